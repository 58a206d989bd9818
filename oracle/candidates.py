"""TEST INFRASTRUCTURE ONLY: CPU restatement of the candidate-group test, following the reference
member by member: CandidateGroupSet (CoMap/CoETools.h:139-300), nextCandidateSite (CoETools.cpp:900-933),
analyseSimulations (:950-1000), addSimulatedSite (:1004-1038), computePValuesForCandidateGroups (:1042-1087), group
statistics (CoMap/Statistics.h:121-133, :267-294).  Never imported by comap_amd/.

PARITY UNPINNED: the reference ships no output of this analysis; and its random stream (Bio++ RandomTools) is not
reproducible here, so simulated sites come from the shared counter RNG (batch t = sites t*repRAM ..)."""
import numpy as np

import oracle


def group_stat(kind, vectors, params=None):
    """vectors: list of [B][K] arrays"""
    if kind == oracle.ST_COMPENSATION:
        tot = [v.sum(axis=1) for v in vectors]
        return 1.0 - np.sqrt((np.sum(tot, axis=0) ** 2).sum()) / sum(np.sqrt((t * t).sum()) for t in tot)
    mini = np.inf
    for i in range(1, len(vectors)):
        for j in range(i):
            val = oracle.stat_pair(kind, vectors[i], vectors[j], params)
            if val < mini:
                mini = val
    return mini


class CandidateGroupSet:
    def __init__(self, kind, windows, analysable, observed, min_sim, params=None):
        self.kind, self.params = kind, params
        self.windows = windows                       # [group][site] -> (normMin, normMax)
        self.analysable = list(analysable)
        self.observed = list(observed)
        self.min_sim = min_sim
        self.simulations = [[[] for _ in g] for g in windows]
        self.n1 = [0] * len(windows)
        self.n2 = [0] * len(windows)
        self.near_ties = [0] * len(windows)
        self.nb_completed = 0
        self.nb_analysable = sum(1 for a in analysable if a)
        self.nb_trials = 0
        self.group_pos = 0
        self.site_pos = 0

    def size(self):
        return len(self.windows)

    def next_candidate_site(self):
        if self.nb_completed == self.size():
            raise RuntimeError("enough simulations")
        if self.n2[self.group_pos] < self.min_sim:
            self.site_pos += 1
            if self.site_pos >= len(self.windows[self.group_pos]):
                self.group_pos += 1
                if self.group_pos >= self.size():
                    self.group_pos = 0
                self.site_pos = 0
        start_search = self.group_pos
        if self.n2[self.group_pos] >= self.min_sim or not self.analysable[self.group_pos]:
            while self.n2[self.group_pos] >= self.min_sim or not self.analysable[self.group_pos]:
                self.group_pos += 1
                if self.group_pos >= self.size():
                    self.group_pos = 0
                if self.group_pos == start_search:
                    raise RuntimeError("no more site to complete")
            self.site_pos = 0
        return (self.group_pos, self.site_pos)

    def add_simulated_site(self, g, s, v):
        group = self.simulations[g]
        group[s].append(v)
        if any(len(q) == 0 for q in group):
            return False
        vectors = [q.pop(0) for q in group]
        self.n2[g] += 1
        st = group_stat(self.kind, vectors, self.params)
        if st >= self.observed[g]:
            self.n1[g] += 1
        if abs(st - self.observed[g]) <= 1e-6 * max(1.0, abs(self.observed[g])):
            self.near_ties[g] += 1       # decided by the last bits: implementations may legitimately differ here
        if self.n2[g] == self.min_sim:
            self.nb_completed += 1
        return True

    def analyse_simulations(self, counts, norms):
        test, test_free = True, True
        i = 0
        while test and i < len(norms):
            first, test_norm = True, False
            while test and not test_norm:
                pos = self.next_candidate_site()
                if first:
                    start, first = pos, False
                elif pos == start:
                    break
                lo, hi = self.windows[pos[0]][pos[1]]
                test_norm = lo <= norms[i] <= hi
                if test_norm:
                    if self.add_simulated_site(pos[0], pos[1], counts[i]):
                        test_free = False
                    if self.nb_completed == self.nb_analysable:
                        test = False
            i += 1
        if test_free:
            self.nb_trials += 1
        self.simulations = [[[] for _ in g] for g in self.windows]
        return test


def candidate_groups(model, kind, windows, analysable, observed, min_sim, rep_ram, max_trials, seed, max_batches=0, params=None):
    cs = CandidateGroupSet(kind, windows, analysable, observed, min_sim, params)
    test, nb = True, 0
    while test and (max_batches == 0 or nb < max_batches):
        aln, _ = oracle.simulate(model, seed, nb * rep_ram, rep_ram)
        mp = oracle.map_sites(model, aln)
        nb += 1
        test = cs.analyse_simulations(mp["counts"], mp["norm"]) and cs.nb_trials < max_trials
    return dict(n1=np.array(cs.n1), n2=np.array(cs.n2), trials=cs.nb_trials, batches=nb, near_ties=np.array(cs.near_ties))
