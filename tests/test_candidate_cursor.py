"""The bookkeeping of the candidate-group test (host logic of cmx_candidate_groups, no GPU): which simulated site is
handed to which candidate site and which pseudo-groups result, against the member-by-member restatement of
CandidateGroupSet in oracle/candidates.py driven by the same norms."""
import numpy as np
import pytest

from oracle import candidates as ocand
from comap_amd import engine


def _oracle_run(windows, analysable, min_sim, norms, max_trials, monkeypatch):
    made = []
    monkeypatch.setattr(ocand, "group_stat", lambda kind, vectors, params=None: made.append(list(vectors)) or 0.0)
    cs = ocand.CandidateGroupSet(0, windows, analysable, [1.0] * len(windows), min_sim)
    pgs, nb, test = [], 0, True
    while test and nb < len(norms):
        before = len(made)
        test = cs.analyse_simulations([(nb, i) for i in range(norms.shape[1])], norms[nb]) and cs.nb_trials < max_trials
        for vec in made[before:]:
            pgs.append((nb, [i for (_, i) in vec]))
        nb += 1
    return cs, pgs, nb


@pytest.mark.parametrize("seed", range(8))
def test_cursor_matches_restatement(seed, monkeypatch):
    rng = np.random.default_rng(seed)
    G = int(rng.integers(1, 7))
    windows = []
    for _ in range(G):
        m = int(rng.integers(1, 5))
        c = rng.uniform(0.5, 3.0, size=m)
        w = rng.uniform(0.05, 0.6)
        windows.append([(float(x - w), float(x + w)) for x in c])
    analysable = [int(rng.random() < 0.8) for _ in range(G)]
    if not any(analysable):
        analysable[int(rng.integers(0, G))] = 1
    min_sim, max_trials = int(rng.integers(1, 9)), 3
    norms = rng.uniform(0.0, 3.5, size=(12, 23))
    got = engine.debug_candidate_cursor(windows, analysable, min_sim, norms, max_trials)
    cs, pgs, nb = _oracle_run(windows, analysable, min_sim, norms, max_trials, monkeypatch)
    assert list(got["n2"]) == cs.n2 and got["trials"] == cs.nb_trials and got["batches"] == nb
    assert [(b, s) for (_, b, s) in got["pseudo_groups"]] == pgs
    for g, _, s in got["pseudo_groups"]:
        assert len(s) == len(windows[g]) and analysable[g]


def test_cursor_skips_the_first_site_on_its_first_move():
    """the reference's iterator advances before it is read: the very first simulated site is offered to site 1 of group 0,
    not site 0 (CoETools.cpp:904-913)"""
    windows = [[(0.0, 1.0), (5.0, 6.0)], [(0.0, 1.0)]]
    got = engine.debug_candidate_cursor(windows, [1, 1], 1, np.array([[0.5, 0.5, 5.5]]), 5)
    # site 0 (norm 0.5) fits neither (0,1) [5,6] -> goes on to group 1 site 0 and completes it
    assert got["pseudo_groups"][0] == (1, 0, [0])
    assert list(got["n2"]) == [1, 1] and got["pseudo_groups"][1] == (0, 0, [1, 2])
