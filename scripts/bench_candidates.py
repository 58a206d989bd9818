#!/usr/bin/env python3
"""Candidate-group test on one GPU: 100 candidate groups of 2-5 sites from a 500-site protein
alignment on 64 taxa, candidates.null.min = 1000, candidates.null.nb_rep_RAM = 1000 (the reference's defaults).
Prints one JSON line: seconds per analysis, batches, simulated+mapped sites/s, pseudo-groups/s."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from comap_amd import engine, synthetic  # noqa: E402

ntaxa, nobs, ngroups, omega, min_sim, rep_ram = 64, 500, 100, 0.25, 1000, 1000
parent, blen, lot = synthetic.random_tree(ntaxa, 20260103)
mdl = synthetic.protein_model(0.5, 4)
eng = engine.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
aln, _ = eng.simulate(7, 10 ** 7, nobs)
mp = eng.map_sites(aln)
rng = np.random.default_rng(1)
groups = [list(rng.choice(nobs, size=int(rng.integers(2, 6)), replace=False)) for _ in range(ngroups)]
windows = [[(mp["norm"][i] - omega, mp["norm"][i] + omega) for i in g] for g in groups]
observed = eng.group_stats(engine.STAT_CORRELATION, mp["counts"], groups)
eng.candidate_groups(engine.STAT_CORRELATION, windows, [1] * ngroups, observed, 10, rep_ram, 10, 1)      # warm-up
t0 = time.perf_counter()
r = eng.candidate_groups(engine.STAT_CORRELATION, windows, [1] * ngroups, observed, min_sim, rep_ram, 10, 2)
dt = time.perf_counter() - t0
print(json.dumps({"workload": f"candidate groups: {ngroups} groups x min {min_sim} pseudo-groups, repRAM {rep_ram}, {ntaxa} taxa protein",
                  "seconds": dt, "batches": r["batches"], "trials": r["trials"], "simulated_sites_per_s": r["batches"] * rep_ram / dt,
                  "pseudo_groups": int(r["n2"].sum()), "pseudo_groups_per_s": float(r["n2"].sum()) / dt,
                  "groups_completed": int((r["n2"] == min_sim).sum()), "median_pvalue": float(np.median(r["pvalue"]))}))
