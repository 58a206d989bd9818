#!/usr/bin/env python3
"""Mica's permutation test (null.method = permutations, CoMap/Mica.cpp:93-118) on one GPU: columns of BASELINE
configs[4]'s shape (256 taxa, protein alphabet, fully resolved), null.max_number_of_permutations = 1000.
Prints one JSON line: pairs/s, permutations/s."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from comap_amd import engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2000)
ap.add_argument("--taxa", type=int, default=256)
ap.add_argument("--alpha", type=int, default=20)
ap.add_argument("--max-perm", type=int, default=1000)
ap.add_argument("--shared", type=float, default=0.3, help="fraction of each column copied from a common ancestor column (0: independent columns)")
ap.add_argument("--gap-columns", type=float, default=0.0, help="fraction of columns that get gaps (10 % of their symbols)")
a = ap.parse_args()
rng = np.random.default_rng(20260103)
T, A, n = a.taxa, a.alpha, a.n
base = rng.integers(0, A, size=(T, 1))
aln = np.where(rng.random((T, n)) < a.shared, base, rng.integers(0, A, size=(T, n))).astype(np.uint8)
if a.gap_columns > 0:
    cols = rng.random(n) < a.gap_columns
    hit = (rng.random((T, n)) < 0.10) & cols[None, :]
    aln[hit] = A                                # unknown: resolveUnknowns = true spreads it over the states
d = torch.from_numpy(aln).cuda()
npairs = n * (n - 1) // 2
pv = torch.empty(npairs, dtype=torch.float64, device="cuda")
npm = torch.empty(npairs, dtype=torch.int32, device="cuda")
eng = engine.Engine()
lib, vp, sz = eng._lib, engine._vp, engine._sz
import ctypes  # noqa: E402


def run(p0, p1):
    eng._check(lib.cmx_mica_permutation_test_dev(eng._ctx, A, T, vp(d), sz(n), sz(n), ctypes.c_uint32(a.max_perm),
                                                 ctypes.c_uint64(7), sz(p0), sz(p1), vp(pv), vp(npm), eng._stream()))


run(0, min(npairs, 20000))
torch.cuda.synchronize()
t0 = time.perf_counter()
run(0, npairs)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
perms = int(npm.sum().item())
print(json.dumps({"workload": f"mica permutation test: {n} columns ({npairs} pairs), {T} taxa, A={A}, max {a.max_perm}, gaps in {a.gap_columns:.0%} of the columns",
                  "seconds": dt, "pairs_per_s": npairs / dt, "permutations": perms, "permutations_per_s": perms / dt,
                  "mean_permutations_per_pair": perms / npairs, "median_pvalue": float(pv.median().item()),
                  "frac_p_below_0.05": float((pv < 0.05).double().mean().item())}))
