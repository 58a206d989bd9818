#!/usr/bin/env python3
"""Diagnostic build only (stamps.patch applied, library given by COMAP_MI355X_LIB): the four waves of workgroup 7 write
their s_memtime totals over hjoint[0, 0:32] (plain instantiation) / [0, 64:96] (weighted): cycles at the barrier, in the
first half, in the finished tile's reductions and stores, in the second half, in the loop tail, the number of tiles, and
the workgroup's lifetime."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from comap_amd import engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--taxa", type=int, default=256)
ap.add_argument("--gap-columns", type=float, default=0.0)
a = ap.parse_args()
rng = np.random.default_rng(20260103)
T, A, n = a.taxa, 20, 5000
base = rng.integers(0, A, size=(T, 1))
a1 = np.where(rng.random((T, n)) < 0.6, base, rng.integers(0, A, size=(T, n))).astype(np.uint8)
a2 = np.where(rng.random((T, n)) < 0.4, base, rng.integers(0, A, size=(T, n))).astype(np.uint8)
if a.gap_columns > 0:
    for arr in (a1, a2):
        cols = rng.random(arr.shape[1]) < a.gap_columns
        arr[(rng.random(arr.shape) < 0.05) & cols[None, :]] = A
dev = torch.device("cuda:0")
d1, d2 = torch.from_numpy(a1).to(dev), torch.from_numpy(a2).to(dev)
mi = torch.empty((n, n), dtype=torch.float64, device=dev)
hj = torch.empty_like(mi)
h1 = torch.empty(n, dtype=torch.float64, device=dev)
h2 = torch.empty(n, dtype=torch.float64, device=dev)
eng = engine.Engine()
for _ in range(3):
    eng.mi_columns_dev(d1, mi, hj, d2, A, None, h1, h2)
torch.cuda.synchronize()
off = 64 if a.gap_columns >= 1.0 else 0
v = hj[0, off:off + 32].cpu().numpy().reshape(4, 8)
names = ["barrier", "first half", "results", "second half", "tail", "tiles", "lifetime"]
for w in range(4):
    tiles = max(v[w, 5], 1.0)
    print(f"taxa {T} gaps {a.gap_columns} wave {w}: " + ", ".join(f"{names[k]} {v[w, k] / tiles:.0f}" for k in range(5)) +
          f"; tiles {v[w, 5]:.0f}; lifetime / tile {v[w, 6] / tiles:.0f}")
