#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU: Mica column MI of two alignments of 5000 + 5000 columns, 256 taxa, protein alphabet,
all 25e6 cross pairs (seed 20260103; synthetic columns drawn around a shared ancestral column so that MI is
not trivially zero).  Prints one JSON line: pairs/s, ms, algorithmic one-hot Gram rate 2*A^2*T flop/pair."""
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
ap.add_argument("--n1", type=int, default=5000)
ap.add_argument("--n2", type=int, default=5000)
ap.add_argument("--taxa", type=int, default=256)
ap.add_argument("--alpha", type=int, default=20)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--gap-columns", type=float, default=0.0, help="fraction of columns that carry gaps (code = alpha), 5 %% of their rows")
a = ap.parse_args()
rng = np.random.default_rng(20260103)
T, A = a.taxa, a.alpha
base = rng.integers(0, A, size=(T, 1))
a1 = np.where(rng.random((T, a.n1)) < 0.6, base, rng.integers(0, A, size=(T, a.n1))).astype(np.uint8)
a2 = np.where(rng.random((T, a.n2)) < 0.4, base, rng.integers(0, A, size=(T, a.n2))).astype(np.uint8)
if a.gap_columns > 0:
    for arr in (a1, a2):
        cols = rng.random(arr.shape[1]) < a.gap_columns
        mask = (rng.random(arr.shape) < 0.05) & cols[None, :]
        arr[mask] = A
dev = torch.device("cuda:0")
d1, d2 = torch.from_numpy(a1).to(dev), torch.from_numpy(a2).to(dev)
mi = torch.empty((a.n1, a.n2), dtype=torch.float64, device=dev)
hj = torch.empty_like(mi)
h1 = torch.empty(a.n1, dtype=torch.float64, device=dev)
h2 = torch.empty(a.n2, dtype=torch.float64, device=dev)
eng = engine.Engine()
eng.mi_columns_dev(d1, mi, hj, d2, A, None, h1, h2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps):
    eng.mi_columns_dev(d1, mi, hj, d2, A, None, h1, h2)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.steps
pairs = a.n1 * a.n2
# APC needs the row means of MI over the concatenation (Mica.cpp:349-363): one reduction, shown for completeness
apc_ms0 = time.perf_counter()
rm, cm, mm = mi.mean(dim=1), mi.mean(dim=0), mi.mean()
torch.cuda.synchronize()
apc_ms = (time.perf_counter() - apc_ms0) * 1e3
ident = float((mi - (h1[:, None] + h2[None, :] - hj)).abs().max())
print(json.dumps({"workload": f"mica {a.n1}x{a.n2} columns, {T} taxa, A={A}", "pairs": pairs, "ms": ms,
                  "pairs_per_s": pairs / ms * 1e3, "onehot_gram_TFLOPs_algorithmic": pairs * 2 * A * A * T / ms / 1e9,
                  "output_GBps": pairs * 16 / ms / 1e6, "row_means_ms": apc_ms, "max_identity_residual": ident,
                  "mi_mean": float(mm)}))
