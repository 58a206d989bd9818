#!/usr/bin/env python3
"""Clustering null on one GPU: ClusterTools::computeGlobalDistanceDistribution for an alignment of
BASELINE configs[2]'s shape (2 000 sites x 64 taxa, protein, 4 rate classes): per replicate simulate 2 000 sites, map
them, build the 2 000 x 2 000 distance matrix and cluster it.  Prints one JSON line: replicates/s end to end, the
agglomeration alone (batch of matrices resident in HBM), and scipy's linkage on one matrix on the host for context."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from comap_amd import engine, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=2000)
ap.add_argument("--taxa", type=int, default=64)
ap.add_argument("--reps", type=int, default=256)
ap.add_argument("--link", default="complete")
ap.add_argument("--dist", default="cor")
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--no-scipy", action="store_true")
a = ap.parse_args()
link, dist = engine.LINK_BY_NAME[a.link], engine.DIST_BY_NAME[a.dist]
parent, blen, lot = synthetic.random_tree(a.taxa, 20260103)
mdl = synthetic.protein_model(0.5, 4)
eng = engine.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
eng.cluster_null(dist, link, 1, 0, min(a.reps, 8), a.sites)          # warm-up: scratch allocation
t0 = time.perf_counter()
for s in range(a.steps):
    r = eng.cluster_null(dist, link, 1, s * a.reps, (s + 1) * a.reps, a.sites)
t_null = (time.perf_counter() - t0) / a.steps

# agglomeration alone on resident matrices
n, batch = a.sites, min(a.reps, 256)
rng = np.random.default_rng(3)
x = torch.from_numpy(rng.normal(size=(batch, n, 8))).cuda()
d = torch.cdist(x, x)
d = (d + d.transpose(1, 2)) / 2
merge = torch.empty((batch, n - 1, 2), dtype=torch.int32, device="cuda")
dmax = torch.empty((batch, n - 1), dtype=torch.float64, device="cuda")
size = torch.empty((batch, n - 1), dtype=torch.int32, device="cuda")
lib, vp, sz = eng._lib, engine._vp, engine._sz
work = d.clone()
eng._check(lib.cmx_hclust_dev(eng._ctx, link, vp(work), sz(n), sz(n), sz(batch), vp(merge), vp(dmax), vp(size), eng._stream()))
torch.cuda.synchronize()
work.copy_(d)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
eng._check(lib.cmx_hclust_dev(eng._ctx, link, vp(work), sz(n), sz(n), sz(batch), vp(merge), vp(dmax), vp(size), eng._stream()))
e1.record()
torch.cuda.synchronize()
hc_ms = e0.elapsed_time(e1)
out = {"workload": f"clustering null: {a.reps} replicates x {a.sites} sites x {a.taxa} taxa, protein, {a.dist}/{a.link}",
       "replicates_per_s": a.reps / t_null, "s_per_step": t_null,
       "hclust_batch": batch, "hclust_ms": hc_ms, "hclust_us_per_join_per_matrix": hc_ms * 1e3 / (n - 1),
       "hclust_matrices_per_s": batch / hc_ms * 1e3,
       "monotone": bool(np.all(np.diff(r["dmax"], axis=1) >= -1e-12)), "root_size_ok": bool(np.all(r["size"][:, -1] == a.sites))}
if not a.no_scipy:
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import squareform
    dh = d[0].cpu().numpy()
    np.fill_diagonal(dh, 0.0)
    t0 = time.perf_counter()
    linkage(squareform(dh, checks=False), a.link)
    out["scipy_linkage_one_matrix_s"] = time.perf_counter() - t0
print(json.dumps(out))
