#!/usr/bin/env python3
"""Does the operator stream's L2 footprint matter?  The target's mapping launch with ONE rate class (operators: 1 MB, always
L2-resident) against four (3.9 MB ~ an XCD's 4 MiB of L2): time per (site, class pass).  GPU box, repo root."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
import bench
from comap_amd import engine as E, synthetic as sy

w = bench.WORKLOADS["target"]
parent, blen, lot = sy.random_tree(w["ntaxa"], w["seed"])
dev = torch.device("cuda", 0)
for ncat, nrep in ((4, 1000), (1, 1000), (1, 4000), (2, 2000)):
    mdl = sy.protein_model(0.5, ncat)
    eng = E.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
    ram = 10000
    n = nrep * ram
    aln = torch.empty(nrep * 2 * eng.T * ram, dtype=torch.uint8, device=dev)
    eng.null_simulate_dev(5, 0, nrep, ram, aln)
    stat = torch.empty(n, dtype=torch.float64, device=dev)
    nm = torch.empty(n, dtype=torch.float64, device=dev)
    ts = []
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.null_intra_dev(0, 5, 0, nrep, ram, stat, None, None, nm, supplied=aln)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = min(ts[1:])
    print(f"classes {ncat} sites {2 * n:.2e}: launch {ms:.1f} ms = {ms * 1e6 / (2 * n * ncat):.2f} ns per (site, class pass)", flush=True)
    eng.close(); del aln, stat, nm
    torch.cuda.empty_cache()
