#!/usr/bin/env python3
"""does the mapping kernel's HBM read traffic per site depend on how long a launch runs?
Runs the north-star target's null (1 000 replicates x 10 000 sites, 2e7 mapped sites) once as ONE mapping launch and once as
NSPLIT back-to-back launches (waves re-phase at every launch), alignments simulated beforehand.  Under
`rocprofv3 --pmc FETCH_SIZE ...` / `--pmc TCC_HIT_sum TCC_MISS_sum` / `--kernel-trace` the per-dispatch rows of the mapping
kernel tell the two apart (scripts/summarize_split_null.py).  usage: profile_split_null.py [nsplit] [workload]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from comap_amd import engine as E  # noqa: E402

nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 40
wl = sys.argv[2] if len(sys.argv) > 2 else "target"
w = bench.WORKLOADS[wl]
parent, blen, lot, mdl, Bk, clamp = bench.build_inputs(w)
eng = E.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], Bk=Bk, clamp_negative=clamp, device=0)
dev = torch.device("cuda:0")
nrep, ram = w["nrep"](1), w["rep_ram"]
kind = E.STAT_BY_NAME[w["statistic"]]
aln = torch.empty(nrep * 2 * eng.T * ram, dtype=torch.uint8, device=dev)
eng.null_simulate_dev(w["seed"] + 7, 0, nrep, ram, aln)
stat = torch.empty(nrep * ram, dtype=torch.float64, device=dev)
nmin = torch.empty(nrep * ram, dtype=torch.float64, device=dev)
per = 2 * eng.T * ram


def run(parts):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step = -(-nrep // parts)
    for r0 in range(0, nrep, step):
        r1 = min(nrep, r0 + step)
        eng.null_intra_dev(kind, w["seed"] + 7, r0, r1, ram, stat[r0 * ram:r1 * ram], nmin=nmin[r0 * ram:r1 * ram],
                           supplied=aln[r0 * per:r1 * per])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


run(1)   # warm-up (first touch of the workspace)
one = run(1)
ref = stat.clone()
many = run(nsplit)
assert torch.equal(torch.nan_to_num(ref), torch.nan_to_num(stat)), "splitting the launch changed the null"
print(f"workload {wl}: {2 * nrep * ram} sites; 1 launch {one:.1f} ms; {nsplit} launches {many:.1f} ms")
