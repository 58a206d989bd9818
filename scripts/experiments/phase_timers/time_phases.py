#!/usr/bin/env python3
"""Per-phase s_memtime totals of the mapping kernel's waves (a -DCMX_TIMING build of cmx_kernels.hip, see README.md here).
usage: COMAP_MI355X_LIB=build/abl/libcmx_timing.so python scripts/experiments/phase_timers/time_phases.py cfg4|cfg3|target [observed]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import bench
from comap_amd import engine as E
from comap_amd.pipeline import IntraAnalysis

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
observed = len(sys.argv) > 2 and sys.argv[2] == "observed"
w = bench.WORKLOADS[wl]
parent, blen, lot, mdl, Bk, clamp = bench.build_inputs(w)
eng = E.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], Bk=Bk, clamp_negative=clamp)
print(eng.info())
dev = torch.device("cuda", 0)
aln_h, _ = eng.simulate(w["seed"] + 1, 0, w["nsites"])
d_aln = torch.from_numpy(aln_h).to(dev)
ana = IntraAnalysis(eng, d_aln, w["statistic"], w["nclasses"])
lib = E.load_library()
buf = np.zeros((4096, 16), dtype=np.uint32)
nrep = w["nrep"](1)


def run():
    if observed:
        ana.get_vectors()
    else:
        ana.null_distribution(w["seed"] + 7, 0, nrep, w["rep_ram"])
    torch.cuda.synchronize()


run()
lib.cmx_debug_read_timers(buf.ctypes.data_as(ctypes.c_void_p), 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
lib.cmx_debug_read_timers(buf.ctypes.data_as(ctypes.c_void_p), 1)
ms = e0.elapsed_time(e1)
act = buf[:, 10] > 0
b = buf[act].astype(np.float64)
names = ["issue(dma)", "wait_vm", "mv", "leaf", "rec", "ld", "st", "elementwise"]
tot = b[:, 10]
print(f"{wl} {'observed' if observed else 'null'}: call {ms:.3f} ms (timers on), {act.sum()} active waves; kernel cycles per wave mean {tot.mean():.0f} max {tot.max():.0f}"
      f" -> counter rate {tot.max() / (ms * 1e-3) / 1e6:.0f} MHz if the longest wave spans the call")
nmv, nlf = b[:, 8].mean(), b[:, 9].mean()
print(f"ops per wave: products {nmv:.0f} leaf {nlf:.0f}")
acc = 0.0
for i, n in enumerate(names):
    f = b[:, i].mean() / tot.mean()
    acc += f
    per = b[:, i].mean() / (nmv + nlf)
    print(f"  {n:12s} {100 * f:5.1f} %   {per:7.1f} cycles per operator op")
print(f"  {'walk + rest':12s} {100 * (1 - acc):5.1f} %   {(1 - acc) * tot.mean() / (nmv + nlf):7.1f} cycles per operator op")
print(f"  per product: mv {b[:, 2].mean() / nmv:.0f} cycles; per leaf op: {b[:, 3].mean() / max(nlf, 1):.0f} cycles; total per op {tot.mean() / (nmv + nlf):.0f}")
