#!/usr/bin/env python3
"""The observed pair stage alone (operand preparation + Gram + record pass for all pairs, with a null already on the device):
usage: python scripts/time_pair_stage.py [cfg4|target|cfg3]   (GPU box, repo root; COMAP_MI355X_LIB selects the build)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from comap_amd import engine as E
from comap_amd.pipeline import IntraAnalysis
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
w = bench.WORKLOADS[wl]
parent, blen, lot, mdl, Bk, clamp = bench.build_inputs(w)
eng = E.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], Bk=Bk, clamp_negative=clamp)
dev = torch.device("cuda:0")
aln_h, _ = eng.simulate(w["seed"] + 1, 0, w["nsites"])
ana = IntraAnalysis(eng, torch.from_numpy(aln_h).to(dev), w["statistic"], w["nclasses"])
ana.get_vectors()
nb = ana.null_distribution(w["seed"] + 7, 0, min(w["nrep"](1), 50), w["rep_ram"])
ns, nm = nb["stat"].clone(), nb["nmin"].clone()
def run():
    ana.compute_intra_compact(ns, nm)
run(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print(f"{wl}: pair stage {min(ts):.3f} ms (best of 5), median {sorted(ts)[2]:.3f}")
