#!/usr/bin/env python3
"""How long does the observed-alignment mapping take alone, and next to a null launch of a given size?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from comap_amd import engine as E
from comap_amd.pipeline import IntraAnalysis
w = bench.WORKLOADS["target"]
parent, blen, lot, mdl, Bk, clamp = bench.build_inputs(w)
eng = E.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], device=0)
dev = torch.device("cuda:0")
aln_h, _ = eng.simulate(w["seed"] + 1, 0, w["nsites"])
ana = IntraAnalysis(eng, torch.from_numpy(aln_h).to(dev), "Correlation", 10)
def t_ms(f, n=3):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("observed mapping alone (10000 sites): %.3f ms" % t_ms(ana.get_vectors))
side = torch.cuda.Stream(device=dev); main = torch.cuda.current_stream()
for nrep in (125, 1000):
    print("null alone, %d replicates: %.3f ms" % (nrep, t_ms(lambda: ana.null_distribution(7, 0, nrep, 10000), 2)))
    def both():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ana.get_vectors()
        ana.null_distribution(7, 0, nrep, 10000)
        main.wait_stream(side)
    print("observed || null, %d replicates: %.3f ms" % (nrep, t_ms(both, 2)))
    def seq():
        ana.get_vectors()
        ana.null_distribution(7, 0, nrep, 10000)
    print("observed then null, %d replicates: %.3f ms" % (nrep, t_ms(seq, 2)))
