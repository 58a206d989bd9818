#!/usr/bin/env python3
"""One GPU doing what ONE rank of an 8-GPU target run does between the barriers (no communication): the null for its
125 replicates, then statistic + p-values + rows of its row range against the merged 10^7-entry null.  Tells which part
of a step does not shrink with the number of GPUs."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from comap_amd import engine as E
from comap_amd.pipeline import IntraAnalysis
from comap_amd.distributed import replicate_shard, row_shard
w = bench.WORKLOADS["target"]
parent, blen, lot, mdl, Bk, clamp = bench.build_inputs(w)
eng = E.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], device=0)
dev = torch.device("cuda:0")
aln_h, _ = eng.simulate(w["seed"] + 1, 0, w["nsites"])
ana = IntraAnalysis(eng, torch.from_numpy(aln_h).to(dev), "Correlation", w["nclasses"])
ana.get_vectors()
def t_ms(f, n=3):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
world, ram, nrep = 8, w["rep_ram"], 1000
for rank in (0, 7):
    rb, re = replicate_shard(rank, world, nrep)
    r0, r1 = row_shard(rank, world, w["nsites"])
    nb = ana.null_distribution(w["seed"] + 7, rb, re, ram)
    # the merged null: this rank's 125 replicates tiled 8 times (same size and spread as the real one)
    ns, nm = nb["stat"].repeat(world), nb["nmin"].repeat(world)
    print("rank %d: null %d replicates %.2f ms | rows [%d, %d) of the pair loop with a %d-entry null %.2f ms | observed mapping %.2f ms"
          % (rank, re - rb, t_ms(lambda: ana.null_distribution(w["seed"] + 7, rb, re, ram), 2), r0, r1, ns.numel(),
             t_ms(lambda: ana.compute_intra_rows(ns, nm, r0, r1), 2), t_ms(ana.get_vectors, 2)))
