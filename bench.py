#!/usr/bin/env python3
"""Benchmark of the substitution-mapping + pairwise-coevolution hot path on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input (DESIGN.md section "Measurement"):
  map the observed alignment -> parametric-bootstrap null for this rank's replicates (simulate, re-map twice,
  score) -> [N > 1: one RCCL all-gather of the null] -> all-pairs statistic on MFMA -> p-values.
Metric (BASELINE.json): site-pair coevolution statistics/s incl. null sims = (observed pairs + null pairs of all
ranks) / time, inputs resident in HBM.  Default workload = BASELINE configs[1] alignment (2 000 sites x 64 taxa
protein, JTT92+G4, correlation) with configs[2]'s null sharded as 125 replicates x 2 000 per GPU (weak scaling:
8 GPUs = the 1 000 replicates of configs[2]).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: ntaxa, nsites, nstates, tree seed, statistic, replicates per GPU, rep_ram, norm classes
    "cfg3": dict(ntaxa=64, nsites=2000, nstates=20, seed=20260101, statistic="Correlation", rep_per_gpu=125,
                 rep_ram=2000, nclasses=10,
                 desc="2000x64 protein JTT92+G4(a=0.5), correlation, null 125 rep/GPU x 2000 (cfg3 sharded)"),
    "target": dict(ntaxa=64, nsites=10000, nstates=20, seed=20260101, statistic="Correlation", rep_per_gpu=1000,
                   rep_ram=10000, nclasses=10,
                   desc="north-star target: 10000x64 protein, correlation, 1000 null replicates x 10000 per GPU"),
    "cfg4": dict(ntaxa=256, nsites=10000, nstates=4, seed=20260102, statistic="Compensation", rep_per_gpu=12,
                 rep_ram=10000, nclasses=10,
                 desc="10000x256 DNA GTR+G4, compensation (W=idx[y]-idx[x]), null 12 rep/GPU x 10000"),
}
FP64_PEAK_TFLOPS = 78.6  # MI355X public spec, vector == matrix fp64 (the microarch guide has no fp64 row);
# measured here (scripts/ubench_f64.hip): v_mfma_f64_16x16x4 72-75, v_fma_f64 55-59 TFLOP/s


def build_inputs(w):
    from comap_amd import synthetic as sy
    parent, blen, lot = sy.random_tree(w["ntaxa"], w["seed"])
    if w["nstates"] == 20:
        mdl = sy.protein_model(0.5, 4)
        Bk, clamp = None, True
    else:
        mdl = sy.dna_model(0.5, 4)
        Bk = sy.weighted_register(mdl["Q"], sy.compensation_weights_dna())[None] if w["statistic"] == "Compensation" else None
        clamp = Bk is None
    return parent, blen, lot, mdl, Bk, clamp


def flops_per_site(B, C, S, K, nn, ni_nonroot, nleaves):
    algorithmic = 7.0 * B * C * S * S                      # F_map = 7 B C S^2 (K = 1)
    executed = C * (ni_nonroot * 2.0 * S * S * (3 + K)     # inside, recomputed sibling message, J.D, outside
                    + nleaves * 2.0 * S * K + nn * 3.0 * S)
    return algorithmic, executed


def cpu_baseline(w, parent, blen, lot, mdl, Bk, clamp, n_gpu_units, world):
    """Oracle (CPU restatement, single thread, reference loop structure) on a bounded sample, extrapolated to the
    benchmark's unit mix.  kind = "port": the reference itself cannot be built here (Bio++ absent)."""
    import oracle
    kind = {"Correlation": 0, "Compensation": 1}[w["statistic"]]
    om = oracle.Model(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], Bk=Bk, nonneg=clamp)
    n_obs = min(w["nsites"], 2000)
    nrep_s, ram_s = 16, min(w["rep_ram"], 2000)   # ~64 000 sites re-mapped: 10-30 s of host work
    aln, _ = oracle.simulate(om, w["seed"] + 1, 0, n_obs)
    t0 = time.perf_counter()
    m = oracle.map_sites(om, aln)
    t_map = time.perf_counter() - t0
    t0 = time.perf_counter()
    st = oracle.pair_stats_intra(kind, m["counts"])
    t_pairs = time.perf_counter() - t0
    t0 = time.perf_counter()
    nl = oracle.null_intra(om, kind, 1, 0, nrep_s, ram_s)
    t_null = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.intra_pvalues(st, m["norm"], w["nclasses"], nl["stat"], nl["nmin"])
    t_pv = time.perf_counter() - t0
    # extrapolate the sample to one benchmark step of ONE rank's share (CPU has no ranks: total work / 1 thread)
    scale_obs = w["nsites"] / n_obs
    pairs_obs = w["nsites"] * (w["nsites"] - 1) / 2
    null_pairs_total = world * w["rep_per_gpu"] * w["rep_ram"]
    t_full = (t_map * scale_obs + t_pairs * scale_obs ** 2 + t_null * null_pairs_total / (nrep_s * ram_s)
              + t_pv * scale_obs ** 2 * null_pairs_total / (nrep_s * ram_s))   # linear-scan p-values scale with nsim
    value = (pairs_obs + null_pairs_total) / t_full
    sample = (f"oracle/oracle.c -O2, 1 thread: map {n_obs} sites {t_map:.2f}s, {n_obs * (n_obs - 1) // 2} pair stats "
              f"{t_pairs:.2f}s, null {nrep_s} rep x {ram_s} (={2 * nrep_s * ram_s} sites re-mapped) {t_null:.2f}s, "
              f"linear-scan p-values {t_pv:.2f}s; extrapolated to the step's unit mix ({t_full:.0f}s of CPU work)")
    return dict(value=value, unit="site-pair statistics/s", cores=1, kind="port", sample=sample)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from comap_amd import engine as E
    from comap_amd.pipeline import IntraAnalysis

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # rehearsal on a one-GPU box (CMX_BENCH_REHEARSAL=1): all ranks share device 0 and the exchange runs over gloo;
    # it checks the N > 1 control flow only, its throughput means nothing
    rehearsal = os.environ.get("CMX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    w = WORKLOADS[args.workload]
    parent, blen, lot, mdl, Bk, clamp = build_inputs(w)
    eng = E.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], Bk=Bk, clamp_negative=clamp,
                   device=local_rank)
    info = eng.info()
    # observed alignment: simulated by the engine's own simulator (seed + 1), resident in HBM before timing starts
    aln_h, _ = eng.simulate(w["seed"] + 1, 0, w["nsites"])
    d_aln = torch.from_numpy(aln_h).to(dev)
    ana = IntraAnalysis(eng, d_aln, w["statistic"], w["nclasses"])
    from comap_amd.distributed import gather_null, replicate_shard
    ram = w["rep_ram"]
    nrep_total = w["rep_per_gpu"] * world                  # weak scaling: fixed replicates per GPU
    rep_begin, rep_end = replicate_shard(rank, world, nrep_total)
    n_local = (rep_end - rep_begin) * ram
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    side = torch.cuda.Stream(device=dev)   # observed-alignment mapping overlaps the null kernel (independent work)
    main = torch.cuda.current_stream()

    def step(i, timed):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ana.get_vectors()
        if timed:
            ev[i][0].record()
        nb = ana.null_distribution(w["seed"] + 7, rep_begin, rep_end, ram)
        if timed:
            ev[i][1].record()
        main.wait_stream(side)
        # the path's one exchange: every rank needs the merged null before p-values (one RCCL all-gather)
        ns, nm = gather_null(nb["stat"], nb["nmin"], nrep_total, ram)
        ana.compute_intra_stats(ns, nm)

    for i in range(args.warmup):
        step(i, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cpu") if rehearsal else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    pairs_obs = w["nsites"] * (w["nsites"] - 1) // 2
    units_per_step = pairs_obs + nrep_total * ram
    ms_per_step = 1e3 * elapsed / args.steps
    value = units_per_step * args.steps / elapsed

    # roofline of the dominant kernel (map_kernel<S, null>): HIP events on the launch stream
    null_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    nn = len(parent)
    nleaves = len(lot)
    ni_nonroot = nn - nleaves - 1
    alg, exe = flops_per_site(eng.B, eng.C, eng.S, eng.K, nn, ni_nonroot, nleaves)
    sites_per_launch = 2 * n_local
    achieved = sites_per_launch * alg / (null_ms * 1e-3) / 1e12
    roofline = dict(bound="mfma", kernel=f"map_kernel<{eng.S},null>", achieved=achieved, peak=FP64_PEAK_TFLOPS,
                    unit="TFLOP/s", frac=achieved / FP64_PEAK_TFLOPS, traffic=None,
                    launch_ms=null_ms, sites_per_launch=sites_per_launch, flops_per_site_algorithmic=alg,
                    flops_per_site_executed=exe,
                    achieved_executed=sites_per_launch * exe / (null_ms * 1e-3) / 1e12)
    traffic_file = os.path.join(ROOT, "profiles", "traffic_r01.json")
    if os.path.exists(traffic_file):
        try:
            roofline["traffic"] = json.load(open(traffic_file)).get(args.workload)
        except Exception:
            pass

    out = dict(metric="site-pair coevolution statistics/s (incl. null sims)", value=value,
               unit="site-pair statistics/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=ms_per_step, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64",
               data="synthetic",
               config=dict(workload=f"{args.workload}: {w['desc']}", observed_pairs=pairs_obs,
                           null_pairs_per_gpu=n_local, sites_mapped_per_step_per_gpu=w["nsites"] + 2 * n_local,
                           parallelism=f"null replicates sharded x{world}, one RCCL all-gather" if world > 1 else "single GPU",
                           cu_count=info["cu_count"], mapping_waves=info["waves"]),
               roofline=roofline)
    if rank == 0:
        if args.no_cpu_baseline or world > 1:   # CPU baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = None
        else:
            out["cpu_baseline"] = cpu_baseline(w, parent, blen, lot, mdl, Bk, clamp, units_per_step, world)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
