"""N > 1 path on CPU: two gloo ranks each produce their replicate shard of the null (through the oracle, since there
is no GPU here), all-gather it with comap_amd.distributed, and must reproduce the single-process null bit for bit in
the reference's replicate order (sharding invariance comes from the counter-based RNG keyed by global indices)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, nrep, rep_ram, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from comap_amd.distributed import gather_null, replicate_shard
    from conftest import make_case
    case = make_case(8, 4, 4, 77)
    om = oracle.Model(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    b, e = replicate_shard(rank, world, nrep)
    loc = oracle.null_intra(om, 0, 99, b, e, rep_ram)
    stat, nmin = gather_null(torch.from_numpy(loc["stat"]), torch.from_numpy(loc["nmin"]), nrep, rep_ram)
    if rank == 0:
        full = oracle.null_intra(om, 0, 99, 0, nrep, rep_ram)
        q.put((np.array_equal(stat.numpy(), full["stat"], equal_nan=True),
               np.array_equal(nmin.numpy(), full["nmin"]), int(stat.numel())))
    dist.barrier()
    dist.destroy_process_group()


def test_replicate_shard_is_a_balanced_partition():
    from comap_amd.distributed import replicate_shard
    for nrep in (1, 7, 125, 1000):
        for world in (1, 2, 3, 8):
            parts = [replicate_shard(r, world, nrep) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == nrep
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("nrep", [4, 5])       # even and uneven shards
def test_two_rank_gloo_null_matches_single_process(nrep):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + nrep) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nrep, 10, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_stat, ok_nmin, n = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_stat and ok_nmin and n == nrep * 10


def _worker_shards(rank, world, port, nitems, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from comap_amd.distributed import gather_shards, replicate_shard
    import oracle
    rng = np.random.default_rng(0)
    aln = rng.integers(0, 4, size=(20, 7)).astype(np.uint8)
    b, e = replicate_shard(rank, world, nitems)
    pv, npm = oracle.mica_permutation_test(aln, 4, 50, 11, b, e)          # this rank's pairs of Mica's permutation test
    loc = torch.stack([torch.from_numpy(pv), torch.from_numpy(npm.astype(np.float64))], dim=1)
    full = gather_shards(loc, nitems)
    if rank == 0:
        pv0, npm0 = oracle.mica_permutation_test(aln, 4, 50, 11)
        q.put((np.array_equal(full[:, 0].numpy(), pv0), np.array_equal(full[:, 1].numpy(), npm0.astype(np.float64))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_pair_shards_match_single_process():
    """pairs of Mica's permutation test sharded over two ranks (uneven: 21 pairs) and gathered in pair order"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_shards, args=(r, 2, port, 21, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_pv, ok_n = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_pv and ok_n


def test_row_shard_partitions_the_upper_triangle_by_pair_count():
    from comap_amd.distributed import row_shard
    from comap_amd.pipeline import sum_pairs
    for n in (2, 5, 129, 2000, 10000):
        for world in (1, 2, 3, 8):
            sh = [row_shard(r, world, n) for r in range(world)]
            assert sh[0][0] == 0 and sh[-1][1] == n and all(sh[i][1] == sh[i + 1][0] for i in range(world - 1))
            pc = [sum_pairs(n, a, b) for a, b in sh]
            assert sum(pc) == n * (n - 1) // 2
            if n >= 2000:
                assert max(pc) <= 1.01 * sum(pc) / world


def _worker_rows_and_mica(rank, world, port, q):
    """Observed stage sharded over two ranks, through the oracle: each rank computes the statistics.txt rows of its
    row_shard range after the null all-gather; rank order concatenation must equal the single-process rows.  Then Mica's
    rectangle: row blocks of MI + ONE all-reduce of the column sums must reproduce the single-process averages."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from comap_amd.distributed import combine_mica_sums, gather_null, gather_shards, replicate_shard, row_shard
    from conftest import make_case
    case = make_case(9, 40, 20, 5)
    om = oracle.Model(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    m = oracle.map_sites(om, case["aln"])
    nrep, ram, n = 6, 25, 40
    b, e = replicate_shard(rank, world, nrep)
    loc = oracle.null_intra(om, 0, 5, b, e, ram)
    ns, nm = gather_null(torch.from_numpy(loc["stat"]), torch.from_numpy(loc["nmin"]), nrep, ram)
    st = oracle.pair_stats_intra(0, m["counts"])
    pv, nsim = oracle.intra_pvalues(st, m["norm"], 4, ns.numpy(), nm.numpy())
    r0, r1 = row_shard(rank, world, n)
    rows = [(i, j, st[i, j], pv[i, j], nsim[i, j]) for i in range(r0, r1) for j in range(i + 1, n)]
    mine = torch.tensor(rows, dtype=torch.float64).reshape(-1, 5)
    # (rows stay on their rank in production; gathered here only to compare)
    sizes = [sum(n - 1 - i for i in range(*row_shard(r, world, n))) for r in range(world)]
    pad = torch.full((max(sizes), 5), float("nan"), dtype=torch.float64)
    pad[: mine.shape[0]] = mine
    allr = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(allr, pad)
    cat = torch.cat([allr[r][: sizes[r]] for r in range(world)]).numpy()
    # Mica rectangle
    rng = np.random.default_rng(3)
    a1 = rng.integers(0, 20, size=(30, 11)).astype(np.uint8)
    a2 = rng.integers(0, 20, size=(30, 7)).astype(np.uint8)
    mb, me = replicate_shard(rank, world, 11)
    blk = oracle.mi_columns(a1[:, mb:me], a2, 20)["mi"]
    col_sum, tot = combine_mica_sums(torch.from_numpy(blk.sum(axis=0)))
    if rank == 0:
        nl = oracle.null_intra(om, 0, 5, 0, nrep, ram)
        pv0, ns0 = oracle.intra_pvalues(st, m["norm"], 4, nl["stat"], nl["nmin"])
        ref = np.array([(i, j, st[i, j], pv0[i, j], ns0[i, j]) for i in range(n) for j in range(i + 1, n)])
        full = oracle.mi_columns(a1, a2, 20)["mi"]
        q.put((np.array_equal(cat, ref, equal_nan=True), bool(np.allclose(col_sum.numpy(), full.sum(axis=0), rtol=1e-13)),
               bool(abs(float(tot) - full.sum()) < 1e-12 * abs(full.sum()))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_observed_rows_and_mica_rectangle():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_rows_and_mica, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_rows, ok_cols, ok_tot = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_rows and ok_cols and ok_tot


class _OracleColumns:
    """stands where the engine stands in mica_rectangle (same method, same argument order), columns through the oracle:
    there is no GPU in the CPU suite, and what is under test is the sharding, the skipped empty block and the averages"""

    def mi_columns_dev(self, blk, mi, hj, other, nalpha, masks, h1, h2):
        import oracle
        assert blk.shape[1] > 0, "an empty block must not reach the engine"
        r = oracle.mi_columns(blk.numpy(), other.numpy(), nalpha)
        mi.copy_(torch.from_numpy(r["mi"])); hj.copy_(torch.from_numpy(r["hjoint"]))
        h1.copy_(torch.from_numpy(r["h1"])); h2.copy_(torch.from_numpy(r["h2"]))


def _worker_rectangle(rank, world, port, n1, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from comap_amd.distributed import mica_rectangle
    rng = np.random.default_rng(11)
    a1 = rng.integers(0, 20, size=(25, n1)).astype(np.uint8)
    a2 = rng.integers(0, 20, size=(25, 6)).astype(np.uint8)
    t1, t2 = torch.from_numpy(a1), torch.from_numpy(a2)
    intra = mica_rectangle(_OracleColumns(), t1, None, 20)
    rect = mica_rectangle(_OracleColumns(), t1, t2, 20)
    parts = [None] * world
    dist.all_gather_object(parts, (intra["rows"], intra["row_mean"].numpy(), rect["row_mean"].numpy()))
    if rank == 0:
        full = oracle.mi_columns(a1, a1, 20)["mi"]
        avg, fullavg = oracle.mica_average_mi(full)          # Mica.cpp:346-363
        fr = oracle.mi_columns(a1, a2, 20)["mi"]
        ok = [np.allclose(np.concatenate([p[1] for p in parts]), avg, rtol=1e-13, atol=1e-15),
              np.allclose(intra["col_mean"].numpy(), avg, rtol=1e-13, atol=1e-15),
              abs(float(intra["full_mean"]) - fullavg) <= 1e-13 * abs(fullavg),
              np.allclose(np.concatenate([p[2] for p in parts]), fr.mean(axis=1), rtol=1e-13),
              np.allclose(rect["col_mean"].numpy(), fr.mean(axis=0), rtol=1e-13),
              abs(float(rect["full_mean"]) - fr.mean()) <= 1e-13 * fr.mean(),
              [p[0] for p in parts][0][0] == 0 and [p[0] for p in parts][-1][1] == n1]
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n1", [(2, 9), (3, 2)])     # (3, 2): one rank has no column of its own
def test_gloo_mica_rectangle_shards_and_reference_averages(world, n1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker_rectangle, args=(r, world, port, n1, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok), ok
