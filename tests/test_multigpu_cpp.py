"""include/comap_mi355x_multigpu.hpp: the one-process, N-device driver over RCCL.
CPU: the shard arithmetic and the reassembly order equal comap_amd/distributed.py (what the torch.distributed path
uses).  GPU: with the one device of the test box the driver's rows and null equal the single-context adapter path."""
import os
import struct
import subprocess

import numpy as np
import pytest

from comap_amd import distributed, engine
from conftest import make_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "multigpu_main")


@pytest.fixture(scope="module")
def multigpu_exe():
    src = os.path.join(ROOT, "tests", "cpp", "multigpu_main.cpp")
    deps = [src, engine.LIB_PATH] + [os.path.join(ROOT, "include", h) for h in
                                      ("comap_mi355x_multigpu.hpp", "comap_mi355x_adapter.hpp", "comap_mi355x.h")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                               "-I", "/opt/rocm/include", src, "-o", EXE, "-L", os.path.dirname(engine.LIB_PATH),
                               "-lcomap_mi355x", "-L", "/opt/rocm/lib", "-lrccl", "-lamdhip64",
                               "-Wl,-rpath," + os.path.dirname(engine.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


@pytest.mark.parametrize("world,nrep,n", [(1, 7, 5), (2, 100, 1000), (3, 10, 100), (8, 100, 1000), (8, 5, 6), (4, 0, 2), (8, 100, 37)])
def test_shard_arithmetic_equals_the_torch_distributed_path(multigpu_exe, world, nrep, n):
    out = subprocess.check_output([multigpu_exe, "shards", str(world), str(nrep), str(n)], text=True).split("\n")
    reps = [tuple(int(v) for v in l.split()[1:]) for l in out if l.startswith("rep")]
    rows = [tuple(int(v) for v in l.split()[1:]) for l in out if l.startswith("row")]
    assert reps == [distributed.replicate_shard(r, world, nrep) for r in range(world)]
    assert rows == [distributed.row_shard(r, world, n) for r in range(world)]
    # contiguous cover in rank order: concatenating the ranks' rows gives the reference's (i, j) order
    assert reps[0][0] == 0 and reps[-1][1] == nrep and all(a[1] == b[0] for a, b in zip(reps, reps[1:]))
    assert rows[0][0] == 0 and rows[-1][1] == n and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))


def test_driver_posts_one_grouped_all_gather():
    """the exchange is a single ncclAllGather per communicator inside one ncclGroupStart / ncclGroupEnd"""
    text = open(os.path.join(ROOT, "include", "comap_mi355x_multigpu.hpp")).read()
    code = "\n".join(l.split("//")[0] for l in text.split("\n"))
    assert code.count("ncclAllGather(") == 1 and code.count("ncclGroupStart(") == 1 and code.count("ncclGroupEnd(") == 1
    for other in ("ncclAllReduce(", "ncclBroadcast(", "ncclSend(", "ncclRecv(", "ncclReduceScatter("):
        assert other not in code


def _check_against_single_context(out_path, case, N, rep_cpu, rep_ram, ncls, seed):
    """rows and merged null written by tests/cpp/multigpu_main.cpp == the single-context engine's, byte for byte"""
    raw = open(out_path, "rb").read()
    nrows = struct.unpack_from("<q", raw, 0)[0]
    rec = np.dtype([("i", "<i8"), ("j", "<i8"), ("stat", "<f8"), ("pr", "<f8"), ("nm", "<f8"), ("pv", "<f8"),
                    ("rc", "<i4"), ("ns", "<i4")])
    assert nrows == N * (N - 1) // 2
    rows = np.frombuffer(raw, dtype=rec, count=nrows, offset=8)
    off = 8 + nrows * rec.itemsize
    nnull = struct.unpack_from("<q", raw, off)[0]
    assert nnull == rep_cpu * rep_ram
    nstat = np.frombuffer(raw, dtype="<f8", count=nnull, offset=off + 8)
    nnmin = np.frombuffer(raw, dtype="<f8", count=nnull, offset=off + 8 + 8 * nnull)
    eng = engine.Engine(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    nl = eng.null_intra(engine.STAT_CORRELATION, seed, 0, rep_cpu, rep_ram)
    assert np.array_equal(nstat, nl["stat"], equal_nan=True) and np.array_equal(nnmin, nl["nmin"], equal_nan=True)
    m = eng.map_sites(case["aln"])
    ref, count = eng.intra_rows(engine.STAT_CORRELATION, m["counts"], m["rate_class"], m["post_rate"], m["norm"], nl["stat"],
                                nl["nmin"], nclasses=ncls)
    assert count == nrows
    for a, b in (("i", "i"), ("j", "j"), ("stat", "stat"), ("pr", "pr_min"), ("nm", "n_min"), ("pv", "pvalue"), ("rc", "rc_min"), ("ns", "nsim")):
        assert np.array_equal(rows[a], ref[b], equal_nan=True), a


@pytest.mark.gpu
def test_one_device_driver_equals_the_single_context_path(multigpu_exe, tmp_path):
    from test_adapter_cpp import _write_case
    case = make_case(9, 70, 20, 61)
    N, rep_cpu, rep_ram, ncls, seed = 70, 3, 64, 5, 4242
    inp, out1 = tmp_path / "in.bin", tmp_path / "o1.bin"
    _write_case(inp, case, N, rep_cpu, rep_ram, ncls, seed)
    subprocess.check_call([multigpu_exe, "run", str(inp), str(out1), "1"])
    _check_against_single_context(out1, case, N, rep_cpu, rep_ram, ncls, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("nranks,rep_cpu,rep_ram,nsites", [(2, 5, 37, 70), (3, 7, 64, 91), (3, 2, 50, 33), (4, 4, 21, 57)])
def test_n_rank_logic_through_the_loopback_exchange(multigpu_exe, tmp_path, nranks, rep_cpu, rep_ram, nsites):
    """the N > 1 path of cmx::MultiGpu -- uneven replicate shards (5 over 2, 7 over 3, 2 over 3: one rank
    with NO replicate), NaN-padded send buffers, the all-gather, the reassembly into replicate order, row ranges balanced
    by pair count -- executed with N contexts on the ONE device of the test box (LoopbackExchange = the same all-gather as
    device-to-device copies), and compared byte for byte with the single-context path: rows in the reference's (i, j)
    order, p-values against the merged null, the merged null itself.  The driver also checks that a second call on the
    warm arena, the rows left on the devices and every rank's copy of the null agree (tests/cpp/multigpu_main.cpp)."""
    from test_adapter_cpp import _write_case
    case = make_case(9, nsites, 20, 61 + nranks)
    ncls, seed = 5, 4242 + nranks
    inp, out = tmp_path / "in.bin", tmp_path / "o.bin"
    _write_case(inp, case, nsites, rep_cpu, rep_ram, ncls, seed)
    r = subprocess.run([multigpu_exe, "loopback", str(inp), str(out), str(nranks)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    _check_against_single_context(out, case, nsites, rep_cpu, rep_ram, ncls, seed)


def test_rccl_exchange_refuses_two_ranks_on_one_device():
    text = open(os.path.join(ROOT, "include", "comap_mi355x_multigpu.hpp")).read()
    assert "RCCL needs one distinct device per rank" in text and "using LoopbackMultiGpu = BasicMultiGpu<LoopbackExchange>" in text
