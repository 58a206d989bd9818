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


@pytest.mark.gpu
def test_one_device_driver_equals_the_single_context_path(multigpu_exe, tmp_path):
    from test_adapter_cpp import _write_case
    case = make_case(9, 70, 20, 61)
    N, rep_cpu, rep_ram, ncls, seed = 70, 3, 64, 5, 4242
    inp, out1, out2 = tmp_path / "in.bin", tmp_path / "o1.bin", tmp_path / "o2.bin"
    _write_case(inp, case, N, rep_cpu, rep_ram, ncls, seed)
    subprocess.check_call([multigpu_exe, "run", str(inp), str(out1), "1"])
    raw = open(out1, "rb").read()
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
