"""The 16-byte pair records of the unfiltered pair loop (cmx_pair_compact, include/comap_mi355x.h) and their expansion to
the 48-byte statistics.txt rows of CoMap/CoETools.cpp:662-722.

CPU: cmx_expand_compact_rows is host code -- positions, minima, the p-value quotient (nsim - below + 1) / (nsim + 1) of
CoETools.cpp:717 and the NA rule against a plain Python loop, for row ranges and any thread count.
GPU: records written by the device + expansion == the rows the device compacts itself, byte for byte."""
import numpy as np
import pytest

from comap_amd import engine
from conftest import make_case


def _synthetic(n, seed):
    rng = np.random.default_rng(seed)
    rc = rng.integers(0, 4, size=n).astype(np.int32)
    pr = rng.uniform(0.1, 3.0, size=n)
    nm = rng.uniform(0.0, 5.0, size=n)
    return rc, pr, nm


def _reference_rows(n, rb, re_, rc, pr, nm, rec):
    rows = []
    k = 0
    for i in range(rb, re_):
        for j in range(i + 1, n):
            c = rec[k]
            k += 1
            if c["below"] == 0xffffffff:
                pv, ns = np.nan, 0
            else:
                ns = int(c["nsim"])
                pv = float(ns - int(c["below"]) + 1) / float(ns + 1)
            rows.append((i, j, c["stat"], min(rc[i], rc[j]), ns, min(pr[i], pr[j]), min(nm[i], nm[j]), pv))
    return np.array(rows, dtype=engine.PAIR_ROW)


@pytest.mark.parametrize("n,rb,re_,threads", [(2, 0, 2, 1), (37, 0, 37, 1), (37, 5, 20, 3), (64, 63, 64, 2), (101, 0, 101, 16), (50, 10, 10, 4)])
def test_expansion_equals_the_reference_row_loop(n, rb, re_, threads):
    rc, pr, nm = _synthetic(n, n + rb)
    npairs = (re_ - rb) * (n - 1) - (re_ * (re_ - 1) - rb * (rb - 1)) // 2
    rng = np.random.default_rng(3)
    rec = np.zeros(npairs, dtype=engine.PAIR_COMPACT)
    rec["stat"] = rng.normal(size=npairs)
    rec["nsim"] = rng.integers(1, 100000, size=npairs)
    rec["below"] = (rng.random(npairs) * (rec["nsim"] + 1)).astype(np.uint32)
    na = rng.random(npairs) < 0.1
    rec["below"][na] = 0xffffffff
    rec["nsim"][na] = 0
    got = engine.expand_compact_rows(n, rb, re_, rc, pr, nm, rec, threads)
    ref = _reference_rows(n, rb, re_, rc, pr, nm, rec)
    assert got.tobytes() == ref.tobytes()


def test_expansion_rejects_a_wrong_record_count():
    rc, pr, nm = _synthetic(10, 1)
    with pytest.raises(engine.CmxError):
        engine.expand_compact_rows(10, 0, 10, rc, pr, nm, np.zeros(44, dtype=engine.PAIR_COMPACT))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,nstates", [(engine.STAT_CORRELATION, 20), (engine.STAT_COMPENSATION, 4), (engine.STAT_DISCRETE_MI, 20)])
def test_device_records_expand_to_the_device_rows(kind, nstates):
    import torch
    from comap_amd.pipeline import IntraAnalysis
    case = make_case(10, 203, nstates, 17 + kind)
    eng = engine.Engine(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    d_aln = torch.from_numpy(case["aln"]).cuda()
    ana = IntraAnalysis(eng, d_aln, kind, 6, threshold=0.05)
    ana.get_vectors()
    nb = ana.null_distribution(99, 0, 40, 64)
    n = 203
    for (rb, re_), with_null in (((0, n), True), ((0, n), False), ((17, 130), True), ((202, 203), True), ((60, 60), True)):
        ns, nm = (nb["stat"], nb["nmin"]) if with_null else (None, None)
        rows, count = ana.compute_intra_rows(ns, nm, rb, re_)
        nrows = int(count.item())
        ref = rows[:nrows * engine.PAIR_ROW.itemsize].cpu().numpy().view(engine.PAIR_ROW)
        rec, npairs = ana.compute_intra_compact(ns, nm, rb, re_)
        assert npairs == nrows
        rec_h = rec[:npairs * engine.PAIR_COMPACT.itemsize].cpu().numpy().view(engine.PAIR_COMPACT)
        got = engine.expand_compact_rows(n, rb, re_, ana.rate_class.cpu().numpy(), ana.post_rate.cpu().numpy(), ana.norm.cpu().numpy(),
                                         rec_h, nthreads=3)
        assert got.tobytes() == ref.tobytes()
        if with_null and npairs > 100:
            assert (rec_h["below"] != 0xffffffff).any() and np.isfinite(got["pvalue"]).any()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,nstates", [(engine.STAT_CORRELATION, 20), (engine.STAT_COMPENSATION, 4), (engine.STAT_DISCRETE_MI, 20)])
def test_prefetched_gram_gives_the_same_records(kind, nstates):
    """cmx_intra_gram_prefetch_dev: the observed pairs' statistics enqueued ahead (on a side stream, beside the null) and
    kept for the next record pass with the same arguments.  Same bytes as without; consumed by one call; ignored when the
    row range differs; a statistic with a parameter (DiscreteMI's threshold) is not kept at all (the calls still agree)."""
    import torch
    from comap_amd.pipeline import IntraAnalysis
    case = make_case(10, 310, nstates, 5 + kind)
    eng = engine.Engine(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    d_aln = torch.from_numpy(case["aln"]).cuda()
    ana = IntraAnalysis(eng, d_aln, kind, 6, threshold=0.05)
    n = 310
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def records(rb, re_, ns, nm):
        rec, npairs = ana.compute_intra_compact(ns, nm, rb, re_)
        return rec[:npairs * engine.PAIR_COMPACT.itemsize].cpu().numpy().tobytes()

    for rb, re_ in ((0, n), (33, 290), (309, 310)):
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            ana.get_vectors()
            ana.prefetch_intra_gram(rb, re_)
        nb = ana.null_distribution(7, 0, 30, 64)
        main_s.wait_stream(side)
        with_kept = records(rb, re_, nb["stat"], nb["nmin"])
        plain = records(rb, re_, nb["stat"], nb["nmin"])        # the kept blocks were consumed: this one computes its own
        from comap_amd.pipeline import sum_pairs
        assert with_kept == plain and len(plain) == sum_pairs(n, rb, re_) * engine.PAIR_COMPACT.itemsize
        # kept for another range: ignored, and dropped
        ana.prefetch_intra_gram(0, n)
        torch.cuda.synchronize()
        assert records(rb, re_, nb["stat"], nb["nmin"]) == plain
        assert records(0, n, None, None) == records(0, n, None, None)
    # Gram blocks kept, then the vectors rewritten by another mapping call into the same buffer: the kept blocks are dropped
    aln2 = d_aln.clone()
    aln2[:, 0:100] = d_aln[:, 100:200]
    ana.get_vectors()
    ana.prefetch_intra_gram(0, n)
    ana.get_vectors(aln2)
    changed = records(0, n, None, None)
    assert changed == records(0, n, None, None)
    ana.get_vectors()
    assert changed != records(0, n, None, None)
