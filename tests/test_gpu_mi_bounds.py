"""-m gpu: DiscreteMutualInformationStatistic with a bounds vector (CoMap/Statistics.h:307-327) behind the C-ABI
(CMX_STAT_DISCRETE_MI_BOUNDS) against the oracle's restatement (oracle.c: orc_stat_pair, ST_DISCRETE_MI with
params = [nbounds, bounds..]) -- all pairs, two data sets, groups, the null, p-values and the rows of statistics.txt --
including the analysis the bounds exist for: nijt = Label + nijt.average = no + statistic = MI (CoETools.cpp:577-588),
13 unit bins for nucleotides, 381 for proteins.  Parity unpinned against the reference itself (it ships no statistic
output); the restatement follows Statistics.h / Domain.cpp / VectorTools::miDiscrete line by line.

Tolerance: the device sums a pair's cell terms in 2^-46 fixed point (order-free, see cmx_stat_mi.hip), so values agree
with the restatement to 1e-12 absolute; class indices, Nsim and the NaN pattern are exact."""
import numpy as np
import pytest

import oracle
from comap_amd import engine
from conftest import make_case, rel_close

pytestmark = pytest.mark.gpu
KIND = engine.STAT_DISCRETE_MI_BOUNDS


def _oparams(bounds):
    return np.concatenate([[float(len(bounds))], np.asarray(bounds, dtype=np.float64)])


def _engine(case, **kw):
    return engine.Engine(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"], **kw)


def _label_case(S, ntaxa, nsites, seed):
    """observed data mapped as nijt = Label, nijt.average = no: integer labels 0 .. S(S-1) per branch"""
    case = make_case(ntaxa, nsites, S, seed)
    W = engine.label_substitution_weights(S)
    eng = _engine(case, count_method=engine.COUNT_NAIVE, naive_weights=W)
    eng.set_mapping_options(False, True)
    return case, eng, eng.map_sites(case["aln"])


@pytest.mark.parametrize("S,ntaxa,nsites", [(4, 14, 90), (20, 40, 70)])
def test_label_bounds_all_pairs(S, ntaxa, nsites):
    case, eng, r = _label_case(S, ntaxa, nsites, 500 + S)
    bounds = engine.label_mi_bounds(S)
    assert len(bounds) == S * (S - 1) + 2 and bounds[0] == -0.5 and bounds[-1] == S * (S - 1) + 0.5
    got = eng.pair_stats(KIND, r["counts"], threshold=bounds)
    want = oracle.pair_stats_intra(oracle.ST_DISCRETE_MI, r["counts"], params=_oparams(bounds))
    rel_close(got, want, 1e-9, 1e-12)
    iu = np.triu_indices(nsites, 1)
    assert np.isfinite(got[iu]).all() and (got[iu] >= -1e-12).all() and got[iu].max() > 0.05
    # independent tables give exactly 0 on both sides (every term is log(1)): ties with the null must stay ties
    assert np.array_equal(got[iu] == 0.0, want[iu] == 0.0)


def test_arbitrary_bounds_on_real_counts_and_two_data_sets():
    """any non-decreasing bounds vector on ordinary (averaged, real-valued) counts; duplicate bounds make an empty class"""
    case = make_case(20, 130, 20, 71)
    eng = _engine(case)
    c1 = eng.map_sites(case["aln"])["counts"]
    c2 = eng.map_sites(case["aln"][:, ::-1][:, :77].copy())["counts"]
    bounds = np.array([0.0, 1e-4, 1e-3, 1e-3, 0.01, 0.05, 0.3, 1.0, 50.0])
    p = _oparams(bounds)
    rel_close(eng.pair_stats(KIND, c1, threshold=bounds), oracle.pair_stats_intra(oracle.ST_DISCRETE_MI, c1, params=p), 1e-9, 1e-12)
    rel_close(eng.pair_stats(KIND, c1, c2, threshold=bounds), oracle.pair_stats_inter(oracle.ST_DISCRETE_MI, c1, c2, params=p), 1e-9, 1e-12)
    # the factory's two-class form {0, threshold, 10000} through this kernel == through the indicator Gram (kind 5)
    for thr in (0.01, 0.99):
        a = eng.pair_stats(KIND, c1, threshold=[0.0, thr, 10000.0])
        b = eng.pair_stats(engine.STAT_DISCRETE_MI, c1, threshold=thr)
        rel_close(a, b, 1e-9, 1e-12)


def test_totals_outside_the_bounds_give_nan_and_bad_bounds_are_refused():
    """Domain::getIndex throws OutOfRangeException for x outside [b_0, b_last) (Domain.cpp:115): every statistic of such a
    site is NaN here; Domain::Domain(bounds) throws for decreasing bounds (Domain.cpp:62-72)"""
    case = make_case(10, 40, 4, 3)
    eng = _engine(case)
    c = eng.map_sites(case["aln"])["counts"].copy()
    c[5, 2, 0] = 7.0                       # beyond the last bound
    c[9, 0, 0] = -1e-3                     # below the first
    bounds = np.array([0.0, 0.05, 0.5, 5.0])
    got = eng.pair_stats(KIND, c, threshold=bounds)
    want = oracle.pair_stats_intra(oracle.ST_DISCRETE_MI, c, params=_oparams(bounds))
    rel_close(got, want, 1e-9, 1e-12)
    assert np.isnan(got[5, 6:]).all() and np.isnan(got[:5, 5]).all() and np.isnan(got[9, 10:]).all()
    assert np.isfinite(got[0, 1:5]).all()
    with pytest.raises(engine.CmxError, match="is < to bound"):
        eng.pair_stats(KIND, c, threshold=[0.0, 1.0, 0.5])
    with pytest.raises(engine.CmxError):
        eng.pair_stats(KIND, c, threshold=[0.0])


@pytest.mark.parametrize("S,ntaxa", [(4, 12), (20, 10)])
def test_label_mi_end_to_end_observed_null_pvalues(S, ntaxa):
    """nijt = Label + nijt.average = no + statistic = MI, the way CoETools::computeIntraStats runs it: observed pairs, the
    parametric-bootstrap null (replicates simulated, mapped without averaging, site j against site j), Domain classes of
    the min norms, p = (nsim - #{null < stat} + 1) / (nsim + 1), rows of statistics.txt"""
    nsites, ram, nrep, seed = 60, 50, 6, 4242
    case, eng, r = _label_case(S, ntaxa, nsites, 900 + S)
    bounds = engine.label_mi_bounds(S)
    p = _oparams(bounds)
    # ---- the null: engine vs simulate -> map -> oracle statistic (the engine's own simulator and mapping are checked
    # against the oracle elsewhere; the restatement of the NoAveraging mapping may pick another of two equally probable
    # ancestral pairs, so the statistic is compared on the engine's own integer labels)
    nl = eng.null_intra(KIND, seed, 0, nrep, ram, threshold=bounds)
    for rep in range(nrep):
        a0, _ = eng.simulate(seed, (rep * 2 + 0) * ram, ram)
        a1, _ = eng.simulate(seed, (rep * 2 + 1) * ram, ram)
        m0, m1 = eng.map_sites(a0), eng.map_sites(a1)
        st = np.array([oracle.stat_pair(oracle.ST_DISCRETE_MI, m0["counts"][j], m1["counts"][j], p) for j in range(ram)])
        sl = slice(rep * ram, (rep + 1) * ram)
        rel_close(nl["stat"][sl], st, 1e-9, 1e-12)
        assert np.array_equal(nl["stat"][sl] == 0.0, st == 0.0)
        rel_close(nl["nmin"][sl], np.minimum(m0["norm"], m1["norm"]), 1e-12)
        assert np.array_equal(nl["rcmin"][sl], np.minimum(m0["rate_class"], m1["rate_class"]))
    # ---- observed statistic, p-values: the reference's rule applied by the oracle to the engine's values -- exact
    stat = eng.pair_stats(KIND, r["counts"], threshold=bounds)
    rel_close(stat, oracle.pair_stats_intra(oracle.ST_DISCRETE_MI, r["counts"], params=p), 1e-9, 1e-12)
    nclasses = 4
    pv, ns = eng.intra_pvalues(stat, r["norm"], nclasses, nl["stat"], nl["nmin"])
    po, nso = oracle.intra_pvalues(stat, r["norm"], nclasses, nl["stat"], nl["nmin"])
    iu = np.triu_indices(nsites, 1)
    assert np.array_equal(ns, nso) and np.array_equal(pv[iu], po[iu], equal_nan=True)
    assert (ns[iu] > 0).sum() > 100
    # ---- rows of statistics.txt through the compacting entry point: the same numbers, pair by pair
    rows, count = eng.intra_rows(KIND, r["counts"], r["rate_class"], r["post_rate"], r["norm"], nl["stat"], nl["nmin"], nclasses,
                                 threshold=bounds)
    assert count == nsites * (nsites - 1) // 2 == len(rows)
    assert np.array_equal(rows["stat"], stat[iu]) and np.array_equal(rows["pvalue"], pv[iu], equal_nan=True)
    assert np.array_equal(rows["nsim"], ns[iu])
    # a discrete statistic ties with its null all the time: the strict "<" of CoETools.cpp:715 must see the ties
    tied = sum(int((nl["stat"] == v).any()) for v in rows["stat"][:300])
    assert tied > 30


def test_rows_range_blocks_equal_the_dense_path():
    """the default large-N path (row blocks, no N x N matrix) with the bounds statistic: any split of the rows gives the
    bytes of the dense path"""
    import torch
    case, eng, r = _label_case(4, 10, 150, 77)
    bounds = engine.label_mi_bounds(4)
    n = 150
    nl = eng.null_intra(KIND, 5, 0, 4, 60, threshold=bounds)
    rows, count = eng.intra_rows(KIND, r["counts"], r["rate_class"], r["post_rate"], r["norm"], nl["stat"], nl["nmin"], 5, threshold=bounds)
    dev = torch.device("cuda:0")
    BK = r["counts"].shape[1] * r["counts"].shape[2]
    d_counts = torch.from_numpy(np.ascontiguousarray(r["counts"].reshape(n, BK).T)).to(dev)
    d_rc = torch.from_numpy(r["rate_class"].astype(np.int32)).to(dev)
    d_pr, d_nm = torch.from_numpy(r["post_rate"]).to(dev), torch.from_numpy(r["norm"]).to(dev)
    d_ns, d_nn = torch.from_numpy(nl["stat"]).to(dev), torch.from_numpy(nl["nmin"]).to(dev)
    got = []
    for rb, re in ((0, 37), (37, 100), (100, 150)):
        buf = torch.zeros(n * n // 2 * engine.PAIR_ROW.itemsize, dtype=torch.uint8, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        eng.intra_rows_range_dev(KIND, d_counts, d_rc, d_pr, d_nm, d_ns, d_nn, 5, buf, cnt, row_begin=rb, row_end=re, threshold=bounds)
        torch.cuda.synchronize()
        k = int(cnt.item())
        got.append(np.frombuffer(buf.cpu().numpy().tobytes()[: k * engine.PAIR_ROW.itemsize], dtype=engine.PAIR_ROW))
    got = np.concatenate(got)
    assert len(got) == count and got.tobytes() == rows.tobytes()


def test_groups_and_two_data_set_null():
    case, eng, r = _label_case(4, 9, 40, 12)
    bounds = engine.label_mi_bounds(4)
    p = _oparams(bounds)
    groups = [[0, 3, 7], [1, 2], [5, 9, 11, 30], [4, 4, 6]]
    got = eng.group_stats(KIND, r["counts"], groups, threshold=bounds)
    for g, v in zip(groups, got):          # AbstractMinimumStatistic::getValueForGroup, Statistics.h:121-133
        want = min(oracle.stat_pair(oracle.ST_DISCRETE_MI, r["counts"][g[i]], r["counts"][g[j]], p) for i in range(1, len(g)) for j in range(i))
        assert abs(v - want) <= 1e-12
    # two data sets (AnalysisTools::getNullDistributionInterDR): same tree, other branch lengths
    case2 = dict(case)
    case2["blen"] = case["blen"] * 1.7
    W = engine.label_substitution_weights(4)
    eng2 = _engine(case2, count_method=engine.COUNT_NAIVE, naive_weights=W)
    eng2.set_mapping_options(False, True)
    nl = eng.null_inter(eng2, KIND, 99, 0, 3, 30, threshold=bounds)
    for rep in range(3):
        a0, _ = eng.simulate(99, (rep * 2 + 0) * 30, 30)
        a1, _ = eng2.simulate(99, (rep * 2 + 1) * 30, 30)
        m0, m1 = eng.map_sites(a0), eng2.map_sites(a1)
        st = np.array([oracle.stat_pair(oracle.ST_DISCRETE_MI, m0["counts"][j], m1["counts"][j], p) for j in range(30)])
        rel_close(nl["stat"][rep * 30:(rep + 1) * 30], st, 1e-9, 1e-12)
