"""CMX_SCRATCH_GUARD (include/comap_mi355x.h): one call of each pipeline with SEVERAL replicates and ragged sizes while
every hand-sized device buffer carries a canary behind its last byte.  Round 3 shipped a scratch buffer sized [nn][rep_ram]
for a kernel that writes [nn][nrep * rep_ram]: it passed a round of nrep = 2 tests on allocator slack and
surfaced as an abort.  Here such a buffer makes the call fail and names itself.

The sizes are deliberately awkward: replicate blocks that are not multiples of 64 sites, odd replicate counts, row
capacities below the row count, two data sets with different rate distributions."""
import numpy as np
import pytest

import oracle
from oracle import candidates as ocand
from oracle import cluster as oc
from comap_amd import engine, mica, synthetic
from conftest import make_case, rel_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def guard():
    was = engine.scratch_guard(True)
    engine.scratch_guard_failures(clear=True)
    engine.scratch_shrink(None, 0)
    yield
    engine.scratch_shrink(None, 0)
    engine.scratch_guard(was)


def _eng(case, **kw):
    return engine.Engine(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"], **kw)


def _om(case, **kw):
    return oracle.Model(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"], **kw)


def _clean(*engines):
    for e in engines:
        e.synchronize()       # verifies every canary of the context
        e.scratch_check()
    assert engine.scratch_guard_failures() == [], engine.scratch_guard_failures()


@pytest.mark.parametrize("nstates,ntaxa", [(20, 11), (4, 13)])
def test_observed_pipeline_and_fused_null_under_the_guard(nstates, ntaxa):
    case = make_case(ntaxa, 157, nstates, 5 + nstates)
    eng, om = _eng(case), _om(case)
    m = eng.map_sites(case["aln"])
    rel_close(m["counts"], oracle.map_sites(om, case["aln"])["counts"], 1e-6, 1e-12)
    st = eng.pair_stats(0, m["counts"])
    nl = eng.null_intra(0, 17, 2, 7, 37)                    # 5 replicates of 37 sites
    o = oracle.null_intra(om, 0, 17, 2, 7, 37)
    rel_close(nl["stat"], o["stat"], 1e-6, 1e-12)
    eng.intra_pvalues(st, m["norm"], 5, nl["stat"], nl["nmin"])
    rows, total = eng.intra_rows(0, m["counts"], m["rate_class"], m["post_rate"], m["norm"], nl["stat"], nl["nmin"], 5)
    assert total == 157 * 156 // 2 == len(rows)
    few, total = eng.intra_rows(0, m["counts"], m["rate_class"], m["post_rate"], m["norm"], nl["stat"], nl["nmin"], 5, capacity=33)
    assert len(few) == 33 and total == 157 * 156 // 2       # capacity < count
    sup = np.stack([np.stack([oracle.simulate(om, 5, (r * 2 + h) * 21, 21)[0] for h in range(2)]) for r in range(3)])
    eng.null_intra(1, 0, 0, 3, 21, supplied=sup)
    _clean(eng)


@pytest.mark.parametrize("kind", [engine.STAT_CORRELATION, engine.STAT_DISCRETE_MI_BOUNDS])
def test_unfused_nulls_with_three_replicates_under_the_guard(kind):
    """nijt.average = no and the bounds statistic take null_unfused_dev: the path the round-3 overflow lived in"""
    case = make_case(9, 10, 4, 23)
    W = engine.label_substitution_weights(4)
    eng = _eng(case, count_method=engine.COUNT_NAIVE, naive_weights=W)
    eng.set_mapping_options(False, True)
    thr = engine.label_mi_bounds(4) if kind == engine.STAT_DISCRETE_MI_BOUNDS else 0.99
    nl = eng.null_intra(kind, 3, 1, 4, 45, threshold=thr)   # 3 replicates of 45
    assert nl["stat"].shape == (135,)
    eng.set_mapping_options(True, False)
    eng.null_intra(engine.STAT_CORRELATION, 3, 0, 3, 29)
    _clean(eng)


def test_inter_null_and_inter_rows_under_the_guard():
    c1 = make_case(10, 61, 20, 61)
    c2 = dict(c1)
    c2["blen"] = c1["blen"] * np.random.default_rng(3).uniform(0.5, 1.5, size=len(c1["blen"]))
    mdl2 = synthetic.protein_model(0.9, 3)
    c2.update(rates=mdl2["rates"], probs=mdl2["probs"])
    e1, e2, o1, o2 = _eng(c1), _eng(c2), _om(c1), _om(c2)
    g = e1.null_inter(e2, 0, 4242, 1, 6, 23)                # 5 replicates of 23
    o = oracle.null_inter(o1, o2, 0, 4242, 1, 6, 23)
    rel_close(g["stat"], o["stat"], 1e-6, 1e-12)
    m1, m2 = e1.map_sites(c1["aln"]), e2.map_sites(c1["aln"][:, :40])
    rows, total = e1.inter_rows(0, m1, m2)
    assert total == 61 * 40
    few, total = e1.inter_rows(0, m1, m2, capacity=17)
    assert len(few) == 17 and total == 61 * 40
    _clean(e1, e2)


def test_the_round3_sizing_trips_the_guard():
    """the simulator's node states are [nn][nrep * rep_ram]; pretend the buffer was asked for as [nn][rep_ram] (round 3's
    bug).  The allocation keeps its real size, so the kernel's correct writes land on the canary, not outside."""
    case = make_case(10, 8, 20, 61)
    e1, e2 = _eng(case), _eng(case)
    nn = len(case["parent"])
    rep_ram, nrep = 23, 3
    engine.scratch_shrink("inter0_st", nn * rep_ram)
    try:
        e1.null_inter(e2, 0, 1, 0, nrep, rep_ram)           # the call itself may pass: nothing re-requests the buffer in it
        with pytest.raises(engine.CmxError, match="inter0_st.*written past its end"):
            e1.synchronize()
    finally:
        engine.scratch_shrink("inter0_st", 0)
    found = engine.scratch_guard_failures(clear=True)
    assert len(found) == 1 and "scratch:inter0_st" in found[0]
    e1.null_inter(e2, 0, 1, 0, nrep, rep_ram)               # the real size is clean
    _clean(e1, e2)


def test_clustering_and_candidate_groups_under_the_guard():
    case = make_case(11, 90, 20, 4)
    eng, om = _eng(case), _om(case)
    g = eng.cluster_null(oc.DIST_CORRELATION, oc.LINK_COMPLETE, 123, 1, 4, 50)      # 3 replicates of 50 sites
    assert len(g["merge"]) == 3     # (trees are compared with the oracle in tests/test_gpu_cluster.py, on alignments without duplicate columns)
    counts = eng.map_sites(case["aln"])["counts"]
    eng.cluster_sites(oc.DIST_COMPENSATION, oc.LINK_AVERAGE, counts)
    mp = oracle.map_sites(om, case["aln"])
    rng = np.random.default_rng(7)
    groups = [list(rng.choice(90, size=int(rng.integers(2, 5)), replace=False)) for _ in range(5)]
    windows = [[(mp["norm"][i] - 0.3, mp["norm"][i] + 0.3) for i in g] for g in groups]
    observed = eng.group_stats(0, mp["counts"], groups)
    r = eng.candidate_groups(0, windows, [1, 1, 0, 1, 1], observed, min_sim=25, rep_ram=47, max_trials=4, seed=2024)
    ro = ocand.candidate_groups(om, 0, windows, [1, 1, 0, 1, 1], observed, min_sim=25, rep_ram=47, max_trials=4, seed=2024)
    assert np.array_equal(r["n2"], ro["n2"])
    _clean(eng)


@pytest.mark.parametrize("A,T", [(20, 40), (4, 300)])
def test_mica_pipelines_under_the_guard(A, T):
    case = make_case(T, 75, A, 31 + A)
    eng = _eng(case)
    aln = case["aln"].copy()
    aln[np.random.default_rng(2).random(aln.shape) < 0.05] = A      # unknowns: the weighted kernels
    r = eng.mi_columns(aln, nalpha=A)
    assert r["mi"].shape == (75, 75)
    eng.mi_columns(aln[:, :31], aln[:, 31:], nalpha=A)
    eng.mica_parametric_null(5, 3, 37, with_norms=True)             # 3 replicates of 37 column pairs
    i1, i2 = mica.bootstrap_indices(9, 75, 3, 41)
    eng.mi_pairs(aln, i1, i2, None, A)                              # the non-parametric bootstrap's pair list
    eng.mica_permutation_test(aln[:, :9], 130, 4, nalpha=A)
    _clean(eng)


def test_continuous_rates_and_codon_alphabets_under_the_guard():
    case = make_case(9, 40, 20, 12)
    eng = _eng(case)
    eng.null_intra_continuous(0, 8, 0, 3, 27, 0.7)
    eng.simulate_continuous(8, 5, 77, 0.7, 0.1)
    from comap_amd import protein_models as pm
    Q, pi = pm.synthetic_reversible(9, 105)                          # a plain-kernel alphabet
    parent, blen, lot = synthetic.random_tree(7, 3)
    mdl = synthetic.dna_model(0.5, 4)
    e9 = engine.Engine(parent, blen, lot, Q, pi, mdl["rates"], mdl["probs"])
    a, _ = e9.simulate(4, 0, 83)
    e9.map_sites(a)
    e9.null_intra(0, 4, 0, 3, 31)
    _clean(eng, e9)


def test_guard_switched_on_after_buffers_exist():
    """a scratch buffer from before the guard was switched on has no room for a canary: it is re-created, not checked"""
    engine.scratch_guard(False)
    try:
        case = make_case(9, 80, 20, 2)
        eng = _eng(case)
        m = eng.map_sites(case["aln"])
        a = eng.pair_stats(0, m["counts"])                   # allocates the pair operands without a guard tail
        engine.scratch_guard(True)
        b = eng.pair_stats(0, m["counts"])                   # the same buffers, now re-created with their canaries
        assert np.array_equal(a, b, equal_nan=True)
        eng.null_intra(0, 3, 0, 3, 31)
        _clean(eng)
    finally:
        engine.scratch_guard(True)
