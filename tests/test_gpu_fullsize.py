"""-m gpu: BASELINE.json's full sizes through size-independent properties (the oracle cannot finish these in
seconds): site independence (any split / permutation of the sites gives the same bits), replicate-sharding invariance
of the null, structure of the p-value rule, symmetry / range / scale invariance of the statistics, norm and
likelihood identities -- plus oracle spot checks on small slices of the same inputs."""
import numpy as np
import pytest

import oracle
from comap_amd import engine, synthetic as sy
from conftest import rel_close

pytestmark = pytest.mark.gpu


def _protein_case(nsites, seed=20260101, ntaxa=64):
    parent, blen, lot = sy.random_tree(ntaxa, seed)
    mdl = sy.protein_model(0.5, 4)
    om = oracle.Model(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
    eng = engine.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
    aln, _ = eng.simulate(seed + 1, 0, nsites)
    eng.rates_ = mdl["rates"]
    return eng, om, aln


def test_configuration2_mapping_2000x64_protein_properties():
    eng, om, aln = _protein_case(2000)
    r = eng.map_sites(aln)
    c = r["counts"]
    assert c.shape == (2000, 125, 1) and np.isfinite(c).all() and (c >= 0).all()
    # computeNormForSite identity and likelihood sanity
    rel_close(r["norm"], np.sqrt((c.sum(axis=2) ** 2).sum(axis=1)), 1e-12)
    assert (r["logL"] < 0).all() and r["rate_class"].min() >= 0 and r["rate_class"].max() <= 3
    assert (r["post_rate"] >= eng.rates_.min()).all() and (r["post_rate"] <= eng.rates_.max()).all()  # a posterior mean
    # sites are independent: any split and any permutation reproduce the same bits
    h1, h2 = eng.map_sites(aln[:, :777]), eng.map_sites(aln[:, 777:])
    assert np.array_equal(np.concatenate([h1["counts"], h2["counts"]]), c)
    perm = np.random.default_rng(1).permutation(2000)
    rp = eng.map_sites(np.ascontiguousarray(aln[:, perm]))
    assert np.array_equal(rp["counts"], c[perm]) and np.array_equal(rp["norm"], r["norm"][perm])
    assert np.array_equal(rp["rate_class"], r["rate_class"][perm])
    # oracle on a slice of the very same alignment
    o = oracle.map_sites(om, aln[:, 1000:1024])
    rel_close(c[1000:1024], o["counts"], 1e-6, 1e-300)
    rel_close(r["logL"][1000:1024], o["logL"], 1e-9)
    assert np.array_equal(r["rate_class"][1000:1024], o["rate_class"])
    # all 1 999 000 pairs: symmetric, |r| <= 1, invariant to a rescaling of the vectors, slice against the oracle
    st = eng.pair_stats(0, c)
    iu = np.triu_indices(2000, 1)
    assert np.nanmax(np.abs(st[iu])) <= 1 + 1e-12
    st2 = eng.pair_stats(0, c * 3.5)
    rel_close(st2[iu], st[iu], 1e-9, 1e-13)
    rel_close(st[1000:1024, 1000:1024][np.triu_indices(24, 1)],
              oracle.pair_stats_intra(0, o["counts"])[np.triu_indices(24, 1)], 1e-6, 1e-12)


def test_configuration3_null_125x2000_sharding_and_pvalue_rule():
    eng, om, aln = _protein_case(2000)
    nrep, ram = 125, 2000                                     # one GPU's share of configs[2] (1000 replicates / 8)
    full = eng.null_intra(0, 20260101, 0, nrep, ram)
    assert np.isfinite(full["stat"]).all() and np.abs(full["stat"]).max() <= 1 + 1e-12
    a, b = eng.null_intra(0, 20260101, 0, 60, ram), eng.null_intra(0, 20260101, 60, nrep, ram)
    for k in ("stat", "nmin", "prmin"):
        assert np.array_equal(np.concatenate([a[k], b[k]]), full[k])
    assert np.array_equal(np.concatenate([a["rcmin"], b["rcmin"]]), full["rcmin"])
    # one replicate against the oracle (same counter-based simulator)
    o = oracle.null_intra(om, 0, 20260101, 7, 8, 64)
    g = eng.null_intra(0, 20260101, 7, 8, 64)
    rel_close(g["stat"], o["stat"], 1e-6, 1e-12)
    assert np.array_equal(g["rcmin"], o["rcmin"])
    # p-values over all pairs: (nsim + 1) p is the integer nsim - count + 1; within a norm class p never increases with
    # the statistic; pairs at the maximum norm are NA (half-open Domain)
    m = eng.map_sites(aln)
    st = eng.pair_stats(0, m["counts"])
    pv, ns = eng.intra_pvalues(st, m["norm"], 10, full["stat"], full["nmin"])
    iu = np.triu_indices(2000, 1)
    p, n, s = pv[iu], ns[iu], st[iu]
    ok = ~np.isnan(p)
    k = p[ok] * (n[ok] + 1)
    assert np.max(np.abs(k - np.rint(k))) < 1e-6 and np.rint(k).min() >= 1 and (np.rint(k) <= n[ok] + 1).all()
    nm = np.minimum(m["norm"][iu[0]], m["norm"][iu[1]])
    top = nm == m["norm"].max()
    assert np.isnan(p[top]).all() and (n[top] == 0).all()
    cls = np.floor(nm / (m["norm"].max() / 10)).astype(int)
    for q in range(10):
        sel = ok & (cls == q) & (n == np.bincount(n[ok & (cls == q)]).argmax() if (ok & (cls == q)).any() else False)
        if sel.sum() > 2:
            order = np.argsort(s[sel], kind="stable")
            assert (np.diff(p[sel][order]) <= 1e-15).all()


def test_configuration4_dna_10000x256_compensation_properties():
    parent, blen, lot = sy.random_tree(256, 20260102)
    mdl = sy.dna_model(0.5, 4)
    Bk = sy.weighted_register(mdl["Q"], sy.compensation_weights_dna())[None]
    eng = engine.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], Bk=Bk, clamp_negative=False)
    om = oracle.Model(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], Bk=Bk, nonneg=False)
    aln, _ = eng.simulate(20260103, 0, 10000)
    r = eng.map_sites(aln)
    c = r["counts"]
    assert c.shape == (10000, 509, 1) and np.isfinite(c).all()
    rel_close(r["norm"], np.sqrt((c.sum(axis=2) ** 2).sum(axis=1)), 1e-12)
    o = oracle.map_sites(om, aln[:, 5000:5016])
    rel_close(c[5000:5016], o["counts"], 1e-6, 1e-300)
    # weighted counts are antisymmetric in the weights: reversing the sign of W reverses the vectors
    eng_m = engine.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], Bk=-Bk, clamp_negative=False)
    rm = eng_m.map_sites(aln[:, :512])
    rel_close(rm["counts"], -c[:512], 1e-9, 1e-300)
    # compensation over a 3 000-site block (4.5e6 pairs; all 5e7 are the benchmark's job): in [0, 1], symmetric
    sub = c[:3000]
    st = eng.pair_stats(1, sub)
    iu = np.triu_indices(3000, 1)
    v = st[iu]
    assert np.isfinite(v).all() and v.min() >= -1e-12 and v.max() <= 1 + 1e-12
    rel_close(eng.pair_stats(1, sub[::-1].copy())[::-1, ::-1].T[iu], v, 1e-9, 1e-13)   # reversed site order, (j, i)
    rel_close(st[:16, :16][np.triu_indices(16, 1)], oracle.pair_stats_intra(1, c[:16])[np.triu_indices(16, 1)], 1e-6, 1e-12)


def _rows_range(eng, counts_bm, rc, pr, norm, ns, nm, nclasses, kind, a, b, filters=None, offset=0, cap=None):
    import torch
    from comap_amd.pipeline import sum_pairs
    dev = counts_bm.device
    n = norm.shape[0]
    cap = max(sum_pairs(n, a, b), 1) if cap is None else cap
    rows = torch.zeros(cap * engine.PAIR_ROW.itemsize + offset, dtype=torch.uint8, device=dev)[offset:]
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    eng.intra_rows_range_dev(kind, counts_bm, rc, pr, norm, ns, nm, nclasses, rows, count, a, b, filters)
    torch.cuda.synchronize()
    k = min(int(count.item()), cap)
    return np.frombuffer(rows[: k * engine.PAIR_ROW.itemsize].cpu().numpy().tobytes(), dtype=engine.PAIR_ROW).copy()


def test_rows_range_equals_dense_path_and_concatenates():
    """cmx_intra_rows_range_dev (no N x N matrix, a block of rows at a time) against the dense path, for the whole
    triangle, for unbalanced ranges that do not fall on tile boundaries, and with filters"""
    import torch
    from comap_amd.distributed import row_shard
    eng, om, aln = _protein_case(1500)
    r = eng.map_sites(aln)
    nl = eng.null_intra(0, 11, 0, 40, 500)
    f = engine.PairFilters(min_rate_class=1, max_rate_class_diff=2, min_statistic=0.05)
    dev = torch.device("cuda:0")
    cbm = torch.from_numpy(np.ascontiguousarray(r["counts"].reshape(1500, -1).T)).to(dev)
    rc, pr, nm = (torch.from_numpy(r[k]).to(dev) for k in ("rate_class", "post_rate", "norm"))
    ns, nn = torch.from_numpy(nl["stat"]).to(dev), torch.from_numpy(nl["nmin"]).to(dev)
    for flt in (None, f):
        ref, cnt = eng.intra_rows(0, r["counts"], r["rate_class"], r["post_rate"], r["norm"], nl["stat"], nl["nmin"], 10, flt)
        full = _rows_range(eng, cbm, rc, pr, nm, ns, nn, 10, 0, 0, 1500, flt)
        assert len(full) == cnt == len(ref)
        for k in ("i", "j", "rc_min", "nsim"):
            assert np.array_equal(full[k], ref[k]), k
        for k in ("stat", "pr_min", "n_min"):
            assert np.array_equal(full[k], ref[k]), k
        assert np.array_equal(np.isnan(full["pvalue"]), np.isnan(ref["pvalue"]))
        assert np.array_equal(np.nan_to_num(full["pvalue"]), np.nan_to_num(ref["pvalue"]))
        parts = [_rows_range(eng, cbm, rc, pr, nm, ns, nn, 10, 0, a, b, flt) for a, b in ((0, 1), (1, 130), (130, 777), (777, 1499), (1499, 1500))]
        assert np.array_equal(np.concatenate(parts).tobytes(), full.tobytes())
        shards = [_rows_range(eng, cbm, rc, pr, nm, ns, nn, 10, 0, *row_shard(q, 3, 1500), flt) for q in range(3)]
        assert np.array_equal(np.concatenate(shards).tobytes(), full.tobytes())
        # a rows buffer that is only 8-byte aligned (rows leave one per lane instead of packed), and one that is too
        # small (the first `capacity` rows are written, the rest dropped)
        assert _rows_range(eng, cbm, rc, pr, nm, ns, nn, 10, 0, 0, 1500, flt, offset=8).tobytes() == full.tobytes()
        for off in (0, 8):
            short = _rows_range(eng, cbm, rc, pr, nm, ns, nn, 10, 0, 0, 1500, flt, offset=off, cap=len(full) - 37)
            assert short.tobytes() == full[: len(full) - 37].tobytes()


def test_configuration4_pair_stage_all_10000_sites_through_row_ranges():
    """BASELINE configs[3] at full size: all 49 995 000 compensation statistics of the 10 000 x 256 DNA alignment through
    cmx_intra_rows_range_dev (compacted on the device, nothing dense crosses PCIe), oracle blocks at the matrix corners"""
    import torch
    parent, blen, lot = sy.random_tree(256, 20260102)
    mdl = sy.dna_model(0.5, 4)
    Bk = sy.weighted_register(mdl["Q"], sy.compensation_weights_dna())[None]
    eng = engine.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], Bk=Bk, clamp_negative=False)
    n = 10000
    aln, _ = eng.simulate(20260103, 0, n)
    dev = torch.device("cuda:0")
    d_aln = torch.from_numpy(aln).to(dev)
    counts = torch.empty((eng.B * eng.K, n), dtype=torch.float64, device=dev)
    logL, pr, nm = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
    rc = torch.empty(n, dtype=torch.int32, device=dev)
    eng.map_sites_dev(d_aln, counts, logL, pr, rc, nm)
    # rows in two halves of the pair count (what two ranks would do); only a thinned set of rows crosses PCIe
    from comap_amd.distributed import row_shard
    f = engine.PairFilters(min_statistic=0.2)
    tot, keep = 0, []
    for q in range(2):
        a, b = row_shard(q, 2, n)
        rows = _rows_range(eng, counts, rc, pr, nm, None, None, 10, 1, a, b, f)
        assert (rows["i"] >= a).all() and (rows["i"] < b).all() and (rows["j"] > rows["i"]).all()
        key = rows["i"].astype(np.int64) * n + rows["j"]
        assert (np.diff(key) > 0).all()                      # the reference's (i, j) order
        assert (np.abs(rows["stat"]) >= 0.2).all() and rows["stat"].max() <= 1 + 1e-12
        tot += len(rows)
        keep.append(rows)
    rows = np.concatenate(keep)
    assert 0 < tot < n * (n - 1) // 2
    # oracle at the corners of the matrix: first rows x last columns, last rows, a diagonal block in the middle
    c = counts.T.cpu().numpy().reshape(n, eng.B, eng.K)
    key = rows["i"].astype(np.int64) * n + rows["j"]
    for ia, ja in ((0, n - 16), (n - 16, n - 16), (5000, 5000), (0, 0)):
        blk_i, blk_j = np.arange(ia, ia + 16), np.arange(ja, ja + 16)
        o = oracle.pair_stats_inter(1, c[blk_i], c[blk_j])
        for x, i in enumerate(blk_i):
            for y, j in enumerate(blk_j):
                if j <= i:
                    continue
                pos = int(np.searchsorted(key, i * n + j))
                found = pos < len(key) and key[pos] == i * n + j
                if abs(o[x, y]) >= 0.2 * (1 + 1e-9):
                    assert found and abs(rows["stat"][pos] - o[x, y]) <= 1e-6 * abs(o[x, y]) + 1e-12
                elif abs(o[x, y]) < 0.2 * (1 - 1e-9):
                    assert not found


def test_target_size_null_replicate_against_the_oracle():
    """the north-star workload's null: one replicate of 10 000 simulated site pairs (20 000 sites re-mapped) against the
    oracle, statistic and all three minima"""
    eng, om, aln = _protein_case(8)
    g = eng.null_intra(0, 20260108, 3, 4, 10000)
    o = oracle.null_intra(om, 0, 20260108, 3, 4, 10000)
    rel_close(g["stat"], o["stat"], 1e-6, 1e-12)
    rel_close(g["nmin"], o["nmin"], 1e-6)
    rel_close(g["prmin"], o["prmin"], 1e-6)
    assert np.array_equal(g["rcmin"], o["rcmin"])


def test_configuration5_mica_5000x5000x256_identity_and_oracle_tiles():
    """BASELINE configs[4] at full size: MI = H1 + H2 - Hjoint on all 25e6 cross pairs, and the oracle on scattered 8 x 8
    tiles including the last (ragged: 5000 = 78 * 64 + 8) one"""
    import torch
    rng = np.random.default_rng(20260103)
    T, A, n1, n2 = 256, 20, 5000, 5000
    base = rng.integers(0, A, size=(T, 1))
    a1 = np.where(rng.random((T, n1)) < 0.6, base, rng.integers(0, A, size=(T, n1))).astype(np.uint8)
    a2 = np.where(rng.random((T, n2)) < 0.4, base, rng.integers(0, A, size=(T, n2))).astype(np.uint8)
    a2[:, 4990:][rng.random((T, 10)) < 0.05] = A            # unknowns in the last columns
    dev = torch.device("cuda:0")
    d1, d2 = torch.from_numpy(a1).to(dev), torch.from_numpy(a2).to(dev)
    mi = torch.empty((n1, n2), dtype=torch.float64, device=dev)
    hj = torch.empty_like(mi)
    h1 = torch.empty(n1, dtype=torch.float64, device=dev)
    h2 = torch.empty(n2, dtype=torch.float64, device=dev)
    eng = engine.Engine()
    eng.mi_columns_dev(d1, mi, hj, d2, A, None, h1, h2)
    torch.cuda.synchronize()
    resid = (mi - (h1[:, None] + h2[None, :] - hj)).abs().max().item()
    assert resid < 1e-12
    assert torch.isfinite(mi).all() and mi.min().item() > -1e-12
    mih, hjh = mi.cpu().numpy(), hj.cpu().numpy()
    for i0, j0 in ((0, 0), (4992, 4992), (0, 4992), (4992, 0), (2496, 1234), (63, 4989)):
        o = oracle.mi_columns(a1[:, i0:i0 + 8], a2[:, j0:j0 + 8], A)
        assert np.max(np.abs(o["mi"] - mih[i0:i0 + 8, j0:j0 + 8])) < 1e-12
        assert np.max(np.abs(o["hjoint"] - hjh[i0:i0 + 8, j0:j0 + 8])) < 1e-12


def test_mica_unknowns_at_size_mixed_tiles_and_intra_layout():
    """gaps in a third of the columns of a 1 501 x 1 300 x 256 protein rectangle (tiles with none, some and only gapped
    pairs; ragged last tiles of the packed 12 x 6 layout), partial ambiguity codes in a few columns (those pairs go to
    the LDS-table kernel), and the intra layout of the first alignment: identity on every pair, oracle on scattered tiles"""
    import torch
    rng = np.random.default_rng(7)
    T, A, n1, n2 = 256, 20, 1501, 1300
    base = rng.integers(0, A, size=(T, 1))
    a1 = np.where(rng.random((T, n1)) < 0.5, base, rng.integers(0, A, size=(T, n1))).astype(np.uint8)
    a2 = np.where(rng.random((T, n2)) < 0.5, base, rng.integers(0, A, size=(T, n2))).astype(np.uint8)
    for a in (a1, a2):
        cols = rng.random(a.shape[1]) < 0.33
        a[(rng.random(a.shape) < 0.12) & cols[None, :]] = A          # unknown (default mask: every state)
    masks = oracle.default_masks(A)
    masks[A + 1] = 0b1100000                                          # a two-state ambiguity code
    a1[rng.random(T) < 0.05, 17] = A + 1
    a2[rng.random(T) < 0.05, 1294] = A + 1
    dev = torch.device("cuda:0")
    d1, d2 = torch.from_numpy(a1).to(dev), torch.from_numpy(a2).to(dev)
    d_masks = torch.from_numpy(masks.astype(np.int64)).to(torch.int32).to(dev)
    eng = engine.Engine()
    mi = torch.empty((n1, n2), dtype=torch.float64, device=dev)
    hj = torch.empty_like(mi)
    h1 = torch.empty(n1, dtype=torch.float64, device=dev)
    h2 = torch.empty(n2, dtype=torch.float64, device=dev)
    eng.mi_columns_dev(d1, mi, hj, d2, A, d_masks, h1, h2)
    torch.cuda.synchronize()
    assert (mi - (h1[:, None] + h2[None, :] - hj)).abs().max().item() < 1e-11
    mih, hjh = mi.cpu().numpy(), hj.cpu().numpy()
    for i0, j0 in ((0, 0), (1493, 1292), (12, 1288), (1490, 0), (700, 650), (10, 1290)):
        o = oracle.mi_columns(a1[:, i0:i0 + 8], a2[:, j0:j0 + 8], A, masks=masks)
        assert np.max(np.abs(o["mi"] - mih[i0:i0 + 8, j0:j0 + 8])) < 1e-11
        assert np.max(np.abs(o["hjoint"] - hjh[i0:i0 + 8, j0:j0 + 8])) < 1e-11
    # intra layout: upper triangle filled, the rest NaN
    mii = torch.empty((n1, n1), dtype=torch.float64, device=dev)
    hji = torch.empty_like(mii)
    eng.mi_columns_dev(d1, mii, hji, None, A, d_masks, h1, None)
    torch.cuda.synchronize()
    m = mii.cpu().numpy()
    iu = np.triu_indices(n1, 1)
    assert np.isfinite(m[iu]).all() and np.isnan(m[np.tril_indices(n1)]).all()
    for i0, j0 in ((0, 8), (12, 24), (1480, 1490), (5, 1493)):
        o = oracle.mi_columns(a1[:, i0:i0 + 8], a1[:, j0:j0 + 8], A, masks=masks)
        blk = m[i0:i0 + 8, j0:j0 + 8]
        keep = np.add.outer(np.arange(i0, i0 + 8), np.zeros(8, int)) < np.add.outer(np.zeros(8, int), np.arange(j0, j0 + 8))
        assert np.max(np.abs(o["mi"][keep] - blk[keep])) < 1e-11


@pytest.mark.parametrize("n", [1, 2, 3, 65, 130])
def test_rows_of_tiny_alignments_and_tiny_nulls(n):
    """The row passes cut a row into eight runs of whole 64-column steps and the p-value lookup cuts a class into bins:
    alignments of one, two and three sites (zero, one, three pairs: most runs empty), one and two 64-column steps per row,
    a null of seven values (no bins at all), an empty null, a capacity of zero (count only) -- all against the dense
    oracle path."""
    eng, om, aln = _protein_case(max(n, 8))
    r = eng.map_sites(aln)
    r = {k: v[:n] for k, v in r.items()}
    st = oracle.pair_stats_intra(0, r["counts"]) if n > 1 else np.zeros((1, 1))
    iu = np.triu_indices(n, 1)
    rng = np.random.default_rng(n)
    for nnull in (0, 7, 900):
        ns = rng.normal(0.0, 0.3, nnull)
        nm = rng.uniform(0.0, float(r["norm"].max()), nnull)
        rows, cnt = eng.intra_rows(0, r["counts"], r["rate_class"], r["post_rate"], r["norm"], ns if nnull else None, nm if nnull else None, 3)
        assert cnt == n * (n - 1) // 2 == len(rows)
        assert np.array_equal(rows["i"], iu[0]) and np.array_equal(rows["j"], iu[1])
        if n > 1:
            rel_close(rows["stat"], st[iu], 1e-9, 1e-12)
            if nnull:
                opv, ons = oracle.intra_pvalues(st, r["norm"], 3, ns, nm)
                # (the statistic differs from the oracle's in the last bits: compare the counts where no null value is that close)
                assert np.array_equal(rows["nsim"], ons[iu])
                far = np.array([np.min(np.abs(ns - v)) > 1e-9 for v in rows["stat"]])
                assert np.array_equal(rows["pvalue"][far], opv[iu][far], equal_nan=True)
            else:
                assert np.isnan(rows["pvalue"]).all() and (rows["nsim"] == 0).all()
        none, cnt0 = eng.intra_rows(0, r["counts"], r["rate_class"], r["post_rate"], r["norm"], capacity=0)
        assert cnt0 == cnt and len(none) == 0


def test_kept_gram_blocks_over_several_row_blocks_at_7000_sites():
    """cmx_intra_gram_prefetch_dev with more than one row block of the pair loop (256 MiB / 8 n = 4 736 rows at 7 000 sites:
    two blocks for the whole triangle, and a row range that starts inside the first and ends inside the second): the record
    pass over the kept blocks writes the bytes of the pass that computes its own."""
    import torch
    from comap_amd.pipeline import IntraAnalysis, sum_pairs
    parent, blen, lot = sy.random_tree(24, 77)
    mdl = sy.dna_model(0.5, 4)
    eng = engine.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
    n = 7000
    aln, _ = eng.simulate(5, 0, n)
    d_aln = torch.from_numpy(aln).cuda()
    ana = IntraAnalysis(eng, d_aln, engine.STAT_CORRELATION, 8)
    ana.get_vectors()
    nb = ana.null_distribution(3, 0, 20, 500)
    for rb, re_ in ((0, n), (3000, 6500)):
        rec, npairs = ana.compute_intra_compact(nb["stat"], nb["nmin"], rb, re_)
        assert npairs == sum_pairs(n, rb, re_)
        own = rec[:npairs * engine.PAIR_COMPACT.itemsize].clone()
        ana.prefetch_intra_gram(rb, re_)
        rec, _ = ana.compute_intra_compact(nb["stat"], nb["nmin"], rb, re_)
        assert torch.equal(rec[:npairs * engine.PAIR_COMPACT.itemsize], own)
        h = own[:16 * 1000].cpu().numpy().view(engine.PAIR_COMPACT)
        assert np.isfinite(h["stat"]).all() and (h["below"] != 0xffffffff).any()
