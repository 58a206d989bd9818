"""TEST INFRASTRUCTURE ONLY: loader for the C oracle (oracle/oracle.c) and the numpy restatement.

Never imported by comap_amd/.  See the header of oracle.c for what is pinned and what is not."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ST_CORRELATION, ST_COMPENSATION, ST_COSUBSTITUTION, ST_COSINUS, ST_COVARIANCE, ST_DISCRETE_MI, ST_CORRECTED_CORRELATION, \
    ST_EUCLIDIAN_DISTANCE = range(8)
METHOD_UNIF, METHOD_DECOMP, METHOD_NAIVE = 0, 1, 2


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return os.path.join(_HERE, "_build", "liboracle.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "oracle.c")):
            try:
                build()
            except Exception:
                if not os.path.exists(path):
                    raise
        _LIB = ctypes.CDLL(path)
        _LIB.orc_stat_pair.restype = ctypes.c_double
        _LIB.orc_uniform.restype = ctypes.c_double
        _LIB.orc_uniform.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32]
    return _LIB


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Model:
    """Plain-array description of tree + model shared by the oracle entry points."""

    def __init__(self, parent, blen, leaf_of_taxon, Q, pi, rates, probs, Bk=None, method=METHOD_UNIF, nonneg=True,
                 naive_W=None):
        self.parent, self.blen, self.lot = _i32(parent), _f64(blen), _i32(leaf_of_taxon)
        self.Q, self.pi, self.rates, self.probs = _f64(Q), _f64(pi), _f64(rates), _f64(probs)
        self.S, self.C = len(self.pi), len(self.rates)
        if Bk is None:
            B0 = self.Q.copy()
            np.fill_diagonal(B0, 0.0)
            Bk = B0[None]
        self.Bk = _f64(Bk).reshape(-1, self.S, self.S)
        self.K = self.Bk.shape[0]
        self.method, self.nonneg = int(method), int(bool(nonneg))
        self.naive_W = None if naive_W is None else _f64(naive_W)
        self.nn, self.T, self.B = len(self.parent), len(self.lot), len(self.parent) - 1


def default_masks(S):
    m = np.zeros(256, dtype=np.uint32)
    if S > 32:          # no mask table for alphabets of more than 32 states: every code >= S is an unknown (oracle.c)
        m[:] = 0xFFFFFFFF
        return m
    m[:S] = 1 << np.arange(S, dtype=np.uint32)
    m[S:] = (1 << S) - 1
    return m


def map_sites(m, aln, masks=None):
    aln = np.ascontiguousarray(aln, dtype=np.uint8)
    T, N = aln.shape
    assert T == m.T
    masks = default_masks(m.S) if masks is None else np.ascontiguousarray(masks, dtype=np.uint32)
    counts = np.zeros((N, m.B, m.K))
    logL, pr, norm = np.zeros(N), np.zeros(N), np.zeros(N)
    rc = np.zeros(N, dtype=np.int32)
    D, I, U8, U32 = ctypes.c_double, ctypes.c_int, ctypes.c_uint8, ctypes.c_uint32
    lib().orc_map_sites(m.nn, _p(m.parent, I), _p(m.blen, D), m.T, _p(m.lot, I), ctypes.c_long(N), _p(aln, U8),
                        _p(masks, U32), m.S, m.C, m.K, _p(m.Q, D), _p(m.pi, D), _p(m.rates, D), _p(m.probs, D),
                        _p(m.Bk, D), m.method, m.nonneg, _p(m.naive_W, D), _p(counts, D), _p(logL, D), _p(pr, D),
                        _p(rc, I), _p(norm, D))
    return dict(counts=counts, logL=logL, post_rate=pr, rate_class=rc, norm=norm)


def map_sites_noavg(m, aln, masks=None):
    """nijt.average = no, nijt.joint = yes (computeSubstitutionVectorsNoAveraging; oracle.c orc_map_sites_noavg) ->
    counts [N, B, K], norm [N], argmax [N, B] (x * S + y of the most probable pair of ancestral states), margin [N, B]"""
    aln = np.ascontiguousarray(aln, dtype=np.uint8)
    T, N = aln.shape
    assert T == m.T
    masks = default_masks(m.S) if masks is None else np.ascontiguousarray(masks, dtype=np.uint32)
    counts = np.zeros((N, m.B, m.K))
    norm, margin = np.zeros(N), np.zeros((N, m.B))
    arg = np.zeros((N, m.B), dtype=np.int32)
    D, I, U8, U32 = ctypes.c_double, ctypes.c_int, ctypes.c_uint8, ctypes.c_uint32
    lib().orc_map_sites_noavg(m.nn, _p(m.parent, I), _p(m.blen, D), m.T, _p(m.lot, I), ctypes.c_long(N), _p(aln, U8),
                              _p(masks, U32), m.S, m.C, m.K, _p(m.Q, D), _p(m.pi, D), _p(m.rates, D), _p(m.probs, D),
                              _p(m.Bk, D), m.method, m.nonneg, _p(m.naive_W, D), _p(counts, D), _p(norm, D), _p(arg, I),
                              _p(margin, D))
    return dict(counts=counts, norm=norm, argmax=arg, margin=margin)


def map_sites_marginal(m, aln, average, masks=None, want_post=False):
    """nijt.joint = no (computeSubstitutionVectorsMarginal / ...NoAveragingMarginal; oracle.c orc_map_sites_marginal) ->
    counts [N, B, K], norm [N], anc [N, nn] marginal ancestral states, margin [N, nn], post [N, nn, C, S] on request"""
    aln = np.ascontiguousarray(aln, dtype=np.uint8)
    T, N = aln.shape
    assert T == m.T
    masks = default_masks(m.S) if masks is None else np.ascontiguousarray(masks, dtype=np.uint32)
    counts = np.zeros((N, m.B, m.K))
    norm, margin = np.zeros(N), np.zeros((N, m.nn))
    anc = np.zeros((N, m.nn), dtype=np.int32)
    post = np.zeros((N, m.nn, m.C, m.S)) if want_post else None
    D, I, U8, U32 = ctypes.c_double, ctypes.c_int, ctypes.c_uint8, ctypes.c_uint32
    lib().orc_map_sites_marginal(m.nn, _p(m.parent, I), _p(m.blen, D), m.T, _p(m.lot, I), ctypes.c_long(N), _p(aln, U8),
                                 _p(masks, U32), m.S, m.C, m.K, _p(m.Q, D), _p(m.pi, D), _p(m.rates, D), _p(m.probs, D),
                                 _p(m.Bk, D), m.method, m.nonneg, _p(m.naive_W, D), int(bool(average)), _p(counts, D),
                                 _p(norm, D), _p(post, D) if want_post else None, _p(anc, I), _p(margin, D))
    return dict(counts=counts, norm=norm, anc=anc, margin=margin, post=post)


def simulate(m, seed, g0, n):
    aln = np.zeros((m.T, n), dtype=np.uint8)
    cls = np.zeros(n, dtype=np.int32)
    D, I = ctypes.c_double, ctypes.c_int
    lib().orc_simulate(m.nn, _p(m.parent, I), _p(m.blen, D), m.T, _p(m.lot, I), m.S, m.C, _p(m.Q, D), _p(m.pi, D),
                       _p(m.rates, D), _p(m.probs, D), ctypes.c_uint64(seed), ctypes.c_uint64(g0), ctypes.c_long(n),
                       _p(aln, ctypes.c_uint8), _p(cls, I))
    return aln, cls


def simulate_continuous(m, seed, g0, n, alpha, p_inv=0.0):
    """simulations.continuous = yes: -> (aln [T, n], rates [n])"""
    aln = np.zeros((m.T, n), dtype=np.uint8)
    rates = np.zeros(n)
    D, I = ctypes.c_double, ctypes.c_int
    lib().orc_simulate_continuous(m.nn, _p(m.parent, I), _p(m.blen, D), m.T, _p(m.lot, I), m.S, _p(m.Q, D), _p(m.pi, D),
                                  D(alpha), D(p_inv), ctypes.c_uint64(seed), ctypes.c_uint64(g0), ctypes.c_long(n),
                                  _p(aln, ctypes.c_uint8), _p(rates, D))
    return aln, rates


def gamma_quantile(a, u):
    f = lib().orc_gamma_quantile
    f.restype = ctypes.c_double
    return f(ctypes.c_double(a), ctypes.c_double(u))


def stat_params(kind, threshold=0.99):
    if kind == ST_DISCRETE_MI:
        return _f64([3, 0.0, threshold, 10000.0])
    return _f64([0])


def stat_pair(kind, v1, v2, params=None):
    v1, v2 = _f64(v1), _f64(v2)
    if v1.ndim == 1:
        v1, v2 = v1[:, None], v2[:, None]
    params = stat_params(kind) if params is None else _f64(params)
    return lib().orc_stat_pair(kind, v1.shape[0], v1.shape[1], _p(v1, ctypes.c_double), _p(v2, ctypes.c_double),
                               _p(params, ctypes.c_double))


def pair_stats_intra(kind, counts, params=None):
    counts = _f64(counts)
    N, B, K = counts.shape
    params = stat_params(kind) if params is None else _f64(params)
    out = np.zeros((N, N))
    lib().orc_pair_stats_intra(kind, ctypes.c_long(N), B, K, _p(counts, ctypes.c_double), _p(params, ctypes.c_double),
                               _p(out, ctypes.c_double))
    return out


def pair_stats_inter(kind, c1, c2, params=None):
    c1, c2 = _f64(c1), _f64(c2)
    N1, B, K = c1.shape
    N2 = c2.shape[0]
    params = stat_params(kind) if params is None else _f64(params)
    out = np.zeros((N1, N2))
    lib().orc_pair_stats_inter(kind, ctypes.c_long(N1), ctypes.c_long(N2), B, K, _p(c1, ctypes.c_double),
                               _p(c2, ctypes.c_double), _p(params, ctypes.c_double), _p(out, ctypes.c_double))
    return out


def null_inter(m1, m2, kind, seed, rep_begin, rep_end, repRAM, params=None):
    """AnalysisTools::getNullDistributionInterDR (CoMap/AnalysisTools.cpp:662-735) composed from the oracle's pieces:
    per replicate simulate + map repRAM sites under each data set, score site j against site j.  Simulated-site indices
    as in orc_null_intra: g = ((rep*2 + h)*repRAM + j), h = 0 for data set 1 and 1 for data set 2."""
    stat, rcmin, prmin, nmin = [], [], [], []
    for rep in range(rep_begin, rep_end):
        res = []
        for h, m in enumerate((m1, m2)):
            aln, _ = simulate(m, seed, (rep * 2 + h) * repRAM, repRAM)
            res.append(map_sites(m, aln))
        a, b = res
        for j in range(repRAM):
            stat.append(stat_pair(kind, a["counts"][j], b["counts"][j], params))
        rcmin.append(np.minimum(a["rate_class"], b["rate_class"]))
        prmin.append(np.minimum(a["post_rate"], b["post_rate"]))
        nmin.append(np.minimum(a["norm"], b["norm"]))
    return dict(stat=np.array(stat), rcmin=np.concatenate(rcmin).astype(np.int32), prmin=np.concatenate(prmin),
                nmin=np.concatenate(nmin))


def null_intra(m, kind, seed, rep_begin, rep_end, repRAM, supplied=None, params=None):
    n = (rep_end - rep_begin) * repRAM
    stat, prmin, nmin = np.zeros(n), np.zeros(n), np.zeros(n)
    rcmin = np.zeros(n, dtype=np.int32)
    params = stat_params(kind) if params is None else _f64(params)
    if supplied is not None:
        supplied = np.ascontiguousarray(supplied, dtype=np.uint8)
        assert supplied.shape == (rep_end - rep_begin, 2, m.T, repRAM)
    D, I = ctypes.c_double, ctypes.c_int
    lib().orc_null_intra(m.nn, _p(m.parent, I), _p(m.blen, D), m.T, _p(m.lot, I), m.S, m.C, m.K, _p(m.Q, D),
                         _p(m.pi, D), _p(m.rates, D), _p(m.probs, D), _p(m.Bk, D), m.method, m.nonneg, kind,
                         _p(params, D), ctypes.c_uint64(seed), ctypes.c_long(rep_begin), ctypes.c_long(rep_end),
                         ctypes.c_long(repRAM), _p(supplied, ctypes.c_uint8), _p(stat, D), _p(rcmin, I), _p(prmin, D),
                         _p(nmin, D))
    return dict(stat=stat, rcmin=rcmin, prmin=prmin, nmin=nmin)


def intra_pvalues(stat, norms, nclasses, null_stat, null_nmin):
    stat, norms, null_stat, null_nmin = _f64(stat), _f64(norms), _f64(null_stat), _f64(null_nmin)
    N = len(norms)
    pv = np.zeros((N, N))
    nsim = np.zeros((N, N), dtype=np.int32)
    D = ctypes.c_double
    lib().orc_intra_pvalues(ctypes.c_long(N), _p(stat, D), _p(norms, D), int(nclasses), ctypes.c_long(len(null_stat)),
                            _p(null_stat, D), _p(null_nmin, D), _p(pv, D), _p(nsim, ctypes.c_int))
    return pv, nsim


def domain_index(lo, hi, n, x):
    return lib().orc_domain_index(ctypes.c_double(lo), ctypes.c_double(hi), int(n), ctypes.c_double(x))


def mi_columns(aln1, aln2, A, masks=None):
    aln1 = np.ascontiguousarray(aln1, dtype=np.uint8)
    aln2 = np.ascontiguousarray(aln2, dtype=np.uint8)
    T, N1 = aln1.shape
    N2 = aln2.shape[1]
    masks = default_masks(A) if masks is None else np.ascontiguousarray(masks, dtype=np.uint32)
    mi, hj = np.zeros((N1, N2)), np.zeros((N1, N2))
    h1, h2 = np.zeros(N1), np.zeros(N2)
    D = ctypes.c_double
    lib().orc_mi_columns(T, A, _p(masks, ctypes.c_uint32), ctypes.c_long(N1), _p(aln1, ctypes.c_uint8),
                         ctypes.c_long(N2), _p(aln2, ctypes.c_uint8), _p(mi, D), _p(hj, D), _p(h1, D), _p(h2, D))
    return dict(mi=mi, hjoint=hj, h1=h1, h2=h2)


def mica_average_mi(mi):
    """averageMI / fullAverageMI of CoMap/Mica.cpp:346-363 (mi: [n, n], upper triangle used)."""
    m = np.triu(np.asarray(mi, dtype=np.float64), 1)
    m = m + m.T
    n = m.shape[0]
    avg = np.array([(m[i, :i].sum() + m[i, i + 1:].sum()) / (n - 1) for i in range(n)])
    return avg, float(avg.mean())


def mica_zscore_null(which, mi, key):
    """null.method = z-score, CoMap/Mica.cpp:565-603: (statistic, min key) of every pair i < j in row order;
    which: 0 MI, 1 MIp = MI - APC, 2 MIc = MI / RCW."""
    mi = np.asarray(mi, dtype=np.float64)
    avg, full = mica_average_mi(mi)
    iu = np.triu_indices(mi.shape[0], 1)
    v = mi[iu].copy()
    if which == 1:
        v = v - avg[iu[0]] * avg[iu[1]] / full
    elif which == 2:
        with np.errstate(divide="ignore", invalid="ignore"):
            v = v / (avg[iu[0]] * avg[iu[1]] / 2.0)
    key = np.asarray(key, dtype=np.float64)
    return v, np.minimum(key[iu[0]], key[iu[1]])


def mica_permutation_test(aln, A, max_perm, seed, pair_begin=0, pair_end=None, masks=None):
    """miTest of CoMap/Mica.cpp:93-118 for column pairs in (i < j) row order -> (pvalue, nperm).  masks: table indexed by
    alignment code (bit a = compatible with state a); codes >= A without an entry are unknowns (gap, X, N)."""
    a = np.ascontiguousarray(aln, dtype=np.uint8)
    T, n = a.shape
    npairs = n * (n - 1) // 2
    pair_end = npairs if pair_end is None else pair_end
    pv, npm = np.zeros(pair_end - pair_begin), np.zeros(pair_end - pair_begin, dtype=np.int32)
    L = lib()
    mk = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint32)
    L.orc_mica_permutation_test_masks.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_void_p,
                                                  ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_long, ctypes.c_long,
                                                  ctypes.c_void_p, ctypes.c_void_p]
    st = L.orc_mica_permutation_test_masks(a.ctypes.data, T, n, A, None if mk is None else mk.ctypes.data,
                                           0 if mk is None else len(mk), max_perm, seed, pair_begin, pair_end,
                                           pv.ctypes.data, npm.ctypes.data)
    if st != 0:
        raise ValueError(f"oracle: permutation test: bad mask table or too many ambiguity codes ({st})")
    return pv, npm
