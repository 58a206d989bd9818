"""ctypes binding of libcomap_mi355x.so (include/comap_mi355x.h) -- plumbing only.

All compute happens in the HIP kernels behind the C-ABI; this module never falls back to a CPU
implementation: if the shared library is missing or no MI355X is visible, calls raise.
numpy arrays go through the host-pointer entry points; torch CUDA tensors go through the `_dev`
entry points (raw device pointers + the current torch stream).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COMAP_MI355X_LIB", os.path.join(_HERE, "libcomap_mi355x.so"))  # override: diagnostic builds

STAT_CORRELATION, STAT_COMPENSATION, STAT_COSUBSTITUTION, STAT_COSINUS, STAT_COVARIANCE, STAT_DISCRETE_MI, \
    STAT_CORRECTED_CORRELATION, STAT_EUCLIDIAN_DISTANCE, STAT_DISCRETE_MI_BOUNDS, STAT_SCALAR_PRODUCT = range(10)
STAT_BY_NAME = {
    # names of the reference's `statistic=` option (CoMap/CoETools.cpp:540-599)
    "Correlation": STAT_CORRELATION, "Compensation": STAT_COMPENSATION, "Cosubstitution": STAT_COSUBSTITUTION,
    "Cosinus": STAT_COSINUS, "Covariance": STAT_COVARIANCE, "MI": STAT_DISCRETE_MI,
    "CorrectedCorrelation": STAT_CORRECTED_CORRELATION, "EuclidianDistance": STAT_EUCLIDIAN_DISTANCE,
    "MI(bounds)": STAT_DISCRETE_MI_BOUNDS,   # DiscreteMutualInformationStatistic(const Vdouble& bounds): pass the bounds as `threshold`
}


def label_mi_bounds(nstates):
    """bounds of the MI statistic under nijt = Label (CoMap/CoETools.cpp:577-588): -0.5, 0.5, .., S(S-1) + 0.5, i.e. one
    unit bin per substitution label 0 (none) .. S(S-1)"""
    n = nstates * (nstates - 1)
    return -0.5 + np.arange(n + 2, dtype=np.float64)


def _stat_params(kind, threshold, mean_vectors):
    """host parameter block of a statistic: the MI threshold, or for CorrectedCorrelation the two per-branch mean
    vectors [2][B] (Statistics.h:176-204; one vector given = used for both operands, CoMap.cpp:350-359)"""
    if int(kind) == STAT_CORRECTED_CORRELATION:
        if mean_vectors is None:
            raise CmxError(-1, "CorrectedCorrelation needs mean_vectors")
        mv = _f64(mean_vectors)
        if mv.ndim == 1:
            mv = np.stack([mv, mv])
        return np.ascontiguousarray(mv)
    if int(kind) == STAT_DISCRETE_MI_BOUNDS:      # [nbounds, b_0 .. b_{n-1}]; the bounds arrive in `threshold`
        b = _f64(threshold).ravel()
        return np.ascontiguousarray(np.concatenate([[float(len(b))], b]))
    return _f64([threshold])
COUNT_EXPECTED, COUNT_NAIVE = 0, 1

EXPORTS = [
    "cmx_version", "cmx_ctx_create", "cmx_ctx_destroy", "cmx_last_error", "cmx_get_info",
    "cmx_get_transition_matrices", "cmx_synchronize", "cmx_debug_walk", "cmx_map_sites", "cmx_set_mapping_options", "cmx_map_sites_dev", "cmx_simulate", "cmx_simulate_dev", "cmx_simulate_continuous",
    "cmx_simulate_continuous_dev", "cmx_null_intra_continuous", "cmx_null_intra_continuous_dev", "cmx_mi_pairs_dev",
    "cmx_pair_stats", "cmx_pair_stats_dev", "cmx_null_intra", "cmx_null_simulate_dev", "cmx_null_intra_dev", "cmx_null_inter",
    "cmx_null_inter_dev", "cmx_intra_pvalues", "cmx_intra_rows", "cmx_intra_rows_dev", "cmx_intra_rows_range_dev",
    "cmx_intra_pvalues_dev", "cmx_inter_rows", "cmx_inter_rows_dev", "cmx_mica_bootstrap_indices", "cmx_mica_parametric_null", "cmx_mi_columns", "cmx_mi_columns_dev", "cmx_mi_pairs",
    "cmx_mica_permutation_test", "cmx_mica_permutation_test_dev", "cmx_mica_permutation_test_masks", "cmx_mica_permutation_test_masks_dev", "cmx_mica_average_mi", "cmx_mica_average_mi_dev", "cmx_mica_zscore_null", "cmx_mica_zscore_null_dev",
    "cmx_group_stats", "cmx_group_stats_dev", "cmx_candidate_groups", "cmx_debug_candidate_cursor",
    "cmx_hclust", "cmx_hclust_dev", "cmx_cluster_sites", "cmx_cluster_sites_dev", "cmx_cluster_null",
    "cmx_intra_compact_range_dev", "cmx_intra_gram_prefetch_dev", "cmx_expand_compact_rows", "cmx_vector_matrix",
    "cmx_scratch_check", "cmx_debug_scratch_guard", "cmx_debug_scratch_guard_failures", "cmx_debug_scratch_shrink",
]
# clustering.distance / clustering.method options of the reference (CoMap/CoMap.cpp:402-428, :460-472)
DIST_CORRELATION, DIST_COMPENSATION, DIST_EUCLIDIAN = range(3)
LINK_COMPLETE, LINK_SINGLE, LINK_AVERAGE = range(3)
DIST_BY_NAME = {"cor": DIST_CORRELATION, "Correlation": DIST_CORRELATION, "comp": DIST_COMPENSATION,
                "Compensation": DIST_COMPENSATION, "euclidian": DIST_EUCLIDIAN, "Euclidian": DIST_EUCLIDIAN}
LINK_BY_NAME = {"complete": LINK_COMPLETE, "single": LINK_SINGLE, "average": LINK_AVERAGE}
CLUSTER_MAX_SITES = 5000
MICA_MI, MICA_MIP, MICA_MIC = range(3)      # null.method_zscore.stat (Mica.cpp:551-559)


def scratch_guard(on=None):
    """CMX_SCRATCH_GUARD (include/comap_mi355x.h): switch the canary guard on/off for contexts created from now on
    (None = query); returns the previous state"""
    lib = load_library()
    return bool(lib.cmx_debug_scratch_guard(ctypes.c_int(-1 if on is None else int(bool(on)))))


def scratch_guard_failures(clear=False):
    """buffers the guard has found written past their end so far (process-wide), one string each"""
    lib = load_library()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.cmx_debug_scratch_guard_failures.restype = ctypes.c_size_t
    n = lib.cmx_debug_scratch_guard_failures(buf, ctypes.c_size_t(len(buf)), ctypes.c_int(int(clear)))
    return [l for l in buf.value.decode().split("\n") if l][:n]


def scratch_shrink(name, nbytes):
    """test hook of the guard: pretend scratch buffer `name` is asked for with only nbytes (0 removes, name None removes all)"""
    lib = load_library()
    lib.cmx_debug_scratch_shrink.restype = None
    lib.cmx_debug_scratch_shrink(None if name is None else name.encode(), ctypes.c_size_t(nbytes))


def mica_bootstrap_indices(seed, nsites, nrep_cpu, nrep_ram):
    """index1 / index2 of Mica's non-parametric bootstrap (SiteContainerTools::sampleSites, Mica.cpp:426-430) from the engine's
    counter RNG (cmx_mica_bootstrap_indices; host-side, no GPU): [nrep_cpu * nrep_ram] each"""
    lib = load_library()
    n = nrep_cpu * nrep_ram
    i1, i2 = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    st = lib.cmx_mica_bootstrap_indices(ctypes.c_uint64(seed), _sz(nsites), _sz(nrep_cpu), _sz(nrep_ram), _vp(i1), _vp(i2))
    if st != 0:
        raise CmxError(st, "cmx_mica_bootstrap_indices: bad arguments")
    return i1, i2


def label_substitution_weights(nstates):
    """nijt = Label (bpp::LabelSubstitutionCount, doc/comap.texi nijt option): every ordered pair of
    distinct states carries its own label 1 .. S(S-1), numbered row by row; the "count" of a branch is then the
    label of the substitution when there is exactly one.  It is the Naive count with these weights:
    Engine(..., count_method=COUNT_NAIVE, naive_weights=label_substitution_weights(S))."""
    W = np.zeros((nstates, nstates))
    W[~np.eye(nstates, dtype=bool)] = np.arange(1, nstates * (nstates - 1) + 1)
    return W


class CmxError(RuntimeError):
    """Raised for any non-zero cmx_status (the reference throws bpp::Exception)."""

    def __init__(self, status, message):
        super().__init__(f"cmx status {status}: {message}")
        self.status = status


class _Model(ctypes.Structure):
    _fields_ = [("nstates", ctypes.c_int32), ("nclasses", ctypes.c_int32), ("ntypes", ctypes.c_int32),
                ("Q", ctypes.c_void_p), ("pi", ctypes.c_void_p), ("rates", ctypes.c_void_p),
                ("probs", ctypes.c_void_p), ("Bk", ctypes.c_void_p), ("count_method", ctypes.c_int32),
                ("clamp_negative", ctypes.c_int32), ("naive_weights", ctypes.c_void_p),
                ("nmodels", ctypes.c_int32), ("Qs", ctypes.c_void_p), ("pis", ctypes.c_void_p), ("Bks", ctypes.c_void_p),
                ("model_of_branch", ctypes.c_void_p), ("root_freqs", ctypes.c_void_p)]


class _Tree(ctypes.Structure):
    _fields_ = [("nnodes", ctypes.c_int32), ("parent", ctypes.c_void_p), ("blen", ctypes.c_void_p),
                ("ntaxa", ctypes.c_int32), ("leaf_of_taxon", ctypes.c_void_p)]


class PairFilters(ctypes.Structure):
    """cmx_pair_filters: the pair filters of CoETools::computeIntraStats (CoETools.cpp:674-693)."""
    _fields_ = [("min_rate_class", ctypes.c_int32), ("max_rate_class_diff", ctypes.c_int32), ("min_rate", ctypes.c_double),
                ("max_rate_diff", ctypes.c_double), ("min_statistic", ctypes.c_double)]

    def __init__(self, min_rate_class=0, max_rate_class_diff=-1, min_rate=0.0, max_rate_diff=-1.0, min_statistic=0.0):
        super().__init__(min_rate_class, max_rate_class_diff, min_rate, max_rate_diff, min_statistic)


class InterFilters(ctypes.Structure):
    """cmx_inter_filters: the filters of CoETools::computeInterStats (CoETools.cpp:755-812)."""
    _fields_ = [("min_rate_class1", ctypes.c_int32), ("min_rate_class2", ctypes.c_int32), ("max_rate_class_diff", ctypes.c_int32),
                ("independent_comparisons", ctypes.c_int32), ("min_rate1", ctypes.c_double), ("min_rate2", ctypes.c_double),
                ("max_rate_diff", ctypes.c_double), ("min_statistic", ctypes.c_double), ("reference_norm_quirk", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]

    def __init__(self, min_rate_class1=0, min_rate_class2=0, max_rate_class_diff=-1, independent_comparisons=False, min_rate1=0.0,
                 min_rate2=0.0, max_rate_diff=-1.0, min_statistic=0.0, reference_norm_quirk=False):
        super().__init__(min_rate_class1, min_rate_class2, max_rate_class_diff, int(independent_comparisons), min_rate1, min_rate2,
                         max_rate_diff, min_statistic, int(reference_norm_quirk), 0)


PAIR_ROW = np.dtype([("i", np.int32), ("j", np.int32), ("stat", np.float64), ("rc_min", np.int32), ("nsim", np.int32),
                     ("pr_min", np.float64), ("n_min", np.float64), ("pvalue", np.float64)], align=True)
# cmx_pair_compact (include/comap_mi355x.h): the unfiltered pair loop as 16 bytes per pair; below = 0xffffffff: PValue NA
PAIR_COMPACT = np.dtype([("stat", np.float64), ("below", np.uint32), ("nsim", np.uint32)])


def expand_compact_rows(n, row_begin, row_end, rate_class, post_rate, norm, compact, nthreads=1):
    """cmx_expand_compact_rows (host side, no GPU): PAIR_COMPACT records of the rows [row_begin, row_end) -> PAIR_ROW records"""
    lib = load_library()
    compact = np.ascontiguousarray(compact).view(PAIR_COMPACT).ravel()
    rows = np.zeros(len(compact), dtype=PAIR_ROW)
    rc, pr, nm = np.ascontiguousarray(rate_class, dtype=np.int32), _f64(post_rate), _f64(norm)
    st = lib.cmx_expand_compact_rows(_sz(n), _sz(row_begin), _sz(row_end), _vp(rc), _vp(pr), _vp(nm), _vp(compact), _sz(len(compact)),
                                     _vp(rows), ctypes.c_int(int(nthreads)))
    if st != 0:
        raise CmxError(st, "cmx_expand_compact_rows: bad arguments")
    return rows



class _Info(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("nstates", "nclasses", "ntypes", "nnodes", "nbranches", "ntaxa",
                                               "ninternal", "device", "cu_count", "waves")] + \
               [("workspace_bytes", ctypes.c_size_t)] + \
               [(n, ctypes.c_int32) for n in ("device_states", "device_classes", "products_per_pass", "leaf_ops_per_pass",
                                              "ws_loads_per_pass", "ws_stores_per_pass", "products_per_pass_null",
                                              "leaf_ops_per_pass_null", "cherry_tables")]


_lib = None


def load_library():
    """Load the C-ABI library; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not built; run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
        lib = ctypes.CDLL(LIB_PATH)
        lib.cmx_version.restype = ctypes.c_char_p
        lib.cmx_last_error.restype = ctypes.c_char_p
        lib.cmx_last_error.argtypes = [ctypes.c_void_p]
        lib.cmx_ctx_destroy.restype = None
        lib.cmx_ctx_destroy.argtypes = [ctypes.c_void_p]
        _lib = lib
    return _lib


def _vp(a):
    if a is None:
        return ctypes.c_void_p(None)
    if isinstance(a, np.ndarray):
        return ctypes.c_void_p(a.ctypes.data)
    return ctypes.c_void_p(int(a.data_ptr()))  # torch tensor


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _sz(x):
    return ctypes.c_size_t(int(x))


def debug_walk(parent, blen, leaf_of_taxon, Q, pi, rates, probs, Bk=None):
    """Host-side compilation of the tree into what the mapping kernel's walk reads (no GPU): dict(nrec[NV,16], ldsched,
    msched[nops,2], slot[nnodes], loads, stores, products, leaf_ops, products_tables, leaf_ops_tables, cherry_tables) --
    counts are per rate-class pass; *_tables: the walk of resolved alignments with cherry tables (class-fused nucleotide
    models; cherry_tables = 0: no such walk).  The call fails if the engine's numerical self-check of either walk does."""
    lib = load_library()
    keep = [np.ascontiguousarray(parent, dtype=np.int32), _f64(blen), np.ascontiguousarray(leaf_of_taxon, dtype=np.int32),
            _f64(Q), _f64(pi), _f64(rates), _f64(probs), None if Bk is None else _f64(Bk)]
    p, bl, lot, Qa, pia, ra, pr, Bka = keep
    S, C = len(pia), len(ra)
    K = 1 if Bka is None else Bka.reshape(-1, S, S).shape[0]
    model = _Model(S, C, K, _vp(Qa), _vp(pia), _vp(ra), _vp(pr), _vp(Bka), 0, 1, _vp(None))
    tree = _Tree(len(p), _vp(p), _vp(bl), len(lot), _vp(lot))
    nn = len(p)
    cap = 64 * nn * max(K, 1) + 64
    nrec, ld, ms = (np.zeros(cap, dtype=np.int32) for _ in range(3))
    slot = np.zeros(nn, dtype=np.int32)
    stats = np.zeros(8, dtype=np.uint64)
    n1, n2, n3 = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    st = lib.cmx_debug_walk(ctypes.byref(model), ctypes.byref(tree), _vp(nrec), _sz(cap), ctypes.byref(n1), _vp(ld),
                            _sz(cap), ctypes.byref(n2), _vp(ms), _sz(cap), ctypes.byref(n3), _vp(slot), _vp(stats))
    if st != 0:
        raise CmxError(st, lib.cmx_last_error(None).decode())
    return dict(nrec=nrec[: n1.value].reshape(-1, 16).copy(), ldsched=ld[: n2.value].copy(),
                msched=ms[: n3.value].reshape(-1, 2).copy(), slot=slot, loads=int(stats[0]), stores=int(stats[1]),
                products=int(stats[2]), leaf_ops=int(stats[3]), products_tables=int(stats[4]), leaf_ops_tables=int(stats[5]),
                cherry_tables=int(stats[6]))


def debug_candidate_cursor(norm_windows, analysable, min_sim, norms, max_trials):
    """Host-side run of the candidate cursor (no GPU).  norms: [nbatches, rep_ram].
    -> dict(n2, trials, batches, pseudo_groups=[(group, batch, [sites])])"""
    lib = load_library()
    off = np.zeros(len(norm_windows) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(g) for g in norm_windows])
    lo = _f64([w[0] for g in norm_windows for w in g])
    hi = _f64([w[1] for g in norm_windows for w in g])
    ok = np.ascontiguousarray(analysable, dtype=np.uint8)
    nm = _f64(norms)
    nb, rr = nm.shape
    G = len(norm_windows)
    n2 = np.zeros(G, dtype=np.uint32)
    cap = nb * rr
    pg_group, pg_batch = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32)
    pg_off, pg_sites = np.zeros(cap + 1, dtype=np.int64), np.zeros(cap, dtype=np.int32)
    trials, used, npg = ctypes.c_uint32(0), ctypes.c_uint64(0), ctypes.c_size_t(0)
    st = lib.cmx_debug_candidate_cursor(_sz(G), _vp(off), _vp(lo), _vp(hi), _vp(ok), ctypes.c_uint32(min_sim), _vp(nm), _sz(rr),
                                        _sz(nb), ctypes.c_uint32(max_trials), _vp(n2), ctypes.byref(trials), ctypes.byref(used),
                                        _vp(pg_group), _vp(pg_batch), _vp(pg_off), _vp(pg_sites), _sz(cap), _sz(cap),
                                        ctypes.byref(npg))
    if st != 0:
        raise CmxError(st, "cmx_debug_candidate_cursor: bad arguments")
    pgs = [(int(pg_group[q]), int(pg_batch[q]), [int(x) for x in pg_sites[pg_off[q]:pg_off[q + 1]]]) for q in range(npg.value)]
    return dict(n2=n2, trials=trials.value, batches=used.value, pseudo_groups=pgs)


class Engine:
    """One context per GPU (cmx_ctx).  tree/model given as plain arrays (see include/comap_mi355x.h)."""

    def __init__(self, parent=None, blen=None, leaf_of_taxon=None, Q=None, pi=None, rates=None, probs=None, Bk=None,
                 count_method=COUNT_EXPECTED, clamp_negative=True, naive_weights=None, device=0, model_of_branch=None,
                 root_freqs=None):
        """Non-homogeneous model set (CoETools.cpp:126-206): Q [M, S, S], pi [M, S], Bk [M, K, S, S] or None,
        model_of_branch [nnodes], root_freqs [S]."""
        self._lib = load_library()
        self._ctx = ctypes.c_void_p(None)
        self.device = device
        if parent is None:
            st = self._lib.cmx_ctx_create(None, None, int(device), ctypes.byref(self._ctx))
            self.S = self.C = self.K = self.B = self.T = 0
        else:
            self._keep = [np.ascontiguousarray(parent, dtype=np.int32), _f64(blen),
                          np.ascontiguousarray(leaf_of_taxon, dtype=np.int32), _f64(Q), _f64(pi), _f64(rates),
                          _f64(probs), None if Bk is None else _f64(Bk),
                          None if naive_weights is None else _f64(naive_weights)]
            p, bl, lot, Qa, pia, ra, pr, Bka, nw = self._keep
            C = len(ra)
            if model_of_branch is None:
                S = len(pia)
                K = 1 if Bka is None else Bka.reshape(-1, S, S).shape[0]
                model = _Model(S, C, K, _vp(Qa), _vp(pia), _vp(ra), _vp(pr), _vp(Bka), int(count_method),
                               int(bool(clamp_negative)), _vp(nw))
            else:
                M, S = pia.shape
                K = 1 if Bka is None else Bka.reshape(M, -1, S, S).shape[1]
                mob = np.ascontiguousarray(model_of_branch, dtype=np.int32)
                rf = _f64(root_freqs)
                self._keep += [mob, rf]
                model = _Model(S, C, K, _vp(None), _vp(None), _vp(ra), _vp(pr), _vp(None), int(count_method),
                               int(bool(clamp_negative)), _vp(nw), M, _vp(Qa), _vp(pia), _vp(Bka), _vp(mob), _vp(rf))
            tree = _Tree(len(p), _vp(p), _vp(bl), len(lot), _vp(lot))
            st = self._lib.cmx_ctx_create(ctypes.byref(model), ctypes.byref(tree), int(device), ctypes.byref(self._ctx))
            self.S, self.C, self.K, self.B, self.T = S, C, K, len(p) - 1, len(lot)
        if st != 0:
            raise CmxError(st, self._lib.cmx_last_error(None).decode())

    # -- plumbing
    def _check(self, st):
        if st != 0:
            raise CmxError(st, self._lib.cmx_last_error(self._ctx).decode())

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.cmx_ctx_destroy(self._ctx)
            self._ctx = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = _Info()
        self._check(self._lib.cmx_get_info(self._ctx, ctypes.byref(i)))
        return {n: getattr(i, n) for n, _ in _Info._fields_}

    def transition_matrices(self):
        P = np.zeros((self.C, self.B, self.S, self.S))
        self._check(self._lib.cmx_get_transition_matrices(self._ctx, _vp(P)))
        return P

    def synchronize(self):
        self._check(self._lib.cmx_synchronize(self._ctx))

    def scratch_check(self):
        """verify every canary of this context (CMX_SCRATCH_GUARD); raises CmxError naming the buffer written past its end"""
        self._check(self._lib.cmx_scratch_check(self._ctx))

    @staticmethod
    def _stream():
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    # -- host-pointer entry points (numpy in, numpy out; reference layouts)
    def map_sites(self, aln, masks=None, want_counts=True):
        """aln: uint8 [T, N].  Returns dict(counts[N,B,K], logL, post_rate, rate_class, norm)."""
        aln = np.ascontiguousarray(aln, dtype=np.uint8)
        T, N = aln.shape
        if T != self.T:
            raise CmxError(-1, f"alignment has {T} rows, tree has {self.T} taxa")
        counts = np.zeros((N, self.B, self.K)) if want_counts else None
        logL, pr, norm = np.zeros(N), np.zeros(N), np.zeros(N)
        rc = np.zeros(N, dtype=np.int32)
        mk = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint32)
        self._check(self._lib.cmx_map_sites(self._ctx, _vp(aln), _sz(N), _sz(N), _vp(mk),
                                            _sz(0 if mk is None else len(mk)), _vp(counts), _vp(logL), _vp(pr),
                                            _vp(rc), _vp(norm)))
        return dict(counts=counts, logL=logL, post_rate=pr, rate_class=rc, norm=norm)

    def set_mapping_options(self, average=True, joint=True):
        """nijt.average / nijt.joint (CoETools.cpp:393-406): which of computeSubstitutionVectors{, NoAveraging, Marginal,
        NoAveragingMarginal} every later mapping of this engine (observed data and nulls) uses; default (True, True)."""
        self._check(self._lib.cmx_set_mapping_options(self._ctx, int(bool(average)), int(bool(joint))))

    def simulate(self, seed, g0, n):
        aln = np.zeros((self.T, n), dtype=np.uint8)
        cls = np.zeros(n, dtype=np.int32)
        self._check(self._lib.cmx_simulate(self._ctx, ctypes.c_uint64(seed), ctypes.c_uint64(g0), _sz(n), _vp(aln),
                                           _vp(cls)))
        return aln, cls

    def simulate_continuous(self, seed, g0, n, gamma_alpha, p_invariant=0.0):
        """simulations.continuous = yes (CoMap.cpp:146, 213) -> (aln uint8 [T, n], rates float64 [n])"""
        aln = np.zeros((self.T, n), dtype=np.uint8)
        rates = np.zeros(n)
        self._check(self._lib.cmx_simulate_continuous(self._ctx, ctypes.c_uint64(seed), ctypes.c_uint64(g0), _sz(n),
                                                      ctypes.c_double(gamma_alpha), ctypes.c_double(p_invariant), _vp(aln),
                                                      _vp(rates)))
        return aln, rates

    def null_intra_continuous(self, kind, seed, rep_begin, rep_end, rep_ram, gamma_alpha, p_invariant=0.0, threshold=0.99,
                              mean_vectors=None):
        """AnalysisTools::getNullDistributionIntraDR with a continuous-rate simulator (simulations.continuous = yes): the
        replicates' alignments come from simulate_continuous (global site index g = ((rep * 2 + h) * rep_ram + j), as
        everywhere), the re-mapping and scoring from the fused null kernel on supplied alignments."""
        n = (rep_end - rep_begin) * rep_ram
        stat, pr, nm = np.zeros(n), np.zeros(n), np.zeros(n)
        rc = np.zeros(n, dtype=np.int32)
        params = _stat_params(kind, threshold, mean_vectors)
        # simulator and mapping both on the device, the alignments never leave it (cmx_null_intra_continuous)
        self._check(self._lib.cmx_null_intra_continuous(self._ctx, int(kind), _vp(params), ctypes.c_uint64(seed), _sz(rep_begin),
                                                        _sz(rep_end), _sz(rep_ram), ctypes.c_double(gamma_alpha),
                                                        ctypes.c_double(p_invariant), _vp(stat), _vp(rc), _vp(pr), _vp(nm)))
        return dict(stat=stat, rcmin=rc, prmin=pr, nmin=nm)

    def null_intra_continuous_via_host(self, kind, seed, rep_begin, rep_end, rep_ram, gamma_alpha, p_invariant=0.0,
                                       threshold=0.99, mean_vectors=None):
        """the same null assembled from the two public pieces through host memory (simulate_continuous, then null_intra on
        supplied alignments): what null_intra_continuous must equal bit for bit"""
        nrep = rep_end - rep_begin
        aln, _ = self.simulate_continuous(seed, rep_begin * 2 * rep_ram, nrep * 2 * rep_ram, gamma_alpha, p_invariant)
        sup = np.ascontiguousarray(aln.reshape(self.T, nrep, 2, rep_ram).transpose(1, 2, 0, 3))
        return self.null_intra(kind, seed, rep_begin, rep_end, rep_ram, supplied=sup, threshold=threshold,
                               mean_vectors=mean_vectors)

    def pair_stats(self, kind, counts1, counts2=None, threshold=0.99, mean_vectors=None):
        c1 = _f64(counts1).reshape(len(counts1), self.B, self.K)
        n1 = c1.shape[0]
        c2 = None if counts2 is None else _f64(counts2).reshape(len(counts2), self.B, self.K)
        n2 = n1 if c2 is None else c2.shape[0]
        out = np.zeros((n1, n2))
        params = _stat_params(kind, threshold, mean_vectors)
        self._check(self._lib.cmx_pair_stats(self._ctx, int(kind), _vp(params), _vp(c1), _sz(n1), _vp(c2), _sz(n2),
                                             _vp(out)))
        return out

    def null_intra(self, kind, seed, rep_begin, rep_end, rep_ram, supplied=None, threshold=0.99, mean_vectors=None):
        n = (rep_end - rep_begin) * rep_ram
        stat, prmin, nmin = np.zeros(n), np.zeros(n), np.zeros(n)
        rcmin = np.zeros(n, dtype=np.int32)
        sup = None
        if supplied is not None:
            sup = np.ascontiguousarray(supplied, dtype=np.uint8)
            if sup.shape != (rep_end - rep_begin, 2, self.T, rep_ram):
                raise CmxError(-1, "supplied alignments must be [nrep, 2, T, rep_ram]")
        params = _stat_params(kind, threshold, mean_vectors)
        self._check(self._lib.cmx_null_intra(self._ctx, int(kind), _vp(params), ctypes.c_uint64(seed), _sz(rep_begin),
                                             _sz(rep_end), _sz(rep_ram), _vp(sup), _vp(stat), _vp(rcmin), _vp(prmin),
                                             _vp(nmin)))
        return dict(stat=stat, rcmin=rcmin, prmin=prmin, nmin=nmin)

    def null_inter(self, other, kind, seed, rep_begin, rep_end, rep_ram, threshold=0.99, mean_vectors=None):
        """AnalysisTools::getNullDistributionInterDR: self = data set 1, other = data set 2 (same branches)."""
        n = (rep_end - rep_begin) * rep_ram
        stat, prmin, nmin = np.zeros(n), np.zeros(n), np.zeros(n)
        rcmin = np.zeros(n, dtype=np.int32)
        params = _stat_params(kind, threshold, mean_vectors)
        self._check(self._lib.cmx_null_inter(self._ctx, other._ctx, int(kind), _vp(params), ctypes.c_uint64(seed),
                                             _sz(rep_begin), _sz(rep_end), _sz(rep_ram), _vp(stat), _vp(rcmin),
                                             _vp(prmin), _vp(nmin)))
        return dict(stat=stat, rcmin=rcmin, prmin=prmin, nmin=nmin)

    def vector_matrix(self, kind, v1, v2=None, independent=False):
        """AnalysisTools::compute{ScalarProduct,Cosinus,Correlation,Covariance}Matrix (AnalysisTools.cpp:102-339) for plain
        vectors v1 [n1, dim] (and v2 [n2, dim]); kind = STAT_SCALAR_PRODUCT / STAT_COSINUS / STAT_CORRELATION / STAT_COVARIANCE"""
        a = np.ascontiguousarray(v1, dtype=np.float64)
        b = None if v2 is None else np.ascontiguousarray(v2, dtype=np.float64)
        n1, dim = a.shape
        n2 = n1 if b is None else b.shape[0]
        out = np.zeros((n1, n2))
        self._check(self._lib.cmx_vector_matrix(self._ctx, int(kind), _sz(dim), _vp(a), _sz(n1), _vp(b), _sz(n2),
                                                ctypes.c_int(int(bool(independent))), _vp(out)))
        return out

    def intra_pvalues(self, stat, norms, nclasses, null_stat, null_nmin):
        stat, norms, ns, nm = _f64(stat), _f64(norms), _f64(null_stat), _f64(null_nmin)
        n = len(norms)
        pv = np.zeros((n, n))
        nsim = np.zeros((n, n), dtype=np.int32)
        self._check(self._lib.cmx_intra_pvalues(self._ctx, _vp(stat), _vp(norms), _sz(n), int(nclasses), _vp(ns),
                                                _vp(nm), _sz(len(ns)), _vp(pv), _vp(nsim)))
        return pv, nsim

    def intra_rows(self, kind, counts, rate_class, post_rate, norm, null_stat=None, null_nmin=None, nclasses=10,
                   filters=None, capacity=None, threshold=0.99, mean_vectors=None):
        """statistics.txt rows (structured array PAIR_ROW, reference order), compacted on the device."""
        c = _f64(counts).reshape(len(counts), self.B, self.K)
        n = c.shape[0]
        rc = np.ascontiguousarray(rate_class, dtype=np.int32)
        pr, nm = _f64(post_rate), _f64(norm)
        ns = None if null_stat is None else _f64(null_stat)
        nn = None if null_nmin is None else _f64(null_nmin)
        cap = n * (n - 1) // 2 if capacity is None else int(capacity)
        rows = np.zeros(max(cap, 1), dtype=PAIR_ROW)
        count = ctypes.c_uint64(0)
        f = filters if filters is not None else PairFilters()
        params = _stat_params(kind, threshold, mean_vectors)
        self._check(self._lib.cmx_intra_rows(self._ctx, int(kind), _vp(params), _vp(c), _sz(n), _vp(rc), _vp(pr), _vp(nm),
                                             _vp(ns), _vp(nn), _sz(0 if ns is None else len(ns)), int(nclasses),
                                             ctypes.byref(f), _vp(rows), _sz(cap), ctypes.byref(count)))
        return rows[: min(cap, count.value)], count.value

    def inter_rows(self, kind, m1, m2, filters=None, capacity=None, threshold=0.99, mean_vectors=None):
        """rows of the inter-gene statistics file (CoETools::computeInterStats, CoETools.cpp:786-828), compacted on the
        device.  m1 / m2: dicts with counts [N, B, K], rate_class, post_rate, norm (what map_sites returns)."""
        c1 = _f64(m1["counts"]).reshape(len(m1["counts"]), self.B, self.K)
        c2 = _f64(m2["counts"]).reshape(len(m2["counts"]), self.B, self.K)
        n1, n2 = c1.shape[0], c2.shape[0]
        f = filters if filters is not None else InterFilters()
        cap = (n1 if f.independent_comparisons else n1 * n2) if capacity is None else int(capacity)
        rows = np.zeros(max(cap, 1), dtype=PAIR_ROW)
        count = ctypes.c_uint64(0)
        params = _stat_params(kind, threshold, mean_vectors)
        args = []
        for m in (m1, m2):
            args += [np.ascontiguousarray(m["rate_class"], dtype=np.int32), _f64(m["post_rate"]), _f64(m["norm"])]
        self._check(self._lib.cmx_inter_rows(self._ctx, int(kind), _vp(params), _vp(c1), _sz(n1), _vp(args[0]), _vp(args[1]),
                                             _vp(args[2]), _vp(c2), _sz(n2), _vp(args[3]), _vp(args[4]), _vp(args[5]),
                                             ctypes.byref(f), _vp(rows), _sz(cap), ctypes.byref(count)))
        return rows[: min(cap, count.value)], count.value

    def mi_columns(self, aln1, aln2=None, nalpha=20, masks=None):
        a1 = np.ascontiguousarray(aln1, dtype=np.uint8)
        T, n1 = a1.shape
        a2 = None if aln2 is None else np.ascontiguousarray(aln2, dtype=np.uint8)
        n2 = n1 if a2 is None else a2.shape[1]
        mi, hj = np.zeros((n1, n2)), np.zeros((n1, n2))
        h1, h2 = np.zeros(n1), np.zeros(n2)
        mk = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint32)
        self._check(self._lib.cmx_mi_columns(self._ctx, int(nalpha), int(T), _vp(mk), _sz(0 if mk is None else len(mk)),
                                             _vp(a1), _sz(n1), _vp(a2), _sz(n2), _vp(mi), _vp(hj), _vp(h1), _vp(h2)))
        return dict(mi=mi, hjoint=hj, h1=h1, h2=h2)

    def mi_pairs(self, aln1, idx1, idx2, aln2=None, nalpha=20, masks=None):
        """MI / joint entropy of the listed column pairs (Mica's bootstrap nulls, Mica.cpp:399-548)."""
        a1 = np.ascontiguousarray(aln1, dtype=np.uint8)
        T, n1 = a1.shape
        a2 = None if aln2 is None else np.ascontiguousarray(aln2, dtype=np.uint8)
        i1, i2 = np.ascontiguousarray(idx1, dtype=np.int64), np.ascontiguousarray(idx2, dtype=np.int64)
        mi, hj = np.zeros(len(i1)), np.zeros(len(i1))
        mk = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint32)
        self._check(self._lib.cmx_mi_pairs(self._ctx, int(nalpha), int(T), _vp(mk), _sz(0 if mk is None else len(mk)),
                                           _vp(a1), _sz(n1), _vp(a2), _sz(0 if a2 is None else a2.shape[1]), _vp(i1),
                                           _vp(i2), _sz(len(i1)), _vp(mi), _vp(hj)))
        return dict(mi=mi, hjoint=hj)

    def mica_parametric_null(self, seed, nrep_cpu, nrep_ram, with_norms=False, gamma_alpha=0.0, p_invariant=0.0):
        """null.method = parametric-bootstrap (Mica.cpp:469-548) in one call (cmx_mica_parametric_null) -> dict(mi, hjoint[, nmin])"""
        n = nrep_cpu * nrep_ram
        mi, hj = np.zeros(n), np.zeros(n)
        nm = np.zeros(n) if with_norms else None
        self._check(self._lib.cmx_mica_parametric_null(self._ctx, int(self.S), ctypes.c_uint64(seed), _sz(nrep_cpu), _sz(nrep_ram),
                                                       ctypes.c_double(gamma_alpha), ctypes.c_double(p_invariant), _vp(mi), _vp(hj),
                                                       _vp(nm)))
        out = dict(mi=mi, hjoint=hj)
        if with_norms:
            out["nmin"] = nm
        return out

    # -- groups of sites / candidate-group test
    @staticmethod
    def _flatten_groups(groups):
        off = np.zeros(len(groups) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(g) for g in groups])
        flat = np.array([x for g in groups for x in g], dtype=np.int32)
        return off, flat

    def group_stats(self, kind, counts, groups, threshold=0.99, mean_vectors=None):
        """Statistic::getValueForGroup for each list of site indices in `groups` (Statistics.h:121-133, :267-294)"""
        c = _f64(counts)
        off, flat = self._flatten_groups(groups)
        out = np.zeros(len(groups))
        params = _stat_params(kind, threshold, mean_vectors)
        self._check(self._lib.cmx_group_stats(self._ctx, int(kind), _vp(params), _vp(c), _sz(c.shape[0]), _vp(off), _vp(flat),
                                              _sz(len(groups)), _vp(out)))
        return out

    def candidate_groups(self, kind, norm_windows, analysable, observed, min_sim, rep_ram, max_trials, seed, max_batches=0,
                         threshold=0.99, mean_vectors=None):
        """CoETools::computePValuesForCandidateGroups.  norm_windows: per group a list of (lo, hi) per candidate site.
        -> dict(n1, n2, pvalue, trials, batches)"""
        off = np.zeros(len(norm_windows) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(g) for g in norm_windows])
        lo = _f64([w[0] for g in norm_windows for w in g])
        hi = _f64([w[1] for g in norm_windows for w in g])
        G = len(norm_windows)
        ok = np.ascontiguousarray(analysable, dtype=np.uint8)
        obs = _f64(observed)
        n1, n2 = np.zeros(G, dtype=np.uint32), np.zeros(G, dtype=np.uint32)
        trials, batches = ctypes.c_uint32(0), ctypes.c_uint64(0)
        params = _stat_params(kind, threshold, mean_vectors)
        self._check(self._lib.cmx_candidate_groups(self._ctx, int(kind), _vp(params), _sz(G), _vp(off), _vp(lo), _vp(hi), _vp(ok),
                                                   _vp(obs), ctypes.c_uint32(min_sim), _sz(rep_ram), ctypes.c_uint32(max_trials),
                                                   ctypes.c_uint64(max_batches), ctypes.c_uint64(seed), _vp(n1), _vp(n2),
                                                   ctypes.byref(trials), ctypes.byref(batches)))
        return dict(n1=n1, n2=n2, pvalue=(n1 + 1.0) / (n2 + 1.0), trials=trials.value, batches=batches.value)

    # -- Mica after the all-pairs matrix
    def mica_average_mi(self, mi):
        """-> (averageMI [n], fullAverageMI) (Mica.cpp:346-363); mi: [n, n], upper triangle read"""
        m = _f64(mi)
        n = m.shape[0]
        avg, full = np.zeros(n), np.zeros(1)
        self._check(self._lib.cmx_mica_average_mi(self._ctx, _vp(m), _sz(n), _vp(avg), _vp(full)))
        return avg, float(full[0])

    def mica_permutation_test(self, aln, max_perm, seed, nalpha=20, masks=None):
        """miTest of every column pair (Mica.cpp:93-118) -> (pvalue, nperm) [n(n-1)/2] in (i < j) row order.  masks: table
        indexed by alignment code (bit a = compatible with state a); codes >= nalpha without an entry are unknowns."""
        a = np.ascontiguousarray(aln, dtype=np.uint8)
        T, n = a.shape
        npairs = n * (n - 1) // 2
        pv, npm = np.zeros(npairs), np.zeros(npairs, dtype=np.int32)
        mk = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint32)
        self._check(self._lib.cmx_mica_permutation_test_masks(self._ctx, int(nalpha), int(T), _vp(mk) if mk is not None else None,
                                                              _sz(0 if mk is None else len(mk)), _vp(a), _sz(n),
                                                              ctypes.c_uint32(max_perm), ctypes.c_uint64(seed), _vp(pv), _vp(npm)))
        return pv, npm

    def mica_zscore_null(self, which, mi, key):
        """null.method = z-score (Mica.cpp:549-607) -> (null_stat, null_key) [n(n-1)/2]; which: MICA_MI / MIP / MIC"""
        m, k = _f64(mi), _f64(key)
        n = m.shape[0]
        ns, nk = np.zeros(n * (n - 1) // 2), np.zeros(n * (n - 1) // 2)
        self._check(self._lib.cmx_mica_zscore_null(self._ctx, int(which), _vp(m), _sz(n), _vp(k), _vp(ns), _vp(nk)))
        return ns, nk

    # -- clustering analysis
    def hclust(self, dist, linkage):
        """dist [n, n] or [batch, n, n] -> dict(merge [.., n-1, 2], dmax, size)"""
        d = _f64(dist)
        single = d.ndim == 2
        d = d.reshape((-1,) + d.shape[-2:])
        batch, n = d.shape[0], d.shape[1]
        merge = np.zeros((batch, n - 1, 2), dtype=np.int32)
        dmax = np.zeros((batch, n - 1))
        size = np.zeros((batch, n - 1), dtype=np.int32)
        self._check(self._lib.cmx_hclust(self._ctx, int(linkage), _vp(d), _sz(n), _sz(batch), _vp(merge), _vp(dmax), _vp(size)))
        out = dict(merge=merge, dmax=dmax, size=size)
        return {k: v[0] for k, v in out.items()} if single else out

    def cluster_sites(self, dist_kind, linkage, counts, want_dist=True):
        """counts [N, B, K] -> dict(dist [N, N], merge, dmax, size, stat, nmin) (CoMap.cpp:432-550)"""
        c = _f64(counts)
        n = c.shape[0]
        dist = np.zeros((n, n)) if want_dist else None
        merge = np.zeros((n - 1, 2), dtype=np.int32)
        size = np.zeros(n - 1, dtype=np.int32)
        dmax, stat, nmin = np.zeros(n - 1), np.zeros(n - 1), np.zeros(n - 1)
        self._check(self._lib.cmx_cluster_sites(self._ctx, int(dist_kind), int(linkage), _vp(c), _sz(n), _vp(dist), _vp(merge),
                                                _vp(dmax), _vp(size), _vp(stat), _vp(nmin)))
        return dict(dist=dist, merge=merge, dmax=dmax, size=size, stat=stat, nmin=nmin)

    def cluster_null(self, dist_kind, linkage, seed, rep_begin, rep_end, nsites):
        """ClusterTools::computeGlobalDistanceDistribution: dict of [nrep, nsites-1] arrays (merge: [.., 2])"""
        nrep, nm = rep_end - rep_begin, nsites - 1
        merge = np.zeros((nrep, nm, 2), dtype=np.int32)
        size = np.zeros((nrep, nm), dtype=np.int32)
        dmax, stat, nmin = np.zeros((nrep, nm)), np.zeros((nrep, nm)), np.zeros((nrep, nm))
        self._check(self._lib.cmx_cluster_null(self._ctx, int(dist_kind), int(linkage), ctypes.c_uint64(seed), _sz(rep_begin),
                                               _sz(rep_end), _sz(nsites), _vp(merge), _vp(dmax), _vp(size), _vp(stat), _vp(nmin)))
        return dict(merge=merge, dmax=dmax, size=size, stat=stat, nmin=nmin)

    # -- device-pointer entry points (torch CUDA tensors, engine-native layouts, asynchronous)
    def map_sites_dev(self, d_aln, counts=None, logL=None, post_rate=None, rate_class=None, norm=None, masks=None):
        """d_aln: uint8 [T, ld] CUDA tensor; counts: float64 [B*K, ldc]; per-site outputs: [N]."""
        n = d_aln.shape[1]
        self._check(self._lib.cmx_map_sites_dev(self._ctx, _vp(d_aln), _sz(n), _sz(d_aln.stride(0)), _vp(masks),
                                                _vp(counts), _sz(0 if counts is None else counts.stride(0)), _vp(logL),
                                                _vp(post_rate), _vp(rate_class), _vp(norm), self._stream()))

    def simulate_dev(self, seed, g0, n, aln, classes=None):
        """aln: uint8 [T, ld] CUDA tensor (ld >= n)"""
        self._check(self._lib.cmx_simulate_dev(self._ctx, ctypes.c_uint64(seed), ctypes.c_uint64(g0), _sz(n), _vp(aln),
                                               _sz(aln.stride(0)), _vp(classes), self._stream()))

    def mi_pairs_dev(self, d_aln1, idx1, idx2, mi, hjoint, d_aln2=None, nalpha=20, masks=None):
        """MI / joint entropy of the listed column pairs, everything on the device (int64 index tensors, float64 outputs)"""
        self._check(self._lib.cmx_mi_pairs_dev(self._ctx, int(nalpha), int(d_aln1.shape[0]), _vp(masks), _vp(d_aln1),
                                               _sz(d_aln1.shape[1]), _sz(d_aln1.stride(0)), _vp(d_aln2),
                                               _sz(0 if d_aln2 is None else d_aln2.shape[1]),
                                               _sz(0 if d_aln2 is None else d_aln2.stride(0)), _vp(idx1), _vp(idx2),
                                               _sz(idx1.numel()), _vp(mi), _vp(hjoint), self._stream()))

    def pair_stats_dev(self, kind, counts1, out, counts2=None, threshold=0.99, mean_vectors=None):
        params = _stat_params(kind, threshold, mean_vectors)
        n1 = counts1.shape[1]
        n2 = n1 if counts2 is None else counts2.shape[1]
        self._check(self._lib.cmx_pair_stats_dev(self._ctx, int(kind), _vp(params), _vp(counts1), _sz(n1),
                                                 _sz(counts1.stride(0)), _vp(counts2), _sz(n2),
                                                 _sz(0 if counts2 is None else counts2.stride(0)), _vp(out),
                                                 _sz(out.stride(0)), self._stream()))

    def null_simulate_dev(self, seed, rep_begin, rep_end, rep_ram, aln):
        """the null's simulated alignments into `aln` (uint8 CUDA tensor, >= nrep * 2 * T * rep_ram bytes, laid out
        [replicate][batch][taxon][rep_ram]): what null_intra_dev takes as `supplied`"""
        self._check(self._lib.cmx_null_simulate_dev(self._ctx, ctypes.c_uint64(seed), _sz(rep_begin), _sz(rep_end), _sz(rep_ram),
                                                    _vp(aln), self._stream()))

    def null_intra_dev(self, kind, seed, rep_begin, rep_end, rep_ram, stat, rcmin=None, prmin=None, nmin=None,
                       supplied=None, threshold=0.99, mean_vectors=None):
        params = _stat_params(kind, threshold, mean_vectors)
        self._check(self._lib.cmx_null_intra_dev(self._ctx, int(kind), _vp(params), ctypes.c_uint64(seed),
                                                 _sz(rep_begin), _sz(rep_end), _sz(rep_ram), _vp(supplied), _vp(stat),
                                                 _vp(rcmin), _vp(prmin), _vp(nmin), self._stream()))

    def null_inter_dev(self, other, kind, seed, rep_begin, rep_end, rep_ram, stat, rcmin=None, prmin=None, nmin=None,
                       threshold=0.99, mean_vectors=None):
        params = _stat_params(kind, threshold, mean_vectors)
        self._check(self._lib.cmx_null_inter_dev(self._ctx, other._ctx, int(kind), _vp(params), ctypes.c_uint64(seed),
                                                 _sz(rep_begin), _sz(rep_end), _sz(rep_ram), _vp(stat), _vp(rcmin),
                                                 _vp(prmin), _vp(nmin), self._stream()))

    def intra_pvalues_dev(self, stat, norms, nclasses, null_stat, null_nmin, pvalue, nsim):
        n = norms.shape[0]
        nnull = 0 if null_stat is None else null_stat.shape[0]
        self._check(self._lib.cmx_intra_pvalues_dev(self._ctx, _vp(stat), _sz(stat.stride(0)), _vp(norms), _sz(n),
                                                    int(nclasses), _vp(null_stat), _vp(null_nmin), _sz(nnull),
                                                    _vp(pvalue), _vp(nsim), self._stream()))

    def mi_columns_dev(self, d_aln1, mi, hjoint, d_aln2=None, nalpha=20, masks=None, h1=None, h2=None):
        T, n1 = d_aln1.shape
        n2 = n1 if d_aln2 is None else d_aln2.shape[1]
        self._check(self._lib.cmx_mi_columns_dev(self._ctx, int(nalpha), int(T), _vp(masks), _vp(d_aln1), _sz(n1),
                                                 _sz(d_aln1.stride(0)), _vp(d_aln2), _sz(n2),
                                                 _sz(0 if d_aln2 is None else d_aln2.stride(0)), _vp(mi), _vp(hjoint),
                                                 _sz(mi.stride(0)), _vp(h1), _vp(h2), self._stream()))

    def intra_compact_range_dev(self, kind, counts, norm, null_stat, null_nmin, nclasses, out, row_begin=0, row_end=None,
                                threshold=0.99, mean_vectors=None):
        """the UNFILTERED pair loop for rows [row_begin, row_end) as PAIR_COMPACT records (16 B per pair, (i, j) order) in the
        CUDA uint8 tensor `out`; expand_compact_rows rebuilds the PAIR_ROW records on the host"""
        n = norm.shape[0]
        row_end = n if row_end is None else int(row_end)
        nnull = 0 if null_stat is None else null_stat.shape[0]
        params = _stat_params(kind, threshold, mean_vectors)
        self._check(self._lib.cmx_intra_compact_range_dev(
            self._ctx, int(kind), _vp(params), _vp(counts), _sz(n), _sz(counts.stride(0)), _vp(norm), _vp(null_stat), _vp(null_nmin),
            _sz(nnull), int(nclasses), _sz(row_begin), _sz(row_end), _vp(out), _sz(out.numel() // PAIR_COMPACT.itemsize), self._stream()))

    def intra_gram_prefetch_dev(self, kind, counts, n, row_begin=0, row_end=None):
        """enqueue the statistics of the observed pairs of rows [row_begin, row_end) on the current stream and keep them for the
        next intra_compact_range_dev with the same arguments (cmx_intra_gram_prefetch_dev: beside the null, not behind it)"""
        row_end = int(n) if row_end is None else int(row_end)
        self._check(self._lib.cmx_intra_gram_prefetch_dev(self._ctx, int(kind), _vp(counts), _sz(n), _sz(counts.stride(0)),
                                                          _sz(row_begin), _sz(row_end), self._stream()))

    def intra_rows_range_dev(self, kind, counts, rate_class, post_rate, norm, null_stat, null_nmin, nclasses, rows, count,
                             row_begin=0, row_end=None, filters=None, threshold=0.99, mean_vectors=None):
        """CoETools::computeIntraStats' pair loop for rows [row_begin, row_end) of the upper triangle, everything on the
        device and no N x N matrix: counts float64 [B*K, ldc] CUDA (as map_sites_dev writes them), rows = CUDA uint8 tensor
        of capacity * 48 bytes (PAIR_ROW records, reference (i, j) order), count = CUDA int64 tensor [1]."""
        n = norm.shape[0]
        row_end = n if row_end is None else int(row_end)
        nnull = 0 if null_stat is None else null_stat.shape[0]
        f = filters if filters is not None else PairFilters()
        params = _stat_params(kind, threshold, mean_vectors)
        cap = rows.numel() // PAIR_ROW.itemsize
        self._check(self._lib.cmx_intra_rows_range_dev(
            self._ctx, int(kind), _vp(params), _vp(counts), _sz(n), _sz(counts.stride(0)), _vp(rate_class), _vp(post_rate),
            _vp(norm), _vp(null_stat), _vp(null_nmin), _sz(nnull), int(nclasses), ctypes.byref(f), _sz(row_begin),
            _sz(row_end), _vp(rows), _sz(cap), _vp(count), self._stream()))
