/* comap_mi355x.h -- C-ABI of the MI355X-native substitution-mapping + pairwise-coevolution engine.
 *
 * Drop-in boundary for ONE hot path of jydu/comap (paths below are relative to the reference tree):
 *   CoETools::getVectors                         CoMap/CoETools.h:317-322, CoMap/CoETools.cpp:366-416
 *   AnalysisTools::computeNorms                  CoMap/AnalysisTools.h:198,   CoMap/AnalysisTools.cpp:343-350
 *   AnalysisTools::getNullDistributionIntraDR    CoMap/AnalysisTools.h:248-260, CoMap/AnalysisTools.cpp:564-658
 *   Statistic::getValueForPair (all-pairs form)  CoMap/Statistics.h:72, loops CoMap/CoETools.cpp:672-724, 786-828
 *   Distance::getDistanceForPair                 CoMap/Distance.h:71-72, loops CoMap/CoMap.cpp:432-440
 *   Mica all-pairs MI                            CoMap/Mica.cpp:93-118, 346-361, 646-689
 * The reference has no FFI layer of its own (its seams are C++ static methods over Bio++ types), so this header
 * is what a binding for that path would bind; INTEGRATION.md shows the C++ adapter a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no exceptions cross the boundary: every call returns a cmx_status and
 *     cmx_last_error(ctx) holds the message (the reference throws bpp::Exception, caught in main).
 *   - one ctx per GPU; calls on one ctx are serialised by the caller (the reference is single-threaded).
 *   - tree: nodes in post-order, root last, parent[root] = -1.  Branch index b == id of the branch's lower node,
 *     i.e. the row order of the reference's .vec files (LegacySubstitutionMappingTools::writeToStream).
 *   - alignment: uint8 codes, taxon-major: code of (taxon t, site i) at aln[t * ld + i].  code < S is a state;
 *     code >= S indexes `masks` (bit z set <=> state z is compatible; X and gaps = all ones).
 *   - "host" entry points take host pointers in the reference's layouts (counts site-major [N][B][K] ==
 *     mapping[i][b][k], CoMap/Statistics.h:154-160).  "_dev" entry points take device pointers in the engine's
 *     native layouts (counts branch-major [B*K][ld], one row per (branch, type), sites contiguous) and a
 *     hipStream_t passed as void*; they never synchronise the device.
 */
#ifndef COMAP_MI355X_H
#define COMAP_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cmx_ctx cmx_ctx;

typedef enum {
  CMX_OK = 0,
  CMX_ERR_INVALID = -1,      /* bad argument (reference: bpp::Exception / DimensionException) */
  CMX_ERR_UNSUPPORTED = -2,  /* e.g. nstates > 64 */
  CMX_ERR_DEVICE = -3,       /* HIP error; no CPU fallback exists */
  CMX_ERR_NOMEM = -4,
  CMX_ERR_INTERNAL = -5      /* a self-check of the engine failed (CMX_SCRATCH_GUARD: a device buffer was written past its end) */
} cmx_status;

/* statistic kinds: CoMap/Statistics.h:164-329, factory CoMap/CoETools.cpp:535-600 */
typedef enum {
  CMX_STAT_CORRELATION = 0,    /* Statistics.h:164-174 */
  CMX_STAT_COMPENSATION = 1,   /* Statistics.h:247-265 */
  CMX_STAT_COSUBSTITUTION = 2, /* Statistics.h:230-245 */
  CMX_STAT_COSINUS = 3,        /* Statistics.h:218-228 */
  CMX_STAT_COVARIANCE = 4,     /* Statistics.h:206-216 */
  CMX_STAT_DISCRETE_MI = 5,    /* Statistics.h:307-327 with bounds {0, threshold, 10000} (CoETools.cpp:590-593) */
  CMX_STAT_EUCLIDIAN_DISTANCE = 7, /* Distance.h:157-171: sqrt(sum_b (total2_b - total1_b)^2), a distance (clustering) */
  CMX_STAT_CORRECTED_CORRELATION = 6, /* Statistics.h:176-204: correlation after subtracting a per-branch mean vector from
                                        either operand; params = [2][nbranches] (meanVector1_, meanVector2_; CoMap.cpp:350-359
                                        sets both to the mean total substitution vector of the data) */
  CMX_STAT_SCALAR_PRODUCT = 9,       /* the raw Gram: sum_b v1_b v2_b on the type-0 counts (VectorTools::scalar,
                                      * AnalysisTools::computeScalarProductMatrix, AnalysisTools.cpp:102-158) */
  CMX_STAT_DISCRETE_MI_BOUNDS = 8    /* Statistics.h:307-327 with ANY bounds vector (DiscreteMutualInformationStatistic(const
                                        Vdouble& bounds)): params = [nbounds, b_0 .. b_{nbounds-1}], non-decreasing; per-branch
                                        totals are binned with Domain(bounds)::getIndex (Domain.cpp:113-122).  The bounds of
                                        nijt = Label (CoETools.cpp:577-588): -0.5, 0.5, .., S(S-1) + 0.5.  A total outside
                                        [b_0, b_last) makes every statistic of that site NaN (reference: OutOfRangeException).
                                        Nulls with this statistic run unfused (simulate -> map -> score).  At most 4096
                                        branches. */
} cmx_stat_kind;

/* substitution-count flavour (SubstitutionCountInterface::getAllNumbersOfSubstitutions, CoMap/CoMap.cpp:152) */
typedef enum {
  CMX_COUNT_EXPECTED = 0, /* Uniformization == Decomposition (exact conditional expectation) */
  CMX_COUNT_NAIVE = 1     /* N(x,y) = W(x,y) [x != y] */
} cmx_count_method;

typedef struct {
  int32_t nstates;             /* S: 4 and 20 run on the matrix cores; any other S in 2..64 (codon alphabets,
                                * CoETools.cpp:95-100) on plain kernels: same results API, no ambiguity table (every
                                * code >= S is an unknown), every null through the unfused sequence */
  int32_t nclasses;            /* C: discrete rate classes */
  int32_t ntypes;              /* K: substitution types (1 for the total register) */
  const double* Q;             /* [S*S] row-major generator, reversible w.r.t. pi */
  const double* pi;            /* [S] */
  const double* rates;         /* [C] */
  const double* probs;         /* [C] */
  const double* Bk;            /* [K*S*S] Q o register_k o weights, zero diagonal; NULL => K = 1, unweighted total */
  int32_t count_method;        /* cmx_count_method */
  int32_t clamp_negative;      /* 1: clamp negative conditional counts to 0 (Bio++ rule for unweighted counts) */
  const double* naive_weights; /* [S*S] for CMX_COUNT_NAIVE, NULL => 1 */
  /* Non-homogeneous models (nonhomogeneous = one_per_branch | general, CoMap/CoETools.cpp:126-206:
   * DRNonHomogeneousTreeLikelihood over a SubstitutionModelSet): nmodels > 0 gives every branch its own generator.
   * Leave all six fields zero for the homogeneous case above (zero-initialise the struct). */
  int32_t nmodels;                /* number of generators in the set */
  const double* Qs;               /* [nmodels][S*S], each reversible w.r.t. its row of pis */
  const double* pis;              /* [nmodels][S] equilibrium frequencies of each generator */
  const double* Bks;              /* [nmodels][K][S*S] registers per generator; NULL => K = 1, unweighted total */
  const int32_t* model_of_branch; /* [nnodes] generator of the branch above each node (root entry ignored) */
  const double* root_freqs;       /* [S] frequencies at the root (the model set's root frequency set) */
} cmx_model;

typedef struct {
  int32_t nnodes;
  const int32_t* parent;        /* [nnodes], post-order, root last */
  const double* blen;           /* [nnodes], blen[root] ignored */
  int32_t ntaxa;
  const int32_t* leaf_of_taxon; /* [ntaxa] node id of each alignment row */
} cmx_tree;

typedef struct {
  int32_t nstates, nclasses, ntypes, nnodes, nbranches, ntaxa, ninternal;
  int32_t device, cu_count, waves; /* waves = resident mapping waves (workspace is sized for them) */
  size_t workspace_bytes;
  /* the mapping kernel's walk of ONE rate-class pass over one block of 64 sites (device classes: nucleotide models with
   * >= 4 classes run `device_classes` = ceil(C / 4 or 5) fused passes): matrix products issued, leaf gathers, S-vectors
   * loaded from / stored to the per-wave workspace -- what roofline accounting needs */
  int32_t device_states, device_classes;
  int32_t products_per_pass, leaf_ops_per_pass, ws_loads_per_pass, ws_stores_per_pass;
  /* the walk of the null's (fully resolved) alignments when it differs: class-fused nucleotide models take an inlined
   * cherry's message and outside visit from tables indexed by its two symbols (0 = same walk as above) */
  int32_t products_per_pass_null, leaf_ops_per_pass_null, cherry_tables;
} cmx_info;

const char* cmx_version(void);
cmx_status cmx_ctx_create(const cmx_model* model, const cmx_tree* tree, int device, cmx_ctx** out);
void cmx_ctx_destroy(cmx_ctx* ctx);
const char* cmx_last_error(const cmx_ctx* ctx); /* ctx may be NULL: last error of a failed cmx_ctx_create */
cmx_status cmx_get_info(const cmx_ctx* ctx, cmx_info* info);
/* transition probabilities the engine uses, for inspection/tests: P[C][B][S][S] (row x -> column y) */
cmx_status cmx_get_transition_matrices(const cmx_ctx* ctx, double* P);
cmx_status cmx_synchronize(cmx_ctx* ctx);
/* Scratch guard (debugging aid, no counterpart in the reference).  With CMX_SCRATCH_GUARD=1 in the environment when the
 * first context is created, or after cmx_debug_scratch_guard(1), every device buffer the engine sizes by hand (named
 * scratch buffers, per-wave workspaces, temporaries of the host-pointer entry points) is followed by a 4 KiB canary.
 * The engine verifies a buffer's canary before it hands the buffer out again; cmx_synchronize, cmx_scratch_check and
 * cmx_ctx_destroy verify all of them.  A trampled canary makes the call fail with CMX_ERR_INTERNAL and cmx_last_error
 * names the buffer.  Every check synchronises the device: not for timed runs. */
cmx_status cmx_scratch_check(cmx_ctx* ctx);                 /* CMX_ERR_UNSUPPORTED when the guard is off */
int cmx_debug_scratch_guard(int on);                        /* on < 0: query only; returns the previous state.  Set BEFORE cmx_ctx_create */
size_t cmx_debug_scratch_guard_failures(char* buf, size_t cap, int clear);   /* findings so far (process-wide), one per line; returns their number */
/* test hook: under the guard, pretend the named scratch buffer was asked for with only `bytes` bytes (the allocation keeps
 * its real size, so the kernel's own writes land on the canary instead of past the allocation).  bytes = 0 removes the
 * override, name = NULL removes all. */
void cmx_debug_scratch_shrink(const char* name, size_t bytes);
/* host-side only (no GPU needed): compile the tree into what the mapping kernel's walk of a rate-class pass reads
 * (comap_amd/csrc/cmx_walk.h) and copy it out for inspection/tests.  nrec: [nvisited][16] node records; ldsched:
 * workspace loads; msched: operator uses, pairs (matrix index in a class block, taxon or -1), in program order.  The walk
 * has already passed the engine's self-check (run numerically on the host against a direct pruning computation).
 * Each *_cap is the capacity (in int32) of the caller's buffer; sizes are returned in *_n. */
cmx_status cmx_debug_walk(const cmx_model* model, const cmx_tree* tree, int32_t* nrec, size_t nrec_cap, size_t* nrec_n,
                          int32_t* ldsched, size_t ld_cap, size_t* ld_n, int32_t* msched, size_t m_cap, size_t* m_n,
                          int32_t* slot_of_node /*[nnodes] or NULL*/,
                          uint64_t* stats /*[7] or NULL: workspace loads, stores, matrix products, leaf ops per pass; products and
                                            leaf ops of the cherry-table walk (resolved alignments of class-fused nucleotide
                                            models), cherries with tables (0: that walk is the first one)*/);

/* ---- substitution mapping: replaces DRHomogeneousTreeLikelihood::initialize + getLogLikelihoodPerSite /
 * getPosteriorRatePerSite / getRateClassWithMaxPostProbPerSite + computeSubstitutionVectors + computeNormForSite
 * as reached from CoETools::getVectors (CoETools.cpp:397) and AnalysisTools.cpp:592-612.  Any output may be NULL. */
cmx_status cmx_map_sites(cmx_ctx* ctx, const uint8_t* aln, size_t nsites, size_t ld, const uint32_t* masks,
                         size_t nmasks, double* counts /*[N][B][K]*/, double* logL, double* post_rate,
                         int32_t* rate_class, double* norm);
cmx_status cmx_map_sites_dev(cmx_ctx* ctx, const uint8_t* d_aln, size_t nsites, size_t ld, const uint32_t* d_masks,
                             double* d_counts /*[B*K][ldc]*/, size_t ldc, double* d_logL, double* d_post_rate,
                             int32_t* d_rate_class, double* d_norm, void* stream);
/* nijt.average / nijt.joint (CoMap/CoETools.cpp:393-406, CoMap/AnalysisTools.cpp:598-633: which of
 * LegacySubstitutionMappingTools::computeSubstitutionVectors{, NoAveraging, Marginal, NoAveragingMarginal} maps the
 * sites; "for benchmarking only" in the reference, but nijt = Label with the MI statistic requires average = no,
 * CoETools.cpp:577-588).  Default (1, 1).  Any other combination: every later cmx_map_sites* / cmx_null_* / clustering /
 * candidate call of this context returns the counts and norms of that variant -- (0, 1): the conditional count
 * N^k(x*, y*; t_b) of the most probable PAIR of ancestral states of each branch; (1, 0): the conditional counts weighted
 * with the product of the two ends' marginal posteriors per state and rate; (0, 0): N^k at the two ends' marginal
 * ancestral states -- through plain (slow) kernels; the null then runs unfused.  Likelihood, posterior rate and rate
 * class do not depend on the variant.  Parity unpinned: bpp-phyl is not in the reference tree. */
cmx_status cmx_set_mapping_options(cmx_ctx* ctx, int average, int joint);

/* ---- sequence simulator (NonHomogeneousSequenceSimulator::simulate, AnalysisTools.cpp:591): counter-based RNG,
 * global site indices g0 .. g0+n-1 (see DESIGN.md "RNG").  aln_out: [T][n]. */
cmx_status cmx_simulate(cmx_ctx* ctx, uint64_t seed, uint64_t g0, size_t n, uint8_t* aln_out, int32_t* classes_out);
/* the same into device memory: d_aln [T][ld] (ld >= n), d_classes [n] or NULL -- what Mica's parametric bootstrap feeds to
 * cmx_mi_pairs_dev without leaving the device */
cmx_status cmx_simulate_dev(cmx_ctx* ctx, uint64_t seed, uint64_t g0, size_t n, uint8_t* d_aln, size_t ld, int32_t* d_classes,
                            void* stream);
/* simulations.continuous = yes (CoMap.cpp:146, 213: NonHomogeneousSequenceSimulator::enableContinuousRates): every site
 * draws its rate from the continuous Gamma(alpha, beta = alpha) distribution -- with probability p_invariant the rate is 0
 * and the Gamma draw is divided by 1 - p_invariant, as Invariant(Gamma) does -- and every branch uses exp(Q r t) of that
 * rate.  Same counter RNG and site indexing as cmx_simulate.  rates_out (may be NULL): the drawn rates.  A null
 * distribution under continuous rates = this simulator + cmx_null_intra with `supplied` alignments (the mapping of the
 * simulated sites still uses the discrete classes, as in the reference). */
cmx_status cmx_simulate_continuous(cmx_ctx* ctx, uint64_t seed, uint64_t g0, size_t n, double gamma_alpha, double p_invariant,
                                   uint8_t* aln_out, double* rates_out);
cmx_status cmx_simulate_continuous_dev(cmx_ctx* ctx, uint64_t seed, uint64_t g0, size_t n, double gamma_alpha, double p_invariant,
                                       uint8_t* d_aln, size_t ld, double* d_rates, void* stream);

/* ---- all-pairs statistic.  counts2 == NULL: intra (CoETools.cpp:672-692), out[i*N1+j] filled for j > i, NaN
 * elsewhere.  Otherwise inter (CoETools.cpp:786-810), out[i*N2+j].  params: for CMX_STAT_DISCRETE_MI params[0] is
 * the threshold (bounds {0, threshold, 10000}); for CMX_STAT_DISCRETE_MI_BOUNDS [nbounds, bounds..]; for
 * CMX_STAT_CORRECTED_CORRELATION the two mean vectors; ignored otherwise (may be NULL). */
cmx_status cmx_pair_stats(cmx_ctx* ctx, int kind, const double* params, const double* counts1, size_t n1,
                          const double* counts2, size_t n2, double* out);
cmx_status cmx_pair_stats_dev(cmx_ctx* ctx, int kind, const double* params, const double* d_counts1, size_t n1,
                              size_t ld1, const double* d_counts2, size_t n2, size_t ld2, double* d_out, size_t ldo,
                              void* stream);

/* The simulated alignments of a null, on the device: [replicate][batch 0 / 1][taxon][rep_ram] bytes -- the two
 * seqSim.simulate(repRAM) calls of every replicate (AnalysisTools.cpp:591, 612).  cmx_null_intra_dev does this itself
 * when no alignments are supplied; a caller that wants the mapping launch alone on its clock (bench.py) fills a buffer
 * here and passes it as `d_supplied`. */
cmx_status cmx_null_simulate_dev(cmx_ctx* ctx, uint64_t seed, size_t rep_begin, size_t rep_end, size_t rep_ram,
                                 uint8_t* d_aln, void* stream);
/* ---- parametric-bootstrap null, replicates [rep_begin, rep_end) of AnalysisTools::getNullDistributionIntraDR
 * (AnalysisTools.cpp:587-653): per replicate two batches of rep_ram simulated sites are mapped and site j of
 * batch 1 is scored against site j of batch 2.  Outputs have (rep_end-rep_begin)*rep_ram entries, the four
 * columns of AnalysisTools.cpp:642.  supplied (optional): [rep_end-rep_begin][2][T][rep_ram] alignments to map
 * instead of simulating (deterministic cross-implementation checks).  Supplied alignments must be FULLY RESOLVED (every
 * code a state, as a simulator's output is): the null's kernel stages only the state rows of a leaf operator and, for
 * class-fused nucleotide models, takes cherries from tables indexed by their two symbols.  The host-pointer entry checks
 * it (CMX_ERR_INVALID); with the device-pointer entry it is the caller's contract.  Sharding replicates over GPUs gives
 * results independent of the number of shards. */
cmx_status cmx_null_intra(cmx_ctx* ctx, int kind, const double* params, uint64_t seed, size_t rep_begin,
                          size_t rep_end, size_t rep_ram, const uint8_t* supplied, double* stat, int32_t* rcmin,
                          double* prmin, double* nmin);
cmx_status cmx_null_intra_dev(cmx_ctx* ctx, int kind, const double* params, uint64_t seed, size_t rep_begin,
                              size_t rep_end, size_t rep_ram, const uint8_t* d_supplied, double* d_stat,
                              int32_t* d_rcmin, double* d_prmin, double* d_nmin, void* stream);

/* the same null under simulations.continuous = yes (CoMap.cpp:146, 213): the two batches of every replicate are drawn by
 * the continuous-rate simulator on the device and mapped there (cmx_simulate_continuous_dev + cmx_null_intra_dev with
 * supplied alignments in one call; results equal that sequence bit for bit) */
cmx_status cmx_null_intra_continuous(cmx_ctx* ctx, int kind, const double* params, uint64_t seed, size_t rep_begin, size_t rep_end,
                                     size_t rep_ram, double gamma_alpha, double p_invariant, double* stat, int32_t* rcmin,
                                     double* prmin, double* nmin);
cmx_status cmx_null_intra_continuous_dev(cmx_ctx* ctx, int kind, const double* params, uint64_t seed, size_t rep_begin, size_t rep_end,
                                         size_t rep_ram, double gamma_alpha, double p_invariant, double* d_stat, int32_t* d_rcmin,
                                         double* d_prmin, double* d_nmin, void* stream);

/* Inter-gene null: AnalysisTools::getNullDistributionInterDR (CoMap/AnalysisTools.cpp:662-735, driver
 * CoETools::computeInterNullDistribution CoETools.cpp:873-897).  Per replicate rep_ram sites are simulated and mapped
 * under ctx1 (data set 1) and under ctx2 (data set 2, its own model / branch lengths, same branches) and site j of
 * the one is scored against site j of the other; outputs are the four columns of the null file
 * (Stat, RCmin, PRmin, Nmin), [(rep_end - rep_begin) * rep_ram].  Both contexts must live on the same device. */
cmx_status cmx_null_inter(cmx_ctx* ctx1, cmx_ctx* ctx2, int kind, const double* params, uint64_t seed, size_t rep_begin,
                          size_t rep_end, size_t rep_ram, double* stat, int32_t* rcmin, double* prmin, double* nmin);
cmx_status cmx_null_inter_dev(cmx_ctx* ctx1, cmx_ctx* ctx2, int kind, const double* params, uint64_t seed,
                              size_t rep_begin, size_t rep_end, size_t rep_ram, double* d_stat, int32_t* d_rcmin,
                              double* d_prmin, double* d_nmin, void* stream);

/* ---- p-values of CoETools::computeIntraStats (CoETools.cpp:636-652, 695-721): null stats are binned by
 * Domain(0, max(norms), nclasses) on nmin (out-of-range and NaN dropped), sorted per class;
 * p = (nsim - #{null < stat} + 1)/(nsim + 1).  pvalue = NaN / nsim = 0 where the reference prints "NA\t0". */
cmx_status cmx_intra_pvalues(cmx_ctx* ctx, const double* stat /*[N*N]*/, const double* norms, size_t n,
                             int nclasses, const double* null_stat, const double* null_nmin, size_t nnull,
                             double* pvalue, int32_t* nsim);
cmx_status cmx_intra_pvalues_dev(cmx_ctx* ctx, const double* d_stat, size_t ldo, const double* d_norms, size_t n,
                                 int nclasses, const double* d_null_stat, const double* d_null_nmin, size_t nnull,
                                 double* d_pvalue, int32_t* d_nsim, void* stream);

/* ---- the rows of statistics.txt, compacted on the device: the pair loop of CoETools::computeIntraStats
 * (CoMap/CoETools.cpp:672-724) with its filters (:674-693: min rate class / rate per site, max differences per pair,
 * |stat| >= statistic.min) applied to the dense statistic (and p-value / Nsim) matrices; rows come out in the
 * reference's (i, j) order, j > i.  pvalue NaN == "NA" (Nsim 0).  *count receives the number of rows that pass;
 * at most `capacity` of them are written. */
typedef struct cmx_pair_filters {
  int32_t min_rate_class;      /* statistic.min_rate_class, CoETools.cpp:420-433 */
  int32_t max_rate_class_diff; /* < 0: off */
  double min_rate;             /* statistic.min_rate */
  double max_rate_diff;        /* < 0: off */
  double min_statistic;        /* statistic.min, CoETools.cpp:693 */
} cmx_pair_filters;
typedef struct cmx_pair_row {
  int32_t i, j;                /* site indices (the caller maps them to coordinates, CoETools.cpp:699-703) */
  double stat;
  int32_t rc_min, nsim;
  double pr_min, n_min, pvalue;
} cmx_pair_row;
cmx_status cmx_intra_rows_dev(cmx_ctx* ctx, const double* d_stat, size_t ldo, const double* d_pvalue /*or NULL*/,
                              const int32_t* d_nsim /*or NULL*/, size_t n, const int32_t* d_rate_class,
                              const double* d_post_rate, const double* d_norm, const cmx_pair_filters* filters,
                              cmx_pair_row* d_rows, size_t capacity, uint64_t* d_count, void* stream);

/* The same pair loop for the rows [row_begin, row_end) of the upper triangle only, from the substitution vectors
 * (branch-major device counts as cmx_map_sites_dev writes them) and the merged null, a block of rows at a time: no
 * dense N x N matrix exists anywhere (scratch = one row block, <= 256 MiB).  This is what a multi-GPU job calls, each
 * rank with its own row range (rows come out in the reference's (i, j) order, so the ranks' outputs concatenate to the
 * single-GPU output), and the default single-GPU path for large N.  d_null_stat == NULL: no p-values (NaN, Nsim 0). */
cmx_status cmx_intra_rows_range_dev(cmx_ctx* ctx, int kind, const double* params, const double* d_counts, size_t n, size_t ldc,
                                    const int32_t* d_rate_class, const double* d_post_rate, const double* d_norm,
                                    const double* d_null_stat, const double* d_null_nmin, size_t nnull, int nclasses,
                                    const cmx_pair_filters* filters, size_t row_begin, size_t row_end, cmx_pair_row* d_rows,
                                    size_t capacity, uint64_t* d_count, void* stream);
/* The UNFILTERED pair loop as 16 bytes per pair (round 4): statistic, number of null values of the pair's norm class that
 * are smaller (CoETools.cpp:712-717) and the size of that class, for the pairs (i, j > i) of the rows [row_begin, row_end)
 * at their position in the reference's (i, j) order -- which, without filters, is arithmetic, so no counting pass runs.
 * below = 0xffffffff, nsim = 0: PValue NA (smaller norm outside the Domain, CoETools.cpp:718-720) or no null given.
 * Everything else a statistics.txt row carries is a function of per-site arrays: cmx_expand_compact_rows (host side, no
 * GPU) rebuilds the cmx_pair_row array bit for bit.  A job without pair filters brings a third of the bytes home. */
typedef struct cmx_pair_compact {
  double stat;
  uint32_t below, nsim;
} cmx_pair_compact;
cmx_status cmx_intra_compact_range_dev(cmx_ctx* ctx, int kind, const double* params, const double* d_counts, size_t n, size_t ldc,
                                       const double* d_norm, const double* d_null_stat, const double* d_null_nmin, size_t nnull,
                                       int nclasses, size_t row_begin, size_t row_end, cmx_pair_compact* d_out, size_t capacity,
                                       void* stream);
/* Optional first half of cmx_intra_compact_range_dev, for a caller that maps the observed alignment beside the null (the
 * reference runs CoETools::computeIntraStats' null, CoETools.cpp:836-872, before its pair loop, :672-724; the statistics
 * of the observed pairs do not depend on it): the operand preparation and the Gram blocks of the rows [row_begin, row_end)
 * are enqueued on `stream` NOW and kept (rows x n doubles) for the next cmx_intra_compact_range_dev with the same kind,
 * d_counts, n, ldc and row range, which then only runs its record pass; any other call to it, and a mapping call that
 * writes to d_counts, discards them.  The vectors behind d_counts must not change in between by other means, and that next call's stream must be ordered behind this one's.  Kept only
 * when it fits 2 GiB and for statistics without parameters; otherwise nothing happens here and the later call does all
 * the work.  Returns CMX_OK either way. */
cmx_status cmx_intra_gram_prefetch_dev(cmx_ctx* ctx, int kind, const double* d_counts, size_t n, size_t ldc, size_t row_begin,
                                       size_t row_end, void* stream);
/* host only: rows[k] for the pairs of the rows [row_begin, row_end) in (i, j) order from their compact records and the
 * per-site arrays; nthreads <= 1: the calling thread alone.  npairs must equal the number of pairs of the row range. */
cmx_status cmx_expand_compact_rows(size_t n, size_t row_begin, size_t row_end, const int32_t* rate_class, const double* post_rate,
                                   const double* norm, const cmx_pair_compact* compact, size_t npairs, cmx_pair_row* rows, int nthreads);
/* AnalysisTools::compute{ScalarProduct,Cosinus,Correlation,Covariance}Matrix (CoMap/AnalysisTools.h:93-190,
 * AnalysisTools.cpp:102-339) for plain vectors: v1 [n1][dim], v2 [n2][dim] (NULL: the one-set form) in host memory, any
 * dimension (nothing here depends on the context's tree or model: a context created without them serves too).
 * kind: CMX_STAT_SCALAR_PRODUCT / COSINUS / CORRELATION / COVARIANCE.  out [n1][n2] (one-set: [n1][n1], symmetric, the
 * diagonal as the reference sets it: scalar(v, v), 1, 1, var(v)).  independent != 0 (two-set form only, n1 == n2, else
 * CMX_ERR_INVALID = the reference's DimensionException): only out[i][i] is computed, the rest is 0. */
cmx_status cmx_vector_matrix(cmx_ctx* ctx, int kind, size_t dim, const double* v1, size_t n1, const double* v2, size_t n2,
                             int independent, double* out);
/* host pointers: counts [N][B][K] -> statistic -> (optional) p-values from a null -> compacted rows; only the rows
 * cross PCIe on the way back.  null_stat == NULL: no p-values (pvalue NaN, Nsim 0 in every row). */
cmx_status cmx_intra_rows(cmx_ctx* ctx, int kind, const double* params, const double* counts, size_t n,
                          const int32_t* rate_class, const double* post_rate, const double* norm,
                          const double* null_stat, const double* null_nmin, size_t nnull, int nclasses,
                          const cmx_pair_filters* filters, cmx_pair_row* rows, size_t capacity, uint64_t* count);

/* ---- the rows of the inter-gene statistics file: CoETools::computeInterStats' pair loop (CoMap/CoETools.cpp:786-828) for
 * the rows of data set 1 against data set 2 -- statistic on the matrix cores a block of rows at a time, filters, rows
 * compacted in the reference's (i, j) order; no N1 x N2 matrix leaves the device (or exists beyond a <= 256 MiB block).
 * Both data sets: branch-major counts of the same branches, rate class / posterior rate / norm per site.  Rows carry
 * pvalue NaN, nsim 0 (this path has no p-values in the reference either). */
typedef struct cmx_inter_filters {
  int32_t min_rate_class1, min_rate_class2; /* statistic.min_rate_class, .min_rate_class2 (CoETools.cpp:755-756) */
  int32_t max_rate_class_diff;              /* < 0: off */
  int32_t independent_comparisons;          /* independant_comparisons = yes: only the pairs (i, i), n1 == n2 (:744-748) */
  double min_rate1, min_rate2;
  double max_rate_diff;                     /* < 0: off */
  double min_statistic;
  int32_t reference_norm_quirk;             /* 0: Nmin = min(norm1[i], norm2[j]).  1: the reference's own column,
                                               min(norms1[i], norms2[i]) -- CoETools.cpp:803 reads norms2[i] */
  int32_t reserved;
} cmx_inter_filters;
cmx_status cmx_inter_rows_dev(cmx_ctx* ctx, int kind, const double* params, const double* d_counts1, size_t n1, size_t ld1,
                              const int32_t* d_rate_class1, const double* d_post_rate1, const double* d_norm1,
                              const double* d_counts2, size_t n2, size_t ld2, const int32_t* d_rate_class2,
                              const double* d_post_rate2, const double* d_norm2, const cmx_inter_filters* filters,
                              cmx_pair_row* d_rows, size_t capacity, uint64_t* d_count, void* stream);
/* host pointers, counts [N][B][K]; only the rows cross PCIe on the way back */
cmx_status cmx_inter_rows(cmx_ctx* ctx, int kind, const double* params, const double* counts1, size_t n1,
                          const int32_t* rate_class1, const double* post_rate1, const double* norm1, const double* counts2,
                          size_t n2, const int32_t* rate_class2, const double* post_rate2, const double* norm2,
                          const cmx_inter_filters* filters, cmx_pair_row* rows, size_t capacity, uint64_t* count);

/* ---- Mica: mutual information between alignment columns over taxa (Mica.cpp:93-95, 349-361, 646-660).
 * aln2 == NULL: intra (filled for j > i, NaN elsewhere).  Outputs dense [n1][n2] (mi, hjoint) and per-column entropies;
 * nalpha = alphabet size.  Alignment codes >= nalpha index `masks` (bit a set = compatible with state a; masks == NULL:
 * every such code is an unknown): SiteTools::*(.., resolveUnknowns = true) spreads such a symbol evenly over its
 * compatible states.  Unknowns (gap, X, N: all states) cost nothing extra; a column with a partial ambiguity code sends
 * its pairs through a slower kernel. */
cmx_status cmx_mi_columns(cmx_ctx* ctx, int nalpha, int ntaxa, const uint32_t* masks, size_t nmasks,
                          const uint8_t* aln1, size_t n1, const uint8_t* aln2, size_t n2, double* mi,
                          double* hjoint, double* h1, double* h2);
cmx_status cmx_mi_columns_dev(cmx_ctx* ctx, int nalpha, int ntaxa, const uint32_t* d_masks, const uint8_t* d_aln1,
                              size_t n1, size_t ld1, const uint8_t* d_aln2, size_t n2, size_t ld2, double* d_mi,
                              double* d_hjoint, size_t ldo, double* d_h1, double* d_h2, void* stream);
/* MI and joint entropy of listed column pairs (idx1[p] of aln1, idx2[p] of aln2; aln2 == NULL: both from aln1): the
 * statistic of Mica's null distributions -- non-parametric bootstrap over random site pairs (CoMap/Mica.cpp:399-468)
 * and parametric bootstrap over pairs (j, j) of two simulated alignments (:469-548, with cmx_simulate). */
cmx_status cmx_mi_pairs(cmx_ctx* ctx, int nalpha, int ntaxa, const uint32_t* masks, size_t nmasks, const uint8_t* aln1,
                        size_t n1, const uint8_t* aln2, size_t n2, const int64_t* idx1, const int64_t* idx2,
                        size_t npairs, double* mi, double* hjoint);
/* device pointers throughout (alignments [T][ld], indices, outputs); d_masks: 256 masks or NULL; indices are not checked */
cmx_status cmx_mi_pairs_dev(cmx_ctx* ctx, int nalpha, int ntaxa, const uint32_t* d_masks, const uint8_t* d_aln1, size_t n1, size_t ld1,
                            const uint8_t* d_aln2, size_t n2, size_t ld2, const int64_t* d_idx1, const int64_t* d_idx2, size_t npairs,
                            double* d_mi, double* d_hjoint, void* stream);

/* Mica's bootstrap nulls.
 * cmx_mica_bootstrap_indices (host-side only, no GPU): the site indices of null.method = nonparametric-bootstrap
 * (SiteContainerTools::sampleSites, CoMap/Mica.cpp:426-430) from the engine's counter RNG, so that every binding draws the
 * same pairs: idx_h[r * nrep_ram + j] = floor(u(seed, g = (r * 2 + h) * nrep_ram + j) * nsites).  Score them with cmx_mi_pairs.
 * cmx_mica_parametric_null: null.method = parametric-bootstrap (Mica.cpp:469-548) -- per replicate two alignments of
 * nrep_ram sites simulated under the context's model (gamma_alpha > 0: simulations.continuous = yes with that Gamma shape
 * and p_invariant), column j against column j: mi, hjoint [nrep_cpu * nrep_ram]; nmin (may be NULL) = the smaller of the
 * two simulated sites' norms (`use_model`).  Simulation, MI and mapping stay on the device. */
cmx_status cmx_mica_bootstrap_indices(uint64_t seed, size_t nsites, size_t nrep_cpu, size_t nrep_ram, int64_t* idx1, int64_t* idx2);
cmx_status cmx_mica_parametric_null(cmx_ctx* ctx, int nalpha, uint64_t seed, size_t nrep_cpu, size_t nrep_ram, double gamma_alpha,
                                    double p_invariant, double* mi, double* hjoint, double* nmin);

/* ---- Mica, after the all-pairs matrix (intra, n columns, upper triangle j > i is what is read):
 * cmx_mica_average_mi: averageMI[i] = sum_{j != i} MI(i,j) / (n-1) and fullAverageMI = mean(averageMI)
 * (CoMap/Mica.cpp:346-363); the APC and RCW columns of the output are averageMI[i]*averageMI[j]/fullAverageMI and
 * averageMI[i]*averageMI[j]/2 (Mica.cpp:656-657).
 * cmx_mica_zscore_null: null.method = z-score (Mica.cpp:549-607): every pair of the data set is one draw; `which`
 * selects null.method_zscore.stat (MI, MIp = MI - APC, MIc = MI / RCW); key = entropy per column, or the norms when a
 * model is used (Mica.cpp:573-601).  Outputs [n(n-1)/2] in the reference's (i, j) order: the statistic and
 * min(key[i], key[j]).  p-values (Mica.cpp:671-683) then come from cmx_intra_pvalues(_dev) with stat = the MI
 * matrix and norms = key: same Domain(0, max(key), null.nb_rate_classes) binning, same count, same "NA". */
enum { CMX_MICA_MI = 0, CMX_MICA_MIP = 1, CMX_MICA_MIC = 2 };
cmx_status cmx_mica_average_mi_dev(cmx_ctx* ctx, const double* d_mi, size_t n, size_t ldo, double* d_average,
                                   double* d_full_average, void* stream);
cmx_status cmx_mica_average_mi(cmx_ctx* ctx, const double* mi, size_t n, double* average, double* full_average);
cmx_status cmx_mica_zscore_null_dev(cmx_ctx* ctx, int which, const double* d_mi, size_t n, size_t ldo,
                                    const double* d_average, const double* d_full_average, const double* d_key,
                                    double* d_null_stat, double* d_null_key, void* stream);
cmx_status cmx_mica_zscore_null(cmx_ctx* ctx, int which, const double* mi, size_t n, const double* key,
                                double* null_stat, double* null_key);

/* ---- groups of sites.  Statistic::getValueForGroup (CoMap/Statistics.h:121-133: the smallest pairwise value; :267-294:
 * Compensation's closed form) of ngroups groups; group g = sites[offsets[g] .. offsets[g+1]) (indices into the
 * mapping).  This is CandidateGroup::computeStatisticValue (CoMap/CoETools.h:106-117). */
cmx_status cmx_group_stats_dev(cmx_ctx* ctx, int kind, const double* params, const double* d_counts, size_t n, size_t ldc,
                               const int64_t* d_offsets, const int32_t* d_sites, size_t ngroups, double* d_out,
                               void* stream);
cmx_status cmx_group_stats(cmx_ctx* ctx, int kind, const double* params, const double* counts, size_t n,
                           const int64_t* offsets, const int32_t* sites, size_t ngroups, double* out);
/* Candidate-group test: CoETools::computePValuesForCandidateGroups + CandidateGroupSet::analyseSimulations
 * (CoMap/CoETools.cpp:1042-1087, :950-1038, cursor :900-947).  Batches of rep_ram sites are simulated and mapped on the
 * device; their norms drive the reference's greedy assembly of pseudo-groups (a simulated site is given to the next
 * candidate site, in cursor order, whose norm window [norm_lo, norm_hi] contains its norm -- windows from
 * CandidateGroup::computeNormRanges, CoETools.h:118-128); the statistic of every completed pseudo-group is evaluated on
 * the device and counted in n1 when >= observed[g] (n2 counts the pseudo-groups).  Stops when every analysable group
 * has min_sim pseudo-groups, after max_trials batches that completed nothing (candidates.nb_max_trials), or after
 * max_batches batches (0 = no limit; a safety valve the reference does not have).  p-value of group g =
 * (n1[g] + 1) / (n2[g] + 1) (CoETools.h:235-238).  Batch t simulates sites t*rep_ram .. of the counter RNG. */
cmx_status cmx_candidate_groups(cmx_ctx* ctx, int kind, const double* params, size_t ngroups, const int64_t* offsets,
                                const double* norm_lo, const double* norm_hi, const uint8_t* analysable,
                                const double* observed, uint32_t min_sim, size_t rep_ram, uint32_t max_trials,
                                uint64_t max_batches, uint64_t seed, uint32_t* n1, uint32_t* n2, uint32_t* trials,
                                uint64_t* batches);
/* host-side only (no GPU needed): the cursor of cmx_candidate_groups run over caller-supplied norms
 * [nbatches][rep_ram]; lists the pseudo-groups it assembles (candidate group, batch, stand-in sites within the batch;
 * pg_offsets has cap_groups + 1 entries).  *npg receives their number even beyond the capacities. */
cmx_status cmx_debug_candidate_cursor(size_t ngroups, const int64_t* offsets, const double* norm_lo, const double* norm_hi,
                                      const uint8_t* analysable, uint32_t min_sim, const double* norms, size_t rep_ram,
                                      size_t nbatches, uint32_t max_trials, uint32_t* n2, uint32_t* trials,
                                      uint64_t* batches_used, int32_t* pg_group, int32_t* pg_batch, int64_t* pg_offsets,
                                      int32_t* pg_sites, size_t cap_groups, size_t cap_sites, size_t* npg);

/* Mica's permutation test, null.method = permutations: miTest (CoMap/Mica.cpp:93-118) for the pairs
 * [pair_begin, pair_end) of the row-major (i < j) order: columns are shuffled until 5 shuffles reach the observed MI or
 * max_perm (null.max_number_of_permutations) were done; pvalue = (count + 1) / (nperm + 1); pairs with a constant
 * column (SiteTools::isConstant(site, ignoreUnknown = true): at most one distinct symbol besides unknowns) get pvalue 1,
 * nperm 0.  "MI of the shuffle >= MI" is decided on sum m ln m of the joint table in fixed point (exact, order-free;
 * DESIGN.md 4.4).  Shuffles come from the counter RNG (seed, pair, permutation, position), so results do not depend on
 * how pairs are sharded.  ntaxa <= 2047.
 * Gaps, unknowns and ambiguity codes count as SiteTools::mutualInformation(.., resolveUnknowns = true) counts them: a
 * symbol with k compatible states adds 1/k to each.  `masks` is a HOST table indexed by alignment code (bit a =
 * compatible with state a), as for cmx_mi_columns; codes >= nalpha without an entry (and masks == NULL) are unknowns.
 * Partial ambiguity codes must be < 31.  Pairs of fully resolved columns take the same path, and give the same
 * results, with or without a mask table.
 * A column WITHOUT ANY resolved symbol (gaps / unknowns only) is treated like a constant column: pvalue 1, nperm 0.
 * (Whether Bio++'s SiteTools::isConstant(site, true) says the same of such a site could not be checked: bpp-seq is not
 * in the reference tree.  Its MI with any column is 0 either way.)
 * The call blocks the host once (it has to learn whether any pair carries unknowns); the fixed-point table of the
 * unknowns' path is built once per (alphabet, ambiguity codes, ntaxa) and kept by the context. */
cmx_status cmx_mica_permutation_test_masks_dev(cmx_ctx* ctx, int nalpha, int ntaxa, const uint32_t* masks, size_t nmasks,
                                               const uint8_t* d_aln, size_t n, size_t ld, uint32_t max_perm, uint64_t seed,
                                               size_t pair_begin, size_t pair_end, double* d_pvalue, int32_t* d_nperm,
                                               void* stream);
cmx_status cmx_mica_permutation_test_masks(cmx_ctx* ctx, int nalpha, int ntaxa, const uint32_t* masks, size_t nmasks,
                                           const uint8_t* aln, size_t n, uint32_t max_perm, uint64_t seed,
                                           double* pvalue /*[n(n-1)/2]*/, int32_t* nperm);
/* the same with masks == NULL */
cmx_status cmx_mica_permutation_test_dev(cmx_ctx* ctx, int nalpha, int ntaxa, const uint8_t* d_aln, size_t n, size_t ld,
                                         uint32_t max_perm, uint64_t seed, size_t pair_begin, size_t pair_end,
                                         double* d_pvalue, int32_t* d_nperm, void* stream);
cmx_status cmx_mica_permutation_test(cmx_ctx* ctx, int nalpha, int ntaxa, const uint8_t* aln, size_t n, uint32_t max_perm,
                                     uint64_t seed, double* pvalue /*[n(n-1)/2]*/, int32_t* nperm);

/* ---- clustering analysis (CoMap/CoMap.cpp:395-560; null: ClusterTools::computeGlobalDistanceDistribution,
 * CoMap/ClusterTools.cpp:200-294).  Distances of CoMap.cpp:402-428: 1 - correlation (StatisticBasedDistance(cor, 1.),
 * Distance.h:321-336), 1 - compensation (CompensationDistance, Distance.h:376-385), Euclidian (Distance.h:161-181).
 * Agglomeration = bpp::HierarchicalClustering as constructed at CoMap.cpp:460-472.
 * A clustering tree over n sites is returned as its n-1 joins in the order they were made: leaves are 0..n-1, join m
 * creates node n+m with sons merge[m][0], merge[m][1] (the reference's son order); dmax[m] is the distance at which
 * the two were joined (= 2 * node height = the "Dmax" column), size[m] the number of sites below, stat[m] / nmin[m]
 * the "Stat" / "Nmin" node properties (Distance.h:113-126, :353-366, :393-413; ClusterTools.cpp:302-320).
 * ClusterTools::getGroups (ClusterTools.cpp:61-113) is a post-order walk of these joins (adapter / comap_amd.formats).
 * Ties between equal distances go to the first pair in index order (CoMap/Cluster.cpp:55-79); NaN distances are
 * treated as +inf.  n is limited to CMX_CLUSTER_MAX_SITES (the per-matrix state lives in LDS). */
enum { CMX_DIST_CORRELATION = 0, CMX_DIST_COMPENSATION = 1, CMX_DIST_EUCLIDIAN = 2 };
enum { CMX_LINK_COMPLETE = 0, CMX_LINK_SINGLE = 1, CMX_LINK_AVERAGE = 2 };
#define CMX_CLUSTER_MAX_SITES 5000
/* agglomeration only: `batch` independent symmetric n x n matrices (row stride ld, matrix stride n*ld); the device
 * variant overwrites them.  Needs no model: works on a context created with model == NULL. */
cmx_status cmx_hclust_dev(cmx_ctx* ctx, int linkage, double* d_dist, size_t n, size_t ld, size_t batch, int32_t* d_merge,
                          double* d_dmax, int32_t* d_size, void* stream);
cmx_status cmx_hclust(cmx_ctx* ctx, int linkage, const double* dist, size_t n, size_t batch, int32_t* merge, double* dmax,
                      int32_t* size);
/* observed data: substitution vectors -> distance matrix (optionally returned, [n][n], zero diagonal: the matrix of
 * clustering.output.matrix.file) -> clustering tree with its group properties.  counts as in cmx_pair_stats(_dev);
 * d_norm from cmx_map_sites_dev. */
cmx_status cmx_cluster_sites_dev(cmx_ctx* ctx, int dist_kind, int linkage, const double* d_counts, size_t n, size_t ldc,
                                 const double* d_norm, double* d_dist_out, int32_t* d_merge, double* d_dmax,
                                 int32_t* d_size, double* d_stat, double* d_nmin, void* stream);
cmx_status cmx_cluster_sites(cmx_ctx* ctx, int dist_kind, int linkage, const double* counts, size_t n, double* dist_out,
                             int32_t* merge, double* dmax, int32_t* size, double* stat, double* nmin);
/* the null: replicates [rep_begin, rep_end), each one simulates nsites sites (global site indices rep*nsites ..),
 * maps them, builds their distance matrix and clusters it, all on the device; replicates are processed in batches
 * of independent matrices, one workgroup per matrix.  Host outputs, [(rep_end-rep_begin)][nsites-1] (merge: [..][2]). */
cmx_status cmx_cluster_null(cmx_ctx* ctx, int dist_kind, int linkage, uint64_t seed, size_t rep_begin, size_t rep_end,
                            size_t nsites, int32_t* merge, double* dmax, int32_t* size, double* stat, double* nmin);


#ifdef __cplusplus
}
#endif
#endif /* COMAP_MI355X_H */
