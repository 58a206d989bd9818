/* TEST INFRASTRUCTURE ONLY.  CPU (fp64, single thread) restatement of the reference hot path:
 *   substitution mapping (DR likelihood + expected per-branch counts), pair statistics,
 *   parametric-bootstrap null, p-values, Mica column MI.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (comap_amd/, libcomap_mi355x.so) never links, imports or calls it.
 *
 * Where the algorithm lives: the arithmetic below is Bio++'s (bpp-core / bpp-seq / bpp-phyl >= 3.0.0,
 * /root/reference/CMakeLists.txt:107), a third-party dependency that is NOT vendored under
 * /root/reference and is absent from this image, so the reference cannot be built here.  Its published
 * algorithms are restated and anchored on the reference's own call sites:
 *   CoMap/CoETools.cpp:124,209,397 (likelihood + computeSubstitutionVectors), CoMap/AnalysisTools.cpp:343-350
 *   (norms), :564-658 (null loop), CoMap/Statistics.h:164-329 (scorers), CoMap/Domain.cpp:46-59,113-122,
 *   CoMap/CoETools.cpp:672-724 (pair loop + p-value rule), CoMap/Mica.cpp:93-118,346-361.
 * PINNED: mapping/likelihood/rates against the reference's committed fixtures
 *   examples/Proteins/Benchmark/CoMap/Myo_{unif,naive,unif_grantham,naive_grantham}.vec and Myo.infos
 *   (tests/golden/myoglobin.npz, tests/test_golden_myoglobin.py).
 * PARITY UNPINNED (no reference outputs exist): pair statistics, null distribution, p-values, simulator,
 *   Mica MI -- these follow the in-tree formulas cited at each function.
 *
 * Conventions: tree nodes in post-order, root last, parent[root] = -1; branch b == node id of the branch's
 * lower node (row order of the reference .vec files).  Alignment codes: aln[t*N + i] (taxon-major);
 * code < S is a state, otherwise an index into masks[] (bit z set <=> state z compatible).  Alphabets of more than 32
 * states (codon models, CoETools.cpp:95-100) have no mask table: every code >= S is an unknown, compatible with all.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* leaf initialisation of the DR likelihood: 1 where state x is compatible with the observed code */
static inline double leaf_compatible(const uint32_t* masks, int S, unsigned code, int x) {
  if (code < (unsigned)S) return (unsigned)x == code ? 1.0 : 0.0;
  if (S > 32) return 1.0;
  return (double)((masks[code] >> x) & 1u);
}

#include <string.h>

#define IDX3(a, b, c, nb, nc) (((size_t)(a) * (nb) + (b)) * (nc) + (c))

/* ------------------------------------------------------------------ small dense helpers */
static void matmul(int n, const double* A, const double* B, double* C) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int k = 0; k < n; k++) s += A[i * n + k] * B[k * n + j];
      C[i * n + j] = s;
    }
}

/* cyclic Jacobi on a symmetric matrix; A is destroyed, U gets the eigenvectors in columns */
static void jacobi_sym(int n, double* A, double* U, double* lam) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) U[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0;
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
    if (off < 1e-60) break;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) {
        double apq = A[p * n + q];
        if (fabs(apq) < 1e-300) continue;
        double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
        double c = 1 / sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < n; k++) {
          double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {
          double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; k++) {
          double ukp = U[k * n + p], ukq = U[k * n + q];
          U[k * n + p] = c * ukp - s * ukq;
          U[k * n + q] = s * ukp + c * ukq;
        }
      }
  }
  for (int i = 0; i < n; i++) lam[i] = A[i * n + i];
}

/* Q = V diag(lam) Vinv for a reversible generator (pi_x Q_xy = pi_y Q_yx), through the symmetrised form.
 * Bio++ diagonalises Q in AbstractSubstitutionModel::updateMatrices; P(t) = V e^{lam t} Vinv (A.2). */
int orc_eigen_reversible(int S, const double* Q, const double* pi, double* lam, double* V, double* Vinv) {
  double* A = (double*)malloc(sizeof(double) * S * S);
  double* U = (double*)malloc(sizeof(double) * S * S);
  for (int i = 0; i < S; i++)
    for (int j = 0; j < S; j++) A[i * S + j] = sqrt(pi[i]) * Q[i * S + j] / sqrt(pi[j]);
  for (int i = 0; i < S; i++)
    for (int j = i + 1; j < S; j++) A[i * S + j] = A[j * S + i] = 0.5 * (A[i * S + j] + A[j * S + i]);
  jacobi_sym(S, A, U, lam);
  for (int i = 0; i < S; i++)
    for (int j = 0; j < S; j++) {
      V[i * S + j] = U[i * S + j] / sqrt(pi[i]);
      Vinv[j * S + i] = U[i * S + j] * sqrt(pi[i]);
    }
  free(A);
  free(U);
  return 0;
}

void orc_transition(int S, const double* lam, const double* V, const double* Vinv, double t, double* P) {
  for (int i = 0; i < S; i++)
    for (int j = 0; j < S; j++) {
      double s = 0;
      for (int k = 0; k < S; k++) s += V[i * S + k] * exp(lam[k] * t) * Vinv[k * S + j];
      P[i * S + j] = s;
    }
}

/* Uniformization (reference default nijt, doc/comap.texi:155):
 * J(t) = sum_n Pois(n+1; mu t)/mu * sum_{l=0..n} R^l B R^{n-l},  R = I + Q/mu. */
void orc_count_unif(int S, const double* Q, const double* Bm, double t, double* J) {
  int n2 = S * S;
  double mu = 0;
  for (int i = 0; i < S; i++)
    if (fabs(Q[i * S + i]) > mu) mu = fabs(Q[i * S + i]);
  double* R = (double*)malloc(sizeof(double) * n2 * 5);
  double *Rn = R + n2, *s = R + 2 * n2, *tmp = R + 3 * n2, *tmp2 = R + 4 * n2;
  for (int i = 0; i < n2; i++) {
    R[i] = Q[i] / mu + ((i / S) == (i % S));
    Rn[i] = ((i / S) == (i % S));
    s[i] = Bm[i];
    J[i] = 0;
  }
  double mt = mu * t;
  int nmax = (int)ceil(4 + 6 * sqrt(mt) + mt) + 10;
  double logp = -mt + log(mt); /* Pois(1) */
  for (int n = 0; n <= nmax; n++) {
    double w = exp(logp) / mu;
    for (int i = 0; i < n2; i++) J[i] += w * s[i];
    matmul(S, Rn, R, tmp); /* R^{n+1} */
    memcpy(Rn, tmp, sizeof(double) * n2);
    matmul(S, R, s, tmp);
    matmul(S, Bm, Rn, tmp2);
    for (int i = 0; i < n2; i++) s[i] = tmp[i] + tmp2[i];
    logp += log(mt) - log((double)(n + 2));
  }
  free(R);
}

/* Decomposition: J = V [ (Vinv B V) o Phi ] Vinv, Phi_ij = (e^{li t}-e^{lj t})/(li-lj) or t e^{li t}.
 * The difference quotient is evaluated through expm1 (the plain difference loses the O(t^2) diagonal on
 * 1e-6 branches -- the reference's own Myo_decomp.vec shows that loss; Myo_unif.vec is the pinned one). */
void orc_count_decomp(int S, const double* lam, const double* V, const double* Vinv, const double* Bm, double t,
                      double* J) {
  int n2 = S * S;
  double* W = (double*)malloc(sizeof(double) * n2 * 2);
  double* X = W + n2;
  matmul(S, Vinv, Bm, X);
  matmul(S, X, V, W);
  for (int i = 0; i < S; i++)
    for (int j = 0; j < S; j++) {
      double d = (lam[i] - lam[j]) * t, phi;
      if (fabs(d) < 1e-14) phi = t * exp(lam[i] * t);
      else phi = t * exp(lam[j] * t) * expm1(d) / d;
      W[i * S + j] *= phi;
    }
  matmul(S, V, W, X);
  matmul(S, X, Vinv, J);
  free(W);
}

/* ------------------------------------------------------------------ mapping */
static void child_lists(int nn, const int* parent, int** first, int** next) {
  int* f = (int*)malloc(sizeof(int) * nn);
  int* nx = (int*)malloc(sizeof(int) * nn);
  int* last = (int*)malloc(sizeof(int) * nn);
  for (int i = 0; i < nn; i++) f[i] = nx[i] = last[i] = -1;
  for (int i = 0; i < nn; i++) {
    int p = parent[i];
    if (p < 0) continue;
    if (f[p] < 0) f[p] = i; else nx[last[p]] = i;
    last[p] = i;
  }
  free(last);
  *first = f;
  *next = nx;
}

/* counts N = J / P with Bio++'s guards (A.4): non-finite -> 0, unweighted negatives -> 0; returns P o N */
static void joint_count_matrix(int S, const double* J, const double* P, int nonneg, double* PN) {
  for (int i = 0; i < S * S; i++) {
    double nxy = J[i] / P[i];
    if (isnan(nxy) || isinf(nxy)) nxy = 0;
    if (nonneg && nxy < 0) nxy = 0;
    PN[i] = P[i] * nxy;
  }
}

/* DRHomogeneousTreeLikelihood::initialize + getLogLikelihoodPerSite / getPosteriorRatePerSite /
 * getRateClassWithMaxPostProbPerSite (A.2, A.5) followed by
 * LegacySubstitutionMappingTools::computeSubstitutionVectors(average, joint) (A.3) and computeNormForSite (A.6).
 * method: 0 uniformization, 1 decomposition, 2 naive (N(x,y) = W(x,y)[x != y]).
 * counts is site-major [N][B][K] == mapping[i][b][k] (CoMap/Statistics.h:154-160). */
int orc_map_sites(int nn, const int* parent, const double* blen, int T, const int* leaf_of_taxon, long N,
                  const uint8_t* aln, const uint32_t* masks, int S, int C, int K, const double* Q, const double* pi,
                  const double* rates, const double* probs, const double* Bk, int method, int nonneg,
                  const double* naiveW, double* counts, double* logL, double* post_rate, int* rate_class,
                  double* norm) {
  int B = nn - 1, root = nn - 1, S2 = S * S;
  int *first, *next;
  child_lists(nn, parent, &first, &next);
  int* taxon_of = (int*)malloc(sizeof(int) * nn);
  for (int i = 0; i < nn; i++) taxon_of[i] = -1;
  for (int t = 0; t < T; t++) taxon_of[leaf_of_taxon[t]] = t;
  double* lam = (double*)malloc(sizeof(double) * (S + 2 * S2));
  double *V = lam + S, *Vinv = V + S2;
  orc_eigen_reversible(S, Q, pi, lam, V, Vinv);
  /* per branch x class: P and (P o N^k) */
  double* P = (double*)malloc(sizeof(double) * (size_t)B * C * S2);
  double* PN = (double*)malloc(sizeof(double) * (size_t)B * C * K * S2);
  double* Jtmp = (double*)malloc(sizeof(double) * S2);
  for (int b = 0; b < B; b++)
    for (int c = 0; c < C; c++) {
      double t = blen[b] * rates[c];
      double* Pbc = P + ((size_t)b * C + c) * S2;
      orc_transition(S, lam, V, Vinv, t, Pbc);
      for (int k = 0; k < K; k++) {
        double* PNk = PN + (((size_t)b * C + c) * K + k) * S2;
        if (method == 2) {
          for (int i = 0; i < S2; i++) PNk[i] = ((i / S) == (i % S)) ? 0.0 : Pbc[i] * (naiveW ? naiveW[i] : 1.0);
        } else {
          if (method == 0) orc_count_unif(S, Q, Bk + (size_t)k * S2, t, Jtmp);
          else orc_count_decomp(S, lam, V, Vinv, Bk + (size_t)k * S2, t, Jtmp);
          joint_count_matrix(S, Jtmp, Pbc, nonneg, PNk);
        }
      }
    }
  /* per site work arrays: D[node][c][x], M[node][c][x] (= P_node D_node), Up[node][c][x] */
  size_t vsz = (size_t)nn * C * S;
  double* D = (double*)malloc(sizeof(double) * vsz * 3);
  double *M = D + vsz, *Up = M + vsz;
  double* U = (double*)malloc(sizeof(double) * S);
  double* Lc = (double*)malloc(sizeof(double) * C);
  for (long i = 0; i < N; i++) {
    for (int n = 0; n < nn; n++) {
      for (int c = 0; c < C; c++) {
        double* Dn = D + ((size_t)n * C + c) * S;
        if (first[n] < 0) {
          uint8_t code = aln[(size_t)taxon_of[n] * N + i];
          for (int x = 0; x < S; x++) Dn[x] = leaf_compatible(masks, S, code, x);
        } else {
          for (int x = 0; x < S; x++) Dn[x] = 1.0;
          for (int e = first[n]; e >= 0; e = next[e]) {
            const double* Me = M + ((size_t)e * C + c) * S;
            for (int x = 0; x < S; x++) Dn[x] *= Me[x];
          }
        }
        if (n != root) {
          const double* Pn = P + ((size_t)n * C + c) * S2;
          double* Mn = M + ((size_t)n * C + c) * S;
          for (int x = 0; x < S; x++) {
            double s = 0;
            for (int z = 0; z < S; z++) s += Pn[x * S + z] * Dn[z];
            Mn[x] = s;
          }
        }
      }
    }
    double L = 0, pr = 0, best = -1;
    int bestc = 0;
    for (int c = 0; c < C; c++) {
      const double* Dr = D + ((size_t)root * C + c) * S;
      double s = 0;
      for (int x = 0; x < S; x++) s += pi[x] * Dr[x];
      Lc[c] = s;
      L += probs[c] * s;
      pr += rates[c] * probs[c] * s;
      if (probs[c] * s > best) { best = probs[c] * s; bestc = c; }
    }
    logL[i] = log(L);
    post_rate[i] = pr / L;
    rate_class[i] = bestc;
    /* outside pass, root first */
    for (int c = 0; c < C; c++)
      for (int x = 0; x < S; x++) Up[((size_t)root * C + c) * S + x] = pi[x];
    double nrm = 0;
    double* ci = counts + (size_t)i * B * K;
    for (int b = 0; b < B * K; b++) ci[b] = 0;
    for (int f = nn - 1; f >= 0; f--) {
      if (first[f] < 0) continue;
      for (int n = first[f]; n >= 0; n = next[n]) {
        for (int c = 0; c < C; c++) {
          const double* Upf = Up + ((size_t)f * C + c) * S;
          for (int x = 0; x < S; x++) U[x] = Upf[x];
          for (int m = first[f]; m >= 0; m = next[m])
            if (m != n) {
              const double* Mm = M + ((size_t)m * C + c) * S;
              for (int x = 0; x < S; x++) U[x] *= Mm[x];
            }
          const double* Dn = D + ((size_t)n * C + c) * S;
          for (int k = 0; k < K; k++) {
            const double* PNk = PN + (((size_t)n * C + c) * K + k) * S2;
            double tot = 0;
            for (int x = 0; x < S; x++) {
              double s = 0;
              for (int y = 0; y < S; y++) s += PNk[x * S + y] * Dn[y];
              tot += U[x] * s;
            }
            ci[(size_t)n * K + k] += probs[c] * tot;
          }
          if (first[n] >= 0) {
            const double* Pn = P + ((size_t)n * C + c) * S2;
            double* Upn = Up + ((size_t)n * C + c) * S;
            for (int z = 0; z < S; z++) {
              double s = 0;
              for (int x = 0; x < S; x++) s += Pn[x * S + z] * U[x];
              Upn[z] = s;
            }
          }
        }
      }
    }
    for (int b = 0; b < B; b++) {
      double tot = 0;
      for (int k = 0; k < K; k++) {
        ci[(size_t)b * K + k] /= L;
        tot += ci[(size_t)b * K + k];
      }
      nrm += tot * tot;
    }
    norm[i] = sqrt(nrm);
  }
  free(D); free(U); free(Lc); free(P); free(PN); free(Jtmp); free(lam); free(taxon_of); free(first); free(next);
  return 0;
}

/* LegacySubstitutionMappingTools::computeSubstitutionVectorsNoAveraging -- nijt.average = no, nijt.joint = yes
 * (call sites CoMap/CoETools.cpp:395-403, CoMap/AnalysisTools.cpp:598-610; "for benchmarking only" there, but the only
 * way the reference runs nijt = Label with the MI statistic, CoETools.cpp:577-588).  The algorithm lives in bpp-phyl
 * (Bio++ 3.0 legacy classes), which is not part of the reference tree: PARITY UNPINNED, restated from the published
 * source as far as it is remembered:
 *   per branch b (father f, son n) and site i:  pxy(x, y) = sum_c p_c U_b(i,c,x) P_c,b(x,y) D_n(i,c,y)   (the joint
 *   posterior of the two ancestral states up to the factor 1 / L_i; U_b and D_n as in orc_map_sites / A.3);
 *   (x*, y*) = MatrixTools::whichMax(pxy): the FIRST maximum in row-major order;
 *   count(b, i, k) = N^k(x*, y*; t_b), the conditional expectation J/P of A.4 at the branch length itself (the rate
 *   class has been summed out, no r_c here).
 * Assumption where memory does not decide: the class weight p_c inside pxy (it does not matter for equiprobable
 * classes, i.e. for every Gamma(n) distribution).
 * margin[i*B + b] = (best - second best) / best of pxy: where it is below ~1e-9 another implementation may pick the
 * other cell; the parity tests skip those entries. */
int orc_map_sites_noavg(int nn, const int* parent, const double* blen, int T, const int* leaf_of_taxon, long N,
                        const uint8_t* aln, const uint32_t* masks, int S, int C, int K, const double* Q, const double* pi,
                        const double* rates, const double* probs, const double* Bk, int method, int nonneg,
                        const double* naiveW, double* counts, double* norm, int* argmax_xy, double* margin) {
  int B = nn - 1, root = nn - 1, S2 = S * S;
  int *first, *next;
  child_lists(nn, parent, &first, &next);
  int* taxon_of = (int*)malloc(sizeof(int) * nn);
  for (int i = 0; i < nn; i++) taxon_of[i] = -1;
  for (int t = 0; t < T; t++) taxon_of[leaf_of_taxon[t]] = t;
  double* lam = (double*)malloc(sizeof(double) * (S + 2 * S2));
  double *V = lam + S, *Vinv = V + S2;
  orc_eigen_reversible(S, Q, pi, lam, V, Vinv);
  double* P = (double*)malloc(sizeof(double) * (size_t)B * C * S2);
  double* N1 = (double*)malloc(sizeof(double) * (size_t)B * K * S2);   /* N^k(x, y; t_b) at rate 1 */
  double* Jtmp = (double*)malloc(sizeof(double) * S2 * 2);
  double* P1 = Jtmp + S2;
  for (int b = 0; b < B; b++) {
    for (int c = 0; c < C; c++) orc_transition(S, lam, V, Vinv, blen[b] * rates[c], P + ((size_t)b * C + c) * S2);
    orc_transition(S, lam, V, Vinv, blen[b], P1);
    for (int k = 0; k < K; k++) {
      double* Nk = N1 + ((size_t)b * K + k) * S2;
      if (method == 2) {
        for (int i = 0; i < S2; i++) Nk[i] = ((i / S) == (i % S)) ? 0.0 : (naiveW ? naiveW[i] : 1.0);
      } else {
        if (method == 0) orc_count_unif(S, Q, Bk + (size_t)k * S2, blen[b], Jtmp);
        else orc_count_decomp(S, lam, V, Vinv, Bk + (size_t)k * S2, blen[b], Jtmp);
        for (int i = 0; i < S2; i++) {
          double nxy = Jtmp[i] / P1[i];
          if (isnan(nxy) || isinf(nxy)) nxy = 0;
          if (nonneg && nxy < 0) nxy = 0;
          Nk[i] = nxy;
        }
      }
    }
  }
  size_t vsz = (size_t)nn * C * S;
  double* D = (double*)malloc(sizeof(double) * vsz * 3);
  double *M = D + vsz, *Up = M + vsz;
  double* U = (double*)malloc(sizeof(double) * (size_t)C * S);
  double* pxy = (double*)malloc(sizeof(double) * S2);
  for (long i = 0; i < N; i++) {
    for (int n = 0; n < nn; n++)
      for (int c = 0; c < C; c++) {
        double* Dn = D + ((size_t)n * C + c) * S;
        if (first[n] < 0) {
          uint8_t code = aln[(size_t)taxon_of[n] * N + i];
          for (int x = 0; x < S; x++) Dn[x] = leaf_compatible(masks, S, code, x);
        } else {
          for (int x = 0; x < S; x++) Dn[x] = 1.0;
          for (int e = first[n]; e >= 0; e = next[e]) {
            const double* Me = M + ((size_t)e * C + c) * S;
            for (int x = 0; x < S; x++) Dn[x] *= Me[x];
          }
        }
        if (n != root) {
          const double* Pn = P + ((size_t)n * C + c) * S2;
          double* Mn = M + ((size_t)n * C + c) * S;
          for (int x = 0; x < S; x++) {
            double s = 0;
            for (int z = 0; z < S; z++) s += Pn[x * S + z] * Dn[z];
            Mn[x] = s;
          }
        }
      }
    for (int c = 0; c < C; c++)
      for (int x = 0; x < S; x++) Up[((size_t)root * C + c) * S + x] = pi[x];
    double* ci = counts + (size_t)i * B * K;
    for (int f = nn - 1; f >= 0; f--) {
      if (first[f] < 0) continue;
      for (int n = first[f]; n >= 0; n = next[n]) {
        for (int e = 0; e < S2; e++) pxy[e] = 0;
        for (int c = 0; c < C; c++) {
          const double* Upf = Up + ((size_t)f * C + c) * S;
          double* Uc = U + (size_t)c * S;
          for (int x = 0; x < S; x++) Uc[x] = Upf[x];
          for (int m = first[f]; m >= 0; m = next[m])
            if (m != n) {
              const double* Mm = M + ((size_t)m * C + c) * S;
              for (int x = 0; x < S; x++) Uc[x] *= Mm[x];
            }
          const double* Dn = D + ((size_t)n * C + c) * S;
          const double* Pn = P + ((size_t)n * C + c) * S2;
          for (int x = 0; x < S; x++)
            for (int y = 0; y < S; y++) pxy[x * S + y] += probs[c] * (Uc[x] * Pn[x * S + y] * Dn[y]);
          if (first[n] >= 0) {
            double* Upn = Up + ((size_t)n * C + c) * S;
            for (int z = 0; z < S; z++) {
              double s = 0;
              for (int x = 0; x < S; x++) s += Pn[x * S + z] * Uc[x];
              Upn[z] = s;
            }
          }
        }
        int best = 0;
        double bv = -INFINITY, second = -INFINITY;
        for (int e = 0; e < S2; e++) {
          if (pxy[e] > bv) { second = bv; bv = pxy[e]; best = e; }
          else if (pxy[e] > second) second = pxy[e];
        }
        for (int k = 0; k < K; k++) ci[(size_t)n * K + k] = N1[((size_t)n * K + k) * S2 + best];
        if (argmax_xy) argmax_xy[(size_t)i * B + n] = best;
        if (margin) margin[(size_t)i * B + n] = bv > 0 ? (bv - second) / bv : 0.0;
      }
    }
    double nrm = 0;
    for (int b = 0; b < B; b++) {
      double tot = 0;
      for (int k = 0; k < K; k++) tot += ci[(size_t)b * K + k];
      nrm += tot * tot;
    }
    norm[i] = sqrt(nrm);
  }
  free(D); free(U); free(pxy); free(P); free(N1); free(Jtmp); free(lam); free(taxon_of); free(first); free(next);
  return 0;
}

/* nijt.joint = no: LegacySubstitutionMappingTools::computeSubstitutionVectorsMarginal (average = yes) and
 * computeSubstitutionVectorsNoAveragingMarginal (average = no) -- CoMap/CoETools.cpp:399-405,
 * CoMap/AnalysisTools.cpp:598-610, 620-633; "for benchmarking only" in the reference.  bpp-phyl (Bio++ 3.0 legacy
 * classes) is not part of the reference tree: PARITY UNPINNED, restated from the published source as far as it is
 * remembered, and pinned to its definition by brute force in tests/test_oracle_marginal.py:
 *   DRTreeLikelihoodTools::getPosteriorProbabilitiesPerStatePerRate(drtl, node)[i][c][x]
 *     internal node: p_c L_node(i, c, x) / L_i with L_node(i, c, x) = P(data, state x at the node | class c) -- the
 *       likelihood re-rooted at the node, root frequencies included (computeLikelihoodAtNode); in the arrays of
 *       orc_map_sites: Up_node(c, x) D_node(c, x) (Up_root = pi).  This IS the joint posterior of (class, state).
 *     leaf: e(x) p_c / sum_s e(s), e the leaf's 0/1 compatibility vector -- the PRIOR class weight, as the remembered
 *       source has it (`larray[x] * rcProbs[c] / sumprobs`), not the posterior one.
 *   average = yes:  n(b, i, k) = sum_c sum_x sum_y post_father(i,c,x) post_node(i,c,y) N^k(x, y; r_c t_b)
 *     (the product of the two marginal posteriors in place of the joint one; note the class weight enters twice).
 *   average = no:   MarginalAncestralStateReconstruction: x*(node) = first maximum over x of sum_c p_c L_node(i,c,x)
 *     (leaf: first maximum of e, i.e. the observed state, state 0 for an unknown); n(b, i, k) = N^k(x*(father),
 *     x*(node); t_b) at the branch length itself.
 * post (optional): [N][nn][C][S]; anc (optional): [N][nn]; margin (optional, [N][nn]): (best - second) / best of the
 * marginal state posterior -- where it is tiny another implementation may pick the other state. */
int orc_map_sites_marginal(int nn, const int* parent, const double* blen, int T, const int* leaf_of_taxon, long N,
                           const uint8_t* aln, const uint32_t* masks, int S, int C, int K, const double* Q, const double* pi,
                           const double* rates, const double* probs, const double* Bk, int method, int nonneg,
                           const double* naiveW, int average, double* counts, double* norm, double* post, int* anc,
                           double* margin) {
  int B = nn - 1, root = nn - 1, S2 = S * S;
  int *first, *next;
  child_lists(nn, parent, &first, &next);
  int* taxon_of = (int*)malloc(sizeof(int) * nn);
  for (int i = 0; i < nn; i++) taxon_of[i] = -1;
  for (int t = 0; t < T; t++) taxon_of[leaf_of_taxon[t]] = t;
  double* lam = (double*)malloc(sizeof(double) * (S + 2 * S2));
  double *V = lam + S, *Vinv = V + S2;
  orc_eigen_reversible(S, Q, pi, lam, V, Vinv);
  double* P = (double*)malloc(sizeof(double) * (size_t)B * C * S2);
  /* conditional counts N^k(x, y; t): per class at r_c t_b (slots 0 .. C-1) and at t_b itself (slot C) */
  double* NC = (double*)malloc(sizeof(double) * (size_t)B * (C + 1) * K * S2);
  double* Jtmp = (double*)malloc(sizeof(double) * S2 * 2);
  double* Pt = Jtmp + S2;
  for (int b = 0; b < B; b++)
    for (int c = 0; c <= C; c++) {
      double t = blen[b] * (c < C ? rates[c] : 1.0);
      orc_transition(S, lam, V, Vinv, t, Pt);
      if (c < C) memcpy(P + ((size_t)b * C + c) * S2, Pt, sizeof(double) * S2);
      for (int k = 0; k < K; k++) {
        double* Nk = NC + (((size_t)b * (C + 1) + c) * K + k) * S2;
        if (method == 2) {
          for (int i = 0; i < S2; i++) Nk[i] = ((i / S) == (i % S)) ? 0.0 : (naiveW ? naiveW[i] : 1.0);
        } else {
          if (method == 0) orc_count_unif(S, Q, Bk + (size_t)k * S2, t, Jtmp);
          else orc_count_decomp(S, lam, V, Vinv, Bk + (size_t)k * S2, t, Jtmp);
          for (int i = 0; i < S2; i++) {
            double nxy = Jtmp[i] / Pt[i];
            if (isnan(nxy) || isinf(nxy)) nxy = 0;
            if (nonneg && nxy < 0) nxy = 0;
            Nk[i] = nxy;
          }
        }
      }
    }
  size_t vsz = (size_t)nn * C * S;
  double* D = (double*)malloc(sizeof(double) * vsz * 4);
  double *M = D + vsz, *Up = M + vsz, *PP = Up + vsz;   /* PP: posterior per node, class, state */
  double* U = (double*)malloc(sizeof(double) * S);
  int* st = (int*)malloc(sizeof(int) * nn);
  for (long i = 0; i < N; i++) {
    for (int n = 0; n < nn; n++)
      for (int c = 0; c < C; c++) {
        double* Dn = D + ((size_t)n * C + c) * S;
        if (first[n] < 0) {
          uint8_t code = aln[(size_t)taxon_of[n] * N + i];
          for (int x = 0; x < S; x++) Dn[x] = leaf_compatible(masks, S, code, x);
        } else {
          for (int x = 0; x < S; x++) Dn[x] = 1.0;
          for (int e = first[n]; e >= 0; e = next[e]) {
            const double* Me = M + ((size_t)e * C + c) * S;
            for (int x = 0; x < S; x++) Dn[x] *= Me[x];
          }
        }
        if (n != root) {
          const double* Pn = P + ((size_t)n * C + c) * S2;
          double* Mn = M + ((size_t)n * C + c) * S;
          for (int x = 0; x < S; x++) {
            double s = 0;
            for (int z = 0; z < S; z++) s += Pn[x * S + z] * Dn[z];
            Mn[x] = s;
          }
        }
      }
    double L = 0;
    for (int c = 0; c < C; c++) {
      const double* Dr = D + ((size_t)root * C + c) * S;
      double s = 0;
      for (int x = 0; x < S; x++) s += pi[x] * Dr[x];
      L += probs[c] * s;
    }
    for (int c = 0; c < C; c++)
      for (int x = 0; x < S; x++) Up[((size_t)root * C + c) * S + x] = pi[x];
    for (int f = nn - 1; f >= 0; f--) {
      if (first[f] < 0) continue;
      for (int n = first[f]; n >= 0; n = next[n]) {
        if (first[n] < 0) continue;
        for (int c = 0; c < C; c++) {
          const double* Upf = Up + ((size_t)f * C + c) * S;
          for (int x = 0; x < S; x++) U[x] = Upf[x];
          for (int m = first[f]; m >= 0; m = next[m])
            if (m != n) {
              const double* Mm = M + ((size_t)m * C + c) * S;
              for (int x = 0; x < S; x++) U[x] *= Mm[x];
            }
          const double* Pn = P + ((size_t)n * C + c) * S2;
          double* Upn = Up + ((size_t)n * C + c) * S;
          for (int z = 0; z < S; z++) {
            double s = 0;
            for (int x = 0; x < S; x++) s += Pn[x * S + z] * U[x];
            Upn[z] = s;
          }
        }
      }
    }
    /* posteriors per state per rate, marginal ancestral states */
    for (int n = 0; n < nn; n++) {
      double bestv = -INFINITY, second = -INFINITY;
      int best = 0;
      if (first[n] < 0) {
        const double* e = D + ((size_t)n * C) * S;
        double se = 0;
        for (int x = 0; x < S; x++) se += e[x];
        for (int c = 0; c < C; c++)
          for (int x = 0; x < S; x++) PP[((size_t)n * C + c) * S + x] = e[x] * probs[c] / se;
        for (int x = 0; x < S; x++) {
          if (e[x] > bestv) { second = bestv; bestv = e[x]; best = x; }
          else if (e[x] > second) second = e[x];
        }
      } else {
        for (int c = 0; c < C; c++)
          for (int x = 0; x < S; x++)
            PP[((size_t)n * C + c) * S + x] = Up[((size_t)n * C + c) * S + x] * D[((size_t)n * C + c) * S + x] * probs[c] / L;
        for (int x = 0; x < S; x++) {
          double s = 0;
          for (int c = 0; c < C; c++) s += PP[((size_t)n * C + c) * S + x];
          if (s > bestv) { second = bestv; bestv = s; best = x; }
          else if (s > second) second = s;
        }
      }
      st[n] = best;
      if (anc) anc[(size_t)i * nn + n] = best;
      if (margin) margin[(size_t)i * nn + n] = bestv > 0 ? (bestv - second) / bestv : 0.0;
    }
    if (post) memcpy(post + (size_t)i * vsz, PP, sizeof(double) * vsz);
    double* ci = counts + (size_t)i * B * K;
    double nrm = 0;
    for (int b = 0; b < B; b++) {
      int f = parent[b];
      double tot = 0;
      for (int k = 0; k < K; k++) {
        double v = 0;
        if (average) {
          for (int c = 0; c < C; c++) {
            const double* Nk = NC + (((size_t)b * (C + 1) + c) * K + k) * S2;
            const double *pf = PP + ((size_t)f * C + c) * S, *pn = PP + ((size_t)b * C + c) * S;
            for (int x = 0; x < S; x++) {
              double s = 0;
              for (int y = 0; y < S; y++) s += Nk[x * S + y] * pn[y];
              v += pf[x] * s;
            }
          }
        } else {
          v = NC[(((size_t)b * (C + 1) + C) * K + k) * S2 + st[f] * S + st[b]];
        }
        ci[(size_t)b * K + k] = v;
        tot += v;
      }
      nrm += tot * tot;
    }
    norm[i] = sqrt(nrm);
  }
  free(D); free(U); free(st); free(P); free(NC); free(Jtmp); free(lam); free(taxon_of); free(first); free(next);
  return 0;
}

/* ------------------------------------------------------------------ simulator (A.8; RNG scheme is this build's)
 * Bio++ draws from a global, time-seeded generator, so simulated alignments are not reproducible across
 * implementations; the product and this oracle share a counter-based scheme instead (Philox2x32-10, Random123):
 *   key = seed_lo ^ seed_hi * 0x9E3779B9 ^ 'CMX2'; counter = (g_lo, g_hi[14:0] | draw << 15) with g the global index of
 *   the simulated site and draw = 0 (rate class), 1 (root state), 2 + node (state at the lower end of branch `node`);
 *   draws 0 and 1: u = ((r0 << 32 | r1) >> 11) * 2^-53.  Node draws: the sites 2k and 2k + 1 share the call with counter
 *   g >> 1 and take r0 resp. r1 as a 32-bit uniform, u = r * 2^-32 (orc_node_uniform) -- they are 99 % of the calls, and a
 *   device thread holding both sites makes one call for two draws.  index = #{ j < n-1 : u >= cum[j] }, cum the running sum.
 *   (Philox4x32-10 below serves the Mica permutation test.) */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                 uint32_t* out) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1,
             n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline void philox2x32_10(uint64_t seed, uint64_t g, uint32_t draw, uint32_t* w) {
  uint32_t c0 = (uint32_t)g, c1 = ((uint32_t)(g >> 32) & 0x7fffu) | (draw << 15);
  uint32_t k = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u) ^ 0x434d5832u;
  for (int r = 0; r < 10; r++) {
    uint64_t p = (uint64_t)0xD256D193u * c0;
    uint32_t n0 = (uint32_t)(p >> 32) ^ k ^ c1;
    c1 = (uint32_t)p;
    c0 = n0;
    k += 0x9E3779B9u;
  }
  w[0] = c0;
  w[1] = c1;
}
double orc_uniform(uint64_t seed, uint64_t g, uint32_t draw) {
  uint32_t w[2];
  philox2x32_10(seed, g, draw, w);
  uint64_t bits = (((uint64_t)w[0] << 32) | w[1]) >> 11;
  return (double)bits * (1.0 / 9007199254740992.0);
}
double orc_node_uniform(uint64_t seed, uint64_t g, uint32_t node) {
  uint32_t w[2];
  philox2x32_10(seed, g >> 1, 2u + node, w);
  return (double)w[g & 1] * (1.0 / 4294967296.0);
}

static int draw_index(double u, const double* p, int n) {
  double cum = 0;
  int idx = 0;
  for (int j = 0; j < n - 1; j++) {
    cum += p[j];
    if (u >= cum) idx = j + 1;
  }
  return idx;
}

/* NonHomogeneousSequenceSimulator::simulate(n) restated (homogeneous model, discrete rates):
 * aln[t*n + j]; global site indices g0 .. g0+n-1. */
void orc_simulate(int nn, const int* parent, const double* blen, int T, const int* leaf_of_taxon, int S, int C,
                  const double* Q, const double* pi, const double* rates, const double* probs, uint64_t seed,
                  uint64_t g0, long n, uint8_t* aln, int* classes) {
  int S2 = S * S, B = nn - 1, root = nn - 1;
  double* lam = (double*)malloc(sizeof(double) * (S + 2 * S2));
  double *V = lam + S, *Vinv = V + S2;
  orc_eigen_reversible(S, Q, pi, lam, V, Vinv);
  double* P = (double*)malloc(sizeof(double) * (size_t)B * C * S2);
  for (int b = 0; b < B; b++)
    for (int c = 0; c < C; c++) orc_transition(S, lam, V, Vinv, blen[b] * rates[c], P + ((size_t)b * C + c) * S2);
  int* taxon_of = (int*)malloc(sizeof(int) * nn);
  for (int i = 0; i < nn; i++) taxon_of[i] = -1;
  for (int t = 0; t < T; t++) taxon_of[leaf_of_taxon[t]] = t;
  uint8_t* st = (uint8_t*)malloc(nn);
  for (long j = 0; j < n; j++) {
    uint64_t g = g0 + (uint64_t)j;
    int c = draw_index(orc_uniform(seed, g, 0), probs, C);
    if (classes) classes[j] = c;
    st[root] = (uint8_t)draw_index(orc_uniform(seed, g, 1), pi, S);
    for (int node = nn - 2; node >= 0; node--) { /* parents have larger ids (post-order) */
      int x = st[parent[node]];
      st[node] = (uint8_t)draw_index(orc_node_uniform(seed, g, (uint32_t)node), P + ((size_t)node * C + c) * S2 + (size_t)x * S, S);
      if (taxon_of[node] >= 0) aln[(size_t)taxon_of[node] * n + j] = st[node];
    }
  }
  free(st); free(taxon_of); free(P); free(lam);
}

/* ------------------------------------------------------------------ simulations.continuous = yes (CoMap/CoMap.cpp:146, 213)
 * NonHomogeneousSequenceSimulator::enableContinuousRates: per site a rate from the continuous Gamma(alpha, beta = alpha)
 * (Invariant(Gamma): 0 with probability p_inv, else the draw / (1 - p_inv)), per branch P = exp(Q r t) of that rate (A.8).
 * Quantile by the textbook route: regularised incomplete gamma (series / continued fraction), bracket by doubling, 110
 * bisection steps.  orc_gamma_quantile is checked against scipy.stats.gamma.ppf in tests/test_oracle_continuous.py. */
void orc_gamma_pq(double a, double x, double* p, double* q) {
  /* regularised incomplete gamma, lower P and upper Q = 1 - P, each from the expansion that gives it without
   * cancellation: series for x < a + 1 (P), Lentz continued fraction otherwise (Q) */
  if (x <= 0.0) { *p = 0.0; *q = 1.0; return; }
  const double pre = exp(-x + a * log(x) - lgamma(a));
  if (x < a + 1.0) {
    double term = 1.0 / a, sum = term;
    for (int n = 1; n < 1000; ++n) {
      term *= x / (a + n);
      sum += term;
      if (fabs(term) < fabs(sum) * 1e-17) break;
    }
    *p = sum * pre;
    *q = 1.0 - *p;
    return;
  }
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 1000; ++i) {
    const double an = -(double)i * ((double)i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < tiny) d = tiny;
    c = b + an / c;
    if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  *q = pre * h;
  *p = 1.0 - *q;
}
/* "x is below the u-quantile": decided on the tail that carries the information (P < u, or Q > 1 - u for u > 1/2) */
int orc_gamma_below(double a, double x, double u) {
  double p, q;
  orc_gamma_pq(a, x, &p, &q);
  return u <= 0.5 ? p < u : q > 1.0 - u;
}
/* quantile of Gamma(shape a, scale 1): bracket [lo, 2 lo] by doubling / halving from 1, then 110 bisection steps
 * (deterministic, no tolerance test) */
double orc_gamma_quantile(double a, double u) {
  if (u <= 0.0) return 0.0;
  double lo = 1.0, hi;
  if (orc_gamma_below(a, lo, u)) {
    for (int i = 0; i < 1100 && orc_gamma_below(a, 2.0 * lo, u); ++i) lo *= 2.0;
    hi = 2.0 * lo;
  } else {
    hi = lo;
    lo = 0.5 * hi;
    for (int i = 0; i < 1070 && !orc_gamma_below(a, lo, u); ++i) { hi = lo; lo *= 0.5; }
  }
  for (int i = 0; i < 110; ++i) {
    const double mid = 0.5 * (lo + hi);
    if (orc_gamma_below(a, mid, u)) lo = mid; else hi = mid;
  }
  return 0.5 * (lo + hi);
}

void orc_simulate_continuous(int nn, const int* parent, const double* blen, int T, const int* leaf_of_taxon, int S,
                             const double* Q, const double* pi, double alpha, double p_inv, uint64_t seed, uint64_t g0,
                             long n, uint8_t* aln, double* rates) {
  int S2 = S * S, root = nn - 1;
  double* lam = (double*)malloc(sizeof(double) * (S + 3 * S2));
  double *V = lam + S, *Vinv = V + S2, *P = Vinv + S2;
  orc_eigen_reversible(S, Q, pi, lam, V, Vinv);
  int* taxon_of = (int*)malloc(sizeof(int) * nn);
  for (int i = 0; i < nn; i++) taxon_of[i] = -1;
  for (int t = 0; t < T; t++) taxon_of[leaf_of_taxon[t]] = t;
  uint8_t* st = (uint8_t*)malloc(nn);
  for (long j = 0; j < n; j++) {
    uint64_t g = g0 + (uint64_t)j;
    double u0 = orc_uniform(seed, g, 0), r = 0.0;
    if (u0 >= p_inv) r = orc_gamma_quantile(alpha, (u0 - p_inv) / (1.0 - p_inv)) / alpha / (1.0 - p_inv);
    if (rates) rates[j] = r;
    st[root] = (uint8_t)draw_index(orc_uniform(seed, g, 1), pi, S);
    for (int node = nn - 2; node >= 0; node--) {
      int x = st[parent[node]];
      orc_transition(S, lam, V, Vinv, blen[node] * r, P);   /* getPij_t(d * rate) of this very site */
      st[node] = (uint8_t)draw_index(orc_node_uniform(seed, g, (uint32_t)node), P + (size_t)x * S, S);
      if (taxon_of[node] >= 0) aln[(size_t)taxon_of[node] * n + j] = st[node];
    }
  }
  free(st); free(taxon_of); free(lam);
}

/* ------------------------------------------------------------------ Domain (CoMap/Domain.cpp:46-59, 113-122) */
int orc_domain_index(double a, double b, int n, double x) {
  double mini = a < b ? a : b, maxi = a < b ? b : a;
  double w = (maxi - mini) / (double)n;
  if (x < mini || x >= mini + (double)n * w) return -1; /* OutOfRangeException */
  for (int i = 1; i < n + 1; i++)
    if (x < mini + (double)i * w) return i - 1;
  return -1;
}

int orc_domain_index_bounds(const double* bounds, int nb, double x) {
  if (x < bounds[0] || x >= bounds[nb - 1]) return -1;
  for (int i = 1; i < nb; i++)
    if (x < bounds[i]) return i - 1;
  return -1;
}

/* ------------------------------------------------------------------ statistics (CoMap/Statistics.h) */
enum { ST_CORRELATION = 0, ST_COMPENSATION = 1, ST_COSUBSTITUTION = 2, ST_COSINUS = 3, ST_COVARIANCE = 4,
       ST_DISCRETE_MI = 5, ST_CORRECTED_CORRELATION = 6, ST_EUCLIDIAN_DISTANCE = 7 };

static double vsum(const double* v, int K) {
  double s = 0;
  for (int k = 0; k < K; k++) s += v[k];
  return s;
}

/* v1, v2: [B][K].  params: for ST_DISCRETE_MI params[0] = number of bounds, params[1..] = bounds
 * (CoETools.cpp:590-593: {0, threshold, 10000}). */
double orc_stat_pair(int kind, int B, int K, const double* v1, const double* v2, const double* params) {
  switch (kind) {
    case ST_CORRELATION: case ST_COVARIANCE: { /* Statistics.h:164-174,206-216 -> VectorTools::cor/cov on type 0 (A.7) */
      double m1 = 0, m2 = 0;
      for (int b = 0; b < B; b++) { m1 += v1[(size_t)b * K]; m2 += v2[(size_t)b * K]; }
      m1 /= B; m2 /= B;
      double sxy = 0, sxx = 0, syy = 0;
      for (int b = 0; b < B; b++) {
        double dx = v1[(size_t)b * K] - m1, dy = v2[(size_t)b * K] - m2;
        sxy += dx * dy; sxx += dx * dx; syy += dy * dy;
      }
      double cov = sxy / (B - 1);
      if (kind == ST_COVARIANCE) return cov;
      return cov / (sqrt(sxx / (B - 1)) * sqrt(syy / (B - 1)));
    }
    case ST_CORRECTED_CORRELATION: { /* Statistics.h:176-204: cor(v1 - meanVector1_, v2 - meanVector2_), params = [2][B] */
      double m1 = 0, m2 = 0;
      for (int b = 0; b < B; b++) { m1 += v1[(size_t)b * K] - params[b]; m2 += v2[(size_t)b * K] - params[B + b]; }
      m1 /= B; m2 /= B;
      double sxy = 0, sxx = 0, syy = 0;
      for (int b = 0; b < B; b++) {
        double dx = v1[(size_t)b * K] - params[b] - m1, dy = v2[(size_t)b * K] - params[B + b] - m2;
        sxy += dx * dy; sxx += dx * dx; syy += dy * dy;
      }
      return (sxy / (B - 1)) / (sqrt(sxx / (B - 1)) * sqrt(syy / (B - 1)));
    }
    case ST_EUCLIDIAN_DISTANCE: { /* Distance.h:157-171 */
      double d = 0;
      for (int b = 0; b < B; b++) {
        double sv1 = vsum(v1 + (size_t)b * K, K), sv2 = vsum(v2 + (size_t)b * K, K);
        d += pow(sv2 - sv1, 2);
      }
      return sqrt(d);
    }
    case ST_COSINUS: { /* Statistics.h:218-228 */
      double sxy = 0, sxx = 0, syy = 0;
      for (int b = 0; b < B; b++) {
        double x = v1[(size_t)b * K], y = v2[(size_t)b * K];
        sxy += x * y; sxx += x * x; syy += y * y;
      }
      return sxy / (sqrt(sxx) * sqrt(syy));
    }
    case ST_COMPENSATION: { /* Statistics.h:247-265 */
      double s1 = 0, s2 = 0, s3 = 0;
      for (int b = 0; b < B; b++) {
        double a = vsum(v1 + (size_t)b * K, K), c = vsum(v2 + (size_t)b * K, K);
        s1 += pow(a, 2); s2 += pow(c, 2); s3 += pow(a + c, 2);
      }
      return 1. - sqrt(s3) / (sqrt(s1) + sqrt(s2));
    }
    case ST_COSUBSTITUTION: { /* Statistics.h:230-245 */
      double c = 0;
      for (int b = 0; b < B; b++)
        if (vsum(v1 + (size_t)b * K, K) >= 1. && vsum(v2 + (size_t)b * K, K) >= 1.) c++;
      return c;
    }
    case ST_DISCRETE_MI: { /* Statistics.h:307-327 -> VectorTools::miDiscrete(c1, c2, base=2.7182818, unbiased=false) */
      int nb = (int)params[0], ncls = nb - 1;
      double* n12 = (double*)calloc((size_t)ncls * ncls + 2 * ncls, sizeof(double));
      double *n1 = n12 + (size_t)ncls * ncls, *n2 = n1 + ncls;
      for (int b = 0; b < B; b++) {
        int a = orc_domain_index_bounds(params + 1, nb, vsum(v1 + (size_t)b * K, K));
        int c = orc_domain_index_bounds(params + 1, nb, vsum(v2 + (size_t)b * K, K));
        if (a < 0 || c < 0) { free(n12); return NAN; } /* OutOfRangeException propagates in the reference */
        n12[a * ncls + c] += 1; n1[a] += 1; n2[c] += 1;
      }
      double s = 0, np = B;
      for (int a = 0; a < ncls; a++)
        for (int c = 0; c < ncls; c++)
          if (n12[a * ncls + c] > 0) s += (n12[a * ncls + c] / np) * log(n12[a * ncls + c] * np / (n1[a] * n2[c]));
      free(n12);
      return s / log(2.7182818);
    }
  }
  return NAN;
}

/* CoETools::computeIntraStats pair loop, dense form (CoETools.cpp:672-692): out[i*N+j] for i<j, NaN elsewhere */
void orc_pair_stats_intra(int kind, long N, int B, int K, const double* counts, const double* params, double* out) {
  for (long i = 0; i < N; i++)
    for (long j = 0; j < N; j++)
      out[i * N + j] = (j > i) ? orc_stat_pair(kind, B, K, counts + (size_t)i * B * K, counts + (size_t)j * B * K, params) : NAN;
}

void orc_pair_stats_inter(int kind, long N1, long N2, int B, int K, const double* c1, const double* c2,
                          const double* params, double* out) {
  for (long i = 0; i < N1; i++)
    for (long j = 0; j < N2; j++)
      out[i * N2 + j] = orc_stat_pair(kind, B, K, c1 + (size_t)i * B * K, c2 + (size_t)j * B * K, params);
}

/* ------------------------------------------------------------------ null distribution (AnalysisTools.cpp:564-658)
 * Loop structure kept: per replicate simulate repRAM sites twice, re-initialise the likelihood, re-map, then
 * score site j of batch 1 against site j of batch 2.  Replicates rep_begin..rep_end-1; global simulated-site
 * index g = ((rep*2 + batch) * repRAM + j).  Outputs have (rep_end-rep_begin)*repRAM entries. */
void orc_null_intra(int nn, const int* parent, const double* blen, int T, const int* leaf_of_taxon, int S, int C,
                    int K, const double* Q, const double* pi, const double* rates, const double* probs,
                    const double* Bk, int method, int nonneg, int kind, const double* sparams, uint64_t seed,
                    long rep_begin, long rep_end, long repRAM, const uint8_t* supplied, double* stat, int* rcmin,
                    double* prmin, double* nmin) {
  int B = nn - 1;
  uint8_t* aln = (uint8_t*)malloc((size_t)T * repRAM);
  uint32_t* masks = (uint32_t*)malloc(sizeof(uint32_t) * 256);
  for (int i = 0; i < 256; i++) masks[i] = i < 32 ? (1u << i) : 0xffffffffu;
  double* cnt[2];
  double *ll = (double*)malloc(sizeof(double) * repRAM), *pr[2], *nr[2];
  int* rc[2];
  for (int h = 0; h < 2; h++) {
    cnt[h] = (double*)malloc(sizeof(double) * (size_t)repRAM * B * K);
    pr[h] = (double*)malloc(sizeof(double) * repRAM);
    nr[h] = (double*)malloc(sizeof(double) * repRAM);
    rc[h] = (int*)malloc(sizeof(int) * repRAM);
  }
  for (long rep = rep_begin; rep < rep_end; rep++) {
    for (int h = 0; h < 2; h++) {
      const uint8_t* a = aln;
      if (supplied) a = supplied + ((size_t)(rep - rep_begin) * 2 + h) * T * repRAM;
      else orc_simulate(nn, parent, blen, T, leaf_of_taxon, S, C, Q, pi, rates, probs, seed,
                        ((uint64_t)rep * 2 + h) * (uint64_t)repRAM, repRAM, aln, 0);
      orc_map_sites(nn, parent, blen, T, leaf_of_taxon, repRAM, a, masks, S, C, K, Q, pi, rates, probs, Bk, method,
                    nonneg, 0, cnt[h], ll, pr[h], rc[h], nr[h]);
    }
    for (long j = 0; j < repRAM; j++) {
      size_t o = (size_t)(rep - rep_begin) * repRAM + j;
      stat[o] = orc_stat_pair(kind, B, K, cnt[0] + (size_t)j * B * K, cnt[1] + (size_t)j * B * K, sparams);
      rcmin[o] = rc[0][j] < rc[1][j] ? rc[0][j] : rc[1][j];
      prmin[o] = pr[0][j] < pr[1][j] ? pr[0][j] : pr[1][j];
      nmin[o] = nr[0][j] < nr[1][j] ? nr[0][j] : nr[1][j];
    }
  }
  for (int h = 0; h < 2; h++) { free(cnt[h]); free(pr[h]); free(nr[h]); free(rc[h]); }
  free(ll); free(masks); free(aln);
}

/* ------------------------------------------------------------------ p-values (CoETools.cpp:636-652, 695-721)
 * null stats are binned by Domain(0, max(norms), nclasses) on nmin (AnalysisTools.cpp:643-652, out of range dropped),
 * each class sorted ascending; p = (nsim - #{null < stat} + 1)/(nsim + 1) by a linear scan with strict '<'.
 * NaN null values: std::sort on NaN is undefined in the reference; here NaNs are dropped from the class (documented
 * divergence).  Pairs whose min norm falls outside [0, max norm) get pvalue = NaN ("NA"), nsim = 0. */
static int cmp_double(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

void orc_intra_pvalues(long N, const double* stat /*[N*N], i<j*/, const double* norms, int nclasses, long nnull,
                       const double* null_stat, const double* null_nmin, double* pvalue, int* nsim) {
  double maxn = 0;
  for (long i = 0; i < N; i++)
    if (norms[i] > maxn) maxn = norms[i];
  long* cnt = (long*)calloc(nclasses, sizeof(long));
  int* cls = (int*)malloc(sizeof(int) * (nnull > 0 ? nnull : 1));
  for (long s = 0; s < nnull; s++) {
    cls[s] = isnan(null_stat[s]) ? -1 : orc_domain_index(0, maxn, nclasses, null_nmin[s]);
    if (cls[s] >= 0) cnt[cls[s]]++;
  }
  double** sim = (double**)malloc(sizeof(double*) * nclasses);
  long* fill = (long*)calloc(nclasses, sizeof(long));
  for (int c = 0; c < nclasses; c++) sim[c] = (double*)malloc(sizeof(double) * (cnt[c] > 0 ? cnt[c] : 1));
  for (long s = 0; s < nnull; s++)
    if (cls[s] >= 0) sim[cls[s]][fill[cls[s]]++] = null_stat[s];
  for (int c = 0; c < nclasses; c++) qsort(sim[c], cnt[c], sizeof(double), cmp_double);
  for (long i = 0; i < N; i++)
    for (long j = 0; j < N; j++) {
      size_t o = (size_t)i * N + j;
      if (j <= i) { pvalue[o] = NAN; nsim[o] = 0; continue; }
      double mn = norms[i] < norms[j] ? norms[i] : norms[j];
      int cat = orc_domain_index(0, maxn, nclasses, mn);
      if (cat < 0) { pvalue[o] = NAN; nsim[o] = 0; continue; }
      long ns = cnt[cat], count;
      for (count = 0; count < ns && sim[cat][count] < stat[o]; ++count) {}
      pvalue[o] = (double)(ns - count + 1) / (double)(ns + 1);
      nsim[o] = (int)ns;
    }
  for (int c = 0; c < nclasses; c++) free(sim[c]);
  free(sim); free(fill); free(cls); free(cnt);
}

/* ------------------------------------------------------------------ Mica column MI (Mica.cpp:93-118, 346-361; A.9)
 * SiteTools::mutualInformation / jointEntropy / entropy over taxa, natural log, resolveUnknowns = true:
 * an ambiguous symbol spreads its unit count uniformly over its compatible states.
 * aln1: [T][N1], aln2: [T][N2] (pass the same pointer for the intra case).  Outputs dense [N1][N2]. */
void orc_mi_columns(int T, int A, const uint32_t* masks, long N1, const uint8_t* aln1, long N2, const uint8_t* aln2,
                    double* mi, double* hjoint, double* h1, double* h2) {
  double* p12 = (double*)malloc(sizeof(double) * (size_t)(A * A + 2 * A));
  double *p1 = p12 + A * A, *p2 = p1 + A;
  for (int which = 0; which < 2; which++) {
    long Nn = which ? N2 : N1;
    const uint8_t* aln = which ? aln2 : aln1;
    double* h = which ? h2 : h1;
    for (long i = 0; i < Nn; i++) {
      for (int a = 0; a < A; a++) p1[a] = 0;
      for (int t = 0; t < T; t++) {
        uint8_t code = aln[(size_t)t * Nn + i];
        uint32_t m = code < A ? (1u << code) : masks[code];
        int n = __builtin_popcount(m);
        for (int a = 0; a < A; a++)
          if ((m >> a) & 1u) p1[a] += 1.0 / n;
      }
      double s = 0;
      for (int a = 0; a < A; a++)
        if (p1[a] > 0) s -= (p1[a] / T) * log(p1[a] / T);
      h[i] = s;
    }
  }
  for (long i = 0; i < N1; i++)
    for (long j = 0; j < N2; j++) {
      for (int a = 0; a < A * A + 2 * A; a++) p12[a] = 0;
      for (int t = 0; t < T; t++) {
        uint8_t c1 = aln1[(size_t)t * N1 + i], c2 = aln2[(size_t)t * N2 + j];
        uint32_t m1 = c1 < A ? (1u << c1) : masks[c1], m2 = c2 < A ? (1u << c2) : masks[c2];
        double w = 1.0 / (__builtin_popcount(m1) * __builtin_popcount(m2));
        for (int a = 0; a < A; a++)
          if ((m1 >> a) & 1u)
            for (int b = 0; b < A; b++)
              if ((m2 >> b) & 1u) { p12[a * A + b] += w; p1[a] += w; p2[b] += w; }
      }
      double s = 0, hj = 0;
      for (int a = 0; a < A; a++)
        for (int b = 0; b < A; b++) {
          double pab = p12[a * A + b] / T;
          if (pab > 0) {
            s += pab * log(pab / ((p1[a] / T) * (p2[b] / T)));
            hj -= pab * log(pab);
          }
        }
      mi[(size_t)i * N2 + j] = s;
      hjoint[(size_t)i * N2 + j] = hj;
    }
  free(p12);
}


/* ------------------------------------------------------------------ Mica permutation test
 * miTest (CoMap/Mica.cpp:93-118) for the column pairs [pair_begin, pair_end) of the row-major (i < j) order; first the
 * scheme for fully resolved pairs, below it the one for pairs with unknowns.  PARITY UNPINNED (no reference output; Bio++'s Site::shuffle draws from a global generator).
 * This build's scheme, shared with the product: only column j is shuffled (same distribution of joint tables), by a
 * forward Fisher-Yates over the positions 0..T-1 with jj = t + mulhi32(r, T - t), r = word (t & 3) of Philox4x32-10
 * (key = seed, counter = (pair, pair >> 32, permutation, 'P' << 24 | t >> 2)); position t carries column i's states in
 * sorted order; "MI of the shuffle >= MI" is decided on sum_xy F[c_xy], F[c] = round(c ln c * 2^40) in int64. */
/* Columns with gaps / unknowns / ambiguity codes (SiteTools::mutualInformation(.., resolveUnknowns = true): a symbol
 * with k compatible states counts 1/k for each, a pair 1/(k_a k_b) for each compatible (x, y)).  A pair with such a
 * column takes this path; fully resolved pairs keep the one above bit for bit.
 *   extended codes: states 0..A-1; code c in [A, min(nmasks, 31)) -> itself with masks[c]; every other code -> 31 =
 *   unknown (all states).  L = lcm of the state counts of all extended codes; the joint table is kept as integers
 *   m_xy = sum_t [x in a_t][y in b_t] (L/k_a)(L/k_b) <= L^2 T, and "MI of the shuffle >= MI" is decided on
 *   sum_xy F[m_xy], F[m] = round(m ln m * 2^sh), sh = min(40, 62 - ceil(log2(M ln M))), M = L^2 T (H1, H2 and
 *   sum_xy m_xy do not change under a shuffle).
 *   isConstant(site, ignoreUnknown = true): at most one distinct code among the symbols that are not unknowns.
 *   positions are taken in the order (ambiguous / unknown codes of column i ascending, then states ascending; stable),
 *   column j is carried along, and the Fisher-Yates of the resolved path shuffles that copy. */
static long orc_lcm(long a, long b) {
  long x = a, y = b;
  while (y) { long r = x % y; x = y; y = r; }
  return a / x * b;
}
typedef struct { int ext[256]; uint32_t mask[32]; int k[32]; long L; int sh; } orc_perm_codes;
static int orc_perm_codes_init(orc_perm_codes* pc, int A, const uint32_t* masks, int nmasks, int T) {
  const uint32_t all = A >= 32 ? 0xffffffffu : ((1u << A) - 1u);
  for (int e = 0; e < 32; e++) { pc->mask[e] = e < A ? (1u << e) : all; pc->k[e] = e < A ? 1 : A; }
  pc->L = A;
  for (int c = 0; c < 256; c++) {
    if (c < A) { pc->ext[c] = c; continue; }
    if (!masks || c >= nmasks) { pc->ext[c] = 31; continue; }
    const uint32_t m = masks[c] & all;
    if (m == 0) return -1;
    if (m == all) { pc->ext[c] = 31; continue; }
    if (c >= 31) return -3;                      /* more partial ambiguity codes than the extended alphabet holds */
    pc->ext[c] = c; pc->mask[c] = m; pc->k[c] = __builtin_popcount(m);
    pc->L = orc_lcm(pc->L, pc->k[c]);
  }
  const double M = (double)pc->L * (double)pc->L * (double)T;
  if (M > 67108864.0) return -3;
  int sh = 62 - (int)ceil(log2(M * log(M)));
  pc->sh = sh > 40 ? 40 : sh;
  return 0;
}
static long long orc_perm_table_sum(const orc_perm_codes* pc, int A, int T, const uint8_t* xs, const uint8_t* q, const long long* F,
                                    long* m) {
  memset(m, 0, sizeof(long) * (size_t)A * A);
  for (int t = 0; t < T; t++) {
    const int a = xs[t], b = q[t];
    const long w = (pc->L / pc->k[a]) * (pc->L / pc->k[b]);
    for (int x = 0; x < A; x++)
      if ((pc->mask[a] >> x) & 1u)
        for (int y = 0; y < A; y++)
          if ((pc->mask[b] >> y) & 1u) m[x * A + y] += w;
  }
  long long s = 0;
  for (int e = 0; e < A * A; e++) s += F[m[e]];
  return s;
}

/* the integer joint table of two columns (test hook: H_joint = ln(L^2 T) - sum m ln m / (L^2 T) must agree with
 * orc_mi_columns); returns L, or a negative status */
long orc_mica_joint_table(int A, const uint32_t* masks, int nmasks, int T, const uint8_t* col_i, const uint8_t* col_j, long* m) {
  orc_perm_codes pc;
  const int st = orc_perm_codes_init(&pc, A, masks, nmasks, T);
  if (st) return st;
  uint8_t* xi = (uint8_t*)malloc((size_t)T);
  uint8_t* xj = (uint8_t*)malloc((size_t)T);
  long long* F = (long long*)calloc((size_t)(pc.L * pc.L * T) + 1, sizeof(long long));
  for (int t = 0; t < T; t++) { xi[t] = (uint8_t)pc.ext[col_i[t]]; xj[t] = (uint8_t)pc.ext[col_j[t]]; }
  (void)orc_perm_table_sum(&pc, A, T, xi, xj, F, m);
  free(xi); free(xj); free(F);
  return pc.L;
}

int orc_mica_permutation_test_masks(const uint8_t* aln, int T, long n, int A, const uint32_t* masks, int nmasks, uint32_t max_perm,
                                    uint64_t seed, long pair_begin, long pair_end, double* pvalue, int32_t* nperm);

int orc_mica_permutation_test(const uint8_t* aln, int T, long n, int A, uint32_t max_perm, uint64_t seed, long pair_begin,
                              long pair_end, double* pvalue, int32_t* nperm) {
  return orc_mica_permutation_test_masks(aln, T, n, A, NULL, 0, max_perm, seed, pair_begin, pair_end, pvalue, nperm);
}

int orc_mica_permutation_test_masks(const uint8_t* aln, int T, long n, int A, const uint32_t* masks, int nmasks, uint32_t max_perm,
                                    uint64_t seed, long pair_begin, long pair_end, double* pvalue, int32_t* nperm) {
  orc_perm_codes pc;
  { const int st = orc_perm_codes_init(&pc, A, masks, nmasks, T); if (st) return st; }
  const uint32_t all = (1u << A) - 1u;
  long long* FG = NULL;                           /* general table, built on first use */
  long* mtab = (long*)malloc(sizeof(long) * (size_t)A * A);
  uint8_t* gxs = (uint8_t*)malloc((size_t)T);
  uint8_t* gbase = (uint8_t*)malloc((size_t)T);

  long long* F = (long long*)calloc((size_t)T + 1, sizeof(long long));
  uint8_t* q = (uint8_t*)malloc((size_t)T);
  uint8_t* xs = (uint8_t*)malloc((size_t)T);
  int* joint = (int*)malloc(sizeof(int) * (size_t)A * A);
  for (int c = 1; c <= T; c++) F[c] = llround((double)c * log((double)c) * 1099511627776.0);
  long p = 0;
  for (long i = 0; i < n - 1; i++)
    for (long j = i + 1; j < n; j++, p++) {
      if (p < pair_begin || p >= pair_end) continue;
      int ci[32] = {0}, cj[32] = {0}, nzi = 0, nzj = 0, general = 0;
      for (int t = 0; t < T; t++) {
        const int ei = pc.ext[aln[(size_t)t * n + i]], ej = pc.ext[aln[(size_t)t * n + j]];
        general |= (ei >= A) | (ej >= A);
        ci[ei]++;
        cj[ej]++;
      }
      if (general) {
        for (int e = 0; e < 32; e++) { nzi += ci[e] > 0 && pc.mask[e] != all; nzj += cj[e] > 0 && pc.mask[e] != all; }
        if (nzi <= 1 || nzj <= 1) { pvalue[p - pair_begin] = 1.0; nperm[p - pair_begin] = 0; continue; }
        if (!FG) {
          const long M = pc.L * pc.L * (long)T;
          FG = (long long*)calloc((size_t)M + 1, sizeof(long long));
          for (long m = 1; m <= M; m++) FG[m] = llround((double)m * log((double)m) * ldexp(1.0, pc.sh));
        }
        { /* stable sort of the positions: ambiguous / unknown codes of column i ascending, then states ascending */
          int tt = 0;
          for (int pass = 0; pass < 2; pass++)
            for (int e = pass ? 0 : A; e < (pass ? A : 32); e++)
              for (int t = 0; t < T; t++)
                if (pc.ext[aln[(size_t)t * n + i]] == e) { gxs[tt] = (uint8_t)e; gbase[tt] = (uint8_t)pc.ext[aln[(size_t)t * n + j]]; tt++; }
        }
        const long long sobs = orc_perm_table_sum(&pc, A, T, gxs, gbase, FG, mtab);
        uint32_t k = 0, count = 0;
        for (; count < 5 && k < max_perm; k++) {
          memcpy(q, gbase, (size_t)T);
          uint32_t r[4] = {0, 0, 0, 0};
          for (int t = 0; t < T; t++) {
            if ((t & 3) == 0)
              philox4x32_10((uint32_t)p, (uint32_t)((uint64_t)p >> 32), k, 0x50000000u | (uint32_t)(t >> 2), (uint32_t)seed,
                            (uint32_t)(seed >> 32), r);
            const int jj = t + (int)(((uint64_t)r[t & 3] * (uint32_t)(T - t)) >> 32);
            const uint8_t vj = q[jj];
            q[jj] = q[t];
            q[t] = vj;
          }
          if (orc_perm_table_sum(&pc, A, T, gxs, q, FG, mtab) >= sobs) count++;
        }
        pvalue[p - pair_begin] = (double)(count + 1) / (double)(k + 1);
        nperm[p - pair_begin] = (int32_t)k;
        continue;
      }
      for (int x = 0; x < A; x++) { nzi += ci[x] > 0; nzj += cj[x] > 0; }
      if (nzi <= 1 || nzj <= 1) { pvalue[p - pair_begin] = 1.0; nperm[p - pair_begin] = 0; continue; }
      memset(joint, 0, sizeof(int) * (size_t)A * A);
      for (int t = 0; t < T; t++) joint[aln[(size_t)t * n + i] * A + aln[(size_t)t * n + j]]++;
      long long sobs = 0;
      for (int e = 0; e < A * A; e++) sobs += F[joint[e]];
      { int t = 0; for (int x = 0; x < A; x++) for (int c = 0; c < ci[x]; c++) xs[t++] = (uint8_t)x; }
      uint32_t k = 0, count = 0;
      for (; count < 5 && k < max_perm; k++) {
        for (int t = 0; t < T; t++) q[t] = aln[(size_t)t * n + j];
        memset(joint, 0, sizeof(int) * (size_t)A * A);
        uint32_t r[4] = {0, 0, 0, 0};
        for (int t = 0; t < T; t++) {
          if ((t & 3) == 0)
            philox4x32_10((uint32_t)p, (uint32_t)((uint64_t)p >> 32), k, 0x50000000u | (uint32_t)(t >> 2), (uint32_t)seed,
                          (uint32_t)(seed >> 32), r);
          const int jj = t + (int)(((uint64_t)r[t & 3] * (uint32_t)(T - t)) >> 32);
          const uint8_t vj = q[jj];
          q[jj] = q[t];
          q[t] = vj;
          joint[xs[t] * A + vj]++;
        }
        long long s = 0;
        for (int e = 0; e < A * A; e++) s += F[joint[e]];
        if (s >= sobs) count++;
      }
      pvalue[p - pair_begin] = (double)(count + 1) / (double)(k + 1);
      nperm[p - pair_begin] = (int32_t)k;
    }
  free(F); free(q); free(xs); free(joint); free(FG); free(mtab); free(gxs); free(gbase);
  return 0;
}
