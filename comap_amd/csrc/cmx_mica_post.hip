// What Mica does with the all-pairs MI matrix once it exists (CoMap/Mica.cpp:346-363, 549-607): per-site average MI
// (the APC / RCW corrections of :656-657 are products of these), and the "z-score" null that turns every pair of the
// data set itself into one draw of the null distribution.  Bandwidth-bound passes over the n x n matrix.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "cmx_device.h"

namespace cmx {

namespace {

__device__ __forceinline__ double block_sum_256(double v, double* sm) {
  for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  const double t = sm[0] + sm[1] + sm[2] + sm[3];
  __syncthreads();
  return t;
}

// averageMI[i] = sum_{j != i} MI(i, j) / (n - 1)   (Mica.cpp:349-361); one workgroup per site.
// Element (i, j) is read from the upper triangle (row min(i,j), column max(i,j)): the matrix is symmetric and the
// intra entry points of the engine only guarantee that triangle.
__global__ __launch_bounds__(256) void mica_average_kernel(const double* __restrict__ mi, size_t n, size_t ld,
                                                           double* __restrict__ avg) {
  __shared__ double sm[4];
  const size_t i = blockIdx.x;
  double s = 0.0;
  for (size_t j = threadIdx.x; j < n; j += 256)
    if (j != i) s += j > i ? mi[i * ld + j] : mi[j * ld + i];
  s = block_sum_256(s, sm);
  if (threadIdx.x == 0) avg[i] = s / (double)(n - 1);
}

// fullAverageMI = mean(averageMI)   (Mica.cpp:363); one workgroup, fixed order
__global__ __launch_bounds__(256) void mica_full_average_kernel(const double* __restrict__ avg, size_t n, double* __restrict__ full) {
  __shared__ double sm[4];
  double s = 0.0;
  for (size_t j = threadIdx.x; j < n; j += 256) s += avg[j];
  s = block_sum_256(s, sm);
  if (threadIdx.x == 0) *full = s / (double)n;
}

// z-score null (Mica.cpp:565-603): pair (i, j), j > i, in row-major order -> (statistic, min key); which: 0 MI,
// 1 MIp = MI - avg_i avg_j / full, 2 MIc = MI / (avg_i avg_j / 2)
__global__ __launch_bounds__(256) void mica_zscore_kernel(int which, const double* __restrict__ mi, size_t n, size_t ld,
                                                          const double* __restrict__ avg, const double* __restrict__ full,
                                                          const double* __restrict__ key, double* __restrict__ out_stat,
                                                          double* __restrict__ out_key) {
  const size_t i = blockIdx.y;
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j <= i || j >= n) return;
  const size_t p = i * n - i * (i + 1) / 2 + (j - i - 1);
  double v = mi[i * ld + j];
  if (which == 1) v -= avg[i] * avg[j] / *full;
  else if (which == 2) v /= avg[i] * avg[j] / 2.;
  out_stat[p] = v;
  out_key[p] = fmin(key[i], key[j]);
}


// ------------------------------------------------------------------------------------------------ permutation test
// miTest (CoMap/Mica.cpp:93-118): MI of the pair against MI after shuffling the columns, until 5 shuffles reach the
// observed value or max_perm were done.  What the device does with it:
//  * shuffling both columns or one is the same distribution of joint tables; H1 and H2 do not change under a shuffle,
//    so "rep >= mi" is "sum_xy f(c_xy) of the shuffle >= the observed one", f(c) = c ln c.  That sum is kept in fixed
//    point (F[c] = round(c ln c * 2^40), int64), so it is exact and order-free: tables with the same counts tie
//    exactly, on any device and in the CPU restatement (the reference's own floating-point sums decide such ties by
//    rounding noise);
//  * one wave per pair, one LANE per permutation: 64 shuffles of the same pair run side by side, each lane with its
//    own copy of column j in LDS (forward Fisher-Yates from the counter RNG: Philox4x32-10, key = seed, counter =
//    (pair, pair >> 32, permutation, 'P' << 24 | position / 4)) and a 20-counter histogram for the current state of
//    column i -- positions are taken in the order of column i's states, so only the counts of the current state are
//    live, and the sum is accumulated from the increments (F[c+1] - F[c]), no pass over the A x A table;
//  * the sequential stopping rule is applied to the 64 results in order (ballot + popcount).
// Only fully resolved columns (no gaps / ambiguity codes); T <= kPermMaxTaxa.
constexpr int kPermMaxTaxa = 2047;

__device__ __forceinline__ void philox4(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// per column: counts of the resolved states (fast path), histogram of the extended codes (states, partial ambiguity
// codes, 31 = unknown; emap: alignment code -> extended code), whether the column has any code that is not a state, and
// (global maximum) how many distinct such codes one column holds
__global__ __launch_bounds__(256) void mica_colcount_kernel(const uint8_t* __restrict__ aln, int T, size_t n, size_t ld, int A,
                                                            const uint8_t* __restrict__ emap, uint16_t* __restrict__ cnt /*[n][A]*/,
                                                            uint16_t* __restrict__ ext /*[n][32]*/, uint8_t* __restrict__ hasamb,
                                                            int* __restrict__ bad /*[2]: any, max distinct*/) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint16_t c[32];
  for (int x = 0; x < 32; ++x) c[x] = 0;
  for (int t = 0; t < T; ++t) {
    const int v = emap[aln[(size_t)t * ld + i]];
    for (int x = 0; x < 32; ++x) c[x] += (v == x) ? 1 : 0;
  }
  int namb = 0;
  for (int x = 0; x < 32; ++x) {
    ext[i * 32 + x] = c[x];
    if (x < A) cnt[i * A + x] = c[x];
    else namb += c[x] > 0;
  }
  hasamb[i] = namb > 0;
  if (namb > 0) { bad[0] = 1; atomicMax(&bad[1], namb); }
}

// positions of a column in the order the general permutation kernel visits them: codes that are not states first
// (ascending), then the states (ascending); stable.  One thread per column.
__global__ __launch_bounds__(256) void mica_colorder_kernel(const uint8_t* __restrict__ aln, int T, size_t n, size_t ld, int A,
                                                            const uint8_t* __restrict__ emap, const uint16_t* __restrict__ ext,
                                                            uint16_t* __restrict__ order /*[n][T]*/) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint16_t start[32];
  int run = 0;
  for (int e = A; e < 32; ++e) { start[e] = (uint16_t)run; run += ext[i * 32 + e]; }
  for (int e = 0; e < A; ++e) { start[e] = (uint16_t)run; run += ext[i * 32 + e]; }
  for (int t = 0; t < T; ++t) {
    const int v = emap[aln[(size_t)t * ld + i]];
    int pos = 0;
    for (int e = 0; e < 32; ++e)
      if (e == v) { pos = start[e]; start[e] = (uint16_t)(pos + 1); }
    order[i * (size_t)T + pos] = (uint16_t)t;
  }
}

struct PermArgs {
  const uint8_t* aln; int T; size_t n, ld; int A;
  const uint16_t* colcnt;
  const uint8_t* hasamb;     // [n]: the column has gaps / unknowns / ambiguity codes (its pairs belong to the general kernel)
  const long long* dF;       // [T]: F[c+1] - F[c]
  uint32_t max_perm; uint64_t seed;
  size_t pair_begin, pair_end;   // pairs in row-major (i < j) order
  double* pvalue; int32_t* nperm;   // indexed by pair - pair_begin
  int qstride, cstride;      // bytes per lane of the private column copy / counters
  int wave_bytes;
  uint32_t first;            // shuffles already done by the opening pass (0 if it was skipped)
};

// G = column pairs per wave.  G = 4 is the opening pass: four pairs, 16 shuffles each -- half of all pairs between
// unrelated columns collect their 5 hits within the first 16 shuffles, and a whole wave per pair would run 64.  A pair
// that is not decided yet leaves nperm = -1 - hits; the G = 1 pass (one pair per wave, 64 shuffles per round) picks
// those up at shuffle 16.  Shuffle k of a pair is the same in either pass: the result does not depend on the split.
constexpr int kPermFirst = 16;
template <int G>
__global__ void mica_perm_kernel(PermArgs a) {
  constexpr int LPG = 64 / G;                                                     // lanes (= shuffles per round) per pair
  extern __shared__ __attribute__((aligned(16))) uint8_t perm_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int grp = lane / LPG, gl = lane % LPG;
  const int T = a.T, A = a.A, Tr = (T + 3) & ~3;
  long long* dF = reinterpret_cast<long long*>(perm_smem);                       // [T] block-shared
  uint8_t* wbase = perm_smem + (size_t)T * 8 + (size_t)wave * a.wave_bytes;
  const int gbytes = A * A * 4 + 64 + Tr;                                         // per pair: joint | seg | column j
  uint32_t* joint = reinterpret_cast<uint32_t*>(wbase + (size_t)grp * gbytes);   // [A*A] observed table
  uint16_t* seg = reinterpret_cast<uint16_t*>(wbase + (size_t)grp * gbytes + A * A * 4);   // [A] end of state x in column i's order
  uint8_t* qbase = wbase + (size_t)grp * gbytes + A * A * 4 + 64;                // [T] column j
  uint8_t* priv = wbase + (size_t)G * gbytes;
  uint8_t* q = priv + (size_t)lane * a.qstride;                                  // private copy
  uint16_t* cnt = reinterpret_cast<uint16_t*>(priv + (size_t)64 * a.qstride + (size_t)lane * a.cstride);
  for (int c = threadIdx.x; c < T; c += blockDim.x) dF[c] = a.dF[c];
  __syncthreads();
  const size_t n = a.n;
  for (size_t pb = a.pair_begin + ((size_t)blockIdx.x * nwaves + wave) * G; pb < a.pair_end; pb += (size_t)gridDim.x * nwaves * G) {
    const bool exists = pb + grp < a.pair_end;
    const size_t p = exists ? pb + grp : a.pair_end - 1;
    uint32_t done = 0, count = 0;
    bool open = exists;
    if (G == 1) {                      // second pass: only what the opening pass left undecided
      const int st = a.nperm[p - a.pair_begin];
      if (st >= 0) continue;
      count = (uint32_t)(-1 - st);
      done = a.first;
    }
    // (i, j) of pair p in row-major order: p = i*n - i(i+1)/2 + (j - i - 1)
    size_t i = (size_t)((2.0 * n - 1.0 - sqrt((2.0 * n - 1.0) * (2.0 * n - 1.0) - 8.0 * (double)p)) / 2.0);
    while (i > 0 && i * n - i * (i + 1) / 2 > p) --i;
    while ((i + 1) * n - (i + 1) * (i + 2) / 2 <= p) ++i;
    const size_t j = p - (i * n - i * (i + 1) / 2) + i + 1;
    // a pair with a column that has gaps / unknowns / ambiguity codes was decided by mica_perm_general_kernel: nothing
    // is written for it here (its symbols are clamped below only to keep the table indices in range)
    if (a.hasamb[i] | a.hasamb[j]) {
      open = false;
      if (G == 1) continue;
    }
    // column setup (cooperative within the pair's lanes)
    for (int e = gl; e < A * A; e += LPG) joint[e] = 0;
    int nz_i = 0, nz_j = 0;
    if (gl == 0) {
      int run = 0;
      for (int x = 0; x < A; ++x) {
        const int ci = a.colcnt[i * A + x], cj = a.colcnt[j * A + x];
        nz_i += ci > 0; nz_j += cj > 0;
        run += ci;
        seg[x] = (uint16_t)run;
      }
      if (run < T) seg[A - 1] = (uint16_t)T;   // column with unknowns (pair skipped, see above): keep the walk in range
    }
    nz_i = __shfl(nz_i, grp * LPG); nz_j = __shfl(nz_j, grp * LPG);
    for (int t = gl; t < T; t += LPG) qbase[t] = (uint8_t)min((int)a.aln[(size_t)t * a.ld + j], A - 1);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (nz_i <= 1 || nz_j <= 1) {     // SiteTools::isConstant(site, ignoreUnknown = true), Mica.cpp:100-104
      if (open && gl == 0) { a.pvalue[p - a.pair_begin] = 1.0; a.nperm[p - a.pair_begin] = 0; }
      open = false;
      if (G == 1) continue;
    }
    long long sobs = 0;
    for (int t = gl; t < T; t += LPG) {
      const int x = min((int)a.aln[(size_t)t * a.ld + i], A - 1), y = qbase[t];
      const uint32_t old = atomicAdd(&joint[x * A + y], 1u);
      sobs += dF[old];
    }
    for (int off = LPG / 2; off; off >>= 1) sobs += __shfl_xor(sobs, off);
    bool stop = false;
    do {
      const uint32_t k = done + (uint32_t)gl;
      for (int t = 0; t < T; t += 4) *reinterpret_cast<uint32_t*>(q + t) = *reinterpret_cast<const uint32_t*>(qbase + t);
      long long s = 0;
      int x = -1, send = 0;
      uint32_t r[4] = {0, 0, 0, 0};
      for (int t = 0; t < T; ++t) {
        while (t >= send) {            // next state of column i that occurs (uniform within the pair's lanes)
          ++x;
          send = seg[x];
          for (int y = 0; y < A; y += 2) *reinterpret_cast<uint32_t*>(cnt + y) = 0;
        }
        if ((t & 3) == 0) philox4(a.seed, (uint32_t)p, (uint32_t)((uint64_t)p >> 32), k, 0x50000000u | (uint32_t)(t >> 2), r);
        const int jj = t + (int)__umulhi(r[t & 3], (uint32_t)(T - t));
        const uint8_t vj = q[jj], vt = q[t];
        q[jj] = vt;                    // q[t] itself is not read again
        const int c = cnt[vj];
        s += dF[c];
        cnt[vj] = (uint16_t)(c + 1);
      }
      const bool hit = k < a.max_perm && s >= sobs;
      const unsigned long long mw = __ballot(hit);
      const unsigned long long m = G == 1 ? mw : (mw >> (grp * LPG)) & ((1ull << LPG) - 1ull);
      const uint32_t avail = min((uint32_t)LPG, a.max_perm - done);
      const int need = 5 - (int)count;
      if (__popcll(m) >= need) {       // the need-th hit ends the loop: find its position
        unsigned long long mm = m;
        for (int z = 1; z < need; ++z) mm &= mm - 1;
        const int pos = __ffsll((long long)mm) - 1;
        done += (uint32_t)pos + 1;
        count = 5;
        stop = true;
      } else {
        count += (uint32_t)__popcll(m);
        done += avail;
      }
    } while (G == 1 && !stop && done < a.max_perm);
    if (open && gl == 0) {
      if (stop || done >= a.max_perm) {
        a.pvalue[p - a.pair_begin] = (double)(count + 1) / (double)(done + 1);
        a.nperm[p - a.pair_begin] = (int32_t)done;
      } else {
        a.nperm[p - a.pair_begin] = -1 - (int32_t)count;   // undecided after kPermFirst shuffles
      }
    }
  }
}


// ---- pairs with gaps / unknowns / ambiguity codes (SiteTools::mutualInformation(.., resolveUnknowns = true): a symbol
// with k compatible states counts 1/k for each of them).  Same test, same shuffles, but the joint table is the integer
// m_xy = sum_t [x in a_t][y in b_t] (L/k_a)(L/k_b), L = lcm of the k's, and the sum kept in fixed point is
// sum_xy F[m_xy], F[m] = round(m ln m * 2^sh) from a table in global memory (L^2 T + 1 entries).  One wave per pair, one
// lane per shuffle.  Positions are visited in the order (codes of column i that are not states, ascending; then the
// states), so that a lane only keeps: the counts of column j's codes against the current code of column i (32
// counters), and for each non-state code a of column i the row R_a[y] = sum_b [y in b] (L/k_b) c_ab (A words each).
// Row x of the table is then L R_x + sum_{a containing x} (L/k_a) R_a, summed as soon as state x is done.
// The observed value is the same evaluation without a shuffle.  oracle/oracle.c orc_mica_permutation_test_masks builds
// the whole table per shuffle instead; the tests compare p-values and permutation counts exactly.
struct PermGenArgs {
  const uint8_t* aln; int T; size_t n, ld;
  const uint8_t* emap;       // [256] alignment code -> extended code
  const uint16_t* ext;       // [n][32]
  const uint16_t* order;     // [n][T]
  const uint8_t* hasamb;     // [n]
  const uint32_t* emask;     // [32]
  const uint32_t* ewgt;      // [32] L / k
  const long long* F;        // [L*L*T + 1]
  uint32_t L, max_perm; uint64_t seed;
  size_t pair_begin, pair_end;
  double* pvalue; int32_t* nperm;
  int qstride, astride, wave_bytes;   // bytes per lane of the private column copy / of the stored rows; per wave
};
constexpr int kPermCntStride = 68;     // 32 counters of 16 bits + one word of padding (LDS banks)

template <int A, bool SHUF>
__device__ __forceinline__ long long perm_general_eval(const PermGenArgs& a, const uint8_t* qbase, uint8_t* q, uint16_t* cnt,
                                                       uint32_t* amb, const uint8_t* ord, const uint16_t* send, const uint8_t* jb,
                                                       int ne, int namb, int njb, uint32_t cover, size_t p, uint32_t k) {
  const int T = a.T;
  for (int t = 0; t < T; t += 4) *reinterpret_cast<uint32_t*>(q + t) = *reinterpret_cast<const uint32_t*>(qbase + t);
  for (int y = 0; y < 32; y += 2) *reinterpret_cast<uint32_t*>(cnt + y) = 0;
  long long s = 0;
  uint32_t r[4] = {0, 0, 0, 0};
  int t = 0;
  for (int ei = 0; ei < ne; ++ei) {
    const int e = ord[ei], end = send[ei];
    const bool touched = t < end;
    for (; t < end; ++t) {
      uint8_t vj;
      if (SHUF) {
        if ((t & 3) == 0) philox4(a.seed, (uint32_t)p, (uint32_t)((uint64_t)p >> 32), k, 0x50000000u | (uint32_t)(t >> 2), r);
        const int jj = t + (int)__umulhi(r[t & 3], (uint32_t)(T - t));
        vj = q[jj];
        q[jj] = q[t];
      } else {
        vj = q[t];
      }
      cnt[vj] = (uint16_t)(cnt[vj] + 1);
    }
    const bool isamb = ei < namb;
    if (!isamb && !touched && !((cover >> e) & 1u)) continue;   // empty row that no code of column i reaches: all zero
    uint32_t R[A];
#pragma unroll
    for (int y = 0; y < A; ++y) R[y] = a.L * (uint32_t)cnt[y];
    for (int bi = 0; bi < njb; ++bi) {
      const int b = jb[bi];
      const uint32_t w = a.ewgt[b] * (uint32_t)cnt[b], mk = a.emask[b];
#pragma unroll
      for (int y = 0; y < A; ++y) R[y] += ((mk >> y) & 1u) ? w : 0u;
    }
    if (isamb) {
#pragma unroll
      for (int y = 0; y < A; ++y) amb[ei * A + y] = R[y];
    } else {
#pragma unroll
      for (int y = 0; y < A; ++y) R[y] *= a.L;
      for (int qi = 0; qi < namb; ++qi) {
        const int eq = ord[qi];
        if ((a.emask[eq] >> e) & 1u) {
          const uint32_t wq = a.ewgt[eq];
#pragma unroll
          for (int y = 0; y < A; ++y) R[y] += wq * amb[qi * A + y];
        }
      }
#pragma unroll
      for (int y = 0; y < A; ++y) s += a.F[R[y]];
    }
    if (touched)
      for (int y = 0; y < 32; y += 2) *reinterpret_cast<uint32_t*>(cnt + y) = 0;
  }
  return s;
}

template <int A>
__global__ void mica_perm_general_kernel(PermGenArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t perm_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int T = a.T, Tr = (T + 3) & ~3;
  uint8_t* wbase = perm_smem + (size_t)wave * a.wave_bytes;
  uint8_t* qbase = wbase;                                            // [Tr] column j (extended codes) in visiting order
  uint16_t* send = reinterpret_cast<uint16_t*>(wbase + Tr);          // [64] end of each entry's positions
  uint8_t* ord = wbase + Tr + 128;                                   // [64] extended code of each entry
  uint8_t* jb = ord + 64;                                            // [32] non-state codes present in column j
  int* hdr = reinterpret_cast<int*>(jb + 32);                        // ne, namb, njb, constant, cover
  uint8_t* priv = wbase + Tr + 128 + 64 + 32 + 32;
  uint8_t* q = priv + (size_t)lane * a.qstride;
  uint16_t* cnt = reinterpret_cast<uint16_t*>(priv + (size_t)64 * a.qstride + (size_t)lane * kPermCntStride);
  uint32_t* amb = reinterpret_cast<uint32_t*>(priv + (size_t)64 * a.qstride + (size_t)64 * kPermCntStride + (size_t)lane * a.astride);
  const uint32_t all = (1u << A) - 1u;
  const size_t n = a.n;
  for (size_t p = a.pair_begin + (size_t)blockIdx.x * nwaves + wave; p < a.pair_end; p += (size_t)gridDim.x * nwaves) {
    size_t i = (size_t)((2.0 * n - 1.0 - sqrt((2.0 * n - 1.0) * (2.0 * n - 1.0) - 8.0 * (double)p)) / 2.0);
    while (i > 0 && i * n - i * (i + 1) / 2 > p) --i;
    while ((i + 1) * n - (i + 1) * (i + 2) / 2 <= p) ++i;
    const size_t j = p - (i * n - i * (i + 1) / 2) + i + 1;
    if (!(a.hasamb[i] | a.hasamb[j])) continue;      // fully resolved pair: mica_perm_kernel
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      int ne = 0, run = 0, njb = 0, nzi = 0, nzj = 0;
      uint32_t cover = 0;
      for (int e = A; e < 32; ++e) {
        const int ci = a.ext[i * 32 + e], cj = a.ext[j * 32 + e];
        if (ci > 0) { ord[ne] = (uint8_t)e; run += ci; send[ne] = (uint16_t)run; ++ne; cover |= a.emask[e]; nzi += a.emask[e] != all; }
        if (cj > 0) { jb[njb++] = (uint8_t)e; nzj += a.emask[e] != all; }
      }
      const int namb = ne;
      for (int x = 0; x < A; ++x) {
        const int ci = a.ext[i * 32 + x], cj = a.ext[j * 32 + x];
        ord[ne] = (uint8_t)x; run += ci; send[ne] = (uint16_t)run; ++ne;
        nzi += ci > 0; nzj += cj > 0;
      }
      hdr[0] = ne; hdr[1] = namb; hdr[2] = njb; hdr[3] = (nzi <= 1 || nzj <= 1) ? 1 : 0; hdr[4] = (int)cover;
    }
    for (int t = lane; t < T; t += 64) qbase[t] = a.emap[a.aln[(size_t)a.order[i * (size_t)T + t] * a.ld + j]];
    for (int t = T + lane; t < Tr; t += 64) qbase[t] = 0;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const int ne = hdr[0], namb = hdr[1], njb = hdr[2];
    const uint32_t cover = (uint32_t)hdr[4];
    if (hdr[3]) {                      // SiteTools::isConstant(site, ignoreUnknown = true), Mica.cpp:100-104
      if (lane == 0) { a.pvalue[p - a.pair_begin] = 1.0; a.nperm[p - a.pair_begin] = 0; }
      continue;
    }
    const long long sobs = perm_general_eval<A, false>(a, qbase, q, cnt, amb, ord, send, jb, ne, namb, njb, cover, p, 0);
    uint32_t done = 0, count = 0;
    bool stop = false;
    do {
      const uint32_t k = done + (uint32_t)lane;
      const long long s = perm_general_eval<A, true>(a, qbase, q, cnt, amb, ord, send, jb, ne, namb, njb, cover, p, k);
      const bool hit = k < a.max_perm && s >= sobs;
      const unsigned long long m = __ballot(hit);
      const uint32_t avail = min(64u, a.max_perm - done);
      const int need = 5 - (int)count;
      if (__popcll(m) >= need) {       // the need-th hit ends the loop: find its position
        unsigned long long mm = m;
        for (int z = 1; z < need; ++z) mm &= mm - 1;
        done += (uint32_t)(__ffsll((long long)mm) - 1) + 1;
        count = 5;
        stop = true;
      } else {
        count += (uint32_t)__popcll(m);
        done += avail;
      }
    } while (!stop && done < a.max_perm);
    if (lane == 0) {
      a.pvalue[p - a.pair_begin] = (double)(count + 1) / (double)(done + 1);
      a.nperm[p - a.pair_begin] = (int32_t)done;
    }
  }
}

}  // namespace

hipError_t launch_mica_average(const double* d_mi, size_t n, size_t ld, double* d_avg, double* d_full, hipStream_t stream) {
  hipLaunchKernelGGL(mica_average_kernel, dim3((unsigned)n), dim3(256), 0, stream, d_mi, n, ld, d_avg);
  hipLaunchKernelGGL(mica_full_average_kernel, dim3(1), dim3(256), 0, stream, d_avg, n, d_full);
  return hipGetLastError();
}

hipError_t launch_mica_zscore(int which, const double* d_mi, size_t n, size_t ld, const double* d_avg, const double* d_full,
                              const double* d_key, double* d_stat, double* d_outkey, hipStream_t stream) {
  hipLaunchKernelGGL(mica_zscore_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, stream, which, d_mi, n,
                     ld, d_avg, d_full, d_key, d_stat, d_outkey);
  return hipGetLastError();
}

}  // namespace cmx

namespace cmx {
int mica_perm_max_taxa() { return kPermMaxTaxa; }

hipError_t launch_mica_colcount(const uint8_t* d_aln, int T, size_t n, size_t ld, int A, const uint8_t* d_emap, uint16_t* d_cnt,
                                uint16_t* d_ext, uint8_t* d_hasamb, int* d_bad, hipStream_t stream) {
  hipLaunchKernelGGL(mica_colcount_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_aln, T, n, ld, A, d_emap, d_cnt,
                     d_ext, d_hasamb, d_bad);
  return hipGetLastError();
}

hipError_t launch_mica_colorder(const uint8_t* d_aln, int T, size_t n, size_t ld, int A, const uint8_t* d_emap, const uint16_t* d_ext,
                                uint16_t* d_order, hipStream_t stream) {
  hipLaunchKernelGGL(mica_colorder_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_aln, T, n, ld, A, d_emap, d_ext,
                     d_order);
  return hipGetLastError();
}

// LDS one wave of the general kernel needs with `namb` stored rows per lane (0 = does not fit one workgroup)
size_t mica_perm_general_lds(int T, int A, int namb) {
  int sd = (T + 3) / 4; if (sd % 2 == 0) ++sd;
  int ad = std::max(1, namb * A); if (ad % 2 == 0) ++ad;
  const size_t w = ((size_t)((T + 3) & ~3) + 128 + 64 + 32 + 32 + 64 * (size_t)(4 * sd) + 64 * (size_t)kPermCntStride + 64 * (size_t)(4 * ad) + 15) & ~(size_t)15;
  return w <= 160 * 1024 ? w : 0;
}

hipError_t launch_mica_perm_general(const uint8_t* d_aln, int T, size_t n, size_t ld, int A, const uint8_t* d_emap,
                                    const uint16_t* d_ext, const uint16_t* d_order, const uint8_t* d_hasamb, const uint32_t* d_emask,
                                    const uint32_t* d_ewgt, const long long* d_F, uint32_t L, int namb, uint32_t max_perm, uint64_t seed,
                                    size_t pair_begin, size_t pair_end, double* d_pvalue, int32_t* d_nperm, int cu_count,
                                    hipStream_t stream) {
  PermGenArgs a{};
  a.aln = d_aln; a.T = T; a.n = n; a.ld = ld; a.emap = d_emap; a.ext = d_ext; a.order = d_order; a.hasamb = d_hasamb;
  a.emask = d_emask; a.ewgt = d_ewgt; a.F = d_F; a.L = L; a.max_perm = max_perm; a.seed = seed;
  a.pair_begin = pair_begin; a.pair_end = pair_end; a.pvalue = d_pvalue; a.nperm = d_nperm;
  int sd = (T + 3) / 4; if (sd % 2 == 0) ++sd;
  a.qstride = 4 * sd;
  int ad = std::max(1, namb * A); if (ad % 2 == 0) ++ad;
  a.astride = 4 * ad;
  const size_t wb = mica_perm_general_lds(T, A, namb);
  if (wb == 0) return hipErrorInvalidValue;
  a.wave_bytes = (int)wb;
  const int waves = (int)std::max<size_t>(1, std::min<size_t>(4, (150 * 1024) / wb));
  const size_t lds = (size_t)waves * wb;
  const size_t npairs = pair_end - pair_begin;
  const size_t per_cu = std::max<size_t>(1, (160 * 1024) / lds);
  const unsigned grid = (unsigned)std::min<size_t>((npairs + waves - 1) / waves, (size_t)cu_count * per_cu);
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), lds, stream, a);
    return hipGetLastError();
  };
  return A == 4 ? go(mica_perm_general_kernel<4>) : go(mica_perm_general_kernel<20>);
}

bool mica_perm_opening_fits(int T, int A) {
  int sd = (T + 3) / 4; if (sd % 2 == 0) ++sd;
  int cd = (A * 2 + 3) / 4; if (cd % 2 == 0) ++cd;
  const size_t lds4 = (size_t)T * 8 + (((size_t)4 * (A * A * 4 + 64 + ((T + 3) & ~3)) + 64 * (size_t)(4 * sd) + 64 * (size_t)(4 * cd) + 15) & ~(size_t)15);
  return lds4 <= 160 * 1024;
}

hipError_t launch_mica_perm(const uint8_t* d_aln, int T, size_t n, size_t ld, int A, const uint16_t* d_colcnt,
                            const uint8_t* d_hasamb, const long long* d_dF, bool nperm_preset, uint32_t max_perm, uint64_t seed, size_t pair_begin, size_t pair_end, double* d_pvalue, int32_t* d_nperm,
                            int cu_count, hipStream_t stream) {
  PermArgs a{};
  a.aln = d_aln; a.T = T; a.n = n; a.ld = ld; a.A = A; a.colcnt = d_colcnt; a.hasamb = d_hasamb; a.dF = d_dF; a.max_perm = max_perm; a.seed = seed;
  a.pair_begin = pair_begin; a.pair_end = pair_end; a.pvalue = d_pvalue; a.nperm = d_nperm;
  int sd = (T + 3) / 4; if (sd % 2 == 0) ++sd;
  a.qstride = 4 * sd;
  int cd = (A * 2 + 3) / 4; if (cd % 2 == 0) ++cd;
  a.cstride = 4 * cd;
  const size_t npairs = pair_end - pair_begin;
  auto go = [&](auto kern, int G) -> hipError_t {
    a.wave_bytes = (G * (A * A * 4 + 64 + ((T + 3) & ~3)) + 64 * a.qstride + 64 * a.cstride + 15) & ~15;
    const int budget = 150 * 1024 - T * 8;
    const int waves = std::max(1, std::min(4, budget / a.wave_bytes));
    const size_t lds = (size_t)T * 8 + (size_t)waves * a.wave_bytes;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const size_t per_cu = std::max<size_t>(1, (160 * 1024) / lds);
    const size_t units = (npairs + G - 1) / G;
    const unsigned grid = (unsigned)std::min<size_t>((units + waves - 1) / waves, (size_t)cu_count * per_cu);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), lds, stream, a);
    return hipGetLastError();
  };
  // opening pass (four pairs per wave, 16 shuffles each) if its four column copies fit the LDS, then one pair per wave
  // for what is still undecided
  hipError_t e;
  if (mica_perm_opening_fits(T, A)) {
    a.first = kPermFirst;
    if ((e = go(mica_perm_kernel<4>, 4)) != hipSuccess) return e;
  } else {
    a.first = 0;
    // -1: undecided, no hits.  nperm_preset: the caller did this before the general kernel wrote its pairs' results
    if (!nperm_preset && (e = hipMemsetAsync(d_nperm, 0xFF, sizeof(int32_t) * npairs, stream)) != hipSuccess) return e;
  }
  return go(mica_perm_kernel<1>, 1);
}
}  // namespace cmx
