// Host side of the nucleotide mapping path (cmx_nuc.h): cuts the tree into register-resident blocks, writes the visit
// records both phases of map_nuc_kernel read, and checks the whole program numerically against a direct pruning
// computation before a context accepts it.  Plain C++17, no device code.
#include "cmx_nuc.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>

namespace cmx {
namespace {

struct BinTree {
  int nn = 0, nd = 0, root = 0;
  std::vector<std::array<int, 2>> ch;   // children of the binary device tree; nodes >= nn are pseudo nodes
  std::vector<int> taxon_of;
  bool leaf(int n) const { return n < nn && taxon_of[n] >= 0; }
  bool pseudo(int n) const { return n >= nn; }
};

// every multifurcation (the trifurcating root of an unrooted tree included) becomes a chain of pseudo nodes on zero-length
// branches: ((..((c1, c2), c3) ..), ck) -- the same split the 20-state walk uses (cmx_host_model.cpp: build_records)
BinTree binarise(const HostModel& hm) {
  BinTree t;
  t.nn = hm.nn;
  t.root = hm.root;
  t.taxon_of = hm.taxon_of;
  t.ch.assign(hm.nn, {-1, -1});
  for (int n = 0; n < hm.nn; ++n) {
    if (hm.taxon_of[n] >= 0) continue;
    std::vector<int> c;
    for (int e = hm.first_child[n]; e >= 0; e = hm.next_sib[e]) c.push_back(e);
    int left = c[0];
    for (size_t i = 1; i + 1 < c.size(); ++i) {
      t.ch.push_back({left, c[i]});
      left = (int)t.ch.size() - 1;
    }
    t.ch[n] = {left, c.back()};
  }
  t.nd = (int)t.ch.size();
  return t;
}

}  // namespace

std::string build_nuc_program(const HostModel& hm, int NB, NucProgram* np) {
  if (hm.S != 4) return "nucleotide program: 4 states required";
  if (NB < 2 || NB > 16) return "nucleotide program: block capacity out of range";
  const BinTree t = binarise(hm);
  const int K = hm.K, C = hm.C, B = hm.B;
  *np = NucProgram();
  np->NB = NB; np->C = C; np->K = K; np->B = B; np->T = hm.T;
  np->nops = B * (K + 1);
  // ---- post-order of the internal nodes of the binary tree (explicit stack: caterpillar trees are deep)
  std::vector<int> ipost;
  {
    std::vector<std::pair<int, int>> st;
    st.push_back({t.root, 0});
    while (!st.empty()) {
      auto& top = st.back();
      const int n = top.first;
      if (t.leaf(n)) { st.pop_back(); continue; }
      if (top.second < 2) st.push_back({t.ch[n][top.second++], 0});
      else { ipost.push_back(n); st.pop_back(); }
    }
  }
  std::vector<int> pidx(t.nd, -1);
  for (size_t i = 0; i < ipost.size(); ++i) pidx[ipost[i]] = (int)i;
  // ---- blocks: bottom-up, a node keeps its open children until the open part would exceed NB; then the largest open
  // child subtrees are closed (become blocks of their own) until it fits
  std::vector<int> open(t.nd, 0);
  std::vector<char> closed(t.nd, 0);
  for (int n : ipost) {
    int sz = 1;
    for (int e : t.ch[n]) if (!t.leaf(e) && !closed[e]) sz += open[e];
    while (sz > NB) {
      int big = -1;
      for (int e : t.ch[n]) if (!t.leaf(e) && !closed[e] && (big < 0 || open[e] > open[big])) big = e;
      closed[big] = 1;
      sz -= open[big];
    }
    open[n] = sz;
  }
  closed[t.root] = 1;
  // block roots in phase-1 order = by post-order index; HBM root slots for all but the tree root
  std::vector<int> roots;
  for (int n : ipost) if (closed[n]) roots.push_back(n);
  std::vector<int> hslot(t.nd, -1);
  int nroots = 0;
  for (int r : roots) if (r != t.root) hslot[r] = nroots++;
  np->nblocks = (int)roots.size();
  np->nroots = nroots;
  // ---- records
  std::vector<int> slot(t.nd, -1);
  auto child_desc = [&](int e, int* kind, int* arg, int* tx) {
    if (t.leaf(e)) { *kind = NK_LEAF; *arg = e * (K + 1); *tx = hm.taxon_of[e]; }
    else if (closed[e]) { *kind = NK_HBM; *arg = hslot[e]; *tx = -1; }
    else { *kind = NK_SLOT; *arg = slot[e]; *tx = -1; }
  };
  size_t n_apply = 0, n_rl = 0, n_rs = 0;
  for (int r : roots) {
    // nodes of the block: post-order (inside) and pre-order (outside) over the unclosed internal children
    std::vector<int> post, pre;
    {
      std::vector<std::pair<int, int>> st;
      st.push_back({r, 0});
      while (!st.empty()) {
        auto& top = st.back();
        const int n = top.first;
        if (top.second == 0) pre.push_back(n);
        if (top.second < 2) {
          const int e = t.ch[n][top.second++];
          if (!t.leaf(e) && !closed[e]) st.push_back({e, 0});
        } else { post.push_back(n); st.pop_back(); }
      }
    }
    if ((int)post.size() > NB) return "nucleotide program: a block exceeds its capacity";
    for (size_t i = 0; i < post.size(); ++i) slot[post[i]] = (int)i;
    const int i0 = (int)(np->irec.size() / 8), o0 = (int)(np->orec.size() / 16);
    for (int n : post) {
      int rec[8] = {0, 0, -1, 0, -1, -1, 0, 0};
      int ka, kb;
      child_desc(t.ch[n][0], &ka, &rec[NI_A], &rec[NI_ATX]);
      child_desc(t.ch[n][1], &kb, &rec[NI_B], &rec[NI_BTX]);
      int flags = ka | (kb << 2);
      if (t.pseudo(n)) flags |= NF_PSEUDO;
      if (n == t.root) flags |= NF_ROOT;
      if (n == r) flags |= NF_BLOCKROOT;
      rec[NI_FLAGS] = flags;
      rec[NI_OP] = (t.pseudo(n) || n == t.root) ? -1 : n * (K + 1);
      rec[NI_DST] = (n == r && n != t.root) ? hslot[n] : slot[n];
      np->irec.insert(np->irec.end(), rec, rec + 8);
      // accounting (per class pass): phase 1 visits every node, the recomputation all but the block root
      const int leaves = (ka == NK_LEAF) + (kb == NK_LEAF), hb = (ka == NK_HBM) + (kb == NK_HBM);
      const int own = (t.pseudo(n) || n == t.root) ? 0 : 1;
      n_apply += leaves + own;
      n_rl += hb;
      if (n == r && n != t.root) n_rs += 1;
      if (n != r) { n_apply += leaves + own; n_rl += hb; }
    }
    for (int f : pre) {
      int rec[16];
      std::fill(rec, rec + 16, -1);
      int ka, kb;
      child_desc(t.ch[f][0], &ka, &rec[NO_A], &rec[NO_ATX]);
      child_desc(t.ch[f][1], &kb, &rec[NO_B], &rec[NO_BTX]);
      int flags = ka | (kb << 2);
      if (t.pseudo(f)) flags |= NF_PSEUDO;
      if (f == t.root) flags |= NF_ROOT;
      if (f == r) flags |= NF_BLOCKROOT;
      rec[NO_FLAGS] = flags;
      rec[NO_USRC] = (f == r) ? (f == t.root ? -1 : hslot[f]) : slot[f];
      const bool real = !t.pseudo(f) && f != t.root;
      rec[NO_OPJ] = real ? f * (K + 1) + 1 : -1;
      rec[NO_OPP] = real ? f * (K + 1) : -1;
      rec[NO_ROW] = real ? f * K : -1;
      for (int side = 0; side < 2; ++side) {
        const int e = t.ch[f][side], kind = side ? kb : ka;
        int* disp = &rec[side ? NO_BDISP : NO_ADISP];
        int* row = &rec[side ? NO_BROW : NO_AROW];
        if (kind == NK_LEAF) { *disp = e * (K + 1) + 1; *row = e * K; n_apply += 1 + K; }   // its message + its K counts
        else if (kind == NK_HBM) { *disp = hslot[e]; n_rl += 1; n_rs += 1; }
        else *disp = slot[e];
      }
      if (real) n_apply += K + 1;
      if (f == r && f != t.root) n_rl += 1;
      np->orec.insert(np->orec.end(), rec, rec + 16);
    }
    const int bd[4] = {i0, (int)post.size(), o0, (int)pre.size()};
    np->blk.insert(np->blk.end(), bd, bd + 4);
  }
  np->n_apply = n_apply;
  np->n_root_loads = n_rl;
  np->n_root_stores = n_rs;
  // ---- operators, row-major 4x4 (x -> y): the inside pass applies them as they are, the outside pass transposed
  np->ops.assign((size_t)C * np->nops * 16, 0.0);
  for (int c = 0; c < C; ++c)
    for (int b = 0; b < B; ++b) {
      double* o = &np->ops[((size_t)c * np->nops + (size_t)b * (K + 1)) * 16];
      std::memcpy(o, &hm.P[((size_t)c * B + b) * 16], sizeof(double) * 16);
      for (int k = 0; k < K; ++k) std::memcpy(o + 16 * (1 + k), &hm.PN[(((size_t)c * B + b) * K + k) * 16], sizeof(double) * 16);
    }
  return std::string();
}

// The program run in plain doubles for one site exactly as the device runs it (register slots, HBM root slots, count rows
// accumulated over the classes with weights p_c / L), against a direct computation on the original tree.
std::string verify_nuc_program(const HostModel& hm, const NucProgram& np) {
  const int K = np.K, C = np.C, B = np.B, nn = hm.nn, root = hm.root;
  const double nan = std::nan("");
  std::vector<int> code(hm.T);
  for (int tx = 0; tx < hm.T; ++tx) {      // splitmix-style hash of the taxon index: no global RNG state
    uint64_t z = 0x9E3779B97F4A7C15ull * (uint64_t)(tx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    code[tx] = (int)(z % 5);               // 4 = an unknown: every state compatible
  }
  auto fail = [](const std::string& m) { return "nucleotide program self-check failed: " + m; };
  typedef std::array<double, 4> V4;
  auto evec = [&](int tx) { V4 e; for (int z = 0; z < 4; ++z) e[z] = (code[tx] == 4 || code[tx] == z) ? 1.0 : 0.0; return e; };
  auto op = [&](int c, int idx) -> const double* {
    return (idx < 0 || idx >= np.nops) ? nullptr : &np.ops[((size_t)c * np.nops + idx) * 16];
  };
  auto mvn = [](const double* A, const V4& x) { V4 y; for (int i = 0; i < 4; ++i) { double s = 0; for (int j = 0; j < 4; ++j) s += A[4 * i + j] * x[j]; y[i] = s; } return y; };
  auto mvt = [](const double* A, const V4& x) { V4 y; for (int j = 0; j < 4; ++j) { double s = 0; for (int i = 0; i < 4; ++i) s += A[4 * i + j] * x[i]; y[j] = s; } return y; };
  const V4 bad = {nan, nan, nan, nan};
  std::vector<V4> WM((size_t)C * std::max(1, np.nroots), bad), WU((size_t)C * std::max(1, np.nroots), bad), slots(np.NB, bad);
  std::vector<double> Lc(C, nan), cnt((size_t)B * K, nan);
  std::vector<int> ncount((size_t)B * K, 0);
  std::string err;
  auto child = [&](int c, int kind, int arg, int tx) -> V4 {
    if (kind == NK_LEAF) { const double* A = op(c, arg); if (!A || tx < 0 || tx >= hm.T) { err = "leaf operator"; return bad; } return mvn(A, evec(tx)); }
    if (kind == NK_SLOT) { if (arg < 0 || arg >= np.NB) { err = "register slot out of range"; return bad; } return slots[arg]; }
    if (kind == NK_HBM) { if (arg < 0 || arg >= np.nroots) { err = "root slot out of range"; return bad; } return WM[(size_t)c * np.nroots + arg]; }
    err = "unknown child kind";
    return bad;
  };
  auto inside = [&](int c, const int* r, bool phase1) {
    const int fl = r[NI_FLAGS];
    const V4 Ma = child(c, fl & 3, r[NI_A], r[NI_ATX]), Mb = child(c, (fl >> 2) & 3, r[NI_B], r[NI_BTX]);
    V4 D;
    for (int x = 0; x < 4; ++x) D[x] = Ma[x] * Mb[x];
    if (fl & NF_ROOT) { if (phase1) { double s = 0; for (int x = 0; x < 4; ++x) s += hm.pi[x] * D[x]; Lc[c] = s; } return; }
    V4 M = D;
    if (!(fl & NF_PSEUDO)) { const double* A = op(c, r[NI_OP]); if (!A) { err = "inside operator"; return; } M = mvn(A, D); }
    if (fl & NF_BLOCKROOT) { if (phase1) WM[(size_t)c * np.nroots + r[NI_DST]] = M; }
    else slots[r[NI_DST]] = M;
  };
  // phase 1
  for (int c = 0; c < C; ++c)
    for (int b = 0; b < np.nblocks; ++b) {
      std::fill(slots.begin(), slots.end(), bad);
      const int* bd = &np.blk[(size_t)b * 4];
      for (int i = 0; i < bd[1]; ++i) inside(c, &np.irec[(size_t)(bd[0] + i) * 8], true);
    }
  if (!err.empty()) return fail(err);
  double L = 0;
  for (int c = 0; c < C; ++c) L += hm.probs[c] * Lc[c];
  // phase 2
  auto add_count = [&](int c, int row, double v) {
    if (row < 0 || row >= B * K) { err = "count row out of range"; return; }
    if (ncount[row] != c) { err = "a branch is not counted once per class in class order"; return; }
    ncount[row]++;
    cnt[row] = (c == 0 ? 0.0 : cnt[row]) + hm.probs[c] / L * v;
  };
  for (int b = np.nblocks - 1; b >= 0; --b) {
    const int* bd = &np.blk[(size_t)b * 4];
    for (int c = 0; c < C; ++c) {
      std::fill(slots.begin(), slots.end(), bad);
      for (int i = 0; i + 1 < bd[1]; ++i) inside(c, &np.irec[(size_t)(bd[0] + i) * 8], false);   // all but the block root
      for (int i = 0; i < bd[3]; ++i) {
        const int* r = &np.orec[(size_t)(bd[2] + i) * 16];
        const int fl = r[NO_FLAGS], ka = fl & 3, kb = (fl >> 2) & 3;
        V4 U;
        if (fl & NF_ROOT) U = bad;
        else if (fl & NF_BLOCKROOT) U = WU[(size_t)c * np.nroots + r[NO_USRC]];
        else U = slots[r[NO_USRC]];
        const V4 Ma = child(c, ka, r[NO_A], r[NO_ATX]), Mb = child(c, kb, r[NO_B], r[NO_BTX]);
        V4 Up;
        if (fl & NF_ROOT) { for (int x = 0; x < 4; ++x) Up[x] = hm.pi[x]; }
        else if (fl & NF_PSEUDO) Up = U;
        else {
          for (int k = 0; k < K; ++k) {
            const double* J = op(c, r[NO_OPJ] + k);
            if (!J) { err = "count operator"; break; }
            const V4 W = mvt(J, U);
            double s = 0;
            for (int x = 0; x < 4; ++x) s += W[x] * Ma[x] * Mb[x];
            add_count(c, r[NO_ROW] + k, s);
          }
          const double* P = op(c, r[NO_OPP]);
          if (!P) { err = "transition operator"; break; }
          Up = mvt(P, U);
        }
        for (int side = 0; side < 2; ++side) {
          const int kind = side ? kb : ka, disp = r[side ? NO_BDISP : NO_ADISP], row = r[side ? NO_BROW : NO_AROW];
          const int tx = r[side ? NO_BTX : NO_ATX];
          V4 Uc;
          for (int x = 0; x < 4; ++x) Uc[x] = Up[x] * (side ? Ma[x] : Mb[x]);
          if (kind == NK_LEAF) {
            for (int k = 0; k < K; ++k) {
              const double* J = op(c, disp + k);
              if (!J) { err = "leaf count operator"; break; }
              const V4 Je = mvn(J, evec(tx));
              double s = 0;
              for (int x = 0; x < 4; ++x) s += Uc[x] * Je[x];
              add_count(c, row + k, s);
            }
          } else if (kind == NK_HBM) WU[(size_t)c * np.nroots + disp] = Uc;
          else slots[disp] = Uc;
        }
        if (!err.empty()) break;
      }
      if (!err.empty()) return fail(err);
    }
  }
  // ---- direct computation on the original (possibly multifurcating) tree
  std::vector<double> ref((size_t)B * K, 0.0);
  double Lref = 0;
  auto kids = [&](int n) { std::vector<int> v; for (int e = hm.first_child[n]; e >= 0; e = hm.next_sib[e]) v.push_back(e); return v; };
  std::vector<double> Lcr(C, 0.0);
  std::vector<std::vector<double>> cc(C, std::vector<double>((size_t)B * K, 0.0));
  for (int c = 0; c < C; ++c) {
    std::vector<V4> D(nn), M(nn), U(nn), Up(nn);
    for (int n = 0; n < nn; ++n) {
      if (hm.taxon_of[n] >= 0) D[n] = evec(hm.taxon_of[n]);
      else { D[n] = {1, 1, 1, 1}; for (int e : kids(n)) for (int x = 0; x < 4; ++x) D[n][x] *= M[e][x]; }
      if (n != root) M[n] = mvn(&hm.P[((size_t)c * B + n) * 16], D[n]);
    }
    for (int x = 0; x < 4; ++x) { Lcr[c] += hm.pi[x] * D[root][x]; Up[root][x] = hm.pi[x]; }
    for (int f = nn - 1; f >= 0; --f) {
      if (hm.taxon_of[f] >= 0) continue;
      const std::vector<int> ch = kids(f);
      for (int n : ch) {
        U[n] = Up[f];
        for (int m : ch) if (m != n) for (int x = 0; x < 4; ++x) U[n][x] *= M[m][x];
        for (int k = 0; k < K; ++k) {
          const V4 JD = mvn(&hm.PN[(((size_t)c * B + n) * K + k) * 16], D[n]);
          double s = 0;
          for (int x = 0; x < 4; ++x) s += U[n][x] * JD[x];
          cc[c][(size_t)n * K + k] = s;
        }
        if (hm.taxon_of[n] < 0) Up[n] = mvt(&hm.P[((size_t)c * B + n) * 16], U[n]);
      }
    }
    Lref += hm.probs[c] * Lcr[c];
  }
  for (int c = 0; c < C; ++c)
    for (size_t r = 0; r < ref.size(); ++r) ref[r] += hm.probs[c] * cc[c][r] / Lref;
  auto close = [](double a, double b) { return std::fabs(a - b) <= 1e-9 * (std::fabs(a) + std::fabs(b)) + 1e-290; };
  if (!close(L, Lref)) return fail("site likelihood differs from the direct computation");
  for (size_t r = 0; r < ref.size(); ++r) {
    if (ncount[r] != C) return fail("branch " + std::to_string(r / K) + " is not counted in every class");
    if (!close(cnt[r], ref[r])) return fail("count of branch " + std::to_string(r / K) + " differs from the direct computation");
  }
  return std::string();
}

}  // namespace cmx
