// Host side of the nucleotide mapping path (cmx_nuc.h): cuts the tree into LDS-resident blocks, writes the packet stream
// map_nuc_kernel walks and the leaf tables it gathers from, and checks the whole program numerically -- executed with
// the device's read-ahead semantics -- against a direct pruning computation before a context accepts it.  Plain C++17.
#include "cmx_nuc.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <map>

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
  if (NB < 2 || NB > kNucMaxNB) return "nucleotide program: block capacity must be in 2 .. " + std::to_string(kNucMaxNB);
  if (hm.C > 255) return "nucleotide program: at most 255 rate classes";
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
  const int nblocks = (int)roots.size();
  np->nblocks = nblocks;
  np->nroots = nroots;
  // ---- per block: nodes in post-order (inside) and pre-order (outside), LDS slots, leaf list
  struct Block { std::vector<int> post, pre, leaves; };
  std::vector<Block> blocks(nblocks);
  std::vector<int> slot(t.nd, -1), lpos(t.nd, -1);
  for (int bi = 0; bi < nblocks; ++bi) {
    Block& bk = blocks[bi];
    std::vector<std::pair<int, int>> st;
    st.push_back({roots[bi], 0});
    while (!st.empty()) {
      auto& top = st.back();
      const int n = top.first;
      if (top.second == 0) bk.pre.push_back(n);
      if (top.second < 2) {
        const int e = t.ch[n][top.second++];
        if (!t.leaf(e) && !closed[e]) st.push_back({e, 0});
      } else { bk.post.push_back(n); st.pop_back(); }
    }
    if ((int)bk.post.size() > NB) return "nucleotide program: a block exceeds its capacity";
    for (size_t i = 0; i < bk.post.size(); ++i) slot[bk.post[i]] = (int)i;
    for (int n : bk.post)
      for (int e : t.ch[n])
        if (t.leaf(e)) { lpos[e] = (int)bk.leaves.size(); bk.leaves.push_back(e); }
    if ((int)bk.leaves.size() > kNucMaxLeaves) return "nucleotide program: a block has too many leaves";
  }
  // ---- packets
  auto child_desc = [&](int e, int* kind, int* arg, int* pos) {
    if (t.leaf(e)) { *kind = NK_LEAF; *arg = hm.taxon_of[e]; *pos = lpos[e]; }
    else if (closed[e]) { *kind = NK_HBM; *arg = hslot[e]; *pos = 0; }
    else { *kind = NK_SLOT; *arg = slot[e]; *pos = 0; }
  };
  std::vector<int>& pk = np->pk;
  auto emit = [&](const int (&rec)[8]) { pk.insert(pk.end(), rec, rec + 8); return (int)(pk.size() / 8) - 1; };
  auto emit_block = [&](int bi) {
    int rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const std::vector<int>& lv = blocks[bi].leaves;
    rec[PK_FLAGS] = NF_BLOCKPKT | ((int)lv.size() << 24);
    for (size_t i = 0; i < lv.size(); ++i) rec[1 + i / 2] |= (hm.taxon_of[lv[i]] & 0xffff) << (16 * (i & 1));
    emit(rec);
  };
  if (hm.T > 65535) return "nucleotide program: at most 65535 taxa";
  // who wrote a root message / count row last (packet index): a reader right behind its writer must not read ahead
  std::map<std::pair<int, int>, int> wM, wU;
  std::vector<int> wC((size_t)B * K, -2);
  size_t n_apply = 0, n_leaf = 0, n_rl = 0, n_rs = 0;
  auto inside_pkt = [&](int bi, int n, int c, bool phase1, int extra) {
    const int r = roots[bi];
    int rec[8] = {0, 0, 0, 0, 0, -1, 0, 0};
    int ka, kb;
    child_desc(t.ch[n][0], &ka, &rec[PK_A], &rec[PK_APOS]);
    child_desc(t.ch[n][1], &kb, &rec[PK_B], &rec[PK_BPOS]);
    int flags = ka | (kb << 2) | extra | (c << 24);
    if (t.pseudo(n)) flags |= NF_PSEUDO;
    if (n == t.root) flags |= NF_ROOT;
    if (n == r) flags |= NF_BLOCKROOT;
    if (phase1) flags |= NF_PHASE1;
    rec[PK_NODE] = (t.pseudo(n) || n == t.root) ? -1 : n;
    rec[PK_SLOT] = (n == r && n != t.root) ? hslot[n] : slot[n];
    const int me = (int)(pk.size() / 8);
    for (int side = 0; side < 2; ++side)
      if ((side ? kb : ka) == NK_HBM) {
        auto it = wM.find({c, rec[side ? PK_B : PK_A]});
        if (it == wM.end()) return std::string("nucleotide program: a root message is read before it is written");
        if (it->second == me - 1) flags |= side ? NF_NOPF_B : NF_NOPF_A;
        if (c == 0) n_rl++;
      }
    rec[PK_FLAGS] = flags;
    emit(rec);
    if (phase1 && n == r && n != t.root) { wM[{c, hslot[n]}] = me; if (c == 0) n_rs++; }
    if (c == 0) {
      n_leaf += (ka == NK_LEAF) + (kb == NK_LEAF);
      if (!t.pseudo(n) && n != t.root) n_apply++;
    }
    return std::string();
  };
  auto outside_pkt = [&](int bi, int f, int c, int extra) {
    const int r = roots[bi];
    int rec[8] = {0, 0, 0, 0, 0, -1, 0, 0};
    int ka, kb;
    child_desc(t.ch[f][0], &ka, &rec[PK_A], &rec[PK_APOS]);
    child_desc(t.ch[f][1], &kb, &rec[PK_B], &rec[PK_BPOS]);
    int flags = ka | (kb << 2) | extra | NF_OUTSIDE | (c << 24);
    if (t.pseudo(f)) flags |= NF_PSEUDO;
    if (f == t.root) flags |= NF_ROOT;
    if (f == r) flags |= NF_BLOCKROOT;
    if (c == 0) flags |= NF_FIRSTCLASS;
    if (c == C - 1) flags |= NF_LASTCLASS;
    const bool real = !t.pseudo(f) && f != t.root;
    rec[PK_NODE] = real ? f : -1;
    rec[PK_SLOT] = (f == r) ? (f == t.root ? 0 : hslot[f]) : slot[f];
    rec[PK_KIDS] = (t.leaf(t.ch[f][0]) ? t.ch[f][0] : 0) | ((t.leaf(t.ch[f][1]) ? t.ch[f][1] : 0) << 16);
    const int me = (int)(pk.size() / 8);
    for (int side = 0; side < 2; ++side)
      if ((side ? kb : ka) == NK_HBM) {
        auto it = wM.find({c, rec[side ? PK_B : PK_A]});
        if (it == wM.end()) return std::string("nucleotide program: a root message is read before it is written");
        if (it->second == me - 1) flags |= side ? NF_NOPF_B : NF_NOPF_A;
        if (c == 0) n_rl++;
      }
    if (f == r && f != t.root) {
      auto it = wU.find({c, hslot[f]});
      if (it == wU.end()) return std::string("nucleotide program: an outside message is read before it is written");
      if (it->second == me - 1) flags |= NF_NOPF_U;
      if (c == 0) n_rl++;
    }
    // count rows this packet updates: its own branch and its leaf children's
    std::vector<int> rows;
    if (real) for (int k = 0; k < K; ++k) rows.push_back(f * K + k);
    for (int e : t.ch[f]) if (t.leaf(e)) for (int k = 0; k < K; ++k) rows.push_back(e * K + k);
    for (int row : rows) {
      if (c > 0 && wC[row] == me - 1) flags |= NF_NOPF_CNT;
      wC[row] = me;
    }
    rec[PK_FLAGS] = flags;
    emit(rec);
    for (int side = 0; side < 2; ++side)
      if ((side ? kb : ka) == NK_HBM) { wU[{c, rec[side ? PK_B : PK_A]}] = me; if (c == 0) n_rs++; }
    if (c == 0) {
      if (real) n_apply += K + 1;
      n_leaf += ((ka == NK_LEAF) + (kb == NK_LEAF)) * (1 + K);
    }
    return std::string();
  };
  // One (block, class) list of packets; in the last class of a block the leaf list of the block that follows is read
  // ahead, right behind the list's first packet (by then the current block has taken its own masks over)
  std::string err;
  auto run_list = [&](int bi, int c, bool phase1, bool first_of_block, int next_block) {
    const Block& bk = blocks[bi];
    std::vector<std::pair<int, int>> items;   // (0 inside / 1 outside, node)
    if (phase1) for (int n : bk.post) items.push_back({0, n});
    else {
      for (size_t i = 0; i + 1 < bk.post.size(); ++i) items.push_back({0, bk.post[i]});   // recomputation: all but the root
      for (int f : bk.pre) items.push_back({1, f});
    }
    for (size_t i = 0; i < items.size() && err.empty(); ++i) {
      const int extra = (i == 0 && first_of_block) ? NF_SWAPSYM : 0;
      err = items[i].first ? outside_pkt(bi, items[i].second, c, extra) : inside_pkt(bi, items[i].second, c, phase1, extra);
      if (i == 0 && next_block >= 0) emit_block(next_block);
    }
  };
  emit_block(0);
  for (int bi = 0; bi < nblocks && err.empty(); ++bi)
    for (int c = 0; c < C && err.empty(); ++c) run_list(bi, c, true, c == 0, (c == C - 1 && bi + 1 < nblocks) ? bi + 1 : -1);
  if (!err.empty()) return err;
  pk[(pk.size() / 8 - 1) * 8 + PK_FLAGS] |= NF_FINISH1;   // the tree root's packet of the last class
  for (int bi = nblocks - 1; bi >= 0 && err.empty(); --bi)
    for (int c = 0; c < C && err.empty(); ++c)
      run_list(bi, c, false, c == 0 && bi != nblocks - 1, (c == C - 1 && bi > 0) ? bi - 1 : -1);
  if (!err.empty()) return err;
  np->npk = (int)(pk.size() / 8);
  {
    int end[8] = {NF_END, 0, 0, 0, 0, 0, 0, 0};
    emit(end);
    emit(end);
    emit(end);
  }
  np->n_apply = n_apply;
  np->n_leaf = n_leaf;
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
  // ---- leaf tables: the leaf vector e of a symbol is 1 for every compatible state (DR likelihood leaf initialisation),
  // so the leaf's message is Op e = the sum of the operator's columns over the mask
  np->ltab.assign((size_t)C * hm.T * (K + 1) * 64, 0.0);
  for (int c = 0; c < C; ++c)
    for (int b = 0; b < B; ++b) {
      if (hm.taxon_of[b] < 0) continue;
      for (int w = 0; w <= K; ++w) {
        const double* o = &np->ops[((size_t)c * np->nops + (size_t)b * (K + 1) + w) * 16];
        double* tb = &np->ltab[(((size_t)c * hm.T + hm.taxon_of[b]) * (K + 1) + w) * 64];
        for (int m = 0; m < 16; ++m)
          for (int x = 0; x < 4; ++x) {
            double s = 0.0;
            for (int z = 0; z < 4; ++z) if ((m >> z) & 1) s += o[4 * x + z];
            tb[m * 4 + x] = s;
          }
      }
    }
  return std::string();
}

// The stream run in plain doubles for one site as the device runs it -- LDS slots, root messages and count rows in
// "memory", the leaf masks of a block read ahead, and every read the device issues one packet ahead done one packet
// ahead -- against a direct computation on the original tree.
std::string verify_nuc_program(const HostModel& hm, const NucProgram& np) {
  const int K = np.K, C = np.C, B = np.B, nn = hm.nn, root = hm.root;
  const double nan = std::nan("");
  std::vector<int> mask(hm.T);
  for (int tx = 0; tx < hm.T; ++tx) {      // splitmix-style hash of the taxon index: no global RNG state
    uint64_t z = 0x9E3779B97F4A7C15ull * (uint64_t)(tx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    const int code = (int)(z % 6);         // 4 = an unknown (every state), 5 = a two-state ambiguity code
    mask[tx] = code < 4 ? (1 << code) : (code == 4 ? 15 : 5);
  }
  auto fail = [](const std::string& m) { return "nucleotide program self-check failed: " + m; };
  typedef std::array<double, 4> V4;
  auto op = [&](int c, int idx) -> const double* {
    return (idx < 0 || idx >= np.nops) ? nullptr : &np.ops[((size_t)c * np.nops + idx) * 16];
  };
  auto mvn = [](const double* A, const V4& x) { V4 y; for (int i = 0; i < 4; ++i) { double s = 0; for (int j = 0; j < 4; ++j) s += A[4 * i + j] * x[j]; y[i] = s; } return y; };
  auto mvt = [](const double* A, const V4& x) { V4 y; for (int j = 0; j < 4; ++j) { double s = 0; for (int i = 0; i < 4; ++i) s += A[4 * i + j] * x[i]; y[j] = s; } return y; };
  const V4 bad = {nan, nan, nan, nan};
  const int NR = std::max(1, np.nroots);
  std::vector<V4> WM((size_t)C * NR, bad), WU((size_t)C * NR, bad), slots(np.NB, bad);
  std::vector<double> cnt((size_t)B * K, nan);
  std::vector<int> ncount((size_t)B * K, 0);
  std::string err;
  uint64_t symw = ~0ull, symw_next = ~0ull;
  bool sym_ok = false, symn_ok = false;
  double Lsum = 0.0, rL = nan;
  auto ltab = [&](int c, int tx, int w, int m) -> V4 {
    if (tx < 0 || tx >= hm.T || m <= 0 || m > 15) { err = "leaf table index"; return bad; }
    const double* p = &np.ltab[((((size_t)c * hm.T + tx) * (K + 1) + w) * 16 + m) * 4];
    return V4{p[0], p[1], p[2], p[3]};
  };
  // what a packet's read-ahead delivers
  struct Ahead { V4 G[2], GJ[2], GU; double OC[3]; bool has = false; };
  auto kindA = [](int fl) { return fl & 3; };
  auto kindB = [](int fl) { return (fl >> 2) & 3; };
  auto nib = [&](const int* r, int side, bool next) -> int {
    const uint64_t w = next ? symw_next : symw;
    if (!(next ? symn_ok : sym_ok)) { err = "leaf masks used before their block's leaf list was read"; return 1; }
    return (int)((w >> (4 * r[side ? PK_BPOS : PK_APOS])) & 15);
  };
  auto read_ahead = [&](const int* r, bool ahead) -> Ahead {   // ahead: issued during the previous packet
    Ahead a;
    const int fl = r[PK_FLAGS];
    if (fl & (NF_BLOCKPKT | NF_END)) return a;
    a.has = true;
    const int c = (fl >> 24) & 255;
    const bool swap = ahead && (fl & NF_SWAPSYM);     // its masks are still the "next" ones while the previous packet runs
    for (int side = 0; side < 2; ++side) {
      const int kind = side ? kindB(fl) : kindA(fl), arg = r[side ? PK_B : PK_A];
      a.G[side] = bad; a.GJ[side] = bad;
      if (kind == NK_LEAF) {
        const int m = nib(r, side, swap);
        a.G[side] = ltab(c, arg, 0, m);
        if (fl & NF_OUTSIDE) a.GJ[side] = ltab(c, arg, 1, m);
      } else if (kind == NK_HBM && !(ahead && (fl & (side ? NF_NOPF_B : NF_NOPF_A)))) {
        if (arg < 0 || arg >= np.nroots) { err = "root slot out of range"; return a; }
        a.G[side] = WM[(size_t)c * NR + arg];
      }
    }
    a.GU = bad;
    a.OC[0] = a.OC[1] = a.OC[2] = nan;
    if (fl & NF_OUTSIDE) {
      if ((fl & NF_BLOCKROOT) && !(fl & NF_ROOT) && !(ahead && (fl & NF_NOPF_U))) a.GU = WU[(size_t)c * NR + r[PK_SLOT]];
      if (!(fl & NF_FIRSTCLASS) && !(ahead && (fl & NF_NOPF_CNT))) {
        if (r[PK_NODE] >= 0) a.OC[0] = cnt[(size_t)r[PK_NODE] * K];
        if (kindA(fl) == NK_LEAF) a.OC[1] = cnt[(size_t)(r[PK_KIDS] & 0xffff) * K];
        if (kindB(fl) == NK_LEAF) a.OC[2] = cnt[(size_t)((r[PK_KIDS] >> 16) & 0xffff) * K];
      }
    }
    return a;
  };
  auto add_count = [&](int c, int row, double old_ahead, bool use_ahead, double v) {
    if (row < 0 || row >= B * K) { err = "count row out of range"; return; }
    if (ncount[row] != c) { err = "a branch is not counted once per class in class order"; return; }
    ncount[row]++;
    const double old = c == 0 ? 0.0 : (use_ahead ? old_ahead : cnt[row]);
    cnt[row] = old + hm.probs[c] * rL * v;
  };
  auto body = [&](const int* r, const Ahead& pa) {
    const int fl = r[PK_FLAGS], c = (fl >> 24) & 255, ka = kindA(fl), kb = kindB(fl);
    Ahead now;   // reads that were not allowed to run ahead
    const bool needA = ka == NK_HBM && (fl & NF_NOPF_A), needB = kb == NK_HBM && (fl & NF_NOPF_B);
    const bool needU = (fl & NF_OUTSIDE) && (fl & NF_NOPF_U), needC = (fl & NF_OUTSIDE) && (fl & NF_NOPF_CNT);
    if (needA || needB || needU || needC) now = read_ahead(r, false);
    V4 Ma, Mb;
    if (ka == NK_SLOT) Ma = slots[r[PK_A]]; else Ma = needA ? now.G[0] : pa.G[0];
    if (kb == NK_SLOT) Mb = slots[r[PK_B]]; else Mb = needB ? now.G[1] : pa.G[1];
    if (!(fl & NF_OUTSIDE)) {
      V4 D;
      for (int x = 0; x < 4; ++x) D[x] = Ma[x] * Mb[x];
      if (fl & NF_ROOT) {
        if (fl & NF_PHASE1) { double s = 0; for (int x = 0; x < 4; ++x) s += hm.pi[x] * D[x]; Lsum += hm.probs[c] * s; }
      } else {
        V4 M = D;
        if (!(fl & NF_PSEUDO)) { const double* A = op(c, r[PK_NODE] * (K + 1)); if (!A) { err = "inside operator"; return; } M = mvn(A, D); }
        if (fl & NF_BLOCKROOT) { if (fl & NF_PHASE1) WM[(size_t)c * NR + r[PK_SLOT]] = M; }
        else slots[r[PK_SLOT]] = M;
      }
      if (fl & NF_FINISH1) rL = 1.0 / Lsum;
      return;
    }
    V4 U = bad, Up;
    if (!(fl & NF_ROOT)) U = (fl & NF_BLOCKROOT) ? (needU ? now.GU : pa.GU) : slots[r[PK_SLOT]];
    const Ahead& oc = needC ? now : pa;
    if (fl & NF_ROOT) { for (int x = 0; x < 4; ++x) Up[x] = hm.pi[x]; }
    else if (fl & NF_PSEUDO) Up = U;
    else {
      for (int k = 0; k < K; ++k) {
        const double* J = op(c, r[PK_NODE] * (K + 1) + 1 + k);
        if (!J) { err = "count operator"; return; }
        const V4 W = mvt(J, U);
        double s = 0;
        for (int x = 0; x < 4; ++x) s += W[x] * Ma[x] * Mb[x];
        add_count(c, r[PK_NODE] * K + k, oc.OC[0], k == 0, s);
      }
      const double* P = op(c, r[PK_NODE] * (K + 1));
      if (!P) { err = "transition operator"; return; }
      Up = mvt(P, U);
    }
    for (int side = 0; side < 2; ++side) {
      const int kind = side ? kb : ka, arg = r[side ? PK_B : PK_A];
      V4 Uc;
      for (int x = 0; x < 4; ++x) Uc[x] = Up[x] * (side ? Ma[x] : Mb[x]);
      if (kind == NK_LEAF) {
        const int leaf = (r[PK_KIDS] >> (16 * side)) & 0xffff;
        for (int k = 0; k < K; ++k) {
          const V4 Je = k == 0 ? pa.GJ[side] : ltab(c, arg, 1 + k, nib(r, side, false));
          double s = 0;
          for (int x = 0; x < 4; ++x) s += Uc[x] * Je[x];
          add_count(c, leaf * K + k, oc.OC[1 + side], k == 0, s);
        }
      } else if (kind == NK_HBM) WU[(size_t)c * NR + arg] = Uc;
      else slots[arg] = Uc;
    }
  };
  Ahead pa;
  for (int i = 0; i < np.npk; ++i) {
    const int* r = &np.pk[(size_t)i * 8];
    const int* r1 = &np.pk[(size_t)(i + 1) * 8];
    const bool blockpkt = (r[PK_FLAGS] & NF_BLOCKPKT) != 0;
    if (blockpkt) {
      const int nl = (r[PK_FLAGS] >> 24) & 255;
      symw_next = 0;
      for (int q = 0; q < nl; ++q) {
        const int tx = (r[1 + q / 2] >> (16 * (q & 1))) & 0xffff;
        if (tx >= hm.T) return fail("taxon out of range in a leaf list");
        symw_next |= (uint64_t)mask[tx] << (4 * q);
      }
      symn_ok = true;
    }
    if (!blockpkt && (r[PK_FLAGS] & NF_SWAPSYM)) { symw = symw_next; sym_ok = symn_ok; }   // before the read-ahead, as on the device
    const Ahead nxt = read_ahead(r1, true);
    if (!blockpkt) body(r, pa);
    if (!err.empty()) return fail(err + " (packet " + std::to_string(i) + ")");
    pa = nxt;
  }
  if (!(np.pk[(size_t)np.npk * 8] & NF_END)) return fail("stream is not terminated");
  // ---- direct computation on the original (possibly multifurcating) tree
  auto evec = [&](int tx) { V4 e; for (int z = 0; z < 4; ++z) e[z] = ((mask[tx] >> z) & 1) ? 1.0 : 0.0; return e; };
  std::vector<double> ref((size_t)B * K, 0.0);
  double Lref = 0;
  auto kids = [&](int n) { std::vector<int> v; for (int e = hm.first_child[n]; e >= 0; e = hm.next_sib[e]) v.push_back(e); return v; };
  std::vector<std::vector<double>> cc(C, std::vector<double>((size_t)B * K, 0.0));
  for (int c = 0; c < C; ++c) {
    std::vector<V4> D(nn), M(nn), U(nn), Up(nn);
    for (int n = 0; n < nn; ++n) {
      if (hm.taxon_of[n] >= 0) D[n] = evec(hm.taxon_of[n]);
      else { D[n] = {1, 1, 1, 1}; for (int e : kids(n)) for (int x = 0; x < 4; ++x) D[n][x] *= M[e][x]; }
      if (n != root) M[n] = mvn(&hm.P[((size_t)c * B + n) * 16], D[n]);
    }
    double Lc = 0;
    for (int x = 0; x < 4; ++x) { Lc += hm.pi[x] * D[root][x]; Up[root][x] = hm.pi[x]; }
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
    Lref += hm.probs[c] * Lc;
  }
  for (int c = 0; c < C; ++c)
    for (size_t r = 0; r < ref.size(); ++r) ref[r] += hm.probs[c] * cc[c][r] / Lref;
  auto close = [](double a, double b) { return std::fabs(a - b) <= 1e-9 * (std::fabs(a) + std::fabs(b)) + 1e-290; };
  if (!close(Lsum, Lref)) return fail("site likelihood differs from the direct computation");
  for (size_t r = 0; r < ref.size(); ++r) {
    if (ncount[r] != C) return fail("branch " + std::to_string(r / K) + " is not counted in every class");
    if (!close(cnt[r], ref[r])) return fail("count of branch " + std::to_string(r / K) + " differs from the direct computation");
  }
  return std::string();
}

}  // namespace cmx
