// Host-side model preparation (see cmx_host_model.h).  Plain C++17, no device code.
#include "cmx_host_model.h"
#include "cmx_device.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace cmx {
namespace {

using Mat = std::vector<double>;

Mat matmul(int n, const Mat& A, const Mat& B) {
  Mat C((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < n; ++k) {
      const double a = A[(size_t)i * n + k];
      for (int j = 0; j < n; ++j) C[(size_t)i * n + j] += a * B[(size_t)k * n + j];
    }
  return C;
}

// cyclic Jacobi eigen-solver for a symmetric matrix (n <= 64 here); columns of U are eigenvectors
void jacobi(int n, Mat A, Mat* U, std::vector<double>* lam) {
  U->assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) (*U)[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 200; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += A[(size_t)i * n + i] * A[(size_t)i * n + i];
      for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j];
    }
    if (off <= 1e-40 * (diag + 1e-300)) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
          A[(size_t)k * n + p] = c * akp - s * akq;
          A[(size_t)k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
          A[(size_t)p * n + k] = c * apk - s * aqk;
          A[(size_t)q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double ukp = (*U)[(size_t)k * n + p], ukq = (*U)[(size_t)k * n + q];
          (*U)[(size_t)k * n + p] = c * ukp - s * ukq;
          (*U)[(size_t)k * n + q] = s * ukp + c * ukq;
        }
      }
  }
  lam->resize(n);
  for (int i = 0; i < n; ++i) (*lam)[i] = A[(size_t)i * n + i];
}

// row-major SxS -> 4x4-block packed (tile t = (bi, bj), element k = (k/4, k%4)); layout consumed by matvec_s
void pack_blocks(int S, const double* M, double* out) {
  const int NB = S / 4;
  for (int t = 0; t < NB * NB; ++t)
    for (int k = 0; k < 16; ++k) out[t * 16 + k] = M[(size_t)(4 * (t / NB) + k / 4) * S + 4 * (t % NB) + k % 4];
}

}  // namespace

// Builds the traversal of one rate-class pass: the per-node records the kernel walks, and -- by simulating exactly
// the kernel's control flow (map_sites_wave in cmx_kernels.hip) -- the order of its workspace loads (ldsched, with
// prefetchability) and of its matrix products (msched).
//
// Bytes are ~10x dearer than flops in this kernel (a 10 KiB workspace vector costs a wave ~5 us of its HBM share, a
// 20x20 product ~1 us), so the traversal avoids workspace traffic wherever a vector can stay in registers or be
// rebuilt from leaves:
//   * "inlined cherries": an internal node whose two children are leaves (and whose parent is binary) is never
//     visited on its own.  Its inside vector is rebuilt from two leaf gathers where needed, and the counts of its
//     two leaf branches are taken right where its outside message is produced -- no store, no load for either.
//   * a binary node's child Y = its largest-id child that is a visited node: Y is visited right before (inside
//     pass) / right after (outside pass) its parent, so its vectors are handed over in registers both ways.
void build_load_schedule(HostModel* hm) {
  const int NI = hm->NI, root = hm->root, K = hm->K, nn = hm->nn;
  auto internal = [&](int n) { return hm->taxon_of[n] < 0; };
  auto kids = [&](int n) { std::vector<int> v; for (int e = hm->first_child[n]; e >= 0; e = hm->next_sib[e]) v.push_back(e); return v; };
  std::vector<char> inlined(nn, 0);
  for (int n = 0; n < nn; ++n) {
    if (!internal(n) || n == root) continue;
    const std::vector<int> c = kids(n);
    if (c.size() == 2 && !internal(c[0]) && !internal(c[1]) && kids(hm->parent[n]).size() == 2) inlined[n] = 1;
  }
  auto kind = [&](int e) { return !internal(e) ? 0 : (inlined[e] ? 2 : 1); };
  std::vector<int> visited;
  for (int idx = 0; idx < NI; ++idx) if (!inlined[hm->int_post[idx]]) visited.push_back(hm->int_post[idx]);
  const int NV = (int)visited.size();
  hm->NV = NV;
  hm->nrec.assign((size_t)NV * 16, -1);
  // child descriptor, 5 ints: kind, node (= branch), slot (stored) / taxon (leaf), and for an inlined cherry its two
  // leaf nodes (their taxa are named by the op stream)
  auto fill_child = [&](int* r, int e) {
    r[0] = kind(e); r[1] = e;
    r[2] = internal(e) ? hm->slot[e] : hm->taxon_of[e];
    if (kind(e) == 2) {
      const std::vector<int> c = kids(e);
      r[3] = c[0]; r[4] = c[1];
    }
  };
  std::vector<int> Xof(nn, -1), Yof(nn, -1);
  for (int v = 0; v < NV; ++v) {
    const int n = visited[v];
    int* r = &hm->nrec[(size_t)v * 16];
    const std::vector<int> c = kids(n);
    r[0] = n; r[1] = hm->slot[n]; r[2] = (int)c.size(); r[3] = 0; r[14] = -1;
    if (c.size() == 2) {
      int y = (kind(c[1]) == 1) ? c[1] : ((kind(c[0]) == 1) ? c[0] : c[1]);
      int x = (y == c[1]) ? c[0] : c[1];
      Xof[n] = x; Yof[n] = y;
      fill_child(r + 4, x);
      fill_child(r + 9, y);
      if (kind(y) == 1) r[3] |= 2;   // Y's vectors are handed over in registers
    } else {
      const int last = c.back();
      if (kind(last) == 1) r[14] = last;   // general node: its last child is the node finished right before it
    }
  }
  // ---- simulate the kernel
  struct Ev { int arr, slot; long t, src_store; };
  std::vector<Ev> pops;
  std::vector<long> storeD(NI, -1), storeU(NI, -1);
  long t = 0;
  hm->stores_D = hm->stores_U = 0;
  auto pop = [&](int arr, int slot) { pops.push_back({arr, slot, t++, arr ? storeU[slot] : storeD[slot]}); };
  hm->msched.clear();
  // op stream: one (matrix index within a class block, taxon or -1) pair per matrix use, in kernel program order.
  // Class block layout (HostModel::MAT): P[slot] | J[slot*K+k] | leaf P^T[taxon] | leaf J^T[k*T+taxon]
  const int T = hm->T;
  auto op = [&](int mat, int tx) { hm->msched.push_back(mat); hm->msched.push_back(tx); };
  auto mvP = [&](int slot) { op(slot, -1); };
  auto mvJ = [&](int slot) { for (int k = 0; k < K; ++k) op(NI + slot * K + k, -1); };
  auto leafP = [&](int tx) { op(NI + NI * K + tx, tx); };
  auto leafJ = [&](int tx) { for (int k = 0; k < K; ++k) op(NI + NI * K + T + k * T + tx, tx); };
  auto txof = [&](int e) { return hm->taxon_of[e]; };
  auto cherry_leaves = [&](int e) { const std::vector<int> c = kids(e); leafP(txof(c[0])); leafP(txof(c[1])); };
  auto cherry_counts = [&](int e) {
    const std::vector<int> c = kids(e);
    leafP(txof(c[1])); leafJ(txof(c[0])); leafP(txof(c[0])); leafJ(txof(c[1]));
  };
  // inside vector of child edge e into registers (CMX_GET_D)
  auto getD = [&](int e) { if (kind(e) == 1) pop(0, hm->slot[e]); else cherry_leaves(e); };
  // inside pass
  for (int v = 0; v < NV; ++v) {
    const int n = visited[v];
    const std::vector<int> c = kids(n);
    if (c.size() == 2) {
      const int x = Xof[n], y = Yof[n];
      if (kind(y) == 1) mvP(hm->slot[y]);                       // carried
      auto edge = [&](int e) {
        if (kind(e) == 0) leafP(txof(e));
        else { getD(e); mvP(hm->slot[e]); }
      };
      edge(x);
      if (kind(y) != 1) edge(y);
    } else {
      const int carry = hm->nrec[(size_t)v * 16 + 14];
      if (carry >= 0) mvP(hm->slot[carry]);
      for (int e : c) {
        if (!internal(e)) leafP(txof(e));
        else if (e != carry) { pop(0, hm->slot[e]); mvP(hm->slot[e]); }
      }
    }
    if (n != root) { storeD[hm->slot[n]] = t++; hm->stores_D++; }
  }
  // outside pass
  std::vector<char> up_in_acc(nn, 0);
  for (int v = NV - 1; v >= 0; --v) {
    const int f = visited[v];
    if (f != root) {
      if (up_in_acc[f]) hm->nrec[(size_t)v * 16 + 3] |= 4;
      else pop(1, hm->slot[f]);
    }
    const std::vector<int> c = kids(f);
    if (c.size() == 2) {
      const int x = Xof[f], y = Yof[f];
      if (kind(y) == 0) leafP(txof(y)); else { getD(y); mvP(hm->slot[y]); }
      if (kind(x) == 0) { leafJ(txof(x)); leafP(txof(x)); }
      else {
        getD(x); mvJ(hm->slot[x]); mvP(hm->slot[x]); mvP(hm->slot[x]);
        if (kind(x) == 1) { storeU[hm->slot[x]] = t++; hm->stores_U++; } else cherry_counts(x);
      }
      if (kind(y) == 0) leafJ(txof(y));
      else {
        getD(y); mvJ(hm->slot[y]); mvP(hm->slot[y]);
        if (kind(y) == 1) up_in_acc[y] = 1; else cherry_counts(y);
      }
    } else {
      for (int n : c) {
        for (int sb : c) {
          if (sb == n) continue;
          if (!internal(sb)) leafP(txof(sb)); else { pop(0, hm->slot[sb]); mvP(hm->slot[sb]); }
        }
        if (!internal(n)) leafJ(txof(n));
        else { pop(0, hm->slot[n]); mvJ(hm->slot[n]); mvP(hm->slot[n]); storeU[hm->slot[n]] = t++; hm->stores_U++; }
      }
    }
  }
  hm->ldsched.clear();
  hm->loads_D = hm->loads_U = 0;
  for (size_t j = 0; j < pops.size(); ++j) {
    const Ev& e = pops[j];
    unsigned w = (unsigned)e.slot | (e.arr ? 0x40000000u : 0u);
    // prefetchable: its producer store is issued before the previous pop (where the prefetch is issued)
    if (j > 0 && e.src_store >= 0 && e.src_store < pops[j - 1].t) w |= 0x80000000u;
    hm->ldsched.push_back((int)w);
    if (e.arr) hm->loads_U++; else hm->loads_D++;
  }
}

// Host-side dry run of map_sites_wave's control flow, driven by the SAME records and schedules the kernel reads:
// every index is bounds-checked, every workspace load must name the vector the code needs and must have been stored
// before, every matrix product must find its matrix next in msched.  A mismatch here would be an out-of-bounds or
// stale access on the GPU, so a context is refused instead (cmx_ctx_create).
std::string verify_traversal(const HostModel& hm) {
  const int NI = hm.NI, NV = hm.NV, K = hm.K, T = hm.T, nn = hm.nn, root = hm.root;
  if ((int)hm.nrec.size() != NV * 16) return "nrec size";
  size_t fi = 0, mi = 0;
  std::vector<char> haveD(NI, 0), haveU(NI, 0), counted((size_t)hm.B * K, 0);
  std::string err;
  auto fail = [&](const std::string& m) { if (err.empty()) err = "traversal self-check failed: " + m; };
  auto pop = [&](int arr, int slot) {
    if (fi >= hm.ldsched.size()) return fail("more workspace loads than scheduled");
    const int e = hm.ldsched[fi++];
    if ((((unsigned)e >> 30) & 1) != (unsigned)arr || (e & 0xffffff) != slot) return fail("load " + std::to_string(fi - 1) + " names the wrong vector");
    if (slot < 0 || slot >= NI) return fail("slot out of range");
    if (!(arr ? haveU[slot] : haveD[slot])) return fail("load of a vector that was never stored");
  };
  const int MC = NI + NI * K + T + K * T;  // matrices per class block
  auto opchk = [&](int want_mat, int want_tx, const char* what) {
    if (2 * mi + 1 >= hm.msched.size()) return fail(std::string("more matrix uses than scheduled (") + what + ")");
    const int mat = hm.msched[2 * mi], tx = hm.msched[2 * mi + 1];
    ++mi;
    if (mat < 0 || mat >= MC) return fail("matrix index out of range");
    if (mat != want_mat || tx != want_tx) return fail("op " + std::to_string(mi - 1) + " (" + what + ") finds the wrong matrix staged");
  };
  auto mv = [&](bool isJ, int idx) {
    if (isJ ? (idx < 0 || idx >= NI * K) : (idx < 0 || idx >= NI)) return fail("matrix index out of range");
    opchk(isJ ? NI + idx : idx, -1, isJ ? "count product" : "product");
  };
  auto leaf = [&](int tx) {  // leaf edge, transition matrix
    if (tx < 0 || tx >= T) return fail("taxon out of range");
    opchk(NI + NI * K + tx, tx, "leaf P");
  };
  auto leafJ = [&](int tx) {
    if (tx < 0 || tx >= T) return fail("taxon out of range");
    for (int k = 0; k < K; ++k) opchk(NI + NI * K + T + k * T + tx, tx, "leaf J");
  };
  auto count = [&](int node) {
    if (node < 0 || node >= nn - 1) return fail("branch out of range");
    for (int k = 0; k < K; ++k) { if (counted[(size_t)node * K + k]) fail("branch counted twice"); counted[(size_t)node * K + k] = 1; }
  };
  auto tx_of = [&](int node) { return (node >= 0 && node < nn) ? hm.taxon_of[node] : -1; };  // leaf node -> taxon
  auto getD = [&](const int* ch) {
    if (ch[0] == 1) pop(0, ch[2]);
    else if (ch[0] == 2) { leaf(tx_of(ch[3])); leaf(tx_of(ch[4])); }
    else fail("bad child kind");
  };
  auto cherry_counts = [&](const int* ch) {
    leaf(tx_of(ch[4])); leafJ(tx_of(ch[3])); count(ch[3]); leaf(tx_of(ch[3])); leafJ(tx_of(ch[4])); count(ch[4]);
  };
  auto kids = [&](int n) { std::vector<int> v; for (int e = hm.first_child[n]; e >= 0; e = hm.next_sib[e]) v.push_back(e); return v; };
  bool acc_is_D_of_prev = false;
  int prev_node = -1;
  // inside pass
  for (int idx = 0; idx < NV && err.empty(); ++idx) {
    const int* r = &hm.nrec[(size_t)idx * 16];
    const int n = r[0];
    if (n < 0 || n >= nn || r[1] != hm.slot[n]) return "traversal self-check failed: bad node record";
    if (r[2] == 2) {
      const int* X = r + 4; const int* Y = r + 9;
      if (r[3] & 2) {
        if (!acc_is_D_of_prev || prev_node != Y[1] || Y[0] != 1) fail("Y is not the node finished last");
        mv(false, Y[2]);
      }
      if (X[0] == 0) leaf(X[2]); else { getD(X); mv(false, X[2]); }
      if (!(r[3] & 2)) { if (Y[0] == 0) leaf(Y[2]); else { if (Y[0] != 2) fail("stored Y not handed over"); getD(Y); mv(false, Y[2]); } }
    } else {
      const int carry = r[14];
      if (carry >= 0) { if (!acc_is_D_of_prev || prev_node != carry) fail("general carry"); mv(false, hm.slot[carry]); }
      for (int e : kids(n)) {
        if (hm.taxon_of[e] >= 0) leaf(hm.taxon_of[e]);
        else if (e != carry) { pop(0, hm.slot[e]); mv(false, hm.slot[e]); }
      }
    }
    if (n != root) haveD[r[1]] = 1;
    acc_is_D_of_prev = true;
    prev_node = n;
  }
  // outside pass
  int up_node_in_acc = -1;
  for (int idx = NV - 1; idx >= 0 && err.empty(); --idx) {
    const int* r = &hm.nrec[(size_t)idx * 16];
    const int f = r[0];
    if (f != root) {
      if (r[3] & 4) { if (up_node_in_acc != f) fail("outside message not in registers"); }
      else pop(1, r[1]);
    }
    up_node_in_acc = -1;
    if (r[2] == 2) {
      const int* X = r + 4; const int* Y = r + 9;
      if (Y[0] == 0) leaf(Y[2]); else { getD(Y); mv(false, Y[2]); }
      if (X[0] == 0) { leafJ(X[2]); count(X[1]); leaf(X[2]); }
      else {
        getD(X);
        for (int k = 0; k < K; ++k) mv(true, X[2] * K + k);
        count(X[1]);
        mv(false, X[2]); mv(false, X[2]);
        if (X[0] == 1) haveU[X[2]] = 1; else cherry_counts(X);
      }
      if (Y[0] == 0) { leafJ(Y[2]); count(Y[1]); }
      else {
        getD(Y);
        for (int k = 0; k < K; ++k) mv(true, Y[2] * K + k);
        count(Y[1]);
        mv(false, Y[2]);
        if (Y[0] == 1) up_node_in_acc = Y[1]; else cherry_counts(Y);
      }
    } else {
      const std::vector<int> c = kids(f);
      for (int n : c) {
        for (int sb : c)
          if (sb != n) { if (hm.taxon_of[sb] >= 0) leaf(hm.taxon_of[sb]); else { pop(0, hm.slot[sb]); mv(false, hm.slot[sb]); } }
        if (hm.taxon_of[n] >= 0) { leafJ(hm.taxon_of[n]); count(n); }
        else {
          pop(0, hm.slot[n]);
          for (int k = 0; k < K; ++k) mv(true, hm.slot[n] * K + k);
          count(n);
          mv(false, hm.slot[n]);
          haveU[hm.slot[n]] = 1;
        }
      }
    }
  }
  if (!err.empty()) return err;
  if (fi != hm.ldsched.size()) return "traversal self-check failed: unused workspace loads in the schedule";
  if (2 * mi != hm.msched.size()) return "traversal self-check failed: unused matrix uses in the schedule";
  for (char c : counted) if (!c) return "traversal self-check failed: a branch is never counted";
  return std::string();
}

std::string build_host_model(const cmx_model* model, const cmx_tree* tree, HostModel* hm, int* code) {
  *code = CMX_ERR_INVALID;
  if (!model || !tree) return "model and tree are required";
  const int S = model->nstates, C = model->nclasses;
  const int K = (model->nmodels > 0 ? model->Bks : model->Bk) ? model->ntypes : 1;
  if (S != 4 && S != 20) {
    *code = CMX_ERR_UNSUPPORTED;
    return "nstates must be 4 (nucleotides) or 20 (proteins); got " + std::to_string(S);
  }
  if (C < 1 || C > 64) return "nclasses out of range";
  if (K < 1 || K > 64) return "ntypes out of range";
  if (!model->rates || !model->probs) return "rates and probs are required";
  if (model->nmodels <= 0 && (!model->Q || !model->pi)) return "Q, pi, rates and probs are required";
  const int nn = tree->nnodes, T = tree->ntaxa;
  if (nn < 3 || T < 2 || !tree->parent || !tree->blen || !tree->leaf_of_taxon) return "tree is incomplete";
  if (nn > 65535) return "tree too large";
  hm->S = S; hm->C = C; hm->K = K; hm->nn = nn; hm->B = nn - 1; hm->T = T; hm->root = nn - 1;
  hm->parent.assign(tree->parent, tree->parent + nn);
  hm->blen.assign(tree->blen, tree->blen + nn);
  // ---- tree checks: post-order, root last
  if (hm->parent[nn - 1] != -1) return "parent[root] must be -1 with the root last";
  for (int i = 0; i < nn - 1; ++i) {
    if (hm->parent[i] <= i || hm->parent[i] >= nn) return "nodes must be in post-order (parent id > child id)";
    if (!(hm->blen[i] >= 0.0) || !std::isfinite(hm->blen[i])) return "branch lengths must be finite and >= 0";
  }
  hm->first_child.assign(nn, -1);
  hm->next_sib.assign(nn, -1);
  std::vector<int> last(nn, -1), nchild(nn, 0);
  for (int i = 0; i < nn - 1; ++i) {
    const int p = hm->parent[i];
    if (hm->first_child[p] < 0) hm->first_child[p] = i; else hm->next_sib[last[p]] = i;
    last[p] = i;
    nchild[p]++;
  }
  hm->taxon_of.assign(nn, -1);
  for (int t = 0; t < T; ++t) {
    const int n = tree->leaf_of_taxon[t];
    if (n < 0 || n >= nn || nchild[n] != 0) return "leaf_of_taxon must name leaves";
    if (hm->taxon_of[n] >= 0) return "leaf_of_taxon has duplicates";
    hm->taxon_of[n] = t;
  }
  hm->slot.assign(nn, -1);
  hm->int_post.clear();
  for (int i = 0; i < nn; ++i) {
    if (nchild[i] == 0) {
      if (hm->taxon_of[i] < 0) return "every leaf needs an alignment row";
    } else {
      if (nchild[i] < 2 && i != nn - 1) return "internal nodes need at least two children";
      hm->slot[i] = (int)hm->int_post.size();
      hm->int_post.push_back(i);
    }
  }
  if (nchild[nn - 1] < 2) return "the root needs at least two children";
  hm->NI = (int)hm->int_post.size();
  build_load_schedule(hm);
  {
    const std::string bad = verify_traversal(*hm);
    if (!bad.empty()) return bad;
  }
  // ---- model checks
  const bool nh = model->nmodels > 0;
  const int NM = nh ? model->nmodels : 1;
  if (nh && (!model->Qs || !model->pis || !model->model_of_branch || !model->root_freqs))
    return "non-homogeneous model: Qs, pis, model_of_branch and root_freqs are required";
  if (!nh && (model->Qs || model->pis || model->Bks || model->model_of_branch || model->root_freqs))
    return "non-homogeneous fields are set but nmodels is 0";
  // frequencies at the root: the stationary ones, or the model set's root frequency set
  const double* rootf = nh ? model->root_freqs : model->pi;
  hm->pi.assign(rootf, rootf + S);
  hm->rates.assign(model->rates, model->rates + C);
  hm->probs.assign(model->probs, model->probs + C);
  double spi = 0, spr = 0;
  for (double v : hm->pi) { if (!(v > 0)) return "pi must be positive"; spi += v; }
  for (double v : hm->probs) { if (!(v >= 0)) return "probs must be >= 0"; spr += v; }
  if (std::fabs(spi - 1.0) > 1e-6 || std::fabs(spr - 1.0) > 1e-6) return "pi and probs must sum to one";
  for (double v : hm->rates) if (!(v >= 0) || !std::isfinite(v)) return "rates must be finite and >= 0";
  const size_t S2 = (size_t)S * S;
  std::vector<int> model_of(hm->B, 0);
  if (nh)
    for (int b = 0; b < hm->B; ++b) {
      model_of[b] = model->model_of_branch[b];
      if (model_of[b] < 0 || model_of[b] >= NM) return "model_of_branch: generator index out of range";
    }
  // ---- per generator: checks and eigen-decomposition through the symmetrised generator
  struct Eig { Mat V, Vi; std::vector<double> lam; std::vector<Mat> W; };
  std::vector<Eig> eig(NM);
  for (int m = 0; m < NM; ++m) {
    const double* Qp = nh ? model->Qs + (size_t)m * S2 : model->Q;
    const double* pip = nh ? model->pis + (size_t)m * S : model->pi;
    const double* Bp = nh ? (model->Bks ? model->Bks + (size_t)m * K * S2 : nullptr) : model->Bk;
    Mat Q(Qp, Qp + S2);
    double sp = 0;
    for (int x = 0; x < S; ++x) { if (!(pip[x] > 0)) return "pi must be positive"; sp += pip[x]; }
    if (std::fabs(sp - 1.0) > 1e-6) return "pi and probs must sum to one";
    for (int x = 0; x < S; ++x) {
      double rs = 0;
      for (int y = 0; y < S; ++y) {
        rs += Q[(size_t)x * S + y];
        const double a = pip[x] * Q[(size_t)x * S + y], b = pip[y] * Q[(size_t)y * S + x];
        if (std::fabs(a - b) > 1e-9 * (std::fabs(a) + std::fabs(b) + 1e-300) + 1e-14)
          return "Q must be reversible with respect to pi (pi_x Q_xy == pi_y Q_yx)";
      }
      if (std::fabs(rs) > 1e-9) return "rows of Q must sum to zero";
    }
    std::vector<Mat> Bk(K);
    for (int k = 0; k < K; ++k) {
      if (Bp) Bk[k].assign(Bp + k * S2, Bp + (k + 1) * S2);
      else { Bk[k] = Q; for (int x = 0; x < S; ++x) Bk[k][(size_t)x * S + x] = 0.0; }
    }
    Mat A(S2), U;
    for (int x = 0; x < S; ++x)
      for (int y = 0; y < S; ++y) A[(size_t)x * S + y] = std::sqrt(pip[x]) * Q[(size_t)x * S + y] / std::sqrt(pip[y]);
    for (int x = 0; x < S; ++x)
      for (int y = x + 1; y < S; ++y) A[(size_t)x * S + y] = A[(size_t)y * S + x] = 0.5 * (A[(size_t)x * S + y] + A[(size_t)y * S + x]);
    Eig& e = eig[m];
    jacobi(S, A, &U, &e.lam);
    e.V.resize(S2); e.Vi.resize(S2);
    for (int x = 0; x < S; ++x)
      for (int j = 0; j < S; ++j) {
        e.V[(size_t)x * S + j] = U[(size_t)x * S + j] / std::sqrt(pip[x]);
        e.Vi[(size_t)j * S + x] = U[(size_t)x * S + j] * std::sqrt(pip[x]);
      }
    e.W.resize(K);   // Vinv B_k V
    for (int k = 0; k < K; ++k) e.W[k] = matmul(S, matmul(S, e.Vi, Bk[k]), e.V);
  }
  // ---- per (class, branch) matrices, each branch with its own generator
  const int B = hm->B;
  hm->P.assign((size_t)C * B * S2, 0.0);
  hm->PN.assign((size_t)C * B * K * S2, 0.0);
  Mat E(S2), Phi(S2);
  for (int c = 0; c < C; ++c)
    for (int b = 0; b < B; ++b) {
      const Eig& e = eig[model_of[b]];
      const Mat &V = e.V, &Vi = e.Vi;
      const std::vector<double>& lam = e.lam;
      const std::vector<Mat>& W = e.W;
      const double t = hm->blen[b] * hm->rates[c];
      double* P = &hm->P[((size_t)c * B + b) * S2];
      for (int x = 0; x < S; ++x)
        for (int j = 0; j < S; ++j) E[(size_t)x * S + j] = V[(size_t)x * S + j] * std::exp(lam[j] * t);
      Mat Pm = matmul(S, E, Vi);
      std::memcpy(P, Pm.data(), sizeof(double) * S2);
      for (int k = 0; k < K; ++k) {
        double* PNk = &hm->PN[(((size_t)c * B + b) * K + k) * S2];
        if (model->count_method == CMX_COUNT_NAIVE) {
          for (size_t i = 0; i < S2; ++i)
            PNk[i] = (i / S == i % S) ? 0.0 : P[i] * (model->naive_weights ? model->naive_weights[i] : 1.0);
          continue;
        }
        // J = V [ (Vinv B V) o Phi ] Vinv; Phi through expm1 (accurate O(t^2) diagonal on 1e-6 branches)
        for (int i = 0; i < S; ++i)
          for (int j = 0; j < S; ++j) {
            const double d = (lam[i] - lam[j]) * t;
            const double phi = std::fabs(d) < 1e-14 ? t * std::exp(lam[i] * t) : t * std::exp(lam[j] * t) * std::expm1(d) / d;
            Phi[(size_t)i * S + j] = W[k][(size_t)i * S + j] * phi;
          }
        Mat J = matmul(S, matmul(S, V, Phi), Vi);
        for (size_t i = 0; i < S2; ++i) {
          double nxy = J[i] / P[i];  // conditional count; Bio++ guards: non-finite -> 0, unweighted negatives -> 0
          if (std::isnan(nxy) || std::isinf(nxy)) nxy = 0.0;
          if (model->clamp_negative && nxy < 0.0) nxy = 0.0;
          PNk[i] = P[i] * nxy;
        }
      }
    }
  // ---- device layouts
  // One allocation, per class a block of MC matrices of S*S doubles (the unit the kernel DMAs into LDS):
  //   [0, NI)                    P of internal edges, 4x4-block packed (matrix-vector products)
  //   [NI, NI + NI*K)            P o N^k of internal edges, packed, index slot*K + k
  //   [.., + T)                  P of leaf edges transposed, [z][x] = P[x][z] (per-lane row gather by observed symbol)
  //   [.., + K*T)                P o N^k of leaf edges transposed, index k*T + taxon
  const int NI = hm->NI;
  const int MC = NI + NI * K + T + K * T;
  hm->MC = MC;
  hm->fuse = (S == 4 && C >= 4) ? (C == 4 ? 4 : 5) : 1;
  const int F = hm->fuse;
  const int dS = S * F, dC = (C + F - 1) / F;
  hm->dS = dS;
  hm->dC = dC;
  const size_t MU = (size_t)mat_unit(dS);   // doubles per device matrix: dS*dS plus max_ambig(dS) extra leaf rows
  const size_t dS2 = (size_t)dS * dS;
  hm->MAT.assign((size_t)dC * MC * MU, 0.0);
  hm->CP.assign((size_t)C * nn * S2, 0.0);
  for (int c = 0; c < C; ++c)
    for (int b = 0; b < B; ++b) {
      const double* P = &hm->P[((size_t)c * B + b) * S2];
      double* cp = &hm->CP[((size_t)c * nn + b) * S2];
      for (int x = 0; x < S; ++x) {
        double cum = 0.0;
        for (int y = 0; y < S; ++y) { cum += P[(size_t)x * S + y]; cp[(size_t)x * S + y] = cum; }
      }
    }
  Mat dense(dS2);
  for (int dc = 0; dc < dC; ++dc) {
    double* blk = &hm->MAT[(size_t)dc * MC * MU];
    for (int b = 0; b < B; ++b) {
      // operator of edge b for device class dc: diagonal blocks g = 0..F-1 <-> true class dc*F + g (classes beyond C
      // pad the last group: identity transition, zero weight).  which = -1: P, k >= 0: w_c (P o N^k), w_c = class
      // probability when fused (the kernel then uses weight 1 per pass), else 1.
      auto block = [&](int g, int which) -> const double* {
        const int c = dc * F + g;
        if (c >= C) return nullptr;
        return which < 0 ? &hm->P[((size_t)c * B + b) * S2] : &hm->PN[(((size_t)c * B + b) * K + which) * S2];
      };
      auto weight = [&](int g, int which) { return (which >= 0 && F > 1) ? hm->probs[dc * F + g] : 1.0; };
      if (hm->taxon_of[b] >= 0) {
        const int tx = hm->taxon_of[b];
        for (int which = -1; which < K; ++which) {
          double* lt = blk + (size_t)(which < 0 ? NI + NI * K + tx : NI + NI * K + T + which * T + tx) * MU;
          for (int g = 0; g < F; ++g) {
            const double* M = block(g, which);
            for (int x = 0; x < S; ++x)
              for (int z = 0; z < S; ++z) {  // row = observed state z, column = device state X = (class g, state x): transposed.
                // Columns are stored state-in-tile major (X % 4) * (dS / 4) + X / 4: the values one lane of the
                // matrix-core layout needs from a row are contiguous.
                const int X = g * S + x, pos = (X % 4) * (dS / 4) + X / 4;
                lt[(size_t)z * dS + pos] = M ? weight(g, which) * M[(size_t)x * S + z] : (which < 0 && x == z ? 1.0 : 0.0);
              }
          }
        }
      } else {
        const int sl = hm->slot[b];
        for (int which = -1; which < K; ++which) {
          std::fill(dense.begin(), dense.end(), 0.0);
          for (int g = 0; g < F; ++g) {
            const double* M = block(g, which);
            for (int x = 0; x < S; ++x)
              for (int y = 0; y < S; ++y)
                dense[(size_t)(g * S + x) * dS + g * S + y] =
                    M ? weight(g, which) * M[(size_t)x * S + y] : (which < 0 && x == y ? 1.0 : 0.0);
          }
          pack_blocks(dS, dense.data(), blk + (size_t)(which < 0 ? sl : NI + sl * K + which) * MU);
        }
      }
    }
  }
  hm->cum_pi.resize(S);
  hm->cum_probs.resize(C);
  double cum = 0.0;
  for (int x = 0; x < S; ++x) { cum += hm->pi[x]; hm->cum_pi[x] = cum; }
  cum = 0.0;
  for (int c = 0; c < C; ++c) { cum += hm->probs[c]; hm->cum_probs[c] = cum; }
  *code = CMX_OK;
  return std::string();
}

}  // namespace cmx
