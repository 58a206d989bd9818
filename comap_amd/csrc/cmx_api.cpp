// C-ABI of the engine (include/comap_mi355x.h).  Host-side orchestration only: uploads the prepared model,
// owns the per-wave workspace, launches the HIP kernels.  There is no CPU compute path: without a HIP device every
// compute entry point fails with CMX_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/comap_mi355x.h"
#include "cmx_device.h"
#include "cmx_host_model.h"

using namespace cmx;

namespace {
thread_local std::string g_create_error;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;     // capacity the callers may use (the allocation is kGuardBytes longer under the guard)
  size_t logical = 0;   // guard only: what the last caller asked for; the canary sits at [logical, logical + kGuardBytes)
  bool guarded = false; // allocated with room for the canary (a buffer from before the guard was switched on is not)
};

// CMX_SCRATCH_GUARD=1 (or cmx_debug_scratch_guard(1)): every scratch buffer, per-wave workspace and temporary is
// allocated kGuardBytes longer, the bytes after what the caller asked for hold a canary, and scratch() (before it hands a
// buffer out again), cmx_synchronize, cmx_scratch_check and cmx_ctx_destroy verify it and name the buffer that was
// written past its end.  Round 3: a scratch buffer sized [nn][rep_ram] for a kernel that writes [nn][nrep * rep_ram] lived
// through a round of green tests on allocator slack.  Debug mode: every check synchronises the device.
constexpr size_t kGuardBytes = 4096;
constexpr int kGuardByte = 0xC5;
std::atomic<int> g_guard{-1};   // -1: not decided yet (environment read at the first context)
std::mutex g_guard_mu;
std::vector<std::string> g_guard_failures;   // buffers found trampled, in the order found (process-wide)
std::map<std::string, size_t> g_guard_shrink;   // test hook: logical size override per buffer name

bool guard_on() {
  int g = g_guard.load();
  if (g < 0) {
    const char* e = getenv("CMX_SCRATCH_GUARD");
    g = (e && e[0] == '1') ? 1 : 0;
    g_guard.store(g);
  }
  return g == 1;
}

void guard_record(const std::string& what) {
  std::lock_guard<std::mutex> lk(g_guard_mu);
  g_guard_failures.push_back(what);
  std::fprintf(stderr, "CMX_SCRATCH_GUARD: %s\n", what.c_str());
}

hipError_t guard_arm(void* base, size_t logical) {
  return hipMemset(static_cast<char*>(base) + logical, kGuardByte, kGuardBytes);
}

// true when the canary after `logical` bytes is intact; the device must be idle
bool guard_intact(const void* base, size_t logical, size_t* first_bad) {
  static thread_local std::vector<unsigned char> h(kGuardBytes);
  if (hipMemcpy(h.data(), static_cast<const char*>(base) + logical, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
  for (size_t i = 0; i < kGuardBytes; ++i)
    if (h[i] != (unsigned char)kGuardByte) { if (first_bad) *first_bad = i; return false; }
  return true;
}
}  // namespace

struct cmx_ctx {
  int device = 0;
  bool has_model = false;
  HostModel hm;
  DevModel dm{};
  Workspace ws{};       // null-distribution launches (persistent grid: 1 wave per SIMD on every CU)
  Workspace ws_obs{};   // observed-alignment launches: own slices, so both kinds can overlap on two streams
  int obs_blocks = 0;
  int cu_count = 0, waves = 0, grid_blocks = 0;
  size_t ws_bytes = 0;
  std::vector<void*> model_allocs;
  std::map<std::string, DevBuf> scratch;
  struct GuardedFixed { std::string name; void* p; size_t bytes; };
  std::vector<GuardedFixed> guarded_fixed;   // CMX_SCRATCH_GUARD: the per-wave workspaces, each with a canary after its last byte
  uint32_t* d_default_masks = nullptr;
  unsigned stat_mean_turn = 0;
  // host copies of asynchronously uploaded parameter blocks (mean vectors, MI bounds): the source of a hipMemcpyAsync must
  // outlive the copy, and the caller's array need not
  std::vector<double> param_host[8];
  bool leaf_rows_custom = false;   // the leaf operators' ambiguity rows were built from a caller's mask table
  bool map_average = true;         // nijt.average (cmx_set_mapping_options); false: the no-averaging mapping of cmx_variants.hip
  bool map_joint = true;           // nijt.joint; false: the ...Marginal variants of cmx_variants.hip
  // Mica's permutation test: host-side sources of its asynchronous table uploads (they must outlive the copies, also
  // when a later call fails), and which (L, taxa, shift) the fixed-point table F on the device was built for
  std::vector<long long> perm_dF_host, perm_F_host;
  std::vector<uint8_t> perm_tab_host;
  unsigned long long perm_F_L = 0;
  int perm_F_T = 0, perm_F_sh = -1;
  // cmx_intra_gram_prefetch_dev: the Gram blocks kept for the next cmx_intra_compact_range_dev with the same arguments
  struct GramKept { bool valid = false; int kind = 0; const double* counts = nullptr; size_t n = 0, ldc = 0, row_begin = 0, row_end = 0; const double* stat = nullptr; } gram_kept;
  const double *va_P = nullptr, *va_N1 = nullptr, *va_NC = nullptr;   // their operators, uploaded at first use
  const double *va_PN = nullptr, *va_pi = nullptr;                     // plain path: joint counts and frequencies, padded
  const int *va_first = nullptr, *va_next = nullptr;
  mutable std::string err;
};

#define HIP_TRY(ctx, expr)                                                                                   \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess) {                                                                                  \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                                        \
      return CMX_ERR_DEVICE;                                                                                 \
    }                                                                                                        \
  } while (0)

namespace {

cmx_status fail(cmx_ctx* ctx, cmx_status s, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return s;
}

template <class T>
cmx_status upload(cmx_ctx* ctx, const std::vector<T>& h, const T** d) {
  void* p = nullptr;
  const size_t bytes = sizeof(T) * (h.empty() ? 1 : h.size());
  HIP_TRY(ctx, hipMalloc(&p, bytes));
  ctx->model_allocs.push_back(p);
  if (!h.empty()) HIP_TRY(ctx, hipMemcpy(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
  *d = static_cast<const T*>(p);
  return CMX_OK;
}

// verify every guarded buffer of a context (device idle); the first trampled one is named in ctx->err
cmx_status guard_verify_all(cmx_ctx* ctx) {
  if (!guard_on()) return CMX_OK;
  cmx_status rc = CMX_OK;
  auto bad = [&](const std::string& name, size_t logical, size_t off) {
    const std::string msg = "buffer '" + name + "' was written past its end (" + std::to_string(logical) + " bytes asked for, first bad byte at +" + std::to_string(off) + ")";
    guard_record(msg);
    if (rc == CMX_OK) { ctx->err = "CMX_SCRATCH_GUARD: " + msg; rc = CMX_ERR_INTERNAL; }
  };
  size_t off = 0;
  for (auto& kv : ctx->scratch)
    if (kv.second.p && kv.second.guarded && !guard_intact(kv.second.p, kv.second.logical, &off)) {
      bad("scratch:" + kv.first, kv.second.logical, off);
      (void)guard_arm(kv.second.p, kv.second.logical);   // report an overflow once, not at every later check
    }
  for (auto& f : ctx->guarded_fixed)
    if (!guard_intact(f.p, f.bytes, &off)) {
      bad(f.name, f.bytes, off);
      (void)guard_arm(f.p, f.bytes);
    }
  return rc;
}

// grow-only named scratch buffers (allocated on first use, released with the context)
cmx_status scratch(cmx_ctx* ctx, const char* name, size_t bytes, void** out) {
  DevBuf& b = ctx->scratch[name];
  if (guard_on()) {
    // the previous user's canary is checked before the buffer is handed out again (it may move with the size asked for)
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (b.p && !b.guarded) {   // allocated before the guard was switched on: no room for a canary, start over
      HIP_TRY(ctx, hipFree(b.p));
      b.p = nullptr;
      b.bytes = 0;
    }
    size_t off = 0;
    if (b.p && !guard_intact(b.p, b.logical, &off)) {
      const std::string msg = std::string("buffer 'scratch:") + name + "' was written past its end (" + std::to_string(b.logical) +
                              " bytes asked for, first bad byte at +" + std::to_string(off) + ")";
      guard_record(msg);
      (void)guard_arm(b.p, b.logical);
      return fail(ctx, CMX_ERR_INTERNAL, "CMX_SCRATCH_GUARD: " + msg);
    }
    size_t logical = bytes ? bytes : 16;
    {
      std::lock_guard<std::mutex> lk(g_guard_mu);
      auto it = g_guard_shrink.find(name);
      if (it != g_guard_shrink.end() && it->second < logical) logical = it->second;   // test hook: pretend the caller asked for less
    }
    if (b.bytes < bytes || !b.p) {
      if (b.p) HIP_TRY(ctx, hipFree(b.p));
      b.p = nullptr;
      b.bytes = 0;
      HIP_TRY(ctx, hipMalloc(&b.p, (bytes ? bytes : 16) + kGuardBytes));
      b.bytes = bytes;
      b.guarded = true;
    }
    b.logical = logical;
    HIP_TRY(ctx, guard_arm(b.p, b.logical));
    *out = b.p;
    return CMX_OK;
  }
  if (b.bytes < bytes) {
    if (b.p) HIP_TRY(ctx, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    HIP_TRY(ctx, hipMalloc(&b.p, bytes ? bytes : 16));
    b.bytes = bytes;
    b.guarded = false;
  }
  *out = b.p;
  return CMX_OK;
}

struct TmpDev {  // RAII device temporaries for the host-pointer entry points
  std::vector<void*> ptrs;
  std::vector<size_t> sizes;   // guard only
  ~TmpDev() {
    if (!sizes.empty()) {
      (void)hipDeviceSynchronize();
      for (size_t i = 0; i < ptrs.size(); ++i) {
        size_t off = 0;
        if (!guard_intact(ptrs[i], sizes[i], &off))
          guard_record("temporary #" + std::to_string(i) + " of a host-pointer entry point was written past its end (" +
                       std::to_string(sizes[i]) + " bytes asked for, first bad byte at +" + std::to_string(off) + ")");
      }
    }
    for (void* p : ptrs) (void)hipFree(p);
  }
  hipError_t alloc(void** p, size_t bytes) {
    if (!bytes) bytes = 16;
    const bool g = guard_on();
    hipError_t e = hipMalloc(p, bytes + (g ? kGuardBytes : 0));
    if (e != hipSuccess) return e;
    ptrs.push_back(*p);
    if (g) {
      sizes.push_back(bytes);
      e = guard_arm(*p, bytes);
    }
    return e;
  }
};

cmx_status need_model(cmx_ctx* ctx) {
  if (!ctx) return CMX_ERR_INVALID;
  if (!ctx->has_model) return fail(ctx, CMX_ERR_INVALID, "this context was created without a model/tree");
  return CMX_OK;
}

}  // namespace

extern "C" {

const char* cmx_version(void) { return "comap_mi355x 0.1 (gfx950)"; }

const char* cmx_last_error(const cmx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

cmx_status cmx_ctx_create(const cmx_model* model, const cmx_tree* tree, int device, cmx_ctx** out) {
  if (!out) return CMX_ERR_INVALID;
  *out = nullptr;
  cmx_ctx* ctx = new cmx_ctx();
  ctx->device = device;
  auto bail = [&](cmx_status s) {
    g_create_error = ctx->err;
    cmx_ctx_destroy(ctx);
    return s;
  };
  if ((model == nullptr) != (tree == nullptr)) {
    ctx->err = "model and tree must be given together (both NULL creates a context for cmx_mi_columns only)";
    return bail(CMX_ERR_INVALID);
  }
  if (model) {
    int code = CMX_OK;
    std::string msg = build_host_model(model, tree, &ctx->hm, &code);
    if (!msg.empty()) {
      ctx->err = msg;
      return bail((cmx_status)code);
    }
    ctx->has_model = true;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    ctx->err = std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0") +
               "); this engine has no CPU path";
    return bail(CMX_ERR_DEVICE);
  }
  if (device < 0 || device >= ndev) {
    ctx->err = "device index out of range";
    return bail(CMX_ERR_INVALID);
  }
  auto dev_init = [&]() -> cmx_status {
    HIP_TRY(ctx, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(ctx, CMX_ERR_UNSUPPORTED, std::string("built for gfx950 only, device is ") + prop.gcnArchName);
    ctx->cu_count = prop.multiProcessorCount;
    std::vector<uint32_t> dm(256);
    for (int i = 0; i < 256; ++i) dm[i] = i < 32 ? (1u << i) : 0xffffffffu;
    const uint32_t* p = nullptr;
    cmx_status s = upload(ctx, dm, &p);
    if (s != CMX_OK) return s;
    ctx->d_default_masks = const_cast<uint32_t*>(p);
    if (!ctx->has_model) return CMX_OK;
    const HostModel& h = ctx->hm;
    DevModel& d = ctx->dm;
    d.S = h.dS; d.C = h.dC; d.S0 = h.S; d.C0 = h.C; d.fuse = h.fuse; d.K = h.K; d.nn = h.nn; d.B = h.B; d.T = h.T; d.NI = h.NI; d.NIW = h.NIW; d.NV = h.NV; d.root = h.root;
#define UP(field) if ((s = upload(ctx, h.field, &d.field)) != CMX_OK) return s
    UP(taxon_of); UP(parent);
    if (h.plain) {
      // alphabets other than 4 / 20 states: simulator tables and tree only; the sites are mapped by the plain kernels of
      // cmx_variants.hip on scratch buffers (map_plain below), no operator stream, no per-wave workspaces
      UP(simg); UP(simord);
      d.nsimg = (int)(h.simg.size() / 16);
      UP(eigV); UP(eigVi); UP(eigLam); UP(model_of); UP(blen);
      UP(CP); UP(CPG); UP(pi); UP(rates); UP(probs); UP(cum_pi); UP(cum_probs);
      return CMX_OK;
    }
    {
      const double* mat = nullptr;
      if ((s = upload(ctx, h.MAT, &mat)) != CMX_OK) return s;
      d.MAT = const_cast<double*>(mat);
    }
    d.MC = h.MC;
    {  // device copy of the operator stream: operator indices premultiplied to element offsets, and the first two entries
       // repeated after the last one so that "the entry two ops ahead" never needs a wrap test
      std::vector<int> ms(h.msched);
      const int unit = mat_unit(h.dS);
      for (size_t i = 0; i < ms.size(); i += 2) ms[i] *= unit;
      const size_t n2 = ms.size();
      for (size_t i = 0; i < 4; ++i) ms.push_back(ms[i % n2]);
      const int* dms = nullptr;
      if ((s = upload(ctx, ms, &dms)) != CMX_OK) return s;
      d.msched = dms;
      d.nmv = (int)(h.msched.size() / 2);
      d.msched_r = nullptr;
      d.nmv_r = 0;
      if (!h.msched_r.empty()) {   // the cherry-table walk's stream, prepared the same way
        std::vector<int> mr(h.msched_r);
        for (size_t i = 0; i < mr.size(); i += 2) mr[i] *= unit;
        const size_t nr = mr.size();
        for (size_t i = 0; i < 4; ++i) mr.push_back(mr[i % nr]);
        const int* dmr = nullptr;
        if ((s = upload(ctx, mr, &dmr)) != CMX_OK) return s;
        d.msched_r = dmr;
        d.nmv_r = (int)(h.msched_r.size() / 2);
      }
    }
    UP(nrec); UP(simg); UP(simord);
    d.nsimg = (int)(h.simg.size() / 16);
    {  // two zero entries (not prefetchable) after the last load: the kernel reads one entry ahead without a bounds test
      std::vector<int> ld(h.ldsched);
      ld.push_back(0);
      ld.push_back(0);
      const int* dld = nullptr;
      if ((s = upload(ctx, ld, &dld)) != CMX_OK) return s;
      d.ldsched = dld;
    }
    UP(eigV); UP(eigVi); UP(eigLam); UP(model_of); UP(blen);
    UP(CP); UP(CPG); UP(pi); UP(rates); UP(probs); UP(cum_pi); UP(cum_probs);
#undef UP
    // ambiguity rows of the leaf operators: default "every state compatible" until a call brings a mask table
    HIP_TRY(ctx, launch_extend_leaf_rows(d, nullptr, nullptr));
    HIP_TRY(ctx, hipDeviceSynchronize());
    ctx->leaf_rows_custom = false;
    // ambiguity masks default: code c >= S compatible with every state; fix the table for this S
    for (int i = 0; i < 256; ++i) dm[i] = i < h.S ? (1u << i) : ((h.S >= 32) ? 0xffffffffu : ((1u << h.S) - 1u));
    HIP_TRY(ctx, hipMemcpy(ctx->d_default_masks, dm.data(), sizeof(uint32_t) * 256, hipMemcpyHostToDevice));
    // per-wave workspaces: 1 wave per SIMD on every CU for the null; a quarter of that for observed alignments
    ctx->grid_blocks = ctx->cu_count * map_waves_per_simd(h.dS);   // 4-wave workgroups, that many per CU
    ctx->waves = ctx->grid_blocks * kWavesPerBlock;
    ctx->obs_blocks = std::max(1, ctx->grid_blocks / 4);
    auto alloc_ws = [&](Workspace* ws, size_t w, size_t* bytes) -> cmx_status {
      const size_t ks = (size_t)map_sites_per_wave(h.dS);   // sites per mapping wave
      const size_t bD = w * h.NIW * h.dS * ks * sizeof(double);
      // (rows of the per-site scratch arrays: the sites of a wave, or 64 lanes for the 48-site experiment layout)
      const size_t kr = map_ng(h.dS) == 3 ? 64 : ks;
      const size_t bC = w * 2 * h.B * h.K * kr * sizeof(double);
      const size_t bP = w * h.dC * h.B * h.K * kr * sizeof(double);
      const size_t bS = w * h.nn * ks, bA = w * h.T * ks;
      const bool g = guard_on();
      auto one = [&](const char* nm, void** p, size_t bytes) -> cmx_status {
        HIP_TRY(ctx, hipMalloc(p, bytes + (g ? kGuardBytes : 0)));
        if (g) {
          HIP_TRY(ctx, guard_arm(*p, bytes));
          ctx->guarded_fixed.push_back({std::string(ws == &ctx->ws ? "workspace:" : "workspace_obs:") + nm, *p, bytes});
        }
        return CMX_OK;
      };
      cmx_status sa;
      if ((sa = one("D", (void**)&ws->D, bD)) != CMX_OK) return sa;
      if ((sa = one("U", (void**)&ws->U, bD)) != CMX_OK) return sa;
      if ((sa = one("cnt", (void**)&ws->cnt, bC)) != CMX_OK) return sa;
      if ((sa = one("part", (void**)&ws->part, bP)) != CMX_OK) return sa;
      if ((sa = one("st", (void**)&ws->st, bS)) != CMX_OK) return sa;
      if ((sa = one("aln", (void**)&ws->aln, bA)) != CMX_OK) return sa;
      ws->waves = (int)w;
      *bytes += 2 * bD + bC + bP + bS + bA;
      return CMX_OK;
    };
    ctx->ws_bytes = 0;
    if ((s = alloc_ws(&ctx->ws, (size_t)ctx->waves, &ctx->ws_bytes)) != CMX_OK) return s;
    if ((s = alloc_ws(&ctx->ws_obs, (size_t)ctx->obs_blocks * kWavesPerBlock, &ctx->ws_bytes)) != CMX_OK) return s;
    return CMX_OK;
  };
  cmx_status s = dev_init();
  if (s != CMX_OK) return bail(s);
  *out = ctx;
  return CMX_OK;
}

void cmx_ctx_destroy(cmx_ctx* ctx) {
  if (!ctx) return;
  if (guard_on() && hipSetDevice(ctx->device) == hipSuccess && hipDeviceSynchronize() == hipSuccess)
    (void)guard_verify_all(ctx);   // destroy cannot fail: findings go to stderr and to cmx_debug_scratch_guard_failures
  for (void* p : ctx->model_allocs) (void)hipFree(p);
  for (auto& kv : ctx->scratch) if (kv.second.p) (void)hipFree(kv.second.p);
  for (Workspace* ws : {&ctx->ws, &ctx->ws_obs}) {
    if (ws->D) (void)hipFree(ws->D);
    if (ws->U) (void)hipFree(ws->U);
    if (ws->cnt) (void)hipFree(ws->cnt);
    if (ws->part) (void)hipFree(ws->part);
    if (ws->st) (void)hipFree(ws->st);
    if (ws->aln) (void)hipFree(ws->aln);
  }
  delete ctx;
}

cmx_status cmx_get_info(const cmx_ctx* ctx, cmx_info* info) {
  if (!ctx || !info) return CMX_ERR_INVALID;
  std::memset(info, 0, sizeof(*info));
  info->nstates = ctx->hm.S; info->nclasses = ctx->hm.C; info->ntypes = ctx->hm.K; info->nnodes = ctx->hm.nn;
  info->nbranches = ctx->hm.B; info->ntaxa = ctx->hm.T; info->ninternal = ctx->hm.NI;
  info->device = ctx->device; info->cu_count = ctx->cu_count; info->waves = ctx->waves;
  info->workspace_bytes = ctx->ws_bytes;
  info->device_states = ctx->hm.dS; info->device_classes = ctx->hm.dC;
  info->products_per_pass = (int32_t)ctx->hm.n_products; info->leaf_ops_per_pass = (int32_t)ctx->hm.n_leaf_ops;
  info->ws_loads_per_pass = (int32_t)ctx->hm.n_loads; info->ws_stores_per_pass = (int32_t)ctx->hm.n_stores;
  info->products_per_pass_null = (int32_t)ctx->hm.n_products_r; info->leaf_ops_per_pass_null = (int32_t)ctx->hm.n_leaf_ops_r;
  info->cherry_tables = ctx->hm.msched_r.empty() ? 0 : ctx->hm.ncherry;
  return CMX_OK;
}

cmx_status cmx_get_transition_matrices(const cmx_ctx* ctx, double* P) {
  if (!ctx || !P || !ctx->has_model) return CMX_ERR_INVALID;
  std::memcpy(P, ctx->hm.P.data(), sizeof(double) * ctx->hm.P.size());
  return CMX_OK;
}

cmx_status cmx_debug_walk(const cmx_model* model, const cmx_tree* tree, int32_t* nrec, size_t nrec_cap, size_t* nrec_n,
                          int32_t* ldsched, size_t ld_cap, size_t* ld_n, int32_t* msched, size_t m_cap, size_t* m_n,
                          int32_t* slot_of_node, uint64_t* stats /*[7]: loads, stores, products, leaf ops per pass; products, leaf ops of the cherry-table walk, cherries with tables*/) {
  HostModel hm;
  int code = CMX_OK;
  const std::string msg = build_host_model(model, tree, &hm, &code);
  if (!msg.empty()) {
    g_create_error = msg;
    return (cmx_status)code;
  }
  if (hm.nrec.size() > nrec_cap || hm.ldsched.size() > ld_cap || hm.msched.size() > m_cap) {
    g_create_error = "cmx_debug_walk: buffers too small";
    return CMX_ERR_INVALID;
  }
  std::memcpy(nrec, hm.nrec.data(), hm.nrec.size() * sizeof(int32_t));
  std::memcpy(ldsched, hm.ldsched.data(), hm.ldsched.size() * sizeof(int32_t));
  std::memcpy(msched, hm.msched.data(), hm.msched.size() * sizeof(int32_t));
  *nrec_n = hm.nrec.size(); *ld_n = hm.ldsched.size(); *m_n = hm.msched.size();
  if (slot_of_node) std::memcpy(slot_of_node, hm.slot.data(), hm.slot.size() * sizeof(int32_t));
  if (stats) {
    stats[0] = hm.n_loads; stats[1] = hm.n_stores; stats[2] = hm.n_products; stats[3] = hm.n_leaf_ops;
    stats[4] = hm.n_products_r; stats[5] = hm.n_leaf_ops_r; stats[6] = hm.msched_r.empty() ? 0 : (uint64_t)hm.ncherry;
  }
  return CMX_OK;
}

cmx_status cmx_synchronize(cmx_ctx* ctx) {
  if (!ctx) return CMX_ERR_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  return guard_verify_all(ctx);
}

cmx_status cmx_scratch_check(cmx_ctx* ctx) {
  if (!ctx) return CMX_ERR_INVALID;
  if (!guard_on()) return fail(ctx, CMX_ERR_UNSUPPORTED, "the scratch guard is off (CMX_SCRATCH_GUARD=1 or cmx_debug_scratch_guard(1) before the context is created)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  return guard_verify_all(ctx);
}

int cmx_debug_scratch_guard(int on) {
  const int was = guard_on() ? 1 : 0;
  if (on >= 0) g_guard.store(on ? 1 : 0);
  return was;
}

size_t cmx_debug_scratch_guard_failures(char* buf, size_t cap, int clear) {
  std::lock_guard<std::mutex> lk(g_guard_mu);
  const size_t n = g_guard_failures.size();
  if (buf && cap) {
    std::string all;
    for (const std::string& f : g_guard_failures) { all += f; all += '\n'; }
    std::snprintf(buf, cap, "%s", all.c_str());
  }
  if (clear) g_guard_failures.clear();
  return n;
}

void cmx_debug_scratch_shrink(const char* name, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_guard_mu);
  if (!name) { g_guard_shrink.clear(); return; }
  if (bytes == 0) g_guard_shrink.erase(name);
  else g_guard_shrink[name] = bytes;
}

// nijt.average = no (cmx_set_mapping_options): counts and norms of the sites just mapped are replaced by those of
// computeSubstitutionVectorsNoAveraging (cmx_variants.hip); likelihood, posterior rate and rate class stay.
// full_grid names the caller as in map_sites_impl: the engine's own null / clustering / candidate pipelines (true) and the
// public observed-alignment mapping (false) may run on two streams at once, so each has its own scratch -- the averaged
// path keeps ws and ws_obs apart for the same reason.
// S x S matrices padded with zeros to SP x SP (the plain path's kernels run at kPlainStates states)
static std::vector<double> pad_mats(const std::vector<double>& m, int S, int SP) {
  if (S == SP) return m;
  const size_t nm = m.size() / ((size_t)S * S);
  std::vector<double> o(nm * (size_t)SP * SP, 0.0);
  for (size_t q = 0; q < nm; ++q)
    for (int x = 0; x < S; ++x)
      for (int y = 0; y < S; ++y) o[(q * SP + x) * SP + y] = m[(q * S + x) * S + y];
  return o;
}

// plain: the caller is map_plain (alphabets other than 4 / 20 states): every mapping option, the default one included, and
// the site scalars come from these kernels
static cmx_status map_variant(cmx_ctx* ctx, const uint8_t* d_aln, size_t nsites, size_t ld, const uint32_t* d_masks,
                              double* d_counts, size_t ldc, double* d_norm, void* stream, bool full_grid,
                              double* d_logL = nullptr, double* d_post_rate = nullptr, int32_t* d_rate_class = nullptr) {
  const HostModel& h = ctx->hm;
  const bool scalars = h.plain && (d_logL || d_post_rate || d_rate_class);
  if (!h.plain && ((ctx->map_average && ctx->map_joint) || (!d_counts && !d_norm))) return CMX_OK;
  if (h.plain && !d_counts && !d_norm && !scalars) return CMX_OK;
  const int SD = h.plain ? kPlainStates : h.S;   // device states
  cmx_status s;
  if (!ctx->va_P) {
    if ((s = upload(ctx, pad_mats(h.P, h.S, SD), &ctx->va_P)) != CMX_OK) return s;
    if ((s = upload(ctx, pad_mats(h.N1, h.S, SD), &ctx->va_N1)) != CMX_OK) return s;
    if ((s = upload(ctx, pad_mats(h.NC, h.S, SD), &ctx->va_NC)) != CMX_OK) return s;
    if ((s = upload(ctx, h.first_child, &ctx->va_first)) != CMX_OK) return s;
    if ((s = upload(ctx, h.next_sib, &ctx->va_next)) != CMX_OK) return s;
    if (h.plain) {
      std::vector<double> pi(SD, 0.0);
      std::copy(h.pi.begin(), h.pi.end(), pi.begin());
      if ((s = upload(ctx, pad_mats(h.PN, h.S, SD), &ctx->va_PN)) != CMX_OK) return s;
      if ((s = upload(ctx, pi, &ctx->va_pi)) != CMX_OK) return s;
    }
  }
  const bool want_counts = d_counts || d_norm;
  if (!d_counts && want_counts) {   // only the norms were asked for: they still need the counts
    if ((s = scratch(ctx, full_grid ? "va_counts_null" : "va_counts_obs", sizeof(double) * (size_t)h.B * h.K * nsites, (void**)&d_counts)) != CMX_OK) return s;
    ldc = nsites;
  }
  NoAvgArgs a{};
  a.S = SD; a.Sreal = h.S; a.C = h.C; a.K = h.K; a.nn = h.nn; a.B = h.B; a.root = h.root;
  a.mode = ctx->map_joint ? (ctx->map_average ? kVariantJoint : kVariantNoAvg) : (ctx->map_average ? kVariantMarginal : kVariantNoAvgMarginal);
  a.first_child = ctx->va_first; a.next_sib = ctx->va_next; a.taxon_of = ctx->dm.taxon_of; a.parent = ctx->dm.parent;
  a.P = ctx->va_P; a.N1 = ctx->va_N1; a.NC = ctx->va_NC; a.PN = ctx->va_PN; a.pi = h.plain ? ctx->va_pi : ctx->dm.pi; a.probs = ctx->dm.probs;
  a.rates = ctx->dm.rates; a.logL = d_logL; a.post_rate = d_post_rate; a.rate_class = d_rate_class;
  a.masks = d_masks; a.aln = d_aln; a.ld = ld;
  // sites per pass: per-node vectors of a pass stay under 1 GiB
  const size_t per_site = sizeof(double) * noavg_scratch_doubles(SD, h.C, h.nn, 1);
  a.chunk = std::max<size_t>(256, std::min<size_t>(nsites, ((size_t)1 << 30) / per_site / 256 * 256));
  a.counts = d_counts; a.ldc = ldc;
  double* buf;
  if ((s = scratch(ctx, full_grid ? "va_nodes_null" : "va_nodes_obs", sizeof(double) * noavg_scratch_doubles(SD, h.C, h.nn, a.chunk), (void**)&buf)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_map_noavg(a, nsites, buf, d_norm, (hipStream_t)stream));
  return CMX_OK;
}

// ------------------------------------------------------------------------------------------------ mapping
// full_grid: use the whole-chip workspace of the null launches instead of the quarter-chip slice reserved for
// observed alignments (which exists so that a caller can overlap the observed mapping with cmx_null_intra_dev on a
// second stream).  Only the engine's own simulate -> map pipelines (inter null, clustering null, candidate groups)
// ask for it: they are blocking calls on the null stream and map hundreds of thousands of simulated sites.
static cmx_status map_sites_impl(cmx_ctx* ctx, const uint8_t* d_aln, size_t nsites, size_t ld, const uint32_t* d_masks,
                                 double* d_counts, size_t ldc, double* d_logL, double* d_post_rate, int32_t* d_rate_class,
                                 double* d_norm, void* stream, bool full_grid) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (!d_aln || nsites == 0 || ld < nsites) return fail(ctx, CMX_ERR_INVALID, "cmx_map_sites: bad alignment arguments");
  if (d_counts && ldc < nsites) return fail(ctx, CMX_ERR_INVALID, "cmx_map_sites: ldc < nsites");
  if (d_counts && d_counts == ctx->gram_kept.counts) ctx->gram_kept.valid = false;   // the vectors the kept Gram blocks were made from are rewritten
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->hm.plain) {
    // alphabets other than 4 / 20 states (codon models): likelihood, rates and every mapping option from the plain kernels.
    // No mask table: every code >= nstates is an unknown.
    if (d_masks) return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_map_sites: no ambiguity table for alphabets other than 4 / 20 states (codes >= nstates are unknowns)");
    return map_variant(ctx, d_aln, nsites, ld, d_masks, d_counts, ldc, d_norm, stream, full_grid, d_logL, d_post_rate, d_rate_class);
  }
  const int max_blocks = full_grid ? ctx->grid_blocks : ctx->obs_blocks;
  MapArgs a{};
  a.m = ctx->dm; a.ws = full_grid ? ctx->ws : ctx->ws_obs;
  a.aln = d_aln; a.ld = ld; a.nsites = nsites;
  // ambiguity ids S .. S+max_ambig(S)-1: rebuild the extra rows of the leaf operators when the table changes
  if (d_masks || ctx->leaf_rows_custom) {
    HIP_TRY(ctx, launch_extend_leaf_rows(ctx->dm, d_masks, (hipStream_t)stream));
    ctx->leaf_rows_custom = d_masks != nullptr;
  }
  a.counts = d_counts; a.ldc = ldc; a.logL = d_logL; a.post_rate = d_post_rate; a.rate_class = d_rate_class;
  a.norm = d_norm;
  size_t ks = (size_t)map_sites_per_wave(ctx->hm.dS);
  size_t nblocks = (nsites + ks - 1) / ks;
  const size_t obs_waves = (size_t)max_blocks * kWavesPerBlock;
  if (nblocks * (size_t)ctx->hm.dC <= obs_waves && ctx->hm.dC > 1 && map_ng(ctx->hm.dS) != 3) {   // (48-site experiment builds: no class split)
    // small alignment: one (site block, class) per wave, classes summed by a second kernel (same arithmetic order)
    // Proteins, when even that leaves most of the chip idle: 16-site blocks (one site group per wave) -- four times the
    // tasks, a quarter of the matrix work per operator op, and a wave's slices of the workspaces are a quarter as large,
    // so 4 * obs_waves of them fit the same allocation
    if (ctx->hm.dS == 20 && ctx->hm.fuse == 1 && ks == 64 && ((nsites + 15) / 16) * (size_t)ctx->hm.dC <= 4 * obs_waves) {
      ks = 16;
      nblocks = (nsites + ks - 1) / ks;
    }
    a.split_sites = (int)ks;
    const size_t ntasks = nblocks * (size_t)ctx->hm.dC, BK = (size_t)ctx->hm.B * ctx->hm.K;
    // (per caller, like the workspaces: the public observed mapping and the engine's own pipelines may overlap on two streams)
    if ((s = scratch(ctx, full_grid ? "split_part_null" : "split_part_obs", sizeof(double) * ntasks * BK * ks, (void**)&a.split_part)) != CMX_OK) return s;
    if ((s = scratch(ctx, full_grid ? "split_lc_null" : "split_lc_obs", sizeof(double) * 4 * ntasks * ks, (void**)&a.split_lc)) != CMX_OK) return s;
    const int grid = (int)((ntasks + kWavesPerBlock - 1) / kWavesPerBlock);
    HIP_TRY(ctx, launch_map(a, kModeObservedSplit, grid, (hipStream_t)stream));
    HIP_TRY(ctx, launch_map_finalize(a, (hipStream_t)stream));
    return map_variant(ctx, d_aln, nsites, ld, d_masks, d_counts, ldc, d_norm, stream, full_grid);
  }
  const size_t blocks_needed = (nblocks + kWavesPerBlock - 1) / kWavesPerBlock;
  const int grid = (int)std::min<size_t>(blocks_needed, (size_t)max_blocks);
  HIP_TRY(ctx, launch_map(a, kModeObserved, grid, (hipStream_t)stream));
  return map_variant(ctx, d_aln, nsites, ld, d_masks, d_counts, ldc, d_norm, stream, full_grid);
}

cmx_status cmx_map_sites_dev(cmx_ctx* ctx, const uint8_t* d_aln, size_t nsites, size_t ld, const uint32_t* d_masks,
                             double* d_counts, size_t ldc, double* d_logL, double* d_post_rate, int32_t* d_rate_class,
                             double* d_norm, void* stream) {
  return map_sites_impl(ctx, d_aln, nsites, ld, d_masks, d_counts, ldc, d_logL, d_post_rate, d_rate_class, d_norm, stream, false);
}

cmx_status cmx_map_sites(cmx_ctx* ctx, const uint8_t* aln, size_t nsites, size_t ld, const uint32_t* masks,
                         size_t nmasks, double* counts, double* logL, double* post_rate, int32_t* rate_class,
                         double* norm) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (!aln || nsites == 0 || ld < nsites) return fail(ctx, CMX_ERR_INVALID, "cmx_map_sites: bad alignment arguments");
  const HostModel& h = ctx->hm;
  if (masks && h.plain)
    return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_map_sites: no ambiguity table for alphabets other than 4 / 20 states (codes >= nstates are unknowns)");
  if (masks && nmasks > (size_t)(h.S + max_ambig(h.S)))
    return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_map_sites: at most " + std::to_string(max_ambig(h.S)) +
                                              " ambiguity ids (codes >= nstates) are supported for this alphabet");
  // every code must be a state or a known mask (the reference throws BadCharException at alignment parsing)
  for (int t = 0; t < h.T; ++t)
    for (size_t i = 0; i < nsites; ++i) {
      const unsigned c = aln[(size_t)t * ld + i];
      if (c >= (unsigned)h.S && masks && c >= nmasks) return fail(ctx, CMX_ERR_INVALID, "cmx_map_sites: alignment code without a mask");
    }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  TmpDev tmp;
  uint8_t* d_aln = nullptr;
  uint32_t* d_masks = nullptr;
  double *d_counts = nullptr, *d_logL = nullptr, *d_pr = nullptr, *d_norm = nullptr;
  int32_t* d_rc = nullptr;
  const size_t BK = (size_t)h.B * h.K;
  HIP_TRY(ctx, tmp.alloc((void**)&d_aln, (size_t)h.T * nsites));
  HIP_TRY(ctx, hipMemcpy2D(d_aln, nsites, aln, ld, nsites, h.T, hipMemcpyHostToDevice));
  if (masks) {
    std::vector<uint32_t> mk(256, h.S >= 32 ? 0xffffffffu : ((1u << h.S) - 1u));
    for (size_t i = 0; i < nmasks && i < 256; ++i) mk[i] = masks[i];
    HIP_TRY(ctx, tmp.alloc((void**)&d_masks, 256 * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMemcpy(d_masks, mk.data(), 256 * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  if (counts) HIP_TRY(ctx, tmp.alloc((void**)&d_counts, BK * nsites * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_logL, nsites * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_pr, nsites * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_norm, nsites * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_rc, nsites * sizeof(int32_t)));
  s = cmx_map_sites_dev(ctx, d_aln, nsites, nsites, d_masks, d_counts, nsites, d_logL, d_pr, d_rc, d_norm, nullptr);
  if (s != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  if (counts) {  // branch-major [B*K][N] -> site-major [N][B][K] (reference layout mapping[i][b][k])
    std::vector<double> bm(BK * nsites);
    HIP_TRY(ctx, hipMemcpy(bm.data(), d_counts, bm.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t r = 0; r < BK; ++r)
      for (size_t i = 0; i < nsites; ++i) counts[i * BK + r] = bm[r * nsites + i];
  }
  if (logL) HIP_TRY(ctx, hipMemcpy(logL, d_logL, nsites * sizeof(double), hipMemcpyDeviceToHost));
  if (post_rate) HIP_TRY(ctx, hipMemcpy(post_rate, d_pr, nsites * sizeof(double), hipMemcpyDeviceToHost));
  if (norm) HIP_TRY(ctx, hipMemcpy(norm, d_norm, nsites * sizeof(double), hipMemcpyDeviceToHost));
  if (rate_class) HIP_TRY(ctx, hipMemcpy(rate_class, d_rc, nsites * sizeof(int32_t), hipMemcpyDeviceToHost));
  return CMX_OK;
}

// nijt.average / nijt.joint of CoETools.cpp:393-394 ("really for benchmarking only" there)
cmx_status cmx_set_mapping_options(cmx_ctx* ctx, int average, int joint) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  ctx->map_average = average != 0;
  ctx->map_joint = joint != 0;
  return CMX_OK;
}

// the simulator's counter layout (cmx_kernels.hip philox_uniform): 47 bits of simulated-site index, 17 bits of draw index
static cmx_status rng_range(cmx_ctx* ctx, uint64_t g_end, const char* who) {
  if (g_end > (1ull << 47) || (uint64_t)ctx->hm.nn + 2 > (1ull << 17))
    return fail(ctx, CMX_ERR_UNSUPPORTED, std::string(who) + ": simulated-site index beyond 2^47 or more than 2^17 - 2 nodes");
  return CMX_OK;
}

// ------------------------------------------------------------------------------------------------ simulator
cmx_status cmx_simulate_dev(cmx_ctx* ctx, uint64_t seed, uint64_t g0, size_t n, uint8_t* d_aln, size_t ld, int32_t* d_classes,
                            void* stream) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (!d_aln || n == 0 || ld < n) return fail(ctx, CMX_ERR_INVALID, "cmx_simulate: bad arguments");
  if ((s = rng_range(ctx, g0 + n, "cmx_simulate")) != CMX_OK) return s;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // node states of the n sites: a scratch of their own per stream would be needed to overlap two simulations of one
  // context; calls on one context are serialised by the caller (header)
  uint8_t* d_st;
  int32_t* d_cls = d_classes;
  if ((s = scratch(ctx, "sim_states", (size_t)ctx->hm.nn * ld, (void**)&d_st)) != CMX_OK) return s;
  if (!d_cls && (s = scratch(ctx, "sim_classes", sizeof(int32_t) * n, (void**)&d_cls)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_simulate(ctx->dm, seed, g0, n, d_aln, ld, d_cls, d_st, (hipStream_t)stream));
  return CMX_OK;
}

cmx_status cmx_simulate(cmx_ctx* ctx, uint64_t seed, uint64_t g0, size_t n, uint8_t* aln_out, int32_t* classes_out) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (!aln_out || n == 0) return fail(ctx, CMX_ERR_INVALID, "cmx_simulate: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const HostModel& h = ctx->hm;
  TmpDev tmp;
  uint8_t* d_aln = nullptr;
  int32_t* d_cls = nullptr;
  HIP_TRY(ctx, tmp.alloc((void**)&d_aln, (size_t)h.T * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_cls, n * sizeof(int32_t)));
  if ((s = cmx_simulate_dev(ctx, seed, g0, n, d_aln, n, d_cls, nullptr)) != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(aln_out, d_aln, (size_t)h.T * n, hipMemcpyDeviceToHost));
  if (classes_out) HIP_TRY(ctx, hipMemcpy(classes_out, d_cls, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  return CMX_OK;
}

cmx_status cmx_simulate_continuous_dev(cmx_ctx* ctx, uint64_t seed, uint64_t g0, size_t n, double gamma_alpha, double p_invariant,
                                       uint8_t* d_aln, size_t ld, double* d_rates, void* stream) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (!d_aln || n == 0 || ld < n || !(gamma_alpha > 0.0) || !(p_invariant >= 0.0 && p_invariant < 1.0))
    return fail(ctx, CMX_ERR_INVALID, "cmx_simulate_continuous: bad arguments (alpha > 0, 0 <= p_invariant < 1)");
  if ((s = rng_range(ctx, g0 + n, "cmx_simulate_continuous")) != CMX_OK) return s;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint8_t* d_st;
  if ((s = scratch(ctx, "sim_states", (size_t)ctx->hm.nn * ld, (void**)&d_st)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_simulate_continuous(ctx->dm, seed, g0, n, gamma_alpha, p_invariant, d_aln, ld, d_rates, d_st, (hipStream_t)stream));
  return CMX_OK;
}

cmx_status cmx_simulate_continuous(cmx_ctx* ctx, uint64_t seed, uint64_t g0, size_t n, double gamma_alpha, double p_invariant,
                                   uint8_t* aln_out, double* rates_out) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (!aln_out || n == 0) return fail(ctx, CMX_ERR_INVALID, "cmx_simulate_continuous: bad arguments (alpha > 0, 0 <= p_invariant < 1)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const HostModel& h = ctx->hm;
  TmpDev tmp;
  uint8_t* d_aln = nullptr;
  double* d_r = nullptr;
  HIP_TRY(ctx, tmp.alloc((void**)&d_aln, (size_t)h.T * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_r, n * sizeof(double)));
  if ((s = cmx_simulate_continuous_dev(ctx, seed, g0, n, gamma_alpha, p_invariant, d_aln, n, d_r, nullptr)) != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(aln_out, d_aln, (size_t)h.T * n, hipMemcpyDeviceToHost));
  if (rates_out) HIP_TRY(ctx, hipMemcpy(rates_out, d_r, n * sizeof(double), hipMemcpyDeviceToHost));
  return CMX_OK;
}

// ------------------------------------------------------------------------------------------------ pair statistics
static cmx_status check_kind(cmx_ctx* ctx, int kind) {
  if (kind < CMX_STAT_CORRELATION || kind > CMX_STAT_SCALAR_PRODUCT) return fail(ctx, CMX_ERR_INVALID, "unknown statistic kind");
  return CMX_OK;
}
// CorrectedCorrelation: params = the two mean vectors [2][B] in host memory -> device copy (null for other kinds)
// CorrectedCorrelation's two mean vectors on the device for an asynchronous call: copied on the CALLER's stream (a
// blocking copy on the null stream does not order against torch's non-blocking streams) into one of eight rotating
// buffers, so that a later call cannot overwrite vectors a kernel still in flight is reading.
static cmx_status stat_mean_vectors(cmx_ctx* ctx, int kind, const double* params, const double** d_mean, void* stream) {
  *d_mean = nullptr;
  if (kind != CMX_STAT_CORRECTED_CORRELATION) return CMX_OK;
  if (!params) return fail(ctx, CMX_ERR_INVALID, "CorrectedCorrelation needs its mean vectors: params = [2][nbranches]");
  void* p = nullptr;
  const unsigned turn = ctx->stat_mean_turn++ & 7;
  const std::string name = "stat_mean" + std::to_string(turn);
  cmx_status s = scratch(ctx, name.c_str(), sizeof(double) * 2 * ctx->hm.B, &p);
  if (s != CMX_OK) return s;
  ctx->param_host[turn].assign(params, params + 2 * (size_t)ctx->hm.B);
  HIP_TRY(ctx, hipMemcpyAsync(p, ctx->param_host[turn].data(), sizeof(double) * 2 * ctx->hm.B, hipMemcpyHostToDevice, (hipStream_t)stream));
  *d_mean = static_cast<const double*>(p);
  return CMX_OK;
}

// ---- DiscreteMutualInformationStatistic with a bounds vector (CMX_STAT_DISCRETE_MI_BOUNDS, cmx_stat_mi.hip):
// params = [nbounds, b_0 .. b_{nbounds-1}]
struct MiBounds {
  int nb = 0;
  const double* d_bounds = nullptr;
};
static cmx_status mi_bounds(cmx_ctx* ctx, const double* params, MiBounds* out, void* stream) {
  if (!params) return fail(ctx, CMX_ERR_INVALID, "DiscreteMI with bounds: params = [nbounds, b_0 .. b_{nbounds-1}] is required");
  const double nbd = params[0];
  if (!(nbd >= 2.0) || nbd > 65535.0 || nbd != std::floor(nbd))
    return fail(ctx, CMX_ERR_INVALID, "DiscreteMI with bounds: the number of bounds must be an integer in 2 .. 65535");
  const int nb = (int)nbd;
  for (int i = 0; i < nb; ++i) {
    if (params[1 + i] != params[1 + i]) return fail(ctx, CMX_ERR_INVALID, "DiscreteMI with bounds: a bound is NaN");
    if (i && params[1 + i] < params[i])   // Domain::Domain(const Vdouble&) throws for decreasing bounds (Domain.cpp:62-72)
      return fail(ctx, CMX_ERR_INVALID, "DiscreteMI with bounds: bound " + std::to_string(i) + " is < to bound " + std::to_string(i - 1));
  }
  if (ctx->hm.B > 4096) return fail(ctx, CMX_ERR_UNSUPPORTED, "DiscreteMI with bounds: at most 4096 branches (the joint table of a pair lives in LDS)");
  void* p = nullptr;
  const unsigned turn = ctx->stat_mean_turn++ & 7;
  const std::string name = "mi_bounds" + std::to_string(turn);
  cmx_status s = scratch(ctx, name.c_str(), sizeof(double) * nb, &p);
  if (s != CMX_OK) return s;
  ctx->param_host[turn].assign(params + 1, params + 1 + nb);
  HIP_TRY(ctx, hipMemcpyAsync(p, ctx->param_host[turn].data(), sizeof(double) * nb, hipMemcpyHostToDevice, (hipStream_t)stream));
  out->nb = nb;
  out->d_bounds = static_cast<const double*>(p);
  return CMX_OK;
}
// class words of n sites (branch-major [B][ldx]) + per-site out-of-range flags, in scratch buffers named after `tag`
static cmx_status mi_classify(cmx_ctx* ctx, const MiBounds& mb, const double* d_counts, size_t n, size_t ldc, const char* tag,
                              uint32_t** cls, uint8_t** bad, size_t* ldx, void* stream) {
  const HostModel& h = ctx->hm;
  *ldx = (n + 15) / 16 * 16;
  cmx_status s;
  if ((s = scratch(ctx, (std::string("mi_cls_") + tag).c_str(), sizeof(uint32_t) * (size_t)h.B * *ldx, (void**)cls)) != CMX_OK) return s;
  if ((s = scratch(ctx, (std::string("mi_bad_") + tag).c_str(), n, (void**)bad)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_mi_classify(d_counts, n, ldc, h.B, h.K, mb.d_bounds, mb.nb, *cls, *ldx, *bad, (hipStream_t)stream));
  return CMX_OK;
}

cmx_status cmx_pair_stats_dev(cmx_ctx* ctx, int kind, const double* params, const double* d_counts1, size_t n1,
                              size_t ld1, const double* d_counts2, size_t n2, size_t ld2, double* d_out, size_t ldo,
                              void* stream) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if ((s = check_kind(ctx, kind)) != CMX_OK) return s;
  const bool intra = d_counts2 == nullptr;
  if (intra) { n2 = n1; ld2 = ld1; }
  if (!d_counts1 || !d_out || n1 == 0 || n2 == 0 || ld1 < n1 || ld2 < n2 || ldo < n2)
    return fail(ctx, CMX_ERR_INVALID, "cmx_pair_stats: bad arguments");
  const HostModel& h = ctx->hm;
  if (h.B < 2) return fail(ctx, CMX_ERR_INVALID, "cmx_pair_stats: need at least two branches");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  if (kind == CMX_STAT_DISCRETE_MI_BOUNDS) {
    MiBounds mb;
    uint32_t *c1, *c2;
    uint8_t *b1, *b2;
    size_t lx1, lx2;
    if ((s = mi_bounds(ctx, params, &mb, stream)) != CMX_OK) return s;
    if ((s = mi_classify(ctx, mb, d_counts1, n1, ld1, "1", &c1, &b1, &lx1, stream)) != CMX_OK) return s;
    if (intra) { c2 = c1; b2 = b1; lx2 = lx1; }
    else if ((s = mi_classify(ctx, mb, d_counts2, n2, ld2, "2", &c2, &b2, &lx2, stream)) != CMX_OK) return s;
    HIP_TRY(ctx, launch_mi_pairs_block(h.B, c1, b1, n1, lx1, c2, b2, n2, lx2, intra ? 1 : 0, d_out, ldo, 0, st));
    return CMX_OK;
  }
  const double param = (kind == CMX_STAT_DISCRETE_MI) ? (params ? params[0] : 0.99) : 0.0;
  const double* d_mean = nullptr;
  if ((s = stat_mean_vectors(ctx, kind, params, &d_mean, stream)) != CMX_OK) return s;
  const int gk = kind == CMX_STAT_CORRECTED_CORRELATION ? CMX_STAT_CORRELATION : kind;   // same Gram + epilogue
  const int Bp = (h.B + 3) / 4 * 4;
  const size_t ldx1 = (n1 + 15) / 16 * 16, ldx2 = (n2 + 15) / 16 * 16;
  double *X1, *s1, *r1, *X2, *s2, *r2;
  if ((s = scratch(ctx, "pair_X1", sizeof(double) * Bp * ldx1, (void**)&X1)) != CMX_OK) return s;
  if ((s = scratch(ctx, "pair_s1", sizeof(double) * n1, (void**)&s1)) != CMX_OK) return s;
  if ((s = scratch(ctx, "pair_r1", sizeof(double) * n1, (void**)&r1)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_pair_prep(gk, param, d_counts1, n1, ld1, h.B, h.K, X1, ldx1, Bp, s1, r1, d_mean, st));
  if (intra) { X2 = X1; s2 = s1; r2 = r1; }
  else {
    if ((s = scratch(ctx, "pair_X2", sizeof(double) * Bp * ldx2, (void**)&X2)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_s2", sizeof(double) * n2, (void**)&s2)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_r2", sizeof(double) * n2, (void**)&r2)) != CMX_OK) return s;
    HIP_TRY(ctx, launch_pair_prep(gk, param, d_counts2, n2, ld2, h.B, h.K, X2, ldx2, Bp, s2, r2, d_mean ? d_mean + h.B : nullptr, st));
  }
  HIP_TRY(ctx, launch_pair_gram(gk, h.B, Bp, X1, s1, r1, n1, ldx1, X2, s2, r2, n2, intra ? ldx1 : ldx2, intra ? 1 : 0,
                                d_out, ldo, st));
  return CMX_OK;
}

static void to_branch_major(const double* sm, size_t n, size_t BK, std::vector<double>* bm) {
  bm->resize(BK * n);
  for (size_t i = 0; i < n; ++i)
    for (size_t r = 0; r < BK; ++r) (*bm)[r * n + i] = sm[i * BK + r];
}

// AnalysisTools::compute*Matrix (AnalysisTools.cpp:102-339): the same operand preparation, Gram kernel and epilogues as
// the pair statistics, with the vector length as the "number of branches" and one "substitution type"
cmx_status cmx_vector_matrix(cmx_ctx* ctx, int kind, size_t dim, const double* v1, size_t n1, const double* v2, size_t n2,
                             int independent, double* out) {
  if (!ctx) return CMX_ERR_INVALID;
  if (kind != CMX_STAT_SCALAR_PRODUCT && kind != CMX_STAT_COSINUS && kind != CMX_STAT_CORRELATION && kind != CMX_STAT_COVARIANCE)
    return fail(ctx, CMX_ERR_INVALID, "cmx_vector_matrix: kind must be scalar product, cosinus, correlation or covariance");
  const bool one = v2 == nullptr;
  if (one) n2 = n1;
  if (!v1 || !out || n1 == 0 || n2 == 0 || dim == 0 || dim > 0x7fffffffull) return fail(ctx, CMX_ERR_INVALID, "cmx_vector_matrix: bad arguments");
  if (independent && (one || n1 != n2))
    return fail(ctx, CMX_ERR_INVALID, "cmx_vector_matrix: when performing independant comparisons, the two datasets must have the same length");
  if ((kind == CMX_STAT_CORRELATION || kind == CMX_STAT_COVARIANCE) && dim < 2)
    return fail(ctx, CMX_ERR_INVALID, "cmx_vector_matrix: correlation / covariance need vectors of at least two elements");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int B = (int)dim, Bp = (B + 3) / 4 * 4;
  TmpDev tmp;
  std::vector<double> bm;
  double *d1 = nullptr, *d2 = nullptr, *d_out = nullptr;
  to_branch_major(v1, n1, dim, &bm);
  HIP_TRY(ctx, tmp.alloc((void**)&d1, bm.size() * sizeof(double)));
  HIP_TRY(ctx, hipMemcpy(d1, bm.data(), bm.size() * sizeof(double), hipMemcpyHostToDevice));
  if (!one) {
    to_branch_major(v2, n2, dim, &bm);
    HIP_TRY(ctx, tmp.alloc((void**)&d2, bm.size() * sizeof(double)));
    HIP_TRY(ctx, hipMemcpy(d2, bm.data(), bm.size() * sizeof(double), hipMemcpyHostToDevice));
  } else d2 = d1;
  if (independent) {   // AnalysisTools.cpp:150-157: j runs over i alone
    HIP_TRY(ctx, tmp.alloc((void**)&d_out, n1 * sizeof(double)));
    HIP_TRY(ctx, launch_pair_diag(kind, 0.0, B, 1, d1, n1, d2, n2, n1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d_out, nullptr,
                                  nullptr, nullptr, nullptr, nullptr));
    std::vector<double> dg(n1);
    HIP_TRY(ctx, hipMemcpy(dg.data(), d_out, n1 * sizeof(double), hipMemcpyDeviceToHost));
    std::fill(out, out + n1 * n2, 0.0);
    for (size_t i = 0; i < n1; ++i) out[i * n2 + i] = dg[i];
    return CMX_OK;
  }
  const size_t ldx1 = (n1 + 15) / 16 * 16, ldx2 = (n2 + 15) / 16 * 16;
  double *X1, *s1, *r1, *X2, *s2, *r2;
  cmx_status s;
  if ((s = scratch(ctx, "pair_X1", sizeof(double) * Bp * ldx1, (void**)&X1)) != CMX_OK) return s;
  if ((s = scratch(ctx, "pair_s1", sizeof(double) * n1, (void**)&s1)) != CMX_OK) return s;
  if ((s = scratch(ctx, "pair_r1", sizeof(double) * n1, (void**)&r1)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_pair_prep(kind, 0.0, d1, n1, n1, B, 1, X1, ldx1, Bp, s1, r1, nullptr, nullptr));
  if (one) { X2 = X1; s2 = s1; r2 = r1; }
  else {
    if ((s = scratch(ctx, "pair_X2", sizeof(double) * Bp * ldx2, (void**)&X2)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_s2", sizeof(double) * n2, (void**)&s2)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_r2", sizeof(double) * n2, (void**)&r2)) != CMX_OK) return s;
    HIP_TRY(ctx, launch_pair_prep(kind, 0.0, d2, n2, n2, B, 1, X2, ldx2, Bp, s2, r2, nullptr, nullptr));
  }
  HIP_TRY(ctx, tmp.alloc((void**)&d_out, n1 * n2 * sizeof(double)));
  // the full rectangle (one-set form too: the reference fills both triangles and the diagonal)
  HIP_TRY(ctx, launch_pair_gram(kind, B, Bp, X1, s1, r1, n1, ldx1, X2, s2, r2, n2, one ? ldx1 : ldx2, 0, d_out, n2, nullptr));
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, d_out, n1 * n2 * sizeof(double), hipMemcpyDeviceToHost));
  if (one) {
    // matrix[i][i] = 1 for cosinus / correlation (AnalysisTools.cpp:178, 236); the lower triangle mirrors the upper one
    // (matrix[i][j] = matrix[j][i] = f(v_i, v_j), j > i)
    for (size_t i = 0; i < n1; ++i) {
      if (kind == CMX_STAT_COSINUS || kind == CMX_STAT_CORRELATION) out[i * n1 + i] = 1.0;
      for (size_t j = 0; j < i; ++j) out[i * n1 + j] = out[j * n1 + i];
    }
  }
  return CMX_OK;
}

cmx_status cmx_pair_stats(cmx_ctx* ctx, int kind, const double* params, const double* counts1, size_t n1,
                          const double* counts2, size_t n2, double* out) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (!counts1 || !out || n1 == 0 || (counts2 && n2 == 0)) return fail(ctx, CMX_ERR_INVALID, "cmx_pair_stats: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t BK = (size_t)ctx->hm.B * ctx->hm.K;
  if (!counts2) n2 = n1;
  TmpDev tmp;
  std::vector<double> bm;
  double *d1 = nullptr, *d2 = nullptr, *d_out = nullptr;
  to_branch_major(counts1, n1, BK, &bm);
  HIP_TRY(ctx, tmp.alloc((void**)&d1, bm.size() * sizeof(double)));
  HIP_TRY(ctx, hipMemcpy(d1, bm.data(), bm.size() * sizeof(double), hipMemcpyHostToDevice));
  if (counts2) {
    to_branch_major(counts2, n2, BK, &bm);
    HIP_TRY(ctx, tmp.alloc((void**)&d2, bm.size() * sizeof(double)));
    HIP_TRY(ctx, hipMemcpy(d2, bm.data(), bm.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  HIP_TRY(ctx, tmp.alloc((void**)&d_out, n1 * n2 * sizeof(double)));
  s = cmx_pair_stats_dev(ctx, kind, params, d1, n1, n1, d2, n2, n2, d_out, n2, nullptr);
  if (s != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, d_out, n1 * n2 * sizeof(double), hipMemcpyDeviceToHost));
  return CMX_OK;
}

// ------------------------------------------------------------------------------------------------ null distribution
// The null's alignments, [replicate][batch][taxon][rep_ram] bytes (what cmx_null_intra_dev takes as `supplied`):
// NonHomogeneousSequenceSimulator::simulate(repRAM) twice per replicate (AnalysisTools.cpp:591, 612), simulated-site index
// g = ((rep * 2 + batch) * rep_ram + j) as everywhere.
cmx_status cmx_null_simulate_dev(cmx_ctx* ctx, uint64_t seed, size_t rep_begin, size_t rep_end, size_t rep_ram, uint8_t* d_aln,
                                 void* stream) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (rep_end <= rep_begin || rep_ram == 0 || !d_aln) return fail(ctx, CMX_ERR_INVALID, "cmx_null_simulate: bad arguments");
  if ((s = rng_range(ctx, (uint64_t)rep_end * 2 * rep_ram, "cmx_null_simulate")) != CMX_OK) return s;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t nsites = (rep_end - rep_begin) * 2 * rep_ram;
  // node states of a pass: nn bytes per site, up to 4 GiB -- one pass for the 2 * 10^7 sites of the target's null (nine
  // passes of 2^21 sites left nine tails of half-empty CUs)
  const size_t chunk = std::min<size_t>(nsites, std::max<size_t>((size_t)1 << 21, (((size_t)4 << 30) / (size_t)ctx->hm.nn) & ~(size_t)1023));
  uint8_t* d_states;
  if ((s = scratch(ctx, "null_states", (size_t)ctx->hm.nn * chunk, (void**)&d_states)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_simulate_blocked(ctx->dm, seed, (uint64_t)rep_begin * 2 * rep_ram, nsites, rep_ram, d_aln, d_states, chunk,
                                       (hipStream_t)stream));
  return CMX_OK;
}

static cmx_status null_unfused_dev(cmx_ctx* ctx1, cmx_ctx* ctx2, int kind, const double* params, uint64_t seed,
                                   size_t rep_begin, size_t rep_end, size_t rep_ram, const uint8_t* d_supplied, double* d_stat,
                                   int32_t* d_rcmin, double* d_prmin, double* d_nmin, void* stream);

cmx_status cmx_null_intra_dev(cmx_ctx* ctx, int kind, const double* params, uint64_t seed, size_t rep_begin,
                              size_t rep_end, size_t rep_ram, const uint8_t* d_supplied, double* d_stat,
                              int32_t* d_rcmin, double* d_prmin, double* d_nmin, void* stream) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if ((s = check_kind(ctx, kind)) != CMX_OK) return s;
  if (rep_end <= rep_begin || rep_ram == 0 || !d_stat) return fail(ctx, CMX_ERR_INVALID, "cmx_null_intra: bad arguments");
  if ((s = rng_range(ctx, (uint64_t)rep_end * 2 * rep_ram, "cmx_null_intra")) != CMX_OK) return s;
  if (!ctx->map_average || !ctx->map_joint || kind == CMX_STAT_DISCRETE_MI_BOUNDS || ctx->hm.plain) {
    // nijt.average = no (AnalysisTools.cpp:598-610): the fused kernel only knows the averaged mapping; and a statistic that
    // needs a joint table per pair cannot be evaluated per lane inside the mapping wave.  The same simulate -> map ->
    // score sequence then runs unfused, which is what the two-data-set null does with both sides equal.
    return null_unfused_dev(ctx, ctx, kind, params, seed, rep_begin, rep_end, rep_ram, d_supplied, d_stat, d_rcmin, d_prmin, d_nmin, stream);
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!d_supplied) {
    // simulate first, at full occupancy, then map the alignments as "supplied" ones: the same draws, the same results
    // as a simulator inside the mapping waves (round 1; 7.8 % of the launch there, latency nobody could hide).  The
    // alignments of a pass stay under 4 GiB: a larger null runs as several passes over replicate ranges.
    const size_t per_rep = 2 * (size_t)ctx->hm.T * rep_ram;
    static const size_t pass_bytes = [] {   // CMX_NULL_PASS_BYTES: tests exercise the multi-pass path with small nulls
      const char* e = getenv("CMX_NULL_PASS_BYTES");
      return e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)4 << 30);
    }();
    const size_t reps_per_pass = std::max<size_t>(1, pass_bytes / per_rep);
    if (rep_end - rep_begin > reps_per_pass) {
      for (size_t r0 = rep_begin; r0 < rep_end; r0 += reps_per_pass) {
        const size_t r1 = std::min(rep_end, r0 + reps_per_pass), o = (r0 - rep_begin) * rep_ram;
        if ((s = cmx_null_intra_dev(ctx, kind, params, seed, r0, r1, rep_ram, nullptr, d_stat + o, d_rcmin ? d_rcmin + o : nullptr,
                                    d_prmin ? d_prmin + o : nullptr, d_nmin ? d_nmin + o : nullptr, stream)) != CMX_OK)
          return s;
      }
      return CMX_OK;
    }
    uint8_t* d_aln;
    if ((s = scratch(ctx, "null_aln", (rep_end - rep_begin) * per_rep, (void**)&d_aln)) != CMX_OK) return s;
    if ((s = cmx_null_simulate_dev(ctx, seed, rep_begin, rep_end, rep_ram, d_aln, stream)) != CMX_OK) return s;
    d_supplied = d_aln;
  }
  MapArgs a{};
  a.m = ctx->dm; a.ws = ctx->ws;
  a.nsites = (rep_end - rep_begin) * rep_ram;
  a.stat_kind = kind;
  a.stat_param = (kind == CMX_STAT_DISCRETE_MI) ? (params ? params[0] : 0.99) : 0.0;
  if ((s = stat_mean_vectors(ctx, kind, params, &a.stat_mean, stream)) != CMX_OK) return s;
  a.seed = seed; a.rep_begin = rep_begin; a.rep_ram = rep_ram; a.supplied = d_supplied;
  a.null_stat = d_stat; a.null_rcmin = d_rcmin; a.null_prmin = d_prmin; a.null_nmin = d_nmin;
  const size_t ks = (size_t)map_sites_per_wave(ctx->hm.dS);
  const size_t blocks_needed = ((a.nsites + ks - 1) / ks + kWavesPerBlock - 1) / kWavesPerBlock;
  const int grid = (int)std::min<size_t>(blocks_needed, (size_t)ctx->grid_blocks);
  HIP_TRY(ctx, launch_map(a, kModeNull, grid, (hipStream_t)stream));
  return CMX_OK;
}

// simulations.continuous = yes (CoMap.cpp:146, 213): the replicates' alignments come from the continuous-rate simulator,
// straight into the device buffer cmx_null_intra_dev maps as "supplied" alignments -- nothing crosses PCIe
cmx_status cmx_null_intra_continuous_dev(cmx_ctx* ctx, int kind, const double* params, uint64_t seed, size_t rep_begin, size_t rep_end,
                                         size_t rep_ram, double gamma_alpha, double p_invariant, double* d_stat, int32_t* d_rcmin,
                                         double* d_prmin, double* d_nmin, void* stream) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (rep_end <= rep_begin || rep_ram == 0 || !d_stat) return fail(ctx, CMX_ERR_INVALID, "cmx_null_intra_continuous: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t T = (size_t)ctx->hm.T, per_rep = 2 * T * rep_ram;
  const size_t reps_per_pass = std::max<size_t>(1, ((size_t)4 << 30) / per_rep);   // alignments of a pass stay under 4 GiB
  uint8_t* d_aln;
  if ((s = scratch(ctx, "null_aln", std::min(reps_per_pass, rep_end - rep_begin) * per_rep, (void**)&d_aln)) != CMX_OK) return s;
  for (size_t r0 = rep_begin; r0 < rep_end; r0 += reps_per_pass) {
    const size_t r1 = std::min(rep_end, r0 + reps_per_pass), o = (r0 - rep_begin) * rep_ram;
    for (size_t r = r0; r < r1; ++r)
      for (int h = 0; h < 2; ++h) {   // [replicate][batch][taxon][rep_ram]; simulated-site index g = (rep * 2 + batch) * rep_ram + j
        if ((s = cmx_simulate_continuous_dev(ctx, seed, ((uint64_t)r * 2 + h) * rep_ram, rep_ram, gamma_alpha, p_invariant,
                                             d_aln + ((r - r0) * 2 + h) * T * rep_ram, rep_ram, nullptr, stream)) != CMX_OK)
          return s;
      }
    if ((s = cmx_null_intra_dev(ctx, kind, params, seed, r0, r1, rep_ram, d_aln, d_stat + o, d_rcmin ? d_rcmin + o : nullptr,
                                d_prmin ? d_prmin + o : nullptr, d_nmin ? d_nmin + o : nullptr, stream)) != CMX_OK)
      return s;
  }
  return CMX_OK;
}

cmx_status cmx_null_intra_continuous(cmx_ctx* ctx, int kind, const double* params, uint64_t seed, size_t rep_begin, size_t rep_end,
                                     size_t rep_ram, double gamma_alpha, double p_invariant, double* stat, int32_t* rcmin,
                                     double* prmin, double* nmin) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (rep_end <= rep_begin || rep_ram == 0 || !stat) return fail(ctx, CMX_ERR_INVALID, "cmx_null_intra_continuous: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n = (rep_end - rep_begin) * rep_ram;
  TmpDev tmp;
  double *d_stat, *d_pr, *d_nm;
  int32_t* d_rc;
  HIP_TRY(ctx, tmp.alloc((void**)&d_stat, n * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_pr, n * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_nm, n * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_rc, n * sizeof(int32_t)));
  s = cmx_null_intra_continuous_dev(ctx, kind, params, seed, rep_begin, rep_end, rep_ram, gamma_alpha, p_invariant, d_stat, d_rc, d_pr,
                                    d_nm, nullptr);
  if (s != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(stat, d_stat, n * sizeof(double), hipMemcpyDeviceToHost));
  if (rcmin) HIP_TRY(ctx, hipMemcpy(rcmin, d_rc, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (prmin) HIP_TRY(ctx, hipMemcpy(prmin, d_pr, n * sizeof(double), hipMemcpyDeviceToHost));
  if (nmin) HIP_TRY(ctx, hipMemcpy(nmin, d_nm, n * sizeof(double), hipMemcpyDeviceToHost));
  return CMX_OK;
}

cmx_status cmx_null_intra(cmx_ctx* ctx, int kind, const double* params, uint64_t seed, size_t rep_begin, size_t rep_end,
                          size_t rep_ram, const uint8_t* supplied, double* stat, int32_t* rcmin, double* prmin,
                          double* nmin) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (rep_end <= rep_begin || rep_ram == 0 || !stat) return fail(ctx, CMX_ERR_INVALID, "cmx_null_intra: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n = (rep_end - rep_begin) * rep_ram;
  TmpDev tmp;
  uint8_t* d_sup = nullptr;
  double *d_stat, *d_pr, *d_nm;
  int32_t* d_rc;
  if (supplied) {
    const size_t bytes = (rep_end - rep_begin) * 2 * (size_t)ctx->hm.T * rep_ram;
    for (size_t i = 0; i < bytes; ++i)
      if (supplied[i] >= (unsigned)ctx->hm.S) return fail(ctx, CMX_ERR_INVALID, "cmx_null_intra: supplied alignments must be fully resolved");
    HIP_TRY(ctx, tmp.alloc((void**)&d_sup, bytes));
    HIP_TRY(ctx, hipMemcpy(d_sup, supplied, bytes, hipMemcpyHostToDevice));
  }
  HIP_TRY(ctx, tmp.alloc((void**)&d_stat, n * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_pr, n * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_nm, n * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_rc, n * sizeof(int32_t)));
  s = cmx_null_intra_dev(ctx, kind, params, seed, rep_begin, rep_end, rep_ram, d_sup, d_stat, d_rc, d_pr, d_nm, nullptr);
  if (s != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(stat, d_stat, n * sizeof(double), hipMemcpyDeviceToHost));
  if (rcmin) HIP_TRY(ctx, hipMemcpy(rcmin, d_rc, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (prmin) HIP_TRY(ctx, hipMemcpy(prmin, d_pr, n * sizeof(double), hipMemcpyDeviceToHost));
  if (nmin) HIP_TRY(ctx, hipMemcpy(nmin, d_nm, n * sizeof(double), hipMemcpyDeviceToHost));
  return CMX_OK;
}

// ------------------------------------------------------------------------------------------------ inter-gene null
// AnalysisTools::getNullDistributionInterDR (AnalysisTools.cpp:662-735): per replicate simulate + map rep_ram sites
// under data set 1 and rep_ram sites under data set 2, then score site j of the one against site j of the other.
// Not fused (a "next" row of SURVEY 8f): simulate -> map -> diagonal-pair kernel, everything resident in HBM.
// Simulated-site indices follow the intra scheme: g = ((rep*2 + h)*rep_ram + j), h = 0 for ctx1 and 1 for ctx2.
// d_supplied (intra use only: both contexts the same): [nrep][2][T][rep_ram] alignments to map instead of simulating
static cmx_status null_unfused_dev(cmx_ctx* ctx1, cmx_ctx* ctx2, int kind, const double* params, uint64_t seed,
                                   size_t rep_begin, size_t rep_end, size_t rep_ram, const uint8_t* d_supplied, double* d_stat,
                                   int32_t* d_rcmin, double* d_prmin, double* d_nmin, void* stream) {
  cmx_status s = need_model(ctx1);
  if (s != CMX_OK) return s;
  if (!ctx2 || !ctx2->has_model) return fail(ctx1, CMX_ERR_INVALID, "cmx_null_inter: second context has no model");
  if ((s = check_kind(ctx1, kind)) != CMX_OK) return s;
  if (rep_end <= rep_begin || rep_ram == 0 || !d_stat) return fail(ctx1, CMX_ERR_INVALID, "cmx_null_inter: bad arguments");
  if ((s = rng_range(ctx1, (uint64_t)rep_end * 2 * rep_ram, "cmx_null_inter")) != CMX_OK) return s;
  if ((s = rng_range(ctx2, (uint64_t)rep_end * 2 * rep_ram, "cmx_null_inter")) != CMX_OK) return s;
  if (ctx1->device != ctx2->device) return fail(ctx1, CMX_ERR_INVALID, "cmx_null_inter: contexts live on different devices");
  if (ctx1->hm.B != ctx2->hm.B || ctx1->hm.K != ctx2->hm.K)
    return fail(ctx1, CMX_ERR_INVALID, "cmx_null_inter: the two data sets must have the same branches and substitution types "
                                       "(Statistic::getValueForPair throws DimensionException otherwise)");
  HIP_TRY(ctx1, hipSetDevice(ctx1->device));
  hipStream_t st = (hipStream_t)stream;
  const size_t nrep = rep_end - rep_begin, n = nrep * rep_ram;
  const size_t BK = (size_t)ctx1->hm.B * ctx1->hm.K;
  double *cnt[2], *pr[2], *nm[2];
  int32_t* rc[2];
  cmx_ctx* cx[2] = {ctx1, ctx2};
  for (int h = 0; h < 2; ++h) {
    cmx_ctx* c = cx[h];
    const std::string tag = std::string("inter") + char('0' + h);
    uint8_t *d_aln, *d_st;
    if ((s = scratch(ctx1, (tag + "_aln").c_str(), (size_t)c->hm.T * n, (void**)&d_aln)) != CMX_OK) return s;
    // (node states share the alignment's row stride n in simulate_kernel: [nn][n], not [nn][rep_ram] -- sized for rep_ram
    // this overflowed as soon as a call held more than one replicate)
    if ((s = scratch(ctx1, (tag + "_st").c_str(), (size_t)c->hm.nn * n, (void**)&d_st)) != CMX_OK) return s;
    if ((s = scratch(ctx1, (tag + "_cnt").c_str(), sizeof(double) * BK * n, (void**)&cnt[h])) != CMX_OK) return s;
    if ((s = scratch(ctx1, (tag + "_pr").c_str(), sizeof(double) * n, (void**)&pr[h])) != CMX_OK) return s;
    if ((s = scratch(ctx1, (tag + "_nm").c_str(), sizeof(double) * n, (void**)&nm[h])) != CMX_OK) return s;
    if ((s = scratch(ctx1, (tag + "_rc").c_str(), sizeof(int32_t) * n, (void**)&rc[h])) != CMX_OK) return s;
    if (d_supplied) {
      for (size_t r = 0; r < nrep; ++r)   // batch h of replicate r: [T][rep_ram] -> columns r * rep_ram .. of the [T][n] alignment
        HIP_TRY(ctx1, hipMemcpy2DAsync(d_aln + r * rep_ram, n, d_supplied + ((r * 2 + h) * (size_t)c->hm.T) * rep_ram, rep_ram, rep_ram,
                                       (size_t)c->hm.T, hipMemcpyDeviceToDevice, st));
    } else {
      // ONE launch for all replicates of this side (round 3: one per replicate and side -- 2 000 tiny launches for
      // nb_rep_CPU = 1000): block r of rep_ram columns holds the global sites ((rep_begin + r) * 2 + h) * rep_ram ..
      const uint64_t g0 = ((uint64_t)rep_begin * 2 + h) * (uint64_t)rep_ram;
      HIP_TRY(ctx1, launch_simulate(c->dm, seed, g0, n, d_aln, n, nullptr, d_st, st, rep_ram, 2 * (uint64_t)rep_ram));
    }
    s = map_sites_impl(c, d_aln, n, n, nullptr, cnt[h], n, nullptr, pr[h], rc[h], nm[h], stream, true);
    if (s != CMX_OK) { if (c != ctx1) ctx1->err = c->err; return s; }
  }
  if (kind == CMX_STAT_DISCRETE_MI_BOUNDS) {
    MiBounds mb;
    uint32_t *c1, *c2;
    uint8_t *b1, *b2;
    size_t lx1, lx2;
    if ((s = mi_bounds(ctx1, params, &mb, stream)) != CMX_OK) return s;
    if ((s = mi_classify(ctx1, mb, cnt[0], n, n, "n1", &c1, &b1, &lx1, stream)) != CMX_OK) return s;
    if ((s = mi_classify(ctx1, mb, cnt[1], n, n, "n2", &c2, &b2, &lx2, stream)) != CMX_OK) return s;
    HIP_TRY(ctx1, launch_mi_pairs_diag(ctx1->hm.B, c1, b1, lx1, c2, b2, lx2, n, d_stat, st));
    HIP_TRY(ctx1, launch_pair_diag(kind, 0.0, ctx1->hm.B, ctx1->hm.K, cnt[0], n, cnt[1], n, n, rc[0], rc[1], pr[0], pr[1], nm[0], nm[1],
                                   nullptr, d_rcmin, d_prmin, d_nmin, nullptr, st));   // the minima only
    return CMX_OK;
  }
  const double param = (kind == CMX_STAT_DISCRETE_MI) ? (params ? params[0] : 0.99) : 0.0;
  const double* d_mean = nullptr;
  if ((s = stat_mean_vectors(ctx1, kind, params, &d_mean, stream)) != CMX_OK) return s;
  HIP_TRY(ctx1, launch_pair_diag(kind, param, ctx1->hm.B, ctx1->hm.K, cnt[0], n, cnt[1], n, n, rc[0], rc[1], pr[0], pr[1],
                                 nm[0], nm[1], d_stat, d_rcmin, d_prmin, d_nmin, d_mean, st));
  return CMX_OK;
}

cmx_status cmx_null_inter_dev(cmx_ctx* ctx1, cmx_ctx* ctx2, int kind, const double* params, uint64_t seed,
                              size_t rep_begin, size_t rep_end, size_t rep_ram, double* d_stat, int32_t* d_rcmin,
                              double* d_prmin, double* d_nmin, void* stream) {
  return null_unfused_dev(ctx1, ctx2, kind, params, seed, rep_begin, rep_end, rep_ram, nullptr, d_stat, d_rcmin, d_prmin, d_nmin, stream);
}

cmx_status cmx_null_inter(cmx_ctx* ctx1, cmx_ctx* ctx2, int kind, const double* params, uint64_t seed, size_t rep_begin,
                          size_t rep_end, size_t rep_ram, double* stat, int32_t* rcmin, double* prmin, double* nmin) {
  cmx_status s = need_model(ctx1);
  if (s != CMX_OK) return s;
  if (rep_end <= rep_begin || rep_ram == 0 || !stat) return fail(ctx1, CMX_ERR_INVALID, "cmx_null_inter: bad arguments");
  HIP_TRY(ctx1, hipSetDevice(ctx1->device));
  const size_t n = (rep_end - rep_begin) * rep_ram;
  TmpDev tmp;
  double *d_stat, *d_pr, *d_nm;
  int32_t* d_rc;
  HIP_TRY(ctx1, tmp.alloc((void**)&d_stat, n * sizeof(double)));
  HIP_TRY(ctx1, tmp.alloc((void**)&d_pr, n * sizeof(double)));
  HIP_TRY(ctx1, tmp.alloc((void**)&d_nm, n * sizeof(double)));
  HIP_TRY(ctx1, tmp.alloc((void**)&d_rc, n * sizeof(int32_t)));
  s = cmx_null_inter_dev(ctx1, ctx2, kind, params, seed, rep_begin, rep_end, rep_ram, d_stat, d_rc, d_pr, d_nm, nullptr);
  if (s != CMX_OK) return s;
  HIP_TRY(ctx1, hipDeviceSynchronize());
  HIP_TRY(ctx1, hipMemcpy(stat, d_stat, n * sizeof(double), hipMemcpyDeviceToHost));
  if (rcmin) HIP_TRY(ctx1, hipMemcpy(rcmin, d_rc, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (prmin) HIP_TRY(ctx1, hipMemcpy(prmin, d_pr, n * sizeof(double), hipMemcpyDeviceToHost));
  if (nmin) HIP_TRY(ctx1, hipMemcpy(nmin, d_nm, n * sizeof(double), hipMemcpyDeviceToHost));
  return CMX_OK;
}

// ------------------------------------------------------------------------------------------------ p-values
// The null distribution as the p-value kernel wants it (CoETools.cpp:636-652): Domain(0, max norm, nclasses) classes of
// the null pairs' min norms, every class sorted ascending, classes laid out one after the other; hist = class sizes.
static cmx_status prepare_null(cmx_ctx* ctx, const double* d_norms, size_t n, int nclasses, const double* d_null_stat,
                               const double* d_null_nmin, size_t nnull, hipStream_t st, NullTable* out) {
  cmx_status s;
  double *maxnorm, *sa, *sb;
  uint32_t *ca, *cb, *hist, *bins;
  NullClass* cls;
  const size_t nn = nnull ? nnull : 1;
  if ((s = scratch(ctx, "pv_max", sizeof(double), (void**)&maxnorm)) != CMX_OK) return s;
  if ((s = scratch(ctx, "pv_sa", sizeof(double) * nn, (void**)&sa)) != CMX_OK) return s;
  if ((s = scratch(ctx, "pv_sb", sizeof(double) * nn, (void**)&sb)) != CMX_OK) return s;
  if ((s = scratch(ctx, "pv_ca", sizeof(uint32_t) * nn, (void**)&ca)) != CMX_OK) return s;
  if ((s = scratch(ctx, "pv_cb", sizeof(uint32_t) * nn, (void**)&cb)) != CMX_OK) return s;
  if ((s = scratch(ctx, "pv_hist", sizeof(uint32_t) * 66, (void**)&hist)) != CMX_OK) return s;
  if ((s = scratch(ctx, "pv_cls", sizeof(NullClass) * 66, (void**)&cls)) != CMX_OK) return s;
  if ((s = scratch(ctx, "pv_bins", sizeof(uint32_t) * ((nn >> kNullBinShift) + 2 * 66), (void**)&bins)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_max_reduce(d_norms, n, maxnorm, st));
  HIP_TRY(ctx, launch_null_classify(d_null_stat, d_null_nmin, nnull, maxnorm, nclasses, ca, hist, st));
  if (nnull > 0) {
    HIP_TRY(ctx, hipMemcpyAsync(sa, d_null_stat, sizeof(double) * nnull, hipMemcpyDeviceToDevice, st));
    size_t tmp_bytes = 0;
    HIP_TRY(ctx, sort_null_by_class(nullptr, tmp_bytes, sa, sb, ca, cb, nnull, st));
    void* tmp;
    if ((s = scratch(ctx, "pv_sorttmp", tmp_bytes, &tmp)) != CMX_OK) return s;
    HIP_TRY(ctx, sort_null_by_class(tmp, tmp_bytes, sa, sb, ca, cb, nnull, st));
  }
  HIP_TRY(ctx, launch_null_index(sa, hist, nclasses, nnull, cls, bins, st));
  *out = NullTable{sa, cls, bins, maxnorm, nclasses};
  return CMX_OK;
}

cmx_status cmx_intra_pvalues_dev(cmx_ctx* ctx, const double* d_stat, size_t ldo, const double* d_norms, size_t n,
                                 int nclasses, const double* d_null_stat, const double* d_null_nmin, size_t nnull,
                                 double* d_pvalue, int32_t* d_nsim, void* stream) {
  if (!ctx) return CMX_ERR_INVALID;
  if (!d_stat || !d_norms || !d_pvalue || !d_nsim || n == 0 || ldo < n || nclasses < 1 || nclasses > 64 ||
      (nnull > 0 && (!d_null_stat || !d_null_nmin)) || nnull > 0xfffffff0ull)
    return fail(ctx, CMX_ERR_INVALID, "cmx_intra_pvalues: bad arguments (nclasses must be in 1..64)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  cmx_status s;
  NullTable nt;
  if ((s = prepare_null(ctx, d_norms, n, nclasses, d_null_stat, d_null_nmin, nnull, st, &nt)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_pvalues(d_stat, ldo, d_norms, n, nt, d_pvalue, d_nsim, st));
  return CMX_OK;
}

cmx_status cmx_intra_pvalues(cmx_ctx* ctx, const double* stat, const double* norms, size_t n, int nclasses,
                             const double* null_stat, const double* null_nmin, size_t nnull, double* pvalue,
                             int32_t* nsim) {
  if (!ctx) return CMX_ERR_INVALID;
  if (!stat || !norms || !pvalue || !nsim || n == 0) return fail(ctx, CMX_ERR_INVALID, "cmx_intra_pvalues: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  TmpDev tmp;
  double *d_stat, *d_norms, *d_ns = nullptr, *d_nm = nullptr, *d_pv;
  int32_t* d_nsim;
  HIP_TRY(ctx, tmp.alloc((void**)&d_stat, n * n * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_norms, n * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_pv, n * n * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_nsim, n * n * sizeof(int32_t)));
  HIP_TRY(ctx, hipMemcpy(d_stat, stat, n * n * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_norms, norms, n * sizeof(double), hipMemcpyHostToDevice));
  if (nnull) {
    HIP_TRY(ctx, tmp.alloc((void**)&d_ns, nnull * sizeof(double)));
    HIP_TRY(ctx, tmp.alloc((void**)&d_nm, nnull * sizeof(double)));
    HIP_TRY(ctx, hipMemcpy(d_ns, null_stat, nnull * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(d_nm, null_nmin, nnull * sizeof(double), hipMemcpyHostToDevice));
  }
  cmx_status s = cmx_intra_pvalues_dev(ctx, d_stat, n, d_norms, n, nclasses, d_ns, d_nm, nnull, d_pv, d_nsim, nullptr);
  if (s != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(pvalue, d_pv, n * n * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(nsim, d_nsim, n * n * sizeof(int32_t), hipMemcpyDeviceToHost));
  return CMX_OK;
}

// ------------------------------------------------------------------------------------------------ compacted rows
cmx_status cmx_intra_rows_dev(cmx_ctx* ctx, const double* d_stat, size_t ldo, const double* d_pvalue, const int32_t* d_nsim,
                              size_t n, const int32_t* d_rate_class, const double* d_post_rate, const double* d_norm,
                              const cmx_pair_filters* filters, cmx_pair_row* d_rows, size_t capacity, uint64_t* d_count,
                              void* stream) {
  if (!ctx) return CMX_ERR_INVALID;
  if (!d_stat || n == 0 || ldo < n || !d_rate_class || !d_post_rate || !d_norm || !d_count || (capacity && !d_rows) ||
      n > 0x7fffffffull)
    return fail(ctx, CMX_ERR_INVALID, "cmx_intra_rows: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  cmx_pair_filters f{0, -1, 0.0, -1.0, 0.0};
  if (filters) f = *filters;
  unsigned long long* rowcount;
  cmx_status s;
  if ((s = scratch(ctx, "rows_count", sizeof(unsigned long long) * (n * kPairRowSegs + 1), (void**)&rowcount)) != CMX_OK) return s;
  size_t tmp_bytes = 0;
  HIP_TRY(ctx, launch_pair_rows(d_stat, ldo, d_pvalue, d_nsim, n, d_rate_class, d_post_rate, d_norm, f, rowcount, nullptr,
                                tmp_bytes, d_rows, capacity, reinterpret_cast<unsigned long long*>(d_count), (hipStream_t)stream));
  void* tmp = nullptr;
  if ((s = scratch(ctx, "rows_scan", tmp_bytes ? tmp_bytes : 16, &tmp)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_pair_rows(d_stat, ldo, d_pvalue, d_nsim, n, d_rate_class, d_post_rate, d_norm, f, rowcount, tmp,
                                tmp_bytes, d_rows, capacity, reinterpret_cast<unsigned long long*>(d_count), (hipStream_t)stream));
  return CMX_OK;
}

// CoETools::computeIntraStats' pair loop (CoETools.cpp:672-724) for the rows [row_begin, row_end) of the upper triangle,
// a block of rows at a time: Gram block on the matrix cores -> p-values -> filters -> compaction.  No N x N matrix exists;
// the dense scratch is one row block (<= 256 MiB).  Ranks of a multi-GPU job call it with disjoint row ranges: rows come
// out in the reference's (i, j) order, so the ranks' outputs concatenate to the single-GPU output.
cmx_status cmx_intra_rows_range_dev(cmx_ctx* ctx, int kind, const double* params, const double* d_counts, size_t n, size_t ldc,
                                    const int32_t* d_rate_class, const double* d_post_rate, const double* d_norm,
                                    const double* d_null_stat, const double* d_null_nmin, size_t nnull, int nclasses,
                                    const cmx_pair_filters* filters, size_t row_begin, size_t row_end, cmx_pair_row* d_rows,
                                    size_t capacity, uint64_t* d_count, void* stream) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if ((s = check_kind(ctx, kind)) != CMX_OK) return s;
  const bool with_null = d_null_stat != nullptr;
  if (!d_counts || n == 0 || ldc < n || !d_rate_class || !d_post_rate || !d_norm || !d_count || (capacity && !d_rows) ||
      n > 0x7fffffffull || row_begin > row_end || row_end > n || (with_null && (!d_null_nmin || nclasses < 1 || nclasses > 64)) ||
      nnull > 0xfffffff0ull)
    return fail(ctx, CMX_ERR_INVALID, "cmx_intra_rows_range: bad arguments");
  const HostModel& h = ctx->hm;
  if (h.B < 2) return fail(ctx, CMX_ERR_INVALID, "cmx_intra_rows_range: need at least two branches");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  cmx_pair_filters f{0, -1, 0.0, -1.0, 0.0};
  if (filters) f = *filters;
  HIP_TRY(ctx, hipMemsetAsync(d_count, 0, sizeof(uint64_t), st));
  if (row_begin == row_end) return CMX_OK;
  // operand of the Gram kernel for all n sites (both sides of every pair)
  const double param = (kind == CMX_STAT_DISCRETE_MI) ? (params ? params[0] : 0.99) : 0.0;
  const double* d_mean = nullptr;
  if ((s = stat_mean_vectors(ctx, kind, params, &d_mean, stream)) != CMX_OK) return s;
  const int gk = kind == CMX_STAT_CORRECTED_CORRELATION ? CMX_STAT_CORRELATION : kind;
  const int Bp = (h.B + 3) / 4 * 4;
  const size_t ldx = (n + 15) / 16 * 16;
  double *X = nullptr, *sv = nullptr, *rv = nullptr;
  uint32_t* mcls = nullptr;
  uint8_t* mbad = nullptr;
  size_t mldx = 0;
  if (kind == CMX_STAT_DISCRETE_MI_BOUNDS) {
    MiBounds mb;
    if ((s = mi_bounds(ctx, params, &mb, stream)) != CMX_OK) return s;
    if ((s = mi_classify(ctx, mb, d_counts, n, ldc, "1", &mcls, &mbad, &mldx, stream)) != CMX_OK) return s;
  } else {
    if ((s = scratch(ctx, "pair_X1", sizeof(double) * Bp * ldx, (void**)&X)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_s1", sizeof(double) * n, (void**)&sv)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_r1", sizeof(double) * n, (void**)&rv)) != CMX_OK) return s;
    HIP_TRY(ctx, launch_pair_prep(gk, param, d_counts, n, ldc, h.B, h.K, X, ldx, Bp, sv, rv, d_mean, st));
  }
  NullTable nt{};
  if (with_null && (s = prepare_null(ctx, d_norm, n, nclasses, d_null_stat, d_null_nmin, nnull, st, &nt)) != CMX_OK) return s;
  // row blocks: a multiple of 64 rows (Gram tiles), dense scratch (the f64 statistic; the p-values are looked up by the
  // pass that writes the rows, for the pairs it writes) <= 256 MiB
  size_t RB = ((size_t)256 << 20) / (8 * n) / 64 * 64;
  RB = std::max<size_t>(64, std::min<size_t>(RB, (row_end - row_begin + 63) / 64 * 64));
  double* blk_stat;
  unsigned long long* rowcount;
  if ((s = scratch(ctx, "blk_stat", sizeof(double) * RB * n, (void**)&blk_stat)) != CMX_OK) return s;
  if ((s = scratch(ctx, "rows_count", sizeof(unsigned long long) * (RB * kPairRowSegs + 1), (void**)&rowcount)) != CMX_OK) return s;
  size_t tmp_bytes = 0;
  HIP_TRY(ctx, launch_pair_rows(blk_stat, n, nullptr, nullptr, n, d_rate_class, d_post_rate, d_norm, f, rowcount, nullptr, tmp_bytes,
                                d_rows, capacity, reinterpret_cast<unsigned long long*>(d_count), st, 0, RB));
  void* tmp = nullptr;
  if ((s = scratch(ctx, "rows_scan", tmp_bytes ? tmp_bytes : 16, &tmp)) != CMX_OK) return s;
  for (size_t i0 = row_begin; i0 < row_end; i0 += RB) {
    const size_t rb = std::min(RB, row_end - i0);
    if (gk == CMX_STAT_EUCLIDIAN_DISTANCE) return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_intra_rows_range: EuclidianDistance is a distance, not a statistic");
    if (mcls) HIP_TRY(ctx, launch_mi_pairs_block(h.B, mcls + i0, mbad + i0, rb, mldx, mcls, mbad, n, mldx, 2, blk_stat, n, i0, st));
    else HIP_TRY(ctx, launch_pair_gram(gk, h.B, Bp, X + i0, sv + i0, rv + i0, rb, ldx, X, sv, rv, n, ldx, 2, blk_stat, n, st, 1, 0, 0, 0, i0));
    HIP_TRY(ctx, launch_pair_rows(blk_stat, n, nullptr, nullptr, n, d_rate_class, d_post_rate, d_norm, f, rowcount, tmp, tmp_bytes,
                                  d_rows, capacity, reinterpret_cast<unsigned long long*>(d_count), st, i0, rb,
                                  reinterpret_cast<unsigned long long*>(d_count), with_null ? &nt : nullptr));
  }
  return CMX_OK;
}

// the row blocks of the pair loop: <= 256 MiB of statistics, whole 64-row tiles
static size_t pair_row_block(size_t n, size_t rows) {
  size_t RB = ((size_t)256 << 20) / (8 * n) / 64 * 64;
  return std::max<size_t>(64, std::min<size_t>(RB, (rows + 63) / 64 * 64));
}

cmx_status cmx_intra_gram_prefetch_dev(cmx_ctx* ctx, int kind, const double* d_counts, size_t n, size_t ldc, size_t row_begin,
                                       size_t row_end, void* stream) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if ((s = check_kind(ctx, kind)) != CMX_OK) return s;
  if (!d_counts || n == 0 || ldc < n || n > 0x7fffffffull || row_begin > row_end || row_end > n)
    return fail(ctx, CMX_ERR_INVALID, "cmx_intra_gram_prefetch: bad arguments");
  ctx->gram_kept.valid = false;
  const HostModel& h = ctx->hm;
  const size_t rows = row_end - row_begin;
  // statistics with parameters (mean vectors, thresholds, bounds) and distances: left to the later call
  if (rows == 0 || h.B < 2 || kind == CMX_STAT_CORRECTED_CORRELATION || kind == CMX_STAT_DISCRETE_MI || kind == CMX_STAT_DISCRETE_MI_BOUNDS ||
      kind == CMX_STAT_EUCLIDIAN_DISTANCE)
    return CMX_OK;
  const size_t RB = pair_row_block(n, rows), nblk = (rows + RB - 1) / RB;
  if (nblk * RB * n * sizeof(double) > ((size_t)2 << 30)) return CMX_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  const int Bp = (h.B + 3) / 4 * 4;
  const size_t ldx = (n + 15) / 16 * 16;
  // (scratch of its own: the null's scoring may be using the pair loop's on another stream)
  double *X = nullptr, *sv = nullptr, *rv = nullptr, *kept = nullptr;
  if ((s = scratch(ctx, "gram_X1", sizeof(double) * Bp * ldx, (void**)&X)) != CMX_OK) return s;
  if ((s = scratch(ctx, "gram_s1", sizeof(double) * n, (void**)&sv)) != CMX_OK) return s;
  if ((s = scratch(ctx, "gram_r1", sizeof(double) * n, (void**)&rv)) != CMX_OK) return s;
  if ((s = scratch(ctx, "gram_kept", sizeof(double) * nblk * RB * n, (void**)&kept)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_pair_prep(kind, 0.0, d_counts, n, ldc, h.B, h.K, X, ldx, Bp, sv, rv, nullptr, st));
  for (size_t i0 = row_begin; i0 < row_end; i0 += RB) {
    const size_t rb = std::min(RB, row_end - i0);
    HIP_TRY(ctx, launch_pair_gram(kind, h.B, Bp, X + i0, sv + i0, rv + i0, rb, ldx, X, sv, rv, n, ldx, 2, kept + (i0 - row_begin) * n, n, st, 1, 0, 0, 0, i0));
  }
  ctx->gram_kept = {true, kind, d_counts, n, ldc, row_begin, row_end, kept};
  return CMX_OK;
}

cmx_status cmx_intra_compact_range_dev(cmx_ctx* ctx, int kind, const double* params, const double* d_counts, size_t n, size_t ldc,
                                       const double* d_norm, const double* d_null_stat, const double* d_null_nmin, size_t nnull,
                                       int nclasses, size_t row_begin, size_t row_end, cmx_pair_compact* d_out, size_t capacity,
                                       void* stream) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if ((s = check_kind(ctx, kind)) != CMX_OK) return s;
  const bool with_null = d_null_stat != nullptr;
  if (!d_counts || n == 0 || ldc < n || !d_norm || (capacity && !d_out) || n > 0x7fffffffull || row_begin > row_end || row_end > n ||
      (with_null && (!d_null_nmin || nclasses < 1 || nclasses > 64)) || nnull > 0xfffffff0ull)
    return fail(ctx, CMX_ERR_INVALID, "cmx_intra_compact_range: bad arguments");
  const HostModel& h = ctx->hm;
  if (h.B < 2) return fail(ctx, CMX_ERR_INVALID, "cmx_intra_compact_range: need at least two branches");
  const int gk = kind == CMX_STAT_CORRECTED_CORRELATION ? CMX_STAT_CORRELATION : kind;
  if (gk == CMX_STAT_EUCLIDIAN_DISTANCE) return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_intra_compact_range: EuclidianDistance is a distance, not a statistic");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  if (row_begin == row_end) return CMX_OK;
  // the same operand, null index and row blocks as cmx_intra_rows_range_dev; the pass after each Gram block writes the
  // records at their arithmetic position (no filters: no counting pass, no scan)
  // the Gram blocks may be there already (cmx_intra_gram_prefetch_dev with these arguments): then only the record pass runs
  const cmx_ctx::GramKept gk0 = ctx->gram_kept;
  ctx->gram_kept.valid = false;
  const double* kept = gk0.valid && gk0.kind == kind && gk0.counts == d_counts && gk0.n == n && gk0.ldc == ldc && gk0.row_begin == row_begin &&
                               gk0.row_end == row_end ? gk0.stat : nullptr;
  const double param = (kind == CMX_STAT_DISCRETE_MI) ? (params ? params[0] : 0.99) : 0.0;
  const double* d_mean = nullptr;
  if ((s = stat_mean_vectors(ctx, kind, params, &d_mean, stream)) != CMX_OK) return s;
  const int Bp = (h.B + 3) / 4 * 4;
  const size_t ldx = (n + 15) / 16 * 16;
  double *X = nullptr, *sv = nullptr, *rv = nullptr;
  uint32_t* mcls = nullptr;
  uint8_t* mbad = nullptr;
  size_t mldx = 0;
  if (kept) {
  } else if (kind == CMX_STAT_DISCRETE_MI_BOUNDS) {
    MiBounds mb;
    if ((s = mi_bounds(ctx, params, &mb, stream)) != CMX_OK) return s;
    if ((s = mi_classify(ctx, mb, d_counts, n, ldc, "1", &mcls, &mbad, &mldx, stream)) != CMX_OK) return s;
  } else {
    if ((s = scratch(ctx, "pair_X1", sizeof(double) * Bp * ldx, (void**)&X)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_s1", sizeof(double) * n, (void**)&sv)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_r1", sizeof(double) * n, (void**)&rv)) != CMX_OK) return s;
    HIP_TRY(ctx, launch_pair_prep(gk, param, d_counts, n, ldc, h.B, h.K, X, ldx, Bp, sv, rv, d_mean, st));
  }
  NullTable nt{};
  if (with_null && (s = prepare_null(ctx, d_norm, n, nclasses, d_null_stat, d_null_nmin, nnull, st, &nt)) != CMX_OK) return s;
  const size_t RB = pair_row_block(n, row_end - row_begin);
  double* blk_stat = nullptr;
  if (!kept && (s = scratch(ctx, "blk_stat", sizeof(double) * RB * n, (void**)&blk_stat)) != CMX_OK) return s;
  for (size_t i0 = row_begin; i0 < row_end; i0 += RB) {
    const size_t rb = std::min(RB, row_end - i0);
    const double* blk = kept ? kept + (i0 - row_begin) * n : blk_stat;
    if (kept) {
    } else if (mcls) HIP_TRY(ctx, launch_mi_pairs_block(h.B, mcls + i0, mbad + i0, rb, mldx, mcls, mbad, n, mldx, 2, blk_stat, n, i0, st));
    else HIP_TRY(ctx, launch_pair_gram(gk, h.B, Bp, X + i0, sv + i0, rv + i0, rb, ldx, X, sv, rv, n, ldx, 2, blk_stat, n, st, 1, 0, 0, 0, i0));
    HIP_TRY(ctx, launch_pair_compact(blk, n, n, d_norm, with_null ? &nt : nullptr, d_out, capacity, st, i0, rb, row_begin));
  }
  return CMX_OK;
}

cmx_status cmx_expand_compact_rows(size_t n, size_t row_begin, size_t row_end, const int32_t* rate_class, const double* post_rate,
                                   const double* norm, const cmx_pair_compact* compact, size_t npairs, cmx_pair_row* rows, int nthreads) {
  if (!rate_class || !post_rate || !norm || row_begin > row_end || row_end > n || n > 0x7fffffffull || (npairs && (!compact || !rows)))
    return CMX_ERR_INVALID;
  auto prefix = [n, row_begin](size_t i) { return (i - row_begin) * (n - 1) - (i * (i - 1) - row_begin * (row_begin - 1)) / 2; };
  if (npairs != prefix(row_end)) return CMX_ERR_INVALID;
  auto expand = [&](size_t i_begin, size_t i_end) {
    for (size_t i = i_begin; i < i_end; ++i) {
      const cmx_pair_compact* c = compact + prefix(i);
      cmx_pair_row* r = rows + prefix(i);
      const int32_t ci = rate_class[i];
      const double ri = post_rate[i], ni = norm[i];
      for (size_t j = i + 1; j < n; ++j, ++c, ++r) {
        // the same expressions as pair_rows_kernel / null_pvalue on the device (one IEEE division: bit-identical)
        r->i = (int32_t)i; r->j = (int32_t)j; r->stat = c->stat;
        r->rc_min = ci < rate_class[j] ? ci : rate_class[j];
        r->pr_min = ri < post_rate[j] ? ri : post_rate[j];
        r->n_min = ni < norm[j] ? ni : norm[j];
        if (c->below == 0xffffffffu) { r->pvalue = __builtin_nan(""); r->nsim = 0; }
        else { r->pvalue = (double)(c->nsim - c->below + 1) / (double)(c->nsim + 1); r->nsim = (int32_t)c->nsim; }
      }
    }
  };
  const size_t nrow = row_end - row_begin;
  if (nthreads <= 1 || nrow < 2) { expand(row_begin, row_end); return CMX_OK; }
  // rows cut by pair count: thread t takes the rows whose prefix lies in [t, t + 1) * npairs / nthreads
  std::vector<std::thread> pool;
  size_t i0 = row_begin;
  for (int t = 0; t < nthreads; ++t) {
    const size_t target = (size_t)((unsigned long long)npairs * (t + 1) / nthreads);
    size_t i1 = i0;
    while (i1 < row_end && (t + 1 == nthreads || prefix(i1) < target)) ++i1;
    if (t + 1 == nthreads) i1 = row_end;
    if (i1 > i0) pool.emplace_back(expand, i0, i1);
    i0 = i1;
  }
  for (auto& th : pool) th.join();
  return CMX_OK;
}

cmx_status cmx_intra_rows(cmx_ctx* ctx, int kind, const double* params, const double* counts, size_t n,
                          const int32_t* rate_class, const double* post_rate, const double* norm, const double* null_stat,
                          const double* null_nmin, size_t nnull, int nclasses, const cmx_pair_filters* filters,
                          cmx_pair_row* rows, size_t capacity, uint64_t* count) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (!counts || n == 0 || !rate_class || !post_rate || !norm || !count || (capacity && !rows) ||
      (null_stat && (!null_nmin || nclasses < 1)))
    return fail(ctx, CMX_ERR_INVALID, "cmx_intra_rows: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t BK = (size_t)ctx->hm.B * ctx->hm.K;
  TmpDev tmp;
  std::vector<double> bm;
  to_branch_major(counts, n, BK, &bm);
  double *d_cnt, *d_stat, *d_pr, *d_nm, *d_pv = nullptr, *d_ns = nullptr, *d_nn = nullptr;
  int32_t *d_rc, *d_nsim = nullptr;
  cmx_pair_row* d_rows = nullptr;
  uint64_t* d_count;
  HIP_TRY(ctx, tmp.alloc((void**)&d_cnt, bm.size() * sizeof(double)));
  HIP_TRY(ctx, hipMemcpy(d_cnt, bm.data(), bm.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, tmp.alloc((void**)&d_stat, n * n * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_pr, n * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_nm, n * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_rc, n * sizeof(int32_t)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_count, sizeof(uint64_t)));
  if (capacity) HIP_TRY(ctx, tmp.alloc((void**)&d_rows, capacity * sizeof(cmx_pair_row)));
  HIP_TRY(ctx, hipMemcpy(d_pr, post_rate, n * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_nm, norm, n * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_rc, rate_class, n * sizeof(int32_t), hipMemcpyHostToDevice));
  if ((s = cmx_pair_stats_dev(ctx, kind, params, d_cnt, n, n, nullptr, 0, 0, d_stat, n, nullptr)) != CMX_OK) return s;
  if (null_stat) {
    HIP_TRY(ctx, tmp.alloc((void**)&d_pv, n * n * sizeof(double)));
    HIP_TRY(ctx, tmp.alloc((void**)&d_nsim, n * n * sizeof(int32_t)));
    if (nnull) {
      HIP_TRY(ctx, tmp.alloc((void**)&d_ns, nnull * sizeof(double)));
      HIP_TRY(ctx, tmp.alloc((void**)&d_nn, nnull * sizeof(double)));
      HIP_TRY(ctx, hipMemcpy(d_ns, null_stat, nnull * sizeof(double), hipMemcpyHostToDevice));
      HIP_TRY(ctx, hipMemcpy(d_nn, null_nmin, nnull * sizeof(double), hipMemcpyHostToDevice));
    }
    if ((s = cmx_intra_pvalues_dev(ctx, d_stat, n, d_nm, n, nclasses, d_ns, d_nn, nnull, d_pv, d_nsim, nullptr)) != CMX_OK) return s;
  }
  if ((s = cmx_intra_rows_dev(ctx, d_stat, n, d_pv, d_nsim, n, d_rc, d_pr, d_nm, filters, d_rows, capacity, d_count, nullptr)) != CMX_OK)
    return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(count, d_count, sizeof(uint64_t), hipMemcpyDeviceToHost));
  const size_t nw = std::min<size_t>((size_t)*count, capacity);
  if (nw) HIP_TRY(ctx, hipMemcpy(rows, d_rows, nw * sizeof(cmx_pair_row), hipMemcpyDeviceToHost));
  return CMX_OK;
}

// ------------------------------------------------------------------------------------------------ inter-gene rows
cmx_status cmx_inter_rows_dev(cmx_ctx* ctx, int kind, const double* params, const double* d_counts1, size_t n1, size_t ld1,
                              const int32_t* d_rc1, const double* d_pr1, const double* d_nm1, const double* d_counts2, size_t n2,
                              size_t ld2, const int32_t* d_rc2, const double* d_pr2, const double* d_nm2,
                              const cmx_inter_filters* filters, cmx_pair_row* d_rows, size_t capacity, uint64_t* d_count, void* stream) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if ((s = check_kind(ctx, kind)) != CMX_OK) return s;
  if (!d_counts1 || !d_counts2 || n1 == 0 || n2 == 0 || ld1 < n1 || ld2 < n2 || !d_rc1 || !d_pr1 || !d_nm1 || !d_rc2 || !d_pr2 ||
      !d_nm2 || !d_count || (capacity && !d_rows) || n1 > 0x7fffffffull || n2 > 0x7fffffffull)
    return fail(ctx, CMX_ERR_INVALID, "cmx_inter_rows: bad arguments");
  if (kind == CMX_STAT_EUCLIDIAN_DISTANCE) return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_inter_rows: EuclidianDistance is a distance, not a statistic");
  cmx_inter_filters f{0, 0, -1, 0, 0.0, 0.0, -1.0, 0.0, 0, 0};
  if (filters) f = *filters;
  if (f.independent_comparisons && n1 != n2)   // CoETools.cpp:745-749
    return fail(ctx, CMX_ERR_INVALID, "When performing independant comparisons, the two datasets must have the same length.");
  const HostModel& h = ctx->hm;
  if (h.B < 2) return fail(ctx, CMX_ERR_INVALID, "cmx_inter_rows: need at least two branches");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(ctx, hipMemsetAsync(d_count, 0, sizeof(uint64_t), st));
  unsigned long long* rowcount;
  size_t tmp_bytes = 0;
  void* tmp = nullptr;
  if (f.independent_comparisons) {
    // the n1 pairs (i, i): statistic of the diagonal, then the same filters and compaction
    double* dstat;
    if ((s = scratch(ctx, "inter_diag", sizeof(double) * n1, (void**)&dstat)) != CMX_OK) return s;
    if (kind == CMX_STAT_DISCRETE_MI_BOUNDS) {
      MiBounds mb;
      uint32_t *c1, *c2;
      uint8_t *b1, *b2;
      size_t lx1, lx2;
      if ((s = mi_bounds(ctx, params, &mb, stream)) != CMX_OK) return s;
      if ((s = mi_classify(ctx, mb, d_counts1, n1, ld1, "1", &c1, &b1, &lx1, stream)) != CMX_OK) return s;
      if ((s = mi_classify(ctx, mb, d_counts2, n2, ld2, "2", &c2, &b2, &lx2, stream)) != CMX_OK) return s;
      HIP_TRY(ctx, launch_mi_pairs_diag(h.B, c1, b1, lx1, c2, b2, lx2, n1, dstat, st));
    } else {
      const double param = (kind == CMX_STAT_DISCRETE_MI) ? (params ? params[0] : 0.99) : 0.0;
      const double* d_mean = nullptr;
      if ((s = stat_mean_vectors(ctx, kind, params, &d_mean, stream)) != CMX_OK) return s;
      HIP_TRY(ctx, launch_pair_diag(kind, param, h.B, h.K, d_counts1, ld1, d_counts2, ld2, n1, nullptr, nullptr, nullptr, nullptr, nullptr,
                                    nullptr, dstat, nullptr, nullptr, nullptr, d_mean, st));
    }
    if ((s = scratch(ctx, "rows_count", sizeof(unsigned long long) * (n1 + 1), (void**)&rowcount)) != CMX_OK) return s;
    HIP_TRY(ctx, launch_inter_rows(dstat, 1, n2, d_rc1, d_pr1, d_nm1, d_rc2, d_pr2, d_nm2, f, rowcount, nullptr, tmp_bytes, d_rows, capacity,
                                   reinterpret_cast<unsigned long long*>(d_count), st, 0, n1, nullptr));
    if ((s = scratch(ctx, "rows_scan", tmp_bytes ? tmp_bytes : 16, &tmp)) != CMX_OK) return s;
    HIP_TRY(ctx, launch_inter_rows(dstat, 1, n2, d_rc1, d_pr1, d_nm1, d_rc2, d_pr2, d_nm2, f, rowcount, tmp, tmp_bytes, d_rows, capacity,
                                   reinterpret_cast<unsigned long long*>(d_count), st, 0, n1, nullptr));
    return CMX_OK;
  }
  // operands of both data sets once, then row blocks of data set 1 (dense scratch <= 256 MiB)
  const bool mi = kind == CMX_STAT_DISCRETE_MI_BOUNDS;
  const double param = (kind == CMX_STAT_DISCRETE_MI) ? (params ? params[0] : 0.99) : 0.0;
  const double* d_mean = nullptr;
  const int gk = kind == CMX_STAT_CORRECTED_CORRELATION ? CMX_STAT_CORRELATION : kind;
  const int Bp = (h.B + 3) / 4 * 4;
  const size_t ldx1 = (n1 + 15) / 16 * 16, ldx2 = (n2 + 15) / 16 * 16;
  double *X1 = nullptr, *s1 = nullptr, *r1 = nullptr, *X2 = nullptr, *s2 = nullptr, *r2 = nullptr;
  uint32_t *c1 = nullptr, *c2 = nullptr;
  uint8_t *b1 = nullptr, *b2 = nullptr;
  size_t lx1 = 0, lx2 = 0;
  if (mi) {
    MiBounds mb;
    if ((s = mi_bounds(ctx, params, &mb, stream)) != CMX_OK) return s;
    if ((s = mi_classify(ctx, mb, d_counts1, n1, ld1, "1", &c1, &b1, &lx1, stream)) != CMX_OK) return s;
    if ((s = mi_classify(ctx, mb, d_counts2, n2, ld2, "2", &c2, &b2, &lx2, stream)) != CMX_OK) return s;
  } else {
    if ((s = stat_mean_vectors(ctx, kind, params, &d_mean, stream)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_X1", sizeof(double) * Bp * ldx1, (void**)&X1)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_s1", sizeof(double) * n1, (void**)&s1)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_r1", sizeof(double) * n1, (void**)&r1)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_X2", sizeof(double) * Bp * ldx2, (void**)&X2)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_s2", sizeof(double) * n2, (void**)&s2)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_r2", sizeof(double) * n2, (void**)&r2)) != CMX_OK) return s;
    HIP_TRY(ctx, launch_pair_prep(gk, param, d_counts1, n1, ld1, h.B, h.K, X1, ldx1, Bp, s1, r1, d_mean, st));
    HIP_TRY(ctx, launch_pair_prep(gk, param, d_counts2, n2, ld2, h.B, h.K, X2, ldx2, Bp, s2, r2, d_mean ? d_mean + h.B : nullptr, st));
  }
  size_t RB = ((size_t)256 << 20) / (8 * n2) / 64 * 64;
  RB = std::max<size_t>(64, std::min<size_t>(RB, (n1 + 63) / 64 * 64));
  double* blk;
  if ((s = scratch(ctx, "blk_stat", sizeof(double) * RB * n2, (void**)&blk)) != CMX_OK) return s;
  if ((s = scratch(ctx, "rows_count", sizeof(unsigned long long) * (RB + 1), (void**)&rowcount)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_inter_rows(blk, n2, n2, d_rc1, d_pr1, d_nm1, d_rc2, d_pr2, d_nm2, f, rowcount, nullptr, tmp_bytes, d_rows, capacity,
                                 reinterpret_cast<unsigned long long*>(d_count), st, 0, RB, nullptr));
  if ((s = scratch(ctx, "rows_scan", tmp_bytes ? tmp_bytes : 16, &tmp)) != CMX_OK) return s;
  for (size_t i0 = 0; i0 < n1; i0 += RB) {
    const size_t rb = std::min(RB, n1 - i0);
    if (mi) HIP_TRY(ctx, launch_mi_pairs_block(h.B, c1 + i0, b1 + i0, rb, lx1, c2, b2, n2, lx2, 0, blk, n2, i0, st));
    else HIP_TRY(ctx, launch_pair_gram(gk, h.B, Bp, X1 + i0, s1 + i0, r1 + i0, rb, ldx1, X2, s2, r2, n2, ldx2, 0, blk, n2, st));
    HIP_TRY(ctx, launch_inter_rows(blk, n2, n2, d_rc1, d_pr1, d_nm1, d_rc2, d_pr2, d_nm2, f, rowcount, tmp, tmp_bytes, d_rows, capacity,
                                   reinterpret_cast<unsigned long long*>(d_count), st, i0, rb, reinterpret_cast<unsigned long long*>(d_count)));
  }
  return CMX_OK;
}

cmx_status cmx_inter_rows(cmx_ctx* ctx, int kind, const double* params, const double* counts1, size_t n1, const int32_t* rate_class1,
                          const double* post_rate1, const double* norm1, const double* counts2, size_t n2, const int32_t* rate_class2,
                          const double* post_rate2, const double* norm2, const cmx_inter_filters* filters, cmx_pair_row* rows,
                          size_t capacity, uint64_t* count) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (!counts1 || !counts2 || n1 == 0 || n2 == 0 || !rate_class1 || !post_rate1 || !norm1 || !rate_class2 || !post_rate2 || !norm2 ||
      !count || (capacity && !rows))
    return fail(ctx, CMX_ERR_INVALID, "cmx_inter_rows: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t BK = (size_t)ctx->hm.B * ctx->hm.K;
  TmpDev tmp;
  std::vector<double> bm;
  double *d_c[2], *d_pr[2], *d_nm[2];
  int32_t* d_rc[2];
  const double* cs[2] = {counts1, counts2};
  const size_t ns[2] = {n1, n2};
  const int32_t* rcs[2] = {rate_class1, rate_class2};
  const double *prs[2] = {post_rate1, post_rate2}, *nms[2] = {norm1, norm2};
  for (int q = 0; q < 2; ++q) {
    to_branch_major(cs[q], ns[q], BK, &bm);
    HIP_TRY(ctx, tmp.alloc((void**)&d_c[q], bm.size() * sizeof(double)));
    HIP_TRY(ctx, hipMemcpy(d_c[q], bm.data(), bm.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(ctx, tmp.alloc((void**)&d_pr[q], ns[q] * sizeof(double)));
    HIP_TRY(ctx, tmp.alloc((void**)&d_nm[q], ns[q] * sizeof(double)));
    HIP_TRY(ctx, tmp.alloc((void**)&d_rc[q], ns[q] * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemcpy(d_pr[q], prs[q], ns[q] * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(d_nm[q], nms[q], ns[q] * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(d_rc[q], rcs[q], ns[q] * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  cmx_pair_row* d_rows = nullptr;
  uint64_t* d_count;
  HIP_TRY(ctx, tmp.alloc((void**)&d_count, sizeof(uint64_t)));
  if (capacity) HIP_TRY(ctx, tmp.alloc((void**)&d_rows, capacity * sizeof(cmx_pair_row)));
  if ((s = cmx_inter_rows_dev(ctx, kind, params, d_c[0], n1, n1, d_rc[0], d_pr[0], d_nm[0], d_c[1], n2, n2, d_rc[1], d_pr[1], d_nm[1],
                              filters, d_rows, capacity, d_count, nullptr)) != CMX_OK)
    return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(count, d_count, sizeof(uint64_t), hipMemcpyDeviceToHost));
  const size_t nw = std::min<size_t>((size_t)*count, capacity);
  if (nw) HIP_TRY(ctx, hipMemcpy(rows, d_rows, nw * sizeof(cmx_pair_row), hipMemcpyDeviceToHost));
  return CMX_OK;
}

// ------------------------------------------------------------------------------------------------ Mica MI
cmx_status cmx_mi_columns_dev(cmx_ctx* ctx, int nalpha, int ntaxa, const uint32_t* d_masks, const uint8_t* d_aln1,
                              size_t n1, size_t ld1, const uint8_t* d_aln2, size_t n2, size_t ld2, double* d_mi,
                              double* d_hjoint, size_t ldo, double* d_h1, double* d_h2, void* stream) {
  if (!ctx) return CMX_ERR_INVALID;
  if (nalpha != 4 && nalpha != 20) return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_mi_columns: alphabet size must be 4 or 20");
  const bool intra = d_aln2 == nullptr;
  if (intra) { d_aln2 = d_aln1; n2 = n1; ld2 = ld1; }
  if (!d_masks) {  // no ambiguity table: every code >= nalpha is "unknown" (compatible with all states)
    std::vector<uint32_t> mk(256, (1u << nalpha) - 1u);
    for (int i = 0; i < nalpha; ++i) mk[i] = 1u << i;
    void* p = nullptr;
    cmx_status s = scratch(ctx, "mi_masks", 256 * sizeof(uint32_t), &p);
    if (s != CMX_OK) return s;
    HIP_TRY(ctx, hipMemcpy(p, mk.data(), 256 * sizeof(uint32_t), hipMemcpyHostToDevice));
    d_masks = static_cast<const uint32_t*>(p);
  }
  if (!d_aln1 || !d_mi || !d_hjoint || ntaxa < 1 || n1 == 0 || n2 == 0 || ld1 < n1 || ld2 < n2 || ldo < n2)
    return fail(ctx, CMX_ERR_INVALID, "cmx_mi_columns: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // MFMA path (one-hot Gram) whenever the c ln c table fits the kernel's LDS; columns with ambiguous symbols are
  // left to the LDS-table kernel pair by pair
  MicaWork w{};
  const bool mfma = ntaxa <= 2047;   // table + operand buffers within 64 KiB of LDS
  if (mfma) {
    cmx_status s;
    w.Tp = (ntaxa + 31) / 32 * 32;
    const size_t hb = 32 * (size_t)w.Tp;
    const bool needH = mica_needs_onehot(nalpha, w.Tp);   // 32 Tp bytes per column: only where a kernel reads them
    if ((s = scratch(ctx, "mica_H1", needH ? hb * n1 : 16, (void**)&w.H1)) != CMX_OK) return s;
    if ((s = scratch(ctx, "mica_C1", (size_t)w.Tp * (n1 + kMicaCodePad), (void**)&w.C1)) != CMX_OK) return s;
    if ((s = scratch(ctx, "mica_f1", n1, (void**)&w.flag1)) != CMX_OK) return s;
    if ((s = scratch(ctx, "mica_g1", n1, (void**)&w.gap1)) != CMX_OK) return s;
    if ((s = scratch(ctx, "mica_S1", sizeof(double) * n1, (void**)&w.S1)) != CMX_OK) return s;
    if (!intra) {
      if ((s = scratch(ctx, "mica_H2", needH ? hb * n2 : 16, (void**)&w.H2)) != CMX_OK) return s;
      if ((s = scratch(ctx, "mica_C2", (size_t)w.Tp * (n2 + kMicaCodePad), (void**)&w.C2)) != CMX_OK) return s;
      if ((s = scratch(ctx, "mica_f2", n2, (void**)&w.flag2)) != CMX_OK) return s;
      if ((s = scratch(ctx, "mica_g2", n2, (void**)&w.gap2)) != CMX_OK) return s;
      if ((s = scratch(ctx, "mica_S2", sizeof(double) * n2, (void**)&w.S2)) != CMX_OK) return s;
    }
    if ((s = scratch(ctx, "mica_ftab", sizeof(double) * ((size_t)(ntaxa + 1) + 2 * ((size_t)nalpha * nalpha * ntaxa + 1) + 2), (void**)&w.ftab)) != CMX_OK) return s;
    if ((s = scratch(ctx, "mica_any", sizeof(int), (void**)&w.anyflag)) != CMX_OK) return s;
    if (nalpha == 20) {   // block info of the four-wave kernel, padded to whole tiles of 12 columns
      if ((s = scratch(ctx, "mica_info1", sizeof(unsigned) * ((n1 + 11) / 12 * 4 + 4), (void**)&w.info1)) != CMX_OK) return s;
      if ((s = scratch(ctx, "mica_order1", sizeof(unsigned) * n1, (void**)&w.order1)) != CMX_OK) return s;
      if ((s = scratch(ctx, "mica_Cs1", (size_t)w.Tp * (n1 + kMicaCodePad), (void**)&w.Cs1)) != CMX_OK) return s;
      if ((s = scratch(ctx, "mica_Ss1", sizeof(double) * n1, (void**)&w.Ss1)) != CMX_OK) return s;
      if (mica4_serves(nalpha, w.Tp, n1, intra ? n1 : n2) && (s = scratch(ctx, "mica_img2", mica4_image_bytes(w.Tp, intra ? n1 : n2), &w.img2)) != CMX_OK) return s;
      if (!intra) {
        if ((s = scratch(ctx, "mica_info2", sizeof(unsigned) * ((n2 + 11) / 12 * 4 + 4), (void**)&w.info2)) != CMX_OK) return s;
        if ((s = scratch(ctx, "mica_order2", sizeof(unsigned) * n2, (void**)&w.order2)) != CMX_OK) return s;
        if ((s = scratch(ctx, "mica_Cs2", (size_t)w.Tp * (n2 + kMicaCodePad), (void**)&w.Cs2)) != CMX_OK) return s;
        if ((s = scratch(ctx, "mica_Ss2", sizeof(double) * n2, (void**)&w.Ss2)) != CMX_OK) return s;
      }
    }
  }
  HIP_TRY(ctx, launch_mi_columns(nalpha, ntaxa, d_masks, d_aln1, n1, ld1, d_aln2, n2, ld2, intra ? 1 : 0, d_mi, d_hjoint,
                                 ldo, d_h1, intra ? nullptr : d_h2, mfma ? &w : nullptr, (hipStream_t)stream));
  return CMX_OK;
}

cmx_status cmx_mi_columns(cmx_ctx* ctx, int nalpha, int ntaxa, const uint32_t* masks, size_t nmasks, const uint8_t* aln1,
                          size_t n1, const uint8_t* aln2, size_t n2, double* mi, double* hjoint, double* h1, double* h2) {
  if (!ctx) return CMX_ERR_INVALID;
  if (!aln1 || !mi || !hjoint || n1 == 0 || ntaxa < 1 || nalpha < 2 || nalpha > 31) return fail(ctx, CMX_ERR_INVALID, "cmx_mi_columns: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!aln2) n2 = n1;
  std::vector<uint32_t> mk(256, (1u << nalpha) - 1u);
  for (int i = 0; i < nalpha; ++i) mk[i] = 1u << i;
  if (masks) for (size_t i = 0; i < nmasks && i < 256; ++i) mk[i] = masks[i];
  for (size_t i = 0; i < 256; ++i) if (mk[i] == 0) mk[i] = (1u << nalpha) - 1u;
  TmpDev tmp;
  uint32_t* d_masks;
  uint8_t *d1, *d2 = nullptr;
  double *d_mi, *d_hj, *d_h1, *d_h2 = nullptr;
  HIP_TRY(ctx, tmp.alloc((void**)&d_masks, 256 * sizeof(uint32_t)));
  HIP_TRY(ctx, hipMemcpy(d_masks, mk.data(), 256 * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(ctx, tmp.alloc((void**)&d1, (size_t)ntaxa * n1));
  HIP_TRY(ctx, hipMemcpy(d1, aln1, (size_t)ntaxa * n1, hipMemcpyHostToDevice));
  if (aln2) {
    HIP_TRY(ctx, tmp.alloc((void**)&d2, (size_t)ntaxa * n2));
    HIP_TRY(ctx, hipMemcpy(d2, aln2, (size_t)ntaxa * n2, hipMemcpyHostToDevice));
    HIP_TRY(ctx, tmp.alloc((void**)&d_h2, n2 * sizeof(double)));
  }
  HIP_TRY(ctx, tmp.alloc((void**)&d_mi, n1 * n2 * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_hj, n1 * n2 * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_h1, n1 * sizeof(double)));
  cmx_status s = cmx_mi_columns_dev(ctx, nalpha, ntaxa, d_masks, d1, n1, n1, d2, n2, n2, d_mi, d_hj, n2, d_h1, d_h2, nullptr);
  if (s != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(mi, d_mi, n1 * n2 * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(hjoint, d_hj, n1 * n2 * sizeof(double), hipMemcpyDeviceToHost));
  if (h1) HIP_TRY(ctx, hipMemcpy(h1, d_h1, n1 * sizeof(double), hipMemcpyDeviceToHost));
  if (h2) HIP_TRY(ctx, hipMemcpy(h2, aln2 ? d_h2 : d_h1, n2 * sizeof(double), hipMemcpyDeviceToHost));
  return CMX_OK;
}

// d_masks: device table of 256 compatibility masks (NULL: codes >= nalpha are unknowns); column indices are validated by
// the caller (the host entry point checks them)
cmx_status cmx_mi_pairs_dev(cmx_ctx* ctx, int nalpha, int ntaxa, const uint32_t* d_masks, const uint8_t* d_aln1, size_t n1, size_t ld1,
                            const uint8_t* d_aln2, size_t n2, size_t ld2, const int64_t* d_idx1, const int64_t* d_idx2, size_t npairs,
                            double* d_mi, double* d_hjoint, void* stream) {
  if (!ctx) return CMX_ERR_INVALID;
  if (nalpha != 4 && nalpha != 20) return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_mi_pairs: alphabet size must be 4 or 20");
  if (!d_aln2) { d_aln2 = d_aln1; n2 = n1; ld2 = ld1; }
  if (!d_aln1 || !d_idx1 || !d_idx2 || !d_mi || !d_hjoint || n1 == 0 || n2 == 0 || ld1 < n1 || ld2 < n2 || ntaxa < 1 || npairs == 0)
    return fail(ctx, CMX_ERR_INVALID, "cmx_mi_pairs: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!d_masks) {
    std::vector<uint32_t> mk(256, (1u << nalpha) - 1u);
    for (int i = 0; i < nalpha; ++i) mk[i] = 1u << i;
    void* p = nullptr;
    cmx_status s = scratch(ctx, "mi_masks", 256 * sizeof(uint32_t), &p);
    if (s != CMX_OK) return s;
    HIP_TRY(ctx, hipMemcpy(p, mk.data(), 256 * sizeof(uint32_t), hipMemcpyHostToDevice));
    d_masks = static_cast<const uint32_t*>(p);
  }
  HIP_TRY(ctx, launch_mi_pairs(nalpha, ntaxa, d_masks, d_aln1, ld1, d_aln2, ld2, d_idx1, d_idx2, npairs, d_mi, d_hjoint, (hipStream_t)stream));
  return CMX_OK;
}

cmx_status cmx_mi_pairs(cmx_ctx* ctx, int nalpha, int ntaxa, const uint32_t* masks, size_t nmasks, const uint8_t* aln1,
                        size_t n1, const uint8_t* aln2, size_t n2, const int64_t* idx1, const int64_t* idx2, size_t npairs,
                        double* mi, double* hjoint) {
  if (!ctx) return CMX_ERR_INVALID;
  if (nalpha != 4 && nalpha != 20) return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_mi_pairs: alphabet size must be 4 or 20");
  if (!aln1 || !idx1 || !idx2 || !mi || !hjoint || n1 == 0 || ntaxa < 1 || npairs == 0)
    return fail(ctx, CMX_ERR_INVALID, "cmx_mi_pairs: bad arguments");
  if (!aln2) n2 = n1;
  for (size_t p = 0; p < npairs; ++p)
    if (idx1[p] < 0 || (size_t)idx1[p] >= n1 || idx2[p] < 0 || (size_t)idx2[p] >= n2)
      return fail(ctx, CMX_ERR_INVALID, "cmx_mi_pairs: column index out of range");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::vector<uint32_t> mk(256, (1u << nalpha) - 1u);
  for (int i = 0; i < nalpha; ++i) mk[i] = 1u << i;
  if (masks) for (size_t i = 0; i < nmasks && i < 256; ++i) mk[i] = masks[i];
  for (size_t i = 0; i < 256; ++i) if (mk[i] == 0) mk[i] = (1u << nalpha) - 1u;
  TmpDev tmp;
  uint32_t* d_masks;
  uint8_t *d1, *d2 = nullptr;
  int64_t *di1, *di2;
  double *d_mi, *d_hj;
  HIP_TRY(ctx, tmp.alloc((void**)&d_masks, 256 * sizeof(uint32_t)));
  HIP_TRY(ctx, hipMemcpy(d_masks, mk.data(), 256 * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(ctx, tmp.alloc((void**)&d1, (size_t)ntaxa * n1));
  HIP_TRY(ctx, hipMemcpy(d1, aln1, (size_t)ntaxa * n1, hipMemcpyHostToDevice));
  if (aln2) {
    HIP_TRY(ctx, tmp.alloc((void**)&d2, (size_t)ntaxa * n2));
    HIP_TRY(ctx, hipMemcpy(d2, aln2, (size_t)ntaxa * n2, hipMemcpyHostToDevice));
  }
  HIP_TRY(ctx, tmp.alloc((void**)&di1, npairs * sizeof(int64_t)));
  HIP_TRY(ctx, tmp.alloc((void**)&di2, npairs * sizeof(int64_t)));
  HIP_TRY(ctx, hipMemcpy(di1, idx1, npairs * sizeof(int64_t), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(di2, idx2, npairs * sizeof(int64_t), hipMemcpyHostToDevice));
  HIP_TRY(ctx, tmp.alloc((void**)&d_mi, npairs * sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_hj, npairs * sizeof(double)));
  cmx_status s = cmx_mi_pairs_dev(ctx, nalpha, ntaxa, d_masks, d1, n1, n1, d2, n2, n2, di1, di2, npairs, d_mi, d_hj, nullptr);
  if (s != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(mi, d_mi, npairs * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(hjoint, d_hj, npairs * sizeof(double), hipMemcpyDeviceToHost));
  return CMX_OK;
}


// ---- Mica's bootstrap nulls.  Site indices of the non-parametric bootstrap (SiteContainerTools::sampleSites,
// CoMap/Mica.cpp:426-430) come from the engine's counter RNG (cmx_kernels.hip philox_uniform: Philox2x32-10, key from the
// seed, counter = (g, draw)), so that every binding -- this library's C++ adapter, the Python mirror, a Mica.cpp linked
// against the C-ABI -- draws the same pairs: idx_h[r * rep_ram + j] = floor(u(seed, g = (r * 2 + h) * rep_ram + j, draw 0) * nsites).
static double host_philox_uniform(uint64_t seed, uint64_t g, uint32_t draw) {
  uint32_t c0 = (uint32_t)g, c1 = ((uint32_t)(g >> 32) & 0x7fffu) | (draw << 15);
  uint32_t k = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u) ^ 0x434d5832u;
  for (int r = 0; r < 10; ++r) {
    const uint64_t p = (uint64_t)0xD256D193u * (uint64_t)c0;
    c0 = (uint32_t)(p >> 32) ^ k ^ c1;
    c1 = (uint32_t)p;
    k += 0x9E3779B9u;
  }
  const uint64_t bits = (((uint64_t)c0 << 32) | c1) >> 11;
  return (double)bits * (1.0 / 9007199254740992.0);
}

cmx_status cmx_mica_bootstrap_indices(uint64_t seed, size_t nsites, size_t nrep_cpu, size_t nrep_ram, int64_t* idx1, int64_t* idx2) {
  if (nsites == 0 || !idx1 || !idx2 || (uint64_t)nrep_cpu * 2 * nrep_ram > (1ull << 47)) return CMX_ERR_INVALID;
  for (size_t r = 0; r < nrep_cpu; ++r)
    for (size_t j = 0; j < nrep_ram; ++j)
      for (int h = 0; h < 2; ++h) {
        size_t v = (size_t)(host_philox_uniform(seed, ((uint64_t)r * 2 + h) * nrep_ram + j, 0) * (double)nsites);
        if (v >= nsites) v = nsites - 1;
        (h ? idx2 : idx1)[r * nrep_ram + j] = (int64_t)v;
      }
  return CMX_OK;
}

// null.method = parametric-bootstrap (CoMap/Mica.cpp:469-548): per replicate two alignments of nrep_ram sites are simulated
// under the context's model, column j of the one is scored against column j of the other (MI, joint entropy), and -- Mica's
// `use_model` case -- both are mapped for their norms.  One simulation, one MI launch and one mapping over all replicates,
// none of it leaving the device; only the null's columns come back.
cmx_status cmx_mica_parametric_null(cmx_ctx* ctx, int nalpha, uint64_t seed, size_t nrep_cpu, size_t nrep_ram, double gamma_alpha,
                                    double p_invariant, double* mi, double* hjoint, double* nmin) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (nrep_cpu == 0 || nrep_ram == 0 || !mi || !hjoint) return fail(ctx, CMX_ERR_INVALID, "cmx_mica_parametric_null: bad arguments");
  if (nalpha != ctx->hm.S) return fail(ctx, CMX_ERR_INVALID, "cmx_mica_parametric_null: the alphabet is the model's");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n = nrep_cpu * nrep_ram, T = (size_t)ctx->hm.T;
  TmpDev tmp;
  uint8_t* d_aln;
  int64_t *d_i1, *d_i2;
  double *d_mi, *d_hj, *d_norm = nullptr;
  HIP_TRY(ctx, tmp.alloc((void**)&d_aln, T * 2 * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_i1, sizeof(int64_t) * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_i2, sizeof(int64_t) * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_mi, sizeof(double) * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_hj, sizeof(double) * n));
  // simulated-site index g = (rep * 2 + batch) * nrep_ram + j = its column in the [T][2 n] alignment
  if (gamma_alpha > 0.0) s = cmx_simulate_continuous_dev(ctx, seed, 0, 2 * n, gamma_alpha, p_invariant, d_aln, 2 * n, nullptr, nullptr);
  else s = cmx_simulate_dev(ctx, seed, 0, 2 * n, d_aln, 2 * n, nullptr, nullptr);
  if (s != CMX_OK) return s;
  std::vector<int64_t> i1(n), i2(n);
  for (size_t q = 0; q < n; ++q) {
    const size_t rep = q / nrep_ram, j = q % nrep_ram;
    i1[q] = (int64_t)((rep * 2) * nrep_ram + j);
    i2[q] = (int64_t)((rep * 2 + 1) * nrep_ram + j);
  }
  HIP_TRY(ctx, hipMemcpy(d_i1, i1.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_i2, i2.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice));
  if ((s = cmx_mi_pairs_dev(ctx, nalpha, (int)T, nullptr, d_aln, 2 * n, 2 * n, nullptr, 0, 0, d_i1, d_i2, n, d_mi, d_hj, nullptr)) != CMX_OK) return s;
  if (nmin) {
    HIP_TRY(ctx, tmp.alloc((void**)&d_norm, sizeof(double) * 2 * n));
    if ((s = map_sites_impl(ctx, d_aln, 2 * n, 2 * n, nullptr, nullptr, 0, nullptr, nullptr, nullptr, d_norm, nullptr, true)) != CMX_OK) return s;
  }
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(mi, d_mi, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(hjoint, d_hj, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (nmin) {
    std::vector<double> norm(2 * n);
    HIP_TRY(ctx, hipMemcpy(norm.data(), d_norm, sizeof(double) * 2 * n, hipMemcpyDeviceToHost));
    for (size_t q = 0; q < n; ++q) nmin[q] = std::min(norm[(size_t)i1[q]], norm[(size_t)i2[q]]);
  }
  return CMX_OK;
}

// ------------------------------------------------------------------------------------------------ groups of sites
cmx_status cmx_group_stats_dev(cmx_ctx* ctx, int kind, const double* params, const double* d_counts, size_t n, size_t ldc,
                               const int64_t* d_offsets, const int32_t* d_sites, size_t ngroups, double* d_out, void* stream) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if ((s = check_kind(ctx, kind)) != CMX_OK) return s;
  if (!d_counts || n == 0 || ldc < n || !d_offsets || !d_sites || !d_out) return fail(ctx, CMX_ERR_INVALID, "cmx_group_stats: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (kind == CMX_STAT_DISCRETE_MI_BOUNDS) {
    MiBounds mb;
    uint32_t* cls;
    uint8_t* bad;
    size_t ldx;
    if ((s = mi_bounds(ctx, params, &mb, stream)) != CMX_OK) return s;
    if ((s = mi_classify(ctx, mb, d_counts, n, ldc, "g", &cls, &bad, &ldx, stream)) != CMX_OK) return s;
    HIP_TRY(ctx, launch_mi_group(ctx->hm.B, cls, bad, ldx, d_offsets, d_sites, ngroups, d_out, (hipStream_t)stream));
    return CMX_OK;
  }
  const double param = (kind == CMX_STAT_DISCRETE_MI) ? (params ? params[0] : 0.99) : 0.0;
  const double* d_mean = nullptr;
  if ((s = stat_mean_vectors(ctx, kind, params, &d_mean, stream)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_group_stats(kind, param, ctx->hm.B, ctx->hm.K, d_counts, ldc, d_offsets, d_sites, ngroups, d_out, d_mean,
                                  (hipStream_t)stream));
  return CMX_OK;
}

static cmx_status check_groups(cmx_ctx* ctx, const int64_t* offsets, const int32_t* sites, size_t ngroups, size_t n) {
  if (offsets[0] != 0) return fail(ctx, CMX_ERR_INVALID, "groups: offsets[0] must be 0");
  for (size_t g = 0; g < ngroups; ++g)
    if (offsets[g + 1] < offsets[g]) return fail(ctx, CMX_ERR_INVALID, "groups: offsets must not decrease");
  if (sites)
    for (int64_t q = 0; q < offsets[ngroups]; ++q)
      if (sites[q] < 0 || (size_t)sites[q] >= n) return fail(ctx, CMX_ERR_INVALID, "groups: site index out of range");
  return CMX_OK;
}

cmx_status cmx_group_stats(cmx_ctx* ctx, int kind, const double* params, const double* counts, size_t n, const int64_t* offsets,
                           const int32_t* sites, size_t ngroups, double* out) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if (!counts || n == 0 || !offsets || !sites || !out) return fail(ctx, CMX_ERR_INVALID, "cmx_group_stats: bad arguments");
  if (ngroups == 0) return CMX_OK;
  if ((s = check_groups(ctx, offsets, sites, ngroups, n)) != CMX_OK) return s;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t BK = (size_t)ctx->hm.B * ctx->hm.K;
  std::vector<double> bm;
  to_branch_major(counts, n, BK, &bm);
  TmpDev tmp;
  double *d_c, *d_out;
  int64_t* d_off;
  int32_t* d_sites;
  HIP_TRY(ctx, tmp.alloc((void**)&d_c, sizeof(double) * bm.size()));
  HIP_TRY(ctx, tmp.alloc((void**)&d_out, sizeof(double) * ngroups));
  HIP_TRY(ctx, tmp.alloc((void**)&d_off, sizeof(int64_t) * (ngroups + 1)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_sites, sizeof(int32_t) * (size_t)offsets[ngroups]));
  HIP_TRY(ctx, hipMemcpy(d_c, bm.data(), sizeof(double) * bm.size(), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_off, offsets, sizeof(int64_t) * (ngroups + 1), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_sites, sites, sizeof(int32_t) * (size_t)offsets[ngroups], hipMemcpyHostToDevice));
  if ((s = cmx_group_stats_dev(ctx, kind, params, d_c, n, n, d_off, d_sites, ngroups, d_out, nullptr)) != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, d_out, sizeof(double) * ngroups, hipMemcpyDeviceToHost));
  return CMX_OK;
}

namespace {
// The bookkeeping of CandidateGroupSet (CoMap/CoETools.h:139-300, CoETools.cpp:900-1038) on norms alone: which
// simulated site goes to which candidate site, and which pseudo-groups are thereby completed.  Whether a completed
// pseudo-group also counts for n1 needs its statistic -- evaluated afterwards for the whole batch on the device.
struct CandidateCursor {
  size_t G = 0;
  const int64_t* off = nullptr;
  const double *lo = nullptr, *hi = nullptr;
  const uint8_t* usable = nullptr;
  uint32_t min_sim = 0, completed = 0, n_usable = 0, trials = 0;
  size_t gpos = 0, spos = 0;                       // the reference's groupPos_ / sitePos_ (kept across batches)
  std::vector<uint32_t> n2;
  std::vector<std::vector<int32_t>> waiting;       // per candidate site: simulated sites not yet used, oldest first
  std::vector<size_t> head;                        // per candidate site: first unused entry of `waiting`
  std::vector<int64_t> pg_off;                     // completed pseudo-groups of the current batch
  std::vector<int32_t> pg_sites, pg_group;

  size_t gsize(size_t g) const { return (size_t)(off[g + 1] - off[g]); }
  bool open(size_t g) const { return n2[g] < min_sim && usable[g]; }
  // nextCandidateSite: step to the next site of the current group (or the next group), then skip groups that are
  // complete or not analysable.  false: no group is left (the reference throws)
  bool advance() {
    if (n2[gpos] < min_sim && ++spos >= gsize(gpos)) { gpos = (gpos + 1) % G; spos = 0; }
    if (!open(gpos)) {
      const size_t start = gpos;
      do {
        gpos = (gpos + 1) % G;
        if (gpos == start) return false;
      } while (!open(gpos));
      spos = 0;
    }
    return true;
  }
  // addSimulatedSite: true if the group now has one simulated site for each of its members
  bool give(size_t g, size_t sidx, int32_t sim) {
    waiting[off[g] + sidx].push_back(sim);
    for (size_t q = off[g]; q < (size_t)off[g + 1]; ++q)
      if (head[q] >= waiting[q].size()) return false;
    for (size_t q = off[g]; q < (size_t)off[g + 1]; ++q) pg_sites.push_back(waiting[q][head[q]++]);
    pg_off.push_back((int64_t)pg_sites.size());
    pg_group.push_back((int32_t)g);
    if (++n2[g] == min_sim) ++completed;
    return true;
  }
  // analyseSimulations: 1 more batches needed, 0 done, -1 cursor error
  int batch(const double* norms, size_t nsim) {
    pg_off.assign(1, 0); pg_sites.clear(); pg_group.clear();
    bool more = true, nothing = true;
    for (size_t i = 0; more && i < nsim; ++i) {
      bool first = true, hit = false;
      size_t g0 = 0, s0 = 0;
      while (more && !hit) {
        if (!advance()) return -1;
        if (first) { g0 = gpos; s0 = spos; first = false; }
        else if (gpos == g0 && spos == s0) break;              // went round the whole set: this site fits nowhere
        const size_t q = off[gpos] + spos;
        hit = norms[i] >= lo[q] && norms[i] <= hi[q];
        if (hit) {
          if (give(gpos, spos, (int32_t)i)) nothing = false;
          if (completed == n_usable) more = false;
        }
      }
    }
    if (nothing) ++trials;
    for (size_t q = 0; q < waiting.size(); ++q) { waiting[q].clear(); head[q] = 0; }   // resetSimulations
    return more ? 1 : 0;
  }
};
}  // namespace

// host-side only (no GPU): run the candidate cursor over caller-supplied norms, batch by batch, and list the
// pseudo-groups it assembles.  For tests of the bookkeeping.
cmx_status cmx_debug_candidate_cursor(size_t ngroups, const int64_t* offsets, const double* norm_lo, const double* norm_hi,
                                      const uint8_t* analysable, uint32_t min_sim, const double* norms, size_t rep_ram,
                                      size_t nbatches, uint32_t max_trials, uint32_t* n2, uint32_t* trials,
                                      uint64_t* batches_used, int32_t* pg_group, int32_t* pg_batch, int64_t* pg_offsets,
                                      int32_t* pg_sites, size_t cap_groups, size_t cap_sites, size_t* npg) {
  if (ngroups == 0 || !offsets || !norm_lo || !norm_hi || !analysable || min_sim == 0 || !norms || rep_ram == 0 || !n2 || !npg)
    return CMX_ERR_INVALID;
  CandidateCursor cur;
  cur.G = ngroups; cur.off = offsets; cur.lo = norm_lo; cur.hi = norm_hi; cur.usable = analysable; cur.min_sim = min_sim;
  cur.n2.assign(ngroups, 0);
  cur.waiting.resize((size_t)offsets[ngroups]);
  cur.head.assign((size_t)offsets[ngroups], 0);
  for (size_t g = 0; g < ngroups; ++g)
    if (analysable[g]) { if (cur.gsize(g) == 0) return CMX_ERR_INVALID; ++cur.n_usable; }
  if (cur.n_usable == 0) return CMX_ERR_INVALID;
  size_t ng = 0, ns = 0;
  uint64_t nb = 0;
  int more = 1;
  if (pg_offsets && cap_groups) pg_offsets[0] = 0;
  while (more == 1 && cur.trials < max_trials && nb < nbatches) {
    more = cur.batch(norms + nb * rep_ram, rep_ram);
    if (more < 0) return CMX_ERR_INVALID;
    for (size_t q = 0; q < cur.pg_group.size(); ++q, ++ng) {
      const size_t m = (size_t)(cur.pg_off[q + 1] - cur.pg_off[q]);
      if (ng < cap_groups && ns + m <= cap_sites) {
        pg_group[ng] = cur.pg_group[q];
        pg_batch[ng] = (int32_t)nb;
        for (size_t e = 0; e < m; ++e) pg_sites[ns + e] = cur.pg_sites[cur.pg_off[q] + e];
        pg_offsets[ng + 1] = (int64_t)(ns + m);
      }
      ns += m;
    }
    ++nb;
  }
  std::copy(cur.n2.begin(), cur.n2.end(), n2);
  if (trials) *trials = cur.trials;
  if (batches_used) *batches_used = nb;
  *npg = ng;
  return CMX_OK;
}

cmx_status cmx_candidate_groups(cmx_ctx* ctx, int kind, const double* params, size_t ngroups, const int64_t* offsets,
                                const double* norm_lo, const double* norm_hi, const uint8_t* analysable,
                                const double* observed, uint32_t min_sim, size_t rep_ram, uint32_t max_trials,
                                uint64_t max_batches, uint64_t seed, uint32_t* n1, uint32_t* n2, uint32_t* trials,
                                uint64_t* batches) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if ((s = check_kind(ctx, kind)) != CMX_OK) return s;
  if (ngroups == 0 || !offsets || !norm_lo || !norm_hi || !analysable || !observed || min_sim == 0 || rep_ram == 0 || !n1 || !n2)
    return fail(ctx, CMX_ERR_INVALID, "cmx_candidate_groups: bad arguments");
  if ((s = check_groups(ctx, offsets, nullptr, ngroups, 0)) != CMX_OK) return s;
  CandidateCursor cur;
  cur.G = ngroups; cur.off = offsets; cur.lo = norm_lo; cur.hi = norm_hi; cur.usable = analysable; cur.min_sim = min_sim;
  cur.n2.assign(ngroups, 0);
  cur.waiting.resize((size_t)offsets[ngroups]);
  cur.head.assign((size_t)offsets[ngroups], 0);
  for (size_t g = 0; g < ngroups; ++g) {
    if (analysable[g] && cur.gsize(g) == 0) return fail(ctx, CMX_ERR_INVALID, "cmx_candidate_groups: an analysable group is empty");
    if (analysable[g]) ++cur.n_usable;
  }
  if (cur.n_usable == 0) return fail(ctx, CMX_ERR_INVALID, "cmx_candidate_groups: no analysable group");
  std::fill(n1, n1 + ngroups, 0u);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const HostModel& h = ctx->hm;
  // The device works ahead of the cursor: a round simulates and maps several batches in one launch, the cursor then
  // consumes them batch by batch exactly as the reference would (batches it does not get to are discarded), and the
  // statistics of all pseudo-groups of the round are evaluated in one launch.
  const size_t BK = (size_t)h.B * h.K;
  const size_t round_batches = std::max<size_t>(1, std::min<size_t>(32, 32768 / rep_ram));
  const size_t N = round_batches * rep_ram;
  uint8_t *d_aln, *d_states;
  int32_t *d_cls, *d_pgs;
  int64_t* d_pgo;
  double *d_cnt, *d_norm, *d_st;
  if ((s = scratch(ctx, "cg_aln", (size_t)h.T * N, (void**)&d_aln)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cg_states", (size_t)h.nn * N, (void**)&d_states)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cg_cls", sizeof(int32_t) * N, (void**)&d_cls)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cg_cnt", sizeof(double) * BK * N, (void**)&d_cnt)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cg_norm", sizeof(double) * N, (void**)&d_norm)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cg_pgs", sizeof(int32_t) * N, (void**)&d_pgs)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cg_pgo", sizeof(int64_t) * (N + 1), (void**)&d_pgo)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cg_stat", sizeof(double) * N, (void**)&d_st)) != CMX_OK) return s;
  std::vector<double> norms(N), stats(N);
  std::vector<int64_t> r_off;
  std::vector<int32_t> r_sites, r_group;
  uint64_t nb = 0;
  int more = 1;
  auto go_on = [&]() { return more == 1 && cur.trials < max_trials && (max_batches == 0 || nb < max_batches); };
  while (go_on()) {
    HIP_TRY(ctx, launch_simulate(ctx->dm, seed, nb * (uint64_t)rep_ram, N, d_aln, N, d_cls, d_states, nullptr));
    if ((s = map_sites_impl(ctx, d_aln, N, N, nullptr, d_cnt, N, nullptr, nullptr, nullptr, d_norm, nullptr, true)) != CMX_OK) return s;
    HIP_TRY(ctx, hipMemcpy(norms.data(), d_norm, sizeof(double) * N, hipMemcpyDeviceToHost));
    r_off.assign(1, 0); r_sites.clear(); r_group.clear();
    for (size_t t = 0; t < round_batches && go_on(); ++t) {
      ++nb;
      more = cur.batch(norms.data() + t * rep_ram, rep_ram);
      if (more < 0) return fail(ctx, CMX_ERR_INVALID, "cmx_candidate_groups: candidate cursor found no open group");
      for (size_t q = 0; q < cur.pg_group.size(); ++q) {
        for (int64_t e = cur.pg_off[q]; e < cur.pg_off[q + 1]; ++e) r_sites.push_back((int32_t)(t * rep_ram) + cur.pg_sites[e]);
        r_off.push_back((int64_t)r_sites.size());
        r_group.push_back(cur.pg_group[q]);
      }
    }
    const size_t npg = r_group.size();
    if (npg) {
      HIP_TRY(ctx, hipMemcpy(d_pgo, r_off.data(), sizeof(int64_t) * (npg + 1), hipMemcpyHostToDevice));
      HIP_TRY(ctx, hipMemcpy(d_pgs, r_sites.data(), sizeof(int32_t) * r_sites.size(), hipMemcpyHostToDevice));
      if ((s = cmx_group_stats_dev(ctx, kind, params, d_cnt, N, N, d_pgo, d_pgs, npg, d_st, nullptr)) != CMX_OK) return s;
      HIP_TRY(ctx, hipMemcpy(stats.data(), d_st, sizeof(double) * npg, hipMemcpyDeviceToHost));
      for (size_t q = 0; q < npg; ++q)
        if (stats[q] >= observed[r_group[q]]) ++n1[r_group[q]];
    }
  }
  std::copy(cur.n2.begin(), cur.n2.end(), n2);
  if (trials) *trials = cur.trials;
  if (batches) *batches = nb;
  return CMX_OK;
}

// ------------------------------------------------------------------------------------------------ Mica post-processing
cmx_status cmx_mica_average_mi_dev(cmx_ctx* ctx, const double* d_mi, size_t n, size_t ldo, double* d_average,
                                   double* d_full_average, void* stream) {
  if (!ctx) return CMX_ERR_INVALID;
  if (!d_mi || n < 2 || ldo < n || !d_average || !d_full_average) return fail(ctx, CMX_ERR_INVALID, "cmx_mica_average_mi: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, launch_mica_average(d_mi, n, ldo, d_average, d_full_average, (hipStream_t)stream));
  return CMX_OK;
}

cmx_status cmx_mica_average_mi(cmx_ctx* ctx, const double* mi, size_t n, double* average, double* full_average) {
  if (!ctx) return CMX_ERR_INVALID;
  if (!mi || n < 2 || !average || !full_average) return fail(ctx, CMX_ERR_INVALID, "cmx_mica_average_mi: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  TmpDev tmp;
  double *d_mi, *d_avg, *d_full;
  HIP_TRY(ctx, tmp.alloc((void**)&d_mi, sizeof(double) * n * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_avg, sizeof(double) * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_full, sizeof(double)));
  HIP_TRY(ctx, hipMemcpy(d_mi, mi, sizeof(double) * n * n, hipMemcpyHostToDevice));
  cmx_status s = cmx_mica_average_mi_dev(ctx, d_mi, n, n, d_avg, d_full, nullptr);
  if (s != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(average, d_avg, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(full_average, d_full, sizeof(double), hipMemcpyDeviceToHost));
  return CMX_OK;
}

cmx_status cmx_mica_zscore_null_dev(cmx_ctx* ctx, int which, const double* d_mi, size_t n, size_t ldo, const double* d_average,
                                    const double* d_full_average, const double* d_key, double* d_null_stat,
                                    double* d_null_key, void* stream) {
  if (!ctx) return CMX_ERR_INVALID;
  if (which < CMX_MICA_MI || which > CMX_MICA_MIC) return fail(ctx, CMX_ERR_INVALID, "cmx_mica_zscore_null: unknown statistic");
  if (!d_mi || n < 2 || ldo < n || !d_key || !d_null_stat || !d_null_key || (which != CMX_MICA_MI && (!d_average || !d_full_average)))
    return fail(ctx, CMX_ERR_INVALID, "cmx_mica_zscore_null: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, launch_mica_zscore(which, d_mi, n, ldo, d_average, d_full_average, d_key, d_null_stat, d_null_key,
                                  (hipStream_t)stream));
  return CMX_OK;
}

cmx_status cmx_mica_zscore_null(cmx_ctx* ctx, int which, const double* mi, size_t n, const double* key, double* null_stat,
                                double* null_key) {
  if (!ctx) return CMX_ERR_INVALID;
  if (!mi || n < 2 || !key || !null_stat || !null_key) return fail(ctx, CMX_ERR_INVALID, "cmx_mica_zscore_null: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t np = n * (n - 1) / 2;
  TmpDev tmp;
  double *d_mi, *d_avg, *d_full, *d_key, *d_ns, *d_nk;
  HIP_TRY(ctx, tmp.alloc((void**)&d_mi, sizeof(double) * n * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_avg, sizeof(double) * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_full, sizeof(double)));
  HIP_TRY(ctx, tmp.alloc((void**)&d_key, sizeof(double) * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_ns, sizeof(double) * np));
  HIP_TRY(ctx, tmp.alloc((void**)&d_nk, sizeof(double) * np));
  HIP_TRY(ctx, hipMemcpy(d_mi, mi, sizeof(double) * n * n, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_key, key, sizeof(double) * n, hipMemcpyHostToDevice));
  cmx_status s = cmx_mica_average_mi_dev(ctx, d_mi, n, n, d_avg, d_full, nullptr);
  if (s != CMX_OK) return s;
  if ((s = cmx_mica_zscore_null_dev(ctx, which, d_mi, n, n, d_avg, d_full, d_key, d_ns, d_nk, nullptr)) != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(null_stat, d_ns, sizeof(double) * np, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(null_key, d_nk, sizeof(double) * np, hipMemcpyDeviceToHost));
  return CMX_OK;
}

// extended codes of the permutation test: states 0..A-1; a code in [A, min(nmasks, 31)) with a partial mask keeps its
// number; everything else (codes without an entry, gap, X, N: all states) is 31 = unknown.  L = lcm of the state counts.
namespace {
struct PermCodes {
  uint8_t emap[256];
  uint32_t emask[32], ewgt[32];
  uint32_t L;
  int sh;
};
cmx_status perm_codes(cmx_ctx* ctx, int A, const uint32_t* masks, size_t nmasks, int T, PermCodes* pc) {
  const uint32_t all = (1u << A) - 1u;
  unsigned long long L = (unsigned long long)A;
  auto lcm = [](unsigned long long a, unsigned long long b) {
    unsigned long long x = a, y = b;
    while (y) { const unsigned long long r = x % y; x = y; y = r; }
    return a / x * b;
  };
  int k[32];
  for (int e = 0; e < 32; ++e) { pc->emask[e] = e < A ? (1u << e) : all; k[e] = e < A ? 1 : A; }
  for (int c = 0; c < 256; ++c) {
    if (c < A) { pc->emap[c] = (uint8_t)c; continue; }
    if (!masks || (size_t)c >= nmasks) { pc->emap[c] = 31; continue; }
    const uint32_t m = masks[c] & all;
    if (m == 0) return fail(ctx, CMX_ERR_INVALID, "cmx_mica_permutation_test: mask of code " + std::to_string(c) + " has no state");
    if (m == all) { pc->emap[c] = 31; continue; }
    if (c >= 31) return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_mica_permutation_test: partial ambiguity codes must be < 31");
    pc->emap[c] = (uint8_t)c;
    pc->emask[c] = m;
    k[c] = __builtin_popcount(m);
    L = lcm(L, (unsigned long long)k[c]);
  }
  const double M = (double)L * (double)L * (double)T;
  if (M > 67108864.0)
    return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_mica_permutation_test: lcm of the ambiguity codes' state counts too large for the fixed-point table");
  pc->L = (uint32_t)L;
  for (int e = 0; e < 32; ++e) pc->ewgt[e] = (uint32_t)(L / (unsigned long long)k[e]);
  pc->sh = std::min(40, 62 - (int)std::ceil(std::log2(M * std::log(M))));
  return CMX_OK;
}
}  // namespace

cmx_status cmx_mica_permutation_test_masks_dev(cmx_ctx* ctx, int nalpha, int ntaxa, const uint32_t* masks, size_t nmasks,
                                               const uint8_t* d_aln, size_t n, size_t ld, uint32_t max_perm, uint64_t seed,
                                               size_t pair_begin, size_t pair_end, double* d_pvalue, int32_t* d_nperm, void* stream) {
  if (!ctx) return CMX_ERR_INVALID;
  if (nalpha != 4 && nalpha != 20) return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_mica_permutation_test: alphabet size must be 4 or 20");
  if (ntaxa < 2 || ntaxa > mica_perm_max_taxa())
    return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_mica_permutation_test: 2 <= ntaxa <= " + std::to_string(mica_perm_max_taxa()));
  if (!d_aln || n < 2 || ld < n || max_perm == 0 || pair_end <= pair_begin || pair_end > n * (n - 1) / 2 || !d_pvalue || !d_nperm)
    return fail(ctx, CMX_ERR_INVALID, "cmx_mica_permutation_test: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  PermCodes pc;
  cmx_status s;
  if ((s = perm_codes(ctx, nalpha, masks, nmasks, ntaxa, &pc)) != CMX_OK) return s;
  uint16_t *d_cnt, *d_ext;
  uint8_t *d_emap, *d_hasamb;
  uint32_t* d_tab;   // emask[32] | ewgt[32]
  int* d_bad;
  long long* d_dF;
  if ((s = scratch(ctx, "perm_cnt", sizeof(uint16_t) * n * nalpha, (void**)&d_cnt)) != CMX_OK) return s;
  if ((s = scratch(ctx, "perm_ext", sizeof(uint16_t) * n * 32, (void**)&d_ext)) != CMX_OK) return s;
  if ((s = scratch(ctx, "perm_emap", 256, (void**)&d_emap)) != CMX_OK) return s;
  if ((s = scratch(ctx, "perm_hasamb", n, (void**)&d_hasamb)) != CMX_OK) return s;
  if ((s = scratch(ctx, "perm_tab", sizeof(uint32_t) * 64, (void**)&d_tab)) != CMX_OK) return s;
  if ((s = scratch(ctx, "perm_bad", 2 * sizeof(int), (void**)&d_bad)) != CMX_OK) return s;
  if ((s = scratch(ctx, "perm_dF", sizeof(long long) * ntaxa, (void**)&d_dF)) != CMX_OK) return s;
  // resolved pairs: F[c] = round(c ln c * 2^40); the kernel accumulates F[c+1] - F[c] per increment of a joint count
  // (sources owned by the context: a failing call further down must not free memory an upload still reads)
  std::vector<long long>& dF = ctx->perm_dF_host;
  dF.assign(ntaxa, 0);
  long long prev = 0;
  for (int c = 1; c <= ntaxa; ++c) {
    const long long f = std::llround((double)c * std::log((double)c) * 1099511627776.0);
    dF[c - 1] = f - prev;
    prev = f;
  }
  ctx->perm_tab_host.assign(256 + 64 * sizeof(uint32_t), 0);
  std::memcpy(ctx->perm_tab_host.data(), pc.emap, 256);
  std::memcpy(ctx->perm_tab_host.data() + 256, pc.emask, 32 * sizeof(uint32_t));
  std::memcpy(ctx->perm_tab_host.data() + 256 + 32 * sizeof(uint32_t), pc.ewgt, 32 * sizeof(uint32_t));
  HIP_TRY(ctx, hipMemcpyAsync(d_dF, dF.data(), sizeof(long long) * ntaxa, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d_emap, ctx->perm_tab_host.data(), 256, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d_tab, ctx->perm_tab_host.data() + 256, sizeof(uint32_t) * 64, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemsetAsync(d_bad, 0, 2 * sizeof(int), st));
  HIP_TRY(ctx, launch_mica_colcount(d_aln, ntaxa, n, ld, nalpha, d_emap, d_cnt, d_ext, d_hasamb, d_bad, st));
  int bad[2] = {0, 0};
  HIP_TRY(ctx, hipMemcpyAsync(bad, d_bad, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));   // the one host decision of this call: are there pairs with unknowns at all
  int cus = 0;
  HIP_TRY(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
  const size_t npairs = pair_end - pair_begin;
  bool preset = false;
  if (bad[0]) {
    // pairs with gaps / unknowns / ambiguity codes (SiteTools::*(.., resolveUnknowns = true)): their own kernel, first
    if (mica_perm_general_lds(ntaxa, nalpha, bad[1]) == 0)
      return fail(ctx, CMX_ERR_UNSUPPORTED, "cmx_mica_permutation_test: " + std::to_string(ntaxa) + " taxa with " + std::to_string(bad[1]) +
                                            " distinct ambiguity codes in one column do not fit the LDS");
    const size_t M = (size_t)pc.L * pc.L * (size_t)ntaxa;
    long long* d_F;
    uint16_t* d_order;
    if ((s = scratch(ctx, "perm_F", sizeof(long long) * (M + 1), (void**)&d_F)) != CMX_OK) return s;
    if ((s = scratch(ctx, "perm_order", sizeof(uint16_t) * n * (size_t)ntaxa, (void**)&d_order)) != CMX_OK) return s;
    // F[m] = round(m ln m 2^sh), up to 2^26 entries with a logarithm each: built once per (L, taxa, shift) and kept on the
    // device -- a caller that shards the pairs over several calls pays for it once.  (Built on the host with the oracle's
    // logarithm so that the fixed-point sums, and with them every tie, are the oracle's.)
    if (ctx->perm_F_L != pc.L || ctx->perm_F_T != ntaxa || ctx->perm_F_sh != pc.sh || ctx->perm_F_host.size() != M + 1) {
      ctx->perm_F_sh = -1;
      std::vector<long long>& F = ctx->perm_F_host;
      F.assign(M + 1, 0);
      const double scale = std::ldexp(1.0, pc.sh);
      for (size_t m = 1; m <= M; ++m) F[m] = std::llround((double)m * std::log((double)m) * scale);
      HIP_TRY(ctx, hipMemcpyAsync(d_F, F.data(), sizeof(long long) * (M + 1), hipMemcpyHostToDevice, st));
      ctx->perm_F_L = pc.L; ctx->perm_F_T = ntaxa; ctx->perm_F_sh = pc.sh;
    }
    HIP_TRY(ctx, launch_mica_colorder(d_aln, ntaxa, n, ld, nalpha, d_emap, d_ext, d_order, st));
    if (!mica_perm_opening_fits(ntaxa, nalpha)) {
      HIP_TRY(ctx, hipMemsetAsync(d_nperm, 0xFF, sizeof(int32_t) * npairs, st));   // -1: undecided, no hits (resolved pairs)
      preset = true;
    }
    HIP_TRY(ctx, launch_mica_perm_general(d_aln, ntaxa, n, ld, nalpha, d_emap, d_ext, d_order, d_hasamb, d_tab, d_tab + 32, d_F, pc.L,
                                          bad[1], max_perm, seed, pair_begin, pair_end, d_pvalue, d_nperm, cus, st));
  }
  HIP_TRY(ctx, launch_mica_perm(d_aln, ntaxa, n, ld, nalpha, d_cnt, d_hasamb, d_dF, preset, max_perm, seed, pair_begin, pair_end,
                                d_pvalue, d_nperm, cus, st));
  return CMX_OK;
}

cmx_status cmx_mica_permutation_test_dev(cmx_ctx* ctx, int nalpha, int ntaxa, const uint8_t* d_aln, size_t n, size_t ld,
                                         uint32_t max_perm, uint64_t seed, size_t pair_begin, size_t pair_end,
                                         double* d_pvalue, int32_t* d_nperm, void* stream) {
  return cmx_mica_permutation_test_masks_dev(ctx, nalpha, ntaxa, nullptr, 0, d_aln, n, ld, max_perm, seed, pair_begin, pair_end,
                                             d_pvalue, d_nperm, stream);
}

cmx_status cmx_mica_permutation_test_masks(cmx_ctx* ctx, int nalpha, int ntaxa, const uint32_t* masks, size_t nmasks,
                                           const uint8_t* aln, size_t n, uint32_t max_perm, uint64_t seed, double* pvalue,
                                           int32_t* nperm) {
  if (!ctx) return CMX_ERR_INVALID;
  if (!aln || n < 2 || ntaxa < 2 || !pvalue || !nperm) return fail(ctx, CMX_ERR_INVALID, "cmx_mica_permutation_test: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t np = n * (n - 1) / 2;
  TmpDev tmp;
  uint8_t* d_aln;
  double* d_pv;
  int32_t* d_np;
  HIP_TRY(ctx, tmp.alloc((void**)&d_aln, (size_t)ntaxa * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_pv, sizeof(double) * np));
  HIP_TRY(ctx, tmp.alloc((void**)&d_np, sizeof(int32_t) * np));
  HIP_TRY(ctx, hipMemcpy(d_aln, aln, (size_t)ntaxa * n, hipMemcpyHostToDevice));
  cmx_status s = cmx_mica_permutation_test_masks_dev(ctx, nalpha, ntaxa, masks, nmasks, d_aln, n, n, max_perm, seed, 0, np, d_pv, d_np,
                                                     nullptr);
  if (s != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(pvalue, d_pv, sizeof(double) * np, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(nperm, d_np, sizeof(int32_t) * np, hipMemcpyDeviceToHost));
  return CMX_OK;
}

cmx_status cmx_mica_permutation_test(cmx_ctx* ctx, int nalpha, int ntaxa, const uint8_t* aln, size_t n, uint32_t max_perm,
                                     uint64_t seed, double* pvalue, int32_t* nperm) {
  return cmx_mica_permutation_test_masks(ctx, nalpha, ntaxa, nullptr, 0, aln, n, max_perm, seed, pvalue, nperm);
}

// ------------------------------------------------------------------------------------------------ clustering
static cmx_status check_cluster(cmx_ctx* ctx, int dist_kind, int linkage, size_t n) {
  if (dist_kind < CMX_DIST_CORRELATION || dist_kind > CMX_DIST_EUCLIDIAN) return fail(ctx, CMX_ERR_INVALID, "unknown clustering distance");
  if (linkage < CMX_LINK_COMPLETE || linkage > CMX_LINK_AVERAGE) return fail(ctx, CMX_ERR_INVALID, "unknown clustering method");
  if (n < 2) return fail(ctx, CMX_ERR_INVALID, "clustering needs at least two sites");
  if (n > CMX_CLUSTER_MAX_SITES)
    return fail(ctx, CMX_ERR_UNSUPPORTED, "clustering is limited to " + std::to_string(CMX_CLUSTER_MAX_SITES) + " sites per matrix");
  return CMX_OK;
}

cmx_status cmx_hclust_dev(cmx_ctx* ctx, int linkage, double* d_dist, size_t n, size_t ld, size_t batch, int32_t* d_merge,
                          double* d_dmax, int32_t* d_size, void* stream) {
  if (!ctx) return CMX_ERR_INVALID;
  cmx_status s = check_cluster(ctx, CMX_DIST_CORRELATION, linkage, n);
  if (s != CMX_OK) return s;
  if (!d_dist || ld < n || batch == 0 || !d_merge || !d_dmax || !d_size) return fail(ctx, CMX_ERR_INVALID, "cmx_hclust: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  double* rmin;
  int* nn;
  if ((s = scratch(ctx, "hc_rmin", sizeof(double) * batch * n, (void**)&rmin)) != CMX_OK) return s;
  if ((s = scratch(ctx, "hc_nn", sizeof(int) * batch * n, (void**)&nn)) != CMX_OK) return s;
  HIP_TRY(ctx, launch_hclust(linkage, d_dist, n, ld, n * ld, batch, rmin, nn, d_merge, d_dmax, d_size, (hipStream_t)stream));
  return CMX_OK;
}

cmx_status cmx_hclust(cmx_ctx* ctx, int linkage, const double* dist, size_t n, size_t batch, int32_t* merge, double* dmax,
                      int32_t* size) {
  if (!ctx) return CMX_ERR_INVALID;
  if (!dist || !merge || !dmax || !size || batch == 0) return fail(ctx, CMX_ERR_INVALID, "cmx_hclust: bad arguments");
  cmx_status s = check_cluster(ctx, CMX_DIST_CORRELATION, linkage, n);
  if (s != CMX_OK) return s;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  TmpDev tmp;
  double *d_D, *d_dm;
  int32_t *d_mg, *d_sz;
  const size_t nm = batch * (n - 1);
  HIP_TRY(ctx, tmp.alloc((void**)&d_D, sizeof(double) * batch * n * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_dm, sizeof(double) * nm));
  HIP_TRY(ctx, tmp.alloc((void**)&d_mg, sizeof(int32_t) * 2 * nm));
  HIP_TRY(ctx, tmp.alloc((void**)&d_sz, sizeof(int32_t) * nm));
  HIP_TRY(ctx, hipMemcpy(d_D, dist, sizeof(double) * batch * n * n, hipMemcpyHostToDevice));
  if ((s = cmx_hclust_dev(ctx, linkage, d_D, n, n, batch, d_mg, d_dm, d_sz, nullptr)) != CMX_OK) return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(merge, d_mg, sizeof(int32_t) * 2 * nm, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(dmax, d_dm, sizeof(double) * nm, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(size, d_sz, sizeof(int32_t) * nm, hipMemcpyDeviceToHost));
  return CMX_OK;
}

// `batch` replicates of n sites each in consecutive column blocks of d_counts / d_norm: statistic (upper triangles) ->
// distances -> agglomeration -> group properties.  d_dist_out (batch == 1 only): copy of the distance matrix.
static cmx_status cluster_batch_dev(cmx_ctx* ctx, int dist_kind, int linkage, const double* d_counts, size_t ldc, size_t n,
                                    size_t batch, const double* d_norm, double* d_dist_out, int32_t* d_merge, double* d_dmax,
                                    int32_t* d_size, double* d_stat, double* d_nmin, hipStream_t st) {
  cmx_status s;
  const HostModel& h = ctx->hm;
  double *D, *sigma = nullptr;
  if ((s = scratch(ctx, "cl_D", sizeof(double) * batch * n * n, (void**)&D)) != CMX_OK) return s;
  const int stat_kind = dist_kind == CMX_DIST_CORRELATION ? CMX_STAT_CORRELATION
                        : dist_kind == CMX_DIST_COMPENSATION ? CMX_STAT_COMPENSATION : CMX_STAT_EUCLIDIAN_DISTANCE;
  {  // one operand preparation over all batch * n sites (an operand block per replicate), one Gram launch with a
     // replicate per grid.z slice
    const size_t N = batch * n, ldx = (n + 15) / 16 * 16;
    const int Bp = (h.B + 3) / 4 * 4;
    double *X, *sv, *rv;
    if ((s = scratch(ctx, "pair_X1", sizeof(double) * Bp * ldx * batch, (void**)&X)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_s1", sizeof(double) * N, (void**)&sv)) != CMX_OK) return s;
    if ((s = scratch(ctx, "pair_r1", sizeof(double) * N, (void**)&rv)) != CMX_OK) return s;
    HIP_TRY(ctx, launch_pair_prep(stat_kind, 0.0, d_counts, N, ldc, h.B, h.K, X, ldx, Bp, sv, rv, nullptr, st, n));
    HIP_TRY(ctx, launch_pair_gram(stat_kind, h.B, Bp, X, sv, rv, n, ldx, X, sv, rv, n, ldx, 2 /* upper triangle only */, D, n, st,
                                  batch, n, n * n, (size_t)Bp * ldx));
  }
  HIP_TRY(ctx, launch_dist_finish(dist_kind, D, n, n, n * n, batch, st));
  if (d_dist_out) HIP_TRY(ctx, hipMemcpyAsync(d_dist_out, D, sizeof(double) * n * n, hipMemcpyDeviceToDevice, st));
  if ((s = cmx_hclust_dev(ctx, linkage, D, n, n, batch, d_merge, d_dmax, d_size, st)) != CMX_OK) return s;
  if (dist_kind == CMX_DIST_COMPENSATION &&
      (s = scratch(ctx, "cl_sigma", sizeof(double) * batch * (2 * n - 1) * h.B, (void**)&sigma)) != CMX_OK)
    return s;
  HIP_TRY(ctx, launch_cluster_props(dist_kind, (int)n, h.B, h.K, batch, d_merge, d_dmax, d_norm, d_counts, ldc, n, sigma, d_stat,
                                    d_nmin, st));
  return CMX_OK;
}

cmx_status cmx_cluster_sites_dev(cmx_ctx* ctx, int dist_kind, int linkage, const double* d_counts, size_t n, size_t ldc,
                                 const double* d_norm, double* d_dist_out, int32_t* d_merge, double* d_dmax, int32_t* d_size,
                                 double* d_stat, double* d_nmin, void* stream) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if ((s = check_cluster(ctx, dist_kind, linkage, n)) != CMX_OK) return s;
  if (!d_counts || ldc < n || !d_norm || !d_merge || !d_dmax || !d_size || !d_stat || !d_nmin)
    return fail(ctx, CMX_ERR_INVALID, "cmx_cluster_sites: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return cluster_batch_dev(ctx, dist_kind, linkage, d_counts, ldc, n, 1, d_norm, d_dist_out, d_merge, d_dmax, d_size, d_stat,
                           d_nmin, (hipStream_t)stream);
}

cmx_status cmx_cluster_sites(cmx_ctx* ctx, int dist_kind, int linkage, const double* counts, size_t n, double* dist_out,
                             int32_t* merge, double* dmax, int32_t* size, double* stat, double* nmin) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if ((s = check_cluster(ctx, dist_kind, linkage, n)) != CMX_OK) return s;
  if (!counts || !merge || !dmax || !size || !stat || !nmin) return fail(ctx, CMX_ERR_INVALID, "cmx_cluster_sites: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const HostModel& h = ctx->hm;
  const size_t BK = (size_t)h.B * h.K, nm = n - 1;
  std::vector<double> bm, norm(n);
  to_branch_major(counts, n, BK, &bm);
  for (size_t i = 0; i < n; ++i) {       // computeNormForSite: sqrt(sum_b (sum_k n_bk)^2)
    double q = 0.0;
    for (int b = 0; b < h.B; ++b) {
      double t = 0.0;
      for (int k = 0; k < h.K; ++k) t += counts[i * BK + (size_t)b * h.K + k];
      q += t * t;
    }
    norm[i] = std::sqrt(q);
  }
  TmpDev tmp;
  double *d_c, *d_norm, *d_dist = nullptr, *d_dm, *d_st, *d_nmn;
  int32_t *d_mg, *d_sz;
  HIP_TRY(ctx, tmp.alloc((void**)&d_c, sizeof(double) * bm.size()));
  HIP_TRY(ctx, tmp.alloc((void**)&d_norm, sizeof(double) * n));
  if (dist_out) HIP_TRY(ctx, tmp.alloc((void**)&d_dist, sizeof(double) * n * n));
  HIP_TRY(ctx, tmp.alloc((void**)&d_dm, sizeof(double) * nm));
  HIP_TRY(ctx, tmp.alloc((void**)&d_st, sizeof(double) * nm));
  HIP_TRY(ctx, tmp.alloc((void**)&d_nmn, sizeof(double) * nm));
  HIP_TRY(ctx, tmp.alloc((void**)&d_mg, sizeof(int32_t) * 2 * nm));
  HIP_TRY(ctx, tmp.alloc((void**)&d_sz, sizeof(int32_t) * nm));
  HIP_TRY(ctx, hipMemcpy(d_c, bm.data(), sizeof(double) * bm.size(), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_norm, norm.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  if ((s = cmx_cluster_sites_dev(ctx, dist_kind, linkage, d_c, n, n, d_norm, d_dist, d_mg, d_dm, d_sz, d_st, d_nmn, nullptr)) != CMX_OK)
    return s;
  HIP_TRY(ctx, hipDeviceSynchronize());
  if (dist_out) HIP_TRY(ctx, hipMemcpy(dist_out, d_dist, sizeof(double) * n * n, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(merge, d_mg, sizeof(int32_t) * 2 * nm, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(dmax, d_dm, sizeof(double) * nm, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(size, d_sz, sizeof(int32_t) * nm, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(stat, d_st, sizeof(double) * nm, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(nmin, d_nmn, sizeof(double) * nm, hipMemcpyDeviceToHost));
  return CMX_OK;
}

cmx_status cmx_cluster_null(cmx_ctx* ctx, int dist_kind, int linkage, uint64_t seed, size_t rep_begin, size_t rep_end,
                            size_t nsites, int32_t* merge, double* dmax, int32_t* size, double* stat, double* nmin) {
  cmx_status s = need_model(ctx);
  if (s != CMX_OK) return s;
  if ((s = check_cluster(ctx, dist_kind, linkage, nsites)) != CMX_OK) return s;
  if (rep_end <= rep_begin || !merge || !dmax || !size || !stat || !nmin) return fail(ctx, CMX_ERR_INVALID, "cmx_cluster_null: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const HostModel& h = ctx->hm;
  const size_t n = nsites, nm = n - 1, nrep = rep_end - rep_begin, BK = (size_t)h.B * h.K;
  // replicates per batch: every replicate keeps its own n x n matrix in HBM; 16 GiB of them at most
  const size_t per_rep = sizeof(double) * (n * n + BK * n) + (size_t)(h.T + h.nn) * n;
  const size_t R = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(nrep, 1024), ((size_t)16 << 30) / per_rep));
  const size_t N = R * n;
  uint8_t *d_aln, *d_states;
  int32_t *d_cls, *d_mg, *d_sz;
  double *d_cnt, *d_norm, *d_dm, *d_st, *d_nmn;
  if ((s = scratch(ctx, "cl_aln", (size_t)h.T * N, (void**)&d_aln)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cl_states", (size_t)h.nn * N, (void**)&d_states)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cl_cls", sizeof(int32_t) * N, (void**)&d_cls)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cl_cnt", sizeof(double) * BK * N, (void**)&d_cnt)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cl_norm", sizeof(double) * N, (void**)&d_norm)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cl_merge", sizeof(int32_t) * 2 * R * nm, (void**)&d_mg)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cl_size", sizeof(int32_t) * R * nm, (void**)&d_sz)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cl_dmax", sizeof(double) * R * nm, (void**)&d_dm)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cl_stat", sizeof(double) * R * nm, (void**)&d_st)) != CMX_OK) return s;
  if ((s = scratch(ctx, "cl_nmin", sizeof(double) * R * nm, (void**)&d_nmn)) != CMX_OK) return s;
  for (size_t r0 = 0; r0 < nrep; r0 += R) {
    const size_t rb = std::min(R, nrep - r0), nb = rb * n;
    HIP_TRY(ctx, launch_simulate(ctx->dm, seed, (uint64_t)(rep_begin + r0) * n, nb, d_aln, nb, d_cls, d_states, nullptr));
    if ((s = map_sites_impl(ctx, d_aln, nb, nb, nullptr, d_cnt, nb, nullptr, nullptr, nullptr, d_norm, nullptr, true)) != CMX_OK) return s;
    if ((s = cluster_batch_dev(ctx, dist_kind, linkage, d_cnt, nb, n, rb, d_norm, nullptr, d_mg, d_dm, d_sz, d_st, d_nmn, nullptr)) != CMX_OK)
      return s;
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(merge + 2 * r0 * nm, d_mg, sizeof(int32_t) * 2 * rb * nm, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(size + r0 * nm, d_sz, sizeof(int32_t) * rb * nm, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(dmax + r0 * nm, d_dm, sizeof(double) * rb * nm, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(stat + r0 * nm, d_st, sizeof(double) * rb * nm, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(nmin + r0 * nm, d_nmn, sizeof(double) * rb * nm, hipMemcpyDeviceToHost));
  }
  return CMX_OK;
}

}  // extern "C"
