// comap_mi355x_multigpu.hpp -- one process driving the N MI355X of one node through the C-ABI (comap_mi355x.h) and ONE
// RCCL collective: SURVEY.md 8(b) "multi-GPU driver owns 8 ctxs + one RCCL communicator", 8(e).
//
// What shards (same arithmetic as comap_amd/distributed.py, which the one-process-per-GPU Python path uses):
//   * the parametric-bootstrap null (AnalysisTools::getNullDistributionIntraDR, CoMap/AnalysisTools.cpp:564-658): replicates
//     are independent; rank r maps the contiguous range replicateShard(r, N, repCPU); the counter RNG is keyed by the
//     GLOBAL simulated-site index, so the merged null is bit-identical for any N.  ONE all-gather gives every rank the
//     whole null in the reference's replicate order;
//   * the observed pair loop (CoETools::computeIntraStats, CoMap/CoETools.cpp:672-724): rows of the upper triangle split
//     by PAIR count (rowShard); every rank maps the observed alignment itself (it is small) and compacts the rows of
//     its range; ranges are contiguous in i, so the ranks' rows concatenate to the single-GPU output in the
//     reference's (i, j) order -- no second collective on the data path.
// xGMI is point-to-point: the exchange is 16 bytes per null pair (20 MB per device at the north-star target with 8
// devices), far below any link limit, so it is a single all-gather and not a ring of smaller ones.
//
// Round 4: a throughput path, not only a mirror.
//   * The exchange is a policy.  RcclExchange (default; `MultiGpu`) = one grouped ncclAllGather over the N communicators
//     of this process, one rank per distinct device.  LoopbackExchange (`LoopbackMultiGpu`) = the SAME all-gather written
//     as device-to-device copies between N contexts that may all live on ONE device: every line of the N > 1 logic
//     (uneven shards, NaN padding, reassembly into replicate order, row ranges) runs on a one-GPU box and is tested there
//     byte for byte against the single-context path (tests/test_multigpu_cpp.py).
//   * Every buffer lives in a grow-only per-rank arena owned by the object (nothing is allocated or freed per call
//     once the sizes have been seen), uploads are asynchronous from pinned staging, and nothing blocks the host between
//     the first launch on rank 0 and the last launch on rank N - 1.
//   * Rows stay on the devices (enqueueIntraStats + deviceRows) or come home with ONE asynchronous pinned copy per rank
//     (fetchRows: spans of cmx_pair_row in rank order, no per-row conversion).  computeIntraStats keeps the
//     reference-shaped std::vector<IntraStatRow> for callers that want it (one resize + one bulk conversion).
//
// Needs the HIP runtime API and RCCL headers (/opt/rocm/include); link with -lcomap_mi355x -lrccl -lamdhip64.
#ifndef COMAP_MI355X_MULTIGPU_HPP
#define COMAP_MI355X_MULTIGPU_HPP

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstring>
#include <memory>
#include <thread>
#include <utility>

#include "comap_mi355x_adapter.hpp"

namespace cmx {

// contiguous, balanced split of [0, nrep): the first nrep % world ranks get one extra replicate
inline std::pair<size_t, size_t> replicateShard(size_t rank, size_t world, size_t nrep) {
  const size_t q = nrep / world, r = nrep % world;
  const size_t begin = rank * q + std::min(rank, r);
  return {begin, begin + q + (rank < r ? 1 : 0)};
}
// rows [begin, end) of the upper triangle for this rank, balanced by pair count (row i holds n - 1 - i pairs)
inline std::pair<size_t, size_t> rowShard(size_t rank, size_t world, size_t n) {
  const unsigned long long total = (unsigned long long)n * (n - 1) / 2;
  auto firstRowWithPrefixAtLeast = [n](unsigned long long target) {
    size_t lo = 0, hi = n;   // smallest r with pairs(rows < r) >= target; pairs(rows < r) = r (n - 1) - r (r - 1) / 2
    while (lo < hi) {
      const size_t mid = (lo + hi) / 2;
      const unsigned long long m = mid;
      if (m * (n - 1) - m * (m - 1) / 2 >= target) hi = mid; else lo = mid + 1;
    }
    return lo;
  };
  const size_t begin = rank ? firstRowWithPrefixAtLeast(total * rank / world) : 0;
  const size_t end = rank + 1 < world ? firstRowWithPrefixAtLeast(total * (rank + 1) / world) : n;
  return {begin, end};
}
// pairs (i, j), j > i, of the rows [begin, end)
inline size_t pairsOfRows(size_t begin, size_t end, size_t n) {
  return (end - begin) * (n - 1) - (end * (end - 1) - begin * (begin - 1)) / 2;
}

namespace detail {
inline void hipCheck(hipError_t e, const char* what = "MultiGpu") {
  if (e != hipSuccess) throw Exception(std::string(what) + ": HIP error: " + hipGetErrorString(e));
}
inline void ncclCheck(ncclResult_t e) {
  if (e != ncclSuccess) throw Exception(std::string("MultiGpu: RCCL error: ") + ncclGetErrorString(e));
}
}  // namespace detail

// ---- exchange policies: allGather(send[r], recv[r], count) leaves recv[r] = send[0] | send[1] | ... | send[N-1] on every
// rank, ordered after what stream r held before the call and before what it is given afterwards.
struct RcclExchange {   // one communicator per rank, all in this process; ranks must sit on DISTINCT devices
  std::vector<ncclComm_t> comms;
  void init(const std::vector<int>& devices, const std::vector<hipStream_t>&) {
    for (size_t a = 0; a < devices.size(); ++a)
      for (size_t b = a + 1; b < devices.size(); ++b)
        if (devices[a] == devices[b]) throw Exception("MultiGpu: RCCL needs one distinct device per rank (LoopbackMultiGpu runs N ranks on one device).");
    comms.assign(devices.size(), nullptr);
    detail::ncclCheck(ncclCommInitAll(comms.data(), (int)devices.size(), devices.data()));
  }
  void allGather(const std::vector<double*>& send, const std::vector<double*>& recv, size_t count, const std::vector<int>&,
                 const std::vector<hipStream_t>& streams) {
    detail::ncclCheck(ncclGroupStart());
    for (size_t r = 0; r < comms.size(); ++r) detail::ncclCheck(ncclAllGather(send[r], recv[r], count, ncclDouble, comms[r], streams[r]));
    detail::ncclCheck(ncclGroupEnd());
  }
  void destroy(const std::vector<int>& devices) {
    for (size_t r = 0; r < comms.size(); ++r)
      if (comms[r]) { (void)hipSetDevice(devices[r]); (void)ncclCommDestroy(comms[r]); comms[r] = nullptr; }
  }
};

struct LoopbackExchange {   // the same all-gather as N x N device-to-device copies; ranks may share a device
  std::vector<hipEvent_t> ready;
  void init(const std::vector<int>& devices, const std::vector<hipStream_t>&) {
    ready.assign(devices.size(), nullptr);
    for (size_t r = 0; r < devices.size(); ++r) {
      detail::hipCheck(hipSetDevice(devices[r]));
      detail::hipCheck(hipEventCreateWithFlags(&ready[r], hipEventDisableTiming));
    }
  }
  void allGather(const std::vector<double*>& send, const std::vector<double*>& recv, size_t count, const std::vector<int>& devices,
                 const std::vector<hipStream_t>& streams) {
    const size_t N = ready.size();
    for (size_t q = 0; q < N; ++q) {   // rank q's shard is complete when its stream reaches this point
      detail::hipCheck(hipSetDevice(devices[q]));
      detail::hipCheck(hipEventRecord(ready[q], streams[q]));
    }
    for (size_t r = 0; r < N; ++r) {
      detail::hipCheck(hipSetDevice(devices[r]));
      for (size_t q = 0; q < N; ++q) {
        if (q != r) detail::hipCheck(hipStreamWaitEvent(streams[r], ready[q], 0));
        detail::hipCheck(hipMemcpyAsync(recv[r] + q * count, send[q], sizeof(double) * count, hipMemcpyDeviceToDevice, streams[r]));
      }
    }
  }
  void destroy(const std::vector<int>& devices) {
    for (size_t r = 0; r < ready.size(); ++r)
      if (ready[r]) { (void)hipSetDevice(devices[r]); (void)hipEventDestroy(ready[r]); ready[r] = nullptr; }
  }
};

template <class Exchange>
class BasicMultiGpu {
 public:
  // one context and one stream per rank; rank r runs on devices[r]
  BasicMultiGpu(const TreeArrays& tree, const ModelArrays& model, const std::vector<int>& devices) : devices_(devices) {
    if (devices.empty()) throw Exception("MultiGpu: no device given.");
    for (int d : devices) engines_.emplace_back(new Engine(tree, model, d));
    streams_.assign(devices.size(), nullptr);
    side_.assign(devices.size(), nullptr);
    sideDone_.assign(devices.size(), nullptr);
    dv_.resize(devices.size());
    for (size_t r = 0; r < devices.size(); ++r) {
      hip(hipSetDevice(devices[r]));
      hip(hipStreamCreate(&streams_[r]));
      hip(hipStreamCreate(&side_[r]));   // the observed alignment's mapping (and Gram) beside the null: independent work
      hip(hipEventCreateWithFlags(&sideDone_[r], hipEventDisableTiming));
    }
    exchange_.init(devices_, streams_);
  }
  ~BasicMultiGpu() {
    for (size_t r = 0; r < devices_.size(); ++r) {
      (void)hipSetDevice(devices_[r]);
      if (streams_[r]) (void)hipStreamSynchronize(streams_[r]);
      if (side_[r]) (void)hipStreamSynchronize(side_[r]);
      for (Buf* b : dv_[r].all()) if (b->p) (void)hipFree(b->p);
      if (dv_[r].hostRows.p) (void)hipHostFree(dv_[r].hostRows.p);
      if (dv_[r].hostSite.p) (void)hipHostFree(dv_[r].hostSite.p);
      if (dv_[r].hostCount) (void)hipHostFree(dv_[r].hostCount);
    }
    exchange_.destroy(devices_);
    for (size_t r = 0; r < devices_.size(); ++r) {
      (void)hipSetDevice(devices_[r]);
      if (streams_[r]) (void)hipStreamDestroy(streams_[r]);
      if (side_[r]) (void)hipStreamDestroy(side_[r]);
      if (sideDone_[r]) (void)hipEventDestroy(sideDone_[r]);
    }
    if (hostAln_.p) (void)hipHostFree(hostAln_.p);
    if (hostMasks_) (void)hipHostFree(hostMasks_);
  }
  BasicMultiGpu(const BasicMultiGpu&) = delete;
  BasicMultiGpu& operator=(const BasicMultiGpu&) = delete;
  size_t size() const { return devices_.size(); }
  // Rows wanted on the HOST and no pair filter set: let the devices write the 16-byte records of
  // cmx_intra_compact_range_dev (a third of the bytes over PCIe, no counting pass) and rebuild the 48-byte rows on the host
  // (cmx_expand_compact_rows: bit for bit the rows the devices would have written).  deviceRows() has no rows to show in
  // that mode and throws.  computeIntraStats() switches it on by itself; off by default for enqueueIntraStats().
  void enableCompactTransfer(bool on = true) { compactWanted_ = on; }
  const Engine& engine(size_t r) const { return *engines_[r]; }
  hipStream_t stream(size_t r) const { return streams_[r]; }

  // ---- step 1: everything onto the devices' streams, nothing waited for.  CoETools::getVectors + computeIntraStats with
  // their null (CoMap.cpp:155, 363) over all ranks.  aln: [taxon][site] codes (host).  After this call (and a
  // synchronize / fetchRows) rank r holds the compacted rows of its range, and every rank the merged null.
  void enqueueIntraStats(const uint8_t* aln, size_t nbSites, const uint32_t* masks, size_t nbMasks, const Statistic& statistic,
                         bool computeNull, uint64_t seed, size_t nbRepCPU = 100, size_t nbRepRAM = 1000, size_t nbRateClasses = 10,
                         const PairFilters& f = PairFilters()) {
    const size_t N = size(), n = nbSites, T = engines_[0]->getNumberOfTaxa();
    const size_t BK = engines_[0]->getNumberOfBranches() * engines_[0]->getNumberOfSubstitutionTypes();
    if (n < 2) throw Exception("MultiGpu::computeIntraStats: at least two sites are needed.");
    nnull_ = computeNull ? nbRepCPU * nbRepRAM : 0;
    size_t mx = 0;   // entries of the largest null shard: shards are padded to it for the collective
    for (size_t r = 0; r < N; ++r) {
      const auto s = replicateShard(r, N, nbRepCPU);
      mx = std::max(mx, (s.second - s.first) * nbRepRAM);
    }
    cmx_pair_filters pf;
    pf.min_rate_class = f.minRateClass; pf.max_rate_class_diff = f.maxRateClassDiff;
    pf.min_rate = f.minRate; pf.max_rate_diff = f.maxRateDiff; pf.min_statistic = f.minStatistic;
    // pinned staging of the inputs: ONE host copy, then an asynchronous upload per rank (a pageable source would make
    // every hipMemcpyAsync a blocking staged copy on the thread that is supposed to keep N devices fed)
    if (hostAln_.bytes < T * n) {
      if (hostAln_.p) hip(hipHostFree(hostAln_.p));
      hostAln_.p = nullptr;
      hip(hipHostMalloc(&hostAln_.p, T * n, hipHostMallocDefault));
      hostAln_.bytes = T * n;
    }
    for (size_t r = 0; r < N; ++r) {   // the staging buffers may still feed the previous call's uploads
      hip(hipSetDevice(devices_[r]));
      hip(hipStreamSynchronize(streams_[r]));
      hip(hipStreamSynchronize(side_[r]));
    }
    std::memcpy(hostAln_.p, aln, T * n);
    if (masks) {
      if (!hostMasks_) hip(hipHostMalloc((void**)&hostMasks_, 256 * sizeof(uint32_t), hipHostMallocDefault));
      const int S = engines_[0]->getNumberOfStates();
      for (size_t i = 0; i < 256; ++i) hostMasks_[i] = i < nbMasks ? masks[i] : (S >= 32 ? 0xffffffffu : ((1u << S) - 1u));
    }
    // ---- every rank: observed alignment up, mapped; its null shard into the send buffer [2][mx] (NaN padded)
    for (size_t r = 0; r < N; ++r) {
      Dev& d = dv_[r];
      const Engine& e = *engines_[r];
      hip(hipSetDevice(devices_[r]));
      hipStream_t st = streams_[r];
      const auto rs = rowShard(r, N, n);
      d.rowBegin = rs.first; d.rowEnd = rs.second;
      d.cap = pairsOfRows(d.rowBegin, d.rowEnd, n);
      ensure(d.aln, T * n);
      ensure(d.counts, sizeof(double) * BK * n);
      ensure(d.pr, sizeof(double) * n);
      ensure(d.norm, sizeof(double) * n);
      ensure(d.rc, sizeof(int32_t) * n);
      // the unfiltered loop as 16-byte records when the rows are wanted on the host (enableCompactTransfer)
      d.compact = compactWanted_ && f.minRateClass <= 0 && f.maxRateClassDiff < 0 && !(f.minRate > 0.) && f.maxRateDiff < 0. &&
                  !(f.minStatistic > 0.);
      ensure(d.rows, (d.compact ? sizeof(cmx_pair_compact) : sizeof(cmx_pair_row)) * std::max<size_t>(d.cap, 1));
      ensure(d.count, sizeof(uint64_t));
      if (!d.hostCount) hip(hipHostMalloc((void**)&d.hostCount, sizeof(uint64_t), hipHostMallocDefault));
      // the observed alignment on the side stream: up, mapped, (records mode) its pairs' Gram blocks kept for the record
      // pass and the per-site arrays the host expansion needs on their way home -- all beside the null on the main stream
      hipStream_t sd = side_[r];
      hip(hipMemcpyAsync(d.aln.p, hostAln_.p, T * n, hipMemcpyHostToDevice, sd));
      if (masks) {
        ensure(d.masks, 256 * sizeof(uint32_t));
        hip(hipMemcpyAsync(d.masks.p, hostMasks_, 256 * sizeof(uint32_t), hipMemcpyHostToDevice, sd));
      }
      e.check(cmx_map_sites_dev(e.ctx(), d.aln.template as<uint8_t>(), n, n, masks ? d.masks.template as<uint32_t>() : nullptr,
                                d.counts.template as<double>(), n, nullptr, d.pr.template as<double>(), d.rc.template as<int32_t>(),
                                d.norm.template as<double>(), sd));
      if (d.compact) {
        e.check(cmx_intra_gram_prefetch_dev(e.ctx(), statistic.kind(), d.counts.template as<double>(), n, n, d.rowBegin, d.rowEnd, sd));
        const size_t sb = n * (sizeof(int32_t) + 2 * sizeof(double));
        if (d.hostSite.bytes < sb) {
          if (d.hostSite.p) hip(hipHostFree(d.hostSite.p));
          d.hostSite.p = nullptr;
          hip(hipHostMalloc(&d.hostSite.p, sb, hipHostMallocDefault));
          d.hostSite.bytes = sb;
        }
        char* hs = static_cast<char*>(d.hostSite.p);   // [n] posterior rate | [n] norm | [n] rate class
        hip(hipMemcpyAsync(hs, d.pr.p, sizeof(double) * n, hipMemcpyDeviceToHost, sd));
        hip(hipMemcpyAsync(hs + sizeof(double) * n, d.norm.p, sizeof(double) * n, hipMemcpyDeviceToHost, sd));
        hip(hipMemcpyAsync(hs + 2 * sizeof(double) * n, d.rc.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, sd));
      }
      hip(hipEventRecord(sideDone_[r], sd));
      if (computeNull) {
        const auto s = replicateShard(r, N, nbRepCPU);
        ensure(d.send, sizeof(double) * 2 * mx);
        ensure(d.recv, sizeof(double) * 2 * mx * N);
        ensure(d.nstat, sizeof(double) * nnull_);
        ensure(d.nnmin, sizeof(double) * nnull_);
        hip(hipMemsetAsync(d.send.p, 0xFF, sizeof(double) * 2 * mx, st));   // all-ones bytes are a NaN
        if (s.second > s.first)
          e.check(cmx_null_intra_dev(e.ctx(), statistic.kind(), statistic.params(), seed, s.first, s.second, nbRepRAM, nullptr,
                                     d.send.template as<double>(), nullptr, nullptr, d.send.template as<double>() + mx, st));
      }
    }
    // ---- the path's one exchange
    if (computeNull) {
      std::vector<double*> send(N), recv(N);
      for (size_t r = 0; r < N; ++r) { send[r] = dv_[r].send.template as<double>(); recv[r] = dv_[r].recv.template as<double>(); }
      exchange_.allGather(send, recv, 2 * mx, devices_, streams_);
    }
    // ---- every rank: the shards back into replicate order, then the rows of its range against the merged null
    for (size_t r = 0; r < N; ++r) {
      Dev& d = dv_[r];
      const Engine& e = *engines_[r];
      hip(hipSetDevice(devices_[r]));
      hipStream_t st = streams_[r];
      hip(hipStreamWaitEvent(st, sideDone_[r], 0));   // the observed alignment's vectors (and kept Gram blocks) are there
      if (computeNull) reassembleNull(d.recv.template as<double>(), d.nstat.template as<double>(), d.nnmin.template as<double>(), N, nbRepCPU, nbRepRAM, mx, st);
      if (d.compact)
        e.check(cmx_intra_compact_range_dev(e.ctx(), statistic.kind(), statistic.params(), d.counts.template as<double>(), n, n,
                                            d.norm.template as<double>(), computeNull ? d.nstat.template as<double>() : nullptr,
                                            computeNull ? d.nnmin.template as<double>() : nullptr, nnull_, (int)nbRateClasses, d.rowBegin,
                                            d.rowEnd, d.rows.template as<cmx_pair_compact>(), d.cap, st));
      else {
        e.check(cmx_intra_rows_range_dev(e.ctx(), statistic.kind(), statistic.params(), d.counts.template as<double>(), n, n,
                                         d.rc.template as<int32_t>(), d.pr.template as<double>(), d.norm.template as<double>(),
                                         computeNull ? d.nstat.template as<double>() : nullptr, computeNull ? d.nnmin.template as<double>() : nullptr,
                                         nnull_, (int)nbRateClasses, &pf, d.rowBegin, d.rowEnd, d.rows.template as<cmx_pair_row>(), d.cap,
                                         d.count.template as<uint64_t>(), st));
        hip(hipMemcpyAsync(d.hostCount, d.count.p, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      }
      d.nSites = n;
      d.fetched = false;
    }
  }

  // the gathered shards [rank][2][mx] (stat | nmin, NaN padded) -> the null in replicate order.  Static and public: it is
  // the one piece of N > 1 arithmetic besides the shard functions, and the tests call it with made-up layouts too.
  static void reassembleNull(const double* d_recv, double* d_nstat, double* d_nnmin, size_t N, size_t nbRepCPU, size_t nbRepRAM, size_t mx,
                             hipStream_t st) {
    for (size_t q = 0; q < N; ++q) {
      const auto s = replicateShard(q, N, nbRepCPU);
      const size_t cnt = (s.second - s.first) * nbRepRAM, off = s.first * nbRepRAM;
      if (!cnt) continue;
      hip(hipMemcpyAsync(d_nstat + off, d_recv + q * 2 * mx, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
      hip(hipMemcpyAsync(d_nnmin + off, d_recv + q * 2 * mx + mx, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
    }
  }

  void synchronize() {
    for (size_t r = 0; r < size(); ++r) {
      hip(hipSetDevice(devices_[r]));
      hip(hipStreamSynchronize(side_[r]));
      hip(hipStreamSynchronize(streams_[r]));
    }
  }

  // ---- step 2a: rows left on the device (valid after synchronize(); the caller reads them with its own kernels / copies)
  struct DeviceRows { const cmx_pair_row* rows; size_t count, rowBegin, rowEnd; };
  DeviceRows deviceRows(size_t r) {
    hip(hipSetDevice(devices_[r]));
    hip(hipStreamSynchronize(streams_[r]));
    const Dev& d = dv_[r];
    if (d.compact) throw Exception("MultiGpu::deviceRows: the devices hold 16-byte records (enableCompactTransfer), not rows.");
    return {d.rows.template as<cmx_pair_row>(), (size_t)std::min<uint64_t>(*d.hostCount, d.cap), d.rowBegin, d.rowEnd};
  }
  const double* deviceNullStat(size_t r) const { return dv_[r].nstat.template as<double>(); }   // merged null, replicate order, [nnull]
  const double* deviceNullNmin(size_t r) const { return dv_[r].nnmin.template as<double>(); }
  size_t nullSize() const { return nnull_; }

  // ---- step 2b: rows home.  One asynchronous copy per rank into its own grow-only pinned buffer; a rank's copy is issued
  // as soon as ITS row count is known, so the N copies overlap on the N PCIe links.  Spans are in rank order == the
  // reference's (i, j) order; they stay valid until the next enqueueIntraStats.
  struct HostRows {
    std::vector<const cmx_pair_row*> rows;
    std::vector<size_t> count;
    size_t total() const { size_t t = 0; for (size_t c : count) t += c; return t; }
  };
  const HostRows& fetchRows() {
    const size_t N = size();
    host_.rows.assign(N, nullptr);
    host_.count.assign(N, 0);
    for (size_t r = 0; r < N; ++r) {
      Dev& d = dv_[r];
      hip(hipSetDevice(devices_[r]));
      hip(hipStreamSynchronize(streams_[r]));   // ranks run concurrently: while this one is waited for the others proceed
      const size_t cnt = d.compact ? d.cap : (size_t)std::min<uint64_t>(*d.hostCount, d.cap);   // (records mode: every pair)
      const size_t each = d.compact ? sizeof(cmx_pair_compact) : sizeof(cmx_pair_row);
      if (d.hostRows.bytes < each * cnt) {
        if (d.hostRows.p) hip(hipHostFree(d.hostRows.p));
        d.hostRows.p = nullptr;
        hip(hipHostMalloc(&d.hostRows.p, each * cnt, hipHostMallocDefault));
        d.hostRows.bytes = each * cnt;
      }
      if (cnt && !d.fetched) hip(hipMemcpyAsync(d.hostRows.p, d.rows.p, each * cnt, hipMemcpyDeviceToHost, streams_[r]));
      host_.rows[r] = static_cast<const cmx_pair_row*>(d.hostRows.p);
      host_.count[r] = cnt;
    }
    synchronize();
    // records mode: the 48-byte rows rebuilt from the records and the per-site arrays (host threads, no GPU)
    for (size_t r = 0; r < N; ++r) {
      Dev& d = dv_[r];
      if (d.compact && !d.fetched && host_.count[r]) {
        if (d.expandedCap < host_.count[r]) {
          d.expanded.reset(new cmx_pair_row[host_.count[r]]);
          d.expandedCap = host_.count[r];
        }
        const char* hs = static_cast<const char*>(d.hostSite.p);
        const size_t n = d.nSites;
        unsigned nt = std::thread::hardware_concurrency();
        nt = nt == 0 ? 1u : (nt > 16 ? 16u : nt);
        const cmx_status st = cmx_expand_compact_rows(n, d.rowBegin, d.rowEnd, reinterpret_cast<const int32_t*>(hs + 2 * sizeof(double) * n),
                                                      reinterpret_cast<const double*>(hs), reinterpret_cast<const double*>(hs + sizeof(double) * n),
                                                      static_cast<const cmx_pair_compact*>(d.hostRows.p), host_.count[r], d.expanded.get(), (int)nt);
        if (st != CMX_OK) throw Exception("MultiGpu::fetchRows: cmx_expand_compact_rows failed.");
      }
      if (d.compact) host_.rows[r] = d.expanded.get();
      d.fetched = true;
    }
    return host_;
  }

  // ---- the reference-shaped call: rows of statistics.txt in the reference's order; nullRows (optional) receives the
  // merged null (Stat, Nmin; RCmin / PRmin are not exchanged: 0 / NaN) in replicate order.
  std::vector<IntraStatRow> computeIntraStats(const uint8_t* aln, size_t nbSites, const uint32_t* masks, size_t nbMasks,
                                              const Statistic& statistic, bool computeNull, uint64_t seed, size_t nbRepCPU = 100,
                                              size_t nbRepRAM = 1000, size_t nbRateClasses = 10, const PairFilters& f = PairFilters(),
                                              std::vector<NullDistributionRow>* nullRows = nullptr) {
    const bool was = compactWanted_;
    compactWanted_ = true;   // only host rows are asked for here
    try {
      enqueueIntraStats(aln, nbSites, masks, nbMasks, statistic, computeNull, seed, nbRepCPU, nbRepRAM, nbRateClasses, f);
    } catch (...) {
      compactWanted_ = was;
      throw;
    }
    compactWanted_ = was;
    const HostRows& h = fetchRows();
    std::vector<IntraStatRow> rows(h.total());
    size_t k = 0;
    for (size_t r = 0; r < h.rows.size(); ++r)
      for (size_t q = 0; q < h.count[r]; ++q, ++k) {
        const cmx_pair_row& s = h.rows[r][q];
        IntraStatRow& o = rows[k];
        o.i = (size_t)s.i; o.j = (size_t)s.j; o.stat = s.stat; o.rcMin = s.rc_min; o.prMin = s.pr_min; o.nMin = s.n_min;
        o.pValue = s.pvalue; o.nSim = s.nsim;
      }
    if (nullRows && computeNull) {
      Vdouble s(nnull_), m(nnull_);
      hip(hipSetDevice(devices_[0]));
      hip(hipMemcpy(s.data(), dv_[0].nstat.p, sizeof(double) * nnull_, hipMemcpyDeviceToHost));
      hip(hipMemcpy(m.data(), dv_[0].nnmin.p, sizeof(double) * nnull_, hipMemcpyDeviceToHost));
      nullRows->reserve(nullRows->size() + nnull_);
      for (size_t q = 0; q < nnull_; ++q) nullRows->push_back({s[q], 0, std::numeric_limits<double>::quiet_NaN(), m[q]});
    }
    return rows;
  }

 private:
  static void hip(hipError_t e) { detail::hipCheck(e); }
  struct Buf {   // grow-only device buffer of a rank's arena
    void* p = nullptr;
    size_t bytes = 0;
    template <class T> T* as() const { return static_cast<T*>(p); }
  };
  struct Dev {
    Buf aln, masks, counts, pr, norm, rc, send, recv, nstat, nnmin, rows, count;
    Buf hostRows;                 // pinned: the rows, or (records mode) the 16-byte records
    Buf hostSite;                 // pinned, records mode: posterior rate | norm | rate class of every site
    std::unique_ptr<cmx_pair_row[]> expanded;   // records mode: the rows rebuilt on the host
    size_t expandedCap = 0, nSites = 0;
    uint64_t* hostCount = nullptr;   // pinned
    size_t cap = 0, rowBegin = 0, rowEnd = 0;
    bool fetched = false, compact = false;
    std::vector<Buf*> all() { return {&aln, &masks, &counts, &pr, &norm, &rc, &send, &recv, &nstat, &nnmin, &rows, &count}; }
  };
  static void ensure(Buf& b, size_t bytes) {   // the current device is the rank's
    if (b.bytes >= bytes && b.p) return;
    if (b.p) hip(hipFree(b.p));
    b.p = nullptr; b.bytes = 0;
    hip(hipMalloc(&b.p, bytes ? bytes : 16));
    b.bytes = bytes;
  }
  std::vector<int> devices_;
  std::vector<std::unique_ptr<Engine>> engines_;
  std::vector<hipStream_t> streams_, side_;
  std::vector<hipEvent_t> sideDone_;
  bool compactWanted_ = false;
  std::vector<Dev> dv_;
  Exchange exchange_;
  Buf hostAln_;
  uint32_t* hostMasks_ = nullptr;
  HostRows host_;
  size_t nnull_ = 0;
};

using MultiGpu = BasicMultiGpu<RcclExchange>;              // N distinct devices, one grouped RCCL all-gather
using LoopbackMultiGpu = BasicMultiGpu<LoopbackExchange>;  // N ranks on any devices (all on one: the N > 1 logic on a one-GPU box)

}  // namespace cmx
#endif  // COMAP_MI355X_MULTIGPU_HPP
