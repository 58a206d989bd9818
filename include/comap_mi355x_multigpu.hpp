// comap_mi355x_multigpu.hpp -- one process driving the N MI355X of one node through the C-ABI (comap_mi355x.h) and ONE
// RCCL collective: SURVEY.md 8(b) "multi-GPU driver owns 8 ctxs + one RCCL communicator", 8(e).
//
// What shards (same arithmetic as comap_amd/distributed.py, which the one-process-per-GPU Python path uses):
//   * the parametric-bootstrap null (AnalysisTools::getNullDistributionIntraDR, CoMap/AnalysisTools.cpp:564-658): replicates
//     are independent; device r maps the contiguous range replicateShard(r, N, repCPU); the counter RNG is keyed by the
//     GLOBAL simulated-site index, so the merged null is bit-identical for any N.  ONE ncclAllGather (grouped over the N
//     communicators of this process) gives every device the whole null in the reference's replicate order;
//   * the observed pair loop (CoETools::computeIntraStats, CoMap/CoETools.cpp:672-724): rows of the upper triangle split
//     by PAIR count (rowShard); every device maps the observed alignment itself (it is small) and compacts the rows of
//     its range; ranges are contiguous in i, so the devices' rows concatenate to the single-GPU output in the
//     reference's (i, j) order -- no second collective on the data path.
// xGMI is point-to-point: the exchange is 16 bytes per null pair (20 MB per device at the north-star target with 8
// devices), far below any link limit, so it is a single all-gather and not a ring of smaller ones.
//
// Needs the HIP runtime API and RCCL headers (/opt/rocm/include); link with -lcomap_mi355x -lrccl -lamdhip64.
#ifndef COMAP_MI355X_MULTIGPU_HPP
#define COMAP_MI355X_MULTIGPU_HPP

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

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

class MultiGpu {
 public:
  // one context, one stream and one RCCL communicator per listed device (distinct devices of this node)
  MultiGpu(const TreeArrays& tree, const ModelArrays& model, const std::vector<int>& devices) : devices_(devices) {
    if (devices.empty()) throw Exception("MultiGpu: no device given.");
    for (int d : devices) engines_.emplace_back(new Engine(tree, model, d));
    streams_.assign(devices.size(), nullptr);
    comms_.assign(devices.size(), nullptr);
    for (size_t r = 0; r < devices.size(); ++r) {
      hip(hipSetDevice(devices[r]));
      hip(hipStreamCreate(&streams_[r]));
    }
    nccl(ncclCommInitAll(comms_.data(), (int)devices.size(), devices.data()));
  }
  ~MultiGpu() {
    for (size_t r = 0; r < devices_.size(); ++r) {
      (void)hipSetDevice(devices_[r]);
      if (comms_[r]) (void)ncclCommDestroy(comms_[r]);
      if (streams_[r]) (void)hipStreamDestroy(streams_[r]);
    }
  }
  MultiGpu(const MultiGpu&) = delete;
  MultiGpu& operator=(const MultiGpu&) = delete;
  size_t size() const { return devices_.size(); }
  const Engine& engine(size_t r) const { return *engines_[r]; }

  // CoETools::getVectors + computeIntraStats with their null (CoMap.cpp:155, 363) over all devices.  aln: [taxon][site]
  // codes.  Returns the rows of statistics.txt in the reference's order; nullRows (optional) receives the merged null
  // (Stat, Nmin; RCmin / PRmin are not exchanged: 0 / NaN) in replicate order.
  std::vector<IntraStatRow> computeIntraStats(const uint8_t* aln, size_t nbSites, const uint32_t* masks, size_t nbMasks,
                                              const Statistic& statistic, bool computeNull, uint64_t seed, size_t nbRepCPU = 100,
                                              size_t nbRepRAM = 1000, size_t nbRateClasses = 10, const PairFilters& f = PairFilters(),
                                              std::vector<NullDistributionRow>* nullRows = nullptr) {
    const size_t N = size(), n = nbSites, T = engines_[0]->getNumberOfTaxa();
    const size_t BK = engines_[0]->getNumberOfBranches() * engines_[0]->getNumberOfSubstitutionTypes();
    if (n < 2) throw Exception("MultiGpu::computeIntraStats: at least two sites are needed.");
    const size_t nnull = computeNull ? nbRepCPU * nbRepRAM : 0;
    size_t mx = 0;   // entries of the largest null shard: shards are padded to it for the collective
    for (size_t r = 0; r < N; ++r) {
      const auto s = replicateShard(r, N, nbRepCPU);
      mx = std::max(mx, (s.second - s.first) * nbRepRAM);
    }
    cmx_pair_filters pf;
    pf.min_rate_class = f.minRateClass; pf.max_rate_class_diff = f.maxRateClassDiff;
    pf.min_rate = f.minRate; pf.max_rate_diff = f.maxRateDiff; pf.min_statistic = f.minStatistic;
    struct Dev {
      uint8_t* aln = nullptr; uint32_t* masks = nullptr;
      double *counts = nullptr, *pr = nullptr, *norm = nullptr, *send = nullptr, *recv = nullptr, *nstat = nullptr, *nnmin = nullptr;
      int32_t* rc = nullptr; cmx_pair_row* rows = nullptr; uint64_t* count = nullptr;
      size_t cap = 0, rowBegin = 0, rowEnd = 0;
    };
    std::vector<Dev> dv(N);
    auto freeAll = [&]() {
      for (size_t r = 0; r < N; ++r) {
        (void)hipSetDevice(devices_[r]);
        Dev& d = dv[r];
        for (void* p : {(void*)d.aln, (void*)d.masks, (void*)d.counts, (void*)d.pr, (void*)d.norm, (void*)d.send, (void*)d.recv,
                        (void*)d.nstat, (void*)d.nnmin, (void*)d.rc, (void*)d.rows, (void*)d.count})
          if (p) (void)hipFree(p);
      }
    };
    try {
      // ---- every device: observed alignment up, mapped; its null shard into the send buffer [2][mx] (NaN padded)
      for (size_t r = 0; r < N; ++r) {
        Dev& d = dv[r];
        const Engine& e = *engines_[r];
        hip(hipSetDevice(devices_[r]));
        hipStream_t st = streams_[r];
        const auto rs = rowShard(r, N, n);
        d.rowBegin = rs.first; d.rowEnd = rs.second;
        d.cap = (d.rowEnd - d.rowBegin) * (n - 1) - (d.rowEnd * (d.rowEnd - 1) - d.rowBegin * (d.rowBegin - 1)) / 2;
        hip(hipMalloc((void**)&d.aln, T * n));
        hip(hipMalloc((void**)&d.counts, sizeof(double) * BK * n));
        hip(hipMalloc((void**)&d.pr, sizeof(double) * n));
        hip(hipMalloc((void**)&d.norm, sizeof(double) * n));
        hip(hipMalloc((void**)&d.rc, sizeof(int32_t) * n));
        hip(hipMalloc((void**)&d.rows, sizeof(cmx_pair_row) * std::max<size_t>(d.cap, 1)));
        hip(hipMalloc((void**)&d.count, sizeof(uint64_t)));
        hip(hipMemcpyAsync(d.aln, aln, T * n, hipMemcpyHostToDevice, st));
        if (masks) {
          const int S = e.getNumberOfStates();
          std::vector<uint32_t> mk(256, S >= 32 ? 0xffffffffu : ((1u << S) - 1u));
          for (size_t i = 0; i < nbMasks && i < 256; ++i) mk[i] = masks[i];
          hip(hipMalloc((void**)&d.masks, 256 * sizeof(uint32_t)));
          hip(hipMemcpy(d.masks, mk.data(), 256 * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        e.check(cmx_map_sites_dev(e.ctx(), d.aln, n, n, d.masks, d.counts, n, nullptr, d.pr, d.rc, d.norm, st));
        if (computeNull) {
          const auto s = replicateShard(r, N, nbRepCPU);
          hip(hipMalloc((void**)&d.send, sizeof(double) * 2 * mx));
          hip(hipMalloc((void**)&d.recv, sizeof(double) * 2 * mx * N));
          hip(hipMalloc((void**)&d.nstat, sizeof(double) * nnull));
          hip(hipMalloc((void**)&d.nnmin, sizeof(double) * nnull));
          hip(hipMemsetAsync(d.send, 0xFF, sizeof(double) * 2 * mx, st));   // all-ones bytes are a NaN
          if (s.second > s.first)
            e.check(cmx_null_intra_dev(e.ctx(), statistic.kind(), statistic.params(), seed, s.first, s.second, nbRepRAM, nullptr, d.send,
                                       nullptr, nullptr, d.send + mx, st));
        }
      }
      // ---- the path's one exchange
      if (computeNull) {
        nccl(ncclGroupStart());
        for (size_t r = 0; r < N; ++r) nccl(ncclAllGather(dv[r].send, dv[r].recv, 2 * mx, ncclDouble, comms_[r], streams_[r]));
        nccl(ncclGroupEnd());
      }
      // ---- every device: the shards back into replicate order, then the rows of its range against the merged null
      for (size_t r = 0; r < N; ++r) {
        Dev& d = dv[r];
        const Engine& e = *engines_[r];
        hip(hipSetDevice(devices_[r]));
        hipStream_t st = streams_[r];
        for (size_t q = 0; computeNull && q < N; ++q) {
          const auto s = replicateShard(q, N, nbRepCPU);
          const size_t cnt = (s.second - s.first) * nbRepRAM, off = s.first * nbRepRAM;
          if (!cnt) continue;
          hip(hipMemcpyAsync(d.nstat + off, d.recv + q * 2 * mx, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
          hip(hipMemcpyAsync(d.nnmin + off, d.recv + q * 2 * mx + mx, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
        }
        e.check(cmx_intra_rows_range_dev(e.ctx(), statistic.kind(), statistic.params(), d.counts, n, n, d.rc, d.pr, d.norm,
                                         computeNull ? d.nstat : nullptr, computeNull ? d.nnmin : nullptr, nnull, (int)nbRateClasses, &pf,
                                         d.rowBegin, d.rowEnd, d.rows, d.cap, d.count, st));
      }
      // ---- rows home, in rank order == the reference's (i, j) order
      std::vector<IntraStatRow> rows;
      for (size_t r = 0; r < N; ++r) {
        Dev& d = dv[r];
        hip(hipSetDevice(devices_[r]));
        hip(hipStreamSynchronize(streams_[r]));
        uint64_t count = 0;
        hip(hipMemcpy(&count, d.count, sizeof(uint64_t), hipMemcpyDeviceToHost));
        std::vector<cmx_pair_row> raw((size_t)std::min<uint64_t>(count, d.cap));
        if (!raw.empty()) hip(hipMemcpy(raw.data(), d.rows, sizeof(cmx_pair_row) * raw.size(), hipMemcpyDeviceToHost));
        for (const cmx_pair_row& q : raw) {
          IntraStatRow o;
          o.i = (size_t)q.i; o.j = (size_t)q.j; o.stat = q.stat; o.rcMin = q.rc_min; o.prMin = q.pr_min; o.nMin = q.n_min;
          o.pValue = q.pvalue; o.nSim = q.nsim;
          rows.push_back(o);
        }
        if (r == 0 && nullRows && computeNull) {
          Vdouble s(nnull), m(nnull);
          hip(hipMemcpy(s.data(), d.nstat, sizeof(double) * nnull, hipMemcpyDeviceToHost));
          hip(hipMemcpy(m.data(), d.nnmin, sizeof(double) * nnull, hipMemcpyDeviceToHost));
          for (size_t q = 0; q < nnull; ++q) nullRows->push_back({s[q], 0, std::numeric_limits<double>::quiet_NaN(), m[q]});
        }
      }
      freeAll();
      return rows;
    } catch (...) {
      freeAll();
      throw;
    }
  }

 private:
  static void hip(hipError_t e) {
    if (e != hipSuccess) throw Exception(std::string("MultiGpu: HIP error: ") + hipGetErrorString(e));
  }
  static void nccl(ncclResult_t e) {
    if (e != ncclSuccess) throw Exception(std::string("MultiGpu: RCCL error: ") + ncclGetErrorString(e));
  }
  std::vector<int> devices_;
  std::vector<std::unique_ptr<Engine>> engines_;
  std::vector<hipStream_t> streams_;
  std::vector<ncclComm_t> comms_;
};

}  // namespace cmx
#endif  // COMAP_MI355X_MULTIGPU_HPP
