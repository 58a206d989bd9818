// Test driver for include/comap_mi355x_multigpu.hpp (one process, N ranks, one all-gather).
//   multigpu_main shards <world> <nrep> <n>        -> "rep <begin> <end>" and "row <begin> <end>" per rank (host only, no GPU call)
//   multigpu_main run <input.bin> <output.bin> <ndev>   -> MultiGpu::computeIntraStats with null on devices 0..ndev-1 (RCCL);
//      input.bin / the rows of output.bin as tests/cpp/adapter_main.cpp "run"; then int64 nnull; f64 null stat[nnull], nmin[nnull]
//   multigpu_main loopback <input.bin> <output.bin> <nranks>   -> the same through LoopbackMultiGpu: nranks contexts on device 0,
//      the all-gather as device-to-device copies.  The call runs TWICE (the second one on the warm arena) and both results,
//      the rows left on the devices (deviceRows) and the rows fetched home (fetchRows) must agree byte for byte.
//   multigpu_main time <input.bin> <nranks> <reps> [loopback]  -> host-to-host milliseconds per enqueueIntraStats + fetchRows
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "comap_mi355x_multigpu.hpp"

template <class T>
static void rd(std::ifstream& f, T* p, size_t n) { f.read(reinterpret_cast<char*>(p), sizeof(T) * n); }
template <class T>
static void wr(std::ofstream& f, const T* p, size_t n) { f.write(reinterpret_cast<const char*>(p), sizeof(T) * n); }

int main(int argc, char** argv) {
  try {
    if (argc == 5 && std::strcmp(argv[1], "shards") == 0) {
      const size_t world = std::strtoull(argv[2], nullptr, 10), nrep = std::strtoull(argv[3], nullptr, 10), n = std::strtoull(argv[4], nullptr, 10);
      for (size_t r = 0; r < world; ++r) {
        const auto a = cmx::replicateShard(r, world, nrep);
        const auto b = cmx::rowShard(r, world, n);
        std::cout << "rep " << a.first << " " << a.second << "\nrow " << b.first << " " << b.second << "\n";
      }
      return 0;
    }
    if (argc == 5 && std::strcmp(argv[1], "run") == 0) {
      std::ifstream in(argv[2], std::ios::binary);
      int32_t h[8];
      uint64_t seed;
      rd(in, h, 8);
      rd(in, &seed, 1);
      const int nn = h[0], T = h[1], S = h[2], C = h[3], N = h[4];
      cmx::TreeArrays t;
      cmx::ModelArrays m;
      t.parent.resize(nn); t.branchLengths.resize(nn); t.leafOfTaxon.resize(T);
      rd(in, t.parent.data(), nn); rd(in, t.branchLengths.data(), nn); rd(in, t.leafOfTaxon.data(), T);
      m.nbStates = S;
      m.generator.resize(S * S); m.frequencies.resize(S); m.rates.resize(C); m.rateProbabilities.resize(C);
      rd(in, m.generator.data(), S * S); rd(in, m.frequencies.data(), S); rd(in, m.rates.data(), C);
      rd(in, m.rateProbabilities.data(), C);
      std::vector<uint8_t> aln(static_cast<size_t>(T) * N);
      rd(in, aln.data(), aln.size());
      std::vector<int> devices(std::atoi(argv[4]));
      for (size_t d = 0; d < devices.size(); ++d) devices[d] = static_cast<int>(d);
      cmx::MultiGpu mg(t, m, devices);
      cmx::CorrelationStatistic stat;
      std::vector<cmx::NullDistributionRow> nul;
      const auto rows = mg.computeIntraStats(aln.data(), N, nullptr, 0, stat, true, seed, h[5], h[6], h[7], cmx::PairFilters(), &nul);
      std::ofstream out(argv[3], std::ios::binary);
      int64_t nr = static_cast<int64_t>(rows.size());
      wr(out, &nr, 1);
      for (const auto& r : rows) {
        int64_t ij[2] = {static_cast<int64_t>(r.i), static_cast<int64_t>(r.j)};
        double v[4] = {r.stat, r.prMin, r.nMin, r.pValue};
        int32_t k[2] = {r.rcMin, r.nSim};
        wr(out, ij, 2); wr(out, v, 4); wr(out, k, 2);
      }
      int64_t nnull = static_cast<int64_t>(nul.size());
      wr(out, &nnull, 1);
      for (const auto& q : nul) wr(out, &q.stat, 1);
      for (const auto& q : nul) wr(out, &q.nMin, 1);
      return 0;
    }
    if ((argc == 5 && std::strcmp(argv[1], "loopback") == 0) || ((argc == 5 || argc == 6) && std::strcmp(argv[1], "time") == 0)) {
      const bool timing = std::strcmp(argv[1], "time") == 0;
      std::ifstream in(argv[2], std::ios::binary);
      int32_t h[8];
      uint64_t seed;
      rd(in, h, 8);
      rd(in, &seed, 1);
      const int nn = h[0], T = h[1], S = h[2], C = h[3], N = h[4];
      cmx::TreeArrays t;
      cmx::ModelArrays m;
      t.parent.resize(nn); t.branchLengths.resize(nn); t.leafOfTaxon.resize(T);
      rd(in, t.parent.data(), nn); rd(in, t.branchLengths.data(), nn); rd(in, t.leafOfTaxon.data(), T);
      m.nbStates = S;
      m.generator.resize(S * S); m.frequencies.resize(S); m.rates.resize(C); m.rateProbabilities.resize(C);
      rd(in, m.generator.data(), S * S); rd(in, m.frequencies.data(), S); rd(in, m.rates.data(), C);
      rd(in, m.rateProbabilities.data(), C);
      std::vector<uint8_t> aln(static_cast<size_t>(T) * N);
      rd(in, aln.data(), aln.size());
      cmx::CorrelationStatistic stat;
      if (timing) {
        const int nranks = std::atoi(argv[3]), reps = std::atoi(argv[4]);
        const bool loop = argc == 6;
        auto run = [&](auto& mg) {
          double best = 1e300, sum = 0;
          for (int it = 0; it < reps + 1; ++it) {   // the first call sizes the arena and is not counted
            const auto t0 = std::chrono::steady_clock::now();
            mg.enqueueIntraStats(aln.data(), N, nullptr, 0, stat, true, seed, h[5], h[6], h[7]);
            const auto& hr = mg.fetchRows();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (it) { best = std::min(best, ms); sum += ms; }
            if (hr.total() != static_cast<size_t>(N) * (N - 1) / 2) throw cmx::Exception("row count");
          }
          const double pairs = 0.5 * N * (N - 1.0) + static_cast<double>(h[5]) * h[6];
          std::printf("{\"ranks\": %d, \"exchange\": \"%s\", \"ms_per_step_mean\": %.3f, \"ms_per_step_best\": %.3f, \"pairs_per_s\": %.4e}\n", nranks,
                      loop ? "loopback" : "rccl", sum / reps, best, pairs / (sum / reps * 1e-3));
        };
        const bool records = std::getenv("CMX_MG_RECORDS") != nullptr;   // 16-byte records over PCIe, rows rebuilt on the host
        if (loop) { cmx::LoopbackMultiGpu mg(t, m, std::vector<int>(nranks, 0)); mg.enableCompactTransfer(records); run(mg); }
        else { std::vector<int> dv(nranks); for (int d = 0; d < nranks; ++d) dv[d] = d; cmx::MultiGpu mg(t, m, dv); mg.enableCompactTransfer(records); run(mg); }
        return 0;
      }
      const int nranks = std::atoi(argv[4]);
      cmx::LoopbackMultiGpu mg(t, m, std::vector<int>(nranks, 0));
      std::vector<cmx::NullDistributionRow> nul;
      const auto first = mg.computeIntraStats(aln.data(), N, nullptr, 0, stat, true, seed, h[5], h[6], h[7], cmx::PairFilters(), nullptr);
      const auto rows = mg.computeIntraStats(aln.data(), N, nullptr, 0, stat, true, seed, h[5], h[6], h[7], cmx::PairFilters(), &nul);
      if (first.size() != rows.size()) throw cmx::Exception("second call on the warm arena returns a different row count");
      for (size_t q = 0; q < rows.size(); ++q)
        if (first[q].i != rows[q].i || first[q].j != rows[q].j || std::memcmp(&first[q].stat, &rows[q].stat, 8) || first[q].nSim != rows[q].nSim ||
            std::memcmp(&first[q].pValue, &rows[q].pValue, 8))
          throw cmx::Exception("second call on the warm arena differs from the first");
      // (computeIntraStats asks for host rows only: the devices wrote 16-byte records and the host rebuilt the rows.)  The same
      // analysis in rows mode: rows left on the devices == rows fetched home == the rows rebuilt from the records; every rank
      // holds the same merged null
      bool threw = false;
      try { (void)mg.deviceRows(0); } catch (const cmx::Exception&) { threw = true; }
      if (!threw) throw cmx::Exception("deviceRows in records mode did not throw");
      mg.enqueueIntraStats(aln.data(), N, nullptr, 0, stat, true, seed, h[5], h[6], h[7]);
      const auto& hr = mg.fetchRows();
      {
        size_t k = 0;
        for (int r = 0; r < nranks; ++r)
          for (size_t q = 0; q < hr.count[r]; ++q, ++k) {
            const cmx_pair_row& a = hr.rows[r][q];
            if (k >= rows.size() || static_cast<size_t>(a.i) != rows[k].i || static_cast<size_t>(a.j) != rows[k].j || std::memcmp(&a.stat, &rows[k].stat, 8) ||
                std::memcmp(&a.pvalue, &rows[k].pValue, 8) || a.nsim != rows[k].nSim || a.rc_min != rows[k].rcMin ||
                std::memcmp(&a.pr_min, &rows[k].prMin, 8) || std::memcmp(&a.n_min, &rows[k].nMin, 8))
              throw cmx::Exception("rows mode differs from records mode");
          }
        if (k != rows.size()) throw cmx::Exception("rows mode: row count differs from records mode");
      }
      std::vector<double> n0(nul.size()), nr(nul.size());
      for (int r = 0; r < nranks; ++r) {
        const auto d = mg.deviceRows(r);
        if (d.count != hr.count[r]) throw cmx::Exception("deviceRows / fetchRows counts differ");
        std::vector<cmx_pair_row> tmp(d.count);
        if (d.count && hipMemcpy(tmp.data(), d.rows, sizeof(cmx_pair_row) * d.count, hipMemcpyDeviceToHost) != hipSuccess) throw cmx::Exception("copy");
        if (d.count && std::memcmp(tmp.data(), hr.rows[r], sizeof(cmx_pair_row) * d.count)) throw cmx::Exception("deviceRows differ from fetchRows");
        if (hipMemcpy(nr.data(), mg.deviceNullStat(r), sizeof(double) * nr.size(), hipMemcpyDeviceToHost) != hipSuccess) throw cmx::Exception("copy");
        if (r == 0) n0 = nr;
        else if (std::memcmp(n0.data(), nr.data(), sizeof(double) * nr.size())) throw cmx::Exception("ranks hold different merged nulls");
      }
      std::ofstream out(argv[3], std::ios::binary);
      int64_t nrw = static_cast<int64_t>(rows.size());
      wr(out, &nrw, 1);
      for (const auto& r : rows) {
        int64_t ij[2] = {static_cast<int64_t>(r.i), static_cast<int64_t>(r.j)};
        double v[4] = {r.stat, r.prMin, r.nMin, r.pValue};
        int32_t k[2] = {r.rcMin, r.nSim};
        wr(out, ij, 2); wr(out, v, 4); wr(out, k, 2);
      }
      int64_t nnull = static_cast<int64_t>(nul.size());
      wr(out, &nnull, 1);
      for (const auto& q : nul) wr(out, &q.stat, 1);
      for (const auto& q : nul) wr(out, &q.nMin, 1);
      return 0;
    }
    std::cerr << "usage: multigpu_main shards world nrep n | run in.bin out.bin ndev | loopback in.bin out.bin nranks | time in.bin nranks reps [loopback]\n";
    return 2;
  } catch (cmx::Exception& e) {
    std::cerr << "cmx::Exception: " << e.what() << "\n";
    return 1;
  }
}
