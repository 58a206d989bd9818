// Test driver for include/comap_mi355x_multigpu.hpp (one process, N devices, one RCCL all-gather).
//   multigpu_main shards <world> <nrep> <n>        -> "rep <begin> <end>" and "row <begin> <end>" per rank (host only, no GPU call)
//   multigpu_main run <input.bin> <output.bin> <ndev>   -> MultiGpu::computeIntraStats with null on devices 0..ndev-1;
//      input.bin / the rows of output.bin as tests/cpp/adapter_main.cpp "run"; then int64 nnull; f64 null stat[nnull], nmin[nnull]
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
    std::cerr << "usage: multigpu_main shards world nrep n | run in.bin out.bin ndev\n";
    return 2;
  } catch (cmx::Exception& e) {
    std::cerr << "cmx::Exception: " << e.what() << "\n";
    return 1;
  }
}
