// Test driver for include/comap_mi355x_adapter.hpp (the C++ mirror of the reference's interface).
//   adapter_main domain <lo> <hi> <n> <x>...         -> prints getIndex(x) or -1 per x (host logic only, no GPU)
//   adapter_main run <input.bin> <output.bin>        -> getVectors + computeIntraStats with null on the GPU
//   adapter_main groups <tree.bin>                   -> ClusterTools::getGroups + io::writeGroups to stdout (host only);
//      tree.bin: int32 n, maxGroupSize; int32 merge[2(n-1)]; f64 dmax, stat, nmin [n-1]; int32 coords[n], isConstant[n]
//   adapter_main cluster <input.bin> <method> <maxsize>   -> ClusterTools::cluster + getGroups + writeGroups of the observed data (GPU)
//   adapter_main clusternull <input.bin> cor|euclidian <method> <nsites> <nrep> <maxsize>   -> null groups file (GPU)
//   adapter_main candidates <input.bin> <omega> <minSim> <repRAM> <maxTrials> <seed>   -> candidate-group test (GPU)
//   adapter_main mica <input.bin> <method> <withModel 0|1> <zstat> -> cmx::Mica::analyse: output.file to stdout, null.output.file to stderr (GPU)
//   adapter_main inter <input.bin> <indep 0|1> <quirk 0|1> -> CoETools::computeInterStats of the alignment's two halves (GPU)
//   adapter_main vec <input.bin>                     -> cmx::io::writeToStream of a mapping to stdout (host only);
//      input.bin: int32 N, B; int32 coords[N]; f64 blen[B]; f64 counts[N*B] (site-major)
// input.bin (little endian): int32 nn, T, S, C, N, repCPU, repRAM, nclasses; uint64 seed;
//   int32 parent[nn]; f64 blen[nn]; int32 lot[T]; f64 Q[S*S], pi[S], rates[C], probs[C]; uint8 aln[T*N]
// output.bin: int64 nrows; per row: int64 i, j; f64 stat, prMin, nMin, pValue; int32 rcMin, nSim;
//   then f64 counts[N*B], f64 norms[N]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "comap_mi355x_adapter.hpp"

template <class T>
static void rd(std::ifstream& f, T* p, size_t n) { f.read(reinterpret_cast<char*>(p), sizeof(T) * n); }
template <class T>
static void wr(std::ofstream& f, const T* p, size_t n) { f.write(reinterpret_cast<const char*>(p), sizeof(T) * n); }

int main(int argc, char** argv) {
  try {
    if (argc >= 6 && std::strcmp(argv[1], "domain") == 0) {
      cmx::Domain d(std::atof(argv[2]), std::atof(argv[3]), static_cast<size_t>(std::atoi(argv[4])));
      for (int a = 5; a < argc; ++a) {
        try {
          std::cout << d.getIndex(std::strtod(argv[a], nullptr)) << "\n";
        } catch (cmx::OutOfRangeException&) {
          std::cout << -1 << "\n";
        }
      }
      return 0;
    }
    if (argc == 4 && std::strcmp(argv[1], "matrices") == 0) {   // GPU: <in.bin: int32 n1, n2, dim; f64 v1[n1][dim], v2[n2][dim]> <out.bin>
      // cmx::AnalysisTools::compute{ScalarProduct,Cosinus,Correlation,Covariance}Matrix (AnalysisTools.h:93-190) on a
      // model-less engine: for each of the four, the one-set form of v1, the two-set form, and (n1 == n2) the independent one
      std::ifstream in(argv[2], std::ios::binary);
      int32_t h[3];
      rd(in, h, 3);
      cmx::VVdouble v1(h[0], cmx::Vdouble(h[2])), v2(h[1], cmx::Vdouble(h[2]));
      for (auto& v : v1) rd(in, v.data(), v.size());
      for (auto& v : v2) rd(in, v.data(), v.size());
      cmx::Engine eng(0);
      std::ofstream out(argv[3], std::ios::binary);
      auto put = [&](const cmx::VVdouble& m) { for (const auto& r : m) wr(out, r.data(), r.size()); };
      put(cmx::AnalysisTools::computeScalarProductMatrix(eng, v1)); put(cmx::AnalysisTools::computeScalarProductMatrix(eng, v1, v2, false));
      put(cmx::AnalysisTools::computeCosinusMatrix(eng, v1)); put(cmx::AnalysisTools::computeCosinusMatrix(eng, v1, v2, false));
      put(cmx::AnalysisTools::computeCorrelationMatrix(eng, v1)); put(cmx::AnalysisTools::computeCorrelationMatrix(eng, v1, v2, false));
      put(cmx::AnalysisTools::computeCovarianceMatrix(eng, v1)); put(cmx::AnalysisTools::computeCovarianceMatrix(eng, v1, v2, false));
      if (h[0] == h[1]) put(cmx::AnalysisTools::computeCorrelationMatrix(eng, v1, v2, true));
      else {
        try { cmx::AnalysisTools::computeCorrelationMatrix(eng, v1, v2, true); return 3; }
        catch (cmx::DimensionException& e) { std::cout << "DimensionException: " << e.what() << "\n"; }
      }
      return 0;
    }
    if (argc == 4 && std::strcmp(argv[1], "run") == 0) {
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
      cmx::Engine eng(t, m, 0);
      auto mapping = cmx::CoETools::getVectors(eng, aln.data(), N);
      cmx::CorrelationStatistic stat;
      auto rows = cmx::CoETools::computeIntraStats(eng, *mapping, stat, true, seed, h[5], h[6], h[7]);
      std::ofstream out(argv[3], std::ios::binary);
      int64_t nr = static_cast<int64_t>(rows.size());
      wr(out, &nr, 1);
      for (const auto& r : rows) {
        int64_t ij[2] = {static_cast<int64_t>(r.i), static_cast<int64_t>(r.j)};
        double v[4] = {r.stat, r.prMin, r.nMin, r.pValue};
        int32_t k[2] = {r.rcMin, r.nSim};
        wr(out, ij, 2); wr(out, v, 4); wr(out, k, 2);
      }
      wr(out, mapping->data(), static_cast<size_t>(N) * eng.getNumberOfBranches());
      cmx::Vdouble norms = cmx::AnalysisTools::computeNorms(*mapping);
      wr(out, norms.data(), norms.size());
      return 0;
    }
    if (argc == 3 && std::strcmp(argv[1], "groups") == 0) {   // host only: a clustering tree -> the groups table
      std::ifstream in(argv[2], std::ios::binary);
      int32_t h[2];
      rd(in, h, 2);
      const size_t n = h[0], maxSize = h[1];
      cmx::ClusteringTree t;
      t.n = n;
      t.merge.resize(2 * (n - 1)); t.dmax.resize(n - 1); t.stat.resize(n - 1); t.nmin.resize(n - 1);
      std::vector<int32_t> coords(n), isc(n);
      rd(in, t.merge.data(), t.merge.size()); rd(in, t.dmax.data(), n - 1); rd(in, t.stat.data(), n - 1);
      rd(in, t.nmin.data(), n - 1); rd(in, coords.data(), n); rd(in, isc.data(), n);
      std::vector<std::string> names(n);
      std::vector<bool> isConst(n);
      for (size_t i = 0; i < n; ++i) { names[i] = std::to_string(coords[i]); isConst[i] = isc[i] != 0; }
      cmx::io::writeGroups(cmx::ClusterTools::getGroups(t), names, isConst, maxSize, std::cout);
      return 0;
    }
    if (argc == 5 && std::strcmp(argv[1], "cluster") == 0) {   // GPU: <input.bin of "run"> <method> <maxsize>: observed clustering
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
      cmx::Engine eng(t, m, 0);
      auto mapping = cmx::CoETools::getVectors(eng, aln.data(), N);
      cmx::StatisticBasedDistance dist(std::make_shared<cmx::CorrelationStatistic>(), 1.);
      cmx::ClusteringTree tree = cmx::ClusterTools::cluster(eng, dist, std::atoi(argv[3]), *mapping, true);
      std::vector<std::string> names(N);
      std::vector<bool> isConst(N, false);
      for (int i = 0; i < N; ++i) names[i] = std::to_string(10 + i);
      cmx::io::writeGroups(cmx::ClusterTools::getGroups(tree), names, isConst, static_cast<size_t>(std::atoi(argv[4])), std::cout);
      std::cout << tree.distances[1] << " " << tree.distances[static_cast<size_t>(N) * 2 + 1] << "\n";
      return 0;
    }
    if (argc == 8 && std::strcmp(argv[1], "clusternull") == 0) {   // GPU: <input.bin of "run"> dist method nsites nrep maxsize
      std::ifstream in(argv[2], std::ios::binary);
      int32_t h[8];
      uint64_t seed;
      rd(in, h, 8);
      rd(in, &seed, 1);
      const int nn = h[0], T = h[1], S = h[2], C = h[3];
      cmx::TreeArrays t;
      cmx::ModelArrays m;
      t.parent.resize(nn); t.branchLengths.resize(nn); t.leafOfTaxon.resize(T);
      rd(in, t.parent.data(), nn); rd(in, t.branchLengths.data(), nn); rd(in, t.leafOfTaxon.data(), T);
      m.nbStates = S;
      m.generator.resize(S * S); m.frequencies.resize(S); m.rates.resize(C); m.rateProbabilities.resize(C);
      rd(in, m.generator.data(), S * S); rd(in, m.frequencies.data(), S); rd(in, m.rates.data(), C);
      rd(in, m.rateProbabilities.data(), C);
      cmx::Engine eng(t, m, 0);
      const int method = std::atoi(argv[4]);
      const size_t nsites = std::atoi(argv[5]), nrep = std::atoi(argv[6]), maxSize = std::atoi(argv[7]);
      if (std::strcmp(argv[3], "euclidian") == 0)
        cmx::ClusterTools::computeGlobalDistanceDistribution(eng, cmx::EuclidianDistance(), method, seed, nsites, nrep, maxSize, &std::cout);
      else
        cmx::ClusterTools::computeGlobalDistanceDistribution(
            eng, cmx::StatisticBasedDistance(std::make_shared<cmx::CorrelationStatistic>(), 1.), method, seed, nsites, nrep,
            maxSize, &std::cout);
      return 0;
    }
    if (argc == 8 && std::strcmp(argv[1], "candidates") == 0) {   // GPU: <input.bin of "run"> omega minSim repRAM maxTrials seed2
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
      cmx::Engine eng(t, m, 0);
      auto mapping = cmx::CoETools::getVectors(eng, aln.data(), N);
      cmx::CorrelationStatistic stat;
      cmx::CandidateGroupSet set(&stat, static_cast<unsigned>(std::atoi(argv[4])));
      // groups: sites (3g, 3g+1, 3g+2) for g = 0..3, the last one flagged not analysable
      for (int g = 0; g < 4; ++g) {
        cmx::CandidateGroup c;
        for (int k = 0; k < 3; ++k) c.addSite(cmx::CandidateSite(static_cast<size_t>(3 * g + k)));
        c.computeNormRanges(std::atof(argv[3]), *mapping);
        c.computeStatisticValue(eng, stat, *mapping);
        if (g == 3) c.setAnalysable(false);
        set.addCandidate(c);
      }
      cmx::CoETools::computePValuesForCandidateGroups(set, eng, std::strtoull(argv[7], nullptr, 10),
                                                      static_cast<unsigned>(std::atoi(argv[5])), static_cast<unsigned>(std::atoi(argv[6])));
      std::cout.precision(17);
      for (size_t g = 0; g < set.size(); ++g)
        std::cout << set[g].getStatisticValue() << " " << set.getN1ForGroup(g) << " " << set.getN2ForGroup(g) << " "
                  << set.getPValueForGroup(g) << "\n";
      std::cout << set.getNumberOfTrials() << " " << set.getNumberOfBatches() << "\n";
      return 0;
    }
    if ((argc == 6 && std::strcmp(argv[1], "mica") == 0) || (argc == 5 && std::strcmp(argv[1], "inter") == 0)) {
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
      cmx::Engine eng(t, m, 0);
      std::cout.precision(6);
      if (std::strcmp(argv[1], "inter") == 0) {   // data set 1 = the first half of the columns, data set 2 = the second half
        const int n1 = N / 2, n2 = N - n1;
        std::vector<uint8_t> a1(static_cast<size_t>(T) * n1), a2(static_cast<size_t>(T) * n2);
        for (int tx = 0; tx < T; ++tx) {
          std::copy(aln.begin() + static_cast<size_t>(tx) * N, aln.begin() + static_cast<size_t>(tx) * N + n1, a1.begin() + static_cast<size_t>(tx) * n1);
          std::copy(aln.begin() + static_cast<size_t>(tx) * N + n1, aln.begin() + static_cast<size_t>(tx + 1) * N, a2.begin() + static_cast<size_t>(tx) * n2);
        }
        auto m1 = cmx::CoETools::getVectors(eng, a1.data(), n1);
        auto m2 = cmx::CoETools::getVectors(eng, a2.data(), n2);
        cmx::CorrelationStatistic stat;
        cmx::PairFilters f1, f2;
        f1.minRateClass = 1; f2.minRate = 0.2; f1.minStatistic = 0.05;
        const auto rows = cmx::CoETools::computeInterStats(eng, *m1, *m2, stat, std::atoi(argv[3]) != 0, f1, f2, std::atoi(argv[4]) != 0);
        std::vector<int> c1(n1), c2(n2);
        for (int i = 0; i < n1; ++i) c1[i] = 100 + i;
        for (int i = 0; i < n2; ++i) c2[i] = 500 + i;
        cmx::io::writeIntraStats(rows, c1, false, std::cout, &c2);
        return 0;
      }
      cmx::Mica::Options opt;
      opt.nullMethod = argv[3];
      opt.nbRepCPU = h[5]; opt.nbRepRAM = h[6]; opt.nbRateClasses = h[7];
      opt.seed = seed;
      opt.zScoreStat = argv[5];
      opt.maxNbPermutations = 200;
      const bool withModel = std::atoi(argv[4]) != 0;
      cmx::Vdouble norms;
      if (withModel) norms = cmx::AnalysisTools::computeNorms(*cmx::CoETools::getVectors(eng, aln.data(), N));
      const cmx::Mica::Result res = cmx::Mica::analyse(eng, aln.data(), T, N, S, nullptr, 0, withModel ? &norms : nullptr, opt);
      std::vector<int> coords(N);
      for (int i = 0; i < N; ++i) coords[i] = 10 + i;
      cmx::io::writeMica(res, coords, std::cout);
      std::cerr.precision(6);
      if (!res.null.empty()) cmx::io::writeMicaNull(res, withModel, std::cerr);
      return 0;
    }
    if (argc == 3 && std::strcmp(argv[1], "vec") == 0) {
      std::ifstream in(argv[2], std::ios::binary);
      int32_t h[2];
      rd(in, h, 2);
      const size_t N = h[0], B = h[1];
      std::vector<int> coords(N);
      cmx::Vdouble bl(B);
      rd(in, coords.data(), N); rd(in, bl.data(), B);
      cmx::ProbabilisticSubstitutionMapping mapping(N, B, 1);
      rd(in, mapping.data(), N * B);
      cmx::io::writeToStream(mapping, bl, coords, 0, std::cout);
      std::vector<cmx::NullDistributionRow> nul = {{0.5, 1, 0.25, 3.5}, {-1e-7, 0, 2., 0.125}};
      cmx::io::writeNull(nul, std::cerr);
      return 0;
    }
    std::cerr << "usage: adapter_main domain lo hi n x... | run in.bin out.bin\n";
    return 2;
  } catch (cmx::Exception& e) {   // the reference's main catches bpp::Exception and exits (CoMap.cpp:730-734)
    std::cerr << "cmx::Exception: " << e.what() << "\n";
    return 1;
  }
}
