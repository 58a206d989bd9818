// comap_mi355x_adapter.hpp -- C++ host-side mirror of the reference's interface for the pairwise path, on top of
// the C-ABI (comap_mi355x.h).  Header-only, no Bio++: it uses the same class and method names and the same
// argument meaning / error behaviour as the reference so that call sites read alike:
//
//   reference (jydu/comap)                                   here (namespace cmx)
//   ------------------------------------------------------   -------------------------------------------------
//   Domain(a, b, n), getIndex, OutOfRangeException           cmx::Domain, cmx::OutOfRangeException
//     CoMap/Domain.h:59-152, CoMap/Domain.cpp:46-122
//   Statistic / CorrelationStatistic / CompensationStatistic cmx::Statistic hierarchy: kind() + getValuesForAllPairs()
//     ... getValueForPair(v1, v2)  CoMap/Statistics.h:57-329    (the all-pairs loop of CoETools.cpp:672-692 in one call)
//   Distance / StatisticBasedDistance / CompensationDistance cmx::StatisticBasedDistance etc. (CoMap/Distance.h:316-424)
//   CoETools::getVectors             CoMap/CoETools.cpp:366   cmx::CoETools::getVectors -> ProbabilisticSubstitutionMapping
//   AnalysisTools::computeNorms      AnalysisTools.cpp:343    cmx::AnalysisTools::computeNorms
//   AnalysisTools::getNullDistributionIntraDR  :564-658        cmx::AnalysisTools::getNullDistributionIntraDR
//   CoETools::computeIntraStats      CoETools.cpp:604-728     cmx::CoETools::computeIntraStats (rows instead of a TSV stream)
//
// The Bio++-typed originals take DRTreeLikelihoodInterface / SubstitutionCountInterface / SequenceSimulatorInterface
// objects; their roles (tree + model + rates, count registers/weights, simulator) are all carried by cmx::Engine,
// built from plain arrays.  INTEGRATION.md shows the glue a CoMap maintainer would write to fill those arrays from
// the Bio++ objects.  Errors: every failing C-ABI status becomes a cmx::Exception (the reference throws
// bpp::Exception and catches it in main, CoMap/CoMap.cpp:730-734).
#ifndef COMAP_MI355X_ADAPTER_HPP
#define COMAP_MI355X_ADAPTER_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <ostream>
#include <vector>

#include "comap_mi355x.h"

namespace cmx {

typedef std::vector<double> Vdouble;
typedef std::vector<Vdouble> VVdouble;

class Exception : public std::runtime_error {
 public:
  explicit Exception(const std::string& what) : std::runtime_error(what) {}
};
class DimensionException : public Exception {
 public:
  DimensionException(const std::string& where, size_t got, size_t expected)
      : Exception(where + " dimension " + std::to_string(got) + " != " + std::to_string(expected)), got_(got), expected_(expected) {}
  size_t got() const { return got_; }
  size_t expected() const { return expected_; }

 private:
  size_t got_, expected_;
};
class OutOfRangeException : public Exception {
 public:
  OutOfRangeException(const std::string& where, double x, double lo, double hi)
      : Exception(where + ": " + std::to_string(x) + " out of [" + std::to_string(lo) + ", " + std::to_string(hi) + "[") {}
};

// ------------------------------------------------------------------------------------------------ Domain
// CoMap/Domain.cpp:46-59 (equal-width bounds), :113-122 (half-open getIndex).
class Domain {
 public:
  Domain(double a, double b, size_t n) : bounds_(n + 1) {
    if (n == 0) throw Exception("Domain::constructor1. Number of classes should be > 0.");
    const double mini = std::min(a, b), maxi = std::max(a, b);
    const double w = (maxi - mini) / static_cast<double>(n);
    bounds_[0] = mini;
    for (size_t i = 1; i < n + 1; i++) bounds_[i] = mini + static_cast<double>(i) * w;
  }
  explicit Domain(const Vdouble& bounds) : bounds_(bounds) {
    for (size_t i = 0; i + 1 < bounds_.size(); i++)
      if (bounds_[i + 1] < bounds_[i]) throw Exception("Domain: bounds must be increasing.");
  }
  size_t getSize() const { return bounds_.size() - 1; }
  double getLowerBound() const { return bounds_.front(); }
  double getUpperBound() const { return bounds_.back(); }
  size_t getIndex(double x) const {
    if (x < getLowerBound() || x >= getUpperBound())
      throw OutOfRangeException("Domain::getIndex", x, getLowerBound(), getUpperBound());
    for (size_t i = 1; i < bounds_.size(); i++)
      if (x < bounds_[i]) return i - 1;
    throw Exception("Unexpected error!");
  }

 private:
  Vdouble bounds_;
};

// ------------------------------------------------------------------------------------------------ Engine
struct TreeArrays {  // nodes in post-order, root last (== row order of the reference's .vec files)
  std::vector<int32_t> parent;
  std::vector<double> branchLengths;
  std::vector<int32_t> leafOfTaxon;
};
struct ModelArrays {
  int nbStates = 0;
  std::vector<double> generator;    // Q, row-major
  std::vector<double> frequencies;  // pi
  std::vector<double> rates, rateProbabilities;
  std::vector<double> registers;    // K * S * S: Q o register_k o weights; empty => unweighted total count
  int nbTypes = 1;
  // non-homogeneous model set (leave empty for a homogeneous model): generators [M][S*S], their equilibrium
  // frequencies [M][S], registers [M][K][S*S] (optional), the generator of the branch above each node [nnodes] =
  // SubstitutionModelSet::getModelIndexForNode, and the root frequency set [S]
  std::vector<double> generators, generatorFrequencies, generatorRegisters, rootFrequencies;
  std::vector<int32_t> modelOfBranch;
  bool naive = false;               // nijt = Naive
  std::vector<double> naiveWeights;
  bool clampNegative = true;        // unweighted counts (Bio++ clamps negative round-off)
};

// Owns one cmx_ctx (one per GPU).  Stands in for the (drtl, substitutionCount, seqSim) triple of the reference.
class Engine {
 public:
  Engine(const TreeArrays& t, const ModelArrays& m, int device = 0) : S_(m.nbStates) {
    cmx_model cm{};   // zero: homogeneous unless the non-homogeneous fields are filled below
    cm.nstates = m.nbStates;
    cm.nclasses = static_cast<int32_t>(m.rates.size());
    cm.ntypes = m.registers.empty() ? 1 : m.nbTypes;
    cm.Q = m.generator.data();
    cm.pi = m.frequencies.data();
    cm.rates = m.rates.data();
    cm.probs = m.rateProbabilities.data();
    cm.Bk = m.registers.empty() ? nullptr : m.registers.data();
    cm.count_method = m.naive ? CMX_COUNT_NAIVE : CMX_COUNT_EXPECTED;
    cm.clamp_negative = m.clampNegative ? 1 : 0;
    cm.naive_weights = m.naiveWeights.empty() ? nullptr : m.naiveWeights.data();
    if (!m.modelOfBranch.empty()) {   // SubstitutionModelSet: one generator per branch (CoETools.cpp:126-206)
      cm.nmodels = static_cast<int32_t>(m.generators.size() / (static_cast<size_t>(m.nbStates) * m.nbStates));
      cm.Qs = m.generators.data();
      cm.pis = m.generatorFrequencies.data();
      cm.Bks = m.generatorRegisters.empty() ? nullptr : m.generatorRegisters.data();
      cm.model_of_branch = m.modelOfBranch.data();
      cm.root_freqs = m.rootFrequencies.data();
      cm.ntypes = m.generatorRegisters.empty() ? 1 : m.nbTypes;
    }
    cmx_tree ct;
    ct.nnodes = static_cast<int32_t>(t.parent.size());
    ct.parent = t.parent.data();
    ct.blen = t.branchLengths.data();
    ct.ntaxa = static_cast<int32_t>(t.leafOfTaxon.size());
    ct.leaf_of_taxon = t.leafOfTaxon.data();
    if (cmx_ctx_create(&cm, &ct, device, &ctx_) != CMX_OK) throw Exception(cmx_last_error(nullptr));
    nbBranches_ = static_cast<size_t>(ct.nnodes) - 1;
    nbTypes_ = static_cast<size_t>(cm.ntypes);
    nbTaxa_ = static_cast<size_t>(ct.ntaxa);
  }
  // model-free context: pair statistics / distances / Mica column MI only (cmx_ctx_create(NULL, NULL, ...))
  explicit Engine(int device = 0) : S_(0) {
    if (cmx_ctx_create(nullptr, nullptr, device, &ctx_) != CMX_OK) throw Exception(cmx_last_error(nullptr));
  }
  ~Engine() { cmx_ctx_destroy(ctx_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  // nijt.average / nijt.joint (CoETools.cpp:393-406; "for benchmarking only" there).  average = no, joint = yes maps with
  // computeSubstitutionVectorsNoAveraging from here on (observed data and nulls; what nijt = Label with the MI statistic
  // needs, CoETools.cpp:577-588); joint = no selects the two ...Marginal variants (CoETools.cpp:399-405).
  void setMappingOptions(bool average, bool joint) { check(cmx_set_mapping_options(ctx_, average ? 1 : 0, joint ? 1 : 0)); }
  // simulations.continuous = yes (CoMap.cpp:146, 213: seqSim->enableContinuousRates(true)): n simulated sites [taxon][site]
  // with global site indices g0 .. g0 + n - 1, every site with its own rate from the continuous Gamma(alpha, alpha)
  // (+ invariant mass pInvariant); the drawn rates are returned in *rates when given
  std::vector<uint8_t> simulateContinuous(uint64_t seed, uint64_t g0, size_t n, double gammaAlpha, double pInvariant = 0.,
                                          Vdouble* rates = nullptr) const {
    std::vector<uint8_t> aln(nbTaxa_ * n);
    if (rates) rates->assign(n, 0.);
    check(cmx_simulate_continuous(ctx_, seed, g0, n, gammaAlpha, pInvariant, aln.data(), rates ? rates->data() : nullptr));
    return aln;
  }
  cmx_ctx* ctx() const { return ctx_; }
  size_t getNumberOfBranches() const { return nbBranches_; }
  size_t getNumberOfSubstitutionTypes() const { return nbTypes_; }
  size_t getNumberOfTaxa() const { return nbTaxa_; }
  int getNumberOfStates() const { return S_; }
  void check(cmx_status s) const {
    if (s != CMX_OK) throw Exception(cmx_last_error(ctx_));
  }

 private:
  cmx_ctx* ctx_ = nullptr;
  int S_;
  size_t nbBranches_ = 0, nbTypes_ = 0, nbTaxa_ = 0;
};

// ------------------------------------------------------------------------------------------------ mapping
// LegacyProbabilisticSubstitutionMapping stand-in: mapping[i] is the VVdouble v[branch][type] of site i
// (CoMap/Statistics.h:154-160), operator()(branch, site, type) the scalar accessor (CoMap/ClusterTools.cpp:237).
class ProbabilisticSubstitutionMapping {
 public:
  ProbabilisticSubstitutionMapping(size_t nbSites, size_t nbBranches, size_t nbTypes)
      : n_(nbSites), b_(nbBranches), k_(nbTypes), counts_(nbSites * nbBranches * nbTypes) {}
  size_t getNumberOfSites() const { return n_; }
  size_t getNumberOfBranches() const { return b_; }
  size_t getNumberOfSubstitutionTypes() const { return k_; }
  double operator()(size_t branch, size_t site, size_t type) const { return counts_[(site * b_ + branch) * k_ + type]; }
  double& operator()(size_t branch, size_t site, size_t type) { return counts_[(site * b_ + branch) * k_ + type]; }
  VVdouble operator[](size_t site) const {
    VVdouble v(b_, Vdouble(k_));
    for (size_t b = 0; b < b_; ++b)
      for (size_t k = 0; k < k_; ++k) v[b][k] = (*this)(b, site, k);
    return v;
  }
  double* data() { return counts_.data(); }
  const double* data() const { return counts_.data(); }
  // filled by getVectors alongside the counts (the reference reads them back from the likelihood object,
  // CoMap/CoETools.cpp:507-510, 669-670)
  Vdouble logLikelihoods, posteriorRates, norms;
  std::vector<int32_t> rateClasses;

 private:
  size_t n_, b_, k_;
  Vdouble counts_;
};

// ------------------------------------------------------------------------------------------------ statistics
class Statistic {
 public:
  virtual ~Statistic() {}
  virtual int kind() const = 0;                       // cmx_stat_kind
  virtual const double* params() const { return nullptr; }
  // all-pairs form of getValueForPair: out[i * n + j] for j > i (NaN elsewhere), CoMap/CoETools.cpp:672-692
  Vdouble getValuesForAllPairs(const Engine& eng, const ProbabilisticSubstitutionMapping& mapping) const {
    const size_t n = mapping.getNumberOfSites();
    Vdouble out(n * n);
    eng.check(cmx_pair_stats(eng.ctx(), kind(), params(), mapping.data(), n, nullptr, 0, out.data()));
    return out;
  }
  // rectangular form, CoMap/CoETools.cpp:786-810
  Vdouble getValuesForAllPairs(const Engine& eng, const ProbabilisticSubstitutionMapping& m1,
                               const ProbabilisticSubstitutionMapping& m2) const {
    if (m1.getNumberOfBranches() != m2.getNumberOfBranches())
      throw DimensionException("Statistic::getValuesForAllPairs.", m2.getNumberOfBranches(), m1.getNumberOfBranches());
    Vdouble out(m1.getNumberOfSites() * m2.getNumberOfSites());
    eng.check(cmx_pair_stats(eng.ctx(), kind(), params(), m1.data(), m1.getNumberOfSites(), m2.data(),
                             m2.getNumberOfSites(), out.data()));
    return out;
  }
};
class CorrelationStatistic : public Statistic { public: int kind() const override { return CMX_STAT_CORRELATION; } };
class CovarianceStatistic : public Statistic { public: int kind() const override { return CMX_STAT_COVARIANCE; } };
class CosinusStatistic : public Statistic { public: int kind() const override { return CMX_STAT_COSINUS; } };
class CosubstitutionNumberStatistic : public Statistic { public: int kind() const override { return CMX_STAT_COSUBSTITUTION; } };
class CompensationStatistic : public Statistic { public: int kind() const override { return CMX_STAT_COMPENSATION; } };
// Statistics.h:296-329.  The reference constructs it from a bounds vector (DiscreteMutualInformationStatistic(const
// Vdouble& bounds)); its factory (CoETools.cpp:577-593) builds {0, threshold, 10000} for MI(threshold) and, for
// nijt = Label, -0.5, 0.5, .., S(S-1) + 0.5 (labelBounds).  The two-bound-interval form runs on the indicator Gram
// (CMX_STAT_DISCRETE_MI), any other vector on the joint-table kernel (CMX_STAT_DISCRETE_MI_BOUNDS).
class DiscreteMutualInformationStatistic : public Statistic {
 public:
  explicit DiscreteMutualInformationStatistic(double threshold = 0.99) : params_{3., 0., threshold, 10000.} {}
  explicit DiscreteMutualInformationStatistic(const Vdouble& bounds) {
    if (bounds.size() < 2) throw Exception("DiscreteMutualInformationStatistic: at least two bounds are needed.");
    for (size_t i = 0; i + 1 < bounds.size(); ++i)   // Domain::Domain(const Vdouble&), Domain.cpp:62-72
      if (bounds[i + 1] < bounds[i])
        throw Exception("Bound " + std::to_string(i + 1) + " (" + std::to_string(bounds[i + 1]) + ") is < to bound " +
                        std::to_string(i) + " (" + std::to_string(bounds[i]) + ").");
    params_.push_back(static_cast<double>(bounds.size()));
    params_.insert(params_.end(), bounds.begin(), bounds.end());
  }
  // CoETools.cpp:583-588: one unit bin per substitution label 0 .. S(S-1)
  static Vdouble labelBounds(size_t alphabetSize) {
    const size_t n = alphabetSize * (alphabetSize - 1);
    Vdouble b(n + 2);
    b[0] = -0.5;
    for (size_t i = 0; i < n + 1; ++i) b[i + 1] = b[i] + 1;
    return b;
  }
  bool isThresholdForm() const { return params_.size() == 4 && params_[1] == 0. && params_[3] == 10000.; }
  int kind() const override { return isThresholdForm() ? CMX_STAT_DISCRETE_MI : CMX_STAT_DISCRETE_MI_BOUNDS; }
  // threshold form: the C-ABI takes the threshold alone; else [nbounds, bounds..]
  const double* params() const override { return isThresholdForm() ? &params_[2] : params_.data(); }
  Vdouble getBounds() const { return Vdouble(params_.begin() + 1, params_.end()); }

 private:
  Vdouble params_;   // [nbounds, bounds..]
};

// Statistics.h:176-204: correlation of the vectors minus a per-branch mean vector; CoMap.cpp:350-359 sets it to the mean
// total substitution vector of the data (computeMeanVector below)
class CorrectedCorrelationStatistic : public Statistic {
 public:
  CorrectedCorrelationStatistic() {}
  explicit CorrectedCorrelationStatistic(const Vdouble& meanVector) { setMeanVector(meanVector); }
  CorrectedCorrelationStatistic(const Vdouble& meanVector1, const Vdouble& meanVector2) { setMeanVectors(meanVector1, meanVector2); }
  void setMeanVector(const Vdouble& meanVector) { setMeanVectors(meanVector, meanVector); }
  void setMeanVectors(const Vdouble& meanVector1, const Vdouble& meanVector2) {
    if (meanVector1.size() != meanVector2.size())
      throw DimensionException("CorrectedCorrelationStatistic::setMeanVectors.", meanVector2.size(), meanVector1.size());
    means_ = meanVector1;
    means_.insert(means_.end(), meanVector2.begin(), meanVector2.end());
  }
  int kind() const override { return CMX_STAT_CORRECTED_CORRELATION; }
  const double* params() const override { return means_.empty() ? nullptr : means_.data(); }
  // CoMap.cpp:350-359: mean over the sites of computeTotalSubstitutionVectorForSitePerBranch (sum over types)
  static Vdouble computeMeanVector(const ProbabilisticSubstitutionMapping& mapping) {
    Vdouble mv(mapping.getNumberOfBranches(), 0.);
    for (size_t i = 0; i < mapping.getNumberOfSites(); ++i)
      for (size_t b = 0; b < mv.size(); ++b)
        for (size_t k = 0; k < mapping.getNumberOfSubstitutionTypes(); ++k) mv[b] += mapping(b, i, k);
    for (double& v : mv) v /= static_cast<double>(mapping.getNumberOfSites());
    return mv;
  }

 private:
  Vdouble means_;   // [2][B]
};

// CoMap/Distance.h:316-370 (comp - stat) and :372-424 (1 - stat); matrix fill loops CoMap/CoMap.cpp:432-440
class StatisticBasedDistance {
 public:
  StatisticBasedDistance(std::shared_ptr<Statistic> stat, double comp) : stat_(stat), comp_(comp) {}
  // the distance as the device names it (CMX_DIST_*); only the distances CoMap.cpp:402-428 can construct exist there
  int distanceKind() const {
    if (stat_->kind() == CMX_STAT_CORRELATION && comp_ == 1.) return CMX_DIST_CORRELATION;
    if (stat_->kind() == CMX_STAT_COMPENSATION && comp_ == 1.) return CMX_DIST_COMPENSATION;
    throw Exception("StatisticBasedDistance: only 1 - Correlation and 1 - Compensation are clustering distances");
  }
  Vdouble getDistancesForAllPairs(const Engine& eng, const ProbabilisticSubstitutionMapping& mapping) const {
    Vdouble d = stat_->getValuesForAllPairs(eng, mapping);
    for (double& v : d) v = comp_ - v;
    return d;
  }

 private:
  std::shared_ptr<Statistic> stat_;
  double comp_;
};
// CoMap/Distance.h:150-173
class EuclidianDistance {
 public:
  int distanceKind() const { return CMX_DIST_EUCLIDIAN; }
  Vdouble getDistancesForAllPairs(const Engine& eng, const ProbabilisticSubstitutionMapping& mapping) const {
    const size_t n = mapping.getNumberOfSites();
    Vdouble out(n * n);
    eng.check(cmx_pair_stats(eng.ctx(), CMX_STAT_EUCLIDIAN_DISTANCE, nullptr, mapping.data(), n, nullptr, 0, out.data()));
    return out;
  }
};
class CompensationDistance : public StatisticBasedDistance {
 public:
  CompensationDistance() : StatisticBasedDistance(std::make_shared<CompensationStatistic>(), 1.) {}
};

// ------------------------------------------------------------------------------------------------ AnalysisTools
struct NullDistributionRow { double stat; int32_t rcMin; double prMin, nMin; };  // columns of AnalysisTools.cpp:642
// simulations.continuous = yes (CoMap.cpp:146, 213: seqSim->enableContinuousRates(true)): every simulated site draws its
// rate from the continuous Gamma(alpha, alpha) (+ an invariant mass) instead of the discrete classes
struct ContinuousRates { double gammaAlpha = 1.; double pInvariant = 0.; };

class AnalysisTools {
  static VVdouble vectorMatrix(const Engine& eng, int kind, const VVdouble& v1, const VVdouble* v2, bool independant, const char* where) {
    const size_t n1 = v1.size(), n2 = v2 ? v2->size() : n1;
    if (independant && n1 != n2)
      throw DimensionException(std::string(where) + "\nWhen performing independant comparisons, the two datasets must have the same length.", n1, n2);
    VVdouble matrix(n1, Vdouble(n2, 0.0));
    if (n1 == 0 || n2 == 0) return matrix;
    const size_t dim = v1[0].size();
    auto flat = [dim](const VVdouble& v) {
      Vdouble f(v.size() * dim);
      for (size_t i = 0; i < v.size(); ++i) {
        if (v[i].size() != dim) throw DimensionException("AnalysisTools: vectors of different lengths.", v[i].size(), dim);
        std::copy(v[i].begin(), v[i].end(), f.begin() + i * dim);
      }
      return f;
    };
    const Vdouble f1 = flat(v1), f2 = v2 ? flat(*v2) : Vdouble();
    Vdouble out(n1 * n2);
    eng.check(cmx_vector_matrix(eng.ctx(), kind, dim, f1.data(), n1, v2 ? f2.data() : nullptr, n2, independant ? 1 : 0, out.data()));
    for (size_t i = 0; i < n1; ++i) std::copy(out.begin() + i * n2, out.begin() + (i + 1) * n2, matrix[i].begin());
    return matrix;
  }

 public:
  // AnalysisTools.cpp:343-350 (the norms come out of the mapping kernel; recomputed here from the counts on request)
  static Vdouble computeNorms(const ProbabilisticSubstitutionMapping& mapping) {
    if (!mapping.norms.empty()) return mapping.norms;
    Vdouble v(mapping.getNumberOfSites());
    for (size_t i = 0; i < v.size(); ++i) {
      double s = 0;
      for (size_t b = 0; b < mapping.getNumberOfBranches(); ++b) {
        double t = 0;
        for (size_t k = 0; k < mapping.getNumberOfSubstitutionTypes(); ++k) t += mapping(b, i, k);
        s += t * t;
      }
      v[i] = std::sqrt(s);
    }
    return v;
  }
  // ---- AnalysisTools::compute{ScalarProduct,Cosinus,Correlation,Covariance}Matrix (AnalysisTools.h:93-190,
  // AnalysisTools.cpp:102-339): the matrix of a pairwise function of plain vectors, on the device's Gram kernel
  // (cmx_vector_matrix).  One-set forms: symmetric, diagonal scalar(v, v) / 1 / 1 / var(v).  Two-set forms with
  // independantComparisons: the two sets must have the same length (DimensionException as in the reference) and only
  // matrix[i][i] is computed.  `eng`: any engine (a model-less one serves: nothing here depends on tree or model).
  static VVdouble computeScalarProductMatrix(const Engine& eng, const VVdouble& vectors) { return vectorMatrix(eng, CMX_STAT_SCALAR_PRODUCT, vectors, nullptr, false, ""); }
  static VVdouble computeScalarProductMatrix(const Engine& eng, const VVdouble& vectors1, const VVdouble& vectors2, bool independantComparisons) {
    return vectorMatrix(eng, CMX_STAT_SCALAR_PRODUCT, vectors1, &vectors2, independantComparisons, "AnalysisTools::computeScalarProductMatrix.");
  }
  static VVdouble computeCosinusMatrix(const Engine& eng, const VVdouble& vectors) { return vectorMatrix(eng, CMX_STAT_COSINUS, vectors, nullptr, false, ""); }
  static VVdouble computeCosinusMatrix(const Engine& eng, const VVdouble& vectors1, const VVdouble& vectors2, bool independantComparisons) {
    return vectorMatrix(eng, CMX_STAT_COSINUS, vectors1, &vectors2, independantComparisons, "AnalysisTools::computeCosinusMatrix.");
  }
  static VVdouble computeCorrelationMatrix(const Engine& eng, const VVdouble& vectors) { return vectorMatrix(eng, CMX_STAT_CORRELATION, vectors, nullptr, false, ""); }
  static VVdouble computeCorrelationMatrix(const Engine& eng, const VVdouble& vectors1, const VVdouble& vectors2, bool independantComparisons) {
    return vectorMatrix(eng, CMX_STAT_CORRELATION, vectors1, &vectors2, independantComparisons, "AnalysisTools::computeCorrelationMatrix.");
  }
  static VVdouble computeCovarianceMatrix(const Engine& eng, const VVdouble& vectors) { return vectorMatrix(eng, CMX_STAT_COVARIANCE, vectors, nullptr, false, ""); }
  static VVdouble computeCovarianceMatrix(const Engine& eng, const VVdouble& vectors1, const VVdouble& vectors2, bool independantComparisons) {
    return vectorMatrix(eng, CMX_STAT_COVARIANCE, vectors1, &vectors2, independantComparisons, "AnalysisTools::computeCovarianceMatrix.");
  }

  // AnalysisTools.cpp:564-658.  simstats (optional, one vector per class of rateDomain) receives the statistics as
  // in the reference (pairs whose min norm is out of the domain are dropped, :645-648); rows receives all of them.
  static void getNullDistributionIntraDR(const Engine& eng, const Statistic& statistic, uint64_t seed, size_t repCPU,
                                         size_t repRAM, std::vector<NullDistributionRow>* rows,
                                         std::vector<std::vector<double>>* simstats, const Domain* rateDomain,
                                         size_t repBegin = 0, const ContinuousRates* continuous = nullptr) {
    if (simstats && rateDomain && rateDomain->getSize() != simstats->size())
      throw Exception("AnalysisTools::getNullDistributionIntraDR. Input vector should be of same size as rate domain.");
    if (simstats && !rateDomain && simstats->size() != 1)
      throw Exception("AnalysisTools::getNullDistributionIntraDR. Input vector should be of same size 1 as no rate domain was specified.");
    const size_t n = repCPU * repRAM;
    Vdouble stat(n), pr(n), nm(n);
    std::vector<int32_t> rc(n);
    if (continuous)   // simulator and mapping both on the device, the alignments never leave it
      eng.check(cmx_null_intra_continuous(eng.ctx(), statistic.kind(), statistic.params(), seed, repBegin, repBegin + repCPU, repRAM,
                                          continuous->gammaAlpha, continuous->pInvariant, stat.data(), rc.data(), pr.data(), nm.data()));
    else
      eng.check(cmx_null_intra(eng.ctx(), statistic.kind(), statistic.params(), seed, repBegin, repBegin + repCPU, repRAM,
                               nullptr, stat.data(), rc.data(), pr.data(), nm.data()));
    for (size_t q = 0; q < n; ++q) {
      if (rows) rows->push_back({stat[q], rc[q], pr[q], nm[q]});
      if (simstats) {
        if (rateDomain) {
          try {
            (*simstats)[rateDomain->getIndex(nm[q])].push_back(stat[q]);
          } catch (OutOfRangeException&) {
          }
        } else {
          (*simstats)[0].push_back(stat[q]);
        }
      }
    }
  }
  // AnalysisTools.cpp:662-735: eng1 / eng2 hold the two data sets (own model and branch lengths, same branches);
  // rows = the lines of statistics.null.txt (Stat, RCmin, PRmin, Nmin).
  static void getNullDistributionInterDR(const Engine& eng1, const Engine& eng2, const Statistic& statistic,
                                         uint64_t seed, size_t repCPU, size_t repRAM,
                                         std::vector<NullDistributionRow>* rows, size_t repBegin = 0) {
    const size_t n = repCPU * repRAM;
    Vdouble stat(n), pr(n), nm(n);
    std::vector<int32_t> rc(n);
    eng1.check(cmx_null_inter(eng1.ctx(), eng2.ctx(), statistic.kind(), statistic.params(), seed, repBegin,
                              repBegin + repCPU, repRAM, stat.data(), rc.data(), pr.data(), nm.data()));
    if (rows)
      for (size_t q = 0; q < n; ++q) rows->push_back({stat[q], rc[q], pr[q], nm[q]});
  }
};

// ------------------------------------------------------------------------------------------------ CoETools
struct IntraStatRow {  // one line of statistics.txt, CoETools.cpp:662-722
  size_t i, j;
  double stat;
  int32_t rcMin;
  double prMin, nMin, pValue;  // pValue NaN == "NA"
  int32_t nSim;
};
struct PairFilters {  // CoETools.cpp:420-481
  int minRateClass = 0, maxRateClassDiff = -1;
  double minRate = 0., maxRateDiff = -1., minStatistic = 0.;
};

class CoETools {
 public:
  // CoETools.cpp:366-416: aln[t * nbSites + i] are state codes (>= nbStates: index into masks)
  static std::unique_ptr<ProbabilisticSubstitutionMapping> getVectors(const Engine& eng, const uint8_t* aln,
                                                                      size_t nbSites, const uint32_t* masks = nullptr,
                                                                      size_t nbMasks = 0) {
    std::unique_ptr<ProbabilisticSubstitutionMapping> m(new ProbabilisticSubstitutionMapping(
        nbSites, eng.getNumberOfBranches(), eng.getNumberOfSubstitutionTypes()));
    m->logLikelihoods.resize(nbSites);
    m->posteriorRates.resize(nbSites);
    m->norms.resize(nbSites);
    m->rateClasses.resize(nbSites);
    eng.check(cmx_map_sites(eng.ctx(), aln, nbSites, nbSites, masks, nbMasks, m->data(), m->logLikelihoods.data(),
                            m->posteriorRates.data(), m->rateClasses.data(), m->norms.data()));
    return m;
  }

  // CoETools.cpp:604-728 with the null of :836-872; returns the rows the reference would write, in its (i, j) order.
  static std::vector<IntraStatRow> computeIntraStats(const Engine& eng, const ProbabilisticSubstitutionMapping& mapping,
                                                     const Statistic& statistic, bool computeNull, uint64_t seed,
                                                     size_t nbRepCPU = 100, size_t nbRepRAM = 1000,
                                                     size_t nbRateClasses = 10, const PairFilters& f = PairFilters(),
                                                     const ContinuousRates* continuous = nullptr) {
    const size_t n = mapping.getNumberOfSites();
    const Vdouble norms = AnalysisTools::computeNorms(mapping);
    Vdouble ns, nm;
    if (computeNull) {
      const size_t nn = nbRepCPU * nbRepRAM;
      ns.resize(nn);
      nm.resize(nn);
      if (continuous)   // simulations.continuous = yes
        eng.check(cmx_null_intra_continuous(eng.ctx(), statistic.kind(), statistic.params(), seed, 0, nbRepCPU, nbRepRAM,
                                            continuous->gammaAlpha, continuous->pInvariant, ns.data(), nullptr, nullptr, nm.data()));
      else
        eng.check(cmx_null_intra(eng.ctx(), statistic.kind(), statistic.params(), seed, 0, nbRepCPU, nbRepRAM, nullptr,
                                 ns.data(), nullptr, nullptr, nm.data()));
    }
    // statistic, p-values, filters and the (i, j) ordering all happen on the device (cmx_intra_rows): only the rows
    // that the reference would write come back
    cmx_pair_filters pf;
    pf.min_rate_class = f.minRateClass; pf.max_rate_class_diff = f.maxRateClassDiff;
    pf.min_rate = f.minRate; pf.max_rate_diff = f.maxRateDiff; pf.min_statistic = f.minStatistic;
    std::vector<cmx_pair_row> raw(n * (n - 1) / 2 + 1);
    uint64_t count = 0;
    eng.check(cmx_intra_rows(eng.ctx(), statistic.kind(), statistic.params(), mapping.data(), n, mapping.rateClasses.data(),
                             mapping.posteriorRates.data(), norms.data(), computeNull ? ns.data() : nullptr,
                             computeNull ? nm.data() : nullptr, ns.size(), static_cast<int>(nbRateClasses), &pf, raw.data(),
                             raw.size(), &count));
    std::vector<IntraStatRow> rows(static_cast<size_t>(count));
    for (size_t q = 0; q < rows.size(); ++q) {
      IntraStatRow& r = rows[q];
      r.i = static_cast<size_t>(raw[q].i); r.j = static_cast<size_t>(raw[q].j); r.stat = raw[q].stat;
      r.rcMin = raw[q].rc_min; r.prMin = raw[q].pr_min; r.nMin = raw[q].n_min;
      r.pValue = raw[q].pvalue; r.nSim = raw[q].nsim;
    }
    return rows;
  }

  // CoETools.cpp:732-832 (no p-values there: the reference leaves them to computePValues.R): statistic, filters and the
  // compaction of the surviving pairs run on the device (cmx_inter_rows), only the rows come back.  Nmin is
  // min(norm1[i], norm2[j]); referenceNormQuirk = true reproduces the reference's own column, which reads norms2[i]
  // (CoETools.cpp:803).
  static std::vector<IntraStatRow> computeInterStats(const Engine& eng, const ProbabilisticSubstitutionMapping& mapping1,
                                                     const ProbabilisticSubstitutionMapping& mapping2,
                                                     const Statistic& statistic, bool independentComparisons = false,
                                                     const PairFilters& f1 = PairFilters(),
                                                     const PairFilters& f2 = PairFilters(), bool referenceNormQuirk = false) {
    const size_t n1 = mapping1.getNumberOfSites(), n2 = mapping2.getNumberOfSites();
    if (independentComparisons && n1 != n2)
      throw Exception("When performing independant comparisons, the two datasets must have the same length.");
    if (mapping1.getNumberOfBranches() != mapping2.getNumberOfBranches())
      throw DimensionException("CoETools::computeInterStats.", mapping2.getNumberOfBranches(), mapping1.getNumberOfBranches());
    const Vdouble norms1 = AnalysisTools::computeNorms(mapping1), norms2 = AnalysisTools::computeNorms(mapping2);
    cmx_inter_filters f;
    f.min_rate_class1 = f1.minRateClass; f.min_rate_class2 = f2.minRateClass;
    f.max_rate_class_diff = f1.maxRateClassDiff;
    f.independent_comparisons = independentComparisons ? 1 : 0;
    f.min_rate1 = f1.minRate; f.min_rate2 = f2.minRate;
    f.max_rate_diff = f1.maxRateDiff;
    f.min_statistic = f1.minStatistic;
    f.reference_norm_quirk = referenceNormQuirk ? 1 : 0;
    f.reserved = 0;
    const size_t cap = independentComparisons ? n1 : n1 * n2;
    std::vector<cmx_pair_row> raw(cap ? cap : 1);
    uint64_t count = 0;
    eng.check(cmx_inter_rows(eng.ctx(), statistic.kind(), statistic.params(), mapping1.data(), n1, mapping1.rateClasses.data(),
                             mapping1.posteriorRates.data(), norms1.data(), mapping2.data(), n2, mapping2.rateClasses.data(),
                             mapping2.posteriorRates.data(), norms2.data(), &f, raw.data(), cap, &count));
    std::vector<IntraStatRow> rows(count);
    for (size_t q = 0; q < count; ++q) {
      const cmx_pair_row& r = raw[q];
      rows[q].i = (size_t)r.i; rows[q].j = (size_t)r.j; rows[q].stat = r.stat;
      rows[q].rcMin = r.rc_min; rows[q].prMin = r.pr_min; rows[q].nMin = r.n_min;
      rows[q].pValue = std::numeric_limits<double>::quiet_NaN();
      rows[q].nSim = 0;
    }
    return rows;
  }

  // CoETools.cpp:1042-1087 (defined after CandidateGroupSet below).  seed replaces the global RandomTools state;
  // repRAM = candidates.null.nb_rep_RAM, maxTrials = candidates.nb_max_trials.
  static void computePValuesForCandidateGroups(class CandidateGroupSet& candidates, const Engine& eng, uint64_t seed,
                                               unsigned int repRAM, unsigned int maxTrials);
};

// ------------------------------------------------------------------------------------------------ text outputs
// The files either side of the path, written with default ostream formatting exactly as the reference does.
// ------------------------------------------------------------------------------------------------ candidate groups
// CoMap/CoETools.h:71-137: a candidate site is a position of the mapping with the norm window its simulated stand-ins
// must fall into; a candidate group carries its observed statistic.
class CandidateSite {
 public:
  explicit CandidateSite(size_t index) : index_(index) {}
  void setNormRange(double min, double max) { normMin_ = min; normMax_ = max; }
  bool checkNorm(double norm) const { return norm >= normMin_ && norm <= normMax_; }
  size_t getIndex() const { return index_; }
  double getNormMin() const { return normMin_; }
  double getNormMax() const { return normMax_; }

 private:
  size_t index_;
  double normMin_ = 0, normMax_ = 0;
};

class CandidateGroup {
 public:
  double getStatisticValue() const { return statistic_; }
  void setStatisticValue(double v) { statistic_ = v; }
  // Statistic::getValueForGroup on the device (CoETools.h:106-117)
  void computeStatisticValue(const Engine& eng, const Statistic& stat, const ProbabilisticSubstitutionMapping& mapping) {
    if (!analysable_) throw Exception("CandidateGroup::computeStatisticValue. Group is not analysable.");
    if (sites_.empty()) throw Exception("CandidateGroup::computeStatisticValue. Group is empty!");
    std::vector<int32_t> idx(sites_.size());
    for (size_t i = 0; i < sites_.size(); ++i) idx[i] = static_cast<int32_t>(sites_[i].getIndex());
    const int64_t off[2] = {0, static_cast<int64_t>(idx.size())};
    eng.check(cmx_group_stats(eng.ctx(), stat.kind(), stat.params(), mapping.data(), mapping.getNumberOfSites(), off, idx.data(), 1,
                              &statistic_));
  }
  // CoETools.h:118-128
  void computeNormRanges(double omega, const ProbabilisticSubstitutionMapping& mapping) {
    if (!analysable_) throw Exception("CandidateGroup::computeNormRanges. Group is not analyzable.");
    if (sites_.empty()) throw Exception("CandidateGroup::computeNormRanges. Group is empty!");
    const Vdouble norms = AnalysisTools::computeNorms(mapping);
    for (CandidateSite& s : sites_) s.setNormRange(norms[s.getIndex()] - omega, norms[s.getIndex()] + omega);
  }
  void setAnalysable(bool yn) { analysable_ = yn; }
  bool isAnalysable() const { return analysable_; }
  size_t size() const { return sites_.size(); }
  const CandidateSite& operator[](size_t i) const { return sites_[i]; }
  CandidateSite& operator[](size_t i) { return sites_[i]; }
  void addSite(const CandidateSite& cs) { sites_.push_back(cs); }

 private:
  std::vector<CandidateSite> sites_;
  double statistic_ = 0;
  bool analysable_ = true;
};

// CoETools.h:139-300.  The simulation bookkeeping lives behind cmx_candidate_groups; this class holds the candidates
// and, afterwards, the counts.
class CandidateGroupSet {
 public:
  CandidateGroupSet(const Statistic* statistic, unsigned int minSim) : statistic_(statistic), minSim_(minSim) {}
  void addCandidate(const CandidateGroup& g) { candidates_.push_back(g); n1_.push_back(0); n2_.push_back(0); }
  double getPValueForGroup(size_t g) const { return (static_cast<double>(n1_[g]) + 1.) / (static_cast<double>(n2_[g]) + 1.); }
  double getN1ForGroup(size_t g) const { return n1_[g]; }
  double getN2ForGroup(size_t g) const { return n2_[g]; }
  size_t size() const { return candidates_.size(); }
  const CandidateGroup& operator[](size_t g) const { return candidates_[g]; }
  unsigned int getNumberOfTrials() const { return nbTrials_; }
  uint64_t getNumberOfBatches() const { return nbBatches_; }

 private:
  friend class CoETools;
  const Statistic* statistic_;
  unsigned int minSim_;
  std::vector<CandidateGroup> candidates_;
  std::vector<uint32_t> n1_, n2_;
  unsigned int nbTrials_ = 0;
  uint64_t nbBatches_ = 0;
};

inline void CoETools::computePValuesForCandidateGroups(CandidateGroupSet& candidates, const Engine& eng, uint64_t seed,
                                                       unsigned int repRAM, unsigned int maxTrials) {
  const size_t G = candidates.size();
  std::vector<int64_t> off(G + 1, 0);
  Vdouble lo, hi, observed(G);
  std::vector<uint8_t> ok(G);
  for (size_t g = 0; g < G; ++g) {
    const CandidateGroup& c = candidates[g];
    for (size_t i = 0; i < c.size(); ++i) { lo.push_back(c[i].getNormMin()); hi.push_back(c[i].getNormMax()); }
    off[g + 1] = static_cast<int64_t>(lo.size());
    observed[g] = c.getStatisticValue();
    ok[g] = c.isAnalysable() ? 1 : 0;
  }
  uint32_t trials = 0;
  eng.check(cmx_candidate_groups(eng.ctx(), candidates.statistic_->kind(), candidates.statistic_->params(), G, off.data(),
                                 lo.data(), hi.data(), ok.data(), observed.data(), candidates.minSim_, repRAM, maxTrials, 0, seed,
                                 candidates.n1_.data(), candidates.n2_.data(), &trials, &candidates.nbBatches_));
  candidates.nbTrials_ = trials;
}

// ------------------------------------------------------------------------------------------------ clustering
// bpp::HierarchicalClustering's method names as CoMap.cpp:460-472 passes them
struct HierarchicalClustering {
  static constexpr int COMPLETE = CMX_LINK_COMPLETE, SINGLE = CMX_LINK_SINGLE, AVERAGE = CMX_LINK_AVERAGE;
};

// The clustering tree with the node properties the reference attaches to it ("Stat", "Nmin"; Dmax = 2 * height):
// leaves 0..n-1, join m creates node n+m with sons merge[2m], merge[2m+1] (include/comap_mi355x.h).
struct ClusteringTree {
  size_t n = 0;
  std::vector<int32_t> merge, size;
  Vdouble dmax, stat, nmin;
  Vdouble distances;   // [n][n], only if asked for (clustering.output.matrix.file, CoMap.cpp:442-449)
};

// CoMap/ClusterTools.h:57-143 (the parts its callers read)
struct Group {
  std::vector<size_t> sites;   // positions in the mapping, in son order
  size_t join = 0;             // index of the inner node
  double dmax = 0, stat = 0, nmin = 0;
  size_t size() const { return sites.size(); }
  double getHeight() const { return dmax / 2.; }
  std::string toString() const {
    std::string t = "[";
    for (size_t i = 0; i < sites.size(); ++i) t += (i ? ";" : "") + std::to_string(sites[i]);
    return t + "]";
  }
  std::string toString(const std::vector<std::string>& names) const {
    std::string t = "[";
    for (size_t i = 0; i < sites.size(); ++i) t += (i ? ";" : "") + names[sites[i]];
    return t + "]";
  }
};

class ClusterTools {
 public:
  // distance matrix + clustering + node properties of the observed mapping: CoMap.cpp:432-491
  template <class DistanceT>
  static ClusteringTree cluster(const Engine& eng, const DistanceT& distance, int method,
                                const ProbabilisticSubstitutionMapping& mapping, bool wantMatrix = false) {
    ClusteringTree t;
    t.n = mapping.getNumberOfSites();
    if (t.n < 2) throw Exception("ClusterTools::cluster: at least two sites are needed");
    t.merge.resize(2 * (t.n - 1)); t.size.resize(t.n - 1);
    t.dmax.resize(t.n - 1); t.stat.resize(t.n - 1); t.nmin.resize(t.n - 1);
    if (wantMatrix) t.distances.resize(t.n * t.n);
    eng.check(cmx_cluster_sites(eng.ctx(), distance.distanceKind(), method, mapping.data(), t.n,
                                wantMatrix ? t.distances.data() : nullptr, t.merge.data(), t.dmax.data(), t.size.data(),
                                t.stat.data(), t.nmin.data()));
    return t;
  }

  // ClusterTools::getGroups (ClusterTools.cpp:55-113): one group per inner node, sons first; members in son order
  static std::vector<Group> getGroups(size_t n, const int32_t* merge, const double* dmax, const double* stat,
                                      const double* nmin) {
    std::vector<Group> groups;
    if (n < 2) return groups;
    std::vector<std::vector<size_t>> members(2 * n - 1);
    for (size_t i = 0; i < n; ++i) members[i] = {i};
    // joins are already in an order where sons precede their parent; the reference's order is the post-order walk
    std::vector<std::pair<size_t, bool>> stack{{2 * n - 2, false}};
    while (!stack.empty()) {
      auto [node, seen] = stack.back();
      stack.pop_back();
      if (node < n) continue;
      const size_t a = static_cast<size_t>(merge[2 * (node - n)]), b = static_cast<size_t>(merge[2 * (node - n) + 1]);
      if (!seen) {
        stack.push_back({node, true});
        stack.push_back({b, false});
        stack.push_back({a, false});
      } else {
        members[node] = members[a];
        members[node].insert(members[node].end(), members[b].begin(), members[b].end());
        if (a >= n) std::vector<size_t>().swap(members[a]);
        if (b >= n) std::vector<size_t>().swap(members[b]);
        Group g;
        g.sites = members[node];
        g.join = node - n;
        g.dmax = dmax[g.join]; g.stat = stat[g.join]; g.nmin = nmin[g.join];
        groups.push_back(std::move(g));
      }
    }
    return groups;
  }
  static std::vector<Group> getGroups(const ClusteringTree& t) {
    return getGroups(t.n, t.merge.data(), t.dmax.data(), t.stat.data(), t.nmin.data());
  }

  // ClusterTools::computeGlobalDistanceDistribution (ClusterTools.cpp:200-294): the rows of the null file go to *out
  // (may be null).  All replicates are simulated, mapped and clustered on the device; seed replaces the reference's
  // global RandomTools state.
  template <class DistanceT>
  static void computeGlobalDistanceDistribution(const Engine& eng, const DistanceT& distance, int method, uint64_t seed,
                                                size_t sizeOfDataSet, size_t nrep, size_t maxGroupSize,
                                                std::ostream* out) {
    if (sizeOfDataSet < 2 || nrep == 0) throw Exception("computeGlobalDistanceDistribution: nothing to do");
    const size_t nm = sizeOfDataSet - 1;
    std::vector<int32_t> merge(2 * nm * nrep), size(nm * nrep);
    Vdouble dmax(nm * nrep), stat(nm * nrep), nmin(nm * nrep);
    eng.check(cmx_cluster_null(eng.ctx(), distance.distanceKind(), method, seed, 0, nrep, sizeOfDataSet, merge.data(),
                               dmax.data(), size.data(), stat.data(), nmin.data()));
    if (!out) return;
    *out << "Rep\tGroup\tSize\tDmax\tStat\tNmin" << std::endl;
    for (size_t k = 0; k < nrep; ++k)
      for (const Group& g : getGroups(sizeOfDataSet, &merge[2 * nm * k], &dmax[nm * k], &stat[nm * k], &nmin[nm * k])) {
        if (g.size() > maxGroupSize) continue;
        *out << k << "\t" << g.toString() << "\t" << g.size() << "\t" << g.dmax << "\t" << g.stat << "\t" << g.nmin
             << std::endl;
      }
  }
};

// ------------------------------------------------------------------------------------------------ Mica
// CoMap/Mica.cpp on top of the C-ABI: all-pairs MI / joint entropy / entropies (one-hot Gram on the matrix cores),
// averageMI -> APC / RCW (:346-363, :656-657), the four null methods (:399-632) and the p-value rule of :671-683.
// The alignment is [taxon][site] state codes; codes >= alphabetSize index `masks` (bit a = compatible with state a),
// as SiteTools::*(.., resolveUnknowns = true) resolves them.  An Engine WITH a model is Mica's `use_model` case: the
// parametric bootstrap needs it, and norms (from CoETools::getVectors on the same engine) then bin the p-values
// (:383-386) and add the Nmin column.
class Mica {
 public:
  struct Options {
    std::string nullMethod = "none";      // none | nonparametric-bootstrap | parametric-bootstrap | z-score | permutations (:370)
    size_t nbRepCPU = 10, nbRepRAM = 100; // null.nb_rep_CPU / null.nb_rep_RAM (:415-416, :499-500)
    size_t nbRateClasses = 10;            // null.nb_rate_classes (:390)
    bool computePValues = true;           // null.compute_pvalues (:387; forced on for z-score, off for permutations)
    std::string zScoreStat = "MIp";       // null.method_zscore.stat (:551): MI | MIp | MIc
    size_t maxNbPermutations = 1000;      // null.max_number_of_permutations (:610)
    bool continuousSim = false;           // simulations.continuous (:473), with the Gamma shape / invariant mass below
    ContinuousRates continuous;
    uint64_t seed = 0;                    // replaces Bio++'s global generator
  };
  struct NullRow { double mi, hjoint, hmin, nmin; };               // lines of null.output.file (:411-415, :494)
  struct Row {                                                     // lines of output.file (:646-689)
    size_t i, j;
    double mi, apc, rcw, hjoint, hmin, nmin;
    double permPValue; int32_t permNb;                             // Perm.p.value / Perm.nb
    double bsPValue; int32_t bsNb;                                 // Bs.p.value (NaN == "NA") / Bs.nb
  };
  struct Result {
    bool withModel = false, withPermutations = false, withPValues = false;
    Vdouble entropy, averageMI;
    double fullAverageMI = 0.;
    std::vector<Row> rows;
    std::vector<NullRow> null;                                     // bootstrap nulls only (the z-score null is the data itself)
  };

  // miTest of every pair (Mica.cpp:93-118): shuffles until 5 of them reach the observed MI or maxNbPermutations were done.
  // pvalue / nbPermutations: [n (n - 1) / 2] in the reference's (i, j) order
  static void miTest(const Engine& eng, const uint8_t* aln, size_t nbTaxa, size_t nbSites, int alphabetSize, const uint32_t* masks,
                     size_t nbMasks, size_t maxNbPermutations, uint64_t seed, Vdouble& pvalue, std::vector<int32_t>& nbPermutations) {
    const size_t np = nbSites * (nbSites - 1) / 2;
    pvalue.assign(np, 0.);
    nbPermutations.assign(np, 0);
    eng.check(cmx_mica_permutation_test_masks(eng.ctx(), alphabetSize, (int)nbTaxa, masks, nbMasks, aln, nbSites,
                                              (uint32_t)maxNbPermutations, seed, pvalue.data(), nbPermutations.data()));
  }

  static Result analyse(const Engine& eng, const uint8_t* aln, size_t nbTaxa, size_t nbSites, int alphabetSize,
                        const uint32_t* masks, size_t nbMasks, const Vdouble* norms, const Options& opt) {
    if (nbSites < 2) throw Exception("Mica: at least two sites are needed.");
    const size_t n = nbSites;
    Result res;
    res.withModel = norms != nullptr;
    if (norms && norms->size() != n) throw DimensionException("Mica::analyse: norms.", norms->size(), n);
    // ---- all pairs at once (the reference computes every MI twice, :349-361 and :658)
    Vdouble mi(n * n), hj(n * n);
    res.entropy.assign(n, 0.);
    eng.check(cmx_mi_columns(eng.ctx(), alphabetSize, (int)nbTaxa, masks, nbMasks, aln, n, nullptr, 0, mi.data(), hj.data(),
                             res.entropy.data(), nullptr));
    res.averageMI.assign(n, 0.);
    eng.check(cmx_mica_average_mi(eng.ctx(), mi.data(), n, res.averageMI.data(), &res.fullAverageMI));
    const Vdouble& key = norms ? *norms : res.entropy;             // what the p-values are binned on (:383-386)
    // ---- null distribution
    Vdouble nullStat, nullKey;
    bool computePValues = false;
    if (opt.nullMethod != "none") {
      if (opt.nullMethod == "z-score") computePValues = true;
      else if (opt.nullMethod == "permutations") computePValues = false;
      else computePValues = opt.computePValues;
      if (opt.nullMethod == "nonparametric-bootstrap") {
        const size_t m = opt.nbRepCPU * opt.nbRepRAM;
        std::vector<int64_t> i1(m), i2(m);
        eng.check(cmx_mica_bootstrap_indices(opt.seed, n, opt.nbRepCPU, opt.nbRepRAM, i1.data(), i2.data()));
        Vdouble bmi(m), bhj(m);
        eng.check(cmx_mi_pairs(eng.ctx(), alphabetSize, (int)nbTaxa, masks, nbMasks, aln, n, nullptr, 0, i1.data(), i2.data(), m,
                               bmi.data(), bhj.data()));
        for (size_t q = 0; q < m; ++q) {
          const double hm = std::min(res.entropy[i1[q]], res.entropy[i2[q]]);
          const double nm = norms ? std::min((*norms)[i1[q]], (*norms)[i2[q]]) : std::numeric_limits<double>::quiet_NaN();
          res.null.push_back({bmi[q], bhj[q], hm, nm});
          nullStat.push_back(bmi[q]);
          nullKey.push_back(norms ? nm : hm);
        }
      } else if (opt.nullMethod == "parametric-bootstrap") {
        if (!norms) throw Exception("You need to specify a model of sequence evolution in order to use a parametric bootstrap approach!");
        const size_t m = opt.nbRepCPU * opt.nbRepRAM;
        Vdouble bmi(m), bhj(m), bnm(m);
        eng.check(cmx_mica_parametric_null(eng.ctx(), alphabetSize, opt.seed, opt.nbRepCPU, opt.nbRepRAM,
                                           opt.continuousSim ? opt.continuous.gammaAlpha : 0., opt.continuous.pInvariant, bmi.data(),
                                           bhj.data(), bnm.data()));
        for (size_t q = 0; q < m; ++q) {
          // (Hmin of this file is min(entropy[i], entropy[j]) with the REPLICATE and site counters in the reference,
          // :528 -- meaningless; reproduced where the indices exist, NaN beyond)
          const size_t rep = q / opt.nbRepRAM, j = q % opt.nbRepRAM;
          const double hm = (rep < n && j < n) ? std::min(res.entropy[rep], res.entropy[j]) : std::numeric_limits<double>::quiet_NaN();
          res.null.push_back({bmi[q], bhj[q], hm, bnm[q]});
        }
        nullStat = bmi;
        nullKey = bnm;
      } else if (opt.nullMethod == "z-score") {
        int which;
        if (opt.zScoreStat == "MI") which = CMX_MICA_MI;
        else if (opt.zScoreStat == "MIp") which = CMX_MICA_MIP;
        else if (opt.zScoreStat == "MIc") which = CMX_MICA_MIC;
        else throw Exception("Unkown statistic, should be 'MI', 'MIp' or 'MIc'.");
        nullStat.assign(n * (n - 1) / 2, 0.);
        nullKey.assign(n * (n - 1) / 2, 0.);
        eng.check(cmx_mica_zscore_null(eng.ctx(), which, mi.data(), n, key.data(), nullStat.data(), nullKey.data()));
      } else if (opt.nullMethod == "permutations") {
        if (opt.maxNbPermutations == 0) throw Exception("Permutation number should be greater than 0!");
      } else {
        throw Exception("Unvalid null distribution method specified: " + opt.nullMethod);
      }
    }
    // ---- p-values of :671-683 (same Domain(0, max key, nbRateClasses) binning and count as CoMap's; NaN == "NA")
    Vdouble pv;
    std::vector<int32_t> nsim;
    if (computePValues) {
      pv.assign(n * n, 0.);
      nsim.assign(n * n, 0);
      eng.check(cmx_intra_pvalues(eng.ctx(), mi.data(), key.data(), n, (int)opt.nbRateClasses, nullStat.data(), nullKey.data(),
                                  nullStat.size(), pv.data(), nsim.data()));
    }
    Vdouble permP;
    std::vector<int32_t> permN;
    if (opt.nullMethod == "permutations") miTest(eng, aln, nbTaxa, n, alphabetSize, masks, nbMasks, opt.maxNbPermutations, opt.seed, permP, permN);
    res.withPermutations = !permP.empty();
    res.withPValues = computePValues;
    // ---- the table (:646-689)
    size_t q = 0;
    for (size_t i = 0; i + 1 < n; ++i)
      for (size_t j = i + 1; j < n; ++j, ++q) {
        Row r;
        r.i = i; r.j = j;
        r.mi = mi[i * n + j];
        r.apc = res.averageMI[i] * res.averageMI[j] / res.fullAverageMI;
        r.rcw = res.averageMI[i] * res.averageMI[j] / 2.;
        r.hjoint = hj[i * n + j];
        r.hmin = std::min(res.entropy[i], res.entropy[j]);
        r.nmin = norms ? std::min((*norms)[i], (*norms)[j]) : std::numeric_limits<double>::quiet_NaN();
        r.permPValue = res.withPermutations ? permP[q] : std::numeric_limits<double>::quiet_NaN();
        r.permNb = res.withPermutations ? permN[q] : 0;
        r.bsPValue = computePValues ? pv[i * n + j] : std::numeric_limits<double>::quiet_NaN();
        r.bsNb = computePValues ? nsim[i * n + j] : 0;
        res.rows.push_back(r);
      }
    return res;
  }
};

namespace io {
// LegacySubstitutionMappingTools::writeToStream as called at CoETools.cpp:408-412 (format: one row per branch)
inline void writeToStream(const ProbabilisticSubstitutionMapping& mapping, const Vdouble& branchLengths,
                          const std::vector<int>& coordinates, size_t type, std::ostream& out) {
  if (coordinates.size() != mapping.getNumberOfSites())
    throw DimensionException("writeToStream: site coordinates.", coordinates.size(), mapping.getNumberOfSites());
  if (branchLengths.size() < mapping.getNumberOfBranches())
    throw DimensionException("writeToStream: branch lengths.", branchLengths.size(), mapping.getNumberOfBranches());
  out << "Branches" << "\t" << "Mean";
  for (int c : coordinates) out << "\tSite" << c;
  out << std::endl;
  for (size_t b = 0; b < mapping.getNumberOfBranches(); ++b) {
    out << b << "\t" << branchLengths[b];
    for (size_t i = 0; i < mapping.getNumberOfSites(); ++i) out << "\t" << mapping(b, i, type);
    out << std::endl;
  }
}
// statistics.txt rows of CoETools.cpp:698-722
inline void writeIntraStats(const std::vector<IntraStatRow>& rows, const std::vector<int>& coordinates, bool withNull,
                            std::ostream& out, const std::vector<int>* coordinates2 = nullptr) {
  out << "Group\tStat\tRCmin\tPRmin\tNmin";
  if (withNull) out << "\tPValue\tNsim";
  out << std::endl;
  const std::vector<int>& c2 = coordinates2 ? *coordinates2 : coordinates;
  for (const IntraStatRow& r : rows) {
    out << "[" << coordinates[r.i] << ";" << c2[r.j] << "]\t" << r.stat << "\t" << r.rcMin << "\t" << r.prMin << "\t"
        << r.nMin;
    if (withNull) {
      if (std::isnan(r.pValue)) out << "\tNA\t0";
      else out << "\t" << r.pValue << "\t" << r.nSim;
    }
    out << std::endl;
  }
}
// statistics.null.txt: AnalysisTools.cpp:580, 642 (inter: :680, 732)
// clustering.output.groups.file: the DataTable of CoMap.cpp:493-548
inline void writeGroups(const std::vector<Group>& groups, const std::vector<std::string>& siteNames,
                        const std::vector<bool>& isConstant, size_t maxGroupSize, std::ostream& out) {
  out << "Group\tSize\tIsConstant\tDmax\tStat\tNmin\n";
  for (const Group& g : groups) {
    if (g.size() > maxGroupSize) continue;
    bool c = false;
    for (size_t s : g.sites) c = c || isConstant[s];
    out << g.toString(siteNames) << "\t" << g.size() << "\t" << (c ? "yes" : "no") << "\t" << g.dmax << "\t" << g.stat
        << "\t" << g.nmin << "\n";
  }
}

inline void writeNull(const std::vector<NullDistributionRow>& rows, std::ostream& out) {
  out << "Stat\tRCmin\tPRmin\tNmin" << std::endl;
  for (const NullDistributionRow& r : rows) out << r.stat << "\t" << r.rcMin << "\t" << r.prMin << "\t" << r.nMin << std::endl;
}
// Mica's output.file (Mica.cpp:634-690) and null.output.file (:411-415, :494, :534)
inline void writeMica(const Mica::Result& res, const std::vector<int>& coordinates, std::ostream& out) {
  out << "Group\tMI\tAPC\tRCW\tHjoint\tHmin";
  if (res.withModel) out << "\tNmin";
  if (res.withPermutations) out << "\tPerm.p.value\tPerm.nb";
  if (res.withPValues) out << "\tBs.p.value\tBs.nb";
  out << std::endl;
  for (const Mica::Row& r : res.rows) {
    out << "[" << coordinates[r.i] << ";" << coordinates[r.j] << "]\t" << r.mi << "\t" << r.apc << "\t" << r.rcw << "\t" << r.hjoint
        << "\t" << r.hmin;
    if (res.withModel) out << "\t" << r.nmin;
    if (res.withPermutations) out << "\t" << r.permPValue << "\t" << r.permNb;
    if (res.withPValues) {
      if (std::isnan(r.bsPValue)) out << "\tNA\t0";
      else out << "\t" << r.bsPValue << "\t" << r.bsNb;
    }
    out << std::endl;
  }
}
inline void writeMicaNull(const Mica::Result& res, bool withNmin, std::ostream& out) {
  out << "MI\tHjoint\tHmin";
  if (withNmin) out << "\tNmin";
  out << std::endl;
  for (const Mica::NullRow& r : res.null) {
    out << r.mi << "\t" << r.hjoint << "\t" << r.hmin;
    if (withNmin) out << "\t" << r.nmin;
    out << std::endl;
  }
}
}  // namespace io

}  // namespace cmx
#endif  // COMAP_MI355X_ADAPTER_HPP
