// comap_mi355x_bpp.hpp -- the reference's OWN call seams, with their exact Bio++-typed signatures, on top of the MI355X
// engine, so that CoMap.cpp / Mica.cpp call them unchanged (BASELINE.json north_star).
//
//   reference declaration (jydu/comap)                                         defined here as
//   CoETools::getVectors                      CoMap/CoETools.h:317-322         cmx::bpp::CoETools::getVectors
//   CoETools::computeIntraStats               CoMap/CoETools.h:363-371         cmx::bpp::CoETools::computeIntraStats
//   CoETools::computeInterStats               CoMap/CoETools.h:373-381         cmx::bpp::CoETools::computeInterStats
//   CoETools::computeIntraNullDistribution    CoMap/CoETools.h:383-389         cmx::bpp::CoETools::computeIntraNullDistribution
//   CoETools::computeInterNullDistribution    CoMap/CoETools.h:391-399         cmx::bpp::CoETools::computeInterNullDistribution
//   AnalysisTools::getNullDistributionIntraDR CoMap/AnalysisTools.h:248-260    cmx::bpp::AnalysisTools::getNullDistributionIntraDR
//   AnalysisTools::getNullDistributionInterDR CoMap/AnalysisTools.h:262-275    cmx::bpp::AnalysisTools::getNullDistributionInterDR
//   AnalysisTools::computeNorms               CoMap/AnalysisTools.h:198        cmx::bpp::AnalysisTools::computeNorms
//   AnalysisTools::compute{ScalarProduct,Cosinus,Correlation,Covariance}Matrix
//                                             CoMap/AnalysisTools.h:93-190     cmx::bpp::AnalysisTools::compute...Matrix (8 overloads)
//
// A maintainer swaps the engine in with two lines per translation unit:
//     #include "comap_mi355x_bpp.hpp"
//     namespace CoETools_impl = cmx::bpp;     // or: replace the bodies in CoETools.cpp / AnalysisTools.cpp by forwarding calls
//
// THIS BLOCK CANNOT BE COMPILED IN THIS REPOSITORY'S IMAGE: Bio++ >= 3.0.0 (CMakeLists.txt:107 of the reference) is
// neither vendored under /root/reference nor installed here, and there is no network.  Everything below the guard is
// therefore written against the Bio++ 3 interfaces as the reference itself uses them (the calls are the ones that
// appear in CoMap/CoETools.cpp, AnalysisTools.cpp, Mica.cpp; each is cited) and is excluded from the build when the
// Bio++ headers are absent; tests/test_adapter_cpp.py compiles this header in that state to make sure the guard holds.
// The Bio++-free layer it forwards to (comap_mi355x_adapter.hpp) IS compiled and tested here.
#ifndef COMAP_MI355X_BPP_HPP
#define COMAP_MI355X_BPP_HPP

#include "comap_mi355x_adapter.hpp"

#if defined(__has_include)
#if __has_include(<Bpp/Phyl/Legacy/Likelihood/DRTreeLikelihood.h>)
#define CMX_HAVE_BPP 1
#endif
#endif

#ifdef CMX_HAVE_BPP
#include <Bpp/App/ApplicationTools.h>
#include <Bpp/Numeric/Prob/DiscreteDistribution.h>
#include <Bpp/Phyl/Legacy/Likelihood/DRTreeLikelihood.h>
#include <Bpp/Phyl/Legacy/Likelihood/DRHomogeneousTreeLikelihood.h>
#include <Bpp/Phyl/Legacy/Likelihood/DRNonHomogeneousTreeLikelihood.h>
#include <Bpp/Phyl/Legacy/Mapping/ProbabilisticSubstitutionMapping.h>
#include <Bpp/Phyl/Legacy/Simulation/SequenceSimulator.h>
#include <Bpp/Phyl/Mapping/SubstitutionCount.h>
#include <Bpp/Phyl/Mapping/WeightedSubstitutionCount.h>
#include <Bpp/Phyl/Mapping/NaiveSubstitutionCount.h>
#include <Bpp/Phyl/Mapping/LabelSubstitutionCount.h>
#include <Bpp/Phyl/Tree/TreeTemplate.h>
#include <Bpp/Seq/Container/SiteContainer.h>

#include <fstream>
#include <map>
#include <tuple>

// the reference's own scorer / domain classes (CoMap/Statistics.h, CoMap/Domain.h): the seams take them by reference
#include "Statistics.h"
#include "Domain.h"

namespace cmx {
namespace bpp {

using ::bpp::ApplicationTools;

// ------------------------------------------------------------------------------------------------ Bio++ -> plain arrays
// tree: nodes in post-order with the root last == TreeTemplate::getNodes() order after the DR likelihood has unrooted
// the tree; branch b == position of the lower node == row order of writeToStream (.vec files).
struct TreeView {
  cmx::TreeArrays arrays;
  std::vector<int> nodeId;                 // position -> Bio++ node id
  std::map<int, int> position;             // Bio++ node id -> position
};

inline TreeView toArrays(const ::bpp::TreeTemplate<::bpp::Node>& tree, const ::bpp::AlignmentDataInterface& sites) {
  TreeView v;
  const auto nodes = tree.getNodes();      // post-order, root last (CoMap/ClusterTools.cpp:231 relies on the same order)
  for (size_t i = 0; i < nodes.size(); ++i) { v.position[nodes[i]->getId()] = (int)i; v.nodeId.push_back(nodes[i]->getId()); }
  for (const auto* n : nodes) {
    v.arrays.parent.push_back(n->hasFather() ? v.position[n->getFather()->getId()] : -1);
    v.arrays.branchLengths.push_back(n->hasFather() ? n->getDistanceToFather() : 0.);
  }
  for (size_t s = 0; s < sites.getNumberOfSequences(); ++s)      // alignment row s <-> leaf
    v.arrays.leafOfTaxon.push_back(v.position[tree.getNode(sites.sequence(s).getName())->getId()]);
  return v;
}

// generator, frequencies, rates; registers / weights of the substitution count:
//   B_k(x, y) = Q(x, y) [register(x, y) == k + 1] weight(x, y), zero diagonal  
inline cmx::ModelArrays toArrays(const ::bpp::SubstitutionModelInterface& model,
                                 const ::bpp::DiscreteDistributionInterface& rDist,
                                 const ::bpp::SubstitutionCountInterface& nijt) {
  cmx::ModelArrays m;
  const size_t S = model.getNumberOfStates();
  m.nbStates = (int)S;
  for (size_t x = 0; x < S; ++x) {
    m.frequencies.push_back(model.freq(x));
    for (size_t y = 0; y < S; ++y) m.generator.push_back(model.Qij(x, y));
  }
  for (size_t c = 0; c < rDist.getNumberOfCategories(); ++c) {
    m.rates.push_back(rDist.getCategory(c));
    m.rateProbabilities.push_back(rDist.getProbability(c));
  }
  const auto* weighted = dynamic_cast<const ::bpp::WeightedSubstitutionCountInterface*>(&nijt);
  const bool hasWeights = weighted && weighted->hasWeights();
  const bool naive = dynamic_cast<const ::bpp::NaiveSubstitutionCount*>(&nijt) != nullptr;
  const bool label = dynamic_cast<const ::bpp::LabelSubstitutionCount*>(&nijt) != nullptr;
  if (naive || label) {
    m.naive = true;
    m.naiveWeights.assign(S * S, 0.);
    size_t lab = 0;
    for (size_t x = 0; x < S; ++x)
      for (size_t y = 0; y < S; ++y) {
        if (x == y) continue;
        ++lab;   // LabelSubstitutionCount numbers the ordered pairs row by row, 1 .. S(S-1)
        m.naiveWeights[x * S + y] = label ? (double)lab : (hasWeights ? weighted->weights()->getIndex(model.getAlphabetStateAsInt(x), model.getAlphabetStateAsInt(y)) : 1.);
      }
  } else {
    const size_t K = nijt.getNumberOfSubstitutionTypes();
    const auto& reg = nijt.substitutionRegister();
    m.registers.assign(K * S * S, 0.);
    m.nbTypes = (int)K;
    for (size_t x = 0; x < S; ++x)
      for (size_t y = 0; y < S; ++y) {
        if (x == y) continue;
        const size_t type = reg.getType(x, y);           // 0 = not counted, k + 1 = type k
        if (type == 0) continue;
        const double w = hasWeights ? weighted->weights()->getIndex(model.getAlphabetStateAsInt(x), model.getAlphabetStateAsInt(y)) : 1.;
        m.registers[(type - 1) * S * S + x * S + y] = model.Qij(x, y) * w;
      }
  }
  m.clampNegative = !hasWeights;             // Bio++ clamps round-off negatives of UNWEIGHTED counts only (A.4)
  return m;
}

// alignment as the engine wants it: [taxon][site] codes; code < S = model state, code >= S = ambiguity id whose bit
// mask lists the compatible states (X / gap = all ones) -- what the DR likelihood's leaf initialisation does (A.2)
struct AlignmentView {
  std::vector<uint8_t> codes;               // [T][N]
  std::vector<uint32_t> masks;              // [S + number of ambiguity ids]
  size_t nbSites = 0;
};

inline AlignmentView toArrays(const ::bpp::SiteContainerInterface& sites, const ::bpp::SubstitutionModelInterface& model) {
  AlignmentView a;
  const size_t T = sites.getNumberOfSequences(), N = sites.getNumberOfSites(), S = model.getNumberOfStates();
  a.nbSites = N;
  a.codes.assign(T * N, 0);
  for (size_t x = 0; x < S; ++x) a.masks.push_back(1u << x);
  std::map<uint32_t, uint8_t> idOfMask;
  const auto alphabet = sites.getAlphabet();
  for (size_t s = 0; s < T; ++s)
    for (size_t i = 0; i < N; ++i) {
      const int sym = sites.site(i)[s];
      uint32_t mask = 0;
      for (size_t x = 0; x < S; ++x)
        if (alphabet->isGap(sym) || alphabet->isUnresolved(sym)
                ? (alphabet->isGap(sym) || [&] { for (int al : alphabet->getAlias(sym)) if (al == model.getAlphabetStateAsInt(x)) return true; return false; }())
                : sym == model.getAlphabetStateAsInt(x))
          mask |= 1u << x;
      uint8_t code;
      if (mask && !(mask & (mask - 1))) {     // exactly one state
        code = 0;
        while (!((mask >> code) & 1u)) ++code;
      } else {
        auto it = idOfMask.find(mask);
        if (it == idOfMask.end()) {
          it = idOfMask.emplace(mask, (uint8_t)a.masks.size()).first;
          a.masks.push_back(mask);
        }
        code = it->second;
      }
      a.codes[s * N + i] = code;
    }
  return a;
}

// the engine behind a likelihood object + substitution count (the reference re-initialises its likelihood for every
// simulated batch, AnalysisTools.cpp:592-593 -- the engine keeps tree, model and count operators on the device instead)
struct Seam {
  TreeView tree;
  cmx::Engine engine;
  Seam(const ::bpp::DRTreeLikelihoodInterface& drtl, const ::bpp::SubstitutionCountInterface& nijt)
      : tree(toArrays(dynamic_cast<const ::bpp::TreeTemplate<::bpp::Node>&>(drtl.tree()), drtl.data())),
        engine(tree.arrays, toArrays(drtl.substitutionModel(0, 0), *drtl.getRateDistribution(), nijt)) {}
};

// One Seam (= one cmx_ctx: host-side walk verification + a multi-GB device workspace) per (likelihood, substitution
// count) pair and parameter state, kept for the life of the process: CoMap.cpp calls getVectors, computeIntraStats and
// the null seams on the same pair in turn.  The fingerprint covers what a context depends on, so that an optimised or
// re-parameterised likelihood gets a fresh context.
inline Seam& seamFor(const ::bpp::DRTreeLikelihoodInterface& drtl, const ::bpp::SubstitutionCountInterface& nijt) {
  struct Key {
    const void *tl, *nijt;
    std::vector<double> print;
    bool operator<(const Key& o) const { return std::tie(tl, nijt, print) < std::tie(o.tl, o.nijt, o.print); }
  };
  static std::map<Key, std::unique_ptr<Seam>> cache;
  Key k{&drtl, &nijt, {}};
  for (const auto* n : dynamic_cast<const ::bpp::TreeTemplate<::bpp::Node>&>(drtl.tree()).getNodes())
    k.print.push_back(n->hasFather() ? n->getDistanceToFather() : 0.);
  const auto& model = drtl.substitutionModel(0, 0);
  for (size_t x = 0; x < model.getNumberOfStates(); ++x) { k.print.push_back(model.freq(x)); k.print.push_back(model.Qij(x, (x + 1) % model.getNumberOfStates())); }
  const auto& rDist = *drtl.getRateDistribution();
  for (size_t c = 0; c < rDist.getNumberOfCategories(); ++c) { k.print.push_back(rDist.getCategory(c)); k.print.push_back(rDist.getProbability(c)); }
  auto it = cache.find(k);
  if (it == cache.end()) it = cache.emplace(std::move(k), std::make_unique<Seam>(drtl, nijt)).first;
  return *it->second;
}

// ---- the caller's sequence simulator.  The engine simulates under the likelihood's own tree / model / rate distribution
// with its counter-based generator; what a NonHomogeneousSequenceSimulator built from the same objects (CoMap.cpp:209-219)
// adds is ONE switch, enableContinuousRates (CoMap.cpp:213, option simulations.continuous).  The legacy simulator keeps
// that flag private, so the seams learn it from a registry: replace CoMap.cpp:213 by
//     cmx::bpp::enableContinuousRates(*seqSim, continuousSim);
// Any other kind of SequenceSimulatorInterface cannot be honoured and is refused -- never silently replaced.
inline std::map<const void*, bool>& continuousRegistry() {
  static std::map<const void*, bool> reg;
  return reg;
}
inline void enableContinuousRates(::bpp::NonHomogeneousSequenceSimulator& seqSim, bool yn) {
  seqSim.enableContinuousRates(yn);
  continuousRegistry()[&seqSim] = yn;
}
// alpha (and the invariant mass) of Gamma(n, alpha) / Invariant(dist = Gamma(n, alpha), p): what the continuous draw needs
inline cmx::ContinuousRates continuousRatesOf(const ::bpp::DiscreteDistributionInterface& rDist) {
  cmx::ContinuousRates cr;
  bool haveAlpha = false;
  const auto& pl = rDist.getParameters();
  for (size_t q = 0; q < pl.size(); ++q) {
    const std::string name = pl[q].getName();
    if (name.size() >= 5 && name.compare(name.size() - 5, 5, "alpha") == 0) { cr.gammaAlpha = pl[q].getValue(); haveAlpha = true; }
    if (name == "p" || (name.size() >= 2 && name.compare(name.size() - 2, 2, ".p") == 0)) cr.pInvariant = pl[q].getValue();
  }
  if (!haveAlpha) throw cmx::Exception("cmx::bpp: simulations.continuous needs a Gamma rate distribution (no alpha parameter found)");
  return cr;
}
// nullptr: the discrete simulator (the engine's default); otherwise the continuous-rate parameters
inline std::unique_ptr<cmx::ContinuousRates> simulatorMode(const ::bpp::SequenceSimulatorInterface& seqSim,
                                                           const ::bpp::DRTreeLikelihoodInterface& drtl) {
  if (!dynamic_cast<const ::bpp::NonHomogeneousSequenceSimulator*>(&seqSim))
    throw cmx::Exception("cmx::bpp: only a NonHomogeneousSequenceSimulator built from the likelihood's own tree, model and rate "
                         "distribution (CoMap.cpp:209-219) can be replaced by the engine's simulator");
  const auto it = continuousRegistry().find(&seqSim);
  if (it == continuousRegistry().end() || !it->second) return nullptr;
  return std::make_unique<cmx::ContinuousRates>(continuousRatesOf(*drtl.getRateDistribution()));
}

// reference Statistic object -> the engine's statistic (CoETools::getStatistic builds exactly these, CoETools.cpp:535-600)
inline std::unique_ptr<cmx::Statistic> toEngineStatistic(const ::Statistic& statistic) {
  if (dynamic_cast<const ::CorrectedCorrelationStatistic*>(&statistic)) {
    auto s = std::make_unique<cmx::CorrectedCorrelationStatistic>();
    const auto& ref = dynamic_cast<const ::CorrectedCorrelationStatistic&>(statistic);
    s->setMeanVector(ref.getMeanVector());
    return s;
  }
  if (dynamic_cast<const ::CorrelationStatistic*>(&statistic)) return std::make_unique<cmx::CorrelationStatistic>();
  if (dynamic_cast<const ::CovarianceStatistic*>(&statistic)) return std::make_unique<cmx::CovarianceStatistic>();
  if (dynamic_cast<const ::CosinusStatistic*>(&statistic)) return std::make_unique<cmx::CosinusStatistic>();
  if (dynamic_cast<const ::CosubstitutionNumberStatistic*>(&statistic)) return std::make_unique<cmx::CosubstitutionNumberStatistic>();
  if (dynamic_cast<const ::CompensationStatistic*>(&statistic)) return std::make_unique<cmx::CompensationStatistic>();
  // The reference keeps the bounds in a private Domain (Statistics.h:309) without an accessor: INTEGRATION.md section 2 adds
  // the one-line getter `const Domain& getDomain() const { return domain_; }` this needs.  Every bounds vector the factory
  // builds (CoETools.cpp:577-593: MI(threshold) and the unit bins of nijt = Label) goes through as it is.
  if (const auto* mi = dynamic_cast<const ::DiscreteMutualInformationStatistic*>(&statistic))
    return std::make_unique<cmx::DiscreteMutualInformationStatistic>(mi->getDomain().getBounds());
  throw cmx::Exception("cmx::bpp: this Statistic has no device kernel (MutualInformationStatistic on continuous counts)");
}

// engine mapping -> the Bio++ table the callers index as mapping(branch, site, type) (CoMap/ClusterTools.cpp:237)
inline std::unique_ptr<::bpp::LegacyProbabilisticSubstitutionMapping> toBpp(const cmx::ProbabilisticSubstitutionMapping& m,
                                                                           const ::bpp::Tree& tree, const TreeView& view,
                                                                           std::shared_ptr<const ::bpp::SubstitutionCountInterface> nijt) {
  auto out = std::make_unique<::bpp::LegacyProbabilisticSubstitutionMapping>(tree, nijt, m.getNumberOfSites());
  for (size_t i = 0; i < m.getNumberOfSites(); ++i)
    for (size_t b = 0; b < m.getNumberOfBranches(); ++b)
      for (size_t k = 0; k < m.getNumberOfSubstitutionTypes(); ++k)
        (*out)(out->getNodeIndex(view.nodeId[b]), i, k) = m(b, i, k);
  return out;
}

inline cmx::ProbabilisticSubstitutionMapping fromBpp(const ::bpp::LegacyProbabilisticSubstitutionMapping& mapping, const TreeView& view) {
  cmx::ProbabilisticSubstitutionMapping m(mapping.getNumberOfSites(), view.nodeId.size() - 1, mapping.getNumberOfSubstitutionTypes());
  for (size_t i = 0; i < m.getNumberOfSites(); ++i)
    for (size_t b = 0; b < m.getNumberOfBranches(); ++b)
      for (size_t k = 0; k < m.getNumberOfSubstitutionTypes(); ++k)
        m(b, i, k) = mapping(mapping.getNodeIndex(view.nodeId[b]), i, k);
  m.norms = cmx::AnalysisTools::computeNorms(m);
  return m;
}

// seed of the engine's counter-based simulator: the application's `seed=` option when given (Bio++ seeds its global
// generator from it), else a draw from that generator -- null distributions agree in distribution, not draw for draw
inline uint64_t seedFrom(std::map<std::string, std::string>& params) {
  if (params.count("seed")) return (uint64_t)ApplicationTools::getParameter<long>("seed", params, 0, "", true, 4);
  return (uint64_t)::bpp::RandomTools::giveIntRandomNumberBetweenZeroAndEntry<long>(1L << 62);
}

// ------------------------------------------------------------------------------------------------ AnalysisTools seams
class AnalysisTools {
 public:
  // CoMap/AnalysisTools.h:198 -- norms of every site's substitution vector (fused into the mapping kernel; recomputed
  // here from the table because the caller may have loaded it from a .vec file)
  static ::bpp::Vdouble computeNorms(const ::bpp::LegacyProbabilisticSubstitutionMapping& mapping) {
    ::bpp::Vdouble n(mapping.getNumberOfSites());
    for (size_t i = 0; i < n.size(); ++i) n[i] = ::bpp::LegacySubstitutionMappingTools::computeNormForSite(mapping, i);
    return n;
  }

  // CoMap/AnalysisTools.h:93-190, bodies AnalysisTools.cpp:102-339: matrices of a pairwise function of plain vectors (not
  // called from the reference's main() today; kept because they are public members of the seam).  They need no tree and no
  // model: one model-less context on device 0, created at the first call.
  static const cmx::Engine& plainEngine() { static cmx::Engine e(0); return e; }
  static ::bpp::VVdouble computeScalarProductMatrix(const ::bpp::VVdouble& vectors) { return cmx::AnalysisTools::computeScalarProductMatrix(plainEngine(), vectors); }
  static ::bpp::VVdouble computeScalarProductMatrix(const ::bpp::VVdouble& vectors1, const ::bpp::VVdouble& vectors2, bool independantComparisons) {
    return rethrow([&] { return cmx::AnalysisTools::computeScalarProductMatrix(plainEngine(), vectors1, vectors2, independantComparisons); });
  }
  static ::bpp::VVdouble computeCosinusMatrix(const ::bpp::VVdouble& vectors) { return cmx::AnalysisTools::computeCosinusMatrix(plainEngine(), vectors); }
  static ::bpp::VVdouble computeCosinusMatrix(const ::bpp::VVdouble& vectors1, const ::bpp::VVdouble& vectors2, bool independantComparisons) {
    return rethrow([&] { return cmx::AnalysisTools::computeCosinusMatrix(plainEngine(), vectors1, vectors2, independantComparisons); });
  }
  static ::bpp::VVdouble computeCorrelationMatrix(const ::bpp::VVdouble& vectors) { return cmx::AnalysisTools::computeCorrelationMatrix(plainEngine(), vectors); }
  static ::bpp::VVdouble computeCorrelationMatrix(const ::bpp::VVdouble& vectors1, const ::bpp::VVdouble& vectors2, bool independantComparisons) {
    return rethrow([&] { return cmx::AnalysisTools::computeCorrelationMatrix(plainEngine(), vectors1, vectors2, independantComparisons); });
  }
  static ::bpp::VVdouble computeCovarianceMatrix(const ::bpp::VVdouble& vectors) { return cmx::AnalysisTools::computeCovarianceMatrix(plainEngine(), vectors); }
  static ::bpp::VVdouble computeCovarianceMatrix(const ::bpp::VVdouble& vectors1, const ::bpp::VVdouble& vectors2, bool independantComparisons) {
    return rethrow([&] { return cmx::AnalysisTools::computeCovarianceMatrix(plainEngine(), vectors1, vectors2, independantComparisons); });
  }
  // the adapter's DimensionException as the reference's (bpp::DimensionException, AnalysisTools.cpp:139-147)
  template <class F> static ::bpp::VVdouble rethrow(F f) {
    try { return f(); }
    catch (const cmx::DimensionException& e) { throw ::bpp::DimensionException(e.what(), e.got(), e.expected()); }
  }

  // CoMap/AnalysisTools.h:248-260, body AnalysisTools.cpp:564-658.  The engine simulates under the likelihood's own tree /
  // model / rates with its counter-based generator; `seqSim` decides between discrete and continuous rates (simulatorMode
  // above) and is refused when it is anything else.  average / joint select the mapping variant (AnalysisTools.cpp:598-633).
  static void getNullDistributionIntraDR(std::shared_ptr<::bpp::DRTreeLikelihoodInterface> drtl,
                                         const ::bpp::SequenceSimulatorInterface& seqSim,
                                         std::shared_ptr<::bpp::SubstitutionCountInterface> nijt, const ::Statistic& statistic,
                                         std::ostream* out, ::bpp::VVdouble* simstats, const ::Domain* rateDomain, size_t repCPU,
                                         size_t repRAM, bool average, bool joint, bool verbose = true) {
    Seam& seam = seamFor(*drtl, *nijt);
    seam.engine.setMappingOptions(average, joint);
    const auto continuous = simulatorMode(seqSim, *drtl);
    const auto stat = toEngineStatistic(statistic);
    std::vector<cmx::NullDistributionRow> rows;
    std::unique_ptr<cmx::Domain> dom;
    if (rateDomain) dom = std::make_unique<cmx::Domain>(rateDomain->getLowerBound(), rateDomain->getUpperBound(), rateDomain->getSize());
    cmx::VVdouble sims;
    if (simstats) sims = *simstats;
    cmx::AnalysisTools::getNullDistributionIntraDR(seam.engine, *stat, seedOf(drtl.get()), repCPU, repRAM, out ? &rows : nullptr,
                                                   simstats ? &sims : nullptr, dom.get(), 0, continuous.get());
    if (simstats) *simstats = sims;
    if (out) {
      *out << "Stat\tRCmin\tPRmin\tNmin" << std::endl;                 // AnalysisTools.cpp:580
      cmx::io::writeNull(rows, *out);
    }
    if (verbose) ApplicationTools::displayTaskDone();
  }

  // CoMap/AnalysisTools.h:262-275, body AnalysisTools.cpp:662-735
  static void getNullDistributionInterDR(std::shared_ptr<::bpp::DRTreeLikelihoodInterface> drtl1,
                                         std::shared_ptr<::bpp::DRTreeLikelihoodInterface> drtl2,
                                         const ::bpp::SequenceSimulatorInterface& seqSim1,
                                         const ::bpp::SequenceSimulatorInterface& seqSim2,
                                         std::shared_ptr<::bpp::SubstitutionCountInterface> nijt1,
                                         std::shared_ptr<::bpp::SubstitutionCountInterface> nijt2, const ::Statistic& statistic,
                                         std::ostream& out, size_t repCPU, size_t repRAM, bool average, bool joint,
                                         bool verbose = true) {
    Seam &seam1 = seamFor(*drtl1, *nijt1), &seam2 = seamFor(*drtl2, *nijt2);
    seam1.engine.setMappingOptions(average, joint);
    seam2.engine.setMappingOptions(average, joint);
    if (simulatorMode(seqSim1, *drtl1) || simulatorMode(seqSim2, *drtl2))
      throw cmx::Exception("cmx::bpp: simulations.continuous is not available for the two-data-set null (cmx_null_inter simulates with the discrete classes)");
    const auto stat = toEngineStatistic(statistic);
    std::vector<cmx::NullDistributionRow> rows;
    cmx::AnalysisTools::getNullDistributionInterDR(seam1.engine, seam2.engine, *stat, seedOf(drtl1.get()), repCPU, repRAM, &rows);
    out << "Stat\tRCmin\tPRmin\tNmin" << std::endl;                    // AnalysisTools.cpp:677
    cmx::io::writeNull(rows, out);
    if (verbose) ApplicationTools::displayTaskDone();
  }

 private:
  // one stream of seeds per likelihood object and process (the reference draws from one global generator)
  static uint64_t seedOf(const void* key) {
    static std::map<const void*, uint64_t> turn;
    return (uint64_t)::bpp::RandomTools::giveIntRandomNumberBetweenZeroAndEntry<long>(1L << 62) + turn[key]++;
  }
};

// ------------------------------------------------------------------------------------------------ CoETools seams
class CoETools {
 public:
  // CoMap/CoETools.h:317-322, body CoETools.cpp:366-416: load a .vec file when input.vectors.file is set, otherwise
  // map every site on the device; write output.vectors.file.  Same option keys, same file bytes.
  static std::unique_ptr<::bpp::LegacyProbabilisticSubstitutionMapping> getVectors(
      std::shared_ptr<const ::bpp::DRTreeLikelihoodInterface> drtl, std::shared_ptr<::bpp::SubstitutionCountInterface> substitutionCount,
      const ::bpp::SiteContainerInterface& completeSites, std::map<std::string, std::string>& params, const std::string& suffix = "") {
    const std::string inputVectorsFilePath = ApplicationTools::getAFilePath("input.vectors.file", params, false, true, suffix, false);
    if (inputVectorsFilePath != "none") {                                // CoETools.cpp:374-385: unchanged behaviour
      ApplicationTools::displayResult("Substitution mapping in file:", inputVectorsFilePath);
      std::ifstream sc(inputVectorsFilePath.c_str(), std::ios::in);
      auto substitutions = std::make_unique<::bpp::LegacyProbabilisticSubstitutionMapping>(drtl->tree(), substitutionCount, completeSites.getNumberOfSites());
      ::bpp::LegacySubstitutionMappingTools::readFromStream(sc, *substitutions, 0);
      return substitutions;
    }
    const bool average = ApplicationTools::getBooleanParameter("nijt.average", params, true, "", true, 4);
    const bool joint = ApplicationTools::getBooleanParameter("nijt.joint", params, true, "", true, 4);
    Seam& seam = seamFor(*drtl, *substitutionCount);
    seam.engine.setMappingOptions(average, joint);                       // CoETools.cpp:393-406
    const AlignmentView aln = toArrays(completeSites, drtl->substitutionModel(0, 0));
    const auto mapping = cmx::CoETools::getVectors(seam.engine, aln.codes.data(), aln.nbSites, aln.masks.data(), aln.masks.size());
    auto substitutions = toBpp(*mapping, drtl->tree(), seam.tree, substitutionCount);
    const std::string outputVectorsFilePath = ApplicationTools::getAFilePath("output.vectors.file", params, false, false, suffix, false);
    if (outputVectorsFilePath != "none") {                               // CoETools.cpp:408-412
      std::ofstream outputVectors(outputVectorsFilePath.c_str(), std::ios::out);
      ::bpp::LegacySubstitutionMappingTools::writeToStream(*substitutions, completeSites, 0, outputVectors);
      ApplicationTools::displayResult("Wrote substitution vectors to file", outputVectorsFilePath);
    }
    return substitutions;
  }

  // CoMap/CoETools.h:383-389, body CoETools.cpp:836-872
  static std::vector<std::vector<double>>* computeIntraNullDistribution(std::shared_ptr<::bpp::DRTreeLikelihoodInterface> drtl,
                                                                      const ::Domain* rateDomain,
                                                                      const ::bpp::SequenceSimulatorInterface& seqSim,
                                                                      std::shared_ptr<::bpp::SubstitutionCountInterface> nijt,
                                                                      const ::Statistic& statistic,
                                                                      std::map<std::string, std::string>& params) {
    const std::string path = ApplicationTools::getAFilePath("statistic.null.output.file", params, false, false, "", false);   // CoETools.cpp:844
    const size_t nbRepCPU = ApplicationTools::getParameter<size_t>("statistic.null.nb_rep_CPU", params, 100);                   // :853
    const size_t nbRepRAM = ApplicationTools::getParameter<size_t>("statistic.null.nb_rep_RAM", params, 1000);                  // :854
    const bool average = ApplicationTools::getBooleanParameter("nijt.average", params, true, "", true, 4);
    const bool joint = ApplicationTools::getBooleanParameter("nijt.joint", params, true, "", true, 4);
    const bool computePValue = ApplicationTools::getBooleanParameter("statistic.null.compute_pvalue", params, false);           // :858
    std::unique_ptr<std::ofstream> simout;
    if (path != "none") simout = std::make_unique<std::ofstream>(path.c_str(), std::ios::out);
    std::vector<std::vector<double>>* simstats = nullptr;
    if (computePValue) simstats = new std::vector<std::vector<double>>(rateDomain ? rateDomain->getSize() : 1);
    AnalysisTools::getNullDistributionIntraDR(drtl, seqSim, nijt, statistic, simout.get(), simstats, rateDomain, nbRepCPU, nbRepRAM, average, joint, true);
    return simstats;
  }

  // CoMap/CoETools.h:391-399, body CoETools.cpp:873-897
  static void computeInterNullDistribution(std::shared_ptr<::bpp::DRTreeLikelihoodInterface> drtl1,
                                           std::shared_ptr<::bpp::DRTreeLikelihoodInterface> drtl2,
                                           const ::bpp::SequenceSimulatorInterface& seqSim1, const ::bpp::SequenceSimulatorInterface& seqSim2,
                                           std::shared_ptr<::bpp::SubstitutionCountInterface> nijt1,
                                           std::shared_ptr<::bpp::SubstitutionCountInterface> nijt2, const ::Statistic& statistic,
                                           std::map<std::string, std::string>& params) {
    const std::string path = ApplicationTools::getAFilePath("statistic.null.output.file", params, true, false);                 // CoETools.cpp:886
    const size_t nbRepCPU = ApplicationTools::getParameter<size_t>("statistic.null.nb_rep_CPU", params, 10);                    // :889
    const size_t nbRepRAM = ApplicationTools::getParameter<size_t>("statistic.null.nb_rep_RAM", params, 1000);                  // :890
    const bool average = ApplicationTools::getBooleanParameter("nijt.average", params, true, "", true, 4);
    const bool joint = ApplicationTools::getBooleanParameter("nijt.joint", params, true, "", true, 4);
    std::ofstream simout(path.c_str(), std::ios::out);
    AnalysisTools::getNullDistributionInterDR(drtl1, drtl2, seqSim1, seqSim2, nijt1, nijt2, statistic, simout, nbRepCPU, nbRepRAM, average, joint, true);
  }

  // CoMap/CoETools.h:363-371, body CoETools.cpp:604-728: statistic of every pair of sites, filters, conditional
  // p-values; statistics.txt is written by the same writer the engine's tests pin to the reference's column layout.
  static void computeIntraStats(const ::bpp::DRTreeLikelihoodInterface& tl, const ::bpp::SequenceSimulatorInterface& seqSim,
                                const ::bpp::SiteContainerInterface& completeSites, ::bpp::LegacyProbabilisticSubstitutionMapping& mapping,
                                std::shared_ptr<::bpp::SubstitutionCountInterface> nijt, const ::Statistic& statistic, bool computeNull,
                                std::map<std::string, std::string>& params) {
    const std::string path = ApplicationTools::getAFilePath("statistic.output.file", params, true, false);                      // CoETools.cpp:617
    std::ofstream statOut(path.c_str(), std::ios::out);
    Seam& seam = seamFor(tl, *nijt);
    seam.engine.setMappingOptions(ApplicationTools::getBooleanParameter("nijt.average", params, true, "", true, 4),
                                  ApplicationTools::getBooleanParameter("nijt.joint", params, true, "", true, 4));
    const auto continuous = simulatorMode(seqSim, tl);                    // the null of CoETools.cpp:641-653 draws from seqSim
    cmx::ProbabilisticSubstitutionMapping m = fromBpp(mapping, seam.tree);
    const std::vector<size_t> classes = tl.getRateClassWithMaxPostProbPerSite();                                                // :669
    const ::bpp::Vdouble rates = tl.getPosteriorRatePerSite();                                                                  // :670
    m.rateClasses.assign(classes.begin(), classes.end());
    m.posteriorRates = rates;
    cmx::PairFilters f;
    f.minRateClass = ::CoETools::getMinRateClass(params);
    f.minRate = ::CoETools::getMinRate(params);
    f.maxRateClassDiff = ::CoETools::getMaxRateClassDiff(params);
    f.maxRateDiff = ::CoETools::getMaxRateDiff(params);
    f.minStatistic = ::CoETools::getStatisticMin(params);
    const size_t nbRepCPU = ApplicationTools::getParameter<size_t>("statistic.null.nb_rep_CPU", params, 100);
    const size_t nbRepRAM = ApplicationTools::getParameter<size_t>("statistic.null.nb_rep_RAM", params, 1000);
    const unsigned nbRateClasses = ApplicationTools::getParameter<unsigned>("statistic.null.nb_rate_classes", params, 10);      // :638
    const auto stat = toEngineStatistic(statistic);
    const auto rows = cmx::CoETools::computeIntraStats(seam.engine, m, *stat, computeNull, seedFrom(params), nbRepCPU, nbRepRAM, nbRateClasses, f,
                                                       continuous.get());
    std::vector<int> coordinates(completeSites.getNumberOfSites());
    for (size_t i = 0; i < coordinates.size(); ++i) coordinates[i] = completeSites.site(i).getCoordinate();
    cmx::io::writeIntraStats(rows, coordinates, computeNull, statOut);
  }

  // CoMap/CoETools.h:373-381, body CoETools.cpp:732-832 (no p-values on this path in the reference either)
  static void computeInterStats(const ::bpp::DiscreteRatesAcrossSitesTreeLikelihoodInterface& tl1,
                                const ::bpp::DiscreteRatesAcrossSitesTreeLikelihoodInterface& tl2,
                                const ::bpp::SiteContainerInterface& completeSites1, const ::bpp::SiteContainerInterface& completeSites2,
                                ::bpp::LegacyProbabilisticSubstitutionMapping& mapping1, ::bpp::LegacyProbabilisticSubstitutionMapping& mapping2,
                                const ::Statistic& statistic, std::map<std::string, std::string>& params) {
    const std::string path = ApplicationTools::getAFilePath("statistic.output.file", params, true, false);                      // CoETools.cpp:745
    std::ofstream statOut(path.c_str(), std::ios::out);
    // the pair kernel needs no model: a model-free context serves both tables (same branches, CoETools.cpp:752-756)
    cmx::Engine engine;
    TreeView v1, v2;
    for (size_t b = 0; b + 1 < mapping1.getNumberOfBranches() + 1; ++b) { v1.nodeId.push_back(mapping1.getNode(b)->getId()); v2.nodeId.push_back(mapping2.getNode(b)->getId()); }
    v1.nodeId.push_back(-1); v2.nodeId.push_back(-1);
    cmx::ProbabilisticSubstitutionMapping m1 = fromBpp(mapping1, v1), m2 = fromBpp(mapping2, v2);
    const std::vector<size_t> c1 = tl1.getRateClassWithMaxPostProbPerSite(), c2 = tl2.getRateClassWithMaxPostProbPerSite();    // :781-784
    m1.rateClasses.assign(c1.begin(), c1.end()); m2.rateClasses.assign(c2.begin(), c2.end());
    m1.posteriorRates = tl1.getPosteriorRatePerSite(); m2.posteriorRates = tl2.getPosteriorRatePerSite();
    cmx::PairFilters f1, f2;
    f1.minRateClass = ::CoETools::getMinRateClass(params); f2.minRateClass = ::CoETools::getMinRateClass(params, "2");
    f1.minRate = ::CoETools::getMinRate(params); f2.minRate = ::CoETools::getMinRate(params, "2");
    f1.maxRateClassDiff = f2.maxRateClassDiff = ::CoETools::getMaxRateClassDiff(params);
    f1.maxRateDiff = f2.maxRateDiff = ::CoETools::getMaxRateDiff(params);
    f1.minStatistic = f2.minStatistic = ::CoETools::getStatisticMin(params);
    const bool indepComp = ::CoETools::haveToPerformIndependantComparisons(params);                                             // :759
    const auto stat = toEngineStatistic(statistic);
    const auto rows = cmx::CoETools::computeInterStats(engine, m1, m2, *stat, indepComp, f1, f2);
    std::vector<int> coord1(completeSites1.getNumberOfSites()), coord2(completeSites2.getNumberOfSites());
    for (size_t i = 0; i < coord1.size(); ++i) coord1[i] = completeSites1.site(i).getCoordinate();
    for (size_t i = 0; i < coord2.size(); ++i) coord2[i] = completeSites2.site(i).getCoordinate();
    cmx::io::writeIntraStats(rows, coord1, false, statOut, &coord2);   // same columns, CoETools.cpp:777, 814-826
  }
};

}  // namespace bpp
}  // namespace cmx
#endif  // CMX_HAVE_BPP

#endif  // COMAP_MI355X_BPP_HPP
