"""Host-side mirror of Mica's null distributions on top of the C-ABI (first two of three):

  bootstrap_null          <- null.method = nonparametric-bootstrap, CoMap/Mica.cpp:399-468: nbRepCPU x nbRepRAM pairs of
                             sites drawn with replacement from the data, scored (MI, Hjoint, Hmin)
  parametric_null         <- null.method = parametric-bootstrap, CoMap/Mica.cpp:469-548: per replicate two alignments of
                             nbRepRAM sites are simulated under the model and column j of the one is scored against
                             column j of the other

Site indices come from the engine's counter-based generator scheme (Philox keyed by seed), not from Bio++'s global
generator: null distributions agree with the reference in distribution, not draw for draw.
  zscore_null / analysis  <- null.method = z-score, CoMap/Mica.cpp:549-607, and the output table of :634-690
  permutation_test        <- null.method = permutations, miTest CoMap/Mica.cpp:93-118 (fully resolved columns only)"""
import numpy as np


def bootstrap_indices(seed, nsites, nrep_cpu, nrep_ram):
    """index1 / index2 of SiteContainerTools::sampleSites (Mica.cpp:426-430), [nrep_cpu * nrep_ram] each: drawn behind the
    C-ABI (cmx_mica_bootstrap_indices, the engine's counter RNG) so that the C++ adapter and this mirror agree."""
    from .engine import mica_bootstrap_indices
    return mica_bootstrap_indices(seed, nsites, nrep_cpu, nrep_ram)


def bootstrap_null(engine, aln, entropy, seed, nrep_cpu=10, nrep_ram=100, nalpha=20, masks=None, norms=None):
    """-> dict(mi, hjoint, hmin[, nmin], index1, index2): the columns of the null output file (Mica.cpp:411-415)."""
    i1, i2 = bootstrap_indices(seed, aln.shape[1], nrep_cpu, nrep_ram)
    r = engine.mi_pairs(aln, i1, i2, None, nalpha, masks)
    out = dict(mi=r["mi"], hjoint=r["hjoint"], hmin=np.minimum(entropy[i1], entropy[i2]), index1=i1, index2=i2)
    if norms is not None:
        out["nmin"] = np.minimum(np.asarray(norms)[i1], np.asarray(norms)[i2])
    return out


def parametric_null(engine, seed, nrep_cpu=10, nrep_ram=100, nalpha=20, with_norms=False):
    """engine must hold the model (tree, Q, rates).  Simulated-site indices follow the intra null's scheme:
    g = ((rep * 2 + h) * nrep_ram + j).  All replicates go through the device in ONE batch per stage: one simulation of
    2 * nrep_cpu * nrep_ram sites, one MI call over the nrep_cpu * nrep_ram (j, j) column pairs and -- with_norms, Mica's
    `use_model` case, Mica.cpp:505-530 -- one mapping of all simulated sites for their norms.
    -> dict(mi, hjoint[, nmin]) in replicate order (the rows of the null output file, Mica.cpp:411-415)."""
    import torch
    n = nrep_cpu * nrep_ram
    dev = torch.device("cuda", engine.device)
    # simulator -> MI of the (j, j) pairs -> (norms) without leaving the device (cmx_simulate_dev, cmx_mi_pairs_dev,
    # cmx_map_sites_dev); only the null's columns come back
    d_aln = torch.empty((engine.T, 2 * n), dtype=torch.uint8, device=dev)
    engine.simulate_dev(seed, 0, 2 * n, d_aln)
    q = torch.arange(n, dtype=torch.int64, device=dev)
    rep, j = q // nrep_ram, q % nrep_ram
    i1, i2 = (rep * 2) * nrep_ram + j, (rep * 2 + 1) * nrep_ram + j
    mi, hj = torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)
    engine.mi_pairs_dev(d_aln, i1, i2, mi, hj, nalpha=nalpha)
    out = dict(mi=mi.cpu().numpy(), hjoint=hj.cpu().numpy())
    if with_norms:
        norm = torch.empty(2 * n, dtype=torch.float64, device=dev)
        engine.map_sites_dev(d_aln, norm=norm)
        out["nmin"] = torch.minimum(norm[i1], norm[i2]).cpu().numpy()
    return out


def parametric_null_via_host(engine, seed, nrep_cpu=10, nrep_ram=100, nalpha=20, with_norms=False):
    """the same null through host memory (simulate -> numpy -> mi_pairs): what parametric_null must equal bit for bit"""
    n = nrep_cpu * nrep_ram
    aln, _ = engine.simulate(seed, 0, 2 * n)
    rep, j = np.divmod(np.arange(n, dtype=np.int64), nrep_ram)
    i1 = (rep * 2) * nrep_ram + j
    i2 = (rep * 2 + 1) * nrep_ram + j
    r = engine.mi_pairs(aln, i1, i2, None, nalpha)
    out = dict(mi=r["mi"], hjoint=r["hjoint"])
    if with_norms:
        norm = engine.map_sites(aln, want_counts=False)["norm"]
        out["nmin"] = np.minimum(norm[i1], norm[i2])
    return out


def permutation_test(engine, aln, max_perm=1000, seed=0, nalpha=20, masks=None):
    """-> dense (pvalue [n, n], nperm [n, n]) filled for j > i: Perm.p.value / Perm.nb of Mica.cpp:667-668"""
    n = aln.shape[1]
    pv, npm = engine.mica_permutation_test(aln, max_perm, seed, nalpha, masks=masks)
    iu = np.triu_indices(n, 1)
    P, N = np.full((n, n), np.nan), np.zeros((n, n), dtype=np.int32)
    P[iu], N[iu] = pv, npm
    return P, N


def analysis(engine, aln, nalpha=20, masks=None, norms=None, null=None, nclasses=10, permutations=None):
    """The table Mica writes (Mica.cpp:634-690) as dense arrays: MI, Hjoint per pair, entropy / averageMI per column,
    fullAverageMI, and -- if `null` = (null_stat, null_key) is given -- Bs.p.value / Bs.nb per pair, binned on the
    model norms when given (withModel) and on min entropy otherwise (Mica.cpp:383-386, 672).  All numbers come from the
    device entry points; comap_amd.formats.write_mica turns the result into the reference's text."""
    r = engine.mi_columns(aln, None, nalpha, masks)
    avg, full = engine.mica_average_mi(r["mi"])
    out = dict(mi=r["mi"], hjoint=r["hjoint"], entropy=r["h1"], average_mi=avg, full_average_mi=full,
               norms=None if norms is None else np.asarray(norms, dtype=np.float64))
    if null is not None:
        key = out["entropy"] if norms is None else out["norms"]
        out["pvalue"], out["nsim"] = engine.intra_pvalues(r["mi"], key, nclasses, null[0], null[1])
    if permutations is not None:          # (max_perm, seed)
        out["perm_pvalue"], out["perm_nb"] = permutation_test(engine, aln, permutations[0], permutations[1], nalpha)
    return out


def zscore_null(engine, mi, entropy, which="MIp", norms=None):
    """-> (null_stat, null_key): every pair of the data set as one draw of the null (Mica.cpp:565-603)."""
    from .engine import MICA_MI, MICA_MIP, MICA_MIC
    kind = {"MI": MICA_MI, "MIp": MICA_MIP, "MIc": MICA_MIC}[which]
    return engine.mica_zscore_null(kind, mi, entropy if norms is None else norms)
