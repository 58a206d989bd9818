"""Host-side mirror of Mica's null distributions on top of the C-ABI (first two of three):

  bootstrap_null          <- null.method = nonparametric-bootstrap, CoMap/Mica.cpp:399-468: nbRepCPU x nbRepRAM pairs of
                             sites drawn with replacement from the data, scored (MI, Hjoint, Hmin)
  parametric_null         <- null.method = parametric-bootstrap, CoMap/Mica.cpp:469-548: per replicate two alignments of
                             nbRepRAM sites are simulated under the model and column j of the one is scored against
                             column j of the other

Site indices come from the engine's counter-based generator scheme (Philox keyed by seed), not from Bio++'s global
generator: null distributions agree with the reference in distribution, not draw for draw.
The permutation test (miTest, Mica.cpp:93-118) is sequential per pair and is not offered."""
import numpy as np


def bootstrap_indices(seed, nsites, nrep_cpu, nrep_ram):
    """index1 / index2 of SiteContainerTools::sampleSites (Mica.cpp:426-430), [nrep_cpu * nrep_ram] each."""
    rng = np.random.Generator(np.random.Philox(key=int(seed)))
    n = nrep_cpu * nrep_ram
    idx = rng.integers(0, nsites, size=(nrep_cpu, 2, nrep_ram))
    return idx[:, 0, :].reshape(n).astype(np.int64), idx[:, 1, :].reshape(n).astype(np.int64)


def bootstrap_null(engine, aln, entropy, seed, nrep_cpu=10, nrep_ram=100, nalpha=20, masks=None, norms=None):
    """-> dict(mi, hjoint, hmin[, nmin], index1, index2): the columns of the null output file (Mica.cpp:411-415)."""
    i1, i2 = bootstrap_indices(seed, aln.shape[1], nrep_cpu, nrep_ram)
    r = engine.mi_pairs(aln, i1, i2, None, nalpha, masks)
    out = dict(mi=r["mi"], hjoint=r["hjoint"], hmin=np.minimum(entropy[i1], entropy[i2]), index1=i1, index2=i2)
    if norms is not None:
        out["nmin"] = np.minimum(np.asarray(norms)[i1], np.asarray(norms)[i2])
    return out


def parametric_null(engine, seed, nrep_cpu=10, nrep_ram=100, nalpha=20):
    """engine must hold the model (tree, Q, rates).  Simulated-site indices follow the intra null's scheme:
    g = ((rep * 2 + h) * nrep_ram + j)."""
    mi, hj, h1, h2 = [], [], [], []
    idx = np.arange(nrep_ram, dtype=np.int64)
    for rep in range(nrep_cpu):
        a1, _ = engine.simulate(seed, (rep * 2 + 0) * nrep_ram, nrep_ram)
        a2, _ = engine.simulate(seed, (rep * 2 + 1) * nrep_ram, nrep_ram)
        r = engine.mi_pairs(a1, idx, idx, a2, nalpha)
        mi.append(r["mi"])
        hj.append(r["hjoint"])
    return dict(mi=np.concatenate(mi), hjoint=np.concatenate(hj))
