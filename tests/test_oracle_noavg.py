"""nijt.average = no, nijt.joint = yes (LegacySubstitutionMappingTools::computeSubstitutionVectorsNoAveraging, called at
CoMap/CoETools.cpp:401 and CoMap/AnalysisTools.cpp:605).  bpp-phyl is not in the reference tree, so the oracle's
restatement (oracle.c orc_map_sites_noavg) is pinned here against the DEFINITION it implements: the most probable pair
of states at the two ends of a branch, found by brute force over all ancestral assignments of a small tree."""
import itertools

import numpy as np
import scipy.linalg

import oracle
from comap_amd import synthetic as sy


def _brute_pairs(parent, blen, lot, Q, pi, rates, probs, aln):
    """[N, B] argmax over (x, y) of sum_c p_c P(data, father = x, node = y | c), by enumeration of the internal nodes"""
    nn, S = len(parent), len(pi)
    T, N = aln.shape
    root = nn - 1
    leaves = {int(lot[t]): t for t in range(T)}
    internal = [n for n in range(nn) if n not in leaves]
    out = np.zeros((N, nn - 1), dtype=np.int64)
    P = [[scipy.linalg.expm(Q * blen[b] * r) for b in range(nn - 1)] for r in rates]
    for i in range(N):
        joint = np.zeros((nn - 1, S, S))
        for states in itertools.product(range(S), repeat=len(internal)):
            st = dict(zip(internal, states))
            for n, t in leaves.items():
                st[n] = int(aln[t, i])
            for c, pc in enumerate(probs):
                w = pi[st[root]] * pc
                for b in range(nn - 1):
                    w *= P[c][b][st[int(parent[b])], st[b]]
                for b in range(nn - 1):
                    joint[b, st[int(parent[b])], st[b]] += w
        out[i] = joint.reshape(nn - 1, -1).argmax(axis=1)       # first maximum in row-major order, as whichMax
    return out


def test_noavg_is_the_most_probable_pair_of_ancestral_states():
    rng = np.random.default_rng(3)
    parent, blen, lot = sy.random_tree(5, 9)
    mdl = sy.dna_model(0.7, 3)
    om = oracle.Model(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
    aln = rng.integers(0, 4, size=(5, 12)).astype(np.uint8)
    aln[:, :4] = aln[:1, :4]                                # a few conserved sites
    r = oracle.map_sites_noavg(om, aln)
    ref = _brute_pairs(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], aln)
    clear = r["margin"] > 1e-9
    assert clear.mean() > 0.9
    assert np.array_equal(r["argmax"][clear], ref[clear])


def test_noavg_counts_are_table_entries_and_norm_follows():
    parent, blen, lot = sy.random_tree(7, 4)
    mdl = sy.protein_model(0.5, 4)
    om = oracle.Model(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
    aln, _ = oracle.simulate(om, 5, 0, 30)
    r = oracle.map_sites_noavg(om, aln)
    avg = oracle.map_sites(om, aln)
    B = om.B
    # a site whose pair is (x, x) on every branch maps to zero substitutions; conserved columns do that
    same = (r["argmax"] // om.S) == (r["argmax"] % om.S)
    assert np.all(r["counts"][same] < 0.5)                  # expected substitutions given x -> x on a short branch
    assert np.allclose(r["norm"], np.sqrt((r["counts"].sum(axis=2) ** 2).sum(axis=1)))
    # the no-averaging count of a branch is an entry of N(t_b): bounded, unlike nothing in particular -- and on
    # average it tracks the averaged mapping
    assert abs(r["counts"].sum() - avg["counts"].sum()) / avg["counts"].sum() < 0.35
    assert r["counts"].shape == (30, B, om.K)
