"""nijt.joint = no (LegacySubstitutionMappingTools::computeSubstitutionVectorsMarginal and ...NoAveragingMarginal, called
at CoMap/CoETools.cpp:399-405 and CoMap/AnalysisTools.cpp:598-633).  bpp-phyl is not in the reference tree, so the
oracle's restatement (oracle.c orc_map_sites_marginal) is pinned here against the DEFINITIONS it implements, by brute force
over all ancestral assignments of a small tree:
  * getPosteriorProbabilitiesPerStatePerRate at an internal node = the joint posterior P(class c, state x at the node |
    data);
  * the marginal ancestral state = the first maximum of that posterior summed over the classes;
  * the Marginal count = sum_c sum_xy post_father(c, x) post_node(c, y) N(x, y; r_c t_b), the NoAveragingMarginal count =
    N(x*_father, x*_node; t_b) -- rebuilt here from scipy's expm and a quadrature-free formula for N.
Parity against the reference itself: unpinned (it ships no output of these variants)."""
import itertools

import numpy as np
import scipy.linalg

import oracle
from comap_amd import synthetic as sy


def _brute_posteriors(parent, blen, lot, Q, pi, rates, probs, aln):
    """post[N, nn, C, S] = P(class c, state x at node | data) for internal nodes by enumeration; leaves: e(x) p_c / sum e"""
    nn, S, C = len(parent), len(pi), len(rates)
    T, N = aln.shape
    root = nn - 1
    leaves = {int(lot[t]): t for t in range(T)}
    internal = [n for n in range(nn) if n not in leaves]
    P = [[scipy.linalg.expm(Q * blen[b] * r) for b in range(nn - 1)] for r in rates]
    post = np.zeros((N, nn, C, S))
    for i in range(N):
        for states in itertools.product(range(S), repeat=len(internal)):
            st = dict(zip(internal, states))
            for n, t in leaves.items():
                st[n] = int(aln[t, i])
            for c, pc in enumerate(probs):
                w = pi[st[root]] * pc
                for b in range(nn - 1):
                    w *= P[c][b][st[int(parent[b])], st[b]]
                for n in internal:
                    post[i, n, c, st[n]] += w
        post[i] /= post[i, root].sum()
        for n, t in leaves.items():
            post[i, n] = 0.0
            post[i, n, :, int(aln[t, i])] = probs
    return post


def _conditional_counts(Q, t):
    """N(x, y; t) = E[# substitutions | x at 0, y at t] for the total register: J / P with J = int_0^t e^{Qs} B e^{Q(t-s)} ds
    (B = Q without its diagonal), from the block-matrix exponential identity expm([[Q, B], [0, Q]] t)[0:S, S:2S]"""
    S = Q.shape[0]
    Bm = Q - np.diag(np.diag(Q))
    big = np.zeros((2 * S, 2 * S))
    big[:S, :S] = Q
    big[:S, S:] = Bm
    big[S:, S:] = Q
    E = scipy.linalg.expm(big * t)
    return E[:S, S:] / E[:S, :S]


def _case():
    rng = np.random.default_rng(11)
    parent, blen, lot = sy.random_tree(5, 9)
    blen = np.maximum(blen, 0.02)
    mdl = sy.dna_model(0.7, 3)
    aln = rng.integers(0, 4, size=(5, 14)).astype(np.uint8)
    aln[:, :4] = aln[:1, :4]                                # a few conserved sites
    return parent, blen, lot, mdl, aln


def test_posterior_per_state_per_rate_is_the_joint_posterior():
    parent, blen, lot, mdl, aln = _case()
    om = oracle.Model(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
    r = oracle.map_sites_marginal(om, aln, True, want_post=True)
    ref = _brute_posteriors(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], aln)
    assert np.allclose(r["post"], ref, rtol=1e-9, atol=1e-13)
    assert np.allclose(r["post"].sum(axis=(2, 3)), 1.0)     # a distribution over (class, state) at every node


def test_marginal_counts_follow_the_definition():
    parent, blen, lot, mdl, aln = _case()
    Q, rates, probs = mdl["Q"], mdl["rates"], mdl["probs"]
    om = oracle.Model(parent, blen, lot, Q, mdl["pi"], rates, probs)
    post = _brute_posteriors(parent, blen, lot, Q, mdl["pi"], rates, probs, aln)
    nn = len(parent)
    # average = yes
    r = oracle.map_sites_marginal(om, aln, True)
    want = np.zeros((aln.shape[1], nn - 1))
    for b in range(nn - 1):
        f = int(parent[b])
        for c, rc in enumerate(rates):
            Nc = _conditional_counts(Q, blen[b] * rc)
            want[:, b] += np.einsum("ix,xy,iy->i", post[:, f, c], Nc, post[:, b, c])
    assert np.allclose(r["counts"][:, :, 0], want, rtol=1e-7, atol=1e-12)
    assert np.allclose(r["norm"], np.sqrt((r["counts"].sum(axis=2) ** 2).sum(axis=1)))
    # average = no: marginal ancestral states, then one table entry per branch
    r0 = oracle.map_sites_marginal(om, aln, False)
    marg = post.sum(axis=2)                                  # [N, nn, S]
    states = marg.argmax(axis=2)                             # first maximum, as VectorTools::whichMax
    clear = r0["margin"] > 1e-9
    assert clear.mean() > 0.9 and np.array_equal(r0["anc"][clear], states[clear])
    for b in range(nn - 1):
        N1 = _conditional_counts(Q, blen[b])
        ok = clear[:, b] & clear[:, int(parent[b])]
        assert np.allclose(r0["counts"][ok, b, 0], N1[states[ok, int(parent[b])], states[ok, b]], rtol=1e-7, atol=1e-12)


def test_unknown_leaf_and_protein_shapes():
    parent, blen, lot = sy.random_tree(7, 4)
    mdl = sy.protein_model(0.5, 4)
    om = oracle.Model(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
    aln, _ = oracle.simulate(om, 5, 0, 20)
    aln[2, ::3] = 20                                          # unknowns at one leaf
    r = oracle.map_sites_marginal(om, aln, True, want_post=True)
    leaf = int(lot[2])
    # a leaf with an unknown: e = all ones -> uniform over the states, prior class weights
    assert np.allclose(r["post"][0, leaf], np.outer(mdl["probs"], np.full(20, 1 / 20)))
    assert r["counts"].shape == (20, om.B, om.K) and np.isfinite(r["counts"]).all() and (r["counts"] >= 0).all()
    r0 = oracle.map_sites_marginal(om, aln, False)
    assert (r0["anc"][::3, leaf] == 0).all()                  # first maximum of an all-ones vector
    avg = oracle.map_sites(om, aln)
    # the product of marginals is not the joint, but the total number of mapped substitutions stays in its neighbourhood
    # (about half of it on this tree: 0.52 - 0.55 over alignments of 20 to 200 sites)
    assert abs(r["counts"].sum() - avg["counts"].sum()) / avg["counts"].sum() < 0.7
