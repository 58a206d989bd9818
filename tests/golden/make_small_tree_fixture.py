#!/usr/bin/env python3
"""Known-answer vectors for small random trees, computed in 40-digit arithmetic (mpmath) from the DEFINITIONS, not from
the oracle's formulas ("golden fixtures to commit (ii)"):

  * P_c,b = exp(Q r_c t_b) from the symmetrised generator's eigen-system in 40 digits (checked against mpmath.expm);
  * the joint substitution count J_c,b(x, y) = integral_0^t [exp(Q' s) B exp(Q' (t - s))]_xy ds with Q' = r_c Q, B = the
    off-diagonal part of Q' (total register) or Q' o W (weighted), by Gauss-Legendre quadrature of the matrix-valued
    integrand (entire function: 48 nodes are far past 40 digits for ||Q' t|| < 10) -- the oracle instead uses the
    eigen-decomposition closed form with expm1 (uniformization and decomposition agree with it);
  * Felsenstein pruning down and up, site likelihood, posterior rate, rate class, counts
    n(b, i) = sum_c p_c sum_xy U_b(x) J_c,b(x, y) D_b(y) / L_i, norms.

Writes tests/golden/small_trees.npz: the inputs and the float64 roundings of the 40-digit results, for S = 4 and S = 20
(7 taxa, 12 sites, one unknown symbol, Gamma(4) rates, the protein case also with a weighted register).
tests/test_golden_small_trees.py checks the oracle against it on the CPU and the HIP path on the GPU.

    python tests/golden/make_small_tree_fixture.py        # ~15 minutes"""
import os
import sys

import mpmath as mp
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from comap_amd import protein_models as pm, synthetic as sy  # noqa: E402

mp.mp.dps = 40
# Gauss-Legendre nodes on [-1, 1], ascending (48 of them): node k and node n - 1 - k are mirror images
NODES, WEIGHTS = zip(*sorted((x, w) for x, w in mp.calculus.quadrature.GaussLegendre(mp.mp).calc_nodes(5, mp.mp.prec)))


def mpm(a):
    return mp.matrix([[mp.mpf(float(v)) for v in row] for row in np.asarray(a, dtype=np.float64)])


class Expo:
    """s -> exp(Q s) for a generator that is reversible w.r.t. pi: the symmetrised generator's eigen-system in 40 digits
    (mpmath.expm needs half a second per 20 x 20 matrix; this needs one product)"""

    def __init__(self, Qm, pim):
        S = Qm.rows
        sq = [mp.sqrt(p) for p in pim]
        A = mp.matrix(S, S)
        for x in range(S):
            for y in range(S):
                A[x, y] = sq[x] * Qm[x, y] / sq[y]
        A = (A + A.T) / 2
        lam, V = mp.eigsy(A)
        self.lam = [lam[k] for k in range(S)]
        self.L = mp.matrix(S, S)      # D^-1/2 V
        self.R = mp.matrix(S, S)      # V^T D^1/2
        for x in range(S):
            for k in range(S):
                self.L[x, k] = V[x, k] / sq[x]
                self.R[k, x] = V[x, k] * sq[x]

    def __call__(self, s):
        S = self.L.rows
        Ls = mp.matrix(S, S)
        e = [mp.exp(l * s) for l in self.lam]
        for x in range(S):
            for k in range(S):
                Ls[x, k] = self.L[x, k] * e[k]
        return Ls * self.R


def joint_counts(expo, Bm, t):
    """integral_0^t exp(Q s) Bm exp(Q (t - s)) ds"""
    S = Bm.rows
    J = mp.zeros(S, S)
    half = mp.mpf(t) / 2
    E = [expo(half * (x + 1)) for x in NODES]          # exp(Q s_k); t - s_k = s_(n-1-k)
    n = len(NODES)
    for k in range(n):
        J += (WEIGHTS[k] * half) * (E[k] * Bm * E[n - 1 - k])
    return J


def build(S, ntaxa, nsites, seed, weighted):
    parent, blen, lot = sy.random_tree(ntaxa, seed)
    mdl = sy.protein_model(0.7, 4) if S == 20 else sy.dna_model(0.7, 4)
    Q, pi, rates, probs = mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"]
    rng = np.random.default_rng(seed)
    base = rng.integers(0, S, size=(1, nsites))
    aln = np.where(rng.random((ntaxa, nsites)) < 0.5, base, rng.integers(0, S, size=(ntaxa, nsites))).astype(np.uint8)
    aln[1, 3] = S                                             # an unknown: compatible with every state
    W = pm.grantham_distance() if weighted else None
    nn, B, C = len(parent), len(parent) - 1, len(rates)
    root = nn - 1
    Qm, pim = mpm(Q), [mp.mpf(float(v)) for v in pi]
    P = [[None] * B for _ in range(C)]
    J = [[None] * B for _ in range(C)]
    for c in range(C):
        Qr = Qm * mp.mpf(float(rates[c]))
        Bm = mp.matrix(S, S)
        for x in range(S):
            for y in range(S):
                if x != y:
                    Bm[x, y] = Qr[x, y] * (mp.mpf(float(W[x, y])) if weighted else 1)
        expo = Expo(Qr, pim)
        if c == 0:   # the eigen-system against mpmath's own matrix exponential, once
            chk = expo(mp.mpf("0.37")) - mp.expm(Qr * mp.mpf("0.37"))
            assert max(abs(chk[x, y]) for x in range(S) for y in range(S)) < mp.mpf(10) ** -13   # (Q in float64 is reversible to rounding only: the symmetrised generator differs from it by ~1e-17)
        for b in range(B):
            t = float(blen[b])
            P[c][b] = expo(mp.mpf(t))
            J[c][b] = joint_counts(expo, Bm, t)
        print(f"  S={S} class {c} operators done", flush=True)
    children = [[] for _ in range(nn)]
    for i in range(nn - 1):
        children[parent[i]].append(i)
    taxon_of = {int(n): t for t, n in enumerate(lot)}
    counts = np.zeros((nsites, B))
    logL, prate, rclass = np.zeros(nsites), np.zeros(nsites), np.zeros(nsites, dtype=np.int32)
    for i in range(nsites):
        Lc, cnt = [], [[mp.mpf(0)] * B for _ in range(C)]
        for c in range(C):
            D, M = [None] * nn, [None] * nn      # D: below the node; M: message of the branch above the node to its father
            for n in range(nn):                  # post-order: children before fathers
                if not children[n]:
                    code = int(aln[taxon_of[n], i])
                    D[n] = [mp.mpf(1) if (code >= S or x == code) else mp.mpf(0) for x in range(S)]
                else:
                    D[n] = [mp.mpf(1)] * S
                    for e in children[n]:
                        D[n] = [D[n][x] * M[e][x] for x in range(S)]
                if n != root:
                    M[n] = [mp.fsum(P[c][n][x, z] * D[n][z] for z in range(S)) for x in range(S)]
            Lc.append(mp.fsum(pim[x] * D[root][x] for x in range(S)))
            U = [None] * nn                      # U[n]: everything except the subtree of n, as a function of the father's state
            Up = [None] * nn                     # Up[n]: the same seen from n itself (after the branch)
            Up[root] = pim
            for f in range(nn - 1, -1, -1):
                for n in children[f]:
                    u = list(Up[f])
                    for m in children[f]:
                        if m != n:
                            u = [u[x] * M[m][x] for x in range(S)]
                    U[n] = u
                    Up[n] = [mp.fsum(P[c][n][x, z] * u[x] for x in range(S)) for z in range(S)]
            for b in range(B):
                cnt[c][b] = mp.fsum(U[b][x] * J[c][b][x, y] * D[b][y] for x in range(S) for y in range(S))
        pr = [mp.mpf(float(p)) for p in probs]
        L = mp.fsum(pr[c] * Lc[c] for c in range(C))
        logL[i] = float(mp.log(L))
        prate[i] = float(mp.fsum(mp.mpf(float(rates[c])) * pr[c] * Lc[c] for c in range(C)) / L)
        rclass[i] = int(np.argmax([float(pr[c] * Lc[c]) for c in range(C)]))
        for b in range(B):
            counts[i, b] = float(mp.fsum(pr[c] * cnt[c][b] for c in range(C)) / L)
    norm = np.sqrt((counts ** 2).sum(axis=1))
    tag = f"s{S}" + ("w" if weighted else "")
    return {f"{tag}_{k}": v for k, v in dict(parent=parent, blen=blen, lot=lot, Q=Q, pi=pi, rates=rates, probs=probs, aln=aln,
                                            counts=counts, logL=logL, post_rate=prate, rate_class=rclass, norm=norm,
                                            W=(W if weighted else np.zeros((S, S)))).items()}


out = {}
out.update(build(4, 7, 12, 31, False))
out.update(build(20, 7, 12, 32, False))
out.update(build(20, 7, 6, 33, True))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "small_trees.npz"), **out)
print("wrote tests/golden/small_trees.npz", {k: np.asarray(v).shape for k, v in out.items() if k.endswith("counts")})
