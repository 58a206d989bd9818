#!/usr/bin/env python3
"""Diagnostic (not product, not test): which entries of the recalled JTT92-DCmut table, if any,
disagree with the reference's Myoglobin fixtures?  Fits log-exchangeabilities by Gauss-Newton on
the relative residuals of Myo_decomp.vec and reports entries that move by more than print noise.
Used once in round 1 to vet comap_amd/protein_models.py; kept for provenance."""
import sys
import time
import numpy as np

sys.path.insert(0, ".")
from oracle import np_oracle as o  # noqa: E402
from comap_amd import protein_models as pm  # noqa: E402

g = np.load("tests/golden/myoglobin.npz")
rates, probs = pm.gamma_rates(float(g["alpha"]), 4)
S0 = pm._lower_to_sym(pm._JTT_DCMUT_LOWER)
f0 = np.array([float(x) for x in pm._JTT_DCMUT_FREQ.split()])
iu = np.tril_indices(20, -1)
v = g["vec_decomp"].T
pos = v > 0


def model(theta):
    S = np.zeros((20, 20))
    S[iu] = S0[iu] * np.exp(theta[:190])
    S = S + S.T
    f = f0 * np.exp(np.concatenate([theta[190:209], [0.0]]))
    return pm.reversible_generator(S, f)


def resid(theta):
    Q, pi = model(theta)
    B = o.rate_matrix_register(Q)
    r = o.map_sites(g["parent"], g["blen"], g["leaf_of_taxon"], g["aln"], g["masks"], Q, pi, rates, probs, [B],
                    method="decomp")
    c = r["counts"][:, :, 0]
    rc = ((c - v) / np.where(pos, v, 1.0))[pos]
    rl = (r["logL"] - g["infos_logl"]) / np.abs(g["infos_logl"])
    rp = (r["post_rate"] - g["infos_pr"]) / g["infos_pr"]
    return np.concatenate([rc, rl, rp])


theta = np.zeros(209)
r0 = resid(theta)
print("initial rms", np.sqrt((r0 ** 2).mean()), "max", np.abs(r0).max())
for it in range(3):
    t = time.time()
    J = np.zeros((len(r0), 209))
    h = 1e-4
    for k in range(209):
        th = theta.copy()
        th[k] += h
        J[:, k] = (resid(th) - r0) / h
    # ridge-regularised GN step (damp directions the data cannot see)
    lam = 1e-10 * np.trace(J.T @ J) / 209
    step = np.linalg.solve(J.T @ J + lam * np.eye(209), -J.T @ r0)
    theta = theta + step
    r0 = resid(theta)
    print(f"iter {it} rms {np.sqrt((r0 ** 2).mean()):.3e} max {np.abs(r0).max():.3e}  ({time.time() - t:.0f}s)")
    np.save("/tmp/jtt_theta.npy", theta)

rel = np.exp(theta[:190]) - 1
order = np.argsort(-np.abs(rel))
print("largest relative changes of exchangeabilities:")
for k in order[:25]:
    i, j = iu[0][k], iu[1][k]
    print(f"  S[{pm.AA_ORDER[i]},{pm.AA_ORDER[j]}] {S0[i, j]:.6f} -> {S0[i, j] * np.exp(theta[k]):.6f}  ({rel[k]:+.2e})")
print("freq changes:", np.exp(theta[190:]) - 1)
