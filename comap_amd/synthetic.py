"""Seeded synthetic inputs of the benchmark configurations.

Host-side helpers only (numpy); nothing here is on the device path."""
import numpy as np

from . import protein_models as pm


def random_tree(ntaxa, seed, mean_blen=0.1, min_blen=1e-6):
    """Random unrooted binary topology by sequential random leaf attachment, branch lengths ~ Exp(mean) floored
    at min_blen.  Returns (parent, blen, leaf_of_taxon) with nodes in post-order, root (trifurcation) last."""
    rng = np.random.default_rng(seed)
    # adjacency as child lists from a trifurcating root 0 with leaves 1,2,3 (temporary ids)
    children = {0: [1, 2, 3], 1: [], 2: [], 3: []}
    parent = {1: 0, 2: 0, 3: 0}
    taxon = {1: 0, 2: 1, 3: 2}
    nxt = 4
    for t in range(3, ntaxa):
        edges = sorted(parent.keys())
        v = edges[rng.integers(len(edges))]      # split the branch above v
        u = parent[v]
        mid, leaf = nxt, nxt + 1
        nxt += 2
        children[u][children[u].index(v)] = mid
        children[mid] = [v, leaf]
        children[leaf] = []
        parent[mid] = u
        parent[v] = mid
        parent[leaf] = mid
        taxon[leaf] = t
    order = []

    def visit(u):
        stack = [(u, 0)]
        while stack:
            node, i = stack.pop()
            if i < len(children[node]):
                stack.append((node, i + 1))
                stack.append((children[node][i], 0))
            else:
                order.append(node)

    visit(0)
    remap = {old: new for new, old in enumerate(order)}
    nn = len(order)
    par = np.full(nn, -1, dtype=np.int32)
    for old in order:
        if old != 0:
            par[remap[old]] = remap[parent[old]]
    blen = np.maximum(rng.exponential(mean_blen, size=nn), min_blen)
    blen[nn - 1] = 0.0
    lot = np.zeros(ntaxa, dtype=np.int32)
    for leaf, t in taxon.items():
        lot[t] = remap[leaf]
    return par, blen, lot


def protein_model(alpha=0.5, ncat=4):
    Q, pi = pm.jtt92()
    rates, probs = pm.gamma_rates(alpha, ncat)
    return dict(Q=Q, pi=pi, rates=rates, probs=probs)


def dna_model(alpha=0.5, ncat=4):
    # GTR parameters of examples/RNA/BacteriaSSU/options.comap:42 (a, b, c, d, e, theta, theta1, theta2)
    Q, pi = pm.gtr(a=1.595119085705, b=0.551507085060, c=0.350972557796, d=0.304670173544, e=0.282819006597,
                   theta=0.523619444641, theta1=0.512962941602, theta2=0.585047306118)
    rates, probs = pm.gamma_rates(alpha, ncat)
    return dict(Q=Q, pi=pi, rates=rates, probs=probs)


def compensation_weights_dna():
    """Non-symmetric weights W(x,y) = idx[y] - idx[x] (cfg 4)."""
    idx = np.array([-1.5, -0.5, 0.5, 1.5])
    return idx[None, :] - idx[:, None]


def weighted_register(Q, W=None):
    B = np.array(Q, dtype=np.float64, copy=True)
    np.fill_diagonal(B, 0.0)
    if W is not None:
        B = B * W
    return B
