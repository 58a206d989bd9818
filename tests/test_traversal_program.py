"""Host logic without a GPU: what the mapping kernel's tree walk reads (node records, operator stream, workspace-load
schedule) is compiled and self-checked by cmx_debug_walk for many tree shapes -- binary, caterpillar, star, random
multifurcations.  The self-check (verify_walk, cmx_host_model.cpp) runs the SAME walk (cmx_walk.h) numerically on the
host, from the device matrix layouts and through the recorded operator stream, and compares site likelihood and every
joint count with a direct pruning computation on the original (not binarised) tree."""
import numpy as np
import pytest

from comap_amd import engine, synthetic as sy


def _model():
    m = sy.protein_model(0.5, 4)
    return m["Q"], m["pi"], m["rates"], m["probs"]


def _random_multifurcating(ntaxa, seed):
    """post-order parent array of a random tree whose internal nodes have 2..5 children (root >= 3)"""
    rng = np.random.default_rng(seed)
    # grow by merging random groups of subtrees until one root remains
    roots, parent, blen = list(range(ntaxa)), {}, {}
    nxt = ntaxa
    while len(roots) > 1:
        k = int(min(len(roots), rng.integers(2, 6)))
        if len(roots) - k == 1:        # avoid a final binary root: merge everything that is left
            k = len(roots)
        pick = list(rng.choice(len(roots), size=k, replace=False))
        kids = [roots[i] for i in pick]
        for c in kids:
            parent[c] = nxt
        roots = [r for i, r in enumerate(roots) if i not in pick] + [nxt]
        nxt += 1
    root = roots[0]
    # relabel in post-order
    children = {}
    for c, p in parent.items():
        children.setdefault(p, []).append(c)
    order = []

    def visit(u):
        for c in children.get(u, []):
            visit(c)
        order.append(u)

    visit(root)
    new = {old: i for i, old in enumerate(order)}
    nn = len(order)
    par = np.full(nn, -1, dtype=np.int32)
    for c, p in parent.items():
        par[new[c]] = new[p]
    bl = rng.exponential(0.1, nn) + 1e-6
    bl[nn - 1] = 0.0
    lot = np.array([new[t] for t in range(ntaxa)], dtype=np.int32)
    return par, bl, lot


@pytest.mark.parametrize("ntaxa", [3, 4, 5, 9, 33, 64, 257])
def test_binary_trees_compile(ntaxa):
    Q, pi, rates, probs = _model()
    parent, blen, lot = sy.random_tree(ntaxa, 1000 + ntaxa)
    d = engine.debug_walk(parent, blen, lot, Q, pi, rates, probs)
    nn = len(parent)
    ninternal = nn - ntaxa
    # a trifurcating root is split by one pseudo node
    assert d["nrec"].shape[1] == 16 and 1 <= d["nrec"].shape[0] <= ninternal + 1
    ops = d["msched"]
    assert ((ops[:, 1] >= -1) & (ops[:, 1] < ntaxa)).all()
    leaf_ops = ops[ops[:, 1] >= 0]
    assert set(leaf_ops[:, 1]) == set(range(ntaxa))            # every leaf edge is applied
    # per class pass: every leaf needs its transition operator on the way down and up, and its count operator once
    assert len(leaf_ops) >= 3 * ntaxa and len(leaf_ops) == d["leaf_ops"]
    # message scheme: one product per internal branch on the way up, two (count + transposed) on the way down; an
    # inlined cherry's message is rebuilt once more
    assert d["products"] <= 4 * (ninternal - 1)
    # every stored vector is loaded; a message at most twice (by its parent in either pass), an outside message once
    if len(d["ldsched"]):
        slots, counts = np.unique(d["ldsched"] & 0x40ffffff, return_counts=True)
        assert counts.max() <= 2
        assert d["loads"] == len(d["ldsched"])


def test_caterpillar_and_star():
    Q, pi, rates, probs = _model()
    n = 20
    # caterpillar: ((((a,b),c),d),...) with a trifurcating root
    parent = np.full(2 * n - 2, -1, dtype=np.int32)
    lot = np.zeros(n, dtype=np.int32)
    # nodes: leaves interleaved in post-order: a, b, i1, c, i2, d, ..., last two leaves hang off the root
    idx, last = 0, None
    lot[0], lot[1] = 0, 1
    idx = 2
    cur = idx          # first internal node
    parent[0] = parent[1] = cur
    idx += 1
    for t in range(2, n - 2):
        lot[t] = idx
        leaf = idx
        idx += 1
        parent[cur] = parent[leaf] = idx
        cur = idx
        idx += 1
    root = 2 * n - 3
    for t in (n - 2, n - 1):
        lot[t] = idx
        parent[idx] = root
        idx += 1
    parent[cur] = root
    assert idx == root
    blen = np.full(2 * n - 2, 0.05)
    engine.debug_walk(parent, blen, lot, Q, pi, rates, probs)
    # star tree: all leaves on the root -> a chain of n - 2 pseudo nodes under a binary root, no operator for any of them
    parent = np.array([n] * n + [-1], dtype=np.int32)
    d = engine.debug_walk(parent, np.full(n + 1, 0.1), np.arange(n, dtype=np.int32), Q, pi, rates, probs)
    assert d["nrec"].shape[0] == n - 1 and d["products"] == 0


@pytest.mark.parametrize("seed", range(12))
def test_random_multifurcating_trees_compile(seed):
    Q, pi, rates, probs = _model()
    ntaxa = int(np.random.default_rng(seed).integers(4, 70))
    parent, blen, lot = _random_multifurcating(ntaxa, seed)
    d = engine.debug_walk(parent, blen, lot, Q, pi, rates, probs)
    ops = d["msched"]
    assert set(ops[ops[:, 1] >= 0][:, 1]) == set(range(ntaxa))


def test_malformed_trees_are_rejected():
    Q, pi, rates, probs = _model()
    parent, blen, lot = sy.random_tree(8, 3)
    bad = parent.copy()
    bad[0], bad[1] = bad[1], bad[0] if bad[0] != bad[1] else bad[0]
    bad2 = parent.copy()
    bad2[2] = 1                          # parent id below the child id: not post-order
    with pytest.raises(engine.CmxError):
        engine.debug_walk(bad2, blen, lot, Q, pi, rates, probs)
    with pytest.raises(engine.CmxError):
        engine.debug_walk(parent, -blen, lot, Q, pi, rates, probs)


def _shape(kind, ntaxa, seed):
    if kind == "random":
        return sy.random_tree(ntaxa, seed)
    if kind == "multifurcating":
        return _random_multifurcating(ntaxa, seed)
    rng = np.random.default_rng(seed)
    if kind == "caterpillar":
        nn = 2 * ntaxa - 1
        par = np.full(nn, -1, dtype=np.int32)
        lot = np.zeros(ntaxa, dtype=np.int32)
        lot[0], lot[1] = 0, 1
        par[0] = par[1] = 2
        cur = 2
        for t in range(2, ntaxa):
            lot[t] = cur + 1
            par[cur] = par[cur + 1] = cur + 2
            cur += 2
    else:   # balanced: every leaf in a cherry
        level, parent, nxt = list(range(ntaxa)), {}, ntaxa
        while len(level) > 1:
            up = []
            for k in range(0, len(level) - 1, 2):
                parent[level[k]] = parent[level[k + 1]] = nxt
                up.append(nxt)
                nxt += 1
            if len(level) % 2:
                up.append(level[-1])
            level = up
        kids = {}
        for c, p in parent.items():
            kids.setdefault(p, []).append(c)
        order = []

        def visit(n):
            for c in kids.get(n, []):
                visit(c)
            order.append(n)
        visit(level[0])
        new = {old: i for i, old in enumerate(order)}
        par = np.full(nxt, -1, dtype=np.int32)
        for c, p in parent.items():
            par[new[c]] = new[p]
        lot = np.array([new[t] for t in range(ntaxa)], dtype=np.int32)
    blen = np.maximum(rng.exponential(0.1, size=len(par)), 1e-6)
    blen[-1] = 0.0
    return par, blen, lot


@pytest.mark.parametrize("kind,ntaxa", [("random", 5), ("random", 40), ("random", 256), ("multifurcating", 30), ("caterpillar", 12),
                                        ("balanced", 16), ("balanced", 21), ("random", 3)])
@pytest.mark.parametrize("ncat,ntypes", [(4, 1), (5, 1), (4, 2), (8, 1)])
def test_cherry_table_walk_compiles_and_checks(kind, ntaxa, ncat, ntypes):
    """class-fused nucleotide models (>= 4 rate classes): the walk of resolved alignments takes inlined cherries from tables
    (cmx_walk.h).  cmx_debug_walk fails unless BOTH walks reproduce likelihood and every joint count of a direct pruning
    computation from the device layouts (verify_walk), so a call that returns has checked the tables numerically."""
    parent, blen, lot = _shape(kind, ntaxa, 31 * ntaxa + ncat)
    m = sy.dna_model(0.6, ncat)
    Bk = None
    if ntypes == 2:
        rng = np.random.default_rng(ntaxa)
        Bk = np.stack([sy.weighted_register(m["Q"], rng.uniform(-1, 1, size=(4, 4))), sy.weighted_register(m["Q"])])
    w = engine.debug_walk(parent, blen, lot, m["Q"], m["pi"], m["rates"], m["probs"], Bk=Bk)
    # a cherry with tables: its message is 1 op instead of 3, its outside visit 3 K instead of 5 + 3 K
    nch = w["cherry_tables"]
    assert w["products_tables"] + w["leaf_ops_tables"] == w["products"] + w["leaf_ops"] - nch * 2 * 2 - nch * 5 or nch == 0 or ntypes == 2
    assert w["products_tables"] <= w["products"] and w["leaf_ops_tables"] <= w["leaf_ops"]
    if kind == "balanced" and ntaxa == 16:
        assert nch >= 6
    if kind == "random" and ntaxa == 3:
        assert nch == 0 and w["products_tables"] == w["products"]      # no cherry: the second stream is the first


def test_fewer_than_four_classes_have_no_table_walk():
    parent, blen, lot = sy.random_tree(20, 5)
    m = sy.dna_model(0.6, 3)
    w = engine.debug_walk(parent, blen, lot, m["Q"], m["pi"], m["rates"], m["probs"])
    assert w["cherry_tables"] == 0 and w["products_tables"] == 0
