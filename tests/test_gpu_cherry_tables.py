"""Cherry tables (cmx_walk.h): on the null's resolved alignments a class-fused nucleotide model takes an inlined
cherry's message and its whole outside visit from tables indexed by its two symbols.  The null kernel (table walk) and the
observed kernel (generic walk: leaf gathers and products) must give the same substitution vectors for the same columns --
checked here at 1e-11, far below the 1e-6 of the oracle comparisons -- for tree shapes from "every leaf in a cherry" to
"no cherry at all", four and five rate classes (16- and 20-state fused layouts), one and two substitution types."""
import numpy as np
import pytest

import oracle
from comap_amd import engine, synthetic
from conftest import rel_close

pytestmark = pytest.mark.gpu


def _balanced(ntaxa):
    """perfectly balanced rooted binary tree, leaves first (post-order with the root last): every leaf sits in a cherry"""
    nodes, parent = list(range(ntaxa)), {}
    nxt = ntaxa
    level = nodes
    while len(level) > 1:
        up = []
        for k in range(0, len(level) - 1, 2):
            parent[level[k]] = parent[level[k + 1]] = nxt
            up.append(nxt)
            nxt += 1
        if len(level) % 2:
            up.append(level[-1])
        level = up
    nn = nxt
    # renumber in post-order (children before parents, root last)
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
    par = np.full(nn, -1, dtype=np.int32)
    for c, p in parent.items():
        par[new[c]] = new[p]
    lot = np.array([new[t] for t in range(ntaxa)], dtype=np.int32)
    return par, lot


def _caterpillar(ntaxa):
    """((((t0, t1), t2), t3) ...): one cherry at the bottom, every other leaf pendant"""
    nn = 2 * ntaxa - 1
    par = np.full(nn, -1, dtype=np.int32)
    lot = np.zeros(ntaxa, dtype=np.int32)
    # post-order: t0, t1, i0, t2, i1, t3, i2, ...
    lot[0], lot[1] = 0, 1
    par[0] = par[1] = 2
    cur = 2
    for t in range(2, ntaxa):
        leaf, inner = cur + 1, cur + 2
        lot[t] = leaf
        par[cur] = par[leaf] = inner
        cur = inner
    return par, lot


@pytest.mark.parametrize("shape,ntaxa", [("balanced", 16), ("balanced", 13), ("caterpillar", 9), ("random", 40)])
@pytest.mark.parametrize("ncat,ntypes", [(4, 1), (5, 1), (4, 2)])
def test_table_walk_equals_the_generic_walk(shape, ntaxa, ncat, ntypes):
    rng = np.random.default_rng(ntaxa * 10 + ncat + ntypes)
    if shape == "random":
        parent, blen, lot = synthetic.random_tree(ntaxa, 77)
    else:
        parent, lot = _balanced(ntaxa) if shape == "balanced" else _caterpillar(ntaxa)
        blen = np.maximum(rng.exponential(0.1, size=len(parent)), 1e-6)
        blen[-1] = 0.0
    mdl = synthetic.dna_model(0.7, ncat)
    kw = {}
    if ntypes == 2:
        W1 = rng.uniform(-1, 1, size=(4, 4))
        kw = dict(Bk=np.stack([synthetic.weighted_register(mdl["Q"], W1), synthetic.weighted_register(mdl["Q"], np.abs(W1))]), clamp_negative=False)
    eng = engine.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], **kw)
    info = eng.info()
    assert info["device_states"] == 4 * (4 if ncat == 4 else 5)
    if shape == "balanced" and ntaxa == 16:
        assert info["cherry_tables"] >= 6                      # (the two cherries under the root's children may be visited)
        assert info["products_per_pass_null"] < info["products_per_pass"] and info["leaf_ops_per_pass_null"] < info["leaf_ops_per_pass"]
    nrep, ram = 3, 53
    sup = np.stack([np.stack([eng.simulate(5, (r * 2 + h) * ram, ram)[0] for h in range(2)]) for r in range(nrep)])
    nl = eng.null_intra(engine.STAT_CORRELATION if ntypes == 1 else engine.STAT_COMPENSATION, 0, 0, nrep, ram, supplied=sup)   # the table walk
    for r in range(nrep):
        m0, m1 = eng.map_sites(sup[r, 0]), eng.map_sites(sup[r, 1])                                                    # the generic walk
        sl = slice(r * ram, (r + 1) * ram)
        rel_close(nl["nmin"][sl], np.minimum(m0["norm"], m1["norm"]), 1e-11, 1e-300)
        rel_close(nl["prmin"][sl], np.minimum(m0["post_rate"], m1["post_rate"]), 1e-12)
        assert np.array_equal(nl["rcmin"][sl], np.minimum(m0["rate_class"], m1["rate_class"]))
        kind = oracle.ST_CORRELATION if ntypes == 1 else oracle.ST_COMPENSATION
        st = np.array([oracle.stat_pair(kind, m0["counts"][j], m1["counts"][j]) for j in range(ram)])
        rel_close(nl["stat"][sl], st, 1e-8, 1e-11)


def test_table_walk_against_the_oracle_with_many_cherries():
    parent, lot = _balanced(32)
    rng = np.random.default_rng(3)
    blen = np.maximum(rng.exponential(0.15, size=len(parent)), 1e-6)
    mdl = synthetic.dna_model(0.5, 4)
    eng = engine.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
    om = oracle.Model(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
    g, o = eng.null_intra(1, 91, 2, 5, 70), oracle.null_intra(om, 1, 91, 2, 5, 70)
    rel_close(g["stat"], o["stat"], 1e-6, 1e-12)
    rel_close(g["nmin"], o["nmin"], 1e-6)
    assert np.array_equal(g["rcmin"], o["rcmin"])
