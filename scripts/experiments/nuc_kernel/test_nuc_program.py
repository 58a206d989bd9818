"""Host logic without a GPU: the program of the nucleotide mapping kernel (comap_amd/csrc/cmx_nuc.h) -- the tree cut into
blocks of at most NB internal nodes, one linear stream of visit packets, operators, leaf tables -- is built and
self-checked by cmx_debug_nuc_program for many tree shapes and block capacities.  The self-check (verify_nuc_program)
executes the stream in plain doubles exactly as the device does (block slots, block-root messages through "HBM", count
rows accumulated over the rate classes, leaf masks read a block ahead, and every read the device issues one packet
ahead DONE one packet ahead) for a site with unknown and ambiguous symbols in it and compares likelihood and every count
with a direct pruning computation on the original (not binarised) tree."""
import numpy as np
import pytest

from comap_amd import engine, synthetic as sy
from test_traversal_program import _random_multifurcating


def _model(ncat=4):
    m = sy.dna_model(0.5, ncat)
    return m["Q"], m["pi"], m["rates"], m["probs"]


@pytest.mark.parametrize("ntaxa", [3, 4, 5, 9, 33, 64, 256, 320])
@pytest.mark.parametrize("nb", [2, 3, 6, 10, 13])
def test_binary_trees(ntaxa, nb):
    Q, pi, rates, probs = _model()
    parent, blen, lot = sy.random_tree(ntaxa, 2000 + ntaxa)
    d = engine.debug_nuc_program(parent, blen, lot, Q, pi, rates, probs, block_capacity=nb)
    ninternal = len(parent) - ntaxa + 1          # + the pseudo node that splits the trifurcating root
    # packets: 4 classes x (phase 1: every node; phase 2: recomputation of all but the block roots + outside visits) and
    # one leaf list per block and phase (the root block's masks survive from phase 1 to phase 2)
    assert d["packets"] == 4 * (3 * ninternal - d["blocks"]) + 2 * d["blocks"] - 1
    assert d["blocks"] == d["roots"] + 1 and d["blocks"] >= -(-ninternal // nb)
    # greedy bottom-up cut: blocks are at least about half full on these trees
    assert d["blocks"] <= max(1, 2 * -(-ninternal // nb) + 1)
    assert d["operators"] == 2 * (len(parent) - 1)
    # every block root but the tree's goes to HBM once per pass (message) and once (outside message)
    assert d["root_stores"] == 2 * d["roots"]


def test_weighted_types_and_class_counts():
    parent, blen, lot = sy.random_tree(40, 7)
    for ncat in (1, 2, 4, 5, 8):
        Q, pi, rates, probs = _model(ncat)
        W = sy.compensation_weights_dna()
        Bk = np.stack([sy.weighted_register(Q, W), sy.weighted_register(Q), sy.weighted_register(Q, np.abs(W))])
        d = engine.debug_nuc_program(parent, blen, lot, Q, pi, rates, probs, Bk=Bk, block_capacity=6)
        assert d["operators"] == 4 * (len(parent) - 1)


def test_caterpillar_and_star():
    Q, pi, rates, probs = _model()
    n = 40
    parent = np.full(2 * n - 2, -1, dtype=np.int32)
    lot = np.zeros(n, dtype=np.int32)
    lot[0], lot[1] = 0, 1
    idx = 2
    cur = idx
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
    blen = np.full(2 * n - 2, 0.05)
    for nb in (2, 5, 10):
        d = engine.debug_nuc_program(parent, blen, lot, Q, pi, rates, probs, block_capacity=nb)
        assert d["blocks"] == -(-(n - 1) // nb)          # a chain cuts into full blocks
    # star tree: a chain of pseudo nodes under the root, no internal operator at all
    parent = np.array([n] * n + [-1], dtype=np.int32)
    d = engine.debug_nuc_program(parent, np.full(n + 1, 0.1), np.arange(n, dtype=np.int32), Q, pi, rates, probs, block_capacity=10)
    assert d["applies"] == 0 and d["leaf_gathers"] > 3 * n


@pytest.mark.parametrize("seed", range(10))
def test_random_multifurcating_trees(seed):
    Q, pi, rates, probs = _model()
    ntaxa = int(np.random.default_rng(seed).integers(4, 90))
    parent, blen, lot = _random_multifurcating(ntaxa, seed)
    for nb in (3, 10):
        engine.debug_nuc_program(parent, blen, lot, Q, pi, rates, probs, block_capacity=nb)


def test_rejects_protein_models_and_bad_capacity():
    parent, blen, lot = sy.random_tree(8, 3)
    m = sy.protein_model(0.5, 4)
    with pytest.raises(engine.CmxError):
        engine.debug_nuc_program(parent, blen, lot, m["Q"], m["pi"], m["rates"], m["probs"])
    Q, pi, rates, probs = _model()
    with pytest.raises(engine.CmxError):
        engine.debug_nuc_program(parent, blen, lot, Q, pi, rates, probs, block_capacity=14)
