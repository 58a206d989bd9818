"""Known answers in 40-digit arithmetic for small random trees (tests/golden/small_trees.npz, generated from the
DEFINITIONS by tests/golden/make_small_tree_fixture.py: matrix exponentials, the substitution-count integral by
quadrature, pruning): the oracle on the CPU, the HIP path on the GPU.  This pins the arithmetic of oracle and engine to
the mathematics; what pins them to the REFERENCE are the Myoglobin fixtures (tests/test_golden_myoglobin.py)."""
import os

import numpy as np
import pytest

import oracle
from comap_amd import engine
from conftest import ROOT, rel_close

CASES = ["s4", "s20", "s20w"]


@pytest.fixture(scope="module")
def small():
    return np.load(os.path.join(ROOT, "tests", "golden", "small_trees.npz"))


def _get(small, tag):
    return {k[len(tag) + 1:]: small[k] for k in small.files if k.startswith(tag + "_")}


def _register(c):
    """Q o W off the diagonal (weighted register) or None (total register)"""
    if not c["W"].any():
        return None
    B = c["Q"] * c["W"]
    np.fill_diagonal(B, 0.0)
    return B[None]


def _check(r, c, rtol):
    rel_close(r["counts"][:, :, 0], c["counts"], rtol, 1e-300)
    rel_close(r["logL"], c["logL"], 1e-12)
    rel_close(r["post_rate"], c["post_rate"], 1e-12)
    rel_close(r["norm"], c["norm"], rtol)
    assert np.array_equal(r["rate_class"], c["rate_class"])


@pytest.mark.parametrize("tag", CASES)
@pytest.mark.parametrize("method", [oracle.METHOD_UNIF, oracle.METHOD_DECOMP])
def test_oracle_matches_high_precision_known_answers(small, tag, method):
    c = _get(small, tag)
    Bk = _register(c)
    m = oracle.Model(c["parent"], c["blen"], c["lot"], c["Q"], c["pi"], c["rates"], c["probs"], Bk=Bk, method=method,
                     nonneg=Bk is None)
    _check(oracle.map_sites(m, c["aln"]), c, 1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_device_matches_high_precision_known_answers(small, tag):
    c = _get(small, tag)
    Bk = _register(c)
    eng = engine.Engine(c["parent"], c["blen"], c["lot"], c["Q"], c["pi"], c["rates"], c["probs"], Bk=Bk,
                        clamp_negative=Bk is None)
    _check(eng.map_sites(c["aln"]), c, 1e-9)
