import os
import sys

import numpy as np
import pytest

try:
    # torch bundles its own HIP runtime; it has to be the first one this process loads, or a test that later moves a
    # tensor to the device finds "no HIP GPUs" (the engine library then binds to the runtime torch brought, as in bench.py)
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_close(a, b, rtol=1e-6, atol=0.0):
    """max relative error with the reference value in the denominator; NaN must match NaN."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), "NaN pattern differs"
    ok = ~nan_a
    err = np.abs(a[ok] - b[ok])
    tol = rtol * np.abs(b[ok]) + atol
    bad = err > tol
    assert not bad.any(), f"{bad.sum()} of {bad.size} entries differ; worst rel {np.max(err / np.maximum(np.abs(b[ok]), 1e-300)):.3e}"


def make_case(ntaxa, nsites, nstates, seed, alpha=0.5, ncat=4):
    """Random tree + model + alignment simulated by the ORACLE (test infrastructure) for parity tests."""
    import oracle
    from comap_amd import synthetic as sy
    parent, blen, lot = sy.random_tree(ntaxa, seed)
    mdl = sy.protein_model(alpha, ncat) if nstates == 20 else sy.dna_model(alpha, ncat)
    om = oracle.Model(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
    aln, _ = oracle.simulate(om, seed + 1, 0, nsites)
    return dict(parent=parent, blen=blen, lot=lot, aln=aln, **mdl)


@pytest.fixture(scope="session")
def myo():
    return np.load(os.path.join(ROOT, "tests", "golden", "myoglobin.npz"))
