"""simulations.continuous = yes (CoMap.cpp:146, 213): the oracle's restatement of the continuous-rate simulator.
Its Gamma quantile is pinned to scipy; the drawn rates have the moments of Gamma(alpha, alpha) mixed with the invariant
class; with all rates equal the simulator reduces to the discrete one on a one-class model."""
import numpy as np
import pytest
from scipy.stats import gamma

import oracle
from conftest import make_case


def test_gamma_quantile_against_scipy():
    for a in (0.2, 0.5, 0.985435, 2.0, 7.5, 40.0):
        for u in (1e-9, 1e-4, 0.01, 0.3, 0.5, 0.9, 0.9999, 1 - 1e-9):
            q, r = oracle.gamma_quantile(a, u), gamma.ppf(u, a)
            assert abs(q - r) <= 1e-10 * r + 1e-300, (a, u, q, r)


def test_rates_have_the_distribution_of_invariant_gamma():
    case = make_case(6, 4, 4, 3)
    om = oracle.Model(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    _, r = oracle.simulate_continuous(om, 5, 0, 40000, 0.7, 0.0)
    assert abs(r.mean() - 1.0) < 0.03 and abs(r.var() - 1 / 0.7) < 0.12      # Gamma(a, a): mean 1, variance 1 / a
    _, ri = oracle.simulate_continuous(om, 5, 0, 40000, 0.7, 0.25)
    assert abs((ri == 0).mean() - 0.25) < 0.01 and abs(ri.mean() - 1.0) < 0.03   # Invariant(Gamma) keeps the mean at 1
    # same uniforms: sites above the invariant mass are the same quantiles rescaled
    assert np.all(ri[ri > 0] > 0)


def test_alignment_columns_follow_the_site_rates():
    """a site with rate ~0 is constant; fast sites differ more often from the root state than slow ones"""
    case = make_case(24, 4, 20, 9)
    om = oracle.Model(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    aln, r = oracle.simulate_continuous(om, 11, 0, 3000, 0.5, 0.2)
    const = (aln == aln[0]).all(axis=0)
    assert const[r == 0].all()
    slow, fast = r < np.quantile(r, 0.3), r > np.quantile(r, 0.8)
    assert const[slow].mean() > const[fast].mean() + 0.3
