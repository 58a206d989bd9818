"""Pins the oracle(s) to the reference's committed fixtures (examples/Proteins/Benchmark/CoMap/Myo_*.vec, Myo.infos).

Fixture values are printed with 6 significant digits (default ostream precision, CoMap/CoETools.cpp:698-722) and were
produced by CoMap 1.4/1.5 + Bio++ 2.x; the measured agreement is: logL 4e-6 rel, posterior rate 1e-5 rel, rate class
exact, counts median 1.6e-6 / max 7.4e-5 rel.  Tolerances below are those measurements with a little headroom."""
import numpy as np
import pytest

import oracle
from comap_amd import protein_models as pm
from oracle import np_oracle as npo


def _model(myo, **kw):
    Q, pi = pm.jtt92()
    rates, probs = pm.gamma_rates(float(myo["alpha"]), int(myo["ncat"]))
    return oracle.Model(myo["parent"], myo["blen"], myo["leaf_of_taxon"], Q, pi, rates, probs, **kw), Q


def _check_counts(c, v, max_rel, med_rel):
    rel = np.abs(c - v) / np.where(np.abs(v) > 0, np.abs(v), 1.0)
    assert rel.max() < max_rel, rel.max()
    assert np.median(rel) < med_rel, np.median(rel)


def test_gamma_rates_match_infos_max_rate(myo):
    rates, _ = pm.gamma_rates(float(myo["alpha"]), 4)
    assert abs(rates[3] - myo["infos_pr"].max()) / rates[3] < 5e-6   # mean-of-category, SURVEY A.5
    assert abs(rates.mean() - 1.0) < 1e-12


def test_site_selection_matches_vec_header(myo):
    assert myo["aln"].shape == (100, 129) and len(myo["parent"]) == 198
    assert list(myo["coords"][:6]) == [162, 163, 164, 165, 166, 168]
    assert np.allclose(myo["vec_blen"], myo["blen"][:197], rtol=1e-6)   # "Mean" column = branch length, row = node id


def test_infos_and_unif_counts(myo):
    m, _ = _model(myo)
    r = oracle.map_sites(m, myo["aln"], myo["masks"])
    assert np.array_equal(r["rate_class"], myo["infos_rc"])
    assert np.max(np.abs(r["logL"] - myo["infos_logl"]) / np.abs(myo["infos_logl"])) < 1e-5
    assert np.max(np.abs(r["post_rate"] - myo["infos_pr"]) / myo["infos_pr"]) < 2e-5
    _check_counts(r["counts"][:, :, 0], myo["vec_unif"].T, 1e-4, 5e-6)
    assert abs(r["logL"].sum() - (-4885.276)) < 2e-3
    assert np.allclose(r["norm"][:5], [4.98613, 1.04684, 1.99140, 3.33126, 4.50954], rtol=5e-6)


def test_naive_counts(myo):
    m, _ = _model(myo, method=oracle.METHOD_NAIVE)
    r = oracle.map_sites(m, myo["aln"], myo["masks"])
    _check_counts(r["counts"][:, :, 0], myo["vec_naive"].T, 2e-4, 5e-6)


def test_grantham_weighted_counts(myo):
    W = pm.grantham_distance()
    Q, _ = pm.jtt92()
    B = npo.rate_matrix_register(Q, W)
    m, _ = _model(myo, Bk=B[None], nonneg=False)
    r = oracle.map_sites(m, myo["aln"], myo["masks"])
    _check_counts(r["counts"][:, :, 0], myo["vec_unif_grantham"].T, 1e-4, 5e-6)
    m2, _ = _model(myo, method=oracle.METHOD_NAIVE, naive_W=W)
    r2 = oracle.map_sites(m2, myo["aln"], myo["masks"])
    _check_counts(r2["counts"][:, :, 0], myo["vec_naive_grantham"].T, 2e-4, 5e-6)


def test_numpy_restatement_agrees_with_c_oracle(myo):
    m, Q = _model(myo)
    r = oracle.map_sites(m, myo["aln"], myo["masks"])
    r2 = npo.map_sites(myo["parent"], myo["blen"], myo["leaf_of_taxon"], myo["aln"], myo["masks"], m.Q, m.pi, m.rates,
                       m.probs, [npo.rate_matrix_register(Q)], method="unif")
    assert np.max(np.abs(r2["counts"] - r["counts"]) / np.abs(r["counts"])) < 1e-6
    assert np.max(np.abs(r2["logL"] - r["logL"])) < 1e-9
    assert np.array_equal(r2["rate_class"], r["rate_class"])


def test_decomposition_fixture_shows_cancellation_on_short_branches(myo):
    """The reference's own Myo_decomp.vec differs from Myo_unif.vec by up to 0.65 % on the O(t^2) counts of the
    1e-6 branches (plain difference quotient); the expm1 form used here agrees with uniformization."""
    a, b = myo["vec_unif"], myo["vec_decomp"]
    assert 1e-3 < np.max(np.abs(a - b) / np.where(b > 0, b, 1)) < 1e-2
    m, _ = _model(myo)
    md, _ = _model(myo, method=oracle.METHOD_DECOMP)
    ru, rd = oracle.map_sites(m, myo["aln"], myo["masks"]), oracle.map_sites(md, myo["aln"], myo["masks"])
    assert np.max(np.abs(ru["counts"] - rd["counts"]) / np.abs(ru["counts"])) < 1e-6
