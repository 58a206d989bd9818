"""Alphabets other than 4 / 20 states (codon models, CoMap/CoETools.cpp:95-100, 216-226): the plain kernels of
cmx_variants.hip with the states padded to 64.  Device against the oracle (which is generic in the state count):
mapping with every nijt.average / nijt.joint option, site scalars, unknowns, the simulator bit for bit, pair
statistics, the parametric-bootstrap null and the compacted statistics rows.  Parity unpinned against the reference like
everything past the Myoglobin fixtures (no codon output is shipped)."""
import numpy as np
import pytest

import oracle
from comap_amd import engine, protein_models as pm
from conftest import make_case, rel_close

pytestmark = pytest.mark.gpu


def _case(S, ntaxa=9, nsites=70, seed=5, K=1):
    case = make_case(ntaxa, nsites, 20, seed)
    Q, pi = pm.synthetic_reversible(S, seed + 100)
    rng = np.random.default_rng(seed)
    aln = rng.integers(0, S, size=case["aln"].shape).astype(np.uint8)
    base = rng.integers(0, S, size=(1, nsites))
    aln = np.where(rng.random(aln.shape) < 0.6, base, aln).astype(np.uint8)     # columns with signal
    aln[2, ::7] = S                                                               # unknowns (gap / NNN) at a leaf
    aln[5, 3::11] = 200                                                           # any code >= S is an unknown
    case.update(Q=Q, pi=pi, aln=aln)
    return case


def _pair(case, **kw):
    eng = engine.Engine(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"], **kw)
    om = oracle.Model(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    return eng, om


@pytest.mark.parametrize("S", [61, 64, 7])
def test_plain_alphabet_mapping_matches_oracle(S):
    case = _case(S)
    eng, om = _pair(case)
    assert eng.info()["nstates"] == S
    r = eng.map_sites(case["aln"])
    o = oracle.map_sites(om, case["aln"])
    rel_close(r["counts"], o["counts"], 1e-9, 1e-300)
    rel_close(r["logL"], o["logL"], 1e-12)
    rel_close(r["post_rate"], o["post_rate"], 1e-12)
    rel_close(r["norm"], o["norm"], 1e-9)
    assert np.array_equal(r["rate_class"], o["rate_class"])
    assert (r["counts"] >= 0).all()


def test_plain_alphabet_mapping_variants_match_oracle():
    case = _case(61, nsites=40)
    eng, om = _pair(case)
    eng.set_mapping_options(False, True)
    r = eng.map_sites(case["aln"])
    o = oracle.map_sites_noavg(om, case["aln"])
    same = np.isclose(r["counts"], o["counts"], rtol=1e-9, atol=1e-300)
    assert same.mean() > 0.995            # elsewhere the two best cells tie to rounding (tests/test_gpu_parity.py, NoAveraging)
    for average in (True, False):
        eng.set_mapping_options(average, False)
        r = eng.map_sites(case["aln"])
        o = oracle.map_sites_marginal(om, case["aln"], average)
        if average:
            rel_close(r["counts"], o["counts"], 1e-9, 1e-300)
        else:
            assert np.isclose(r["counts"], o["counts"], rtol=1e-9, atol=1e-300).mean() > 0.995
    eng.set_mapping_options(True, True)
    rel_close(eng.map_sites(case["aln"])["counts"], oracle.map_sites(om, case["aln"])["counts"], 1e-9, 1e-300)


def test_plain_alphabet_simulator_null_and_rows():
    case = _case(61, ntaxa=8, nsites=50)
    eng, om = _pair(case)
    aln, cls = eng.simulate(77, 0, 300)
    oaln, ocls = oracle.simulate(om, 77, 0, 300)
    assert np.array_equal(aln, oaln) and np.array_equal(cls, ocls) and aln.max() < 61
    nl = eng.null_intra(engine.STAT_CORRELATION, 11, 0, 3, 40)
    onl = oracle.null_intra(om, 0, 11, 0, 3, 40)
    rel_close(nl["stat"], onl["stat"], 1e-6, 1e-12)
    rel_close(nl["nmin"], onl["nmin"], 1e-9)
    assert np.array_equal(nl["rcmin"], onl["rcmin"])
    m = eng.map_sites(case["aln"])
    st = eng.pair_stats(engine.STAT_CORRELATION, m["counts"])
    ost = oracle.pair_stats_intra(0, oracle.map_sites(om, case["aln"])["counts"])
    iu = np.triu_indices(50, 1)
    rel_close(st[iu], ost[iu], 1e-6, 1e-12)
    rows, count = eng.intra_rows(engine.STAT_CORRELATION, m["counts"], m["rate_class"], m["post_rate"], m["norm"], nl["stat"], nl["nmin"],
                                 nclasses=3)
    assert count == 50 * 49 // 2
    opv, ons = oracle.intra_pvalues(st, m["norm"], 3, nl["stat"], nl["nmin"])
    assert np.array_equal(rows["nsim"], ons[iu])


def test_plain_alphabet_refuses_a_mask_table():
    case = _case(61, nsites=10)
    eng, _ = _pair(case)
    with pytest.raises(engine.CmxError) as e:
        eng.map_sites(case["aln"], masks=np.full(64, 0xFFFFFFFF, dtype=np.uint32))
    assert "ambiguity table" in str(e.value)
