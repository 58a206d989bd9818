"""Group statistics and the candidate-group test through the C-ABI against oracle/candidates.py.
Counters (n2, trials, batches) are integer bookkeeping driven by the norms: exact.  n1 compares a floating-point
statistic with a threshold: exact unless a pseudo-group's statistic ties the observed one to 1e-6 (checked)."""
import numpy as np
import pytest

import oracle
from oracle import candidates as ocand
from comap_amd import engine, synthetic
from conftest import rel_close

pytestmark = pytest.mark.gpu


def _setup(nstates=20, ntaxa=11, seed=4, weighted=False):
    parent, blen, lot = synthetic.random_tree(ntaxa, seed)
    mdl = synthetic.protein_model(0.8, 4) if nstates == 20 else synthetic.dna_model(0.8, 4)
    kw = {}
    if weighted:
        rng = np.random.default_rng(seed)
        kw = dict(Bk=np.stack([synthetic.weighted_register(mdl["Q"], rng.uniform(-1, 1, size=(nstates, nstates)))]))
    om = oracle.Model(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], nonneg=not weighted, **kw)
    eng = engine.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], clamp_negative=not weighted, **kw)
    return om, eng


@pytest.mark.parametrize("kind", [oracle.ST_CORRELATION, oracle.ST_COMPENSATION, oracle.ST_COSUBSTITUTION, oracle.ST_COSINUS,
                                  oracle.ST_COVARIANCE, oracle.ST_DISCRETE_MI])
def test_group_stats_against_oracle(kind):
    om, eng = _setup(weighted=kind == oracle.ST_COMPENSATION)
    aln, _ = oracle.simulate(om, 3, 0, 90)
    counts = oracle.map_sites(om, aln)["counts"]
    rng = np.random.default_rng(kind)
    groups = [list(rng.choice(90, size=int(m), replace=False)) for m in (2, 3, 5, 12, 2, 40)]
    got = eng.group_stats(kind, counts, groups, threshold=0.05)
    params = oracle.stat_params(kind, 0.05)
    exp = np.array([ocand.group_stat(kind, [counts[i] for i in g], params) for g in groups])
    rel_close(got, exp, 1e-6, 1e-12)


def test_group_stats_rejects_bad_groups():
    om, eng = _setup()
    counts = np.random.default_rng(0).random((5, eng.B, 1))
    with pytest.raises(engine.CmxError, match="out of range"):
        eng.group_stats(0, counts, [[0, 7]])


def _candidates(om, eng, kind, seed, ngroups, omega, nobs=60):
    """observed data set -> candidate groups with their norm windows and observed statistics (CoMap.cpp:629-677)"""
    aln, _ = oracle.simulate(om, seed, 10 ** 6, nobs)
    mp = oracle.map_sites(om, aln)
    rng = np.random.default_rng(seed)
    groups = [list(rng.choice(nobs, size=int(rng.integers(2, 5)), replace=False)) for _ in range(ngroups)]
    windows = [[(mp["norm"][i] - omega, mp["norm"][i] + omega) for i in g] for g in groups]
    observed = eng.group_stats(kind, mp["counts"], groups)
    return groups, windows, observed


@pytest.mark.parametrize("kind,analysable", [(oracle.ST_CORRELATION, [1, 1, 1, 1, 1]), (oracle.ST_COMPENSATION, [1, 0, 1, 1, 0]),
                                             (oracle.ST_COSUBSTITUTION, [0, 1, 1, 1, 1])])
def test_candidate_groups_against_oracle(kind, analysable):
    om, eng = _setup(weighted=kind == oracle.ST_COMPENSATION)
    groups, windows, observed = _candidates(om, eng, kind, 11 + kind, 5, 0.3)
    args = dict(min_sim=25, rep_ram=48, max_trials=4, seed=2024)
    g = eng.candidate_groups(kind, windows, analysable, observed, **args)
    o = ocand.candidate_groups(om, kind, windows, analysable, observed, **args)
    assert np.array_equal(g["n2"], o["n2"]) and g["trials"] == o["trials"] and g["batches"] == o["batches"]
    # n1 compares a floating-point statistic with the observed one: pseudo-groups made of the same column patterns as
    # the observed group reproduce its statistic to the last bits, and those ties fall either way
    assert np.all(np.abs(g["n1"].astype(np.int64) - o["n1"]) <= o["near_ties"])
    assert all(g["n2"][k] == 0 for k in range(5) if not analysable[k])
    assert np.all(g["n2"] <= 25) and g["batches"] >= 1
    assert np.allclose(g["pvalue"], (o["n1"] + 1.0) / (o["n2"] + 1.0))


def test_candidate_groups_gives_up_after_max_trials():
    """windows no simulated norm can fall into: every batch completes nothing, nb_max_trials ends the loop"""
    om, eng = _setup()
    windows = [[(1e6, 1e6 + 1)] * 2, [(2e6, 2e6 + 1)] * 3]
    g = eng.candidate_groups(oracle.ST_CORRELATION, windows, [1, 1], [0.5, 0.5], min_sim=10, rep_ram=32, max_trials=3, seed=5)
    o = ocand.candidate_groups(om, oracle.ST_CORRELATION, windows, [1, 1], [0.5, 0.5], 10, 32, 3, 5)
    assert g["trials"] == 3 == o["trials"] and g["batches"] == 3 == o["batches"] and not g["n2"].any()
    with pytest.raises(engine.CmxError, match="no analysable group"):
        eng.candidate_groups(oracle.ST_CORRELATION, windows, [0, 0], [0.5, 0.5], min_sim=10, rep_ram=32, max_trials=3, seed=5)


def test_candidate_groups_full_size_properties():
    """candidates.null.min = 1000 (the reference's default) for 40 groups: every analysable group reaches exactly
    min_sim pseudo-groups, p-values are in (0, 1], and a wide-open group set under the null is not all-significant"""
    om, eng = _setup(ntaxa=24, seed=9)
    groups, windows, observed = _candidates(om, eng, oracle.ST_CORRELATION, 77, 40, 0.5, nobs=200)
    g = eng.candidate_groups(oracle.ST_CORRELATION, windows, [1] * 40, observed, min_sim=1000, rep_ram=1000, max_trials=10, seed=1)
    assert np.all(g["n2"] == 1000) and g["trials"] == 0
    assert np.all(g["pvalue"] > 0) and np.all(g["pvalue"] <= 1) and np.median(g["pvalue"]) > 0.05


def test_cpp_candidate_group_set_matches_python(tmp_path):
    """cmx::CandidateGroup / CandidateGroupSet / CoETools::computePValuesForCandidateGroups (C++ mirror) == the ctypes path"""
    import os
    import struct
    import subprocess
    from conftest import make_case
    from test_adapter_cpp import EXE, ROOT
    src = os.path.join(ROOT, "tests", "cpp", "adapter_main.cpp")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                           "-L", os.path.dirname(engine.LIB_PATH), "-lcomap_mi355x",
                           "-Wl,-rpath," + os.path.dirname(engine.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    case = make_case(9, 30, 20, 61)
    nn, T, S, C, N = len(case["parent"]), len(case["lot"]), 20, 4, 30
    omega, min_sim, rep_ram, max_trials, seed = 0.4, 20, 64, 3, 31337
    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<8i", nn, T, S, C, N, 1, 1, 1) + struct.pack("<Q", 1))
        f.write(case["parent"].astype(np.int32).tobytes() + case["blen"].tobytes() + case["lot"].astype(np.int32).tobytes())
        f.write(case["Q"].tobytes() + case["pi"].tobytes() + case["rates"].tobytes() + case["probs"].tobytes())
        f.write(np.ascontiguousarray(case["aln"]).tobytes())
    out = subprocess.run([EXE, "candidates", str(inp), str(omega), str(min_sim), str(rep_ram), str(max_trials), str(seed)],
                         capture_output=True, text=True, check=True).stdout.split("\n")
    eng = engine.Engine(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    mp = eng.map_sites(case["aln"])
    groups = [[3 * g, 3 * g + 1, 3 * g + 2] for g in range(4)]
    observed = eng.group_stats(engine.STAT_CORRELATION, mp["counts"], groups)
    windows = [[(mp["norm"][i] - omega, mp["norm"][i] + omega) for i in g] for g in groups]
    r = eng.candidate_groups(engine.STAT_CORRELATION, windows, [1, 1, 1, 0], observed, min_sim, rep_ram, max_trials, seed)
    for g in range(4):
        st, n1, n2, pv = (float(x) for x in out[g].split())
        assert st == observed[g] and n1 == r["n1"][g] and n2 == r["n2"][g] and pv == r["pvalue"][g]
    assert [int(x) for x in out[4].split()] == [r["trials"], r["batches"]]


@pytest.mark.parametrize("seed", range(8))
def test_random_candidate_configurations_against_oracle(seed):
    """seeded sweep over group sets (sizes 1-6, some not analysable), window widths, min_sim, batch sizes, statistics"""
    rng = np.random.default_rng(300 + seed)
    kind = int(rng.choice([oracle.ST_CORRELATION, oracle.ST_COSINUS, oracle.ST_COVARIANCE, oracle.ST_COSUBSTITUTION]))
    om, eng = _setup(nstates=20 if seed % 2 == 0 else 4, ntaxa=int(rng.integers(6, 14)), seed=50 + seed)
    aln, _ = oracle.simulate(om, seed, 10 ** 6, 80)
    mp = oracle.map_sites(om, aln)
    G = int(rng.integers(1, 8))
    groups = [list(rng.choice(80, size=int(rng.integers(1, 7)), replace=False)) for _ in range(G)]
    omega = float(rng.uniform(0.05, 0.8))
    windows = [[(mp["norm"][i] - omega, mp["norm"][i] + omega) for i in g] for g in groups]
    analysable = [int(len(g) >= 2 and rng.random() < 0.85) for g in groups]
    if not any(analysable):
        analysable[0], groups[0] = 1, [0, 1]
        windows[0] = [(mp["norm"][i] - omega, mp["norm"][i] + omega) for i in groups[0]]
    observed = np.where(analysable, eng.group_stats(kind, mp["counts"], [g if len(g) >= 2 else [0, 1] for g in groups]), 0.0)
    args = dict(min_sim=int(rng.integers(1, 30)), rep_ram=int(rng.integers(5, 70)), max_trials=int(rng.integers(1, 4)), seed=1000 + seed,
                max_batches=60)
    g = eng.candidate_groups(kind, windows, analysable, observed, **args)
    o = ocand.candidate_groups(om, kind, windows, analysable, observed, **args)
    assert np.array_equal(g["n2"], o["n2"]) and g["trials"] == o["trials"] and g["batches"] == o["batches"]
    # n1 compares a floating-point statistic with the observed one: pseudo-groups made of the same column patterns as
    # the observed group reproduce its statistic to the last bits, and those ties fall either way
    assert np.all(np.abs(g["n1"].astype(np.int64) - o["n1"]) <= o["near_ties"])
