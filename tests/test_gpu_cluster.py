"""Clustering analysis on the device against oracle/cluster.py, through the C-ABI.
The agglomeration is integer/compare work plus one explicitly rounded formula: merges, sizes and join distances must be
bit-identical for the same input matrix.  Distances and group statistics are floating point: 1e-6 relative."""
import numpy as np
import pytest

import oracle
from oracle import cluster as oc
from comap_amd import engine, synthetic

pytestmark = pytest.mark.gpu

LINKS = [oc.LINK_COMPLETE, oc.LINK_SINGLE, oc.LINK_AVERAGE]


@pytest.fixture(scope="module")
def bare():
    return engine.Engine()


def _random_dist(n, seed, dims=5):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, dims))
    d = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(-1))
    d = (d + d.T) / 2
    np.fill_diagonal(d, 0.0)
    return d


def _same_tree(g, o):
    assert np.array_equal(g["merge"], o[0])
    assert np.array_equal(g["dmax"], o[1])
    assert np.array_equal(g["size"], o[2])


@pytest.mark.parametrize("link", LINKS)
@pytest.mark.parametrize("n", [2, 3, 5, 64, 65, 300, 777])
def test_hclust_bit_exact(bare, link, n):
    d = _random_dist(n, 10 * n + link)
    _same_tree(bare.hclust(d, link), oc.hclust(d, link))


@pytest.mark.parametrize("link", LINKS)
def test_hclust_ties_everywhere(bare, link):
    """small integer distances: most steps have several candidate pairs -> the first pair in index order must win"""
    rng = np.random.default_rng(4)
    n = 150
    d = rng.integers(1, 4, size=(n, n)).astype(np.float64)
    d = np.maximum(d, d.T)
    np.fill_diagonal(d, 0.0)
    _same_tree(bare.hclust(d, link), oc.hclust(d, link))
    d[:] = 1.0
    np.fill_diagonal(d, 0.0)
    _same_tree(bare.hclust(d, link), oc.hclust(d, link))


@pytest.mark.parametrize("link", LINKS)
def test_hclust_nan_and_inf(bare, link):
    d = _random_dist(40, 9)
    d[7, :] = d[:, 7] = np.nan
    d[20, 3] = d[3, 20] = np.inf
    d[7, 7] = 0.0
    _same_tree(bare.hclust(d, link), oc.hclust(d, link))
    d[:] = np.nan
    _same_tree(bare.hclust(d, link), oc.hclust(d, link))


def test_hclust_batch_of_independent_matrices(bare):
    ds = np.stack([_random_dist(90, 50 + b) for b in range(7)])
    g = bare.hclust(ds, oc.LINK_AVERAGE)
    for b in range(7):
        o = oc.hclust(ds[b], oc.LINK_AVERAGE)
        assert np.array_equal(g["merge"][b], o[0]) and np.array_equal(g["dmax"][b], o[1]) and np.array_equal(g["size"][b], o[2])


def test_hclust_full_size_against_scipy(bare):
    """n = 2000 (BASELINE cfg3's alignment length): too slow for the O(n^3) restatement, so check the dendrogram
    against scipy's: same sorted join distances, monotone, complete bookkeeping"""
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import squareform
    n = 2000
    d = _random_dist(n, 77, dims=12)
    for link, name in ((oc.LINK_COMPLETE, "complete"), (oc.LINK_AVERAGE, "average"), (oc.LINK_SINGLE, "single")):
        g = bare.hclust(d, link)
        Z = linkage(squareform(d, checks=False), name)
        assert np.allclose(np.sort(g["dmax"]), Z[:, 2], rtol=1e-12, atol=0)
        assert np.all(np.diff(g["dmax"]) >= -1e-12) and g["size"][-1] == n
        used = np.sort(g["merge"].ravel())
        assert np.array_equal(used, np.arange(2 * n - 2))             # every node is a son exactly once, root excepted


def test_hclust_argument_errors(bare):
    with pytest.raises(engine.CmxError):
        bare.hclust(np.zeros((1, 1)), oc.LINK_COMPLETE)
    with pytest.raises(engine.CmxError):
        bare.hclust(np.zeros((4, 4)), 7)
    with pytest.raises(engine.CmxError, match="limited"):
        bare.hclust(np.zeros((engine.CLUSTER_MAX_SITES + 1,) * 2), oc.LINK_COMPLETE)


def _setup(nstates, ntaxa, seed, ncat=4, Bk=False):
    """long branches: identical alignment columns give identical vectors, i.e. distances that tie to the last bit or
    not depending on rounding -- the tree is then not a function of the data (see _no_duplicate_columns)"""
    parent, blen, lot = synthetic.random_tree(ntaxa, seed)
    blen = np.asarray(blen) * (6.0 if nstates == 20 else 15.0)
    mdl = synthetic.protein_model(5.0, ncat) if nstates == 20 else synthetic.dna_model(5.0, ncat)
    kw = {}
    if Bk:      # compensation distance needs signed weights (CoMap.cpp:414-423)
        rng = np.random.default_rng(seed)
        kw = dict(Bk=np.stack([synthetic.weighted_register(mdl["Q"], rng.uniform(-1, 1, size=(nstates, nstates)))]))
    om = oracle.Model(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], nonneg=not Bk, **kw)
    eng = engine.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"], clamp_negative=not Bk, **kw)
    return om, eng


def _no_duplicate_columns(aln):
    assert np.unique(aln, axis=1).shape[1] == aln.shape[1], "test precondition: simulated columns must be distinct"


@pytest.mark.parametrize("dist", [oc.DIST_CORRELATION, oc.DIST_COMPENSATION, oc.DIST_EUCLIDIAN])
@pytest.mark.parametrize("link", LINKS)
def test_cluster_sites_against_oracle(dist, link):
    om, eng = _setup(20, 14, 31, Bk=dist == oc.DIST_COMPENSATION)
    aln, _ = oracle.simulate(om, 5, 0, 130)
    _no_duplicate_columns(aln)
    counts = oracle.map_sites(om, aln)["counts"]
    g = eng.cluster_sites(dist, link, counts)
    d = oc.distance_matrix(dist, counts)
    assert np.array_equal(g["dist"], g["dist"].T) and np.all(np.diag(g["dist"]) == 0)
    assert np.allclose(g["dist"], d, rtol=1e-6, atol=1e-12)
    # the agglomeration of the device's own matrix is exact; against the oracle's matrix it is the same tree
    _same_tree(g, oc.hclust(g["dist"], link))
    merge, dmax, size = oc.hclust(d, link)
    assert np.array_equal(g["merge"], merge) and np.array_equal(g["size"], size)
    assert np.allclose(g["dmax"], dmax, rtol=1e-6, atol=1e-12)
    stat, nmin = oc.group_properties(dist, merge, dmax, counts)
    assert np.allclose(g["stat"], stat, rtol=1e-6, atol=1e-9)
    assert np.allclose(g["nmin"], nmin, rtol=1e-6, atol=0)


@pytest.mark.parametrize("nstates,dist,link", [(20, oc.DIST_CORRELATION, oc.LINK_COMPLETE), (4, oc.DIST_EUCLIDIAN, oc.LINK_AVERAGE),
                                               (20, oc.DIST_COMPENSATION, oc.LINK_SINGLE)])
def test_cluster_null_against_oracle(nstates, dist, link):
    om, eng = _setup(nstates, 10 if nstates == 20 else 18, 8, Bk=dist == oc.DIST_COMPENSATION)
    nsites, r0, r1 = 70, 2, 6
    for k in range(r0, r1):
        _no_duplicate_columns(oracle.simulate(om, 123, k * nsites, nsites)[0])
    g = eng.cluster_null(dist, link, 123, r0, r1, nsites)
    o = oc.cluster_null(om, dist, link, 123, r0, r1, nsites)
    for k in range(r1 - r0):
        assert np.array_equal(g["merge"][k], o[k]["merge"]) and np.array_equal(g["size"][k], o[k]["size"])
        assert np.allclose(g["dmax"][k], o[k]["dmax"], rtol=1e-6, atol=1e-12)
        assert np.allclose(g["stat"][k], o[k]["stat"], rtol=1e-6, atol=1e-9)
        assert np.allclose(g["nmin"][k], o[k]["nmin"], rtol=1e-6, atol=0)


def test_cluster_null_does_not_depend_on_how_replicates_are_sharded():
    om, eng = _setup(20, 9, 3)
    a = eng.cluster_null(oc.DIST_CORRELATION, oc.LINK_COMPLETE, 9, 0, 6, 50)
    b = eng.cluster_null(oc.DIST_CORRELATION, oc.LINK_COMPLETE, 9, 0, 3, 50)
    c = eng.cluster_null(oc.DIST_CORRELATION, oc.LINK_COMPLETE, 9, 3, 6, 50)
    for k in a:
        assert np.array_equal(a[k], np.concatenate([b[k], c[k]]))


def test_cpp_cluster_null_writes_the_same_table_as_python(tmp_path):
    """cmx::ClusterTools::computeGlobalDistanceDistribution (C++ mirror) == engine.cluster_null + formats writer"""
    import struct
    import subprocess
    from comap_amd import formats
    from conftest import make_case
    from test_adapter_cpp import EXE, ROOT
    import os
    src = os.path.join(ROOT, "tests", "cpp", "adapter_main.cpp")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                           "-L", os.path.dirname(engine.LIB_PATH), "-lcomap_mi355x",
                           "-Wl,-rpath," + os.path.dirname(engine.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    case = make_case(9, 4, 20, 61)
    nn, T, S, C = len(case["parent"]), len(case["lot"]), 20, 4
    seed, nsites, nrep, maxsize = 99, 45, 3, 6
    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<8i", nn, T, S, C, 4, 1, 1, 1) + struct.pack("<Q", seed))
        f.write(case["parent"].astype(np.int32).tobytes() + case["blen"].tobytes() + case["lot"].astype(np.int32).tobytes())
        f.write(case["Q"].tobytes() + case["pi"].tobytes() + case["rates"].tobytes() + case["probs"].tobytes())
    eng = engine.Engine(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    for name, dist in (("cor", oc.DIST_CORRELATION), ("euclidian", oc.DIST_EUCLIDIAN)):
        got = subprocess.run([EXE, "clusternull", str(inp), name, str(oc.LINK_AVERAGE), str(nsites), str(nrep), str(maxsize)],
                             capture_output=True, text=True, check=True).stdout
        exp = formats.to_text(formats.write_cluster_null, eng.cluster_null(dist, oc.LINK_AVERAGE, seed, 0, nrep, nsites), maxsize)
        assert got == exp and got.count("\n") > nrep


def test_cluster_sites_at_the_size_limit():
    """n = CMX_CLUSTER_MAX_SITES: the per-matrix state (100 KB) and the group-property state (160 KB) still fit LDS;
    dendrogram against scipy, group properties against their definitions"""
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import squareform
    n = engine.CLUSTER_MAX_SITES
    parent, blen, lot = synthetic.random_tree(6, 2)
    mdl = synthetic.dna_model(1.0, 1)
    eng = engine.Engine(parent, blen, lot, mdl["Q"], mdl["pi"], mdl["rates"], mdl["probs"])
    rng = np.random.default_rng(8)
    counts = rng.random((n, eng.B, 1))
    g = eng.cluster_sites(oc.DIST_EUCLIDIAN, oc.LINK_AVERAGE, counts)
    d = g["dist"]
    ref = np.sqrt(((counts[:50, None, :, 0] - counts[None, :50, :, 0]) ** 2).sum(-1))
    assert np.allclose(d[:50, :50], ref, rtol=1e-9, atol=1e-12)
    Z = linkage(squareform(d, checks=False), "average")
    assert np.allclose(np.sort(g["dmax"]), Z[:, 2], rtol=1e-10, atol=0)
    assert g["size"][-1] == n and np.array_equal(g["stat"], g["dmax"])
    norm = np.sqrt((counts.sum(2) ** 2).sum(1))
    assert np.isclose(g["nmin"][-1], norm.min(), rtol=1e-12)
    for m, mem in oc.groups(g["merge"], max_group_size=3)[:200]:
        assert np.isclose(g["nmin"][m], norm[mem].min(), rtol=1e-12)


@pytest.mark.parametrize("seed", range(10))
def test_random_cluster_configurations_against_oracle(seed):
    """seeded sweep: alphabet, taxa, number of sites (2 ..), distance, linkage, weighted counts"""
    rng = np.random.default_rng(900 + seed)
    nstates = 20 if rng.random() < 0.6 else 4
    dist = int(rng.integers(0, 3))
    link = int(rng.integers(0, 3))
    om, eng = _setup(nstates, int(rng.integers(8, 16)) if nstates == 20 else 18, 40 + seed, Bk=dist == oc.DIST_COMPENSATION)
    n = int(rng.integers(2, 90))
    aln, _ = oracle.simulate(om, 77 + seed, 0, n)
    if np.unique(aln, axis=1).shape[1] != n:
        pytest.skip("simulated columns not distinct for this seed")
    counts = oracle.map_sites(om, aln)["counts"]
    g = eng.cluster_sites(dist, link, counts)
    d = oc.distance_matrix(dist, counts)
    merge, dmax, size = oc.hclust(d, link)
    assert np.array_equal(g["merge"], merge) and np.array_equal(g["size"], size)
    assert np.allclose(g["dmax"], dmax, rtol=1e-6, atol=1e-12)
    stat, nmin = oc.group_properties(dist, merge, dmax, counts)
    assert np.allclose(g["stat"], stat, rtol=1e-6, atol=1e-9) and np.allclose(g["nmin"], nmin, rtol=1e-6, atol=0)


def test_cpp_observed_clustering_matches_python(tmp_path):
    """cmx::ClusterTools::cluster + getGroups + io::writeGroups (C++ mirror of CoMap.cpp:432-548) == the ctypes path"""
    import os
    import struct
    import subprocess
    from comap_amd import cluster as pc, formats
    from conftest import make_case
    from test_adapter_cpp import EXE, ROOT
    src = os.path.join(ROOT, "tests", "cpp", "adapter_main.cpp")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                           "-L", os.path.dirname(engine.LIB_PATH), "-lcomap_mi355x",
                           "-Wl,-rpath," + os.path.dirname(engine.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    case = make_case(10, 55, 20, 7)
    nn, T, S, C, N = len(case["parent"]), len(case["lot"]), 20, 4, 55
    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<8i", nn, T, S, C, N, 1, 1, 1) + struct.pack("<Q", 1))
        f.write(case["parent"].astype(np.int32).tobytes() + case["blen"].tobytes() + case["lot"].astype(np.int32).tobytes())
        f.write(case["Q"].tobytes() + case["pi"].tobytes() + case["rates"].tobytes() + case["probs"].tobytes())
        f.write(np.ascontiguousarray(case["aln"]).tobytes())
    got = subprocess.run([EXE, "cluster", str(inp), str(oc.LINK_COMPLETE), "6"], capture_output=True, text=True, check=True).stdout
    eng = engine.Engine(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    mp = eng.map_sites(case["aln"])
    g = eng.cluster_sites(oc.DIST_CORRELATION, oc.LINK_COMPLETE, mp["counts"])
    exp = formats.to_text(formats.write_groups, pc.get_groups(g["merge"], 6), 10 + np.arange(N), np.zeros(N), g["dmax"], g["stat"],
                          g["nmin"])
    exp += formats.fmt(g["dist"][0, 1]) + " " + formats.fmt(g["dist"][2, 1]) + "\n"
    assert got == exp
