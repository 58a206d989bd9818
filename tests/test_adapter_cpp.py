"""The C++ mirror of the reference interface (include/comap_mi355x_adapter.hpp): compiles with g++ against the C-ABI
library; Domain matches the oracle on CPU; the full getVectors -> computeIntraStats call sequence matches the oracle
on the GPU."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle
from comap_amd import engine
from conftest import make_case, rel_close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "adapter_main")


@pytest.fixture(scope="module")
def adapter_exe():
    src = os.path.join(ROOT, "tests", "cpp", "adapter_main.cpp")
    deps = [src, engine.LIB_PATH, os.path.join(ROOT, "include", "comap_mi355x_adapter.hpp"),
            os.path.join(ROOT, "include", "comap_mi355x.h")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                               "-L", os.path.dirname(engine.LIB_PATH), "-lcomap_mi355x",
                               "-Wl,-rpath," + os.path.dirname(engine.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_domain_matches_oracle(adapter_exe):
    rng = np.random.default_rng(0)
    hi = 4.98613
    xs = np.concatenate([rng.uniform(-0.1, hi * 1.05, 200), [0.0, hi, hi / 10 * 3, np.nextafter(hi, 0)]])
    out = subprocess.check_output([adapter_exe, "domain", "0", repr(hi), "10"] + [repr(float(x)) for x in xs]).split()
    got = np.array([int(v) for v in out])
    exp = np.array([oracle.domain_index(0, hi, 10, float(x)) for x in xs])
    assert np.array_equal(got, exp)
    assert got[-3] == -1 and got[-4] == 0          # upper bound exclusive, lower inclusive (Domain.cpp:115)


def test_errors_surface_as_exceptions(adapter_exe, tmp_path):
    bad = tmp_path / "bad.bin"
    # S = 65 is rejected by the engine -> cmx::Exception -> exit code 1 (reference: bpp::Exception caught in main)
    nn, T, S, C, N = 4, 3, 65, 1, 1
    with open(bad, "wb") as f:
        f.write(struct.pack("<8i", nn, T, S, C, N, 1, 1, 1) + struct.pack("<Q", 1))
        f.write(np.array([3, 3, 3, -1], dtype=np.int32).tobytes() + np.ones(4).tobytes())
        f.write(np.array([0, 1, 2], dtype=np.int32).tobytes())
        f.write(np.zeros(S * S).tobytes() + np.full(S, 1.0 / S).tobytes() + np.ones(1).tobytes() + np.ones(1).tobytes())
        f.write(bytes(T * N))
    r = subprocess.run([adapter_exe, "run", str(bad), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "nstates" in r.stderr


def test_cpp_vec_writer_reproduces_reference_text(adapter_exe, tmp_path, myo):
    """cmx::io::writeToStream (the C++ side of SURVEY 8f row 1) against the first lines of the reference's Myo_unif.vec"""
    counts = np.ascontiguousarray(myo["vec_unif"].T)       # [N, B] site-major
    inp = tmp_path / "vec.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<2i", counts.shape[0], counts.shape[1]))
        f.write(myo["coords"].astype(np.int32).tobytes() + myo["vec_blen"].astype(np.float64).tobytes())
        f.write(counts.astype(np.float64).tobytes())
    r = subprocess.run([adapter_exe, "vec", str(inp)], capture_output=True, text=True, check=True)
    assert r.stdout.startswith(str(myo["vec_unif_text_head"]))
    assert r.stderr == "Stat\tRCmin\tPRmin\tNmin\n0.5\t1\t0.25\t3.5\n-1e-07\t0\t2\t0.125\n"


@pytest.mark.gpu
def test_reference_call_sequence_matches_oracle(adapter_exe, tmp_path):
    case = make_case(9, 70, 20, 61)
    nn, T, S, C, N = len(case["parent"]), len(case["lot"]), 20, 4, 70
    rep_cpu, rep_ram, ncls, seed = 3, 64, 5, 4242
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<8i", nn, T, S, C, N, rep_cpu, rep_ram, ncls) + struct.pack("<Q", seed))
        f.write(case["parent"].astype(np.int32).tobytes() + case["blen"].tobytes() + case["lot"].astype(np.int32).tobytes())
        f.write(case["Q"].tobytes() + case["pi"].tobytes() + case["rates"].tobytes() + case["probs"].tobytes())
        f.write(np.ascontiguousarray(case["aln"]).tobytes())
    subprocess.check_call([adapter_exe, "run", str(inp), str(outp)])
    raw = open(outp, "rb").read()
    nrows = struct.unpack_from("<q", raw, 0)[0]
    assert nrows == N * (N - 1) // 2
    rec = np.dtype([("i", "<i8"), ("j", "<i8"), ("stat", "<f8"), ("pr", "<f8"), ("nm", "<f8"), ("pv", "<f8"),
                    ("rc", "<i4"), ("ns", "<i4")])
    rows = np.frombuffer(raw, dtype=rec, count=nrows, offset=8)
    B = nn - 1
    counts = np.frombuffer(raw, dtype="<f8", count=N * B, offset=8 + nrows * rec.itemsize).reshape(N, B, 1)
    om = oracle.Model(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    o = oracle.map_sites(om, case["aln"])
    rel_close(counts, o["counts"], 1e-6, 1e-300)
    st = oracle.pair_stats_intra(0, o["counts"])
    nl = oracle.null_intra(om, 0, seed, 0, rep_cpu, rep_ram)
    pv, ns = oracle.intra_pvalues(st, o["norm"], ncls, nl["stat"], nl["nmin"])
    iu = np.triu_indices(N, 1)
    assert np.array_equal(rows["i"], iu[0]) and np.array_equal(rows["j"], iu[1])      # the reference's row order
    rel_close(rows["stat"], st[iu], 1e-6, 1e-12)
    assert np.array_equal(rows["ns"], ns[iu])
    assert np.array_equal(rows["rc"], np.minimum(o["rate_class"][iu[0]], o["rate_class"][iu[1]]))
    rel_close(rows["nm"], np.minimum(o["norm"][iu[0]], o["norm"][iu[1]]), 1e-6)
    # p-values: the rule is p = (nsim - #{null < stat} + 1) / (nsim + 1) over the pair's norm class.  The device's null
    # agrees with the oracle's to 1e-6, so the two counts may differ only by null values that tie the statistic to that
    # tolerance (the cosine of two sparse count vectors takes few distinct values, so several replicates can tie at once)
    same = (rows["pv"] == pv[iu]) | (np.isnan(rows["pv"]) & np.isnan(pv[iu]))
    assert np.mean(same) > 0.99
    maxn = float(np.max(o["norm"]))
    ncl = np.array([-1 if np.isnan(s) else oracle.domain_index(0, maxn, ncls, float(m))
                    for s, m in zip(nl["stat"], nl["nmin"])])
    for q in np.flatnonzero(~same):
        i, j = iu[0][q], iu[1][q]
        cat = oracle.domain_index(0, maxn, ncls, float(min(o["norm"][i], o["norm"][j])))
        pool = nl["stat"][ncl == cat]
        ties = int(np.sum(np.abs(pool - st[i, j]) <= 2e-6 * max(abs(st[i, j]), 1e-12)))
        off = abs(rows["pv"][q] - pv[i, j]) * (ns[i, j] + 1)
        assert abs(off - round(off)) < 1e-9 and 1 <= round(off) <= ties, (i, j, off, ties)


def test_bpp_seam_header_guard_and_signatures(tmp_path):
    """include/comap_mi355x_bpp.hpp carries the reference's exact Bio++-typed seams (CoETools.h:317-399,
    AnalysisTools.h:198-275).  Bio++ is absent from this image: the header must then compile to nothing but the adapter,
    and its guarded block must declare every seam with the reference's parameter list."""
    src = tmp_path / "tu.cpp"
    src.write_text('#include "comap_mi355x_bpp.hpp"\n#ifdef CMX_HAVE_BPP\n#error "Bio++ unexpectedly present"\n#endif\n'
                   'int main() { cmx::Domain d(0., 1., 4); return d.getIndex(0.3) == 1 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
    text = open(os.path.join(ROOT, "include", "comap_mi355x_bpp.hpp")).read()
    flat = " ".join(text.split())
    for sig in [
        "getVectors( std::shared_ptr<const ::bpp::DRTreeLikelihoodInterface> drtl, std::shared_ptr<::bpp::SubstitutionCountInterface> substitutionCount, const ::bpp::SiteContainerInterface& completeSites, std::map<std::string, std::string>& params, const std::string& suffix = \"\")",
        "getNullDistributionIntraDR(std::shared_ptr<::bpp::DRTreeLikelihoodInterface> drtl, const ::bpp::SequenceSimulatorInterface& seqSim, std::shared_ptr<::bpp::SubstitutionCountInterface> nijt, const ::Statistic& statistic, std::ostream* out, ::bpp::VVdouble* simstats, const ::Domain* rateDomain, size_t repCPU, size_t repRAM, bool average, bool joint, bool verbose = true)",
        "computeIntraStats(const ::bpp::DRTreeLikelihoodInterface& tl, const ::bpp::SequenceSimulatorInterface& seqSim, const ::bpp::SiteContainerInterface& completeSites, ::bpp::LegacyProbabilisticSubstitutionMapping& mapping, std::shared_ptr<::bpp::SubstitutionCountInterface> nijt, const ::Statistic& statistic, bool computeNull, std::map<std::string, std::string>& params)",
    ]:
        assert sig in flat, sig
    assert "__has_include(<Bpp/Phyl/Legacy/Likelihood/DRTreeLikelihood.h>)" in text


def _write_case(path, case, N, rep_cpu, rep_ram, ncls, seed):
    nn, T, S, C = len(case["parent"]), len(case["lot"]), len(case["pi"]), len(case["rates"])
    with open(path, "wb") as f:
        f.write(struct.pack("<8i", nn, T, S, C, N, rep_cpu, rep_ram, ncls) + struct.pack("<Q", seed))
        f.write(case["parent"].astype(np.int32).tobytes() + case["blen"].tobytes() + case["lot"].astype(np.int32).tobytes())
        f.write(case["Q"].tobytes() + case["pi"].tobytes() + case["rates"].tobytes() + case["probs"].tobytes())
        f.write(np.ascontiguousarray(case["aln"]).tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("method,with_model,zstat", [("none", 0, "MI"), ("nonparametric-bootstrap", 0, "MI"),
                                                     ("nonparametric-bootstrap", 1, "MI"), ("parametric-bootstrap", 1, "MI"),
                                                     ("z-score", 0, "MIp"), ("z-score", 1, "MIc"), ("permutations", 0, "MI")])
def test_cpp_mica_seam_equals_the_python_mirror(adapter_exe, tmp_path, method, with_model, zstat):
    """cmx::Mica (all-pairs MI, APC / RCW, the four null methods, Bs.p.value, the output table of
    Mica.cpp:646-689) through the C++ adapter writes the bytes the Python mirror writes -- both are thin over the same
    C-ABI calls, bootstrap site indices included (cmx_mica_bootstrap_indices); the numbers themselves are checked
    against the oracle in tests/test_gpu_parity.py"""
    import io
    from comap_amd import formats, mica
    case = make_case(14, 40, 20, 33)
    N, rep_cpu, rep_ram, ncls, seed = 40, 3, 30, 4, 99
    inp = tmp_path / "in.bin"
    _write_case(inp, case, N, rep_cpu, rep_ram, ncls, seed)
    r = subprocess.run([adapter_exe, "mica", str(inp), method, str(with_model), zstat], capture_output=True, text=True, check=True)
    eng = engine.Engine(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    aln = case["aln"]
    norms = eng.map_sites(aln)["norm"] if with_model else None
    base = mica.analysis(eng, aln, 20, norms=norms)
    key = base["entropy"] if norms is None else norms
    null = perm = None
    null_rows = None
    if method == "nonparametric-bootstrap":
        nb = mica.bootstrap_null(eng, aln, base["entropy"], seed, rep_cpu, rep_ram, norms=norms)
        null = (nb["mi"], nb["nmin"] if with_model else nb["hmin"])
        null_rows = [nb["mi"], nb["hjoint"], nb["hmin"]] + ([nb["nmin"]] if with_model else [])
    elif method == "parametric-bootstrap":
        pn = eng.mica_parametric_null(seed, rep_cpu, rep_ram, with_norms=True)
        q = np.arange(rep_cpu * rep_ram)
        hm = np.minimum(base["entropy"][q // rep_ram], base["entropy"][q % rep_ram])      # Mica.cpp:528 as it is
        null = (pn["mi"], pn["nmin"])
        null_rows = [pn["mi"], pn["hjoint"], hm, pn["nmin"]]
    elif method == "z-score":
        null = mica.zscore_null(eng, base["mi"], base["entropy"], zstat, norms=norms)
    elif method == "permutations":
        perm = (200, seed)
    res = mica.analysis(eng, aln, 20, norms=norms, null=null, nclasses=ncls, permutations=perm)
    buf = io.StringIO()
    formats.write_mica(buf, 10 + np.arange(N), res)
    assert r.stdout == buf.getvalue()
    if null_rows is not None:
        head = "MI\tHjoint\tHmin" + ("\tNmin" if with_model else "") + "\n"
        body = "".join("\t".join(formats.fmt(c[k]) for c in null_rows) + "\n" for k in range(len(null_rows[0])))
        assert r.stderr == head + body
    else:
        assert r.stderr == ""


@pytest.mark.gpu
@pytest.mark.parametrize("indep,quirk", [(0, 0), (0, 1), (1, 0)])
def test_cpp_inter_stats_through_the_device_loop(adapter_exe, tmp_path, indep, quirk):
    """cmx::CoETools::computeInterStats (device pair loop, cmx_inter_rows) writes the statistics file of
    CoETools.cpp:777, 814-826: same rows as the Python binding of the same entry point, in the reference's order"""
    import io
    case = make_case(10, 80, 20, 17)
    inp = tmp_path / "in.bin"
    _write_case(inp, case, 80, 1, 1, 1, 1)
    r = subprocess.run([adapter_exe, "inter", str(inp), str(indep), str(quirk)], capture_output=True, text=True, check=True)
    eng = engine.Engine(case["parent"], case["blen"], case["lot"], case["Q"], case["pi"], case["rates"], case["probs"])
    m1, m2 = eng.map_sites(case["aln"][:, :40].copy()), eng.map_sites(case["aln"][:, 40:].copy())
    f = engine.InterFilters(min_rate_class1=1, min_rate2=0.2, min_statistic=0.05, independent_comparisons=bool(indep),
                            reference_norm_quirk=bool(quirk))
    rows, count = eng.inter_rows(engine.STAT_CORRELATION, m1, m2, f)
    assert count > 0
    from comap_amd import formats
    lines = ["Group\tStat\tRCmin\tPRmin\tNmin\n"]
    for q in rows:
        lines.append("[%d;%d]\t%s\t%d\t%s\t%s\n" % (100 + q["i"], 500 + q["j"], formats.fmt(q["stat"]), q["rc_min"],
                                                       formats.fmt(q["pr_min"]), formats.fmt(q["n_min"])))
    assert r.stdout == "".join(lines)


@pytest.mark.gpu
@pytest.mark.parametrize("n1,n2,dim", [(9, 9, 30), (12, 5, 125)])
def test_cpp_analysis_tools_matrices(adapter_exe, tmp_path, n1, n2, dim):
    """cmx::AnalysisTools::compute{ScalarProduct,Cosinus,Correlation,Covariance}Matrix (the reference's
    AnalysisTools.h:93-190 with an engine argument) against the definitions: one-set, two-set and independantComparisons forms,
    and the reference's DimensionException when the independent form gets sets of different lengths"""
    rng = np.random.default_rng(n1 * 100 + n2)
    a, b = rng.normal(size=(n1, dim)), rng.normal(loc=0.2, size=(n2, dim))
    inp, out = tmp_path / "m.bin", tmp_path / "o.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<3i", n1, n2, dim) + a.tobytes() + b.tobytes())
    r = subprocess.run([adapter_exe, "matrices", str(inp), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.frombuffer(open(out, "rb").read(), dtype="<f8")

    def cor(x, y, norm=True):
        xc, yc = x - x.mean(1, keepdims=True), y - y.mean(1, keepdims=True)
        cov = (xc @ yc.T) / (dim - 1)
        return cov / np.outer(np.sqrt((xc ** 2).sum(1) / (dim - 1)), np.sqrt((yc ** 2).sum(1) / (dim - 1))) if norm else cov
    cosm = lambda x, y: (x @ y.T) / np.outer(np.linalg.norm(x, axis=1), np.linalg.norm(y, axis=1))
    refs = [a @ a.T, a @ b.T, cosm(a, a), cosm(a, b), cor(a, a), cor(a, b), cor(a, a, False), cor(a, b, False)]
    off = 0
    for k, ref in enumerate(refs):
        got = raw[off:off + ref.size].reshape(ref.shape)
        off += ref.size
        if k in (2, 4):
            assert np.all(np.diag(got) == 1.0)
        rel_close(got, ref, 1e-9, 1e-12)
    if n1 == n2:
        ind = raw[off:off + n1 * n2].reshape(n1, n2)
        assert np.all(ind[~np.eye(n1, dtype=bool)] == 0.0)
        rel_close(np.diag(ind), np.diag(cor(a, b)), 1e-9, 1e-12)
    else:
        assert "DimensionException" in r.stdout and "independant comparisons" in r.stdout
