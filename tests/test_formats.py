"""Text formats either side of the path: the writers must reproduce the reference's own files.
The first lines of Myo_unif.vec / Myo.infos are committed verbatim in tests/golden/myoglobin.npz."""
import io

import numpy as np

from comap_amd import formats


def test_vec_writer_reproduces_reference_text(myo):
    head = str(myo["vec_unif_text_head"])
    counts = myo["vec_unif"].T[:, :, None]               # [N, B, 1]
    text = formats.to_text(formats.write_vec, counts, myo["vec_blen"], myo["coords"])
    # re-printing the parsed 6-digit values gives back the very same characters
    assert text.startswith(head)
    assert text.count("\n") == 1 + counts.shape[1]


def test_vec_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    counts = rng.random((7, 5, 1)) * np.array([1e-12, 1e-3, 1, 30, 1e5])[None, :, None]
    bl = np.array([1e-6, 0.01, 0.123456789, 2.5, 0.0])
    path = tmp_path / "x.vec"
    formats.write_vec(path, counts, bl, [3, 4, 9, 10, 11, 40, 41])
    back = formats.read_vec(path)
    assert back["coords"].tolist() == [3, 4, 9, 10, 11, 40, 41]
    assert back["branch_ids"].tolist() == [0, 1, 2, 3, 4]
    assert np.allclose(back["counts"], counts, rtol=5e-6, atol=0)
    assert np.allclose(back["branch_lengths"], bl, rtol=5e-6)
    # a second write of what was read is a fixed point
    assert formats.to_text(formats.write_vec, back["counts"], back["branch_lengths"], back["coords"]) == path.read_text()


def test_infos_writer_reproduces_reference_text(myo):
    head = str(myo["infos_text_head"])
    complete, const = formats.site_flags(myo["aln"], 20)
    text = formats.to_text(formats.write_infos, myo["coords"], complete, const, myo["infos_rc"], myo["infos_pr"], None,
                           myo["infos_logl"], with_norm=False)
    assert text.startswith(head)
    # current layout (CoETools.cpp:515) has the N column
    t2 = formats.to_text(formats.write_infos, myo["coords"][:2], complete[:2], const[:2], [3, 0], [2.34128, 0.290275],
                         [4.98613, 1.04684], [-60.8878, -10.9915])
    assert t2.splitlines()[0] == "Group\tIsComplete\tIsConstant\tRC\tPR\tN\tlogLn"
    assert t2.splitlines()[1] == "[162]\t1\t0\t3\t2.34128\t4.98613\t-60.8878"


def test_number_formatting_matches_default_ostream():
    cases = {0.0: "0", 1.0: "1", 0.000878519: "0.000878519", 1e-6: "1e-06", 123456.7: "123457", 1234567.0: "1.23457e+06",
             -60.8878: "-60.8878", 7.71622e-15: "7.71622e-15", float("inf"): "inf"}
    for v, s in cases.items():
        assert formats.fmt(v) == s


def test_intra_stats_rows_filters_and_na():
    coords = [5, 6, 9]
    stat = np.array([[1, 0.5, -0.2], [0.5, 1, 0.05], [-0.2, 0.05, 1.0]])
    pv = np.array([[np.nan] * 3, [np.nan, np.nan, 0.25], [np.nan] * 3])
    ns = np.array([[0] * 3, [0, 0, 3], [0] * 3])
    rc, pr, nm = [0, 2, 3], [0.2, 1.1, 2.0], [1.0, 3.0, 2.0]
    txt = formats.to_text(formats.write_intra_stats, coords, stat, rc, pr, nm, pv, ns)
    lines = txt.splitlines()
    assert lines[0] == "Group\tStat\tRCmin\tPRmin\tNmin\tPValue\tNsim"
    assert lines[1] == "[5;6]\t0.5\t0\t0.2\t1\tNA\t0"
    assert lines[3] == "[6;9]\t0.05\t2\t1.1\t2\t0.25\t3"
    txt = formats.to_text(formats.write_intra_stats, coords, stat, rc, pr, nm, min_rate_class=1, min_statistic=0.1)
    assert txt.splitlines() == ["Group\tStat\tRCmin\tPRmin\tNmin"]       # [6;9] fails |stat| >= 0.1
    buf = io.StringIO()
    formats.write_null(buf, [0.5, -1e-7], [1, 0], [0.25, 2], [3.5, 0.125])
    assert buf.getvalue() == "Stat\tRCmin\tPRmin\tNmin\n0.5\t1\t0.25\t3.5\n-1e-07\t0\t2\t0.125\n"
