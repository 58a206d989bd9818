"""Host side of the clustering analysis: group extraction, the groups / null tables, Newick, and the C++ mirror of
ClusterTools, against the restatement in oracle/cluster.py (CPU only; the tree comes from the oracle here)."""
import struct
import subprocess

import numpy as np
import pytest

from oracle import cluster as oc
from comap_amd import cluster as pc, formats
from test_adapter_cpp import adapter_exe  # noqa: F401  (fixture)


def _tree(n, seed, link=oc.LINK_COMPLETE):
    rng = np.random.default_rng(seed)
    counts = rng.random((n, 9, 1))
    d = oc.distance_matrix(oc.DIST_CORRELATION, counts)
    merge, dmax, size = oc.hclust(d, link)
    stat, nmin = oc.group_properties(oc.DIST_CORRELATION, merge, dmax, counts)
    return merge, dmax, size, stat, nmin


@pytest.mark.parametrize("n,maxsize", [(2, None), (3, 2), (40, 10), (41, None)])
def test_get_groups_matches_the_restatement(n, maxsize):
    merge = _tree(n, n)[0]
    assert pc.get_groups(merge, maxsize) == oc.groups(merge, max_group_size=maxsize)


def test_groups_table_text():
    merge = np.array([[0, 1], [3, 4], [2, 6], [5, 7]], dtype=np.int32)
    dmax, stat, nmin = np.array([0.1, 0.25, 0.5, 1.5]), np.array([0.9, 0.75, 0.5, -0.5]), np.array([2.0, 1.0, 0.5, 0.5])
    coords = [10, 11, 15, 20, 21]
    txt = formats.to_text(formats.write_groups, pc.get_groups(merge, 3), coords, [0, 0, 1, 0, 0], dmax, stat, nmin)
    assert txt == ("Group\tSize\tIsConstant\tDmax\tStat\tNmin\n"
                   "[10;11]\t2\tno\t0.1\t0.9\t2\n"
                   "[20;21]\t2\tno\t0.25\t0.75\t1\n"
                   "[15;20;21]\t3\tyes\t0.5\t0.5\t0.5\n")
    null = dict(merge=merge[None], dmax=dmax[None], stat=stat[None], nmin=nmin[None])
    txt = formats.to_text(formats.write_cluster_null, null, 2, rep_begin=7)
    assert txt == "Rep\tGroup\tSize\tDmax\tStat\tNmin\n7\t[0;1]\t2\t0.1\t0.9\t2\n7\t[3;4]\t2\t0.25\t0.75\t1\n"


def test_newick_heights_are_half_the_join_distance():
    merge = np.array([[0, 1], [2, 3]], dtype=np.int32)
    assert pc.newick(merge, [0.5, 2.0], names=["a", "b", "c"]) == "(c:1,(a:0.25,b:0.25):0.75);"


def test_cpp_groups_table_matches_python(adapter_exe, tmp_path):  # noqa: F811
    n, maxsize = 60, 8
    merge, dmax, size, stat, nmin = _tree(n, 5, oc.LINK_AVERAGE)
    coords = np.arange(n) * 3 + 1
    isc = (np.arange(n) % 11 == 0).astype(np.int32)
    f = tmp_path / "tree.bin"
    with open(f, "wb") as fh:
        fh.write(struct.pack("<2i", n, maxsize) + merge.astype(np.int32).tobytes() + dmax.tobytes() + stat.tobytes() +
                 nmin.tobytes() + coords.astype(np.int32).tobytes() + isc.tobytes())
    got = subprocess.run([adapter_exe, "groups", str(f)], capture_output=True, text=True, check=True).stdout
    exp = formats.to_text(formats.write_groups, pc.get_groups(merge, maxsize), coords, isc, dmax, stat, nmin)
    assert got == exp and got.count("\n") > 10
