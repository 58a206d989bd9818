"""The clustering restatement (oracle/cluster.py) against an independent implementation: scipy's linkage.  Merge
order and son order may differ between the two (tie-breaking, bookkeeping), the dendrogram may not: same merge
distances, same member sets."""
import numpy as np
import pytest
from scipy.cluster.hierarchy import linkage
from scipy.spatial.distance import squareform

from oracle import cluster as oc

_SCIPY = {oc.LINK_COMPLETE: "complete", oc.LINK_SINGLE: "single", oc.LINK_AVERAGE: "average"}


def _random_dist(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, 5))
    d = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(-1))
    np.fill_diagonal(d, 0.0)
    return (d + d.T) / 2


def member_sets(merge):
    return sorted(tuple(sorted(m)) for _, m in oc.groups(merge))


def scipy_member_sets(Z, n):
    mem = {i: (i,) for i in range(n)}
    out = []
    for m, row in enumerate(Z):
        mem[n + m] = tuple(sorted(mem[int(row[0])] + mem[int(row[1])]))
        out.append(mem[n + m])
    return sorted(out)


@pytest.mark.parametrize("link", [oc.LINK_COMPLETE, oc.LINK_SINGLE, oc.LINK_AVERAGE])
@pytest.mark.parametrize("n", [2, 3, 17, 120])
def test_hclust_matches_scipy(link, n):
    d = _random_dist(n, 100 + n)
    merge, dmax, size = oc.hclust(d, link)
    Z = linkage(squareform(d, checks=False), _SCIPY[link])
    assert np.allclose(np.sort(dmax), Z[:, 2], rtol=1e-12, atol=0)
    assert member_sets(merge) == scipy_member_sets(Z, n)
    assert size[-1] == n and np.all(np.diff(dmax) >= -1e-12)        # these linkages are monotone


def test_groups_are_listed_sons_first_members_in_son_order():
    # ((0,1),(2,(3,4))) built by hand: merges (0,1) (3,4) (2,n+1) (n+0,n+2)
    merge = np.array([[0, 1], [3, 4], [2, 6], [5, 7]], dtype=np.int32)
    g = oc.groups(merge)
    assert g == [(0, [0, 1]), (1, [3, 4]), (2, [2, 3, 4]), (3, [0, 1, 2, 3, 4])]
    assert oc.groups(merge, max_group_size=2) == [(0, [0, 1]), (1, [3, 4])]


def test_ties_take_the_first_pair_in_index_order():
    d = np.ones((4, 4)) - np.eye(4)
    merge, dmax, _ = oc.hclust(d, oc.LINK_COMPLETE)
    assert merge.tolist() == [[0, 1], [4, 2], [5, 3]] and np.all(dmax == 1.0)


def test_nan_distances_are_never_preferred():
    d = _random_dist(6, 3)
    d[2, :] = d[:, 2] = np.nan
    d[2, 2] = 0
    merge, dmax, _ = oc.hclust(d, oc.LINK_AVERAGE)
    assert 2 in merge[-1] and np.isinf(dmax[-1]) and np.all(np.isfinite(dmax[:-1]))


def test_group_properties():
    rng = np.random.default_rng(5)
    counts = rng.random((9, 7, 2))
    for kind in (oc.DIST_CORRELATION, oc.DIST_COMPENSATION, oc.DIST_EUCLIDIAN):
        d = oc.distance_matrix(kind, counts)
        assert np.array_equal(d, d.T) and np.all(np.diag(d) == 0)
        merge, dmax, size = oc.hclust(d, oc.LINK_COMPLETE)
        stat, nmin = oc.group_properties(kind, merge, dmax, counts)
        norm = np.sqrt((counts.sum(2) ** 2).sum(1))
        for m, mem in oc.groups(merge):
            assert nmin[m] == norm[mem].min()
            pair_d = max(d[a, b] for a in mem for b in mem)        # complete linkage: Dmax is the group's diameter
            assert np.isclose(dmax[m], pair_d, rtol=1e-12)
            if kind == oc.DIST_COMPENSATION:
                sg = counts[mem].sum(2).sum(0)
                assert np.isclose(stat[m], 1 - np.sqrt((sg ** 2).sum()) / norm[mem].sum(), rtol=1e-12)
            elif kind == oc.DIST_EUCLIDIAN:
                assert stat[m] == dmax[m]
            else:
                assert stat[m] == 1 - dmax[m]
