"""TEST INFRASTRUCTURE ONLY: CPU restatement of CoMap's clustering analysis.  Never imported by
comap_amd/.

What it follows
* distance matrix: CoMap/CoMap.cpp:432-440 and ClusterTools.cpp:242-252 with the distances of CoMap.cpp:402-428
  (Distance.h: EuclidianDistance :161-181, StatisticBasedDistance(cor, 1.) :321-336, CompensationDistance :376-385);
* agglomeration: bpp::HierarchicalClustering (bpp-phyl >= 3.0.0, Bpp/Phyl/Distance/HierarchicalClustering.{h,cpp},
  ABSENT from /root/reference) as CoMap.cpp:460-472 constructs it for "complete", "single" and "average".  Its loop
  is AbstractAgglomerativeDistanceMethod::computeTree: take the closest pair, join it under a new node placed at
  half the pair's distance, replace the first member's row by the linkage update, drop the second.  The pair scan
  and the son order are restated from the reference's own subclass of the same base, SumClustering
  (CoMap/Cluster.cpp:55-79 getBestPair: i < j in index order, strict `<`, first minimum wins; :126-139
  getParentNode(son1 = first member, son2 = second)).
* groups and their properties: ClusterTools::getGroups (ClusterTools.cpp:61-113, one group per inner node, sons
  first; members in son order), "Nmin" ClusterTools.cpp:302-320, "Stat" Distance.h:113-126 / :353-366 / :393-413,
  "Dmax" = 2 * height (CoMap.cpp:541, ClusterTools.cpp:283);
* the null: ClusterTools::computeGlobalDistanceDistribution (ClusterTools.cpp:200-294).

PARITY UNPINNED: the reference ships no clustering output, and HierarchicalClustering itself is not in the tree, so
tie-breaking and son order rest on the SumClustering analogy above.  They only matter when two distances are equal
to the last bit.  hclust() is cross-checked against scipy.cluster.hierarchy.linkage in tests/test_cluster_oracle.py.

Representation: leaves are 0..n-1, the node made by merge m is n+m (scipy's convention); merge[m] = (son1, son2)."""
import numpy as np

import oracle

DIST_CORRELATION, DIST_COMPENSATION, DIST_EUCLIDIAN = 0, 1, 2
LINK_COMPLETE, LINK_SINGLE, LINK_AVERAGE = 0, 1, 2

_STAT_OF_DIST = {DIST_CORRELATION: oracle.ST_CORRELATION, DIST_COMPENSATION: oracle.ST_COMPENSATION,
                 DIST_EUCLIDIAN: oracle.ST_EUCLIDIAN_DISTANCE}


def distance_matrix(dist_kind, counts):
    """counts [N][B][K] -> symmetric [N][N], zero diagonal."""
    s = oracle.pair_stats_intra(_STAT_OF_DIST[dist_kind], counts)
    n = s.shape[0]
    iu = np.triu_indices(n, 1)
    d = np.zeros((n, n))
    d[iu] = s[iu] if dist_kind == DIST_EUCLIDIAN else 1.0 - s[iu]
    d.T[iu] = d[iu]
    return d


def hclust(dist, linkage):
    """-> merge int32 [n-1][2], dmax float64 [n-1] (distance of the joined pair = 2 * node height), size int32 [n-1].
    NaN distances are treated as +inf (never preferred; the reference leaves that case undefined)."""
    D = np.array(dist, dtype=np.float64, copy=True)
    n = D.shape[0]
    D[np.isnan(D)] = np.inf
    cid = np.arange(n)
    csz = np.ones(n, dtype=np.int64)
    active = np.ones(n, dtype=bool)
    merge = np.zeros((max(n - 1, 0), 2), dtype=np.int32)
    dmax = np.zeros(max(n - 1, 0))
    size = np.zeros(max(n - 1, 0), dtype=np.int32)
    iu = np.triu(np.ones((n, n), dtype=bool), 1)
    for m in range(n - 1):
        ok = iu & active[:, None] & active[None, :]
        masked = np.where(ok, D, np.inf)
        flat = int(np.argmin(masked))               # first minimum in row-major order == the reference's scan
        if not ok.flat[flat]:                       # everything left is +inf: first active pair
            flat = int(np.flatnonzero(ok.ravel())[0])
        i, j = divmod(flat, n)
        merge[m] = (cid[i], cid[j])
        dmax[m] = D[i, j]
        ni, nj = float(csz[i]), float(csz[j])
        if linkage == LINK_COMPLETE:
            new = np.maximum(D[i], D[j])
        elif linkage == LINK_SINGLE:
            new = np.minimum(D[i], D[j])
        else:
            new = (ni * D[i] + nj * D[j]) / (ni + nj)
        keep = active.copy()
        keep[[i, j]] = False
        D[i, keep] = new[keep]
        D[keep, i] = new[keep]
        active[j] = False
        cid[i] = n + m
        csz[i] += csz[j]
        size[m] = csz[i]
    return merge, dmax, size


def group_properties(dist_kind, merge, dmax, counts):
    """"Stat" and "Nmin" of every inner node.  counts [N][B][K]."""
    n = counts.shape[0]
    sigma = counts.sum(axis=2)                                   # total substitution vector per branch
    norm = np.sqrt((sigma * sigma).sum(axis=1))                  # computeNormForSite
    nm = np.concatenate([norm, np.zeros(n - 1)])
    sg = np.concatenate([sigma, np.zeros((n - 1, sigma.shape[1]))])
    sn = nm.copy()
    stat = np.zeros(n - 1)
    for m in range(n - 1):
        a, b = merge[m]
        nm[n + m] = min(nm[a], nm[b])
        if dist_kind == DIST_COMPENSATION:
            sg[n + m] = sg[a] + sg[b]
            sn[n + m] = sn[a] + sn[b]
            stat[m] = 1.0 - np.sqrt((sg[n + m] ** 2).sum()) / sn[n + m]
        elif dist_kind == DIST_EUCLIDIAN:
            stat[m] = dmax[m]
        else:
            stat[m] = 1.0 - dmax[m]
    return stat, nm[n:]


def groups(merge, size=None, max_group_size=None):
    """ClusterTools::getGroups order: (node index m, member list) per inner node, sons first, root last."""
    n = len(merge) + 1
    out = []
    if n < 2:
        return out
    members = {}
    stack = [(2 * n - 2, False)]
    while stack:
        node, done = stack.pop()
        if node < n:
            members[node] = [node]
            continue
        a, b = merge[node - n]
        if not done:
            stack.append((node, True))
            stack.append((int(b), False))
            stack.append((int(a), False))
        else:
            mem = members.pop(int(a)) + members.pop(int(b))
            members[node] = mem
            if max_group_size is None or len(mem) <= max_group_size:
                out.append((node - n, list(mem)))
    return out


def cluster_null(model, dist_kind, linkage, seed, rep_begin, rep_end, nsites):
    """computeGlobalDistanceDistribution: replicate k simulates sites k*nsites .. (k+1)*nsites-1 of the counter RNG."""
    res = []
    for k in range(rep_begin, rep_end):
        aln, _ = oracle.simulate(model, seed, k * nsites, nsites)
        mp = oracle.map_sites(model, aln)
        d = distance_matrix(dist_kind, mp["counts"])
        merge, dmax, size = hclust(d, linkage)
        stat, nmin = group_properties(dist_kind, merge, dmax, mp["counts"])
        res.append(dict(merge=merge, dmax=dmax, size=size, stat=stat, nmin=nmin, dist=d))
    return res
