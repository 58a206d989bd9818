"""On-disk formats either side of the hot path, byte-compatible with what the reference writes so
that its R scripts and `input.vectors.file` restarts keep working:

* substitution vectors `.vec`  -- LegacySubstitutionMappingTools::writeToStream / readFromStream, called at
  CoMap/CoETools.cpp:374-385, 408-412 (format: SURVEY Appendix B.1)
* `output.infos`               -- CoETools::writeInfos, CoMap/CoETools.cpp:496-531
* pairwise `statistics.txt`    -- CoETools::computeIntraStats, CoMap/CoETools.cpp:662-722 (inter: :775-826)
* null `statistics.null.txt`   -- AnalysisTools.cpp:580, 642 (intra), :680, 732 (inter)

Numbers are printed like a default-constructed C++ ostream does (`%g`, 6 significant digits); that formatting is
pinned by the first lines of the reference's own Myo_unif.vec / Myo.infos (tests/golden/myoglobin.npz).
Host-side text only: nothing here touches the GPU."""
import io
import math

import numpy as np


def fmt(x):
    """operator<<(ostream&, double) with default flags and precision 6."""
    x = float(x)
    if math.isnan(x):
        return "nan" if not math.copysign(1.0, x) < 0 else "-nan"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    return "%g" % x


def _open(path_or_file, mode):
    if isinstance(path_or_file, (str, bytes)) or hasattr(path_or_file, "__fspath__"):
        return open(path_or_file, mode), True
    return path_or_file, False


# --------------------------------------------------------------------------------------------------------- .vec
def write_vec(out, counts, branch_lengths, coords, substitution_type=0):
    """counts: [N, B, K] (engine.map_sites()["counts"]); one row per branch = node id (post-order, root last),
    branch length (column labelled "Mean" by the reference), then the type-`substitution_type` count per site."""
    counts = np.asarray(counts)
    N, B, _ = counts.shape
    fh, close = _open(out, "w")
    try:
        fh.write("Branches\tMean" + "".join("\tSite%d" % int(c) for c in coords) + "\n")
        for b in range(B):
            fh.write(str(b) + "\t" + fmt(branch_lengths[b]))
            fh.write("".join("\t" + fmt(v) for v in counts[:, b, substitution_type]))
            fh.write("\n")
    finally:
        if close:
            fh.close()


def read_vec(src):
    """-> dict(coords[N], branch_ids[B], branch_lengths[B], counts[N, B, 1]) (readFromStream, CoETools.cpp:384)."""
    fh, close = _open(src, "r")
    try:
        header = fh.readline().rstrip("\n").split("\t")
        if header[:2] != ["Branches", "Mean"]:
            raise ValueError("not a substitution-vector file: header starts with %r" % header[:2])
        coords = np.array([int(h[4:]) for h in header[2:]], dtype=np.int64)
        ids, bl, rows = [], [], []
        for line in fh:
            f = line.rstrip("\n").split("\t")
            if len(f) < 3:
                continue
            if len(f) != len(header):
                raise ValueError("row of branch %s has %d columns, header has %d" % (f[0], len(f), len(header)))
            ids.append(int(f[0]))
            bl.append(float(f[1]))
            rows.append([float(x) for x in f[2:]])
    finally:
        if close:
            fh.close()
    counts = np.array(rows).T[:, :, None] if rows else np.zeros((len(coords), 0, 1))
    return dict(coords=coords, branch_ids=np.array(ids), branch_lengths=np.array(bl), counts=np.ascontiguousarray(counts))


# --------------------------------------------------------------------------------------------------------- infos
def write_infos(out, coords, is_complete, is_constant, rate_class, post_rate, norm, logl, with_norm=True):
    """CoETools::writeInfos.  with_norm=False reproduces the pre-3.0 layout of the committed Myo.infos (no N column)."""
    fh, close = _open(out, "w")
    try:
        fh.write("Group\tIsComplete\tIsConstant\tRC\tPR" + ("\tN" if with_norm else "") + "\tlogLn\n")
        for i in range(len(coords)):
            f = ["[%d]" % int(coords[i]), str(int(is_complete[i])), str(int(is_constant[i])), str(int(rate_class[i])),
                 fmt(post_rate[i])]
            if with_norm:
                f.append(fmt(norm[i]))
            f.append(fmt(logl[i]))
            fh.write("\t".join(f) + "\n")
    finally:
        if close:
            fh.close()


def site_flags(aln, nstates):
    """SiteTools::isComplete (no gap / unresolved symbol) and SiteTools::isConstant(site, ignoreUnknown=true) for a
    coded alignment [T, N] (codes >= nstates are ambiguity ids), CoETools.cpp:520-521."""
    aln = np.asarray(aln)
    complete = (aln < nstates).all(axis=0)
    const = np.zeros(aln.shape[1], dtype=bool)
    for i in range(aln.shape[1]):
        col = aln[:, i]
        col = col[col < nstates]
        const[i] = len(np.unique(col)) <= 1
    return complete, const


# --------------------------------------------------------------------------------------------------------- pairwise TSV
def write_intra_stats(out, coords, stat, rate_class, post_rate, norm, pvalue=None, nsim=None, min_rate_class=0,
                      min_rate=0.0, max_rate_class_diff=-1, max_rate_diff=-1.0, min_statistic=0.0):
    """statistics.txt of CoETools::computeIntraStats (filters of CoETools.cpp:674-693, row of :698-722).
    stat / pvalue / nsim: dense [N, N]; pvalue NaN is written "NA" with Nsim 0 (:718-720)."""
    n = len(coords)
    fh, close = _open(out, "w")
    rows = 0
    try:
        fh.write("Group\tStat\tRCmin\tPRmin\tNmin" + ("\tPValue\tNsim" if pvalue is not None else "") + "\n")
        for i in range(n):
            if rate_class[i] < min_rate_class or post_rate[i] < min_rate:
                continue
            for j in range(i + 1, n):
                if rate_class[j] < min_rate_class or post_rate[j] < min_rate:
                    continue
                if max_rate_class_diff >= 0 and abs(int(rate_class[j]) - int(rate_class[i])) > max_rate_class_diff:
                    continue
                if max_rate_diff >= 0.0 and abs(post_rate[j] - post_rate[i]) > max_rate_diff:
                    continue
                s = stat[i, j]
                if abs(s) < min_statistic:
                    continue
                f = ["[%d;%d]" % (int(coords[i]), int(coords[j])), fmt(s), str(int(min(rate_class[i], rate_class[j]))),
                     fmt(min(post_rate[i], post_rate[j])), fmt(min(norm[i], norm[j]))]
                if pvalue is not None:
                    if math.isnan(pvalue[i, j]):
                        f += ["NA", "0"]
                    else:
                        f += [fmt(pvalue[i, j]), str(int(nsim[i, j]))]
                fh.write("\t".join(f) + "\n")
                rows += 1
    finally:
        if close:
            fh.close()
    return rows


def write_inter_stats(out, coords1, coords2, stat, rc1, rc2, pr1, pr2, norm1, norm2, independent=False):
    """statistics.txt of CoETools::computeInterStats (CoETools.cpp:775-826), stat dense [N1, N2].
    Nmin is min(norm1[i], norm2[j]): the reference reads norms2[i] there (CoETools.cpp:803), a bug not reproduced."""
    fh, close = _open(out, "w")
    try:
        fh.write("Group\tStat\tRCmin\tPRmin\tNmin\n")
        for i in range(len(coords1)):
            js = [i] if independent else range(len(coords2))
            for j in js:
                fh.write("\t".join(["[%d;%d]" % (int(coords1[i]), int(coords2[j])), fmt(stat[i, j]),
                                    str(int(min(rc1[i], rc2[j]))), fmt(min(pr1[i], pr2[j])),
                                    fmt(min(norm1[i], norm2[j]))]) + "\n")
    finally:
        if close:
            fh.close()


def write_null(out, stat, rcmin, prmin, nmin):
    """statistics.null.txt: header and rows of AnalysisTools.cpp:580, 642 (same columns for the inter null, :680, 732)."""
    fh, close = _open(out, "w")
    try:
        fh.write("Stat\tRCmin\tPRmin\tNmin\n")
        for q in range(len(stat)):
            fh.write(fmt(stat[q]) + "\t" + str(int(rcmin[q])) + "\t" + fmt(prmin[q]) + "\t" + fmt(nmin[q]) + "\n")
    finally:
        if close:
            fh.close()


# --------------------------------------------------------------------------------------------------------- Mica
def write_mica(out, coords, res):
    """Mica's output.file (CoMap/Mica.cpp:634-690).  res: comap_amd.mica.analysis(...)."""
    n = len(coords)
    mi, hj, h, avg, full = res["mi"], res["hjoint"], res["entropy"], res["average_mi"], res["full_average_mi"]
    norms, pv, pp = res.get("norms"), res.get("pvalue"), res.get("perm_pvalue")
    fh, close = _open(out, "w")
    try:
        fh.write("Group\tMI\tAPC\tRCW\tHjoint\tHmin" + ("\tNmin" if norms is not None else "") +
                 ("\tPerm.p.value\tPerm.nb" if pp is not None else "") + ("\tBs.p.value\tBs.nb" if pv is not None else "") + "\n")
        for i in range(n - 1):
            for j in range(i + 1, n):
                f = ["[%d;%d]" % (int(coords[i]), int(coords[j])), fmt(mi[i, j]), fmt(avg[i] * avg[j] / full),
                     fmt(avg[i] * avg[j] / 2.0), fmt(hj[i, j]), fmt(min(h[i], h[j]))]
                if norms is not None:
                    f.append(fmt(min(norms[i], norms[j])))
                if pp is not None:
                    f += [fmt(pp[i, j]), str(int(res["perm_nb"][i, j]))]
                if pv is not None:
                    f += ["NA", "0"] if math.isnan(pv[i, j]) else [fmt(pv[i, j]), str(int(res["nsim"][i, j]))]
                fh.write("\t".join(f) + "\n")
    finally:
        if close:
            fh.close()


# --------------------------------------------------------------------------------------------------------- clustering
def write_groups(out, groups, coords, is_constant, dmax, stat, nmin):
    """clustering.output.groups.file: the DataTable of CoMap/CoMap.cpp:493-548 (no row names, tab separated).
    groups: comap_amd.cluster.get_groups(merge, clustering.maximum_group_size); dmax / stat / nmin: per join."""
    fh, close = _open(out, "w")
    try:
        fh.write("Group\tSize\tIsConstant\tDmax\tStat\tNmin\n")
        for m, mem in groups:
            const = any(bool(is_constant[i]) for i in mem)
            fh.write("\t".join(["[" + ";".join(str(int(coords[i])) for i in mem) + "]", str(len(mem)),
                                "yes" if const else "no", fmt(dmax[m]), fmt(stat[m]), fmt(nmin[m])]) + "\n")
    finally:
        if close:
            fh.close()


def write_cluster_null(out, null, max_group_size, rep_begin=0):
    """clustering.null.output.file: ClusterTools::computeGlobalDistanceDistribution (CoMap/ClusterTools.cpp:219-220,
    :283-289); groups are named by the position of their sites in the simulated data set.
    null: engine.cluster_null(...) (arrays [nrep, nsites-1])."""
    from .cluster import get_groups
    fh, close = _open(out, "w")
    try:
        fh.write("Rep\tGroup\tSize\tDmax\tStat\tNmin\n")
        for k in range(null["merge"].shape[0]):
            for m, mem in get_groups(null["merge"][k], max_group_size):
                fh.write("\t".join([str(rep_begin + k), "[" + ";".join(str(i) for i in mem) + "]", str(len(mem)),
                                    fmt(null["dmax"][k][m]), fmt(null["stat"][k][m]), fmt(null["nmin"][k][m])]) + "\n")
    finally:
        if close:
            fh.close()


def to_text(writer, *args, **kw):
    buf = io.StringIO()
    writer(buf, *args, **kw)
    return buf.getvalue()
