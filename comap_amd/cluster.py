"""Host side of the clustering analysis: what the reference does with the clustering tree once it
exists.  The tree itself (distance matrix, agglomeration, node properties) comes from the device through
engine.cluster_sites / engine.cluster_null; nothing here computes a distance.

  get_groups   <- ClusterTools::getGroups (CoMap/ClusterTools.cpp:55-113): one group per inner node, sons first, root
                  last; members in son order; filter `clustering.maximum_group_size` as applied by the callers
                  (CoMap.cpp:513, ClusterTools.cpp:271)
  newick       <- the clustering tree written by CoMap.cpp:552-560 (leaves renamed to site coordinates by
                  ClusterTools::translate); node height = Dmax / 2 (Cluster.cpp:83-91 for the branch lengths)
A tree is (merge [n-1, 2], dmax [n-1]): leaves 0..n-1, join m creates node n+m (include/comap_mi355x.h)."""
from .formats import fmt


def get_groups(merge, max_group_size=None):
    """-> list of (join index m, [site indices]) in the reference's output order."""
    n = len(merge) + 1
    out, members = [], {}
    if n < 2:
        return out
    stack = [(2 * n - 2, False)]
    while stack:
        node, seen = stack.pop()
        if node < n:
            members[node] = [node]
        elif not seen:
            a, b = int(merge[node - n][0]), int(merge[node - n][1])
            stack.extend([(node, True), (b, False), (a, False)])
        else:
            a, b = int(merge[node - n][0]), int(merge[node - n][1])
            mem = members.pop(a) + members.pop(b)
            members[node] = mem
            if max_group_size is None or len(mem) <= max_group_size:
                out.append((node - n, mem))
    return out


def group_string(members, names=None):
    """Group::toString (CoMap/ClusterTools.h:96-118): "[a;b;c]"."""
    return "[" + ";".join(str(m if names is None else names[m]) for m in members) + "]"


def newick(merge, dmax, names=None):
    n = len(merge) + 1
    height = [0.0] * n + [float(d) / 2 for d in dmax]
    text = {}
    for m in range(n - 1):          # a node's sons are always older joins
        parts = []
        for son in (int(merge[m][0]), int(merge[m][1])):
            label = text.pop(son) if son >= n else str(son if names is None else names[son])
            parts.append(label + ":" + fmt(height[n + m] - height[son]))
        text[n + m] = "(" + ",".join(parts) + ")"
    return text[2 * n - 2] + ";"
