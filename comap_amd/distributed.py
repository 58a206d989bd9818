"""Multi-GPU sharding of the parametric-bootstrap null: replicates are independent
(CoMap/AnalysisTools.cpp:587 loop body carries no state besides appending results), so rank r maps the contiguous
replicate range replicate_shard(r, world, nrep) with the counter-based RNG keyed by the GLOBAL replicate index, and
ONE all-gather reassembles the reference's replicate order.  The result is bit-identical for any number of shards.
torch.distributed only (backend "nccl" == RCCL over xGMI on MI355X; "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def replicate_shard(rank, world, nrep):
    """Contiguous, balanced split of [0, nrep): the first nrep % world ranks get one extra replicate."""
    q, r = divmod(nrep, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def gather_null(local_stat, local_nmin, nrep, rep_ram, group=None):
    """All-gather the (stat, nmin) columns of every rank's shard into replicate order.

    local_*: 1-D float64 tensors of (end - begin) * rep_ram entries on the rank's device.  Shards may differ by one
    replicate; they are padded to the largest shard for the collective and trimmed afterwards."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_stat, local_nmin
    world = dist.get_world_size(group)
    sizes = [(e - b) * rep_ram for b, e in (replicate_shard(r, world, nrep) for r in range(world))]
    mx = max(sizes)
    # RCCL moves device memory; gloo (CPU tests, one-GPU rehearsals) wants host tensors
    dev = local_stat.device
    xdev = torch.device("cpu") if dist.get_backend(group) == "gloo" else dev
    send = torch.full((2, mx), float("nan"), dtype=torch.float64, device=xdev)
    send[0, : local_stat.numel()] = local_stat
    send[1, : local_nmin.numel()] = local_nmin
    recv = torch.empty((world, 2, mx), dtype=torch.float64, device=xdev)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    stat = torch.cat([recv[r, 0, : sizes[r]] for r in range(world)]).to(dev)
    nmin = torch.cat([recv[r, 1, : sizes[r]] for r in range(world)]).to(dev)
    return stat, nmin


def gather_shards(local, nitems, group=None):
    """All-gather of per-item rows computed for the contiguous shard replicate_shard(rank, world, nitems): the
    clustering null shards its replicates this way (ClusterTools.cpp:221 loop, independent iterations) and Mica's
    permutation test its column pairs (Mica.cpp:642-648 loop).  local: tensor [items_of_this_rank, ...]; returns
    [nitems, ...] in item order.  One collective; shards may differ by one item."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [e - b for b, e in (replicate_shard(r, world, nitems) for r in range(world))]
    mx = max(sizes)
    dev = local.device
    xdev = torch.device("cpu") if dist.get_backend(group) == "gloo" else dev
    send = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=xdev)
    send[: local.shape[0]] = local
    recv = torch.empty((world,) + tuple(send.shape), dtype=local.dtype, device=xdev)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    return torch.cat([recv[r, : sizes[r]] for r in range(world)]).to(dev)


def row_shard(rank, world, n):
    """Rows [begin, end) of the upper triangle for this rank, balanced by PAIR count (row i holds n - 1 - i pairs): the
    observed pair loop of CoETools.cpp:672-724 split over the ranks.  Contiguous in i, so the ranks' compacted rows
    concatenate to the single-GPU output in the reference's (i, j) order."""
    total = n * (n - 1) // 2

    def first_row_with_prefix_at_least(target):
        # smallest r with pairs(rows < r) >= target; pairs(rows < r) = r (n - 1) - r (r - 1) / 2
        lo, hi = 0, n
        while lo < hi:
            mid = (lo + hi) // 2
            if mid * (n - 1) - mid * (mid - 1) // 2 >= target:
                hi = mid
            else:
                lo = mid + 1
        return lo

    begin = first_row_with_prefix_at_least(total * rank // world) if rank else 0
    end = first_row_with_prefix_at_least(total * (rank + 1) // world) if rank + 1 < world else n
    return begin, end


def mica_rectangle(engine, d_aln1, d_aln2=None, nalpha=20, group=None):
    """Column MI split by blocks of columns of the first alignment over the ranks: every rank holds the alignments (they
    are small), computes MI / Hjoint of its contiguous block of columns against all columns of the second on its own GPU,
    and ONE all-reduce (sum) of the per-column MI sums gives every rank the averages APC / RCW need.

    d_aln2 None -- the reference's case, one alignment against itself (Mica.cpp:346-361, 646-689): the averages follow
    the reference, averageMI_i = sum over j != i of MI_ij / (n - 1) and fullAverageMI = the mean of those.
    d_aln2 given -- an N1 x N2 rectangle between two alignments.  The reference has no such mode (its Mica reads one
    alignment): this is an extension of this engine, and row_mean / col_mean / full_mean are then the plain means over
    the rectangle, by this repo's own definition and not a reference-parity quantity.
    A rank whose block is empty (world > N1) skips the engine call but still enters the all-reduce.
    -> dict(rows=(begin, end), mi, hjoint [rows_local, N2] CUDA, h1 [rows_local], h2 [N2], row_mean [rows_local],
            col_mean [N2], full_mean) -- APC_ij = row_mean[i] * col_mean[j] / full_mean."""
    import torch
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    intra = d_aln2 is None
    other = d_aln1 if intra else d_aln2
    n1, n2 = d_aln1.shape[1], other.shape[1]
    b, e = replicate_shard(rank, world, n1)
    dev = d_aln1.device
    mi = torch.zeros((e - b, n2), dtype=torch.float64, device=dev)
    hj = torch.zeros_like(mi)
    h1 = torch.zeros(e - b, dtype=torch.float64, device=dev)
    h2 = torch.zeros(n2, dtype=torch.float64, device=dev)
    if e > b:
        engine.mi_columns_dev(d_aln1[:, b:e].contiguous(), mi, hj, other, nalpha, None, h1, h2)
    if intra:
        # MI_ii (= H_i) is not part of the reference's averages
        own = torch.arange(b, e, device=dev)
        off = mi.clone()
        off[own - b, own] = 0.0
        col_sum, _ = combine_mica_sums(off.sum(dim=0), group)
        col_mean = col_sum / (n1 - 1)
        return dict(rows=(b, e), mi=mi, hjoint=hj, h1=h1, h2=h2, row_mean=off.sum(dim=1) / (n1 - 1), col_mean=col_mean,
                    full_mean=col_mean.mean())
    col_sum, tot = combine_mica_sums(mi.sum(dim=0), group)
    return dict(rows=(b, e), mi=mi, hjoint=hj, h1=h1, h2=h2, row_mean=mi.mean(dim=1), col_mean=col_sum / n1,
                full_mean=tot / (n1 * n2))


def combine_mica_sums(col_sum_local, group=None):
    """the path's one exchange for Mica: all-reduce (sum) of the per-column MI sums of every rank's row block
    -> (column sums over all rows, grand total)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return col_sum_local, col_sum_local.sum()
    dev = col_sum_local.device
    x = col_sum_local.to("cpu") if dist.get_backend(group) == "gloo" else col_sum_local.clone()
    dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
    x = x.to(dev)
    return x, x.sum()
