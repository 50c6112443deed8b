"""Multi-GPU plumbing: one process per GPU, reads sharded across ranks, no collective on the per-read path.

The only exchange of the whole path is the pass-1 used-barcode histogram: a dense
uint32 vector indexed by barcode ordinal, summed with one all-reduce (RCCL over xGMI when the tensors live on
the GPUs -- torch.distributed backend "nccl" -- or gloo on CPU tensors in the tests), after which every rank runs
the deterministic host finalize on identical input and so ends with the identical used list (rank 0's result is
broadcast anyway, to make that an invariant rather than an assumption).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import lib as _lib


def shard_range(n_items, rank, world):
    """contiguous shard [lo, hi) of n_items for this rank (reads, files or 10,000-read chunks)"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_histogram(hist, record_count, group=None):
    """hist: int32/int64 tensor [n_keys] (device or CPU); record_count: this rank's number of pass-1 chunks.
    Returns (summed hist, summed record_count).  In-place on `hist`."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
        rc = torch.tensor([int(record_count)], dtype=torch.int64, device=hist.device)
        dist.all_reduce(rc, op=dist.ReduceOp.SUM, group=group)
        record_count = int(rc.item())
    return hist, record_count


def pass1_finalize(hist, sorted_keys, record_count, merge_ed=1, min_count_fold=10, cells_fold_below_max=500, group=None):
    """All-reduce the pass-1 histogram, run the host finalize, broadcast the used list.

    hist[i] counts barcode sorted_keys[i] (ascending key order = the ordinal order of smi_hist_*_device).
    Returns (keys uint64, counts uint32, ranks uint32) -- identical on every rank."""
    hist, record_count = allreduce_histogram(hist, record_count, group)
    if hist.is_cuda:
        # the counters that are not zero are picked on the device: a used list is a few thousand of the 3.6 M counters, and the whole histogram
        # through pageable memory + a host scan for non-zeros cost ~ 6 ms of every finalize
        nz_t = torch.nonzero(hist.detach()).flatten()
        nz = nz_t.cpu().numpy()
        counts = hist.detach()[nz_t].cpu().numpy().astype(np.uint32)
    else:
        h = hist.detach().numpy()
        nz = np.nonzero(h)[0]
        counts = h[nz].astype(np.uint32)
    keys = np.asarray(sorted_keys, dtype=np.uint64)[nz]
    k, c, r = _lib.finalize_used_list(keys, counts, record_count, merge_ed, min_count_fold, cells_fold_below_max)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        n = torch.tensor([k.size], dtype=torch.int64, device=hist.device)
        dist.broadcast(n, src=0, group=group)
        buf = torch.zeros((3, int(n.item())), dtype=torch.int64, device=hist.device)
        if dist.get_rank(group) == 0:
            buf[0] = torch.from_numpy(k.astype(np.int64))
            buf[1] = torch.from_numpy(c.astype(np.int64))
            buf[2] = torch.from_numpy(r.astype(np.int64))
        dist.broadcast(buf, src=0, group=group)
        b = buf.cpu().numpy()
        k, c, r = b[0].astype(np.uint64), b[1].astype(np.uint32), b[2].astype(np.uint32)
    return k, c, r


def assigned_counts_tsv(counts, sorted_keys, max_ed=1, group=None):
    """End of pass 2: sum the per-(barcode, ed) counters of smi_bc_counts_device over the ranks (the second, tiny exchange
    of SURVEY 8e: `BarcodesAssigned.tsv` counters) and format the file on every rank.  counts: [n_keys, 3] int32/uint32
    tensor (host or device)."""
    import torch.distributed as dist

    from . import lib as _lib

    c = counts.to(torch.int32) if counts.dtype != torch.int32 else counts
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
    return _lib.assigned_tsv(sorted_keys, c.cpu().numpy().astype(np.uint32), max_ed=max_ed)
