"""Sharding candidate fusions across GPUs (one process per GPU).

Fusions are independent in the split-alignment path (the candidate de-duplication key contains the
fusion, tools/SplitAlignment.cpp:292), so a rank simply takes a subset of the fusions together with
their candidate pairs; no collective is needed on the data path.  The merge restores the order a
single process would have produced: by position of the pair in the original candidate list.
"""
import numpy as np

from .dsa import RECORD_DTYPE


def shard_fusions(fusions, pairs, rank, world):
    """Contiguous fusion ranges balanced by candidate count.  Returns (fusion_lo, fusion_hi)."""
    counts = np.bincount(pairs["fusion_idx"], minlength=len(fusions)).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(counts)])
    total = cum[-1]
    bounds = [int(np.searchsorted(cum, total * r / world, side="left")) for r in range(world + 1)]
    bounds[0], bounds[-1] = 0, len(fusions)
    return bounds[rank], bounds[rank + 1]


def shard_batch(ref_bytes, fusions, read_bytes, pairs, rank, world):
    """The sub-batch of one rank plus the original index of each of its pairs."""
    lo, hi = shard_fusions(fusions, pairs, rank, world)
    sel = np.nonzero((pairs["fusion_idx"] >= lo) & (pairs["fusion_idx"] < hi))[0]
    sub_pairs = pairs[sel].copy()
    sub_pairs["fusion_idx"] -= lo
    return ref_bytes, fusions[lo:hi].copy(), read_bytes, sub_pairs, sel


def merge_records(parts):
    """parts: list of (records, original_pair_index_of_each_local_pair).  Returns the records a single
    process would have produced, pair_idx renumbered to the original candidate list."""
    out = []
    for recs, orig in parts:
        r = recs.copy()
        if len(r):
            r["pair_idx"] = orig[r["pair_idx"]]
        out.append(r)
    allr = np.concatenate(out) if out else np.zeros(0, dtype=RECORD_DTYPE)
    return allr[np.argsort(allr["pair_idx"], kind="stable")]


RECORD_WORDS = RECORD_DTYPE.itemsize // 4


def gather_records(recs, pair_base=0, dst=0, group=None):
    """recs: this rank's records as an int32 tensor [n, RECORD_WORDS] (on the GPU with RCCL, on the CPU with gloo);
    pair_base is added to its pair_idx column so that the gathered records carry job-wide pair numbers.
    Returns (on dst) the records of all ranks in rank order as one tensor plus the per-rank counts, else (None, counts).
    Collectives: one all_gather of the counts, then one group of isend/irecv with exact sizes (no padding)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    assert recs.dtype == torch.int32 and recs.dim() == 2 and recs.shape[1] == RECORD_WORDS
    if pair_base:
        recs = recs.clone()
        recs[:, RECORD_WORDS - 1] += int(pair_base)
    recs = recs.contiguous()
    mine = torch.tensor([recs.shape[0]], dtype=torch.int64, device=recs.device)
    counts = torch.zeros(world, dtype=torch.int64, device=recs.device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts = [int(c) for c in counts.tolist()]
    if rank != dst:
        if counts[rank]:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, recs, dst, group)]):
                w.wait()
        return None, counts
    out = torch.empty((sum(counts), RECORD_WORDS), dtype=torch.int32, device=recs.device)
    starts = np.concatenate([[0], np.cumsum(counts)])
    out[starts[rank]:starts[rank + 1]] = recs
    ops = [dist.P2POp(dist.irecv, out[starts[r]:starts[r + 1]], r, group) for r in range(world) if r != dst and counts[r]]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return out, counts
