"""The N>1 path on CPU: 2 gloo ranks shard the candidate fusions, align their shard (with the CPU
oracle here — the GPU path is covered by the -m gpu tests), gather, and the merge must equal the
single-process result.  Also exercises the barrier / MAX-reduction pattern bench.py uses."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from defuse_amd import shard
    from oracle import dosplitalign_oracle as ora
    from tests import cases
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = cases.mixed_batch(17, n_fusions=9, reads_per_fusion=12, lq=30, lr=(60, 100))
    ref, fus, reads, pairs, orig = shard.shard_batch(*batch, rank=rank, world=world)
    recs = ora.align_batch(ref, fus, reads, pairs)
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                       # bench.py's max-over-ranks timing
    gathered = [None] * world
    dist.all_gather_object(gathered, (recs, orig))
    # the tensor gather bench.py and a multi-GPU caller use (RCCL there, gloo here): records as int32 rows,
    # pair_idx renumbered to the job-wide candidate list before sending
    r = recs.copy()
    if len(r):
        r["pair_idx"] = orig[r["pair_idx"]]
    rows = torch.from_numpy(np.ascontiguousarray(r).view(np.int32).reshape(-1, shard.RECORD_WORDS))
    allrows, counts = shard.gather_records(rows, dst=0)
    if rank == 0:
        merged = shard.merge_records(gathered)
        exp = ora.align_batch(*batch)
        from defuse_amd.dsa import RECORD_DTYPE
        viat = allrows.numpy().view(RECORD_DTYPE).reshape(-1)
        viat = viat[np.argsort(viat["pair_idx"], kind="stable")]
        same = merged.tobytes() == exp.tobytes() and viat.tobytes() == exp.tobytes() and counts == [len(g[0]) for g in gathered]
        q.put((float(t.item()), same, len(exp), [len(g[0]) for g in gathered]))
    else:
        assert allrows is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_merge(built):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tmax, same, n, per_rank = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0 and same and n > 0
    assert all(c > 0 for c in per_rank)                            # both ranks really had work


def test_shard_ranges_cover_everything():
    from defuse_amd import shard
    from tests import cases
    ref, fus, reads, pairs = cases.mixed_batch(3, n_fusions=7, reads_per_fusion=5, lq=20, lr=(40, 60))
    for world in (1, 2, 3, 8):
        seen = np.zeros(len(pairs), dtype=int)
        for r in range(world):
            *_, orig = shard.shard_batch(ref, fus, reads, pairs, r, world)
            seen[orig] += 1
        assert (seen == 1).all()


def _gather_worker(rank, world, port, q, counts):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from defuse_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = counts[rank]
    rows = (torch.arange(n * shard.RECORD_WORDS, dtype=torch.int32).reshape(n, shard.RECORD_WORDS) + 1000000 * rank)
    out, got = shard.gather_records(rows, pair_base=7 * rank, dst=0)
    if rank == 0:
        ok = got == list(counts) and out.shape[0] == sum(counts)
        lo = 0
        for r, c in enumerate(counts):
            exp = torch.arange(c * shard.RECORD_WORDS, dtype=torch.int32).reshape(c, shard.RECORD_WORDS) + 1000000 * r
            exp[:, shard.RECORD_WORDS - 1] += 7 * r
            ok = ok and bool((out[lo:lo + c] == exp).all())
            lo += c
        q.put(ok)
    else:
        assert out is None and got == list(counts)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("counts", [(0, 5), (6, 0), (0, 0), (3, 1000), (4, 0, 9)])
def test_gather_records_with_empty_and_unequal_ranks(counts):
    """The gather of bench.py / a multi-GPU caller with ranks that hold no records at all and with very unequal counts."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, len(counts), port, q, counts)) for r in range(len(counts))]
    for p in procs:
        p.start()
    assert q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
