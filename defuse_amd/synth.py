"""Synthetic candidate batches for the split-read DP (BASELINE.json configs).

make_batch() builds what dosplitalign's candidate enumeration would hand to the aligner for a set
of synthetic fusions: per fusion two reference windows of `lr` random bases and `reads_per_fusion`
reads of `lq` bases that cross the fusion junction at read offset 4..lq-4, with 1 % substitutions
and 0.5 % of the reads carrying one 'N'.  Deterministic in (seed, sizes); numpy only.
"""
import numpy as np

from .dsa import FUSION_DTYPE, PAIR_DTYPE

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def window_length(ufrag, sfrag, min_read, max_read, region_len):
    """Lr of SURVEY 8(a): breakLen + maxRead with the reference's integer arithmetic
    (tools/SplitAlignment.cpp:37-38,61-76,637-655)."""
    max_frag = int(ufrag + 3 * sfrag)
    push = min(max_read, int(0.5 * region_len))
    return max_frag - region_len - min_read + 2 * push + max_read


def make_batch(n_fusions, reads_per_fusion, lq=76, lr=389, seed=2, sub_rate=0.01, n_rate=0.005,
               decoy_frac=0.0, inside_frac=0.0):
    """Returns (ref_bytes, fusions, read_bytes, pairs) as numpy arrays in the C-ABI layout.

    decoy_frac: fraction of reads replaced by random sequence (candidates that do not align);
    inside_frac: fraction of reads that lie wholly inside one of the two windows, nowhere near the junction — what most of the
    mates DoAlignment enumerates look like (tools/SplitAlignment.cpp:266-303 takes every mate whose partner falls in a mate
    region): one matrix scores 2 * lq, the other side stays below the split minimum (the zero-side rule)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    F, P = int(n_fusions), int(reads_per_fusion)
    ref = _ACGT[rng.integers(0, 4, size=(F, 2, lr), dtype=np.uint8)]
    ref_bytes = ref.reshape(-1)
    fusions = np.zeros(F, dtype=FUSION_DTYPE)
    fusions["fusion_id"] = np.arange(F, dtype=np.int32)
    fusions["ref0_off"] = np.arange(F, dtype=np.int64) * 2 * lr
    fusions["ref0_len"] = lr
    fusions["ref1_off"] = np.arange(F, dtype=np.int64) * 2 * lr + lr
    fusions["ref1_len"] = lr
    # junction: read = ref0[first-a:first] + ref1[s1:s1+lq-a]
    first = rng.integers(lq, lr, size=F)               # prefix length of window 0 kept by the fusion
    s1 = rng.integers(0, lr - lq + 1, size=F)          # first base of window 1 kept by the fusion
    n = F * P
    if n * lq >= 2 ** 31 or F * 2 * lr >= 2 ** 31:
        raise ValueError("one batch holds less than 2 GiB of read bytes and of window bytes (32-bit offsets in dsa_pair / dsa_fusion, "
                         "include/defuse_dsa.h): split %d x %d reads of %d bases into several batches, as bin/dosplitalign does" % (F, P, lq))
    fidx = np.repeat(np.arange(F, dtype=np.int64), P)
    a = rng.integers(4, lq - 4 + 1, size=n)
    k = np.arange(lq, dtype=np.int64)[None, :]
    left = k < a[:, None]
    src = np.where(left,
                   (fidx * 2 * lr + first[fidx] - a)[:, None] + k,
                   (fidx * 2 * lr + lr + s1[fidx] - a)[:, None] + k)
    reads = ref_bytes[src]
    sub = rng.random(size=reads.shape) < sub_rate
    if sub.any():
        cur = np.searchsorted(_ACGT, reads[sub])  # A,C,G,T are sorted ascending in ASCII
        reads[sub] = _ACGT[(cur + rng.integers(1, 4, size=cur.size)) % 4]
    has_n = rng.random(size=n) < n_rate
    if has_n.any():
        rows = np.nonzero(has_n)[0]
        reads[rows, rng.integers(0, lq, size=rows.size)] = ord("N")
    if inside_frac > 0:
        ins = np.nonzero(rng.random(size=n) < inside_frac)[0]
        side = rng.integers(0, 2, size=ins.size)
        start = rng.integers(0, lr - lq + 1, size=ins.size)
        reads[ins] = ref_bytes[(fidx[ins] * 2 * lr + side * lr + start)[:, None] + k]
    if decoy_frac > 0:
        dec = np.nonzero(rng.random(size=n) < decoy_frac)[0]
        reads[dec] = _ACGT[rng.integers(0, 4, size=(dec.size, lq), dtype=np.uint8)]
    pairs = np.zeros(n, dtype=PAIR_DTYPE)
    pairs["fusion_idx"] = fidx
    pairs["read_off"] = np.arange(n, dtype=np.int64) * lq
    pairs["read_len"] = lq
    pairs["frag"] = np.arange(n, dtype=np.int32)
    pairs["read_end"] = (np.arange(n) & 1).astype(np.uint8)
    pairs["revcomp"] = ((np.arange(n) >> 1) & 1).astype(np.uint8)
    return ref_bytes.copy(), fusions, reads.reshape(-1).copy(), pairs


def cells_per_align(lq, lr):
    return 2 * (lr + 1) * (lq + 1)


def algorithmic_bytes_per_align(lq, lr, reads_per_fusion, records_per_align):
    """2-bit bases + 1-bit non-ACGT mask, references amortised over the reads of a
    fusion, a 12-byte work descriptor in and one 36-byte record out per emitted alignment."""
    c4 = lambda x: -(-x // 4)
    c8 = lambda x: -(-x // 8)
    return c4(lq) + c8(lq) + 2.0 * (c4(lr) + c8(lr)) / reads_per_fusion + 12 + 36.0 * records_per_align


def make_batch_device(n_fusions, reads_per_fusion, lq, lr, seed, device, fusion_id_base=0, sub_rate=0.01, n_rate=0.005):
    """The same recipe as make_batch, drawn on the GPU with torch (plumbing: a share of BASELINE configs[3] is tens of
    millions of reads, which numpy index arrays would take minutes and tens of GB of host memory to build), returned as
    host numpy arrays in the C-ABI layout.  Not bit-identical to make_batch (different generator); deterministic in
    (seed, sizes).  Reads are drawn in chunks so the index tensors stay small."""
    import torch
    F, P = int(n_fusions), int(reads_per_fusion)
    n = F * P
    if n * lq >= 2 ** 31 or F * 2 * lr >= 2 ** 31:
        raise ValueError("one upload holds less than 2 GiB of read bytes and of window bytes (32-bit offsets, include/defuse_dsa.h)")
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    ref = torch.randint(0, 4, (F * 2 * lr,), generator=g, device=device, dtype=torch.uint8)
    first = torch.randint(lq, lr, (F,), generator=g, device=device)
    s1 = torch.randint(0, lr - lq + 1, (F,), generator=g, device=device)
    reads = torch.empty((n, lq), dtype=torch.uint8, device=device)
    k = torch.arange(lq, device=device)[None, :]
    step = max(1, (1 << 22) // P) * P                      # whole fusions per chunk, about 4 M reads
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        m = hi - lo
        fidx = torch.arange(lo, hi, device=device) // P
        a = torch.randint(4, lq - 4 + 1, (m,), generator=g, device=device)
        base0 = fidx * (2 * lr) + first[fidx] - a
        base1 = fidx * (2 * lr) + lr + s1[fidx] - a
        src = torch.where(k < a[:, None], base0[:, None] + k, base1[:, None] + k)
        code = ref[src]
        sub = torch.rand((m, lq), generator=g, device=device) < sub_rate
        shift = torch.randint(1, 4, (m, lq), generator=g, device=device, dtype=torch.uint8)
        code = torch.where(sub, (code + shift) % 4, code)
        chunk = acgt[code.long()]
        has_n = torch.rand((m,), generator=g, device=device) < n_rate
        pos = torch.randint(0, lq, (m,), generator=g, device=device)
        rows = torch.nonzero(has_n)[:, 0]
        chunk[rows, pos[rows]] = ord("N")
        reads[lo:hi] = chunk
        del src, code, sub, shift, chunk
    ref_bytes = acgt[ref.long()].cpu().numpy()
    read_bytes = reads.reshape(-1).cpu().numpy()
    del reads, ref
    torch.cuda.empty_cache()
    fusions = np.zeros(F, dtype=FUSION_DTYPE)
    fusions["fusion_id"] = fusion_id_base + np.arange(F, dtype=np.int64)
    fusions["ref0_off"] = np.arange(F, dtype=np.int64) * 2 * lr
    fusions["ref0_len"] = lr
    fusions["ref1_off"] = np.arange(F, dtype=np.int64) * 2 * lr + lr
    fusions["ref1_len"] = lr
    pairs = np.zeros(n, dtype=PAIR_DTYPE)
    pairs["fusion_idx"] = np.repeat(np.arange(F, dtype=np.int32), P)
    pairs["read_off"] = np.arange(n, dtype=np.int64) * lq
    pairs["read_len"] = lq
    pairs["frag"] = (np.arange(n, dtype=np.int64) & 0x7FFFFFFF).astype(np.int32)
    pairs["read_end"] = (np.arange(n) & 1).astype(np.uint8)
    pairs["revcomp"] = ((np.arange(n) >> 1) & 1).astype(np.uint8)
    return ref_bytes, fusions, read_bytes, pairs
