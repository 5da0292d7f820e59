"""Seeded candidate batches for the parity tests (shared by CPU and GPU tests)."""
import numpy as np

from defuse_amd.dsa import FUSION_DTYPE, PAIR_DTYPE


class BatchBuilder:
    def __init__(self):
        self.ref = bytearray()
        self.reads = bytearray()
        self.fusions = []
        self.pairs = []

    def add_fusion(self, ref0: bytes, ref1: bytes, fusion_id=None):
        idx = len(self.fusions)
        o0 = len(self.ref)
        self.ref += ref0
        o1 = len(self.ref)
        self.ref += ref1
        self.fusions.append((idx if fusion_id is None else fusion_id, o0, len(ref0), o1, len(ref1)))
        return idx

    def add_read(self, fusion_idx, read: bytes, frag=None, read_end=0, revcomp=0):
        off = len(self.reads)
        self.reads += read
        self.pairs.append((fusion_idx, off, len(read), len(self.pairs) if frag is None else frag, read_end, revcomp, (0, 0)))

    def arrays(self):
        ref = np.frombuffer(bytes(self.ref), dtype=np.uint8).copy()
        reads = np.frombuffer(bytes(self.reads), dtype=np.uint8).copy()
        return ref, np.array(self.fusions, dtype=FUSION_DTYPE), reads, np.array(self.pairs, dtype=PAIR_DTYPE)


def rnd(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n).tolist())


def mutate(rng, s: bytes, rate):
    b = bytearray(s)
    for i in range(len(b)):
        if rng.random() < rate:
            b[i] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8))
    return bytes(b)


def split_read(rng, ref0, ref1, lq, first=None, s1=None, a=None):
    first = int(rng.integers(min(lq, len(ref0)), len(ref0) + 1)) if first is None else first
    s1 = int(rng.integers(0, max(1, len(ref1) - lq + 1))) if s1 is None else s1
    a = int(rng.integers(0, min(lq, first) + 1)) if a is None else a
    return (ref0[first - a:first] + ref1[s1:s1 + lq - a])[:lq]


def mixed_batch(seed, n_fusions=12, reads_per_fusion=40, lq=50, lr=(100, 260)):
    """Split reads, unsplit reads, random reads, Ns, lowercase, variable lengths, odd window sizes."""
    rng = np.random.default_rng(seed)
    bb = BatchBuilder()
    for f in range(n_fusions):
        l0 = int(rng.integers(lr[0], lr[1]))
        l1 = int(rng.integers(lr[0], lr[1]))
        alpha = b"ACGT" if f % 4 else b"ACGTN"
        ref0, ref1 = rnd(rng, l0, alpha), rnd(rng, l1, alpha)
        if f % 5 == 3:
            ref0 = ref0[: l0 // 2] + ref0[: l0 // 2].lower()[::-1] + ref0[l0 // 2:]
        fi = bb.add_fusion(ref0, ref1, fusion_id=1000 + 7 * f)
        n = reads_per_fusion + (int(rng.integers(0, 120)) if f % 3 == 0 else 0)
        for r in range(n):
            kind = rng.integers(0, 10)
            q = lq if kind < 7 else int(rng.integers(0, lq + 20))
            if kind <= 5:
                read = split_read(rng, ref0, ref1, q)
            elif kind == 6:
                p = int(rng.integers(0, max(1, l0 - q)))
                read = ref0[p:p + q]                      # aligns unsplit to window 0
            elif kind == 7:
                p = int(rng.integers(0, max(1, l1 - q)))
                read = ref1[p:p + q]                      # aligns unsplit to window 1
            else:
                read = rnd(rng, q)
            read = mutate(rng, read, 0.02)
            if rng.random() < 0.05 and len(read) > 0:
                b = bytearray(read)
                b[int(rng.integers(0, len(b)))] = ord("N")
                read = bytes(b)
            if rng.random() < 0.03:
                read = read.lower()
            bb.add_read(fi, read, read_end=int(r & 1), revcomp=int((r >> 1) & 1))
    return bb.arrays()


def tie_batch(seed):
    """Low-complexity sequence: homopolymers and tandem repeats around the junction, so that many
    columns and many read splits tie (tools/SplitReadAligner.cpp:233-269 emits all of them)."""
    rng = np.random.default_rng(seed)
    bb = BatchBuilder()
    unit = b"ACG"
    cases = [
        (b"A" * 90, b"A" * 70, [b"A" * 30, b"A" * 12, b"A" * 7]),
        (rnd(rng, 40) + b"T" * 40 + rnd(rng, 50), rnd(rng, 30) + b"T" * 25 + rnd(rng, 60), None),
        (rnd(rng, 30) + unit * 20 + rnd(rng, 40), rnd(rng, 50) + unit * 15 + rnd(rng, 20), None),
        (rnd(rng, 140), rnd(rng, 150), None),
    ]
    for ref0, ref1, reads in cases:
        fi = bb.add_fusion(ref0, ref1)
        if reads is None:
            reads = []
            for _ in range(24):
                q = int(rng.integers(20, 45))
                reads.append(split_read(rng, ref0, ref1, q))
            # a read made of the repeat itself
            reads.append((unit * 12)[:30])
            reads.append(b"T" * 28)
        for r in reads:
            bb.add_read(fi, r)
    # one fusion whose window 1 contains window 0's tail twice (two tied columns on one side)
    core = rnd(rng, 25)
    ref0 = rnd(rng, 60) + core
    ref1 = rnd(rng, 10) + rnd(rng, 30) + rnd(rng, 70)
    fi = bb.add_fusion(ref0 + rnd(rng, 20), ref1)
    dup0 = rnd(rng, 50) + core + rnd(rng, 37) + core + rnd(rng, 11)
    fj = bb.add_fusion(dup0, ref1)
    for s1 in (0, 5, 17):
        bb.add_read(fi, core + ref1[s1:s1 + 20])
        bb.add_read(fj, core + ref1[s1:s1 + 20])
        bb.add_read(fj, core[5:] + ref1[s1:s1 + 25])
    return bb.arrays()


def repeat_batch(seed, n_fusions=40, reads_per_fusion=100, lq=60):
    """Full workgroups of repeat-rich fusions: windows of several tiles that hold a tandem repeat, a duplicated segment or a
    junction on a tile boundary, with a hundred reads each, so that pairs have many kept splits in several tiles (the emit
    paths for many masks, runs of such pairs in one wave, many generic replay tasks, device buffers that overflow and make the
    host re-run the slice) next to ordinary fusions."""
    rng = np.random.default_rng(seed)
    bb = BatchBuilder()
    for f in range(n_fusions):
        kind = f % 5
        if kind == 0:                      # tandem repeat across two tiles of window 0, plain window 1
            unit = rnd(rng, int(rng.integers(2, 7)))
            ref0 = rnd(rng, 40) + (unit * 60)[:150] + rnd(rng, 30)
            ref1 = rnd(rng, 200)
        elif kind == 1:                    # window 1 holds a segment three times, tiles apart
            seg = rnd(rng, 30)
            ref0 = rnd(rng, 230)
            ref1 = seg + rnd(rng, 50) + seg + rnd(rng, 70) + seg + rnd(rng, 20)
        elif kind == 2:                    # junction on a tile boundary, both sides continue alike (several kept splits)
            shared = rnd(rng, 6)
            ref0 = rnd(rng, 128 - 3) + shared + rnd(rng, 80)
            ref1 = shared + rnd(rng, 180)
        elif kind == 3:                    # homopolymer stretches on both sides
            ref0 = rnd(rng, 70) + b"A" * 80 + rnd(rng, 60)
            ref1 = b"A" * 40 + rnd(rng, 160)
        else:
            ref0, ref1 = rnd(rng, 389), rnd(rng, 389)
        fi = bb.add_fusion(ref0, ref1)
        for r in range(reads_per_fusion):
            if kind == 2:
                a = int(rng.integers(8, lq - 8))
                read = ref0[125 + 3 - a:125 + 3] + ref1[3:3 + lq - a] if r % 3 else split_read(rng, ref0, ref1, lq)
            elif kind == 1 and r % 2:
                a = int(rng.integers(6, lq - 20))
                s0 = int(rng.integers(a, len(ref0)))
                read = ref0[s0 - a:s0] + ref1[:lq - a]       # the suffix starts with the repeated segment
            elif kind == 0 and r % 2:
                a = int(rng.integers(10, lq - 10))
                read = ref0[40 + 150 - a:40 + 150] + ref1[:lq - a]      # the prefix lies in the repeat
            elif kind == 3 and r % 2:
                a = int(rng.integers(10, lq - 10))
                read = (b"A" * a) + ref1[40 - (lq - a) // 3:][:lq - a]
            else:
                read = split_read(rng, ref0, ref1, lq)
            if r % 7 == 0:
                read = mutate(rng, read, 0.03)
            bb.add_read(fi, read, read_end=int(r & 1), revcomp=int((r >> 1) & 1))
    return bb.arrays()


def edge_batch():
    """Degenerate shapes: empty read, read shorter than the 4-base anchor, empty window, window
    shorter than one tile, window of exactly one/two tiles, ungrouped pairs."""
    rng = np.random.default_rng(99)
    bb = BatchBuilder()
    r0, r1 = rnd(rng, 64), rnd(rng, 128)
    f0 = bb.add_fusion(r0, r1)
    f1 = bb.add_fusion(rnd(rng, 10), rnd(rng, 200))
    f2 = bb.add_fusion(b"", rnd(rng, 80))
    f3 = bb.add_fusion(rnd(rng, 65), b"")
    f4 = bb.add_fusion(rnd(rng, 63), rnd(rng, 129))
    refs = {f0: (r0, r1)}
    for f in (f0, f1, f2, f3, f4, f0, f4, f1):     # deliberately not grouped by fusion
        bb.add_read(f, b"")
        bb.add_read(f, rnd(rng, 3))
        bb.add_read(f, rnd(rng, 8))
        bb.add_read(f, rnd(rng, 33))
        bb.add_read(f, r0[30:64] + r1[0:30])
        bb.add_read(f, r0[40:64] + r1[100:128])
    return bb.arrays()
