"""Randomised parity stress of the HIP split-read path against the CPU oracle (run by hand on a GPU box:
`python tests/stress_dsa.py [rounds] [seed0]`).  Each round draws the batch geometry at random: reads per
fusion from 1 to 300 (all three fill tiers), read lengths 8..260, windows 30..1400 bases, clean and dirty
alphabets; records must agree byte for byte."""
import sys

import numpy as np

sys.path.insert(0, ".")
from defuse_amd import dsa
from oracle import dosplitalign_oracle as ora
from tests import cases


def borderline(rng, read):
    """Edits whose penalties add up to about the budget 2*Lq - minScore (mismatch 3, inserted base 4, deleted
    base 2 + the lost match): alignments that sit exactly on the pruning threshold, or just beyond it."""
    b = bytearray(read)
    if len(b) < 12:
        return bytes(b)
    budget = 2 * len(b) - int(np.float32(len(b)) * np.float32(2) * 0.90) + int(rng.integers(-1, 2))
    while budget >= 3 and len(b) > 10:
        pos = int(rng.integers(1, len(b) - 1))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            b[pos] = ord("ACGT"[("ACGT".find(chr(b[pos])) + 1) % 4]) if chr(b[pos]) in "ACGT" else ord("A")
            budget -= 3
        elif kind == 1 and budget >= 4:
            b.insert(pos, int(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8))))
            budget -= 4
        else:
            del b[pos]
            budget -= 4
    return bytes(b)


def random_batch(rng):
    bb = cases.BatchBuilder()
    clean = rng.random() < 0.7                        # reads over {A,C,G,T,N} only
    n_fusions = int(rng.integers(1, 40))
    rpf_hi = int(rng.choice([4, 20, 60, 150, 300]))
    lq_hi = int(rng.choice([30, 76, 101, 150, 260]))
    lr_hi = int(rng.choice([120, 400, 600, 1400]))
    for _ in range(n_fusions):
        ref0 = cases.rnd(rng, int(rng.integers(30, lr_hi + 1)), b"ACGTN" if rng.random() < 0.3 else b"ACGT")
        ref1 = cases.rnd(rng, int(rng.integers(30, lr_hi + 1)))
        f = bb.add_fusion(ref0, ref1)
        for _ in range(int(rng.integers(1, rpf_hi + 1))):
            lq = int(rng.integers(8, lq_hi + 1))
            kind = rng.random()
            if kind < 0.25 and len(ref0) > 8 and len(ref1) > 8:
                read = borderline(rng, cases.split_read(rng, ref0, ref1, lq))
            elif kind < 0.7 and len(ref0) > 8 and len(ref1) > 8:
                read = cases.mutate(rng, cases.split_read(rng, ref0, ref1, lq), float(rng.choice([0.0, 0.01, 0.05])))
            elif kind < 0.85:
                read = cases.rnd(rng, lq)
            else:
                read = (ref0 + ref0)[: lq]             # unsplit: the zero-side rule
            if not clean and rng.random() < 0.05:
                read = read.lower()
            if rng.random() < 0.03:
                b = bytearray(read)
                b[int(rng.integers(0, len(b)))] = ord("N")
                read = bytes(b)
            bb.add_read(f, read)
    return bb.arrays()


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    ctx = dsa.Context(0)
    total = 0
    for r in range(rounds):
        rng = np.random.default_rng(seed0 + r)
        batch = random_batch(rng)
        got = ctx.align_batch(*batch)
        exp = ora.align_batch(*batch)
        if len(got) != len(exp) or got.tobytes() != exp.tobytes():
            print("MISMATCH at seed %d: %d vs %d records" % (seed0 + r, len(got), len(exp)))
            sys.exit(1)
        total += len(batch[3])
        if r % 10 == 9:
            print("round %d ok, %d pairs so far" % (r + 1, total), flush=True)
    print("all %d rounds agree (%d pairs)" % (rounds, total))


if __name__ == "__main__":
    main()
