"""Throughput probe of the localalign scorer (include/defuse_la.h) on a pipeline-like batch:
references of 2001 bases (dna_concordant_len window), sequences of 100-400 bases, -m 10 -x -5 -g -5.
Usage: python profiles/microbench/la_throughput.py [n_pairs] [concordant fraction, default 0.3]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from defuse_amd import la
from oracle import localalign_oracle as o


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    rng = np.random.default_rng(7)
    al = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = rng.choice(al, size=4_000_000)
    pairs = []
    for _ in range(n):
        a = int(rng.integers(0, len(genome) - 2001))
        ls = int(rng.integers(100, 401))
        if rng.random() < frac:                     # concordant: the other end lies inside the window
            b = a + int(rng.integers(0, 2001 - ls))
        else:
            b = int(rng.integers(0, len(genome) - ls))
        pairs.append((genome[a:a + 2001].tobytes(), genome[b:b + ls].tobytes()))
    la.align_batch(pairs[:1000], 10, -5, -5)        # warm-up
    t0 = time.time()
    scores, t = la.align_batch(pairs, 10, -5, -5)
    wall = time.time() - t0
    print("pairs %d  cells %.3e  kernel %.1f ms  pack %.1f ms  call %.1f ms (host prep + H2D included; wall %.1f ms)"
          % (n, t.cells, t.kernel_ms, t.pack_ms, t.total_ms, wall * 1e3))
    print("kernel: %.1f GCUPS   %.2f M pairs/s" % (t.cells / t.kernel_ms / 1e6, n / t.kernel_ms / 1e3))
    need = np.array([int(np.ceil(0.8 * 10 * len(s))) for _, s in pairs], dtype=np.int32)     # the pipeline's -t 0.8
    s2, t2 = la.align_batch(pairs, 10, -5, -5, min_score=need)
    keep = scores >= need
    assert np.array_equal(s2[keep], scores[keep]) and np.all(s2[~keep] < need[~keep])
    print("with the -t 0.8 minimum: kernel %.1f ms, %.2f M pairs/s, %d of %d pairs reach it"
          % (t2.kernel_ms, n / t2.kernel_ms / 1e3, int(keep.sum()), n))
    k = 40
    t0 = time.time()
    want = [o.simple_align(10, -5, -5, r, s) for r, s in pairs[:k]]
    dt = time.time() - t0
    cells = sum((len(r) + 1) * (len(s) + 1) for r, s in pairs[:k])
    assert list(scores[:k]) == want
    print("oracle (1 thread): %.3f GCUPS on %d pairs" % (cells / dt / 1e9, k))


if __name__ == "__main__":
    main()
