"""Throughput of the split-read path on a batch whose fusions have very different numbers of reads (the
benchmark config has exactly 100 each): the three fill tiers (one table per fusion, split tables, generic)
all take part.  Usage: python profiles/microbench/mixed_sizes.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from defuse_amd import dsa, synth


def concat(batches):
    refs, fuss, readss, pairss = [], [], [], []
    ro = rdo = fo = 0
    for ref, fus, reads, pairs in batches:
        fus = fus.copy(); pairs = pairs.copy()
        fus["ref0_off"] += ro; fus["ref1_off"] += ro
        fus["fusion_id"] += fo
        pairs["read_off"] += rdo; pairs["fusion_idx"] += fo
        refs.append(ref); fuss.append(fus); readss.append(reads); pairss.append(pairs)
        ro += len(ref); rdo += len(reads); fo += len(fus)
    return np.concatenate(refs), np.concatenate(fuss), np.concatenate(readss), np.concatenate(pairss)


def main():
    mix = [(200, 1500), (60, 5000), (25, 8000), (8, 25000)]          # (reads per fusion, fusions)
    batch = concat([synth.make_batch(nf, rpf, seed=10 + k)[:4] for k, (rpf, nf) in enumerate(mix)])
    n = len(batch[3])
    ctx = dsa.Context(0)
    ctx.upload(*batch)
    for _ in range(3):
        ctx.run()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.run()
    dt = (time.perf_counter() - t0) / 10
    t = ctx.timing()
    print("mix %s: %d pairs, %.2f ms per run, %.1f M aligns/s (fill %.2f ms, finish %.2f ms)" % (mix, n, dt * 1e3, n / dt / 1e6, t.fill_ms, t.finish_ms))


if __name__ == "__main__":
    main()
