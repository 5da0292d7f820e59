"""Throughput probe of the batched average-linkage clusterer (include/defuse_hc.h): tables of clustered points in the plane.
Usage: python profiles/microbench/hc_throughput.py [n_tables] [items per table]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from defuse_amd import hc


def main():
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    rng = np.random.default_rng(5)
    tabs = []
    for _ in range(nt):
        c = rng.uniform(0, 1000, size=(max(1, n // 8), 2))
        pts = c[rng.integers(0, len(c), size=n)] + rng.normal(0, 15, size=(n, 2))
        tabs.append(np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)))
    hc.cluster_batch(tabs[:8], [60.0] * 8)
    t0 = time.time()
    out, t = hc.cluster_batch(tabs, [60.0] * nt)
    wall = time.time() - t0
    print("%d tables of %d items: kernel %.1f ms, upload %.1f ms, call %.1f ms (wall %.0f ms incl. python unpacking); %d merges, %.2f M merges/s; "
          "%.1f clusters per table" % (nt, n, t.kernel_ms, t.upload_ms, t.total_ms, wall * 1e3, t.n_merges, t.n_merges / t.kernel_ms / 1e3,
                                       np.mean([len(o) for o in out])))


if __name__ == "__main__":
    main()
