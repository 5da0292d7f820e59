"""Records per candidate of the bench workload (BASELINE configs[1]): the emit kernels work one lane per candidate, so the
candidates with the most records set their duration.  Run on a GPU box."""
import sys

import numpy as np

sys.path.insert(0, ".")
from defuse_amd import dsa, synth

ref, fus, reads, pairs = synth.make_batch(10000, 100, lq=76, lr=389, seed=2)
ctx = dsa.Context(0)
ctx.upload(ref, fus, reads, pairs)
ctx.run()
rec = ctx.download()
n = np.bincount(rec["pair_idx"], minlength=len(pairs))
print("pairs", len(pairs), "records", len(rec), "max per pair", n.max())
for q in (50, 90, 99, 99.9, 99.99):
    print("percentile", q, np.percentile(n, q))
big = np.argsort(n)[-5:]
for p in big:
    r = rec[rec["pair_idx"] == p]
    print("pair", p, "records", n[p], "distinct read splits", len(np.unique(r["read_first"])), "distinct ref_first", len(np.unique(r["ref_first"])),
          "distinct ref_second", len(np.unique(r["ref_second"])))
hist = np.bincount(np.minimum(n, 40))
print("histogram (last bin = 40+):", hist.tolist())
