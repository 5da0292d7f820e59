import sys
import numpy as np
sys.path.insert(0, ".")
from defuse_amd import dsa, synth
ref, fus, reads, pairs = synth.make_batch(2000, 100, lq=76, lr=389, seed=2)
ctx = dsa.Context(0)
ctx.upload(ref, fus, reads, pairs)
ctx.run()
rec = ctx.download()
# same pair, same read_first, different ref_first tile
order = np.lexsort((rec["ref_first"], rec["read_first"], rec["pair_idx"]))
r = rec[order]
same = (r["pair_idx"][1:] == r["pair_idx"][:-1]) & (r["read_first"][1:] == r["read_first"][:-1])
d = r["ref_first"][1:] - r["ref_first"][:-1]
tile_a = (r["ref_first"][:-1] - 1) // 64
tile_b = (r["ref_first"][1:] - 1) // 64
sel = same & (tile_a != tile_b) & (r["read_first"][:-1] >= 16) & (r["read_first"][:-1] <= 60)
print("ties across tiles at read_first in [16,60]:", sel.sum())
idx = np.nonzero(sel)[0][:12]
for i in idx:
    print(r[i], r[i + 1])
vals, cnt = np.unique(d[sel], return_counts=True)
print(list(zip(vals.tolist(), cnt.tolist()))[:20])
