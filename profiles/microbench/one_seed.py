import sys
import numpy as np
sys.path.insert(0, ".")
from defuse_amd import dsa
from oracle import dosplitalign_oracle as ora
from tests import stress_dsa as sd
seed = int(sys.argv[1])
batch = sd.random_batch(np.random.default_rng(seed))
pairs = batch[3]
ctx = dsa.Context(0)
got = ctx.align_batch(*batch)
exp = ora.align_batch(*batch)
print("seed", seed, "got", len(got), "exp", len(exp), "pairs", len(pairs))
ng = np.bincount(got["pair_idx"], minlength=len(pairs)); ne = np.bincount(exp["pair_idx"], minlength=len(pairs))
bad = np.nonzero(ng != ne)[0]
print("bad pairs", len(bad))
for p in bad[:30]:
    g = got[got["pair_idx"] == p]; e = exp[exp["pair_idx"] == p]
    print("pair", p, "wg", p // 256, "lane", p % 256, "fusion", pairs[p]["fusion_idx"], "lq", pairs[p]["read_len"], "got", len(g), "exp", len(e),
          "exp (a, i1, i2):", [(int(r["read_first"]), int(r["ref_first"]), int(r["ref_second"])) for r in e][:4],
          "got distinct i1", len(np.unique(g["ref_first"])), "i2", len(np.unique(g["ref_second"])))
