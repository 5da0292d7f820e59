import sys
import numpy as np
sys.path.insert(0, ".")
from defuse_amd import dsa
from oracle import dosplitalign_oracle as ora
from tests import cases
rng = np.random.default_rng(77)
bb = cases.BatchBuilder()
ref0, ref1 = cases.rnd(rng, 389), cases.rnd(rng, 389, b"ACGTN")
f = bb.add_fusion(ref0, ref1)
for r in range(700):
    read = cases.mutate(rng, cases.split_read(rng, ref0, ref1, 76), 0.01)
    if r % 50 == 0:
        b = bytearray(read)
        b[int(rng.integers(0, len(b)))] = ord("N")
        read = bytes(b)
    bb.add_read(f, read)
batch = bb.arrays()
ctx = dsa.Context(0)
for rep in range(2):
    got = ctx.align_batch(*batch)
    exp = ora.align_batch(*batch)
    print("rep", rep, "got", len(got), "exp", len(exp))
    ng = np.bincount(got["pair_idx"], minlength=700); ne = np.bincount(exp["pair_idx"], minlength=700)
    bad = np.nonzero(ng != ne)[0]
    print("pairs with different counts:", len(bad), bad[:40].tolist())
    print("got counts", ng[bad[:20]].tolist(), "exp", ne[bad[:20]].tolist())
    g = set(map(tuple, got.tolist())); e = set(map(tuple, exp.tolist()))
    print("extra", len(g - e), sorted(g - e, key=lambda r: r[-1])[:5])
    print("missing", len(e - g), sorted(e - g, key=lambda r: r[-1])[:5])
