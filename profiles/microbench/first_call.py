"""What the first calls into the library cost a fresh process, against the same calls later (the drop-in tools are
short-lived processes: their fixed costs are these).  Usage: python profiles/microbench/first_call.py [pairs per batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

t0 = time.perf_counter()
from defuse_amd import dsa, synth
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
nf = npairs // 100
ref, fus, reads, pairs = synth.make_batch(nf, 100, lq=76, lr=389, seed=3)
t = time.perf_counter()
dsa.load_library()
print("dlopen of the library (+ HIP runtime): %.1f ms" % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter()
ctx = dsa.Context(0)
print("dsa_create: %.1f ms" % (1e3 * (time.perf_counter() - t)))
out = np.zeros(2 * len(pairs) + 1024, dtype=dsa.RECORD_DTYPE)
for k in range(4):
    t = time.perf_counter()
    n = ctx.align_batch_into(ref, fus, reads, pairs, out)
    tm = ctx.timing()
    print("dsa_align_batch #%d (%d pairs, pageable buffers): %.2f ms (plan %.2f fill %.2f finish %.2f of it on the device), %d records" %
          (k + 1, len(pairs), 1e3 * (time.perf_counter() - t), tm.plan_ms, tm.fill_ms, tm.finish_ms, n))
t = time.perf_counter()
st = dsa.Stream(0, depth=3)
print("dsa_stream_create(depth 3): %.1f ms" % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter()
pins = [[dsa.pinned_copy(a) for a in (ref, fus, reads, pairs)] for _ in range(3)]
pouts = [dsa.PinnedArray((2 * len(pairs) + 1024,), dsa.RECORD_DTYPE) for _ in range(3)]
print("pinned buffers of 3 slots (%.0f MB in all, with the copies into them): %.1f ms" %
      (3 * (ref.nbytes + fus.nbytes + reads.nbytes + pairs.nbytes + pouts[0].array.nbytes) / 1e6, 1e3 * (time.perf_counter() - t)))
for rnd in range(3):
    t = time.perf_counter()
    sub = 0
    for k in range(8):
        while sub < 8 and sub - k < 3:
            st.submit(*[p.array for p in pins[sub % 3]], pouts[sub % 3].array)
            sub += 1
        st.collect()
    print("round %d: 8 batches of %d pairs through the stream: %.2f ms (%.2f ms per batch)" % (rnd + 1, len(pairs), 1e3 * (time.perf_counter() - t), 1e3 * (time.perf_counter() - t) / 8))
print("all of the above from the import on: %.2f s" % (time.perf_counter() - t0))
