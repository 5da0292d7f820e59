"""PCIe-inclusive rate of the one-shot entry point (host buffers in, host records out) — reported in
DESIGN.md next to the HBM-resident `value` of bench.py; never used as `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from defuse_amd import dsa, synth
ref, fus, reads, pairs = synth.make_batch(10000, 100, lq=76, lr=389, seed=2)
ctx = dsa.Context(0)
ctx.align_batch(ref, fus, reads, pairs)
t0 = time.perf_counter()
n = 3
for _ in range(n):
    recs = ctx.align_batch(ref, fus, reads, pairs)
dt = (time.perf_counter() - t0) / n
print("one-shot dsa_align_batch: %.1f ms per 1M aligns -> %.1f M aligns/s (records %d, input %.1f MB, output %.1f MB)" %
      (dt * 1e3, len(pairs) / dt / 1e6, len(recs), (ref.nbytes + reads.nbytes + pairs.nbytes + fus.nbytes) / 1e6, recs.nbytes / 1e6))
t0 = time.perf_counter(); ctx.upload(ref, fus, reads, pairs); t1 = time.perf_counter(); ctx.run(); t2 = time.perf_counter(); r = ctx.download(); t3 = time.perf_counter()
print("upload %.1f ms (incl. host task build), run %.1f ms, download %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
