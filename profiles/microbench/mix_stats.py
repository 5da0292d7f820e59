"""The fill kernel on workload mixes, with a -DDSA_PRUNE_STATS build if DEFUSE_DSA_LIB names one (its [stats] lines go to
stderr): headline (every read crosses the junction), reads wholly inside one window, random decoys, 2x150 bp.
Usage: DEFUSE_DSA_LIB=build_var/lib_stats.so python profiles/microbench/mix_stats.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from defuse_amd import dsa, synth

ctx = dsa.Context(0)
for name, kw in (("headline 2x76", dict()), ("inside 50 %", dict(inside_frac=0.5)), ("inside 100 %", dict(inside_frac=1.0)), ("decoys 50 %", dict(decoy_frac=0.5)),
                 ("headline 2x150, windows 590", dict(lq=150, lr=590)), ("2x150, inside 50 %", dict(lq=150, lr=590, inside_frac=0.5))):
    lq, lr = kw.pop("lq", 76), kw.pop("lr", 389)
    b = synth.make_batch(5000, 100, lq=lq, lr=lr, seed=2, **kw)
    ctx.upload(*b)
    ctx.run()
    sys.stderr.write("== %s\n" % name)
    sys.stderr.flush()
    ctx.plan()
    n = ctx.run()
    t = ctx.timing()
    print("%-32s %d pairs: plan %.3f fill %.3f finish %.3f ms, %.2f records per pair, %.1f M aligns/s in the kernels, %.2f TCUPS in the fill" %
          (name, len(b[3]), t.plan_ms, t.fill_ms, t.finish_ms, n / len(b[3]), len(b[3]) / (t.plan_ms + t.fill_ms + t.finish_ms) / 1e3,
           len(b[3]) * 2.0 * (lr + 1) * (lq + 1) / (t.fill_ms * 1e-3) / 1e12))
    sys.stdout.flush()
