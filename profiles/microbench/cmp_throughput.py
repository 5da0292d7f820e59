"""Throughput probe of the drop-in clustermatepairs/setcover on a synthetic input shaped like SURVEY.md
section 6's probe (4000 loci x 3-60 pairs, ~125k fragments)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tests import cmp_cases
rng = np.random.default_rng(1)
lines, frag = [], 0
chroms = ["chr%d" % k for k in range(1, 9)]
for l in range(4000):
    ca, cb = rng.choice(chroms, size=2)
    n = int(rng.integers(3, 61))
    ba, bb = int(rng.integers(5000, 200_000_000)), int(rng.integers(5000, 200_000_000))
    lines += cmp_cases.locus_fragments(rng, frag, n, ca, "+-"[int(rng.integers(0, 2))], ba, cb, "+-"[int(rng.integers(0, 2))], bb)
    frag += n
os.makedirs("gpurun_out", exist_ok=True)
p = "gpurun_out/spanning_probe.txt"
open(p, "w").write("".join(lines))
env = dict(os.environ, DEFUSE_TIMING="1")
t0 = time.time()
r = subprocess.run([ROOT + "/bin/clustermatepairs", "-a", p, "-c", "gpurun_out/clusters_probe.txt", "-u", "300", "-s", "30", "-p", "0.95", "-m", "5"],
                   capture_output=True, text=True, env=env)
t1 = time.time()
print(r.stdout.strip().splitlines()[-1], "|", r.stderr.strip(), "| %d fragments in %.2f s -> %.1f k fragments/s" % (frag, t1 - t0, frag / (t1 - t0) / 1e3))
t0 = time.time()
r = subprocess.run([ROOT + "/bin/setcover", "-c", "gpurun_out/clusters_probe.txt", "-m", "5", "-o", "gpurun_out/clusters_probe.sc"], capture_output=True, text=True, env=env)
t1 = time.time()
n = sum(1 for _ in open("gpurun_out/clusters_probe.txt"))
print("setcover:", r.stderr.strip(), "| %d cluster lines in %.2f s" % (n, t1 - t0))
