"""Where do the EM iteration counts of the device and of the C restatement differ?  BASELINE configs[2] at 1 M fragments: the
tool's host stages dump the arrays they hand to the device, mpe_cluster_batch and oracle/mpe_oracle.c run on them, and the
per-problem, per-K iteration counts are compared (DEFUSE_MPE_DUMP_ITERS / ora_mpe_diag.iters_by_k).
    gpurun -- python3 profiles/microbench/em_iterations.py [fragments]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from defuse_amd import build, mpe           # noqa: E402
from oracle import mpe_c                    # noqa: E402
from tests import cmp_cases                 # noqa: E402
from tests.mpe_dump import read_em_dump     # noqa: E402

n_frag = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
out = "/tmp/em_iterations"            # large intermediates: not under gpurun_out (only the log travels back)
os.makedirs(out, exist_ok=True)
build.build_tools()
build.build_oracle()
span, dump, its = out + "/spanning.txt", out + "/em.bin", out + "/iters.bin"
cmp_cases.config3_write(n_frag, span)
r = subprocess.run([os.path.join(ROOT, "bin", "clustermatepairs"), "-a", span, "-c", out + "/unused", "-u", "300", "-s", "30", "-p", "0.95", "-m", "5"],
                   capture_output=True, text=True, env=dict(os.environ, DEFUSE_CMP_DUMP_EM=dump))
assert r.returncode == 0, r.stderr
d = read_em_dump(dump)
args = (d["mean"], d["sd"], d["min_prob"], d["min_size"], d["prob_off"], d["x"], d["y"], d["u"], d["to_xo"], d["to_yo"])
os.environ["DEFUSE_MPE_DUMP_ITERS"] = its
g_ncl, g_member, g_status, t = mpe.cluster_batch(*args)
o_ncl, o_member, o_status, dg, pd = mpe_c.cluster_batch(*args, per_problem_diag=True)
n = len(d["prob_off"]) - 1
raw = np.fromfile(its, dtype=np.int64)
dev = raw[:n * 12].reshape(n, 12)
dev_ll = raw[n * 12:].view(np.float64).reshape(n, 12)
ora = np.array([list(pd[p].iters_by_k) for p in range(n)], dtype=np.int64)
ora_ll = np.array([list(pd[p].ll_by_k) for p in range(n)], dtype=np.float64)
print("problems %d, mate pairs %d; device iterations %d, restatement %d; memberships equal: %s" % (
    n, len(d["x"]), t.em_iterations, dg.em_iterations, g_member.tobytes() == o_member.tobytes()))
print("sum of the dumped device counts (fits + refit): %d" % int(dev[:, 1:].sum()))
diff = np.nonzero((dev != ora).any(axis=1))[0]
print("problems whose counts differ: %d" % len(diff))
sizes = np.diff(d["prob_off"])
for p in diff[:40]:
    ks = [k for k in range(1, 12) if dev[p, k] != ora[p, k]]
    print("  problem %d  N %d  chosen K dev %d ora %d  differing K %s  dev %s  ora %s" % (
        p, sizes[p], dev[p, 0], ora[p, 0], ks, [int(dev[p, k]) for k in ks], [int(ora[p, k]) for k in ks]))
    for k in ks:
        if k <= 10:
            print("      K %d: log-likelihood at the end dev %.12g ora %.12g (relative difference %.2e)" % (
                k, dev_ll[p, k], ora_ll[p, k], abs(dev_ll[p, k] - ora_ll[p, k]) / max(1e-300, abs(ora_ll[p, k]))))
same = (dev == ora).all(axis=1)
both = (dev_ll != 0) & (ora_ll != 0)
rel = np.abs(dev_ll - ora_ll)[both] / np.abs(ora_ll[both])
print("log-likelihoods at the end of all fits that gave one: largest relative difference %.3e (fits of the differing problems included)" % rel.max())
print("chosen K equal everywhere: %s; iteration counts of the chosen K's fit and of the refit equal everywhere: %s" % (
    bool((dev[:, 0] == ora[:, 0]).all()),
    bool(all(dev[p, dev[p, 0]] == ora[p, ora[p, 0]] and dev[p, 11] == ora[p, 11] for p in range(n) if dev[p, 0] >= 1))))
print("differing fits have more components than the chosen K in every case: %s" % bool(all(k > dev[p, 0] for p in diff for k in range(1, 11) if dev[p, k] != ora[p, k])))
