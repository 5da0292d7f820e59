"""The EM kernel alone on the config-3 probe: the host stages of bin/clustermatepairs dump the arrays they hand to the device
(DEFUSE_CMP_DUMP_EM), mpe_cluster_batch runs on them a few times; prints kernel_ms (and, with a -DMPE_PHASE_STATS build named by
DEFUSE_DSA_LIB, the wave cycles per phase on stderr).
    gpurun -- python3 profiles/microbench/em_probe.py [fragments] [repeats]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from defuse_amd import mpe                  # noqa: E402
from tests import cmp_cases                 # noqa: E402
from tests.mpe_dump import read_em_dump     # noqa: E402

n_frag = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
out = "/tmp/em_probe"
os.makedirs(out, exist_ok=True)
span, dump = out + "/spanning.txt", out + "/em.bin"
if not os.path.exists(dump):
    cmp_cases.config3_write(n_frag, span)
    env = {k: v for k, v in os.environ.items() if k != "DEFUSE_DSA_LIB"}
    r = subprocess.run([os.path.join(ROOT, "bin", "clustermatepairs"), "-a", span, "-c", out + "/unused", "-u", "300", "-s", "30", "-p", "0.95", "-m", "5"],
                       capture_output=True, text=True, env=dict(env, DEFUSE_CMP_DUMP_EM=dump))
    assert r.returncode == 0, r.stderr
d = read_em_dump(dump)
args = (d["mean"], d["sd"], d["min_prob"], d["min_size"], d["prob_off"], d["x"], d["y"], d["u"], d["to_xo"], d["to_yo"])
import hashlib
for k in range(reps):
    ncl, member, status, t = mpe.cluster_batch(*args)
    print("fragments %d problems %d mate pairs %d: kernel %.1f ms, EM iterations %d, clusters %d, member hash %s" % (
        n_frag, t.n_problems, t.n_mate_pairs, t.kernel_ms, t.em_iterations, int(ncl.sum()), hashlib.sha1(member.tobytes()).hexdigest()[:12]), flush=True)
