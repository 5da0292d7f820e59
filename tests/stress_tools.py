"""Randomised parity stress of the other device paths against their CPU oracles (run by hand on a GPU box:
`python tests/stress_tools.py [rounds] [seed0]`): set cover, mate-pair EM clustering (through the drop-in
binary), the localalign scorer and the average-linkage clusterer."""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, ".")
from defuse_amd import hc, la, sc
from oracle import clustermatepairs_oracle as cmp_o
from oracle import hierarchical_oracle as hc_o
from oracle import localalign_oracle as la_o
from oracle import setcover_oracle as sc_o
from tests import cmp_cases
from tests.test_hierarchical import _random_tables
from tests.test_localalign import random_pairs
from tests.test_setcover import random_clusters

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    tmp = tempfile.mkdtemp(prefix="stress_tools_")
    for r in range(rounds):
        seed = seed0 + r
        rng = np.random.default_rng(seed)
        # set cover
        clusters = random_clusters(seed, n_clusters=int(rng.integers(50, 600)), n_frag=int(rng.integers(100, 2000)),
                                   big=int(rng.integers(0, 3)))
        sol, _ = sc.cover(clusters)
        exp = sc_o.set_cover(clusters)
        assert [sorted(set(s)) for s in sol] == [sorted(set(s)) for s in exp], ("setcover", seed)
        # average-linkage clusterer (continuous distances and tables full of ties in turn)
        tabs, thr = _random_tables(seed, count=12, nmax=int(rng.integers(5, 90)), quantised=bool(r % 2))
        got_hc, _ = hc.cluster_batch(tabs, thr)
        assert got_hc == [hc_o.do_clustering(t, th) for t, th in zip(tabs, thr)], ("hierarchical", seed)
        # localalign scores
        prm = [(10, -5, -5), (2, -1, -2), (5, 2, -1), (1, -3, -1), (3, -2, -2)][r % 5]
        pairs = random_pairs(seed, 400, lr=(0, int(rng.integers(50, 900))), ls=(0, int(rng.integers(20, 300))),
                             alphabet=b"ACGTN" if r % 3 else b"ACGTNacgt")
        got, _ = la.align_batch(pairs, *prm)
        want = [la_o.simple_align(*prm, a, b) for a, b in pairs]
        assert list(got) == want, ("localalign", seed, prm)
        # clustermatepairs through the binary
        lines = cmp_cases.many_loci(seed)
        p = os.path.join(tmp, "spanning.txt")
        out = os.path.join(tmp, "clusters.txt")
        open(p, "w").write("".join(lines))
        m = int(rng.integers(3, 7))
        run = subprocess.run([os.path.join(ROOT, "bin", "clustermatepairs"), "-a", p, "-c", out, "-u", "300", "-s", "30", "-p", "0.95",
                              "-m", str(m)], capture_output=True, text=True,
                             env=dict(os.environ, DEFUSE_MPE_WAVE_MIN=["0", "40", "1000000000"][r % 3]))   # all waves (default) / split by size / all lanes
        exp_txt, _ = cmp_o.clustermatepairs(lines, 300, 30, 0.95, m)
        assert run.returncode == 0 and open(out).read() == exp_txt, ("clustermatepairs", seed, m, run.stderr[-300:])
        if r % 5 == 4:
            print("round %d ok" % (r + 1), flush=True)
    print("all %d rounds agree" % rounds)


if __name__ == "__main__":
    main()
