"""setcover: oracle behaviour (CPU) and GPU parity, including the drop-in binary."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "bin", "setcover")


def random_clusters(seed, n_clusters=300, n_frag=900, big=0):
    rng = np.random.default_rng(seed)
    clusters = []
    for c in range(n_clusters):
        kind = rng.integers(0, 10)
        if kind == 0:
            clusters.append([])                                          # id gap
            continue
        base = int(rng.integers(0, n_frag - 40))
        n = int(rng.integers(1, 14))
        frs = [base + int(x) for x in rng.integers(0, 30, size=n)]       # overlapping neighbourhoods, duplicates
        if kind == 1 and clusters:
            frs = list(clusters[int(rng.integers(0, len(clusters)))])    # exact copy: size ties
        clusters.append(frs)
    for b in range(big):                                                 # one large component: a chain of overlapping clusters
        start = n_frag + 10 + 400 * b
        for k in range(70):
            clusters.append([start + 3 * k + d for d in range(5)])
    return clusters


def write_cluster_file(path, clusters, seed=0):
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        for cid, frs in enumerate(clusters):
            for e in (0, 1):
                for fr in frs:
                    f.write("%d\t%d\t%d\t%d\tchr%d\t%s\t%d\t%d\n" % (cid, e, fr, int(rng.integers(0, 2)), 1 + cid % 3, "+-"[e],
                                                                   1000 + fr, 1050 + fr))


def test_oracle_tie_rule():
    from oracle import setcover_oracle as o
    # equal sizes: the highest index wins first; after a decrement the most recently changed wins
    assert o.set_cover([[1, 2, 3], [3, 4], [4, 5, 6], [1, 2, 3]]) == [[], [], [4, 5, 6], [1, 2, 3]]


def test_oracle_chain_exact():
    from oracle import setcover_oracle as o
    # sizes 2,2,2: cluster 2 wins (last arrival) and takes 3,4; cluster 1 drops to 1 (fragment 3 gone);
    # cluster 0 (size 2) then takes 1,2; cluster 1 ends empty
    assert o.set_cover([[1, 2], [2, 3], [3, 4]]) == [[1, 2], [], [3, 4]]


def test_oracle_file_roundtrip(tmp_path):
    from oracle import setcover_oracle as o
    clusters = random_clusters(1, n_clusters=40, n_frag=120)
    p = tmp_path / "clusters.txt"
    write_cluster_file(p, clusters)
    assert o.read_clusters(str(p)) == [c for c in clusters][:len(o.read_clusters(str(p)))]
    out = o.setcover(str(p), 3)
    kept = {}
    for line in out.splitlines():
        f = line.split("\t")
        kept.setdefault(int(f[0]), set()).add(int(f[2]))
    assert kept and all(len(v) >= 3 for v in kept.values())
    allf = [fr for v in kept.values() for fr in v]
    assert len(allf) == len(set(allf))                                   # every fragment in at most one cluster


def test_cli(built):
    from defuse_amd import build
    build.build_tools()
    r = subprocess.run([TOOL, "-c", "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "One or more required arguments missing!" in r.stderr
    r = subprocess.run([TOOL, "--help"], capture_output=True, text=True)
    assert "Set cover for maximum parsimony" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("seed,big", [(1, 0), (2, 1), (3, 2)])
def test_gpu_cover_matches_oracle(built, seed, big):
    from defuse_amd import sc
    from oracle import setcover_oracle as o
    clusters = random_clusters(seed, big=big)
    sol, t = sc.cover(clusters)
    exp = o.set_cover(clusters)
    assert [sorted(set(s)) for s in sol] == [sorted(set(s)) for s in exp]
    assert t.n_components > 10 and t.n_large >= (1 if big else 0)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [None, "1", "7", "64"])
def test_setcover_tool_matches_oracle(built, tmp_path, threads):
    import os
    from defuse_amd import build
    from oracle import setcover_oracle as o
    build.build_tools()
    clusters = random_clusters(7, n_clusters=500, n_frag=1500, big=1)
    p = tmp_path / "clusters.txt"
    write_cluster_file(p, clusters)
    outp = tmp_path / "clusters.sc"
    env = dict(os.environ, DEFUSE_THREADS=threads) if threads else None        # host pieces: one, several, more than lines
    r = subprocess.run([TOOL, "-c", str(p), "-m", "3", "-o", str(outp)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout == "Reading clusters\nCalculating set cover solution\nWriting out clusters\n"
    exp = o.setcover(str(p), 3)
    assert outp.read_text() == exp and len(exp.splitlines()) > 100
