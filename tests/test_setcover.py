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


GOOD_LINES = ["%d\t%d\t%d\t1\tchr1\t+\t%d\t%d" % (c, e, 10 * c + k, 1000 + k, 1050 + k) for c in range(40) for e in (0, 1) for k in range(6)]


@pytest.mark.parametrize("threads", [None, "3", "16"])
@pytest.mark.parametrize("bad,message", [
    ("", "Error: Empty clusters line {n} of {path}\n"),
    ("7\t0", "Error: Format error for clusters line {n} of {path}\n"),
    ("7\tx\t12\t1\tchr1\t+\t5\t9", "Failed to interpret line:\n7\tx\t12\t1\tchr1\t+\t5\t9\n"),
    ("7\t0\t12 \t1\tchr1\t+\t5\t9", "Failed to interpret line:\n7\t0\t12 \t1\tchr1\t+\t5\t9\n"),
    ("7\t0\t99999999999\t1", "Failed to interpret line:\n7\t0\t99999999999\t1\n"),
    ("-3\t0\t12\t1\tchr1\t+\t5\t9", "Error: Invalid cluster ID for line {n} of {path}\n"),
])
def test_setcover_tool_reports_the_first_bad_line_as_a_serial_reader_would(built, tmp_path, threads, bad, message):
    """ReadClusters' error exits (tools/Parsers.cpp:23-84) from the threaded, one-pass reader: the first bad line of the file
    decides, whichever piece it is in and whatever comes later; these runs end before the tool needs a GPU."""
    import os
    lines = list(GOOD_LINES)
    n = 301
    lines.insert(n - 1, bad)
    lines.insert(n + 60, "9\ty\t1")                      # a later bad line of another kind: never reported
    p = tmp_path / "clusters.txt"
    p.write_text("\n".join(lines) + "\n")
    env = dict(os.environ, DEFUSE_THREADS=threads) if threads else dict(os.environ)
    r = subprocess.run([TOOL, "-c", str(p), "-m", "3", "-o", str(tmp_path / "out.sc")], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 1
    assert r.stdout == "Reading clusters\n"
    assert r.stderr == message.format(n=n, path=str(p))


def test_setcover_tool_negative_cluster_id_on_end_one_is_not_an_error_in_the_reader(built, tmp_path):
    """The reader skips end-1 lines before it looks at the cluster id (tools/Parsers.cpp:60-70): such a file gets as far as the
    set cover (and, without a GPU, to the tool's own message about that)."""
    lines = list(GOOD_LINES)
    lines.insert(100, "-3\t1\t12\t1\tchr1\t+\t5\t9")
    p = tmp_path / "clusters.txt"
    p.write_text("\n".join(lines) + "\n")
    r = subprocess.run([TOOL, "-c", str(p), "-m", "3", "-o", str(tmp_path / "out.sc")], capture_output=True, text=True, timeout=120)
    assert r.stdout.startswith("Reading clusters\nCalculating set cover solution\n")
    assert "Invalid cluster ID" not in r.stderr or r.stderr.startswith("Error: Invalid cluster ID for line 101")      # the writer (:86-170) does look at it


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [None, "5"])
def test_setcover_tool_writer_reports_a_negative_cluster_id_on_an_end_one_line(built, tmp_path, threads):
    """WriteClusters (tools/Parsers.cpp:86-170) checks the id of every line, whatever its end, and names the OUTPUT file."""
    import os
    lines = list(GOOD_LINES)
    lines.insert(100, "-3\t1\t12\t1\tchr1\t+\t5\t9")
    p, outp = tmp_path / "clusters.txt", tmp_path / "out.sc"
    p.write_text("\n".join(lines) + "\n")
    env = dict(os.environ, DEFUSE_THREADS=threads) if threads else dict(os.environ)
    r = subprocess.run([TOOL, "-c", str(p), "-m", "3", "-o", str(outp)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 1
    assert r.stdout == "Reading clusters\nCalculating set cover solution\nWriting out clusters\n"
    assert r.stderr == "Error: Invalid cluster ID for line 101 of %s\n" % outp


@pytest.mark.gpu
@pytest.mark.parametrize("threads", ["1", "6"])
def test_setcover_tool_cluster_lines_in_any_order(built, tmp_path, threads):
    """clusters[id] is built in file order whatever the order of the lines (tools/Parsers.cpp:72-73): a file whose lines are
    shuffled — every host piece then sees every cluster id — against the oracle reading the same file."""
    import os
    from oracle import setcover_oracle as o
    clusters = random_clusters(11, n_clusters=300, n_frag=900, big=1)
    p, outp = tmp_path / "clusters.txt", tmp_path / "clusters.sc"
    write_cluster_file(p, clusters)
    lines = p.read_text().splitlines()
    np.random.default_rng(5).shuffle(lines)
    p.write_text("\n".join(lines) + "\n")
    r = subprocess.run([TOOL, "-c", str(p), "-m", "3", "-o", str(outp)], capture_output=True, text=True,
                       env=dict(os.environ, DEFUSE_THREADS=threads), timeout=300)
    assert r.returncode == 0, r.stderr
    exp = o.setcover(str(p), 3)
    assert outp.read_text() == exp and len(exp.splitlines()) > 100
