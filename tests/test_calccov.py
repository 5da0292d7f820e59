"""calccov: the drop-in binary against oracle/calccov_oracle.py; the oracle's restatement of glibc's
rand() against the platform's libc."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "bin", "calccov")


@pytest.fixture(scope="module")
def tools(built):
    from defuse_amd import build
    build.build_tools()
    return True


def make_case(d, seed=3, n_genes=40, n_frag=3000, rl=50):
    """Exon table with single- and multi-transcript genes, and a concordant SAM of paired alignments on `gene|transcript`
    references (two records per fragment), some on transcripts that are not sampled, some with flag-named read ends."""
    rng = np.random.default_rng(seed)
    os.makedirs(d, exist_ok=True)
    tx = []
    with open(os.path.join(d, "cdna.regions"), "w") as f:
        for g in range(n_genes):
            gene = "ENSG%05d" % int(rng.integers(0, 99999))
            for t in range(1 if g % 3 else 2):
                name = "ENST%05d" % (g * 10 + t)
                exons, pos = [], int(rng.integers(1000, 50000))
                for _ in range(int(rng.integers(1, 5))):
                    ln = int(rng.integers(200, 1500))
                    exons.append((pos, pos + ln - 1))
                    pos += ln + int(rng.integers(100, 3000))
                f.write("\t".join([gene, name, "chr%d" % (1 + g % 5), "+-"[g % 2]] + [str(x) for e in exons for x in e]) + "\t\n")
                tx.append((gene + "|" + name, sum(e - b + 1 for b, e in exons)))
        f.write("short\tline\n\n")
    lines = ["@HD\tVN:1.0", "@SQ\tSN:x\tLN:1"]
    for fr in range(n_frag):
        ref, ln = tx[int(rng.integers(0, len(tx)))]
        flen = int(rng.normal(200, 30))
        if ln < flen + 10:
            continue
        s = int(rng.integers(1, ln - flen))
        a = (s, 0), (s + flen - rl, 16)
        if fr % 2:
            a = a[::-1]
        for k, (pos, flag) in enumerate(a):
            qname = "%d/%d" % (fr, k + 1) if fr % 7 else "frag%d" % fr
            fl = flag if fr % 7 else flag | (0x40 if k == 0 else 0x80)
            lines.append("%s\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s" % (qname, fl, ref, pos, rl, "A" * rl, "I" * rl))
    with open(os.path.join(d, "cdna.pair.sam"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return os.path.join(d, "cdna.pair.sam"), os.path.join(d, "cdna.regions")


def run_tool(sam, regions, out, density="0.01", anchor="4", trim="50", extra=(), env=None):
    return subprocess.run([TOOL, "-c", sam, "-g", regions, "-l", out + ".len", "-p", out + ".pos", "-m", out + ".min", "-d", density,
                           "-a", anchor, "-t", trim] + list(extra), capture_output=True, text=True,
                          env=dict(os.environ, **env) if env else None)


def test_glibc_rand_restatement_equals_libc():
    from oracle import calccov_oracle as c
    libc = ctypes.CDLL("libc.so.6")
    for seed in (11, 1, 0, 12345, 2 ** 31 + 5):
        libc.srand(seed)
        g = c.GlibcRand(seed)
        assert [libc.rand() for _ in range(2000)] == [g.rand() for _ in range(2000)]


def test_oracle_shapes(tmp_path):
    from oracle import calccov_oracle as c
    sam, regions = make_case(str(tmp_path))
    ln, pos, mn = c.calccov(sam, regions, 0.01, 4, 50)
    assert len(ln.splitlines()) > 100 and len(pos.splitlines()) == len(mn.splitlines()) > 100
    ln2, pos2, _ = c.calccov(sam, regions, 0.01, 4, 50, multiexon=True)
    assert len(ln2.splitlines()) > len(ln.splitlines())                 # multi-transcript genes are sampled too
    vals = [float(l.split("\t")[1]) for l in pos.splitlines()]
    assert 0.0 <= min(vals) and max(vals) <= 1.0


def test_cli_and_errors(tools, tmp_path):
    r = subprocess.run([TOOL, "-c", "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "One or more required arguments missing!" in r.stderr
    r = subprocess.run([TOOL, "--help"], capture_output=True, text=True)
    assert "Calculate covariance stats from concordant alignments" in r.stdout and "--multiexon" in r.stdout
    sam, regions = make_case(str(tmp_path))
    bad = tmp_path / "bad.sam"
    lines = open(sam).read().splitlines()
    bad.write_text("\n".join(lines[:40] + [lines[40]] + lines[40:]) + "\n")      # three alignments for one fragment
    r = run_tool(str(bad), regions, str(tmp_path / "o"))
    assert r.returncode == 1 and "Error: expected 2 alignments per fragment" in r.stderr and "retrieved 3 alignments for" in r.stderr
    r = run_tool(sam, str(tmp_path / "nope"), str(tmp_path / "o"))
    assert r.returncode == 1 and "Error: Unable to gene transcripts file" in r.stderr
    # no fragment on a sampled transcript: three empty files, no GPU needed
    empty = tmp_path / "none.sam"
    empty.write_text("@HD\tVN:1.0\nq/1\t0\tother|tx\t5\t255\t50M\t*\t0\t0\t%s\t*\nq/2\t16\tother|tx\t200\t255\t50M\t*\t0\t0\t%s\t*\n" % ("A" * 50, "A" * 50))
    r = run_tool(str(empty), regions, str(tmp_path / "e"))
    assert r.returncode == 0, r.stderr
    assert all(os.path.getsize(str(tmp_path / "e") + x) == 0 for x in (".len", ".pos", ".min"))


@pytest.mark.gpu
@pytest.mark.parametrize("multiexon,threads", [(False, None), (True, "5"), (False, "64")])
def test_tool_matches_oracle(tools, tmp_path, multiexon, threads):
    from oracle import calccov_oracle as c
    sam, regions = make_case(str(tmp_path), seed=4, n_frag=6000)
    exp = c.calccov(sam, regions, 0.01, 4, 50, multiexon=multiexon)
    out = str(tmp_path / "cov")
    r = run_tool(sam, regions, out, extra=["--multiexon"] if multiexon else [], env={"DEFUSE_THREADS": threads} if threads else None)
    assert r.returncode == 0, r.stderr
    assert (open(out + ".len").read(), open(out + ".pos").read(), open(out + ".min").read()) == exp
    assert len(exp[0].splitlines()) > 200
