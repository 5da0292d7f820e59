"""localalign: oracle behaviour on the CPU, GPU parity of the scores and of the drop-in binary."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "bin", "localalign")
PIPELINE = (10, -5, -5)          # scripts/defuse_run.pl:498: -m 10 -x -5 -g -5 -t 0.8


def random_pairs(seed, n, lr=(0, 300), ls=(0, 120), alphabet=b"ACGT", related=0.5):
    """Pairs of byte strings; `related` of them carry a mutated copy of a reference segment in the sequence."""
    rng = np.random.default_rng(seed)
    al = np.frombuffer(alphabet, dtype=np.uint8)
    out = []
    for _ in range(n):
        r = rng.choice(al, size=int(rng.integers(lr[0], lr[1] + 1)))
        n_s = int(rng.integers(ls[0], ls[1] + 1))
        if rng.random() < related and len(r) > 0 and n_s > 0:
            start = int(rng.integers(0, len(r)))
            s = r[start:start + n_s].copy()
            if len(s) < n_s:
                s = np.concatenate([s, rng.choice(al, size=n_s - len(s))])
            flips = rng.random(len(s)) < 0.05
            s[flips] = rng.choice(al, size=int(flips.sum()))
            if rng.random() < 0.3 and len(s) > 4:                        # an indel
                cut = int(rng.integers(1, len(s) - 1))
                s = np.concatenate([s[:cut], s[cut + 1:]])
        else:
            s = rng.choice(al, size=n_s)
        out.append((r.tobytes(), s.tobytes()))
    return out


# ---------------------------------------------------------------------------------------------- CPU
def test_oracle_known_values():
    from oracle import localalign_oracle as o
    assert o.simple_align(10, -5, -5, b"ACGTACGT", b"ACGTACGT") == 80
    assert o.simple_align(10, -5, -5, b"AAAA", b"TTTT") == 0                 # maximum starts at 0
    assert o.simple_align(10, -5, -5, b"", b"ACGT") == 0 and o.simple_align(10, -5, -5, b"ACGT", b"") == 0
    assert o.simple_align(10, -5, -5, b"TTACGTTT", b"ACGT") == 40            # free start in the reference (H(i,0) = 0)
    assert o.simple_align(10, -5, -5, b"ACGT", b"TTACGT") == 30              # but not in the sequence: H(0,j) = j*gap
    assert o.simple_align(10, -5, -5, b"acgt", b"ACGT") == 0                 # exact byte comparison
    assert o.simple_align(10, -5, -5, b"ACGTTACGT", b"ACGTACGT") == 75       # one gap


def test_oracle_matches_independent_recursion():
    from oracle import localalign_oracle as o
    for seed, prm in enumerate([(10, -5, -5), (2, -1, -2), (1, -3, -1), (5, 2, -1), (3, -2, 1), (0, 0, 0), (-1, -2, -3), (7, -20, -3)]):
        for r, s in random_pairs(seed, 40, lr=(0, 60), ls=(0, 40), alphabet=b"ACGTNacg"):
            assert o.simple_align(*prm, r, s) == o.simple_align_py(*prm, r, s)


def test_oracle_tool_protocol():
    from oracle import localalign_oracle as o
    out, err, rc = o.run(["a\tACGTACGT\tACGTACGT\textra\n", "b\tAAAA\tTTTT\n", "c\tACGTACGTAC\tACGTTCGTAC\n"], 10, -5, -5, 0.8)
    assert (out, err, rc) == ("a\t80\t1\nc\t85\t0.85\n", "", 0)
    out, err, rc = o.run(["a\tACGT\tACGT\n", "\n", "b\tACGT\tACGT\n"], 10, -5, -5)
    assert (out, err, rc) == ("a\t40\t1\n", "Error: Empty line 2\n", 1)
    out, err, rc = o.run(["a\tACGT\n"], 10, -5, -5)
    assert (out, err, rc) == ("", "Error: Format error for line 1\n", 1)
    assert o.run(["a\tACGT\t\n"], 10, -5, -5)[0] == "a\t0\t-nan\n"        # 0.0/0.0, never below the threshold
    assert o.format_double(1.0 / 3) == "0.333333" and o.format_double(0.85) == "0.85"


def test_library_exports_la():
    import ctypes
    from defuse_amd.dsa import LIB_PATH
    lib = ctypes.CDLL(LIB_PATH)
    for sym in ("la_align_batch", "la_align_batch_min", "la_last_error"):
        assert hasattr(lib, sym)


# ---------------------------------------------------------------------------------------------- GPU
def _check(pairs, prm):
    from defuse_amd import la
    from oracle import localalign_oracle as o
    got, t = la.align_batch(pairs, *prm)
    want = np.array([o.simple_align(*prm, r, s) for r, s in pairs], dtype=np.int32)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (prm, int(bad[0]), pairs[int(bad[0])], int(got[bad[0]]), int(want[bad[0]]))
    return t


@pytest.mark.gpu
def test_gpu_scores_pipeline_scoring(built):
    t = _check(random_pairs(11, 3000, lr=(0, 700), ls=(0, 260)), PIPELINE)
    assert t.n_packed16 == 3000 and t.n_int32 == 0


@pytest.mark.gpu
def test_gpu_scores_edge_cases(built):
    pairs = [(b"", b""), (b"ACGT", b""), (b"", b"ACGT"), (b"A", b"A"), (b"A", b"C"), (b"acgt", b"ACGT"),
             (b"ACGT" * 16, b"ACGT" * 16), (b"ACGT" * 16 + b"A", b"ACGT" * 16),            # 64 / 65 columns: tile edge
             (b"T" * 63 + b"ACGTACGT" + b"T" * 70, b"ACGTACGT"), (b"N" * 200, b"N" * 100),
             (b"ACGTTACGT", b"ACGTACGT"), (b"ACGTACGT", b"ACGTTACGT")]
    _check(pairs, PIPELINE)
    _check(pairs + random_pairs(5, 200, alphabet=b"ACGTNacgt"), PIPELINE)


@pytest.mark.gpu
def test_gpu_scores_other_scorings(built):
    pairs = random_pairs(21, 500, lr=(0, 200), ls=(0, 90), alphabet=b"ACGTN")
    for prm in [(2, -1, -2), (3, -2, -2), (1, -1, -1), (4, 0, -1), (10, -5, 0), (-1, -2, -3)]:       # packed kernel: gap <= mismatch <= 0
        assert _check(pairs, prm).n_int32 == 0
    for prm in [(5, 2, -1), (3, -2, 1), (1, -3, -1), (7, -20, -3), (10, -5, -2000)]:   # int32 kernel
        assert _check(pairs, prm).n_packed16 == 0


@pytest.mark.gpu
def test_gpu_long_sequences_use_int32(built):
    # 15 per row: sequences above ~1860 rows leave the 16-bit range and must be routed to the int32 kernel
    rng = np.random.default_rng(3)
    ref = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=2500).tobytes()
    pairs = [(ref, ref[100:100 + n]) for n in (1800, 1900, 2300)] + random_pairs(4, 50)
    t = _check(pairs, PIPELINE)
    assert t.n_int32 == 2 and t.n_packed16 == len(pairs) - 2


@pytest.mark.gpu
def test_gpu_minimum_scores(built):
    """With a minimum per pair (the tool's threshold) the device prunes: scores that reach the minimum are exact,
    the others only have to stay below it."""
    from defuse_amd import la
    from oracle import localalign_oracle as o
    pairs = random_pairs(51, 3000, lr=(50, 700), ls=(20, 260), related=0.5)
    want = np.array([o.simple_align(*PIPELINE, r, s) for r, s in pairs], dtype=np.int64)
    for frac in (0.8, 0.5, 1.0):
        need = np.array([int(np.ceil(frac * 10 * len(s))) for _, s in pairs], dtype=np.int32)
        got, _ = la.align_batch(pairs, *PIPELINE, min_score=need)
        hit = want >= need
        assert hit.sum() > 20 and (~hit).sum() > 100
        assert np.array_equal(got[hit], want[hit])
        assert np.all(got[~hit] < need[~hit]) and np.all(got >= 0)
    got, _ = la.align_batch(pairs, *PIPELINE, min_score=np.full(len(pairs), -5, dtype=np.int32))
    assert np.array_equal(got, want)                          # a minimum every score reaches: all exact


@pytest.mark.gpu
def test_gpu_small_scratch_chunks(built, monkeypatch):
    monkeypatch.setenv("DEFUSE_LA_SCRATCH_MB", "1")         # many launch groups
    _check(random_pairs(31, 1500, lr=(100, 400), ls=(50, 150)), PIPELINE)


@pytest.mark.gpu
def test_gpu_tool_matches_oracle(built, tmp_path):
    from oracle import localalign_oracle as o
    pairs = random_pairs(41, 400, lr=(1, 500), ls=(0, 200), related=0.8)
    lines = ["c%d\t%s\t%s\n" % (k, r.decode(), s.decode()) for k, (r, s) in enumerate(pairs)]
    for args, thr in ((["-m", "10", "-x", "-5", "-g", "-5", "-t", "0.8"], 0.8), (["--match", "10", "--mismatch", "-5", "--gap", "-5"], 0.0)):
        want = o.run(lines, 10, -5, -5, thr)
        p = subprocess.run([TOOL] + args, input="".join(lines), capture_output=True, text=True)
        assert (p.stdout, p.returncode) == (want[0], 0)
    bad = lines[:7] + ["\n"] + lines[7:]
    want = o.run(bad, 10, -5, -5, 0.0)
    p = subprocess.run([TOOL, "-m", "10", "-x", "-5", "-g", "-5"], input="".join(bad), capture_output=True, text=True)
    assert (p.stdout, p.stderr, p.returncode) == want
    p = subprocess.run([TOOL, "-m", "10", "-x", "-5"], input="", capture_output=True, text=True)
    assert p.returncode == 1 and "One or more required arguments missing!" in p.stderr and "-g <int>" in p.stdout    # TCLAP's texts
