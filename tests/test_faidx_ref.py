"""FASTA index and region fetch: the oracle's FastaIndex (oracle/dosplitalign_oracle.py) and the index file the tools write
against the REFERENCE'S OWN code — faidx.c of the vendored samtools 0.1.8 (external/samtools-0.1.8), which tools/FastaIndex.cpp
wraps, compiled as it lies into oracle/_ref/libfaidx_ref.so by oracle/Makefile.  The window and remainder sequences of every
fusion, and the clipped start / length values the break positions are computed from (SURVEY a-5), come through this call.
The file travels prebuilt to the GPU box; without it the tests skip."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libfaidx_ref.so")


@pytest.fixture(scope="module")
def faidx(built):
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/libfaidx_ref.so not built (reference sources absent)")
    lib = C.CDLL(REF)
    lib.fai_load.restype = C.c_void_p
    lib.fai_load.argtypes = [C.c_char_p]
    lib.fai_fetch.restype = C.c_void_p
    lib.fai_fetch.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
    lib.fai_build.argtypes = [C.c_char_p]
    lib.fai_destroy.argtypes = [C.c_void_p]
    return lib


def write_fasta(path, seed=5):
    rng = np.random.default_rng(seed)
    seqs = {}
    with open(path, "wb") as f:
        for k, (name, n, width) in enumerate((("chr1", 1000, 60), ("ENSG01|ENST01", 333, 70), ("short", 7, 60), ("chrM", 601, 50),
                                               ("lower", 240, 60), ("oneline", 95, 200))):
            s = bytes(rng.choice(np.frombuffer(b"ACGTN" if k != 4 else b"acgtn", dtype=np.uint8), size=n))
            seqs[name] = s
            f.write(b">" + name.encode() + (b" description text\n" if k % 2 else b"\n"))
            for o in range(0, n, width):
                f.write(s[o:o + width] + b"\n")
    return seqs


def ref_get(lib, fai, reference, start, length):
    """FastaIndex::Get (tools/FastaIndex.cpp:23-61) around the reference's fai_fetch, plus strand."""
    if length < 0:
        return b"", start, length
    if start < 1:
        length -= 1 - start
        start = 1
    end = start + length - 1
    n = C.c_int(length)
    p = lib.fai_fetch(fai, ("%s:%d-%d" % (reference, start, end)).encode(), C.byref(n))
    assert p
    s = C.string_at(p)
    C.CDLL("libc.so.6").free(C.c_void_p(p))
    return s, start, n.value


def test_region_fetch_equals_the_reference(faidx, tmp_path):
    from oracle import dosplitalign_oracle as ora
    fa = str(tmp_path / "ref.fa")
    seqs = write_fasta(fa)
    fai = faidx.fai_load(fa.encode())                    # builds ref.fa.fai
    assert fai
    o = ora.FastaIndex(fa)
    rng = np.random.default_rng(9)
    n_checked = 0
    for name, s in seqs.items():
        L = len(s)
        cases = [(1, L), (1, 1), (L, 1), (L, 5), (L + 1, 3), (L + 50, 10), (0, 10), (-5, 3), (-5, 20), (-100, 50), (5, 0), (5, -1),
                 (1, L + 100), (2, L - 2)]
        cases += [(int(rng.integers(-30, L + 30)), int(rng.integers(-2, L + 40))) for _ in range(300)]
        for start, length in cases:
            exp = ref_get(faidx, fai, name, start, length)
            got = o.get(name, ora.PLUS, start, length)
            assert (got[0], got[1], got[2]) == exp, (name, start, length)
            n_checked += 1
            rc = o.get(name, ora.MINUS, start, length)
            assert rc[0] == ora.reverse_complement(exp[0]) and rc[1:] == exp[1:]
    assert n_checked > 1500
    faidx.fai_destroy(fai)


def test_index_file_of_the_tools_equals_the_reference(faidx, tmp_path):
    """bin/dosplitalign builds <fasta>.fai when it is missing, as fai_load does; the bytes must be fai_build's."""
    from defuse_amd import build
    build.build_tools()
    a, b = tmp_path / "a", tmp_path / "b"
    os.makedirs(a)
    os.makedirs(b)
    write_fasta(str(a / "ref.fa"))
    shutil.copy(a / "ref.fa", b / "ref.fa")
    assert faidx.fai_build(str(a / "ref.fa").encode()) == 0
    (b / "exons.txt").write_text("g\tt\tchr1\t+\t1\t900\t\n")
    (b / "regions.txt").write_text("0\t0\tchr1\t+\t300\t400\n0\t1\tchr1\t-\t700\t800\n")
    (b / "improper.sam").write_text("@HD\tVN:1.0\n")
    for e in (1, 2):
        (b / ("reads.%d.fastq" % e)).write_text("@0/%d\nACGT\n+\nIIII\n" % e)
    r = subprocess.run([os.path.join(ROOT, "bin", "dosplitalign"), "-f", str(b / "ref.fa"), "-e", str(b / "exons.txt"), "-u", "200", "-s", "30",
                        "-n", "50", "-x", "50", "-r", str(b / "regions.txt"), "-i", str(b / "improper.sam"), "-1", str(b / "reads.1.fastq"),
                        "-2", str(b / "reads.2.fastq"), "-a", str(b / "out.align")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "[fai_load] build FASTA index." in r.stderr
    assert (b / "ref.fa.fai").read_bytes() == (a / "ref.fa.fai").read_bytes()
