"""CPU tests: the oracle against the reference's known-answer vector, and host-side checks."""
import ctypes
import os

import numpy as np
import pytest

from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMOKE = os.path.join(ROOT, "tests", "golden", "smoke")


@pytest.fixture(scope="module")
def ora(built):
    from oracle import dosplitalign_oracle as o
    return o


def test_smoke_inputs_reproducible(tmp_path):
    """The committed inputs are exactly what the documented recipe generates."""
    from tests.golden import make_smoke
    make_smoke.main(str(tmp_path))
    for name in ("ref.fa", "exons.txt", "regions.txt", "reads.1.fastq", "reads.2.fastq", "improper.sam"):
        assert open(os.path.join(tmp_path, name)).read() == open(os.path.join(SMOKE, name)).read(), name


def test_oracle_reproduces_reference_dosplitalign(ora):
    """22 lines recorded from the reference's own dosplitalign."""
    d = SMOKE + "/"
    txt = ora.dosplitalign(d + "ref.fa", d + "exons.txt", 300, 30, 50, 50, d + "regions.txt", d + "improper.sam",
                           d + "reads.1.fastq", d + "reads.2.fastq")
    lines = txt.split("\n")
    assert lines[-1] == ""
    assert all(l.endswith("\t") for l in lines[:-1])          # trailing tab of WriteAlignment
    got = [l.rstrip("\t").split("\t") for l in lines[:-1]]
    exp = [l.split() for l in open(d + "expected.split.align.txt")]
    assert got == exp


def test_oracle_reproduces_reference_evalsplitalign(ora, tmp_path):
    d = SMOKE + "/"
    txt = ora.dosplitalign(d + "ref.fa", d + "exons.txt", 300, 30, 50, 50, d + "regions.txt", d + "improper.sam",
                           d + "reads.1.fastq", d + "reads.2.fastq")
    p = tmp_path / "split.align"
    p.write_text(txt)
    seq, brk, pred = ora.evalsplitalign(d + "ref.fa", d + "exons.txt", 300, 30, 50, 50, d + "regions.txt", str(p))
    assert brk == open(d + "expected.break.txt").read()
    assert len(pred.splitlines()) == 14
    f = seq.rstrip("\n").split("\t")
    assert f[0] == "0" and f[2:] == ["0", "14", "0.511905", "0.5"]
    left, right = f[1].split("|")
    ref = ora.FastaIndex(d + "ref.fa")
    assert ref.seqs["chrA"][500:650].endswith(left.encode()[-150:]) and left.encode().endswith(ref.seqs["chrA"][500:650])
    assert right.encode().startswith(ref.seqs["chrB"][999:1050])


def test_min_score_expression(ora):
    # 76->136, 100->180, 150->270, 50->90
    for lq, ms in ((76, 136), (100, 180), (150, 270), (50, 90), (0, 0), (1, 1), (7, 12)):
        assert ora.lib().ora_min_score(lq) == ms


def test_fill_matrix_boundaries(ora):
    ref, read = b"ACGTACGT", b"CGTA"
    m = (ctypes.c_int * ((len(ref) + 1) * (len(read) + 1)))()
    ora.lib().ora_fill_matrix(ref, len(ref), read, len(read), m)
    L = len(ref) + 1
    assert all(m[i] == 0 for i in range(L))                       # row j=0
    assert [m[j * L] for j in range(len(read) + 1)] == [0, -2, -4, -6, -8]
    assert m[4 * L + 5] == 8                                      # CGTA matches ref[1:5]


def test_zero_side_rule(ora):
    """A split whose one side has no entry >= 8 still competes for the maximum but emits nothing."""
    rng = np.random.default_rng(5)
    ref0, ref1 = cases.rnd(rng, 120), cases.rnd(rng, 120)
    read = ref1[40:90]                     # aligns unsplit on window 1 with a = 0
    out = ora.task_align(read, ref0, ref1)
    assert out == [] or all(r[2] >= 4 for r in out)


def test_library_exports(built):
    """The C-ABI library loads without a GPU and exports every symbol declared in include/defuse_dsa.h."""
    import re
    from defuse_amd import dsa
    lib = ctypes.CDLL(dsa.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "defuse_dsa.h")).read()
    declared = set(re.findall(r"\b(dsa_[a-z_]+)\s*\(", header))
    assert declared == set(dsa.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in ctypes.cast(lib.dsa_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()
    # and every function the other headers of include/ declare
    for h, prefix in (("defuse_sc.h", "sc"), ("defuse_mpe.h", "mpe"), ("defuse_la.h", "la"), ("defuse_hc.h", "hc"), ("defuse_cov.h", "cov"), ("defuse_cmp.h", "cmp")):
        text = open(os.path.join(ROOT, "include", h)).read()
        names = set(re.findall(r"\b(%s_[a-z_]+)\s*\(" % prefix, text))
        assert names, h
        for name in names:
            assert getattr(lib, name) is not None, name


def test_no_cpu_fallback_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from defuse_amd import dsa
    with pytest.raises(dsa.DsaError):
        dsa.Context(0)


def test_struct_layouts_match_header(built):
    from defuse_amd import dsa
    assert dsa.FUSION_DTYPE.itemsize == 20 and dsa.PAIR_DTYPE.itemsize == 20 and dsa.RECORD_DTYPE.itemsize == 40
    assert ctypes.sizeof(dsa.Timing) == 56 and ctypes.sizeof(dsa.Limits) == 12      # dsa_timing grew by plan_ms + pad_


def test_oracle_batch_matches_python_loop(ora):
    ref, fus, reads, pairs = cases.mixed_batch(3, n_fusions=3, reads_per_fusion=8, lq=30, lr=(60, 90))
    recs = ora.align_batch(ref, fus, reads, pairs)
    exp = []
    for p in pairs:
        f = fus[p["fusion_idx"]]
        r0 = ref[f["ref0_off"]:f["ref0_off"] + f["ref0_len"]].tobytes()
        r1 = ref[f["ref1_off"]:f["ref1_off"] + f["ref1_len"]].tobytes()
        rd = reads[p["read_off"]:p["read_off"] + p["read_len"]].tobytes()
        for (a, b, c, d, s) in ora.task_align(rd, r0, r1):
            exp.append((f["fusion_id"], p["frag"], p["read_end"], p["revcomp"], a, b, c, d, s))
    assert [tuple(int(x) for x in r)[:9] for r in recs] == [tuple(int(x) for x in e) for e in exp]
    assert (np.diff(recs["pair_idx"]) >= 0).all()
    assert len(exp) > 0


def test_synth_batch_shapes():
    from defuse_amd import synth
    ref, fus, reads, pairs = synth.make_batch(5, 7, lq=76, lr=389, seed=2)
    assert ref.size == 5 * 2 * 389 and reads.size == 35 * 76 and len(pairs) == 35
    assert synth.window_length(300, 30, 76, 76, 151) == 389
    assert synth.window_length(300, 30, 100, 100, 150) == 390
    assert synth.cells_per_align(76, 389) == 60060
    ref2, *_ = synth.make_batch(5, 7, lq=76, lr=389, seed=2)
    assert (ref == ref2).all()
