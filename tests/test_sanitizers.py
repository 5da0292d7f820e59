"""The host code of the tools under AddressSanitizer + UBSan and under ThreadSanitizer (CPU builds, bin/asan and bin/tsan;
defuse_amd/build.py:build_sanitized): the threaded parsers, binners, joins, the glue steps and the piece-wise evaluator run
their host stages — the parts that need no GPU — and must finish clean.  The GPU pool offers no sanitizers (and none are
needed for these stages); the kernels are covered by the parity tests."""
import os
import shutil
import subprocess

import pytest

from tests import cmp_cases, pipeline_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "tests", "golden", "glue")
ENV = {"asan": {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"},
       "tsan": {"TSAN_OPTIONS": "halt_on_error=0:report_signal_unsafe=0"}}
BAD = ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "ERROR: ThreadSanitizer")


@pytest.fixture(scope="module", params=["asan", "tsan"])
def san(request, built):
    from defuse_amd import build
    kind = request.param
    tools = build.build_sanitized(kind)
    probe = subprocess.run([tools["defuse_glue"], "merge_clusters", os.path.join(GLUE, "merge_in0.txt")], capture_output=True, text=True,
                           env=dict(os.environ, **ENV[kind]))
    if probe.returncode != 0 and "FATAL: ThreadSanitizer" in probe.stderr:
        pytest.skip("ThreadSanitizer cannot start in this environment: " + probe.stderr.splitlines()[0])
    return kind, tools


def run(san, tool, args, env=None, stdin=None, ok=(0,)):
    kind, tools = san
    r = subprocess.run([tools[tool]] + args, capture_output=True, text=True, input=stdin, env=dict(os.environ, **ENV[kind], **(env or {})))
    assert not any(b in r.stderr for b in BAD), r.stderr[-3000:]
    assert r.returncode in ok, (r.returncode, r.stderr[-2000:])
    return r


def test_clustermatepairs_host_stages(san, tmp_path):
    lines = cmp_cases.many_loci(3)
    span = tmp_path / "spanning.txt"
    span.write_text("".join(lines))
    dumps = []
    for threads in ("1", "7"):
        d = tmp_path / ("dump" + threads)
        run(san, "clustermatepairs", ["-a", str(span), "-c", str(tmp_path / "c"), "-u", "300", "-s", "30", "-p", "0.95", "-m", "5"],
            env={"DEFUSE_THREADS": threads, "DEFUSE_CMP_DUMP_PROBLEMS": str(d), "DEFUSE_CMP_HOST_BINNING": "1"})
        dumps.append(d.read_bytes())
    assert dumps[0] == dumps[1] and len(dumps[0]) > 1000
    big = tmp_path / "big.txt"
    cmp_cases.config3_write(20000, str(big))
    run(san, "clustermatepairs", ["-a", str(big), "-c", str(tmp_path / "c"), "-u", "300", "-s", "30", "-p", "0.95", "-m", "5"],
        env={"DEFUSE_THREADS": "6", "DEFUSE_CMP_DUMP_EM": str(tmp_path / "em.bin"), "DEFUSE_CMP_HOST_BINNING": "1"})
    assert os.path.getsize(tmp_path / "em.bin") > 10000


def test_glue_steps(san, tmp_path):
    kind, tools = san                                   # (the relative file names need cwd)
    p = subprocess.run([tools["defuse_glue"], "merge_clusters", "merge_in0.txt", "merge_in1.txt", "merge_in2.txt"], cwd=GLUE, capture_output=True,
                       text=True, env=dict(os.environ, **ENV[kind]))
    assert p.returncode == 0 and not any(b in p.stderr for b in BAD), p.stderr[-2000:]
    assert p.stdout == open(os.path.join(GLUE, "merge_out.txt")).read()
    for step, args, inp in (("get_align_regions", [], "regions_in.txt"), ("remove_duplicates", ["3"], "dups_in.txt")):
        out = run(san, "defuse_glue", [step] + args, stdin=open(os.path.join(GLUE, inp)).read()).stdout
        assert len(out) > 0
    body = "".join(l for l in open(os.path.join(GLUE, "improper.sam")) if not l.startswith("@"))
    assert run(san, "defuse_glue", ["filter_unmatched"], stdin=body).stdout == open(os.path.join(GLUE, "matched.sam")).read()
    shutil.copy(os.path.join(GLUE, "trans_chr.txt"), tmp_path / "trans_chr.txt")
    os.makedirs(tmp_path / "div")
    p = subprocess.run([tools["defuse_glue"], "divide_sam_chr_pairs", "-t", "trans_chr.txt", "-p", "div/"], cwd=tmp_path, capture_output=True, text=True,
                       input=open(os.path.join(GLUE, "matched.sam")).read(), env=dict(os.environ, **ENV[kind]))
    assert p.returncode == 0 and not any(b in p.stderr for b in BAD), p.stderr[-2000:]


def test_evalsplitalign_and_dosplitalign_host_paths(san, tmp_path):
    from oracle import dosplitalign_oracle as ora
    case = pipeline_case.build(str(tmp_path / "case"), seed=11, n_fusions=70, reads_per_fusion=12)      # > 64 fusions: tasks set up on threads
    txt = ora.dosplitalign(case["fasta"], case["exons"], case["ufrag"], case["sfrag"], case["minread"], case["maxread"],
                           case["regions"], case["improper"], case["seq1"], case["seq2"])
    lines = sorted(txt.splitlines(True), key=lambda l: int(l.split("\t")[0]))
    align = tmp_path / "sorted.align"
    align.write_text("".join(lines))
    exp = ora.evalsplitalign(case["fasta"], case["exons"], case["ufrag"], case["sfrag"], case["minread"], case["maxread"], case["regions"], str(align))
    out = str(tmp_path / "pred")
    common = ["-f", case["fasta"], "-e", case["exons"], "-u", str(case["ufrag"]), "-s", str(case["sfrag"]), "-n", str(case["minread"]),
              "-x", str(case["maxread"]), "-r", case["regions"]]
    run(san, "evalsplitalign", common + ["-a", str(align), "-q", out + ".seq", "-b", out + ".break", "-p", out + ".predalign"], env={"DEFUSE_THREADS": "5"})
    assert (open(out + ".seq").read(), open(out + ".break").read(), open(out + ".predalign").read()) == exp
    # dosplitalign up to the first batch: regions no alignment reaches, so parsing, the parallel task set-up, the FASTQ threads and
    # the SAM pieces run, and the GPU is never asked for
    far = tmp_path / "far.sam"
    far.write_text("@HD\tVN:1.0\n" + "".join("%d/1\t0\tnowhere\t%d\t255\t50M\t*\t0\t0\t%s\t%s\n" % (k, 10 + k, "A" * 50, "I" * 50) for k in range(3000)))
    o = tmp_path / "split.align"
    run(san, "dosplitalign", common + ["-i", str(far), "-1", case["seq1"], "-2", case["seq2"], "-a", str(o)], env={"DEFUSE_THREADS": "6"})
    assert o.read_text() == ""


def test_calccov_and_setcover_host_paths(san, tmp_path):
    from tests import test_calccov
    sam, regions = test_calccov.make_case(str(tmp_path / "cov"))
    lines = open(sam).read().splitlines()
    bad = tmp_path / "bad.sam"
    bad.write_text("\n".join(lines[:40] + [lines[40]] + lines[40:]) + "\n")
    r = run(san, "calccov", ["-c", str(bad), "-g", regions, "-l", str(tmp_path / "l"), "-p", str(tmp_path / "p"), "-m", str(tmp_path / "m"),
                             "-d", "0.01", "-a", "4", "-t", "50"], env={"DEFUSE_THREADS": "4"}, ok=(1,))
    assert "expected 2 alignments per fragment" in r.stderr
    r = run(san, "setcover", ["-c", str(tmp_path / "missing"), "-m", "3", "-o", str(tmp_path / "o")], ok=(1,))
