"""The drop-in tool binaries: command line behaviour (CPU) and byte-exact outputs (GPU)."""
import os
import shutil
import subprocess

import pytest

from tests import pipeline_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMOKE = os.path.join(ROOT, "tests", "golden", "smoke")
TOOL = os.path.join(ROOT, "bin", "dosplitalign")


@pytest.fixture(scope="module")
def tools(built):
    from defuse_amd import build
    build.build_tools()
    return True


def smoke_args(tmp, out, regions="regions.txt"):
    for n in ("ref.fa", "exons.txt", "regions.txt", "improper.sam", "reads.1.fastq", "reads.2.fastq"):
        shutil.copy(os.path.join(SMOKE, n), tmp)          # the tool writes ref.fa.fai next to the FASTA
    d = str(tmp) + "/"
    return ["-f", d + "ref.fa", "-e", d + "exons.txt", "-u", "300", "-s", "30", "-n", "50", "-x", "50", "-r", d + regions,
            "-i", d + "improper.sam", "-1", d + "reads.1.fastq", "-2", d + "reads.2.fastq", "-a", out]


def test_cli_errors(tools, tmp_path):
    r = subprocess.run([TOOL, "-f", "x"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("PARSE ERROR:") and "One or more required arguments missing!" in r.stderr
    r = subprocess.run([TOOL, "--bogus", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "Couldn't find match for argument" in r.stderr
    args = smoke_args(tmp_path, str(tmp_path / "o"))
    args[args.index("-n") + 1] = "fifty"
    r = subprocess.run([TOOL] + args, capture_output=True, text=True)
    assert r.returncode == 1 and "Couldn't read argument value" in r.stderr
    r = subprocess.run([TOOL, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "Fusion sequence prediction by split reads" in r.stdout
    for flag in ("--fasta", "--exons", "--ufrag", "--sfrag", "--minread", "--maxread", "--regions", "--improper",
                 "--seq1", "--seq2", "--align"):
        assert flag in r.stdout                                 # tools/dosplitalign.cpp:43-56


def test_missing_inputs_fail(tools, tmp_path):
    args = smoke_args(tmp_path, str(tmp_path / "o"))
    args[args.index("-r") + 1] = str(tmp_path / "nope.txt")
    r = subprocess.run([TOOL] + args, capture_output=True, text=True)
    assert r.returncode != 0 and "Unable to open align region pairs file" in r.stderr
    args = smoke_args(tmp_path, str(tmp_path / "o"))
    args[args.index("-1") + 1] = str(tmp_path / "reads.1.txt")
    r = subprocess.run([TOOL] + args, capture_output=True, text=True)
    assert r.returncode != 0 and "unrecognized extension" in r.stderr


def test_no_candidates_needs_no_gpu(tools, tmp_path):
    """Regions that no mate alignment touches: empty output file, exit 0, .fai index built."""
    args = smoke_args(tmp_path, str(tmp_path / "out.align"))
    with open(tmp_path / "far.txt", "w") as f:
        f.write("7\t0\tchrA\t+\t2500\t2600\n7\t1\tchrA\t-\t200\t300\n")
    args[args.index("-r") + 1] = str(tmp_path / "far.txt")
    r = subprocess.run([TOOL] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "[fai_load] build FASTA index." in r.stderr
    assert os.path.getsize(tmp_path / "out.align") == 0
    fai = open(tmp_path / "ref.fa.fai").read().split("\n")
    assert fai[0] == "chrA\t3000\t6\t60\t61" and fai[1] == "chrB\t3000\t3062\t60\t61"


@pytest.mark.gpu
def test_dosplitalign_smoke_vector(tools, tmp_path):
    """The reference's own known-answer vector through the drop-in binary, byte for byte."""
    out = tmp_path / "split.align"
    r = subprocess.run([TOOL] + smoke_args(tmp_path, str(out)), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exp = "".join("\t".join(l.split()) + "\t\n" for l in open(os.path.join(SMOKE, "expected.split.align.txt")))
    assert out.read_text() == exp


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [5, 6])
def test_dosplitalign_matches_oracle(tools, tmp_path, seed):
    from oracle import dosplitalign_oracle as ora
    case = pipeline_case.build(str(tmp_path / "case"), seed=seed)
    exp = ora.dosplitalign(case["fasta"], case["exons"], case["ufrag"], case["sfrag"], case["minread"], case["maxread"],
                           case["regions"], case["improper"], case["seq1"], case["seq2"])
    os.remove(case["fasta"] + ".fai") if os.path.exists(case["fasta"] + ".fai") else None
    out = tmp_path / "split.align"
    r = subprocess.run([TOOL] + pipeline_case.tool_args(case, str(out)), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert len(exp.splitlines()) > 20
    assert out.read_text() == exp
    # the same run streamed to the GPU in batches of a few candidates: identical file
    out2 = tmp_path / "split.align.batched"
    r = subprocess.run([TOOL] + pipeline_case.tool_args(case, str(out2)), capture_output=True, text=True,
                       env=dict(os.environ, DEFUSE_DSA_BATCH_PAIRS="7"))
    assert r.returncode == 0, r.stderr
    assert out2.read_text() == exp


EVAL = os.path.join(ROOT, "bin", "evalsplitalign")


def eval_args(case, align, out):
    return ["-f", case["fasta"], "-e", case["exons"], "-u", str(case["ufrag"]), "-s", str(case["sfrag"]),
            "-n", str(case["minread"]), "-x", str(case["maxread"]), "-r", case["regions"], "-a", align,
            "-q", out + ".seq", "-b", out + ".break", "-p", out + ".predalign"]


def test_evalsplitalign_smoke_vector(tools, tmp_path):
    """evalsplitalign on the reference's known-answer vector."""
    args = smoke_args(tmp_path, "unused")
    case = dict(fasta=args[1], exons=args[3], ufrag=300, sfrag=30, minread=50, maxread=50, regions=args[13])
    align = tmp_path / "split.align"
    align.write_text("".join("\t".join(l.split()) + "\t\n" for l in open(os.path.join(SMOKE, "expected.split.align.txt"))))
    out = str(tmp_path / "pred")
    r = subprocess.run([EVAL] + eval_args(case, str(align), out), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out + ".break").read() == open(os.path.join(SMOKE, "expected.break.txt")).read()
    assert len(open(out + ".predalign").read().splitlines()) == 14
    f = open(out + ".seq").read().rstrip("\n").split("\t")
    assert f[0] == "0" and f[2:] == ["0", "14", "0.511905", "0.5"]


@pytest.mark.parametrize("seed", [5, 8])
def test_evalsplitalign_matches_oracle(tools, tmp_path, seed):
    from oracle import dosplitalign_oracle as ora
    case = pipeline_case.build(str(tmp_path / "case"), seed=seed)
    txt = ora.dosplitalign(case["fasta"], case["exons"], case["ufrag"], case["sfrag"], case["minread"], case["maxread"],
                           case["regions"], case["improper"], case["seq1"], case["seq2"])
    # the pipeline sorts by fusion id between the two tools (scripts/defuse_run.pl:528)
    lines = sorted(txt.splitlines(True), key=lambda l: int(l.split("\t")[0]))
    align = tmp_path / "sorted.align"
    align.write_text("".join(lines))
    exp = ora.evalsplitalign(case["fasta"], case["exons"], case["ufrag"], case["sfrag"], case["minread"], case["maxread"],
                             case["regions"], str(align))
    out = str(tmp_path / "pred")
    r = subprocess.run([EVAL] + eval_args(case, str(align), out), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out + ".seq").read() == exp[0]
    assert open(out + ".break").read() == exp[1]
    assert open(out + ".predalign").read() == exp[2]
    assert len(exp[1].splitlines()) >= 6


@pytest.mark.parametrize("threads", ["1", "3", "16"])
def test_evalsplitalign_pieces_give_the_same_files(tools, tmp_path, threads):
    """The alignment file is evaluated in one piece per host thread, cut at changes of the fusion id: any thread count
    gives the oracle's three files (more threads than groups included)."""
    from oracle import dosplitalign_oracle as ora
    case = pipeline_case.build(str(tmp_path / "case"), seed=11, n_fusions=9, reads_per_fusion=25)
    txt = ora.dosplitalign(case["fasta"], case["exons"], case["ufrag"], case["sfrag"], case["minread"], case["maxread"],
                           case["regions"], case["improper"], case["seq1"], case["seq2"])
    lines = sorted(txt.splitlines(True), key=lambda l: int(l.split("\t")[0]))
    align = tmp_path / "sorted.align"
    align.write_text("".join(lines))
    exp = ora.evalsplitalign(case["fasta"], case["exons"], case["ufrag"], case["sfrag"], case["minread"], case["maxread"],
                             case["regions"], str(align))
    out = str(tmp_path / "pred")
    r = subprocess.run([EVAL] + eval_args(case, str(align), out), capture_output=True, text=True, env=dict(os.environ, DEFUSE_THREADS=threads))
    assert r.returncode == 0, r.stderr
    assert (open(out + ".seq").read(), open(out + ".break").read(), open(out + ".predalign").read()) == exp
    assert len(exp[1].splitlines()) >= 6


def test_evalsplitalign_malformed_lines_end_the_run_like_the_reference(tools, tmp_path):
    """ReadSortedAlignments (tools/SplitAlignment.cpp:319-369) reads one line ahead: a malformed line that still opens a new
    group (seven fields, readable id) lets the running group out first; a malformed line inside a group, or one whose id
    cannot be read, ends the run before the running group is evaluated."""
    from oracle import dosplitalign_oracle as ora
    case = pipeline_case.build(str(tmp_path / "case"), seed=12, n_fusions=6, reads_per_fusion=20)
    txt = ora.dosplitalign(case["fasta"], case["exons"], case["ufrag"], case["sfrag"], case["minread"], case["maxread"],
                           case["regions"], case["improper"], case["seq1"], case["seq2"])
    lines = sorted(txt.splitlines(True), key=lambda l: int(l.split("\t")[0]))
    ids = [int(l.split("\t")[0]) for l in lines]
    groups = sorted(set(ids))
    assert len(groups) >= 3
    second = ids.index(groups[1])                              # first line of the second group
    third = ids.index(groups[2])

    def run(mutated):
        align = tmp_path / "bad.align"
        align.write_text("".join(mutated))
        out = str(tmp_path / "bad")
        r = subprocess.run([EVAL] + eval_args(case, str(align), out), capture_output=True, text=True, env=dict(os.environ, DEFUSE_THREADS="4"))
        return r, [int(l.split("\t")[0]) for l in open(out + ".seq")]

    bad = lines[third].split("\t")
    bad[5] = "x7"                                              # opens group three with a readable id: group two is written first
    r, done = run(lines[:third] + ["\t".join(bad)] + lines[third + 1:])
    assert r.returncode == 1 and "bad integer 'x7'" in r.stderr and done == groups[:2]
    bad = lines[second + 1].split("\t")
    bad[5] = "x7"                                              # inside group two: only group one is written
    r, done = run(lines[:second + 1] + ["\t".join(bad)] + lines[second + 2:])
    assert r.returncode == 1 and done == groups[:1]
    r, done = run(lines[:third] + ["short\tline\n"] + lines[third:])     # fewer than seven fields at a group change
    assert r.returncode == 1 and "Format error for candidate reads line" in r.stderr and done == groups[:1]


def test_outputs_may_be_pipes(tools, tmp_path):
    """The reference writes its outputs through ofstream, which accepts /dev/stdout, FIFOs and process substitution; the
    ordered writer of the tools falls back to plain ordered writes when the target is not a regular file."""
    from oracle import dosplitalign_oracle as ora
    case = pipeline_case.build(str(tmp_path / "case"), seed=13, n_fusions=5, reads_per_fusion=15)
    txt = ora.dosplitalign(case["fasta"], case["exons"], case["ufrag"], case["sfrag"], case["minread"], case["maxread"],
                           case["regions"], case["improper"], case["seq1"], case["seq2"])
    lines = sorted(txt.splitlines(True), key=lambda l: int(l.split("\t")[0]))
    align = tmp_path / "sorted.align"
    align.write_text("".join(lines))
    exp = ora.evalsplitalign(case["fasta"], case["exons"], case["ufrag"], case["sfrag"], case["minread"], case["maxread"],
                             case["regions"], str(align))
    out = str(tmp_path / "pred")
    args = eval_args(case, str(align), out)
    args[args.index("-b") + 1] = "/dev/stdout"                       # the break file goes down a pipe
    r = subprocess.run([EVAL] + args, capture_output=True, text=True, env=dict(os.environ, DEFUSE_THREADS="3"))
    assert r.returncode == 0, r.stderr
    assert r.stdout == exp[1] and open(out + ".seq").read() == exp[0] and open(out + ".predalign").read() == exp[2]


def test_device_pick_spreads_concurrent_processes(tools, tmp_path):
    """dsa_pick_device_among: processes that live side by side get different devices (lock files), a lone process gets
    pid mod n; a taken lock is skipped."""
    import ctypes
    import sys
    code = ("import ctypes, os, sys, time\n"
            "lib = ctypes.CDLL(sys.argv[1])\n"
            "print(lib.dsa_pick_device_among(4), flush=True)\n"
            "time.sleep(float(sys.argv[2]))\n")
    from defuse_amd import dsa
    env = dict(os.environ, DEFUSE_GPU_LOCK_DIR=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, "-c", code, dsa.LIB_PATH, "3"], stdout=subprocess.PIPE, text=True, env=env) for _ in range(4)]
    picked = [int(p.stdout.readline()) for p in procs]
    assert sorted(picked) == [0, 1, 2, 3]                              # four live processes, four devices
    fifth = subprocess.run([sys.executable, "-c", code, dsa.LIB_PATH, "0"], capture_output=True, text=True, env=env)
    assert int(fifth.stdout) in (0, 1, 2, 3)                           # all taken: pid mod n
    for p in procs:
        p.wait()
    again = subprocess.run([sys.executable, "-c", code, dsa.LIB_PATH, "0"], capture_output=True, text=True, env=env)
    assert int(again.stdout) in (0, 1, 2, 3)
    lib = ctypes.CDLL(dsa.LIB_PATH)
    assert lib.dsa_pick_device_among(1) == 0 and lib.dsa_pick_device_among(0) == 0
    # a process that asks twice keeps its device (its own lock must not push it on to the next one) and holds ONE lock
    twice = ("import ctypes, os, sys\n"
             "lib = ctypes.CDLL(sys.argv[1])\n"
             "a, b, c = lib.dsa_pick_device_among(4), lib.dsa_pick_device_among(4), lib.dsa_pick_device_among(4)\n"
             "locks = [os.readlink('/proc/self/fd/' + f) for f in os.listdir('/proc/self/fd') if os.path.exists('/proc/self/fd/' + f)]\n"
             "print(a, b, c, sum('defuse_gpu.' in l for l in locks))\n")
    r = subprocess.run([sys.executable, "-c", twice, dsa.LIB_PATH], capture_output=True, text=True, env=env)
    a, b, c, n_locks = (int(x) for x in r.stdout.split())
    assert a == b == c and n_locks == 1, r.stdout


def test_no_candidates_never_loads_the_library(tools, tmp_path):
    """A run without a single candidate does not even load the C-ABI library (let alone start a HIP runtime): with
    DEFUSE_DSA_LIB pointing nowhere it still succeeds; with a candidate the same setting is a clean error exit (status 1,
    message on stderr) however far the other threads are."""
    args = smoke_args(tmp_path, str(tmp_path / "out.align"))
    with open(tmp_path / "far.txt", "w") as f:
        f.write("7\t0\tchrA\t+\t2500\t2600\n7\t1\tchrA\t-\t200\t300\n")
    far = list(args)
    far[far.index("-r") + 1] = str(tmp_path / "far.txt")
    env = dict(os.environ, DEFUSE_DSA_LIB=str(tmp_path / "no_such_library.so"))
    r = subprocess.run([TOOL] + far, capture_output=True, text=True, env=env)
    assert r.returncode == 0 and os.path.getsize(tmp_path / "out.align") == 0, r.stderr
    for _ in range(5):                                   # timing-dependent before: the helper thread used to be mid-dlopen
        r = subprocess.run([TOOL] + args, capture_output=True, text=True, env=env)
        assert r.returncode == 1 and "cannot load the split alignment library" in r.stderr, (r.returncode, r.stderr)


def test_read_store_follows_the_number_of_reads_not_the_largest_id(tools, tmp_path):
    """deFuse's fragment indices are global to the run: a chunk whose ids sit around 10^8 must not make the read table 10^8
    slots long.  Also: later reads replace earlier ones, lookups of absent ids fail, sparse and far ids still resolve."""
    src = tmp_path / "rs.cpp"
    src.write_text('''
#include "%s/tools_src/defuse_host.hpp"
#include <cassert>
using namespace defuse;
int main() {
    ReadStore rs;
    const int base = 123456789;
    for (int k = 0; k < 50000; ++k) { std::string s = "ACGT" + std::to_string(k); rs.put(base + k, k & 1, s.data(), s.size()); }
    rs.put(base + 7, 1, "TTTT", 4);                       // replaces
    rs.put(5, 0, "GG", 2);                                // far below the chunk: hash map
    rs.put(base + 40000000, 1, "CC", 2);                  // far above: hash map
    rs.put(base + 40000000, 1, "CCC", 3);                 // replaces there too
    const char* s; size_t n;
    assert(rs.get(base + 7, 1, s, n) && std::string(s, n) == "TTTT");
    assert(rs.get(base + 8, 0, s, n) && std::string(s, n) == "ACGT8");
    assert(!rs.get(base + 8, 1, s, n));
    assert(rs.get(5, 0, s, n) && std::string(s, n) == "GG");
    assert(rs.get(base + 40000000, 1, s, n) && std::string(s, n) == "CCC");
    assert(!rs.get(base - 5000, 0, s, n) && !rs.get(base + 49999 + 3, 0, s, n));
    assert(rs.table_slots() <= 8 * 50010 + 4096);
    // ids spread thinly (every 1000th): the table stays small and everything resolves
    ReadStore thin;
    for (int k = 0; k < 2000; ++k) thin.put(1000 * k, 0, "A", 1);
    for (int k = 0; k < 2000; ++k) assert(thin.get(1000 * k, 0, s, n) && n == 1);
    assert(!thin.get(1500, 0, s, n) && thin.table_slots() <= 8 * 2001 + 4096);
    return 0;
}
''' % ROOT)
    exe = tmp_path / "rs"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-o", str(exe), str(src)])
    assert subprocess.run([str(exe)]).returncode == 0
