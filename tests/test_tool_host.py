"""The host logic of bin/dosplitalign where there is no GPU: parsing, de-duplication, the candidate table, batch cutting, the
counting sort by fusion, the worker process and its shared slots, formatting and writing — with DEFUSE_DSA_LIB pointing at a
TEST DOUBLE of the streaming ABI (tests/shim/dsa_abi_double.c: the CPU oracle behind dsa_stream_*).  The output file must be
the Python oracle's, byte for byte, whatever the batch size, the byte limits, the thread count and the worker's place (process
or thread) are.  The GPU tests (tests/test_tools.py, -m gpu) run the same binary on the real library."""
import os
import subprocess

import pytest

from tests import pipeline_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "bin", "dosplitalign")


@pytest.fixture(scope="module")
def double(built):
    from defuse_amd import build
    build.build_tools()
    out_dir = os.path.join(ROOT, "tests", "shim", "_build")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libdsa_abi_double.so")
    src = os.path.join(ROOT, "tests", "shim", "dsa_abi_double.c")
    deps = [src, os.path.join(ROOT, "oracle", "dsa_oracle.c"), os.path.join(ROOT, "include", "defuse_dsa.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.check_call(["gcc", "-O2", "-g", "-fPIC", "-shared", "-std=c11", "-Wall", "-o", lib, src])
    return lib


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    from oracle import dosplitalign_oracle as ora
    d = tmp_path_factory.mktemp("toolhost")
    c = pipeline_case.build(str(d / "case"), seed=21, n_fusions=40, reads_per_fusion=25, lq=50)
    exp = ora.dosplitalign(c["fasta"], c["exons"], c["ufrag"], c["sfrag"], c["minread"], c["maxread"], c["regions"], c["improper"], c["seq1"], c["seq2"])
    assert exp.count("\n") > 500
    return c, exp, d


def run_tool(case, out, lib, env=None, tool=TOOL, extra_env=None):
    c = case
    e = dict(os.environ, DEFUSE_DSA_LIB=lib, **(env or {}), **(extra_env or {}))
    return subprocess.run([tool] + pipeline_case.tool_args(c, out), capture_output=True, text=True, env=e, timeout=600)


@pytest.mark.parametrize("env", [
    {},                                                             # defaults: one batch, worker process
    {"DEFUSE_THREADS": "5"},                                        # teams of five on a small input
    {"DEFUSE_THREADS": "3", "DEFUSE_DSA_BATCH_PAIRS": "7"},         # dozens of batches through the three slots
    {"DEFUSE_THREADS": "4", "DEFUSE_DSA_BATCH_PAIRS": "1"},         # one record's candidates per batch
    {"DEFUSE_THREADS": "4", "DEFUSE_DSA_INPROCESS": "1", "DEFUSE_DSA_BATCH_PAIRS": "50"},      # the worker as a thread
    {"DEFUSE_THREADS": "6", "DEFUSE_DSA_BATCH_READ_BYTES": "333"},  # batches cut by read bytes (a few reads each)
    {"DEFUSE_THREADS": "2", "DEFUSE_DSA_BATCH_REF_BYTES": "3000"},  # windows over the limit: batches halved until they fit
    {"DEFUSE_THREADS": "1", "DEFUSE_DSA_BATCH_PAIRS": "64", "DEFUSE_DSA_PINNED": "1"},
], ids=lambda e: ",".join("%s=%s" % (k.replace("DEFUSE_", ""), v) for k, v in e.items()) or "defaults")
def test_output_equals_the_oracle_whatever_the_batching(double, case, env):
    c, exp, d = case
    out = str(d / ("out_%d.align" % abs(hash(tuple(sorted(env.items()))))))
    r = run_tool(c, out, double, env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(out).read() == exp


def test_timing_lines_and_worker_report(double, case):
    c, exp, d = case
    r = run_tool(c, str(d / "t.align"), double, {"DEFUSE_TIMING": "1", "DEFUSE_DSA_BATCH_PAIRS": "100"})
    assert r.returncode == 0 and "GPU worker (process)" in r.stderr and "of which GPU calls" in r.stderr and "main()" in r.stderr, r.stderr
    r = run_tool(c, str(d / "t.align"), double, {"DEFUSE_TIMING": "1", "DEFUSE_DSA_INPROCESS": "1"})
    assert r.returncode == 0 and "GPU worker (thread)" in r.stderr, r.stderr


def test_a_worker_without_a_device_is_a_clean_error(double, case):
    """dsa_stream_create fails in the worker (no GPU): the main process says so and exits 1 — as a process and as a thread."""
    c, exp, d = case
    for extra in ({}, {"DEFUSE_DSA_INPROCESS": "1"}):
        r = run_tool(c, str(d / "e.align"), double, {"DSA_DOUBLE_NO_DEVICE": "1"}, extra_env=extra)
        assert r.returncode == 1 and "no usable MI355X/HIP device" in r.stderr, (r.returncode, r.stderr)


def test_a_worker_that_dies_is_noticed(double, case, tmp_path):
    """The worker process killed from outside while the main process waits for its records: exit 1 with a message, no hang."""
    c, exp, d = case
    lib = tmp_path / "libkill.so"
    src = tmp_path / "kill.c"
    src.write_text('#include <signal.h>\n#include <unistd.h>\n#include "%s/tests/shim/dsa_abi_double.c"\n'
                   '__attribute__((constructor)) static void die_soon(void) { }\n'
                   'int dsa_pick_device_dummy;\n' % ROOT)
    # a double whose collect never comes back: the worker kills itself in it
    src.write_text(open(os.path.join(ROOT, "tests", "shim", "dsa_abi_double.c")).read().replace('#include "../../oracle/dsa_oracle.c"', '#include "%s/oracle/dsa_oracle.c"\n#include <signal.h>\n#include <unistd.h>' % ROOT)
                   .replace("int dsa_stream_collect(dsa_stream* s, int64_t* out_n)\n{", "int dsa_stream_collect(dsa_stream* s, int64_t* out_n)\n{\n    if (s->n_collected >= 1) kill(getpid(), SIGKILL);"))
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-std=gnu11", "-I" + os.path.join(ROOT, "tests", "shim"), "-o", str(lib), str(src)])
    r = run_tool(c, str(d / "k.align"), str(lib), {"DEFUSE_DSA_BATCH_PAIRS": "200"})
    assert r.returncode == 1 and "GPU worker process ended unexpectedly" in r.stderr, (r.returncode, r.stderr[-500:])


def test_fastq_table_equals_the_serial_reader(built, tmp_path):
    """ReadTable (mapped file, a team of threads) against ReadStore + AddReads (the serial reader it replaces) on files with
    duplicates (the later record wins), ids far from the others, a truncated last record, a last line without a newline, and
    every kind of record that ends the reading: same reads, same messages, same verdict."""
    src = tmp_path / "rt.cpp"
    src.write_text(r'''
#include "%s/tools_src/defuse_host.hpp"
using namespace defuse;
int main(int argc, char** argv) {
    int bad = 0;
    for (int a = 1; a < argc; ++a) {
        for (unsigned nt : {1u, 3u, 8u}) {
            Team team(nt);
            ReadTable table; ReadStore store;
            std::ostringstream e1, e2; std::string f1, f2;
            const bool ok1 = table.load(argv[a], team, e1, &f1);
            const bool ok2 = AddReads(argv[a], store, e2, &f2);
            if (ok1 != ok2 || e1.str() != e2.str() || f1 != f2) { std::cerr << argv[a] << ": verdicts differ: [" << e1.str() << f1 << "] [" << e2.str() << f2 << "]\n"; ++bad; continue; }
            for (int frag = -5; frag < 70000; ++frag)
                for (int end = 0; end < 2; ++end) {
                    const char *s1 = nullptr, *s2 = nullptr; size_t n1 = 0, n2 = 0;
                    const bool g1 = table.get(frag, end, s1, n1), g2 = store.get(frag, end, s2, n2);
                    if (g1 != g2 || (g1 && (n1 != n2 || memcmp(s1, s2, n1)))) { std::cerr << argv[a] << ": read " << frag << "/" << end << " differs at " << nt << " threads\n"; ++bad; frag = 70000; break; }
                }
            const char* s; size_t n;
            if (table.get(40000000, 1, s, n) != store.get(40000000, 1, s, n)) ++bad;
        }
    }
    std::cout << (bad ? "differs" : "same") << std::endl;
    return bad != 0;
}
''' % ROOT)
    exe = tmp_path / "rt"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-o", str(exe), str(src)])
    import random
    rnd = random.Random(5)

    def rec(frag, end, n=None):
        s = "".join(rnd.choice("ACGT") for _ in range(n if n is not None else rnd.randint(0, 90)))
        return "@%d/%d\n%s\n+\n%s\n" % (frag, end, s, "I" * len(s))
    body = "".join(rec(1000 + k // 2, 1 + k % 2) for k in range(40000))
    files = {
        "plain.fastq": body,
        "dups.fastq": body + "".join(rec(1000 + k, 1) for k in range(0, 20000, 7)),           # later records replace earlier ones
        "far.fastq": body + rec(40000000, 2) + rec(3, 1) + rec(40000000, 2),                   # ids the dense table does not reach
        "truncated.fastq": body + "@77/1\nACGT\n+\n",                                          # incomplete last record: ignored
        "nonewline.fastq": body + "@78/2\nACGT\n+\nIIII",                                      # last line without a newline: a record
        "badname.fastq": body[:len(body) // 2] + "X9/1\nAC\n+\nII\n" + body[len(body) // 2:],  # stops the reading there
        "badend.fastq": rec(5, 1) * 300 + "@9/3\nAC\n+\nII\n" + body,
        "noslash.fastq": rec(5, 1) * 3000 + "@9\nAC\n+\nII\n" + body,
        "badint.fastq": body[:len(body) // 3] + "@9x/1\nAC\n+\nII\n" + body,
        "empty.fastq": "",
        "tiny.fq": rec(1, 1, 10),
        "wrong.txt": body,
    }
    # records are cut at multiples of four lines wherever the pieces begin: a sequence line that starts with '@'
    files["atsign.fastq"] = "".join("@%d/1\n@CGT\n+\n@III\n" % k for k in range(5000))
    for name, text in files.items():
        (tmp_path / name).write_text(text)
    r = subprocess.run([str(exe)] + [str(tmp_path / n) for n in files] + [str(tmp_path / "missing.fastq")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == "same", r.stderr[-3000:]


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_the_whole_host_pipeline_under_sanitizers(double, case, kind):
    """The same runs with the AddressSanitizer+UBSan and the ThreadSanitizer builds of the tool (bin/asan, bin/tsan): teams,
    barriers, the shared slots, the writer and the worker (process and thread) finish clean and give the oracle's bytes."""
    from defuse_amd import build
    from tests.test_sanitizers import ENV, BAD
    tool = build.build_sanitized(kind)["dosplitalign"]
    c, exp, d = case
    probe = subprocess.run([tool, "--version"], capture_output=True, text=True, env=dict(os.environ, **ENV[kind]))
    if "FATAL: ThreadSanitizer" in probe.stderr:
        pytest.skip("ThreadSanitizer cannot start in this environment: " + probe.stderr.splitlines()[0])
    for env in ({"DEFUSE_THREADS": "4", "DEFUSE_DSA_BATCH_PAIRS": "40"},
                {"DEFUSE_THREADS": "3", "DEFUSE_DSA_BATCH_PAIRS": "90", "DEFUSE_DSA_INPROCESS": "1"},
                {"DEFUSE_THREADS": "5", "DEFUSE_DSA_BATCH_REF_BYTES": "5000"}):
        out = str(d / ("san_%s.align" % kind))
        r = run_tool(c, out, double, env, tool=tool, extra_env=ENV[kind])
        assert not any(b in r.stderr for b in BAD), r.stderr[-3000:]
        assert r.returncode == 0, r.stderr[-2000:]
        assert open(out).read() == exp


@pytest.mark.parametrize("env", [{"DEFUSE_THREADS": "1"}, {"DEFUSE_THREADS": "5", "DEFUSE_DSA_BATCH_PAIRS": "9"}, {"DEFUSE_THREADS": "16", "DEFUSE_DSA_BATCH_PAIRS": "40"}],
                         ids=lambda e: ",".join("%s=%s" % (k.replace("DEFUSE_", ""), v) for k, v in e.items()))
def test_sorted_output_is_gnu_sorts(double, case, env):
    """Fused mode's --sorted (DEFUSE_FUSED=1): the alignments in the order of `LC_ALL=C sort -n -k 1` — indexed by a team over
    the batches' texts, brought into fusion order by a counting sort (stable: ties keep the order of arrival), every fusion's
    lines in byte order.  Against GNU sort on the oracle's text, with one text and with dozens."""
    c, exp, d = case
    out = str(d / ("sorted_%d.align" % abs(hash(tuple(sorted(env.items()))))))
    e = dict(os.environ, DEFUSE_DSA_LIB=double, DEFUSE_FUSED="1", **env)
    r = subprocess.run([TOOL] + pipeline_case.tool_args(c, out) + ["--sorted"], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = subprocess.run(["sort", "-n", "-k", "1"], input=exp, capture_output=True, text=True, env=dict(os.environ, LC_ALL="C"), check=True).stdout
    assert open(out).read() == want


def test_runs_inside_a_limit_on_address_space(double, case):
    """the pipeline's submitters give a job 6 GB.  Where that is enforced on ADDRESS SPACE (`ulimit -v`, h_vmem) the
    batch slots — reserved at their largest useful size, which is address space only — are reserved smaller instead of failing, and
    the batches are cut to fit: the same file.  (The HIP runtime itself wants more address space than such a limit leaves: on a GPU
    box the limit has to be on resident memory.  The test double has no runtime.)"""
    c, exp, d = case
    out = str(d / "vlimit.align")
    cmd = "ulimit -v 6000000; exec %s %s" % (TOOL, " ".join(pipeline_case.tool_args(c, out)))
    r = subprocess.run(["bash", "-c", cmd], capture_output=True, text=True, env=dict(os.environ, DEFUSE_DSA_LIB=double, DEFUSE_THREADS="4"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(out).read() == exp


def test_sorted_by_environment_keeps_the_reference_command_line(double, case):
    """DEFUSE_DSA_SORTED=1 (no fused-mode options): the same alignments in the order of `LC_ALL=C sort -n -k 1`."""
    c, exp, d = case
    out = str(d / "sorted_env.align")
    r = run_tool(c, out, double, {"DEFUSE_DSA_SORTED": "1", "DEFUSE_THREADS": "3", "DEFUSE_DSA_BATCH_PAIRS": "30"})
    assert r.returncode == 0, r.stderr[-2000:]
    want = subprocess.run(["sort", "-n", "-k", "1"], input=exp, capture_output=True, text=True, env=dict(os.environ, LC_ALL="C"), check=True).stdout
    assert open(out).read() == want
