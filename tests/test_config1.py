"""BASELINE.json configs[0]: dosplitalign on the reference's bundled tools/discordant.test.sam + tools/rna.breaks through
the companion fixture of tests/config1_case.py.  The literal rna.breaks and its six-column
form need no GPU (parse error / no candidate); the regions derived from the SAM's dominant gene pair run the DP."""
import os
import subprocess

import pytest

from tests import config1_case as c1

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "bin", "dosplitalign")
EVAL = os.path.join(ROOT, "bin", "evalsplitalign")


@pytest.fixture(scope="module")
def case(built, tmp_path_factory):
    from defuse_amd import build
    build.build_tools()
    return c1.build(str(tmp_path_factory.mktemp("config1")))


def expected(tag, name):
    return open(os.path.join(c1.DATA, "expected.%s.%s.txt" % (tag, name))).read()


def args(case, regions, out):
    return ["-f", case["fasta"], "-e", case["exons"], "-u", "200", "-s", "30", "-n", "50", "-x", "50", "-r", regions,
            "-i", case["improper"], "-1", case["seq1"], "-2", case["seq2"], "-a", out]


def test_fixture_shape(case):
    assert case["n_fragments"] == 183
    sam = open(case["improper"]).read().splitlines()
    assert len(sam) == 1647 and all(l.split("\t")[0].split("/")[0].isdigit() for l in sam)
    assert open(case["regions_literal"]).read() == open(os.path.join(c1.DATA, "rna.breaks")).read()
    assert len(open(case["regions_derived"]).read().splitlines()) == 4


def test_literal_rna_breaks_is_a_parse_error(case, tmp_path):
    """Five columns: ReadAlignRegionPairs casts the reference name to an int (tools/Parsers.cpp:243-244) and exits 1 with
    the line on stdout.  The drop-in binary must do exactly that, before any GPU work."""
    from oracle import dosplitalign_oracle as ora
    r = subprocess.run([TOOL] + args(case, case["regions_literal"], str(tmp_path / "o")), capture_output=True, text=True)
    first = open(case["regions_literal"]).readline()
    assert r.returncode == 1 and r.stdout.endswith("Failed to interpret region:\n" + first)
    with pytest.raises(SystemExit):
        ora.read_align_region_pairs(case["regions_literal"])


def test_rna_breaks_regions_reach_no_candidate(case, tmp_path):
    """With the cluster-end column put back the 12 fusions parse, but no mate alignment of the SAM lies in their mate
    regions on the right strand: empty output, exit 0, no GPU needed (the reference's CPU path gives the same)."""
    from oracle import dosplitalign_oracle as ora
    out = tmp_path / "split.align"
    r = subprocess.run([TOOL] + args(case, case["regions_6col"], str(out)), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.read_text() == "" == expected("6col", "align")
    tasks = ora.create_tasks(case["fasta"], case["exons"], 200.0, 30.0, 50, 50, ora.read_align_region_pairs(case["regions_6col"]))
    assert len(tasks) == 12
    reads = {}
    ora.read_fastq(case["seq1"], reads)
    ora.read_fastq(case["seq2"], reads)
    assert list(ora.enumerate_candidates(tasks, reads, case["improper"])) == []
    # evalsplitalign on an empty alignment file: three empty outputs
    pre = str(tmp_path / "pred")
    r = subprocess.run([EVAL, "-f", case["fasta"], "-e", case["exons"], "-u", "200", "-s", "30", "-n", "50", "-x", "50",
                        "-r", case["regions_6col"], "-a", str(out), "-q", pre + ".seq", "-b", pre + ".break", "-p", pre + ".predalign"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert all(open(pre + e).read() == "" for e in (".seq", ".break", ".predalign"))


def test_oracle_on_the_derived_regions_is_stable(case):
    """The oracle's outputs on the regions derived from the SAM's dominant gene pair are committed (regression only)."""
    align, seq, brk, pred = c1.oracle_outputs(case, case["regions_derived"])
    assert align == expected("derived", "align") and len(align.splitlines()) == 41
    assert (seq, brk, pred) == (expected("derived", "seq"), expected("derived", "break"), expected("derived", "predalign"))


@pytest.mark.gpu
def test_derived_regions_through_the_tools(case, tmp_path):
    """218 candidates (one fusion with a region wider than any fragment, hence empty windows; one on transcripts with
    remapped mate regions): dosplitalign, sort, evalsplitalign byte for byte against the oracle."""
    out = tmp_path / "split.align"
    r = subprocess.run([TOOL] + args(case, case["regions_derived"], str(out)), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.read_text() == expected("derived", "align")
    rows = sorted(out.read_text().splitlines(True), key=lambda l: int(l.split("\t")[0]))
    srt = tmp_path / "sorted.align"
    srt.write_text("".join(rows))
    pre = str(tmp_path / "pred")
    r = subprocess.run([EVAL, "-f", case["fasta"], "-e", case["exons"], "-u", "200", "-s", "30", "-n", "50", "-x", "50",
                        "-r", case["regions_derived"], "-a", str(srt), "-q", pre + ".seq", "-b", pre + ".break", "-p", pre + ".predalign"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for e in ("seq", "break", "predalign"):
        assert open(pre + "." + e).read() == expected("derived", e), e
