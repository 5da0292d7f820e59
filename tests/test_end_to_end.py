"""The drop-in tools chained as scripts/defuse_run.pl chains them (:455,461,476,506,512,521,528,542):
clustermatepairs -> merge_clusters -> setcover -> remove_duplicates -> get_align_regions (bin/defuse_glue; the regions
rule of scripts/get_align_regions.pl:14-53 is also restated here as a cross-check) -> dosplitalign -> sort -n -k 1 ->
evalsplitalign, on the genome of the known-answer vector
with spanning fragments planted around its fusion chrA:+:650 | chrB:-:1000.  The final breakpoints must
be the planted ones (the same check SURVEY.md Appendix A records for the reference + its Perl glue)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests import cmp_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMOKE = os.path.join(ROOT, "tests", "golden", "smoke")
BIN = os.path.join(ROOT, "bin")


def align_regions(cluster_text):
    """scripts/get_align_regions.pl: per (cluster, end) the reference, strand and min start / max end."""
    reg = {}
    for line in cluster_text.splitlines():
        f = line.split("\t")
        key = (int(f[0]), int(f[1]))
        s, e = int(f[6]), int(f[7])
        if key not in reg:
            reg[key] = [f[4], f[5], s, e]
        reg[key][2] = min(reg[key][2], s)
        reg[key][3] = max(reg[key][3], e)
    return "".join("%d\t%d\t%s\t%s\t%d\t%d\n" % (k[0], k[1], v[0], v[1], v[2], v[3]) for k, v in sorted(reg.items()))


@pytest.mark.gpu
def test_pipeline_recovers_planted_breakpoint(built, tmp_path):
    from defuse_amd import build
    build.build_tools()
    for n in ("ref.fa", "exons.txt", "improper.sam", "reads.1.fastq", "reads.2.fastq"):
        shutil.copy(os.path.join(SMOKE, n), tmp_path)
    d = str(tmp_path) + "/"
    rng = np.random.default_rng(4)
    # spanning fragments: end 1 upstream of chrA:650 on +, end 2 downstream of chrB:1000 on -; a second, weaker locus
    lines = cmp_cases.locus_fragments(rng, 100, 14, "chrA", "+", 650, "chrB", "-", 1000) + \
        cmp_cases.locus_fragments(rng, 200, 3, "chrA", "-", 2300, "chrB", "+", 2600)
    (tmp_path / "spanning.txt").write_text("".join(lines))

    def run(tool, *args, **kw):
        r = subprocess.run([os.path.join(BIN, tool)] + list(args), capture_output=True, text=True, **kw)
        assert r.returncode == 0, (tool, r.stderr)
        return r

    r = run("clustermatepairs", "-m", "5", "-p", "0.95", "-u", "300", "-s", "30", "-a", "-", "-c", d + "clusters.txt",
            input="".join(lines))
    assert "Created 1 clusters" in r.stdout                       # the 3-fragment locus is below -m 5
    merged = run("defuse_glue", "merge_clusters", d + "clusters.txt").stdout          # one chromosome pair here: ids unchanged
    assert merged == open(d + "clusters.txt").read()
    (tmp_path / "clusters.all").write_text(merged)
    run("setcover", "-m", "5", "-c", d + "clusters.all", "-o", d + "clusters.sc.all")
    sc = run("defuse_glue", "remove_duplicates", "5", input=open(d + "clusters.sc.all").read()).stdout
    assert len(sc.splitlines()) >= 20
    regions = run("defuse_glue", "get_align_regions", input=sc).stdout
    assert regions == align_regions(sc)
    (tmp_path / "regions.txt").write_text(regions)
    common = ["-f", d + "ref.fa", "-e", d + "exons.txt", "-u", "300", "-s", "30", "-n", "50", "-x", "50", "-r", d + "regions.txt"]
    run("dosplitalign", *common, "-i", d + "improper.sam", "-1", d + "reads.1.fastq", "-2", d + "reads.2.fastq", "-a", d + "split.align")
    rows = sorted(open(d + "split.align").read().splitlines(True), key=lambda l: int(l.split("\t")[0]))
    assert len(rows) >= 10
    (tmp_path / "split.sorted").write_text("".join(rows))
    run("evalsplitalign", *common, "-a", d + "split.sorted", "-q", d + "out.seq", "-b", d + "out.break", "-p", d + "out.predalign")
    assert open(d + "out.break").read() == "0\t0\tchrA\t+\t650\n0\t1\tchrB\t-\t1000\n"
    seq = open(d + "out.seq").read().split("\t")
    assert "|" in seq[1] and int(seq[3]) >= 10
