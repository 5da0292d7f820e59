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
        kw.setdefault("env", dict(os.environ, DEFUSE_FUSED="1"))          # (the fused-mode options of dosplitalign exist only on request)
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


# ------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4] at test size: 2x150 bp, mu = 450, sigma = 45, the four tools and
# the glue of scripts/defuse_run.pl:455-542, EVERY intermediate file compared: tool stages with the oracles, glue
# stages with the outputs of the reference's own Perl scripts (tests/golden/config5/, tests/golden/make_config5.py).
# ------------------------------------------------------------------------------------------------------------
G5 = os.path.join(ROOT, "tests", "golden", "config5")


def _pos_pairs(lines):
    from collections import defaultdict
    frag = defaultdict(dict)
    for l in lines:
        f = l.split("\t")
        frag[int(f[2])][f[1]] = int(f[6]) if f[5] == "+" else int(f[7])
    return sorted((v["0"], v["1"]) for v in frag.values())


def _by_cluster(text):
    from collections import defaultdict
    d = defaultdict(list)
    for l in text.splitlines():
        d[int(l.split("\t")[0])].append(l)
    return d


@pytest.mark.gpu
def test_config5_shaped_chain_every_intermediate_file(built, tmp_path):
    import hashlib
    import json
    from defuse_amd import build
    from oracle import clustermatepairs_oracle as co, setcover_oracle as so, dosplitalign_oracle as do
    from tests import e2e_case
    build.build_tools()
    case = e2e_case.build(str(tmp_path / "case"))
    digest = json.load(open(os.path.join(G5, "inputs.md5.json")))
    for p in [case[k] for k in ("fasta", "exons", "improper", "seq1", "seq2")] + case["spanning"]:
        assert hashlib.md5(open(p, "rb").read()).hexdigest() == digest[os.path.basename(p)], p      # the inputs the Perl fixtures were made on
    d = str(tmp_path) + "/"

    def run(tool, *args, **kw):
        kw.setdefault("env", dict(os.environ, DEFUSE_FUSED="1"))          # (the fused-mode options of dosplitalign exist only on request)
        r = subprocess.run([os.path.join(BIN, tool)] + list(args), capture_output=True, text=True, **kw)
        assert r.returncode == 0, (tool, r.stderr)
        return r

    # clustermatepairs per chromosome pair, fed through `cat files |` as the pipeline does (-a -)
    cl_paths, n_clusters = [], 0
    for sp in case["spanning"]:
        out = d + "clusters." + os.path.basename(sp).split(".", 1)[1]
        run("clustermatepairs", "-m", "5", "-p", "0.95", "-u", "450", "-s", "45", "-a", "-", "-c", out, input=open(sp).read())
        exp, n = co.clustermatepairs(open(sp).readlines(), 450.0, 45.0, 0.95, 5, em="c")
        assert open(out).read() == exp, sp
        n_clusters += n
        cl_paths.append(out)
    assert n_clusters >= 6
    merged = run("defuse_glue", "merge_clusters", *cl_paths).stdout
    assert merged == open(os.path.join(G5, "clusters.all.perl.txt")).read()                    # deterministic script: byte for byte
    open(d + "clusters.all", "w").write(merged)
    run("setcover", "-m", "5", "-c", d + "clusters.all", "-o", d + "clusters.sc.all")
    assert open(d + "clusters.sc.all").read() == so.setcover(d + "clusters.all", 5)
    sc = run("defuse_glue", "remove_duplicates", "5", input=open(d + "clusters.sc.all").read()).stdout
    g, e = _by_cluster(sc), _by_cluster(open(os.path.join(G5, "clusters.sc.perl.txt")).read())
    assert sorted(g) == sorted(e) and len(sc.splitlines()) < len(open(d + "clusters.sc.all").read().splitlines())   # duplicates were planted
    for cid in g:                                                   # which duplicate stays is Perl hash order in the script
        assert len(g[cid]) == len(e[cid]) and _pos_pairs(g[cid]) == _pos_pairs(e[cid])
    regions = run("defuse_glue", "get_align_regions", input=sc).stdout
    assert sorted(regions.splitlines()) == sorted(open(os.path.join(G5, "clusters.sc.regions.perl.txt")).read().splitlines())
    open(d + "clusters.sc.regions", "w").write(regions)
    common = ["-f", case["fasta"], "-e", case["exons"], "-u", "450", "-s", "45", "-n", "150", "-x", "150", "-r", d + "clusters.sc.regions"]
    ocommon = (case["fasta"], case["exons"], 450.0, 45.0, 150, 150, d + "clusters.sc.regions")
    run("dosplitalign", *common, "-i", case["improper"], "-1", case["seq1"], "-2", case["seq2"], "-a", d + "splitreads.alignments")
    exp = do.dosplitalign(*ocommon, case["improper"], case["seq1"], case["seq2"])
    assert open(d + "splitreads.alignments").read() == exp and len(exp.splitlines()) > 1000
    r = subprocess.run(["sort", "-n", "-k", "1", d + "splitreads.alignments"], capture_output=True, text=True, env=dict(os.environ, LC_ALL="C"))
    assert r.returncode == 0
    open(d + "splitreads.alignments.sorted", "w").write(r.stdout)
    run("evalsplitalign", *common, "-a", d + "splitreads.alignments.sorted", "-q", d + "out.seq", "-b", d + "out.break", "-p", d + "out.predalign")
    seq, brk, pred = do.evalsplitalign(*ocommon, d + "splitreads.alignments.sorted")
    assert open(d + "out.seq").read() == seq and open(d + "out.break").read() == brk and open(d + "out.predalign").read() == pred
    # the fused mode: one dosplitalign process from the set-cover clusters to the predictions, writing the same
    # intermediate files as the separate steps above
    open(d + "clusters.sc", "w").write(sc)
    run("dosplitalign", "-f", case["fasta"], "-e", case["exons"], "-u", "450", "-s", "45", "-n", "150", "-x", "150",
        "--clusters", d + "clusters.sc", "-r", d + "fused.regions", "-i", case["improper"], "-1", case["seq1"], "-2", case["seq2"],
        "-a", d + "fused.alignments", "--sorted", "-q", d + "fused.seq", "-b", d + "fused.break", "-p", d + "fused.predalign")
    assert open(d + "fused.regions").read() == regions
    assert open(d + "fused.alignments").read() == open(d + "splitreads.alignments.sorted").read()
    assert (open(d + "fused.seq").read(), open(d + "fused.break").read(), open(d + "fused.predalign").read()) == (seq, brk, pred)
    found = {tuple(l.split("\t")[2:5]) for l in brk.splitlines()}
    for (ca, sa, ba, cb, sb, bb) in case["planted"][2:5]:            # the planted junctions come back (those without microhomology exactly)
        assert (ca, sa, str(ba)) in found and (cb, sb, str(bb)) in found
