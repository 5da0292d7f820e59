"""BASELINE.json configs[0]: dosplitalign on the reference's bundled tools/discordant.test.sam + tools/rna.breaks
(copies under tests/golden/config1/).  Test infrastructure.

The two files are not runnable as they are: fragment names are not integers
(tools/SplitAlignment.cpp:282 casts them), no FASTA / exon table / FASTQ files come with them, and rna.breaks has the
older five-column layout (id, reference, strand, start, end), on which ReadAlignRegionPairs (tools/Parsers.cpp:229-254)
exits with "Failed to interpret region".  The companion fixture follows SURVEY.md 8(d) config 1:

  * fragments renamed 0..182 in order of first appearance;
  * FASTQ 1/2 rebuilt from the SAM's SEQ column (reverse-complemented when flag 0x10 is set; first record of a read wins;
    reads whose end never appears in the SAM are missing: the empty-read path of DoAlignment);
  * FASTA: one random sequence (seed 1) per distinct reference name, long enough for its largest POS and for the
    coordinates of rna.breaks, with every record's SEQ written in at its POS so that the windows carry real sequence;
  * exon table: every bare gene name (treated as a chromosome) gets one transcript over its whole length, every
    `gene|transcript` name one transcript on the chromosome named after its gene;
  * region files: rna.breaks literally; rna.breaks with the cluster-end column put back (rows of one id alternate 0/1);
    and regions derived by the rule of scripts/get_align_regions.pl from the SAM's dominant gene pair
    ENSG00000068323 / ENSG00000124782 (bare-gene alignments -> fusion 0, their most frequent transcripts -> fusion 1).
  mu = 200, sigma = 30, min = max read length = 50."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "config1")
COMP = bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")
UFRAG, SFRAG, RL = 200.0, 30.0, 50
GENE_A, GENE_B = "ENSG00000068323", "ENSG00000124782"


def rc(s: bytes) -> bytes:
    return s[::-1].translate(COMP)


def build(outdir):
    os.makedirs(outdir, exist_ok=True)
    P = lambda n: os.path.join(outdir, n)
    recs = []
    for line in open(os.path.join(DATA, "discordant.test.sam")):
        f = line.rstrip("\n").split("\t")
        name, end = f[0].rsplit("/", 1)
        recs.append((name, int(end), int(f[1]), f[2], int(f[3]), f[9].encode(), f))
    frag_id = {}
    for r in recs:
        frag_id.setdefault(r[0], len(frag_id))
    # reference sequences
    need = {}
    for (_, _, _, rname, pos, seq, _) in recs:
        need[rname] = max(need.get(rname, 0), pos + len(seq) + 1000)
    breaks = [l.rstrip("\n").split("\t") for l in open(os.path.join(DATA, "rna.breaks")) if l.strip()]
    for b in breaks:
        need[b[1]] = max(need.get(b[1], 0), int(b[4]) + 2000)
    for rname in list(need):
        gene = rname.split("|")[0]
        need.setdefault(gene, 3000)                       # the chromosome a transcript's exon row points at
    rng = np.random.default_rng(1)
    seqs = {}
    for rname in sorted(need):
        seqs[rname] = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=need[rname]).tobytes())
    for (_, _, _, rname, pos, seq, _) in recs:
        seqs[rname][pos - 1:pos - 1 + len(seq)] = seq
    with open(P("ref.fa"), "wb") as f:
        for rname in sorted(seqs):
            s = bytes(seqs[rname])
            f.write(b">" + rname.encode() + b"\n")
            for k in range(0, len(s), 60):
                f.write(s[k:k + 60] + b"\n")
    with open(P("exons.txt"), "w") as f:
        for rname in sorted(seqs):
            if "|" in rname:
                g, t = rname.split("|")
                f.write("%s\t%s\t%s\t+\t1\t%d\t\n" % (g, t, g, len(seqs[rname])))
            else:
                f.write("%s\t%s\t%s\t+\t1\t%d\t\n" % (rname, "ENST_whole_" + rname[4:], rname, len(seqs[rname])))
    # FASTQ and the SAM with integer fragment names
    reads = {1: {}, 2: {}}
    with open(P("improper.sam"), "w") as f:
        for (name, end, flag, rname, pos, seq, fields) in recs:
            fid = frag_id[name]
            reads[end].setdefault(fid, rc(seq) if flag & 16 else seq)
            f.write("\t".join(["%d/%d" % (fid, end)] + fields[1:]) + "\n")
    for e in (1, 2):
        with open(P("reads.%d.fastq" % e), "wb") as f:
            for fid in sorted(reads[e]):
                s = reads[e][fid]
                f.write(b"@%d/%d\n%s\n+\n%s\n" % (fid, e, s, b"I" * len(s)))
    # region files
    with open(P("rna.breaks"), "w") as f:
        f.write(open(os.path.join(DATA, "rna.breaks")).read())
    with open(P("rna.breaks.6col"), "w") as f:
        seen = {}
        for b in breaks:
            ce = seen.get(b[0], 0)
            seen[b[0]] = ce + 1
            f.write("\t".join([b[0], str(ce)] + b[1:]) + "\n")

    def derived(ref_a, ref_b):
        """get_align_regions.pl on the fragments with one end on ref_a and the other on ref_b, restricted - as a cluster of
        clustermatepairs is - to the most frequent strand combination."""
        by_frag = {}
        for (name, end, flag, rname, pos, seq, _) in recs:
            if rname in (ref_a, ref_b):
                by_frag.setdefault(name, []).append((end, rname, "-" if flag & 16 else "+", pos, pos + len(seq) - 1))
        pairs = []
        for alns in by_frag.values():
            a = [x for x in alns if x[1] == ref_a]
            b = [x for x in alns if x[1] == ref_b]
            if a and b and a[0][0] != b[0][0]:
                pairs.append((a[0], b[0]))
        combos = {}
        for a, b in pairs:
            combos[(a[2], b[2])] = combos.get((a[2], b[2]), 0) + 1
        best = max(sorted(combos), key=lambda k: combos[k])
        reg = {}
        for a, b in pairs:
            if (a[2], b[2]) != best:
                continue
            for ce, x in ((0, a), (1, b)):
                cur = reg.setdefault(ce, [x[1], x[2], x[3], x[4]])
                cur[2], cur[3] = min(cur[2], x[3]), max(cur[3], x[4])
        return reg

    def most_frequent_transcript(gene):
        cnt = {}
        for r in recs:
            if r[3].startswith(gene + "|"):
                cnt[r[3]] = cnt.get(r[3], 0) + 1
        return max(sorted(cnt), key=lambda k: cnt[k])
    with open(P("derived.regions"), "w") as f:
        for fid, (ra, rb) in enumerate(((GENE_A, GENE_B), (most_frequent_transcript(GENE_A), most_frequent_transcript(GENE_B)))):
            reg = derived(ra, rb)
            assert len(reg) == 2, (ra, rb)
            for ce in (0, 1):
                f.write("%d\t%d\t%s\t%s\t%d\t%d\n" % (fid, ce, reg[ce][0], reg[ce][1], reg[ce][2], reg[ce][3]))
    return dict(fasta=P("ref.fa"), exons=P("exons.txt"), improper=P("improper.sam"), seq1=P("reads.1.fastq"), seq2=P("reads.2.fastq"),
                ufrag=UFRAG, sfrag=SFRAG, minread=RL, maxread=RL, n_fragments=len(frag_id),
                regions_literal=P("rna.breaks"), regions_6col=P("rna.breaks.6col"), regions_derived=P("derived.regions"))


def oracle_outputs(case, regions):
    from oracle import dosplitalign_oracle as ora
    common = (case["fasta"], case["exons"], case["ufrag"], case["sfrag"], case["minread"], case["maxread"], regions)
    align = ora.dosplitalign(*common, case["improper"], case["seq1"], case["seq2"])
    rows = sorted(align.splitlines(True), key=lambda l: int(l.split("\t")[0]))
    path = regions + ".sorted.align"
    open(path, "w").write("".join(rows))
    return (align,) + tuple(ora.evalsplitalign(*common, path))


if __name__ == "__main__":            # rewrites tests/golden/config1/expected.*
    import tempfile
    sys.path.insert(0, os.path.dirname(HERE))
    with tempfile.TemporaryDirectory() as tmp:
        case = build(tmp)
        for tag, regions in (("6col", case["regions_6col"]), ("derived", case["regions_derived"])):
            align, seq, brk, pred = oracle_outputs(case, regions)
            print(tag, len(align.splitlines()), "alignment lines;", len(brk.splitlines()), "break lines")
            for name, txt in (("align", align), ("seq", seq), ("break", brk), ("predalign", pred)):
                open(os.path.join(DATA, "expected.%s.%s.txt" % (tag, name)), "w").write(txt)
        print(open(case["regions_derived"]).read())
