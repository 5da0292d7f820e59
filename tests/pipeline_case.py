"""File-level synthetic inputs for the dosplitalign / evalsplitalign tools (test infrastructure).

Builds FASTA + exon table + fusion regions, then uses the oracle's own task geometry to place mate
alignments inside the mate regions and to cut split reads out of the reference windows, so that the
candidate enumeration and the DP both have real work.  Seeded; small."""
import os

import numpy as np

from tests.cases import rnd, mutate


def rc(s: bytes) -> bytes:
    return s[::-1].translate(bytes.maketrans(b"ACGTacgt", b"TGCAtgca"))


def build(outdir, seed=5, n_fusions=6, reads_per_fusion=30, lq=50, ufrag=300.0, sfrag=30.0, chrom_len=5000, transcripts=True):
    from oracle import dosplitalign_oracle as ora
    rng = np.random.default_rng(seed)
    os.makedirs(outdir, exist_ok=True)
    P = lambda n: os.path.join(outdir, n)
    chroms = {"chr%d" % k: rnd(rng, chrom_len + 700 * k) for k in range(1, 4)}   # chrom_len: spreads many fusions out
    # two transcripts given as cDNA sequences with their own names "gene|transcript"
    tx = {"ENSG01|ENST01": ("chr1", "+", [(501, 900), (1501, 2100), (3001, 3800)]),
          "ENSG02|ENST02": ("chr2", "-", [(801, 1500), (2501, 3300)])}
    seqs = dict(chroms)
    for name, (c, strand, exons) in tx.items():
        s = b"".join(chroms[c][b - 1:e] for b, e in exons)
        seqs[name] = s if strand == "+" else rc(s)
    with open(P("ref.fa"), "wb") as f:
        for name, s in seqs.items():
            f.write(b">" + name.encode() + b" some description\n")
            for k in range(0, len(s), 70):
                f.write(s[k:k + 70] + b"\n")
    with open(P("exons.txt"), "w") as f:
        for name, (c, strand, exons) in tx.items():
            g, t = name.split("|")
            f.write("\t".join([g, t, c, strand] + [str(x) for be in exons for x in be]) + "\t\n")
        f.write("ENSG03\tENST03\tchr3\t+\t100\t4000\t\n")
        f.write("short line\n\n")
    # fusion regions: all strand combinations, chromosome and transcript references, one near a sequence start
    names = list(seqs) if transcripts else list(chroms)       # transcripts=False: regions on chromosomes only
    regions = []
    for k in range(n_fusions):
        ends = []
        for ce in (0, 1):
            name = names[int(rng.integers(0, len(names)))]
            L = len(seqs[name])
            if k == 0 and ce == 0:
                start = 20                                         # window clipped at the sequence start
            elif k == 1 and ce == 1:
                start = L - 130                                    # window clipped at the sequence end
            else:
                start = int(rng.integers(400, L - 600))
            ends.append((name, "+-"[int(rng.integers(0, 2))], start, start + int(rng.integers(80, 130))))
        regions.append(ends)
    with open(P("regions.txt"), "w") as f:
        for k, ends in enumerate(regions):
            for ce, (name, strand, s, e) in enumerate(ends):
                f.write("%d\t%d\t%s\t%s\t%d\t%d\n" % (10 + 3 * k, ce, name, strand, s, e))
        f.write("\nbad\tline\n")
    tasks = ora.create_tasks(P("ref.fa"), P("exons.txt"), ufrag, sfrag, lq, lq, ora.read_align_region_pairs(P("regions.txt")))
    fq = {1: [], 2: []}
    sam = ["@HD\tVN:1.0", "@SQ\tSN:chr1\tLN:5700"]
    frag = 0
    for t in tasks.values():
        for r in range(reads_per_fusion):
            ce = int(rng.integers(0, 2))
            loc = t.mate_regions[ce][int(rng.integers(0, len(t.mate_regions[ce])))]
            lo, hi = max(1, loc["start"]), max(1, loc["end"])
            pos = int(rng.integers(lo, hi + 1)) if hi >= lo else lo
            mate_end = int(rng.integers(1, 3))                     # which FASTQ end the aligned mate is
            other = 3 - mate_end
            s0, s1 = t.seq[0], t.seq[1]
            kind = int(rng.integers(0, 6))
            if kind <= 3 and len(s0) > lq and len(s1) > lq:
                a = int(rng.integers(0, lq + 1))
                x = int(rng.integers(a, len(s0) + 1))
                y = int(rng.integers(0, len(s1) - (lq - a) + 1))
                target = s0[x - a:x] + s1[y:y + lq - a]
            elif kind == 4:
                target = rnd(rng, lq)
            else:
                target = s0[:lq]
            target = mutate(rng, target, 0.02)
            revcomp = ce == 0
            read = rc(target) if revcomp else target
            if r % 11 != 10:                                        # every 11th read is missing from the FASTQ
                fq[other].append((frag, read))
            fq[mate_end].append((frag, rnd(rng, lq)))
            flag = 16 if loc["strand"] == 1 else 0
            line = "%d/%d\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s" % (frag, mate_end, flag, loc["refName"], pos, lq,
                                                                      "A" * lq, "I" * lq)
            sam.append(line)
            if r % 7 == 0:
                sam.append(line)                                    # duplicate record: de-duplicated candidate
            if r % 13 == 0:
                sam.append("%d/%d\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s" % (frag, mate_end, "A" * lq, "I" * lq))
            frag += 1
    for e in (1, 2):
        with open(P("reads.%d.fastq" % e), "wb") as f:
            for fr, s in fq[e]:
                f.write(b"@%d/%d\n%s\n+\n%s\n" % (fr, e, s, b"I" * len(s)))
    with open(P("improper.sam"), "w") as f:
        f.write("\n".join(sam) + "\n")
    return dict(fasta=P("ref.fa"), exons=P("exons.txt"), regions=P("regions.txt"), improper=P("improper.sam"),
                seq1=P("reads.1.fastq"), seq2=P("reads.2.fastq"), ufrag=ufrag, sfrag=sfrag, minread=lq, maxread=lq)


def tool_args(case, out):
    return ["-f", case["fasta"], "-e", case["exons"], "-u", str(case["ufrag"]), "-s", str(case["sfrag"]),
            "-n", str(case["minread"]), "-x", str(case["maxread"]), "-r", case["regions"], "-i", case["improper"],
            "-1", case["seq1"], "-2", case["seq2"], "-a", out]
