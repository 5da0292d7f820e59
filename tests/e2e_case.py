"""A BASELINE.json configs[4]-shaped end-to-end input at test size (2x150 bp, mu = 450,
sigma = 45, the four tools with the glue of scripts/defuse_run.pl:455-542 in between).  Test infrastructure.

A small genome with planted fusions in all four strand combinations (two of them on the same chromosome pair and 150 bp
apart, one below the support threshold), paired-end fragments drawn from the fused transcripts: fragments whose two
reads lie on either side of the junction become spanning (discordant) alignments in the compact format
`divide_sam_chr_pairs.pl` writes (one file per chromosome pair), fragments with a read across the junction become
candidates of dosplitalign (their anchored mate is a record of improper.sam, the crossing read comes from the FASTQ
files).  PCR duplicates, concordant decoys and a multi-mapping end are mixed in.  Seeded; everything is regenerated at
test time, only the outputs of the reference's Perl glue on it are committed (tests/golden/config5/)."""
import os

import numpy as np

UFRAG, SFRAG, RL = 450.0, 45.0, 150
CHROM_LEN = 40000
COMP = bytes.maketrans(b"ACGT", b"TGCA")

# (chrA, strandA, breakA, chrB, strandB, breakB, number of fragments drawn around the junction)
FUSIONS = [
    ("chr1", "+", 12000, "chr2", "-", 21000, 90),
    ("chr1", "+", 12150, "chr2", "-", 21040, 60),     # same chromosome pair, 150 bp away: mixtures with K > 1
    ("chr3", "-", 9000, "chr1", "+", 30500, 70),
    ("chr2", "+", 33000, "chr4", "+", 8000, 50),
    ("chr4", "-", 25000, "chr4", "-", 31000, 40),     # both sides on one chromosome, far apart
    ("chr3", "+", 30000, "chr2", "-", 5000, 6),       # too few spanning fragments for -m 5
]


def rc(s: bytes) -> bytes:
    return s[::-1].translate(COMP)


def _mutate(rng, s: bytes, rate):
    b = bytearray(s)
    for k in np.nonzero(rng.random(len(b)) < rate)[0]:
        b[k] = b"ACGT"[(b"ACGT".index(b[k]) + int(rng.integers(1, 4))) % 4]
    return bytes(b)


def _side_a(chrom: bytes, strand, brk):
    """Upstream part of the fused transcript and the map from an offset in it to (1-based start, end) of a length-n piece."""
    if strand == "+":
        part = chrom[:brk]
        return part, lambda q, n: (q + 1, q + n)
    part = rc(chrom[brk - 1:])
    L = len(chrom)
    return part, lambda q, n: (L - q - n + 1, L - q)


def _side_b(chrom: bytes, strand, brk):
    if strand == "-":
        part = chrom[brk - 1:]
        return part, lambda r, n: (brk + r, brk + r + n - 1)
    part = rc(chrom[:brk])
    return part, lambda r, n: (brk - r - n + 1, brk - r)


def build(outdir, seed=5):
    rng = np.random.default_rng(seed)
    os.makedirs(outdir, exist_ok=True)
    P = lambda n: os.path.join(outdir, n)
    chroms = {"chr%d" % k: bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=CHROM_LEN + 1000 * k)) for k in range(1, 5)}
    with open(P("ref.fa"), "wb") as f:
        for name, s in chroms.items():
            f.write(b">" + name.encode() + b"\n")
            for k in range(0, len(s), 60):
                f.write(s[k:k + 60] + b"\n")
    with open(P("exons.txt"), "w") as f:
        for k, (name, s) in enumerate(chroms.items(), 1):
            f.write("ENSG%02d\tENST%02d\t%s\t+\t1\t%d\t\n" % (k, k, name, len(s)))
    spanning = []                     # (fragment, [(read_end 1|2, chrom, strand, start, end), ...])
    fq = {1: [], 2: []}
    sam = ["@HD\tVN:1.0\tSO:unsorted"]
    frag = 0
    planted = []

    def sam_line(fr, read_end, name, strand, start, seq):
        return "%d/%d\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s" % (fr, read_end, 16 if strand == "-" else 0, name, start, len(seq),
                                                                 seq.decode(), "I" * len(seq))

    for (ca, sa, ba, cb, sb, bb, n) in FUSIONS:
        left, map_a = _side_a(chroms[ca], sa, ba)
        right, map_b = _side_b(chroms[cb], sb, bb)
        fused = left + right
        J = len(left)
        planted.append((ca, sa, ba, cb, sb, bb))
        drawn = []
        for k in range(n):
            flen = max(2 * RL + 10, int(rng.normal(UFRAG, SFRAG)))
            p = int(rng.integers(J - flen + 12, J - 12))
            drawn.append((p, flen))
            if k % 9 == 0:
                drawn.append((p, flen))                                  # a PCR duplicate: same positions, new fragment
        for (p, flen) in drawn:
            piece = fused[p:p + flen]
            r1, r2 = _mutate(rng, piece[:RL], 0.01), _mutate(rng, rc(piece[-RL:]), 0.01)
            fq[1].append((frag, r1))
            fq[2].append((frag, r2))
            one_in_left, two_in_right = p + RL <= J, p + flen - RL >= J
            a_reg = map_a(p, RL) if one_in_left else None                # read 1 aligns as a whole on side A
            b_reg = map_b(p + flen - RL - J, RL) if two_in_right else None
            strand2 = "-" if sb == "-" else "+"                           # read 2 is the reverse complement of the tail
            if a_reg and b_reg:
                spanning.append((frag, [(1, ca, sa, a_reg[0], a_reg[1]), (2, cb, strand2, b_reg[0], b_reg[1])]))
            # improper.sam: every end that aligns as a whole (the other end is then a candidate read of dosplitalign)
            if a_reg:
                fwd = chroms[ca][a_reg[0] - 1:a_reg[1]]
                sam.append(sam_line(frag, 1, ca, sa, a_reg[0], fwd))
            if b_reg:
                fwd = chroms[cb][b_reg[0] - 1:b_reg[1]]
                sam.append(sam_line(frag, 2, cb, strand2, b_reg[0], fwd))
            frag += 1
    # concordant decoys (dropped by clustermatepairs) and one multi-mapping end
    for k in range(25):
        name = "chr%d" % (1 + k % 4)
        p = int(rng.integers(1000, CHROM_LEN - 2000))
        flen = int(rng.normal(UFRAG, SFRAG))
        s = chroms[name]
        fq[1].append((frag, s[p - 1:p - 1 + RL]))
        fq[2].append((frag, rc(s[p - 1 + flen - RL:p - 1 + flen])))
        spanning.append((frag, [(1, name, "+", p, p + RL - 1), (2, name, "-", p + flen - RL, p + flen - 1)]))
        frag += 1
    for k in range(6):
        p, q = 14000 + 40 * k, 27000 + 30 * k
        fq[1].append((frag, chroms["chr2"][p - 1:p - 1 + RL]))
        fq[2].append((frag, rc(chroms["chr3"][q - 1:q - 1 + RL])))
        spanning.append((frag, [(1, "chr2", "+", p, p + RL - 1), (2, "chr3", "-", q, q + RL - 1), (2, "chr3", "-", q + 9, q + RL + 8)]))
        frag += 1
    # shuffle fragments as an aligner's output would be ordered by read, then split by chromosome pair exactly as
    # divide_sam_chr_pairs.pl does: file <chrX>-<chrY> with the two names in ascending order, read end column = end - 1
    order = rng.permutation(len(spanning))
    pair_files = {}
    for idx in order:
        fr, alns = spanning[idx]
        by_end = {1: [a for a in alns if a[0] == 1], 2: [a for a in alns if a[0] == 2]}
        for a1 in by_end[1]:
            for a2 in by_end[2]:
                pair = tuple(sorted((a1[1], a2[1])))
                lines = pair_files.setdefault(pair, {})
                lst = lines.setdefault(fr, [])
                for a in (a1, a2):
                    l = "%d\t%d\t%s\t%s\t%d\t%d\n" % (fr, a[0] - 1, a[1], a[2], a[3], a[4])
                    if l not in lst:
                        lst.append(l)
    span_paths = []
    for pair in sorted(pair_files):
        path = P("spanning.%s-%s" % pair)
        with open(path, "w") as f:
            for fr, lst in pair_files[pair].items():
                f.writelines(lst)
        span_paths.append(path)
    for e in (1, 2):
        with open(P("reads.%d.fastq" % e), "wb") as f:
            for fr, s in fq[e]:
                f.write(b"@%d/%d\n%s\n+\n%s\n" % (fr, e, s, b"I" * len(s)))
    with open(P("improper.sam"), "w") as f:
        f.write("\n".join(sam) + "\n")
    return dict(fasta=P("ref.fa"), exons=P("exons.txt"), improper=P("improper.sam"), seq1=P("reads.1.fastq"), seq2=P("reads.2.fastq"),
                spanning=span_paths, ufrag=UFRAG, sfrag=SFRAG, minread=RL, maxread=RL, planted=planted, n_fragments=frag)
