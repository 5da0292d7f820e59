"""Synthetic compact spanning-alignment inputs for clustermatepairs (test infrastructure)."""
import numpy as np


def locus_fragments(rng, frag0, n, chr_a, strand_a, break_a, chr_b, strand_b, break_b, read_len=50, ufrag=300.0, sfrag=30.0):
    """n fragments spanning a fusion: end 1 upstream of break_a on (chr_a,strand_a), end 2 upstream of break_b."""
    lines = []
    for k in range(n):
        frag_len = int(rng.normal(ufrag, sfrag))
        inner = max(0, frag_len - 2 * read_len)
        da = int(rng.integers(0, inner + 1))
        db = inner - da

        def place(strand, brk, d):
            if strand == "+":
                end = brk - d
                return end - read_len + 1, end
            start = brk + d
            return start, start + read_len - 1
        sa, ea = place(strand_a, break_a, da)
        sb, eb = place(strand_b, break_b, db)
        lines.append("%d\t0\t%s\t%s\t%d\t%d\n" % (frag0 + k, chr_a, strand_a, sa, ea))
        lines.append("%d\t1\t%s\t%s\t%d\t%d\n" % (frag0 + k, chr_b, strand_b, sb, eb))
    return lines


def two_loci(seed=1):
    """The shape of SURVEY.md Appendix A's mini check: 12 fragments chrA+/chrB-, 8 fragments chrA-/chrB+."""
    rng = np.random.default_rng(seed)
    return locus_fragments(rng, 0, 12, "chrA", "+", 650, "chrB", "-", 1000) + \
        locus_fragments(rng, 12, 8, "chrA", "-", 2000, "chrB", "+", 2600)


def many_loci(seed, n_loci=12, decoys=True):
    rng = np.random.default_rng(seed)
    lines = []
    frag = 0
    chroms = ["chr1", "chr2", "chr3"]
    for l in range(n_loci):
        ca, cb = rng.choice(chroms, size=2, replace=True)
        n = int(rng.integers(3, 40))
        ba, bb = int(rng.integers(2000, 150000)), int(rng.integers(2000, 150000))
        if ca == cb and abs(ba - bb) < 5000:
            bb += 20000
        lines += locus_fragments(rng, frag, n, ca, "+-"[int(rng.integers(0, 2))], ba, cb, "+-"[int(rng.integers(0, 2))], bb)
        frag += n
        if l % 3 == 0:                                   # a second breakpoint 150 bp away: K > 1 mixtures
            n2 = int(rng.integers(5, 20))
            lines += locus_fragments(rng, frag, n2, ca, "+", ba + 150, cb, "-", bb + 40)
            frag += n2
    if decoys:
        for k in range(15):                              # concordant fragments: both ends close on one chromosome
            p = int(rng.integers(1000, 100000))
            lines.append("%d\t0\tchr1\t+\t%d\t%d\n" % (frag, p, p + 49))
            lines.append("%d\t1\tchr1\t-\t%d\t%d\n" % (frag, p + 200, p + 249))
            frag += 1
        for k in range(6):                               # multi-mapping end
            p = int(rng.integers(1000, 100000))
            lines.append("%d\t0\tchr2\t+\t%d\t%d\n" % (frag, p, p + 49))
            lines.append("%d\t1\tchr3\t-\t%d\t%d\n" % (frag, p + 7000, p + 7049))
            lines.append("%d\t1\tchr3\t-\t%d\t%d\n" % (frag, p + 7010, p + 7059))
            frag += 1
    return lines
