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


# ------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2] at any fraction: loci pairs over 24 synthetic chromosomes
# (50-250 Mb, seed 3), support per locus ~ 1/k on 1..500 (most loci fall below -m 5), ends placed
# U(0, mu+3sigma-2*100) upstream of each breakpoint, 2x100 bp, 5 % of the fragments multi-map one end to a second
# locus, 10 % concordant decoys; the tool runs with -m 5 -p 0.95 -u 300 -s 30.  Vectorised (numpy + pandas).
# ------------------------------------------------------------------------------------------------------------
MU, SIGMA, RL = 300.0, 30.0, 100


def config3_write(n_fragments, path, seed=3):
    import pandas as pd
    rng = np.random.default_rng(seed)
    chrom_len = rng.integers(50_000_000, 250_000_001, size=24)
    k = np.arange(1, 501)
    pk = (1.0 / k) / (1.0 / k).sum()
    n_disc = int(n_fragments * 0.9)
    n_loci = max(1, int(n_disc / float((k * pk).sum())))
    support = rng.choice(k, size=n_loci, p=pk)
    n_disc = int(support.sum())
    ca, cb = rng.integers(0, 24, size=n_loci), rng.integers(0, 24, size=n_loci)
    sa, sb = rng.integers(0, 2, size=n_loci), rng.integers(0, 2, size=n_loci)          # 0 '+', 1 '-'
    ba = (rng.random(n_loci) * (chrom_len[ca] - 20000)).astype(np.int64) + 10000
    bb = (rng.random(n_loci) * (chrom_len[cb] - 20000)).astype(np.int64) + 10000
    locus = np.repeat(np.arange(n_loci), support)
    inner_max = int(MU + 3 * SIGMA) - 2 * RL

    def place(strand, brk, d):
        plus = strand == 0
        start = np.where(plus, brk - d - RL + 1, brk + d)
        return start, start + RL - 1

    da, db = rng.integers(0, inner_max + 1, size=n_disc), rng.integers(0, inner_max + 1, size=n_disc)
    s0, e0 = place(sa[locus], ba[locus], da)
    s1, e1 = place(sb[locus], bb[locus], db)
    frag = np.arange(n_disc)
    parts = [pd.DataFrame({"f": frag, "o": 0, "e": 0, "c": ca[locus], "s": sa[locus], "a": s0, "b": e0}),
             pd.DataFrame({"f": frag, "o": 1, "e": 1, "c": cb[locus], "s": sb[locus], "a": s1, "b": e1})]
    # 5 %: end 2 also aligns at a second locus
    mm = np.nonzero(rng.random(n_disc) < 0.05)[0]
    l2 = rng.integers(0, n_loci, size=len(mm))
    s2, e2 = place(sb[l2], bb[l2], rng.integers(0, inner_max + 1, size=len(mm)))
    parts.append(pd.DataFrame({"f": frag[mm], "o": 2, "e": 1, "c": cb[l2], "s": sb[l2], "a": s2, "b": e2}))
    # 10 % concordant decoys: both ends on one chromosome, a fragment length apart, opposite strands
    n_dec = n_fragments - n_disc if n_fragments > n_disc else 0
    cd = rng.integers(0, 24, size=n_dec)
    p = (rng.random(n_dec) * (chrom_len[cd] - 20000)).astype(np.int64) + 10000
    fl = np.maximum(2 * RL, rng.normal(MU, SIGMA, size=n_dec).astype(np.int64))
    fd = n_disc + np.arange(n_dec)
    parts.append(pd.DataFrame({"f": fd, "o": 0, "e": 0, "c": cd, "s": 0, "a": p, "b": p + RL - 1}))
    parts.append(pd.DataFrame({"f": fd, "o": 1, "e": 1, "c": cd, "s": 1, "a": p + fl - RL, "b": p + fl - 1}))
    df = pd.concat(parts, ignore_index=True)
    # shuffle the fragments (the aligner's output order is by read, not by locus), keep a fragment's lines together
    perm = rng.permutation(n_disc + n_dec)
    df["f"] = perm[df["f"].to_numpy()]
    df.sort_values(["f", "o"], inplace=True, kind="stable")
    df["c"] = ("chr" + (df["c"] + 1).astype(str))
    df["s"] = np.where(df["s"].to_numpy() == 0, "+", "-")
    df[["f", "e", "c", "s", "a", "b"]].to_csv(path, sep="\t", header=False, index=False)
    return n_disc + n_dec, n_loci, len(df)




def config3_lines(n_fragments, seed=3):
    """The same input as a list of text lines (for the Python oracle)."""
    import os, tempfile
    fd, path = tempfile.mkstemp(suffix=".spanning.txt")
    os.close(fd)
    try:
        config3_write(n_fragments, path, seed)
        with open(path) as f:
            return f.readlines()
    finally:
        os.remove(path)


def adversarial_em_batch(seed, n_problems=4000, mean=300.0):
    """Problems for mpe_cluster_batch that leave the beaten path of MatePairEM (found by searching with the C oracle's
    counters): far-apart tight groups and far outliers make components lose every responsibility (MaxLikelihood returns
    false, NK == 0, tools/MatePairEM.cpp:277: A/B keep their values) and make LogLikelihood underflow to -DBL_MAX (:127-131:
    that K is dropped); groups of identical points stop the KKZ seeding (:375-378); points on a line tie distances.
    Returns (prob_off, x, y, u, to_xo, to_yo) as numpy arrays."""
    from oracle import mpe_c
    rng = np.random.default_rng(seed)
    xs, ys, off = [], [], [0]
    for p in range(n_problems):
        n = int(rng.integers(5, 40))
        kind = p % 4
        if kind == 0:
            k = int(rng.integers(2, 6))
            cx, cy = rng.integers(-100000, 100000, size=k), rng.integers(-100000, 100000, size=k)
            a = rng.integers(0, k, size=n)
            x, y = cx[a] + rng.integers(0, 30, size=n), cy[a] + rng.integers(0, 30, size=n)
        elif kind == 1:
            x, y = rng.integers(0, 200, size=n), rng.integers(0, 200, size=n)
            m = int(rng.integers(1, 4))
            x[:m] += rng.integers(2000, 30000, size=m)
            y[:m] -= rng.integers(2000, 30000, size=m)
        elif kind == 2:
            k = int(rng.integers(2, 5))
            px, py = rng.integers(0, 5000, size=k), rng.integers(0, 5000, size=k)
            a = rng.integers(0, k, size=n)
            x, y = px[a], py[a]
        else:
            x = np.sort(rng.integers(0, 20000, size=n))
            y = -x + rng.integers(0, 400, size=n)
        xs.append(x.astype(np.float64))
        ys.append(y.astype(np.float64))
        off.append(off[-1] + n)
    x, y = np.concatenate(xs), np.concatenate(ys)
    u = np.full(len(x), mean - 100.0)
    to_xo = np.concatenate([mpe_c.ranks_desc(a) for a in xs])
    to_yo = np.concatenate([mpe_c.ranks_desc(a) for a in ys])
    return np.array(off, dtype=np.int64), x, y, u, to_xo, to_yo


def tie_heavy_em_batch(seed, n_problems=120, mean=300.0):
    """Larger problems (60-500 mate pairs) whose coordinates repeat: the M step's breakpoint search (mpe_api.hip,
    max_likelihood_groups) works on runs of equal coordinates and on prefix sums that tie between the two orders — with K = 1
    (every responsibility 1.0) and after the hard k-means start every sum is a whole number — so runs, ties, runs of ties and
    components without any responsibility over long stretches all occur.  Returns the arrays of adversarial_em_batch."""
    from oracle import mpe_c
    rng = np.random.default_rng(seed)
    xs, ys, off = [], [], [0]
    for p in range(n_problems):
        n = int(rng.integers(60, 500))
        kind = p % 5
        k = int(rng.integers(1, 7))
        cx, cy = rng.integers(0, 3000, size=k) * 10, rng.integers(0, 3000, size=k) * 10
        a = rng.integers(0, k, size=n)
        if kind == 0:                      # few distinct values per locus
            x, y = cx[a] + rng.integers(0, 4, size=n) * 7, cy[a] + rng.integers(0, 4, size=n) * 7
        elif kind == 1:                    # the same multiset in both orders: every prefix sum ties
            x = cx[a] + rng.integers(0, 25, size=n)
            y = x.copy()
        elif kind == 2:                    # one coordinate constant (a single run), the other spread
            x, y = np.full(n, 500), cy[a] + rng.integers(0, 200, size=n)
        elif kind == 3:                    # ordinary loci with integer jitter
            x, y = cx[a] + rng.integers(0, 120, size=n), cy[a] + rng.integers(0, 120, size=n)
        else:                              # long runs in x, all distinct in y
            x, y = cx[a] + rng.integers(0, 2, size=n), rng.permutation(n) * 3 + cy[a]
        xs.append(x.astype(np.float64))
        ys.append(y.astype(np.float64))
        off.append(off[-1] + n)
    x, y = np.concatenate(xs), np.concatenate(ys)
    u = np.full(len(x), mean - 100.0)
    to_xo = np.concatenate([mpe_c.ranks_desc(a) for a in xs])
    to_yo = np.concatenate([mpe_c.ranks_desc(a) for a in ys])
    return np.array(off, dtype=np.int64), x, y, u, to_xo, to_yo
