"""CPU restatement of the reference's `calccov` tool (tools/calccov.cpp:66-250).  TEST INFRASTRUCTURE ONLY.

  main                                tools/calccov.cpp:66-234
  BinnedSamplePositions               :30-64   (bins only pre-select; the exact range test follows, :181-187, :199-207)
  CalculateSplitPos / CalculateSplitMin   :236-250
  ExonRegions::Read / GetGenes        tools/ExonRegions.cpp:21-112, :126-129
  SamAlignmentStream / FragmentAlignmentStream   tools/AlignmentStream.cpp:39-130, :190-221 (through dosplitalign_oracle)

The sample positions come from the C library's rand() after srand(11) (:116, :137): GlibcRand below restates glibc's
TYPE_3 generator (random_r.c: 31-word additive feedback r[i] = r[i-3] + r[i-31], seeded by the Lehmer sequence
16807 x mod 2^31-1, 310 outputs discarded, result >> 1); tests/test_calccov.py checks it against the platform's libc.
Orders the reference leaves to boost::unordered_set follow SURVEY.md 8(c): genes ascending by name, a fragment's samples
ascending by index.

Parity status: UNPINNED (calccov.cpp includes Common.h and needs Boost; the reference holds no vector for it)."""
import math


class GlibcRand:
    def __init__(self, seed):
        seed = seed & 0xFFFFFFFF
        if seed == 0:
            seed = 1
        r = [0] * 34
        r[0] = seed
        for i in range(1, 31):
            prev = r[i - 1] if r[i - 1] < 0x80000000 else r[i - 1] - (1 << 32)      # int32 arithmetic of random_r.c
            hi = prev // 127773 if prev >= 0 else -((-prev) // 127773)
            lo = prev - hi * 127773
            word = 16807 * lo - 2836 * hi
            if word < 0:
                word += 2147483647
            r[i] = word & 0xFFFFFFFF
        for i in range(31, 34):
            r[i] = r[i - 31]
        self.r = r
        for _ in range(310):
            self._step()

    def _step(self):
        v = (self.r[-31] + self.r[-3]) & 0xFFFFFFFF
        self.r.append(v)
        self.r.pop(0)
        return v

    def rand(self):
        return self._step() >> 1


def read_exons(path):
    """gene -> [transcripts in file order], transcript -> length (ExonRegions::Read)."""
    gene_tr, length = {}, {}
    for line in open(path):
        line = line.rstrip("\n")
        if not line:
            continue
        f = line.split("\t")
        if len(f) < 6:
            continue
        exons = [(int(f[k - 1]), int(f[k])) for k in range(5, len(f), 2)]
        if f[3] not in ("+", "-"):
            raise SystemExit("Error: Unable to intepret strand " + f[3])
        length[f[1]] = sum(e - b + 1 for b, e in exons)
        gene_tr.setdefault(f[0], []).append(f[1])
    return gene_tr, length


def _fmt(x):
    s = "%g" % x
    return s


def calccov(conc_sam, genetran, density, anchor, trim, multiexon=False):
    """Returns the three file texts (length samples, split positions, split minimums)."""
    from oracle import dosplitalign_oracle as ora
    gene_tr, length = read_exons(genetran)
    rng = GlibcRand(11)
    ref_index, sample_pos, sample_off = {}, [], [0]
    for gene in sorted(gene_tr):
        tr = gene_tr[gene]
        if len(tr) == 1 or multiexon:
            ref_index[gene + "|" + tr[0]] = len(sample_off) - 1
            n = int(length[tr[0]] * density)
            for _ in range(n):
                sample_pos.append(rng.rand() % length[tr[0]] + 1)
            sample_off.append(len(sample_pos))
    out_len, out_pos, out_min = [], [], []

    def fragment(alns):
        if len(alns) != 2:
            raise SystemExit("Error: expected 2 alignments per fragment\nretrieved %d alignments for %s" % (len(alns), alns[0][0]))
        if alns[0][2] not in ref_index:
            return
        r = ref_index[alns[0][2]]
        (s0, e0), (s1, e1) = (alns[0][4], alns[0][5]), (alns[1][4], alns[1][5])
        us, ue = min(s0 + trim, s1 + trim), max(e0 - trim, e1 - trim)
        flen = max(e0, e1) - min(s0, s1)
        idx = range(sample_off[r], sample_off[r + 1])
        for k in idx:
            if us <= sample_pos[k] <= ue:
                out_len.append("%d\t%d\n" % (k, flen))
        for (s, e) in ((s0, e0), (s1, e1)):
            a_s, a_e = s + anchor, e - anchor + 1
            for k in idx:
                p = sample_pos[k]
                if a_s <= p <= a_e:
                    pos_value = max(0.0, float(p - s - anchor))
                    pos_range = e - s + 1.0 - 2.0 * anchor
                    min_value = max(0.0, float(min(p - s - anchor, e + 1 - p - anchor)))
                    min_range = math.floor(0.5 * (e - s + 1.0 - 2.0 * anchor))
                    out_pos.append("%d\t%s\n" % (k, _fmt(_div(pos_value, pos_range))))
                    out_min.append("%d\t%s\n" % (k, _fmt(_div(min_value, min_range))))
    cur = []
    for rec in ora.sam_alignments(conc_sam):
        if cur and rec[0] != cur[0][0]:
            fragment(cur)
            cur = []
        cur.append(rec)
    if cur:
        fragment(cur)
    return "".join(out_len), "".join(out_pos), "".join(out_min)


def _div(a, b):      # IEEE division: x/0 is +-inf or nan, not an exception
    if b == 0.0:
        return float("nan") if a == 0.0 else math.copysign(float("inf"), a)
    return a / b
