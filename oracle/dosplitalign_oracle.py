"""CPU restatement of the `dosplitalign` / `evalsplitalign` host logic.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product never does.  It follows the reference file by file (citations are relative to
/root/reference) and calls oracle/libdsa_oracle.so for the DP itself.

Iteration orders that the reference leaves to boost::unordered_* are fixed to the canonical order
of SURVEY.md section 8(c): overlapping cluster ids ascending as signed int, Evaluate ties to the
lexicographically smallest refSplit.
"""
import ctypes
import math
import os
from collections import OrderedDict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PLUS, MINUS = 0, 1

# tools/SplitAlignment.cpp:25-29
NUM_BREAK_PADDING = 10
MATCH, MISMATCH, GAP, MIN_ANCHOR = 2, -1, -2, 4


# ------------------------------------------------------------------------------------------
# C oracle binding (oracle/dsa_oracle.c)
# ------------------------------------------------------------------------------------------
class OraSplit(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("ref_first", "ref_second", "read_first", "read_second", "score", "score1", "score2")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libdsa_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libdsa_oracle.so missing: run `make -C oracle`")
        _lib = ctypes.CDLL(path)
        _lib.ora_task_align.restype = ctypes.c_long
        _lib.ora_task_align.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                        ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(OraSplit), ctypes.c_long]
        _lib.ora_min_score.restype = ctypes.c_int
        _lib.ora_min_score.argtypes = [ctypes.c_int]
        _lib.ora_row_maxima.restype = None
        _lib.ora_row_maxima.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_int)]
    return _lib


def task_align(read: bytes, ref1: bytes, ref2: bytes):
    """SplitAlignmentTask::Align (tools/SplitAlignment.cpp:371-400) -> list of
    (ref_first, ref_second, read_first, read_second, score)."""
    cap = 256
    while True:
        buf = (OraSplit * cap)()
        n = lib().ora_task_align(read, len(read), ref1, len(ref1), ref2, len(ref2), buf, cap)
        if n <= cap:
            return [(b.ref_first, b.ref_second, b.read_first, b.read_second, b.score) for b in buf[:n]]
        cap = int(n)


def row_maxima(ref: bytes, read: bytes):
    out = (ctypes.c_int * (len(read) + 1))()
    lib().ora_row_maxima(ref, len(ref), read, len(read), out)
    return np.array(out[:], dtype=np.int32)


def align_batch(ref_bytes: np.ndarray, fusions: np.ndarray, read_bytes: np.ndarray, pairs: np.ndarray):
    """ora_align_batch with the numpy structured dtypes of defuse_amd.dsa (same layout as the C ABI)."""
    from defuse_amd import dsa as _dsa  # dtypes only; no GPU needed
    L = lib()
    L.ora_align_batch.restype = ctypes.c_int
    cap = max(1024, 4 * len(pairs))
    while True:
        out = np.zeros(cap, dtype=_dsa.RECORD_DTYPE)
        n = ctypes.c_int64(0)
        rc = L.ora_align_batch(
            ctypes.c_void_p(ref_bytes.ctypes.data), ctypes.c_int64(ref_bytes.size),
            ctypes.c_void_p(fusions.ctypes.data), ctypes.c_int32(len(fusions)),
            ctypes.c_void_p(read_bytes.ctypes.data), ctypes.c_int64(read_bytes.size),
            ctypes.c_void_p(pairs.ctypes.data), ctypes.c_int64(len(pairs)),
            ctypes.c_void_p(out.ctypes.data), ctypes.c_int64(cap), ctypes.byref(n))
        if rc == 0:
            return out[:n.value].copy()
        cap = int(n.value)


# ------------------------------------------------------------------------------------------
# tools/Common.cpp:32-54
# ------------------------------------------------------------------------------------------
_RC = bytes.maketrans(b"ACTGactg", b"TGACtgac")


def reverse_complement(seq: bytes) -> bytes:
    return seq[::-1].translate(_RC)


def lexical_cast_int(s: str) -> int:
    """boost::lexical_cast<int>: optional sign, digits only, no whitespace, no trailing junk."""
    t = s[1:] if s[:1] in "+-" else s
    if not t or not t.isdigit() or not t.isascii():
        raise ValueError("bad_lexical_cast: %r" % s)
    return int(s)


# ------------------------------------------------------------------------------------------
# FASTA random access: tools/FastaIndex.cpp:23-61 + external/samtools-0.1.8/faidx.c:305-357
# ------------------------------------------------------------------------------------------
class FastaIndex:
    def __init__(self, path):
        self.seqs = {}
        name = None
        chunks = []
        with open(path, "rb") as f:
            for line in f:
                if line.startswith(b">"):
                    if name is not None:
                        self.seqs[name] = b"".join(chunks)
                    fields = line[1:].split()
                    name = fields[0].decode() if fields else ""
                    chunks = []
                else:
                    chunks.append(bytes(c for c in line if 33 <= c <= 126))  # isgraph
        if name is not None:
            self.seqs[name] = b"".join(chunks)

    def get(self, reference, strand, start, length):
        """Returns (sequence, start, length) with start/length mutated as FastaIndex::Get does
        through its int& parameters (tools/FastaIndex.h:24)."""
        if length < 0:
            return b"", start, length
        if start < 1:
            length -= 1 - start
            start = 1
        end = start + length - 1
        if reference not in self.seqs:
            raise SystemExit("Error: Unable to find sequence for " + reference)
        seq = self.seqs[reference]
        beg = start  # atoi of the text written by the reference; start >= 1 here
        if beg > 0:
            beg -= 1
        e = end
        if beg >= len(seq):
            beg = len(seq)
        # faidx.c:337 compares the int `end` with the unsigned 32-bit field val.len: a NEGATIVE end (a window that lies wholly
        # before the sequence start: "name:1--12") converts to a huge unsigned number and is clipped to the sequence END,
        # so the reference fetches the whole sequence from `beg` on (pinned by tests/test_faidx_ref.py)
        if e < 0 or e >= len(seq):
            e = len(seq)
        if beg > e:
            beg = e
        out = seq[beg:e] if beg >= 0 else b""
        length = len(out)
        if strand == MINUS:
            out = reverse_complement(out)
        return out, start, length


# ------------------------------------------------------------------------------------------
# tools/ExonRegions.cpp
# ------------------------------------------------------------------------------------------
class ExonRegions:
    BIN = 100000  # :19

    def __init__(self, path):
        self.chromosome, self.strand, self.exons, self.length = {}, {}, {}, {}
        self.transcript_gene, self.exons_str, self.region, self.lookup = {}, {PLUS: {}, MINUS: {}}, {}, {}
        with open(path) as f:
            for line in f:
                line = line.rstrip("\n")
                if not line:
                    continue
                fields = line.split("\t")
                if len(fields) < 6:
                    continue
                gene, transcript, chrom, strand = fields[:4]
                exons = []
                for k in range(5, len(fields), 2):
                    exons.append((lexical_cast_int(fields[k - 1]), lexical_cast_int(fields[k])))
                if strand not in "+-" or len(strand) != 1:
                    raise SystemExit("Error: Unable to intepret strand " + strand)
                s = PLUS if strand == "+" else MINUS
                self.chromosome[transcript] = chrom
                self.strand[transcript] = s
                self.exons[transcript] = exons
                self.length[transcript] = sum(e - b + 1 for b, e in exons)
                self.transcript_gene[transcript] = gene
                self.exons_str[PLUS][transcript] = list(exons)
                self.exons_str[MINUS][transcript] = [(-e, -b) for b, e in exons][::-1]  # TransformExons :114-124
                self.region[transcript] = (exons[0][0], exons[-1][1])
                for b in range(_cdiv(exons[0][0], self.BIN), _cdiv(exons[-1][1], self.BIN) + 1):
                    self.lookup.setdefault(chrom, {}).setdefault(b, []).append(transcript)

    def is_transcript(self, t):
        return t in self.transcript_gene

    def region_transcripts(self, chrom, region):  # :131-161
        if chrom not in self.lookup:
            raise SystemExit("Error: Data mismatch, invalid chromosome " + chrom)
        found = set()
        for b in range(_cdiv(region[0], self.BIN), _cdiv(region[1], self.BIN) + 1):
            for t in self.lookup[chrom].get(b, []):
                r = self.region[t]
                if not (r[1] < region[0] or r[0] > region[1]):
                    found.add(t)
        return sorted(found)

    def remap_transcript_to_genome(self, transcript, strand, position):  # :258-302
        exons = self.exons[transcript]
        tlen, tstrand = self.length[transcript], self.strand[transcript]
        remap_strand = PLUS if tstrand == strand else MINUS
        if tstrand == MINUS:
            position = tlen - position + 1
        off = 0
        for b, e in exons:
            n = e - b + 1
            if position <= off + n:
                return self.chromosome[transcript], remap_strand, position - (off + 1) + b
            off += n
        return self.chromosome[transcript], remap_strand, position - tlen + exons[-1][1]

    def remap_through_transcript(self, transcript, position, strand, ext_min, ext_max):  # :421-482
        exons = self.exons_str[strand][transcript]
        tlen, tstrand = self.length[transcript], self.strand[transcript]
        remap_strand = PLUS if strand == tstrand else MINUS
        sp = position if strand == PLUS else -position
        if sp > exons[-1][1]:
            return None
        off = 0
        start = end = None
        for b, e in exons:
            n = e - b + 1
            if sp <= e:
                rs = sp - b + ext_min + 1
                re_ = sp - b + ext_max + 1
                if re_ < 1:
                    return None
                start = max(1, rs) + off
                end = max(1, re_) + off
                break
            off += n
        if end < 1 or start > tlen:
            return None
        if strand != tstrand:
            start, end = tlen - end + 1, tlen - start + 1
        return remap_strand, start, end


def _cdiv(a, b):
    """C++ int division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


# ------------------------------------------------------------------------------------------
# tools/Parsers.cpp:211-264
# ------------------------------------------------------------------------------------------
def read_align_region_pairs(path):
    pairs = {}
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            fields = line.split("\t")
            if len(fields) < 5:
                continue
            try:                                        # the catch clause of :254-258: message on stdout, exit(1)
                pid, pend = lexical_cast_int(fields[0]), lexical_cast_int(fields[1])
                assert pend in (0, 1)
                strand = PLUS if fields[3] == "+" else MINUS
                if fields[3] not in ("+", "-"):
                    raise SystemExit("Error: Unable to intepret strand " + fields[3])
                loc = dict(refName=fields[2], strand=strand, start=lexical_cast_int(fields[4]),
                           end=lexical_cast_int(fields[5]))
            except ValueError:
                raise SystemExit("Failed to interpret region:\n" + line)
            pairs.setdefault(pid, [None, None])[pend] = loc
    return OrderedDict(sorted(pairs.items()))  # std::map<int,...> iterates ascending


# ------------------------------------------------------------------------------------------
# tools/SplitAlignment.cpp:31-175, :637-655
# ------------------------------------------------------------------------------------------
class Task:
    pass


def calculate_break_region(min_read, max_read, max_frag, astart, aend, strand):
    region_len = aend - astart + 1
    push = min(max_read, int(0.5 * region_len))
    break_len = max_frag - region_len - min_read + 2 * push
    break_start = aend - push + 1 if strand == PLUS else astart + push - 1
    return break_start, break_len


def make_task(fid, align_pair, fasta, exons, ufrag, sfrag, min_read, max_read):
    t = Task()
    t.fusion_id = fid
    min_frag = int(ufrag - 3 * sfrag)
    max_frag = int(ufrag + 3 * sfrag)
    t.ref_name, t.strand = [None, None], [None, None]
    t.seq_start, t.seq_len, t.seq_strand = [0, 0], [0, 0], [0, 0]
    t.seq, t.remainder, t.mate_regions = [b"", b""], [b"", b""], [[], []]
    for ce in (0, 1):
        loc = align_pair[ce]
        name, strand, astart, aend = loc["refName"], loc["strand"], loc["start"], loc["end"]
        t.ref_name[ce], t.strand[ce] = name, strand
        ref_strand = strand if ce == 0 else 1 - strand
        bstart, blen = calculate_break_region(min_read, max_read, max_frag, astart, aend, strand)
        t.seq_strand[ce] = ref_strand
        if strand == PLUS:
            s0, l0 = bstart - max_read, blen + max_read
        else:
            s0, l0 = bstart - blen + 1, blen + max_read
        t.seq[ce], t.seq_start[ce], t.seq_len[ce] = fasta.get(name, ref_strand, s0, l0)
        if strand == PLUS:
            if astart < t.seq_start[ce]:
                t.remainder[ce] = fasta.get(name, ref_strand, astart, t.seq_start[ce] - 1 - astart + 1)[0]
        else:
            if aend > t.seq_start[ce] + t.seq_len[ce] - 1:
                rs = t.seq_start[ce] + t.seq_len[ce]
                t.remainder[ce] = fasta.get(name, ref_strand, rs, aend - rs + 1)[0]
        parts = name.split("|")
        if len(parts) >= 2 and exons.is_transcript(parts[1]):
            chrom, gstrand, gbstart = exons.remap_transcript_to_genome(parts[1], strand, bstart)
        else:
            chrom, gstrand, gbstart = name, strand, bstart
        mate_min = min_frag - blen - max_read + 1
        mate_max = max_frag - min_read
        if gstrand == PLUS:
            mr = (gbstart - mate_max, gbstart - mate_min)
        else:
            mr = (gbstart + mate_min, gbstart + mate_max)
        t.mate_regions[ce].append(dict(refName=chrom, strand=gstrand, start=mr[0], end=mr[1]))
        for tr in exons.region_transcripts(chrom, mr):
            r = exons.remap_through_transcript(tr, gbstart, 1 - gstrand, mate_min, mate_max)
            if r is not None:
                rstrand, ms, me = r
                t.mate_regions[ce].append(dict(refName=exons.transcript_gene[tr] + "|" + tr,
                                               strand=1 - rstrand, start=ms, end=me))
    return t


def create_tasks(fasta_path, exons_path, ufrag, sfrag, min_read, max_read, regions):
    fasta = FastaIndex(fasta_path)
    exons = ExonRegions(exons_path)
    return OrderedDict((fid, make_task(fid, pair, fasta, exons, ufrag, sfrag, min_read, max_read))
                       for fid, pair in regions.items())


# ------------------------------------------------------------------------------------------
# tools/SplitAlignment.cpp:177-303
# ------------------------------------------------------------------------------------------
def cluster_id(fid, cend):
    """ClusterID.id: clusterID + (clusterEnd<<31) as signed int (tools/Common.h:206-218)."""
    v = (fid & 0x7FFFFFFF) | (cend << 31)
    return v - (1 << 32) if v >= (1 << 31) else v


def read_id(frag, rend):
    v = (frag & 0x7FFFFFFF) | (rend << 31)
    return v - (1 << 32) if v >= (1 << 31) else v


class BinnedLocations:
    def __init__(self, spacing):
        self.spacing, self.ids, self.regions = spacing, [], []
        self.binned = [{}, {}]

    def add(self, cid, loc):
        idx = len(self.ids)
        self.ids.append(cid)
        self.regions.append((loc["start"], loc["end"]))
        for b in range(_cdiv(loc["start"], self.spacing), _cdiv(loc["end"], self.spacing) + 1):
            self.binned[loc["strand"]].setdefault(loc["refName"], {}).setdefault(b, []).append(idx)

    def overlapping(self, ref, strand, start, end):
        ids = set()
        byref = self.binned[strand].get(ref)
        if byref is None:
            return ids
        for b in range(_cdiv(start, self.spacing), _cdiv(end, self.spacing) + 1):
            for idx in byref.get(b, []):
                rs, re_ = self.regions[idx]
                if rs <= end and re_ >= start:
                    ids.add(self.ids[idx])
        return ids


def read_fastq(path, reads):
    """FastqReadStream::GetNextRead (tools/ReadStream.cpp:57-104) + AddReads (SplitAlignment.cpp:253-264)."""
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    for k in range(0, len(lines) - 3, 4):
        hdr = lines[k]
        if not hdr.startswith(b"@"):
            break
        slash = hdr.find(b"/")
        end_ch = hdr[slash + 1:slash + 2]
        if end_ch not in (b"1", b"2"):
            break
        frag = lexical_cast_int(hdr[1:slash].decode())
        reads[read_id(frag, 0 if end_ch == b"1" else 1)] = lines[k + 1]


def sam_alignments(path):
    """SamAlignmentStream::GetNextAlignment (tools/AlignmentStream.cpp:39-130)."""
    with open(path) as f:
        for n, line in enumerate(f, 1):
            line = line.rstrip("\n")
            if not line:
                raise SystemExit("Error: Empty alignment line %d" % n)
            if line[0] == "@":
                continue
            fields = line.split("\t")
            if len(fields) < 10:
                raise SystemExit("Error: Format error for alignment line %d" % n)
            qname, flag, rname, pos, seq = fields[0], lexical_cast_int(fields[1]), fields[2], \
                lexical_cast_int(fields[3]), fields[9]
            if rname == "*":
                continue
            strand = PLUS if (flag & 0x10) == 0 else MINUS
            q = qname.split("/")
            if len(q) == 2:
                if q[1] not in ("1", "2"):
                    raise SystemExit("Error: Unable to interpret qname for alignment line %d" % n)
                frag, rend = q[0], (0 if q[1] == "1" else 1)
            else:
                frag, rend = qname, (0 if flag & 0x40 else 1)
            yield frag, rend, rname, strand, pos, pos + len(seq) - 1


def enumerate_candidates(tasks, reads, sam_path):
    """The candidate loop of SplitReadRealigner::DoAlignment (tools/SplitAlignment.cpp:266-303)
    without the DP: yields (task, frag, read_end, revcomp, oriented read bytes) in canonical order."""
    binned = BinnedLocations(2000)
    for t in tasks.values():
        for ce in (0, 1):
            for loc in t.mate_regions[ce]:
                binned.add(cluster_id(t.fusion_id, ce), loc)
    seen = {}
    for frag, rend, rname, strand, start, end in sam_alignments(sam_path):
        for cid in sorted(binned.overlapping(rname, strand, start, end)):
            cend = 1 if cid < 0 else 0
            fid = cid & 0x7FFFFFFF
            frag_i = lexical_cast_int(frag)
            read_end = 1 if rend == 0 else 0
            revcomp = 1 if cend == 0 else 0
            seq = reads.get(read_id(frag_i, read_end), b"")
            if revcomp:
                seq = reverse_complement(seq)
            key = (read_id(frag_i, read_end), revcomp)
            s = seen.setdefault(fid, set())
            if key in s:
                continue
            s.add(key)
            yield tasks[fid], frag_i, read_end, revcomp, seq


def dosplitalign(fasta, exons, ufrag, sfrag, min_read, max_read, regions, improper, seq1, seq2):
    """tools/dosplitalign.cpp:25-111: returns the output file text."""
    tasks = create_tasks(fasta, exons, ufrag, sfrag, min_read, max_read, read_align_region_pairs(regions))
    reads = {}
    read_fastq(seq1, reads)
    read_fastq(seq2, reads)
    out = []
    for t, frag, rend, revcomp, seq in enumerate_candidates(tasks, reads, improper):
        for (rf, rs, qf, qs, score) in task_align(seq, t.seq[0], t.seq[1]):
            out.append("%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t\n" % (t.fusion_id, frag, rend, revcomp, rf, rs, qf, qs, score))
    return "".join(out)


# ------------------------------------------------------------------------------------------
# tools/SplitAlignment.cpp:484-635 + tools/evalsplitalign.cpp:96-114
# ------------------------------------------------------------------------------------------
def _fmt_double(x):
    """ostream << double with default precision 6 (%g)."""
    return "%g" % x


def evaluate(task, alignments):
    """alignments: list of tuples (fusion, frag, readEnd, revComp, rf, rs, qf, qs, score)."""
    split_score = {}
    for a in alignments:
        split_score[(a[4], a[5])] = split_score.get((a[4], a[5]), 0) + a[8]
    best, max_score = None, -1
    for split in sorted(split_score):  # canonical: ascending key order
        if split_score[split] > max_score:
            best, max_score = split, split_score[split]
    kept = [a for a in alignments if (a[4], a[5]) == best]
    seq = task.remainder[0] + task.seq[0][:best[0]] + b"|" + task.seq[1][best[1] + 1:] + task.remainder[1]
    if task.seq_strand[0] == PLUS:
        bp0 = task.seq_start[0] + best[0] - 1
    else:
        bp0 = task.seq_start[0] + task.seq_len[0] - best[0]
    if task.seq_strand[1] == PLUS:
        bp1 = task.seq_start[1] + best[1] + 1
    else:
        bp1 = task.seq_start[1] + task.seq_len[1] - best[1] - 2
    pos_sum = min_sum = 0.0
    for a in kept:
        left, right = a[6], a[7]
        pos_range = float(left + right - 2 * MIN_ANCHOR)
        pos_value = float(max(0, left - MIN_ANCHOR))
        min_range = math.floor(0.5 * float(left + right - 2 * MIN_ANCHOR))
        min_value = float(max(0, min(left - MIN_ANCHOR, right - MIN_ANCHOR)))
        pos_sum += _div(pos_value, pos_range)
        min_sum += _div(min_value, min_range)
    n = len(kept)
    return dict(seq=seq, break_pos=(bp0, bp1), count=n, pos_avg=pos_sum / n, min_avg=min_sum / n, kept=kept)


def _div(a, b):
    if b == 0.0:
        return float("nan") if a == 0.0 else math.copysign(float("inf"), a)
    return a / b


def evalsplitalign(fasta, exons, ufrag, sfrag, min_read, max_read, regions, align_path):
    """Returns (seq_text, break_text, predalign_text)."""
    tasks = create_tasks(fasta, exons, ufrag, sfrag, min_read, max_read, read_align_region_pairs(regions))
    rows = []
    with open(align_path) as f:
        for line in f:
            fields = line.rstrip("\n").split("\t")
            if len(fields) < 7:
                raise SystemExit("Error: Format error for candidate reads line:\n" + line)
            rows.append(tuple(lexical_cast_int(x) for x in fields[:9]))
    seq_out, brk_out, pred_out = [], [], []
    k = 0
    while k < len(rows):
        e = k
        while e < len(rows) and rows[e][0] == rows[k][0]:
            e += 1
        group = rows[k:e]
        k = e
        t = tasks[group[0][0]]
        p = evaluate(t, group)
        seq_out.append("%d\t%s\t0\t%d\t%s\t%s\n" % (t.fusion_id, p["seq"].decode(), p["count"],
                                                  _fmt_double(p["pos_avg"]), _fmt_double(p["min_avg"])))
        for ce in (0, 1):
            brk_out.append("%d\t%d\t%s\t%s\t%d\n" % (t.fusion_id, ce, t.ref_name[ce],
                                                     "+" if t.strand[ce] == PLUS else "-", p["break_pos"][ce]))
        for a in p["kept"]:
            pred_out.append("%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t\n" % a)
    return "".join(seq_out), "".join(brk_out), "".join(pred_out)
