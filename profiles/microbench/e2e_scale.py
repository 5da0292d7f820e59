"""BASELINE.json configs[4] on one GPU: synthetic 2x150 bp fusion-derived fragments (mu = 450, sigma = 45) through
clustermatepairs -> merge_clusters -> setcover -> remove_duplicates -> get_align_regions -> dosplitalign (one process per
chunk of a million fragments, as scripts/defuse_run.pl:518-523 runs it, reads_per_job = 1000000 in scripts/config.txt:112) ->
sort -n -k 1 (per chunk, then sort -m) -> evalsplitalign, every stage timed by itself: wall seconds, the GPU seconds the tools
report (DEFUSE_TIMING), bytes in and out.  One JSON document on stdout / --json.

    python profiles/microbench/e2e_scale.py --fragments 20000000 --out /tmp/e2e --json gpurun_out/e2e.json
    python profiles/microbench/e2e_scale.py --fragments 4000 --check        # tools against the oracle chain, timed as the CPU baseline

The generator (numpy, fixed-width text lines so that whole files are written as arrays): a random genome of 24 chromosomes,
fusions with 20-60 fragments each, fragments placed over the junction as tests/e2e_case.py places them — both reads clear of
the junction: a spanning pair for clustermatepairs (and two records of improper.sam); one read across it: a candidate of
dosplitalign (its mate is a record of improper.sam, the crossing read comes from the FASTQ chunk).  Fragments are shuffled:
every chunk holds a couple of reads of every fusion, which is what a chunk of a real run looks like to dosplitalign (few
candidates per fusion and chunk) — the opposite corner from bench.py's 100 reads per fusion."""
import argparse
import shutil
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "bin")
RL, UFRAG, SFRAG = 150, 450.0, 45.0
WIN = 700                                  # bases kept on either side of a junction
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
COMP[list(b"ACGT")] = list(b"TGCA")


def digits(v, width):
    """(n, width) uint8: zero-padded decimal text of v (the tools read integers with leading zeros as integers)."""
    v = np.asarray(v, dtype=np.int64)
    out = np.empty((len(v), width), dtype=np.uint8)
    for k in range(width):
        out[:, width - 1 - k] = 48 + (v // 10 ** k) % 10
    return out


def text_rows(n, parts):
    """Rows of fixed-width text from constant byte strings and (n, w) uint8 blocks."""
    width = sum(len(p) if isinstance(p, bytes) else p.shape[1] for p in parts)
    out = np.empty((n, width), dtype=np.uint8)
    at = 0
    for p in parts:
        if isinstance(p, bytes):
            out[:, at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
            at += len(p)
        else:
            out[:, at:at + p.shape[1]] = p
            at += p.shape[1]
    return out


def generate(out, n_fragments, seed=7, n_chrom=24, chrom_len=60_000_000, chunk=1_000_000, lo=100, hi=300):
    rng = np.random.default_rng(seed)
    os.makedirs(out, exist_ok=True)
    P = lambda n: os.path.join(out, n)
    t0 = time.time()
    chrom_len -= chrom_len % 60
    names = ["chr%02d" % (k + 1) for k in range(n_chrom)]
    genome = []
    with open(P("ref.fa"), "wb") as fa, open(P("ref.fa.fai"), "w") as fai, open(P("exons.txt"), "w") as ex:
        off = 0
        for k, name in enumerate(names):
            s = ACGT[rng.integers(0, 4, size=chrom_len, dtype=np.uint8)]
            genome.append(s)
            head = (">%s\n" % name).encode()
            fa.write(head)
            off += len(head)
            fai.write("%s\t%d\t%d\t60\t61\n" % (name, chrom_len, off))
            lines = np.empty((chrom_len // 60, 61), dtype=np.uint8)
            lines[:, :60] = s.reshape(-1, 60)
            lines[:, 60] = 10
            lines.tofile(fa)
            off += lines.size
            ex.write("ENSG%02d\tENST%02d\t%s\t+\t1\t%d\t\n" % (k + 1, k + 1, name, chrom_len))
    # fusions and the 2 x WIN bases around every junction, in transcript orientation (left part ... junction ... right part)
    sup = []
    total = 0
    while total < n_fragments:
        s = int(rng.integers(lo, hi + 1))
        sup.append(min(s, n_fragments - total))
        total += sup[-1]
    sup = np.array(sup, dtype=np.int64)
    F = len(sup)
    ca, cb = rng.integers(0, n_chrom, size=F), rng.integers(0, n_chrom, size=F)
    sa, sb = rng.integers(0, 2, size=F), rng.integers(0, 2, size=F)            # 0 '+', 1 '-'
    ba, bb = rng.integers(5000, chrom_len - 5000, size=F), rng.integers(5000, chrom_len - 5000, size=F)
    win = np.empty((F, 2 * WIN), dtype=np.uint8)
    k = np.arange(WIN)
    for c in range(n_chrom):
        g = genome[c]
        m = (ca == c) & (sa == 0)                       # left = chrom[:brk]: its last WIN bases
        win[m, :WIN] = g[(ba[m] - WIN)[:, None] + k]
        m = (ca == c) & (sa == 1)                       # left = rc(chrom[brk-1:]): its last WIN bases = rc(chrom[brk-1 : brk-1+WIN])
        win[m, :WIN] = COMP[g[(ba[m] - 1 + WIN - 1)[:, None] - k]]
        m = (cb == c) & (sb == 1)                       # right = chrom[brk-1:]: its first WIN bases
        win[m, WIN:] = g[(bb[m] - 1)[:, None] + k]
        m = (cb == c) & (sb == 0)                       # right = rc(chrom[:brk]): its first WIN bases = rc(chrom[brk-WIN : brk])
        win[m, WIN:] = COMP[g[(bb[m] - 1)[:, None] - k]]
    del genome
    # fragments, shuffled
    fus = np.repeat(np.arange(F), sup)[rng.permutation(n_fragments)]
    flen = np.clip(rng.normal(UFRAG, SFRAG, size=n_fragments).astype(np.int64), 2 * RL + 10, WIN - 20)
    p_rel = -flen + 12 + (rng.random(n_fragments) * (flen - 24)).astype(np.int64)          # start of the fragment relative to the junction
    one_left = p_rel + RL <= 0                                # read 1 wholly in the left part
    two_right = p_rel + flen - RL >= 0                        # read 2 wholly in the right part
    # positions (1-based start) of the whole-read alignments
    q = p_rel                                                 # read 1 covers fused[J+q : J+q+RL)
    a_start = np.where(sa[fus] == 0, ba[fus] + q + 1, ba[fus] - q - RL)
    r = p_rel + flen - RL                                     # read 2 covers fused[J+r : J+r+RL), r >= 0 when whole
    b_start = np.where(sb[fus] == 1, bb[fus] + r, bb[fus] - r - RL + 1)
    frag_id = np.arange(n_fragments, dtype=np.int64)
    chr_digits = lambda c: digits(c + 1, 2)
    strand_a = np.where(sa[fus] == 0, ord("+"), ord("-")).astype(np.uint8)[:, None]
    strand_b = np.where(sb[fus] == 1, ord("-"), ord("+")).astype(np.uint8)[:, None]      # read 2 is the reverse complement of the tail
    # spanning alignments (compact format of divide_sam_chr_pairs.pl: fragment, end - 1, reference, strand, start, end)
    span = np.nonzero(one_left & two_right)[0]
    rows = np.empty((2 * len(span), 0), dtype=np.uint8)
    r1 = text_rows(len(span), [digits(frag_id[span], 9), b"\t0\tchr", chr_digits(ca[fus[span]]), b"\t", strand_a[span], b"\t", digits(a_start[span], 9),
                               b"\t", digits(a_start[span] + RL - 1, 9), b"\n"])
    r2 = text_rows(len(span), [digits(frag_id[span], 9), b"\t1\tchr", chr_digits(cb[fus[span]]), b"\t", strand_b[span], b"\t", digits(b_start[span], 9),
                               b"\t", digits(b_start[span] + RL - 1, 9), b"\n"])
    rows = np.empty((2 * len(span), r1.shape[1]), dtype=np.uint8)
    rows[0::2] = r1
    rows[1::2] = r2
    rows.tofile(P("spanning.txt"))
    del rows, r1, r2
    # FASTQ chunks and improper.sam chunks
    n_chunks = -(-n_fragments // chunk)
    seqA, qual = np.full((1, RL), ord("A"), dtype=np.uint8), np.full((1, RL), ord("I"), dtype=np.uint8)
    kk = np.arange(RL)
    for c in range(n_chunks):
        s = slice(c * chunk, min(n_fragments, (c + 1) * chunk))
        n = s.stop - s.start
        f = fus[s]
        read1 = win[f[:, None], (WIN + p_rel[s])[:, None] + kk]
        read2 = COMP[win[f[:, None], (WIN + p_rel[s] + flen[s] - 1)[:, None] - kk]]
        for rd in (read1, read2):                             # 1 % substitutions
            m = rng.random(rd.shape) < 0.01
            cur = np.searchsorted(ACGT, rd[m])
            rd[m] = ACGT[(cur + rng.integers(1, 4, size=cur.size)) % 4]
        for e, rd in ((1, read1), (2, read2)):
            text_rows(n, [b"@", digits(frag_id[s], 9), b"/%d\n" % e, rd, b"\n+\n", np.broadcast_to(qual, (n, RL)), b"\n"]).tofile(P("reads.%d.%d.fastq" % (c, e)))
        # improper.sam: every end that aligns as a whole, in fragment order, end 1 before end 2
        l, rgt = one_left[s], two_right[s]
        flag_a = np.where(sa[f] == 0, 0, 16)
        flag_b = np.where(sb[f] == 1, 16, 0)
        sam1 = text_rows(n, [digits(frag_id[s], 9), b"/1\t", digits(flag_a, 2), b"\tchr", chr_digits(ca[f]), b"\t", digits(a_start[s], 9),
                             b"\t255\t150M\t*\t0\t0\t", np.broadcast_to(seqA, (n, RL)), b"\t", np.broadcast_to(qual, (n, RL)), b"\n"])
        sam2 = text_rows(n, [digits(frag_id[s], 9), b"/2\t", digits(flag_b, 2), b"\tchr", chr_digits(cb[f]), b"\t", digits(b_start[s], 9),
                             b"\t255\t150M\t*\t0\t0\t", np.broadcast_to(seqA, (n, RL)), b"\t", np.broadcast_to(qual, (n, RL)), b"\n"])
        both = np.empty((2 * n, sam1.shape[1]), dtype=np.uint8)
        both[0::2] = sam1
        both[1::2] = sam2
        keep = np.empty(2 * n, dtype=bool)
        keep[0::2] = l
        keep[1::2] = rgt
        with open(P("improper.%d.sam" % c), "wb") as fh:
            fh.write(b"@HD\tVN:1.0\tSO:unsorted\n")
            both[keep].tofile(fh)
    planted = dict(chr_a=ca, strand_a=sa, break_a=ba, chr_b=cb, strand_b=sb, break_b=bb, support=sup)
    info = {"fragments": int(n_fragments), "fusions": int(F), "chunks": int(n_chunks), "spanning_fragments": int(len(span)),
            "split_candidates": int(np.count_nonzero(one_left ^ two_right)), "generate_s": round(time.time() - t0, 1),
            "genome_bases": int(n_chrom * chrom_len)}
    return info, planted, names


class Stages:
    def __init__(self):
        self.rows = []

    def run(self, name, cmd, inputs, outputs, stdin=None, stdout=None, env=None, shell=False):
        e = dict(os.environ, DEFUSE_TIMING="1", **(env or {}))
        t0 = time.time()
        fin = open(stdin, "rb") if stdin else None
        fout = open(stdout, "wb") if stdout else subprocess.PIPE
        p = subprocess.run(cmd, stdin=fin, stdout=fout, stderr=subprocess.PIPE, env=e, shell=shell)
        dt = time.time() - t0
        if fin:
            fin.close()
        if stdout:
            fout.close()
        err = p.stderr.decode(errors="replace")
        if p.returncode != 0:
            raise SystemExit("%s failed (%d): %s" % (name, p.returncode, err[-2000:]))
        gpu = 0.0
        for m in re.finditer(r"kernel ([0-9.e+-]+) ms", err):
            gpu += float(m.group(1)) * 1e-3
        for m in re.finditer(r"kernels \+ sorts ([0-9.e+-]+) ms", err):
            gpu += float(m.group(1)) * 1e-3
        for m in re.finditer(r"of which GPU calls ([0-9.e+-]+) s", err):
            gpu += float(m.group(1))
        for m in re.finditer(r"build ([0-9.e+-]+) ms, components ([0-9.e+-]+) ms, greedy ([0-9.e+-]+) ms", err):
            gpu += (float(m.group(1)) + float(m.group(2)) + float(m.group(3))) * 1e-3
        row = {"stage": name, "wall_s": round(dt, 3), "gpu_s": round(gpu, 4),
               "bytes_in": int(sum(os.path.getsize(f) for f in inputs if os.path.exists(f))),
               "bytes_out": int(sum(os.path.getsize(f) for f in outputs if os.path.exists(f)))}
        self.rows.append(row)
        return row, err


def pipeline(out, n_chunks, stages, parallel=1, threads=16):
    """The chain on the files generate() wrote; returns the paths of the final files."""
    P = lambda n: os.path.join(out, n)
    T = lambda n: os.path.join(BIN, n)
    cm = ["-u", str(UFRAG), "-s", str(SFRAG)]
    th = {"DEFUSE_THREADS": str(threads)}
    stages.run("clustermatepairs", [T("clustermatepairs"), "-m", "5", "-p", "0.95"] + cm + ["-a", P("spanning.txt"), "-c", P("clusters.0")],
               [P("spanning.txt")], [P("clusters.0")], env=th)
    stages.run("merge_clusters", [T("defuse_glue"), "merge_clusters", P("clusters.0")], [P("clusters.0")], [P("clusters.all")], stdout=P("clusters.all"))
    stages.run("setcover", [T("setcover"), "-m", "5", "-c", P("clusters.all"), "-o", P("clusters.sc.all")], [P("clusters.all")], [P("clusters.sc.all")], env=th)
    stages.run("remove_duplicates", [T("defuse_glue"), "remove_duplicates", "5"], [P("clusters.sc.all")], [P("clusters.sc")], stdin=P("clusters.sc.all"),
               stdout=P("clusters.sc"))
    stages.run("get_align_regions", [T("defuse_glue"), "get_align_regions"], [P("clusters.sc")], [P("clusters.sc.regions")], stdin=P("clusters.sc"),
               stdout=P("clusters.sc.regions"))
    common = ["-f", P("ref.fa"), "-e", P("exons.txt")] + cm + ["-n", str(RL), "-x", str(RL), "-r", P("clusters.sc.regions")]
    t0 = time.time()
    gpu = 0.0
    bi = bo = 0
    detail = []

    def one(c):
        st = Stages()
        row, err = st.run("dosplitalign.%d" % c, [T("dosplitalign")] + common + ["-i", P("improper.%d.sam" % c), "-1", P("reads.%d.1.fastq" % c),
                                                                                 "-2", P("reads.%d.2.fastq" % c), "-a", P("split.%d" % c)],
                          [P("improper.%d.sam" % c), P("reads.%d.1.fastq" % c), P("reads.%d.2.fastq" % c), P("clusters.sc.regions")], [P("split.%d" % c)])
        return row, err
    if parallel > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=parallel) as ex:
            res = list(ex.map(one, range(n_chunks)))
    else:
        res = [one(c) for c in range(n_chunks)]
    for row, err in res:
        gpu += row["gpu_s"]
        bi += row["bytes_in"]
        bo += row["bytes_out"]
        detail.append(row["wall_s"])
    last_err = res[-1][1]
    stages.rows.append({"stage": "dosplitalign x %d chunks%s" % (n_chunks, (", %d at a time" % parallel) if parallel > 1 else ""), "wall_s": round(time.time() - t0, 3),
                        "gpu_s": round(gpu, 4), "bytes_in": bi, "bytes_out": bo, "per_chunk_wall_s": detail,
                        "last_chunk_timing": [l for l in last_err.splitlines() if l.startswith("[dosplitalign]")]})
    t0 = time.time()
    for c in range(n_chunks):
        subprocess.check_call("LC_ALL=C sort -n -k 1 %s > %s" % (P("split.%d" % c), P("split.%d.sorted" % c)), shell=True)
    subprocess.check_call("LC_ALL=C sort -m -n -k 1 %s > %s" % (" ".join(P("split.%d.sorted" % c) for c in range(n_chunks)), P("splitreads.alignments")), shell=True)
    stages.rows.append({"stage": "sort -n -k 1 per chunk + sort -m", "wall_s": round(time.time() - t0, 3), "gpu_s": 0.0,
                        "bytes_in": int(sum(os.path.getsize(P("split.%d" % c)) for c in range(n_chunks))), "bytes_out": os.path.getsize(P("splitreads.alignments"))})
    stages.run("evalsplitalign", [T("evalsplitalign")] + common + ["-a", P("splitreads.alignments"), "-q", P("splitreads.seq"), "-b", P("splitreads.break"),
                                                                 "-p", P("splitreads.predalign")],
               [P("splitreads.alignments"), P("clusters.sc.regions")], [P("splitreads.seq"), P("splitreads.break"), P("splitreads.predalign")])
    return P("splitreads.break")


def fused_legs(out, n_chunks):
    """The back half of the chain (get_align_regions -> dosplitalign -> sort -> evalsplitalign) in dosplitalign's fused mode
    (DEFUSE_FUSED=1), after pipeline() has run: (1) per chunk with --sorted (regions file as in the chain), so that GNU sort only merges;
    (2) one process over all reads, from the clusters to the three final files.  Both must give pipeline()'s final files."""
    P = lambda n: os.path.join(out, n)
    T = lambda n: os.path.join(BIN, n)
    cm = ["-u", str(UFRAG), "-s", str(SFRAG)]
    base = ["-f", P("ref.fa"), "-e", P("exons.txt")] + cm + ["-n", str(RL), "-x", str(RL)]
    env = dict(os.environ, DEFUSE_FUSED="1", DEFUSE_TIMING="1")
    final = [P("splitreads.seq"), P("splitreads.break"), P("splitreads.predalign")]
    want = [open(f, "rb").read() for f in final]
    legs = {}
    # (1) per chunk, sorted in the process
    rows = []
    t0 = time.time()
    per = []
    for c in range(n_chunks):
        t1 = time.time()
        r = subprocess.run([T("dosplitalign")] + base + ["-r", P("clusters.sc.regions"), "--sorted", "-i", P("improper.%d.sam" % c),
                                                         "-1", P("reads.%d.1.fastq" % c), "-2", P("reads.%d.2.fastq" % c), "-a", P("fsplit.%d" % c)],
                           env=env, capture_output=True, text=True)
        if r.returncode:
            raise SystemExit("fused dosplitalign, chunk %d: %s" % (c, r.stderr[-2000:]))
        per.append(round(time.time() - t1, 3))
    rows.append({"stage": "dosplitalign --sorted x %d chunks (alignments in sort order)" % n_chunks, "wall_s": round(time.time() - t0, 3),
                 "per_chunk_wall_s": per})
    t0 = time.time()
    subprocess.check_call("LC_ALL=C sort -m -n -k 1 %s > %s" % (" ".join(P("fsplit.%d" % c) for c in range(n_chunks)), P("f.alignments")), shell=True)
    rows.append({"stage": "sort -m", "wall_s": round(time.time() - t0, 3)})
    t0 = time.time()
    subprocess.check_call([T("evalsplitalign")] + base + ["-r", P("clusters.sc.regions"), "-a", P("f.alignments"), "-q", P("f.seq"), "-b", P("f.break"), "-p", P("f.predalign")])
    rows.append({"stage": "evalsplitalign", "wall_s": round(time.time() - t0, 3)})
    got = [open(P(n), "rb").read() for n in ("f.seq", "f.break", "f.predalign")]
    legs["per_chunk_sorted"] = {"stages": rows, "wall_s": round(sum(r["wall_s"] for r in rows), 3),
                                "replaces": "dosplitalign x chunks + sort + evalsplitalign of the chain above",
                                "same_alignments_file": open(P("f.alignments"), "rb").read() == open(P("splitreads.alignments"), "rb").read(),
                                "same_final_files": got == want}
    # (1b) the unmodified chain with DEFUSE_DSA_SORTED=1 in the environment: dosplitalign writes in sort order, GNU sort still runs
    rows = []
    t0 = time.time()
    env_sorted = dict(os.environ, DEFUSE_DSA_SORTED="1")
    for c in range(n_chunks):
        r = subprocess.run([T("dosplitalign")] + base + ["-r", P("clusters.sc.regions"), "-i", P("improper.%d.sam" % c), "-1", P("reads.%d.1.fastq" % c),
                                                         "-2", P("reads.%d.2.fastq" % c), "-a", P("ssplit.%d" % c)], env=env_sorted, capture_output=True, text=True)
        if r.returncode:
            raise SystemExit("dosplitalign with DEFUSE_DSA_SORTED, chunk %d: %s" % (c, r.stderr[-2000:]))
    rows.append({"stage": "dosplitalign x %d chunks, DEFUSE_DSA_SORTED=1" % n_chunks, "wall_s": round(time.time() - t0, 3)})
    t0 = time.time()
    for c in range(n_chunks):
        subprocess.check_call("LC_ALL=C sort -n -k 1 %s > %s" % (P("ssplit.%d" % c), P("ssplit.%d.sorted" % c)), shell=True)
    subprocess.check_call("LC_ALL=C sort -m -n -k 1 %s > %s" % (" ".join(P("ssplit.%d.sorted" % c) for c in range(n_chunks)), P("s.alignments")), shell=True)
    rows.append({"stage": "sort -n -k 1 per chunk + sort -m (input in order already)", "wall_s": round(time.time() - t0, 3)})
    legs["sorted_by_environment"] = {"stages": rows, "wall_s": round(sum(r["wall_s"] for r in rows), 3),
                                     "replaces": "dosplitalign x chunks + sort of the chain above (the pipeline unchanged, one variable in its environment)",
                                     "same_alignments_file": open(P("s.alignments"), "rb").read() == open(P("splitreads.alignments"), "rb").read()}
    for c in range(n_chunks):
        for n in ("ssplit.%d" % c, "ssplit.%d.sorted" % c):
            os.unlink(P(n))
    os.unlink(P("s.alignments"))
    # (2) one process: the chunks' inputs concatenated (the pipeline cuts them from such files), everything else in the tool
    t0 = time.time()
    for kind in ("improper.%d.sam", "reads.%d.1.fastq", "reads.%d.2.fastq"):
        with open(P("all." + kind.replace("%d.", "")), "wb") as fh:
            for c in range(n_chunks):
                with open(P(kind % c), "rb") as src:
                    if kind.endswith(".sam") and c > 0:
                        for line in src:                     # one header is enough
                            if not line.startswith(b"@"):
                                fh.write(line)
                                break
                    shutil.copyfileobj(src, fh, 16 << 20)
    prep = time.time() - t0
    t0 = time.time()
    r = subprocess.run([T("dosplitalign")] + base + ["-c", P("clusters.sc"), "-r", P("g.regions"), "--sorted", "-i", P("all.improper.sam"), "-1", P("all.reads.1.fastq"),
                                                     "-2", P("all.reads.2.fastq"), "-a", P("g.alignments"), "-q", P("g.seq"), "-b", P("g.break"), "-p", P("g.predalign")],
                       env=env, capture_output=True, text=True)
    if r.returncode:
        raise SystemExit("fused dosplitalign, one process: %s" % r.stderr[-2000:])
    wall = time.time() - t0
    got = [open(P(n), "rb").read() for n in ("g.seq", "g.break", "g.predalign")]
    legs["one_process"] = {"wall_s": round(wall, 3), "inputs_concatenated_s_not_counted": round(prep, 3),
                           "replaces": "get_align_regions + dosplitalign x chunks + sort + evalsplitalign of the chain above",
                           "same_alignments_file": open(P("g.alignments"), "rb").read() == open(P("splitreads.alignments"), "rb").read(),
                           "same_final_files": got == want, "timing": [l for l in r.stderr.splitlines() if l.startswith("[dosplitalign]")]}
    return legs


def recovered(break_file, planted, names):
    """How many of the planted junctions with enough support come back exactly (both ends) among the predicted breakpoints."""
    found = set()
    for line in open(break_file):
        f = line.rstrip("\n").split("\t")
        found.add((f[2], f[3], int(f[4])))
    ok = n = 0
    for k in range(len(planted["support"])):
        if planted["support"][k] < 20:            # (the last fusion of a data set may have fewer)
            continue
        n += 1
        a = (names[planted["chr_a"][k]], "+-"[planted["strand_a"][k]], int(planted["break_a"][k]))
        b = (names[planted["chr_b"][k]], "+-"[planted["strand_b"][k]], int(planted["break_b"][k]))
        ok += 1 if a in found and b in found else 0
    return ok, n


def oracle_chain(out, n_chunks):
    """The CPU restatements chained the same way (oracle/*.py, C restatements of the DP and the EM behind them); returns the
    three final texts and the seconds per stage."""
    from oracle import clustermatepairs_oracle as co, setcover_oracle as so, dosplitalign_oracle as do
    P = lambda n: os.path.join(out, n)
    t = {}
    t0 = time.time()
    txt, n = co.clustermatepairs(open(P("spanning.txt")).readlines(), UFRAG, SFRAG, 0.95, 5, em="c")
    t["clustermatepairs"] = time.time() - t0
    open(P("o.clusters.all"), "w").write(txt)
    t0 = time.time()
    sc_all = so.setcover(P("o.clusters.all"), 5)
    t["setcover"] = time.time() - t0
    glue = lambda step, *a, **kw: subprocess.run([os.path.join(BIN, "defuse_glue"), step] + list(a), capture_output=True, text=True, check=True, **kw).stdout
    sc = glue("remove_duplicates", "5", input=sc_all)
    regions = glue("get_align_regions", input=sc)
    open(P("o.regions"), "w").write(regions)
    oc = (P("ref.fa"), P("exons.txt"), UFRAG, SFRAG, RL, RL, P("o.regions"))
    t0 = time.time()
    al = "".join(do.dosplitalign(*oc, P("improper.%d.sam" % c), P("reads.%d.1.fastq" % c), P("reads.%d.2.fastq" % c)) for c in range(n_chunks))
    t["dosplitalign"] = time.time() - t0
    rows = sorted(al.splitlines(True), key=lambda l: (int(l.split("\t")[0]), l))
    open(P("o.sorted"), "w").write("".join(rows))
    t0 = time.time()
    res = do.evalsplitalign(*oc, P("o.sorted"))
    t["evalsplitalign"] = time.time() - t0
    return res, regions, t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fragments", type=int, default=20_000_000)
    ap.add_argument("--out", default="/tmp/e2e_scale")
    ap.add_argument("--json", default=None)
    ap.add_argument("--chrom-len", type=int, default=60_000_000)
    ap.add_argument("--chunk", type=int, default=1_000_000)
    ap.add_argument("--parallel", type=int, default=4, help="a second pass over the dosplitalign chunks with this many processes at a time")
    ap.add_argument("--support", type=int, nargs=2, default=[100, 300], metavar=("LO", "HI"),
                    help="fragments per fusion, uniform in [LO, HI]: 100 300 is the proportion of BASELINE configs[3] (200 reads per fusion); 20 60 makes five "
                         "times as many fusions with a couple of candidates each per chunk (the table tiers of the fill kernel cannot amortise a fusion's tables then)")
    ap.add_argument("--threads", type=int, default=16, help="DEFUSE_THREADS of clustermatepairs and setcover (dosplitalign: its own default)")
    ap.add_argument("--check", action="store_true", help="small sizes: compare the final files with the oracle chain and time it (the CPU baseline)")
    ap.add_argument("--generate-only", action="store_true")
    ap.add_argument("--no-fused", dest="fused", action="store_false", help="skip the two fused-mode legs (DEFUSE_FUSED=1, SURVEY 8(f)-2)")
    args = ap.parse_args()
    info, planted, names = generate(args.out, args.fragments, chrom_len=args.chrom_len, chunk=args.chunk, lo=args.support[0], hi=args.support[1])
    info["fragments_per_fusion"] = args.support
    res = {"what": "BASELINE configs[4] shape on one GPU: %d fragments 2x%d bp, mu %g sigma %g, %d fusions, dosplitalign per chunk of %d fragments"
                   % (args.fragments, RL, UFRAG, SFRAG, info["fusions"], args.chunk), "input": info}
    if args.generate_only:
        print(json.dumps(res))
        return
    st = Stages()
    t0 = time.time()
    brk = pipeline(args.out, info["chunks"], st, threads=args.threads)
    wall = time.time() - t0
    ok, n = recovered(brk, planted, names)
    res["stages"] = st.rows
    res["wall_s"] = round(wall, 3)
    res["stages_sum_s"] = round(sum(r["wall_s"] for r in st.rows), 3)
    res["stages_over_wall"] = round(res["stages_sum_s"] / wall, 4)
    res["gpu_busy_s"] = round(sum(r["gpu_s"] for r in st.rows), 3)
    res["fragments_per_s_end_to_end"] = round(args.fragments / wall)
    res["planted_junctions_recovered"] = {"exactly": ok, "of_those_with_20_or_more_fragments": n}
    res["break_lines"] = sum(1 for _ in open(brk))
    if args.parallel > 1 and info["chunks"] > 1:
        P = lambda nme: os.path.join(args.out, nme)
        T = lambda nme: os.path.join(BIN, nme)
        common = ["-f", P("ref.fa"), "-e", P("exons.txt"), "-u", str(UFRAG), "-s", str(SFRAG), "-n", str(RL), "-x", str(RL), "-r", P("clusters.sc.regions")]
        from concurrent.futures import ThreadPoolExecutor
        t1 = time.time()

        def one(c):
            return subprocess.run([T("dosplitalign")] + common + ["-i", P("improper.%d.sam" % c), "-1", P("reads.%d.1.fastq" % c), "-2", P("reads.%d.2.fastq" % c),
                                                                  "-a", P("psplit.%d" % c)], capture_output=True).returncode
        with ThreadPoolExecutor(max_workers=args.parallel) as ex:
            rcs = list(ex.map(one, range(info["chunks"])))
        same = all(open(P("psplit.%d" % c), "rb").read() == open(P("split.%d" % c), "rb").read() for c in range(info["chunks"]))
        res["dosplitalign_chunks_in_parallel"] = {"at_a_time": args.parallel, "wall_s": round(time.time() - t1, 3), "all_ok": all(r == 0 for r in rcs),
                                                  "same_files_as_one_at_a_time": same}
    if args.fused:
        res["fused"] = fused_legs(args.out, info["chunks"])
    if args.check:
        (seq, brk_txt, pred), regions, t = oracle_chain(args.out, info["chunks"])
        P = lambda nme: os.path.join(args.out, nme)
        same = {"regions": sorted(open(P("clusters.sc.regions")).read().splitlines()) == sorted(regions.splitlines()),
                "break": open(P("splitreads.break")).read() == brk_txt, "seq": open(P("splitreads.seq")).read() == seq,
                "predalign": open(P("splitreads.predalign")).read() == pred}
        res["cpu_baseline"] = {"kind": "port", "cores": 1, "sample": "the same %d fragments through the oracle chain (Python + the C restatements of the DP and the EM)" % args.fragments,
                               "stage_s": {k: round(v, 3) for k, v in t.items()}, "total_s": round(sum(t.values()), 3),
                               "fragments_per_s": round(args.fragments / max(sum(t.values()), 1e-9), 1), "tools_equal_the_oracle_chain": same}
        if not all(same.values()):
            print(json.dumps(res, indent=1))
            raise SystemExit("the tools' final files differ from the oracle chain's")
    text = json.dumps(res, indent=1)
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        open(args.json, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
