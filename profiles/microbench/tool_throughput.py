"""End-to-end wall time of the drop-in dosplitalign / evalsplitalign binaries on a synthetic case
(tests/pipeline_case.py): shows how much of a tool run is host text I/O and how much is the GPU.
Usage: python profiles/microbench/tool_throughput.py [n_fusions] [reads_per_fusion] [copies]
copies > 1: the case is also replicated that many times (own chromosomes, fusion ids and fragment ids per copy) and run as ONE
job, e.g. 10000 100 10 = ten million candidates."""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import pipeline_case


def timed(args, env=None, reps=3, label="dosplitalign"):
    best = None
    for rep in range(reps):
        t0 = time.time()
        p = subprocess.run(["bin/dosplitalign"] + args, capture_output=True, text=True, env=dict(os.environ, **(env or {})))
        dt = time.time() - t0
        print("%s run %d: rc %d, %.3f s wall" % (label, rep + 1, p.returncode, dt))
        if p.returncode != 0:
            print(p.stderr[-1500:])
        best = dt if best is None else min(best, dt)
    return best, p


def replicate(case, d, copies):
    """`copies` copies of the case in one set of files: copy k has chromosomes <name>_k, fusion ids + k * 10^6, fragments + k * F."""
    frags = 1 + max(int(subprocess.check_output("tail -4 %s | head -1" % case[k], shell=True).decode().split("/")[0][1:]) for k in ("seq1", "seq2"))
    big = dict(case)
    for key, name in (("fasta", "big.fa"), ("exons", "big.exons.txt"), ("regions", "big.regions.txt"), ("improper", "big.sam"), ("seq1", "big.1.fastq"), ("seq2", "big.2.fastq")):
        big[key] = os.path.join(d, name)
        if os.path.exists(big[key]):
            os.unlink(big[key])
    for k in range(copies):
        sh = lambda c: subprocess.check_call(c, shell=True)
        sh("sed -e 's/^>\\(chr[0-9]*\\)/>\\1_%d/' -e 's/^>\\([^| ]*\\)|\\([^ ]*\\)/>\\1_%d|\\2_%d/' %s >> %s" % (k, k, k, case["fasta"], big["fasta"]))
        sh("awk 'BEGIN{FS=OFS=\"\\t\"} NF>=6 {$1=$1\"_%d\"; $2=$2\"_%d\"; $3=$3\"_%d\"; print}' %s >> %s" % (k, k, k, case["exons"], big["exons"]))
        sh("awk 'BEGIN{FS=OFS=\"\\t\"} NF>=6 && $1+0==$1 {$1=$1+%d; $3=$3\"_%d\"; print}' %s >> %s" % (k * 1000000, k, case["regions"], big["regions"]))
        sh("awk 'BEGIN{FS=OFS=\"\\t\"} /^@/{if (%d==0) print; next} {split($1,a,\"/\"); $1=(a[1]+%d)\"/\"a[2]; if ($3!=\"*\") {n=split($3,b,\"|\"); $3=(n==2 ? b[1]\"_%d|\"b[2]\"_%d\" : $3\"_%d\")} print}' %s >> %s"
           % (k, k * frags, k, k, k, case["improper"], big["improper"]))
        for key in ("seq1", "seq2"):
            sh("awk 'NR%%4==1{split(substr($0,2),a,\"/\"); print \"@\"(a[1]+%d)\"/\"a[2]; next} {print}' %s >> %s" % (k * frags, case[key], big[key]))
    return big


def main():
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    copies = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    d = tempfile.mkdtemp(prefix="tooltp_")
    t0 = time.time()
    # chromosomes only and long ones: regions of different fusions rarely overlap, as in a real run
    case = pipeline_case.build(d, seed=9, n_fusions=nf, reads_per_fusion=rp, lq=76, chrom_len=max(5000, nf * 3000),
                               transcripts=False)
    print("case built in %.1f s: %d fusions x %d reads" % (time.time() - t0, nf, rp))
    out = os.path.join(d, "split.align.txt")
    args = pipeline_case.tool_args(case, out)
    t0 = time.time()
    p = subprocess.run(["bin/dosplitalign"] + args, capture_output=True, text=True)
    dt = time.time() - t0
    n = sum(1 for _ in open(out)) if os.path.exists(out) else -1
    print("dosplitalign (builds the FASTA index): rc %d, %.3f s wall, %d alignment lines" % (p.returncode, dt, n))
    print(p.stdout[-600:])
    print(p.stderr[-600:])
    ref_out = open(out, "rb").read()
    # again, as in a pipeline whose reference already has its index (the first run wrote <fasta>.fai)
    best, _ = timed(args, reps=5, label="dosplitalign with the index in place")
    print("best of 5: %.3f s wall -> %.2f M candidates/s tool-level" % (best, nf * rp / best / 1e6))
    _, p = timed(args, env={"DEFUSE_TIMING": "1"}, reps=1, label="with DEFUSE_TIMING")
    print(p.stderr[-3000:])
    assert open(out, "rb").read() == ref_out
    b2, p = timed(args, env={"DEFUSE_DSA_INPROCESS": "1", "DEFUSE_TIMING": "1"}, reps=3, label="worker as a thread of the one process (DEFUSE_DSA_INPROCESS=1)")
    print(p.stderr[-1800:])
    assert open(out, "rb").read() == ref_out
    b3, _ = timed(args, env={"DEFUSE_DSA_PINNED": "1"}, reps=3, label="slots pinned by the worker (DEFUSE_DSA_PINNED=1)")
    assert open(out, "rb").read() == ref_out
    b4, _ = timed(args, env={"DEFUSE_THREADS": "16"}, reps=3, label="16 threads per team")
    print("summary: default %.3f s, in-process %.3f s, pinned %.3f s, 16 threads %.3f s" % (best, b2, b3, b4))
    # the pipeline's next steps: sort -n -k 1 (scripts/defuse_run.pl:528) and evalsplitalign
    srt = out + ".sorted"
    t0 = time.time()
    subprocess.run("LC_ALL=C sort -n -k 1 %s > %s" % (out, srt), shell=True, check=True)
    print("sort: %.2f s" % (time.time() - t0))
    a = ["-f", case["fasta"], "-e", case["exons"], "-u", str(case["ufrag"]), "-s", str(case["sfrag"]), "-n", str(case["minread"]),
         "-x", str(case["maxread"]), "-r", case["regions"], "-a", srt, "-q", os.path.join(d, "seq.txt"), "-b", os.path.join(d, "break.txt"),
         "-p", os.path.join(d, "predalign.txt")]
    t0 = time.time()
    p = subprocess.run(["bin/evalsplitalign"] + a, capture_output=True, text=True)
    print("evalsplitalign: rc %d, %.2f s wall, %d break lines" % (p.returncode, time.time() - t0,
          sum(1 for _ in open(os.path.join(d, "break.txt"))) if p.returncode == 0 else -1))
    print(p.stderr[-300:])
    if copies > 1:
        t0 = time.time()
        big = replicate(case, d, copies)
        print("replicated %d x in %.1f s: %.2f GB of SAM, %.2f GB per FASTQ" % (copies, time.time() - t0, os.path.getsize(big["improper"]) / 1e9, os.path.getsize(big["seq1"]) / 1e9))
        bargs = pipeline_case.tool_args(big, os.path.join(d, "big.align.txt"))
        subprocess.run(["bin/dosplitalign"] + bargs, capture_output=True, text=True)          # (builds the index of the big FASTA)
        best, p = timed(bargs, env={"DEFUSE_TIMING": "1"}, reps=3, label="%d x: dosplitalign" % copies)
        nlines = sum(1 for _ in open(os.path.join(d, "big.align.txt")))
        print(p.stderr[-3000:])
        print("%d candidates, %d alignment lines (%d x %d), best %.3f s wall -> %.2f M candidates/s tool-level" % (copies * nf * rp, nlines, copies, n, best, copies * nf * rp / best / 1e6))
        assert nlines == copies * n
        b16, _ = timed(bargs, env={"DEFUSE_THREADS": "16"}, reps=2, label="%d x, 16 threads per team" % copies)
        print("%d x with 16 threads: %.3f s -> %.2f M candidates/s" % (copies, b16, copies * nf * rp / b16 / 1e6))


if __name__ == "__main__":
    main()
