"""End-to-end wall time of the drop-in dosplitalign / evalsplitalign binaries on a synthetic case
(tests/pipeline_case.py): shows how much of a tool run is host text I/O and how much is the GPU.
Usage: python profiles/microbench/tool_throughput.py [n_fusions] [reads_per_fusion]"""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import pipeline_case


def main():
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
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
    print("dosplitalign: rc %d, %.2f s wall, %d alignment lines" % (p.returncode, dt, n))
    print(p.stdout[-600:])
    print(p.stderr[-600:])
    # again, as in a pipeline whose reference already has its index (the first run wrote <fasta>.fai), with the stage times
    for rep in (1, 2):
        t0 = time.time()
        p = subprocess.run(["bin/dosplitalign"] + args, capture_output=True, text=True, env=dict(os.environ, DEFUSE_TIMING="1"))
        print("dosplitalign with the index in place, run %d: rc %d, %.2f s wall" % (rep, p.returncode, time.time() - t0))
    print(p.stderr[-1500:])
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


if __name__ == "__main__":
    main()
