"""clustermatepairs (+ setcover) at a fraction of BASELINE.json configs[2]:
loci pairs over 24 synthetic chromosomes (50-250 Mb, seed 3), support per locus ~ 1/k on 1..500 (so most loci
fall below -m 5), ends placed U(0, mu+3sigma-2*100) upstream of each breakpoint, 2x100 bp, 5 % of the fragments
multi-map one end to a second locus, 10 % concordant decoys; -m 5 -p 0.95 -u 300 -s 30.

    python profiles/microbench/cmp_scale.py --fragments 5000000

The full configuration is 50 M fragments; the generator is vectorised (numpy + pandas) so the input text is
written at a few million lines per second.  Prints one JSON line with the tools' stage timings (DEFUSE_TIMING)."""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd

MU, SIGMA, RL = 300.0, 30.0, 100


def generate(n_fragments, path, seed=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fragments", type=int, default=5_000_000)
    ap.add_argument("--out", default="gpurun_out/cmp_scale")
    ap.add_argument("--generate-only", action="store_true")
    ap.add_argument("--keep", action="store_true", help="leave the input and output files in --out")
    ap.add_argument("--compare-lane", action="store_true", help="run again with a lane per fit for every problem and compare the cluster files")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    span = os.path.join(args.out, "spanning.txt")
    t0 = time.time()
    n_frag, n_loci, n_lines = generate(args.fragments, span)
    res = {"fragments": n_frag, "loci": n_loci, "lines": n_lines, "input_bytes": os.path.getsize(span), "generate_s": round(time.time() - t0, 2)}
    if args.generate_only:
        print(json.dumps(res))
        return
    env = dict(os.environ, DEFUSE_TIMING="1")
    cl, sc = os.path.join(args.out, "clusters.txt"), os.path.join(args.out, "clusters.sc")
    t0 = time.time()
    r = subprocess.run([ROOT + "/bin/clustermatepairs", "-a", span, "-c", cl, "-u", "300", "-s", "30", "-p", "0.95", "-m", "5"],
                       capture_output=True, text=True, env=env)
    res["clustermatepairs_s"] = round(time.time() - t0, 2)
    res["clustermatepairs_rc"] = r.returncode
    res["clustermatepairs_stdout_tail"] = r.stdout.strip().splitlines()[-1:] if r.stdout.strip() else []
    res["clustermatepairs_timing"] = r.stderr.strip().splitlines()[-3:]
    res["fragments_per_s"] = round(n_frag / max(res["clustermatepairs_s"], 1e-9))
    if r.returncode == 0 and args.compare_lane:
        cl2 = cl + ".lane"
        t0 = time.time()
        r2 = subprocess.run([ROOT + "/bin/clustermatepairs", "-a", span, "-c", cl2, "-u", "300", "-s", "30", "-p", "0.95", "-m", "5"],
                            capture_output=True, text=True, env=dict(env, DEFUSE_MPE_WAVE_MIN="1000000000"))
        res["lane_only_s"] = round(time.time() - t0, 2)
        res["lane_only_timing"] = r2.stderr.strip().splitlines()[-1:]
        res["lane_only_identical"] = r2.returncode == 0 and subprocess.run(["cmp", "-s", cl, cl2]).returncode == 0
        if os.path.exists(cl2):
            os.remove(cl2)
    if r.returncode == 0:
        t0 = time.time()
        r = subprocess.run([ROOT + "/bin/setcover", "-c", cl, "-m", "5", "-o", sc], capture_output=True, text=True, env=env)
        res["setcover_s"] = round(time.time() - t0, 2)
        res["setcover_rc"] = r.returncode
        res["setcover_timing"] = r.stderr.strip().splitlines()[-3:]
        res["cluster_lines"] = sum(1 for _ in open(cl))
        res["cover_lines"] = sum(1 for _ in open(sc)) if os.path.exists(sc) else None
    for f in (span, cl, sc):                      # large scratch files do not travel back
        if os.path.exists(f) and not args.keep:
            os.remove(f)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
