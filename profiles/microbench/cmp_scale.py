"""clustermatepairs (+ setcover) at a fraction of BASELINE.json configs[2]:
loci pairs over 24 synthetic chromosomes (50-250 Mb, seed 3), support per locus ~ 1/k on 1..500 (so most loci
fall below -m 5), ends placed U(0, mu+3sigma-2*100) upstream of each breakpoint, 2x100 bp, 5 % of the fragments
multi-map one end to a second locus, 10 % concordant decoys; -m 5 -p 0.95 -u 300 -s 30.

    python profiles/microbench/cmp_scale.py --fragments 5000000

The full configuration is 50 M fragments; the generator is vectorised (numpy + pandas) so the input text is
written at a few million lines per second.  Prints one JSON line with the tools' stage timings (DEFUSE_TIMING)."""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

sys.path.insert(0, ROOT)
from tests.cmp_cases import config3_write as generate   # the generator lives with the tests (it is a parity-test input)


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
