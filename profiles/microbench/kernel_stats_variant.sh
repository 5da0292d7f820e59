# per-kernel times (rocprofv3 --kernel-trace --stats) of bench.py against ablation builds: $@ = names of build_var/lib_<name>.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  O=$R/gpurun_out/variant_$v
  mkdir -p $O
  export DEFUSE_DSA_LIB=$R/build_var/lib_$v.so
  rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline --warmup 1 --steps 10 > $O/kt.log 2>&1 || exit 1
  echo "variant $v"
  python3 - $O <<'PY'
import csv, glob, sys
ks = glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(ks)):
    if any(k in r["Name"] for k in ("k_emit", "k_replay", "k_fill_fast<0")):
        print("  %-30s avg_us %10.1f" % (r["Name"].split("(")[0][-30:], float(r["AverageNs"]) / 1e3))
PY
done
