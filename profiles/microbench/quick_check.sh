# Iteration check on the GPU box: DP parity tests, a short randomised stress, the bench line, the per-kernel times of one
# bench run and (if build_var/lib_stats.so is there) the pruning / stage statistics of the -DDSA_PRUNE_STATS build.
#   gpurun -- bash profiles/microbench/quick_check.sh [stress rounds]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/quick
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_dsa_gpu.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 600 python tests/stress_dsa.py ${1:-60} 7000 > $O/stress.log 2>&1 || { tail -30 $O/stress.log; exit 1; }
tail -2 $O/stress.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/quick/bench.json").read().strip().splitlines()[-1])
print("aligns/s %.1f M  ms_per_step %.4f  stage_ms %s" % (d["value"] / 1e6, d["ms_per_step"], d["stage_ms"]))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline --warmup 1 --steps 20 > $O/kt.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/quick"
ks = glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(ks)):
    if float(r["Percentage"]) > 0.05:
        print("%-40s calls %5s avg_us %10.1f pct %s" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cd $R
if [ -f build_var/lib_stats.so ]; then
  DEFUSE_DSA_LIB=build_var/lib_stats.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 0 2>&1 >/dev/null | grep "\[stats\]" | tail -1
fi
