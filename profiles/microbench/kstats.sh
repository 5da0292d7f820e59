# per-kernel times of bench.py (rocprofv3 --kernel-trace --stats), the kernels above 0.05 %; extra arguments / environment go to bench.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --profile-run --warmup 1 --steps 20 "$@" > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 1; }
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/kstats"
ks = glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(ks)):
    if float(r["Percentage"]) > 0.05:
        print("%-44s calls %5s avg_us %10.1f pct %s" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
