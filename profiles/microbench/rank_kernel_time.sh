cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/kt3 -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/kt3.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/kt3/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_rank" in r["Name"]:
        print("k_rank_in_fusion us", float(r["AverageNs"]) / 1e3)
PY
rm -rf gpurun_out/kt3
