# per-kernel durations of the bench workload (rocprofv3 kernel trace), summary to gpurun_out/kstats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/kt.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(f"{R}/gpurun_out/kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(f"{R}/gpurun_out/kstats.txt", "w") as o:
    for r in rows:
        line = f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:10.1f} pct {r["Percentage"]}'
        print(line); o.write(line + "\n")
PY
