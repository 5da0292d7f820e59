#!/bin/bash
# per-kernel times of clustermatepairs' EM kernels on the config-3 probe (rocprofv3 kernel trace)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
D=/tmp/cmp_scale
python3 $R/profiles/microbench/cmp_scale.py --fragments ${1:-5000000} --out $D --keep > $R/gpurun_out/mpe_prof_base.json || exit 1
cd /tmp
DEFUSE_FULL_EXIT=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/mpe_prof -o mpe --output-format csv -- $R/bin/clustermatepairs -a $D/spanning.txt -u 300 -s 30 -p 0.95 -m 5 -c $D/cl.prof > $R/gpurun_out/mpe_prof.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
f = glob.glob(f"{R}/gpurun_out/mpe_prof/**/*kernel_stats.csv", recursive=True)[0]
with open(f"{R}/gpurun_out/mpe_kstats.txt", "w") as o:
    for r in csv.DictReader(open(f)):
        line = f'{r["Name"][:60]:60s} calls {r["Calls"]:>4s} total_ms {float(r["TotalDurationNs"])/1e6:10.2f} avg_ms {float(r["AverageNs"])/1e6:10.2f} max_ms {float(r["MaxNs"])/1e6:10.2f} pct {r["Percentage"]}'
        print(line); o.write(line + "\n")
PY
rm -rf $D $R/gpurun_out/mpe_prof
