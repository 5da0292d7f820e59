# rocprofv3 kernel stats of bin/clustermatepairs at 50 M fragments (BASELINE configs[2]): the bin-pair kernels and sorts of cmp_api.hip, the EM kernels.
#   gpurun -- bash profiles/microbench/cmp50_kernels.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cmp50_kernels; mkdir -p $O; cd $R
python profiles/microbench/cmp_scale.py --fragments ${1:-50000000} --out /tmp/cmp50 --generate-only > $O/gen.json 2>&1 || { cat $O/gen.json; exit 1; }
cd /tmp
DEFUSE_FULL_EXIT=1 DEFUSE_THREADS=16 DEFUSE_TIMING=1 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $R/bin/clustermatepairs -a /tmp/cmp50/spanning.txt -c /tmp/cmp50/clusters.txt -u 300 -s 30 -p 0.95 -m 5 > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 1; }
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/cmp50_kernels"
f = glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True)[0]
with open(O + "/cmp50_kernel_stats.txt", "w") as out:
    for r in csv.DictReader(open(f)):
        line = "%-90s calls %5s total_ms %10.2f avg_us %10.1f pct %s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"])
        print(line)
        out.write(line + "\n")
PY
grep -E "bin pairs on the device|kernel " $O/kt.log
rm -rf /tmp/cmp50 $O/kt
