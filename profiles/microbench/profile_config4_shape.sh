# rocprofv3 kernel stats of one GPU on the shape of BASELINE configs[3] (2x100 bp, 200 reads per fusion): one resident upload of
# 50 000 fusions = 10 M aligns, the unit a rank of the strong-scaling bench repeats.  Output: gpurun_out/c4/ (copy to profiles/rNN/).
#   gpurun -- bash profiles/microbench/profile_config4_shape.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c4
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline --workload config4 --fusions 50000 --steps 3 --warmup 1 > $O/bench_config4_shape.json 2> $O/bench.err || exit 1
python3 - <<'PY'
import csv, glob, json, os, shutil
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/c4"
ks = glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(ks, O + "/config4_shape_kernel_stats.csv")
for r in csv.DictReader(open(ks)):
    if float(r["Percentage"]) > 0.3:
        print("%-40s calls %5s avg_us %10.1f pct %s" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
d = json.loads(open(O + "/bench_config4_shape.json").read().strip().splitlines()[-1])
print(d["value"] / 1e6, "M aligns/s", d["ms_per_step"], "ms per step", d["stage_ms"])
PY
