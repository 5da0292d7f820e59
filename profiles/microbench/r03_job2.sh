set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_job2; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_dsa_gpu.py -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 600 python tests/stress_dsa.py 40 7000 > $O/stress.log 2>&1 || { tail -30 $O/stress.log; exit 1; }
tail -2 $O/stress.log
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_job2/bench.json").read().strip().splitlines()[-1])
print("aligns/s %.1f M  ms_per_step %.4f  stage_ms %s rerun %.1f M one_shot %s" % (d["value"] / 1e6, d["ms_per_step"], d["stage_ms"], d["resident_rerun"]["value"]/1e6, d.get("one_shot")))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --profile-run --warmup 1 --steps 20 > $O/kt.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_job2"
ks = glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(ks)):
    print("%-44s calls %5s avg_us %10.1f pct %s" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
