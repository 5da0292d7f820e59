#!/bin/bash
# SQ counters of the EM kernel of clustermatepairs on the config-3 probe (one rocprofv3 --pmc pass, kernel trace only):
# how many of a wave's 64 lanes its VALU instructions keep busy, and how much of the time the SIMDs issue
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
D=/tmp/cmp_scale
python3 $R/profiles/microbench/cmp_scale.py --fragments ${1:-1000000} --out $D --keep > /dev/null || exit 1
cd /tmp
DEFUSE_FULL_EXIT=1 rocprofv3 --kernel-trace --pmc ${PMC:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVES} -d $R/gpurun_out/pmc_mpe -o p --output-format csv -- $R/bin/clustermatepairs -a $D/spanning.txt -u 300 -s 30 -p 0.95 -m 5 -c $D/cl.pmc > $R/gpurun_out/pmc_mpe.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, os, collections, json
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
out = {}
for f in glob.glob(f"{R}/gpurun_out/pmc_mpe/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        acc["k_mpe_problem_wave" if "k_mpe_problem_wave" in name else name[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if "mpe" in k:
            out[k] = dict(v)
for k, v in out.items():
    if v.get("SQ_ACTIVE_INST_VALU"):
        v["lanes_busy_per_valu_instruction"] = v.get("SQ_THREAD_CYCLES_VALU", 0.0) / v["SQ_ACTIVE_INST_VALU"]
print(json.dumps(out, indent=1))
open(f"{R}/gpurun_out/pmc_mpe.json", "w").write(json.dumps(out, indent=1))
PY
rm -rf $D $R/gpurun_out/pmc_mpe
