# SQ counters of the EM kernels on the 5 M fragment probe (two rocprofv3 --pmc passes), per kernel:  gpurun -- bash profiles/microbench/em_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_em
mkdir -p $O
N=${EM_FRAGMENTS:-5000000}
python3 $R/profiles/microbench/em_probe.py $N 1 > $O/probe.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/pmc_km -o p --output-format csv -- python3 $R/profiles/microbench/em_probe.py $N 1 > $O/pmc_km.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT -d $O/pmc_km2 -o p --output-format csv -- python3 $R/profiles/microbench/em_probe.py $N 1 > $O/pmc_km2.log 2>&1 || exit 1
python3 - <<'PY'
import collections, csv, glob, json, os
R = os.environ["GRAFT_REPO_ROOT"]
out = {}
for d in ("pmc_km", "pmc_km2"):
    f = glob.glob(R + "/gpurun_out/r04_em/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mpe" not in k:
            continue
        k = k.replace("(anonymous namespace)::", "").split("(")[0]
        out.setdefault(k, collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in out.items():
    print(k, {a: "%.4g" % b for a, b in v.items()})
json.dump(out, open(R + "/gpurun_out/r04_em/em_pmc.json", "w"), indent=1)
PY
