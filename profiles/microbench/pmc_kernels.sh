# SQ counters per kernel of one bench.py run (rocprofv3 --pmc, no other trace domain): $1 = substring of the kernel names to print
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_kernels
mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --warmup 1 --steps 3"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/p1 -o p --output-format csv -- $B > $O/p1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM -d $O/p2 -o p --output-format csv -- $B > $O/p2.log 2>&1 || exit 1
python3 - "$1" <<'PY'
import collections, csv, glob, os, sys
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_kernels"
for d in ("p1", "p2"):
    f = glob.glob(O + "/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k in sorted(tot):
        if sys.argv[1] in k:
            print(k[-28:], {c: "%.4g" % (v / len(disp[k])) for c, v in sorted(tot[k].items())})
PY
