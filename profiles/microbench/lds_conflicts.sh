# Where do the LDS bank conflicts of k_fill_fast come from?  SQ_LDS_* counters of the fill kernel for the product build,
# ablation builds (build_var/lib_<name>.so: norep = no tail replay, notail = no tail at all) and for a workload whose fusions
# fill whole waves (128 reads per fusion: no wave straddles two fusions' tables).
#   gpurun -- bash profiles/microbench/lds_conflicts.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lds_conflicts
mkdir -p $O
run() {   # name, lib (or ""), extra bench args
  local name=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export DEFUSE_DSA_LIB=$lib; else unset DEFUSE_DSA_LIB; fi
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
    -d $O/$name -o p --output-format csv -- python3 $R/bench.py --profile-run --warmup 1 --steps 3 "$@" > $O/$name.log 2>&1 || { tail -5 $O/$name.log; return 1; }
  python3 - $O/$name $name <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k in sorted(tot):
    if "k_fill_fast" in k or "k_replay" in k or "k_plan" in k or "k_rank" in k:
        d = {c: v / len(disp[k]) for c, v in tot[k].items()}
        print("%-8s %-22s conflict/active %.3f  %s" % (sys.argv[2], k[-22:], d.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, d.get("SQ_LDS_IDX_ACTIVE", 0)),
              {c: "%.4g" % v for c, v in sorted(d.items())}))
PY
}
run main "" && run norep $R/build_var/lib_norep.so && run notail $R/build_var/lib_notail.so && run reads128 "" --fusions 7813 --reads 128 && run reads64 "" --fusions 15625 --reads 64
