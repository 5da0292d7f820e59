# HBM traffic of the bench workload's kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes
# (MI355X_MICROARCH.md: TCC slots do not fit both), summed per kernel and averaged per launch.
# Writes gpurun_out/pmc_traffic_raw.json; profiles/rNN/pmc_traffic.json is made from it.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$c -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/pmc_$c.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, json, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{R}/gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    tot = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k in tot:
        res[k][c + "_KB_per_launch"] = tot[k] / max(1, len(disp[k]))
        res[k]["launches"] = len(disp[k])
json.dump(res, open(f"{R}/gpurun_out/pmc_traffic_raw.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("WRITE_SIZE_KB_per_launch", 0))[:8]:
    print(k[:50], v)
PY
