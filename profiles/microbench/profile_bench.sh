# One call on the GPU box: rocprofv3 kernel stats + HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) + SQ counters
# (two passes) of `python3 bench.py` (BASELINE configs[1], N = 1), summarised into gpurun_out/profile_bench/ (copy them to profiles/rNN/):
#   bench_kernel_stats.csv, pmc_traffic.json, pmc_sq.json        (copy them to profiles/rNN/)
# Every JSON carries the workload and the SOURCE HASH of the library that ran (dsa_version()); bench.py reports the counters
# only while that hash equals the loaded library's.
#   gpurun -- bash profiles/microbench/profile_bench.sh
# PB_ARGS / PB_SUFFIX / PB_WORKLOAD profile another workload into files with a suffix, e.g. the shape of BASELINE configs[3] (one
# upload of 50 000 fusions x 200 reads, what a rank of the N > 1 bench repeats):
#   PB_ARGS="--workload config4 --fusions 50000" PB_SUFFIX=_config4 PB_WORKLOAD=config4:50000 bash profiles/microbench/profile_bench.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profile_bench${PB_SUFFIX}
mkdir -p $O
export PB_SUFFIX PB_WORKLOAD
B="python3 $R/bench.py --profile-run --warmup 1 $PB_ARGS"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $B --steps 20 > $O/kt.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p --output-format csv -- $B --steps 3 > $O/pmc_$c.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $O/pmc1 -o p1 --output-format csv -- $B --steps 3 > $O/pmc1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES -d $O/pmc2 -o p2 --output-format csv -- $B --steps 3 > $O/pmc2.log 2>&1 || exit 1
python3 - <<'PY'
import collections, csv, glob, json, os, shutil, sys
R = os.environ["GRAFT_REPO_ROOT"]
SUF = os.environ.get("PB_SUFFIX", "")
O = R + "/gpurun_out/profile_bench" + SUF
sys.path.insert(0, R)
import bench
h = bench.library_hash()
wname, _, wf = (os.environ.get("PB_WORKLOAD") or "config2").partition(":")
workload = dict(bench.WORKLOADS[wname])
if wf:
    workload["fusions"] = int(wf)
ks = glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(ks, O + "/bench_kernel_stats%s.csv" % SUF)
for r in csv.DictReader(open(ks)):
    print("%-60s calls %5s avg_us %10.1f pct %s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))

def per_kernel(path, counters):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if r["Counter_Name"] in counters:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return tot, disp

FILL = "k_fill_fast<0>"
def pick(tot):
    ks = [k for k in tot if "k_fill_fast<0" in k] or [k for k in tot if "k_fill_fast" in k]
    return max(ks, key=lambda k: sum(tot[k].values()))

raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(O + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    tot, disp = per_kernel(f, (c,))
    k = pick(tot)
    raw[c] = {"kernel": k, "KB_per_launch": tot[k][c] / len(disp[k]), "launches": len(disp[k])}
# MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads (this kernel's
# loads are dwordx4): doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Both counters are in KB.
read_b = 2.0 * raw["FETCH_SIZE"]["KB_per_launch"] * 1024
write_b = raw["WRITE_SIZE"]["KB_per_launch"] * 1024
# aligns per launch of the dominant kernel, from the bench line of the kernel-trace pass (a batch is cut into slices, one fill launch each)
line = [l for l in open(O + "/kt.log") if l.startswith("{") and '"roofline"' in l][-1]
bl = json.loads(line)
aligns_per_launch = bl["config"]["aligns_per_step_this_gpu"] / bl["roofline"]["launches_per_step"]
json.dump({"workload": workload, "source_hash": h, "kernel": raw["FETCH_SIZE"]["kernel"], "raw": raw, "aligns_per_launch": aligns_per_launch,
           "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b, "hbm_bytes_per_launch": read_b + write_b,
           "correction": "FETCH_SIZE x2 (gfx950, 16 B/lane reads), WRITE_SIZE as is; separate rocprofv3 --pmc passes",
           "note": "per launch of the dominant fill kernel; a batch of more than 2^19 pairs runs as two launches per step"},
          open(O + "/pmc_traffic%s.json" % SUF, "w"), indent=1)
sq = {"workload": workload, "source_hash": h, "aligns_per_launch": aligns_per_launch}
for name, d in (("pass1", "pmc1"), ("pass2", "pmc2")):
    f = glob.glob(O + "/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    tot, disp = per_kernel(f, None if False else set(r for r in ("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES").split()))
    k = pick(tot)
    sq["kernel"] = k
    sq["launches"] = len(disp[k])
    sq[name] = dict(tot[k])
json.dump(sq, open(O + "/pmc_sq%s.json" % SUF, "w"), indent=1)
print(json.dumps({"traffic_GB_per_launch": (read_b + write_b) / 1e9, "read": read_b / 1e9, "write": write_b / 1e9, "hash": h}))
print({k: "%.4g" % (v / sq["launches"]) for k, v in sq["pass1"].items()})
PY
