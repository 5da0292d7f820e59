#!/bin/bash
# Roofline figures of the EM kernels of clustermatepairs (k_mpe_problem_wave; k_mpe_kmeans and k_mpe_seed listed beside it) on the config-3 probe (default 5 M fragments):
# kernel time (rocprofv3 kernel trace), SQ counters (VALU issue, lanes busy, waits), HBM-side traffic (FETCH_SIZE / WRITE_SIZE,
# separate passes) and L2 hits, each in its own rocprofv3 --pmc pass.  Writes gpurun_out/mpe_roofline/roofline.json
# (copy to profiles/rNN/clustermatepairs/).    gpurun -- bash profiles/microbench/mpe_roofline.sh [fragments]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
D=/tmp/cmp_scale
O=$R/gpurun_out/mpe_roofline
mkdir -p $O
python3 $R/profiles/microbench/cmp_scale.py --fragments ${1:-5000000} --out $D --generate-only > $O/gen.json || exit 1
cd /tmp
T="$R/bin/clustermatepairs -a $D/spanning.txt -u 300 -s 30 -p 0.95 -m 5 -c $D/cl.pmc"
export DEFUSE_CMP_CHUNKS=1 DEFUSE_TIMING=1 DEFUSE_FULL_EXIT=1
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $T > $O/kt.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/sq -o p --output-format csv -- $T > $O/sq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_SMEM -d $O/sq2 -o p --output-format csv -- $T > $O/sq2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $T > $O/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $T > $O/write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc -o p --output-format csv -- $T > $O/tcc.log 2>&1 || exit 1
python3 - <<'PY'
import collections, csv, glob, json, os, re, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
O = R + "/gpurun_out/mpe_roofline"
sys.path.insert(0, R)
import bench
K = "k_mpe_problem_wave"
def counters(d, kernel=K):
    acc = collections.defaultdict(float)
    for f in glob.glob(O + "/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
    return dict(acc)
def kernel_time(kernel):
    ms = calls = 0.0
    for f in glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Name"]:
                ms += float(r["TotalDurationNs"]) / 1e6
                calls += float(r["Calls"])
    return ms, calls
# The problems run in shares on streams of their own (mpe_api.hip): the EM kernel's launches overlap each other and the
# k-means kernel, so the sum of their durations is device work, not elapsed time; the batch's elapsed time is the tool's.
ms, calls = kernel_time(K)
log = open(O + "/kt.log").read()
m = re.search(r"(\d+) bin pairs, (\d+) mate pairs, (\d+) EM iterations", log)
bin_pairs, mate_pairs, em_iters = (int(v) for v in m.groups())
mb = re.search(r"EM kernels? ([0-9.]+) ms", log) or re.search(r"kernel[s]? ([0-9.]+) ms", log)
batch_ms = float(mb.group(1)) if mb else None
sq, sq2, fe, wr, tcc = counters("sq"), counters("sq2"), counters("fetch"), counters("write"), counters("tcc")
others = {k: {"sq": counters("sq", k), "fetch": counters("fetch", k), "write": counters("write", k)} for k in ("k_mpe_kmeans", "k_mpe_seed")}
# the batch as a whole: elapsed time of the tool's HIP events around the three kernels (the shares' launches overlap), all
# three kernels' instructions and bytes
elapsed_ms = batch_ms if batch_ms else ms
cycles = elapsed_ms * 1e-3 * 2.4e9                       # nominal clock
simds = 256 * 4
valu_all = sq["SQ_ACTIVE_INST_VALU"] + sum(o["sq"].get("SQ_ACTIVE_INST_VALU", 0.0) for o in others.values())
thread_all = sq["SQ_THREAD_CYCLES_VALU"] + sum(o["sq"].get("SQ_THREAD_CYCLES_VALU", 0.0) for o in others.values())
issue = 4.0 * valu_all / (cycles * simds)          # SQ counts quad-cycles
lanes = thread_all / valu_all
for o in others.values():
    fe["FETCH_SIZE"] = fe.get("FETCH_SIZE", 0.0) + o["fetch"].get("FETCH_SIZE", 0.0)
    wr["WRITE_SIZE"] = wr.get("WRITE_SIZE", 0.0) + o["write"].get("WRITE_SIZE", 0.0)
ms_work, ms = ms, elapsed_ms
fetch_b, write_b = fe.get("FETCH_SIZE", 0.0) * 1024, wr.get("WRITE_SIZE", 0.0) * 1024
out = {
    "kernel": K, "source_hash": bench.library_hash(), "workload": json.load(open(O + "/gen.json")),
    "kernel_ms": ms, "launches": calls,
    "kernel_ms_note": "elapsed time of the batch (HIP events of mpe_cluster_batch around k_mpe_seed, k_mpe_kmeans, k_mpe_problem_wave; the shares' launches overlap); every fraction below is over this time and over all three kernels",
    "sum_of_k_mpe_problem_wave_launch_durations_ms": ms_work,
    "other_kernels": {k: dict(zip(("ms", "launches"), kernel_time(k)), **{"sq": counters("sq", k), "fetch_KB": counters("fetch", k).get("FETCH_SIZE"), "write_KB": counters("write", k).get("WRITE_SIZE")})
                      for k in ("k_mpe_kmeans", "k_mpe_seed")},
    "bin_pairs": bin_pairs, "mate_pairs": mate_pairs, "em_iterations": em_iters,
    "bound": "latency of dependent FP64 chains and their loads (neither HBM nor MFMA): see waits",
    "valu": {"issue_busy_frac_of_simd_cycles": issue, "lanes_busy_per_instruction": lanes,
             "fp64_lane_throughput_frac": issue * lanes / 64.0,
             "note": "nominal 2.4 GHz, one quad-cycle of issue per wave instruction; the serial sums the reference prescribes "
                     "keep one lane per component (M step) or per fit (likelihood chain) busy"},
    "waits": {"wait_any_over_wave_cycles": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
              "wait_inst_any_over_wave_cycles": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
              "salu_per_valu": sq["SQ_INSTS_SALU"] / sq["SQ_INSTS_VALU"]},
    "memory": {"fetch_bytes_raw": fetch_b, "write_bytes": write_b,
               "fabric_GBps_raw": (fetch_b + write_b) / (ms * 1e-3) / 1e9, "hbm_peak_GBps": 8000.0,
               "frac_of_hbm_peak_raw": (fetch_b + write_b) / (ms * 1e-3) / 1e9 / 8000.0,
               "bytes_per_em_iteration_raw": (fetch_b + write_b) / max(1, em_iters),
               "l2_hit_rate": tcc.get("TCC_HIT_sum", 0.0) / max(1.0, tcc.get("TCC_HIT_sum", 0.0) + tcc.get("TCC_MISS_sum", 0.0)),
               "vmem_read_instructions": sq2.get("SQ_INSTS_VMEM_RD"), "vmem_write_instructions": sq2.get("SQ_INSTS_VMEM_WR"),
               "note": "FETCH_SIZE as counted (8-byte scattered reads: the gfx950 x2 correction is calibrated for 16-B streaming reads only)"},
    "raw": {"sq": sq, "sq2": sq2, "tcc": tcc},
}
json.dump(out, open(O + "/roofline.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel_ms", "valu", "waits")}, indent=1))
print(json.dumps(out["memory"], indent=1))
PY
rm -rf $D $O/kt $O/sq $O/sq2 $O/fetch $O/write $O/tcc
