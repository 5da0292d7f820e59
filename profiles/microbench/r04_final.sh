# The round's committed evidence in one call: full GPU test suite, rocprofv3 kernel stats + PMC passes of the bench workload and of the
# configs[3] shape, the bench lines (which print the counters only if they carry the running library's hash), stress runs.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_final; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
bash profiles/microbench/profile_bench.sh > $O/profile_bench.log 2>&1 || { tail -30 $O/profile_bench.log; exit 1; }
tail -3 $O/profile_bench.log
PB_ARGS="--workload config4 --fusions 50000" PB_SUFFIX=_config4 PB_WORKLOAD=config4:50000 bash profiles/microbench/profile_bench.sh > $O/profile_bench_config4.log 2>&1 || { tail -30 $O/profile_bench_config4.log; exit 1; }
tail -3 $O/profile_bench_config4.log
mkdir -p $R/profiles/r04
cp $R/gpurun_out/profile_bench/bench_kernel_stats.csv $R/gpurun_out/profile_bench/pmc_traffic.json $R/gpurun_out/profile_bench/pmc_sq.json $O/
cp $R/gpurun_out/profile_bench_config4/bench_kernel_stats_config4.csv $R/gpurun_out/profile_bench_config4/pmc_traffic_config4.json $R/gpurun_out/profile_bench_config4/pmc_sq_config4.json $O/
# the bench line carries the counters only while they belong to the library that runs: put them where bench.py looks (this copy
# of the tree lives on the box; the committed ones come back through gpurun_out/)
cp $O/pmc_traffic.json $O/pmc_sq.json $O/pmc_traffic_config4.json $O/pmc_sq_config4.json $R/profiles/r04/
python profiles/microbench/fill_mix.py > $O/fill_mix.log 2>&1 && cp $R/profiles/r04/fill_mix.json $O/
cd $R && timeout -k 10 500 python bench.py > $O/bench_r04.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
cd $R && timeout -k 10 400 python bench.py --no-cpu-baseline --workload config4 --fusions 50000 > $O/bench_config4_shape.json 2> $O/bench_c4.err || { tail -20 $O/bench_c4.err; exit 1; }
cd $R && DEFUSE_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --fusions 100000 --steps 3 > $O/bench_gloo2_rehearsal.json 2> $O/bench_g2.err || { tail -20 $O/bench_g2.err; exit 1; }
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_final/"
for f in ("bench_r04.json", "bench_config4_shape.json", "bench_gloo2_rehearsal.json"):
    d = json.loads(open(O + f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "aligns/s %.1f M  ms_per_step %.4f  stage_ms %s rerun %.1f M; roofline %s frac %.4f hbm_frac %.5f traffic %s" % (
        d["value"] / 1e6, d["ms_per_step"], d["stage_ms"], d["resident_rerun"]["value"] / 1e6, r["bound"], r["frac"], r["hbm_frac"], r["traffic"]))
    if "strong_scaling" in d:
        print("  strong_scaling", d["strong_scaling"]["efficiency"], d["strong_scaling"]["n1_aligns_per_s"])
PY
timeout -k 10 500 python tests/stress_dsa.py 60 7000 > $O/stress_dsa.log 2>&1 || { tail -30 $O/stress_dsa.log; exit 1; }
tail -1 $O/stress_dsa.log
timeout -k 10 400 python tests/stress_tools.py 20 > $O/stress_tools.log 2>&1 || { tail -30 $O/stress_tools.log; exit 1; }
tail -1 $O/stress_tools.log
