# The round's committed evidence in one call: full GPU test suite, the bench line, rocprofv3 kernel stats + PMC passes, static mix.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_final; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
bash profiles/microbench/profile_bench.sh > $O/profile_bench.log 2>&1 || { tail -30 $O/profile_bench.log; exit 1; }
tail -3 $O/profile_bench.log
cp $R/gpurun_out/profile_bench/bench_kernel_stats.csv $R/gpurun_out/profile_bench/pmc_traffic.json $R/gpurun_out/profile_bench/pmc_sq.json $O/
# the bench line carries the counters only while they belong to the library that runs: put them where bench.py looks (this copy
# of the tree lives on the box; the committed ones come back through gpurun_out/)
cp $O/bench_kernel_stats.csv $O/pmc_traffic.json $O/pmc_sq.json $R/profiles/r03/
cd $R && timeout -k 10 500 python bench.py > $O/bench_r03.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
cd $R && timeout -k 10 300 python bench.py --no-cpu-baseline --workload config4 --fusions 50000 > $O/bench_config4_shape.json 2> $O/bench_c4.err || { tail -20 $O/bench_c4.err; exit 1; }
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_final/"
for f in ("bench_r03.json", "bench_config4_shape.json"):
    d = json.loads(open(O + f).read().strip().splitlines()[-1])
    print(f, "aligns/s %.1f M  ms_per_step %.4f  stage_ms %s rerun %.1f M one_shot %s" % (d["value"] / 1e6, d["ms_per_step"], d["stage_ms"], d["resident_rerun"]["value"]/1e6, d.get("one_shot", {}).get("ms_per_1M_aligns")))
PY
