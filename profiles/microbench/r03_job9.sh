set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_job9; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_dsa_gpu.py -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 600 python tests/stress_dsa.py 60 7000 > $O/stress.log 2>&1 || { tail -30 $O/stress.log; exit 1; }
tail -1 $O/stress.log
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_job9/bench.json").read().strip().splitlines()[-1])
print("aligns/s %.1f M  ms_per_step %.4f  stage_ms %s rerun %.1f M one_shot %.2f ms" % (d["value"] / 1e6, d["ms_per_step"], d["stage_ms"], d["resident_rerun"]["value"]/1e6, d["one_shot"]["ms_per_1M_aligns"]))
PY
DEFUSE_DSA_LIB=build_var/lib_stats.so timeout -k 10 200 python bench.py --profile-run --steps 2 --warmup 0 2>&1 >/dev/null | grep "\[stats\]" | tail -5
