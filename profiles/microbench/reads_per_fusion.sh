# The DP at a fixed 1 M aligns (2x76, Lref 389: the cells of BASELINE configs[1]) with fewer and fewer reads per fusion: which fill
# kernel a workgroup of 256 pairs gets (<= 4 fusions: 25-row tables; <= 20 / <= 40: split tables; more: k_fill_generic) and what it costs.
#   gpurun -- bash profiles/microbench/reads_per_fusion.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/reads_per_fusion; mkdir -p $O; cd $R
for rf in "10000 100" "40000 25" "100000 10" "200000 5" "333334 3" "1000000 1"; do
  set -- $rf
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-sensitivity --fusions $1 --reads $2 --steps 20 > $O/f$1.json 2> $O/f$1.err || { tail -5 $O/f$1.err; exit 1; }
  python3 - $O/f$1.json $1 $2 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%8s fusions x %3s reads: %6.1f M aligns/s, ms per step %.3f, stages %s" % (sys.argv[2], sys.argv[3], d["value"] / 1e6, d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}))
PY
done | tee $O/summary.txt
