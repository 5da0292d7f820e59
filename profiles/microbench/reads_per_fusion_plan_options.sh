R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/reads_per_fusion; mkdir -p $O; cd $R
for env in "" "DEFUSE_DSA_NO_RANK=1 DEFUSE_DSA_NO_TIGHTEN=1" "DEFUSE_DSA_NO_TIGHTEN=1" "DEFUSE_DSA_NO_RANK=1"; do
for rf in "100000 10" "200000 5" "333334 3"; do
  set -- $rf
  env $env timeout -k 10 300 python bench.py --no-cpu-baseline --no-sensitivity --fusions $1 --reads $2 --steps 20 > $O/x.json 2> $O/x.err || { tail -5 $O/x.err; exit 1; }
  python3 - $O/x.json $1 $2 "$env" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-50s %8s fusions x %3s reads: %6.1f M aligns/s, ms per step %.3f, stages %s" % (sys.argv[4], sys.argv[2], sys.argv[3], d["value"] / 1e6, d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}))
PY
done; done
