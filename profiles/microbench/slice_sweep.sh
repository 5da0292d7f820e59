# sweeps the slice size of the two-lane pipeline (DEFUSE_DSA_SLICE_PAIRS) on the bench workload
for sp in ${SWEEP:-250112 333568 500224 2000000}; do
  echo "slice_pairs=$sp"
  DEFUSE_DSA_SLICE_PAIRS=$sp timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['stage_ms'], d['roofline']['kernel_ms'])" || exit 1
done
