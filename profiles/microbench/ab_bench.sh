# A/B on one box: bench.py (fill kernel time, step time) for the product build and for build_var/lib_<name>.so variants, three
# rounds interleaved so that a drift of the box shows.   gpurun -- bash profiles/microbench/ab_bench.sh <name> [<name> ...]
R=$GRAFT_REPO_ROOT; cd $R
for round in 1 2 3; do
  for v in product "$@"; do
    if [ $v = product ]; then unset DEFUSE_DSA_LIB; else export DEFUSE_DSA_LIB=$R/build_var/lib_$v.so; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --profile-run --steps 60 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('round $round %-10s step %.4f ms  fill %.4f  plan %.4f  finish %.4f' % ('$v', d['ms_per_step'], d['stage_ms']['fill'], d['stage_ms']['plan'], d['stage_ms']['finish']))" || exit 1
  done
done
