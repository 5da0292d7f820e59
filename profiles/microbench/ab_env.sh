# A/B of an environment switch on one box and one binary: bench.py with and without "$1" (e.g. DEFUSE_DSA_NO_RESTART=1), three rounds
R=$GRAFT_REPO_ROOT; cd $R
for round in 1 2 3; do
  for v in default "$1"; do
    ( if [ "$v" != default ]; then export "$v"; fi
      timeout -k 10 200 python bench.py --no-cpu-baseline --profile-run --steps 60 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('round $round %-26s step %.4f ms  fill %.4f  plan %.4f  finish %.4f' % ('$v', d['ms_per_step'], d['stage_ms']['fill'], d['stage_ms']['plan'], d['stage_ms']['finish']))" ) || exit 1
  done
done
