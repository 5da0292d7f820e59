# bench.py against alternative builds of the library (build_var/lib_<NAME>.so)
for v in "$@"; do
  echo "variant=$v"
  DEFUSE_DSA_LIB=build_var/lib_$v.so timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['stage_ms'], d['roofline']['kernel_ms'])" || exit 1
done
