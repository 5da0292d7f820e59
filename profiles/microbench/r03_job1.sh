set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_job1; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_dsa_gpu.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 400 python bench.py > $O/bench_baseline.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
cat $O/bench_baseline.json
bash profiles/microbench/variant_bench.sh lds norep notail > $O/variants.txt 2>&1; cat $O/variants.txt
bash profiles/microbench/lds_conflicts.sh > $O/lds.txt 2>&1; cat $O/lds.txt
