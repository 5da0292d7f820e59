set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_job4; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_dsa_gpu.py -x -q --durations=5 > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -12 $O/tests.log
