set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_job3; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
