# The EM kernel with and without the breakpoint search (DEFUSE_MPE_NO_JUMP): parity tests (unless "quick"), then the 5 M fragment
# probe; with build_var/lib_mpestats.so present (-DMPE_PHASE_STATS -DMPE_PATH_STATS) its counters as well.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/em_jump; mkdir -p $O; cd $R
if [ "$1" != quick ]; then
  timeout -k 10 900 python -m pytest tests/test_clustermatepairs.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
  tail -2 $O/tests.log
fi
timeout -k 10 400 python3 profiles/microbench/em_probe.py 5000000 3 > $O/probe.txt 2>&1 || { tail -20 $O/probe.txt; exit 1; }
if [ "$1" != quick ]; then
  DEFUSE_MPE_NO_JUMP=1 timeout -k 10 400 python3 profiles/microbench/em_probe.py 5000000 2 >> $O/probe.txt 2>&1 || { tail -20 $O/probe.txt; exit 1; }
fi
if [ -f build_var/lib_mpestats.so ]; then
  DEFUSE_DSA_LIB=$R/build_var/lib_mpestats.so timeout -k 10 400 python3 profiles/microbench/em_probe.py 5000000 1 >> $O/probe.txt 2>&1 || { tail -20 $O/probe.txt; exit 1; }
fi
cat $O/probe.txt
