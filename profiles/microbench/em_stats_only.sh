set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/em_jump; mkdir -p $O; cd $R
DEFUSE_DSA_LIB=$R/build_var/lib_mpestats.so timeout -k 10 400 python3 profiles/microbench/em_probe.py 5000000 1 > $O/stats.txt 2>&1 || { tail -20 $O/stats.txt; exit 1; }
grep "M step" $O/stats.txt
