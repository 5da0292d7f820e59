#!/bin/bash
# EM kernel time of clustermatepairs against the workspace chunk size (DEFUSE_MPE_SCRATCH_MB): how much of it is cache misses
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
D=/tmp/cmp_scale
python3 $R/profiles/microbench/cmp_scale.py --fragments ${1:-5000000} --out $D --keep > /dev/null || exit 1
for mb in 128 512 2048 8192 65536; do
  DEFUSE_TIMING=1 DEFUSE_MPE_SCRATCH_MB=$mb $R/bin/clustermatepairs -a $D/spanning.txt -u 300 -s 30 -p 0.95 -m 5 -c $D/cl.$mb 2>&1 | grep "EM iterations" | sed "s/^/scratch_mb=$mb /"
  rm -f $D/cl.$mb
done
rm -rf $D
