#!/bin/bash
# EM kernel time of clustermatepairs against the share of the wave fits that run from LDS, and the wave/lane split
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
D=/tmp/cmp_scale
python3 $R/profiles/microbench/cmp_scale.py --fragments ${1:-5000000} --out $D --keep > /dev/null || exit 1
A="-a $D/spanning.txt -u 300 -s 30 -p 0.95 -m 5"
for sh in 0 0.25 0.4 0.5 0.6 0.75 1; do
  DEFUSE_TIMING=1 DEFUSE_MPE_SCRATCH_MB=65536 DEFUSE_MPE_LDS_SHARE=$sh $R/bin/clustermatepairs $A -c $D/cl.s 2>&1 | grep "EM iterations" | sed "s/.*kernel/lds_share=$sh one chunk: kernel/"
  cmp -s $D/clusters.txt $D/cl.s || echo "  DIFFERENT"
done
for w in 8 16 64; do
  DEFUSE_TIMING=1 DEFUSE_MPE_SCRATCH_MB=65536 DEFUSE_MPE_WAVE_MIN=$w $R/bin/clustermatepairs $A -c $D/cl.s 2>&1 | grep "EM iterations" | sed "s/.*kernel/wave_min=$w one chunk: kernel/"
done
DEFUSE_TIMING=1 $R/bin/clustermatepairs $A -c $D/cl.s 2>&1 | grep "EM iterations" | sed "s/.*kernel/defaults: kernel/"
rm -rf $D
