#!/bin/bash
# EM kernel of clustermatepairs on the config-3 probe: per-kernel times (rocprofv3) and the wave/lane split point.
# Run from the repo root on the GPU box; scratch goes to /tmp, summaries to gpurun_out/.
set -e
export TMPDIR=/tmp
D=/tmp/cmp_scale
python profiles/microbench/cmp_scale.py --fragments ${1:-5000000} --out $D --keep > gpurun_out/mpe_sweep_base.json
A="-a $D/spanning.txt -u 300 -s 30 -p 0.95 -m 5"
for w in 0 8 12 16 24 40; do
  DEFUSE_TIMING=1 DEFUSE_MPE_WAVE_MIN=$w bin/clustermatepairs $A -c $D/cl.$w 2>&1 | grep "EM iterations" | sed "s/^/wave_min=$w /"
  cmp -s $D/clusters.txt $D/cl.$w && echo "  identical" || echo "  DIFFERENT"
  rm -f $D/cl.$w
done
DEFUSE_FULL_EXIT=1 rocprofv3 --kernel-trace --stats -d gpurun_out/mpe_prof -o mpe -- bin/clustermatepairs $A -c $D/cl.prof > /dev/null 2>&1 || true
find gpurun_out/mpe_prof -name "*kernel_stats.csv" -exec cat {} \; | cut -d, -f1-8 | head -12
rm -rf $D
