#!/bin/bash
# host stages of clustermatepairs against the number of host threads (DEFUSE_THREADS) on the config-3 probe
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
D=/tmp/cmp_scale
python3 $R/profiles/microbench/cmp_scale.py --fragments ${1:-5000000} --out $D --keep > /dev/null || exit 1
nproc
for t in 1 2 4 8 12 16; do
  t0=$(date +%s.%N)
  DEFUSE_THREADS=$t DEFUSE_TIMING=1 $R/bin/clustermatepairs -a $D/spanning.txt -u 300 -s 30 -p 0.95 -m 5 -c $D/cl.$t 2>&1 | grep "clustermatepairs\]" | grep -v "lines per\|EM iter" | sed "s/\[clustermatepairs\]//" | tr '\n' ';'
  t1=$(date +%s.%N)
  echo " threads=$t wall $(echo "$t1 - $t0" | bc) s"
  cmp -s $D/clusters.txt $D/cl.$t && echo "  identical" || echo "  DIFFERENT"
  rm -f $D/cl.$t
done
rm -rf $D
