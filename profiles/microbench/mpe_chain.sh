#!/bin/bash
# EM kernel with the M step's serial sums loading 4, 8 or 16 elements ahead (-DMPE_CHAIN)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
D=/tmp/cmp_scale
C=$R/defuse_amd/csrc
python3 $R/profiles/microbench/cmp_scale.py --fragments ${1:-5000000} --out $D --keep > /dev/null || exit 1
for n in 4 8 16; do
  B=$R/build_var/chain$n
  mkdir -p $B/defuse_amd $B/bin
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMPE_CHAIN=$n -o $B/defuse_amd/libdefuse_dsa.so $C/dsa_api.hip $C/sc_api.hip $C/mpe_api.hip $C/la_api.hip $C/hc_api.hip || exit 1
  g++ -std=c++17 -O2 -pthread -o $B/bin/clustermatepairs $R/tools_src/clustermatepairs.cpp $B/defuse_amd/libdefuse_dsa.so '-Wl,-rpath,$ORIGIN/../defuse_amd' || exit 1
  DEFUSE_TIMING=1 $B/bin/clustermatepairs -a $D/spanning.txt -u 300 -s 30 -p 0.95 -m 5 -c $D/cl.$n 2>&1 | grep "EM iterations" | sed "s/.*kernel/chain=$n: kernel/"
  cmp -s $D/clusters.txt $D/cl.$n && echo "  identical" || echo "  DIFFERENT"
done
rm -rf $D
