#!/bin/bash
# where the wave EM kernel's cycles go: -DMPE_PROFILE build of the library (build_var/), config-3 probe
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
D=/tmp/cmp_scale
mkdir -p $R/build_var/defuse_amd $R/build_var/bin
C=$R/defuse_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMPE_PROFILE -o $R/build_var/defuse_amd/libdefuse_dsa.so $C/dsa_api.hip $C/sc_api.hip $C/mpe_api.hip $C/la_api.hip $C/hc_api.hip || exit 1
g++ -std=c++17 -O2 -o $R/build_var/bin/clustermatepairs $R/tools_src/clustermatepairs.cpp $R/build_var/defuse_amd/libdefuse_dsa.so '-Wl,-rpath,$ORIGIN/../defuse_amd' || exit 1
python3 $R/profiles/microbench/cmp_scale.py --fragments ${1:-5000000} --out $D --keep > /dev/null || exit 1
DEFUSE_TIMING=1 DEFUSE_MPE_SCRATCH_MB=65536 $R/build_var/bin/clustermatepairs -a $D/spanning.txt -u 300 -s 30 -p 0.95 -m 5 -c $D/cl.p 2>&1 | grep "mpe profile\|EM iterations"
rm -rf $D
