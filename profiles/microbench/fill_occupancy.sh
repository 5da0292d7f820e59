#!/bin/bash
# k_fill_fast compiled for 3 or 4 workgroups per CU (-DDSA_FAST_WGS): registers against waves in flight
R=${GRAFT_REPO_ROOT:-$PWD}
C=$R/defuse_amd/csrc
mkdir -p $R/build_var
for n in 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDSA_FAST_WGS=$n -o $R/build_var/lib_wgs$n.so $C/dsa_api.hip $C/sc_api.hip $C/mpe_api.hip $C/la_api.hip $C/hc_api.hip || exit 1
done
cd $R && bash profiles/microbench/variant_bench.sh wgs3 wgs4
