# An ablation / diagnostic build of the library: build_var/lib_<name>.so with extra -D flags (only dsa_api.hip differs; the
# other objects are shared).  bash profiles/microbench/build_variant.sh <name> [-DFLAG ...]
name=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/build_var
H=$(cd $R && python3 -c "from defuse_amd import build; import sys; print(build.source_hash(sys.argv[1:]))" "$@")
C=$R/defuse_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -DDSA_BUILD_HASH="\"$H\"" -o $R/build_var/lib_$name.so \
  $C/dsa_api.hip $C/sc_api.hip $C/mpe_api.hip $C/la_api.hip $C/hc_api.hip $C/cov_api.hip && echo "built build_var/lib_$name.so ($H)"
