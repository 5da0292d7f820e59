# An ablation / diagnostic build of the library: build_var/lib_<name>.so with extra flags on every file (dsa_api.hip keeps the
# scheduler flag of the product build, defuse_amd/build.py).  bash profiles/microbench/build_variant.sh <name> [-DFLAG ...]
name=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/build_var
cd $R && python3 - "$name" "$@" <<'PY'
import sys
from defuse_amd import build
name, flags = sys.argv[1], sys.argv[2:]
out = build.compile_lib(build.ROOT + "/build_var/lib_%s.so" % name, flags)
print("built build_var/lib_%s.so (%s)" % (name, build.source_hash(flags)))
PY
