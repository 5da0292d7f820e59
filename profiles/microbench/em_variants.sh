# the 5 M fragment probe for the library and for build_var/lib_<name>.so of every name given
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/em_jump; mkdir -p $O; cd $R
timeout -k 10 400 python3 profiles/microbench/em_probe.py 5000000 3 > $O/variants.txt 2>&1 || { tail -20 $O/variants.txt; exit 1; }
for v in "$@"; do
  echo "variant $v" >> $O/variants.txt
  DEFUSE_DSA_LIB=$R/build_var/lib_$v.so timeout -k 10 400 python3 profiles/microbench/em_probe.py 5000000 2 >> $O/variants.txt 2>&1 || { tail -20 $O/variants.txt; exit 1; }
done
cat $O/variants.txt
