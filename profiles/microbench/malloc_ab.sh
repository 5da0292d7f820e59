# clustermatepairs and setcover at full size with the allocator left alone (DEFUSE_MALLOC_DEFAULT=1) and with freed memory kept
# (the default), alternating on one box; wall time and the shell's user / system times.
cd $GRAFT_REPO_ROOT
O=gpurun_out/malloc_ab; mkdir -p $O; rm -f $O/*.txt
python profiles/microbench/cmp_scale.py --fragments ${1:-50000000} --out /tmp/cmp50 --generate-only > $O/gen.json 2>&1 || { cat $O/gen.json; exit 1; }
run() {   # label, command...
  local label=$1; shift
  local t0=$(date +%s.%N)
  ( time "$@" > /dev/null 2> $O/stderr_last.txt ) 2> $O/time_last.txt || { cat $O/stderr_last.txt; exit 1; }
  local t1=$(date +%s.%N)
  echo "$label: $(python3 -c "print('%.2f' % ($t1 - $t0))") s wall; $(grep -E 'user|sys' $O/time_last.txt | tr '\n' ' ')" >> $O/wall.txt
}
for rep in 1 2; do
  for mode in default kept; do
    export DEFUSE_THREADS=16
    if [ $mode = default ]; then export DEFUSE_MALLOC_DEFAULT=1; else unset DEFUSE_MALLOC_DEFAULT; fi
    rm -f /tmp/cmp50/clusters.txt /tmp/cmp50/clusters.sc
    run "clustermatepairs $mode rep $rep" bin/clustermatepairs -a /tmp/cmp50/spanning.txt -c /tmp/cmp50/clusters.txt -u 300 -s 30 -p 0.95 -m 5
    run "setcover $mode rep $rep" bin/setcover -c /tmp/cmp50/clusters.txt -m 5 -o /tmp/cmp50/clusters.sc
    md5sum /tmp/cmp50/clusters.txt /tmp/cmp50/clusters.sc | awk '{print $1}' | tr '\n' ' ' >> $O/wall.txt; echo >> $O/wall.txt
  done
done
cat $O/wall.txt
