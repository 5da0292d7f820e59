# setcover before (build_var/setcover_old, the round-2 tool) and after on the same box and the same 141 M line cluster file,
# alternating, DEFUSE_TIMING stage lines of the last repetition of either.
cd $GRAFT_REPO_ROOT
O=gpurun_out/setcover_ab; mkdir -p $O; rm -f $O/*.txt
python profiles/microbench/cmp_scale.py --fragments ${1:-50000000} --out /tmp/cmp50 --generate-only > $O/gen.json 2>&1 || { cat $O/gen.json; exit 1; }
DEFUSE_THREADS=16 bin/clustermatepairs -a /tmp/cmp50/spanning.txt -c /tmp/cmp50/clusters.txt -u 300 -s 30 -p 0.95 -m 5 > /dev/null 2>&1 || exit 1
rm /tmp/cmp50/spanning.txt
for rep in 1 2 3; do
  for which in old new; do
    T=bin/setcover; [ $which = old ] && T=build_var/setcover_old
    for th in 8 16; do
      t0=$(date +%s.%N)
      DEFUSE_THREADS=$th DEFUSE_TIMING=1 $T -c /tmp/cmp50/clusters.txt -m 5 -o /tmp/cmp50/clusters.$which.sc > /dev/null 2> $O/stages_${which}_$th.txt || { cat $O/stages_${which}_$th.txt; exit 1; }
      t1=$(date +%s.%N)
      echo "$which threads $th rep $rep: $(python3 -c "print('%.2f' % ($t1 - $t0))") s wall" >> $O/wall.txt
    done
  done
done
cmp /tmp/cmp50/clusters.old.sc /tmp/cmp50/clusters.new.sc && echo "outputs identical ($(stat -c %s /tmp/cmp50/clusters.new.sc) bytes, input $(stat -c %s /tmp/cmp50/clusters.txt) bytes)" >> $O/wall.txt
cat $O/wall.txt; for f in $O/stages_*; do echo $f; cat $f; done
