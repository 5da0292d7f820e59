# clustermatepairs at 50 M fragments three times on a fresh box (the tool alone: the first thing the box does), then setcover twice.
cd $GRAFT_REPO_ROOT
O=gpurun_out/cmp50x; mkdir -p $O; rm -f $O/timing.txt
python profiles/microbench/cmp_scale.py --fragments 50000000 --out /tmp/cmp50 --generate-only > $O/gen.json 2>&1 || { cat $O/gen.json; exit 1; }
for rep in 1 2 3; do
  echo "clustermatepairs, threads 16, run $rep" >> $O/timing.txt
  ( time DEFUSE_THREADS=16 DEFUSE_TIMING=1 bin/clustermatepairs -a /tmp/cmp50/spanning.txt -c /tmp/cmp50/clusters.txt -u 300 -s 30 -p 0.95 -m 5 ) >> $O/timing.txt 2>&1 || { tail $O/timing.txt; exit 1; }
done
for th in 8 16; do
  echo "setcover, threads $th" >> $O/timing.txt
  ( time DEFUSE_THREADS=$th DEFUSE_TIMING=1 bin/setcover -c /tmp/cmp50/clusters.txt -m 5 -o /tmp/cmp50/clusters.sc ) >> $O/timing.txt 2>&1 || { tail $O/timing.txt; exit 1; }
done
grep -E "run |real|sys|kernel|stages overlapped|bin pairs on|read \+ bin|records on|parsed|setcover, th" $O/timing.txt
