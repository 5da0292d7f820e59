# BASELINE configs[2] at full size through the two tools, stage by stage (DEFUSE_TIMING): clustermatepairs at 16 host threads,
# setcover at 8 and 16.  Intermediate files stay in /tmp on the box.
cd $GRAFT_REPO_ROOT
O=gpurun_out/cmp50; mkdir -p $O; rm -f $O/timing.txt
python profiles/microbench/cmp_scale.py --fragments ${1:-50000000} --out /tmp/cmp50 --generate-only > $O/gen.json 2>&1 || { cat $O/gen.json; exit 1; }
for th in 16; do
  echo "clustermatepairs, threads $th" >> $O/timing.txt
  ( time DEFUSE_THREADS=$th DEFUSE_TIMING=1 bin/clustermatepairs -a /tmp/cmp50/spanning.txt -c /tmp/cmp50/clusters.txt -u 300 -s 30 -p 0.95 -m 5 ) >> $O/timing.txt 2>&1 || { tail $O/timing.txt; exit 1; }
done
for th in 8 16; do
  echo "setcover, threads $th" >> $O/timing.txt
  ( time DEFUSE_THREADS=$th DEFUSE_TIMING=1 bin/setcover -c /tmp/cmp50/clusters.txt -m 5 -o /tmp/cmp50/clusters.sc ) >> $O/timing.txt 2>&1 || { tail $O/timing.txt; exit 1; }
done
ls -la /tmp/cmp50 >> $O/timing.txt
cat $O/gen.json; cat $O/timing.txt
