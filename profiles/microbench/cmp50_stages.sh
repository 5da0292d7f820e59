cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cmp50
python profiles/microbench/cmp_scale.py --fragments 50000000 --out /tmp/cmp50 --generate-only > gpurun_out/cmp50/gen.json 2>&1
for th in 8 16; do
  echo "threads $th" >> gpurun_out/cmp50/timing.txt
  ( time DEFUSE_THREADS=$th DEFUSE_TIMING=1 bin/clustermatepairs -a /tmp/cmp50/spanning.txt -c /tmp/cmp50/clusters.txt -u 300 -s 30 -p 0.95 -m 5 ) >> gpurun_out/cmp50/timing.txt 2>&1
done
( time DEFUSE_TIMING=1 bin/setcover -c /tmp/cmp50/clusters.txt -m 5 -o /tmp/cmp50/clusters.sc ) >> gpurun_out/cmp50/timing.txt 2>&1
cat gpurun_out/cmp50/gen.json; cat gpurun_out/cmp50/timing.txt
