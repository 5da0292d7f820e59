# clustermatepairs (and setcover) at 50 M fragments with and without the huge-page mappings of large blocks (DEFUSE_NO_HUGE_BLOCKS=1 = without):
# real / user / sys.   gpurun -- bash profiles/microbench/cmp50_thp.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/cmp50_thp; mkdir -p $O; rm -f $O/timing.txt
echo "THP: $(cat /sys/kernel/mm/transparent_hugepage/enabled) defrag: $(cat /sys/kernel/mm/transparent_hugepage/defrag) shmem: $(cat /sys/kernel/mm/transparent_hugepage/shmem_enabled 2>/dev/null)" | tee -a $O/timing.txt
python profiles/microbench/cmp_scale.py --fragments ${1:-50000000} --out /tmp/cmp50 --generate-only > $O/gen.json 2>&1 || { cat $O/gen.json; exit 1; }
for tun in ${CMP50_ENVS:-DEFUSE_NO_HUGE_BLOCKS=1 DEFUSE_THREADS=16 DEFUSE_NO_HUGE_BLOCKS=1 DEFUSE_THREADS=16}; do
  echo "clustermatepairs, env $tun" >> $O/timing.txt
  rm -f /tmp/cmp50/clusters.txt
  ( time env $tun DEFUSE_THREADS=16 DEFUSE_TIMING=1 bin/clustermatepairs -a /tmp/cmp50/spanning.txt -c /tmp/cmp50/clusters.txt -u 300 -s 30 -p 0.95 -m 5 ) 2>&1 | grep -E "real|user|sys|problems \+ clustering|read \+ bin" >> $O/timing.txt
done
for tun in ${CMP50_SC_ENVS:-}; do
  echo "setcover, env $tun" >> $O/timing.txt
  ( time env $tun DEFUSE_THREADS=16 DEFUSE_TIMING=1 bin/setcover -c /tmp/cmp50/clusters.txt -m 5 -o /tmp/cmp50/clusters.sc ) 2>&1 | grep -E "real|user|sys" >> $O/timing.txt
done
rm -rf /tmp/cmp50
cat $O/timing.txt
