# the EM batch on the 5 M fragment probe, the cut points of the shares varied:  gpurun -- bash profiles/microbench/em_heavy.sh
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04_em
python3 profiles/microbench/em_probe.py 5000000 1 > /dev/null 2>&1
for f in ${EM_SHARES:-0 0.3 0.1,0.4 0.05,0.2,0.5 0.15,0.45 0.2,0.3,0.4}; do
  echo "DEFUSE_MPE_SHARES=$f"
  DEFUSE_MPE_SHARES=$f python3 profiles/microbench/em_probe.py 5000000 3 2>&1 | tail -3
done | tee gpurun_out/r04_em/shares.txt
