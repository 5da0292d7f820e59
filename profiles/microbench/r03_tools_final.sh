# The tools with the round's final build: dosplitalign / evalsplitalign on 1 M candidates through files, both clustering tools
# stage by stage at 50 M fragments, and a longer randomised stress of the split-read path.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_tools; mkdir -p $O; cd $R
timeout -k 10 600 python profiles/microbench/tool_throughput.py 10000 100 > $O/tool_throughput_10k.txt 2>&1 || { tail -20 $O/tool_throughput_10k.txt; exit 1; }
tail -12 $O/tool_throughput_10k.txt
bash profiles/microbench/cmp50_stages.sh > $O/cmp50.log 2>&1 || { tail -20 $O/cmp50.log; exit 1; }
cp gpurun_out/cmp50/timing.txt $O/cmp50_stage_timing.txt
grep -E "real|kernel|^\[setcover\] (read|write|set cover)" $O/cmp50_stage_timing.txt
timeout -k 10 900 python tests/stress_dsa.py ${1:-150} 7000 > $O/stress.log 2>&1 || { tail -30 $O/stress.log; exit 1; }
tail -1 $O/stress.log
