# Round 4, the tools as a user runs them: GPU tests of the changed tools, the stream's per-batch trace inside dosplitalign, clustermatepairs +
# setcover at 50 M fragments stage by stage, the end-to-end chain at test size against the oracle chain and at 20 M fragments.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_tools; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_clustermatepairs.py tests/test_cmp_bins.py tests/test_tools.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
timeout -k 10 300 python profiles/microbench/e2e_scale.py --fragments 6000 --chrom-len 600000 --chunk 3000 --out /tmp/e2e_small --check --json $O/e2e_check.json > $O/e2e_check.log 2>&1 || { tail -30 $O/e2e_check.log; exit 1; }
grep -E "tools_equal|fragments_per_s|exactly" $O/e2e_check.json
timeout -k 10 400 python profiles/microbench/tool_throughput.py 10000 100 10 > $O/tool_throughput.txt 2>&1 || { tail -20 $O/tool_throughput.txt; exit 1; }
grep -E "best|summary|candidates/s" $O/tool_throughput.txt
timeout -k 10 600 bash profiles/microbench/cmp50_stages.sh > $O/cmp50.log 2>&1 || { tail -20 $O/cmp50.log; exit 1; }
cp gpurun_out/cmp50/timing.txt $O/cmp50_stage_timing.txt
grep -E "real|user|sys|kernel|stages overlapped|records on|read \+ bin|bin pairs on" $O/cmp50_stage_timing.txt | head -12
rm -rf /tmp/cmp50
timeout -k 10 900 python profiles/microbench/e2e_scale.py --fragments ${1:-20000000} --out /tmp/e2e --json $O/e2e.json > $O/e2e.log 2>&1 || { tail -30 $O/e2e.log; exit 1; }
python - <<PY
import json
d = json.load(open("$O/e2e.json"))
print({k: d[k] for k in ("wall_s", "stages_sum_s", "stages_over_wall", "gpu_busy_s", "fragments_per_s_end_to_end", "planted_junctions_recovered")})
for r in d["stages"]:
    print(r["stage"], r["wall_s"], r["gpu_s"])
print(d.get("dosplitalign_chunks_in_parallel"))
for k, v in d.get("fused", {}).items():
    print("fused", k, v["wall_s"], "same alignments", v["same_alignments_file"], "same final files", v.get("same_final_files"), [(r["stage"][:40], r["wall_s"]) for r in v.get("stages", [])])
PY
rm -rf /tmp/e2e /tmp/e2e_small
timeout -k 10 900 bash profiles/microbench/mpe_roofline.sh > $O/mpe_roofline.log 2>&1 || { tail -30 $O/mpe_roofline.log; exit 1; }
cp gpurun_out/mpe_roofline/roofline.json $O/mpe_roofline.json
tail -25 $O/mpe_roofline.log
