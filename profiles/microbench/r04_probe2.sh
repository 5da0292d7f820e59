# One sparse and one dense chunk of the end-to-end data through dosplitalign with the stream's per-batch trace (worker as a thread, so
# that its stderr is seen), then the bench line with the round's new fields.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_probe2; mkdir -p $O; cd $R
for shape in "20 60" "100 300"; do
  tag=$(echo $shape | tr ' ' _)
  timeout -k 10 300 python profiles/microbench/e2e_scale.py --fragments 2000000 --support $shape --out /tmp/e2e_$tag --parallel 1 --json $O/e2e_2M_$tag.json > $O/e2e_2M_$tag.log 2>&1 || { tail -20 $O/e2e_2M_$tag.log; exit 1; }
  D=/tmp/e2e_$tag
  DEFUSE_DSA_INPROCESS=1 DEFUSE_DSA_STREAM_TRACE=1 DEFUSE_TIMING=1 bin/dosplitalign -f $D/ref.fa -e $D/exons.txt -u 450 -s 45 -n 150 -x 150 -r $D/clusters.sc.regions \
      -i $D/improper.0.sam -1 $D/reads.0.1.fastq -2 $D/reads.0.2.fastq -a $D/trace.align 2> $O/trace_$tag.txt || { tail $O/trace_$tag.txt; exit 1; }
  DEFUSE_TIMING=1 bin/dosplitalign -f $D/ref.fa -e $D/exons.txt -u 450 -s 45 -n 150 -x 150 -r $D/clusters.sc.regions \
      -i $D/improper.0.sam -1 $D/reads.0.1.fastq -2 $D/reads.0.2.fastq -a $D/trace2.align 2> $O/timing_$tag.txt
  cmp $D/trace.align $D/trace2.align && cmp $D/trace.align $D/split.0 || exit 1
  grep -E "dsa_stream|main\(\)|regions|fasta index|reads:" $O/trace_$tag.txt | head -14
  rm -rf $D
done
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "steps", "scaling", "sched", "hip_runtime")})
print(d["roofline"]["bound"], d["roofline"]["frac"], d["roofline"].get("hbm_frac"), d["stage_ms"])
print(json.dumps(d.get("sensitivity"), indent=1))
print(d["cpu_baseline"]["value"], d["one_shot"]["ms_per_1M_aligns"])
PY
