# One dosplitalign chunk of the e2e chain with the library's stream trace: where a batch of many fusions with few candidates each
# spends its device time.   gpurun -- bash profiles/microbench/e2e_chunk_trace.sh [fragments] [support lo] [support hi]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_e2e; mkdir -p $O; D=/tmp/e2e_trace
python3 $R/profiles/microbench/e2e_scale.py --fragments ${1:-2000000} --support ${2:-10} ${3:-30} --no-fused --parallel 1 --out $D > $O/trace_chain.json 2> $O/trace_chain.err || { tail -20 $O/trace_chain.err; exit 1; }
C="$R/bin/dosplitalign -f $D/ref.fa -e $D/exons.txt -u 450 -s 45 -n 150 -x 150 -r $D/clusters.sc.regions -i $D/improper.0.sam -1 $D/reads.0.1.fastq -2 $D/reads.0.2.fastq -a $D/trace.split"
DEFUSE_TIMING=1 DEFUSE_DSA_INPROCESS=1 DEFUSE_DSA_STREAM_TRACE=1 $C 2> $O/chunk_trace.txt || { tail -20 $O/chunk_trace.txt; exit 1; }
grep -E "dsa_stream|batches:|main\(\)" $O/chunk_trace.txt
cd /tmp && export TMPDIR=/tmp && DEFUSE_DSA_INPROCESS=1 DEFUSE_FULL_EXIT=1 rocprofv3 --kernel-trace --stats -d $O/chunk_kt -o kt --output-format csv -- $C > $O/chunk_kt.log 2>&1 || { tail -20 $O/chunk_kt.log; exit 1; }
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_e2e"
f = glob.glob(O + "/chunk_kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(O + "/chunk_kernel_stats.txt", "w") as out:
    for r in rows[:16]:
        line = "%-64s calls %5s total_ms %9.2f avg_us %9.1f pct %s" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"])
        print(line)
        out.write(line + "\n")
PY
