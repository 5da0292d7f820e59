# after `gpurun -- bash profiles/microbench/r04_final.sh`: copy the run's files from gpurun_out/r04_final into profiles/r04 (run here, in the repo)
O=gpurun_out/r04_final; P=profiles/r04
for f in bench_r04.json bench_config4_shape.json bench_gloo2_rehearsal.json bench_kernel_stats.csv pmc_traffic.json pmc_sq.json bench_kernel_stats_config4.csv pmc_traffic_config4.json pmc_sq_config4.json fill_mix.json; do cp $O/$f $P/$f; done
tail -3 $O/tests.log > $P/gpu_tests_tail.txt
cp $O/stress_dsa.log $P/stress_dsa_60_rounds.txt
cp $O/stress_tools.log $P/stress_tools_20_rounds.txt
python3 -c "
import json
d = json.loads(open('$P/bench_r04.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['library_source_hash'], d['roofline']['traffic'], d['roofline']['traffic_over_algorithmic'])"
