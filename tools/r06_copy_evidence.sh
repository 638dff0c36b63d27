#!/bin/bash
# gpurun_out/r06/ (what tools/r06_final.sh wrote on the GPU box) -> profiles/r06_* (tracked). Run from the repo root.
set -eu
S=gpurun_out/r06; D=profiles
n=$(grep -o '[0-9]* passed' $S/gputest.txt | head -1 | cut -d' ' -f1)
rm -f $D/r06_gputest_*_passed.txt
cp $S/gputest.txt $D/r06_gputest_${n}_passed.txt
for f in api_one_thread_latency.json contract_clock_five_runs.jsonl modular_split_probe.json bench_default.json bench_steps20_warmup5.json smoke.txt \
	kernel_stats_device_output_b256.txt kernel_stats_lf_launch_alone_b256.txt kernel_stats_one_batch_alone_b256.txt \
	probe_device_output_b256.json probe_lf_launch_alone_b256.json probe_one_batch_alone_b256.json pmc_traffic.json rccl_dry_run.json \
	timeline_device_output_b256.txt timeline_one_batch_alone_b256.txt api_64_threads.json api_128_threads.json restoration_kernel_stats.csv; do
	cp $S/$f $D/r06_$f
done
tail -n 4 $S/fuzz_gpu.txt > $D/r06_fuzz_gpu.txt
tail -n 3 $S/restoration_probe.txt > $D/r06_restoration_probe.txt
grep "j40 api\|j40hip upload\|j40hip parse" $S/api_one_thread_latency.err | tail -n 24 > $D/r06_api_one_thread_phases.txt
cp $S/pmc_fetch.txt $D/r06_pmc_fetch_device_output.txt
cp $S/pmc_write.txt $D/r06_pmc_write_device_output.txt
cp $S/pmc_sq1.txt $D/r06_sq_counters_1.txt
cp $S/pmc_sq2.txt $D/r06_sq_counters_2.txt
echo "evidence of commit $(cat $S/commit.txt) copied"
