#!/bin/bash
# Round 6's closing evidence in one GPU call, every part on the SAME commit (gpurun_out/r06/; the summaries are then copied into
# profiles/ as r06_*). Every step under a timeout.
#   1. the GPU suite, smoke()
#   2. the default bench line and the driver's (--steps 20 --warmup 5)
#   3. rocprofv3 --kernel-trace --stats (+ tools/kernel_timeline.py: who runs beside whom): one batch alone (the stages' kernels with the device to themselves), the LfGroup launch
#      alone, the pipeline with the pixels left in HBM (the stages overlapped as in the steady state)
#   4. PMC passes over that last command, one counter group per pass (FETCH_SIZE; WRITE_SIZE; two SQ groups), j40hip's kernels only
#      -> r06_pmc_traffic.json (bench.py's roofline.stages[].traffic)
#   0. (first, on the fresh box) the public API: one caller (latency), 64 and 128 callers; config 5 with the LfGroup streams on either side
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
COMMIT=$(cat $R/.commit_for_profiles 2>/dev/null || echo unknown)
echo $COMMIT > $O/commit.txt
python - <<'PY' > $O/synth.log 2>&1
import sys; sys.path.insert(0, "tests")
from streams import synth
for i in range(4): synth("vardct", 7680, 4320, 3 + 1000 * i, forward=1)
PY
P8K=$(ls $R/build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
J40HIP_API_TIMING=1 J40HIP_SERVE=0 timeout 200 $R/build/api_threads 1 8 --warm 2 $P8K > $O/api_one_thread_latency.json 2> $O/api_one_thread_latency.err
for n in 64 128; do timeout 300 $R/build/api_threads $n 8 --warm 3 --verify-every 8 $P8K > $O/api_${n}_threads.json 2> $O/api_${n}_threads.err; echo "api_$n rc=$?" >> $O/rc.txt; done
timeout 1500 python -u -m pytest tests -m gpu -q -p no:cacheprovider > $O/gputest_full.txt 2>&1; tail -n 6 $O/gputest_full.txt > $O/gputest.txt; echo "gputest rc=$?" >> $O/rc.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 600 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench_default rc=$?" >> $O/rc.txt
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench_steps20_warmup5.err; echo "bench_steps20 rc=$?" >> $O/rc.txt
# the contract clock in five fresh processes at the driver's flags, the copies' engines on each line
for i in 1 2 3 4 5; do timeout 200 python $R/bench.py --skip-sections --no-cpu-baseline --steps 20 --warmup 5 >> $O/contract_clock_five_runs.jsonl 2>> $O/contract_clock_five_runs.err; echo "clock$i rc=$?" >> $O/rc.txt; done
timeout 300 python $R/tools/modular_split_probe.py > $O/modular_split_probe.json 2> $O/modular_split_probe.err; echo "split_probe rc=$?" >> $O/rc.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_restore -- python $R/tools/restore_probe.py 3 > $O/restoration_probe.txt 2>&1 ); echo "restore_probe rc=$?" >> $O/rc.txt
cp "$(find /tmp/kt_restore -name '*kernel_stats.csv' | head -1)" $O/restoration_kernel_stats.csv 2>/dev/null; rm -rf /tmp/kt_restore
FUZZ_FLIPS=0.3 timeout 1500 python $R/tools/fuzz_parity.py 220 606 gpu > $O/fuzz_gpu.txt 2>&1; echo "fuzz_gpu rc=$?" >> $O/rc.txt
kt() { name=$1; shift; ( cd /tmp && timeout 240 env "$@" rocprofv3 --kernel-trace --stats -d /tmp/kt_$name -- python $R/tools/r05_probe.py 256 16 6 > $O/kt_$name.log 2>&1 ); echo "kt_$name rc=$?" >> $O/rc.txt
	python tools/prof_summary.py /tmp/kt_$name $O/kernel_stats_$name.txt > /dev/null 2>&1; python tools/kernel_timeline.py /tmp/kt_$name $O/timeline_$name.txt 1.0 0 > /dev/null 2>&1; rm -rf /tmp/kt_$name; grep -h '^{' $O/kt_$name.log > $O/probe_$name.json; }
kt one_batch_alone_b256 PROBE_ONLY=alone
kt lf_launch_alone_b256 PROBE_ONLY=lf_alone
kt device_output_b256 PROBE_ONLY=device
pmc() { name=$1; counters=$2; ( cd /tmp && timeout 300 env PROBE_ONLY=device rocprofv3 --pmc $counters --kernel-include-regex "j40hip" -d /tmp/pmc_$name --output-format csv -- python $R/tools/r05_probe.py 256 16 2 > $O/pmc_$name.log 2>&1 ); echo "pmc_$name rc=$?" >> $O/rc.txt
	python tools/pmc_summary.py /tmp/pmc_$name $O/pmc_$name.txt > /dev/null 2>&1; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
LFF=$(grep -h '^{' $O/pmc_fetch.log | python -c "import json,sys; print(json.loads(sys.stdin.readline())['device']['lf_frames_per_launch'])" 2>/dev/null || echo 256)
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write $O/pmc_traffic.json 256 ${LFF:-256} "$COMMIT" "PROBE_ONLY=device python tools/r05_probe.py 256 16 2" > $O/pmc_traffic.log 2>&1
rm -rf /tmp/pmc_fetch /tmp/pmc_write
pmc sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
pmc sq2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
rm -rf /tmp/pmc_sq1 /tmp/pmc_sq2
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 $R/tools/rccl_dry_run.py > $O/rccl_dry_run.json 2> $O/rccl_dry_run.err
cat $O/rc.txt; cat $O/gputest.txt; tail -n 4 $O/smoke.txt; cut -c1-700 $O/bench_default.json; echo; cut -c1-300 $O/bench_steps20_warmup5.json; echo; cat $O/probe_*.json | cut -c1-900; grep -h "k_lf_rows\|k_hf_lanes\|k_vardct_dct<3, 3\|k_vardct_special\|k_plan_place" $O/kernel_stats_*.txt | cut -c1-60,108-190; cat $O/pmc_traffic.log | cut -c1-900; tail -n 1 $O/api_one_thread_latency.json | cut -c1-400; tail -n 1 $O/rccl_dry_run.json | cut -c1-300; python -c "
import json
for l in open('$O/contract_clock_five_runs.jsonl'):
    if l.startswith('{'):
        r=json.loads(l); print('clock', r['value'], r['ms_per_step'], r['pcie']['achieved_gb_per_s'], r['pcie'].get('slow_run'), r['pcie'].get('copy_engine',{}).get('engine'))
"; cat $O/modular_split_probe.json | head -60; tail -n 3 $O/fuzz_gpu.txt; cat $O/restoration_probe.txt | tail -n 3; grep -E "k_epf|k_gaborish" $O/restoration_kernel_stats.csv | cut -c1-160; for n in 64 128; do tail -n 1 $O/api_${n}_threads.json | cut -c1-300; done
