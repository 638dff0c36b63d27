#!/bin/bash
# round 5, call K: (1) k_hf_lanes with the coefficient symbols on a straight path of their own -- parity, then the launch alone, beside
# the build without it, with the event rings, with the rings and the refill as selects, and with the lighter wavefront of a SIMD at a
# lower priority; (2) the pipeline with the pixels left in HBM as a kernel timeline (who runs beside whom) with the launching
# thread's own timings and what the device memory cache did meanwhile
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05k; mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
timeout 300 python -u -m pytest tests/test_device_stages.py tests/test_pipeline.py -q -x -m gpu -k "not config5 and not large_transforms" > $O/tests.txt 2>&1; echo "tests rc=$? $(tail -n 1 $O/tests.txt)" >> $O/rc.txt
timeout 200 env J40HIP_LIB=$V/libj40hip_ring8_selects.so python -u -m pytest tests/test_device_stages.py -q -x -m gpu -k "not config5 and not large_transforms and not lftree" > $O/tests_ring8_selects.txt 2>&1; echo "tests_ring8_selects rc=$? $(tail -n 1 $O/tests_ring8_selects.txt)" >> $O/rc.txt
probe() { name=$1; shift; ( timeout 150 env "$@" python tools/r05_probe.py 256 16 6 ) >> $O/probes.jsonl 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
probe alone_straight PROBE_ONLY=alone
probe alone_nostraight PROBE_ONLY=alone J40HIP_LIB=$V/libj40hip_nostraight.so
probe alone_ring8 PROBE_ONLY=alone J40HIP_LIB=$V/libj40hip_ring8.so
probe alone_ring8_selects PROBE_ONLY=alone J40HIP_LIB=$V/libj40hip_ring8_selects.so
probe alone_rank_prio PROBE_ONLY=alone J40HIP_K1_RANK_PRIO=1
( cd /tmp && timeout 240 env PROBE_ONLY=device J40HIP_ASYNC_TIMING=1 rocprofv3 --kernel-trace --stats -d /tmp/kt_dev -- python $GRAFT_REPO_ROOT/tools/r05_probe.py 256 16 6 > $O/kt_device.log 2> $O/kt_device.err ); echo "kt_device rc=$?" >> $O/rc.txt
python tools/kernel_timeline.py /tmp/kt_dev $O/timeline_device_output.txt 1.0 0 > /dev/null 2> $O/timeline.err
python tools/prof_summary.py /tmp/kt_dev $O/kernel_stats_device_output.txt > /dev/null 2>&1
grep -h '^{' $O/kt_device.log >> $O/probes.jsonl
probe device_nostraight PROBE_ONLY=device J40HIP_LIB=$V/libj40hip_nostraight.so
probe device_ring8_selects PROBE_ONLY=device J40HIP_LIB=$V/libj40hip_ring8_selects.so
cat $O/rc.txt
python - <<'PY'
import json
for l in open("gpurun_out/r05k/probes.jsonl"):
    r = json.loads(l)
    for k in ("alone", "device"):
        if k in r: d = r[k]; print(r["lib"], r["env"], k, "k1", d["k_hf_lanes_ms"], "k2", d["pixel_stage_ms"], "plan", d["plan_tail_ms"], "lf", d.get("lf_kernel_ms"), "step", d.get("ms_per_step"))
PY
grep -h "batch launch\|gpu thread" $O/kt_device.err | tail -n 8 | cut -c1-420
tail -n 30 $O/timeline_device_output.txt
