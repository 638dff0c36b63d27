#!/bin/bash
# round 5, call L: k_plan_place in two parts (the serial walk kept to positions, the records 64 at a time) against the walk that does
# everything (J40HIP_PLAN_PLACE_FORM=1); the LfGroup launch at the head of a burst waiting for its batch's worth
# (J40HIP_LF_WAIT_BURST=1: the old rule); parity first
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05l; mkdir -p $O
timeout 400 python -u -m pytest tests/test_device_stages.py tests/test_pipeline.py -q -x -m gpu -k "not config5 and not large_transforms" > $O/tests.txt 2>&1; echo "tests rc=$? $(tail -n 1 $O/tests.txt)" >> $O/rc.txt
probe() { name=$1; shift; ( timeout 150 env "$@" python tools/r05_probe.py 256 16 6 ) >> $O/probes.jsonl 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
probe alone PROBE_ONLY=alone
probe alone_walk PROBE_ONLY=alone J40HIP_PLAN_PLACE_FORM=1
probe device PROBE_ONLY=device
probe device_old_lf_wait PROBE_ONLY=device J40HIP_LF_WAIT_BURST=1
probe device_walk PROBE_ONLY=device J40HIP_PLAN_PLACE_FORM=1
probe device_12 PROBE_ONLY=device PROBE_STEPS=12
( cd /tmp && timeout 240 env PROBE_ONLY=device J40HIP_ASYNC_TIMING=1 rocprofv3 --kernel-trace --stats -d /tmp/kt_dev -- python $GRAFT_REPO_ROOT/tools/r05_probe.py 256 16 6 > $O/kt_device.log 2> $O/kt_device.err ); echo "kt_device rc=$?" >> $O/rc.txt
python tools/kernel_timeline.py /tmp/kt_dev $O/timeline_device_output.txt 1.0 0 > /dev/null 2> $O/timeline.err
python tools/prof_summary.py /tmp/kt_dev $O/kernel_stats_device_output.txt > /dev/null 2>&1
cat $O/rc.txt
python - <<'PY'
import json
for l in open("gpurun_out/r05l/probes.jsonl"):
    r = json.loads(l)
    for k in ("alone", "device"):
        if k in r: d = r[k]; print(r["lib"], r["env"], k, "k1", d["k_hf_lanes_ms"], "k2", d["pixel_stage_ms"], "plan", d["plan_tail_ms"], "lf", d.get("lf_kernel_ms"), "step", d.get("ms_per_step"))
PY
grep "k_plan_place\|k_hf_lanes\|k_lf_rows" $O/kernel_stats_device_output.txt | cut -c1-60,108-190
grep -v "^#" $O/timeline_device_output.txt | awk '$2>=20.0' | cut -c1-70 | head -60
tail -n 32 $O/timeline_device_output.txt
