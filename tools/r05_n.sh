#!/bin/bash
# round 5, call N: the 8x8 special transforms as three kernels (DctSelect 1-3, 12-13, AFV), 16 blocks per workgroup, eight (AFV: five)
# wavefronts per SIMD -- parity of the pixel path, then the pixel stage alone and in the pipeline beside the geometry it replaces
# (sp32: 32 blocks per workgroup, registers left alone, i.e. three workgroups per compute unit)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05n; mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
timeout 500 python -u -m pytest tests/test_gpu_parity.py tests/test_forward_streams.py tests/test_pipeline.py -q -x -m gpu -k "not 16384 and not config5 and not baseline_config" > $O/tests.txt 2>&1; echo "tests rc=$? $(tail -n 1 $O/tests.txt)" >> $O/rc.txt
probe() { name=$1; shift; ( timeout 150 env "$@" python tools/r05_probe.py 256 16 12 ) >> $O/probes.jsonl 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
probe alone PROBE_ONLY=alone
probe alone_sp32 PROBE_ONLY=alone J40HIP_LIB=$V/libj40hip_sp32.so
probe alone_sp16w6 PROBE_ONLY=alone J40HIP_LIB=$V/libj40hip_sp16w6.so
probe device PROBE_ONLY=device
probe device_sp32 PROBE_ONLY=device J40HIP_LIB=$V/libj40hip_sp32.so
probe device_sp16w6 PROBE_ONLY=device J40HIP_LIB=$V/libj40hip_sp16w6.so
( cd /tmp && timeout 240 env PROBE_ONLY=alone rocprofv3 --kernel-trace --stats -d /tmp/kt_alone -- python $GRAFT_REPO_ROOT/tools/r05_probe.py 256 16 4 > $O/kt_alone.log 2> $O/kt_alone.err ); echo "kt_alone rc=$?" >> $O/rc.txt
python tools/kernel_timeline.py /tmp/kt_alone $O/timeline_one_batch_alone.txt 0.5 0 > /dev/null 2> $O/timeline.err
python tools/prof_summary.py /tmp/kt_alone $O/kernel_stats_one_batch_alone.txt > /dev/null 2>&1
( timeout 200 env PROBE_K2_PHASES=1 J40HIP_LIB=$V/libj40hip_phases.so python tools/stages_alone_probe.py 256 3 8 ) > $O/k2_phases.jsonl 2> $O/k2_phases.err; echo "k2_phases rc=$?" >> $O/rc.txt
cat $O/rc.txt
python - <<'PY'
import json
for l in open("gpurun_out/r05n/probes.jsonl"):
    r = json.loads(l)
    for k in ("alone", "device"):
        if k in r: d = r[k]; print(r["lib"], r["env"], k, "k1", d["k_hf_lanes_ms"], "k2", d["pixel_stage_ms"], "plan", d["plan_tail_ms"], "lf", d.get("lf_kernel_ms"), "step", d.get("ms_per_step"))
PY
cut -c1-60,108-190 $O/kernel_stats_one_batch_alone.txt | head -16
grep -v "^#" $O/timeline_one_batch_alone.txt | awk '$1>400 && $2>=1.0' | cut -c1-75 | tail -n 45
head -n 3 $O/k2_phases.jsonl | cut -c1-500
