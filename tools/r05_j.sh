#!/bin/bash
# round 5, call J: the pipeline's two regimes with the pixels left in HBM -- the threads' own timings (J40HIP_ASYNC_TIMING) of four runs in a row
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05j; mkdir -p $O
for i in 1 2 3 4; do
( timeout 150 env PROBE_ONLY=device J40HIP_ASYNC_TIMING=1 python tools/r05_probe.py 256 16 12 ) >> $O/probes.jsonl 2> $O/run$i.err; echo "run$i rc=$?" >> $O/rc.txt
cat /sys/fs/cgroup/cpu.stat > $O/cpustat$i.txt 2>/dev/null
rocm-smi --showmeminfo vram --showclocks > $O/smi$i.txt 2>&1
done
( timeout 150 env PROBE_ONLY=device J40HIP_ASYNC_TIMING=1 J40HIP_LF_CAP=512 python tools/r05_probe.py 256 16 12 ) >> $O/probes.jsonl 2> $O/run5_lfcap512.err
( timeout 150 env PROBE_ONLY=device J40HIP_ASYNC_TIMING=1 J40HIP_LF_FLIGHT_FRAMES=256 python tools/r05_probe.py 256 16 12 ) >> $O/probes.jsonl 2> $O/run6_flight256.err
cat $O/rc.txt; python - <<'PY'
import json
for l in open("gpurun_out/r05j/probes.jsonl"):
    r = json.loads(l); d = r["device"]; print(r["env"], d["ms_per_step"], d["lf_launches"], d["lf_frames_per_launch"], d["lf_kernel_ms"], d["k_hf_lanes_ms"], d["pixel_stage_ms"])
PY
for i in 1 2 3 4; do echo == run$i; grep "gpu thread\|lf launch" $O/run$i.err | tail -n 6 | cut -c1-260; grep -h "nr_throttled\|throttled_usec" $O/cpustat$i.txt | tr '\n' ' '; echo; done
grep "gpu thread" $O/run5_lfcap512.err $O/run6_flight256.err | cut -c1-260
