#!/bin/bash
# round 5, call M: (1) how many LfGroup workgroups may sit on the compute units beside the entropy decoder's (99 KB of a CU's 160 KB
# of LDS each; an LfGroup workgroup takes up to 48): flights of 1024 frames (the default, up to four such workgroups per CU), of 256
# (one), one flight of 512 (one); (2) where the 8x8 special transforms' time goes: the instrumented build's phases
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05m; mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
probe() { name=$1; shift; ( timeout 150 env "$@" python tools/r05_probe.py 256 16 12 ) >> $O/probes.jsonl 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
probe device PROBE_ONLY=device
probe device_flight256 PROBE_ONLY=device J40HIP_LF_FLIGHT_FRAMES=256
probe device_one_flight512 PROBE_ONLY=device J40HIP_LF_FLIGHT_FRAMES=512 J40HIP_LF_FLIGHTS=1
probe device_flight256_lds30 PROBE_ONLY=device J40HIP_LF_FLIGHT_FRAMES=256 J40HIP_LF_ROWS_LDS_KB=30
probe device_b PROBE_ONLY=device
probe device_flight256_b PROBE_ONLY=device J40HIP_LF_FLIGHT_FRAMES=256
( timeout 200 env PROBE_K2_PHASES=1 J40HIP_LIB=$V/libj40hip_phases.so python tools/stages_alone_probe.py 256 3 8 ) > $O/k2_phases.jsonl 2> $O/k2_phases.err; echo "k2_phases rc=$?" >> $O/rc.txt
cat $O/rc.txt
python - <<'PY'
import json
for l in open("gpurun_out/r05m/probes.jsonl"):
    r = json.loads(l); d = r["device"]
    print(r["env"], "step", d["ms_per_step"], "k1", d["k_hf_lanes_ms"], "k2", d["pixel_stage_ms"], "plan", d["plan_tail_ms"], "lf", d["lf_kernel_ms"], d["lf_launches"], d["lf_frames_per_launch"], d["lf_waves_per_launch"])
PY
cut -c1-600 $O/k2_phases.jsonl
