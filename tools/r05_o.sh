#!/bin/bash
# round 5, call O: the pixel kernels' prologue two tiles ahead (records and block_events entries copied to LDS by global_load_lds_dword
# while the tile before is worked on) -- parity of the pixel path first, then the stage alone and in the pipeline beside the build
# without it (J40_K2_AHEAD=0: the ordinals prefetched into registers, as before), then the instrumented build's phases
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05o; mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
timeout 600 python -u -m pytest tests/test_gpu_parity.py tests/test_forward_streams.py tests/test_pipeline.py tests/test_device_stages.py -q -x -m gpu -k "not 16384 and not config5 and not baseline_config" > $O/tests.txt 2>&1; echo "tests rc=$? $(tail -n 1 $O/tests.txt)" >> $O/rc.txt
probe() { name=$1; shift; ( timeout 150 env "$@" python tools/r05_probe.py 256 16 12 ) >> $O/probes.jsonl 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
probe alone PROBE_ONLY=alone
probe alone_noahead PROBE_ONLY=alone J40HIP_LIB=$V/libj40hip_noahead.so
probe device PROBE_ONLY=device
probe device_noahead PROBE_ONLY=device J40HIP_LIB=$V/libj40hip_noahead.so
probe alone_b PROBE_ONLY=alone
probe device_b PROBE_ONLY=device
( timeout 200 env PROBE_K2_PHASES=1 J40HIP_LIB=$V/libj40hip_phases.so python tools/stages_alone_probe.py 256 3 8 ) > $O/k2_phases.jsonl 2> $O/k2_phases.err; echo "k2_phases rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -n 3 $O/tests.txt
python - <<'PY'
import json
for l in open("gpurun_out/r05o/probes.jsonl"):
    r = json.loads(l)
    for k in ("alone", "device"):
        if k in r: d = r[k]; print(r["lib"], r["env"], k, "k1", d["k_hf_lanes_ms"], "k2", d["pixel_stage_ms"], "plan", d["plan_tail_ms"], "lf", d.get("lf_kernel_ms"), "step", d.get("ms_per_step"))
PY
head -n 6 $O/k2_phases.jsonl | cut -c1-420
