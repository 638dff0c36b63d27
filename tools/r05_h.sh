#!/bin/bash
# round 5, call H: the pixel kernels' prefetch of the next tile's block ordinals (against none, and with eight wavefronts per SIMD asked
# of the small shapes' kernels), parity first; 96 callers with the burst threshold at twelve frames per thread
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
timeout 300 python -u -m pytest tests/test_pipeline.py tests/test_gpu_parity.py -q -x -m gpu -k "batches_give or forward_encoded or all_transforms or large_transforms or batch_throughput or 4k_frame or matches_single" > $O/tests.txt 2>&1; echo "tests rc=$? $(tail -n 1 $O/tests.txt)" >> $O/rc.txt
probe() { name=$1; shift; ( timeout 150 env "$@" PROBE_ONLY=alone python tools/r05_probe.py 256 16 4 ) >> $O/probes.jsonl 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
for i in 1 2; do
probe base A=1
probe nopf J40HIP_LIB=$V/libj40hip_nopf.so
probe pfw8 J40HIP_LIB=$V/libj40hip_pfw8.so
done
P8K=$(ls build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
( timeout 200 build/api_threads 96 8 --warm 3 --verify-every 8 $P8K ) > $O/api96.json 2> $O/api96.err; echo "api96 $(tail -n 1 $O/api96.json | cut -c1-300)" >> $O/rc.txt
cat $O/rc.txt; python - <<'PY'
import json
for l in open("gpurun_out/r05h/probes.jsonl"):
    r = json.loads(l); print(r["lib"], r["alone"]["pixel_stage_ms"], r["alone"]["k_hf_lanes_ms"], r["alone"]["plan_tail_ms"])
PY
