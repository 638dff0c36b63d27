#!/bin/bash
# round 4: where a tile's time goes in k_vardct_dct (instrumented build, -DJ40_K2_PHASES), stages alone on the device. Writes gpurun_out/r04PH/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04PH; mkdir -p $O
PROBE_K2_PHASES=1 J40HIP_LIB=$GRAFT_REPO_ROOT/build/variants/libj40hip_k2phases.so timeout 90 python tools/stages_alone_probe.py 256 3 8 > $O/phases.jsonl 2> $O/phases.err; echo "rc=$?"
cat $O/phases.jsonl; tail -3 $O/phases.err
