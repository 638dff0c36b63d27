#!/bin/bash
# round 5, call E: the event ring against none in the steady state (12 steps, pixels left in HBM), then the whole bench line once
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05e; mkdir -p $O
V=$GRAFT_REPO_ROOT/build/variants
probe() { name=$1; shift; ( timeout 200 env "$@" python tools/r05_probe.py 256 16 12 ) >> $O/probes.jsonl 2>> $O/$name.err; echo "$name rc=$?" >> $O/rc.txt; }
probe ev8_dev PROBE_ONLY=device
probe ev0_dev J40HIP_LIB=$V/libj40hip_ev0.so PROBE_ONLY=device
probe ev8_dev_b PROBE_ONLY=device
probe ev0_dev_b J40HIP_LIB=$V/libj40hip_ev0.so PROBE_ONLY=device
timeout 400 python bench.py --steps 8 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; cat $O/probes.jsonl; tail -n 5 $O/bench.err; cut -c1-6000 $O/bench.json
