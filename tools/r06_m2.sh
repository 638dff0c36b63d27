# scratch of call M: the single-image path's phases as J40HIP_API_TIMING=1 prints them (tools/latency_probe.py)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r06m; mkdir -p $O
J40HIP_API_TIMING=1 timeout 300 python tools/latency_probe.py 5 > $O/timing_probe.json 2> $O/timing_probe.err
grep "two phases\|j40 api" $O/timing_probe.err | tail -12
