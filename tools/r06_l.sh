#!/bin/bash
# round 6, call L (second run): the latency entropy kernel (k_hf_entropy_fast, hf_uni_dev.h) -- J40_UNI_SLOAD / _SMUL / _PREV each alone against
# none (first run: 17.55 -> 17.30 / 16.95 / 16.77, all three 16.24 ms), then J40_UNI_FLOW on top (variant libraries from tools/build_variant.sh),
# then the GPU parity suite of the single-image path
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06l; mkdir -p $O
for rep in 1 2; do
for v in ${VARIANTS:-base noflow all}; do
	lib=$GRAFT_REPO_ROOT/build/variants/libj40hip_uni_$v.so; [ $v = all ] && lib=$GRAFT_REPO_ROOT/build/libj40hip.so
	( timeout 300 env J40HIP_LIB=$lib python tools/latency_probe.py 7 ) >> $O/latency2_$v.jsonl 2>> $O/probe.err; echo "latency $v rc=$?" >> $O/rc.txt
done
done
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_forward_streams.py -m gpu -x -q ) > $O/pytest.txt 2>&1; echo "tests rc=$?" >> $O/rc.txt
tail -3 $O/pytest.txt
cat $O/rc.txt; for v in ${VARIANTS:-base noflow all}; do echo $v; cut -c1-500 $O/latency2_$v.jsonl; done
