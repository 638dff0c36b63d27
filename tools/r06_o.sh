#!/bin/bash
# round 6, call O: k_modular_tokens with LZ77 runs at distance 1 as stores of the value the wavefront just decoded -- the Modular GPU tests,
# the two-pass probe (config 1 and four more), a GPU sweep of random option mixes against the reference
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06o; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_modular_coop.py tests/test_squeeze.py -m gpu -x -q -k "modular or two_pass or lz77 or baseline or golden or corruption or squeeze or coop" ) > $O/pytest.txt 2>&1; echo "tests rc=$?" >> $O/rc.txt; tail -3 $O/pytest.txt
timeout 400 python tools/modular_split_probe.py > $O/modular_split_probe.json 2> $O/probe.err; echo "probe rc=$?" >> $O/rc.txt
FUZZ_FLIPS=0.3 timeout 900 python tools/fuzz_parity.py 160 808 gpu > $O/fuzz_gpu.txt 2>&1; echo "fuzz rc=$?" >> $O/rc.txt; tail -2 $O/fuzz_gpu.txt
cat $O/rc.txt; python - <<'PY'
import json
d=json.load(open("gpurun_out/r06o/modular_split_probe.json"))
for k,v in d["two_pass"].items(): print(k, v["ms"], v["equals_reference"], v["status"])
PY
