#!/bin/bash
# round 4, third GPU call: whole suite on four-byte events + queued K1; serving gather time; host-to-host variants; the full default bench
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
python - > $O/synth.log 2>&1 <<'PY'
import sys, concurrent.futures
sys.path.insert(0, "tests")
from streams import synth
with concurrent.futures.ThreadPoolExecutor(16) as ex:
    list(ex.map(lambda i: synth("vardct", 7680, 4320, 3 + 1000 * i, forward=1), range(64)))
PY
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
P8K=$(ls build/streams/vardct_7680_4320_*forward-1.jxl | head -4 | tr '\n' ' ')
for cfg in "64 64 6 20" "64 64 6 40" "64 64 6 70" "64 64 6 0" "128 64 6 40" "16 64 6 40" "256 128 6 40"; do set -- $cfg
  J40HIP_SERVE_BATCH=$2 J40HIP_SERVE_IN_FLIGHT=$3 J40HIP_SERVE_WAIT_MS=$4 timeout 300 ./build/api_threads $1 8 --warm 3 $P8K > $O/api_t$1_b$2_f$3_w$4.json 2> $O/api_t$1_b$2_f$3_w$4.err
done
for cfg in "host 256 2" "auto 256 2" "device 256 2" "host 128 2"; do set -- $cfg
  timeout 600 python bench.py --skip-sections --no-cpu-baseline --steps 12 --warmup 2 --lf-streams $1 --pipe-batch $2 --in-flight $3 > $O/bench_$1_$2_$3.json 2> $O/bench_$1_$2_$3.err
done
rocm-smi --showmeminfo vram > $O/mem_before_full.txt 2>&1
timeout 1200 python bench.py --lf-streams auto > $O/bench_full.json 2> $O/bench_full.err
ls -la $O
