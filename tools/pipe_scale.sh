for t in 8 16 32 64 128; do echo "threads $t"; python tools/pipe_bench.py 128 $t 32 8 2>&1 | grep "rep 1"; done
echo "no mallopt, 64 threads"; J40HIP_KEEP_MALLOC_DEFAULTS=1 python tools/pipe_bench.py 128 64 32 8 2>&1 | grep "rep 1"
