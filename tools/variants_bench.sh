# profiling aid: runs bench.py against every build/variants/lib_*.so (knock-out builds of the pixel kernels)
for so in build/variants/lib_*.so; do
  echo "== $so"
  J40HIP_LIB=$PWD/$so python bench.py --batch ${BATCH:-64} --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['kernels_ms'])"
done
