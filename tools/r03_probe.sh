# round-3 probe: pipeline bench lines for the LF-stream modes with the host-stage breakdown, and rocprofv3 kernel stats of the host mode
R=${GRAFT_REPO_ROOT:-$PWD}
export J40HIP_ASYNC_TIMING=1
cd $R
for m in ${MODES:-host auto}; do
  timeout 250 python bench.py --skip-sections --steps ${STEPS:-8} --warmup ${WARM:-2} --distinct 16 --no-cpu-baseline --lf-streams $m > gpurun_out/p_$m.json 2> gpurun_out/p_$m.err
  python -c "import json; d=json.load(open('gpurun_out/p_$m.json')); print('$m', d['value'], d['ms_per_step'], {k: v for k, v in d['pipeline'].items() if k != 'note'})"
  grep "host stage" gpurun_out/p_$m.err | head -3
done
if [ -n "$PROF" ]; then
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_p -- python $R/bench.py --skip-sections --steps 3 --warmup 1 --distinct 16 --no-cpu-baseline --lf-streams $PROF > $R/gpurun_out/kt_p.log 2>&1
cd $R && python tools/prof_summary.py gpurun_out/kt_p gpurun_out/kernel_stats_p_$PROF.txt | head -40; rm -rf gpurun_out/kt_p
fi
