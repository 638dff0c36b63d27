# pipeline bench over worker-thread counts / LF modes (host CPU quota study)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for t in ${THREADS:-12 14 16 20}; do for m in ${MODES:-host}; do
  timeout 250 python bench.py --skip-sections --steps ${STEPS:-10} --warmup 3 --distinct 16 --no-cpu-baseline --lf-streams $m --host-threads $t > gpurun_out/t_${t}_$m.json 2> gpurun_out/t_${t}_$m.err
  python -c "import json; d=json.load(open('gpurun_out/t_${t}_$m.json')); p=d['pipeline']; print('threads $t', '$m', d['value'], d['ms_per_step'], 'host_stage', p['host_stage_ms_per_frame'], 'dev frames', p['lf_streams_on_device_frames'], 'k1', p['entropy_ms_per_launch'], 'k2', p['pixel_kernels_ms_per_launch'])"
done; done
