# clean per-stage times of the batched kernels: one batch in flight, LfGroup streams on the host threads (nothing else on the GPU)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 250 python bench.py --skip-sections --steps ${STEPS:-4} --warmup 2 --distinct 16 --no-cpu-baseline --lf-streams host --in-flight 1 ${EXTRA:-} > gpurun_out/k_$1.json 2> gpurun_out/k_$1.err
python -c "import json; d=json.load(open('gpurun_out/k_$1.json')); p=d['pipeline']; print('$1', 'plan+tail', p['lf_streams_plan_tail_ms_per_launch'], 'K1', p['entropy_ms_per_launch'], 'K2', p['pixel_kernels_ms_per_launch'], 'step', d['ms_per_step'])"
