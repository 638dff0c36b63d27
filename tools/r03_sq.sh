# SQ counters of the batched kernels (one pass; kernels run one at a time under --pmc)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/sq_r03 --output-format csv -- python $R/bench.py --skip-sections --no-cpu-baseline --steps 1 --warmup 1 --distinct 16 --lf-streams host --in-flight 1 > $R/gpurun_out/sq_r03.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/sq_r03 gpurun_out/sq_r03.txt > /dev/null 2>&1; rm -rf gpurun_out/sq_r03; tail -2 gpurun_out/sq_r03.log | cut -c1-200
