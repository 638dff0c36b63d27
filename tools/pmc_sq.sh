cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/sq1 --output-format csv -- python $R/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/sq2 --output-format csv -- python $R/bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/sq2.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/sq1 gpurun_out/sq1.txt > /dev/null; python tools/pmc_summary.py gpurun_out/sq2 gpurun_out/sq2.txt > /dev/null; rm -rf gpurun_out/sq1 gpurun_out/sq2; tail -2 gpurun_out/sq1.log | cut -c1-300
