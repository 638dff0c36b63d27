# round-2 evidence in one GPU call: the default bench line, rocprofv3 kernel stats of the same command, HBM counters (separate
# passes), SQ counters (two passes), and the SQ instruction counts of the Modular section kernels (tools/k3_probe.py)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && bash tools/profile_round.sh r02 > gpurun_out/profile_round_r02.log 2>&1
cd $R && bash tools/pmc_sq.sh > gpurun_out/pmc_sq_r02.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SMEM -d $R/gpurun_out/sq_k3 --output-format csv -- python $R/tools/k3_probe.py > $R/gpurun_out/sq_k3.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/sq_k3 gpurun_out/sq_k3.txt > /dev/null 2>&1; rm -rf gpurun_out/sq_k3
tail -3 gpurun_out/profile_round_r02.log | cut -c1-600; grep -A8 "k_modular_coop" gpurun_out/sq_k3.txt | head -12
