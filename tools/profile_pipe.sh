# rocprofv3 evidence for the TIMED REGION of the default bench only (--skip-sections: every k_hf_lanes launch is one of the
# 256-frame launches the bench line's roofline object describes): kernel stats, then FETCH_SIZE and WRITE_SIZE in separate passes
# usage: bash tools/profile_pipe.sh <tag>
tag=${1:-run}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --skip-sections"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ktp_$tag -- $CMD > $R/gpurun_out/ktp_$tag.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmcp_fetch_$tag --output-format csv -- $CMD > $R/gpurun_out/pmcp_fetch_$tag.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmcp_write_$tag --output-format csv -- $CMD > $R/gpurun_out/pmcp_write_$tag.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/ktp_$tag gpurun_out/kernel_stats_pipe_$tag.txt > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmcp_fetch_$tag gpurun_out/pmc_fetch_pipe_$tag.txt > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmcp_write_$tag gpurun_out/pmc_write_pipe_$tag.txt > /dev/null 2>&1
rm -rf gpurun_out/ktp_$tag gpurun_out/pmcp_fetch_$tag gpurun_out/pmcp_write_$tag
tail -1 gpurun_out/ktp_$tag.log | cut -c1-400; head -6 gpurun_out/kernel_stats_pipe_$tag.txt | cut -c1-170; grep -A1 k_hf_lanes gpurun_out/pmc_fetch_pipe_$tag.txt gpurun_out/pmc_write_pipe_$tag.txt
