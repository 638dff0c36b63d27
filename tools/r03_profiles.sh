# round-3 evidence in one GPU call: the default bench line; rocprofv3 kernel stats and HBM counters (separate passes) of the timed
# region (`bench.py --skip-sections`, same pipeline, same streams)
# usage: bash tools/r03_profiles.sh [tag]
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
python $R/bench.py ${BENCH_ARGS:-} > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.err
T="--skip-sections --no-cpu-baseline --steps 4 --warmup 2"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$tag -- python $R/bench.py $T > $R/gpurun_out/kt_$tag.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$tag --output-format csv -- python $R/bench.py $T > $R/gpurun_out/pmc_fetch_$tag.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$tag --output-format csv -- python $R/bench.py $T > $R/gpurun_out/pmc_write_$tag.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/kt_$tag gpurun_out/kernel_stats_$tag.txt > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_fetch_$tag gpurun_out/pmc_fetch_$tag.txt > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_write_$tag gpurun_out/pmc_write_$tag.txt > /dev/null 2>&1
rm -rf gpurun_out/kt_$tag gpurun_out/pmc_fetch_$tag gpurun_out/pmc_write_$tag   # (raw traces: tens of MB; gpurun_out/ travels back only below 64 MiB)
cut -c1-2500 gpurun_out/bench_$tag.json; head -12 gpurun_out/kernel_stats_$tag.txt; grep -A3 "k_hf_lanes" gpurun_out/pmc_fetch_$tag.txt gpurun_out/pmc_write_$tag.txt | head
