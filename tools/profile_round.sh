# refreshes the evidence under gpurun_out/ for the default bench: JSON line, rocprofv3 kernel stats, HBM counters (separate passes)
# usage: bash tools/profile_round.sh <tag>
tag=${1:-run}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$tag -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/kt_$tag.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$tag --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch_$tag.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$tag --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write_$tag.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/kt_$tag gpurun_out/kernel_stats_$tag.txt > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_fetch_$tag gpurun_out/pmc_fetch_$tag.txt > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_write_$tag gpurun_out/pmc_write_$tag.txt > /dev/null 2>&1
rm -rf gpurun_out/kt_$tag gpurun_out/pmc_fetch_$tag gpurun_out/pmc_write_$tag   # (raw traces: tens of MB; gpurun_out/ travels back only below 64 MiB)
cut -c1-1800 gpurun_out/bench_$tag.json; head -8 gpurun_out/kernel_stats_$tag.txt
