# round-3 evidence in one GPU call: the default bench line; rocprofv3 kernel stats and HBM counters (separate passes) of the timed
# region (`bench.py --skip-sections`, same pipeline, same streams); kernel stats of a coefficient-domain batch with the 128/256-sized
# transforms in the mix
# usage: bash tools/r03_profiles.sh [tag]
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
python $R/bench.py ${BENCH_ARGS:-} > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.err
T="--skip-sections --no-cpu-baseline --steps 6 --warmup 2"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$tag -- python $R/bench.py $T > $R/gpurun_out/kt_$tag.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$tag --output-format csv -- python $R/bench.py --skip-sections --no-cpu-baseline --steps 2 --warmup 2 > $R/gpurun_out/pmc_fetch_$tag.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$tag --output-format csv -- python $R/bench.py --skip-sections --no-cpu-baseline --steps 2 --warmup 2 > $R/gpurun_out/pmc_write_$tag.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_maxlog8_$tag -- python $R/bench.py --skip-sections --no-cpu-baseline --stream coefficient --maxlog 8 --batch 64 --pipe-batch 64 --distinct 8 --steps 3 --warmup 1 > $R/gpurun_out/kt_maxlog8_$tag.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/kt_$tag gpurun_out/kernel_stats_$tag.txt > /dev/null 2>&1
python tools/prof_summary.py gpurun_out/kt_maxlog8_$tag gpurun_out/kernel_stats_maxlog8_$tag.txt > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_fetch_$tag gpurun_out/pmc_fetch_$tag.txt > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_write_$tag gpurun_out/pmc_write_$tag.txt > /dev/null 2>&1
rm -rf gpurun_out/kt_$tag gpurun_out/kt_maxlog8_$tag gpurun_out/pmc_fetch_$tag gpurun_out/pmc_write_$tag   # (raw traces: tens of MB; gpurun_out/ travels back only below 64 MiB)
python - "$tag" <<'PY'
import json, re, sys
tag = sys.argv[1]
def counter(path, kernel, name):
    lines = open(path).read().split("\n")
    for i, l in enumerate(lines):
        if l.startswith(kernel):
            for m in lines[i + 1:i + 12]:
                if name in m:
                    return float(re.search(r"avg=([0-9.e+]+)", m).group(1))
    return None
f = counter("gpurun_out/pmc_fetch_%s.txt" % tag, "j40hip::k_hf_lanes", "FETCH_SIZE")
w = counter("gpurun_out/pmc_write_%s.txt" % tag, "j40hip::k_hf_lanes", "WRITE_SIZE")
if f and w:
    json.dump({"kernel": "k_hf_lanes", "stream": "forward", "frames_per_launch": 256, "frame": "7680x4320, tools/jxlsynth forward=1, seeds 3 + 1000 i",
               "fetch_size_kb": f, "write_size_kb": w, "fetch_correction": 2.0,
               "source": "profiles/%s_pmc_fetch_timed_region.txt, profiles/%s_pmc_write_timed_region.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --skip-sections`, per k_hf_lanes launch; KB of 1024 B; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md (128-B requests tallied at 64 B) -- calibrated there for wide coalesced reads only, so for this kernel's 4-byte per-lane reads it is an upper bound; WRITE_SIZE as reported)" % (tag, tag)},
              open("gpurun_out/%s_pmc_traffic.json" % tag, "w"))
    print("traffic", f, w)
PY
cut -c1-2200 gpurun_out/bench_$tag.json; head -14 gpurun_out/kernel_stats_$tag.txt; grep -i "large\|dct<6\|dct<5, 6\|dct<6, 5" gpurun_out/kernel_stats_maxlog8_$tag.txt | head
