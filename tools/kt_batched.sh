cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_batched --output-format csv -- env J40HIP_K2_BATCHED=1 python $R/bench.py --batch 128 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/kt_batched.log 2>&1
cd $R; f=$(ls gpurun_out/kt_batched/*/*kernel_stats.csv | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]: print("%-90s calls %6s avg_us %10.1f pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
