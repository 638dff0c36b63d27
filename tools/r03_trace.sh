# kernel timeline of the pipeline bench (csv of every dispatch: name, stream/queue, start, end)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -- python $R/bench.py --skip-sections --steps ${STEPS:-8} --warmup ${WARM:-5} --distinct 16 --no-cpu-baseline --lf-streams ${MODE:-host} > $R/gpurun_out/trace.log 2>&1
cd $R
f=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "dispatches; columns:", list(rows[0].keys()))
out = open("gpurun_out/trace_small.csv", "w")
out.write("name,queue,stream,start,end,grid,wg\n")
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    n = r["Kernel_Name"].split("(")[0][:60]
    out.write("%s,%s,%s,%d,%d,%s,%s\n" % (n, r.get("Queue_Id", ""), r.get("Stream_Id", ""), int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
out.close()
PY
rm -rf gpurun_out/trace
tail -3 gpurun_out/trace.log | cut -c1-300
