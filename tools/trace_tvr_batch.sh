# Kernel timeline of ONE 50-query batch at the as-trained shape (tools/bench_tvr_val.py, eager chain).
#   gpurun -- bash tools/trace_tvr_batch.sh     (writes gpurun_out/trace_tvr/)
mkdir -p gpurun_out/trace_tvr
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/trace_tvr -o step --output-format csv -- python $R/tools/bench_tvr_val.py 50 > $R/gpurun_out/trace_tvr/bench.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_tvr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k6 = [i for i, r in enumerate(rows) if "q2c_persist" in r["Kernel_Name"]]
a, b = k6[-2], k6[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev = t0
print("one eager 50-query batch, K6 launch to K6 launch:")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  gap %6.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:100]))
    prev = e
PY
tail -1 gpurun_out/trace_tvr/bench.log | cut -c1-900
rm -rf gpurun_out/trace_tvr/*/
