# Kernel timeline of ONE exact-rank pass (bench.py --exact-rank): everything between the last two K6 launches.
#   gpurun -- bash tools/trace_exact.sh     (writes gpurun_out/trace_exact/)
mkdir -p gpurun_out/trace_exact
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/trace_exact -o step --output-format csv -- python $R/bench.py --exact-rank --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $R/gpurun_out/trace_exact/bench.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_exact/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k6 = [i for i, r in enumerate(rows) if "q2c_persist" in r["Kernel_Name"]]
a, b = k6[-2], k6[-1]
t0 = int(rows[a]["End_Timestamp"])
prev = t0
print("kernels between the last two K6 launches (one step minus K6):")
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  gap %6.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:100]))
    prev = e
PY
tail -2 gpurun_out/trace_exact/bench.log | cut -c1-1500
rm -rf gpurun_out/trace_exact/*/  # raw traces stay off the merge-back
