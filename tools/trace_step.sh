# Kernel timeline of ONE bench.py step (c3): everything between the last two K6 launches, with the idle gaps.
#   gpurun -- bash tools/trace_step.sh     (writes gpurun_out/trace/)
mkdir -p gpurun_out/trace
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/trace -o step --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $R/gpurun_out/trace/bench.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k6 = [i for i, r in enumerate(rows) if "q2c_persist" in r["Kernel_Name"]]
a, b = k6[-2], k6[-1]
t0 = int(rows[a]["End_Timestamp"])
prev = t0
print("kernels between the last two K6 launches (one step minus K6):")
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  gap %6.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev = e
PY
tail -2 gpurun_out/trace/bench.log
