# Kernel timeline of ONE graphed training step (tools/bench_train.py --graph): adam_update to adam_update, per stream.
#   gpurun -- bash tools/trace_train.sh     (writes gpurun_out/trace_train.txt)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/trace_train
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/trace_train -o step --output-format csv -- python $R/tools/bench_train.py --graph > $R/gpurun_out/trace_train/bench.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/trace_train/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adam_update" in r["Kernel_Name"]]
a, b = ad[-2], ad[-1]
t0 = int(rows[a]["End_Timestamp"])
qs = {}
print("one graphed step, adam_update to adam_update: %.1f us, %d kernels" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3, b - a))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    print("%8.1f us  q%d  dur %7.1f  %s" % ((s - t0) / 1e3, q, (e - s) / 1e3, r["Kernel_Name"][:84]))
    k = agg[r["Kernel_Name"][:60]]; k[0] += 1; k[1] += (e - s) / 1e3
print("--- per kernel")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-62s %3d  %8.1f us" % (k, n, t))
PY
tail -1 gpurun_out/trace_train/bench.log | cut -c1-200
rm -rf gpurun_out/trace_train/*/
