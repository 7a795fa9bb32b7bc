#!/bin/bash
# Per-kernel times (and, with PMC=1, HBM traffic) of ONE pass at the reference's as-trained shape (bench.py --workload tvr_val).
#   gpurun -- bash tools/prof_tvr_val.sh   -> gpurun_out/r05_tvr_val_kernel_stats.csv [, r05_tvr_val_traffic.txt]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload tvr_val --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
rm -rf /tmp/tv_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tv_stats -o b -- $CMD > /tmp/tv_stats.log 2>&1
tail -1 /tmp/tv_stats.log | cut -c1-600
f=$(find /tmp/tv_stats -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/${TAG:-r05}_tvr_val_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/${TAG:-r05}_tvr_val_kernel_stats.csv")))
for r in rows[:22]:
    print("%-70s calls %5s avg %9.1f us  total %8.2f ms  %5s%%" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
if [ "${PMC:-0}" = "1" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tv_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tv_$c -o b -- $CMD > /tmp/tv_$c.log 2>&1
  done
  python - > $OUT/${TAG:-r05}_tvr_val_traffic.txt <<'PY'
import csv, glob, collections
dur = {}
for f in glob.glob("/tmp/tv_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = (float(r["AverageNs"]), int(r["Calls"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("/tmp/tv_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            a = cnt[r["Kernel_Name"]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("# bench.py --workload tvr_val: per-launch averages; read = FETCH_SIZE x 2 x 1 KiB (gfx950 correction), written = WRITE_SIZE x 1 KiB")
print("%-64s %6s %10s %10s %10s %9s" % ("kernel", "calls", "avg us", "read MB", "written MB", "TB/s"))
rows = []
for k, (ns, calls) in dur.items():
    if k not in cnt or ns < 20000:
        continue
    rd = cnt[k]["FETCH_SIZE"][0] / max(cnt[k]["FETCH_SIZE"][1], 1) * 2 * 1024
    wr = cnt[k]["WRITE_SIZE"][0] / max(cnt[k]["WRITE_SIZE"][1], 1) * 1024
    rows.append((ns, k, calls, rd, wr))
for ns, k, calls, rd, wr in sorted(rows, reverse=True):
    print("%-64s %6d %10.1f %10.1f %10.1f %9.2f" % (k.split("(")[0][:64], calls, ns / 1e3, rd / 1e6, wr / 1e6, (rd + wr) / ns / 1e3))
PY
  cat $OUT/${TAG:-r05}_tvr_val_traffic.txt
fi
