#!/bin/bash
# HBM-side traffic and achieved bandwidth of the kernels of one bench step BESIDE K6 (top-k, ConvSE, moment top-n, the query
# encoder's kernels): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each, --kernel-trace only) joined with the kernel
# durations of a --stats pass.  FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section), both counters in KiB.
#   gpurun -- bash tools/tail_traffic.sh   -> gpurun_out/r04_tail_traffic.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt_stats -o b -- $CMD > /tmp/tt_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tt_$c -o b -- $CMD > /tmp/tt_$c.log 2>&1
done
python - > $OUT/r04_tail_traffic.txt <<'PY'
import csv, glob, collections
dur = {}
for f in glob.glob("/tmp/tt_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = (float(r["AverageNs"]), int(r["Calls"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("/tmp/tt_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            a = cnt[r["Kernel_Name"]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("# bench.py --steps 3 --warmup 1 --no-extras: per-launch averages; read = FETCH_SIZE x 2 x 1 KiB (gfx950 correction), written = WRITE_SIZE x 1 KiB")
print("%-64s %6s %10s %10s %10s %9s" % ("kernel", "calls", "avg us", "read MB", "written MB", "TB/s"))
keep = ("topk_rows", "convse_kernel", "moment_topk", "attention_core_small", "modular_pool_small", "add_layernorm_vec", "gemm256p", "gather_pos", "q2c_persist", "gemm256_kernel")
rows = []
for k, (ns, calls) in dur.items():
    if not any(x in k for x in keep) or k not in cnt:
        continue
    rd = cnt[k]["FETCH_SIZE"][0] / max(cnt[k]["FETCH_SIZE"][1], 1) * 2 * 1024
    wr = cnt[k]["WRITE_SIZE"][0] / max(cnt[k]["WRITE_SIZE"][1], 1) * 1024
    rows.append((ns, k, calls, rd, wr))
for ns, k, calls, rd, wr in sorted(rows, reverse=True):
    print("%-64s %6d %10.1f %10.1f %10.1f %9.2f" % (k.split("(")[0][:64], calls, ns / 1e3, rd / 1e6, wr / 1e6, (rd + wr) / ns / 1e3))
PY
cat $OUT/r04_tail_traffic.txt
