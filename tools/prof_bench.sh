#!/bin/bash
# rocprofv3 runs of the exact bench.py command: (1) --kernel-trace --stats, (2)(3) separate PMC passes for HBM traffic.
# Raw traces stay in /tmp on the GPU box; only the small summaries are written to gpurun_out/ (copied to profiles/).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
RAW=/tmp/prof_bench; OUT=$R/gpurun_out; mkdir -p $RAW $OUT
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o bench -- $CMD > $RAW/stats.log 2>&1
echo "stats rc=$?"; tail -1 $RAW/stats.log | cut -c1-400
f=$(find $RAW/stats -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $RAW/$c -o bench -- $CMD > $RAW/$c.log 2>&1
  echo "$c rc=$?"
done
python - <<PY > $OUT/bench_pmc_traffic.txt
import csv, glob, collections
print("# rocprofv3 --pmc <counter> --kernel-trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline  (one pass per counter)")
print("# per-launch averages; FETCH_SIZE/WRITE_SIZE are in KiB as reported (uncorrected)")
for f in sorted(glob.glob("$RAW/*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        a = agg[(r["Kernel_Name"][:60], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        print("%-62s %-12s per-launch avg %.6g  (%d launches)" % (k, c, s / n, n))
PY
cat $OUT/bench_pmc_traffic.txt | head -30
