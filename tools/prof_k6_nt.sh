#!/bin/bash
# K6 persistent kernel: default vs non-temporal clip-tile loads (ablation 3): time + FETCH_SIZE
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
RAW=/tmp/prof_k6nt; mkdir -p $RAW $R/gpurun_out
for abl in 0 3; do
  python $R/tools/bench_k6.py 10000 21793 768 --variants=4 --ablation=$abl 2>&1 | grep variant
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/a$abl -o pmc -- python $R/tools/bench_k6.py 10000 21793 768 --variants=4 --ablation=$abl > $RAW/a$abl.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$RAW/a*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "q2c" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in agg.items():
        print("%-20s %-24s per-launch avg %.6g  (%d launches)" % (f.split("/")[3], k, s / n, n))
PY
