#!/bin/bash
# rocprofv3 counter passes for the K6 micro-benchmark (run on the GPU box through gpurun).
# Raw rocprof output stays in /tmp on the box; only the per-kernel summary goes to gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
RAW=/tmp/prof_k6
OUT=$R/gpurun_out
mkdir -p $RAW $OUT
ARGS="${K6_ARGS:-10000 21793 768 --variants=2}"
i=0
for pmc in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $RAW/p$i -o pmc -- python $R/tools/bench_k6.py $ARGS > $RAW/p$i.log 2>&1
  echo "pass $i rc=$? : $(tail -2 $RAW/p$i.log | tr '\n' ' ')"
done
find $RAW -name "*.csv" | head -20
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$RAW/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "q2c" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in agg.items():
        print("%-28s per-launch avg %.6g  (%d launches)" % (k, s / n, n))
PY
