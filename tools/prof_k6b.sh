#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
RAW=/tmp/prof_k6b; mkdir -p $RAW
ARGS="${K6_ARGS:-10000 21793 768 --variants=4}"
i=0
for pmc in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_WAVES SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $RAW/p$i -o pmc -- python $R/tools/bench_k6.py $ARGS > $RAW/p$i.log 2>&1
  echo "pass $i rc=$?"
done
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
