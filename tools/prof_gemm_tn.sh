#!/bin/bash
# PMC passes over the weight-gradient GEMM alone (TN_SHAPE=rows,N,K; default 12800,768,768), debug library so that XML_ABL
# selects the kernel (300: the 128 x 128 kernel).  Output: gpurun_out/gemm_tn_pmc_<abl>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export TN_SHAPE=${TN_SHAPE:-12800,768,768} XMLHIP_LIB=$R/tvretrieval_amd/csrc/libxmlhip_dbg.so XML_ABL=${XML_ABL:-0}
RAW=/tmp/prof_tn_$XML_ABL; rm -rf $RAW; mkdir -p $RAW $R/gpurun_out
python $R/tools/bench_gemm_tn.py
CTRS=("FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM" "TA_BUSY_sum" "TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum" "SQ_INST_LEVEL_VMEM" "SQ_INST_LEVEL_LDS")
[ -n "$TN_CTRS" ] && IFS=";" read -ra CTRS <<< "$TN_CTRS"
for c in "${CTRS[@]}"; do
  d=$(echo $c | tr ' ' '_')
  timeout 100 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $RAW/$d -o tn -- python $R/tools/bench_gemm_tn.py > $RAW/$d.log 2>&1
  echo "$d rc=$?"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("$RAW/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gemm_tn" in r["Kernel_Name"]:
            a = agg[(r["Kernel_Name"][:60], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
out = ["# TN_SHAPE=$TN_SHAPE XML_ABL=$XML_ABL ; per-launch averages (FETCH_SIZE / WRITE_SIZE in KiB as reported, uncorrected)"]
for (k, c), (s, n) in sorted(agg.items()):
    out.append("%-62s %-30s %.6g  (%d launches)" % (k, c, s / n, n))
open("$R/gpurun_out/gemm_tn_pmc_$XML_ABL.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
