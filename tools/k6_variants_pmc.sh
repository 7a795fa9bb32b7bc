#!/bin/bash
# K6 kernel variants side by side (debug library): the production two-waves-per-SIMD persistent kernel (variant 4, one
# modality, row-major operands) against the two abandoned shapes kept in the debug build -- variant 5: FOUR waves per
# workgroup, 128 x 128 per wave in 256 AGPRs, MFMA issued from inline asm, fragment reads and DMA pieces hand-placed behind
# individual MFMAs (q2c_persist4.hip); variant 6: 32x32x16 MFMA (q2c_persist32.hip).  Times in one process, then one
# rocprofv3 --pmc pass per counter group.    gpurun -- bash tools/k6_variants_pmc.sh   -> gpurun_out/r03_k6_variants_pmc.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
export XMLHIP_LIB=$R/tvretrieval_amd/csrc/libxmlhip_dbg.so
cd /tmp && export TMPDIR=/tmp
{
echo "# python tools/bench_k6.py 10000 21793 768 --variants=4,5,6,4,5,6   (one modality, bf16, random L2-normalised operands)"
python $R/tools/bench_k6.py 10000 21793 768 --variants=4,5,6,4,5,6 2>&1 | grep -v amdgpu.ids
for v in 4 5 6; do
  for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    d=/tmp/k6v_${v}_$(echo $c | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python $R/tools/bench_k6.py 10000 21793 768 --variants=$v > $d.log 2>&1
  done
done
python - <<'PY'
import csv, glob, collections
for v in (4, 5, 6):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("/tmp/k6v_%d_*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "q2c" in r["Kernel_Name"]:
                a = agg[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print("# variant %d: per-launch averages" % v)
    for (k, c), (s, n) in sorted(agg.items()):
        print("  %-62s %-30s %.6g  (%d launches)" % (k, c, s / n, n))
PY
} > $OUT/r03_k6_variants_pmc.txt 2>&1
tail -50 $OUT/r03_k6_variants_pmc.txt
