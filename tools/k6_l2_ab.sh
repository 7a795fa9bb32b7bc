#!/bin/bash
# K6 tile-walk A/B on ONE box: times of every configuration in one process, then FETCH_SIZE / WRITE_SIZE per configuration
# (one rocprofv3 --pmc pass each, --kernel-trace only).  -> gpurun_out/r05_k6_l2_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
export XMLHIP_LIB=$R/tvretrieval_amd/csrc/libxmlhip_dbg.so
cd /tmp && export TMPDIR=/tmp
{
echo "# times (one process, one box)"
python $R/tools/k6_l2_ab.py 2>/dev/null
echo "# fabric traffic per launch: FETCH_SIZE x 2 x 1 KiB (gfx950 correction) + WRITE_SIZE x 1 KiB"
for cfg in "default(qsh=3)" "qsh=2" "qsh=4" "chunk=2" "qsh=2,chunk=2"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/k6ab; K6_ONLY="$cfg" rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/k6ab -o b -- python $R/tools/k6_l2_ab.py > /tmp/k6ab.log 2>&1
    python - "$cfg" $c <<'PY'
import csv, glob, sys
tot = n = 0
for f in glob.glob("/tmp/k6ab/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "q2c_persist" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]:
            tot += float(r["Counter_Value"]); n += 1
mult = 2 if sys.argv[2] == "FETCH_SIZE" else 1
print("%-16s %-10s %8.1f GB per launch (%d launches)" % (sys.argv[1], sys.argv[2], tot / max(n, 1) * 1024 * mult / 1e9, n))
PY
  done
done
} | tee $OUT/r05_k6_l2_ab.txt
