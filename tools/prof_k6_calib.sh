#!/bin/bash
# Calibrates rocprofv3 FETCH_SIZE for K6's access pattern (16-B lanes, 1 KiB contiguous pieces of slice-major tiles
# through global_load_lds; the row-major predecessor -- 64-B row segments -- calibrated to the same factor 1/2):
# with ONE query tile (nq = 256) every clip tile is fetched by exactly one workgroup, so the fabric reads must equal
# the corpus bytes (nv*128*h*2).  Then the same counter at nq = 2048 (one query group: 8 workgroups share each clip
# tile through L2) and at the full nq.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
RAW=/tmp/prof_k6c; mkdir -p $RAW $R/gpurun_out
for nq in 256 2048 10000; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    tag=$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $RAW/q${nq}_$tag -o pmc -- python $R/tools/bench_k6.py $nq 21793 768 --tiled1 > $RAW/q${nq}_$tag.log 2>&1
    echo "nq=$nq $c rc=$?"
  done
done
python - <<PY > $R/gpurun_out/k6_fetch_calibration.txt
import csv, glob, collections
print("# rocprofv3 --pmc <counter> --kernel-trace -- python tools/bench_k6.py <nq> 21793 768 --tiled1   (persistent K6 on slice-major tiles, one modality, bf16)")
print("# corpus bytes = 21793*128*768*2 = %.4g" % (21793*128*768*2))
for f in sorted(glob.glob("$RAW/q*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "q2c_persist" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in agg.items():
        print("%-40s %-24s per-launch avg %.6g  (%d launches)" % (f.split("/")[3], k, s / n, n))
PY
cat $R/gpurun_out/k6_fetch_calibration.txt
