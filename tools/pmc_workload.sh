#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of K6 for another bench workload (separate --pmc passes, --kernel-trace only):
#   gpurun -- bash tools/pmc_workload.sh c3r   -> gpurun_out/r06_<workload>_k6_traffic.txt
W=${1:-c3r}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; RAW=/tmp/pmc_$W; OUT=$R/gpurun_out; mkdir -p $RAW $OUT
CMD="python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $RAW/$c -o b -- $CMD > $RAW/$c.log 2>&1; echo "$c rc=$?"
done
grep "^{\"metric\"" $RAW/WRITE_SIZE.log | tail -1 > $RAW/line.json
python - <<PY > $OUT/r06_${W}_k6_traffic.txt
import csv, glob, collections, json
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$RAW/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "q2c_persist_kernel" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
fetch, write = (agg[c][0] / max(agg[c][1], 1) * 1024 for c in ("FETCH_SIZE", "WRITE_SIZE"))
d = json.loads(open("$RAW/line.json").read())
rc = d.get("ragged_corpus", {})
print("# bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-extras; one --pmc pass per counter; K6 per launch")
print("FETCH_SIZE reported %.3f GB (x2 gfx950 correction = %.3f GB), WRITE_SIZE %.3f GB -> traffic %.1f GB per launch" % (fetch / 1e9, 2 * fetch / 1e9, write / 1e9, (2 * fetch + write) / 1e9))
print("executed clip rows %.0f, valid %.0f; K6 avg launch %.2f ms under the counters" % (rc.get("executed_clip_rows", 0), rc.get("valid_clip_rows", 0), d["roofline"]["avg_launch_ms"]))
PY
cat $OUT/r06_${W}_k6_traffic.txt
