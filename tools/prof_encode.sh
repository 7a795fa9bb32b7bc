#!/bin/bash
# rocprofv3 kernel stats of the corpus encode alone (tools/prof_encode.py --videos 8192 --reps 3 --no-queries): per-kernel table
# with the share of the encode and microseconds per 2 048-video batch.   gpurun -- bash tools/prof_encode.sh > gpurun_out/x.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_e
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o e -- python $R/tools/prof_encode.py --videos 8192 --reps 3 --no-queries > /tmp/prof_e.log 2>&1
tail -1 /tmp/prof_e.log
f=$(find /tmp/prof_e -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/encode_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
nb = (3 * 8192 + 2048) / 2048.0      # batches of 2048 videos in the run (3 reps + the warm-up batch)
print("total kernel time %.1f ms; per 2048-video batch %.2f ms" % (tot/1e6, tot/1e6/nb))
for r in rows[:22]:
    print("%-86s calls %5d avg_us %8.1f  %5.1f %%  us/batch %8.1f" % (r["Name"][:86], int(r["Calls"]), float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot, float(r["TotalDurationNs"])/1e3/nb))
PY
