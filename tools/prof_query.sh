#!/bin/bash
# rocprofv3 kernel stats of the query-encode stage (6 passes over 10 000 queries) [+ 1024 videos of context encode with "ctx"]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q -o q -- python $R/tools/prof_query.py "$@" > /tmp/prof_q.log 2>&1
f=$(find /tmp/prof_q -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:24]:
    print("%-70s calls %5d avg_us %9.1f total_ms %8.2f" % (r["Name"][:70], int(r["Calls"]), float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
