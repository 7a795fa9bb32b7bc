#!/bin/bash
# Measurement artefacts of the bench command, one GPU call (ROUND=r03 bash tools/measure_k6.sh; default tag r03):
#   (1) rocprofv3 --kernel-trace --stats of `python bench.py --steps 5 --warmup 2 --no-cpu-baseline`  -> r02_bench_kernel_stats.csv
#   (2) separate --pmc passes (no trace domains beside --kernel-trace): FETCH_SIZE, WRITE_SIZE, and
#       SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES + GRBM_GUI_ACTIVE                                    -> r02_bench_pmc.txt
#   (3) r02_k6_traffic.json: K6 bytes per launch (FETCH x2 per the gfx950 note of MI355X_MICROARCH.md + WRITE) stamped with
#       the sha256 of the K6 sources it was measured on -- bench.py refuses the file when the sources have changed.
# Raw traces stay in /tmp; summaries go to gpurun_out/ (copy them to profiles/).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${ROUND:-r03}
RAW=/tmp/prof_$TAG; OUT=$R/gpurun_out; mkdir -p $RAW $OUT
# (--no-extras: the headline pass only -- the extra legs of the default run would be traced / counted once per PMC pass)
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
CMD_S="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o bench -- $CMD > $RAW/stats.log 2>&1
echo "stats rc=$?"; grep "^{\"metric\"" $RAW/stats.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json.log
f=$(find $RAW/stats -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/${TAG}_bench_kernel_stats.csv
# FETCH_SIZE / WRITE_SIZE: PASSES (default 3) separate passes each -- the figure moves with how closely the workgroups that
# share a tile stay together (130 vs 158 GB for one walk in round 5): the json carries every pass, the MEDIAN is what bench.py
# prints as roofline.traffic, next to the spread
PASSES=${PASSES:-3}
for i in $(seq 1 $PASSES); do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $RAW/${c}_$i -o bench -- $CMD_S > $RAW/${c}_$i.log 2>&1
    echo "${c}_$i rc=$?"
  done
done
c="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; d=$(echo $c | tr ' ' '_')
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $RAW/$d -o bench -- $CMD_S > $RAW/$d.log 2>&1
echo "$d rc=$?"
python - <<PY
import csv, glob, collections, hashlib, json, os
R = "$R"; RAW = "$RAW"; OUT = "$OUT"; TAG = "$TAG"
agg = collections.defaultdict(lambda: [0.0, 0])
per_pass = collections.defaultdict(list)       # counter -> K6 per-launch average of every pass
for f in sorted(glob.glob(RAW + "/*/**/*counter_collection.csv", recursive=True)):
    one = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        a = agg[(r["Kernel_Name"][:70], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
        if "q2c_persist_kernel" in r["Kernel_Name"]:
            b = one[r["Counter_Name"]]; b[0] += float(r["Counter_Value"]); b[1] += 1
    for c, (s_, n_) in one.items():
        per_pass[c].append(s_ / n_)
lines = ["# rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras (one pass per group)",
         "# per-launch averages; FETCH_SIZE / WRITE_SIZE in KiB as reported (uncorrected)"]
for (k, c), (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    lines.append("%-72s %-26s per-launch avg %.6g  (%d launches)" % (k, c, s / n, n))
k6 = {c: s / n for (k, c), (s, n) in agg.items() if "q2c_persist_kernel" in k}
if k6:
    h = hashlib.sha256()
    for f in ("q2c_persist.hip", "common.h"):
        h.update(open(os.path.join(R, "tvretrieval_amd", "csrc", f), "rb").read())
    med = lambda v: sorted(v)[len(v) // 2] if v else 0.0      # noqa: E731
    fetches, writes = [x * 1024 for x in per_pass.get("FETCH_SIZE", [])], [x * 1024 for x in per_pass.get("WRITE_SIZE", [])]
    fetch, write = med(fetches), med(writes)
    totals = sorted(2 * f_ + write for f_ in fetches)          # (the writes do not move: 3.0 GB every time)
    rec = dict(kernel="q2c_persist_kernel", kernel_source_sha256=h.hexdigest(),
               fetch_bytes_reported=fetch, write_bytes_reported=write,
               traffic_bytes_per_launch=2 * fetch + write,
               traffic_passes=dict(n=len(fetches), fetch_bytes_reported=fetches, write_bytes_reported=writes,
                                   min=totals[0] if totals else None, max=totals[-1] if totals else None,
                                   what="per-launch averages of separate rocprofv3 --pmc passes; traffic_bytes_per_launch is "
                                        "2 x median(FETCH) + median(WRITE)"),
               correction="FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section; "
                          "re-calibrated on this access pattern in profiles/r01_k6_fetch_calibration_tiled.txt), WRITE_SIZE as is",
               command="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras (c3, 1 GPU), %d --pmc passes per counter" % len(fetches))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in k6 and "GRBM_GUI_ACTIVE" in k6:
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE over its 8 XCDs (per-launch
        # value / 8 / launch time = the ~1.9 GHz shader clock)
        cyc = k6["GRBM_GUI_ACTIVE"] / 8.0
        rec["mfma_busy_fraction"] = k6["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)
        rec["shader_cycles_per_launch"] = cyc
        lines.append("# K6: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = %.4f" % rec["mfma_busy_fraction"])
    json.dump(rec, open(OUT + "/" + TAG + "_k6_traffic.json", "w"), indent=1)
open(OUT + "/" + TAG + "_bench_pmc.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:24]))
PY
