set -x
export XMLHIP_LIB=$PWD/tvretrieval_amd/csrc/libxmlhip_dbg.so
python tools/bench_k6_chunk.py --values=20 --lines=0,1,2,3 2>&1 | grep -v amdgpu.ids
mkdir -p gpurun_out/line
cd /tmp && export TMPDIR=/tmp
for ln in 0 2; do
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/line/w$ln -o w$ln --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_k6_chunk.py --values=20 --lines=$ln --nq=10000 > $GRAFT_REPO_ROOT/gpurun_out/line/w$ln.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for ln in (0, 2):
    for f in glob.glob("gpurun_out/line/w%d/**/*counter_collection.csv" % ln, recursive=True):
        tot, n = 0.0, 0
        for r in csv.DictReader(open(f)):
            if "q2c_persist" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE":
                tot += float(r["Counter_Value"]); n += 1
        print("line", ln, f.split("/")[-1], "launches", n, "WRITE_SIZE per launch (KiB?)", tot / max(n, 1))
PY
