#!/bin/bash
# Builds tvretrieval_amd/csrc/libxmlhip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   bash build.sh            product library: stateless dispatch, no experiment kernels
#   XML_DEBUG=1 bash build.sh   additionally libxmlhip_dbg.so (-DXML_DEBUG_VARIANTS): kernel-variant / ablation switches
#                               (xml_debug_*) and the abandoned K6 variants, which live OUTSIDE the product sources in
#                               tools/microbench/k6_variants/ (q2c256.hip, q2c_persist4/32.hip) -- for tools/ only
#                               (XMLHIP_LIB=.../libxmlhip_dbg.so python tools/bench_k6.py ...)
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
SRCS="api.hip linear.hip gemm256.hip gemm256p.hip attention.hip attention_train.hip q2c.hip q2c_ring.hip q2c_persist.hip topk.hip convse.hip moment.hip decode.hip conv1d.hip postproc.hip train.hip gemm_tn.hip index_build.hip ingest.hip collectives.hip exact.hip split16.hip loss_tail.hip"
DBG_DIR=../../tools/microbench/k6_variants
DBG_SRCS="$DBG_DIR/q2c256.hip $DBG_DIR/q2c_persist4.hip $DBG_DIR/q2c_persist32.hip"

build() {   # $1 = object dir, $2 = extra flags, $3 = output, $4.. = sources
  local dir=$1 extra=$2 out=$3; shift 3
  mkdir -p "$dir"
  local pids=() objs=()
  for s in "$@"; do
    [ -f "$s" ] || continue
    local b=$(basename "$s")
    local o=$dir/${b%.hip}.o
    objs+=("$o")
    if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ common.h -nt "$o" ] || [ gemm.h -nt "$o" ] || [ internal.h -nt "$o" ] \
       || [ debug.h -nt "$o" ] || [ l2norm.h -nt "$o" ] || [ ../../include/xmlhip.h -nt "$o" ]; then
      $HIPCC $FLAGS $extra -I. -c "$s" -o "$o" &
      pids+=($!)
    fi
  done
  for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}" ${XML_LINK_LIBS:-}
  echo "built $(pwd)/$out"
}

build build "" libxmlhip.so $SRCS
# libxmlpy.so: the reference's nested-list result format built against the CPython API (host only; ctypes.PyDLL)
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
if [ ! -f libxmlpy.so ] || [ pylists.c -nt libxmlpy.so ]; then
  gcc -O2 -fPIC -shared -Wall -I"$PYINC" pylists.c -o libxmlpy.so
  echo "built $(pwd)/libxmlpy.so"
fi
if [ "${XML_DEBUG:-0}" = "1" ]; then
  build build_dbg "-DXML_DEBUG_VARIANTS ${XML_DEBUG_EXTRA:-}" libxmlhip_dbg.so $SRCS $DBG_SRCS      # (XML_DEBUG_EXTRA=-DXML_TN_PROBE: stage timers of gemm_tn)
fi
