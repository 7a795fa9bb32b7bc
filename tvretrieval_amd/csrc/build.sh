#!/bin/bash
# Builds tvretrieval_amd/csrc/libxmlhip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
SRCS="api.hip linear.hip gemm256.hip gemm256p.hip attention.hip q2c.hip q2c256.hip q2c_ring.hip q2c_persist.hip q2c_persist4.hip q2c_persist32.hip topk.hip convse.hip moment.hip postproc.hip train.hip"
mkdir -p build
pids=()
for s in $SRCS; do
  o=build/${s%.hip}.o
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ common.h -nt "$o" ] || [ gemm.h -nt "$o" ] || [ internal.h -nt "$o" ] \
     || [ ../../include/xmlhip.h -nt "$o" ]; then
    $HIPCC $FLAGS -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libxmlhip.so build/*.o
echo "built $(pwd)/libxmlhip.so"
