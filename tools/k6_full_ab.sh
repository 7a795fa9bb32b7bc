#!/bin/bash
# A/B of the mask-free K6 (headline shape, every video 128 clips) against the previous round's snapshot (_ab_r05/).
mkdir -p gpurun_out
{
for rep in 1 2 3; do
  for tree in . _ab_r05; do
    [ -d "$tree/tvretrieval_amd" ] || continue
    echo "== tree $tree"
    XML_PKG_ROOT=$PWD/$tree python tools/bench_k6_ragged.py 10000 21793 768 --full
    XML_PKG_ROOT=$PWD/$tree python tools/bench_k6_ragged.py 10895 2179 256 --full
  done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/k6_full_ab.txt
