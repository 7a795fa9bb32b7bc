#!/bin/bash
# A/B of K6 on the real TVR clip counts: this tree's packed image against a snapshot of the previous round's package
# (_ab_r05/, built before the change; not tracked).  Same box, same process order repeated twice.
mkdir -p gpurun_out
{
for rep in 1 2; do
  for tree in . _ab_r05; do
    [ -d "$tree/tvretrieval_amd" ] || continue
    echo "== tree $tree, headline shape (10000 x 21793, H = 768)"
    XML_PKG_ROOT=$PWD/$tree python tools/bench_k6_ragged.py 10000 21793 768
    echo "== tree $tree, as-trained shape (10895 x 2179, H = 256, max 100 clips)"
    XML_PKG_ROOT=$PWD/$tree python tools/bench_k6_ragged.py 10895 2179 256 100
  done
done
} 2>&1 | tee gpurun_out/k6_ragged_ab.txt
