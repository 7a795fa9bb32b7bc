#!/bin/bash
# Corpus-encode A/B between package trees on ONE box, interleaved:  gpurun -- bash tools/ab_encode.sh _ab_r05 _ab_cur .
# (each tree: its own tvretrieval_amd/ + built libxmlhip.so; tools/bench_encode_ab.py of THIS tree drives all of them)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for t in "$@"; do
    p=$(cd $R/$t && pwd)
    XML_PKG_ROOT=$p XMLHIP_LIB=$p/tvretrieval_amd/csrc/libxmlhip.so python $R/tools/bench_encode_ab.py 8192 2>/dev/null | tail -1
  done
done
