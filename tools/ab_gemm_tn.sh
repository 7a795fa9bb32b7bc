#!/bin/bash
# A/B of the weight-gradient GEMM kernels (debug library; XML_DEBUG=1 bash tvretrieval_amd/csrc/build.sh first): XML_ABL=300 the
# 128 x 128 kernel everywhere, 299 the XCD-partitioned kernel wherever it can run, 0 the product dispatch, 301 no output,
# 304 / 306 L2- / L1-hot rows, 307 no loads in the loop; PROBE=1 (library built with XML_DEBUG_EXTRA=-DXML_TN_PROBE): stage timers.
cd "$(dirname "$0")/.."
[ -n "$SKIP_TEST" ] || python -m pytest tests/test_gpu_train.py -q -x -k "weight_gradient_gemm" 2>&1 | tail -5
echo "== product library"; python tools/bench_gemm_tn.py 2>&1 | grep "gemm_tn" | sed "s/transposes.*//"
for a in ${ABLS:-300 0 301}; do
  echo "== XML_ABL=$a"
  XMLHIP_LIB=$PWD/tvretrieval_amd/csrc/libxmlhip_dbg.so XML_ABL=$a python tools/bench_gemm_tn.py 2>&1 | grep "gemm_tn\|rror\|fault" | sed 's/transposes.*//'
done
if [ -n "$PROBE" ]; then XMLHIP_LIB=$PWD/tvretrieval_amd/csrc/libxmlhip_dbg.so XML_ABL=308 TN_SHAPE=12800,768,768 python tools/bench_gemm_tn.py 2>&1 | tail -10; fi
