export XMLHIP_LIB=$PWD/tvretrieval_amd/csrc/libxmlhip_dbg.so
python tools/probe_gemm.py 2>&1 | grep -v amdgpu.ids
