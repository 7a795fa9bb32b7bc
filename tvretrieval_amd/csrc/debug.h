// Experiment switches (kernel variants, ablations) exist ONLY in the -DXML_DEBUG_VARIANTS build (libxmlhip_dbg.so, used by
// tools/ for A/B measurements).  In the product library they are compile-time constants: every `g_* == N` branch folds
// away, no ablation kernel is instantiated, no process-global state is read on a launch (the C ABI is stateless and
// re-entrant per stream, include/xmlhip.h).
#pragma once
#include <hip/hip_runtime.h>

#ifdef XML_DEBUG_VARIANTS
extern int g_q2c_variant;       // 0 auto, 1: 128x128 register-staged, 2: 256x256 LDS-DMA double buffer, 3: ring, 4: persistent,
                                // 5 / 6: the abandoned 4-wave and 32x32-MFMA persistent kernels (q2c_persist4/32.hip)
extern int g_q2c_ablation;      // per-kernel ablation id (see the ABL template parameters)
extern int g_gemm_variant;      // 0 auto, 1 force the 128x128 register-staged kernel, 2 never the persistent 256x256 one
extern int g_q2c_xcd_swizzle;
extern int g_q2c_chunk_log2;    // -1 auto; K6 corpus walk: rounds per MALL-resident chunk = 2^v (30 = one chunk = old order)
#else
static constexpr int g_q2c_variant = 0, g_q2c_ablation = 0, g_gemm_variant = 0, g_q2c_xcd_swizzle = 1,
                     g_q2c_chunk_log2 = -1;
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize once per kernel (thread-safe function-local static), not on every launch
template <auto KERN>
static bool xml_lds_attr_once(int lds_bytes) {
  static const bool ok =
      hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) == hipSuccess;
  return ok;
}
