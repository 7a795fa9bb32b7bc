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
extern int g_gemm_variant;      // 0 auto, 1 force the 128x128 register-staged kernel, 2 never the persistent 256x256 one, 3 never the few-row form
extern int g_q2c_xcd_swizzle;
extern int g_q2c_chunk_log2;    // -1 auto; K6 corpus walk: rounds per MALL-resident chunk = 2^v (30 = one chunk = old order)
extern int g_q2c_line_log2;     // K6 walk: 2^v consecutive rounds of an XCD on adjacent clip tiles (Q2cPersistArgs::lsh)
extern int g_q2c_qsh;           // -1 auto; K6 super-tile of an XCD: 2^v query tiles x 2^(5-v) clip tiles (v = 0..4)
#else
static constexpr int g_q2c_variant = 0, g_q2c_ablation = 0, g_gemm_variant = 0, g_q2c_xcd_swizzle = 1,
                     g_q2c_chunk_log2 = -1, g_q2c_line_log2 = 0, g_q2c_qsh = -1;
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device) and per growth of the requested size -- not on
// every launch.  The attribute belongs to the function ON THE CURRENT DEVICE: a process that drives several GPUs, or asks
// for a larger dynamic LDS size later, sets it again there (a per-process flag left the second device at the 64 KiB
// default and its 138-160 KiB kernels failed to launch).  Lock-free: racing threads at worst set the same value twice.
#include <atomic>
template <auto KERN>
static bool xml_lds_attr_once(int lds_bytes) {
  constexpr int MAX_DEV = 64;
  static std::atomic<int> done[MAX_DEV];          // largest size set so far per device (zero-initialised)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (dev < 0 || dev >= MAX_DEV)
    return hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) == hipSuccess;
  if (done[dev].load(std::memory_order_acquire) >= lds_bytes) return true;
  if (hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
    return false;
  int cur = done[dev].load(std::memory_order_relaxed);
  while (cur < lds_bytes && !done[dev].compare_exchange_weak(cur, lds_bytes, std::memory_order_release)) {}
  return true;
}
