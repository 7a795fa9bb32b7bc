// Split-f16 operand preparation: f32-grade values on the 16-bit MFMA pipe (exact-rank mode, include/xmlhip.h
// "Exact-rank mode on the 16-bit pipe").
//
// The reference computes in f32; v_mfma_f32_16x16x4_f32 runs at 1/16 of the f16 / bf16 MFMA rate, and the
// three stages that have to reproduce f32 scores -- query encoder, candidate re-score, ConvSE -- sat on it.  An f32 value
// x is carried instead as two halves  x S = hi + lo  (S a power of two, hi = rn_f16(x S), lo = rn_f16(x S - hi);
// |x S - hi - lo| <= 2^-22 |x S|) and a dot product as  hi.hi + lo.hi + hi.lo  with f32 accumulation: three f16 MFMAs
// per 32 k instead of eight f32 ones, i.e. 16/3 of the f32 rate, representation error 4 f32 ulps, dropped term
// |lo||lo| <= 2^-22 |x||y|.  Two layouts:
//   ROWS  (xml_split_f16_rows)   resident operands of the gathered-pair kernels (re-score, ConvSE) and their query rows:
//         per 32 elements [32 x hi | 32 x lo] = one 128-byte K step of gemm_mainloop{,_dma}; 4 bytes per element.
//         Optionally also the plain hi plane (rows, k) -- the f16 FILTER operand of K6 -- and the row's rounding-error
//         norm || x - hi / S ||_2 for the certificate.
//   KCAT  (xml_split_f16_kcat / xml_pack_weights_f16s)   operands of the projection GEMMs: A' (M, 3K) = [hi | lo | hi],
//         W' (N, 3K) = [hi | hi | lo], so that the UNCHANGED 256 x 256 LDS-DMA GEMM run over K' = 3 K in f16 produces
//         hi.hi + lo.hi + hi.lo.  A' is a per-call temporary; W' is packed once per weight.
// Scales: a power of two per ROW for activations (row maximum -> [2^13, 2^14); returned as inv_scale[row] = 1 / S and
// applied in the consumer's epilogue), one per TENSOR for weights (stored behind W', folded into inv_scale by the A split),
// a FIXED 2^fixed_log2 for unit-norm rows (the K6 filter's epilogue cannot afford per-column scales).
// Subnormal halves are flushed to zero on write: what an MFMA consumes is exactly what the error norms describe.
#include "common.h"

namespace {

__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((uint32_t)(e + 127) << 23); }   // e in [-126, 127]

// power-of-two scale that maps a row / tensor maximum m into [2^13, 2^14); m == 0 or non-finite -> 1
__device__ __forceinline__ int scale_log2_of(float m) {
  if (!(m > 0.f) || !(m < 3.0e38f)) return 0;
  int q;
  (void)frexpf(m, &q);                    // m = f 2^q, f in [0.5, 1)
  const int e = 14 - q;
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}

// 8 consecutive elements -> 8 hi halves + 8 lo halves (raw bits packed two per word), sum of squares of the hi residual
__device__ __forceinline__ float split8(const float* f, float s, uint4& hi, uint4& lo) {
  uint32_t h[8], l[8];
  float e2 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float xs = f[i] * s;                        // exact (power of two) unless it over/underflows
    h[i] = f32_to_f16_bits_ftz(xs);
    const float r = xs - f16_bits_to_f32(h[i]);       // exact in f32: |r| <= ulp_f16(xs) / 2 (or xs itself when flushed)
    l[i] = f32_to_f16_bits_ftz(r);
    e2 += r * r;
  }
  hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
  lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
  return e2;
}

__device__ __forceinline__ float row_absmax(const float* px, int k, int lane) {
  float m = 0.f;
  for (int c = lane * 8; c < k; c += 64 * 8) {
    float f[8];
    ld8<float>(px + c, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf(f[i]));
  }
  return wave_max(m);
}

// one wave per row.  MODE 0: interleaved ROWS (+ optional hi plane / err); MODE 1: KCAT A' = [hi | lo | hi]
template <int MODE>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ src_row,
                                                         char* __restrict__ y, float* __restrict__ inv_scale,
                                                         unsigned short* __restrict__ hi_plane, float* __restrict__ err,
                                                         const float* __restrict__ extra_inv, int64_t rows, int k,
                                                         int fixed_log2) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* px = x + (src_row ? (int64_t)src_row[row] : row) * k;
  const int e = fixed_log2 > -1000 ? fixed_log2 : scale_log2_of(row_absmax(px, k, lane));
  const float s = pow2f(e);
  float e2 = 0.f;
  for (int c = lane * 8; c < k; c += 64 * 8) {
    float f[8];
    ld8<float>(px + c, f);
    uint4 hi, lo;
    e2 += split8(f, s, hi, lo);
    if (MODE == 0) {
      char* g = y + row * (int64_t)k * 4 + (int64_t)(c >> 5) * 128 + (c & 31) * 2;
      *reinterpret_cast<uint4*>(g) = hi;
      *reinterpret_cast<uint4*>(g + 64) = lo;
      if (hi_plane) *reinterpret_cast<uint4*>(hi_plane + row * k + c) = hi;
    } else {
      char* g = y + row * (int64_t)k * 6 + (int64_t)c * 2;
      *reinterpret_cast<uint4*>(g) = hi;
      *reinterpret_cast<uint4*>(g + (int64_t)k * 2) = lo;
      *reinterpret_cast<uint4*>(g + (int64_t)k * 4) = hi;
    }
  }
  if (err) {
    e2 = wave_sum(e2);
    if (lane == 0) err[row] = sqrtf(e2) * pow2f(-e);
  }
  if (inv_scale && lane == 0) inv_scale[row] = pow2f(-e) * (extra_inv ? *extra_inv : 1.f);
}

// weights: tensor maximum (atomicMax on the bits of |w|: order-preserving for non-negative floats) ...
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ w, int64_t n, uint32_t* __restrict__ out_bits) {
  float m = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(w + i);
      m = fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
    } else {
      for (int64_t j = i; j < n; ++j) m = fmaxf(m, fabsf(w[j]));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));
}
// ... then W' (n, 3k) = [hi | hi | lo] at the tensor's scale; trailer[0] = 1 / S (f32), trailer[1] = bits of the maximum
__global__ __launch_bounds__(256) void pack_weights_f16s_kernel(const float* __restrict__ w, char* __restrict__ dst,
                                                                float* __restrict__ trailer, int n, int k) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int e = scale_log2_of(__uint_as_float(reinterpret_cast<const uint32_t*>(trailer)[1]));
  if (blockIdx.x == 0 && threadIdx.x == 0) trailer[0] = pow2f(-e);
  if (row >= n) return;
  const float s = pow2f(e);
  const float* px = w + (int64_t)row * k;
  for (int c = lane * 8; c < k; c += 64 * 8) {
    float f[8];
    ld8<float>(px + c, f);
    uint4 hi, lo;
    (void)split8(f, s, hi, lo);
    char* g = dst + (int64_t)row * k * 6 + (int64_t)c * 2;
    *reinterpret_cast<uint4*>(g) = hi;
    *reinterpret_cast<uint4*>(g + (int64_t)k * 2) = hi;
    *reinterpret_cast<uint4*>(g + (int64_t)k * 4) = lo;
  }
}

// back to f32 (tests, the CPU baseline's view of a split index): x = (hi + lo) * inv_scale[row]
__global__ __launch_bounds__(256) void unsplit_rows_kernel(const char* __restrict__ y, const float* __restrict__ inv_scale,
                                                           float* __restrict__ x, int64_t rows, int k) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float inv = inv_scale[row];
  for (int c = lane * 8; c < k; c += 64 * 8) {
    const char* g = y + row * (int64_t)k * 4 + (int64_t)(c >> 5) * 128 + (c & 31) * 2;
    const uint4 hi = *reinterpret_cast<const uint4*>(g), lo = *reinterpret_cast<const uint4*>(g + 64);
    const uint32_t hw[4] = {hi.x, hi.y, hi.z, hi.w}, lw[4] = {lo.x, lo.y, lo.z, lo.w};
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t hb = (hw[i >> 1] >> ((i & 1) * 16)) & 0xffffu, lb = (lw[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
      f[i] = (f16_bits_to_f32(hb) + f16_bits_to_f32(lb)) * inv;
    }
    st8<float>(x + row * k + c, f);
  }
}

}  // namespace

// internal (linear.hip: the A operand of a split projection GEMM)
int xmli_split_f16_kcat(const float* x, void* a_cat, float* inv_scale, const float* w_trailer, int64_t rows, int k,
                        hipStream_t st) {
  if (!x || !a_cat || !inv_scale || rows <= 0 || k <= 0 || k % 8) return XML_ERR_BAD_ARG;
  hipLaunchKernelGGL(split_rows_kernel<1>, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, (const int32_t*)nullptr, (char*)a_cat,
                     inv_scale, (unsigned short*)nullptr, (float*)nullptr, w_trailer, rows, k, -100000);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" int xml_split_f16_rows(const float* x, void* y, float* inv_scale, void* hi_plane, float* err, int64_t rows,
                                  int k, int fixed_log2, xml_stream_t stream) {
  XML_ENTER();
  if (!x || !y || rows <= 0 || k <= 0) return XML_ERR_BAD_ARG;
  if (k % 32) return XML_ERR_UNSUPPORTED;
  if (fixed_log2 > 30 || (fixed_log2 < -30 && fixed_log2 != -1)) return XML_ERR_BAD_ARG;
  hipLaunchKernelGGL(split_rows_kernel<0>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x,
                     (const int32_t*)nullptr, (char*)y, inv_scale, (unsigned short*)hi_plane, err, (const float*)nullptr,
                     rows, k, fixed_log2 == -1 ? -100000 : fixed_log2);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" int xml_unsplit_f16_rows(const void* y, const float* inv_scale, float* x, int64_t rows, int k,
                                    xml_stream_t stream) {
  XML_ENTER();
  if (!x || !y || !inv_scale || rows <= 0 || k <= 0) return XML_ERR_BAD_ARG;
  if (k % 32) return XML_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(unsplit_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const char*)y, inv_scale,
                     x, rows, k);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" size_t xml_pack_weights_f16s_bytes(int n, int k) {
  if (n <= 0 || k <= 0) return 0;
  return align_up((size_t)n * k * 6, 16) + 16;
}

extern "C" int xml_pack_weights_f16s(const float* w, void* dst, int n, int k, xml_stream_t stream) {
  XML_ENTER();
  if (!w || !dst || n <= 0 || k <= 0) return XML_ERR_BAD_ARG;
  if (k % 8) return XML_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  float* trailer = reinterpret_cast<float*>((char*)dst + align_up((size_t)n * k * 6, 16));
  if (!xml_zero_async(trailer, 16, st)) return XML_ERR_LAUNCH;
  const int64_t total = (int64_t)n * k;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(total / 1024 < 1 ? 1 : (total / 1024 > 1024 ? 1024 : total / 1024))),
                     dim3(256), 0, st, w, total, reinterpret_cast<uint32_t*>(trailer) + 1);
  XML_CHECK_LAUNCH();
  hipLaunchKernelGGL(pack_weights_f16s_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, w, (char*)dst, trailer, n, k);
  XML_CHECK_LAUNCH();
  return XML_OK;
}
