// F.normalize(x, dim=-1) row arithmetic shared by xml_l2norm_rows (linear.hip) and the fused normalise-and-tile index
// build (index_build.hip): both produce bitwise the same values, so an index built by the fused kernel holds exactly what
// l2norm_rows + tile_rows would have written.
//   A row is handled by HALF a wave (32 lanes): lane l sums the squares of its 16-byte chunks c = l, l + 32, l + 64, ...
//   (elements in order), the 32 partial sums are combined by an xor butterfly (16, 8, 4, 2, 1), and every element is
//   divided by max(sqrt(sum), 1e-12).
#pragma once
#include "common.h"

constexpr int L2N_MAXJ = 8;      // chunks per lane: rows of up to 32 * 8 chunks = 4096 bytes

template <typename T>
__device__ __forceinline__ bool l2n_ok(int d) {
  constexpr int VEC = 16 / (int)sizeof(T);
  return d % VEC == 0 && d / VEC <= 32 * L2N_MAXJ;
}

// loads the row's chunks of this lane into `v` (zeros beyond the row, or for a missing row) and returns 1 / max(norm, eps)
// as the DIVISOR (the callers divide, like the reference)
template <typename T>
__device__ __forceinline__ float l2n_load_row(const T* px, int d, int l32, uint4 (&v)[L2N_MAXJ]) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int chunks = d / VEC;
#pragma unroll
  for (int j = 0; j < L2N_MAXJ; ++j) {
    const int c = l32 + 32 * j;
    v[j] = make_uint4(0u, 0u, 0u, 0u);
    if (px && c < chunks) v[j] = ld_global16(px + c * VEC);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < L2N_MAXJ; ++j) {
    float f[8];
    unpack16<T>(v[j], f);
#pragma unroll
    for (int e = 0; e < VEC; ++e) s += f[e] * f[e];
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return fmaxf(sqrtf(s), 1e-12f);
}

template <typename T>
__device__ __forceinline__ uint4 l2n_scale_chunk(const uint4& v, float nrm) {
  constexpr int VEC = 16 / (int)sizeof(T);
  float f[8];
  unpack16<T>(v, f);
#pragma unroll
  for (int e = 0; e < VEC; ++e) f[e] = f[e] / nrm;
  return pack16<T>(f);
}
