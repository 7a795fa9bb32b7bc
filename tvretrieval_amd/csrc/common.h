// Shared device helpers for libxmlhip (gfx950 / CDNA4 only: wave64, MFMA 16x16 f32-accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/xmlhip.h"
#include "debug.h"

// hipGetLastError() also reports stale, non-sticky errors left by OTHER runtime users in this thread (e.g. the
// caching allocator's hipErrorNotReady event polls), so every entry point clears the slot before launching.
#define XML_ENTER() (void)hipGetLastError()

#define XML_CHECK_LAUNCH()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return XML_ERR_LAUNCH; \
  } while (0)

typedef unsigned short bf16_t;  // raw bf16 bits
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_v __attribute__((ext_vector_type(8)));

static constexpr int WAVE = 64;

// ---- dtype traits ------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// f32 -> bf16, round to nearest even: v_cvt_pk_bf16_f32 (gfx950).  The former integer version (NaN test + add + shift)
// compiled into a divergent branch per element: ~25 instructions per value in every bf16 epilogue.
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 xml_bf16x2_t;
typedef __attribute__((__vector_size__(2 * sizeof(float)))) float xml_f32x2_t;
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {      // lo -> bits [15:0]
  const xml_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, xml_bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(f32x2_to_bf16x2(f, 0.f) & 0xffffu); }

template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int id = XML_F32;
  static constexpr int vec = 4;  // elements per 16-byte vector
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
  static constexpr int id = XML_BF16;
  static constexpr int vec = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 16-byte vector <-> floats
template <typename T> __device__ __forceinline__ void unpack16(const uint4& v, float* out);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& v, float* out) {
  out[0] = __uint_as_float(v.x); out[1] = __uint_as_float(v.y);
  out[2] = __uint_as_float(v.z); out[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& v, float* out) {
  out[0] = __uint_as_float(v.x << 16); out[1] = __uint_as_float(v.x & 0xffff0000u);
  out[2] = __uint_as_float(v.y << 16); out[3] = __uint_as_float(v.y & 0xffff0000u);
  out[4] = __uint_as_float(v.z << 16); out[5] = __uint_as_float(v.z & 0xffff0000u);
  out[6] = __uint_as_float(v.w << 16); out[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* in);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* in) {
  return make_uint4(__float_as_uint(in[0]), __float_as_uint(in[1]), __float_as_uint(in[2]), __float_as_uint(in[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* in) {
  uint4 v;
  v.x = f32x2_to_bf16x2(in[0], in[1]);
  v.y = f32x2_to_bf16x2(in[2], in[3]);
  v.z = f32x2_to_bf16x2(in[4], in[5]);
  v.w = f32x2_to_bf16x2(in[6], in[7]);
  return v;
}

// 8 consecutive elements as floats (16-byte loads for bf16, 2 x 16 bytes for f32)
template <typename T> __device__ __forceinline__ void ld8(const T* p, float* out) {
  if constexpr (sizeof(T) == 2) {
    unpack16<bf16_t>(*reinterpret_cast<const uint4*>(p), out);
  } else {
    unpack16<float>(*reinterpret_cast<const uint4*>(p), out);
    unpack16<float>(*reinterpret_cast<const uint4*>(p + 4), out + 4);
  }
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float* in) {
  if constexpr (sizeof(T) == 2) {
    *reinterpret_cast<uint4*>(p) = pack16<bf16_t>(in);
  } else {
    *reinterpret_cast<uint4*>(p) = pack16<float>(in);
    *reinterpret_cast<uint4*>(p + 4) = pack16<float>(in + 4);
  }
}

// ---- MFMA "chunk": one 64-byte slice of K for a 16x16 output tile -------------------------------
// Lane l supplies 16 bytes of row (l & 15) at byte offset (l >> 4) * 16 of the 64-byte K chunk, for both
// operands (A rows = output rows, B rows = output columns; both K-contiguous).
//   bf16: the 16 bytes are 8 consecutive k  -> one v_mfma_f32_16x16x32_bf16
//   f32 : the 16 bytes are 4 consecutive k  -> four v_mfma_f32_16x16x4_f32; MFMA j consumes element j of
//         every lane, i.e. k = 4*(l>>4) + j.  Both operands use the same k permutation, so the dot product
//         is over the same 16 k values (summation order differs from sequential k; exact f32 fma chain).
// Accumulator layout (both): acc[r] = D[row = (l >> 4) * 4 + r][col = l & 15].
template <typename T> struct Mma;
template <> struct Mma<float> {
  __device__ static __forceinline__ void chunk(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};
template <> struct Mma<bf16_t> {
  __device__ static __forceinline__ void chunk(f32x4& acc, const uint4& a, const uint4& b) {
    union { uint4 u; bf16x8_v v; } ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc, 0, 0, 0);
  }
};

// ---- f16 / split-f16 (exact-rank mode on the 16-bit MFMA pipe) --------------------------------------------------------
// f16_t : IEEE half storage (raw bits), one v_mfma_f32_16x16x32_f16 per 64-byte chunk -- same rate as bf16, 3 more
//         mantissa bits (the exact-rank FILTER: rounding error 8x smaller than bf16's, so half the candidates suffice).
// f16s_t: "split f16" -- an f32-grade value x carried as hi + lo, hi = rn_f16(x S), lo = rn_f16(x S - hi) (S a power of two
//         per row): |x S - hi - lo| <= 2^-22 |x S|.  Rows are stored INTERLEAVED per 32 elements: [32 x hi | 32 x lo] =
//         128 bytes, the K step of gemm_mainloop{,_dma} -- 4 bytes per element like f32.  A step is multiplied as
//         hi.hi + lo.hi + hi.lo (three f16 MFMAs, f32 accumulate; the dropped lo.lo term is 2^-22 relative): f32-grade dot
//         products at 16/3 of the f32 MFMA rate.  sizeof(f16s_t) == 4 so that k_bytes = K * sizeof(T) holds.
// Subnormal halves are flushed to zero when the planes are WRITTEN (split16.hip), so what the MFMA consumes is what the
// error norms were computed from, whatever its denormal mode.
struct f16_t { unsigned short bits; };
struct f16s_t { uint32_t pair; };
typedef _Float16 f16x8_v __attribute__((ext_vector_type(8)));
template <> struct Mma<f16_t> {
  __device__ static __forceinline__ void chunk(f32x4& acc, const uint4& a, const uint4& b) {
    union { uint4 u; f16x8_v v; } ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ua.v, ub.v, acc, 0, 0, 0);
  }
};
template <typename T> struct IsSplit16 { static constexpr bool value = false; };
template <> struct IsSplit16<f16s_t> { static constexpr bool value = true; };
// rn_f16(v) as raw bits, subnormal results flushed to (signed) zero
__device__ __forceinline__ uint32_t f32_to_f16_bits_ftz(float v) {
  const _Float16 h = (_Float16)v;
  const uint32_t b = (uint32_t)__builtin_bit_cast(unsigned short, h);
  return (b & 0x7c00u) ? b : (b & 0x8000u);
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t b) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)b);
}

// ---- wave / block reductions ---------------------------------------------------------------------
// reductions over the 16 lanes that share (lane >> 4)
__device__ __forceinline__ float lane16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// same reduction with DPP row rotations (v_max_f32_dpp row_ror:8/4/2/1): no LDS crossbar traffic, all 16 lanes
// of each row end up with the row maximum
__device__ __forceinline__ float lane16_max_dpp(float v) {
#define XML_ROR(x, n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + (n), 0xf, 0xf, false))
  v = fmaxf(v, XML_ROR(v, 8));
  v = fmaxf(v, XML_ROR(v, 4));
  v = fmaxf(v, XML_ROR(v, 2));
  v = fmaxf(v, XML_ROR(v, 1));
#undef XML_ROR
  return v;
}
__device__ __forceinline__ float lane16_sum_dpp(float v) {
#define XML_ROR(x, n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + (n), 0xf, 0xf, false))
  v += XML_ROR(v, 8);
  v += XML_ROR(v, 4);
  v += XML_ROR(v, 2);
  v += XML_ROR(v, 1);
#undef XML_ROR
  return v;
}
// 64-lane reductions: four DPP row rotations inside each 16-lane row, then two crossbar steps (xor 16, 32).  The plain
// xor butterfly was six dependent ds_bpermute round trips; every row-wise kernel (LayerNorm forward / backward, ConvSE
// softmax, pooling) chains several of these per row.
__device__ __forceinline__ float wave_sum(float v) {
  v = lane16_sum_dpp(v);
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float wave_max(float v) {
  v = lane16_max_dpp(v);
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float lane16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ uint4 ld_global16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
// 16-byte load through an explicit GLOBAL pointer.  A pointer that reaches a kernel inside a by-value struct, or comes
// out of a lambda that may return nullptr, is a generic ("flat") pointer to hipcc; a load through it that also sits in a
// `cond ? load : zero` select was emitted as FOUR flat_load_dword (4 x the instructions, each touching 64 x 4 bytes at a
// 16-byte stride) -- this is what every user of gemm_mainloop was running on.  p must be a valid global address.
typedef unsigned int xml_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_global16_as1(const void* p) {
  const xml_u32x4_t v = *reinterpret_cast<const __attribute__((address_space(1))) xml_u32x4_t*>(
      reinterpret_cast<uintptr_t>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_global16(void* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = v; }

// Zero-fill as a KERNEL.  hipMemsetAsync is fine on a stream, but as a memset NODE of a captured HIP graph it did not
// run again on the second replay (ROCm 7.2; found with inference.GraphedVcmrSearch -- stale K7 cursors); every fill inside
// an entry point that may be captured (the search pass, the training step) is therefore a kernel.  bytes % 4 == 0.
static __global__ void xml_zero_words_kernel(uint32_t* __restrict__ p, int64_t n_words) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n_words && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    *reinterpret_cast<uint4*>(p + i) = make_uint4(0u, 0u, 0u, 0u);
  } else {
    for (int k = 0; k < 4; ++k) if (i + k < n_words) p[i + k] = 0u;
  }
}
static inline bool xml_zero_async(void* p, size_t bytes, hipStream_t st) {
  const int64_t n_words = (int64_t)((bytes + 3) / 4);
  if (n_words <= 0) return true;
  hipLaunchKernelGGL(xml_zero_words_kernel, dim3((unsigned)((n_words + 1023) / 1024)), dim3(256), 0, st, (uint32_t*)p, n_words);
  return hipGetLastError() == hipSuccess;
}

// dropout seed words of xml_dropout / the fused training attention: host seed + optional device-resident base seed (a
// captured training step advances the base seed on the device, so every replay draws fresh masks)
__device__ __forceinline__ void xml_seed_words(uint64_t seed, const uint64_t* seed_dev, uint32_t& s0, uint32_t& s1) {
  if (seed_dev) seed += *seed_dev;
  s0 = (uint32_t)seed;
  s1 = (uint32_t)(seed >> 32) * 0x27D4EB2Fu + 0x165667B1u;
}

// counter-based dropout mask of xml_dropout: element i of a tensor is KEPT when drop_hash(i, s0, s1) >= p * 2^32
__device__ __forceinline__ uint32_t drop_hash(uint64_t i, uint32_t s0, uint32_t s1) {
  uint32_t h = (uint32_t)i * 0x9E3779B1u + s0;
  h ^= (uint32_t)(i >> 32) * 0x85EBCA77u;
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h += s1; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
// The same masks applied to 8 consecutive elements i0 .. i0 + 7 of a tensor: v[j] = keep(i0 + j) ? v[j] * scale : 0.  The index
// multiply and the high-word term are taken once per vector -- two of drop_hash's four integer multiplies (quarter rate on
// CDNA) per element instead of four; bit for bit drop_hash(i0 + j, ..) (the one vector in 2^32 elements whose low index word
// wraps inside it takes the element-wise form).
__device__ __forceinline__ void drop_mask8(uint64_t i0, uint32_t s0, uint32_t s1, uint32_t thresh, float scale, float* v) {
  if ((uint32_t)i0 > 0xfffffff8u) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = drop_hash(i0 + j, s0, s1) >= thresh ? v[j] * scale : 0.f;
    return;
  }
  const uint32_t base = (uint32_t)i0 * 0x9E3779B1u + s0;
  const uint32_t hi = (uint32_t)(i0 >> 32) * 0x85EBCA77u;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    uint32_t h = (base + (uint32_t)j * 0x9E3779B1u) ^ hi;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h += s1; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    v[j] = h >= thresh ? v[j] * scale : 0.f;
  }
}
// dropout sites fused into another kernel's loads / stores (xml_add_layernorm_drop, xml_layernorm_bwd_drop):
// thresh == 0 means "no dropout at this site"
struct XmlDropSite {
  uint32_t thresh;
  float scale;
  uint64_t seed;
};
static inline XmlDropSite xml_drop_site(float p, uint64_t seed) {
  XmlDropSite s;
  s.thresh = p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u;
  s.scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  s.seed = seed;
  return s;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline size_t dt_size(int dt) { return (dt == XML_F32 || dt == XML_F16S) ? 4 : 2; }
