// Projection GEMM on the 256 x 256 LDS-DMA ring (the K-loop schedule of q2c_persist.hip, one tile per workgroup):
//   out[m][n] = act(A[m] . W[n] + bias[n]) + addend        A (M, K), W (N, K) both K-contiguous
// Used for the encoder projections (K1, QKV, output dense) when K * sizeof(T) is a multiple of 128 bytes and
// M >= 256; the 128 x 128 register-staged kernel (linear.hip) covers every other shape.
#include <type_traits>

#include "common.h"
#include "internal.h"

__device__ __forceinline__ void g256_dma(uint32_t voff, const char* sbase, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
// the four pieces of a slice in one statement, M0 advanced by immediates (see q2c_persist.hip: every scalar
// instruction and branch in the per-slice path of a wave is issue time its partner wave's MFMAs do not fully cover)
__device__ __forceinline__ void g256_dma4(uint32_t va0, uint32_t va1, uint32_t vb0, uint32_t vb1, const char* sa,
                                          const char* sb, uint32_t lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
      "s_add_u32 m0, m0, 0x3c00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6"
      :
      : "v"(va0), "v"(va1), "v"(vb0), "v"(vb1), "s"(lds_dst), "s"(sa), "s"(sb)
      : "memory", "scc");
}
template <typename T> struct G256Init;      // first K chunk of the tile: C = 0 as an inline constant
template <> struct G256Init<float> {
  __device__ static __forceinline__ void chunk(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};
template <> struct G256Init<bf16_t> {
  __device__ static __forceinline__ void chunk(f32x4& acc, const uint4& a, const uint4& b) {
    union { uint4 u; bf16x8_v v; } ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  }
};
template <> struct G256Init<f16_t> {
  __device__ static __forceinline__ void chunk(f32x4& acc, const uint4& a, const uint4& b) {
    union { uint4 u; f16x8_v v; } ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ua.v, ub.v, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  }
};
template <int CTRL>
__device__ __forceinline__ uint4 g256_dpp_u4(const uint4& v) {
  uint4 r;
  r.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, CTRL, 0xf, 0xf, false);
  r.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, CTRL, 0xf, 0xf, false);
  r.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.z, CTRL, 0xf, 0xf, false);
  r.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, CTRL, 0xf, 0xf, false);
  return r;
}
__device__ __forceinline__ int g256_swz(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

// RAGGED_N: N % 8 != 0 -> per-element tail stores (kept out of the common instantiation: its 64-bit modulo and
// per-element branches, unrolled 16 times, made the kernel 640 KB of code and the epilogue instruction-fetch bound)
template <typename T, typename OutT, typename AddT, bool RAGGED_N = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                         const float* __restrict__ bias,
                                                         const AddT* __restrict__ addend, OutT* __restrict__ out,
                                                         int64_t M, int N, int K, int relu, int add_mode, int seq_len,
                                                         int tm, int tn, const float* __restrict__ row_scale) {
  // T == f16_t: the split-f16 projection (split16.hip): A / W are the K-concatenated halves, K = 3 x the logical K, and
  // every accumulator row is multiplied by row_scale[m] = 1 / (S_row S_weights) before bias / activation
  constexpr bool SCALED = std::is_same<T, f16_t>::value;
  constexpr int ROWB = 64;
  constexpr int OPER_BYTES = 256 * ROWB;
  constexpr int SLOT_BYTES = 2 * OPER_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware tile order: 8 (M tiles) x 4 (N tiles) super-tiles per XCD
  const int b = blockIdx.x;
  const int xcd = b & 7, local = b >> 3;
  const int sup = (local >> 5) * 8 + xcd;
  const int w32 = local & 31;
  const int sm = (tm + 7) >> 3;
  const int mt_ = (sup % sm) * 8 + (w32 & 7);
  const int nt_ = (sup / sm) * 4 + (w32 >> 3);
  if (mt_ >= tm || nt_ >= tn) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int64_t m0 = (int64_t)mt_ * 256;
  const int n0 = nt_ * 256;
  const int k_bytes = K * (int)sizeof(T);
  const int n_slices = k_bytes / ROWB;

  uint32_t voff_a[2], voff_b[2];
  {
    const int rsub = lane >> 2, pslot = lane & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave * 2 + i) * 16 + rsub;
      const int slot = pslot ^ g256_swz(row);
      voff_a[i] = (uint32_t)((m0 + row < M) ? row : 0) * k_bytes + slot * 16;
      voff_b[i] = (uint32_t)((n0 + row < N) ? row : 0) * k_bytes + slot * 16;
    }
  }
  const char* sbase_a = reinterpret_cast<const char*>(A) + m0 * k_bytes;
  const char* sbase_b = reinterpret_cast<const char*>(W) + (int64_t)n0 * k_bytes;
  const uint32_t lds_wave = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 2048;
  int i_slice = 0;
  auto issue_slice = [&]() {        // callers guarantee i_slice < n_slices
    const int koff = i_slice * ROWB;
    const uint32_t dst = lds_wave + (i_slice & 3) * SLOT_BYTES;
    g256_dma4(voff_a[0], voff_a[1], voff_b[0], voff_b[1], sbase_a + koff, sbase_b + koff, dst);
    ++i_slice;
  };

  const int a_off = (wm * 64 + fr) * ROWB + ((fg ^ g256_swz(fr)) << 4);
  // DIRECT epilogue (N % 8 == 0; see gemm256p.hip): swapped MFMA operands -> a lane's accumulator holds 4 consecutive
  // columns of ONE row; 2-byte outputs pair the accumulators (W row of fragment n, MFMA row index i:
  // (n >> 1) * 32 + (i >> 2) * 8 + (n & 1) * 4 + (i & 3)) so that a lane owns 8 consecutive columns.  Same products and
  // K order per element: bitwise the results of the staged epilogue (which the ragged-N instantiation keeps).
  constexpr bool DIRECT = !RAGGED_N;
  constexpr bool PAIRED = DIRECT && sizeof(OutT) == 2;
  const int rb_e = PAIRED ? (fr >> 2) * 8 + (fr & 3) : fr;
  const int b_off = OPER_BYTES + (wn * 128 + rb_e) * ROWB + ((fg ^ g256_swz(rb_e)) << 4);
  const int b_off_o = OPER_BYTES + (wn * 128 + rb_e + 4) * ROWB + ((fg ^ g256_swz(rb_e + 4)) << 4);   // PAIRED, odd n
  auto b_frag = [&](const char* slot, int n) -> uint4 {
    if constexpr (PAIRED) return *reinterpret_cast<const uint4*>(slot + ((n & 1) ? b_off_o : b_off) + (n >> 1) * 32 * ROWB);
    else return *reinterpret_cast<const uint4*>(slot + b_off + n * 16 * ROWB);
  };

  issue_slice(); issue_slice(); issue_slice(); issue_slice();      // n_slices >= 4 (eligibility)
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x4 acc[4][8];                  // written by the first slice (G256Init: C = 0)
  uint4 faA[4], faB[4], fbL[4], fbH[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) faA[m] = *reinterpret_cast<const uint4*>(smem + a_off + m * 16 * ROWB);
#pragma unroll
  for (int n = 0; n < 4; ++n) fbL[n] = b_frag(smem, n);

  // One step = one 32-wide K slice (see q2c_persist.hip for the schedule).  The tail of the stream is PEELED instead
  // of tested: MODE 0 = steady state (wait for slice c+1 with two younger slices in flight, read, issue slice c+4),
  // MODE 1 / 2 / 3 = the last three steps that still prefetch fragments (2 / 1 / 0 slices left in flight, nothing to
  // issue), MODE 4 = the last slice (nothing to wait for or to read).  GRP1: the second wave of each SIMD runs its
  // second MFMA half before its preamble, so the two waves' non-MFMA phases do not coincide.
  int c_slice = 0;
  auto run = [&](auto grp_tag) {
    constexpr bool GRP1 = decltype(grp_tag)::value;
    auto slice_step = [&](uint4 (&fc)[4], uint4 (&fn)[4], auto mode_tag, auto init_tag) {
      constexpr int MODE = decltype(mode_tag)::value;
      constexpr bool INIT = decltype(init_tag)::value;
      const char* slot = smem + (c_slice & 3) * SLOT_BYTES;
#pragma unroll
      for (int n = 0; n < 4; ++n) fbH[n] = b_frag(slot, n + 4);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if constexpr (DIRECT) {
            if constexpr (INIT) G256Init<T>::chunk(acc[m][n], fbL[n], fc[m]);
            else Mma<T>::chunk(acc[m][n], fbL[n], fc[m]);
          } else {
            if constexpr (INIT) G256Init<T>::chunk(acc[m][n], fc[m], fbL[n]);
            else Mma<T>::chunk(acc[m][n], fc[m], fbL[n]);
          }
        }
      if constexpr (MODE <= 1) __builtin_amdgcn_s_waitcnt(0x0078);        // vmcnt(8) lgkmcnt(0)
      else if constexpr (MODE == 2) __builtin_amdgcn_s_waitcnt(0x0074);   // vmcnt(4) lgkmcnt(0)
      else if constexpr (MODE == 3) __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
      if constexpr (MODE <= 3) __builtin_amdgcn_s_barrier();
      ++c_slice;
      auto preamble = [&]() {
        if constexpr (MODE <= 3) {
          const char* nslot = smem + (c_slice & 3) * SLOT_BYTES;
#pragma unroll
          for (int m = 0; m < 4; ++m) fn[m] = *reinterpret_cast<const uint4*>(nslot + a_off + m * 16 * ROWB);
#pragma unroll
          for (int n = 0; n < 4; ++n) fbL[n] = b_frag(nslot, n);
        }
        if constexpr (MODE == 0) issue_slice();
      };
      if (!GRP1) preamble();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if constexpr (DIRECT) {
            if constexpr (INIT) G256Init<T>::chunk(acc[m][n + 4], fbH[n], fc[m]);
            else Mma<T>::chunk(acc[m][n + 4], fbH[n], fc[m]);
          } else {
            if constexpr (INIT) G256Init<T>::chunk(acc[m][n + 4], fc[m], fbH[n]);
            else Mma<T>::chunk(acc[m][n + 4], fc[m], fbH[n]);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
      if (GRP1) preamble();
    };
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    using M3 = std::integral_constant<int, 3>;
    using M4 = std::integral_constant<int, 4>;
    const int n_main = n_slices - 4;             // even: steps 0 .. n_slices - 5 run in steady state
    if (n_main > 0) {
      slice_step(faA, faB, M0{}, std::true_type{});
      slice_step(faB, faA, M0{}, std::false_type{});
      for (int s2 = 2; s2 < n_main; s2 += 2) {
        slice_step(faA, faB, M0{}, std::false_type{});
        slice_step(faB, faA, M0{}, std::false_type{});
      }
      slice_step(faA, faB, M1{}, std::false_type{});
    } else {
      slice_step(faA, faB, M1{}, std::true_type{});
    }
    slice_step(faB, faA, M2{}, std::false_type{});
    slice_step(faA, faB, M3{}, std::false_type{});
    slice_step(faB, faA, M4{}, std::false_type{});
  };
  if (wave >> 2) run(std::true_type{});
  else run(std::false_type{});

  // ---- DIRECT epilogue: accumulators -> global memory, whole 128-byte lines per store instruction (gemm256p.hip) ------
  if constexpr (DIRECT) {
    constexpr int GC = PAIRED ? 8 : 4;            // columns per group: group q of a lane = GC consecutive columns of row fr
    constexpr int NGRP = 128 / (4 * GC);
    static_assert(GC * sizeof(AddT) <= 16, "addend group wider than one 16-byte load");
    int gcol[NGRP], pcol[NGRP / 2];
    bool gok[NGRP];
    float gb[NGRP][GC];
#pragma unroll
    for (int q = 0; q < NGRP; ++q) {
      gcol[q] = n0 + wn * 128 + q * (4 * GC) + fg * GC;
      gok[q] = gcol[q] + GC <= N;
#pragma unroll
      for (int e = 0; e < GC; e += 4) {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && gok[q]) b4 = *reinterpret_cast<const float4*>(bias + gcol[q] + e);
        gb[q][e] = b4.x; gb[q][e + 1] = b4.y; gb[q][e + 2] = b4.z; gb[q][e + 3] = b4.w;
      }
    }
#pragma unroll
    for (int pq = 0; pq < NGRP / 2; ++pq) pcol[pq] = n0 + wn * 128 + pq * (8 * GC) + (fr & 1) * (4 * GC) + fg * GC;
    const uint32_t m0_mod = add_mode == 1 ? (uint32_t)(m0 % seq_len) : 0u;       // one 64-bit modulo per tile
    const bool odd = (fr & 1) != 0;
#pragma unroll
    for (int m4 = 0; m4 < 4; ++m4) {
      const int lrow = wm * 64 + m4 * 16 + fr;
      const int64_t m = m0 + lrow;
      const bool rok = m < M;
      float rsc = 1.f;
      if constexpr (SCALED) rsc = rok ? row_scale[m] : 0.f;
      float v[NGRP][GC];
#pragma unroll
      for (int q = 0; q < NGRP; ++q)
#pragma unroll
        for (int e = 0; e < GC; ++e) {
          float x = (PAIRED ? acc[m4][2 * q + (e >> 2)][e & 3] : acc[m4][q][e & 3]);
          if constexpr (SCALED) x *= rsc;
          x += gb[q][e];
          if (relu) x = fmaxf(x, 0.f);
          v[q][e] = x;
        }
      if (add_mode) {
        const int64_t arow = add_mode == 1 ? (int64_t)((m0_mod + (uint32_t)lrow) % (uint32_t)seq_len) : m;
        constexpr int AB = GC * (int)sizeof(AddT);
#pragma unroll
        for (int qb = 0; qb < NGRP; qb += 4) {
          uint4 ad[4];
#pragma unroll
          for (int qi = 0; qi < 4; ++qi) {
            const int q = qb + qi;
            ad[qi] = make_uint4(0u, 0u, 0u, 0u);
            if (rok && gok[q]) {
              if constexpr (AB == 16) ad[qi] = ld_global16(addend + arow * N + gcol[q]);
              else { const uint2 u = *reinterpret_cast<const uint2*>(addend + arow * N + gcol[q]); ad[qi].x = u.x; ad[qi].y = u.y; }
            }
          }
#pragma unroll
          for (int qi = 0; qi < 4; ++qi) {
            float av[8];
            if constexpr (sizeof(AddT) == 2) unpack16<bf16_t>(ad[qi], av);
            else unpack16<float>(ad[qi], av);
#pragma unroll
            for (int e = 0; e < GC; ++e) v[qb + qi][e] += av[e];
          }
        }
      }
      // groups 2 p and 2 p + 1 of a row are the halves of one 128-byte line: the EVEN store writes rows fr & ~1 (odd lanes
      // bring group 2 p + 1 of the row below them, DPP row_shr:1), the ODD store rows fr | 1 (row_shl:1)
      uint4 pk[NGRP];
#pragma unroll
      for (int q = 0; q < NGRP; ++q) pk[q] = pack16<OutT>(v[q]);
      const int64_t m_even = m0 + wm * 64 + m4 * 16 + (fr & ~1);
#pragma unroll
      for (int pq = 0; pq < NGRP / 2; ++pq) {
        const uint4 own_e = pk[2 * pq], own_o = pk[2 * pq + 1];
        const uint4 dn = g256_dpp_u4<0x111>(own_o);
        const uint4 up = g256_dpp_u4<0x101>(own_e);
        const uint4 d_even = make_uint4(odd ? dn.x : own_e.x, odd ? dn.y : own_e.y, odd ? dn.z : own_e.z, odd ? dn.w : own_e.w);
        const uint4 d_odd = make_uint4(odd ? own_o.x : up.x, odd ? own_o.y : up.y, odd ? own_o.z : up.z, odd ? own_o.w : up.w);
        if (pcol[pq] + GC <= N) {
          if (m_even < M) st_global16(out + m_even * N + pcol[pq], d_even);
          if (m_even + 1 < M) st_global16(out + (m_even + 1) * N + pcol[pq], d_odd);
        }
      }
    }
    return;
  }
  // ---- epilogue through LDS: 16-byte coalesced stores --------------------------------------------------------
  // An MFMA accumulator holds 4 rows x 1 column per lane, so direct stores are 2- or 4-byte pieces (32-64 B
  // segments).  Each wave instead parks 32 rows x 128 columns of f32 in its private 16.5 KiB patch of the (now idle)
  // ring and re-reads it row-major: lane l then owns 8 consecutive columns of one row -> 16 B (bf16) / 2 x 16 B
  // (f32) stores, 256-512 contiguous bytes per row, with bias / ReLU / addend applied on the way out.
  __syncthreads();                                  // every wave is done reading the ring
  constexpr int PLD = 132;                          // padded patch row (floats): conflict-free column writes
  float* patch = reinterpret_cast<float*>(smem) + wave * (32 * PLD);
  // a lane owns two groups of 4 consecutive columns of a row.  bf16 out: adjacent groups -> one 16-byte store, 16 lanes =
  // the row's 256 bytes.  f32 out: groups 64 columns apart -> two 16-byte stores, each 16 lanes = 256 contiguous bytes
  // (see gemm256p.hip: the CU's write path retires about one request per 5 cycles and bounds this epilogue)
  constexpr bool F32O = sizeof(OutT) == 4;
  const int orow = lane >> 4;
  const int oc0 = (lane & 15) * (F32O ? 4 : 8);
  const int oc1 = F32O ? oc0 + 64 : oc0 + 4;
  const int nc0 = n0 + wn * 128 + oc0, nc1 = n0 + wn * 128 + oc1;
  const bool ok0 = nc0 + 4 <= N, ok1 = nc1 + 4 <= N;
  float bv[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bv[j] = (bias && nc0 + j < N) ? bias[nc0 + j] : 0.f;
    bv[4 + j] = (bias && nc1 + j < N) ? bias[nc1 + j] : 0.f;
  }
  // (inline-asm DMA: hipcc guards every use of a loaded register with s_waitcnt vmcnt(0), which inside the loop below also
  // waits for the stores of the previous iteration; an empty asm makes that one wait happen here)
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(bv[j]));
  const uint32_t m0_mod = add_mode == 1 ? (uint32_t)(m0 % seq_len) : 0u;       // one 64-bit modulo per tile
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) patch[(mt * 16 + fg * 4 + r) * PLD + nt * 16 + fr] = acc[half * 2 + mt][nt][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int prow = it * 4 + orow;               // 0..31
      const int lrow = wm * 64 + half * 32 + prow;
      const int64_t m = m0 + lrow;
      const float4 v0 = *reinterpret_cast<const float4*>(patch + prow * PLD + oc0);
      const float4 v1 = *reinterpret_cast<const float4*>(patch + prow * PLD + oc1);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      if (m >= M) continue;
      if constexpr (SCALED) {
        const float rsc = row_scale[m];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= rsc;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] += bv[j];
        if (relu) v[j] = fmaxf(v[j], 0.f);
      }
      const int64_t arow = add_mode == 1 ? (int64_t)((m0_mod + (uint32_t)lrow) % (uint32_t)seq_len) : m;
      if (!RAGGED_N) {                              // N % 8 == 0: whole 4-column groups, 16-byte aligned rows
        if (add_mode) {
          float av[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (sizeof(AddT) == 2) {
            uint2 u0 = make_uint2(0u, 0u), u1 = u0;
            if (ok0) u0 = *reinterpret_cast<const uint2*>(addend + arow * N + nc0);
            if (ok1) u1 = *reinterpret_cast<const uint2*>(addend + arow * N + nc1);
            av[0] = __uint_as_float(u0.x << 16); av[1] = __uint_as_float(u0.x & 0xffff0000u);
            av[2] = __uint_as_float(u0.y << 16); av[3] = __uint_as_float(u0.y & 0xffff0000u);
            av[4] = __uint_as_float(u1.x << 16); av[5] = __uint_as_float(u1.x & 0xffff0000u);
            av[6] = __uint_as_float(u1.y << 16); av[7] = __uint_as_float(u1.y & 0xffff0000u);
          } else {
            if (ok0) unpack16<float>(ld_global16(addend + arow * N + nc0), av);
            if (ok1) unpack16<float>(ld_global16(addend + arow * N + nc1), av + 4);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += av[j];
        }
        if (F32O) {
          if (ok0) st_global16(out + m * N + nc0, pack16<float>(v));
          if (ok1) st_global16(out + m * N + nc1, pack16<float>(v + 4));
        } else if (ok1) {
          st_global16(out + m * N + nc0, pack16<bf16_t>(v));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int n = (j < 4 ? nc0 : nc1 - 4) + j;
          if (n >= N) continue;
          float x = v[j];
          if (add_mode) x += DT<AddT>::ld(addend + arow * N + n);
          DT<OutT>::st(out + m * N + n, x);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename T, typename OutT, typename AddT>
static int launch_gemm256(const void* A, const void* W, const float* bias, const void* addend, void* out, int64_t M,
                          int N, int K, int relu, int add_mode, int seq_len, hipStream_t st,
                          const float* row_scale = nullptr) {
  const int tm = cdiv(M, 256), tn = cdiv(N, 256);
  const int64_t nsup = (int64_t)((tm + 7) / 8) * ((tn + 3) / 4);
  const unsigned grid = (unsigned)(((nsup + 7) / 8) * 8 * 32);
  const int lds = 8 * 32 * 132 * 4;     // epilogue patches (135 168 B) >= the 4 x 32 KiB ring
  auto kern = (N & 7) ? gemm256_kernel<T, OutT, AddT, true> : gemm256_kernel<T, OutT, AddT, false>;
  if (!((N & 7) ? xml_lds_attr_once<gemm256_kernel<T, OutT, AddT, true>>(lds)
                : xml_lds_attr_once<gemm256_kernel<T, OutT, AddT, false>>(lds)))
    return XML_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, (const T*)A, (const T*)W, bias, (const AddT*)addend,
                     (OutT*)out, M, N, K, relu, add_mode, seq_len, tm, tn, row_scale);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

bool xmli_gemm256_eligible(int64_t M, int N, int K, int dt) {
  const size_t kb = (size_t)K * dt_size(dt);
  return M >= 256 && N >= 128 && kb % 128 == 0 && kb >= 256;
}

// split-f16 projection: A' (M, K3) / W' (N, K3) f16, K3 = 3 x the logical K; out / addend f32
bool xmli_gemm256_f16s_eligible(int64_t M, int N, int K3) {
  return M >= 256 && N >= 128 && N % 8 == 0 && (K3 * 2) % 128 == 0 && K3 * 2 >= 256;
}
int xmli_gemm256_f16s(const void* A, const void* W, const float* bias, const void* addend, void* out,
                      const float* row_scale, int64_t M, int N, int K3, int relu, int add_mode, int seq_len,
                      hipStream_t st) {
  return launch_gemm256<f16_t, float, float>(A, W, bias, addend, out, M, N, K3, relu, add_mode, seq_len, st, row_scale);
}

int xmli_gemm256(const void* A, const void* W, const float* bias, const void* addend, void* out, int64_t M, int N,
                 int K, int relu, int add_mode, int seq_len, int out_f32, int dt, hipStream_t st) {
  if (dt == XML_F32)
    return launch_gemm256<float, float, float>(A, W, bias, addend, out, M, N, K, relu, add_mode, seq_len, st);
  if (out_f32)
    return launch_gemm256<bf16_t, float, bf16_t>(A, W, bias, addend, out, M, N, K, relu, add_mode, seq_len, st);
  return launch_gemm256<bf16_t, bf16_t, bf16_t>(A, W, bias, addend, out, M, N, K, relu, add_mode, seq_len, st);
}
