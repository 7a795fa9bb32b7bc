// K6, pipelined variant: 256 x 256 tile, 8 waves, 4-deep LDS ring of 32-element K slices fed by LDS-DMA with a
// COUNTED vmcnt, two staggered wave groups (one runs its MFMA cluster while the other reads LDS / issues DMA).
//
// Measured motivation (profiles/r01_k6_notes.md): with one 64 KiB stage in flight and `vmcnt(0)` + barrier per K
// tile, the K-tile period equals the DMA round trip (~2.2 us under load: 18 % of the lines miss L2), i.e. the
// kernel is latency-bound at 33 % of the MFMA peak although L2/HBM bandwidth and LDS are far from saturated; the
// MFMA+LDS structure alone (no DMA) reaches 46 %.  This kernel keeps ~2.5 slices (80 KiB) of DMA in flight per CU
// and never drains it inside the K loop.
//
// Ring / schedule
//   slice  = 32 K-elements (bf16; 16 for f32) = 64 bytes per row; a ring slot holds A (256 rows) + B (256 rows)
//            = 32 KiB; 4 slots = 128 KiB.
//   unit   = half a slot (A part or B part, 16 KiB) = 2 global_load_lds_dwordx4 per lane per wave.
//   phase p (two per slice: h = p & 1):
//        L(p): ds_read the fragments of slice p >> 1 needed by M(p)  [h = 0: A (4) + B tiles 0-3 (4); h = 1: B 4-7 (4)]
//              issue DMA unit p + 5;  if h == 1: s_waitcnt vmcnt(6)  (slice (p + 1) / 2 has landed, 3 units may fly)
//              s_barrier
//        M(p): 16 MFMAs under s_setprio(1);  s_barrier
//   Waves 4-7 execute one extra s_barrier up front, so in every barrier interval one group is in L and the other
//   in M; waves w and w + 4 share a SIMD, i.e. every SIMD always has one MFMA-issuing and one loading wave.
//   Hazards: a slice is read (L(2s), L(2s+1)) only after every wave's counted vmcnt for it and one more barrier;
//   a slot is re-filled (unit issue in L(2s+1+...)) only after the barrier that follows every wave's M(2s-1), whose
//   MFMAs consumed the last registers loaded from it.  Lead of 5 units is the largest that satisfies the second.
//   LDS image: lane-linear per DMA instruction (16 rows x 64 B); 16-byte slot j of row r lives at j ^ g((r >> 2) & 3),
//   g = {0,2,3,1}, which makes every ds_read_b128 lane group hit 16 distinct 16-byte bank slots; the permutation is
//   applied on the per-lane SOURCE address of the DMA and again on the read (guide rule 21).
#include "common.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// LDS-DMA issued from inline asm: hipcc models a compiler-visible global_load_lds as an LDS store and then drains
// it (s_waitcnt vmcnt(0)) in front of every ds_read that may alias it -- i.e. in front of every K step of this
// pipeline.  Hidden in asm, the DMA is counted by hand (the s_waitcnt vmcnt(N) below are the only waits on it).
// M0 carries the wave-uniform LDS byte address; lane l lands at M0 + 16 l (guide 5.7, glds16 recipe).
__device__ __forceinline__ void dma16(const char* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

__device__ __forceinline__ int swz4(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

template <typename T, int ABL = 0>   // ABL (perf ablations only): 1 no DMA in the loop, 2 no MFMA, 3 no ds_read in the loop
__global__ __launch_bounds__(512, 2) void q2c_scores_kernel_ring(const T* __restrict__ qn, const T* __restrict__ cn,
                                                                 const float* __restrict__ mask,
                                                                 float* __restrict__ out, int64_t ld_out, int nq,
                                                                 int nv, int lpad, int hidden, int combine, int tq,
                                                                 int tc) {
  constexpr int ROWB = 64;                  // bytes of K per row per slice
  constexpr int OPER_BYTES = 256 * ROWB;    // 16 KiB
  constexpr int SLOT_BYTES = 2 * OPER_BYTES;
  constexpr int LEAD = 5;                   // DMA units in front of the phase being computed
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int b = blockIdx.x;
  const int xcd = b & 7, local = b >> 3;
  const int sup = (local >> 5) * 8 + xcd;
  const int w32 = local & 31;
  const int sq = (tq + 7) >> 3;
  const int qt = (sup % sq) * 8 + (w32 & 7);
  const int ct = (sup / sq) * 4 + (w32 >> 3);
  if (qt >= tq || ct >= tc) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int vpg = 128 / lpad;
  const int gcols = vpg * lpad;
  const int q0 = qt * 256;
  const int v0 = ct * 2 * vpg;
  const int k_bytes = hidden * (int)sizeof(T);
  const int n_units = 2 * (k_bytes / ROWB);   // two units (A, B) per slice

  // ---- DMA sources: per unit each wave moves row groups 2*wave and 2*wave+1 (16 rows x 64 B each) --------
  const char* a_src[2];
  const char* b_src[2];
  {
    const int rsub = lane >> 2, pslot = lane & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave * 2 + i) * 16 + rsub;
      const int slot = pslot ^ swz4(row);
      int qrow = q0 + row;
      qrow = qrow < nq ? qrow : nq - 1;
      a_src[i] = reinterpret_cast<const char*>(qn) + (int64_t)qrow * k_bytes + slot * 16;
      const int grp = row >> 7, col = row & 127;
      int vid = v0 + grp * vpg + col / lpad;
      int clip = col % lpad;
      if (col >= gcols || vid >= nv) { vid = 0; clip = 0; }
      b_src[i] = reinterpret_cast<const char*>(cn) + ((int64_t)vid * lpad + clip) * k_bytes + slot * 16;
    }
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 2048;
  auto issue_unit = [&](int u) {
    const int slice = u >> 1;
    const uint32_t dst = lds_base + (slice & 3) * SLOT_BYTES + (u & 1) * OPER_BYTES;
    const int koff = slice * ROWB;
    if (u & 1) {
      dma16(b_src[0] + koff, dst);
      dma16(b_src[1] + koff, dst + 1024);
    } else {
      dma16(a_src[0] + koff, dst);
      dma16(a_src[1] + koff, dst + 1024);
    }
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // read offsets inside a slot: row = base + 16 t + fr -> swizzle depends on fr only
  const int a_off = (wm * 64 + fr) * ROWB + ((fg ^ swz4(fr)) << 4);
  const int b_off = OPER_BYTES + (wn * 128 + fr) * ROWB + ((fg ^ swz4(fr)) << 4);

  // ---- prologue -------------------------------------------------------------------------------------------
#pragma unroll
  for (int u = 0; u < LEAD; ++u)
    if (u < n_units) issue_unit(u);
  __builtin_amdgcn_s_waitcnt(0x0f76);       // vmcnt(6): units 0, 1 (slice 0) landed
  __builtin_amdgcn_s_barrier();
  if (wave >= 4) __builtin_amdgcn_s_barrier();   // stagger the second wave group by one barrier interval

  uint4 fa[4], fb[4];
  const int n_phases = n_units;             // one phase per unit: 2 per slice
  for (int p = 0; p < n_phases; p += 2) {
    const char* slot = smem + ((p >> 1) & 3) * SLOT_BYTES;
    // ---------------- phase p (h = 0) ----------------
    if (ABL != 3 || p == 0) {
#pragma unroll
      for (int m = 0; m < 4; ++m) fa[m] = *reinterpret_cast<const uint4*>(slot + a_off + m * 16 * ROWB);
#pragma unroll
      for (int n = 0; n < 4; ++n) fb[n] = *reinterpret_cast<const uint4*>(slot + b_off + n * 16 * ROWB);
    }
    if (p + LEAD < n_units && ABL != 1) issue_unit(p + LEAD);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (ABL == 2) { asm volatile("" ::"v"(fa[m].x), "v"(fb[n].x), "v"(fa[m].w), "v"(fb[n].w)); }
        else Mma<T>::chunk(acc[m][n], fa[m], fb[n]);
      }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase p + 1 (h = 1) ----------------
    if (ABL != 3) {
#pragma unroll
      for (int n = 0; n < 4; ++n) fb[n] = *reinterpret_cast<const uint4*>(slot + b_off + (n + 4) * 16 * ROWB);
    }
    if (p + 1 + LEAD < n_units && ABL != 1) issue_unit(p + 1 + LEAD);
    {   // next slice (units p+2, p+3) must have landed; units issued after them may stay in flight
      const int newest = min(p + 1 + LEAD, n_units - 1);
      const int fly = newest - (p + 3);     // units allowed outstanding
      if (fly >= 3) __builtin_amdgcn_s_waitcnt(0x0f76);
      else if (fly == 2) __builtin_amdgcn_s_waitcnt(0x0f74);
      else if (fly == 1) __builtin_amdgcn_s_waitcnt(0x0f72);
      else __builtin_amdgcn_s_waitcnt(0x0f70);
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (ABL == 2) { asm volatile("" ::"v"(fa[m].x), "v"(fb[n].x), "v"(fa[m].w), "v"(fb[n].w)); }
        else Mma<T>::chunk(acc[m][n + 4], fa[m], fb[n]);
      }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  }
  if (wave < 4) __builtin_amdgcn_s_barrier();    // balance the stagger

  // ---- epilogue (same as the 256 kernel): mask_logits + max over each video's clips inside the wave -------
  float mk[8], fill[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int col = n * 16 + fr;
    const int vid = v0 + wn * vpg + col / lpad;
    const bool ok = col < gcols && vid < nv;
    mk[n] = ok ? mask[(int64_t)vid * lpad + (col % lpad)] : 0.f;
    fill[n] = (1.f - mk[n]) * -1e10f;
  }
  const int tpv = lpad >> 4;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = q0 + wm * 64 + m * 16 + fg * 4 + r;
      float mx = -INFINITY;
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        mx = fmaxf(mx, acc[m][n][r] * mk[n] + fill[n]);
        if ((n + 1) % tpv == 0) {
          const float red = lane16_max(mx);
          const int vid = v0 + wn * vpg + n / tpv;
          if (fr == 0 && row < nq && vid < nv && (n / tpv) < vpg) {
            float* po = out + (int64_t)row * ld_out + vid;
            *po = combine ? (*po + red) * 0.5f : red;
          }
          mx = -INFINITY;
        }
      }
    }
  }
}

template <typename T>
static int launch_q2c_ring(const void* qn, const void* cn, const float* mask, float* out, int64_t ld_out, int nq,
                           int nv, int lpad, int hidden, int combine, hipStream_t st) {
  const int vpg = 128 / lpad;
  const int tq = cdiv(nq, 256), tc = cdiv(nv, 2 * vpg);
  const int64_t nsup = (int64_t)((tq + 7) / 8) * ((tc + 3) / 4);
  const unsigned grid = (unsigned)(((nsup + 7) / 8) * 8 * 32);
  const int lds = 4 * 2 * 256 * 64;
  auto kern = q2c_scores_kernel_ring<T, 0>;
  bool ok = xml_lds_attr_once<q2c_scores_kernel_ring<T, 0>>(lds);
#ifdef XML_DEBUG_VARIANTS
  if (g_q2c_ablation == 1) { kern = q2c_scores_kernel_ring<T, 1>; ok = xml_lds_attr_once<q2c_scores_kernel_ring<T, 1>>(lds); }
  if (g_q2c_ablation == 2) { kern = q2c_scores_kernel_ring<T, 2>; ok = xml_lds_attr_once<q2c_scores_kernel_ring<T, 2>>(lds); }
  if (g_q2c_ablation == 3) { kern = q2c_scores_kernel_ring<T, 3>; ok = xml_lds_attr_once<q2c_scores_kernel_ring<T, 3>>(lds); }
#endif
  if (!ok) return XML_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, (const T*)qn, (const T*)cn, mask, out, ld_out, nq, nv, lpad,
                     hidden, combine, tq, tc);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// eligible when hidden * sizeof(T) is a multiple of 64 bytes and there are at least 3 slices
int xmli_q2c_scores_ring(const void* qn, const void* cn, const float* mask, float* out, int64_t ld_out, int nq, int nv,
                         int lpad, int hidden, int combine, int dt, hipStream_t st) {
  if (dt == XML_BF16) return launch_q2c_ring<bf16_t>(qn, cn, mask, out, ld_out, nq, nv, lpad, hidden, combine, st);
  return launch_q2c_ring<float>(qn, cn, mask, out, ld_out, nq, nv, lpad, hidden, combine, st);
}
