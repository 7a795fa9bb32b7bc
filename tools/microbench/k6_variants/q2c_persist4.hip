// K6, persistent fused variant with FOUR waves per workgroup (one per SIMD), each owning a 128 x 128 quadrant of
// the 256 x 256 tile in 256 accumulator registers.  Same ring / DMA stream / epilogue as q2c_persist.hip (8 waves,
// 64 x 128 per wave); what changes is the LDS traffic per MFMA: a wave reads (128 + 128) rows x 64 B per slice for
// 64 MFMAs instead of (64 + 128) rows for 32, i.e. 64 KiB instead of 96 KiB of ds_read per slice and CU next to the
// 32 KiB the DMA writes, and a barrier joins 4 waves instead of 8.  One wave per SIMD has nobody to hide behind, so
// every LDS read is issued a full 32-MFMA block ahead of its use.
//
// (original description of the schedule, q2c_persist.hip:)
// K6, persistent fused variant: one workgroup per CU walks a static list of
// 256 x 256 tiles, both modalities of a tile back to back, with ONE continuous LDS-DMA stream.
//
// Why (measured, profiles/r01_k6_notes.md): with one launch-time workgroup per tile, ~10 us of every ~26 us tile
// was fixed cost -- workgroup launch on a CU that can hold only one (128 KiB of LDS), a cold DMA pipeline
// (~2.2 us round trip), epilogue + drain.  t(K) = 17.4 ms + 1.11 ms per 32-wide K slice at the TVR shape: 40 % of
// the kernel was not the K loop.  Here
//   * 256 workgroups stay resident; the DMA unit stream (see q2c_ring.hip for the ring / phase / stagger schedule,
//     which is unchanged) runs LEAD units ahead of the MFMA phases ACROSS modality and tile boundaries, so the
//     pipeline never drains and the epilogue of a tile overlaps the loads of the next one;
//   * video and sub scores of a tile are produced back to back and combined in registers:
//     out = (max_l s_video + max_l s_sub) * 0.5  -- one plain store per (query, video), no read-modify-write;
//   * the clip masks of a tile arrive by the same DMA stream (1 KiB) and are read from LDS in the epilogue, so the
//     only VMEM ops besides the stream are the result stores (a compiler-visible global load would make hipcc
//     drain the hand-counted stream with vmcnt(0));
//   * DMA addressing is SGPR base (tile / slice, scalar adds) + loop-invariant 32-bit VGPR row offset.
// Tile order: workgroup j runs on XCD j % 8 (observed, speed only).  The 32 workgroups of an XCD form an
// 8 (query tiles) x 4 (clip tiles) super-tile; an XCD keeps its query group and walks clip groups, so the 3 MiB of
// query operands stay in that XCD's 4 MiB L2 and only clip tiles stream in (6 % instead of 19 % line misses).
#include <type_traits>

#include "common.h"

struct Q2cPersist4Args {
  const void* qn[2];
  const void* cn[2];
  const float* mask[2];
  float* out;
  int64_t ld_out;
  int nq, nv, hidden, n_mod, tq, tc;
};

__device__ __forceinline__ void dma16q(uint32_t voff, const char* sbase, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

// same, with the non-temporal hint: streamed clip tiles should not displace the L2-resident query group
__device__ __forceinline__ void dma16q_nt(uint32_t voff, const char* sbase, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

// MFMA with the accumulator pinned to AGPRs.  With 256 accumulator registers per lane hipcc keeps the builtin's C/D
// operand in AGPRs but copies it to VGPRs and back around EVERY v_mfma (9 v_accvgpr moves per MFMA, measured
// 704 TF); the "a" constraint makes the instruction use the AGPRs directly.  INIT: C = 0 (first slice of a segment).
// Hazards the compiler no longer sees: the epilogue waits explicitly before reading the accumulators.
typedef unsigned int u32x4_v __attribute__((ext_vector_type(4)));
template <typename T, bool INIT> struct MmaAcc;
template <bool INIT> struct MmaAcc<bf16_t, INIT> {
  __device__ static __forceinline__ void chunk(f32x4& acc, const u32x4_v& a, const u32x4_v& b) {
    if constexpr (INIT) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  }
};
template <bool INIT> struct MmaAcc<float, INIT> {
  __device__ static __forceinline__ void chunk(f32x4& acc, const u32x4_v& a, const u32x4_v& b) {
    if constexpr (INIT)
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(a.x), "v"(b.x));
    else
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a.x), "v"(b.x));
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a.y), "v"(b.y));
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a.z), "v"(b.z));
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a.w), "v"(b.w));
  }
};

__device__ __forceinline__ int swz4q(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

// Specialised to lpad == 128 (one video per 128-column group -- the TVR shape): no per-column divisions, one
// reduction per accumulator row.  Other clip paddings use the per-modality kernels (q2c_ring.hip / q2c256.hip).
//
// K-loop schedule ("one barrier per slice"; measured reason in profiles/r01_k6_notes.md: with two barriers per
// 16-MFMA phase the waves spent 41 % of their cycles parked at barriers / waitcnts and issued 2.4 SALU per MFMA):
//   slice g (32 K-elements; ring slot g & 3) is consumed as two 16-MFMA halves h0 (column tiles 0-3) and h1 (4-7);
//   fragments are double-buffered in registers, so LDS reads always run under the other half's MFMAs:
//
//     read  fbH <- B[4..7](g)
//     MFMA  h0(g):  acc[:,0..3] += fa x fbL
//     s_waitcnt vmcnt(8)      my DMAs of slice g+1 have landed (slices g+2, g+3 may still fly)
//     s_waitcnt lgkmcnt(0)    all my LDS reads of slice g have returned
//     s_barrier               => slice g+1 is readable by everyone, slot g & 3 is free for everyone
//     DMA   slice g+4 -> slot g & 3          (3 slices = 96 KiB in flight per CU)
//     read  fa' <- A(g+1), fbL <- B[0..3](g+1)
//     MFMA  h1(g):  acc[:,4..7] += fa x fbH
//     fa <-> fa'
//
//   No wave-group stagger and no s_setprio: the two waves of a SIMD drift apart by themselves and keep the MFMA
//   pipe busy from either wave's ready cluster.
template <typename T, int ABL = 0>   // ABL (perf ablations only): 1 no DMA after the prologue, 2 no MFMA
__global__ __launch_bounds__(256, 1) void q2c_persist4_kernel(Q2cPersist4Args a) {
  constexpr int ROWB = 64;
  constexpr int OPER_BYTES = 256 * ROWB;
  constexpr int SLOT_BYTES = 2 * OPER_BYTES;
  constexpr int RING_BYTES = 4 * SLOT_BYTES;
  constexpr int MASK_OFF = RING_BYTES;            // 2 x 1 KiB mask patches (256 columns x f32)
  constexpr int STASH_OFF = RING_BYTES + 2048;    // 256 rows x 2 videos f32: modality-0 maxima of the current tile
  constexpr int PATCH_OFF = RING_BYTES + 4096;    // 4 x 4 KiB: per-wave bounce patch of the epilogue
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int xcd = blockIdx.x & 7;
  const int qt_off = (blockIdx.x >> 3) & 7, ct_off = blockIdx.x >> 6;   // position inside the 8 x 4 super-tile
  const int k_bytes = a.hidden * (int)sizeof(T);
  const int slices_per_seg = k_bytes / ROWB;      // even (k_bytes % 128 == 0)
  const int n_qgroups = (a.tq + 7) >> 3;
  const int cr = (((a.tc + 3) >> 2) + 7) >> 3;    // rounds per query group on one XCD

  // tile of (query group g, round c):  qt = 8 g + qt_off,  ct = 4 (8 c + xcd) + ct_off
  auto tile_valid = [&](int g, int c) -> bool { return (8 * g + qt_off) < a.tq && (4 * (8 * c + xcd) + ct_off) < a.tc; };
  auto advance = [&](int& g, int& c) {            // next valid (g, c) in walk order, g == n_qgroups when exhausted
    do {
      if (++c == cr) { c = 0; ++g; }
    } while (g < n_qgroups && !tile_valid(g, c));
  };

  // ---- issue side (DMA stream, runs up to 3 slices ahead of the slice being computed) ---------------------
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t lds_wave = lds0 + wave * 4096;
  int i_g = 0, i_c = -1, i_mod = 0, i_slice = 0, i_seg = 0;
  uint32_t i_gs = 0;                               // slices issued so far (global) -> ring slot
  uint32_t voff_a[4] = {0, 0, 0, 0}, voff_b[4] = {0, 0, 0, 0};
  const char* sbase_a = nullptr;
  const char* sbase_b = nullptr;

  auto setup_issue_segment = [&](bool new_tile) {
    const int q0 = (8 * i_g + qt_off) * 256, v0 = (4 * (8 * i_c + xcd) + ct_off) * 2;
    int lane_o = lane;                              // opaque copy: keeps LICM from hoisting (and keeping live across
    asm volatile("" : "+v"(lane_o));                // the MFMA loop) everything derived from the lane id below
    if (new_tile) {
      const int rsub = lane_o >> 2, pslot = lane_o & 3;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 16 + rsub;                     // 0..255
        const int slot = pslot ^ swz4q(row);
        const int qrow = (q0 + row < a.nq) ? row : 0;                   // clamp to the tile's first row
        const int brow = (v0 + (row >> 7) < a.nv) ? row : (row & 127);  // second video missing: re-read the first
        const uint32_t va = (uint32_t)qrow * k_bytes + slot * 16;
        const uint32_t vb = (uint32_t)brow * k_bytes + slot * 16;
        voff_a[i] = va;
        voff_b[i] = vb;
      }
    }
    sbase_a = reinterpret_cast<const char*>(a.qn[i_mod]) + (int64_t)q0 * k_bytes;
    sbase_b = reinterpret_cast<const char*>(a.cn[i_mod]) + (int64_t)v0 * 128 * k_bytes;
    if (wave == 0) {   // mask patch of the segment: lane l carries columns 4 l .. 4 l + 3 of the tile's 256 columns
      const int mrow = (v0 + (lane_o >> 5) < a.nv) ? lane_o : (lane_o & 31);
      const char* sbase_m = reinterpret_cast<const char*>(a.mask[i_mod]) + (int64_t)v0 * 128 * 4;
      dma16q((uint32_t)mrow * 16, sbase_m, lds0 + MASK_OFF + (i_seg & 1) * 1024);
    }
  };
  // One slice = 8 DMA instructions (this wave's 64 rows of A and of B) + bookkeeping; issued piecewise so that the
  // single wave of a SIMD can slot them between MFMAs (a 16-cycle MFMA leaves ~3 issue slots).
  auto issue_dma = [&](int i) {
    if (i_g >= n_qgroups) return;
    if (ABL == 1 && i_gs >= 4) return;
    const int koff = i_slice * ROWB;
    const uint32_t dst = lds_wave + (i_gs & 3) * SLOT_BYTES;
    if (i < 4) dma16q(voff_a[i], sbase_a + koff, dst + i * 1024);
    else if (ABL == 3) dma16q_nt(voff_b[i - 4], sbase_b + koff, dst + OPER_BYTES + (i - 4) * 1024);
    else dma16q(voff_b[i - 4], sbase_b + koff, dst + OPER_BYTES + (i - 4) * 1024);
  };
  auto issue_finish = [&]() {
    if (i_g >= n_qgroups) return;
    ++i_gs;
    if (++i_slice == slices_per_seg) {   // next segment: other modality of the tile, or the next tile
      i_slice = 0;
      ++i_seg;
      bool new_tile = false;
      if (++i_mod == a.n_mod) {
        i_mod = 0;
        advance(i_g, i_c);
        new_tile = true;
      }
      if (i_g < n_qgroups) setup_issue_segment(new_tile);
    }
  };
  auto issue_slice = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_dma(i);
    issue_finish();
  };

  advance(i_g, i_c);
  if (i_g >= n_qgroups) return;
  setup_issue_segment(true);

  // ---- compute side ---------------------------------------------------------------------------------------------
  const int a_off = (wm * 128 + fr) * ROWB + ((fg ^ swz4q(fr)) << 4);
  const int b_off = OPER_BYTES + (wn * 128 + fr) * ROWB + ((fg ^ swz4q(fr)) << 4);
  int c_g = i_g, c_c = i_c, c_mod = 0, c_seg = 0;
  uint32_t c_gs = 0;                               // global index of the slice being computed
  bool more = true;                                // a slice c_gs + 1 exists

  // prologue: slices 0..3 in flight, slice 0 landed, first fragments in registers.  The issue side needs
  // slices_per_seg >= 4 here (no segment end inside the first 3 issues is required; 4th may end a segment).
  issue_slice(); issue_slice(); issue_slice(); issue_slice();
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  u32x4_v faA[8], faB[8], fbL[4], fbH[4];
  {
    const char* slot = smem;
#pragma unroll
    for (int m = 0; m < 8; ++m) faA[m] = *reinterpret_cast<const u32x4_v*>(slot + a_off + m * 16 * ROWB);
#pragma unroll
    for (int n = 0; n < 4; ++n) fbL[n] = *reinterpret_cast<const u32x4_v*>(slot + b_off + n * 16 * ROWB);
  }

  for (;;) {      // one iteration = one (tile, modality) segment
    f32x4 acc[8][8];       // written by the INIT MFMAs of the segment's first slice

    auto slice_step = [&](u32x4_v (&fc)[8], u32x4_v (&fn)[8], auto init_tag) {
      constexpr bool INIT = decltype(init_tag)::value;
      const char* slot = smem + (c_gs & 3) * SLOT_BYTES;
      // h0: 32 MFMAs on column tiles 0-3; the 4 reads of column tiles 4-7 ride behind the first four
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int m = i >> 2, n = i & 3;
        if (ABL == 2) asm volatile("" ::"v"(fc[m].x), "v"(fbL[n].x), "v"(fc[m].w), "v"(fbL[n].w));
        else MmaAcc<T, INIT>::chunk(acc[m][n], fc[m], fbL[n]);
        if (i < 4) fbH[i] = *reinterpret_cast<const u32x4_v*>(slot + b_off + (i + 4) * 16 * ROWB);
        __builtin_amdgcn_sched_barrier(0);
      }
      {   // slice c_gs + 1 must have landed (mine) before the barrier; later slices may stay in flight
        const int fly = (int)(i_gs - (c_gs + 2));
        if (fly >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (fly == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0), as a builtin: hipcc must KNOW the fbH reads have
      __builtin_amdgcn_s_barrier();                     // returned, or it waits for the reads issued below before h1
      ++c_gs;
      const char* nslot = smem + (c_gs & 3) * SLOT_BYTES;
      // h1: 32 MFMAs on column tiles 4-7; behind MFMA i: read i of the next slice's fragments (12 reads), then the
      // 8 DMA pieces of slice c_gs + 3 (into the slot the barrier just released), one per second MFMA
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int m = i >> 2, n = i & 3;
        if (ABL == 2) asm volatile("" ::"v"(fc[m].x), "v"(fbH[n].x), "v"(fc[m].w), "v"(fbH[n].w));
        else MmaAcc<T, INIT>::chunk(acc[m][n + 4], fc[m], fbH[n]);
        if (more) {
          if (i < 8) fn[i] = *reinterpret_cast<const u32x4_v*>(nslot + a_off + i * 16 * ROWB);
          else if (i < 12) fbL[i - 8] = *reinterpret_cast<const u32x4_v*>(nslot + b_off + (i - 8) * 16 * ROWB);
        }
        if (i >= 12 && i < 28 && !(i & 1)) issue_dma((i - 12) >> 1);
        if (i == 28) issue_finish();
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // slices_per_seg is even and >= 6: the first pair initialises the accumulators, the last pair looks ahead
    slice_step(faA, faB, std::true_type{});
    slice_step(faB, faA, std::false_type{});
    for (int c_slice = 2; c_slice < slices_per_seg; c_slice += 2) {
      slice_step(faA, faB, std::false_type{});
      if (c_slice + 2 >= slices_per_seg) {            // the slice after the next one closes the segment:
        // does a further segment exist?  (compute-side lookahead of the walk, scalar only)
        int ng = c_g, nc = c_c;
        bool has_next = c_mod + 1 < a.n_mod;
        if (!has_next) { advance(ng, nc); has_next = ng < n_qgroups; }
        more = has_next;
      }
      slice_step(faB, faA, std::false_type{});
    }
    // the MFMAs are opaque to the hazard recogniser: let the last ones retire before the accumulators are read
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // ---- end of a (tile, modality) segment: mask_logits + max over the video's 128 clips, inside the wave -----
    {
      const int q0 = (8 * c_g + qt_off) * 256, vid = (4 * (8 * c_c + xcd) + ct_off) * 2 + wn;
      int fr_e = fr, fg_e = fg, lane_e = lane;      // opaque copies (see setup_issue_segment)
      asm volatile("" : "+v"(fr_e), "+v"(fg_e), "+v"(lane_e));
      const float* mpatch = reinterpret_cast<const float*>(smem + MASK_OFF + (c_seg & 1) * 1024) + wn * 128 + fr_e;
      float* stash = reinterpret_cast<float*>(smem + STASH_OFF) + wn;
      const bool last_mod = c_mod == a.n_mod - 1;
      const bool vid_ok = vid < a.nv;
      float mk[8];
#pragma unroll
      for (int n = 0; n < 8; ++n) mk[n] = vid_ok ? mpatch[n * 16] : 0.f;
      // The accumulators leave the AGPRs through LDS (ds_write_b128 takes AGPR data on gfx90a+): every use of `acc`
      // then is an AGPR use.  With v_accvgpr_read in C++ the register allocator splits ~160 accumulator live ranges
      // into VGPRs at the loop exit and pays for it by spilling the DMA row offsets INSIDE the K loop (scratch
      // reloads + vmcnt waits in front of every DMA).  4 tiles (4 KiB per wave) at a time.
      const uint32_t patch = lds0 + PATCH_OFF + wave * 4096 + lane_e * 16;
      const char* patch_rd = smem + PATCH_OFF + wave * 4096 + lane_e * 16;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        f32x4 t[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(patch), "a"(acc[m][h * 4 + j]), "i"(j * 1024) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int j = 0; j < 4; ++j) t[h * 4 + j] = *reinterpret_cast<const f32x4*>(patch_rd + j * 1024);
          __builtin_amdgcn_s_waitcnt(0xc07f);       // the patch is rewritten by the next batch
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lrow = wm * 128 + m * 16 + fg_e * 4 + r;
          float mx = -INFINITY;
#pragma unroll
          for (int n = 0; n < 8; ++n)
            mx = fmaxf(mx, t[n][r] * mk[n] + (1.f - mk[n]) * -1e10f);   // mask_logits, xml/model_xml.py:640-641
          float red = lane16_max_dpp(mx);
          if (fr_e == 0) {
            if (!last_mod) {
              stash[lrow * 2] = red;
            } else {
              if (a.n_mod == 2) red = (stash[lrow * 2] + red) * 0.5f;        // (video + sub) / 2, xml/model_xml.py:574
              if (q0 + lrow < a.nq && vid_ok) a.out[(int64_t)(q0 + lrow) * a.ld_out + vid] = red;
            }
          }
          __builtin_amdgcn_sched_barrier(0);   // one row at a time: keeps the epilogue's register peak low
        }
      }
    }
    ++c_seg;
    if (++c_mod == a.n_mod) {
      c_mod = 0;
      advance(c_g, c_c);
      if (c_g >= n_qgroups) break;
    }
  }
}

template <typename T>
static int launch_q2c_persist4(const Q2cPersist4Args& a, hipStream_t st) {
  const int lds = 4 * 2 * 256 * 64 + 2048 + 2048 + 4 * 4096;
  extern int g_q2c_ablation;
  auto kern = g_q2c_ablation == 1 ? q2c_persist4_kernel<T, 1> : g_q2c_ablation == 2 ? q2c_persist4_kernel<T, 2>
             : g_q2c_ablation == 3 ? q2c_persist4_kernel<T, 3> : q2c_persist4_kernel<T, 0>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
    return XML_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, st, a);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// Requirements (checked by the caller, which otherwise uses the per-modality kernels): lpad == 128,
// hidden * sizeof(T) a multiple of 128 bytes (an even number of 64-byte slices) and at least 6 slices:
// the mask patch of segment s+2 is fetched 4 slices ahead and must not land before the epilogue of segment s.
int xmli_q2c_scores_persist4(int n_mod, const void* const* qn, const void* const* cn, const float* const* mask,
                            float* out, int64_t ld_out, int nq, int nv, int lpad, int hidden, int dt, hipStream_t st) {
  Q2cPersist4Args a;
  for (int m = 0; m < 2; ++m) {
    a.qn[m] = qn[m < n_mod ? m : 0];
    a.cn[m] = cn[m < n_mod ? m : 0];
    a.mask[m] = mask[m < n_mod ? m : 0];
  }
  if (lpad != 128) return XML_ERR_UNSUPPORTED;
  a.out = out; a.ld_out = ld_out; a.nq = nq; a.nv = nv; a.hidden = hidden; a.n_mod = n_mod;
  a.tq = cdiv(nq, 256); a.tc = cdiv(nv, 2);
  if (dt == XML_BF16) return launch_q2c_persist4<bf16_t>(a, st);
  return launch_q2c_persist4<float>(a, st);
}
