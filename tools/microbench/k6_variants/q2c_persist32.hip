// K6, persistent fused variant on v_mfma_f32_32x32x16_bf16 (bf16 operands only; f32 stays on q2c_persist.hip).
// Same ring / DMA stream / walk as q2c_persist.hip; only the compute side differs: a wave's 64 x 128 quadrant is
// 2 x 4 tiles of 32 x 32 (16 accumulator registers each), a 32-wide K slice is two 16-wide MFMA steps.  The
// 16x16x32 instruction issues at ~17-20 cycles per 16 K flop, the 32x32x16 one at 32 cycles per 32 K flop
// (MI355X_MICROARCH.md, throughput table): same bytes from LDS, up to 1.2x the matrix rate.
//
// (description of the shared schedule, from q2c_persist.hip:)
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
#include "common.h"

typedef float f32x16_v __attribute__((ext_vector_type(16)));

struct Q2cPersist32Args {
  const void* qn[2];
  const void* cn[2];
  const float* mask[2];
  float* out;
  int64_t ld_out;
  int nq, nv, hidden, n_mod, tq, tc;
  int qsh;     // log2 of the query tiles per XCD super-tile (0..3): the 32 workgroups of an XCD form 2^qsh x 2^(5-qsh)
};

__device__ __forceinline__ void dma16t(uint32_t voff, const char* sbase, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

// same, with the non-temporal hint: streamed clip tiles should not displace the L2-resident query group
__device__ __forceinline__ void dma16t_nt(uint32_t voff, const char* sbase, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

// value of `v` in the lane selected by a DPP control word (row_mirror 0x140, row_half_mirror 0x141, quad_perm 0x00-0xff)
template <int CTRL>
__device__ __forceinline__ float dpp_read32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ int swz4t(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

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
template <typename T, int ABL = 0, bool PHASED = true>   // ABL (perf ablations only): 1 no DMA after the prologue, 2 no MFMA
__global__ __launch_bounds__(512, 2) void q2c_persist32_kernel(Q2cPersist32Args a) {
  constexpr int ROWB = 64;
  constexpr int OPER_BYTES = 256 * ROWB;
  constexpr int SLOT_BYTES = 2 * OPER_BYTES;
  constexpr int RING_BYTES = 4 * SLOT_BYTES;
  constexpr int MASK_OFF = RING_BYTES;            // 2 x 1 KiB mask patches (256 columns x f32)
  constexpr int STASH_OFF = RING_BYTES + 2048;    // 256 rows x 2 videos f32: modality-0 maxima of the current tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int grp = wave >> 2;                      // waves w and w + 4 share a SIMD
  const int fr = lane & 15, fg = lane >> 4;
  const int xcd = blockIdx.x & 7;
  const int qsh = a.qsh, csh = 5 - a.qsh;        // super-tile = 2^qsh query tiles x 2^csh clip tiles (8 x 4 when nq is large)
  const int qt_off = (blockIdx.x >> 3) & ((1 << qsh) - 1), ct_off = blockIdx.x >> (3 + qsh);
  const int k_bytes = a.hidden * (int)sizeof(T);
  const int slices_per_seg = k_bytes / ROWB;      // even (k_bytes % 128 == 0)
  const int n_qgroups = (a.tq + (1 << qsh) - 1) >> qsh;
  const int cr = (((a.tc + (1 << csh) - 1) >> csh) + 7) >> 3;    // rounds per query group on one XCD

  // tile of (query group g, round c):  qt = 2^qsh g + qt_off,  ct = 2^csh (8 c + xcd) + ct_off
  auto tile_valid = [&](int g, int c) -> bool {
    return ((g << qsh) + qt_off) < a.tq && ((((c << 3) + xcd) << csh) + ct_off) < a.tc;
  };
  auto advance = [&](int& g, int& c) {            // next valid (g, c) in walk order, g == n_qgroups when exhausted
    do {
      if (++c == cr) { c = 0; ++g; }
    } while (g < n_qgroups && !tile_valid(g, c));
  };

  // ---- issue side (DMA stream, runs up to 3 slices ahead of the slice being computed) ---------------------
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t lds_wave = lds0 + wave * 2048;
  int i_g = 0, i_c = -1, i_mod = 0, i_slice = 0, i_seg = 0;
  uint32_t i_gs = 0;                               // slices issued so far (global) -> ring slot
  uint32_t voff_a0 = 0, voff_a1 = 0, voff_b0 = 0, voff_b1 = 0;
  const char* sbase_a = nullptr;
  const char* sbase_b = nullptr;

  auto setup_issue_segment = [&](bool new_tile) {
    const int q0 = ((i_g << qsh) + qt_off) * 256, v0 = ((((i_c << 3) + xcd) << csh) + ct_off) * 2;
    int lane_o = lane;                              // opaque copy: keeps LICM from hoisting (and keeping live across
    asm volatile("" : "+v"(lane_o));                // the MFMA loop) everything derived from the lane id below
    if (new_tile) {
      const int rsub = lane_o >> 2, pslot = lane_o & 3;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + rsub;                     // 0..255
        const int slot = pslot ^ swz4t(row);
        const int qrow = (q0 + row < a.nq) ? row : 0;                   // clamp to the tile's first row
        const int brow = (v0 + (row >> 7) < a.nv) ? row : (row & 127);  // second video missing: re-read the first
        const uint32_t va = (uint32_t)qrow * k_bytes + slot * 16;
        const uint32_t vb = (uint32_t)brow * k_bytes + slot * 16;
        if (i == 0) { voff_a0 = va; voff_b0 = vb; } else { voff_a1 = va; voff_b1 = vb; }
      }
    }
    sbase_a = reinterpret_cast<const char*>(a.qn[i_mod]) + (int64_t)q0 * k_bytes;
    sbase_b = reinterpret_cast<const char*>(a.cn[i_mod]) + (int64_t)v0 * 128 * k_bytes;
    if (wave == 0) {   // mask patch of the segment: lane l carries columns 4 l .. 4 l + 3 of the tile's 256 columns
      const int mrow = (v0 + (lane_o >> 5) < a.nv) ? lane_o : (lane_o & 31);
      const char* sbase_m = reinterpret_cast<const char*>(a.mask[i_mod]) + (int64_t)v0 * 128 * 4;
      dma16t((uint32_t)mrow * 16, sbase_m, lds0 + MASK_OFF + (i_seg & 1) * 1024);
    }
  };
  auto issue_slice = [&]() {      // 4 DMA instructions: this wave's 32 rows of A and of B
    if (i_g >= n_qgroups) return;
    const int koff = i_slice * ROWB;
    const uint32_t dst = lds_wave + (i_gs & 3) * SLOT_BYTES;
    if (ABL != 1 || i_gs < 4) {
      dma16t(voff_a0, sbase_a + koff, dst);
      dma16t(voff_a1, sbase_a + koff, dst + 1024);
      if (ABL == 3) {
        dma16t_nt(voff_b0, sbase_b + koff, dst + OPER_BYTES);
        dma16t_nt(voff_b1, sbase_b + koff, dst + OPER_BYTES + 1024);
      } else {
        dma16t(voff_b0, sbase_b + koff, dst + OPER_BYTES);
        dma16t(voff_b1, sbase_b + koff, dst + OPER_BYTES + 1024);
      }
    }
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

  advance(i_g, i_c);
  if (i_g >= n_qgroups) return;
  setup_issue_segment(true);

  // ---- compute side ---------------------------------------------------------------------------------------------
  // fragment of a 32x32x16 step: lane l supplies row (l & 31), K bytes [ks * 32 + (l >> 5) * 16, +16) of the slice
  const int l31 = lane & 31, lh = lane >> 5;
  const int a_off0 = (wm * 64 + l31) * ROWB + (((0 + lh) ^ swz4t(l31)) << 4);       // K step 0; tile m adds m * 32 rows
  const int a_off1 = (wm * 64 + l31) * ROWB + (((2 + lh) ^ swz4t(l31)) << 4);       // K step 1
  const int b_off0 = OPER_BYTES + (wn * 128 + l31) * ROWB + (((0 + lh) ^ swz4t(l31)) << 4);
  const int b_off1 = OPER_BYTES + (wn * 128 + l31) * ROWB + (((2 + lh) ^ swz4t(l31)) << 4);
  int c_g = i_g, c_c = i_c, c_mod = 0, c_seg = 0;
  uint32_t c_gs = 0;                               // global index of the slice being computed
  bool more = true;                                // a slice c_gs + 1 exists

  // prologue: slices 0..3 in flight, slice 0 landed, first fragments in registers.  The issue side needs
  // slices_per_seg >= 4 here (no segment end inside the first 3 issues is required; 4th may end a segment).
  issue_slice(); issue_slice(); issue_slice(); issue_slice();
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // K step 0 fragments of the current slice (fa0 / fb0, loaded one slice ahead) and K step 1 fragments (fa1 / fb1,
  // loaded under the step-0 MFMAs of the same slice)
  bf16x8_v fa0[2], fa1[2], fb0[4], fb1[4];
  {
    const char* slot = smem;
#pragma unroll
    for (int m = 0; m < 2; ++m) fa0[m] = *reinterpret_cast<const bf16x8_v*>(slot + a_off0 + m * 32 * ROWB);
#pragma unroll
    for (int n = 0; n < 4; ++n) fb0[n] = *reinterpret_cast<const bf16x8_v*>(slot + b_off0 + n * 32 * ROWB);
  }

  for (;;) {      // one iteration = one (tile, modality) segment
    f32x16_v acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[m][n][v] = 0.f;

    auto mfma_step = [&](bf16x8_v (&fa)[2], bf16x8_v (&fb)[4]) {      // 8 MFMAs: one K step on all 8 tiles
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if (ABL == 2) asm volatile("" ::"v"(fa[m]), "v"(fb[n]));
          else acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
        }
    };
    auto slice_step = [&]() {
      const char* slot = smem + (c_gs & 3) * SLOT_BYTES;
#pragma unroll
      for (int m = 0; m < 2; ++m) fa1[m] = *reinterpret_cast<const bf16x8_v*>(slot + a_off1 + m * 32 * ROWB);
#pragma unroll
      for (int n = 0; n < 4; ++n) fb1[n] = *reinterpret_cast<const bf16x8_v*>(slot + b_off1 + n * 32 * ROWB);
      mfma_step(fa0, fb0);
      {   // slice c_gs + 1 must have landed (mine) before the barrier; later slices may stay in flight
        const int fly = (int)(i_gs - (c_gs + 2));
        if (fly >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (fly == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0), as a builtin: hipcc must KNOW the step-1 reads have
      __builtin_amdgcn_s_barrier();                     // returned, or it waits for the reads issued below first
      ++c_gs;
      auto next_reads = [&]() {                         // K step 0 fragments of the next slice (fa0 / fb0 are free: the
        if (more) {                                     // step-0 MFMAs above have consumed them)
          const char* nslot = smem + (c_gs & 3) * SLOT_BYTES;
#pragma unroll
          for (int m = 0; m < 2; ++m) fa0[m] = *reinterpret_cast<const bf16x8_v*>(nslot + a_off0 + m * 32 * ROWB);
#pragma unroll
          for (int n = 0; n < 4; ++n) fb0[n] = *reinterpret_cast<const bf16x8_v*>(nslot + b_off0 + n * 32 * ROWB);
        }
      };
      // de-phasing of the two waves of a SIMD: see q2c_persist.hip
      if (!PHASED || !grp) {
        next_reads();
        issue_slice();                                  // slice c_gs + 3 -> the slot just released
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      if (PHASED && grp) {
        next_reads();
        issue_slice();
      }
    };

    for (int c_slice = 0; c_slice < slices_per_seg; c_slice += 2) {
      slice_step();
      if (c_slice + 2 >= slices_per_seg) {            // the slice after the next one closes the segment:
        // does a further segment exist?  (compute-side lookahead of the walk, scalar only)
        int ng = c_g, nc = c_c;
        bool has_next = c_mod + 1 < a.n_mod;
        if (!has_next) { advance(ng, nc); has_next = ng < n_qgroups; }
        more = has_next;
      }
      slice_step();
    }
    // ---- end of a (tile, modality) segment: mask_logits + max over the video's 128 clips, inside the wave -----
    {
      const int q0 = ((c_g << qsh) + qt_off) * 256, vid = ((((c_c << 3) + xcd) << csh) + ct_off) * 2 + wn;
      int l31_e = l31, lh_e = lh;                   // opaque copies (see setup_issue_segment)
      asm volatile("" : "+v"(l31_e), "+v"(lh_e));
      const float* mpatch = reinterpret_cast<const float*>(smem + MASK_OFF + (c_seg & 1) * 1024) + wn * 128 + l31_e;
      float* stash = reinterpret_cast<float*>(smem + STASH_OFF) + wn;
      const bool last_mod = c_mod == a.n_mod - 1;
      const bool vid_ok = vid < a.nv;
      float mk[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) mk[n] = vid_ok ? mpatch[n * 32] : 0.f;
      // every clip of this wave's video valid: x * 1 + (1 - 1) * -1e10 == x exactly -> skip the mask_logits arithmetic
      bool all_on = true;
#pragma unroll
      for (int n = 0; n < 4; ++n) all_on = all_on && (mk[n] == 1.f);
      const bool fast = __all(all_on);
      // accumulator element v of tile (m, n) in lane l: row m * 32 + 8 * (v / 4) + 4 * (l >> 5) + (v % 4), column
      // n * 32 + (l & 31).  In-lane maxima over the 4 column tiles: x[m * 16 + v]
      float x[32];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          float mx = -INFINITY;
          if (fast) {
#pragma unroll
            for (int n = 0; n < 4; ++n) mx = fmaxf(mx, acc[m][n][v]);
          } else {
#pragma unroll
            for (int n = 0; n < 4; ++n)
              mx = fmaxf(mx, acc[m][n][v] * mk[n] + (1.f - mk[n]) * -1e10f);   // mask_logits, xml/model_xml.py:640-641
          }
          x[m * 16 + v] = mx;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // reduce-scatter over the 32 lanes that share (l >> 5): lane c = l & 31 ends with the maximum of x[c]
      {
        const bool b16 = (l31_e & 16) != 0, b8 = (l31_e & 8) != 0, b4 = (l31_e & 4) != 0, b2 = (l31_e & 2) != 0,
                   b1 = (l31_e & 1) != 0;
        float t[16], y[8], z[4], u[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {       // lane <-> lane ^ 16 through the LDS crossbar (ds_swizzle, xor mask 0x10)
          const float give = b16 ? x[i] : x[i + 16];
          const float got = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(give), 0x401F));
          t[i] = fmaxf(b16 ? x[i + 16] : x[i], got);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          y[i] = fmaxf(b8 ? t[i + 8] : t[i], dpp_read32<0x140>(b8 ? t[i] : t[i + 8]));       // row_mirror
#pragma unroll
        for (int i = 0; i < 4; ++i)
          z[i] = fmaxf(b4 ? y[i + 4] : y[i], dpp_read32<0x141>(b4 ? y[i] : y[i + 4]));       // row_half_mirror
#pragma unroll
        for (int i = 0; i < 2; ++i)
          u[i] = fmaxf(b2 ? z[i + 2] : z[i], dpp_read32<0x1B>(b2 ? z[i] : z[i + 2]));        // quad_perm [3,2,1,0]
        float red = fmaxf(b1 ? u[1] : u[0], dpp_read32<0xB1>(b1 ? u[0] : u[1]));             // quad_perm [1,0,3,2]
        const int c = l31_e, vv = c & 15;
        const int lrow = wm * 64 + (c >> 4) * 32 + (vv >> 2) * 8 + lh_e * 4 + (vv & 3);
        if (!last_mod) {
          stash[lrow * 2] = red;
        } else {
          if (a.n_mod == 2) red = (stash[lrow * 2] + red) * 0.5f;            // (video + sub) / 2, xml/model_xml.py:574
          if (q0 + lrow < a.nq && vid_ok) a.out[(int64_t)(q0 + lrow) * a.ld_out + vid] = red;
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
static int launch_q2c_persist32(const Q2cPersist32Args& a, hipStream_t st) {
  const int lds = 4 * 2 * 256 * 64 + 2048 + 2048;
  extern int g_q2c_ablation;
  auto kern = g_q2c_ablation == 1 ? q2c_persist32_kernel<T, 1> : g_q2c_ablation == 2 ? q2c_persist32_kernel<T, 2>
             : g_q2c_ablation == 3 ? q2c_persist32_kernel<T, 3> : g_q2c_ablation == 4 ? q2c_persist32_kernel<T, 0, false>
                                                                                     : q2c_persist32_kernel<T, 0, true>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
    return XML_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, st, a);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// Requirements (checked by the caller, which otherwise uses the per-modality kernels): lpad == 128,
// hidden * sizeof(T) a multiple of 128 bytes (an even number of 64-byte slices) and at least 6 slices:
// the mask patch of segment s+2 is fetched 4 slices ahead and must not land before the epilogue of segment s.
int xmli_q2c_scores_persist32(int n_mod, const void* const* qn, const void* const* cn, const float* const* mask,
                            float* out, int64_t ld_out, int nq, int nv, int lpad, int hidden, int dt, hipStream_t st) {
  Q2cPersist32Args a;
  for (int m = 0; m < 2; ++m) {
    a.qn[m] = qn[m < n_mod ? m : 0];
    a.cn[m] = cn[m < n_mod ? m : 0];
    a.mask[m] = mask[m < n_mod ? m : 0];
  }
  if (lpad != 128) return XML_ERR_UNSUPPORTED;
  a.out = out; a.ld_out = ld_out; a.nq = nq; a.nv = nv; a.hidden = hidden; a.n_mod = n_mod;
  a.tq = cdiv(nq, 256); a.tc = cdiv(nv, 2);
  a.qsh = a.tq >= 5 ? 3 : a.tq >= 3 ? 2 : a.tq == 2 ? 1 : 0;     // few queries: more workgroups share a query tile
  if (dt != XML_BF16) return XML_ERR_UNSUPPORTED;
  return launch_q2c_persist32<bf16_t>(a, st);
}
