// K6, persistent fused variant (the one the engine uses): one workgroup per CU walks a static list of
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

struct Q2cPersistArgs {
  const void* qn[2];
  const void* cn[2];
  const float* mask[2];
  const uint32_t* mbits[2];   // BITMASK kernels: (nv, 4) words, bit l of a video = clip l valid
                              // PACKED kernels: (2 * tc, 4) words, bit c of wave tile w = column c of its 128 columns valid
  const int32_t* slot_ids;    // PACKED: (2 * tc, 8) one code per 16-column block of wave tile w (xml_q2c_scores_packed):
                              //         id >= 0 last block of video id, -1 inside a video, -2 unused, -3 continues in the next wave tile
  float* out;
  int64_t ld_out;
  int nq, nv, hidden, n_mod, tq, tc;
  int qsh;     // log2 of the query tiles per XCD super-tile (0..3): the 32 workgroups of an XCD form 2^qsh x 2^(5-qsh)
  int lsh;     // log2 of the consecutive rounds an XCD spends on ADJACENT clip tiles (0: round c of XCD x takes tiles
               // 2^csh (8c + x) ..; 2: four rounds cover 16 adjacent tiles = 32 videos = one 128-byte line of `out` rows)
  int rsh;     // log2 of the rounds per corpus chunk: all query groups visit a chunk before the walk moves on, so that
               // passes 2..n over a chunk's clip tiles are served by the 256 MB Infinity Cache instead of HBM
};

__device__ __forceinline__ void dma16s(uint32_t voff, const char* sbase, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

// two consecutive 1 KiB pieces of one operand
__device__ __forceinline__ void dma_pair(uint32_t v0, uint32_t v1, const char* sb, uint32_t lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3"
      :
      : "v"(v0), "v"(v1), "s"(lds_dst), "s"(sb)
      : "memory", "scc");
}
// four consecutive 1 KiB pieces of one operand
__device__ __forceinline__ void dma_quad(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, const char* sb,
                                         uint32_t lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5"
      :
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(lds_dst), "s"(sb)
      : "memory", "scc");
}

// first K chunk of a segment: C = 0 as an inline constant (no 128 v_mov per segment to clear the accumulators)
template <typename T> struct MmaInit;
template <> struct MmaInit<float> {
  __device__ static __forceinline__ void chunk(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};
template <> struct MmaInit<bf16_t> {
  __device__ static __forceinline__ void chunk(f32x4& acc, const uint4& a, const uint4& b) {
    union { uint4 u; bf16x8_v v; } ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  }
};

template <> struct MmaInit<f16_t> {
  __device__ static __forceinline__ void chunk(f32x4& acc, const uint4& a, const uint4& b) {
    union { uint4 u; f16x8_v v; } ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ua.v, ub.v, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  }
};
// XML_F16 operands are the hi planes of xml_split_f16_rows(fixed_log2 = XML_F16_UNIT_LOG2): unit-norm rows scaled by
// 2^14 on both sides, so a score leaves the accumulators scaled by 2^28 (the exact-rank FILTER, inference.stage_exact_topk)
static constexpr float K6_F16_OUT_SCALE = 1.f / (float)(1u << (2 * XML_F16_UNIT_LOG2));

// MFMA issue order inside a 4 x 4 block.  Boustrophedon (row m walks n upwards, row m + 1 downwards: exactly ONE operand
// changes between consecutive MFMAs, 15 operand switches per 16 instead of 18) measured on one box, same process pair:
// f16 operands 68.2 vs 69.3 ms (+1.5 %), bf16 operands 65.1 vs 64.8 ms (-0.5 %) -- kept for f16 only.  Same accumulators.
template <typename T> struct K6Order { static constexpr bool snake = false; };
template <> struct K6Order<f16_t> { static constexpr bool snake = true; };

// ABL 8 (timing probe): per wave of workgroup 0, shader-clock cycles spent between "about to wait" and "barrier
// released" summed over all slices, and the wave's total; read back with xml_debug_read_k6_probe
#ifdef XML_DEBUG_VARIANTS
__device__ unsigned long long g_k6_probe[32];
extern "C" int xml_debug_read_k6_probe(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_k6_probe), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -4;
}
#endif

// value of `v` in the lane selected by a DPP control word (row_mirror 0x140, row_half_mirror 0x141, quad_perm 0x00-0xff)
template <int CTRL>
__device__ __forceinline__ float dpp_read(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// lane id from the hardware (v_mbcnt), as a VOLATILE asm: hipcc can neither hoist it out of the segment loop nor keep a
// copy of threadIdx live across the K loop (the persistent kernels sit at 246-256 VGPRs; a spilled lane id comes back
// through a scratch load whose s_waitcnt vmcnt(0) would drain the hand-counted DMA stream once per segment)
__device__ __forceinline__ int lane_id_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

__device__ __forceinline__ int swz4p(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

// Specialised to lpad == 128 (one video per 128-column group -- the TVR shape): no per-column divisions, one
// reduction per accumulator row.  Other clip paddings use the per-modality kernels (q2c_ring.hip).
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
// PACKED (ragged corpora): videos are padded to a multiple of 16 clips and laid back to back into the 256 columns of a
// tile (xml_q2c_scores_packed), so padding rows cost (almost) no MFMA work.  The MFMA operands are SWAPPED (clips as the A
// operand): a lane then holds ONE query column and 4 clip rows of every 16 x 16 block, so the maximum over a video's
// clips is an in-lane v_max3 chain over its blocks plus two permlane swaps across the four row groups (6 VALU per
// video instead of the 45-op 16-lane reduce-scatter of the plain layout).  A video may straddle the two wave tiles of
// a tile: the left wave parks its partial maxima in a 1 KiB LDS patch, one extra s_barrier, the right wave combines.
// Scores land in the videos' ORIGINAL columns of `out` and are bitwise those of the unpacked layout.
template <typename T, int ABL = 0, bool PHASED = true, bool TILED = false, bool NOMASK = false, bool BITMASK = false,
          bool PACKED = false>   // ABL (debug library only): 1 no DMA after the prologue, 2 no MFMA, 8 timing probe
__global__ __launch_bounds__(512, 2) void q2c_persist_kernel(Q2cPersistArgs a) {
  constexpr int ROWB = 64;
  constexpr int OPER_BYTES = 256 * ROWB;
  constexpr int SLOT_BYTES = 2 * OPER_BYTES;
  // Uneven DMA duty: the timing probe shows the first wave group parked at the barrier ~35 % of the time while the
  // second is on the critical path, so the first group issues 6 of the 8 pieces per SIMD pair -- its wave w loads A
  // pieces 4w..4w+3 and B pieces 2w, 2w+1; wave 4+w loads B pieces 8+2w, 9+2w (+1 % measured against the even 4 / 4
  // split; all 8 on the first group gives the gain back).  (Measured and not kept, profiles/r01-r02_k6_notes.md: one
  // address register + `offset:` immediates for a wave's pieces, wider s_nop spacing between them, DMA pieces issued
  // between the MFMAs, non-temporal clip loads.  Their code lives in the history, not here.)
  // NOMASK (the caller vouches that every clip mask is 1: full-length videos, e.g. the TVR benchmark shape): no mask
  // patches are needed, the 2 KiB they occupy are what a FIFTH ring slot was missing (5 x 32 KiB = all 160 KiB of LDS):
  // three slices in flight behind the awaited one instead of two, +1.1-1.3 % measured.
  // BITMASK: ragged corpora get the fifth slot too -- binary clip masks packed 128 bits per video arrive through SCALAR
  // loads (lgkmcnt, invisible to the hand-counted vmcnt of the DMA stream), no mask DMA, no LDS patches.
  constexpr bool FIVE = NOMASK || BITMASK;       // (PACKED: four slots, the patch area carries the straddle hand-off)
  constexpr int NSLOT = FIVE ? 5 : 4;
  constexpr int RING_BYTES = NSLOT * SLOT_BYTES;
  constexpr int MASK_OFF = RING_BYTES;            // 2 x 1 KiB mask patches (256 columns x f32)
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
  // operand addressing.  Row-major (rows of k_bytes): a DMA piece is 16 rows x 64 B of the K slice, every 128-byte line
  // is requested by two consecutive slices.  TILED (xml_q2c_tile_rows): a 256-row operand tile is stored slice-major,
  // [slice][row][64 B] = the LDS image of each slice: a piece is 1 KiB contiguous and every line is requested once.
  // Tile bases are the same byte offsets in both layouts (256 * k_bytes per tile).
  const int row_stride = TILED ? ROWB : k_bytes;
  constexpr int SLICE_STRIDE = TILED ? 256 * ROWB : ROWB;
  const int n_qgroups = (a.tq + (1 << qsh) - 1) >> qsh;
  const int lsh = a.lsh, lmask = (1 << a.lsh) - 1;
  const int cr = ((a.tc + (8 << (csh + lsh)) - 1) / (8 << (csh + lsh))) << lsh;    // rounds per query group on one XCD

  // tile of (query group g, round c):  qt = 2^qsh g + qt_off,  ct = 2^(csh+lsh) (8 (c >> lsh) + xcd) + 2^csh (c & lmask) + ct_off
  auto clip_tile = [&](int c) -> int {
    return ((((((c >> lsh) << 3) + xcd) << lsh) + (c & lmask)) << csh) + ct_off;
  };
  auto tile_valid = [&](int g, int c) -> bool {
    return ((g << qsh) + qt_off) < a.tq && clip_tile(c) < a.tc;
  };
  // Walk order: the corpus is visited in chunks of 2^rsh rounds (one round = the 8 XCDs x 2^csh clip tiles the chip
  // works on at a time); EVERY query group visits a chunk before the walk moves to the next chunk.  A chunk is sized to
  // stay in the Infinity Cache, so only the first query group streams it from HBM (the straight order -- all chunks
  // per query group -- read the whole corpus from HBM once per query group).
  const int rsh = a.rsh, rmask = (1 << a.rsh) - 1;
  auto advance = [&](int& g, int& c) {            // next valid (g, c) in walk order, g == n_qgroups when exhausted
    do {
      const int c1 = c + 1;
      if (c >= 0 && ((c1 & rmask) == 0 || c1 == cr)) {     // c was the last round of its chunk
        if (g + 1 < n_qgroups) { ++g; c = (c >> rsh) << rsh; }
        else { g = 0; c = c1; }
      } else {
        c = c1;
      }
      if (c >= cr) { g = n_qgroups; break; }
    } while (!tile_valid(g, c));
  };

  // ---- issue side (DMA stream, runs up to 3 slices ahead of the slice being computed) ---------------------
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // this wave's byte offsets inside a slot's operand images, pinned in SGPRs (opaque to the compiler: rematerialising the
  // shift in front of every DMA group is one more scalar instruction per slice that the partner wave's MFMAs do not cover)
  uint32_t wave_2k = (uint32_t)wave * 2048u, wave_4k = (uint32_t)wave * 4096u;
  asm volatile("" : "+s"(wave_2k), "+s"(wave_4k));
  int i_g = 0, i_c = -1, i_mod = 0, i_slice = 0, i_seg = 0;
  uint32_t i_gs = 0;                               // slices issued so far (global) -> ring slot
  uint32_t voff_a0 = 0, voff_a1 = 0, voff_b0 = 0, voff_b1 = 0;
  uint32_t voff_x[2] = {};                         // A pieces 2, 3 of the first group's waves
  const char* sbase_a = nullptr;
  const char* sbase_b = nullptr;

  auto setup_issue_segment = [&](bool new_tile) {
    const int q0 = ((i_g << qsh) + qt_off) * 256, v0 = clip_tile(i_c) * 2;
    const int lane_o = lane_id_now();               // re-derived here: nothing lane-dependent stays live across the MFMA loop
    if (new_tile) {
      const int rsub = lane_o >> 2, pslot = lane_o & 3;
      {
        int q_left = a.nq - q0;                     // (a scalar here and now: under SGPR pressure hipcc otherwise parks a
        asm volatile("" : "+s"(q_left));            // VGPR copy of nq across the K loop)
        auto off_a = [&](int piece) -> uint32_t {
          const int row = piece * 16 + rsub;
          const int qrow = (row < q_left) ? row : 0;
          return (uint32_t)qrow * row_stride + (pslot ^ swz4p(row)) * 16;
        };
        auto off_b = [&](int piece) -> uint32_t {
          const int row = piece * 16 + rsub;
          // (PACKED: nv = 2 * tc wave tiles, every tile of the image is complete)
          const int brow = (PACKED || v0 + (row >> 7) < a.nv) ? row : (row & 127);
          return (uint32_t)brow * row_stride + (pslot ^ swz4p(row)) * 16;
        };
        if (grp == 0) {
          voff_a0 = off_a(wave * 4); voff_a1 = off_a(wave * 4 + 1); voff_x[0] = off_a(wave * 4 + 2); voff_x[1] = off_a(wave * 4 + 3);
          voff_b0 = off_b(wave * 2); voff_b1 = off_b(wave * 2 + 1);
        } else {
          voff_b0 = off_b(8 + (wave - 4) * 2); voff_b1 = off_b(9 + (wave - 4) * 2);
        }
      }
    }
    sbase_a = reinterpret_cast<const char*>(a.qn[i_mod]) + (int64_t)q0 * k_bytes;
    sbase_b = reinterpret_cast<const char*>(a.cn[i_mod]) + (int64_t)v0 * 128 * k_bytes;
    if (!FIVE && !PACKED && wave == 0) {   // mask patch of the segment: lane l carries columns 4 l .. 4 l + 3 of the tile's 256 columns
      const int mrow = (v0 + (lane_o >> 5) < a.nv) ? lane_o : (lane_o & 31);
      const char* sbase_m = reinterpret_cast<const char*>(a.mask[i_mod]) + (int64_t)v0 * 128 * 4;
      dma16s((uint32_t)mrow * 16, sbase_m, lds0 + MASK_OFF + (i_seg & 1) * 1024);
    }
  };
  // 4 DMA instructions: this wave's 32 rows of A and of B.  The stream SATURATES: once the walk is exhausted it keeps
  // re-fetching slice 0 of the last segment into the slots it would have used (valid addresses, dead data), so that
  // the consumer side needs no end-of-stream cases: always 2 slices in flight behind the awaited one (vmcnt(8)),
  // fragment reads unconditional.  ~8 fewer scalar branches per slice; the kernel drains the stream before it exits.
  int i_slot = 0;
  int i_left = slices_per_seg, i_inc = 1;              // slices left to issue in the current segment; slice increment
  auto issue_advance = [&]() {
    ++i_gs;
    if (++i_slot == NSLOT) i_slot = 0;
    i_slice += i_inc;
    if (--i_left == 0) {                                 // next segment: other modality of the tile, or the next tile
      i_slice = 0;
      i_left = slices_per_seg;
      ++i_seg;
      bool new_tile = false;
      if (++i_mod == a.n_mod) {
        i_mod = 0;
        advance(i_g, i_c);
        new_tile = true;
      }
      if (i_g < n_qgroups) setup_issue_segment(new_tile);
      else { i_left = 0x7fffffff; i_inc = 0; }           // exhausted: slice 0 of the last segment from now on
    }
  };
  auto issue_slice = [&]() {
    const int koff = i_slice * SLICE_STRIDE;
    if (ABL != 1 || i_gs < 4) {
      const uint32_t slot0 = lds0 + i_slot * SLOT_BYTES;
      if (grp == 0) {
        dma_quad(voff_a0, voff_a1, voff_x[0], voff_x[1], sbase_a + koff, slot0 + wave_4k);
        dma_pair(voff_b0, voff_b1, sbase_b + koff, slot0 + OPER_BYTES + wave_2k);
      } else {
        dma_pair(voff_b0, voff_b1, sbase_b + koff, slot0 + OPER_BYTES + wave_2k);      // (8192 + (wave - 4) * 2048)
      }
    }
    issue_advance();
  };

  advance(i_g, i_c);
  if (i_g >= n_qgroups) return;
  setup_issue_segment(true);

  // ---- compute side ---------------------------------------------------------------------------------------------
  const int a_off = (wm * 64 + fr) * ROWB + ((fg ^ swz4p(fr)) << 4);
  const int b_off = OPER_BYTES + (wn * 128 + fr) * ROWB + ((fg ^ swz4p(fr)) << 4);
  int c_g = i_g, c_c = i_c, c_mod = 0, c_seg = 0;
  int c_slot = 0;                                  // its ring slot

  // prologue: slices 0..3 in flight, slice 0 landed, first fragments in registers.  The issue side needs
  // slices_per_seg >= 4 here (no segment end inside the first 3 issues is required; 4th may end a segment).
  issue_slice(); issue_slice(); issue_slice(); issue_slice();
  if (NSLOT == 5) {
    issue_slice();                                       // five slices in flight, the first one awaited
    if (grp == 0) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    if (grp == 0) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  uint4 faA[4], faB[4], fbL[4], fbH[4];
  {
    const char* slot = smem;
#pragma unroll
    for (int m = 0; m < 4; ++m) faA[m] = *reinterpret_cast<const uint4*>(slot + a_off + m * 16 * ROWB);
#pragma unroll
    for (int n = 0; n < 4; ++n) fbL[n] = *reinterpret_cast<const uint4*>(slot + b_off + n * 16 * ROWB);
  }

  // The whole segment loop exists twice, once per wave group: the de-phased order of [preamble, MFMA] is then
  // straight-line code instead of two wave-uniform branches per slice.
  auto run = [&](auto grp_tag) {
  constexpr bool GRP1 = decltype(grp_tag)::value;
  float stash = 0.f;                   // modality-0 maximum of this lane's row of the current tile
  unsigned long long probe_wait = 0, probe_bar = 0, probe_t0 = 0;
  unsigned long long probe_r0 = 0;
  if (ABL == 8) { probe_t0 = __builtin_amdgcn_s_memtime(); probe_r0 = __builtin_amdgcn_s_memrealtime(); }
  for (;;) {      // one iteration = one (tile, modality) segment
    f32x4 acc[4][8];       // written by the first slice of the segment (MmaInit: C = 0)

    auto slice_step = [&](uint4 (&fc)[4], uint4 (&fn)[4], auto init_tag) {
      constexpr bool INIT = decltype(init_tag)::value;
      const char* slot = smem + c_slot * SLOT_BYTES;
#pragma unroll
      for (int n = 0; n < 4; ++n) fbH[n] = *reinterpret_cast<const uint4*>(slot + b_off + (n + 4) * 16 * ROWB);
      // (issue order inside the block: K6Order)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          const int n = (K6Order<T>::snake && (m & 1)) ? 3 - nn : nn;
          if (ABL == 2) asm volatile("" ::"v"(fc[m].x), "v"(fbL[n].x), "v"(fc[m].w), "v"(fbL[n].w));
          else if constexpr (INIT) MmaInit<T>::chunk(acc[m][n], fbL[n], fc[m]);      // clips as the A operand: see the epilogue
          else Mma<T>::chunk(acc[m][n], fbL[n], fc[m]);
        }
      // ONE wait: vmcnt(8) -- my DMAs of slice c_gs + 1 have landed, the two younger slices stay in flight -- and
      // lgkmcnt(0) -- all my LDS reads of slice c_gs have returned (as a builtin: hipcc must KNOW the fbH reads are
      // complete, or it waits for the reads issued below before h1)
      unsigned long long t_a = 0;
      if (ABL == 8) t_a = __builtin_amdgcn_s_memtime();
      if (NSLOT == 5 && !GRP1) __builtin_amdgcn_s_waitcnt(0x4072);        // vmcnt(18): 3 younger slices x 6 pieces
      else if (NSLOT == 5) __builtin_amdgcn_s_waitcnt(0x0076);            // vmcnt(6): 3 x 2
      else if (!GRP1) __builtin_amdgcn_s_waitcnt(0x007c);                 // 6 pieces per slice: vmcnt(12)
      else __builtin_amdgcn_s_waitcnt(0x0074);                            // 2 pieces per slice: vmcnt(4)
      if (ABL == 8) { probe_wait += __builtin_amdgcn_s_memtime() - t_a; t_a = __builtin_amdgcn_s_memtime(); }
      __builtin_amdgcn_s_barrier();
      if (ABL == 8) probe_bar += __builtin_amdgcn_s_memtime() - t_a;
      if (++c_slot == NSLOT) c_slot = 0;
      auto next_reads = [&]() {                         // their latency hides under the DMA issue + MFMAs
        const char* nslot = smem + c_slot * SLOT_BYTES;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          fn[m] = *reinterpret_cast<const uint4*>(nslot + a_off + m * 16 * ROWB);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) fbL[n] = *reinterpret_cast<const uint4*>(nslot + b_off + n * 16 * ROWB);
      };
      auto h1 = [&]() {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int nn = 0; nn < 4; ++nn) {
            const int n = (K6Order<T>::snake && (m & 1)) ? 3 - nn : nn;
            if (ABL == 2) asm volatile("" ::"v"(fc[m].x), "v"(fbH[n].x), "v"(fc[m].w), "v"(fbH[n].w));
            else if constexpr (INIT) MmaInit<T>::chunk(acc[m][n + 4], fbH[n], fc[m]);
            else Mma<T>::chunk(acc[m][n + 4], fbH[n], fc[m]);
          }
      };
      // The two waves of a SIMD (w and w + 4) leave the barrier together.  If both ran [reads, DMA issue, MFMA]
      // the matrix pipe would idle through both preambles (~250 cycles per slice); so the second group runs its
      // 16 MFMAs FIRST (their operands were complete before the barrier) and its preamble under the first
      // group's MFMAs.
      // (The MFMA block itself stays outside any branch: accumulators updated in both arms of a branch get phi
      // copies -- 350 spilled VGPRs when tried.)
      // (Reading the next fragments before h1 in BOTH groups was tried after the timing probe: no gain, and the second
      // group's variant then spills.)
      if (!GRP1) {
        next_reads();
        issue_slice();                                  // slice c_gs + 3 -> the slot just released
      }
      __builtin_amdgcn_sched_barrier(0);
      h1();
      __builtin_amdgcn_sched_barrier(0);
      if (GRP1) {
        next_reads();
        issue_slice();
      }
    };

    slice_step(faA, faB, std::true_type{});               // slices_per_seg is even and >= 6
    slice_step(faB, faA, std::false_type{});
    for (int c_slice = 2; c_slice < slices_per_seg; c_slice += 2) {
      slice_step(faA, faB, std::false_type{});
      slice_step(faB, faA, std::false_type{});
    }
    // ---- end of a (tile, modality) segment: mask_logits + max over the video's 128 clips, inside the wave -----
    if constexpr (PACKED) {
      // acc[m][n][r] = score of query (wm * 64 + m * 16 + fr) against packed clip column (n * 16 + fg * 4 + r) of this wave tile
      const int q0 = ((c_g << qsh) + qt_off) * 256, ct = clip_tile(c_c);
      const int lane_e = lane_id_now(), fr_e = lane_e & 15, fg_e = lane_e >> 4;      // (see setup_issue_segment)
      const bool last_mod = c_mod == a.n_mod - 1;
      const int wt_s = __builtin_amdgcn_readfirstlane(ct * 2 + wn);
      typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
      typedef int32_t i32x8_t __attribute__((ext_vector_type(8)));
      u32x4_t wv;
      i32x8_t ids;
      int tail;       // code of block 7 of the tile's LEFT wave tile, read by all eight waves: -3 <=> a video straddles the
                      // wave tiles <=> every wave takes the extra barrier (uniform by construction, whatever the table holds)
      const uint32_t* mb = (c_mod == 0 ? a.mbits[0] : a.mbits[1]) + (int64_t)wt_s * 4;
      const int32_t* idp = a.slot_ids + (int64_t)wt_s * 8;
      const int32_t* tp = a.slot_ids + (int64_t)(wt_s - wn) * 8 + 7;
      asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx8 %1, %4, 0x0\n\ts_load_dword %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(wv), "=&s"(ids), "=&s"(tail) : "s"(mb), "s"(idp), "s"(tp) : "memory");
      const bool straddle = tail == -3;
      bool head = straddle && wn == 1;          // the first video that ends in the right wave tile began in the left one
      float headred = -INFINITY;
      int head_id = -1;
      // lane (fr, fg) ends up with the row of query block m = {0, 2, 1, 3}[fg] (see the swaps below)
      const int lrow = wm * 64 + ((((fg_e & 1) << 1) | (fg_e >> 1)) << 4) + fr_e;
      int q_left = a.nq - q0;
      asm volatile("" : "+s"(q_left));
      const bool row_ok = lrow < q_left;
      float* orow = a.out + (int64_t)(q0 + lrow) * a.ld_out;
      float* patch = reinterpret_cast<float*>(smem + MASK_OFF) + wm * 64 + lane_e;
      // modality-0 maxima wait for modality 1 in LDS (the four-slot ring leaves 32 KiB): slot n = the video that ends in
      // block n of this wave tile, slot 8 = the video that straddles into it.  No register lives across the K loop.
      float* stash = reinterpret_cast<float*>(smem + MASK_OFF + 1024) + wave * (9 * 64) + lane_e;
      const int sh = fg_e * 4;
      auto finish = [&](float red, int slot, int id) {
        if (!last_mod) {
          stash[slot * 64] = red;
        } else {
          if (a.n_mod == 2) red = (stash[slot * 64] + red) * 0.5f;                        // (video + sub) / 2, xml/model_xml.py:574
          if constexpr (std::is_same<T, f16_t>::value) red *= K6_F16_OUT_SCALE;
          if (row_ok) orow[id] = red;
        }
      };
      float run[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const int id = n == 0 ? ids.s0 : n == 1 ? ids.s1 : n == 2 ? ids.s2 : n == 3 ? ids.s3 : n == 4 ? ids.s4
                       : n == 5 ? ids.s5 : n == 6 ? ids.s6 : ids.s7;
        if (id == -2) continue;                                              // unused block (wave-uniform)
        const uint32_t w32 = (n >> 1) == 0 ? wv.x : (n >> 1) == 1 ? wv.y : (n >> 1) == 2 ? wv.z : wv.w;
        const uint32_t w16 = (n & 1) ? (w32 >> 16) : (w32 & 0xffffu);
        if (w16 == 0xffffu) {                                                // all 16 clips of the block valid
#pragma unroll
          for (int m = 0; m < 4; ++m)
            run[m] = fmaxf(fmaxf(fmaxf(fmaxf(run[m], acc[m][n][0]), acc[m][n][1]), acc[m][n][2]), acc[m][n][3]);
        } else {                                                             // mask_logits, xml/model_xml.py:640-641:
          const uint32_t lw = w16 >> sh;                                     // x * 1 + 0 * -1e10 == x, x * 0 + 1 * -1e10 == -1e10
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool on = (lw >> r) & 1u;
#pragma unroll
            for (int m = 0; m < 4; ++m) run[m] = fmaxf(run[m], on ? acc[m][n][r] : -1e10f);
          }
        }
        if (id >= 0 || (n == 7 && id == -3)) {                               // a video ends here (or leaves this wave tile)
          // maximum over the four row groups fg: lanes l, l ^ 32 through permlane32_swap, then l ^ 16 through
          // permlane16_swap; each swap also halves the number of live values, so 4 query blocks cost 3 swaps + 3 max
          const auto s01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(run[0]), __float_as_uint(run[1]), false, false);
          const auto s23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(run[2]), __float_as_uint(run[3]), false, false);
          const float p = fmaxf(__uint_as_float(s01[0]), __uint_as_float(s01[1]));
          const float q = fmaxf(__uint_as_float(s23[0]), __uint_as_float(s23[1]));
          const auto spq = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(q), false, false);
          const float red = fmaxf(__uint_as_float(spq[0]), __uint_as_float(spq[1]));
          run[0] = run[1] = run[2] = run[3] = -INFINITY;
          if (n == 7 && id == -3) *patch = red;
          else if (head) { headred = red; head_id = id; head = false; }
          else finish(red, n, id);
        }
      }
      if (straddle) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (wn == 1 && head_id >= 0) finish(fmaxf(headred, *patch), 8, head_id);
      }
    } else {
      // One video per wave tile.  acc[m][n][r] = score of query (wm * 64 + m * 16 + fr) against clip (n * 16 + fg * 4 + r):
      // the maximum over the clips is an in-lane v_max3 chain (64 ops) and two permlane swaps across the four row groups.
      const int q0 = ((c_g << qsh) + qt_off) * 256, vid = clip_tile(c_c) * 2 + wn;
      const int lane_e = lane_id_now(), fr_e = lane_e & 15, fg_e = lane_e >> 4;      // (see setup_issue_segment)
      const bool last_mod = c_mod == a.n_mod - 1;
      const bool vid_ok = vid < a.nv;
      float run[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      auto plain_block = [&](int n) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          run[m] = fmaxf(fmaxf(fmaxf(fmaxf(run[m], acc[m][n][0]), acc[m][n][1]), acc[m][n][2]), acc[m][n][3]);
      };
      if constexpr (BITMASK) {
        // the video's 128 mask bits: four scalar loads (the address is wave-uniform)
        // (readfirstlane: hipcc must KNOW the index is uniform, or it emits VMEM loads whose s_waitcnt vmcnt(0) would drain
        // the DMA stream at every tile)
        const int vid_s = __builtin_amdgcn_readfirstlane(vid_ok ? vid : 0);
        const uint32_t* mb = (c_mod == 0 ? a.mbits[0] : a.mbits[1]) + (int64_t)vid_s * 4;
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t wv;
        asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wv) : "s"(mb) : "memory");
        const int sh = fg_e * 4;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          const uint32_t w32 = (n >> 1) == 0 ? wv.x : (n >> 1) == 1 ? wv.y : (n >> 1) == 2 ? wv.z : wv.w;
          const uint32_t w16 = (n & 1) ? (w32 >> 16) : (w32 & 0xffffu);
          if (w16 == 0xffffu) {                     // all 16 clips of the block valid (wave-uniform)
            plain_block(n);
          } else {                                  // mask_logits (xml/model_xml.py:640-641) for binary masks:
            const uint32_t lw = w16 >> sh;          // x * 1 + 0 * -1e10 == x, x * 0 + 1 * -1e10 == -1e10
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool on = (lw >> r) & 1u;
#pragma unroll
              for (int m = 0; m < 4; ++m) run[m] = fmaxf(run[m], on ? acc[m][n][r] : -1e10f);
            }
          }
        }
      } else if constexpr (NOMASK) {
#pragma unroll
        for (int n = 0; n < 8; ++n) plain_block(n);
      } else {
        // f32 masks (any values) through the LDS patch of the segment: lane (fr, fg) needs columns n * 16 + fg * 4 + 0..3
        const float4* mpatch = reinterpret_cast<const float4*>(smem + MASK_OFF + (c_seg & 1) * 1024) + wn * 32 + fg_e;
        float4 mk[8];
        bool all_on = true;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          mk[n] = mpatch[n * 4];
          all_on = all_on && mk[n].x == 1.f && mk[n].y == 1.f && mk[n].z == 1.f && mk[n].w == 1.f;
        }
        // every clip of this wave's video valid (the common case): x * 1 + (1 - 1) * -1e10 == x exactly, so the
        // multiply-adds of mask_logits are skipped (wave-uniform branch)
        if (__all(all_on)) {
#pragma unroll
          for (int n = 0; n < 8; ++n) plain_block(n);
        } else {
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            const float mv[4] = {mk[n].x, mk[n].y, mk[n].z, mk[n].w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int m = 0; m < 4; ++m)
                run[m] = fmaxf(run[m], acc[m][n][r] * mv[r] + (1.f - mv[r]) * -1e10f);   // mask_logits, xml/model_xml.py:640-641
          }
        }
      }
      // maximum over the four row groups fg (lanes l, l ^ 32, then l ^ 16); each swap also halves the number of live
      // values: lane (fr, fg) ends up with the row of query block {0, 2, 1, 3}[fg]
      {
        const auto s01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(run[0]), __float_as_uint(run[1]), false, false);
        const auto s23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(run[2]), __float_as_uint(run[3]), false, false);
        const float p = fmaxf(__uint_as_float(s01[0]), __uint_as_float(s01[1]));
        const float q = fmaxf(__uint_as_float(s23[0]), __uint_as_float(s23[1]));
        const auto spq = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(q), false, false);
        float red = fmaxf(__uint_as_float(spq[0]), __uint_as_float(spq[1]));
        const int lrow = wm * 64 + ((((fg_e & 1) << 1) | (fg_e >> 1)) << 4) + fr_e;
        if (!last_mod) {
          stash = red;                       // the lane <-> row mapping is the same for both modalities of a tile
        } else {
          if (a.n_mod == 2) red = (stash + red) * 0.5f;                      // (video + sub) / 2, xml/model_xml.py:574
          if constexpr (std::is_same<T, f16_t>::value) red *= K6_F16_OUT_SCALE;
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
#ifdef XML_DEBUG_VARIANTS
  if (ABL == 8 && blockIdx.x == 0 && lane == 0) {
    g_k6_probe[wave * 4 + 0] = probe_wait;
    g_k6_probe[wave * 4 + 1] = probe_bar;
    g_k6_probe[wave * 4 + 2] = __builtin_amdgcn_s_memtime() - probe_t0;
    g_k6_probe[wave * 4 + 3] = __builtin_amdgcn_s_memrealtime() - probe_r0;      // 100 MHz reference: shader clock = [2] / [3] x 100 MHz
  }
#endif
  };
  if (PHASED && grp) run(std::true_type{});
  else run(std::false_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the saturated stream still has DMAs in flight: nothing may
  __builtin_amdgcn_s_barrier();                         // land in this LDS allocation after the workgroup is gone
}

template <typename T>
static int launch_q2c_persist(const Q2cPersistArgs& a, hipStream_t st, bool tiled, int mask_mode) {
  const bool five = tiled && (mask_mode == 1 || mask_mode == 2);     // mask_mode 3: PACKED (four slots + the straddle patch)
  // ring + two mask patches; PACKED: + the straddle patch (1 KiB) and the modality-0 stash (8 waves x 9 x 256 B)
  const int lds = five ? 5 * 2 * 256 * 64 : mask_mode == 3 ? 4 * 2 * 256 * 64 + 1024 + 8 * 9 * 256 : 4 * 2 * 256 * 64 + 2048;
  void (*kern)(Q2cPersistArgs) = nullptr;
  bool ok = false;
#define XML_K6_PICK(...) do { kern = q2c_persist_kernel<__VA_ARGS__>; ok = xml_lds_attr_once<q2c_persist_kernel<__VA_ARGS__>>(lds); } while (0)
  if (tiled && mask_mode == 3) XML_K6_PICK(T, 0, true, true, false, false, true);
  else if (tiled && mask_mode == 1) XML_K6_PICK(T, 0, true, true, true);
  else if (tiled && mask_mode == 2) XML_K6_PICK(T, 0, true, true, false, true);
  else if (tiled) XML_K6_PICK(T, 0, true, true);
#ifdef XML_DEBUG_VARIANTS
  else if (g_q2c_ablation == 1) XML_K6_PICK(T, 1);
  else if (g_q2c_ablation == 2) XML_K6_PICK(T, 2);
  else if (g_q2c_ablation == 8) XML_K6_PICK(T, 8, true);
  else if (g_q2c_ablation == 4) XML_K6_PICK(T, 0, false);
#endif
  else XML_K6_PICK(T, 0, true);
#undef XML_K6_PICK
  if (!ok) return XML_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, st, a);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// Requirements (checked by the caller, which otherwise uses the per-modality kernels): lpad == 128,
// hidden * sizeof(T) a multiple of 128 bytes (an even number of 64-byte slices) and at least 6 slices:
// the mask patch of segment s+2 is fetched 4 slices ahead and must not land before the epilogue of segment s.
int xmli_q2c_scores_persist(int n_mod, const void* const* qn, const void* const* cn, const float* const* mask,
                            float* out, int64_t ld_out, int nq, int nv, int lpad, int hidden, int dt, hipStream_t st,
                            bool tiled, int mask_mode, const uint32_t* const* mbits, const int32_t* slot_ids) {
  Q2cPersistArgs a;
  a.slot_ids = slot_ids;
  for (int m = 0; m < 2; ++m) {
    a.qn[m] = qn[m < n_mod ? m : 0];
    a.cn[m] = cn[m < n_mod ? m : 0];
    a.mask[m] = mask[m < n_mod ? m : 0];
    a.mbits[m] = mbits ? mbits[m < n_mod ? m : 0] : nullptr;
  }
  if ((mask_mode == 2 || mask_mode == 3) && (!mbits || !a.mbits[0] || !a.mbits[1])) return XML_ERR_BAD_ARG;
  if (mask_mode == 3 && (!slot_ids || !tiled || (nv & 1))) return XML_ERR_BAD_ARG;
  if (lpad != 128) return XML_ERR_UNSUPPORTED;
  a.out = out; a.ld_out = ld_out; a.nq = nq; a.nv = nv; a.hidden = hidden; a.n_mod = n_mod;
  a.tq = cdiv(nq, 256); a.tc = cdiv(nv, 2);
  a.qsh = a.tq >= 5 ? 3 : a.tq >= 3 ? 2 : a.tq == 2 ? 1 : 0;     // few queries: more workgroups share a query tile
  if (a.qsh == 3) {
    // The walk hands every workgroup ceil(tq / 2^qsh) x ceil(tc / (8 x 2^(5 - qsh))) tiles, valid or not: with 43 query
    // tiles (TVR val, 10 895 queries) the 8 x 4 super-tile walks 48 x 576 slots for 43 x 573 tiles, the 4 x 8 one 44 x 576
    // (-8 % of the kernel's time).  Take the 4 x 8 shape when it saves more than 3 % of the slots -- it fetches every clip
    // tile for 4 instead of 8 query tiles per XCD, which is what the 8 x 4 shape is there to avoid when the two tie.
    auto slots = [&](int qsh) -> int64_t {
      const int64_t qg = (a.tq + (1 << qsh) - 1) >> qsh, per = 8ll << (5 - qsh);
      return (qg << qsh) * ((a.tc + per - 1) / per * per);
    };
    if (slots(2) * 100 < slots(3) * 97) a.qsh = 2;
  }
  if (g_q2c_qsh >= 0 && g_q2c_qsh <= 4) a.qsh = g_q2c_qsh;       // (debug build: tools/k6_l2_ab.py)
  // Walk order: rsh = 20 is the straight order (every query group walks the whole corpus).  The chunked order
  // (2^rsh rounds per Infinity-Cache-sized chunk, all query groups per chunk; debug knob xml_debug_set_q2c_chunk) was
  // measured in round 2: 4-round chunks cut the HBM passes over the corpus from one per query group to one per launch
  // and were +0.6 % faster, but the fabric-side FETCH_SIZE -- which counts Infinity-Cache hits -- rose from 129 to 209 GB
  // per launch (the query tiles are re-fetched at every chunk switch): not kept (profiles/r02_k6_notes.md).
  a.rsh = g_q2c_chunk_log2 >= 0 ? g_q2c_chunk_log2 : 20;
  a.lsh = g_q2c_line_log2;
  if (dt == XML_BF16) return launch_q2c_persist<bf16_t>(a, st, tiled, mask_mode);
  if (dt == XML_F16) return tiled ? launch_q2c_persist<f16_t>(a, st, tiled, mask_mode) : XML_ERR_UNSUPPORTED;
  return launch_q2c_persist<float>(a, st, tiled, mask_mode);
}


// ---- slice-major operand tiles ------------------------------------------------------------------------------------
// (rows, k_bytes) row-major  ->  [tile = row / 256][slice = byte / 64][row % 256][64 B]; rows beyond `rows` are zero.
// One 16-byte chunk per thread; consecutive threads write consecutive chunks of the tiled image (coalesced stores,
// 64-byte-segment loads).
__global__ __launch_bounds__(256) void q2c_tile_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                            int64_t rows, int k_bytes, int64_t n_chunks) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // chunk index in the tiled image
  if (i >= n_chunks) return;
  const int slices = k_bytes >> 6;
  const int c = (int)(i & 3);
  const int r = (int)((i >> 2) & 255);
  const int64_t ts = i >> 10;                                     // tile * slices + slice
  const int64_t tile = ts / slices;
  const int sl = (int)(ts - tile * slices);
  const int64_t row = tile * 256 + r;
  uint4 v = {0u, 0u, 0u, 0u};
  if (row < rows) v = src[(row * k_bytes + sl * 64 + c * 16) >> 4];
  dst[i] = v;
}

// Same tiled image, rows gathered through a table: dst row i = src row row_map[i] (zero when row_map[i] < 0).  Builds the
// length-bucketed corpus image (xml_q2c_tile_rows_gather).
__global__ __launch_bounds__(256) void q2c_tile_rows_gather_kernel(const uint4* __restrict__ src,
                                                                   const int32_t* __restrict__ row_map,
                                                                   uint4* __restrict__ dst, int k_bytes, int64_t n_chunks) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_chunks) return;
  const int slices = k_bytes >> 6;
  const int c = (int)(i & 3);
  const int r = (int)((i >> 2) & 255);
  const int64_t ts = i >> 10;
  const int64_t tile = ts / slices;
  const int sl = (int)(ts - tile * slices);
  const int64_t srow = row_map[tile * 256 + r];
  uint4 v = {0u, 0u, 0u, 0u};
  if (srow >= 0) v = src[(srow * k_bytes + sl * 64 + c * 16) >> 4];
  dst[i] = v;
}

extern "C" int xml_q2c_tile_rows_gather(const void* src, const int32_t* row_map, void* dst, int64_t rows_packed,
                                        int hidden, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!src || !row_map || !dst || rows_packed <= 0 || (rows_packed & 255) || hidden <= 0) return XML_ERR_BAD_ARG;
  if (dt != XML_F32 && dt != XML_BF16 && dt != XML_F16) return XML_ERR_BAD_ARG;
  const size_t kb = (size_t)hidden * dt_size(dt);
  if (kb % 64) return XML_ERR_UNSUPPORTED;
  const int64_t n_chunks = rows_packed * (int64_t)kb / 16;
  hipLaunchKernelGGL(q2c_tile_rows_gather_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const uint4*)src, row_map, (uint4*)dst, (int)kb, n_chunks);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" int xml_q2c_scores_packed(int n_mod, const void* qt0, const void* ct0, const void* qt1, const void* ct1,
                                     float* out, int64_t ld_out, int nq, int n_tiles, const int32_t* slot_ids, const uint32_t* mbits0, const uint32_t* mbits1, int hidden,
                                     int dt, xml_stream_t stream) {
  XML_ENTER();
  if ((n_mod != 1 && n_mod != 2) || !qt0 || !ct0 || !out || !slot_ids || !mbits0) return XML_ERR_BAD_ARG;
  if (n_mod == 2 && (!qt1 || !ct1 || !mbits1)) return XML_ERR_BAD_ARG;
  if (nq <= 0 || n_tiles <= 0 || hidden <= 0 || ld_out <= 0) return XML_ERR_BAD_ARG;
  if (!xml_q2c_tiled_ok(128, hidden, dt)) return XML_ERR_UNSUPPORTED;
  const void* q[2] = {qt0, n_mod == 2 ? qt1 : qt0};
  const void* c[2] = {ct0, n_mod == 2 ? ct1 : ct0};
  const float* m[2] = {nullptr, nullptr};
  const uint32_t* mb[2] = {mbits0, n_mod == 2 ? mbits1 : mbits0};
  // the kernel sees 2 * n_tiles "videos" of 128 columns (wave tiles); real ids come from slot_ids
  return xmli_q2c_scores_persist(n_mod, q, c, m, out, ld_out, nq, 2 * n_tiles, 128, hidden, dt, (hipStream_t)stream, true,
                                 3, mb, slot_ids);
}

extern "C" int xml_q2c_tiled_ok(int lpad, int hidden, int dt) {
  if (dt != XML_F32 && dt != XML_BF16 && dt != XML_F16) return 0;
  const size_t kb = (size_t)hidden * dt_size(dt);
  return lpad == 128 && kb % 128 == 0 && kb >= 384;
}

extern "C" int64_t xml_q2c_tiled_bytes(int64_t rows, int hidden, int dt) {
  if (rows < 0 || hidden <= 0 || (dt != XML_F32 && dt != XML_BF16 && dt != XML_F16)) return -1;
  return (rows + 255) / 256 * 256 * (int64_t)hidden * (int64_t)dt_size(dt);
}

extern "C" int xml_q2c_tile_rows(const void* src, void* dst, int64_t rows, int hidden, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!src || !dst || rows <= 0 || hidden <= 0) return XML_ERR_BAD_ARG;
  if (dt != XML_F32 && dt != XML_BF16 && dt != XML_F16) return XML_ERR_BAD_ARG;
  const size_t kb = (size_t)hidden * dt_size(dt);
  if (kb % 64) return XML_ERR_UNSUPPORTED;
  const int64_t n_chunks = xml_q2c_tiled_bytes(rows, hidden, dt) / 16;
  hipLaunchKernelGGL(q2c_tile_rows_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)src, (uint4*)dst, rows, (int)kb, n_chunks);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" int xml_q2c_scores_tiled(int n_mod, const void* qt0, const void* ct0, const float* mask0, const void* qt1,
                                    const void* ct1, const float* mask1, float* out, int64_t ld_out, int nq, int nv,
                                    int lpad, int hidden, int dt, int mask_mode, const uint32_t* mbits0,
                                    const uint32_t* mbits1, xml_stream_t stream) {
  XML_ENTER();
  if ((n_mod != 1 && n_mod != 2) || !qt0 || !ct0 || !mask0 || !out) return XML_ERR_BAD_ARG;
  if (n_mod == 2 && (!qt1 || !ct1 || !mask1)) return XML_ERR_BAD_ARG;
  if (nq <= 0 || nv <= 0 || hidden <= 0 || ld_out < nv) return XML_ERR_BAD_ARG;
  if (!xml_q2c_tiled_ok(lpad, hidden, dt)) return XML_ERR_UNSUPPORTED;
  const void* q[2] = {qt0, n_mod == 2 ? qt1 : qt0};
  const void* c[2] = {ct0, n_mod == 2 ? ct1 : ct0};
  const float* m[2] = {mask0, n_mod == 2 ? mask1 : mask0};
  if (mask_mode < 0 || mask_mode > 2) return XML_ERR_BAD_ARG;
  const uint32_t* mb[2] = {mbits0, n_mod == 2 ? mbits1 : mbits0};
  return xmli_q2c_scores_persist(n_mod, q, c, m, out, ld_out, nq, nv, lpad, hidden, dt, (hipStream_t)stream, true,
                                 mask_mode, mb, nullptr);
}
