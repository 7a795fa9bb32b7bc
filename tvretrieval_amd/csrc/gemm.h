// LDS-tiled MFMA mainloop shared by every contraction of the hot path (K1/K3/K4 projections, K6, K7).
//
// D[m][n] = sum_k A[m][k] * B[n][k]      A rows and B rows are both K-contiguous ("B^T input"), which is
// how nn.Linear weights (out,in), query vectors (Nq,H) and clip features (Nv*L,H) are laid out in HBM.
//
// Geometry (v1, register-staged, one barrier per K step):
//   block tile BM x BN, K step = 128 bytes per row (64 bf16 / 32 f32) = two 64-byte MFMA chunks;
//   WM x WN waves, each owning (BM/WM) x (BN/WN) outputs as 16x16 MFMA tiles, f32 accumulators;
//   LDS: 2 stages x (BM + BN) rows x 128 B.  16-byte slot c of row r is stored at slot c ^ ((r >> 1) & 7):
//   ds_write_b128 (8 lanes = one row) and ds_read_b128 (16 rows x one K slot) are then both conflict-free
//   on the 64-bank LDS (cdna guide 5.5 T2).
//   Global loads are 16 B per lane, 8 lanes per 128-byte row segment (fully coalesced), issued one K step
//   ahead into registers and written to the other LDS stage after the MFMAs of the current step.
#pragma once
#include <type_traits>

#include "common.h"

template <typename T, int BM_, int BN_, int WM_, int WN_>
struct GemmCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int NTHREADS = WM * WN * 64;
  static constexpr int ROWB = 128;                       // bytes of K per row per step
  static constexpr int KSTEP = ROWB / (int)sizeof(T);    // elements of K per step
  static constexpr int MT = BM / WM / 16;                // 16x16 tiles per wave along M
  static constexpr int NT = BN / WN / 16;                // ... along N
  static constexpr int ROWS_PER_PASS = NTHREADS / 8;     // 8 lanes cover one 128-byte row segment
  static constexpr int A_PASSES = BM / ROWS_PER_PASS;
  static constexpr int B_PASSES = BN / ROWS_PER_PASS;
  static constexpr int STAGE_BYTES = (BM + BN) * ROWB;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
  static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile/threads mismatch");
};

// v or zero, component-wise (a `c ? v : zero` on the 16-byte struct makes hipcc select between two ADDRESSES and park
// both operands in scratch memory)
__device__ __forceinline__ uint4 keep16(bool c, const uint4& v) {
  return make_uint4(c ? v.x : 0u, c ? v.y : 0u, c ? v.z : 0u, c ? v.w : 0u);
}

__device__ __forceinline__ int lds_slot_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// a_row(m) / b_row(n): return the byte pointer of tile-local row m / n, or nullptr when out of range
// (rows are then zero-filled).  k_bytes = K * sizeof(T), must be a multiple of 16.
template <typename T, typename Cfg, bool ZERO_INIT = true, typename ARow, typename BRow>
__device__ __forceinline__ void gemm_mainloop(f32x4 (&acc)[Cfg::MT][Cfg::NT], ARow a_row, BRow b_row,
                                              int k_bytes, char* smem, int mt_used = Cfg::MT) {
  // mt_used (block-uniform): only the first mt_used 16-row tiles of a wave's rows carry real rows (a ragged chunk of a
  // gathered pair list); the MFMAs and fragment reads of the others are skipped, their accumulators keep what they hold
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int lrow = tid >> 3;   // row within a pass
  const int lslot = tid & 7;   // 16-byte slot within the 128-byte segment

  // Row pointers.  Out-of-range rows / K tails load from a valid stand-in address and are zeroed when they are written
  // to LDS: an unconditional 16-byte GLOBAL load (see ld_global16_as1) instead of a predicated generic one.  The stand-in
  // is row 0 of either operand (precondition: valid for at least one of them -- true for every tile the callers launch).
  const char* const safe = a_row(0) ? a_row(0) : b_row(0);
  const char* aps[Cfg::A_PASSES];
  const char* bps[Cfg::B_PASSES];
  uint32_t row_ok = 0;                       // bit i: A pass i valid, bit 16 + i: B pass i valid
#pragma unroll
  for (int i = 0; i < Cfg::A_PASSES; ++i) {
    const char* p = a_row(lrow + i * Cfg::ROWS_PER_PASS);
    if (p) row_ok |= 1u << i;
    aps[i] = p ? p : safe;
  }
#pragma unroll
  for (int i = 0; i < Cfg::B_PASSES; ++i) {
    const char* p = b_row(lrow + i * Cfg::ROWS_PER_PASS);
    if (p) row_ok |= 1u << (16 + i);
    bps[i] = p ? p : safe;
  }

  if (ZERO_INIT) {
#pragma unroll
    for (int m = 0; m < Cfg::MT; ++m)
#pragma unroll
      for (int n = 0; n < Cfg::NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  uint4 ra[Cfg::A_PASSES], rb[Cfg::B_PASSES];
  const uint4 zero = make_uint4(0, 0, 0, 0);
  bool kin_loaded = true;
  auto gload = [&](int step) {
    const int off = step * Cfg::ROWB + lslot * 16;
    const bool kin = off < k_bytes;
    const int offc = kin ? off : 0;
    // raw loads only: the zeroing of invalid rows happens in lstore, where the values are consumed anyway (a select
    // here would make hipcc wait for the loads right away and lose the overlap with this step's MFMAs)
#pragma unroll
    for (int i = 0; i < Cfg::A_PASSES; ++i) ra[i] = ld_global16_as1(aps[i] + offc);
#pragma unroll
    for (int i = 0; i < Cfg::B_PASSES; ++i) rb[i] = ld_global16_as1(bps[i] + offc);
    kin_loaded = kin;
  };
  auto lstore = [&](int stage) {
    char* sa = smem + stage * Cfg::STAGE_BYTES;
    char* sb = sa + Cfg::BM * Cfg::ROWB;
#pragma unroll
    for (int i = 0; i < Cfg::A_PASSES; ++i) {
      const int r = lrow + i * Cfg::ROWS_PER_PASS;
      *reinterpret_cast<uint4*>(sa + lds_slot_off(r, lslot)) = keep16(kin_loaded && ((row_ok >> i) & 1u), ra[i]);
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_PASSES; ++i) {
      const int r = lrow + i * Cfg::ROWS_PER_PASS;
      *reinterpret_cast<uint4*>(sb + lds_slot_off(r, lslot)) = keep16(kin_loaded && ((row_ok >> (16 + i)) & 1u), rb[i]);
    }
  };

  const int nsteps = (k_bytes + Cfg::ROWB - 1) / Cfg::ROWB;
  gload(0);
  lstore(0);
  __syncthreads();
  const int fr = lane & 15, fg = lane >> 4;
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) gload(s + 1);
    const char* sa = smem + (s & 1) * Cfg::STAGE_BYTES;
    const char* sb = sa + Cfg::BM * Cfg::ROWB;
    if constexpr (IsSplit16<T>::value) {
      // split-f16 rows: the step's 128 bytes are [32 x hi | 32 x lo] of the same 32 k -> hi.hi + lo.hi + hi.lo
      uint4 fa[2][Cfg::MT], fb[2][Cfg::NT];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int m = 0; m < Cfg::MT; ++m) {
          const int r = wm * (Cfg::BM / Cfg::WM) + m * 16 + fr;
          if (m < mt_used) fa[c][m] = *reinterpret_cast<const uint4*>(sa + lds_slot_off(r, c * 4 + fg));
        }
#pragma unroll
        for (int n = 0; n < Cfg::NT; ++n) {
          const int r = wn * (Cfg::BN / Cfg::WN) + n * 16 + fr;
          fb[c][n] = *reinterpret_cast<const uint4*>(sb + lds_slot_off(r, c * 4 + fg));
        }
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int m = 0; m < Cfg::MT; ++m)
          if (m < mt_used) {
#pragma unroll
            for (int n = 0; n < Cfg::NT; ++n) Mma<f16_t>::chunk(acc[m][n], fa[t == 1 ? 1 : 0][m], fb[t == 2 ? 1 : 0][n]);
          }
    } else {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      uint4 fa[Cfg::MT], fb[Cfg::NT];
#pragma unroll
      for (int m = 0; m < Cfg::MT; ++m) {
        const int r = wm * (Cfg::BM / Cfg::WM) + m * 16 + fr;
        if (m < mt_used) fa[m] = *reinterpret_cast<const uint4*>(sa + lds_slot_off(r, c * 4 + fg));
      }
#pragma unroll
      for (int n = 0; n < Cfg::NT; ++n) {
        const int r = wn * (Cfg::BN / Cfg::WN) + n * 16 + fr;
        fb[n] = *reinterpret_cast<const uint4*>(sb + lds_slot_off(r, c * 4 + fg));
      }
#pragma unroll
      for (int m = 0; m < Cfg::MT; ++m)
        if (m < mt_used) {
#pragma unroll
          for (int n = 0; n < Cfg::NT; ++n) Mma<T>::chunk(acc[m][n], fa[m], fb[n]);
        }
    }
    }
    if (s + 1 < nsteps) lstore((s + 1) & 1);
    __syncthreads();
  }
}


// ---- deep-prefetch variant for few-row products ---------------------------------------------------------------------------
// Same LDS image, fragment reads and MFMA order as gemm_mainloop (bitwise the same accumulators), but the global loads run
// D K-steps ahead through D register sets instead of one.  A product with few rows has few workgroups and a short, strictly
// serial K loop: with one step of lead every step costs a full memory round trip (128 x 768 x 768: 12 steps, ~2 us each,
// whatever the tile size); with D sets in flight the loop costs nsteps / D round trips.  Requires nsteps % D == 0 (the loop
// body exists D times, one per register set: no dynamic register indexing, no branch between a load and its use -- hipcc
// keeps its vmcnt count and waits for exactly the set it stores).  Loads past the last step re-read step 0 and are dropped.
template <typename T, typename Cfg, int D, typename ARow, typename BRow>
__device__ __forceinline__ void gemm_mainloop_deep(f32x4 (&acc)[Cfg::MT][Cfg::NT], ARow a_row, BRow b_row, int k_bytes,
                                                   char* smem) {
  static_assert(!IsSplit16<T>::value, "plain element types only");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int lrow = tid >> 3, lslot = tid & 7;
  const char* const safe = a_row(0) ? a_row(0) : b_row(0);
  const char* aps[Cfg::A_PASSES];
  const char* bps[Cfg::B_PASSES];
  uint32_t row_ok = 0;
#pragma unroll
  for (int i = 0; i < Cfg::A_PASSES; ++i) {
    const char* p = a_row(lrow + i * Cfg::ROWS_PER_PASS);
    if (p) row_ok |= 1u << i;
    aps[i] = p ? p : safe;
  }
#pragma unroll
  for (int i = 0; i < Cfg::B_PASSES; ++i) {
    const char* p = b_row(lrow + i * Cfg::ROWS_PER_PASS);
    if (p) row_ok |= 1u << (16 + i);
    bps[i] = p ? p : safe;
  }
#pragma unroll
  for (int m = 0; m < Cfg::MT; ++m)
#pragma unroll
    for (int n = 0; n < Cfg::NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Regs { uint4 a[Cfg::A_PASSES], b[Cfg::B_PASSES]; };
  Regs r[D];
  const int nsteps = (k_bytes + Cfg::ROWB - 1) / Cfg::ROWB;
  auto gload = [&](Regs& q, int step) {
    const int off = step * Cfg::ROWB + lslot * 16;
    const int offc = off < k_bytes ? off : lslot * 16;      // (past the end: valid bytes that lstore drops)
#pragma unroll
    for (int i = 0; i < Cfg::A_PASSES; ++i) q.a[i] = ld_global16_as1(aps[i] + offc);
#pragma unroll
    for (int i = 0; i < Cfg::B_PASSES; ++i) q.b[i] = ld_global16_as1(bps[i] + offc);
  };
  auto lstore = [&](const Regs& q, int step) {
    char* sa = smem + (step & 1) * Cfg::STAGE_BYTES;
    char* sb = sa + Cfg::BM * Cfg::ROWB;
    const bool kin = step * Cfg::ROWB + lslot * 16 < k_bytes;
#pragma unroll
    for (int i = 0; i < Cfg::A_PASSES; ++i) {
      const int rr = lrow + i * Cfg::ROWS_PER_PASS;
      *reinterpret_cast<uint4*>(sa + lds_slot_off(rr, lslot)) = keep16(kin && ((row_ok >> i) & 1u), q.a[i]);
    }
#pragma unroll
    for (int i = 0; i < Cfg::B_PASSES; ++i) {
      const int rr = lrow + i * Cfg::ROWS_PER_PASS;
      *reinterpret_cast<uint4*>(sb + lds_slot_off(rr, lslot)) = keep16(kin && ((row_ok >> (16 + i)) & 1u), q.b[i]);
    }
  };
  const int fr = lane & 15, fg = lane >> 4;
  auto compute = [&](int step) {
    const char* sa = smem + (step & 1) * Cfg::STAGE_BYTES;
    const char* sb = sa + Cfg::BM * Cfg::ROWB;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      uint4 fa[Cfg::MT], fb[Cfg::NT];
#pragma unroll
      for (int m = 0; m < Cfg::MT; ++m)
        fa[m] = *reinterpret_cast<const uint4*>(sa + lds_slot_off(wm * (Cfg::BM / Cfg::WM) + m * 16 + fr, c * 4 + fg));
#pragma unroll
      for (int n = 0; n < Cfg::NT; ++n)
        fb[n] = *reinterpret_cast<const uint4*>(sb + lds_slot_off(wn * (Cfg::BN / Cfg::WN) + n * 16 + fr, c * 4 + fg));
#pragma unroll
      for (int m = 0; m < Cfg::MT; ++m)
#pragma unroll
        for (int n = 0; n < Cfg::NT; ++n) Mma<T>::chunk(acc[m][n], fa[m], fb[n]);
    }
  };
#pragma unroll
  for (int j = 0; j < D; ++j) gload(r[j], j);
  lstore(r[0], 0);
  __syncthreads();
  for (int s = 0; s < nsteps; s += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {      // step s + j: its set is in LDS, refill it with step s + j + D; sets j + 1 .. stay in flight
      gload(r[j], s + j + D);
      compute(s + j);
      lstore(r[(j + 1) % D], s + j + 1);      // (after the last step: a dropped store into the other stage)
      __syncthreads();
    }
  }
}

// ---- LDS-DMA variant of the mainloop (gathered-pair contractions: K7, exact-rank re-score) -------------------------------
// Same LDS image, same fragment reads and MFMA order as gemm_mainloop (bitwise the same accumulators), but the K steps
// arrive through `global_load_lds_dwordx4` into a THREE-stage ring: two steps (2 x 24 KiB per workgroup) are in flight
// behind the one being multiplied, no staging registers, one barrier per step.  The register-staged loop has one step in
// flight for the length of one step's MFMAs: a pair-list chunk spent most of its time waiting for HBM (f32 ConvSE:
// 2.5 TB/s and a half-busy MFMA pipe at three workgroups per CU).
//   A 1 KiB DMA piece = 8 tile rows x 128 B; lane i of a piece lands at LDS byte 16 i of it = row i / 8, PHYSICAL slot
//   i % 8, so it fetches the row's LOGICAL slot (i % 8) ^ ((row >> 1) & 7) -- the swizzle of lds_slot_off moves to the
//   global address.  Piece j of a step covers tile rows 8 j .. 8 j + 7 (A rows first), wave w issues pieces w, w + 4, ...
//   Rows are addressed as SGPR base + 32-bit byte offset (a_off / b_off: offset of tile row r from a_base / b_base; rows
//   without data must return the offset of some VALID row -- their products are never read, and with mt_used mostly not
//   computed).  Requirements: Cfg = <T, 64, 128, 1, 4> shape family (WM == 1, 256 threads), k_bytes % 128 == 0.
// vmcnt: the DMAs are invisible to hipcc's counters, the waits are hand-counted (PPW pieces per wave and step); any VMEM
// operation of the caller issued BEFORE this loop is older than every DMA and therefore covered by the first wait.
template <typename Cfg, int STAGES_ = 3> struct GemmDma {
  static constexpr int STAGES = STAGES_;
  static constexpr int PIECES = (Cfg::BM + Cfg::BN) / 8;
  static constexpr int PPW = PIECES / 4;
  static constexpr int LDS_BYTES = STAGES * Cfg::STAGE_BYTES;
  static_assert(Cfg::WM == 1 && Cfg::WN == 4 && Cfg::BM % 32 == 0 && Cfg::BN % 32 == 0, "piece / wave split");
};

__device__ __forceinline__ void gemm_dma_piece(uint32_t voff, const char* sbase, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}

template <typename T, typename Cfg, bool ZERO_INIT = true, int STAGES = 3, typename AOff, typename BOff>
__device__ __forceinline__ void gemm_mainloop_dma(f32x4 (&acc)[Cfg::MT][Cfg::NT], const char* a_base, AOff a_off,
                                                  const char* b_base, BOff b_off, int k_bytes, char* smem,
                                                  int mt_used = Cfg::MT) {
  using D = GemmDma<Cfg, STAGES>;
  static_assert(STAGES == 2 || STAGES == 3, "ring depth");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave;
  constexpr int A_PIECES = Cfg::BM / 8 / 4;                 // per wave
  uint32_t voff[D::PPW];
#pragma unroll
  for (int i = 0; i < D::PPW; ++i) {
    const int row = (wave + 4 * i) * 8 + (lane >> 3), phys = lane & 7;
    if (i < A_PIECES) voff[i] = a_off(row) + (uint32_t)((phys ^ ((row >> 1) & 7)) << 4);
    else { const int r = row - Cfg::BM; voff[i] = b_off(r) + (uint32_t)((phys ^ ((r >> 1) & 7)) << 4); }
  }
  // wave-uniform values the DMA instruction takes from SGPRs
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane(
      (int)((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)wave * 1024u));
  auto uni = [](const char* p) -> const char* {
    const uint64_t u = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
  };
  const char* ab = uni(a_base);
  const char* bb = uni(b_base);
  auto issue = [&](int step, int stage) {
    const uint32_t dst = lds0 + (uint32_t)stage * (uint32_t)Cfg::STAGE_BYTES;
    const char* sa = ab + (int64_t)step * Cfg::ROWB;
    const char* sb = bb + (int64_t)step * Cfg::ROWB;
#pragma unroll
    for (int i = 0; i < D::PPW; ++i) gemm_dma_piece(voff[i], i < A_PIECES ? sa : sb, dst + (uint32_t)i * 4096u);
  };
  if (ZERO_INIT) {
#pragma unroll
    for (int m = 0; m < Cfg::MT; ++m)
#pragma unroll
      for (int n = 0; n < Cfg::NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nsteps = k_bytes / Cfg::ROWB;
  const int fr = lane & 15, fg = lane >> 4;
  // the whole step loop once per number of live row tiles: straight-line MFMA blocks, no per-tile branches
  auto run = [&](auto mtu_c) {
    constexpr int MTU = decltype(mtu_c)::value;
    issue(0, 0);
    if (STAGES == 3 && nsteps > 1) issue(1, 1);
    int stage = 0;
    for (int s = 0; s < nsteps; ++s) {
      if (STAGES == 3 && s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(D::PPW) : "memory");   // my pieces of step s have landed
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();        // everyone's have; everyone is done reading the stage the next issue goes to
      if (STAGES == 3) { if (s + 2 < nsteps) issue(s + 2, stage >= 1 ? stage - 1 : 2); }
      else if (s + 1 < nsteps) issue(s + 1, stage ^ 1);
      const char* sa = smem + stage * Cfg::STAGE_BYTES;
      const char* sb = sa + Cfg::BM * Cfg::ROWB;
      if constexpr (IsSplit16<T>::value) {
        // split-f16 rows (see gemm_mainloop): [32 x hi | 32 x lo] per step -> hi.hi + lo.hi + hi.lo
        uint4 fa[2][MTU], fb[2][Cfg::NT];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
          for (int m = 0; m < MTU; ++m) fa[c][m] = *reinterpret_cast<const uint4*>(sa + lds_slot_off(m * 16 + fr, c * 4 + fg));
#pragma unroll
          for (int n = 0; n < Cfg::NT; ++n) {
            const int r = wn * (Cfg::BN / Cfg::WN) + n * 16 + fr;
            fb[c][n] = *reinterpret_cast<const uint4*>(sb + lds_slot_off(r, c * 4 + fg));
          }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int m = 0; m < MTU; ++m)
#pragma unroll
            for (int n = 0; n < Cfg::NT; ++n) Mma<f16_t>::chunk(acc[m][n], fa[t == 1 ? 1 : 0][m], fb[t == 2 ? 1 : 0][n]);
      } else {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        uint4 fa[MTU], fb[Cfg::NT];
#pragma unroll
        for (int m = 0; m < MTU; ++m) fa[m] = *reinterpret_cast<const uint4*>(sa + lds_slot_off(m * 16 + fr, c * 4 + fg));
#pragma unroll
        for (int n = 0; n < Cfg::NT; ++n) {
          const int r = wn * (Cfg::BN / Cfg::WN) + n * 16 + fr;
          fb[n] = *reinterpret_cast<const uint4*>(sb + lds_slot_off(r, c * 4 + fg));
        }
#pragma unroll
        for (int m = 0; m < MTU; ++m)
#pragma unroll
          for (int n = 0; n < Cfg::NT; ++n) Mma<T>::chunk(acc[m][n], fa[m], fb[n]);
      }
      }
      stage = stage == STAGES - 1 ? 0 : stage + 1;
    }
  };
  static_assert(Cfg::MT == 4 || Cfg::MT == 8, "row-tile dispatch below");
  if constexpr (Cfg::MT == 8) {      // 128-row chunks: live row tiles rounded up to 2 (four loop bodies, not eight)
    if (mt_used > 6) run(std::integral_constant<int, 8>{});
    else if (mt_used > 4) run(std::integral_constant<int, 6>{});
    else if (mt_used > 2) run(std::integral_constant<int, 4>{});
    else run(std::integral_constant<int, 2>{});
  } else {
    if (mt_used >= 4) run(std::integral_constant<int, 4>{});
    else if (mt_used == 3) run(std::integral_constant<int, 3>{});
    else if (mt_used == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 1>{});
  }
}
