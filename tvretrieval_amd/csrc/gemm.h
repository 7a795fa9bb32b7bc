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
                                              int k_bytes, char* smem) {
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
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      uint4 fa[Cfg::MT], fb[Cfg::NT];
#pragma unroll
      for (int m = 0; m < Cfg::MT; ++m) {
        const int r = wm * (Cfg::BM / Cfg::WM) + m * 16 + fr;
        fa[m] = *reinterpret_cast<const uint4*>(sa + lds_slot_off(r, c * 4 + fg));
      }
#pragma unroll
      for (int n = 0; n < Cfg::NT; ++n) {
        const int r = wn * (Cfg::BN / Cfg::WN) + n * 16 + fr;
        fb[n] = *reinterpret_cast<const uint4*>(sb + lds_slot_off(r, c * 4 + fg));
      }
#pragma unroll
      for (int m = 0; m < Cfg::MT; ++m)
#pragma unroll
        for (int n = 0; n < Cfg::NT; ++n) Mma<T>::chunk(acc[m][n], fa[m], fb[n]);
    }
    if (s + 1 < nsteps) lstore((s + 1) & 1);
    __syncthreads();
  }
}

