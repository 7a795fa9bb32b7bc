// Weight-gradient GEMM of the training step, straight from the row-major operands:
//   dW (N, K) f32  =  dY^T X  =  sum_r dY[r][n] X[r][k]        dY (rows, N), X (rows, K) bf16, both as the layers hold them
//   reference: the autograd of nn.Linear inside XML.forward (xml/model_components.py:156-163; loss.backward(), xml/train.py:81)
// Both operands are contracted over their ROW index, i.e. both are needed transposed.  The unfused path transposed dY and X
// explicitly (two launches writing rows x (N + K) elements) and ran a split-K NT GEMM on the copies; here 32-row slabs of
// dY and X go to LDS as they lie (row-major, 16-byte loads / stores) and the MFMA fragments -- 8 consecutive rows of one
// column per lane -- come out of ds_read_b64_tr_b16 (attention.hip has the lane semantics).
// One workgroup = 4 waves = a 128 (n) x 128 (k) output tile over one range of rows; the row ranges (blockIdx.z) are combined
// with f32 atomics into the pre-zeroed output, as the split-K kernel did.
#include "common.h"

namespace {

constexpr int TN_STRIDE = 128 * 2 + 16;      // LDS row of a slab: 128 columns bf16 + 16 bytes pad

typedef short tn_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 tn_read_tr16(const char* p) {
  const tn_v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_v4s*)p);
  return __builtin_bit_cast(uint2, r);
}

// colsum (optional): colsum[n] += sum_r A[r][n] -- the bias gradient of the same layer, taken from the dY slabs the
// workgroups of the first k tile load anyway (one launch and one pass over dY less per layer).
__global__ __launch_bounds__(256, 3) void gemm_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                      float* __restrict__ out, float* __restrict__ colsum, int rows, int N,
                                                      int K, int rows_per_split, int accumulate) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 32 * TN_STRIDE];      // [buffer][operand][32 rows]
  __shared__ float s_cs[128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = blockIdx.y * 128, k0 = blockIdx.x * 128;
  const int r_begin = blockIdx.z * rows_per_split;
  const int r_end = min(rows, r_begin + rows_per_split);
  const int n_slabs = ((r_end - r_begin + 31) / 32 + 1) & ~1;      // even: the two-step loop body below has no tail case (a
                                                                   // slab past r_end is all zeros)

  // a thread's two 16-byte pieces of a slab of each operand: slab row lr + 16 i, columns 8 lc .. + 7
  const int lr = tid >> 4, lc = tid & 15;
  const bool a_ok = n0 + lc * 8 + 8 <= N, b_ok = k0 + lc * 8 + 8 <= K;       // N, K multiples of 8 (checked by the entry)
  const int ca = a_ok ? n0 + lc * 8 : 0, cb = b_ok ? k0 + lc * 8 : 0;        // clamped columns
  auto load = [&](uint4 (&ra)[2], uint4 (&rb)[2], int slab) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // UNCONDITIONAL loads from clamped (always valid) addresses; rows / columns outside the operands are zeroed when the
      // registers are parked in LDS (stash).  With the loads inside `if`s hipcc loses its vmcnt count at the branch joins and
      // waits vmcnt(0) before every LDS store, and a select right here would wait for the data at once: either way the
      // two-slab lead is gone.
      const int r = r_begin + slab * 32 + lr + 16 * i;
      const int rc = r < r_end ? r : r_begin;
      ra[i] = ld_global16(A + (int64_t)rc * N + ca);
      rb[i] = ld_global16(B + (int64_t)rc * K + cb);
    }
  };
  const bool do_cs = colsum != nullptr && blockIdx.x == 0;      // (workgroup-uniform)
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto stash = [&](const uint4 (&ra)[2], const uint4 (&rb)[2], int buf, int slab) {
    char* sa = smem + buf * (2 * 32 * TN_STRIDE);
    char* sb = sa + 32 * TN_STRIDE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool rok = r_begin + slab * 32 + lr + 16 * i < r_end;
      const bool ka = rok && a_ok, kb = rok && b_ok;
      const uint4 za = make_uint4(ka ? ra[i].x : 0u, ka ? ra[i].y : 0u, ka ? ra[i].z : 0u, ka ? ra[i].w : 0u);
      if (do_cs) {
        float f[8];
        unpack16<bf16_t>(za, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] += f[e];
      }
      *reinterpret_cast<uint4*>(sa + (lr + 16 * i) * TN_STRIDE + lc * 16) = za;
      *reinterpret_cast<uint4*>(sb + (lr + 16 * i) * TN_STRIDE + lc * 16) =
          make_uint4(kb ? rb[i].x : 0u, kb ? rb[i].y : 0u, kb ? rb[i].z : 0u, kb ? rb[i].w : 0u);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Loads run TWO slabs ahead through two register sets (a 16-MFMA slab is ~500 cycles of work, a global round trip
  // several times that: with one slab of lead every iteration waited for memory -- 64 us per 768 x 768 x 12 800 product):
  //   step s:  issue the loads of slab s + 2 into the set slab s came from, compute slab s from its LDS buffer, then park
  //            slab s + 1 (loaded one step earlier) in the other buffer.
  uint4 r0a[2], r0b[2], r1a[2], r1b[2];
  load(r0a, r0b, 0);
  load(r1a, r1b, 1);
  stash(r0a, r0b, 0, 0);
  __syncthreads();
  // fragment of column tile t (16 columns from column c0 + 16 t): lane = column fr, slab rows 8 fg .. 8 fg + 7
  const int frag_off = (fg * 8 + (fr >> 2)) * TN_STRIDE + (fr & 3) * 8;
  auto step = [&](int s, uint4 (&la)[2], uint4 (&lb)[2], const uint4 (&sa_)[2], const uint4 (&sb_)[2]) {
    const int buf = s & 1;
    load(la, lb, s + 2 < n_slabs ? s + 2 : n_slabs - 1);      // (unconditional: a branch here costs the vmcnt count again)
    __builtin_amdgcn_sched_barrier(0);                         // keep the loads HERE (hipcc sinks them below the MFMAs)
    const char* sa = smem + buf * (2 * 32 * TN_STRIDE) + frag_off + wn * 128;
    const char* sb = smem + buf * (2 * 32 * TN_STRIDE) + 32 * TN_STRIDE + frag_off + wk * 128;
    uint4 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint2 lo = tn_read_tr16(sa + i * 32), hi = tn_read_tr16(sa + i * 32 + 4 * TN_STRIDE);
      fa[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint2 lo = tn_read_tr16(sb + j * 32), hi = tn_read_tr16(sb + j * 32 + 4 * TN_STRIDE);
      fb[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) Mma<bf16_t>::chunk(acc[i][j], fa[i], fb[j]);
    stash(sa_, sb_, buf ^ 1, s + 1);      // the other buffer (after the last slab: zeros nobody reads): its last readers passed the previous barrier
    __syncthreads();
  };
  for (int s = 0; s < n_slabs; s += 2) {
    step(s, r0a, r0b, r1a, r1b);
    step(s + 1, r1a, r1b, r0a, r0b);
  }

  if (do_cs) {        // the 16 threads that share a column group meet in LDS, one global atomic per column
    if (tid < 128) s_cs[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(&s_cs[lc * 8 + e], cs[e]);
    __syncthreads();
    if (tid < 128 && n0 + tid < N) unsafeAtomicAdd(colsum + n0 + tid, s_cs[tid]);
  }
  const bool split = gridDim.z > 1 || accumulate;      // accumulate: out already holds a partial sum (a .grad buffer)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * 64 + i * 16 + fg * 4 + r;
      if (n >= N) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wk * 64 + j * 16 + fr;
        if (k >= K) continue;
        if (split) unsafeAtomicAdd(out + (int64_t)n * K + k, acc[i][j][r]);       // hardware f32 add (no CAS loop)
        else out[(int64_t)n * K + k] = acc[i][j][r];
      }
    }
}


// ---- the XCD-partitioned kernel (round 5) --------------------------------------------------------------------------------
// The kernel above runs ONE wave per SIMD (252 workgroups of 4 waves) and every step is a serial chain -- global load ->
// VGPR -> ds_write -> barrier -> transposed LDS read -> 16 MFMAs -- and its operands are re-fetched over the fabric by every
// tile (36 x 6.5 MB for a 768 x 768 product: 236 MB per launch).  Here
//   * the 8 XCDs split the ROWS: XCD x owns rows [x R, (x + 1) R) and its 32 workgroups tile the whole (N, K) output with
//     96 x 192 tiles (768 x 768 = 8 x 4 tiles = one per workgroup), all walking the same rows at the same time: an
//     operand byte crosses the fabric once per launch (FETCH_SIZE 39.5 MB, corrected), the other 3 / 7 readers of a panel
//     hit the XCD's L2 (TCC hit rate 0.75);
//   * a workgroup is 8 waves = two groups of 4; a step is 64 rows, group g multiplies rows 32 g .. 32 g + 31 of it
//     (two waves per SIMD, the second group runs its MFMAs first so the matrix pipe works through the other's preamble);
//   * slabs go global -> LDS by DMA (global_load_lds_dwordx4) into a 4-step ring, three steps in flight; no VGPR staging,
//     no ds_write; ONE barrier per 64 rows; the 5 DMA pieces of a wave are issued between the MFMAs of its block;
//   * the LDS image is laid out for ds_read_b64_tr_b16 (SQ_LDS_BANK_CONFLICT = 0; the padded row-major image of the kernel
//     above is 2-way conflicted on every read) AND so that every 8 consecutive lanes of a DMA instruction fetch one whole,
//     aligned 128-byte line of a global row.  A's 96 columns are fetched as the aligned 128-column window around them
//     (N % 192 == 0: the tile starts 0 or 32 columns into its window).
//     A lane group of a transposed read (32 lanes) takes 8 rows x 32 bytes (two 16-byte granules = 16 columns): rows
//     {4h .. 4h + 3} and {8 + 4h .. 8 + 4h + 3} (+ 16 for the upper lanes); conflict-free means those 8 x 32 bytes cover the
//     256-byte bank row once.  The image is made of 128-byte segments (one line: 8 granules of one row, 64 columns); a bank
//     row holds the segments of rows r (bit 3 clear) and r + 8 of one 64-column block, and row r stores its granules at
//     position g ^ (2 * (r & 3)) -- its 32-byte pairs permuted by XOR, so the four rows i = 0..3 of a lane group put the
//     pair they are asked for in four different places, while each lane quad still fetches one aligned 64-byte piece:
//         bank row = (c >> 3) * 16 + (r >> 4) * 8 + (r & 7),   half = (r >> 3) & 1,   slot = (c & 7) ^ (2 * (r & 3))
//     (c = granule index inside the operand's window);
//   * the two groups' accumulators meet in LDS after the loop and leave as 256-byte-wide rows (the 16 x 16 MFMA layout
//     gives 64-byte pieces).
// Where the time goes (768 x 768 over 12 800 rows, accumulate mode: 38 us against 50): ~13 us are the output atomics (8 row
// ranges x 2.4 MB of device-scope f32 adds, 1.6 us per range: abl 1) and the loop runs at ~1600-1750 cycles per 64-row step
// against 2 x 18 MFMAs = 600 per SIMD.  Stage timers (-DXML_TN_PROBE) and ablations say what it is NOT: not the fabric or
// the L2 (rows kept L2- or even L1-resident: same time, abl 4 / 6), not the LDS write-back of the DMA (the same fetches into
// registers: same time), not request size (64-byte pieces per lane quad: same time).  Every DMA piece costs the issuing wave
// ~100 cycles wherever it is issued (burst or between MFMAs) -- ~40 cycles of the CU's address unit per wave-wide 1 KB
// instruction, i.e. ~25 B/clk per CU, the rate K6 also streams at -- and a 96 x 192 tile needs 40 of them per 36 MFMAs per
// SIMD.  A larger tile would halve that ratio and double the row ranges (and the atomics); see DESIGN.md 14g.
constexpr int X_TN = 96, X_TK = 192;
constexpr int X_A_SLAB = 32 * 128 * 2;            // 8192: the 128-column (2 full lines per row) window that holds the tile's 96 columns
constexpr int X_B_SLAB = 32 * X_TK * 2;           // 12288
constexpr int X_SLAB = X_A_SLAB + X_B_SLAB;       // one group's 32 rows of both operands
constexpr int X_STEP = 2 * X_SLAB;                // 40960
constexpr int X_RING = 4;
constexpr int X_LDS = X_RING * X_STEP;            // 163840 = all of a CU's LDS
constexpr int X_STAGE_LD = X_TK + 4;              // f32 staging of the finished tile: 96 x 196 floats (75 264 bytes)

__device__ __forceinline__ void tn_dma16(uint32_t voff, const void* sbase, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

#ifdef XML_TN_PROBE      // (with -DXML_DEBUG_VARIANTS -DXML_TN_PROBE: the timers cost registers, the kernel then spills)
}  // namespace
__device__ unsigned long long g_tn_probe[8 * 8];      // [wave][stage] cycles of workgroup 0 (ablation 8)
extern "C" int xml_debug_read_tn_probe(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tn_probe), sizeof(g_tn_probe)) == hipSuccess ? 0 : -4;
}
namespace {
#define TN_T() (abl == 8 ? __builtin_amdgcn_s_memtime() : 0ull)
#else
#define TN_T() 0ull
#endif
__global__ __launch_bounds__(512, 1) void gemm_tn_xcd_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                          float* __restrict__ out, float* __restrict__ colsum, int rows,
                                                          int N, int K, int range_rows, int sub, int atomic_out, int abl_arg) {
#ifdef XML_DEBUG_VARIANTS
  const int abl = abl_arg;      // 1: no output, 4 / 6: L2- / L1-hot rows, 7: no loads in the loop, 8: stage timers (with -DXML_TN_PROBE)
#else
  constexpr int abl = 0;
#endif
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3, wn = w4 >> 1, wk = w4 & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int tiles_k = K / X_TK, n_tiles = (N / X_TN) * tiles_k, units = n_tiles * sub;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // ---- DMA source of this lane: (row, granule) of the LDS slot its 16 bytes land in ------------------------------------
  // instructions d = w4, w4 + 4 (A: 8 KB) and d = w4, w4 + 4, w4 + 8 (B: 12 KB); lane l of instruction d fills LDS slot 64 d + l
  auto dma_rc = [&](int q, int& r, int& c) {
    const int br = 4 * (w4 + 4 * q) + (lane >> 4), rr = br & 15;
    r = (rr >> 3) * 16 + ((lane >> 3) & 1) * 8 + (rr & 7);
    c = (br >> 4) * 8 + ((lane & 7) ^ (2 * (rr & 3)));
  };
  uint32_t dma_a[2], dma_b[3];      // byte offsets from the slab's first row (rows inside the matrix)
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    int r, c;
    dma_rc(q, r, c);
    if (q < 2) dma_a[q] = (uint32_t)r * (uint32_t)(N * 2) + c * 16;
    dma_b[q] = (uint32_t)r * (uint32_t)(K * 2) + c * 16;
  }
  // ---- transposed-read addresses (bytes inside an operand slab; "hi" half = + 1024): row r = 8 fg + 4 h + (fr >> 2),
  // granule c = c0 + 2 t + ((fr >> 1) & 1) of 16-column tile t
  auto tr_off = [&](int c_even) {
    const int c = c_even + ((fr >> 1) & 1);
    return (uint32_t)(((c >> 3) * 16 + (fg >> 1) * 8 + (fr >> 2)) * 256 + (fg & 1) * 128 + ((c & 7) ^ (2 * (fr >> 2))) * 16 +
                      (fr & 1) * 8);
  };
  uint32_t a_off[3], b_off[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) b_off[j] = X_A_SLAB + tr_off(2 * (wk * 6 + j));
  const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);      // bf16 1.0 x 8

  for (int u = blockIdx.x >> 3; u < units; u += per_xcd) {
    const int tile = u % n_tiles, range = xcd * sub + u / n_tiles;
    const int n0 = (tile / tiles_k) * X_TN, k0 = (tile % tiles_k) * X_TK;
    const int w0 = n0 & ~63;                                  // A's window: the aligned 128 columns around the tile's 96
#pragma unroll
    for (int i = 0; i < 3; ++i) a_off[i] = tr_off(((n0 - w0) >> 3) + 2 * (wn * 3 + i));
    const int r_begin = range * range_rows;
    const int r_end = min(rows, r_begin + range_rows);
    if (r_begin >= r_end) continue;
    const int n_steps = (r_end - r_begin + 63) >> 6;
    const bool do_cs = colsum != nullptr && k0 == 0 && wk == 0;

    // The 5 DMA pieces of a step are issued one at a time BETWEEN the MFMAs of a block (issue_piece); as a burst in front of the
    // block they stall the issuing wave just as long (measured, both ways ~450 cycles per wave and step).
    struct Piece { const bf16_t* sa; const bf16_t* sb; uint32_t dst; int lim; };
    auto prepare = [&](int s) {
      Piece p;
      const int row0 = min(r_begin + (abl == 4 ? (s & 3) : abl == 6 ? 0 : s) * 64 + grp * 32, rows - 1);      // (abl 4 / 6: L2- / L1-hot rows, wrong result)
      p.lim = abl == 6 ? 3 : rows - 1 - row0;
      p.sa = A + (int64_t)row0 * N + w0;
      p.sb = B + (int64_t)row0 * K + k0;
      p.dst = lds0 + (s & (X_RING - 1)) * X_STEP + grp * X_SLAB + w4 * 1024;
      return p;
    };
    auto issue_piece = [&](const Piece& p, int q) {      // q = 0, 1: A; 2 .. 4: B
      uint32_t v = q < 2 ? dma_a[q] : dma_b[q - 2];
      if (p.lim < 31) {      // (uniform, the last step of the matrix only) rows past the end: clamp to the last row
        int r, c;
        dma_rc(q < 2 ? q : q - 2, r, c);
        v = (uint32_t)min(r, p.lim) * (uint32_t)((q < 2 ? N : K) * 2) + c * 16;
      }
      if (q < 2) tn_dma16(v, p.sa, p.dst + q * 4096);
      else tn_dma16(v, p.sb, p.dst + X_A_SLAB + (q - 2) * 4096);
    };
    auto issue = [&](int s) {
      const Piece p = prepare(s);
#pragma unroll
      for (int q = 0; q < 5; ++q) issue_piece(p, q);
    };
    auto read_frags = [&](uint4 (&fa)[3], uint4 (&fb)[6], int s) {
      const char* slab = smem + (s & (X_RING - 1)) * X_STEP + grp * X_SLAB;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const uint2 lo = tn_read_tr16(slab + a_off[i]), hi = tn_read_tr16(slab + a_off[i] + 1024);
        fa[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const uint2 lo = tn_read_tr16(slab + b_off[j]), hi = tn_read_tr16(slab + b_off[j] + 1024);
        fb[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
      const int limit = r_end - (r_begin + s * 64 + grp * 32);      // rows of this group's slab inside the range
      if (limit < 32) {      // (group-uniform) the slab's rows past the range were fetched from clamped addresses: zero them in
        const int nv = limit - fg * 8;      // A, the products vanish.  This lane's 8 rows: the first nv are inside
        uint32_t m[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) m[d] = (nv > 2 * d ? 0x0000ffffu : 0u) | (nv > 2 * d + 1 ? 0xffff0000u : 0u);
#pragma unroll
        for (int i = 0; i < 3; ++i) fa[i] = make_uint4(fa[i].x & m[0], fa[i].y & m[1], fa[i].z & m[2], fa[i].w & m[3]);
      }
    };

    f32x4 acc[3][6], cs[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto mfma_block = [&](const uint4 (&fa)[3], const uint4 (&fb)[6], int s_issue) {
      const bool with_dma = s_issue < n_steps && abl != 7;      // (abl 7: no loads in the loop)
      Piece p = {};
      if (with_dma) p = prepare(s_issue);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          Mma<bf16_t>::chunk(acc[i][j], fa[i], fb[j]);
          const int idx = i * 6 + j;
          if (idx % 3 == 2 && idx < 15) {      // after MFMA 3, 6, 9, 12, 15: one piece
            __builtin_amdgcn_sched_barrier(0);
            if (with_dma) issue_piece(p, idx / 3);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      if (do_cs) {
#pragma unroll
        for (int i = 0; i < 3; ++i) Mma<bf16_t>::chunk(cs[i], fa[i], ones);
      }
    };
    unsigned long long pr_wait = 0, pr_bar = 0, pr_pre = 0, pr_mma = 0, pr_post = 0;
    auto iter = [&](int s, uint4 (&ca)[3], uint4 (&cb)[6], uint4 (&na)[3], uint4 (&nb)[6]) {
      const bool more = s + 1 < n_steps;
      unsigned long long t0 = TN_T(), t1 = t0, t2 = t0;
      __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0), every iteration: hipcc then knows the current fragments are complete
                                               // and does not make their MFMAs wait for the NEXT fragments' reads
      if (more) {
        // my DMAs of step s + 1 have landed (the two younger steps stay in flight), my LDS reads of step s have returned
        if (s + 3 < n_steps) __builtin_amdgcn_s_waitcnt(0x0f7a);      // vmcnt(10): 5 pieces per wave and step
        else __builtin_amdgcn_s_waitcnt(0x0f70);
        t1 = TN_T();
        __builtin_amdgcn_s_barrier();      // everyone's pieces of step s + 1 are in LDS; nobody reads step s's slot any more
        t2 = TN_T();
      }
      if (grp == 0 && more) read_frags(na, nb, s + 1);
      const unsigned long long t3 = TN_T();
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(ca, cb, s + X_RING);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t4 = TN_T();
      if (grp == 1 && more) read_frags(na, nb, s + 1);
#ifdef XML_TN_PROBE
      if (abl == 8) {
        const unsigned long long t5 = TN_T();
        pr_wait += t1 - t0; pr_bar += t2 - t1; pr_pre += t3 - t2; pr_mma += t4 - t3; pr_post += t5 - t4;
      }
#endif
    };

    uint4 f0a[3], f0b[6], f1a[3], f1b[6];
#pragma unroll
    for (int p = 0; p < X_RING; ++p)
      if (p < n_steps) issue(p);
    if (n_steps >= X_RING) {
      __builtin_amdgcn_s_waitcnt(0x007f);                      // vmcnt(15): three younger steps x 5 pieces
    } else {
      __builtin_amdgcn_s_waitcnt(0x0070);
    }
    __builtin_amdgcn_s_barrier();
    read_frags(f0a, f0b, 0);
    for (int s = 0; s < n_steps; s += 2) {
      iter(s, f0a, f0b, f1a, f1b);
      if (s + 1 < n_steps) iter(s + 1, f1a, f1b, f0a, f0b);
    }

#ifdef XML_TN_PROBE
    if (abl == 8 && blockIdx.x == 0 && lane == 0) {
      unsigned long long* p = g_tn_probe + wave * 8;
      p[0] = pr_wait; p[1] = pr_bar; p[2] = pr_pre; p[3] = pr_mma; p[4] = pr_post; p[5] = n_steps;
    }
#endif
    // ---- the two groups' halves meet in LDS; the tile leaves as 256-byte rows ---------------------------------------------
    __syncthreads();
    float* stage = reinterpret_cast<float*>(smem);
    if (grp == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stage[(wn * 48 + i * 16 + fg * 4 + r) * X_STAGE_LD + wk * 96 + j * 16 + fr] = acc[i][j][r];
    }
    __syncthreads();
    if (grp == 1) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stage[(wn * 48 + i * 16 + fg * 4 + r) * X_STAGE_LD + wk * 96 + j * 16 + fr] += acc[i][j][r];
    }
    __syncthreads();
    if (abl != 1) {
      for (int row = wave; row < X_TN; row += 8) {
        float* o = out + (int64_t)(n0 + row) * K + k0 + lane;
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
          const float v = stage[row * X_STAGE_LD + c3 * 64 + lane];
          if (atomic_out) unsafeAtomicAdd(o + c3 * 64, v);
          else o[c3 * 64] = v;
        }
      }
    }
    if (do_cs && fr == 0) {      // every column of the 16 x 16 result holds the same sums
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(colsum + n0 + wn * 48 + i * 16 + fg * 4 + r, cs[i][r]);
    }
    __syncthreads();      // the staging area is the ring of the next tile
  }
}

}  // namespace

extern "C" int xml_gemm_tn_supported(int64_t rows, int N, int K, int dt) {
  return dt == XML_BF16 && rows > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0;
}

extern "C" int xml_gemm_tn(const void* A, const void* B, float* out, float* colsum_a, int64_t rows, int N, int K, int dt,
                           int accumulate, xml_stream_t stream) {
  XML_ENTER();
  if (!A || !B || !out || rows <= 0 || rows > 0x7fffffff || N <= 0 || K <= 0) return XML_ERR_BAD_ARG;
  if (!xml_gemm_tn_supported(rows, N, K, dt)) return XML_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  // The XCD-partitioned kernel: whole 96 x 192 tiles, A rows of whole 128-column windows, 16-byte aligned operands.  It always
  // splits the rows 8 ways (one range per XCD), so its output atomics grow with N x K x 8: used where that is cheaper than the
  // 128 x 128 kernel's re-fetching (accumulate mode, us: 768 x 768 x 12 800 rows 38 vs 50, 1536 x 768 70 vs 77, 2304 x 768 97 vs
  // 104, 768 x 768 x 3 840 rows 23 vs 28; NOT 768 x 3072 x 12 800: 126 vs 101, nor 2304 x 768 x 3 840: 60 vs 41).
  const int x_tiles = (N / X_TN) * (K / X_TK);
  bool xcd_path = N % (2 * X_TN) == 0 && K % X_TK == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0 &&
                  ((x_tiles <= 32 && rows >= 2048) || (x_tiles <= 96 && rows >= 8192));
#ifdef XML_DEBUG_VARIANTS
  if (g_q2c_ablation == 300) xcd_path = false;                 // A/B: the 128 x 128 kernel
  if (g_q2c_ablation == 299) xcd_path = N % (2 * X_TN) == 0 && K % X_TK == 0 && rows >= 2048;      // A/B: whenever it can
#endif
  if (xcd_path) {
    const int n_tiles = x_tiles;
    const int sub = n_tiles >= 32 ? 1 : 32 / n_tiles;          // fewer than 32 tiles: the XCD's workgroups split its rows further
    const int n_ranges = 8 * sub;
    const int range_rows = (int)((cdiv(rows, n_ranges) + 63) / 64 * 64);
    if (!accumulate) {
      if (colsum_a == out + (size_t)N * K) {
        if (!xml_zero_async(out, ((size_t)N * K + N) * 4, st)) return XML_ERR_LAUNCH;
      } else {
        if (!xml_zero_async(out, (size_t)N * K * 4, st)) return XML_ERR_LAUNCH;
        if (colsum_a && !xml_zero_async(colsum_a, (size_t)N * 4, st)) return XML_ERR_LAUNCH;
      }
    }
    if (!xml_lds_attr_once<gemm_tn_xcd_kernel>(X_LDS)) return XML_ERR_LAUNCH;
    int abl = 0;
#ifdef XML_DEBUG_VARIANTS
    if (g_q2c_ablation > 300 && g_q2c_ablation < 320) abl = g_q2c_ablation - 300;   // 301: no output, 304 / 306: L2- / L1-hot rows, 307: no loads, 308: timers (see the kernel)
#endif
    hipLaunchKernelGGL(gemm_tn_xcd_kernel, dim3(256), dim3(512), X_LDS, st, (const bf16_t*)A, (const bf16_t*)B, out,
                       colsum_a, (int)rows, N, K, range_rows, sub, 1, abl);
    XML_CHECK_LAUNCH();
    return XML_OK;
  }
  const int tiles = cdiv(N, 128) * cdiv(K, 128);
  // workgroups aimed at.  Every row range adds N x K f32 atomics, and device-scope float atomics are slow enough to show
  // (768 x 768 over 12 800 rows: 58.6 us with 7 ranges, 69.6 with 22, 94 with 43 -- tools/bench_gemm_tn.py); with many
  // output tiles the extra ranges pay for themselves by hiding the global-load latency (768 x 3072: 134 vs 160 us)
  int target = (tiles <= 48 || rows < 8192) ? 256 : 512;      // (3 840 rows x 2304 x 768: 46 us at 256, 55 at 512, 67 at 768)
#ifdef XML_DEBUG_VARIANTS
  if (g_q2c_ablation >= 200 && g_q2c_ablation < 300) target = (g_q2c_ablation - 200) * 64;      // A/B: workgroups aimed at = (XML_ABL - 200) x 64
#endif
  int splits = cdiv(target, tiles);
  int rps = (cdiv(rows, splits) + 31) / 32 * 32;               // rows per workgroup, whole 32-row slabs
  if (rps < 256) rps = 256;
  splits = cdiv(rows, rps);
  if (accumulate) {
    // out / colsum_a are gradient buffers that already hold a (possibly zero) partial sum: every contribution is an atomic add
  } else if (splits > 1 && colsum_a == out + (size_t)N * K) {   // caller laid them out back to back: one fill
    if (!xml_zero_async(out, ((size_t)N * K + N) * 4, st)) return XML_ERR_LAUNCH;
  } else {
    if (splits > 1 && !xml_zero_async(out, (size_t)N * K * 4, st)) return XML_ERR_LAUNCH;
    if (colsum_a && !xml_zero_async(colsum_a, (size_t)N * 4, st)) return XML_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(cdiv(K, 128), cdiv(N, 128), splits), dim3(256), 0, st, (const bf16_t*)A,
                     (const bf16_t*)B, out, colsum_a, (int)rows, N, K, rps, accumulate ? 1 : 0);
  XML_CHECK_LAUNCH();
  return XML_OK;
}
