// Persistent projection GEMM: the K-loop of gemm256.hip / q2c_persist.hip, one workgroup per CU walking many tiles
// with a DMA stream that never drains.
//   out[m][n] = act(A[m] . W[n] + bias[n]) + addend        A (M, K), W (N, K) both K-contiguous
// Why: with one 256 x 256 tile per workgroup the encoder projections (K = 768: 24 slices) spent as long outside the
// K loop as inside it -- T(tile) = 19.5 us + 0.81 us x slices (tools/bench_gemm.py: 660-700 TF at K = 768, 1060 TF at
// K = 3072).  A timing probe in this kernel showed where: not the launch or the first DMA round trip (~1.5 us) but the
// EPILOGUE, 34 K of 74 K cycles per tile:
//   * 640 KB of code: the per-element tail path (N % 8 != 0) with its 64-bit modulo, unrolled 16 times, sat between the
//     hot instructions -> instruction-fetch bound.  Now a separate instantiation of gemm256.hip; here N % 8 == 0.
//   * software f32 -> bf16 rounding with a NaN branch per element (~25 instructions each) -> v_cvt_pk_bf16_f32.
//   * hipcc cannot count the inline-asm DMAs and guarded every use of the bias registers with s_waitcnt vmcnt(0),
//     which also waits for the stores just issued -> the values pass through an empty asm once per tile.
//   * f32 rows written as 16-byte pieces at a 32-byte stride (every 128-byte line half-written per instruction, twice
//     the write requests; the CU retires about one write request per 5 cycles) -> two column groups 64 apart.
// What persistence itself adds: the issue side of the stream runs into the next tile while the current one finishes:
// 750-810 -> 815-865 TF at K = 768 for >= 1024 tiles (the fixes above lifted both kernels from 660-700).
// Round 2: the epilogue no longer goes through LDS at all (DIRECT, below): swapped MFMA operands put 4 consecutive
// columns of one row in a lane, DPP row shifts pair neighbouring rows so that every store instruction writes whole
// 128-byte lines: 734-793 -> 826-893 TF at K = 768, 1065 -> 1170-1220 TF at K = 3072 (same box, tools/bench_gemm.py,
// XML_GEMM_VARIANT=3 is the staged epilogue).  A phase probe (tools/probe_gemm.py) puts the K loop at 83 % of a tile's
// time now.  Tried on the way and NOT kept, both neutral: starting the workgroups a quarter tile apart (the epilogues of
// the chip are not what limits the write path -- each CU's own store queue is), and skipping the counted vmcnt waits for
// the first slices of a tile, which are already resident (peeling three steps made hipcc spill into the K loop prologue).
// A fifth ring slot for this kernel (the direct epilogue leaves LDS free; bias in registers): neutral as well -- unlike K6,
// this K loop is not short of bytes in flight.
// Same MFMA sequence per accumulator as gemm256_kernel: bitwise the same results.
#include <type_traits>

#include "common.h"
#include "internal.h"

struct G256pArgs {
  const void* A;
  const void* W;
  const float* bias;
  const void* addend;
  void* out;
  int64_t M, n_tiles;
  int N, K, relu, add_mode, seq_len, tn;
  // LayerNorm epilogue (LNE kernels): y = LN(act(A W^T + bias) + addend) * g + b over the full rows of N = tn * 256 columns
  float* ln_scratch;    // per workgroup (tn - 1) x 256 x 256 f32: the pre-LayerNorm values of a row block's first tn - 1 tiles
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
  int probe;            // debug build only: phase probe on
};

__device__ __forceinline__ void g256p_dma_pair(uint32_t v0, uint32_t v1, const char* sb, uint32_t lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3"
      :
      : "v"(v0), "v"(v1), "s"(lds_dst), "s"(sb)
      : "memory", "scc");
}
__device__ __forceinline__ void g256p_dma_quad(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, const char* sb,
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
template <typename T> struct G256pInit;      // first K chunk of a tile: C = 0 as an inline constant
template <> struct G256pInit<float> {
  __device__ static __forceinline__ void chunk(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};
template <> struct G256pInit<bf16_t> {
  __device__ static __forceinline__ void chunk(f32x4& acc, const uint4& a, const uint4& b) {
    union { uint4 u; bf16x8_v v; } ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  }
};
#ifdef XML_DEBUG_VARIANTS
// phase probe (debug build, XML_ABL = 9): s_memtime ticks of workgroup 0, summed over its tiles, per wave:
//   [0] K loop  [1] epilogue until the bias is in LDS (includes the vmcnt(0) drain)  [2] row blocks + stores
//   [3] LNE publish / wait / statistics  [4] LNE pass 2  [5] whole kernel  [6] tiles
__device__ unsigned long long g_g256p_probe[8 * 8];
__device__ unsigned int g_g256p_steps[8 * 64];       // per wave: ticks summed per slice step index (first 64 steps of a tile)
extern "C" int xml_debug_read_gemm_steps(unsigned int* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_g256p_steps), sizeof(unsigned int) * 512) == hipSuccess ? 0 : -4;
}
extern "C" int xml_debug_read_gemm_probe(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_g256p_probe), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -4;
}
#define G256P_T() ((a.probe & 1) ? __builtin_amdgcn_s_memtime() : 0ull)
#else
#define G256P_T() 0ull
#endif
template <int CTRL>
__device__ __forceinline__ uint4 dpp_u4(const uint4& v) {
  uint4 r;
  r.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, CTRL, 0xf, 0xf, false);
  r.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, CTRL, 0xf, 0xf, false);
  r.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.z, CTRL, 0xf, 0xf, false);
  r.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, CTRL, 0xf, 0xf, false);
  return r;
}
// lane id from the hardware, as a VOLATILE asm: neither hoisted out of the tile loop nor kept live across the K loop
// (see lane_id_now in q2c_persist.hip: a spilled lane id returns through a scratch load and its s_waitcnt vmcnt(0))
__device__ __forceinline__ int g256p_lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ int g256p_swz(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

// LNE: LayerNorm fused into the epilogue (K2 behind K1, BertSelfOutput; xml/model_components.py:76-89,313-317).  A row of
// the output spans tn = N / 256 column tiles (tn <= 3).  ONE workgroup computes all tn tiles of a row block, back to back:
//   tiles 0 .. tn - 2: the pre-LayerNorm values (f32) go to the workgroup's private scratch, lane for lane as they sit in
//     the accumulators (1 KiB per store instruction), and per row and 128-column segment (sum, centred sum of squares) --
//     computed from the f32 values -- go to LDS;
//   tile tn - 1: its values stay in the accumulators; the 2 tn segment partials of every row are combined in a fixed order
//     (Chan's formula: deterministic, no E[x^2] - mean^2 cancellation), the tile is normalised from the registers and the
//     earlier tiles from the scratch (written by this CU a tile ago: L2 hits), all stored once in the output type.
// Nothing leaves the workgroup: no counters, no waiting for other workgroups, no assumption about who is resident
// (rounds 2-5 had the column tiles on tn workgroups that exchanged partials through memory and waited for each other).
// The arithmetic is that exchange's, value for value: the results are bitwise unchanged.
//
// DIRECT epilogue (default): the MFMA operands are SWAPPED (first operand = W fragment, second = A fragment), so an
// accumulator holds 4 consecutive output COLUMNS of ONE output row per lane (row = lane & 15, columns 4 (lane >> 4) + r)
// instead of 4 rows of one column, and W's rows are assigned to MFMA row indices such that two accumulators of a lane sit
// side by side (bf16 out: 8 columns = one 16-byte store).  The results go from the accumulators straight to global
// memory -- no LDS transpose (the former epilogue moved the tile through 4 KiB patches: 256 ds_write_b32 + 32
// ds_read_b128 + 16 wave barriers per wave and tile; debug variant 3 keeps it for A/B measurements).  The products and the
// order of the K summation per output element are unchanged: bitwise the same results.
template <typename T, typename OutT, typename AddT, bool LNE = false, bool DIRECT = true>
__global__ __launch_bounds__(512) void gemm256p_kernel(G256pArgs a) {
  static_assert(!LNE || DIRECT, "the LayerNorm epilogue works on the DIRECT accumulator layout");
  constexpr int ROWB = 64;
  constexpr int OPER_BYTES = 256 * ROWB;
  constexpr int SLOT_BYTES = 2 * OPER_BYTES;
  constexpr int NSLOT = 4;
  constexpr int RING_BYTES = NSLOT * SLOT_BYTES;      // 128 KiB; the eight 4 KiB epilogue patches follow
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int grp = wave >> 2;                       // waves w and w + 4 share a SIMD
  const int fr = lane & 15, fg = lane >> 4;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int k_bytes = a.K * (int)sizeof(T);
  const int n_slices = k_bytes / ROWB;             // even, >= 4 (eligibility)
  const T* A = reinterpret_cast<const T*>(a.A);
  const T* W = reinterpret_cast<const T*>(a.W);
  const AddT* addend = reinterpret_cast<const AddT*>(a.addend);
  OutT* out = reinterpret_cast<OutT*>(a.out);
  const int64_t M = a.M;
  const int N = a.N;

  // tile walk: round k hands the 32 workgroups of XCD x the 32 consecutive tiles (8 k + x) * 32 .. + 31, N fastest:
  // the workgroups of an XCD share a few row tiles of A through its L2, W stays resident
  // LNE: the unit is a ROW BLOCK -- round r hands workgroup (x, loc) row block (8 r + x) * 32 + loc, whose tn tiles it runs
  // consecutively (k = r * tn + nt): the A rows are fetched from HBM once and twice more from L2, the 32 workgroups of an
  // XCD walk the same W tile at about the same time.
  const int tn_s = __builtin_amdgcn_readfirstlane(a.tn);
  const int64_t n_rb = a.n_tiles / tn_s;
  auto lne_split = [&](int k, int& r, int& nt) {            // k -> (round, column tile); tn <= 3, k < 98 304
    r = tn_s == 1 ? k : tn_s == 2 ? (k >> 1) : (int)(((uint32_t)k * 43691u) >> 17);
    nt = k - r * tn_s;
  };
  auto tile_of = [&](int k) -> int64_t {
    if constexpr (LNE) {
      int r, nt;
      lne_split(k, r, nt);
      const int64_t rb = ((int64_t)r * 8 + xcd) * 32 + loc;
      return rb < n_rb ? rb * tn_s + nt : a.n_tiles;
    }
    return ((int64_t)k * 8 + xcd) * 32 + loc;
  };
  auto tile_mt_nt = [&](int k, int64_t& mt, int& nt) {      // (row block, column tile) of this workgroup's k-th tile
    if constexpr (LNE) {
      int r;
      lne_split(k, r, nt);
      mt = ((int64_t)r * 8 + xcd) * 32 + loc;               // scalar arithmetic on uniform ints: the DMA bases stay in SGPRs
    } else {
      const int64_t lin = tile_of(k);
      mt = lin / a.tn;
      nt = (int)(lin - mt * a.tn);
    }
  };

  // ---- issue side ---------------------------------------------------------------------------------------------
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  int i_k = 0, i_slice = 0, i_slot = 0, i_left = n_slices, i_inc = 1;
  uint32_t va[4] = {}, vb[2] = {};
  const char* sbase_a = nullptr;
  const char* sbase_b = nullptr;
  auto setup_issue_tile = [&]() {
    int64_t mt; int nt;
    tile_mt_nt(i_k, mt, nt);
    const int64_t m0 = mt * 256;
    const int n0 = nt * 256;
    const int lane_o = g256p_lane_now();           // re-derived here: nothing lane-dependent stays live across the MFMA loop
    const int rsub = lane_o >> 2, pslot = lane_o & 3;
    auto off_a = [&](int piece) -> uint32_t {
      const int row = piece * 16 + rsub;
      return (uint32_t)((m0 + row < M) ? row : 0) * k_bytes + (pslot ^ g256p_swz(row)) * 16;
    };
    auto off_b = [&](int piece) -> uint32_t {
      const int row = piece * 16 + rsub;
      return (uint32_t)((n0 + row < N) ? row : 0) * k_bytes + (pslot ^ g256p_swz(row)) * 16;
    };
    if (grp == 0) {     // uneven duty (see q2c_persist.hip): the group that parks at the barrier issues 6 of 8 pieces
      va[0] = off_a(wave * 4); va[1] = off_a(wave * 4 + 1); va[2] = off_a(wave * 4 + 2); va[3] = off_a(wave * 4 + 3);
      vb[0] = off_b(wave * 2); vb[1] = off_b(wave * 2 + 1);
    } else {
      vb[0] = off_b(8 + (wave - 4) * 2); vb[1] = off_b(9 + (wave - 4) * 2);
    }
    sbase_a = reinterpret_cast<const char*>(A) + ((a.probe & 2) ? (int64_t)(blockIdx.x & 31) * 256 : m0) * k_bytes;   // probe bit 1: A from a resident 8192-row window (timing only)
    sbase_b = reinterpret_cast<const char*>(W) + (int64_t)n0 * k_bytes;
  };
  auto issue_slice = [&]() {
    const int koff = i_slice * ROWB;
    const uint32_t slot0 = lds0 + i_slot * SLOT_BYTES;
    if (grp == 0) {
      g256p_dma_quad(va[0], va[1], va[2], va[3], sbase_a + koff, slot0 + wave * 4096);
      g256p_dma_pair(vb[0], vb[1], sbase_b + koff, slot0 + OPER_BYTES + wave * 2048);
    } else {
      g256p_dma_pair(vb[0], vb[1], sbase_b + koff, slot0 + OPER_BYTES + 8192 + (wave - 4) * 2048);
    }
    if (++i_slot == NSLOT) i_slot = 0;
    i_slice += i_inc;
    if (--i_left == 0) {                           // next tile; once the walk is exhausted the stream SATURATES:
      i_slice = 0;                                 // it keeps re-fetching slice 0 of the last tile (valid addresses,
      i_left = n_slices;                           // dead data), so the consumer side has no end-of-stream cases
      ++i_k;
      if (tile_of(i_k) < a.n_tiles) setup_issue_tile();
      else { i_left = 0x7fffffff; i_inc = 0; }
    }
  };
  if (tile_of(0) >= a.n_tiles) return;
  setup_issue_tile();

  // ---- compute side ---------------------------------------------------------------------------------------------
  const int a_off = (wm * 64 + fr) * ROWB + ((fg ^ g256p_swz(fr)) << 4);
  // B (= W) fragment n, MFMA row index i = fr  ->  row of the wave's 128 W rows:
  //   staged epilogue / f32 out:  n * 16 + i                                   (a lane's 4 columns: n * 16 + 4 fg + r)
  //   DIRECT, 2-byte out (PAIRED): (n >> 1) * 32 + (i >> 2) * 8 + (n & 1) * 4 + (i & 3)
  //                               (accumulators 2 q and 2 q + 1 of a lane = columns q * 32 + 8 fg + 0..7)
  // Either way the 16 lanes of a ds_read_b128 group hit 16 distinct (row mod 4, swizzled slot) pairs: no bank conflicts.
  constexpr bool PAIRED = DIRECT && sizeof(OutT) == 2;
  const int rb_e = PAIRED ? (fr >> 2) * 8 + (fr & 3) : fr;
  const int b_off = OPER_BYTES + (wn * 128 + rb_e) * ROWB + ((fg ^ g256p_swz(rb_e)) << 4);
  const int b_off_o = OPER_BYTES + (wn * 128 + rb_e + 4) * ROWB + ((fg ^ g256p_swz(rb_e + 4)) << 4);   // PAIRED, odd n
  auto b_frag = [&](const char* slot, int n) -> uint4 {
    if constexpr (PAIRED) return *reinterpret_cast<const uint4*>(slot + ((n & 1) ? b_off_o : b_off) + (n >> 1) * 32 * ROWB);
    else return *reinterpret_cast<const uint4*>(slot + b_off + n * 16 * ROWB);
  };
  int c_k = 0, c_slot = 0;
  issue_slice(); issue_slice(); issue_slice(); issue_slice();
  if (grp == 0) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  uint4 faA[4], faB[4], fbL[4], fbH[4];

  auto run = [&](auto grp_tag) {
  constexpr bool GRP1 = decltype(grp_tag)::value;
  unsigned long long pr[8] = {};
  const unsigned long long pr_t0 = G256P_T();
  const unsigned long long pr_r0 = (a.probe & 1) ? __builtin_amdgcn_s_memrealtime() : 0ull;      // 100 MHz reference clock
#ifdef XML_DEBUG_VARIANTS
  unsigned int* sp = reinterpret_cast<unsigned int*>(smem + RING_BYTES + wave * 4096 + 2048);     // this wave's patch, upper half
  if ((a.probe & 1) && lane < 64) sp[lane] = 0;
  int pstep = 0;
  unsigned long long pst = 0;
#endif
  for (;;) {      // one iteration = one tile
    unsigned long long pt = G256P_T();
#ifdef XML_DEBUG_VARIANTS
    pstep = 0; pst = pt;
#endif
    f32x4 acc[4][8];
    // first fragments of the tile (read here rather than prefetched by the previous tile's last step: 32 registers that
    // would otherwise stay live through the epilogue, next to 128 accumulators and a row of results)
    {
      const char* slot0 = smem + c_slot * SLOT_BYTES;
#pragma unroll
      for (int m = 0; m < 4; ++m) faA[m] = *reinterpret_cast<const uint4*>(slot0 + a_off + m * 16 * ROWB);
#pragma unroll
      for (int n = 0; n < 4; ++n) fbL[n] = b_frag(slot0, n);
    }
    auto slice_step = [&](uint4 (&fc)[4], uint4 (&fn)[4], auto init_tag) {
      constexpr bool INIT = decltype(init_tag)::value;
      const char* slot = smem + c_slot * SLOT_BYTES;
#pragma unroll
      for (int n = 0; n < 4; ++n) fbH[n] = b_frag(slot, n + 4);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if constexpr (DIRECT) {
            if constexpr (INIT) G256pInit<T>::chunk(acc[m][n], fbL[n], fc[m]);
            else Mma<T>::chunk(acc[m][n], fbL[n], fc[m]);
          } else {
            if constexpr (INIT) G256pInit<T>::chunk(acc[m][n], fc[m], fbL[n]);
            else Mma<T>::chunk(acc[m][n], fc[m], fbL[n]);
          }
        }
      if (!GRP1) __builtin_amdgcn_s_waitcnt(0x007c);      // vmcnt(12) lgkmcnt(0): 6 pieces per slice, two slices young
      else __builtin_amdgcn_s_waitcnt(0x0074);            // vmcnt(4)
      __builtin_amdgcn_s_barrier();
      if (++c_slot == NSLOT) c_slot = 0;
      auto next_reads = [&]() {
        const char* nslot = smem + c_slot * SLOT_BYTES;
#pragma unroll
        for (int m = 0; m < 4; ++m) fn[m] = *reinterpret_cast<const uint4*>(nslot + a_off + m * 16 * ROWB);
#pragma unroll
        for (int n = 0; n < 4; ++n) fbL[n] = b_frag(nslot, n);
      };
      if (!GRP1) { next_reads(); issue_slice(); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if constexpr (DIRECT) {
            if constexpr (INIT) G256pInit<T>::chunk(acc[m][n + 4], fbH[n], fc[m]);
            else Mma<T>::chunk(acc[m][n + 4], fbH[n], fc[m]);
          } else {
            if constexpr (INIT) G256pInit<T>::chunk(acc[m][n + 4], fc[m], fbH[n]);
            else Mma<T>::chunk(acc[m][n + 4], fc[m], fbH[n]);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
      if (GRP1) { next_reads(); issue_slice(); }
#ifdef XML_DEBUG_VARIANTS
      if (a.probe & 1) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0 && pstep < 64) sp[pstep] += (unsigned int)(t - pst);
        pst = t; ++pstep;
      }
#endif
    };
    slice_step(faA, faB, std::true_type{});
    slice_step(faB, faA, std::false_type{});
    for (int s2 = 2; s2 < n_slices; s2 += 2) {
      slice_step(faA, faB, std::false_type{});
      slice_step(faB, faA, std::false_type{});
    }

    { const unsigned long long t = G256P_T(); pr[0] += t - pt; pt = t; }
    {
      int64_t mt; int nt;
      tile_mt_nt(c_k, mt, nt);
      const int64_t m0 = mt * 256;
      const int n0 = nt * 256;
      const uint32_t m0_mod = a.add_mode == 1 ? (uint32_t)(m0 % a.seq_len) : 0u;     // one 64-bit modulo per tile
      const int lane_e = g256p_lane_now();
      const int fr_e = lane_e & 15, fg_e = lane_e >> 4;
      const int tid_e = wave * 64 + lane_e;
      float* patch = reinterpret_cast<float*>(smem + RING_BYTES) + wave * 1024;
      // a lane owns two groups of 4 consecutive columns of a row.  bf16 out: adjacent groups -> one 16-byte store, 16
      // lanes = the row's 256 bytes.  f32 out: groups 64 columns apart -> two 16-byte stores, each again 16 lanes =
      // 256 contiguous bytes (adjacent groups would leave every 128-byte line half-written per instruction: twice the
      // write requests, and the CU's write path retires about one request per 5 cycles -- it is what bounds this epilogue)
      constexpr bool F32O = sizeof(OutT) == 4;
      const int orow = lane_e >> 4;
      const int oc0 = (lane_e & 15) * (F32O ? 4 : 8);
      const int oc1 = F32O ? oc0 + 64 : oc0 + 4;
      const int nc0 = n0 + wn * 128 + oc0, nc1 = n0 + wn * 128 + oc1;
      const bool ok0 = nc0 + 4 <= N, ok1 = nc1 + 4 <= N;           // N % 8 == 0 (eligibility): whole groups only
      // DIRECT: group q of this lane = GC consecutive columns of ONE row (row = fr of row block m):
      //   2-byte out: q = 0..3, columns q * 32 + 8 fg .. + 7  (accumulators 2 q, 2 q + 1);  4-byte out: q = 0..7, q * 16 + 4 fg .. + 3
      // A store instruction writes 16 rows x 64 contiguous bytes.
      constexpr int GC = PAIRED ? 8 : 4;
      constexpr int NGRP = 128 / (4 * GC);
      static_assert(GC * sizeof(AddT) <= 16, "addend group wider than one 16-byte load");
      int gcol[DIRECT ? NGRP : 1];
      bool gok[DIRECT ? NGRP : 1];
      int pcol[DIRECT ? NGRP / 2 : 1];      // column this lane STORES for group pair pq (see the stores below)
      // Stores in FULL 128-byte lines.  Written as they lie, groups (q, fg = 0..3) of a row are 64 contiguous bytes: every
      // store instruction touches 16 rows x half a line, 2048 line requests per tile -- and the CU's write path retires
      // one request per 5-8 cycles (the phase probe: 17 K cycles to drain a tile, a third of its time; vmcnt counts the
      // stores, so the K loop's counted waits block behind them).  Groups 2 p and 2 p + 1 of a row are the two halves of
      // one line: the EVEN store of a pair writes rows fr & ~1 -- even lanes their own group 2 p, odd lanes group 2 p + 1
      // of the row BELOW them (DPP row_shr:1) --, the ODD store rows fr | 1 likewise (row_shl:1): 8 rows x one whole
      // line per instruction, 1024 requests per tile, no LDS.
      auto store_rows = [&](int m4, auto& v, int dcol = 0) {        // (DIRECT only; v: float [NGRP][GC]; dcol: another tile's columns)
        uint4 pk[NGRP];
#pragma unroll
        for (int q = 0; q < NGRP; ++q) pk[q] = pack16<OutT>(v[q]);
        const bool odd = (fr_e & 1) != 0;
        const int64_t m_even = m0 + wm * 64 + m4 * 16 + (fr_e & ~1);
#pragma unroll
        for (int pq = 0; pq < NGRP / 2; ++pq) {
          const uint4 dn = dpp_u4<0x111>(pk[2 * pq + 1]);       // row_shr:1: lane i <- lane i - 1
          const uint4 up = dpp_u4<0x101>(pk[2 * pq]);           // row_shl:1: lane i <- lane i + 1
          // (component-wise selects: `odd ? a : b` on the struct type becomes a select of stack ADDRESSES + a scratch load)
          const uint4 own_e = pk[2 * pq], own_o = pk[2 * pq + 1];
          const uint4 d_even = make_uint4(odd ? dn.x : own_e.x, odd ? dn.y : own_e.y, odd ? dn.z : own_e.z, odd ? dn.w : own_e.w);
          const uint4 d_odd = make_uint4(odd ? own_o.x : up.x, odd ? own_o.y : up.y, odd ? own_o.z : up.z, odd ? own_o.w : up.w);
          const int col = pcol[pq] + dcol;
          if (col + GC <= N) {
            // non-temporal: this kernel only runs on outputs of >= 3072 tiles (400 MB and up), nothing of which survives
            // in a cache until its consumer starts; streaming stores retire ~2.5 % faster here (886 -> 906 TF at
            // N = 2304, same box).  LNE too: its outputs are final -- written once, after the row statistics, never re-read
            // by this kernel (rounds 2-5 re-read the tile in a second pass through this XCD's L2 and kept plain stores).
            if (!(a.probe & 8)) {      // (probe bit 3: plain stores, A/B)
              typedef unsigned int g_u4 __attribute__((ext_vector_type(4)));
              if (m_even < M && !(a.probe & 4)) __builtin_nontemporal_store(g_u4{d_even.x, d_even.y, d_even.z, d_even.w}, reinterpret_cast<g_u4*>(out + m_even * N + col));
              if (m_even + 1 < M && !(a.probe & 4)) __builtin_nontemporal_store(g_u4{d_odd.x, d_odd.y, d_odd.z, d_odd.w}, reinterpret_cast<g_u4*>(out + (m_even + 1) * N + col));
            } else {
              if (m_even < M && !(a.probe & 4)) st_global16(out + m_even * N + col, d_even);        // probe bit 2: no stores
              if (m_even + 1 < M && !(a.probe & 4)) st_global16(out + (m_even + 1) * N + col, d_odd);
            }
          }
        }
      };
      if constexpr (DIRECT) {
#pragma unroll
        for (int q = 0; q < NGRP; ++q) {
          gcol[q] = n0 + wn * 128 + q * (4 * GC) + fg_e * GC;
          gok[q] = gcol[q] + GC <= N;                              // N % 8 == 0 (eligibility): whole groups only
        }
#pragma unroll
        for (int pq = 0; pq < NGRP / 2; ++pq) pcol[pq] = n0 + wn * 128 + pq * (8 * GC) + (fr_e & 1) * (4 * GC) + fg_e * GC;
        // bias of this wave's 128 columns -> its LDS patch (floats 256..383), read back per row block below: held in
        // registers for the whole epilogue it is 32 VGPRs on top of 128 accumulators + a row of results + its addend.
        // (hipcc cannot count the inline-asm DMAs and guards the use of a loaded register with s_waitcnt vmcnt(0): one such
        // wait per tile, here)
        {
          float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
          const int bc = n0 + wn * 128 + (lane_e & 31) * 4;
          if (a.bias && bc + 4 <= N) b4 = *reinterpret_cast<const float4*>(a.bias + bc);
          if (lane_e < 32) *reinterpret_cast<float4*>(patch + 256 + lane_e * 4) = b4;
          __builtin_amdgcn_s_waitcnt(0xc07f);
          __builtin_amdgcn_wave_barrier();
        }
        { const unsigned long long t = G256P_T(); pr[1] += t - pt; pt = t; }
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) {
          const int lrow = wm * 64 + m4 * 16 + fr_e;
          const int64_t m = m0 + lrow;
          const bool rok = m < M;
          float v[NGRP][GC];
          int boff = fg_e * GC;                                       // opaque per row block: keeps the bias reads from being
          asm volatile("" : "+v"(boff));                              // hoisted (and held in 32 registers) across the loop
#pragma unroll
          for (int q = 0; q < NGRP; ++q) {
            float gb[GC];
#pragma unroll
            for (int e = 0; e < GC; e += 4) {
              const float4 b4 = *reinterpret_cast<const float4*>(patch + 256 + q * (4 * GC) + boff + e);
              gb[e] = b4.x; gb[e + 1] = b4.y; gb[e + 2] = b4.z; gb[e + 3] = b4.w;
            }
#pragma unroll
            for (int e = 0; e < GC; ++e) {
              float x = (PAIRED ? acc[m4][2 * q + (e >> 2)][e & 3] : acc[m4][q][e & 3]) + gb[e];
              if (a.relu) x = fmaxf(x, 0.f);
              v[q][e] = x;
            }
          }
          if (a.add_mode) {
            const int64_t arow = a.add_mode == 1 ? (int64_t)((m0_mod + (uint32_t)lrow) % (uint32_t)a.seq_len) : m;
            constexpr int AB = GC * (int)sizeof(AddT);            // addend bytes per group: 16 or 8
#pragma unroll
            for (int qb = 0; qb < NGRP; qb += 4) {                   // the loads of four groups first, then the arithmetic
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
                if constexpr (sizeof(AddT) == 2) unpack16<bf16_t>(ad[qi], av);   // (AB == 8: the upper four are zeros, unused)
                else unpack16<float>(ad[qi], av);
#pragma unroll
                for (int e = 0; e < GC; ++e) v[qb + qi][e] += av[e];
              }
            }
          }
          if constexpr (LNE) {      // (N % 256 == 0: every group is whole)
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < NGRP; ++q)
#pragma unroll
              for (int e = 0; e < GC; ++e) s += v[q][e];
            s += __shfl_xor(s, 16, 64);                               // the 4 lanes (fg) that share the row: this wave's 128 columns
            s += __shfl_xor(s, 32, 64);
            const float seg_mean = s * (1.0f / 128.0f);
            float c2 = 0.f;
#pragma unroll
            for (int q = 0; q < NGRP; ++q)
#pragma unroll
              for (int e = 0; e < GC; ++e) { const float c = v[q][e] - seg_mean; c2 += c * c; }
            c2 += __shfl_xor(c2, 16, 64);
            c2 += __shfl_xor(c2, 32, 64);
            // this wave's patch, floats 640 + (row of its 64, column tile) x 2: the segment (tile nt, half wn) of the row
            if (fg_e == 0) {
              float* ps = patch + 640 + ((m4 * 16 + fr_e) * tn_s + nt) * 2;
              ps[0] = s;
              ps[1] = c2;
            }
          }
          if constexpr (LNE) {
            if (nt == tn_s - 1) {
              // the last tile's pre-LayerNorm values stay in the accumulators (f32) until the row statistics are complete
#pragma unroll
              for (int q = 0; q < NGRP; ++q)
#pragma unroll
                for (int e = 0; e < GC; ++e) {
                  if constexpr (PAIRED) acc[m4][2 * q + (e >> 2)][e & 3] = v[q][e];
                  else acc[m4][q][e & 3] = v[q][e];
                }
            } else {
              // earlier tiles: f32 to the workgroup's scratch, lane for lane (every store instruction writes 1 KiB contiguous)
              float4* sc = reinterpret_cast<float4*>(a.ln_scratch) + (int64_t)blockIdx.x * (tn_s - 1) * 16384 +
                           (int64_t)((nt * 4 + m4) * 8) * 512 + tid_e;
              // (non-temporal: 512 KB per workgroup and row block, read back once a tile or two later -- 16 MB per XCD would
              // sweep the A / W tiles out of its 4 MB L2; the Infinity Cache holds them)
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const float* vf = &v[0][0] + j * 4;
                typedef float xml_nt4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(xml_nt4{vf[0], vf[1], vf[2], vf[3]}, reinterpret_cast<xml_nt4*>(sc + j * 512));
              }
            }
          } else {
            store_rows(m4, v);
          }
        }
      } else {
      float bv[8];
      {
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        if (a.bias && ok0) b0 = *reinterpret_cast<const float4*>(a.bias + nc0);
        if (a.bias && ok1) b1 = *reinterpret_cast<const float4*>(a.bias + nc1);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
      }
      // The DMA stream is inline asm, so hipcc cannot count outstanding VMEM operations and guards every use of a loaded
      // register with s_waitcnt vmcnt(0) -- which, inside the passes below, also waits for the stores of the previous
      // pass.  Passing the values through an empty asm here makes that one wait happen now, once per tile.
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(bv[j]));
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        if ((fg_e >> 1) == (p & 1)) {
          const int prow0 = (fg_e & 1) * 4;
#pragma unroll
          for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              patch[(prow0 + r) * 128 + ((n * 16 + fr_e) ^ ((fg_e & 1) << 4))] = acc[p >> 1][n][r];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int prow = it * 4 + orow;               // 0..7
          const int lrow = wm * 64 + p * 8 + prow;
          const int64_t m = m0 + lrow;
          const float4 v0 = *reinterpret_cast<const float4*>(patch + prow * 128 + (oc0 ^ (it << 4)));
          const float4 v1 = *reinterpret_cast<const float4*>(patch + prow * 128 + (oc1 ^ (it << 4)));
          float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          if (m >= M) continue;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[j] += bv[j];
            if (a.relu) v[j] = fmaxf(v[j], 0.f);
          }
          if (a.add_mode) {
            const int64_t arow = a.add_mode == 1 ? (int64_t)((m0_mod + (uint32_t)lrow) % (uint32_t)a.seq_len) : m;
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
          } else if (ok1) {                               // (ok1 implies ok0; N % 8 == 0)
            st_global16(out + m * N + nc0, pack16<bf16_t>(v));
          }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
      }
      }   // !DIRECT
      { const unsigned long long t = G256P_T(); pr[2] += t - pt; pt = t; }
      if constexpr (LNE) {
        if (nt == tn_s - 1) {
          // ---- the row block is complete: statistics from the 2 tn segment partials, normalise, store -------------------
          __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): my partials are in LDS
          __builtin_amdgcn_s_barrier();                // ... and so are the partner wave's (the other 128-column half)
          {   // lane l: statistics of row wm * 64 + l, combined in a fixed order (tile 0 left, tile 0 right, tile 1 left, ...)
            const float* p0 = reinterpret_cast<const float*>(smem + RING_BYTES) + (wm * 2) * 1024 + 640 + lane_e * tn_s * 2;
            const float* p1 = p0 + 1024;
            typedef float xml_f4 __attribute__((ext_vector_type(4)));
            xml_f4 pv[3];
#pragma unroll
            for (int t = 0; t < 3; ++t)
              pv[t] = t < tn_s ? xml_f4{p0[t * 2], p0[t * 2 + 1], p1[t * 2], p1[t * 2 + 1]} : xml_f4{0.f, 0.f, 0.f, 0.f};
            float tot = 0.f;
#pragma unroll
            for (int t = 0; t < 3; ++t) tot += pv[t].x + pv[t].z;
            tot += 0.f + 0.f;                          // (the fourth tile of the former four-tile form: the same additions)
            const float mean = tot / (float)N;
            float m2 = 0.f;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
              if (t < tn_s) {
                const float d0 = pv[t].x * (1.0f / 128.0f) - mean, d1 = pv[t].z * (1.0f / 128.0f) - mean;
                m2 += pv[t].y + 128.0f * d0 * d0 + pv[t].w + 128.0f * d1 * d1;
              }
            }
            const float rstd = 1.0f / sqrtf(m2 / (float)N + a.ln_eps);
            patch[lane_e * 2] = mean;
            patch[lane_e * 2 + 1] = rstd;
          }
          { const unsigned long long t = G256P_T(); pr[3] += t - pt; pt = t; }
          // tiles tn - 1 (from the accumulators), then tn - 2 .. 0 (from the scratch); gamma / beta of the tile's columns go
          // through the wave's LDS patch (floats 384..511 / 512..639): in registers they would be 64 VGPRs next to the 128
          // live accumulators
          for (int t = tn_s - 1; t >= 0; --t) {
            __builtin_amdgcn_wave_barrier();
            {
              const int bc = t * 256 + wn * 128 + (lane_e & 31) * 4;
              const float4 g4 = *reinterpret_cast<const float4*>(a.ln_g + bc);
              const float4 b4 = *reinterpret_cast<const float4*>(a.ln_b + bc);
              if (lane_e < 32) {
                *reinterpret_cast<float4*>(patch + 384 + lane_e * 4) = g4;
                *reinterpret_cast<float4*>(patch + 512 + lane_e * 4) = b4;
              }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const float4* sc = reinterpret_cast<const float4*>(a.ln_scratch) + (int64_t)blockIdx.x * (tn_s - 1) * 16384 +
                               (int64_t)(t * 32) * 512 + tid_e;
#pragma unroll
            for (int m4 = 0; m4 < 4; ++m4) {
              const int lr64 = m4 * 16 + fr_e;
              const float mean = patch[lr64 * 2], rstd = patch[lr64 * 2 + 1];
              int boff = fg_e * GC;
              asm volatile("" : "+v"(boff));
              float x[NGRP * GC];
              if (t == tn_s - 1) {
#pragma unroll
                for (int q = 0; q < NGRP; ++q)
#pragma unroll
                  for (int e = 0; e < GC; ++e)
                    x[q * GC + e] = PAIRED ? acc[m4][2 * q + (e >> 2)][e & 3] : acc[m4][q][e & 3];
              } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  typedef float xml_nt4 __attribute__((ext_vector_type(4)));
                  const xml_nt4 f = __builtin_nontemporal_load(reinterpret_cast<const xml_nt4*>(sc + (m4 * 8 + j) * 512));
                  x[j * 4] = f.x; x[j * 4 + 1] = f.y; x[j * 4 + 2] = f.z; x[j * 4 + 3] = f.w;
                }
              }
              float v[NGRP][GC];
#pragma unroll
              for (int q = 0; q < NGRP; ++q)
#pragma unroll
                for (int e = 0; e < GC; e += 4) {
                  const float4 g4 = *reinterpret_cast<const float4*>(patch + 384 + q * (4 * GC) + boff + e);
                  const float4 b4 = *reinterpret_cast<const float4*>(patch + 512 + q * (4 * GC) + boff + e);
                  const float gq[4] = {g4.x, g4.y, g4.z, g4.w}, bq[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                  for (int j = 0; j < 4; ++j) v[q][e + j] = (x[q * GC + e + j] - mean) * rstd * gq[j] + bq[j];
                }
              store_rows(m4, v, (t - nt) * 256);
            }
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    { const unsigned long long t = G256P_T(); pr[4] += t - pt; pt = t; }
    ++pr[6];
    ++c_k;
    if (tile_of(c_k) >= a.n_tiles) break;
  }
#ifdef XML_DEBUG_VARIANTS
  if ((a.probe & 1) && blockIdx.x == 0 && lane == 0) {
    pr[5] = __builtin_amdgcn_s_memtime() - pr_t0;
    pr[7] = __builtin_amdgcn_s_memrealtime() - pr_r0;
    for (int i = 0; i < 8; ++i) g_g256p_probe[wave * 8 + i] = pr[i];
    for (int i = 0; i < 64; ++i) g_g256p_steps[wave * 64 + i] = sp[i];
  }
#endif
  };
  if (grp) run(std::true_type{});
  else run(std::false_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the saturated stream still has DMAs in flight
  __builtin_amdgcn_s_barrier();
}

template <typename T, typename OutT, typename AddT, bool LNE = false, bool DIRECT = true>
static int launch_gemm256p(const void* A, const void* W, const float* bias, const void* addend, void* out, int64_t M,
                           int N, int K, int relu, int add_mode, int seq_len, hipStream_t st, const float* ln_g = nullptr,
                           const float* ln_b = nullptr, void* ln_ws = nullptr) {
  G256pArgs a;
  a.A = A; a.W = W; a.bias = bias; a.addend = addend; a.out = out;
  a.M = M; a.N = N; a.K = K; a.relu = relu; a.add_mode = add_mode; a.seq_len = seq_len;
  a.tn = cdiv(N, 256);
  a.n_tiles = (int64_t)cdiv(M, 256) * a.tn;
  a.ln_scratch = (float*)ln_ws; a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = 1e-5f;
  a.probe = g_q2c_ablation == 9 ? 1 : g_q2c_ablation == 21 ? 2 : g_q2c_ablation == 22 ? 4 : g_q2c_ablation == 23 ? 6 : g_q2c_ablation == 24 ? 8 : 0;
  const int lds = 4 * 2 * 256 * 64 + 8 * 4096;          // ring + patches = 160 KiB
#ifdef XML_DEBUG_VARIANTS
  if (!LNE && DIRECT && g_gemm_variant == 3)      // A/B: the LDS-staged epilogue
    return launch_gemm256p<T, OutT, AddT, false, false>(A, W, bias, addend, out, M, N, K, relu, add_mode, seq_len, st);
#endif
  auto kern = gemm256p_kernel<T, OutT, AddT, LNE, DIRECT>;
  if (!xml_lds_attr_once<gemm256p_kernel<T, OutT, AddT, LNE, DIRECT>>(lds)) return XML_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, st, a);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---- GEMM with the LayerNorm in its epilogue --------------------------------------------------------------------
// Taken for a SHAPE CLASS, never for a batch size above some tile count: rows that span 1-3 whole 256-column tiles
// (N = 256 / 512 / 768), K a whole number of 128-byte steps, and at least LN_FUSED_MIN_ROWS rows.  Whether a projection
// runs fused therefore does not depend on how many videos a context batch holds (any batch of >= 16 videos of 128 clips
// qualifies): the encoder's bits are the same for context batches of 2 048, 200 or 37 videos
// (tests/test_gpu_model.py::test_index_bits_do_not_depend_on_the_context_batch).  Below the row threshold (a 50-query batch's
// 1 500 tokens) the callers run GEMM (f32 out) + LayerNorm: small tiles fill the chip there, six row blocks would not.
// The fused kernel waits for nothing outside its workgroup, so it has no residency requirement and no failure path.
static constexpr int64_t LN_FUSED_MIN_ROWS = 2048;

bool xmli_gemm_ln_eligible(int64_t M, int N, int K, int dt) {
  if (dt != XML_F32 && dt != XML_BF16) return false;      // (split-f16 projections take the three-launch path)
  const size_t kb = (size_t)K * dt_size(dt);
  return kb % 128 == 0 && kb >= 256 && N % 256 == 0 && N / 256 <= 3 && M >= LN_FUSED_MIN_ROWS;
}
// scratch of the fused kernel: per workgroup the f32 values of a row block's first tn - 1 tiles
size_t xmli_gemm_ln_workspace_bytes(int64_t M, int N) {
  (void)M;
  return align_up((size_t)256 * (size_t)(N / 256 > 1 ? N / 256 - 1 : 0) * 256 * 256 * 4 + 256, 256);
}
int xmli_gemm_ln(const void* A, const void* W, const float* bias, const void* addend, const float* ln_g, const float* ln_b,
                 void* y, int64_t M, int N, int K, int relu, int add_mode, int seq_len, int dt, void* ln_ws,
                 hipStream_t st) {
  if (!ln_ws) return XML_ERR_WORKSPACE;
  if (dt == XML_F32)
    return launch_gemm256p<float, float, float, true>(A, W, bias, addend, y, M, N, K, relu, add_mode, seq_len, st, ln_g,
                                                      ln_b, ln_ws);
  return launch_gemm256p<bf16_t, bf16_t, bf16_t, true>(A, W, bias, addend, y, M, N, K, relu, add_mode, seq_len, st, ln_g,
                                                        ln_b, ln_ws);
}

// worth it when every workgroup gets several tiles; below that the one-tile-per-workgroup kernel is as good
bool xmli_gemm256p_eligible(int64_t M, int N, int K, int dt) {
  const size_t kb = (size_t)K * dt_size(dt);
  return kb % 128 == 0 && kb >= 256 && N >= 128 && N % 8 == 0 && (int64_t)cdiv(M, 256) * cdiv(N, 256) >= 3072;   // (2304 tiles: 2 % behind the one-tile kernel, 3516: 7 % ahead)
}

int xmli_gemm256p(const void* A, const void* W, const float* bias, const void* addend, void* out, int64_t M, int N,
                  int K, int relu, int add_mode, int seq_len, int out_f32, int dt, hipStream_t st) {
  if (dt == XML_F32)
    return launch_gemm256p<float, float, float>(A, W, bias, addend, out, M, N, K, relu, add_mode, seq_len, st);
  if (out_f32)
    return launch_gemm256p<bf16_t, float, bf16_t>(A, W, bias, addend, out, M, N, K, relu, add_mode, seq_len, st);
  return launch_gemm256p<bf16_t, bf16_t, bf16_t>(A, W, bias, addend, out, M, N, K, relu, add_mode, seq_len, st);
}
