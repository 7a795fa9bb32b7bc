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
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "common.h"
#include "internal.h"

// LayerNorm-epilogue exchange: per-row partials published and read with agent-scope accesses (on gfx950 the sc1 form of
// the instruction: past the per-XCD L2, coherent across the 8 XCDs of the device)
__device__ __forceinline__ void ln_publish(float* pp, float s, float m2) {
  __hip_atomic_store(pp, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(pp + 1, m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
typedef float xml_ln_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ xml_ln_f4 ln_read4(const float* pp) {
  xml_ln_f4 v;
  v.x = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.y = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.z = __hip_atomic_load(pp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.w = __hip_atomic_load(pp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}

// DIAGNOSTIC count of row-block waits of the LayerNorm-epilogue kernel that gave up (see the exchange below); read and cleared
// by xml_ln_fusion_status.  No kernel reads it: what makes the remaining waits of a launch give up at once is that LAUNCH's own
// flag in its workspace (G256pArgs::ln_fail, zeroed per launch), so a count left behind by an earlier launch cannot touch a
// later one.
__device__ int g_ln_timeouts = 0;

struct G256pArgs {
  const void* A;
  const void* W;
  const float* bias;
  const void* addend;
  void* out;
  int64_t M, n_tiles;
  int N, K, relu, add_mode, seq_len, tn;
  // LayerNorm epilogue (LNE kernels): y = LN(act(A W^T + bias) + addend) * g + b over the full rows of N = tn * 256 columns
  float* ln_part;       // (M, 2 tn, 2) f32: per row and per 128-column segment (sum, centred sum of squares)
  int* ln_count;        // (ceil(M / 256)): column tiles of a row block that have published their partials (zeroed per launch)
  int* ln_fail;         // 1 int behind ln_count, zeroed per launch: some wait of THIS launch gave up -> the others do not wait
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
__device__ __forceinline__ int g256p_swz(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

// LNE: LayerNorm fused into the epilogue (K2 behind K1, BertSelfOutput; xml/model_components.py:76-89,313-317).  A row of
// the output spans tn = N / 256 column tiles = tn workgroups.  Every workgroup (1) writes its part of the pre-LayerNorm
// rows in the storage type and publishes, per row and 128-column segment, (sum, centred sum of squares) computed from
// the f32 values; (2) signals a per-row-block counter and waits until the tn tiles of the block have signalled -- all
// 256 workgroups are resident (one per CU) and walk the tiles in the same order, so the partners are at most a tile
// apart; (3) combines the 2 tn partials of its rows in a fixed order (Chan's formula: deterministic, no E[x^2] - mean^2
// cancellation) and normalises its own part in place.  No f32 round trip through HBM, no separate LayerNorm launch.
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
  // LNE: the tn column tiles of a row block must meet (row statistics), so they go to tn NEIGHBOURING workgroups of ONE
  // XCD in the same round -- one L2 serves their exchange, no device-wide cache maintenance -- : an XCD takes 32 / tn row
  // blocks per round (tn = 3: 10 blocks, 2 of its 32 workgroups idle).
  const int lne_g = LNE ? __builtin_amdgcn_readfirstlane(32 / a.tn) : 0;
  const int lne_rb = LNE ? __builtin_amdgcn_readfirstlane(loc / a.tn) : 0, lne_nt = LNE ? loc - lne_rb * a.tn : 0;
  auto tile_of = [&](int k) -> int64_t {
    if constexpr (LNE) {
      if (lne_rb >= lne_g) return a.n_tiles;
      return (((int64_t)k * 8 + xcd) * lne_g + lne_rb) * a.tn + lne_nt;
    }
    return ((int64_t)k * 8 + xcd) * 32 + loc;
  };
  auto tile_mt_nt = [&](int k, int64_t& mt, int& nt) {      // (row block, column tile) of this workgroup's k-th tile
    if constexpr (LNE) {
      mt = ((int64_t)k * 8 + xcd) * lne_g + lne_rb;         // scalar arithmetic on uniform ints: the DMA bases stay in SGPRs
      nt = lne_nt;
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
    int lane_o = lane;                             // opaque copy: keeps the address arithmetic out of the MFMA loop
    asm volatile("" : "+v"(lane_o));
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
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int fr_e = lane_e & 15, fg_e = lane_e >> 4;
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
      auto store_rows = [&](int m4, auto& v) {        // (DIRECT only; v: float [NGRP][GC])
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
          const int col = pcol[pq];
          if (col + GC <= N) {
            // non-temporal: this kernel only runs on outputs of >= 3072 tiles (400 MB and up), nothing of which survives
            // in a cache until its consumer starts; streaming stores retire ~2.5 % faster here (886 -> 906 TF at
            // N = 2304, same box).  Not for LNE: pass 2 re-reads the tile through this XCD's L2.
            if (!LNE && !(a.probe & 8)) {      // (probe bit 3: plain stores, A/B)
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
            if (fg_e == 0 && rok) {
              float* pp = a.ln_part + ((m * (2 * a.tn)) + nt * 2 + wn) * 2;
              ln_publish(pp, s, c2);
            }
          }
          if constexpr (LNE) {
            // the pre-LayerNorm values stay in the accumulators (f32) until the row statistics are complete: no store of the
            // pre-LN rows, no re-read after the exchange, and the normalisation sees unrounded values
#pragma unroll
            for (int q = 0; q < NGRP; ++q)
#pragma unroll
              for (int e = 0; e < GC; ++e) {
                if constexpr (PAIRED) acc[m4][2 * q + (e >> 2)][e & 3] = v[q][e];
                else acc[m4][q][e & 3] = v[q][e];
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
          if constexpr (LNE) {      // (N % 256 == 0: every lane of the row holds 8 real columns)
            float s8 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s8 += v[j];
            const float seg_sum = lane16_sum_dpp(s8);                 // the 16 lanes of a row: this wave's 128 columns
            const float seg_mean = seg_sum * (1.0f / 128.0f);
            float q8 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float c = v[j] - seg_mean; q8 += c * c; }
            const float seg_m2 = lane16_sum_dpp(q8);
            if ((lane_e & 15) == 0) {
              float* pp = a.ln_part + ((m * (2 * a.tn)) + nt * 2 + wn) * 2;
              ln_publish(pp, seg_sum, seg_m2);
            }
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
        // ---- publish, wait for the row block's other column tiles, normalise in place -------------------------------
        // The exchange does NOT depend on which XCD a partner runs on: the partials are published and read with
        // agent-scope (device-coherent) accesses (ln_publish / ln_read4: they go past the per-XCD L2 to the memory side, a
        // few hundred bytes per tile), the counter is an agent-scope atomic, and "my stores are acknowledged" (vmcnt(0))
        // orders the two.  The tile walk above still puts partners on ONE XCD in the same round -- that is speed (they finish
        // together), not correctness.  A device-scope FENCE here would write back and invalidate the whole L2 per tile --
        // measured: it doubled the kernel's time; coherent accesses to the partials alone cost nothing measurable.
        // The wait is bounded: if the partners cannot get onto the chip (CUs held by another process that is itself waiting,
        // a CU mask smaller than the grid), the kernel traps after ~4 s instead of hanging the device or normalising with
        // missing statistics -- the runtime reports the fault to the caller.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // wave 0's patch, float 640: a word nothing else uses between here and the end of the tile (0..127 statistics,
        // 256..639 bias / gamma / beta)
        volatile float* ln_fail_lds = reinterpret_cast<volatile float*>(smem + RING_BYTES) + 640;
        if (tid == 0) {
          __hip_atomic_fetch_add(a.ln_count + mt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();        // 100 MHz
          float failed = 0.f;
          while (__hip_atomic_load(a.ln_count + mt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.tn) {
            __builtin_amdgcn_s_sleep(1);
            // bounded wait, no trap (a trap is a sticky device fault: the process loses its context).  A wait that gives up
            // marks THIS launch (ln_fail: every later wait of the launch gives up at once instead of 4 s each), bumps the
            // diagnostic counter the host reads at its next synchronisation point (xml_ln_fusion_status), and the tile is
            // written as NaN -- never as a LayerNorm over missing statistics: a caller that does not look at the counter
            // sees NaN scores, not plausible wrong ones.
            if (__hip_atomic_load(a.ln_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                __builtin_amdgcn_s_memrealtime() - t_start > 400000000ull) {
              __hip_atomic_fetch_add(a.ln_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_fetch_add(&g_ln_timeouts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              failed = 1.f;
              break;
            }
          }
          *ln_fail_lds = failed;
          __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the word is in LDS before this wave reaches the barrier
        }
        __builtin_amdgcn_s_barrier();
        const bool ln_failed = *ln_fail_lds != 0.f;
        {   // lane l: statistics of row wm * 64 + l of the tile, combined from the 2 tn segment partials in fixed order
          const int64_t m = m0 + wm * 64 + lane_e;
          float mean = 0.f, rstd = 0.f;
          if (m < M) {
            // all partials of the row in one batch of 16-byte loads (tn <= 4: at most 16 floats), then the arithmetic --
            // a load -> use loop would pay one memory round trip per iteration (hipcc waits vmcnt(0) at every use here)
            typedef float xml_f4 __attribute__((ext_vector_type(4)));
            const float* pp = a.ln_part + m * (2 * a.tn) * 2;
            xml_f4 pv[4];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) pv[t4] = t4 < a.tn ? ln_read4(pp + 4 * t4) : xml_f4{0.f, 0.f, 0.f, 0.f};
            float tot = 0.f;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) tot += pv[t4].x + pv[t4].z;
            mean = tot / (float)N;
            float m2 = 0.f;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
              if (t4 < a.tn) {
                const float d0 = pv[t4].x * (1.0f / 128.0f) - mean, d1 = pv[t4].z * (1.0f / 128.0f) - mean;
                m2 += pv[t4].y + 128.0f * d0 * d0 + pv[t4].w + 128.0f * d1 * d1;
              }
            }
            rstd = 1.0f / sqrtf(m2 / (float)N + a.ln_eps);
            if (ln_failed) mean = rstd = __builtin_nanf("");
          }
          patch[lane_e * 2] = mean;
          patch[lane_e * 2 + 1] = rstd;
        }
        { const unsigned long long t = G256P_T(); pr[3] += t - pt; pt = t; }
        if constexpr (DIRECT) {
          // gamma / beta of this wave's 128 columns -> its LDS patch (floats 384..511 / 512..639), read per row block like the
          // bias (held in registers they would be 64 VGPRs next to the 128 live accumulators)
          {
            const int bc = n0 + wn * 128 + (lane_e & 31) * 4;
            float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = g4;
            if (bc + 4 <= N) {
              g4 = *reinterpret_cast<const float4*>(a.ln_g + bc);
              b4 = *reinterpret_cast<const float4*>(a.ln_b + bc);
            }
            if (lane_e < 32) {
              *reinterpret_cast<float4*>(patch + 384 + lane_e * 4) = g4;
              *reinterpret_cast<float4*>(patch + 512 + lane_e * 4) = b4;
            }
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);
          __builtin_amdgcn_wave_barrier();
          // the pre-LayerNorm values are still in the accumulators (f32): normalise and store, once
#pragma unroll
          for (int m4 = 0; m4 < 4; ++m4) {
            const int lr64 = m4 * 16 + fr_e;
            const float mean = patch[lr64 * 2], rstd = patch[lr64 * 2 + 1];
            int boff = fg_e * GC;
            asm volatile("" : "+v"(boff));
            float v[NGRP][GC];
#pragma unroll
            for (int q = 0; q < NGRP; ++q)
#pragma unroll
              for (int e = 0; e < GC; e += 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(patch + 384 + q * (4 * GC) + boff + e);
                const float4 b4 = *reinterpret_cast<const float4*>(patch + 512 + q * (4 * GC) + boff + e);
                const float gq[4] = {g4.x, g4.y, g4.z, g4.w}, bq[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const float x = PAIRED ? acc[m4][2 * q + ((e + j) >> 2)][(e + j) & 3] : acc[m4][q][(e + j) & 3];
                  v[q][e + j] = (x - mean) * rstd * gq[j] + bq[j];
                }
              }
            store_rows(m4, v);
          }
        } else {
        float gv[8], bb[8];
        {
          const float4 g0 = *reinterpret_cast<const float4*>(a.ln_g + nc0), g1 = *reinterpret_cast<const float4*>(a.ln_g + nc1);
          const float4 b0 = *reinterpret_cast<const float4*>(a.ln_b + nc0), b1 = *reinterpret_cast<const float4*>(a.ln_b + nc1);
          gv[0] = g0.x; gv[1] = g0.y; gv[2] = g0.z; gv[3] = g0.w; gv[4] = g1.x; gv[5] = g1.y; gv[6] = g1.z; gv[7] = g1.w;
          bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // the same (row, column group) walk as above: a lane re-reads exactly what it wrote.  ALL loads of the tile first
        // (the accumulators are dead: 64 / 128 registers are free), one wait, then arithmetic and stores
        constexpr int NLD = F32O ? 2 : 1;
        uint4 rb[16][NLD];
#pragma unroll
        for (int pi = 0; pi < 16; ++pi) {
          const int lr64 = (pi >> 1) * 8 + (pi & 1) * 4 + orow;
          const int64_t m = m0 + wm * 64 + lr64;
          const int64_t ms = m < M ? m : m0;              // (rows beyond M: any valid row, the result is not stored)
          rb[pi][0] = ld_global16(out + ms * N + nc0);
          if (F32O) rb[pi][NLD - 1] = ld_global16(out + ms * N + nc1);
        }
#pragma unroll
        for (int pi = 0; pi < 16; ++pi) {
          const int lr64 = (pi >> 1) * 8 + (pi & 1) * 4 + orow;
          const int64_t m = m0 + wm * 64 + lr64;
          const float mean = patch[lr64 * 2], rstd = patch[lr64 * 2 + 1];
          float v[8];
          if (F32O) {
            unpack16<float>(rb[pi][0], v);
            unpack16<float>(rb[pi][NLD - 1], v + 4);
          } else {
            unpack16<OutT>(rb[pi][0], v);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (v[j] - mean) * rstd * gv[j] + bb[j];
          if (m < M) {
            if (F32O) {
              st_global16(out + m * N + nc0, pack16<float>(v));
              st_global16(out + m * N + nc1, pack16<float>(v + 4));
            } else {
              st_global16(out + m * N + nc0, pack16<bf16_t>(v));
            }
          }
        }
        }   // !DIRECT
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
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

__global__ void g256p_zero_kernel(int* p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0;
}

static bool ln_coop_wanted();

template <typename T, typename OutT, typename AddT, bool LNE = false, bool DIRECT = true>
static int launch_gemm256p(const void* A, const void* W, const float* bias, const void* addend, void* out, int64_t M,
                           int N, int K, int relu, int add_mode, int seq_len, hipStream_t st, const float* ln_g = nullptr,
                           const float* ln_b = nullptr, void* ln_ws = nullptr) {
  G256pArgs a;
  a.A = A; a.W = W; a.bias = bias; a.addend = addend; a.out = out;
  a.M = M; a.N = N; a.K = K; a.relu = relu; a.add_mode = add_mode; a.seq_len = seq_len;
  a.tn = cdiv(N, 256);
  a.n_tiles = (int64_t)cdiv(M, 256) * a.tn;
  a.ln_part = nullptr; a.ln_count = nullptr; a.ln_fail = nullptr; a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = 1e-5f;
  a.probe = g_q2c_ablation == 9 ? 1 : g_q2c_ablation == 21 ? 2 : g_q2c_ablation == 22 ? 4 : g_q2c_ablation == 23 ? 6 : g_q2c_ablation == 24 ? 8 : 0;
  if (LNE) {
    const int n_blocks = (int)cdiv(M, 256);
    a.ln_count = (int*)ln_ws;
    a.ln_fail = a.ln_count + n_blocks;
    a.ln_part = (float*)((char*)ln_ws + align_up((size_t)(n_blocks + 1) * 4, 256));
    hipLaunchKernelGGL(g256p_zero_kernel, dim3(cdiv(n_blocks + 1, 256)), dim3(256), 0, st, a.ln_count, n_blocks + 1);
  }
  const int lds = 4 * 2 * 256 * 64 + 8 * 4096;          // ring + patches = 160 KiB
#ifdef XML_DEBUG_VARIANTS
  if (DIRECT && g_gemm_variant == 3)      // A/B: the LDS-staged epilogue
    return launch_gemm256p<T, OutT, AddT, LNE, false>(A, W, bias, addend, out, M, N, K, relu, add_mode, seq_len, st, ln_g, ln_b,
                                                      ln_ws);
#endif
  auto kern = gemm256p_kernel<T, OutT, AddT, LNE, DIRECT>;
  if (!xml_lds_attr_once<gemm256p_kernel<T, OutT, AddT, LNE, DIRECT>>(lds)) return XML_ERR_LAUNCH;
  if (LNE) {
    // The workgroups of a row block wait for each other, so all 256 must get onto the chip.  With one workgroup per CU
    // (160 KiB of LDS) and a grid of 256 they do as soon as the CUs are free -- kernels of OTHER kinds that still hold CUs
    // only delay them.  Two kernels of THIS kind running at once could each hold part of the chip and wait for partners
    // that cannot start: launches on different streams are therefore chained through an event (per device).
    // (hipLaunchCooperativeKernel would give the same guarantee, but rocprofiler-sdk 7.2 crashes at process exit after
    // a traced cooperative launch -- `rocprofv3 --kernel-trace -- python bench.py` ended with SIGSEGV, outputs written.)
    static std::mutex mu;
    static hipEvent_t last_ev[16] = {};
    static hipStream_t last_st[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return XML_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(mu);
    if (!last_ev[dev] && hipEventCreateWithFlags(&last_ev[dev], hipEventDisableTiming) != hipSuccess) return XML_ERR_LAUNCH;
    else if (last_st[dev] != st && hipStreamWaitEvent(st, last_ev[dev], 0) != hipSuccess) return XML_ERR_LAUNCH;
    // Cooperative launch (the runtime refuses a grid that cannot be co-resident instead of letting it wait): the default
    // outside profiler runs and stream captures; XML_LN_COOP=0 / 1 forces it off / on.  A refused launch returns an error
    // here and the caller takes the three-launch path.
    bool coop = ln_coop_wanted();
    if (coop) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) coop = false;
      (void)hipGetLastError();
    }
    if (coop) {
      void* kargs[] = {(void*)&a};
      if (hipLaunchCooperativeKernel((const void*)kern, dim3(256), dim3(512), kargs, (unsigned)lds, st) != hipSuccess) {
        (void)hipGetLastError();
        return XML_ERR_LAUNCH;
      }
    } else {
      hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, st, a);
    }
    const bool ok = hipGetLastError() == hipSuccess && hipEventRecord(last_ev[dev], st) == hipSuccess;
    last_st[dev] = st;
    return ok ? XML_OK : XML_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, st, a);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---- GEMM with the LayerNorm in its epilogue --------------------------------------------------------------------
// Eligible when the rows span whole 256-column tiles (N % 256 == 0, at most 4 of them) and there is enough work for the
// persistent kernel; the callers fall back to GEMM (f32 out) + LayerNorm otherwise.
// The fused kernel's workgroups wait for each other: it runs only where all 256 of them fit at once -- a device with at
// least 256 CUs visible to this process (MI355X unpartitioned; a CPX / partitioned device or a CU mask reports fewer and
// takes the three-launch path).
static std::atomic<int> g_ln_fusion_off{0};     // set by xml_ln_fusion_status after a timed-out exchange, or XML_LN_FUSION=0

static bool ln_coop_wanted() {
  static const int mode = [] {
    // OPT-IN (XML_LN_COOP=1).  Measured in round 4: after ONE cooperative launch in a process, every later kernel of that
    // process's neighbours on the GPU -- and graph replays of the process itself -- ran 1.5-3x slower for the rest of the
    // run (bench.py's extras child next to its idle parent: K6 94.9 instead of 37.7 ms, training step 16.3 instead of
    // 5.05 ms; the queue a cooperative launch goes through stays gang-scheduled).  And rocprofiler-sdk 7.2 crashes at
    // process exit after a traced cooperative launch (round 2).  The default is therefore the plain launch + the bounded,
    // COUNTED wait (xml_ln_fusion_status).
    const char* e = getenv("XML_LN_COOP");
    return (e && *e && atoi(e)) ? 1 : 0;
  }();
  if (!mode) return false;
  int dev = 0, ok = 0;
  return hipGetDevice(&dev) == hipSuccess &&
         hipDeviceGetAttribute(&ok, hipDeviceAttributeCooperativeLaunch, dev) == hipSuccess && ok != 0;
}

// How many row-block exchanges of the LayerNorm-epilogue GEMM gave up since the last call (their tiles were written as NaN).
// Synchronises the device.  disable != 0: a non-zero count also switches the fused path off for this process -- every later
// projection takes GEMM + LayerNorm launches, which wait for nothing.  Returns the count, or a negative xml_status.
extern "C" int xml_ln_fusion_status(int disable) {
  XML_ENTER();
  int n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_ln_timeouts), sizeof(int)) != hipSuccess) return XML_ERR_LAUNCH;
  if (n != 0) {
    const int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ln_timeouts), &zero, sizeof(int)) != hipSuccess) return XML_ERR_LAUNCH;
    if (disable) g_ln_fusion_off.store(1);
  }
  return n;
}
extern "C" int xml_ln_fusion_enabled(void) {
  static const int env_off = [] { const char* e = getenv("XML_LN_FUSION"); return (e && *e && !atoi(e)) ? 1 : 0; }();
  return (env_off || g_ln_fusion_off.load()) ? 0 : 1;
}

static bool ln_fusion_device_ok() {
  if (!xml_ln_fusion_enabled()) return false;
  static std::atomic<int> cached[64];                  // 0 unknown, 1 ok, 2 not ok  (per device)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  int c = cached[dev].load(std::memory_order_relaxed);
  if (c == 0) {
    int cus = 0;
    const bool ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 256;
    c = ok ? 1 : 2;
    cached[dev].store(c, std::memory_order_relaxed);
  }
  return c == 1;
}

bool xmli_gemm_ln_eligible(int64_t M, int N, int K, int dt) {
  if (dt != XML_F32 && dt != XML_BF16) return false;      // (split-f16 projections take the three-launch path)
  const size_t kb = (size_t)K * dt_size(dt);
  if (!ln_fusion_device_ok()) return false;
  // N = 256 (the reference's as-trained hidden size, xml/config.py:143): a row is ONE tile, the statistics never leave the
  // workgroup, nothing waits for a partner -- worth it from one tile per workgroup on (the query encoder of TVR val: 745
  // tiles; it saves the f32 round trip and the LayerNorm launch behind each of its three projections)
  const int64_t tiles = (int64_t)cdiv(M, 256) * (N / 256);
  return kb % 128 == 0 && kb >= 256 && N % 256 == 0 && N / 256 <= 4 && tiles >= (N == 256 ? 256 : 768);
}
size_t xmli_gemm_ln_workspace_bytes(int64_t M, int N) {
  return align_up((size_t)(cdiv(M, 256) + 1) * 4, 256) + align_up((size_t)M * 2 * (N / 256 + 1) * 2 * 4, 256);
}
int xmli_gemm_ln(const void* A, const void* W, const float* bias, const void* addend, const float* ln_g, const float* ln_b,
                 void* y, int64_t M, int N, int K, int relu, int add_mode, int seq_len, int dt, void* ln_ws,
                 hipStream_t st) {
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
