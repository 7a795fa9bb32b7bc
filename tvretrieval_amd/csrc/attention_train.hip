// Fused multi-head attention for the TRAINING step (bf16 storage): forward with dropout on the probabilities, and the
// whole backward pass, one workgroup per (sequence, head) each.
//   reference: BertSelfAttention.forward   xml/model_components.py:266-303 (probabilities dropout :297)
// Replaces, per attention layer, the chain split_heads x3 -> batched QK^T -> softmax -> dropout -> batched PV ->
// merge_heads (8 launches) and its 17-launch backward (train.hip / autograd.py) with one launch each way: L <= 128, so
// the L x L score tile of a head lives in registers and every operand tile in LDS.
//
//   forward    S = Q K^T / sqrt(dh) + (1 - qm (x) km) * -1e4 ;  P = softmax_rows(S) ;  Pd = P o M / (1 - p) ;  O = Pd V
//   backward   dPd = dO V^T ;  dP = dPd o M / (1 - p) ;  delta_i = sum_j dP_ij P_ij ;  dS = P o (dP - delta) / sqrt(dh)
//              dQ = dS K ;  dK = dS^T Q ;  dV = Pd^T dO
// Nothing but Q, K, V is saved by the forward: the backward recomputes S and P (same instructions as the forward), and the
// dropout mask M is the counter-based hash of train.hip's xml_dropout evaluated at the element's index in the
// (N * heads, L8, L8) layout the unfused path used -- the two paths drop the SAME elements for a given seed.
//
// Workgroup = 4 waves; wave w owns query row tiles w and w + 4 (16 rows each) for everything that is row-wise in the
// queries (S, P, dP, dS, O, dQ) and key row tiles w, w + 4 for dK / dV.  LDS (DH = 192: 138 KiB, one workgroup per CU):
//   R1  [128][DH] rows (+16 B pad)   K        -> dO      -> Q
//   R2  69.6 KiB                     V        -> Pd^T [key][query] and dS^T [key][query] (272-byte rows)
//   4 per-wave patches [16][128]     C-layout -> A-operand layout for dQ (and for O in the forward)
// Operands that are needed TRANSPOSED (K^T for dQ, dO^T for dV, Q^T for dK, V^T for O) stay row-major in LDS and are read
// with ds_read_b64_tr_b16 (see attention.hip); Pd^T / dS^T are written transposed straight from the accumulator layout
// (a lane holds 4 consecutive query rows of one key column = one 8-byte store).
#include "gemm.h"
#include "internal.h"

namespace {

constexpr int AT_MAXNT = 8;            // 16-wide tiles along a sequence (L <= 128)
constexpr int AT_PSTRIDE = 128 * 2 + 16;

struct AttnTrainArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;
  int ldq, ldk, ldv;
  const float* q_mask; const float* k_mask;
  const bf16_t* dout; bf16_t* out; int ldo;
  bf16_t* dq; bf16_t* dk; bf16_t* dv;
  int lddq, lddk, lddv;
  int lq, lk, lq8, lk8;
  float sqrt_dh;
  uint32_t thresh; float scale;                        // dropout: keep(i) = hash(i) >= thresh, kept values * scale
  uint64_t seed; const uint64_t* seed_dev;             // host seed + optional device-resident base seed (xml_seed_words)
};

typedef short at_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 at_read_tr16(const char* p) {
  const at_v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) at_v4s*)p);
  return __builtin_bit_cast(uint2, r);
}
// the hash of train.hip's dropout_kernel (must stay identical)
__device__ __forceinline__ uint32_t at_drop_hash(uint64_t i, uint32_t s0, uint32_t s1) {
  uint32_t h = (uint32_t)i * 0x9E3779B1u + s0;
  h ^= (uint32_t)(i >> 32) * 0x85EBCA77u;
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h += s1; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

// rows [0, valid) of a (rows, ld) operand's head slice -> LDS rows of DH * 2 + 16 bytes; rows [valid, pad) zero.  All loads
// of the tile are issued before the first LDS store.
template <int DH>
__device__ __forceinline__ void at_stage_rows(char* dst, const bf16_t* src, int ld, int valid, int pad, int tid) {
  constexpr int VPR = DH / 8;                  // 16-byte vectors per row
  constexpr int NVB = 128 * VPR / 256;
  constexpr int KS = DH * 2 + 16;
  uint4 b[NVB];
#pragma unroll
  for (int j = 0; j < NVB; ++j) {
    const int i = tid + j * 256;
    const int r = i / VPR, c = i % VPR;
    b[j] = make_uint4(0, 0, 0, 0);
    if (r < valid) b[j] = ld_global16(src + (int64_t)r * ld + c * 8);
  }
#pragma unroll
  for (int j = 0; j < NVB; ++j) {
    const int i = tid + j * 256;
    const int r = i / VPR, c = i % VPR;
    if (r < pad) *reinterpret_cast<uint4*>(dst + r * KS + c * 16) = b[j];
  }
}

// A-operand fragments of this wave's two row tiles, straight from global memory
template <int DH>
__device__ __forceinline__ void at_load_frags(uint4 (&f)[2][DH / 32], const bf16_t* base, int ld, int l, int wave, int fr,
                                              int fg) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = (wave + t * 4) * 16 + fr;
#pragma unroll
    for (int c = 0; c < DH / 32; ++c) {
      f[t][c] = make_uint4(0, 0, 0, 0);
      if (row < l) f[t][c] = ld_global16(base + (int64_t)row * ld + c * 32 + fg * 8);
    }
  }
}

// x[t][j] = A-rows (fragments) . B-rows^T for the wave's two row tiles against the 8 row tiles of an LDS operand
template <int DH>
__device__ __forceinline__ void at_rows_dot_rows(f32x4 (&x)[2][AT_MAXNT], const uint4 (&fa)[2][DH / 32], const char* s_b,
                                                 int fr, int fg) {
  constexpr int KS = DH * 2 + 16;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int j = 0; j < AT_MAXNT; ++j) x[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < DH / 32; ++c) {
      uint4 b[AT_MAXNT];
#pragma unroll
      for (int j = 0; j < AT_MAXNT; ++j) b[j] = *reinterpret_cast<const uint4*>(s_b + (j * 16 + fr) * KS + c * 64 + fg * 16);
#pragma unroll
      for (int j = 0; j < AT_MAXNT; ++j) Mma<bf16_t>::chunk(x[t][j], fa[t][c], b[j]);
    }
  }
}

// scores -> probabilities in place (the arithmetic of attention_core_kernel's bf16 path)
__device__ __forceinline__ void at_softmax(f32x4 (&p)[2][AT_MAXNT], const float (&km)[AT_MAXNT], const float (&qmk)[2][4],
                                           int lk, float inv_sqrt_dh, int fr) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float qm = qmk[t][r];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < AT_MAXNT; ++j) {
        const int col = j * 16 + fr;
        float s = -INFINITY;
        if (col < lk) s = p[t][j][r] * inv_sqrt_dh + (1.f - qm * km[j]) * -10000.f;
        p[t][j][r] = s;
        mx = fmaxf(mx, s);
      }
      mx = lane16_max_dpp(mx);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < AT_MAXNT; ++j) {
        const float e = __builtin_amdgcn_exp2f((p[t][j][r] - mx) * 1.4426950408889634f);
        p[t][j][r] = e;
        sum += e;
      }
      sum = lane16_sum_dpp(sum);
      const float inv_sum = 1.0f / sum;
#pragma unroll
      for (int j = 0; j < AT_MAXNT; ++j) p[t][j][r] *= inv_sum;
    }
}

// o[d] (16 rows x DH) = X (16 rows x lkp, C layout in x) . B, with B = rows of an LDS operand [k][DH] read transposed.
// X goes through the wave's patch to reach the A-operand layout.
template <int DH>
__device__ __forceinline__ void at_tile_times_rows(f32x4 (&o)[DH / 16], const f32x4 (&x)[AT_MAXNT], char* s_patch,
                                                   const char* s_b, int kp, int fr, int fg) {
  constexpr int KS = DH * 2 + 16;
#pragma unroll
  for (int j = 0; j < AT_MAXNT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      DT<bf16_t>::st(reinterpret_cast<bf16_t*>(s_patch + (fg * 4 + r) * AT_PSTRIDE) + j * 16 + fr, x[j][r]);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int d = 0; d < DH / 16; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < kp / 32; ++c) {
    const uint4 a = *reinterpret_cast<const uint4*>(s_patch + fr * AT_PSTRIDE + c * 64 + fg * 16);
    const char* vb0 = s_b + (c * 32 + fg * 8 + (fr >> 2)) * KS + (fr & 3) * 8;
#pragma unroll
    for (int d = 0; d < DH / 16; ++d) {
      const uint2 lo = at_read_tr16(vb0 + d * 32), hi = at_read_tr16(vb0 + d * 32 + 4 * KS);
      Mma<bf16_t>::chunk(o[d], a, make_uint4(lo.x, lo.y, hi.x, hi.y));
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// o[d] (16 rows x DH) = A^T-tile rows (already transposed in LDS: [row][k], 272-byte rows) . B read transposed
template <int DH>
__device__ __forceinline__ void at_trows_times_rows(f32x4 (&o)[DH / 16], const char* s_at, int row0, const char* s_b, int kp,
                                                    int fr, int fg) {
  constexpr int KS = DH * 2 + 16;
#pragma unroll
  for (int d = 0; d < DH / 16; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < kp / 32; ++c) {
    const uint4 a = *reinterpret_cast<const uint4*>(s_at + (row0 + fr) * AT_PSTRIDE + c * 64 + fg * 16);
    const char* vb0 = s_b + (c * 32 + fg * 8 + (fr >> 2)) * KS + (fr & 3) * 8;
#pragma unroll
    for (int d = 0; d < DH / 16; ++d) {
      const uint2 lo = at_read_tr16(vb0 + d * 32), hi = at_read_tr16(vb0 + d * 32 + 4 * KS);
      Mma<bf16_t>::chunk(o[d], a, make_uint4(lo.x, lo.y, hi.x, hi.y));
    }
  }
}

template <int DH>
__device__ __forceinline__ void at_store_tile(bf16_t* base, int ld, int row0, int l, const f32x4 (&o)[DH / 16], int fr, int fg) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + fg * 4 + r;
    if (row >= l) continue;
    bf16_t* po = base + (int64_t)row * ld;
#pragma unroll
    for (int d = 0; d < DH / 16; ++d) DT<bf16_t>::st(po + d * 16 + fr, o[d][r]);
  }
}

template <int DH>
__device__ __forceinline__ void at_load_masks(float (&km)[AT_MAXNT], float (&qmk)[2][4], const AttnTrainArgs& a, int n, int wave,
                                              int fr, int fg) {
#pragma unroll
  for (int j = 0; j < AT_MAXNT; ++j) {
    const int col = j * 16 + fr;
    km[j] = (col < a.lk) ? a.k_mask[(int64_t)n * a.lk + col] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (wave + t * 4) * 16 + fg * 4 + r;
      qmk[t][r] = (a.q_mask && row < a.lq) ? a.q_mask[(int64_t)n * a.lq + row] : 1.f;
    }
}

// ---- forward ---------------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_train_fwd_kernel(AttnTrainArgs a) {
  constexpr int KS = DH * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int head = blockIdx.x, n = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int lq = a.lq, lk = a.lk;
  const int lkp = (lk + 31) / 32 * 32;
  const int nqt = (lq + 15) / 16;
  char* s_k = smem;
  char* s_v = smem + 128 * KS;
  char* s_p = s_v + 128 * KS + wave * (16 * AT_PSTRIDE);
  const bf16_t* qbase = a.q + (int64_t)n * lq * a.ldq + head * DH;

  at_stage_rows<DH>(s_k, a.k + (int64_t)n * lk * a.ldk + head * DH, a.ldk, lk, 128, tid);
  at_stage_rows<DH>(s_v, a.v + (int64_t)n * lk * a.ldv + head * DH, a.ldv, lk, 128, tid);
  uint4 qa[2][DH / 32];
  at_load_frags<DH>(qa, qbase, a.ldq, lq, wave, fr, fg);
  float km[AT_MAXNT], qmk[2][4];
  at_load_masks<DH>(km, qmk, a, n, wave, fr, fg);
  __syncthreads();

  f32x4 p[2][AT_MAXNT];
  at_rows_dot_rows<DH>(p, qa, s_k, fr, fg);
  at_softmax(p, km, qmk, lk, 1.0f / a.sqrt_dh, fr);
  if (a.thresh) {
    const int64_t unit = (int64_t)n * gridDim.x + head;
    uint32_t s0, s1;
    xml_seed_words(a.seed, a.seed_dev, s0, s1);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < AT_MAXNT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (wave + t * 4) * 16 + fg * 4 + r, col = j * 16 + fr;
          const uint64_t idx = (uint64_t)((unit * a.lq8 + row) * a.lk8 + col);
          p[t][j][r] = at_drop_hash(idx, s0, s1) >= a.thresh ? p[t][j][r] * a.scale : 0.f;
        }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int qt = wave + t * 4;
    if (qt >= nqt) continue;
    f32x4 o[DH / 16];
    at_tile_times_rows<DH>(o, p[t], s_p, s_v, lkp, fr, fg);
    at_store_tile<DH>(a.out + (int64_t)n * lq * a.ldo + head * DH, a.ldo, qt * 16, lq, o, fr, fg);
  }
}

// ---- backward --------------------------------------------------------------------------------------------------------
// Where a workgroup's ~70 K cycles go (128 x 100 x 768, 4 heads; XML_DEBUG_EXTRA=-DXML_AT_PROBE, tools/bench_attn_train.py):
// first loads + K / V staging 14 K -- all 256 resident workgroups fetch their 154 KB at the same moment, a bandwidth burst with
// nothing to overlap it at one workgroup per CU (issuing every load before the first use and both tiles' loads together: no
// change) --, S 2.5 K, softmax 4 K, dP 2.3 K, dropout + dS 8.3 K (hash multiplies hoisted per row: no change), dQ 11.5 K,
// transposes to LDS 2.4 K, dO staging 2.3-3.7 K, dV 8.6 K, Q staging 3-4 K, dK 8.8 K; the 480 MFMAs are 7.7 K of it.
#ifdef XML_AT_PROBE
}  // namespace
__device__ unsigned long long g_at_probe[4 * 16];      // stage timers of workgroup (head 1, sequence 7)
extern "C" int xml_debug_read_at_probe(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_at_probe), sizeof(g_at_probe)) == hipSuccess ? 0 : -4;
}
namespace {
#define AT_MARK(i) do { if (blockIdx.x == 1 && blockIdx.y == 7 && lane == 0) g_at_probe[wave * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AT_MARK(i) do { } while (0)
#endif
template <int DH>
__global__ __launch_bounds__(256) void attn_train_bwd_kernel(AttnTrainArgs a) {
  constexpr int KS = DH * 2 + 16;
  constexpr int R2_BYTES = (128 * KS > 2 * 128 * AT_PSTRIDE) ? 128 * KS : 2 * 128 * AT_PSTRIDE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int head = blockIdx.x, n = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int lq = a.lq, lk = a.lk;
  const int lkp = (lk + 31) / 32 * 32, lqp = (lq + 31) / 32 * 32;
  const int nqt = (lq + 15) / 16, nkt = (lk + 15) / 16;
  char* r1 = smem;
  char* r2 = smem + 128 * KS;
  char* s_p = r2 + R2_BYTES + wave * (16 * AT_PSTRIDE);
  const bf16_t* qbase = a.q + (int64_t)n * lq * a.ldq + head * DH;
  const bf16_t* kbase = a.k + (int64_t)n * lk * a.ldk + head * DH;
  const bf16_t* vbase = a.v + (int64_t)n * lk * a.ldv + head * DH;
  const bf16_t* dobase = a.dout + (int64_t)n * lq * a.ldo + head * DH;

  // ---- step 1: K -> R1, V -> R2; this wave's Q and dO row fragments and the masks in registers
  AT_MARK(0);
  at_stage_rows<DH>(r1, kbase, a.ldk, lk, 128, tid);
  at_stage_rows<DH>(r2, vbase, a.ldv, lk, 128, tid);
  uint4 qa[2][DH / 32], da[2][DH / 32];
  at_load_frags<DH>(qa, qbase, a.ldq, lq, wave, fr, fg);
  at_load_frags<DH>(da, dobase, a.ldo, lq, wave, fr, fg);
  float km[AT_MAXNT], qmk[2][4];
  at_load_masks<DH>(km, qmk, a, n, wave, fr, fg);
  __syncthreads();
  AT_MARK(1);

  // ---- step 2: P (recomputed), dPd = dO V^T, then Pd and dS in registers
  f32x4 p[2][AT_MAXNT], dp[2][AT_MAXNT];
  at_rows_dot_rows<DH>(p, qa, r1, fr, fg);
  const float inv_sqrt_dh = 1.0f / a.sqrt_dh;
  AT_MARK(2);
  at_softmax(p, km, qmk, lk, inv_sqrt_dh, fr);
  AT_MARK(3);
  at_rows_dot_rows<DH>(dp, da, r2, fr, fg);
  AT_MARK(4);
  {
    const int64_t unit = (int64_t)n * gridDim.x + head;
    uint32_t s0 = 0, s1 = 0;
    if (a.thresh) xml_seed_words(a.seed, a.seed_dev, s0, s1);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (wave + t * 4) * 16 + fg * 4 + r;
        uint32_t keep = 0xffu;                     // bit j: element (row, j * 16 + fr) kept
        if (a.thresh) {
          keep = 0u;
#pragma unroll
          for (int j = 0; j < AT_MAXNT; ++j) {
            const uint64_t idx = (uint64_t)((unit * a.lq8 + row) * a.lk8 + j * 16 + fr);
            keep |= (at_drop_hash(idx, s0, s1) >= a.thresh ? 1u : 0u) << j;
          }
        }
        float delta = 0.f;
#pragma unroll
        for (int j = 0; j < AT_MAXNT; ++j) {
          const float g = ((keep >> j) & 1u) ? dp[t][j][r] * a.scale : 0.f;     // dP: through the dropout
          dp[t][j][r] = g;
          delta += g * p[t][j][r];
        }
        delta = lane16_sum_dpp(delta);
#pragma unroll
        for (int j = 0; j < AT_MAXNT; ++j) {
          const float pr = p[t][j][r];
          dp[t][j][r] = pr * (dp[t][j][r] - delta) * inv_sqrt_dh;                 // dS (scaled: the scores were q.k / sqrt(dh))
          p[t][j][r] = ((keep >> j) & 1u) ? pr * a.scale : 0.f;                   // Pd
        }
      }
  }

  AT_MARK(5);
  // ---- step 3: dQ = dS K for this wave's rows (K^T by transpose reads of R1)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int qt = wave + t * 4;
    if (qt >= nqt) continue;
    f32x4 o[DH / 16];
    at_tile_times_rows<DH>(o, dp[t], s_p, r1, lkp, fr, fg);
    at_store_tile<DH>(a.dq + (int64_t)n * lq * a.lddq + head * DH, a.lddq, qt * 16, lq, o, fr, fg);
  }
  AT_MARK(6);
  __syncthreads();                                  // everyone is done with K (R1) and V (R2)
  AT_MARK(7);

  // ---- step 4: Pd^T and dS^T -> R2 ([key][query], 272-byte rows); dO -> R1
  {
    char* s_pdt = r2;
    char* s_dst = r2 + 128 * AT_PSTRIDE;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int qt = wave + t * 4;
#pragma unroll
      for (int j = 0; j < AT_MAXNT; ++j) {
        const int off = (j * 16 + fr) * AT_PSTRIDE + (qt * 16 + fg * 4) * 2;
        uint2 u, w;
        u.x = f32x2_to_bf16x2(p[t][j][0], p[t][j][1]); u.y = f32x2_to_bf16x2(p[t][j][2], p[t][j][3]);
        w.x = f32x2_to_bf16x2(dp[t][j][0], dp[t][j][1]); w.y = f32x2_to_bf16x2(dp[t][j][2], dp[t][j][3]);
        *reinterpret_cast<uint2*>(s_pdt + off) = u;
        *reinterpret_cast<uint2*>(s_dst + off) = w;
      }
    }
  }
  AT_MARK(8);
  at_stage_rows<DH>(r1, dobase, a.ldo, lq, 128, tid);
  __syncthreads();
  AT_MARK(9);

  // ---- step 5: dV = Pd^T dO for key tiles w, w + 4
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int jt = wave + t * 4;
    if (jt >= nkt) continue;
    f32x4 o[DH / 16];
    at_trows_times_rows<DH>(o, r2, jt * 16, r1, lqp, fr, fg);
    at_store_tile<DH>(a.dv + (int64_t)n * lk * a.lddv + head * DH, a.lddv, jt * 16, lk, o, fr, fg);
  }
  AT_MARK(10);
  __syncthreads();
  at_stage_rows<DH>(r1, qbase, a.ldq, lq, 128, tid);
  __syncthreads();
  AT_MARK(11);

  // ---- step 6: dK = dS^T Q
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int jt = wave + t * 4;
    if (jt >= nkt) continue;
    f32x4 o[DH / 16];
    at_trows_times_rows<DH>(o, r2 + 128 * AT_PSTRIDE, jt * 16, r1, lqp, fr, fg);
    at_store_tile<DH>(a.dk + (int64_t)n * lk * a.lddk + head * DH, a.lddk, jt * 16, lk, o, fr, fg);
  }
  AT_MARK(12);
}

template <int DH> size_t at_fwd_lds() { return (size_t)2 * 128 * (DH * 2 + 16) + 4 * 16 * AT_PSTRIDE; }
template <int DH> size_t at_bwd_lds() {
  const size_t ks = DH * 2 + 16;
  const size_t r2 = 128 * ks > (size_t)2 * 128 * AT_PSTRIDE ? 128 * ks : (size_t)2 * 128 * AT_PSTRIDE;
  return 128 * ks + r2 + 4 * 16 * AT_PSTRIDE;
}

template <int DH>
int at_launch(bool bwd, const AttnTrainArgs& a, int64_t n, int n_heads, hipStream_t st) {
  if (bwd) {
    const size_t lds = at_bwd_lds<DH>();
    if (lds > 160 * 1024 || !xml_lds_attr_once<attn_train_bwd_kernel<DH>>(160 * 1024)) return XML_ERR_LAUNCH;
    hipLaunchKernelGGL(attn_train_bwd_kernel<DH>, dim3(n_heads, (unsigned)n), dim3(256), lds, st, a);
  } else {
    const size_t lds = at_fwd_lds<DH>();
    if (lds > 160 * 1024 || !xml_lds_attr_once<attn_train_fwd_kernel<DH>>(160 * 1024)) return XML_ERR_LAUNCH;
    hipLaunchKernelGGL(attn_train_fwd_kernel<DH>, dim3(n_heads, (unsigned)n), dim3(256), lds, st, a);
  }
  XML_CHECK_LAUNCH();
  return XML_OK;
}

int at_dispatch(bool bwd, AttnTrainArgs& a, int64_t n, int hidden, int n_heads, float p_drop, uint64_t seed,
                const uint64_t* seed_dev, hipStream_t st) {
  if (n <= 0 || a.lq <= 0 || a.lk <= 0 || n_heads <= 0 || hidden % n_heads || !(p_drop >= 0.f) || p_drop >= 1.f)
    return XML_ERR_BAD_ARG;
  if (a.lq > 128 || a.lk > 128) return XML_ERR_UNSUPPORTED;
  const int dh = hidden / n_heads;
  a.lq8 = (a.lq + 7) / 8 * 8; a.lk8 = (a.lk + 7) / 8 * 8;
  a.sqrt_dh = sqrtf((float)dh);
  a.thresh = (uint32_t)((double)p_drop * 4294967296.0);          // xml_dropout's threshold / scale / seed words
  a.scale = 1.f / (1.f - p_drop);
  a.seed = seed; a.seed_dev = seed_dev;
  switch (dh) {
    case 32: return at_launch<32>(bwd, a, n, n_heads, st);
    case 64: return at_launch<64>(bwd, a, n, n_heads, st);
    case 96: return at_launch<96>(bwd, a, n, n_heads, st);
    case 128: return at_launch<128>(bwd, a, n, n_heads, st);
    case 192: return at_launch<192>(bwd, a, n, n_heads, st);
    default: return XML_ERR_UNSUPPORTED;
  }
}

}  // namespace

extern "C" int xml_attention_train_supported(int lq, int lk, int hidden, int n_heads, int dt) {
  if (dt != XML_BF16 || n_heads <= 0 || hidden % n_heads || lq <= 0 || lk <= 0 || lq > 128 || lk > 128) return 0;
  const int dh = hidden / n_heads;
  return dh == 32 || dh == 64 || dh == 96 || dh == 128 || dh == 192;
}

extern "C" int xml_attention_train_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                                       const float* q_mask, const float* k_mask, void* out, int ldo, int64_t n, int lq, int lk,
                                       int hidden, int n_heads, float p_drop, uint64_t seed, const uint64_t* seed_dev, int dt,
                                       xml_stream_t stream) {
  XML_ENTER();
  if (!q || !k || !v || !k_mask || !out) return XML_ERR_BAD_ARG;
  if (dt != XML_BF16) return XML_ERR_UNSUPPORTED;
  if ((ldq | ldk | ldv) % 8) return XML_ERR_BAD_ARG;        // 16-byte row-fragment loads
  AttnTrainArgs a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
  a.q_mask = q_mask; a.k_mask = k_mask; a.out = (bf16_t*)out; a.ldo = ldo; a.lq = lq; a.lk = lk;
  return at_dispatch(false, a, n, hidden, n_heads, p_drop, seed, seed_dev, (hipStream_t)stream);
}

extern "C" int xml_attention_train_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                                       const float* q_mask, const float* k_mask, const void* dout, int ldo, void* dq, int lddq,
                                       void* dk, int lddk, void* dv, int lddv, int64_t n, int lq, int lk, int hidden,
                                       int n_heads, float p_drop, uint64_t seed, const uint64_t* seed_dev, int dt,
                                       xml_stream_t stream) {
  XML_ENTER();
  if (!q || !k || !v || !k_mask || !dout || !dq || !dk || !dv) return XML_ERR_BAD_ARG;
  if (dt != XML_BF16) return XML_ERR_UNSUPPORTED;
  if ((ldq | ldk | ldv | ldo) % 8) return XML_ERR_BAD_ARG;
  AttnTrainArgs a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
  a.q_mask = q_mask; a.k_mask = k_mask; a.dout = (const bf16_t*)dout; a.ldo = ldo;
  a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.lq = lq; a.lk = lk;
  return at_dispatch(true, a, n, hidden, n_heads, p_drop, seed, seed_dev, (hipStream_t)stream);
}
