// Exact-rank mode of the corpus search (bf16 K6 as a FILTER, f32 re-score of its candidates, per-query certificate).
//
// The reference ranks videos by f32 scores (xml/inference.py:317,347-348).  The bf16 similarity pass reproduces those
// scores to ~3e-5, which moves ~1 % of the top-100 memberships when neighbouring scores are closer than that.  In exact
// mode the corpus and the queries are encoded in f32; the bf16 pass sees both operands ROUNDED ONCE to bf16 and only
// proposes M >= k candidates per query, which xml_q2c_rescore (convse.hip) re-scores against the f32 operands.  The two
// helpers here make that filter safe:
//
//   xml_round_bf16_rows_err   yb = rne_bf16(y) and err[row] = || y - yb ||_2 for every (already L2-normalised) f32 row --
//                             the rounding-error norms the certificate's bound is made of;
//   xml_exact_certificate     per query: no video OUTSIDE the candidate set can belong to the f32 top-k if
//                                 b_M + eps_q < T_k
//                             (b_M = the M-th largest FILTER score, T_k = the k-th largest RE-SCORED value) with
//                                 eps_q = mean_m( e_q,m * |c| + |q_b| * E_c,m ) + slack        (Cauchy-Schwarz:
//                                 | q.c - q_b.c_b | <= |q - q_b| |c| + |q_b| |c - c_b| , masked clips are -1e10 on both sides,
//                                 and | max_l a_l - max_l b_l | <= max_l | a_l - b_l |),
//                             e_q,m this query's rounding-error norm, E_c,m the largest one of the corpus, slack the f32
//                             accumulation bound of both dot products.  Queries that fail are re-done against the whole
//                             corpus in f32 by the caller.  Also turns the raw top-k values into exp(alpha * s) (what K8
//                             emits, xml/inference.py:317).
//   xml_select_ge_rows        the second tier for the few queries whose certificate fails: every video whose FILTER score
//                             reaches T_k - eps_q (the re-scored k-th value is a lower bound of the final one, so nothing
//                             below that line can enter the top-k) -- usually a few dozen more than M; they are re-scored
//                             like the first M and the lists re-selected, no second certificate needed.
#include "common.h"

namespace {

// one wave per row; 16-byte loads, 8 elements per lane and trip
__global__ __launch_bounds__(256) void round_bf16_rows_err_kernel(const float* __restrict__ y, bf16_t* __restrict__ yb,
                                                                  float* __restrict__ err, int64_t rows, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* px = y + row * d;
  bf16_t* pb = yb + row * d;
  float s = 0.f;
  for (int c = lane * 8; c < d; c += 64 * 8) {
    float f[8];
    ld8<float>(px + c, f);
    const uint4 packed = pack16<bf16_t>(f);
    float g[8];
    unpack16<bf16_t>(packed, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float dlt = f[e] - g[e];          // exact in f32 (Sterbenz-like: g is f rounded to 8 significant bits)
      s += dlt * dlt;
    }
    *reinterpret_cast<uint4*>(pb + c) = packed;
  }
  s = wave_sum(s);
  if (lane == 0) err[row] = sqrtf(s);
}

__global__ void exact_certificate_kernel(const float* __restrict__ filt, int m, float* __restrict__ top_val, int k,
                                         const float* __restrict__ eq0, const float* __restrict__ eq1, float ec0,
                                         float ec1, int n_mod, float slack, float alpha, int outside,
                                         int32_t* __restrict__ fail, float* __restrict__ eps_out,
                                         float* __restrict__ thr_out, int32_t* __restrict__ n_fail, int nq) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const float c_norm = 1.f + 1e-6f;                       // | c | of an f32-normalised row
  float eps = eq0[q] * c_norm + (c_norm + eq0[q]) * ec0;
  if (n_mod == 2) eps = (eps + eq1[q] * c_norm + (c_norm + eq1[q]) * ec1) * 0.5f;
  eps += slack;
  const float b_m = filt[(int64_t)q * m + (m - 1)];
  const float t_k = top_val[(int64_t)q * k + (k - 1)];
  const int f = (outside && !(b_m + eps < t_k)) ? 1 : 0;
  fail[q] = f;
  if (eps_out) eps_out[q] = eps;
  if (thr_out) thr_out[q] = t_k - eps;                    // second tier: filter scores below this line cannot enter the top-k
  if (f) atomicAdd(n_fail, 1);
  if (alpha != 0.f)
    for (int j = 0; j < k; ++j) top_val[(int64_t)q * k + j] = expf(alpha * top_val[(int64_t)q * k + j]);
}

// one workgroup per row: the columns whose score reaches the row's threshold.  idx == NULL: count only.
__global__ __launch_bounds__(256) void select_ge_rows_kernel(const float* __restrict__ scores, int64_t ld, const float* __restrict__ thr,
                                                             int32_t* __restrict__ idx, int cap, int32_t* __restrict__ cnt,
                                                             int n) {
  __shared__ int s_cnt;
  const int row = blockIdx.x, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const float t = thr[row];
  const float* r = scores + (int64_t)row * ld;
  for (int base = 0; base < n; base += 256) {
    const int i = base + threadIdx.x;
    const bool take = i < n && r[i] >= t;
    const unsigned long long bal = __ballot(take);
    if (bal) {
      int off = 0;
      const int leader = __ffsll((long long)bal) - 1;
      if (lane == leader) off = atomicAdd(&s_cnt, (int)__popcll(bal));
      off = __shfl(off, leader, 64) + (int)__popcll(bal & ((1ull << lane) - 1ull));
      if (take && idx && off < cap) idx[(int64_t)row * cap + off] = i;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) cnt[row] = s_cnt;
}

}  // namespace

extern "C" int xml_select_ge_rows(const float* scores, int64_t ld, const float* thr, int32_t* idx, int cap, int32_t* cnt,
                                  int rows, int n, xml_stream_t stream) {
  XML_ENTER();
  if (!scores || !thr || !cnt || rows <= 0 || n <= 0 || ld < n || (idx && cap <= 0)) return XML_ERR_BAD_ARG;
  hipLaunchKernelGGL(select_ge_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, scores, ld, thr, idx, cap, cnt, n);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" int xml_round_bf16_rows_err(const float* y, void* yb, float* err, int64_t rows, int d, xml_stream_t stream) {
  XML_ENTER();
  if (!y || !yb || !err || rows <= 0 || d <= 0) return XML_ERR_BAD_ARG;
  if (d % 8) return XML_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(round_bf16_rows_err_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, y, (bf16_t*)yb,
                     err, rows, d);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" int xml_exact_certificate(const float* filter_scores, int m, float* top_val, int k, const float* eq0,
                                     const float* eq1, float ec0, float ec1, int n_mod, float slack, float alpha,
                                     int outside, int32_t* fail, float* eps_out, float* thr_out, int32_t* n_fail, int nq,
                                     xml_stream_t stream) {
  XML_ENTER();
  if (!filter_scores || !top_val || !eq0 || !fail || !n_fail || nq <= 0 || m <= 0 || k <= 0 || k > m) return XML_ERR_BAD_ARG;
  if (n_mod < 1 || n_mod > 2 || (n_mod == 2 && !eq1)) return XML_ERR_BAD_ARG;
  hipLaunchKernelGGL(exact_certificate_kernel, dim3(cdiv(nq, 128)), dim3(128), 0, (hipStream_t)stream, filter_scores, m,
                     top_val, k, eq0, eq1 ? eq1 : eq0, ec0, ec1, n_mod, slack, alpha, outside, fail, eps_out, thr_out, n_fail,
                     nq);
  XML_CHECK_LAUNCH();
  return XML_OK;
}
