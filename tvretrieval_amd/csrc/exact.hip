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
                                         int32_t* __restrict__ n_fail, int nq) {
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
  if (f) atomicAdd(n_fail, 1);
  if (alpha != 0.f)
    for (int j = 0; j < k; ++j) top_val[(int64_t)q * k + j] = expf(alpha * top_val[(int64_t)q * k + j]);
}

}  // namespace

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
                                     int outside, int32_t* fail, float* eps_out, int32_t* n_fail, int nq,
                                     xml_stream_t stream) {
  XML_ENTER();
  if (!filter_scores || !top_val || !eq0 || !fail || !n_fail || nq <= 0 || m <= 0 || k <= 0 || k > m) return XML_ERR_BAD_ARG;
  if (n_mod < 1 || n_mod > 2 || (n_mod == 2 && !eq1)) return XML_ERR_BAD_ARG;
  hipLaunchKernelGGL(exact_certificate_kernel, dim3(cdiv(nq, 128)), dim3(128), 0, (hipStream_t)stream, filter_scores, m,
                     top_val, k, eq0, eq1 ? eq1 : eq0, ec0, ec1, n_mod, slack, alpha, outside, fail, eps_out, n_fail, nq);
  XML_CHECK_LAUNCH();
  return XML_OK;
}
