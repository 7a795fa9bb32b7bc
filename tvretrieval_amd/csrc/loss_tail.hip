// The serial middle of the training step: between the forward pass of the context encoders and their
// backward pass sits the chain  query encoder -> modular pooling -> video-level scores -> ranking loss -> loss sum ->
// and back.  Nothing else can run beside it, so its ~90 small launches (5 us each in a HIP graph, longer when a kernel
// walks rows one after the other) were 1 ms of the 4.8 ms step at the configs[4] shape.  This file holds the fused forms:
//   xml_q2c_scores_l2norm_bwd   VideoLevelScoresFn backward in ONE launch (was: 2 fills, arg-max re-scan with f32 atomics,
//                               slice copy, 2 l2norm backward passes over mostly-zero f32 gradients -- 135 us per modality)
//   xml_modular_pool_bwd        16-byte loads, clips in flight (was 124 us: one thread walked the clips one by one)
//   xml_loss_combine            the weighted loss sum and its backward (was ~15 scalar torch kernels)
// reference: get_video_level_scores xml/model_xml.py:436-453, get_modularized_queries :410-423, forward :241-251.
#include "gemm.h"
#include "internal.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// arg max_l mask_logits(qn . cn[l]) of one (query, video) pair, by a whole 256-thread block: wave w scans the clips
// l = w, w + 4, ...; four clips in flight; lanes hold 8-element chunks; first clip on ties (what max() picks in the forward).
// s_red: 8 floats of LDS scratch.  Returns the clip in every thread.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ int pair_argmax(const T* __restrict__ q, const T* __restrict__ cbase, const float* __restrict__ mrow,
                                           int L, int hidden, float* s_best, int* s_bl) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunks = hidden >> 3;
  float best = -INFINITY;
  int best_l = 0x7fffffff;
  for (int l0 = wave; l0 < L; l0 += 16) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < chunks; c += 64) {
      float qv[8];
      ld8<T>(q + c * 8, qv);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int l = min(l0 + 4 * u, L - 1);
        float cv[8];
        ld8<T>(cbase + (int64_t)l * hidden + c * 8, cv);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[u] += qv[e] * cv[e];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int l = l0 + 4 * u;
      float v = wave_sum(s[u]);
      if (l < L) {
        const float mk = mrow[l];
        v = v * mk + (1.f - mk) * -1e10f;
        if (v > best) { best = v; best_l = l; }
      }
    }
  }
  __syncthreads();                      // (s_best / s_bl may still be read from the previous pair)
  if (lane == 0) { s_best[wave] = best; s_bl[wave] = best_l; }
  __syncthreads();
  best = s_best[0]; best_l = s_bl[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float ob = s_best[w];
    const int ol = s_bl[w];
    if (ob > best || (ob == best && ol < best_l)) { best = ob; best_l = ol; }
  }
  return best_l;
}

// ---------------------------------------------------------------------------------------------------------------------
// In-batch video-level scores with the arg-max clip kept (training forward).  Same tile scheme as q2c_scores_kernel
// (q2c.hip): 128 queries x 128 clip columns per workgroup, a tile holds vpt = 128 / lpad whole videos, lpad = ceil16(L) --
// but the clip rows are read from the UNPADDED (nv, L, hidden) tensor (columns >= L of a video are zero rows with mask 0), and
// the epilogue carries (value, column) through the 16-lane and cross-tile reductions: first clip on ties.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane16_min_int_dpp(int v) {
#define XML_ROR_I(x, n) __builtin_amdgcn_update_dpp(0x7fffffff, (x), 0x120 + (n), 0xf, 0xf, false)
  v = min(v, XML_ROR_I(v, 8));
  v = min(v, XML_ROR_I(v, 4));
  v = min(v, XML_ROR_I(v, 2));
  v = min(v, XML_ROR_I(v, 1));
#undef XML_ROR_I
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void q2c_scores_arg_kernel(const T* __restrict__ qn, const T* __restrict__ cn,
                                                             const float* __restrict__ mask, float* __restrict__ out,
                                                             int64_t ld_out, int32_t* __restrict__ arg, int64_t ld_arg, int nq,
                                                             int nv, int L, int lpad, int hidden, int combine, int tq) {
  using Cfg = GemmCfg<T, 128, 128, 2, 2>;
  __shared__ __attribute__((aligned(16))) char smem[Cfg::LDS_BYTES];
  const int qt = blockIdx.x % tq, ct = blockIdx.x / tq;
  const int vpt = 128 / lpad;
  const int cols = vpt * lpad;
  const int q0 = qt * 128, v0 = ct * vpt;

  f32x4 acc[Cfg::MT][Cfg::NT];
  auto a_row = [&](int r) -> const char* {
    return (q0 + r) < nq ? reinterpret_cast<const char*>(qn + (int64_t)(q0 + r) * hidden) : nullptr;
  };
  auto b_row = [&](int r) -> const char* {
    const int v = r / lpad, l = r - v * lpad;
    return (r < cols && v0 + v < nv && l < L) ? reinterpret_cast<const char*>(cn + ((int64_t)(v0 + v) * L + l) * hidden)
                                             : nullptr;
  };
  gemm_mainloop<T, Cfg>(acc, a_row, b_row, hidden * (int)sizeof(T), smem);

  float* red_v = reinterpret_cast<float*>(smem);            // [128 rows][8 column tiles]
  int* red_c = reinterpret_cast<int*>(smem) + 128 * 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int fr = lane & 15, fg = lane >> 4;
  __syncthreads();                                          // (the mainloop's last LDS reads)
#pragma unroll
  for (int nt = 0; nt < Cfg::NT; ++nt) {
    const int col = wn * 64 + nt * 16 + fr;
    const int v = col / lpad, l = col - v * lpad;
    const float m = (col < cols && v0 + v < nv && l < L) ? mask[(int64_t)(v0 + v) * L + l] : 0.f;
    const float fill = (1.f - m) * -1e10f;
#pragma unroll
    for (int mt = 0; mt < Cfg::MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sv = acc[mt][nt][r] * m + fill;         // mask_logits, xml/model_xml.py:640-641
        const float mx = lane16_max_dpp(sv);
        const int c = lane16_min_int_dpp(sv == mx ? col : 0x7fffffff);
        if (fr == 0) {
          const int i = (wm * 64 + mt * 16 + fg * 4 + r) * 8 + wn * 4 + nt;
          red_v[i] = mx;
          red_c[i] = c;
        }
      }
    }
  }
  __syncthreads();
  const int tiles_per_video = lpad >> 4;
  for (int i = threadIdx.x; i < 128 * vpt; i += 256) {
    const int row = i % 128, v = i / 128;
    if (q0 + row >= nq || v0 + v >= nv) continue;
    float mx = -INFINITY;
    int mc = 0;
    for (int t = 0; t < tiles_per_video; ++t) {
      const float x = red_v[row * 8 + v * tiles_per_video + t];
      if (x > mx) { mx = x; mc = red_c[row * 8 + v * tiles_per_video + t]; }
    }
    mc -= v * lpad;
    // every clip masked out: all columns tie at -1e10 (or the padding columns at -1e10 beat nothing): the first clip, as
    // torch.max picks; a padding column (l >= L) can only win such a tie and never comes first
    float* po = out + (int64_t)(q0 + row) * ld_out + v0 + v;
    *po = combine ? (*po + mx) * 0.5f : mx;
    arg[(int64_t)(q0 + row) * ld_arg + v0 + v] = mc < L ? mc : 0;
  }
}

// In-order compaction of the non-zero entries of g[0 .. count) (stride `stride`) into s_idx / s_val (capacity cap, checked by
// the caller: count <= cap).  Returns the number of entries; all threads see it after the trailing barrier.
__device__ __forceinline__ int collect_active(const float* __restrict__ g, int64_t stride, int count, float scale, int* s_idx,
                                              float* s_val, int* s_cnt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int base = 0;
  for (int i0 = 0; i0 < count; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const float v = i < count ? g[(int64_t)i * stride] * scale : 0.f;
    const unsigned long long b = __ballot(v != 0.f);
    if (lane == 0) s_cnt[wave] = __popcll(b);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += s_cnt[w];
    if (v != 0.f) {
      const int pos = off + __popcll(b & ((1ull << lane) - 1ull));
      s_idx[pos] = i;
      s_val[pos] = v;
    }
    base += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __syncthreads();
  }
  return base;
}

constexpr int Q2C_BWD_CAP = 1024;      // pairs with a gradient per row / column of dscores (all of them when nq, nv <= 1024)
constexpr int Q2C_BWD_MAXH = 2048;

// (F.normalize backward of a row, as in l2norm_bwd_kernel: dx = (dy - x (x . dy) / den^2) / den, den = max(|x|, eps); when
// the norm is clamped it is a constant and only dy / den survives.)
struct Q2cBwdSet {                      // one modality
  const void *query, *feat, *qn, *cn;
  const float* mask;
  void *dq, *dfeat;
  const int32_t* arg;
  int L, lpad;
};
struct Q2cBwdArgs {
  Q2cBwdSet s[2];                       // blockIdx.y picks the modality
  const float* dscores;
  int64_t ld_ds, ld_arg;
  float scale, eps;
  int nq, nv, hidden;
};

template <typename T>
__global__ __launch_bounds__(256) void q2c_l2_bwd_kernel(Q2cBwdArgs a) {
  const Q2cBwdSet& set = a.s[blockIdx.y];
  const T* __restrict__ query = (const T*)set.query;
  const T* __restrict__ feat = (const T*)set.feat;
  const T* __restrict__ qn = (const T*)set.qn;
  const T* __restrict__ cn = (const T*)set.cn;
  const float* __restrict__ mask = set.mask;
  const float* __restrict__ dscores = a.dscores;
  T* __restrict__ dq = (T*)set.dq;
  T* __restrict__ dfeat = (T*)set.dfeat;
  const int32_t* __restrict__ arg = set.arg;
  const int64_t ld_ds = a.ld_ds, ld_arg = a.ld_arg;
  const float scale = a.scale, eps = a.eps;
  const int nq = a.nq, nv = a.nv, L = set.L, lpad = set.lpad, hidden = a.hidden;
  __shared__ int s_idx[Q2C_BWD_CAP];
  __shared__ float s_val[Q2C_BWD_CAP];
  __shared__ int s_l[Q2C_BWD_CAP];
  __shared__ float s_best[4], s_red[8];
  __shared__ int s_bl[4], s_cnt[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunks = hidden >> 3;
  if ((int)blockIdx.x < nq) {
    // ---- query side: dqn[m] = sum_v g[m][v] mask cn[v][l*];  dq[m] = normalize'(query[m]) dqn[m] ----
    const int m = blockIdx.x;
    const int cnt = collect_active(dscores + (int64_t)m * ld_ds, 1, nv, scale, s_idx, s_val, s_cnt);
    const bool own = tid < chunks;                      // thread t owns chunk t (hidden <= 2048)
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const T* q = qn + (int64_t)m * hidden;
    for (int j = 0; j < cnt; ++j) {
      const int n = s_idx[j];
      const T* cbase = cn + (int64_t)n * lpad * hidden;
      const int bl = arg ? arg[(int64_t)m * ld_arg + n]          // kept by the forward pass (xml_q2c_scores_arg)
                         : pair_argmax<T>(q, cbase, mask + (int64_t)n * lpad, L, hidden, s_best, s_bl);
      const float gm = s_val[j] * mask[(int64_t)n * lpad + bl];
      if (gm != 0.f && own) {
        float cv[8];
        ld8<T>(cbase + (int64_t)bl * hidden + tid * 8, cv);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += gm * cv[e];
      }
    }
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (own) ld8<T>(query + (int64_t)m * hidden + tid * 8, x);
    float ss = 0.f, xd = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ss += x[e] * x[e]; xd += x[e] * acc[e]; }
    ss = wave_sum(ss);
    xd = wave_sum(xd);
    __syncthreads();
    if (lane == 0) { s_red[wave] = ss; s_red[4 + wave] = xd; }
    __syncthreads();
    ss = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    xd = s_red[4] + s_red[5] + s_red[6] + s_red[7];
    const float nrm = sqrtf(ss), den = fmaxf(nrm, eps);
    if (own) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = nrm > eps ? (acc[e] - x[e] * xd / (den * den)) / den : acc[e] / den;
      st8<T>(dq + (int64_t)m * hidden + tid * 8, o);
    }
    return;
  }
  // ---- context side: video n; the pairs of column n each send their gradient to ONE clip row; every other row is zero ----
  const int n = blockIdx.x - nq;
  const int cnt = collect_active(dscores + n, ld_ds, nq, scale, s_idx, s_val, s_cnt);
  const T* cbase = cn + (int64_t)n * lpad * hidden;
  const float* mrow = mask + (int64_t)n * lpad;
  for (int j = 0; j < cnt; ++j) {
    const int bl = arg ? arg[(int64_t)s_idx[j] * ld_arg + n]
                       : pair_argmax<T>(qn + (int64_t)s_idx[j] * hidden, cbase, mrow, L, hidden, s_best, s_bl);
    if (tid == 0) {
      s_l[j] = bl;
      s_val[j] *= mrow[bl];
    }
  }
  __syncthreads();
  for (int l = wave; l < L; l += 4) {                   // one wave per clip row; lanes own chunks lane, lane + 64, ...
    T* out = dfeat + ((int64_t)n * L + l) * hidden;
    bool any = false;
    for (int j = 0; j < cnt; ++j) any = any || (s_l[j] == l && s_val[j] != 0.f);
    if (!any) {
      const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int c = lane; c < chunks; c += 64) st8<T>(out + c * 8, z);
      continue;
    }
    float acc[4][8], x[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane + 64 * k;
#pragma unroll
      for (int e = 0; e < 8; ++e) { acc[k][e] = 0.f; x[k][e] = 0.f; }
      if (c < chunks) ld8<T>(feat + ((int64_t)n * L + l) * hidden + c * 8, x[k]);
    }
    for (int j = 0; j < cnt; ++j) {
      if (s_l[j] != l || s_val[j] == 0.f) continue;
      const float gm = s_val[j];
      const T* q = qn + (int64_t)s_idx[j] * hidden;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = lane + 64 * k;
        if (c < chunks) {
          float qv[8];
          ld8<T>(q + c * 8, qv);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[k][e] += gm * qv[e];
        }
      }
    }
    float ss = 0.f, xd = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) { ss += x[k][e] * x[k][e]; xd += x[k][e] * acc[k][e]; }
    ss = wave_sum(ss);
    xd = wave_sum(xd);
    const float nrm = sqrtf(ss), den = fmaxf(nrm, eps);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane + 64 * k;
      if (c < chunks) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = nrm > eps ? (acc[k][e] - x[k][e] * xd / (den * den)) / den : acc[k][e] / den;
        st8<T>(out + c * 8, o);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// modular query pooling backward, 16-byte version (scalar form + maths: train.hip modular_pool_bwd_kernel)
//   a = softmax_l(mask_logits(enc w_m)),  mq[m] = sum_l a[l][m] enc[l]
// One block per query.  w_m and dout[m][q] are staged in LDS as f32; phase 1: a wave per clip, four clips in flight;
// phase 3: a thread owns an 8-element chunk of the hidden dimension and every G-th clip (G = 256 / chunks groups), four
// clips in flight; dw_m meets in LDS first, one global atomic per element and block.  hidden % 8 == 0, hidden <= 2048.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void modular_pool_bwd_vec_kernel(const T* __restrict__ enc, const float* __restrict__ mask,
                                                                   const float* __restrict__ wm, const T* __restrict__ dout,
                                                                   T* __restrict__ denc, float* __restrict__ dwm, int64_t n,
                                                                   int lq, int hidden, int n_mod) {
  __shared__ float s_att[2][128], s_da[2][128], s_dsc[2][128];
  extern __shared__ float s_dyn[];                      // [n_mod][hidden] w, [n_mod][hidden] dout, [n_mod][hidden] dw
  float* s_w = s_dyn;
  float* s_dm = s_dyn + n_mod * hidden;
  float* s_dw = s_dyn + 2 * n_mod * hidden;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunks = hidden >> 3;
  const T* e = enc + (int64_t)q * lq * hidden;
  for (int i = tid; i < n_mod * hidden; i += 256) {
    const int m = i / hidden, h = i - m * hidden;
    s_w[i] = wm[i];
    s_dm[i] = DT<T>::ld(dout + ((int64_t)m * n + q) * hidden + h);
    s_dw[i] = 0.f;
  }
  __syncthreads();
  for (int l0 = wave; l0 < lq; l0 += 16) {
    float s[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, d[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int c = lane; c < chunks; c += 64) {
      float ev[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) ld8<T>(e + (int64_t)min(l0 + 4 * u, lq - 1) * hidden + c * 8, ev[u]);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (m < n_mod) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float w = s_w[m * hidden + c * 8 + k], dm = s_dm[m * hidden + c * 8 + k];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s[m][u] += ev[u][k] * w; d[m][u] += ev[u][k] * dm; }
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (m < n_mod) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float sv = wave_sum(s[m][u]), dv = wave_sum(d[m][u]);
          const int l = l0 + 4 * u;
          if (lane == 0 && l < lq) {
            const float mk = mask[(int64_t)q * lq + l];
            s_att[m][l] = sv * mk + (1.f - mk) * -1e10f;
            s_da[m][l] = dv;
          }
        }
      }
    }
  }
  __syncthreads();
  if (wave < n_mod) {                                   // softmax over the clips and its backward: one wave per modular vector
    const int m = wave;
    float v[2], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 2; ++k) { v[k] = lane + 64 * k < lq ? s_att[m][lane + 64 * k] : -INFINITY; mx = fmaxf(mx, v[k]); }
    for (int off = 32; off; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) { v[k] = lane + 64 * k < lq ? expf(v[k] - mx) : 0.f; sum += v[k]; }
    sum = wave_sum(sum);
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) { v[k] /= sum; if (lane + 64 * k < lq) dot += v[k] * s_da[m][lane + 64 * k]; }
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int l = lane + 64 * k;
      if (l < lq) {
        s_att[m][l] = v[k];
        s_dsc[m][l] = v[k] * (s_da[m][l] - dot) * mask[(int64_t)q * lq + l];
      }
    }
  }
  __syncthreads();
  const int groups = 256 / chunks;                      // >= 1 (hidden <= 2048)
  const int c = tid % chunks, grp = tid / chunks;
  if (grp < groups) {
    float w[2][8], dm[2][8], dw[2][8];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        w[m][k] = m < n_mod ? s_w[m * hidden + c * 8 + k] : 0.f;
        dm[m][k] = m < n_mod ? s_dm[m * hidden + c * 8 + k] : 0.f;
        dw[m][k] = 0.f;
      }
    for (int l0 = grp; l0 < lq; l0 += 4 * groups) {
      float ev[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) ld8<T>(e + (int64_t)min(l0 + u * groups, lq - 1) * hidden + c * 8, ev[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int l = l0 + u * groups;
        if (l >= lq) break;
        float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          if (m < n_mod) {
            const float a = s_att[m][l], ds = s_dsc[m][l];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              g[k] += a * dm[m][k] + ds * w[m][k];
              dw[m][k] += ds * ev[u][k];
            }
          }
        }
        st8<T>(denc + ((int64_t)q * lq + l) * hidden + c * 8, g);
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
      if (m < n_mod)
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&s_dw[m * hidden + c * 8 + k], dw[m][k]);
  }
  __syncthreads();
  for (int i = tid; i < n_mod * hidden; i += 256) unsafeAtomicAdd(dwm + i, s_dw[i]);
}

// parts4 = {w0 a, w1 r[0], w2 r[1], their sum}, overall = the sum; a / r may be NULL (term switched off: 0)
__global__ void loss_combine_kernel(const float* __restrict__ a, const float* __restrict__ r, float w0, float w1, float w2,
                                    float* __restrict__ parts4, float* __restrict__ overall) {
  if (threadIdx.x) return;
  const float x0 = a ? w0 * a[0] : 0.f, x1 = r ? w1 * r[0] : 0.f, x2 = r ? w2 * r[1] : 0.f;
  const float sum = x0 + x1 + x2;
  parts4[0] = x0; parts4[1] = x1; parts4[2] = x2; parts4[3] = sum;
  overall[0] = sum;
}
// gradients of the inputs from the gradient g of the SUM (out4[3]): da = w0 g, dr = {w1 g, w2 g}
__global__ void loss_combine_bwd_kernel(const float* __restrict__ g, float w0, float w1, float w2, float* __restrict__ da,
                                        float* __restrict__ dr) {
  if (threadIdx.x) return;
  if (da) da[0] = w0 * g[0];
  if (dr) { dr[0] = w1 * g[0]; dr[1] = w2 * g[0]; }
}

}  // namespace

extern "C" int xml_q2c_scores_l2norm_bwd_supported(int nq, int nv, int l, int hidden, int dt) {
  return (dt == XML_F32 || dt == XML_BF16) && nq > 0 && nv > 0 && nq <= Q2C_BWD_CAP && nv <= Q2C_BWD_CAP && l > 0 &&
         hidden % 8 == 0 && hidden > 0 && hidden <= Q2C_BWD_MAXH;
}

static int q2c_l2_bwd_launch(int n_mod, const void* const* query, const void* const* feat, const void* const* qn,
                             const void* const* cn, const float* const* mask, const float* dscores, int64_t ld_ds, float scale,
                             void* const* dq, void* const* dfeat, int nq, int nv, const int* l, const int* lpad, int hidden,
                             const int32_t* const* arg, int64_t ld_arg, int dt, hipStream_t st) {
  if (n_mod < 1 || n_mod > 2 || !dscores || ld_ds < nv) return XML_ERR_BAD_ARG;
  Q2cBwdArgs a;
  for (int m = 0; m < n_mod; ++m) {
    if (!query[m] || !feat[m] || !qn[m] || !cn[m] || !mask[m] || !dq[m] || !dfeat[m] || lpad[m] < l[m]) return XML_ERR_BAD_ARG;
    if (arg && arg[m] && ld_arg < nv) return XML_ERR_BAD_ARG;
    if (!xml_q2c_scores_l2norm_bwd_supported(nq, nv, l[m], hidden, dt)) return XML_ERR_UNSUPPORTED;
    a.s[m] = Q2cBwdSet{query[m], feat[m], qn[m], cn[m], mask[m], dq[m], dfeat[m], arg ? arg[m] : nullptr, l[m], lpad[m]};
  }
  if (n_mod == 1) a.s[1] = a.s[0];
  a.dscores = dscores; a.ld_ds = ld_ds; a.ld_arg = ld_arg; a.scale = scale; a.eps = 1e-12f;
  a.nq = nq; a.nv = nv; a.hidden = hidden;
  const dim3 grid(nq + nv, n_mod), blk(256);
  if (dt == XML_F32) hipLaunchKernelGGL(q2c_l2_bwd_kernel<float>, grid, blk, 0, st, a);
  else hipLaunchKernelGGL(q2c_l2_bwd_kernel<bf16_t>, grid, blk, 0, st, a);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" int xml_q2c_scores_l2norm_bwd(const void* query, const void* feat, const void* qn, const void* cn,
                                         const float* mask, const float* dscores, int64_t ld_ds, float scale, void* dq,
                                         void* dfeat, int nq, int nv, int l, int lpad, int hidden, const int32_t* arg,
                                         int64_t ld_arg, int dt, xml_stream_t stream) {
  XML_ENTER();
  return q2c_l2_bwd_launch(1, &query, &feat, &qn, &cn, &mask, dscores, ld_ds, scale, &dq, &dfeat, nq, nv, &l, &lpad, hidden,
                           &arg, ld_arg, dt, (hipStream_t)stream);
}

extern "C" int xml_q2c_scores_l2norm_bwd_multi(int n_mod, const void* const* query, const void* const* feat,
                                               const void* const* qn, const void* const* cn, const float* const* mask,
                                               const float* dscores, int64_t ld_ds, float scale, void* const* dq,
                                               void* const* dfeat, int nq, int nv, const int* l, const int* lpad, int hidden,
                                               const int32_t* const* arg, int64_t ld_arg, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!query || !feat || !qn || !cn || !mask || !dq || !dfeat || !l || !lpad) return XML_ERR_BAD_ARG;
  return q2c_l2_bwd_launch(n_mod, query, feat, qn, cn, mask, dscores, ld_ds, scale, dq, dfeat, nq, nv, l, lpad, hidden, arg,
                           ld_arg, dt, (hipStream_t)stream);
}

extern "C" int xml_q2c_scores_arg(const void* qn, const void* cn, const float* mask, float* out, int64_t ld_out,
                                  int32_t* arg, int64_t ld_arg, int nq, int nv, int l, int hidden, int combine, int dt,
                                  xml_stream_t stream) {
  XML_ENTER();
  if (!qn || !cn || !mask || !out || !arg || nq <= 0 || nv <= 0 || l <= 0 || hidden <= 0 || ld_out < nv || ld_arg < nv)
    return XML_ERR_BAD_ARG;
  if (l > 128 || hidden % 8) return XML_ERR_UNSUPPORTED;
  const int lpad = (l + 15) / 16 * 16, vpt = 128 / lpad;
  const int tq = cdiv(nq, 128), tc = cdiv(nv, vpt);
  const dim3 grid((unsigned)((int64_t)tq * tc)), blk(256);
  if (dt == XML_F32)
    hipLaunchKernelGGL(q2c_scores_arg_kernel<float>, grid, blk, 0, (hipStream_t)stream, (const float*)qn, (const float*)cn,
                       mask, out, ld_out, arg, ld_arg, nq, nv, l, lpad, hidden, combine, tq);
  else if (dt == XML_BF16)
    hipLaunchKernelGGL(q2c_scores_arg_kernel<bf16_t>, grid, blk, 0, (hipStream_t)stream, (const bf16_t*)qn,
                       (const bf16_t*)cn, mask, out, ld_out, arg, ld_arg, nq, nv, l, lpad, hidden, combine, tq);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// (called by xml_modular_pool_bwd, train.hip, for the shapes it serves)
int xmli_modular_pool_bwd_vec(const void* enc, const float* mask, const float* w_m, const void* dout, void* denc, float* dw_m,
                              int64_t n, int lq, int hidden, int n_mod, int dt, hipStream_t st) {
  if (hidden % 8 || hidden > 2048 || lq > 128 || n_mod < 1 || n_mod > 2 || (dt != XML_F32 && dt != XML_BF16)) return -1;
  const size_t lds = (size_t)3 * n_mod * hidden * 4;
  if (dt == XML_F32)
    hipLaunchKernelGGL(modular_pool_bwd_vec_kernel<float>, dim3((unsigned)n), dim3(256), lds, st, (const float*)enc, mask,
                       w_m, (const float*)dout, (float*)denc, dw_m, n, lq, hidden, n_mod);
  else
    hipLaunchKernelGGL(modular_pool_bwd_vec_kernel<bf16_t>, dim3((unsigned)n), dim3(256), lds, st, (const bf16_t*)enc, mask,
                       w_m, (const bf16_t*)dout, (bf16_t*)denc, dw_m, n, lq, hidden, n_mod);
  return 0;
}

extern "C" int xml_loss_combine(const float* st_ed, const float* rank2, float w_st_ed, float w_neg_ctx, float w_neg_q,
                                float* parts4, float* overall, xml_stream_t stream) {
  XML_ENTER();
  if (!parts4 || !overall) return XML_ERR_BAD_ARG;
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, st_ed, rank2, w_st_ed, w_neg_ctx,
                     w_neg_q, parts4, overall);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" int xml_loss_combine_bwd(const float* g, float w_st_ed, float w_neg_ctx, float w_neg_q, float* d_st_ed,
                                    float* d_rank2, xml_stream_t stream) {
  XML_ENTER();
  if (!g) return XML_ERR_BAD_ARG;
  hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g, w_st_ed, w_neg_ctx, w_neg_q,
                     d_st_ed, d_rank2);
  XML_CHECK_LAUNCH();
  return XML_OK;
}
