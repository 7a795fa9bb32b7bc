// Projections and row-wise normalisations of the XML encoders (K1, K2, K4 and the query linears).
//   reference: LinearLayer.forward           xml/model_components.py:156-163
//              TrainablePositionalEncoding   xml/model_components.py:76-89
//              BertSelfOutput.forward        xml/model_components.py:313-317
#include "gemm.h"
#include "internal.h"
#include "l2norm.h"

// ---------------------------------------------------------------------------------------------------
// GEMM + epilogue:  out[m][n] = act(acc + bias[n]) + addend      (OutT = float when a LayerNorm follows)
//   add_mode 0: none; 1: addend[(m % seq_len)][n] (positional table); 2: addend[m][n] (residual)
// ---------------------------------------------------------------------------------------------------
template <typename T, typename OutT, typename AddT, int BMN = 128>
__global__ __launch_bounds__(256) void gemm_bias_act_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                            const float* __restrict__ bias,
                                                            const AddT* __restrict__ addend, OutT* __restrict__ out,
                                                            int64_t M, int N, int K, int relu, int add_mode,
                                                            int seq_len, const float* __restrict__ row_scale) {
  // T == f16_t: split-f16 projection on K-concatenated halves (split16.hip); accumulator rows are scaled by row_scale[m]
  using Cfg = GemmCfg<T, BMN, BMN, 2, 2>;
  __shared__ __attribute__((aligned(16))) char smem[Cfg::LDS_BYTES];
  const int64_t m0 = (int64_t)blockIdx.y * Cfg::BM;
  const int n0 = blockIdx.x * Cfg::BN;
  f32x4 acc[Cfg::MT][Cfg::NT];
  auto a_row = [&](int r) -> const char* {
    const int64_t m = m0 + r;
    return m < M ? reinterpret_cast<const char*>(A + m * K) : nullptr;
  };
  auto b_row = [&](int r) -> const char* {
    const int n = n0 + r;
    return n < N ? reinterpret_cast<const char*>(W + (int64_t)n * K) : nullptr;
  };
  if constexpr (BMN < 128 && !IsSplit16<T>::value && !std::is_same<T, f16_t>::value) {
    // few rows: the K loop is a chain of memory round trips -- loads 3 (or 2) steps ahead (gemm_mainloop_deep)
    const int nsteps = (K * (int)sizeof(T) + Cfg::ROWB - 1) / Cfg::ROWB;
    if (nsteps % 3 == 0) gemm_mainloop_deep<T, Cfg, 3>(acc, a_row, b_row, K * (int)sizeof(T), smem);
    else if (nsteps % 2 == 0) gemm_mainloop_deep<T, Cfg, 2>(acc, a_row, b_row, K * (int)sizeof(T), smem);
    else gemm_mainloop<T, Cfg>(acc, a_row, b_row, K * (int)sizeof(T), smem);
  } else {
    gemm_mainloop<T, Cfg>(acc, a_row, b_row, K * (int)sizeof(T), smem);
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
#pragma unroll
  for (int mt = 0; mt < Cfg::MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < Cfg::NT; ++nt) {
      const int n = n0 + wn * (Cfg::BN / Cfg::WN) + nt * 16 + (lane & 15);
      if (n >= N) continue;
      const float b = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + wm * (Cfg::BM / Cfg::WM) + mt * 16 + (lane >> 4) * 4 + r;
        if (m >= M) continue;
        float v = acc[mt][nt][r];
        if constexpr (std::is_same<T, f16_t>::value) v *= row_scale[m];
        v += b;
        if (relu) v = fmaxf(v, 0.f);
        if (add_mode == 1) v += DT<AddT>::ld(addend + (int64_t)(m % seq_len) * N + n);
        else if (add_mode == 2) v += DT<AddT>::ld(addend + m * N + n);
        DT<OutT>::st(out + m * N + n, v);
      }
    }
  }
}

template <typename T, typename OutT, typename AddT>
static int launch_gemm(const void* A, const void* W, const float* bias, const void* addend, void* out, int64_t M,
                       int N, int K, int relu, int add_mode, int seq_len, hipStream_t st,
                       const float* row_scale = nullptr) {
  // Few rows (the pooled-query linears of a 128-pair training batch, a 50-query batch's encoder): 128 x 128 tiles leave most
  // of the chip idle behind a serial K loop (128 x 768 x 768: 6 workgroups, 32 us).  Smaller tiles of the SAME mainloop -- every
  // output element sees the same MFMA sequence over K, the results are bit-identical -- until a few hundred workgroups exist
  // (the small tiles also take the deep-prefetch K loop, gemm_mainloop_deep).
  const int64_t wg128 = (int64_t)cdiv(N, 128) * cdiv(M, 128);
  const int bmn = wg128 >= 512 ? 128 : wg128 * 4 >= 192 ? 64 : 32;
  dim3 grid(cdiv(N, bmn), cdiv(M, bmn));
  if (bmn == 128)
    hipLaunchKernelGGL((gemm_bias_act_kernel<T, OutT, AddT, 128>), grid, dim3(256), 0, st, (const T*)A, (const T*)W, bias,
                       (const AddT*)addend, (OutT*)out, M, N, K, relu, add_mode, seq_len, row_scale);
  else if (bmn == 64)
    hipLaunchKernelGGL((gemm_bias_act_kernel<T, OutT, AddT, 64>), grid, dim3(256), 0, st, (const T*)A, (const T*)W, bias,
                       (const AddT*)addend, (OutT*)out, M, N, K, relu, add_mode, seq_len, row_scale);
  else
    hipLaunchKernelGGL((gemm_bias_act_kernel<T, OutT, AddT, 32>), grid, dim3(256), 0, st, (const T*)A, (const T*)W, bias,
                       (const AddT*)addend, (OutT*)out, M, N, K, relu, add_mode, seq_len, row_scale);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

bool xmli_gemm256_eligible(int64_t M, int N, int K, int dt);
int xmli_gemm256(const void* A, const void* W, const float* bias, const void* addend, void* out, int64_t M, int N,
                 int K, int relu, int add_mode, int seq_len, int out_f32, int dt, hipStream_t st);
bool xmli_gemm256p_eligible(int64_t M, int N, int K, int dt);
int xmli_gemm256p(const void* A, const void* W, const float* bias, const void* addend, void* out, int64_t M, int N,
                  int K, int relu, int add_mode, int seq_len, int out_f32, int dt, hipStream_t st);
#ifdef XML_DEBUG_VARIANTS
int g_gemm_variant = 0;
extern "C" void xml_debug_set_gemm_variant(int v) { g_gemm_variant = v; }
#endif

bool xmli_gemm256_f16s_eligible(int64_t M, int N, int K3);
int xmli_gemm256_f16s(const void* A, const void* W, const float* bias, const void* addend, void* out,
                      const float* row_scale, int64_t M, int N, int K3, int relu, int add_mode, int seq_len,
                      hipStream_t st);
int xmli_split_f16_kcat(const float* x, void* a_cat, float* inv_scale, const float* w_trailer, int64_t rows, int k,
                        hipStream_t st);

// scratch of one split-f16 projection: A' (M, 3K) f16 + the row scales
size_t xmli_gemm_split_ws_bytes(int64_t M, int K, int dt) {
  if (dt != XML_F16S) return 0;
  return align_up((size_t)M * K * 6, 256) + align_up((size_t)M * 4, 256);
}

// out_f32: write f32 regardless of dt (pre-LayerNorm values keep full precision)
// dt == XML_F16S: A / addend / out are f32, W is the packed split weight (xml_pack_weights_f16s), split_ws holds
// xmli_gemm_split_ws_bytes(M, K, dt) bytes: the A operand is split per row into [hi | lo | hi] halves and the f16 GEMM
// runs over K' = 3 K -- f32-grade results (2^-22 relative per product) at 16/3 of the f32 MFMA rate.
int xmli_gemm(const void* A, const void* W, const float* bias, const void* addend, void* out, int64_t M, int N,
              int K, int relu, int add_mode, int seq_len, int out_f32, int dt, hipStream_t st, void* split_ws) {
  XML_ENTER();
  if (M <= 0 || N <= 0 || K <= 0) return XML_ERR_BAD_ARG;
  if (dt == XML_F16S) {
    if (!split_ws) return XML_ERR_WORKSPACE;
    if (K % 8 || N % 8) return XML_ERR_UNSUPPORTED;
    char* a_cat = (char*)split_ws;
    float* rs = reinterpret_cast<float*>(a_cat + align_up((size_t)M * K * 6, 256));
    const float* trailer = reinterpret_cast<const float*>((const char*)W + align_up((size_t)N * K * 6, 16));
    const int rc = xmli_split_f16_kcat((const float*)A, a_cat, rs, trailer, M, K, st);
    if (rc) return rc;
    if (xmli_gemm256_f16s_eligible(M, N, 3 * K))
      return xmli_gemm256_f16s(a_cat, W, bias, addend, out, rs, M, N, 3 * K, relu, add_mode, seq_len, st);
    return launch_gemm<f16_t, float, float>(a_cat, W, bias, addend, out, M, N, 3 * K, relu, add_mode, seq_len, st, rs);
  }
  if (dt != XML_F32 && dt != XML_BF16) return XML_ERR_BAD_ARG;
  // fewer than 64 tiles of 256 x 256 (a 50-query batch's encoder, the query branch of a training step): a quarter of the
  // chip at best behind a serial K loop -- the small-tile, deep-prefetch form of the register-staged kernel instead
  // (3 840 x 768 x 768: 28 -> 11 us)
  const bool few = (int64_t)cdiv(M, 256) * cdiv(N, 256) < 64 && g_gemm_variant != 3;
  if (!few && g_gemm_variant == 0 && xmli_gemm256p_eligible(M, N, K, dt))
    return xmli_gemm256p(A, W, bias, addend, out, M, N, K, relu, add_mode, seq_len, out_f32, dt, st);
  if (!few && g_gemm_variant != 1 && xmli_gemm256_eligible(M, N, K, dt))
    return xmli_gemm256(A, W, bias, addend, out, M, N, K, relu, add_mode, seq_len, out_f32, dt, st);
  if (dt == XML_F32) {
    if (K % 4) return XML_ERR_UNSUPPORTED;
    return launch_gemm<float, float, float>(A, W, bias, addend, out, M, N, K, relu, add_mode, seq_len, st);
  } else if (dt == XML_BF16) {
    if (K % 8) return XML_ERR_UNSUPPORTED;
    if (out_f32) return launch_gemm<bf16_t, float, bf16_t>(A, W, bias, addend, out, M, N, K, relu, add_mode, seq_len, st);
    return launch_gemm<bf16_t, bf16_t, bf16_t>(A, W, bias, addend, out, M, N, K, relu, add_mode, seq_len, st);
  }
  return XML_ERR_BAD_ARG;
}

// ---------------------------------------------------------------------------------------------------
// Row-wise LayerNorm of (a [+ b]); one wave per row, f32 statistics (two-pass mean / variance).
// ---------------------------------------------------------------------------------------------------
template <typename InT, typename BT, typename OutT>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const InT* __restrict__ a, const BT* __restrict__ b,
                                                            const float* __restrict__ g,
                                                            const float* __restrict__ beta, OutT* __restrict__ y,
                                                            int64_t rows, int d, int ld_out, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const InT* pa = a + row * d;
  const BT* pb = b ? b + row * d : nullptr;
  OutT* py = y + row * ld_out;
  if (d <= 64 * 16) {
    // the row stays in registers (<= 16 values per lane): one read of the inputs instead of three
    float x[16];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = lane + k * 64;
      x[k] = (i < d) ? DT<InT>::ld(pa + i) + (pb ? DT<BT>::ld(pb + i) : 0.f) : 0.f;
      s += x[k];
    }
    const float mean = wave_sum(s) / (float)d;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float c = (lane + k * 64 < d) ? x[k] - mean : 0.f;
      v += c * c;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)d + eps);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = lane + k * 64;
      if (i < d) DT<OutT>::st(py + i, (x[k] - mean) * rstd * g[i] + beta[i]);
    }
  } else {
    float s = 0.f;
    for (int i = lane; i < d; i += 64) s += DT<InT>::ld(pa + i) + (pb ? DT<BT>::ld(pb + i) : 0.f);
    const float mean = wave_sum(s) / (float)d;
    float v = 0.f;
    for (int i = lane; i < d; i += 64) {
      const float x = DT<InT>::ld(pa + i) + (pb ? DT<BT>::ld(pb + i) : 0.f) - mean;
      v += x * x;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)d + eps);
    for (int i = lane; i < d; i += 64) {
      const float x = DT<InT>::ld(pa + i) + (pb ? DT<BT>::ld(pb + i) : 0.f);
      DT<OutT>::st(py + i, (x - mean) * rstd * g[i] + beta[i]);
    }
  }
  for (int i = d + lane; i < ld_out; i += 64) DT<OutT>::st(py + i, 0.f);  // zero K padding
}

// Vectorised variant (d % 8 == 0, d <= 64 * 8 * VPL): lane owns the 8-element vectors lane + 64 k, the row stays in
// registers, inputs are read once with 16-byte loads.  VPL = 2 covers the hidden size (<= 1024), 6 / 8 the raw
// feature rows of the input LayerNorm (3072 / 4096).
// DROP (training, xml_add_layernorm_drop): y = drop_out( LN( drop_in(a) + b ) * g + beta ) with xml_dropout's masks -- element
// i of `a` / of `y` is kept when drop_hash(i, seed) >= thresh -- applied in the loads / stores instead of by two elementwise
// passes (22 dropout launches per training step at the configs[4] shape).
template <typename InT, typename BT, typename OutT, int VPL, bool DROP = false>
__global__ __launch_bounds__(256) void add_layernorm_vec_kernel(const InT* __restrict__ a, const BT* __restrict__ b,
                                                                const float* __restrict__ g,
                                                                const float* __restrict__ beta, OutT* __restrict__ y,
                                                                int64_t rows, int d, int ld_out, float eps,
                                                                const int32_t* __restrict__ src_row,
                                                                XmlDropSite din = XmlDropSite{0u, 1.f, 0ull},
                                                                XmlDropSite dout = XmlDropSite{0u, 1.f, 0ull},
                                                                const uint64_t* __restrict__ seed_dev = nullptr) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = d >> 3;
  const InT* pa = a + (src_row ? (int64_t)src_row[row] : row) * d;        // (packed tokens: row i reads source row src_row[i])
  const BT* pb = b ? b + row * d : nullptr;
  OutT* py = y + row * ld_out;
  uint32_t si0 = 0, si1 = 0, so0 = 0, so1 = 0;
  if constexpr (DROP) {
    xml_seed_words(din.seed, seed_dev, si0, si1);
    xml_seed_words(dout.seed, seed_dev, so0, so1);
  }
  float x[VPL * 8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int v = lane + k * 64;
    if (v < nvec) {
      ld8<InT>(pa + v * 8, x + k * 8);
      if constexpr (DROP) {
        if (din.thresh) {
          const uint64_t i0 = (uint64_t)row * (uint64_t)d + (uint64_t)v * 8u;
          drop_mask8(i0, si0, si1, din.thresh, din.scale, x + k * 8);
        }
      }
      if (pb) {
        float t[8];
        ld8<BT>(pb + v * 8, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[k * 8 + j] += t[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[k * 8 + j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[k * 8 + j];
  }
  const float mean = wave_sum(s) / (float)d;
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < VPL; ++k)
    if (lane + k * 64 < nvec)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float c = x[k * 8 + j] - mean; var += c * c; }
  const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)d + eps);
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int v = lane + k * 64;
    if (v < nvec) {
      float gv[8], bv[8], o[8];
      ld8<float>(g + v * 8, gv);
      ld8<float>(beta + v * 8, bv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (x[k * 8 + j] - mean) * rstd * gv[j] + bv[j];
      if constexpr (DROP) {
        if (dout.thresh) {
          const uint64_t i0 = (uint64_t)row * (uint64_t)ld_out + (uint64_t)v * 8u;
          drop_mask8(i0, so0, so1, dout.thresh, dout.scale, o);
        }
      }
      st8<OutT>(py + v * 8, o);
    }
  }
  for (int i = d + lane; i < ld_out; i += 64) DT<OutT>::st(py + i, 0.f);  // zero K padding
}

// training: LayerNorm with the dropout sites before / behind it applied in place (d % 8 == 0, d <= 4096, ld_out == d)
template <typename InT, typename BT, typename OutT>
static int launch_ln_drop(const void* a, const void* b, const float* g, const float* beta, void* y, int64_t rows, int d,
                          XmlDropSite din, XmlDropSite dout, const uint64_t* seed_dev, hipStream_t st) {
  const dim3 grid(cdiv(rows, 4)), blk(256);
#define XML_LN_DROP(VPL)                                                                                                    \
  hipLaunchKernelGGL((add_layernorm_vec_kernel<InT, BT, OutT, VPL, true>), grid, blk, 0, st, (const InT*)a, (const BT*)b, g, \
                     beta, (OutT*)y, rows, d, d, 1e-5f, (const int32_t*)nullptr, din, dout, seed_dev)
  if (d <= 1024) XML_LN_DROP(2);
  else if (d <= 3072) XML_LN_DROP(6);
  else XML_LN_DROP(8);
#undef XML_LN_DROP
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" int xml_add_layernorm_drop(const void* a, int a_dt, const void* b, const float* g, const float* beta, void* y,
                                      int64_t rows, int d, int dt, float p_in, uint64_t seed_in, float p_out,
                                      uint64_t seed_out, const uint64_t* seed_dev, xml_stream_t stream) {
  XML_ENTER();
  if (!a || !g || !beta || !y || rows <= 0 || d <= 0) return XML_ERR_BAD_ARG;
  if (!(p_in >= 0.f) || p_in >= 1.f || !(p_out >= 0.f) || p_out >= 1.f) return XML_ERR_BAD_ARG;
  if (d % 8 || d > 4096) return XML_ERR_UNSUPPORTED;
  const XmlDropSite din = xml_drop_site(p_in, seed_in), dout = xml_drop_site(p_out, seed_out);
  hipStream_t st = (hipStream_t)stream;
  if (dt == XML_F32) {
    if (a_dt != XML_F32) return XML_ERR_BAD_ARG;
    return launch_ln_drop<float, float, float>(a, b, g, beta, y, rows, d, din, dout, seed_dev, st);
  }
  if (dt != XML_BF16) return XML_ERR_BAD_ARG;
  if (a_dt == XML_F32) return launch_ln_drop<float, bf16_t, bf16_t>(a, b, g, beta, y, rows, d, din, dout, seed_dev, st);
  return launch_ln_drop<bf16_t, bf16_t, bf16_t>(a, b, g, beta, y, rows, d, din, dout, seed_dev, st);
}

template <typename InT, typename BT, typename OutT>
static int launch_ln(const void* a, const void* b, const float* g, const float* beta, void* y, int64_t rows, int d,
                     int ld_out, hipStream_t st, const int32_t* src_row = nullptr) {
  const dim3 grid(cdiv(rows, 4)), blk(256);
  const bool vec = (d % 8 == 0) && (ld_out % 8 == 0);     // 16-byte aligned rows on both sides
#define XML_LN_VEC(VPL)                                                                                              \
  hipLaunchKernelGGL((add_layernorm_vec_kernel<InT, BT, OutT, VPL>), grid, blk, 0, st, (const InT*)a, (const BT*)b, g, \
                     beta, (OutT*)y, rows, d, ld_out, 1e-5f, src_row)
  if (src_row && !(vec && d <= 4096)) return XML_ERR_UNSUPPORTED;
  if (vec && d <= 1024) XML_LN_VEC(2);
  else if (vec && d <= 3072) XML_LN_VEC(6);
  else if (vec && d <= 4096) XML_LN_VEC(8);
  else
    hipLaunchKernelGGL((add_layernorm_kernel<InT, BT, OutT>), grid, blk, 0, st, (const InT*)a, (const BT*)b, g, beta,
                       (OutT*)y, rows, d, ld_out, 1e-5f);
#undef XML_LN_VEC
  XML_CHECK_LAUNCH();
  return XML_OK;
}

static int add_layernorm_rows(const void* a, int a_dt, const void* b, const float* g, const float* beta, void* y,
                              int64_t rows, int d, int ld_out, int dt, hipStream_t st, const int32_t* src_row) {
  if (rows <= 0 || d <= 0) return XML_ERR_BAD_ARG;
  if (dt == XML_F32) {
    if (a_dt != XML_F32) return XML_ERR_BAD_ARG;
    return launch_ln<float, float, float>(a, b, g, beta, y, rows, d, ld_out, st, src_row);
  }
  if (a_dt == XML_F32) return launch_ln<float, bf16_t, bf16_t>(a, b, g, beta, y, rows, d, ld_out, st, src_row);
  return launch_ln<bf16_t, bf16_t, bf16_t>(a, b, g, beta, y, rows, d, ld_out, st, src_row);
}

int xmli_add_layernorm(const void* a, int a_dt, const void* b, const float* g, const float* beta, void* y,
                       int64_t rows, int d, int ld_out, int dt, hipStream_t st) {
  XML_ENTER();
  return add_layernorm_rows(a, a_dt, b, g, beta, y, rows, d, ld_out, dt, st, nullptr);
}

extern "C" int xml_add_layernorm(const void* a, int a_dt, const void* b, const float* g, const float* beta,
                                 void* y, int64_t rows, int d, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!a || !g || !beta || !y) return XML_ERR_BAD_ARG;
  return xmli_add_layernorm(a, a_dt, b, g, beta, y, rows, d, d, dt, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------
// F.normalize(x, dim=-1): x / max(||x||_2, 1e-12)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows,
                                                          int d) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* px = x + row * d;
  float s = 0.f;
  for (int i = lane; i < d; i += 64) {
    const float v = DT<T>::ld(px + i);
    s += v * v;
  }
  const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
  for (int i = lane; i < d; i += 64) DT<T>::st(y + row * d + i, DT<T>::ld(px + i) / nrm);
}

// dataset-side normalisation, l2_normalize_np_array (utils/basic_utils.py:82-84): x / (||x||_2 + eps); f32 in,
// f32 out; zero (padding) rows stay zero
__global__ __launch_bounds__(256) void l2norm_add_eps_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int64_t rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* px = x + row * d;
  float s = 0.f;
  for (int i = lane; i < d; i += 64) s += px[i] * px[i];
  const float den = sqrtf(wave_sum(s)) + eps;
  for (int i = lane; i < d; i += 64) y[row * d + i] = px[i] / den;
}

extern "C" int xml_l2norm_rows_eps(const float* x, float* y, int64_t rows, int d, float eps, xml_stream_t stream) {
  XML_ENTER();
  if (!x || !y || rows <= 0 || d <= 0) return XML_ERR_BAD_ARG;
  hipLaunchKernelGGL(l2norm_add_eps_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, rows, d, eps);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// the same with 16-byte loads / stores, half a wave per row (l2norm.h: the arithmetic the fused index build shares)
template <typename T>
__global__ __launch_bounds__(256) void l2norm_rows_vec_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows,
                                                              int d) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int l32 = threadIdx.x & 31;
  const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  const bool live = row < rows;
  uint4 v[L2N_MAXJ];
  const float nrm = l2n_load_row<T>(live ? x + row * d : nullptr, d, l32, v);
  if (!live) return;
  const int chunks = d / VEC;
#pragma unroll
  for (int j = 0; j < L2N_MAXJ; ++j) {
    const int c = l32 + 32 * j;
    if (c < chunks) st_global16(y + row * d + c * VEC, l2n_scale_chunk<T>(v[j], nrm));
  }
}

extern "C" int xml_l2norm_rows(const void* x, void* y, int64_t rows, int d, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!x || !y || rows <= 0 || d <= 0) return XML_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dt == XML_F32 && d % 4 == 0 && d / 4 <= 32 * L2N_MAXJ)
    hipLaunchKernelGGL(l2norm_rows_vec_kernel<float>, dim3(cdiv(rows, 8)), dim3(256), 0, st, (const float*)x, (float*)y, rows, d);
  else if (dt == XML_BF16 && d % 8 == 0 && d / 8 <= 32 * L2N_MAXJ)
    hipLaunchKernelGGL(l2norm_rows_vec_kernel<bf16_t>, dim3(cdiv(rows, 8)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, rows, d);
  else if (dt == XML_F32)
    hipLaunchKernelGGL(l2norm_rows_kernel<float>, dim3(cdiv(rows, 4)), dim3(256), 0, st, (const float*)x, (float*)y, rows, d);
  else if (dt == XML_BF16)
    hipLaunchKernelGGL(l2norm_rows_kernel<bf16_t>, dim3(cdiv(rows, 4)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, rows, d);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------
// dtype conversion
// ---------------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ void convert_kernel(const S* __restrict__ s, D* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    DT<D>::st(d + i, DT<S>::ld(s + i));
}

extern "C" int xml_convert(const void* src, int src_dt, void* dst, int dst_dt, int64_t n, xml_stream_t stream) {
  XML_ENTER();
  if (!src || !dst || n < 0) return XML_ERR_BAD_ARG;
  if (n == 0) return XML_OK;
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  if (src_dt == XML_F32 && dst_dt == XML_BF16)
    hipLaunchKernelGGL((convert_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n);
  else if (src_dt == XML_BF16 && dst_dt == XML_F32)
    hipLaunchKernelGGL((convert_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n);
  else if (src_dt == XML_F32 && dst_dt == XML_F32)
    hipLaunchKernelGGL((convert_kernel<float, float>), dim3(grid), dim3(256), 0, st, (const float*)src, (float*)dst, n);
  else if (src_dt == XML_BF16 && dst_dt == XML_BF16)
    hipLaunchKernelGGL((convert_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" int xml_pack_weights(const float* src, void* dst, int dt, int64_t n, xml_stream_t stream) {
  XML_ENTER();
  return xml_convert(src, XML_F32, dst, dt, n, stream);
}

// ---------------------------------------------------------------------------------------------------
// public: plain linear
// ---------------------------------------------------------------------------------------------------
extern "C" int xml_linear(const void* x, const void* w, const float* b, void* y, int64_t rows, int n, int k, int relu,
                          int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!x || !w || !y) return XML_ERR_BAD_ARG;
  return xmli_gemm(x, w, b, nullptr, y, rows, n, k, relu, 0, 1, 0, dt, (hipStream_t)stream);
}

extern "C" int xml_linear_add(const void* x, const void* w, const float* b, const void* addend, void* y, int64_t rows, int n,
                              int k, int relu, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!x || !w || !y || !addend) return XML_ERR_BAD_ARG;
  if (dt != XML_F32 && dt != XML_BF16) return XML_ERR_UNSUPPORTED;
  return xmli_gemm(x, w, b, addend, y, rows, n, k, relu, 2, 1, 0, dt, (hipStream_t)stream);
}

// y = x W^T + b with a split-f16 weight (xml_pack_weights_f16s): x / y f32, f32-grade results on the 16-bit MFMA pipe
extern "C" size_t xml_linear_f16s_workspace_bytes(int64_t rows, int k) { return xmli_gemm_split_ws_bytes(rows, k, XML_F16S); }
extern "C" int xml_linear_f16s(const float* x, const void* w, const float* b, float* y, int64_t rows, int n, int k, int relu,
                               void* ws, size_t ws_bytes, xml_stream_t stream) {
  XML_ENTER();
  if (!x || !w || !y || !ws) return XML_ERR_BAD_ARG;
  if (rows > 0 && ws_bytes < xmli_gemm_split_ws_bytes(rows, k, XML_F16S)) return XML_ERR_WORKSPACE;
  return xmli_gemm(x, w, b, nullptr, y, rows, n, k, relu, 0, 1, 1, XML_F16S, (hipStream_t)stream, ws);
}

// ---------------------------------------------------------------------------------------------------
// public: K1 + K2
//   ws layout: [ LN_in(x) as dt (rows x d_pad) | pre-LN f32 (rows x hidden) ],  d_pad = d_in rounded up to 8
// ---------------------------------------------------------------------------------------------------
static inline int k_pad8(int d_in) { return (d_in + 7) & ~7; }
// (the second part holds either the f32 pre-LayerNorm rows of the 3-launch path or, when the GEMM takes the LayerNorm in
// its epilogue, that kernel's per-workgroup scratch -- the larger of the two is reserved)
extern "C" size_t xml_linear_ln_relu_pos_workspace_bytes(int64_t rows, int d_in, int hidden, int dt) {
  const size_t pre = align_up((size_t)rows * hidden * 4, 256);
  const size_t lnw = xmli_gemm_ln_eligible(rows, hidden, k_pad8(d_in), dt) ? xmli_gemm_ln_workspace_bytes(rows, hidden) : 0;
  return align_up((size_t)rows * k_pad8(d_in) * dt_size(dt), 256) + (pre > lnw ? pre : lnw) +
         xmli_gemm_split_ws_bytes(rows, k_pad8(d_in), dt);
}

extern "C" int xml_linear_ln_relu_pos(const void* x, int x_dt, const float* ln_in_g, const float* ln_in_b,
                                      const void* w, const float* b, const void* pos, const float* ln_pos_g,
                                      const float* ln_pos_b, void* y, int64_t rows, int seq_len, int d_in, int hidden,
                                      int dt, void* ws, size_t ws_bytes, xml_stream_t stream) {
  XML_ENTER();
  if (!x || !ln_in_g || !ln_in_b || !w || !b || !pos || !ln_pos_g || !ln_pos_b || !y || !ws) return XML_ERR_BAD_ARG;
  if (rows <= 0 || seq_len <= 0 || !xmli_model_dt_ok(dt)) return XML_ERR_BAD_ARG;
  if (d_in <= 0 || hidden % 8) return XML_ERR_UNSUPPORTED;
  if (ws_bytes < xml_linear_ln_relu_pos_workspace_bytes(rows, d_in, hidden, dt)) return XML_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  // TEF inputs (d_in = 3074 / 770, xml/config.py:251-254): LN statistics over d_in, the normalised row is written with
  // zero columns up to d_pad and the weight arrives zero-padded in K alike, so the GEMM sees K = d_pad
  const int d_pad = k_pad8(d_in);
  const int adt = xmli_act_dt(dt);            // storage type of the activations (XML_F16S: f32 rows, split weights)
  char* xn = (char*)ws;
  char* pre = xn + align_up((size_t)rows * d_pad * dt_size(dt), 256);
  char* sws = dt == XML_F16S ? (char*)ws + xml_linear_ln_relu_pos_workspace_bytes(rows, d_in, hidden, dt) -
                                   xmli_gemm_split_ws_bytes(rows, d_pad, dt) : nullptr;
  int rc = xmli_add_layernorm(x, x_dt, nullptr, ln_in_g, ln_in_b, xn, rows, d_in, d_pad, adt, st);
  if (rc) return rc;
  if (xmli_gemm_ln_eligible(rows, hidden, d_pad, dt) &&    // LN_pos in the GEMM epilogue: two launches, no f32 round trip
      xmli_gemm_ln(xn, w, b, pos, ln_pos_g, ln_pos_b, y, rows, hidden, d_pad, /*relu*/ 1, /*add_mode*/ 1, seq_len, dt, pre,
                   st) == XML_OK)
    return XML_OK;                                         // (a refused launch falls through to the 3-launch path)
  rc = xmli_gemm(xn, w, b, pos, pre, rows, hidden, d_pad, /*relu*/ 1, /*add_mode*/ 1, seq_len, /*out_f32*/ 1, dt, st, sws);
  if (rc) return rc;
  return xmli_add_layernorm(pre, XML_F32, nullptr, ln_pos_g, ln_pos_b, y, rows, hidden, hidden, adt, st);
}

// ---------------------------------------------------------------------------------------------------
// Packing plan of a padded token batch + K1+K2 on the packed tokens (the query encoder without its padding rows)
// ---------------------------------------------------------------------------------------------------
// Three small launches instead of a dozen framework kernels with a host sync in between (10 000 x 30 tokens: ~1 MB of mask):
//   lengths   one wave per sequence: lane t reads mask[t]; the row must be a non-empty PREFIX of ones (values exactly 0 / 1)
//   scan      one workgroup: cu_seqlens = exclusive scan of the lengths; status[0] = packed rows, or -1 for a bad mask
//   fill      one wave per sequence: source row of every packed token
__global__ __launch_bounds__(256) void pack_len_kernel(const float* __restrict__ mask, int n, int lq, int32_t* __restrict__ cu,
                                                       int32_t* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const float v = lane < lq ? mask[(int64_t)r * lq + lane] : 0.f;
  const unsigned long long ones = __ballot(v == 1.f), other = __ballot(v != 1.f && v != 0.f);
  const int len = (int)__popcll(ones);
  const bool ok = other == 0 && len > 0 && ones == (len == 64 ? ~0ull : ((1ull << len) - 1ull));
  if (lane == 0) {
    cu[r + 1] = len;
    if (!ok) atomicOr(&status[1], 1);
  }
}

__global__ __launch_bounds__(1024) void pack_scan_kernel(int n, int32_t* __restrict__ cu, int32_t* __restrict__ status) {
  __shared__ int s_wave[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n + 1023) / 1024;
  const int r0 = min(n, tid * per), r1 = min(n, r0 + per);
  int sum = 0;
  for (int r = r0; r < r1; ++r) sum += cu[r + 1];
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int base = inc - sum;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  for (int r = r0; r < r1; ++r) {            // (a thread reads only its own entries before it overwrites them)
    base += cu[r + 1];
    cu[r + 1] = base;
  }
  if (tid == 0) {
    cu[0] = 0;
    int total = 0;
    for (int w = 0; w < 16; ++w) total += s_wave[w];
    status[0] = status[1] ? -1 : total;
  }
}

__global__ __launch_bounds__(256) void pack_fill_kernel(int n, int lq, const int32_t* __restrict__ cu, int32_t* __restrict__ src_row) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int b = cu[r], len = cu[r + 1] - b;
  if (lane < len) src_row[b + lane] = r * lq + lane;
}

extern "C" int xml_pack_plan(const float* mask, int64_t n, int lq, int32_t* cu_seqlens, int32_t* src_row, int32_t* status,
                             xml_stream_t stream) {
  XML_ENTER();
  if (!mask || !cu_seqlens || !src_row || !status || n <= 0 || lq <= 0) return XML_ERR_BAD_ARG;
  if (lq > 64 || n * lq > (int64_t)INT32_MAX) return XML_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  xml_zero_async(status, 8, st);
  hipLaunchKernelGGL(pack_len_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, mask, (int)n, lq, cu_seqlens, status);
  hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(1024), 0, st, (int)n, cu_seqlens, status);
  hipLaunchKernelGGL(pack_fill_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, (int)n, lq, cu_seqlens, src_row);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// dst[i] = table[src_row[i] % lq]: the positional row of every packed token (16-byte pieces; row_bytes % 16 == 0)
__global__ __launch_bounds__(256) void gather_pos_rows_kernel(const uint4* __restrict__ table, const int32_t* __restrict__ src_row,
                                                              uint4* __restrict__ dst, int64_t rows, int lq, int vec_per_row) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * vec_per_row) return;
  const int64_t r = i / vec_per_row;
  const int v = (int)(i - r * vec_per_row);
  dst[i] = table[(int64_t)(src_row[r] % lq) * vec_per_row + v];
}

extern "C" size_t xml_linear_ln_relu_pos_packed_workspace_bytes(int64_t rows, int d_in, int hidden, int dt) {
  return xml_linear_ln_relu_pos_workspace_bytes(rows, d_in, hidden, dt) + align_up((size_t)rows * hidden * dt_size(dt), 256);
}

extern "C" int xml_linear_ln_relu_pos_packed(const void* x, int x_dt, const int32_t* src_row, int lq, const float* ln_in_g,
                                             const float* ln_in_b, const void* w, const float* b, const void* pos,
                                             const float* ln_pos_g, const float* ln_pos_b, void* y, int64_t rows, int d_in,
                                             int hidden, int dt, void* ws, size_t ws_bytes, xml_stream_t stream) {
  XML_ENTER();
  if (!x || !src_row || !ln_in_g || !ln_in_b || !w || !b || !pos || !ln_pos_g || !ln_pos_b || !y || !ws) return XML_ERR_BAD_ARG;
  if (rows <= 0 || rows > (int64_t)INT32_MAX || lq <= 0 || !xmli_model_dt_ok(dt)) return XML_ERR_BAD_ARG;
  if (d_in <= 0 || d_in % 8 || d_in > 4096 || hidden % 8) return XML_ERR_UNSUPPORTED;
  if (ws_bytes < xml_linear_ln_relu_pos_packed_workspace_bytes(rows, d_in, hidden, dt)) return XML_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int adt = xmli_act_dt(dt);
  char* pg = (char*)ws;                                            // gathered positional rows (rows, hidden) dt
  char* xn = pg + align_up((size_t)rows * hidden * dt_size(dt), 256);
  char* pre = xn + align_up((size_t)rows * d_in * dt_size(dt), 256);
  char* sws = dt == XML_F16S ? (char*)ws + xml_linear_ln_relu_pos_packed_workspace_bytes(rows, d_in, hidden, dt) -
                                   xmli_gemm_split_ws_bytes(rows, d_in, dt) : nullptr;
  const int vpr = hidden * (int)dt_size(dt) / 16;
  hipLaunchKernelGGL(gather_pos_rows_kernel, dim3(cdiv(rows * vpr, 256)), dim3(256), 0, st, (const uint4*)pos, src_row,
                     (uint4*)pg, rows, lq, vpr);
  XML_CHECK_LAUNCH();
  int rc = add_layernorm_rows(x, x_dt, nullptr, ln_in_g, ln_in_b, xn, rows, d_in, d_in, adt, st, src_row);
  if (rc) return rc;
  // with seq_len = rows, "row % seq_len" addresses the gathered positional rows one to one
  if (xmli_gemm_ln_eligible(rows, hidden, d_in, dt) &&
      xmli_gemm_ln(xn, w, b, pg, ln_pos_g, ln_pos_b, y, rows, hidden, d_in, /*relu*/ 1, /*add_mode*/ 1, (int)rows, dt, pre,
                   st) == XML_OK)
    return XML_OK;
  rc = xmli_gemm(xn, w, b, pg, pre, rows, hidden, d_in, /*relu*/ 1, /*add_mode*/ 1, (int)rows, /*out_f32*/ 1, dt, st, sws);
  if (rc) return rc;
  return xmli_add_layernorm(pre, XML_F32, nullptr, ln_pos_g, ln_pos_b, y, rows, hidden, hidden, adt, st);
}
