// K6: video-level retrieval scores = similarity GEMM #1 with a fused masked max-over-clips epilogue.
//   reference: XML.get_video_level_scores  xml/model_xml.py:436-453
//              (video + sub) / divisor     xml/model_xml.py:572-574
//
//   out[q][v] = max_l ( s*m + (1-m)*-1e10 ),  s = qn[q] . cn[v][l]
//
// The (Nq, Nv, L) similarity tensor never exists in memory: each workgroup contracts a 128-query x
// 128-clip-row tile over H on the MFMA pipe (gemm.h) and reduces the clip axis in registers / LDS.
// A tile's 128 columns hold floor(128 / lpad) whole videos (lpad % 16 == 0, so a 16-wide MFMA tile never
// straddles two videos); at the TVR shape lpad = 128 and a tile is exactly one video.
//
// Tile order is XCD-aware: MI355X dispatches workgroup b to XCD b % 8, each XCD has a private 4 MiB L2.
// Workgroups are grouped in 8x8 super-tiles (8 query tiles x 8 clip tiles = 1.5 MiB + 1.5 MiB of bf16
// operands at H=768) and every XCD walks its own sequence of super-tiles, so the 64 workgroups resident on
// one XCD share operands through that XCD's L2 instead of each re-fetching from Infinity Cache / HBM.
#include "gemm.h"

template <typename T>
__global__ __launch_bounds__(256) void q2c_scores_kernel(const T* __restrict__ qn, const T* __restrict__ cn,
                                                         const float* __restrict__ mask, float* __restrict__ out,
                                                         int64_t ld_out, int nq, int nv, int lpad, int hidden,
                                                         int combine, int tq, int tc, int xcd_swizzle) {
  using Cfg = GemmCfg<T, 128, 128, 2, 2>;
  __shared__ __attribute__((aligned(16))) char smem[Cfg::LDS_BYTES];

  int qt, ct;
  if (xcd_swizzle) {
    const int b = blockIdx.x;
    const int xcd = b & 7, local = b >> 3;
    const int sup = (local >> 6) * 8 + xcd;  // super-tile index owned by this XCD
    const int w = local & 63;
    const int sq = (tq + 7) >> 3;
    const int s_q = sup % sq, s_c = sup / sq;
    qt = s_q * 8 + (w & 7);
    ct = s_c * 8 + (w >> 3);
  } else {
    qt = blockIdx.x % tq;
    ct = blockIdx.x / tq;
  }
  if (qt >= tq || ct >= tc) return;

  const int vpt = 128 / lpad;            // videos per column tile
  const int cols = vpt * lpad;           // used columns of the tile
  const int q0 = qt * 128;
  const int v0 = ct * vpt;
  const int64_t row0 = (int64_t)v0 * lpad;  // first clip row of the tile in cn
  const int64_t nrows = (int64_t)nv * lpad;

  f32x4 acc[Cfg::MT][Cfg::NT];
  auto a_row = [&](int r) -> const char* {
    return (q0 + r) < nq ? reinterpret_cast<const char*>(qn + (int64_t)(q0 + r) * hidden) : nullptr;
  };
  auto b_row = [&](int r) -> const char* {
    return (r < cols && row0 + r < nrows) ? reinterpret_cast<const char*>(cn + (row0 + r) * hidden) : nullptr;
  };
  gemm_mainloop<T, Cfg>(acc, a_row, b_row, hidden * (int)sizeof(T), smem);

  // ---- epilogue: mask, max over the 16 columns of each MFMA tile, then over the tiles of a video ----
  float* red = reinterpret_cast<float*>(smem);  // [128 rows][8 column tiles]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < Cfg::NT; ++nt) {
    const int col = wn * 64 + nt * 16 + fr;
    const int64_t crow = row0 + col;
    const float m = (col < cols && crow < nrows) ? mask[crow] : 0.f;
    const float fill = (1.f - m) * -1e10f;
#pragma unroll
    for (int mt = 0; mt < Cfg::MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = acc[mt][nt][r];
        // XML_F16: hi planes of unit-norm rows at the fixed scale 2^XML_F16_UNIT_LOG2 on both sides (split16.hip)
        if constexpr (std::is_same<T, f16_t>::value) s *= 1.f / (float)(1u << (2 * XML_F16_UNIT_LOG2));
        s = s * m + fill;  // mask_logits, xml/model_xml.py:640-641
        s = lane16_max(s);
        if (fr == 0) red[(wm * 64 + mt * 16 + fg * 4 + r) * 8 + wn * 4 + nt] = s;
      }
    }
  }
  __syncthreads();
  const int tiles_per_video = lpad >> 4;
  for (int i = threadIdx.x; i < 128 * vpt; i += 256) {
    const int row = i % 128, v = i / 128;
    if (q0 + row >= nq || v0 + v >= nv) continue;
    float mx = -INFINITY;
    for (int t = 0; t < tiles_per_video; ++t) mx = fmaxf(mx, red[row * 8 + v * tiles_per_video + t]);
    float* po = out + (int64_t)(q0 + row) * ld_out + v0 + v;
    *po = combine ? (*po + mx) * 0.5f : mx;
  }
}

#ifdef XML_DEBUG_VARIANTS
int g_q2c_xcd_swizzle = 1, g_q2c_variant = 0, g_q2c_ablation = 0, g_q2c_chunk_log2 = -1, g_q2c_line_log2 = 0, g_q2c_qsh = -1;
extern "C" void xml_debug_set_q2c_swizzle(int on) { g_q2c_xcd_swizzle = on; }
extern "C" void xml_debug_set_q2c_variant(int v) { g_q2c_variant = v; }
extern "C" void xml_debug_set_q2c_ablation(int v) { g_q2c_ablation = v; }
extern "C" void xml_debug_set_q2c_chunk(int v) { g_q2c_chunk_log2 = v; }
extern "C" void xml_debug_set_q2c_line(int v) { g_q2c_line_log2 = v; }
extern "C" void xml_debug_set_q2c_qsh(int v) { g_q2c_qsh = v; }
#endif

int xmli_q2c_scores_persist(int n_mod, const void* const* qn, const void* const* cn, const float* const* mask,
                            float* out, int64_t ld_out, int nq, int nv, int lpad, int hidden, int dt, hipStream_t st,
                            bool tiled = false, int mask_mode = 0, const uint32_t* const* mbits = nullptr,
                            const int32_t* slot_ids = nullptr);
int xmli_q2c_scores_persist32(int n_mod, const void* const* qn, const void* const* cn, const float* const* mask,
                              float* out, int64_t ld_out, int nq, int nv, int lpad, int hidden, int dt, hipStream_t st);
int xmli_q2c_scores_persist4(int n_mod, const void* const* qn, const void* const* cn, const float* const* mask,
                             float* out, int64_t ld_out, int nq, int nv, int lpad, int hidden, int dt, hipStream_t st);
int xmli_q2c_scores_ring(const void* qn, const void* cn, const float* mask, float* out, int64_t ld_out, int nq, int nv,
                         int lpad, int hidden, int combine, int dt, hipStream_t st);
int xmli_q2c_scores_256(const void* qn, const void* cn, const float* mask, float* out, int64_t ld_out, int nq, int nv,
                        int lpad, int hidden, int combine, int dt, hipStream_t st);

extern "C" int xml_q2c_scores(const void* qn, const void* cn, const float* mask, float* out, int64_t ld_out, int nq,
                              int nv, int lpad, int hidden, int combine, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!qn || !cn || !mask || !out || nq <= 0 || nv <= 0 || lpad <= 0 || hidden <= 0 || ld_out < nv)
    return XML_ERR_BAD_ARG;
  if (lpad % 16 || lpad > 128 || hidden % 8) return XML_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dt != XML_F32 && dt != XML_BF16 && dt != XML_F16) return XML_ERR_BAD_ARG;
  // (XML_F16 row-major operands -- corpora too small / hidden sizes too short for the tiled persistent kernel -- take the
  // register-staged kernel below)
  const bool dma_ok = dt != XML_F16 && ((size_t)hidden * dt_size(dt)) % 128 == 0;
  const bool persist_ok = dt != XML_F16 && lpad == 128 && ((size_t)hidden * dt_size(dt)) % 128 == 0 &&
                          (size_t)hidden * dt_size(dt) >= 384;
  if ((g_q2c_variant == 0 || g_q2c_variant == 4) && persist_ok && !combine) {
    const void* q[2] = {qn, qn};
    const void* c[2] = {cn, cn};
    const float* m[2] = {mask, mask};
    return xmli_q2c_scores_persist(1, q, c, m, out, ld_out, nq, nv, lpad, hidden, dt, st);
  }
#ifdef XML_DEBUG_VARIANTS
  if (g_q2c_variant == 6 && persist_ok && !combine && dt == XML_BF16) {
    const void* q[2] = {qn, qn};
    const void* c[2] = {cn, cn};
    const float* m[2] = {mask, mask};
    return xmli_q2c_scores_persist32(1, q, c, m, out, ld_out, nq, nv, lpad, hidden, dt, st);
  }
  if (g_q2c_variant == 5 && persist_ok && !combine) {
    const void* q[2] = {qn, qn};
    const void* c[2] = {cn, cn};
    const float* m[2] = {mask, mask};
    return xmli_q2c_scores_persist4(1, q, c, m, out, ld_out, nq, nv, lpad, hidden, dt, st);
  }
#endif
  if ((g_q2c_variant == 0 || g_q2c_variant == 3) && dma_ok)
    return xmli_q2c_scores_ring(qn, cn, mask, out, ld_out, nq, nv, lpad, hidden, combine, dt, st);
#ifdef XML_DEBUG_VARIANTS
  if (g_q2c_variant == 2 && dma_ok)      // q2c256.hip: debug library only
    return xmli_q2c_scores_256(qn, cn, mask, out, ld_out, nq, nv, lpad, hidden, combine, dt, st);
#endif
  const int vpt = 128 / lpad;
  const int tq = cdiv(nq, 128), tc = cdiv(nv, vpt);
  const int swz = g_q2c_xcd_swizzle;
  unsigned grid;
  if (swz) {
    const int64_t nsup = (int64_t)((tq + 7) / 8) * ((tc + 7) / 8);
    grid = (unsigned)(((nsup + 7) / 8) * 8 * 64);
  } else {
    grid = (unsigned)((int64_t)tq * tc);
  }
  if (dt == XML_F32)
    hipLaunchKernelGGL(q2c_scores_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)qn, (const float*)cn,
                       mask, out, ld_out, nq, nv, lpad, hidden, combine, tq, tc, swz);
  else if (dt == XML_BF16)
    hipLaunchKernelGGL(q2c_scores_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)qn, (const bf16_t*)cn,
                       mask, out, ld_out, nq, nv, lpad, hidden, combine, tq, tc, swz);
  else if (dt == XML_F16)
    hipLaunchKernelGGL(q2c_scores_kernel<f16_t>, dim3(grid), dim3(256), 0, st, (const f16_t*)qn, (const f16_t*)cn,
                       mask, out, ld_out, nq, nv, lpad, hidden, combine, tq, tc, swz);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" int xml_q2c_scores_fused(int n_mod, const void* qn0, const void* cn0, const float* mask0, const void* qn1,
                                    const void* cn1, const float* mask1, float* out, int64_t ld_out, int nq, int nv,
                                    int lpad, int hidden, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (n_mod < 1 || n_mod > 2 || !qn0 || !cn0 || !mask0 || !out) return XML_ERR_BAD_ARG;
  if (n_mod == 2 && (!qn1 || !cn1 || !mask1)) return XML_ERR_BAD_ARG;
  if (nq <= 0 || nv <= 0 || lpad <= 0 || hidden <= 0 || ld_out < nv) return XML_ERR_BAD_ARG;
  if (dt != XML_F32 && dt != XML_BF16 && dt != XML_F16) return XML_ERR_BAD_ARG;
  if (lpad % 16 || lpad > 128 || hidden % 8) return XML_ERR_UNSUPPORTED;
  const size_t kb = (size_t)hidden * dt_size(dt);
  const bool persist_ok = dt != XML_F16 && lpad == 128 && kb % 128 == 0 && kb >= 384;
  if ((g_q2c_variant == 0 || g_q2c_variant == 4) && persist_ok) {
    const void* q[2] = {qn0, qn1};
    const void* c[2] = {cn0, cn1};
    const float* m[2] = {mask0, mask1};
    return xmli_q2c_scores_persist(n_mod, q, c, m, out, ld_out, nq, nv, lpad, hidden, dt, (hipStream_t)stream);
  }
#ifdef XML_DEBUG_VARIANTS
  if (g_q2c_variant == 6 && persist_ok && dt == XML_BF16) {
    const void* q[2] = {qn0, qn1};
    const void* c[2] = {cn0, cn1};
    const float* m[2] = {mask0, mask1};
    return xmli_q2c_scores_persist32(n_mod, q, c, m, out, ld_out, nq, nv, lpad, hidden, dt, (hipStream_t)stream);
  }
  if (g_q2c_variant == 5 && persist_ok) {
    const void* q[2] = {qn0, qn1};
    const void* c[2] = {cn0, cn1};
    const float* m[2] = {mask0, mask1};
    return xmli_q2c_scores_persist4(n_mod, q, c, m, out, ld_out, nq, nv, lpad, hidden, dt, (hipStream_t)stream);
  }
#endif
  int rc = xml_q2c_scores(qn0, cn0, mask0, out, ld_out, nq, nv, lpad, hidden, 0, dt, stream);
  if (rc || n_mod == 1) return rc;
  return xml_q2c_scores(qn1, cn1, mask1, out, ld_out, nq, nv, lpad, hidden, 1, dt, stream);
}
