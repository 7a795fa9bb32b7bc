// K6, MI355X-tuned variant: 256 x 256 workgroup tile, 8 waves, LDS-DMA staging (global_load_lds, 16 B/lane).
//
// Why this geometry (cdna guide section 5 / MI355X_MICROARCH LDS table): at the dense bf16 MFMA rate a CU retires
// 4096 flop/clk, while ds_write tops out at 64-85 B/clk and ds_read_b128 at 256 B/clk.  A 128x128 tile staged
// through VGPRs needs 64 B/clk of LDS writes at peak -- the write path alone caps it.  Here
//   * the tile is 256 queries x 256 clip rows (two 128-column groups = whole videos), K step 128 bytes per row;
//   * operands go HBM/L2 -> LDS by DMA (no VGPR round trip, no ds_write issue), double buffered: 2 x 64 KiB;
//   * waves are laid out 4 (M) x 2 (N): each owns 64 query rows x one 128-column group, i.e. ONE WHOLE VIDEO at
//     L = 128, so the max-over-clips epilogue stays inside the wave (8 MFMA tiles in registers + a 16-lane
//     butterfly); 12 ds_read_b128 feed 32 MFMAs per 64-byte K chunk;
//   * LDS image is lane-linear per DMA instruction (8 rows x 128 B per wave-instruction); the XOR swizzle
//     slot ^= (row >> 1) & 7 that keeps ds_read_b128 conflict-free is applied on the per-lane SOURCE address
//     and again on the read address (guide rule 21: both sides or neither).
// Tile order is XCD-aware (workgroup b runs on XCD b % 8): each XCD walks 8 x 4 super-tiles so its 32 resident
// workgroups share 8 query tiles and 4 clip tiles through that XCD's private L2.
#include "common.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <typename T, int ABL = 0>   // ABL: perf ablations only (1: DMA for the first K tile only, 2: no MFMA)
__global__ __launch_bounds__(512, 2) void q2c_scores_kernel_256(const T* __restrict__ qn, const T* __restrict__ cn,
                                                                const float* __restrict__ mask,
                                                                float* __restrict__ out, int64_t ld_out, int nq,
                                                                int nv, int lpad, int hidden, int combine, int tq,
                                                                int tc) {
  constexpr int ROWB = 128;                 // bytes of K per row per stage
  constexpr int OPER_BYTES = 256 * ROWB;    // one operand, one stage: 32 KiB
  constexpr int STAGE_BYTES = 2 * OPER_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- XCD-aware tile assignment ---------------------------------------------------------------------
  const int b = blockIdx.x;
  const int xcd = b & 7, local = b >> 3;
  const int sup = (local >> 5) * 8 + xcd;
  const int w32 = local & 31;
  const int sq = (tq + 7) >> 3;
  const int qt = (sup % sq) * 8 + (w32 & 7);
  const int ct = (sup / sq) * 4 + (w32 >> 3);
  if (qt >= tq || ct >= tc) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int vpg = 128 / lpad;               // videos per 128-column group
  const int gcols = vpg * lpad;             // used columns of a group
  const int q0 = qt * 256;
  const int v0 = ct * 2 * vpg;              // first video of the tile
  const int k_bytes = hidden * (int)sizeof(T);
  const int nk = k_bytes / ROWB;

  // ---- DMA source pointers: wave `wave` moves row groups g = wave + 8 i (8 rows x 128 B each) ----------
  const char* a_src[4];
  const char* b_src[4];
  {
    const int rsub = lane >> 3, pslot = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave + 8 * i) * 8 + rsub;             // tile-local row 0..255
      const int slot = pslot ^ ((row >> 1) & 7);             // logical 16-byte slot this lane must fetch
      int qrow = q0 + row;
      qrow = qrow < nq ? qrow : nq - 1;                       // clamp: finite data, results discarded
      a_src[i] = reinterpret_cast<const char*>(qn) + (int64_t)qrow * k_bytes + slot * 16;
      const int grp = row >> 7, col = row & 127;              // column group / column inside it
      int vid = v0 + grp * vpg + col / lpad;
      int clip = col % lpad;
      if (col >= gcols || vid >= nv) { vid = 0; clip = 0; }
      b_src[i] = reinterpret_cast<const char*>(cn) + ((int64_t)vid * lpad + clip) * k_bytes + slot * 16;
    }
  }
  auto issue = [&](int kt, int stage) {
    char* sa = smem + stage * STAGE_BYTES;
    char* sb = sa + OPER_BYTES;
    const int koff = kt * ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((glb_void*)(a_src[i] + koff), (lds_void*)(sa + (wave + 8 * i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void*)(b_src[i] + koff), (lds_void*)(sb + (wave + 8 * i) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane read offsets: row = base + 16 t + fr  ->  swizzle key (row >> 1) & 7 == fr >> 1
  const int key = fr >> 1;
  const int a_off0 = (wm * 64 + fr) * ROWB + ((fg ^ key) << 4);
  const int a_off1 = (wm * 64 + fr) * ROWB + (((4 + fg) ^ key) << 4);
  const int b_off0 = OPER_BYTES + (wn * 128 + fr) * ROWB + ((fg ^ key) << 4);
  const int b_off1 = OPER_BYTES + (wn * 128 + fr) * ROWB + (((4 + fg) ^ key) << 4);

  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's DMA pieces of stage kt have landed
    __syncthreads();                      // ... and everybody's; everybody is done reading the other stage
    if (kt + 1 < nk && ABL != 1) issue(kt + 1, (kt + 1) & 1);
    const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      uint4 fa[4], fb[8];
#pragma unroll
      for (int m = 0; m < 4; ++m)
        fa[m] = *reinterpret_cast<const uint4*>(st + (c ? a_off1 : a_off0) + m * 16 * ROWB);
#pragma unroll
      for (int n = 0; n < 8; ++n)
        fb[n] = *reinterpret_cast<const uint4*>(st + (c ? b_off1 : b_off0) + n * 16 * ROWB);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          if (ABL == 2) {
            asm volatile("" ::"v"(fa[m].x), "v"(fb[n].x));
            asm volatile("" ::"v"(fa[m].w), "v"(fb[n].w));
          } else {
            Mma<T>::chunk(acc[m][n], fa[m], fb[n]);
          }
        }
    }
  }

  // ---- epilogue: mask_logits + max over the clips of each video, all inside the wave --------------------
  float mk[8], fill[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int col = n * 16 + fr;
    const int vid = v0 + wn * vpg + col / lpad;
    const bool ok = col < gcols && vid < nv;
    mk[n] = ok ? mask[(int64_t)vid * lpad + (col % lpad)] : 0.f;
    fill[n] = (1.f - mk[n]) * -1e10f;
  }
  const int tpv = lpad >> 4;   // MFMA column tiles per video
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = q0 + wm * 64 + m * 16 + fg * 4 + r;
      float mx = -INFINITY;
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        mx = fmaxf(mx, acc[m][n][r] * mk[n] + fill[n]);      // mask_logits, xml/model_xml.py:640-641
        if ((n + 1) % tpv == 0) {
          const float red = lane16_max(mx);
          const int vid = v0 + wn * vpg + n / tpv;
          if (fr == 0 && row < nq && vid < nv && (n / tpv) < vpg) {
            float* po = out + (int64_t)row * ld_out + vid;
            *po = combine ? (*po + red) * 0.5f : red;
          }
          mx = -INFINITY;
        }
      }
    }
  }
}


template <typename T>
static int launch_q2c256(const void* qn, const void* cn, const float* mask, float* out, int64_t ld_out, int nq, int nv,
                         int lpad, int hidden, int combine, hipStream_t st) {
  const int vpg = 128 / lpad;
  const int tq = cdiv(nq, 256), tc = cdiv(nv, 2 * vpg);
  const int64_t nsup = (int64_t)((tq + 7) / 8) * ((tc + 3) / 4);
  const unsigned grid = (unsigned)(((nsup + 7) / 8) * 8 * 32);
  const int lds = 2 * 2 * 256 * 128;
  auto kern = q2c_scores_kernel_256<T, 0>;
  bool ok = xml_lds_attr_once<q2c_scores_kernel_256<T, 0>>(lds);
#ifdef XML_DEBUG_VARIANTS
  if (g_q2c_ablation == 1) { kern = q2c_scores_kernel_256<T, 1>; ok = xml_lds_attr_once<q2c_scores_kernel_256<T, 1>>(lds); }
  if (g_q2c_ablation == 2) { kern = q2c_scores_kernel_256<T, 2>; ok = xml_lds_attr_once<q2c_scores_kernel_256<T, 2>>(lds); }
#endif
  if (!ok) return XML_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, (const T*)qn, (const T*)cn, mask, out, ld_out, nq, nv, lpad,
                     hidden, combine, tq, tc);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// eligible when a K step never straddles the row end: hidden * sizeof(T) % 128 == 0
int xmli_q2c_scores_256(const void* qn, const void* cn, const float* mask, float* out, int64_t ld_out, int nq, int nv,
                        int lpad, int hidden, int combine, int dt, hipStream_t st) {
  if (dt == XML_BF16) return launch_q2c256<bf16_t>(qn, cn, mask, out, ld_out, nq, nv, lpad, hidden, combine, st);
  return launch_q2c256<float>(qn, cn, mask, out, ld_out, nq, nv, lpad, hidden, combine, st);
}
