// K7: similarity contraction #2 + ConvSE start/end scorer, evaluated only on the selected (query, video) pairs.
//   reference: get_merged_st_ed_prob   xml/model_xml.py:455-502   (merge_two_stream)
//              _get_st_ed_prob         xml/model_xml.py:512-551   (single stream / no merge)
//              softmax over clips      xml/inference.py:321-322
//
// The reference contracts every query with every clip of every video ((Nq,Nv,L) fp32, 111.6 GB at the TVR
// shape) and then keeps only the rows of the top-k videos (xml/inference.py:365-367).  Here the pair list
// is inverted on the device (video -> the queries that selected it), so each video's feat2 tile (L x H) is
// fetched from HBM once and contracted on the MFMA pipe against the <= TM query vectors of a chunk:
//   bytes ~= Nv*L*H*b*modalities (+ L2-resident query vectors) instead of Nq*k*L*H*b*modalities.
// Epilogue per pair row, in LDS: k-tap zero-padded cross-correlation x2 (start / end filters), mask_logits,
// softmax over clips.
#include "band.h"
#include "gemm.h"

static constexpr int TM = 64;            // pairs per workgroup chunk (a video has ~46 pairs at C3: one chunk each)
static constexpr int LP = 128 + 8;       // row of the LDS similarity patch (floats): [4 halo | 128 clips | 4 halo]
static constexpr int LH = 4;             // the clips start at float LH of a row (the fast epilogue reads 2 before / 5 past)

// Popular videos: with few videos and many pairs (TVR val: 2 179 videos, 1.09 M pairs = 500 per counter) the counting
// atomics serialise on their addresses -- 138 us of a 1.1 ms K7.  Every video then gets CONVSE_SUB_MAX sub-counters, a pair
// uses sub-counter (pair id % sub): 16 x fewer collisions per address; the scan runs over the sub-buckets in (video, sub)
// order, so a video's bucket is still one contiguous range.
static constexpr int CONVSE_SUB_MAX = 16;
static inline int convse_sub(int64_t P, int nv) { return P >= 128 * (int64_t)nv ? CONVSE_SUB_MAX : 1; }

struct ConvseWs {
  int32_t* counts;     // [nv * sub]   pairs per (video, sub-counter)
  int32_t* offsets;    // [nv * sub + 1] exclusive scan of counts; video v's bucket = [offsets[v * sub], offsets[(v + 1) * sub])
  int32_t* pos;        // [P]    rank of a pair among the pairs of its video (the value its counting atomic returned)
  int32_t* chunk_off;  // [nv+1] exclusive scan of ceil(count / TM)
  int32_t* bucket;     // [P]    pair ids grouped by video
  int32_t* chunk_vid;  // [P / TM + min(P, nv)] video of every chunk (saves a 15-step dependent binary search per workgroup)
};

static size_t convse_ws_layout(const xml_convse_desc* d, ConvseWs* w, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  };
  const size_t nv = (size_t)d->nv, P = (size_t)d->nq * d->kpairs;
  char* counts = take(nv * CONVSE_SUB_MAX * 4);             // (sub-counters: see convse_sub)
  char* offsets = take((nv * CONVSE_SUB_MAX + 1) * 4);
  char* chunk_off = take((nv + 1) * 4);
  char* bucket = take(P * 4);
  char* pos = take(P * 4);
  char* chunk_vid = take((P / TM + (P < nv ? P : nv) + 1) * 4);
  if (w) {
    w->chunk_vid = (int32_t*)chunk_vid;
    w->counts = (int32_t*)counts; w->pos = (int32_t*)pos; w->offsets = (int32_t*)offsets;
    w->chunk_off = (int32_t*)chunk_off; w->bucket = (int32_t*)bucket;
  }
  return off;
}

extern "C" size_t xml_convse_rerank_workspace_bytes(const xml_convse_desc* d) {
  if (!d) return 0;
  return convse_ws_layout(d, nullptr, nullptr);
}

// Inverting the pair list (video -> the pairs that selected it) in three passes with ONE atomic per pair: the counting
// atomic's return value is the pair's rank inside its video's bucket, so the fill pass is a plain scatter.  (The order of a
// bucket depends on the atomics' arrival order; every pair writes its own output row, so the results do not.)
__global__ void convse_count_kernel(const int32_t* __restrict__ pair_vid, int32_t* __restrict__ counts,
                                    int32_t* __restrict__ pos, int64_t P, int nv, int sub) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int v = pair_vid[p];
  pos[p] = (v >= 0 && v < nv) ? atomicAdd(&counts[(int64_t)v * sub + (int)(p & (sub - 1))], 1) : -1;
}

// rows of skipped pairs (pair_vid < 0: owned by another shard / padding) are zero-filled: one thread per pair looks, the
// (rare) skipped ones write their two rows
__global__ void convse_zero_skipped_kernel(const int32_t* __restrict__ pair_vid, float* __restrict__ st_out,
                                           float* __restrict__ ed_out, int64_t P, int nv, int lpad4) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int v = pair_vid[p];
  if (v >= 0 && v < nv) return;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4* s4 = reinterpret_cast<float4*>(st_out) + p * lpad4;
  float4* e4 = reinterpret_cast<float4*>(ed_out) + p * lpad4;
  for (int i = 0; i < lpad4; ++i) { s4[i] = z; e4[i] = z; }
}

// single workgroup: exclusive scans of counts and of ceil(counts / TM).  Thread t owns the consecutive videos
// [t * per, (t + 1) * per): serial sums, one shuffle scan over the 1024 partial sums, serial write-out.
__global__ __launch_bounds__(1024) void convse_scan_kernel(const int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ offsets,
                                                           int32_t* __restrict__ chunk_off,
                                                           int32_t* __restrict__ chunk_vid, int nv, int TM, int sub) {
  __shared__ int32_t wa[16], wb[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (nv + 1023) / 1024;
  const int r0 = min(nv, tid * per), r1 = min(nv, r0 + per);
  int sa = 0, sb = 0;
  for (int i = r0; i < r1; ++i) {
    int c = 0;
    for (int j = 0; j < sub; ++j) c += counts[(int64_t)i * sub + j];
    sa += c; sb += (c + TM - 1) / TM;
  }
  int ia = sa, ib = sb;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int va = __shfl_up(ia, o, 64), vb = __shfl_up(ib, o, 64);
    if (lane >= o) { ia += va; ib += vb; }
  }
  if (lane == 63) { wa[wave] = ia; wb[wave] = ib; }
  __syncthreads();
  int a = ia - sa, b = ib - sb;
  for (int w = 0; w < wave; ++w) { a += wa[w]; b += wb[w]; }
  for (int i = r0; i < r1; ++i) {
    int c = 0;
    for (int j = 0; j < sub; ++j) {
      const int cj = counts[(int64_t)i * sub + j];
      offsets[(int64_t)i * sub + j] = a + c;
      c += cj;
    }
    const int ch = (c + TM - 1) / TM;
    chunk_off[i] = b;
    for (int j = 0; j < ch; ++j) chunk_vid[b + j] = i;
    a += c; b += ch;
  }
  if (tid == 1023) { offsets[(int64_t)nv * sub] = a; chunk_off[nv] = b; }    // the last thread's running sums are the totals
}

__global__ void convse_fill_kernel(const int32_t* __restrict__ pair_vid, const int32_t* __restrict__ offsets,
                                   const int32_t* __restrict__ pos, int32_t* __restrict__ bucket, int64_t P, int sub) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int r = pos[p];
  if (r >= 0) bucket[offsets[(int64_t)pair_vid[p] * sub + (int)(p & (sub - 1))] + r] = (int32_t)p;
}

// Small pair lists (the reference's 50-query batches: 5 000 pairs): the five launches above -- zero, count, zero skipped
// rows, scan, fill -- cost 24 us of launch floor in a 350 us batch.  ONE workgroup does all of it: counters and bucket
// starts in LDS, ranks kept in the `pos` scratch by the thread that took them.
static constexpr int CONVSE_SMALL_NV = 4096, CONVSE_SMALL_P = 32768;
__global__ __launch_bounds__(1024) void convse_invert_small_kernel(const int32_t* __restrict__ pair_vid,
                                                                   int32_t* __restrict__ offsets, int32_t* __restrict__ chunk_off,
                                                                   int32_t* __restrict__ chunk_vid, int32_t* __restrict__ pos,
                                                                   int32_t* __restrict__ bucket, float* __restrict__ st_out,
                                                                   float* __restrict__ ed_out, int P, int nv, int TM, int lpad4,
                                                                   int zero_skipped) {
  __shared__ int32_t s_cnt[CONVSE_SMALL_NV], s_off[CONVSE_SMALL_NV];
  __shared__ int32_t wa[16], wb[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < nv; i += 1024) s_cnt[i] = 0;
  // up to 8 pairs per thread (8 192 pairs: a 50-query batch has 5 000): video and rank of a thread's pairs stay in its
  // registers between the counting and the fill pass instead of going through the `pos` scratch and a second read of pair_vid
  const bool in_regs = P <= 8 * 1024;
  int v_r[8], pos_r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int p = tid + j * 1024;
    v_r[j] = (in_regs && p < P) ? pair_vid[p] : -1;
  }
  __syncthreads();
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = tid + j * 1024;
      const bool ok = v_r[j] >= 0 && v_r[j] < nv;
      pos_r[j] = ok ? atomicAdd(&s_cnt[v_r[j]], 1) : -1;
      if (p < P && !ok && zero_skipped) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        float4* s4 = reinterpret_cast<float4*>(st_out) + (int64_t)p * lpad4;
        float4* e4 = reinterpret_cast<float4*>(ed_out) + (int64_t)p * lpad4;
        for (int i = 0; i < lpad4; ++i) { s4[i] = z; e4[i] = z; }
      }
    }
  } else {
  for (int p = tid; p < P; p += 1024) {
    const int v = pair_vid[p];
    const bool ok = v >= 0 && v < nv;
    pos[p] = ok ? atomicAdd(&s_cnt[v], 1) : -1;
    if (!ok && zero_skipped) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      float4* s4 = reinterpret_cast<float4*>(st_out) + (int64_t)p * lpad4;
      float4* e4 = reinterpret_cast<float4*>(ed_out) + (int64_t)p * lpad4;
      for (int i = 0; i < lpad4; ++i) { s4[i] = z; e4[i] = z; }
    }
  }
  }
  __syncthreads();
  const int per = (nv + 1023) / 1024;
  const int r0 = min(nv, tid * per), r1 = min(nv, r0 + per);
  int sa = 0, sb = 0;
  for (int i = r0; i < r1; ++i) { const int c = s_cnt[i]; sa += c; sb += (c + TM - 1) / TM; }
  int ia = sa, ib = sb;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int va = __shfl_up(ia, o, 64), vb = __shfl_up(ib, o, 64);
    if (lane >= o) { ia += va; ib += vb; }
  }
  if (lane == 63) { wa[wave] = ia; wb[wave] = ib; }
  __syncthreads();
  int a = ia - sa, b = ib - sb;
  for (int w = 0; w < wave; ++w) { a += wa[w]; b += wb[w]; }
  for (int i = r0; i < r1; ++i) {
    const int c = s_cnt[i], ch = (c + TM - 1) / TM;
    offsets[i] = a; s_off[i] = a;
    chunk_off[i] = b;
    for (int j = 0; j < ch; ++j) chunk_vid[b + j] = i;
    a += c; b += ch;
  }
  if (tid == 1023) { offsets[nv] = a; chunk_off[nv] = b; }
  __syncthreads();
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (pos_r[j] >= 0) bucket[s_off[v_r[j]] + pos_r[j]] = tid + j * 1024;
    return;
  }
  for (int p = tid; p < P; p += 1024) {
    const int r = pos[p];                            // (written by this very thread above)
    if (r >= 0) bucket[s_off[pair_vid[p]] + r] = p;
  }
}

struct ConvseArgs {
  const void* q_lin[2];
  const void* feat2[2];
  const float* mask[2];
  const float* conv_w;
  float* st_out;
  float* ed_out;
  const int32_t* offsets;
  const int32_t* chunk_off;
  const int32_t* bucket;
  const int32_t* chunk_vid;
  int nv, kpairs, lpad, l_ref, hidden, n_mod, merged, ksize, softmax;
  int sub;   // sub-counters per video in `offsets` (convse_sub)
  int dbg;   // perf ablations (xml_debug_set_q2c_ablation): 1 skip the GEMMs, 2 skip the conv / softmax / store epilogue
  int dma;   // rows are whole 128-byte K steps and every row offset fits 32 bits: the LDS-DMA mainloop (gemm.h)
  // XML_F16S operands (split-f16 rows, split16.hip): 1 / S of every query row (nq) and of every clip row (nv * lpad),
  // per modality -- powers of two, so moving an accumulator from one modality's units to the other's is exact
  const float* q_inv[2];
  const float* c_inv[2];
  // candidate summaries for K9 (xml_convse_rerank_ex): summ[p][g] = the largest banded row maximum
  // (st[i] * pair_w[p]) * max_{min_l <= d < max_l} ed[i + d] over the rows i of group g (i % 64 in [8 g, 8 g + 8)) of pair p,
  // taken while its rows are in this wave's registers
  const float* pair_w;
  float* summ;
  int min_l, max_l;
  // ragged corpora (xml_convse_rerank_ex): valid clips of every video (1 + index of its last unmasked clip), or NULL.
  // Given: clip rows >= vid_len[v] + ksize / 2 of a video are not fetched (the rows the taps of a valid position can reach
  // end there) and entries l >= vid_len[v] of st_out / ed_out -- exactly 0 after the masked softmax -- are NOT WRITTEN: the
  // consumer (xml_moment_topk_ex with the same vid_len) does not read them.  TVR: 51 of 128 clip rows on average.
  const int32_t* vid_len;
};

// 3 waves per SIMD (<= 168 VGPRs) for the f32 / bf16 instantiations.  The split-f16 one holds both halves of every fragment:
// at 3 waves it spills 24 VGPRs, at 2 (191 VGPRs, no scratch) it is 1.6 % faster (4.54 vs 4.62 ms, same box).
template <typename T>
__global__ __launch_bounds__(256, IsSplit16<T>::value ? 2 : 3) void convse_kernel(ConvseArgs a) {
  using Cfg = GemmCfg<T, TM, 128, 1, 4>;
  // dynamic LDS: [ GEMM staging | similarity patches ].  With ONE similarity patch (merged streams or a single
  // modality) the patch is written once after the last GEMM and overlays the staging area (49 KiB -> 3 workgroups
  // per CU); two independent streams need their patches while the second GEMM still stages.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int32_t s_pair[TM];
  const int n_sim_l = a.merged ? 1 : a.n_mod;
  float (*sim)[TM][LP] = reinterpret_cast<float (*)[TM][LP]>(n_sim_l == 1 ? smem : smem + Cfg::LDS_BYTES);
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x;
  if (chunk >= a.chunk_off[a.nv]) return;
  const int v = a.chunk_vid[chunk];                // video owning this chunk
  const int first = a.offsets[(int64_t)v * a.sub] + (chunk - a.chunk_off[v]) * TM;
  const int cnt = min(TM, a.offsets[(int64_t)(v + 1) * a.sub] - first);
  const int vlen = a.vid_len ? max(0, min(a.vid_len[v], a.l_ref)) : a.l_ref;          // clips whose outputs are stored
  const int b_rows = a.vid_len ? max(1, min(a.lpad, vlen + (a.ksize >> 1))) : a.lpad;   // clip rows the GEMM needs
  constexpr bool SPLIT = IsSplit16<T>::value;
  __shared__ float s_qinv[SPLIT ? 2 : 1][SPLIT ? TM : 1];
  if (tid < TM) {
    const int p = tid < cnt ? a.bucket[first + tid] : -1;
    s_pair[tid] = p;
    if constexpr (SPLIT) {
#pragma unroll
      for (int m = 0; m < 2; ++m) s_qinv[m][tid] = (p >= 0 && m < a.n_mod) ? a.q_inv[m][p / a.kpairs] : 1.f;
    }
  }
  __syncthreads();

  const int lane = tid & 63, wn = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int n_sim = a.merged ? 1 : a.n_mod;
  const int mt_used = (cnt + 15) >> 4;             // a video has ~46 pairs at the TVR shape: 3 of the 4 row tiles
  f32x4 acc[Cfg::MT][Cfg::NT];
  for (int m = 0; m < a.n_mod; ++m) {
    float cinv[Cfg::NT];                           // SPLIT: 1 / S of this lane's clip columns, this modality
#pragma unroll
    for (int nt = 0; nt < Cfg::NT; ++nt) {
      cinv[nt] = 1.f;
      if constexpr (SPLIT) cinv[nt] = a.c_inv[m][(int64_t)v * a.lpad + min(wn * 32 + nt * 16 + fr, a.lpad - 1)];
    }
    const T* ql = reinterpret_cast<const T*>(a.q_lin[m]);
    const T* f2 = reinterpret_cast<const T*>(a.feat2[m]);
    auto a_row = [&](int r) -> const char* {
      const int p = s_pair[r];
      return p >= 0 ? reinterpret_cast<const char*>(ql + (int64_t)(p / a.kpairs) * a.hidden) : nullptr;
    };
    auto b_row = [&](int r) -> const char* {
      return r < b_rows ? reinterpret_cast<const char*>(f2 + ((int64_t)v * a.lpad + r) * a.hidden) : nullptr;
    };
    const int k_bytes = a.hidden * (int)sizeof(T);
    auto a_off = [&](int r) -> uint32_t {             // rows beyond the chunk: any valid row (their products are never read)
      const int p = s_pair[r] >= 0 ? s_pair[r] : s_pair[0];
      return (uint32_t)(p / a.kpairs) * (uint32_t)k_bytes;
    };
    // (rows the taps of a valid clip cannot reach re-read the last needed row: an L1 / L2 hit instead of an HBM fetch)
    auto b_off = [&](int r) -> uint32_t { return (uint32_t)min(r, b_rows - 1) * (uint32_t)k_bytes; };
    const char* b_base = reinterpret_cast<const char*>(f2 + (int64_t)v * a.lpad * a.hidden);
    if (a.dbg == 1) {
#pragma unroll
      for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if (a.dma) {
      // two-stage ring: 49 KiB of staging = three workgroups per CU.  (Three stages / two workgroups measured 8 % SLOWER
      // than the register-staged loop in bf16 -- the epilogue of one workgroup needs the GEMM phases of two others to
      // hide behind --, two stages 1 % / 7 % faster in bf16 / f32: tools/bench_k7.py, ablations 40 / 41.)
      if (m > 0) __syncthreads();                     // the previous modality's last step is still being read
      if (a.merged && m > 0)
        gemm_mainloop_dma<T, Cfg, false, 2>(acc, reinterpret_cast<const char*>(ql), a_off, b_base, b_off, k_bytes, smem, mt_used);
      else
        gemm_mainloop_dma<T, Cfg, true, 2>(acc, reinterpret_cast<const char*>(ql), a_off, b_base, b_off, k_bytes, smem, mt_used);
    } else if (a.merged && m > 0)
      gemm_mainloop<T, Cfg, false>(acc, a_row, b_row, a.hidden * (int)sizeof(T), smem, mt_used);
    else
      gemm_mainloop<T, Cfg, true>(acc, a_row, b_row, a.hidden * (int)sizeof(T), smem, mt_used);
    if constexpr (SPLIT) {
      if (a.merged && m + 1 < a.n_mod) {
        // the next modality accumulates on top: move the accumulators into ITS units (exact: ratios of powers of two)
        float cnext[Cfg::NT];
#pragma unroll
        for (int nt = 0; nt < Cfg::NT; ++nt)
          cnext[nt] = cinv[nt] / a.c_inv[m + 1][(int64_t)v * a.lpad + min(wn * 32 + nt * 16 + fr, a.lpad - 1)];
#pragma unroll
        for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = mt * 16 + fg * 4 + r;
            const float qr = s_qinv[m][row] / s_qinv[m + 1][row];
#pragma unroll
            for (int nt = 0; nt < Cfg::NT; ++nt) acc[mt][nt][r] *= qr * cnext[nt];
          }
      }
    }
    if (!a.merged || m == a.n_mod - 1) {
      const float scale = (a.merged && a.n_mod == 2) ? 0.5f : 1.f;
      const int si = a.merged ? 0 : m;
      if (n_sim_l == 1) __syncthreads();     // every wave is done with the staging area the patch overlays
#pragma unroll
      for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = acc[mt][nt][r] * scale;
            if constexpr (SPLIT) x *= s_qinv[m][mt * 16 + fg * 4 + r] * cinv[nt];
            sim[si][mt * 16 + fg * 4 + r][LH + wn * 32 + nt * 16 + fr] = x;
          }
    }
  }
  __syncthreads();

  if (a.dbg == 2) return;
  // ---- ConvSE epilogue: each wave owns TM/4 pair rows; a lane owns clips lane and lane + 64 ----------
  // Everything that does not depend on the pair row is fetched ONCE per workgroup: the clip masks of this video (two
  // values per lane and stream) and the filter taps (scalar loads into LDS).  Left inside the row loop they were a
  // dependent global load per row -- 16 round trips per wave, most of the epilogue's time.
  const int half = a.ksize >> 1;
  const float inv_mod = 1.f / (float)n_sim;
  __shared__ float s_taps[2 * 2 * 16];               // [st | ed][stream][tap]
  if (tid < 2 * n_sim * a.ksize) {
    const int which = tid / (n_sim * a.ksize), rest = tid - which * n_sim * a.ksize;
    const int si = rest / a.ksize, t = rest - si * a.ksize;
    s_taps[(which * 2 + si) * 16 + t] = a.conv_w[(which * n_sim + si) * a.ksize + t];
  }
  float mk_r[2][2];
#pragma unroll
  for (int si = 0; si < 2; ++si)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int l = lane + h * 64;
      mk_r[si][h] = (si < n_sim && l < a.l_ref) ? a.mask[a.merged ? 0 : si][(int64_t)v * a.lpad + l] : 0.f;
    }
  __syncthreads();
  if (a.ksize == 5 && !a.summ) {
    // ---- fast epilogue (5 taps, no candidate summaries): HALF a wave per pair row, a lane owns 4 CONSECUTIVE clips ------
    // The row loop below costs a wave ~30 dependent LDS reads, four 64-lane reductions and four 256-byte stores per pair row
    // and was 0.6 of K7's 1.1 ms at the as-trained shape (tools/bench_k7.py, ablation 2).  Here a lane reads its 4 clips +
    // the 2 + 2 halo values once (four 8-byte LDS reads per stream), runs the 2 x 4 x 5 multiply-adds from registers, the
    // reductions span 32 lanes, two rows are in flight per wave instruction, and a row leaves as two 16-byte stores per lane.
    const int hsel = lane >> 5, l4 = (lane & 31) * 4;
    float tw[2][2][5];                                  // [st | ed][stream][tap]
#pragma unroll
    for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
      for (int si = 0; si < 2; ++si)
#pragma unroll
        for (int t = 0; t < 5; ++t) tw[w2][si][t] = si < n_sim ? s_taps[(w2 * 2 + si) * 16 + t] : 0.f;
    float mk4[2][4];
#pragma unroll
    for (int si = 0; si < 2; ++si)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        mk4[si][c] = (si < n_sim && l4 + c < a.l_ref) ? a.mask[a.merged ? 0 : si][(int64_t)v * a.lpad + l4 + c] : 0.f;
    const int store_end = a.vid_len ? ((vlen + 3) & ~3) : a.lpad;     // (whole 16-byte pieces: up to 3 exact zeros past vlen)
    for (int row = wn * 2 + hsel; row < ((cnt + 7) & ~7); row += 8) {
      const bool live = row < cnt;
      const int rr = live ? row : 0;
      const int p = s_pair[rr];
      float sv[4] = {0.f, 0.f, 0.f, 0.f}, ev[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int si = 0; si < 2; ++si) {
        if (si < n_sim) {
          float x[8];
          const float2* px = reinterpret_cast<const float2*>(&sim[si][rr][LH + l4 - 2]);       // 8-byte aligned
#pragma unroll
          for (int i = 0; i < 4; ++i) { const float2 t2 = px[i]; x[2 * i] = t2.x; x[2 * i + 1] = t2.y; }
#pragma unroll
          for (int i = 0; i < 8; ++i) { const int j = l4 - 2 + i; x[i] = (j >= 0 && j < a.l_ref) ? x[i] : 0.f; }   // zero padding at L
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float cs = 0.f, ce = 0.f;
#pragma unroll
            for (int t = 0; t < 5; ++t) { cs += tw[0][si][t] * x[c + t]; ce += tw[1][si][t] * x[c + t]; }
            const float fill = (1.f - mk4[si][c]) * -1e10f;
            sv[c] += cs * mk4[si][c] + fill;            // mask_logits, xml/model_xml.py:640-641
            ev[c] += ce * mk4[si][c] + fill;
          }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (n_sim > 1) { sv[c] *= inv_mod; ev[c] *= inv_mod; }
        if (l4 + c >= a.l_ref) { sv[c] = -INFINITY; ev[c] = -INFINITY; }
      }
      if (a.softmax) {
        float ms = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), me = fmaxf(fmaxf(ev[0], ev[1]), fmaxf(ev[2], ev[3]));
        ms = lane16_max_dpp(ms); me = lane16_max_dpp(me);
        ms = fmaxf(ms, __shfl_xor(ms, 16, 64)); me = fmaxf(me, __shfl_xor(me, 16, 64));
        float ss = 0.f, se = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) { sv[c] = expf(sv[c] - ms); ev[c] = expf(ev[c] - me); ss += sv[c]; se += ev[c]; }
        ss = lane16_sum_dpp(ss); se = lane16_sum_dpp(se);
        ss += __shfl_xor(ss, 16, 64); se += __shfl_xor(se, 16, 64);
#pragma unroll
        for (int c = 0; c < 4; ++c) { sv[c] /= ss; ev[c] /= se; }
      }
      if (live && l4 < store_end) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (l4 + c >= a.l_ref) { sv[c] = 0.f; ev[c] = 0.f; }
        *reinterpret_cast<float4*>(a.st_out + (int64_t)p * a.lpad + l4) = make_float4(sv[0], sv[1], sv[2], sv[3]);
        *reinterpret_cast<float4*>(a.ed_out + (int64_t)p * a.lpad + l4) = make_float4(ev[0], ev[1], ev[2], ev[3]);
      }
    }
    return;
  }
  for (int row = wn; row < cnt; row += 4) {
    const int p = s_pair[row];
    float st[2], ed[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int l = lane + h * 64;
      float s_acc = 0.f, e_acc = 0.f;
      if (l < a.l_ref) {
#pragma unroll
        for (int si = 0; si < 2; ++si) {
          if (si < n_sim) {
            const float* wst = s_taps + si * 16;
            const float* wed = s_taps + (2 + si) * 16;
            float cs = 0.f, ce = 0.f;
            for (int t = 0; t < a.ksize; ++t) {
              const int j = l + t - half;
              const float x = (j >= 0 && j < a.l_ref) ? sim[si][row][LH + j] : 0.f;
              cs += wst[t] * x;
              ce += wed[t] * x;
            }
            const float mk = mk_r[si][h];
            const float fill = (1.f - mk) * -1e10f;
            s_acc += cs * mk + fill;   // mask_logits, xml/model_xml.py:640-641
            e_acc += ce * mk + fill;
          }
        }
        if (n_sim > 1) { s_acc *= inv_mod; e_acc *= inv_mod; }
      } else {
        s_acc = -INFINITY; e_acc = -INFINITY;
      }
      st[h] = s_acc; ed[h] = e_acc;
    }
    if (a.softmax) {
      const float ms = wave_max(fmaxf(st[0], st[1])), me = wave_max(fmaxf(ed[0], ed[1]));
      float es[2], ee[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) { es[h] = expf(st[h] - ms); ee[h] = expf(ed[h] - me); }
      const float ss = wave_sum(es[0] + es[1]), se = wave_sum(ee[0] + ee[1]);
#pragma unroll
      for (int h = 0; h < 2; ++h) { st[h] = es[h] / ss; ed[h] = ee[h] / se; }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int l = lane + h * 64;
      if (a.vid_len ? l < vlen : l < a.lpad) {
        const bool in = l < a.l_ref;
        a.st_out[(int64_t)p * a.lpad + l] = in ? st[h] : 0.f;
        a.ed_out[(int64_t)p * a.lpad + l] = in ? ed[h] : 0.f;
      }
    }
    if (a.summ) {
      // K9's row maxima, from the very values just stored (same two f32 multiplies, so the same bits K9 would compute):
      // a = st * w, m = a * (window maximum of ed); then the 8 largest of the pair's 128, one wave maximum each
      const float wv = a.pair_w ? a.pair_w[p] : 1.f;
      const bool in0 = lane < a.l_ref, in1 = lane + 64 < a.l_ref;
      float e_lo = in0 ? ed[0] : 0.f, e_hi = in1 ? ed[1] : 0.f;
      band_window_max(e_lo, e_hi, a.max_l - a.min_l, a.min_l, lane);
      const float m_lo = fmaxf((in0 ? st[0] : 0.f) * wv * e_lo, 0.f), m_hi = fmaxf((in1 ? st[1] : 0.f) * wv * e_hi, 0.f);
      // ... and 8 of them, one per group of 16 rows (lanes 8 g .. 8 g + 7 hold rows 8 g .. + 7 and 64 + 8 g .. + 7): the
      // group maxima, three DPP steps.  (The 8 LARGEST of the pair -- eight sequential wave maxima -- cost K7 0.45-0.75 ms
      // per pass and bought K9 0.2: measured, profiles/r04_notes.md.)  Disjoint groups = distinct rows, which is all
      // K9's bound needs.
      float mg = fmaxf(m_lo, m_hi);
      mg = fmaxf(mg, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mg), 0xB1, 0xf, 0xf, false)));    // quad_perm [1,0,3,2]
      mg = fmaxf(mg, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mg), 0x4E, 0xf, 0xf, false)));    // quad_perm [2,3,0,1]
      mg = fmaxf(mg, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mg), 0x141, 0xf, 0xf, false)));   // row_half_mirror
      if ((lane & 7) == 0) a.summ[(int64_t)p * XML_MOMENT_SUMM + (lane >> 3)] = mg;
    }
  }
}

// ---- exact-rank rescoring (xml_q2c_rescore) -----------------------------------------------------------------------
// Video-level scores of the LISTED (query, video) pairs only, in the storage type of the operands (f32 for the exact-rank
// mode: the bf16 K6 pass is a filter, its top-M candidates are re-scored here against the f32 index).  Same inverted pair
// list as K7 (a video's feat1 tile is fetched once per 64-pair chunk), same MFMA mainloop; the epilogue is K6's:
//   out[q, j] = ( sum_m max_l mask_logits( qn_m[q] . cn_m[v, l] ) ) / n_mod        (xml/model_xml.py:448-452, :572-574)
// taken straight from the accumulators (a wave holds 32 clip columns of all 64 pair rows): DPP row maximum over the 16
// lanes of a column group, then a 4 x 64 LDS patch across the four waves -- no similarity patch, 49 KiB of LDS.
struct RescoreArgs {
  const void* qn[2];
  const void* cn[2];
  const float* mask[2];
  float* out;
  const int32_t* offsets;
  const int32_t* chunk_off;
  const int32_t* bucket;
  const int32_t* chunk_vid;
  int nv, kpairs, lpad, hidden, n_mod;
  int sub;   // see ConvseArgs
  int dma;   // see ConvseArgs
};

static constexpr int RS_STAGES = 2;   // re-score ring depth: 2 stages = 49 KiB = three workgroups per CU (f32 MFMA-bound: waves per SIMD matter more than depth)
// TM = pairs per chunk: 64 (three workgroups per CU) or 128 -- with 256 candidates per query a video is listed by ~117
// queries, and one 128-row chunk fetches its tile ONCE where two 64-row chunks fetch it twice (the kernel is bound by
// those fetches: 393 KiB per modality and chunk in split-f16 form)
template <typename T, int TM = 64>
__global__ __launch_bounds__(256, TM == 64 ? 3 : 2) void rescore_kernel(RescoreArgs a) {
  using Cfg = GemmCfg<T, TM, 128, 1, 4>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int32_t s_pair[TM];
  __shared__ float s_max[4][TM];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x;
  if (chunk >= a.chunk_off[a.nv]) return;
  const int v = a.chunk_vid[chunk];
  const int first = a.offsets[(int64_t)v * a.sub] + (chunk - a.chunk_off[v]) * TM;
  const int cnt = min(TM, a.offsets[(int64_t)(v + 1) * a.sub] - first);
  if (tid < TM) s_pair[tid] = tid < cnt ? a.bucket[first + tid] : -1;
  __syncthreads();
  const int lane = tid & 63, wn = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  float tot = 0.f;                                   // threads < TM: sum over modalities of the row maximum
  f32x4 acc[Cfg::MT][Cfg::NT];
  for (int m = 0; m < a.n_mod; ++m) {
    const T* qn = reinterpret_cast<const T*>(a.qn[m]);
    const T* cn = reinterpret_cast<const T*>(a.cn[m]);
    auto a_row = [&](int r) -> const char* {
      const int p = s_pair[r];
      return p >= 0 ? reinterpret_cast<const char*>(qn + (int64_t)(p / a.kpairs) * a.hidden) : nullptr;
    };
    auto b_row = [&](int r) -> const char* {
      return r < a.lpad ? reinterpret_cast<const char*>(cn + ((int64_t)v * a.lpad + r) * a.hidden) : nullptr;
    };
    float mk[Cfg::NT];
#pragma unroll
    for (int nt = 0; nt < Cfg::NT; ++nt) {
      const int c = wn * 32 + nt * 16 + fr;
      mk[nt] = c < a.lpad ? a.mask[m][(int64_t)v * a.lpad + c] : 0.f;
    }
    if (a.dma) {
      const int k_bytes = a.hidden * (int)sizeof(T);
      auto a_off = [&](int r) -> uint32_t {
        const int p = s_pair[r] >= 0 ? s_pair[r] : s_pair[0];
        return (uint32_t)(p / a.kpairs) * (uint32_t)k_bytes;
      };
      auto b_off = [&](int r) -> uint32_t { return (uint32_t)min(r, a.lpad - 1) * (uint32_t)k_bytes; };
      // (the barrier behind the previous modality's row maxima also frees its last ring stage)
      gemm_mainloop_dma<T, Cfg, true, RS_STAGES>(acc, reinterpret_cast<const char*>(qn), a_off,
                                                 reinterpret_cast<const char*>(cn + (int64_t)v * a.lpad * a.hidden), b_off,
                                                 k_bytes, smem, (cnt + 15) >> 4);
    } else {
      gemm_mainloop<T, Cfg, true>(acc, a_row, b_row, a.hidden * (int)sizeof(T), smem, (cnt + 15) >> 4);
    }
#pragma unroll
    for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float best = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < Cfg::NT; ++nt) {
          float x = acc[mt][nt][r];
          // XML_F16S: both operands are unit-norm rows at the fixed scale 2^XML_F16_UNIT_LOG2
          if constexpr (IsSplit16<T>::value) x *= 1.f / (float)(1u << (2 * XML_F16_UNIT_LOG2));
          best = fmaxf(best, x * mk[nt] + (1.f - mk[nt]) * -1e10f);      // mask_logits
        }
        best = lane16_max_dpp(best);
        if (fr == 0) s_max[wn][mt * 16 + fg * 4 + r] = best;
      }
    __syncthreads();
    if (tid < TM) tot += fmaxf(fmaxf(s_max[0][tid], s_max[1][tid]), fmaxf(s_max[2][tid], s_max[3][tid]));
    // (the next modality's mainloop passes several barriers before anyone writes s_max again)
  }
  if (tid < cnt) a.out[s_pair[tid]] = a.n_mod == 2 ? tot * 0.5f : tot;
}

__global__ void rescore_fill_skipped_kernel(const int32_t* __restrict__ pair_vid, float* __restrict__ out, int64_t P,
                                            int nv) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int v = pair_vid[p];
  if (v < 0 || v >= nv) out[p] = -INFINITY;
}

__global__ void convse_zero_words_kernel(uint32_t* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}

struct ConvseSumm {          // optional candidate summaries / ragged-corpus lengths (xml_convse_rerank_ex)
  const float* pair_w;
  float* summ;
  int min_l, max_l;
  const int32_t* vid_len;
};
static int convse_rerank_impl(const xml_convse_desc* d, const void* q_lin0, const void* q_lin1, const void* feat2_0,
                              const void* feat2_1, const float* mask0, const float* mask1, const int32_t* pair_vid,
                              const float* conv_w, float* st_out, float* ed_out, void* ws, size_t ws_bytes,
                              const float* q_inv0, const float* q_inv1, const float* c_inv0, const float* c_inv1,
                              xml_stream_t stream, ConvseSumm sm = ConvseSumm{nullptr, nullptr, 0, 0, nullptr});

// xml_convse_rerank / xml_convse_rerank_f16s (by desc.dt) + the candidate summaries K9 starts from
extern "C" int xml_convse_rerank_ex(const xml_convse_desc* d, const void* q_lin0, const void* q_lin1, const float* q_inv0,
                                    const float* q_inv1, const void* feat2_0, const void* feat2_1, const float* c_inv0,
                                    const float* c_inv1, const float* mask0, const float* mask1, const int32_t* pair_vid,
                                    const float* conv_w, const float* pair_w, int min_l, int max_l,
                                    const int32_t* vid_len, float* st_out, float* ed_out, float* summ_out, void* ws,
                                    size_t ws_bytes, xml_stream_t stream) {
  XML_ENTER();
  if (!d || !q_lin0 || !feat2_0 || !mask0 || !pair_vid || !conv_w || !st_out || !ed_out || !ws) return XML_ERR_BAD_ARG;
  if (!summ_out && !vid_len) return XML_ERR_BAD_ARG;          // (neither extension asked for: xml_convse_rerank / _f16s)
  if (d->nq <= 0 || d->nv <= 0 || d->kpairs <= 0 || d->hidden <= 0) return XML_ERR_BAD_ARG;
  if (d->n_mod < 1 || d->n_mod > 2 || (d->n_mod == 2 && (!q_lin1 || !feat2_1))) return XML_ERR_BAD_ARG;
  if (d->n_mod == 2 && !d->merged && !mask1) return XML_ERR_BAD_ARG;
  if (d->merged && d->n_mod != 2) return XML_ERR_BAD_ARG;
  if (d->lpad % 16 || d->lpad > 128 || d->l_ref > d->lpad || d->l_ref <= 0 || d->hidden % 8) return XML_ERR_UNSUPPORTED;
  if (!(d->ksize & 1) || d->ksize > 15 || d->ksize < 1) return XML_ERR_UNSUPPORTED;
  if (summ_out && (min_l < 0 || max_l <= min_l)) return XML_ERR_BAD_ARG;
  if (!(d->softmax & 1)) return XML_ERR_BAD_ARG;      // summaries are of PROBABILITIES; unwritten entries are exact zeros of a softmax
  if (d->dt == XML_F16S) {
    if (!q_inv0 || !c_inv0 || (d->n_mod == 2 && (!q_inv1 || !c_inv1))) return XML_ERR_BAD_ARG;
    if (d->hidden % 32) return XML_ERR_UNSUPPORTED;
  } else if (d->dt != XML_F32 && d->dt != XML_BF16) {
    return XML_ERR_BAD_ARG;
  }
  return convse_rerank_impl(d, q_lin0, q_lin1, feat2_0, feat2_1, mask0, mask1, pair_vid, conv_w, st_out, ed_out, ws, ws_bytes,
                            d->dt == XML_F16S ? q_inv0 : nullptr, d->dt == XML_F16S ? q_inv1 : nullptr,
                            d->dt == XML_F16S ? c_inv0 : nullptr, d->dt == XML_F16S ? c_inv1 : nullptr, stream,
                            ConvseSumm{pair_w, summ_out, min_l, max_l, vid_len});
}

extern "C" int xml_convse_rerank(const xml_convse_desc* d, const void* q_lin0, const void* q_lin1,
                                 const void* feat2_0, const void* feat2_1, const float* mask0, const float* mask1,
                                 const int32_t* pair_vid, const float* conv_w, float* st_out, float* ed_out, void* ws,
                                 size_t ws_bytes, xml_stream_t stream) {
  XML_ENTER();
  if (!d || !q_lin0 || !feat2_0 || !mask0 || !pair_vid || !conv_w || !st_out || !ed_out || !ws) return XML_ERR_BAD_ARG;
  if (d->nq <= 0 || d->nv <= 0 || d->kpairs <= 0 || d->hidden <= 0) return XML_ERR_BAD_ARG;
  if (d->n_mod < 1 || d->n_mod > 2 || (d->n_mod == 2 && (!q_lin1 || !feat2_1))) return XML_ERR_BAD_ARG;
  if (d->n_mod == 2 && !d->merged && !mask1) return XML_ERR_BAD_ARG;
  if (d->merged && d->n_mod != 2) return XML_ERR_BAD_ARG;
  if (d->lpad % 16 || d->lpad > 128 || d->l_ref > d->lpad || d->l_ref <= 0 || d->hidden % 8) return XML_ERR_UNSUPPORTED;
  if (!(d->ksize & 1) || d->ksize > 15 || d->ksize < 1) return XML_ERR_UNSUPPORTED;
  if (d->dt != XML_F32 && d->dt != XML_BF16) return XML_ERR_BAD_ARG;
  return convse_rerank_impl(d, q_lin0, q_lin1, feat2_0, feat2_1, mask0, mask1, pair_vid, conv_w, st_out, ed_out, ws, ws_bytes,
                            nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int xml_convse_rerank_f16s(const xml_convse_desc* d, const void* q_lin0, const void* q_lin1, const float* q_inv0,
                                      const float* q_inv1, const void* feat2_0, const void* feat2_1, const float* c_inv0,
                                      const float* c_inv1, const float* mask0, const float* mask1, const int32_t* pair_vid,
                                      const float* conv_w, float* st_out, float* ed_out, void* ws, size_t ws_bytes,
                                      xml_stream_t stream) {
  XML_ENTER();
  if (!d || !q_lin0 || !feat2_0 || !mask0 || !pair_vid || !conv_w || !st_out || !ed_out || !ws || !q_inv0 || !c_inv0)
    return XML_ERR_BAD_ARG;
  if (d->nq <= 0 || d->nv <= 0 || d->kpairs <= 0 || d->hidden <= 0) return XML_ERR_BAD_ARG;
  if (d->n_mod < 1 || d->n_mod > 2 || (d->n_mod == 2 && (!q_lin1 || !feat2_1 || !q_inv1 || !c_inv1))) return XML_ERR_BAD_ARG;
  if (d->n_mod == 2 && !d->merged && !mask1) return XML_ERR_BAD_ARG;
  if (d->merged && d->n_mod != 2) return XML_ERR_BAD_ARG;
  if (d->lpad % 16 || d->lpad > 128 || d->l_ref > d->lpad || d->l_ref <= 0 || d->hidden % 32) return XML_ERR_UNSUPPORTED;
  if (!(d->ksize & 1) || d->ksize > 15 || d->ksize < 1) return XML_ERR_UNSUPPORTED;
  if (d->dt != XML_F16S) return XML_ERR_BAD_ARG;
  return convse_rerank_impl(d, q_lin0, q_lin1, feat2_0, feat2_1, mask0, mask1, pair_vid, conv_w, st_out, ed_out, ws, ws_bytes,
                            q_inv0, q_inv1, c_inv0, c_inv1, stream);
}

static int convse_rerank_impl(const xml_convse_desc* d, const void* q_lin0, const void* q_lin1, const void* feat2_0,
                              const void* feat2_1, const float* mask0, const float* mask1, const int32_t* pair_vid,
                              const float* conv_w, float* st_out, float* ed_out, void* ws, size_t ws_bytes,
                              const float* q_inv0, const float* q_inv1, const float* c_inv0, const float* c_inv1,
                              xml_stream_t stream, ConvseSumm sm) {
  if (ws_bytes < xml_convse_rerank_workspace_bytes(d)) return XML_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  ConvseWs w;
  convse_ws_layout(d, &w, (char*)ws);
  const int64_t P = (int64_t)d->nq * d->kpairs;
  int sub = 1;
  if (P <= CONVSE_SMALL_P && d->nv <= CONVSE_SMALL_NV) {
    hipLaunchKernelGGL(convse_invert_small_kernel, dim3(1), dim3(1024), 0, st, pair_vid, w.offsets, w.chunk_off, w.chunk_vid,
                       w.pos, w.bucket, st_out, ed_out, (int)P, d->nv, TM, d->lpad / 4, (d->softmax & 2) ? 0 : 1);
    XML_CHECK_LAUNCH();
  } else {
    // counts is the first (256-aligned) region.  Zeroed by a kernel, not hipMemsetAsync:
    // as a memset NODE of a captured HIP graph the reset did not happen on the second replay (stale cursors ->
    // out-of-bounds bucket writes, seen with inference.GraphedVcmrSearch); a kernel node replays faithfully.
    {
      const int64_t nwords = (int64_t)((char*)w.offsets - (char*)w.counts) / 4;
      hipLaunchKernelGGL(convse_zero_words_kernel, dim3(cdiv(nwords, 256)), dim3(256), 0, st, (uint32_t*)w.counts, nwords);
      XML_CHECK_LAUNCH();
    }
    sub = convse_sub(P, d->nv);
    hipLaunchKernelGGL(convse_count_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, pair_vid, w.counts, w.pos, P, d->nv, sub);
    XML_CHECK_LAUNCH();
    if (!(d->softmax & 2)) {     // bit 1: the caller never reads the rows of skipped pairs
      hipLaunchKernelGGL(convse_zero_skipped_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, pair_vid, st_out, ed_out, P,
                         d->nv, d->lpad / 4);
      XML_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(convse_scan_kernel, dim3(1), dim3(1024), 0, st, w.counts, w.offsets, w.chunk_off, w.chunk_vid,
                       d->nv, TM, sub);
    XML_CHECK_LAUNCH();
    hipLaunchKernelGGL(convse_fill_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, pair_vid, w.offsets, w.pos, w.bucket, P, sub);
    XML_CHECK_LAUNCH();
  }
  ConvseArgs a;
  a.sub = sub;
  a.q_lin[0] = q_lin0; a.q_lin[1] = q_lin1;
  a.feat2[0] = feat2_0; a.feat2[1] = feat2_1;
  a.mask[0] = mask0; a.mask[1] = mask1 ? mask1 : mask0;
  a.conv_w = conv_w; a.st_out = st_out; a.ed_out = ed_out;
  a.offsets = w.offsets; a.chunk_off = w.chunk_off; a.bucket = w.bucket; a.chunk_vid = w.chunk_vid;
  a.nv = d->nv; a.kpairs = d->kpairs; a.lpad = d->lpad; a.l_ref = d->l_ref; a.hidden = d->hidden;
  a.n_mod = d->n_mod; a.merged = d->merged; a.ksize = d->ksize; a.softmax = d->softmax & 1;
  a.dbg = g_q2c_ablation;       // constant 0 in the product build (debug.h)
  a.pair_w = sm.pair_w; a.summ = sm.summ; a.min_l = sm.min_l; a.max_l = sm.max_l; a.vid_len = sm.vid_len;
  a.q_inv[0] = q_inv0; a.q_inv[1] = q_inv1 ? q_inv1 : q_inv0;
  a.c_inv[0] = c_inv0; a.c_inv[1] = c_inv1 ? c_inv1 : c_inv0;
  {
    const int64_t kb = (int64_t)d->hidden * (int64_t)dt_size(d->dt);
    a.dma = (kb % 128 == 0 && (int64_t)d->nq * kb < (1ll << 32) && (int64_t)d->lpad * kb < (1ll << 32)) ? 1 : 0;
    if (g_q2c_ablation == 40) a.dma = 0;             // (debug build: A/B against the register-staged mainloop)
  }
  const int64_t max_chunks = P / TM + (P < d->nv ? P : d->nv);
  const int n_sim = d->merged ? 1 : d->n_mod;
  const size_t patch = (size_t)TM * LP * 4;
  if (d->dt == XML_F16S) {
    if (!a.dma) return XML_ERR_UNSUPPORTED;          // split rows are whole 128-byte steps by construction (hidden % 32 == 0)
    using Cfg = GemmCfg<f16s_t, TM, 128, 1, 4>;
    constexpr size_t stg = Cfg::LDS_BYTES;
    const size_t lds = n_sim == 1 ? (stg > patch ? stg : patch) : stg + 2 * patch;
    if (!xml_lds_attr_once<convse_kernel<f16s_t>>((int)(stg + 2 * patch))) return XML_ERR_LAUNCH;
    hipLaunchKernelGGL(convse_kernel<f16s_t>, dim3((unsigned)max_chunks), dim3(256), lds, st, a);
  } else if (d->dt == XML_F32) {
    using Cfg = GemmCfg<float, TM, 128, 1, 4>;
    constexpr size_t stg = Cfg::LDS_BYTES;           // = GemmDma<Cfg, 2>::LDS_BYTES: both mainloops stage two steps
    const size_t lds = n_sim == 1 ? (stg > patch ? stg : patch) : stg + 2 * patch;
    if (!xml_lds_attr_once<convse_kernel<float>>((int)(stg + 2 * patch))) return XML_ERR_LAUNCH;
    hipLaunchKernelGGL(convse_kernel<float>, dim3((unsigned)max_chunks), dim3(256), lds, st, a);
  } else {
    using Cfg = GemmCfg<bf16_t, TM, 128, 1, 4>;
    constexpr size_t stg = Cfg::LDS_BYTES;           // = GemmDma<Cfg, 2>::LDS_BYTES: both mainloops stage two steps
    const size_t lds = n_sim == 1 ? (stg > patch ? stg : patch) : stg + 2 * patch;
    if (!xml_lds_attr_once<convse_kernel<bf16_t>>((int)(stg + 2 * patch))) return XML_ERR_LAUNCH;
    hipLaunchKernelGGL(convse_kernel<bf16_t>, dim3((unsigned)max_chunks), dim3(256), lds, st, a);
  }
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" size_t xml_q2c_rescore_workspace_bytes(int nq, int nv, int kpairs) {
  xml_convse_desc d{};
  d.nq = nq; d.nv = nv; d.kpairs = kpairs;
  return convse_ws_layout(&d, nullptr, nullptr);
}

extern "C" int xml_q2c_rescore(int n_mod, const void* qn0, const void* qn1, const void* cn0, const void* cn1,
                               const float* mask0, const float* mask1, const int32_t* pair_vid, float* out, int nq, int nv,
                               int kpairs, int lpad, int hidden, int dt, void* ws, size_t ws_bytes, xml_stream_t stream) {
  XML_ENTER();
  if (!qn0 || !cn0 || !mask0 || !pair_vid || !out || !ws) return XML_ERR_BAD_ARG;
  if (n_mod < 1 || n_mod > 2 || (n_mod == 2 && (!qn1 || !cn1 || !mask1))) return XML_ERR_BAD_ARG;
  if (nq <= 0 || nv <= 0 || kpairs <= 0 || hidden <= 0) return XML_ERR_BAD_ARG;
  if (lpad % 16 || lpad > 128 || lpad <= 0 || hidden % 8) return XML_ERR_UNSUPPORTED;
  if (dt != XML_F32 && dt != XML_BF16 && dt != XML_F16S) return XML_ERR_BAD_ARG;
  if (dt == XML_F16S && hidden % 32) return XML_ERR_UNSUPPORTED;
  if (ws_bytes < xml_q2c_rescore_workspace_bytes(nq, nv, kpairs)) return XML_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  xml_convse_desc d{};
  d.nq = nq; d.nv = nv; d.kpairs = kpairs;
  ConvseWs w;
  convse_ws_layout(&d, &w, (char*)ws);
  const int64_t P = (int64_t)nq * kpairs;
  {
    const int64_t nwords = (int64_t)((char*)w.offsets - (char*)w.counts) / 4;
    hipLaunchKernelGGL(convse_zero_words_kernel, dim3(cdiv(nwords, 256)), dim3(256), 0, st, (uint32_t*)w.counts, nwords);
    XML_CHECK_LAUNCH();
  }
  const int sub = convse_sub(P, nv);
  hipLaunchKernelGGL(convse_count_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, pair_vid, w.counts, w.pos, P, nv, sub);
  XML_CHECK_LAUNCH();
  hipLaunchKernelGGL(rescore_fill_skipped_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, pair_vid, out, P, nv);
  XML_CHECK_LAUNCH();
  // rows per chunk: 128 when the average video is listed by more than 64 pairs (then most videos would need two 64-row
  // chunks, each fetching the tile again)
  const bool big = dt == XML_F16S && P > 64 * (int64_t)nv;
  hipLaunchKernelGGL(convse_scan_kernel, dim3(1), dim3(1024), 0, st, w.counts, w.offsets, w.chunk_off, w.chunk_vid, nv,
                     big ? 128 : TM, sub);
  XML_CHECK_LAUNCH();
  hipLaunchKernelGGL(convse_fill_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, pair_vid, w.offsets, w.pos, w.bucket, P, sub);
  XML_CHECK_LAUNCH();
  RescoreArgs a;
  a.sub = sub;
  a.qn[0] = qn0; a.qn[1] = qn1; a.cn[0] = cn0; a.cn[1] = cn1;
  a.mask[0] = mask0; a.mask[1] = mask1 ? mask1 : mask0;
  a.out = out;
  a.offsets = w.offsets; a.chunk_off = w.chunk_off; a.bucket = w.bucket; a.chunk_vid = w.chunk_vid;
  a.nv = nv; a.kpairs = kpairs; a.lpad = lpad; a.hidden = hidden; a.n_mod = n_mod;
  {
    const int64_t kb = (int64_t)hidden * (int64_t)dt_size(dt);
    a.dma = (kb % 128 == 0 && (int64_t)nq * kb < (1ll << 32) && (int64_t)lpad * kb < (1ll << 32)) ? 1 : 0;
    if (g_q2c_ablation == 40) a.dma = 0;
  }
  const int64_t max_chunks = P / TM + (P < nv ? P : nv);
  if (dt == XML_F16S && big) {
    if (!a.dma) return XML_ERR_UNSUPPORTED;
    using Cfg = GemmCfg<f16s_t, 128, 128, 1, 4>;
    constexpr int lds = GemmDma<Cfg, RS_STAGES>::LDS_BYTES;
    if (!xml_lds_attr_once<rescore_kernel<f16s_t, 128>>(lds)) return XML_ERR_LAUNCH;
    const int64_t chunks128 = P / 128 + (P < nv ? P : nv);
    hipLaunchKernelGGL((rescore_kernel<f16s_t, 128>), dim3((unsigned)chunks128), dim3(256), lds, st, a);
  } else if (dt == XML_F16S) {
    using Cfg = GemmCfg<f16s_t, TM, 128, 1, 4>;
    constexpr int lds = GemmDma<Cfg, RS_STAGES>::LDS_BYTES;
    if (!xml_lds_attr_once<rescore_kernel<f16s_t>>(lds)) return XML_ERR_LAUNCH;
    hipLaunchKernelGGL(rescore_kernel<f16s_t>, dim3((unsigned)max_chunks), dim3(256), lds, st, a);
  } else if (dt == XML_F32) {
    using Cfg = GemmCfg<float, TM, 128, 1, 4>;
    constexpr int lds = GemmDma<Cfg, RS_STAGES>::LDS_BYTES;
    if (!xml_lds_attr_once<rescore_kernel<float>>(lds)) return XML_ERR_LAUNCH;
    hipLaunchKernelGGL(rescore_kernel<float>, dim3((unsigned)max_chunks), dim3(256), lds, st, a);
  } else {
    using Cfg = GemmCfg<bf16_t, TM, 128, 1, 4>;
    constexpr int lds = GemmDma<Cfg, RS_STAGES>::LDS_BYTES;
    if (!xml_lds_attr_once<rescore_kernel<bf16_t>>(lds)) return XML_ERR_LAUNCH;
    hipLaunchKernelGGL(rescore_kernel<bf16_t>, dim3((unsigned)max_chunks), dim3(256), lds, st, a);
  }
  XML_CHECK_LAUNCH();
  return XML_OK;
}
