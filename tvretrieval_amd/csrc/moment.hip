// K9 + K10 (+ K11): banded moment candidates and per-query top-n.
//   reference: einsum("qvm,qv,qvn->qvmn") * min/max-length mask, flat descending sort, [:max_before_nms]
//              xml/inference.py:365-386, generate_min_max_length_mask :170-192, index decode :423-431
//              SVMR variant: get_svmr_res_from_st_ed_probs :195-241 + utils/tensor_utils.py:133-141
//
// The reference materialises (Nq, 100, L, L) products and fully sorts 1.64 M values per query.  Only the band
// min_l <= j-i < max_l can be non-zero, so one workgroup per query enumerates the k*L*(max_l-min_l) band
// candidates from LDS-resident st/ed rows and keeps the best n_out:
//   1. stage (st*w) and ed rows in LDS (the product order (st*w)*ed is the reference's einsum order);
//   2. row maxima m(r,i) = max_d score(r,i,i+d) stay in registers; the n_out-th largest row maximum is a
//      lower bound T_lb of the n_out-th largest candidate (radix-select over <= 16 K register values);
//   3. only rows with m(r,i) >= T_lb are re-expanded; candidates >= T_lb go to an LDS list;
//   4. bitonic sort of the list on (score desc, flat index asc); emit n_out.
//   If the list overflows (flat score distributions), an exact radix-select over all candidates replaces 2-3.
// Zero products (masked clips) are not candidates: the reference's order among zeros is unspecified.
#include "common.h"

static constexpr int MT_CAP = 4096;   // LDS candidate list capacity
static constexpr int MT_RPT = 64;     // (pair, start) rows per thread -> kpairs * l_ref <= 16384

struct RadixState {
  uint32_t prefix, need, eq_total;
};

// Block-wide MSB-first radix select (11/11/10 bits) of the `need`-th largest non-zero key.
// each(f): calls f(key) for every element owned by this thread.  Returns T (0 if there are no keys);
// *need_eq = how many keys == T belong to the top-`need`, *eq_total = how many keys == T exist.
template <typename Each>
__device__ uint32_t block_radix_select(Each each, uint32_t need, uint32_t* hist /*2048*/, RadixState* rs,
                                       uint32_t* need_eq, uint32_t* eq_total) {
  const int tid = threadIdx.x;
  const int shifts[3] = {21, 10, 0};
  const uint32_t widths[3] = {11, 11, 10};
  if (tid == 0) { rs->prefix = 0; rs->need = need; rs->eq_total = 0; }
  uint32_t mask = 0;
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = shifts[pass];
    const uint32_t bins = 1u << widths[pass];
    for (int i = tid; i < 2048; i += 256) hist[i] = 0;
    __syncthreads();
    const uint32_t prefix = rs->prefix;
    each([&](uint32_t key) {
      if (key != 0 && (key & mask) == prefix) atomicAdd(&hist[(key >> shift) & (bins - 1)], 1u);
    });
    __syncthreads();
    if (tid == 0) {
      uint32_t nd = rs->need, above = 0;
      int b = (int)bins - 1;
      for (; b > 0; --b) {
        if (above + hist[b] >= nd) break;
        above += hist[b];
      }
      // b == 0 with too few keys: everything non-zero qualifies
      if (above + hist[b] < nd) { rs->need = hist[b]; } else { rs->need = nd - above; }
      rs->prefix = prefix | ((uint32_t)b << shift);
      rs->eq_total = hist[b];
    }
    mask |= (bins - 1) << shift;
    __syncthreads();
  }
  *need_eq = rs->need;
  *eq_total = rs->eq_total;
  return rs->prefix;
}

__global__ __launch_bounds__(256) void moment_topk_kernel(const float* __restrict__ st, const float* __restrict__ ed,
                                                          const float* __restrict__ w, float* __restrict__ out_score,
                                                          int32_t* __restrict__ out_flat, int kpairs, int lpad,
                                                          int l_ref, int min_l, int max_l, int n_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int q = blockIdx.x;
  const int R = kpairs * l_ref;
  float* s_st = reinterpret_cast<float*>(smem);            // [kpairs][l_ref]  st * w
  float* s_ed = s_st + R;                                  // [kpairs][l_ref]
  unsigned long long* s_list = reinterpret_cast<unsigned long long*>(s_ed + R + (R & 1));  // [MT_CAP]
  uint32_t* s_hist = reinterpret_cast<uint32_t*>(s_list + MT_CAP);                          // [2048]
  __shared__ RadixState rs;
  __shared__ uint32_t s_cnt;

  const float* gst = st + (int64_t)q * kpairs * lpad;
  const float* ged = ed + (int64_t)q * kpairs * lpad;
  for (int p = tid; p < R; p += 256) {
    const int r = p / l_ref, i = p - r * l_ref;
    const float wv = w ? w[(int64_t)q * kpairs + r] : 1.f;
    s_st[p] = gst[r * lpad + i] * wv;
    s_ed[p] = ged[r * lpad + i];
  }
  if (tid == 0) s_cnt = 0;
  __syncthreads();

  // ---- row maxima in registers ---------------------------------------------------------------------
  float rmax[MT_RPT];
#pragma unroll
  for (int t = 0; t < MT_RPT; ++t) {
    const int p = tid + t * 256;
    float m = 0.f;
    if (p < R) {
      const int r = p / l_ref, i = p - r * l_ref;
      const float a = s_st[p];
      const int jend = min(l_ref, i + max_l);
      for (int j = i + min_l; j < jend; ++j) m = fmaxf(m, a * s_ed[r * l_ref + j]);
    }
    rmax[t] = m;
  }
  uint32_t need_eq, eq_total;
  const uint32_t t_lb = block_radix_select(
      [&](auto f) {
#pragma unroll
        for (int t = 0; t < MT_RPT; ++t) f(__float_as_uint(rmax[t]));
      },
      (uint32_t)n_out, s_hist, &rs, &need_eq, &eq_total);
  const uint32_t lb = t_lb == 0 ? 1u : t_lb;  // fewer than n_out positive rows: keep every positive candidate

  // ---- expand surviving rows, collect candidates >= lb ----------------------------------------------
#pragma unroll
  for (int t = 0; t < MT_RPT; ++t) {
    if (__float_as_uint(rmax[t]) >= lb) {
      const int p = tid + t * 256;
      const int r = p / l_ref, i = p - r * l_ref;
      const float a = s_st[p];
      const int jend = min(l_ref, i + max_l);
      for (int j = i + min_l; j < jend; ++j) {
        const uint32_t key = __float_as_uint(a * s_ed[r * l_ref + j]);
        if (key >= lb) {
          const uint32_t slot = atomicAdd(&s_cnt, 1u);
          if (slot < (uint32_t)MT_CAP)
            s_list[slot] = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - (uint32_t)(p * l_ref + j));
        }
      }
    }
  }
  __syncthreads();
  uint32_t cnt = s_cnt;
  __syncthreads();

  if (cnt > (uint32_t)MT_CAP) {
    // ---- fallback: exact threshold over every candidate (flat score distributions) -----------------
    auto each_cand = [&](auto f) {
      for (int p = tid; p < R; p += 256) {
        const int r = p / l_ref, i = p - r * l_ref;
        const float a = s_st[p];
        const int jend = min(l_ref, i + max_l);
        for (int j = i + min_l; j < jend; ++j) f(__float_as_uint(a * s_ed[r * l_ref + j]), p * l_ref + j);
      }
    };
    const uint32_t T = block_radix_select([&](auto f) { each_cand([&](uint32_t key, int) { f(key); }); },
                                          (uint32_t)n_out, s_hist, &rs, &need_eq, &eq_total);
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    each_cand([&](uint32_t key, int flat) {
      if (key > T || (key == T && T != 0)) {
        const uint32_t slot = atomicAdd(&s_cnt, 1u);  // > T first-come; ties at T beyond capacity are dropped
        if (slot < (uint32_t)MT_CAP)
          s_list[slot] = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - (uint32_t)flat);
      }
    });
    __syncthreads();
    cnt = min(s_cnt, (uint32_t)MT_CAP);
    __syncthreads();
  }

  // ---- bitonic sort (descending) of the list, padded with zeros to a power of two ------------------
  int npow = 256;
  while ((uint32_t)npow < cnt) npow <<= 1;
  for (int i = (int)cnt + tid; i < npow; i += 256) s_list[i] = 0ull;
  __syncthreads();
  for (int size = 2; size <= npow; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (npow >> 1); i += 256) {
        const int lo = ((i / stride) * stride << 1) + (i % stride);
        const int hi = lo + stride;
        const unsigned long long a = s_list[lo], b = s_list[hi];
        const bool desc = (lo & size) == 0;
        if (desc ? (a < b) : (a > b)) { s_list[lo] = b; s_list[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < n_out; i += 256) {
    float sc = 0.f;
    int32_t flat = -1;
    if (i < npow) {
      const unsigned long long c = s_list[i];
      if (c != 0ull) {
        sc = __uint_as_float((uint32_t)(c >> 32));
        flat = (int32_t)(0xffffffffu - (uint32_t)(c & 0xffffffffull));
      }
    }
    out_score[(int64_t)q * n_out + i] = sc;
    out_flat[(int64_t)q * n_out + i] = flat;
  }
}

extern "C" int xml_moment_topk(const float* st, const float* ed, const float* w, float* out_score, int32_t* out_flat,
                               int nq, int kpairs, int lpad, int l_ref, int min_l, int max_l, int n_out,
                               xml_stream_t stream) {
  XML_ENTER();
  if (!st || !ed || !out_score || !out_flat || nq <= 0 || kpairs <= 0 || lpad <= 0 || l_ref <= 0 || n_out <= 0)
    return XML_ERR_BAD_ARG;
  if (l_ref > lpad || min_l < 0 || max_l <= min_l) return XML_ERR_BAD_ARG;
  if (n_out > 1024 || lpad > 128 || (int64_t)kpairs * l_ref > 256 * MT_RPT) return XML_ERR_UNSUPPORTED;
  const size_t R = (size_t)kpairs * l_ref;
  const size_t lds = (2 * R + (R & 1)) * 4 + (size_t)MT_CAP * 8 + 2048 * 4;
  if (lds > 160 * 1024 - 64) return XML_ERR_UNSUPPORTED;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)moment_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return XML_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(moment_topk_kernel, dim3(nq), dim3(256), lds, (hipStream_t)stream, st, ed, w, out_score, out_flat,
                     kpairs, lpad, l_ref, min_l, max_l, n_out);
  XML_CHECK_LAUNCH();
  return XML_OK;
}
