// K8: per-row top-k  (reference: torch.exp(alpha * s) ; torch.topk(.., k, dim=1), xml/inference.py:317,347-348)
//
// One workgroup per row.  exp(alpha * s) is monotone, so selection runs on the raw f32 score bits:
//   1. 4 x 8-bit MSB-first radix-select over the row (LDS histogram; the row is read ONCE into registers when it has
//      <= 256 * VPT elements, else re-read from L2) -> exact key T of the k-th largest element and how many elements
//      equal to T are still needed; the bin holding the k-th element is found by a parallel suffix scan;
//   2. gather every element > T plus the needed ones == T (when ties exceed the need: lowest payload first, i.e. lowest
//      column without payloads);
//   3. the <= 256 survivors in (score desc, payload asc) order, each ranked against the others in LDS; emit exp(alpha*s) +
//      payload.
// HBM/L2-bound integer work: no reshaping into a GEMM.
#include "common.h"

__device__ __forceinline__ uint32_t ord_key(float f) {
  uint32_t u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0u;  // -0.0 ties with +0.0, as in torch.topk
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending uint order == ascending float order
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// VPT > 0: keys live in registers (thread t owns columns t, t + 256, ...; n <= 256 * VPT).
// VPT == 0: any n.  Long rows (n >= 4096) are pre-filtered so that the row is read ONCE instead of five times:
//   a. the r-th largest key T0 of a 2048-column sample (the first columns; 8 keys per thread in registers), with r chosen
//      so that about 3 k elements of the whole row reach T0;
//   b. one pass over the row collects every element >= T0 into an LDS candidate list (wave-aggregated append);
//   c. if the list holds between k and 1024 entries (and the threshold ties need no column-ordered tie-break), the exact
//      selection runs on the list; otherwise -- adversarially ordered rows, massive ties -- the kernel falls back to
//      re-reading the row for every radix pass.  Either way the result is the exact top-k with the same tie rule.
template <int VPT>
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ scores, int64_t ld,
                                                        const int32_t* __restrict__ idx_in,
                                                        float* __restrict__ out_val, int32_t* __restrict__ out_idx,
                                                        int n, int k, float alpha) {
  constexpr int CAND_CAP = VPT > 0 ? 1 : 1024;
  constexpr int SAMPLE = 2048;
  __shared__ uint32_t hist[256];
  __shared__ unsigned long long comp[256];
  __shared__ unsigned long long cand[CAND_CAP];      // (key << 32) | column
  __shared__ uint32_t s_prefix, s_need, s_cnt, s_eq_total, s_eq_taken;
  __shared__ uint32_t s_wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = scores + (int64_t)blockIdx.x * ld;
  const int32_t* pay = idx_in ? idx_in + (int64_t)blockIdx.x * ld : nullptr;
  constexpr int NK = VPT > 0 ? VPT : SAMPLE / 256;
  const int n_reg = VPT > 0 ? n : min(n, SAMPLE);
  uint32_t keys[NK];
  {   // all loads first, then the conversions: written as load -> convert per element, hipcc waits for each load in turn
    float raw[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const int i = tid + j * 256;
      raw[j] = i < n_reg ? row[i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NK; ++j)
      keys[j] = tid + j * 256 < n_reg ? ord_key(raw[j]) : 0u;   // 0 sorts below every real key (ord_key(x) >= 1)
  }
  enum { SRC_REG = 0, SRC_ROW = 1, SRC_CAND = 2 };
  int src = VPT > 0 ? SRC_REG : SRC_ROW;
  int n_cand = 0;
  // walk the current source: f(key, column, valid); every lane of a wave runs the same number of iterations
  auto walk = [&](auto&& f) {
    if (src == SRC_REG) {
#pragma unroll
      for (int j = 0; j < NK; ++j) f(keys[j], tid + j * 256, tid + j * 256 < n_reg);
    } else if (src == SRC_ROW) {
      for (int base = 0; base < n; base += 256 * 8) {        // eight loads in flight per thread (see above)
        float raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = base + u * 256 + tid;
          raw[u] = i < n ? row[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = base + u * 256 + tid;
          if (base + u * 256 < n) f(i < n ? ord_key(raw[u]) : 0u, i, i < n);      // (wave-uniform condition)
        }
      }
    } else {
      for (int base = 0; base < n_cand; base += 256) {
        const int i = base + tid;
        const unsigned long long e = i < n_cand ? cand[i] : 0ull;
        f((uint32_t)(e >> 32), (int)(uint32_t)(e & 0xffffffffull), i < n_cand);
      }
    }
  };
  // histogram pass.  Scores of one row share their top bits (cosines: a handful of exponents), so in the FIRST pass
  // nearly all 64 lanes of a wave hit one or two bins and a plain atomicAdd serialises up to 64-fold; there the adds are
  // wave-aggregated (up to 4 leader rounds, then plain adds for what is left).  Later passes see spread-out bins.
  auto hist_pass = [&](uint32_t mask, uint32_t prefix, int shift) {
    walk([&](uint32_t key, int, bool valid) {
      const bool act = valid && (key & mask) == prefix;
      const uint32_t bin = (key >> shift) & 0xff;
      if (shift == 24) {
        unsigned long long todo = __ballot(act);
#pragma unroll 1
        for (int r = 0; r < 4 && todo; ++r) {
          const int leader = __ffsll((long long)todo) - 1;
          const uint32_t lb = __shfl(bin, leader, 64);
          const unsigned long long same = __ballot(act && bin == lb) & todo;
          if (lane == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(same));
          todo &= ~same;
        }
        if ((todo >> lane) & 1ull) atomicAdd(&hist[bin], 1u);
      } else if (act) {
        atomicAdd(&hist[bin], 1u);
      }
    });
  };
  // 4 x 8-bit MSB-first radix select of the kth largest key of the current source (which holds >= kth elements):
  // s_prefix = its key T, s_need = elements == T still to take, s_eq_total = elements == T in the source
  auto select = [&](uint32_t kth) {
    if (tid == 0) { s_prefix = 0; s_need = kth; }
    uint32_t mask = 0;
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      hist[tid] = 0;
      __syncthreads();
      const uint32_t prefix = s_prefix;
      hist_pass(mask, prefix, shift);
      __syncthreads();
      {   // bin b with  sum_{x > b} hist[x] < need <= sum_{x >= b} hist[x]:  inclusive scan from the top bin down
        const uint32_t need = s_need;
        const uint32_t h = hist[255 - tid];
        uint32_t inc = h;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t v = __shfl_up(inc, o, 64);
          if (lane >= o) inc += v;
        }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        for (int w = 0; w < wave; ++w) inc += s_wsum[w];
        if (inc >= need && inc - h < need) {        // exactly one thread (the source holds >= need candidates)
          s_need = need - (inc - h);
          s_prefix = prefix | ((uint32_t)(255 - tid) << shift);
          s_eq_total = h;
        }
      }
      mask |= 0xffu << shift;
      __syncthreads();
    }
  };
  auto emit = [&](uint32_t key, int i) {
    const uint32_t slot = atomicAdd(&s_cnt, 1u);
    const uint32_t p = pay ? (uint32_t)pay[i] : (uint32_t)i;
    if (slot < 256) comp[slot] = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - p);
  };

  bool done = false;
  if constexpr (VPT == 0) {
    const uint32_t r = (uint32_t)(((int64_t)3 * k * SAMPLE + n - 1) / n) + 2;
    if (n >= 2 * SAMPLE && r <= SAMPLE / 8) {
      src = SRC_REG;
      select(r);                                           // a. threshold from the sample
      const uint32_t t0 = s_prefix;
      if (tid == 0) s_cnt = 0;
      __syncthreads();
      src = SRC_ROW;
      walk([&](uint32_t key, int i, bool valid) {          // b. one pass: everything >= t0
        const bool take = valid && key >= t0;
        const unsigned long long bal = __ballot(take);
        if (bal) {
          uint32_t base = 0;
          const int leader = __ffsll((long long)bal) - 1;
          if (lane == leader) base = atomicAdd(&s_cnt, (uint32_t)__popcll(bal));
          base = __shfl(base, leader, 64);
          const uint32_t slot = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
          if (take && slot < (uint32_t)CAND_CAP) cand[slot] = ((unsigned long long)key << 32) | (unsigned long long)(uint32_t)i;
        }
      });
      __syncthreads();
      const uint32_t cnt = s_cnt;
      __syncthreads();
      if (cnt >= (uint32_t)k && cnt <= (uint32_t)CAND_CAP) {   // c. exact selection on the candidates
        n_cand = (int)cnt;
        src = SRC_CAND;
        select((uint32_t)k);
        if (s_eq_total == s_need) {
          const uint32_t T = s_prefix;
          if (tid == 0) s_cnt = 0;
          comp[tid] = 0ull;
          __syncthreads();
          walk([&](uint32_t key, int i, bool valid) {
            if (valid && key >= T) emit(key, i);
          });
          done = true;
        }
      }
      src = SRC_ROW;
    }
  }
  if (!done) {
    __syncthreads();                          // (the pre-filter's readers of s_need / s_eq_total are done)
    select((uint32_t)k);
    const uint32_t T = s_prefix;
    const uint32_t need_eq = s_need;          // elements == T still to take (>= 1)
    const uint32_t eq_total = s_eq_total;     // elements == T in the row
    if (tid == 0) { s_cnt = 0; s_eq_taken = 0; }
    comp[tid] = 0ull;
    __syncthreads();
    if (eq_total == need_eq) {
      walk([&](uint32_t key, int i, bool valid) {
        if (valid && key >= T) emit(key, i);
      });
    } else if (pay) {
      // ties at the threshold exceed the need AND the caller ordered by payload: the need_eq SMALLEST PAYLOADS of the tied
      // elements (then lowest column) -- the documented order (score desc, payload asc) also at the list's last position.
      // (Taking the lowest columns here made a merged sharded list depend on which rank a tied candidate came from: one of
      // 10 000 queries at the TVR shape ended on a different one of two equal-score moments than the single-GPU pass.)
      // Rare path, short rows (merges: n = world x c): one block-wide minimum per element taken.
      for (int j = tid; j < n; j += 256) {
        const uint32_t kj = ord_key(row[j]);
        if (kj > T) emit(kj, j);
      }
      __shared__ unsigned long long s_min;
      unsigned long long last = 0ull;
      for (uint32_t t = 0; t < need_eq; ++t) {
        if (tid == 0) s_min = ~0ull;
        __syncthreads();
        unsigned long long best = ~0ull;
        for (int j = tid; j < n; j += 256) {
          if (ord_key(row[j]) == T) {
            const unsigned long long c = ((unsigned long long)(uint32_t)pay[j] << 32) | (unsigned long long)(uint32_t)j;
            if ((t == 0 || c > last) && c < best) best = c;
          }
        }
        if (best != ~0ull) atomicMin(&s_min, best);
        __syncthreads();
        last = s_min;
        if (tid == 0 && last != ~0ull) emit(T, (int)(uint32_t)(last & 0xffffffffull));
        __syncthreads();
      }
    } else {
      // ties at the threshold exceed the need: take the lowest columns, in order (rare path)
      for (int base = 0; base < n; base += 256) {
        const int i = base + tid;
        uint32_t key = 0;
        bool eq = false;
        if (i < n) {
          key = ord_key(row[i]);      // rare path: plain re-read (dynamic register indexing would go to scratch)
          if (key > T) emit(key, i);
          eq = key == T;
        }
        const unsigned long long bal = __ballot(eq);
        __shared__ uint32_t wcount[4];
        if ((tid & 63) == 0) wcount[tid >> 6] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = s_eq_taken;
        for (int w = 0; w < (tid >> 6); ++w) before += wcount[w];
        before += (uint32_t)__popcll(bal & ((1ull << (tid & 63)) - 1ull));
        if (eq && before < need_eq) emit(key, i);
        __syncthreads();
        if (tid == 0) s_eq_taken += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
        if (s_eq_taken >= need_eq) {
          // remaining chunks can only contribute elements > T
          for (int j = base + 256 + tid; j < n; j += 256) {
            const uint32_t kj = ord_key(row[j]);
            if (kj > T) emit(kj, j);
          }
          break;
        }
      }
    }
  }
  __syncthreads();

  // order on the composite (score desc, payload asc).  The composites are distinct (the payload is part of them), so an
  // entry's position is the number of entries above it: every thread ranks ITS entry against the <= 256 survivors (all lanes
  // read the same LDS word: a broadcast) and stores it at its rank -- one pass without a barrier instead of the 36
  // barrier-separated stages of a 256-wide bitonic network.  Slots beyond the survivors (k > row length) stay empty rows.
  const uint32_t n_comp = min(s_cnt, 256u);
  const unsigned long long mine = comp[tid];
  uint32_t rank = 0;
#pragma unroll 8
  for (uint32_t j = 0; j < n_comp; ++j) rank += comp[j] > mine ? 1u : 0u;
  const uint32_t pos = (uint32_t)tid < n_comp ? rank : (uint32_t)tid;
  if (pos < (uint32_t)k && ((uint32_t)tid < n_comp || tid < k)) {
    const float s = key_to_float((uint32_t)(mine >> 32));
    out_val[(int64_t)blockIdx.x * k + pos] = (alpha != 0.f) ? expf(alpha * s) : s;
    out_idx[(int64_t)blockIdx.x * k + pos] = (int32_t)(0xffffffffu - (uint32_t)(mine & 0xffffffffull));
  }
}

extern "C" size_t xml_topk_rows_workspace_bytes(int rows, int n, int k) {
  (void)rows; (void)n; (void)k;
  return 0;
}

extern "C" int xml_topk_rows(const float* scores, int64_t ld, const int32_t* idx_in, float* out_val,
                             int32_t* out_idx, int rows, int n, int k, float alpha, void* ws, size_t ws_bytes,
                             xml_stream_t stream) {
  XML_ENTER();
  (void)ws; (void)ws_bytes;
  if (!scores || !out_val || !out_idx || rows <= 0 || n <= 0 || k <= 0 || ld < n) return XML_ERR_BAD_ARG;
  if (k > 256 || k > n) return XML_ERR_UNSUPPORTED;
  // register-resident keys pay for short rows (shard-local / merge passes); at n = 21 793 (96 keys per thread) the
  // L2 re-read variant measured faster (1.45 vs 1.88 ms per 10 000 rows)
  auto kern = n <= 256 * 12 ? topk_rows_kernel<12> : topk_rows_kernel<0>;
  hipLaunchKernelGGL(kern, dim3(rows), dim3(256), 0, (hipStream_t)stream, scores, ld, idx_in, out_val,
                     out_idx, n, k, alpha);
  XML_CHECK_LAUNCH();
  return XML_OK;
}
