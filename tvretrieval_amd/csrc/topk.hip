// K8: per-row top-k  (reference: torch.exp(alpha * s) ; torch.topk(.., k, dim=1), xml/inference.py:317,347-348)
//
// One workgroup per row.  exp(alpha * s) is monotone, so selection runs on the raw f32 score bits:
//   1. 4 x 8-bit MSB-first radix-select over the row (LDS histogram; the row is read ONCE into registers when it has
//      <= 256 * VPT elements, else re-read from L2) -> exact key T of the k-th largest element and how many elements
//      equal to T are still needed; the bin holding the k-th element is found by a parallel suffix scan;
//   2. gather every element > T plus the needed ones == T (lowest column first when ties exceed the need);
//   3. bitonic sort of the <= 256 survivors by (score desc, payload asc) in LDS; emit exp(alpha*s) + payload.
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

// VPT > 0: keys live in registers (thread t owns columns t, t + 256, ...; n <= 256 * VPT); VPT == 0: any n, re-reads.
template <int VPT>
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ scores, int64_t ld,
                                                        const int32_t* __restrict__ idx_in,
                                                        float* __restrict__ out_val, int32_t* __restrict__ out_idx,
                                                        int n, int k, float alpha) {
  __shared__ uint32_t hist[256];
  __shared__ unsigned long long comp[256];
  __shared__ uint32_t s_prefix, s_need, s_cnt, s_eq_total, s_eq_taken;
  __shared__ uint32_t s_wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = scores + (int64_t)blockIdx.x * ld;
  const int32_t* pay = idx_in ? idx_in + (int64_t)blockIdx.x * ld : nullptr;
  constexpr int NK = VPT > 0 ? VPT : 1;
  uint32_t keys[NK];
  if (VPT > 0) {
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const int i = tid + j * 256;
      keys[j] = i < n ? ord_key(row[i]) : 0u;     // 0 sorts below every real key (ord_key(x) >= 1 for finite / inf x)
    }
  }
  // walk the row: f(key, column)
  auto for_each = [&](auto&& f) {
    if constexpr (VPT > 0) {
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        const int i = tid + j * 256;
        if (i < n) f(keys[j], i);
      }
    } else {
      for (int i = tid; i < n; i += 256) f(ord_key(row[i]), i);
    }
  };

  if (tid == 0) { s_prefix = 0; s_need = (uint32_t)k; }
  uint32_t mask = 0;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix;
    for_each([&](uint32_t key, int) {
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xff], 1u);
    });
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
      if (inc >= need && inc - h < need) {        // exactly one thread (the row holds >= need candidates)
        s_need = need - (inc - h);
        s_prefix = prefix | ((uint32_t)(255 - tid) << shift);
        s_eq_total = h;
      }
    }
    mask |= 0xffu << shift;
    __syncthreads();
  }
  const uint32_t T = s_prefix;
  const uint32_t need_eq = s_need;          // elements == T still to take (>= 1)
  const uint32_t eq_total = s_eq_total;     // elements == T in the row
  if (tid == 0) { s_cnt = 0; s_eq_taken = 0; }
  comp[tid] = 0ull;
  __syncthreads();

  auto emit = [&](uint32_t key, int i) {
    const uint32_t slot = atomicAdd(&s_cnt, 1u);
    const uint32_t p = pay ? (uint32_t)pay[i] : (uint32_t)i;
    if (slot < 256) comp[slot] = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - p);
  };
  if (eq_total == need_eq) {
    for_each([&](uint32_t key, int i) {
      if (key >= T) emit(key, i);
    });
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
  __syncthreads();

  // bitonic sort, descending on the composite (score desc, payload asc); empty slots (0) sink to the end
  for (int size = 2; size <= 256; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int partner = tid ^ stride;
      if (partner > tid) {
        const unsigned long long a = comp[tid], b = comp[partner];
        const bool desc = (tid & size) == 0;
        if (desc ? (a < b) : (a > b)) { comp[tid] = b; comp[partner] = a; }
      }
      __syncthreads();
    }
  }
  if (tid < k) {
    const unsigned long long c = comp[tid];
    const float s = key_to_float((uint32_t)(c >> 32));
    out_val[(int64_t)blockIdx.x * k + tid] = (alpha != 0.f) ? expf(alpha * s) : s;
    out_idx[(int64_t)blockIdx.x * k + tid] = (int32_t)(0xffffffffu - (uint32_t)(c & 0xffffffffull));
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
