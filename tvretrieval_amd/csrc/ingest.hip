// Feature ingest on the device ("next" row 8f-3): the dataset-side half of the reference's context collate, applied to raw
// rows as they arrive from the host.
//   reference: StartEndEvalDataset._get_item_context  xml/start_end_dataset.py:297-321  (truncate to max_ctx_len, then
//                                                      l2_normalize_np_array: x / (||x||_2 + 1e-5), utils/basic_utils.py:82-84)
//              start_end_collate / pad_sequences_1d   xml/start_end_dataset.py:346-359, utils/tensor_utils.py:5-53
//                                                      (zero padding to the batch maximum + float mask)
// The host hands over the batch's rows back to back in the STORE's dtype (f16 on disk: half the bytes over PCIe, no
// host-side conversion, no per-video Python) plus a prefix array; one launch writes the padded (n, lmax, d) tensor the
// encoder reads -- converted, normalised, zero beyond each video's length -- and the (n, lmax) mask.  HBM-bound:
// reads rows * d * 2 B, writes n * lmax * d * 4 B (f32) or * 2 B (bf16).
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

template <typename S> __device__ __forceinline__ float ing_ld(const S* p);
template <> __device__ __forceinline__ float ing_ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ing_ld<__half>(const __half* p) { return __half2float(*p); }

// one wave per destination row (video i, clip l)
template <typename S, typename D>
__global__ __launch_bounds__(256) void ingest_rows_kernel(const S* __restrict__ src, const int64_t* __restrict__ row_start,
                                                          D* __restrict__ dst, float* __restrict__ mask, int n, int lmax,
                                                          int d, int max_len, float eps, int normalize) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (int64_t)n * lmax) return;
  const int i = (int)(r / lmax), l = (int)(r - (int64_t)i * lmax);
  const int64_t first = row_start[i];
  const int len = (int)min((int64_t)max_len, row_start[i + 1] - first);
  D* out = dst + r * d;
  if (lane == 0 && mask) mask[r] = l < len ? 1.f : 0.f;
  if (l >= len) {
    for (int c = lane; c < d; c += 64) DT<D>::st(out + c, 0.f);
    return;
  }
  const S* px = src + (first + l) * d;
  float s = 0.f;
  if (normalize) {
    for (int c = lane; c < d; c += 64) { const float v = ing_ld<S>(px + c); s += v * v; }
    s = sqrtf(wave_sum(s)) + eps;          // l2_normalize_np_array: x / (||x|| + eps)
  }
  for (int c = lane; c < d; c += 64) {
    const float v = ing_ld<S>(px + c);
    DT<D>::st(out + c, normalize ? v / s : v);
  }
}

}  // namespace

extern "C" int xml_ingest_rows(const void* src, int src_dt, const int64_t* row_start, void* dst, int dst_dt, float* mask, int n,
                               int lmax, int d, int max_len, float eps, int normalize, xml_stream_t stream) {
  XML_ENTER();
  if (!src || !row_start || !dst || n <= 0 || lmax <= 0 || d <= 0 || max_len <= 0) return XML_ERR_BAD_ARG;
  if ((src_dt != XML_F32 && src_dt != XML_F16) || (dst_dt != XML_F32 && dst_dt != XML_BF16)) return XML_ERR_BAD_ARG;
  const int64_t rows = (int64_t)n * lmax;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define XML_INGEST(S, D) \
  hipLaunchKernelGGL((ingest_rows_kernel<S, D>), grid, block, 0, st, (const S*)src, row_start, (D*)dst, mask, n, lmax, d, max_len, eps, normalize)
  if (src_dt == XML_F32 && dst_dt == XML_F32) XML_INGEST(float, float);
  else if (src_dt == XML_F32) XML_INGEST(float, bf16_t);
  else if (dst_dt == XML_F32) XML_INGEST(__half, float);
  else XML_INGEST(__half, bf16_t);
#undef XML_INGEST
  XML_CHECK_LAUNCH();
  return XML_OK;
}
