// The span predictors as modules of their own: nn.Conv1d(1, 1, k, padding = k // 2, bias = False) on rows of similarities.
//   reference: self.merged_st_predictor(similarity) / merged_ed_predictor  xml/model_xml.py:114-117,476-477
//              (called directly by baselines/profiling/profile_main.py:204-205)
// The retrieval pass never goes through here -- K7 (convse.hip) applies the taps to the similarities while they are in
// registers; this entry serves callers that hold a similarity tensor already.  HBM-bound: 8 bytes per element.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void conv1d_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, int64_t rows, int l, int ksize) {
  __shared__ float taps[16];
  if (threadIdx.x < ksize) taps[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const int half = ksize >> 1;
  const int64_t total = rows * l;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / l;
    const int i = (int)(e - r * l);
    const float* row = x + r * l;
    float acc = 0.f;
    for (int t = 0; t < ksize; ++t) {          // cross-correlation, zero padding: y[i] = sum_t w[t] x[i + t - k / 2]
      const int j = i + t - half;
      if (j >= 0 && j < l) acc += taps[t] * row[j];
    }
    y[e] = acc;
  }
}

}  // namespace

extern "C" int xml_conv1d_rows(const float* x, const float* w, float* y, int64_t rows, int l, int ksize, xml_stream_t stream) {
  XML_ENTER();
  if (!x || !w || !y || rows <= 0 || l <= 0) return XML_ERR_BAD_ARG;
  if (ksize <= 0 || ksize > 15 || (ksize & 1) == 0) return XML_ERR_UNSUPPORTED;
  const int64_t blocks = (rows * l + 255) / 256;
  hipLaunchKernelGGL(conv1d_rows_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, x, w,
                     y, rows, l, ksize);
  XML_CHECK_LAUNCH();
  return XML_OK;
}
