// Corpus index build, last step: L2-normalise the clip rows of feat1 (F.normalize, xml/model_xml.py:447-448) and write them
// straight into K6's slice-major tile image ([tile of 256 rows][64-byte K slice][row][64 B], see q2c_persist.hip) -- one
// pass over the index instead of l2norm_rows (read + write) followed by tile_rows (read + write).  Optionally through a
// row map (the length-bucketed image of ragged corpora: destination row i = source row row_map[i], zeros when < 0).
// The arithmetic is l2norm.h's: bitwise the values the two separate kernels produce.
#include "l2norm.h"

namespace {

template <typename T, bool GATHER>
__global__ __launch_bounds__(256) void tile_rows_l2norm_kernel(const T* __restrict__ src, const int32_t* __restrict__ row_map,
                                                               T* __restrict__ dst, int64_t rows_src, int64_t rows_dst,
                                                               int d) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int l32 = threadIdx.x & 31;
  const int64_t drow = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);      // row of the tiled image (rows_dst % 256 == 0)
  int64_t srow = -1;
  if (drow < rows_dst) srow = GATHER ? (int64_t)row_map[drow] : (drow < rows_src ? drow : -1);
  uint4 v[L2N_MAXJ];
  const float nrm = l2n_load_row<T>(srow >= 0 ? src + srow * d : nullptr, d, l32, v);
  if (drow >= rows_dst) return;
  const int chunks = d / VEC;
  const int slices = chunks >> 2;                                          // 64-byte slices per row
  char* tile = reinterpret_cast<char*>(dst) + (drow >> 8) * (int64_t)slices * (256 * 64) + (drow & 255) * 64;
#pragma unroll
  for (int j = 0; j < L2N_MAXJ; ++j) {
    const int c = l32 + 32 * j;
    if (c < chunks)       // chunk c = piece c & 3 of slice c >> 2; a missing row stays zero (0 / 1e-12 = 0)
      *reinterpret_cast<uint4*>(tile + (int64_t)(c >> 2) * (256 * 64) + (c & 3) * 16) = l2n_scale_chunk<T>(v[j], nrm);
  }
}

}  // namespace

extern "C" int xml_q2c_tile_rows_l2norm_ok(int hidden, int dt) {
  if (dt != XML_F32 && dt != XML_BF16) return 0;
  const int vec = dt == XML_F32 ? 4 : 8;
  return hidden > 0 && hidden % (4 * vec) == 0 && hidden / vec <= 32 * L2N_MAXJ;      // whole 64-byte slices
}

extern "C" int xml_q2c_tile_rows_l2norm(const void* src, const int32_t* row_map, void* dst, int64_t rows_src,
                                        int64_t rows_dst, int hidden, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!src || !dst || rows_src <= 0 || rows_dst <= 0 || (rows_dst & 255)) return XML_ERR_BAD_ARG;
  if (!xml_q2c_tile_rows_l2norm_ok(hidden, dt)) return XML_ERR_UNSUPPORTED;
  if (!row_map && rows_dst < rows_src) return XML_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((rows_dst + 7) / 8));
  if (dt == XML_F32) {
    if (row_map) hipLaunchKernelGGL((tile_rows_l2norm_kernel<float, true>), grid, dim3(256), 0, st, (const float*)src, row_map, (float*)dst, rows_src, rows_dst, hidden);
    else hipLaunchKernelGGL((tile_rows_l2norm_kernel<float, false>), grid, dim3(256), 0, st, (const float*)src, row_map, (float*)dst, rows_src, rows_dst, hidden);
  } else {
    if (row_map) hipLaunchKernelGGL((tile_rows_l2norm_kernel<bf16_t, true>), grid, dim3(256), 0, st, (const bf16_t*)src, row_map, (bf16_t*)dst, rows_src, rows_dst, hidden);
    else hipLaunchKernelGGL((tile_rows_l2norm_kernel<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)src, row_map, (bf16_t*)dst, rows_src, rows_dst, hidden);
  }
  XML_CHECK_LAUNCH();
  return XML_OK;
}
