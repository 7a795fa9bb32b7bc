// K10: the index tail of compute_query2ctx_info as a device epilogue of K9 / K8.
//   reference: np.unravel_index over (max_n_videos, max_ctx_l, max_ctx_l), sorted_q2c_indices[i, local],
//              st = st_idx.astype(f32) * clip_length, ed = ed_idx.astype(f32) * clip_length + clip_length,
//              video2idx[video_metas[meta]["vid_name"]]              xml/inference.py:415-439 (VCMR), :402-413 (VR)
//              _sorted_triples[:, 1] += 1; [:, :2] * clip_length      xml/inference.py:229-233        (SVMR, float64)
// One record per list entry -- xml_moment {vid, st, ed, score}, 16 bytes -- so a batch leaves the device in ONE copy and the
// host never loops over queries.  HBM-bound trivia: 10 000 x 200 records = 32 MB written, 16 MB read.
#include "common.h"

namespace {

// one workgroup per query row; n <= 1024 entries walked 256 at a time
__global__ __launch_bounds__(256) void moments_decode_kernel(
    const int32_t* __restrict__ flat, const float* __restrict__ score, const int32_t* __restrict__ top_idx,
    const int32_t* __restrict__ row_vid, const int32_t* __restrict__ meta2vid, int n, int64_t ld_in, int k, int l_ref,
    float clip, int seconds, xml_moment* __restrict__ out, int64_t ld_out, int32_t* __restrict__ out_count) {
  const int q = blockIdx.x, tid = threadIdx.x;
  const int ll = l_ref * l_ref;
  int valid = 0;
  for (int i = tid; i < n; i += 256) {
    xml_moment m;
    m.vid = -1; m.st = 0.f; m.ed = 0.f;
    m.score = score[(int64_t)q * ld_in + i];
    if (flat) {
      const int32_t f = flat[(int64_t)q * ld_in + i];
      if (f >= 0) {
        const int r = f / ll, rem = f - r * ll;
        const int si = rem / l_ref, ei = rem - si * l_ref;
        int meta = top_idx ? top_idx[(int64_t)q * k + r] : (row_vid ? row_vid[q] : r);
        m.vid = (meta2vid && meta >= 0) ? meta2vid[meta] : meta;
        if (seconds) {     // numpy float32 arithmetic: one rounding per operation.  hipcc contracts a * b + c into an fma
#pragma clang fp contract(off)   // by default (-ffp-contract=fast, and HIP's __fmul_rn / __fadd_rn are plain operators)
          const float prod = (float)ei * clip;
          m.st = (float)si * clip;
          m.ed = prod + clip;
        } else {           // clip units (st_idx, ed_idx + 1), exact; the caller scales in float64 like the reference's SVMR tail
          m.st = (float)si;
          m.ed = (float)(ei + 1);
        }
        ++valid;
      } else {
        m.score = 0.f;
      }
    } else {               // video-retrieval list: [video_idx, 0, 0, score]
      const int meta = top_idx[(int64_t)q * ld_in + i];
      m.vid = (meta2vid && meta >= 0) ? meta2vid[meta] : meta;
      valid += meta >= 0;
    }
    *reinterpret_cast<uint4*>(&out[(int64_t)q * ld_out + i]) = *reinterpret_cast<const uint4*>(&m);
  }
  if (out_count) {
    __shared__ int s_cnt[4];
    for (int o = 32; o > 0; o >>= 1) valid += __shfl_down(valid, o, 64);
    if ((tid & 63) == 0) s_cnt[tid >> 6] = valid;
    __syncthreads();
    if (tid == 0) out_count[q] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  }
}

}  // namespace

extern "C" int xml_moments_decode(const int32_t* flat, const float* score, const int32_t* top_idx, const int32_t* row_vid,
                                  const int32_t* meta2vid, int nq, int n, int64_t ld_in, int k, int l_ref, float clip_length,
                                  int seconds, xml_moment* out, int64_t ld_out, int32_t* out_count, xml_stream_t stream) {
  XML_ENTER();
  static_assert(sizeof(xml_moment) == 16, "xml_moment is a 16-byte record");
  if (!score || !out || nq <= 0 || n <= 0 || ld_in < n || ld_out < n) return XML_ERR_BAD_ARG;
  if (flat && (l_ref <= 0 || (top_idx && k <= 0))) return XML_ERR_BAD_ARG;
  if (!flat && !top_idx) return XML_ERR_BAD_ARG;
  if (((uintptr_t)out & 15) != 0) return XML_ERR_BAD_ARG;
  if (flat && (int64_t)l_ref * l_ref * (top_idx ? k : 1) > INT32_MAX) return XML_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(moments_decode_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, flat, score, top_idx, row_vid,
                     meta2vid, n, ld_in, k, l_ref, clip_length, seconds, out, ld_out, out_count);
  XML_CHECK_LAUNCH();
  return XML_OK;
}
