// Weight-gradient GEMM of the training step, straight from the row-major operands:
//   dW (N, K) f32  =  dY^T X  =  sum_r dY[r][n] X[r][k]        dY (rows, N), X (rows, K) bf16, both as the layers hold them
//   reference: the autograd of nn.Linear inside XML.forward (xml/model_components.py:156-163; loss.backward(), xml/train.py:81)
// Both operands are contracted over their ROW index, i.e. both are needed transposed.  The unfused path transposed dY and X
// explicitly (two launches writing rows x (N + K) elements) and ran a split-K NT GEMM on the copies; here 32-row slabs of
// dY and X go to LDS as they lie (row-major, 16-byte loads / stores) and the MFMA fragments -- 8 consecutive rows of one
// column per lane -- come out of ds_read_b64_tr_b16 (attention.hip has the lane semantics).
// One workgroup = 4 waves = a 128 (n) x 128 (k) output tile over one range of rows; the row ranges (blockIdx.z) are combined
// with f32 atomics into the pre-zeroed output, as the split-K kernel did.  Register double buffer: the loads of slab s + 1
// are in flight under the MFMAs of slab s.
#include "common.h"

namespace {

constexpr int TN_STRIDE = 128 * 2 + 16;      // LDS row of a slab: 128 columns bf16 + 16 bytes pad

typedef short tn_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 tn_read_tr16(const char* p) {
  const tn_v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_v4s*)p);
  return __builtin_bit_cast(uint2, r);
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                      float* __restrict__ out, int rows, int N, int K, int rows_per_split) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 32 * TN_STRIDE];      // [buffer][operand][32 rows]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = blockIdx.y * 128, k0 = blockIdx.x * 128;
  const int r_begin = blockIdx.z * rows_per_split;
  const int r_end = min(rows, r_begin + rows_per_split);
  const int n_slabs = (r_end - r_begin + 31) / 32;

  // a thread's two 16-byte pieces of a slab of each operand: slab row lr + 16 i, columns 8 lc .. + 7
  const int lr = tid >> 4, lc = tid & 15;
  const bool a_ok = n0 + lc * 8 + 8 <= N, b_ok = k0 + lc * 8 + 8 <= K;       // N, K multiples of 8 (checked by the entry)
  auto load = [&](uint4 (&ra)[2], uint4 (&rb)[2], int slab) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = r_begin + slab * 32 + lr + 16 * i;
      ra[i] = make_uint4(0, 0, 0, 0);
      rb[i] = make_uint4(0, 0, 0, 0);
      if (r < r_end && a_ok) ra[i] = ld_global16(A + (int64_t)r * N + n0 + lc * 8);
      if (r < r_end && b_ok) rb[i] = ld_global16(B + (int64_t)r * K + k0 + lc * 8);
    }
  };
  auto stash = [&](const uint4 (&ra)[2], const uint4 (&rb)[2], int buf) {
    char* sa = smem + buf * (2 * 32 * TN_STRIDE);
    char* sb = sa + 32 * TN_STRIDE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<uint4*>(sa + (lr + 16 * i) * TN_STRIDE + lc * 16) = ra[i];
      *reinterpret_cast<uint4*>(sb + (lr + 16 * i) * TN_STRIDE + lc * 16) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 ra[2], rb[2];
  if (n_slabs > 0) {
    load(ra, rb, 0);
    stash(ra, rb, 0);
  }
  __syncthreads();
  // fragment of column tile t (16 columns from column c0 + 16 t): lane = column fr, slab rows 8 fg .. 8 fg + 7
  const int frag_off = (fg * 8 + (fr >> 2)) * TN_STRIDE + (fr & 3) * 8;
  for (int s = 0; s < n_slabs; ++s) {
    const int buf = s & 1;
    if (s + 1 < n_slabs) load(ra, rb, s + 1);
    const char* sa = smem + buf * (2 * 32 * TN_STRIDE) + frag_off + wn * 128;
    const char* sb = smem + buf * (2 * 32 * TN_STRIDE) + 32 * TN_STRIDE + frag_off + wk * 128;
    uint4 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint2 lo = tn_read_tr16(sa + i * 32), hi = tn_read_tr16(sa + i * 32 + 4 * TN_STRIDE);
      fa[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint2 lo = tn_read_tr16(sb + j * 32), hi = tn_read_tr16(sb + j * 32 + 4 * TN_STRIDE);
      fb[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) Mma<bf16_t>::chunk(acc[i][j], fa[i], fb[j]);
    if (s + 1 < n_slabs) stash(ra, rb, buf ^ 1);        // the other buffer: its last readers passed the barrier below
    __syncthreads();
  }

  const bool split = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * 64 + i * 16 + fg * 4 + r;
      if (n >= N) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wk * 64 + j * 16 + fr;
        if (k >= K) continue;
        if (split) unsafeAtomicAdd(out + (int64_t)n * K + k, acc[i][j][r]);       // hardware f32 add (no CAS loop)
        else out[(int64_t)n * K + k] = acc[i][j][r];
      }
    }
}

}  // namespace

extern "C" int xml_gemm_tn_supported(int64_t rows, int N, int K, int dt) {
  return dt == XML_BF16 && rows > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0;
}

extern "C" int xml_gemm_tn(const void* A, const void* B, float* out, int64_t rows, int N, int K, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!A || !B || !out || rows <= 0 || rows > 0x7fffffff || N <= 0 || K <= 0) return XML_ERR_BAD_ARG;
  if (!xml_gemm_tn_supported(rows, N, K, dt)) return XML_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int tiles = cdiv(N, 128) * cdiv(K, 128);
  int splits = cdiv(768, tiles);                               // aim at >= 3 workgroups per CU
  int rps = (cdiv(rows, splits) + 31) / 32 * 32;               // rows per workgroup, whole 32-row slabs
  if (rps < 256) rps = 256;
  splits = cdiv(rows, rps);
  if (splits > 1 && hipMemsetAsync(out, 0, (size_t)N * K * 4, st) != hipSuccess) return XML_ERR_LAUNCH;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(cdiv(K, 128), cdiv(N, 128), splits), dim3(256), 0, st, (const bf16_t*)A,
                     (const bf16_t*)B, out, (int)rows, N, K, rps);
  XML_CHECK_LAUNCH();
  return XML_OK;
}
