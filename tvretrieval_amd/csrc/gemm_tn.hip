// Weight-gradient GEMM of the training step, straight from the row-major operands:
//   dW (N, K) f32  =  dY^T X  =  sum_r dY[r][n] X[r][k]        dY (rows, N), X (rows, K) bf16, both as the layers hold them
//   reference: the autograd of nn.Linear inside XML.forward (xml/model_components.py:156-163; loss.backward(), xml/train.py:81)
// Both operands are contracted over their ROW index, i.e. both are needed transposed.  The unfused path transposed dY and X
// explicitly (two launches writing rows x (N + K) elements) and ran a split-K NT GEMM on the copies; here 32-row slabs of
// dY and X go to LDS as they lie (row-major, 16-byte loads / stores) and the MFMA fragments -- 8 consecutive rows of one
// column per lane -- come out of ds_read_b64_tr_b16 (attention.hip has the lane semantics).
// One workgroup = 4 waves = a 128 (n) x 128 (k) output tile over one range of rows; the row ranges (blockIdx.z) are combined
// with f32 atomics into the pre-zeroed output, as the split-K kernel did.
#include "common.h"

namespace {

constexpr int TN_STRIDE = 128 * 2 + 16;      // LDS row of a slab: 128 columns bf16 + 16 bytes pad

typedef short tn_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 tn_read_tr16(const char* p) {
  const tn_v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_v4s*)p);
  return __builtin_bit_cast(uint2, r);
}

// colsum (optional): colsum[n] += sum_r A[r][n] -- the bias gradient of the same layer, taken from the dY slabs the
// workgroups of the first k tile load anyway (one launch and one pass over dY less per layer).
__global__ __launch_bounds__(256, 3) void gemm_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                      float* __restrict__ out, float* __restrict__ colsum, int rows, int N,
                                                      int K, int rows_per_split, int accumulate) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 32 * TN_STRIDE];      // [buffer][operand][32 rows]
  __shared__ float s_cs[128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = blockIdx.y * 128, k0 = blockIdx.x * 128;
  const int r_begin = blockIdx.z * rows_per_split;
  const int r_end = min(rows, r_begin + rows_per_split);
  const int n_slabs = ((r_end - r_begin + 31) / 32 + 1) & ~1;      // even: the two-step loop body below has no tail case (a
                                                                   // slab past r_end is all zeros)

  // a thread's two 16-byte pieces of a slab of each operand: slab row lr + 16 i, columns 8 lc .. + 7
  const int lr = tid >> 4, lc = tid & 15;
  const bool a_ok = n0 + lc * 8 + 8 <= N, b_ok = k0 + lc * 8 + 8 <= K;       // N, K multiples of 8 (checked by the entry)
  const int ca = a_ok ? n0 + lc * 8 : 0, cb = b_ok ? k0 + lc * 8 : 0;        // clamped columns
  auto load = [&](uint4 (&ra)[2], uint4 (&rb)[2], int slab) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // UNCONDITIONAL loads from clamped (always valid) addresses; rows / columns outside the operands are zeroed when the
      // registers are parked in LDS (stash).  With the loads inside `if`s hipcc loses its vmcnt count at the branch joins and
      // waits vmcnt(0) before every LDS store, and a select right here would wait for the data at once: either way the
      // two-slab lead is gone.
      const int r = r_begin + slab * 32 + lr + 16 * i;
      const int rc = r < r_end ? r : r_begin;
      ra[i] = ld_global16(A + (int64_t)rc * N + ca);
      rb[i] = ld_global16(B + (int64_t)rc * K + cb);
    }
  };
  const bool do_cs = colsum != nullptr && blockIdx.x == 0;      // (workgroup-uniform)
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto stash = [&](const uint4 (&ra)[2], const uint4 (&rb)[2], int buf, int slab) {
    char* sa = smem + buf * (2 * 32 * TN_STRIDE);
    char* sb = sa + 32 * TN_STRIDE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool rok = r_begin + slab * 32 + lr + 16 * i < r_end;
      const bool ka = rok && a_ok, kb = rok && b_ok;
      const uint4 za = make_uint4(ka ? ra[i].x : 0u, ka ? ra[i].y : 0u, ka ? ra[i].z : 0u, ka ? ra[i].w : 0u);
      if (do_cs) {
        float f[8];
        unpack16<bf16_t>(za, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] += f[e];
      }
      *reinterpret_cast<uint4*>(sa + (lr + 16 * i) * TN_STRIDE + lc * 16) = za;
      *reinterpret_cast<uint4*>(sb + (lr + 16 * i) * TN_STRIDE + lc * 16) =
          make_uint4(kb ? rb[i].x : 0u, kb ? rb[i].y : 0u, kb ? rb[i].z : 0u, kb ? rb[i].w : 0u);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Loads run TWO slabs ahead through two register sets (a 16-MFMA slab is ~500 cycles of work, a global round trip
  // several times that: with one slab of lead every iteration waited for memory -- 64 us per 768 x 768 x 12 800 product):
  //   step s:  issue the loads of slab s + 2 into the set slab s came from, compute slab s from its LDS buffer, then park
  //            slab s + 1 (loaded one step earlier) in the other buffer.
  uint4 r0a[2], r0b[2], r1a[2], r1b[2];
  load(r0a, r0b, 0);
  load(r1a, r1b, 1);
  stash(r0a, r0b, 0, 0);
  __syncthreads();
  // fragment of column tile t (16 columns from column c0 + 16 t): lane = column fr, slab rows 8 fg .. 8 fg + 7
  const int frag_off = (fg * 8 + (fr >> 2)) * TN_STRIDE + (fr & 3) * 8;
  auto step = [&](int s, uint4 (&la)[2], uint4 (&lb)[2], const uint4 (&sa_)[2], const uint4 (&sb_)[2]) {
    const int buf = s & 1;
    load(la, lb, s + 2 < n_slabs ? s + 2 : n_slabs - 1);      // (unconditional: a branch here costs the vmcnt count again)
    __builtin_amdgcn_sched_barrier(0);                         // keep the loads HERE (hipcc sinks them below the MFMAs)
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
    stash(sa_, sb_, buf ^ 1, s + 1);      // the other buffer (after the last slab: zeros nobody reads): its last readers passed the previous barrier
    __syncthreads();
  };
  for (int s = 0; s < n_slabs; s += 2) {
    step(s, r0a, r0b, r1a, r1b);
    step(s + 1, r1a, r1b, r0a, r0b);
  }

  if (do_cs) {        // the 16 threads that share a column group meet in LDS, one global atomic per column
    if (tid < 128) s_cs[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(&s_cs[lc * 8 + e], cs[e]);
    __syncthreads();
    if (tid < 128 && n0 + tid < N) unsafeAtomicAdd(colsum + n0 + tid, s_cs[tid]);
  }
  const bool split = gridDim.z > 1 || accumulate;      // accumulate: out already holds a partial sum (a .grad buffer)
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

extern "C" int xml_gemm_tn(const void* A, const void* B, float* out, float* colsum_a, int64_t rows, int N, int K, int dt,
                           int accumulate, xml_stream_t stream) {
  XML_ENTER();
  if (!A || !B || !out || rows <= 0 || rows > 0x7fffffff || N <= 0 || K <= 0) return XML_ERR_BAD_ARG;
  if (!xml_gemm_tn_supported(rows, N, K, dt)) return XML_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int tiles = cdiv(N, 128) * cdiv(K, 128);
  // workgroups aimed at.  Every row range adds N x K f32 atomics, and device-scope float atomics are slow enough to show
  // (768 x 768 over 12 800 rows: 58.6 us with 7 ranges, 69.6 with 22, 94 with 43 -- tools/bench_gemm_tn.py); with many
  // output tiles the extra ranges pay for themselves by hiding the global-load latency (768 x 3072: 134 vs 160 us)
  int target = (tiles <= 48 || rows < 8192) ? 256 : 512;      // (3 840 rows x 2304 x 768: 46 us at 256, 55 at 512, 67 at 768)
#ifdef XML_DEBUG_VARIANTS
  if (g_q2c_ablation >= 200) target = (g_q2c_ablation - 200) * 64;      // A/B: workgroups aimed at = (XML_ABL - 200) x 64
#endif
  int splits = cdiv(target, tiles);
  int rps = (cdiv(rows, splits) + 31) / 32 * 32;               // rows per workgroup, whole 32-row slabs
  if (rps < 256) rps = 256;
  splits = cdiv(rows, rps);
  if (accumulate) {
    // out / colsum_a are gradient buffers that already hold a (possibly zero) partial sum: every contribution is an atomic add
  } else if (splits > 1 && colsum_a == out + (size_t)N * K) {   // caller laid them out back to back: one fill
    if (!xml_zero_async(out, ((size_t)N * K + N) * 4, st)) return XML_ERR_LAUNCH;
  } else {
    if (splits > 1 && !xml_zero_async(out, (size_t)N * K * 4, st)) return XML_ERR_LAUNCH;
    if (colsum_a && !xml_zero_async(colsum_a, (size_t)N * 4, st)) return XML_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(cdiv(K, 128), cdiv(N, 128), splits), dim3(256), 0, st, (const bf16_t*)A,
                     (const bf16_t*)B, out, colsum_a, (int)rows, N, K, rps, accumulate ? 1 : 0);
  XML_CHECK_LAUNCH();
  return XML_OK;
}
