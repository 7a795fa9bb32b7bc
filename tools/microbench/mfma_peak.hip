// Sustained MFMA issue rate on gfx950, no memory traffic: the practical ceiling K6 is priced against in
// profiles/r01_k6_notes.md.  Each wave keeps NACC independent accumulators and loops over them.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__device__ __forceinline__ float rnd(uint32_t x) {      // hash -> (-1, 1): full-entropy mantissas (switching power)
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (float)(int32_t)x * (1.0f / 2147483648.0f);
}

template <int NACC>
__global__ __launch_bounds__(512) void mfma16_kernel(float* out, int iters, float seed) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[4], b[8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      a[i][j] = (__bf16)(seed < 0.f ? rnd(threadIdx.x * 64 + i * 8 + j + blockIdx.x * 77777) : seed + threadIdx.x * 0.001f + i);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      b[i][j] = (__bf16)(seed < 0.f ? rnd(threadIdx.x * 64 + i * 8 + j + 12345 + blockIdx.x * 99991) : seed - threadIdx.x * 0.002f + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(512) void mfma32_kernel(float* out, int iters, float seed) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (__bf16)(seed + threadIdx.x * 0.001f + i);
      b[i][j] = (__bf16)(seed - threadIdx.x * 0.002f + i);
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kernel, int threads, int blocks, int iters, double flops_per_mfma, int nacc,
                float* out, float seed = 1.0f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, iters, seed);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64.0;
    const double mfmas = waves * (double)iters * nacc;
    const double tf = mfmas * flops_per_mfma / (ms * 1e-3) / 1e12;
    // cycles per MFMA and SIMD at an assumed clock: 1024 SIMDs
    const double per_simd = mfmas / 1024.0;
    if (rep == 2)
      printf("%-34s %4d thr x %4d WG: %8.3f ms  %7.1f TFLOP/s   %.2f ns per MFMA and SIMD\n", name, threads, blocks, ms,
             tf, ms * 1e6 / per_simd);
  }
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* out;
  hipMalloc(&out, 4096 * 512 * sizeof(float));
  const double f16 = 2.0 * 16 * 16 * 32, f32 = 2.0 * 32 * 32 * 16;
  run("16x16x32 bf16, 32 acc, 1 wave/SIMD", mfma16_kernel<32>, 256, 256, iters, f16, 32, out);
  run("16x16x32 bf16, 32 acc, 2 waves/SIMD", mfma16_kernel<32>, 512, 256, iters, f16, 32, out);
  run("16x16x32 bf16, 16 acc, 4 waves/SIMD", mfma16_kernel<16>, 512, 512, iters, f16, 16, out);
  run("32x32x16 bf16,  8 acc, 1 wave/SIMD", mfma32_kernel<8>, 256, 256, iters, f32, 8, out);
  run("32x32x16 bf16,  8 acc, 2 waves/SIMD", mfma32_kernel<8>, 512, 256, iters, f32, 8, out);
  // long run: sustained clocks (the first launches run before DVFS settles)
  run("16x16x32 bf16, 32 acc, 2 waves/SIMD, x8 long", mfma16_kernel<32>, 512, 256, iters * 8, f16, 32, out);
  run("32x32x16 bf16,  8 acc, 2 waves/SIMD, x8 long", mfma32_kernel<8>, 512, 256, iters * 8, f32, 8, out);
  run("16x16x32 bf16, 32 acc, 2 w/SIMD, RANDOM data, long", mfma16_kernel<32>, 512, 256, iters * 8, f16, 32, out, -1.0f);
  run("16x16x32 bf16, 32 acc, 2 w/SIMD, const data, long", mfma16_kernel<32>, 512, 256, iters * 8, f16, 32, out, 1.0f);
  hipFree(out);
  return 0;
}
