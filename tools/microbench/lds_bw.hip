// LDS read bandwidth per CU on gfx950 (ds_read_b128, conflict-free), alone and interleaved with independent MFMAs
// at K6's ratio (12 reads per 32 MFMAs and wave).   hipcc --offload-arch=gfx950 -O3 -o lds_bw lds_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4;

// MODE 0: reads only; 1: reads + MFMAs consuming them; 2: MFMAs only (same loop shape)
// BAR: 0 = no barrier, n = s_barrier every n-th slice
template <int MODE, int BAR = 0>
__global__ __launch_bounds__(512) void lds_kernel(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 128 * 1024 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = (float)(i & 255) * 0.01f;
  __syncthreads();
  f32x4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x4 keep = {0, 0, 0, 0};
  // each wave walks its own 12 KB window per "slice": 12 x (64 lanes x 16 B), consecutive lanes -> consecutive 16 B
  const char* base = smem + wave * 12288 + lane * 16;
  for (int it = 0; it < iters; ++it) {
    if (BAR && (it % BAR) == 0) __builtin_amdgcn_s_barrier();
    const char* p = base + (it & 1) * 1024 * 0;   // same window: bandwidth, not capacity
    i32x4 f[12];
    if (MODE != 2) {
#pragma unroll
      for (int r = 0; r < 12; ++r) f[r] = *reinterpret_cast<const i32x4*>(p + r * 1024);
    } else {
#pragma unroll
      for (int r = 0; r < 12; ++r) f[r] = keep + r;
    }
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 12; ++r) keep ^= f[r];
      asm volatile("" : "+v"(keep));
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        bf16x8 a = __builtin_bit_cast(bf16x8, f[i & 3]);
        bf16x8 b = __builtin_bit_cast(bf16x8, f[4 + (i >> 2)]);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      }
    }
  }
  float s = (float)(keep[0] + keep[1] + keep[2] + keep[3]);
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 512 + tid] = s;
}

template <typename K>
static void run(const char* name, K kernel, int iters, bool reads, bool mfma, float* out) {
  hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(256), dim3(512), 128 * 1024, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double bytes_cu = (double)iters * 8 * 12 * 1024;            // per CU
  const double mf = (double)iters * 8 * 32 * 256 * 2.0 * 16 * 16 * 32;
  printf("%-28s %8.3f ms", name, ms);
  if (reads) printf("   LDS read %7.1f GB/s per CU (%5.1f B/clk at 2.4 GHz, %5.1f at 1.95)", bytes_cu / ms / 1e6,
                    bytes_cu / (ms * 1e-3) / 2.4e9, bytes_cu / (ms * 1e-3) / 1.95e9);
  if (mfma) printf("   %7.1f TFLOP/s", mf / (ms * 1e-3) / 1e12);
  printf("\n");
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  run("reads only", lds_kernel<0>, iters, true, false, out);
  run("reads + 32 MFMA per 12 reads", lds_kernel<1>, iters, true, true, out);
  run("MFMA only (same loop)", lds_kernel<2>, iters, false, true, out);
  run("reads + MFMA, barrier / slice", lds_kernel<1, 1>, iters, true, true, out);
  run("reads + MFMA, barrier / 2 slices", lds_kernel<1, 2>, iters, true, true, out);
  run("reads + MFMA, barrier / 4 slices", lds_kernel<1, 4>, iters, true, true, out);
  run("MFMA only, barrier / slice", lds_kernel<2, 1>, iters, false, true, out);
  hipFree(out);
  return 0;
}
