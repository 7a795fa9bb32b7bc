// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): which LDS elements does lane l get?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
// LDS holds lds[i] = i.  Each lane passes the address of 4 consecutive shorts; the result is printed per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int row_stride) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  // 16-lane group g: lane i of the group points at row (i >> 2), columns (i & 3) * 4 .. + 3 of a [4 rows][16 cols] block
  // whose rows are row_stride shorts apart; group g's block starts 4 rows further down
  const short* p = lds + ((l >> 4) * 4 + ((l & 15) >> 2)) * row_stride + (l & 3) * 4;
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {16, 200}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("row_stride %d shorts: lane -> 4 values as (row, col)\n", stride);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf("  (%d,%2d)", h[l * 4 + j] / stride, h[l * 4 + j] % stride);
      printf("\n");
    }
  }
  return 0;
}
