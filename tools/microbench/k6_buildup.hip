// Build-up of K6's inner loop from clean ingredients, to price each one (profiles/r01_k6_notes.md):
//   LDS ring of 4 x 32 KiB slices, 8 waves; per slice and wave: 4 LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB
//   each) for the slice three ahead, 12 ds_read_b128 fragment reads of the next slice (register double buffer),
//   32 MFMA 16x16x32 bf16 on the current one, one s_barrier.
//   hipcc --offload-arch=gfx950 -O3 -o k6_buildup k6_buildup.hip && ./k6_buildup
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4;

__device__ __forceinline__ void dma_quad(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, const char* sb,
                                         uint32_t lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5"
      :
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(lds_dst), "s"(sb)
      : "memory", "scc");
}

__device__ __forceinline__ void g_dma_pair(uint32_t v0, const char* sb, uint32_t lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3"
      :
      : "v"(v0), "v"(v0 + 1024), "s"(lds_dst), "s"(sb)
      : "memory", "scc");
}

#define G_DMA_PAIR_MOD(NAME, MOD)                                                                         \
  __device__ __forceinline__ void NAME(uint32_t v0, const char* sb, uint32_t lds_dst) {                   \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3 " MOD "\n\t"              \
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 " MOD              \
                 :                                                                                        \
                 : "v"(v0), "v"(v0 + 1024), "s"(lds_dst), "s"(sb)                                         \
                 : "memory", "scc");                                                                      \
  }
G_DMA_PAIR_MOD(g_dma_pair_nt, "nt")
G_DMA_PAIR_MOD(g_dma_pair_sc0, "sc0")
G_DMA_PAIR_MOD(g_dma_pair_sc1, "sc1")
G_DMA_PAIR_MOD(g_dma_pair_sc01, "sc0 sc1")

struct Frags { i32x4 a[4], b[8]; };

template <bool READS>
__device__ __forceinline__ void read_frags(Frags& f, const char* slot, int wave, int lane, int salt) {
  if (READS) {
    const char* pa = slot + (wave >> 1) * 4096 + lane * 16;
    const char* pb = slot + 16384 + (wave & 1) * 8192 + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = *reinterpret_cast<const i32x4*>(pa + i * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i) f.b[i] = *reinterpret_cast<const i32x4*>(pb + i * 1024);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = i32x4{salt, i, lane, 1};
#pragma unroll
    for (int i = 0; i < 8; ++i) f.b[i] = i32x4{salt, i, lane, 2};
  }
}

template <bool MFMA>
__device__ __forceinline__ void mma(f32x4 (&acc)[32], const Frags& f, i32x4& sink) {
  if (MFMA) {
#pragma unroll
    for (int i = 0; i < 32; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.a[i & 3]),
                                                       __builtin_bit_cast(bf16x8, f.b[i >> 2]), acc[i], 0, 0, 0);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) sink ^= f.a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) sink ^= f.b[i];
    asm volatile("" : "+v"(sink));
  }
}

// SRC 0: every workgroup re-reads its own 64 KiB window, pieces are 1 KiB contiguous (best case for the memory side).
// SRC 1: K6's access pattern: operand rows of 1536 B (H = 768 bf16), a piece = 16 rows x 64 B of the current K slice;
//        the 32 workgroups of an XCD form an 8 x 4 super-tile: 8 query tiles (A, 384 KiB each, re-read for every
//        clip tile -> L2 hits) x 4 clip tiles (B, streamed from a 2 GiB buffer, each shared by 8 workgroups).
// SRC 2: same tiles and sharing as SRC 1, but every operand tile is stored slice-major: [24 K slices][256 rows][64 B]
//        (= the LDS image of each slice, 16 KiB contiguous): every piece is 1 KiB contiguous, every line used once.
// SRC 3: SRC 1 with rows padded to 1664 B (13 x 128 B: an odd number of lines, spreads the L2 channels).
template <bool DMA, bool READS, bool MFMA, bool BARRIER, int SRC = 0>
__global__ __launch_bounds__(512) void k6_loop(const char* __restrict__ src, float* out, int slices) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 128 * 1024 / 4; i += 512) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u + (i & 0xff);
  __syncthreads();
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const char* sb = src + (size_t)blockIdx.x * 65536;         // 64 KiB source window per workgroup: L2 hits
  f32x4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x4 sink = {0, 0, 0, 0};
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  constexpr uint32_t ROWB = SRC == 3 ? 1664 : 1536;
  constexpr size_t TILEB = (size_t)256 * ROWB;
  const char* a_base = src + ((size_t)(xcd * 8 + (loc & 7))) * TILEB;                     // 64 A tiles at the front
  const char* b_base = src + (size_t)64 * TILEB;
  const uint32_t rowoff = SRC == 2 ? (uint32_t)(wave * 2048 + lane * 16)                  // pieces 2w, 2w+1 of a slice
                                   : (uint32_t)((wave * 32 + (lane >> 2)) * ROWB + (lane & 3) * 16);
  auto issue = [&](int s) {
    if (DMA && SRC == 0) {
      const uint32_t v = (uint32_t)(((s & 1) * 32768) + wave * 4096 + lane * 16);
      dma_quad(v, v + 1024, v + 2048, v + 3072, sb, lds0 + (uint32_t)((s & 3) * 32768 + wave * 4096));
    } else if (DMA) {
      const int t = s / 24, ks = s - t * 24;                                              // tile, K slice in the tile
      const uint32_t blk = (uint32_t)(((t * 8 + xcd) * 4 + (loc >> 3)) % 5000);           // B tile of this round
      const char* bb = b_base + (size_t)blk * TILEB;
      const uint32_t v = rowoff + (uint32_t)ks * (SRC == 2 ? 16384 : 64);
      const uint32_t v2 = v + (SRC == 2 ? 1024 : 16 * ROWB);
      asm volatile(
          "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
          "s_add_u32 m0, m0, 0x3c00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %4\n\t"
          "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4"
          :
          : "v"(v), "v"(v2), "s"(lds0 + (uint32_t)((s & 3) * 32768 + wave * 2048)), "s"(a_base), "s"(bb)
          : "memory", "scc");
    }
  };
  issue(0); issue(1); issue(2); issue(3);
  if (DMA) __builtin_amdgcn_s_waitcnt(0x0f70 | 12);          // slice 0 landed (12 pieces still in flight)
  __syncthreads();
  Frags f0, f1;
  read_frags<READS>(f0, smem, wave, lane, 0);
  auto step = [&](int s, Frags& cur, Frags& nxt) {
    if (DMA) __builtin_amdgcn_s_waitcnt(0x0f70 | 8);          // own pieces of slice s + 1 landed
    if (BARRIER) __builtin_amdgcn_s_barrier();                // everyone's: slice s + 1 readable, slot s & 3 free
    issue(s + 4);
    read_frags<READS>(nxt, smem + ((s + 1) & 3) * 32768, wave, lane, s);
    mma<MFMA>(acc, cur, sink);
  };
  for (int s = 0; s < slices; s += 2) {
    step(s, f0, f1);
    step(s + 1, f1, f0);
  }
  if (DMA) __builtin_amdgcn_s_waitcnt(0x0f70);
  float r = (float)(sink[0] ^ sink[1] ^ sink[2] ^ sink[3]) + (float)f0.a[0][0];
#pragma unroll
  for (int i = 0; i < 32; ++i) r += acc[i][0] + acc[i][3];
  out[blockIdx.x * 512 + tid] = r;
}


// ---- 4-wave variant: wave tile 128 x 128 (64 accumulators, AGPRs), one wave per SIMD --------------------------------
// LDS fragment traffic per MFMA drops by a third (16 KiB per 64 MFMA instead of 12 KiB per 32); the single wave has
// to interleave its own DMA issue / fragment reads with its MFMA stream.
struct Frags4 { i32x4 a[8], b[8]; };

template <bool DMA, bool MFMA, int SRC>
__global__ __launch_bounds__(256) void k6_loop4(const char* __restrict__ src, float* out, int slices) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 128 * 1024 / 4; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u + (i & 0xff);
  __syncthreads();
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const char* sb = src + (size_t)blockIdx.x * 65536;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  constexpr size_t TILEB = (size_t)256 * 1536;
  const char* a_base = src + ((size_t)(xcd * 8 + (loc & 7))) * TILEB;
  const char* b_base = src + (size_t)64 * TILEB;
  f32x4 acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto issue = [&](int s) {
    if (!DMA) return;
    const uint32_t dst = lds0 + (uint32_t)((s & 3) * 32768 + wave * 4096);
    if (SRC == 0) {
      const uint32_t v = (uint32_t)(((s & 1) * 32768) + wave * 4096 + lane * 16);
      dma_quad(v, v + 1024, v + 2048, v + 3072, sb, dst);
      dma_quad(v + 16384, v + 17408, v + 18432, v + 19456, sb, dst + 16384);
    } else {      // slice-major tiles, K6's sharing pattern: 4 pieces of A and 4 of B per wave
      const int t = s / 24, ks = s - t * 24;
      const uint32_t blk = (uint32_t)(((t * 8 + xcd) * 4 + (loc >> 3)) % 5000);
      const uint32_t v = (uint32_t)(ks * 16384 + wave * 4096 + lane * 16);
      dma_quad(v, v + 1024, v + 2048, v + 3072, a_base, dst);
      dma_quad(v, v + 1024, v + 2048, v + 3072, b_base + (size_t)blk * TILEB, dst + 16384);
    }
  };
  auto read = [&](Frags4& f, int s) {
    const char* slot = smem + (s & 3) * 32768;
    const char* pa = slot + (wave >> 1) * 8192 + lane * 16;
    const char* pb = slot + 16384 + (wave & 1) * 8192 + lane * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) f.a[i] = *reinterpret_cast<const i32x4*>(pa + i * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i) f.b[i] = *reinterpret_cast<const i32x4*>(pb + i * 1024);
  };
  issue(0); issue(1); issue(2); issue(3);
  if (DMA) __builtin_amdgcn_s_waitcnt(0x4f70 | 8);            // vmcnt(24): slice 0 landed
  __syncthreads();
  Frags4 f0, f1;
  read(f0, 0);
  i32x4 sink = {0, 0, 0, 0};
  auto step = [&](int s, Frags4& cur, Frags4& nxt) {
    if (DMA) __builtin_amdgcn_s_waitcnt(0x4f70 | 0);          // vmcnt(16): own pieces of slice s + 1 landed
    __builtin_amdgcn_s_barrier();
    issue(s + 4);
    read(nxt, s + 1);
    if (MFMA) {
#pragma unroll
      for (int i = 0; i < 64; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.a[i & 7]),
                                                         __builtin_bit_cast(bf16x8, cur.b[i >> 3]), acc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) sink ^= cur.a[i] ^ cur.b[i];
      asm volatile("" : "+v"(sink));
    }
  };
  for (int s = 0; s < slices; s += 2) {
    step(s, f0, f1);
    step(s + 1, f1, f0);
  }
  if (DMA) __builtin_amdgcn_s_waitcnt(0x0f70);
  float r = (float)(sink[0] ^ sink[3]) + (float)f0.a[0][0];
#pragma unroll
  for (int i = 0; i < 64; ++i) r += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + tid] = r;
}

template <typename K>
static void run4(const char* name, K kernel, const char* src, float* out, int slices, bool mfma) {
  hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(256), dim3(256), 128 * 1024, 0, src, out, slices);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double tf = (double)slices * 4 * 64 * 256 * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12;
  printf("%-44s %8.3f ms  %7.1f ns per slice", name, ms, ms * 1e6 / slices);
  if (mfma) printf("  %7.1f TFLOP/s", tf);
  printf("\n");
}


// ---- ring-depth experiment: separate rings for the two operands, SA slots of 16 KiB for A (L2-resident re-reads), SB
// for B (streamed) -- K6's source pattern on slice-major tiles.  (4,4) is the symmetric 4-slot ring, (5,5) the 5-slot one.
// MA / MB: cache-policy modifier of the A / B loads (0 none, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1)
template <int SA, int SB, int MA = 0, int MB = 0>
__global__ __launch_bounds__(512) void k6_ring(const char* __restrict__ src, float* out, int slices) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int A_BYTES = SA * 16384;
  for (int i = tid; i < (SA + SB) * 16384 / 4; i += 512) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u + (i & 0xff);
  __syncthreads();
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  constexpr size_t TILEB = (size_t)256 * 1536;
  const char* a_base = src + ((size_t)(xcd * 8 + (loc & 7))) * TILEB;
  const char* b_base = src + (size_t)64 * TILEB;
  f32x4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t rowoff = (uint32_t)(wave * 2048 + lane * 16);
  auto issue_a = [&](int s) {
    const int t = s / 24, ks = s - t * 24;
    const uint32_t v = (uint32_t)(rowoff + ks * 16384), d = lds0 + (uint32_t)((s % SA) * 16384 + wave * 2048);
    if (MA == 1) g_dma_pair_nt(v, a_base, d); else if (MA == 2) g_dma_pair_sc0(v, a_base, d);
    else if (MA == 3) g_dma_pair_sc1(v, a_base, d); else if (MA == 4) g_dma_pair_sc01(v, a_base, d);
    else g_dma_pair(v, a_base, d);
  };
  auto issue_b = [&](int s) {
    const int t = s / 24, ks = s - t * 24;
    const uint32_t blk = (uint32_t)(((t * 8 + xcd) * 4 + (loc >> 3)) % 5000);
    const uint32_t v = (uint32_t)(rowoff + ks * 16384), d = lds0 + (uint32_t)(A_BYTES + (s % SB) * 16384 + wave * 2048);
    const char* bb = b_base + (size_t)blk * TILEB;
    if (MB == 1) g_dma_pair_nt(v, bb, d); else if (MB == 2) g_dma_pair_sc0(v, bb, d);
    else if (MB == 3) g_dma_pair_sc1(v, bb, d); else if (MB == 4) g_dma_pair_sc01(v, bb, d);
    else g_dma_pair(v, bb, d);
  };
  // prologue: A slices 0 .. SA-1, B slices 0 .. SB-1, in the steady-state order [A(t + SA), B(t + SB)] of steps t < 0
  for (int t = -SB; t < 0; ++t) {
    if (t + SA >= 0) issue_a(t + SA);
    issue_b(t + SB);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  Frags f0, f1;
  auto read = [&](Frags& f, int s) {
    const char* pa = smem + (s % SA) * 16384 + (wave >> 1) * 4096 + lane * 16;
    const char* pb = smem + A_BYTES + (s % SB) * 16384 + (wave & 1) * 8192 + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = *reinterpret_cast<const i32x4*>(pa + i * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i) f.b[i] = *reinterpret_cast<const i32x4*>(pb + i * 1024);
  };
  read(f0, 0);
  i32x4 sink = {0, 0, 0, 0};
  constexpr int YOUNGER = 2 + 4 * (SA - 2);          // pieces issued after A(s + 1) (B(s + 1) is older still)
  auto step = [&](int s, Frags& cur, Frags& nxt) {
    __builtin_amdgcn_s_waitcnt(0x0f70 | (YOUNGER & 15) | ((YOUNGER >> 4) << 14));
    __builtin_amdgcn_s_barrier();
    issue_a(s + SA);
    issue_b(s + SB);
    read(nxt, s + 1);
    mma<true>(acc, cur, sink);
  };
  for (int s = 0; s < slices; s += 2) {
    step(s, f0, f1);
    step(s + 1, f1, f0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float r = (float)f0.a[0][0];
#pragma unroll
  for (int i = 0; i < 32; ++i) r += acc[i][0] + acc[i][3];
  out[blockIdx.x * 512 + tid] = r;
}

template <typename K>
static void run_ring(const char* name, K kernel, int lds, const char* src, float* out, int slices) {
  hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(256), dim3(512), lds, 0, src, out, slices);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double tf = (double)slices * 8 * 32 * 256 * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12;
  printf("%-44s %8.3f ms  %7.1f ns per slice  %7.1f TFLOP/s\n", name, ms, ms * 1e6 / slices, tf);
}

template <typename K>
static void run(const char* name, K kernel, const char* src, float* out, int slices, bool mfma) {
  hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(256), dim3(512), 128 * 1024, 0, src, out, slices);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double per_slice_ns = ms * 1e6 / slices;
  const double tf = (double)slices * 8 * 32 * 256 * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12;
  printf("%-44s %8.3f ms  %7.1f ns per slice", name, ms, per_slice_ns);
  if (mfma) printf("  %7.1f TFLOP/s", tf);
  printf("\n");
}

int main(int argc, char** argv) {
  int slices = argc > 1 ? atoi(argv[1]) : 40000;
  char* src;
  float* out;
  const size_t src_bytes = (size_t)(64 + 5000 + 1) * 256 * 1664;      // 64 A tiles + 5000 B tiles (1.97 GB)
  hipMalloc(&src, src_bytes);
  const bool random = argc > 2 && atoi(argv[2]) != 0;             // full-entropy bf16 operands: switching power
  if (random) {
    uint16_t* h = (uint16_t*)malloc(64 << 20);
    uint32_t x = 12345;
    for (size_t i = 0; i < (64u << 20) / 2; ++i) {
      x = x * 1664525u + 1013904223u;
      h[i] = (uint16_t)(((x >> 16) & 0x807f) | (((x >> 9) & 7) + 0x78) << 7);   // sign, 3-bit exponent spread, mantissa
    }
    for (size_t off = 0; off < src_bytes; off += (64u << 20))
      hipMemcpy(src + off, h, src_bytes - off < (64u << 20) ? src_bytes - off : (64u << 20), hipMemcpyHostToDevice);
    free(h);
  } else {
    hipMemset(src, 0x3c, src_bytes);
  }
  printf("operands: %s\n", random ? "random bf16" : "constant");
  hipMalloc(&out, 256 * 512 * sizeof(float));
  //                                                  DMA    READS  MFMA   BARRIER
  run("MFMA only", k6_loop<false, false, true, false>, src, out, slices, true);
  run("MFMA + barrier", k6_loop<false, false, true, true>, src, out, slices, true);
  run("reads + MFMA", k6_loop<false, true, true, false>, src, out, slices, true);
  run("reads + MFMA + barrier", k6_loop<false, true, true, true>, src, out, slices, true);
  run("DMA + MFMA + barrier (no reads)", k6_loop<true, false, true, true>, src, out, slices, true);
  run("DMA + reads + MFMA + barrier  (= K6 loop)", k6_loop<true, true, true, true>, src, out, slices, true);
  run("K6 loop, K6 source pattern (rows, L2 + HBM)", k6_loop<true, true, true, true, 1>, src, out, slices, true);
  run("  same without MFMA", k6_loop<true, true, false, true, 1>, src, out, slices, false);
  run("K6 loop, slice-major tiles (L2 + HBM)", k6_loop<true, true, true, true, 2>, src, out, slices, true);
  run("  same without MFMA", k6_loop<true, true, false, true, 2>, src, out, slices, false);
  run("K6 loop, rows padded to 1664 B", k6_loop<true, true, true, true, 3>, src, out, slices, true);
  run("  same without MFMA", k6_loop<true, true, false, true, 3>, src, out, slices, false);
  run4("4 waves 128x128: ideal source", k6_loop4<true, true, 0>, src, out, slices, true);
  run4("4 waves 128x128: slice-major tiles (L2+HBM)", k6_loop4<true, true, 2>, src, out, slices, true);
  run4("  same without MFMA", k6_loop4<true, false, 2>, src, out, slices, false);
  run4("4 waves 128x128: no DMA", k6_loop4<false, true, 0>, src, out, slices, true);
  run_ring("rings A 4 / B 4 slots (128 KiB)", k6_ring<4, 4>, 8 * 16384, src, out, slices);
  run_ring("rings A 5 / B 5 slots (160 KiB)", k6_ring<5, 5>, 10 * 16384, src, out, slices);
  run_ring("rings A 4 / B 6 slots (160 KiB)", k6_ring<4, 6>, 10 * 16384, src, out, slices);
  run_ring("rings A 3 / B 7 slots (160 KiB)", k6_ring<3, 7>, 10 * 16384, src, out, slices);
  run_ring("rings A 6 / B 4 slots (160 KiB)", k6_ring<6, 4>, 10 * 16384, src, out, slices);
  run_ring("5 / 5, B loads nt", k6_ring<5, 5, 0, 1>, 10 * 16384, src, out, slices);
  run_ring("5 / 5, B loads sc0", k6_ring<5, 5, 0, 2>, 10 * 16384, src, out, slices);
  run_ring("5 / 5, B loads sc1", k6_ring<5, 5, 0, 3>, 10 * 16384, src, out, slices);
  run_ring("5 / 5, B loads sc0 sc1", k6_ring<5, 5, 0, 4>, 10 * 16384, src, out, slices);
  run_ring("5 / 5, A loads nt", k6_ring<5, 5, 1, 0>, 10 * 16384, src, out, slices);
  run_ring("5 / 5, A loads sc1", k6_ring<5, 5, 3, 0>, 10 * 16384, src, out, slices);
  run("DMA + reads + barrier (no MFMA)", k6_loop<true, true, false, true>, src, out, slices, false);
  run("DMA + barrier only", k6_loop<true, false, false, true>, src, out, slices, false);
  run("reads + barrier only", k6_loop<false, true, false, true>, src, out, slices, false);
  hipFree(src);
  hipFree(out);
  return 0;
}
