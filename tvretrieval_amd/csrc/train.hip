// Training-step kernels (SURVEY.md 8 a14 / BASELINE config 5): backward of the XML encoders and scorers + BertAdam.
//   reference: XML.forward + losses      xml/model_xml.py:212-251,588-637   (autograd does the backward there)
//              BertAdam.step             xml/optimization.py:273-338
// Round-1 goal is a CORRECT hand-written backward (parity with the reference's autograd on the golden training-step
// fixture); these kernels reuse the tested 128x128 MFMA mainloop (gemm.h) and simple wave-per-row reductions and
// are not tuned yet.  Activation gradients use the activation dtype, parameter gradients are f32 and ACCUMULATE.
#include "gemm.h"
#include "internal.h"

// ---------------------------------------------------------------------------------------------------------
// batched transpose: y[b][c][r] = x[b][r][c]
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ x, T* __restrict__ y, int rows,
                                                        int cols, int ld_out) {
  __shared__ T tile[32][33];
  const int64_t boff = (int64_t)blockIdx.z * rows * cols;
  const int64_t yoff = (int64_t)blockIdx.z * cols * ld_out;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  T v[4];                    // the four loads first, then the LDS stores (hipcc otherwise waits for each load in turn)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + i * 8, c = c0 + tx;
    v[i] = (r < rows && c < cols) ? x[boff + (int64_t)r * cols + c] : T(0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) tile[ty + i * 8][tx] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + i * 8, r = r0 + tx;
    if (r < rows && c < cols) y[yoff + (int64_t)c * ld_out + r] = tile[tx][ty + i * 8];
  }
}

// bf16, every dimension a multiple of 8: 64 x 64 tiles moved with 16-byte global accesses on both sides (the 32 x 32
// kernel above moves 2 bytes per lane: 64 write requests per store instruction on a write path that retires one request
// per ~5 cycles).  LDS rows are 132 bytes: the 4-byte writes of a row vector and the 2-byte column reads are both
// conflict-free.
__global__ __launch_bounds__(256) void transpose64_bf16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                               int rows, int cols, int ld_out) {
  __shared__ uint32_t tile[64 * 33];
  const int64_t boff = (int64_t)blockIdx.z * rows * cols;
  const int64_t yoff = (int64_t)blockIdx.z * cols * ld_out;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  uint4 in[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {                      // both loads first
    const int v = threadIdx.x + j * 256;
    const int r = v >> 3, c8 = v & 7;
    in[j] = make_uint4(0u, 0u, 0u, 0u);
    if (r0 + r < rows && c0 + c8 * 8 < cols) in[j] = ld_global16(x + boff + (int64_t)(r0 + r) * cols + c0 + c8 * 8);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int v = threadIdx.x + j * 256;
    const int r = v >> 3, c8 = v & 7;
    uint32_t* d = tile + r * 33 + c8 * 4;
    d[0] = in[j].x; d[1] = in[j].y; d[2] = in[j].z; d[3] = in[j].w;
  }
  __syncthreads();
  const unsigned short* t16 = reinterpret_cast<const unsigned short*>(tile);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int v = threadIdx.x + j * 256;
    const int oc = v >> 3, r8 = v & 7;               // output row (= input column) oc, input rows 8 r8 .. 8 r8 + 7
    if (c0 + oc >= cols || r0 + r8 * 8 >= rows) continue;
    uint32_t e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = t16[(r8 * 8 + k) * 66 + oc];
    uint4 o;
    o.x = e[0] | (e[1] << 16); o.y = e[2] | (e[3] << 16); o.z = e[4] | (e[5] << 16); o.w = e[6] | (e[7] << 16);
    st_global16(y + yoff + (int64_t)(c0 + oc) * ld_out + r0 + r8 * 8, o);
  }
}

// y rows have stride ld_out >= rows; columns [rows, ld_out) are left untouched (callers pre-zero padded buffers)
extern "C" int xml_transpose_batched(const void* x, void* y, int batch, int rows, int cols, int ld_out, int dt,
                                     xml_stream_t stream) {
  XML_ENTER();
  if (!x || !y || batch <= 0 || rows <= 0 || cols <= 0 || ld_out < rows) return XML_ERR_BAD_ARG;
  if (dt == XML_BF16 && rows % 8 == 0 && cols % 8 == 0 && ld_out % 8 == 0 && rows >= 64 && cols >= 64) {
    hipLaunchKernelGGL(transpose64_bf16_kernel, dim3(cdiv(cols, 64), cdiv(rows, 64), batch), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, rows, cols, ld_out);
    XML_CHECK_LAUNCH();
    return XML_OK;
  }
  dim3 grid(cdiv(cols, 32), cdiv(rows, 32), batch);
  if (dt == XML_F32)
    hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, rows, cols, ld_out);
  else if (dt == XML_BF16)
    hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, rows, cols, ld_out);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// Transposed bf16 copies of many f32 matrices of one flat buffer in ONE launch (xml_transpose_segments): blockIdx.y = table
// entry {src_off, n, k, dst pointer, ld_dst, col0}, blockIdx.x = 64 x 64 tile of that (n, k) matrix (blocks beyond its tile
// count leave).  dst[kk * ld_dst + col0 + nn] = bf16(src[src_off + nn * k + kk]); 256-byte row reads, 128-byte row writes.
__global__ __launch_bounds__(256) void transpose_segments_kernel(const float* __restrict__ src,
                                                                 const int64_t* __restrict__ table) {
  __shared__ float tile[64][65];
  const int64_t* e = table + (int64_t)blockIdx.y * 6;
  const int64_t src_off = e[0];
  const int n = (int)e[1], k = (int)e[2];
  bf16_t* dst = reinterpret_cast<bf16_t*>(e[3]);
  const int64_t ld = e[4], col0 = e[5];
  const int tk = (k + 63) >> 6, tn = (n + 63) >> 6;
  if ((int)blockIdx.x >= tk * tn) return;
  const int n0 = ((int)blockIdx.x / tk) * 64, k0 = ((int)blockIdx.x % tk) * 64;
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int r = r4 + j * 4;
    tile[r][c] = (n0 + r < n && k0 + c < k) ? src[src_off + (int64_t)(n0 + r) * k + k0 + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int r = r4 + j * 4;                            // output row k0 + r, output column n0 + c
    if (k0 + r < k && n0 + c < n) DT<bf16_t>::st(dst + (int64_t)(k0 + r) * ld + col0 + n0 + c, tile[c][r]);
  }
}

extern "C" int xml_transpose_segments(const float* src, const int64_t* table, int n_ent, int max_tiles, int dt,
                                      xml_stream_t stream) {
  XML_ENTER();
  if (!src || !table || n_ent <= 0 || max_tiles <= 0) return XML_ERR_BAD_ARG;
  if (dt != XML_BF16) return XML_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(transpose_segments_kernel, dim3(max_tiles, n_ent), dim3(256), 0, (hipStream_t)stream, src, table);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// column sums: out[c] (+)= sum_r x[r][c]   (bias / positional-table gradients)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                     int cols, int64_t rows_per_block) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float s = 0.f;
  int64_t r = r0;
  for (; r + 8 <= r1; r += 8) {      // eight loads in flight, summed in row order (same result as the plain loop)
    T v[8];                            // raw loads; converted only after all eight are issued
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[(r + u) * cols + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += DT<T>::ld(&v[u]);
  }
  for (; r < r1; ++r) s += DT<T>::ld(x + r * cols + c);
  unsafeAtomicAdd(out + c, s);      // hardware f32 add throughout this file: plain atomicAdd(float*) is a CAS loop here, and many
                                    // blocks meeting on few addresses make it thrash (adam_norm: 104 -> 20 us)
}

extern "C" int xml_colsum(const void* x, int x_dt, float* out, int64_t rows, int cols, int accumulate,
                          xml_stream_t stream) {
  XML_ENTER();
  if (!x || !out || rows <= 0 || cols <= 0) return XML_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate && !xml_zero_async(out, (size_t)cols * 4, st)) return XML_ERR_LAUNCH;
  const int64_t rpb = rows >= 4096 ? 32 : 256;
  dim3 grid(cdiv(cols, 256), cdiv(rows, rpb));
  if (x_dt == XML_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)x, out, rows, cols, rpb);
  else if (x_dt == XML_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, out, rows, cols, rpb);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// elementwise: relu backward, in-place add, scaled copy to f32 accumulators
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    DT<T>::st(dx + i, DT<T>::ld(y + i) > 0.f ? DT<T>::ld(dy + i) : 0.f);
}
template <typename Y, typename X>
__global__ void add_inplace_kernel(Y* __restrict__ y, const X* __restrict__ x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    DT<Y>::st(y + i, DT<Y>::ld(y + i) + DT<X>::ld(x + i));
}
static inline int ew_grid(int64_t n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

extern "C" int xml_relu_bwd(const void* y, const void* dy, void* dx, int64_t n, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!y || !dy || !dx || n <= 0) return XML_ERR_BAD_ARG;
  if (dt == XML_F32)
    hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const float*)y, (const float*)dy, (float*)dx, n);
  else if (dt == XML_BF16)
    hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, (const bf16_t*)dy, (bf16_t*)dx, n);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// y (y_dt) += x (x_dt)
extern "C" int xml_add_inplace(void* y, int y_dt, const void* x, int x_dt, int64_t n, xml_stream_t stream) {
  XML_ENTER();
  if (!y || !x || n <= 0) return XML_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(ew_grid(n)), b(256);
  if (y_dt == XML_F32 && x_dt == XML_F32) hipLaunchKernelGGL((add_inplace_kernel<float, float>), g, b, 0, st, (float*)y, (const float*)x, n);
  else if (y_dt == XML_F32 && x_dt == XML_BF16) hipLaunchKernelGGL((add_inplace_kernel<float, bf16_t>), g, b, 0, st, (float*)y, (const bf16_t*)x, n);
  else if (y_dt == XML_BF16 && x_dt == XML_BF16) hipLaunchKernelGGL((add_inplace_kernel<bf16_t, bf16_t>), g, b, 0, st, (bf16_t*)y, (const bf16_t*)x, n);
  else if (y_dt == XML_BF16 && x_dt == XML_F32) hipLaunchKernelGGL((add_inplace_kernel<bf16_t, float>), g, b, 0, st, (bf16_t*)y, (const float*)x, n);
  else return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm backward.  x = a (+ b);  y = (x - mean) * rstd * g + beta
//   dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)),  dxhat = dy * g
//   dg += sum_rows dy * xhat,  dbeta += sum_rows dy          (f32, accumulated with one atomic per column per block)
// One wave per row (d <= 1024 values in registers); a block walks ROWS_PER_BLOCK rows and keeps column partials.
// ---------------------------------------------------------------------------------------------------------
// d % 8 == 0, d <= 1024: lane owns the 8-element vectors v = lane + 64*k (k < 2).  A block (4 waves) walks
// 4 * rows_per_wave rows, combines the column partials of its waves in LDS and issues one atomic per column.
// DROP (xml_layernorm_bwd_drop): the forward pass was y = drop_out( LN( drop_in(a) + b ) * g + beta ) (xml_add_layernorm_drop):
// dy is masked with the output site's mask on load, x is rebuilt from the masked a, and the gradient of a (dx times the
// input site's mask) goes to `dxa` next to dx (the gradient of b).
template <typename InT, typename BT, typename T, bool DROP = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const InT* __restrict__ a, const BT* __restrict__ b,
                                                            const float* __restrict__ g, const T* __restrict__ dy,
                                                            T* __restrict__ dx, float* __restrict__ dg,
                                                            float* __restrict__ dbeta, int64_t rows, int d,
                                                            float eps, int rows_per_wave,
                                                            XmlDropSite din = XmlDropSite{0u, 1.f, 0ull},
                                                            XmlDropSite dout = XmlDropSite{0u, 1.f, 0ull},
                                                            const uint64_t* __restrict__ seed_dev = nullptr,
                                                            T* __restrict__ dxa = nullptr) {
  __shared__ float s_pg[4][1024], s_pb[4][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = d >> 3;
  uint32_t si0 = 0, si1 = 0, so0 = 0, so1 = 0;
  if constexpr (DROP) {
    xml_seed_words(din.seed, seed_dev, si0, si1);
    xml_seed_words(dout.seed, seed_dev, so0, so1);
  }
  float pg[16], pb[16], gv[16];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int v = lane + k * 64;
#pragma unroll
    for (int j = 0; j < 8; ++j) { pg[k * 8 + j] = 0.f; pb[k * 8 + j] = 0.f; gv[k * 8 + j] = v < nvec ? g[v * 8 + j] : 0.f; }
  }
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave;
  // The raw 16-byte vectors of the NEXT row are fetched before the four dependent wave reductions of the current one
  // (a row is a memory round trip + ~1 us of reductions; issued load -> use per row, the kernel sat at 1.6 TB/s).
  // Loads are unconditional from clamped vector indices -- inside `if (v < nvec)` hipcc waits vmcnt(0) at the join.
  constexpr int VA = 16 / (int)sizeof(InT) == 8 ? 1 : 2, VB = 16 / (int)sizeof(BT) == 8 ? 1 : 2, VD = 16 / (int)sizeof(T) == 8 ? 1 : 2;
  uint4 ra[2][VA], rb[2][VB], rd[2][VD];
  auto fetch = [&](int64_t row) {
    const int64_t rc = row < rows ? row : rows - 1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int v = lane + k * 64;
      const int vc = v < nvec ? v : 0;
#pragma unroll
      for (int h = 0; h < VA; ++h) ra[k][h] = ld_global16(a + rc * d + vc * 8 + h * (8 / VA));
      if (b) {
#pragma unroll
        for (int h = 0; h < VB; ++h) rb[k][h] = ld_global16(b + rc * d + vc * 8 + h * (8 / VB));
      }
#pragma unroll
      for (int h = 0; h < VD; ++h) rd[k][h] = ld_global16(dy + rc * d + vc * 8 + h * (8 / VD));
    }
  };
  fetch(row0);
  for (int rr = 0; rr < rows_per_wave; ++rr) {
    const int64_t row = row0 + rr;
    if (row >= rows) break;
    float x[16], gy[16];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool on = lane + k * 64 < nvec;
#pragma unroll
      for (int h = 0; h < VA; ++h) unpack16<InT>(ra[k][h], x + k * 8 + h * (8 / VA));
      [[maybe_unused]] const uint64_t i0 = (uint64_t)row * (uint64_t)d + (uint64_t)(lane + k * 64) * 8u;
      if constexpr (DROP) {
        if (din.thresh) {
          drop_mask8(i0, si0, si1, din.thresh, din.scale, x + k * 8);
        }
      }
      if (b) {
        float t[8];
#pragma unroll
        for (int h = 0; h < VB; ++h) unpack16<BT>(rb[k][h], t + h * (8 / VB));
#pragma unroll
        for (int j = 0; j < 8; ++j) x[k * 8 + j] += t[j];
      }
#pragma unroll
      for (int h = 0; h < VD; ++h) unpack16<T>(rd[k][h], gy + k * 8 + h * (8 / VD));
      if constexpr (DROP) {
        if (dout.thresh) {
          drop_mask8(i0, so0, so1, dout.thresh, dout.scale, gy + k * 8);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x[k * 8 + j] = on ? x[k * 8 + j] : 0.f;
        gy[k * 8 + j] = on ? gy[k * 8 + j] : 0.f;
        s += x[k * 8 + j];
      }
    }
    if (rr + 1 < rows_per_wave) fetch(row + 1);         // (wave-uniform; clamped inside)
    const float mean = wave_sum(s) / (float)d;
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (lane + k * 64 < nvec)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float c = x[k * 8 + j] - mean; var += c * c; }
    const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)d + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {        // padding vectors hold x = gy = g = 0 -> xh = -mean*rstd but dy = 0
      const float xh = (x[i] - mean) * rstd;
      const float dyv = gy[i];
      x[i] = xh;
      gy[i] = dyv * gv[i];
      s1 += gy[i];
      s2 += gy[i] * xh;
      pg[i] += dyv * xh;
      pb[i] += dyv;
    }
    if (!dx) continue;                  // (block-uniform) parameter gradients only: the input needs no gradient
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int v = lane + k * 64;
      if (v < nvec) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rstd * (gy[k * 8 + j] - s1 - x[k * 8 + j] * s2);
        st8<T>(dx + row * d + v * 8, o);
        if constexpr (DROP) {
          if (dxa) {
            const uint64_t i0 = (uint64_t)row * (uint64_t)d + (uint64_t)v * 8u;
            drop_mask8(i0, si0, si1, din.thresh, din.scale, o);
            st8<T>(dxa + row * d + v * 8, o);
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int v = lane + k * 64;
    if (v < nvec)
#pragma unroll
      for (int j = 0; j < 8; ++j) { s_pg[wave][v * 8 + j] = pg[k * 8 + j]; s_pb[wave][v * 8 + j] = pb[k * 8 + j]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) {
    if (dg) unsafeAtomicAdd(dg + c, s_pg[0][c] + s_pg[1][c] + s_pg[2][c] + s_pg[3][c]);       // hardware f32 add, no CAS loop
    if (dbeta) unsafeAtomicAdd(dbeta + c, s_pb[0][c] + s_pb[1][c] + s_pb[2][c] + s_pb[3][c]);
  }
}

// wide rows (d > 1024, e.g. the 3072-d input LayerNorm): row statistics -> column sums -> elementwise dx
template <typename InT, typename T>
__global__ __launch_bounds__(256) void ln_bwd_stats_kernel(const InT* __restrict__ a, const float* __restrict__ g,
                                                           const T* __restrict__ dy, float* __restrict__ stats,
                                                           int64_t rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const InT* pa = a + row * d;
  float s = 0.f;
  for (int i = lane; i < d; i += 64) s += DT<InT>::ld(pa + i);
  const float mean = wave_sum(s) / (float)d;
  float v = 0.f;
  for (int i = lane; i < d; i += 64) { const float c = DT<InT>::ld(pa + i) - mean; v += c * c; }
  const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)d + eps);
  float s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < d; i += 64) {
    const float gy = DT<T>::ld(dy + row * d + i) * g[i];
    s1 += gy;
    s2 += gy * (DT<InT>::ld(pa + i) - mean) * rstd;
  }
  s1 = wave_sum(s1) / (float)d;
  s2 = wave_sum(s2) / (float)d;
  if (lane == 0) { stats[row * 4 + 0] = mean; stats[row * 4 + 1] = rstd; stats[row * 4 + 2] = s1; stats[row * 4 + 3] = s2; }
}
template <typename InT, typename T>
__global__ __launch_bounds__(256) void ln_bwd_cols_kernel(const InT* __restrict__ a, const float* __restrict__ g,
                                                          const T* __restrict__ dy, const float* __restrict__ stats,
                                                          T* __restrict__ dx, float* __restrict__ dg,
                                                          float* __restrict__ dbeta, int64_t rows, int d,
                                                          int64_t rows_per_block) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  const float gc = g[c];
  float pg = 0.f, pb = 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    const float mean = stats[r * 4], rstd = stats[r * 4 + 1];
    const float xh = (DT<InT>::ld(a + r * d + c) - mean) * rstd;
    const float dyv = DT<T>::ld(dy + r * d + c);
    pg += dyv * xh;
    pb += dyv;
    if (dx) DT<T>::st(dx + r * d + c, rstd * (dyv * gc - stats[r * 4 + 2] - xh * stats[r * 4 + 3]));
  }
  if (dg) unsafeAtomicAdd(dg + c, pg);
  if (dbeta) unsafeAtomicAdd(dbeta + c, pb);
}

// Wide rows, parameter gradients only (dx == NULL: the 3072-d input LayerNorm of the video features, whose input needs no
// gradient): ONE pass -- a wave keeps a row in registers (16-byte loads), takes mean / rstd, and accumulates dy * xhat and
// dy per column; the four waves of a block meet in LDS (ds_add_f32), one global atomic per column and block.  The
// two-kernel path above read the f32 row four times with scalar loads (190 us per step at 12 800 x 3072).
template <typename InT, typename T, int KV>
__global__ __launch_bounds__(256) void ln_bwd_wide_params_kernel(const InT* __restrict__ a, const T* __restrict__ dy,
                                                                 float* __restrict__ dg, float* __restrict__ dbeta,
                                                                 int64_t rows, int d, float eps, int rows_per_wave,
                                                                 XmlDropSite dout, const uint64_t* __restrict__ seed_dev) {
  extern __shared__ float s_acc[];               // [2][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = d >> 3;
  uint32_t so0 = 0, so1 = 0;
  if (dout.thresh) xml_seed_words(dout.seed, seed_dev, so0, so1);   // (output dropout site of xml_add_layernorm_drop)
  for (int i = threadIdx.x; i < 2 * d; i += 256) s_acc[i] = 0.f;
  __syncthreads();
  float pg[KV * 8], pb[KV * 8];
#pragma unroll
  for (int i = 0; i < KV * 8; ++i) { pg[i] = 0.f; pb[i] = 0.f; }
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave;
  for (int rr = 0; rr < rows_per_wave; ++rr) {
    const int64_t row = row0 + rr;
    if (row >= rows) break;
    float x[KV * 8], gy[KV * 8];
#pragma unroll
    for (int k = 0; k < KV; ++k) {               // unconditional loads from clamped vectors, zeroed by selects below
      const int v = lane + k * 64;
      const int vc = v < nvec ? v : 0;
      ld8<InT>(a + row * d + vc * 8, x + k * 8);
      ld8<T>(dy + row * d + vc * 8, gy + k * 8);
      if (dout.thresh) {
        const uint64_t i0 = (uint64_t)row * (uint64_t)d + (uint64_t)vc * 8u;
        drop_mask8(i0, so0, so1, dout.thresh, dout.scale, gy + k * 8);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const bool on = lane + k * 64 < nvec;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x[k * 8 + j] = on ? x[k * 8 + j] : 0.f;
        gy[k * 8 + j] = on ? gy[k * 8 + j] : 0.f;
        s += x[k * 8 + j];
      }
    }
    const float mean = wave_sum(s) / (float)d;
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < KV; ++k)
      if (lane + k * 64 < nvec)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float c = x[k * 8 + j] - mean; var += c * c; }
    const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < KV * 8; ++i) {           // (padding vectors: dy = 0)
      pg[i] += gy[i] * ((x[i] - mean) * rstd);
      pb[i] += gy[i];
    }
  }
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int v = lane + k * 64;
    if (v < nvec)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&s_acc[v * 8 + j], pg[k * 8 + j]);
        atomicAdd(&s_acc[d + v * 8 + j], pb[k * 8 + j]);
      }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) {
    if (dg) unsafeAtomicAdd(dg + c, s_acc[c]);
    if (dbeta) unsafeAtomicAdd(dbeta + c, s_acc[d + c]);
  }
}

// dg[c] += sum_b partial[b][c], dbeta[c] += sum_b partial[b][d + c]: the row chunks' column sums, written as rows of a scratch
// by ln_wide_cols_kernel, summed here by 8 slices of the rows in parallel (8 device-scope adds per address).
__global__ __launch_bounds__(256) void ln_partials_reduce_kernel(const float* __restrict__ partial, int n_blocks, int d,
                                                                 float* __restrict__ dg, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= 2 * d) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int64_t ld = 2 * (int64_t)d;
  int b = blockIdx.y;
  for (; b + 3 * (int)gridDim.y < n_blocks; b += 4 * gridDim.y) {
    s0 += partial[b * ld + c];
    s1 += partial[(b + (int64_t)gridDim.y) * ld + c];
    s2 += partial[(b + 2 * (int64_t)gridDim.y) * ld + c];
    s3 += partial[(b + 3 * (int64_t)gridDim.y) * ld + c];
  }
  for (; b < n_blocks; b += gridDim.y) s0 += partial[b * ld + c];
  const float v = (s0 + s1) + (s2 + s3);
  if (c < d) { if (dg) unsafeAtomicAdd(dg + c, v); }
  else if (dbeta) unsafeAtomicAdd(dbeta + c - d, v);
}

// ---- the same sums with a scratch: row statistics first, then a column pass ------------------------------------------------
// The one-pass kernel above keeps a row's x and dy AND its running column sums in registers (276 VGPRs at d = 3072: one wave
// per SIMD, every row a serial load -> two wave reductions -> accumulate chain; 12 800 x 3072 f32: 108 us alone, 190 us beside
// the other branch of the training step, 1.2 TB/s).  With rows * 8 bytes of scratch the row statistics are a streaming pass of
// their own (x only, ~70 VGPRs) and the column sums a second one in which a thread owns 8 columns and walks a chunk of rows
// with independent 32- / 16-byte loads; the chunks' sums go through the partials scratch like above.
template <typename InT, int KV>
__global__ __launch_bounds__(256) void ln_wide_stats_kernel(const InT* __restrict__ a, float2* __restrict__ stats, int64_t rows,
                                                            int d, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int nvec = d >> 3;
  float x[KV * 8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int v = lane + k * 64;
    ld8<InT>(a + row * d + (v < nvec ? v : 0) * 8, x + k * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x[k * 8 + j] = v < nvec ? x[k * 8 + j] : 0.f;
      s += x[k * 8 + j];
    }
  }
  const float mean = wave_sum(s) / (float)d;
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < KV; ++k)
    if (lane + k * 64 < nvec)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float c = x[k * 8 + j] - mean; var += c * c; }
  const float rstd = 1.0f / sqrtf(wave_sum(var) / (float)d + eps);      // (the arithmetic of the one-pass kernel)
  if (lane == 0) stats[row] = make_float2(mean, rstd);
}

template <typename InT, typename T>
__global__ __launch_bounds__(128) void ln_wide_cols_kernel(const InT* __restrict__ a, const T* __restrict__ dy,
                                                           const float2* __restrict__ stats, float* __restrict__ partial,
                                                           int64_t rows, int d, int rows_per_chunk, XmlDropSite dout,
                                                           const uint64_t* __restrict__ seed_dev) {
  const int v = blockIdx.x * 128 + threadIdx.x;      // this thread's 8 columns
  if (v >= (d >> 3)) return;
  uint32_t so0 = 0, so1 = 0;
  if (dout.thresh) xml_seed_words(dout.seed, seed_dev, so0, so1);
  float pg[8], pb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { pg[j] = 0.f; pb[j] = 0.f; }
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t r1 = min(rows, r0 + rows_per_chunk);
  auto one = [&](int64_t row) {
    float x[8], gy[8];
    const float2 st = stats[row];
    ld8<InT>(a + row * d + v * 8, x);
    ld8<T>(dy + row * d + v * 8, gy);
    if (dout.thresh) {
      const uint64_t i0 = (uint64_t)row * (uint64_t)d + (uint64_t)v * 8u;
      drop_mask8(i0, so0, so1, dout.thresh, dout.scale, gy);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pg[j] += gy[j] * ((x[j] - st.x) * st.y);
      pb[j] += gy[j];
    }
  };
  int64_t row = r0;
  for (; row + 4 <= r1; row += 4) { one(row); one(row + 1); one(row + 2); one(row + 3); }
  for (; row < r1; ++row) one(row);
  float* p = partial + (int64_t)blockIdx.y * (2 * d) + v * 8;
  *reinterpret_cast<float4*>(p) = make_float4(pg[0], pg[1], pg[2], pg[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(pg[4], pg[5], pg[6], pg[7]);
  *reinterpret_cast<float4*>(p + d) = make_float4(pb[0], pb[1], pb[2], pb[3]);
  *reinterpret_cast<float4*>(p + d + 4) = make_float4(pb[4], pb[5], pb[6], pb[7]);
}

static inline int ln_wide_rows_per_chunk(int64_t rows) { return rows >= 8192 ? 32 : 16; }
static inline size_t ln_wide_ws_bytes(int64_t rows, int d) {
  return align_up((size_t)rows * 8, 256) + (size_t)cdiv(rows, ln_wide_rows_per_chunk(rows)) * 2 * d * 4;
}

static inline int ln_wide_rows_per_wave(int64_t rows, bool) {
  return rows >= 8192 ? 8 : rows >= 4096 ? 4 : 1;      // every block ends in 2 d global atomics: fewer, longer blocks (12 800 x 3072: 198 -> 108 us)
}

template <typename InT, typename T>
static int ln_bwd_wide_params(const void* a, const void* dy, float* dg, float* dbeta, int64_t rows, int d, hipStream_t st,
                              XmlDropSite dout = XmlDropSite{0u, 1.f, 0ull}, const uint64_t* seed_dev = nullptr,
                              void* ws = nullptr, size_t ws_bytes = 0) {
  if (ws && rows >= 1024 && ws_bytes >= ln_wide_ws_bytes(rows, d) && ((uintptr_t)ws & 15) == 0) {
    float2* stats = (float2*)ws;
    float* partial = (float*)((char*)ws + align_up((size_t)rows * 8, 256));
    const int kv = cdiv(d >> 3, 64), rpc = ln_wide_rows_per_chunk(rows), chunks = cdiv(rows, rpc);
#define XML_LNS(KV)                                                                                                    \
  hipLaunchKernelGGL((ln_wide_stats_kernel<InT, KV>), dim3(cdiv(rows, 4)), dim3(256), 0, st, (const InT*)a, stats, rows, d, 1e-5f)
    if (kv <= 2) XML_LNS(2);
    else if (kv <= 4) XML_LNS(4);
    else if (kv <= 6) XML_LNS(6);
    else XML_LNS(8);
#undef XML_LNS
    hipLaunchKernelGGL((ln_wide_cols_kernel<InT, T>), dim3(cdiv(d >> 3, 128), chunks), dim3(128), 0, st, (const InT*)a,
                       (const T*)dy, stats, partial, rows, d, rpc, dout, seed_dev);
    hipLaunchKernelGGL(ln_partials_reduce_kernel, dim3(cdiv(2 * d, 256), 8), dim3(256), 0, st, partial, chunks, d, dg, dbeta);
    XML_CHECK_LAUNCH();
    return XML_OK;
  }
  const int rpw = ln_wide_rows_per_wave(rows, false);
  const dim3 grid(cdiv(rows, 4 * rpw)), blk(256);
  const size_t lds = (size_t)2 * d * 4;
  const int kv = cdiv(d >> 3, 64);
#define XML_LNW(KV)                                                                                                       \
  hipLaunchKernelGGL((ln_bwd_wide_params_kernel<InT, T, KV>), grid, blk, lds, st, (const InT*)a, (const T*)dy, dg, dbeta, rows, \
                     d, 1e-5f, rpw, dout, seed_dev)
  if (kv <= 2) XML_LNW(2);
  else if (kv <= 4) XML_LNW(4);
  else if (kv <= 6) XML_LNW(6);
  else XML_LNW(8);
#undef XML_LNW
  XML_CHECK_LAUNCH();
  return XML_OK;
}

template <typename InT, typename T>
static int ln_bwd_wide(const void* a, const float* g, const void* dy, void* dx, float* dg, float* dbeta, int64_t rows,
                       int d, float* stats, hipStream_t st) {
  hipLaunchKernelGGL((ln_bwd_stats_kernel<InT, T>), dim3(cdiv(rows, 4)), dim3(256), 0, st, (const InT*)a, g, (const T*)dy,
                     stats, rows, d, 1e-5f);
  const int64_t rpb = 64;
  hipLaunchKernelGGL((ln_bwd_cols_kernel<InT, T>), dim3(cdiv(d, 256), cdiv(rows, rpb)), dim3(256), 0, st, (const InT*)a,
                     g, (const T*)dy, stats, (T*)dx, dg, dbeta, rows, d, rpb);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// rows per wave of layernorm_bwd_kernel: a wave handles its rows one after the other, each a memory round trip plus four
// dependent wave reductions, so the launch must be wide -- 16 rows per wave left 200 workgroups for the step's 12 800 rows
// (58 us per call) -- but every workgroup ends in 2 d device-scope atomics on the same 2 d addresses: at 12 800 x 768 bf16,
// 4 rows per wave (800 workgroups) 36 us, of which 14 us atomics; 8 rows per wave 26.5 us; 3 840 rows: 15 us (4) vs 17 us (8).
static inline int ln_bwd_rows_per_wave(int64_t rows) { return rows >= 65536 ? 16 : rows >= 8192 ? 8 : rows >= 1024 ? 4 : 1; }

// dx may be NULL (input features need no gradient: dg / dbeta only); ws: rows*16 bytes, needed when d > 1024 and dx is wanted
extern "C" int xml_layernorm_bwd(const void* a, int a_dt, const void* b, const float* g, const void* dy, void* dx,
                                 float* dg, float* dbeta, int64_t rows, int d, int dt, void* ws, size_t ws_bytes,
                                 xml_stream_t stream) {
  XML_ENTER();
  if (!a || !g || !dy || rows <= 0 || d <= 0) return XML_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (d > 1024 || (d & 7)) {
    if (b) return XML_ERR_UNSUPPORTED;
    if (!dx && !(d & 7) && d <= 4096) {          // parameter gradients only: one pass (the f32 compute path keeps two)
      const XmlDropSite none = XmlDropSite{0u, 1.f, 0ull};
      if (dt == XML_BF16 && a_dt == XML_F32)
        return ln_bwd_wide_params<float, bf16_t>(a, dy, dg, dbeta, rows, d, st, none, nullptr, ws, ws_bytes);
      if (dt == XML_BF16 && a_dt == XML_BF16)
        return ln_bwd_wide_params<bf16_t, bf16_t>(a, dy, dg, dbeta, rows, d, st, none, nullptr, ws, ws_bytes);
    }
    if (!ws || ws_bytes < (size_t)rows * 16) return XML_ERR_WORKSPACE;
    if (dt == XML_F32 && a_dt == XML_F32) return ln_bwd_wide<float, float>(a, g, dy, dx, dg, dbeta, rows, d, (float*)ws, st);
    if (dt == XML_BF16 && a_dt == XML_F32) return ln_bwd_wide<float, bf16_t>(a, g, dy, dx, dg, dbeta, rows, d, (float*)ws, st);
    if (dt == XML_BF16 && a_dt == XML_BF16) return ln_bwd_wide<bf16_t, bf16_t>(a, g, dy, dx, dg, dbeta, rows, d, (float*)ws, st);
    return XML_ERR_BAD_ARG;
  }
  const int rpw = ln_bwd_rows_per_wave(rows);      // (dx NULL: dg / dbeta only -- raw input features need no gradient)
  const dim3 grid(cdiv(rows, 4 * rpw)), blk(256);
  if (dt == XML_F32) {
    if (a_dt != XML_F32) return XML_ERR_BAD_ARG;
    hipLaunchKernelGGL((layernorm_bwd_kernel<float, float, float>), grid, blk, 0, st, (const float*)a, (const float*)b, g,
                       (const float*)dy, (float*)dx, dg, dbeta, rows, d, 1e-5f, rpw);
  } else if (dt == XML_BF16) {
    if (a_dt == XML_F32)
      hipLaunchKernelGGL((layernorm_bwd_kernel<float, bf16_t, bf16_t>), grid, blk, 0, st, (const float*)a, (const bf16_t*)b,
                         g, (const bf16_t*)dy, (bf16_t*)dx, dg, dbeta, rows, d, 1e-5f, rpw);
    else
      hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, bf16_t, bf16_t>), grid, blk, 0, st, (const bf16_t*)a,
                         (const bf16_t*)b, g, (const bf16_t*)dy, (bf16_t*)dx, dg, dbeta, rows, d, 1e-5f, rpw);
  } else {
    return XML_ERR_BAD_ARG;
  }
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// Backward of xml_add_layernorm_drop.  dx: gradient of b (and of a when p_in == 0); dxa: gradient of a when p_in > 0
// (required then).  d <= 1024: everything; wider rows: parameter gradients only (dx == NULL), bf16, p_in == 0.
extern "C" size_t xml_layernorm_bwd_partials_bytes(int64_t rows, int d) {
  if (rows < 1024 || d <= 1024 || d > 4096 || (d & 7)) return 0;
  return ln_wide_ws_bytes(rows, d);
}

extern "C" int xml_layernorm_bwd_drop(const void* a, int a_dt, const void* b, const float* g, const void* dy, void* dx,
                                      void* dxa, float* dg, float* dbeta, int64_t rows, int d, int dt, float p_in,
                                      uint64_t seed_in, float p_out, uint64_t seed_out, const uint64_t* seed_dev,
                                      xml_stream_t stream) {
  return xml_layernorm_bwd_drop_ws(a, a_dt, b, g, dy, dx, dxa, dg, dbeta, rows, d, dt, p_in, seed_in, p_out, seed_out, seed_dev,
                                   nullptr, 0, stream);
}

extern "C" int xml_layernorm_bwd_drop_ws(const void* a, int a_dt, const void* b, const float* g, const void* dy, void* dx,
                                         void* dxa, float* dg, float* dbeta, int64_t rows, int d, int dt, float p_in,
                                         uint64_t seed_in, float p_out, uint64_t seed_out, const uint64_t* seed_dev, void* ws,
                                         size_t ws_bytes, xml_stream_t stream) {
  XML_ENTER();
  if (!a || !g || !dy || rows <= 0 || d <= 0) return XML_ERR_BAD_ARG;
  if (!(p_in >= 0.f) || p_in >= 1.f || !(p_out >= 0.f) || p_out >= 1.f) return XML_ERR_BAD_ARG;
  if (d & 7) return XML_ERR_UNSUPPORTED;
  const XmlDropSite din = xml_drop_site(p_in, seed_in), dout = xml_drop_site(p_out, seed_out);
  hipStream_t st = (hipStream_t)stream;
  if (d > 1024) {
    if (b || dx || dxa || din.thresh || d > 4096 || dt != XML_BF16) return XML_ERR_UNSUPPORTED;
    if (a_dt == XML_F32) return ln_bwd_wide_params<float, bf16_t>(a, dy, dg, dbeta, rows, d, st, dout, seed_dev, ws, ws_bytes);
    if (a_dt == XML_BF16) return ln_bwd_wide_params<bf16_t, bf16_t>(a, dy, dg, dbeta, rows, d, st, dout, seed_dev, ws, ws_bytes);
    return XML_ERR_BAD_ARG;
  }
  if (din.thresh && dx && !dxa) return XML_ERR_BAD_ARG;
  if (!din.thresh || !dx) dxa = nullptr;
  const int rpw = ln_bwd_rows_per_wave(rows);
  const dim3 grid(cdiv(rows, 4 * rpw)), blk(256);
  if (dt == XML_F32) {
    if (a_dt != XML_F32) return XML_ERR_BAD_ARG;
    hipLaunchKernelGGL((layernorm_bwd_kernel<float, float, float, true>), grid, blk, 0, st, (const float*)a, (const float*)b,
                       g, (const float*)dy, (float*)dx, dg, dbeta, rows, d, 1e-5f, rpw, din, dout, seed_dev, (float*)dxa);
  } else if (dt == XML_BF16) {
    if (a_dt == XML_F32)
      hipLaunchKernelGGL((layernorm_bwd_kernel<float, bf16_t, bf16_t, true>), grid, blk, 0, st, (const float*)a,
                         (const bf16_t*)b, g, (const bf16_t*)dy, (bf16_t*)dx, dg, dbeta, rows, d, 1e-5f, rpw, din, dout,
                         seed_dev, (bf16_t*)dxa);
    else
      hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, bf16_t, bf16_t, true>), grid, blk, 0, st, (const bf16_t*)a,
                         (const bf16_t*)b, g, (const bf16_t*)dy, (bf16_t*)dx, dg, dbeta, rows, d, 1e-5f, rpw, din, dout,
                         seed_dev, (bf16_t*)dxa);
  } else {
    return XML_ERR_BAD_ARG;
  }
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// contiguous batched GEMM: out[z] = scale * A[z] B[z]^T,  A (M,K), B (N,K) K-contiguous, out (M,N) as T or f32
// ---------------------------------------------------------------------------------------------------------
// KSPLIT > 1 (f32 output only, batch == 1): blockIdx.z walks K ranges of `kc` elements and the partial products are
// accumulated with f32 atomics into a pre-zeroed output -- the weight-gradient GEMMs (N x K_in outputs, reduction
// over all batch rows) would otherwise fill only a few dozen workgroups.
template <typename T, typename OutT, bool SPLIT>
__global__ __launch_bounds__(256) void gemm_batched_kernel(const T* __restrict__ A, const T* __restrict__ B,
                                                           OutT* __restrict__ out, int M, int N, int K, int kc,
                                                           float scale) {
  using Cfg = GemmCfg<T, 128, 128, 2, 2>;
  __shared__ __attribute__((aligned(16))) char smem[Cfg::LDS_BYTES];
  const int64_t z = SPLIT ? 0 : blockIdx.z;
  const int k0 = SPLIT ? blockIdx.z * kc : 0;
  const int klen = SPLIT ? (K - k0 < kc ? K - k0 : kc) : K;
  const T* Az = A + z * (int64_t)M * K + k0;
  const T* Bz = B + z * (int64_t)N * K + k0;
  OutT* Oz = out + z * (int64_t)M * N;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  f32x4 acc[Cfg::MT][Cfg::NT];
  auto a_row = [&](int r) -> const char* { return (m0 + r) < M ? reinterpret_cast<const char*>(Az + (int64_t)(m0 + r) * K) : nullptr; };
  auto b_row = [&](int r) -> const char* { return (n0 + r) < N ? reinterpret_cast<const char*>(Bz + (int64_t)(n0 + r) * K) : nullptr; };
  gemm_mainloop<T, Cfg>(acc, a_row, b_row, klen * (int)sizeof(T), smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
#pragma unroll
  for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < Cfg::NT; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + (lane & 15);
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + mt * 16 + (lane >> 4) * 4 + r;
        if (m >= M) continue;
        if constexpr (SPLIT) unsafeAtomicAdd(reinterpret_cast<float*>(Oz) + (int64_t)m * N + n, acc[mt][nt][r] * scale);
        else DT<OutT>::st(Oz + (int64_t)m * N + n, acc[mt][nt][r] * scale);
      }
    }
}

extern "C" int xml_gemm_batched(const void* A, const void* B, void* out, int batch, int M, int N, int K, float scale,
                                int out_f32, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!A || !B || !out || batch <= 0 || M <= 0 || N <= 0 || K <= 0) return XML_ERR_BAD_ARG;
  if (dt != XML_F32 && dt != XML_BF16) return XML_ERR_BAD_ARG;
  if (K % (dt == XML_F32 ? 4 : 8)) return XML_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int tiles = cdiv(N, 128) * cdiv(M, 128);
  const bool f32_out = (dt == XML_F32) || out_f32;
  if (batch == 1 && f32_out && K >= 1024 && tiles < 256) {
    int ks = cdiv(768, tiles);                       // aim at >= 3 workgroups per CU
    int kc = (cdiv(K, ks) + 63) / 64 * 64;           // K range per workgroup, 64-element granules
    if (kc < 256) kc = 256;
    ks = cdiv(K, kc);
    if (ks > 1) {
      if (!xml_zero_async(out, (size_t)M * N * 4, st)) return XML_ERR_LAUNCH;
      dim3 grid(cdiv(N, 128), cdiv(M, 128), ks);
      if (dt == XML_F32)
        hipLaunchKernelGGL((gemm_batched_kernel<float, float, true>), grid, dim3(256), 0, st, (const float*)A,
                           (const float*)B, (float*)out, M, N, K, kc, scale);
      else
        hipLaunchKernelGGL((gemm_batched_kernel<bf16_t, float, true>), grid, dim3(256), 0, st, (const bf16_t*)A,
                           (const bf16_t*)B, (float*)out, M, N, K, kc, scale);
      XML_CHECK_LAUNCH();
      return XML_OK;
    }
  }
  dim3 grid(cdiv(N, 128), cdiv(M, 128), batch);
  if (dt == XML_F32)
    hipLaunchKernelGGL((gemm_batched_kernel<float, float, false>), grid, dim3(256), 0, st, (const float*)A,
                       (const float*)B, (float*)out, M, N, K, 0, scale);
  else if (out_f32)
    hipLaunchKernelGGL((gemm_batched_kernel<bf16_t, float, false>), grid, dim3(256), 0, st, (const bf16_t*)A,
                       (const bf16_t*)B, (float*)out, M, N, K, 0, scale);
  else
    hipLaunchKernelGGL((gemm_batched_kernel<bf16_t, bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)A,
                       (const bf16_t*)B, (bf16_t*)out, M, N, K, 0, scale);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// head split / merge between the (n*L, ld) token-major layout and per-(sequence, head) contiguous matrices
//   dst  [n][h][l8][dh]   (rows >= L zero)      dstT [n][h][dh][l8]   (columns >= L zero)
// ---------------------------------------------------------------------------------------------------------
// block = (64-row tile of l, one (sequence, head)); the tile goes through LDS so that both the row-major copy and
// the transposed copy are written with 16-byte vectors
template <typename T>
__global__ __launch_bounds__(256) void split_heads_kernel(const T* __restrict__ src, int ld, int col0, int L, int l8,
                                                          int heads, int dh, T* __restrict__ dst, T* __restrict__ dstT) {
  constexpr int EV = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);            // [64][dh + EV]
  const int pitch = dh + EV;
  const int l0 = blockIdx.x * 64;
  const int64_t z = blockIdx.y;
  const int64_t n = z / heads;
  const int h = (int)(z % heads);
  const int nvec = dh / EV;
  for (int idx = threadIdx.x; idx < 64 * nvec; idx += 256) {
    const int row = idx / nvec, vc = idx % nvec;
    const int l = l0 + row;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (l < L) v = *reinterpret_cast<const uint4*>(src + (n * L + l) * ld + col0 + h * dh + vc * EV);
    *reinterpret_cast<uint4*>(tile + row * pitch + vc * EV) = v;
    if (dst && l < l8) *reinterpret_cast<uint4*>(dst + (z * l8 + l) * dh + vc * EV) = v;
  }
  if (!dstT) return;
  __syncthreads();
  constexpr int LV = 64 / EV;                          // l-vectors per tile row of the transposed copy
  for (int idx = threadIdx.x; idx < dh * LV; idx += 256) {
    const int d = idx / LV, lc = idx % LV;
    const int l = l0 + lc * EV;
    if (l >= l8) continue;
    T tmp[EV];
#pragma unroll
    for (int j = 0; j < EV; ++j) tmp[j] = tile[(lc * EV + j) * pitch + d];
    *reinterpret_cast<uint4*>(dstT + (z * dh + d) * l8 + l) = *reinterpret_cast<const uint4*>(tmp);
  }
}
template <typename T>
__global__ void merge_heads_kernel(const T* __restrict__ src, T* __restrict__ dst, int ld, int col0, int L, int l8,
                                   int heads, int dh, int64_t total_vec) {
  constexpr int EV = 16 / sizeof(T);
  const int nvec = dh / EV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * blockDim.x) {
    const int vc = (int)(i % nvec);
    const int h = (int)((i / nvec) % heads);
    const int l = (int)((i / ((int64_t)nvec * heads)) % L);
    const int64_t n = i / ((int64_t)nvec * heads * L);
    *reinterpret_cast<uint4*>(dst + (n * L + l) * ld + col0 + h * dh + vc * EV) =
        *reinterpret_cast<const uint4*>(src + ((n * heads + h) * l8 + l) * dh + vc * EV);
  }
}

extern "C" int xml_split_heads(const void* src, int ld, int col0, int64_t n, int L, int l8, int heads, int dh, void* dst,
                               void* dstT, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!src || (!dst && !dstT) || n <= 0 || L <= 0 || l8 < L || (l8 & 7)) return XML_ERR_BAD_ARG;
  if ((dh & 7) || (ld & 7) || (col0 & 7) || dh > 256) return XML_ERR_UNSUPPORTED;     // 16-byte vectors, <= 64 KB LDS
  dim3 grid(cdiv(l8, 64), (unsigned)(n * heads));
  if (dt == XML_F32) {
    const size_t lds = (size_t)64 * (dh + 4) * 4;
    if (lds > 65536) return XML_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(split_heads_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, (const float*)src, ld, col0, L, l8, heads, dh, (float*)dst, (float*)dstT);
  } else if (dt == XML_BF16) {
    const size_t lds = (size_t)64 * (dh + 8) * 2;
    hipLaunchKernelGGL(split_heads_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)src, ld, col0, L, l8, heads, dh, (bf16_t*)dst, (bf16_t*)dstT);
  } else {
    return XML_ERR_BAD_ARG;
  }
  XML_CHECK_LAUNCH();
  return XML_OK;
}

extern "C" int xml_merge_heads(const void* src, void* dst, int ld, int col0, int64_t n, int L, int l8, int heads, int dh,
                               int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!src || !dst || n <= 0 || L <= 0 || l8 < L) return XML_ERR_BAD_ARG;
  if ((dh & 7) || (ld & 7) || (col0 & 7)) return XML_ERR_UNSUPPORTED;
  if (dt == XML_F32) {
    const int64_t total = n * L * heads * (dh / 4);
    hipLaunchKernelGGL(merge_heads_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)src, (float*)dst, ld, col0, L, l8, heads, dh, total);
  } else if (dt == XML_BF16) {
    const int64_t total = n * L * heads * (dh / 8);
    hipLaunchKernelGGL(merge_heads_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, ld, col0, L, l8, heads, dh, total);
  } else {
    return XML_ERR_BAD_ARG;
  }
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// attention softmax (recompute) and its backward on materialised per-head score matrices
//   S (f32) [n*h][lq8][lk8] raw QK^T;   P = softmax_j( S / sqrt_dh + (1 - qm*km) * -1e4 )  over j < lk
//   fwd writes P and P^T (T, zero padded);  bwd: dS = P * (dP - sum_j P*dP) / sqrt_dh  -> dS, dS^T (T)
// one wave per (batch, query row); lk <= 128
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_softmax_kernel(const float* __restrict__ S, const float* __restrict__ dP,
                                                           const float* __restrict__ q_mask,
                                                           const float* __restrict__ k_mask, T* __restrict__ P,
                                                           T* __restrict__ PT, T* __restrict__ dS, T* __restrict__ dST,
                                                           int heads, int lq, int lk, int lq8, int lk8, float sqrt_dh,
                                                           int64_t total_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const int i = (int)(row % lq8);
  const int64_t z = row / lq8;               // (sequence, head)
  const int64_t n = z / heads;
  const float* srow = S + (z * lq8 + i) * lk8;
  const float qm = (q_mask && i < lq) ? q_mask[n * lq + i] : 1.f;
  float p[2], mx = -INFINITY;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int j = lane + h * 64;
    float s = -INFINITY;
    if (i < lq && j < lk) s = srow[j] / sqrt_dh + (1.f - qm * k_mask[n * lk + j]) * -10000.f;
    p[h] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    p[h] = (i < lq) ? expf(p[h] - mx) : 0.f;
    sum += p[h];
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int h = 0; h < 2; ++h) p[h] = (i < lq) ? p[h] / sum : 0.f;
  if (!dP) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = lane + h * 64;
      if (j < lk8) {
        DT<T>::st(P + (z * lq8 + i) * lk8 + j, p[h]);
        if (PT) DT<T>::st(PT + (z * lk8 + j) * lq8 + i, p[h]);
      }
    }
    return;
  }
  const float* drow = dP + (z * lq8 + i) * lk8;
  float dp[2], dot = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int j = lane + h * 64;
    dp[h] = (i < lq && j < lk) ? drow[j] : 0.f;
    dot += p[h] * dp[h];
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int j = lane + h * 64;
    if (j < lk8) {
      const float v = p[h] * (dp[h] - dot) / sqrt_dh;
      DT<T>::st(dS + (z * lq8 + i) * lk8 + j, v);
      DT<T>::st(dST + (z * lk8 + j) * lq8 + i, v);
    }
  }
}

// dP == NULL: forward (writes P, optionally P^T);  dP != NULL: backward (writes dS and dS^T)
extern "C" int xml_attn_softmax(const float* S, const float* dP, const float* q_mask, const float* k_mask, void* P,
                                void* PT, void* dS, void* dST, int64_t n, int heads, int lq, int lk, int lq8, int lk8,
                                float sqrt_dh, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!S || !k_mask || n <= 0 || lq <= 0 || lk <= 0 || lk > 128 || lk8 > 128) return XML_ERR_BAD_ARG;
  if ((!dP && !P) || (dP && (!dS || !dST))) return XML_ERR_BAD_ARG;
  const int64_t rows = n * heads * lq8;
  if (dt == XML_F32)
    hipLaunchKernelGGL(attn_softmax_kernel<float>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, S, dP, q_mask,
                       k_mask, (float*)P, (float*)PT, (float*)dS, (float*)dST, heads, lq, lk, lq8, lk8, sqrt_dh, rows);
  else if (dt == XML_BF16)
    hipLaunchKernelGGL(attn_softmax_kernel<bf16_t>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, S, dP, q_mask,
                       k_mask, (bf16_t*)P, (bf16_t*)PT, (bf16_t*)dS, (bf16_t*)dST, heads, lq, lk, lq8, lk8, sqrt_dh, rows);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// modular query pooling backward (forward: attention.hip modular_pool_kernel; xml/model_xml.py:410-423)
//   a = softmax_l(mask_logits(enc w_m)),  mq[m] = sum_l a[l][m] enc[l]
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void modular_pool_bwd_kernel(const T* __restrict__ enc, const float* __restrict__ mask,
                                                               const float* __restrict__ wm, const T* __restrict__ dout,
                                                               T* __restrict__ denc, float* __restrict__ dwm, int64_t n,
                                                               int lq, int hidden, int n_mod) {
  __shared__ float s_att[2][128], s_da[2][128], s_dsc[2][128];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* e = enc + (int64_t)q * lq * hidden;
  for (int l = wave; l < lq; l += 4)
    for (int m = 0; m < n_mod; ++m) {
      const T* dm = dout + ((int64_t)m * n + q) * hidden;
      float s = 0.f, d = 0.f;
      for (int h = lane; h < hidden; h += 64) {
        const float ev = DT<T>::ld(e + (int64_t)l * hidden + h);
        s += ev * wm[m * hidden + h];
        d += ev * DT<T>::ld(dm + h);
      }
      s = wave_sum(s);
      d = wave_sum(d);
      if (lane == 0) {
        const float mk = mask[(int64_t)q * lq + l];
        s_att[m][l] = s * mk + (1.f - mk) * -1e10f;
        s_da[m][l] = d;
      }
    }
  __syncthreads();
  if (tid < n_mod) {
    float mx = -INFINITY;
    for (int l = 0; l < lq; ++l) mx = fmaxf(mx, s_att[tid][l]);
    float sum = 0.f;
    for (int l = 0; l < lq; ++l) { const float ev = expf(s_att[tid][l] - mx); s_att[tid][l] = ev; sum += ev; }
    float dot = 0.f;
    for (int l = 0; l < lq; ++l) { s_att[tid][l] /= sum; dot += s_att[tid][l] * s_da[tid][l]; }
    for (int l = 0; l < lq; ++l)
      s_dsc[tid][l] = s_att[tid][l] * (s_da[tid][l] - dot) * mask[(int64_t)q * lq + l];
  }
  __syncthreads();
  for (int h = tid; h < hidden; h += 256) {
    float dw[2] = {0.f, 0.f}, dm[2] = {0.f, 0.f}, w[2] = {0.f, 0.f};
    for (int m = 0; m < n_mod; ++m) {
      dm[m] = DT<T>::ld(dout + ((int64_t)m * n + q) * hidden + h);
      w[m] = wm[m * hidden + h];
    }
    for (int l = 0; l < lq; ++l) {
      const float ev = DT<T>::ld(e + (int64_t)l * hidden + h);
      float g = 0.f;
      for (int m = 0; m < n_mod; ++m) {
        g += s_att[m][l] * dm[m] + s_dsc[m][l] * w[m];
        dw[m] += s_dsc[m][l] * ev;
      }
      DT<T>::st(denc + ((int64_t)q * lq + l) * hidden + h, g);
    }
    for (int m = 0; m < n_mod; ++m) unsafeAtomicAdd(dwm + m * hidden + h, dw[m]);
  }
}

extern "C" int xml_modular_pool_bwd(const void* enc, const float* mask, const float* w_m, const void* dout, void* denc,
                                    float* dw_m, int64_t n, int lq, int hidden, int n_mod, int dt,
                                    xml_stream_t stream) {
  XML_ENTER();
  if (!enc || !mask || !w_m || !dout || !denc || !dw_m || n <= 0 || lq <= 0 || hidden <= 0) return XML_ERR_BAD_ARG;
  if (n_mod < 1 || n_mod > 2 || lq > 128) return XML_ERR_UNSUPPORTED;
  if (xmli_modular_pool_bwd_vec(enc, mask, w_m, dout, denc, dw_m, n, lq, hidden, n_mod, dt, (hipStream_t)stream) == 0) {
    XML_CHECK_LAUNCH();                  // loss_tail.hip: hidden % 8 == 0, <= 2048
    return XML_OK;
  }
  if (dt == XML_F32)
    hipLaunchKernelGGL(modular_pool_bwd_kernel<float>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const float*)enc, mask, w_m, (const float*)dout, (float*)denc, dw_m, n, lq, hidden, n_mod);
  else if (dt == XML_BF16)
    hipLaunchKernelGGL(modular_pool_bwd_kernel<bf16_t>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)enc, mask, w_m, (const bf16_t*)dout, (bf16_t*)denc, dw_m, n, lq, hidden, n_mod);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// F.normalize backward: y = x / max(|x|, eps);  dx = (dy - y (y . dy)) / max(|x|, eps)    dy is f32
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dy,
                                                         T* __restrict__ dx, int64_t rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float ss = 0.f, xd = 0.f;
  for (int i = lane; i < d; i += 64) {
    const float v = DT<T>::ld(x + row * d + i);
    ss += v * v;
    xd += v * dy[row * d + i];
  }
  ss = wave_sum(ss);
  xd = wave_sum(xd);
  const float nrm = sqrtf(ss);
  const float den = fmaxf(nrm, eps);
  for (int i = lane; i < d; i += 64) {
    const float v = DT<T>::ld(x + row * d + i);
    // norm clamped at eps is a constant: only the (dy / den) term survives there
    const float g = nrm > eps ? (dy[row * d + i] - v * xd / (den * den)) / den : dy[row * d + i] / den;
    DT<T>::st(dx + row * d + i, g);
  }
}

extern "C" int xml_l2norm_bwd(const void* x, const float* dy, void* dx, int64_t rows, int d, int dt,
                              xml_stream_t stream) {
  XML_ENTER();
  if (!x || !dy || !dx || rows <= 0 || d <= 0) return XML_ERR_BAD_ARG;
  if (dt == XML_F32)
    hipLaunchKernelGGL(l2norm_bwd_kernel<float>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const float*)x, dy, (float*)dx, rows, d, 1e-12f);
  else if (dt == XML_BF16)
    hipLaunchKernelGGL(l2norm_bwd_kernel<bf16_t>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, dy, (bf16_t*)dx, rows, d, 1e-12f);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// in-batch video-level scores backward (forward: xml_q2c_scores; xml/model_xml.py:436-453)
//   scores[m][n] = max_l mask_logits(qn[m] . cn[n][l]);  the gradient goes to the arg-max clip (first on ties)
//   dqn / dcn are f32 and are ZEROED here; pairs with dscores == 0 are skipped (the ranking loss touches <= 3N pairs)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void q2c_scores_bwd_kernel(const T* __restrict__ qn, const T* __restrict__ cn,
                                                             const float* __restrict__ mask,
                                                             const float* __restrict__ dscores, int ld_ds,
                                                             float scale, float* __restrict__ dqn,
                                                             float* __restrict__ dcn, int nq, int nv, int L,
                                                             int hidden) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.x;
  if (m >= nq) return;
  const float g = dscores[(int64_t)m * ld_ds + n] * scale;
  if (g == 0.f) return;
  const T* q = qn + (int64_t)m * hidden;
  float best = -INFINITY;
  int best_l = 0x7fffffff;
  for (int l = lane; l < L; l += 64) {
    const T* c = cn + ((int64_t)n * L + l) * hidden;
    float s = 0.f;
    for (int h = 0; h < hidden; ++h) s += DT<T>::ld(q + h) * DT<T>::ld(c + h);
    const float mk = mask[(int64_t)n * L + l];
    s = s * mk + (1.f - mk) * -1e10f;
    if (s > best) { best = s; best_l = l; }
  }
  for (int off = 32; off; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const int ol = __shfl_xor(best_l, off);
    if (ob > best || (ob == best && ol < best_l)) { best = ob; best_l = ol; }
  }
  const float gm = g * mask[(int64_t)n * L + best_l];
  if (gm == 0.f) return;
  const T* c = cn + ((int64_t)n * L + best_l) * hidden;
  for (int h = lane; h < hidden; h += 64) {
    unsafeAtomicAdd(dqn + (int64_t)m * hidden + h, gm * DT<T>::ld(c + h));
    unsafeAtomicAdd(dcn + ((int64_t)n * L + best_l) * hidden + h, gm * DT<T>::ld(q + h));
  }
}

// The same with 16-byte loads: one WAVE per (query, video) pair that carries a gradient; the lanes split the hidden
// dimension in 8-element chunks (a clip row is one coalesced read) and every clip costs one wave reduction, four clips in
// flight.  The scalar kernel above put one clip per lane -- 64 rows walked element by element, 2 bytes per load -- and took
// 217 us for the 3 N pairs of a 128-video batch; this one takes a few.  hidden % 8 == 0.
template <typename T>
__global__ __launch_bounds__(256) void q2c_scores_bwd_vec_kernel(const T* __restrict__ qn, const T* __restrict__ cn,
                                                                 const float* __restrict__ mask,
                                                                 const float* __restrict__ dscores, int ld_ds, float scale,
                                                                 float* __restrict__ dqn, float* __restrict__ dcn, int nq,
                                                                 int nv, int L, int hidden) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.x;
  if (m >= nq) return;
  const float g = dscores[(int64_t)m * ld_ds + n] * scale;
  if (g == 0.f) return;
  const T* q = qn + (int64_t)m * hidden;
  const T* cbase = cn + (int64_t)n * L * hidden;
  const int chunks = hidden >> 3;
  float best = -INFINITY;
  int best_l = 0;
  for (int l0 = 0; l0 < L; l0 += 4) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < chunks; c += 64) {
      float qv[8];
      ld8<T>(q + c * 8, qv);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int l = min(l0 + u, L - 1);
        float cv[8];
        ld8<T>(cbase + (int64_t)l * hidden + c * 8, cv);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[u] += qv[e] * cv[e];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int l = l0 + u;
      float v = wave_sum(s[u]);
      if (l < L) {
        const float mk = mask[(int64_t)n * L + l];
        v = v * mk + (1.f - mk) * -1e10f;
        if (v > best) { best = v; best_l = l; }          // first clip on ties
      }
    }
  }
  const float gm = g * mask[(int64_t)n * L + best_l];
  if (gm == 0.f) return;
  const T* c = cbase + (int64_t)best_l * hidden;
  for (int ch = lane; ch < chunks; ch += 64) {
    float qv[8], cv[8];
    ld8<T>(q + ch * 8, qv);
    ld8<T>(c + ch * 8, cv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsafeAtomicAdd(dqn + (int64_t)m * hidden + ch * 8 + e, gm * cv[e]);
      unsafeAtomicAdd(dcn + ((int64_t)n * L + best_l) * hidden + ch * 8 + e, gm * qv[e]);
    }
  }
}

extern "C" int xml_q2c_scores_bwd(const void* qn, const void* cn, const float* mask, const float* dscores, int64_t ld_ds,
                                  float scale, float* dqn, float* dcn, int nq, int nv, int l, int hidden, int dt,
                                  xml_stream_t stream) {
  XML_ENTER();
  if (!qn || !cn || !mask || !dscores || !dqn || !dcn || nq <= 0 || nv <= 0 || l <= 0 || hidden <= 0) return XML_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (!xml_zero_async(dqn, (size_t)nq * hidden * 4, st)) return XML_ERR_LAUNCH;
  if (!xml_zero_async(dcn, (size_t)nv * l * hidden * 4, st)) return XML_ERR_LAUNCH;
  dim3 grid(nv, cdiv(nq, 4));
  if (hidden % 8 == 0 && (dt == XML_F32 || dt == XML_BF16)) {
    if (dt == XML_F32)
      hipLaunchKernelGGL(q2c_scores_bwd_vec_kernel<float>, grid, dim3(256), 0, st, (const float*)qn, (const float*)cn, mask, dscores, (int)ld_ds, scale, dqn, dcn, nq, nv, l, hidden);
    else
      hipLaunchKernelGGL(q2c_scores_bwd_vec_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)qn, (const bf16_t*)cn, mask, dscores, (int)ld_ds, scale, dqn, dcn, nq, nv, l, hidden);
    XML_CHECK_LAUNCH();
    return XML_OK;
  }
  if (dt == XML_F32)
    hipLaunchKernelGGL(q2c_scores_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)qn, (const float*)cn, mask, dscores, (int)ld_ds, scale, dqn, dcn, nq, nv, l, hidden);
  else if (dt == XML_BF16)
    hipLaunchKernelGGL(q2c_scores_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)qn, (const bf16_t*)cn, mask, dscores, (int)ld_ds, scale, dqn, dcn, nq, nv, l, hidden);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// paired query-clip similarity (cross=False branch, einsum("bd,bld->bl"), xml/model_xml.py:478-479,532)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pair_sim_kernel(const T* __restrict__ q, const T* __restrict__ f2,
                                                       float* __restrict__ sim, int64_t n, int L, int hidden) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (b, l)
  if (row >= n * L) return;
  const int64_t b = row / L;
  float s = 0.f;
  for (int h = lane; h < hidden; h += 64) s += DT<T>::ld(q + b * hidden + h) * DT<T>::ld(f2 + row * hidden + h);
  s = wave_sum(s);
  if (lane == 0) sim[row] = s;
}
template <typename T>
__global__ __launch_bounds__(256) void pair_sim_bwd_kernel(const T* __restrict__ q, const T* __restrict__ f2,
                                                           const float* __restrict__ dsim, T* __restrict__ dq,
                                                           T* __restrict__ df2, int L, int hidden) {
  const int64_t b = blockIdx.x;
  for (int h = threadIdx.x; h < hidden; h += 256) {
    const float qv = DT<T>::ld(q + b * hidden + h);
    float acc = 0.f;
    for (int l = 0; l < L; ++l) {
      const float g = dsim[b * L + l];
      acc += g * DT<T>::ld(f2 + (b * L + l) * hidden + h);
      DT<T>::st(df2 + (b * L + l) * hidden + h, g * qv);
    }
    DT<T>::st(dq + b * hidden + h, acc);
  }
}

// 16-byte version: one workgroup per batch row, its four waves take the clips l = w, w + 4, ...; lanes hold 8-element
// chunks of the hidden dimension; four clips in flight per wave; dq is combined across the waves in LDS.  (The kernel above
// walks the L clips one after the other per thread: 122 us per call at L = 100, 128 workgroups.)  hidden % 8 == 0, <= 2048.
template <typename T>
__global__ __launch_bounds__(256) void pair_sim_bwd_vec_kernel(const T* __restrict__ q, const T* __restrict__ f2,
                                                               const float* __restrict__ dsim, T* __restrict__ dq,
                                                               T* __restrict__ df2, int L, int hidden) {
  __shared__ float s_acc[4][2048];
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunks = hidden >> 3;
  for (int c0 = 0; c0 < chunks; c0 += 64) {
    const int c = c0 + lane;
    const bool ok = c < chunks;
    const int cc = ok ? c : 0;
    float qv[8], acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    ld8<T>(q + b * hidden + cc * 8, qv);
    for (int l0 = wave; l0 < L; l0 += 16) {
      float fv[4][8], g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int l = min(l0 + 4 * u, L - 1);
        g[u] = (l0 + 4 * u < L) ? dsim[b * L + l] : 0.f;
        ld8<T>(f2 + (b * L + l) * hidden + cc * 8, fv[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int l = l0 + 4 * u;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc[e] += g[u] * fv[u][e]; o[e] = g[u] * qv[e]; }
        if (ok && l < L) st8<T>(df2 + (b * L + l) * hidden + c * 8, o);
      }
    }
    if (ok)
#pragma unroll
      for (int e = 0; e < 8; ++e) s_acc[wave][c * 8 + e] = acc[e];
  }
  __syncthreads();
  for (int h = threadIdx.x; h < hidden; h += 256)
    DT<T>::st(dq + b * hidden + h, (s_acc[0][h] + s_acc[1][h]) + (s_acc[2][h] + s_acc[3][h]));
}

extern "C" int xml_pair_sim(const void* q, const void* f2, float* sim, int64_t n, int l, int hidden, int dt,
                            xml_stream_t stream) {
  XML_ENTER();
  if (!q || !f2 || !sim || n <= 0 || l <= 0 || hidden <= 0) return XML_ERR_BAD_ARG;
  if (dt == XML_F32)
    hipLaunchKernelGGL(pair_sim_kernel<float>, dim3(cdiv(n * l, 4)), dim3(256), 0, (hipStream_t)stream, (const float*)q, (const float*)f2, sim, n, l, hidden);
  else if (dt == XML_BF16)
    hipLaunchKernelGGL(pair_sim_kernel<bf16_t>, dim3(cdiv(n * l, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)f2, sim, n, l, hidden);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}
extern "C" int xml_pair_sim_bwd(const void* q, const void* f2, const float* dsim, void* dq, void* df2, int64_t n, int l,
                                int hidden, int dt, xml_stream_t stream) {
  XML_ENTER();
  if (!q || !f2 || !dsim || !dq || !df2 || n <= 0 || l <= 0 || hidden <= 0) return XML_ERR_BAD_ARG;
  if (hidden % 8 == 0 && hidden <= 2048 && (dt == XML_F32 || dt == XML_BF16)) {
    if (dt == XML_F32)
      hipLaunchKernelGGL(pair_sim_bwd_vec_kernel<float>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const float*)q, (const float*)f2, dsim, (float*)dq, (float*)df2, l, hidden);
    else
      hipLaunchKernelGGL(pair_sim_bwd_vec_kernel<bf16_t>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)f2, dsim, (bf16_t*)dq, (bf16_t*)df2, l, hidden);
    XML_CHECK_LAUNCH();
    return XML_OK;
  }
  if (dt == XML_F32)
    hipLaunchKernelGGL(pair_sim_bwd_kernel<float>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const float*)q, (const float*)f2, dsim, (float*)dq, (float*)df2, l, hidden);
  else if (dt == XML_BF16)
    hipLaunchKernelGGL(pair_sim_bwd_kernel<bf16_t>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)f2, dsim, (bf16_t*)dq, (bf16_t*)df2, l, hidden);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// span (start / end) loss head, forward and backward in one kernel (xml/model_xml.py:237-240,478-500,532-550):
//   merged : s = (sim0 + sim1) / 2 ; st = mask_logits(conv(s, w_st), mask0)
//   split  : st = sum_i mask_logits(conv(sim_i, w_st_i), mask_i) / n_sim
//   loss   = mean_b [ CE(st_b, idx_st_b) + CE(ed_b, idx_ed_b) ]
// conv_w = [st filters (n_filt x ks) | ed filters (n_filt x ks)], n_filt = merged ? 1 : n_sim.
// gout == NULL: writes loss_out (pre-zeroed by the entry).  gout != NULL: writes dsim_i and accumulates dconv_w,
// all scaled by *gout.  One wave per batch row, L <= 128.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void span_loss_kernel(const float* __restrict__ sim0, const float* __restrict__ sim1,
                                                        const float* __restrict__ conv_w,
                                                        const float* __restrict__ mask0, const float* __restrict__ mask1,
                                                        const int64_t* __restrict__ st_ed, int merged, int n_sim, int ks,
                                                        int n, int L, const float* __restrict__ gout,
                                                        float* __restrict__ loss_out, float* __restrict__ dsim0,
                                                        float* __restrict__ dsim1, float* __restrict__ dconv_w) {
  __shared__ float s_sim[4][2][160];    // per wave, per stream, zero halo of 16 on both sides
  __shared__ float s_dc[4][2][160];     // d(conv output) per filter
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= n) return;
  const int pad = ks / 2;
  const int n_filt = merged ? 1 : n_sim;
  float (*S)[160] = s_sim[wave];
  float (*DC)[160] = s_dc[wave];
  for (int i = lane; i < 160; i += 64) { S[0][i] = 0.f; S[1][i] = 0.f; }
  for (int l = lane; l < L; l += 64) {
    const float a = sim0[(int64_t)b * L + l];
    const float c = n_sim > 1 ? sim1[(int64_t)b * L + l] : 0.f;
    if (merged) S[0][16 + l] = (a + c) * 0.5f;
    else { S[0][16 + l] = a; S[1][16 + l] = c; }
  }
  __builtin_amdgcn_wave_barrier();
  float total = 0.f;
  for (int se = 0; se < 2; ++se) {              // 0 = start head, 1 = end head
    const float* w = conv_w + se * n_filt * ks;
    const int target = (int)st_ed[(int64_t)b * 2 + se];
    float logit[2], mx = -INFINITY;
    for (int h = 0; h < 2; ++h) {
      const int l = lane + h * 64;
      float v = -INFINITY;
      if (l < L) {
        v = 0.f;
        for (int f = 0; f < n_filt; ++f) {
          float c = 0.f;
          for (int t = 0; t < ks; ++t) c += w[f * ks + t] * S[f][16 + l + t - pad];
          const float mk = (f == 0 ? mask0 : mask1)[(int64_t)b * L + l];
          v += c * mk + (1.f - mk) * -1e10f;
        }
        if (!merged) v /= (float)n_sim;
      }
      logit[h] = v;
      mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int h = 0; h < 2; ++h) sum += (lane + h * 64 < L) ? expf(logit[h] - mx) : 0.f;
    sum = wave_sum(sum);
    const float lse = mx + logf(sum);
    float tl = 0.f;
    for (int h = 0; h < 2; ++h) if (lane + h * 64 == target) tl = logit[h];
    tl = wave_sum(tl);
    total += lse - tl;
    if (gout) {
      const float go = gout[0] / (float)n;
      for (int i = lane; i < 160; i += 64) { DC[0][i] = 0.f; DC[1][i] = 0.f; }
      __builtin_amdgcn_wave_barrier();
      for (int h = 0; h < 2; ++h) {
        const int l = lane + h * 64;
        if (l < L) {
          float dl = (expf(logit[h] - lse) - (l == target ? 1.f : 0.f)) * go;
          if (!merged) dl /= (float)n_sim;
          for (int f = 0; f < n_filt; ++f) DC[f][16 + l] = dl * (f == 0 ? mask0 : mask1)[(int64_t)b * L + l];
        }
      }
      __builtin_amdgcn_wave_barrier();
      for (int f = 0; f < n_filt; ++f) {
        for (int t = 0; t < ks; ++t) {       // dw[t] = sum_l dc[l] * s[l + t - pad]
          float a = 0.f;
          for (int l = lane; l < L; l += 64) a += DC[f][16 + l] * S[f][16 + l + t - pad];
          a = wave_sum(a);
          if (lane == 0) unsafeAtomicAdd(dconv_w + (se * n_filt + f) * ks + t, a);
        }
        for (int l = lane; l < L; l += 64) {  // ds[l] = sum_t w[t] * dc[l - t + pad]
          float a = 0.f;
          for (int t = 0; t < ks; ++t) a += w[f * ks + t] * DC[f][16 + l - t + pad];
          if (merged) {
            unsafeAtomicAdd(dsim0 + (int64_t)b * L + l, 0.5f * a);
            if (n_sim > 1) unsafeAtomicAdd(dsim1 + (int64_t)b * L + l, 0.5f * a);
          } else {
            unsafeAtomicAdd((f == 0 ? dsim0 : dsim1) + (int64_t)b * L + l, a);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (!gout && lane == 0) unsafeAtomicAdd(loss_out, total / (float)n);
}

extern "C" int xml_span_loss(const float* sim0, const float* sim1, const float* conv_w, const float* mask0,
                             const float* mask1, const int64_t* st_ed, int merged, int n_sim, int ks, int n, int l,
                             const float* gout, float* loss_out, float* dsim0, float* dsim1, float* dconv_w,
                             xml_stream_t stream) {
  XML_ENTER();
  if (!sim0 || !conv_w || !mask0 || !st_ed || n <= 0 || l <= 0) return XML_ERR_BAD_ARG;
  if (n_sim < 1 || n_sim > 2 || (n_sim == 2 && (!sim1 || (!merged && !mask1))) || (merged && n_sim != 2)) return XML_ERR_BAD_ARG;
  if (l > 128 || ks < 1 || ks > 31 || !(ks & 1)) return XML_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (!gout) {
    if (!loss_out) return XML_ERR_BAD_ARG;
    if (!xml_zero_async(loss_out, 4, st)) return XML_ERR_LAUNCH;
  } else {
    if (!dsim0 || !dconv_w || (n_sim == 2 && !dsim1)) return XML_ERR_BAD_ARG;
    const int n_filt = merged ? 1 : n_sim;
    const size_t sb = (size_t)n * l * 4, cb = (size_t)2 * n_filt * ks * 4;
    if (n_sim == 2 && (char*)dsim1 == (char*)dsim0 + sb && (char*)dconv_w == (char*)dsim1 + sb) {
      if (!xml_zero_async(dsim0, 2 * sb + cb, st)) return XML_ERR_LAUNCH;      // caller laid the three out back to back: one fill
    } else {
      if (!xml_zero_async(dsim0, sb, st)) return XML_ERR_LAUNCH;
      if (n_sim == 2 && !xml_zero_async(dsim1, sb, st)) return XML_ERR_LAUNCH;
      if (!xml_zero_async(dconv_w, cb, st)) return XML_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL(span_loss_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, sim0, sim1, conv_w, mask0, mask1 ? mask1 : mask0,
                     st_ed, merged, n_sim, ks, n, l, gout, loss_out, dsim0, dsim1, dconv_w);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// in-batch ranking loss, forward and backward (get_video_level_loss / get_neg_scores / get_ranking_loss,
// xml/model_xml.py:588-637).  For query i the negative context is the entry of row i (diagonal masked to 999,
// sorted descending) at position ranks_ctx[i]; the negative query likewise on column i with ranks_q[i].
//   losses[0] = mean_i rank(pos_i, neg_ctx_i)   losses[1] = mean_i rank(pos_i, neg_q_i)
//   hinge: max(0, margin + neg - pos);  lse: log1p(exp(neg - pos))
// grid (n, 2); gout == NULL: losses (pre-zeroed);  gout[2] != NULL: dscores (pre-zeroed) += gout[side] * d loss
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rank_loss_kernel(const float* __restrict__ scores, const int* __restrict__ ranks_ctx,
                                                        const int* __restrict__ ranks_q, float margin, int lse, int n,
                                                        const float* __restrict__ gout, float* __restrict__ losses,
                                                        float* __restrict__ dscores) {
  __shared__ float row[1024];
  const int i = blockIdx.x, side = blockIdx.y;
  auto at = [&](int a, int b2) { return side == 0 ? (int64_t)a * n + b2 : (int64_t)b2 * n + a; };
  for (int j = threadIdx.x; j < n; j += 256) row[j] = (j == i) ? 999.f : scores[at(i, j)];
  __syncthreads();
  const int want = (side == 0 ? ranks_ctx : ranks_q)[i];
  for (int j = threadIdx.x; j < n; j += 256) {
    const float v = row[j];
    int cnt = 0;
    for (int k = 0; k < n; ++k) cnt += (row[k] > v || (row[k] == v && k < j)) ? 1 : 0;
    if (cnt != want) continue;
    const float pos = scores[(int64_t)i * n + i];
    const float neg = scores[at(i, j)];
    const float x = neg - pos;
    if (!gout) {
      const float lv = lse ? log1pf(expf(x)) : fmaxf(margin + x, 0.f);
      unsafeAtomicAdd(losses + side, lv / (float)n);
    } else {
      const float d = lse ? 1.f / (1.f + expf(-x)) : (margin + x > 0.f ? 1.f : 0.f);
      const float g = gout[side] * d / (float)n;
      if (g != 0.f) {
        unsafeAtomicAdd(dscores + at(i, j), g);
        unsafeAtomicAdd(dscores + (int64_t)i * n + i, -g);
      }
    }
  }
}

extern "C" int xml_rank_loss(const float* scores, const int* ranks_ctx, const int* ranks_q, float margin, int lse, int n,
                             const float* gout, float* losses, float* dscores, xml_stream_t stream) {
  XML_ENTER();
  if (!scores || !ranks_ctx || !ranks_q || n <= 0) return XML_ERR_BAD_ARG;
  if (n > 1024) return XML_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (!gout) {
    if (!losses) return XML_ERR_BAD_ARG;
    if (!xml_zero_async(losses, 8, st)) return XML_ERR_LAUNCH;
  } else {
    if (!dscores) return XML_ERR_BAD_ARG;
    if (!xml_zero_async(dscores, (size_t)n * n * 4, st)) return XML_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(rank_loss_kernel, dim3(n, 2), dim3(256), 0, st, scores, ranks_ctx, ranks_q, margin, lse, n, gout,
                     losses, dscores);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// BertAdam over one flat f32 parameter buffer (xml/optimization.py:273-338): per-TENSOR gradient-norm clip
// (clip_grad_norm_(p, max_grad_norm), coefficient max_norm / (norm + 1e-6) clamped to 1, applied to the gradient
// in place), no bias correction, decoupled weight decay, lr = seg_lr[s] * lr_mult (schedule multiplier).
//   seg_off (n_seg + 1) int64: tensor s owns [seg_off[s], seg_off[s+1]).  norms: n_seg floats of scratch.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_seg(const int64_t* __restrict__ seg_off, int n_seg, int64_t i) {
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (seg_off[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}
// one block = 4096 consecutive elements; when they all belong to one tensor (the common case) the block reduces
// in registers / LDS and issues a single atomic
__global__ __launch_bounds__(256) void adam_norm_kernel(const float* __restrict__ g, const int64_t* __restrict__ seg_off,
                                                        int n_seg, int64_t total, float* __restrict__ norms) {
  __shared__ float s_part[4];
  const int64_t base = (int64_t)blockIdx.x * 4096;
  const int64_t last = base + 4095 < total ? base + 4095 : total - 1;
  const int s_lo = find_seg(seg_off, n_seg, base);
  const bool uniform = last < seg_off[s_lo + 1];
  if (uniform) {
    float acc = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int64_t i = base + it * 1024 + threadIdx.x * 4;
      if (i + 3 < total) {
        const float4 v = *reinterpret_cast<const float4*>(g + i);
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      } else {
        for (int k = 0; k < 4; ++k) if (i + k < total) acc += g[i + k] * g[i + k];
      }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(norms + s_lo, s_part[0] + s_part[1] + s_part[2] + s_part[3]);      // (hardware f32 add: thousands of blocks meet on ~100 addresses, a CAS loop thrashes)
    return;
  }
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < 16; ++it) {       // tensor boundary inside the block: segmented reduction per wave
    const int64_t i = base + it * 256 + threadIdx.x;
    const bool valid = i < total;
    const int seg = valid ? find_seg(seg_off, n_seg, i) : -1;
    const float val = valid ? g[i] * g[i] : 0.f;
    unsigned long long todo = __ballot(valid);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int s_cur = __shfl(seg, leader);
      const bool mine = valid && seg == s_cur;
      const float part = wave_sum(mine ? val : 0.f);
      if (lane == leader) unsafeAtomicAdd(norms + s_cur, part);
      todo &= ~__ballot(mine);
    }
  }
}
// The same without the atomics of the common case (norm_ws given): a block that lies inside one tensor stores its partial sum
// (loads issued before the tensor search); adam_norm_reduce_kernel -- one wave per tensor -- adds them up in a fixed order.
// 4 930 device-scope atomics on ~100 addresses (576 of them on the largest tensor) made the kernel above 92 us for 80 MB.
template <int NB>      // elements per block (a multiple of 1024)
__global__ __launch_bounds__(256) void adam_norm_partial_kernel(const float* __restrict__ g, const int64_t* __restrict__ seg_off,
                                                                int n_seg, int64_t total, float* __restrict__ norms,
                                                                float* __restrict__ partials) {
  __shared__ float s_part[4];
  const int64_t base = (int64_t)blockIdx.x * NB;
  const int64_t last = base + NB - 1 < total ? base + NB - 1 : total - 1;
  constexpr int NV = NB / 1024;
  float4 v[NV];
  const bool vec_ok = (total & 3) == 0;                  // (tensors are padded to 4 elements: always, see BertAdam._flatten)
#pragma unroll
  for (int it = 0; it < NV; ++it) {          // unconditional loads from clamped addresses (a load inside an `if` is waited
    const int64_t i = base + it * 1024 + threadIdx.x * 4;      // for at the join: the NV loads would run one after the other)
    const bool ok = vec_ok && i + 3 < total;
    const float4 t = *reinterpret_cast<const float4*>(g + (ok ? i : 0));
    v[it] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
  }
  const int s_lo = find_seg(seg_off, n_seg, base);
  const bool uniform = last < seg_off[s_lo + 1] && vec_ok;
  if (uniform) {
    float acc = 0.f;
#pragma unroll
    for (int it = 0; it < NV; ++it) acc += v[it].x * v[it].x + v[it].y * v[it].y + v[it].z * v[it].z + v[it].w * v[it].w;
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    return;
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = 0.f;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < NB / 256; ++it) {       // tensor boundary inside the block: segmented reduction per wave, atomics
    const int64_t i = base + it * 256 + threadIdx.x;
    const bool valid = i < total;
    const int seg = valid ? find_seg(seg_off, n_seg, i) : -1;
    const float val = valid ? g[i] * g[i] : 0.f;
    unsigned long long todo = __ballot(valid);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int s_cur = __shfl(seg, leader);
      const bool mine = valid && seg == s_cur;
      const float part = wave_sum(mine ? val : 0.f);
      if (lane == leader) unsafeAtomicAdd(norms + s_cur, part);
      todo &= ~__ballot(mine);
    }
  }
}
__global__ __launch_bounds__(64) void adam_norm_reduce_kernel(int nb, const int64_t* __restrict__ seg_off, int n_seg, int64_t total,
                                                              const float* __restrict__ partials, float* __restrict__ norms) {
  const int s = blockIdx.x;
  const bool vec_ok = (total & 3) == 0;
  const int64_t b0 = (seg_off[s] + nb - 1) / nb;
  const int64_t b1 = s == n_seg - 1 ? (total + nb - 1) / nb : seg_off[s + 1] / nb;
  float acc = 0.f;
  if (vec_ok)
    for (int64_t b = b0 + threadIdx.x; b < b1; b += 64) acc += partials[b];
  acc = wave_sum(acc);
  if (threadIdx.x == 0 && acc != 0.f) unsafeAtomicAdd(norms + s, acc);     // (after the boundary blocks' atomics: same stream)
}
// one element: the reference's update (xml/optimization.py:289-338) for element i of tensor s
__device__ __forceinline__ void adam_update_one(float& pi, float& gi, float& mi, float& vi, bool& g_changed, float clip_coef,
                                                float lr, float wd, float b1, float b2, float eps) {
  if (clip_coef < 1.f) { gi *= clip_coef; g_changed = true; }
  mi = mi * b1 + (1.f - b1) * gi;
  vi = vi * b2 + (1.f - b2) * gi * gi;
  float upd = mi / (sqrtf(vi) + eps);
  if (wd > 0.f) upd += wd * pi;
  pi -= lr * upd;
}

// A block = 1024 consecutive elements, a thread = 4 of them (16-byte loads / stores).  When the block lies inside one
// tensor -- all but ~n_seg of the blocks -- the tensor lookup (a binary search over the offsets: dependent loads) and its
// per-tensor constants are block-uniform, i.e. scalar loads, done once; the former kernel searched per ELEMENT and moved
// 4 bytes per load (203 us for 20 M parameters).
__global__ __launch_bounds__(256) void adam_update_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, const int64_t* __restrict__ seg_off,
                                                          const float* __restrict__ seg_lr,
                                                          const float* __restrict__ seg_wd, int n_seg, int64_t total,
                                                          const float* __restrict__ norms, float lr_mult, float b1,
                                                          float b2, float eps, float max_grad_norm,
                                                          const uint8_t* __restrict__ seg_active,
                                                          const float* __restrict__ seg_lr_mult) {
  const int64_t base = (int64_t)blockIdx.x * 1024;
  const int64_t last = base + 1023 < total ? base + 1023 : total - 1;
  const int s_lo = find_seg(seg_off, n_seg, base);
  const bool uniform = last < seg_off[s_lo + 1] && last - base == 1023 && (total & 3) == 0;
  const int64_t i0 = base + (int64_t)threadIdx.x * 4;
  if (uniform) {
    // `if p.grad is None: continue` (xml/optimization.py:289-291): a tensor that has never received a gradient is not
    // touched at all -- no moment update, no weight decay, no schedule step
    if (seg_active && !seg_active[s_lo]) return;
    const float lm = seg_lr_mult ? seg_lr_mult[s_lo] : lr_mult;          // per-tensor state['step'] (xml/optimization.py:325-330)
    const float coef = max_grad_norm > 0.f ? max_grad_norm / (sqrtf(norms[s_lo]) + 1e-6f) : 1.f;
    const float lr = seg_lr[s_lo] * lm, wd = seg_wd[s_lo];
    float4 pv = *reinterpret_cast<float4*>(p + i0), gv = *reinterpret_cast<float4*>(g + i0);
    float4 mv = *reinterpret_cast<float4*>(m + i0), vv = *reinterpret_cast<float4*>(v + i0);
    bool gc = false;
    adam_update_one(pv.x, gv.x, mv.x, vv.x, gc, coef, lr, wd, b1, b2, eps);
    adam_update_one(pv.y, gv.y, mv.y, vv.y, gc, coef, lr, wd, b1, b2, eps);
    adam_update_one(pv.z, gv.z, mv.z, vv.z, gc, coef, lr, wd, b1, b2, eps);
    adam_update_one(pv.w, gv.w, mv.w, vv.w, gc, coef, lr, wd, b1, b2, eps);
    if (gc) *reinterpret_cast<float4*>(g + i0) = gv;
    *reinterpret_cast<float4*>(m + i0) = mv;
    *reinterpret_cast<float4*>(v + i0) = vv;
    *reinterpret_cast<float4*>(p + i0) = pv;
    return;
  }
  for (int k = 0; k < 4; ++k) {       // a tensor boundary (or the tail) inside the block: per element
    const int64_t i = i0 + k;
    if (i >= total) return;
    const int s = find_seg(seg_off, n_seg, i);
    if (seg_active && !seg_active[s]) continue;
    const float lm = seg_lr_mult ? seg_lr_mult[s] : lr_mult;
    const float coef = max_grad_norm > 0.f ? max_grad_norm / (sqrtf(norms[s]) + 1e-6f) : 1.f;
    float pi = p[i], gi = g[i], mi = m[i], vi = v[i];
    bool gc = false;
    adam_update_one(pi, gi, mi, vi, gc, coef, seg_lr[s] * lm, seg_wd[s], b1, b2, eps);
    if (gc) g[i] = gi;
    m[i] = mi; v[i] = vi; p[i] = pi;
  }
}

extern "C" int xml_bert_adam_step(float* p, float* g, float* m, float* v, const int64_t* seg_off, const float* seg_lr,
                                  const float* seg_wd, int n_seg, int64_t total, float lr_mult, float b1, float b2,
                                  float eps, float max_grad_norm, float* norms, const uint8_t* seg_active,
                                  const float* seg_lr_mult, float* norm_ws, xml_stream_t stream) {
  XML_ENTER();
  if (!p || !g || !m || !v || !seg_off || !seg_lr || !seg_wd || !norms || n_seg <= 0 || total <= 0) return XML_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (max_grad_norm > 0.f) {
    if (!xml_zero_async(norms, (size_t)n_seg * 4, st)) return XML_ERR_LAUNCH;
    if (norm_ws) {
      // 1024 elements per block: measured 25 us (19.7 K blocks) vs 28 / 44 / 59 us at 2048 / 4096 / 8192 -- the block's
      // latency (tensor search + one round of loads + reduction) grows faster than its size
      hipLaunchKernelGGL(adam_norm_partial_kernel<XML_ADAM_NORM_BLOCK>, dim3(cdiv(total, XML_ADAM_NORM_BLOCK)), dim3(256), 0, st,
                         g, seg_off, n_seg, total, norms, norm_ws);
      hipLaunchKernelGGL(adam_norm_reduce_kernel, dim3(n_seg), dim3(64), 0, st, XML_ADAM_NORM_BLOCK, seg_off, n_seg, total,
                         norm_ws, norms);
    } else {
      hipLaunchKernelGGL(adam_norm_kernel, dim3(cdiv(total, 4096)), dim3(256), 0, st, g, seg_off, n_seg, total, norms);
    }
  }
  hipLaunchKernelGGL(adam_update_kernel, dim3(cdiv(total, 1024)), dim3(256), 0, st, p, g, m, v, seg_off, seg_lr, seg_wd,
                     n_seg, total, norms, lr_mult, b1, b2, eps, max_grad_norm, seg_active, seg_lr_mult);
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// nn.Dropout (training mode): y = keep(i) ? x / (1 - p) : 0 with a counter-based mask keep(i) = hash(seed, i) >= p 2^32.
// The mask is a pure function of (seed, element index), so the backward pass re-applies the same call to the
// gradient instead of storing a mask.  In place (y == x) is allowed.  The random stream is NOT torch's Philox
// stream: dropout is statistically, not bitwise, equivalent to the reference's.
// ---------------------------------------------------------------------------------------------------------
// (drop_hash: common.h -- shared with the LayerNorm kernels that apply the same masks in their loads / stores)
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, uint32_t thresh, float scale,
                               uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  uint32_t s0, s1;
  xml_seed_words(seed, seed_dev, s0, s1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    DT<T>::st(y + i, drop_hash((uint64_t)i, s0, s1) >= thresh ? DT<T>::ld(x + i) * scale : 0.f);
}

extern "C" int xml_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, int dt,
                           xml_stream_t stream) {
  XML_ENTER();
  if (!x || !y || n <= 0 || !(p >= 0.f) || p >= 1.f) return XML_ERR_BAD_ARG;
  const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
  const float scale = 1.f / (1.f - p);
  if (dt == XML_F32)
    hipLaunchKernelGGL(dropout_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, n, thresh, scale, seed, seed_dev);
  else if (dt == XML_BF16)
    hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n, thresh, scale, seed, seed_dev);
  else
    return XML_ERR_BAD_ARG;
  XML_CHECK_LAUNCH();
  return XML_OK;
}

// ---------------------------------------------------------------------------------------------------------
// torch.nn.utils.clip_grad_norm_ over ALL gradients (xml/train.py:88-90, `grad_clip`, off by default): one flat f32
// buffer, total L2 norm, g *= max_norm / (norm + 1e-6) when that coefficient is < 1.  ws: one float of scratch.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out) {
  __shared__ float s_part[4];
  float acc = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (int k = 0; k < 4 && i + k < n; ++k) acc += g[i + k] * g[i + k];
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}
__global__ void clip_scale_kernel(float* __restrict__ g, int64_t n, const float* __restrict__ sumsq, float max_norm) {
  const float coef = max_norm / (sqrtf(sumsq[0]) + 1e-6f);
  if (coef >= 1.f) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) g[i] *= coef;
}

extern "C" int xml_clip_grad_norm(float* g, int64_t n, float max_norm, float* ws, xml_stream_t stream) {
  XML_ENTER();
  if (!g || !ws || n <= 0 || !(max_norm > 0.f)) return XML_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (!xml_zero_async(ws, 4, st)) return XML_ERR_LAUNCH;
  const int blocks = (int)((n + 1023) / 1024 < 2048 ? (n + 1023) / 1024 : 2048);
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, st, g, n, ws);
  hipLaunchKernelGGL(clip_scale_kernel, dim3(ew_grid(n)), dim3(256), 0, st, g, n, ws, max_norm);
  XML_CHECK_LAUNCH();
  return XML_OK;
}
