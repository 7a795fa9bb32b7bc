// Cross-file internal entry points of libxmlhip (not part of the C ABI).
#pragma once
#include "common.h"

// out = act(A W^T + bias) + addend ; add_mode 0 none / 1 positional table row (m % seq_len) / 2 residual row m
// dt == XML_F16S (split-f16 projection: f32 activations, weights from xml_pack_weights_f16s): split_ws must hold
// xmli_gemm_split_ws_bytes(M, K, dt) bytes; every other kernel of such a call runs on act_dt(dt) = f32 storage
int xmli_gemm(const void* A, const void* W, const float* bias, const void* addend, void* out, int64_t M, int N,
              int K, int relu, int add_mode, int seq_len, int out_f32, int dt, hipStream_t st, void* split_ws = nullptr);
size_t xmli_gemm_split_ws_bytes(int64_t M, int K, int dt);
static inline int xmli_act_dt(int dt) { return dt == XML_F16S ? XML_F32 : dt; }
static inline bool xmli_model_dt_ok(int dt) { return dt == XML_F32 || dt == XML_BF16 || dt == XML_F16S; }
// y = LN(act(A W^T + bias) + addend) * g + b with the LayerNorm in the GEMM epilogue (gemm256p.hip); callers check
// xmli_gemm_ln_eligible and pass xmli_gemm_ln_workspace_bytes(M, N) bytes of scratch
bool xmli_gemm_ln_eligible(int64_t M, int N, int K, int dt);
size_t xmli_gemm_ln_workspace_bytes(int64_t M, int N);
int xmli_gemm_ln(const void* A, const void* W, const float* bias, const void* addend, const float* ln_g, const float* ln_b,
                 void* y, int64_t M, int N, int K, int relu, int add_mode, int seq_len, int dt, void* ln_ws,
                 hipStream_t st);
// y = LN(a + b) ; a may be f32 while b / y are dt ; rows of y have stride ld_out (>= d, tail zero-filled)
int xmli_add_layernorm(const void* a, int a_dt, const void* b, const float* g, const float* beta, void* y,
                       int64_t rows, int d, int ld_out, int dt, hipStream_t st);
// multi-head attention core on projected Q / K / V (see attention.hip)
int xmli_attention_core(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                        const float* q_mask, const float* k_mask, void* out, int out_f32, int64_t n, int lq, int lk,
                        int hidden, int n_heads, int dt, hipStream_t st);
// loss_tail.hip: 16-byte modular-pooling backward; 0 = launched, -1 = shape not served (caller keeps the scalar kernel)
int xmli_modular_pool_bwd_vec(const void* enc, const float* mask, const float* w_m, const void* dout, void* denc, float* dw_m,
                              int64_t n, int lq, int hidden, int n_mod, int dt, hipStream_t st);
