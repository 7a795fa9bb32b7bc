/*
 * xmlhip.h -- C ABI of libxmlhip.so: the MI355X (gfx950) hot path of the XML corpus-level
 * moment-retrieval model (jayleicn/TVRetrieval, baselines/crossmodal_moment_localization = "xml/").
 *
 * The reference has no FFI: its boundary is the Python method surface of XML(nn.Module)
 * (SURVEY.md 8b).  Each entry point below replaces one stock-PyTorch op sequence of that surface and
 * cites it.  The host-side mirror (tvretrieval_amd/model_xml.py) binds these through ctypes; a
 * maintainer of the reference would bind them the same way.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named h_*; tensors are dense row-major, batch first;
 *   - `dt` selects the storage type of activations and weights: XML_F32 (exact-f32 MFMA, parity
 *     config) or XML_BF16 (bf16 storage, f32 accumulation and f32 LayerNorm/softmax statistics);
 *     LayerNorm affine parameters, biases, masks, scores and probabilities are always f32;
 *   - masks are float32 1=valid / 0=pad exactly as the reference passes them (start_end_dataset.py:346-370);
 *   - all calls are asynchronous on `stream`, never allocate, never synchronise and are re-entrant per
 *     stream; scratch comes from the caller: `ws`/`ws_bytes`, sized by the matching *_workspace_bytes();
 *   - return value: 0 = XML_OK, negative = xml_status (never throws, never aborts).
 */
#ifndef XMLHIP_H
#define XMLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* xml_stream_t; /* hipStream_t */

/* XML_F32 / XML_BF16: the two compute dtypes of the model.  XML_F16 / XML_F16S exist for the exact-rank mode only (see
 * "Exact-rank mode on the 16-bit pipe"): XML_F16 = IEEE half rows (the similarity FILTER operands), XML_F16S = "split f16",
 * f32-grade values carried as hi + lo halves (4 bytes per element). */
typedef enum { XML_F32 = 0, XML_BF16 = 1, XML_F16 = 2, XML_F16S = 3 } xml_dtype;
/* log2 of the FIXED scale of unit-norm rows in XML_F16 / XML_F16S form (xml_split_f16_rows(fixed_log2 = this)): the
 * similarity kernels (xml_q2c_scores_tiled / _packed with dt = XML_F16, xml_q2c_rescore with dt = XML_F16S) undo 2^-28. */
#define XML_F16_UNIT_LOG2 14

typedef enum {
  XML_OK = 0,
  XML_ERR_BAD_ARG = -1,       /* null pointer, negative size, unsupported dtype            */
  XML_ERR_UNSUPPORTED = -2,   /* shape outside what the kernels implement (see each entry) */
  XML_ERR_WORKSPACE = -3,     /* ws_bytes smaller than *_workspace_bytes()                 */
  XML_ERR_LAUNCH = -4         /* hipGetLastError() != hipSuccess after a launch            */
} xml_status;

/* version / introspection --------------------------------------------------------------------- */
int xml_abi_version(void);                 /* bumps on any signature change */
const char* xml_build_arch(void);          /* "gfx950" */
const char* xml_status_string(int status);

/* ---------------------------------------------------------------------------------------------
 * Weight packing.  load_state_dict-time conversion of f32 checkpoint tensors (xml/train.py:219-223
 * layout) into the device dtype; `n` elements, round-to-nearest-even for bf16.
 * --------------------------------------------------------------------------------------------- */
int xml_pack_weights(const float* src, void* dst, int dt, int64_t n, xml_stream_t stream);
/* generic dtype conversion of activations (either direction) */
int xml_convert(const void* src, int src_dt, void* dst, int dst_dt, int64_t n, xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K1+K2: LinearLayer + TrainablePositionalEncoding
 *   y = LN_pos( ReLU( LN_in(x) W^T + b ) + E[l] )      l = row % seq_len
 * replaces LinearLayer.forward (xml/model_components.py:156-163) followed by
 * TrainablePositionalEncoding.forward (:76-89) as called from XML.encode_input (xml/model_xml.py:387-390).
 *   x      (rows, d_in)  raw features, f32 or dt (x_dt)
 *   w      (hidden, d_pad) dt, d_pad = d_in rounded up to a multiple of 8, columns >= d_in ZERO (pack time);
 *          b (hidden) f32;  ln_in_{g,b} (d_in) f32
 *   pos    (>= seq_len, hidden) dt;  ln_pos_{g,b} (hidden) f32
 *   y      (rows, hidden) dt
 * Requirements: hidden % 8 == 0.  Any d_in: the TEF context modes (visual 3074 / sub 770, xml/config.py:251-254,
 * xml/start_end_dataset.py:127-142) take the LayerNorm statistics over d_in and run the GEMM over d_pad.
 * --------------------------------------------------------------------------------------------- */
size_t xml_linear_ln_relu_pos_workspace_bytes(int64_t rows, int d_in, int hidden, int dt);
int xml_linear_ln_relu_pos(const void* x, int x_dt, const float* ln_in_g, const float* ln_in_b,
                           const void* w, const float* b, const void* pos, const float* ln_pos_g,
                           const float* ln_pos_b, void* y, int64_t rows, int seq_len, int d_in,
                           int hidden, int dt, void* ws, size_t ws_bytes, xml_stream_t stream);

/* The large projections (K1 + K2, the output dense of K3 + K4; hidden = 256 / 512 / 768, >= 2 048 rows) run as ONE kernel
 * with the LayerNorm in the GEMM's epilogue: one workgroup computes all column tiles of a 256-row block and normalises it
 * itself -- nothing is exchanged between workgroups, nothing waits, there is no failure path and no process-wide switch.
 * Whether a projection takes that kernel depends on its shape class, never on how full the chip is or how many videos a
 * batch holds: the encoder's results are bitwise independent of the context batch size (for batches of >= 16 videos). */

/* ---------------------------------------------------------------------------------------------
 * K3+K4: BertAttention = BertSelfAttention + BertSelfOutput (no FFN)
 *   y = LN( dense( MHA(x, x, x, key_mask) ) + x )
 * replaces BertAttention.forward (xml/model_components.py:207-216, :266-303, :313-317).
 * Additive mask (1-m)*-10000, scale 1/sqrt(dh) applied after QK^T, as the reference does.
 *   x (n, seq_len, hidden) dt; key_mask (n, seq_len) f32
 *   wqkv (3*hidden, hidden) dt = [query; key; value] weights stacked; bqkv (3*hidden) f32
 *   wo (hidden, hidden) dt; bo (hidden) f32; ln_{g,b} (hidden) f32
 * Requirements: seq_len <= 128, hidden % (32*n_heads) == 0.
 * --------------------------------------------------------------------------------------------- */
size_t xml_attention_block_workspace_bytes(int64_t n, int seq_len, int hidden, int dt);
int xml_attention_block(const void* x, const float* key_mask, const void* wqkv, const float* bqkv,
                        const void* wo, const float* bo, const float* ln_g, const float* ln_b,
                        void* y, int64_t n, int seq_len, int hidden, int n_heads, int dt, void* ws,
                        size_t ws_bytes, xml_stream_t stream);

/* The attention core alone: BertSelfAttention.forward (xml/model_components.py:266-303) behind its three projections --
 *   out (n, lq, hidden) dt, head h = columns [h dh, (h+1) dh)  =  softmax(Q_h K_h^T / sqrt(dh) + (1 - m) * -10000) V_h
 * q (n, lq, ldq), k / v (n, lk, ldk / ldv) dt: projected states, leading dimensions in elements (column blocks of a stacked
 * QKV tensor are passed as offset pointers).  The mask is the outer product q_mask (n, lq) x k_mask (n, lk) f32 -- the only
 * forms the reference builds (key mask broadcast over queries, q_mask = NULL; cross attention, xml/model_xml.py:357-359).
 * lq, lk <= 128, hidden % (32 * n_heads) == 0, dt in {XML_F32, XML_BF16}.  For callers of the sub-module; the encoders use
 * xml_attention_block / xml_cross_attention. */
int xml_attention_core(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* q_mask,
                       const float* k_mask, void* out, int64_t n, int lq, int lk, int hidden, int n_heads, int dt,
                       xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3+K4 and K5 on PACKED variable-length sequences: the query encoder without its padding rows.
 * The reference pads every query to the batch maximum (start_end_dataset.py:346-359; TVR: 30 tokens, mean 17.5 valid)
 * and computes all padded rows; here the valid tokens of n sequences lie back to back -- x / y (rows, hidden),
 * sequence i = rows cu_seqlens[i] .. cu_seqlens[i+1]-1 (cu_seqlens: n+1 int32, cu_seqlens[n] == rows), every sequence
 * 1 .. max_len <= 32 tokens.  Per valid token the result is that of xml_attention_block / xml_modular_pool on the padded
 * batch: projections and LayerNorm are row-wise, and a padded key adds exp(-10000 + s - max) = +0 to the softmax sum and
 * 0 * v to P V.  (K1+K2 on packed rows: xml_linear_ln_relu_pos_packed below.)  hidden % (32 * n_heads) == 0 as for xml_attention_block; xml_modular_pool_varlen: hidden <= 1024.
 * --------------------------------------------------------------------------------------------- */
/* Packing plan of a padded batch: mask (n, lq) f32, every row a non-empty prefix of ones (lq <= 64).
 *   cu_seqlens (n + 1) int32, src_row (>= n * lq) int32: packed token i is row src_row[i] = seq * lq + t of the padded batch;
 *   status (2) int32: status[0] = number of packed rows, or -1 when some mask row is not such a prefix (the caller keeps
 *   the padded path).  Three small launches; the caller reads status[0] back (its launch shapes depend on it).
 * K1+K2 on the packed tokens: xml_linear_ln_relu_pos with the source rows read through src_row (x is the PADDED batch,
 * (n * lq, d_in)) and the positional row of token i = pos[src_row[i] % lq].  d_in % 8 == 0, d_in <= 4096. */
int xml_pack_plan(const float* mask, int64_t n, int lq, int32_t* cu_seqlens, int32_t* src_row, int32_t* status,
                  xml_stream_t stream);
size_t xml_linear_ln_relu_pos_packed_workspace_bytes(int64_t rows, int d_in, int hidden, int dt);
int xml_linear_ln_relu_pos_packed(const void* x, int x_dt, const int32_t* src_row, int lq, const float* ln_in_g,
                                  const float* ln_in_b, const void* w, const float* b, const void* pos,
                                  const float* ln_pos_g, const float* ln_pos_b, void* y, int64_t rows, int d_in,
                                  int hidden, int dt, void* ws, size_t ws_bytes, xml_stream_t stream);
size_t xml_attention_block_varlen_workspace_bytes(int64_t rows, int hidden, int dt);
int xml_attention_block_varlen(const void* x, const int32_t* cu_seqlens, const void* wqkv, const float* bqkv,
                               const void* wo, const float* bo, const float* ln_g, const float* ln_b, void* y,
                               int64_t rows, int64_t n, int max_len, int hidden, int n_heads, int dt, void* ws,
                               size_t ws_bytes, xml_stream_t stream);
int xml_modular_pool_varlen(const void* enc, const int32_t* cu_seqlens, const float* w_m, void* out, int64_t n,
                            int max_len, int hidden, int n_mod, int dt, xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Cross-attention step of XML.cross_context_encoder (xml/model_xml.py:369-371):
 *   y = LN( MHA(q=main, k=v=side, mask = main_mask (x) side_mask) + main )
 *   main (n, lq, hidden), side (n, lk, hidden) dt; masks f32
 *   wq (hidden,hidden), wkv (2*hidden,hidden) = [key; value] dt; bq (hidden), bkv (2*hidden) f32
 * --------------------------------------------------------------------------------------------- */
size_t xml_cross_attention_workspace_bytes(int64_t n, int lq, int lk, int hidden, int dt);
int xml_cross_attention(const void* main_x, const float* main_mask, const void* side_x,
                        const float* side_mask, const void* wq, const float* bq, const void* wkv,
                        const float* bkv, const float* ln_g, const float* ln_b, void* y, int64_t n,
                        int lq, int lk, int hidden, int n_heads, int dt, void* ws, size_t ws_bytes,
                        xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K5: modular query pooling, XML.get_modularized_queries (xml/model_xml.py:410-423)
 *   a = softmax_l( mask_logits(enc W_m^T) );  out[m] = sum_l a[l,m] enc[l]
 *   enc (n, lq, hidden) dt; mask (n, lq) f32; w_m (n_mod, hidden) f32, n_mod in {1,2}
 *   out (n_mod, n, hidden) dt     (n_mod == 1: the reference returns the same vector twice)
 * --------------------------------------------------------------------------------------------- */
int xml_modular_pool(const void* enc, const float* mask, const float* w_m, void* out, int64_t n,
                     int lq, int hidden, int n_mod, int dt, xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Plain linear y = x W^T + b  (video_query_linear / sub_query_linear, xml/model_xml.py:459-460,524)
 *   x (rows, k) dt; w (n, k) dt; b (n) f32 or NULL; y (rows, n) dt.  k % 8 == 0, n % 8 == 0.
 * --------------------------------------------------------------------------------------------- */
int xml_linear(const void* x, const void* w, const float* b, void* y, int64_t rows, int n, int k,
               int relu, int dt, xml_stream_t stream);
/* y = act(x W^T + b) + addend, addend (rows, n) dt added in the GEMM epilogue (one rounding).  The training step's dX of a
 * projection whose input also feeds a residual connection: the residual's gradient rides in as the addend instead of an
 * accumulation launch behind the GEMM (autograd.QkvResFn). */
int xml_linear_add(const void* x, const void* w, const float* b, const void* addend, void* y, int64_t rows, int n, int k,
                   int relu, int dt, xml_stream_t stream);

/* Row-wise L2 normalisation, F.normalize(x, dim=-1) eps=1e-12 (xml/model_xml.py:446-447).
 * Done once per corpus for feat1 ("ctx normalisation precomputed at corpus-encode time"). */
int xml_l2norm_rows(const void* x, void* y, int64_t rows, int d, int dt, xml_stream_t stream);

/* Dataset-side feature normalisation ("next" row 8f-3), l2_normalize_np_array (utils/basic_utils.py:82-84):
 * y = x / (||x||_2 + eps), eps = 1e-5 in the reference; f32 (rows, d); all-zero padding rows stay zero. */
int xml_l2norm_rows_eps(const float* x, float* y, int64_t rows, int d, float eps, xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K6: video-level scores = similarity GEMM #1 with fused masked max over clips
 *   out[q,v] = max_l mask_logits( qn[q] . cn[v,l] )      (xml/model_xml.py:448-452)
 *   combine == 0: out = s;   combine == 1: out = (out + s) * 0.5   ((video+sub)/divisor, :572-574)
 *   qn (nq, hidden) dt, L2-normalised;  cn (nv, lpad, hidden) dt, L2-normalised (zero rows beyond a
 *   video's stored length);  mask (nv, lpad) f32;  out (nq, nv) f32, row stride ld_out.
 * Requirements: lpad % 16 == 0, lpad <= 128, hidden % 8 == 0.
 * --------------------------------------------------------------------------------------------- */
int xml_q2c_scores(const void* qn, const void* cn, const float* mask, float* out, int64_t ld_out,
                   int nq, int nv, int lpad, int hidden, int combine, int dt, xml_stream_t stream);

/* K6 for all modalities of the model in ONE launch (the form the retrieval engine uses):
 *   out[q,v] = ( sum_m max_l mask_logits( qn[m][q] . cn[m][v,l] ) ) / n_mod        n_mod in {1, 2}
 * = get_video_level_scores per modality + the (video + sub) / divisor of xml/model_xml.py:572-574.
 * At lpad == 128 this is a persistent kernel: one workgroup per CU, both modalities of a 256 x 256 tile back to
 * back, a single never-drained LDS-DMA stream (tvretrieval_amd/csrc/q2c_persist.hip); other paddings run the
 * per-modality kernels.  qn1 / cn1 / mask1 are ignored when n_mod == 1. */
int xml_q2c_scores_fused(int n_mod, const void* qn0, const void* cn0, const float* mask0, const void* qn1,
                         const void* cn1, const float* mask1, float* out, int64_t ld_out, int nq, int nv,
                         int lpad, int hidden, int dt, xml_stream_t stream);

/* Slice-major operand tiles for K6 (a private layout of the resident index: the reference has no counterpart, it
 * re-normalises and re-reads row-major video_feat1 for every query batch, xml/model_xml.py:446-448).
 *   xml_q2c_tile_rows: src (rows, hidden) row-major  ->  dst [row / 256][byte / 64][row % 256][64 B]; rows beyond
 *     `rows` are zero; dst holds xml_q2c_tiled_bytes(rows, hidden, dt) bytes.  A 256-row tile keeps its byte offset
 *     (256 * hidden * sizeof(dt) per tile), inside it every 64-byte K slice of all 256 rows is contiguous (16 KiB) =
 *     the LDS image the persistent kernel streams: every DMA piece is 1 KiB contiguous and every 128-byte line is
 *     requested once instead of twice.
 *   xml_q2c_scores_tiled: xml_q2c_scores_fused on tiled operands: qt_m = tiles of qn[m] (nq, hidden),
 *     ct_m = tiles of cn[m] viewed as (nv * 128, hidden).  Same arithmetic, same summation order: bitwise the same
 *     scores.  Requires xml_q2c_tiled_ok(lpad, hidden, dt) (lpad == 128, hidden * sizeof(dt) % 128 == 0, >= 384).
 *     mask_mode 0: the f32 masks are applied through LDS patches (four ring slots).
 *     mask_mode 1: the caller vouches that every entry of the masks is 1 (full-length videos); the masks are not read
 *       and the LDS they would occupy becomes a fifth ring slot (+3.8 % measured).
 *     mask_mode 2: BINARY masks packed as bits, mbits_m (nv, 4) uint32, bit l of video v = mask_m[v][l] != 0; they reach
 *       the kernel through scalar loads, so ragged corpora get the fifth slot as well.  Same scores in every mode. */
int xml_q2c_tiled_ok(int lpad, int hidden, int dt);
/* Packed corpus image for ragged corpora (real TVR: mean 51 of 128 clips): every video is padded to a
 * multiple of 16 clips and the videos are laid back to back into the 256 columns of a K6 tile (any arrangement whose padded
 * lengths sum to <= 256 per tile), so padding rows cost (almost) no MFMA work.
 *   xml_q2c_tile_rows_gather: tiled image (as xml_q2c_tile_rows) of the rows src[row_map[i]], i < rows_packed
 *     (rows_packed % 256 == 0; row_map[i] < 0: zero row).  src (any rows, hidden) row-major.
 *   xml_q2c_scores_packed: K6 on that image, n_tiles 256-row tiles.  A tile is two wave tiles of 128 columns = 8 blocks of
 *     16 columns.  slot_ids (2*n_tiles, 8) int32, one code per block: id >= 0 = the block is the LAST block of video id
 *     (original numbering); -1 = the video continues in the next block; -2 = unused block; -3 (block 7 of an even wave tile
 *     only) = the video continues in block 0 of the odd wave tile of the same tile.  mbits_m (2*n_tiles, 4) uint32 = the
 *     binary clip masks of a wave tile's 128 columns (bit c of the 128 = column c).  out[q, id] is written for every
 *     id >= 0 -- in the video's original column, so everything downstream is unchanged and the scores are bitwise those of
 *     xml_q2c_scores_tiled (mask_logits then max over clips; a video with no valid clip scores -1e10 as there). */
int xml_q2c_tile_rows_gather(const void* src, const int32_t* row_map, void* dst, int64_t rows_packed, int hidden, int dt,
                             xml_stream_t stream);
/* F.normalize(dim=-1) of the clip rows fused into the tiling pass: dst = tiled image (rows_dst rows, a multiple of 256) of
 * l2norm(src rows); row_map NULL: destination row i = source row i (zeros beyond rows_src); row_map (rows_dst entries):
 * destination row i = source row row_map[i], zeros when < 0 (the length-bucketed image).  Bitwise the values of
 * xml_l2norm_rows followed by xml_q2c_tile_rows / _gather.  hidden * sizeof(dt) a multiple of 64 and <= 4096 bytes
 * (xml_q2c_tile_rows_l2norm_ok). */
int xml_q2c_tile_rows_l2norm_ok(int hidden, int dt);
int xml_q2c_tile_rows_l2norm(const void* src, const int32_t* row_map, void* dst, int64_t rows_src, int64_t rows_dst,
                             int hidden, int dt, xml_stream_t stream);
int xml_q2c_scores_packed(int n_mod, const void* qt0, const void* ct0, const void* qt1, const void* ct1, float* out,
                          int64_t ld_out, int nq, int n_tiles, const int32_t* slot_ids, const uint32_t* mbits0,
                          const uint32_t* mbits1, int hidden, int dt, xml_stream_t stream);
int64_t xml_q2c_tiled_bytes(int64_t rows, int hidden, int dt);
int xml_q2c_tile_rows(const void* src, void* dst, int64_t rows, int hidden, int dt, xml_stream_t stream);
int xml_q2c_scores_tiled(int n_mod, const void* qt0, const void* ct0, const float* mask0, const void* qt1,
                         const void* ct1, const float* mask1, float* out, int64_t ld_out, int nq, int nv,
                         int lpad, int hidden, int dt, int mask_mode, const uint32_t* mbits0, const uint32_t* mbits1,
                         xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K8: per-row top-k, torch.topk(exp(alpha*s), k) (xml/inference.py:317,347-348)
 *   scores (rows, n) f32 row stride ld; optional idx_in (rows, n) int32 payload (NULL: column index)
 *   out_val (rows, k) f32 = alpha != 0 ? expf(alpha*s) : s, descending; out_idx (rows, k) int32.
 *   Order: value descending, then payload/column index ascending (torch leaves ties unspecified).
 *   k <= 256, k <= n.
 * --------------------------------------------------------------------------------------------- */
size_t xml_topk_rows_workspace_bytes(int rows, int n, int k);
int xml_topk_rows(const float* scores, int64_t ld, const int32_t* idx_in, float* out_val,
                  int32_t* out_idx, int rows, int n, int k, float alpha, void* ws, size_t ws_bytes,
                  xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Exact-rank mode: the bf16 K6 pass as a FILTER in front of f32 scores (inference.vcmr_search on an index built with
 * exact_filter=True).  The reference ranks videos by f32 scores (torch.topk, xml/inference.py:347-348); a bf16 K6 moves
 * ~1 % of the top-100 memberships where neighbouring scores are closer than its ~3e-5 resolution.  Here corpus and queries
 * are encoded in f32, K6 runs on both operands rounded once to bf16 and proposes m >= k candidates per query (K8), the
 * candidates are re-scored against the f32 operands, and a per-query certificate says whether any video outside the
 * candidate set could still belong to the f32 top-k; failing queries get a full f32 K6 row from the caller.
 *
 *   xml_round_bf16_rows_err: y (rows, d) f32 L2-normalised rows -> yb (rows, d) bf16 = round-to-nearest-even(y),
 *     err (rows) f32 = || y - yb ||_2  (d % 8 == 0).
 *   xml_q2c_rescore: get_video_level_scores (xml/model_xml.py:436-453) + (video + sub) / divisor (:572-574) for the
 *     LISTED pairs only: out[q, j] = ( sum_m max_l mask_logits( qn_m[q] . cn_m[pair_vid[q, j], l] ) ) / n_mod.
 *     qn_m (nq, hidden), cn_m (nv, lpad, hidden) dt, ROW-major, L2-normalised; mask_m (nv, lpad) f32; pair_vid (nq, kpairs)
 *     int32 (< 0 or >= nv: out = -inf); out (nq, kpairs) f32.  The pair list is inverted on the device like K7's, so a
 *     video's tile is fetched once per 64 pairs.  lpad % 16 == 0, lpad <= 128, hidden % 8 == 0.
 *   xml_exact_certificate: filter_scores (nq, m) f32 = the bf16 pass's top-m values per query, descending (xml_topk_rows,
 *     alpha = 0); top_val (nq, k) f32 = the top-k RE-SCORED values, descending, raw on entry -- on return
 *     expf(alpha * s) when alpha != 0 (what K8 emits); eq_m (nq) f32 = rounding-error norms of the query vectors
 *     (xml_round_bf16_rows_err), ec_m = the largest rounding-error norm of the corpus rows of modality m; slack = bound of
 *     the f32 accumulation error of the dot products; outside = 1 if videos exist outside the candidate set (nv > m).
 *     fail[q] = outside && !( filter_scores[q, m-1] + eps_q < top_val[q, k-1] ),
 *     eps_q = mean_m( eq_m[q] * c + (c + eq_m[q]) * ec_m ) + slack, c = 1 + 1e-6 (Cauchy-Schwarz on the two rounding errors).
 *     eps_out (nq) f32 or NULL; thr_out (nq) f32 or NULL = top_val[q, k-1] (raw) - eps_q, the second-tier line of
 *     xml_select_ge_rows; *n_fail (int32, zeroed by the caller) += number of failing queries.
 * --------------------------------------------------------------------------------------------- */
int xml_round_bf16_rows_err(const float* y, void* yb, float* err, int64_t rows, int d, xml_stream_t stream);
/* Second tier of the exact-rank mode: per row of scores (rows, n) f32 (row stride ld) the columns whose value is >= thr[row].
 * idx == NULL: cnt[row] = how many.  idx (rows, cap) int32: the first cap of them in unspecified order (entries beyond
 * cnt[row] are left untouched), cnt[row] = how many exist.  With thr = T_k - eps_q of xml_exact_certificate these are ALL
 * videos that can still belong to a failing query's f32 top-k: T_k (the k-th re-scored value over the first candidates) is a
 * lower bound of the final k-th score, and a video below thr has an f32 score < thr + eps_q = T_k. */
int xml_select_ge_rows(const float* scores, int64_t ld, const float* thr, int32_t* idx, int cap, int32_t* cnt, int rows,
                       int n, xml_stream_t stream);
size_t xml_q2c_rescore_workspace_bytes(int nq, int nv, int kpairs);
int xml_q2c_rescore(int n_mod, const void* qn0, const void* qn1, const void* cn0, const void* cn1, const float* mask0,
                    const float* mask1, const int32_t* pair_vid, float* out, int nq, int nv, int kpairs, int lpad,
                    int hidden, int dt, void* ws, size_t ws_bytes, xml_stream_t stream);
int xml_exact_certificate(const float* filter_scores, int m, float* top_val, int k, const float* eq0, const float* eq1,
                          float ec0, float ec1, int n_mod, float slack, float alpha, int outside, int32_t* fail,
                          float* eps_out, float* thr_out, int32_t* n_fail, int nq, xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Exact-rank mode on the 16-bit pipe (XML_F16 / XML_F16S).  The three stages that must reproduce the reference's f32 scores
 * -- query encoder, candidate re-score, ConvSE (xml/model_xml.py:291-295,436-453,455-502) -- ran on v_mfma_f32_16x16x4_f32,
 * 1/16 of the f16 / bf16 MFMA rate.  An f32 value x is carried instead as two halves, x S = hi + lo (S a power of two,
 * hi = rn_f16(x S), lo = rn_f16(x S - hi): |x S - hi - lo| <= 2^-22 |x S|), and a dot product as hi.hi + lo.hi + hi.lo with
 * f32 accumulation: three f16 MFMAs per 32 k instead of eight f32 ones; the dropped lo.lo term is <= 2^-22 |x||y|.  The
 * FILTER (K6) runs on the hi planes alone (XML_F16): same MFMA rate as bf16, rounding error 8x smaller, so half as many
 * candidates per query carry the same certificate.
 *
 *   xml_split_f16_rows: x (rows, k) f32 -> y (rows, k) XML_F16S = per 32 elements [32 x hi | 32 x lo] f16 (4 bytes per
 *     element; the 128-byte K step of the gathered-pair kernels), inv_scale (rows) f32 = 1 / S or NULL; optionally the plain
 *     hi plane (rows, k) f16 -- K6's XML_F16 operand, to be tiled with xml_q2c_tile_rows -- and err (rows) f32 =
 *     || x - hi / S ||_2, the rounding-error norm the certificate is made of.  fixed_log2 = -1: S = the power of two that
 *     brings the row maximum into [2^13, 2^14); otherwise S = 2^fixed_log2 for every row (XML_F16_UNIT_LOG2 for the
 *     unit-norm similarity operands: |x| <= 1 required).  Subnormal halves are flushed to zero.  k % 32 == 0.
 *   xml_unsplit_f16_rows: the inverse, x = (hi + lo) * inv_scale[row]  (tests, exporting a split index).
 *   xml_pack_weights_f16s: w (n, k) f32 nn.Linear weight -> dst: (n, 3k) f16 = [hi | hi | lo] at ONE power-of-two scale,
 *     followed (16-byte aligned) by a 16-byte trailer whose first float is 1 / S; xml_pack_weights_f16s_bytes(n, k) bytes.
 *     A projection with dt = XML_F16S (xml_linear_ln_relu_pos[_packed], xml_attention_block[_varlen], xml_cross_attention,
 *     xml_linear_f16s) takes such weights, f32 activations / positional table / outputs, splits its input rows
 *     into [hi | lo | hi] per call (inside its workspace) and runs the unchanged f16 GEMM over K' = 3k.  k % 8 == 0.
 *   xml_linear_f16s: y (rows, n) f32 = [relu](x W^T + b), x (rows, k) f32, W from xml_pack_weights_f16s
 *     (the query linears, xml/model_xml.py:462,516).
 *   xml_convse_rerank_f16s: xml_convse_rerank with q_lin / feat2 as XML_F16S rows (desc.dt = XML_F16S, hidden % 32 == 0)
 *     and their per-row 1 / S: q_inv_m (nq) f32, c_inv_m (nv * lpad) f32.
 *   xml_q2c_scores_tiled / xml_q2c_scores_packed take dt = XML_F16 (tiles of hi planes at the fixed unit scale) and
 *   xml_q2c_rescore takes dt = XML_F16S (rows from xml_split_f16_rows(fixed_log2 = XML_F16_UNIT_LOG2)); both undo 2^-28.
 * --------------------------------------------------------------------------------------------- */
int xml_split_f16_rows(const float* x, void* y, float* inv_scale, void* hi_plane, float* err, int64_t rows, int k,
                       int fixed_log2, xml_stream_t stream);
int xml_unsplit_f16_rows(const void* y, const float* inv_scale, float* x, int64_t rows, int k, xml_stream_t stream);
size_t xml_pack_weights_f16s_bytes(int n, int k);
int xml_pack_weights_f16s(const float* w, void* dst, int n, int k, xml_stream_t stream);
size_t xml_linear_f16s_workspace_bytes(int64_t rows, int k);
int xml_linear_f16s(const float* x, const void* w, const float* b, float* y, int64_t rows, int n, int k, int relu, void* ws,
                    size_t ws_bytes, xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K7: similarity contraction #2 + ConvSE start/end scorer on selected (query, video) pairs
 *   sim_m[l]  = q'_m . feat2_m[v,l]                       m in {video, sub}
 *   merged:   x = (sim_v + sim_s)/2 ; st = conv(x, w_st[0]) ; ed = conv(x, w_ed[0])
 *   separate: st = (conv(sim_v, w_st[0]) (+ conv(sim_s, w_st[1]))) / n_mod   (same for ed)
 *   then mask_logits, optionally softmax over l in [0, l_ref)
 * replaces get_merged_st_ed_prob / _get_st_ed_prob (xml/model_xml.py:455-551) and the softmax at
 * xml/inference.py:321-322.  The reference evaluates all (q,v) pairs and discards all but the
 * top-k videos (xml/inference.py:365-367); here only the listed pairs are evaluated.
 *   q_lin[m]  (nq, hidden) dt  = module_query_linear(modular query)
 *   feat2[m]  (nv, lpad, hidden) dt (rows l >= l_ref are zero); mask (nv, lpad) f32
 *   pair_vid  (nq, kpairs) int32 video index per pair, < 0 = skip (output rows zero-filled)
 *   conv_w    (2 * n_conv * ksize) f32: [st filters..., ed filters...]; n_conv = merged ? 1 : n_mod
 *   st_out, ed_out (nq, kpairs, lpad) f32
 * Requirements: lpad % 16 == 0, lpad <= 128, ksize odd <= 15, n_mod in {1,2}.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  int nq, nv, kpairs, lpad, l_ref, hidden;
  int n_mod;        /* number of modalities present (1 or 2) */
  int merged;       /* 1: average similarities then one conv pair (merge_two_stream) */
  int ksize;        /* conv kernel size (config.conv_kernel_size, default 5) */
  int softmax;      /* bit 0: 1 = emit probabilities, 0 = emit masked logits;
                       bit 1: leave the rows of skipped pairs (pair_vid < 0) unwritten instead of zero-filling */
  int dt;
} xml_convse_desc;
size_t xml_convse_rerank_workspace_bytes(const xml_convse_desc* d);
int xml_convse_rerank(const xml_convse_desc* d, const void* q_lin0, const void* q_lin1,
                      const void* feat2_0, const void* feat2_1, const float* mask0,
                      const float* mask1, const int32_t* pair_vid, const float* conv_w,
                      float* st_out, float* ed_out, void* ws, size_t ws_bytes, xml_stream_t stream);
/* split-f16 operands (see "Exact-rank mode on the 16-bit pipe"); same workspace as xml_convse_rerank */
int xml_convse_rerank_f16s(const xml_convse_desc* d, const void* q_lin0, const void* q_lin1, const float* q_inv0,
                           const float* q_inv1, const void* feat2_0, const void* feat2_1, const float* c_inv0,
                           const float* c_inv1, const float* mask0, const float* mask1, const int32_t* pair_vid,
                           const float* conv_w, float* st_out, float* ed_out, void* ws, size_t ws_bytes,
                           xml_stream_t stream);

/* K7 with its two extensions (xml_convse_rerank or _f16s by desc.dt; the q_inv / c_inv pointers are read for XML_F16S only;
 * desc.softmax bit 0 must be set; at least one of summ_out / vid_len is given):
 *  - vid_len (nv) int32 or NULL -- RAGGED CORPORA: valid clips of every video, 1 + the index of its last unmasked clip (TVR:
 *    51 of 128 on average); l_ref for a video without any unmasked clip (its masked softmax is uniform, not zero).  Given: clip rows >= vid_len[v] + ksize / 2 of a video are not fetched (no tap of a valid
 *    position reaches them) and the entries l >= vid_len[v] of st_out / ed_out -- exact zeros of the masked softmax -- are
 *    NOT WRITTEN (whole 16-byte pieces are stored: up to 3 of those zeros behind vid_len[v] may be); their consumer, xml_moment_topk_ex with the same vid_len, does not read them.  Halves K7's writes and
 *    K9's reads at the TVR length distribution; every stored value is bitwise that of xml_convse_rerank.
 *  - summ_out (nq, kpairs, XML_MOMENT_SUMM) f32 or NULL -- per pair, XML_MOMENT_SUMM banded row maxima
 *    (st[i] * pair_w[p]) * max_{min_l <= d < max_l} ed[i + d]  -- the largest one of each group of 16 rows
 *    { i : i % 64 in [8 g, 8 g + 8) } -- taken while the pair's rows are still in registers.  xml_moment_topk_ex builds its
 *    selection threshold from those values instead of a first pass over the st / ed rows.  pair_w (nq, kpairs) f32 = the
 *    video weights exp(alpha * s) (NULL: 1); rows of skipped pairs are not written. */
#define XML_MOMENT_SUMM 8
int xml_convse_rerank_ex(const xml_convse_desc* d, const void* q_lin0, const void* q_lin1, const float* q_inv0,
                         const float* q_inv1, const void* feat2_0, const void* feat2_1, const float* c_inv0,
                         const float* c_inv1, const float* mask0, const float* mask1, const int32_t* pair_vid,
                         const float* conv_w, const float* pair_w, int min_l, int max_l, const int32_t* vid_len,
                         float* st_out, float* ed_out, float* summ_out, void* ws, size_t ws_bytes, xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K9+K10: banded moment candidates + per-query top-n
 *   score(r,i,j) = (st[q,r,i] * w[q,r]) * ed[q,r,j]   for min_l <= j-i < max_l, j < l_ref
 * replaces einsum("qvm,qv,qvn->qvmn") * length mask, flat sort, [:max_before_nms]
 * (xml/inference.py:365-386, :170-192) and the index decoding of :423-431; with kpairs == 1 it is the
 * SVMR path (get_svmr_res_from_st_ed_probs, xml/inference.py:195-241, utils/tensor_utils.py:133-141).
 *   st, ed (nq, kpairs, lpad) f32 probabilities (>= 0: the pruning bound uses it); w (nq, kpairs) f32 >= 0 or NULL (= 1)
 *   out_score (nq, n_out) f32 descending; out_flat (nq, n_out) int32 = (r*l_ref + i)*l_ref + j
 *   (the reference's flat index); ties broken by ascending flat index; when fewer than n_out
 *   candidates exist the tail is score 0 / flat -1.
 * Requirements: n_out <= 1024, lpad <= 128.
 * --------------------------------------------------------------------------------------------- */
int xml_moment_topk(const float* st, const float* ed, const float* w, float* out_score,
                    int32_t* out_flat, int nq, int kpairs, int lpad, int l_ref, int min_l, int max_l,
                    int n_out, xml_stream_t stream);
/* summ != NULL: (nq, kpairs, XML_MOMENT_SUMM) f32 from xml_convse_rerank_ex (same pair_w, min_l, max_l, l_ref): the selection
 * threshold comes from these instead of a pass over the rows; pairs of weight 0 are ignored.  Same lists, bit for bit.
 * pair_vid (nq, kpairs) int32 + vid_len (n_videos) int32 (both or neither): ragged corpora -- the entries l >= vid_len[v] of
 * the st / ed rows of a pair with video v are taken as 0 WITHOUT being read (xml_convse_rerank_ex with the same vid_len left
 * them unwritten), pairs with pair_vid < 0 as empty.  Same lists, bit for bit.
 * Batches too small to fill the chip with one 256-thread workgroup per query (nq <= 128: the reference's eval_query_bsz = 50)
 * run 1024-thread workgroups -- the same algorithm over sixteen waves, the same lists, bit for bit. */
int xml_moment_topk_ex(const float* st, const float* ed, const float* w, const float* summ, const int32_t* pair_vid,
                       const int32_t* vid_len, float* out_score, int32_t* out_flat, int nq, int kpairs, int lpad, int l_ref,
                       int min_l, int max_l, int n_out, xml_stream_t stream);
/* The span predictor as a module of its own: nn.Conv1d(1, 1, ksize, padding = ksize / 2, bias = False) on rows of
 * similarities (self.merged_st_predictor(similarity), xml/model_xml.py:476-477; profile_main.py:204-205 calls it directly).
 *   x, y (rows, l) f32; w (ksize) f32; y[r][i] = sum_t w[t] x[r][i + t - ksize / 2], zero beyond the row.  ksize odd <= 15.
 * The retrieval pass applies the taps inside xml_convse_rerank; this entry is for callers that already hold similarities. */
int xml_conv1d_rows(const float* x, const float* w, float* y, int64_t rows, int l, int ksize, xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Feature ingest ("next" row 8f-3): the dataset-side half of the context collate on the device -- truncate to max_len,
 * x / (||x||_2 + eps) per clip (l2_normalize_np_array, utils/basic_utils.py:82-84; xml/start_end_dataset.py:311-321), zero
 * padding to the batch maximum + float mask (pad_sequences_1d, xml/start_end_dataset.py:346-359).
 *   src (rows, d) XML_F32 or XML_F16 (IEEE half: the feature store's on-disk type): the batch's clip rows back to back
 *   row_start (n + 1) int64 DEVICE: video i = source rows [row_start[i], row_start[i + 1]), the first max_len of them kept
 *   dst (n, lmax, d) XML_F32 or XML_BF16; mask (n, lmax) f32 or NULL; normalize = 0: convert + pad only
 * --------------------------------------------------------------------------------------------- */
int xml_ingest_rows(const void* src, int src_dt, const int64_t* row_start, void* dst, int dst_dt, float* mask, int n,
                    int lmax, int d, int max_len, float eps, int normalize, xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K10: the index tail of compute_query2ctx_info as a device epilogue -- one 16-byte record per list entry, so a query batch
 * leaves the device in ONE copy and the host never loops over queries or list entries.
 *   replaces np.unravel_index(flat, (max_n_videos, max_ctx_l, max_ctx_l)), sorted_q2c_indices[i, local],
 *   st_idx.astype(np.float32) * clip_length, ed_idx.astype(np.float32) * clip_length + clip_length and the
 *   video2idx[video_metas[meta]["vid_name"]] lookup (xml/inference.py:415-439); the VR list (:402-413); and, with
 *   seconds = 0, the index part of get_svmr_res_from_st_ed_probs (:229-233).
 *   flat  (nq, n) int32 row stride ld_in = xml_moment_topk's out_flat (< 0: no candidate), score (nq, n) f32 same stride
 *   top_idx (nq, k) int32: video (meta) index of local rank r = flat / l_ref^2; NULL: row_vid[q] (SVMR: the query's one
 *     video) or, when that is NULL too, r itself
 *   meta2vid (n_videos) int32: meta index -> video2idx value, NULL = identity
 *   seconds != 0: st = f32(st_idx) * clip_length, ed = f32(ed_idx) * clip_length + clip_length, each operation rounded to
 *     f32 once (numpy float32 arithmetic, no fma); seconds == 0: st = st_idx, ed = ed_idx + 1 in clip units (exact) -- the
 *     reference's SVMR tail scales these in float64 on the host
 *   flat == NULL: the video-retrieval list -- top_idx (nq, n) stride ld_in, records {meta2vid[top_idx], 0, 0, score}
 *   out (nq, n) xml_moment, row stride ld_out records, 16-byte aligned; entries without a candidate: {-1, 0, 0, 0}
 *   out_count (nq) int32 or NULL: number of entries with a candidate (they form a prefix of the row)
 * --------------------------------------------------------------------------------------------- */
typedef struct { int32_t vid; float st; float ed; float score; } xml_moment;
int xml_moments_decode(const int32_t* flat, const float* score, const int32_t* top_idx, const int32_t* row_vid,
                       const int32_t* meta2vid, int nq, int n, int64_t ld_in, int k, int l_ref, float clip_length,
                       int seconds, xml_moment* out, int64_t ld_out, int32_t* out_count, xml_stream_t stream);

/* Row-wise LayerNorm of (a [+ b]) -- exposed for the host-side mirror and tests.
 *   y = LN(a + b) * g + beta;  a,b,y (rows, d) dt (b may be NULL), x_dt of `a` may be XML_F32. */
int xml_add_layernorm(const void* a, int a_dt, const void* b, const float* g, const float* beta,
                      void* y, int64_t rows, int d, int dt, xml_stream_t stream);

/* =============================================================================================
 * TRAINING STEP (BASELINE config 5): backward kernels + loss heads + BertAdam.
 * The reference relies on torch autograd for XML.forward (xml/model_xml.py:212-251) and
 * loss.backward() / optimizer.step() (xml/train.py:78-95); these entries are the hand-written backward of
 * each forward op above.  Activation gradients use dt, parameter gradients are f32 and ACCUMULATE (+=)
 * unless stated otherwise.
 * ============================================================================================= */

/* y[b][c][r] = x[b][r][c]; x (batch, rows, cols) contiguous, y rows have stride ld_out >= rows
 * (columns beyond `rows` are not written: pre-zero a padded destination). */
int xml_transpose_batched(const void* x, void* y, int batch, int rows, int cols, int ld_out, int dt,
                          xml_stream_t stream);
/* out[c] (+)= sum_r x[r][c]   -- bias / positional-table gradients. */
int xml_colsum(const void* x, int x_dt, float* out, int64_t rows, int cols, int accumulate,
               xml_stream_t stream);
/* dx = dy * (y > 0)   -- F.relu backward (LinearLayer, xml/model_components.py:163). */
int xml_relu_bwd(const void* y, const void* dy, void* dx, int64_t n, int dt, xml_stream_t stream);
/* y (y_dt) += x (x_dt). */
int xml_add_inplace(void* y, int y_dt, const void* x, int x_dt, int64_t n, xml_stream_t stream);
/* LayerNorm backward of y = LN(a [+ b]) * g + beta: dx (grad of a and of b), dg +=, dbeta +=.
 *   dx may be NULL (raw input features need no gradient: dg / dbeta only).
 *   d <= 1024: one wave per row.  d > 1024 (input LayerNorm over raw features): b must be NULL,
 *   ws = rows * 16 bytes of scratch (not needed when dx is NULL, bf16, d % 8 == 0, d <= 4096). */
int xml_layernorm_bwd(const void* a, int a_dt, const void* b, const float* g, const void* dy, void* dx,
                      float* dg, float* dbeta, int64_t rows, int d, int dt, void* ws, size_t ws_bytes,
                      xml_stream_t stream);
/* out[z] = scale * A[z] B[z]^T for z < batch; A (M,K), B (N,K) contiguous, out (M,N) dt or f32.
 * K % 8 == 0 (bf16) / % 4 (f32).  Used for the per-head products of attention forward/backward and,
 * with batch = 1 and transposed operands, for dX = dY W and dW = dY^T X of nn.Linear. */
int xml_gemm_batched(const void* A, const void* B, void* out, int batch, int M, int N, int K, float scale,
                     int out_f32, int dt, xml_stream_t stream);
/* token-major (n*L, ld) columns [col0 + h*dh, +dh)  <->  per-(sequence, head) matrices
 *   dst (n, heads, l8, dh) rows >= L zero;  dstT (n, heads, dh, l8) columns >= L zero; either may be NULL. */
int xml_split_heads(const void* src, int ld, int col0, int64_t n, int L, int l8, int heads, int dh,
                    void* dst, void* dstT, int dt, xml_stream_t stream);
int xml_merge_heads(const void* src, void* dst, int ld, int col0, int64_t n, int L, int l8, int heads,
                    int dh, int dt, xml_stream_t stream);
/* BertSelfAttention softmax on materialised scores (xml/model_components.py:288-296) and its backward.
 *   S (n*heads, lq8, lk8) f32 = Q K^T;  P = softmax_j(S / sqrt_dh + (1 - q_mask*k_mask) * -1e4), j < lk.
 *   dP == NULL: writes P and (optionally) P^T, zero padded.   dP != NULL: writes dS = P*(dP - sum P*dP)/sqrt_dh
 *   and dS^T.  q_mask may be NULL (= 1).  lk <= 128. */
int xml_attn_softmax(const float* S, const float* dP, const float* q_mask, const float* k_mask, void* P,
                     void* PT, void* dS, void* dST, int64_t n, int heads, int lq, int lk, int lq8, int lk8,
                     float sqrt_dh, int dt, xml_stream_t stream);
/* Weight-gradient GEMM from the row-major operands: out (N, K) f32 = A^T B = sum_r A[r][n] B[r][k], A (rows, N), B (rows, K)
 * bf16, N and K multiples of 8 (xml_gemm_tn_supported; callers keep the transpose + xml_gemm_batched path otherwise).
 * Row ranges are combined with f32 atomics: the summation order, hence the last bits, vary from run to run (as with
 * xml_gemm_batched's split-K).  colsum_a (N) f32 or NULL: column sums of A from the same launch (a layer's bias gradient).
 * accumulate == 0: out / colsum_a are overwritten.  accumulate != 0: out += A^T B, colsum_a += column sums -- the outputs are
 * gradient buffers that already hold a partial sum (the optimizer's flat .grad buffer, zeroed once per step), no fill.
 * The f32 adds are hardware atomics: out / colsum_a must live in ordinary (coarse-grained) device memory -- on
 * fine-grained or host-coherent allocations the hardware add is silently dropped. */
int xml_gemm_tn_supported(int64_t rows, int N, int K, int dt);
int xml_gemm_tn(const void* A, const void* B, float* out, float* colsum_a, int64_t rows, int N, int K, int dt,
                int accumulate, xml_stream_t stream);      /* colsum_a (N) f32 or NULL: = sum_r A[r][n] (the layer's bias gradient), same launch */
/* Fused training attention (bf16 storage; xml/model_components.py:266-303 incl. the probabilities dropout :297), one
 * launch each way instead of the split_heads / batched GEMM / xml_attn_softmax / xml_dropout / merge_heads chain:
 *   fwd   out (n, lq, ldo) head h columns [h dh, (h+1) dh)  =  dropout(softmax(Q K^T / sqrt(dh) + mask bias)) V
 *   bwd   dq / dk / dv (same row layouts as q / k / v, leading dimensions lddq / lddk / lddv) from dout; P is recomputed,
 *         nothing but q, k, v has to be kept from the forward pass.
 * q (n, lq, ldq), k / v (n, lk, ldk / ldv): head h = columns [h dh, (h+1) dh) from the given pointers (column blocks of a
 * fused QKV tensor are passed as offset pointers).  Leading dimensions in elements, multiples of 8.  lq, lk <= 128;
 * dh = hidden / n_heads in {32, 64, 96, 128, 192}.  The dropout mask is xml_dropout's hash at the element's index in a
 * (n * heads, ceil8(lq), ceil8(lk)) tensor: the same elements the unfused chain drops for this seed.  p_drop = 0: none.
 * xml_attention_train_supported says whether a shape / dtype is served (callers keep the unfused chain otherwise).
 * seed_dev (here and in xml_dropout): NULL, or a DEVICE pointer to a 64-bit base seed that is ADDED to `seed` when the
 * kernel runs -- a training step captured into a HIP graph keeps `seed` (a per-site constant) in its nodes and advances
 * the base seed on the device, so every replay draws fresh masks. */
int xml_attention_train_supported(int lq, int lk, int hidden, int n_heads, int dt);
int xml_attention_train_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* q_mask,
                            const float* k_mask, void* out, int ldo, int64_t n, int lq, int lk, int hidden, int n_heads,
                            float p_drop, uint64_t seed, const uint64_t* seed_dev, int dt, xml_stream_t stream);
int xml_attention_train_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* q_mask,
                            const float* k_mask, const void* dout, int ldo, void* dq, int lddq, void* dk, int lddk,
                            void* dv, int lddv, int64_t n, int lq, int lk, int hidden, int n_heads, float p_drop,
                            uint64_t seed, const uint64_t* seed_dev, int dt, xml_stream_t stream);
/* get_modularized_queries backward (xml/model_xml.py:410-423): denc (n, lq, hidden) dt written, dw_m += . */
int xml_modular_pool_bwd(const void* enc, const float* mask, const float* w_m, const void* dout, void* denc,
                         float* dw_m, int64_t n, int lq, int hidden, int n_mod, int dt, xml_stream_t stream);
/* F.normalize(dim=-1) backward; dy is f32, dx is dt. */
int xml_l2norm_bwd(const void* x, const float* dy, void* dx, int64_t rows, int d, int dt,
                   xml_stream_t stream);
/* xml_q2c_scores backward for the in-batch (N x N) training scores: the gradient of max_l goes to the arg-max
 * clip.  qn (nq,hidden), cn (nv,l,hidden), mask (nv,l); dqn / dcn f32, zeroed here; `scale` multiplies dscores
 * (1 / number of modalities when the scores were averaged). */
int xml_q2c_scores_bwd(const void* qn, const void* cn, const float* mask, const float* dscores,
                       int64_t ld_ds, float scale, float* dqn, float* dcn, int nq, int nv, int l, int hidden,
                       int dt, xml_stream_t stream);
/* get_video_level_scores backward (xml/model_xml.py:436-453) in ONE launch, for the sparse score gradients of the ranking
 * loss: replaces xml_q2c_scores_bwd (two f32 fills + atomics) + a slice copy + two xml_l2norm_bwd passes.
 *   query (nq, hidden), feat (nv, l, hidden): the un-normalised inputs of the forward pass; qn (nq, hidden), cn (nv, lpad,
 *   hidden), mask (nv, lpad): what the forward pass scored (F.normalize'd rows, clips padded to lpad >= l);
 *   dscores (nq, nv) f32 row stride ld_ds, multiplied by `scale`;  dq (nq, hidden), dfeat (nv, l, hidden) dt: every element
 *   written (rows no pair points at are zero).  Per pair with a gradient the arg-max clip is re-derived (first clip on
 *   ties), its gradient goes through the F.normalize backward of that one row.  Any dscores is handled, the cost grows with
 *   the number of non-zeros.  arg (nq, nv) int32 row stride ld_arg: the arg-max clips kept by xml_q2c_scores_arg (no
 *   re-derivation: the pairs' rows are plain gathers), or NULL.
 *   _supported: dt f32 / bf16, nq, nv <= 1024, hidden % 8 == 0, hidden <= 2048.
 * xml_q2c_scores_arg: the forward pass for these in-batch scores -- xml_q2c_scores (same `combine`) on the UNPADDED
 *   cn (nv, l, hidden) / mask (nv, l), l <= 128, that also writes arg[q, v] = the first clip attaining the maximum. */
int xml_q2c_scores_l2norm_bwd_supported(int nq, int nv, int l, int hidden, int dt);
int xml_q2c_scores_l2norm_bwd(const void* query, const void* feat, const void* qn, const void* cn, const float* mask,
                              const float* dscores, int64_t ld_ds, float scale, void* dq, void* dfeat, int nq, int nv, int l,
                              int lpad, int hidden, const int32_t* arg, int64_t ld_arg, int dt, xml_stream_t stream);
/* xml_q2c_scores_l2norm_bwd for the n_mod (1 or 2) modalities of the model in ONE launch: every per-modality argument is a
 * HOST array of n_mod entries (arg may be NULL, or hold NULL entries); dscores / scale / nq / nv / hidden are shared. */
int xml_q2c_scores_l2norm_bwd_multi(int n_mod, const void* const* query, const void* const* feat, const void* const* qn,
                                    const void* const* cn, const float* const* mask, const float* dscores, int64_t ld_ds,
                                    float scale, void* const* dq, void* const* dfeat, int nq, int nv, const int* l,
                                    const int* lpad, int hidden, const int32_t* const* arg, int64_t ld_arg, int dt,
                                    xml_stream_t stream);
int xml_q2c_scores_arg(const void* qn, const void* cn, const float* mask, float* out, int64_t ld_out, int32_t* arg,
                       int64_t ld_arg, int nq, int nv, int l, int hidden, int combine, int dt, xml_stream_t stream);
/* The loss sum of XML.forward (xml/model_xml.py:241-251): parts4 = {w_st_ed * st_ed[0], w_neg_ctx * rank2[0],
 * w_neg_q * rank2[1], their sum}, overall[0] = the sum; st_ed / rank2 NULL: that term is 0.  _bwd: from the gradient g[0] of
 * the sum, d_st_ed[0] = w_st_ed g, d_rank2 = {w_neg_ctx g, w_neg_q g} (NULL: not wanted).  All device f32. */
int xml_loss_combine(const float* st_ed, const float* rank2, float w_st_ed, float w_neg_ctx, float w_neg_q, float* parts4,
                     float* overall, xml_stream_t stream);
int xml_loss_combine_bwd(const float* g, float w_st_ed, float w_neg_ctx, float w_neg_q, float* d_st_ed, float* d_rank2,
                         xml_stream_t stream);
/* einsum("bd,bld->bl") of the cross=False branch (xml/model_xml.py:478-479,532) and its backward. */
int xml_pair_sim(const void* q, const void* f2, float* sim, int64_t n, int l, int hidden, int dt,
                 xml_stream_t stream);
int xml_pair_sim_bwd(const void* q, const void* f2, const float* dsim, void* dq, void* df2, int64_t n, int l,
                     int hidden, int dt, xml_stream_t stream);
/* Span loss head = conv1d start/end predictors + mask_logits + cross entropy (xml/model_xml.py:237-240,
 * 478-500, 532-550, F.cross_entropy mean reduction).  sim_i (n, l) f32; conv_w = [st filters | ed filters],
 * each n_filt x ks with n_filt = merged ? 1 : n_sim; st_ed (n, 2) int64 targets.
 *   gout == NULL: *loss_out = mean_b CE(st) + CE(ed).
 *   gout != NULL: dsim_i and dconv_w (both overwritten) = *gout * d loss. */
int xml_span_loss(const float* sim0, const float* sim1, const float* conv_w, const float* mask0,
                  const float* mask1, const int64_t* st_ed, int merged, int n_sim, int ks, int n, int l,
                  const float* gout, float* loss_out, float* dsim0, float* dsim1, float* dconv_w,
                  xml_stream_t stream);
/* In-batch ranking loss (get_video_level_loss, xml/model_xml.py:588-637) on scores (n, n) f32 with the
 * torch.randint draws of get_neg_scores (:622) passed in as rank positions ranks_ctx / ranks_q (n) int32.
 *   gout == NULL: losses[0] = loss_neg_ctx, losses[1] = loss_neg_q (unweighted).
 *   gout (2 floats) != NULL: dscores (overwritten) = gout[0] * d loss_neg_ctx + gout[1] * d loss_neg_q. */
int xml_rank_loss(const float* scores, const int* ranks_ctx, const int* ranks_q, float margin, int lse,
                  int n, const float* gout, float* losses, float* dscores, xml_stream_t stream);
/* nn.Dropout in training mode (LinearLayer, TrainablePositionalEncoding, BertSelfAttention probabilities,
 * BertSelfOutput; xml/model_components.py:88,151,239,297,315): y = keep(i) ? x / (1 - p) : 0 with a counter-based
 * mask that is a pure function of (seed, element index) -- the backward pass calls it again on the gradient.
 * 0 <= p < 1; y == x allowed.  Not torch's Philox stream: statistically, not bitwise, equal to the reference. */
int xml_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, int dt,
                xml_stream_t stream);
/* Transposed bf16 copies of many f32 weight matrices of ONE flat buffer in one launch -- the W^T operands of the training
 * step's dX = dY W GEMMs, refreshed once per step from the optimizer's flat parameter buffer (BertAdam.refresh_shadows)
 * instead of one xml_transpose_batched per layer and step.  table (device memory): n_ent rows of six int64
 *   {src_off (elements into src), n, k, dst (device address of bf16 storage), ld_dst, col0}:
 *   dst[kk * ld_dst + col0 + nn] = bf16(src[src_off + nn * k + kk]),  nn < n, kk < k   (other dst elements untouched).
 * max_tiles >= max over the entries of ceil(n / 64) * ceil(k / 64).  dt: XML_BF16 only. */
int xml_transpose_segments(const float* src, const int64_t* table, int n_ent, int max_tiles, int dt, xml_stream_t stream);
/* Training LayerNorm with its neighbouring dropout sites applied in place (xml/model_components.py:201-210 BertSelfOutput
 * dense -> dropout -> LayerNorm(+ residual); :103-114 LinearLayer LayerNorm -> dropout -> Linear; :139-156
 * TrainablePositionalEncoding LayerNorm -> dropout):
 *     y = drop_out( LN( drop_in(a) + b ) * g + beta )
 * with xml_dropout's masks (element i of `a` / of `y`, seeds seed_in / seed_out [+ *seed_dev]); p_in / p_out = 0 switch a
 * site off.  a (a_dt f32 or dt), b (dt or NULL), y (dt) are (rows, d) contiguous, d % 8 == 0, d <= 4096; dt f32 or bf16.
 * xml_layernorm_bwd_drop is its backward pass: dy is masked with the output site, dx = gradient of b (and of a when
 * p_in == 0; NULL: parameter gradients only), dxa = gradient of a when p_in > 0 (required then unless dx is NULL, ignored
 * otherwise); dg / dbeta f32 are accumulated into.
 * d <= 1024: all of it.  1024 < d <= 4096: parameter gradients only (b, dx, dxa NULL, p_in == 0, dt bf16). */
int xml_add_layernorm_drop(const void* a, int a_dt, const void* b, const float* g, const float* beta, void* y,
                           int64_t rows, int d, int dt, float p_in, uint64_t seed_in, float p_out, uint64_t seed_out,
                           const uint64_t* seed_dev, xml_stream_t stream);
int xml_layernorm_bwd_drop(const void* a, int a_dt, const void* b, const float* g, const void* dy, void* dx, void* dxa,
                           float* dg, float* dbeta, int64_t rows, int d, int dt, float p_in, uint64_t seed_in, float p_out,
                           uint64_t seed_out, const uint64_t* seed_dev, xml_stream_t stream);
/* The same with scratch for the parameter gradients: every workgroup of the backward launch ends in 2 d column sums for the
 * same 2 d addresses; with ws >= xml_layernorm_bwd_partials_bytes(rows, d) bytes (0: this shape has no use for one) they are
 * written as rows of the scratch and combined by a second small launch instead of rows / 32 contended device-scope adds per
 * address (raw-feature input LayerNorm, 12 800 x 3072: 108 -> 45 us).  ws NULL or too small: the atomics path.  The scratch of
 * xml_layernorm_bwd (rows * 16 bytes for dx at d > 1024) serves the same purpose there when dx is NULL. */
size_t xml_layernorm_bwd_partials_bytes(int64_t rows, int d);
int xml_layernorm_bwd_drop_ws(const void* a, int a_dt, const void* b, const float* g, const void* dy, void* dx, void* dxa,
                              float* dg, float* dbeta, int64_t rows, int d, int dt, float p_in, uint64_t seed_in, float p_out,
                              uint64_t seed_out, const uint64_t* seed_dev, void* ws, size_t ws_bytes, xml_stream_t stream);
/* torch.nn.utils.clip_grad_norm_ over all gradients (xml/train.py:88-90, `--grad_clip`, off by default): g (n) f32 is
 * the flat gradient buffer; scaled in place by max_norm / (||g||_2 + 1e-6) when that is < 1.  ws: 4 bytes of scratch. */
int xml_clip_grad_norm(float* g, int64_t n, float max_norm, float* ws, xml_stream_t stream);
/* BertAdam.step (xml/optimization.py:273-338) over one flat f32 buffer holding every tensor:
 * per-tensor clip_grad_norm_ (gradient rescaled in place), m/v update, m/(sqrt(v)+eps) + wd*p, no bias
 * correction, p -= seg_lr[s] * lr_mult * update.  seg_off (n_seg+1) int64, seg_lr / seg_wd / norms (n_seg) f32,
 * all device memory.
 *   seg_active (n_seg bytes) or NULL (= all): 0 marks a tensor that has never received a gradient -- the reference
 *     skips it entirely (`if p.grad is None: continue`, :289-291): no moments, no weight decay;
 *   seg_lr_mult (n_seg) f32 or NULL: per-tensor schedule multiplier replacing the scalar lr_mult -- the reference keeps
 *     state['step'] per tensor (:325-330), so tensors that join the training later (train_span_start_epoch) restart
 *     their warm-up;
 *   norm_ws: ceil(total / XML_ADAM_NORM_BLOCK) floats of device scratch or NULL.  Given: the per-tensor gradient norms are
 *     summed from per-block partials in a fixed order (25 us for 20 M parameters) instead of by one f32 atomic per block
 *     (92 us). */
#define XML_ADAM_NORM_BLOCK 1024
int xml_bert_adam_step(float* p, float* g, float* m, float* v, const int64_t* seg_off, const float* seg_lr,
                       const float* seg_wd, int n_seg, int64_t total, float lr_mult, float b1, float b2,
                       float eps, float max_grad_norm, float* norms, const uint8_t* seg_active,
                       const float* seg_lr_mult, float* norm_ws, xml_stream_t stream);

/* =============================================================================================
 * MULTI-GPU (BASELINE config 4): collectives of the corpus-sharded pass, RCCL over xGMI, one process per
 * GPU.  The reference has no distributed path; what these entries preserve is its driver semantics -- moments only from
 * the GLOBAL top-k videos of a query (xml/inference.py:347-348,365-367).  The communicator is a plain ncclComm_t: create
 * it here from a 128-byte ncclUniqueId (rank 0 makes it, any side channel -- e.g. torch.distributed's store --
 * distributes it) or pass one made elsewhere with the same RCCL.  RCCL is resolved at run time (dlopen).
 * ============================================================================================= */
typedef void* xml_comm_t; /* ncclComm_t */
int xml_rccl_available(void);                      /* 1 when an RCCL library could be resolved */
int xml_rccl_unique_id(void* id128);               /* HOST pointer to 128 bytes (ncclGetUniqueId) */
int xml_rccl_comm_init(xml_comm_t* comm, int nranks, int rank, const void* id128);   /* collective: all ranks call it */
int xml_rccl_comm_destroy(xml_comm_t comm);
/* recv (nranks * bytes_per_rank) = concatenation over ranks of send (bytes_per_rank): the modular query vectors of every
 * owner's slice (encode_query runs on 1/P of the queries per rank). */
int xml_rccl_allgather(xml_comm_t comm, const void* send, void* recv, int64_t bytes_per_rank, xml_stream_t stream);
/* in-place average of an f32 buffer over ranks (data-parallel gradient buckets, BASELINE config 5; xml/train.py:81-85
 * is the single-process loop it extends). */
int xml_rccl_allreduce_avg_f32(xml_comm_t comm, float* buf, int64_t n, xml_stream_t stream);
/* The plain scheme: ncclAllGather of every rank's local top-c lists (loc_score / loc_id (nq, c) as below), then EVERY rank
 * merges all nq queries: out_val / out_id (nq, k), same order and tie rule as xml_topk_rows.  Two all-gathers in one group,
 * one un-permute kernel, one top-k kernel.  (SURVEY.md 8b names this entry; the engine's default pass uses the by-owner
 * exchange below, which moves and merges 1/world of it per rank.) */
size_t xml_rccl_allgather_topk_workspace_bytes(int world, int nq, int c);
int xml_rccl_allgather_topk(xml_comm_t comm, int world, const float* loc_score, const int32_t* loc_id, int nq, int c,
                            int k, float alpha, float* out_val, int32_t* out_id, void* ws, size_t ws_bytes,
                            xml_stream_t stream);
/* Exact global top-k by query owner.  Queries are split into `world` contiguous slices of per = ceil(nq / world) rows;
 * rank r owns slice r.  Every rank passes its LOCAL top-c of all nq queries -- loc_score (nq, c) f32 descending,
 * loc_id (nq, c) int32 GLOBAL video ids (pad short shards with -inf / INT32_MAX) -- and receives the merged global top-k
 * of its own slice: own_val (rows_owned, k) f32 = alpha != 0 ? exp(alpha * s) : s, own_id (rows_owned, k), ordered by
 * (score desc, id asc) like xml_topk_rows, i.e. exactly the single-GPU list.  k <= 256, k <= world * c.
 * One grouped send/recv (an all-to-all with ragged counts), one un-permute kernel, one top-k kernel. */
size_t xml_rccl_topk_by_owner_workspace_bytes(int world, int per, int c);
int xml_rccl_topk_by_owner(xml_comm_t comm, int world, int rank, const float* loc_score, const int32_t* loc_id,
                           int nq, int c, int k, float alpha, float* own_val, int32_t* own_id, void* ws,
                           size_t ws_bytes, xml_stream_t stream);

/* The owner's side of the two entries above WITHOUT the wire: recv_score / recv_id (world, n_rows, c) = what the grouped
 * receive leaves in the workspace (source-rank-major: [p][row][c], pad with -inf / INT32_MAX) -> one un-permute kernel, one
 * top-k kernel -> out_val / out_id (n_rows, k), ordered like xml_topk_rows.  The collectives call exactly this after their
 * receive; exported so that the merge of a sharded pass can be checked (and reused) without a communicator -- e.g. walking
 * the 8 shards of BASELINE configs[3] through one GPU (tests/test_gpu_fullsize.py).  k <= 256, k <= world * c. */
size_t xml_merge_shard_topk_workspace_bytes(int world, int n_rows, int c);
int xml_merge_shard_topk(const float* recv_score, const int32_t* recv_id, int world, int n_rows, int c, int k, float alpha,
                         float* out_val, int32_t* out_id, void* ws, size_t ws_bytes, xml_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * HOST post-processing ("next" row 8f-1; pointers are HOST memory): greedy temporal NMS.
 *   xml_nms_vcmr_host = filter_vcmr_by_nms (baselines/clip_alignment_with_language/inference.py:189-225):
 *     first max_before predictions, grouped by video in order of appearance, NMS per video
 *     (temporal_non_maximum_suppression, utils/temporal_nms.py:25-74, IoU over the hull, keep while IoU <= thd),
 *     merged, stable-sorted by score, truncated to max_after.
 *   xml_nms_svmr_host = post_processing_svmr_nms (:247-265) for one query.
 *   out_index receives indices into the input arrays in output order; *n_out their number.
 * --------------------------------------------------------------------------------------------- */
int xml_nms_vcmr_host(const int64_t* vid, const double* st, const double* ed, const double* score, int n,
                      double thd, int max_before, int max_after, int32_t* out_index, int32_t* n_out);
int xml_nms_svmr_host(const double* st, const double* ed, const double* score, int n, double thd,
                      int max_before, int max_after, int32_t* out_index, int32_t* n_out);
/* The same for a whole result set at once: (nq, ld) arrays (the columns of K10's records widened to the Python types the
 * reference operates on), count (nq) = valid entries per row; out_index (nq, ld_out >= max_after) int32 indices into each row,
 * out_count (nq).  Queries are spread over n_threads host threads (0: one per hardware thread, at most 64). */
int xml_nms_vcmr_batched_host(const int64_t* vid, const double* st, const double* ed, const double* score,
                              const int32_t* count, int nq, int64_t ld, double thd, int max_before, int max_after,
                              int32_t* out_index, int64_t ld_out, int32_t* out_count, int n_threads);
int xml_nms_svmr_batched_host(const double* st, const double* ed, const double* score, const int32_t* count, int nq,
                              int64_t ld, double thd, int max_before, int max_after, int32_t* out_index, int64_t ld_out,
                              int32_t* out_count, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* XMLHIP_H */
