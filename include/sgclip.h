/*
 * sgclip.h - C ABI of the MI355X (gfx950) SGCLIP scoring kernels.
 *
 * The reference (video-fm/LASER) has no FFI or operator registry for this
 * path: its boundary is the Python call PredicateModel.forward()
 * (laser/models/llava_clip_model_v3.py:86-107), below which every step is a
 * third-party library call.  Each entry point here replaces one of those
 * library calls; the citation says which.
 *
 * Conventions
 *   - every function returns 0 on success, a negative SGCLIP_E* code on error;
 *     sgclip_last_error() returns a thread-local message for the last error;
 *   - pointers named *_dev are device pointers owned by the caller, everything
 *     else is host memory; no function allocates or frees device memory, and
 *     no function synchronises: work is enqueued on `stream` (a hipStream_t
 *     passed as void*, NULL = the default stream);
 *   - matrices are row-major; "bf16" is the upper half of an IEEE fp32.
 */
#ifndef SGCLIP_H
#define SGCLIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGCLIP_OK 0
#define SGCLIP_EINVAL (-1)   /* bad argument (shape, alignment, range)        */
#define SGCLIP_ENOSPC (-2)   /* caller-provided workspace too small           */
#define SGCLIP_ELAUNCH (-3)  /* HIP reported an error at launch               */
#define SGCLIP_EBBOX (-4)    /* empty crop box: the reference's "image bbox issue" assert */

int sgclip_abi_version(void);
const char *sgclip_last_error(void);

/* Opt-in per-kernel timing for benchmarks (no counterpart in the reference).  While
 * enabled, every launch of a timed kernel is bracketed by two HIP events on its launch
 * stream.  kind: 0..4 = sgclip_gemm_bf16 by epilogue, 8 = crop, 9 = attention.
 * sgclip_timing_read waits for the recorded events and returns the summed duration,
 * launch count and summed algorithmic work (FLOPs for GEMM/attention, bytes for crop). */
int sgclip_timing_enable(int on, int max_records);
int sgclip_timing_read(int kind, double *total_ms, int64_t *launches, double *work);

/* ------------------------------------------------------------------------
 * (1) mask-guided crop + resize + normalize.
 *
 * Replaces, per crop, the host chain
 *   extract_single_object / extract_object_subject  (laser/models/model_utils.py:29-56, 119-140)
 *   crop_image_contain_bboxes                        (laser/models/model_utils.py:87-117)
 *   clip_processor(images=...)                       (laser/models/llava_clip_model_v3.py:244, 443)
 * The box in each entry is the already resolved slice [y1:y2, x1:x2] of the
 * frame (0 <= x1 < x2 <= frame_w, same for y).
 * ---------------------------------------------------------------------- */
#define SGCLIP_CROP_OBJECT 0 /* inside mask_a: keep; outside: blend to white            */
#define SGCLIP_CROP_PAIR 1   /* mask_a tints channel 2, mask_b tints channel 0; outside both: white */
#define SGCLIP_CROP_PLAIN 2  /* no blend (the reference's dummy whole-frame "pair")    */

typedef struct sgclip_crop {
    int64_t frame_off;  /* byte offset of the u8[H,W,3] frame inside frames_dev  */
    int64_t mask_a_off; /* byte offset of the u8[H,W] mask inside masks_dev, -1 = none */
    int64_t mask_b_off;
    int32_t frame_w, frame_h;
    int32_t x1, y1, x2, y2;
    int32_t mode;
    int32_t order;      /* scheduling hint, 0 in every record = table order.  Otherwise 1 + the index of the crop that the
                         * j-th group of workgroups takes (record j carries it; the values form a permutation of 1..n).
                         * Largest boxes first evens out the end of a launch whose crops differ in size (-10 % on the
                         * pair crops of a 720p clip).  A crop's output position is its own index either way. */
} sgclip_crop_t; /* 56 bytes */

#define SGCLIP_LAYOUT_CHW 0     /* out[n][c][y][x], n-major, row length S                 */
#define SGCLIP_LAYOUT_PATCHES 1 /* out[n*P + py*G + px][c*p*p + ky*p + kx], row length kpad */
#define SGCLIP_DTYPE_F32 0
#define SGCLIP_DTYPE_BF16 1

/* luts: host u8[768] = white[256] | keep[256] | tint[256] (level tables of the
 * three blends); norm_lut: host f32[768] = value of level v in channel c at
 * [c*256+v] after rescale+normalize.  frames_dev / masks_dev must be 4-byte
 * aligned and readable for 16 bytes past frames_bytes / masks_bytes (rows are
 * copied in 16-byte pieces).
 * crops: host table; crops_dev: device buffer of n_crops entries that this
 * call fills (H2D copy on `stream`) unless crops_dev_ready != 0.
 * lut_dev: device buffer of SGCLIP_CROP_LUT_BYTES that this call fills from
 * luts / norm_lut; with luts == norm_lut == NULL it already holds them (filled
 * by an earlier call or by the caller: bytes 0..767 = luts, then norm_lut).  */
#define SGCLIP_CROP_LUT_BYTES (768 + 768 * 4)
int sgclip_crop_preprocess(const uint8_t *frames_dev, size_t frames_bytes, const uint8_t *masks_dev,
                           size_t masks_bytes, const sgclip_crop_t *crops, sgclip_crop_t *crops_dev,
                           int crops_dev_ready, int n_crops, const uint8_t *luts, const float *norm_lut,
                           void *lut_dev, int S, int layout, int patch, int kpad, int out_dtype,
                           void *out_dev, void *stream);

/* ------------------------------------------------------------------------
 * (2) CLIP towers.
 *
 * Replace model.get_image_features / model.get_text_features
 * (laser/models/llava_clip_model_v3.py:63-82; transformers
 * models/clip/modeling_clip.py CLIPVisionModel.forward, CLIPTextModel.forward
 * and the two projections).  Weights live in two caller-owned device blobs per
 * tower (bf16 matrices, f32 vectors) whose layout this library defines:
 * query it with sgclip_tower_layout().
 * ---------------------------------------------------------------------- */
typedef struct sgclip_tower_desc {
    int32_t kind;    /* 0 = vision, 1 = text                                  */
    int32_t hidden;  /* d                                                      */
    int32_t ffn;     /* f                                                      */
    int32_t layers;
    int32_t heads;   /* head_dim must be 64                                    */
    int32_t tokens;  /* vision: patches+1; text: max positions                 */
    int32_t proj;    /* E                                                      */
    int32_t patch;   /* vision only                                            */
    int32_t image;   /* vision only                                            */
    int32_t vocab;   /* text only                                              */
    int32_t eos_id;  /* text only                                              */
    float ln_eps;
} sgclip_tower_desc_t;

/* tensor ids for sgclip_tower_layout */
enum {
    SGCLIP_T_PATCH_W = 0, /* bf16 [d, kpad]  (vision; kpad = 3*p*p rounded up to 64, zero padded) */
    SGCLIP_T_CLS = 1,     /* f32 [d]         (vision)                          */
    SGCLIP_T_POS = 2,     /* f32 [tokens, d]                                   */
    SGCLIP_T_PRE_LN_W = 3, SGCLIP_T_PRE_LN_B = 4, /* f32 [d] (vision)          */
    SGCLIP_T_LN1_W = 5, SGCLIP_T_LN1_B = 6,       /* per layer f32 [d]         */
    SGCLIP_T_QKV_W = 7,   /* per layer bf16 [3d, d] rows q | k | v             */
    SGCLIP_T_QKV_B = 8,   /* per layer f32 [3d]                                */
    SGCLIP_T_O_W = 9, SGCLIP_T_O_B = 10,
    SGCLIP_T_LN2_W = 11, SGCLIP_T_LN2_B = 12,
    SGCLIP_T_FC1_W = 13, SGCLIP_T_FC1_B = 14, SGCLIP_T_FC2_W = 15, SGCLIP_T_FC2_B = 16,
    SGCLIP_T_POST_LN_W = 17, SGCLIP_T_POST_LN_B = 18, /* vision post_layernorm / text final_layer_norm */
    SGCLIP_T_PROJ_W = 19, /* f32 [E, d]  (the projection runs at fp32-grade precision) */
    SGCLIP_T_TOK_EMB = 20, /* f32 [vocab, d] (text)                            */
    SGCLIP_T_COUNT = 21
};

/* Element offset and element count of tensor `id` (layer index for per-layer
 * tensors, else 0) inside the bf16 blob (*is_bf16 = 1) or the f32 blob.
 * id = -1 returns the total element counts: *offset = bf16 elements, *count = f32 elements. */
int sgclip_tower_layout(const sgclip_tower_desc_t *desc, int id, int layer, int64_t *offset, int64_t *count,
                        int *is_bf16);
int sgclip_patch_kpad(int patch);

size_t sgclip_vit_workspace_bytes(const sgclip_tower_desc_t *desc, int n_images);
/* patches_dev: bf16 [n_images * P, kpad] as written by sgclip_crop_preprocess
 * with SGCLIP_LAYOUT_PATCHES; out_dev: f32 [n_images, E] un-normalised features. */
int sgclip_vit_encode(const sgclip_tower_desc_t *desc, const void *w_bf16_dev, const float *w_f32_dev,
                      const void *patches_dev, int n_images, void *workspace_dev, size_t workspace_bytes,
                      float *out_dev, void *stream);

size_t sgclip_text_workspace_bytes(const sgclip_tower_desc_t *desc, int n_texts, int seq_len);
/* ids_dev / attn_mask_dev: int32 [n_texts, seq_len] (mask may be NULL = all ones);
 * out_dev: f32 [n_texts, E]. */
int sgclip_text_encode(const sgclip_tower_desc_t *desc, const void *w_bf16_dev, const float *w_f32_dev,
                       const int32_t *ids_dev, const int32_t *attn_mask_dev, int n_texts, int seq_len,
                       void *workspace_dev, size_t workspace_bytes, float *out_dev, void *stream);

/* ------------------------------------------------------------------------
 * (3) similarity + softmax / sigmoid.
 *
 * Replaces clip_sim (laser/models/llava_clip_model_v3.py:41-47), its inline twin
 * for pairs (:451-455) and the .softmax(dim=0) / .sigmoid() that follow
 * (:275, :327, :459-461): out[k][n] = f(exp_logit_scale * <text_k, img_n> /
 * (|text_k| |img_n|)), f applied along k independently for every column n.
 * ---------------------------------------------------------------------- */
#define SGCLIP_SIM_LOGIT 0
#define SGCLIP_SIM_SOFTMAX 1
#define SGCLIP_SIM_SIGMOID 2
size_t sgclip_similarity_workspace_bytes(int K, int N);
int sgclip_similarity(const float *text_dev, const float *img_dev, int K, int N, int E, float exp_logit_scale,
                      int mode, float *out_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* The same computation for callers that cannot allocate device memory or pass a workspace (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * The inverse row norms that sgclip_similarity keeps in its workspace live in a static device array of
 * sgclip_similarity_static_max() floats (65 536 = 256 KiB, held on every GPU the library runs on until the
 * process ends): K + N beyond that returns SGCLIP_ENOSPC.  The call allocates and frees nothing.  All calls on
 * one GPU share the array: two calls that overlap on one GPU (different streams or threads) race; calls on
 * one stream are ordered and safe.  The current HIP device must be the one the pointers live on. */
int sgclip_similarity_static_max(void);
int sgclip_similarity_static(const float *text_dev, const float *img_dev, int K, int N, int E, float exp_logit_scale,
                             int mode, float *out_dev, void *stream);

/* ------------------------------------------------------------------------
 * (4) per-object mean of category probabilities over frames.
 *
 * Replaces the Python accumulation + torch.mean of
 * laser/models/llava_clip_model_v3.py:296-311: out[k][g] = mean of
 * probs[k][n] over the columns n with seg_ids[n] == g, summed in column order.
 * ---------------------------------------------------------------------- */
int sgclip_segment_mean(const float *probs_dev, const int32_t *seg_ids_dev, int K, int N, int G, float *out_dev,
                        void *stream);

/* ------------------------------------------------------------------------
 * (5) Data formats either side of the path.
 *
 * sgclip_topk_columns: for every column n of probs[K][N] (row stride ld) the k
 * best rows, best first, ranked by (value, tie_rank[row]) descending - the order
 * sorted((p, name), reverse=True) gives in compute_metrics
 * (scripts/eval_sgclip.py:112-115,132-133,161-164,170-171) when tie_rank is the
 * rank of the row's name in ascending string order (NULL: lower row first).
 * out_idx / out_val: [k][N]; rows beyond K hold -1 / 0.  k <= 16.
 *
 * sgclip_mask_unpack: bit-packed masks (rows padded to whole bytes, most
 * significant bit first = numpy.packbits(mask, axis=-1)) -> byte-per-pixel
 * masks at out_dev + out_off (the layout sgclip_crop_preprocess reads), and per
 * mask boxes_dev[i] = {x1, y1, x2, y2, count}: the inclusive tight box of
 * bitmasks2bboxes (laser/loading.py:84-104) and the pixel count of the
 * minimum-size filter (laser/training/llava_dataset.py:135); an empty mask
 * gives {INT_MAX, INT_MAX, -1, -1, 0}.  masks: host table; masks_dev: device
 * buffer of n_masks records that this call fills (H2D copy on `stream`).
 * ---------------------------------------------------------------------- */
int sgclip_topk_columns(const float *probs_dev, int64_t ld, int K, int N, const int32_t *tie_rank_dev, int k,
                        int32_t *out_idx_dev, float *out_val_dev, void *stream);

typedef struct sgclip_mask {
    int64_t packed_off; /* byte offset of the packed mask inside packed_dev   */
    int64_t out_off;    /* byte offset of the unpacked u8[H,W] mask inside out_dev */
    int32_t H, W;
} sgclip_mask_t; /* 24 bytes */
int sgclip_mask_unpack(const uint8_t *packed_dev, size_t packed_bytes, const sgclip_mask_t *masks,
                       sgclip_mask_t *masks_dev, int n_masks, uint8_t *out_dev, size_t out_bytes,
                       int32_t *boxes_dev, void *stream);

/* ------------------------------------------------------------------------
 * Building blocks, exported for unit tests and profiling.
 * ---------------------------------------------------------------------- */
#define SGCLIP_EPI_BIAS_BF16 0      /* C_bf16 = A W^T + bias                   */
#define SGCLIP_EPI_BIAS_GELU_BF16 1 /* C_bf16 = quick_gelu(A W^T + bias)       */
#define SGCLIP_EPI_BIAS_RESID_F32 2 /* C_f32 = R + A W^T + bias; R = aux (f32 [M, N], row stride ldc) or, aux NULL, C itself */
#define SGCLIP_EPI_PATCH_F32 3      /* C_f32[(m/P)*T + 1 + m%P] = A W^T + aux[(1 + m%P)]  (aux = pos emb, P = aux_rows) */
#define SGCLIP_EPI_F32 4            /* C_f32 = A W^T                           */
#define SGCLIP_EPI_F32_ATOMIC 5     /* C_f32 += A W^T, the K range split over several workgroups that add their
                                     * partial tiles with fp32 atomics (weight gradients: few tiles, very long K) */
/* The two halves of the MLP activation of the train step (modeling_clip.py:346-350 under autograd): */
#define SGCLIP_EPI_BIAS_GELU_BOTH_BF16 6   /* C_bf16 = u = A W^T + bias AND aux (bf16 [M, N], row stride ldc, WRITTEN)
                                            * = quick_gelu(u): the pre-activation the backward pass keeps and fc2's input */
#define SGCLIP_EPI_BIAS_MULGELUGRAD_BF16 7 /* C_bf16 = (A W^T + bias) * quick_gelu'(u), u = aux (bf16 [M, N], row stride
                                            * ldc): the data gradient through fc2 and the activation in one pass */
/* A: bf16 [M, K] (lda), W: bf16 [N, K] (ldw), C: [.., N] (ldc); K % 64 == 0.  aux_dev: see the epilogue. */
int sgclip_gemm_bf16(const void *A_dev, int lda, const void *W_dev, int ldw, void *C_dev, int ldc, int M, int N,
                     int K, int epilogue, const float *bias_dev, const void *aux_dev, int aux_rows,
                     void *stream);
/* sgclip_gemm_bf16 with SGCLIP_EPI_BIAS_MULGELUGRAD_BF16 (u_dev = its aux) that also accumulates colsum[n] += sum_m C[m][n]
 * (f32 [N], caller zeroes; NULL = none): C is the gradient at fc1's output, its column sums fc1's bias gradient. */
int sgclip_gemm_mulgelugrad_bf16(const void *A_dev, int lda, const void *W_dev, int ldw, void *C_dev, int ldc, int M, int N, int K,
                                 const float *bias_dev, const void *u_dev, float *colsum_dev, void *stream);
/* y[rows, out] = x[rows, in] * w[out, in]^T for fp32 operands, each split into bf16
 * hi + lo halves on the MFMA (three products), error ~2^-16 relative; in % 16 == 0. */
int sgclip_linear_f32(const float *x_dev, const float *w_dev, int rows, int out_features, int in_features,
                      float *y_dev, void *stream);
/* The linear layers and the attention of an encoder layer in the reference's own fp32 arithmetic
 * (nn.Linear / softmax attention as transformers models/clip/modeling_clip.py:269-277, 298-350 runs them under
 * laser/models/llava_clip_model_v3.py:77-83): exact fp32 products with fp32 accumulation on the fp32 matrix
 * cores.  Used to re-score the few crops on which the training-mode pair selection hangs
 * (llava_clip_model_v3.py:303-314, 347-386) when the bf16 towers leave the ranking undecided.
 * y = epi(x W^T + bias), x f32 [rows, in] (row stride ldx), W f32 [out, in] (ldw), bias f32 [out] or NULL,
 * y f32 (row stride ldy); in % 8 == 0, operand rows 16-byte aligned.  mode:                            */
#define SGCLIP_F32_PLAIN 0 /* y[m] = x W^T + bias                                                      */
#define SGCLIP_F32_GELU 1  /* y[m] = quick_gelu(x W^T + bias)                                          */
#define SGCLIP_F32_RESID 2 /* y[m] = aux[m] + x W^T + bias (aux f32, row stride ldy; may be y itself)  */
#define SGCLIP_F32_PATCH 3 /* y[(m / aux_rows) * out_tokens + 1 + m % aux_rows] = x W^T + aux[1 + m % aux_rows]:
                            * patch embedding + position rows (aux = position table, row stride ldy)   */
int sgclip_linear_exact_f32(const float *x_dev, int64_t ldx, const float *w_dev, int64_t ldw, const float *bias_dev,
                            int rows, int out_features, int in_features, int mode, const float *aux_dev,
                            int aux_rows, int out_tokens, float *y_dev, int64_t ldy, void *stream);
/* qkv f32 [n_seq*T, 3d] -> out f32 [n_seq*T, d]: softmax(q k^T / 8 + mask) v per head of 64, all in fp32;
 * causal and key_mask_dev (int32 [n_seq, T], NULL = all ones) as in sgclip_attention.                   */
int sgclip_attention_f32(const float *qkv_dev, float *out_dev, int n_seq, int T, int heads, int causal,
                         const int32_t *key_mask_dev, void *stream);
/* x[k][n] <- scale * x[k][n], then (do_softmax) softmax over k for every column n, in place: the tail of clip_sim
 * and the .softmax(dim=0) behind it (llava_clip_model_v3.py:41-47, :275) for logits already formed in fp32. */
int sgclip_scale_softmax_columns(float *x_dev, int K, int N, float scale, int do_softmax, void *stream);
/* y[r] = LayerNorm(x[row(r)]) * w + b; row(r) = row_idx_dev ? row_idx_dev[r] : r;
 * x f32 with row stride x_stride elements; y bf16 or f32, contiguous rows.   */
int sgclip_layernorm(const float *x_dev, int64_t x_stride, const int32_t *row_idx_dev, int rows, int d,
                     const float *w_dev, const float *b_dev, float eps, int out_dtype, void *y_dev,
                     void *stream);
/* qkv: bf16 [n_seq*T, 3d]; out: bf16 [n_seq*T, d]; softmax(q k^T / 8 + mask) v per head. */
int sgclip_attention(const void *qkv_dev, void *out_dev, int n_seq, int T, int heads, int causal,
                     const int32_t *key_mask_dev, void *stream);
/* The same attention for ONE query per sequence, the row q_rows_dev[s] (global row index into qkv):
 * out: bf16 [n_seq, d].  Used for the last encoder layer, of which only the pooled token
 * (class token / end-of-text token) reaches the output (modeling_clip.py:650-651, 574-581). */
int sgclip_attention_pooled(const void *qkv_dev, const int32_t *q_rows_dev, void *out_dev, int n_seq, int T,
                            int heads, int causal, const int32_t *key_mask_dev, void *stream);
/* Fused q/k/v projection + attention of one encoder layer, for unmasked sequences of T <= 128 tokens
 * (CLIPAttention.forward, modeling_clip.py:298-335, minus its out_proj): x bf16 [n_seq*T, d] (the
 * layer-norm output, row stride ldx), w_qkv bf16 [3d, d] (q, k, v rows stacked, row stride ldw),
 * bias f32 [3d]; out bf16 [n_seq*T, d] = concat_heads(softmax(q k^T / 8) v).  Same result as
 * sgclip_gemm_bf16(BIAS_BF16) followed by sgclip_attention, without the [rows, 3d] activation in HBM.
 * sgclip_qkv_attention_supported says whether (T, heads) is in range (1 = yes).                   */
int sgclip_qkv_attention_supported(int T, int heads);
int sgclip_qkv_attention(const void *x_dev, int ldx, const void *w_qkv_dev, int ldw, const float *bias_dev,
                         void *out_dev, int n_seq, int T, int heads, void *stream);

/* ------------------------------------------------------------------------
 * (6) Adjoints of the kernels above, for the train step
 * (scripts/train_sgclip.py:682-694 runs forward() under autograd and :877-890
 * back-propagates the loss into the CLIP weights).  The matrix products of the
 * backward pass (data gradient dY W, weight gradient dY^T X) run on
 * sgclip_gemm_bf16 with transposed operands; these entry points supply the
 * transposes, reductions and the non-GEMM adjoints.
 * ---------------------------------------------------------------------- */
/* C f32 [M, N] += A^T B for A bf16 [Kd, M] (row stride lda) and B bf16 [Kd, N] (ldb), both row major with the
 * reduction index as the row: the weight gradient dW = dY^T X of a linear layer (torch.autograd's
 * grad of nn.Linear reached from loss.backward(), scripts/train_sgclip.py:884) straight from the
 * activations, no transposed copies.  M, N multiples of 8; the reduction is split over workgroups
 * that add atomically, so repeated calls accumulate (the caller zeroes C once). */
int sgclip_gemm_tn_bf16(const void *A_dev, int lda, const void *B_dev, int ldb, float *C_dev, int ldc, int M, int N,
                        int Kd, void *stream);
/* out[c][r] = in[r][c], bf16; in [rows, cols] (ld_in), out [cols, ld_out], ld_out >= rows,
 * columns rows..ld_out-1 zero-filled (pads the reduction dimension of a weight gradient). */
int sgclip_transpose_bf16(const void *in_dev, int64_t ld_in, int rows, int cols, void *out_dev, int64_t ld_out,
                          void *stream);
/* out[n] += sum_m in[m][n] (bias gradients); in bf16 or f32 [M, N] (ld); out f32, caller zeroes it. */
int sgclip_colsum(const void *in_dev, int in_dtype, int64_t ld, int M, int N, float *out_dev, void *stream);
/* g = u sigmoid(1.702 u) (modeling_clip.py quick_gelu activation) and du = dg g'(u); bf16, n elements. */
int sgclip_quick_gelu(const void *u_dev, void *g_dev, int64_t n, void *stream);
int sgclip_quick_gelu_bwd(const void *u_dev, const void *dg_dev, void *du_dev, int64_t n, void *stream);
int sgclip_cast_f32_bf16(const float *in_dev, void *out_dev, int64_t n, void *stream);
/* out bf16 [M, N] = in f32 [M, N] and colsum[n] += sum_m in[m][n] in one pass (a gradient cast for the bf16 products
 * that is also a bias gradient); contiguous rows, N % 8 == 0, N <= 2048; the caller zeroes colsum. */
int sgclip_cast_colsum_f32_bf16(const float *in_dev, void *out_dev, int M, int N, float *colsum_dev, void *stream);
/* LayerNorm adjoint.  x f32 [rows, d] (row stride x_stride), dy bf16 or f32 [rows, d];
 * dx f32 [rows, d] (accumulate_dx: added to what dx holds); dw, db f32 [d], accumulated (caller zeroes). */
int sgclip_layernorm_bwd(const float *x_dev, int64_t x_stride, int rows, int d, const float *w_dev, float eps,
                         const void *dy_dev, int dy_dtype, float *dx_dev, int accumulate_dx, float *dw_dev,
                         float *db_dev, void *stream);
/* Adjoint of sgclip_attention: qkv bf16 [n_seq*T, 3d], dout bf16 [n_seq*T, d] -> dqkv bf16 [n_seq*T, 3d];
 * the probabilities are recomputed, nothing is saved by the forward.  T <= 128 runs whole sequences per
 * workgroup, longer ones (up to ~6 000 tokens) a blocked kernel with recomputed row statistics. */
int sgclip_attention_bwd(const void *qkv_dev, const void *dout_dev, void *dqkv_dev, int n_seq, int T, int heads,
                         int causal, const int32_t *key_mask_dev, void *stream);
/* The same, and dbias_dev f32 [3d] += the column sums of dqkv: the gradient of the fused q | k | v bias, which
 * autograd forms as a sum over the rows of the projections' output gradient (nn.Linear under loss.backward(),
 * scripts/train_sgclip.py:884).  For T <= 128 the sums leave the attention kernel itself (no pass over dqkv);
 * the caller zeroes or pre-loads dbias_dev. */
int sgclip_attention_bwd_bias(const void *qkv_dev, const void *dout_dev, void *dqkv_dev, float *dbias_dev, int n_seq, int T,
                              int heads, int causal, const int32_t *key_mask_dev, void *stream);
/* C[M][N] (+)= alpha op(A) op(B), fp32; op(A) = A [M,K] or, transA, A^T with A [K,M]; likewise B.
 * For the small fp32 tail (projection, similarity) where bf16 would cost accuracy. */
int sgclip_matmul_f32(const float *A_dev, int64_t lda, int transA, const float *B_dev, int64_t ldb, int transB,
                      float *C_dev, int64_t ldc, int M, int N, int K, float alpha, int accumulate, void *stream);
/* *out += <a, b> over n fp32 elements (caller zeroes out). */
int sgclip_dot_f32(const float *a_dev, const float *b_dev, int64_t n, float *out_dev, void *stream);
/* dlogits [K][N] from dprobs: softmax over k per column n, sigmoid, or copy (mode = SGCLIP_SIM_*). */
int sgclip_prob_bwd(const float *probs_dev, const float *dprobs_dev, int K, int N, int mode, float *dlogits_dev,
                    void *stream);
/* xh = x / |x| per row: dx = (dxh - xh <xh, dxh>) / |x|; [rows, E] f32. */
int sgclip_normalize_bwd(const float *x_dev, const float *dxhat_dev, int rows, int E, float *dx_dev, void *stream);
/* xhat = x / |x| per row, [rows, E] f32 (the normalisation inside clip_sim, llava_clip_model_v3.py:43-44). */
int sgclip_normalize_rows(const float *x_dev, int rows, int E, float *xhat_dev, void *stream);
/* Adjoint of sgclip_segment_mean: dprobs[k][n] = dmean[k][seg[n]] * inv_count[seg[n]]. */
int sgclip_segment_mean_bwd(const float *dmean_dev, const int32_t *seg_ids_dev, const float *inv_count_dev, int K, int N,
                            int G, float *dprobs_dev, void *stream);
/* Embeddings of the training forward from fp32 tables: x[s*T+t] = table[ids[s*T+t]] + pos[t] (text,
 * modeling_clip.py:190-207); x[s*T] = cls + pos[0] (class-token row of every image, :160-171). */
int sgclip_embed_text(const float *table_dev, const float *pos_dev, const int32_t *ids_dev, int n_texts, int T, int d,
                      float *x_dev, void *stream);
int sgclip_embed_cls(const float *cls_dev, const float *pos_dev, int n_images, int T, int d, float *x_dev, void *stream);
/* dst[idx[r]][:] += src[r][:]; f32 rows of d elements (embedding-table gradients). */
int sgclip_rows_scatter_add(const float *src_dev, const int32_t *idx_dev, int rows, int d, float *dst_dev,
                            void *stream);
/* One AdamW step on fp32 parameters (decoupled weight decay; step counts from 1). */
int sgclip_adamw(float *param_dev, const float *grad_dev, float *m_dev, float *v_dev, int64_t n, float lr,
                 float beta1, float beta2, float eps, float weight_decay, int step, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SGCLIP_H */
