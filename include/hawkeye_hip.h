/*
 * libhawkeye_hip.so -- C ABI of the MI355X (gfx950) encode_img kernels.
 *
 * The reference (yellow-binary-tree/HawkEye) has no FFI: the "operators" on its
 * hot path are the PyTorch ops listed in SURVEY.md §2.1.  Each entry point
 * below replaces one group of those ops and cites the reference call sites it
 * stands in for.  Conventions:
 *   - the caller owns every buffer (inputs, outputs, workspace) and passes raw
 *     device pointers; the library never allocates, frees or synchronises;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*);
 *   - tensors are row-major; bf16 tensors and fp32 rows must be 16-byte aligned;
 *   - return value 0 = success, otherwise an HK_E* code; hk_last_error() holds
 *     the text of the last failure on the calling thread.
 */
#ifndef HAWKEYE_HIP_H
#define HAWKEYE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HK_OK 0
#define HK_EINVAL 1   /* bad shape / alignment / null pointer */
#define HK_ELAUNCH 2  /* hipGetLastError() after a launch was not hipSuccess */

#define HK_ACT_NONE 0
#define HK_ACT_GELU_ERF 1
/* SwiGLU on interleaved columns (row f1, modeling_llama.py:143-155 `down_proj(act_fn(gate_proj(x)) * up_proj(x))`): W holds the rows of
 * gate_proj and up_proj interleaved (gate_0, up_0, gate_1, up_1, ...), N counts the rows of W, and C is bf16 [M, N / 2] with
 * C[m, k] = silu(y[m, 2k]) * y[m, 2k + 1], y = A . W^T + bias -- the [M, N] gate|up matrix is never written.  Needs
 * N % 32 == 0, bf16 output, no residual / folded LayerNorm / row_part / column scale / split_k; c_rows addresses the N / 2-column output. */
#define HK_ACT_SWIGLU 2

#define HK_DT_BF16 0
#define HK_DT_F32 1

/* Row addressing shared by the GEMM operands: logical row m of an operand lives at
 *   base + (m / rows_per_batch) * batch_stride + (m % rows_per_batch) * ld   (elements)
 * rows_per_batch == 0 means "plain": base + m * ld.  This lets one launch address
 * the query rows [:, :96] of a [B,S,768] Q-Former buffer (Qformer.py:463) or
 * broadcast the [N,1024] position table over the batch (vit.py:307). */
typedef struct hk_rows {
    int64_t ld;
    int64_t rows_per_batch;
    int64_t batch_stride;
} hk_rows;

/* C = act((A . W^T + bias) * colscale) + residual
 *   A [M,K] bf16, W [N,K] bf16 (nn.Linear / flattened Conv3d weight layout), bias [N] fp32 or NULL,
 *   residual fp32 or bf16 (res_dtype) or NULL -- it may be C itself (in-place residual stream), C bf16 or
 *   fp32 (out_dtype).  fp32 accumulation on MFMA; the output is rounded once, after the residual add.
 *   Columns n < scale_cols are multiplied by `scale` after the bias (used to fold the
 *   attention scale into Q: vit.py:91, Qformer.py:257).
 * Replaces: F.linear qkv vit.py:87; proj vit.py:98 + residual vit.py:126; fc1+GELU vit.py:49-50;
 *   fc2 vit.py:52 + residual vit.py:127; Conv3d patch-embed vit.py:158 + pos table vit.py:307;
 *   BERT query/key/value/dense Qformer.py:199-211,314,388,402; llama_proj hawkeye_it.py:269.
 * Requires K % 64 == 0, N % 16 == 0. */
typedef struct hk_gemm_desc {
    const void* A;
    const void* W;
    const float* bias;
    const void* residual;
    void* C;
    int32_t M, N, K;
    hk_rows a_rows;
    int64_t ldw;
    hk_rows c_rows;
    hk_rows r_rows;
    int32_t act;        /* HK_ACT_* */
    int32_t out_dtype;  /* HK_DT_* */
    int32_t res_dtype;  /* HK_DT_* of `residual` */
    int32_t scale_cols;
    float scale;
    /* LayerNorm of A folded into the GEMM (vit.py:125-127: norm1 -> attn.qkv, norm2 -> mlp.fc1), both NULL when unused.
     * With row_stats[m] = (mean_m, rstd_m) of row m of A (hk_row_stats), W holding gamma[k] * weight[n,k] rounded to
     * bf16, col_c[n] = sum_k W[n,k] (of the rounded values) and bias[n] = linear bias + sum_k beta[k] * weight[n,k]:
     *   C = act((rstd_m * (A . W^T)[m,n] - rstd_m * mean_m * col_c[n] + bias[n]) * colscale)
     * = act((LayerNorm(A) . weight^T + linear bias) * colscale) without ever writing LayerNorm(A).
     * Needs bias, bf16 output, no residual. */
    const float* row_stats; /* [ceil(M / 256) * 256][2] fp32: whole 256-row tiles are read (rows >= M: any value) */
    const float* col_c;     /* [N] fp32 */
    /* Optional output for the LayerNorm that FOLLOWS this GEMM (vit.py:126-127: proj / fc2 + residual, then norm2 /
     * the next block's norm1): row_part[m][g] = (sum, sum of squares) over columns 64 g .. 64 g + 63 of row m of C,
     * taken on the bf16 values as stored (the matrix the consumer reads); hk_row_stats_finalize turns them into (mean, rstd).
     * Needs N % 128 == 0 and the bf16 in-place residual epilogue; NULL when unused. */
    float* row_part;        /* [M][N / 64][2] fp32 */
    /* Split-K for problems with few output tiles and a long contraction (the Q-Former's 3072 -> 768 dense layers on a
     * few hundred rows: 12 workgroups walking 48 K-tiles one after the other).  split_k > 1: slice z < split_k
     * multiplies K-tiles [z * ceil(KT / split_k), ...) and writes its fp32 partial product to C + z * split_stride
     * (elements); slice 0 carries bias and residual.  The consumer sums the slices (hk_layernorm_parts).  Needs fp32
     * output, no activation, no folded LayerNorm, no row_part.  0 or 1 = off. */
    int32_t split_k;
    /* 1: W is stored in 64 x 64 blocks, [N/64][K/64][64 rows][64 elements] bf16 (element k of row n at
     * ((n / 64) * (K / 64) + k / 64) * 4096 + (n % 64) * 64 + k % 64; ldw is ignored): a workgroup's weight tile is whole 8 KB
     * pieces and its walk over K contiguous, which is what weights that arrive cold from HBM want (the LLM's, read once per
     * pass).  N and K multiples of 64.  0: row-major with row stride ldw. */
    int32_t w_tiled;
    int64_t split_stride;
} hk_gemm_desc;

int hk_gemm_bf16(const hk_gemm_desc* d, void* stream);

/* Fused softmax(Q K^T + key_bias [+ causal mask]) V, head size 64 or 128, fp32 online softmax, bf16 in/out.
 *   Q [B,Nq,H,64], K/V [B,Nk,H,64], O [B,Nq,H,64] addressed through element strides
 *   (token stride ld*, batch stride bs*; head h at column h*64).  Q must already carry
 *   softmax_scale * log2(e) (folded into the producing GEMM); key_bias is an optional
 *   fp32 [B,Nk] additive term in natural-log units (the -10000 mask of Qformer.py:834).
 * Replaces: vit.py:91-97 (q*scale, q k^T, softmax, @v, transpose/reshape) and
 *   Qformer.py:218-281 (scores, /sqrt(d), +mask, softmax, context, permute). */
typedef struct hk_attn_desc {
    const void* Q;
    const void* K;
    const void* V;
    void* O;
    const float* key_bias;
    int32_t B, H, Nq, Nk;
    int64_t ldq, ldk, ldv, ldo;
    int64_t bsq, bsk, bsv, bso;
    int32_t head_dim; /* 0 or 64 (ViT, Q-Former), 128 (LLaMA, row f1) */
    int32_t causal;   /* != 0: query q attends keys <= q + (Nk - Nq)  (modeling_llama.py:25-47) */
    /* Key split for SMALL calls (B = 1 eval calls: 12..304 workgroups walking all keys one after the other): the keys
     * are cut into key_splits slices of whole 64-key tiles, each (batch, head, query block, slice) is a workgroup that
     * writes its normalised partial output and the rows' log2-sum-exp, and a second launch of the same call combines
     * them (weights 2^(lse_s - max), slices in index order).  0 or 1 = off.  Head size 64, no key_bias, not causal.
     * o_part: bf16 [B * key_splits * Nq * H * 64]; lse_part: fp32 [B * key_splits * H * Nq]. */
    int32_t key_splits;
    void* o_part;
    float* lse_part;
} hk_attn_desc;

int hk_flash_attn_bf16(const hk_attn_desc* d, void* stream);

/* y = LayerNorm(x [+ add]) * gamma + beta over the last dim (D % 4 == 0, D <= 4096), fp32 statistics.
 *   x: in_dtype (bf16 | fp32) [rows, D] plain; add: optional fp32 [rows, D] summed before the norm
 *   (BERT post-LN, Qformer.py:317,405); out_bf16 and/or out_f32 may be NULL; out_rows (may be NULL =
 *   plain [rows, D]) maps output row m like the GEMM operands, so a [B*nq, d] result can land in
 *   rows [:, :nq] of a [B, S, d] buffer (the torch.cat of Qformer.py:495).
 * Replaces: nn.LayerNorm vit.py:108,114 (eps 1e-6); vision_layernorm hawkeye_it.py:234 (eps 1e-12);
 *   BertSelfOutput/BertOutput LayerNorm Qformer.py:317,405. */
int hk_layernorm(const void* x, int32_t in_dtype, const float* add, const float* gamma, const float* beta,
                 void* out_bf16, float* out_f32, const hk_rows* out_rows, int64_t rows, int32_t D, float eps,
                 void* stream);

/* hk_layernorm with n_add fp32 addends add[a * add_stride + row * D + col], a < n_add, summed in index order before the
 * norm: the partial products of a split-K GEMM (hk_gemm_desc.split_k) are reduced here instead of in a pass of their
 * own, so dense -> (+ residual) -> LayerNorm (Qformer.py:314-317, 402-405) stays two launches however K is split. */
int hk_layernorm_parts(const void* x, int32_t in_dtype, const float* add, int32_t n_add, int64_t add_stride,
                       const float* gamma, const float* beta, void* out_bf16, float* out_f32, const hk_rows* out_rows,
                       int64_t rows, int32_t D, float eps, void* stream);

/* Row statistics of a bf16 matrix: stats[m] = (mean_m, 1 / sqrt(var_m + eps)), two-pass fp32 over the row held in
 * registers (the same arithmetic as hk_layernorm's).  Feeds hk_gemm_desc.row_stats: the statistics half of
 * nn.LayerNorm vit.py:108,114 when the normalisation itself is folded into the consuming GEMM.
 * x [rows, D] plain, D % 8 == 0, D <= 2048; stats [rows][2] fp32. */
int hk_row_stats(const void* x_bf16, float* stats, int64_t rows, int32_t D, float eps, void* stream);

/* Tail of a split-K proj / fc2 (vit.py:126-127) on a small call: x (bf16 [rows, D], in place) += sum over a < n_parts of
 * parts[a * part_stride + row * D + col] (fp32 slices of hk_gemm_desc.split_k; slice 0 carries the bias), rounded once;
 * stats[row] = (mean, rstd) of the row as stored (NULL: not wanted) -- what hk_row_stats would return for it -- for the
 * GEMM that consumes x with its LayerNorm folded in.  D % 8 == 0, D <= 2048. */
int hk_residual_add_parts(void* x_bf16, const float* parts, int32_t n_parts, int64_t part_stride, float* stats, int64_t rows,
                          int32_t D, float eps, void* stream);

/* stats[m] = (mean, rstd) from the per-group sums a GEMM left in hk_gemm_desc.row_part: part [rows][groups][2],
 * D = 64 * groups columns per row; sums combined in float64, var = E[x^2] - mean^2 clamped at 0. */
int hk_row_stats_finalize(const float* part, float* stats, int64_t rows, int32_t groups, float eps, void* stream);

/* Patch gather for the per-frame 16x16 patch GEMM: clip [B,T,3,H,W] fp32 (the caller's
 * layout, tasks/test.py:490-492) -> A [B*T*(H/p)*(W/p), 3*p*p] bf16 with rows ordered (b,t,gy,gx) and
 * K ordered (c,ky,kx) -- the flattening of Conv3d weight [D,3,1,p,p] (vit.py:146-150,158). */
int hk_patchify_bf16(const float* clip, void* out, int32_t B, int32_t T, int32_t H, int32_t W, int32_t patch,
                     void* stream);

/* Row f3: the eval pipeline's frame transform fused into the patch gather.  frames uint8 [B,T,H,W,3] (decoded,
 * resized and cropped frames as decord/PIL hand them over, dataset/video_utils.py:154-176) ->
 * ((x / 255) - mean[c]) / std[c] in fp32, each step rounded as the reference's ToTorchFormatTensor
 * (dataset/video_transforms.py:367-389) and GroupNormalize (:94-107) round it -> the same bf16 [rows, 3*p*p]
 * operand hk_patchify_bf16 produces from the normalised fp32 clip.  mean3/std3 are HOST pointers to 3 floats. */
int hk_patchify_u8(const uint8_t* frames, void* out, int32_t B, int32_t T, int32_t H, int32_t W, int32_t patch,
                   const float* mean3, const float* std3, void* stream);

/* Row f4: fixed sinusoid position table on the device, fp32 [T*L, D].  get_sinusoid_encoding_table
 * (models/blip2/vit.py:163-194): float64 angles, sin/cos, cast to fp32; if ckpt_frames > 0 and != T the
 * [ckpt_frames, L, D] table is linearly interpolated along the frame axis (align_corners=False, vit.py:186). */
int hk_pos_embed(float* out, int32_t T, int32_t L, int32_t D, int32_t ckpt_frames, void* stream);

/* ---- row f1: LLaMA / Vicuna prefill over the spliced [prompt | 96 video tokens | prompt] sequence ---- */

/* LlamaRMSNorm (models/blip2/modeling_llama.py:57-75): out = weight * x * rsqrt(mean(x^2) + eps), fp32 statistics.
 * x [rows, D] bf16 or fp32 (in_dtype), weight fp32 [D], out bf16 [rows, D]; D a multiple of 256. */
int hk_rmsnorm(const void* x, int32_t in_dtype, const float* weight, void* out_bf16, int64_t rows, int32_t D,
               float eps, void* stream);

/* The residual add of o_proj / down_proj and the RMSNorm that follows it (modeling_llama.py:297-307, 57-75) on split-K slices:
 * x[rows, D] fp32 (out: the new residual stream) = sum over the n_parts slices of `parts` (slice q at parts + q * part_stride; slice 0
 * of hk_gemm_bf16's split_k already carries the old residual stream), out_bf16 = RMSNorm(x) * weight.  Used for skinny prompts
 * (B = 1 eval calls), where o_proj / down_proj are a few workgroups walking 64-172 K-tiles unless K is split. */
int hk_rmsnorm_add_parts(const float* parts, int32_t n_parts, int64_t part_stride, float* x, const float* weight,
                         void* out_bf16, int64_t rows, int32_t D, float eps, void* stream);

/* apply_rotary_pos_emb (modeling_llama.py:108-123) in place on the q and k thirds of a fused bf16
 * [rows, 3*H*head_dim] buffer; cos/sin fp32 [table_rows, head_dim/2] (the first half of the reference's cached
 * table, :82-89, whose two halves are equal); position_ids int32 [rows] or NULL (= row % seq_len, the default
 * arange of LlamaModel.forward :526-533).  q is also multiplied by q_scale (softmax scale * log2 e). */
int hk_rope_bf16(void* qkv, const float* cos_table, const float* sin_table, const int32_t* position_ids,
                 int64_t rows, int32_t H, int32_t head_dim, int32_t seq_len, int32_t table_rows, float q_scale,
                 void* stream);

/* ---- decode: one new token per sequence and step (row f1 continued) --------------------------------------------------
 * What `generate` runs after the prefill (tasks/test.py:121-133, test_recursive_grounding.py:122-134 with do_sample=False,
 * num_beams=1): LlamaForCausalLM.forward on ONE token per sequence with past_key_values (modeling_llama.py:176-231,
 * 482-611, 744-773).  With 1-16 rows every projection streams its weights once: these entries are HBM-bound. */

/* y[M, N] = x[M, K] . W[N, K]^T for M <= 16 rows (nn.Linear without bias: q/k/v, o_proj, gate/up, down_proj, lm_head;
 * modeling_llama.py:134-140, 160-163, 622).  x bf16 (row stride ldx; fp32 with norm_weight), W bf16 row-major; out bf16 or fp32 (row stride ldo);
 * residual fp32 with out's layout or NULL, added before the store (may alias out: the residual stream, :297-307);
 * act HK_ACT_SWIGLU: W rows interleaved (gate_0, up_0, ...) and out[M, N/2] = silu(gate) * up in bf16, as hk_gemm_bf16. */
typedef struct hk_gemv_desc {
    const void* x;
    const void* W;
    void* out;
    const float* residual;
    int32_t M, N, K;
    int32_t out_f32;
    int64_t ldx, ldo;
    int32_t act;
    float norm_eps;
    /* fused LlamaRMSNorm (modeling_llama.py:57-75) of the input, M <= 4: x is then the fp32 residual stream [M, K] and the
     * operand is bf16(norm_weight * (x * rsqrt(mean(x^2) + norm_eps))), what hk_rmsnorm would have written; NULL = x is bf16 */
    const float* norm_weight;
    /* 1: W is stored in 16-row x 64-element blocks, [ceil(N/16)][K/64][2][64][8] bf16 -- block (r, b) holds rows 16 r .. 16 r + 15
     * (rows past N zero), elements 64 b .. 64 b + 63, as [half h][lane l = 16 q + i][8]: element 8 e of row 16 r + i at k =
     * 64 b + 16 q + 8 h + e.  A wave then streams one contiguous region (a decode-only copy of the weights: HBM is large). */
    int32_t w_tiled;
    int32_t reserved;      /* 0 */
} hk_gemv_desc;
int hk_gemv_bf16(const hk_gemv_desc* d, void* stream);

/* The new token's fused q|k|v row per sequence (bf16 [B, 3*H*head_dim]): rotary embedding on q and k at position
 * pos0[b] + pos_add (apply_rotary_pos_emb, modeling_llama.py:116-123; the position of generate's new token is
 * cumsum(attention_mask)[-1] - 1, :751-756), q * q_scale -> q_out bf16 [B, H*head_dim], k and v -> row `slot` of the caches
 * (bf16 [B, cache_rows, H*head_dim]; the torch.cat of past_key_value, :201-206). */
int hk_decode_rope_append(const void* qkv, const float* cos_table, const float* sin_table, const int32_t* pos0,
                          int32_t pos_add, void* q_out, void* k_cache, void* v_cache, int32_t B, int32_t H,
                          int32_t head_dim, int32_t slot, int64_t cache_rows, int32_t table_rows, float q_scale,
                          void* stream);

/* Attention of one query row per (sequence, head) over cache rows [0, n_keys) (modeling_llama.py:195-224 with q_len = 1:
 * fp32 softmax, no causal term).  q as hk_decode_rope_append leaves it; key_bias fp32 [B, bias_stride] additive in the
 * natural-log domain (0 / -10000 from hk_qformer_key_bias for left-padded batches) or NULL; out bf16 [B, H*head_dim]. */
int hk_decode_attn_bf16(const void* q, const void* k_cache, const void* v_cache, const float* key_bias, void* out,
                        int32_t B, int32_t H, int32_t head_dim, int32_t n_keys, int64_t cache_rows, int64_t bias_stride,
                        int32_t key_splits, float* split_ws, void* stream);
/* key_splits > 1 with split_ws != NULL (fp32, B * H * min(key_splits, 16) * (head_dim + 2) elements, caller-owned): the keys of a
 * long cache are divided over that many workgroups per (sequence, head) and merged by a second launch -- B * H workgroups are all
 * the parallelism a decode step has otherwise.  Same result up to the order of the fp32 sums. */

/* hk_decode_rope_append + hk_decode_attn_bf16 in one launch (the workgroup of a (sequence, head) is the only reader of that
 * head's new cache row, so it writes the row itself): qkv as hk_decode_rope_append takes it, n_keys = slot + 1.  Same
 * results as the two calls. */
int hk_decode_attn_rope_bf16(const void* qkv, const float* cos_table, const float* sin_table, const int32_t* pos0,
                             int32_t pos_add, void* k_cache, void* v_cache, const float* key_bias, void* out, int32_t B,
                             int32_t H, int32_t head_dim, int32_t slot, int64_t cache_rows, int64_t bias_stride,
                             int32_t table_rows, float q_scale, int32_t key_splits, float* split_ws, void* stream);

/* Greedy next token (do_sample=False, num_beams=1): out[r] = index of the largest of logits[r, 0:V] (row stride ld),
 * the lowest index among equals. */
int hk_argmax_f32(const float* logits, int64_t* out, int32_t rows, int32_t V, int64_t ld, void* stream);

/* LlamaMLP gate (modeling_llama.py:139-140): out[rows, I] = silu(gate_up[:, :I]) * gate_up[:, I:], bf16. */
int hk_silu_mul_bf16(const void* gate_up, void* out, int64_t rows, int32_t I, void* stream);

/* embed_tokens (modeling_llama.py:444-455, num_frame_tokens = 0) fused with the splice of the video tokens into the
 * prompt (tasks/test.py:79-97; hawkeye_it.py:287-375): out fp32 [rows, D]; row r = table[src[r]] if src[r] >= 0,
 * else video[-src[r] - 1].  table bf16 [V, D]; video [*, D] bf16 or fp32 (video_dtype; may be NULL if unused). */
int hk_embed_splice(const int64_t* src, const void* table_bf16, const void* video, int32_t video_dtype, float* out,
                    int64_t rows, int32_t D, void* stream);

/* Q-Former embeddings: rows [0,nq) = query_tokens; rows [nq, nq+Lt) = word_emb[ids] + pos_emb[0..Lt);
 * LayerNorm(eps) -> out_f32 and out_bf16 [B, nq+Lt, d].  (Qformer.py:97-121; hawkeye_it.py:237-241) */
int hk_qformer_embed_ln(const float* query_tokens, const float* word_emb, const float* pos_emb,
                        const int64_t* input_ids, const float* gamma, const float* beta, float* out_f32,
                        void* out_bf16, int32_t B, int32_t nq, int32_t Lt, int32_t d, int32_t vocab, float eps,
                        void* stream);

/* Additive key bias of the Q-Former self-attention: out fp32 [B, nq+Lt]; columns [0,nq) = 0 (queries are never
 * masked, hawkeye_it.py:250), column nq+j = (1 - text_mask[b,j]) * -10000 (get_extended_attention_mask,
 * Qformer.py:746-835).  text_mask int64 [B,Lt] (NULL when Lt == 0).  Feeds hk_flash_attn_bf16's key_bias. */
int hk_qformer_key_bias(const int64_t* text_mask, float* out, int32_t B, int32_t nq, int32_t Lt, void* stream);

/* fp32 -> bf16 (round-to-nearest-even) copy of n elements; weight packing at load time. */
int hk_cast_f32_bf16(const float* x, void* out, int64_t n, void* stream);

/* ---- row f2: training of the Q-Former / query tokens / llama_proj (stage 3: frozen ViT, tasks/train_it.py:63-80) ----
 *
 * Dropout and drop-path (Qformer.py:120,271,315-316,403-404; timm drop_path) use a counter-based hash instead of a
 * stateful generator, so the backward regenerates every mask from (seed, site, element index) and none is stored:
 *   key  = fmix32(seed ^ fmix32(site * 0x9E3779B9 + 0x7F4A7C15))        (fmix32 = MurmurHash3's 32-bit finaliser)
 *   keep = fmix32(idx ^ key) >= (uint32)(p * 2^32) ; kept values are scaled by 1 / (1 - p)
 * idx = flat element index of the site's own tensor ([rows, D] hidden states; [B,H,Nq,Nk] attention probabilities;
 * the sample index for drop-path).  hk_dropout_keep_u32 returns fmix32(idx ^ key) on the host (test aid). */
typedef struct hk_dropout {
    float p;                 /* element dropout probability, 0 = off */
    float p_path;            /* drop-path probability (one draw per sample), 0 = off */
    uint32_t seed;
    uint32_t site;           /* site id of the element dropout */
    uint32_t path_site;      /* site id of the drop-path draw */
    int32_t rows_per_sample; /* rows of one sample (drop-path granularity) */
} hk_dropout;

uint32_t hk_dropout_keep_u32(uint32_t seed, uint32_t site, uint32_t idx);

/* Attention with dropout on the probabilities, head size 64, and its backward.
 *   forward:  O = dropout(softmax(Q K^T + key_bias)) V ; lse[b,h,q] = log2-domain logsumexp of row q (saved for backward)
 *   backward: dQ (gradient w.r.t. the UNscaled query: the kernel multiplies by softmax_scale), dK, dV in bf16;
 *             dsum fp32 [B,H,Nq] is workspace.  Deterministic: no atomics (a query-side and a key-side kernel).
 * Q must carry softmax_scale * log2(e) like hk_flash_attn_bf16.  O and dO share ldo/bso.  With o_f32 the row sums
 * dsum = sum_d dO.O use the unrounded output, which keeps dS exact where attention is diffuse (dP - dsum cancels).
 * Replaces: BertSelfAttention.forward in train mode, Qformer.py:218-281 (incl. nn.Dropout :271), and its autograd. */
typedef struct hk_attn_train_desc {
    const void* Q;
    const void* K;
    const void* V;
    void* O;
    float* lse;
    const float* key_bias;
    const void* dO;
    void* dQ;
    void* dK;
    void* dV;
    float* dsum;
    int32_t B, H, Nq, Nk;
    int64_t ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso;
    int64_t lddq, lddk, lddv, bsdq, bsdk, bsdv;
    float softmax_scale;
    float p_drop;
    uint32_t seed, site;
    float* o_f32;  /* optional fp32 copy of O, dense [B,Nq,H*64]: written by the forward, read by the backward for dsum */
} hk_attn_train_desc;

int hk_attn_train_fwd(const hk_attn_train_desc* d, void* stream);
int hk_attn_train_bwd(const hk_attn_train_desc* d, void* stream);

/* z = drop_path(dropout(t)) + res ; y = LayerNorm(z) * gamma + beta [; y = dropout_post(y)], D <= 1024.
 *   t fp32 [rows, D] plain or NULL; res fp32 row-mapped (res_rows, NULL = plain) or NULL; z_out fp32 [rows, D] or NULL
 *   (saved for hk_layernorm_bwd); y_f32 / y_bf16 row-mapped by out_rows.  drop / post_drop may be NULL (eval).
 * Replaces: BertSelfOutput / BertOutput in train mode (dense output -> dropout -> drop_path -> LayerNorm(h + input),
 *   Qformer.py:313-318, 401-406) and BertEmbeddings' LayerNorm -> dropout (Qformer.py:119-120; post_drop). */
int hk_drop_add_ln(const float* t, const float* res, const hk_rows* res_rows, const hk_dropout* drop,
                   const hk_dropout* post_drop, const float* gamma, const float* beta, float* z_out, float* y_f32,
                   void* y_bf16, const hk_rows* out_rows, int64_t rows, int32_t D, float eps, void* stream);

/* Backward of hk_drop_add_ln.  dy fp32 row-mapped (dy_rows); z = the saved pre-norm rows.
 *   dz fp32 [rows, D] = gradient of z (= gradient of the residual input); dt_bf16 [rows, D] = dz through the
 *   drop-path / dropout mask (= gradient of t, the next GEMM operand); dgamma/dbeta fp32 [D] (dbeta == dgamma + D);
 *   ws: fp32 workspace of HK_LNBWD_MAX_BLOCKS * 2 * D floats (per-workgroup partial sums; deterministic). */
#define HK_LNBWD_MAX_BLOCKS 256
int hk_layernorm_bwd(const float* dy, const hk_rows* dy_rows, const float* z, const float* gamma,
                     const hk_dropout* drop, const hk_dropout* post_drop, float* dz, void* dt_bf16, float* dgamma,
                     float* dbeta, float* ws, int64_t rows, int32_t D, float eps, void* stream);

/* out[c] = sum over rows of x[r][c] (bias gradients).  x bf16 or fp32 [R, C] with row stride ld; ws: optional fp32
 * workspace of HK_COLSUM_MAX_SPLIT * C floats (enables a two-stage reduction for tall inputs). */
#define HK_COLSUM_MAX_SPLIT 64
int hk_colsum(const void* x, int32_t dtype, int64_t ld, int64_t R, int32_t C, float* out, float* ws, void* stream);

/* hk_colsum for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  No counterpart in the reference.  Same result, bit for bit, as hk_colsum with a workspace; the
 * [HK_COLSUM_MAX_SPLIT, C] partial sums live in a static device array of the library (2.25 MiB per GPU, held until
 * the process ends) instead of a caller workspace.  C <= HK_COLSUM_STATIC_MAX_C, otherwise HK_EINVAL.  Calls no
 * hipMalloc / hipFree.  Every call on a GPU shares that GPU's array: calls that overlap in time on one GPU (two
 * streams, two threads) race -- serialise them, or give the others hk_colsum with a workspace of their own.  The
 * array's device address is looked up on the first call on each device (hipGetSymbolAddress; the device that is
 * current at the call must be the one `stream` belongs to) -- make that first call outside graph capture. */
#define HK_COLSUM_STATIC_MAX_C 9216
int hk_colsum_static(const void* x, int32_t dtype, int64_t ld, int64_t R, int32_t C, float* out, void* stream);

/* out bf16 [C, ldo]: out[c][r] = x[r][c] for r < R, 0 for R <= r < roundup(R, 64).  x bf16 or fp32, rows mapped by
 * x_rows (NULL = plain [R, C]).  Makes the contraction-contiguous operands of the backward GEMMs:
 *   dgrad  dX = dY . W      -> hk_gemm_bf16(A = dY,   W = W^T  [in, out])
 *   wgrad  dW = dY^T . X    -> hk_gemm_bf16(A = dY^T, W = X^T  [in, rows]) */
int hk_transpose_bf16(const void* x, int32_t in_dtype, const hk_rows* x_rows, int64_t R, int32_t C, void* out,
                      int64_t ldo, void* stream);

/* Many fp32 [R, C] (row stride ld) -> bf16 [C, >= R] (row stride ldo) transposes in one launch: the W^T operands of
 * every trainable Linear after an optimizer step.  `jobs_dev` is a DEVICE array the caller fills once; job j owns the
 * workgroups [tile0, tile0 + ceil(R / 64) * ceil(C / 64)), tile0 ascending from 0; total_tiles = their sum.  Elements
 * out[c][r] with r >= R are not written. */
typedef struct hk_transpose_job {
    const void* x;
    void* out;
    int64_t ld, ldo;
    int32_t R, C, tile0, pad_;
} hk_transpose_job;
int hk_transpose_f32_bf16_batch(const hk_transpose_job* jobs_dev, int32_t n_jobs, int32_t total_tiles, void* stream);

/* Weight gradient of a Linear in one launch (what autograd computes for nn.Linear under tasks/train_it.py:63-80):
 *   dw[n][k] = sum_r dy[r][n] * x[r][k]   (fp32, row stride lddw, overwritten)
 *   db[n]    = sum_r dy[r][n]             (fp32, when db != NULL)
 * dy bf16 [R, N] plain rows of stride ldy; x bf16 [R, K] with rows mapped by x_rows (NULL = plain [R, K]).  The
 * contraction runs over the ROW index of both operands: tiles are staged row-major and read with gfx950's transposing
 * LDS read, so no transposed copy of either operand is made (it replaces hk_transpose_bf16 x 2 + hk_gemm_bf16 +
 * hk_colsum of round 2).  Any R >= 1; N and K multiples of 128.  Summation order is fixed: bit-reproducible. */
int hk_wgrad_bf16(const void* dy, int64_t ldy, const void* x, const hk_rows* x_rows, float* dw, int64_t lddw, float* db,
                  int64_t R, int32_t N, int32_t K, void* stream);

/* exact-erf GELU on bf16 and its backward du = dg * gelu'(u) (BertIntermediate, Qformer.py:387-390); n % 8 == 0. */
int hk_gelu_bf16(const void* u, void* g, int64_t n, void* stream);
int hk_gelu_bwd_bf16(const void* dg, const void* u, void* du, int64_t n, void* stream);

/* Pre-LayerNorm rows of BertEmbeddings (Qformer.py:97-118): z fp32 [B, nq+Lt, d] = cat(query_tokens, word[ids] + pos),
 * and the backward: dquery [nq, d] and dpos rows [0, Lt) are written (sums over the batch), dword rows are
 * accumulated with atomic adds (the caller zeroes dword first). */
int hk_qformer_embed_gather(const float* query_tokens, const float* word_emb, const float* pos_emb,
                            const int64_t* input_ids, float* z, int32_t B, int32_t nq, int32_t Lt, int32_t d,
                            int32_t vocab, void* stream);
int hk_qformer_embed_bwd(const float* dz, const int64_t* input_ids, float* dquery, float* dword, float* dpos,
                         int32_t B, int32_t nq, int32_t Lt, int32_t d, int32_t vocab, void* stream);

/* Measurement aid, no reference counterpart (SURVEY.md §8d asks for a register-resident MFMA-peak probe):
 * `grid` workgroups of 4 waves each issue `iters` x 8 independent v_mfma_f32_16x16x32_bf16 (kind 0) or
 * `iters` x 4 v_mfma_f32_32x32x16_bf16 (kind 1) on fragments loaded once from seed_bf16_2048 (2048 bf16 values);
 * one launch = grid * iters * 524288 flops.  out: fp32 [grid * 256]. */
int hk_mfma_probe(const void* seed_bf16_2048, float* out, int32_t grid, int32_t iters, int32_t kind, void* stream);

/* Tuning switch, no reference counterpart: which epilogue kinds of hk_gemm_bf16's large-problem path run on the
 * two-workgroups-per-CU 256x128 kernel (one workgroup's store-bound epilogue under the other's main loop) instead of
 * the one-workgroup 256x256 kernel.  Bit e of `mask` = epilogue kind e (0 bf16 out, 1 fp32 out / fp32 residual, 2 bf16
 * residual stream in place, 3 folded LayerNorm, 4 = 2 + row_part).  mask < 0 only queries.  Returns the mask in force
 * before the call.  Process-wide; the environment variable HK_GEMM_DUO (read once) sets the initial value. */
int hk_gemm_duo_mask(int mask);

/* Library identification and last-error text. */
int hk_version(void);
const char* hk_arch(void);
size_t hk_last_error(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* HAWKEYE_HIP_H */
