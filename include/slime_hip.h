/*
 * slime_hip.h -- C ABI of libslime_hip.so: the MI355X (gfx950) native visual-encoding hot path of SliME.
 *
 * The reference (yfzhang114/SliME) has NO native layer: its hot path is Python calling
 * transformers / torch.nn.  This header is therefore a NEW boundary; each entry
 * point names the reference interface whose arithmetic it replaces (paths relative to the reference
 * repo; "HF" = transformers/models/clip/modeling_clip.py, third-party, lines of v5.15.0).
 *
 * Conventions (all functions):
 *   - extern "C", plain pointers + sizes, a hipStream_t passed as void*; no torch types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - return 0 on success, a negative SLIME_E* code otherwise; never throws; slime_last_error()
 *     returns a thread-local message for the last failure;
 *   - no hidden allocation: scratch comes from the caller (`ws`, sized by the matching
 *     *_workspace_bytes query, 256-byte aligned); no global mutable state (libslime_hip.so exports exactly the
 *     functions declared here; the process-global tuning / ablation hooks used by tools/ live in a separate
 *     diagnostic build, libslime_hip_diag.so, compiled from the same sources with -DSLIME_DIAG); calls on
 *     distinct streams with distinct workspaces may run concurrently;
 *   - "T" is the 16-bit MFMA operand type selected by `dtype` (SLIME_BF16 or SLIME_F16); accumulation and the
 *     LayerNorm / softmax statistics are fp32; the tower's residual stream is SPLIT since ABI 5: hi = T(h), which is at once the
 *     next GEMM's operand, and a lower part -- since ABI 7 ONE SIGNED BYTE per element, lo8 = the next 8 bits of h's fp32 pattern
 *     as the signed distance from hi's pattern (|join(hi, lo8) - h| <= ulp(hi) / 512: 16 significant bits of h with bf16 halves,
 *     19 with fp16; it was a second T value, 16 / 22 bits, in ABI 5-6) -- SLIME_EPI_BIAS_RESID_SPLIT_LN; 3 bytes per element, not 4; the
 *     Llama decoder layer's is 16-bit as in HF (SLIME_EPI_BIAS_RESID_T); primitive-level fp32 forms remain
 *     (SLIME_EPI_BIAS_RESID_F32[_LN]).
 */
#ifndef SLIME_HIP_H
#define SLIME_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLIME_ABI_VERSION 7

enum { SLIME_BF16 = 0, SLIME_F16 = 1, SLIME_F32 = 2, SLIME_U8 = 3 };

enum {
    SLIME_OK = 0,
    SLIME_EINVAL = -1,      /* bad argument / unsupported shape */
    SLIME_EWORKSPACE = -2,  /* workspace too small or misaligned */
    SLIME_ELAUNCH = -3      /* HIP launch error */
};

/* GEMM epilogues:  C = epi(A[M,K] * B[N,K]^T + bias[N]) */
enum {
    SLIME_EPI_BIAS_T = 0,        /* -> T                                   (q/k/v projections)          */
    SLIME_EPI_BIAS_QUICKGELU_T,  /* x*sigmoid(1.702x) -> T                  (CLIP fc1, HF :346-350)      */
    SLIME_EPI_BIAS_GELU_T,       /* exact erf GELU -> T                     (projector/builder.py:53-57) */
    SLIME_EPI_BIAS_F32,          /* -> fp32                                                              */
    SLIME_EPI_BIAS_RESID_F32,    /* C(fp32) += A*B^T + bias, in place       (out_proj / fc2 + residual)  */
    SLIME_EPI_BIAS_RESID_F32_LN, /* the same, and it prepares the NEXT LayerNorm: x16 = T(C), per-row partial sums (slime_gemm_ex) */
    SLIME_EPI_BIAS_RESID_T,      /* C = T(A*B^T + bias + resid): 16-bit residual stream (Llama decoder layer; slime_gemm_ex)       */
    SLIME_EPI_BIAS_GELU_MIX_T,   /* C[t] = T(g0[t] gelu(A[t] B^T + bias) + g1[t] gelu(A2[t] B^T + bias)): GatedBlock hidden rows, mixed in fp32 (slime_gemm_ex) */
    SLIME_EPI_BIAS_RESID_SPLIT_LN /* the residual update of BIAS_RESID_F32_LN on the SPLIT residual stream (ABI 5; the lower part is a byte since
                                   * ABI 7; slime_gemm_ex): h = join(C, lo8) -- the fp32 number whose bit pattern is pattern(float(C)) + lo8 2^SH + 2^(SH-1),
                                   * SH = 8 (bf16) / 5 (fp16): exact, integer arithmetic on IEEE patterns, which are monotonic in the magnitude --;
                                   * c = A*B^T + bias + h; C = T(c) (RNE), lo8 = clamp((pattern(c) - pattern(float(C))) >> SH, -128, 127) (arithmetic shift);
                                   * stats_out = partial sums of the UNROUNDED c, as _LN (slime_patch_embed_prenorm, the producer of layer 0's
                                   * table, sums the ROUNDED rows T(h): the two definitions differ by ~2^-9 relative per element with random
                                   * sign -- noise far below what a LayerNorm statistic resolves, stated here because both feed the same fold).
                                   * C is at once the upper part of the stream and the next GEMM's operand (no separate x16 copy). */
};

int slime_abi_version(void);
const char* slime_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Primitive operators (each is one kernel launch; exported so parity tests can pin every kernel)
 * ---------------------------------------------------------------------------------------------- */

/* MFMA GEMM with fused epilogue.  A: T [M, lda>=K] row-major; B: T [N, K] row-major (nn.Linear
 * weight layout); bias: fp32 [N] or NULL; C: T or fp32 [M, ldc].  Requires K % 64 == 0, N % 128 == 0.
 * Replaces torch.nn.functional.linear / F.conv2d-as-GEMM on the path (HF :148-154,309-311,333,346-350). */
int slime_gemm(const void* A, int lda, const void* B, const float* bias, void* C, int ldc,
               int M, int N, int K, int dtype, int epilogue, void* stream);

/* slime_gemm with LayerNorm FOLDED across two GEMMs (HF CLIPEncoderLayer :370-383: LN1 -> q/k/v, LN2 -> fc1), which removes
 * the LayerNorm launches and the fp32 re-read of the residual stream from the tower:
 *   producer (epilogue SLIME_EPI_BIAS_RESID_F32_LN, the out_proj / fc2 GEMM that updates the residual stream C): also writes
 *     x16[M, ldx] = T(C) and stats_out[M, N/64, 2] = (sum, sum of squares) of the updated fp32 row over each 64-column group;
 *   consumer (ln_stats != NULL; epilogue BIAS_T or BIAS_QUICKGELU_T): A = x16 (un-normalised), B = W . diag(gamma) rounded to T,
 *     bias = b + W beta, ln_colsum[n] = sum_k B[n, k]; the epilogue evaluates rstd * (acc - mu * ln_colsum[n]) + bias[n] with
 *     mu / rstd = rsqrt(var + ln_eps) from the ln_groups partial sums of each row (K = the normalised width).
 * Identical to LayerNorm followed by the GEMM up to rounding (x is rounded to T before instead of after the normalisation). */
typedef struct {
    const void* A; int lda; const void* B; const float* bias; void* C; int ldc;
    int M, N, K, dtype, epilogue;
    const float* ln_stats; int ln_groups; const float* ln_colsum; float ln_eps;     /* consumer side, or NULL / 0 */
    void* x16; int ldx; float* stats_out;                                             /* producer side, or NULL / 0 */
    const void* B_frag;          /* optional: B in MFMA-fragment order (slime_gemm_pack_b), or NULL                  */
    const void* resid; int ldr;  /* epilogue BIAS_RESID_T: residual rows T [M, ldr] (may alias C), else NULL / 0     */
    const void* A2;              /* epilogue BIAS_GELU_MIX_T: the second expert's operand rows T [M, lda], else NULL  */
    const float* mix_gates;      /* epilogue BIAS_GELU_MIX_T: fp32 [M, 2] gate pair per row (slime_gate_weights)      */
    void* lo8; int ldlo;         /* epilogue BIAS_RESID_SPLIT_LN: lower part of the split residual stream, int8 [M, ldlo] (ldlo in BYTES, a multiple of 8; 8-byte aligned), read and written in place */
    const int* row_map;          /* optional (epilogues BIAS_T / BIAS_QUICKGELU_T / BIAS_GELU_T / BIAS_F32 without the LayerNorm fold): output row r
                                  * is stored at C row row_map[r] (device int32 [M], a permutation / scatter into a larger buffer) instead of r:
                                  * lets a GEMM write straight into its consumer's layout (the adapter's token buffer: no merge pass)   */
} slime_gemm_args;
int slime_gemm_ex(const slime_gemm_args* args, void* stream);
/* Epilogue SLIME_EPI_BIAS_GELU_MIX_T (round 4; needs B_frag, N % 256 == 0): ONE launch computes projection[0] + GELU of BOTH experts'
 * rows of a token (A = T(x), A2 = attn(x): a workgroup's 128-row tile is 64 rows of A and the same 64 rows of A2, so a lane holds
 * both results of a token) and stores their gate mix, rounded once: C[t, :] = T(g0 a0 + g1 a1), M = tokens.  projection[2] applied to
 * C equals the mix of the two expert outputs (GatedBlock.forward, projector/builder.py:190-206) up to 1e-6 |b2|, see slime_gate_premix.
 * Which of the two forms slime_gated_forward / slime_adapter_forward take depends on the descriptor: with w1_frag (hidden % 256 == 0) the
 * mix happens here, in fp32, rounded to T once; without it slime_gate_premix mixes the already-rounded hidden rows (two roundings) --
 * the outputs of the two forms differ in the last bit of T for the same inputs.  Eval mode only (no noisy gating, builder.py:150-157). */

/* A STATIC B operand (nn.Linear weights: every GEMM of this path) can additionally be handed over in MFMA-fragment order:
 * out[((t*(K/32) + s)*4 + f)*64 + lane] (16-byte units) = B[64t + 32(f>>1) + 8((lane&15)>>2) + 4(f&1) + (lane&3)][32s + 8(lane>>4) .. +8].
 * With args.B_frag set (B must still be given: small grids keep the LDS-staged kernels) slime_gemm_ex may run the direct-B
 * kernel -- B fragments by plain 16-byte loads straight into registers, 128 x 256 tiles, two workgroups per CU -- whose
 * results are bit-identical to the other kernels' (same k order, same epilogues).  Pack once at weight-load time:
 * N % 64 == 0, K % 64 == 0, out = slime_gemm_packed_b_bytes(N, K) bytes, out != B.  No reference counterpart (layout only). */
size_t slime_gemm_packed_b_bytes(int N, int K);
int slime_gemm_pack_b(const void* B, int N, int K, void* out, void* stream);
/* ABI 5: every kernel slime_gemm_ex dispatches to can take the static operand from the fragment-order image ALONE (the LDS-staged
 * kernels fetch the same 16-byte chunks from it through permuted per-lane source addresses: bit-identical results), so a caller
 * that packed a weight may pass args.B = NULL and free the row-major copy.  Returns 1 if that holds for an [N, K] operand
 * (N % 64 == 0, K % 64 == 0), else 0 (then B is required and B_frag is ignored).  Host-only query. */
int slime_gemm_b_frag_usable(int N, int K);

/* Name (as rocprofv3 prints it) of the kernel instantiation slime_gemm launches for this shape: host-only query. */
int slime_gemm_kernel_name(int M, int N, int K, int dtype, int epilogue, int has_b_frag, char* out_host, size_t out_len);

/* Row-wise LayerNorm over fp32 rows (statistics in fp32, two-pass variance).
 *   y = (x - mean) * rstd * w + b            (normalize != 0)    or   y = x   (normalize == 0)
 *   out_f32[r]  = y                            if out_f32
 *   out_t[r]    = T(y)                         if out_t
 *   out_t2[r]   = T(y + add[r % add_period])   if out_t2   (key position term, sampler.py:164)
 * D must be 128, 256 or 1024.  Replaces nn.LayerNorm (HF :370,379; sampler.py:106,158,168). */
int slime_layernorm(const float* x, int ldx, int rows, int D, const float* w, const float* b, float eps,
                    int normalize, float* out_f32, void* out_t, void* out_t2, const float* add,
                    int add_period, int dtype, void* stream);

/* The tower's front end in ONE launch (ABI 5; replaces slime_im2col + patch GEMM + slime_embed_prenorm):
 *   h[n, 1+P, D] = pre_layrnorm( cat(class_embedding, conv_{patch x patch, stride patch}(pixels)) + position_embedding )
 * (HF CLIPVisionEmbeddings.forward modeling_clip.py:148-154, 209-217 + pre_layrnorm :642, via clip_encoder.py:51,55).
 * pixels [n,3,image,image] (pix_dtype F32, or T = dtype: rounded to T first as `images.to(dtype=self.dtype)` does);
 * patch_w_frag = slime_gemm_pack_b of the conv weight T [D, kpad] (column order (c, ky, kx) = Conv2d.weight.flatten(1), zero
 * padded to kpad, a multiple of 64); cls f32 [D], pos f32 [1+P, D], ln_w / ln_b f32 [D].  A workgroup stages whole image rows
 * in LDS with coalesced 16-byte loads, re-tiles the patch x patch tiles into the MFMA operand there, multiplies, and
 * normalises its rows in registers.  Outputs (any may be NULL except that one of h / x16 is required):
 *   h     f32 [n*(1+P), D]      the residual stream;
 *   x16   T   [n*(1+P), D]      T(h) -- the first GEMM's operand and the upper half of the split residual stream;
 *   lo8   i8  [n*(1+P), D]      the lower part of the split stream: the next 8 bits of h's pattern relative to x16 (SLIME_EPI_BIAS_RESID_SPLIT_LN);
 *   stats f32 [n*(1+P), D/64, 2] (sum, sum of squares) of the ROUNDED rows per 64-column group (first folded LayerNorm).
 * Limits (checked here, by slime_vit_check at pack time and by every slime_vit_forward*): D in {128, 256, 1024};
 * image % patch == 0 and image % 8 == 0; image / patch <= 24 patches per side; 6 * patch * image < 65535; 16-bit pixels
 * must already be of type T (fp32 pixels are rounded on the way in).  CLIP-L/14-336 and -224 fit; a 448 / 14 tower does not. */
int slime_patch_embed_prenorm(const void* pixels, int pix_dtype, const void* patch_w_frag, const float* cls, const float* pos,
                              const float* ln_w, const float* ln_b, float eps, float* h, void* x16, void* lo8, float* stats,
                              int dtype, int n, int image, int patch, int kpad, int D, void* stream);

/* Fused multi-head attention, softmax(Q K^T) V with fp32 online softmax; Q is expected PRE-SCALED
 * by head_dim^-0.5 * log2(e) (SLIME_ATTN_Q_PRESCALE, folded into the q projection weights and bias at pack
 * time): the logits then arrive in log2 units and the kernels evaluate 2^x with v_exp_f32 directly.
 *   q: T, element (b, i, h, d) at q[b*q_bs + i*q_rs + h*head_dim + d]   (q_bs may be 0: shared queries)
 *   k, v likewise with (k_bs,k_rs), (v_bs,v_rs);  o: T at o[b*o_bs + i*o_rs + h*head_dim + d].
 * head_dim 64 (CLIP self-attention, HF :259-277) or 128 (Resampler nn.MultiheadAttention,
 * sampler.py:128,162-165). */
#define SLIME_LOG2E 1.4426950408889634
int slime_attention(const void* q, long q_bs, long q_rs, const void* k, long k_bs, long k_rs,
                    const void* v, long v_bs, long v_rs, void* o, long o_bs, long o_rs,
                    int batch, int heads, int head_dim, int n_q, int n_kv, int dtype, void* stream);

/* out[r, :] = g0*e0[r, :] + g1*e1[r, :],  (g0,g1) = softmax(x[r,:] @ w_gate) / (sum + 1e-6)
 * (GatedBlock.noisy_top_k_gating eval path + mix, projector/builder.py:148,158-165,203-206). */
int slime_gate_mix(const float* x, int D, const float* w_gate /* [D,2] */, const float* e0,
                   const float* e1, float* out, int rows, int H, void* stream);

/* The same gates on the projection MLP's HIDDEN rows (round 4): out[r, :] = T(g0*a0[r, :] + g1*a1[r, :]) with a0 = GELU(W1 x + b1),
 * a1 = GELU(W1 attn(x) + b1), T [rows, H]; out may alias a1.  projection[2] is linear and g0 + g1 = 1/(1 + 1e-6), so
 * projection[2](out) equals the mix of the two expert outputs of projector/builder.py:190-206 up to 1e-6 |b2| -- and runs
 * over one row per token instead of two.  Gate arithmetic as slime_gate_mix (builder.py:148,158-165). */
int slime_gate_premix(const float* x, int D, const float* w_gate /* [D,2] */, const void* a0, const void* a1, void* out,
                      int dtype, int rows, int H, void* stream);

/* The gate pair alone: out[r] = (g0, g1) = softmax(x[r,:] @ w_gate) / (sum + 1e-6), fp32 [rows, 2] (builder.py:148,158-165);
 * the operand `mix_gates` of the SLIME_EPI_BIAS_GELU_MIX_T epilogue.  Arithmetic as slime_gate_mix / slime_gate_premix. */
int slime_gate_weights(const float* x, int D, const float* w_gate /* [D,2] */, float* out, int rows, void* stream);

/* Row gather + cast: out[(g*rows_out + r), :] = cast(in[(g*rows_in + row_off + r), :]),
 * g < groups, r < rows_out.  in fp32, out dtype BF16/F16/F32.  (feature_select's [:,1:],
 * clip_encoder.py:38-39, and output-dtype casts :52,56.) */
int slime_gather_rows(const float* in, int rows_in, int row_off, void* out, int out_dtype,
                      int groups, int rows_out, int C, void* stream);
/* The same from the split residual stream (ABI 5; byte lower part since ABI 7): in = join(hi, lo8), hi T / lo8 int8 [groups*rows_in, C]. */
int slime_gather_rows_split(const void* hi, const void* lo, int dtype, int rows_in, int row_off, void* out, int out_dtype,
                            int groups, int rows_out, int C, void* stream);

/* Spatial merge of compressed local tokens (llava_arch.py:235-244): in fp32 [n=nh*nw, g*g, C] ->
 * out[dst_row0 + ((gy*g+qy)*nw + gx)*g + qx, :]; merge == 0 is the 'flat' order (:233-234). */
int slime_merge_rows(const float* in, void* out, int out_dtype, long dst_row0, int nw, int nh, int g,
                     int C, int merge, void* stream);

/* Batched variants used by the fused adapter (same arithmetic): image b reads in + b*in_image_stride rows and
 * writes out rows b*out_image_stride + dst_row0 + ...; gate_mix_ex writes row r to
 * (r / rows_per_group) * group_stride + row0 + r % rows_per_group in out_dtype (F32/BF16/F16). */
int slime_merge_rows_batched(const float* in, long in_image_stride, void* out, int out_dtype,
                             long out_image_stride, long dst_row0, int images, int nw, int nh, int g, int C,
                             int merge, void* stream);
int slime_gate_mix_ex(const float* x, int D, const float* w_gate, const float* e0, const float* e1, void* out,
                      int out_dtype, int rows, int H, int rows_per_group, long group_stride, long row0,
                      void* stream);

/* Select crops of every image from the tower output (the feat[0] / feat[1:] split of llava_arch.py:224-226):
 * feats T [images*period, P, C]; output crop j = image (j / per_image) * period + first + j % per_image, written
 * as fp32 rows (out_f32) and/or T rows (out_t), either may be NULL. */
int slime_select_crops(const void* feats, int dtype, int P, int C, int period, int first, int per_image,
                       int images, float* out_f32, void* out_t, void* stream);

/* Tile + normalise on device: canvas uint8 [Hc, Wc, 3] (Hc, Wc multiples of `crop`) -> T/fp32
 * crops [ (Hc/crop)*(Wc/crop), 3, crop, crop ] in row-major tile order, value =
 * (u8 * (1/255) - mean[c]) / std[c]  (divide_to_patches mm_utils.py:134-153 + CLIPImageProcessor
 * rescale/normalize). */
int slime_tile_normalize(const uint8_t* canvas, int Hc, int Wc, int crop, const float* mean3_host,
                         const float* std3_host, void* out, int out_dtype, void* stream);

/* ---- image slicer: Pillow-exact bicubic resize (replaces PIL Image.resize in resize_and_pad_image,
 * mm_utils.py:99-131, and the global thumbnail, mm_utils.py:200) -------------------------------------
 * Pillow's 8-bit resize is a two-pass separable convolution (horizontal, rounded to uint8, then vertical)
 * whose per-output-pixel weights are normalised in double precision and converted to 22-bit fixed point.
 * slime_resample_coeffs computes those tables on the HOST (pure C, no device work): bounds[2*o] = first
 * source index, bounds[2*o+1] = tap count, kk[o*ksize + t] = fixed-point weight; ksize from
 * slime_resample_ksize.  slime_resize_bicubic_u8 applies them on the device with integer arithmetic, so the
 * result is bit-identical to Image.resize((out_w, out_h)) for RGB uint8 input. */
int slime_resample_ksize(int in_size, int out_size);
int slime_resample_coeffs(int in_size, int out_size, int* bounds_host, int* kk_host);

/* src uint8 [src_h, src_w, 3] (row stride src_stride bytes) -> dst uint8 [out_h, out_w, 3] (row stride
 * dst_stride bytes: dst may point inside a larger canvas = the centred paste of resize_and_pad_image).
 * bounds_* / kk_* are DEVICE copies of the tables for (src_w -> out_w) and (src_h -> out_h); a pass whose
 * sizes agree is skipped exactly as Pillow does (tables may be NULL for it).  tmp: src_h*out_w*3 bytes
 * (only used when both passes run). */
int slime_resize_bicubic_u8(const uint8_t* src, int src_h, int src_w, long src_stride, uint8_t* dst,
                            long dst_stride, int out_h, int out_w, const int* bounds_h, const int* kk_h,
                            int ksize_h, const int* bounds_v, const int* kk_v, int ksize_v, uint8_t* tmp,
                            size_t tmp_bytes, void* stream);

/* Batched forms for `images` equally sized images (one launch per pass for the whole batch): image b reads
 * src + b*src_image_stride and writes dst + b*dst_image_stride (bytes); tmp: images*src_h*out_w*3 bytes.
 * slime_tile_normalize_batched writes the tiles of canvas b to crops [b*out_image_crops + tile] of `out`, so the
 * global thumbnail (one tile) and the local tiles of an image can be laid out as its (1 + n) consecutive crops. */
int slime_resize_bicubic_u8_batched(const uint8_t* src, int images, long src_image_stride, int src_h, int src_w,
                                    long src_stride, uint8_t* dst, long dst_image_stride, long dst_stride, int out_h,
                                    int out_w, const int* bounds_h, const int* kk_h, int ksize_h, const int* bounds_v,
                                    const int* kk_v, int ksize_v, uint8_t* tmp, size_t tmp_bytes, void* stream);
int slime_tile_normalize_batched(const uint8_t* canvas, int images, long canvas_image_stride, int Hc, int Wc,
                                 int crop, const float* mean3_host, const float* std3_host, void* out,
                                 long out_image_crops, int out_dtype, void* stream);

/* Text-guided router, scores (TextGuidedRouterCosine.forward, resampler/builder.py:186-201):
 * scores[t] = sum_l mask[l] * cos(img[t], text[l]) (mean over l if mask is NULL); img fp32 [T,H], text fp32
 * [L,H], mask uint8 [L]; ws: L+H+4 floats of scratch. */
int slime_router_scores(const float* img, int T, const float* text, int L, const unsigned char* mask, int H,
                        float* scores, float* ws, void* stream);

/* Top-p selection (TextGuidedSampler.forward eval path, resampler/builder.py:258-272): softmax(scores/temp),
 * sort descending, keep the prefix whose cumulative sum is <= topp plus one more token; writes the kept
 * token indices in ascending order to keep_idx[0..*keep_count) (device). probs_out (optional) gets the
 * softmax.  T <= 4096. */
int slime_router_select(const float* scores, int T, float temp, float topp, int* keep_idx, int* keep_count,
                        float* probs_out, void* stream);

/* Batched router for the B images of a step (one launch sequence, per-image results identical to the single calls):
 * image b owns rows row_off[b] .. row_off[b] + n_rows[b] - 1 of img (row_off int64 / n_rows int32, DEVICE arrays -- the
 * local-token rows of a fused [B, 576 + T, H] token buffer or a ragged concatenation); text fp32 [B, L, H], mask uint8
 * [B, L] or NULL; scores / keep_idx are padded [B, T_max], keep_count [B]: ONE D2H read of keep_count serves the batch.
 * ws: slime_router_batched_workspace_floats(B, L, H) floats. */
size_t slime_router_batched_workspace_floats(int B, int L, int H);
int slime_router_scores_batched(const float* img, const long long* row_off, const int* n_rows, int B, int T_max,
                                const float* text, int L, const unsigned char* mask, int H, float* scores, float* ws,
                                void* stream);
int slime_router_select_batched(const float* scores, const int* n_rows, int B, int T_max, float temp, float topp,
                                int* keep_idx, int* keep_count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * CLIP vision tower (CLIPVisionTower.forward + feature_select, clip_encoder.py:36-58)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int hidden, inter, heads, layers_run;   /* layers_run = layers that feed hidden_states[select_layer] */
    int image, patch, kpad;                 /* kpad = 3*patch*patch rounded up to a multiple of 64       */
    int dtype;                              /* SLIME_BF16 / SLIME_F16                                    */
    float eps;
    const void*  patch_w;                   /* T   [hidden, kpad]  (zero padded); unused since ABI 5 (may be NULL): the front end reads patch_w_frag */
    const float* cls;                       /* f32 [hidden]                                               */
    const float* pos;                       /* f32 [1+P, hidden]                                          */
    const float* pre_ln_w; const float* pre_ln_b;
    /* per-layer tensors, contiguous over layers (layer stride = the per-layer element count).  layer_norm1 / layer_norm2
     * are FOLDED into the q/k/v and fc1 GEMMs at pack time (slime_gemm_ex): W' = W . diag(gamma), b' = b + W beta,
     * colsum[n] = sum_k T(W')[n, k]. */
    const void*  w_qkv;                     /* T   [L, 3*hidden, hidden] = Wqkv . diag(ln1_w), q rows x dh^-0.5*log2(e)     */
    const float* b_qkv;                     /* f32 [L, 3*hidden] = b + Wqkv ln1_b   (q part scaled likewise)               */
    const float* colsum_qkv;                /* f32 [L, 3*hidden]                                                          */
    const void*  w_o;   const float* b_o;   /* T   [L, hidden, hidden]; f32 [L, hidden]                                   */
    const void*  w_fc1;                     /* T   [L, inter, hidden] = W1 . diag(ln2_w)                                  */
    const float* b_fc1;                     /* f32 [L, inter] = b + W1 ln2_b                                              */
    const float* colsum_fc1;                /* f32 [L, inter]                                                             */
    const void*  w_fc2; const float* b_fc2; /* T   [L, hidden, inter];  f32 [L, hidden]                                   */
    /* fragment-order copies of the four per-layer weights (slime_gemm_pack_b per layer, same layer stride), or NULL.  ABI 5: where
     * a _frag copy is given the row-major tensor of the same weight may be NULL (slime_gemm_b_frag_usable) */
    const void*  w_qkv_frag; const void* w_o_frag; const void* w_fc1_frag; const void* w_fc2_frag;
    const void*  patch_w_frag;              /* slime_gemm_pack_b of patch_w [hidden, kpad]: REQUIRED since ABI 5 (slime_patch_embed_prenorm) */
} slime_vit_desc;

size_t slime_vit_workspace_bytes(const slime_vit_desc* d, int n_crops);
/* Validate a descriptor without running it (host-only): the checks every slime_vit_forward* starts with -- dtype, widths,
 * head_dim 64, the fused front end's geometry limits (slime_patch_embed_prenorm), presence of every weight.  Callers run it
 * when they PACK a tower, so an unsupported geometry fails there with slime_last_error() naming the limit. */
int slime_vit_check(const slime_vit_desc* d);
/* The epilogue (SLIME_EPI_*) of the tower's out_proj / fc2 launches in this build: SLIME_EPI_BIAS_RESID_SPLIT_LN (split
 * residual stream: 16-bit operand part + one byte, ABI 7) -- what a profiler label for those kernels has to be generated with (slime_gemm_kernel_name).  Host-only. */
int slime_vit_residual_epilogue(void);

/* Optional in-situ timing probe: slime_vit_forward_ex records the HIP events `start` / `stop` (hipEvent_t,
 * owned by the caller) immediately before / after the launch of one kernel of one layer, on the call's
 * stream.  kernel: 1 qkv GEMM, 2 attention, 3 out_proj GEMM, 5 fc1 GEMM, 6 fc2 GEMM (0 and 4 were the LayerNorm
 * launches of ABI 1; they are folded into the GEMMs now and never fire). */
typedef struct { int layer; int kernel; void* start; void* stop; } slime_probe;

/* pixels [n,3,image,image] -> out [n, P(+1 if keep_cls), hidden] in out_dtype (BF16/F16/F32).
 * If hidden_f32 is non-NULL it also receives the fp32 residual stream [n, 1+P, hidden]
 * (the selected hidden state before the cls drop / cast). */
int slime_vit_forward(const slime_vit_desc* d, const void* pixels, int pix_dtype, int n_crops,
                      void* out, int out_dtype, int keep_cls, float* hidden_f32,
                      void* ws, size_t ws_bytes, void* stream);

/* Same, with an optional probe (NULL = none). */
int slime_vit_forward_ex(const slime_vit_desc* d, const void* pixels, int pix_dtype, int n_crops,
                         void* out, int out_dtype, int keep_cls, float* hidden_f32,
                         void* ws, size_t ws_bytes, void* stream, const slime_probe* probe);

/* All hidden states of ONE pass (HF CLIPVisionModel(..., output_hidden_states=True), which CLIPVisionTower.forward requests,
 * clip_encoder.py:51,55): states_f32 [layers_run + 1, n, 1 + P, hidden] fp32 -- entry 0 = embeddings after pre_layrnorm,
 * entry i = output of encoder layer i.  Run with a descriptor whose layers_run covers the last state wanted. */
int slime_vit_forward_states(const slime_vit_desc* d, const void* pixels, int pix_dtype, int n_crops, float* states_f32,
                             void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Resampler (sampler.py:91-173) with kv_proj = proj = Identity, as post_qformer / GatedBlock.attn
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int dim, heads, n_query, n_kv, dtype;
    float eps;
    const void*  q_proj;                    /* T   [n_query, dim]: ((ln_q(query)+pos_embed) Wq^T + bq) * dh^-0.5 * log2(e), input independent */
    const float* pos_k;                     /* f32 [n_kv, dim]: get_abs_pos(pos_embed, kv grid)           */
    const float* ln_kv_w; const float* ln_kv_b;
    const void*  w_k; const float* b_k;     /* T [dim, dim]; f32 [dim]                                    */
    const void*  w_v; const float* b_v;
    const void*  w_o; const float* b_o;
    const float* ln_post_w; const float* ln_post_b;
    const void*  w_k_frag; const void* w_v_frag; const void* w_o_frag;   /* optional slime_gemm_pack_b copies, or NULL */
} slime_resampler_desc;

size_t slime_resampler_workspace_bytes(const slime_resampler_desc* d, int n);

/* x fp32 [n, n_kv, dim] (row stride ldx) -> out_f32 [n*n_query, dim] and/or out_t T [n*n_query, dim] */
int slime_resampler_forward(const slime_resampler_desc* d, const float* x, int ldx, int n,
                            float* out_f32, void* out_t, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * mm_projector (projector/builder.py): MLP and GatedBlock
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int in_dim, hidden, dtype;
    const void* w1; const float* b1;        /* T [hidden, in_dim]                                         */
    const void* w2; const float* b2;        /* T [hidden, hidden]                                         */
    const void* w1_frag; const void* w2_frag; /* optional slime_gemm_pack_b copies, or NULL                */
} slime_mlp_desc;

size_t slime_mlp_workspace_bytes(const slime_mlp_desc* d, int rows);

/* Linear -> exact GELU -> Linear (projector/builder.py:53-57).  Input either x_t (T [rows,in_dim]) or,
 * if x_t is NULL, x_f32 (cast to T first).  out fp32 [rows, hidden]. */
int slime_mlp_forward(const slime_mlp_desc* d, const float* x_f32, const void* x_t, int rows,
                      float* out, void* ws, size_t ws_bytes, void* stream);

size_t slime_gated_workspace_bytes(const slime_mlp_desc* mlp, const slime_resampler_desc* attn, int n);

/* GatedBlock.forward full path (projector/builder.py:183-209) on x fp32 [n, 576, in_dim]:
 * learnable_gated < 0 -> softmax-gated mix of E0 = mlp(x) and E1 = mlp(attn(x)); 0/1 -> that expert. */
int slime_gated_forward(const slime_mlp_desc* mlp, const slime_resampler_desc* attn,
                        const float* w_gate /* f32 [in_dim, 2] */, int learnable_gated,
                        const float* x, int n, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- fused adapter: everything between the tower and the router for a batch of images with the same
 * crop layout (llava_arch.py:217-246: feat[0] -> mm_projector (GatedBlock), feat[1:] -> post_qformer ->
 * mm_projector (MLP) -> spatial merge), as ONE launch sequence.  The three MLP passes (projection(x),
 * projection(attn(x)), projection(post_qformer(local))) share weights and are row independent, so they run as
 * one GEMM pair over the stacked rows; per-row results are identical to the separate calls.
 *   feats: T [n_images*(1+n_local), 576, D] tower output, crop 0 of each image = global view.
 *   out:   out_dtype [n_images, out_image_stride rows, H]; image i gets rows [0,576) = gated global tokens and
 *          rows [576, 576 + n_local*g*g) = merged local tokens (raster order if merge != 0; nw*nh == n_local).
 * projection[2] stores every row straight into `out` (slime_gemm_args.row_map) when out_dtype is fp32 or the operand type T;
 * a 16-bit out_dtype OTHER than T (bf16 tokens from an fp16 adapter or vice versa) takes fp32 rows + slime_merge_rows_batched
 * instead: same values, two more passes over the token rows. */
size_t slime_adapter_workspace_bytes(const slime_mlp_desc* mlp, const slime_resampler_desc* attn,
                                     const slime_resampler_desc* post, int n_images, int n_local);
int slime_adapter_forward(const slime_mlp_desc* mlp, const slime_resampler_desc* attn, const float* w_gate,
                          int learnable_gated, const slime_resampler_desc* post, const void* feats,
                          int n_images, int n_local, int nw, int nh, int merge, void* out, int out_dtype,
                          long out_image_stride, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * After the visual tokens: splice into the text embeddings, Llama prefill attention
 * ---------------------------------------------------------------------------------------------- */

/* new_input_embeds of prepare_inputs_labels_for_multimodal (llava/model/llava_arch.py:343-459), all rows of the padded
 * batch in one launch: out[r, :] = table[src[r], :] if src[r] >= 0 (embed_tokens row of a text token),
 * feats[-2 - src[r], :] if src[r] <= -2 (row of the concatenated image features), zeros if src[r] == -1 (padding).
 * src is a DEVICE int64 array built from the index plan (integer host logic, slime_amd/model/llava_arch.py).
 * Equal source / destination dtypes are copied bit for bit.  H * sizeof(out element) must be a multiple of 16. */
int slime_splice_rows(const void* table, int table_dtype, long table_rows, const void* feats, int feats_dtype,
                      long feat_rows, const int64_t* src, void* out, int out_dtype, long rows, int H, void* stream);

/* Rotary position embedding, in place, on the first n_rot_heads heads (head_dim 128) of every row of a packed qkv buffer
 * (row stride in elements): HF apply_rotary_pos_emb / rotate_half as called at llama_flash_attn_monkey_patch.py:51-54,
 * angle = (float)pos[row] * inv_freq[i], i < 64 (inv_freq: DEVICE fp32 [64], computed by the host exactly as
 * LlamaRotaryEmbedding does).  The first n_q_heads heads are additionally multiplied by q_scale (the attention kernels
 * expect q pre-scaled by head_dim^-0.5 * log2 e). */
int slime_rope(void* qkv, long row_stride, const int32_t* pos, long rows, int n_rot_heads, int n_q_heads, int head_dim,
               const float* inv_freq, float q_scale, int dtype, void* stream);

/* Causal grouped-query attention over the un-padded tokens of every sequence (llama_flash_attn_monkey_patch.py:65-90:
 * repeat_kv + unpad_input + flash_attn_unpadded_qkvpacked_func(causal=True) + pad_input): query head h uses kv head
 * h / (n_heads / n_kv_heads); sequence b attends within its token range [kv_start[b], kv_start[b] + kv_len[b]) (NULL:
 * the whole sequence), query i sees keys <= i; output rows outside the range are zero.  Addressing as slime_attention
 * (q pre-scaled by head_dim^-0.5 * log2 e, RoPE already applied); head_dim 128. */
int slime_prefill_attention(const void* q, long q_bs, long q_rs, const void* k, long k_bs, long k_rs, const void* v,
                            long v_bs, long v_rs, void* o, long o_bs, long o_rs, int batch, int n_heads, int n_kv_heads,
                            int head_dim, int S, const int32_t* kv_start, const int32_t* kv_len, int dtype, void* stream);

/* LlamaAttention.forward as patched by llava/train/llama_flash_attn_monkey_patch.py:16-93 (no KV cache: prefill). */
typedef struct {
    int hidden, n_heads, n_kv_heads, head_dim, dtype;
    const void*  w_qkv;                     /* T   [(n_heads + 2 n_kv_heads) * head_dim, hidden]: q_proj, k_proj, v_proj rows */
    const void*  w_o;                       /* T   [hidden, n_heads * head_dim]                                               */
    const float* inv_freq;                  /* f32 [head_dim / 2]                                                             */
    const void*  w_qkv_frag; const void* w_o_frag;   /* optional slime_gemm_pack_b copies, or NULL                                */
} slime_llama_attn_desc;

size_t slime_llama_attn_workspace_bytes(const slime_llama_attn_desc* d, int batch, int S);

/* hidden T [batch*S, hidden], position_ids int32 [batch*S] -> out [batch*S, hidden] (T or fp32). */
int slime_llama_attn_forward(const slime_llama_attn_desc* d, const void* hidden, const int32_t* position_ids,
                             const int32_t* kv_start, const int32_t* kv_len, int batch, int S, void* out, int out_dtype,
                             void* ws, size_t ws_bytes, void* stream);

/* The same sub-layer with the decoder layer's residual add fused into o_proj's epilogue (HF LlamaDecoderLayer.forward:
 * hidden_states = residual + self_attn(norm(hidden_states)); the reference reaches it through the patched
 * LlamaAttention.forward, llama_flash_attn_monkey_patch.py:16-93):  out = T(resid + o_proj(attention(hidden))), all T
 * [batch*S, hidden]; out may alias resid (in-place stream), hidden may be resid itself (no norm in between) but must not alias out.
 * No torch arithmetic is left between two layers. */
int slime_llama_attn_forward_resid(const slime_llama_attn_desc* d, const void* hidden, const int32_t* position_ids,
                                   const int32_t* kv_start, const int32_t* kv_len, int batch, int S, const void* resid,
                                   void* out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement aid (bench.py; no reference counterpart)
 * ---------------------------------------------------------------------------------------------- */

/* One launch of a bare v_mfma_f32_16x16x32_{bf16,f16} stream on register-resident operands: one workgroup of 8 waves per CU,
 * every wave multiplies the fragments of a 64 x 64 tile (2 k-steps, 32 MFMAs per iteration, 16 independent accumulators) `iters`
 * times with no memory instruction in the loop.  operands: >= 16 KiB of finite random 16-bit values (wave w reads the 16 KiB
 * slot w mod (operand_bytes / 16 KiB)); out: >= CUs x 512 floats (the sums, so that the loop is live); *flops_host receives the
 * FLOPs one launch performs.  Timed by the caller with HIP events, it is THIS chip's MFMA ceiling under its power cap at this
 * moment -- the denominator bench.py reports beside the 2.5 PFLOP/s dense peak. */
int slime_mfma_stream_probe(int dtype, int iters, const void* operands, size_t operand_bytes, float* out, size_t out_bytes,
                            double* flops_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLIME_HIP_H */