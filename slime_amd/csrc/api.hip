// api.hip -- C-ABI drivers of libslime_hip: the CLIP tower layer loop, the Resampler, the projector
// MLP and the GatedBlock, each a fixed sequence of the primitive kernels on one stream.  No
// allocation, no synchronisation, no global mutable state (the last-error string is thread-local): callable under
// hipGraph capture.
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void slime_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* slime_last_error(void) { return g_err; }
// slime_gemm with the optional fragment-order copy of the static operand
static int gemm_w(const void* A, int lda, const void* B, const void* B_frag, const float* bias, void* C, int ldc, int M, int N, int K,
                  int dtype, int epilogue, void* stream) {
    slime_gemm_args a{};
    a.A = A; a.lda = lda; a.B = B; a.bias = bias; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.dtype = dtype; a.epilogue = epilogue;
    a.B_frag = B_frag;
    return slime_gemm_ex(&a, stream);
}

// projection[0] + GELU of both experts' rows and their gate mix in ONE launch (SLIME_EPI_BIAS_GELU_MIX_T; direct-B kernel only)
static bool mix_in_gemm(const slime_mlp_desc* m) {
    return m->w1_frag && m->hidden % 256 == 0 && m->in_dim % 64 == 0 && ((uintptr_t)m->w1_frag % 16) == 0;
}
static int gemm_mix(const void* A, const void* A2, const float* gates, const slime_mlp_desc* m, void* C, int tokens, void* stream) {
    slime_gemm_args a{};
    a.A = A; a.A2 = A2; a.mix_gates = gates; a.lda = m->in_dim; a.B = m->w1; a.B_frag = m->w1_frag; a.bias = m->b1; a.C = C; a.ldc = m->hidden;
    a.M = tokens; a.N = m->hidden; a.K = m->in_dim; a.dtype = m->dtype; a.epilogue = SLIME_EPI_BIAS_GELU_MIX_T;
    return slime_gemm_ex(&a, stream);
}

extern "C" int slime_abi_version(void) { return SLIME_ABI_VERSION; }

#define TRY(call)                     \
    do {                              \
        int rc_ = (call);             \
        if (rc_ != SLIME_OK) return rc_; \
    } while (0)

namespace {
static bool is16(int dt) { return dt == SLIME_BF16 || dt == SLIME_F16; }
}  // namespace

// ------------------------------------------------------------------------------------------------
// CLIP tower
// ------------------------------------------------------------------------------------------------
#ifndef SLIME_OPT_ALIAS_WS
#define SLIME_OPT_ALIAS_WS 1
#endif
// Round 5: the residual stream between the layers is SPLIT (hi = T(h), which IS the next GEMM's operand, + a lower part) instead of
// fp32 rows plus a separate T(h) copy; round 6 (ABI 7): the lower part is one signed byte per element (common.h resid_delta): the
// out_proj / fc2 epilogues move 6 bytes per element -- 8 with a 16-bit lower part (round 5), 10 with the fp32 stream
// (SLIME_EPI_BIAS_RESID_SPLIT_LN).  0 = the fp32 stream of rounds 1-4 (tools/build_variants.sh A/B).
#ifndef SLIME_OPT_SPLIT_RESID
#define SLIME_OPT_SPLIT_RESID 1
#endif
struct VitPlan {
    size_t xn, qkv, ctx, ff, h, stats, total;   // offsets (h: the fp32 residual rows, or the lower part of the split stream: one byte per element since ABI 7)
};

static VitPlan vit_plan(const slime_vit_desc* d, int n) {
    const int g = d->image / d->patch, P = g * g, S = P + 1;
    const size_t M = (size_t)n * S, D = d->hidden;
    VitPlan p{};
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = align_up(off, 256); off = o + b; return o; };
    p.h = take(M * D * (SLIME_OPT_SPLIT_RESID ? 1 : 4));
    p.xn = take(M * D * 2);
    const size_t ff_bytes = M * (size_t)d->inter * 2;
#if SLIME_OPT_ALIAS_WS
    // q/k/v and the attention context die before fc1 writes the MLP's intermediate rows, and those die before the next layer's q/k/v
    // GEMM: the three share one region (for CLIP-L, 3 D + D = the intermediate width exactly).  The launches of a stream are serial, so
    // nothing else changes -- except that a 20-crop stream now cycles through 165 MB instead of 260, and the 256 MiB memory-side
    // cache, which two such streams share, keeps more of what the next kernel reads (profiles/r04_fabric_traffic.txt).
    const size_t qkv_bytes = align_up(M * 3 * D * 2, 256), attn_bytes = qkv_bytes + M * D * 2;
    const size_t region = ff_bytes > attn_bytes ? ff_bytes : attn_bytes;
    p.ff = take(region);
    p.qkv = p.ff;
    p.ctx = p.ff + qkv_bytes;
#else
    p.qkv = take(M * 3 * D * 2);
    p.ctx = take(M * D * 2);
    p.ff = take(ff_bytes);
#endif
    p.stats = take(M * (D / 64) * 2 * sizeof(float));      // LayerNorm fold: (sum, sum of squares) per row and 64-column group
    p.total = align_up(off, 256);
    return p;
}

int slime_patch_embed_geometry(int image, int patch, int kpad, int D, size_t* lds_out);   // patch_embed.hip: the fused front end's limits

static int vit_validate(const slime_vit_desc* d) {
    SLIME_REQUIRE(d, "vit: null descriptor");
    SLIME_REQUIRE(is16(d->dtype), "vit: dtype must be BF16 or F16");
    SLIME_REQUIRE(d->hidden == 128 || d->hidden == 256 || d->hidden == 1024, "vit: hidden=%d unsupported", d->hidden);
    SLIME_REQUIRE(d->hidden % 64 == 0, "vit: hidden must be a multiple of 64 (LayerNorm partial sums per 64 columns)");
    SLIME_REQUIRE(d->heads > 0 && d->hidden % d->heads == 0 && d->hidden / d->heads == 64, "vit: head_dim must be 64");
    SLIME_REQUIRE(d->inter % 128 == 0 && d->inter % 64 == 0, "vit: intermediate size %d must be a multiple of 128", d->inter);
    SLIME_REQUIRE(d->patch > 0 && d->image % d->patch == 0, "vit: image %d not a multiple of patch %d", d->image, d->patch);
    TRY(slime_patch_embed_geometry(d->image, d->patch, d->kpad, d->hidden, nullptr));      // the one-launch front end's geometry limits
    SLIME_REQUIRE(d->layers_run >= 0, "vit: layers_run < 0");
    SLIME_REQUIRE(d->patch_w_frag && d->cls && d->pos && d->pre_ln_w && d->pre_ln_b,
                  "vit: missing embedding weights (patch_w_frag = slime_gemm_pack_b of the conv weight is required since ABI 5)");
    // a per-layer weight is given as its fragment-order image, its row-major image, or both (slime_gemm_b_frag_usable)
    SLIME_REQUIRE(d->layers_run == 0 || ((d->w_qkv || d->w_qkv_frag) && d->b_qkv && d->colsum_qkv && (d->w_o || d->w_o_frag) && d->b_o &&
                                         (d->w_fc1 || d->w_fc1_frag) && d->b_fc1 && d->colsum_fc1 && (d->w_fc2 || d->w_fc2_frag) && d->b_fc2),
                  "vit: missing layer weights");
    return SLIME_OK;
}

extern "C" int slime_vit_check(const slime_vit_desc* d) { return vit_validate(d); }

extern "C" int slime_vit_residual_epilogue(void) {
    return SLIME_OPT_SPLIT_RESID ? SLIME_EPI_BIAS_RESID_SPLIT_LN : SLIME_EPI_BIAS_RESID_F32_LN;
}

extern "C" size_t slime_vit_workspace_bytes(const slime_vit_desc* d, int n_crops) {
    if (!d || n_crops <= 0 || d->patch <= 0) return 0;
    return vit_plan(d, n_crops).total;
}

extern "C" int slime_vit_forward(const slime_vit_desc* d, const void* pixels, int pix_dtype, int n, void* out,
                                 int out_dtype, int keep_cls, float* hidden_f32, void* ws, size_t ws_bytes,
                                 void* stream) {
    return slime_vit_forward_ex(d, pixels, pix_dtype, n, out, out_dtype, keep_cls, hidden_f32, ws, ws_bytes, stream, nullptr);
}

// diagnostic build only (libslime_hip_diag.so; results become wrong): bit k set = the tower skips kernel id k of every layer, to read each kernel's
// MARGINAL cost inside the two-stream tower (tools/marginal_bench.py)
#ifdef SLIME_DIAG
static int g_vit_skip_mask = 0;
extern "C" void slime_vit_set_skip_mask(int m) { g_vit_skip_mask = m; }
#else
static constexpr int g_vit_skip_mask = 0;
#endif

#define PROBED(kid, call)                                                                             \
    do {                                                                                              \
        const bool on_ = probe && probe->layer == l && probe->kernel == (kid);                        \
        if (on_ && probe->start) (void)hipEventRecord((hipEvent_t)probe->start, (hipStream_t)stream);       \
        if (!((g_vit_skip_mask >> (kid)) & 1)) TRY(call);                                             \
        if (on_ && probe->stop) (void)hipEventRecord((hipEvent_t)probe->stop, (hipStream_t)stream);          \
    } while (0)

static int vit_run(const slime_vit_desc* d, const void* pixels, int pix_dtype, int n, void* out, int out_dtype, int keep_cls,
                   float* hidden_f32, void* ws, size_t ws_bytes, void* stream, const slime_probe* probe, float* states);

extern "C" int slime_vit_forward_ex(const slime_vit_desc* d, const void* pixels, int pix_dtype, int n, void* out,
                                    int out_dtype, int keep_cls, float* hidden_f32, void* ws, size_t ws_bytes,
                                    void* stream, const slime_probe* probe) {
    return vit_run(d, pixels, pix_dtype, n, out, out_dtype, keep_cls, hidden_f32, ws, ws_bytes, stream, probe, nullptr);
}

extern "C" int slime_vit_forward_states(const slime_vit_desc* d, const void* pixels, int pix_dtype, int n, float* states_f32,
                                        void* ws, size_t ws_bytes, void* stream) {
    SLIME_REQUIRE(states_f32, "vit_forward_states: states_f32 is required");
    return vit_run(d, pixels, pix_dtype, n, nullptr, SLIME_F32, 1, nullptr, ws, ws_bytes, stream, nullptr, states_f32);
}

static int vit_run(const slime_vit_desc* d, const void* pixels, int pix_dtype, int n, void* out, int out_dtype, int keep_cls,
                   float* hidden_f32, void* ws, size_t ws_bytes, void* stream, const slime_probe* probe, float* states) {
    TRY(vit_validate(d));
    SLIME_REQUIRE(pixels && n > 0, "vit: bad input");
    SLIME_REQUIRE(out || hidden_f32 || states, "vit: no output requested");
    SLIME_REQUIRE(!out || out_dtype == SLIME_F32 || is16(out_dtype), "vit: bad out dtype");
    const VitPlan p = vit_plan(d, n);
    if (!ws || ws_bytes < p.total || ((uintptr_t)ws % 256) != 0) {
        slime_set_error("vit: workspace %zu B (need %zu, 256-B aligned)", ws_bytes, p.total);
        return SLIME_EWORKSPACE;
    }
    const int g = d->image / d->patch, P = g * g, S = P + 1, D = d->hidden, F = d->inter;
    const int M = n * S;
    char* w = (char*)ws;
    void* xn = w + p.xn;
    char* qkv = w + p.qkv;
    void* ctx = w + p.ctx;
    void* ff = w + p.ff;
    const int dt = d->dtype;
    float* stats = (float*)(w + p.stats);
    const int G = D / 64;
#if SLIME_OPT_SPLIT_RESID
    // residual stream = (xn, lo): xn = T(h) is the upper part AND the operand of the q/k/v / fc1 GEMMs, lo = one signed byte per element (common.h resid_delta)
    void* lo = w + p.h;
    float* h = nullptr;
#else
    void* lo = nullptr;
    float* h = hidden_f32 ? hidden_f32 : (float*)(w + p.h);
#endif

    // patch embed (MFMA conv), class token, position table, pre-LayerNorm and the first folded LN1's operand + partial sums: ONE launch
    TRY(slime_patch_embed_prenorm(pixels, pix_dtype, d->patch_w_frag, d->cls, d->pos, d->pre_ln_w, d->pre_ln_b, d->eps, h, xn, lo, stats,
                                  dt, n, d->image, d->patch, d->kpad, D, stream));
    // hidden_states[i] of HF's output_hidden_states=True (entry 0 = the pre-LayerNorm'd embeddings, entry i = after layer i):
    // a device-side copy of the residual stream on the call's stream (capturable), one per state
#if SLIME_OPT_SPLIT_RESID
#define SNAPSHOT(idx) TRY(slime_gather_rows_split(xn, lo, dt, S, 0, states + (size_t)(idx) * M * D, SLIME_F32, n, S, D, stream))
#else
#define SNAPSHOT(idx)                                                                                                           \
    do {                                                                                                                        \
        if (hipMemcpyAsync(states + (size_t)(idx) * M * D, h, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice,         \
                           (hipStream_t)stream) != hipSuccess) { slime_set_error("vit: hidden-state snapshot failed"); return SLIME_ELAUNCH; } \
    } while (0)
#endif
    if (states) SNAPSHOT(0);

    // Layer loop, 5 launches per layer.  Both LayerNorms are FOLDED into the GEMMs around them (slime_gemm_ex): the GEMM that
    // updates the residual stream (previous fc2 / out_proj, or the embedding kernel) leaves the rows rounded to T (`xn`) and
    // their partial sums (`stats`); the q/k/v and fc1 GEMMs run on those un-normalised rows with gamma folded into their
    // weights and apply mean / rstd in the epilogue.  No LayerNorm kernel, no fp32 re-read of the residual stream.
    for (int l = 0; l < d->layers_run; ++l) {
        auto layer_w = [&](const void* base, size_t per_layer) -> const void* {     // this layer's slice of a per-layer weight (or NULL)
            return base ? (const char*)base + (size_t)l * per_layer * 2 : nullptr;
        };
        const bool last = l + 1 == d->layers_run;
        slime_gemm_args ga{};
        ga.M = M; ga.dtype = dt; ga.ln_eps = d->eps;
        // q/k/v = LN1(h) Wqkv^T + b  (HF :370-371, :309-311)
        ga.A = xn; ga.lda = D; ga.B = layer_w(d->w_qkv, (size_t)3 * D * D); ga.bias = d->b_qkv + (size_t)l * 3 * D; ga.C = qkv; ga.ldc = 3 * D;
        ga.N = 3 * D; ga.K = D;
        ga.epilogue = SLIME_EPI_BIAS_T; ga.ln_stats = stats; ga.ln_groups = G; ga.ln_colsum = d->colsum_qkv + (size_t)l * 3 * D;
        ga.B_frag = layer_w(d->w_qkv_frag, (size_t)3 * D * D);
        PROBED(1, slime_gemm_ex(&ga, stream));
        PROBED(2, slime_attention(qkv, (long)S * 3 * D, 3 * D, qkv + (size_t)D * 2, (long)S * 3 * D, 3 * D,
                                  qkv + (size_t)2 * D * 2, (long)S * 3 * D, 3 * D, ctx, (long)S * D, D, n, d->heads, 64, S, S,
                                  dt, stream));
        // the two GEMMs that update the residual stream and prepare the next folded LayerNorm
        auto resid_update = [&](slime_gemm_args& r, bool with_ln) {
#if SLIME_OPT_SPLIT_RESID
            (void)with_ln;                                   // the last layer's partial sums are written and never read
            r.C = xn; r.ldc = D; r.lo8 = lo; r.ldlo = D; r.stats_out = stats; r.epilogue = SLIME_EPI_BIAS_RESID_SPLIT_LN;
#else
            r.C = h; r.ldc = D;
            r.epilogue = with_ln ? SLIME_EPI_BIAS_RESID_F32_LN : SLIME_EPI_BIAS_RESID_F32;
            if (with_ln) { r.x16 = xn; r.ldx = D; r.stats_out = stats; }
#endif
        };
        // h += ctx Wo^T + b; leaves T(h) and its partial sums for LN2  (HF :372-377)
        ga = slime_gemm_args{};
        ga.M = M; ga.dtype = dt;
        ga.A = ctx; ga.lda = D; ga.B = layer_w(d->w_o, (size_t)D * D); ga.bias = d->b_o + (size_t)l * D; ga.N = D; ga.K = D;
        resid_update(ga, true);
        ga.B_frag = layer_w(d->w_o_frag, (size_t)D * D);
        PROBED(3, slime_gemm_ex(&ga, stream));
        // ff = quick_gelu(LN2(h) W1^T + b)  (HF :379-380, :346-350)
        ga = slime_gemm_args{};
        ga.M = M; ga.dtype = dt; ga.ln_eps = d->eps;
        ga.A = xn; ga.lda = D; ga.B = layer_w(d->w_fc1, (size_t)F * D); ga.bias = d->b_fc1 + (size_t)l * F; ga.C = ff; ga.ldc = F; ga.N = F; ga.K = D;
        ga.epilogue = SLIME_EPI_BIAS_QUICKGELU_T; ga.ln_stats = stats; ga.ln_groups = G; ga.ln_colsum = d->colsum_fc1 + (size_t)l * F;
        ga.B_frag = layer_w(d->w_fc1_frag, (size_t)F * D);
        PROBED(5, slime_gemm_ex(&ga, stream));
        // h += ff W2^T + b; prepares the next layer's LN1 unless this is the last layer that runs  (HF :381-383)
        ga = slime_gemm_args{};
        ga.M = M; ga.dtype = dt;
        ga.A = ff; ga.lda = F; ga.B = layer_w(d->w_fc2, (size_t)D * F); ga.bias = d->b_fc2 + (size_t)l * D; ga.N = D; ga.K = F;
        resid_update(ga, !last);
        ga.B_frag = layer_w(d->w_fc2_frag, (size_t)D * F);
        PROBED(6, slime_gemm_ex(&ga, stream));
        if (states) SNAPSHOT(l + 1);
    }
#if SLIME_OPT_SPLIT_RESID
    if (hidden_f32) TRY(slime_gather_rows_split(xn, lo, dt, S, 0, hidden_f32, SLIME_F32, n, S, D, stream));
    if (out) {
        // feature_select: 'patch' drops the class token (clip_encoder.py:38-39), cast to out dtype (:52,56)
        TRY(slime_gather_rows_split(xn, lo, dt, S, keep_cls ? 0 : 1, out, out_dtype, n, keep_cls ? S : P, D, stream));
    }
#else
    if (out) {
        TRY(slime_gather_rows(h, S, keep_cls ? 0 : 1, out, out_dtype, n, keep_cls ? S : P, D, stream));
    }
#endif
    return SLIME_OK;
}

// ------------------------------------------------------------------------------------------------
// Resampler
// ------------------------------------------------------------------------------------------------
static int resampler_validate(const slime_resampler_desc* d) {
    SLIME_REQUIRE(d, "resampler: null descriptor");
    SLIME_REQUIRE(is16(d->dtype), "resampler: dtype must be BF16 or F16");
    SLIME_REQUIRE(d->dim == 128 || d->dim == 256 || d->dim == 1024, "resampler: dim=%d unsupported", d->dim);
    SLIME_REQUIRE(d->heads > 0 && d->dim % d->heads == 0, "resampler: heads");
    const int dh = d->dim / d->heads;
    SLIME_REQUIRE(dh == 64 || dh == 128, "resampler: head_dim %d unsupported", dh);
    SLIME_REQUIRE(d->n_query > 0 && d->n_kv > 0, "resampler: empty query/key grid");
    // a projection weight is given as its row-major image, its fragment-order image, or both (slime_gemm_b_frag_usable)
    SLIME_REQUIRE(d->q_proj && d->pos_k && d->ln_kv_w && d->ln_kv_b && (d->w_k || d->w_k_frag) && d->b_k && (d->w_v || d->w_v_frag) && d->b_v &&
                  (d->w_o || d->w_o_frag) && d->b_o && d->ln_post_w && d->ln_post_b, "resampler: missing weights");
    return SLIME_OK;
}

struct ResPlan { size_t xn, xk, kp, vp, ctx, o32, total; };
static ResPlan res_plan(const slime_resampler_desc* d, int n) {
    const size_t Rk = (size_t)n * d->n_kv, Rq = (size_t)n * d->n_query, D = d->dim;
    ResPlan p{};
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = align_up(off, 256); off = o + b; return o; };
    p.xn = take(Rk * D * 2); p.xk = take(Rk * D * 2); p.kp = take(Rk * D * 2); p.vp = take(Rk * D * 2);
    p.ctx = take(Rq * D * 2); p.o32 = take(Rq * D * 4);
    p.total = align_up(off, 256);
    return p;
}

extern "C" size_t slime_resampler_workspace_bytes(const slime_resampler_desc* d, int n) {
    if (!d || n <= 0) return 0;
    return res_plan(d, n).total;
}

int layernorm_crops_launch(const void* feats, int dtype, int P, int period, int first, int per_image, int images, int D, const float* w,
                           const float* b, float eps, void* out_t, void* out_t2, const float* add, int add_period, void* stream);   // rowwise.hip (internal)

// crops != nullptr (fused adapter, round 5): the input rows are the selected crops of the tower's T features, read by the first
// LayerNorm directly (no fp32 copy of them); otherwise x fp32 [n, n_kv, dim]
struct ResCrops { const void* feats; int period, first, per_image, images; };
static int resampler_run(const slime_resampler_desc* d, const float* x, int ldx, const ResCrops* crops, int n, float* out_f32,
                         void* out_t, void* ws, size_t ws_bytes, void* stream) {
    TRY(resampler_validate(d));
    SLIME_REQUIRE((x || crops) && n > 0 && (crops || ldx >= d->dim), "resampler: bad input");
    SLIME_REQUIRE(out_f32 || out_t, "resampler: no output requested");
    const ResPlan p = res_plan(d, n);
    if (!ws || ws_bytes < p.total || ((uintptr_t)ws % 256) != 0) {
        slime_set_error("resampler: workspace %zu B (need %zu, 256-B aligned)", ws_bytes, p.total);
        return SLIME_EWORKSPACE;
    }
    char* w = (char*)ws;
    const int D = d->dim, Rk = n * d->n_kv, Rq = n * d->n_query, dh = D / d->heads, dt = d->dtype;
    // x = ln_kv(x); K input = x + pos (sampler.py:158,164); V input = x
    if (crops)
        TRY(layernorm_crops_launch(crops->feats, dt, d->n_kv, crops->period, crops->first, crops->per_image, crops->images, D, d->ln_kv_w,
                                   d->ln_kv_b, d->eps, w + p.xn, w + p.xk, d->pos_k, d->n_kv, stream));
    else
        TRY(slime_layernorm(x, ldx, Rk, D, d->ln_kv_w, d->ln_kv_b, d->eps, 1, nullptr, w + p.xn, w + p.xk, d->pos_k,
                            d->n_kv, dt, stream));
    TRY(gemm_w(w + p.xk, D, d->w_k, d->w_k_frag, d->b_k, w + p.kp, D, Rk, D, D, dt, SLIME_EPI_BIAS_T, stream));
    TRY(gemm_w(w + p.xn, D, d->w_v, d->w_v_frag, d->b_v, w + p.vp, D, Rk, D, D, dt, SLIME_EPI_BIAS_T, stream));
    TRY(slime_attention(d->q_proj, 0, D, w + p.kp, (long)d->n_kv * D, D, w + p.vp, (long)d->n_kv * D, D, w + p.ctx,
                        (long)d->n_query * D, D, n, d->heads, dh, d->n_query, d->n_kv, dt, stream));
    TRY(gemm_w(w + p.ctx, D, d->w_o, d->w_o_frag, d->b_o, w + p.o32, D, Rq, D, D, dt, SLIME_EPI_BIAS_F32, stream));
    TRY(slime_layernorm((const float*)(w + p.o32), D, Rq, D, d->ln_post_w, d->ln_post_b, d->eps, 1, out_f32, out_t,
                        nullptr, nullptr, 0, dt, stream));
    return SLIME_OK;
}

extern "C" int slime_resampler_forward(const slime_resampler_desc* d, const float* x, int ldx, int n, float* out_f32,
                                       void* out_t, void* ws, size_t ws_bytes, void* stream) {
    SLIME_REQUIRE(x, "resampler: bad input");
    return resampler_run(d, x, ldx, nullptr, n, out_f32, out_t, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// Projector MLP
// ------------------------------------------------------------------------------------------------
static int mlp_validate(const slime_mlp_desc* d) {
    SLIME_REQUIRE(d, "mlp: null descriptor");
    SLIME_REQUIRE(is16(d->dtype), "mlp: dtype must be BF16 or F16");
    SLIME_REQUIRE(d->in_dim == 128 || d->in_dim == 256 || d->in_dim == 1024, "mlp: in_dim=%d unsupported", d->in_dim);
    SLIME_REQUIRE(d->hidden % 128 == 0, "mlp: hidden=%d must be a multiple of 128", d->hidden);
    SLIME_REQUIRE((d->w1 || d->w1_frag) && d->b1 && (d->w2 || d->w2_frag) && d->b2, "mlp: missing weights (row-major or fragment-order)");
    return SLIME_OK;
}

struct MlpPlan { size_t xt, mid, total; };
static MlpPlan mlp_plan(const slime_mlp_desc* d, int rows) {
    MlpPlan p{};
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = align_up(off, 256); off = o + b; return o; };
    p.xt = take((size_t)rows * d->in_dim * 2);
    p.mid = take((size_t)rows * d->hidden * 2);
    p.total = align_up(off, 256);
    return p;
}

extern "C" size_t slime_mlp_workspace_bytes(const slime_mlp_desc* d, int rows) {
    if (!d || rows <= 0) return 0;
    return mlp_plan(d, rows).total;
}

extern "C" int slime_mlp_forward(const slime_mlp_desc* d, const float* x_f32, const void* x_t, int rows, float* out,
                                 void* ws, size_t ws_bytes, void* stream) {
    TRY(mlp_validate(d));
    SLIME_REQUIRE((x_f32 || x_t) && out && rows > 0, "mlp: bad input");
    const MlpPlan p = mlp_plan(d, rows);
    if (!ws || ws_bytes < p.total || ((uintptr_t)ws % 256) != 0) {
        slime_set_error("mlp: workspace %zu B (need %zu, 256-B aligned)", ws_bytes, p.total);
        return SLIME_EWORKSPACE;
    }
    char* w = (char*)ws;
    const void* a = x_t;
    if (!a) {
        TRY(slime_layernorm(x_f32, d->in_dim, rows, d->in_dim, nullptr, nullptr, 0.f, 0, nullptr, w + p.xt, nullptr,
                            nullptr, 0, d->dtype, stream));
        a = w + p.xt;
    }
    TRY(gemm_w(a, d->in_dim, d->w1, d->w1_frag, d->b1, w + p.mid, d->hidden, rows, d->hidden, d->in_dim, d->dtype,
                   SLIME_EPI_BIAS_GELU_T, stream));
    TRY(gemm_w(w + p.mid, d->hidden, d->w2, d->w2_frag, d->b2, out, d->hidden, rows, d->hidden, d->hidden, d->dtype,
                   SLIME_EPI_BIAS_F32, stream));
    return SLIME_OK;
}

// ------------------------------------------------------------------------------------------------
// GatedBlock
// ------------------------------------------------------------------------------------------------
struct GatedPlan { size_t stack, mlp, res, gates, total; };
static GatedPlan gated_plan(const slime_mlp_desc* m, const slime_resampler_desc* r, int n) {
    const size_t rows = (size_t)n * r->n_query;
    GatedPlan p{};
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = align_up(off, 256); off = o + b; return o; };
    p.stack = take(2 * rows * m->in_dim * 2);                  // [T(x) | attn(x)]: the MLP's stacked operand
    p.mlp = take(mlp_plan(m, (int)(2 * rows)).total);
    p.res = take(res_plan(r, n).total);
    p.gates = take(rows * 2 * sizeof(float));
    p.total = align_up(off, 256);
    return p;
}

extern "C" size_t slime_gated_workspace_bytes(const slime_mlp_desc* mlp, const slime_resampler_desc* attn, int n) {
    if (!mlp || !attn || n <= 0) return 0;
    return gated_plan(mlp, attn, n).total;
}

extern "C" int slime_gated_forward(const slime_mlp_desc* mlp, const slime_resampler_desc* attn, const float* w_gate,
                                   int learnable_gated, const float* x, int n, float* out, void* ws, size_t ws_bytes,
                                   void* stream) {
    TRY(mlp_validate(mlp));
    TRY(resampler_validate(attn));
    SLIME_REQUIRE(x && out && n > 0, "gated: bad input");
    SLIME_REQUIRE(attn->dim == mlp->in_dim && attn->n_query == attn->n_kv, "gated: attn must map the token grid onto itself");
    SLIME_REQUIRE(learnable_gated >= 0 || w_gate, "gated: missing w_gate");
    SLIME_REQUIRE(learnable_gated <= 1, "gated: expert index %d", learnable_gated);
    const GatedPlan p = gated_plan(mlp, attn, n);
    if (!ws || ws_bytes < p.total || ((uintptr_t)ws % 256) != 0) {
        slime_set_error("gated: workspace %zu B (need %zu, 256-B aligned)", ws_bytes, p.total);
        return SLIME_EWORKSPACE;
    }
    char* w = (char*)ws;
    const int rows = n * attn->n_query, D = mlp->in_dim, H = mlp->hidden, dt = mlp->dtype;
    const MlpPlan mp = mlp_plan(mlp, 2 * rows);
    const size_t res_ws = res_plan(attn, n).total;
    char* stack = w + p.stack;
    char* attn_t = stack + (size_t)rows * D * 2;
    if (learnable_gated == 0)                                   // expert 0: projection(x)
        return slime_mlp_forward(mlp, x, nullptr, rows, out, w + p.mlp, mp.total, stream);
    TRY(slime_resampler_forward(attn, x, attn->dim, n, nullptr, attn_t, w + p.res, res_ws, stream));
    if (learnable_gated == 1)                                   // expert 1: projection(attn(x))
        return slime_mlp_forward(mlp, nullptr, attn_t, rows, out, w + p.mlp, mp.total, stream);
    // Both experts, mixed by the gate (builder.py:190-206).  projection[0] + GELU run over the stacked rows [T(x) | attn(x)]; the gate
    // mixes the HIDDEN rows (slime_gate_premix: projection[2] is linear and g0 + g1 = 1 / (1 + 1e-6)), so projection[2] runs over ONE
    // row per token and writes the block's output directly -- a third less GEMM work than two complete experts (round 4).
    char* mid = w + p.mlp + mp.mid;
    char* mixed = mid + (size_t)rows * H * 2;
    TRY(slime_layernorm(x, D, rows, D, nullptr, nullptr, 0.f, 0, nullptr, stack, nullptr, nullptr, 0, dt, stream));
    if (mix_in_gemm(mlp)) {
        // the mix in fp32 inside projection[0]'s epilogue (one rounding, no pass over the hidden rows): SLIME_EPI_BIAS_GELU_MIX_T
        float* gates = (float*)(w + p.gates);
        TRY(slime_gate_weights(x, D, w_gate, gates, rows, stream));
        TRY(gemm_mix(stack, attn_t, gates, mlp, mixed, rows, stream));
    } else {
        TRY(gemm_w(stack, D, mlp->w1, mlp->w1_frag, mlp->b1, mid, H, 2 * rows, H, D, dt, SLIME_EPI_BIAS_GELU_T, stream));
        TRY(slime_gate_premix(x, D, w_gate, mid, mixed, mixed, dt, rows, H, stream));
    }
    TRY(gemm_w(mixed, H, mlp->w2, mlp->w2_frag, mlp->b2, out, H, rows, H, H, dt, SLIME_EPI_BIAS_F32, stream));
    return SLIME_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused adapter (GatedBlock on the global crops + post_qformer/MLP/merge on the local crops)
// ------------------------------------------------------------------------------------------------
#ifndef SLIME_OPT_ADAPTER_DIRECT
#define SLIME_OPT_ADAPTER_DIRECT 1      // 0 = round 4's fp32 rows + two merge_rows passes (tools/build_variants.sh A/B)
#endif
int adapter_row_map_launch(int* map, long rows_g, int P, long rows_l, long per_image_local, int g, int nw, int merge, long out_image_stride,
                           void* stream);                                   // rowwise.hip (internal)
struct AdapterPlan { size_t xg32, xl32, stack, e, mlp, res, gates, rmap, total; long rows_g, rows_l, rows_all; int segs_g; };
static AdapterPlan adapter_plan(const slime_mlp_desc* m, const slime_resampler_desc* attn, const slime_resampler_desc* post,
                                int n_images, int n_local, int learnable_gated) {
    AdapterPlan p{};
    const size_t D = m->in_dim, H = m->hidden;
    p.rows_g = (long)n_images * attn->n_kv;
    p.rows_l = post ? (long)n_images * n_local * post->n_query : 0;
    p.segs_g = learnable_gated < 0 ? 2 : 1;
    p.rows_all = p.segs_g * p.rows_g + p.rows_l;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = align_up(off, 256); off = o + b; return o; };
    p.xg32 = take((size_t)p.rows_g * D * 4);
    p.xl32 = take(0);                 // (round 5: the local crops' fp32 copy is gone -- post_qformer's LayerNorm reads the T features)
    p.stack = take((size_t)p.rows_all * D * 2);
    p.e = take((size_t)p.rows_all * H * 4);
    p.mlp = take(mlp_plan(m, (int)p.rows_all).total);
    size_t res = res_plan(attn, n_images).total;
    if (post && n_local > 0) { const size_t r2 = res_plan(post, n_images * n_local).total; if (r2 > res) res = r2; }
    p.res = take(res);
    p.gates = take((size_t)p.rows_g * 2 * sizeof(float));
    p.rmap = take((size_t)(p.rows_g + p.rows_l) * sizeof(int));
    p.total = align_up(off, 256);
    return p;
}

extern "C" size_t slime_adapter_workspace_bytes(const slime_mlp_desc* mlp, const slime_resampler_desc* attn,
                                                const slime_resampler_desc* post, int n_images, int n_local) {
    if (!mlp || !attn || n_images <= 0 || n_local < 0 || (n_local > 0 && !post)) return 0;
    return adapter_plan(mlp, attn, n_local > 0 ? post : nullptr, n_images, n_local, -1).total;
}

extern "C" int slime_adapter_forward(const slime_mlp_desc* mlp, const slime_resampler_desc* attn, const float* w_gate,
                                     int learnable_gated, const slime_resampler_desc* post, const void* feats,
                                     int n_images, int n_local, int nw, int nh, int merge, void* out, int out_dtype,
                                     long out_image_stride, void* ws, size_t ws_bytes, void* stream) {
    TRY(mlp_validate(mlp));
    TRY(resampler_validate(attn));
    SLIME_REQUIRE(feats && out && n_images > 0 && n_local >= 0, "adapter: bad input");
    SLIME_REQUIRE(attn->dim == mlp->in_dim && attn->n_query == attn->n_kv, "adapter: attn must map the token grid onto itself");
    SLIME_REQUIRE(learnable_gated >= 0 || w_gate, "adapter: missing w_gate");
    SLIME_REQUIRE(learnable_gated <= 1, "adapter: expert index %d", learnable_gated);
    SLIME_REQUIRE(attn->dtype == mlp->dtype, "adapter: mixed operand dtypes");
    int g = 0;
    if (n_local > 0) {
        TRY(resampler_validate(post));
        SLIME_REQUIRE(post->dim == mlp->in_dim && post->n_kv == attn->n_kv && post->dtype == mlp->dtype, "adapter: post_qformer does not match the tower grid");
        while (g * g < post->n_query) ++g;
        SLIME_REQUIRE(g * g == post->n_query, "adapter: post_qformer query count %d is not a square grid", post->n_query);
        SLIME_REQUIRE(nw > 0 && nh > 0 && nw * nh == n_local, "adapter: grid %dx%d does not hold %d local crops", nw, nh, n_local);
    } else {
        post = nullptr;
    }
    const int P = attn->n_kv, D = mlp->in_dim, H = mlp->hidden, dt = mlp->dtype;
    const AdapterPlan p = adapter_plan(mlp, attn, post, n_images, n_local, learnable_gated);
    SLIME_REQUIRE(out_image_stride >= P + (long)n_local * (post ? post->n_query : 0), "adapter: out_image_stride %ld too small", out_image_stride);
    if (!ws || ws_bytes < p.total || ((uintptr_t)ws % 256) != 0) {
        slime_set_error("adapter: workspace %zu B (need %zu, 256-B aligned)", ws_bytes, p.total);
        return SLIME_EWORKSPACE;
    }
    char* w = (char*)ws;
    float* xg32 = (float*)(w + p.xg32);
    char* stack = w + p.stack;
    float* e = (float*)(w + p.e);
    const int period = 1 + n_local;

    // stacked MLP input: [x_global | attn(x_global) | post_qformer(x_local)] (segments that are not needed are dropped)
    size_t row = 0;
    long seg_x = -1, seg_attn = -1, seg_local = -1;
    if (learnable_gated != 1) { seg_x = (long)row; row += p.rows_g; }
    if (learnable_gated != 0) { seg_attn = (long)row; row += p.rows_g; }
    if (post) { seg_local = (long)row; row += p.rows_l; }

    // global crops: fp32 copy for the resampler's LayerNorm and the gate logits, T copy straight into the stack
    TRY(slime_select_crops(feats, dt, P, D, period, 0, 1, n_images, xg32, seg_x >= 0 ? stack + (size_t)seg_x * D * 2 : nullptr, stream));
    if (seg_attn >= 0)
        TRY(slime_resampler_forward(attn, xg32, D, n_images, nullptr, stack + (size_t)seg_attn * D * 2, w + p.res,
                                    res_plan(attn, n_images).total, stream));
    if (post) {
        // local crops: post_qformer's first LayerNorm reads them straight from the tower's T features (round 5: no fp32 copy --
        // 75 MB written and read back per 8 x (1+4) step --, no select_crops launch; float(T) either way: bit-identical)
        const ResCrops rc{feats, period, 1, n_local, n_images};
        TRY(resampler_run(post, nullptr, 0, &rc, n_images * n_local, nullptr, stack + (size_t)seg_local * D * 2, w + p.res,
                          res_plan(post, n_images * n_local).total, stream));
    }
    // projection MLP over the stack.  With both experts (learnable_gated < 0) the gate mixes the HIDDEN rows of the two global
    // segments (slime_gate_premix, into the second one) and projection[2] runs over [mixed global | local] = one row per output
    // token; e then holds [global | local] rows.  Otherwise e holds the stack's rows.
    const MlpPlan mp = mlp_plan(mlp, (int)p.rows_all);
    char* mid = w + p.mlp + mp.mid;
    long e_glob = 0, e_local = seg_local;                       // rows of e
    // Round 5: projection[2] stores every output row straight into the token buffer (slime_gemm_args.row_map: global rows to rows
    // [0, P) of their image, local rows to their raster position behind them) in the output dtype -- no fp32 rows `e`, no merge
    // passes (two launches, 151 MB written + read back per 8 x (1+4) step).  The same values, rounded once by the same RNE pack as the
    // merge kernel's: bit-identical to the per-module sequence.  Other output dtypes (a 16-bit type that is not the operand type) keep
    // the fp32 rows + merge path.
    const bool direct = SLIME_OPT_ADAPTER_DIRECT && (out_dtype == SLIME_F32 || out_dtype == dt);
    int* rmap = (int*)(w + p.rmap);
    if (direct) {
        SLIME_REQUIRE((long)n_images * out_image_stride < (1L << 31), "adapter: token buffer too large for the 32-bit row map");
        TRY(adapter_row_map_launch(rmap, p.rows_g, P, p.rows_l, post ? (long)n_local * post->n_query : 1, g > 0 ? g : 1, nw > 0 ? nw : 1, merge,
                                   out_image_stride, stream));
    }
    auto projection2 = [&](const void* A, int rows) -> int {     // rows = [global | local] rows of the hidden activations
        slime_gemm_args a{};
        a.A = A; a.lda = H; a.B = mlp->w2; a.B_frag = mlp->w2_frag; a.bias = mlp->b2; a.M = rows; a.N = H; a.K = H; a.dtype = dt;
        if (direct) { a.C = out; a.ldc = H; a.row_map = rmap; a.epilogue = out_dtype == SLIME_F32 ? SLIME_EPI_BIAS_F32 : SLIME_EPI_BIAS_T; }
        else { a.C = e; a.ldc = H; a.epilogue = SLIME_EPI_BIAS_F32; }
        return slime_gemm_ex(&a, stream);
    };
    if (learnable_gated < 0) {
        char* mixed = mid + (size_t)seg_attn * H * 2;
        if (mix_in_gemm(mlp)) {
            // the mix in fp32 inside projection[0]'s epilogue (SLIME_EPI_BIAS_GELU_MIX_T: one rounding, no pass over the hidden rows,
            // half the global hidden rows written); the local rows follow in their own launch, directly behind the mixed rows
            float* gates = (float*)(w + p.gates);
            TRY(slime_gate_weights(xg32, D, w_gate, gates, (int)p.rows_g, stream));
            TRY(gemm_mix(stack + (size_t)seg_x * D * 2, stack + (size_t)seg_attn * D * 2, gates, mlp, mixed, (int)p.rows_g, stream));
            if (post)
                TRY(gemm_w(stack + (size_t)seg_local * D * 2, D, mlp->w1, mlp->w1_frag, mlp->b1, mid + (size_t)seg_local * H * 2, H,
                           (int)p.rows_l, H, D, dt, SLIME_EPI_BIAS_GELU_T, stream));
        } else {
            TRY(gemm_w(stack, D, mlp->w1, mlp->w1_frag, mlp->b1, mid, H, (int)p.rows_all, H, D, dt, SLIME_EPI_BIAS_GELU_T, stream));
            TRY(slime_gate_premix(xg32, D, w_gate, mid + (size_t)seg_x * H * 2, mixed, mixed, dt, (int)p.rows_g, H, stream));
        }
        TRY(projection2(mixed, (int)(p.rows_g + p.rows_l)));
        e_local = p.rows_g;
    } else {
        TRY(gemm_w(stack, D, mlp->w1, mlp->w1_frag, mlp->b1, mid, H, (int)p.rows_all, H, D, dt, SLIME_EPI_BIAS_GELU_T, stream));
        TRY(projection2(mid, (int)p.rows_all));
    }
    if (!direct) {
        // global tokens -> rows [0, P) of every image: flat cast-copy of P rows per image (nw = P, nh = g = 1, merge = 0)
        TRY(slime_merge_rows_batched(e + (size_t)e_glob * H, P, out, out_dtype, out_image_stride, 0, n_images, P, 1, 1, H, 0, stream));
        if (post)
            TRY(slime_merge_rows_batched(e + (size_t)e_local * H, (long)n_local * post->n_query, out, out_dtype, out_image_stride, P,
                                         n_images, nw, nh, g, H, merge, stream));
    }
    return SLIME_OK;
}
