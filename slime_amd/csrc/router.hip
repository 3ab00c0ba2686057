// router.hip -- text-guided top-p token router of SliME (llava/model/multimodal_resampler/builder.py:
// TextGuidedRouterCosine.forward :186-201 and TextGuidedSampler.forward :248-281, eval path).
//
// scores[t] = sum_l mask[l] * cos(img[t], text[l])       (mean over l when there is no mask)
// torch's cosine_similarity normalises both operands first (x / max(|x|, eps)) and then takes the
// dot product, so the sum over text tokens factorises exactly:
//     scores[t] = (img[t] / max(|img[t]|, eps)) . u,   u = sum_l mask[l] * text[l] / max(|text[l]|, eps)
// which turns a [T x L x H] contraction into two HBM-bound passes (L*H + T*H reads).
// The selection (softmax / temperature, descending sort, cumulative sum, "<= top-p plus one",
// ascending re-sort) runs in ONE workgroup with the sorted list in LDS; the cumulative sum is
// evaluated sequentially in sort order like torch's CPU cumsum.
#include "common.h"

// w[l] = mask[l] / max(|text[l]|, eps)   (one wave per text token)
// (all kernels below: blockIdx.y = image of the batch; per-image strides are passed in elements)
__global__ void __launch_bounds__(256) router_text_weight_kernel(const float* text, const unsigned char* mask, float* w,
                                                                 int L, int H, float eps, int have_mask, float no_mask_w, int w_stride) {
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= L) return;
    text += (size_t)blockIdx.y * L * H;
    if (have_mask) mask += (size_t)blockIdx.y * L;
    w += (size_t)blockIdx.y * w_stride;
    const float* r = text + (size_t)l * H;
    float s = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(r + c);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float m = have_mask ? (mask[l] ? 1.f : 0.f) : no_mask_w;
        w[l] = m / fmaxf(sqrtf(s), eps);
    }
}

// u[h] = sum_l w[l] * text[l][h]   (thread per column, coalesced over h; fixed summation order)
__global__ void __launch_bounds__(256) router_text_dir_kernel(const float* text, const float* w, float* u, int L, int H, int w_stride) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    text += (size_t)blockIdx.y * L * H;
    w += (size_t)blockIdx.y * w_stride;
    u += (size_t)blockIdx.y * w_stride;
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += w[l] * text[(size_t)l * H + h];
    u[h] = s;
}

// scores[t] = (img[t] . u) / max(|img[t]|, eps)
// image b: rows row_off[b] .. row_off[b] + n_rows[b] - 1 of img (NULL tables: one image of T rows at row 0)
__global__ void __launch_bounds__(256) router_scores_kernel(const float* img, const float* u, float* scores, int T, int H, float eps,
                                                            const long long* row_off, const int* n_rows, int u_stride) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n_rows) {
        if (t >= n_rows[blockIdx.y]) return;
        img += (size_t)row_off[blockIdx.y] * H;
        u += (size_t)blockIdx.y * u_stride;
        scores += (size_t)blockIdx.y * T;
    }
    if (t >= T) return;
    const float* r = img + (size_t)t * H;
    float d = 0.f, s = 0.f;
    for (int c = lane * 4; c < H; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(r + c);
        const float4 q = *reinterpret_cast<const float4*>(u + c);
        d += v.x * q.x + v.y * q.y + v.z * q.z + v.w * q.w;
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    d = wave_sum(d); s = wave_sum(s);
    if (lane == 0) scores[t] = d / fmaxf(sqrtf(s), eps);
}

#define ROUTER_MAX_T 4096
__global__ void __launch_bounds__(1024) router_select_kernel(const float* scores, int T, float temp, float topp,
                                                             int* keep_idx, int* keep_count, float* probs_out, const int* n_rows) {
    if (n_rows) {                                     // batched: image blockIdx.x, padded [B, T] layouts
        scores += (size_t)blockIdx.x * T; keep_idx += (size_t)blockIdx.x * T; keep_count += blockIdx.x;
        if (probs_out) probs_out += (size_t)blockIdx.x * T;
        T = n_rows[blockIdx.x];
        if (T <= 0) { if (threadIdx.x == 0) *keep_count = 0; return; }
    }
    __shared__ float key[ROUTER_MAX_T];
    __shared__ int idx[ROUTER_MAX_T];
    __shared__ float red[32];
    __shared__ int scan[1024];
    __shared__ int s_k;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int P = 1;
    while (P < T) P <<= 1;
    // softmax(scores / temp)
    float mx = -INFINITY;
    for (int i = tid; i < T; i += 1024) mx = fmaxf(mx, scores[i] / temp);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = red[0];
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red[i]);
    __syncthreads();
    float sm = 0.f;
    for (int i = tid; i < P; i += 1024) {
        const float e = i < T ? expf(scores[i] / temp - mx) : 0.f;
        key[i] = e; idx[i] = i;
        sm += e;
    }
    sm = wave_sum(sm);
    if (lane == 0) red[wv] = sm;
    __syncthreads();
    sm = 0.f;
    for (int i = 0; i < 16; ++i) sm += red[i];
    for (int i = tid; i < P; i += 1024) {
        key[i] = i < T ? key[i] / sm : -1.f;          // padding sorts last
        if (probs_out && i < T) probs_out[i] = key[i];
    }
    __syncthreads();
    // bitonic sort: descending probability, ties by ascending index
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const float a = key[i], b = key[p];
                    const int ia = idx[i], ib = idx[p];
                    const bool a_first = (a > b) || (a == b && ia < ib);     // a belongs before b (descending)
                    const bool up = (i & k) == 0;
                    if (up ? !a_first : a_first) { key[i] = b; key[p] = a; idx[i] = ib; idx[p] = ia; }
                }
            }
            __syncthreads();
        }
    }
    // sequential cumulative sum in sort order; k = #(cum <= topp); keep k+1 unless everything is kept
    if (tid == 0) {
        float c = 0.f;
        int k = 0;
        for (int i = 0; i < T; ++i) { c += key[i]; if (c <= topp) ++k; }
        s_k = k < T ? k + 1 : T;
    }
    __syncthreads();
    const int nkeep = s_k;
    // flags over original positions, then an ascending compaction
    for (int i = tid; i < P; i += 1024) key[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < nkeep; i += 1024) key[idx[i]] = 1.f;
    __syncthreads();
    const int per = (T + 1023) / 1024;                // consecutive positions per thread
    int local = 0;
    for (int r = 0; r < per; ++r) { const int i = tid * per + r; if (i < T && key[i] != 0.f) ++local; }
    scan[tid] = local;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    int pos = scan[tid] - local;
    for (int r = 0; r < per; ++r) { const int i = tid * per + r; if (i < T && key[i] != 0.f) keep_idx[pos++] = i; }
    if (tid == 0) *keep_count = nkeep;
}

extern "C" int slime_router_scores(const float* img, int T, const float* text, int L, const unsigned char* mask, int H,
                                   float* scores, float* ws /* L + H floats */, void* stream) {
    SLIME_REQUIRE(img && text && scores && ws && T > 0 && L > 0 && H % 4 == 0, "router_scores: bad input");
    hipStream_t s = (hipStream_t)stream;
    float* w = ws;
    float* u = ws + ((L + 3) / 4 * 4);
    hipLaunchKernelGGL(router_text_weight_kernel, dim3((L + 3) / 4), dim3(256), 0, s, text, mask, w, L, H, 1e-8f,
                       mask ? 1 : 0, 1.0f / (float)L, 0);
    hipLaunchKernelGGL(router_text_dir_kernel, dim3((H + 255) / 256), dim3(256), 0, s, text, w, u, L, H, 0);
    hipLaunchKernelGGL(router_scores_kernel, dim3((T + 3) / 4), dim3(256), 0, s, img, u, scores, T, H, 1e-8f, nullptr, nullptr, 0);
    SLIME_CHECK_LAUNCH("router_scores");
    return SLIME_OK;
}

extern "C" int slime_router_select(const float* scores, int T, float temp, float topp, int* keep_idx, int* keep_count,
                                   float* probs_out, void* stream) {
    SLIME_REQUIRE(scores && keep_idx && keep_count, "router_select: null pointer");
    SLIME_REQUIRE(T > 0 && T <= ROUTER_MAX_T, "router_select: T=%d outside 1..%d", T, ROUTER_MAX_T);
    SLIME_REQUIRE(temp > 0.f, "router_select: temperature must be positive");
    hipLaunchKernelGGL(router_select_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, scores, T, temp, topp,
                       keep_idx, keep_count, probs_out, nullptr);
    SLIME_CHECK_LAUNCH("router_select");
    return SLIME_OK;
}

// ---- the B images of a step in one launch sequence (same arithmetic per image; results identical to B single calls) ----
extern "C" size_t slime_router_batched_workspace_floats(int B, int L, int H) {
    if (B <= 0 || L <= 0 || H <= 0) return 0;
    const size_t per = (size_t)((L + 3) / 4 * 4) > (size_t)H ? (size_t)((L + 3) / 4 * 4) : (size_t)H;
    return 2 * per * B;
}

extern "C" int slime_router_scores_batched(const float* img, const long long* row_off, const int* n_rows, int B, int T_max,
                                           const float* text, int L, const unsigned char* mask, int H, float* scores,
                                           float* ws, void* stream) {
    SLIME_REQUIRE(img && row_off && n_rows && text && scores && ws && B > 0 && B <= 65535 && T_max > 0 && L > 0 && H % 4 == 0,
                  "router_scores_batched: bad input");
    hipStream_t s = (hipStream_t)stream;
    const int per = (L + 3) / 4 * 4 > H ? (L + 3) / 4 * 4 : H;     // one stride for the w [L] and u [H] tables of an image
    float* w = ws;
    float* u = ws + (size_t)per * B;
    hipLaunchKernelGGL(router_text_weight_kernel, dim3((L + 3) / 4, B), dim3(256), 0, s, text, mask, w, L, H, 1e-8f,
                       mask ? 1 : 0, 1.0f / (float)L, per);
    hipLaunchKernelGGL(router_text_dir_kernel, dim3((H + 255) / 256, B), dim3(256), 0, s, text, w, u, L, H, per);
    hipLaunchKernelGGL(router_scores_kernel, dim3((T_max + 3) / 4, B), dim3(256), 0, s, img, u, scores, T_max, H, 1e-8f, row_off,
                       n_rows, per);
    SLIME_CHECK_LAUNCH("router_scores_batched");
    return SLIME_OK;
}

extern "C" int slime_router_select_batched(const float* scores, const int* n_rows, int B, int T_max, float temp, float topp,
                                           int* keep_idx, int* keep_count, void* stream) {
    SLIME_REQUIRE(scores && n_rows && keep_idx && keep_count && B > 0, "router_select_batched: bad input");
    SLIME_REQUIRE(T_max > 0 && T_max <= ROUTER_MAX_T, "router_select_batched: T_max=%d outside 1..%d", T_max, ROUTER_MAX_T);
    SLIME_REQUIRE(temp > 0.f, "router_select_batched: temperature must be positive");
    hipLaunchKernelGGL(router_select_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, scores, T_max, temp, topp, keep_idx,
                       keep_count, (float*)nullptr, n_rows);
    SLIME_CHECK_LAUNCH("router_select_batched");
    return SLIME_OK;
}
