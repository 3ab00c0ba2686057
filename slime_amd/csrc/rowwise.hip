// rowwise.hip -- HBM-bound row kernels: LayerNorm (+cast/+position term), gate mix, row gather / spatial merge,
// tile+normalise.  (The patch-embed front end -- im2col, class token, position table, pre-LayerNorm -- is ONE MFMA kernel
// since round 5: patch_embed.hip.)
// One wave (64 lanes) owns one row; loads are 16 B/lane (float4) wherever the width allows.
#include "common.h"

template <typename T> __device__ __forceinline__ unsigned short to_t1(float x) {
    return (unsigned short)(T::pack2(x, 0.f) & 0xffffu);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over fp32 rows; VPL = values per lane (D = 64*VPL).  Two-pass variance in registers.
// ------------------------------------------------------------------------------------------------
struct LnArgs {
    const float* x; int ldx; int rows;
    const float* w; const float* b; float eps; int normalize;
    float* out_f32; void* out_t; void* out_t2; const float* add; int add_period;
    // round 5 (fused adapter): the rows come straight from the tower's T features [crops, P, D] -- row r = (selected crop j = r / P, token
    // r % P), crop j = image (j / per_image) * period + first + j % per_image, as slime_select_crops picks them -- and are widened to
    // fp32 here (the same float(T) the separate select_crops copy produced: bit-identical), instead of being copied out as fp32 first
    const char* xt; int P, period, first, per_image;
};

template <typename T, int VPL>
__global__ void __launch_bounds__(256) layernorm_kernel(LnArgs a) {
    constexpr int D = 64 * VPL;
    constexpr int VEC = (VPL >= 4) ? 4 : 2;          // floats per load
    constexpr int NV = VPL / VEC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    float v[VPL];
    if (a.xt) {
        const long j = row / a.P, tok = row % a.P;
        const long crop = (j / a.per_image) * a.period + a.first + j % a.per_image;
        const char* xr = a.xt + ((size_t)crop * a.P + tok) * D * 2;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * VEC;
            if constexpr (VEC == 4) {
                const u32x2 t = *reinterpret_cast<const u32x2*>(xr + (size_t)c * 2);
                v[i * 4 + 0] = T::lo(t[0]); v[i * 4 + 1] = T::hi(t[0]); v[i * 4 + 2] = T::lo(t[1]); v[i * 4 + 3] = T::hi(t[1]);
            } else {
                const unsigned t = *reinterpret_cast<const unsigned*>(xr + (size_t)c * 2);
                v[i * 2 + 0] = T::lo(t); v[i * 2 + 1] = T::hi(t);
            }
        }
    } else {
        const float* xr = a.x + (size_t)row * a.ldx;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * VEC;
            if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(xr + c);
                v[i * 4 + 0] = t.x; v[i * 4 + 1] = t.y; v[i * 4 + 2] = t.z; v[i * 4 + 3] = t.w;
            } else {
                const float2 t = *reinterpret_cast<const float2*>(xr + c);
                v[i * 2 + 0] = t.x; v[i * 2 + 1] = t.y;
            }
        }
    }
    if (a.normalize) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) s += v[i];
        const float mean = wave_sum(s) * (1.0f / D);
        // the operations are written out (no contraction left to the compiler): adding the T-input branch above moved the generated
        // code of this block and with it the last bit of every output -- pinned since round 5
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { const float d = __fsub_rn(v[i], mean); q = __fmaf_rn(d, d, q); }
        const float rstd = rsqrtf(__fmaf_rn(wave_sum(q), 1.0f / D, a.eps));
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * VEC;
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                v[i * VEC + j] = __fmaf_rn(__fmul_rn(__fsub_rn(v[i * VEC + j], mean), rstd), a.w[c + j], a.b[c + j]);
        }
    }
    const float* addr = a.out_t2 ? a.add + (size_t)(row % a.add_period) * D : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * VEC;
        if (a.out_f32) {
            float* o = a.out_f32 + (size_t)row * D + c;
            if constexpr (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]);
            else *reinterpret_cast<float2*>(o) = make_float2(v[i * 2], v[i * 2 + 1]);
        }
        if (a.out_t) {
            char* o = reinterpret_cast<char*>(a.out_t) + ((size_t)row * D + c) * 2;
            if constexpr (VEC == 4) {
                u32x2 p = {T::pack2(v[i * 4], v[i * 4 + 1]), T::pack2(v[i * 4 + 2], v[i * 4 + 3])};
                *reinterpret_cast<u32x2*>(o) = p;
            } else {
                *reinterpret_cast<unsigned*>(o) = T::pack2(v[i * 2], v[i * 2 + 1]);
            }
        }
        if (a.out_t2) {
            char* o = reinterpret_cast<char*>(a.out_t2) + ((size_t)row * D + c) * 2;
            if constexpr (VEC == 4) {
                u32x2 p = {T::pack2(v[i * 4] + addr[c], v[i * 4 + 1] + addr[c + 1]),
                           T::pack2(v[i * 4 + 2] + addr[c + 2], v[i * 4 + 3] + addr[c + 3])};
                *reinterpret_cast<u32x2*>(o) = p;
            } else {
                *reinterpret_cast<unsigned*>(o) = T::pack2(v[i * 2] + addr[c], v[i * 2 + 1] + addr[c + 1]);
            }
        }
    }
}

template <typename T>
static int launch_ln(const LnArgs& a, int D, hipStream_t s) {
    const dim3 grid((a.rows + 3) / 4), block(256);
    switch (D) {
        case 128: hipLaunchKernelGGL((layernorm_kernel<T, 2>), grid, block, 0, s, a); break;
        case 256: hipLaunchKernelGGL((layernorm_kernel<T, 4>), grid, block, 0, s, a); break;
        case 1024: hipLaunchKernelGGL((layernorm_kernel<T, 16>), grid, block, 0, s, a); break;
        default: slime_set_error("layernorm: D=%d unsupported (128, 256, 1024)", D); return SLIME_EINVAL;
    }
    SLIME_CHECK_LAUNCH("layernorm");
    return SLIME_OK;
}

extern "C" int slime_layernorm(const float* x, int ldx, int rows, int D, const float* w, const float* b,
                               float eps, int normalize, float* out_f32, void* out_t, void* out_t2,
                               const float* add, int add_period, int dtype, void* stream) {
    SLIME_REQUIRE(x && rows > 0 && ldx >= D, "layernorm: bad input");
    SLIME_REQUIRE(!normalize || (w && b), "layernorm: missing affine parameters");
    SLIME_REQUIRE(!out_t2 || (add && add_period > 0), "layernorm: out_t2 needs add/add_period");
    SLIME_REQUIRE(ldx % 4 == 0, "layernorm: ldx must be a multiple of 4");
    LnArgs a{x, ldx, rows, w, b, eps, normalize, out_f32, out_t, out_t2, add, add_period, nullptr, 0, 0, 0, 0};
    if (dtype == SLIME_F16) return launch_ln<F16>(a, D, (hipStream_t)stream);
    return launch_ln<BF16>(a, D, (hipStream_t)stream);
}

// internal (not part of the C ABI): slime_layernorm over the rows of SELECTED crops of the tower's T features (see LnArgs.xt)
int layernorm_crops_launch(const void* feats, int dtype, int P, int period, int first, int per_image, int images, int D, const float* w,
                           const float* b, float eps, void* out_t, void* out_t2, const float* add, int add_period, void* stream) {
    SLIME_REQUIRE(feats && w && b && P > 0 && per_image > 0 && images > 0 && first >= 0 && first + per_image <= period, "layernorm_crops: bad input");
    LnArgs a{nullptr, 0, images * per_image * P, w, b, eps, 1, nullptr, out_t, out_t2, add, add_period, (const char*)feats, P, period, first, per_image};
    if (dtype == SLIME_F16) return launch_ln<F16>(a, D, (hipStream_t)stream);
    return launch_ln<BF16>(a, D, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// Gate mix: (g0,g1) = softmax(x @ w_gate) / (sum + 1e-6);  out = g0*e0 + g1*e1.  One wave per row.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void store4_cast(void* base, size_t elem, int out_dtype, float4 v) {
    if (out_dtype == SLIME_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + elem) = v;
    } else {
        u32x2 p = {T::pack2(v.x, v.y), T::pack2(v.z, v.w)};
        *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(base) + elem * 2) = p;
    }
}

// out row of input row r: (r / rows_per_group) * group_stride + row0 + r % rows_per_group  (token buffer of an image batch)
__global__ void __launch_bounds__(256) gate_mix_kernel(const float* x, int D, const float* wg, const float* e0,
                                                       const float* e1, void* out, int out_dtype, int rows, int H,
                                                       int rows_per_group, long group_stride, long row0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    float l0 = 0.f, l1 = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float xv = xr[c];
        const float2 w = *reinterpret_cast<const float2*>(wg + 2 * c);
        l0 += xv * w.x; l1 += xv * w.y;
    }
    l0 = wave_sum(l0); l1 = wave_sum(l1);
    const float m = fmaxf(l0, l1);
    const float p0 = expf(l0 - m), p1 = expf(l1 - m);
    const float ps = p0 + p1;
    const float s0 = p0 / ps, s1 = p1 / ps;            // softmax
    const float den = s0 + s1 + 1e-6f;                 // top-2-of-2 renormalisation
    const float g0 = s0 / den, g1 = s1 / den;
    const float* a = e0 + (size_t)row * H;
    const float* b = e1 + (size_t)row * H;
    const size_t orow = (size_t)(row / rows_per_group) * group_stride + row0 + row % rows_per_group;
    for (int c = lane * 4; c < H; c += 256) {
        const float4 u = *reinterpret_cast<const float4*>(a + c);
        const float4 w = *reinterpret_cast<const float4*>(b + c);
        const float4 v = make_float4(g0 * u.x + g1 * w.x, g0 * u.y + g1 * w.y, g0 * u.z + g1 * w.z, g0 * u.w + g1 * w.w);
        if (out_dtype == SLIME_F16) store4_cast<F16>(out, orow * H + c, out_dtype, v);
        else store4_cast<BF16>(out, orow * H + c, out_dtype, v);
    }
}

extern "C" int slime_gate_mix_ex(const float* x, int D, const float* w_gate, const float* e0, const float* e1, void* out,
                                 int out_dtype, int rows, int H, int rows_per_group, long group_stride, long row0,
                                 void* stream) {
    SLIME_REQUIRE(x && w_gate && e0 && e1 && out && rows > 0 && H % 4 == 0 && rows_per_group > 0, "gate_mix: bad input");
    SLIME_REQUIRE(out_dtype == SLIME_F32 || out_dtype == SLIME_BF16 || out_dtype == SLIME_F16, "gate_mix: bad out dtype");
    hipLaunchKernelGGL(gate_mix_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, D, w_gate, e0, e1, out,
                       out_dtype, rows, H, rows_per_group, group_stride, row0);
    SLIME_CHECK_LAUNCH("gate_mix");
    return SLIME_OK;
}

// The gate pair of every row, for the GEMM epilogue that mixes the two experts' hidden rows in registers (SLIME_EPI_BIAS_GELU_MIX_T).
// The arithmetic is gate_mix_kernel's, instruction for instruction.
__global__ void __launch_bounds__(256) gate_weights_kernel(const float* x, int D, const float* wg, float* out, int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    float l0 = 0.f, l1 = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float xv = xr[c];
        const float2 w = *reinterpret_cast<const float2*>(wg + 2 * c);
        l0 += xv * w.x; l1 += xv * w.y;
    }
    l0 = wave_sum(l0); l1 = wave_sum(l1);
    const float m = fmaxf(l0, l1);
    const float p0 = expf(l0 - m), p1 = expf(l1 - m);
    const float ps = p0 + p1;
    const float s0 = p0 / ps, s1 = p1 / ps;            // softmax
    const float den = s0 + s1 + 1e-6f;                 // top-2-of-2 renormalisation
    if (lane == 0) *reinterpret_cast<float2*>(out + 2 * (size_t)row) = make_float2(s0 / den, s1 / den);
}

extern "C" int slime_gate_weights(const float* x, int D, const float* w_gate, float* out, int rows, void* stream) {
    SLIME_REQUIRE(x && w_gate && out && rows > 0 && D > 0, "gate_weights: bad input");
    SLIME_REQUIRE(((uintptr_t)out % 8) == 0, "gate_weights: out must be 8-byte aligned");
    hipLaunchKernelGGL(gate_weights_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, D, w_gate, out, rows);
    SLIME_CHECK_LAUNCH("gate_weights");
    return SLIME_OK;
}

// The same gates applied to the MLP's HIDDEN rows (round 4): the projector's second Linear is linear and g0 + g1 = 1 / (1 + 1e-6), so
// W2 (g0 a0 + g1 a1) + b2 replaces g0 (W2 a0 + b2) + g1 (W2 a1 + b2) -- the second Linear then runs over ONE row per token instead of
// two.  a0 / a1 / out: T [rows, H]; out may be a1 (each thread reads its elements before it writes them).  The gate arithmetic is
// gate_mix_kernel's, instruction for instruction.
template <typename T>
__global__ void __launch_bounds__(256) gate_premix_kernel(const float* x, int D, const float* wg, const char* a0, const char* a1,
                                                          char* out, int rows, int H) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    float l0 = 0.f, l1 = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float xv = xr[c];
        const float2 w = *reinterpret_cast<const float2*>(wg + 2 * c);
        l0 += xv * w.x; l1 += xv * w.y;
    }
    l0 = wave_sum(l0); l1 = wave_sum(l1);
    const float m = fmaxf(l0, l1);
    const float p0 = expf(l0 - m), p1 = expf(l1 - m);
    const float ps = p0 + p1;
    const float s0 = p0 / ps, s1 = p1 / ps;            // softmax
    const float den = s0 + s1 + 1e-6f;                 // top-2-of-2 renormalisation
    const float g0 = s0 / den, g1 = s1 / den;
    const size_t base = (size_t)row * H * 2;
    for (int c = lane * 8; c < H; c += 512) {
        const u32x4 u = *reinterpret_cast<const u32x4*>(a0 + base + (size_t)c * 2);
        const u32x4 w = *reinterpret_cast<const u32x4*>(a1 + base + (size_t)c * 2);
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = g0 * T::lo(u[k]) + g1 * T::lo(w[k]);
            v[2 * k + 1] = g0 * T::hi(u[k]) + g1 * T::hi(w[k]);
        }
        *reinterpret_cast<u32x4*>(out + base + (size_t)c * 2) = pack8<T>(v);
    }
}

extern "C" int slime_gate_premix(const float* x, int D, const float* w_gate, const void* a0, const void* a1, void* out, int dtype,
                                 int rows, int H, void* stream) {
    SLIME_REQUIRE(x && w_gate && a0 && a1 && out && rows > 0 && H > 0 && H % 8 == 0, "gate_premix: bad input");
    SLIME_REQUIRE(dtype == SLIME_BF16 || dtype == SLIME_F16, "gate_premix: dtype must be BF16 or F16");
    if (dtype == SLIME_F16)
        hipLaunchKernelGGL(gate_premix_kernel<F16>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, D, w_gate,
                           (const char*)a0, (const char*)a1, (char*)out, rows, H);
    else
        hipLaunchKernelGGL(gate_premix_kernel<BF16>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, D, w_gate,
                           (const char*)a0, (const char*)a1, (char*)out, rows, H);
    SLIME_CHECK_LAUNCH("gate_premix");
    return SLIME_OK;
}

extern "C" int slime_gate_mix(const float* x, int D, const float* w_gate, const float* e0, const float* e1,
                              float* out, int rows, int H, void* stream) {
    return slime_gate_mix_ex(x, D, w_gate, e0, e1, out, SLIME_F32, rows, H, rows > 0 ? rows : 1, 0, 0, stream);
}

// ------------------------------------------------------------------------------------------------
// Row gather / spatial merge with cast.  One wave per output row, 4 floats per lane per step.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void store_row_cast(const float* src, void* dst_base, size_t dst_row, int C, int out_dtype, int lane) {
    if (out_dtype == SLIME_F32) {
        float* d = reinterpret_cast<float*>(dst_base) + dst_row * C;
        for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<float4*>(d + c) = *reinterpret_cast<const float4*>(src + c);
    } else {
        char* d = reinterpret_cast<char*>(dst_base) + dst_row * C * 2;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 t = *reinterpret_cast<const float4*>(src + c);
            u32x2 p = {T::pack2(t.x, t.y), T::pack2(t.z, t.w)};
            *reinterpret_cast<u32x2*>(d + (size_t)c * 2) = p;
        }
    }
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const float* in, int rows_in, int row_off, void* out,
                                                          int out_dtype, long total, int rows_out, int C) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= total) return;
    const long g = r / rows_out, i = r % rows_out;
    const float* src = in + ((size_t)g * rows_in + row_off + i) * C;
    if (out_dtype == SLIME_F16) store_row_cast<F16>(src, out, (size_t)r, C, out_dtype, lane);
    else store_row_cast<BF16>(src, out, (size_t)r, C, out_dtype, lane);
}

extern "C" int slime_gather_rows(const float* in, int rows_in, int row_off, void* out, int out_dtype, int groups,
                                 int rows_out, int C, void* stream) {
    SLIME_REQUIRE(in && out && groups > 0 && rows_out > 0 && C % 4 == 0, "gather_rows: bad input");
    SLIME_REQUIRE(row_off >= 0 && row_off + rows_out <= rows_in, "gather_rows: window outside the group");
    const long total = (long)groups * rows_out;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       in, rows_in, row_off, out, out_dtype, total, rows_out, C);
    SLIME_CHECK_LAUNCH("gather_rows");
    return SLIME_OK;
}

// The same from the split residual stream (tower, round 5; ABI 7: the lower part is one signed byte per element): in = join(hi, lo8), 8 elements per lane per step.
template <typename T>
__global__ void __launch_bounds__(256) gather_rows_split_kernel(const char* hi, const char* lo, int rows_in, int row_off, void* out,
                                                                int out_dtype, long total, int rows_out, int C) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= total) return;
    const long g = r / rows_out, i = r % rows_out;
    const size_t src = ((size_t)g * rows_in + row_off + i) * C;
    for (int c = lane * 8; c < C; c += 512) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(hi + (src + c) * 2);
        const u32x2 b = *reinterpret_cast<const u32x2*>(lo + src + c);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = resid_join<T>(T::lo(a[e]), sext_byte(b[e >> 1], 2 * (e & 1)));
            v[2 * e + 1] = resid_join<T>(T::hi(a[e]), sext_byte(b[e >> 1], 2 * (e & 1) + 1));
        }
        if (out_dtype == SLIME_F32) {
            float* d = reinterpret_cast<float*>(out) + (size_t)r * C + c;
            *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else if (out_dtype == SLIME_F16) {
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(out) + ((size_t)r * C + c) * 2) = pack8<F16>(v);
        } else {
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(out) + ((size_t)r * C + c) * 2) = pack8<BF16>(v);
        }
    }
}

extern "C" int slime_gather_rows_split(const void* hi, const void* lo, int dtype, int rows_in, int row_off, void* out, int out_dtype,
                                       int groups, int rows_out, int C, void* stream) {
    SLIME_REQUIRE(hi && lo && out && groups > 0 && rows_out > 0 && C % 8 == 0, "gather_rows_split: bad input");
    SLIME_REQUIRE(dtype == SLIME_BF16 || dtype == SLIME_F16, "gather_rows_split: the stream halves are BF16 or F16");
    SLIME_REQUIRE(out_dtype == SLIME_F32 || out_dtype == SLIME_BF16 || out_dtype == SLIME_F16, "gather_rows_split: bad out dtype");
    SLIME_REQUIRE(row_off >= 0 && row_off + rows_out <= rows_in, "gather_rows_split: window outside the group");
    SLIME_REQUIRE(((uintptr_t)hi % 16) == 0 && ((uintptr_t)lo % 8) == 0 && ((uintptr_t)out % 16) == 0, "gather_rows_split: hi / out 16-byte, lo8 8-byte aligned");
    const long total = (long)groups * rows_out;
    const dim3 grid((unsigned)((total + 3) / 4)), block(256);
    if (dtype == SLIME_F16)
        hipLaunchKernelGGL(gather_rows_split_kernel<F16>, grid, block, 0, (hipStream_t)stream, (const char*)hi, (const char*)lo, rows_in, row_off,
                           out, out_dtype, total, rows_out, C);
    else
        hipLaunchKernelGGL(gather_rows_split_kernel<BF16>, grid, block, 0, (hipStream_t)stream, (const char*)hi, (const char*)lo, rows_in, row_off,
                           out, out_dtype, total, rows_out, C);
    SLIME_CHECK_LAUNCH("gather_rows_split");
    return SLIME_OK;
}

__global__ void __launch_bounds__(256) merge_rows_kernel(const float* in, long in_image_stride, void* out, int out_dtype,
                                                         long out_image_stride, long dst_row0, int nw, int nh, int g,
                                                         int C, int merge) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);       // input row of this image: (crop k, qy, qx)
    const long total = (long)nw * nh * g * g;
    if (r >= total) return;
    long dst = r;
    if (merge) {
        const int qx = (int)(r % g), qy = (int)((r / g) % g), k = (int)(r / ((long)g * g));
        const int gx = k % nw, gy = k / nw;
        dst = ((long)(gy * g + qy) * nw + gx) * g + qx;
    }
    const float* src = in + ((size_t)blockIdx.y * in_image_stride + (size_t)r) * C;
    const size_t drow = (size_t)blockIdx.y * out_image_stride + dst_row0 + dst;
    if (out_dtype == SLIME_F16) store_row_cast<F16>(src, out, drow, C, out_dtype, lane);
    else store_row_cast<BF16>(src, out, drow, C, out_dtype, lane);
}

extern "C" int slime_merge_rows_batched(const float* in, long in_image_stride, void* out, int out_dtype,
                                        long out_image_stride, long dst_row0, int images, int nw, int nh, int g, int C,
                                        int merge, void* stream) {
    SLIME_REQUIRE(in && out && images > 0 && nw > 0 && nh > 0 && g > 0 && C % 4 == 0, "merge_rows: bad input");
    const long total = (long)nw * nh * g * g;
    hipLaunchKernelGGL(merge_rows_kernel, dim3((unsigned)((total + 3) / 4), images), dim3(256), 0, (hipStream_t)stream,
                       in, in_image_stride, out, out_dtype, out_image_stride, dst_row0, nw, nh, g, C, merge);
    SLIME_CHECK_LAUNCH("merge_rows");
    return SLIME_OK;
}

// Row map of the fused adapter (round 5): GEMM row r of projection[2]'s output [global rows of every image | local rows of every image]
// -> row of the [images, out_image_stride, H] token buffer -- the arithmetic of the two merge_rows launches above as an int32 table,
// so that the GEMM's epilogue stores every row where it belongs (slime_gemm_args.row_map) and the fp32 round trip + merge pass go.
__global__ void __launch_bounds__(256) adapter_row_map_kernel(int* map, long rows_g, int P, long rows_l, long per_image_local, int g, int nw,
                                                              int merge, long out_image_stride) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows_g + rows_l) return;
    long dst;
    if (r < rows_g) {
        dst = (r / P) * out_image_stride + r % P;
    } else {
        const long q = r - rows_g, b = q / per_image_local, rr = q % per_image_local;
        long d = rr;
        if (merge) {
            const int qx = (int)(rr % g), qy = (int)((rr / g) % g), k = (int)(rr / ((long)g * g));
            const int gx = k % nw, gy = k / nw;
            d = ((long)(gy * g + qy) * nw + gx) * g + qx;
        }
        dst = b * out_image_stride + P + d;
    }
    map[r] = (int)dst;
}

// internal (not part of the C ABI: the version script keeps it local)
int adapter_row_map_launch(int* map, long rows_g, int P, long rows_l, long per_image_local, int g, int nw, int merge, long out_image_stride,
                           void* stream) {
    const long total = rows_g + rows_l;
    if (total <= 0) return SLIME_OK;
    hipLaunchKernelGGL(adapter_row_map_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, map, rows_g, P, rows_l,
                       per_image_local, g, nw, merge, out_image_stride);
    SLIME_CHECK_LAUNCH("adapter_row_map");
    return SLIME_OK;
}

extern "C" int slime_merge_rows(const float* in, void* out, int out_dtype, long dst_row0, int nw, int nh, int g,
                                int C, int merge, void* stream) {
    return slime_merge_rows_batched(in, 0, out, out_dtype, 0, dst_row0, 1, nw, nh, g, C, merge, stream);
}

// ------------------------------------------------------------------------------------------------
// Crop gather for the adapter: T features [crops, P, C] -> fp32 (and optionally T) rows of the selected
// crops: output crop j = image (j / per_image) * period + first + j % per_image.  One wave per row,
// 16 B per lane per step.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) select_crops_kernel(const char* feats, int P, int C, int period, int first,
                                                           int per_image, long rows, float* out_f32, char* out_t) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long j = r / P, i = r % P;
    const long crop = (j / per_image) * period + first + j % per_image;
    const char* src = feats + ((size_t)crop * P + i) * C * 2;
    for (int c = lane * 8; c < C; c += 512) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + (size_t)c * 2);
        if (out_t) *reinterpret_cast<u32x4*>(out_t + ((size_t)r * C + c) * 2) = v;
        if (out_f32) {
            float* d = out_f32 + (size_t)r * C + c;
            *reinterpret_cast<float4*>(d) = make_float4(T::lo(v[0]), T::hi(v[0]), T::lo(v[1]), T::hi(v[1]));
            *reinterpret_cast<float4*>(d + 4) = make_float4(T::lo(v[2]), T::hi(v[2]), T::lo(v[3]), T::hi(v[3]));
        }
    }
}

extern "C" int slime_select_crops(const void* feats, int dtype, int P, int C, int period, int first, int per_image,
                                  int images, float* out_f32, void* out_t, void* stream) {
    SLIME_REQUIRE(feats && (out_f32 || out_t) && P > 0 && C % 8 == 0 && per_image > 0 && images > 0, "select_crops: bad input");
    SLIME_REQUIRE(first >= 0 && first + per_image <= period, "select_crops: window outside the image's crops");
    SLIME_REQUIRE(dtype == SLIME_BF16 || dtype == SLIME_F16, "select_crops: features must be BF16 or F16");
    const long rows = (long)images * per_image * P;
    dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == SLIME_F16) hipLaunchKernelGGL(select_crops_kernel<F16>, grid, dim3(256), 0, (hipStream_t)stream, (const char*)feats, P, C, period, first, per_image, rows, out_f32, (char*)out_t);
    else hipLaunchKernelGGL(select_crops_kernel<BF16>, grid, dim3(256), 0, (hipStream_t)stream, (const char*)feats, P, C, period, first, per_image, rows, out_f32, (char*)out_t);
    SLIME_CHECK_LAUNCH("select_crops");
    return SLIME_OK;
}

// ------------------------------------------------------------------------------------------------
// Tile + normalise: uint8 HWC canvas -> planar normalised crops.  A workgroup handles one canvas
// row segment of one crop: the 3*crop interleaved bytes are read coalesced, de-interleaved through
// LDS, and each channel row is written coalesced.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) tile_normalize_kernel(const uint8_t* canvas, long canvas_image_stride, int Hc, int Wc,
                                                             int crop, float3 mean, float3 sd, void* out, long out_image_crops,
                                                             int out_dtype) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t* line = reinterpret_cast<uint8_t*>(smem);
    const int tiles_x = Wc / crop;
    const int y = blockIdx.x % crop;
    const int tile = blockIdx.x / crop;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    // image b (blockIdx.y): canvas b -> crops [b * out_image_crops + tile]
    const uint8_t* src = canvas + (size_t)blockIdx.y * canvas_image_stride + ((size_t)(ty * crop + y) * Wc + (size_t)tx * crop) * 3;
    for (int i = threadIdx.x; i < crop * 3; i += blockDim.x) line[i] = src[i];
    __syncthreads();
    const float m[3] = {mean.x, mean.y, mean.z}, sdv[3] = {sd.x, sd.y, sd.z};
    for (int i = threadIdx.x; i < crop * 3; i += blockDim.x) {
        const int c = i / crop, x = i % crop;
        // HF image_transforms: rescale = float32(float64(u8) * (1/255)); normalise = (x - mean) / std in fp32
        const float v = ((float)((double)line[x * 3 + c] * (1.0 / 255.0)) - m[c]) / sdv[c];
        const size_t o = ((((size_t)blockIdx.y * out_image_crops + tile) * 3 + c) * crop + y) * crop + x;
        if (out_dtype == SLIME_F32) reinterpret_cast<float*>(out)[o] = v;
        else reinterpret_cast<unsigned short*>(out)[o] = to_t1<T>(v);
    }
}

extern "C" int slime_tile_normalize_batched(const uint8_t* canvas, int images, long canvas_image_stride, int Hc, int Wc,
                                            int crop, const float* mean3, const float* std3, void* out,
                                            long out_image_crops, int out_dtype, void* stream) {
    SLIME_REQUIRE(canvas && out && mean3 && std3, "tile_normalize: null pointer (mean3/std3 are HOST pointers)");
    SLIME_REQUIRE(crop > 0 && Hc % crop == 0 && Wc % crop == 0, "tile_normalize: canvas %dx%d is not a multiple of %d", Hc, Wc, crop);
    SLIME_REQUIRE(images > 0 && images <= 65535, "tile_normalize: %d images", images);
    const int tiles = (Hc / crop) * (Wc / crop);
    SLIME_REQUIRE(images == 1 || out_image_crops >= tiles, "tile_normalize: out_image_crops %ld < %d tiles", out_image_crops, tiles);
    const float3 mean = make_float3(mean3[0], mean3[1], mean3[2]);
    const float3 istd = make_float3(std3[0], std3[1], std3[2]);
    const dim3 grid(tiles * crop, images), block(256);
    const size_t lds = (size_t)crop * 3;
    if (out_dtype == SLIME_F16) hipLaunchKernelGGL((tile_normalize_kernel<F16>), grid, block, lds, (hipStream_t)stream, canvas, canvas_image_stride, Hc, Wc, crop, mean, istd, out, out_image_crops, out_dtype);
    else hipLaunchKernelGGL((tile_normalize_kernel<BF16>), grid, block, lds, (hipStream_t)stream, canvas, canvas_image_stride, Hc, Wc, crop, mean, istd, out, out_image_crops, out_dtype);
    SLIME_CHECK_LAUNCH("tile_normalize");
    return SLIME_OK;
}

extern "C" int slime_tile_normalize(const uint8_t* canvas, int Hc, int Wc, int crop, const float* mean3,
                                    const float* std3, void* out, int out_dtype, void* stream) {
    return slime_tile_normalize_batched(canvas, 1, 0, Hc, Wc, crop, mean3, std3, out, 0, out_dtype, stream);
}
