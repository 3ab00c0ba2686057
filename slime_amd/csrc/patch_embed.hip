// patch_embed.hip -- the tower's front end in ONE launch: patch-embed conv (MFMA) + class token + position table + pre-LayerNorm.
//
//   h[n, 1 + P, D] = pre_layrnorm( cat(class_embedding, conv14x14/14(pixels)) + position_embedding )
//   (HF CLIPVisionEmbeddings.forward, modeling_clip.py:148-154, 209-217, and pre_layrnorm :642; reached through
//    llava/model/multimodal_encoder/clip_encoder.py:51,55)
//
// Until round 4 this was three launches and two HBM round trips: im2col_kernel wrote the [n P, 640] conv operand, a GEMM wrote
// the fp32 conv output, embed_prenorm_kernel read it back (84 us per 20 crops for 15 GF).  Here a workgroup owns TWO ROWS OF
// PATCHES of one crop (2 x 24 = 48 output rows = three 16-row MFMA tiles, no padding rows) and ALL D output columns, so the
// LayerNorm statistics of a row never leave the workgroup:
//   1. the 2 x 3 x 14 image rows it needs are whole 336-pixel rows: fetched with 16-byte coalesced loads into LDS (fp32 pixels are
//      rounded to T on the way, as `images.to(dtype=self.dtype)` does, clip_encoder.py:55) -- 56 KB, each pixel read from HBM once;
//   2. the 14 x 14 patch tiles are re-tiled INSIDE LDS into the MFMA operand image X[48][kpad] (column k = (c, ky, kx), the order of
//      Conv2d.weight.flatten(1); columns >= 3 p^2 zero) through a k -> tile-offset table; row pitch kpad + 8 elements, so the
//      16 rows of a fragment read (16 lanes x 16 B, same k) start 4 banks apart: conflict-free ds_read_b128 without a swizzle;
//   3. wave w owns columns [w CW, (w + 1) CW) of all 48 rows: the weight fragments come straight from the fragment-order image
//      (slime_gemm_pack_b of the [D, kpad] conv weight) into VGPRs, one k-step ahead; X fragments are shared by all waves;
//      3 x CW/16 MFMAs of 16x16x32 per k-step, product transposed (mfma(W, X)) so that a lane ends with 8 consecutive columns;
//   4. epilogue in registers: + position row, two-pass LayerNorm statistics (lane -> the wave's four 16-lane rows by permlane
//      swaps -> across the waves through 3 KB of LDS, summed in wave order: fixed, batch-independent), affine, then the outputs
//      the layer stack wants: the fp32 residual rows (or their split form: T(h) + one byte), T(h) and its per-64-column partial sums
//      (the first folded LayerNorm of slime_gemm_ex).  Row 0 of every crop (class token + position 0) is input independent; the
//      workgroup that owns a crop's first patch rows writes it.
#include "gemm_shared.h"

namespace {

__device__ __forceinline__ float pe_rows4_allsum(float x) {      // sum over lanes l, l+16, l+32, l+48 (in every one of them)
    const unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned w = __float_as_uint(y);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

struct PEArgs {
    const void* px; const char* wf; const float* cls; const float* pos; const float* ln_w; const float* ln_b; float eps;
    float* h; char* x16; char* lo; float* stats;
    int n, image, patch, kpad, g;
};

__host__ __device__ constexpr size_t pe_align16(size_t x) { return (x + 15) & ~(size_t)15; }
constexpr int PE_MT = 3;                 // 16-row MFMA tiles per workgroup: two rows of <= 24 patches
constexpr int PE_ROWS = 16 * PE_MT;

// One 1 + P row written from fp32 values held as VPL per lane (column = (i * 64 + lane) * VEC + j): the class-token row.
template <typename T, int D>
__device__ __forceinline__ void pe_cls_row(const PEArgs& a, const long grow, const int lane) {
    constexpr int VPL = D / 64, VEC = (VPL >= 4) ? 4 : 2, NV = VPL / VEC, LPG = 64 / VEC;
    float v[VPL];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[i * VEC + j] = a.cls[c + j] + a.pos[c + j];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { const float d = __fsub_rn(v[i], mean); q = __fmaf_rn(d, d, q); }
    const float rstd = rsqrtf(__fmaf_rn(wave_sum(q), 1.0f / D, a.eps));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * VEC;
        float sx = 0.f, sq = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; j += 2) {
            const float y0 = __fmaf_rn(__fmul_rn(__fsub_rn(v[i * VEC + j], mean), rstd), a.ln_w[c + j], a.ln_b[c + j]);
            const float y1 = __fmaf_rn(__fmul_rn(__fsub_rn(v[i * VEC + j + 1], mean), rstd), a.ln_w[c + j + 1], a.ln_b[c + j + 1]);
            if (a.h) { a.h[grow * D + c + j] = y0; a.h[grow * D + c + j + 1] = y1; }
            const unsigned pk = T::pack2(y0, y1);
            const float r0 = T::lo(pk), r1 = T::hi(pk);
            if (a.x16) *reinterpret_cast<unsigned*>(a.x16 + ((size_t)grow * D + c + j) * 2) = pk;
            if (a.lo) {                                   // the stream's lower part: one signed byte per element (common.h resid_delta)
                a.lo[(size_t)grow * D + c + j] = (char)resid_delta<T>(y0, r0);
                a.lo[(size_t)grow * D + c + j + 1] = (char)resid_delta<T>(y1, r1);
            }
            sx += r0; sx += r1; sq = fmaf(r0, r0, sq); sq = fmaf(r1, r1, sq);
        }
        if (a.stats) {
#pragma unroll
            for (int off = LPG / 2; off > 0; off >>= 1) { sx += __shfl_xor(sx, off); sq += __shfl_xor(sq, off); }
            if ((lane & (LPG - 1)) == 0)
                *reinterpret_cast<float2*>(a.stats + ((size_t)grow * (D / 64) + (c >> 6)) * 2) = make_float2(sx, sq);
        }
    }
}

template <typename T, typename PixT, int D, int NW>
__global__ void __launch_bounds__(NW * 64) patch_embed_kernel(PEArgs a) {
    constexpr int CW = D / NW, NT = CW / 64, NF = CW / 16, NP = NF / 2, NTHR = NW * 64;
    static_assert(CW % 64 == 0, "a wave owns whole 64-column groups (LayerNorm partial sums are wave local)");
    extern __shared__ __attribute__((aligned(16))) char smem_pe[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lg = lane >> 4, li = lane & 15;
    const int g = a.g, patch = a.patch, image = a.image, kpad = a.kpad, P = g * g;
    const int pairs = (g + 1) / 2;
    const int crop = blockIdx.x / pairs, pair = blockIdx.x % pairs;
    const int prow0 = 2 * pair, nprow = min(2, g - prow0);            // patch rows of this workgroup
    const int rows_valid = nprow * g;                                  // <= 48 output rows
    const int pp = patch * patch, kreal = 3 * pp;
    const int npx_rows = 2 * 3 * patch;                                // image rows in the LDS tile: (patch row, channel, ky)
    const int pitch = (kpad + 8) * 2;                                  // bytes per X row

    unsigned short* tile = reinterpret_cast<unsigned short*>(smem_pe);                       // [2 * 3 * patch][image] T
    char* X = smem_pe + pe_align16((size_t)npx_rows * image * 2);                           // [48][kpad + 8] T
    unsigned short* lut = reinterpret_cast<unsigned short*>(X + (size_t)PE_ROWS * pitch);     // [kpad]
    float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(lut) + pe_align16((size_t)kpad * 2));   // [NW][48]

    // ---- 1. pixels -> LDS (coalesced 16-byte loads of whole image rows), k -> offset table ---------------------------------
    for (int k = tid; k < kpad; k += NTHR) {
        unsigned short v = 0xffffu;
        if (k < kreal) { const int c = k / pp, r = k % pp; v = (unsigned short)((c * patch + r / patch) * image + r % patch); }
        lut[k] = v;
    }
    {
        constexpr int EPC = 16 / (int)sizeof(PixT);
        const int cpr = image / EPC;
        const PixT* px = reinterpret_cast<const PixT*>(a.px);
        for (int i = tid; i < npx_rows * cpr; i += NTHR) {
            const int r = i / cpr, q = i % cpr;
            const int pr = r / (3 * patch), rc = r % (3 * patch), c = rc / patch, ky = rc % patch;
            unsigned short* dst = tile + (size_t)r * image + q * EPC;
            if (pr < nprow) {
                const PixT* src = px + (((size_t)crop * 3 + c) * image + (size_t)(prow0 + pr) * patch + ky) * image + (size_t)q * EPC;
                if constexpr (sizeof(PixT) == 4) {
                    const float4 f = *reinterpret_cast<const float4*>(src);
                    *reinterpret_cast<u32x2*>(dst) = u32x2{T::pack2(f.x, f.y), T::pack2(f.z, f.w)};
                } else {
                    *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
                }
            } else {
                if constexpr (sizeof(PixT) == 4) *reinterpret_cast<u32x2*>(dst) = u32x2{0u, 0u};
                else *reinterpret_cast<u32x4*>(dst) = u32x4{0u, 0u, 0u, 0u};
            }
        }
    }
    __syncthreads();

    // ---- 2. re-tile 14 x 14 patches into the operand image X[row = (patch row, px)][k] -------------------------------------
    {
        const int chunks = kpad / 8;
        const int prow_stride = 3 * patch * image;                   // tile elements per patch row
        for (int i = tid; i < PE_ROWS * chunks; i += NTHR) {
            const int r = i / chunks, ch = i % chunks;
            unsigned w[4] = {0u, 0u, 0u, 0u};
            if (r < rows_valid) {
                const int pr = r / g, pxi = r % g;
                const unsigned short* tb = tile + pr * prow_stride + pxi * patch;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned short o0 = lut[ch * 8 + 2 * j], o1 = lut[ch * 8 + 2 * j + 1];
                    const unsigned lo = o0 == 0xffffu ? 0u : tb[o0];
                    const unsigned hi = o1 == 0xffffu ? 0u : tb[o1];
                    w[j] = lo | (hi << 16);
                }
            }
            *reinterpret_cast<u32x4*>(X + (size_t)r * pitch + ch * 16) = u32x4{w[0], w[1], w[2], w[3]};
        }
    }
    __syncthreads();

    // ---- 3. MFMA main loop: acc[mi][f] = 16 rows x 16 columns (fragment f of the wave's CW columns) -------------------------
    f32x4 acc[PE_MT][NF];
#pragma unroll
    for (int mi = 0; mi < PE_MT; ++mi)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[mi][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int KS = kpad / 32;                                          // even: kpad % 64 == 0
    // Fragment-order weights (slime_gemm_pack_b): 16-byte unit ((t KS + s) 4 + f) 64 + lane for 64-column tile t, k-step s.  The wave
    // streams its NT tiles' fragments into two register sets, ONE K-STEP AHEAD of the MFMAs that use them.  The loads are written
    // out (SGPR base + lane offset + immediate, destination "+v", counted vmcnt): left to hipcc, every fragment load was sunk to
    // its first use behind an s_waitcnt vmcnt(0) -- eight exposed L2 round trips per k-step, 93 us per 20 crops instead of ~30.
    const char* wtile = a.wf + ((size_t)(wave * NT) * KS) * 4096;
    const unsigned wlane = lane * 16;
    const char* xb = X + (size_t)li * pitch + lg * 16;
    u32x4 wA[NF], wB[NF];
    auto load_w = [&](u32x4 (&w)[NF], int s) {
        static_for<0, NT>([&](auto t) {
            const char* base = uniform_ptr(wtile + ((size_t)t.value * KS + s) * 4096);
            gload16_frag<0>(w[t.value * 4 + 0], wlane, base);
            gload16_frag<1024>(w[t.value * 4 + 1], wlane, base);
            gload16_frag<2048>(w[t.value * 4 + 2], wlane, base);
            gload16_frag<3072>(w[t.value * 4 + 3], wlane, base);
        });
    };
    auto mul_step = [&](u32x4 (&w)[NF], int s) {
        u32x4 xf[PE_MT];
#pragma unroll
        for (int mi = 0; mi < PE_MT; ++mi) xf[mi] = *reinterpret_cast<const u32x4*>(xb + (size_t)mi * 16 * pitch + s * 64);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int mi = 0; mi < PE_MT; ++mi) acc[mi][f] = T::mfma16(w[f], xf[mi], acc[mi][f]);
    };
    auto wait_w = [&](u32x4 (&w)[NF], auto in_flight) {               // the NF loads issued BEFORE the last `in_flight` ones have landed
        vm_wait_frag<decltype(in_flight)::value>(w[0]);
#pragma unroll
        for (int f = 1; f < NF; ++f) asm volatile("" : "+v"(w[f]));
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int f = 0; f < NF; ++f) { wA[f] = u32x4{0u, 0u, 0u, 0u}; wB[f] = u32x4{0u, 0u, 0u, 0u}; }
    load_w(wA, 0);
    for (int s = 0; s < KS; s += 2) {
        load_w(wB, s + 1);
        wait_w(wA, std::integral_constant<int, NF>{});
        mul_step(wA, s);
        load_w(wA, min(s + 2, KS - 1));                                // the last iteration re-requests a resident line (no branch in the loop)
        wait_w(wB, std::integral_constant<int, NF>{});
        mul_step(wB, s + 1);
    }
    wait_w(wA, std::integral_constant<int, 0>{});                      // nothing in flight past this point: the registers are free again

    // ---- 4. epilogue: + position row, LayerNorm over the D columns of each row, outputs ------------------------------------
    // lane (lg, li) holds, for row 16 mi + li and fragment pair p: columns cw0 + 32 p + 8 lg + e, e = 0..7 = acc[mi][2p + (e >> 2)][e & 3]
    const int cw0 = wave * CW;
    long grow[PE_MT];
    bool ok[PE_MT];
    float part[PE_MT];
#pragma unroll
    for (int mi = 0; mi < PE_MT; ++mi) {
        const int r = 16 * mi + li;
        ok[mi] = r < rows_valid;
        const int t = 1 + prow0 * g + min(r, rows_valid - 1);         // token index (clamped: masked rows compute on a valid row's table)
        grow[mi] = (long)crop * (P + 1) + t;
        const float* pr = a.pos + (size_t)t * D + cw0 + 8 * lg;
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float4 p0 = *reinterpret_cast<const float4*>(pr + 32 * p), p1 = *reinterpret_cast<const float4*>(pr + 32 * p + 4);
            acc[mi][2 * p][0] += p0.x; acc[mi][2 * p][1] += p0.y; acc[mi][2 * p][2] += p0.z; acc[mi][2 * p][3] += p0.w;
            acc[mi][2 * p + 1][0] += p1.x; acc[mi][2 * p + 1][1] += p1.y; acc[mi][2 * p + 1][2] += p1.z; acc[mi][2 * p + 1][3] += p1.w;
#pragma unroll
            for (int k = 0; k < 4; ++k) { s += acc[mi][2 * p][k]; s += acc[mi][2 * p + 1][k]; }
        }
        part[mi] = pe_rows4_allsum(s);
    }
    auto all_waves = [&](float (&v)[PE_MT]) {                        // v[mi] (this wave's share of row 16 mi + li) -> sum over the waves
        __syncthreads();                                              // the previous round's reads are done
        if (lg == 0) {
#pragma unroll
            for (int mi = 0; mi < PE_MT; ++mi) red[wave * PE_ROWS + 16 * mi + li] = v[mi];
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < PE_MT; ++mi) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += red[w * PE_ROWS + 16 * mi + li];      // wave order: fixed
            v[mi] = s;
        }
    };
    all_waves(part);
    float mean[PE_MT];
#pragma unroll
    for (int mi = 0; mi < PE_MT; ++mi) {
        mean[mi] = part[mi] * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float d = __fsub_rn(acc[mi][f][k], mean[mi]); q = __fmaf_rn(d, d, q); }
        part[mi] = pe_rows4_allsum(q);
    }
    all_waves(part);
    float gw[NP][8], gb[NP][8];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int c = cw0 + 32 * p + 8 * lg;
        const float4 w0 = *reinterpret_cast<const float4*>(a.ln_w + c), w1 = *reinterpret_cast<const float4*>(a.ln_w + c + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(a.ln_b + c), b1 = *reinterpret_cast<const float4*>(a.ln_b + c + 4);
        gw[p][0] = w0.x; gw[p][1] = w0.y; gw[p][2] = w0.z; gw[p][3] = w0.w; gw[p][4] = w1.x; gw[p][5] = w1.y; gw[p][6] = w1.z; gw[p][7] = w1.w;
        gb[p][0] = b0.x; gb[p][1] = b0.y; gb[p][2] = b0.z; gb[p][3] = b0.w; gb[p][4] = b1.x; gb[p][5] = b1.y; gb[p][6] = b1.z; gb[p][7] = b1.w;
    }
#pragma unroll
    for (int mi = 0; mi < PE_MT; ++mi) {
        const float rstd = rsqrtf(__fmaf_rn(part[mi], 1.0f / D, a.eps));       // (operations written out: no contraction left to the compiler)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float sx = 0.f, sq = 0.f;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int p = 2 * t + h2;
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = __fmaf_rn(__fmul_rn(__fsub_rn(acc[mi][2 * p + (e >> 2)][e & 3], mean[mi]), rstd), gw[p][e], gb[p][e]);
                const size_t off = (size_t)grow[mi] * D + cw0 + 32 * p + 8 * lg;
                const u32x4 w = pack8<T>(y);
                float r[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { r[2 * e] = T::lo(w[e]); r[2 * e + 1] = T::hi(w[e]); }
                if (ok[mi]) {
                    if (a.h) {
                        *reinterpret_cast<f32x4*>(a.h + off) = f32x4{y[0], y[1], y[2], y[3]};
                        *reinterpret_cast<f32x4*>(a.h + off + 4) = f32x4{y[4], y[5], y[6], y[7]};
                    }
                    if (a.x16) *reinterpret_cast<u32x4*>(a.x16 + off * 2) = w;
                    if (a.lo) {                           // the stream's lower part: one signed byte per element (common.h resid_delta)
                        int d[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) d[e] = resid_delta<T>(y[e], r[e]);
                        *reinterpret_cast<u32x2*>(a.lo + off) = u32x2{pack_bytes(d[0], d[1], d[2], d[3]), pack_bytes(d[4], d[5], d[6], d[7])};
                    }
                }
                // partial sums of the ROUNDED row per 64-column group (what the consuming GEMM multiplies), lane-local first
#pragma unroll
                for (int e = 0; e < 8; ++e) { sx += r[e]; sq = fmaf(r[e], r[e], sq); }
            }
            sx = pe_rows4_allsum(sx);
            sq = pe_rows4_allsum(sq);
            if (a.stats && lg == 0 && ok[mi])
                *reinterpret_cast<float2*>(a.stats + ((size_t)grow[mi] * (D / 64) + ((cw0 + 64 * t) >> 6)) * 2) = make_float2(sx, sq);
        }
    }
    // the crop's class-token row (token 0): input independent, written by the workgroup that owns the first patch rows
    if (pair == 0 && wave == 0) pe_cls_row<T, D>(a, (long)crop * (P + 1), lane);
}

template <typename T, typename PixT, int D, int NW>
int launch_pe(const PEArgs& a, size_t lds, hipStream_t s) {
    auto kern = patch_embed_kernel<T, PixT, D, NW>;
    SLIME_SET_LDS_ONCE(kern, 160 * 1024, "patch_embed");       // the geometry (image, patch, kpad) is a run-time argument: opt in to the maximum
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.n * ((a.g + 1) / 2))), dim3(NW * 64), lds, s, a);
    SLIME_CHECK_LAUNCH("patch_embed");
    return SLIME_OK;
}

template <typename T, typename PixT>
int launch_pe_d(const PEArgs& a, int D, size_t lds, hipStream_t s) {
    switch (D) {
        case 128: return launch_pe<T, PixT, 128, 2>(a, lds, s);
        case 256: return launch_pe<T, PixT, 256, 4>(a, lds, s);
        case 1024: return launch_pe<T, PixT, 1024, 8>(a, lds, s);
    }
    slime_set_error("patch_embed: D=%d unsupported (128, 256, 1024)", D);
    return SLIME_EINVAL;
}

}  // namespace

// Geometry limits of the fused front end, shared with vit_validate / slime_vit_check (api.hip) so that a tower the front end cannot
// run is refused when it is PACKED, with the limit named, not at its first forward: two rows of at most PE_ROWS / 2 patches per
// workgroup, whole image rows of 16-byte pieces, a 16-bit LDS offset table over the 2 x 3 x patch staged image rows, <= 160 KiB of LDS.
int slime_patch_embed_geometry(int image, int patch, int kpad, int D, size_t* lds_out) {
    SLIME_REQUIRE(patch > 0 && image > 0 && image % patch == 0 && image % 8 == 0,
                  "patch_embed: image %d must be a multiple of the patch %d and of 8 (16-byte pixel pieces)", image, patch);
    SLIME_REQUIRE(kpad % 64 == 0 && kpad >= 3 * patch * patch, "patch_embed: kpad=%d (3 * patch^2 rounded up to a multiple of 64)", kpad);
    SLIME_REQUIRE(D == 128 || D == 256 || D == 1024, "patch_embed: D=%d unsupported (128, 256, 1024)", D);
    const int g = image / patch;
    SLIME_REQUIRE(2 * g <= PE_ROWS, "patch_embed: %d patches per side -- the front end holds two rows of at most %d patches per workgroup "
                  "(CLIP-L/14-336: 24; a 448 / 14 tower would need 32)", g, PE_ROWS / 2);
    SLIME_REQUIRE((size_t)6 * patch * image < 65535, "patch_embed: 6 * patch * image = %zu elements of staged image rows exceed the 16-bit offset table",
                  (size_t)6 * patch * image);
    const int nw = D == 128 ? 2 : D == 256 ? 4 : 8;
    const size_t lds = pe_align16((size_t)6 * patch * image * 2) + (size_t)PE_ROWS * (kpad + 8) * 2 + pe_align16((size_t)kpad * 2) +
                       (size_t)nw * PE_ROWS * sizeof(float);
    SLIME_REQUIRE(lds <= 160 * 1024, "patch_embed: %zu bytes of LDS needed (160 KiB per CU)", lds);
    if (lds_out) *lds_out = lds;
    return SLIME_OK;
}

extern "C" int slime_patch_embed_prenorm(const void* pixels, int pix_dtype, const void* patch_w_frag, const float* cls, const float* pos,
                                         const float* ln_w, const float* ln_b, float eps, float* h, void* x16, void* lo8, float* stats,
                                         int dtype, int n, int image, int patch, int kpad, int D, void* stream) {
    SLIME_REQUIRE(pixels && patch_w_frag && cls && pos && ln_w && ln_b && n > 0, "patch_embed: bad input");
    SLIME_REQUIRE(h || x16, "patch_embed: no output requested (h and / or x16)");
    SLIME_REQUIRE(!lo8 || x16, "patch_embed: the split residual's lower part (lo8) comes with its upper part x16");
    SLIME_REQUIRE(!stats || x16, "patch_embed: stats are the partial sums of x16");
    SLIME_REQUIRE(dtype == SLIME_BF16 || dtype == SLIME_F16, "patch_embed: dtype must be BF16 or F16");
    SLIME_REQUIRE(pix_dtype == SLIME_F32 || pix_dtype == dtype, "patch_embed: 16-bit pixels must already be in the tower dtype");
    size_t lds = 0;
    if (const int rc = slime_patch_embed_geometry(image, patch, kpad, D, &lds)) return rc;
    const int g = image / patch;
    SLIME_REQUIRE(((uintptr_t)pixels % 16) == 0 && ((uintptr_t)patch_w_frag % 16) == 0 && ((uintptr_t)pos % 16) == 0 &&
                  ((uintptr_t)ln_w % 16) == 0 && ((uintptr_t)ln_b % 16) == 0 && (!h || (uintptr_t)h % 16 == 0) &&
                  (!x16 || (uintptr_t)x16 % 16 == 0) && (!lo8 || (uintptr_t)lo8 % 8 == 0) && (!stats || (uintptr_t)stats % 8 == 0),
                  "patch_embed: pointers must be 16-byte aligned");
    PEArgs a{pixels, (const char*)patch_w_frag, cls, pos, ln_w, ln_b, eps, h, (char*)x16, (char*)lo8, stats, n, image, patch, kpad, g};
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SLIME_F16)
        return pix_dtype == SLIME_F32 ? launch_pe_d<F16, float>(a, D, lds, s) : launch_pe_d<F16, unsigned short>(a, D, lds, s);
    return pix_dtype == SLIME_F32 ? launch_pe_d<BF16, float>(a, D, lds, s) : launch_pe_d<BF16, unsigned short>(a, D, lds, s);
}
