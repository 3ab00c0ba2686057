// prefill.hip -- the step after the visual hot path (BASELINE configs 4 / 5): splicing the visual
// tokens into the text embeddings and the Llama-3 prefill self-attention over the spliced sequence.
//
//   slime_splice_rows        new_input_embeds of prepare_inputs_labels_for_multimodal (llava/model/llava_arch.py:343-459): every
//                            output row is a row of the embedding table, a row of the image features, or zero padding.  The
//                            index plan is integer host logic (slime_amd/model/llava_arch.py); ALL tensor movement -- what the
//                            reference does with per-sequence torch.cat / split / stack calls -- is this one launch.
//   slime_rope               RoPE on the q and k heads of a packed qkv buffer, in place (HF rotate_half convention: pairs
//                            (i, i + d/2); fp32 cos/sin of pos * inv_freq), q additionally scaled by head_dim^-0.5 * log2 e
//                            (the attention kernels work in log2 units).
//   slime_prefill_attention  causal grouped-query attention (head_dim 128) over the un-padded tokens [start, start + len) of
//                            every sequence -- llava/train/llama_flash_attn_monkey_patch.py:65-90 (repeat_kv, unpad_input,
//                            flash_attn_unpadded_qkvpacked_func(causal=True), pad_input): a workgroup owns one kv head and a
//                            block of 256 / group query rows and serves all `group` query heads of that kv head from ONE copy
//                            of the K / V chunk in LDS (repeat_kv never materialises); rows outside the token range are zero.
//   slime_llama_attn_forward the LlamaAttention.forward of the monkey patch (:16-93): fused q/k/v projection GEMM, RoPE, the
//                            attention above, o_proj GEMM.
//
// The attention kernel follows attention.hip's transposed formulation (S^T = K Q^T, O^T = V^T P^T: a lane owns one query
// column, softmax statistics are lane-local, P never touches LDS) with K/V chunks of 192 rows staged through registers
// (issue early / write late: the next chunk is in flight during the current chunk's arithmetic).
#include "common.h"

// ------------------------------------------------------------------------------------------------ splice
__device__ __forceinline__ float load_as_float(const void* p, int dtype, size_t i) {
    if (dtype == SLIME_F32) return reinterpret_cast<const float*>(p)[i];
    const unsigned short h = reinterpret_cast<const unsigned short*>(p)[i];
    if (dtype == SLIME_BF16) return __uint_as_float((unsigned)h << 16);
    return (float)__builtin_bit_cast(_Float16, h);
}
__device__ __forceinline__ void store_from_float(void* p, int dtype, size_t i, float v) {
    if (dtype == SLIME_F32) reinterpret_cast<float*>(p)[i] = v;
    else if (dtype == SLIME_BF16) reinterpret_cast<unsigned short*>(p)[i] = (unsigned short)(BF16::pack2(v, 0.f) & 0xffffu);
    else reinterpret_cast<unsigned short*>(p)[i] = (unsigned short)(F16::pack2(v, 0.f) & 0xffffu);
}

// One wave per output row.  src >= 0: row src of `table`; src <= -2: row (-2 - src) of `feats`; src == -1: zeros.
// Equal source / destination dtypes are copied bit for bit (16 B per lane per step when the row allows).
__global__ void __launch_bounds__(256) splice_rows_kernel(const void* table, int table_dtype, const void* feats, int feats_dtype,
                                                          const long long* src, void* out, int out_dtype, long rows, int H,
                                                          long table_rows, long feat_rows) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long long s = src[r];
    const int es_out = out_dtype == SLIME_F32 ? 4 : 2;
    char* o = reinterpret_cast<char*>(out) + (size_t)r * H * es_out;
    // a source row outside its tensor is never dereferenced: the row is zeroed (the host plan raises on such ids first, as
    // nn.Embedding does; this is the device-side backstop for direct C-ABI callers)
    if (s == -1 || (s >= 0 && s >= table_rows) || (s <= -2 && -2 - s >= feat_rows)) {
        for (int b = lane * 16; b < H * es_out; b += 1024) *reinterpret_cast<u32x4*>(o + b) = u32x4{0u, 0u, 0u, 0u};
        return;
    }
    const void* base = s >= 0 ? table : feats;
    const int dt = s >= 0 ? table_dtype : feats_dtype;
    const size_t row = s >= 0 ? (size_t)s : (size_t)(-2 - s);
    if (dt == out_dtype) {
        const char* p = reinterpret_cast<const char*>(base) + row * (size_t)H * es_out;
        for (int b = lane * 16; b < H * es_out; b += 1024) *reinterpret_cast<u32x4*>(o + b) = *reinterpret_cast<const u32x4*>(p + b);
        return;
    }
    for (int c = lane; c < H; c += 64) store_from_float(out, out_dtype, (size_t)r * H + c, load_as_float(base, dt, row * (size_t)H + c));
}

extern "C" int slime_splice_rows(const void* table, int table_dtype, long table_rows, const void* feats, int feats_dtype,
                                 long feat_rows, const int64_t* src, void* out, int out_dtype, long rows, int H, void* stream) {
    SLIME_REQUIRE(src && out && rows > 0 && H > 0, "splice_rows: bad input");
    SLIME_REQUIRE(table || feats, "splice_rows: neither an embedding table nor image features");
    SLIME_REQUIRE(table_rows >= 0 && feat_rows >= 0, "splice_rows: negative source size");
    auto ok = [](int dt) { return dt == SLIME_F32 || dt == SLIME_BF16 || dt == SLIME_F16; };
    SLIME_REQUIRE(ok(out_dtype) && (!table || ok(table_dtype)) && (!feats || ok(feats_dtype)), "splice_rows: bad dtype");
    SLIME_REQUIRE((H * (out_dtype == SLIME_F32 ? 4 : 2)) % 16 == 0, "splice_rows: rows must be multiples of 16 bytes (H=%d)", H);
    SLIME_REQUIRE(((uintptr_t)out % 16) == 0 && ((uintptr_t)table % 16) == 0 && ((uintptr_t)feats % 16) == 0, "splice_rows: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(splice_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, table, table_dtype,
                       feats, feats_dtype, reinterpret_cast<const long long*>(src), out, out_dtype, rows, H,
                       table ? table_rows : 0, feats ? feat_rows : 0);
    SLIME_CHECK_LAUNCH("splice_rows");
    return SLIME_OK;
}

// ------------------------------------------------------------------------------------------------ RoPE
// One workgroup per token row: cos / sin of the 64 angles once (LDS), then every (head, pair of adjacent i) item.
template <typename T>
__global__ void __launch_bounds__(256) rope_kernel(char* qkv, long row_stride, const int* pos, int n_rot, int n_q,
                                                   const float* inv_freq, float q_scale) {
    __shared__ float cs[2][64];
    const long row = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid < 64) {
        const float ang = (float)pos[row] * inv_freq[tid];
        cs[0][tid] = cosf(ang);
        cs[1][tid] = sinf(ang);
    }
    __syncthreads();
    char* base = qkv + (size_t)row * row_stride * 2;
    for (int it = tid; it < n_rot * 32; it += 256) {
        const int h = it >> 5, i = (it & 31) * 2;
        unsigned* lo = reinterpret_cast<unsigned*>(base + ((size_t)h * 128 + i) * 2);
        unsigned* hi = reinterpret_cast<unsigned*>(base + ((size_t)h * 128 + 64 + i) * 2);
        const unsigned a = *lo, b = *hi;
        const float x0 = T::lo(a), x1 = T::hi(a), y0 = T::lo(b), y1 = T::hi(b);
        const float sc = h < n_q ? q_scale : 1.0f;
        const float c0 = cs[0][i], s0 = cs[1][i], c1 = cs[0][i + 1], s1 = cs[1][i + 1];
        *lo = T::pack2((x0 * c0 - y0 * s0) * sc, (x1 * c1 - y1 * s1) * sc);
        *hi = T::pack2((y0 * c0 + x0 * s0) * sc, (y1 * c1 + x1 * s1) * sc);
    }
}

extern "C" int slime_rope(void* qkv, long row_stride, const int32_t* pos, long rows, int n_rot_heads, int n_q_heads, int head_dim,
                          const float* inv_freq, float q_scale, int dtype, void* stream) {
    SLIME_REQUIRE(qkv && pos && inv_freq && rows > 0, "rope: bad input");
    SLIME_REQUIRE(head_dim == 128, "rope: head_dim %d unsupported (128)", head_dim);
    SLIME_REQUIRE(n_rot_heads > 0 && n_q_heads >= 0 && n_q_heads <= n_rot_heads && row_stride >= (long)n_rot_heads * 128, "rope: bad head counts");
    SLIME_REQUIRE(((uintptr_t)qkv % 4) == 0 && row_stride % 2 == 0, "rope: buffer must be 4-byte aligned");
    SLIME_REQUIRE(rows <= 0x7fffffffL, "rope: too many rows");
    if (dtype == SLIME_F16) hipLaunchKernelGGL(rope_kernel<F16>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (char*)qkv, row_stride, pos, n_rot_heads, n_q_heads, inv_freq, q_scale);
    else if (dtype == SLIME_BF16) hipLaunchKernelGGL(rope_kernel<BF16>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (char*)qkv, row_stride, pos, n_rot_heads, n_q_heads, inv_freq, q_scale);
    else { slime_set_error("rope: dtype %d is not a 16-bit type", dtype); return SLIME_EINVAL; }
    SLIME_CHECK_LAUNCH("rope");
    return SLIME_OK;
}

// ------------------------------------------------------------------------------------------------ causal GQA attention
struct PrefillArgs {
    const char* q; long q_bs, q_rs;
    const char* k; long k_bs, k_rs;
    const char* v; long v_bs, v_rs;
    char* o; long o_bs, o_rs;
    const int* kv_start; const int* kv_len;     // per sequence token range, or NULL: [0, S)
    int S, group;
    int n_kv_heads, batch;                      // prefill32_kernel's item space (its grid is a number of workgroups, not of items)
#ifdef SLIME_DIAG
    unsigned long long* dbg;                    // 8 counters, or NULL (prefill32.inc)
#endif
};

__device__ __forceinline__ float quad_rows_allmax(float x) {          // max over lanes l, l+16, l+32, l+48
    const unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned w = __float_as_uint(y);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float quad_rows_allsum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned w = __float_as_uint(y);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ u32x2 lds_tr16(const char* p) {
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4i16_ptr;
    s16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)LDS_PTR(p));
    return __builtin_bit_cast(u32x2, r);
}

template <typename T>
__global__ void __launch_bounds__(512) prefill_attn_kernel(PrefillArgs a) {
    constexpr int DH = 128, RB = 256, CPR = 16, KS = 4, DT = 8, KC = 192, NW = 8, NSUB = 2, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Klds = smem;
    char* Vlds = smem + KC * RB;
    const int kvh = blockIdx.x, b = blockIdx.y;
    const int QB = 256 / a.group;                             // query rows per workgroup
    const int nqb = (a.S + QB - 1) / QB;
    const int qb = nqb - 1 - (int)blockIdx.z;                 // heaviest (latest) blocks are dispatched first
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int ks_ = a.kv_start ? a.kv_start[b] : 0;
    const int ke_ = a.kv_len ? min(a.S, ks_ + a.kv_len[b]) : a.S;   // token range [ks_, ke_)

    // sub-block s of this wave: item i = 2 wave + s -> (query head, 16-row block)
    int head[NSUB], row0[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        const int i = 2 * wave + s;
        head[s] = kvh * a.group + i % a.group;
        row0[s] = qb * QB + (i / a.group) * 16;
    }
    const int q_lo = qb * QB, q_hi = min(q_lo + QB, a.S);     // this workgroup's query rows [q_lo, q_hi)
    const int kv_hi = min(ke_, q_hi);                         // causal: no key beyond the last query row
    const int wave_q_last = max(row0[0], row0[1]) + 15;

    u32x4 qf[NSUB][KS];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        const char* qbase = a.q + ((size_t)b * a.q_bs + (size_t)head[s] * DH) * 2;
        const int qr = min(row0[s] + li, a.S - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[s][ks] = *reinterpret_cast<const u32x4*>(qbase + ((size_t)qr * a.q_rs + ks * 32 + g * 8) * 2);
    }
    f32x4 o[NSUB][DT];
    float m_run[NSUB], l_run[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        m_run[s] = -INFINITY; l_run[s] = 0.f;
#pragma unroll
        for (int d = 0; d < DT; ++d) o[s][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = li * RB + (((ks * 4 + g) ^ (lane & (CPR - 1))) << 4);
    int voff[DT];
    {
        const int vrow = 4 * g + (li >> 2);
        const int sw = (vrow & 1) | (((vrow >> 1) & 3) << 3);       // v_swizzle<128>: depends on (row mod 16) only
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int e = 8 * (dt >> 1) + 2 * (li & 3) + (dt & 1);
            voff[dt] = vrow * RB + ((e ^ sw) << 3);
        }
    }
    const char* kbase = a.k + ((size_t)b * a.k_bs + (size_t)kvh * DH) * 2;
    const char* vbase = a.v + ((size_t)b * a.v_bs + (size_t)kvh * DH) * 2;

    const int kv_first = (ks_ >> 5) << 5;                     // 32-aligned first step that holds a token
    // K/V chunk staging, split (issue early / write late): the global loads of chunk c+1 are issued BEFORE the arithmetic of
    // chunk c and held in registers (9 + 9 x 16 B per lane); they are written to LDS, swizzled, after it.  HBM/L2 latency
    // passes under the MFMAs instead of in front of them.
    constexpr int TOTAL = KC * CPR, U = TOTAL / NT;           // 16-byte pieces per chunk per operand, per lane
    static_assert(TOTAL % NT == 0, "chunk pieces must divide over the threads");
    u32x4 kk[U], vv[U];
    auto fetch = [&](int kv0) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int idx = j * NT + tid;
            const int r = idx / CPR, u = idx % CPR, gr = kv0 + r;
            kk[j] = u32x4{0u, 0u, 0u, 0u}; vv[j] = u32x4{0u, 0u, 0u, 0u};
            if (gr < kv_hi) {
                kk[j] = *reinterpret_cast<const u32x4*>(kbase + ((size_t)gr * a.k_rs) * 2 + u * 16);
                vv[j] = *reinterpret_cast<const u32x4*>(vbase + ((size_t)gr * a.v_rs) * 2 + u * 16);
            }
        }
    };
    auto commit = [&]() {                                     // registers -> swizzled LDS images
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int idx = j * NT + tid;
            const int r = idx / CPR, u = idx % CPR;
            *reinterpret_cast<u32x4*>(Klds + r * RB + ((u ^ (r & (CPR - 1))) << 4)) = kk[j];
            const int sw = (r & 1) | (((r >> 1) & 3) << 3);
            u32x4 w = vv[j];
            if (sw & 1) w = u32x4{vv[j][2], vv[j][3], vv[j][0], vv[j][1]};
            *reinterpret_cast<u32x4*>(Vlds + r * RB + ((u ^ (sw >> 1)) << 4)) = w;
        }
    };
    if (kv_first < kv_hi) fetch(kv_first);
    for (int kv0 = kv_first; kv0 < kv_hi; kv0 += KC) {
        if (kv0 > kv_first) __syncthreads();                  // previous chunk fully consumed
        commit();
        __syncthreads();
        if (kv0 + KC < kv_hi) fetch(kv0 + KC);                // in flight during this chunk's arithmetic
        const int rows = min(KC, kv_hi - kv0);
        const int steps = (rows + 31) >> 5;
        for (int st = 0; st < steps; ++st) {
            const int kvs = kv0 + st * 32;                    // first key of this step
            if (kvs > wave_q_last) break;                     // causal: nothing left for this wave's queries (wave-uniform)
            const char* kp = Klds + st * 32 * RB;
            const char* vp = Vlds + st * 32 * RB;
            f32x4 sc[NSUB][2];
#pragma unroll
            for (int s = 0; s < NSUB; ++s) { sc[s][0] = f32x4{0.f, 0.f, 0.f, 0.f}; sc[s][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 kf = *reinterpret_cast<const u32x4*>(kp + t * 16 * RB + koff[ks]);
#pragma unroll
                    for (int s = 0; s < NSUB; ++s) sc[s][t] = T::mfma16(kf, qf[s][ks], sc[s][t]);
                }
            // masks: causal (key > query), token range.  Steps strictly inside [ks_, min(row0) ] need none (wave-uniform test).
            if (kvs + 31 > min(row0[0], row0[1]) || kvs < ks_ || kvs + 32 > kv_hi) {
#pragma unroll
                for (int s = 0; s < NSUB; ++s) {
                    const int qrow = row0[s] + li;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int kvi = kvs + t * 16 + 4 * g + r;
                            const bool dead = kvi > qrow || kvi < ks_ || kvi >= kv_hi;
                            sc[s][t][r] = dead ? -INFINITY : sc[s][t][r];
                        }
                }
            }
            u32x4 vf[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const u32x2 v0 = lds_tr16(vp + voff[dt]);
                const u32x2 v1 = lds_tr16(vp + 16 * RB + voff[dt]);
                vf[dt] = u32x4{v0[0], v0[1], v1[0], v1[1]};
            }
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                float mx = fmaxf(fmaxf(fmaxf(sc[s][0][0], sc[s][0][1]), fmaxf(sc[s][0][2], sc[s][0][3])),
                                 fmaxf(fmaxf(sc[s][1][0], sc[s][1][1]), fmaxf(sc[s][1][2], sc[s][1][3])));
                mx = quad_rows_allmax(mx);
                // floor keeps rows whose keys are ALL masked so far (padding rows) free of inf - inf
                const float m_new = fmaxf(fmaxf(m_run[s], mx), -1e30f);
                if (__builtin_amdgcn_ballot_w64(m_new != m_run[s]) != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(m_run[s] - m_new);
                    l_run[s] *= alpha;
#pragma unroll
                    for (int d = 0; d < DT; ++d) o[s][d] *= alpha;
                    m_run[s] = m_new;
                }
                const float mneg = -m_run[s];
                float p[8];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[t * 4 + r] = __builtin_amdgcn_exp2f(sc[s][t][r] + mneg);
                l_run[s] += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
                const u32x4 pf = pack8<T>(p);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[s][dt] = T::mfma16(vf[dt], pf, o[s][dt]);
            }
        }
    }

#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        const float l = quad_rows_allsum(l_run[s]);
        const int qr = row0[s] + li;
        const bool live = qr >= ks_ && qr < ke_;             // pad_input: rows outside the token range are zero
        const float inv = (live && l > 0.f) ? 1.0f / l : 0.f;
        if (qr < a.S) {
            char* obase = a.o + ((size_t)b * a.o_bs + (size_t)head[s] * DH) * 2;
#pragma unroll
            for (int p = 0; p < DT / 2; ++p) {
                float v[8] = {o[s][2 * p][0] * inv, o[s][2 * p][1] * inv, o[s][2 * p][2] * inv, o[s][2 * p][3] * inv,
                              o[s][2 * p + 1][0] * inv, o[s][2 * p + 1][1] * inv, o[s][2 * p + 1][2] * inv, o[s][2 * p + 1][3] * inv};
                if (!live) { for (int j = 0; j < 8; ++j) v[j] = 0.f; }      // 0 * garbage could be NaN
                *reinterpret_cast<u32x4*>(obase + ((size_t)qr * a.o_rs + 32 * p + 8 * g) * 2) = pack8<T>(v);
            }
        }
    }
}

#include "prefill32.inc"

#ifdef SLIME_DIAG
static int g_prefill_variant = 0;      // 1 = force the eight-wave kernel, 2 = prefill32 with one item per workgroup
extern "C" void slime_prefill_set_variant(int v) { g_prefill_variant = v; }
static unsigned long long* g_prefill_dbg = nullptr;
extern "C" void slime_prefill_set_debug(void* counters) { g_prefill_dbg = (unsigned long long*)counters; }
#else
static constexpr int g_prefill_variant = 0;
#endif

extern "C" int slime_prefill_attention(const void* q, long q_bs, long q_rs, const void* k, long k_bs, long k_rs, const void* v,
                                       long v_bs, long v_rs, void* o, long o_bs, long o_rs, int batch, int n_heads,
                                       int n_kv_heads, int head_dim, int S, const int32_t* kv_start, const int32_t* kv_len,
                                       int dtype, void* stream) {
    SLIME_REQUIRE(q && k && v && o, "prefill_attention: null pointer");
    SLIME_REQUIRE(batch > 0 && batch <= 65535 && S > 0, "prefill_attention: empty shape");
    SLIME_REQUIRE(head_dim == 128, "prefill_attention: head_dim %d unsupported (128)", head_dim);
    SLIME_REQUIRE(n_kv_heads > 0 && n_heads % n_kv_heads == 0, "prefill_attention: %d query heads / %d kv heads", n_heads, n_kv_heads);
    const int group = n_heads / n_kv_heads;
    SLIME_REQUIRE(group == 1 || group == 2 || group == 4 || group == 8 || group == 16, "prefill_attention: group size %d unsupported", group);
    SLIME_REQUIRE(q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && o_rs % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 && v_bs % 8 == 0 &&
                  o_bs % 8 == 0, "prefill_attention: strides must be multiples of 8 elements");
    SLIME_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16 == 0, "prefill_attention: pointers must be 16-byte aligned");
    SLIME_REQUIRE((kv_start == nullptr) == (kv_len == nullptr), "prefill_attention: kv_start and kv_len come together");
    PrefillArgs a{(const char*)q, q_bs, q_rs, (const char*)k, k_bs, k_rs, (const char*)v, v_bs, v_rs, (char*)o, o_bs, o_rs,
                  kv_start, kv_len, S, group, n_kv_heads, batch};
#ifdef SLIME_DIAG
    a.dbg = g_prefill_dbg;
#endif
    hipStream_t s = (hipStream_t)stream;
    if (group == 4 && dtype == SLIME_BF16 && g_prefill_variant != 1) {
        // Llama-3 geometry: one wave per SIMD, 32x32x16 MFMAs, K/V ring by LDS-DMA (prefill32.inc)
        // (persistent over its work items: one workgroup per CU -- the 128 KiB ring allows no second one -- walks the items in
        // snake order; diagnostic variant 2: one item per workgroup, the round-2 launch)
        constexpr int LDS32 = 2 * 8 * 32 * 256;
        const long items = (long)n_kv_heads * batch * ((S + 63) / 64);
        SLIME_REQUIRE(items < (1L << 30), "prefill_attention: sequence too long");
        const int cus = num_cus() >= 8 ? (num_cus() & ~7) : 8;      // (a multiple of 8: the snake's mirror stays inside an XCD class)
        const int grid = (g_prefill_variant == 2 || items <= cus) ? (int)items : cus;
#ifdef SLIME_DIAG
        if (g_prefill_variant == 3) {                            // look-ahead of 4 key steps instead of 6
            auto kern4 = prefill32_kernel<BF16, 4>;
            SLIME_SET_LDS_ONCE(kern4, LDS32, "prefill_attention");
            hipLaunchKernelGGL(kern4, dim3(grid), dim3(256), LDS32, s, a);
            SLIME_CHECK_LAUNCH("prefill_attention");
            return SLIME_OK;
        }
#endif
        auto kern = prefill32_kernel<BF16>;
        SLIME_SET_LDS_ONCE(kern, LDS32, "prefill_attention");
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS32, s, a);
        SLIME_CHECK_LAUNCH("prefill_attention");
        return SLIME_OK;
    }
    constexpr int LDS = 2 * 192 * 256;
    const int QB = 256 / group, nqb = (S + QB - 1) / QB;
    SLIME_REQUIRE(nqb <= 65535, "prefill_attention: sequence too long");
    if (dtype == SLIME_F16) {
        auto kern = prefill_attn_kernel<F16>;
        SLIME_SET_LDS_ONCE(kern, LDS, "prefill_attention");
        hipLaunchKernelGGL(kern, dim3(n_kv_heads, batch, nqb), dim3(512), LDS, s, a);
    } else if (dtype == SLIME_BF16) {
        auto kern = prefill_attn_kernel<BF16>;
        SLIME_SET_LDS_ONCE(kern, LDS, "prefill_attention");
        hipLaunchKernelGGL(kern, dim3(n_kv_heads, batch, nqb), dim3(512), LDS, s, a);
    } else {
        slime_set_error("prefill_attention: dtype %d is not a 16-bit MFMA type", dtype);
        return SLIME_EINVAL;
    }
    SLIME_CHECK_LAUNCH("prefill_attention");
    return SLIME_OK;
}

// ------------------------------------------------------------------------------------------------ LlamaAttention.forward
static int llama_validate(const slime_llama_attn_desc* d) {
    SLIME_REQUIRE(d, "llama_attn: null descriptor");
    SLIME_REQUIRE(d->dtype == SLIME_BF16 || d->dtype == SLIME_F16, "llama_attn: dtype must be BF16 or F16");
    SLIME_REQUIRE(d->head_dim == 128 && d->n_heads > 0 && d->n_kv_heads > 0 && d->n_heads % d->n_kv_heads == 0, "llama_attn: heads");
    SLIME_REQUIRE(d->hidden % 128 == 0 && d->hidden % 64 == 0, "llama_attn: hidden=%d must be a multiple of 128", d->hidden);
    SLIME_REQUIRE((d->w_qkv || d->w_qkv_frag) && (d->w_o || d->w_o_frag) && d->inv_freq, "llama_attn: missing weights (row-major or fragment-order)");
    return SLIME_OK;
}

extern "C" size_t slime_llama_attn_workspace_bytes(const slime_llama_attn_desc* d, int batch, int S) {
    if (!d || batch <= 0 || S <= 0) return 0;
    const size_t rows = (size_t)batch * S;
    const size_t qkv = align_up(rows * (size_t)(d->n_heads + 2 * d->n_kv_heads) * d->head_dim * 2, 256);
    const size_t ctx = align_up(rows * (size_t)d->n_heads * d->head_dim * 2, 256);
    return qkv + ctx;
}

// One implementation behind both entry points: resid == NULL -> out = o_proj(ctx) (T or fp32); else out = T(resid + o_proj(ctx)).
static int llama_attn_run(const slime_llama_attn_desc* d, const void* hidden, const int32_t* position_ids, const int32_t* kv_start,
                          const int32_t* kv_len, int batch, int S, void* out, int out_dtype, const void* resid,
                          void* ws, size_t ws_bytes, void* stream) {
    int rc = llama_validate(d);
    if (rc != SLIME_OK) return rc;
    SLIME_REQUIRE(hidden && position_ids && out && batch > 0 && S > 0, "llama_attn: bad input");
    const size_t need = slime_llama_attn_workspace_bytes(d, batch, S);
    if (!ws || ws_bytes < need || ((uintptr_t)ws % 256) != 0) {
        slime_set_error("llama_attn: workspace %zu B (need %zu, 256-B aligned)", ws_bytes, need);
        return SLIME_EWORKSPACE;
    }
    const int HQ = d->n_heads, HKV = d->n_kv_heads, DH = d->head_dim, D = d->hidden;
    const int NQKV = (HQ + 2 * HKV) * DH, M = batch * S;
    char* qkv = (char*)ws;
    char* ctx = qkv + align_up((size_t)M * NQKV * 2, 256);
    slime_gemm_args ga{};
    // q/k/v projections as one GEMM (monkey patch :31-45; Llama has no projection biases)
    ga.A = hidden; ga.lda = D; ga.B = d->w_qkv; ga.B_frag = d->w_qkv_frag; ga.C = qkv; ga.ldc = NQKV; ga.M = M; ga.N = NQKV; ga.K = D;
    ga.dtype = d->dtype; ga.epilogue = SLIME_EPI_BIAS_T;
    rc = slime_gemm_ex(&ga, stream);
    if (rc != SLIME_OK) return rc;
    // RoPE on q and k (:51-54); q also takes head_dim^-0.5 * log2(e)
    rc = slime_rope(qkv, NQKV, position_ids, M, HQ + HKV, HQ, DH, d->inv_freq, 0.08838834764831845f * 1.4426950408889634f, d->dtype, stream);
    if (rc != SLIME_OK) return rc;
    rc = slime_prefill_attention(qkv, (long)S * NQKV, NQKV, qkv + (size_t)HQ * DH * 2, (long)S * NQKV, NQKV,
                                 qkv + (size_t)(HQ + HKV) * DH * 2, (long)S * NQKV, NQKV, ctx, (long)S * HQ * DH, HQ * DH, batch,
                                 HQ, HKV, DH, S, kv_start, kv_len, d->dtype, stream);
    if (rc != SLIME_OK) return rc;
    // o_proj (:92), optionally with the decoder layer's residual add in its epilogue
    ga = slime_gemm_args{};
    ga.A = ctx; ga.lda = HQ * DH; ga.B = d->w_o; ga.B_frag = d->w_o_frag; ga.M = M; ga.N = D; ga.K = HQ * DH; ga.dtype = d->dtype;
    ga.C = out; ga.ldc = D;
    if (resid) { ga.epilogue = SLIME_EPI_BIAS_RESID_T; ga.resid = resid; ga.ldr = D; }
    else ga.epilogue = out_dtype == SLIME_F32 ? SLIME_EPI_BIAS_F32 : SLIME_EPI_BIAS_T;
    return slime_gemm_ex(&ga, stream);
}

extern "C" int slime_llama_attn_forward(const slime_llama_attn_desc* d, const void* hidden, const int32_t* position_ids,
                                        const int32_t* kv_start, const int32_t* kv_len, int batch, int S, void* out, int out_dtype,
                                        void* ws, size_t ws_bytes, void* stream) {
    SLIME_REQUIRE(d && out, "llama_attn: bad input");
    SLIME_REQUIRE(out_dtype == d->dtype || out_dtype == SLIME_F32, "llama_attn: out dtype must be the operand type or F32");
    return llama_attn_run(d, hidden, position_ids, kv_start, kv_len, batch, S, out, out_dtype, nullptr, ws, ws_bytes, stream);
}

extern "C" int slime_llama_attn_forward_resid(const slime_llama_attn_desc* d, const void* hidden, const int32_t* position_ids,
                                              const int32_t* kv_start, const int32_t* kv_len, int batch, int S, const void* resid,
                                              void* out, void* ws, size_t ws_bytes, void* stream) {
    SLIME_REQUIRE(d && resid && out, "llama_attn_resid: resid and out are required");
    SLIME_REQUIRE(hidden != out, "llama_attn_resid: out may alias resid, not the layer's input rows");
    return llama_attn_run(d, hidden, position_ids, kv_start, kv_len, batch, S, out, d->dtype, resid, ws, ws_bytes, stream);
}
