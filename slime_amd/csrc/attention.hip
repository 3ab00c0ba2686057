// attention.hip -- fused softmax(Q K^T) V for gfx950, head_dim 64 (CLIP-ViT self-attention, S = 577)
// and 128 (Resampler cross-attention, 144/576 queries x 576 keys).
//
// One workgroup owns one (batch item, head) [x q-split]; its K and V panels live in LDS
// (the whole 577 x 64 K and V of a CLIP head are 2 x 76 KiB -- they fit the 160 KiB LDS, so they are
// read from HBM/L2 exactly once per workgroup and there is no staging pipeline in the main loop;
// head_dim 128 uses kv chunks of 288 rows).  Queries are split in 16-row sub-blocks that are dealt to
// the waves (577 -> 37 sub-blocks -> 5,5,5,5,5,4,4,4); a wave keeps all its sub-blocks in flight so
// every K/V fragment read from LDS feeds NSUB MFMAs.
//
// Everything is computed TRANSPOSED (the "swapped QK^T" form):
//   S^T[kv,q] = K Q^T      mfma(A = K rows from LDS (ds_read_b128), B = Q rows held in VGPRs)
//   O^T[d,q]  = V^T P^T    mfma(A = V^T via ds_read_b64_tr_b16 (hardware transpose), B = P^T)
// With the 16x16x32 C/D layout (col = lane&15, row = 4*(lane>>4)+r) a lane owns ONE query column:
// running max / sum / rescale factors are lane-local scalars, and the fp32 S^T tile a lane holds is
// -- after exp and a 16-bit pack -- bit-for-bit the B operand of the PV MFMA (k-slot order
// {tile0: 4g+0..3, tile1: 4g+0..3}, matched by the two transpose reads of V).  P never touches LDS.
//
// LDS images: K rows are 16-B-chunk XOR-swizzled for conflict-free ds_read_b128; V rows are swizzled
// at 8-B granularity so that the 8 rows x 4 pieces a half-wave transpose-read touches cover all 64
// banks.  Softmax statistics are fp32; scores arrive in log2 units (q carries dh^-0.5 * log2 e), exp via v_exp_f32.
#include "common.h"
#include <type_traits>

struct AttnArgs {
    const char* q; long q_bs, q_rs;
    const char* k; long k_bs, k_rs;
    const char* v; long v_bs, v_rs;
    char* o; long o_bs, o_rs;
    int heads, n_q, n_kv, sb_per_wg;
    unsigned long long* dbg;      // diagnostic: 4 s_memtime stamps per wave (NULL in production)
    int abl;                      // diagnostic timing ablations of attn64 (0 in production): 1 = no softmax VALU, 2 = no MFMA
    int n_full = 0, split = 1, n_cut = 0;   // attn32: items (crop, head) < n_full are one workgroup each, the n_cut others `split` each
    int n_items = 0;              // attn64g: heads x batch (its grid is one-dimensional)
};

template <int DH> __device__ __forceinline__ int v_swizzle(int row) {
    if constexpr (DH == 64) return ((row >> 1) & 1) | (((row >> 2) & 1) << 3);
    else return (row & 1) | (((row >> 1) & 3) << 3);
}

// max over the four 16-lane rows of a wave (lanes l, l+16, l+32, l+48), result in every row.
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of the second,
// v_permlane32_swap the upper half of the first with the lower half of the second: with both operands
// equal to x the two results hold the two partners of every lane.  Pure VALU -- a __shfl_xor would be a
// ds_bpermute plus an LDS round trip (10 dependent ones per kv step here).
__device__ __forceinline__ float rows_allmax(float x) {
    const unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned w = __float_as_uint(y);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_allsum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned w = __float_as_uint(y);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__device__ __forceinline__ u32x2 lds_read_tr16(const char* p) {
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4i16_ptr;
    s16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)LDS_PTR(p));
    return __builtin_bit_cast(u32x2, r);
}

template <typename T, int DH, int KC, int NW, int NSUB>
__device__ __forceinline__ void attn_body(const AttnArgs& a, char* smem, const int b, const int h, const int sb0) {
    constexpr int RB = DH * 2;            // bytes per K/V row
    constexpr int CPR = DH / 8;           // 16-B chunks per row
    constexpr int KS = DH / 32;           // k-steps of the QK^T contraction
    constexpr int DT = DH / 16;           // 16-row tiles of O^T
    constexpr int NT = NW * 64;
    char* Klds = smem;
    char* Vlds = smem + KC * RB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = lane >> 4, li = lane & 15;
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (a.dbg) t0 = __builtin_amdgcn_s_memtime();

    const char* kbase = a.k + ((size_t)b * a.k_bs + (size_t)h * DH) * 2;
    const char* vbase = a.v + ((size_t)b * a.v_bs + (size_t)h * DH) * 2;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds q-row li, d = ks*32 + 8g .. +7 -------
    u32x4 qf[NSUB > 0 ? NSUB : 1][KS];
    if constexpr (NSUB > 0) {
        const char* qbase = a.q + ((size_t)b * a.q_bs + (size_t)h * DH) * 2;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const int qr = min((sb0 + s) * 16 + li, a.n_q - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qf[s][ks] = *reinterpret_cast<const u32x4*>(qbase + ((size_t)qr * a.q_rs + ks * 32 + g * 8) * 2);
        }
    }
    f32x4 o[NSUB > 0 ? NSUB : 1][DT];
    float m_run[NSUB > 0 ? NSUB : 1], l_run[NSUB > 0 ? NSUB : 1];
#pragma unroll
    for (int s = 0; s < (NSUB > 0 ? NSUB : 1); ++s) {
        m_run[s] = -INFINITY; l_run[s] = 0.f;
#pragma unroll
        for (int d = 0; d < DT; ++d) o[s][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // per-lane LDS offsets
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = li * RB + (((ks * 4 + g) ^ (lane & (CPR - 1))) << 4);
    int voff[DT];
    {
        const int vrow = 4 * g + (li >> 2);
        const int sw = v_swizzle<DH>(vrow);                 // depends on (row mod 16) only
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int e = 8 * (dt >> 1) + 2 * (li & 3) + (dt & 1);
            voff[dt] = vrow * RB + ((e ^ sw) << 3);
        }
    }

    for (int kv0 = 0; kv0 < a.n_kv; kv0 += KC) {
        if (kv0 > 0) __syncthreads();                       // previous chunk fully consumed
        // ---- stage K and V chunk: global (16 B/lane) -> registers -> swizzled LDS ---------------
        {
            constexpr int TOTAL = KC * CPR;
            constexpr int U = 5;
            for (int base = 0; base < TOTAL; base += NT * U) {
                u32x4 kk[U], vv[U];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int idx = base + j * NT + tid;
                    const int r = idx / CPR, u = idx % CPR;
                    const int gr = kv0 + r;
                    kk[j] = u32x4{0u, 0u, 0u, 0u}; vv[j] = u32x4{0u, 0u, 0u, 0u};
                    if (idx < TOTAL && gr < a.n_kv) {
                        kk[j] = *reinterpret_cast<const u32x4*>(kbase + ((size_t)gr * a.k_rs) * 2 + u * 16);
                        vv[j] = *reinterpret_cast<const u32x4*>(vbase + ((size_t)gr * a.v_rs) * 2 + u * 16);
                    }
                }
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int idx = base + j * NT + tid;
                    if (idx < TOTAL) {
                        const int r = idx / CPR, u = idx % CPR;
                        *reinterpret_cast<u32x4*>(Klds + r * RB + ((u ^ (r & (CPR - 1))) << 4)) = kk[j];
                        const int sw = v_swizzle<DH>(r);
                        u32x4 w = vv[j];
                        if (sw & 1) w = u32x4{vv[j][2], vv[j][3], vv[j][0], vv[j][1]};
                        *reinterpret_cast<u32x4*>(Vlds + r * RB + ((u ^ (sw >> 1)) << 4)) = w;
                    }
                }
            }
        }
        __syncthreads();
        if (a.dbg && kv0 == 0) t1 = __builtin_amdgcn_s_memtime();

        if constexpr (NSUB > 0) {
            const int rows = min(KC, a.n_kv - kv0);
            const int steps = (rows + 31) >> 5;
            for (int st = 0; st < steps; ++st) {
                const char* kp = Klds + st * 32 * RB;
                const char* vp = Vlds + st * 32 * RB;
                // ---- S^T tiles: 2 x (16 kv) per sub-block ------------------------------------------
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
                if (kv0 + st * 32 + 32 > a.n_kv) {          // ragged tail: mask kv >= n_kv
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool dead = kv0 + st * 32 + t * 16 + 4 * g + r >= a.n_kv;
#pragma unroll
                            for (int s = 0; s < NSUB; ++s) if (dead) sc[s][t][r] = -INFINITY;
                        }
                }
                // ---- V^T fragments of this step (hardware transpose reads), shared by all sub-blocks ------
                u32x4 vf[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const u32x2 v0 = lds_read_tr16(vp + voff[dt]);
                    const u32x2 v1 = lds_read_tr16(vp + 16 * RB + voff[dt]);
                    vf[dt] = u32x4{v0[0], v0[1], v1[0], v1[1]};
                }
                // ---- per sub-block: online softmax (lane-local: one q column per lane), then O^T += V^T P^T.
                // Sub-block-major order lets the PV MFMAs of sub-block s run under the softmax VALU work of
                // sub-block s+1 (separate pipes).
#pragma unroll
                for (int s = 0; s < NSUB; ++s) {
                    float mx = fmaxf(fmaxf(fmaxf(sc[s][0][0], sc[s][0][1]), fmaxf(sc[s][0][2], sc[s][0][3])),
                                     fmaxf(fmaxf(sc[s][1][0], sc[s][1][1]), fmaxf(sc[s][1][2], sc[s][1][3])));
                    mx = rows_allmax(mx);
                    const float m_new = fmaxf(m_run[s], mx);
                    if (__builtin_amdgcn_ballot_w64(m_new != m_run[s]) != 0) {     // wave-uniform: some row's max moved
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
    }

    if (a.dbg) t2 = __builtin_amdgcn_s_memtime();
    // ---- finalize: 1/l, 16-B stores of 8 consecutive d per lane ---------------------------------
    if constexpr (NSUB > 0) {
        char* obase = a.o + ((size_t)b * a.o_bs + (size_t)h * DH) * 2;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const float inv = 1.0f / rows_allsum(l_run[s]);
            const int qr = (sb0 + s) * 16 + li;
            if (qr < a.n_q) {
#pragma unroll
                for (int p = 0; p < DT / 2; ++p) {
                    float v[8] = {o[s][2 * p][0] * inv, o[s][2 * p][1] * inv, o[s][2 * p][2] * inv, o[s][2 * p][3] * inv,
                                  o[s][2 * p + 1][0] * inv, o[s][2 * p + 1][1] * inv, o[s][2 * p + 1][2] * inv, o[s][2 * p + 1][3] * inv};
                    st_stream(reinterpret_cast<u32x4*>(obase + ((size_t)qr * a.o_rs + 32 * p + 8 * g) * 2), pack8<T>(v));
                }
            }
        }
    }
    if (a.dbg && lane == 0) {
        const int wv = tid >> 6;
        unsigned long long* d = a.dbg + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (NW * 4) + wv * 4;
        d[0] = t0; d[1] = t1; d[2] = t2; d[3] = __builtin_amdgcn_s_memtime();
    }
}

template <typename T, int DH, int KC, int NW, int NSUBMAX>
__global__ void __launch_bounds__(NW * 64) attn_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int h = blockIdx.x, b = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total_sb = (a.n_q + 15) >> 4;
    const int wg_sb0 = blockIdx.z * a.sb_per_wg;
    const int nsb = min(a.sb_per_wg, total_sb - wg_sb0);
    const int base = nsb / NW, rem = nsb % NW;
    const int cnt = base + (wave < rem ? 1 : 0);
    const int sb0 = wg_sb0 + wave * base + min(wave, rem);
    // The second-dispatched half of a workgroup loses VALU arbitration to the older half on every SIMD
    // (measured: 70-81k vs 52k compute cycles per wave); one static priority bump evens them out.
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
    switch (cnt) {
        case 0: attn_body<T, DH, KC, NW, 0>(a, smem, b, h, sb0); break;
        case 1: attn_body<T, DH, KC, NW, 1>(a, smem, b, h, sb0); break;
        case 2: attn_body<T, DH, KC, NW, 2>(a, smem, b, h, sb0); break;
        case 3: attn_body<T, DH, KC, NW, 3>(a, smem, b, h, sb0); break;
        default:
            if constexpr (NSUBMAX >= 5) {
                if (cnt == 4) attn_body<T, DH, KC, NW, 4>(a, smem, b, h, sb0);
                else attn_body<T, DH, KC, NW, 5>(a, smem, b, h, sb0);
            }
            break;
    }
}

template <typename T, int DH, int KC, int NW, int NSUBMAX>
static int launch_attn(const AttnArgs& a0, int batch, hipStream_t stream) {
    AttnArgs a = a0;
    constexpr int LDS = 2 * KC * DH * 2;
    auto kern = attn_kernel<T, DH, KC, NW, NSUBMAX>;
    SLIME_SET_LDS_ONCE(kern, LDS, "attention");
    const int total_sb = (a.n_q + 15) / 16;
    const int cap = NW * NSUBMAX;
    const int qsplit = (total_sb + cap - 1) / cap;
    a.sb_per_wg = (total_sb + qsplit - 1) / qsplit;
    hipLaunchKernelGGL(kern, dim3(a.heads, batch, qsplit), dim3(NW * 64), LDS, stream, a);
    SLIME_CHECK_LAUNCH("attention");
    return SLIME_OK;
}


// ================================================================================================
// attn64_kernel: the CLIP self-attention shape (head_dim 64, n_kv <= 608 so K and V of one head are
// LDS resident), software pipelined.
//
// Stamps of the generic kernel above showed 17k of ~95k cycles per workgroup in register-staged K/V
// loading and ~390 cycles per (16-query sub-block x 32-kv step) against 128 of MFMA: within a wave the
// stream was [QK MFMAs][softmax VALU][PV MFMAs], and issue is in order, so nothing overlapped.  Here:
//   * K/V come in by LDS-DMA (no VGPR round trip, no ds_write pass), in two halves: kv rows 0..319 are
//     waited for, rows 320..607 land under the first ten steps.  K is 16-B-chunk XOR-swizzled through
//     the per-lane source address (conflict-free ds_read_b128); V can only be chunk-swizzled by a DMA
//     (key = 64-B half flip on (row>>1)&1), which leaves the transpose reads 2-way conflicting -- 8 reads
//     per 24 MFMAs, irrelevant;  rows >= n_kv are clamped copies of the last row (masked later).
//   * score tiles are double buffered: QK(step+1) MFMAs are issued together with softmax(step) VALU work
//     and PV(step) MFMAs, so the matrix pipe runs under the exp/max/sum stream of the same wave.
//   * 3 sub-blocks per wave and a 2-way split of the 37 query sub-blocks: 2 workgroups per (crop, head),
//     640 workgroups at 20 crops instead of 320 (1.25 rounds of 256 CUs).
// ================================================================================================
// Fragment reads of attn64 go through inline asm: with LDS-DMA in flight hipcc guards every compiler-
// visible ds_read that may alias a DMA destination with s_waitcnt vmcnt(0), which would drain the second
// K/V half before the first MFMA.  The DMA/barrier ordering is explicit in attn64_body; consumers are
// ordered behind these reads by lgkm_wait_*(), which name the destination registers as "+v".
// OFF: compile-time byte offset folded into the instruction's 16-bit offset field (inline asm hides the address
// arithmetic from the compiler, which would otherwise spend one v_add_u32 per read on it).
template <int OFF>
__device__ __forceinline__ u32x4 lds_b128_asm(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF>
__device__ __forceinline__ u32x2 lds_tr16_asm(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF>
__device__ __forceinline__ u32x2 lds_b64_asm(unsigned addr) {       // plain 8-byte read (timing ablation VABL: see attn64r_pass)
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    u32x2 r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
__device__ __forceinline__ unsigned lds_addr(const char* p) {
    return (unsigned)(size_t)((__attribute__((address_space(3))) const char*)p);
}

template <typename T, int NSUB>
__device__ __forceinline__ void attn64_body(const AttnArgs& a, char* smem, const int b, const int h, const int sb0) {
    constexpr int DH = 64, RB = 128, KS = 2, DT = 4, KC = 608, NW = 8;
    constexpr int ROWS1 = 320;                             // first DMA half (10 steps)
    constexpr float RESCALE_TH = 8.0f;                     // log2 units: p = exp2(s - m_ref) <= 256 between refreshes
    char* Klds = smem;
    char* Vlds = smem + KC * RB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;

    // ---- Q fragments first (ordinary loads must be retired before any LDS-DMA is in flight) --------
    u32x4 qf[NSUB > 0 ? NSUB : 1][KS];
    if constexpr (NSUB > 0) {
        const char* qbase = a.q + ((size_t)b * a.q_bs + (size_t)h * DH) * 2;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const int qr = min((sb0 + s) * 16 + li, a.n_q - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                qf[s][ks] = *reinterpret_cast<const u32x4*>(qbase + ((size_t)qr * a.q_rs + ks * 32 + g * 8) * 2);
        }
        // make the compiler retire the Q loads HERE (an empty asm that "uses" them): otherwise it waits
        // vmcnt(0) at their first real use, i.e. with the K/V DMA in flight, draining both halves.
#pragma unroll
        for (int s = 0; s < NSUB; ++s)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[s][ks]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- K/V LDS-DMA: one wave instruction = 8 rows x 128 B; pieces dealt round-robin to the 8 waves ----
    {
        const char* kbase = a.k + ((size_t)b * a.k_bs + (size_t)h * DH) * 2;
        const char* vbase = a.v + ((size_t)b * a.v_bs + (size_t)h * DH) * 2;
        const int lrow = lane >> 3, cpos = lane & 7;
        const int kchunk = cpos ^ lrow;                      // K: chunk ^ (row & 7)
        const int vchunk = cpos ^ (((lrow >> 1) & 1) << 2);  // V: 64-B half flip on (row >> 1) & 1
        auto dma_rows = [&](int piece) {                     // piece = 8-row group index
            const int row = min(piece * 8 + lrow, a.n_kv - 1);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(kbase + ((size_t)row * a.k_rs) * 2 + kchunk * 16),
                                             LDS_PTR(Klds + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(vbase + ((size_t)row * a.v_rs) * 2 + vchunk * 16),
                                             LDS_PTR(Vlds + piece * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int i = 0; i < ROWS1 / 8 / NW; ++i) dma_rows(i * NW + wave);                  // 5 x 2 per wave
#pragma unroll
        for (int i = 0; i < (KC - ROWS1) / 8 / NW; ++i) dma_rows(ROWS1 / 8 + i * NW + wave);   // 4 x 2 (+ tail below)
        if (wave < ((KC - ROWS1) / 8) % NW) dma_rows(ROWS1 / 8 + ((KC - ROWS1) / 8 / NW) * NW + wave);
    }
    // second-half pieces still in flight per wave: 2 * (4 or 5)
    if (wave < ((KC - ROWS1) / 8) % NW) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    const int steps = (a.n_kv + 31) >> 5;
    if constexpr (NSUB == 0) {
        if (steps > ROWS1 / 32) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
        return;
    } else {
        // o[s][0..3]: O^T d-tiles; o[s][4]: the row-sum tile (A operand = a constant fragment whose row 0 is
        // all ones, so the matrix pipe accumulates sum_kv P[q][kv] in row 0 -- the VALU is the bound here;
        // measured: lane-local fp32 sums instead are 3 % slower).
        // Scores are in log2 units (q carries dh^-0.5 * log2 e) and leave the matrix pipe ALREADY relative to the
        // reference max: the QK accumulator starts from cinit = -m_run instead of 0, so p = exp2(score) needs no
        // per-score subtract/scale on the VALU (which cannot overlap with the MFMA stream on this chip: a plain VALU
        // instruction beyond ~1 per MFMA costs ~6 cycles of matrix time, tools/mfma_valu_mix.hip).
        f32x4 o[NSUB][DT + 1], cinit[NSUB];
        float m_run[NSUB];
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            m_run[s] = 0.f;                                  // provisional; the first step always refreshes it
            cinit[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int d = 0; d <= DT; ++d) o[s][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const unsigned one2 = T::pack2(1.0f, 1.0f);
        const unsigned onew = (li == 0) ? one2 : 0u;
        const u32x4 ones = {onew, onew, onew, onew};
        // Running per-lane LDS pointers (advanced by one 32-row kv step per call, 4 v_add per step); everything else
        // is an immediate: K fragment (t, ks) at kptr[ks] + t*16*RB, V^T fragment (t, dt) at vptr[dt >> 1] + t*16*RB
        // + (dt & 1)*8.
        unsigned kptr[KS], vptr[2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kptr[ks] = lds_addr(Klds) + li * RB + (((ks * 4 + g) ^ (lane & 7)) << 4);
        {
            const int vrow = 4 * g + (li >> 2);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                vptr[hh] = lds_addr(Vlds) + vrow * RB + (((4 * hh + (li & 3)) ^ (((li >> 3) & 1) << 2)) << 4);
        }
        auto qk = [&](int st, f32x4 (&sc)[NSUB][2]) {
            u32x4 kf[2][KS];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kf[t][ks] = t == 0 ? lds_b128_asm<0>(kptr[ks]) : lds_b128_asm<16 * RB>(kptr[ks]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kptr[ks] += 32 * RB;      // qk is called for st = 0, 1, 2, ... in order
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[1][0]), "+v"(kf[1][1]));
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int s = 0; s < NSUB; ++s) sc[s][t] = T::mfma16(kf[t][ks], qf[s][ks], ks == 0 ? cinit[s] : sc[s][t]);
        };
        // One kv step for all sub-blocks, branch free (one basic block: the scheduler may interleave the
        // independent sub-block chains and the MFMAs with the VALU stream).
        auto softmax_pv = [&](int st, f32x4 (&sc)[NSUB][2], f32x4 (&nxt)[NSUB][2], bool has_next, bool masked) {
            u32x2 v0[DT], v1[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                v0[dt] = (dt & 1) ? lds_tr16_asm<8>(vptr[dt >> 1]) : lds_tr16_asm<0>(vptr[dt >> 1]);
                v1[dt] = (dt & 1) ? lds_tr16_asm<16 * RB + 8>(vptr[dt >> 1]) : lds_tr16_asm<16 * RB>(vptr[dt >> 1]);
            }
            vptr[0] += 32 * RB; vptr[1] += 32 * RB;         // softmax_pv is called for st = 0, 1, 2, ... in order
            if (masked) {                                    // ragged tail: kv >= n_kv -> -inf (selects, no branch)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool dead = st * 32 + t * 16 + 4 * g + r >= a.n_kv;
#pragma unroll
                        for (int s = 0; s < NSUB; ++s) sc[s][t][r] = dead ? -INFINITY : sc[s][t][r];
                    }
            }
            // Lazy rescale: m_run is a REFERENCE max, refreshed only when some query of the wave sees a score more
            // than 8 log2 units above it (p <= 256 in between: harmless in fp32 sums and in 16-bit P) and on the
            // first step.  Refreshing rescales the accumulators, moves cinit, and re-bases the scores that are
            // already in registers (this step's and, double buffered, the next step's).
            float mc[NSUB];
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                float mx = __builtin_fmaxf(__builtin_fmaxf(sc[s][0][0], sc[s][0][1]), sc[s][0][2]);
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, sc[s][0][3]), sc[s][1][0]);
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, sc[s][1][1]), sc[s][1][2]);
                mx = __builtin_fmaxf(mx, sc[s][1][3]);
                mc[s] = mx;                                  // this lane's 8 kv rows only: enough for the test
            }
            // The refresh is decided PER SUB-BLOCK (round 4): a sub-block's reference moves when one of ITS OWN scores ran away, never
            // because a wave-mate's did -- which sub-blocks share a wave depends on the launch form (one / two / four workgroups per
            // (crop, head), chosen by batch size), and the result must not (the sharded tower reproduces the 1-GPU tensor bit for bit).
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                if (__builtin_amdgcn_ballot_w64(mc[s] > RESCALE_TH) != 0 || st == 0) {    // wave-uniform, rare
                    float d = rows_allmax(mc[s]);            // excess of this step's row max over the reference
                    d = st == 0 ? d : __builtin_fmaxf(d, 0.f);
                    const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
                    for (int dd = 0; dd <= DT; ++dd) o[s][dd] *= alpha;
                    m_run[s] += d;
                    const float c = -m_run[s];
                    cinit[s] = f32x4{c, c, c, c};
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            sc[s][t][r] -= d;
                            if (has_next) nxt[s][t][r] -= d;
                        }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0[0]), "+v"(v0[1]), "+v"(v0[2]), "+v"(v0[3]),
                                                   "+v"(v1[0]), "+v"(v1[1]), "+v"(v1[2]), "+v"(v1[3]));
            u32x4 vf[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vf[dt] = u32x4{v0[dt][0], v0[dt][1], v1[dt][0], v1[dt][1]};
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                float p[8];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[t * 4 + r] = __builtin_amdgcn_exp2f(sc[s][t][r]);
                const u32x4 pf = pack8<T>(p);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[s][dt] = T::mfma16(vf[dt], pf, o[s][dt]);
                o[s][DT] = T::mfma16(ones, pf, o[s][DT]);
            }
        };
        auto second_half_ready = [&](int st_next) {          // before the first read of kv rows >= ROWS1
            if (st_next == ROWS1 / 32) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        f32x4 sA[NSUB][2], sB[NSUB][2];
        const bool ragged = (a.n_kv & 31) != 0;
        qk(0, sA);
        int st = 0;
        for (; st + 2 < steps; st += 2) {                    // neither st nor st+1 is the last step
            second_half_ready(st + 1);
            qk(st + 1, sB);                                  // next step's scores under this step's softmax
            softmax_pv(st, sA, sB, true, false);
            second_half_ready(st + 2);
            qk(st + 2, sA);
            softmax_pv(st + 1, sB, sA, true, false);
        }
        if (st + 2 == steps) {
            second_half_ready(st + 1);
            qk(st + 1, sB);
            softmax_pv(st, sA, sB, true, false);
            softmax_pv(st + 1, sB, sA, false, ragged);
        } else {
            softmax_pv(st, sA, sB, false, ragged);
        }

        char* obase = a.o + ((size_t)b * a.o_bs + (size_t)h * DH) * 2;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const float inv = 1.0f / rows_allsum(o[s][DT][0]);    // row 0 of the sum tile lives in quad 0, r = 0
            const int qr = (sb0 + s) * 16 + li;
            if (qr < a.n_q) {
#pragma unroll
                for (int p = 0; p < DT / 2; ++p) {
                    float v[8] = {o[s][2 * p][0] * inv, o[s][2 * p][1] * inv, o[s][2 * p][2] * inv, o[s][2 * p][3] * inv,
                                  o[s][2 * p + 1][0] * inv, o[s][2 * p + 1][1] * inv, o[s][2 * p + 1][2] * inv, o[s][2 * p + 1][3] * inv};
                    st_stream(reinterpret_cast<u32x4*>(obase + ((size_t)qr * a.o_rs + 32 * p + 8 * g) * 2), pack8<T>(v));
                }
            }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(512) attn64_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 8;
    const int h = blockIdx.x, b = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total_sb = (a.n_q + 15) >> 4;
    const int wg_sb0 = blockIdx.z * a.sb_per_wg;
    const int nsb = min(a.sb_per_wg, total_sb - wg_sb0);
    const int base = nsb / NW, rem = nsb % NW;
    const int cnt = base + (wave < rem ? 1 : 0);
    const int sb0 = wg_sb0 + wave * base + min(wave, rem);
    switch (cnt) {
        case 0: attn64_body<T, 0>(a, smem, b, h, sb0); break;
        case 1: attn64_body<T, 1>(a, smem, b, h, sb0); break;
        case 2: attn64_body<T, 2>(a, smem, b, h, sb0); break;
        default: attn64_body<T, 3>(a, smem, b, h, sb0); break;
    }
}

// ================================================================================================
// attn64r_kernel: ONE workgroup per (crop, head) -- K and V travel HBM/L2 -> LDS once instead of twice.
//
// attn64_kernel splits the 37 query sub-blocks of a CLIP head over two workgroups (3 sub-blocks per wave is what the
// register file holds with double-buffered score tiles), so each of the two stages the full 2 x 76 KiB K/V panel: 304 KiB of
// LDS-DMA per (crop, head) against ~25k cycles of MFMA work -- at the ~11 B/cycle a CU's load path sustains the staging is
// as long as the arithmetic, and half of it sits exposed in front of each workgroup's first MFMA.  Here a workgroup keeps
// the panel and walks its queries in TWO passes: pass 1 is attn64's pipeline on 3 sub-blocks per wave (24 of 37), pass 2
// re-reads the resident panel for the remaining sub-blocks (2 or 1 per wave), with no DMA and no barrier at all.
// The DMA is progressive: 64-row granules (one K and one V piece per wave per granule, issued in row order), and a wave
// waits only for the granule it is about to touch (counted vmcnt + barrier), so the first MFMA starts after 16 KiB, not
// 80 KiB, and the rest of the panel streams in under the arithmetic -- three granules ahead of their use (round 3; requesting
// the whole panel at once made every workgroup's first granule queue behind everybody else's panels).
// ================================================================================================
// RING (rows, a power of two; 0 = the resident 608-row panel): K and V live in a ring of RING rows each -- granule gi sits in slot
// gi mod (RING / 8 NW) -- so a workgroup needs 2 x RING x 128 bytes of LDS instead of 152 KiB (attn64g_kernel below).
#ifndef SLIME_OPT_ATTN_SHORT_TAIL
#define SLIME_OPT_ATTN_SHORT_TAIL 1
#endif
// VABL (diagnostic build, variant 40; WRONG results, right timing): the eight ds_read_b64_tr_b16 V^T reads of a kv step are replaced
// by eight plain ds_read_b64 of the same step's 4 KiB at lane-linear addresses (64 lanes x 8 B = every bank exactly once per pass:
// conflict free) -- the same number of LDS instructions, bytes and waits without the 2-way bank conflict.  The difference to the
// product kernel is the exact price of that conflict (VERDICT r5 item 5 i).
template <typename T, int NSUB, bool RESIDENT, int NW = 8, bool KPF = false, int AHEAD = 3, int RING = 0, int VABL = 0>
__device__ __forceinline__ void attn64r_pass(const AttnArgs& a, char* smem, const int b, const int h, const int sb0,
                                             unsigned long long* t_first = nullptr) {    // diagnostic: when the first granule was ready
    constexpr int DH = 64, RB = 128, KS = 2, DT = 4, KC = 608;
    static_assert(RING == 0 || (!RESIDENT && !KPF && (RING & (RING - 1)) == 0 && RING % (8 * NW) == 0 && RING / (8 * NW) >= AHEAD + 3),
                  "ring: streaming pass only; a slot is re-filled three granules after its last reader (see granule_ready)");
    constexpr unsigned RMASK = RING ? RING * RB - 1 : 0xffffffffu;
    constexpr int NG = (KC / 8 + NW - 1) / NW;            // DMA granules of NW pieces = 8 NW rows (8 waves: 10 x 64 rows, 12 waves: 7 x 96 rows;
                                                           // the last one holds 4 real pieces, rows 576..607)
    constexpr int SPG = NW / 4;                            // 32-row kv steps per granule
    constexpr float RESCALE_TH = 8.0f;
    char* Klds = smem;
    char* Vlds = smem + (RING ? RING : KC) * RB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;

    // K/V panel source (pass 1 only)
    const char* kbase = a.k + ((size_t)b * a.k_bs + (size_t)h * DH) * 2;
    const char* vbase = a.v + ((size_t)b * a.v_bs + (size_t)h * DH) * 2;
    const int lrow = lane >> 3, cpos = lane & 7;
    const int kchunk = cpos ^ lrow;
    const int vchunk = cpos ^ (((lrow >> 1) & 1) << 2);
    auto dma_granule = [&](int gi) {
        // granule gi = pieces 8 gi .. 8 gi + 7; the last granule has 4 pieces: waves 4..7 repeat them (same bytes to the
        // same place) so that every wave has issued exactly 2 (gi + 1) DMAs after granule gi -- the counted waits below
        int piece = gi * NW + wave;
        if (piece >= KC / 8) piece = KC / 8 - 4 + (wave & 3);
        const int row = min(piece * 8 + lrow, a.n_kv - 1);
        const int slot = RING ? piece % (RING / 8) : piece;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(kbase + ((size_t)row * a.k_rs) * 2 + kchunk * 16),
                                         LDS_PTR(Klds + slot * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(vbase + ((size_t)row * a.v_rs) * 2 + vchunk * 16),
                                         LDS_PTR(Vlds + slot * 1024), 16, 0, 0);
    };
    u32x4 qf[NSUB > 0 ? NSUB : 1][KS];
    const char* qbase = a.q + ((size_t)b * a.q_bs + (size_t)h * DH) * 2;
    if constexpr (RESIDENT) {
        if constexpr (NSUB > 0) {
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                const int qr = min((sb0 + s) * 16 + li, a.n_q - 1);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    qf[s][ks] = *reinterpret_cast<const u32x4*>(qbase + ((size_t)qr * a.q_rs + ks * 32 + g * 8) * 2);
            }
#pragma unroll
            for (int s = 0; s < NSUB; ++s)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[s][ks]));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // Round 3 (tools/attn64r_stamps.py): a workgroup spent 31 % of its life between its first instruction and its first K/V
        // granule -- two memory round trips in a row (Q rows, wait, THEN the first DMA) on a CU that holds one workgroup (152 KiB
        // of LDS) and so has nothing else to run.  Now ONE: granule 0's DMAs, the Q rows (asm loads: the compiler must not count
        // them), the other granules -- loads return in order, so the first granule's counted wait (at most 2 (NG - 1) DMAs
        // outstanding) covers the Q rows that sit in the queue in front of those.
        dma_granule(0);
        if constexpr (NSUB > 0) {
            static_assert(KS == 2, "two 16-byte Q pieces per lane and sub-block");
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                const int qr = min((sb0 + s) * 16 + li, a.n_q - 1);
                const char* qp = qbase + ((size_t)qr * a.q_rs + g * 8) * 2;
                asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:64"
                             : "=&v"(qf[s][0]), "=&v"(qf[s][1]) : "v"(qp) : "memory");
            }
        }
#pragma unroll
        for (int gi = 1; gi < NG && gi < AHEAD; ++gi) dma_granule(gi);
    }
    // granule gi has landed for THIS wave when at most 2 (NG - 1 - gi) of its DMAs are outstanding; the barrier extends that
    // to every wave's pieces.  Called by every wave of the workgroup for gi = 0, 1, 2, ... in order (uniform control flow).
    // Only AHEAD granules are requested up front, granule gi + AHEAD behind the barrier of granule gi.  Round 3
    // (tools/attn64r_stamps.py, profiles/r03_attn64r_depth.txt): with the whole panel requested at once (rounds 1-2) a workgroup
    // spent 31 % of its life waiting for its FIRST granule -- behind the up-to-152 KiB the other CUs of its XCD had queued each;
    // with 3 ahead the queues are shallow: start-up 13.4 k -> 7.3 k ticks, the steps themselves 6 % faster, kernel 61.4 -> 53.2 us
    // at 20 crops, 105 -> 94 at 40, 18.6 -> 17.4 at 5; bit-equal (same arithmetic, same LDS layout).  2 ahead measures the same.
    // RING: granule gi + AHEAD goes into the slot of granule gi + AHEAD - RING / (8 NW).  A wave that has passed the barrier of
    // granule gi has finished the K reads of every step before granule gi - 1 and the V reads before granule gi - 2 (qk of step
    // st + 1 and softmax_pv of step st follow before_step(st + 2)), and the barrier extends that to every wave: granules <= gi - 3
    // are dead, hence ring slots >= AHEAD + 3.
    auto granule_ready = [&](int gi) {
        if constexpr (!RESIDENT) {
            const int fly = min(AHEAD - 1, NG - 1 - gi);        // granules still allowed in flight
            switch (fly) {
                case 9: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (AHEAD < NG) { if (gi + AHEAD < NG) dma_granule(gi + AHEAD); }
        }
    };
    auto before_step = [&](int st) { if (st % SPG == 0) granule_ready(st / SPG); };   // step st reads rows 32 st .. 32 st + 31

    const int steps = (a.n_kv + 31) >> 5;
    if constexpr (NSUB == 0) {
        if constexpr (!RESIDENT) {
            for (int st = 0; st < steps; ++st) before_step(st);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        return;
    } else {
        f32x4 o[NSUB][DT + 1], cinit[NSUB];
        float m_run[NSUB];
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            m_run[s] = 0.f;
            cinit[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int d = 0; d <= DT; ++d) o[s][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const unsigned one2 = T::pack2(1.0f, 1.0f);
        const unsigned onew = (li == 0) ? one2 : 0u;
        const u32x4 ones = {onew, onew, onew, onew};
        unsigned kptr[KS], vptr[2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kptr[ks] = lds_addr(Klds) + li * RB + (((ks * 4 + g) ^ (lane & 7)) << 4);
        {
            const int vrow = 4 * g + (li >> 2);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                vptr[hh] = lds_addr(Vlds) + vrow * RB + (((4 * hh + (li & 3)) ^ (((li >> 3) & 1) << 2)) << 4);
        }
        // KPF (measured alternative, off): K fragments fetched one call ahead -- qk() multiplies the fragments the PREVIOUS call
        // requested and requests the next step's before it returns, so their LDS latency passes under the softmax / PV work
        // instead of in front of the first QK MFMA.  Neutral on the clock (58.3 vs 58.1 us per 20 crops) at +16 live VGPRs,
        // like the 12-wave variant below: the kernel is not bound by exposed LDS latency.
        u32x4 kf[2][KS];
        unsigned kring = 0, vring = 0;                        // RING: byte offset of the current 32-row step inside the ring (wave-uniform)
        [[maybe_unused]] unsigned vabl = lds_addr(Vlds) + lane * 8;   // VABL: lane-linear (conflict-free) read address of the current step
        auto k_request = [&]() {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kf[t][ks] = t == 0 ? lds_b128_asm<0>(kptr[ks] + kring) : lds_b128_asm<16 * RB>(kptr[ks] + kring);
            if constexpr (RING) kring = (kring + 32 * RB) & RMASK;
            else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kptr[ks] += 32 * RB;
            }
        };
        // short == true (compile-time at every call site): the LAST step of a ragged key range that fills at most its first 16 rows
        // (CLIP: 577 = 18 x 32 + 1) -- only the first key tile is multiplied; softmax_pv(short) never looks at sc[.][1]
        auto qk = [&](f32x4 (&sc)[NSUB][2], bool request_next, auto short_c) {
            constexpr bool SHORT = decltype(short_c)::value;
            if constexpr (!KPF) k_request();                  // register-lean form: fetch at the point of use
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[1][0]), "+v"(kf[1][1]));
#pragma unroll
            for (int t = 0; t < (SHORT ? 1 : 2); ++t)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int s = 0; s < NSUB; ++s) sc[s][t] = T::mfma16(kf[t][ks], qf[s][ks], ks == 0 ? cinit[s] : sc[s][t]);
            if (KPF && request_next) {
                // the MFMAs above have READ kf when they issued (in order), but the compiler must not hoist the requests
                asm volatile("" : "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[1][0]), "+v"(kf[1][1]));
                k_request();
            }
        };
        auto softmax_pv = [&](int st, f32x4 (&sc)[NSUB][2], f32x4 (&nxt)[NSUB][2], bool has_next, bool masked, auto short_c) {
            // SHORT (the ragged last step with <= 16 live keys; always masked, never has a next step): one key tile -- 4 V^T reads,
            // 4 exponentials per query and K = 16 MFMAs (half the matrix time) instead of a 32-key step that is 1/32 useful
            constexpr bool SHORT = decltype(short_c)::value;
            u32x2 v0[DT], v1[DT];
            if constexpr (VABL != 0) {
                static_assert(DT == 4 && RING == 0, "ablation: CLIP shape, resident panel");
                v0[0] = lds_b64_asm<0>(vabl); v0[1] = lds_b64_asm<512>(vabl); v0[2] = lds_b64_asm<1024>(vabl); v0[3] = lds_b64_asm<1536>(vabl);
                if constexpr (!SHORT) {
                    v1[0] = lds_b64_asm<2048>(vabl); v1[1] = lds_b64_asm<2560>(vabl); v1[2] = lds_b64_asm<3072>(vabl); v1[3] = lds_b64_asm<3584>(vabl);
                }
                vabl += 32 * RB;
            } else {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    v0[dt] = (dt & 1) ? lds_tr16_asm<8>(vptr[dt >> 1] + vring) : lds_tr16_asm<0>(vptr[dt >> 1] + vring);
                    if constexpr (!SHORT)
                        v1[dt] = (dt & 1) ? lds_tr16_asm<16 * RB + 8>(vptr[dt >> 1] + vring) : lds_tr16_asm<16 * RB>(vptr[dt >> 1] + vring);
                }
            }
            if constexpr (RING) vring = (vring + 32 * RB) & RMASK;
            else { vptr[0] += 32 * RB; vptr[1] += 32 * RB; }
            if (masked || SHORT) {
#pragma unroll
                for (int t = 0; t < (SHORT ? 1 : 2); ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool dead = st * 32 + t * 16 + 4 * g + r >= a.n_kv;
#pragma unroll
                        for (int s = 0; s < NSUB; ++s) sc[s][t][r] = dead ? -INFINITY : sc[s][t][r];
                    }
            }
            float mc[NSUB];
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                float mx = __builtin_fmaxf(__builtin_fmaxf(sc[s][0][0], sc[s][0][1]), sc[s][0][2]);
                if constexpr (SHORT) {
                    mx = __builtin_fmaxf(mx, sc[s][0][3]);
                } else {
                    mx = __builtin_fmaxf(__builtin_fmaxf(mx, sc[s][0][3]), sc[s][1][0]);
                    mx = __builtin_fmaxf(__builtin_fmaxf(mx, sc[s][1][1]), sc[s][1][2]);
                    mx = __builtin_fmaxf(mx, sc[s][1][3]);
                }
                mc[s] = mx;
            }
            // per SUB-BLOCK refresh decision (round 4): see attn64_body -- the result must not depend on which sub-blocks share a wave
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                if (__builtin_amdgcn_ballot_w64(mc[s] > RESCALE_TH) != 0 || st == 0) {
                    float d = rows_allmax(mc[s]);
                    d = st == 0 ? d : __builtin_fmaxf(d, 0.f);
                    const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
                    for (int dd = 0; dd <= DT; ++dd) o[s][dd] *= alpha;
                    m_run[s] += d;
                    const float c = -m_run[s];
                    cinit[s] = f32x4{c, c, c, c};
#pragma unroll
                    for (int t = 0; t < (SHORT ? 1 : 2); ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            sc[s][t][r] -= d;
                            if (has_next) nxt[s][t][r] -= d;
                        }
                }
            }
            if constexpr (SHORT) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0[0]), "+v"(v0[1]), "+v"(v0[2]), "+v"(v0[3]));
                const u32x2 ones2 = {onew, onew};
#pragma unroll
                for (int s = 0; s < NSUB; ++s) {
                    // the lane's four scores of key rows 4 g + r are exactly the K = 16 MFMA's B fragment (4 consecutive k per lane);
                    // v0[dt] = V^T[dh = 16 dt + li][kv = 4 g .. 4 g + 3] is its A fragment
                    const u32x2 pf = {T::pack2(__builtin_amdgcn_exp2f(sc[s][0][0]), __builtin_amdgcn_exp2f(sc[s][0][1])),
                                      T::pack2(__builtin_amdgcn_exp2f(sc[s][0][2]), __builtin_amdgcn_exp2f(sc[s][0][3]))};
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) o[s][dt] = T::mfma16_k16(v0[dt], pf, o[s][dt]);
                    o[s][DT] = T::mfma16_k16(ones2, pf, o[s][DT]);
                }
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0[0]), "+v"(v0[1]), "+v"(v0[2]), "+v"(v0[3]),
                                                       "+v"(v1[0]), "+v"(v1[1]), "+v"(v1[2]), "+v"(v1[3]));
                u32x4 vf[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) vf[dt] = u32x4{v0[dt][0], v0[dt][1], v1[dt][0], v1[dt][1]};
#pragma unroll
                for (int s = 0; s < NSUB; ++s) {
                    float p[8];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) p[t * 4 + r] = __builtin_amdgcn_exp2f(sc[s][t][r]);
                    const u32x4 pf = pack8<T>(p);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) o[s][dt] = T::mfma16(vf[dt], pf, o[s][dt]);
                    o[s][DT] = T::mfma16(ones, pf, o[s][DT]);
                }
            }
        };

        f32x4 sA[NSUB][2], sB[NSUB][2];
        const bool ragged = (a.n_kv & 31) != 0;
        // step st's K rows must have landed before they are REQUESTED, i.e. one qk() call earlier than they are multiplied
        before_step(0);
        if constexpr (!RESIDENT) {                           // (the Q rows landed with granule 0)
#pragma unroll
            for (int s = 0; s < NSUB; ++s) asm volatile("" : "+v"(qf[s][0]), "+v"(qf[s][1]));
        }
#ifdef SLIME_DIAG
        if (t_first) *t_first = __builtin_amdgcn_s_memtime();
#endif
        constexpr std::false_type FULL_STEP{};
        constexpr std::true_type SHORT_STEP{};
        // Ragged key range whose last step holds at most 16 live keys (CLIP: 577 = 18 x 32 + 1; round 5, SLIME_OPT_ATTN_SHORT_TAIL):
        // that step is multiplied half-wide (one key tile, K = 16 MFMAs) instead of as a masked 32-key step.  The choice depends on
        // n_kv alone, never on the launch form, so the bit-invariance across forms and batch sizes holds.
        const bool short_tail = SLIME_OPT_ATTN_SHORT_TAIL && !KPF && ragged && (a.n_kv & 31) <= 16 && steps >= 2;
        if constexpr (KPF) k_request();                      // step 0
        if (steps > 1) before_step(1);
        qk(sA, steps > 1, FULL_STEP);                        // multiplies step 0, requests step 1
        int st = 0;
        for (; st + 2 < steps; st += 2) {
            before_step(st + 2);
            qk(sB, true, FULL_STEP);                         // multiplies step st+1, requests step st+2
            softmax_pv(st, sA, sB, true, false, FULL_STEP);
            if (st + 3 < steps) before_step(st + 3);
            // multiplies step st+2, requests step st+3; step st+2 may be the short last one (odd step counts: CLIP's 19)
            if (short_tail && st + 3 == steps) qk(sA, false, SHORT_STEP);
            else qk(sA, st + 3 < steps, FULL_STEP);
            softmax_pv(st + 1, sB, sA, true, false, FULL_STEP);
        }
        if (st + 2 == steps) {
            if (short_tail) qk(sB, false, SHORT_STEP);
            else qk(sB, false, FULL_STEP);
            softmax_pv(st, sA, sB, true, false, FULL_STEP);
            if (short_tail) softmax_pv(st + 1, sB, sA, false, true, SHORT_STEP);
            else softmax_pv(st + 1, sB, sA, false, ragged, FULL_STEP);
        } else {
            if (short_tail) softmax_pv(st, sA, sB, false, true, SHORT_STEP);
            else softmax_pv(st, sA, sB, false, ragged, FULL_STEP);
        }
        if constexpr (!RESIDENT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // granules beyond the last step (short n_kv)

        char* obase = a.o + ((size_t)b * a.o_bs + (size_t)h * DH) * 2;
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            const float inv = 1.0f / rows_allsum(o[s][DT][0]);
            const int qr = (sb0 + s) * 16 + li;
            if (qr < a.n_q) {
#pragma unroll
                for (int p = 0; p < DT / 2; ++p) {
                    float v[8] = {o[s][2 * p][0] * inv, o[s][2 * p][1] * inv, o[s][2 * p][2] * inv, o[s][2 * p][3] * inv,
                                  o[s][2 * p + 1][0] * inv, o[s][2 * p + 1][1] * inv, o[s][2 * p + 1][2] * inv, o[s][2 * p + 1][3] * inv};
                    st_stream(reinterpret_cast<u32x4*>(obase + ((size_t)qr * a.o_rs + 32 * p + 8 * g) * 2), pack8<T>(v));
                }
            }
        }
    }
}

#ifndef SLIME_OPT_ATTN_PAIR
#define SLIME_OPT_ATTN_PAIR 1
#endif
template <typename T, int AHEAD = 3, int VABL = 0>
__global__ void __launch_bounds__(512) attn64r_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 8, P1 = 3 * NW;                        // pass 1: <= 3 sub-blocks per wave
    int h = blockIdx.x, b = blockIdx.y, z = blockIdx.z;
#if SLIME_OPT_ATTN_PAIR
    // Two workgroups share a (crop, head) when the query blocks are split: both stage the same K/V panel.  The launch order deals
    // workgroup L to XCD L & 7 (each XCD has its own L2), and in grid order the partners are heads x crops apart -- the second one
    // fetched the panel across the fabric again (117 MB read per launch against 71 MB of q/k/v, profiles/r04_pmc_kernels.json).
    // Re-deal: partners are L and L + 8, neighbours in time on the same XCD.
    if (gridDim.z > 1 && ((gridDim.x * gridDim.y) & 7) == 0) {
        const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int j = L >> 3, p = (j / (int)gridDim.z) * 8 + (L & 7);
        z = j % (int)gridDim.z; h = p % (int)gridDim.x; b = p / (int)gridDim.x;
    }
#endif
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total_sb = (a.n_q + 15) >> 4;
    const int wg_sb0 = z * a.sb_per_wg;
    const int nsb = min(a.sb_per_wg, total_sb - wg_sb0);      // <= 5 * NW
    const int n1 = min(nsb, P1), n2 = nsb - n1;               // pass 2: <= 2 per wave
#ifdef SLIME_DIAG
    // diagnostic variant 17: a.dbg holds one record of 8 words per workgroup (s_memtime stamps of wave 0)
    unsigned long long dt0 = __builtin_amdgcn_s_memtime(), dtf = dt0;
    unsigned long long* tfp = a.dbg ? &dtf : nullptr;
#else
    constexpr unsigned long long* tfp = nullptr;
#endif
    {
        const int base = n1 / NW, rem = n1 % NW;
        const int cnt = base + (wave < rem ? 1 : 0);
        const int sb0 = wg_sb0 + wave * base + min(wave, rem);
        switch (cnt) {
            case 0: attn64r_pass<T, 0, false, NW, false, AHEAD, 0, VABL>(a, smem, b, h, sb0); break;
            case 1: attn64r_pass<T, 1, false, NW, false, AHEAD, 0, VABL>(a, smem, b, h, sb0, tfp); break;
            case 2: attn64r_pass<T, 2, false, NW, false, AHEAD, 0, VABL>(a, smem, b, h, sb0, tfp); break;
            default: attn64r_pass<T, 3, false, NW, false, AHEAD, 0, VABL>(a, smem, b, h, sb0, tfp); break;
        }
    }
#ifdef SLIME_DIAG
    const unsigned long long dt1 = __builtin_amdgcn_s_memtime();
#endif
    if (n2 > 0) {
        // every wave left pass 1 behind the last granule's barrier: the whole panel is resident and read-only from here on.
        // The waves that got the most work in pass 1 (low ids when n1 % 8 != 0) get the least here.
        const int base = n2 / NW, rem = n2 % NW;
        const int rw = NW - 1 - wave;
        const int cnt = base + (rw < rem ? 1 : 0);
        const int sb0 = wg_sb0 + n1 + rw * base + min(rw, rem);
        switch (cnt) {
            case 0: break;
            case 1: attn64r_pass<T, 1, true, NW, false, 3, 0, VABL>(a, smem, b, h, sb0); break;
            default: attn64r_pass<T, 2, true, NW, false, 3, 0, VABL>(a, smem, b, h, sb0); break;
        }
    }
#ifdef SLIME_DIAG
    if (a.dbg && threadIdx.x == 0) {                          // one record per workgroup (wave 0): entry, first granule, end of pass 1, end, where
        const unsigned long long dt2 = __builtin_amdgcn_s_memtime();
        unsigned long long* r = a.dbg + 8 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        r[0] = dt0; r[1] = dtf; r[2] = dt1; r[3] = dt2;
        r[4] = (unsigned long long)__builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11)) |
               ((unsigned long long)__builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11)) << 32);
    }
#endif
}

// Twelve-wave variant: three waves per SIMD (<= 2 sub-blocks each, <= 168 VGPRs) instead of two with three sub-blocks.
// rocprofv3 on attn64r: 36 % of the wave cycles are parked at s_waitcnt / s_barrier and 33 % stalled at issue -- latency, not
// throughput -- so the third wave per SIMD is there to cover the other two's waits.  24 sub-blocks per workgroup at most,
// i.e. always two workgroups per CLIP (crop, head); pass 1 only.
template <typename T>
__global__ void __launch_bounds__(768) attn64w_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 12;
    const int h = blockIdx.x, b = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total_sb = (a.n_q + 15) >> 4;
    const int wg_sb0 = blockIdx.z * a.sb_per_wg;
    const int nsb = min(a.sb_per_wg, total_sb - wg_sb0);      // <= 2 * NW
    const int base = nsb / NW, rem = nsb % NW;
    const int cnt = base + (wave < rem ? 1 : 0);
    const int sb0 = wg_sb0 + wave * base + min(wave, rem);
    switch (cnt) {
        case 0: attn64r_pass<T, 0, false, NW, false>(a, smem, b, h, sb0); break;
        case 1: attn64r_pass<T, 1, false, NW, false>(a, smem, b, h, sb0); break;
        default: attn64r_pass<T, 2, false, NW, false>(a, smem, b, h, sb0); break;
    }
}

template <typename T>
static int launch_attn64w(const AttnArgs& a0, int batch, hipStream_t stream) {
    AttnArgs a = a0;
    constexpr int LDS = 2 * 608 * 128;
    auto kern = attn64w_kernel<T>;
    SLIME_SET_LDS_ONCE(kern, LDS, "attention");
    const int total_sb = (a.n_q + 15) / 16;
    const int qsplit = (total_sb + 23) / 24;
    a.sb_per_wg = (total_sb + qsplit - 1) / qsplit;
    hipLaunchKernelGGL(kern, dim3(a.heads, batch, qsplit), dim3(768), LDS, stream, a);
    SLIME_CHECK_LAUNCH("attention64w");
    return SLIME_OK;
}

#ifdef SLIME_DIAG
// ================================================================================================
// attn64g_kernel (round 4, measured alternative): the same pass on a K/V RING -- four waves, 2 x 32 KiB of LDS, <= 256 registers --
// so that an attention workgroup no longer monopolises its CU (attn64r: 152 KiB + 8 waves x 245 registers) but can share it with
// another attention workgroup or with a direct-B GEMM workgroup of the tower's other stream (4 waves x 256 registers, 33 KiB):
// VALU / LDS-heavy softmax waves beside MFMA / L2-bound GEMM waves.  12 query sub-blocks per workgroup (3 per wave), i.e. four
// workgroups per CLIP (crop, head), each streaming the whole K/V once through the ring (granule = 32 rows = one step, 8 slots,
// 4 ahead).  One-dimensional grid: the four workgroups of an item are 8 apart in launch order -- same XCD, same L2 -- and items
// of 8 consecutive heads fill the 8 XCDs.  Same arithmetic in the same order per query sub-block: bit-identical to attn64r.
// Measured (tools/attn_ring_ab.py, profiles/r04_attention_ring.txt): stand-alone 52 -> 46-48 us at 20 crops, 87 -> 81-83 at 40,
// 16 -> 13.4 at one crop, equal at 5-9; the two-stream tower 15.28-15.34 -> 15.51-15.52 ms (SLOWER), one stream equal.  Like attn32
// in rounds 2-3: under the power cap a faster attention that does the same work moves the step nowhere.  Diagnostic build only.
// ================================================================================================
template <typename T>
__global__ void __launch_bounds__(256, 2) attn64g_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 4, AHEAD = 4, RING = 256;
    const int qsplit = a.sb_per_wg >> 16, sb_per_wg = a.sb_per_wg & 0xffff;
    // launch index -> (item, split): id = 8 qsplit (item / 8) + 8 split + item % 8
    const int id = blockIdx.x, grp = id / (8 * qsplit), rem = id % (8 * qsplit);
    const int item = grp * 8 + (rem & 7), split = rem >> 3;
    if (item >= a.n_items) return;                            // the grid is padded to whole groups of 8 items
    const int h = item % a.heads, b = item / a.heads;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total_sb = (a.n_q + 15) >> 4;
    const int wg_sb0 = split * sb_per_wg;
    const int nsb = max(0, min(sb_per_wg, total_sb - wg_sb0));     // <= 3 * NW
    const int base = nsb / NW, rm = nsb % NW;
    const int cnt = base + (wave < rm ? 1 : 0);
    const int sb0 = wg_sb0 + wave * base + min(wave, rm);
    switch (cnt) {
        case 0: attn64r_pass<T, 0, false, NW, false, AHEAD, RING>(a, smem, b, h, sb0); break;
        case 1: attn64r_pass<T, 1, false, NW, false, AHEAD, RING>(a, smem, b, h, sb0); break;
        case 2: attn64r_pass<T, 2, false, NW, false, AHEAD, RING>(a, smem, b, h, sb0); break;
        default: attn64r_pass<T, 3, false, NW, false, AHEAD, RING>(a, smem, b, h, sb0); break;
    }
}

template <typename T>
static int launch_attn64g(const AttnArgs& a0, int batch, hipStream_t stream) {
    AttnArgs a = a0;
    constexpr int LDS = 2 * 256 * 128;
    auto kern = attn64g_kernel<T>;
    SLIME_SET_LDS_ONCE(kern, LDS, "attention");
    const int total_sb = (a.n_q + 15) / 16;
    const int qsplit = (total_sb + 11) / 12;                  // <= 3 sub-blocks per wave, 4 waves
    const int per = (total_sb + qsplit - 1) / qsplit;
    a.sb_per_wg = (qsplit << 16) | per;
    a.n_items = a.heads * batch;
    const int groups = (a.n_items + 7) / 8;
    hipLaunchKernelGGL(kern, dim3(groups * 8 * qsplit), dim3(256), LDS, stream, a);
    SLIME_CHECK_LAUNCH("attention64g");
    return SLIME_OK;
}
#endif  // SLIME_DIAG

template <typename T, int AHEAD = 3, int VABL = 0>
static int launch_attn64r(const AttnArgs& a0, int batch, hipStream_t stream) {
    AttnArgs a = a0;
    constexpr int LDS = 2 * 608 * 128;
    auto kern = attn64r_kernel<T, AHEAD, VABL>;
    SLIME_SET_LDS_ONCE(kern, LDS, "attention");
    const int total_sb = (a.n_q + 15) / 16;
    int qsplit = (total_sb + 39) / 40;                        // <= 3 + 2 sub-blocks per wave, 8 waves
    // One or two workgroups per (crop, head)?  Two re-stage the K/V panel (a half-size workgroup costs ~0.56 of a full one,
    // measured) but quantise better: rounds of CUs x cost per workgroup decides.  5 crops: 80 -> 160 workgroups, one round
    // either way, 30 -> 18 us; 20 crops: 320 = 2 rounds x 1.0 vs 640 = 3 rounds x 0.56; 10 crops: 160 = 1 round, stays.
    if (total_sb >= 16) {
        const long cus = num_cus(), items = (long)a.heads * batch * qsplit;
        const double one = (double)((items + cus - 1) / cus), two = (double)((2 * items + cus - 1) / cus) * 0.56;
        if (two < one) qsplit *= 2;
    }
    a.sb_per_wg = (total_sb + qsplit - 1) / qsplit;
    hipLaunchKernelGGL(kern, dim3(a.heads, batch, qsplit), dim3(512), LDS, stream, a);
    SLIME_CHECK_LAUNCH("attention64r");
    return SLIME_OK;
}

template <typename T>
static int launch_attn64(const AttnArgs& a0, int batch, hipStream_t stream) {
    AttnArgs a = a0;
    constexpr int LDS = 2 * 608 * 128;
    auto kern = attn64_kernel<T>;
    SLIME_SET_LDS_ONCE(kern, LDS, "attention");
    const int total_sb = (a.n_q + 15) / 16;
    const int qsplit = (total_sb + 23) / 24;                  // <= 3 sub-blocks per wave, 8 waves
    a.sb_per_wg = (total_sb + qsplit - 1) / qsplit;
    hipLaunchKernelGGL(kern, dim3(a.heads, batch, qsplit), dim3(512), LDS, stream, a);
    SLIME_CHECK_LAUNCH("attention64");
    return SLIME_OK;
}

#include "attention32.inc"

#ifdef SLIME_DIAG   // diagnostic build only (libslime_hip_diag.so): the product library has no mutable globals
static unsigned long long* g_attn_dbg = nullptr;
static int g_attn_abl = 0;
extern "C" void slime_attention_set_ablation(int v) { g_attn_abl = v; }
static int g_attn_variant = 0;      // 1 = force the generic kernel
extern "C" void slime_attention_set_variant(int v) { g_attn_variant = v; }
extern "C" void slime_attention_set_debug(void* p) { g_attn_dbg = (unsigned long long*)p; }
#else
static constexpr unsigned long long* g_attn_dbg = nullptr;
static constexpr int g_attn_abl = 0, g_attn_variant = 0;
#endif

extern "C" int slime_attention(const void* q, long q_bs, long q_rs, const void* k, long k_bs, long k_rs,
                               const void* v, long v_bs, long v_rs, void* o, long o_bs, long o_rs,
                               int batch, int heads, int head_dim, int n_q, int n_kv, int dtype, void* stream) {
    SLIME_REQUIRE(q && k && v && o, "attention: null pointer");
    SLIME_REQUIRE(batch > 0 && heads > 0 && n_q > 0 && n_kv > 0, "attention: empty shape");
    SLIME_REQUIRE(head_dim == 64 || head_dim == 128, "attention: head_dim %d unsupported (64, 128)", head_dim);
    SLIME_REQUIRE(q_rs % 8 == 0 && k_rs % 8 == 0 && v_rs % 8 == 0 && o_rs % 8 == 0 &&
                  q_bs % 8 == 0 && k_bs % 8 == 0 && v_bs % 8 == 0 && o_bs % 8 == 0, "attention: strides must be multiples of 8 elements");
    SLIME_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16 == 0, "attention: pointers must be 16-byte aligned");
    SLIME_REQUIRE(batch <= 65535, "attention: batch %d exceeds grid.y", batch);
    SLIME_REQUIRE(dtype == SLIME_BF16 || dtype == SLIME_F16, "attention: dtype %d is not a 16-bit MFMA type", dtype);
    AttnArgs a{(const char*)q, q_bs, q_rs, (const char*)k, k_bs, k_rs, (const char*)v, v_bs, v_rs,
               (char*)o, o_bs, o_rs, heads, n_q, n_kv, 0, g_attn_dbg, g_attn_abl};
    hipStream_t s = (hipStream_t)stream;
    if (head_dim == 64 && n_kv <= 608 && n_kv >= 321 && g_attn_variant == 0 && !g_attn_dbg) {
        // CLIP shape: software-pipelined kernel, one or two workgroups per (crop, head), K/V staged once (granule DMA).
        // Round 3 re-measured the one-wave-per-SIMD alternative (attention32.inc, diagnostic variants 4-6) now that the GEMMs run two
        // workgroups per CU: its uncut form is worth 1 % of the 40-crop tower (15.18 -> 15.01 ms, profiles/r03_tower_knobs.txt) but
        // costs 0.25 ms per pass at <= 8 crops (27 instead of 18 us per launch), and its cut forms, which fix that, are not bit-equal
        // to it: a handful of isolated rows (4-10 of 157k-370k on random data: the rows whose reference maximum moves, deterministic
        // run to run) depend on how many blocks share their wave -- and the sharded tower must reproduce the 1-GPU tensor bit for
        // bit at every shard size.  It stays a measured alternative (tools/attn32_cut_invariance.py).
        if (dtype == SLIME_F16) return launch_attn64r<F16>(a, batch, s);
        return launch_attn64r<BF16>(a, batch, s);
    }
#ifdef SLIME_DIAG
    if (head_dim == 64 && n_kv <= 608 && n_kv >= 321 && g_attn_variant >= 4 && g_attn_variant <= 6 && dtype == SLIME_BF16)
        return launch_attn32<BF16>(a, batch, g_attn_variant == 5 ? 2 : g_attn_variant == 6 ? -1 : 0, s);
    if (head_dim == 64 && n_kv <= 608 && n_kv >= 321 && g_attn_variant >= 12 && g_attn_variant <= 15 && dtype == SLIME_BF16)
        return launch_attn32<BF16>(a, batch, g_attn_variant - 10, s);                       // every item cut in 2 / 3 / 4 / 5
    if (head_dim == 64 && n_kv <= 608 && n_kv >= 321 && g_attn_variant >= 22 && g_attn_variant <= 27 && dtype == SLIME_BF16) {
        // attn64r with 2 / 3 / 4 / 6 K/V granules requested ahead, 27: the whole panel up front (rounds 1-2); with or without stamp records
        switch (g_attn_variant) {
            case 22: return launch_attn64r<BF16, 2>(a, batch, s);
            case 23: return launch_attn64r<BF16, 3>(a, batch, s);
            case 24: return launch_attn64r<BF16, 4>(a, batch, s);
            case 26: return launch_attn64r<BF16, 6>(a, batch, s);
            default: return launch_attn64r<BF16, 16>(a, batch, s);
        }
    }
    if (head_dim == 64 && n_kv <= 608 && n_kv >= 321 && g_attn_variant == 40 && dtype == SLIME_BF16 && !g_attn_dbg)
        return launch_attn64r<BF16, 3, 1>(a, batch, s);      // round 6 timing ablation: conflict-free plain reads instead of the V^T transpose reads (wrong results)
    if (head_dim == 64 && n_kv <= 608 && n_kv >= 321 && g_attn_variant == 17 && g_attn_dbg) {   // attn64r with one stamp record per workgroup
        if (dtype == SLIME_F16) return launch_attn64r<F16>(a, batch, s);
        return launch_attn64r<BF16>(a, batch, s);
    }
    if (head_dim == 64 && n_kv <= 608 && n_kv >= 321 && g_attn_variant == 7 && !g_attn_dbg) {    // round 2's product kernel, for A/B
        if (dtype == SLIME_F16) return launch_attn64r<F16>(a, batch, s);
        return launch_attn64r<BF16>(a, batch, s);
    }
    if (head_dim == 64 && n_kv <= 608 && n_kv >= 321 && g_attn_variant == 30 && !g_attn_dbg) {   // round 4: K/V ring, four waves, two per CU
        if (dtype == SLIME_F16) return launch_attn64g<F16>(a, batch, s);
        return launch_attn64g<BF16>(a, batch, s);
    }
    if (head_dim == 64 && n_kv <= 608 && n_kv >= 321 && g_attn_variant == 3 && !g_attn_dbg) {
        if (dtype == SLIME_F16) return launch_attn64w<F16>(a, batch, s);
        return launch_attn64w<BF16>(a, batch, s);
    }
    if (head_dim == 64 && n_kv <= 608 && n_kv >= 321 && g_attn_variant == 2 && !g_attn_dbg) {
        // the round-1 kernel (two workgroups per (crop, head), two DMA halves), kept for A/B
        if (dtype == SLIME_F16) return launch_attn64<F16>(a, batch, s);
        return launch_attn64<BF16>(a, batch, s);
    }
#endif
    if (head_dim == 64) {
        // K+V resident up to 608 rows (CLIP S = 577); longer sequences stream in 608-row chunks.
        if (dtype == SLIME_F16) return launch_attn<F16, 64, 608, 8, 5>(a, batch, s);
        return launch_attn<BF16, 64, 608, 8, 5>(a, batch, s);
    }
    if (dtype == SLIME_F16) return launch_attn<F16, 128, 288, 8, 2>(a, batch, s);
    return launch_attn<BF16, 128, 288, 8, 2>(a, batch, s);
}
