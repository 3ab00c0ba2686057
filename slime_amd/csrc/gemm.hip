// gemm.hip -- MFMA GEMM with fused epilogues for gfx950:  C = epi(A[M,K] * B[N,K]^T + bias).
//
// Both operands are K-contiguous (activations row-major, weights in nn.Linear layout), which is
// exactly what v_mfma_f32_16x16x32_{bf16,f16} wants: every lane feeds 8 consecutive k.
//
// Common to every kernel in this file (BM x BN output tile per workgroup, BK = 64):
//   * HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane, no VGPR round trip), two LDS
//     stages; the DMA of later k-tiles is in flight while the current one is multiplied.
//   * LDS image is [rows][64] T (128-B rows) with the 16-B chunk index XOR-ed by (row & 7).  LDS-DMA
//     writes lane-linearly, so the swizzle is applied to the per-lane SOURCE address (each 8-lane
//     group still reads one full 128-B line) and again on the ds_read_b128 side: conflict-free
//     fragment reads for the 16-lane ds_read_b128 groups.
//   * The product is computed TRANSPOSED: D = W_tile * X_tile^T (mfma(bfrag, afrag)), so a lane
//     ends up with 4 consecutive output COLUMNS of one row.  The weight rows are additionally
//     permuted inside every 32-row group while staging (free: the DMA source address is per
//     lane) such that two neighbouring MFMA tiles give a lane 8 consecutive columns: 16-B bf16
//     stores / 32-B fp32 read-modify-writes in the epilogue, bias loaded as float4.
//   * Block ids are remapped XCD-first (block b runs on XCD b % 8), then GROUP_M-swizzled, so the
//     32 CUs of an XCD work on a GROUP_M x 8 patch of tiles that share A/B panels through their L2.
//
// Kernels (launch_epi picks one; slime_gemm_force_tile overrides for A/B tests):
//   gemm_db_kernel   DIRECT-B kernel (round 3), 128x256: the static operand in MFMA-fragment order (slime_gemm_pack_b) loaded
//                    straight into VGPRs, only A through LDS, two workgroups per CU -- wherever the caller supplies B_frag and
//                    the grid is not small (the tower's q/k/v, out_proj, fc1, patch embed; the adapter's and Llama's projections)
//   gemm_w4_kernel   four-wave STREAM kernel, 256x256 / 192x256: reads and DMA interleaved into one MFMA
//                    stream per SIMD, accumulators in AGPRs, one barrier per k-tile -- grids of >= 256 tiles without B_frag
//   gemm_pp_kernel   eight-wave PING-PONG kernel, 256x256 / 192x256 -- sub-round grids (they co-run best
//                    with the other tower stream's kernels)
//   gemm_kernel      lock-step kernel, 128x128 (narrow N, small M; two-stage, or a three-stage ring for grids of at most one
//                    workgroup per CU) and 256x256
//   gemm_ppp_kernel, gemm_pp32b_kernel   persistent / 32x32x16-MFMA ping-pong variants kept as measured alternatives
//
// Epilogues: bias -> T; bias+quick-GELU -> T; bias+erf-GELU -> T; bias -> fp32; fp32 += (residual) [+ T copy and LayerNorm partial
// sums]; T(acc + bias + T residual).
#include "gemm_shared.h"

// Sum over the four 16-lane rows of a wave (lanes l, l+16, l+32, l+48), result in every row: pure VALU (permlane swaps).
__device__ __forceinline__ float rows4_allsum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned w = __float_as_uint(y);
    const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Consumer side of the LayerNorm fold: (rstd, -mu * rstd) of the BM rows of this workgroup's tile -> LDS, from the producer's
// partial sums (sum x, sum x^2 per 64-column group, summed here in a FIXED order: results do not depend on the tile shape
// of either kernel).  Ordinary loads, issued before any LDS-DMA of the prologue.  var = E[x^2] - mu^2 in fp32: sound while
// |mu| is not orders of magnitude above sigma (residual streams are not; tests/test_gpu_path.py stresses x100 outliers).
template <int BM, int NT>
__device__ __forceinline__ void stage_ln_rows(const GemmArgs& g, const int m0, float* lnrow) {
    if (!g.ln_stats) return;
    for (int r = threadIdx.x; r < BM; r += NT) {
        const int row = m0 + r;
        float rstd = 0.f, nmr = 0.f;
        if (row < g.M) {
            const float2* st = reinterpret_cast<const float2*>(g.ln_stats) + (size_t)row * g.ln_groups;
            float sx = 0.f, sq = 0.f;
            if (g.ln_groups == 16) {                          // D = 1024: the row's 128 bytes as 8 independent 16-byte loads
                float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = reinterpret_cast<const float4*>(st)[i];
#pragma unroll
                for (int i = 0; i < 8; ++i) { sx += v[i].x; sq += v[i].y; sx += v[i].z; sq += v[i].w; }     // group order 0, 1, 2, ...
            } else {
                for (int i = 0; i < g.ln_groups; ++i) { const float2 v = st[i]; sx += v.x; sq += v.y; }
            }
            ln_finalize(sx, sq, g.K, g.ln_eps, rstd, nmr);
        }
        lnrow[2 * r] = rstd; lnrow[2 * r + 1] = nmr;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the prologue's barrier publishes the table to the other waves
}

// erf-form GELU 0.5 x (1 + erf(x / sqrt 2)).  The device-library erff is a large branchy routine (it
// put 1.2 KB of scratch into this epilogue and ran the projector GEMM at 114 TF/s); this is the
// branch-free Abramowitz-Stegun 7.1.26 form, |erf error| <= 1.5e-7 absolute -- two orders below the
// rounding of the 16-bit output -- with one v_rcp and one v_exp.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float tail = poly * __expf(-z * z);                       // 1 - erf(|x|/sqrt2)
    const float one_plus_erf = x >= 0.f ? 2.0f - tail : tail;       // 1 + erf(x/sqrt2)
    return 0.5f * x * one_plus_erf;
}

// The static operand staged from its FRAGMENT-ORDER image alone (round 5: B = NULL, no row-major copy of a packed weight).  One
// fragment (64-column tile t, k-step s, fragment f) is 1 KiB contiguous in the image: 64 lanes x 16 B, lane = 16 lq + c for the 16
// weight rows c of MFMA block 4 t + f and the four 8-wide k groups lq of the k-step.  An LDS-DMA piece writes 64 lanes x 16 B
// lane-linearly, so the fragment is copied AS IT IS (one fully coalesced 1 KiB read, no swizzle): the LDS image of a B tile in this
// mode is [16-row block][k-step][lane] -- a block is 2 KiB exactly as in the row-major image ([16 rows][128 B]), so the main loops'
// block offsets stay, and the fragment read of lane l for k-step ks is simply block + ks * 1024 + 16 l (64 consecutive 16-byte
// reads: conflict free).  Same bytes into the same MFMAs: results are bit-identical to the row-major path.
// frag_piece_offset: byte offset of (block blk counted from the tile's first row, k-step ks of the k-tile) inside the image,
// relative to the tile's first fragment; the k-tile advance is FRAG_KTILE_BYTES instead of 128.
constexpr unsigned FRAG_KTILE_BYTES = 2 * 4 * 1024;
__device__ __forceinline__ unsigned frag_piece_offset(const int blk, const int ks, const int K) {
    return (((unsigned)(blk >> 2) * (unsigned)(K >> 5) + (unsigned)ks) * 4u + (unsigned)(blk & 3)) * 1024u;
}

template <int EPI> struct EpiOutIsT { static constexpr bool value = (EPI <= SLIME_EPI_BIAS_GELU_T || EPI == SLIME_EPI_BIAS_RESID_T); };

// Wave-level epilogue.  A lane owns, for every 16-row step i and column-tile pair p, 8 consecutive
// columns starting at col_base + 32 p of row row_base + 16 i.
//
// What the s_memtime stamps showed (256x256 tile, 59k-cycle workgroup at K = 1024): a naive
// "for each 8-column group: load bias/residual, add, store" epilogue costs 13k cycles.  On CDNA4 vmcnt
// counts stores as well as loads, bias/residual may alias C so loads cannot be hoisted over stores,
// and a per-row `if (row < M)` makes the compiler open every block with a conservative vmcnt(0) that
// also drains the previous block's stores.  Hence:
//   * FULL tiles (every row < M; all but the last row tile) take a branch-free path;
//   * the bias is loaded once up front; results are formed in place in the accumulators (distinct
//     registers per store) and stored back to back;
//   * the fp32 residual is fetched in register double-buffered batches, one batch ahead of the stores.
// LN (consumer side of the LayerNorm fold): lnrow points at this lane's first row of the workgroup's (rstd, -mu rstd) table.
template <typename T, int EPI, int MI, int NI, bool FULL, bool LN = false>
__device__ __forceinline__ void epilogue_wave(const GemmArgs& g, f32x4 (&acc)[MI][NI], const int row_base, const int col_base,
                                              const float* lnrow = nullptr) {
    constexpr int NP = NI / 2;
    float bias[NP][8];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (g.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(g.bias + col_base + 32 * p);
            const float4 b1 = *reinterpret_cast<const float4*>(g.bias + col_base + 32 * p + 4);
            bias[p][0] = b0.x; bias[p][1] = b0.y; bias[p][2] = b0.z; bias[p][3] = b0.w;
            bias[p][4] = b1.x; bias[p][5] = b1.y; bias[p][6] = b1.z; bias[p][7] = b1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) bias[p][j] = 0.f;
        }
    }
    auto in_range = [&](int row) { return FULL || row < g.M; };
    if constexpr (LN) {
        // LayerNorm-fold epilogues (BIAS_T / BIAS_QUICKGELU_T only): pre-activation = rstd[m] * (acc - mu[m] * colsum[n]) + bias[n],
        // evaluated as fma(rstd, acc, fma(-mu rstd, colsum, bias)).  Everything that is loaded (bias, colsum, the row table) is
        // fetched up front; each 8-column group is then read from the accumulators, finished and stored at once, so only 8
        // results are live at a time (the in-place form below keeps all of them in VGPRs, which spills next to 64 constants).
        static_assert(EpiOutIsT<EPI>::value, "LayerNorm fold: T outputs only");
        float cs[NP][8];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float4 c0 = *reinterpret_cast<const float4*>(g.ln_colsum + col_base + 32 * p);
            const float4 c1 = *reinterpret_cast<const float4*>(g.ln_colsum + col_base + 32 * p + 4);
            cs[p][0] = c0.x; cs[p][1] = c0.y; cs[p][2] = c0.z; cs[p][3] = c0.w;
            cs[p][4] = c1.x; cs[p][5] = c1.y; cs[p][6] = c1.z; cs[p][7] = c1.w;
        }
        float2 rn[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) rn[i] = *reinterpret_cast<const float2*>(lnrow + 32 * i);   // rows 16 apart, 2 floats each
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = row_base + i * 16;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = fmaf(rn[i].x, acc[i][2 * p][j], fmaf(rn[i].y, cs[p][j], bias[p][j]));
                    v[4 + j] = fmaf(rn[i].x, acc[i][2 * p + 1][j], fmaf(rn[i].y, cs[p][4 + j], bias[p][4 + j]));
                }
                if constexpr (EPI == SLIME_EPI_BIAS_QUICKGELU_T) {
                    // x sigmoid(1.702 x) on PAIRS: the multiply by C, the 1 + e and the final product as v_pk_mul / v_pk_add (two
                    // elements per issue; hipcc packs the first multiply on its own but leaves the other two scalar behind the
                    // scalar v_exp / v_rcp).  Same IEEE operations per element: results unchanged (round 4).
                    constexpr float C = -1.702f * 1.4426950408889634f;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const f32x2 z = {v[j], v[j + 1]};
                        const f32x2 u = z * C;
                        f32x2 e = {__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])};
                        e = e + 1.0f;
                        const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
                        const f32x2 y = z * r;
                        v[j] = y[0]; v[j + 1] = y[1];
                    }
                }
                const u32x4 w = pack8<T>(v);
                if (in_range(row))
                    st_stream(reinterpret_cast<u32x4*>(reinterpret_cast<char*>(g.C) + ((size_t)row * g.ldc + col_base + 32 * p) * 2), w);
            }
        }
        return;
    }
    if constexpr (EPI == SLIME_EPI_BIAS_RESID_F32 || EPI == SLIME_EPI_BIAS_RESID_F32_LN) {
        static_assert(EPI != SLIME_EPI_BIAS_RESID_F32_LN || NP % 2 == 0, "BIAS_RESID_F32_LN: partial sums per whole 64-column group of a wave");
        constexpr int BATCH = (MI >= 2) ? 2 : 1;          // 16-row steps per residual batch
        constexpr int NB = MI / BATCH;
        f32x4 hb[2][BATCH][NP][2];
        float* C = reinterpret_cast<float*>(g.C);
        auto load_batch = [&](int b, int buf) {
#pragma unroll
            for (int ii = 0; ii < BATCH; ++ii) {
                int row = row_base + (b * BATCH + ii) * 16;
                if constexpr (!FULL) row = min(row, g.M - 1);                       // clamp: value unused past M
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const float* src = C + (size_t)row * g.ldc + col_base + 32 * p;
                    hb[buf][ii][p][0] = ld_stream(reinterpret_cast<const f32x4*>(src));
                    hb[buf][ii][p][1] = ld_stream(reinterpret_cast<const f32x4*>(src + 4));
                }
            }
        };
        load_batch(0, 0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b + 1 < NB) load_batch(b + 1, (b + 1) & 1);           // residual loads run one batch ahead of the stores
#pragma unroll
            for (int ii = 0; ii < BATCH; ++ii) {
                const int i = b * BATCH + ii;
                const int row = row_base + i * 16;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const f32x4 h0 = hb[b & 1][ii][p][0], h1 = hb[b & 1][ii][p][1];
                    acc[i][2 * p][0] += bias[p][0] + h0[0]; acc[i][2 * p][1] += bias[p][1] + h0[1];
                    acc[i][2 * p][2] += bias[p][2] + h0[2]; acc[i][2 * p][3] += bias[p][3] + h0[3];
                    acc[i][2 * p + 1][0] += bias[p][4] + h1[0]; acc[i][2 * p + 1][1] += bias[p][5] + h1[1];
                    acc[i][2 * p + 1][2] += bias[p][6] + h1[2]; acc[i][2 * p + 1][3] += bias[p][7] + h1[3];
                    if (in_range(row)) {
                        float* o = C + (size_t)row * g.ldc + col_base + 32 * p;
                        st_stream(reinterpret_cast<f32x4*>(o), acc[i][2 * p]);
                        st_stream(reinterpret_cast<f32x4*>(o + 4), acc[i][2 * p + 1]);
                    }
                }
                if constexpr (EPI == SLIME_EPI_BIAS_RESID_F32_LN) {
                    // the next GEMM's A operand: the updated rows rounded to T, and the partial sums of the rows per 64-column group = lane-local over the pair of 32-column blocks, then
                    // across the wave's four 16-lane rows; same order in every kernel of this file
#pragma unroll
                    for (int pp = 0; pp < NP / 2; ++pp) {
                        float sx = 0.f, sq = 0.f;
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int p = 2 * pp + h2;
                            u32x4 w;
                            w[0] = T::pack2(acc[i][2 * p][0], acc[i][2 * p][1]); w[1] = T::pack2(acc[i][2 * p][2], acc[i][2 * p][3]);
                            w[2] = T::pack2(acc[i][2 * p + 1][0], acc[i][2 * p + 1][1]); w[3] = T::pack2(acc[i][2 * p + 1][2], acc[i][2 * p + 1][3]);
                            if (in_range(row))
                                st_stream(reinterpret_cast<u32x4*>(g.x16 + ((size_t)row * g.ldx + col_base + 32 * p) * 2), w);
                            // sums of the fp32 values (the rounded ones differ by 2^-9 relative per element with random sign:
                            // far below what the statistics need, and unpacking them again would double this loop's VALU work)
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float a0 = acc[i][2 * p][k], a1 = acc[i][2 * p + 1][k];
                                sx += a0; sx += a1;
                                sq = fmaf(a0, a0, sq); sq = fmaf(a1, a1, sq);
                            }
                        }
                        sx = rows4_allsum(sx);
                        sq = rows4_allsum(sq);
                        // col_base = (first column of the wave's 64-aligned span) + 8 * (lane >> 4): lane row 0 writes
                        if ((col_base & 31) == 0 && in_range(row))
                            *reinterpret_cast<float2*>(g.stats_out + ((size_t)row * (g.N >> 6) + ((col_base + 64 * pp) >> 6)) * 2) = make_float2(sx, sq);
                    }
                }
            }
        }
    } else if constexpr (EPI == SLIME_EPI_BIAS_RESID_SPLIT_LN) {
        // The residual update on the SPLIT stream (round 5; ABI 7: the lower part is one signed byte per element, common.h resid_delta /
        // resid_join): h = join(hi, d8) (exact), c = acc + (bias + h), hi' = T(c), d8' = delta(c, hi'), partial sums of the UNROUNDED c as
        // in BIAS_RESID_F32_LN (the front end, producer of layer 0's table, sums the rounded rows: include/slime_hip.h states the
        // difference).  hi' IS the next GEMM's operand: 6 bytes per element cross the fabric here (2 + 1 in, 2 + 1 out) instead of 8
        // (ABI 5/6: two 16-bit halves) or 10 (fp32 stream + 16-bit copy); what is kept of c are 16 (bf16) / 19 (fp16) significant bits.
        // C (= hi) and lo8 are read and written in place and vmcnt counts stores: both planes are fetched one 16-row step ahead
        // (two and three steps ahead measured the same, profiles/r06_resid_lookahead_ab.txt: the epilogue is bound by bytes).
        static_assert(NP % 2 == 0, "BIAS_RESID_SPLIT_LN: a wave must own whole 64-column groups (the stores sit in the per-group loop: a 32-column "
                                   "wave tile would store nothing -- round 6, profiles/r06_ring8_tiles.txt)");
        u32x4 rh[2][NP];
        u32x2 rl[2][NP];
        char* Hi = reinterpret_cast<char*>(g.C);
        char* Lo = g.lo;
        auto load_step = [&](int i, int buf) {
            int row = row_base + i * 16;
            if constexpr (!FULL) row = min(row, g.M - 1);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                rh[buf][p] = ld_stream(reinterpret_cast<const u32x4*>(Hi + ((size_t)row * g.ldc + col_base + 32 * p) * 2));
                rl[buf][p] = ld_stream(reinterpret_cast<const u32x2*>(Lo + ((size_t)row * g.ldlo + col_base + 32 * p)));
            }
        };
        load_step(0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (i + 1 < MI) load_step(i + 1, (i + 1) & 1);
            const int row = row_base + i * 16;
#pragma unroll
            for (int pp = 0; pp < NP / 2; ++pp) {
                float sx = 0.f, sq = 0.f;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int p = 2 * pp + h2;
                    const u32x4 hw = rh[i & 1][p];
                    const u32x2 lw = rl[i & 1][p];
                    u32x4 w;
                    int nd[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {                    // word e = columns 2e, 2e+1 of the lane's 8: accumulator block 2p + (e >> 1)
                        const float h0 = resid_join<T>(T::lo(hw[e]), sext_byte(lw[e >> 1], 2 * (e & 1)));
                        const float h1 = resid_join<T>(T::hi(hw[e]), sext_byte(lw[e >> 1], 2 * (e & 1) + 1));
                        const float c0 = acc[i][2 * p + (e >> 1)][(2 * e) & 3] + (bias[p][2 * e] + h0);
                        const float c1 = acc[i][2 * p + (e >> 1)][(2 * e + 1) & 3] + (bias[p][2 * e + 1] + h1);
                        acc[i][2 * p + (e >> 1)][(2 * e) & 3] = c0;
                        acc[i][2 * p + (e >> 1)][(2 * e + 1) & 3] = c1;
                        w[e] = T::pack2(c0, c1);
                        nd[2 * e] = resid_delta<T>(c0, T::lo(w[e]));
                        nd[2 * e + 1] = resid_delta<T>(c1, T::hi(w[e]));
                    }
                    if (in_range(row)) {
                        st_stream(reinterpret_cast<u32x4*>(Hi + ((size_t)row * g.ldc + col_base + 32 * p) * 2), w);
                        st_stream(reinterpret_cast<u32x2*>(Lo + ((size_t)row * g.ldlo + col_base + 32 * p)),
                                  u32x2{pack_bytes(nd[0], nd[1], nd[2], nd[3]), pack_bytes(nd[4], nd[5], nd[6], nd[7])});
                    }
                    // partial sums in the element order of BIAS_RESID_F32_LN
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float a0 = acc[i][2 * p][k], a1 = acc[i][2 * p + 1][k];
                        sx += a0; sx += a1;
                        sq = fmaf(a0, a0, sq); sq = fmaf(a1, a1, sq);
                    }
                }
                sx = rows4_allsum(sx);
                sq = rows4_allsum(sq);
                if ((col_base & 31) == 0 && in_range(row))
                    *reinterpret_cast<float2*>(g.stats_out + ((size_t)row * (g.N >> 6) + ((col_base + 64 * pp) >> 6)) * 2) = make_float2(sx, sq);
            }
        }
    } else if constexpr (EPI == SLIME_EPI_BIAS_RESID_T) {
        // C = T(acc + bias + float(R)): the decoder layer's residual add (16-bit stream, as HF keeps it).  R may alias C, and vmcnt
        // counts stores too: the residual rows are fetched in register double-buffered batches, one batch ahead of the stores.
        constexpr int BATCH = (MI >= 2) ? 2 : 1;
        constexpr int NB = MI / BATCH;
        u32x4 rb[2][BATCH][NP];
        auto load_batch = [&](int b, int buf) {
#pragma unroll
            for (int ii = 0; ii < BATCH; ++ii) {
                int row = row_base + (b * BATCH + ii) * 16;
                if constexpr (!FULL) row = min(row, g.M - 1);
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    rb[buf][ii][p] = ld_stream(reinterpret_cast<const u32x4*>(g.resid + ((size_t)row * g.ldr + col_base + 32 * p) * 2));
            }
        };
        load_batch(0, 0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b + 1 < NB) load_batch(b + 1, (b + 1) & 1);
#pragma unroll
            for (int ii = 0; ii < BATCH; ++ii) {
                const int i = b * BATCH + ii;
                const int row = row_base + i * 16;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const u32x4 r = rb[b & 1][ii][p];
                    float v[8];
                    v[0] = acc[i][2 * p][0] + bias[p][0] + T::lo(r[0]); v[1] = acc[i][2 * p][1] + bias[p][1] + T::hi(r[0]);
                    v[2] = acc[i][2 * p][2] + bias[p][2] + T::lo(r[1]); v[3] = acc[i][2 * p][3] + bias[p][3] + T::hi(r[1]);
                    v[4] = acc[i][2 * p + 1][0] + bias[p][4] + T::lo(r[2]); v[5] = acc[i][2 * p + 1][1] + bias[p][5] + T::hi(r[2]);
                    v[6] = acc[i][2 * p + 1][2] + bias[p][6] + T::lo(r[3]); v[7] = acc[i][2 * p + 1][3] + bias[p][7] + T::hi(r[3]);
                    const u32x4 w = pack8<T>(v);
                    if (in_range(row))
                        st_stream(reinterpret_cast<u32x4*>(reinterpret_cast<char*>(g.C) + ((size_t)row * g.ldc + col_base + 32 * p) * 2), w);
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = acc[i][2 * p][j] + bias[p][j], b = acc[i][2 * p + 1][j] + bias[p][4 + j];
                    if constexpr (EPI == SLIME_EPI_BIAS_QUICKGELU_T) {          // x*sigmoid(1.702x)
                        constexpr float C = -1.702f * 1.4426950408889634f;             // e^(-1.702 x) = 2^(C x): one multiply, then v_exp
                        a = a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(C * a));
                        b = b * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(C * b));
                    } else if constexpr (EPI == SLIME_EPI_BIAS_GELU_T) {        // erf GELU (nn.GELU())
                        a = gelu_erf(a);
                        b = gelu_erf(b);
                    }
                    acc[i][2 * p][j] = a; acc[i][2 * p + 1][j] = b;
                }
        if constexpr (EpiOutIsT<EPI>::value) {
            u32x4 packed[MI][NP];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    packed[i][p][0] = T::pack2(acc[i][2 * p][0], acc[i][2 * p][1]);
                    packed[i][p][1] = T::pack2(acc[i][2 * p][2], acc[i][2 * p][3]);
                    packed[i][p][2] = T::pack2(acc[i][2 * p + 1][0], acc[i][2 * p + 1][1]);
                    packed[i][p][3] = T::pack2(acc[i][2 * p + 1][2], acc[i][2 * p + 1][3]);
                }
            // g.row_map (optional): the row is stored where its consumer wants it (the adapter's token buffer) instead of at `row`
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = row_base + i * 16;
                if (in_range(row)) {
                    const size_t orow = g.row_map ? (size_t)g.row_map[row] : (size_t)row;
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        st_stream(reinterpret_cast<u32x4*>(reinterpret_cast<char*>(g.C) + (orow * g.ldc + col_base + 32 * p) * 2), packed[i][p]);
                }
            }
        } else {
            float* C = reinterpret_cast<float*>(g.C);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = row_base + i * 16;
                if (in_range(row)) {
                    const size_t orow = g.row_map ? (size_t)g.row_map[row] : (size_t)row;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        float* o = C + orow * g.ldc + col_base + 32 * p;
                        st_stream(reinterpret_cast<f32x4*>(o), acc[i][2 * p]);
                        st_stream(reinterpret_cast<f32x4*>(o + 4), acc[i][2 * p + 1]);
                    }
                }
            }
        }
    }
}

// Epilogue BIAS_GELU_MIX_T (direct-B kernel only; round 4): the workgroup's 128-row operand tile is 64 rows of A (the token rows x)
// on top of the same 64 rows of A2 (their attention expert), so accumulator blocks i and i + 4 of a lane belong to ONE token.  Both
// get bias + erf-GELU, the gate pair of the token mixes them in fp32, and the mix is rounded to T once: C[token] = T(g0 a0 + g1 a1).
template <typename T>
__device__ __forceinline__ void epilogue_mix(const GemmArgs& g, f32x4 (&acc)[8][4], const int tok0, const int li, const int col_base) {
    float bias[2][8];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (g.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(g.bias + col_base + 32 * p);
            const float4 b1 = *reinterpret_cast<const float4*>(g.bias + col_base + 32 * p + 4);
            bias[p][0] = b0.x; bias[p][1] = b0.y; bias[p][2] = b0.z; bias[p][3] = b0.w;
            bias[p][4] = b1.x; bias[p][5] = b1.y; bias[p][6] = b1.z; bias[p][7] = b1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) bias[p][j] = 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tok = tok0 + 16 * i + li;
        const float2 gt = *reinterpret_cast<const float2*>(g.mix_gates + 2 * (size_t)min(tok, g.M - 1));
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = gt.x * gelu_erf(acc[i][2 * p][j] + bias[p][j]) + gt.y * gelu_erf(acc[i + 4][2 * p][j] + bias[p][j]);
                v[4 + j] = gt.x * gelu_erf(acc[i][2 * p + 1][j] + bias[p][4 + j]) + gt.y * gelu_erf(acc[i + 4][2 * p + 1][j] + bias[p][4 + j]);
            }
            if (tok < g.M)
                *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(g.C) + ((size_t)tok * g.ldc + col_base + 32 * p) * 2) = pack8<T>(v);
        }
    }
}

// Epilogue dispatch shared by every kernel: FULL-tile fast path or ragged last row tile; LayerNorm-fold variant when the call
// carries row statistics (only the two epilogues the tower uses it with are instantiated).
template <typename T, int EPI, int MI, int NI>
__device__ __forceinline__ void run_epilogue(const GemmArgs& g, f32x4 (&acc)[MI][NI], const int row_base, const int col_base,
                                             const bool full, const float* lnrow_lane) {
    constexpr bool CAN_LN = (EPI == SLIME_EPI_BIAS_T || EPI == SLIME_EPI_BIAS_QUICKGELU_T);
    if constexpr (CAN_LN) {
        if (g.ln_stats) {
            if (full) epilogue_wave<T, EPI, MI, NI, true, true>(g, acc, row_base, col_base, lnrow_lane);
            else epilogue_wave<T, EPI, MI, NI, false, true>(g, acc, row_base, col_base, lnrow_lane);
            return;
        }
    }
    if (full) epilogue_wave<T, EPI, MI, NI, true>(g, acc, row_base, col_base);
    else epilogue_wave<T, EPI, MI, NI, false>(g, acc, row_base, col_base);
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int SCHED>
__global__ void __launch_bounds__(WAVES_M * WAVES_N * 64)
gemm_kernel(GemmArgs g) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int BK = 64;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
    static_assert(NI % 2 == 0 && A_INSTR >= 1 && B_INSTR >= 1, "tile shape");
    constexpr int GROUP_M = 8;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- tile coordinates: XCD-first remap (bijective for any grid), then GROUP_M swizzle ------
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.N / BN;
    const int nblk = tiles_m * tiles_n;
    int pid;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nblk >> 3, r = nblk & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int in_group = GROUP_M * tiles_n;
    const int group_id = pid / in_group;
    const int first_m = group_id * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (pid % in_group) % gsz;
    const int tn = (pid % in_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // ---- LDS-DMA source pointers (per lane) ---------------------------------------------------
    const int lrow = lane >> 3;                       // row inside the 8-row piece == (LDS row & 7)
    const int lchunk = (lane & 7) ^ lrow;             // swizzled 16-B chunk of the 128-B k-slab
    const char* a_src[A_INSTR];
    const char* b_src[B_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int r = (i * NW + wave) * 8 + lrow;
        const int gm = min(m0 + r, g.M - 1);          // clamp: rows past M re-read the last row
        a_src[i] = g.A + ((size_t)gm * g.lda) * 2 + lchunk * 16;
    }
    // the static operand comes from its row-major image (k-tile advance 128 bytes) or, when the caller passed B = NULL, from the
    // fragment-order image alone (frag_piece_offset: piece q = block q >> 1, k-step q & 1, copied as it is; k-tile advance 8 KiB)
    const bool bfrag = g.B == nullptr;
    const int b_kstep = bfrag ? (int)FRAG_KTILE_BYTES : BK * 2;
    const char* b_tile0 = (bfrag ? g.Bimg : g.B) + (size_t)n0 * g.K * 2;     // BN % 64 == 0: n0 rows = n0 / 64 whole fragment tiles
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int q = i * NW + wave;                  // piece: 8 LDS rows (row-major) or one fragment of one k-step (fragment image)
        const int rho = q * 8 + lrow;                 // LDS row
        const int nl = rho & 15;
        const int nphys = (rho & ~31) + 8 * (nl >> 2) + 4 * ((rho >> 4) & 1) + (nl & 3);
        b_src[i] = b_tile0 + (bfrag ? (size_t)frag_piece_offset(q >> 1, q & 1, g.K) + lane * 16 : ((size_t)nphys * g.K) * 2 + lchunk * 16);
    }

    auto stage = [&](int s) {
        char* base = smem + s * STAGE;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(a_src[i]),
                                             LDS_PTR(base + (i * NW + wave) * 1024), 16, 0, 0);
            a_src[i] += BK * 2;
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(b_src[i]),
                                             LDS_PTR(base + A_BYTES + (i * NW + wave) * 1024), 16, 0, 0);
            b_src[i] += b_kstep;
        }
    };

    // ---- fragment read offsets (per lane, bytes inside a stage) --------------------------------
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int sw = ((ks * 4 + (lane >> 4)) ^ (lane & 7)) << 4;
        a_off[ks] = (wm * TM + (lane & 15)) * 128 + sw;
        b_off[ks] = bfrag ? A_BYTES + wn * TN * 128 + ks * 1024 + lane * 16 : A_BYTES + (wn * TN + (lane & 15)) * 128 + sw;
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // SCHED 2 (round 3): the same tile with a THREE-stage ring and the DMA of k-tile t+2 issued while k-tile t is multiplied.
    // For grids that do not give every CU a workgroup (rank shards of 1-10 crops) there is no second wave per SIMD to cover a
    // stage that has not landed, and with one tile of look-ahead (32 MFMAs per wave = 540 cycles) against an L2 / HBM latency
    // of 500-2000 cycles the two-stage loop waits in every k-tile (fc2 at 2885 rows: 63 us for 24 GF).  The DMA is inline asm
    // (SGPR base + per-lane offset) so that hipcc neither drains it at the barrier nor guards the fragment reads with vmcnt(0);
    // the barrier is the raw instruction behind a COUNTED wait (the next tile's 8 pieces stay in flight).
    constexpr int NSTAGE = SCHED == 2 ? 3 : 2;
    float* lnrow = reinterpret_cast<float*>(smem + NSTAGE * STAGE);
    stage_ln_rows<BM, NW * 64>(g, m0, lnrow);
    const int nk = g.K / BK;
    if constexpr (SCHED == 2) {
        static_assert(A_INSTR + B_INSTR == 8 || A_INSTR + B_INSTR == 4, "counted wait below: 8 (128 x 128) or 4 (64 x 64) pieces per wave and stage");
        unsigned soff[A_INSTR + B_INSTR];
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const int r = (i * NW + wave) * 8 + lrow;
            soff[i] = (unsigned)min(r, g.M - 1 - m0) * (unsigned)g.lda * 2u + lchunk * 16;
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const int rho = (i * NW + wave) * 8 + lrow, nl = rho & 15;
            const int nphys = (rho & ~31) + 8 * (nl >> 2) + 4 * ((rho >> 4) & 1) + (nl & 3);
            const int q = i * NW + wave;
            soff[A_INSTR + i] = bfrag ? frag_piece_offset(q >> 1, q & 1, g.K) + lane * 16 : (unsigned)nphys * (unsigned)g.K * 2u + lchunk * 16;
        }
        const char* a_gbase = g.A + (size_t)m0 * g.lda * 2;
        const char* b_gbase = b_tile0;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_byte_addr(smem));
        auto stage3 = [&](int tile) {
            const unsigned base = lds0 + (tile % 3) * STAGE;
            const char* ga = uniform_ptr(a_gbase + (size_t)tile * (BK * 2));
            const char* gb = uniform_ptr(b_gbase + (size_t)tile * b_kstep);
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i) lds_dma16(soff[i], ga, base + (i * NW + wave) * 1024);
#pragma unroll
            for (int i = 0; i < B_INSTR; ++i) lds_dma16(soff[A_INSTR + i], gb, base + A_BYTES + (i * NW + wave) * 1024);
        };
        stage3(0);
        if (nk > 1) stage3(1);
        for (int kt = 0; kt < nk; ++kt) {
            // k-tile kt has landed (its pieces are older than the 8 of k-tile kt+1); every wave has finished reading k-tile kt-1
            if (kt + 1 < nk) {
                if constexpr (A_INSTR + B_INSTR == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 2 < nk) stage3(kt + 2);                  // into the stage k-tile kt-1 was read from
            const char* sb = smem + (kt % 3) * STAGE;
            u32x4 bf0[NI], af0[MI], bf1[NI], af1[MI];
#pragma unroll
            for (int j = 0; j < NI; ++j) bf0[j] = *reinterpret_cast<const u32x4*>(sb + b_off[0] + j * 2048);
#pragma unroll
            for (int i = 0; i < MI; ++i) af0[i] = *reinterpret_cast<const u32x4*>(sb + a_off[0] + i * 2048);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf1[j] = *reinterpret_cast<const u32x4*>(sb + b_off[1] + j * 2048);
#pragma unroll
            for (int i = 0; i < MI; ++i) af1[i] = *reinterpret_cast<const u32x4*>(sb + a_off[1] + i * 2048);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = T::mfma16(bf0[j], af0[i], acc[i][j]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = T::mfma16(bf1[j], af1[i], acc[i][j]);
            if constexpr (MI * NI >= MI + NI) {               // (the 64 x 64 tile has 8 MFMAs for 10 fragment reads: left to the compiler)
                __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
#pragma unroll
                for (int r = 0; r < MI + NI; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * MI * NI - 2 * (MI + NI), 0);
            }
            // the fragment reads of this k-tile must have RETURNED before the wave may enter the next barrier (the DMA issued
            // behind it overwrites this stage two tiles later: the barrier after next -- but its reads are consumed by the MFMAs
            // above, so they have)
        }
    } else {
    stage(0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage((kt + 1) & 1);
        const char* sb = smem + (kt & 1) * STAGE;
        if constexpr (SCHED == 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 bf[NI], af[MI];
#pragma unroll
                for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const u32x4*>(sb + b_off[ks] + j * 2048);
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const u32x4*>(sb + a_off[ks] + i * 2048);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = T::mfma16(bf[j], af[i], acc[i][j]);
            }
        } else {
            // Software pipeline pinned with sched_group_barrier: all k-step-0 fragments first, then
            // the k-step-1 fragment reads are interleaved 1:2 with the k-step-0 MFMAs.
            u32x4 bf0[NI], af0[MI], bf1[NI], af1[MI];
#pragma unroll
            for (int j = 0; j < NI; ++j) bf0[j] = *reinterpret_cast<const u32x4*>(sb + b_off[0] + j * 2048);
#pragma unroll
            for (int i = 0; i < MI; ++i) af0[i] = *reinterpret_cast<const u32x4*>(sb + a_off[0] + i * 2048);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf1[j] = *reinterpret_cast<const u32x4*>(sb + b_off[1] + j * 2048);
#pragma unroll
            for (int i = 0; i < MI; ++i) af1[i] = *reinterpret_cast<const u32x4*>(sb + a_off[1] + i * 2048);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = T::mfma16(bf0[j], af0[i], acc[i][j]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = T::mfma16(bf1[j], af1[i], acc[i][j]);
            __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);          // DS reads of k-step 0
#pragma unroll
            for (int r = 0; r < MI + NI; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);            // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // 1 DS read of k-step 1
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MI * NI - 2 * (MI + NI), 0);
        }
    }
    }

    // ---- epilogue: lane holds, per (mi, tile pair), 8 consecutive columns of one row ------------
    run_epilogue<T, EPI, MI, NI>(g, acc, m0 + wm * TM + (lane & 15), n0 + wn * TN + 8 * (lane >> 4), m0 + BM <= g.M,
                                 lnrow + 2 * (wm * TM + (lane & 15)));
}


// ================================================================================================
// Ping-pong kernel (256 x 256 x 64 tile, 8 waves = two groups of 4, one group per A half).
//
// Every K-tile is cut in 4 phases (the four 64 x 32 quadrants of a wave's 128 x 64 output); a phase
// is an L section (ds_read the quadrant's fragments + issue 2 LDS-DMA pieces) and an M section
// (16 MFMAs), each closed by a workgroup barrier.  Group 1 runs ONE barrier behind group 0, so in
// every barrier-to-barrier slot one group is in an M section while the other is in an L section:
// each SIMD hosts one wave of either group, hence its matrix pipe always has a wave issuing MFMAs
// while the partner wave hides LDS latency, DMA issue and barrier skew.
//
// LDS-DMA schedule (slot = barrier interval; tile t phase p: group 0 L at 8t+2p, M at 8t+2p+1; group
// 1 one slot later).  A stage buffer is reused for tile t+2 region by region as soon as its last
// reader is done; a group refills its own A quarters and half of either B half:
//     section   group 0 issues             group 1 issues            (first reader)
//     L1(t)     A rows   0..63  of t+2     A rows 128..191 of t+2    L0(t+2)
//     L2(t)     B half 0, part 0 of t+2    B half 0, part 1 of t+2   L0(t+2)
//     L3(t)     B half 1, part 0 of t+2    B half 1, part 1 of t+2   L1(t+2)
//     L0(t+1)   A rows  64..127 of t+2     A rows 192..255 of t+2    L2(t+2)
// Every piece is issued >= 2 slots after the last ds_read of the bytes it overwrites has been
// waited for (lgkmcnt(0) opens each M section), and it is retired by its issuing wave exactly 4 of
// its own sections later with a COUNTED s_waitcnt vmcnt(8) (2 pieces issued per section, 8 stay
// in flight => >= 12 slots of flight), one barrier or more before its first reader.  The main
// loop never drains vmcnt to 0 until no further tile exists.
// ================================================================================================
// XCD-owned rows (round 4).  The hardware deals the workgroups of a launch to the 8 XCDs round-robin (workgroup b -> XCD b & 7), and each
// XCD has its own 4 MiB L2.  XCD x owns a contiguous range of row tiles (as even as 8 allows) and runs ALL their column tiles, so an
// activation panel crosses the fabric into ONE L2 and a weight panel once per XCD; the grid is padded to 8 x (most row tiles an XCD owns)
// x tiles_n and the surplus workgroups of the XCDs that own one row tile less exit at once.  N_FAST: the column tiles of a row tile are
// consecutive (one round = 8 row tiles x 4 column tiles for the N = 1024 GEMMs); otherwise the XCD's row tiles are consecutive.
// Measured (profiles/r04_fabric_traffic.txt, 20-crop half batch): the ping-pong kernel (fc2: 46 row tiles x 4, ONE round, so uneven
// ownership costs nothing) reads 255 -> 211 MB per launch across the fabric at unchanged time -- ON.  The direct-B kernel (fc1: 91 row
// tiles x 16, ~3 rounds) reads 199 -> 156 MB but three XCDs then own 12 row tiles against 11: +9 % work on the critical XCDs, tower
// 14.9 -> 15.2 ms (two streams) / 16.0 -> 17.1 (one) -- OFF, it keeps the tile-balanced GROUP_M walk.
#ifndef SLIME_OPT_XCD_ROWS_PP
#define SLIME_OPT_XCD_ROWS_PP 1
#endif
#ifndef SLIME_OPT_XCD_ROWS_DB
#define SLIME_OPT_XCD_ROWS_DB 0
#endif
template <bool N_FAST>
__device__ __forceinline__ bool xcd_rows_tile(const int tiles_m, const int tiles_n, int& tm, int& tn) {
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3, q = tiles_m >> 3, r = tiles_m & 7;
    const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, cnt = q + (xcd < r ? 1 : 0);
    if (j >= cnt * tiles_n) return false;
    if constexpr (N_FAST) { tm = first + j / tiles_n; tn = j % tiles_n; }
    else { tm = first + j % cnt; tn = j / cnt; }
    return true;
}
static inline int xcd_rows_grid(int tiles_m, int tiles_n) { return 8 * ((tiles_m + 7) / 8) * tiles_n; }
// BALANCED row ownership (round 5; SLIME_OPT_XCD_ROWS_DB == 2).  Whole-row-tile ownership is uneven (91 row tiles = 3 x 12 + 5 x 11) and
// on multi-round grids the launch ends with the XCDs that own one more (round 4: fabric reads -20 %, time +2 %).  Here every XCD owns
// q = tiles_m / 8 row tiles [x q, (x + 1) q) with all their column tiles -- walked row tile fastest, so the XCD's q activation panels
// stay in its L2 while one weight panel after the other streams through -- and the r = tiles_m % 8 left-over row tiles' r x tiles_n
// tiles are dealt evenly over all XCDs (column tile by column tile): work per XCD differs by at most one tile, an activation panel
// enters one L2 (a left-over one: a few), a weight panel enters each L2 once.
__device__ __forceinline__ bool xcd_rows_tile_balanced(const int tiles_m, const int tiles_n, int& tm, int& tn) {
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3, q = tiles_m >> 3, r = tiles_m & 7;
    const int own = q * tiles_n;
    if (j < own) { tm = xcd * q + j % q; tn = j / q; return true; }
    const int left = r * tiles_n, per = (left + 7) >> 3, e = (j - own) + xcd * per;
    if (j - own >= per || e >= left) return false;
    tm = 8 * q + e % r; tn = e / r;
    return true;
}
static inline int xcd_rows_grid_balanced(int tiles_m, int tiles_n) {
    return 8 * ((tiles_m >> 3) * tiles_n + (((tiles_m & 7) * tiles_n + 7) >> 3));
}

#define PP_BARRIER()                                   \
    do {                                               \
        __builtin_amdgcn_sched_barrier(0);             \
        asm volatile("s_barrier" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);             \
    } while (0)

// KTAG only splits the symbol name by contraction-length class (K >= 2048: the fc2 shape) so that
// profilers report the short-K and long-K launches of one epilogue as separate kernels.
// ABL (timing ablations only, results are wrong for ABL != 0): bit0 = no LDS-DMA in the main loop,
// bit1 = no fragment ds_reads in the main loop, bit2 = no barriers in the main loop.
template <typename T, int EPI, int KTAG, int ABL = 0, int MT = 4>
__global__ void __launch_bounds__(512) gemm_pp_kernel(GemmArgs g) {
    // MT = 16-row MFMA tiles per quadrant: 4 -> 256-row workgroup tile, 3 -> 192 rows (a group's A half is 32*MT rows,
    // a phase is 4*MT MFMAs).  The 192-row tile exists for grid quantisation: 11540 rows x N=1024 are 184 tiles of 256
    // rows (0.72 rounds of 256 CUs) but 244 tiles of 192 rows (0.95 rounds of 3/4-size workgroups).
    constexpr int BM = 64 * MT, BN = 256, BK = 64, HALF = 32 * MT, QROWS = 16 * MT;
    constexpr int A_BYTES = BM * BK * 2, STAGE = (BM + BN) * BK * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.N / BN;
#if SLIME_OPT_XCD_ROWS_PP
    int tm, tn;
    if (!xcd_rows_tile<true>(tiles_m, tiles_n, tm, tn)) return;
#else
    const int nblk = tiles_m * tiles_n;
    int pid;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nblk >> 3, r = nblk & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int GROUP_M = g.group_m > 0 ? g.group_m : 4;   // 4 x 8 tiles per XCD round (measured: 4 >= 2,8 > 16 > 46 by ~1-3 %)
    const int in_group = GROUP_M * tiles_n;
    const int group_id = pid / in_group;
    const int first_m = group_id * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (pid % in_group) % gsz;
    const int tn = (pid % in_group) / gsz;
#endif
    const int m0 = tm * BM, n0 = tn * BN;

    unsigned long long t_start = 0, t_pro = 0, t_loop = 0;
    if constexpr ((ABL & 8) != 0) t_start = __builtin_amdgcn_s_memtime();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;            // group == A half (wm)
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;
    const int li = lane & 15, lq = lane >> 4;

    // ---- LDS-DMA pieces owned by this wave: kind 0 = A quarter (rows 0..63 of the group's half),
    // kind 1 = B half 0 part, kind 2 = B half 1 part, kind 3 = A quarter (rows 64..127); 2 pieces each.
    // Sources are (wave-uniform 64-bit base) + (per-lane 32-bit byte offset) so that the DMA is the SGPR-base form of
    // global_load_lds: half the address VGPR traffic per piece and no 64-bit VALU pointer arithmetic in the L sections;
    // the k advance is a scalar add on the base.
    const char* a_base = g.A + (size_t)m0 * g.lda * 2;
    // static operand: row-major image, or (B = NULL) the fragment-order image alone (frag_piece_offset: every fragment copied as it is; k-tile advance 8 KiB, not 128 B)
    const bool bfrag = g.B == nullptr;
    const char* b_base = (bfrag ? g.Bimg : g.B) + (size_t)n0 * g.K * 2;
    const size_t b_kstep = bfrag ? FRAG_KTILE_BYTES : BK * 2;
    unsigned soff[4][2];
    int dst[4][2];                                       // byte offset inside a stage
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = wn * 2 + j;                    // 0..7 inside the group's share
#pragma unroll
        for (int qa = 0; qa < 2; ++qa) {                 // A quarters
            // a quarter has 2*MT pieces of 8 rows; at MT = 3 wave 3's two slots repeat piece 5 (same bytes, keeps the
            // per-wave DMA count -- and with it the counted vmcnt -- uniform)
            const int row = grp * HALF + qa * QROWS + min(piece, 2 * MT - 1) * 8;
            const int rl = min(row + lrow, g.M - 1 - m0);           // clamp: rows past M re-read the last row
            soff[qa ? 3 : 0][j] = (unsigned)rl * (unsigned)g.lda * 2u + lchunk * 16;
            dst[qa ? 3 : 0][j] = row * 128;
        }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {                 // B halves: LDS rows chunk*64 + hb*32 + sub*8
            const int chunk = grp * 2 + (piece >> 2), sub = piece & 3;
            const int rho = chunk * 64 + hb * 32 + sub * 8 + lrow;
            const int nl = rho & 15;
            const int nphys = (rho & ~31) + 8 * (nl >> 2) + 4 * ((rho >> 4) & 1) + (nl & 3);
            // fragment image: half hb of 64-row tile `chunk` = blocks 4 chunk + 2 hb + (sub >> 1), k-step sub & 1 -- the same four
            // 1-KiB pieces per (chunk, hb), each copied as it is
            const int blk = chunk * 4 + 2 * hb + (sub >> 1);
            soff[1 + hb][j] = bfrag ? frag_piece_offset(blk, sub & 1, g.K) + lane * 16 : (unsigned)nphys * (unsigned)g.K * 2u + lchunk * 16;
            dst[1 + hb][j] = bfrag ? A_BYTES + blk * 2048 + (sub & 1) * 1024 : A_BYTES + (chunk * 64 + hb * 32 + sub * 8) * 128;
        }
    }
    auto issue = [&](int kind, int tile) {               // 2 pieces of `kind` for k-tile `tile`
        char* base = smem + (tile & 1) * STAGE;
        const char* gb = uniform_ptr(kind == 0 || kind == 3 ? a_base + (size_t)tile * (BK * 2) : b_base + (size_t)tile * b_kstep);
        const unsigned lb = __builtin_amdgcn_readfirstlane(lds_byte_addr(base));
#pragma unroll
        for (int j = 0; j < 2; ++j) lds_dma16(soff[kind][j], gb, lb + dst[kind][j]);
    };

    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int sw = ((ks * 4 + lq) ^ (lane & 7)) << 4;
        a_off[ks] = (grp * HALF + li) * 128 + sw;
        b_off[ks] = bfrag ? A_BYTES + wn * 64 * 128 + ks * 1024 + lane * 16 : A_BYTES + (wn * 64 + li) * 128 + sw;
    }

    f32x4 acc[2 * MT][4];
#pragma unroll
    for (int i = 0; i < 2 * MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float* lnrow = reinterpret_cast<float*>(smem + 2 * STAGE);
    stage_ln_rows<BM, 512>(g, m0, lnrow);
    const int nk = g.K / BK;
    // ---- prologue: all of tile 0, and the tile-1 pieces that "earlier" sections would have issued
    issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
    if (nk > 1) {
        issue(0, 1); issue(1, 1); issue(2, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PP_BARRIER();
    if constexpr ((ABL & 8) != 0) t_pro = __builtin_amdgcn_s_memtime();
    if (grp == 1) PP_BARRIER();                          // group 1 runs one slot behind

    u32x4 af[MT][2], bf[2][2][2];
    for (int t = 0; t < nk; ++t) {
        const char* sb = smem + (t & 1) * STAGE;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int mh = (p >> 1), nh = (p == 1 || p == 2) ? 1 : 0;   // (0,0) (0,1) (1,1) (1,0)
            // ---------------- L section ----------------
            if ((p == 0) && (!(ABL & 2) || t == 0)) {
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        bf[0][nj][ks] = *reinterpret_cast<const u32x4*>(sb + b_off[ks] + (nj * 16) * 128);
            }
            if ((p == 0 || p == 2) && (!(ABL & 2) || t == 0)) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        af[mi][ks] = *reinterpret_cast<const u32x4*>(sb + a_off[ks] + (mh * QROWS + mi * 16) * 128);
            }
            if ((p == 1) && (!(ABL & 2) || t == 0)) {
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        bf[1][nj][ks] = *reinterpret_cast<const u32x4*>(sb + b_off[ks] + (32 + nj * 16) * 128);
            }
            {
                const int kind = (p == 0) ? 3 : p - 1;    // L0: A rows 64..127 (tile t+1); L1..L3: tile t+2
                const int itile = (p == 0) ? t + 1 : t + 2;
                if (itile < nk && !(ABL & 1)) {
                    issue(kind, itile);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            if (!(ABL & 4)) PP_BARRIER();
            // ---------------- M section ----------------
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj)
                        acc[mh * MT + mi][nh * 2 + nj] = T::mfma16(bf[nh][nj][ks], af[mi][ks], acc[mh * MT + mi][nh * 2 + nj]);
            __builtin_amdgcn_s_setprio(0);
            if (!(ABL & 4)) PP_BARRIER();
        }
    }
    if (grp == 0) PP_BARRIER();                          // balance group 1's extra barrier
    if constexpr ((ABL & 8) != 0) t_loop = __builtin_amdgcn_s_memtime();

    run_epilogue<T, EPI, 2 * MT, 4>(g, acc, m0 + grp * HALF + (lane & 15), n0 + wn * 64 + 8 * (lane >> 4), m0 + BM <= g.M,
                                    lnrow + 2 * (grp * HALF + (lane & 15)));
    if constexpr ((ABL & 8) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0 && g.dbg) {
            unsigned long long* d = g.dbg + (size_t)blockIdx.x * 4;
            d[0] = t_start; d[1] = t_pro; d[2] = t_loop; d[3] = __builtin_amdgcn_s_memtime();
        }
    }
}


// ================================================================================================
// Stream kernel (256 x 256 x 64 tile, FOUR waves, each owning a 128 x 128 quarter in 256 accumulator
// registers).  What tools/mfma_mem_mix.hip measured on this chip: a wave that interleaves its LDS-DMA issues and
// fragment ds_reads one at a time between back-to-back MFMAs -- no L/M phase split, no barrier in between -- runs
// at 1.10x the bare MFMA time at this kernel's ratio (4 DMA + 8 ds_read_b128 per 32 MFMAs), whereas the
// ping-pong kernel's slot structure leaves the matrix pipe idle ~40 % of its main loop (barrier pairs around
// every 16 MFMAs; an L section is as long as an M section).  So here:
//   * per-wave tile 128 x 128: 16 fragment reads per 64 MFMAs (0.25 / MFMA instead of 0.375), 16 DMA pieces per
//     k-tile per wave;
//   * one in-order instruction stream per SIMD: MFMA, MFMA, MFMA + {one ds_read for the NEXT k-step, one DMA piece
//     for the k-tile AFTER next}; fragments are register double buffered (2 x 16 x 4 VGPRs), accumulators live in
//     AGPRs;
//   * ONE workgroup barrier per k-tile (between its two k-steps): it publishes k-tile t+1 (every wave has waited for
//     its own pieces) and retires the reads of k-tile t, after which the stage is refilled with k-tile t+2.
// LDS image, swizzles, weight-row permutation and epilogue are those of the ping-pong kernel.
// ================================================================================================
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128_imm(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}

template <typename T, int EPI, int KTAG, int MI, int ABL = 0>
__global__ void __launch_bounds__(256) gemm_w4_kernel(GemmArgs g) {
    // ABL (timing ablations only, wrong results): bit0 = no LDS-DMA in the main loop, bit1 = no fragment reads in the
    // main loop, bit2 = no barrier in the main loop.  ABL bit3 (results correct): the per-k-tile s_barrier is replaced by a
    // split barrier on an LDS counter -- a wave ARRIVES 16 MFMAs before the end of k-step 0 (its tile t+1 pieces have landed,
    // its reads of tile t have returned) and only CHECKS that all four waves have arrived when k-step 1 starts.  Measured
    // 3-5 % SLOWER than s_barrier (256-row tile; 8-15 % for the 192-row tile): the hardware barrier is cheaper than one LDS
    // atomic + one polled read per k-tile.  Kept as an ablation (slime_gemm_set_ablation(8), BF16 bias epilogue only).
    // MI = 16-row MFMA tiles per wave along M: 8 -> 256 x 256 workgroup tile (256 accumulator registers per lane),
    // 6 -> 192 x 256 (192 accumulators: leaves the register allocator slack, and quantises 11540-row grids better).
    constexpr int WM = 16 * MI, BM = 2 * WM, BN = 256, BK = 64, NF = MI + 8;
    constexpr int A_BYTES = BM * BK * 2, STAGE = (BM + BN) * BK * 2;
    constexpr int A_PIECES = BM / 8 / 4, NP = A_PIECES + 8;         // DMA pieces per wave per k-tile (A, then 8 of B)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.N / BN;
    const int nblk = tiles_m * tiles_n;
    int pid;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nblk >> 3, r = nblk & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int GROUP_M = g.group_m > 0 ? g.group_m : 4;
    const int in_group = GROUP_M * tiles_n;
    const int first_m = (pid / in_group) * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (pid % in_group) % gsz;
    const int tn = (pid % in_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
    const int li = lane & 15, lq = lane >> 4;

    // ---- DMA pieces of this wave: A pieces wave, wave+4, ... (8 rows each), B pieces likewise ----
    unsigned soff[NP];
#pragma unroll
    for (int j = 0; j < A_PIECES; ++j) {
        const int row = (wave + 4 * j) * 8 + lrow;
        const int rl = min(row, g.M - 1 - m0);                       // clamp: rows past M re-read the last row
        soff[j] = (unsigned)rl * (unsigned)g.lda * 2u + lchunk * 16;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int rho = (wave + 4 * j) * 8 + lrow;                   // LDS row of the weight tile
        const int nl = rho & 15;
        const int nphys = (rho & ~31) + 8 * (nl >> 2) + 4 * ((rho >> 4) & 1) + (nl & 3);
        soff[A_PIECES + j] = (unsigned)nphys * (unsigned)g.K * 2u + lchunk * 16;
    }
    const char* a_gbase = g.A + (size_t)m0 * g.lda * 2;
    const char* b_gbase = g.B + (size_t)n0 * g.K * 2;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_byte_addr(smem));
    auto dma = [&](int j, int tile) {                                // piece j (0..NP-1) of k-tile `tile`
        const bool isA = j < A_PIECES;
        const char* gb = uniform_ptr((isA ? a_gbase : b_gbase) + (size_t)tile * (BK * 2));
        const unsigned dst = lds0 + (tile & 1) * STAGE + (isA ? (wave + 4 * j) * 1024 : A_BYTES + (wave + 4 * (j - A_PIECES)) * 1024);
        lds_dma16(soff[j], gb, dst);
    };

    // ---- fragment read bases (stage 0); the stage offset is added per k-tile ----
    int xb[2], wb[2];                                                // byte offsets inside smem
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int sw = ((ks * 4 + lq) ^ (lane & 7)) << 4;
        xb[ks] = (wm * WM + li) * 128 + sw;
        wb[ks] = A_BYTES + (wn * 128 + li) * 128 + sw;
    }

    f32x4 acc[MI][8];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float* lnrow = reinterpret_cast<float*>(smem + 2 * STAGE + 64);
    stage_ln_rows<BM, 256>(g, m0, lnrow);
    const int nk = g.K / BK;
    // ---- prologue ----
#pragma unroll
    for (int j = 0; j < NP; ++j) dma(j, 0);
    if (nk > 1) {
#pragma unroll
        for (int j = 0; j < NP; ++j) dma(j, 1);
        if constexpr (NP == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr ((ABL & 8) != 0) {
        if (tid == 0) *reinterpret_cast<volatile unsigned*>(smem + 2 * STAGE) = 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    u32x4 F[2][NF];                                                  // [buffer][0..MI-1 = X (m tiles), MI.. = W (n tiles)]
    // Fragment reads are ordinary LDS loads (the compiler tracks lgkmcnt per register and places counted waits); the
    // LDS-DMA is inline asm with a memory clobber, so it neither triggers conservative vmcnt(0) guards nor lets the
    // loads be cached across it.
    auto read_frag = [&](u32x4& dstF, int i, int xbase, int wbase) {
        dstF = *reinterpret_cast<const u32x4*>(smem + (i < MI ? xbase + i * 2048 : wbase + (i - MI) * 2048));
    };
    // One k-step: 8*MI MFMAs on `cur`; every third MFMA is followed by one fragment read into `nxt` (READS) and one DMA
    // piece of k-tile `dma_tile` (DMA).  READS / DMA are compile-time so the stream has no branches.
    constexpr bool SPLIT = (ABL & 8) != 0;
    constexpr int ARRIVE_AT = 8 * MI - 16;
    unsigned* sync_cnt = reinterpret_cast<unsigned*>(smem + 2 * STAGE);
    unsigned polled = 0;
    auto kstep = [&](u32x4 (&cur)[NF], u32x4 (&nxt)[NF], auto reads, int xbase, int wbase, auto dmas, int dma_tile, auto arrive) {
        constexpr int RS = decltype(arrive)::value ? 2 : 3;           // an arriving step issues its reads early
#pragma unroll
        for (int m = 0; m < 8 * MI; ++m) {
            const int mi = m % MI, nj = m / MI;
            T::mfma16_agpr(acc[mi][nj], cur[MI + nj], cur[mi]);
            if constexpr (decltype(reads)::value && !(ABL & 2)) {
                if (m % RS == 1 && m / RS < NF) read_frag(nxt[m / RS], m / RS, xbase, wbase);
            }
            if constexpr (decltype(arrive)::value) {
                if (m == ARRIVE_AT) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    if (lane == 0) atomicAdd(sync_cnt, 1u);
                }
                if (m == ARRIVE_AT + 8) polled = *reinterpret_cast<volatile unsigned*>(sync_cnt);
            }
            if constexpr (decltype(dmas)::value && !(ABL & 1)) {
                if (m % 3 == 2 && m / 3 < NP) dma(m / 3, dma_tile);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    auto tile_body = [&](int t, auto more, auto refill) {
        const int so = (t & 1) * STAGE, sn = ((t + 1) & 1) * STAGE;
        // k-step 0 of tile t: prefetch the k-step 1 fragments of the same stage
        kstep(F[0], F[1], Yes{}, xb[1] + so, wb[1] + so, No{}, 0, std::integral_constant<bool, SPLIT && decltype(more)::value>{});
        // publish tile t+1 / retire the reads of tile t
        if constexpr (decltype(more)::value && SPLIT) {
            const unsigned target = 4u * (unsigned)(t + 1);
            int guard = 0;
            while (__builtin_amdgcn_readfirstlane(polled) < target && ++guard < (1 << 22))
                polled = *reinterpret_cast<volatile unsigned*>(sync_cnt);
        } else if constexpr (decltype(more)::value && !(ABL & 4)) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // k-step 1 of tile t: prefetch (t+1, k-step 0), refill this stage with tile t+2
        kstep(F[1], F[0], more, xb[0] + sn, wb[0] + sn, refill, t + 2, No{});
    };

    // fragments of (tile 0, k-step 0)
#pragma unroll
    for (int i = 0; i < NF; ++i) read_frag(F[0][i], i, xb[0], wb[0]);

    int t = 0;
    for (; t + 2 < nk; ++t) tile_body(t, Yes{}, Yes{});
    if (t + 1 < nk) { tile_body(t, Yes{}, No{}); ++t; }
    tile_body(t, No{}, No{});

    // the MFMAs are inline asm, so the compiler does not know their results need the matrix pipe's write-back latency
    // before a VALU instruction may read them: pad by hand (>= 18 wait states for a 16x16x32 MFMA)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    // ... and tie every accumulator to a statement after the padding (volatile asm statements keep their order), so the
    // epilogue's v_accvgpr_read cannot be scheduled in front of it
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
    run_epilogue<T, EPI, MI, 8>(g, acc, m0 + wm * WM + li, n0 + wn * 128 + 8 * lq, m0 + BM <= g.M, lnrow + 2 * (wm * WM + li));
}


// ================================================================================================
// Direct-B kernel (round 3): 128 x 256 x 64 tile, FOUR waves side by side along N (128 x 64 per wave, 128 accumulator
// registers), TWO workgroups per CU (two waves per SIMD) -- the static operand never touches LDS.
//
// Why (tools/gemm_bdirect_probe.hip, main loops only, random operands, M = 11520): the weights are static, so slime_gemm_pack_b
// lays them out ONCE in MFMA-fragment order -- [N/64][K/32][4 fragments][64 lanes][16 B], rows permuted exactly as the other
// kernels permute them while staging -- and a wave fetches its fragments with plain global_load_dwordx4 (1 KiB, fully
// coalesced, L2 resident).  Doing only that at the stream kernel's geometry (256 x 256, one wave per SIMD) is worth 0-3 %:
// 8 LDS-DMA + 16 ds_read_b128 are traded for 16 register loads that cost the in-order stream about as much.  What the layout
// buys is the OTHER geometry: with B out of LDS a workgroup needs 32 KiB (A only) and a wave 128 accumulators + 96 fragment
// registers, so two workgroups share a CU -- and they are independent: the prologue (cold first DMA) and the epilogue (bias /
// GELU / LayerNorm-fold VALU + the store burst, 15-20 % of a K = 1024 workgroup) of one run under the MFMAs of the other,
// which the 512-register stream kernel can never do.  LDS traffic per MFMA halves (8 waves x 16 ds_read_b128 per k-tile, 32 KiB
// of DMA writes per CU); L2 -> CU traffic rises 1.5x (A 16 KiB + B 32 KiB per workgroup k-tile).  Probe: fc1 1226 -> 1339,
// qkv 1055 -> 1257 TF/s (K = 1024); at K = 4096 the two-workgroup form trails (fc2 1278 -> 1182, N = K = 4096 1524 -> 1462).
//
// Stream of one wave, per k-step (32 MFMAs, weight fragment nj outer / activation fragment mi inner):
//   * weight fragment nj of a tile's k-step ks lives in 4 VGPRs (set FB[ks]) for its 8 MFMAs and is re-requested for the same
//     k-step of the NEXT tile right behind the last of them (two rolling sets: a request is in flight for two k-steps = 64 MFMAs
//     ~ 1100 cycles.  The first version requested one k-step ahead: equal on full grids, but a workgroup that has its CU to
//     itself -- sub-round grids -- then waits for every fragment set: fc2 at 5193-6924 rows +8-10 %, tower over 17 / 24 crops
//     7.32 -> 6.87 / 9.92 -> 9.29 ms, profiles/r03_db_deep_*.txt); the loads are inline asm with "+v" destinations (hipcc would
//     otherwise drain every LDS-DMA in front of their first use) and hand-counted vmcnt waits;
//   * activation fragments are double buffered from LDS (ordinary loads, the compiler counts lgkmcnt), one read per 4 MFMAs;
//   * k-step 1 of a tile also issues the wave's 4 LDS-DMA pieces of the tile after next; ONE barrier per k-tile.
// VMEM order inside a k-step: D_j behind MFMA 8 j + 3, G_nj behind MFMA 8 nj + 7.
// Epilogues, LayerNorm fold, tile order and output layout are those of the other kernels (run_epilogue<T, EPI, 8, 4>).
// ================================================================================================
// VMEM requests younger than weight request G_nj -- issued TWO k-steps before its use -- at the moment MFMA group nj of the current
// k-step starts (mi = 16-row MFMA tiles per wave: DMA piece j of the wave's mi/2 follows MFMA mi*j + mi/2 - 1, request G_nj follows
// MFMA mi*nj + mi - 1).  ks: k-step of the tile; more: a next tile exists (both k-steps of this tile request fragments, the previous
// tile refilled a stage); refill: k-step 1 of this tile carries the DMA pieces.
constexpr int db_younger(int mi, int nj, int ks, bool more, bool refill) {
    int c = 3 - nj;                                                       // rest of k-step s-2's requests
    if (ks == 0) c += 4 + (more ? mi / 2 : 0) + (more ? nj : 0);          // k-step s-1 = k-step 1 of the previous tile: 4 requests (+ its mi/2 pieces)
    else {
        for (int j = 0; j < mi / 2; ++j) if (more && mi * j + mi / 2 - 1 > mi * nj + mi - 1) ++c;      // pieces behind G_nj in k-step s-2
        c += (more ? 4 : 0) + (more ? nj : 0);                            // k-step s-1 = k-step 0 of this tile; this k-step's requests so far
        for (int j = 0; j < mi / 2; ++j) if (refill && mi * j + mi / 2 - 1 < mi * nj) ++c;             // this k-step's pieces so far
    }
    return c;
}

// MI = 6 (round 6, tile 19): 96 x 256 tile for the K = 1024 launches whose 128-row grid quantises badly at the 20-crop half batch
// (M = 11540: q/k/v 1092 workgroups = 2.13 rounds of 512 slots -> 1452 = 2.84; out_proj 364 = 0.71 of a round -> 484 = 0.945): same
// stream, 24 MFMAs per k-step and 3 DMA pieces per wave, db_younger() counts for any MI.  Same k order per accumulator: bit-identical.
// MI = 8: 128 x 256 tile, two workgroups per CU (the product form).  MI = 4: 64 x 256 tile (64 accumulators), built into the diagnostic
// library only (tile 13): tried for grids that do not give every CU a 128-row workgroup (rank shards of 5-9 crops, single images) and
// measured SLOWER than the 128 x 128 lock-step kernel there (tower over 5 crops 3.70 -> 3.95 ms with fc2 on it, 9 crops 4.84 -> 5.07;
// profiles/r03_small_batch_latency_b.txt): a 16-MFMA k-step (270 cycles) is shorter than the latency of the weight fragments
// requested one k-step ahead, and an under-filled chip has no second wave per SIMD to cover it.
template <typename T, int EPI, int KTAG, int MI>
__global__ void __launch_bounds__(256, 2) gemm_db_kernel(GemmArgs g) {
    constexpr int NJ = 4, BM = 16 * MI, BN = 256, BK = 64, A_BYTES = BM * BK * 2, AP = MI / 2;
    static_assert(MI == 8 || MI == 6 || MI == 4, "direct-B tile heights: 128, 96 or 64 rows");
    // BIAS_GELU_MIX_T: a tile holds TROWS = 64 tokens -- their rows of A in its upper half, of A2 in its lower half (epilogue_mix)
    constexpr bool MIX = EPI == SLIME_EPI_BIAS_GELU_MIX_T;
    static_assert(!MIX || MI == 8, "the mix epilogue pairs accumulator blocks i and i + 4 of a 128-row tile");
    constexpr int TROWS = MIX ? BM / 2 : BM;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tiles_m = (g.M + TROWS - 1) / TROWS, tiles_n = g.N / BN;
#if SLIME_OPT_XCD_ROWS_DB == 2
    int tm, tn;
    if (!xcd_rows_tile_balanced(tiles_m, tiles_n, tm, tn)) return;
#elif SLIME_OPT_XCD_ROWS_DB
    int tm, tn;
    if (!xcd_rows_tile<false>(tiles_m, tiles_n, tm, tn)) return;
#else
    const int nblk = tiles_m * tiles_n;
    int pid;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nblk >> 3, r = nblk & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int GROUP_M = g.group_m > 0 ? g.group_m : 1024 / BM;
    const int in_group = GROUP_M * tiles_n;
    const int first_m = (pid / in_group) * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (pid % in_group) % gsz;
    const int tn = (pid % in_group) / gsz;
#endif
    const int m0 = tm * TROWS, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // = the wave's 64-column slice
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
    const int li = lane & 15, lq = lane >> 4;

    unsigned soff[AP];
#pragma unroll
    for (int j = 0; j < AP; ++j) {
        const int row0 = (wave + 4 * j) * 8 + lrow;                   // < BM by construction (AP = BM / 32 pieces per wave)
        const int row = MIX ? (row0 & (TROWS - 1)) : row0;            // MIX: pieces 8..15 (j >= AP / 2) hold the same tokens, read from A2
        const int rl = min(row, g.M - 1 - m0);                       // clamp: rows past M re-read the last row
        soff[j] = (unsigned)rl * (unsigned)g.lda * 2u + lchunk * 16;
    }
    const char* a_gbase = g.A + (size_t)m0 * g.lda * 2;
    const char* a_gbase2 = MIX ? g.A2 + (size_t)m0 * g.lda * 2 : a_gbase;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_byte_addr(smem));
    auto dma = [&](int j, int tile) {
        lds_dma16(soff[j], uniform_ptr((MIX && j >= AP / 2 ? a_gbase2 : a_gbase) + (size_t)tile * (BK * 2)),
                  lds0 + (tile & 1) * A_BYTES + (wave + 4 * j) * 1024);
    };
    // this wave's weight stream: 4 KiB per k-step, contiguous over k-steps
    const char* bw = uniform_ptr(g.Bf + ((size_t)(n0 / 64 + wave) * (size_t)(g.K / 32)) * 4096);
    const unsigned boff = lane * 16;
    u32x4 FB[2][NJ];                                                  // FB[0]: fragments of a tile's k-step 0, FB[1]: of its k-step 1
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < NJ; ++j) FB[ks][j] = u32x4{0u, 0u, 0u, 0u};
    auto gl = [&](auto ksc, auto jc, int kstep) {
        constexpr int j = decltype(jc)::value;
        gload16_frag<j * 1024>(FB[decltype(ksc)::value][j], boff, uniform_ptr(bw + (size_t)kstep * 4096));
    };

    int xb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) xb[ks] = li * 128 + (((ks * 4 + lq) ^ (lane & 7)) << 4);

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float* lnrow = reinterpret_cast<float*>(smem + 2 * A_BYTES);
    const int nk = g.K / BK;
#ifdef SLIME_DIAG
    if (g.db_abl & 8) __builtin_amdgcn_s_setprio(3);
#endif
    stage_ln_rows<BM, 256>(g, m0, lnrow);
    // ---- prologue: tiles 0 and 1 of A, the weight fragments of k-steps 0 and 1; drained completely (the counted waits of the
    // main loop are written for its steady state and are merely conservative on top of an empty queue) ----
#pragma unroll
    for (int j = 0; j < AP; ++j) dma(j, 0);
    if (nk > 1) {
#pragma unroll
        for (int j = 0; j < AP; ++j) dma(j, 1);
    }
    static_for<0, NJ>([&](auto jc) { gl(std::integral_constant<int, 0>{}, jc, 0); });
    static_for<0, NJ>([&](auto jc) { gl(std::integral_constant<int, 1>{}, jc, 1); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

#ifdef SLIME_DIAG
    if (g.db_abl & 8) __builtin_amdgcn_s_setprio(0);
#endif
    u32x4 FA[2][MI];
    auto read_frag = [&](u32x4& dstF, int i, int xbase) { dstF = *reinterpret_cast<const u32x4*>(smem + xbase + i * 2048); };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    // One k-step (KS = 0 / 1 of its tile) on the fragment set FB[KS]; every fragment is re-requested for the same k-step of the NEXT
    // tile right behind its last MFMA (rolling buffer, two k-steps = 64 MFMAs ~ 1100 cycles of flight).  more: a next tile exists;
    // dmas: this k-step issues the wave's DMA pieces of tile dma_tile; reads: fetch the activation fragments of the next k-step.
    auto kstep = [&](u32x4 (&cur)[MI], u32x4 (&nxt)[MI], auto ksc, auto more, auto dmas, int dma_tile, int next_kstep, auto reads, int xbase_next) {
        constexpr int KS = decltype(ksc)::value;
        constexpr bool MORE = decltype(more)::value, D = decltype(dmas)::value, R = decltype(reads)::value;
        static_for<0, NJ>([&](auto njc) {
            constexpr int nj = decltype(njc)::value;
            vm_wait_frag<db_younger(MI, nj, KS, MORE, D)>(FB[KS][nj]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = MI * nj + mi;
                T::mfma16_agpr(acc[mi][nj], FB[KS][nj], cur[mi]);
                if constexpr (R) { if (m % 4 == 1) read_frag(nxt[m / 4], m / 4, xbase_next); }
                if constexpr (D) { if (m % MI == MI / 2 - 1 && m / MI < AP) dma(m / MI, dma_tile); }
                if constexpr (MORE) { if (mi == MI - 1) gl(ksc, njc, next_kstep); }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    auto tile_body = [&](int t, auto more, auto refill) {
        const int so = (t & 1) * A_BYTES, sn = ((t + 1) & 1) * A_BYTES;
        kstep(FA[0], FA[1], std::integral_constant<int, 0>{}, more, No{}, 0, 2 * t + 2, Yes{}, xb[1] + so);
        if constexpr (decltype(more)::value) {
            // tile t+1 has landed (younger than its last piece: one request of that k-step and this k-step's four); every read of
            // tile t has returned
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        kstep(FA[1], FA[0], std::integral_constant<int, 1>{}, more, refill, t + 2, 2 * t + 3, more, xb[0] + sn);
    };
#pragma unroll
    for (int i = 0; i < MI; ++i) read_frag(FA[0][i], i, xb[0]);

    int t = 0;
    for (; t + 2 < nk; ++t) tile_body(t, Yes{}, Yes{});
    if (t + 1 < nk) { tile_body(t, Yes{}, No{}); ++t; }
    tile_body(t, No{}, No{});

    // hand-written MFMAs: pad the matrix pipe's write-back latency and tie every accumulator behind the padding (see gemm_w4_kernel)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+a"(acc[i][j]));
#ifdef SLIME_DIAG
    if (g.db_abl & 4) __builtin_amdgcn_s_setprio(3);
    if (g.db_abl & 2) return;
    if constexpr (!MIX) { if (g.db_abl & 1) { run_epilogue<T, EPI, MI, NJ>(g, acc, li, n0 + wave * 64 + 8 * lq, true, lnrow + 2 * li); return; } }
#endif
    if constexpr (MIX) epilogue_mix<T>(g, acc, m0, li, n0 + wave * 64 + 8 * lq);
    else run_epilogue<T, EPI, MI, NJ>(g, acc, m0 + li, n0 + wave * 64 + 8 * lq, m0 + BM <= g.M, lnrow + 2 * li);
}

// Static-operand layout of gemm_db_kernel: out[((t * (K/32) + s) * 4 + nj) * 64 + lane] (16-byte units) =
// B[64 t + 32 (nj >> 1) + 8 ((lane & 15) >> 2) + 4 (nj & 1) + (lane & 3)][32 s + 8 (lane >> 4) .. + 8]  -- the MFMA B-operand
// fragment (column lane & 15, k group lane >> 4) of the column tile whose rows are permuted so that a lane's results of
// fragments 2p, 2p+1 are 8 consecutive output columns (the permutation the LDS kernels apply while staging).
__global__ void __launch_bounds__(256) pack_b_frag_kernel(const u32x4* __restrict__ B, u32x4* __restrict__ out, int N, int K) {
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)N * K / 8;
    if (o >= total) return;
    const int lane = (int)(o & 63), nj = (int)((o >> 6) & 3);
    const size_t ts = o >> 8;
    const int ksteps = K / 32;
    const int s = (int)(ts % ksteps), t = (int)(ts / ksteps);
    const int r = lane & 15;
    const int n = 64 * t + 32 * (nj >> 1) + 8 * (r >> 2) + 4 * (nj & 1) + (r & 3);
    out[o] = B[((size_t)n * K + 32 * s + 8 * (lane >> 4)) / 8];
}


#ifdef SLIME_DIAG   // measured alternatives: compiled into libslime_hip_diag.so only
// ================================================================================================
// Persistent ping-pong kernel: the ping-pong kernel above, but a workgroup walks its output tiles
// (tile = blockIdx.x, + gridDim.x, ...) as ONE continuous k-tile stream.
//
// Why (s_memtime stamps, 256x256 tiles at K = 1024): a one-tile workgroup spends 2.8k cycles in its
// prologue (first DMA latency), 43k in the main loop and 9-13k in the epilogue, and because all CUs run
// in lock step the 256 x 128..256 KB of epilogue stores hit HBM as one burst while the memory system
// idles during main loops.  Here
//   * the LDS-DMA schedule never drains between tiles: "tile t+1 / t+2" of the issue table simply
//     runs into the next output tile, so its k-tiles 0/1 are already in LDS when the current tile ends;
//   * the epilogue only ISSUES its stores; they drain under the next tile's MFMAs.  vmcnt counts stores
//     too and loads/stores may retire out of order with respect to each other, so counted waits are
//     only used where nothing older than the wanted loads can be pending: the wave drains its DMA
//     (vmcnt(0)) right before the epilogue, issues the stores, skips the (unneeded) waits of the next
//     tile's first k-tile, and resumes the counted vmcnt(8) at k-tile 1 -- by then the stores have had
//     >= 4 slots plus the epilogue arithmetic to complete.
// ================================================================================================
template <typename T, int EPI, int KTAG>
__global__ void __launch_bounds__(512) gemm_ppp_kernel(GemmArgs g) {
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int A_BYTES = BM * BK * 2, STAGE = (BM + BN) * BK * 2;
    constexpr int GROUP_M = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.N / BN;
    const int ntiles = tiles_m * tiles_n;
    const int nk = g.K / BK;                                 // >= 2 (checked by the launcher)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;
    const int li = lane & 15, lq = lane >> 4;

    // logical tile index -> (m0, n0): XCD-first remap (L & 7 is the XCD for every tile of this workgroup
    // because gridDim.x is a multiple of 8 or equals ntiles), then GROUP_M swizzle.
    auto tile_origin = [&](int L, int& m0, int& n0) {
        const int xcd = L & 7, q = ntiles >> 3, r = ntiles & 7;
        const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        const int in_group = GROUP_M * tiles_n;
        const int first_m = (pid / in_group) * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        m0 = (first_m + (pid % in_group) % gsz) * BM;
        n0 = ((pid % in_group) / gsz) * BN;
    };

    // ---- DMA pieces of this wave (see gemm_pp_kernel): kinds 0/3 = A quarters, 1/2 = B half parts ----
    int dst[4][2];
    unsigned b_voff[2][2];                                   // tile independent
    int a_row[2][2];                                         // row inside the tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = wn * 2 + j;
#pragma unroll
        for (int qa = 0; qa < 2; ++qa) {
            a_row[qa][j] = grp * 128 + qa * 64 + piece * 8 + lrow;
            dst[qa ? 3 : 0][j] = (grp * 128 + qa * 64 + piece * 8) * 128;
        }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const int chunk = grp * 2 + (piece >> 2), sub = piece & 3;
            const int rho = chunk * 64 + hb * 32 + sub * 8 + lrow;
            const int nl = rho & 15;
            const int nphys = (rho & ~31) + 8 * (nl >> 2) + 4 * ((rho >> 4) & 1) + (nl & 3);
            b_voff[hb][j] = (unsigned)nphys * (unsigned)g.K * 2u + lchunk * 16;
            dst[1 + hb][j] = A_BYTES + (chunk * 64 + hb * 32 + sub * 8) * 128;
        }
    }
    struct TileSrc { unsigned a_voff[2][2]; size_t b_base; int m0, n0; };
    auto make_src = [&](int L, TileSrc& ts) {
        tile_origin(L, ts.m0, ts.n0);
#pragma unroll
        for (int qa = 0; qa < 2; ++qa)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                ts.a_voff[qa][j] = (unsigned)min(ts.m0 + a_row[qa][j], g.M - 1) * (unsigned)g.lda * 2u + lchunk * 16;
        ts.b_base = (size_t)ts.n0 * g.K * 2;
    };
    // 2 pieces of `kind` for k-tile kt of tile `ts`, into stage buffer `buf`
    auto issue = [&](const TileSrc& ts, int kind, int kt, int buf) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const char* src = (kind == 0 || kind == 3)
                                  ? g.A + (size_t)ts.a_voff[kind == 3][j] + (size_t)kt * (BK * 2)
                                  : g.B + ts.b_base + (size_t)b_voff[kind - 1][j] + (size_t)kt * (BK * 2);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(base + dst[kind][j]), 16, 0, 0);
        }
    };

    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int sw = ((ks * 4 + lq) ^ (lane & 7)) << 4;
        a_off[ks] = (grp * 128 + li) * 128 + sw;
        b_off[ks] = A_BYTES + (wn * 64 + li) * 128 + sw;
    }

    TileSrc cur, nxt;
    int L = blockIdx.x;
    make_src(L, cur);
    // prologue of the first tile: all of k-tile 0 and the k-tile-1 pieces of kinds 0..2
    issue(cur, 0, 0, 0); issue(cur, 1, 0, 0); issue(cur, 2, 0, 0); issue(cur, 3, 0, 0);
    issue(cur, 0, 1, 1); issue(cur, 1, 1, 1); issue(cur, 2, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    int gk = 0;                                              // global k-tile counter of tile start (buffer parity)
    u32x4 af[4][2], bf[2][2][2];
    while (true) {
        const int Ln = L + gridDim.x;
        const bool has_next = Ln < ntiles;
        if (has_next) make_src(Ln, nxt);
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        PP_BARRIER();                                        // k-tiles 0/1 of this tile visible to all waves
        if (grp == 1) PP_BARRIER();                          // group 1 runs one slot behind
        for (int t = 0; t < nk; ++t) {
            const char* sb = smem + ((gk + t) & 1) * STAGE;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int mh = (p >> 1), nh = (p == 1 || p == 2) ? 1 : 0;
                if (p == 0) {
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
                            bf[0][nj][ks] = *reinterpret_cast<const u32x4*>(sb + b_off[ks] + (nj * 16) * 128);
                }
                if (p == 0 || p == 2) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
                            af[mi][ks] = *reinterpret_cast<const u32x4*>(sb + a_off[ks] + (mh * 64 + mi * 16) * 128);
                }
                if (p == 1) {
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
                            bf[1][nj][ks] = *reinterpret_cast<const u32x4*>(sb + b_off[ks] + (32 + nj * 16) * 128);
                }
                {
                    const int kind = (p == 0) ? 3 : p - 1;
                    const int itile = (p == 0) ? t + 1 : t + 2;
                    const int buf = (gk + itile) & 1;
                    bool issued = true;
                    if (itile < nk) issue(cur, kind, itile, buf);
                    else if (has_next) issue(nxt, kind, itile - nk, buf);
                    else issued = false;
                    if (!issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (t > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    // t == 0: every k-tile-0/1 piece this tile reads before k-tile 1 was drained before the
                    // previous epilogue (or in the prologue); stores may still be in flight -> no counted wait.
                }
                PP_BARRIER();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int nj = 0; nj < 2; ++nj)
                            acc[mh * 4 + mi][nh * 2 + nj] = T::mfma16(bf[nh][nj][ks], af[mi][ks], acc[mh * 4 + mi][nh * 2 + nj]);
                __builtin_amdgcn_s_setprio(0);
                PP_BARRIER();
            }
        }
        if (grp == 0) PP_BARRIER();                          // balance group 1's extra barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA for the next tile has landed
        if (cur.m0 + BM <= g.M) epilogue_wave<T, EPI, 8, 4, true>(g, acc, cur.m0 + grp * 128 + (lane & 15), cur.n0 + wn * 64 + 8 * (lane >> 4));
        else epilogue_wave<T, EPI, 8, 4, false>(g, acc, cur.m0 + grp * 128 + (lane & 15), cur.n0 + wn * 64 + 8 * (lane >> 4));   // (no LayerNorm fold in this variant)
        if (!has_next) break;
        cur = nxt;
        L = Ln;
        gk += nk;
    }
}

template <typename T, int EPI, int KTAG>
static int launch_ppp_k(const GemmArgs& g, hipStream_t stream) {
    constexpr int LDS = 2 * (256 + 256) * 64 * 2;
    auto kern = gemm_ppp_kernel<T, EPI, KTAG>;
    SLIME_SET_LDS_ONCE(kern, LDS, "gemm_ppp");
    const int g_num_cu = num_cus();
    const int ntiles = ((g.M + 255) / 256) * (g.N / 256);
    int grid = ntiles < g_num_cu ? ntiles : (g_num_cu / 8) * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, g);
    SLIME_CHECK_LAUNCH("gemm_ppp");
    return SLIME_OK;
}
template <typename T, int EPI>
static int launch_ppp(const GemmArgs& g, hipStream_t stream) {
    return g.K >= 2048 ? launch_ppp_k<T, EPI, 1>(g, stream) : launch_ppp_k<T, EPI, 0>(g, stream);
}


// ================================================================================================
// 32x32x16 MFMA variant (measured alternative, not the default): 1024 vs ~915 flop/cycle/SIMD of issue
// bound, half the matrix instructions and operand reads per flop.  A first version with the 4-phase table
// of gemm_pp_kernel (8 MFMAs on 2 accumulators per M section) was bound by the 64-cycle dependent latency
// (-15 %); the 2-phase kernel below removes that and still trails the 16x16 kernel by ~10 % (1075 vs
// 1230 TF/s at K = 4096): its L sections (16 fragment reads drained before the barrier) are long.
//   * fragments: a lane feeds row (lane & 31), k = 8*(lane >> 5) .. +7 of a 16-deep k-step
//     -> ds_read_b128 of chunk 2*ks + (lane >> 5); the 16-lane ds_read_b128 groups then touch 16
//     different rows with one chunk index, so the swizzle key is (row >> 1) & 7 (8 distinct keys per row
//     parity inside every group) instead of row & 7;
//   * D layout: col = lane & 31 (the X row m), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (the W row n);
//     W rows are permuted inside each 32-row tile while staging so that D row index i maps to
//     n = 16*(lane>>5) + r: a lane owns 16 consecutive output columns of one row.
// ================================================================================================
template <typename T, int EPI, bool FULL>
__device__ __forceinline__ void epilogue_wave32(const GemmArgs& g, f32x16 (&acc)[4][2], const int row_base, const int col_base) {
    // row_base: m0 + grp*128 + (lane & 31); col_base: n0 + wn*64 + 16*(lane >> 5); tile (mi, nj) adds (32 mi, 32 nj)
    float bias[2][16];
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.bias) bv = *reinterpret_cast<const float4*>(g.bias + col_base + 32 * nj + 4 * v4);
            bias[nj][4 * v4 + 0] = bv.x; bias[nj][4 * v4 + 1] = bv.y; bias[nj][4 * v4 + 2] = bv.z; bias[nj][4 * v4 + 3] = bv.w;
        }
    auto in_range = [&](int row) { return FULL || row < g.M; };
    if constexpr (EPI == SLIME_EPI_BIAS_RESID_F32) {
        float* C = reinterpret_cast<float*>(g.C);
        float4 hb[2][2][4];                                   // [buffer][nj][v4]: one 32-row step ahead
        auto load_step = [&](int mi, int buf) {
            int row = row_base + mi * 32;
            if constexpr (!FULL) row = min(row, g.M - 1);
#pragma unroll
            for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4)
                    hb[buf][nj][v4] = *reinterpret_cast<const float4*>(C + (size_t)row * g.ldc + col_base + 32 * nj + 4 * v4);
        };
        load_step(0, 0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            if (mi + 1 < 4) load_step(mi + 1, (mi + 1) & 1);
            const int row = row_base + mi * 32;
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    const float4 hv = hb[mi & 1][nj][v4];
                    acc[mi][nj][4 * v4 + 0] += bias[nj][4 * v4 + 0] + hv.x; acc[mi][nj][4 * v4 + 1] += bias[nj][4 * v4 + 1] + hv.y;
                    acc[mi][nj][4 * v4 + 2] += bias[nj][4 * v4 + 2] + hv.z; acc[mi][nj][4 * v4 + 3] += bias[nj][4 * v4 + 3] + hv.w;
                }
                if (in_range(row)) {
                    float* o = C + (size_t)row * g.ldc + col_base + 32 * nj;
#pragma unroll
                    for (int v4 = 0; v4 < 4; ++v4)
                        *reinterpret_cast<float4*>(o + 4 * v4) = make_float4(acc[mi][nj][4 * v4], acc[mi][nj][4 * v4 + 1], acc[mi][nj][4 * v4 + 2], acc[mi][nj][4 * v4 + 3]);
                }
            }
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float a = acc[mi][nj][r] + bias[nj][r];
                    if constexpr (EPI == SLIME_EPI_BIAS_QUICKGELU_T) a = a * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * a));
                    else if constexpr (EPI == SLIME_EPI_BIAS_GELU_T) a = gelu_erf(a);
                    acc[mi][nj][r] = a;
                }
        if constexpr (EpiOutIsT<EPI>::value) {
            u32x4 packed[4][2][2];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                    for (int hv = 0; hv < 2; ++hv)
#pragma unroll
                        for (int w = 0; w < 4; ++w) packed[mi][nj][hv][w] = T::pack2(acc[mi][nj][8 * hv + 2 * w], acc[mi][nj][8 * hv + 2 * w + 1]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int row = row_base + mi * 32;
                if (in_range(row)) {
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj) {
                        char* o = reinterpret_cast<char*>(g.C) + ((size_t)row * g.ldc + col_base + 32 * nj) * 2;
                        *reinterpret_cast<u32x4*>(o) = packed[mi][nj][0];
                        *reinterpret_cast<u32x4*>(o + 16) = packed[mi][nj][1];
                    }
                }
            }
        } else {
            float* C = reinterpret_cast<float*>(g.C);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int row = row_base + mi * 32;
                if (in_range(row)) {
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj) {
                        float* o = C + (size_t)row * g.ldc + col_base + 32 * nj;
#pragma unroll
                        for (int v4 = 0; v4 < 4; ++v4)
                            *reinterpret_cast<float4*>(o + 4 * v4) = make_float4(acc[mi][nj][4 * v4], acc[mi][nj][4 * v4 + 1], acc[mi][nj][4 * v4 + 2], acc[mi][nj][4 * v4 + 3]);
                    }
                }
            }
        }
    }
}

// ================================================================================================
// gemm_pp32b_kernel: 32x32x16 MFMA, TWO phases per k-tile.
// The 4-phase 32x32 kernel above accumulates 8 MFMAs on 2 accumulators per M section and is bound by the
// 64-cycle dependent latency of v_mfma_f32_32x32x16 (measured 15 % slower than the 16x16 kernel).  Here a
// phase is a 64x64 half of the wave tile: 4 independent accumulators x 4 k-steps = 16 MFMAs (~512 cycles),
// so the matrix pipe streams at its issue rate and there are half as many barriers per k-tile.
//
// Slots: tile t phase p -- group 0: L at 4t+2p, M at 4t+2p+1; group 1 one slot later.  Every L section
// ends with lgkmcnt(0) BEFORE its barrier, so a region is free for refill one slot after its last reader's
// L section.  Readers: A(own half, rows 0..63) and all of B in L0, A(rows 64..127) in L1.  Refill of the
// stage buffer for tile t+2 (4 pieces per wave per section):
//     L1(t)   : this group's half of the B tile of t+2      (B last read by group 1 in slot 4t+1)
//     L0(t+1) : this group's 128 A rows of t+2              (last read in L1(t), slots 4t+2 / 4t+3)
// and every L1 section retires all but its own 4 newest pieces (vmcnt(4)) one barrier or more before
// their first reader (L0(t+2)); a piece is in flight for >= 2 slots (~1100 cycles).
// ================================================================================================
template <typename T, int EPI, int KTAG>
__global__ void __launch_bounds__(512) gemm_pp32b_kernel(GemmArgs g) {
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int A_BYTES = BM * BK * 2, STAGE = (BM + BN) * BK * 2;
    constexpr int GROUP_M = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.N / BN;
    const int nblk = tiles_m * tiles_n;
    int pid;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nblk >> 3, r = nblk & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int in_group = GROUP_M * tiles_n;
    const int first_m = (pid / in_group) * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (pid % in_group) % gsz;
    const int tn = (pid % in_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int lrow = lane >> 3, cpos = lane & 7;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- LDS-DMA pieces: kind 0 = 4 pieces of this group's A half (16 pieces, 4 per wave),
    //                      kind 1 = 4 pieces of this group's half of the B tile (rows grp*128 .. +127) ----
    const char* src[2][4];
    int dst[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wn * 4 + j;                                   // 0..15: 8-row piece inside the 128-row half
        {
            const int row = grp * 128 + piece * 8;
            const int key = ((row + lrow) >> 1) & 7;
            const int gm = min(m0 + row + lrow, g.M - 1);
            src[0][j] = g.A + ((size_t)gm * g.lda) * 2 + ((cpos ^ key) << 4);
            dst[0][j] = row * 128;
        }
        {
            const int rho = grp * 128 + piece * 8 + lrow;               // LDS row of the B tile
            const int key = (rho >> 1) & 7;
            const int r32 = rho & 31;
            const int nphys = (rho & ~31) + 16 * ((r32 >> 2) & 1) + 4 * (r32 >> 3) + (r32 & 3);
            src[1][j] = g.B + ((size_t)(n0 + nphys) * g.K) * 2 + ((cpos ^ key) << 4);
            dst[1][j] = A_BYTES + (grp * 128 + piece * 8) * 128;
        }
    }
    auto issue = [&](int kind, int tile) {
        char* base = smem + (tile & 1) * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src[kind][j]), LDS_PTR(base + dst[kind][j]), 16, 0, 0);
            src[kind][j] += BK * 2;
        }
    };

    int a_off[4], b_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int sw = ((ks * 2 + lh) ^ ((lane >> 1) & 7)) << 4;
        a_off[ks] = (grp * 128 + l31) * 128 + sw;
        b_off[ks] = A_BYTES + (wn * 64 + l31) * 128 + sw;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = g.K / BK;
    issue(0, 0); issue(1, 0);                                 // all of tile 0
    if (nk > 1) {
        issue(1, 1);                                          // B of tile 1 (an "L1(-1)" piece set); A(1) comes in L0(0)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PP_BARRIER();
    if (grp == 1) PP_BARRIER();

    u32x4 af[2][4], bf[2][4];
    for (int t = 0; t < nk; ++t) {
        const char* sb = smem + (t & 1) * STAGE;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            // ---------------- L section ----------------
            if (p == 0) {
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) bf[nj][ks] = *reinterpret_cast<const u32x4*>(sb + b_off[ks] + nj * 32 * 128);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    af[mi][ks] = *reinterpret_cast<const u32x4*>(sb + a_off[ks] + (p * 64 + mi * 32) * 128);
            if (p == 0) {                                     // L0(t): this group's A rows of tile t+1
                if (t + 1 < nk) issue(0, t + 1);
            } else {                                          // L1(t): this group's B half of tile t+2; retire the rest
                if (t + 2 < nk) { issue(1, t + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // the fragment reads must have LEFT the LDS before the other group may refill what they read
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]),
                                                   "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]));
            if (p == 0)
                asm volatile("" : "+v"(bf[0][0]), "+v"(bf[0][1]), "+v"(bf[0][2]), "+v"(bf[0][3]),
                                  "+v"(bf[1][0]), "+v"(bf[1][1]), "+v"(bf[1][2]), "+v"(bf[1][3]));
            PP_BARRIER();
            // ---------------- M section: 64 x 64 half, 4 accumulators x 4 k-steps ----------------
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj)
                        acc[p * 2 + mi][nj] = T::mfma32(bf[nj][ks], af[mi][ks], acc[p * 2 + mi][nj]);
            __builtin_amdgcn_s_setprio(0);
            PP_BARRIER();
        }
    }
    if (grp == 0) PP_BARRIER();

    if (m0 + BM <= g.M) epilogue_wave32<T, EPI, true>(g, acc, m0 + grp * 128 + l31, n0 + wn * 64 + 16 * lh);
    else epilogue_wave32<T, EPI, false>(g, acc, m0 + grp * 128 + l31, n0 + wn * 64 + 16 * lh);
}

template <typename T, int EPI, int KTAG>
static int launch_pp32b_k(const GemmArgs& g, hipStream_t stream) {
    constexpr int LDS = 2 * (256 + 256) * 64 * 2;
    auto kern = gemm_pp32b_kernel<T, EPI, KTAG>;
    SLIME_SET_LDS_ONCE(kern, LDS, "gemm_pp32b");
    const int tiles_m = (g.M + 255) / 256, tiles_n = g.N / 256;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), LDS, stream, g);
    SLIME_CHECK_LAUNCH("gemm_pp32b");
    return SLIME_OK;
}
template <typename T, int EPI>
static int launch_pp32b(const GemmArgs& g, hipStream_t stream) {
    return g.K >= 2048 ? launch_pp32b_k<T, EPI, 1>(g, stream) : launch_pp32b_k<T, EPI, 0>(g, stream);
}

#endif  // SLIME_DIAG (persistent / 32x32x16 variants)

// Process-global tuning / ablation hooks exist in the DIAGNOSTIC build only (libslime_hip_diag.so, -DSLIME_DIAG: tools/ and
// the tile-forcing tests).  The product library has no mutable global state: dispatch is a pure function of the shape.
#ifdef SLIME_DIAG
static int g_ablation = 0;
static int g_group_m = 0;
static unsigned long long* g_dbg = nullptr;
extern "C" void slime_gemm_set_debug(void* p) { g_dbg = (unsigned long long*)p; }
extern "C" void slime_gemm_set_ablation(int a) { g_ablation = a; }
extern "C" void slime_gemm_set_group_m(int s) { g_group_m = s; }
static int g_db_abl = 0;
extern "C" void slime_gemm_set_db_ablation(int a) { g_db_abl = a; }
#else
static constexpr int g_db_abl = 0;
static constexpr int g_group_m = 0;
static constexpr unsigned long long* g_dbg = nullptr;
#endif

template <typename T, int EPI, int KTAG, int ABL, int MT = 4>
static int launch_pp_k(const GemmArgs& g, hipStream_t stream) {
    constexpr int BM = 64 * MT;
    constexpr int LDS = 2 * (BM + 256) * 64 * 2 + BM * 8;           // + the (rstd, -mu rstd) table of the LayerNorm fold
    auto kern = gemm_pp_kernel<T, EPI, KTAG, ABL, MT>;
    SLIME_SET_LDS_ONCE(kern, LDS, "gemm_pp");
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.N / 256;
#if SLIME_OPT_XCD_ROWS_PP
    const int grid = xcd_rows_grid(tiles_m, tiles_n);
#else
    const int grid = tiles_m * tiles_n;
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, g);
    SLIME_CHECK_LAUNCH("gemm_pp");
    return SLIME_OK;
}

template <typename T, int EPI>
static int launch_pp(const GemmArgs& g, hipStream_t stream) {
#ifdef SLIME_DIAG
    if constexpr (EPI == SLIME_EPI_BIAS_T && T::id == SLIME_BF16) {     // ablation builds: one epilogue only
        switch (g_ablation) {
            case 1: return launch_pp_k<T, EPI, 0, 1>(g, stream);
            case 2: return launch_pp_k<T, EPI, 0, 2>(g, stream);
            case 3: return launch_pp_k<T, EPI, 0, 3>(g, stream);
            case 4: return launch_pp_k<T, EPI, 0, 4>(g, stream);
            case 7: return launch_pp_k<T, EPI, 0, 7>(g, stream);
            case 8: return launch_pp_k<T, EPI, 0, 8>(g, stream);
            default: break;
        }
    }
#endif
    return g.K >= 2048 ? launch_pp_k<T, EPI, 1, 0>(g, stream) : launch_pp_k<T, EPI, 0, 0>(g, stream);
}

template <typename T, int EPI, int KTAG, int MI, int ABL = 0>
static int launch_w4_k(const GemmArgs& g, hipStream_t stream) {
    constexpr int BM = 32 * MI;
    constexpr int LDS = 2 * (BM + 256) * 64 * 2 + 64 + BM * 8;      // + the split-barrier counter + the LayerNorm-fold row table
    auto kern = gemm_w4_kernel<T, EPI, KTAG, MI, ABL>;
    SLIME_SET_LDS_ONCE(kern, LDS, "gemm_w4");
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.N / 256;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), LDS, stream, g);
    SLIME_CHECK_LAUNCH("gemm_w4");
    return SLIME_OK;
}

template <typename T, int EPI, int MI>
static int launch_w4(const GemmArgs& g, hipStream_t stream) {
#ifdef SLIME_DIAG
    if constexpr (EPI == SLIME_EPI_BIAS_T && T::id == SLIME_BF16 && MI == 8) {     // ablation builds: one configuration only
        switch (g_ablation) {
            case 1: return launch_w4_k<T, EPI, 0, MI, 1>(g, stream);
            case 2: return launch_w4_k<T, EPI, 0, MI, 2>(g, stream);
            case 3: return launch_w4_k<T, EPI, 0, MI, 3>(g, stream);
            case 4: return launch_w4_k<T, EPI, 0, MI, 4>(g, stream);
            case 7: return launch_w4_k<T, EPI, 0, MI, 7>(g, stream);
            case 8: return launch_w4_k<T, EPI, 0, MI, 8>(g, stream);   // split barrier on an LDS counter (correct results)
            default: break;
        }
    }
#endif
    return g.K >= 2048 ? launch_w4_k<T, EPI, 1, MI>(g, stream) : launch_w4_k<T, EPI, 0, MI>(g, stream);
}

template <typename T, int EPI, int KTAG, int MI>
static int launch_db_k(const GemmArgs& g, hipStream_t stream) {
    constexpr int BM = 16 * MI;
    constexpr int LDS = 2 * BM * 64 * 2 + BM * 8;                   // two A stages + the LayerNorm-fold row table
    auto kern = gemm_db_kernel<T, EPI, KTAG, MI>;
    constexpr int TROWS = EPI == SLIME_EPI_BIAS_GELU_MIX_T ? BM / 2 : BM;       // tokens per tile (gemm_db_kernel)
    const int tiles_m = (g.M + TROWS - 1) / TROWS, tiles_n = g.N / 256;
#if SLIME_OPT_XCD_ROWS_DB == 2
    const int grid = xcd_rows_grid_balanced(tiles_m, tiles_n);
#elif SLIME_OPT_XCD_ROWS_DB
    const int grid = xcd_rows_grid(tiles_m, tiles_n);
#else
    const int grid = tiles_m * tiles_n;
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, stream, g);
    SLIME_CHECK_LAUNCH("gemm_db");
    return SLIME_OK;
}
template <typename T, int EPI, int MI>
static int launch_db(const GemmArgs& g, hipStream_t stream) {
    return g.K >= 2048 ? launch_db_k<T, EPI, 1, MI>(g, stream) : launch_db_k<T, EPI, 0, MI>(g, stream);
}

#ifdef SLIME_DIAG
// Round 4's persistent direct-B kernels with the epilogue in the next tile's MFMA stream (gemm_ps.hip: gemm_ps32.inc, gemm_ps.inc):
// bit-identical to the kernels here, measured SLOWER than gemm_db_kernel (profiles/r04_ps_ablation.txt) -- measured alternatives,
// tiles 16 / 17, in their own translation unit of the diagnostic library.
bool slime_diag_ps_usable(const GemmArgs& g, int epi);
int slime_diag_launch_ps(const GemmArgs& g, int dtype, int epi, int tile, hipStream_t stream);
#endif

template <typename T, int EPI>
static int launch_pp192(const GemmArgs& g, hipStream_t stream) {
    return g.K >= 2048 ? launch_pp_k<T, EPI, 1, 0, 3>(g, stream) : launch_pp_k<T, EPI, 0, 0, 3>(g, stream);
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int SCHED>
static int launch_cfg(const GemmArgs& g, hipStream_t stream) {
    constexpr int STAGE = (BM + BN) * 64 * 2;
    constexpr int LDS = (SCHED == 2 ? 3 : 2) * STAGE + BM * 8;      // + the LayerNorm-fold row table
    auto kern = gemm_kernel<T, BM, BN, WAVES_M, WAVES_N, EPI, SCHED>;
    SLIME_SET_LDS_ONCE(kern, LDS, "gemm");
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.N / BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(WAVES_M * WAVES_N * 64), LDS, stream, g);
    SLIME_CHECK_LAUNCH("gemm");
    return SLIME_OK;
}

// Tile choice (a pure function of the shape and of the device's CU count).  The 256-row stream / ping-pong kernels are the
// throughput kernels; 128x128 (4 waves, 64 KiB LDS, 2 WG/CU) covers narrow N (tiny geometries) and small M.  Partial last
// rounds of workgroups are filled by running two half batches on two streams (see HipCLIPVisionModel.encode), not by
// shrinking the tile.  Tile ids: 1 = 256x256 lock-step, 3 = 128x128, 4 = 256x256 ping-pong, 9 = 192x256 ping-pong, 10 / 11 =
// 192x256 / 256x256 four-wave stream kernel, 12 = 128x256 direct-B kernel (needs Bf), 15 = 128x128 with a three-stage ring; diagnostic build only: 5 = persistent ping-pong, 7 = 2-phase 32x32x16 ping-pong.
#ifndef SLIME_OPT_TILE64
#define SLIME_OPT_TILE64 1
#endif
#ifndef SLIME_OPT_PP192
#define SLIME_OPT_PP192 0
#endif
#ifndef SLIME_OPT_DB96
#define SLIME_OPT_DB96 0      // measured alternative (round 6): see auto_tile -- below the 1 % kill rule on the second lease, compiled out
#endif
static int auto_tile(const GemmArgs& g) {
    int tile = (g.N % 256 == 0 && g.M >= 512) ? 4 : 3;               // ping-pong 256x256, else 128x128
    if (tile == 4) {
        // Row-tile height by grid quantisation, for grids of at least one full round of CUs: rounds x per-
        // workgroup time (~ rows + a fixed share).  qkv at 11540 rows: 552 tiles of 256 rows = 3 rounds, 732 tiles
        // of 192 rows = 3 rounds of 3/4-size workgroups (772 -> 866 TF/s).  Sub-round grids (out_proj / fc2: 184
        // tiles) would gain even more stand-alone (994 -> 1216 TF/s) but LOSE 2.5 % inside the two-stream tower,
        // where the idle CUs of a 0.72-round launch are taken by the other stream's kernels: they keep 256 rows.
        const long cus = num_cus();
        const long tn = g.N / 256;
        const long n256 = ((g.M + 255) / 256) * tn, n192 = ((g.M + 191) / 192) * tn;
        const long c256 = (n256 + cus - 1) / cus * (256 + 40);
        const long c192 = (n192 + cus - 1) / cus * (192 + 40);
        // ... and multi-round grids run the four-wave stream kernel (same-box A/B inside the two-stream tower: fc1 and
        // qkv on the stream kernel 16.45 -> 15.72 ms; out_proj / fc2 on it lose 0.1-0.3 ms, they stay ping-pong)
        if (n256 >= cus) tile = c192 < c256 ? 10 : 11;
        // ... and grids that would leave more than half of the CUs without a 256-row workgroup (single images, the adapter's
        // 4608-row projections) take the 128x128 tile: 4x the workgroups, two per CU (M = 4608, N = K = 1024: 29 -> 16.5 us)
        else if (n256 < cus / 2) tile = 3;
        // ... and when the caller supplies the static operand in fragment order, multi-round grids and every K <= 2048 grid run
        // the direct-B kernel (128 x 256 tiles, two workgroups per CU).  Same-box A/B (tools/db_tower_bench.py): stand-alone it
        // wins everywhere but on sub-round K = 4096 grids (fc2 at 11540 rows: 184 tiles, 0.96-1.0x), most at small batches
        // (2885 rows: fc1 669 -> 857, qkv 564 -> 649 TF/s); the two-stream tower is 2.5-3 % faster with qkv / out_proj / fc1 on it.
        if (tile != 3 && g.Bf && (n256 >= cus || g.K <= 2048)) tile = 12;
#if SLIME_OPT_PP192
        // measured alternative (round 6 re-measurement of the rule's first paragraph with today's kernels): sub-round ping-pong grids
        // (fc2 at the 20-crop half batch: 184 tiles) on 192-row tiles (244 tiles = 0.95 of a round)
        if (tile == 4 && n256 < cus && n192 <= cus) tile = 9;
#endif
    }
    // ... and K > 2048 grids (fc2) in the band where the rule above falls back to the two-stage 128 x 128 kernel although its grid
    // exceeds one workgroup per CU (half batches of 8-12 crops): the direct-B kernel's two-k-step fragment flight covers an under-
    // filled chip better (tower over 17 / 20 / 24 crops 7.32 -> 6.87 / 8.30 -> 7.99 / 9.92 -> 9.29 ms; out_proj, K = 1024, does not gain)
    if (tile == 3 && g.Bf && g.K > 2048 && g.N % 256 == 0 && (long)((g.M + 127) / 128) * (g.N / 128) > num_cus()) tile = 12;
    // ... and 128 x 128 grids of at most one workgroup per CU (rank shards of 1-5 crops, single images, small adapter batches) run the
    // three-stage form of that kernel: nothing else on the CU covers a k-tile that has not landed, so the DMA runs two tiles ahead
    // (tower over 1 / 3 / 5 crops 3.12 -> 2.40 / 3.25 -> 2.75 / 3.68 -> 3.18 ms; beyond one workgroup per CU the two-stage form's second
    // resident workgroup is worth more: 9 crops 4.78 vs 5.10 -- tools/rank_shapes.py, profiles/r03_small_batch_latency_c.txt)
    if (tile == 3 && (long)((g.M + 127) / 128) * (g.N / 128) <= num_cus()) tile = 15;
    // Measured alternative, OFF (round 6, VERDICT r5 item 4: a row-tile height that fills the round): 96 x 256 direct-B tiles (tile 19,
    // gemm_db_kernel<.., 6>; bit-identical: same k order per accumulator, same epilogue).  Stand-alone the quantisation argument holds
    // for the sub-round launches -- out_proj at 20 crops 43.7 -> 39.8 us (364 -> 484 workgroups in 512 slots), fc2 at 9 crops 70.0 -> 65.2,
    // q/k/v at 5 crops 30.5 -> 27.1 -- and fails for the multi-round ones (q/k/v, fc1 at 20 crops: +3 / +6 %: the per-workgroup prologue /
    // epilogue of the extra tiles costs more than the round returns).  Inside the product's stream policy it buys nothing robust: as a
    // rule for every direct-B grid that fits one round of slots it is ahead on one stream and 2-5 % BEHIND when two half batches co-run
    // (10 / 12 / 21 / 24 crops), and out_proj on it costs the bench step 0.3-0.5 %; narrowed to fc2 at 8-10 crops per stream (96-row grid
    // <= one workgroup per CU) it measured -2.6 / -1.7 / -1.3 % at 16 / 17 / 20 crops on one lease and -0.75 / -1.3 / -0.3 % (+1.7 % at
    // 21) on the next -- below the 1 % kill rule.  profiles/r06_tile_ab_attention_conflict.txt, r06_small_tiles.txt,
    // r06_db96_first_rule_ab.txt, r06_db96_narrow_rule_ab.txt.
#if SLIME_OPT_DB96
    if (tile == 12 && g.K > 2048 && (long)((g.M + 95) / 96) * (g.N / 256) <= (long)num_cus()) tile = 19;
#endif
    // ... and grids that leave HALF the CUs without even a 128 x 128 workgroup (one to three crops: BASELINE config 1, the smallest rank
    // shards) run 64 x 64 tiles on the same three-stage ring (tile 18, round 5): four times the workgroups, and a wave's chain per k-tile
    // is 8 MFMAs instead of 32 -- these launches are bound by that dependent chain (fc2 at one crop: 40 workgroups x 64 k-tiles), not by
    // throughput.  Same k order per accumulator, same epilogue: bit-identical to every other tile.
#if SLIME_OPT_TILE64 == 2      // measured alternative: every grid of at most one 128 x 128 workgroup per CU (tools/small_latency_ab.py)
    if (tile == 15) tile = 18;
#elif SLIME_OPT_TILE64 == 3    // measured alternative: the same, for the long-K launches only (fc2: 64 k-tiles per workgroup): slower from 4 crops on
    if (tile == 15 && (g.K >= 2048 || (long)((g.M + 127) / 128) * (g.N / 128) * 2 <= num_cus())) tile = 18;
#elif SLIME_OPT_TILE64
    if (tile == 15 && (long)((g.M + 127) / 128) * (g.N / 128) * 2 <= num_cus()) tile = 18;
#endif
    return tile;
}

#ifdef SLIME_DIAG
static int g_force_tile = 0;   // 0 = auto_tile()
static int g_sched = 1;        // 0 = compiler schedule, 1 = pinned software pipeline (lock-step kernels)
extern "C" void slime_gemm_force_tile(int t) { g_force_tile = t; }
extern "C" void slime_gemm_set_sched(int s) { g_sched = s; }
// per-shape tile override table (N, K) -> tile, consulted in auto mode; tile 0 clears the table
static int g_rule_n[8], g_rule_k[8], g_rule_tile[8], g_rules = 0;
extern "C" void slime_gemm_set_shape_tile(int N, int K, int tile) {
    if (tile == 0) { g_rules = 0; return; }
    for (int i = 0; i < g_rules; ++i)
        if (g_rule_n[i] == N && g_rule_k[i] == K) { g_rule_tile[i] = tile; return; }
    if (g_rules < 8) { g_rule_n[g_rules] = N; g_rule_k[g_rules] = K; g_rule_tile[g_rules] = tile; ++g_rules; }
}
#endif

// epilogues the 96-row direct-B tile is built for: fc2's (the one launch the dispatch gives it); the diagnostic build adds the tower's
// other two for the per-shape A/B tools (tools/r6_tile_ab.py, r6_small_tiles.py).  Every other epilogue keeps 128 rows.
constexpr bool db96_epilogue(int epi) {
#ifdef SLIME_DIAG
    if (epi == SLIME_EPI_BIAS_T || epi == SLIME_EPI_BIAS_QUICKGELU_T) return true;
#endif
    return epi == SLIME_EPI_BIAS_RESID_SPLIT_LN;
}

template <typename T, int EPI>
static int launch_epi(const GemmArgs& g, hipStream_t stream) {
    int tile = auto_tile(g);
#ifdef SLIME_DIAG
    if (g_force_tile != 0) tile = g_force_tile;
    else
        for (int i = 0; i < g_rules; ++i)
            if (g_rule_n[i] == g.N && g_rule_k[i] == g.K) tile = g_rule_tile[i];
    if (tile == 2) tile = 1;
    if ((!g.B || g.row_map) && (tile == 5 || tile == 6 || tile == 7 || tile == 8)) tile = 4;   // persistent / 32x32 ping-pong variants: row-major B only, no row map
    if ((tile == 1 || (tile >= 4 && tile != 15 && tile != 18)) && g.N % 256 != 0) tile = 3;
    if ((tile == 12 || tile == 13 || tile == 19) && !g.Bf) tile = tile == 13 ? 3 : 11;
    if (tile == 6 || tile == 8) tile = 7;
    if (tile == 7 && (EPI == SLIME_EPI_BIAS_RESID_F32_LN || EPI == SLIME_EPI_BIAS_RESID_T || EPI == SLIME_EPI_BIAS_RESID_SPLIT_LN)) tile = 4;     // the 32x32 variant has neither epilogue
    if (tile == 7) return launch_pp32b<T, EPI>(g, stream);
    if (tile == 5 && g.K < 128) tile = 4;                              // persistent kernel needs >= 2 k-tiles
    if (tile == 5 && ((size_t)g.M * g.lda * 2 >= (1ull << 32) || (size_t)g.N * g.K * 2 >= (1ull << 32))) tile = 4;   // 32-bit row offsets
    if (tile == 5) return launch_ppp<T, EPI>(g, stream);
#if !SLIME_OPT_PP192
    if (tile == 9) return launch_pp192<T, EPI>(g, stream);
#endif
    if (tile == 1) return g_sched == 0 ? launch_cfg<T, 256, 256, 2, 4, EPI, 0>(g, stream) : launch_cfg<T, 256, 256, 2, 4, EPI, 1>(g, stream);
    if (tile == 3 && g_sched == 0) return launch_cfg<T, 128, 128, 2, 2, EPI, 0>(g, stream);
#endif
#ifdef SLIME_DIAG
    if (tile == 16 || tile == 17) {                                        // persistent direct-B, measured alternatives: 16 = 16x16x32 MFMAs, 17 = 32x32x16
        if (slime_diag_ps_usable(g, EPI)) return slime_diag_launch_ps(g, T::id, EPI, tile, stream);
        tile = g.Bf ? 12 : 11;
    }
#endif
    // B = NULL (the caller holds the fragment-order image only): the lock-step, ping-pong and direct-B kernels read it; the stream
    // kernel (never picked by auto_tile when a fragment image exists) is replaced by the direct-B kernel, or the ping-pong one
    if (!g.B && (tile == 10 || tile == 11)) tile = g.Bf ? 12 : 4;
#if SLIME_OPT_PP192
    if (tile == 9) return launch_pp192<T, EPI>(g, stream);
#endif
    if (tile == 15) return launch_cfg<T, 128, 128, 2, 2, EPI, 2>(g, stream);      // 128 x 128, three-stage ring (small grids)
    if (tile == 18) return launch_cfg<T, 64, 64, 4, 1, EPI, 2>(g, stream);        // 64 x 64, three-stage ring (the smallest grids)
    if (tile == 12) return launch_db<T, EPI, 8>(g, stream);
    if (tile == 19) {                                             // 96-row direct-B tiles: instantiated for the tower's epilogues (db96_epilogue)
        if constexpr (db96_epilogue(EPI)) {
            if (g.Bf) return launch_db<T, EPI, 6>(g, stream);
        }
        return launch_db<T, EPI, 8>(g, stream);
    }
#ifdef SLIME_DIAG
    if (tile == 13) return launch_db<T, EPI, 4>(g, stream);       // measured alternative (64-row direct-B tiles), see gemm_db_kernel
#else
    if (tile == 13) return launch_db<T, EPI, 8>(g, stream);
#endif
    if (tile == 4) return launch_pp<T, EPI>(g, stream);
    if (tile == 10) return launch_w4<T, EPI, 6>(g, stream);
    if (tile == 11) return launch_w4<T, EPI, 8>(g, stream);
    return launch_cfg<T, 128, 128, 2, 2, EPI, 1>(g, stream);
}

template <typename T>
static int launch_T(const GemmArgs& g, int epi, hipStream_t stream) {
    switch (epi) {
        case SLIME_EPI_BIAS_T: return launch_epi<T, SLIME_EPI_BIAS_T>(g, stream);
        case SLIME_EPI_BIAS_QUICKGELU_T: return launch_epi<T, SLIME_EPI_BIAS_QUICKGELU_T>(g, stream);
        case SLIME_EPI_BIAS_GELU_T: return launch_epi<T, SLIME_EPI_BIAS_GELU_T>(g, stream);
        case SLIME_EPI_BIAS_F32: return launch_epi<T, SLIME_EPI_BIAS_F32>(g, stream);
        case SLIME_EPI_BIAS_RESID_F32: return launch_epi<T, SLIME_EPI_BIAS_RESID_F32>(g, stream);
        case SLIME_EPI_BIAS_RESID_F32_LN: return launch_epi<T, SLIME_EPI_BIAS_RESID_F32_LN>(g, stream);
        case SLIME_EPI_BIAS_RESID_T: return launch_epi<T, SLIME_EPI_BIAS_RESID_T>(g, stream);
        case SLIME_EPI_BIAS_GELU_MIX_T: return launch_db<T, SLIME_EPI_BIAS_GELU_MIX_T, 8>(g, stream);    // direct-B only (checked by slime_gemm_ex)
        case SLIME_EPI_BIAS_RESID_SPLIT_LN: return launch_epi<T, SLIME_EPI_BIAS_RESID_SPLIT_LN>(g, stream);
    }
    slime_set_error("gemm: unknown epilogue %d", epi);
    return SLIME_EINVAL;
}

// Name of the kernel instantiation slime_gemm dispatches to for a shape, as rocprofv3 prints it (bench.py labels its per-kernel
// figures with it, so the bench line and the profiler summary name the same symbol).
extern "C" int slime_gemm_kernel_name(int M, int N, int K, int dtype, int epilogue, int b_frag, char* out, size_t out_len) {
    SLIME_REQUIRE(out && out_len > 0 && M > 0 && N > 0 && K > 0, "gemm_kernel_name: bad input");
    GemmArgs g{nullptr, nullptr, nullptr, nullptr, K, N, M, N, K, 0, nullptr, nullptr, 0, nullptr, 0.f, nullptr, 0, nullptr,
               (b_frag && N % 256 == 0) ? "" : nullptr, 0, nullptr, 0};       // a fragment image makes the direct-B kernel eligible at N % 256 == 0 only
    const int tile = auto_tile(g);
    const char* t = dtype == SLIME_F16 ? "F16" : "BF16";
    const int ktag = K >= 2048 ? 1 : 0;
    if (tile == 19) snprintf(out, out_len, "gemm_db_kernel<%s, %d, %d, %d>", t, epilogue, ktag, db96_epilogue(epilogue) ? 6 : 8);
    else if (tile == 12 || tile == 13) snprintf(out, out_len, "gemm_db_kernel<%s, %d, %d, %d>", t, epilogue, ktag, tile == 12 ? 8 : 4);
    else if (tile == 4 || tile == 9) snprintf(out, out_len, "gemm_pp_kernel<%s, %d, %d, 0, %d>", t, epilogue, ktag, tile == 4 ? 4 : 3);
    else if (tile == 10 || tile == 11) snprintf(out, out_len, "gemm_w4_kernel<%s, %d, %d, %d, 0>", t, epilogue, ktag, tile == 10 ? 6 : 8);
    else if (tile == 18) snprintf(out, out_len, "gemm_kernel<%s, 64, 64, 4, 1, %d, 2>", t, epilogue);
    else snprintf(out, out_len, "gemm_kernel<%s, 128, 128, 2, 2, %d, %d>", t, epilogue, tile == 15 ? 2 : 1);
    return SLIME_OK;
}

extern "C" int slime_gemm_ex(const slime_gemm_args* a, void* stream) {
    SLIME_REQUIRE(a, "gemm: null argument block");
    const int M = a->M, N = a->N, K = a->K, lda = a->lda, ldc = a->ldc;
    SLIME_REQUIRE(a->A && a->C && (a->B || a->B_frag), "gemm: null pointer");
    SLIME_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty shape M=%d N=%d K=%d", M, N, K);
    SLIME_REQUIRE(K % 64 == 0, "gemm: K=%d must be a multiple of 64", K);
    SLIME_REQUIRE(N % 128 == 0, "gemm: N=%d must be a multiple of 128", N);
    SLIME_REQUIRE(lda >= K && lda % 8 == 0 && ldc >= N && ldc % 8 == 0, "gemm: bad leading dims lda=%d ldc=%d", lda, ldc);
    SLIME_REQUIRE(a->B || (slime_gemm_b_frag_usable(N, K) && ((uintptr_t)a->B_frag % 16) == 0),
                  "gemm: B = NULL needs a fragment-order image this shape can run from (slime_gemm_b_frag_usable)");
    SLIME_REQUIRE(((uintptr_t)a->A % 16 == 0) && ((uintptr_t)a->B % 16 == 0) && ((uintptr_t)a->C % 16 == 0) &&
                  (!a->bias || (uintptr_t)a->bias % 16 == 0), "gemm: pointers must be 16-byte aligned");
    if (a->ln_stats) {
        SLIME_REQUIRE(a->epilogue == SLIME_EPI_BIAS_T || a->epilogue == SLIME_EPI_BIAS_QUICKGELU_T,
                      "gemm: the LayerNorm fold is built for the BIAS_T / BIAS_QUICKGELU_T epilogues");
        SLIME_REQUIRE(a->ln_colsum && a->ln_groups > 0 && ((uintptr_t)a->ln_colsum % 16) == 0 && ((uintptr_t)a->ln_stats % 16) == 0,
                      "gemm: LayerNorm fold needs ln_colsum [N] and ln_groups partial sums per row (16-byte aligned)");
        SLIME_REQUIRE(a->ln_groups * 64 == K, "gemm: LayerNorm fold: ln_groups=%d partial sums of 64 columns must cover K=%d", a->ln_groups, K);
    }
    if (a->epilogue == SLIME_EPI_BIAS_RESID_F32_LN)
        SLIME_REQUIRE(a->x16 && a->stats_out && a->ldx >= N && a->ldx % 8 == 0 && ((uintptr_t)a->x16 % 16) == 0 &&
                      ((uintptr_t)a->stats_out % 8) == 0 && N % 64 == 0, "gemm: BIAS_RESID_F32_LN needs x16 [M, ldx] and stats_out [M, N/64, 2]");
    if (a->row_map)
        SLIME_REQUIRE(!a->ln_stats && ((uintptr_t)a->row_map % 4) == 0 &&
                      (a->epilogue == SLIME_EPI_BIAS_T || a->epilogue == SLIME_EPI_BIAS_QUICKGELU_T || a->epilogue == SLIME_EPI_BIAS_GELU_T ||
                       a->epilogue == SLIME_EPI_BIAS_F32),
                      "gemm: row_map goes with the plain T / fp32 epilogues (BIAS_T, BIAS_QUICKGELU_T, BIAS_GELU_T, BIAS_F32; no LayerNorm fold)");
    if (a->epilogue == SLIME_EPI_BIAS_RESID_SPLIT_LN)
        SLIME_REQUIRE(a->lo8 && a->stats_out && a->ldlo >= N && a->ldlo % 8 == 0 && ((uintptr_t)a->lo8 % 8) == 0 &&
                      ((uintptr_t)a->stats_out % 8) == 0 && N % 64 == 0 && a->lo8 != a->C,
                      "gemm: BIAS_RESID_SPLIT_LN needs the split residual stream C = hi T [M, ldc], lo8 int8 [M, ldlo] and stats_out [M, N/64, 2]");
    if (a->epilogue == SLIME_EPI_BIAS_RESID_T)
        SLIME_REQUIRE(a->resid && a->ldr >= N && a->ldr % 8 == 0 && ((uintptr_t)a->resid % 16) == 0,
                      "gemm: BIAS_RESID_T needs resid T [M, ldr >= N] (16-byte aligned, ldr a multiple of 8)");
    // the fragment-order copy of B is optional; it is only usable with whole 64-column tiles, >= 2 k-steps per tile and 32-bit A offsets
    const bool frag_ok = a->B_frag && N % 256 == 0 && ((uintptr_t)a->B_frag % 16) == 0 && (size_t)128 * lda * 2 < (1ull << 32);
    if (a->epilogue == SLIME_EPI_BIAS_GELU_MIX_T)
        SLIME_REQUIRE(frag_ok && a->A2 && a->mix_gates && ((uintptr_t)a->A2 % 16) == 0 && ((uintptr_t)a->mix_gates % 8) == 0 && !a->ln_stats,
                      "gemm: BIAS_GELU_MIX_T runs on the direct-B kernel only: needs B_frag (N %% 256 == 0), A2 [M, lda] and mix_gates [M, 2]");
    GemmArgs g{(const char*)a->A, (const char*)a->B, a->bias, a->C, lda, ldc, M, N, K, g_group_m, g_dbg,
               a->ln_stats, a->ln_groups, a->ln_colsum, a->ln_eps, (char*)a->x16, a->ldx, a->stats_out,
               frag_ok ? (const char*)a->B_frag : nullptr, g_db_abl, (const char*)a->resid, a->ldr, (const char*)a->A2, a->mix_gates,
               (char*)a->lo8, a->ldlo, (const char*)a->B_frag, a->row_map};
    hipStream_t s = (hipStream_t)stream;
    if (a->dtype == SLIME_BF16) return launch_T<BF16>(g, a->epilogue, s);
    if (a->dtype == SLIME_F16) return launch_T<F16>(g, a->epilogue, s);
    slime_set_error("gemm: dtype %d is not a 16-bit MFMA type", a->dtype);
    return SLIME_EINVAL;
}

extern "C" int slime_gemm_b_frag_usable(int N, int K) { return (N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0) ? 1 : 0; }

extern "C" size_t slime_gemm_packed_b_bytes(int N, int K) { return (N > 0 && K > 0) ? (size_t)N * K * 2 : 0; }

extern "C" int slime_gemm_pack_b(const void* B, int N, int K, void* out, void* stream) {
    SLIME_REQUIRE(B && out && B != out, "gemm_pack_b: null or aliasing pointers");
    SLIME_REQUIRE(N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0, "gemm_pack_b: N=%d must be a multiple of 64 and K=%d of 64", N, K);
    SLIME_REQUIRE(((uintptr_t)B % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm_pack_b: pointers must be 16-byte aligned");
    const size_t total = (size_t)N * K / 8;
    hipLaunchKernelGGL(pack_b_frag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const u32x4*)B, (u32x4*)out, N, K);
    SLIME_CHECK_LAUNCH("gemm_pack_b");
    return SLIME_OK;
}

extern "C" int slime_gemm(const void* A, int lda, const void* B, const float* bias, void* C, int ldc,
                          int M, int N, int K, int dtype, int epilogue, void* stream) {
    SLIME_REQUIRE(epilogue != SLIME_EPI_BIAS_RESID_F32_LN && epilogue != SLIME_EPI_BIAS_RESID_T && epilogue != SLIME_EPI_BIAS_GELU_MIX_T &&
                  epilogue != SLIME_EPI_BIAS_RESID_SPLIT_LN,
                  "gemm: BIAS_RESID_F32_LN / BIAS_RESID_T / BIAS_GELU_MIX_T / BIAS_RESID_SPLIT_LN take extra operands: use slime_gemm_ex");
    slime_gemm_args a{};
    a.A = A; a.lda = lda; a.B = B; a.bias = bias; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.dtype = dtype; a.epilogue = epilogue;
    return slime_gemm_ex(&a, stream);
}
