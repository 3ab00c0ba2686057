// gemm.hip -- MFMA GEMM with fused epilogues for gfx950:  C = epi(A[M,K] * B[N,K]^T + bias).
//
// Both operands are K-contiguous (activations row-major, weights in nn.Linear layout), which is
// exactly what v_mfma_f32_16x16x32_{bf16,f16} wants: every lane feeds 8 consecutive k.
//
// Structure (per workgroup, BM x BN output tile, BK = 64):
//   * HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane, no VGPR round trip), two LDS
//     stages, one barrier per K-tile; the DMA of tile t+1 is in flight while tile t is multiplied.
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
//     32 CUs of an XCD work on an 8 x 4 patch of tiles that share A/B panels through their L2.
//
// Epilogues: bias -> T; bias+quick-GELU -> T; bias+erf-GELU -> T; bias -> fp32; fp32 += (residual).
#include "common.h"

struct GemmArgs {
    const char* A; const char* B; const float* bias; void* C;
    int lda, ldc, M, N, K;
};

template <int EPI> struct EpiOutIsT { static constexpr bool value = (EPI <= SLIME_EPI_BIAS_GELU_T); };

template <typename T, int EPI>
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, int row, int col, float* v) {
    if (g.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(g.bias + col);
        const float4 b1 = *reinterpret_cast<const float4*>(g.bias + col + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if constexpr (EPI == SLIME_EPI_BIAS_QUICKGELU_T) {
#pragma unroll
        for (int i = 0; i < 8; ++i)   // x * sigmoid(1.702 x)
            v[i] = v[i] * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v[i]));
    } else if constexpr (EPI == SLIME_EPI_BIAS_GELU_T) {
#pragma unroll
        for (int i = 0; i < 8; ++i)   // exact (erf) GELU, as nn.GELU()
            v[i] = 0.5f * v[i] * (1.0f + erff(v[i] * 0.70710678118654752440f));
    }
    if constexpr (EpiOutIsT<EPI>::value) {
        char* p = reinterpret_cast<char*>(g.C) + ((size_t)row * g.ldc + col) * 2;
        *reinterpret_cast<u32x4*>(p) = pack8<T>(v);
    } else {
        float* p = reinterpret_cast<float*>(g.C) + (size_t)row * g.ldc + col;
        if constexpr (EPI == SLIME_EPI_BIAS_RESID_F32) {
            const float4 h0 = *reinterpret_cast<const float4*>(p);
            const float4 h1 = *reinterpret_cast<const float4*>(p + 4);
            v[0] += h0.x; v[1] += h0.y; v[2] += h0.z; v[3] += h0.w;
            v[4] += h1.x; v[5] += h1.y; v[6] += h1.z; v[7] += h1.w;
        }
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
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
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int rho = (i * NW + wave) * 8 + lrow;   // LDS row
        const int nl = rho & 15;
        const int nphys = (rho & ~31) + 8 * (nl >> 2) + 4 * ((rho >> 4) & 1) + (nl & 3);
        b_src[i] = g.B + ((size_t)(n0 + nphys) * g.K) * 2 + lchunk * 16;
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
            b_src[i] += BK * 2;
        }
    };

    // ---- fragment read offsets (per lane, bytes inside a stage) --------------------------------
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int sw = ((ks * 4 + (lane >> 4)) ^ (lane & 7)) << 4;
        a_off[ks] = (wm * TM + (lane & 15)) * 128 + sw;
        b_off[ks] = A_BYTES + (wn * TN + (lane & 15)) * 128 + sw;
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
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

    // ---- epilogue: lane holds, per (mi, tile pair), 8 consecutive columns of one row ------------
    const int q = lane >> 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = m0 + wm * TM + i * 16 + (lane & 15);
        if (row < g.M) {
#pragma unroll
            for (int p = 0; p < NI / 2; ++p) {
                const int col = n0 + wn * TN + 32 * p + 8 * q;
                float v[8] = {acc[i][2 * p][0], acc[i][2 * p][1], acc[i][2 * p][2], acc[i][2 * p][3],
                              acc[i][2 * p + 1][0], acc[i][2 * p + 1][1], acc[i][2 * p + 1][2], acc[i][2 * p + 1][3]};
                epilogue_store<T, EPI>(g, row, col, v);
            }
        }
    }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int SCHED>
static int launch_cfg(const GemmArgs& g, hipStream_t stream) {
    constexpr int STAGE = (BM + BN) * 64 * 2;
    constexpr int LDS = 2 * STAGE;
    auto kern = gemm_kernel<T, BM, BN, WAVES_M, WAVES_N, EPI, SCHED>;
    static bool attr_set = false;   // benign race: idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { slime_set_error("gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return SLIME_ELAUNCH; }
        attr_set = true;
    }
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = g.N / BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(WAVES_M * WAVES_N * 64), LDS, stream, g);
    SLIME_CHECK_LAUNCH("gemm");
    return SLIME_OK;
}

// Tile choice.  256x256 (8 waves, 128 KiB LDS, 1 WG/CU) is the throughput tile; 256x128 is used
// when it fills the last round of CUs better (N = 1024 GEMMs at M ~ 23k: 728 vs 364 workgroups);
// 128x128 (4 waves, 64 KiB) covers narrow N (tiny geometries) and small M.
static int g_force_tile = 0;   // test/bench hook: 0 auto, 1 = 256x256, 2 = 256x128, 3 = 128x128
static int g_sched = 1;        // test/bench hook: 0 = compiler schedule, 1 = pinned software pipeline
extern "C" void slime_gemm_force_tile(int t) { g_force_tile = t; }
extern "C" void slime_gemm_set_sched(int s) { g_sched = s; }

template <typename T, int EPI>
static int launch_epi(const GemmArgs& g, hipStream_t stream) {
    int tile = g_force_tile;
    if (tile == 0) {
        if (g.N % 256 == 0 && g.M >= 1024) {
            const long b256 = (long)((g.M + 255) / 256) * (g.N / 256);
            const long b128 = b256 * 2;
            auto eff = [](long b) { long r = (b + 255) / 256; return (double)b / (double)(r * 256); };
            tile = (eff(b128) > eff(b256) + 0.12) ? 2 : 1;
        } else {
            tile = 3;
        }
    }
    if (tile == 1 && g.N % 256 != 0) tile = 3;
    if (g_sched == 0) {
        switch (tile) {
            case 1: return launch_cfg<T, 256, 256, 2, 4, EPI, 0>(g, stream);
            case 2: return launch_cfg<T, 256, 128, 4, 2, EPI, 0>(g, stream);
            default: return launch_cfg<T, 128, 128, 2, 2, EPI, 0>(g, stream);
        }
    }
    switch (tile) {
        case 1: return launch_cfg<T, 256, 256, 2, 4, EPI, 1>(g, stream);
        case 2: return launch_cfg<T, 256, 128, 4, 2, EPI, 1>(g, stream);
        default: return launch_cfg<T, 128, 128, 2, 2, EPI, 1>(g, stream);
    }
}

template <typename T>
static int launch_T(const GemmArgs& g, int epi, hipStream_t stream) {
    switch (epi) {
        case SLIME_EPI_BIAS_T: return launch_epi<T, SLIME_EPI_BIAS_T>(g, stream);
        case SLIME_EPI_BIAS_QUICKGELU_T: return launch_epi<T, SLIME_EPI_BIAS_QUICKGELU_T>(g, stream);
        case SLIME_EPI_BIAS_GELU_T: return launch_epi<T, SLIME_EPI_BIAS_GELU_T>(g, stream);
        case SLIME_EPI_BIAS_F32: return launch_epi<T, SLIME_EPI_BIAS_F32>(g, stream);
        case SLIME_EPI_BIAS_RESID_F32: return launch_epi<T, SLIME_EPI_BIAS_RESID_F32>(g, stream);
    }
    slime_set_error("gemm: unknown epilogue %d", epi);
    return SLIME_EINVAL;
}

extern "C" int slime_gemm(const void* A, int lda, const void* B, const float* bias, void* C, int ldc,
                          int M, int N, int K, int dtype, int epilogue, void* stream) {
    SLIME_REQUIRE(A && B && C, "gemm: null pointer");
    SLIME_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty shape M=%d N=%d K=%d", M, N, K);
    SLIME_REQUIRE(K % 64 == 0, "gemm: K=%d must be a multiple of 64", K);
    SLIME_REQUIRE(N % 128 == 0, "gemm: N=%d must be a multiple of 128", N);
    SLIME_REQUIRE(lda >= K && lda % 8 == 0 && ldc >= N && ldc % 8 == 0, "gemm: bad leading dims lda=%d ldc=%d", lda, ldc);
    SLIME_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0) &&
                  (!bias || (uintptr_t)bias % 16 == 0), "gemm: pointers must be 16-byte aligned");
    GemmArgs g{(const char*)A, (const char*)B, bias, C, lda, ldc, M, N, K};
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SLIME_BF16) return launch_T<BF16>(g, epilogue, s);
    if (dtype == SLIME_F16) return launch_T<F16>(g, epilogue, s);
    slime_set_error("gemm: dtype %d is not a 16-bit MFMA type", dtype);
    return SLIME_EINVAL;
}
