// gemm_ps.hip -- diagnostic library only: round 4's persistent direct-B GEMMs (measured alternatives, tiles 16 / 17) in their own
// translation unit, so that their sixteen straight-line instantiations compile beside gemm.hip instead of behind it.
#include "gemm_shared.h"
#include "gemm_ps32.inc"
#include "gemm_ps.inc"

// Persistent direct-B kernels (gemm_ps32.inc, gemm_ps.inc; diagnostic build): one workgroup per CU, LDS = 4 A stages + 2 row
// tables + bias / colsum of the launch
bool slime_diag_ps_usable(const GemmArgs& g, int epi) {
    const long cus = num_cus();
    const long nblk = (long)((g.M + 127) / 128) * (g.N / 256);
    return g.Bf && g.N % 256 == 0 && g.N <= 8192 && g.K == 1024 && g.ln_stats && g.ln_groups == 16 && g.ln_colsum &&
           (epi == SLIME_EPI_BIAS_T || epi == SLIME_EPI_BIAS_QUICKGELU_T) && cus % 8 == 0 && nblk >= 2 * cus &&
           (size_t)g.M * g.ldc * 2 < (1ull << 31) && (size_t)128 * g.lda * 2 < (1ull << 32);
}
#define PS_LAUNCH(KERN)                                                                                              \
    do {                                                                                                             \
        auto kern_ = KERN;                                                                                           \
        SLIME_SET_LDS_ONCE(kern_, LDS_FIXED + 2 * 8192 * 4, "gemm_ps");                                              \
        hipLaunchKernelGGL(kern_, dim3(num_cus()), dim3(256), LDS_FIXED + 2 * g.N * 4, stream, g);                   \
        SLIME_CHECK_LAUNCH("gemm_ps");                                                                               \
        return SLIME_OK;                                                                                             \
    } while (0)
template <typename T, int EPI>
static int launch_ps32(const GemmArgs& g, hipStream_t stream) {
    if constexpr (EPI == SLIME_EPI_BIAS_T || EPI == SLIME_EPI_BIAS_QUICKGELU_T) {
        constexpr int LDS_FIXED = 4 * 128 * 64 * 2 + 2 * 128 * 8;
        // slime_gemm_set_db_ablation(16 + DBG): 17 = every counted wait drained; 18 / 20 / 22 / 30 = timing-only ablations
        if constexpr (T::id == SLIME_BF16 && EPI == SLIME_EPI_BIAS_QUICKGELU_T) {
            if (g.db_abl == 17) PS_LAUNCH((gemm_ps32_kernel<T, EPI, 16, 1>));
            if (g.db_abl == 18) PS_LAUNCH((gemm_ps32_kernel<T, EPI, 16, 2>));
            if (g.db_abl == 20) PS_LAUNCH((gemm_ps32_kernel<T, EPI, 16, 4>));
            if (g.db_abl == 22) PS_LAUNCH((gemm_ps32_kernel<T, EPI, 16, 6>));
            if (g.db_abl == 30) PS_LAUNCH((gemm_ps32_kernel<T, EPI, 16, 14>));
            if (g.db_abl == 38) PS_LAUNCH((gemm_ps32_kernel<T, EPI, 16, 6 + 16>));          // + no weight requests
            if (g.db_abl == 54) PS_LAUNCH((gemm_ps32_kernel<T, EPI, 16, 6 + 32>));          // + no LDS-DMA
            if (g.db_abl == 70) PS_LAUNCH((gemm_ps32_kernel<T, EPI, 16, 6 + 16 + 32>));     // + neither
            if (g.db_abl == 134) PS_LAUNCH((gemm_ps32_kernel<T, EPI, 16, 6 + 16 + 32 + 64>)); // + no fragment reads: MFMAs + scalar bookkeeping
        }
        PS_LAUNCH((gemm_ps32_kernel<T, EPI, 16>));
    } else {
        slime_set_error("gemm_ps32: epilogue %d has no persistent form", EPI);
        return SLIME_EINVAL;
    }
}
template <typename T, int EPI>
static int launch_ps16(const GemmArgs& g, hipStream_t stream) {
    if constexpr ((EPI == SLIME_EPI_BIAS_T || EPI == SLIME_EPI_BIAS_QUICKGELU_T) && T::id == SLIME_BF16) {
        constexpr int LDS_FIXED = 4 * 128 * 64 * 2 + 2 * 128 * 8;
        if constexpr (EPI == SLIME_EPI_BIAS_QUICKGELU_T) {
            if (g.db_abl == 18) PS_LAUNCH((gemm_ps_kernel<T, EPI, 16, 2>));
        }
        PS_LAUNCH((gemm_ps_kernel<T, EPI, 16>));
    } else {
        return launch_ps32<T, EPI>(g, stream);
    }
}
#undef PS_LAUNCH

int slime_diag_launch_ps(const GemmArgs& g, int dtype, int epi, int tile, hipStream_t stream) {
#define PS_CASE(T, EPI)                                                                                  \
    if (dtype == T::id && epi == EPI) return tile == 16 ? launch_ps16<T, EPI>(g, stream) : launch_ps32<T, EPI>(g, stream);
    PS_CASE(BF16, SLIME_EPI_BIAS_T) PS_CASE(BF16, SLIME_EPI_BIAS_QUICKGELU_T) PS_CASE(F16, SLIME_EPI_BIAS_T) PS_CASE(F16, SLIME_EPI_BIAS_QUICKGELU_T)
#undef PS_CASE
    slime_set_error("gemm_ps: no persistent form for dtype %d epilogue %d", dtype, epi);
    return SLIME_EINVAL;
}
