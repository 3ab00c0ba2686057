// gemm_shared.h -- what the GEMM translation units share: the argument block, the LayerNorm-statistics finaliser and the asm
// helpers of the direct-B kernels (gemm.hip; gemm_ps.hip = the persistent measured alternatives, diagnostic build).
#pragma once
#include "common.h"
#include <type_traits>

struct GemmArgs {
    const char* A; const char* B; const float* bias; void* C;
    int lda, ldc, M, N, K;
    int group_m;      // row tiles per L2 patch of the ping-pong kernel (0 = default 4)
    unsigned long long* dbg;   // ABL & 8 builds only: 4 s_memtime stamps per workgroup
    // LayerNorm folded into this GEMM (consumer side; epilogues BIAS_T / BIAS_QUICKGELU_T): A holds the UN-normalised 16-bit rows x,
    // B = W . diag(gamma), bias = b + W beta, and the epilogue applies  rstd * (acc - mu * colsum[n]) + bias[n]  with the row
    // statistics (mu, rstd) finalised per workgroup from the producer's per-64-column partial sums.  NULL = plain GEMM.
    const float* ln_stats; int ln_groups; const float* ln_colsum; float ln_eps;
    // producer side (epilogue BIAS_RESID_F32_LN): 16-bit copy of the updated residual rows and their partial sums
    char* x16; int ldx; float* stats_out;
    // B in MFMA-fragment order (slime_gemm_pack_b), or NULL: lets the dispatch pick gemm_db_kernel
    const char* Bf;
    // gemm_db_kernel timing ablations (diagnostic build; wrong results): 1 = every tile's epilogue writes rows 0..127 (the stores
    // stay in L2: no HBM write burst), 2 = no epilogue at all
    int db_abl;
    // epilogue BIAS_RESID_T: 16-bit residual rows added before the rounding (may alias C)
    const char* resid; int ldr;
    // epilogue BIAS_GELU_MIX_T: second stacked operand (rows of the other expert) and the per-token gate pair
    const char* A2; const float* mix_gates;
    // epilogue BIAS_RESID_SPLIT_LN: lower part of the split residual stream, one signed byte per element (C is the upper part), read and written in place; ldlo in bytes
    char* lo; int ldlo;
    // the fragment-order image whatever the shape (Bf is set only where the direct-B kernel is eligible): the LDS-staged kernels
    // DMA from it when B == NULL
    const char* Bimg;
    // optional scatter of the output rows (generic T / fp32 epilogues): row r is stored at C row row_map[r]
    const int* row_map;
};

// (sum x, sum x^2) of a K-wide row -> (rstd, -mu rstd).  One shared definition with the operations written out (no contraction left
// to the compiler's discretion): every kernel family finalises the statistics to the same bits.
__device__ __forceinline__ void ln_finalize(float sx, float sq, int K, float eps, float& rstd, float& nmr) {
    const float inv = 1.0f / (float)K;
    const float mu = __fmul_rn(sx, inv);
    const float m2 = __fmul_rn(mu, mu);
    const float var = fmaxf(__fmaf_rn(sq, inv, -m2), 0.f);
    rstd = rsqrtf(var + eps);
    nmr = __fmul_rn(-mu, rstd);
}

template <int OFF>
__device__ __forceinline__ void gload16_frag(u32x4& d, unsigned voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(d) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait_frag(u32x4& d) {       // the consumer side of gload16_frag: names the register, pins the order
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(d) : "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
