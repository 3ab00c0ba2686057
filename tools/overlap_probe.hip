// Can one SIMD overlap MFMA with VALU?  (a) across two co-resident waves, (b) inside one wave's stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define REP 256
// mode bit0: this wave runs MFMA chain(s); bit1: this wave runs exp chain(s).  role picks by wave id.
__global__ void k(float* out, unsigned long long* cyc, int mode_lo, int mode_hi, int valu_per_mfma) {
    const int wave = threadIdx.x >> 6;
    const int mode = wave < 4 ? mode_lo : mode_hi;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 1) {
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        }
    } else if (mode == 2) {
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
        }
    } else if (mode == 5) {          // plain VALU only: 32 fma per rep (8 independent chains x 4)
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], 1.0001f, 0.5f);
        }
    } else if (mode == 6) {          // one stream: 4 MFMAs, then 8 exps (block structure like the attention step)
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (mode == 3) {          // interleaved in ONE stream: 4 MFMAs + 4*valu_per_mfma plain VALU (fma) per rep
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) if (i < valu_per_mfma) x[i] = fmaf(x[i], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (mode == 4) {          // one stream: 4 MFMAs + 2 exps per MFMA
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
                x[2 * j] = __builtin_amdgcn_exp2f(x[2 * j]); x[2 * j + 1] = __builtin_amdgcn_exp2f(x[2 * j + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][3]; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    struct { const char* name; int threads, lo, hi, v; } cases[] = {
        {"MFMA only, 1 wave/SIMD (4 MFMA/rep)", 256, 1, 0, 0}, {"exp only, 1 wave/SIMD (8 exp/rep)", 256, 2, 0, 0},
        {"MFMA || MFMA, 2 waves/SIMD", 512, 1, 1, 0}, {"exp || exp, 2 waves/SIMD", 512, 2, 2, 0},
        {"MFMA(w0-3) || exp(w4-7), 2 waves/SIMD", 512, 1, 2, 0},
        {"fma only, 1 wave/SIMD (32 fma/rep)", 256, 5, 0, 0}, {"fma || fma, 2 waves/SIMD", 512, 5, 5, 0},
        {"MFMA(w0-3) || fma(w4-7)", 512, 1, 5, 0}, {"fma(w0-3) || MFMA(w4-7)", 512, 5, 1, 0}, {"exp(w0-3) || MFMA(w4-7)", 512, 2, 1, 0},
        {"1 wave: 4 MFMA then 8 exp", 256, 6, 0, 0}, {"2 waves: 4 MFMA then 8 exp each", 512, 6, 6, 0},
        {"1 wave: 4x(MFMA + 4 fma)", 256, 3, 0, 4}, {"1 wave: 4x(MFMA + 8 fma)", 256, 3, 0, 8},
        {"2 waves: 4x(MFMA + 8 fma) each", 512, 3, 3, 8}, {"1 wave: 4x(MFMA + 2 exp)", 256, 4, 0, 0}, {"2 waves: 4x(MFMA + 2 exp) each", 512, 4, 4, 0}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(256), dim3(c.threads), 0, 0, out, cyc, c.lo, c.hi, c.v);
        hipDeviceSynchronize();
        unsigned long long h[2048]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double lo = 0, hi = 0; for (int bI = 0; bI < 256; ++bI) { for (int w = 0; w < 4; ++w) lo += h[bI * 8 + w]; for (int w = 4; w < 8; ++w) hi += h[bI * 8 + w]; }
        lo /= 1024.0 * REP; hi /= 1024.0 * REP;
        printf("%-42s: waves0-3 %7.1f cycles/rep   waves4-7 %7.1f cycles/rep\n", c.name, lo, c.threads == 512 ? hi : 0.0);
    }
    return 0;
}
