// Cost of plain VALU instructions inside a dense MFMA stream (one wave per SIMD and two waves per SIMD).
// Each rep: 16 MFMAs (4 independent accumulators x 4) with `nv` VALU instructions of a given kind inserted
// after every `every`-th MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define REP 128
template <int KIND, int NV, int EVERY>
__global__ void k(float* out, unsigned long long* cyc) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    float x[4]; unsigned u[4];
    for (int i = 0; i < 4; ++i) { x[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x + i; }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
            if ((m % EVERY) == EVERY - 1) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i & 3]) : "v"(x[(i + 1) & 3]));
                    if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i & 3]) : "v"(u[(i + 1) & 3]));
                    if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i & 3]));
                    if (KIND == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i & 3]) : "v"(u[(i + 1) & 3]));
                    if (KIND == 4) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
                    if (KIND == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i & 3]) : "v"(x[(i + 1) & 3]), "v"(x[(i + 2) & 3]));
                    if (KIND == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i & 3]) : "v"(x[(i + 1) & 3]), "v"(x[(i + 2) & 3]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][3] + x[j] + (float)u[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int KIND, int NV, int EVERY>
void run(const char* name, float* out, unsigned long long* cyc) {
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL((k<KIND, NV, EVERY>), dim3(256), dim3(threads), 0, 0, out, cyc);
        hipDeviceSynchronize();
        unsigned long long h[2048]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double mx = 0; int nw = threads / 64;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) mx += h[b * 8 + w];
        mx /= 256.0 * nw * REP;
        // per SIMD: waves per SIMD = threads/256; MFMAs per rep per wave = 16
        printf("%-28s nv=%d every=%2d  %d wave/SIMD: %7.1f cycles per 16-MFMA rep per wave (ideal %5.1f)\n", name, NV, EVERY, threads / 256, mx,
               16 * 17.9 * (threads / 256));
    }
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    run<0, 0, 16>("baseline (no VALU)", out, cyc);
    run<0, 1, 16>("v_fma_f32", out, cyc);  run<0, 4, 16>("v_fma_f32", out, cyc);  run<0, 1, 4>("v_fma_f32", out, cyc);  run<0, 1, 1>("v_fma_f32", out, cyc); run<0, 4, 1>("v_fma_f32", out, cyc);
    run<1, 1, 4>("v_add_u32", out, cyc);   run<1, 1, 1>("v_add_u32", out, cyc);
    run<3, 1, 1>("v_mov_b32", out, cyc);
    run<2, 1, 1>("v_exp_f32", out, cyc);   run<2, 2, 1>("v_exp_f32", out, cyc);
    run<4, 1, 1>("s_add_u32", out, cyc);   run<4, 4, 1>("s_add_u32", out, cyc);
    run<5, 1, 1>("v_max3_f32", out, cyc);  run<6, 1, 1>("v_cvt_pk_bf16_f32", out, cyc);
    return 0;
}
