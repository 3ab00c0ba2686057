// Can a softmax-sized VALU load hide under the matrix pipe?  Per 32-query x 32-key attention block a wave issues the same
// filler mix (16 v_exp_f32, 8 v_cvt_pk_bf16_f32, 8 v_max3_f32, 4 v_add_u32) next to either 18 v_mfma_f32_16x16x32_bf16 or
// 9 v_mfma_f32_32x32x16_bf16 (same flops), fillers spread evenly between the MFMAs, one or two waves per SIMD.
// Prints cycles per block and the attention-equivalent TF/s (4*32*32*64 flops per block per wave... per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma32_filler tools/mfma32_filler.hip && tools/mfma32_filler
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define REP 256

__device__ __forceinline__ void filler(int i, float (&x)[8], unsigned (&u)[8]) {
    const int kind = i % 9;      // 36 fillers per block: 16 exp, 8 cvt, 8 max3, 4 add  (pattern of 9: e e c m e e c m a)
    if (kind == 0 || kind == 1 || kind == 4 || kind == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i & 7]));
    else if (kind == 2 || kind == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i & 7]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]));
    else if (kind == 3 || kind == 7) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(x[(i + 3) & 7]), "v"(x[(i + 5) & 7]));
    else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i & 7]) : "v"(u[(i + 1) & 7]));
}

template <int SHAPE, int NFILL>
__global__ void k(float* out, unsigned long long* cyc) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f + i); b[i] = (__bf16)(i * 0.5f - 1.f); }
    float x[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { x[i] = -(threadIdx.x * 0.001f + i); u[i] = threadIdx.x + i; }
    f32x4 acc4[8]; f32x16 acc16[4];
    for (int j = 0; j < 8; ++j) acc4[j] = f32x4{0, 0, 0, 0};
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc16[j][r] = 0.f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
        if constexpr (SHAPE == 16) {
#pragma unroll
            for (int m = 0; m < 18; ++m) {
                acc4[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[m & 7], 0, 0, 0);
#pragma unroll
                for (int f = (m * NFILL) / 18; f < ((m + 1) * NFILL) / 18; ++f) filler(f, x, u);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 9; ++m) {
                acc16[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc16[m & 3], 0, 0, 0);
#pragma unroll
                for (int f = (m * NFILL) / 9; f < ((m + 1) * NFILL) / 9; ++f) filler(f, x, u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc4[j][0] + x[j] + (float)u[j];
    for (int j = 0; j < 4; ++j) s += acc16[j][0] + acc16[j][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int SHAPE, int NFILL>
void run(float* out, unsigned long long* cyc) {
    for (int threads : {256, 512}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<SHAPE, NFILL>), dim3(1024), dim3(threads), 0, 0, out, cyc); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<SHAPE, NFILL>), dim3(1024), dim3(threads), 0, 0, out, cyc);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double blocks = 10.0 * 1024 * (threads / 64) * REP;                  // 32x32 attention blocks computed
        const double tf = blocks * 4.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e12;
        printf("MFMA %dx%d, %2d fillers per block, %d wave/SIMD: %7.3f ms  -> attention-equivalent %7.1f TF/s\n", SHAPE, SHAPE, NFILL,
               threads / 256, ms, tf);
    }
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 1024 * 8 * 8);
    run<16, 0>(out, cyc);  run<32, 0>(out, cyc);
    run<16, 18>(out, cyc); run<32, 18>(out, cyc);
    run<16, 36>(out, cyc); run<32, 36>(out, cyc);
    run<16, 45>(out, cyc); run<32, 45>(out, cyc);
    return 0;
}
