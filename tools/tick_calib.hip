// What does s_memtime count?  Compare it with s_memrealtime (constant 100 MHz) and with the MFMA issue rate
// (v_mfma_f32_16x16x32_bf16 = 16 shader cycles back to back) in a short and in a long MFMA-saturated kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* st, int reps) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = f32x4{0, 0, 0, 0};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int j = 0; j < 16; ++j) s += acc[j][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { st[blockIdx.x * 2] = t1 - t0; st[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main() {
    float* out; unsigned long long* st; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 16);
    for (int reps : {64, 1024, 16384, 262144}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, st, reps); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[512]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
            double mt = 0, rt = 0; for (int i = 0; i < 256; ++i) { mt += h[2 * i]; rt += h[2 * i + 1]; } mt /= 256; rt /= 256;
            const double mfma_per_simd = 2.0 * 16 * reps;        // 2 waves per SIMD
            const double secs = rt / 100e6;
            printf("reps %7d: wall %9.3f ms | memtime %12.0f ticks, memrealtime %10.0f (=%9.3f ms) -> memtime %.1f MHz | %.2f ticks/MFMA, %.2f ns/MFMA -> shader clock %.0f MHz if 16 cyc/MFMA | %.0f TF/s\n",
                   reps, ms, mt, rt, secs * 1e3, mt / secs / 1e6, mt / mfma_per_simd, secs * 1e9 / mfma_per_simd, 16.0 / (secs * 1e9 / mfma_per_simd) * 1e3,
                   mfma_per_simd * 1024 * 16384 / secs / 1e12);
        }
    }
    return 0;
}
