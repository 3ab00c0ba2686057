// Cost of LDS reads and LDS-DMA issues inside a dense MFMA stream, one wave per SIMD (the regime of a 4-wave,
// 128x128-per-wave GEMM) and two waves per SIMD.  Each rep: 32 MFMAs (32 independent accumulators) with ND
// global_load_lds (SGPR-base form, 1 KiB each) and NR ds_read_b128 spread evenly.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define REP 256
template <int ND, int NR>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, const char* src) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 acc[32];
    for (int j = 0; j < 32; ++j) acc[j] = f32x4{0, 0, 0, 0};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned voff = lane * 16;
    const char* sbase = src + (size_t)(blockIdx.x * 8 + wave) * 65536;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + wave * 16384);
    const unsigned raddr = lds0 + lane * 16;
    u32x4 r[4] = {};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < REP; ++rep) {
        const unsigned long long sbv = (unsigned long long)(sbase + (rep & 31) * 2048);
        const unsigned sblo = __builtin_amdgcn_readfirstlane((unsigned)sbv), sbhi = __builtin_amdgcn_readfirstlane((unsigned)(sbv >> 32));
        const char* sb = (const char*)(((unsigned long long)sbhi << 32) | sblo);
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m], 0, 0, 0);
            if (ND > 0 && (m % (32 / (ND > 0 ? ND : 1))) == 0) {
                const unsigned la = lds0 + ((m * ND / 32) & 7) * 1024 + 8192;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sb), "s"(la) : "memory");
            }
            if (NR > 0 && (m % (32 / (NR > 0 ? NR : 1))) == 1) {
                asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(r[m & 3]) : "v"(raddr + ((m & 7) << 10)));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ND > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (NR > 0) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int j = 0; j < 32; ++j) s += acc[j][0];
    s += (float)(r[0][0] + r[1][1] + r[2][2] + r[3][3]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int ND, int NR>
void run(float* out, unsigned long long* cyc, const char* src, double* base) {
    for (int threads : {256, 512}) {
        auto kern = k<ND, NR>;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 131072, 0, out, cyc, src); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 131072, 0, out, cyc, src); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const int wps = threads / 256;
        const double ns_per_mfma = ms * 1e6 / (REP * 32.0 * wps);         // per SIMD
        double& b0 = base[wps - 1];
        if (ND == 0 && NR == 0) b0 = ns_per_mfma;
        printf("DMA %2d + ds_read %2d per 32 MFMAs, %d wave/SIMD: %6.2f ns per MFMA slot (x%.2f of bare; %.0f TF/s-equivalent)\n", ND, NR, wps,
               ns_per_mfma, ns_per_mfma / b0, 16384.0 * 1024 / ns_per_mfma / 1e3);
    }
}
int main() {
    float* out; unsigned long long* cyc; char* src;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8); hipMalloc(&src, (size_t)256 * 8 * 65536 + (1 << 20)); hipMemset(src, 1, (size_t)256 * 8 * 65536);
    double base[2] = {1, 1};
    run<0, 0>(out, cyc, src, base);
    run<4, 0>(out, cyc, src, base); run<8, 0>(out, cyc, src, base);
    run<0, 8>(out, cyc, src, base); run<0, 16>(out, cyc, src, base);
    run<4, 8>(out, cyc, src, base); run<8, 16>(out, cyc, src, base);
    return 0;
}
