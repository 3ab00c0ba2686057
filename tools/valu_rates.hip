// Issue-rate microbenchmark for the VALU ops of the softmax stream (diagnostic). Cycles per wave64
// instruction with N independent chains, 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
template <int OP> __device__ __forceinline__ void body(float (&x)[16], float c) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if constexpr (OP == 0) x[i] = fmaf(x[i], c, 0.5f);
        else if constexpr (OP == 1) x[i] = __builtin_amdgcn_exp2f(x[i]);
        else if constexpr (OP == 2) x[i] = __builtin_fmaxf(__builtin_fmaxf(x[i], c), x[(i + 1) & 15]);
        else if constexpr (OP == 3) x[i] = x[i] + c;
        else if constexpr (OP == 5) x[i] = __builtin_amdgcn_rcpf(x[i]);
    }
}
template <int OP> __global__ void k(float* out, unsigned long long* cyc, float c) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 y[8];
    for (int i = 0; i < 8; ++i) y[i] = f2{x[2 * i], x[2 * i + 1]};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
        if constexpr (OP == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = y[i] * f2{c, c};       // v_pk_mul_f32
        } else if constexpr (OP == 6) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { unsigned u; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(x[2*i]), "v"(x[2*i+1])); x[2*i] += __uint_as_float(u); }
        } else body<OP>(x, c);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i]; for (int i = 0; i < 8; ++i) s += y[i][0] + y[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int OP> void run(const char* name, int n_instr_per_rep) {
    float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8192 * 8);
    for (int waves = 4; waves <= 16; waves *= 2) {        // waves per CU (block of waves*64 threads, 1 block/CU)
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(waves * 64), 0, 0, out, cyc, 1.0001f);
        hipDeviceSynchronize();
        unsigned long long h[4096]; hipMemcpy(h, cyc, 256 * waves * 8, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256 * waves; ++i) m += h[i]; m /= 256 * waves;
        printf("%-12s waves/SIMD=%d: %7.1f cycles per wave-instruction (per wave), %6.2f per SIMD-issue\n", name, waves / 4,
               m / (REP * n_instr_per_rep), m / (REP * n_instr_per_rep) / (waves / 4));
    }
}
int main() {
    run<0>("v_fma_f32", 16); run<3>("v_add_f32", 16); run<1>("v_exp_f32", 16); run<5>("v_rcp_f32", 16);
    run<2>("v_max3_f32", 16); run<4>("v_pk_mul_f32", 8); run<6>("cvt_pk+add", 16);
    return 0;
}
