// ABLATIONS of the stream-GEMM main loop (copy of mfma_shape_mix.hip: what the k-tile's cycles are made of).
// 16x16x32 vs 32x32x16 MFMA inside the stream-GEMM main loop (four waves, one per SIMD, 128x128 per wave, 256x256x64
// k-tiles, LDS-DMA double buffer, one barrier per k-tile), on RANDOM operands, prologue / epilogue excluded: which
// instruction shape sustains more flops at this chip's power cap?  Both variants move the same LDS bytes (32 ds_read_b128 and
// 16 DMA pieces per wave per k-tile) -- the 32x32 shape halves the matrix instructions and their register-file operand reads.
// Results are not a GEMM (fragments are whatever the access pattern delivers); sums are written out to keep everything live.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_shape_mix tools/mfma_shape_mix.hip && tools/mfma_shape_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void dma16(unsigned voff, const char* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ const char* uptr(const char* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void mfma16(f32x4& c, u32x4 a, u32x4 b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma32(f32x16& c, u32x4 a, u32x4 b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }

constexpr int STAGE = 65536, LDA = 2048;        // bytes; operand rows are 1024 bf16 apart in memory

template <int SHAPE, int ABL>
__global__ void __launch_bounds__(256) k(const char* src, float* out, int ktiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 3;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
    const char* gbase = src + (size_t)(blockIdx.x & 31) * (2u << 20);      // 32 regions of 2 MiB (A rows then B rows)
    unsigned soff[16];
    for (int j = 0; j < 16; ++j) {
        const int row = (wave + 4 * (j & 7)) * 8 + lrow + (j >> 3) * 256;
        const int key = SHAPE == 16 ? (lrow & 7) : ((row >> 1) & 7);
        soff[j] = (unsigned)row * LDA + (((lane & 7) ^ key) << 4);
    }
    auto dma = [&](int j, int tile) {
        dma16(soff[j], uptr(gbase + (size_t)(tile & 15) * 128), lds0 + (tile & 1) * STAGE + (j >> 3) * 32768 + (wave + 4 * (j & 7)) * 1024);
    };
    for (int j = 0; j < 16; ++j) dma(j, 0);
    for (int j = 0; j < 16; ++j) dma(j, 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();
    float sum = 0.f;
    if constexpr (SHAPE == 16) {
        const int li = lane & 15, lq = lane >> 4;
        int xb[2], wb[2];
        for (int ks = 0; ks < 2; ++ks) {
            const int sw = ((ks * 4 + lq) ^ (lane & 7)) << 4;
            xb[ks] = (wm * 128 + li) * 128 + sw; wb[ks] = 32768 + (wn * 128 + li) * 128 + sw;
        }
        f32x4 acc[8][8];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        u32x4 F[2][16];
        auto rd = [&](u32x4& d, int i, int xbase, int wbase) { d = *(const u32x4*)(smem + (i < 8 ? xbase + i * 2048 : wbase + (i - 8) * 2048)); };
        for (int i = 0; i < 16; ++i) rd(F[0][i], i, xb[0], wb[0]);
        for (int t = 0; t < ktiles; ++t) {
            const int so = (t & 1) * STAGE, sn = ((t + 1) & 1) * STAGE;
#pragma unroll
            for (int m = 0; m < 64; ++m) {
                mfma16(acc[m & 7][m >> 3], F[0][8 + (m >> 3)], F[0][m & 7]);
                if (!(ABL & 2) && m % 3 == 1 && m / 3 < 16) rd(F[1][m / 3], m / 3, xb[1] + so, wb[1] + so);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(ABL & 4)) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
            } else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int m = 0; m < 64; ++m) {
                mfma16(acc[m & 7][m >> 3], F[1][8 + (m >> 3)], F[1][m & 7]);
                if (!(ABL & 2) && m % 3 == 1 && m / 3 < 16) rd(F[0][m / 3], m / 3, xb[0] + sn, wb[0] + sn);
                if (!(ABL & 1) && m % 3 == 2 && m / 3 < 16) dma(m / 3, t + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { asm volatile("" : "+a"(acc[i][j])); sum += acc[i][j][0] + acc[i][j][3]; }
    } else {
        const int l31 = lane & 31, lh = lane >> 5;
        int xb[4], wb[4];
        for (int ks = 0; ks < 4; ++ks) {
            const int sw = ((ks * 2 + lh) ^ ((lane >> 1) & 7)) << 4;
            xb[ks] = (wm * 128 + l31) * 128 + sw; wb[ks] = 32768 + (wn * 128 + l31) * 128 + sw;
        }
        f32x16 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        u32x4 F[2][8];
        auto rd = [&](u32x4& d, int i, int xbase, int wbase) { d = *(const u32x4*)(smem + (i < 4 ? xbase + i * 4096 : wbase + (i - 4) * 4096)); };
        for (int i = 0; i < 8; ++i) rd(F[0][i], i, xb[0], wb[0]);
        for (int t = 0; t < ktiles; ++t) {
            const int so = (t & 1) * STAGE, sn = ((t + 1) & 1) * STAGE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int nks = (ks + 1) & 3, nso = ks == 3 ? sn : so;
                if (ks == 3) {       // every wave's reads of tile t have returned, its tile t+1 pieces have landed: publish / retire
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    mfma32(acc[m & 3][m >> 2], F[ks & 1][4 + (m >> 2)], F[ks & 1][m & 3]);
                    if ((m & 1) == 0) rd(F[(ks + 1) & 1][m >> 1], m >> 1, xb[nks] + nso, wb[nks] + nso);
                    if (ks == 3) dma(m, t + 2);                       // refill this stage with tile t+2
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { asm volatile("" : "+a"(acc[i][j])); sum += acc[i][j][0] + acc[i][j][15]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int SHAPE, int ABL>
double run(const char* src, float* out, int wgs, int ktiles, int launches) {
    auto kern = k<SHAPE, ABL>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 2 * STAGE, 0, src, out, ktiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 2 * STAGE, 0, src, out, ktiles);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * 256 * 256 * 64 * ktiles * (double)wgs * launches / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t bytes = (size_t)33 * (2u << 20);
    std::vector<unsigned short> h(bytes / 2);
    unsigned s = 12345u;
    for (auto& v : h) {                  // uniform bf16 in (-1, 1): random sign, exponent and mantissa bits
        s = s * 1664525u + 1013904223u;
        const float f = ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f;
        unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16);
    }
    char* src; float* out;
    hipMalloc(&src, bytes); hipMalloc(&out, 4096 * 256 * 4);
    hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
    for (int ktiles : {16, 64}) {
        for (int round = 0; round < 2; ++round) {
            printf("16x16x32 main loop, k-tiles %3d: full %7.1f | no DMA %7.1f | no ds_read %7.1f | no DMA, no ds_read %7.1f | no barrier %7.1f | MFMA only %7.1f TF/s\n", ktiles,
                   run<16, 0>(src, out, 1024, ktiles, 40), run<16, 1>(src, out, 1024, ktiles, 40), run<16, 2>(src, out, 1024, ktiles, 40),
                   run<16, 3>(src, out, 1024, ktiles, 40), run<16, 4>(src, out, 1024, ktiles, 40), run<16, 7>(src, out, 1024, ktiles, 40));
        }
    }
    return 0;
}
