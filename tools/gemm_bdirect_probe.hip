// Main-loop probe (round 3): what does loading the STATIC operand (weights, pre-shuffled at pack time into
// MFMA-fragment order) straight into VGPRs buy over staging both operands through LDS?
// Every variant runs the real address streams of a GEMM C[M,N] = A[M,K] * B[N,K]^T (XCD-first block remap, GROUP_M patches,
// 16x16x32 MFMAs, accumulators in AGPRs, BK = 64, LDS-DMA for what goes through LDS) but NO epilogue and no fragment-exact
// layout: results are not a GEMM, the instruction mix and the bytes moved are.  Operands are uniform random bf16 in (-1, 1).
//   V0  the product's stream kernel: 4 waves, 256x256 tile, 128x128 per wave, A and B through LDS (16 DMA + 32 ds_read_b128 per wave per k-tile)
//   V1  same tile and waves, A through LDS (8 DMA + 16 ds_read_b128), B fragments by 16 global_load_dwordx4 per wave per k-tile
//       (a rolling single buffer: fragment nj of the next k-step is requested right after its last MFMA of this one)
//   V3  128x256 tile, 4 waves as 1(M) x 4(N), 128x64 per wave (128 accumulators -> two workgroups per CU, two waves per SIMD),
//       A (16 KB per k-tile, shared by the 4 waves) through LDS, B (64 columns per wave, not shared) direct
//   V3s the same kernel held to ONE workgroup per CU by a 100 KB LDS request (what co-residency is worth)
//   hipcc --offload-arch=gfx950 -O3 -o tools/gemm_bdirect_probe tools/gemm_bdirect_probe.hip && tools/gemm_bdirect_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct P { const char* A; const char* B; float* out; int M, N, K; };

__device__ __forceinline__ void dma16(unsigned voff, const char* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ const char* uptr(const char* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void mfma16(f32x4& c, u32x4 a, u32x4 b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
// register-destination load hidden from hipcc's vmcnt bookkeeping (guide 5.7 item 1): "+v" keeps the value in ONE register
// for its whole life; the consumer side is gwait<N>() naming the same register
template <int OFF>
__device__ __forceinline__ void gload16(u32x4& d, unsigned voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(d) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void gwait(u32x4& d) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(d) : "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wg_barrier() {
    __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
}

template <int BM, int BN, int GROUP_M>
__device__ __forceinline__ void tile_of_block(const P& p, int& m0, int& n0) {
    const int tiles_m = p.M / BM, tiles_n = p.N / BN, nblk = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nblk >> 3, r = nblk & 7;
    const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int in_group = GROUP_M * tiles_n, first_m = (pid / in_group) * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    m0 = (first_m + (pid % in_group) % gsz) * BM; n0 = ((pid % in_group) / gsz) * BN;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// VMEM requests younger than weight request G_nj of the PREVIOUS k-step when MFMA group nj of the current one starts.
// Order inside a k-step: D_j after MFMA DSTRIDE*j + MI/2-1 (k-step 1 of a tile only), G_nj after MFMA MI*nj + MI-1.
template <int MI, int NJ, int AP>
constexpr int younger_than_prev_g(int nj, bool cur_has_dma) {
    const int dstride = MI * NJ / AP;
    int c = (NJ - 1 - nj) + nj;
    for (int j = 0; j < AP; ++j) {
        const int pos = dstride * j + MI / 2 - 1;
        if (!cur_has_dma && pos > MI * nj + MI - 1) ++c;     // the previous k-step carried the DMAs
        if (cur_has_dma && pos < MI * nj) ++c;
    }
    return c;
}
using Yes = std::integral_constant<bool, true>;
using No = std::integral_constant<bool, false>;

// ---------------------------------------------------------------- V0: both operands through LDS (the product's main loop)
__global__ void __launch_bounds__(256) k_v0(P p) {
    constexpr int STAGE = 65536;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1, lrow = lane >> 3, lchunk = (lane & 7) ^ lrow, li = lane & 15, lq = lane >> 4;
    int m0, n0; tile_of_block<256, 256, 4>(p, m0, n0);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
    unsigned soff[16];
    for (int j = 0; j < 16; ++j) soff[j] = (unsigned)((wave + 4 * (j & 7)) * 8 + lrow) * (unsigned)p.K * 2u + lchunk * 16;
    const char* ag = p.A + (size_t)m0 * p.K * 2; const char* bg = p.B + (size_t)n0 * p.K * 2;
    auto dma = [&](int j, int tile) {
        dma16(soff[j], uptr((j < 8 ? ag : bg) + (size_t)tile * 128), lds0 + (tile & 1) * STAGE + (j >> 3) * 32768 + (wave + 4 * (j & 7)) * 1024);
    };
    const int nk = p.K / 64;
    for (int j = 0; j < 16; ++j) dma(j, 0);
    for (int j = 0; j < 16; ++j) dma(j, 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    wg_barrier();
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
    for (int t = 0; t < nk; ++t) {
        const int so = (t & 1) * STAGE, sn = ((t + 1) & 1) * STAGE;
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            mfma16(acc[m & 7][m >> 3], F[0][8 + (m >> 3)], F[0][m & 7]);
            if (m % 3 == 1 && m / 3 < 16) rd(F[1][m / 3], m / 3, xb[1] + so, wb[1] + so);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        wg_barrier();
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            mfma16(acc[m & 7][m >> 3], F[1][8 + (m >> 3)], F[1][m & 7]);
            if (m % 3 == 1 && m / 3 < 16) rd(F[0][m / 3], m / 3, xb[0] + sn, wb[0] + sn);
            if (m % 3 == 2 && m / 3 < 16) dma(m / 3, (t + 2 < nk) ? t + 2 : t);      // tail: harmless re-stage (probe only)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { asm volatile("" : "+a"(acc[i][j])); sum += acc[i][j][0] + acc[i][j][3]; }
    p.out[blockIdx.x * 256 + threadIdx.x] = sum;
}

// ---------------------------------------------------------------- direct-B kernels
// MI x NJ MFMA tiles per wave (16 rows x 16 columns each), nj outer / mi inner: weight fragment nj lives for MI consecutive
// MFMAs and is then re-requested for the next k-step (rolling buffer); activation fragments are double-buffered from LDS.
// WAVES_M x WAVES_N waves; A tile = WAVES_M * MI * 16 rows through LDS; packed B = [N / (16 NJ)][K / 32][NJ][64 lanes][16 B].
template <int MI, int NJ, int WAVES_M, int WAVES_N, int GROUP_M>
__device__ __forceinline__ void direct_b_body(const P& p) {
    constexpr int BM = WAVES_M * MI * 16, BN = WAVES_N * NJ * 16;
    constexpr int A_BYTES = BM * 128, AP = BM / 8 / 4;                // DMA pieces per wave per k-tile
    constexpr int NM = MI * NJ;                                       // MFMAs per k-step
    static_assert(WAVES_M * WAVES_N == 4 && NM % AP == 0 && NM >= 2 * MI, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N, lrow = lane >> 3, lchunk = (lane & 7) ^ lrow, li = lane & 15, lq = lane >> 4;
    int m0, n0; tile_of_block<BM, BN, GROUP_M>(p, m0, n0);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)smem));
    unsigned soff[AP];
    for (int j = 0; j < AP; ++j) soff[j] = (unsigned)((wave + 4 * j) * 8 + lrow) * (unsigned)p.K * 2u + lchunk * 16;
    const char* ag = p.A + (size_t)m0 * p.K * 2;
    auto dma = [&](int j, int tile) { dma16(soff[j], uptr(ag + (size_t)tile * 128), lds0 + (tile & 1) * A_BYTES + (wave + 4 * j) * 1024); };
    const int nk = p.K / 64;
    // this wave's weight stream: NJ KiB per k-step, contiguous over k-steps
    const char* bw = uptr(p.B + ((size_t)((n0 / 16 / NJ) + wn) * (p.K / 32)) * (NJ * 1024));
    const unsigned boff = lane * 16;
    u32x4 FB[NJ];
    for (int j = 0; j < NJ; ++j) FB[j] = u32x4{0, 0, 0, 0};
    auto gl = [&](auto jc, int kstep) {                               // fragment jc of k-step `kstep`
        constexpr int j = decltype(jc)::value;
        gload16<(j & 3) * 1024>(FB[j], boff, uptr(bw + (size_t)kstep * (NJ * 1024) + (j >> 2) * 4096));
    };
    for (int j = 0; j < AP; ++j) dma(j, 0);
    for (int j = 0; j < AP; ++j) dma(j, 1);
    static_for<0, NJ>([&](auto jc) { gl(jc, 0); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // prologue only: the counted waits below assume steady state
    wg_barrier();
    int xb[2];
    for (int ks = 0; ks < 2; ++ks) xb[ks] = (wm * MI * 16 + li) * 128 + (((ks * 4 + lq) ^ (lane & 7)) << 4);
    f32x4 acc[MI][NJ];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    u32x4 FA[2][MI];
    auto rd = [&](u32x4& d, int i, int xbase) { d = *(const u32x4*)(smem + xbase + i * 2048); };
    for (int i = 0; i < MI; ++i) rd(FA[0][i], i, xb[0]);
    // one k-step.  DMAS: this is k-step 1 of a tile and refills the stage with tile t+2.  PREV_DMAS: the previous k-step did.
    // VMEM order inside a k-step: (D_j after MFMA NM/AP*j + MI/2 - 1), G_nj after MFMA MI*nj + MI-1.  With AP == NJ pieces the
    // sequence is D0 G0 D1 G1 ...; the wait in front of MFMA group nj leaves exactly the younger requests outstanding.
    auto kstep = [&](u32x4 (&cur)[MI], u32x4 (&nxt)[MI], int xbase_next, auto dmas, int dma_tile, int next_kstep) {
        constexpr bool D = decltype(dmas)::value;
        constexpr int DSTRIDE = NM / AP;
        static_for<0, NJ>([&](auto njc) {
            constexpr int nj = decltype(njc)::value;
            gwait<younger_than_prev_g<MI, NJ, AP>(nj, D)>(FB[nj]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = MI * nj + mi;
                mfma16(acc[mi][nj], FB[nj], cur[mi]);
                if (m % NJ == 1) rd(nxt[m / NJ], m / NJ, xbase_next);
                if constexpr (D) { if (m % DSTRIDE == MI / 2 - 1) dma(m / DSTRIDE, dma_tile); }
                if (mi == MI - 1) gl(njc, next_kstep);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    for (int t = 0; t < nk; ++t) {
        const int so = (t & 1) * A_BYTES, sn = ((t + 1) & 1) * A_BYTES;
        const int last = nk - 1;
        // k-step 0 (the previous k-step carried the refill DMAs, except for the very first tile -- waits are then merely conservative)
        kstep(FA[0], FA[1], xb[1] + so, No{}, 0, min(2 * t + 1, 2 * last + 1));
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NJ) : "memory");          // tile t+1 landed (only this k-step's NJ weight requests are younger)
        wg_barrier();
        kstep(FA[1], FA[0], xb[0] + sn, Yes{}, min(t + 2, last), min(2 * t + 2, 2 * last + 1));
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) { asm volatile("" : "+a"(acc[i][j])); sum += acc[i][j][0] + acc[i][j][3]; }
    for (int j = 0; j < NJ; ++j) sum += __uint_as_float(FB[j][0] & 1u);
    p.out[blockIdx.x * 256 + threadIdx.x] = sum;
}
__global__ void __launch_bounds__(256) k_v1(P p) { direct_b_body<8, 8, 2, 2, 4>(p); }
__global__ void __launch_bounds__(256, 2) k_v3(P p) { direct_b_body<8, 4, 1, 4, 8>(p); }

template <typename Kern>
double run(Kern kern, P p, int bm, int bn, int lds, int launches) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int wgs = (p.M / bm) * (p.N / bn);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, 0, p);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, 0, p);
    hipEventRecord(e1); hipDeviceSynchronize();
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) { printf("launch error: %s\n", hipGetErrorString(err)); return 0; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * p.M * (double)p.N * p.K * launches / (ms * 1e-3) / 1e12;
}

int main() {
    const int M = 11520;                                     // 45 x 256 = 90 x 128 (the tower's half batch is 11540 rows)
    const size_t a_bytes = (size_t)M * 4096 * 2, b_bytes = (size_t)4096 * 4096 * 2;
    std::vector<unsigned short> h((a_bytes + b_bytes) / 2);
    unsigned s = 12345u;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        const float f = ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f;
        unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16);
    }
    char* buf; float* out;
    hipMalloc(&buf, a_bytes + b_bytes); hipMalloc(&out, 4096 * 256 * 4);
    hipMemcpy(buf, h.data(), a_bytes + b_bytes, hipMemcpyHostToDevice);
    struct Shape { const char* name; int N, K; } shapes[] = {{"fc1 N4096 K1024", 4096, 1024}, {"qkv N3072 K1024", 3072, 1024},
                                                            {"fc2 N1024 K4096", 1024, 4096}, {"out N1024 K1024", 1024, 1024}, {"sq  N4096 K4096", 4096, 4096}};
    for (auto& sh : shapes) {
        P p{buf, buf + a_bytes, out, M, sh.N, sh.K};
        for (int round = 0; round < 2; ++round) {
            const double v0 = run(k_v0, p, 256, 256, 131072, 20);
            const double v1 = run(k_v1, p, 256, 256, 65536, 20);
            const double v3 = run(k_v3, p, 128, 256, 32768, 20);
            const double v3s = run(k_v3, p, 128, 256, 100 * 1024, 20);
            printf("%s M%d: V0 (LDS both) %7.1f | V1 (B direct, 256x256) %7.1f | V3 (B direct, 128x256, 2 WG/CU) %7.1f | V3 one WG/CU %7.1f TF/s\n",
                   sh.name, M, v0, v1, v3, v3s);
        }
    }
    return 0;
}
