// What does one wave per SIMD really overlap?  An 11-slot iteration shaped like attn32's (5 MFMAs accumulating one VGPR score
// tile alternating with 6 MFMAs on three AGPR tiles; 16 v_exp_f32, 8 v_cvt_pk, 8 v_max3 in the gaps), ONE workgroup of 4 waves
// per CU (150 KiB of LDS requested), cycles per iteration from s_memtime.  Variants switch parts of the stream off.
//   hipcc --offload-arch=gfx950 -O3 -o tools/attn32_slot_probe tools/attn32_slot_probe.hip && tools/attn32_slot_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define REP 512
#define MFQ0 "v_mfma_f32_32x32x16_bf16 %[sv], %[a], %[b], 0\n\t"
#define MFQ "v_mfma_f32_32x32x16_bf16 %[sv], %[a], %[b], %[sv]\n\t"
#define MFP(t) "v_mfma_f32_32x32x16_bf16 %[" #t "], %[a], %[b], %[" #t "]\n\t"
#define EX(d, s) "v_exp_f32 %[" #d "], %[" #s "]\n\t"
#define MV(d, s) "v_mov_b32 %[" #d "], %[" #s "]\n\t"
#define CV(d, x, y) "v_cvt_pk_bf16_f32 %[" #d "], %[" #x "], %[" #y "]\n\t"
#define MX(d, x, y, z) "v_max3_f32 %[" #d "], %[" #x "], %[" #y "], %[" #z "]\n\t"
// MODE bits: 1 = MFMAs, 2 = exps, 4 = cvt + max, 8 = exps as v_mov, 16 = score-tile MFMAs on independent tiles (no chain)
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc) {
    extern __shared__ char smem[];
    u32x4 a = {threadIdx.x * 3u + 1, 0x3f803f80u, 0x3f003f00u, threadIdx.x}, b = {0x3f803f80u, threadIdx.x * 7u, 0x3e803e80u, 1};
    f32x16 sv, o0, o1, os, s2;
    float x[16], e[16], m0, m1, m2, m3;
    unsigned p[8];
    for (int r = 0; r < 16; ++r) { sv[r] = 0; o0[r] = 0; o1[r] = 0; os[r] = 0; s2[r] = 0; x[r] = -(threadIdx.x * 0.001f + r); e[r] = 0; }
    for (int r = 0; r < 8; ++r) p[r] = 0;
    m0 = m1 = m2 = m3 = 0;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(sv), "+a"(o0), "+a"(o1), "+a"(os), "+v"(s2));
    if (smem[threadIdx.x] == 77) out[0] = 1.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define OUTS [sv] "+v"(sv), [s2] "+v"(s2), [o0] "+a"(o0), [o1] "+a"(o1), [os] "+a"(os), [e0] "+v"(e[0]), [e1] "+v"(e[1]), [e2] "+v"(e[2]), [e3] "+v"(e[3]), \
             [p0] "+v"(p[0]), [p1] "+v"(p[1]), [m0] "+v"(m0), [m1] "+v"(m1)
#define INS [a] "v"(a), [b] "v"(b), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5])
    constexpr bool MF = MODE & 1, EXPS = MODE & 2, OTH = MODE & 4, ASMOV = MODE & 8, NOCHAIN = MODE & 16;
#define E2(d0, s0, d1, s1) (EXPS ? (ASMOV ? 2 : 1) : 0)
    for (int r = 0; r < REP; ++r) {
        // five pair statements + one single, like attn32: [QK mfma] exp exp [PV mfma] exp cvt max max
#define PAIRSTMT(QSTR, PSTR)                                                                                                     \
        if (MF && EXPS && !ASMOV && OTH) asm volatile(QSTR EX(e0, x0) EX(e1, x1) PSTR EX(e2, x2) CV(p0, e0, e1) MX(m0, x3, x4, x5) MX(m1, x0, x1, x2) : OUTS : INS); \
        else if (MF && EXPS && ASMOV && OTH) asm volatile(QSTR MV(e0, x0) MV(e1, x1) PSTR MV(e2, x2) CV(p0, e0, e1) MX(m0, x3, x4, x5) MX(m1, x0, x1, x2) : OUTS : INS); \
        else if (MF && EXPS && !OTH) asm volatile(QSTR EX(e0, x0) EX(e1, x1) PSTR EX(e2, x2) : OUTS : INS);                       \
        else if (MF && !EXPS && OTH) asm volatile(QSTR PSTR CV(p0, e0, e1) MX(m0, x3, x4, x5) MX(m1, x0, x1, x2) : OUTS : INS);    \
        else if (MF) asm volatile(QSTR PSTR : OUTS : INS);                                                                        \
        else if (EXPS && OTH) asm volatile(EX(e0, x0) EX(e1, x1) EX(e2, x2) CV(p0, e0, e1) MX(m0, x3, x4, x5) MX(m1, x0, x1, x2) : OUTS : INS); \
        else if (EXPS) asm volatile(EX(e0, x0) EX(e1, x1) EX(e2, x2) : OUTS : INS);                                               \
        else asm volatile(CV(p0, e0, e1) MX(m0, x3, x4, x5) MX(m1, x0, x1, x2) : OUTS : INS);
        if (NOCHAIN) {
            PAIRSTMT(MFQ, MFP(o0)) PAIRSTMT(MFP(s2), MFP(o1)) PAIRSTMT(MFQ, MFP(os)) PAIRSTMT(MFP(s2), MFP(o0)) PAIRSTMT(MFQ, MFP(o1))
        } else {
            PAIRSTMT(MFQ0, MFP(o0)) PAIRSTMT(MFQ, MFP(o1)) PAIRSTMT(MFQ, MFP(os)) PAIRSTMT(MFQ, MFP(o0)) PAIRSTMT(MFQ, MFP(o1))
        }
        if (MF && EXPS) asm volatile(MFP(os) EX(e3, x3) CV(p1, e2, e3) CV(p0, e0, e1) : OUTS : INS);
        else if (MF) asm volatile(MFP(os) : OUTS : INS);
        else if (EXPS) asm volatile(EX(e3, x3) CV(p1, e2, e3) CV(p0, e0, e1) : OUTS : INS);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(sv), "+a"(o0), "+a"(o1), "+a"(os), "+v"(s2));
    float s = sv[0] + o0[0] + o1[1] + os[2] + s2[3] + e[0] + e[1] + e[2] + e[3] + m0 + m1 + (float)p[0] + (float)p[1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE>
void run(float* out, unsigned long long* cyc, const char* what) {
    for (int threads : {256}) {
        hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 150 * 1024, 0, out, cyc); hipDeviceSynchronize();
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 150 * 1024, 0, out, cyc); hipDeviceSynchronize();
        unsigned long long h[256 * 8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double sum = 0; for (int i = 0; i < 256; ++i) for (int w = 0; w < threads / 64; ++w) sum += (double)h[i * 8 + w];
        printf("%-58s %7.1f cycles per iteration (11 MFMAs = 352)\n", what, sum / (256.0 * (threads / 64)) / REP);
    }
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 1024 * 8 * 8);
    run<1>(out, cyc, "MFMAs only (score chain + 3 AGPR tiles)");
    run<1 + 16>(out, cyc, "MFMAs only, no accumulate chain");
    run<2>(out, cyc, "16 exps only");
    run<4>(out, cyc, "cvt + max only (5 x 3)");
    run<6>(out, cyc, "exps + cvt + max");
    run<1 + 2>(out, cyc, "MFMAs + 16 exps");
    run<1 + 4>(out, cyc, "MFMAs + cvt/max");
    run<1 + 2 + 4>(out, cyc, "MFMAs + exps + cvt/max (the attn32 slot)");
    run<1 + 2 + 4 + 8>(out, cyc, "same, exps replaced by v_mov");
    run<1 + 2 + 4 + 16>(out, cyc, "same as the slot, no accumulate chain");
    return 0;
}
