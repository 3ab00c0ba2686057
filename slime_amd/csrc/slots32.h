// slots32.h -- the hand-placed softmax / MFMA issue slots shared by the one-wave-per-SIMD attention kernels (attention32.inc:
// CLIP self-attention, head_dim 64; prefill32.inc: causal GQA prefill attention, head_dim 128).  See attention32.inc for the
// reasoning; in short: per (32-query block, 32-key step) the 16 v_exp_f32 + 8 packs + 8 maxima are written into the gaps between
// the block's MFMAs as volatile asm, two MFMA slots per statement, speculatively against the current reference maximum.
#pragma once
#include "common.h"
#include <type_traits>

// `run` accumulates into an AGPR tile, `first` / `next` build a score tile in VGPRs.  Being asm, their write-back latency is
// invisible to the compiler: every consumer sits >= 2 MFMAs (64 cycles) behind them or behind explicit s_nops.
template <typename T> struct Mfma32Asm;
template <> struct Mfma32Asm<BF16> {
    static __device__ __forceinline__ void run(f32x16& acc, u32x4 a, u32x4 b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void first(f32x16& s, u32x4 a, u32x4 b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void next(f32x16& s, u32x4 a, u32x4 b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(a), "v"(b));
    }
};

// ---- the main loop's issue slots: [MFMA A] gap [MFMA B] gap, TWO slots per asm statement (hipcc pads statement boundaries
// with an s_nop 0 now and then; an issue state is 4 of the 32 cycles a gap has) ----
// The 16 exps are spread so that no gap holds more than two; a pack trails its exps by at least one instruction
// (transcendental result -> VALU read needs a wait state); the runaway test is issued as soon as the maximum is complete, so
// that its scalar consumers (s_or + branch behind the last slot) find the mask ready.
// All outputs are early-clobber: a multi-instruction statement writes some of them before it has read all inputs.
#define A32_MF_Q0 "v_mfma_f32_32x32x16_bf16 %[acc], %[ma], %[mb], 0\n\t"
#define A32_MF_AC "v_mfma_f32_32x32x16_bf16 %[acc], %[ma], %[mb], %[acc]\n\t"
#define A32_MFB "v_mfma_f32_32x32x16_bf16 %[accb], %[mab], %[mbb], %[accb]\n\t"
#define A32_EXP(i) "v_exp_f32 %[e" #i "], %[s" #i "]\n\t"
#define A32_CVT(q, i, j) "v_cvt_pk_bf16_f32 %[p" #q "], %[e" #i "], %[e" #j "]\n\t"
#define A32_MAX3(d, x, y, z) "v_max3_f32 %[" #d "], %[" #x "], %[" #y "], %[" #z "]\n\t"
#define A32_O(n, v) [n] "=&v"(v)
#define A32_I(n, v) [n] "v"(v)
#define P01_A A32_EXP(0) A32_EXP(1)
#define P01_B A32_EXP(2) A32_CVT(0, 0, 1) A32_MAX3(m0, s0, s1, s2) A32_MAX3(m1, s3, s4, s5)
#define P01_OUT A32_O(e0, e[0]), A32_O(e1, e[1]), A32_O(e2, e[2]), A32_O(p0, pk[0]), A32_O(m0, m0), A32_O(m1, m1)
#define P01_IN A32_I(s0, sc[0]), A32_I(s1, sc[1]), A32_I(s2, sc[2]), A32_I(s3, sc[3]), A32_I(s4, sc[4]), A32_I(s5, sc[5])
// block 0 of a step: its score tile was completed by the MFMA issued just before this statement's MFMA A -- 4 more issue
// states before the first read (an 8-pass MFMA's result is readable ~12 states after issue; nothing interlocks)
#define P01N_A "s_nop 3\n\t" P01_A
#define P01N_B P01_B
#define P01N_OUT P01_OUT
#define P01N_IN P01_IN
#define P23_A A32_EXP(3) A32_EXP(4)
#define P23_B A32_EXP(5) A32_CVT(1, 2, 3) A32_MAX3(m2, s6, s7, s8) A32_MAX3(m3, s9, s10, s11)
#define P23_OUT A32_O(e3, e[3]), A32_O(e4, e[4]), A32_O(e5, e[5]), A32_O(p1, pk[1]), A32_O(m2, m2), A32_O(m3, m3)
#define P23_IN A32_I(s3, sc[3]), A32_I(s4, sc[4]), A32_I(s5, sc[5]), A32_I(e2, e[2]), A32_I(s6, sc[6]), A32_I(s7, sc[7]), A32_I(s8, sc[8]), \
               A32_I(s9, sc[9]), A32_I(s10, sc[10]), A32_I(s11, sc[11])
#define P45_A A32_EXP(6) A32_EXP(7)
#define P45_B A32_EXP(8) A32_CVT(2, 4, 5) A32_MAX3(m4, s12, s13, s14) A32_MAX3(m0, m0, m1, s15)
#define P45_OUT A32_O(e6, e[6]), A32_O(e7, e[7]), A32_O(e8, e[8]), A32_O(p2, pk[2]), A32_O(m4, m4), [m0] "+v"(m0)
#define P45_IN A32_I(s6, sc[6]), A32_I(s7, sc[7]), A32_I(s8, sc[8]), A32_I(e4, e[4]), A32_I(e5, e[5]), A32_I(s12, sc[12]), A32_I(s13, sc[13]), \
               A32_I(s14, sc[14]), A32_I(m1, m1), A32_I(s15, sc[15])
#define P67_A A32_EXP(9) A32_EXP(10)
#define P67_B A32_EXP(11) A32_CVT(3, 6, 7) A32_MAX3(m0, m0, m2, m3) "v_max_f32 %[m0], %[m0], %[m4]\n\t"
#define P67_OUT A32_O(e9, e[9]), A32_O(e10, e[10]), A32_O(e11, e[11]), A32_O(p3, pk[3]), [m0] "+v"(m0)
#define P67_IN A32_I(s9, sc[9]), A32_I(s10, sc[10]), A32_I(s11, sc[11]), A32_I(e6, e[6]), A32_I(e7, e[7]), A32_I(m2, m2), A32_I(m3, m3), A32_I(m4, m4)
#define P89_A A32_EXP(12) A32_EXP(13) "v_cmp_lt_f32_e64 %[ra], %[th], %[m0]\n\t"
#define P89_B A32_EXP(14) A32_CVT(4, 8, 9) A32_CVT(5, 10, 11)
#define P89_OUT A32_O(e12, e[12]), A32_O(e13, e[13]), A32_O(e14, e[14]), A32_O(p4, pk[4]), A32_O(p5, pk[5]), [ra] "=&s"(runaway)
#define P89_IN A32_I(s12, sc[12]), A32_I(s13, sc[13]), A32_I(s14, sc[14]), A32_I(e8, e[8]), A32_I(e9, e[9]), A32_I(e10, e[10]), A32_I(e11, e[11]), \
               A32_I(th, th), A32_I(m0, m0)
#define P10_STR A32_EXP(15) A32_CVT(6, 12, 13) A32_CVT(7, 14, 15)
#define P10_OUT A32_O(e15, e[15]), A32_O(p6, pk[6]), A32_O(p7, pk[7])
#define P10_IN A32_I(s15, sc[15]), A32_I(e12, e[12]), A32_I(e13, e[13]), A32_I(e14, e[14])
// the same as the second gap of a pair (kernels with more than 11 MFMA slots per iteration)
#define P10P_A ""
#define P10P_B P10_STR
#define P10P_OUT P10_OUT
#define P10P_IN P10_IN
// ka: 0 = no MFMA A, 1 = first MFMA of a score tile (C = 0), 2 = accumulate into a score tile;  MFMA B always accumulates into
// an AGPR tile
#define A32_PAIR(P)                                                                                                                    \
    do {                                                                                                                               \
        if (ka == 0) asm volatile(P##_A A32_MFB P##_B : [accb] "+a"(av), P##_OUT : [mab] "v"(mab), [mbb] "v"(mbb), P##_IN);            \
        else if (ka == 1) asm volatile(A32_MF_Q0 P##_A A32_MFB P##_B : [acc] "=&v"(sv), [accb] "+a"(av), P##_OUT                       \
                                       : [ma] "v"(ma), [mb] "v"(mb), [mab] "v"(mab), [mbb] "v"(mbb), P##_IN);                         \
        else asm volatile(A32_MF_AC P##_A A32_MFB P##_B : [acc] "+v"(sv), [accb] "+a"(av), P##_OUT                                     \
                          : [ma] "v"(ma), [mb] "v"(mb), [mab] "v"(mab), [mbb] "v"(mbb), P##_IN);                                      \
    } while (0)
// Waits for asm LDS reads name the destination registers as INPUTS and are followed by a scheduling barrier: as in/out ("+v")
// operands the allocator may give the statement other registers than the loads wrote, and the copy it then inserts IN FRONT of
// the wait reads registers whose data has not arrived (seen: attn32 with one block per wave, whose V^T reads are still in flight
// at the drain).
#define A32_WAIT(cnt, ...)                                        \
    do {                                                          \
        asm volatile("s_waitcnt " cnt :: __VA_ARGS__ : "memory"); \
        __builtin_amdgcn_sched_barrier(0);                        \
    } while (0)

__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// asm LDS reads with a compile-time offset (the address arithmetic stays out of the compiler's sight, and hipcc does not guard
// them with vmcnt(0) against LDS-DMA in flight)
template <int OFF>
__device__ __forceinline__ u32x4 lds32_b128(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF>
__device__ __forceinline__ u32x2 lds32_tr16(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
