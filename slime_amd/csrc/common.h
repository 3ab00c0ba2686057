// common.h -- shared device helpers for libslime_hip (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "slime_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 16-bit MFMA operand types.  `bits` are carried around as raw u32 words.
struct BF16 {
    static constexpr int id = SLIME_BF16;
    static constexpr int RESID_SH = 8;      // fp32 pattern bits below the 8 that follow a bf16 significand (resid_delta / resid_join)
    static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                      __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                      __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    // half-depth form (K = 16: a lane feeds 4 consecutive k), for ragged tails that fill at most 16 of a 32-deep step
    static __device__ __forceinline__ f32x4 mfma16_k16(u32x2 a, u32x2 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b), c, 0, 0, 0);
    }
    // accumulator pinned to the AGPR file (written out: the register allocator otherwise migrates large accumulator
    // sets between the two files around loop back edges)
    static __device__ __forceinline__ void mfma16_agpr(f32x4& acc, u32x4 a, u32x4 b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    }
    // the same with the accumulator pinned to the ARCH VGPR file (kernels whose whole register budget is <= 256: the epilogue
    // then reads its results without a v_accvgpr_read per element)
    static __device__ __forceinline__ void mfma16_vgpr(f32x4& acc, u32x4 a, u32x4 b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
    // first k-step of an output tile: C = 0 (inline constant), the accumulator is written, not read
    static __device__ __forceinline__ void mfma16_agpr_init(f32x4& acc, u32x4 a, u32x4 b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
    }
    // last k-step of an output tile in a kernel that drains a second accumulator set: D = C + A B (vdst != srcC)
    static __device__ __forceinline__ void mfma16_agpr_fin(f32x4& d, const f32x4& c, u32x4 a, u32x4 b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %1" : "=a"(d) : "a"(c), "v"(a), "v"(b));
    }
    static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
        f32x2 v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));   // RNE
    }
    static __device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }
    static __device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
};
struct F16 {
    static constexpr int id = SLIME_F16;
    static constexpr int RESID_SH = 5;      // ... below the 8 that follow an fp16 significand
    static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                     __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                     __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16_k16(u32x2 a, u32x2 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, a), __builtin_bit_cast(f16x4_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mfma16_agpr(f32x4& acc, u32x4 a, u32x4 b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void mfma16_vgpr(f32x4& acc, u32x4 a, u32x4 b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
    // first k-step of an output tile: C = 0 (inline constant), the accumulator is written, not read
    static __device__ __forceinline__ void mfma16_agpr_init(f32x4& acc, u32x4 a, u32x4 b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
    }
    // last k-step of an output tile in a kernel that drains a second accumulator set: D = C + A B (vdst != srcC)
    static __device__ __forceinline__ void mfma16_agpr_fin(f32x4& d, const f32x4& c, u32x4 a, u32x4 b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %1" : "=a"(d) : "a"(c), "v"(a), "v"(b));
    }
    static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
        f32x2 v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));    // RNE
    }
    static __device__ __forceinline__ float lo(unsigned w) {
        return (float)__builtin_bit_cast(f16x2_t, w)[0];
    }
    static __device__ __forceinline__ float hi(unsigned w) {
        return (float)__builtin_bit_cast(f16x2_t, w)[1];
    }
};

template <typename T>
__device__ __forceinline__ u32x4 pack8(const float* v) {
    u32x4 r;
    r[0] = T::pack2(v[0], v[1]); r[1] = T::pack2(v[2], v[3]);
    r[2] = T::pack2(v[4], v[5]); r[3] = T::pack2(v[6], v[7]);
    return r;
}

// Measured alternative (round 4, profiles/r04_fabric_traffic.txt), OFF: results leaving with the non-temporal hint (stream / evict
// first) and read-once inputs (the fp32 residual rows) arriving with it, so that neither pushes the shared GEMM operand panels out of
// an XCD's 4 MiB L2.  It does cut the fabric reads (fc1 199 -> 156 MB per launch together with the row ownership below) -- and the
// 40-crop tower gets SLOWER, 14.9 -> 15.9 ms.  Not separated further; the candidates: a streamed result is not kept for its only
// reader, the NEXT kernel, and the epilogue's 64-byte row segments leave as partial lines instead of being merged in L2 first.
// `make` leaves it off; tools/build_variants.sh builds the A/B libraries.
#ifndef SLIME_OPT_NT
#define SLIME_OPT_NT 0
#endif
template <typename V>
__device__ __forceinline__ void st_stream(V* p, const V v) {
#if SLIME_OPT_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <typename V>
__device__ __forceinline__ V ld_stream(const V* p) {
#if SLIME_OPT_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// ---- the residual stream's lower part (ABI 7): ONE SIGNED BYTE per element -------------------------------------------------------
// The stream is kept as hi = T(h) (RNE: hi IS the next GEMM's operand) and d8 = the next 8 bits of h's fp32 pattern, stored as the
// signed distance of h's pattern from hi's in units of 2^SH (SH = 8 for bf16, 5 for fp16), rounded DOWN, and read back at the
// middle of its cell:
//     d8 = floor((pattern(h) - pattern(float(hi))) / 2^SH)  in [-128, 127],      join = pattern(float(hi)) + d8 2^SH + 2^(SH-1).
// IEEE patterns of one sign are monotonic in the magnitude, so this holds across binade boundaries and whichever way hi was rounded:
// |h - hi| <= ulp(hi) / 2 puts the distance in [-2^(SH+7), 2^(SH+7)], i.e. d8 in [-128, 128], and +128 -- an exact tie that RNE rounded
// down -- is stored as 127, whose cell centre is still within half a cell of h.  So |join - h| <= 2^(SH-1) patterns = ulp(hi) / 512 in
// EVERY case, without bias: 16 significant bits of h with bf16 halves (what the 2 x bf16 stream of ABI 5-6 guaranteed), 19 with fp16
// halves (22 before; fp32 has 24; below fp16's normal range, |h| < 2^-14, hi's spacing is fixed while the byte counts cells of its fp32
// pattern, and the stream keeps what hi keeps: 2^-25, where the 16-bit lower part bottomed out too).  Both directions are integer arithmetic on the patterns -- exact, no exponent handling, the same
// operations in every kernel that touches the stream.  6 bytes per element cross the fabric in a residual update (2 + 1 in, 2 + 1
// out) instead of 8; measured with the traffic alone changed (round 6, profiles/r06_lo8_traffic_ablation.txt): two-stream tower
// -2.1 %, stand-alone launches unchanged.
template <typename T>
__device__ __forceinline__ int resid_delta(float h, float hi_f) {
    const int d = (__float_as_int(h) - __float_as_int(hi_f)) >> T::RESID_SH;      // arithmetic shift: floor
    return min(max(d, -128), 127);
}
template <typename T>
__device__ __forceinline__ float resid_join(float hi_f, int d8) {
    return __int_as_float(__float_as_int(hi_f) + d8 * (1 << T::RESID_SH) + (1 << (T::RESID_SH - 1)));
}
__device__ __forceinline__ int sext_byte(unsigned w, int k) { return (int)(w << (24 - 8 * k)) >> 24; }   // v_bfe_i32
__device__ __forceinline__ unsigned pack_bytes(int b0, int b1, int b2, int b3) {
    return (unsigned)(b0 & 0xff) | ((unsigned)(b1 & 0xff) << 8) | ((unsigned)(b2 & 0xff) << 16) | ((unsigned)b3 << 24);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Pin a wave-uniform pointer into SGPRs (the compiler otherwise folds `uniform + lane offset + uniform` into 64-bit
// VALU adds and loses the SGPR-base addressing mode of global_load_lds).
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

// One LDS-DMA piece in the SGPR-base form (64 lanes x 16 B -> 1 KiB at LDS byte address lds_addr, lane-linear):
// hipcc materialises a 64-bit VGPR address per piece from the builtin (v_lshl_add_u64 + the vaddr form) inside loops,
// so the instruction is written out.  M0 = LDS base; one wait state between the M0 write and the load.
__device__ __forceinline__ void lds_dma16(unsigned voff, const char* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_byte_addr(const char* p) {
    return (unsigned)(size_t)((__attribute__((address_space(3))) const char*)p);
}

// ---- host side error plumbing ------------------------------------------------------------------
void slime_set_error(const char* fmt, ...);

#define SLIME_REQUIRE(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) { slime_set_error(__VA_ARGS__); return SLIME_EINVAL; } \
    } while (0)

#define SLIME_CHECK_LAUNCH(what)                                                     \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            slime_set_error("%s: %s", what, hipGetErrorString(e_));                  \
            return SLIME_ELAUNCH;                                                    \
        }                                                                            \
    } while (0)

// One-time opt-in of a kernel to > 64 KiB of dynamic LDS.  A function-local static is initialised exactly once, thread-safely
// (C++11), per kernel instantiation; one process drives one GPU, so once per process is once per device.
#define SLIME_SET_LDS_ONCE(kern, lds, what)                                                                            \
    do {                                                                                                               \
        static const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                          \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (lds));           \
        if (e_ != hipSuccess) { slime_set_error(what ": hipFuncSetAttribute: %s", hipGetErrorString(e_)); return SLIME_ELAUNCH; } \
    } while (0)

// CU count of the current device (queried once; immutable afterwards): grid-quantisation rules count rounds of this many workgroups
static inline int num_cus() {
    static const int n = [] {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }();
    return n;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
