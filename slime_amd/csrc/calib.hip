// calib.hip -- measurement aid of the C ABI (slime_mfma_stream_probe): what do this chip's matrix pipes deliver, right now, when
// nothing but v_mfma_f32_16x16x32 instructions on register-resident random operands is in flight?  bench.py runs it for ~0.3 s beside
// the timed step and reports the step's MFMA rate against THIS figure as well as against the 2.5 PFLOP/s dense peak: the MI355X
// holds its 1400 W cap by lowering sclk, so the bare stream itself reaches ~0.65 of peak on fresh operands, and the ceiling moves
// from box to box (a constant measured on another box in another round divides nothing meaningfully).
//
// One workgroup of eight waves per CU (two per SIMD, the occupancy of the product GEMMs).  A wave loads the fragments of a 64 x 64
// output tile for two k-steps -- 2 x (4 A + 4 B) fragments = 64 VGPRs of the caller's random 16-bit values, a different 8 KiB per
// wave -- and then issues, per iteration, the 32 MFMAs of such a k-tile: acc[i][j] += A[ks][i] . B[ks][j].  Consecutive MFMAs switch
// both operands (as in a GEMM main loop), every accumulator is reused 16 MFMAs (>= 256 cycles) later, no memory instruction and
// no barrier sits in the loop.  The sums leave through one store per lane so that the loop is live.
#include "common.h"

namespace {

template <typename T>
__global__ void __launch_bounds__(512) mfma_stream_kernel(const u32x4* __restrict__ operands, unsigned slots, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 8u + (threadIdx.x >> 6);
    const u32x4* src = operands + (size_t)(wave % slots) * (16 * 64) + lane;      // 16 fragments x 64 lanes x 16 B = 16 KiB per wave slot
    u32x4 A[2][4], B[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            A[ks][i] = src[(ks * 8 + i) * 64];
            B[ks][i] = src[(ks * 8 + 4 + i) * 64];
        }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = T::mfma16(A[ks][i], B[ks][j], acc[i][j]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

}  // namespace

extern "C" int slime_mfma_stream_probe(int dtype, int iters, const void* operands, size_t operand_bytes, float* out, size_t out_bytes,
                                       double* flops_host, void* stream) {
    SLIME_REQUIRE(dtype == SLIME_BF16 || dtype == SLIME_F16, "mfma_stream_probe: dtype must be BF16 or F16");
    SLIME_REQUIRE(iters > 0 && operands && out && ((uintptr_t)operands % 16) == 0 && ((uintptr_t)out % 4) == 0, "mfma_stream_probe: bad input");
    const unsigned slots = (unsigned)(operand_bytes / (16 * 1024));
    SLIME_REQUIRE(slots >= 1, "mfma_stream_probe: operands must hold at least 16 KiB (one wave's fragments)");
    const int cus = num_cus();
    SLIME_REQUIRE(out_bytes >= (size_t)cus * 512 * sizeof(float), "mfma_stream_probe: out must hold %d x 512 floats", cus);
    if (dtype == SLIME_F16)
        hipLaunchKernelGGL(mfma_stream_kernel<F16>, dim3(cus), dim3(512), 0, (hipStream_t)stream, (const u32x4*)operands, slots, out, iters);
    else
        hipLaunchKernelGGL(mfma_stream_kernel<BF16>, dim3(cus), dim3(512), 0, (hipStream_t)stream, (const u32x4*)operands, slots, out, iters);
    SLIME_CHECK_LAUNCH("mfma_stream_probe");
    // 32 MFMAs of 16 x 16 x 32 (2 x 8192 flop) per wave and iteration
    if (flops_host) *flops_host = (double)cus * 8.0 * (double)iters * 32.0 * 16384.0;
    return SLIME_OK;
}
