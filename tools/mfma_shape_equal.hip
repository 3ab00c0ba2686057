// Do v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16 accumulate a K = 1024 dot product to the SAME fp32 bits when both walk k in
// ascending order?  (Round 4: a persistent GEMM on the 32x32x16 shape hides 4 fillers per MFMA instead of 1 -- profiles/
// r02_mfma_filler_probe.txt --, but it may only replace the 16x16x32 kernels under the tower's bit-wise batch invariance if the two
// shapes agree.)  One wave computes the same 32 x 32 block D = W X^T both ways from global memory; the host compares bit patterns.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_shape_equal tools/mfma_shape_equal.hip && tools/mfma_shape_equal
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// W [32][K], X [32][K] 16-bit, K-contiguous.  out32 / out16: [32 (w row)][32 (x row)] fp32.
template <bool F16>
__global__ void k(const unsigned short* W, const unsigned short* X, int K, float* out32, float* out16) {
    const int l = threadIdx.x;
    {   // 32x32x16: A = W fragment (row l & 31, k group l >> 5), B = X fragment likewise; D[i][j]: j = l & 31, i = 8 (v >> 2) + 4 (l >> 5) + (v & 3)
        f32x16 acc;
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        for (int s = 0; s < K / 16; ++s) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(W + (size_t)(l & 31) * K + 16 * s + 8 * (l >> 5));
            const u32x4 b = *reinterpret_cast<const u32x4*>(X + (size_t)(l & 31) * K + 16 * s + 8 * (l >> 5));
            if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
        }
        for (int v = 0; v < 16; ++v) out32[(8 * (v >> 2) + 4 * (l >> 5) + (v & 3)) * 32 + (l & 31)] = acc[v];
    }
    for (int bi = 0; bi < 2; ++bi)
        for (int bj = 0; bj < 2; ++bj) {   // 16x16x32: row l & 15, k group l >> 4; D[i][j]: j = l & 15, i = 4 (l >> 4) + v
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < K / 32; ++s) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(W + (size_t)(16 * bi + (l & 15)) * K + 32 * s + 8 * (l >> 4));
                const u32x4 b = *reinterpret_cast<const u32x4*>(X + (size_t)(16 * bj + (l & 15)) * K + 32 * s + 8 * (l >> 4));
                if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
            }
            for (int v = 0; v < 4; ++v) out16[(16 * bi + 4 * (l >> 4) + v) * 32 + 16 * bj + (l & 15)] = acc[v];
        }
}

static unsigned short to_bf16(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static unsigned short to_f16(float f) { _Float16 h = (_Float16)f; unsigned short s; memcpy(&s, &h, 2); return s; }

int main() {
    int all_equal = 1;
    for (int f16 = 0; f16 < 2; ++f16)
        for (int trial = 0; trial < 6; ++trial) {
            const int K = trial < 3 ? 1024 : 4096;
            unsigned short *hW = (unsigned short*)malloc(32 * K * 2), *hX = (unsigned short*)malloc(32 * K * 2);
            srand(17 + trial + 100 * f16);
            for (int i = 0; i < 32 * K; ++i) {
                // wide dynamic range and both signs: cancellation makes the result sensitive to the order of the fp32 additions
                const float m = expf(((rand() % 2000) / 1000.f - 1.f) * (trial % 3 == 2 ? 4.f : 1.5f));
                const float a = ((rand() % 2001) / 1000.f - 1.f) * m, b = ((rand() % 2001) / 1000.f - 1.f) * (trial % 3 == 1 ? m : 1.f);
                hW[i] = f16 ? to_f16(a * 0.1f) : to_bf16(a); hX[i] = f16 ? to_f16(b * 0.1f) : to_bf16(b);
            }
            unsigned short *dW, *dX; float *d32, *d16;
            hipMalloc(&dW, 32 * K * 2); hipMalloc(&dX, 32 * K * 2); hipMalloc(&d32, 4096); hipMalloc(&d16, 4096);
            hipMemcpy(dW, hW, 32 * K * 2, hipMemcpyHostToDevice); hipMemcpy(dX, hX, 32 * K * 2, hipMemcpyHostToDevice);
            if (f16) hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, dW, dX, K, d32, d16);
            else hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, dW, dX, K, d32, d16);
            float h32[1024], h16[1024];
            hipMemcpy(h32, d32, 4096, hipMemcpyDeviceToHost); hipMemcpy(h16, d16, 4096, hipMemcpyDeviceToHost);
            int diff = 0; double maxrel = 0, ref_err = 0;
            for (int i = 0; i < 1024; ++i) {
                if (memcmp(&h32[i], &h16[i], 4)) { ++diff; const double r = fabs((double)h32[i] - h16[i]) / (fabs((double)h16[i]) + 1e-30); if (r > maxrel) maxrel = r; }
            }
            printf("%s K %4d trial %d: %4d of 1024 outputs differ between 32x32x16 and 16x16x32 (max rel %.3g)  sample %.9g vs %.9g\n",
                   f16 ? "f16 " : "bf16", K, trial, diff, maxrel, h32[5], h16[5]);
            if (diff) all_equal = 0;
            hipFree(dW); hipFree(dX); hipFree(d32); hipFree(d16); free(hW); free(hX);
        }
    printf("MFMA SHAPES %s\n", all_equal ? "AGREE BITWISE" : "DIFFER");
    return 0;
}
