// Dumps the lane/element mapping of ds_read_b64_tr_b16 on gfx950 (diagnostic, not product code).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(unsigned short* out, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short* l = (unsigned short*)smem;
    for (int i = threadIdx.x; i < 4096; i += 64) l[i] = (unsigned short)i;
    __syncthreads();
    int lane = threadIdx.x;
    // mode 0: contiguous (lane*8 bytes); mode 1: lane (4j+c) in quad -> row j (stride 128 B), chunk c (8 B each at 16 B stride)
    unsigned addr;
    if (mode == 0) addr = lane * 8;
    else { int g = lane >> 4, i = lane & 15; addr = (4 * g + (i >> 2)) * 128 + (i & 3) * 16; }
    typedef __attribute__((address_space(3))) s16x4* p3;
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((p3)((__attribute__((address_space(3))) char*)smem + addr));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)r[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 8192, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (element index = byte_addr/2)\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
