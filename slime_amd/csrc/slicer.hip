// slicer.hip -- the resampling half of the SliME image slicer on the GPU (gfx950).
//
// Pillow's ImagingResample (8 bpc) restated for the device: the weights are produced on the host in double
// precision exactly as Pillow does (precompute_coeffs + normalize_coeffs_8bpc; contraction off so no fma
// sneaks in), the two passes run here in 32-bit integer arithmetic.  Both kernels are HBM/L2 streaming
// kernels over a few MB: the vertical pass is perfectly coalesced (a thread owns one byte column), the
// horizontal pass stages the source span of a 256-pixel output segment in LDS with 16-byte loads.
#include "common.h"
#include <math.h>

namespace {
constexpr int PRECISION_BITS = 32 - 8 - 2;

#pragma clang fp contract(off)
inline double bicubic_weight(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

__device__ __forceinline__ uint8_t clip8(int acc) {
    const int v = acc >> PRECISION_BITS;
    return (uint8_t)min(max(v, 0), 255);
}

// Horizontal pass.  One workgroup = one source row x SEG output pixels.  LDS holds the source bytes the
// segment needs ([xmin(first), xmin(last)+count(last)) x 3), fetched 16 B per lane; spans that do not fit
// (extreme down-scales) are read from global memory directly.
constexpr int SEG = 256, SPAN_BYTES = 48 * 1024;
__global__ void __launch_bounds__(256) resample_h_kernel(const uint8_t* __restrict__ src, long src_image_stride, long src_stride,
                                                         int src_w, const uint8_t* src_begin, const uint8_t* src_end,
                                                         uint8_t* __restrict__ dst, long dst_image_stride, long dst_stride, int out_w,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
    __shared__ __attribute__((aligned(16))) uint8_t span[SPAN_BYTES];
    const int y = blockIdx.y, x0 = blockIdx.x * SEG, x1 = min(x0 + SEG, out_w) - 1;
    const uint8_t* row = src + (size_t)blockIdx.z * src_image_stride + (size_t)y * src_stride;
    dst += (size_t)blockIdx.z * dst_image_stride;
    const int first = bounds[2 * x0], last = bounds[2 * x1] + bounds[2 * x1 + 1];     // source pixels [first, last)
    // stage [g0, g1): g0 = the 16-B aligned address at or below the first needed byte (absolute alignment: rows of
    // a packed RGB image are not 16-B aligned themselves)
    const uint8_t* need0 = row + (size_t)first * 3;
    const uint8_t* g1 = row + (size_t)last * 3;
    const uint8_t* g0 = reinterpret_cast<const uint8_t*>(reinterpret_cast<size_t>(need0) & ~(size_t)15);
    const bool staged = (g1 - g0) <= SPAN_BYTES;
    if (staged) {
        for (const uint8_t* g = g0 + threadIdx.x * 16; g < g1; g += 256 * 16) {
            if (g >= src_begin && g + 16 <= src_end) {
                *reinterpret_cast<u32x4*>(span + (g - g0)) = *reinterpret_cast<const u32x4*>(g);
            } else {
                for (int i = 0; i < 16; ++i)
                    if (g + i >= src_begin && g + i < src_end) span[g - g0 + i] = g[i];
            }
        }
        __syncthreads();
    }
    const int x = x0 + threadIdx.x;
    if (x >= out_w) return;
    const int xmin = bounds[2 * x], cnt = bounds[2 * x + 1];
    const int* k = kk + (size_t)x * ksize;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    if (staged) {
        const uint8_t* p = span + (row + (size_t)xmin * 3 - g0);
        for (int t = 0; t < cnt; ++t) {
            const int w = k[t];
            s0 += p[3 * t] * w; s1 += p[3 * t + 1] * w; s2 += p[3 * t + 2] * w;
        }
    } else {
        const uint8_t* p = row + (size_t)xmin * 3;
        for (int t = 0; t < cnt; ++t) {
            const int w = k[t];
            s0 += p[3 * t] * w; s1 += p[3 * t + 1] * w; s2 += p[3 * t + 2] * w;
        }
    }
    uint8_t* d = dst + (size_t)y * dst_stride + (size_t)x * 3;
    d[0] = clip8(s0); d[1] = clip8(s1); d[2] = clip8(s2);
}

// Vertical pass: a thread owns 4 consecutive bytes of an output row (x*3+c is just a byte column for this
// pass), taps walk down the source rows -> every load/store of a wave is one contiguous 256-B segment.
__global__ void __launch_bounds__(256) resample_v_kernel(const uint8_t* __restrict__ src, long src_image_stride, long src_stride,
                                                         uint8_t* __restrict__ dst, long dst_image_stride, long dst_stride, int row_bytes,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
    const int y = blockIdx.y;
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= row_bytes) return;
    src += (size_t)blockIdx.z * src_image_stride;
    dst += (size_t)blockIdx.z * dst_image_stride;
    const int ymin = bounds[2 * y], cnt = bounds[2 * y + 1];
    const int* k = kk + (size_t)y * ksize;
    int s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = 1 << (PRECISION_BITS - 1);
    const bool full = c + 4 <= row_bytes && ((src_stride | (size_t)src) & 3) == 0;
    for (int t = 0; t < cnt; ++t) {
        const uint8_t* p = src + (size_t)(ymin + t) * src_stride + c;
        const int w = k[t];
        if (full) {
            const unsigned v = *reinterpret_cast<const unsigned*>(p);
            s[0] += (int)(v & 255u) * w; s[1] += (int)((v >> 8) & 255u) * w;
            s[2] += (int)((v >> 16) & 255u) * w; s[3] += (int)(v >> 24) * w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (c + i < row_bytes) s[i] += p[i] * w;
        }
    }
    uint8_t* d = dst + (size_t)y * dst_stride + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (c + i < row_bytes) d[i] = clip8(s[i]);
}
}  // namespace

extern "C" int slime_resample_ksize(int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0) return 0;
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(2.0 * filterscale) * 2 + 1;
}

#pragma clang fp contract(off)
extern "C" int slime_resample_coeffs(int in_size, int out_size, int* bounds, int* kk) {
    SLIME_REQUIRE(in_size > 0 && out_size > 0 && bounds && kk, "resample_coeffs: bad arguments (%d -> %d)", in_size, out_size);
    // Pillow: precompute_coeffs(inSize, 0, inSize, outSize, BICUBIC) followed by normalize_coeffs_8bpc
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    const double ss = 1.0 / filterscale;
    double wbuf[4096];
    SLIME_REQUIRE(ksize <= 4096, "resample_coeffs: scale %d -> %d too large", in_size, out_size);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_weight((x + xmin - center + 0.5) * ss);
            wbuf[x] = w;
            ww += w;
        }
        int* k = kk + (size_t)xx * ksize;
        for (int x = 0; x < ksize; ++x) {
            double w = 0.0;
            if (x < xmax) w = (ww != 0.0) ? wbuf[x] / ww : wbuf[x];
            k[x] = w < 0 ? (int)(-0.5 + w * (1 << PRECISION_BITS)) : (int)(0.5 + w * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return SLIME_OK;
}

extern "C" int slime_resize_bicubic_u8_batched(const uint8_t* src, int images, long src_image_stride, int src_h, int src_w,
                                               long src_stride, uint8_t* dst, long dst_image_stride, long dst_stride, int out_h,
                                               int out_w, const int* bounds_h, const int* kk_h, int ksize_h, const int* bounds_v,
                                               const int* kk_v, int ksize_v, uint8_t* tmp, size_t tmp_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SLIME_REQUIRE(src && dst && src_h > 0 && src_w > 0 && out_h > 0 && out_w > 0, "resize_bicubic_u8: bad arguments");
    SLIME_REQUIRE(images > 0 && images <= 65535, "resize_bicubic_u8: %d images", images);
    SLIME_REQUIRE(src_stride >= (long)src_w * 3 && dst_stride >= (long)out_w * 3, "resize_bicubic_u8: row stride smaller than a row");
    const bool need_h = out_w != src_w, need_v = out_h != src_h;
    SLIME_REQUIRE(!need_h || (bounds_h && kk_h && ksize_h == slime_resample_ksize(src_w, out_w)),
                  "resize_bicubic_u8: horizontal tables missing or ksize %d != %d", ksize_h, slime_resample_ksize(src_w, out_w));
    SLIME_REQUIRE(!need_v || (bounds_v && kk_v && ksize_v == slime_resample_ksize(src_h, out_h)),
                  "resize_bicubic_u8: vertical tables missing or ksize %d != %d", ksize_v, slime_resample_ksize(src_h, out_h));
    if (!need_h && !need_v) {
        for (int b = 0; b < images; ++b) {
            hipError_t e = hipMemcpy2DAsync(dst + (size_t)b * dst_image_stride, dst_stride, src + (size_t)b * src_image_stride, src_stride,
                                            (size_t)src_w * 3, src_h, hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) { slime_set_error("resize_bicubic_u8: copy: %s", hipGetErrorString(e)); return SLIME_ELAUNCH; }
        }
        return SLIME_OK;
    }
    const uint8_t* vsrc = src;
    long vstride = src_stride, vimg = src_image_stride;
    if (need_h) {
        uint8_t* hdst = need_v ? tmp : dst;
        const long hstride = need_v ? (long)out_w * 3 : dst_stride;
        const long himg = need_v ? (long)src_h * out_w * 3 : dst_image_stride;
        if (need_v) SLIME_REQUIRE(tmp && tmp_bytes >= (size_t)images * src_h * out_w * 3, "resize_bicubic_u8: tmp needs %zu bytes", (size_t)images * src_h * out_w * 3);
        dim3 grid((out_w + SEG - 1) / SEG, src_h, images);
        const uint8_t* src_end = src + (size_t)(images - 1) * src_image_stride + (size_t)(src_h - 1) * src_stride + (size_t)src_w * 3;
        hipLaunchKernelGGL(resample_h_kernel, grid, dim3(256), 0, stream, src, src_image_stride, src_stride, src_w, src, src_end, hdst, himg,
                           hstride, out_w, bounds_h, kk_h, ksize_h);
        SLIME_CHECK_LAUNCH("resample_h");
        vsrc = hdst;
        vstride = hstride;
        vimg = himg;
    }
    if (need_v) {
        const int row_bytes = out_w * 3;
        dim3 grid((row_bytes + 1023) / 1024, out_h, images);
        hipLaunchKernelGGL(resample_v_kernel, grid, dim3(256), 0, stream, vsrc, vimg, vstride, dst, dst_image_stride, dst_stride, row_bytes,
                           bounds_v, kk_v, ksize_v);
        SLIME_CHECK_LAUNCH("resample_v");
    }
    return SLIME_OK;
}

extern "C" int slime_resize_bicubic_u8(const uint8_t* src, int src_h, int src_w, long src_stride, uint8_t* dst,
                                       long dst_stride, int out_h, int out_w, const int* bounds_h, const int* kk_h,
                                       int ksize_h, const int* bounds_v, const int* kk_v, int ksize_v, uint8_t* tmp,
                                       size_t tmp_bytes, void* stream) {
    return slime_resize_bicubic_u8_batched(src, 1, 0, src_h, src_w, src_stride, dst, 0, dst_stride, out_h, out_w, bounds_h, kk_h, ksize_h,
                                           bounds_v, kk_v, ksize_v, tmp, tmp_bytes, stream);
}
