// Internal launch interface between the C ABI (api.hip) and the gfx950 kernels (kernels.hip,
// fused.hip).  Everything here takes DEVICE pointers and enqueues on `stream`; nothing allocates.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ipk {

typedef uint8_t u8;

struct Taps {  // 8.8 fixed-point Gaussian taps (sum == 256)
    int k;
    int t[31];
};

struct Bounds {  // cv::inRange bounds after the never-match / clamp rules (lo=1,hi=0 == never)
    int n;
    u8 lo[8][3];
    u8 hi[8][3];
};

struct Element {  // structuring element as per-row bit masks (bit i of row j == elem[j][i])
    int kw, kh, ax, ay;
    int is_rect;
    uint32_t rows[31];
};

// HSV reciprocal tables (sdiv[256] then hdiv180[256]) live in device memory owned by the context
hipError_t launch_bgr2hsv(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, int bidx,
                          const int *d_tables, hipStream_t stream);
hipError_t launch_gaussian(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, int cn,
                           const Taps &taps, hipStream_t stream);
// strip kernel (gauss.hip): odd k <= 15, symmetric taps summing to 256, 4-byte aligned source rows
// long kernels (k = 7..15) as banded contractions on the matrix cores (gauss_mfma.hip)
bool gaussian_mfma_ok(const u8 *src, size_t sstride, int w, int h, int cn, const Taps &taps);
hipError_t launch_gaussian_mfma(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, int cn,
                                const Taps &taps, hipStream_t stream);
bool gaussian_strip_ok(const u8 *src, size_t sstride, int w, int h, int cn, const Taps &taps);
hipError_t launch_gaussian_strip(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, int cn,
                                 const Taps &taps, hipStream_t stream);
// dst plane r at dst + r*plane_stride
hipError_t launch_in_range(const u8 *src, size_t sstride, u8 *dst, size_t dstride, size_t plane_stride, int w,
                           int h, const Bounds &b, hipStream_t stream);
// inRange for 1..8 ranges with bit planes out: dst[range][h][(w+31)/32] words, bit i of word j = pixel 32 j + i
hipError_t launch_in_range_bits(const u8 *src, size_t sstride, uint32_t *dst, int w, int h, const Bounds &b,
                                hipStream_t stream);
// BGR2HSV + GaussianBlur(k = 7..15) + inRange in one kernel, bit planes out (gauss_fused.hip): n frames of BGR (bidx 0) or
// RGB (bidx 2) at src + i * frame_stride -> bits[n][b.n][h][(w+31)/32]
bool gauss_fused_ok(const u8 *src, size_t frame_stride, size_t sstride, int w, int h, const Taps &taps, const Bounds &b);
hipError_t launch_gauss_fused(const u8 *src, size_t frame_stride, size_t sstride, uint32_t *bits, int w, int h, int n,
                              const Taps &taps, const Bounds &b, int bidx, const int *d_hsv_tables, hipStream_t stream,
                              bool overlapped = false);  // overlapped: another batch's launches run beside this one (fewer, taller segments)
hipError_t launch_morph(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, int kind,
                        const Element &e, hipStream_t stream);
// adds into d_m[6] (u64: m00 m10 m01 m20 m11 m02); caller zeroes
hipError_t launch_moments(const u8 *mask, size_t stride, int w, int h, unsigned long long *d_m,
                          hipStream_t stream);
hipError_t launch_demosaic_rggb(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h,
                                hipStream_t stream);
hipError_t launch_mask_and(const u8 *bgr, size_t sstride, const u8 *mask, size_t mstride, u8 *dst,
                           size_t dstride, int w, int h, hipStream_t stream);

}  // namespace ipk
