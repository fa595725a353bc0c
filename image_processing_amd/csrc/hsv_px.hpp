// One pixel of cvtColor(BGR2HSV) in the form the chain kernels use (fused_kernel.hpp, gauss_fused.hip): channel bytes
// times 4 in, H | S << 16 and 4 V out, reciprocal tables (times 4) in LDS.  Arithmetic: SURVEY.md Appendix A.1
// (OpenCV's RGB2HSV_b, hrange 180), bit-exact with k_bgr2hsv and the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ipf {

// hipcc does not reliably select the 24-bit multiply-adds (it falls back to quarter-rate v_mul_lo_u32 /
// v_mad_u64_u32), so they are spelled out.
__device__ __forceinline__ int mad_i24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int max3u(int a, int b, int c)
{
    int r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int min3u(int a, int b, int c)
{
    int r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// byte B of w, times 4, in one SDWA shift (the compiler spends two ops on byte 0)
template <int B>
__device__ __forceinline__ uint32_t byte_x4(uint32_t w)
{
    uint32_t r;
    const uint32_t two = 2u;
    if (B == 0)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(two), "v"(w));
    else if (B == 1)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(two), "v"(w));
    else if (B == 2)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(two), "v"(w));
    else
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(two), "v"(w));
    return r;
}

// One pixel of cvtColor(BGR2HSV).  Inputs are the channel bytes times 4 (they double as
// byte offsets into the LDS reciprocal tables, which hold 4*sdiv and 4*hdiv180);
// (4hh * 4hdiv + 2^15) >> 16 == (hh*hdiv + 2^11) >> 12, so H lands in byte 2 of the product like S does.
template <bool RGB>
__device__ __forceinline__ void hsv_px(uint32_t c0, uint32_t c1, uint32_t c2, const char *tab, uint32_t &hs,
                                       uint32_t &v)
{
    const int b4 = RGB ? c2 : c0, g4 = c1, r4 = RGB ? c0 : c2;
    const int v4 = max3u(b4, g4, r4), m4 = min3u(b4, g4, r4), d4 = v4 - m4;
    const int sdiv4 = *reinterpret_cast<const int *>(tab + v4);        // 4 * sdiv[v]
    const int hdiv4 = *reinterpret_cast<const int *>(tab + 1024 + d4);  // 4 * hdiv180[diff]
    // S << 16 == ((d*sdiv + 2^11) >> 12) << 16 == (4d * 4sdiv + 2^15) & 0xFFFF0000   (4d*4sdiv <= 2^24)
    const uint32_t s16 = mad_u24((uint32_t)d4, (uint32_t)sdiv4, 32768u);  // S in byte 2, byte 3 = 0
    int hr = g4 - b4, hg = (b4 - r4) + 2 * d4, hb = (r4 - g4) + 4 * d4;
    asm("" : "+v"(hr), "+v"(hg), "+v"(hb));  // keep the three candidates live: selects, not branches
    const int hh = (v4 == r4) ? hr : (v4 == g4) ? hg : hb;
    const uint32_t x = (uint32_t)mad_i24(hh, hdiv4, 32768);  // |4hh * 4hdiv| <= 180 << 16
    const uint32_t xw = min(x, x + (180u << 16));              // negative hue: + 180 (as unsigned, the wrapped sum is the smaller)
    hs = __builtin_amdgcn_perm(s16, xw, 0x07060C02u);          // H = byte 2 of xw -> byte 0; S = byte 2 of s16 stays
    v = (uint32_t)v4;  // still times 4; hsv_row unscales a packed pair with one shift
}

}  // namespace ipf
