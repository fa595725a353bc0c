// Fused chain kernel for gfx950 (template + device helpers; instantiated in fused_inst_*.hip, planned and
// launched from fused.hip)
// Fused chain kernel for gfx950: BGR/RGB -> HSV -> [5x5 Gaussian] -> inRange -> up to two 3x3 rect
// erode/dilate -> mask bytes + raster moments, one launch for a whole batch of frames.
//
// Replaces, for the pipelines plan_fused() accepts, the OpenCV call sequence of
// objectPolygonPath.cpp:183-188 / pose_estimate.cpp:203-206 (cvtColor, GaussianBlur 5x5, inRange,
// erode, dilate) and the m00/m10/m01 sums of moments() (objectPolygonPath.cpp:449-450).
// Arithmetic: SURVEY.md Appendix A, bit-exact with the stage kernels and the oracle.
//
// Design (MI355X: HBM-bound u8 work, no MFMA):
//  * One WAVE owns a strip of 256 columns (4 px per lane = one 12-byte load per lane and row; lanes 0
//    and 63 are halo) and marches down a segment of rows.  Waves never talk to each other: no
//    barriers after the table load, no LDS tiles, no vertical halo re-computation inside a segment.
//  * HSV rows live in a register ring; the vertical 1-4-6-4-1 pass runs on it, the horizontal pass
//    gets its +-2 px neighbours from the adjacent lanes with DPP wave shifts.
//  * Pixels are kept as packed 16-bit fields (H|S<<16 per pixel, V pairs), so every Gaussian add /
//    shift-add processes two values per VALU op; sums stay < 2^16 per field (max 255*256+128).
//    The single rounding (x+128)>>8 is never materialised: inRange compares the raw 8.8 sum against
//    (lo<<8 - 128, (hi<<8|255) - 128), the rounding constant having moved into the bounds on the host.
//  * inRange results are shifted into a per-lane 4-col x 8-row bit tile; 3x3 erode/dilate are a dozen
//    bit operations on that tile (horizontal neighbours again by DPP), 32 pixels per op.
//  * Moments are popcounts of the final tile; mask bytes are expanded from the tile rows and stored
//    as one dword per lane and row (256 B per wave row, coalesced).
#pragma once
#include "fused.h"
#include "hsv_px.hpp"

#include <algorithm>

#ifndef IPF_CASCADE
#define IPF_CASCADE 1  // 5x5 Gaussian as cascaded pair sums (plain adds); 0: the ring / shift-add form (kept for A/B)
#endif

namespace ipf {

using ipk::u8;

constexpr int strip_valid(bool) { return 248; }  // valid output columns per wave: lanes 1..62 (Bayer too: its lanes load the
                                                // mosaic columns next to their own four themselves, see demosaic_hsv_row)
constexpr int MAX_NR = 8;  // kernel variants hold 1, 2, 4 or 8 ranges; a.nr of them are live

struct Args {
    const u8 *frames;
    size_t frame_stride, row_stride;
    u8 *masks;
    uint32_t *bits;  // MODE 2: inRange result as bit planes [n][nr][h][bit_words], bit i of word j = pixel 32 j + i
    int bit_words;
    unsigned long long *mom;
    const int *tables;
    int w, h, n;
    int nr;  // live ranges (== NR for the 1- and 2-range variants)
    int seg_rows, n_segs, n_strips, total_items;
    int last_base;  // first valid column of the last strip: (n_strips-1)*SV, or w - SV (right-aligned) when w % 4 != 0
    int split_x;    // the last strip owns columns >= split_x, the others own [strip*SV, min((strip+1)*SV, split_x))
    int op0, op1;  // 0 erode3, 1 dilate3, 2 identity (one-row delay only)
    uint32_t hs_lo[MAX_NR], hs_hi[MAX_NR];  // H | S << 16 bounds (8.8 fields after the 5x5 blur)
    uint32_t v_lo[MAX_NR], v_hi[MAX_NR];    // V: (lo | lo << 16) and hi - lo; lo = 0xFFFF with hi - lo = 0 never matches
};

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t from_left(uint32_t v)  // lane l <- lane l-1
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xF, 0xF, true);  // bound_ctrl: lane 0 reads 0
}
__device__ __forceinline__ uint32_t from_right(uint32_t v)  // lane l <- lane l+1
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xF, 0xF, true);
}
// x[lane-1] + y and x[lane+1] + y in one instruction (lane 0 / lane 63 read 0)
__device__ __forceinline__ uint32_t add_from_left(uint32_t x, uint32_t y)
{
    uint32_t r;
    asm("v_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ uint32_t add_from_right(uint32_t x, uint32_t y)
{
    uint32_t r;
    asm("v_add_u32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, int sh)  // ({hi:lo} >> sh) low 32
{
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}
__device__ __forceinline__ uint32_t pk_clamp(uint32_t x, uint32_t lo, uint32_t hi)
{
    us2 a = __builtin_bit_cast(us2, x), l = __builtin_bit_cast(us2, lo), h = __builtin_bit_cast(us2, hi);
    us2 c = __builtin_elementwise_min(__builtin_elementwise_max(a, l), h);
    return __builtin_bit_cast(uint32_t, c);
}
// t = 2*t + bit, the bit coming straight from the compare's lane mask (carry-in)
__device__ __forceinline__ uint32_t shift_in(uint32_t t, unsigned long long m)
{
    uint32_t r;
    asm("v_addc_co_u32 %0, vcc, %1, %1, %2" : "=v"(r) : "v"(t), "s"(m) : "vcc");
    return r;
}
// BORDER_REFLECT_101 for the rows this kernel can ask for (|overshoot| < h for every row a valid output
// depends on; rows beyond that are prefetch overshoot and only need to stay inside the frame)
__device__ __forceinline__ int reflect_row(int r, int h)
{
    r = abs(r);
    r = r >= h ? 2 * (h - 1) - r : r;
    return min(abs(r), h - 1);
}

struct Row {  // one row of this lane's 4 pixels as packed 16-bit fields
    uint32_t hs[4];  // px k: H | S<<16
    uint32_t v[2];   // (V0,V1), (V2,V3)
};

template <bool RGB>
__device__ __forceinline__ void hsv_row(const uint32_t (&w)[3], const char *tab, Row &o)
{
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
    uint32_t vv[4];
    hsv_px<RGB>(byte_x4<0>(w0), byte_x4<1>(w0), byte_x4<2>(w0), tab, o.hs[0], vv[0]);
    hsv_px<RGB>(byte_x4<3>(w0), byte_x4<0>(w1), byte_x4<1>(w1), tab, o.hs[1], vv[1]);
    hsv_px<RGB>(byte_x4<2>(w1), byte_x4<3>(w1), byte_x4<0>(w2), tab, o.hs[2], vv[2]);
    hsv_px<RGB>(byte_x4<1>(w2), byte_x4<2>(w2), byte_x4<3>(w2), tab, o.hs[3], vv[3]);
    o.v[0] = (vv[0] | (vv[1] << 16)) >> 2;  // 4*V0 is a multiple of 4: no bits cross the field boundary
    o.v[1] = (vv[2] | (vv[3] << 16)) >> 2;
}

struct RowBGR {  // one raw row of this lane's 4 pixels as packed 16-bit fields (channel order as in memory)
    uint32_t c01[4];  // px k: c0 | c1<<16
    uint32_t c2[2];   // (c2 of px0, px1), (px2, px3)
};

// The 3x3 pre-blur runs on 16 * value, so that after its two 1-2-1 passes (x16) the blurred value is byte 1 / byte 3
// of a field and comes out, times 4, with one SDWA shift (fields stay below 2^16: 16 * (255 * 16 + 8) = 65408).
__device__ __forceinline__ void scale16(RowBGR &o)
{
    // + 8 per field: the rounding of the 3x3 blur (8 * (1+2+1) * (1+2+1) = 128 = half an output unit), folded into the
    // same v_lshl_add_u32 that scales
#pragma unroll
    for (int k = 0; k < 4; k++) o.c01[k] = (o.c01[k] << 4) + 0x00080008u;
    o.c2[0] = (o.c2[0] << 4) + 0x00080008u;
    o.c2[1] = (o.c2[1] << 4) + 0x00080008u;
}

__device__ __forceinline__ void unpack_row(const uint32_t (&w)[3], RowBGR &o)
{
    // v_perm_b32 byte selectors: 0-3 = bytes of the second operand, 4-7 = bytes of the first, 0x0C = 0x00
    o.c01[0] = __builtin_amdgcn_perm(w[0], w[0], 0x0C010C00u);
    o.c01[1] = __builtin_amdgcn_perm(w[1], w[0], 0x0C040C03u);
    o.c01[2] = __builtin_amdgcn_perm(w[1], w[1], 0x0C030C02u);
    o.c01[3] = __builtin_amdgcn_perm(w[2], w[2], 0x0C020C01u);
    o.c2[0] = __builtin_amdgcn_perm(w[1], w[0], 0x0C050C02u);
    o.c2[1] = __builtin_amdgcn_perm(w[2], w[2], 0x0C030C00u);
    scale16(o);
}

// GaussianBlur 3x3 (objectPolygonPath.cpp:182) on the raw channels, then HSV.  Fields hold 16 * (16*value + 8) after the two
// 1-2-1 passes (inputs are 16 * byte + 8, see scale16: scale and folded rounding), so the
// channel value that hsv_px wants (times 4) is byte 1 of a field, shifted: byte_x4<1> / byte_x4<3>.
template <bool RGB>
__device__ __forceinline__ void preblur_hsv_row(const RowBGR &r0, const RowBGR &r1, const RowBGR &r2, int border_strip,
                                                bool lb_halo, bool rb_halo, const char *tab, Row &o)
{
    uint32_t v[4], u[2];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = r0.c01[k] + r2.c01[k] + (r1.c01[k] << 1);
#pragma unroll
    for (int k = 0; k < 2; k++) u[k] = r0.c2[k] + r2.c2[k] + (r1.c2[k] << 1);
    if (border_strip) {  // BORDER_REFLECT_101 left/right: col -1 = col 1, col w = col w-2 (see the post-blur patch)
        const uint32_t a1 = from_right(v[1]), au = from_right(u[0]);  // lane+1's px1 / (R0,R1)
        const uint32_t b2 = from_left(v[2]), bu = from_left(u[1]);    // lane-1's px2 / (R2,R3)
        if (lb_halo) { v[3] = a1; u[1] = au; }
        if (rb_halo) { v[0] = b2; u[0] = bu; }
    }
    uint32_t h01[4], h2[2];
    h01[0] = add_from_left(v[3], v[1]) + (v[0] << 1);
    h01[1] = v[0] + v[2] + (v[1] << 1);
    h01[2] = v[1] + v[3] + (v[2] << 1);
    h01[3] = add_from_right(v[0], v[2]) + (v[3] << 1);
    const uint32_t ul = from_left(u[1]), ur = from_right(u[0]);
    const uint32_t m1 = alignbit(u[0], ul, 16);    // (c2[-1], c2[0])
    const uint32_t m2 = alignbit(u[1], u[0], 16);  // (c2[1], c2[2])
    const uint32_t m3 = alignbit(ur, u[1], 16);    // (c2[3], c2[4])
    h2[0] = m1 + m2 + (u[0] << 1);
    h2[1] = m2 + m3 + (u[1] << 1);
    uint32_t vv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t f2 = h2[k >> 1];
        const uint32_t c2 = (k & 1) ? byte_x4<3>(f2) : byte_x4<1>(f2);
        hsv_px<RGB>(byte_x4<1>(h01[k]), byte_x4<3>(h01[k]), c2, tab, o.hs[k], vv[k]);
    }
    o.v[0] = (vv[0] | (vv[1] << 16)) >> 2;
    o.v[1] = (vv[2] | (vv[3] << 16)) >> 2;
}

// BayerRG8 front end (cv_bridge's bayer_rggb8 -> bgr8 == COLOR_BayerBG2BGR at pose_estimate.cpp:180):
// bilinear demosaic of this lane's 4 pixels from the three mosaic rows y-1, y, y+1 (one dword each), SWAR on byte
// pairs.  Index [0] holds pixels (0,2) = even columns, [1] holds pixels (1,3) = odd columns, as 16-bit fields.
struct Pairs {
    uint32_t b[2], g[2], r[2];
};

// The three mosaic rows arrive as 8-byte windows (w[2i], w[2i+1] = low / high dword of row y-1+i): bytes b0..b7 = columns
// x0-1 .. x0+6 of the lane's four pixels x0 .. x0+3, so every neighbour a pixel is interpolated from sits in the lane's own
// registers (v_perm_b32 picks the byte pairs) and no lane depends on its neighbour's load: one halo lane per side is
// enough, as for BGR input.  Field pair (bx, by) of a window:
__device__ __forceinline__ uint32_t win_pair(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
constexpr uint32_t WP_E = 0x0C030C01u;  // (b1, b3): the even columns x0, x0+2
constexpr uint32_t WP_O = 0x0C040C02u;  // (b2, b4): the odd columns x0+1, x0+3
constexpr uint32_t WP_L = 0x0C020C00u;  // (b0, b2): left neighbours of the even columns
constexpr uint32_t WP_R = 0x0C050C03u;  // (b3, b5): right neighbours of the odd columns

// Lanes whose window would start before the row (x0 = 0) or end beyond it load the row's first / last 8 bytes instead;
// this moves the bytes to where the window has them (border strips only; what moves in from outside is never used: the
// image's border columns copy their neighbours)
__device__ __forceinline__ void bayer_window_fix(uint32_t &lo_io, uint32_t &hi_io, bool win_lo, int hi_sh8)
{
    const uint32_t lo = lo_io, hi = hi_io;
    if (win_lo) {
        lo_io = lo << 8;
        hi_io = alignbit(hi, lo, 24);
    } else if (hi_sh8) {
        lo_io = alignbit(hi, lo, hi_sh8);
        hi_io = hi >> hi_sh8;
    }
}

__device__ __forceinline__ void demosaic_row(const uint32_t (&w)[6], bool odd_row, int border_strip, bool first_col,
                                             bool last_col, Pairs &o)
{
    const uint32_t M8 = 0x00FF00FFu;
    const uint32_t me = win_pair(w[3], w[2], WP_E), mo = win_pair(w[3], w[2], WP_O);  // centre row
    const uint32_t vse = win_pair(w[1], w[0], WP_E) + win_pair(w[5], w[4], WP_E);     // up + down
    const uint32_t vso = win_pair(w[1], w[0], WP_O) + win_pair(w[5], w[4], WP_O);
    const uint32_t hme = win_pair(w[3], w[2], WP_L) + mo;  // left + right of the even columns
    const uint32_t hmo = me + win_pair(w[3], w[2], WP_R);  // left + right of the odd columns
    if (!odd_row) {  // R G R G: even columns are R sites, odd columns G with R left/right and B above/below
        const uint32_t hve = win_pair(w[1], w[0], WP_L) + win_pair(w[5], w[4], WP_L) + vso;  // the four diagonals of the even columns
        o.r[0] = me;
        o.g[0] = ((hme + vse + 0x00020002u) >> 2) & M8;
        o.b[0] = ((hve + 0x00020002u) >> 2) & M8;
        o.g[1] = mo;
        o.r[1] = ((hmo + 0x00010001u) >> 1) & M8;
        o.b[1] = ((vso + 0x00010001u) >> 1) & M8;
    } else {  // G B G B: odd columns are B sites, even columns G with B left/right and R above/below
        const uint32_t hvo = vse + win_pair(w[1], w[0], WP_R) + win_pair(w[5], w[4], WP_R);
        o.g[0] = me;
        o.b[0] = ((hme + 0x00010001u) >> 1) & M8;
        o.r[0] = ((vse + 0x00010001u) >> 1) & M8;
        o.b[1] = mo;
        o.g[1] = ((hmo + vso + 0x00020002u) >> 2) & M8;
        o.r[1] = ((hvo + 0x00020002u) >> 2) & M8;
    }
    if (border_strip) {  // OpenCV leaves no interpolated border: column 0 copies column 1, column w-1 copies w-2
        if (first_col) {
            o.b[0] = (o.b[0] & 0xFFFF0000u) | (o.b[1] & 0xFFFFu);
            o.g[0] = (o.g[0] & 0xFFFF0000u) | (o.g[1] & 0xFFFFu);
            o.r[0] = (o.r[0] & 0xFFFF0000u) | (o.r[1] & 0xFFFFu);
        }
        if (last_col) {
            o.b[1] = (o.b[1] & 0xFFFFu) | (o.b[0] & 0xFFFF0000u);
            o.g[1] = (o.g[1] & 0xFFFFu) | (o.g[0] & 0xFFFF0000u);
            o.r[1] = (o.r[1] & 0xFFFFu) | (o.r[0] & 0xFFFF0000u);
        }
    }
}

// 16-bit field H of w, times 4, in one SDWA shift
template <int H>
__device__ __forceinline__ uint32_t word_x4(uint32_t w)
{
    uint32_t r;
    const uint32_t two = 2u;
    if (H == 0)
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(two), "v"(w));
    else
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(two), "v"(w));
    return r;
}

__device__ __forceinline__ void hsv_row_pairs(const Pairs &p, const char *tab, Row &o)
{
    uint32_t vv[4];  // pixel k sits in pair k & 1, field k >> 1
    hsv_px<false>(word_x4<0>(p.b[0]), word_x4<0>(p.g[0]), word_x4<0>(p.r[0]), tab, o.hs[0], vv[0]);
    hsv_px<false>(word_x4<0>(p.b[1]), word_x4<0>(p.g[1]), word_x4<0>(p.r[1]), tab, o.hs[1], vv[1]);
    hsv_px<false>(word_x4<1>(p.b[0]), word_x4<1>(p.g[0]), word_x4<1>(p.r[0]), tab, o.hs[2], vv[2]);
    hsv_px<false>(word_x4<1>(p.b[1]), word_x4<1>(p.g[1]), word_x4<1>(p.r[1]), tab, o.hs[3], vv[3]);
    o.v[0] = (vv[0] | (vv[1] << 16)) >> 2;
    o.v[1] = (vv[2] | (vv[3] << 16)) >> 2;
}

// 16-bit field H of w, masked to a multiple of 4 (bits 2..9), in one SDWA and: the interpolated Bayer channels are kept
// as "4 x value plus junk in bits 0..1" so that this is their whole extraction
template <int H>
__device__ __forceinline__ uint32_t word_and_x4(uint32_t w)
{
    uint32_t r;
    const uint32_t m = 0x3FCu;
    if (H == 0)
        asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(m), "v"(w));
    else
        asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(m), "v"(w));
    return r;
}

// Demosaic + HSV of one fed row without the 8-bit intermediate (the no-pre-blur Bayer chain, BASELINE config 3).
// The four-neighbour averages leave (sum + 2) and the two-neighbour averages (2 sum + 2) in their fields: both are
// 4 x the rounded average plus junk in bits 0..1, which word_and_x4 drops while it extracts -- one op per value
// instead of shift + mask per pair and a shift per value.  Arithmetic identical to demosaic_row + hsv_row_pairs:
// 4 ((a+b+c+d+2) >> 2) = (a+b+c+d+2) & ~3 and 4 ((a+b+1) >> 1) = (2a+2b+2) & ~3.
__device__ __forceinline__ void demosaic_hsv_row(const uint32_t (&w)[6], bool odd_row, int border_strip, bool first_col,
                                                 bool last_col, const char *tab, Row &o)
{
    const uint32_t C2 = 0x00020002u;
    const uint32_t me = win_pair(w[3], w[2], WP_E), mo = win_pair(w[3], w[2], WP_O);  // even / odd columns of the centre row
    const uint32_t vse = win_pair(w[1], w[0], WP_E) + win_pair(w[5], w[4], WP_E);     // up + down, even columns
    const uint32_t vso = win_pair(w[1], w[0], WP_O) + win_pair(w[5], w[4], WP_O);     // up + down, odd columns
    const uint32_t hme = win_pair(w[3], w[2], WP_L) + mo;  // left + right of the even columns
    const uint32_t hmo = me + win_pair(w[3], w[2], WP_R);  // left + right of the odd columns
    uint32_t c4[4][3];  // pixel k: 4b, 4g, 4r (pixel k sits in pair k & 1, field k >> 1)
    if (!odd_row) {  // R G R G: even columns are R sites, odd columns G with R left/right and B above/below
        const uint32_t hve = win_pair(w[1], w[0], WP_L) + win_pair(w[5], w[4], WP_L) + vso;  // the four diagonals of the even columns
        const uint32_t g0 = hme + vse + C2, b0 = hve + C2;             // even columns: G, B
        const uint32_t r1 = (hmo << 1) + C2, b1 = (vso << 1) + C2;     // odd columns: R, B
        c4[0][0] = word_and_x4<0>(b0); c4[0][1] = word_and_x4<0>(g0); c4[0][2] = word_x4<0>(me);
        c4[2][0] = word_and_x4<1>(b0); c4[2][1] = word_and_x4<1>(g0); c4[2][2] = word_x4<1>(me);
        c4[1][0] = word_and_x4<0>(b1); c4[1][1] = word_x4<0>(mo); c4[1][2] = word_and_x4<0>(r1);
        c4[3][0] = word_and_x4<1>(b1); c4[3][1] = word_x4<1>(mo); c4[3][2] = word_and_x4<1>(r1);
    } else {  // G B G B: odd columns are B sites, even columns G with B left/right and R above/below
        const uint32_t hvo = vse + win_pair(w[1], w[0], WP_R) + win_pair(w[5], w[4], WP_R);
        const uint32_t b0 = (hme << 1) + C2, r0 = (vse << 1) + C2;     // even columns: B, R
        const uint32_t g1 = hmo + vso + C2, r1 = hvo + C2;             // odd columns: G, R
        c4[0][0] = word_and_x4<0>(b0); c4[0][1] = word_x4<0>(me); c4[0][2] = word_and_x4<0>(r0);
        c4[2][0] = word_and_x4<1>(b0); c4[2][1] = word_x4<1>(me); c4[2][2] = word_and_x4<1>(r0);
        c4[1][0] = word_x4<0>(mo); c4[1][1] = word_and_x4<0>(g1); c4[1][2] = word_and_x4<0>(r1);
        c4[3][0] = word_x4<1>(mo); c4[3][1] = word_and_x4<1>(g1); c4[3][2] = word_and_x4<1>(r1);
    }
    if (border_strip) {  // OpenCV leaves no interpolated border: column 0 copies column 1, column w-1 copies w-2
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (first_col) c4[0][c] = c4[1][c];
            if (last_col) c4[3][c] = c4[2][c];
        }
    }
    uint32_t vv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) hsv_px<false>(c4[k][0], c4[k][1], c4[k][2], tab, o.hs[k], vv[k]);
    o.v[0] = (vv[0] | (vv[1] << 16)) >> 2;
    o.v[1] = (vv[2] | (vv[3] << 16)) >> 2;
}

__device__ __forceinline__ void pairs_to_row(const Pairs &p, RowBGR &o)
{
    o.c01[0] = __builtin_amdgcn_perm(p.g[0], p.b[0], 0x05040100u);
    o.c01[1] = __builtin_amdgcn_perm(p.g[1], p.b[1], 0x05040100u);
    o.c01[2] = __builtin_amdgcn_perm(p.g[0], p.b[0], 0x07060302u);
    o.c01[3] = __builtin_amdgcn_perm(p.g[1], p.b[1], 0x07060302u);
    o.c2[0] = __builtin_amdgcn_perm(p.r[1], p.r[0], 0x05040100u);
    o.c2[1] = __builtin_amdgcn_perm(p.r[1], p.r[0], 0x07060302u);
    scale16(o);
}

// vertical 1-4-6-4-1 on packed fields.  OpenCV's rounding constant (+128 on the 8.8 result) is NOT added here: the
// host subtracts it from the inRange bounds instead (launch_fused), which saves a three-input add per register.
__device__ __forceinline__ uint32_t vert5(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4)
{
    uint32_t t = a0 + a4;
    t += (a1 + a3) << 2;
    return mad_u24(a2, 6u, t);  // a2 <= 0x00FF00FF < 2^24
}
// horizontal 1-4-6-4-1: a,b = k-2,k+2; c,d = k-1,k+1; e = k
__device__ __forceinline__ uint32_t horz5(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e)
{
    uint32_t t = a + b;
    t += (c + d + e) << 2;
    return t + (e << 1);
}

// same with the outer pair already summed
__device__ __forceinline__ uint32_t horz5s(uint32_t ab, uint32_t c, uint32_t d, uint32_t e)
{
    uint32_t t, u, r;  // ab + 4(c+d+e) + 2e in three ops (the compiler's own choice takes four)
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(t) : "v"(c), "v"(d), "v"(e));
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(u) : "v"(e), "v"(ab));
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(r) : "v"(t), "v"(u));
    return r;
}

// 3x3 rect erode/dilate (or identity) on a 4-col x 8-row bit tile; causal in rows (one-row delay).
// Tile layout: row position p (0 = oldest) at bits 4*(7-p)..+3, column c at bit c of the nibble.
// `inv` = bits outside the image (neutral for the op).  prev = this stage's previous horizontal result.
__device__ __forceinline__ uint32_t morph3(uint32_t cur, uint32_t &prev, int kind, uint32_t inv)
{
    if (kind == 2) {
        uint32_t out = alignbit(prev, cur, 4);
        prev = cur;
        return out;
    }
    const uint32_t km = kind ? 0xFFFFFFFFu : 0u;  // dilate(X) = ~erode(~X)
    const uint32_t x = (cur ^ km) | inv;
    const uint32_t xl = from_left(x), xr = from_right(x);
    const uint32_t l = ((x << 1) & 0xEEEEEEEEu) | ((xl >> 3) & 0x11111111u);
    const uint32_t r = ((x >> 1) & 0x77777777u) | ((xr << 3) & 0x88888888u);
    const uint32_t hh = x & l & r;
    const uint32_t out = hh & alignbit(prev, hh, 4) & alignbit(prev, hh, 8);
    prev = hh;
    return out ^ km;
}

__device__ __forceinline__ uint32_t rows_outside(int y0, int lo, int hi)  // nibble mask of positions with y not in [lo,hi)
{
    if (y0 >= lo && y0 + 8 <= hi) return 0u;
    uint32_t m = 0;
#pragma unroll
    for (int p = 0; p < 8; p++)
        if (y0 + p < lo || y0 + p >= hi) m |= 0xFu << (4 * (7 - p));
    return m;
}

// the owned bytes of a lane that straddles an ownership boundary (px = address of the lane's pixel 0)
__device__ __forceinline__ void store_part(u8 *px, uint32_t bytes, uint32_t own)
{
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (own & (1u << k)) px[k] = (u8)(bytes >> (8 * k));
}

// 8x8 nibble transpose across the 8 lanes of a group (lane j's nibble n <-> lane n's nibble j): three butterfly stages,
// the partner's value through ds_swizzle (LDS crossbar, no VALU), the merge one v_perm / v_bfi with per-lane constants.
struct TransposeConsts {
    uint32_t sel4, sel2, m1, rot1;
};
__device__ __forceinline__ TransposeConsts transpose_consts(int lane)
{
    TransposeConsts c;
    c.sel4 = (lane & 4) ? 0x03020706u : 0x05040100u;  // v_perm_b32(partner, own, sel): bytes 0-3 = own, 4-7 = partner
    c.sel2 = (lane & 2) ? 0x03070105u : 0x06020400u;
    c.m1 = (lane & 1) ? 0xF0F0F0F0u : 0x0F0F0F0Fu;    // nibbles this lane keeps in the last stage
    c.rot1 = (lane & 1) ? 4u : 28u;                    // partner rotated right by this: >> 4 (odd lanes), << 4 (even lanes)
    return c;
}
__device__ __forceinline__ uint32_t transpose_nibbles8(uint32_t v, const TransposeConsts &c)
{
    uint32_t q = (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);  // lane ^ 4
    v = __builtin_amdgcn_perm(q, v, c.sel4);
    q = (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x081F);  // lane ^ 2
    v = __builtin_amdgcn_perm(q, v, c.sel2);
    q = (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x041F);  // lane ^ 1
    return (v & c.m1) | (alignbit(q, q, (int)c.rot1) & ~c.m1);
}

// MODE 0: whole chain (up to two 3x3 ops, moments, final mask).  MODE 1: front end only: the inRange output is
// written as 0/255 bytes to a.masks (the bit-plane back end, morphbits.hip, runs the morph list and the moments).
// MODE 2: the same front end, but the inRange output goes out as bit planes (a.bits, 1/8 B per px and range): the
// kernel holds it as bits anyway, and the back end wants bits (w % 4 == 0 only: strip borders are byte borders then).
// BAYER: the frames are BayerRG8 mosaics (1 B/px); every fed row loads the three mosaic rows it is interpolated from.
template <int PRE_K, int POST_K, bool RGB, int NR, int PF, bool M2, int MODE, bool BAYER>
// The one-range Bayer chain without pre-blur (BASELINE config 3): the compiler's own allocation is 97 VGPRs, one over the 96
// that still give 5 waves per SIMD; asked to keep 5 it fits them without scratch (the other Bayer variants would spill)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BAYER && NR == 1 && !M2 && PRE_K == 1 ? 5 : 1)))
void k_fused_chain(const Args a)
{
    __shared__ int s_tab[512 + 16];
    s_tab[threadIdx.x] = a.tables[threadIdx.x] * 4;  // 4*sdiv (see hsv_px)
    s_tab[threadIdx.x + 256] = a.tables[threadIdx.x + 256] * 4;  // 4*hdiv180
    if (threadIdx.x < 16)  // nibble -> four mask bytes (bit c -> byte c = 0x00 / 0xFF)
        s_tab[512 + threadIdx.x] = (int)(((threadIdx.x & 1) ? 0xFFu : 0u) | ((threadIdx.x & 2) ? 0xFF00u : 0u) |
                                         ((threadIdx.x & 4) ? 0xFF0000u : 0u) | ((threadIdx.x & 8) ? 0xFF000000u : 0u));
    __syncthreads();
    const char *tab = reinterpret_cast<const char *>(s_tab);

    const int lane = threadIdx.x & 63;
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (item >= a.total_items) return;
    const int strip = item % a.n_strips;
    const int rest = item / a.n_strips;
    const int seg = rest % a.n_segs;
    const int frame = rest / a.n_segs;
    const int w = a.w, h = a.h;
    const int ys = seg * a.seg_rows;
    const int ye = min(ys + a.seg_rows, h);
    constexpr int HL = 1;  // halo lanes per side
    constexpr int SV = strip_valid(BAYER);
    // Strips are left-aligned (lane columns = 0 mod 4, so the image's left border is a lane boundary) except the last
    // one when w % 4 != 0: that one is right-aligned (a lane boundary at w).  Every strip is then wrong only within
    // the blur/morph reach of the border it is NOT aligned to, and never owns columns there (launch_fused).
    const bool last_strip = strip == a.n_strips - 1;
    const int base = last_strip ? a.last_base : strip * SV;
    const int x0 = base - 4 * HL + lane * 4;
    const int own_lo = last_strip ? a.split_x : strip * SV;
    const int own_hi = last_strip ? w : min((strip + 1) * SV, a.split_x);
    const int xl = min(max(x0, 0), w - 4);
    const bool in_image = x0 >= 0 && x0 < w;
    uint32_t own = 0;  // bit k: this lane's pixel k is an output of this strip
    if (lane >= HL && lane <= 63 - HL) {
#pragma unroll
        for (int k = 0; k < 4; k++) own |= (x0 + k >= own_lo && x0 + k < own_hi) ? (1u << k) : 0u;
    }
    const bool lane_out = own == 0xFu;              // the common case: one dword store per row
    const bool lane_part = own != 0u && own != 0xFu;  // ownership boundary inside the lane (w % 4 != 0 only)
    const uint32_t own_rep = own * 0x11111111u;
    const bool strip_left = strip == 0;
    const bool strip_right = base + 256 - 4 * HL >= w;
    const bool lb_halo = x0 == -4, rb_halo = x0 == w;  // halo lanes just outside the left / right image border
    const uint32_t colinv = in_image ? 0u : 0xFFFFFFFFu;

    // uniform (SGPR) frame pointer + 32-bit per-lane byte offset: loads use the saddr form
    const u8 *fptr = a.frames + (size_t)frame * a.frame_stride;
    // Bayer: the lane's 8-byte window starts one column left of its pixels; at the row's ends it is the row's first / last
    // 8 bytes, shifted into place after the load (bayer_window_fix; w >= 8 is checked at launch)
    const bool win_lo = BAYER && xl == 0;
    const int win_hi_sh8 = (BAYER && xl + 7 > w) ? 8 * (xl + 7 - w) : 0;
    const uint32_t lane_off = BAYER ? (uint32_t)(win_lo ? 0 : win_hi_sh8 ? w - 8 : xl - 1) : (uint32_t)xl * 3u;
    const bool first_col = x0 == 0, last_col = x0 == w - 4;  // Bayer: the image's border columns copy their neighbours
    const uint32_t rs = (uint32_t)a.row_stride;  // byte offsets inside one frame fit 32 bits (checked at launch)

    constexpr int LEAD = (POST_K == 5 ? 2 : 0) + (PRE_K == 3 ? 1 : 0);  // rows of look-ahead the blurs need
    constexpr int MH = MODE == 0 ? 2 : 0;      // row delay of the two morphology stages
    const int t0 = ys - MH;                    // first inRange row of the tile stream
    const int r_first = t0 - LEAD;             // first fed row
    const int n_batches = (ye - ys + 2 * MH + 7) / 8;
    // kept as a scalar integer: as a bool the compiler re-materialises it through a VGPR on every row
    const int border_strip = __builtin_amdgcn_readfirstlane((int)(strip_left || strip_right));  // int, not bool: see above

    // Bayer: the centre rows of consecutive fed rows move by +1 (the usual case), 0 or -1 (the frame's border rows copy
    // their neighbours; look-ahead beyond the frame reflects), so the three mosaic windows of a fed row are the previous
    // fed row's, shifted by one: ONE window per fed row is prefetched -- the row that moves in, below or above -- and the
    // other two stay in registers (bwin).  The segment's first fed row is primed before the prologue.
    uint32_t raw[8][BAYER ? 2 : 3];
    uint32_t bwin[6] = {0, 0, 0, 0, 0, 0};  // the previous fed row's windows: above, centre, below
    int bwin_c = 0;                         // its centre row (uniform)
    Row ring[8];
    RowBGR rawp[PRE_K == 3 ? 8 : 1];
    // raw row in slot `sl` -> HSV row in ring[sl] (with PRE_K == 3: the HSV of the 3x3-blurred row before it)
    auto bayer_row = [&](int f) { return min(max(reflect_row(r_first + f, h), 1), h - 2); };  // border rows copy too
    auto convert = [&](int sl, bool have_rows, int f) {
        if constexpr (BAYER) {
            const int c = bayer_row(f);
            uint32_t nlo = raw[sl][0], nhi = raw[sl][1];
            if (border_strip) bayer_window_fix(nlo, nhi, win_lo, win_hi_sh8);
            if (c > bwin_c) {  // uniform; the usual case: the window below moves in
                bwin[0] = bwin[2]; bwin[1] = bwin[3]; bwin[2] = bwin[4]; bwin[3] = bwin[5]; bwin[4] = nlo; bwin[5] = nhi;
            } else if (c < bwin_c) {  // the window above moves in
                bwin[4] = bwin[2]; bwin[5] = bwin[3]; bwin[2] = bwin[0]; bwin[3] = bwin[1]; bwin[0] = nlo; bwin[1] = nhi;
            }
            bwin_c = c;
        }
        if constexpr (PRE_K == 3) {
            if constexpr (BAYER) {
                Pairs bp;
                demosaic_row(bwin, bayer_row(f) & 1, border_strip, first_col, last_col, bp);
                pairs_to_row(bp, rawp[sl & 7]);
            } else {
                unpack_row(raw[sl], rawp[sl & 7]);
            }
            if (have_rows)
                preblur_hsv_row<RGB>(rawp[(sl + 6) & 7], rawp[(sl + 7) & 7], rawp[sl & 7], border_strip, lb_halo, rb_halo, tab, ring[sl]);
        } else if constexpr (BAYER) {
            demosaic_hsv_row(bwin, bayer_row(f) & 1, border_strip, first_col, last_col, tab, ring[sl]);
        } else {
            hsv_row<RGB>(raw[sl], tab, ring[sl]);
        }
    };
    auto load_row = [&](int slot, int f) {
        if constexpr (BAYER) {
            // the window that moves in: below the centre when the centre row advances (or at the primed start), above it
            // when it steps back; 8 bytes at any byte alignment: fine on gfx950 (scripts/ubench/unaligned.hip)
            const int c = bayer_row(f), cp = f > 0 ? bayer_row(f - 1) : c - 1;
            const int row = c + (c < cp ? -1 : 1);  // c == cp: loaded and dropped
            const uint32_t *p = reinterpret_cast<const uint32_t *>(fptr + (uint32_t)row * rs + lane_off);
            raw[slot][0] = p[0];
            raw[slot][1] = p[1];
        } else {
            const int r = reflect_row(r_first + f, h);
            const uint32_t off = (uint32_t)r * rs + lane_off;
            const uint32_t *p = reinterpret_cast<const uint32_t *>(fptr + off);
            raw[slot][0] = p[0];
            raw[slot][1] = p[1];
            raw[slot][2] = p[2];
        }
    };

    // fed row f uses raw/ring slot (f + 8 - 2*LEAD) & 7, so that the main step j = f - 2*LEAD uses slot j & 7
    constexpr int S0 = (8 - 2 * LEAD) & 7;
    // Vertical 1-4-6-4-1 as four running [1 1] steps down the rows ([1 1]^4 = 1 4 6 4 1): per register and row four plain
    // v_add_u32 on the previous row's value and the three running sums -- the cheap issue class -- instead of
    // add-shift, three-input add and multiply-add on a ring of five rows.  State: the previous row and its three sums
    // (24 registers, against 30 for the ring's five live rows); the first four rows pushed only warm it up.
    uint32_t cas[4][6];  // [0]: previous row; [1..3]: previous first / second / third sums; H|S of pixels 0..3, then the V pairs
#pragma unroll
    for (int i = 0; i < 24; i++) cas[i / 6][i % 6] = 0;
    auto cascade = [&](const Row &row, uint32_t (&c_hs)[4], uint32_t (&c_v)[2]) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const uint32_t a = i < 4 ? row.hs[i] : row.v[i - 4];
            const uint32_t s1 = a + cas[0][i], s2 = s1 + cas[1][i], s3 = s2 + cas[2][i], out = s3 + cas[3][i];
            cas[0][i] = a; cas[1][i] = s1; cas[2][i] = s2; cas[3][i] = s3;
            if (i < 4) c_hs[i] = out; else c_v[i - 4] = out;
        }
    };
    if constexpr (BAYER) {  // prime: as if a fed row with centre c0 - 1 had come before (its window above is never used)
        const int c0 = bayer_row(0);
        const uint32_t *pm = reinterpret_cast<const uint32_t *>(fptr + (uint32_t)(c0 - 1) * rs + lane_off);
        const uint32_t *pb = reinterpret_cast<const uint32_t *>(fptr + (uint32_t)c0 * rs + lane_off);
        bwin[2] = pm[0]; bwin[3] = pm[1]; bwin[4] = pb[0]; bwin[5] = pb[1];
        if (border_strip) {
            bayer_window_fix(bwin[2], bwin[3], win_lo, win_hi_sh8);
            bayer_window_fix(bwin[4], bwin[5], win_lo, win_hi_sh8);
        }
        bwin_c = c0 - 1;
    }
#pragma unroll
    for (int q = 0; q < PF; q++) load_row((S0 + q) & 7, q);
#pragma unroll
    for (int q = 0; q < 2 * LEAD; q++) {  // prologue: fill the rings
        load_row((S0 + q + PF) & 7, q + PF);
        convert((S0 + q) & 7, q >= (PRE_K == 3 ? 2 : 0), q);
        if (POST_K == 5 && IPF_CASCADE && q >= (PRE_K == 3 ? 2 : 0)) {
            uint32_t d_hs[4], d_v[2];
            cascade(ring[(S0 + q) & 7], d_hs, d_v);
        }
    }

    const TransposeConsts tc = transpose_consts(lane);  // MODE 2 only
    uint32_t acc[NR], prev1[NR], prev2[NR];
    uint32_t m00[NR], m10[NR], m01[NR];
    unsigned long long m20[NR], m11[NR], m02[NR];  // second-order sums (only with M2)
#pragma unroll
    for (int r = 0; r < NR; r++) acc[r] = prev1[r] = prev2[r] = m00[r] = m10[r] = m01[r] = 0, m20[r] = m11[r] = m02[r] = 0;

    const size_t plane = (size_t)w * h;
    u8 *mptr = a.masks ? a.masks + (size_t)frame * a.nr * plane : nullptr;  // uniform
    const uint32_t mask_off = (uint32_t)max(x0, 0);

    for (int b = 0; b < n_batches; b++) {
#pragma unroll
        for (int rr = 0; rr < 8; rr++) {
            const int f = 2 * LEAD + b * 8 + rr;  // fed index of this step's newest row
            load_row((rr + PF) & 7, f + PF);
            convert(rr, true, f);
            uint32_t o_hs[4], o_v[2];
            if (POST_K == 5) {
                uint32_t c_hs[4], c_v[2];
                if (IPF_CASCADE) {
                    cascade(ring[rr], c_hs, c_v);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        c_hs[k] = vert5(ring[(rr + 4) & 7].hs[k], ring[(rr + 5) & 7].hs[k], ring[(rr + 6) & 7].hs[k],
                                        ring[(rr + 7) & 7].hs[k], ring[rr].hs[k]);
#pragma unroll
                    for (int k = 0; k < 2; k++)
                        c_v[k] = vert5(ring[(rr + 4) & 7].v[k], ring[(rr + 5) & 7].v[k], ring[(rr + 6) & 7].v[k],
                                       ring[(rr + 7) & 7].v[k], ring[rr].v[k]);
                }
                if (border_strip) {
                    // BORDER_REFLECT_101 at the image's left/right edge: give the halo lane next to the border
                    // lane the mirrored columns, so the DPP fetches below read col -1 = col 1, col -2 = col 2
                    // (and col w = col w-2, col w+1 = col w-3).  Those halo lanes never produce output.
                    const uint32_t m2b = alignbit(c_v[1], c_v[0], 16);  // (V1,V2)
                    const uint32_t sw = alignbit(m2b, m2b, 16);         // (V2,V1)
                    const uint32_t a2 = from_right(c_hs[2]), a1 = from_right(c_hs[1]), asw = from_right(sw);
                    const uint32_t b2 = from_left(c_hs[2]), b1 = from_left(c_hs[1]), bsw = from_left(sw);
                    if (lb_halo) { c_hs[2] = a2; c_hs[3] = a1; c_v[1] = asw; }
                    if (rb_halo) { c_hs[0] = b2; c_hs[1] = b1; c_v[0] = bsw; }
                }
                const uint32_t vl = from_left(c_v[1]), vr = from_right(c_v[0]);
                const uint32_t m2 = alignbit(c_v[1], c_v[0], 16);  // (V1,V2)
                if (IPF_CASCADE) {
                    // horizontal 1-4-6-4-1 of the H|S registers the same way: pair sums p, their pair sums q (1 2 1), theirs r
                    // (1 3 3 1), out = r[k-1] + r[k]; the terms that cross lanes come with the add (v_add_u32_dpp).
                    // 16 plain adds + 4 DPP adds against 4 DPP ops + 2 adds + 12 three-input ops
                    const uint32_t p0 = c_hs[0] + c_hs[1], p1 = c_hs[1] + c_hs[2], p2 = c_hs[2] + c_hs[3];
                    const uint32_t pm1 = add_from_left(c_hs[3], c_hs[0]);   // c[-1] + c[0]
                    const uint32_t p3 = add_from_right(c_hs[0], c_hs[3]);   // c[3] + c[4]
                    const uint32_t q0 = pm1 + p0, q1 = p0 + p1, q2 = p1 + p2, q3 = p2 + p3;
                    const uint32_t qm1 = add_from_left(p2, pm1);            // p[-2] + p[-1]: lane-1's p2 is c[-2] + c[-1]
                    const uint32_t q4 = add_from_right(p0, p3);             // p[3] + p[4]: lane+1's p0 is c[4] + c[5]
                    const uint32_t rm1 = qm1 + q0, r0 = q0 + q1, r1 = q1 + q2, r2 = q2 + q3, r3 = q3 + q4;
                    o_hs[0] = rm1 + r0;
                    o_hs[1] = r0 + r1;
                    o_hs[2] = r1 + r2;
                    o_hs[3] = r2 + r3;
                } else {
                    const uint32_t l3 = from_left(c_hs[3]), r0 = from_right(c_hs[0]);
                    // the k-2 / k+2 terms that cross lanes are single-use: written as plain adds so the DPP fetch
                    // folds into the add (v_add_u32_dpp)
                    o_hs[0] = horz5s(add_from_left(c_hs[2], c_hs[2]), l3, c_hs[1], c_hs[0]);
                    o_hs[1] = horz5s(l3 + c_hs[3], c_hs[0], c_hs[2], c_hs[1]);
                    o_hs[2] = horz5s(c_hs[0] + r0, c_hs[1], c_hs[3], c_hs[2]);
                    o_hs[3] = horz5s(add_from_right(c_hs[1], c_hs[1]), c_hs[2], r0, c_hs[3]);
                }
                const uint32_t m1 = alignbit(c_v[0], vl, 16);  // (V-1,V0)
                const uint32_t m3 = alignbit(vr, c_v[1], 16);  // (V3,V4)
                o_v[0] = horz5s(vl + c_v[1], m1, m2, c_v[0]);
                o_v[1] = horz5s(c_v[0] + vr, m2, m3, c_v[1]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) o_hs[k] = ring[rr].hs[k];
                o_v[0] = ring[rr].v[0];
                o_v[1] = ring[rr].v[1];
            }
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (NR > 2 && r >= a.nr) break;  // capacity 4 / 8 variants: only a.nr ranges are live
                // V: (v - lo) <= (hi - lo) as unsigned 16-bit fields: one packed subtract per pair, one compare per pixel
                const uint32_t vlo2 = a.v_lo[r], vrng = a.v_hi[r];  // (lo | lo << 16), hi - lo
                uint32_t dv[2];
                asm("v_pk_sub_u16 %0, %1, %2" : "=v"(dv[0]) : "v"(o_v[0]), "s"(vlo2));
                asm("v_pk_sub_u16 %0, %1, %2" : "=v"(dv[1]) : "v"(o_v[1]), "s"(vlo2));
                uint32_t t = acc[r];
#pragma unroll
                for (int k = 3; k >= 0; k--) {
                    const uint32_t vk = (k & 1) ? (dv[k >> 1] >> 16) : (dv[k >> 1] & 0xFFFFu);
                    // two compares write lane masks; the AND is scalar work
                    const unsigned long long ok = __builtin_amdgcn_ballot_w64(pk_clamp(o_hs[k], a.hs_lo[r], a.hs_hi[r]) == o_hs[k]) &
                                                  __builtin_amdgcn_ballot_w64(vk <= vrng);
                    t = shift_in(t, ok);
                }
                acc[r] = t;
            }
        }
        const int yt = t0 + b * 8;  // image row of tile position 0 at the inRange stage
        if (MODE == 2) {
            // ---- front end, bits out.  Lane l <- lane l+1, so that lane l stands for columns base + 4l and every second lane
            // starts a byte of the bitmap.  Then the 8 lanes of a group trade nibbles so that lane i holds the 32-px word of
            // tile row 7 - i. ----
            const int li = lane & 7, xg = base + 32 * (lane >> 3);
            const int y = yt + 7 - li;
            const int fb = max(0, (own_lo - xg) >> 3), lb = min(4, (own_hi - xg + 7) >> 3);  // bytes [fb, lb) of the word are this strip's
            const bool store = y >= ys && y < ye && lb > fb;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (NR > 2 && r >= a.nr) break;  // capacity 4 / 8 variants: only a.nr ranges are live
                const uint32_t word = transpose_nibbles8(from_right(acc[r]), tc);
                if (store) {
                    u8 *dst = reinterpret_cast<u8 *>(a.bits + (((size_t)frame * a.nr + r) * h + y) * a.bit_words) + (xg >> 3);
                    if (fb == 0 && lb == 4) {
                        *reinterpret_cast<uint32_t *>(dst) = word;  // byte-aligned, not always dword-aligned: fine on gfx950
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (k >= fb && k < lb) dst[k] = (u8)(word >> (8 * k));
                    }
                }
            }
            continue;
        }
        if (MODE == 1) {
            // ---- front end only: the inRange rows go out as 0/255 bytes ----
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (NR > 2 && r >= a.nr) break;  // capacity 4 / 8 variants: only a.nr ranges are live
                const uint32_t t = acc[r];
                u8 *rowp = mptr + (size_t)r * plane + (size_t)yt * w;  // uniform
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    const int y = yt + p;
                    if (y >= ys && y < ye && (lane_out || lane_part)) {
                        const int sh = 4 * (7 - p) - 2;
                        const uint32_t idx = (sh >= 0 ? (t >> sh) : (t << 2)) & 0x3Cu;
                        const uint32_t bytes = *reinterpret_cast<const uint32_t *>(tab + 2048 + idx);
                        if (lane_out)
                            *reinterpret_cast<uint32_t *>(rowp + (size_t)p * w + mask_off) = bytes;
                        else
                            store_part(rowp + ((long long)p * w + x0), bytes, own);
                    }
                }
            }
            continue;
        }
        // ---- one 4x8 tile per range is complete: morphology, store, moments ----
        const uint32_t inv1 = colinv | rows_outside(yt, 0, h);
        const uint32_t inv2 = colinv | rows_outside(yt - 1, 0, h);
        const uint32_t keep = own_rep & ~rows_outside(yt - 2, ys, ye);
#pragma unroll
        for (int r = 0; r < NR; r++) {
            if (NR > 2 && r >= a.nr) break;  // capacity 4 / 8 variants: only a.nr ranges are live
            uint32_t t = morph3(acc[r], prev1[r], a.op0, inv1);
            t = morph3(t, prev2[r], a.op1, inv2);
            const uint32_t fm = t & keep;
            if (mptr) {
                // tile row p -> nibble*4 -> LDS lookup of the four mask bytes (two VALU ops + one ds_read per row)
                auto expand = [&](uint32_t tile, int p) {
                    const int sh = 4 * (7 - p) - 2;
                    const uint32_t idx = (sh >= 0 ? (tile >> sh) : (tile << 2)) & 0x3Cu;
                    return *reinterpret_cast<const uint32_t *>(tab + 2048 + idx);
                };
                u8 *rowp = mptr + (size_t)r * plane + (size_t)(yt - 2) * w;  // uniform
                if (yt - 2 >= ys && yt - 2 + 8 <= ye) {  // whole tile inside the segment (the common case)
                    if (lane_out) {
#pragma unroll
                        for (int p = 0; p < 8; p++)
                            *reinterpret_cast<uint32_t *>(rowp + (size_t)p * w + mask_off) = expand(t, p);
                    } else if (lane_part) {
#pragma unroll
                        for (int p = 0; p < 8; p++) store_part(rowp + ((long long)p * w + x0), expand(t, p), own);
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < 8; p++) {
                        const int y = yt - 2 + p;
                        if (y >= ys && y < ye && lane_out)
                            *reinterpret_cast<uint32_t *>(rowp + (size_t)p * w + mask_off) = expand(t, p);
                        else if (y >= ys && y < ye && lane_part)
                            store_part(rowp + ((long long)p * w + x0), expand(t, p), own);
                    }
                }
            }
            const uint32_t cnt = __popc(fm);
            const uint32_t csum = __popc(fm & 0xAAAAAAAAu) + 2 * __popc(fm & 0xCCCCCCCCu);
            const uint32_t q = __popc(fm & 0xF0F0F0F0u) + 2 * __popc(fm & 0xFF00FF00u) + 4 * __popc(fm & 0xFFFF0000u);
            m00[r] += cnt;
            m10[r] = mad_u24((uint32_t)x0, cnt, m10[r]) + csum;   // x0 < 2^24 wherever cnt != 0
            m01[r] = mad_u24((uint32_t)(yt + 5), cnt, m01[r]) - q;  // sum of rows = (yt-2+7)*cnt - q; yt+5 > 0 wherever cnt != 0
            if (M2 && __builtin_amdgcn_ballot_w64(fm != 0)) {
                // x = x0 + c (c = bit & 3), y = Y7 - u (u = bit >> 2): all second-order sums from popcounts of the tile
                const uint32_t CA = 0xAAAAAAAAu, CB = 0xCCCCCCCCu, U0 = 0xF0F0F0F0u, U1 = 0xFF00FF00u, U2 = 0xFFFF0000u;
                const uint32_t c2 = __popc(fm & CA) + 4 * __popc(fm & CB) + 4 * __popc(fm & CA & CB);              // sum c^2
                const uint32_t u2 = __popc(fm & U0) + 4 * __popc(fm & U1) + 16 * __popc(fm & U2) +
                                    4 * __popc(fm & U0 & U1) + 8 * __popc(fm & U0 & U2) + 16 * __popc(fm & U1 & U2);  // sum u^2
                const uint32_t cu = __popc(fm & CA & U0) + 2 * __popc(fm & CA & U1) + 4 * __popc(fm & CA & U2) +
                                    2 * __popc(fm & CB & U0) + 4 * __popc(fm & CB & U1) + 8 * __popc(fm & CB & U2);   // sum c*u
                const unsigned long long X = (unsigned long long)(uint32_t)max(x0, 0), Y7 = (unsigned long long)(uint32_t)max(yt + 5, 0);
                const unsigned long long sy = Y7 * cnt - q;      // sum y
                const unsigned long long scy = Y7 * csum - cu;   // sum c*y
                m20[r] += X * X * cnt + 2 * X * csum + c2;
                m11[r] += X * sy + scy;
                m02[r] += Y7 * Y7 * cnt + u2 - 2 * Y7 * q;
            }
        }
    }

    if (MODE != 0) return;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        if (NR > 2 && r >= a.nr) break;  // capacity 4 / 8 variants: only a.nr ranges are live
        unsigned long long s0 = m00[r], s1 = m10[r], s2 = m01[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s0 += __shfl_down(s0, off, 64);
            s1 += __shfl_down(s1, off, 64);
            s2 += __shfl_down(s2, off, 64);
        }
        if (lane == 0) {
            unsigned long long *o = a.mom + ((size_t)frame * a.nr + r) * 6;
            if (s0) {
                atomicAdd(o + 0, s0);
                atomicAdd(o + 1, s1);
                atomicAdd(o + 2, s2);
            }
        }
        if (M2) {
            unsigned long long t0 = m20[r], t1 = m11[r], t2 = m02[r];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                t0 += __shfl_down(t0, off, 64);
                t1 += __shfl_down(t1, off, 64);
                t2 += __shfl_down(t2, off, 64);
            }
            if (lane == 0 && s0) {
                unsigned long long *o = a.mom + ((size_t)frame * a.nr + r) * 6;
                atomicAdd(o + 3, t0);
                atomicAdd(o + 4, t1);
                atomicAdd(o + 5, t2);
            }
        }
    }
}


// One translation unit per group of instantiations (the build compiles them in parallel): Bayer or not, small (1, 2)
// or large (4, 8) range capacity.  Each dispatches on (pre_k, post_k, rgb, nr capacity, second order, mode).
#ifndef IPF_PF
#define IPF_PF 3  // rows of raw-load prefetch ahead of the row being converted
#endif

struct Variant {
    int pre_k, post_k, rgb, cap, second_order, mode;
    int *waves_per_simd;  // non-null: do not launch, report how many waves of this instantiation fit one SIMD
};
hipError_t launch_variant_plain_small(const Variant &v, const Args &a, dim3 grid, hipStream_t stream);
hipError_t launch_variant_plain_large(const Variant &v, const Args &a, dim3 grid, hipStream_t stream);
hipError_t launch_variant_bayer_small(const Variant &v, const Args &a, dim3 grid, hipStream_t stream);
hipError_t launch_variant_bayer_large(const Variant &v, const Args &a, dim3 grid, hipStream_t stream);

// body shared by the four: NRA / NRB are the two capacities of the group
#define IPF_DEFINE_LAUNCH_VARIANT(NAME, BAYERV, NRA, NRB)                                                              \
    hipError_t NAME(const Variant &v, const Args &a, dim3 grid, hipStream_t stream)                                    \
    {                                                                                                                   \
        const dim3 block(256);                                                                                          \
        auto go = [&](auto pre, auto post, auto rgb, auto cap) {                                                        \
            constexpr int PRE = decltype(pre)::value, POST = decltype(post)::value, CAP = decltype(cap)::value;         \
            constexpr bool RGBV = decltype(rgb)::value;                                                                 \
            auto run = [&](auto kernel) {                                                                               \
                if (v.waves_per_simd) {                                                                                 \
                    int nb = 0; /* 256-thread blocks per CU == waves per SIMD */                                        \
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, 0) != hipSuccess) nb = 0;        \
                    *v.waves_per_simd = nb;                                                                             \
                } else {                                                                                                \
                    hipLaunchKernelGGL(kernel, grid, block, 0, stream, a);                                              \
                }                                                                                                       \
            };                                                                                                          \
            if (v.mode == 2) run(k_fused_chain<PRE, POST, RGBV, CAP, IPF_PF, false, 2, BAYERV>);                        \
            else if (v.mode == 1) run(k_fused_chain<PRE, POST, RGBV, CAP, IPF_PF, false, 1, BAYERV>);                   \
            else if (v.second_order) run(k_fused_chain<PRE, POST, RGBV, CAP, IPF_PF, true, 0, BAYERV>);                 \
            else run(k_fused_chain<PRE, POST, RGBV, CAP, IPF_PF, false, 0, BAYERV>);                                    \
        };                                                                                                              \
        using std::integral_constant;                                                                                   \
        auto with_cap = [&](auto pre, auto post, auto rgb) {                                                            \
            if (v.cap == NRA) go(pre, post, rgb, integral_constant<int, NRA>{});                                        \
            else go(pre, post, rgb, integral_constant<int, NRB>{});                                                     \
        };                                                                                                              \
        auto with_rgb = [&](auto pre, auto post) {                                                                      \
            if (!BAYERV && v.rgb) with_cap(pre, post, integral_constant<bool, !BAYERV>{});                              \
            else with_cap(pre, post, integral_constant<bool, false>{});                                                 \
        };                                                                                                              \
        if (v.pre_k == 3) {                                                                                             \
            if (v.post_k == 5) with_rgb(integral_constant<int, 3>{}, integral_constant<int, 5>{});                      \
            else with_rgb(integral_constant<int, 3>{}, integral_constant<int, 1>{});                                    \
        } else {                                                                                                        \
            if (v.post_k == 5) with_rgb(integral_constant<int, 1>{}, integral_constant<int, 5>{});                      \
            else with_rgb(integral_constant<int, 1>{}, integral_constant<int, 1>{});                                    \
        }                                                                                                               \
        return hipGetLastError();                                                                                       \
    }

}  // namespace ipf
