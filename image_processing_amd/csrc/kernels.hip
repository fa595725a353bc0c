// Stage-by-stage gfx950 kernels: one per cv:: call on the reference's path.  They serve the
// single-stage entry points of the C ABI, arbitrary pipelines (any ksize / element / range
// count) and stage-wise parity; the fused chain kernel for the headline presets is in fused.hip.
//
// Arithmetic follows SURVEY.md Appendix A (OpenCV's published 8-bit fixed point); reference call
// sites are cited per kernel.  All work is u8 / integer: HBM-bound, no MFMA.
#include "kernels.h"

namespace ipk {

// ---------------------------------------------------------------------------------------------
// shared device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
    return p;
}

// One pixel of cvtColor(BGR2HSV) (objectPolygonPath.cpp:183, pose_estimate.cpp:203); returns h|s<<8|v<<16.
__device__ __forceinline__ uint32_t hsv_pixel(int b, int g, int r, const int *sdiv, const int *hdiv)
{
    int v = max(max(b, g), r);
    int vmin = min(min(b, g), r);
    int diff = v - vmin;
    int s = (diff * sdiv[v] + (1 << 11)) >> 12;
    int hh = (v == r) ? (g - b) : (v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff);
    hh = (hh * hdiv[diff] + (1 << 11)) >> 12;  // arithmetic shift
    if (hh < 0) hh += 180;
    return (uint32_t)hh | ((uint32_t)s << 8) | ((uint32_t)v << 16);
}

// ---------------------------------------------------------------------------------------------
// cvtColor BGR2HSV: 4 pixels (12 bytes) per lane, 3 dword loads / 3 dword stores when aligned.
// ---------------------------------------------------------------------------------------------
constexpr int HSV_ROWS = 32;
__global__ __launch_bounds__(256) void k_bgr2hsv(const u8 *__restrict__ src, size_t sstride, u8 *__restrict__ dst,
                                                 size_t dstride, int w, int h, int bidx,
                                                 const int *__restrict__ tables, int vec_ok)
{
    __shared__ int s_tab[512];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    s_tab[tid] = tables[tid];
    s_tab[tid + 256] = tables[tid + 256];
    __syncthreads();
    const int *sdiv = s_tab, *hdiv = s_tab + 256;
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (x4 >= w) return;
    // HSV_ROWS rows per block (the 2 KB of tables are loaded once per block), 4 at a time
#pragma unroll 2
    for (int y = blockIdx.y * HSV_ROWS + threadIdx.y; y < min(h, (int)(blockIdx.y + 1) * HSV_ROWS); y += 4) {
    const u8 *s = src + (size_t)y * sstride + (size_t)x4 * 3;
    u8 *d = dst + (size_t)y * dstride + (size_t)x4 * 3;
    if (vec_ok && x4 + 4 <= w) {
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(s);
        uint32_t w0 = s32[0], w1 = s32[1], w2 = s32[2];
        int c0[4] = {(int)(w0 & 255), (int)(w0 >> 24), (int)((w1 >> 16) & 255), (int)((w2 >> 8) & 255)};
        int c1[4] = {(int)((w0 >> 8) & 255), (int)(w1 & 255), (int)(w1 >> 24), (int)((w2 >> 16) & 255)};
        int c2[4] = {(int)((w0 >> 16) & 255), (int)((w1 >> 8) & 255), (int)(w2 & 255), (int)(w2 >> 24)};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = bidx ? hsv_pixel(c2[i], c1[i], c0[i], sdiv, hdiv) : hsv_pixel(c0[i], c1[i], c2[i], sdiv, hdiv);
        uint32_t *d32 = reinterpret_cast<uint32_t *>(d);
        d32[0] = o[0] | (o[1] << 24);
        d32[1] = (o[1] >> 8) | (o[2] << 16);
        d32[2] = (o[2] >> 16) | (o[3] << 8);
    } else {
        for (int i = 0; i < 4 && x4 + i < w; i++) {
            int b = s[3 * i + bidx], g = s[3 * i + 1], r = s[3 * i + (bidx ^ 2)];
            uint32_t o = hsv_pixel(b, g, r, sdiv, hdiv);
            d[3 * i + 0] = (u8)o;
            d[3 * i + 1] = (u8)(o >> 8);
            d[3 * i + 2] = (u8)(o >> 16);
        }
    }
    }
}

hipError_t launch_bgr2hsv(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, int bidx,
                          const int *d_tables, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return hipSuccess;
    int vec_ok = ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 4 == 0) && (sstride % 4 == 0) && (dstride % 4 == 0);
    dim3 block(64, 4), grid((w + 255) / 256, (h + HSV_ROWS - 1) / HSV_ROWS);
    hipLaunchKernelGGL(k_bgr2hsv, grid, block, 0, stream, src, sstride, dst, dstride, w, h, bidx, d_tables, vec_ok);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// GaussianBlur(src,dst,Size(k,k),0,0) on CV_8U (objectPolygonPath.cpp:182,184), any odd k<=31.
// LDS tile with reflect-101 halo, row pass to 8.8 (u16), column pass to 16.16, one rounding.
// ---------------------------------------------------------------------------------------------
constexpr int G_TWB = 192;  // tile width in BYTES (multiple of 3 so tiles start on a pixel)
constexpr int G_TH = 16;
constexpr int G_RMAX = 15;

template <int CN>
__global__ __launch_bounds__(256) void k_gaussian(const u8 *__restrict__ src, size_t sstride, u8 *__restrict__ dst,
                                                  size_t dstride, int w, int h, Taps taps)
{
    __shared__ u8 s_in[(G_TH + 2 * G_RMAX) * (G_TWB + 2 * G_RMAX * 3)];
    __shared__ uint16_t s_mid[(G_TH + 2 * G_RMAX) * G_TWB];
    const int tid = threadIdx.x;
    const int k = taps.k, r = k >> 1;
    const int LW = G_TWB + 2 * r * CN, LH = G_TH + 2 * r;
    const int p0 = blockIdx.x * (G_TWB / CN) - r;
    const int y0 = blockIdx.y * G_TH - r;
    for (int idx = tid; idx < LH * LW; idx += 256) {
        int ty = idx / LW, tb = idx - ty * LW;
        int lp = tb / CN, ch = tb - lp * CN;
        int px = reflect101(p0 + lp, w), py = reflect101(y0 + ty, h);
        s_in[idx] = src[(size_t)py * sstride + (size_t)px * CN + ch];
    }
    __syncthreads();
    for (int idx = tid; idx < LH * G_TWB; idx += 256) {
        int ty = idx / G_TWB, tb = idx - ty * G_TWB;
        const u8 *p = s_in + ty * LW + tb;
        unsigned acc = 0;
        for (int i = 0; i < k; i++) acc += (unsigned)taps.t[i] * p[i * CN];
        s_mid[idx] = (uint16_t)acc;
    }
    __syncthreads();
    const int wb = w * CN;
    for (int idx = tid; idx < G_TH * G_TWB; idx += 256) {
        int oy = idx / G_TWB, tb = idx - oy * G_TWB;
        int gy = blockIdx.y * G_TH + oy, gb = blockIdx.x * G_TWB + tb;
        if (gy < h && gb < wb) {
            const uint16_t *p = s_mid + oy * G_TWB + tb;
            unsigned acc = 0;
            for (int j = 0; j < k; j++) acc += (unsigned)taps.t[j] * p[j * G_TWB];
            dst[(size_t)gy * dstride + gb] = (u8)((acc + (1u << 15)) >> 16);
        }
    }
}

hipError_t launch_gaussian(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, int cn,
                           const Taps &taps, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return hipSuccess;
    if ((cn != 1 && cn != 3) || taps.k < 1 || taps.k > 31 || !(taps.k & 1)) return hipErrorInvalidValue;
    static const bool generic_only = getenv("IMGPROC_GAUSS_GENERIC") != nullptr;  // test hook: tile kernel only
    {  // k >= 7: the matrix-core kernel (IMGPROC_GAUSS_MFMA=0: A/B and tests; read per call, cheap next to a launch)
        const char *e = getenv("IMGPROC_GAUSS_MFMA");
        if (!generic_only && !(e && e[0] == '0') && gaussian_mfma_ok(src, sstride, w, h, cn, taps) &&
            (unsigned long long)dstride * (unsigned long long)h < (1ull << 32))
            return launch_gaussian_mfma(src, sstride, dst, dstride, w, h, cn, taps, stream);
    }
    if (!generic_only && gaussian_strip_ok(src, sstride, w, h, cn, taps))
        return launch_gaussian_strip(src, sstride, dst, dstride, w, h, cn, taps, stream);
    dim3 grid((w * cn + G_TWB - 1) / G_TWB, (h + G_TH - 1) / G_TH);
    if (cn == 3)
        hipLaunchKernelGGL(k_gaussian<3>, grid, dim3(256), 0, stream, src, sstride, dst, dstride, w, h, taps);
    else
        hipLaunchKernelGGL(k_gaussian<1>, grid, dim3(256), 0, stream, src, sstride, dst, dstride, w, h, taps);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// inRange(hsv, lo, hi) for up to 8 ranges at once (objectPolygonPath.cpp:186, pose_estimate.cpp:205):
// the HSV pixel is read once, one mask plane is written per range.  Per channel a 256-entry byte table in
// LDS holds, for every value, the set of ranges whose [lo, hi] contains it (bit r = range r); a pixel's
// membership in ALL ranges is T0[c0] & T1[c1] & T2[c2], and plane r's four bytes are ((m4 >> r) & 0x01010101) * 255.
// PX pixels per lane: 16 (48-byte loads, 16-byte stores) when rows are 16-byte aligned, else 4.
// ---------------------------------------------------------------------------------------------
template <int PX>
__global__ __launch_bounds__(256) void k_in_range(const u8 *__restrict__ src, size_t sstride, u8 *__restrict__ dst,
                                                  size_t dstride, size_t plane_stride, int w, int h, Bounds b,
                                                  int vec_in, int vec_out)
{
    __shared__ u8 s_t[3][256];
    {
        const int v = threadIdx.y * 64 + threadIdx.x;  // 256 threads: one table column each
#pragma unroll
        for (int c = 0; c < 3; c++) {
            unsigned m = 0;
            for (int r = 0; r < b.n; r++) m |= (v >= b.lo[r][c] && v <= b.hi[r][c]) ? (1u << r) : 0u;
            s_t[c][v] = (u8)m;
        }
    }
    __syncthreads();
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * PX;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (y >= h || x0 >= w) return;
    const u8 *s = src + (size_t)y * sstride + (size_t)x0 * 3;
    const int n = min(PX, w - x0);
    constexpr int NW = PX * 3 / 4;
    uint32_t raw[NW];
    if (vec_in && n == PX) {
        if (PX == 16) {
            const uint4 *p = reinterpret_cast<const uint4 *>(s);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const uint4 q = p[i];
                raw[4 * i] = q.x; raw[4 * i + 1] = q.y; raw[4 * i + 2] = q.z; raw[4 * i + 3] = q.w;
            }
        } else {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(s);
#pragma unroll
            for (int i = 0; i < NW; i++) raw[i] = p[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NW; i++) {
            uint32_t v = 0;
            for (int j = 0; j < 4; j++)
                if ((4 * i + j) < 3 * n) v |= (uint32_t)s[4 * i + j] << (8 * j);
            raw[i] = v;
        }
    }
    uint32_t m4[PX / 4];  // byte k = range set of pixel 4*g + k
#pragma unroll
    for (int g = 0; g < PX / 4; g++) {
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int byte0 = 12 * g + 3 * k;
            uint32_t m = 0xFFu;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int bi = byte0 + c;
                m &= s_t[c][(raw[bi >> 2] >> (8 * (bi & 3))) & 255u];
            }
            acc |= m << (8 * k);
        }
        m4[g] = acc;
    }
    for (int r = 0; r < b.n; r++) {
        uint32_t o[PX / 4];
#pragma unroll
        for (int g = 0; g < PX / 4; g++) {
            const uint32_t t = (m4[g] >> r) & 0x01010101u;
            o[g] = (t << 8) - t;  // * 255 without a multiply; the top byte wraps correctly
        }
        u8 *d = dst + (size_t)r * plane_stride + (size_t)y * dstride + x0;
        if (vec_out && n == PX) {
            if (PX == 16)
                *reinterpret_cast<uint4 *>(d) = make_uint4(o[0], o[1], o[2], o[3]);
            else
                *reinterpret_cast<uint32_t *>(d) = o[0];
        } else {
            for (int i = 0; i < n; i++) d[i] = (u8)(o[i >> 2] >> (8 * (i & 3)));
        }
    }
}

// The same test with the result as bit planes (what the bit-plane morphology wants, morphbits.hip): plane r's word j =
// bit i set iff pixel 32 j + i lies in range r.  1/8 B per pixel and range instead of 1 B.
// A wave takes a run of 256-pixel chunks; lane = 4 pixels = 3 dwords, so a chunk is one contiguous 768-byte load.
// The tables hold DWORDS here, bit 4 r = "value inside range r": the AND of a pixel's three entries is its membership
// spread over the eight nibbles, four pixels shifted by 0..3 fill the nibbles, and an 8x8 nibble transpose across the
// 8 lanes of a 32-pixel word leaves word j of plane r in lane 8 j + r.  ~45 vector instructions per 256 pixels.
__device__ __forceinline__ uint32_t byte_x4(uint32_t w, uint32_t two, int k)  // 4 * byte k of w, one SDWA shift
{
    uint32_t r;
    switch (k) {
    case 0: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(two), "v"(w)); break;
    case 1: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(two), "v"(w)); break;
    case 2: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(two), "v"(w)); break;
    default: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(two), "v"(w)); break;
    }
    return r;
}

// FAST: rows and base are dword-aligned and w % 4 == 0, so a lane's 12 bytes are wholly inside or wholly outside the row.
template <bool FAST>
__global__ __launch_bounds__(256) void k_in_range_bits(const u8 *__restrict__ src, size_t sstride, uint32_t *__restrict__ dst,
                                                       int words, int w, int h, Bounds b, int chunks_per_row,
                                                       int chunks_per_wave, int total_chunks)
{
    __shared__ uint32_t s_t[3][256];
    {
        const int v = threadIdx.x;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            uint32_t m = 0;
#pragma unroll
            for (int r = 0; r < 8; r++)  // constant indices: the bounds stay in scalar registers
                m |= (r < b.n && v >= b.lo[r][c] && v <= b.hi[r][c]) ? (1u << (4 * r)) : 0u;
            s_t[c][v] = m;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int first = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * chunks_per_wave);
    if (first >= total_chunks) return;  // whole waves leave; no barrier below
    const int n_ch = min(chunks_per_wave, total_chunks - first);
    int y = first / chunks_per_row, cx = first - y * chunks_per_row;
    // 8x8 nibble transpose across the lanes of a word: butterflies with lane ^ 4, ^ 2, ^ 1 (partner by ds_swizzle)
    const uint32_t sel4 = (lane & 4) ? 0x03020706u : 0x05040100u;  // v_perm_b32(partner, own, sel)
    const uint32_t sel2 = (lane & 2) ? 0x03070105u : 0x06020400u;
    const uint32_t m1 = (lane & 1) ? 0xF0F0F0F0u : 0x0F0F0F0Fu;
    const uint32_t rot1 = (lane & 1) ? 4u : 28u;
    const uint32_t two = 2u;
    const char *tab = reinterpret_cast<const char *>(&s_t[0][0]);
    const int r = lane & 7;
    uint32_t *const dplane = dst + (size_t)r * h * words;  // this lane's plane (h * words < 2^31: checked at launch)
    auto fetch = [&](int yy, int cc, uint32_t (&d)[3]) {
        const int x0 = cc * 256 + lane * 4;
        const u8 *s = src + (size_t)yy * sstride + (uint32_t)(x0 * 3);
        if (FAST) {
            // no branch around the load (a skipped load would make the compiler wait for ALL loads at the join, the one just
            // issued for the next chunk included): lanes beyond the row re-read its last 4 pixels, their bits are dropped
            const uint32_t *p = reinterpret_cast<const uint32_t *>(src + (size_t)yy * sstride + (uint32_t)(min(x0, w - 4) * 3));
            d[0] = p[0]; d[1] = p[1]; d[2] = p[2];
        } else {
            const int nb = 3 * (w - x0);  // bytes of this lane's 12 that lie inside the row (may be <= 0)
#pragma unroll
            for (int i = 0; i < 3; i++) {
                uint32_t v = 0;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (4 * i + q < nb) v |= (uint32_t)s[4 * i + q] << (8 * q);
                d[i] = v;
            }
        }
    };
    auto advance = [&](int &yy, int &cc) {
        if (++cc == chunks_per_row) cc = 0, yy++;
    };
    auto process = [&](const uint32_t (&d)[3], int yy, int cc) {
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t m = 0xFFFFFFFFu;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int bi = 3 * k + c;
                m &= *reinterpret_cast<const uint32_t *>(tab + 1024 * c + byte_x4(d[bi >> 2], two, bi & 3));
            }
            acc |= m << k;
        }
        uint32_t q = (uint32_t)__builtin_amdgcn_ds_swizzle((int)acc, 0x101F);  // lane ^ 4
        acc = __builtin_amdgcn_perm(q, acc, sel4);
        q = (uint32_t)__builtin_amdgcn_ds_swizzle((int)acc, 0x081F);  // lane ^ 2
        acc = __builtin_amdgcn_perm(q, acc, sel2);
        q = (uint32_t)__builtin_amdgcn_ds_swizzle((int)acc, 0x041F);  // lane ^ 1
        acc = (acc & m1) | (__builtin_amdgcn_alignbit(q, q, rot1) & ~m1);
        // lane 8 j + r: word j of this chunk, plane r.  Pixels beyond w matched the tables on zero bytes: cleared.
        const int j = cc * 8 + (lane >> 3);
        const int n = w - j * 32;
        const uint32_t keep = n >= 32 ? 0xFFFFFFFFu : ((1u << max(n, 0)) - 1u);
        if (r < b.n && j < words) dplane[(uint32_t)(yy * words + j)] = acc & keep;
    };
    // two chunks in flight: the next one's load is issued before this one's table lookups.  The loads run up to two
    // chunks beyond the wave's run (the next wave's chunks; beyond the image the last row again) rather than branch.
    uint32_t da[3], db[3];
    fetch(y, cx, da);
    for (int i = 0; i < n_ch; i += 2) {
        int y1 = y, c1 = cx;
        advance(y1, c1);
        fetch(min(y1, h - 1), c1, db);
        process(da, y, cx);
        int y2 = y1, c2 = c1;
        advance(y2, c2);
        fetch(min(y2, h - 1), c2, da);
        if (i + 1 < n_ch) process(db, y1, c1);
        y = y2; cx = c2;
    }
}

hipError_t launch_in_range_bits(const u8 *src, size_t sstride, uint32_t *dst, int w, int h, const Bounds &b, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return hipSuccess;
    const int words = (w + 31) / 32;
    const bool fast = (uintptr_t)src % 4 == 0 && sstride % 4 == 0 && w % 4 == 0;
    const int cpr = (w + 255) / 256;
    const long long total = (long long)cpr * h;
    if (total > 0x7FFFFFFFLL / 4 || (long long)words * h > 0x7FFFFFFFLL || (long long)w * 3 > 0x7FFFFFFFLL) return hipErrorInvalidValue;
    // runs of chunks per wave: long enough to pay for the 3 KB of tables a workgroup builds, short enough for >= ~2000 workgroups
    long long cpw = total / (4 * 2048);
    cpw = cpw < 1 ? 1 : cpw > 16 ? 16 : cpw;
    const long long waves = (total + cpw - 1) / cpw;
    dim3 block(256), grid((unsigned)((waves + 3) / 4));
    if (fast)
        hipLaunchKernelGGL(k_in_range_bits<true>, grid, block, 0, stream, src, sstride, dst, words, w, h, b, cpr, (int)cpw, (int)total);
    else
        hipLaunchKernelGGL(k_in_range_bits<false>, grid, block, 0, stream, src, sstride, dst, words, w, h, b, cpr, (int)cpw, (int)total);
    return hipGetLastError();
}

hipError_t launch_in_range(const u8 *src, size_t sstride, u8 *dst, size_t dstride, size_t plane_stride, int w,
                           int h, const Bounds &b, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return hipSuccess;
    const bool a16 = ((uintptr_t)src % 16 == 0) && (sstride % 16 == 0) && ((uintptr_t)dst % 16 == 0) && (dstride % 16 == 0) &&
                     (plane_stride % 16 == 0) && w >= 256;
    int vec_in = ((uintptr_t)src % 4 == 0) && (sstride % 4 == 0);
    int vec_out = ((uintptr_t)dst % 4 == 0) && (dstride % 4 == 0) && (plane_stride % 4 == 0);
    dim3 block(64, 4);
    if (a16) {
        dim3 grid((w + 1023) / 1024, (h + 3) / 4);
        hipLaunchKernelGGL(k_in_range<16>, grid, block, 0, stream, src, sstride, dst, dstride, plane_stride, w, h, b, 1, 1);
    } else {
        dim3 grid((w + 255) / 256, (h + 3) / 4);
        hipLaunchKernelGGL(k_in_range<4>, grid, block, 0, stream, src, sstride, dst, dstride, plane_stride, w, h, b, vec_in,
                           vec_out);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// erode / dilate with an arbitrary structuring element (objectPolygonPath.cpp:416-432,
// pose_estimate.cpp:407-414).  Out-of-image samples are neutral (255 for erode, 0 for dilate).
// RECT elements run separably inside the tile (row min/max, then column min/max).
// ---------------------------------------------------------------------------------------------
constexpr int M_TW = 64, M_TH = 16, M_KMAX = 31;

template <int DILATE>
__global__ __launch_bounds__(256) void k_morph(const u8 *__restrict__ src, size_t sstride, u8 *__restrict__ dst,
                                               size_t dstride, int w, int h, Element e)
{
    __shared__ u8 s_t[(M_TH + M_KMAX - 1) * (M_TW + M_KMAX - 1)];
    __shared__ u8 s_r[(M_TH + M_KMAX - 1) * M_TW];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int LW = M_TW + e.kw - 1, LH = M_TH + e.kh - 1;
    const int x0 = blockIdx.x * M_TW - e.ax, y0 = blockIdx.y * M_TH - e.ay;
    const u8 neutral = DILATE ? 0 : 255;
    for (int idx = tid; idx < LH * LW; idx += 256) {
        int ty = idx / LW, tx = idx - ty * LW;
        int gx = x0 + tx, gy = y0 + ty;
        s_t[idx] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? src[(size_t)gy * sstride + gx] : neutral;
    }
    __syncthreads();
    const int ox = threadIdx.x;
    if (e.is_rect) {
        for (int idx = tid; idx < LH * M_TW; idx += 256) {
            int ty = idx / M_TW, tx = idx - ty * M_TW;
            const u8 *p = s_t + ty * LW + tx;
            int acc = neutral;
            for (int i = 0; i < e.kw; i++) acc = DILATE ? max(acc, (int)p[i]) : min(acc, (int)p[i]);
            s_r[idx] = (u8)acc;
        }
        __syncthreads();
        for (int q = 0; q < 4; q++) {
            int oy = threadIdx.y * 4 + q;
            int gx = blockIdx.x * M_TW + ox, gy = blockIdx.y * M_TH + oy;
            if (gx < w && gy < h) {
                int acc = neutral;
                for (int j = 0; j < e.kh; j++) {
                    int v = s_r[(oy + j) * M_TW + ox];
                    acc = DILATE ? max(acc, v) : min(acc, v);
                }
                dst[(size_t)gy * dstride + gx] = (u8)acc;
            }
        }
    } else {
        for (int q = 0; q < 4; q++) {
            int oy = threadIdx.y * 4 + q;
            int gx = blockIdx.x * M_TW + ox, gy = blockIdx.y * M_TH + oy;
            if (gx < w && gy < h) {
                int acc = neutral;
                for (int j = 0; j < e.kh; j++) {
                    uint32_t bits = e.rows[j];
                    const u8 *p = s_t + (oy + j) * LW + ox;
                    while (bits) {
                        int i = __ffs(bits) - 1;
                        bits &= bits - 1;
                        int v = p[i];
                        acc = DILATE ? max(acc, v) : min(acc, v);
                    }
                }
                dst[(size_t)gy * dstride + gx] = (u8)acc;
            }
        }
    }
}

hipError_t launch_morph(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, int kind,
                        const Element &e, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return hipSuccess;
    if (e.kw < 1 || e.kh < 1 || e.kw > M_KMAX || e.kh > M_KMAX || e.ax < 0 || e.ay < 0 || e.ax >= e.kw || e.ay >= e.kh)
        return hipErrorInvalidValue;
    dim3 block(64, 4), grid((w + M_TW - 1) / M_TW, (h + M_TH - 1) / M_TH);
    if (kind)
        hipLaunchKernelGGL(k_morph<1>, grid, block, 0, stream, src, sstride, dst, dstride, w, h, e);
    else
        hipLaunchKernelGGL(k_morph<0>, grid, block, 0, stream, src, sstride, dst, dstride, w, h, e);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Raster moments of a mask == cv::moments(mask,true) (objectPolygonPath.cpp:449-450) plus the
// second-order sums.  Per-lane u64 partials -> wave shuffle -> LDS -> one atomicAdd per block and
// moment.  Integer sums: the result does not depend on the order of the adds.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_moments(const u8 *__restrict__ mask, size_t stride, int w, int h,
                                                 unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long s_p[4][6];
    unsigned long long m[6] = {0, 0, 0, 0, 0, 0};
    for (int y = blockIdx.x; y < h; y += gridDim.x) {
        const u8 *row = mask + (size_t)y * stride;
        unsigned long long c = 0, sx = 0, sxx = 0;
        for (int x = threadIdx.x; x < w; x += 256)
            if (row[x]) {
                c += 1;
                sx += (unsigned)x;
                sxx += (unsigned long long)x * (unsigned)x;
            }
        m[0] += c;
        m[1] += sx;
        m[2] += c * (unsigned)y;
        m[3] += sxx;
        m[4] += sx * (unsigned)y;
        m[5] += c * (unsigned long long)y * (unsigned)y;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        unsigned long long v = wave_sum_u64(m[i]);
        if (lane == 0) s_p[wv][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned long long v = s_p[0][threadIdx.x] + s_p[1][threadIdx.x] + s_p[2][threadIdx.x] + s_p[3][threadIdx.x];
        if (v) atomicAdd(out + threadIdx.x, v);
    }
}

hipError_t launch_moments(const u8 *mask, size_t stride, int w, int h, unsigned long long *d_m, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return hipSuccess;
    int grid = h < 2048 ? h : 2048;
    hipLaunchKernelGGL(k_moments, dim3(grid), dim3(256), 0, stream, mask, stride, w, h, d_m);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// bayer_rggb8 -> BGR8 (cv::COLOR_BayerBG2BGR, the cv_bridge mapping at pose_estimate.cpp:180):
// bilinear on the interior; border pixels take the value of their clamped interior neighbour.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_demosaic_rggb(const u8 *__restrict__ src, size_t sstride,
                                                       u8 *__restrict__ dst, size_t dstride, int w, int h)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    u8 *d = dst + (size_t)y * dstride + (size_t)x * 3;
    if (w <= 2 || h <= 2) {
        d[0] = d[1] = d[2] = 0;
        return;
    }
    const int cx = min(max(x, 1), w - 2), cy = min(max(y, 1), h - 2);
    const u8 *p = src + (size_t)cy * sstride + cx;
    const int s = (int)sstride;
    int c = p[0];
    int n = p[-s], so = p[s], we = p[-1], ea = p[1];
    int B, G, R;
    if (((cy ^ cx) & 1) == 0) {
        int cross = (n + so + we + ea + 2) >> 2;
        int diag = (p[-s - 1] + p[-s + 1] + p[s - 1] + p[s + 1] + 2) >> 2;
        G = cross;
        if (cy & 1) { B = c; R = diag; } else { R = c; B = diag; }
    } else {
        int hor = (we + ea + 1) >> 1, ver = (n + so + 1) >> 1;
        G = c;
        if (cy & 1) { B = hor; R = ver; } else { R = hor; B = ver; }
    }
    d[0] = (u8)B;
    d[1] = (u8)G;
    d[2] = (u8)R;
}

hipError_t launch_demosaic_rggb(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h,
                                hipStream_t stream)
{
    if (w <= 0 || h <= 0) return hipSuccess;
    dim3 block(64, 4), grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_demosaic_rggb, grid, block, 0, stream, src, sstride, dst, dstride, w, h);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// cvtColor(GRAY2BGR) + bitwise_and(frame, maskFrame, procFrame) (objectPolygonPath.cpp:189-190)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mask_and(const u8 *__restrict__ bgr, size_t sstride,
                                                  const u8 *__restrict__ mask, size_t mstride, u8 *__restrict__ dst,
                                                  size_t dstride, int w, int h)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const u8 m = mask[(size_t)y * mstride + x];
    const u8 *s = bgr + (size_t)y * sstride + (size_t)x * 3;
    u8 *d = dst + (size_t)y * dstride + (size_t)x * 3;
    d[0] = s[0] & m;
    d[1] = s[1] & m;
    d[2] = s[2] & m;
}

hipError_t launch_mask_and(const u8 *bgr, size_t sstride, const u8 *mask, size_t mstride, u8 *dst, size_t dstride,
                           int w, int h, hipStream_t stream)
{
    if (w <= 0 || h <= 0) return hipSuccess;
    dim3 block(64, 4), grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_mask_and, grid, block, 0, stream, bgr, sstride, mask, mstride, dst, dstride, w, h);
    return hipGetLastError();
}

}  // namespace ipk
