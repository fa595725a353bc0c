// GaussianBlur(src,dst,Size(k,k),0,0) on CV_8U for odd k <= 15 (objectPolygonPath.cpp:182,184 use 3 and 5, the
// BASELINE stress preset 15): strip kernel, column pass first.
//
// One wave owns a 512-byte column strip of the interleaved image (8 bytes per lane; the outer lanes are halo) and
// marches down a segment of rows.  The last K raw rows live in a register ring as 16-bit byte pairs, so the column
// pass is SWAR: one v_mad_u32_u24 per two bytes and tap (partial sums stay below 2^16 because the taps sum to 256).
// Its 8.8 result row goes through a 1 KB per-wave LDS row, the row pass reads it back as aligned b128 windows and
// forms t_i*(x[-i]+x[+i]) with SDWA word selects -- no byte extraction, no division, no barrier (LDS ops of one wave
// execute in order).  OpenCV's single rounding (sum + 2^15) >> 16 is folded in as +128 per
// 8.8 field of the column pass (256 * 128 = 2^15).  Arithmetic: SURVEY.md Appendix A.2; the generic tile kernel in
// kernels.hip remains for k > 15, unsymmetric taps and unaligned rows.
#include "kernels.h"

#include <algorithm>
#include <utility>

namespace ipk {

namespace {

__device__ __forceinline__ int gs_reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
    return p;
}

__device__ __forceinline__ uint32_t gs_mad(int tap, uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "s"(tap), "v"(x), "v"(acc));
    return r;
}

// (a.word[h0] + b.word[h1]) as a full dword
__device__ __forceinline__ uint32_t gs_addw(uint32_t a, int h0, uint32_t b, int h1)
{
    uint32_t r;
    if (h0 == 0 && h1 == 0)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b));
    else if (h0 == 0 && h1 == 1)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
    else if (h0 == 1 && h1 == 0)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b));
    else
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// tap * a.word[h]
__device__ __forceinline__ uint32_t gs_mulw(uint32_t tapv, uint32_t a, int h)
{
    uint32_t r;
    if (h == 0)
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(tapv), "v"(a));
    else
        asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(tapv), "v"(a));
    return r;
}

constexpr int GS_LANE_BYTES = 8;
constexpr int GS_STRIP = 64 * GS_LANE_BYTES;  // 512 bytes per wave row

constexpr int gs_floor_shr2(int v) { return v >= 0 ? v / 4 : -((-v + 3) / 4); }

template <int CN, int K>
struct GsGeom {
    static constexpr int R = K / 2;
    static constexpr int REACH = R * CN;                                  // bytes to either side
    static constexpr int HP = ((REACH + GS_LANE_BYTES - 1) / GS_LANE_BYTES) * GS_LANE_BYTES;  // halo bytes per side
    static constexpr int HL = HP / GS_LANE_BYTES;                          // halo lanes per side
    static constexpr int VALID = GS_STRIP - 2 * HP;                        // output bytes per wave row
    static constexpr int G0 = gs_floor_shr2(-REACH) - ((gs_floor_shr2(-REACH) & 1) ? 1 : 0);  // first 4-byte group (even)
    static constexpr int G1 = (GS_LANE_BYTES - 1 + REACH) / 4;             // last group
    static constexpr int NQ = (2 * (G1 - G0 + 1) + 3) / 4;                 // b128 reads per lane and row
};

// LDS row layout: byte position p of the strip (0..511) is the 16-bit word (p>>1)&1 of dword 2*(p>>2) + (p&1)
// (each 4-byte group is stored as the two SWAR accumulators (b0,b2), (b1,b3)).

template <class F, int... Is>
__device__ __forceinline__ void gs_unrolled(F &f, std::integer_sequence<int, Is...>)
{
    (void)(f(std::integral_constant<int, Is>{}) && ...);
}

template <int CN, int K>
__global__ __launch_bounds__(256) void k_gaussian_strip(const u8 *__restrict__ src, size_t sstride, u8 *__restrict__ dst,
                                                        size_t dstride, int w, int h, Taps taps, int seg_rows,
                                                        int nstrips, int nitems, int dst_vec)
{
    using G = GsGeom<CN, K>;
    constexpr int R = G::R;
    __shared__ uint4 s_mid[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wv;
    if (item >= nitems) return;  // whole waves leave; the kernel has no barrier
    const int strip = item % nstrips, seg = item / nstrips;
    const int wb = w * CN;
    const int gb = strip * G::VALID - G::HP + lane * GS_LANE_BYTES;  // first byte column of this lane
    const int ys = seg * seg_rows;
    const int ye = min(ys + seg_rows, h);
    const int nin = ye - ys + 2 * R;
    const bool inside = gb >= 0 && gb + GS_LANE_BYTES <= wb;
    uint4 *lrow = s_mid[wv];

    // lanes that straddle the image border read byte-wise through reflect-101 offsets computed once
    int eoff[GS_LANE_BYTES];
#pragma unroll
    for (int i = 0; i < GS_LANE_BYTES; i++) eoff[i] = 0;
    if (!inside) {
#pragma unroll
        for (int i = 0; i < GS_LANE_BYTES; i++) {
            const int pp = gb + i + CN * 65536;  // keep the division on non-negative values
            const int pq = pp / CN;
            eoff[i] = gs_reflect101(pq - 65536, w) * CN + (pp - pq * CN);
        }
    }
    auto load_row = [&](int n) -> uint2 {
        const int y = gs_reflect101(ys - R + n, h);
        const u8 *row = src + (size_t)y * sstride;
        uint2 v;
        if (inside) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(row + gb);
            v.x = p[0];
            v.y = p[1];
        } else {
            v.x = row[eoff[0]] | (row[eoff[1]] << 8) | (row[eoff[2]] << 16) | ((uint32_t)row[eoff[3]] << 24);
            v.y = row[eoff[4]] | (row[eoff[5]] << 8) | (row[eoff[6]] << 16) | ((uint32_t)row[eoff[7]] << 24);
        }
        return v;
    };

    uint32_t ring[K][4];
    const uint32_t tapc = (uint32_t)taps.t[R];
    const bool writer = lane >= G::HL && lane < 64 - G::HL && gb < wb;
    const bool full = dst_vec && gb + GS_LANE_BYTES <= wb;
    uint2 nxt = load_row(0);
    int nb = 0;
    // one ring slot per call; U is the slot, so every ring index below is a compile-time constant
    auto step = [&](auto uc) -> bool {
        constexpr int u = decltype(uc)::value;
        const int n = nb + u;
        if (n >= nin) return false;
        const uint2 cur = nxt;
        if (n + 1 < nin) nxt = load_row(n + 1);
        ring[u][0] = cur.x & 0x00FF00FFu;
        ring[u][1] = (cur.x >> 8) & 0x00FF00FFu;
        ring[u][2] = cur.y & 0x00FF00FFu;
        ring[u][3] = (cur.y >> 8) & 0x00FF00FFu;
        if (n < K - 1) return true;
        // column pass: rows n-2R..n sit in slots (u+1+j) % K, j = 0..K-1
        uint32_t mid[4];
#pragma unroll
        for (int f = 0; f < 4; f++) {
            uint32_t acc = gs_mad(taps.t[R], ring[(u + 1 + R) % K][f], 0x00800080u);
#pragma unroll
            for (int j = 1; j <= R; j++) {  // a pre-added pair would be 25 bits wide: one u24 mad per row instead
                acc = gs_mad(taps.t[R - j], ring[(u + 1 + R - j) % K][f], acc);
                acc = gs_mad(taps.t[R - j], ring[(u + 1 + R + j) % K][f], acc);
            }
            mid[f] = acc;
        }
        lrow[lane] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
        __builtin_amdgcn_wave_barrier();
        uint32_t win[G::NQ * 4];
#pragma unroll
        for (int i = 0; i < G::NQ; i++) {
            const uint4 q = lrow[(lane + G::G0 / 2 + i) & 63];
            win[4 * i + 0] = q.x;
            win[4 * i + 1] = q.y;
            win[4 * i + 2] = q.z;
            win[4 * i + 3] = q.w;
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t acc[GS_LANE_BYTES];
#pragma unroll
        for (int q = 0; q < GS_LANE_BYTES; q++) {
            const int dc = 2 * ((q >> 2) - G::G0) + (q & 1), hc = (q >> 1) & 1;
            // all pair sums first, then the multiply-add chain: an SDWA result consumed by the very next
            // instruction costs a wait state
            uint32_t sp[R > 0 ? R : 1];
#pragma unroll
            for (int i = 1; i <= R; i++) {
                const int rm = q - i * CN, rp = q + i * CN;
                const int gm = gs_floor_shr2(rm), gp = gs_floor_shr2(rp);
                const int qm = rm - 4 * gm, qp = rp - 4 * gp;
                sp[i - 1] = gs_addw(win[2 * (gm - G::G0) + (qm & 1)], qm >> 1, win[2 * (gp - G::G0) + (qp & 1)], qp >> 1);
            }
            uint32_t a = gs_mulw(tapc, win[dc], hc);
#pragma unroll
            for (int i = 1; i <= R; i++) a = gs_mad(taps.t[R - i], sp[i - 1], a);
            acc[q] = a;
        }
        // byte 2 of every accumulator is the result
        const uint32_t o0 = __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u) | __builtin_amdgcn_perm(acc[3], acc[2], 0x06020c0cu);
        const uint32_t o1 = __builtin_amdgcn_perm(acc[5], acc[4], 0x0c0c0602u) | __builtin_amdgcn_perm(acc[7], acc[6], 0x06020c0cu);
        if (writer) {
            u8 *orow = dst + (size_t)(ys + n - (K - 1)) * dstride + gb;
            if (full) {
                uint32_t *p = reinterpret_cast<uint32_t *>(orow);
                p[0] = o0;
                p[1] = o1;
            } else {
                for (int i = 0; i < GS_LANE_BYTES && gb + i < wb; i++) orow[i] = (u8)((i < 4 ? o0 : o1) >> (8 * (i & 3)));
            }
        }
        return true;
    };
    for (; nb < nin; nb += K) gs_unrolled(step, std::make_integer_sequence<int, K>{});
}

template <int CN, int K>
hipError_t launch_strip(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, const Taps &taps,
                        hipStream_t stream)
{
    using G = GsGeom<CN, K>;
    const int wb = w * CN;
    const int nstrips = (wb + G::VALID - 1) / G::VALID;
    // enough waves to fill 256 CUs several times over, segments of at least 32 output rows
    int nsegs = (8192 + nstrips - 1) / nstrips;
    nsegs = std::max(1, std::min(nsegs, (h + 31) / 32));
    const int seg_rows = (h + nsegs - 1) / nsegs;
    nsegs = (h + seg_rows - 1) / seg_rows;
    const int nitems = nstrips * nsegs;
    const int dst_vec = ((uintptr_t)dst % 4 == 0) && (dstride % 4 == 0);
    hipLaunchKernelGGL((k_gaussian_strip<CN, K>), dim3((nitems + 3) / 4), dim3(256), 0, stream, src, sstride, dst, dstride,
                       w, h, taps, seg_rows, nstrips, nitems, dst_vec);
    return hipGetLastError();
}

template <int CN>
hipError_t dispatch_k(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, const Taps &taps,
                      hipStream_t stream)
{
    switch (taps.k) {
    case 3: return launch_strip<CN, 3>(src, sstride, dst, dstride, w, h, taps, stream);
    case 5: return launch_strip<CN, 5>(src, sstride, dst, dstride, w, h, taps, stream);
    case 7: return launch_strip<CN, 7>(src, sstride, dst, dstride, w, h, taps, stream);
    case 9: return launch_strip<CN, 9>(src, sstride, dst, dstride, w, h, taps, stream);
    case 11: return launch_strip<CN, 11>(src, sstride, dst, dstride, w, h, taps, stream);
    case 13: return launch_strip<CN, 13>(src, sstride, dst, dstride, w, h, taps, stream);
    case 15: return launch_strip<CN, 15>(src, sstride, dst, dstride, w, h, taps, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace

bool gaussian_strip_ok(const u8 *src, size_t sstride, int w, int h, int cn, const Taps &taps)
{
    if (taps.k < 3 || taps.k > 15 || !(taps.k & 1) || (cn != 1 && cn != 3)) return false;
    if ((uintptr_t)src % 4 || sstride % 4) return false;
    if (w < 1 || h < 1 || (long long)w * cn > (1 << 28)) return false;
    int sum = 0;
    for (int i = 0; i < taps.k; i++) {
        if (taps.t[i] < 0 || taps.t[i] != taps.t[taps.k - 1 - i]) return false;
        sum += taps.t[i];
    }
    return sum == 256;  // keeps every SWAR partial sum below 2^16
}

hipError_t launch_gaussian_strip(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, int cn,
                                 const Taps &taps, hipStream_t stream)
{
    return cn == 3 ? dispatch_k<3>(src, sstride, dst, dstride, w, h, taps, stream)
                   : dispatch_k<1>(src, sstride, dst, dstride, w, h, taps, stream);
}

}  // namespace ipk
