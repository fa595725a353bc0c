// [cvtColor BGR2HSV] + GaussianBlur(k = 7..15) + inRange for 1..8 ranges in ONE kernel, bit planes out: the colour / blur /
// threshold stages of pipelines whose post-HSV Gaussian is long (BASELINE configs[4]: 7680x4320, 15x15, 8 ranges) and whose
// inRange planes go straight into the bit-plane morphology (morphbits.hip).  Replaces, for those pipelines, the sequence
// k_bgr2hsv -> k_gaussian_mfma -> k_in_range_bits (objectPolygonPath.cpp:183,184,186 with a 15x15 kernel): 3 B/px in,
// nr/8 B/px out instead of 3+3, 3+3 and 3+nr/8 through HBM.
//
// The Gaussian is k_gaussian_mfma's (gauss_mfma.hip): a separable filter as two banded i8 contractions on the matrix cores
// (v_mfma_i32_16x16x32_i8), the 8.8 intermediate split into byte planes in a per-wave LDS ring.  What differs:
//   * a strip is 96 bytes = 32 PIXELS of the interleaved image (6 tiles of 16 columns), so a strip row is exactly one
//     32-pixel word of every bit plane; its input is 144 bytes = 48 pixels, loaded as 12-byte (4-pixel) chunks: one
//     global_load_dwordx3 per lane, 16 rows x 12 chunks = three loads per lane and block, all lanes busy;
//   * the chunks are raw BGR / RGB pixels; each lane converts its 12 pixels (ipf::hsv_px, the chain kernel's
//     form) before the block goes through the LDS tile into MFMA operand order -- the HSV image never exists in HBM;
//   * the blurred bytes never leave the CU either: the output tile is read back pixel by pixel (8 pixels per lane),
//     each pixel's membership in all ranges is T0[h] & T1[s] & T2[v] on dword tables (bit 4 r = inside range r, as in
//     k_in_range_bits), eight pixels' nibbles pack into one byte per range, and a 4x4 byte transpose across the four
//     lanes of a row (DPP quad_perm + v_perm) leaves lane q with the row's 32-pixel word of ranges 2q and 2q + 1.
//   * frames of a batch are one launch (item = frame, segment, strip).
// Arithmetic: SURVEY.md Appendix A.1 / A.3 / A.4, bit-exact with the stage kernels and the oracle
// (tests/test_gpu_stages.py, tests/test_gpu_chain.py).
#include "hsv_px.hpp"
#include "kernels.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace ipk {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int GF_NT = 6;             // tiles of 16 byte columns per strip
constexpr int GF_WX = 16 * GF_NT;    // 96 output bytes = 32 pixels
constexpr int GF_OFF = 24;           // input bytes left of the strip (8 pixels >= the 7 a 15-tap kernel reaches)
// a strip row reads GF_WX + 48 = 144 input bytes = 48 pixels = 12 pixel chunks = 18 operand chunks of 8 bytes
constexpr int GF_PITCH = 152;        // LDS row pitch of the stage / output tile in bytes (19 x 8: operand chunks stay 8-byte aligned)

struct GfArgs {
    const u8 *src;
    size_t frame_stride, sstride;
    uint32_t *bits;  // [n][nr][h][words]
    const int *hsv_tables;
    int words, w, h, n, k;
    int taps[15];
    int nstrips, seg_blocks, nsegs, nblocks, nitems;
    Bounds b;
};

__device__ __forceinline__ int gf_reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
    return p;
}

__device__ __forceinline__ long gf_pack(uint32_t lo, uint32_t hi) { return (long)(((unsigned long long)hi << 32) | lo); }

__device__ __forceinline__ uint32_t quad_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); }  // lane ^ 2
__device__ __forceinline__ uint32_t quad_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }  // lane ^ 1

// 12 bytes = 4 pixels of BGR / RGB -> the same 12 bytes as H S V (interleaved), every byte XOR 0x80 (signed MFMA operand)
template <bool RGB>
__device__ __forceinline__ void hsv_chunk(uint32_t (&d)[3], const char *tab)
{
    using namespace ipf;
    uint32_t hs[4], vv[4];
    hsv_px<RGB>(byte_x4<0>(d[0]), byte_x4<1>(d[0]), byte_x4<2>(d[0]), tab, hs[0], vv[0]);
    hsv_px<RGB>(byte_x4<3>(d[0]), byte_x4<0>(d[1]), byte_x4<1>(d[1]), tab, hs[1], vv[1]);
    hsv_px<RGB>(byte_x4<2>(d[1]), byte_x4<3>(d[1]), byte_x4<0>(d[2]), tab, hs[2], vv[2]);
    hsv_px<RGB>(byte_x4<1>(d[2]), byte_x4<2>(d[2]), byte_x4<3>(d[2]), tab, hs[3], vv[3]);
    const uint32_t v01 = (vv[0] | (vv[1] << 16)) >> 2;  // V0 in byte 0, V1 in byte 2
    const uint32_t v23 = (vv[2] | (vv[3] << 16)) >> 2;
    // hs[k] = H in byte 0, S in byte 2.  v_perm_b32(a, b, sel): selector 0-3 = bytes of b, 4-7 = bytes of a.
    const uint32_t t0 = __builtin_amdgcn_perm(hs[0], v01, 0x0C000604u);   // H0 S0 V0 --
    d[0] = (t0 | (hs[1] << 24)) ^ 0x80808080u;                             // .. H1
    const uint32_t t1 = __builtin_amdgcn_perm(hs[1], v01, 0x0C0C0206u);   // S1 V1 -- --
    d[1] = (t1 | __builtin_amdgcn_perm(hs[2], hs[2], 0x02000C0Cu)) ^ 0x80808080u;  // .. .. H2 S2
    d[2] = __builtin_amdgcn_perm(hs[3], v23, 0x02060400u) ^ 0x80808080u;  // V2 H3 S3 V3
}

template <bool RGB>
__global__ __launch_bounds__(256) void k_hsv_gauss_inrange(const GfArgs a)
{
    __shared__ uint32_t s_mid[4][3][2][GF_NT][4][16];                 // per wave: ring of three blocks, lo / hi byte planes
    __shared__ __align__(16) unsigned char s_tile[4][16 * GF_PITCH];  // per wave: input block, later the block's output bytes
    __shared__ int s_taps[16];
    __shared__ int s_hsv[512];                                        // 4 * sdiv, 4 * hdiv180
    __shared__ uint32_t s_rng[3][256];                                // bit 4 r: value inside range r on that channel
    if (threadIdx.x < 16) s_taps[threadIdx.x] = threadIdx.x < 15 ? a.taps[threadIdx.x] : 0;
    s_hsv[threadIdx.x] = a.hsv_tables[threadIdx.x] * 4;
    s_hsv[threadIdx.x + 256] = a.hsv_tables[threadIdx.x + 256] * 4;
    {
        const int v = threadIdx.x;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            uint32_t m = 0;
#pragma unroll
            for (int r = 0; r < 8; r++) m |= (r < a.b.n && v >= a.b.lo[r][c] && v <= a.b.hi[r][c]) ? (1u << (4 * r)) : 0u;
            s_rng[c][v] = m;
        }
    }
    __syncthreads();  // the only workgroup barrier, before any wave leaves
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // XCD-aware order (as k_gaussian_mfma): neighbouring strips share input lines and meet in one L2
    const int per_xcd = (int)gridDim.x >> 3;  // the grid is a multiple of 8 workgroups
    const int lwg = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int item = __builtin_amdgcn_readfirstlane(lwg * 4 + wv);
    if (item >= a.nitems) return;  // whole waves leave; no workgroup barrier below
    const int strip = item % a.nstrips;
    const int rest = item / a.nstrips;
    const int seg = rest % a.nsegs, frame = rest / a.nsegs;
    const int i16 = lane & 15, c = lane >> 4;
    const int K = a.k, R = K >> 1;
    const int w = a.w, h = a.h;
    const u8 *fsrc = a.src + (size_t)frame * a.frame_stride;
    const char *hsv_tab = reinterpret_cast<const char *>(s_hsv);
    const char *rng_tab = reinterpret_cast<const char *>(&s_rng[0][0]);

    // ---- the two Toeplitz operands, built once per lane: 8 k-values (bytes) per dword pair ----
    uint32_t rowB[2][2] = {{0, 0}, {0, 0}}, colB[2] = {0, 0};
    for (int m = 0; m < 8; m++) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int d = 32 * b + 8 * c + m - i16 - GF_OFF + R * 3;  // (input byte) - (first input byte of output column i16)
            const int t = d / 3;
            const int tap = (d >= 0 && t * 3 == d && t < K) ? s_taps[t] : 0;
            rowB[b][m >> 2] |= (uint32_t)tap << (8 * (m & 3));
        }
        const int t = 8 * c + m - i16 - 8 + R;  // window row k = mid row 16 B - 8 + k; output row i16 needs rows +(t - R)
        colB[m >> 2] |= (uint32_t)((t >= 0 && t < K) ? s_taps[t] : 0) << (8 * (m & 3));
    }
    const long rowB0 = gf_pack(rowB[0][0], rowB[0][1]), rowB1 = gf_pack(rowB[1][0], rowB[1][1]);
    const long colBl = gf_pack(colB[0], colB[1]);

    // ---- loads: a block is 16 rows x 12 pixel chunks (4 pixels = 12 bytes); load j takes chunk index 64 j + lane ----
    const bool edge_strip = 32 * strip - 8 < 0 || 32 * strip + 40 > w;  // uniform: some input pixel lies outside the row
    int lrow[3], lq[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int q = 64 * j + lane;
        lrow[j] = q / 12;
        lq[j] = q - 12 * lrow[j];
    }
    unsigned char *tile = s_tile[wv];
    uint32_t(*mid)[2][GF_NT][4][16] = s_mid[wv];
    auto gray = [](int g) { return g ^ (g >> 1); };  // row group -> slot inside a tile: 0, 1, 3, 2 (bank halves, see gauss_mfma.hip)
    const int Bs = seg * a.seg_blocks, Be = min(Bs + a.seg_blocks, a.nblocks);

    // 4x4 byte transpose across the four lanes of a row (output stage)
    const uint32_t selA = (lane & 2) ? 0x03020706u : 0x05040100u;  // v_perm_b32(partner, own, sel)
    const uint32_t selB = (lane & 1) ? 0x03070105u : 0x06020400u;
    const int orow_l = lane >> 2, oq = lane & 3;
    const int n_last = w - 32 * strip;  // pixels of this strip's word that exist
    const uint32_t keep = n_last >= 32 ? 0xFFFFFFFFu : ((1u << max(n_last, 0)) - 1u);
    const int nr = a.b.n;
    uint32_t *const plane_e = a.bits + (((size_t)frame * nr + 2 * oq) * h) * a.words + strip;      // range 2 q
    uint32_t *const plane_o = a.bits + (((size_t)frame * nr + 2 * oq + 1) * h) * a.words + strip;  // range 2 q + 1

    auto run = [&](auto edge_c) {
    constexpr bool EDGE = decltype(edge_c)::value;
    auto load_block = [&](int blk, uint32_t (&ld)[3][3]) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int y = gf_reflect101(16 * blk + lrow[j], h);
            const u8 *row = fsrc + (uint32_t)y * (uint32_t)a.sstride;  // 32-bit in-frame offsets (checked at launch)
            const int px0 = 32 * strip - 8 + 4 * lq[j];
            if (!EDGE) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(row + 3 * px0);
                ld[j][0] = q[0];
                ld[j][1] = q[1];
                ld[j][2] = q[2];
            } else {
                // BORDER_REFLECT_101 on the pixel index (HSV is per pixel: converting the reflected pixel is the reflected HSV)
                uint32_t bb[12];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const u8 *s = row + 3 * gf_reflect101(px0 + p, w);
                    bb[3 * p] = s[0];
                    bb[3 * p + 1] = s[1];
                    bb[3 * p + 2] = s[2];
                }
#pragma unroll
                for (int i = 0; i < 3; i++) ld[j][i] = bb[4 * i] | (bb[4 * i + 1] << 8) | (bb[4 * i + 2] << 16) | (bb[4 * i + 3] << 24);
            }
        }
    };
    uint32_t bufA[3][3], bufB[3][3];
    auto step = [&](int blk, uint32_t (&buf)[3][3]) {
        // ---- this block: HSV, into the LDS tile in row order, back out in operand order ----
#pragma unroll
        for (int j = 0; j < 3; j++) {
            uint32_t d[3] = {buf[j][0], buf[j][1], buf[j][2]};
            hsv_chunk<RGB>(d, hsv_tab);
            uint32_t *t = reinterpret_cast<uint32_t *>(tile + lrow[j] * GF_PITCH + 12 * lq[j]);
            t[0] = d[0];
            t[1] = d[1];
            t[2] = d[2];
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t cur[GF_NT + 2][2];
#pragma unroll
        for (int u = 0; u < GF_NT + 2; u++) {  // lane (row i16, chunk c) takes the 8-byte chunks c + 2u
            const uint2 v = *reinterpret_cast<const uint2 *>(tile + i16 * GF_PITCH + 8 * (c + 2 * u));
            cur[u][0] = v.x;
            cur[u][1] = v.y;
        }
        __builtin_amdgcn_wave_barrier();
        load_block(min(blk + 2, Be), buf);  // two blocks in flight while this one is multiplied
        // ---- row pass of block blk into the ring ----
        const int slot = (blk + 3) % 3;
        v4i acc[GF_NT];
#pragma unroll
        for (int t = 0; t < GF_NT; t++) {
            const v4i c0 = {32768, 32768, 32768, 32768};  // 128 * sum(taps): undoes the -128 of the data bytes
            acc[t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(gf_pack(cur[t][0], cur[t][1]), rowB0, c0, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < GF_NT; t++)
            acc[t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(gf_pack(cur[t + 2][0], cur[t + 2][1]), rowB1, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < GF_NT; t++) {
            const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)acc[t][1], (uint32_t)acc[t][0], 0x05010400u);
            const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)acc[t][3], (uint32_t)acc[t][2], 0x05010400u);
            mid[slot][0][t][gray(c)][i16] = __builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u;
            mid[slot][1][t][gray(c)][i16] = __builtin_amdgcn_perm(p23, p01, 0x07060302u) ^ 0x80808080u;
        }
        __builtin_amdgcn_wave_barrier();
        if (blk < Bs + 1) return;  // the column pass of block ob needs blocks ob - 1, ob, ob + 1
        // ---- column pass of output block ob ----
        const int ob = blk - 1;
        const int wg = 2 * c + 2;  // this lane's first row group, counted from block ob - 1
        const int rslot = (ob - 1 + (wg >> 2) + 3) % 3, rg = wg & 3;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            v4i dlo[3], dhi[3];
#pragma unroll
            for (int tt = 0; tt < 3; tt++) {
                const int t = 3 * half + tt;
                const long alo = gf_pack(mid[rslot][0][t][gray(rg)][i16], mid[rslot][0][t][gray(rg + 1)][i16]);
                const long ahi = gf_pack(mid[rslot][1][t][gray(rg)][i16], mid[rslot][1][t][gray(rg + 1)][i16]);
                // sum(tap * mid) = S_lo + 256 S_hi + 128 * 256 * (1 + 256); + 2^15 for the rounding
                const v4i k0 = {8454144, 8454144, 8454144, 8454144}, z = {0, 0, 0, 0};
                dlo[tt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(alo, colBl, k0, 0, 0, 0);
                dhi[tt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(ahi, colBl, z, 0, 0, 0);
            }
#pragma unroll
            for (int tt = 0; tt < 3; tt++) {
                const int t = 3 * half + tt;
                uint32_t s[4];
#pragma unroll
                for (int r = 0; r < 4; r++) s[r] = (uint32_t)dlo[tt][r] + ((uint32_t)dhi[tt][r] << 8);
                // lane (row i16, c): s[r] is byte column 16 t + 4 c + r of the strip row; byte 2 of every sum is the result
                *reinterpret_cast<uint32_t *>(tile + i16 * GF_PITCH + 16 * t + 4 * c) =
                    __builtin_amdgcn_perm(s[1], s[0], 0x0c0c0602u) | __builtin_amdgcn_perm(s[3], s[2], 0x06020c0cu);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- inRange on the block's blurred pixels: lane (row l >> 2, q = l & 3) takes pixels 8 q .. 8 q + 7 of its row ----
        {
            using ipf::byte_x4;
            const uint2 *tp = reinterpret_cast<const uint2 *>(tile + orow_l * GF_PITCH + 24 * oq);
            const uint2 v0 = tp[0], v1 = tp[1], v2 = tp[2];
            const uint32_t d[6] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y};
            auto look = [&](int ch, uint32_t idx4) { return *reinterpret_cast<const uint32_t *>(rng_tab + 1024 * ch + idx4); };
            uint32_t A = 0, B = 0;
#pragma unroll
            for (int half = 0; half < 2; half++) {  // four pixels = three dwords
                const uint32_t w0 = d[3 * half], w1 = d[3 * half + 1], w2 = d[3 * half + 2];
                const uint32_t m0 = look(0, byte_x4<0>(w0)) & look(1, byte_x4<1>(w0)) & look(2, byte_x4<2>(w0));
                const uint32_t m1 = look(0, byte_x4<3>(w0)) & look(1, byte_x4<0>(w1)) & look(2, byte_x4<1>(w1));
                const uint32_t m2 = look(0, byte_x4<2>(w1)) & look(1, byte_x4<3>(w1)) & look(2, byte_x4<0>(w2));
                const uint32_t m3 = look(0, byte_x4<1>(w2)) & look(1, byte_x4<2>(w2)) & look(2, byte_x4<3>(w2));
                const uint32_t m = m0 | (m1 << 1) | (m2 << 2) | (m3 << 3);  // nibble r: pixels 0..3 of this half in range r
                if (half == 0) A = m; else B = m;
            }
            // byte i of E: range 2 i, bit j = pixel 8 q + j; byte i of O: range 2 i + 1
            uint32_t E = (A & 0x0F0F0F0Fu) | ((B & 0x0F0F0F0Fu) << 4);
            uint32_t O = ((A >> 4) & 0x0F0F0F0Fu) | (B & 0xF0F0F0F0u);
            // 4x4 byte transpose across the row's four lanes: lane q ends up with the 32-pixel words of ranges 2 q and 2 q + 1
            E = __builtin_amdgcn_perm(quad_xor2(E), E, selA);
            O = __builtin_amdgcn_perm(quad_xor2(O), O, selA);
            E = __builtin_amdgcn_perm(quad_xor1(E), E, selB);
            O = __builtin_amdgcn_perm(quad_xor1(O), O, selB);
            const int y = 16 * ob + orow_l;
            if (y < h) {
                const uint32_t off = (uint32_t)y * (uint32_t)a.words;  // h * words < 2^31 (checked at launch)
                if (2 * oq < nr) plane_e[off] = E & keep;
                if (2 * oq + 1 < nr) plane_o[off] = O & keep;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };  // step
    load_block(Bs - 1, bufA);
    load_block(Bs, bufB);
    int blk = Bs - 1;
    for (; blk + 1 <= Be; blk += 2) {  // whole pairs only: a conditional second step would hide its loads from the first's wait
        step(blk, bufA);
        step(blk + 1, bufB);
    }
    if (blk <= Be) step(blk, bufA);
    };  // run
    if (edge_strip)
        run(std::true_type{});
    else
        run(std::false_type{});
}

}  // namespace

bool gauss_fused_ok(const u8 *src, size_t frame_stride, size_t sstride, int w, int h, const Taps &taps, const Bounds &b)
{
    if (taps.k < 7 || taps.k > 15 || !(taps.k & 1)) return false;
    if ((uintptr_t)src % 4 || sstride % 4 || frame_stride % 4) return false;  // dword loads of 12-byte pixel chunks
    if (w < 1 || h < 1 || w > (1 << 24) || (unsigned long long)sstride * (unsigned long long)h >= (1ull << 32)) return false;
    if ((unsigned long long)h * (unsigned long long)((w + 31) / 32) >= (1ull << 31)) return false;
    if (b.n < 1 || b.n > 8) return false;
    int sum = 0;
    for (int i = 0; i < taps.k; i++) {
        if (taps.t[i] < 0 || taps.t[i] > 127) return false;  // signed 8-bit operands
        sum += taps.t[i];
    }
    return sum == 256;  // the -128 corrections assume it
}

hipError_t launch_gauss_fused(const u8 *src, size_t frame_stride, size_t sstride, uint32_t *bits, int w, int h, int n,
                              const Taps &taps, const Bounds &b, int bidx, const int *d_hsv_tables, hipStream_t stream,
                              bool overlapped)
{
    if (n <= 0 || w <= 0 || h <= 0) return hipSuccess;
    GfArgs a{};
    a.src = src; a.frame_stride = frame_stride; a.sstride = sstride; a.bits = bits; a.hsv_tables = d_hsv_tables;
    a.words = (w + 31) / 32; a.w = w; a.h = h; a.n = n; a.k = taps.k;
    for (int i = 0; i < taps.k; i++) a.taps[i] = taps.t[i];
    a.b = b;
    a.nstrips = a.words;
    a.nblocks = (h + 15) / 16;
    // every segment runs two extra blocks (the one above, the one below), yet short segments win: a block's steps are a
    // chain of dependent LDS / MFMA phases at the 3 waves per SIMD the LDS allows, and a finer grain evens out the rounds.
    // Measured at 4 frames of 7680x4320 (960 strip columns): 8 segments per column 0.433 ms, 12 0.397, 16 0.375, 23 0.360,
    // 34 0.362, 46 0.371, 68 0.416 -> about 24 k waves per launch, at least 4 blocks per segment.  With another batch's
    // launches in flight on the other stream (overlapped) the rounds even out by themselves and fewer, taller segments win:
    // whole step of that configuration at 9 / 13 / 18 / 26 / 34 segments 0.661 / 0.622 / 0.600 / 0.614 / 0.625 ms -> 16 k waves
    const long long cols = (long long)a.nstrips * n;
    const long long waves = overlapped ? 16384 : 24576;
    int nsegs = (int)std::max<long long>(1, std::min<long long>((waves + cols - 1) / cols, (a.nblocks + 3) / 4));
    if (const char *e = getenv("IMGPROC_GAUSS_SEGS")) nsegs = std::max(1, std::min(atoi(e), a.nblocks));
    a.seg_blocks = (a.nblocks + nsegs - 1) / nsegs;
    a.nsegs = (a.nblocks + a.seg_blocks - 1) / a.seg_blocks;
    a.nitems = a.nstrips * a.nsegs * n;
    const dim3 grid(((a.nitems + 3) / 4 + 7) / 8 * 8), block(256);
    if (bidx == 2)
        hipLaunchKernelGGL((k_hsv_gauss_inrange<true>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((k_hsv_gauss_inrange<false>), grid, block, 0, stream, a);
    return hipGetLastError();
}

}  // namespace ipk
