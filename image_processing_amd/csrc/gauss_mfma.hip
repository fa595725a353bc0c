// GaussianBlur(src,dst,Size(k,k),0,0) on CV_8U for the long kernels (odd k = 7..15; the BASELINE stress preset uses 15)
// on the matrix cores.  A separable filter with 15 taps IS a contraction -- a banded (Toeplitz) matrix times the image
// -- and at that length it pays: the strip kernel in gauss.hip spends 24 VALU instructions per output byte on it,
// this one ~0.1 and two MFMAs per 16x16 tile and pass.
//
// One wave owns a strip of 64 output bytes of the interleaved image and marches down in blocks of 16 rows.
//   row pass:    mid[16 rows][16 cols] = in[16 rows][64 bytes] x T_row[64][16]   two v_mfma_i32_16x16x32_i8 per tile;
//                the A operand of lane (row i, chunk c) is the 8 consecutive input bytes it loaded itself -- no
//                shuffles; T_row[k][j] = tap[(k - j - 24) / cn + r] on the pixel stride, 0 elsewhere.
//   mid (8.8, 16 bits) is split into its low and high bytes, four rows of one column per dword, into a per-wave LDS
//                ring of three blocks: exactly the A operand layout of the column pass.
//   column pass: out^T[16 cols][16 rows] = mid^T[16 cols][32 rows] x T_col[32][16], once for the low and once for the
//                high byte plane; out = (S_lo + 256 S_hi + const) >> 16.  Transposed so that a lane ends up with four
//                x-consecutive output bytes of one row.
// Input blocks are loaded coalesced (adjacent lanes = adjacent bytes, four rows per instruction) and turned into operand
// order through a per-wave LDS tile: loading straight into operand order (adjacent lanes = adjacent ROWS) made every
// lane its own memory request, 44 cycles per load instruction.  Workgroups are mapped XCD-aware (neighbouring strips
// share input lines and meet in one L2).  Measured at 7680x4320x3, k = 15: 0.074 ms against 0.127 ms for the strip
// kernel.  Built with -amdgpu-mfma-vgpr-form (build.py): the accumulators stay in VGPRs, no v_accvgpr_read.
// i8 operands are signed: every data byte is XORed with 0x80 (v - 128) and the constant term 128 * sum(taps) = 2^15 goes
// into the accumulator's initial value; taps are <= 127 for k >= 5 (k = 3 has a 128 and stays with the strip kernel).
// The arithmetic is OpenCV's: exact integer sums, one rounding (+2^15) >> 16; results are
// bit-identical to the strip and tile kernels (tests/test_gpu_stages.py).
#include "kernels.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace ipk {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int GM_WX = 64;   // output bytes per strip row = 4 tiles of 16
constexpr int GM_OFF = 24;  // input bytes loaded left of the strip: >= 7 px * 3 channels, a multiple of 8
constexpr int GM_NT = 4;

struct GmArgs {
    const u8 *src;
    size_t sstride;
    u8 *dst;
    size_t dstride;
    int wb, w, h, cn, k;
    int taps[15];
    int nstrips, seg_blocks, nsegs, nitems, nblocks;
    int dst_vec;
};

__device__ __forceinline__ int gm_reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = (p < 0) ? -p : 2 * (n - 1) - p;
    return p;
}

__device__ __forceinline__ long gm_pack(uint32_t lo, uint32_t hi) { return (long)(((unsigned long long)hi << 32) | lo); }

__global__ __launch_bounds__(256) void k_gaussian_mfma(const GmArgs a)
{
    // per wave: mid ring [block slot 0..2][byte plane lo/hi][tile][row group of 4, Gray-coded][column] and the output
    // staging tile.  Layouts are chosen against LDS bank conflicts (they were 53 % of the LDS cycles, and the LDS is what
    // bounds this kernel): a wave's mid write covers 64 consecutive dwords; the column pass reads row groups (2, 3) in
    // lanes c = 0, 2 and (0, 1) in lanes c = 1, 3 -- Gray-coded group slots 0, 1, 3, 2 put the two on different bank halves;
    // the stage tile's rows are 15 chunks apart (30 dwords: 16 rows on 16 distinct even banks); the output tile's 4-dword
    // column groups are XORed with (row >> 1) & 3 (writes 2-way instead of 8-way, the 16-byte reads stay aligned).
    __shared__ uint32_t s_mid[4][3][2][GM_NT][4][16];
    __shared__ __align__(16) uint32_t s_out[4][16][16];
    __shared__ int s_taps[16];
    __shared__ __align__(8) uint2 s_stage[4][16][15];  // per wave: one input block between load order and operand order
    __shared__ int s_off[4][8 * 14];  // edge strips: reflected byte offset of every input byte of the strip
    if (threadIdx.x < 16) s_taps[threadIdx.x] = threadIdx.x < 15 ? a.taps[threadIdx.x] : 0;
    __syncthreads();  // the only workgroup barrier, before any wave leaves
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // XCD-aware order: workgroups go round-robin over the 8 XCDs, each with an L2 of its own, and adjacent strips share
    // input lines (112 bytes in for 64 out).  Workgroup b = 8 i + x is given the i-th group of XCD x's contiguous share
    // of the strips, so neighbours meet in one L2 instead of each fetching the shared lines over the fabric.
    const int per_xcd = (int)gridDim.x >> 3;  // the grid is a multiple of 8 workgroups
    const int lwg = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int item = __builtin_amdgcn_readfirstlane(lwg * 4 + wv);
    if (item >= a.nitems) return;  // whole waves leave; no workgroup barrier below
    const int strip = item % a.nstrips, seg = item / a.nstrips;
    const int X = strip * GM_WX;
    const int i16 = lane & 15, c = lane >> 4;
    const int cn = a.cn, K = a.k, R = K >> 1;
    const int wb = a.wb, h = a.h;

    // ---- the two Toeplitz operands, built once per lane: 8 k-values (bytes) per dword pair ----
    uint32_t rowB[2][2] = {{0, 0}, {0, 0}}, colB[2] = {0, 0};
    for (int m = 0; m < 8; m++) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int d = 32 * b + 8 * c + m - i16 - GM_OFF + R * cn;  // (input byte) - (first input byte of output column i16)
            const int t = cn == 3 ? d / 3 : d;
            const int tap = (d >= 0 && t * cn == d && t < K) ? s_taps[t] : 0;
            rowB[b][m >> 2] |= (uint32_t)tap << (8 * (m & 3));
        }
        const int t = 8 * c + m - i16 - 8 + R;  // window row k = mid row 16 B - 8 + k; output row i16 needs rows +(t - R)
        colB[m >> 2] |= (uint32_t)((t >= 0 && t < K) ? s_taps[t] : 0) << (8 * (m & 3));
    }
    const long rowB0 = gm_pack(rowB[0][0], rowB[0][1]), rowB1 = gm_pack(rowB[1][0], rowB[1][1]);
    const long colBl = gm_pack(colB[0], colB[1]);

    // ---- this lane's six 8-byte input chunks per block: bytes X - 24 + 8 (c + 2u), u = 0..5, of row 16 blk + i16 ----
    const int p0 = X - GM_OFF + 8 * c;
    // Strips whose 112 input bytes lie inside the row (all but the first and the last one or two) load dword pairs.  In the
    // others the chunks that the image's edge cuts go byte by byte through BORDER_REFLECT_101 on the pixel index; the
    // reflected byte offsets are the same for every row, so they are computed once per wave into LDS and the eight byte
    // loads of a chunk are issued together.  The choice is per wave: the common path carries none of this.
    const bool edge_strip = X - GM_OFF < 0 || X - GM_OFF + 8 * 14 > wb;  // uniform
    int *offs = s_off[wv];
    if (edge_strip) {
        for (int q = lane; q < 8 * 14; q += 64) {
            const int pp = X - GM_OFF + q + cn * 65536;  // keep the division on non-negative values
            const int pq = cn == 3 ? pp / 3 : pp;
            offs[q] = gm_reflect101(pq - 65536, a.w) * cn + (pp - pq * cn);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // Coalesced loads: a block is 16 rows x 14 chunks of 8 bytes; load k takes rows 4k..4k+3, lane l < 56 the chunk l % 14 of
    // row 4k + l / 14, so adjacent lanes read adjacent bytes (the MFMA operand layout -- adjacent lanes = adjacent ROWS --
    // made every lane its own memory request: 44 cycles per load instruction).  The block then goes through a per-wave
    // LDS tile and is read back in operand order.
    // The interior strips' loads are unconditional (lanes 56..63 repeat lane 55's; beyond the segment's end the last block
    // is loaded again): a load behind a branch makes the compiler wait for ALL outstanding loads where the paths join --
    // the two blocks "in flight" were being waited for right after they were issued.  The edge strips' byte path keeps
    // its branches and runs as a second copy of the loop (EDGE), chosen per wave.
    const int ll = min(lane, 55);
    const int lrow = ll / 14, lq = ll - 14 * lrow;
    uint2(*stage)[15] = s_stage[wv];  // [16 rows][14 chunks + 1 pad]
    uint32_t(*mid)[2][GM_NT][4][16] = s_mid[wv];
    auto gray = [](int g) { return g ^ (g >> 1); };  // row group -> slot inside a tile: 0, 1, 3, 2
    uint32_t(*ost)[16] = s_out[wv];
    const int Bs = seg * a.seg_blocks, Be = min(Bs + a.seg_blocks, a.nblocks);
    auto run = [&](auto edge_c) {
    constexpr bool EDGE = decltype(edge_c)::value;
    auto load_block = [&](int blk, uint32_t (&ld)[4][2]) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = gm_reflect101(16 * blk + 4 * k + lrow, h);
            const u8 *row = a.src + (uint32_t)y * (uint32_t)a.sstride;  // 32-bit offsets (image < 4 GB, checked at launch): saddr-form loads
            const int p = X - GM_OFF + 8 * lq;
            uint32_t v0 = 0, v1 = 0;
            if (!EDGE) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(row + p);
                v0 = q[0];
                v1 = q[1];
            } else if (p >= 0 && p + 8 <= wb) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(row + p);
                v0 = q[0];
                v1 = q[1];
            } else {
                const int *o = offs + 8 * lq;
                uint32_t bb[8];
#pragma unroll
                for (int m = 0; m < 8; m++) bb[m] = row[o[m]];
                v0 = bb[0] | (bb[1] << 8) | (bb[2] << 16) | (bb[3] << 24);
                v1 = bb[4] | (bb[5] << 8) | (bb[6] << 16) | (bb[7] << 24);
            }
            ld[k][0] = v0;  // raw: the first use (and with it the wait for the load) is the LDS write two steps later
            ld[k][1] = v1;
        }
    };
    // Two register buffers take the blocks alternately (the loop is unrolled twice by hand, so no buffer is ever copied:
    // a copy would be a use, and the wait for the load would move up to it).  A step writes its buffer to the LDS tile
    // and refills it at once with the block two steps ahead.
    uint32_t bufA[4][2], bufB[4][2];
    auto step = [&](int blk, uint32_t (&buf)[4][2]) {
        uint32_t cur[6][2];
        // this block through the LDS tile into operand order: lane (row i16, chunk c) takes chunks c + 2u, u = 0..5
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (lane < 56) stage[4 * k + lrow][lq] = make_uint2(buf[k][0] ^ 0x80808080u, buf[k][1] ^ 0x80808080u);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 6; u++) {
            const uint2 v = stage[i16][c + 2 * u];
            cur[u][0] = v.x;
            cur[u][1] = v.y;
        }
        __builtin_amdgcn_wave_barrier();
        load_block(min(blk + 2, Be), buf);  // two blocks in flight while this one is multiplied
        // ---- row pass of block blk, packed into the ring: all first halves, then all second halves, so that no MFMA
        // waits for the one just issued ----
        const int slot = (blk + 3) % 3;
        v4i acc[GM_NT];
#pragma unroll
        for (int t = 0; t < GM_NT; t++) {
            const v4i c0 = {32768, 32768, 32768, 32768};  // 128 * sum(taps): undoes the -128 of the data bytes
            acc[t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(gm_pack(cur[t][0], cur[t][1]), rowB0, c0, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < GM_NT; t++)
            acc[t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(gm_pack(cur[t + 2][0], cur[t + 2][1]), rowB1, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < GM_NT; t++) {
            // acc[t][r] = mid[row 4 c + r][column 16 t + i16], 16 bits: low bytes of the four rows -> one dword, high bytes another
            const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)acc[t][1], (uint32_t)acc[t][0], 0x05010400u);
            const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)acc[t][3], (uint32_t)acc[t][2], 0x05010400u);
            mid[slot][0][t][gray(c)][i16] = __builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u;
            mid[slot][1][t][gray(c)][i16] = __builtin_amdgcn_perm(p23, p01, 0x07060302u) ^ 0x80808080u;
        }
        __builtin_amdgcn_wave_barrier();  // LDS operations of one wave execute in order; keep the compiler from reordering
        if (blk < Bs + 1) return;         // the column pass of block ob needs blocks ob - 1, ob, ob + 1
        // ---- column pass of output block ob: window rows 16 ob - 8 .. 16 ob + 23 = row groups 4 ob - 2 .. 4 ob + 5 ----
        const int ob = blk - 1;
        const int wg = 2 * c + 2;  // this lane's first row group, counted from block ob - 1
        const int rslot = (ob - 1 + (wg >> 2) + 3) % 3, rg = wg & 3;
        v4i dlo[GM_NT], dhi[GM_NT];
#pragma unroll
        for (int t = 0; t < GM_NT; t++) {
            const long alo = gm_pack(mid[rslot][0][t][gray(rg)][i16], mid[rslot][0][t][gray(rg + 1)][i16]);
            const long ahi = gm_pack(mid[rslot][1][t][gray(rg)][i16], mid[rslot][1][t][gray(rg + 1)][i16]);
            // sum(tap * mid) = S_lo + 256 S_hi + 128 * 256 * (1 + 256); + 2^15 for the rounding
            const v4i k0 = {8454144, 8454144, 8454144, 8454144}, z = {0, 0, 0, 0};
            dlo[t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(alo, colBl, k0, 0, 0, 0);
            dhi[t] = __builtin_amdgcn_mfma_i32_16x16x32_i8(ahi, colBl, z, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < GM_NT; t++) {
            uint32_t s[4];
#pragma unroll
            for (int r = 0; r < 4; r++) s[r] = (uint32_t)dlo[t][r] + ((uint32_t)dhi[t][r] << 8);
            // lane (row i16, c): s[r] is column 16 t + 4 c + r; byte 2 of every sum is the result
            ost[i16][4 * (t ^ ((i16 >> 1) & 3)) + c] = __builtin_amdgcn_perm(s[1], s[0], 0x0c0c0602u) | __builtin_amdgcn_perm(s[3], s[2], 0x06020c0cu);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- store: lane l writes bytes [16 (l & 3), +16) of row l >> 2: 64 contiguous bytes per row ----
        {
            const int orow = 16 * ob + (lane >> 2), ox = X + 16 * (lane & 3);
            const uint4 v = *reinterpret_cast<const uint4 *>(&ost[lane >> 2][4 * ((lane & 3) ^ ((lane >> 3) & 3))]);
            if (orow < h && ox < wb) {
                u8 *o = a.dst + ((uint32_t)orow * (uint32_t)a.dstride + (uint32_t)ox);
                if (a.dst_vec && ox + 16 <= wb) {
                    *reinterpret_cast<uint4 *>(o) = v;
                } else {
                    const uint32_t q[4] = {v.x, v.y, v.z, v.w};
                    for (int i = 0; i < 16 && ox + i < wb; i++) o[i] = (u8)(q[i >> 2] >> (8 * (i & 3)));
                }
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

bool gaussian_mfma_ok(const u8 *src, size_t sstride, int w, int h, int cn, const Taps &taps)
{
    if (taps.k < 7 || taps.k > 15 || !(taps.k & 1) || (cn != 1 && cn != 3)) return false;
    if ((uintptr_t)src % 4 || sstride % 4) return false;
    if (w < 1 || h < 1 || (long long)w * cn > (1 << 28) || (unsigned long long)sstride * (unsigned long long)h >= (1ull << 32)) return false;
    int sum = 0;
    for (int i = 0; i < taps.k; i++) {
        if (taps.t[i] < 0 || taps.t[i] > 127) return false;  // signed 8-bit operands
        sum += taps.t[i];
    }
    return sum == 256;  // the -128 corrections assume it
}

hipError_t launch_gaussian_mfma(const u8 *src, size_t sstride, u8 *dst, size_t dstride, int w, int h, int cn,
                                const Taps &taps, hipStream_t stream)
{
    GmArgs a{};
    a.src = src; a.sstride = sstride; a.dst = dst; a.dstride = dstride;
    a.w = w; a.h = h; a.cn = cn; a.wb = w * cn; a.k = taps.k;
    for (int i = 0; i < taps.k; i++) a.taps[i] = taps.t[i];
    a.nstrips = (a.wb + GM_WX - 1) / GM_WX;
    a.nblocks = (h + 15) / 16;
    // every segment runs two extra blocks (the one above, the one below), yet short segments win: the kernel is bound by
    // the latency of a block's dependent LDS / MFMA steps at the 4 waves per SIMD its LDS allows, and a finer grain evens
    // out the rounds (measured at 7680x4320x3: 12 segments per strip 0.099 ms, 18 0.084, 23 0.080, 34 0.074, 46 0.073)
    int nsegs = std::max(1, std::min((12288 + a.nstrips - 1) / a.nstrips, (a.nblocks + 3) / 4));
    if (const char *e = getenv("IMGPROC_GAUSS_SEGS")) nsegs = std::max(1, std::min(atoi(e), a.nblocks));
    a.seg_blocks = (a.nblocks + nsegs - 1) / nsegs;
    a.nsegs = (a.nblocks + a.seg_blocks - 1) / a.seg_blocks;
    a.nitems = a.nstrips * a.nsegs;
    a.dst_vec = ((uintptr_t)dst % 4 == 0) && (dstride % 4 == 0);
    hipLaunchKernelGGL(k_gaussian_mfma, dim3(((a.nitems + 3) / 4 + 7) / 8 * 8), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace ipk
