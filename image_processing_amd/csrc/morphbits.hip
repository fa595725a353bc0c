// Bit-plane morphology back end: the whole erode/dilate list of a pipeline (morphOps,
// objectPolygonPath.cpp:414-433 / pose_estimate.cpp:405-415, any list of RECT / CROSS / ELLIPSE elements),
// the raster moments (moments(), objectPolygonPath.cpp:449-450) and the final 0/255 mask in ONE launch.
//
// Input is the inRange output (0/255 bytes, 1 B/px); a workgroup packs a tile plus the list's total reach into
// bit rows in LDS (32 px per word), runs every op on the bits (a 3x3 erode is a handful of word operations for 32
// pixels; consecutive same-kind RECT ops are merged on the host into one larger window, which is exact because
// OpenCV's morphology border is neutral), then expands the interior back to bytes and popcounts the moments.
// HBM traffic: 1 B/px read (+ halo from L2) and 1 B/px written instead of 2 B/px per op.
#include "morphbits.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace ipm {

using ipk::u8;

constexpr int TWW = 16;  // largest interior tile width in 32-px words (512 px)
constexpr int TH = 96;   // largest interior tile height
// the launch divides the image into equal tiles no larger than that (1920x1200: 4 x 13 tiles of 15 words x 93 rows)
constexpr int MAX_HALO_ROWS = 56;
constexpr int MAX_HALO_WORDS = 2;  // per side
constexpr int LW_MAX = TWW + 2 * MAX_HALO_WORDS;
constexpr int LS_MAX = LW_MAX + 1;  // row stride in LDS: the data words + one pad word that stays all-ones
constexpr int LH_MAX = TH + MAX_HALO_ROWS;

struct Args {
    const u8 *thresh;  // [planes][h][w] 0/255 bytes, or null when the input comes as bit planes:
    const uint32_t *tbits;  // [planes][h][twords] words, bit i of word j = pixel 32 j + i (bits beyond w are ignored)
    int twords;
    u8 *masks;         // [planes][h][w]
    unsigned long long *mom;  // [planes][6], added into
    const Group *groups;
    int n_groups;
    int w, h, planes;
    int hwl, hwr, ht, hb;  // halo: words left/right, rows top/bottom
    int tiles_x, tiles_y;
    int tww, th;  // interior tile size: words, rows
    int second_order;
};

constexpr int N_WORDS_MAX = LH_MAX * LS_MAX;
constexpr int WORDS_PER_THREAD = (N_WORDS_MAX + 255) / 256;
static_assert(WORDS_PER_THREAD == 13, "the vertical pass is instantiated for 1..13 words per thread");
constexpr int CH = 4;  // words a thread carries through one horizontal window pass at a time

__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, int s) { return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)s); }

// out-of-image masks of one tile word (the pack phase runs before the shared copies of them are visible)
__device__ __forceinline__ uint32_t s_rinv_early(int y, int h) { return (y >= 0 && y < h) ? 0u : 0xFFFFFFFFu; }
__device__ __forceinline__ uint32_t cinv_word(int x_org, int k, int LW, int w)
{
    const int x0 = x_org + k * 32;
    if (k >= LW || x0 >= w || x0 + 32 <= 0) return 0xFFFFFFFFu;
    uint32_t inv = 0;
    if (x0 < 0) inv |= (1u << (-x0)) - 1u;
    if (x0 + 32 > w) inv |= ~((1u << (w - x0)) - 1u);
    return inv;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_bitmorph(const Args a)
{
    // Two bit planes of the tile (current / horizontal-pass output), rows of LS = LW + 1 words: the extra word of every
    // row (and the word before the first row) is all-ones, the neutral value of the erode domain everything is stored
    // in, so the horizontal pass reads its left / right neighbour words without looking at the column index.
    __shared__ uint32_t s_buf[2 * N_WORDS_MAX + 2];
    // bits outside the image (neutral for every op: OpenCV's morphologyDefaultBorderValue) = row part | column part
    __shared__ uint32_t s_rinv[LH_MAX], s_cinv[LS_MAX];
    __shared__ unsigned long long s_red[4][6];
    __shared__ uint32_t s_lut[16];  // nibble -> four 0/255 bytes

    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y; t /= a.tiles_y;
    const int plane = t;
    const int w = a.w, h = a.h;
    const int tww = a.tww, th = a.th;
    const int LW = tww + a.hwl + a.hwr, LS = LW + 1, LH = th + a.ht + a.hb;
    const uint32_t rcp = (65536u + LS - 1) / LS;  // idx / LS == (idx * rcp) >> 16 for idx < 3400 (LS <= 21)
    const int x_org = (tx * tww - a.hwl) * 32;  // pixel x of tile word 0
    const int y_org = ty * th - a.ht;
    const u8 *src = a.thresh ? a.thresh + (size_t)plane * w * h : nullptr;
    const bool vec = true;  // dword accesses may be unaligned (w % 4 != 0): fine for gfx950 global memory (scripts/ubench/unaligned.hip)
    auto mask_of = [](int kind) { return kind ? 0xFFFFFFFFu : 0u; };  // dilate(X) = ~erode(~X)
    uint32_t *const cur = s_buf + 1, *const alt = s_buf + N_WORDS_MAX + 2;

    if (tid == 0) cur[-1] = 0xFFFFFFFFu;
    if (tid < 16) s_lut[tid] = ((tid & 1) ? 0xFFu : 0u) | ((tid & 2) ? 0xFF00u : 0u) | ((tid & 4) ? 0xFF0000u : 0u) | ((tid & 8) ? 0xFF000000u : 0u);
    if (tid < LH) s_rinv[tid] = (y_org + tid >= 0 && y_org + tid < h) ? 0u : 0xFFFFFFFFu;
    if (tid < LS) {
        const int x0 = x_org + tid * 32;
        uint32_t inv = 0xFFFFFFFFu;
        if (tid < LW && x0 < w && x0 + 32 > 0) {
            inv = 0;
            if (x0 < 0) inv |= (1u << (-x0)) - 1u;             // x0 in (-32, 0)
            if (x0 + 32 > w) inv |= ~((1u << (w - x0)) - 1u);  // partially beyond the right edge
        }
        s_cinv[tid] = inv;
    }
    // ---- pack: one 32-px word per thread-iteration, stored in the first group's erode domain ----
    const uint32_t km_first = mask_of(a.groups[0].kind);
    if (a.tbits) {
        // ---- bit planes in (the front kernel / the inRange kernel already hold their result as bits): one aligned
        // word load per tile word; out-of-image bits are forced neutral by the row / column masks ----
        const uint32_t *bsrc = a.tbits + (size_t)plane * h * a.twords;
        const int j_org = tx * tww - a.hwl;  // image word index of tile word 0
        constexpr int PBB = 8;
        for (int base = 0; base < LH * LS; base += 256 * PBB) {
            uint32_t v[PBB];
#pragma unroll
            for (int j = 0; j < PBB; j++) {
                const int idx = base + tid + 256 * j;
                const int r = (int)(((uint32_t)idx * rcp) >> 16), k = idx - r * LS;
                const int y = y_org + r, jw = j_org + k;
                const bool in = idx < LH * LS && k < LW && y >= 0 && y < h && jw >= 0 && jw < a.twords;
                v[j] = in ? bsrc[(size_t)y * a.twords + jw] : 0u;
            }
#pragma unroll
            for (int j = 0; j < PBB; j++) {
                const int idx = base + tid + 256 * j;
                const int r = (int)(((uint32_t)idx * rcp) >> 16), k = idx - r * LS;
                if (idx < LH * LS) cur[idx] = (v[j] ^ km_first) | s_rinv_early(y_org + r, h) | cinv_word(x_org, k, LW, w);
            }
        }
    } else {
        // PB words per thread at a time: all their global loads are issued before the first is used, so a thread waits for
        // HBM two or three times per tile instead of once per word
        constexpr int PB = 5;
        for (int base = 0; base < LH * LS; base += 256 * PB) {
            uint32_t v[PB][8];
            int px0[PB], py[PB];
#pragma unroll
            for (int j = 0; j < PB; j++) {
                const int idx = base + tid + 256 * j;
                const int r = (int)(((uint32_t)idx * rcp) >> 16), k = idx - r * LS;
                const int y = y_org + r, x0 = k == LW ? w : x_org + k * 32;  // the pad word: "outside" -> all-ones
                px0[j] = x0;
                py[j] = (idx < LH * LS && y >= 0 && y < h) ? y : -1;
                const bool fast = vec && py[j] >= 0 && x0 >= 0 && x0 + 32 <= w;
                // unconditional loads (words that take no fast path read the group list instead: always 32 readable bytes)
                const uint32_t *p = fast ? reinterpret_cast<const uint32_t *>(src + (size_t)y * w + x0)
                                         : reinterpret_cast<const uint32_t *>(a.groups);
#pragma unroll
                for (int q = 0; q < 8; q++) v[j][q] = p[q];
            }
#pragma unroll
            for (int j = 0; j < PB; j++) {
                const int idx = base + tid + 256 * j;
                const int y = py[j], x0 = px0[j];
                uint32_t bits = 0, inv = 0xFFFFFFFFu;
                if (y >= 0 && x0 < w && x0 + 32 > 0) {
                    inv = 0;
                    if (vec && x0 >= 0 && x0 + 32 <= w) {
                        // 0/255 bytes -> bits: two dwords (8 px) per byte of the word, one 4x8-bit dot product each
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t hi4 = __builtin_amdgcn_udot4(v[j][2 * q + 1] & 0x01010101u, 0x80402010u, 0u, false);
                            bits |= __builtin_amdgcn_udot4(v[j][2 * q] & 0x01010101u, 0x08040201u, hi4, false) << (8 * q);
                        }
                    } else {  // a word that the image's left / right edge cuts
                        if (x0 < 0) inv |= (1u << (-x0)) - 1u;
                        if (x0 + 32 > w) inv |= ~((1u << (w - x0)) - 1u);
                        const u8 *row = src + (size_t)y * w;
                        for (int i = 0; i < 32; i++)
                            if (x0 + i >= 0 && x0 + i < w && row[x0 + i]) bits |= 1u << i;
                    }
                }
                if (idx < LH * LS) cur[idx] = (bits ^ km_first) | inv;
            }
        }
    }
    __syncthreads();

    // Every group: for each distinct span of its element, a horizontal window AND of the whole tile (cur -> alt), then
    // the AND over the rows that use that span, accumulated in registers; the result goes back into cur already in the
    // NEXT group's erode domain.  Rows a group can no longer produce correctly (reach of the groups so far) are skipped.
    // All loop control is uniform (scalar) and amortised over the CH / WORDS_PER_THREAD words a thread holds.
    for (int gi = 0; gi < a.n_groups; gi++) {
        const Group &g = a.groups[gi];
        const bool last = gi + 1 == a.n_groups;
        const uint32_t flip = mask_of(g.kind) ^ (last ? 0u : mask_of(a.groups[gi + 1].kind));
        const int h0 = g.t_in * LS, h1 = (LH - g.b_in) * LS;
        const int v0 = g.t_out * LS, v1 = (LH - g.b_out) * LS;
        uint32_t acc[WORDS_PER_THREAD];
#pragma unroll
        for (int i = 0; i < WORDS_PER_THREAD; i++) acc[i] = 0xFFFFFFFFu;
        // this thread's output words are v0 + tid + 256 i, i < n_it; the last one is clamped into the valid range (its
        // surplus lanes compute but never store)
        const int n_it = (v1 - v0 + 255) >> 8;
        const int vfirst = v0 + tid, vlast = min(v0 + tid + 256 * (n_it - 1), v1 - 1);
        bool cur_read_late = false;  // did a vertical pass read `cur` itself (a span without horizontal extent)?
        for (int si = 0; si < g.n_spans; si++) {
            const Span sp = g.spans[si];
            const uint32_t *hsrc = cur;
            if (sp.lx != 0 || sp.hx != 0) {
                // window AND of the bits x+lx .. x+hx on the 96-bit string (previous, this, next word): doubling
                // (f_2n(x) = f_n(x) & f_n(x+n)) up to the largest power of two in the window, then the string is read at
                // offset 32 + lx and at the offset that ends the window at x+hx
                const int len = sp.hx - sp.lx + 1, ex = 32 + sp.lx;  // len 1..65, ex 0..64
                auto hpass = [&](auto cn, int base) {  // C words per thread starting at base + tid
                    constexpr int C = decltype(cn)::value;
                    uint32_t a0[C], a1[C], a2[C];
#pragma unroll
                    for (int j = 0; j < C; j++) {
                        const int idx = min(base + tid + 256 * j, h1 - 1);
                        a0[j] = cur[idx - 1];
                        a1[j] = cur[idx];
                        a2[j] = cur[idx + 1];
                    }
                    auto step = [&](int s) {  // a(x) &= a(x + s), 1 <= s <= 32; ones come in from beyond the string
                        if (s == 32) {
#pragma unroll
                            for (int j = 0; j < C; j++) a0[j] &= a1[j], a1[j] &= a2[j];
                        } else {
#pragma unroll
                            for (int j = 0; j < C; j++) {
                                a0[j] &= funnel(a1[j], a0[j], s);
                                a1[j] &= funnel(a2[j], a1[j], s);
                                a2[j] &= funnel(0xFFFFFFFFu, a2[j], s);
                            }
                        }
                    };
                    int have = 1;
#pragma unroll 1
                    while (have * 2 <= len) {
                        step(have);
                        have *= 2;
                    }
                    // a(x) = AND of `have` bits from x on (have = the largest power of two <= len); the window of len
                    // bits is two of those, overlapping: one starting at x+lx, one ending at x+hx
                    auto ext = [&](int j, int e) {  // the string read at bit offset e (0..64)
                        return e >= 64 ? a2[j] : e >= 32 ? funnel(a2[j], a1[j], e - 32) : funnel(a1[j], a0[j], e);
                    };
                    const int ex2 = ex + len - have;
#pragma unroll
                    for (int j = 0; j < C; j++) {
                        const int idx = base + tid + 256 * j;
                        uint32_t res = ext(j, ex);
                        if (have < len) res &= ext(j, ex2);
                        if (idx < h1) alt[idx] = res;
                    }
                };
                int base = h0;
#pragma unroll 1
                for (; h1 - base > 2 * 256; base += 256 * CH) hpass(std::integral_constant<int, CH>(), base);
                if (h1 - base > 256)
                    hpass(std::integral_constant<int, 2>(), base);
                else if (h1 - base > 0)
                    hpass(std::integral_constant<int, 1>(), base);
                __syncthreads();
                hsrc = alt;
            }
            // AND over the rows of the span, N = n_it words per thread: one address per row + immediate offsets
            auto vpass = [&](auto nn) {
                constexpr int N = decltype(nn)::value;
                auto and_row = [&](int dy) {
                    const uint32_t *rowp = hsrc + dy * LS;
                    const uint32_t *pf = rowp + vfirst;
#pragma unroll
                    for (int i = 0; i < N - 1; i++) acc[i] &= pf[256 * i];
                    acc[N - 1] &= rowp[vlast];
                };
                if (sp.rows == 0) {
#pragma unroll 1
                    for (int dy = sp.dy_lo; dy <= sp.dy_hi; dy++) and_row(dy);
                } else {
#pragma unroll 1
                    for (uint32_t m = sp.rows; m; m &= m - 1) and_row(sp.dy_lo + __ffs(m) - 1);
                }
            };
            switch (n_it) {
            case 1: vpass(std::integral_constant<int, 1>()); break;
            case 2: vpass(std::integral_constant<int, 2>()); break;
            case 3: vpass(std::integral_constant<int, 3>()); break;
            case 4: vpass(std::integral_constant<int, 4>()); break;
            case 5: vpass(std::integral_constant<int, 5>()); break;
            case 6: vpass(std::integral_constant<int, 6>()); break;
            case 7: vpass(std::integral_constant<int, 7>()); break;
            case 8: vpass(std::integral_constant<int, 8>()); break;
            case 9: vpass(std::integral_constant<int, 9>()); break;
            case 10: vpass(std::integral_constant<int, 10>()); break;
            case 11: vpass(std::integral_constant<int, 11>()); break;
            case 12: vpass(std::integral_constant<int, 12>()); break;
            default: vpass(std::integral_constant<int, 13>()); break;
            }
            if (hsrc == alt && si + 1 < g.n_spans) __syncthreads();  // alt is rewritten by the next span
            cur_read_late |= hsrc == cur;
        }
        // the write-back below overwrites cur: its last readers were the horizontal passes (all behind a barrier
        // already) unless a vertical pass read it directly
        if (cur_read_late) __syncthreads();
#pragma unroll
        for (int i = 0; i < WORDS_PER_THREAD; i++) {
            const int idx = v0 + tid + 256 * i;
            if (idx < v1) {
                const int r = (int)(((uint32_t)idx * rcp) >> 16), k = idx - r * LS;
                cur[idx] = last ? (acc[i] ^ flip) : ((acc[i] ^ flip) | s_rinv[r] | s_cinv[k]);
            }
        }
        __syncthreads();
    }

    // ---- interior: bytes out + moments ----
    // thread = (column word tid & 15, row tid >> 4), 16 rows per pass: no index division, the row pointer just advances;
    // the first-order sums of a thread's <= 6 words fit 32 bits (x0, y < 2^24: checked at launch)
    unsigned long long m[6] = {0, 0, 0, 0, 0, 0};
    {
        const int k = tid & 15;
        const int x0 = (tx * tww + k) * 32;
        const bool col_ok = k < tww && x0 < w;
        const bool full = x0 + 32 <= w;
        const uint32_t edge = (col_ok && !full) ? (1u << (w - x0)) - 1u : 0xFFFFFFFFu;
        u8 *dst = a.masks ? a.masks + (size_t)plane * w * h : nullptr;
        uint32_t m0 = 0, m1 = 0, m2 = 0;
        if (col_ok) {
            int y = ty * th + (tid >> 4);
            const uint32_t *lp = cur + ((tid >> 4) + a.ht) * LS + (k + a.hwl);
            u8 *o = dst + (size_t)y * w + x0;
            for (int r = tid >> 4; r < th && y < h; r += 16, y += 16, lp += 16 * LS, o += (size_t)16 * w) {
                const uint32_t bits = *lp & edge;
                if (dst) {
                    if (vec && full) {
                        uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
#pragma unroll
                        for (int q = 0; q < 8; q++) {  // nibble -> four 0/255 bytes through the 16-entry table
                            const uint32_t i4 = (q == 0 ? (bits << 2) : (bits >> (4 * q - 2))) & 60u;
                            o32[q] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_lut) + i4);
                        }
                    } else {
                        for (int i = 0; i < 32 && x0 + i < w; i++) o[i] = ((bits >> i) & 1u) ? 255 : 0;
                    }
                }
                const uint32_t cnt = __popc(bits);
                const uint32_t bs = __popc(bits & 0xAAAAAAAAu) + 2 * __popc(bits & 0xCCCCCCCCu) + 4 * __popc(bits & 0xF0F0F0F0u) +
                                    8 * __popc(bits & 0xFF00FF00u) + 16 * __popc(bits & 0xFFFF0000u);  // sum of bit indices
                const uint32_t sx = __umul24((uint32_t)x0, cnt) + bs;
                m0 += cnt;
                m1 += sx;
                m2 += __umul24((uint32_t)y, cnt);
                if (a.second_order && bits) {
                    // sum of squared bit indices i = sum_j 2^j b_j(i): i^2 = sum_j 4^j b_j + sum_{j<k} 2^(j+k+1) b_j b_k
                    const uint32_t B[5] = {0xAAAAAAAAu, 0xCCCCCCCCu, 0xF0F0F0F0u, 0xFF00FF00u, 0xFFFF0000u};
                    unsigned long long bs2 = 0;
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        bs2 += (unsigned long long)__popc(bits & B[j]) << (2 * j);
#pragma unroll
                        for (int kk = j + 1; kk < 5; kk++) bs2 += (unsigned long long)__popc(bits & B[j] & B[kk]) << (j + kk + 1);
                    }
                    m[3] += (unsigned long long)x0 * x0 * cnt + 2ull * x0 * bs + bs2;
                    m[4] += (unsigned long long)y * sx;
                    m[5] += (unsigned long long)y * y * cnt;
                }
            }
        }
        m[0] = m0;
        m[1] = m1;
        m[2] = m2;
    }
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        unsigned long long v = m[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) s_red[wv][i] = v;
    }
    __syncthreads();
    if (tid < 6) {
        const unsigned long long v = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
        if (v) atomicAdd(a.mom + (size_t)plane * 6 + tid, v);
    }
}

// ---------------------------------------------------------------------------------------------
// Strip kernel for the reference's two morphOps lists on bit planes: no LDS tiles, no barriers, no halo recomputation
// along x.  One WAVE owns a strip of 64 words (2048 px; a 1920-px row is one strip) of one plane and marches down a
// segment of rows; lane = one 32-px word.  Every input row runs through all groups of the list in the same iteration:
//   horizontal window AND: the 64-bit string (this word, right neighbour) re-based at x + lx from (left, this, right)
//     -- neighbours by DPP -- then shift-doubling, 4 ops per step;
//   vertical window AND: doubling over rows in registers (h & h[-1] -> p2, p2 & p2[-2] -> p4, p4 & p4[-4] -> p8; the
//     row loop is unrolled 8x so every ring slot is a register name) and one last, overlapping combine with the level-P
//     value of (height - P) rows ago (register ring, or a per-wave LDS ring with constant slots when that is > 2 rows);
//   domain change for the next group: (out ^ flip) | out-of-image bits, one v_bitop3.
// The list is a template argument (window offsets, heights and kinds are constants: no scalar loop control, no
// per-group parameter registers -- a version with run-time lists spent more scalar than vector instructions and ran
// no faster than the tile kernel).  Instantiated for morphOps of objectPolygonPath.cpp:416-432 (merged: E 9x9,
// D 15x15 at -8..+6, E 5x5, D 15x15, E 9x9) and of pose_estimate.cpp:407-414 (E 5x5, D 15x15, E 5x5); every other list
// stays with k_bitmorph.  ~24 VALU per group and 32-px word, against ~25 per word, group AND halo-inflated tile word
// (x1.8 for polygon_path) plus index arithmetic there.
// ---------------------------------------------------------------------------------------------
struct SpanC {  // one distinct row span of a non-RECT element, as the planner's Span: bit i of rows = element row dlo + i
    int lx, hx, dlo;
    unsigned rows;
};
struct WinC {  // one group as the planner holds it: a merged RECT window (nsp == 0), or an element's spans inside the box
    int kind, lx, hx, dlo, dhi;
    int nsp;
    SpanC sp[3];
};
struct ListPolygonPath {
    static constexpr int NG = 5;
    static constexpr WinC get(int g)
    {
        constexpr WinC t[NG] = {{0, -4, 4, -4, 4}, {1, -8, 6, -8, 6}, {0, -2, 2, -2, 2}, {1, -8, 6, -8, 6}, {0, -4, 4, -4, 4}};
        return t[g];
    }
};
struct ListPoseEstimate {
    static constexpr int NG = 3;
    static constexpr WinC get(int g)
    {
        constexpr WinC t[NG] = {{0, -2, 2, -2, 2}, {1, -8, 6, -8, 6}, {0, -2, 2, -2, 2}};
        return t[g];
    }
};

// the stress preset's list (BASELINE configs[4]): erode, then dilate with getStructuringElement(MORPH_ELLIPSE, 9x9):
// rows of 1, 7, 7, 9, 9, 9, 7, 7, 1 pixels
struct ListOpenEllipse9 {
    static constexpr int NG = 2;
    static constexpr WinC get(int g)
    {
        constexpr WinC t[NG] = {{0, -4, 4, -4, 4, 3, {{0, 0, -4, 0x101u}, {-3, 3, -3, 0x63u}, {-4, 4, -1, 0x7u}}},
                                {1, -4, 4, -4, 4, 3, {{0, 0, -4, 0x101u}, {-3, 3, -3, 0x63u}, {-4, 4, -1, 0x7u}}}};
        return t[g];
    }
};

template <class F, int... Is>
__device__ __forceinline__ void bs_for_each(F &f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}

struct StripArgs {
    const uint32_t *tbits;  // [planes][h][twords]
    int twords;
    u8 *masks;  // [planes][h][w] or null
    unsigned long long *mom;
    int w, h, planes;
    int hwl, valid_words, n_strips, seg_rows, n_segs, total_items;
};

constexpr int bs_pow2_le(int v) { return v >= 8 ? 8 : v >= 4 ? 4 : v >= 2 ? 2 : 1; }
constexpr bool bs_uses_lds(const WinC &w)  // a RECT window whose last vertical combine reaches more than 2 rows back
{
    const int H = w.dhi - w.dlo + 1;
    return w.nsp == 0 && (H - bs_pow2_le(H)) > 2;
}

// AND of the bits x+LX .. x+HX of a row (x = this lane's word, L / R = its neighbours' words; ones beyond the string)
template <int LX, int HX>
__device__ __forceinline__ uint32_t bs_hwindow(uint32_t x, uint32_t L, uint32_t R)
{
    static_assert(LX <= 0 && LX >= -31 && HX >= LX && HX - LX < 33, "window offsets");
    constexpr int ex = 32 + LX, len = HX - LX + 1;
    uint32_t b0 = x, b1 = R;
    if (ex != 32) {
        b0 = funnel(x, L, ex);
        b1 = funnel(R, x, ex);
    }
    if (len >= 2) { b0 &= funnel(b1, b0, 1); if (len > 2) b1 &= funnel(0xFFFFFFFFu, b1, 1); }
    if (len >= 4) { b0 &= funnel(b1, b0, 2); if (len > 4) b1 &= funnel(0xFFFFFFFFu, b1, 2); }
    if (len >= 8) { b0 &= funnel(b1, b0, 4); if (len > 8) b1 &= funnel(0xFFFFFFFFu, b1, 4); }
    if (len >= 16) { b0 &= funnel(b1, b0, 8); if (len > 16) b1 &= funnel(0xFFFFFFFFu, b1, 8); }
    if (len >= 32) { b0 &= funnel(b1, b0, 16); if (len > 32) b1 &= funnel(0xFFFFFFFFu, b1, 16); }
    constexpr int have = len >= 32 ? 32 : len >= 16 ? 16 : len >= 8 ? 8 : len >= 4 ? 4 : len >= 2 ? 2 : 1;
    if (have < len) b0 &= funnel(b1, b0, len - have);  // the window is two overlapping runs of `have` bits
    return b0;
}

template <class LIST, bool M2>
__global__ __launch_bounds__(256) void k_bitstrip(const StripArgs a)
{
    constexpr int NG = LIST::NG;
    // per group: window height, doubling level, distance of the last combine; which groups keep that ring in LDS
    constexpr int HT = [] { int s = 0; for (int g = 0; g < NG; g++) s += -LIST::get(g).dlo; return s; }();
    constexpr int HB = [] { int s = 0; for (int g = 0; g < NG; g++) s += LIST::get(g).dhi; return s; }();
    constexpr int N_LDS = [] { int n = 0; for (int g = 0; g < NG; g++) n += bs_uses_lds(LIST::get(g)); return n; }();
    static_assert(N_LDS >= 0 && N_LDS <= NG, "ring count");
    __shared__ uint32_t s_ring[4][N_LDS > 0 ? N_LDS : 1][8][64];  // per wave: the last 8 rows of a group's level-P results
    __shared__ uint32_t s_lut[16];                                // nibble -> four 0/255 bytes
    if (threadIdx.x < 16)
        s_lut[threadIdx.x] = ((threadIdx.x & 1) ? 0xFFu : 0u) | ((threadIdx.x & 2) ? 0xFF00u : 0u) |
                             ((threadIdx.x & 4) ? 0xFF0000u : 0u) | ((threadIdx.x & 8) ? 0xFF000000u : 0u);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
    if (item >= a.total_items) return;  // whole waves leave; no barrier below
    const int strip = item % a.n_strips;
    const int rest = item / a.n_strips;
    const int seg = rest % a.n_segs;
    const int plane = rest / a.n_segs;
    const int w = a.w, h = a.h;
    const int ys = seg * a.seg_rows, ye = min(ys + a.seg_rows, h);
    const int j = strip * a.valid_words - a.hwl + lane;  // this lane's word of the row
    const int x0 = j * 32;
    const bool in_row = j >= 0 && j < a.twords;
    const bool writer = in_row && lane >= a.hwl && lane < a.hwl + a.valid_words;
    uint32_t cinv = 0xFFFFFFFFu;  // bits outside the image (neutral in every group's erode domain)
    if (in_row) cinv = x0 + 32 > w ? ~((1u << (w - x0)) - 1u) : 0u;
    const uint32_t *src = a.tbits + (size_t)plane * h * a.twords + (in_row ? j : 0);
    u8 *mrow = a.masks ? a.masks + (size_t)plane * w * h : nullptr;  // uniform

    uint32_t prev1[NG], ring2[NG][2], ring4[NG][4], ringp[NG][2];
    uint32_t ringe[NG][3][8];  // non-RECT groups: the last 8 rows of every span's horizontal window (slots = register names)
#pragma unroll
    for (int g = 0; g < NG; g++) {
        prev1[g] = ring2[g][0] = ring2[g][1] = ringp[g][0] = ringp[g][1] = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < 4; q++) ring4[g][q] = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < 24; q++) ringe[g][q >> 3][q & 7] = 0xFFFFFFFFu;
    }
    uint32_t m00 = 0, m01 = 0, bsa[5] = {0, 0, 0, 0, 0};  // first-order sums of this lane's word column (u32: <= 512 rows)
    unsigned long long m20 = 0, m11 = 0, m02 = 0;
    uint32_t(*ring)[8][64] = s_ring[wv];
    const int r0 = ys - HT;  // first input row
    const int n_iter = (ye - ys) + HT + HB;
    auto load = [&](int r) -> uint32_t {
        const bool ok = in_row && r >= 0 && r < h;
        return ok ? src[(size_t)r * a.twords] : 0u;
    };
    constexpr uint32_t km_first = LIST::get(0).kind ? 0xFFFFFFFFu : 0u, km_last = LIST::get(NG - 1).kind ? 0xFFFFFFFFu : 0u;
#ifndef BS_PF
#define BS_PF 8  // rows of load look-ahead (a register ring).  The plane words come from HBM through a memory system that is
                 // saturated with this kernel's own mask stores (scripts/ubench/write_bw.hip `like` / `b` / `k`: real loads
                 // beside the stores cost 28 % of the store rate, cache-hit loads nothing, eight rows ahead give 11 % back);
                 // measured here: c5 +1.8 % against 3 rows, pp / pe unchanged
#endif
    static_assert(BS_PF == 1 || BS_PF == 2 || BS_PF == 4 || BS_PF == 8, "the ring slot (row % BS_PF) must be a constant of the 8x unrolled loop");
    uint32_t nxt[8];
#pragma unroll
    for (int q = 0; q < BS_PF; q++) nxt[q] = load(r0 + q);
    // whole batches of 8 rows (ring slots are register names): the rows a last, partial batch runs beyond the segment are
    // computed and dropped
    for (int ib = 0; ib < n_iter; ib += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int r = r0 + ib + u;
            // row r sits in ring slot (r - r0) % BS_PF; with BS_PF in {1, 2, 4, 8} the slot is a constant of the unrolled loop
            const uint32_t in = nxt[u % BS_PF];
            nxt[u % BS_PF] = load(r + BS_PF);
            uint32_t x = (in ^ km_first) | cinv | ((r >= 0 && r < h) ? 0u : 0xFFFFFFFFu);
            auto group = [&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr WinC W = LIST::get(g);
                constexpr int H = W.dhi - W.dlo + 1, P = bs_pow2_le(H), dist = H - P;
                // rows of reach below summed over groups 0..g, and the LDS ring this group uses (if any)
                constexpr int below = [] { int b = 0; for (int q = 0; q <= g; q++) b += LIST::get(q).dhi; return b; }();
                constexpr int lds_i = [] { int n = 0; for (int q = 0; q < g; q++) n += bs_uses_lds(LIST::get(q)); return n; }();
                const uint32_t R = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x130, 0xF, 0xF, true);  // lane+1 (lane 63: 0, halo)
                const uint32_t L = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x138, 0xF, 0xF, true);  // lane-1
                uint32_t out;
                if constexpr (W.nsp == 0) {
                    // ---- horizontal: AND of bits x+lx .. x+hx ----
                    const uint32_t b0 = bs_hwindow<W.lx, W.hx>(x, L, R);
                    // ---- vertical: AND of this and the previous height-1 rows of b0 ----
                    out = b0;
                    if (P >= 2) {
                        out = b0 & prev1[g];
                        prev1[g] = b0;
                    }
                    if (P >= 4) {
                        const uint32_t p2 = out;
                        out = p2 & ring2[g][u & 1];
                        ring2[g][u & 1] = p2;
                    }
                    if (P >= 8) {
                        const uint32_t p4 = out;
                        out = p4 & ring4[g][u & 3];
                        ring4[g][u & 3] = p4;
                    }
                    if (dist > 2) {  // the level-P value of `dist` rows ago: LDS ring, constant slots
                        ring[lds_i][u][lane] = out;
                        __builtin_amdgcn_wave_barrier();  // LDS ops of one wave execute in order; keep the compiler from reordering
                        out &= ring[lds_i][(u - dist) & 7][lane];
                        __builtin_amdgcn_wave_barrier();
                    } else if (dist > 0) {  // 1 or 2 rows ago: two registers
                        const uint32_t pp = out;
                        out = pp & ringp[g][(u - dist) & 1];
                        ringp[g][u & 1] = pp;
                    }
                } else {
                    // ---- an element of row spans (ELLIPSE / CROSS): every distinct span's horizontal window of this row,
                    // kept for 8 rows; the output row is the AND over the element's rows of the window each one uses, `age`
                    // rows ago (the row that just came in is the element's bottom row) ----
                    static_assert(H <= 9, "the span rings hold 8 rows");
                    out = 0xFFFFFFFFu;
                    auto span = [&](auto sc) {
                        constexpr int si = decltype(sc)::value;
                        constexpr SpanC S = W.sp[si];
                        const uint32_t hv = bs_hwindow<S.lx, S.hx>(x, L, R);
#pragma unroll
                        for (int i = 0; i < 9; i++) {
                            if (!((S.rows >> i) & 1u)) continue;
                            const int age = W.dhi - (S.dlo + i);
                            out &= age == 0 ? hv : ringe[g][si][(u - age) & 7];  // age 8: the slot about to be replaced
                        }
                        ringe[g][si][u & 7] = hv;
                    };
                    bs_for_each(span, std::make_integer_sequence<int, W.nsp>{});
                }
                // ---- into the next group's erode domain; its rows outside the image are neutral ----
                if (g + 1 < NG) {
                    constexpr uint32_t flip = (W.kind != LIST::get(g + 1 < NG ? g + 1 : g).kind) ? 0xFFFFFFFFu : 0u;
                    const int yg = r - below;
                    x = (out ^ flip) | cinv | ((yg >= 0 && yg < h) ? 0u : 0xFFFFFFFFu);
                } else {
                    x = out;
                }
            };
            bs_for_each(group, std::make_integer_sequence<int, NG>{});
            const int y = r - HB;
            const bool row_out = y >= ys && y < ye;  // uniform: the pipeline is still filling / past the segment
            const uint32_t bits = row_out ? (x ^ km_last) & ~cinv : 0u;
            if (a.masks && row_out) {
                // Bytes out, each store instruction one contiguous KB per wave, starting at the strip's first owned word
                // (whole 128-byte lines when the row pitch allows): lane l takes the 16 pixels [16 l, 16 l + 16) of the
                // strip's first (second) 1024 from lane hwl + l/2 (hwl + 32 + l/2) through the LDS crossbar (a lane
                // storing its own 32 bytes would leave every 64-byte line to two instructions).
                const uint32_t wbits = writer ? bits : 0u;
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int sl = a.hwl + 32 * half + (lane >> 1);  // source lane = word of the strip (64: none)
                    const uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute(sl * 4, (int)wbits);
                    const uint32_t b16 = (lane & 1) ? word >> 16 : word;
                    uint32_t o[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint32_t i4 = (q == 0 ? (b16 << 2) : (b16 >> (4 * q - 2))) & 60u;
                        o[q] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_lut) + i4);
                    }
                    const int xs = (strip * a.valid_words - a.hwl + sl) * 32 + 16 * (lane & 1);  // first pixel of this lane's piece
                    const bool src_writer = sl >= a.hwl && sl < a.hwl + a.valid_words;
                    const int nd = src_writer && xs >= 0 ? min(4, (w - xs) >> 2) : 0;  // dwords inside the row (w % 4 == 0)
                    uint32_t *o32 = reinterpret_cast<uint32_t *>(mrow + (size_t)y * w + xs);
                    if ((w & 15) == 0) {  // uniform; a lane's 16 bytes are inside the row or outside.  Kept apart from the
                        // general case: with both in one if / else the compiler hoists the store of o[0] out of them and
                        // the 16-byte store becomes a 4-byte and a 12-byte one
                        if (nd == 4) *reinterpret_cast<uint4 *>(o32) = make_uint4(o[0], o[1], o[2], o[3]);
                    } else if (nd == 4) {
                        *reinterpret_cast<uint4 *>(o32) = make_uint4(o[0], o[1], o[2], o[3]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 3; q++)
                            if (q < nd) o32[q] = o[q];
                    }
                }
            }
            if (writer) {
                const uint32_t cnt = __popc(bits);
                m00 += cnt;
                m01 = __umul24(cnt, (uint32_t)(y - ys)) + m01;  // relative to the segment: fits 32 bits for any h
                bsa[0] += __popc(bits & 0xAAAAAAAAu);
                bsa[1] += __popc(bits & 0xCCCCCCCCu);
                bsa[2] += __popc(bits & 0xF0F0F0F0u);
                bsa[3] += __popc(bits & 0xFF00FF00u);
                bsa[4] += __popc(bits & 0xFFFF0000u);
                if (M2 && bits) {
                    const uint32_t B[5] = {0xAAAAAAAAu, 0xCCCCCCCCu, 0xF0F0F0F0u, 0xFF00FF00u, 0xFFFF0000u};
                    unsigned long long bs = 0, bs2 = 0;
#pragma unroll
                    for (int q = 0; q < 5; q++) {
                        bs += (unsigned long long)__popc(bits & B[q]) << q;
                        bs2 += (unsigned long long)__popc(bits & B[q]) << (2 * q);
#pragma unroll
                        for (int kk = q + 1; kk < 5; kk++) bs2 += (unsigned long long)__popc(bits & B[q] & B[kk]) << (q + kk + 1);
                    }
                    const unsigned long long X = (unsigned long long)(uint32_t)x0;
                    m20 += X * X * cnt + 2ull * X * bs + bs2;
                    m11 += (unsigned long long)(uint32_t)y * (X * cnt + bs);
                    m02 += (unsigned long long)(uint32_t)y * (uint32_t)y * cnt;
                }
            }
        }
    }
    // sum of x over this lane's set bits = x0 * count + sum of bit indices
    unsigned long long s0 = m00, s2 = (unsigned long long)m01 + (unsigned long long)(uint32_t)ys * m00;
    unsigned long long s1 = (unsigned long long)(uint32_t)(in_row ? x0 : 0) * m00 + bsa[0] + 2ull * bsa[1] + 4ull * bsa[2] + 8ull * bsa[3] + 16ull * bsa[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_down(s0, off, 64);
        s1 += __shfl_down(s1, off, 64);
        s2 += __shfl_down(s2, off, 64);
        if (M2) {
            m20 += __shfl_down(m20, off, 64);
            m11 += __shfl_down(m11, off, 64);
            m02 += __shfl_down(m02, off, 64);
        }
    }
    if (lane == 0 && s0) {
        unsigned long long *o = a.mom + (size_t)plane * 6;
        atomicAdd(o + 0, s0);
        atomicAdd(o + 1, s1);
        atomicAdd(o + 2, s2);
        if (M2) {
            atomicAdd(o + 3, m20);
            atomicAdd(o + 4, m11);
            atomicAdd(o + 5, m02);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side: turn the pipeline's element list into groups
// ---------------------------------------------------------------------------------------------
namespace {
struct HostGroup {  // one (possibly merged) erode/dilate while planning
    int kind, nrows, ay, rect;
    int lx[128], hx[128];  // per element row (RECT: row 0 only)
};
}  // namespace

bool plan_bitmorph(const imgproc_pipeline &pipe, const ipk::Element *elems, Plan *out)
{
    std::vector<HostGroup> hg;
    for (int i = 0; i < pipe.n_morph; i++) {
        const ipk::Element &e = elems[i];
        HostGroup g{};
        g.kind = pipe.morph[i].kind == IMGPROC_DILATE ? 1 : 0;
        g.nrows = e.kh;
        g.ay = e.ay;
        g.rect = e.is_rect ? 1 : 0;
        for (int j = 0; j < e.kh; j++) {
            const uint32_t bits = e.rows[j];
            if (!bits) {
                g.lx[j] = 1; g.hx[j] = 0;  // empty
                continue;
            }
            const int first = __builtin_ctz(bits), last = 31 - __builtin_clz(bits);
            if (bits != ((last == 31 ? 0u : (1u << (last + 1))) - (1u << first))) return false;  // not one contiguous span
            g.lx[j] = first - e.ax;
            g.hx[j] = last - e.ax;
        }
        // merge with the previous group when both are RECT of the same kind and the merged window stays within +-32
        if (g.rect && !hg.empty()) {
            HostGroup &q = hg.back();
            if (q.rect && q.kind == g.kind) {
                const int lx = q.lx[0] + g.lx[0], hx = q.hx[0] + g.hx[0];
                const int top = q.ay + g.ay, bot = (q.nrows - 1 - q.ay) + (g.nrows - 1 - g.ay);
                if (lx >= -32 && hx <= 32 && top + bot + 1 <= 127) {
                    q.lx[0] = lx; q.hx[0] = hx;
                    q.ay = top; q.nrows = top + bot + 1;
                    continue;
                }
            }
        }
        hg.push_back(g);
    }
    if (hg.empty() || hg.size() > 32) return false;
    Plan p{};
    int hl = 0, hr = 0, ht = 0, hb = 0;
    for (const HostGroup &g : hg) {
        Group d{};
        d.kind = g.kind;
        d.t_in = ht; d.b_in = hb;
        int lo = 0, hi = 0;
        if (g.rect) {
            d.n_spans = 1;
            d.spans[0] = Span{(int8_t)g.lx[0], (int8_t)g.hx[0], (int16_t)(-g.ay), (int16_t)(g.nrows - 1 - g.ay), 0u};
            lo = g.lx[0] < 0 ? g.lx[0] : 0;
            hi = g.hx[0] > 0 ? g.hx[0] : 0;
        } else {
            for (int j = 0; j < g.nrows; j++) {
                if (g.hx[j] < g.lx[j]) continue;
                lo = g.lx[j] < lo ? g.lx[j] : lo;
                hi = g.hx[j] > hi ? g.hx[j] : hi;
                const int dy = j - g.ay;
                int si = 0;
                while (si < d.n_spans && !(d.spans[si].lx == g.lx[j] && d.spans[si].hx == g.hx[j])) si++;
                if (si == d.n_spans) {
                    if (d.n_spans == MAX_SPANS) return false;
                    d.spans[d.n_spans++] = Span{(int8_t)g.lx[j], (int8_t)g.hx[j], (int16_t)dy, (int16_t)dy, 0u};
                }
                Span &sp = d.spans[si];  // rows arrive in increasing dy, so dy_lo is already final
                sp.rows |= 1u << (dy - sp.dy_lo);
                sp.dy_hi = (int16_t)dy;
            }
            if (d.n_spans == 0) {  // an all-empty element: OpenCV then takes the min/max over nothing; keep it off this path
                return false;
            }
        }
        hl += -lo; hr += hi;
        ht += g.ay > 0 ? g.ay : 0;
        hb += (g.nrows - 1 - g.ay) > 0 ? (g.nrows - 1 - g.ay) : 0;
        d.t_out = ht; d.b_out = hb;
        p.groups.push_back(d);
    }
    p.hwl = (hl + 31) / 32; p.hwr = (hr + 31) / 32;
    p.ht = ht; p.hb = hb;
    if (p.hwl > MAX_HALO_WORDS || p.hwr > MAX_HALO_WORDS || ht + hb > MAX_HALO_ROWS) return false;
    *out = p;
    return true;
}

// Does the planner's list equal one of the lists the strip kernel is instantiated for?
template <class LIST>
static bool strip_matches(const Plan &plan)
{
    if ((int)plan.groups.size() != LIST::NG || plan.hwl > 1 || plan.hwr > 1) return false;
    int ht = 0, hb = 0;
    for (int g = 0; g < LIST::NG; g++) {
        const Group &G = plan.groups[g];
        const WinC c = LIST::get(g);
        if (G.kind != c.kind) return false;
        ht += -c.dlo;
        hb += c.dhi;
        if (c.nsp == 0) {  // a (merged) RECT window
            if (G.n_spans != 1 || G.spans[0].rows != 0) return false;
            const Span &sp = G.spans[0];
            if (sp.lx != c.lx || sp.hx != c.hx || sp.dy_lo != c.dlo || sp.dy_hi != c.dhi) return false;
        } else {  // an element of row spans: the same spans on the same rows, in the planner's order (first appearance)
            if (G.n_spans != c.nsp) return false;
            for (int i = 0; i < c.nsp; i++) {
                const Span &sp = G.spans[i];
                if (sp.lx != c.sp[i].lx || sp.hx != c.sp[i].hx || sp.dy_lo != c.sp[i].dlo || sp.rows != c.sp[i].rows) return false;
                if (sp.dy_lo < c.dlo || sp.dy_hi > c.dhi) return false;
            }
        }
    }
    return ht == plan.ht && hb == plan.hb;  // the kernel's reach is the planner's (elements with empty border rows differ)
}

static int g_bs_seg_rows = -1, g_bs_mode = -1;  // IMGPROC_BITSTRIP: 0 never, 1 when the job is large (default), 2 whenever the list matches

template <class LIST>
static hipError_t launch_bitstrip(const Plan &plan, const uint32_t *d_tbits, u8 *d_masks, unsigned long long *d_mom, int w,
                                  int h, int planes, int second_order, hipStream_t stream)
{
    StripArgs a{};
    a.tbits = d_tbits;
    a.twords = (w + 31) / 32;
    a.masks = d_masks;
    a.mom = d_mom;
    a.w = w; a.h = h; a.planes = planes;
    a.hwl = plan.hwl;
    // several strips: a multiple of 4 words each, so that strips start on 128-byte lines
    a.valid_words = a.twords <= 64 - plan.hwl - plan.hwr ? a.twords : ((64 - plan.hwl - plan.hwr) & ~3);
    a.n_strips = (a.twords + a.valid_words - 1) / a.valid_words;
    // Segments: every segment re-runs the list's rows of reach, so they should be long; the chip wants a few thousand
    // waves.  ~4000 waves when the job has them, at least 1.5 x the reach and 48 rows, at most 512 (u32 sums per lane).
    const int reach = plan.ht + plan.hb;
    const long long cols = (long long)planes * a.n_strips;
    int seg = g_bs_seg_rows > 0 ? g_bs_seg_rows : (int)((long long)h * cols / 4096);
    if (g_bs_seg_rows <= 0 && seg < reach + reach / 2) seg = reach + reach / 2;
    seg = seg < (g_bs_seg_rows > 0 ? 8 : 48) ? (g_bs_seg_rows > 0 ? 8 : 48) : seg > 512 ? 512 : seg;
    a.n_segs = (h + seg - 1) / seg;
    a.seg_rows = (h + a.n_segs - 1) / a.n_segs;
    const long long items = cols * a.n_segs;
    if (items > 0x7FFFFFFFLL) return hipErrorInvalidValue;
    a.total_items = (int)items;
    const dim3 grid((unsigned)((items + 3) / 4));
    if (second_order)
        hipLaunchKernelGGL((k_bitstrip<LIST, true>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((k_bitstrip<LIST, false>), grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_bitmorph(const Plan &plan, const Group *d_groups, const u8 *d_thresh, u8 *d_masks,
                           unsigned long long *d_mom, int w, int h, int planes, int second_order, hipStream_t stream,
                           const uint32_t *d_tbits)
{
    if (planes <= 0 || w <= 0 || h <= 0) return hipSuccess;
    if (w >= (1 << 24) || h >= (1 << 24)) return hipErrorInvalidValue;  // 24-bit multiplies in the moment sums
    {  // read on every launch (cheap next to one): tests and A/B runs switch it inside a process
        const char *e = getenv("IMGPROC_BITSTRIP");
        g_bs_mode = e ? atoi(e) : 1;
        const char *r = getenv("IMGPROC_BITSTRIP_SEG_ROWS");
        g_bs_seg_rows = r ? atoi(r) : 0;
    }
    // the reference's two lists on bit planes, with enough independent columns to fill the chip: the strip kernel
    const long long strip_waves = (long long)planes * (((w + 31) / 32 + 61) / 62) * ((h + 63) / 64);
    if (d_tbits && g_bs_mode > 0 && (w & 3) == 0 && (g_bs_mode > 1 || strip_waves >= 512)) {
        if (strip_matches<ListPolygonPath>(plan))
            return launch_bitstrip<ListPolygonPath>(plan, d_tbits, d_masks, d_mom, w, h, planes, second_order, stream);
        if (strip_matches<ListPoseEstimate>(plan))
            return launch_bitstrip<ListPoseEstimate>(plan, d_tbits, d_masks, d_mom, w, h, planes, second_order, stream);
        if (strip_matches<ListOpenEllipse9>(plan))
            return launch_bitstrip<ListOpenEllipse9>(plan, d_tbits, d_masks, d_mom, w, h, planes, second_order, stream);
    }
    Args a{};
    a.thresh = d_tbits ? nullptr : d_thresh;
    a.tbits = d_tbits;
    a.twords = (w + 31) / 32;
    a.masks = d_masks;
    a.mom = d_mom;
    a.groups = d_groups;
    a.n_groups = (int)plan.groups.size();
    a.w = w; a.h = h; a.planes = planes;
    a.hwl = plan.hwl; a.hwr = plan.hwr; a.ht = plan.ht; a.hb = plan.hb;
    const int words = (w + 31) / 32;
    a.tiles_x = (words + TWW - 1) / TWW;
    a.tiles_y = (h + TH - 1) / TH;
    a.tww = (words + a.tiles_x - 1) / a.tiles_x;  // equal tiles: no mostly-empty last tile column / row
    a.th = (h + a.tiles_y - 1) / a.tiles_y;
    a.second_order = second_order;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * planes;
    if (blocks > 0x7FFFFFFFLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bitmorph, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace ipm
