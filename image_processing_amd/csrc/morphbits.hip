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

namespace ipm {

using ipk::u8;

constexpr int TWW = 16;  // interior tile width in 32-px words (512 px)
constexpr int TH = 96;   // interior tile height
constexpr int MAX_HALO_ROWS = 64;
constexpr int MAX_HALO_WORDS = 2;  // per side
constexpr int LW_MAX = TWW + 2 * MAX_HALO_WORDS;
constexpr int LH_MAX = TH + MAX_HALO_ROWS;

struct Args {
    const u8 *thresh;  // [planes][h][w]
    u8 *masks;         // [planes][h][w]
    unsigned long long *mom;  // [planes][6], added into
    const Group *groups;
    int n_groups;
    int w, h, planes;
    int hwl, hwr, ht, hb;  // halo: words left/right, rows top/bottom
    int tiles_x, tiles_y;
    int second_order;
};

// AND of the bits x+lo .. x+hi (|lo|,|hi| <= 32) for the 32 positions x of word `c`; p / n are the previous / next words.
__device__ __forceinline__ uint32_t window_and(uint32_t p, uint32_t c, uint32_t n, int lo, int hi)
{
    // 96-bit row segment, bit i of word k = position 32k+i; doubling: f_len(x) = AND of len bits starting at x
    uint32_t a0 = p, a1 = c, a2 = n;
    const int len = hi - lo + 1;
    int have = 1;
    auto shr96 = [](uint32_t &x0, uint32_t &x1, uint32_t &x2, int s) {  // positions move down by s (value at x+s -> x)
        while (s >= 32) {
            x0 = x1; x1 = x2; x2 = 0xFFFFFFFFu;
            s -= 32;
        }
        if (s) {
            x0 = (x0 >> s) | (x1 << (32 - s));
            x1 = (x1 >> s) | (x2 << (32 - s));
            x2 = (x2 >> s) | (0xFFFFFFFFu << (32 - s));
        }
    };
    while (have * 2 <= len) {
        uint32_t b0 = a0, b1 = a1, b2 = a2;
        shr96(b0, b1, b2, have);
        a0 &= b0; a1 &= b1; a2 &= b2;
        have *= 2;
    }
    if (have < len) {
        uint32_t b0 = a0, b1 = a1, b2 = a2;
        shr96(b0, b1, b2, len - have);
        a0 &= b0; a1 &= b1; a2 &= b2;
    }
    // now a(x) = AND of bits x .. x+len-1; we need it at x+lo: shift by 32+lo positions (lo in [-32, 32])
    const int s = 32 + lo;  // 0..64
    if (s >= 64) return a2;  // s == 64 (lo == 32)
    if (s >= 32) {
        const int t = s - 32;
        return t ? ((a1 >> t) | (a2 << (32 - t))) : a1;
    }
    return s ? ((a0 >> s) | (a1 << (32 - s))) : a0;
}

constexpr int N_WORDS_MAX = LH_MAX * LW_MAX;
constexpr int WORDS_PER_THREAD = (N_WORDS_MAX + 255) / 256;

__global__ __launch_bounds__(256) void k_bitmorph(const Args a)
{
    __shared__ uint32_t s_a[N_WORDS_MAX];
    __shared__ uint32_t s_b[N_WORDS_MAX];
    // bits outside the image (neutral for every op: OpenCV's morphologyDefaultBorderValue) = row part | column part
    __shared__ uint32_t s_rinv[LH_MAX], s_cinv[LW_MAX];
    __shared__ unsigned long long s_red[4][6];

    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y; t /= a.tiles_y;
    const int plane = t;
    const int w = a.w, h = a.h;
    const int LW = TWW + a.hwl + a.hwr, LH = TH + a.ht + a.hb;
    const uint32_t rcp = (65536u + LW - 1) / LW;  // idx / LW == (idx * rcp) >> 16 for idx < 3277 (LW <= 20)
    const int x_org = (tx * TWW - a.hwl) * 32;  // pixel x of tile word 0
    const int y_org = ty * TH - a.ht;
    const u8 *src = a.thresh + (size_t)plane * w * h;
    const bool vec = true;  // dword accesses may be unaligned (w % 4 != 0): fine for gfx950 global memory (scripts/ubench/unaligned.hip)
    auto mask_of = [](int kind) { return kind ? 0xFFFFFFFFu : 0u; };  // dilate(X) = ~erode(~X)

    if (tid < LH) s_rinv[tid] = (y_org + tid >= 0 && y_org + tid < h) ? 0u : 0xFFFFFFFFu;
    if (tid < LW) {
        const int x0 = x_org + tid * 32;
        uint32_t inv = 0xFFFFFFFFu;
        if (x0 < w && x0 + 32 > 0) {
            inv = 0;
            if (x0 < 0) inv |= (1u << (-x0)) - 1u;             // x0 in (-32, 0)
            if (x0 + 32 > w) inv |= ~((1u << (w - x0)) - 1u);  // partially beyond the right edge
        }
        s_cinv[tid] = inv;
    }
    // ---- pack: one 32-px word per thread-iteration, stored in the first group's erode domain ----
    const uint32_t km_first = mask_of(a.groups[0].kind);
    for (int idx = tid; idx < LH * LW; idx += 256) {
        const int r = (int)(((uint32_t)idx * rcp) >> 16), k = idx - r * LW;
        const int y = y_org + r, x0 = x_org + k * 32;
        uint32_t bits = 0, inv = 0xFFFFFFFFu;
        if (y >= 0 && y < h && x0 < w && x0 + 32 > 0) {
            inv = 0;
            if (x0 < 0) inv |= (1u << (-x0)) - 1u;
            if (x0 + 32 > w) inv |= ~((1u << (w - x0)) - 1u);
            const u8 *row = src + (size_t)y * w;
            if (vec && x0 >= 0 && x0 + 32 <= w) {
                const uint32_t *p = reinterpret_cast<const uint32_t *>(row + x0);
#pragma unroll
                for (int q = 0; q < 8; q++) bits |= (((p[q] & 0x01010101u) * 0x10204080u) >> 28) << (4 * q);
            } else {
                for (int i = 0; i < 32; i++)
                    if (x0 + i >= 0 && x0 + i < w && row[x0 + i]) bits |= 1u << i;
            }
        }
        s_a[idx] = (bits ^ km_first) | inv;
    }
    __syncthreads();

    // Every group: for each distinct span of its element, a horizontal window AND of the whole tile (cur -> alt), then
    // the AND over the rows that use that span, accumulated in registers; the result goes back into cur already in the
    // NEXT group's erode domain.  Rows a group can no longer produce correctly (reach of the groups so far) are skipped.
    uint32_t *const cur = s_a, *const alt = s_b;
    for (int gi = 0; gi < a.n_groups; gi++) {
        const Group &g = a.groups[gi];
        const bool last = gi + 1 == a.n_groups;
        const uint32_t flip = mask_of(g.kind) ^ (last ? 0u : mask_of(a.groups[gi + 1].kind));
        const int h0 = g.t_in * LW, h1 = (LH - g.b_in) * LW;
        const int v0 = g.t_out * LW, v1 = (LH - g.b_out) * LW;
        uint32_t acc[WORDS_PER_THREAD];
#pragma unroll
        for (int i = 0; i < WORDS_PER_THREAD; i++) acc[i] = 0xFFFFFFFFu;
        for (int si = 0; si < g.n_spans; si++) {
            const Span sp = g.spans[si];
            const uint32_t *hsrc = cur;
            if (sp.lx != 0 || sp.hx != 0) {
                for (int idx = h0 + tid; idx < h1; idx += 256) {
                    const int r = (int)(((uint32_t)idx * rcp) >> 16), k = idx - r * LW;
                    const uint32_t p = k > 0 ? cur[idx - 1] : 0xFFFFFFFFu;
                    const uint32_t n = k < LW - 1 ? cur[idx + 1] : 0xFFFFFFFFu;
                    alt[idx] = window_and(p, cur[idx], n, sp.lx, sp.hx);
                }
                __syncthreads();
                hsrc = alt;
            }
#pragma unroll
            for (int i = 0; i < WORDS_PER_THREAD; i++) {
                const int idx = v0 + tid + 256 * i;
                if (idx < v1) {
                    uint32_t x = acc[i];
                    const uint32_t *col = hsrc + idx + sp.dy_lo * LW;
                    if (sp.rows == 0) {
                        for (int d = sp.dy_hi - sp.dy_lo; d >= 0; d--) x &= col[d * LW];
                    } else {
                        for (uint32_t m = sp.rows; m; m &= m - 1) x &= col[(__ffs(m) - 1) * LW];
                    }
                    acc[i] = x;
                }
            }
            if (hsrc == alt && si + 1 < g.n_spans) __syncthreads();  // alt is rewritten by the next span
        }
        __syncthreads();  // every read of cur is done
#pragma unroll
        for (int i = 0; i < WORDS_PER_THREAD; i++) {
            const int idx = v0 + tid + 256 * i;
            if (idx < v1) {
                const int r = (int)(((uint32_t)idx * rcp) >> 16), k = idx - r * LW;
                cur[idx] = last ? (acc[i] ^ flip) : ((acc[i] ^ flip) | s_rinv[r] | s_cinv[k]);
            }
        }
        __syncthreads();
    }

    // ---- interior: bytes out + moments ----
    unsigned long long m[6] = {0, 0, 0, 0, 0, 0};
    u8 *dst = a.masks ? a.masks + (size_t)plane * w * h : nullptr;
    for (int idx = tid; idx < TH * TWW; idx += 256) {
        const int r = idx / TWW, k = idx - r * TWW;
        const int y = ty * TH + r, x0 = (tx * TWW + k) * 32;
        if (y >= h || x0 >= w) continue;
        uint32_t bits = cur[(r + a.ht) * LW + (k + a.hwl)];
        if (x0 + 32 > w) bits &= (1u << (w - x0)) - 1u;
        if (dst) {
            u8 *o = dst + (size_t)y * w + x0;
            if (vec && x0 + 32 <= w) {
                uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint32_t nib = (bits >> (4 * q)) & 15u;
                    o32[q] = __builtin_amdgcn_perm(0u, 0x0000FF00u, (nib * 0x00204081u) & 0x01010101u);
                }
            } else {
                for (int i = 0; i < 32 && x0 + i < w; i++) o[i] = ((bits >> i) & 1u) ? 255 : 0;
            }
        }
        if (bits) {
            const unsigned long long cnt = __popc(bits);
            const unsigned long long bs = __popc(bits & 0xAAAAAAAAu) + 2 * __popc(bits & 0xCCCCCCCCu) + 4 * __popc(bits & 0xF0F0F0F0u) +
                                          8 * __popc(bits & 0xFF00FF00u) + 16 * __popc(bits & 0xFFFF0000u);  // sum of bit indices
            const unsigned long long sx = (unsigned long long)x0 * cnt + bs;
            m[0] += cnt;
            m[1] += sx;
            m[2] += (unsigned long long)y * cnt;
            if (a.second_order) {
                // sum of squared bit indices i = sum_j 2^j b_j(i): i^2 = sum_j 4^j b_j + sum_{j<k} 2^(j+k+1) b_j b_k
                const uint32_t B[5] = {0xAAAAAAAAu, 0xCCCCCCCCu, 0xF0F0F0F0u, 0xFF00FF00u, 0xFFFF0000u};
                unsigned long long bs2 = 0;
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    bs2 += (unsigned long long)__popc(bits & B[j]) << (2 * j);
#pragma unroll
                    for (int k = j + 1; k < 5; k++) bs2 += (unsigned long long)__popc(bits & B[j] & B[k]) << (j + k + 1);
                }
                m[3] += (unsigned long long)x0 * x0 * cnt + 2ull * x0 * bs + bs2;
                m[4] += (unsigned long long)y * sx;
                m[5] += (unsigned long long)y * y * cnt;
            }
        }
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

hipError_t launch_bitmorph(const Plan &plan, const Group *d_groups, const u8 *d_thresh, u8 *d_masks,
                           unsigned long long *d_mom, int w, int h, int planes, int second_order, hipStream_t stream)
{
    if (planes <= 0 || w <= 0 || h <= 0) return hipSuccess;
    Args a{};
    a.thresh = d_thresh;
    a.masks = d_masks;
    a.mom = d_mom;
    a.groups = d_groups;
    a.n_groups = (int)plan.groups.size();
    a.w = w; a.h = h; a.planes = planes;
    a.hwl = plan.hwl; a.hwr = plan.hwr; a.ht = plan.ht; a.hb = plan.hb;
    a.tiles_x = (w + TWW * 32 - 1) / (TWW * 32);
    a.tiles_y = (h + TH - 1) / TH;
    a.second_order = second_order;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * planes;
    if (blocks > 0x7FFFFFFFLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bitmorph, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace ipm
