// Planning and launch of the fused chain kernel (fused_kernel.hpp); the instantiations live in fused_inst_*.hip.
#include "fused_kernel.hpp"

#include <array>
#include <cmath>
#include <map>
#include <mutex>

namespace ipf {

// ---------------------------------------------------------------------------------------------
// planning / launch
// ---------------------------------------------------------------------------------------------
static bool plan_common(const imgproc_pipeline &pipe, const ipk::Bounds &bounds, FusedPlan &p)
{
    if (!(pipe.pre_blur_k <= 1 || pipe.pre_blur_k == 3)) return false;
    if (!(pipe.post_blur_k <= 1 || pipe.post_blur_k == 5)) return false;
    if (bounds.n < 1 || bounds.n > MAX_NR) return false;
    p.bidx = pipe.encoding == IMGPROC_ENC_RGB8 ? 2 : 0;
    p.bayer = pipe.encoding == IMGPROC_ENC_BAYER_RGGB8;
    p.second_order = pipe.second_order != 0;
    p.pre_k = pipe.pre_blur_k == 3 ? 3 : 1;
    p.post_k = pipe.post_blur_k == 5 ? 5 : 1;
    p.bounds = bounds;
    return true;
}

bool plan_fused(const imgproc_pipeline &pipe, const ipk::Element *elems, const ipk::Taps &, const ipk::Taps &,
                const ipk::Bounds &bounds, FusedPlan *out)
{
    FusedPlan p{};
    if (!plan_common(pipe, bounds, p) || pipe.n_morph > 2) return false;
    p.mode = 0;
    p.n_groups = pipe.n_morph;
    p.kind[0] = p.kind[1] = 2;
    for (int i = 0; i < pipe.n_morph; i++) {
        const ipk::Element &e = elems[i];
        if (!e.is_rect || e.kw != 3 || e.kh != 3 || e.ax != 1 || e.ay != 1) return false;
        p.kind[i] = pipe.morph[i].kind == IMGPROC_DILATE ? 1 : 0;
    }
    // one halo lane = 4 columns: blur reach (pre 1, post 2) + one column per 3x3 op must fit
    const int reach = (p.pre_k == 3 ? 1 : 0) + (p.post_k == 5 ? 2 : 0) + pipe.n_morph;
    if (reach > 4) return false;
    *out = p;
    return true;
}

bool plan_front(const imgproc_pipeline &pipe, const ipk::Bounds &bounds, FusedPlan *out)
{
    FusedPlan p{};
    if (!plan_common(pipe, bounds, p)) return false;
    p.mode = 1;
    p.kind[0] = p.kind[1] = 2;
    *out = p;
    return true;
}

bool can_launch_fused(const FusedPlan &plan, const void *d_frames, size_t frame_stride, size_t row_stride, int w, int h,
                      const void *d_masks)
{
    if (w < 4 || h < 1 || w > 16384 || h > 16384) return false;
    // w % 4 != 0 (tracker ROI crops, pose_estimate.cpp:197-202) runs with a right-aligned last strip: needs room for
    // two strips' unaligned-border zones, and an even width for the Bayer phase.  Dword accesses may then be unaligned,
    // which gfx950 global memory handles (scripts/ubench/unaligned.hip).
    if ((w & 3) && (w < 32 || (plan.bayer && (w & 1)))) return false;
    if (plan.bayer && h < 3) return false;  // no interior row: OpenCV writes zeros, the stage kernel does that
    if (plan.bayer && w < 8) return false;  // the lanes' 8-byte mosaic windows (fused_kernel.hpp, bayer_window_fix)
    // reflect_row() reflects once: exact BORDER_REFLECT_101 while the blurs' look-ahead (pre 1 + post 2 rows) stays below
    // h; shorter frames take the stage kernels
    if (h < (plan.pre_k == 3 ? 1 : 0) + (plan.post_k == 5 ? 2 : 0) + 1) return false;
    if ((unsigned long long)row_stride * (unsigned long long)h >= (1ull << 32)) return false;  // 32-bit in-frame offsets
    (void)d_frames; (void)frame_stride; (void)d_masks;
    return true;
}

static int g_seg_rows_override = -1;

static hipError_t dispatch_variant(const FusedPlan &plan, const Variant &v, const Args &a, dim3 grid, hipStream_t stream)
{
    if (plan.bayer) return v.cap <= 2 ? launch_variant_bayer_small(v, a, grid, stream) : launch_variant_bayer_large(v, a, grid, stream);
    return v.cap <= 2 ? launch_variant_plain_small(v, a, grid, stream) : launch_variant_plain_large(v, a, grid, stream);
}

// Wave slots of the current device for one instantiation (SIMDs x resident waves per SIMD), looked up once per
// instantiation and device.  0 when the runtime cannot tell.
static int wave_slots(const FusedPlan &plan, const Variant &v)
{
    static std::mutex mu;
    static std::map<std::array<int, 9>, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const std::array<int, 9> key = {dev, plan.bayer, v.pre_k, v.post_k, v.rgb, v.cap, v.second_order, v.mode, 0};
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int cus = 0, occ = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    Variant q = v;
    q.waves_per_simd = &occ;
    (void)dispatch_variant(plan, q, Args{}, dim3(1), nullptr);
    return cache[key] = cus * 4 * occ;
}

hipError_t launch_fused(const FusedPlan &plan, const u8 *d_frames, size_t frame_stride, size_t row_stride, int w,
                        int h, int n, u8 *d_masks, unsigned long long *d_mom, const int *d_tables,
                        hipStream_t stream, uint32_t *d_bits, bool overlapped)
{
    if (n <= 0) return hipSuccess;
    if (d_bits && (plan.mode != 1 || (w & 3))) return hipErrorInvalidValue;  // bit planes: front end only, strip borders on byte borders
    if (g_seg_rows_override < 0) {
        const char *e = getenv("IMGPROC_SEG_ROWS");
        g_seg_rows_override = e ? atoi(e) : 0;
    }
    Args a{};
    a.frames = d_frames;
    a.frame_stride = frame_stride;
    a.row_stride = row_stride;
    a.masks = d_masks;
    a.bits = d_bits;
    a.bit_words = (w + 31) / 32;
    a.mom = d_mom;
    a.tables = d_tables;
    a.w = w;
    a.h = h;
    a.n = n;
    const int sv = strip_valid(plan.bayer != 0);
    a.n_strips = (w + sv - 1) / sv;
    a.last_base = a.split_x = (a.n_strips - 1) * sv;
    if (w & 3) {
        a.n_strips = std::max(2, a.n_strips);
        a.last_base = w - sv;
        a.split_x = std::min((a.n_strips - 1) * sv, (w - 16) & ~3);
    }
    Variant v{};
    v.pre_k = plan.pre_k;
    v.post_k = plan.post_k;
    v.rgb = plan.bidx == 2;
    const int nr = plan.bounds.n;
    v.cap = nr <= 2 ? nr : (nr <= 4 ? 4 : 8);
    v.second_order = plan.second_order;
    v.mode = d_bits ? 2 : plan.mode;
    // ~100 rows per wave: measured optimum at steady clocks (1920x1200 x 256: 0.77 ms against 0.79 at 200, 0.80 at 67; the
    // fill / drain of a segment costs about as much as 9 rows, finer items even out the end of the launch)
    int target = g_seg_rows_override > 0 ? g_seg_rows_override : 100;
    if (g_seg_rows_override <= 0) {
        const long long items100 = (long long)a.n_strips * ((h + 99) / 100) * n;
        if (items100 < 1536) {
            // Small jobs (the live node: ONE frame per callback) are latency-bound: a 100-row segment is one wave's serial
            // work.  Shorter segments (more look-ahead rows re-done, a few % more work) until the chip has ~6 waves per CU.
            target = std::max(32, (int)((long long)h * a.n_strips * n / 1536));
        } else {
            // Large jobs: among segment heights of 90..130 rows take the one that leaves the last round of waves fullest
            // (waves = a whole multiple of the instantiation's wave slots: 12 x 100 rows at 6 waves per SIMD for the chain
            // kernel, 10 x 120 rows at 5 for the pre-blur front end on a 256-frame 1920x1200 batch; measured 2-4 %)
            const int slots = wave_slots(plan, v);
            if (slots > 0 && items100 >= slots) {
                double best = 1e9;
                for (int ns = (h + 129) / 130; ns <= (h + 89) / 90; ns++) {
                    const int rows = (h + ns - 1) / ns;
                    if (rows < 90 || rows > 130) continue;
                    const double rounds = (double)a.n_strips * ns * n / slots;
                    const double waste = std::ceil(rounds) - rounds;  // empty share of the last round
                    const double score = waste / std::ceil(rounds) + 0.0005 * std::abs(rows - 100);
                    if (score < best) best = score, target = rows;
                }
                // Two batches in flight on two streams: the other launch fills the slots this one's last waves leave, and
                // what counts is the look-ahead rows every segment re-does (8 of 108 at 100 rows).  Measured with two
                // 256-frame 1920x1200 batches in flight: 100 / 200 / 300 / 400 rows 0.696 / 0.696 / 0.672 / 0.670 ms per batch
                // (one stream: 0.704 / 0.719 / 0.722); 64 x 2448x2048 Bayer: 0.459 / 0.440 / 0.440 / 0.443.
                const int ns_tall = (h + 359) / 360;
                if (overlapped && 2ll * a.n_strips * ns_tall * n >= slots) target = (h + ns_tall - 1) / ns_tall;
            }
        }
    }
    a.n_segs = (h + target - 1) / target;
    a.seg_rows = (h + a.n_segs - 1) / a.n_segs;
    if (a.seg_rows > 512) {  // u32 moment partials are sized for <= 512 rows per segment
        a.seg_rows = 512;
        a.n_segs = (h + 511) / 512;
    }
    a.total_items = a.n_strips * a.n_segs * n;
    a.op0 = plan.kind[0];
    a.op1 = plan.kind[1];
    const int sh = plan.post_k == 5 ? 8 : 0;  // blurred fields are 8.8
    for (int r = 0; r < MAX_NR; r++) {  // ranges beyond nr (spare capacity of the 4 / 8 variants) never match
        a.hs_lo[r] = 0; a.hs_hi[r] = 0xFFFFFFFFu; a.v_lo[r] = 0xFFFFFFFFu; a.v_hi[r] = 0;
    }
    for (int r = 0; r < nr; r++) {
        const auto &B = plan.bounds;
        bool never = false;
        for (int c = 0; c < 3; c++) never |= B.lo[r][c] > B.hi[r][c];
        // 8.8 fields of the blurred image: the kernel leaves OpenCV's rounding constant (+128) out of the blur, so the
        // bounds move down by 128 instead: lo*256 <= x + 128 <= hi*256 + 255 (a negative lower bound is 0: x >= 0)
        auto lo16 = [&](int c) { const int v = ((int)B.lo[r][c] << sh) - (sh ? 128 : 0); return (uint32_t)(v < 0 ? 0 : v); };
        auto hi16 = [&](int c) { return (((uint32_t)B.hi[r][c] << sh) | (sh ? 0xFFu : 0u)) - (sh ? 128u : 0u); };
        if (never) {  // an empty channel interval empties the whole mask: make the V test unsatisfiable
            a.hs_lo[r] = 0;
            a.hs_hi[r] = 0xFFFFFFFFu;
            a.v_lo[r] = 0xFFFFFFFFu;  // no V field is ever 0xFFFF (8.8 fields stop at 255 * 256 + 128)
            a.v_hi[r] = 0;
        } else {
            a.hs_lo[r] = lo16(0) | (lo16(1) << 16);
            a.hs_hi[r] = hi16(0) | (hi16(1) << 16);
            a.v_lo[r] = lo16(2) | (lo16(2) << 16);
            a.v_hi[r] = hi16(2) - lo16(2);
        }
    }
    a.nr = nr;
    const dim3 grid((a.total_items + 3) / 4);
    return dispatch_variant(plan, v, a, grid, stream);
}

}  // namespace ipf
