// C ABI of libimgproc_hip (include/imgproc.h): context, pipeline plan, streaming slots, batch
// paths and single-stage entry points.  Host-side only; the kernels are in kernels.hip / fused.hip.
// No CPU fallback anywhere: if HIP fails, the call fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/imgproc.h"
#include "fused.h"
#include "kernels.h"
#include "morphbits.h"

#define IMGPROC_API extern "C" __attribute__((visibility("default")))

using ipk::u8;

// ---- imgproc_process_batch_device_noalloc: the scratch of the one entry point that allocates nothing ----------------
// Layout: [0, NA_MOM_BYTES) moment sums u64[frames * ranges][6]; [NA_MOM_BYTES, end) inRange planes handed from the front
// kernel to the bit-plane kernel.  One copy per GPU, held until the process ends; every call on a GPU shares it.
constexpr size_t NA_MOM_ENTRIES = 8192;                                   // frames x ranges per call
constexpr size_t NA_MOM_BYTES = NA_MOM_ENTRIES * 6 * sizeof(unsigned long long);  // 393 216, a multiple of 256
constexpr size_t NA_PLANE_BYTES = 32u << 20;                              // inRange planes: frames x ranges x w x h bytes
constexpr size_t NA_SCRATCH_BYTES = NA_MOM_BYTES + NA_PLANE_BYTES;
static __device__ __align__(256) unsigned char scratch[NA_SCRATCH_BYTES];
// `scratch` is a host-side stand-in in host code, not a device address: the per-device address is looked up once per
// device (imgproc_create, with that device current) and only this pointer is ever used for kernel arguments / memsets / copies.
constexpr int NA_MAX_DEVICES = 64;
static unsigned char *g_scratch_ptr[NA_MAX_DEVICES];
static std::mutex g_scratch_mu;
// one no-allocation call at a time per GPU: the call is synchronous, so the flag is held from its first use of the array to
// after its stream synchronize; a second context / thread on the same GPU gets IMGPROC_ESTATE instead of corrupted sums
static std::atomic<bool> g_scratch_busy[NA_MAX_DEVICES];

namespace {

struct Slot {
    u8 *h_in = nullptr;  // pinned staging (tight rows)
    u8 *d_raw = nullptr, *d_a = nullptr, *d_b = nullptr, *d_bgr = nullptr, *d_t = nullptr;
    u8 *d_thresh = nullptr, *d_mask = nullptr, *d_m0 = nullptr, *d_m1 = nullptr, *d_masked = nullptr;
    unsigned long long *d_mom = nullptr, *h_mom = nullptr;
    hipEvent_t ev_up = nullptr, ev_done = nullptr;
    hipStream_t stream = nullptr;          // every slot runs its frame on its own stream: slot k+1's upload overlaps slot k's kernels
    hipGraphExec_t graph_exec = nullptr;   // captured {H2D, memset, chain kernels, D2H of moments} for (w, h, plan epoch)
    int graph_w = 0, graph_h = 0;
    unsigned graph_epoch = 0;
    int w = 0, h = 0;
    bool in_flight = false, has_frame = false;
    const u8 *blur_ptr = nullptr, *hsv_ptr = nullptr, *bgr_ptr = nullptr;  // where each kept stage lives
    size_t bgr_stride = 0;
    int valid_upto = -1;        // highest of IMGPROC_STAGE_BGR..THRESH that is materialised for the frame in the slot
    int masked_range = -1;      // d_masked holds frame & mask of this range (-1: not built)
    unsigned frame_epoch = 0;   // plan epoch the frame was submitted under
    int nr = 0;                 // ranges / second-order flag the frame in flight was submitted with: imgproc_wait decodes
    bool second = false;        // h_mom with these, not with whatever the plan says by then
};

// Staging of a submitted frame into the slot's pinned buffer by several threads: the frame is cut into row chunks that
// the caller's thread and a few helper threads take from a shared counter, while the caller hands every finished chunk
// to the DMA engine in order.  One core moves ~29 GB/s, which made that copy -- not PCIe, not the kernel -- the ceiling
// of the streaming path; the caller starts on chunk 0 at once, so the first DMA does not wait for a thread to wake up.
struct CopyPool {
    static constexpr int MAX_CHUNKS = 32;
    struct Frame {  // the frame being staged
        u8 *dst = nullptr;
        const u8 *src = nullptr;
        size_t rowb = 0, stride = 0;
        int h = 0, n_chunks = 0;
    };
    static void chunk_rows(const Frame &f, int ci, int *y0, int *y1)
    {
        *y0 = (int)((long long)f.h * ci / f.n_chunks);
        *y1 = (int)((long long)f.h * (ci + 1) / f.n_chunks);
    }
    static void copy_chunk(const Frame &f, int ci)
    {
        int y0, y1;
        chunk_rows(f, ci, &y0, &y1);
        if (f.stride == f.rowb)
            memcpy(f.dst + (size_t)y0 * f.rowb, f.src + (size_t)y0 * f.stride, f.rowb * (size_t)(y1 - y0));
        else
            for (int y = y0; y < y1; y++) memcpy(f.dst + (size_t)y * f.rowb, f.src + (size_t)y * f.stride, f.rowb);
    }
    std::vector<std::thread> threads;
    Frame frame;
    std::atomic<unsigned long long> next{0};       // (generation << 32) | next chunk to take
    std::atomic<unsigned> chunk_done[MAX_CHUNKS];  // generation in which chunk ci was last completed
    std::mutex mu;
    std::condition_variable cv;
    unsigned gen = 0;
    bool stop = false;

    explicit CopyPool(int n)
    {
        for (auto &d : chunk_done) d.store(0);
        try {
            for (int i = 0; i < n; i++) threads.emplace_back([this] { run(); });
        } catch (...) {  // a thread could not be started: end the ones that were (a joinable std::thread must not be destroyed)
            shut_down();
            throw;
        }
    }
    ~CopyPool() { shut_down(); }
    void shut_down()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : threads) t.join();
        threads.clear();
    }
    int helpers() const { return (int)threads.size(); }
    // Takes the next chunk of generation g, or -1 when none is left (or g is no longer the current frame: a helper
    // that wakes up late must not touch a newer frame with the old frame's pointers).
    int take(unsigned g, int n_chunks)
    {
        unsigned long long v = next.load(std::memory_order_acquire);
        for (;;) {
            if ((unsigned)(v >> 32) != g || (int)(v & 0xFFFFFFFFu) >= n_chunks) return -1;
            if (next.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) return (int)(v & 0xFFFFFFFFu);
        }
    }
    void run()
    {
        unsigned seen = 0;
        for (;;) {
            Frame f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                f = frame;
            }
            for (int ci; (ci = take(seen, f.n_chunks)) >= 0;) {
                copy_chunk(f, ci);
                chunk_done[ci].store(seen, std::memory_order_release);
            }
        }
    }
    void start(const Frame &f)  // every chunk of the previous frame is done (stage() returns only then)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            frame = f;
            gen++;
            next.store((unsigned long long)gen << 32, std::memory_order_release);
        }
        cv.notify_all();
    }
    bool done(int ci) const { return chunk_done[ci].load(std::memory_order_acquire) == gen; }
    // Caller's side: copy chunks too, and call issue(ci) for every chunk, in order, as soon as it is complete.  Returns
    // only when all chunks are complete: nobody reads f.src afterwards.
    template <typename Issue>
    void stage(const Frame &f, Issue issue)
    {
        start(f);
        int issued = 0;
        while (issued < f.n_chunks) {
            if (done(issued)) {
                issue(issued++);
                continue;
            }
            const int ci = take(gen, f.n_chunks);
            if (ci >= 0) {
                copy_chunk(f, ci);
                chunk_done[ci].store(gen, std::memory_order_release);
            } else {
                std::this_thread::yield();
            }
        }
    }
};

struct Plan {
    imgproc_pipeline pipe{};
    std::vector<std::vector<u8>> custom;  // deep copies of custom elements
    ipk::Taps pre{}, post{};
    ipk::Element elems[IMGPROC_MAX_MORPH];
    ipk::Bounds bounds{};
    ipf::FusedPlan fused{};
    bool fused_ok = false;
    ipm::Plan bm;        // the morph list on bit planes (morphbits.hip)
    bool bm_ok = false;
    ipf::FusedPlan front{};  // [blur3] HSV [blur5] inRange in one kernel, bytes out; bm finishes (two launches per batch)
    bool front_ok = false;
};

}  // namespace

struct imgproc_ctx {
    imgproc_config cfg{};
    hipStream_t stream = nullptr;
    int *d_tables = nullptr;
    ipm::Group *d_groups = nullptr;  // device copy of plan.bm.groups (32 entries)
    std::vector<Slot> slots;
    Slot work;  // scratch planes of the stage-by-stage device-batch path and imgproc_mask_and (ensure_work: on first use)
    bool work_ready = false;
    Plan plan;
    struct Batch {  // one of the two in-flight device batches (imgproc_batch_enqueue / imgproc_batch_wait)
        unsigned long long *d_mom = nullptr, *h_mom = nullptr;
        hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_done = nullptr;
        bool pending = false, timed = false;
        int n = 0, nr = 0, launches = 0, second = 0;
    } batch[IMGPROC_MAX_TICKETS];
    int last_ticket = -1;  // the batch imgproc_last_kernel_ms reports on
    // inRange planes of a device batch for the front + bit-plane path (grown on demand), one buffer per ticket
    u8 *d_thresh_batch[IMGPROC_MAX_TICKETS] = {};
    size_t thresh_batch_bytes[IMGPROC_MAX_TICKETS] = {};
    // tickets 1.. of the one- and two-launch batch paths run on streams of their own (IMGPROC_BATCH_STREAMS=1: all on `stream`)
    hipStream_t tstream[IMGPROC_MAX_TICKETS] = {};
    hipEvent_t ev_ref = nullptr;    // time zero of imgproc_last_kernel_span (re-recorded by imgproc_set_profiling)
    unsigned char *na_scratch = nullptr;    // this device's copy of `scratch` (hipGetSymbolAddress at create)
    unsigned long long *h_na_mom = nullptr;  // pinned landing buffer of the no-allocation entry point (max_batch frames)
    u8 *d_s0 = nullptr, *d_s1 = nullptr;  // scratch for single-stage calls (max frame, 3 ch)
    unsigned long long *d_s_mom = nullptr;
    bool profiling = false;
    size_t copy_chunk_bytes = 768u << 10;  // IMGPROC_COPY_CHUNK_KB
    bool copy_always = false;              // IMGPROC_COPY_ALWAYS=1: helper threads even for a lone frame
    CopyPool *copy_pool = nullptr;  // null: the caller's thread copies alone (IMGPROC_COPY_THREADS=0, or threads unavailable)
    unsigned plan_epoch = 1;  // bumped by set_pipeline / set_ranges: invalidates the slots' captured graphs
    int graph_mode = 0;  // IMGPROC_GRAPH: 0 direct launches (default, measured faster), 1 replay a captured hipGraph per frame, 2 graph only for the stage-by-stage chain
    char err[512] = {0};
};

namespace {

int fail(imgproc_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof c->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIPCHK(c, expr)                                                                                     \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail((c), e_ == hipErrorOutOfMemory ? IMGPROC_ENOMEM : IMGPROC_EHIP, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                         \
    } while (0)

// --- host-side restatements the plan needs -------------------------
int gaussian_taps(int k, int *taps)
{
    if (k < 1 || !(k & 1) || k > IMGPROC_MAX_KSIZE) return IMGPROC_EINVAL;
    switch (k) {  // OpenCV's fixed table for sigma<=0, k<=7
    case 1: taps[0] = 256; return 0;
    case 3: { const int t[] = {64, 128, 64}; memcpy(taps, t, sizeof t); return 0; }
    case 5: { const int t[] = {16, 64, 96, 64, 16}; memcpy(taps, t, sizeof t); return 0; }
    case 7: { const int t[] = {8, 28, 56, 72, 56, 28, 8}; memcpy(taps, t, sizeof t); return 0; }
    default: break;
    }
    const long double sigma = ((k - 1) * 0.5L - 1.0L) * 0.3L + 0.8L;
    long double g[IMGPROC_MAX_KSIZE];
    long double tot = 0;
    for (int i = 0; i < k; i++) {
        long double x = i - (k - 1) * 0.5L;
        tot += g[i] = expl(-0.5L * x * x / (sigma * sigma));
    }
    long double carry = 0;  // error diffusion toward the centre tap
    int used = 0;
    for (int i = 0; i < k / 2; i++) {
        long double want = g[i] / tot * 256.0L + carry;
        long long q = llrintl(want);
        carry = want - (long double)q;
        taps[i] = taps[k - 1 - i] = (int)q;
        used += 2 * (int)q;
    }
    taps[k / 2] = 256 - used;
    return 0;
}

int make_element(const imgproc_morph &m, const u8 *custom, ipk::Element &e)
{
    if (m.kw < 1 || m.kh < 1 || m.kw > IMGPROC_MAX_KSIZE || m.kh > IMGPROC_MAX_KSIZE) return IMGPROC_EINVAL;
    if (m.kind != IMGPROC_ERODE && m.kind != IMGPROC_DILATE) return IMGPROC_EINVAL;
    e.kw = m.kw;
    e.kh = m.kh;
    e.ax = m.anchor_x < 0 ? m.kw / 2 : m.anchor_x;
    e.ay = m.anchor_y < 0 ? m.kh / 2 : m.anchor_y;
    if (e.ax >= m.kw || e.ay >= m.kh) return IMGPROC_EINVAL;
    u8 bits[IMGPROC_MAX_KSIZE * IMGPROC_MAX_KSIZE];
    if (m.shape == IMGPROC_CUSTOM) {
        if (!custom) return IMGPROC_EINVAL;
        memcpy(bits, custom, (size_t)m.kw * m.kh);
    } else {
        int rc = imgproc_structuring_element(m.shape, m.kw, m.kh, bits);
        if (rc) return rc;
    }
    bool all = true;
    for (int j = 0; j < m.kh; j++) {
        uint32_t r = 0;
        for (int i = 0; i < m.kw; i++)
            if (bits[(size_t)j * m.kw + i]) r |= 1u << i;
            else all = false;
        e.rows[j] = r;
    }
    for (int j = m.kh; j < 31; j++) e.rows[j] = 0;
    e.is_rect = all ? 1 : 0;
    return 0;
}

void make_bounds(int n, const int32_t (*lo)[3], const int32_t (*hi)[3], ipk::Bounds &b)
{
    b.n = n;
    for (int r = 0; r < n; r++)
        for (int c = 0; c < 3; c++) {
            long l = lo[r][c], u = hi[r][c];
            if (l > u || l > 255 || u < 0) {
                b.lo[r][c] = 1;  // empty interval: never matches
                b.hi[r][c] = 0;
            } else {
                b.lo[r][c] = (u8)(l < 0 ? 0 : l);
                b.hi[r][c] = (u8)(u > 255 ? 255 : u);
            }
        }
}

void finish_moments(const unsigned long long *m, imgproc_moments &o, bool second = true)
{
    o.m00 = m[0]; o.m10 = m[1]; o.m01 = m[2];
    o.m20 = second ? m[3] : 0; o.m11 = second ? m[4] : 0; o.m02 = second ? m[5] : 0;
    o.found = m[0] != 0;
    o.reserved = 0;
    o.theta = 0.0;
    if (o.found) {
        o.cx = (double)m[1] / (double)m[0];
        o.cy = (double)m[2] / (double)m[0];
        if (!second) return;
        long double n = (long double)m[0];
        long double mu20 = (long double)m[3] - (long double)m[1] * (long double)m[1] / n;
        long double mu02 = (long double)m[5] - (long double)m[2] * (long double)m[2] / n;
        long double mu11 = (long double)m[4] - (long double)m[1] * (long double)m[2] / n;
        o.theta = 0.5 * std::atan2((double)(2 * mu11), (double)(mu20 - mu02));
    } else {
        o.cx = o.cy = -1.0;
        o.theta = 0.0;
    }
}

void free_slot(Slot &s)
{
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_mom) (void)hipHostFree(s.h_mom);
    u8 *bufs[] = {s.d_raw, s.d_a, s.d_b, s.d_bgr, s.d_t, s.d_thresh, s.d_mask, s.d_m0, s.d_m1, s.d_masked};
    for (u8 *p : bufs)
        if (p) (void)hipFree(p);
    if (s.d_mom) (void)hipFree(s.d_mom);
    if (s.ev_up) (void)hipEventDestroy(s.ev_up);
    if (s.ev_done) (void)hipEventDestroy(s.ev_done);
    if (s.graph_exec) (void)hipGraphExecDestroy(s.graph_exec);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    s = Slot();
}

// staging = false: the context's own work slot (device batches, single-stage calls): no pinned frame buffer
int alloc_slot(imgproc_ctx *c, Slot &s, bool staging = true)
{
    const size_t px = (size_t)c->cfg.max_width * c->cfg.max_height;
    if (staging) HIPCHK(c, hipHostMalloc((void **)&s.h_in, px * 3, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&s.h_mom, sizeof(unsigned long long) * 6 * IMGPROC_MAX_RANGES, hipHostMallocDefault));
    memset(s.h_mom, 0, sizeof(unsigned long long) * 6 * IMGPROC_MAX_RANGES);
    if (staging) HIPCHK(c, hipMalloc((void **)&s.d_raw, px * 3));  // the work slot reads the caller's device frames in place
    HIPCHK(c, hipMalloc((void **)&s.d_a, px * 3));
    HIPCHK(c, hipMalloc((void **)&s.d_b, px * 3));
    HIPCHK(c, hipMalloc((void **)&s.d_bgr, px * 3));
    HIPCHK(c, hipMalloc((void **)&s.d_t, px * 3));
    if (staging) HIPCHK(c, hipMalloc((void **)&s.d_masked, px * 3));
    // inRange planes: bytes (px per range) or bit planes (h rows of ceil(w / 32) words per range) -- for frames narrower
    // than 4 px the bit planes are the larger of the two
    const size_t bit_plane = (size_t)c->cfg.max_height * 4 * (((size_t)c->cfg.max_width + 31) / 32);
    HIPCHK(c, hipMalloc((void **)&s.d_thresh, std::max(px, bit_plane) * IMGPROC_MAX_RANGES));
    HIPCHK(c, hipMalloc((void **)&s.d_mask, px * IMGPROC_MAX_RANGES));
    HIPCHK(c, hipMalloc((void **)&s.d_m0, px));
    HIPCHK(c, hipMalloc((void **)&s.d_m1, px));
    HIPCHK(c, hipMalloc((void **)&s.d_mom, sizeof(unsigned long long) * 6 * IMGPROC_MAX_RANGES));
    HIPCHK(c, hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    HIPCHK(c, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    return 0;
}

// Stage-by-stage chain for one frame whose raw bytes are at d_frame (device).  Final masks land in
// d_masks_out (n_ranges planes of w*h, tight), moments are ADDED into d_mom (caller zeroes).
// `upto` < IMGPROC_STAGE_MASK stops after that stage (imgproc_fetch_stage rebuilding a stage image on demand).
// `bits_out` (only when bit_plane_tail(plan) holds): the frame's inRange bit planes go there and the morphology is left
// to the caller, who runs it for several frames in one launch.
static bool bit_plane_tail(const Plan &p) { return p.bm_ok && !p.pipe.force_unfused && !p.pipe.keep_stages; }

// Long post-HSV Gaussian (k = 7..15) + bit-plane tail: [HSV,] blur and inRange run as ONE kernel (gauss_fused.hip), on the
// BGR / RGB image (`hsv_in`) or on an HSV image a stage kernel produced.  IMGPROC_GAUSS_FUSED=0 keeps the three stage kernels.
static bool gauss_fused_wanted(const Plan &p)
{
    static const bool off = [] { const char *e = getenv("IMGPROC_GAUSS_FUSED"); return e && e[0] == '0'; }();
    return !off && bit_plane_tail(p) && p.pipe.post_blur_k >= 7 && p.pipe.post_blur_k <= 15;
}

int run_unfused(imgproc_ctx *c, Slot &s, const u8 *d_frame, size_t row_stride, int w, int h, u8 *d_masks_out,
                unsigned long long *d_mom, int *launches, hipStream_t st, int upto = IMGPROC_STAGE_MASK,
                uint32_t *bits_out = nullptr)
{
    const Plan &p = c->plan;
    const size_t t3 = (size_t)w * 3, px = (size_t)w * h;
    const u8 *cur = d_frame;
    size_t cur_stride = row_stride;
    int bidx = p.pipe.encoding == IMGPROC_ENC_RGB8 ? 2 : 0;
    int nl = 0;
    // buffers: d_bgr = BGR stage, d_a = HSV, d_b = blurred HSV, d_t = temporary (demosaic before pre-blur)
    if (p.pipe.encoding == IMGPROC_ENC_BAYER_RGGB8) {
        u8 *dst = p.pipe.pre_blur_k > 1 ? s.d_t : s.d_bgr;
        HIPCHK(c, ipk::launch_demosaic_rggb(cur, cur_stride, dst, t3, w, h, st));
        nl++;
        cur = dst;
        cur_stride = t3;
    }
    if (p.pipe.pre_blur_k > 1) {
        HIPCHK(c, ipk::launch_gaussian(cur, cur_stride, s.d_bgr, t3, w, h, 3, p.pre, st));
        nl++;
        cur = s.d_bgr;
        cur_stride = t3;
    }
    s.bgr_ptr = cur;  // may be the caller's frame itself (then the row stride is the caller's)
    s.bgr_stride = cur_stride;
    if (upto == IMGPROC_STAGE_BGR) return 0;
    if (upto > IMGPROC_STAGE_THRESH && gauss_fused_wanted(p) && ipk::gauss_fused_ok(cur, 0, cur_stride, w, h, p.post, p.bounds)) {
        // HSV + long Gaussian + inRange -> bit planes in one kernel, then (unless the caller collects planes of several
        // frames) the whole morph list + moments + final masks in one more
        uint32_t *bits = bits_out ? bits_out : reinterpret_cast<uint32_t *>(s.d_thresh);
        HIPCHK(c, ipk::launch_gauss_fused(cur, 0, cur_stride, bits, w, h, 1, p.post, p.bounds, bidx, c->d_tables, st));
        nl++;
        if (!bits_out) {
            HIPCHK(c, ipm::launch_bitmorph(p.bm, c->d_groups, nullptr, d_masks_out, d_mom, w, h, p.bounds.n, p.pipe.second_order, st, bits));
            nl++;
        }
        if (launches) *launches += nl;
        return 0;
    }
    HIPCHK(c, ipk::launch_bgr2hsv(cur, cur_stride, s.d_a, t3, w, h, bidx, c->d_tables, st));
    nl++;
    s.hsv_ptr = s.d_a;
    if (upto == IMGPROC_STAGE_HSV) return 0;
    const u8 *blur = s.d_a;
    if (p.pipe.post_blur_k > 1) {
        HIPCHK(c, ipk::launch_gaussian(s.d_a, t3, s.d_b, t3, w, h, 3, p.post, st));
        nl++;
        blur = s.d_b;
    }
    s.blur_ptr = blur;
    if (upto == IMGPROC_STAGE_BLUR) return 0;
    const int nr = p.bounds.n;
    const bool to_bitmorph = p.bm_ok && !p.pipe.force_unfused;
    if (bit_plane_tail(p) && upto > IMGPROC_STAGE_THRESH) {
        // nobody asked for the inRange bytes: the planes go to the bit-plane morphology as bits (1/8 B per px and range),
        // then the whole morph list + moments + final masks for all ranges in one launch
        uint32_t *bits = bits_out ? bits_out : reinterpret_cast<uint32_t *>(s.d_thresh);
        HIPCHK(c, ipk::launch_in_range_bits(blur, t3, bits, w, h, p.bounds, st));
        if (bits_out) {
            if (launches) *launches += nl + 1;
            return 0;
        }
        HIPCHK(c, ipm::launch_bitmorph(p.bm, c->d_groups, nullptr, d_masks_out, d_mom, w, h, nr, p.pipe.second_order, st, bits));
        if (launches) *launches += nl + 2;
        return 0;
    }
    HIPCHK(c, ipk::launch_in_range(blur, t3, s.d_thresh, (size_t)w, px, w, h, p.bounds, st));
    nl++;
    if (upto == IMGPROC_STAGE_THRESH) return 0;
    if (to_bitmorph) {
        HIPCHK(c, ipm::launch_bitmorph(p.bm, c->d_groups, s.d_thresh, d_masks_out, d_mom, w, h, nr, p.pipe.second_order, st));
        nl++;
    } else {
        for (int r = 0; r < nr; r++) {
            const u8 *m = s.d_thresh + (size_t)r * px;
            u8 *final_dst = d_masks_out + (size_t)r * px;
            for (int i = 0; i < p.pipe.n_morph; i++) {
                u8 *dst = (i == p.pipe.n_morph - 1) ? final_dst : ((m == s.d_m0) ? s.d_m1 : s.d_m0);
                HIPCHK(c, ipk::launch_morph(m, (size_t)w, dst, (size_t)w, w, h, p.pipe.morph[i].kind, p.elems[i], st));
                nl++;
                m = dst;
            }
            if (p.pipe.n_morph == 0) {
                HIPCHK(c, hipMemcpyAsync(final_dst, m, px, hipMemcpyDeviceToDevice, st));
            }
            HIPCHK(c, ipk::launch_moments(final_dst, (size_t)w, w, h, d_mom + 6 * (size_t)r, st));
            nl++;
        }
    }
    if (launches) *launches += nl;
    return 0;
}

// The context's work slot (planes of the stage-by-stage device-batch path and of imgproc_mask_and) is allocated on its
// first use: ~30 B/px that callers of the streaming path, the chain kernel or the front + bit-plane path never pay.
int ensure_work(imgproc_ctx *c)
{
    if (c->work_ready) return 0;
    int rc = alloc_slot(c, c->work, false);
    if (rc) {
        free_slot(c->work);
        return rc;
    }
    c->work_ready = true;
    return 0;
}

int check_frame(imgproc_ctx *c, int w, int h, size_t row_stride)
{
    if (w < 1 || h < 1) return fail(c, IMGPROC_EINVAL, "frame %dx%d", w, h);
    if (w > c->cfg.max_width || h > c->cfg.max_height || (size_t)w * h > (size_t)c->cfg.max_width * c->cfg.max_height)
        return fail(c, IMGPROC_ESIZE, "frame %dx%d exceeds context maximum %dx%d", w, h, c->cfg.max_width, c->cfg.max_height);
    const size_t need = (size_t)w * (c->plan.pipe.encoding == IMGPROC_ENC_BAYER_RGGB8 ? 1 : 3);
    if (row_stride < need) return fail(c, IMGPROC_EINVAL, "row stride %zu < %zu", row_stride, need);
    return 0;
}

}  // namespace

// ================================================================================================
// lifecycle
// ================================================================================================
IMGPROC_API int imgproc_version(void) { return IMGPROC_VERSION; }

IMGPROC_API const char *imgproc_strerror(int s)
{
    switch (s) {
    case IMGPROC_OK: return "ok";
    case IMGPROC_EINVAL: return "invalid argument";
    case IMGPROC_ENOMEM: return "out of memory";
    case IMGPROC_EHIP: return "HIP runtime error";
    case IMGPROC_ESTATE: return "invalid call order / state";
    case IMGPROC_ESIZE: return "frame exceeds context maximum";
    case IMGPROC_ENODEV: return "no usable HIP device";
    case IMGPROC_EUNSUPPORTED: return "unsupported request";
    default: return "unknown status";
    }
}

IMGPROC_API const char *imgproc_last_error(const imgproc_ctx *ctx) { return ctx ? ctx->err : "null context"; }

IMGPROC_API int imgproc_device_count(int *count)
{
    if (!count) return IMGPROC_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        *count = 0;
        return IMGPROC_ENODEV;
    }
    *count = n;
    return IMGPROC_OK;
}

// No C++ exception crosses the ABI: the two entry points that grow std::vectors (the slots, the plan) catch bad_alloc.
static int set_pipeline_impl(imgproc_ctx *c, const imgproc_pipeline *pp);

IMGPROC_API int imgproc_create(imgproc_ctx **out, const imgproc_config *cfg)
{
    if (!out || !cfg) return IMGPROC_EINVAL;
    *out = nullptr;
    if (cfg->max_width < 1 || cfg->max_height < 1 || cfg->n_slots < 1 || cfg->n_slots > 16 || cfg->max_batch < 0)
        return IMGPROC_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return IMGPROC_ENODEV;
    if (cfg->device < 0 || cfg->device >= n) return IMGPROC_ENODEV;
    imgproc_ctx *c = new (std::nothrow) imgproc_ctx();
    if (!c) return IMGPROC_ENOMEM;
    c->cfg = *cfg;
    if (c->cfg.max_batch < 1) c->cfg.max_batch = 1;
    if (const char *e = getenv("IMGPROC_GRAPH")) c->graph_mode = e[0] == '0' ? 0 : e[0] == '1' ? 1 : 2;
    int copy_helpers = 3;
    if (const char *e = getenv("IMGPROC_COPY_ALWAYS")) c->copy_always = e[0] == '1';
    if (const char *e = getenv("IMGPROC_COPY_CHUNK_KB")) c->copy_chunk_bytes = (size_t)(atoi(e) < 64 ? 64 : atoi(e)) << 10;
    if (const char *e = getenv("IMGPROC_COPY_THREADS")) copy_helpers = atoi(e) < 0 ? 0 : atoi(e) > 7 ? 7 : atoi(e);
    if (copy_helpers > 0) {
        try {
            c->copy_pool = new CopyPool(copy_helpers);
        } catch (...) {
            c->copy_pool = nullptr;  // no threads to be had: the single-threaded copy still works
        }
    }
    auto build = [&]() -> int {
        HIPCHK(c, hipSetDevice(c->cfg.device));
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        // ticket 1 of the device-batch path gets a stream of its own unless IMGPROC_BATCH_STREAMS=1 (A/B: DESIGN.md 4)
        const char *bs = getenv("IMGPROC_BATCH_STREAMS");
        if (!bs || atoi(bs) != 1)
            for (int t = 1; t < IMGPROC_MAX_TICKETS; t++) HIPCHK(c, hipStreamCreateWithFlags(&c->tstream[t], hipStreamNonBlocking));
        HIPCHK(c, hipEventCreate(&c->ev_ref));
        HIPCHK(c, hipEventRecord(c->ev_ref, c->stream));
        // HSV reciprocal tables: cvRound == round-half-even == lrint
        int tab[512];
        tab[0] = tab[256] = 0;
        for (int i = 1; i < 256; i++) {
            tab[i] = (int)lrint((255 << 12) / (double)i);
            tab[256 + i] = (int)lrint((180 << 12) / (6.0 * i));
        }
        HIPCHK(c, hipMalloc((void **)&c->d_tables, sizeof tab));
        HIPCHK(c, hipMemcpy(c->d_tables, tab, sizeof tab, hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc((void **)&c->d_groups, sizeof(ipm::Group) * 32));
        const size_t px = (size_t)c->cfg.max_width * c->cfg.max_height;
        HIPCHK(c, hipMalloc((void **)&c->d_s0, px * 3));
        HIPCHK(c, hipMalloc((void **)&c->d_s1, px * 3));
        HIPCHK(c, hipMalloc((void **)&c->d_s_mom, sizeof(unsigned long long) * 6));
        const size_t mb = sizeof(unsigned long long) * 6 * IMGPROC_MAX_RANGES * (size_t)c->cfg.max_batch;
        for (auto &b : c->batch) {
            HIPCHK(c, hipMalloc((void **)&b.d_mom, mb));
            HIPCHK(c, hipHostMalloc((void **)&b.h_mom, mb, hipHostMallocDefault));
            HIPCHK(c, hipEventCreate(&b.ev_k0));
            HIPCHK(c, hipEventCreate(&b.ev_k1));
            HIPCHK(c, hipEventCreateWithFlags(&b.ev_done, hipEventDisableTiming));
        }
        {  // this device's address of the no-allocation scratch: once per device, with the device current
            if (c->cfg.device >= NA_MAX_DEVICES) return fail(c, IMGPROC_ENODEV, "device ordinal %d >= %d", c->cfg.device, NA_MAX_DEVICES);
            std::lock_guard<std::mutex> lk(g_scratch_mu);
            if (!g_scratch_ptr[c->cfg.device]) {
                unsigned char *ptr = nullptr;
                HIPCHK(c, hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
                if (!ptr) return fail(c, IMGPROC_EHIP, "hipGetSymbolAddress(scratch) returned a null address");
                g_scratch_ptr[c->cfg.device] = ptr;
            }
            c->na_scratch = g_scratch_ptr[c->cfg.device];
            const size_t na_frames = (size_t)c->cfg.max_batch < NA_MOM_ENTRIES ? (size_t)c->cfg.max_batch : NA_MOM_ENTRIES;
            HIPCHK(c, hipHostMalloc((void **)&c->h_na_mom, sizeof(unsigned long long) * 6 * IMGPROC_MAX_RANGES * na_frames, hipHostMallocDefault));
        }
        c->slots.resize(c->cfg.n_slots);
        for (auto &s : c->slots) {
            int rc = alloc_slot(c, s);
            if (rc) return rc;
        }
        return 0;
    };
    int rc;
    try {
        rc = build();
    } catch (...) {
        rc = IMGPROC_ENOMEM;
    }
    if (rc) {
        imgproc_destroy(c);
        return rc;
    }
    // default pipeline: the pose_estimate chain (BGR2HSV + inRange, pose_estimate.cpp:203-205), cfg default range
    imgproc_pipeline p;
    memset(&p, 0, sizeof p);
    imgproc_set_pipeline(c, &p);
    const int32_t lo[1][3] = {{0, 0, 0}}, hi[1][3] = {{256, 256, 256}};
    imgproc_set_ranges(c, 1, lo, hi);
    *out = c;
    return IMGPROC_OK;
}

IMGPROC_API void imgproc_destroy(imgproc_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (hipStream_t t : c->tstream)
        if (t) (void)hipStreamSynchronize(t);
    for (auto &s : c->slots) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        free_slot(s);
    }
    free_slot(c->work);
    if (c->d_tables) (void)hipFree(c->d_tables);
    if (c->d_groups) (void)hipFree(c->d_groups);
    for (u8 *b : c->d_thresh_batch)
        if (b) (void)hipFree(b);
    if (c->d_s0) (void)hipFree(c->d_s0);
    if (c->d_s1) (void)hipFree(c->d_s1);
    if (c->d_s_mom) (void)hipFree(c->d_s_mom);
    if (c->h_na_mom) (void)hipHostFree(c->h_na_mom);
    for (auto &b : c->batch) {
        if (b.d_mom) (void)hipFree(b.d_mom);
        if (b.h_mom) (void)hipHostFree(b.h_mom);
        if (b.ev_k0) (void)hipEventDestroy(b.ev_k0);
        if (b.ev_k1) (void)hipEventDestroy(b.ev_k1);
        if (b.ev_done) (void)hipEventDestroy(b.ev_done);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (hipStream_t t : c->tstream)
        if (t) (void)hipStreamDestroy(t);
    if (c->ev_ref) (void)hipEventDestroy(c->ev_ref);
    delete c->copy_pool;
    delete c;
}

// ================================================================================================
// configuration
// ================================================================================================
static void replan_fused(imgproc_ctx *c)
{
    Plan &p = c->plan;
    p.fused_ok = !p.pipe.force_unfused && !p.pipe.keep_stages &&
                 ipf::plan_fused(p.pipe, p.elems, p.pre, p.post, p.bounds, &p.fused);
    p.front_ok = !p.fused_ok && !p.pipe.force_unfused && !p.pipe.keep_stages && ipf::plan_front(p.pipe, p.bounds, &p.front);
}

IMGPROC_API int imgproc_set_pipeline(imgproc_ctx *c, const imgproc_pipeline *pp)
{
    try {
        return set_pipeline_impl(c, pp);
    } catch (...) {
        return fail(c, IMGPROC_ENOMEM, "set_pipeline: out of host memory");
    }
}

static int set_pipeline_impl(imgproc_ctx *c, const imgproc_pipeline *pp)
{
    if (!c || !pp) return IMGPROC_EINVAL;
    if (pp->encoding < 0 || pp->encoding > IMGPROC_ENC_BAYER_RGGB8) return fail(c, IMGPROC_EINVAL, "encoding %d", pp->encoding);
    if (pp->n_morph < 0 || pp->n_morph > IMGPROC_MAX_MORPH) return fail(c, IMGPROC_EINVAL, "n_morph %d", pp->n_morph);
    Plan np;
    np.pipe = *pp;
    np.bounds = c->plan.bounds;
    const int ks[2] = {pp->pre_blur_k, pp->post_blur_k};
    ipk::Taps *tp[2] = {&np.pre, &np.post};
    for (int i = 0; i < 2; i++) {
        int k = ks[i] < 1 ? 1 : ks[i];
        if (gaussian_taps(k, tp[i]->t)) return fail(c, IMGPROC_EINVAL, "blur ksize %d (odd, <=%d)", ks[i], IMGPROC_MAX_KSIZE);
        tp[i]->k = k;
    }
    np.custom.resize(pp->n_morph);
    for (int i = 0; i < pp->n_morph; i++) {
        const imgproc_morph &m = pp->morph[i];
        if (m.shape < 0 || m.shape > IMGPROC_CUSTOM) return fail(c, IMGPROC_EINVAL, "morph[%d] shape %d", i, m.shape);
        if (m.shape == IMGPROC_CUSTOM && m.custom && m.kw > 0 && m.kh > 0 && m.kw <= 31 && m.kh <= 31)
            np.custom[i].assign(m.custom, m.custom + (size_t)m.kw * m.kh);
        int rc = make_element(m, np.custom[i].empty() ? nullptr : np.custom[i].data(), np.elems[i]);
        if (rc) return fail(c, rc, "morph[%d]: bad element %dx%d anchor (%d,%d)", i, m.kw, m.kh, m.anchor_x, m.anchor_y);
        np.pipe.morph[i].custom = np.custom[i].empty() ? nullptr : np.custom[i].data();
    }
    c->plan = std::move(np);
    c->plan_epoch++;
    for (int i = 0; i < c->plan.pipe.n_morph; i++)
        c->plan.pipe.morph[i].custom = c->plan.custom[i].empty() ? nullptr : c->plan.custom[i].data();
    replan_fused(c);
    c->plan.bm_ok = c->plan.pipe.n_morph > 0 && ipm::plan_bitmorph(c->plan.pipe, c->plan.elems, &c->plan.bm);
    if (c->plan.bm_ok) {
        HIPCHK(c, hipSetDevice(c->cfg.device));
        HIPCHK(c, imgproc_synchronize(c) == IMGPROC_OK ? hipSuccess : hipErrorUnknown);  // nothing in flight may still read the old list
        HIPCHK(c, hipMemcpy(c->d_groups, c->plan.bm.groups.data(), sizeof(ipm::Group) * c->plan.bm.groups.size(), hipMemcpyHostToDevice));
    }
    return IMGPROC_OK;
}

IMGPROC_API int imgproc_set_ranges(imgproc_ctx *c, int n, const int32_t (*lo)[3], const int32_t (*hi)[3])
{
    if (!c || !lo || !hi) return IMGPROC_EINVAL;
    if (n < 1 || n > IMGPROC_MAX_RANGES) return fail(c, IMGPROC_EINVAL, "n ranges %d (1..%d)", n, IMGPROC_MAX_RANGES);
    make_bounds(n, lo, hi, c->plan.bounds);
    c->plan_epoch++;
    replan_fused(c);
    return IMGPROC_OK;
}

IMGPROC_API int imgproc_pipeline_is_fused(const imgproc_ctx *c) { return c && c->plan.fused_ok ? 1 : 0; }

IMGPROC_API int imgproc_pipeline_plan(const imgproc_ctx *c, char *buf, size_t len)
{
    if (!c || !buf || !len) return IMGPROC_EINVAL;
    const Plan &p = c->plan;
    if (p.fused_ok)
        snprintf(buf, len, "fused chain kernel%s", p.fused.bayer ? " (Bayer front end inside)" : "");
    else if (p.front_ok && p.bm_ok)
        snprintf(buf, len, "front kernel%s + bit-plane morphology (%d groups)", p.front.bayer ? " (Bayer front end inside)" : "", (int)p.bm.groups.size());
    else if (p.bm_ok)
        snprintf(buf, len, "stage kernels + bit-plane morphology (%d groups)", (int)p.bm.groups.size());
    else
        snprintf(buf, len, "stage kernels");
    return IMGPROC_OK;
}

// ================================================================================================
// streaming path
// ================================================================================================
static int submit_impl(imgproc_ctx *c, int slot, const uint8_t *data, size_t stride, int w, int h, bool zero_copy);

IMGPROC_API int imgproc_submit(imgproc_ctx *c, int slot, const uint8_t *data, size_t stride, int w, int h)
{
    return submit_impl(c, slot, data, stride, w, h, false);
}

// The same without the staging copy: the DMA engine reads the caller's buffer, which must be page-locked
// (imgproc_host_alloc, hipHostMalloc, hipHostRegister) and stay untouched until imgproc_wait returns for the slot.
IMGPROC_API int imgproc_submit_pinned(imgproc_ctx *c, int slot, const uint8_t *data, size_t stride, int w, int h)
{
    return submit_impl(c, slot, data, stride, w, h, true);
}

static int submit_impl(imgproc_ctx *c, int slot, const uint8_t *data, size_t stride, int w, int h, bool zero_copy)
{
    if (!c || !data) return IMGPROC_EINVAL;
    if (slot < 0 || slot >= (int)c->slots.size()) return fail(c, IMGPROC_EINVAL, "slot %d", slot);
    int rc = check_frame(c, w, h, stride);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    Slot &s = c->slots[slot];
    if (s.in_flight) {  // slot reuse before wait(): finish the old frame first (its result is dropped)
        HIPCHK(c, hipEventSynchronize(s.ev_done));
        s.in_flight = false;
    }
    const Plan &p = c->plan;
    const size_t rowb = (size_t)w * (p.pipe.encoding == IMGPROC_ENC_BAYER_RGGB8 ? 1 : 3);
    // Staging: the caller's frame goes into the slot's pinned buffer (so msg->data may be released on return) and from
    // there to HBM.  Large frames go in row chunks, each chunk's DMA starting while the CPU copies the next one; with a
    // captured graph (IMGPROC_GRAPH) the whole-frame H2D is part of the graph instead.
    const bool graph_path = c->graph_mode != 0 && !zero_copy;  // the captured graph uploads from the slot's staging buffer
    const bool big = rowb * h >= (1u << 20);
    // The helper threads raise throughput (4.1 k -> 6.9 k frames/s at 1920x1200 with three slots) but not the latency of
    // a lone frame (0.26 ms against 0.24 ms: waking them costs more than the single-threaded copy/DMA pipeline loses), so
    // they are used only when the caller is pipelining, i.e. another slot is in flight at this submit.
    bool others_in_flight = false;
    for (const Slot &o : c->slots) others_in_flight |= (&o != &s) && o.in_flight;
    const bool pooled = big && c->copy_pool && (others_in_flight || c->copy_always);
    CopyPool::Frame fr;
    fr.dst = s.h_in;
    fr.src = data;
    fr.rowb = rowb;
    fr.stride = stride;
    fr.h = h;
    fr.n_chunks = 1;
    if (pooled) {
        const size_t want = (rowb * h + c->copy_chunk_bytes - 1) / c->copy_chunk_bytes;
        fr.n_chunks = want > (size_t)CopyPool::MAX_CHUNKS ? CopyPool::MAX_CHUNKS : (int)want;
    } else if (!graph_path && big) {
        fr.n_chunks = 4;
    }
    if (fr.n_chunks > h) fr.n_chunks = h;
    hipError_t e_up = hipSuccess;
    auto issue = [&](int ci) {
        if (graph_path || e_up != hipSuccess) return;
        int y0, y1;
        CopyPool::chunk_rows(fr, ci, &y0, &y1);
        e_up = hipMemcpyAsync(s.d_raw + (size_t)y0 * rowb, s.h_in + (size_t)y0 * rowb, rowb * (size_t)(y1 - y0),
                              hipMemcpyHostToDevice, s.stream);
    };
    if (zero_copy) {
        // straight from the caller's page-locked frame: one DMA (two-dimensional when its rows are padded), no CPU copy
        if (stride == rowb)
            e_up = hipMemcpyAsync(s.d_raw, data, rowb * (size_t)h, hipMemcpyHostToDevice, s.stream);
        else
            e_up = hipMemcpy2DAsync(s.d_raw, rowb, data, stride, rowb, (size_t)h, hipMemcpyHostToDevice, s.stream);
    } else if (pooled) {
        c->copy_pool->stage(fr, issue);
    } else {
        for (int ci = 0; ci < fr.n_chunks; ci++) {
            CopyPool::copy_chunk(fr, ci);
            issue(ci);
        }
    }
    HIPCHK(c, e_up);
    // The per-frame work {H2D, memset, chain kernels, D2H of the moments} has fixed pointers and sizes for a given
    // (slot, w, h, plan): it is captured once into a hipGraph and replayed with one hipGraphLaunch per frame.
    // 0: chain kernel, 1: front kernel + bit-plane back end, 2: stage by stage
    const int path = (p.fused_ok && ipf::can_launch_fused(p.fused, s.d_raw, 0, rowb, w, h, s.d_mask)) ? 0
                     : (p.front_ok && p.bm_ok && ipf::can_launch_fused(p.front, s.d_raw, 0, rowb, w, h, s.d_thresh)) ? 1 : 2;
    auto enqueue = [&](hipStream_t st) -> int {
        if (graph_path) HIPCHK(c, hipMemcpyAsync(s.d_raw, s.h_in, rowb * h, hipMemcpyHostToDevice, st));
        const int nr = p.bounds.n;
        HIPCHK(c, hipMemsetAsync(s.d_mom, 0, sizeof(unsigned long long) * 6 * nr, st));
        if (path == 0) {
            // BGR8 / RGB8 / BayerRG8 all go straight into the chain kernel (the Bayer front end is part of it)
            HIPCHK(c, ipf::launch_fused(p.fused, s.d_raw, 0, rowb, w, h, 1, s.d_mask, s.d_mom, c->d_tables, st));
        } else if (path == 1) {
            // front kernel ([demosaic] [blur3] HSV [blur5] inRange -> bytes) + bit-plane back end (morph list, moments, mask)
            // the inRange planes go from one kernel to the other as bits (1/8 B per px and range) when the width allows
            uint32_t *bits = ipf::front_writes_bits(w) ? reinterpret_cast<uint32_t *>(s.d_thresh) : nullptr;
            HIPCHK(c, ipf::launch_fused(p.front, s.d_raw, 0, rowb, w, h, 1, s.d_thresh, nullptr, c->d_tables, st, bits));
            HIPCHK(c, ipm::launch_bitmorph(p.bm, c->d_groups, s.d_thresh, s.d_mask, s.d_mom, w, h, nr, p.pipe.second_order, st, bits));
        } else {
            int rc2 = run_unfused(c, s, s.d_raw, rowb, w, h, s.d_mask, s.d_mom, nullptr, st);
            if (rc2) return rc2;
            if (p.pipe.keep_stages)
                HIPCHK(c, ipk::launch_mask_and(s.bgr_ptr, s.bgr_stride, s.d_mask, (size_t)w, s.d_masked, (size_t)w * 3, w, h, st));
        }
        HIPCHK(c, hipMemcpyAsync(s.h_mom, s.d_mom, sizeof(unsigned long long) * 6 * nr, hipMemcpyDeviceToHost, st));
        return 0;
    };
    // Measured on MI355X (scripts/stream_bench.py, 1920x1200, 3 slots): replaying the captured graph is SLOWER than
    // direct launches both for the fused chain (3.8 k vs 4.3 k frames/s) and for polygon_path's 23-launch chain (2.2 k vs
    // 3.1 k frames/s), so direct launches are the default and the graph path stays selectable.
    const bool fused_now = path == 0;
    if (!zero_copy && (c->graph_mode == 1 || (c->graph_mode == 2 && !fused_now))) {
        if (!s.graph_exec || s.graph_w != w || s.graph_h != h || s.graph_epoch != c->plan_epoch) {
            if (s.graph_exec) {
                (void)hipGraphExecDestroy(s.graph_exec);
                s.graph_exec = nullptr;
            }
            hipGraph_t g = nullptr;
            HIPCHK(c, hipStreamBeginCapture(s.stream, hipStreamCaptureModeThreadLocal));
            rc = enqueue(s.stream);
            hipError_t e_end = hipStreamEndCapture(s.stream, &g);
            if (rc) {
                if (g) (void)hipGraphDestroy(g);
                return rc;
            }
            HIPCHK(c, e_end);
            hipError_t e_inst = hipGraphInstantiate(&s.graph_exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            HIPCHK(c, e_inst);
            s.graph_w = w;
            s.graph_h = h;
            s.graph_epoch = c->plan_epoch;
        }
        HIPCHK(c, hipGraphLaunch(s.graph_exec, s.stream));
    } else {
        rc = enqueue(s.stream);
        if (rc) return rc;
    }
    HIPCHK(c, hipEventRecord(s.ev_done, s.stream));
    s.w = w;
    s.h = h;
    s.in_flight = true;
    s.has_frame = true;
    s.frame_epoch = c->plan_epoch;
    s.nr = p.bounds.n;
    s.second = p.pipe.second_order != 0;
    // the fast paths leave no colour stage behind; the stage path leaves them all, except that its inRange planes are
    // bits, not bytes, when they went straight to the bit-plane morphology
    const bool thresh_as_bits = p.bm_ok && !p.pipe.force_unfused && !p.pipe.keep_stages;
    s.valid_upto = path == 2 ? (thresh_as_bits ? IMGPROC_STAGE_BLUR : IMGPROC_STAGE_THRESH) : -1;
    s.masked_range = (path == 2 && p.pipe.keep_stages) ? 0 : -1;
    return IMGPROC_OK;
}

IMGPROC_API int imgproc_wait(imgproc_ctx *c, int slot, imgproc_result *out)
{
    if (!c || !out) return IMGPROC_EINVAL;
    if (slot < 0 || slot >= (int)c->slots.size()) return fail(c, IMGPROC_EINVAL, "slot %d", slot);
    Slot &s = c->slots[slot];
    if (!s.in_flight) return fail(c, IMGPROC_ESTATE, "wait on slot %d without a submitted frame", slot);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipEventSynchronize(s.ev_done));
    s.in_flight = false;
    memset(out, 0, sizeof *out);
    out->n_ranges = s.nr;  // as submitted: set_ranges / set_pipeline may have run since
    for (int r = 0; r < out->n_ranges; r++) finish_moments(s.h_mom + 6 * r, out->r[r], s.second);
    return IMGPROC_OK;
}

IMGPROC_API int imgproc_fetch_stage(imgproc_ctx *c, int slot, int stage, int range, uint8_t *dst, size_t dst_stride)
{
    if (!c || !dst) return IMGPROC_EINVAL;
    if (slot < 0 || slot >= (int)c->slots.size()) return fail(c, IMGPROC_EINVAL, "slot %d", slot);
    Slot &s = c->slots[slot];
    if (!s.has_frame) return fail(c, IMGPROC_ESTATE, "slot %d holds no frame", slot);
    if (range < 0 || range >= c->plan.bounds.n) return fail(c, IMGPROC_EINVAL, "range %d", range);
    if (stage < IMGPROC_STAGE_BGR || stage > IMGPROC_STAGE_MASKED) return fail(c, IMGPROC_EINVAL, "stage %d", stage);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (s.in_flight) HIPCHK(c, hipEventSynchronize(s.ev_done));
    // The fast paths (chain kernel, front kernel + bit planes) leave only the final mask behind.  Any other stage image
    // is rebuilt here, on demand, from the raw frame that is still in the slot, with the stage kernels -- bit-identical
    // to what the fast path computed internally (tests/test_gpu_chain.py), paid for only by callers that ask.
    // keep_stages != 0 produces them all at submit instead.
    const int need = stage == IMGPROC_STAGE_MASK ? -1 : stage == IMGPROC_STAGE_MASKED ? IMGPROC_STAGE_BGR : stage;
    const bool rebuild = need > s.valid_upto || (stage == IMGPROC_STAGE_MASKED && s.masked_range != range);
    if (rebuild) {
        if (s.frame_epoch != c->plan_epoch)
            return fail(c, IMGPROC_ESTATE, "stage %d: pipeline or ranges changed since the frame in slot %d was submitted", stage, slot);
        const size_t rawb = (size_t)s.w * (c->plan.pipe.encoding == IMGPROC_ENC_BAYER_RGGB8 ? 1 : 3);
        if (need > s.valid_upto) {
            int rc = run_unfused(c, s, s.d_raw, rawb, s.w, s.h, nullptr, nullptr, nullptr, s.stream, need);
            if (rc) return rc;
            s.valid_upto = need;
        }
        if (stage == IMGPROC_STAGE_MASKED && s.masked_range != range) {
            HIPCHK(c, ipk::launch_mask_and(s.bgr_ptr, s.bgr_stride, s.d_mask + (size_t)s.w * s.h * range, (size_t)s.w, s.d_masked,
                                           (size_t)s.w * 3, s.w, s.h, s.stream));
            s.masked_range = range;
        }
        HIPCHK(c, hipStreamSynchronize(s.stream));
    }
    const size_t px = (size_t)s.w * s.h;
    const u8 *src = nullptr;
    size_t rowb = 0, src_stride = 0;
    switch (stage) {
    case IMGPROC_STAGE_BGR: src = s.bgr_ptr; rowb = (size_t)s.w * 3; src_stride = s.bgr_stride; break;
    case IMGPROC_STAGE_HSV: src = s.hsv_ptr; rowb = (size_t)s.w * 3; break;
    case IMGPROC_STAGE_BLUR: src = s.blur_ptr; rowb = (size_t)s.w * 3; break;
    case IMGPROC_STAGE_THRESH: src = s.d_thresh + px * range; rowb = s.w; break;
    case IMGPROC_STAGE_MASK: src = s.d_mask + px * range; rowb = s.w; break;
    case IMGPROC_STAGE_MASKED: src = s.d_masked; rowb = (size_t)s.w * 3; break;
    default: return fail(c, IMGPROC_EINVAL, "stage %d", stage);
    }
    if (!src) return fail(c, IMGPROC_ESTATE, "stage %d was not produced", stage);
    if (dst_stride < rowb) return fail(c, IMGPROC_EINVAL, "dst stride %zu < %zu", dst_stride, rowb);
    if (!src_stride) src_stride = rowb;
    HIPCHK(c, hipMemcpy2D(dst, dst_stride, src, src_stride, rowb, s.h, hipMemcpyDeviceToHost));
    return IMGPROC_OK;
}

// ================================================================================================
// batch paths
// ================================================================================================
IMGPROC_API int imgproc_process_batch(imgproc_ctx *c, const uint8_t *frames, size_t frame_stride, size_t row_stride,
                                      int w, int h, int n, imgproc_result *out)
{
    if (!c || !frames || !out || n < 0) return IMGPROC_EINVAL;
    const int ns = (int)c->slots.size();
    for (int i = 0; i < n; i++) {
        const int slot = i % ns;
        if (i >= ns) {
            int rc = imgproc_wait(c, slot, &out[i - ns]);
            if (rc) return rc;
        }
        int rc = imgproc_submit(c, slot, frames + (size_t)i * frame_stride, row_stride, w, h);
        if (rc) return rc;
    }
    for (int i = (n > ns ? n - ns : 0); i < n; i++) {
        int rc = imgproc_wait(c, i % ns, &out[i]);
        if (rc) return rc;
    }
    return IMGPROC_OK;
}

// Two batches may be in flight: enqueue returns as soon as the work is on the stream, wait delivers the results.
IMGPROC_API int imgproc_batch_enqueue(imgproc_ctx *c, const void *d_frames, size_t frame_stride, size_t row_stride,
                                      int w, int h, int n, void *d_masks, int ticket)
{
    if (!c || !d_frames || n < 0) return IMGPROC_EINVAL;
    if (ticket < 0 || ticket >= IMGPROC_MAX_TICKETS) return fail(c, IMGPROC_EINVAL, "ticket %d (0..%d)", ticket, IMGPROC_MAX_TICKETS - 1);
    if (n > c->cfg.max_batch) return fail(c, IMGPROC_ESIZE, "batch %d > max_batch %d", n, c->cfg.max_batch);
    imgproc_ctx::Batch &bt = c->batch[ticket];
    if (bt.pending) return fail(c, IMGPROC_ESTATE, "ticket %d is still in flight: imgproc_batch_wait first", ticket);
    int rc = n ? check_frame(c, w, h, row_stride) : 0;
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const Plan &p = c->plan;
    const int nr = p.bounds.n;
    const size_t px = (size_t)w * h;
    const size_t mom_bytes = sizeof(unsigned long long) * 6 * nr * (size_t)n;
    bt.n = n;
    bt.nr = nr;
    bt.second = p.pipe.second_order != 0;
    bt.launches = 0;
    bt.timed = false;
    if (n == 0) {
        bt.pending = true;
        HIPCHK(c, hipEventRecord(bt.ev_done, c->stream));
        return IMGPROC_OK;
    }
    Slot &s = c->work;  // stage-by-stage planes of the batch path; all its users run on c->stream, in order
    const bool one_launch = p.fused_ok && ipf::can_launch_fused(p.fused, d_frames, frame_stride, row_stride, w, h, d_masks);
    const bool two_launch = !one_launch && p.front_ok && p.bm_ok &&
                            ipf::can_launch_fused(p.front, d_frames, frame_stride, row_stride, w, h, d_masks);
    // frames that need no demosaic / pre-blur go through the fused HSV + Gaussian + inRange kernel a whole group at a time
    const bool group_fused = !one_launch && !two_launch && bit_plane_tail(p) && gauss_fused_wanted(p) &&
                             p.pipe.encoding != IMGPROC_ENC_BAYER_RGGB8 && p.pipe.pre_blur_k <= 1 &&
                             ipk::gauss_fused_ok((const u8 *)d_frames, frame_stride, row_stride, w, h, p.post, p.bounds);
    // Paths that touch nothing of the context but read-only tables and this ticket's own buffers may run ticket 1 on a
    // stream of its own (default; IMGPROC_BATCH_STREAMS=1 turns it off), so kernels of consecutive batches overlap on the GPU.
    hipStream_t const st = (c->tstream[ticket] && (one_launch || two_launch || group_fused)) ? c->tstream[ticket] : c->stream;
    bool overlapped = false;  // another ticket's kernels run beside this batch's
    for (int t = 0; t < IMGPROC_MAX_TICKETS; t++) overlapped |= t != ticket && c->batch[t].pending && c->tstream[1];
    u8 *&planes = c->d_thresh_batch[ticket];
    size_t &planes_bytes = c->thresh_batch_bytes[ticket];
    auto grow_planes = [&](size_t need) -> int {  // grown on demand, kept for the life of the context; only this ticket uses it
        if (need <= planes_bytes) return IMGPROC_OK;
        if (planes) HIPCHK(c, hipFree(planes));  // the ticket is not pending: nothing in flight reads it
        planes = nullptr;
        planes_bytes = 0;
        HIPCHK(c, hipMalloc((void **)&planes, need));
        planes_bytes = need;
        return IMGPROC_OK;
    };
    HIPCHK(c, hipMemsetAsync(bt.d_mom, 0, mom_bytes, st));
    if (c->profiling) HIPCHK(c, hipEventRecord(bt.ev_k0, st));
    if (one_launch) {
        HIPCHK(c, ipf::launch_fused(p.fused, (const u8 *)d_frames, frame_stride, row_stride, w, h, n, (u8 *)d_masks, bt.d_mom,
                                    c->d_tables, st, nullptr, overlapped));
        bt.launches = 1;
    } else if (two_launch) {
        const bool as_bits = ipf::front_writes_bits(w);
        rc = grow_planes(as_bits ? (size_t)n * nr * h * ipm::bit_words(w) * 4 : (size_t)n * nr * px);
        if (rc) return rc;
        uint32_t *bits = as_bits ? reinterpret_cast<uint32_t *>(planes) : nullptr;
        HIPCHK(c, ipf::launch_fused(p.front, (const u8 *)d_frames, frame_stride, row_stride, w, h, n, planes,
                                    nullptr, c->d_tables, st, bits, overlapped));
        HIPCHK(c, ipm::launch_bitmorph(p.bm, c->d_groups, planes, (u8 *)d_masks, bt.d_mom, w, h, n * nr,
                                       p.pipe.second_order, st, bits));
        bt.launches = 2;
    } else if (bit_plane_tail(p)) {
        // stage kernels per frame up to the inRange bit planes, then the bit-plane morphology of a whole group of frames in
        // one launch (planes = frames x ranges): at one 8K frame it has too few independent columns to fill the chip
        const size_t per_frame = (size_t)nr * h * ipm::bit_words(w) * 4;
        const size_t cap = (size_t)1 << 30;  // at most 1 GiB of bit planes at a time
        const int group = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, cap / std::max<size_t>(per_frame, 1)));
        rc = grow_planes(per_frame * group);
        if (rc) return rc;
        uint32_t *bits = reinterpret_cast<uint32_t *>(planes);
        if (!group_fused) {  // the stage kernels work in the context's own planes
            rc = ensure_work(c);
            if (rc) return rc;
        }
        for (int g0 = 0; g0 < n; g0 += group) {
            const int gn = std::min(group, n - g0);
            if (group_fused) {
                HIPCHK(c, ipk::launch_gauss_fused((const u8 *)d_frames + (size_t)g0 * frame_stride, frame_stride, row_stride, bits, w, h, gn,
                                                  p.post, p.bounds, p.pipe.encoding == IMGPROC_ENC_RGB8 ? 2 : 0, c->d_tables, st, overlapped));
                bt.launches++;
            }
            for (int i = 0; i < gn && !group_fused; i++) {
                rc = run_unfused(c, s, (const u8 *)d_frames + (size_t)(g0 + i) * frame_stride, row_stride, w, h, nullptr, nullptr,
                                 &bt.launches, st, IMGPROC_STAGE_MASK, bits + (size_t)i * (per_frame / 4));
                if (rc) return rc;
            }
            u8 *mout = d_masks ? (u8 *)d_masks + (size_t)g0 * nr * px : nullptr;  // no mask buffer: moments only
            HIPCHK(c, ipm::launch_bitmorph(p.bm, c->d_groups, nullptr, mout, bt.d_mom + (size_t)g0 * nr * 6, w, h, gn * nr,
                                           p.pipe.second_order, st, bits));
            bt.launches++;
        }
    } else {
        rc = ensure_work(c);
        if (rc) return rc;
        for (int i = 0; i < n; i++) {
            u8 *mout = d_masks ? (u8 *)d_masks + (size_t)i * nr * px : s.d_mask;
            rc = run_unfused(c, s, (const u8 *)d_frames + (size_t)i * frame_stride, row_stride, w, h, mout,
                             bt.d_mom + (size_t)i * nr * 6, &bt.launches, st);
            if (rc) return rc;
        }
    }
    if (c->profiling) {
        HIPCHK(c, hipEventRecord(bt.ev_k1, st));
        bt.timed = true;
    }
    HIPCHK(c, hipMemcpyAsync(bt.h_mom, bt.d_mom, mom_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipEventRecord(bt.ev_done, st));
    bt.pending = true;
    return IMGPROC_OK;
}

IMGPROC_API int imgproc_batch_wait(imgproc_ctx *c, int ticket, imgproc_result *out)
{
    if (!c) return IMGPROC_EINVAL;
    if (ticket < 0 || ticket >= IMGPROC_MAX_TICKETS) return fail(c, IMGPROC_EINVAL, "ticket %d (0..%d)", ticket, IMGPROC_MAX_TICKETS - 1);
    imgproc_ctx::Batch &bt = c->batch[ticket];
    if (!bt.pending) return fail(c, IMGPROC_ESTATE, "ticket %d holds no batch", ticket);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipEventSynchronize(bt.ev_done));
    bt.pending = false;
    c->last_ticket = ticket;
    if (out) {
        for (int i = 0; i < bt.n; i++) {
            memset(&out[i], 0, sizeof out[i]);
            out[i].n_ranges = bt.nr;
            for (int r = 0; r < bt.nr; r++) finish_moments(bt.h_mom + ((size_t)i * bt.nr + r) * 6, out[i].r[r], bt.second != 0);
        }
    }
    return IMGPROC_OK;
}

IMGPROC_API int imgproc_process_batch_device(imgproc_ctx *c, const void *d_frames, size_t frame_stride,
                                             size_t row_stride, int w, int h, int n, void *d_masks,
                                             imgproc_result *out)
{
    if (!c || !d_frames || n < 0) return IMGPROC_EINVAL;
    if (c->batch[0].pending) return fail(c, IMGPROC_ESTATE, "ticket 0 is in flight (imgproc_batch_enqueue): wait for it first");
    int rc = imgproc_batch_enqueue(c, d_frames, frame_stride, row_stride, w, h, n, d_masks, 0);
    if (rc) return rc;
    return imgproc_batch_wait(c, 0, out);
}

// The one entry point that allocates nothing (include/imgproc.h): imgproc_process_batch_device with the library's
// scratch (moment sums, the inRange planes between the front kernel and the bit-plane kernel) in the per-GPU static
// array `scratch` instead of hipMalloc'ed buffers.  No hipMalloc / hipFree / hipHostMalloc on this path: the pinned
// landing buffer and the scratch address were set up by imgproc_create.
IMGPROC_API int imgproc_process_batch_device_noalloc(imgproc_ctx *c, const void *d_frames, size_t frame_stride,
                                                     size_t row_stride, int w, int h, int n, void *d_masks,
                                                     imgproc_result *out)
{
    if (!c || !d_frames || !out || n < 0) return IMGPROC_EINVAL;
    if (n == 0) return IMGPROC_OK;
    int rc = check_frame(c, w, h, row_stride);
    if (rc) return rc;
    const Plan &p = c->plan;
    const int nr = p.bounds.n;
    const size_t px = (size_t)w * h;
    if (n > c->cfg.max_batch) return fail(c, IMGPROC_ESIZE, "batch %d > max_batch %d", n, c->cfg.max_batch);
    if ((size_t)n * nr > NA_MOM_ENTRIES)
        return fail(c, IMGPROC_ESIZE, "no-allocation path: %d frames x %d ranges > %zu moment entries", n, nr, NA_MOM_ENTRIES);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    unsigned long long *d_mom = reinterpret_cast<unsigned long long *>(c->na_scratch);
    u8 *d_planes = c->na_scratch + NA_MOM_BYTES;
    const size_t mom_bytes = sizeof(unsigned long long) * 6 * nr * (size_t)n;
    struct Busy {  // released on every return path below
        std::atomic<bool> *f = nullptr;
        ~Busy() { if (f) f->store(false, std::memory_order_release); }
    } busy;
    const bool chain = p.fused_ok && ipf::can_launch_fused(p.fused, d_frames, frame_stride, row_stride, w, h, d_masks);
    const bool front = !chain && p.front_ok && p.bm_ok && ipf::can_launch_fused(p.front, d_frames, frame_stride, row_stride, w, h, d_masks);
    if (!chain && !front)
        return fail(c, IMGPROC_EUNSUPPORTED, "no-allocation path: this pipeline / frame shape runs stage by stage (%dx%d)", w, h);
    const bool as_bits = ipf::front_writes_bits(w);  // bit planes (1/8 B per px) unless w % 4 != 0
    const size_t plane_bytes = as_bits ? (size_t)h * ipm::bit_words(w) * 4 : px;
    if (front && (size_t)n * nr * plane_bytes > NA_PLANE_BYTES)
        return fail(c, IMGPROC_ESIZE, "no-allocation path: %d frames x %d ranges x %dx%d inRange planes (%zu bytes each) > %zu scratch bytes",
                    n, nr, w, h, plane_bytes, NA_PLANE_BYTES);
    if (g_scratch_busy[c->cfg.device].exchange(true, std::memory_order_acquire))
        return fail(c, IMGPROC_ESTATE, "no-allocation path: another call is using GPU %d's scratch array (one such call per GPU at a time)",
                    c->cfg.device);
    busy.f = &g_scratch_busy[c->cfg.device];
    // the array keeps the last call's contents: the sums must start at zero (the planes are fully rewritten)
    HIPCHK(c, hipMemsetAsync(d_mom, 0, mom_bytes, c->stream));
    if (chain) {
        HIPCHK(c, ipf::launch_fused(p.fused, (const u8 *)d_frames, frame_stride, row_stride, w, h, n, (u8 *)d_masks, d_mom,
                                    c->d_tables, c->stream));
    } else {
        uint32_t *bits = as_bits ? reinterpret_cast<uint32_t *>(d_planes) : nullptr;
        HIPCHK(c, ipf::launch_fused(p.front, (const u8 *)d_frames, frame_stride, row_stride, w, h, n, d_planes, nullptr,
                                    c->d_tables, c->stream, bits));
        HIPCHK(c, ipm::launch_bitmorph(p.bm, c->d_groups, d_planes, (u8 *)d_masks, d_mom, w, h, n * nr, p.pipe.second_order,
                                       c->stream, bits));
    }
    HIPCHK(c, hipMemcpyAsync(c->h_na_mom, d_mom, mom_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) {
        memset(&out[i], 0, sizeof out[i]);
        out[i].n_ranges = nr;
        for (int r = 0; r < nr; r++)
            finish_moments(c->h_na_mom + ((size_t)i * nr + r) * 6, out[i].r[r], p.pipe.second_order != 0);
    }
    return IMGPROC_OK;
}

// ================================================================================================
// single stages on host buffers
// ================================================================================================
namespace {
int stage_io_check(imgproc_ctx *c, const void *src, void *dst, int w, int h, size_t sstride, size_t dstride,
                   size_t srow, size_t drow)
{
    if (!c || !src || !dst) return IMGPROC_EINVAL;
    if (w < 1 || h < 1) return fail(c, IMGPROC_EINVAL, "frame %dx%d", w, h);
    if ((size_t)w * h > (size_t)c->cfg.max_width * c->cfg.max_height)
        return fail(c, IMGPROC_ESIZE, "frame %dx%d exceeds context maximum", w, h);
    if (sstride < srow || dstride < drow) return fail(c, IMGPROC_EINVAL, "stride too small");
    return 0;
}
}  // namespace

#define UPLOAD(c, dptr, hptr, hstride, rowb, h) \
    HIPCHK(c, hipMemcpy2DAsync(dptr, rowb, hptr, hstride, rowb, h, hipMemcpyHostToDevice, (c)->stream))
#define DOWNLOAD(c, hptr, hstride, dptr, rowb, h)                                                           \
    do {                                                                                                    \
        HIPCHK(c, hipMemcpy2DAsync(hptr, hstride, dptr, rowb, rowb, h, hipMemcpyDeviceToHost, (c)->stream)); \
        HIPCHK(c, hipStreamSynchronize((c)->stream));                                                       \
    } while (0)

IMGPROC_API int imgproc_cvt_bgr2hsv(imgproc_ctx *c, const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride,
                                    int w, int h, int encoding)
{
    const size_t rowb = (size_t)w * 3;
    int rc = stage_io_check(c, src, dst, w, h, sstride, dstride, rowb, rowb);
    if (rc) return rc;
    if (encoding != IMGPROC_ENC_BGR8 && encoding != IMGPROC_ENC_RGB8) return fail(c, IMGPROC_EINVAL, "encoding %d", encoding);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    UPLOAD(c, c->d_s0, src, sstride, rowb, h);
    HIPCHK(c, ipk::launch_bgr2hsv(c->d_s0, rowb, c->d_s1, rowb, w, h, encoding == IMGPROC_ENC_RGB8 ? 2 : 0, c->d_tables, c->stream));
    DOWNLOAD(c, dst, dstride, c->d_s1, rowb, h);
    return IMGPROC_OK;
}

IMGPROC_API int imgproc_gaussian_blur(imgproc_ctx *c, const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride,
                                      int w, int h, int channels, int ksize)
{
    if (channels != 1 && channels != 3) return c ? fail(c, IMGPROC_EINVAL, "channels %d", channels) : IMGPROC_EINVAL;
    const size_t rowb = (size_t)w * channels;
    int rc = stage_io_check(c, src, dst, w, h, sstride, dstride, rowb, rowb);
    if (rc) return rc;
    ipk::Taps t;
    t.k = ksize < 1 ? 1 : ksize;
    if (gaussian_taps(t.k, t.t)) return fail(c, IMGPROC_EINVAL, "ksize %d (odd, <=%d)", ksize, IMGPROC_MAX_KSIZE);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    UPLOAD(c, c->d_s0, src, sstride, rowb, h);
    HIPCHK(c, ipk::launch_gaussian(c->d_s0, rowb, c->d_s1, rowb, w, h, channels, t, c->stream));
    DOWNLOAD(c, dst, dstride, c->d_s1, rowb, h);
    return IMGPROC_OK;
}

IMGPROC_API int imgproc_in_range(imgproc_ctx *c, const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride,
                                 int w, int h, const int32_t lo[3], const int32_t hi[3])
{
    if (!lo || !hi) return IMGPROC_EINVAL;
    const size_t rowb = (size_t)w * 3;
    int rc = stage_io_check(c, src, dst, w, h, sstride, dstride, rowb, (size_t)w);
    if (rc) return rc;
    ipk::Bounds b;
    make_bounds(1, reinterpret_cast<const int32_t(*)[3]>(lo), reinterpret_cast<const int32_t(*)[3]>(hi), b);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    UPLOAD(c, c->d_s0, src, sstride, rowb, h);
    HIPCHK(c, ipk::launch_in_range(c->d_s0, rowb, c->d_s1, (size_t)w, (size_t)w * h, w, h, b, c->stream));
    DOWNLOAD(c, dst, dstride, c->d_s1, (size_t)w, h);
    return IMGPROC_OK;
}

IMGPROC_API int imgproc_morphology(imgproc_ctx *c, const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride,
                                   int w, int h, const imgproc_morph *op)
{
    if (!op) return IMGPROC_EINVAL;
    int rc = stage_io_check(c, src, dst, w, h, sstride, dstride, (size_t)w, (size_t)w);
    if (rc) return rc;
    ipk::Element e;
    if (op->shape < 0 || op->shape > IMGPROC_CUSTOM) return fail(c, IMGPROC_EINVAL, "shape %d", op->shape);
    rc = make_element(*op, op->custom, e);
    if (rc) return fail(c, rc, "bad element %dx%d anchor (%d,%d)", op->kw, op->kh, op->anchor_x, op->anchor_y);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    UPLOAD(c, c->d_s0, src, sstride, (size_t)w, h);
    HIPCHK(c, ipk::launch_morph(c->d_s0, (size_t)w, c->d_s1, (size_t)w, w, h, op->kind, e, c->stream));
    DOWNLOAD(c, dst, dstride, c->d_s1, (size_t)w, h);
    return IMGPROC_OK;
}

IMGPROC_API int imgproc_mask_moments(imgproc_ctx *c, const uint8_t *mask, size_t stride, int w, int h, imgproc_moments *out)
{
    int rc = stage_io_check(c, mask, out, w, h, stride, (size_t)w, (size_t)w, (size_t)w);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    UPLOAD(c, c->d_s0, mask, stride, (size_t)w, h);
    HIPCHK(c, hipMemsetAsync(c->d_s_mom, 0, sizeof(unsigned long long) * 6, c->stream));
    HIPCHK(c, ipk::launch_moments(c->d_s0, (size_t)w, w, h, c->d_s_mom, c->stream));
    unsigned long long m[6];
    HIPCHK(c, hipMemcpyAsync(m, c->d_s_mom, sizeof m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    finish_moments(m, *out);
    return IMGPROC_OK;
}

IMGPROC_API int imgproc_demosaic_rggb(imgproc_ctx *c, const uint8_t *src, size_t sstride, uint8_t *dst, size_t dstride,
                                      int w, int h)
{
    int rc = stage_io_check(c, src, dst, w, h, sstride, dstride, (size_t)w, (size_t)w * 3);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    UPLOAD(c, c->d_s0, src, sstride, (size_t)w, h);
    HIPCHK(c, ipk::launch_demosaic_rggb(c->d_s0, (size_t)w, c->d_s1, (size_t)w * 3, w, h, c->stream));
    DOWNLOAD(c, dst, dstride, c->d_s1, (size_t)w * 3, h);
    return IMGPROC_OK;
}

IMGPROC_API int imgproc_mask_and(imgproc_ctx *c, const uint8_t *bgr, size_t sstride, const uint8_t *mask, size_t mstride,
                                 uint8_t *dst, size_t dstride, int w, int h)
{
    if (!mask) return IMGPROC_EINVAL;
    const size_t rowb = (size_t)w * 3;
    int rc = stage_io_check(c, bgr, dst, w, h, sstride, dstride, rowb, rowb);
    if (rc) return rc;
    if (mstride < (size_t)w) return fail(c, IMGPROC_EINVAL, "mask stride too small");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    rc = ensure_work(c);
    if (rc) return rc;
    Slot &s = c->work;
    UPLOAD(c, c->d_s0, bgr, sstride, rowb, h);
    UPLOAD(c, s.d_m0, mask, mstride, (size_t)w, h);
    HIPCHK(c, ipk::launch_mask_and(c->d_s0, rowb, s.d_m0, (size_t)w, c->d_s1, rowb, w, h, c->stream));
    DOWNLOAD(c, dst, dstride, c->d_s1, rowb, h);
    return IMGPROC_OK;
}

// getStructuringElement (objectPolygonPath.cpp:31-32).  Host only.
IMGPROC_API int imgproc_structuring_element(int shape, int kw, int kh, uint8_t *elem)
{
    if (!elem || kw < 1 || kh < 1 || kw > IMGPROC_MAX_KSIZE || kh > IMGPROC_MAX_KSIZE) return IMGPROC_EINVAL;
    if (shape != IMGPROC_RECT && shape != IMGPROC_CROSS && shape != IMGPROC_ELLIPSE) return IMGPROC_EINVAL;
    const int cy = kh / 2, cx = kw / 2;
    memset(elem, 0, (size_t)kw * kh);
    for (int row = 0; row < kh; row++) {
        int from = 0, to = 0;  // [from, to)
        if (shape == IMGPROC_RECT) {
            to = kw;
        } else if (shape == IMGPROC_CROSS) {
            if (row == cy) to = kw;
            else { from = cx; to = cx + 1; }
        } else {
            const int dy = row - cy;
            if (cy == 0) {
                // degenerate one-row ellipse: OpenCV's inv_r2 = 0 -> half width 0
                from = cx; to = cx + 1;
            } else {
                const double half = cx * std::sqrt((double)(cy * cy - dy * dy) / ((double)cy * cy));
                const int dx = (int)lrint(half);
                from = cx - dx < 0 ? 0 : cx - dx;
                to = cx + dx + 1 > kw ? kw : cx + dx + 1;
            }
        }
        for (int i = from; i < to; i++) elem[(size_t)row * kw + i] = 1;
    }
    return IMGPROC_OK;
}

IMGPROC_API int imgproc_gaussian_taps(int ksize, int32_t *taps)
{
    if (!taps) return IMGPROC_EINVAL;
    int t[31];
    int rc = gaussian_taps(ksize, t);
    if (rc) return rc;
    for (int i = 0; i < ksize; i++) taps[i] = t[i];
    return IMGPROC_OK;
}

// ================================================================================================
// device memory helpers / measurement
// ================================================================================================
IMGPROC_API int imgproc_dev_alloc(imgproc_ctx *c, size_t bytes, void **dptr)
{
    if (!c || !dptr) return IMGPROC_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipMalloc(dptr, bytes));
    return IMGPROC_OK;
}
IMGPROC_API int imgproc_dev_free(imgproc_ctx *c, void *dptr)
{
    if (!c) return IMGPROC_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipFree(dptr));
    return IMGPROC_OK;
}
// page-locked host memory for imgproc_submit_pinned (callers without their own HIP allocator)
IMGPROC_API int imgproc_host_alloc(imgproc_ctx *c, size_t bytes, void **hptr)
{
    if (!c || !hptr) return IMGPROC_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return IMGPROC_OK;
}
IMGPROC_API int imgproc_host_free(imgproc_ctx *c, void *hptr)
{
    if (!c) return IMGPROC_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipHostFree(hptr));
    return IMGPROC_OK;
}
IMGPROC_API int imgproc_dev_upload(imgproc_ctx *c, void *dptr, const void *src, size_t bytes)
{
    if (!c || !dptr || !src) return IMGPROC_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipMemcpy(dptr, src, bytes, hipMemcpyHostToDevice));
    return IMGPROC_OK;
}
IMGPROC_API int imgproc_dev_download(imgproc_ctx *c, void *dst, const void *dptr, size_t bytes)
{
    if (!c || !dptr || !dst) return IMGPROC_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(dst, dptr, bytes, hipMemcpyDeviceToHost));
    return IMGPROC_OK;
}
IMGPROC_API int imgproc_set_profiling(imgproc_ctx *c, int on)
{
    if (!c) return IMGPROC_EINVAL;
    c->profiling = on != 0;
    c->last_ticket = -1;
    if (on) {  // time zero of imgproc_last_kernel_span
        HIPCHK(c, hipSetDevice(c->cfg.device));
        HIPCHK(c, hipEventRecord(c->ev_ref, c->stream));
        HIPCHK(c, hipEventSynchronize(c->ev_ref));
    }
    return IMGPROC_OK;
}
IMGPROC_API int imgproc_last_kernel_ms(imgproc_ctx *c, float *ms, int *launches)
{
    if (!c || !ms) return IMGPROC_EINVAL;
    if (c->last_ticket < 0 || !c->batch[c->last_ticket].timed) return fail(c, IMGPROC_ESTATE, "no profiled batch call yet");
    const imgproc_ctx::Batch &bt = c->batch[c->last_ticket];
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipEventSynchronize(bt.ev_k1));
    HIPCHK(c, hipEventElapsedTime(ms, bt.ev_k0, bt.ev_k1));
    if (launches) *launches = bt.launches;
    return IMGPROC_OK;
}
IMGPROC_API int imgproc_last_kernel_span(imgproc_ctx *c, float *t0_ms, float *t1_ms)
{
    if (!c || !t0_ms || !t1_ms) return IMGPROC_EINVAL;
    if (c->last_ticket < 0 || !c->batch[c->last_ticket].timed) return fail(c, IMGPROC_ESTATE, "no profiled batch call yet");
    const imgproc_ctx::Batch &bt = c->batch[c->last_ticket];
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipEventSynchronize(bt.ev_k1));
    HIPCHK(c, hipEventElapsedTime(t0_ms, c->ev_ref, bt.ev_k0));
    HIPCHK(c, hipEventElapsedTime(t1_ms, c->ev_ref, bt.ev_k1));
    return IMGPROC_OK;
}
IMGPROC_API int imgproc_synchronize(imgproc_ctx *c)
{
    if (!c) return IMGPROC_EINVAL;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (hipStream_t t : c->tstream)
        if (t) HIPCHK(c, hipStreamSynchronize(t));
    for (auto &s : c->slots) HIPCHK(c, hipStreamSynchronize(s.stream));
    return IMGPROC_OK;
}
