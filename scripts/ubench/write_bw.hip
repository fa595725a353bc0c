// dev tool: what rate do plain global stores reach on this chip, for the store shapes of the mask-writing kernels?
//   linear : every wave streams 1 KB pieces of one contiguous region (dwordx4 per lane)
//   rows   : the strip kernel's shape -- a wave owns ~2 KB of a 7680-byte row and walks down 48 rows
//   nt     : linear with non-temporal stores
// build: hipcc --offload-arch=gfx950 -O3 -o write_bw write_bw.hip ; run: ./write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(256) void k_linear(uint4 *dst, size_t n16, int per_wave)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint4 v = make_uint4(lane, 1, 2, 3);
    size_t i = wave * per_wave * 64 + lane;
    for (int k = 0; k < per_wave; k++, i += 64)
        if (i < n16) dst[i] = v;
}
__global__ __launch_bounds__(256) void k_linear_nt(uint4 *dst, size_t n16, int per_wave)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint4 v = make_uint4(lane, 1, 2, 3);
    size_t i = wave * per_wave * 64 + lane;
    for (int k = 0; k < per_wave; k++, i += 64)
        if (i < n16) {
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            v4u q = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(q, reinterpret_cast<v4u *>(dst + i));
        }
}
// planes x h rows of w bytes; wave = (plane, seg, strip): strip = 2 KB of a row (w / 2048 strips, the last one short)
__global__ __launch_bounds__(256) void k_rows(unsigned char *dst, int w, int h, int planes, int seg_rows, int strips)
{
    const int lane = threadIdx.x & 63;
    int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int strip = item % strips; item /= strips;
    const int n_segs = (h + seg_rows - 1) / seg_rows;
    const int seg = item % n_segs, plane = item / n_segs;
    if (plane >= planes) return;
    const uint4 v = make_uint4(lane, 1, 2, 3);
    for (int y = seg * seg_rows; y < min(h, (seg + 1) * seg_rows); y++) {
        unsigned char *row = dst + ((size_t)plane * h + y) * w + strip * 2048;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int x = strip * 2048 + half * 1024 + lane * 16;
            if (x < w) *reinterpret_cast<uint4 *>(row + half * 1024 + lane * 16) = v;
        }
    }
}

// rows + K dependent VALU ops per row and half (does the store stream overlap with a wave's own arithmetic?)
template <bool STORE>
__global__ __launch_bounds__(256) void k_rows_valu(unsigned char *dst, int w, int h, int planes, int seg_rows, int strips, int K, uint32_t *sink)
{
    const int lane = threadIdx.x & 63;
    int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int strip = item % strips; item /= strips;
    const int n_segs = (h + seg_rows - 1) / seg_rows;
    const int seg = item % n_segs, plane = item / n_segs;
    if (plane >= planes) return;
    uint32_t acc = lane * 2654435761u;
    for (int y = seg * seg_rows; y < min(h, (seg + 1) * seg_rows); y++) {
        unsigned char *row = dst + ((size_t)plane * h + y) * w + strip * 2048;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            for (int k = 0; k < K; k++) acc = __builtin_amdgcn_alignbit(acc, acc ^ (uint32_t)y, 7) + 0x9E3779B9u;
            const int x = strip * 2048 + half * 1024 + lane * 16;
            if (STORE) {
                if (x < w) *reinterpret_cast<uint4 *>(row + half * 1024 + lane * 16) = make_uint4(acc, acc + 1, acc + 2, acc + 3);
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// round 3: the strip kernel's loop more closely -- WAVES waves per SIMD, K independent VALU ops per row (8 accumulators:
// issue-bound, not latency-bound), optionally a 4-byte-per-lane load three rows ahead (LOAD) and the mask expansion's LDS
// traffic (one ds_bpermute + 4 table reads per half, LDSOPS); two 1 KB stores per row (STORE)
template <int WAVES, bool STORE, bool LOAD, bool LDSOPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_rows_like(
    unsigned char *dst, const uint32_t *bits, int w, int h, int planes, int seg_rows, int strips, int K, uint32_t *sink)
{
    __shared__ uint32_t s_lut[16];
    if (threadIdx.x < 16) s_lut[threadIdx.x] = threadIdx.x * 0x01010101u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int strip = item % strips; item /= strips;
    const int n_segs = (h + seg_rows - 1) / seg_rows;
    const int seg = item % n_segs, plane = item / n_segs;
    if (plane >= planes) return;
    uint32_t acc[8];
    for (int i = 0; i < 8; i++) acc[i] = lane * 2654435761u + i;
    const int words = w / 32;
    const uint32_t *src = bits + (size_t)plane * h * words + min(strip * 64 + lane, words - 1);
    const int y0 = seg * seg_rows, y1 = min(h, (seg + 1) * seg_rows);
    uint32_t n0 = 0, n1 = 0, n2 = 0;
    if (LOAD) { n0 = src[(size_t)y0 * words]; n1 = src[(size_t)min(y0 + 1, h - 1) * words]; n2 = src[(size_t)min(y0 + 2, h - 1) * words]; }
    for (int y = y0; y < y1; y++) {
        uint32_t in = n0;
        if (LOAD) { n0 = n1; n1 = n2; n2 = src[(size_t)min(y + 3, h - 1) * words]; }
        acc[0] ^= in;
        for (int k = 0; k < K; k += 8) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_alignbit(acc[i], acc[(i + 1) & 7], 7);
        }
        unsigned char *row = dst + ((size_t)plane * h + y) * w + strip * 2048;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint32_t o[4] = {acc[0], acc[1], acc[2], acc[3]};
            if (LDSOPS) {
                const uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute((32 * half + (lane >> 1)) * 4, (int)acc[half]);
#pragma unroll
                for (int q = 0; q < 4; q++) o[q] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_lut) + ((word >> (4 * q)) & 60u));
            }
            const int x = strip * 2048 + half * 1024 + lane * 16;
            if (STORE && x < w) *reinterpret_cast<uint4 *>(row + half * 1024 + lane * 16) = make_uint4(o[0], o[1], o[2], o[3]);
            if (!STORE) acc[4 + half] ^= o[0] ^ o[1] ^ o[2] ^ o[3];
        }
    }
    uint32_t t = 0;
    for (int i = 0; i < 8; i++) t ^= acc[i];
    if (t == 0x12345678u) sink[0] = t;
}

// the same with the loads in blocks: every D rows the D words of the NEXT block are requested at once, so the wait for
// them (vmcnt counts loads and stores together, in issue order) leaves the 2 D stores of the current block in flight
template <int WAVES, int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_rows_blockload(
    unsigned char *dst, const uint32_t *bits, int w, int h, int planes, int seg_rows, int strips, int K, uint32_t *sink)
{
    const int lane = threadIdx.x & 63;
    int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int strip = item % strips; item /= strips;
    const int n_segs = (h + seg_rows - 1) / seg_rows;
    const int seg = item % n_segs, plane = item / n_segs;
    if (plane >= planes) return;
    uint32_t acc[8];
    for (int i = 0; i < 8; i++) acc[i] = lane * 2654435761u + i;
    const int words = w / 32;
    const uint32_t *src = bits + (size_t)plane * h * words + min(strip * 64 + lane, words - 1);
    const int y0 = seg * seg_rows, y1 = min(h, (seg + 1) * seg_rows);
    uint32_t nxt[D], cur[D];
#pragma unroll
    for (int i = 0; i < D; i++) nxt[i] = src[(size_t)min(y0 + i, h - 1) * words];
    for (int yb = y0; yb < y1; yb += D) {
#pragma unroll
        for (int i = 0; i < D; i++) cur[i] = nxt[i];
#pragma unroll
        for (int i = 0; i < D; i++) nxt[i] = src[(size_t)min(yb + D + i, h - 1) * words];
#pragma unroll
        for (int i = 0; i < D; i++) {
            const int y = yb + i;
            acc[0] ^= cur[i];
            for (int k = 0; k < K; k += 8) {
#pragma unroll
                for (int q = 0; q < 8; q++) acc[q] = __builtin_amdgcn_alignbit(acc[q], acc[(q + 1) & 7], 7);
            }
            if (y < y1) {
                unsigned char *row = dst + ((size_t)plane * h + y) * w + strip * 2048;
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int x = strip * 2048 + half * 1024 + lane * 16;
                    if (x < w) *reinterpret_cast<uint4 *>(row + half * 1024 + lane * 16) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
                }
            }
        }
    }
    uint32_t t = 0;
    for (int i = 0; i < 8; i++) t ^= acc[i];
    if (t == 0x12345678u) sink[0] = t;
}

// what about the load hurts?  KIND 1: the plane words as the strip kernel reads them; 2: the same 256 bytes every row (cache
// hits: only the vmcnt coupling is left); 3: a scalar load per row (lgkmcnt, no vector-memory load at all); 4: as 1 but the
// loads marked non-temporal; 5: as 1, but ONE lane-group loads for the wave: lanes 0..15 fetch 16 bytes each (same 256 bytes)
template <int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_rows_loadkind(
    unsigned char *dst, const uint32_t *bits, int w, int h, int planes, int seg_rows, int strips, int K, uint32_t *sink)
{
    const int lane = threadIdx.x & 63;
    int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int strip = item % strips; item /= strips;
    const int n_segs = (h + seg_rows - 1) / seg_rows;
    const int seg = item % n_segs, plane = item / n_segs;
    if (plane >= planes) return;
    uint32_t acc[8];
    for (int i = 0; i < 8; i++) acc[i] = lane * 2654435761u + i;
    const int words = w / 32;
    const uint32_t *src = bits + (size_t)plane * h * words + min(strip * 64 + lane, words - 1);
    const uint32_t *usrc = bits + (size_t)plane * h * words + strip * 64;  // uniform
    const int y0 = seg * seg_rows, y1 = min(h, (seg + 1) * seg_rows);
    auto ld = [&](int y) -> uint32_t {
        y = min(y, h - 1);
        if (KIND == 1) return src[(size_t)y * words];
        if (KIND == 2) return src[(size_t)y0 * words];
        if (KIND == 3) return usrc[(size_t)y * words];  // uniform address: the compiler emits s_load_dword
        if (KIND == 4) return __builtin_nontemporal_load(src + (size_t)y * words);
        const uint4 v = *reinterpret_cast<const uint4 *>(usrc + (size_t)y * words + 4 * (lane & 15));
        return v.x ^ v.y ^ v.z ^ v.w;
    };
    uint32_t n0 = ld(y0), n1 = ld(y0 + 1), n2 = ld(y0 + 2);
    for (int y = y0; y < y1; y++) {
        const uint32_t in = n0;
        n0 = n1; n1 = n2; n2 = ld(y + 3);
        acc[0] ^= in;
        for (int k = 0; k < K; k += 8) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_alignbit(acc[i], acc[(i + 1) & 7], 7);
        }
        unsigned char *row = dst + ((size_t)plane * h + y) * w + strip * 2048;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int x = strip * 2048 + half * 1024 + lane * 16;
            if (x < w) *reinterpret_cast<uint4 *>(row + half * 1024 + lane * 16) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
    uint32_t t = 0;
    for (int i = 0; i < 8; i++) t ^= acc[i];
    if (t == 0x12345678u) sink[0] = t;
}

int main(int argc, char **argv)
{
    if (argc > 1) {  // ./write_bw like : the strip-kernel-like sweep only
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int w = 7680, h = 4320, planes = 32;
        const size_t bytes = (size_t)w * h * planes;
        unsigned char *d; uint32_t *bits, *sink;
        if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&bits, bytes / 8) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
        hipMemset(bits, 0x5a, bytes / 8);
        auto time = [&](const char *name, auto launch) {
            for (int i = 0; i < 2; i++) launch();
            hipEventRecord(e0);
            for (int i = 0; i < 10; i++) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-56s %8.1f us  %6.2f TB/s\n", name, ms / 10 * 1e3, bytes / (ms / 10 * 1e-3) / 1e12);
            fflush(stdout);
        };
        const int seg = 64, strips = (w + 2047) / 2048, n_segs = (h + seg - 1) / seg, items = strips * n_segs * planes;
        const dim3 grid((items + 3) / 4), block(256);
#define LIKE(WV, ST, LD, LO, K) { char nm[96]; snprintf(nm, sizeof nm, "waves %d store %d load %d lds %d valu %d", WV, ST, LD, LO, K); \
        time(nm, [&] { hipLaunchKernelGGL((k_rows_like<WV, ST, LD, LO>), grid, block, 0, 0, d, bits, w, h, planes, seg, strips, K, sink); }); }
#define KIND(KD, K) { char nm[96]; snprintf(nm, sizeof nm, "load kind %d, valu %d", KD, K); \
        time(nm, [&] { hipLaunchKernelGGL((k_rows_loadkind<KD>), grid, block, 0, 0, d, bits, w, h, planes, seg, strips, K, sink); }); }
        for (int K : {0, 128}) { KIND(1, K) KIND(2, K) KIND(3, K) KIND(4, K) KIND(5, K) }
        if (argv[1][0] == 'k') return 0;
#define BLK(WV, D, K) { char nm[96]; snprintf(nm, sizeof nm, "waves %d loads in blocks of %d rows, valu %d", WV, D, K); \
        time(nm, [&] { hipLaunchKernelGGL((k_rows_blockload<WV, D>), grid, block, 0, 0, d, bits, w, h, planes, seg, strips, K, sink); }); }
        for (int K : {0, 128, 176}) { BLK(5, 4, K) BLK(5, 8, K) BLK(5, 16, K) BLK(5, 32, K) BLK(4, 8, K) BLK(4, 16, K) BLK(3, 16, K) BLK(3, 32, K) }
        if (argv[1][0] == 'b') return 0;
        for (int K : {0, 64, 128, 176, 256}) {
            LIKE(8, true, false, false, K) LIKE(8, false, false, false, K)
            LIKE(5, true, false, false, K) LIKE(5, false, false, false, K)
            LIKE(5, true, true, false, K) LIKE(5, true, false, true, K)
            LIKE(5, true, true, true, K) LIKE(5, false, true, true, K)
            LIKE(3, true, true, true, K)
        }
        return 0;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // planes = 8: one 8K frame's masks (265 MB, about the size of the Infinity Cache); 32 and 128: far beyond it
    for (int planes : {8, 32, 128}) {
        const int w = 7680, h = 4320;
        const size_t bytes = (size_t)w * h * planes;
        unsigned char *d;
        if (hipMalloc(&d, bytes) != hipSuccess) return 1;
        printf("---- %.2f GB ----\n", bytes / 1e9);
        auto time = [&](const char *name, auto launch) {
            for (int i = 0; i < 2; i++) launch();
            hipEventRecord(e0);
            const int reps = planes > 32 ? 5 : 20;
            for (int i = 0; i < reps; i++) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-28s %8.1f us  %6.2f TB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
        };
        const size_t n16 = bytes / 16;
        for (int per_wave : {4, 96}) {
            const size_t waves = (n16 / 64 + per_wave - 1) / per_wave;
            char nm[64];
            snprintf(nm, sizeof nm, "linear, %d KB per wave", per_wave);
            time(nm, [&] { hipLaunchKernelGGL(k_linear, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, (uint4 *)d, n16, per_wave); });
            snprintf(nm, sizeof nm, "linear nt, %d KB per wave", per_wave);
            time(nm, [&] { hipLaunchKernelGGL(k_linear_nt, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, (uint4 *)d, n16, per_wave); });
        }
        for (int seg : {16, 48, 128}) {
            const int strips = (w + 2047) / 2048, n_segs = (h + seg - 1) / seg;
            const int items = strips * n_segs * planes;
            char nm[64];
            snprintf(nm, sizeof nm, "rows, %d rows per wave", seg);
            time(nm, [&] { hipLaunchKernelGGL(k_rows, dim3((items + 3) / 4), dim3(256), 0, 0, d, w, h, planes, seg, strips); });
        }
        if (planes == 32) {
            uint32_t *sink;
            hipMalloc(&sink, 64);
            for (int K : {0, 16, 32, 64, 128}) {
                const int seg = 135;
                const int strips = (w + 2047) / 2048, n_segs = (h + seg - 1) / seg;
                const int items = strips * n_segs * planes;
                char nm[64];
                snprintf(nm, sizeof nm, "rows+%d valu per half, store", 2 * K);
                time(nm, [&] { hipLaunchKernelGGL(k_rows_valu<true>, dim3((items + 3) / 4), dim3(256), 0, 0, d, w, h, planes, seg, strips, K, sink); });
                snprintf(nm, sizeof nm, "rows+%d valu per half, none", 2 * K);
                time(nm, [&] { hipLaunchKernelGGL(k_rows_valu<false>, dim3((items + 3) / 4), dim3(256), 0, 0, d, w, h, planes, seg, strips, K, sink); });
            }
            hipFree(sink);
        }
        hipFree(d);
    }
    return 0;
}
