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

int main()
{
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
