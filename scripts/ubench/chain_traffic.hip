// dev tool: the chain kernel's memory traffic without its arithmetic.  Same decomposition as ipf::k_fused_chain on the
// headline configuration (256 frames of 1920x1200 BGR8: one wave = a strip of 256 columns, 4 px = 12 bytes per lane and
// row, marching down a segment of SEG rows with 8 rows of look-ahead; 248 valid columns per strip), same loads
// (one 12-byte load per lane and row, PF rows ahead) and the same stores (one dword of mask bytes per lane and row);
// the "arithmetic" is K dependent-free VALU ops per row.  What rate does the memory system give this shape, with
// loads only, stores only, and both?
//   hipcc --offload-arch=gfx950 -O3 -o chain_traffic chain_traffic.hip && ./chain_traffic
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <bool LOAD, bool STORE, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_traffic(
    const unsigned char *frames, unsigned char *masks, int w, int h, int n_strips, int n_segs, int seg_rows, int total, int K, uint32_t *sink)
{
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= total) return;
    const int strip = item % n_strips, rest = item / n_strips, seg = rest % n_segs, frame = rest / n_segs;
    const int x0 = strip * 248 - 4 + lane * 4;
    const int xl = min(max(x0, 0), w - 4);
    const bool out = lane >= 1 && lane <= 62 && x0 >= 0 && x0 + 4 <= w;
    const unsigned char *f = frames + (size_t)frame * w * h * 3 + (size_t)xl * 3;
    unsigned char *m = masks + (size_t)frame * w * h + xl;
    const int ys = seg * seg_rows, ye = min(ys + seg_rows, h);
    uint32_t acc[8];
    for (int i = 0; i < 8; i++) acc[i] = lane + i;
    uint32_t r[4][3];
    auto ld = [&](int slot, int y) {
        y = min(max(y, 0), h - 1);
        const uint32_t *p = reinterpret_cast<const uint32_t *>(f + (size_t)y * w * 3);
        if (LOAD) { r[slot][0] = p[0]; r[slot][1] = p[1]; r[slot][2] = p[2]; }
    };
    for (int q = 0; q < 3; q++) ld(q, ys - 4 + q);
    for (int yb = ys - 4; yb < ye + 4; yb += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int y = yb + u;
            ld((u + 3) & 3, y + 3);
            if (LOAD) acc[0] ^= r[u][0] ^ r[u][1] ^ r[u][2];
            for (int k = 0; k < K; k += 8) {
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_alignbit(acc[i], acc[(i + 1) & 7], 7);
            }
            if (STORE && out && y >= ys && y < ye) *reinterpret_cast<uint32_t *>(m + (size_t)y * w) = acc[0];
        }
    }
    uint32_t t = 0;
    for (int i = 0; i < 8; i++) t ^= acc[i];
    if (t == 0x12345678u) sink[0] = t;
}

// the same with the mask stores of 8 rows gathered: lane (g = l >> 3, i = l & 7) stores the 32 bytes of word g of row i of
// the batch as two 16-byte stores (each instruction: 8 rows x 8 pieces of 16 bytes = 1 KB) -- the shape the chain kernel's
// tile takes after an 8x8 nibble transpose (its MODE 2 already does one)
template <bool LOAD, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_traffic16(
    const unsigned char *frames, unsigned char *masks, int w, int h, int n_strips, int n_segs, int seg_rows, int total, int K, uint32_t *sink)
{
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= total) return;
    const int strip = item % n_strips, rest = item / n_strips, seg = rest % n_segs, frame = rest / n_segs;
    const int x0 = strip * 248 - 4 + lane * 4;
    const int xl = min(max(x0, 0), w - 4);
    const unsigned char *f = frames + (size_t)frame * w * h * 3 + (size_t)xl * 3;
    const int g = lane >> 3, i8 = lane & 7;
    const int xs = strip * 248 + 32 * g;               // first pixel of this lane's word
    const int nb = min(min(32, 248 - 32 * g), w - xs);  // bytes of it inside the strip and the row (24 for g = 7)
    unsigned char *m = masks + (size_t)frame * w * h + min(xs, w - 32);
    const int ys = seg * seg_rows, ye = min(ys + seg_rows, h);
    uint32_t acc[8];
    for (int i = 0; i < 8; i++) acc[i] = lane + i;
    uint32_t r[4][3];
    auto ld = [&](int slot, int y) {
        y = min(max(y, 0), h - 1);
        const uint32_t *p = reinterpret_cast<const uint32_t *>(f + (size_t)y * w * 3);
        if (LOAD) { r[slot][0] = p[0]; r[slot][1] = p[1]; r[slot][2] = p[2]; }
    };
    for (int q = 0; q < 3; q++) ld(q, ys - 4 + q);
    for (int yb = ys - 4; yb < ye + 4; yb += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int y = yb + u;
            ld((u + 3) & 3, y + 3);
            if (LOAD) acc[0] ^= r[u & 3][0] ^ r[u & 3][1] ^ r[u & 3][2];
            for (int k = 0; k < K; k += 8) {
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_alignbit(acc[i], acc[(i + 1) & 7], 7);
            }
        }
        const int y = yb + i8;
        if (y >= ys && y < ye && nb > 0) {
            uint4 *o = reinterpret_cast<uint4 *>(m + (size_t)y * w);
            o[0] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
            if (nb > 16) {
                if (nb == 32) o[1] = make_uint4(acc[4], acc[5], acc[6], acc[7]);
                else *reinterpret_cast<uint2 *>(o + 1) = make_uint2(acc[4], acc[5]);
            }
        }
    }
    uint32_t t = 0;
    for (int i = 0; i < 8; i++) t ^= acc[i];
    if (t == 0x12345678u) sink[0] = t;
}

int main()
{
    const int w = 1920, h = 1200, n = 256;
    unsigned char *frames, *masks;
    uint32_t *sink;
    if (hipMalloc(&frames, (size_t)w * h * 3 * n) != hipSuccess || hipMalloc(&masks, (size_t)w * h * n) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(frames, 0x37, (size_t)w * h * 3 * n);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int n_strips = (w + 247) / 248;
    auto run = [&](const char *name, auto kernel, int seg_rows, int K, double bytes) {
        const int n_segs = (h + seg_rows - 1) / seg_rows, total = n_strips * n_segs * n;
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kernel, dim3((total + 3) / 4), dim3(256), 0, 0, frames, masks, w, h, n_strips, n_segs, seg_rows, total, K, sink);
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(kernel, dim3((total + 3) / 4), dim3(256), 0, 0, frames, masks, w, h, n_strips, n_segs, seg_rows, total, K, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s seg %3d valu/row %3d  %7.1f us  %5.2f TB/s of algorithmic bytes\n", name, seg_rows, K, ms / 10 * 1e3, bytes / (ms / 10 * 1e-3) / 1e12);
        fflush(stdout);
    };
    const double px = (double)w * h * n;
    for (int K : {0, 64, 128, 168}) {
        run("loads + stores, 6 waves/SIMD", k_traffic<true, true, 6>, 100, K, px * 4);
        run("loads only", k_traffic<true, false, 6>, 100, K, px * 3);
        run("stores only", k_traffic<false, true, 6>, 100, K, px * 1);
        run("no memory", k_traffic<false, false, 6>, 100, K, px * 4);
    }
    for (int K : {0, 64, 128, 168}) {
        run("loads + 16-byte stores of 8 rows, 6 waves/SIMD", k_traffic16<true, 6>, 100, K, px * 4);
        run("16-byte stores of 8 rows only", k_traffic16<false, 6>, 100, K, px * 1);
    }
    for (int K : {128, 168}) {  // how well do arithmetic and traffic overlap at 4 / 6 / 8 waves per SIMD?
        run("loads + stores, 4 waves/SIMD", k_traffic<true, true, 4>, 100, K, px * 4);
        run("loads + stores, 8 waves/SIMD", k_traffic<true, true, 8>, 100, K, px * 4);
        run("no memory, 8 waves/SIMD", k_traffic<false, false, 8>, 100, K, px * 4);
    }
    run("loads + stores, 8 waves/SIMD", k_traffic<true, true, 8>, 100, 0, px * 4);
    run("loads + stores, 6 waves, 200-row segments", k_traffic<true, true, 6>, 200, 0, px * 4);
    run("loads + stores, 6 waves, 50-row segments", k_traffic<true, true, 6>, 50, 0, px * 4);
    return 0;
}
