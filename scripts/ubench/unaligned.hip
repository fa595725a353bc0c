#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k(const uint8_t *src, uint32_t *out, uint8_t *dst, int off)
{
    const int t = threadIdx.x;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(src + off + 12 * t);  // deliberately misaligned
    const uint32_t a = p[0], b = p[1], c = p[2];
    out[3 * t] = a; out[3 * t + 1] = b; out[3 * t + 2] = c;
    *reinterpret_cast<uint32_t *>(dst + off + 4 * t) = a ^ 0x55AA55AAu;        // misaligned dword store
}
int main()
{
    const int N = 64, B = 12 * N + 16;
    uint8_t h[B], *d, *dd; uint32_t *o, ho[3 * N]; uint8_t hd[B];
    for (int i = 0; i < B; i++) h[i] = (uint8_t)(i * 7 + 3);
    hipMalloc(&d, B); hipMalloc(&dd, B); hipMalloc(&o, sizeof ho);
    hipMemcpy(d, h, B, hipMemcpyHostToDevice);
    int bad = 0;
    for (int off = 0; off < 4; off++) {
        hipMemset(dd, 0, B);
        k<<<1, N>>>(d, o, dd, off);
        hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
        hipMemcpy(hd, dd, B, hipMemcpyDeviceToHost);
        for (int t = 0; t < N; t++)
            for (int j = 0; j < 3; j++) {
                uint32_t e; memcpy(&e, h + off + 12 * t + 4 * j, 4);
                if (ho[3 * t + j] != e) bad++;
            }
        for (int t = 0; t < N; t++) {
            uint32_t e, g; memcpy(&e, h + off + 12 * t, 4); e ^= 0x55AA55AAu; memcpy(&g, hd + off + 4 * t, 4);
            if (g != e) bad++;
        }
        printf("off %d bad so far %d\n", off, bad);
    }
    return bad != 0;
}
