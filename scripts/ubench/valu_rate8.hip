// Micro-benchmark (dev tool, not part of the library): issue rate of the integer VALU ops the fused
// kernel is made of, on gfx950.  Each kernel runs ITER x 16 independent ops per wave; 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITER 2048

#define BODY16(OPSTR)                                                                        \
    asm volatile(OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7) \
                 OPSTR(8) OPSTR(9) OPSTR(10) OPSTR(11) OPSTR(12) OPSTR(13) OPSTR(14) OPSTR(15) \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) \
                 : "v"(x), "v"(y), "s"(sm) : "vcc")

#define KERNEL(NAME, OPSTR)                                                     \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)   \
    {                                                                           \
        uint32_t r[16];                                                         \
        for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;        \
        uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;                 \
        unsigned long long sm = 0x1234567887654321ull ^ seed;                   \
        for (int it = 0; it < ITER; it++) { BODY16(OPSTR); }                    \
        uint32_t a = 0;                                                         \
        for (int i = 0; i < 16; i++) a ^= r[i];                                 \
        out[blockIdx.x * 256 + threadIdx.x] = a;                                \
    }

#define S_(x) #x
#define OP_ADD(i) "v_add_u32 %" S_(i) ", %" S_(i) ", %16\n"
#define OP_SUB(i) "v_sub_u32 %" S_(i) ", %" S_(i) ", %16\n"
#define OP_ADD3(i) "v_add3_u32 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OP_LSHLADD(i) "v_lshl_add_u32 %" S_(i) ", %" S_(i) ", 2, %16\n"
#define OP_ADDLSHL(i) "v_add_lshl_u32 %" S_(i) ", %" S_(i) ", %16, 2\n"
#define OP_MADU24(i) "v_mad_u32_u24 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OP_MADI24(i) "v_mad_i32_i24 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OP_MULU24(i) "v_mul_u32_u24 %" S_(i) ", %" S_(i) ", %16\n"
#define OP_MULLO(i) "v_mul_lo_u32 %" S_(i) ", %" S_(i) ", %16\n"
#define OP_AND(i) "v_and_b32 %" S_(i) ", %" S_(i) ", %16\n"
#define OP_ANDOR(i) "v_and_or_b32 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OP_MAX(i) "v_max_u32 %" S_(i) ", %" S_(i) ", %16\n"
#define OP_MIN3(i) "v_min3_u32 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OP_ASHR(i) "v_ashrrev_i32 %" S_(i) ", 14, %" S_(i) "\n"
#define OP_CNDMASK(i) "v_cndmask_b32 %" S_(i) ", %" S_(i) ", %16, %18\n"
#define OP_CMP(i) "v_cmp_eq_u32 vcc, %" S_(i) ", %16\n"
#define OP_CMPSDWA(i) "v_cmp_ge_u32_sdwa vcc, %" S_(i) ", %16 src0_sel:WORD_1 src1_sel:DWORD\n"
#define OP_ADDC(i) "v_addc_co_u32 %" S_(i) ", vcc, %" S_(i) ", %" S_(i) ", %18\n"
#define OP_SDWA(i) "v_lshlrev_b32_sdwa %" S_(i) ", %17, %" S_(i) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define OP_DPP(i) "v_mov_b32_dpp %" S_(i) ", %" S_(i) " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define OP_DPPROW(i) "v_mov_b32_dpp %" S_(i) ", %" S_(i) " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define OP_ADDDPP(i) "v_add_u32_dpp %" S_(i) ", %" S_(i) ", %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define OP_PKMAX(i) "v_pk_max_u16 %" S_(i) ", %" S_(i) ", %16\n"
#define OP_PKADD(i) "v_pk_add_u16 %" S_(i) ", %" S_(i) ", %16\n"
#define OP_PKMAD(i) "v_pk_mad_u16 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OP_PKMUL(i) "v_pk_mul_lo_u16 %" S_(i) ", %" S_(i) ", %16\n"
#define OP_PKLSHL(i) "v_pk_lshlrev_b16 %" S_(i) ", 2, %" S_(i) "\n"
#define OP_PKSUB(i) "v_pk_sub_i16 %" S_(i) ", %" S_(i) ", %16\n"
#define OP_ALIGNBIT(i) "v_alignbit_b32 %" S_(i) ", %" S_(i) ", %16, 16\n"
#define OP_PERM(i) "v_perm_b32 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OP_BFE(i) "v_bfe_u32 %" S_(i) ", %" S_(i) ", 8, 8\n"
#define OP_BCNT(i) "v_bcnt_u32_b32 %" S_(i) ", %" S_(i) ", %16\n"
#define OP_DOT4(i) "v_dot4_u32_u8 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OP_FMA(i) "v_fma_f32 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OP_PKFMA(i) "v_pk_fma_f32 %0, %0, %0, %0\n"
#define OP_SAD(i) "v_sad_u8 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OP_MOV(i) "v_mov_b32 %" S_(i) ", %16\n"
#define OP_BFI(i) "v_bfi_b32 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OP_MADU16(i) "v_mad_u16 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OP_CVTPK(i) "v_cvt_pk_u8_f32 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OP_MADU64(i) "v_mul_hi_u32 %" S_(i) ", %" S_(i) ", %16\n"


__global__ __launch_bounds__(256) void k_addmax_16a16b(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (0) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 4;
        for (int q = 0; q < lead; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 1; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 1; q++) { asm volatile(OP_MAX(0) OP_MAX(1) OP_MAX(2) OP_MAX(3) OP_MAX(4) OP_MAX(5) OP_MAX(6) OP_MAX(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_addmax_16a16b_desync(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (1) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 4;
        for (int q = 0; q < lead; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 1; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 1; q++) { asm volatile(OP_MAX(0) OP_MAX(1) OP_MAX(2) OP_MAX(3) OP_MAX(4) OP_MAX(5) OP_MAX(6) OP_MAX(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_addmax_64a64b(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (0) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 16;
        for (int q = 0; q < lead; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 4; q++) { asm volatile(OP_MAX(0) OP_MAX(1) OP_MAX(2) OP_MAX(3) OP_MAX(4) OP_MAX(5) OP_MAX(6) OP_MAX(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_addmax_64a64b_desync(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (1) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 16;
        for (int q = 0; q < lead; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 4; q++) { asm volatile(OP_MAX(0) OP_MAX(1) OP_MAX(2) OP_MAX(3) OP_MAX(4) OP_MAX(5) OP_MAX(6) OP_MAX(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_addmax_256a256b(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (0) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 64;
        for (int q = 0; q < lead; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 16; q++) { asm volatile(OP_MAX(0) OP_MAX(1) OP_MAX(2) OP_MAX(3) OP_MAX(4) OP_MAX(5) OP_MAX(6) OP_MAX(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_addmax_256a256b_desync(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (1) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 64;
        for (int q = 0; q < lead; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 16; q++) { asm volatile(OP_MAX(0) OP_MAX(1) OP_MAX(2) OP_MAX(3) OP_MAX(4) OP_MAX(5) OP_MAX(6) OP_MAX(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_addmax_64a128b(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (0) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 24;
        for (int q = 0; q < lead; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 8; q++) { asm volatile(OP_MAX(0) OP_MAX(1) OP_MAX(2) OP_MAX(3) OP_MAX(4) OP_MAX(5) OP_MAX(6) OP_MAX(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_addmax_64a128b_desync(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (1) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 24;
        for (int q = 0; q < lead; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 8; q++) { asm volatile(OP_MAX(0) OP_MAX(1) OP_MAX(2) OP_MAX(3) OP_MAX(4) OP_MAX(5) OP_MAX(6) OP_MAX(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_addmax_256a512b(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (0) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 96;
        for (int q = 0; q < lead; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 32; q++) { asm volatile(OP_MAX(0) OP_MAX(1) OP_MAX(2) OP_MAX(3) OP_MAX(4) OP_MAX(5) OP_MAX(6) OP_MAX(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_addmax_256a512b_desync(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (1) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 96;
        for (int q = 0; q < lead; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 32; q++) { asm volatile(OP_MAX(0) OP_MAX(1) OP_MAX(2) OP_MAX(3) OP_MAX(4) OP_MAX(5) OP_MAX(6) OP_MAX(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_addmax_128a16b(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (0) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 18;
        for (int q = 0; q < lead; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 1; q++) { asm volatile(OP_MAX(0) OP_MAX(1) OP_MAX(2) OP_MAX(3) OP_MAX(4) OP_MAX(5) OP_MAX(6) OP_MAX(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_addmax_128a16b_desync(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (1) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 18;
        for (int q = 0; q < lead; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) { asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 1; q++) { asm volatile(OP_MAX(0) OP_MAX(1) OP_MAX(2) OP_MAX(3) OP_MAX(4) OP_MAX(5) OP_MAX(6) OP_MAX(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_andmad_16a16b(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (0) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 4;
        for (int q = 0; q < lead; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 1; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 1; q++) { asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_andmad_16a16b_desync(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (1) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 4;
        for (int q = 0; q < lead; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 1; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 1; q++) { asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_andmad_64a64b(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (0) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 16;
        for (int q = 0; q < lead; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 4; q++) { asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_andmad_64a64b_desync(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (1) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 16;
        for (int q = 0; q < lead; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 4; q++) { asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_andmad_256a256b(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (0) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 64;
        for (int q = 0; q < lead; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 16; q++) { asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_andmad_256a256b_desync(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (1) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 64;
        for (int q = 0; q < lead; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 16; q++) { asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_andmad_64a128b(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (0) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 24;
        for (int q = 0; q < lead; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 8; q++) { asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_andmad_64a128b_desync(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (1) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 24;
        for (int q = 0; q < lead; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 8; q++) { asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_andmad_256a512b(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (0) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 96;
        for (int q = 0; q < lead; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 32; q++) { asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_andmad_256a512b_desync(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (1) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 96;
        for (int q = 0; q < lead; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 16; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 32; q++) { asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_andmad_128a16b(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (0) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 18;
        for (int q = 0; q < lead; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 1; q++) { asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_andmad_128a16b_desync(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    if (1) {  // de-synchronise the waves of a SIMD: a wave-dependent number of leading A blocks
        const int lead = ((threadIdx.x >> 6) * 5 + blockIdx.x * 3) % 18;
        for (int q = 0; q < lead; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) { asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
#pragma unroll
        for (int q = 0; q < 1; q++) { asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc"); }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; i++) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef void (*kfn)(uint32_t *, uint32_t, int);
struct Ent { const char *name; kfn f; int blocks16; };
int main()
{
    Ent ents[] = {{"addmax_16a16b", k_addmax_16a16b, 2}, {"addmax_16a16b_desync", k_addmax_16a16b_desync, 2}, {"addmax_64a64b", k_addmax_64a64b, 8}, {"addmax_64a64b_desync", k_addmax_64a64b_desync, 8}, {"addmax_256a256b", k_addmax_256a256b, 32}, {"addmax_256a256b_desync", k_addmax_256a256b_desync, 32}, {"addmax_64a128b", k_addmax_64a128b, 12}, {"addmax_64a128b_desync", k_addmax_64a128b_desync, 12}, {"addmax_256a512b", k_addmax_256a512b, 48}, {"addmax_256a512b_desync", k_addmax_256a512b_desync, 48}, {"addmax_128a16b", k_addmax_128a16b, 9}, {"addmax_128a16b_desync", k_addmax_128a16b_desync, 9}, {"andmad_16a16b", k_andmad_16a16b, 2}, {"andmad_16a16b_desync", k_andmad_16a16b_desync, 2}, {"andmad_64a64b", k_andmad_64a64b, 8}, {"andmad_64a64b_desync", k_andmad_64a64b_desync, 8}, {"andmad_256a256b", k_andmad_256a256b, 32}, {"andmad_256a256b_desync", k_andmad_256a256b_desync, 32}, {"andmad_64a128b", k_andmad_64a128b, 12}, {"andmad_64a128b_desync", k_andmad_64a128b_desync, 12}, {"andmad_256a512b", k_andmad_256a512b, 48}, {"andmad_256a512b_desync", k_andmad_256a512b_desync, 48}, {"andmad_128a16b", k_andmad_128a16b, 9}, {"andmad_128a16b_desync", k_andmad_128a16b_desync, 9}};
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 8;
    uint32_t *d;
    (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (auto &en : ents) {
        const int iters = 2048 / en.blocks16 + 1;
        en.f<<<blocks, 256>>>(d, 1, iters);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int rep = 0; rep < 5; rep++) en.f<<<blocks, 256>>>(d, rep, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %.2f cyc/instr (2.4 GHz)\n", en.name, ms * 1e-3 * 2.4e9 / (5.0 * 8 * iters * en.blocks16 * 16));
        fflush(stdout);
    }
    return 0;
}
