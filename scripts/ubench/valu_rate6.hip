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


__global__ __launch_bounds__(256) void k_add_max_8a8b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_max_12a4b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_max_14a2b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_max_15a1b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_max_alt(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_MAX(1) OP_ADD(2) OP_MAX(3) OP_ADD(4) OP_MAX(5) OP_ADD(6) OP_MAX(7) OP_ADD(8) OP_MAX(9) OP_ADD(10) OP_MAX(11) OP_ADD(12) OP_MAX(13) OP_ADD(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_max_3a1b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_MAX(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_MAX(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_MAX(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_max_16a(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_max_16b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_MAX(0) OP_MAX(1) OP_MAX(2) OP_MAX(3) OP_MAX(4) OP_MAX(5) OP_MAX(6) OP_MAX(7) OP_MAX(8) OP_MAX(9) OP_MAX(10) OP_MAX(11) OP_MAX(12) OP_MAX(13) OP_MAX(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_mad_8a8b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_mad_12a4b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_mad_14a2b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_mad_15a1b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_mad_alt(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_MADU24(1) OP_ADD(2) OP_MADU24(3) OP_ADD(4) OP_MADU24(5) OP_ADD(6) OP_MADU24(7) OP_ADD(8) OP_MADU24(9) OP_ADD(10) OP_MADU24(11) OP_ADD(12) OP_MADU24(13) OP_ADD(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_mad_3a1b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_MADU24(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_MADU24(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_MADU24(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_mad_16a(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_mad_16b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_cnd_8a8b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_CNDMASK(8) OP_CNDMASK(9) OP_CNDMASK(10) OP_CNDMASK(11) OP_CNDMASK(12) OP_CNDMASK(13) OP_CNDMASK(14) OP_CNDMASK(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_cnd_12a4b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_CNDMASK(12) OP_CNDMASK(13) OP_CNDMASK(14) OP_CNDMASK(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_cnd_14a2b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_CNDMASK(14) OP_CNDMASK(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_cnd_15a1b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_CNDMASK(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_cnd_alt(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_CNDMASK(1) OP_AND(2) OP_CNDMASK(3) OP_AND(4) OP_CNDMASK(5) OP_AND(6) OP_CNDMASK(7) OP_AND(8) OP_CNDMASK(9) OP_AND(10) OP_CNDMASK(11) OP_AND(12) OP_CNDMASK(13) OP_AND(14) OP_CNDMASK(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_cnd_3a1b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_CNDMASK(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_CNDMASK(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_CNDMASK(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_CNDMASK(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_cnd_16a(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_AND(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_AND(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_AND(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_cnd_16b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_CNDMASK(0) OP_CNDMASK(1) OP_CNDMASK(2) OP_CNDMASK(3) OP_CNDMASK(4) OP_CNDMASK(5) OP_CNDMASK(6) OP_CNDMASK(7) OP_CNDMASK(8) OP_CNDMASK(9) OP_CNDMASK(10) OP_CNDMASK(11) OP_CNDMASK(12) OP_CNDMASK(13) OP_CNDMASK(14) OP_CNDMASK(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_add3_8a8b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD3(8) OP_ADD3(9) OP_ADD3(10) OP_ADD3(11) OP_ADD3(12) OP_ADD3(13) OP_ADD3(14) OP_ADD3(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_add3_12a4b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD3(12) OP_ADD3(13) OP_ADD3(14) OP_ADD3(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_add3_14a2b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD3(14) OP_ADD3(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_add3_15a1b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD3(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_add3_alt(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD3(1) OP_ADD(2) OP_ADD3(3) OP_ADD(4) OP_ADD3(5) OP_ADD(6) OP_ADD3(7) OP_ADD(8) OP_ADD3(9) OP_ADD(10) OP_ADD3(11) OP_ADD(12) OP_ADD3(13) OP_ADD(14) OP_ADD3(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_add3_3a1b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD3(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD3(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD3(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD3(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_add3_16a(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_add3_16b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD3(0) OP_ADD3(1) OP_ADD3(2) OP_ADD3(3) OP_ADD3(4) OP_ADD3(5) OP_ADD3(6) OP_ADD3(7) OP_ADD3(8) OP_ADD3(9) OP_ADD3(10) OP_ADD3(11) OP_ADD3(12) OP_ADD3(13) OP_ADD3(14) OP_ADD3(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_dpp_8a8b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_DPP(8) OP_DPP(9) OP_DPP(10) OP_DPP(11) OP_DPP(12) OP_DPP(13) OP_DPP(14) OP_DPP(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_dpp_12a4b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_DPP(12) OP_DPP(13) OP_DPP(14) OP_DPP(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_dpp_14a2b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_DPP(14) OP_DPP(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_dpp_15a1b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_DPP(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_dpp_alt(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_DPP(1) OP_ADD(2) OP_DPP(3) OP_ADD(4) OP_DPP(5) OP_ADD(6) OP_DPP(7) OP_ADD(8) OP_DPP(9) OP_ADD(10) OP_DPP(11) OP_ADD(12) OP_DPP(13) OP_ADD(14) OP_DPP(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_dpp_3a1b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_DPP(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_DPP(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_DPP(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_DPP(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_dpp_16a(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_dpp_16b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_DPP(0) OP_DPP(1) OP_DPP(2) OP_DPP(3) OP_DPP(4) OP_DPP(5) OP_DPP(6) OP_DPP(7) OP_DPP(8) OP_DPP(9) OP_DPP(10) OP_DPP(11) OP_DPP(12) OP_DPP(13) OP_DPP(14) OP_DPP(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_fma_mad_8a8b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_FMA(0) OP_FMA(1) OP_FMA(2) OP_FMA(3) OP_FMA(4) OP_FMA(5) OP_FMA(6) OP_FMA(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_fma_mad_12a4b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_FMA(0) OP_FMA(1) OP_FMA(2) OP_FMA(3) OP_FMA(4) OP_FMA(5) OP_FMA(6) OP_FMA(7) OP_FMA(8) OP_FMA(9) OP_FMA(10) OP_FMA(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_fma_mad_14a2b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_FMA(0) OP_FMA(1) OP_FMA(2) OP_FMA(3) OP_FMA(4) OP_FMA(5) OP_FMA(6) OP_FMA(7) OP_FMA(8) OP_FMA(9) OP_FMA(10) OP_FMA(11) OP_FMA(12) OP_FMA(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_fma_mad_15a1b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_FMA(0) OP_FMA(1) OP_FMA(2) OP_FMA(3) OP_FMA(4) OP_FMA(5) OP_FMA(6) OP_FMA(7) OP_FMA(8) OP_FMA(9) OP_FMA(10) OP_FMA(11) OP_FMA(12) OP_FMA(13) OP_FMA(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_fma_mad_alt(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_FMA(0) OP_MADU24(1) OP_FMA(2) OP_MADU24(3) OP_FMA(4) OP_MADU24(5) OP_FMA(6) OP_MADU24(7) OP_FMA(8) OP_MADU24(9) OP_FMA(10) OP_MADU24(11) OP_FMA(12) OP_MADU24(13) OP_FMA(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_fma_mad_3a1b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_FMA(0) OP_FMA(1) OP_FMA(2) OP_MADU24(3) OP_FMA(4) OP_FMA(5) OP_FMA(6) OP_MADU24(7) OP_FMA(8) OP_FMA(9) OP_FMA(10) OP_MADU24(11) OP_FMA(12) OP_FMA(13) OP_FMA(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_fma_mad_16a(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_FMA(0) OP_FMA(1) OP_FMA(2) OP_FMA(3) OP_FMA(4) OP_FMA(5) OP_FMA(6) OP_FMA(7) OP_FMA(8) OP_FMA(9) OP_FMA(10) OP_FMA(11) OP_FMA(12) OP_FMA(13) OP_FMA(14) OP_FMA(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_fma_mad_16b(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_MADU24(0) OP_MADU24(1) OP_MADU24(2) OP_MADU24(3) OP_MADU24(4) OP_MADU24(5) OP_MADU24(6) OP_MADU24(7) OP_MADU24(8) OP_MADU24(9) OP_MADU24(10) OP_MADU24(11) OP_MADU24(12) OP_MADU24(13) OP_MADU24(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

typedef void (*kfn)(uint32_t *, uint32_t);
struct Ent { const char *name; kfn f; };
int main()
{
    Ent ents[] = {{"add_max_8a8b", k_add_max_8a8b}, {"add_max_12a4b", k_add_max_12a4b}, {"add_max_14a2b", k_add_max_14a2b}, {"add_max_15a1b", k_add_max_15a1b}, {"add_max_alt", k_add_max_alt}, {"add_max_3a1b", k_add_max_3a1b}, {"add_max_16a", k_add_max_16a}, {"add_max_16b", k_add_max_16b}, {"add_mad_8a8b", k_add_mad_8a8b}, {"add_mad_12a4b", k_add_mad_12a4b}, {"add_mad_14a2b", k_add_mad_14a2b}, {"add_mad_15a1b", k_add_mad_15a1b}, {"add_mad_alt", k_add_mad_alt}, {"add_mad_3a1b", k_add_mad_3a1b}, {"add_mad_16a", k_add_mad_16a}, {"add_mad_16b", k_add_mad_16b}, {"and_cnd_8a8b", k_and_cnd_8a8b}, {"and_cnd_12a4b", k_and_cnd_12a4b}, {"and_cnd_14a2b", k_and_cnd_14a2b}, {"and_cnd_15a1b", k_and_cnd_15a1b}, {"and_cnd_alt", k_and_cnd_alt}, {"and_cnd_3a1b", k_and_cnd_3a1b}, {"and_cnd_16a", k_and_cnd_16a}, {"and_cnd_16b", k_and_cnd_16b}, {"add_add3_8a8b", k_add_add3_8a8b}, {"add_add3_12a4b", k_add_add3_12a4b}, {"add_add3_14a2b", k_add_add3_14a2b}, {"add_add3_15a1b", k_add_add3_15a1b}, {"add_add3_alt", k_add_add3_alt}, {"add_add3_3a1b", k_add_add3_3a1b}, {"add_add3_16a", k_add_add3_16a}, {"add_add3_16b", k_add_add3_16b}, {"add_dpp_8a8b", k_add_dpp_8a8b}, {"add_dpp_12a4b", k_add_dpp_12a4b}, {"add_dpp_14a2b", k_add_dpp_14a2b}, {"add_dpp_15a1b", k_add_dpp_15a1b}, {"add_dpp_alt", k_add_dpp_alt}, {"add_dpp_3a1b", k_add_dpp_3a1b}, {"add_dpp_16a", k_add_dpp_16a}, {"add_dpp_16b", k_add_dpp_16b}, {"fma_mad_8a8b", k_fma_mad_8a8b}, {"fma_mad_12a4b", k_fma_mad_12a4b}, {"fma_mad_14a2b", k_fma_mad_14a2b}, {"fma_mad_15a1b", k_fma_mad_15a1b}, {"fma_mad_alt", k_fma_mad_alt}, {"fma_mad_3a1b", k_fma_mad_3a1b}, {"fma_mad_16a", k_fma_mad_16a}, {"fma_mad_16b", k_fma_mad_16b}};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;
    uint32_t *d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("CUs %d clock %d kHz\n", cus, prop.clockRate);
    for (auto &en : ents) {
        en.f<<<blocks, 256>>>(d, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; rep++) en.f<<<blocks, 256>>>(d, rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double insts_per_simd = 5.0 * 8 * ITER * 16;
        double cyc = ms * 1e-3 * 2.4e9;
        printf("%-24s %8.3f ms   %.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", en.name, ms, cyc / insts_per_simd);
        fflush(stdout);
    }
    return 0;
}
