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


__global__ __launch_bounds__(256) void k_add_add(uint32_t *out, uint32_t seed)
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

__global__ __launch_bounds__(256) void k_add_sub(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_SUB(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_SUB(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_SUB(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_SUB(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_add3(uint32_t *out, uint32_t seed)
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

__global__ __launch_bounds__(256) void k_add_lshladd(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_LSHLADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_LSHLADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_LSHLADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_LSHLADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_addlshl(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADDLSHL(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADDLSHL(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADDLSHL(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADDLSHL(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_madu24(uint32_t *out, uint32_t seed)
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

__global__ __launch_bounds__(256) void k_add_madi24(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_MADI24(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_MADI24(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_MADI24(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_MADI24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_mulu24(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_MULU24(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_MULU24(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_MULU24(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_MULU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_mullo(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_MULLO(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_MULLO(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_MULLO(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_MULLO(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_and(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_AND(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_AND(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_AND(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_AND(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_andor(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ANDOR(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ANDOR(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ANDOR(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ANDOR(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_max(uint32_t *out, uint32_t seed)
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

__global__ __launch_bounds__(256) void k_add_min3(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_MIN3(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_MIN3(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_MIN3(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_MIN3(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_ashr(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ASHR(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ASHR(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ASHR(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ASHR(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_cndmask(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_CNDMASK(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_CNDMASK(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_CNDMASK(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_CNDMASK(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_cmp(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_CMP(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_CMP(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_CMP(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_CMP(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_cmpsdwa(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_CMPSDWA(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_CMPSDWA(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_CMPSDWA(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_CMPSDWA(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_addc(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADDC(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADDC(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADDC(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADDC(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_sdwa(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_SDWA(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_SDWA(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_SDWA(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_SDWA(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_dpp(uint32_t *out, uint32_t seed)
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

__global__ __launch_bounds__(256) void k_add_dpprow(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_DPPROW(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_DPPROW(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_DPPROW(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_DPPROW(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_adddpp(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ADDDPP(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ADDDPP(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ADDDPP(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ADDDPP(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_pkmax(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_PKMAX(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_PKMAX(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_PKMAX(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_PKMAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_pkadd(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_PKADD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_PKADD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_PKADD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_PKADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_pkmad(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_PKMAD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_PKMAD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_PKMAD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_PKMAD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_pkmul(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_PKMUL(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_PKMUL(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_PKMUL(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_PKMUL(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_pklshl(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_PKLSHL(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_PKLSHL(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_PKLSHL(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_PKLSHL(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_pksub(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_PKSUB(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_PKSUB(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_PKSUB(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_PKSUB(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_alignbit(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_ALIGNBIT(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_ALIGNBIT(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_ALIGNBIT(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_ALIGNBIT(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_perm(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_PERM(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_PERM(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_PERM(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_PERM(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_bfe(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_BFE(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_BFE(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_BFE(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_BFE(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_bcnt(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_BCNT(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_BCNT(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_BCNT(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_BCNT(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_dot4(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_DOT4(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_DOT4(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_DOT4(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_DOT4(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_fma(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_FMA(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_FMA(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_FMA(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_FMA(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_sad(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_SAD(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_SAD(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_SAD(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_SAD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_mov(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_MOV(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_MOV(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_MOV(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_MOV(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_bfi(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_BFI(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_BFI(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_BFI(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_BFI(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_madu16(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_MADU16(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_MADU16(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_MADU16(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_MADU16(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_cvtpk(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_CVTPK(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_CVTPK(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_CVTPK(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_CVTPK(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_add_madu64(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_ADD(0) OP_ADD(1) OP_ADD(2) OP_MADU64(3) OP_ADD(4) OP_ADD(5) OP_ADD(6) OP_MADU64(7) OP_ADD(8) OP_ADD(9) OP_ADD(10) OP_MADU64(11) OP_ADD(12) OP_ADD(13) OP_ADD(14) OP_MADU64(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_add3(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_ADD3(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_ADD3(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_ADD3(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_ADD3(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_lshladd(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_LSHLADD(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_LSHLADD(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_LSHLADD(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_LSHLADD(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_madu24(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_MADU24(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_MADU24(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_MADU24(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_MADU24(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_max(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_MAX(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_MAX(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_MAX(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_MAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_cmp(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_CMP(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_CMP(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_CMP(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_CMP(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_addc(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_ADDC(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_ADDC(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_ADDC(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_ADDC(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_sdwa(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_SDWA(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_SDWA(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_SDWA(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_SDWA(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_dpp(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_DPP(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_DPP(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_DPP(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_DPP(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_pkmax(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_PKMAX(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_PKMAX(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_PKMAX(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_PKMAX(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_alignbit(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_ALIGNBIT(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_ALIGNBIT(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_ALIGNBIT(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_ALIGNBIT(15)
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_and_perm(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile(OP_AND(0) OP_AND(1) OP_AND(2) OP_PERM(3) OP_AND(4) OP_AND(5) OP_AND(6) OP_PERM(7) OP_AND(8) OP_AND(9) OP_AND(10) OP_PERM(11) OP_AND(12) OP_AND(13) OP_AND(14) OP_PERM(15)
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
    Ent ents[] = {{"add_add", k_add_add}, {"add_sub", k_add_sub}, {"add_add3", k_add_add3}, {"add_lshladd", k_add_lshladd}, {"add_addlshl", k_add_addlshl}, {"add_madu24", k_add_madu24}, {"add_madi24", k_add_madi24}, {"add_mulu24", k_add_mulu24}, {"add_mullo", k_add_mullo}, {"add_and", k_add_and}, {"add_andor", k_add_andor}, {"add_max", k_add_max}, {"add_min3", k_add_min3}, {"add_ashr", k_add_ashr}, {"add_cndmask", k_add_cndmask}, {"add_cmp", k_add_cmp}, {"add_cmpsdwa", k_add_cmpsdwa}, {"add_addc", k_add_addc}, {"add_sdwa", k_add_sdwa}, {"add_dpp", k_add_dpp}, {"add_dpprow", k_add_dpprow}, {"add_adddpp", k_add_adddpp}, {"add_pkmax", k_add_pkmax}, {"add_pkadd", k_add_pkadd}, {"add_pkmad", k_add_pkmad}, {"add_pkmul", k_add_pkmul}, {"add_pklshl", k_add_pklshl}, {"add_pksub", k_add_pksub}, {"add_alignbit", k_add_alignbit}, {"add_perm", k_add_perm}, {"add_bfe", k_add_bfe}, {"add_bcnt", k_add_bcnt}, {"add_dot4", k_add_dot4}, {"add_fma", k_add_fma}, {"add_sad", k_add_sad}, {"add_mov", k_add_mov}, {"add_bfi", k_add_bfi}, {"add_madu16", k_add_madu16}, {"add_cvtpk", k_add_cvtpk}, {"add_madu64", k_add_madu64}, {"and_add3", k_and_add3}, {"and_lshladd", k_and_lshladd}, {"and_madu24", k_and_madu24}, {"and_max", k_and_max}, {"and_cmp", k_and_cmp}, {"and_addc", k_and_addc}, {"and_sdwa", k_and_sdwa}, {"and_dpp", k_and_dpp}, {"and_pkmax", k_and_pkmax}, {"and_alignbit", k_and_alignbit}, {"and_perm", k_and_perm}};
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 8;
    uint32_t *d;
    (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (auto &en : ents) {
        en.f<<<blocks, 256>>>(d, 1);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int rep = 0; rep < 5; rep++) en.f<<<blocks, 256>>>(d, rep);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("3a1b %-28s %.2f cyc/instr (2.4 GHz)\n", en.name, ms * 1e-3 * 2.4e9 / (5.0 * 8 * ITER * 16));
        fflush(stdout);
    }
    return 0;
}
