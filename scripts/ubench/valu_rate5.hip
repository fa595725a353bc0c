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

KERNEL(k_add, OP_ADD) KERNEL(k_sub, OP_SUB) KERNEL(k_add3, OP_ADD3) KERNEL(k_lshladd, OP_LSHLADD)
KERNEL(k_addlshl, OP_ADDLSHL) KERNEL(k_madu24, OP_MADU24) KERNEL(k_madi24, OP_MADI24) KERNEL(k_mulu24, OP_MULU24)
KERNEL(k_mullo, OP_MULLO) KERNEL(k_and, OP_AND) KERNEL(k_andor, OP_ANDOR) KERNEL(k_max, OP_MAX) KERNEL(k_min3, OP_MIN3)
KERNEL(k_ashr, OP_ASHR) KERNEL(k_cndmask, OP_CNDMASK) KERNEL(k_cmp, OP_CMP) KERNEL(k_cmpsdwa, OP_CMPSDWA)
KERNEL(k_addc, OP_ADDC) KERNEL(k_sdwa, OP_SDWA) KERNEL(k_dpp, OP_DPP) KERNEL(k_dpprow, OP_DPPROW) KERNEL(k_adddpp, OP_ADDDPP)
KERNEL(k_pkmax, OP_PKMAX) KERNEL(k_pkadd, OP_PKADD) KERNEL(k_pkmad, OP_PKMAD) KERNEL(k_pkmul, OP_PKMUL)
KERNEL(k_pklshl, OP_PKLSHL) KERNEL(k_pksub, OP_PKSUB) KERNEL(k_alignbit, OP_ALIGNBIT) KERNEL(k_perm, OP_PERM)
KERNEL(k_bfe, OP_BFE) KERNEL(k_bcnt, OP_BCNT) KERNEL(k_dot4, OP_DOT4) KERNEL(k_fma, OP_FMA) KERNEL(k_sad, OP_SAD)
KERNEL(k_mov, OP_MOV) KERNEL(k_bfi, OP_BFI) KERNEL(k_madu16, OP_MADU16) KERNEL(k_mulhi, OP_MADU64)


#define BODYMIX(A,B)                                                                        \
    asm volatile(A(0) B(1) A(2) B(3) A(4) B(5) A(6) B(7) A(8) B(9) A(10) B(11) A(12) B(13) A(14) B(15) \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) \
                 : "v"(x), "v"(y), "s"(sm) : "vcc")
#define KERNELMIX(NAME, A, B)                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)   \
    {                                                                           \
        uint32_t r[16];                                                         \
        for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;        \
        uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;                 \
        unsigned long long sm = 0x1234567887654321ull ^ seed;                   \
        for (int it = 0; it < ITER; it++) { BODYMIX(A, B); }                    \
        uint32_t a = 0;                                                         \
        for (int i = 0; i < 16; i++) a ^= r[i];                                 \
        out[blockIdx.x * 256 + threadIdx.x] = a;                                \
    }
#define OPA_mix_add_max(i) "v_add_u32 %" S_(i) ", %" S_(i) ", %16\n"
#define OPB_mix_add_max(i) "v_max_u32 %" S_(i) ", %" S_(i) ", %16\n"
#define OPM_mix_add_max(i) OPM2_mix_add_max(i)
#define OPA_mix_add_mad(i) "v_add_u32 %" S_(i) ", %" S_(i) ", %16\n"
#define OPB_mix_add_mad(i) "v_mad_u32_u24 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OPM_mix_add_mad(i) OPM2_mix_add_mad(i)
#define OPA_mix_fma_pk(i) "v_fma_f32 %" S_(i) ", %" S_(i) ", %16, %17\n"
#define OPB_mix_fma_pk(i) "v_pk_max_u16 %" S_(i) ", %" S_(i) ", %16\n"
#define OPM_mix_fma_pk(i) OPM2_mix_fma_pk(i)
#define OPA_mix_and_cmp(i) "v_and_b32 %" S_(i) ", %" S_(i) ", %16\n"
#define OPB_mix_and_cmp(i) "v_cmp_eq_u32 vcc, %" S_(i) ", %16\n"
#define OPM_mix_and_cmp(i) OPM2_mix_and_cmp(i)
#define OPA_mix_add_salu(i) "v_add_u32 %" S_(i) ", %" S_(i) ", %16\n"
#define OPB_mix_add_salu(i) "s_add_u32 s20, s20, s21\n"
#define OPM_mix_add_salu(i) OPM2_mix_add_salu(i)
#define OPA_mix_max_salu(i) "v_max_u32 %" S_(i) ", %" S_(i) ", %16\n"
#define OPB_mix_max_salu(i) "s_add_u32 s20, s20, s21\n"
#define OPM_mix_max_salu(i) OPM2_mix_max_salu(i)
#define OPA_all_salu(i) "s_add_u32 s20, s20, s21\n"
#define OPB_all_salu(i) "s_and_b64 s[22:23], s[22:23], s[24:25]\n"
#define OPM_all_salu(i) OPM2_all_salu(i)
KERNELMIX(km_mix_add_max, OPA_mix_add_max, OPB_mix_add_max)
KERNELMIX(km_mix_add_mad, OPA_mix_add_mad, OPB_mix_add_mad)
KERNELMIX(km_mix_fma_pk, OPA_mix_fma_pk, OPB_mix_fma_pk)
KERNELMIX(km_mix_and_cmp, OPA_mix_and_cmp, OPB_mix_and_cmp)

__global__ __launch_bounds__(256) void kc_aabb_add_max(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_max_u32 %2, %2, %16\n" "v_max_u32 %3, %3, %16\n" "v_add_u32 %4, %4, %16\n" "v_add_u32 %5, %5, %16\n" "v_max_u32 %6, %6, %16\n" "v_max_u32 %7, %7, %16\n" "v_add_u32 %8, %8, %16\n" "v_add_u32 %9, %9, %16\n" "v_max_u32 %10, %10, %16\n" "v_max_u32 %11, %11, %16\n" "v_add_u32 %12, %12, %16\n" "v_add_u32 %13, %13, %16\n" "v_max_u32 %14, %14, %16\n" "v_max_u32 %15, %15, %16\n" 
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void kc_dep_add_chain(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %1, %1, %16\n" 
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void kc_dep_pairs(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %0, %16\n" "v_add_u32 %2, %2, %16\n" "v_add_u32 %3, %2, %16\n" "v_add_u32 %4, %4, %16\n" "v_add_u32 %5, %4, %16\n" "v_add_u32 %6, %6, %16\n" "v_add_u32 %7, %6, %16\n" "v_add_u32 %8, %8, %16\n" "v_add_u32 %9, %8, %16\n" "v_add_u32 %10, %10, %16\n" "v_add_u32 %11, %10, %16\n" "v_add_u32 %12, %12, %16\n" "v_add_u32 %13, %12, %16\n" "v_add_u32 %14, %14, %16\n" "v_add_u32 %15, %14, %16\n" 
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void kc_aab_add_max(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_max_u32 %2, %2, %16\n" "v_add_u32 %3, %3, %16\n" "v_add_u32 %4, %4, %16\n" "v_max_u32 %5, %5, %16\n" "v_add_u32 %6, %6, %16\n" "v_add_u32 %7, %7, %16\n" "v_max_u32 %8, %8, %16\n" "v_add_u32 %9, %9, %16\n" "v_add_u32 %10, %10, %16\n" "v_max_u32 %11, %11, %16\n" "v_add_u32 %12, %12, %16\n" "v_add_u32 %13, %13, %16\n" "v_max_u32 %14, %14, %16\n" "v_max_u32 %15, %15, %16\n" 
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void kc_add_sub_and_mix(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_add_u32 %0, %0, %16\n" "v_sub_u32 %1, %1, %17\n" "v_and_b32 %2, %2, %16\n" "v_lshrrev_b32 %3, 1, %3\n" "v_add_u32 %4, %4, %16\n" "v_sub_u32 %5, %5, %17\n" "v_and_b32 %6, %6, %16\n" "v_lshrrev_b32 %7, 1, %7\n" "v_add_u32 %8, %8, %16\n" "v_sub_u32 %9, %9, %17\n" "v_and_b32 %10, %10, %16\n" "v_lshrrev_b32 %11, 1, %11\n" "v_add_u32 %12, %12, %16\n" "v_sub_u32 %13, %13, %17\n" "v_and_b32 %14, %14, %16\n" "v_lshrrev_b32 %15, 1, %15\n" 
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void kc_fma_add_mix(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_fma_f32 %0, %0, %16, %17\n" "v_add_u32 %1, %1, %17\n" "v_fma_f32 %2, %2, %16, %17\n" "v_add_u32 %3, %3, %17\n" "v_fma_f32 %4, %4, %16, %17\n" "v_add_u32 %5, %5, %17\n" "v_fma_f32 %6, %6, %16, %17\n" "v_add_u32 %7, %7, %17\n" "v_fma_f32 %8, %8, %16, %17\n" "v_add_u32 %9, %9, %17\n" "v_fma_f32 %10, %10, %16, %17\n" "v_add_u32 %11, %11, %17\n" "v_fma_f32 %12, %12, %16, %17\n" "v_add_u32 %13, %13, %17\n" "v_fma_f32 %14, %14, %16, %17\n" "v_add_u32 %15, %15, %17\n" 
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void kc_a_ds_mix(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %2, %2, %16\n" "v_add_u32 %3, %3, %16\n" "v_add_u32 %4, %4, %16\n" "v_add_u32 %5, %5, %16\n" "v_add_u32 %6, %6, %16\n" "v_add_u32 %7, %7, %16\n" "v_add_u32 %8, %8, %16\n" "v_add_u32 %9, %9, %16\n" "v_add_u32 %10, %10, %16\n" "v_add_u32 %11, %11, %16\n" "v_add_u32 %12, %12, %16\n" "v_add_u32 %13, %13, %16\n" "v_add_u32 %14, %14, %16\n" "v_add_u32 %15, %15, %16\n" 
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void kd_run4(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER * 16 / 16; it++) {
        asm volatile("v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %2, %2, %16\n" "v_add_u32 %3, %3, %16\n" "v_max_u32 %4, %4, %16\n" "v_max_u32 %5, %5, %16\n" "v_max_u32 %6, %6, %16\n" "v_max_u32 %7, %7, %16\n" "v_add_u32 %8, %8, %16\n" "v_add_u32 %9, %9, %16\n" "v_add_u32 %10, %10, %16\n" "v_add_u32 %11, %11, %16\n" "v_max_u32 %12, %12, %16\n" "v_max_u32 %13, %13, %16\n" "v_max_u32 %14, %14, %16\n" "v_max_u32 %15, %15, %16\n" 
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void kd_run8(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER * 16 / 16; it++) {
        asm volatile("v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %2, %2, %16\n" "v_add_u32 %3, %3, %16\n" "v_add_u32 %4, %4, %16\n" "v_add_u32 %5, %5, %16\n" "v_add_u32 %6, %6, %16\n" "v_add_u32 %7, %7, %16\n" "v_max_u32 %8, %8, %16\n" "v_max_u32 %9, %9, %16\n" "v_max_u32 %10, %10, %16\n" "v_max_u32 %11, %11, %16\n" "v_max_u32 %12, %12, %16\n" "v_max_u32 %13, %13, %16\n" "v_max_u32 %14, %14, %16\n" "v_max_u32 %15, %15, %16\n" 
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void kd_run16(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER * 16 / 32; it++) {
        asm volatile("v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %2, %2, %16\n" "v_add_u32 %3, %3, %16\n" "v_add_u32 %4, %4, %16\n" "v_add_u32 %5, %5, %16\n" "v_add_u32 %6, %6, %16\n" "v_add_u32 %7, %7, %16\n" "v_add_u32 %8, %8, %16\n" "v_add_u32 %9, %9, %16\n" "v_add_u32 %10, %10, %16\n" "v_add_u32 %11, %11, %16\n" "v_add_u32 %12, %12, %16\n" "v_add_u32 %13, %13, %16\n" "v_add_u32 %14, %14, %16\n" "v_add_u32 %15, %15, %16\n" "v_max_u32 %0, %0, %16\n" "v_max_u32 %1, %1, %16\n" "v_max_u32 %2, %2, %16\n" "v_max_u32 %3, %3, %16\n" "v_max_u32 %4, %4, %16\n" "v_max_u32 %5, %5, %16\n" "v_max_u32 %6, %6, %16\n" "v_max_u32 %7, %7, %16\n" "v_max_u32 %8, %8, %16\n" "v_max_u32 %9, %9, %16\n" "v_max_u32 %10, %10, %16\n" "v_max_u32 %11, %11, %16\n" "v_max_u32 %12, %12, %16\n" "v_max_u32 %13, %13, %16\n" "v_max_u32 %14, %14, %16\n" "v_max_u32 %15, %15, %16\n" 
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void kd_run32(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER * 16 / 64; it++) {
        asm volatile("v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %2, %2, %16\n" "v_add_u32 %3, %3, %16\n" "v_add_u32 %4, %4, %16\n" "v_add_u32 %5, %5, %16\n" "v_add_u32 %6, %6, %16\n" "v_add_u32 %7, %7, %16\n" "v_add_u32 %8, %8, %16\n" "v_add_u32 %9, %9, %16\n" "v_add_u32 %10, %10, %16\n" "v_add_u32 %11, %11, %16\n" "v_add_u32 %12, %12, %16\n" "v_add_u32 %13, %13, %16\n" "v_add_u32 %14, %14, %16\n" "v_add_u32 %15, %15, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %2, %2, %16\n" "v_add_u32 %3, %3, %16\n" "v_add_u32 %4, %4, %16\n" "v_add_u32 %5, %5, %16\n" "v_add_u32 %6, %6, %16\n" "v_add_u32 %7, %7, %16\n" "v_add_u32 %8, %8, %16\n" "v_add_u32 %9, %9, %16\n" "v_add_u32 %10, %10, %16\n" "v_add_u32 %11, %11, %16\n" "v_add_u32 %12, %12, %16\n" "v_add_u32 %13, %13, %16\n" "v_add_u32 %14, %14, %16\n" "v_add_u32 %15, %15, %16\n" "v_max_u32 %0, %0, %16\n" "v_max_u32 %1, %1, %16\n" "v_max_u32 %2, %2, %16\n" "v_max_u32 %3, %3, %16\n" "v_max_u32 %4, %4, %16\n" "v_max_u32 %5, %5, %16\n" "v_max_u32 %6, %6, %16\n" "v_max_u32 %7, %7, %16\n" "v_max_u32 %8, %8, %16\n" "v_max_u32 %9, %9, %16\n" "v_max_u32 %10, %10, %16\n" "v_max_u32 %11, %11, %16\n" "v_max_u32 %12, %12, %16\n" "v_max_u32 %13, %13, %16\n" "v_max_u32 %14, %14, %16\n" "v_max_u32 %15, %15, %16\n" "v_max_u32 %0, %0, %16\n" "v_max_u32 %1, %1, %16\n" "v_max_u32 %2, %2, %16\n" "v_max_u32 %3, %3, %16\n" "v_max_u32 %4, %4, %16\n" "v_max_u32 %5, %5, %16\n" "v_max_u32 %6, %6, %16\n" "v_max_u32 %7, %7, %16\n" "v_max_u32 %8, %8, %16\n" "v_max_u32 %9, %9, %16\n" "v_max_u32 %10, %10, %16\n" "v_max_u32 %11, %11, %16\n" "v_max_u32 %12, %12, %16\n" "v_max_u32 %13, %13, %16\n" "v_max_u32 %14, %14, %16\n" "v_max_u32 %15, %15, %16\n" 
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void kd_f12s4(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER * 16 / 16; it++) {
        asm volatile("v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %2, %2, %16\n" "v_add_u32 %3, %3, %16\n" "v_add_u32 %4, %4, %16\n" "v_add_u32 %5, %5, %16\n" "v_add_u32 %6, %6, %16\n" "v_add_u32 %7, %7, %16\n" "v_add_u32 %8, %8, %16\n" "v_add_u32 %9, %9, %16\n" "v_add_u32 %10, %10, %16\n" "v_add_u32 %11, %11, %16\n" "v_max_u32 %12, %12, %16\n" "v_max_u32 %13, %13, %16\n" "v_max_u32 %14, %14, %16\n" "v_max_u32 %15, %15, %16\n" 
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
                 : "v"(x), "v"(y), "s"(sm) : "vcc");
    }
    uint32_t a = 0;
    for (int i = 0; i < 16; i++) a ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void kd_f24s8(uint32_t *out, uint32_t seed)
{
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    unsigned long long sm = 0x1234567887654321ull ^ seed;
    for (int it = 0; it < ITER * 16 / 32; it++) {
        asm volatile("v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %2, %2, %16\n" "v_add_u32 %3, %3, %16\n" "v_add_u32 %4, %4, %16\n" "v_add_u32 %5, %5, %16\n" "v_add_u32 %6, %6, %16\n" "v_add_u32 %7, %7, %16\n" "v_add_u32 %8, %8, %16\n" "v_add_u32 %9, %9, %16\n" "v_add_u32 %10, %10, %16\n" "v_add_u32 %11, %11, %16\n" "v_add_u32 %12, %12, %16\n" "v_add_u32 %13, %13, %16\n" "v_add_u32 %14, %14, %16\n" "v_add_u32 %15, %15, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %2, %2, %16\n" "v_add_u32 %3, %3, %16\n" "v_add_u32 %4, %4, %16\n" "v_add_u32 %5, %5, %16\n" "v_add_u32 %6, %6, %16\n" "v_add_u32 %7, %7, %16\n" "v_max_u32 %8, %8, %16\n" "v_max_u32 %9, %9, %16\n" "v_max_u32 %10, %10, %16\n" "v_max_u32 %11, %11, %16\n" "v_max_u32 %12, %12, %16\n" "v_max_u32 %13, %13, %16\n" "v_max_u32 %14, %14, %16\n" "v_max_u32 %15, %15, %16\n" 
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
    Ent ents[] = {{"run4", kd_run4},
{"run8", kd_run8},
{"run16", kd_run16},
{"run32", kd_run32},
{"f12s4", kd_f12s4},
{"f24s8", kd_f24s8},
{"aabb_add_max", kc_aabb_add_max},
{"dep_add_chain", kc_dep_add_chain},
{"dep_pairs", kc_dep_pairs},
{"aab_add_max", kc_aab_add_max},
{"add_sub_and_mix", kc_add_sub_and_mix},
{"fma_add_mix", kc_fma_add_mix},
{"a_ds_mix", kc_a_ds_mix},
{"mix_add_max", km_mix_add_max},
{"mix_add_mad", km_mix_add_mad},
{"mix_fma_pk", km_mix_fma_pk},
{"mix_and_cmp", km_mix_and_cmp},
{"v_add_u32", k_add}, {"v_sub_u32", k_sub}, {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshladd},
                  {"v_add_lshl_u32", k_addlshl}, {"v_mad_u32_u24", k_madu24}, {"v_mad_i32_i24", k_madi24},
                  {"v_mul_u32_u24", k_mulu24}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_and_b32", k_and},
                  {"v_and_or_b32", k_andor}, {"v_max_u32", k_max}, {"v_min3_u32", k_min3}, {"v_ashrrev_i32", k_ashr},
                  {"v_cndmask_b32(sgpr)", k_cndmask}, {"v_cmp_eq_u32", k_cmp}, {"v_cmp_sdwa", k_cmpsdwa},
                  {"v_addc_co_u32", k_addc}, {"v_lshlrev_sdwa", k_sdwa}, {"v_mov_dpp wave_shr", k_dpp},
                  {"v_mov_dpp row_shr", k_dpprow}, {"v_add_u32_dpp", k_adddpp}, {"v_pk_max_u16", k_pkmax},
                  {"v_pk_add_u16", k_pkadd}, {"v_pk_mad_u16", k_pkmad}, {"v_pk_mul_lo_u16", k_pkmul},
                  {"v_pk_lshlrev_b16", k_pklshl}, {"v_pk_sub_i16", k_pksub}, {"v_alignbit_b32", k_alignbit},
                  {"v_perm_b32", k_perm}, {"v_bfe_u32", k_bfe}, {"v_bcnt_u32_b32", k_bcnt}, {"v_dot4_u32_u8", k_dot4},
                  {"v_fma_f32", k_fma}, {"v_sad_u8", k_sad}, {"v_mov_b32", k_mov}, {"v_bfi_b32", k_bfi},
                  {"v_mad_u16", k_madu16}};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;  // 8 blocks x 4 waves = 32 waves per CU = 8 per SIMD
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
        double insts_per_simd = 5.0 * 8 /*waves per simd*/ * ITER * 16;
        double cyc = ms * 1e-3 * 2.4e9;  // at 2.4 GHz nominal
        fflush(stdout); printf("%-24s %8.3f ms   %.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", en.name, ms, cyc / insts_per_simd);
    }
    return 0;
}
