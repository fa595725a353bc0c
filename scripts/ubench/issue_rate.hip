// Micro-benchmark (dev tool): issue cost of wave64 vector instructions on gfx950, alone and mixed, and of the
// 4x4x4 i8 MFMA beside them.  One file, one command:
//   hipcc --offload-arch=gfx950 -O3 -o issue_rate scripts/ubench/issue_rate.hip && ./issue_rate [filter]
// Every kernel runs ITER iterations of a 16-instruction body on 16 independent registers, 8 waves per SIMD on every CU;
// the figure printed is SIMD cycles per wave-instruction at 2.4 GHz (4.0 = one wave64 instruction per 4 cycles).
// It replaces the nine valu_rate*.hip files of round 1; DESIGN.md §3.1 / §7 quote its output.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#define ITER 2048
#define R16(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
// A/B alternate two instruction kinds (even / odd registers); P = A for the first 12, B for the last 4, and so on
#define ALT(A, B) A(0) B(1) A(2) B(3) A(4) B(5) A(6) B(7) A(8) B(9) A(10) B(11) A(12) B(13) A(14) B(15)
#define MIX12_4(A, B) A(0) A(1) A(2) B(3) A(4) A(5) A(6) B(7) A(8) A(9) A(10) B(11) A(12) A(13) A(14) B(15)
#define MIX15_1(A, B) A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11) A(12) A(13) A(14) B(15)
#define SDWA " dst_sel:DWORD dst_unused:UNUSED_PAD "
#define KERNEL(NAME, BODY)                                                                                             \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)                                          \
    {                                                                                                                  \
        uint32_t r[16];                                                                                                \
        for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;                                               \
        uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;                                                        \
        for (int it = 0; it < ITER; it++)                                                                              \
            asm volatile(BODY : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]),    \
                         "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]),       \
                         "+v"(r[14]), "+v"(r[15]) : "v"(x), "v"(y) : "vcc");                                           \
        uint32_t a = 0;                                                                                                \
        for (int i = 0; i < 16; i++) a ^= r[i];                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = a;                                                                       \
    }
// ---- one line per instruction ----
#define T_add(i) "v_add_u32 %" #i ", %" #i ", %16\n"
#define T_sub(i) "v_sub_u32 %" #i ", %" #i ", %16\n"
#define T_and(i) "v_and_b32 %" #i ", %" #i ", %16\n"
#define T_xor(i) "v_xor_b32 %" #i ", %" #i ", %16\n"
#define T_lshr(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define T_lshl(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define T_mov(i) "v_mov_b32 %" #i ", %16\n"
#define T_cndmask(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define T_bitop3(i) "v_bitop3_b32 %" #i ", %" #i ", %16, %17 bitop3:0x96\n"
#define T_add3(i) "v_add3_u32 %" #i ", %" #i ", %16, %17\n"
#define T_lshl_add(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %16\n"
#define T_max3(i) "v_max3_u32 %" #i ", %" #i ", %16, %17\n"
#define T_max(i) "v_max_u32 %" #i ", %" #i ", %16\n"
#define T_mad24(i) "v_mad_u32_u24 %" #i ", %" #i ", %16, %17\n"
#define T_mul_lo(i) "v_mul_lo_u32 %" #i ", %" #i ", %16\n"
#define T_cmp(i) "v_cmp_eq_u32 vcc, %" #i ", %16\n"
#define T_perm(i) "v_perm_b32 %" #i ", %" #i ", %16, %17\n"
#define T_alignbit(i) "v_alignbit_b32 %" #i ", %" #i ", %16, 16\n"
#define T_bfe(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define T_bcnt(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %16\n"
#define T_dot4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %16, %17\n"
#define T_lerp(i) "v_lerp_u8 %" #i ", %" #i ", %16, %17\n"
#define T_sad_u8(i) "v_sad_u8 %" #i ", %" #i ", %16, %17\n"
#define T_pk_add16(i) "v_pk_add_u16 %" #i ", %" #i ", %16\n"
#define T_pk_max16(i) "v_pk_max_u16 %" #i ", %" #i ", %16\n"
#define T_pk_lshr16(i) "v_pk_lshrrev_b16 %" #i ", 2, %" #i "\n"
#define T_pk_mad16(i) "v_pk_mad_u16 %" #i ", %" #i ", %16, %17\n"
#define T_dpp_mov(i) "v_mov_b32_dpp %" #i ", %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define T_dpp_add(i) "v_add_u32_dpp %" #i ", %" #i ", %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define T_sdwa_lshl(i) "v_lshlrev_b32_sdwa %" #i ", %16, %" #i SDWA "src0_sel:DWORD src1_sel:BYTE_1\n"
#define T_sdwa_add(i) "v_add_u32_sdwa %" #i ", %" #i ", %16" SDWA "src0_sel:BYTE_1 src1_sel:BYTE_2\n"
#define T_sdwa_and(i) "v_and_b32_sdwa %" #i ", %" #i ", %16" SDWA "src0_sel:WORD_1 src1_sel:DWORD\n"
#define T_cvt_ub(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define T_fma(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define T_add_u16(i) "v_add_u16 %" #i ", %" #i ", %16\n"
// ---- round 3: the lane-mask readers (v_cndmask 22.7 cycles?) and more opcodes ----
#define T_or(i) "v_or_b32 %" #i ", %" #i ", %16\n"
#define T_ashr(i) "v_ashrrev_i32 %" #i ", 3, %" #i "\n"
#define T_lshl1(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define T_subrev(i) "v_subrev_u32 %" #i ", %" #i ", %16\n"
#define T_not(i) "v_not_b32 %" #i ", %" #i "\n"
#define T_min(i) "v_min_u32 %" #i ", %" #i ", %16\n"
#define T_and_or(i) "v_and_or_b32 %" #i ", %" #i ", %16, %17\n"
#define T_or3(i) "v_or3_b32 %" #i ", %" #i ", %16, %17\n"
#define T_bfi(i) "v_bfi_b32 %" #i ", %" #i ", %16, %17\n"
#define T_add_lshl(i) "v_add_lshl_u32 %" #i ", %" #i ", %16, 2\n"
#define T_mul24(i) "v_mul_u32_u24 %" #i ", %" #i ", %16\n"
#define T_add_co(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %16\n"
#define T_addc_vcc(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n"
#define T_addc_sgpr(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", s[20:21]\n"
#define T_cndmask_sgpr(i) "v_cndmask_b32 %" #i ", %" #i ", %16, s[20:21]\n"
#define T_cmp_sgpr(i) "v_cmp_eq_u32 s[22:23], %" #i ", %16\n"
#define T_cmp_lt(i) "v_cmp_lt_u32 vcc, %" #i ", %16\n"
#define T_cndmask_e32(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %16, vcc\n"
#define T_sub_co(i) "v_sub_co_u32 %" #i ", vcc, %" #i ", %16\n"
#define T_mov_sgpr(i) "v_mov_b32 %" #i ", s20\n"
#define T_add_sgpr(i) "v_add_u32 %" #i ", s20, %" #i "\n"
#define T_max3_sgpr(i) "v_max3_u32 %" #i ", %" #i ", s20, %17\n"
#define T_pk_sub_sgpr(i) "v_pk_sub_u16 %" #i ", %" #i ", s20\n"
#define T_fmac(i) "v_fmac_f32 %" #i ", %16, %17\n"
#define T_mul_f32(i) "v_mul_f32 %" #i ", %" #i ", %16\n"
#define T_add_f32(i) "v_add_f32 %" #i ", %" #i ", %16\n"
#define T_xad(i) "v_xad_u32 %" #i ", %" #i ", %16, %17\n"
#define T_lshl_or(i) "v_lshl_or_b32 %" #i ", %" #i ", 2, %16\n"
#define T_sub_sdwa_free(i) "v_sub_u32 %" #i ", %16, %" #i "\n"
// the KERNEL macro with s[20:23] written before the loop (a lane mask in SGPRs, as a v_cmp would leave it)
#define KERNEL_S(NAME, BODY)                                                                                           \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)                                          \
    {                                                                                                                  \
        uint32_t r[16];                                                                                                \
        for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;                                               \
        uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;                                                        \
        asm volatile("s_mov_b32 s20, 0x55aa55aa\ns_mov_b32 s21, 0x0f0f3c3c\ns_mov_b64 s[22:23], 0\ns_mov_b64 vcc, s[20:21]" ::: "s20", "s21", "s22", "s23", "vcc"); \
        for (int it = 0; it < ITER; it++)                                                                              \
            asm volatile(BODY : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]),    \
                         "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]),       \
                         "+v"(r[14]), "+v"(r[15]) : "v"(x), "v"(y) : "vcc", "s20", "s21", "s22", "s23");               \
        uint32_t a = 0;                                                                                                \
        for (int i = 0; i < 16; i++) a ^= r[i];                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = a;                                                                       \
    }
#define RUN2(A, B) A(0) A(1) B(2) B(3) A(4) A(5) B(6) B(7) A(8) A(9) B(10) B(11) A(12) A(13) B(14) B(15)
#define RUN4(A, B) A(0) A(1) A(2) A(3) B(4) B(5) B(6) B(7) A(8) A(9) A(10) A(11) B(12) B(13) B(14) B(15)
#define RUN8(A, B) A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) B(8) B(9) B(10) B(11) B(12) B(13) B(14) B(15)
#define MIX3_1(A, B) A(0) A(1) A(2) B(3) A(4) A(5) A(6) B(7) A(8) A(9) A(10) B(11) A(12) A(13) A(14) B(15)
#define MIX2_2x(A, B) A(0) A(1) B(2) A(3) A(4) B(5) A(6) A(7) B(8) A(9) A(10) B(11) A(12) A(13) B(14) B(15)
KERNEL_S(k2_or, R16(T_or)) KERNEL_S(k2_ashr, R16(T_ashr)) KERNEL_S(k2_lshl1, R16(T_lshl1)) KERNEL_S(k2_subrev, R16(T_subrev))
KERNEL_S(k2_not, R16(T_not)) KERNEL_S(k2_min, R16(T_min)) KERNEL_S(k2_and_or, R16(T_and_or)) KERNEL_S(k2_or3, R16(T_or3))
KERNEL_S(k2_bfi, R16(T_bfi)) KERNEL_S(k2_add_lshl, R16(T_add_lshl)) KERNEL_S(k2_mul24, R16(T_mul24)) KERNEL_S(k2_add_co, R16(T_add_co))
KERNEL_S(k2_addc_vcc, R16(T_addc_vcc)) KERNEL_S(k2_addc_sgpr, R16(T_addc_sgpr)) KERNEL_S(k2_cndmask_vcc, R16(T_cndmask))
KERNEL_S(k2_cndmask_sgpr, R16(T_cndmask_sgpr)) KERNEL_S(k2_cmp_sgpr, R16(T_cmp_sgpr)) KERNEL_S(k2_cndmask_e32, R16(T_cndmask_e32))
KERNEL_S(k2_alt_cmp_cndmask, ALT(T_cmp_lt, T_cndmask)) KERNEL_S(k2_alt_cmpsgpr_cndmask, ALT(T_cmp_sgpr, T_cndmask_sgpr))
KERNEL_S(k2_alt_add_cndmask, ALT(T_add, T_cndmask)) KERNEL_S(k2_sub_co, R16(T_sub_co))
KERNEL_S(k2_mov_sgpr, R16(T_mov_sgpr)) KERNEL_S(k2_add_sgpr, R16(T_add_sgpr)) KERNEL_S(k2_max3_sgpr, R16(T_max3_sgpr))
KERNEL_S(k2_pk_sub_sgpr, R16(T_pk_sub_sgpr)) KERNEL_S(k2_fmac, R16(T_fmac)) KERNEL_S(k2_mul_f32, R16(T_mul_f32))
KERNEL_S(k2_add_f32, R16(T_add_f32)) KERNEL_S(k2_xad, R16(T_xad)) KERNEL_S(k2_lshl_or, R16(T_lshl_or))
// run lengths of a cheap kind (add) among a full-rate kind (max3), 8 + 8 per iteration
KERNEL_S(k2_run1_add_max3, ALT(T_add, T_max3)) KERNEL_S(k2_run2_add_max3, RUN2(T_add, T_max3)) KERNEL_S(k2_run4_add_max3, RUN4(T_add, T_max3))
KERNEL_S(k2_run8_add_max3, RUN8(T_add, T_max3)) KERNEL_S(k2_mix2_1_add_max3, MIX2_2x(T_add, T_max3))
#define RUN12_4(A, B) A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) A(8) A(9) A(10) A(11) B(12) B(13) B(14) B(15)
#define PAT_1_3(A, B) A(0) B(1) B(2) B(3) A(4) B(5) B(6) B(7) A(8) B(9) B(10) B(11) A(12) B(13) B(14) B(15)
KERNEL_S(k2_run12_4_add_max3, RUN12_4(T_add, T_max3)) KERNEL_S(k2_cmp_cnd3, PAT_1_3(T_cmp_lt, T_cndmask))
KERNEL_S(k2_run2_add_cnd, RUN2(T_add, T_cndmask)) KERNEL_S(k2_run4_add_cnd, RUN4(T_add, T_cndmask))
// longer runs: 16 + 16, 48 + 16 (75 % cheap), 16 + 48 (25 % cheap: the chain kernel's share) per iteration
KERNEL_S(k2_run16_16_add_max3, R16(T_add) R16(T_max3)) KERNEL_S(k2_run48_16_add_max3, R16(T_add) R16(T_add) R16(T_add) R16(T_max3))
KERNEL_S(k2_run16_48_add_max3, R16(T_add) R16(T_max3) R16(T_max3) R16(T_max3))
KERNEL_S(k2_run2_add_cmp, RUN2(T_add, T_cmp)) KERNEL_S(k2_run2_and_perm, RUN2(T_and, T_perm)) KERNEL_S(k2_run2_sub_mad24, RUN2(T_sub, T_mad24))
KERNEL_S(k2_run2_add_dpp, RUN2(T_add, T_dpp_add)) KERNEL_S(k2_run4_add_lshl_add, RUN4(T_add, T_lshl_add))
KERNEL(k_add, R16(T_add)) KERNEL(k_sub, R16(T_sub)) KERNEL(k_and, R16(T_and)) KERNEL(k_xor, R16(T_xor))
KERNEL(k_lshr, R16(T_lshr)) KERNEL(k_lshl, R16(T_lshl)) KERNEL(k_mov, R16(T_mov)) KERNEL(k_cndmask, R16(T_cndmask))
KERNEL(k_bitop3, R16(T_bitop3)) KERNEL(k_add3, R16(T_add3)) KERNEL(k_lshl_add, R16(T_lshl_add)) KERNEL(k_max3, R16(T_max3))
KERNEL(k_max, R16(T_max)) KERNEL(k_mad24, R16(T_mad24)) KERNEL(k_mul_lo, R16(T_mul_lo)) KERNEL(k_cmp, R16(T_cmp))
KERNEL(k_perm, R16(T_perm)) KERNEL(k_alignbit, R16(T_alignbit)) KERNEL(k_bfe, R16(T_bfe)) KERNEL(k_bcnt, R16(T_bcnt))
KERNEL(k_dot4, R16(T_dot4)) KERNEL(k_lerp, R16(T_lerp)) KERNEL(k_sad_u8, R16(T_sad_u8)) KERNEL(k_pk_add16, R16(T_pk_add16))
KERNEL(k_pk_max16, R16(T_pk_max16)) KERNEL(k_pk_lshr16, R16(T_pk_lshr16)) KERNEL(k_pk_mad16, R16(T_pk_mad16))
KERNEL(k_dpp_mov, R16(T_dpp_mov)) KERNEL(k_dpp_add, R16(T_dpp_add)) KERNEL(k_sdwa_lshl, R16(T_sdwa_lshl))
KERNEL(k_sdwa_add, R16(T_sdwa_add)) KERNEL(k_sdwa_and, R16(T_sdwa_and)) KERNEL(k_cvt_ub, R16(T_cvt_ub)) KERNEL(k_fma, R16(T_fma))
KERNEL(k_add_u16, R16(T_add_u16))
// mixes of a 2.4-cycle kind (add) with a 4-cycle kind (max3): alternating, 12 + 4, 15 + 1
KERNEL(k_alt_add_max3, ALT(T_add, T_max3)) KERNEL(k_12add_4max3, MIX12_4(T_add, T_max3)) KERNEL(k_15add_1max3, MIX15_1(T_add, T_max3))
KERNEL(k_alt_add_and, ALT(T_add, T_and)) KERNEL(k_alt_max3_mad24, ALT(T_max3, T_mad24))

// ---- the 4x4x4 i8 MFMA: per lane a 4 -> 4 linear map of the bytes of one VGPR (16 blocks of 4 lanes; lane i of a block
// holds row i of the 4x4 coefficient matrix) -- N_M MFMAs on independent accumulators beside N_V v_max3 per iteration ----
typedef int v4i __attribute__((ext_vector_type(4)));
template <int N_M, int N_V>
__global__ __launch_bounds__(256) void k_mfma_mix(uint32_t *out, uint32_t seed)
{
    v4i acc[8];
    for (int i = 0; i < 8; i++) acc[i] = v4i{(int)seed, i, 2, 3};
    uint32_t r[16];
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 17 + i + seed;
    const int a = (int)(0x04000000u >> (8 * (threadIdx.x & 3))), b = (int)(threadIdx.x * 0x01010101u + seed);
    const uint32_t x = seed * 3 + threadIdx.x, y = seed ^ 0x5555;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (j < N_V) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(r[j]) : "v"(x), "v"(y));
            if (j * N_M / 16 != (j + 1) * N_M / 16) {
                const int m = (j * N_M / 16) & 7;
                asm volatile("v_mfma_i32_4x4x4_16b_i8 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 16; i++) s ^= r[i];
    for (int i = 0; i < 8; i++) s ^= (uint32_t)(acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef void (*kfn)(uint32_t *, uint32_t);
struct Ent { const char *name; kfn f; double insts_per_iter; };
#define E(N) {#N, N, 16.0}
int main(int argc, char **argv)
{
    Ent ents[] = {E(k2_or), E(k2_ashr), E(k2_lshl1), E(k2_subrev), E(k2_not), E(k2_min), E(k2_and_or), E(k2_or3), E(k2_bfi),
                  E(k2_add_lshl), E(k2_mul24), E(k2_add_co), E(k2_sub_co), E(k2_addc_vcc), E(k2_addc_sgpr), E(k2_cndmask_vcc),
                  E(k2_cndmask_sgpr), E(k2_cndmask_e32), E(k2_cmp_sgpr), E(k2_alt_cmp_cndmask), E(k2_alt_cmpsgpr_cndmask),
                  E(k2_alt_add_cndmask), E(k2_mov_sgpr), E(k2_add_sgpr), E(k2_max3_sgpr), E(k2_pk_sub_sgpr), E(k2_fmac),
                  E(k2_mul_f32), E(k2_add_f32), E(k2_xad), E(k2_lshl_or), E(k2_run1_add_max3), E(k2_run2_add_max3),
                  E(k2_run4_add_max3), E(k2_run8_add_max3), E(k2_run12_4_add_max3), {"k2_run16_16_add_max3", k2_run16_16_add_max3, 32.0}, {"k2_run48_16_add_max3", k2_run48_16_add_max3, 64.0},
                  {"k2_run16_48_add_max3", k2_run16_48_add_max3, 64.0}, E(k2_cmp_cnd3), E(k2_run2_add_cnd), E(k2_run4_add_cnd), E(k2_mix2_1_add_max3), E(k2_run2_add_cmp), E(k2_run2_and_perm),
                  E(k2_run2_sub_mad24), E(k2_run2_add_dpp), E(k2_run4_add_lshl_add),
                  E(k_add), E(k_sub), E(k_and), E(k_xor), E(k_lshr), E(k_lshl), E(k_mov), E(k_cndmask), E(k_bitop3), E(k_add3),
                  E(k_lshl_add), E(k_max3), E(k_max), E(k_mad24), E(k_mul_lo), E(k_cmp), E(k_perm), E(k_alignbit), E(k_bfe),
                  E(k_bcnt), E(k_dot4), E(k_lerp), E(k_sad_u8), E(k_pk_add16), E(k_pk_max16), E(k_pk_lshr16), E(k_pk_mad16),
                  E(k_dpp_mov), E(k_dpp_add), E(k_sdwa_lshl), E(k_sdwa_add), E(k_sdwa_and), E(k_cvt_ub), E(k_fma), E(k_add_u16),
                  E(k_alt_add_max3), E(k_12add_4max3), E(k_15add_1max3), E(k_alt_add_and), E(k_alt_max3_mad24),
                  {"mfma_16_valu_0", k_mfma_mix<16, 0>, 16.0}, {"mfma_0_valu_16", k_mfma_mix<0, 16>, 16.0},
                  {"mfma_2_valu_16", k_mfma_mix<2, 16>, 18.0}, {"mfma_4_valu_16", k_mfma_mix<4, 16>, 20.0},
                  {"mfma_8_valu_16", k_mfma_mix<8, 16>, 24.0}, {"mfma_16_valu_16", k_mfma_mix<16, 16>, 32.0},
                  {"mfma_4_valu_8", k_mfma_mix<4, 8>, 12.0}};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int wps = argc > 2 ? atoi(argv[2]) : 8;  // waves per SIMD (one 256-thread block = one wave on each SIMD of a CU)
    const int blocks = prop.multiProcessorCount * wps;
    uint32_t *d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 40; w++) k_fma<<<blocks, 256>>>(d, 1);  // clock warm-up
    for (auto &en : ents) {
        if (argc > 1 && !strstr(en.name, argv[1])) continue;
        en.f<<<blocks, 256>>>(d, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; rep++) en.f<<<blocks, 256>>>(d, rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double iters = 5.0 * wps * ITER;  // per SIMD: wps waves x ITER iterations x 5 launches
        printf("%-20s %7.3f ms  %6.1f cycles/iteration/SIMD-wave  %.2f cycles per instruction (2.4 GHz)\n", en.name, ms,
               ms * 1e-3 * 2.4e9 / iters, ms * 1e-3 * 2.4e9 / (iters * en.insts_per_iter));
        fflush(stdout);
    }
    return 0;
}
