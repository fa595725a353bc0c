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
    Ent ents[] = {E(k_add), E(k_sub), E(k_and), E(k_xor), E(k_lshr), E(k_lshl), E(k_mov), E(k_cndmask), E(k_bitop3), E(k_add3),
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
    const int blocks = prop.multiProcessorCount * 8;
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
        const double iters = 5.0 * 8 * ITER;  // per SIMD: 8 waves x ITER iterations x 5 launches
        printf("%-20s %7.3f ms  %6.1f cycles/iteration/SIMD-wave  %.2f cycles per instruction (2.4 GHz)\n", en.name, ms,
               ms * 1e-3 * 2.4e9 / iters, ms * 1e-3 * 2.4e9 / (iters * en.insts_per_iter));
        fflush(stdout);
    }
    return 0;
}
