// Micro-benchmark (development aid): how fast does one SIMD of gfx950 ISSUE vector instructions?  DESIGN.md 3.1 leans on "a wave64
// VALU instruction per 3.3 SIMD-cycles" (tools/ubench/valu_rates at 8 waves per SIMD), MI355X_MICROARCH.md says 2 cycles across
// waves and 4 for one wave alone.  This program measures the issue interval with the shader clock (s_memtime) for
//   * 1, 2, 4, 8 waves per SIMD (one 256-thread workgroup = one wave on each SIMD; the workgroups per CU are pinned by their LDS),
//   * instruction forms: three-source FMA with its sources in one / in different VGPR banks (register number mod 4), two-source
//     VOP2, an SGPR or an inline constant as a source, v_mov, and packed FP32,
//   * a loop body of 64 independent instructions (8 chains x 8), so that the loop's scalar instructions are 5 % of the stream.
// Output: SIMD-cycles per wave-instruction = (cycles a wave spent in the loop) / (its instructions x resident waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue.hip -o tools/ubench/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITERS = 16384;

// eight accumulators pinned to v[8..15] and two operands to chosen registers, so that the bank of every source is known
#define BODY8(INSTR) INSTR(8) INSTR(9) INSTR(10) INSTR(11) INSTR(12) INSTR(13) INSTR(14) INSTR(15)
#define REP8(X) X X X X X X X X

// second batch (round 3): one template per instruction form, D = accumulator register, A / B = operand registers
#define GROUP(T) T(8, 17, 18) T(9, 18, 19) T(10, 19, 16) T(11, 16, 17) T(12, 17, 18) T(13, 18, 19) T(14, 19, 16) T(15, 16, 17)
#define GROUP4(T) T(8, 9, 16, 17, 18, 19) T(10, 11, 16, 17, 18, 19) T(12, 13, 16, 17, 18, 19) T(14, 15, 16, 17, 18, 19)
#define ACC8 "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"
#define T_ADD_U32(D, A, B) "v_add_u32 v" #D ", v" #D ", v" #A "\n"
#define T_LSHL(D, A, B) "v_lshlrev_b32 v" #D ", 1, v" #D "\n"
#define T_LSHR(D, A, B) "v_lshrrev_b32 v" #D ", 1, v" #D "\n"
#define T_AND(D, A, B) "v_and_b32 v" #D ", v" #D ", v" #A "\n"
#define T_MUL_LO(D, A, B) "v_mul_lo_u32 v" #D ", v" #D ", v" #A "\n"
#define T_MUL_HI(D, A, B) "v_mul_hi_u32 v" #D ", v" #D ", v" #A "\n"
#define T_MAD64(D, E, A, B, C, F) "v_mad_u64_u32 v[" #D ":" #E "], vcc, v" #D ", v" #A ", v[" #C ":" #F "]\n"
#define T_RCP(D, A, B) "v_rcp_f32 v" #D ", v" #D "\n"
#define T_SQRT(D, A, B) "v_sqrt_f32 v" #D ", v" #D "\n"
#define T_RSQ(D, A, B) "v_rsq_f32 v" #D ", v" #D "\n"
#define T_SIN(D, A, B) "v_sin_f32 v" #D ", v" #D "\n"
#define T_EXP(D, A, B) "v_exp_f32 v" #D ", v" #D "\n"
#define T_CVT(D, A, B) "v_cvt_f32_u32 v" #D ", v" #D "\n"
#define T_BFE(D, A, B) "v_bfe_u32 v" #D ", v" #D ", v" #A ", v" #B "\n"
#define T_PERM(D, A, B) "v_perm_b32 v" #D ", v" #D ", v" #A ", v" #B "\n"
#define T_AND_OR(D, A, B) "v_and_or_b32 v" #D ", v" #D ", v" #A ", v" #B "\n"
#define T_LSHL_ADD(D, A, B) "v_lshl_add_u32 v" #D ", v" #D ", 1, v" #A "\n"
#define T_ADD3(D, A, B) "v_add3_u32 v" #D ", v" #D ", v" #A ", v" #B "\n"
#define T_PK_FMA_S(D, E, A, B, C, F) "v_pk_fma_f32 v[" #D ":" #E "], %0, v[" #A ":" #B "], v[" #D ":" #E "]\n"
#define T_PK_FMA_SEL(D, E, A, B, C, F) "v_pk_fma_f32 v[" #D ":" #E "], %0, v[" #A ":" #B "], v[" #D ":" #E "] op_sel_hi:[1,0,1]\n"
#define T_PK_MUL(D, E, A, B, C, F) "v_pk_mul_f32 v[" #D ":" #E "], v[" #D ":" #E "], v[" #A ":" #B "]\n"
#define T_PK_ADD(D, E, A, B, C, F) "v_pk_add_f32 v[" #D ":" #E "], v[" #D ":" #E "], v[" #C ":" #F "]\n"
#define T_PK_MUL_S(D, E, A, B, C, F) "v_pk_mul_f32 v[" #D ":" #E "], v[" #D ":" #E "], %0\n"
#define T_ADD_ABS(D, A, B) "v_add_f32 v" #D ", |v" #D "|, v" #A "\n"
#define T_FMA_NEG(D, A, B) "v_fma_f32 v" #D ", -v" #D ", v" #A ", v" #B "\n"
#define T_MOV_S(D, A, B) "v_mov_b32 v" #D ", %0\n"
#define T_MOV_LIT(D, A, B) "v_mov_b32 v" #D ", 0x12345678\n"
#define T_CND_VCC(D, A, B) "v_cndmask_b32 v" #D ", v" #D ", v" #A ", vcc\n"
#define T_MIN(D, A, B) "v_min_f32 v" #D ", v" #D ", v" #A "\n"
#define T_ADD_F32(D, A, B) "v_add_f32 v" #D ", v" #D ", v" #A "\n"
#define T_FMAC_S(D, A, B) "v_fmac_f32 v" #D ", %0, v" #A "\n"
#define T_CMP_CLASS(D, A, B) "v_cmp_class_f32 vcc, v" #D ", v" #A "\n"
#define T_CMP_U32(D, A, B) "v_cmp_ne_u32 vcc, v" #D ", v" #A "\n"
#define T_CMP_S(D, A, B) "v_cmp_lt_f32 vcc, %0, v" #D "\n"
#define T_MUL_DPP(D, A, B) "v_mul_f32_dpp v" #D ", v" #D ", v" #A " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define T_MUL_U24(D, A, B) "v_mul_u32_u24 v" #D ", v" #D ", v" #A "\n"
#define T_MAD_U24(D, A, B) "v_mad_u32_u24 v" #D ", v" #D ", v" #A ", v" #B "\n"
#define T_XOR_S(D, A, B) "v_xor_b32 v" #D ", %0, v" #D "\n"
#define T_BFI(D, A, B) "v_bfi_b32 v" #D ", v" #A ", v" #D ", v" #B "\n"
#define T_MUL_LEGACY(D, A, B) "v_mul_legacy_f32 v" #D ", v" #D ", v" #A "\n"
#define T_MAX_S(D, A, B) "v_max_f32 v" #D ", %0, v" #D "\n"
#define T_CVT_PK(D, A, B) "v_cvt_pkrtz_f16_f32 v" #D ", v" #D ", v" #A "\n"
#define T_FMA_F16(D, A, B) "v_pk_fma_f16 v" #D ", v" #D ", v" #A ", v" #B "\n"
#define T_LDEXP(D, A, B) "v_ldexp_f32 v" #D ", v" #D ", v" #A "\n"
#define T_FRACT(D, A, B) "v_fract_f32 v" #D ", v" #D "\n"
#define T_FLOOR(D, A, B) "v_floor_f32 v" #D ", v" #D "\n"
#define T_CMP_CND_VCC(D, A, B) "v_cmp_lt_f32 vcc, v" #D ", v" #A "\n v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n"
#define T_CMP_CND_S(D, A, B) "v_cmp_lt_f32 s[20:21], v" #D ", v" #A "\n v_cndmask_b32 v" #D ", v" #D ", v" #B ", s[20:21]\n"
#define T_CND_VCC_E64(D, A, B) "v_cndmask_b32_e64 v" #D ", v" #D ", v" #A ", vcc\n"
#define T_CND_VCC_OTHER(D, A, B) "v_cndmask_b32 v" #D ", v" #A ", v" #B ", vcc\n"
#define T_CMP_CND2(D, A, B) "v_cmp_lt_f32 vcc, v" #D ", v" #A "\n v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n v_cndmask_b32 v" #D ", v" #D ", v" #A ", vcc\n"
#define T_CMP_CND3(D, A, B) "v_cmp_lt_f32 vcc, v" #D ", v" #A "\n v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n v_cndmask_b32 v" #D ", v" #D ", v" #A ", vcc\n v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n"
#define T_CND_MUL(D, A, B) "v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n v_mul_f32 v" #D ", v" #D ", v" #A "\n"
#define T_CND_MUL3(D, A, B) "v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n v_mul_f32 v" #D ", v" #D ", v" #A "\n v_mul_f32 v" #D ", v" #D ", v" #A "\n v_mul_f32 v" #D ", v" #D ", v" #A "\n"
#define T_ADDC(D, A, B) "v_addc_co_u32 v" #D ", vcc, v" #D ", v" #A ", vcc\n"
#define T_CND_NOP(D, A, B) "v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n s_nop 0\n"
#define T_CND_NOP1(D, A, B) "v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n s_nop 1\n"
#define T_CND_VNOP(D, A, B) "v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n v_nop\n"
#define T_CND_E32_E64(D, A, B) "v_cndmask_b32_e32 v" #D ", v" #D ", v" #B ", vcc\n v_cndmask_b32_e64 v" #D ", v" #D ", v" #A ", s[20:21]\n"
#define T_CND_SALU(D, A, B) "v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n s_mov_b32 s22, s23\n"
#define T_CND_IMUL(D, A, B) "v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n v_mul_f32 v20, v20, v" #A "\n"
#define T_CND_IMOV(D, A, B) "v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n v_mov_b32 v20, v" #A "\n"
#define T_CND_CMPS(D, A, B) "v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n v_cmp_lt_f32 s[20:21], v" #D ", v" #A "\n"
#define T_CND2_MUL(D, A, B) "v_cndmask_b32 v" #D ", v" #D ", v" #B ", vcc\n v_cndmask_b32 v20, v20, v" #A ", vcc\n v_mul_f32 v" #D ", v" #D ", v" #A "\n"
#define T_MUL_SALU(D, A, B) "v_mul_f32 v" #D ", v" #D ", v" #A "\n s_add_i32 s22, s22, 1\n"
#define T_MUL_SALU2(D, A, B) "v_mul_f32 v" #D ", v" #D ", v" #A "\n s_add_i32 s22, s22, 1\n s_and_b64 s[20:21], s[20:21], exec\n"
#define T_MUL4_SALU(D, A, B) "v_mul_f32 v" #D ", v" #D ", v" #A "\n s_add_i32 s22, s22, 1\n v_max_f32 v" #D ", v" #D ", v" #B "\n"
#define M_MUL(D, A) "v_mul_f32 v" #D ", v" #D ", v" #A "\n"
#define M_MAX(D, B) "v_max_f32 v" #D ", v" #D ", v" #B "\n"
#define T_M1(D, A, B) M_MUL(D, A) M_MAX(D, B)
#define T_M2(D, A, B) M_MUL(D, A) M_MUL(D, B) M_MAX(D, B)
#define T_M3(D, A, B) M_MUL(D, A) M_MAX(D, A) M_MAX(D, B)
#define T_M4(D, A, B) M_MUL(D, A) "v_mul_f32 v" #D ", %0, v" #D "\n"
#define T_M5(D, A, B) M_MUL(D, A) "v_cmp_lt_f32 vcc, v" #D ", v" #B "\n"
#define T_M6(D, A, B) M_MUL(D, A) M_MUL(D, B) M_MUL(D, A) "v_rcp_f32 v" #D ", v" #D "\n"
#define T_M7(D, A, B) M_MUL(D, A) M_MUL(D, B) "s_add_i32 s22, s22, 1\n"
#define T_M8(D, A, B) M_MUL(D, A) M_MUL(D, B) M_MUL(D, A) M_MUL(D, B) "s_add_i32 s22, s22, 1\n"
#define T_M9(D, A, B) M_MAX(D, A) "s_add_i32 s22, s22, 1\n"
#define T_M10(D, A, B) M_MUL(D, A) "s_mov_b32 s22, s23\n"
#define T_M11(D, A, B) M_MAX(D, A) "s_mov_b32 s22, s23\n s_mov_b32 s24, s23\n"
#define T_M12(D, A, B) M_MUL(D, A) "s_mov_b32 s22, s23\n s_mov_b32 s24, s23\n"
#define T_M13(D, A, B) M_MUL(D, A) M_MUL(D, B) M_MUL(D, A) M_MUL(D, B) M_MUL(D, A) M_MUL(D, B) M_MUL(D, A) "v_rcp_f32 v" #D ", v" #D "\n"
#define T_M14(D, A, B) "v_rcp_f32 v20, v" #A "\n" M_MUL(D, A) M_MUL(D, B) M_MUL(D, A)
#define T_M15(D, A, B) "v_rcp_f32 v20, v" #A "\n" M_MAX(D, A) M_MUL(D, B) M_MAX(D, A)
#define T_M16(D, A, B) "v_rcp_f32 v20, v" #A "\n v_rcp_f32 v21, v" #B "\n" M_MUL(D, A) M_MUL(D, B) M_MUL(D, A) M_MUL(D, B) M_MUL(D, A) M_MUL(D, B)
#define T_M17(D, A, B) "v_sqrt_f32 v20, v" #A "\n" M_MUL(D, A) M_MUL(D, B) M_MUL(D, A) M_MUL(D, B) M_MUL(D, A) M_MUL(D, B) M_MUL(D, A)
#define CASE_M(N, T) if (OP == N) { asm volatile(REP8(GROUP(T)) :: "s"(seed) : ACC8, "vcc", "scc", "s22", "s23", "s24"); }
#define CASE_V(N, T) if (OP == N) { asm volatile(REP8(GROUP(T)) ::: ACC8, "vcc"); }
#define CASE_S(N, T) if (OP == N) { asm volatile(REP8(GROUP(T)) :: "s"(seed) : ACC8, "vcc"); }
#define CASE_P(N, T) if (OP == N) { asm volatile(REP8(GROUP4(T)) ::: ACC8, "vcc"); }
#define CASE_PS(N, T) if (OP == N) { asm volatile(REP8(GROUP4(T)) :: "s"(mask) : ACC8, "vcc"); }
template <int OP> __global__ void __launch_bounds__(256) k(unsigned long long *cycles, float *sink, float seed) {
    extern __shared__ int pin[];
    asm volatile("v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n"
                 "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 1.0 \n v_mov_b32 v23, 0\n"
                 :: "v"(seed) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
    __syncthreads();
    unsigned long long mask = __builtin_amdgcn_ballot_w64((threadIdx.x & 1) != 0);
    asm volatile("v_cmp_lt_f32 vcc, v17, v16" ::: "vcc");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        // dst = dst * v16 + v17 style: acc in v8..v15 (banks 0,1,2,3,0,1,2,3)
        if (OP == 0) { asm volatile(REP8("v_fma_f32 v8, v8, v16, v17\n v_fma_f32 v9, v9, v16, v17\n v_fma_f32 v10, v10, v16, v17\n v_fma_f32 v11, v11, v16, v17\n v_fma_f32 v12, v12, v16, v17\n v_fma_f32 v13, v13, v16, v17\n v_fma_f32 v14, v14, v16, v17\n v_fma_f32 v15, v15, v16, v17\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        // every instruction's three sources in three different banks: acc bank b, operands in banks b+1, b+2
        if (OP == 1) { asm volatile(REP8("v_fma_f32 v8, v8, v17, v18\n v_fma_f32 v9, v9, v18, v19\n v_fma_f32 v10, v10, v19, v16\n v_fma_f32 v11, v11, v16, v17\n v_fma_f32 v12, v12, v17, v18\n v_fma_f32 v13, v13, v18, v19\n v_fma_f32 v14, v14, v19, v16\n v_fma_f32 v15, v15, v16, v17\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        // all three sources in ONE bank (v8,v16,v20 are bank 0 ...)
        if (OP == 2) { asm volatile(REP8("v_fma_f32 v8, v8, v16, v20\n v_fma_f32 v9, v9, v17, v21\n v_fma_f32 v10, v10, v18, v22\n v_fma_f32 v11, v11, v19, v23\n v_fma_f32 v12, v12, v16, v20\n v_fma_f32 v13, v13, v17, v21\n v_fma_f32 v14, v14, v18, v22\n v_fma_f32 v15, v15, v19, v23\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 3) { asm volatile(REP8("v_fmac_f32 v8, v17, v18\n v_fmac_f32 v9, v18, v19\n v_fmac_f32 v10, v19, v16\n v_fmac_f32 v11, v16, v17\n v_fmac_f32 v12, v17, v18\n v_fmac_f32 v13, v18, v19\n v_fmac_f32 v14, v19, v16\n v_fmac_f32 v15, v16, v17\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 4) { asm volatile(REP8("v_mul_f32 v8, v8, v17\n v_mul_f32 v9, v9, v18\n v_mul_f32 v10, v10, v19\n v_mul_f32 v11, v11, v16\n v_mul_f32 v12, v12, v17\n v_mul_f32 v13, v13, v18\n v_mul_f32 v14, v14, v19\n v_mul_f32 v15, v15, v16\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 5) { asm volatile(REP8("v_mul_f32 v8, 1.0, v8\n v_mul_f32 v9, 1.0, v9\n v_mul_f32 v10, 1.0, v10\n v_mul_f32 v11, 1.0, v11\n v_mul_f32 v12, 1.0, v12\n v_mul_f32 v13, 1.0, v13\n v_mul_f32 v14, 1.0, v14\n v_mul_f32 v15, 1.0, v15\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 6) { asm volatile(REP8("v_mul_f32 v8, %0, v8\n v_mul_f32 v9, %0, v9\n v_mul_f32 v10, %0, v10\n v_mul_f32 v11, %0, v11\n v_mul_f32 v12, %0, v12\n v_mul_f32 v13, %0, v13\n v_mul_f32 v14, %0, v14\n v_mul_f32 v15, %0, v15\n") :: "s"(seed) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 7) { asm volatile(REP8("v_mov_b32 v8, v17\n v_mov_b32 v9, v18\n v_mov_b32 v10, v19\n v_mov_b32 v11, v16\n v_mov_b32 v12, v17\n v_mov_b32 v13, v18\n v_mov_b32 v14, v19\n v_mov_b32 v15, v16\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 8) { asm volatile(REP8("v_pk_fma_f32 v[8:9], v[8:9], v[16:17], v[18:19]\n v_pk_fma_f32 v[10:11], v[10:11], v[16:17], v[18:19]\n v_pk_fma_f32 v[12:13], v[12:13], v[16:17], v[18:19]\n v_pk_fma_f32 v[14:15], v[14:15], v[16:17], v[18:19]\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 9) { asm volatile(REP8("v_fma_f32 v8, v8, 1.0, 0\n v_fma_f32 v9, v9, 1.0, 0\n v_fma_f32 v10, v10, 1.0, 0\n v_fma_f32 v11, v11, 1.0, 0\n v_fma_f32 v12, v12, 1.0, 0\n v_fma_f32 v13, v13, 1.0, 0\n v_fma_f32 v14, v14, 1.0, 0\n v_fma_f32 v15, v15, 1.0, 0\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        // dependent chain: every instruction reads the previous one's result
        if (OP == 10) { asm volatile(REP8("v_fma_f32 v8, v8, v17, v18\n v_fma_f32 v8, v8, v17, v18\n v_fma_f32 v8, v8, v17, v18\n v_fma_f32 v8, v8, v17, v18\n v_fma_f32 v8, v8, v17, v18\n v_fma_f32 v8, v8, v17, v18\n v_fma_f32 v8, v8, v17, v18\n v_fma_f32 v8, v8, v17, v18\n") ::: "v8"); }
        if (OP == 11) { asm volatile(REP8("v_xor_b32 v8, v8, v17\n v_xor_b32 v9, v9, v18\n v_xor_b32 v10, v10, v19\n v_xor_b32 v11, v11, v16\n v_xor_b32 v12, v12, v17\n v_xor_b32 v13, v13, v18\n v_xor_b32 v14, v14, v19\n v_xor_b32 v15, v15, v16\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 15) { asm volatile(REP8("v_fma_f32 v8, v8, %0, v17\n v_fma_f32 v9, v9, %0, v18\n v_fma_f32 v10, v10, %0, v19\n v_fma_f32 v11, v11, %0, v16\n v_fma_f32 v12, v12, %0, v17\n v_fma_f32 v13, v13, %0, v18\n v_fma_f32 v14, v14, %0, v19\n v_fma_f32 v15, v15, %0, v16\n") :: "s"(seed) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 16) { asm volatile(REP8("v_sub_f32 v8, %0, v8\n v_sub_f32 v9, %0, v9\n v_sub_f32 v10, %0, v10\n v_sub_f32 v11, %0, v11\n v_sub_f32 v12, %0, v12\n v_sub_f32 v13, %0, v13\n v_sub_f32 v14, %0, v14\n v_sub_f32 v15, %0, v15\n") :: "s"(seed) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 17) { asm volatile(REP8("v_cndmask_b32 v8, v8, v17, %0\n v_cndmask_b32 v9, v9, v18, %0\n v_cndmask_b32 v10, v10, v19, %0\n v_cndmask_b32 v11, v11, v16, %0\n v_cndmask_b32 v12, v12, v17, %0\n v_cndmask_b32 v13, v13, v18, %0\n v_cndmask_b32 v14, v14, v19, %0\n v_cndmask_b32 v15, v15, v16, %0\n") :: "s"(mask) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 18) { asm volatile(REP8("v_mul_f32 v8, 0x40490fdb, v8\n v_mul_f32 v9, 0x40490fdb, v9\n v_mul_f32 v10, 0x40490fdb, v10\n v_mul_f32 v11, 0x40490fdb, v11\n v_mul_f32 v12, 0x40490fdb, v12\n v_mul_f32 v13, 0x40490fdb, v13\n v_mul_f32 v14, 0x40490fdb, v14\n v_mul_f32 v15, 0x40490fdb, v15\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 19) { asm volatile(REP8("v_cmp_lt_f32 %0, v8, v17\n v_cmp_lt_f32 %0, v9, v18\n v_cmp_lt_f32 %0, v10, v19\n v_cmp_lt_f32 %0, v11, v16\n v_cmp_lt_f32 %0, v12, v17\n v_cmp_lt_f32 %0, v13, v18\n v_cmp_lt_f32 %0, v14, v19\n v_cmp_lt_f32 %0, v15, v16\n") : "=s"(mask)); }
        if (OP == 20) { asm volatile(REP8("v_max_f32 v8, v8, v17\n v_max_f32 v9, v9, v18\n v_max_f32 v10, v10, v19\n v_max_f32 v11, v11, v16\n v_max_f32 v12, v12, v17\n v_max_f32 v13, v13, v18\n v_max_f32 v14, v14, v19\n v_max_f32 v15, v15, v16\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 21) { asm volatile(REP8("v_med3_f32 v8, v8, v17, v18\n v_med3_f32 v9, v9, v18, v19\n v_med3_f32 v10, v10, v19, v16\n v_med3_f32 v11, v11, v16, v17\n v_med3_f32 v12, v12, v17, v18\n v_med3_f32 v13, v13, v18, v19\n v_med3_f32 v14, v14, v19, v16\n v_med3_f32 v15, v15, v16, v17\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 22) { asm volatile(REP8("ds_read_b128 v[24:27], %0\n v_mul_f32 v8, v8, v17\n v_mul_f32 v9, v9, v18\n v_mul_f32 v10, v10, v19\n v_mul_f32 v11, v11, v16\n ds_read_b128 v[28:31], %0 offset:16\n v_mul_f32 v12, v12, v17\n v_mul_f32 v13, v13, v18\n v_mul_f32 v14, v14, v19\n v_mul_f32 v15, v15, v16\n s_waitcnt lgkmcnt(0)\n") :: "v"(0) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31"); }
        if (OP == 12) { asm volatile(REP8("v_cndmask_b32 v8, v8, v17, vcc\n v_cndmask_b32 v9, v9, v18, vcc\n v_cndmask_b32 v10, v10, v19, vcc\n v_cndmask_b32 v11, v11, v16, vcc\n v_cndmask_b32 v12, v12, v17, vcc\n v_cndmask_b32 v13, v13, v18, vcc\n v_cndmask_b32 v14, v14, v19, vcc\n v_cndmask_b32 v15, v15, v16, vcc\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 13) { asm volatile(REP8("v_max3_f32 v8, v8, v17, v18\n v_max3_f32 v9, v9, v18, v19\n v_max3_f32 v10, v10, v19, v16\n v_max3_f32 v11, v11, v16, v17\n v_max3_f32 v12, v12, v17, v18\n v_max3_f32 v13, v13, v18, v19\n v_max3_f32 v14, v14, v19, v16\n v_max3_f32 v15, v15, v16, v17\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15"); }
        if (OP == 14) { asm volatile(REP8("v_cmp_lt_f32 vcc, v8, v17\n v_cmp_lt_f32 vcc, v9, v18\n v_cmp_lt_f32 vcc, v10, v19\n v_cmp_lt_f32 vcc, v11, v16\n v_cmp_lt_f32 vcc, v12, v17\n v_cmp_lt_f32 vcc, v13, v18\n v_cmp_lt_f32 vcc, v14, v19\n v_cmp_lt_f32 vcc, v15, v16\n") ::: "vcc"); }
        CASE_V(30, T_ADD_U32) CASE_V(31, T_LSHL) CASE_V(32, T_LSHR) CASE_V(33, T_AND) CASE_V(34, T_MUL_LO) CASE_V(35, T_MUL_HI) CASE_P(36, T_MAD64)
        CASE_V(37, T_RCP) CASE_V(38, T_SQRT) CASE_V(39, T_RSQ) CASE_V(40, T_SIN) CASE_V(41, T_EXP) CASE_V(42, T_CVT) CASE_V(43, T_BFE) CASE_V(44, T_PERM)
        CASE_V(45, T_AND_OR) CASE_V(46, T_LSHL_ADD) CASE_V(47, T_ADD3) CASE_PS(48, T_PK_FMA_S) CASE_PS(49, T_PK_FMA_SEL) CASE_P(50, T_PK_MUL) CASE_P(51, T_PK_ADD)
        CASE_PS(52, T_PK_MUL_S) CASE_V(53, T_ADD_ABS) CASE_V(54, T_FMA_NEG) CASE_S(55, T_MOV_S) CASE_V(56, T_MOV_LIT) CASE_V(57, T_CND_VCC) CASE_V(58, T_MIN)
        CASE_V(59, T_ADD_F32) CASE_S(60, T_FMAC_S) CASE_V(61, T_CMP_CLASS) CASE_V(62, T_CMP_U32) CASE_S(63, T_CMP_S) CASE_V(64, T_MUL_DPP) CASE_V(65, T_MUL_U24)
        CASE_V(66, T_MAD_U24) CASE_S(67, T_XOR_S) CASE_V(68, T_BFI) CASE_V(69, T_MUL_LEGACY) CASE_S(70, T_MAX_S) CASE_V(71, T_CVT_PK) CASE_V(72, T_FMA_F16)
        CASE_V(76, T_CMP_CND_VCC) CASE_V(78, T_CND_VCC_E64) CASE_V(79, T_CND_VCC_OTHER)
        if (OP == 77) { asm volatile(REP8(GROUP(T_CMP_CND_S)) ::: ACC8, "s20", "s21"); }
        CASE_V(80, T_CMP_CND2) CASE_V(81, T_CMP_CND3) CASE_V(82, T_CND_MUL) CASE_V(83, T_CND_MUL3) CASE_V(84, T_ADDC)
        CASE_V(85, T_CND_NOP) CASE_V(86, T_CND_NOP1) CASE_V(87, T_CND_VNOP)
        if (OP == 88) { asm volatile(REP8(GROUP(T_CND_E32_E64)) ::: ACC8, "s20", "s21"); }
        if (OP == 89) { asm volatile(REP8(GROUP(T_CND_SALU)) ::: ACC8, "s22", "s23"); }
        if (OP == 90) { asm volatile(REP8(GROUP(T_CND_IMUL)) ::: ACC8, "v20"); }
        if (OP == 91) { asm volatile(REP8(GROUP(T_CND_IMOV)) ::: ACC8, "v20"); }
        if (OP == 92) { asm volatile(REP8(GROUP(T_CND_CMPS)) ::: ACC8, "s20", "s21"); }
        if (OP == 93) { asm volatile(REP8(GROUP(T_CND2_MUL)) ::: ACC8, "v20"); }
        if (OP == 94) { asm volatile(REP8(GROUP(T_MUL_SALU)) ::: ACC8, "s22", "scc"); }
        if (OP == 95) { asm volatile(REP8(GROUP(T_MUL_SALU2)) ::: ACC8, "s20", "s21", "s22", "scc"); }
        if (OP == 96) { asm volatile(REP8(GROUP(T_MUL4_SALU)) ::: ACC8, "s22", "scc"); }
        CASE_M(100, T_M1) CASE_M(101, T_M2) CASE_M(102, T_M3) CASE_M(103, T_M4) CASE_M(104, T_M5) CASE_M(105, T_M6) CASE_M(106, T_M7) CASE_M(107, T_M8)
        CASE_M(108, T_M9) CASE_M(109, T_M10) CASE_M(110, T_M11) CASE_M(111, T_M12)
        if (OP == 112) { asm volatile(REP8(GROUP(T_M13)) ::: ACC8); }
        if (OP == 113) { asm volatile(REP8(GROUP(T_M14)) ::: ACC8, "v20"); }
        if (OP == 114) { asm volatile(REP8(GROUP(T_M15)) ::: ACC8, "v20"); }
        if (OP == 115) { asm volatile(REP8(GROUP(T_M16)) ::: ACC8, "v20", "v21"); }
        if (OP == 116) { asm volatile(REP8(GROUP(T_M17)) ::: ACC8, "v20"); }
        CASE_V(73, T_LDEXP) CASE_V(74, T_FRACT) CASE_V(75, T_FLOOR)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s; asm volatile("v_add_f32 %0, v8, v9\n v_add_f32 %0, %0, v10\n v_add_f32 %0, %0, v11\n v_add_f32 %0, %0, v12\n v_add_f32 %0, %0, v13\n v_add_f32 %0, %0, v14\n v_add_f32 %0, %0, v15\n" : "=v"(s));
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (s == 12345.678f) sink[0] = s + (float)pin[0];
}
template <int OP> int run(const char *name, int instr_per_iter) {
    printf("%-44s", name);
    for (int waves = 1; waves <= 8; waves *= 2) {
        const int blocks = 256 * waves;                               // `waves` workgroups of 4 waves on every CU
        const size_t lds = (size_t)(160 * 1024 / waves) - 1024;       // ... and no more: the LDS of a CU is full with them
        unsigned long long *cyc; float *sink;
        CHK(hipMalloc(&cyc, blocks * 4 * sizeof(unsigned long long))); CHK(hipMalloc(&sink, 64));
        CHK(hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, cyc, sink, 1.0f);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, cyc, sink, 1.0f);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks * 4);
        CHK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double sum = 0; for (auto c : h) sum += (double)c;
        const double per_wave = sum / h.size();
        // [in-wave cycles per instruction x assumed residency | wall-clock: SIMD-cycles at 2.4 GHz per instruction | wave lifetime / launch time]
        const double wall_cyc = ms * 1e-3 * 2.4e9;
        printf("  %dw: %5.2f|%5.2f|%4.2f", waves, per_wave / ((double)ITERS * instr_per_iter * waves), wall_cyc / ((double)ITERS * instr_per_iter * waves), per_wave / wall_cyc);
        CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
        CHK(hipFree(cyc)); CHK(hipFree(sink));
    }
    printf("\n");
    return 0;
}
int main() {
    printf("SIMD-cycles per wave64 instruction (s_memtime), by resident waves per SIMD\n");
    run<0>("v_fma_f32 acc*v16+v17 (banks b,0,1)", 64);
    run<1>("v_fma_f32, three sources in three banks", 64);
    run<2>("v_fma_f32, three sources in ONE bank", 64);
    run<9>("v_fma_f32 acc*1.0+0 (inline constants)", 64);
    run<3>("v_fmac_f32 (VOP2: acc += a*b)", 64);
    run<4>("v_mul_f32 (VOP2, two VGPR sources)", 64);
    run<5>("v_mul_f32 (inline constant source)", 64);
    run<6>("v_mul_f32 (SGPR source)", 64);
    run<7>("v_mov_b32", 64);
    run<11>("v_xor_b32", 64);
    run<12>("v_cndmask_b32 (vcc)", 64);
    run<13>("v_max3_f32", 64);
    run<14>("v_cmp_lt_f32 -> vcc", 64);
    run<15>("v_fma_f32 with an SGPR source", 64);
    run<16>("v_sub_f32 (SGPR - VGPR)", 64);
    run<18>("v_mul_f32 with a 32-bit literal", 64);
    run<17>("v_cndmask_b32 (SGPR-pair mask)", 64);
    run<19>("v_cmp_lt_f32 -> SGPR pair", 64);
    run<20>("v_max_f32 (VOP2)", 64);
    run<21>("v_med3_f32", 64);
    run<22>("8 v_mul_f32 + 2 ds_read_b128 broadcast (per 8)", 64);
    run<8>("v_pk_fma_f32 (32 per iteration)", 32);
    run<10>("v_fma_f32, ONE dependent chain", 64);
    printf("-- second batch --\n");
    run<59>("v_add_f32 (VOP2)", 64); run<58>("v_min_f32 (VOP2)", 64); run<53>("v_add_f32 with |x| modifier (VOP3)", 64); run<54>("v_fma_f32 with a neg modifier", 64);
    run<69>("v_mul_legacy_f32", 64); run<73>("v_ldexp_f32", 64); run<74>("v_fract_f32", 64); run<75>("v_floor_f32", 64);
    run<60>("v_fmac_f32 with an SGPR source", 64); run<70>("v_max_f32 with an SGPR source", 64); run<67>("v_xor_b32 with an SGPR source", 64);
    run<55>("v_mov_b32 from an SGPR", 64); run<56>("v_mov_b32 of a 32-bit literal", 64);
    run<57>("v_cndmask_b32 (vcc), vcc declared", 64); run<61>("v_cmp_class_f32 -> vcc", 64); run<62>("v_cmp_ne_u32 -> vcc", 64); run<63>("v_cmp_lt_f32 (SGPR source) -> vcc", 64);
    run<76>("v_cmp -> vcc + v_cndmask (vcc) pairs (per instr)", 128); run<77>("v_cmp -> s[20:21] + v_cndmask (s[20:21]) pairs", 128);
    run<78>("v_cndmask_b32_e64 (vcc)", 64); run<79>("v_cndmask_b32 (vcc), dst not a source", 64);
    run<80>("v_cmp -> vcc + 2 v_cndmask (vcc) (per instr)", 192); run<81>("v_cmp -> vcc + 3 v_cndmask (vcc) (per instr)", 256);
    run<82>("v_cndmask (vcc) + v_mul pairs (per instr)", 128); run<83>("v_cndmask (vcc) + 3 v_mul (per instr)", 256); run<84>("v_addc_co_u32 (vcc in and out)", 64);
    run<85>("v_cndmask (vcc) + s_nop 0 (per cndmask)", 64); run<86>("v_cndmask (vcc) + s_nop 1 (per cndmask)", 64); run<87>("v_cndmask (vcc) + v_nop (per cndmask)", 64);
    run<88>("v_cndmask e32 (vcc) + e64 (s[20:21]) pairs (per instr)", 128); run<89>("v_cndmask (vcc) + s_mov_b32 (per cndmask)", 64);
    run<90>("v_cndmask (vcc) + independent v_mul (per instr)", 128); run<91>("v_cndmask (vcc) + independent v_mov (per instr)", 128);
    run<92>("v_cndmask (vcc) + v_cmp -> s[20:21] (per instr)", 128); run<93>("2 v_cndmask (vcc) + v_mul (per instr)", 192);
    run<94>("v_mul_f32 + 1 SALU each (per v_mul)", 64); run<95>("v_mul_f32 + 2 SALU each (per v_mul)", 64); run<96>("v_mul + s_add + v_max (per VALU instr)", 128);
    printf("-- mixes: cycles per GROUP of the instructions named --\n");
    run<100>("mix: v_mul + v_max", 64); run<101>("mix: 2 v_mul + v_max", 64); run<102>("mix: v_mul + 2 v_max", 64); run<103>("mix: v_mul + v_mul(SGPR)", 64);
    run<104>("mix: v_mul + v_cmp", 64); run<105>("mix: 3 v_mul + v_rcp", 64); run<106>("mix: 2 v_mul + s_add", 64); run<107>("mix: 4 v_mul + s_add", 64);
    run<108>("mix: v_max + s_add", 64); run<109>("mix: v_mul + s_mov", 64); run<110>("mix: v_max + 2 s_mov", 64); run<111>("mix: v_mul + 2 s_mov", 64);
    run<112>("mix: 7 v_mul + dependent v_rcp", 64); run<113>("mix: independent v_rcp + 3 v_mul", 64); run<114>("mix: independent v_rcp + max, mul, max", 64);
    run<115>("mix: 2 independent v_rcp + 6 v_mul", 64); run<116>("mix: independent v_sqrt + 7 v_mul", 64);
    run<30>("v_add_u32", 64); run<31>("v_lshlrev_b32", 64); run<32>("v_lshrrev_b32", 64); run<33>("v_and_b32", 64); run<43>("v_bfe_u32", 64); run<68>("v_bfi_b32", 64); run<44>("v_perm_b32", 64);
    run<45>("v_and_or_b32", 64); run<46>("v_lshl_add_u32", 64); run<47>("v_add3_u32", 64);
    run<34>("v_mul_lo_u32", 64); run<35>("v_mul_hi_u32", 64); run<65>("v_mul_u32_u24", 64); run<66>("v_mad_u32_u24", 64); run<36>("v_mad_u64_u32 (32 per iteration)", 32);
    run<37>("v_rcp_f32", 64); run<38>("v_sqrt_f32", 64); run<39>("v_rsq_f32", 64); run<40>("v_sin_f32", 64); run<41>("v_exp_f32", 64); run<42>("v_cvt_f32_u32", 64); run<71>("v_cvt_pkrtz_f16_f32", 64);
    run<64>("v_mul_f32 with a DPP quad_perm source", 64);
    run<50>("v_pk_mul_f32 (32 per iteration)", 32); run<51>("v_pk_add_f32 (32)", 32); run<48>("v_pk_fma_f32, SGPR-pair source (32)", 32); run<49>("v_pk_fma_f32, SGPR pair + op_sel_hi (32)", 32);
    run<52>("v_pk_mul_f32, SGPR-pair source (32)", 32); run<72>("v_pk_fma_f16 (64)", 64);
    return 0;
}
