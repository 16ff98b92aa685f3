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
    return 0;
}
