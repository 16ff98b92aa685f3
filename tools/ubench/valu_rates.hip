// Micro-benchmark (development aid): issue cost of the VALU instructions the path tracer leans on, gfx950.
// Each kernel runs N iterations of 8 independent chains of one instruction; reports wave-instructions per cycle per SIMD
// at a given occupancy.  hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 4096;
static int g_repeat = 1;   // calibration mode: the kernel body is run g_repeat times longer (launch ramp and tail become negligible)

template <int OP> __global__ void __launch_bounds__(256) k(float *out, float seed, int repeat) {
    float a[8]; uint32_t u[8]; uint64_t w[8];
    if (OP == 21 && (threadIdx.x & 63) >= 32) return;   // half-empty waves: does the VALU skip the inactive half?
    if (OP == 22 && (threadIdx.x & 63) >= 16) return;   // a quarter
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; u[i] = (uint32_t)(threadIdx.x * 2654435761u + i); w[i] = u[i]; }
    for (int it = 0; it < ITERS * repeat; ++it) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
            if (OP == 1) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (OP == 2) a[i] = __builtin_amdgcn_sqrtf(a[i]);
            if (OP == 3) a[i] = __builtin_amdgcn_sinf(a[i]);
            if (OP == 4) { w[i] = (uint64_t)(uint32_t)w[i] * 0xD2511F53u + (w[i] >> 32); }      // v_mad_u64_u32
            if (OP == 5) u[i] = u[i] * 0xCD9E8D57u + 12345u;                                      // v_mul_lo_u32 (+add)
            if (OP == 6) u[i] = __umulhi(u[i], 0xCD9E8D57u) ^ u[i];                               // v_mul_hi_u32
            if (OP == 7) u[i] = (u[i] ^ (u[i] >> 7)) + 0x9E3779B9u;                               // cheap int
            if (OP == 8) a[i] = a[i] > 1.0f ? a[i] * 0.5f : a[i] + 0.25f;                         // cmp + cndmask-ish
            if (OP == 9) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7])); }
            if (OP == 11) { asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 7])); }      // VOP2, 4-byte encoding
            if (OP == 12) { asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(seed)); }
            if (OP == 13) { asm volatile("v_mul_f32_e64 %0, %1, %0" : "+v"(a[i]) : "v"(seed)); }                              // same op, 8-byte encoding
            if (OP == 14) { asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7])); }
            if (OP == 15) { asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]) : ); }
            if (OP == 16) { asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(seed)); }
            if (OP == 17) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed)); }                              // VOP3, no cross-chain operand
            if (OP == 18) { asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7])); }
            if (OP == 19) { asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(seed)); }
            if (OP == 20) { asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7])); }                 // VOP3 integer
            if (OP == 21 || OP == 22) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 7])); }
            if (OP == 10) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 7])); }   // one FMA per lane: never packed
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i] + (float)u[i] + (float)(w[i] & 0xffff);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> int run(const char *name, int instr_per_iter_per_chain) {
    const int blocks = 256 * 8;    // 8 blocks of 256 per CU = 8 waves/SIMD if registers allow
    float *out; CHK(hipMalloc(&out, blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.5f, g_repeat);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.5f, g_repeat);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = (double)blocks * 4 * ITERS * g_repeat * 8 * instr_per_iter_per_chain;
    const double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;   // 256 CU x 4 SIMD at 2.4 GHz nominal
    printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (nominal 2.4 GHz)\n", name, ms, simd_cycles / wave_instr);
    CHK(hipFree(out));
    return 0;
}
int main(int argc, char **argv) {
    // `valu_rates calib`: the pure v_fma_f32 loop alone, 64x longer, for reading SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE
    // under rocprofv3 --pmc: the loop that must read "VALU issue utilisation 1.0" 
    // (OP 0 is compiled to v_pk_fma_f32 -- two FMAs per lane and instruction -- by hipcc's SLP pass; OP 10 is the scalar v_fma_f32
    // the production kernels, built with -fno-slp-vectorize, are made of)
    if (argc > 1 && argv[1][0] == 'c') { g_repeat = 64; run<10>("v_fma_f32 (asm, calibration)", 1); run<0>("fmaf x8 (v_pk_fma_f32, calibration)", 1); return 0; }
    if (argc > 1 && argv[1][0] == 'e') {   // encodings: VOP2 (4 bytes) against VOP3 (8 bytes), long runs
        g_repeat = 16;
        run<11>("v_fmac_f32_e32", 1); run<10>("v_fma_f32 (cross operand)", 1); run<17>("v_fma_f32 (own chain)", 1); run<12>("v_mul_f32_e32", 1); run<13>("v_mul_f32_e64", 1);
        run<19>("v_add_f32_e32", 1); run<14>("v_mov_b32_e32", 1); run<15>("v_cndmask_b32_e32", 1); run<16>("v_max_f32_e32", 1); run<18>("v_xor_b32_e32", 1); run<20>("v_and_or_b32", 1);
        run<4>("v_mad_u64_u32", 1); run<1>("v_rcp_f32", 1);
        run<21>("v_fma_f32, lanes 0-31 only", 1); run<22>("v_fma_f32, lanes 0-15 only", 1);
        return 0;
    }
    run<0>("v_fma_f32", 1); run<1>("v_rcp_f32", 1); run<2>("v_sqrt_f32", 1); run<3>("v_sin_f32", 1);
    run<4>("v_mad_u64_u32", 1); run<5>("v_mul_lo_u32(+add)", 2); run<6>("v_mul_hi_u32(+xor)", 2); run<7>("xor/shift/add (3)", 3);
    run<8>("cmp+cndmask+mul+add", 4); run<9>("v_mul_u32_u24", 1); run<10>("v_fma_f32 (asm, unpacked)", 1);
    return 0;
}
