// Micro-benchmark (development aid): what a scattered 64-byte record fetch costs the vector-memory path of a CU on gfx950.
//   mode 0: every active lane fetches its own record with four global_load_dwordx4 (what bvh_closest_hit does)
//   mode 1: a quad of lanes fetches ONE record with one global_load_dwordx4 (lane q of the quad takes piece q): 16 records per instruction
//   mode 2: like 0, but one dwordx4 per lane only (16-byte records)
// `active` lanes of each wave take part (the others are masked off, like rays that have finished).  Table of `n_rec` records:
// 4 K records = 256 KB (L2 resident, the final scene's tree), 1.6 M = 100 MB (the 1M-sphere tree).
// hipcc --offload-arch=gfx950 -O3 gather_rates.hip -o gather_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE> __global__ void __launch_bounds__(64) k(const float4 *__restrict__ tab, uint32_t n_rec, int active, int iters, float *out) {
    const int lane = threadIdx.x;
    uint32_t state = (blockIdx.x * 64u + lane) * 2654435761u + 12345u;
    float acc = 0.0f;
    if (MODE == 1) {
        // 16 records per instruction: `active` records per wave-round take ceil(active / 16) instructions; quad q of round r
        // fetches record r * 16 + q of the round (if there is one), lane p of the quad its p-th 16 bytes
        const int rounds = (active + 15) / 16, quad = lane >> 2;
        uint32_t s = (blockIdx.x * 64u + lane) * 2654435761u + 12345u;
        for (int it = 0; it < iters; ++it) {
            for (int r = 0; r < rounds; ++r) {
                s = s * 1664525u + 1013904223u;
                const uint32_t rec = (__shfl(s, quad * 4, 64) >> 8) % n_rec;     // the quad agrees on one record
                if (r * 16 + quad < active) {
                    const float4 v = tab[(size_t)rec * 4 + (lane & 3)];
                    acc += v.x + v.w;
                    s ^= __float_as_uint(v.y) & 1u;     // the next address depends on the data, as in a traversal
                }
            }
        }
    } else if (lane < active) {
        for (int it = 0; it < iters; ++it) {
            state = state * 1664525u + 1013904223u;
            const uint32_t rec = (state >> 8) % n_rec;
            const float4 *p = tab + (size_t)rec * 4;
            if (MODE == 0) { const float4 a = p[0], b = p[1], c = p[2], d = p[3]; acc += a.x + b.y + c.z + d.w; state ^= __float_as_uint(a.y) & 1u; }
            else { const float4 a = p[0]; acc += a.x + a.w; state ^= __float_as_uint(a.y) & 1u; }
        }
    }
    out[blockIdx.x * 64 + lane] = acc;
}

template <int MODE> int run(const char *name, const float4 *tab, uint32_t n_rec, int active, float *out) {
    const int blocks = 256 * 32, iters = 256;      // 32 one-wave workgroups per CU = 8 waves per SIMD
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, tab, n_rec, active, iters, out);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, tab, n_rec, active, iters, out);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double records = (double)blocks * iters * active;
    printf("%-34s table %7.1f MB active %2d: %7.3f ms  %7.2f G records/s  %6.1f CU-cycles per wave-round (2.4 GHz)\n", name, n_rec * 64.0 / 1e6, active, ms,
           records / ms * 1e-6, ms * 1e-3 * 2.4e9 * 256 / ((double)blocks * iters));
    return 0;
}
int main() {
    const uint32_t sizes[2] = {4096u, 1600000u};
    float *out; CHK(hipMalloc(&out, 256 * 32 * 64 * sizeof(float)));
    for (uint32_t n_rec : sizes) {
        float4 *tab; CHK(hipMalloc(&tab, (size_t)n_rec * 64));
        CHK(hipMemset(tab, 0, (size_t)n_rec * 64));
        for (int active : {64, 28, 15}) {
            run<0>("own record, 4 x dwordx4", tab, n_rec, active, out);
            run<1>("quad per record, 1 x dwordx4", tab, n_rec, active, out);
            run<2>("own 16-byte record, 1 x dwordx4", tab, n_rec, active, out);
        }
        CHK(hipFree(tab));
    }
    return 0;
}
