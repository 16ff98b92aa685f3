// How much dynamic LDS may one workgroup ask for on this device, and does it need the opt-in attribute?  (development probe)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) k(int *out, int words) { extern __shared__ int lds[]; for (int i = threadIdx.x; i < words; i += 1024) lds[i] = i; __syncthreads(); if (threadIdx.x == 0) out[blockIdx.x] = lds[words - 1]; }
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu  sharedMemPerBlockOptin %zu  sharedMemPerMultiprocessor %zu\n", p.sharedMemPerBlock, p.sharedMemPerBlockOptin, p.sharedMemPerMultiprocessor);
    int *out; hipMalloc(&out, 4096);
    for (int kb : {48, 64, 96, 128, 160}) {
        for (int optin = 0; optin < 2; ++optin) {
            hipError_t a = hipSuccess;
            if (optin) a = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
            hipLaunchKernelGGL(k, dim3(4), dim3(1024), (size_t)kb * 1024, 0, out, kb * 256);
            hipError_t l = hipGetLastError(), s = hipDeviceSynchronize();
            printf("%3d KB optin %d: attr %s | launch %s | sync %s\n", kb, optin, hipGetErrorString(a), hipGetErrorString(l), hipGetErrorString(s));
        }
    }
    return 0;
}
