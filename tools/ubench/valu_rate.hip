// Micro-benchmark: issue rate of 32-bit integer VALU and SALU instructions on one CU (wave64), to ground
// the "instruction-issue bound" estimates of DESIGN.md.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 4096
__global__ void k_valu(uint32_t* out, uint32_t n, uint32_t lanes)
{
    uint32_t a = threadIdx.x, b = a * 3u, c = a ^ 5u, d = a + 7u;
    if ((threadIdx.x & 63u) >= lanes) { out[blockIdx.x * blockDim.x + threadIdx.x] = 0; return; }
    for (uint32_t i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < REP / 4; ++u) {
            asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
__global__ void k_salu(uint32_t* out, uint32_t n)
{
    uint32_t a = blockIdx.x, b = a * 3u, c = a ^ 5u, d = a + 7u;
    for (uint32_t i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < REP / 4; ++u) {
            asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0" : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
int main()
{
    uint32_t* out;
    hipMalloc((void**)&out, 256 * 1024 * 4 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t n = 64;
    for (int wpc = 4; wpc <= 32; wpc *= 2)          // waves per CU (one workgroup per CU, 256 workgroups)
        for (uint32_t lanes = 64; lanes >= 16; lanes /= 2) {
            const int threads = wpc * 64 > 1024 ? 1024 : wpc * 64, blocks = 256 * (wpc * 64 / threads);
            hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(threads), 0, 0, out, 1u, lanes);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(threads), 0, 0, out, n, lanes);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double ops = (double)blocks * (threads / 64) * n * REP;       // wave-instructions
            printf("VALU waves/CU %2d active lanes %2u: %.3f ms, %.1f G wave-instr/s, %.2f cycles per wave-instr per SIMD at 2.4 GHz\n",
                   wpc, lanes, ms, ops / ms / 1e6, 2.4e9 * 1024.0 / (ops / (ms * 1e-3)));
        }
    for (int wpc = 4; wpc <= 32; wpc *= 2) {
        const int threads = wpc * 64 > 1024 ? 1024 : wpc * 64, blocks = 256 * (wpc * 64 / threads);
        hipLaunchKernelGGL(k_salu, dim3(blocks), dim3(threads), 0, 0, out, 1u);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_salu, dim3(blocks), dim3(threads), 0, 0, out, n);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)blocks * (threads / 64) * n * REP;
        printf("SALU waves/CU %2d: %.3f ms, %.1f G wave-instr/s, %.2f cycles per instr per CU at 2.4 GHz\n", wpc, ms, ops / ms / 1e6,
               2.4e9 * 256.0 / (ops / (ms * 1e-3)));
    }
    return 0;
}
