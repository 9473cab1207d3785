// hop_latency.hip — what one hop of kp_round's walk can cost at best: a dependent chase through a 1.2 MB table
// (REC of a 300 000-position chain) with one wave of a 1024-thread workgroup,
//   scalar (s_load_dword through K$) or vector (global_load_dword) loads,
//   table cold (first touch after the kernel started) or pulled through this XCD's L2 by the other 15 waves first,
//   the other waves idle or busy on the chaser's SIMD (LDS + VALU work, as the staging / sweep waves of kp_round are).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/hop_latency.hip -o /tmp/hop && /tmp/hop
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define N_TAB (300000u)
#define HOPS 2048u

static __device__ __forceinline__ uint32_t sload(const uint32_t* p)
{
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
static __device__ __forceinline__ const uint32_t* sptr(const uint32_t* p)
{
    const unsigned long long v = (unsigned long long)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (const uint32_t*)(((unsigned long long)hi << 32) | lo);
}

// mode bit 0: scalar chase; bit 1: warm the table through L2 first; bit 2: the other waves keep busy during the chase
__global__ __launch_bounds__(1024) void k_hop(const uint32_t* __restrict__ tab, uint32_t mode, uint32_t* out)
{
    __shared__ uint32_t s_w[16384];
    __shared__ uint32_t s_done;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) s_done = 0;
    for (int i = tid; i < 16384; i += 1024) s_w[i] = i;
    __syncthreads();
    if ((mode & 2u) && wave != 0) {
        uint32_t acc = 0;
        const uint4* t4 = (const uint4*)tab;
        for (uint32_t i = tid - 64; i < N_TAB / 4; i += 960) { const uint4 v = t4[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        asm volatile("" ::"v"(acc));
    }
    __syncthreads();
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        uint32_t x = 1;
        const long long t0 = clock64();
        if (mode & 1u) {
            for (uint32_t h = 0; h < HOPS; ++h) x = sload(sptr(tab + x));
        } else {
            for (uint32_t h = 0; h < HOPS; ++h) { x = tab[x]; x = (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
        }
        const long long t1 = clock64();
        if (lane == 0) { out[0] = (uint32_t)(t1 - t0); out[1] = x; }
        __hip_atomic_store(&s_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (mode & 4u) {
        uint32_t a = tid, acc = 0;
        while (!__hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { a = s_w[(a * 2654435761u >> 18) & 16383u] + k; acc += a * 3u; }
        }
        asm volatile("" ::"v"(acc));
    }
}

int main()
{
    std::vector<uint32_t> perm(N_TAB), tab(N_TAB);
    for (uint32_t i = 0; i < N_TAB; ++i) perm[i] = i;
    uint64_t s = 88172645463325252ull;
    for (uint32_t i = N_TAB - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const uint32_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
    for (uint32_t i = 0; i < N_TAB; ++i) tab[perm[i]] = perm[(i + 1) % N_TAB];      // one cycle through all entries
    uint32_t *d_tab, *d_out, *d_junk;
    hipMalloc(&d_tab, N_TAB * 4);
    hipMalloc(&d_out, 8);
    hipMalloc(&d_junk, 64u << 20);
    hipMemcpy(d_tab, tab.data(), N_TAB * 4, hipMemcpyHostToDevice);
    const char* names[8] = {"vector cold idle", "scalar cold idle", "vector warm idle", "scalar warm idle",
                            "vector cold busy", "scalar cold busy", "vector warm busy", "scalar warm busy"};
    for (int rep = 0; rep < 2; ++rep)
        for (uint32_t mode = 0; mode < 8; ++mode) {
            hipMemset(d_junk, rep + mode, 64u << 20);          // push the table out of every cache between runs
            hipDeviceSynchronize();
            hipLaunchKernelGGL(k_hop, dim3(1), dim3(1024), 0, 0, d_tab, mode, d_out);
            uint32_t h[2];
            hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
            printf("%-18s %6.0f cycles per hop\n", names[mode], (double)h[0] / HOPS);
        }
    return 0;
}
