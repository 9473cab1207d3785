// xwg_hop.hip — what a dependent scalar load costs when ANOTHER workgroup of the same launch wrote the table
// (kp_rounds' walk reads REC / bitmap words the other tiles' workgroups published a moment ago):
//   producer block P writes a 1.2 MB permutation table (plain or write-through `sc1` stores), drains its stores, raises a flag;
//   consumer block C polls the flag, invalidates its scalar cache and chases 2048 dependent hops through the table.
//   C on the producer's XCD (block index = P + 8) or on another one (P + 1); optional: the consumer's other waves pull the
//   table through the L2 with vector loads first.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/xwg_hop.hip -o /tmp/xwg && /tmp/xwg
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
static __device__ __forceinline__ uint32_t xcc()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xFu;
}

// mode bit 0: producer stores write-through (sc1); bit 1: consumer's waves 1-15 pull the table through L2 first (sc1 vector loads);
// bit 2: consumer chases with VECTOR sc1 loads instead of scalar loads; bit 3: the pull uses PLAIN loads; bit 4: the producer reads
// back what it wrote (plain loads) before it raises the flag
__global__ __launch_bounds__(1024) void k_xwg(const uint32_t* __restrict__ src, uint32_t* tab, uint32_t* flag, uint32_t mode,
                                              uint32_t prod, uint32_t cons, uint32_t* out)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (blockIdx.x == prod) {
        if (mode & 1u) for (uint32_t i = tid; i < N_TAB; i += 1024) __hip_atomic_store(tab + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else for (uint32_t i = tid; i < N_TAB; i += 1024) tab[i] = src[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (mode & 16u) {
            uint32_t acc = 0;
            for (uint32_t i = tid * 16u; i < N_TAB; i += 1024 * 16) acc ^= ((volatile uint32_t*)tab)[i];   // a word per 64 bytes
            asm volatile("" ::"v"(acc));
        }
        __syncthreads();
        if (tid == 0) { out[3] = xcc(); __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    } else if (blockIdx.x == cons) {
        if (tid == 0) while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
        __syncthreads();
        asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        if ((mode & 2u) && wave != 0) {
            uint32_t acc = 0;
            if (mode & 8u) for (uint32_t i = (tid - 64) * 16u; i < N_TAB; i += 960 * 16) acc ^= ((volatile uint32_t*)tab)[i];
            else for (uint32_t i = (tid - 64) * 16u; i < N_TAB; i += 960 * 16) acc ^= __hip_atomic_load(tab + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // a word per 64-byte line
            asm volatile("" ::"v"(acc));
        }
        __syncthreads();
        if (wave == 0) {
            uint32_t x = 1;
            const long long t0 = clock64();
            if (mode & 4u) for (uint32_t h = 0; h < HOPS; ++h) { x = __hip_atomic_load(tab + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); x = (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
            else for (uint32_t h = 0; h < HOPS; ++h) x = sload(sptr(tab + x));
            const long long t1 = clock64();
            if (lane == 0) { out[0] = (uint32_t)(t1 - t0); out[1] = x; out[2] = xcc(); }
        }
    }
}

int main()
{
    std::vector<uint32_t> perm(N_TAB), tab(N_TAB);
    for (uint32_t i = 0; i < N_TAB; ++i) perm[i] = i;
    uint64_t s = 88172645463325252ull;
    for (uint32_t i = N_TAB - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const uint32_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
    for (uint32_t i = 0; i < N_TAB; ++i) tab[perm[i]] = perm[(i + 1) % N_TAB];
    // the answer the chase must give (a stale table gives another one)
    uint32_t want = 1;
    for (uint32_t h = 0; h < HOPS; ++h) want = tab[want];
    uint32_t *d_src, *d_tab, *d_flag, *d_out, *d_junk;
    hipMalloc(&d_src, N_TAB * 4); hipMalloc(&d_tab, N_TAB * 4); hipMalloc(&d_flag, 4); hipMalloc(&d_out, 16); hipMalloc(&d_junk, 64u << 20);
    hipMemcpy(d_src, tab.data(), N_TAB * 4, hipMemcpyHostToDevice);
    const uint32_t modes[] = {0, 1, 2, 3, 4, 6, 10, 11, 16, 17, 26};
    for (int place = 0; place < 2; ++place)
        for (uint32_t mode : modes) {
            hipMemset(d_tab, 0, N_TAB * 4);                   // a stale table chases to 0 and stays there
            hipMemset(d_flag, 0, 4);
            hipMemset(d_junk, place + mode, 64u << 20);
            hipDeviceSynchronize();
            const uint32_t prod = 0, cons = place == 0 ? 8 : 1;
            hipLaunchKernelGGL(k_xwg, dim3(16), dim3(1024), 0, 0, d_src, d_tab, d_flag, mode, prod, cons, d_out);
            uint32_t h[4];
            hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
            char nm[160];
            snprintf(nm, sizeof nm, "%s stores%s, %s%s chase", (mode & 1u) ? "sc1" : "plain", (mode & 16u) ? " + read back" : "",
                     (mode & 2u) ? ((mode & 8u) ? "pulled (plain loads), " : "pulled (sc1 loads), ") : "", (mode & 4u) ? "vector sc1" : "scalar");
            printf("%-64s consumer on %s (XCD %u / producer %u): %6.0f cycles per hop%s\n", nm,
                   place == 0 ? "the producer's XCD" : "another XCD      ", h[2], h[3], (double)h[0] / HOPS, h[1] == want ? "" : "   STALE DATA");
        }
    return 0;
}
