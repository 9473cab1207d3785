// trip_latency.hip — what one "trip to memory" of the team path's look-ups costs: a wave issues a handful of VECTOR
// loads whose addresses depend on the previous trip's data (a record here, 64 sub-queue entries there, ten players'
// ratings somewhere else), waits for all of them, and goes on.  Measured: cycles per trip for
//   1 / 4 / 8 loads per trip, each a 64-lane gather (lanes 64 B apart) or a coalesced 256-byte row,
//   all in ONE 64 MB buffer or each in a buffer of its OWN (16 x 8 MB: the engine's arrays are separate allocations),
//   on an idle device or beside a kernel that streams 1 GB through HBM (kt_f beside the chaser).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/trip_latency.hip -o /tmp/trip && /tmp/trip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define TRIPS 1024u
#define NBUF 16u
#define WORDS (2u << 20)          // 8 MB per buffer

struct Bufs { const uint32_t* b[NBUF]; };

// NL loads per trip, all in flight together (unrolled: the wait comes after the last one is issued);
// mode bit 0: gather (lanes 16 words apart) instead of a coalesced row; bit 1: every load in a buffer of its own
template <uint32_t NL>
__global__ __launch_bounds__(64) void k_trip(Bufs B, uint32_t mode, uint32_t* out)
{
    const uint32_t lane = threadIdx.x, gather = mode & 1u, spread = (mode >> 1) & 1u;
    uint32_t x = 12345u;
    const long long t0 = clock64();
    for (uint32_t t = 0; t < TRIPS; ++t) {
        uint32_t v[NL];
#pragma unroll
        for (uint32_t k = 0; k < NL; ++k) {
            const uint32_t* p = B.b[spread ? (k + t) % NBUF : 0u];
            const uint32_t base = (x * 2654435761u + k * 40503u) % (WORDS - 2048u);
            const uint32_t off = gather ? lane * 16u : lane;
            v[k] = p[(base & ~15u) + off];
        }
        uint32_t acc = 0;
#pragma unroll
        for (uint32_t k = 0; k < NL; ++k) acc ^= v[k];
        // the next trip's addresses depend on this one's data
        x = (uint32_t)__builtin_amdgcn_readfirstlane((int)acc) + t;
    }
    const long long t1 = clock64();
    if (lane == 0) { out[0] = (uint32_t)((t1 - t0) / TRIPS); out[1] = x; }
}

__global__ void k_stream(const uint4* src, uint4* dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main()
{
    Bufs B;
    std::vector<uint32_t> h(WORDS);
    for (uint32_t i = 0; i < WORDS; ++i) h[i] = i * 2246822519u;
    uint32_t* big;
    hipMalloc(&big, (size_t)NBUF * WORDS * 4);            // one allocation ...
    uint32_t* own[NBUF];
    for (uint32_t k = 0; k < NBUF; ++k) { hipMalloc(&own[k], WORDS * 4); hipMemcpy(own[k], h.data(), WORDS * 4, hipMemcpyHostToDevice); }
    for (uint32_t k = 0; k < NBUF; ++k) hipMemcpy(big + (size_t)k * WORDS, h.data(), WORDS * 4, hipMemcpyHostToDevice);
    uint32_t* d_out;
    hipMalloc(&d_out, 8);
    uint4 *s0, *s1;
    const size_t sn = (256u << 20) / 16;
    hipMalloc(&s0, sn * 16); hipMalloc(&s1, sn * 16);
    hipStream_t bg, fg;                                    // (non-blocking: the two really run side by side)
    hipStreamCreateWithFlags(&bg, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&fg, hipStreamNonBlocking);
    for (int busy = 0; busy < 2; ++busy)
        for (int alloc = 0; alloc < 2; ++alloc)           // 0: slices of the one allocation, 1: allocations of their own
            for (uint32_t spread = 0; spread < 2; ++spread)
                for (uint32_t gather = 0; gather < 2; ++gather)
                    for (uint32_t nl : {1u, 4u, 8u}) {
                        if (!spread && alloc) continue;
                        for (uint32_t k = 0; k < NBUF; ++k) B.b[k] = alloc ? own[k] : big + (size_t)k * WORDS;
                        if (busy) for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, bg, s0, s1, sn);
                        const uint32_t md = gather | (spread << 1);
                        if (nl == 1u) hipLaunchKernelGGL(k_trip<1>, dim3(1), dim3(64), 0, fg, B, md, d_out);
                        else if (nl == 4u) hipLaunchKernelGGL(k_trip<4>, dim3(1), dim3(64), 0, fg, B, md, d_out);
                        else hipLaunchKernelGGL(k_trip<8>, dim3(1), dim3(64), 0, fg, B, md, d_out);
                        uint32_t r[2];
                        hipStreamSynchronize(fg);
                        hipMemcpy(r, d_out, 8, hipMemcpyDeviceToHost);
                        hipDeviceSynchronize();
                        printf("%-5s %-22s %-7s %u loads a trip: %6u cycles\n", busy ? "busy" : "idle",
                               !spread ? "one buffer" : alloc ? "16 allocations" : "16 slices of one",
                               gather ? "gather" : "row", nl, r[0]);
                    }
    return 0;
}
