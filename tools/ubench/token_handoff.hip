// token_handoff.hip — what passing the walk's cursor from one device to the next would cost per pass (DESIGN.md section 7:
// a POSITION split of one chain over several GPUs makes the cursor — position, lobbies so far, the carried anchor — a token
// that goes from rank to rank once per pass and comes back).  Emulated on ONE GPU as the review asked: two persistent
// workgroups on two streams hand a 64-byte token back and forth
//   (a) through host-pinned memory (what a peer store over the fabric looks like to the receiver: a write that lands outside
//       its caches, a poll that has to leave the chip),
//   (b) through device memory (two workgroups of one device: the floor).
// One hand-off = the sender's payload store + flag store (system / agent scope) -> the receiver's poll sees the flag -> it
// reads the payload.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/token_handoff.hip -o /tmp/tok && /tmp/tok
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <chrono>

struct Token { unsigned long long flag; unsigned long long payload[7]; };

template <int SCOPE>
__global__ void k_side(Token* tok, uint32_t me, uint32_t rounds, unsigned long long* out)
{
    if (threadIdx.x != 0) return;
    unsigned long long acc = 0;
    const long long t0 = wall_clock64();
    for (uint32_t r = 0; r < rounds; ++r) {
        const unsigned long long want = 2ull * r + me;              // side 0 moves on even values, side 1 on odd ones
        while (__hip_atomic_load(&tok->flag, __ATOMIC_ACQUIRE, SCOPE) != want) __builtin_amdgcn_s_sleep(1);
        for (int k = 0; k < 7; ++k) acc += __hip_atomic_load(&tok->payload[k], __ATOMIC_RELAXED, SCOPE);
        for (int k = 0; k < 7; ++k) __hip_atomic_store(&tok->payload[k], acc + k, __ATOMIC_RELAXED, SCOPE);
        __hip_atomic_store(&tok->flag, want + 1ull, __ATOMIC_RELEASE, SCOPE);
    }
    out[me] = (unsigned long long)(wall_clock64() - t0);
    out[2 + me] = acc;
}

int main()
{
    const uint32_t rounds = 20000;
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    unsigned long long* d_out;
    hipMalloc(&d_out, 32);
    for (int where = 0; where < 2; ++where) {
        Token* tok;
        if (where == 0) hipHostMalloc(&tok, sizeof(Token), hipHostMallocCoherent | hipHostMallocMapped);
        else hipMalloc(&tok, sizeof(Token));
        hipMemset(tok, 0, sizeof(Token));
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        if (where == 0) {
            hipLaunchKernelGGL(k_side<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(1), dim3(64), 0, s0, tok, 0u, rounds, d_out);
            hipLaunchKernelGGL(k_side<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(1), dim3(64), 0, s1, tok, 1u, rounds, d_out);
        } else {
            hipLaunchKernelGGL(k_side<__HIP_MEMORY_SCOPE_AGENT>, dim3(1), dim3(64), 0, s0, tok, 0u, rounds, d_out);
            hipLaunchKernelGGL(k_side<__HIP_MEMORY_SCOPE_AGENT>, dim3(1), dim3(64), 0, s1, tok, 1u, rounds, d_out);
        }
        hipStreamSynchronize(s0);
        hipStreamSynchronize(s1);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("%-34s %7.2f us per hand-off (%u round trips of two hand-offs, 64-byte token, wall %0.1f ms)\n",
               where == 0 ? "token in host-pinned memory:" : "token in device memory:", us / (2.0 * rounds), rounds, us / 1000.0);
        if (where == 0) hipHostFree(tok); else hipFree(tok);
    }
    return 0;
}
