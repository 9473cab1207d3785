/*
 * tests/emu/hip/hip_runtime.h — a tiny single-process stand-in for <hip/hip_runtime.h>.
 *
 * TEST INFRASTRUCTURE ONLY.  It lets g++ compile the UNMODIFIED engine source
 * (microservice_matchmaking_amd/csrc/mm_engine.hip) into tests/emu/libmm_engine_emu.so so
 * the kernels' control logic (tile loops, ballots, compaction, lobby handling) can be
 * exercised by `pytest -m "not gpu"` in a container without a GPU.  It is not a backend:
 * the product package only ever loads csrc/libmm_engine.so and fails loudly without it.
 *
 * Model: one OS thread; every HIP thread of a block is a ucontext fiber; blocks of a grid
 * run one after another.  A wave is 64 consecutive threads.  __syncthreads() and the wave
 * collectives (__ballot, __shfl) are rendezvous points; a rendezvous that can never
 * complete (divergent collective, missing barrier) aborts with a message instead of
 * hanging — which is exactly the class of bug this shim exists to catch.
 * It does NOT model memory ordering, caches or timing.
 */
#ifndef MM_EMU_HIP_RUNTIME_H
#define MM_EMU_HIP_RUNTIME_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define MM_EMU 1

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct emu_dim3 {
    unsigned x, y, z;
    emu_dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef emu_dim3 dim3;
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

extern emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipErrorOutOfMemory 2
#define hipErrorNotReady 600
#define hipErrorUnknown 999
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
#define hipHostMallocDefault 0
#define hipStreamNonBlocking 1

hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError(void);
const char* hipGetErrorString(hipError_t e);

/* ---- device-side primitives ---- */
void __syncthreads(void);
unsigned long long __ballot(int pred);
int emu_shfl_i32(int v, int src_lane);
static inline int __shfl(int v, int src) { return emu_shfl_i32(v, src); }
static inline unsigned __shfl(unsigned v, int src) { return (unsigned)emu_shfl_i32((int)v, src); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, sizeof(u)); return u; }

/* wave_sync() of the engine: a fence is a no-op here, the wave barrier is a rendezvous */
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)__ballot(1))
/* priorities do not exist here; a sleep must be a yield point or a polling wave would spin forever */
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define __builtin_amdgcn_readfirstlane(v) (v)
static inline long long clock64(void) { return 0; }
static inline long long wall_clock64(void) { return 0; }
/* relaxed workgroup-scope atomics: plain accesses through a volatile lvalue */
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __HIP_MEMORY_SCOPE_SYSTEM 4
static inline void __threadfence_system(void) {}
#define __hip_atomic_load(p, order, scope) (*(const volatile __typeof__(*(p))*)(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(volatile __typeof__(*(p))*)(p) = (v)))
#define __builtin_amdgcn_readlane(v, l) emu_shfl_i32((v), (l))
#define __builtin_amdgcn_s_sleep(n) ((void)__ballot(1))

template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <class T> static inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <class T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

void emu_launch(emu_dim3 grid, emu_dim3 block, const std::function<void()>& body);

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu_launch(emu_dim3(grid), emu_dim3(block), [&]() { kernel(__VA_ARGS__); })

#endif
