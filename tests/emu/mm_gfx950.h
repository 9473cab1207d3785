// tests/emu/mm_gfx950.h — TEST INFRASTRUCTURE ONLY: the plain-C++ stand-ins of
// microservice_matchmaking_amd/csrc/mm_gfx950.h for the fiber-shim build (no ISA on a CPU).
#ifndef MM_GFX950_H
#define MM_GFX950_H
#define TW_SINK(v) asm volatile("" ::"r"(v))
static inline const uint32_t* tw_sptr(const uint32_t* p) { return p; }
static inline uint32_t tw_sload(const uint32_t* p) { return *p; }
static inline void tw_sload2(const uint32_t* p0, const uint32_t* p1, uint32_t& v0, uint32_t& v1) { v0 = *p0; v1 = *p1; }
static inline uint32_t tw_sload_v(const uint32_t* p) { return *p; }
static inline void tw_sload2_v(const uint32_t* p0, const uint32_t* p1, uint32_t& v0, uint32_t& v1) { v0 = *p0; v1 = *p1; }
static inline void tw_sload4_v(const uint32_t* p, uint32_t& v0, uint32_t& v1, uint32_t& v2, uint32_t& v3) { v0 = p[0]; v1 = p[1]; v2 = p[2]; v3 = p[3]; }
static inline uint32_t mm_xcc_id() { return 0; }
static inline void tw_store_wait() {}
// workgroups run one after another here: the ones that wait for others of their launch come last
#define MM_WAITERS_FIRST 0
#define MM_WAVES_PER_SIMD(n)
// kp_rounds (mm_pair.inc): no caches to get past here; the workgroup-resident state of a persistent launch is kept per
// workgroup across the launches the shim cuts it into (workgroups run one after another: one pass per launch)
static inline void tw_sinv() {}
static inline uint32_t xld(const uint32_t* p) { return *(const volatile uint32_t*)p; }
static inline uint32_t xld16(const uint16_t* p) { return *(const volatile uint16_t*)p; }
static inline unsigned long long xld64(const unsigned long long* p) { return *(const volatile unsigned long long*)p; }
// the walk's fast hops (see the product header): the same loop in plain C++
static inline void tw_hops(const uint32_t* bits, const uint32_t* rec, uint32_t m, uint32_t tmask, uint32_t lim,
                           uint32_t& p, uint32_t& rr, uint32_t& k)
{
    while ((rr >> 30) == 1u) {
        const uint32_t tb = p & tmask;
        const uint32_t nxt = tb + (rr & 0x1FFFFu), q = nxt - 1u;
        if (nxt > lim || nxt >= m) break;
        const uint32_t bw = bits[q >> 5], r2 = rec[nxt];
        if (!((bw >> (q & 31u)) & 1u)) break;
        k += ((rr >> 17) & 0x1FFFu) + 1u;
        p = nxt;
        rr = r2;
    }
}
#define TW_KEEP(v) ((void)(v))
#define TW_LAUNDER(v) ((void)(v))
#include <map>
#define MM_RESIDENT(Type, var)                                                              \
    static std::map<unsigned long long, Type*> var##_all;                                   \
    Type*& var##_p = var##_all[((unsigned long long)blockIdx.y << 32) | blockIdx.x];        \
    if (!var##_p) var##_p = new Type();                                                     \
    Type& var = *var##_p
// the first thread of a workgroup runs first here: poison before anybody has written (a real launch starts on garbage)
#define MM_RESIDENT_FRESH(var) do { if (threadIdx.x == 0) memset(&(var), 0xA5, sizeof(var)); } while (0)
#define MM_PERSIST_SLICE 1u
#endif
