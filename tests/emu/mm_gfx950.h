// tests/emu/mm_gfx950.h — TEST INFRASTRUCTURE ONLY: the plain-C++ stand-ins of
// microservice_matchmaking_amd/csrc/mm_gfx950.h for the fiber-shim build (no ISA on a CPU).
#ifndef MM_GFX950_H
#define MM_GFX950_H
#define TW_SINK(v) asm volatile("" ::"r"(v))
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
#endif
