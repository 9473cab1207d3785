// mm_gfx950.h — the few primitives of the kernels that are written in gfx950 ISA directly.
// Included as <mm_gfx950.h>: the product build finds this file (-I. in csrc/Makefile); the CPU
// fiber-shim build of the kernel source used by the logic tests puts its own file of the same
// name first on the include path (tests/emu/mm_gfx950.h), so the kernel source itself carries
// no test-only branches.
#ifndef MM_GFX950_H
#define MM_GFX950_H

// keep a value alive in a vector register (a load whose result is only wanted in the cache)
#define TW_SINK(v) asm volatile("" ::"v"(v))

// a wave-uniform address the compiler may hold in vector registers: into scalar registers
static __device__ __forceinline__ const uint32_t* tw_sptr(const uint32_t* p)
{
    const unsigned long long v = (unsigned long long)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (const uint32_t*)(((unsigned long long)hi << 32) | lo);
}

// one dword through the scalar cache (K$ -> L2): the dependent load of a pointer chase whose
// address is wave-uniform AND held in scalar registers (the hot loops; tw_sload_v takes any)
static __device__ __forceinline__ uint32_t tw_sload(const uint32_t* p)
{
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}


// two independent dwords through the scalar cache, one wait: both addresses wave-uniform
static __device__ __forceinline__ void tw_sload2(const uint32_t* p0, const uint32_t* p1, uint32_t& v0, uint32_t& v1)
{
    asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(v0), "=&s"(v1) : "s"(p0), "s"(p1) : "memory");
}

// the same for addresses that are wave-uniform but that the compiler holds in vector registers
static __device__ __forceinline__ uint32_t tw_sload_v(const uint32_t* p) { return tw_sload(tw_sptr(p)); }
static __device__ __forceinline__ void tw_sload2_v(const uint32_t* p0, const uint32_t* p1, uint32_t& v0, uint32_t& v1)
{
    tw_sload2(tw_sptr(p0), tw_sptr(p1), v0, v1);
}

// a 16-byte aligned record of four dwords, address wave-uniform (any register class)
static __device__ __forceinline__ void tw_sload4_v(const uint32_t* p, uint32_t& v0, uint32_t& v1, uint32_t& v2, uint32_t& v3)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    const uint32_t* const sp = tw_sptr(p);
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(sp) : "memory");
    v0 = v.x; v1 = v.y; v2 = v.z; v3 = v.w;
}

// every store this wave has issued has been acknowledged (gfx9: vmcnt counts stores too).  After write-through
// (agent-scope atomic) stores: they are in memory, and a flag stored next is seen after them
static __device__ __forceinline__ void tw_store_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Workgroups that WAIT for others of their launch (kt_chase's emitters for its chasers; kt_fc's chasers for kt_f's
// chunks, its emitters for the chasers) come in an order that lets the hardware's in-order dispatch work for them: whoever
// is waited for is on the chip, or can get there, before the one that waits can fill it (mm_team.inc, kt_fc).  The CPU
// shim of the tests runs workgroups one after another and needs the ones that are waited for FIRST
// (tests/emu/mm_gfx950.h).
#define MM_WAITERS_FIRST 1

// a kernel's register budget as waves per SIMD (kt_fc: six = 80 vector registers = three 512-thread workgroups a CU;
// measured: at two workgroups a CU a cfg-3 tick costs 0.5 ms more, at four nothing less)
#define MM_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))

// which of the eight XCDs this wave runs on (diagnostics: workgroup -> XCD placement is observed, not promised)
static __device__ __forceinline__ uint32_t mm_xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xFu;
}

// ---- what the persistent pair rounds (kp_rounds, mm_pair.inc) need ----
// The scalar cache (read only, shared by neighbouring CUs, in front of the XCD's L2) is not refreshed by anybody's stores:
// after a flag barrier among workgroups, before the first scalar load of what the others published.
static __device__ __forceinline__ void tw_sinv() { asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); }
// A value another workgroup of this launch has stored: past this CU's vector L1 (which nobody's stores refresh), served by the L2.
static __device__ __forceinline__ uint32_t xld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ uint32_t xld16(const uint16_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ unsigned long long xld64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// keep the result of a returning atomic alive (the returning form is what makes the wave wait for it)
#define TW_KEEP(v) asm volatile("" ::"v"(v))
// The value as the compiler cannot see through: what is derived from it is computed again where it is used instead of
// being hoisted out of a long loop and kept in registers across it.  kp_rounds<8192> spilled 36 vector registers of
// loop-invariant LDS addresses (tid + k * 1024, clamped) to scratch and reloaded them, a `s_waitcnt vmcnt(0)` behind each,
// in the middle of every pass (round 5, found in the ISA).
#define TW_LAUNDER(v) asm volatile("" : "+v"(v))
// The walk's fast hops (kp_rounds), tight: from the cursor position p with the REC there (rr), follow "exit anchor whose partner
// is still queued" from tile to tile while the partner lies BELOW position `lim` — per hop ONE round trip through the scalar cache
// (the partner's bitmap word and the REC behind the partner) and two dozen scalar instructions; the compiler's version of the same
// loop, with the own tile's bookkeeping inside, takes 45 and four taken branches.  k counts the lobbies on the way.
// Ends at the first hop that is not of that kind (p, rr describe it): the caller's general code takes it from there.
// REC layout (mm_pair.inc, rec_make): kind << 30 | lobbies inside the tile << 17 | landing position in the tile + 1; kind 1 = partner.
static __device__ __forceinline__ void tw_hops(const uint32_t* bits, const uint32_t* rec, uint32_t m, uint32_t tmask, uint32_t lim,
                                               uint32_t& p, uint32_t& rr, uint32_t& k)
{
    uint32_t t0, t1, tb, nxt, q, bw, r2;
    asm volatile(
        ".Ltw_top_%=:\n\t"
        "s_lshr_b32 %[t0], %[rr], 30\n\t"
        "s_cmp_lg_u32 %[t0], 1\n\t"
        "s_cbranch_scc1 .Ltw_end_%=\n\t"
        "s_and_b32 %[tb], %[p], %[tmask]\n\t"
        "s_and_b32 %[t0], %[rr], 0x1ffff\n\t"
        "s_add_u32 %[nxt], %[tb], %[t0]\n\t"
        "s_cmp_gt_u32 %[nxt], %[lim]\n\t"
        "s_cbranch_scc1 .Ltw_end_%=\n\t"
        "s_cmp_ge_u32 %[nxt], %[m]\n\t"
        "s_cbranch_scc1 .Ltw_end_%=\n\t"
        "s_add_u32 %[q], %[nxt], -1\n\t"
        "s_lshr_b32 %[t0], %[q], 5\n\t"
        "s_lshl_b32 %[t0], %[t0], 2\n\t"
        "s_lshl_b32 %[t1], %[nxt], 2\n\t"
        "s_load_dword %[bw], %[bits], %[t0]\n\t"
        "s_load_dword %[r2], %[rec], %[t1]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_lshr_b32 %[bw], %[bw], %[q]\n\t"
        "s_bitcmp1_b32 %[bw], 0\n\t"
        "s_cbranch_scc0 .Ltw_end_%=\n\t"
        "s_bfe_u32 %[t0], %[rr], 0xd0011\n\t"
        "s_add_u32 %[k], %[k], %[t0]\n\t"
        "s_add_u32 %[k], %[k], 1\n\t"
        "s_mov_b32 %[p], %[nxt]\n\t"
        "s_mov_b32 %[rr], %[r2]\n\t"
        "s_branch .Ltw_top_%=\n\t"
        ".Ltw_end_%=:\n\t"
        : [p] "+s"(p), [rr] "+s"(rr), [k] "+s"(k), [t0] "=&s"(t0), [t1] "=&s"(t1), [tb] "=&s"(tb), [nxt] "=&s"(nxt), [q] "=&s"(q),
          [bw] "=&s"(bw), [r2] "=&s"(r2)
        : [bits] "s"(bits), [rec] "s"(rec), [m] "s"(m), [tmask] "s"(tmask), [lim] "s"(lim)
        : "memory", "scc");
}

// A workgroup's state that lives as long as its launch: LDS.  (The CPU shim of the tests runs the workgroups of a launch one
// after another, so it runs a persistent launch as one launch per pass and keeps this state per workgroup between them.)
#define MM_RESIDENT(Type, var) __shared__ Type var
#define MM_RESIDENT_FRESH(var) ((void)0)      // a new launch: LDS holds whatever the CU's last tenant left (the shim poisons it)
// iterations of kp_rounds per launch: 0 = the whole batch in ONE launch (what the kernel is for)
#define MM_PERSIST_SLICE 0u

#endif
