// mm_engine.hip — MI355X (gfx950) implementation of include/mm_engine.h.
//
// One translation unit: HIP kernels + the C-ABI host code that drives them.
// Path replaced: Matchmaking.Search.Worker.consume/5 (reference lib/search/worker.ex:291-324)
// run under the canonical schedule of SURVEY.md §3.4 / docs/MATCH_CHECK.md ("Mode R"),
// plus the bucketing hop in front of it (lib/generic/worker.ex:46-69).
//
// Device data layout (all HBM-resident, SoA, queue-ordered):
//   chain = (game mode, rating group) — an independent FIFO + one open lobby
//           (lib/application.ex:26-40, lib/models/lobby_state.ex:15-29).
//   q_rating[chain][cap] i32, q_cons[chain][cap] u32, q_slot[chain][cap] u32
//           the broker queue `matchmaking.queues.<group>` restricted to one mode, head at
//           index 0; always in ascending arrival order (MATCH_CHECK.md §4).
//   chains[chain]   ChainDev: queue length, the open lobby (LobbyState record), per-tick
//           counters.
//   state[slot] u8  ActiveUser mirror: 1 = in queue, 2 = cancelled (lib/models/active_user.ex).
//   out_*[group][..] per-group emission logs of the last tick (team-ordered slots, score, pass).
//
// Kernels:
//   k_bucket_count / k_bucket_scan / k_bucket_scatter   stable multi-way partition of an
//           arrival batch into chain tails (A1 bucketing); HBM streaming, wave-ballot ranks.
//   k_cancel, k_purge   ActiveUser.remove_user + the "vanish when popped" rule.
//   k_walk   one workgroup per chain: LDS-staged tiles of the queue, wave-0 first-fit chain
//           (ballot + ffs arg-min over queue position), in-place survivor compaction.
//           The chain is inherently sequential (each match decides the next anchor), so
//           the parallelism is: 64 candidates per anchor step, tile load/compaction on the
//           other waves, and independent chains on independent CUs.  Walks every mode; the
//           fallback of the pair path.
//   mm_pair.inc   the 1v1 walk: next[] pointers for every queued player, one launch per pass
//           over all tiles of all chains (kp_round), an LDS-resident speculative pointer chase
//           for short chains (kp_late).  DESIGN.md §4.3.
//   mm_team.inc   the team-mode walk: per pass, role sub-queues and F[a] = the player the
//           cursor picks after a's lobby, for every queued a at once, then a pointer chase
//           (kt_build / kt_f / kt_f2 / kt_chase / kt_fc / kt_late).  DESIGN.md §4.4.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <dlfcn.h>

#include <algorithm>
#include <mutex>
#include <cmath>
#include <new>
#include <vector>

#include "../../include/mm_engine.h"
#include <mm_gfx950.h>

// ------------------------------------------------------------------------------------
// device-side structures
// ------------------------------------------------------------------------------------

#define MM_NO_SLOT 0xFFFFFFFFu
#define MM_ST_FREE 0
#define MM_ST_LIVE 1
#define MM_ST_CANCELLED 2

#define MM_ERRF_QUEUE_OVERFLOW 1u
#define MM_ERRF_OUT_OVERFLOW 2u
#define MM_ERRF_SEAT_INVARIANT 4u
#define MM_ERRF_PASS_LIMIT 8u
#define MM_ERRF_SYNC_TIMEOUT 16u      // an emitter workgroup of kt_chase gave up waiting for its chaser

struct LobbyDev {                                 // the record LobbyState stores
    uint32_t n;
    uint32_t cnt[MM_MAX_TEAMS];
    uint32_t slot[MM_MAX_TEAMS][8];
    int32_t  rating[MM_MAX_TEAMS][8];
    uint32_t cons[MM_MAX_TEAMS][8];
};

struct ChainDev {
    uint32_t len;        // queue length
    uint32_t head_state; // set by k_purge: 0 = queue was empty, 1 = head alive, 2 = head cancelled
    uint32_t n_out;      // lobbies emitted in the last tick
    uint32_t passes;
    uint32_t err;
    uint32_t purged;     // entries k_purge removed (folded into `before` by k_walk)
    uint32_t before;     // queue + lobby population when the tick began
    uint32_t pad;
    unsigned long long pairs;
    unsigned long long scanned;
    LobbyDev lobby;
};

struct ModeDev {
    uint32_t team_size, teams, window, eqmask, n_roles, L;
    uint32_t quota[MM_MAX_ROLES];
};

struct BucketCfg {
    uint32_t n_groups, n_modes, default_group, capacity;
    int32_t from[MM_MAX_GROUPS], to[MM_MAX_GROUPS];
    uint32_t n_roles[MM_MAX_MODES];
    uint32_t quota_mask[MM_MAX_MODES];  // bit r set = role r seatable
};

#define BK_THREADS 256
#define BK_WAVES 4
#define BK_PER_WAVE 512                       // elements per wave per block
#define BK_CHUNK (BK_WAVES * BK_PER_WAVE)     // elements per block
#define BK_MAXC (MM_MAX_GROUPS * MM_MAX_MODES)
#define BK_INVALID 0xFFFFu

#define WK_THREADS 256
#define WK_TILE 4096

static __device__ __forceinline__ uint32_t dev_min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// generic/worker.ex:46-53 on an int32 rating
static __device__ __forceinline__ uint32_t dev_rating_group(const BucketCfg& B, int32_t r)
{
    uint32_t g = B.default_group;
    for (int k = (int)B.n_groups - 1; k >= 0; --k)
        if (r >= B.from[k] && r <= B.to[k]) g = (uint32_t)k;   // lowest matching index wins
    return g;
}

static __device__ __forceinline__ uint32_t dev_chain_of(const BucketCfg& B, int32_t r, uint32_t cons,
                                                         const uint8_t* group, uint32_t i)
{
    const uint32_t mode = cons & 0xFu, role = (cons >> 16) & 0xFu;
    if (mode >= B.n_modes || role >= B.n_roles[mode] || !((B.quota_mask[mode] >> role) & 1u))
        return BK_INVALID;
    const uint32_t g = group ? (uint32_t)group[i] : dev_rating_group(B, r);
    return mode * B.n_groups + g;
}

// Orders LDS traffic between the lanes of ONE wave: everything written before it (by any
// lane) is visible to every lane after it.  The hardware executes a wave's LDS ops in
// order; the fence stops the compiler from forwarding stale register copies across it.
static __device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Up to 256 per-chain counters of one wave, held in registers: lane l owns chains
// l, l+64, l+128, l+192.  Reads are wave shuffles, updates touch one lane.
struct LaneCounters {
    uint32_t v0, v1, v2, v3;
};
static __device__ __forceinline__ uint32_t lc_get(const LaneCounters& k, uint32_t c)
{
    const uint32_t x = c < 64u ? k.v0 : c < 128u ? k.v1 : c < 192u ? k.v2 : k.v3;   // c is wave-uniform
    return (uint32_t)__shfl((int)x, (int)(c & 63u));
}
static __device__ __forceinline__ void lc_add(LaneCounters& k, uint32_t c, uint32_t n, int lane)
{
    if ((uint32_t)lane == (c & 63u)) {
        if (c < 64u) k.v0 += n;
        else if (c < 128u) k.v1 += n;
        else if (c < 192u) k.v2 += n;
        else k.v3 += n;
    }
}

// Inclusive prefix sum over the 64 lanes of a wave.
static __device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = (uint32_t)__shfl((int)v, lane >= d ? lane - d : lane);
        if (lane >= d) v += t;
    }
    return v;
}

// ------------------------------------------------------------------------------------
// bucketing (enqueue) kernels
// ------------------------------------------------------------------------------------

// Per-wave histogram of chain ids: wave_hist[(block*4 + wave) * n_chains + chain].
__global__ __launch_bounds__(BK_THREADS) void k_bucket_count(uint32_t n, const int32_t* __restrict__ rating,
                                                             const uint32_t* __restrict__ cons,
                                                             const uint8_t* __restrict__ group, BucketCfg B,
                                                             uint32_t n_chains, uint32_t* __restrict__ wave_hist,
                                                             uint32_t* __restrict__ rejected)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    LaneCounters hist = { 0u, 0u, 0u, 0u };
    uint32_t rej = 0;
    const uint32_t base = blockIdx.x * BK_CHUNK + wave * BK_PER_WAVE;
    for (uint32_t r = 0; r < BK_PER_WAVE / 64; ++r) {
        const uint32_t i = base + r * 64 + lane;
        const bool valid = i < n;
        uint32_t ch = BK_INVALID;
        if (valid) ch = dev_chain_of(B, rating[i], cons[i] & MM_CONS_USER_MASK, group, i);
        rej += (valid && ch == BK_INVALID) ? 1u : 0u;
        unsigned long long todo = __ballot(ch != BK_INVALID);
        while (todo) {
            const int l = __ffsll(todo) - 1;
            const uint32_t c0 = (uint32_t)__shfl((int)ch, l);
            const unsigned long long m = __ballot(ch == c0);
            lc_add(hist, c0, (uint32_t)__popcll(m), lane);
            todo &= ~m;
        }
    }
    if (__ballot(rej != 0)) {   // rare: reduce the per-lane reject counts
        const uint32_t s = wave_incl_scan(rej, lane);
        if (lane == 63) atomicAdd(rejected, s);
    }
    const uint32_t row = (blockIdx.x * BK_WAVES + wave) * n_chains;
    if ((uint32_t)lane < n_chains) wave_hist[row + lane] = hist.v0;
    if ((uint32_t)lane + 64u < n_chains) wave_hist[row + lane + 64u] = hist.v1;
    if ((uint32_t)lane + 128u < n_chains) wave_hist[row + lane + 128u] = hist.v2;
    if ((uint32_t)lane + 192u < n_chains) wave_hist[row + lane + 192u] = hist.v3;
}

// Exclusive scan of wave_hist down the wave axis, per chain, seeded with the chain's
// current length; updates the chain lengths.  One block.
__global__ __launch_bounds__(BK_THREADS) void k_bucket_scan(uint32_t n_rows, uint32_t n_chains,
                                                            uint32_t* __restrict__ wave_hist,
                                                            ChainDev* __restrict__ chains, uint32_t capacity)
{
    __shared__ uint32_t wtot[BK_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per = (n_rows + BK_THREADS - 1) / BK_THREADS;
    const uint32_t r0 = dev_min_u32(tid * per, n_rows), r1 = dev_min_u32(r0 + per, n_rows);
    // one workgroup per chain (round 5: one workgroup went through the chains one after the other, 33 us for seven)
    {
        const uint32_t c = blockIdx.x;
        uint32_t s = 0;
        for (uint32_t r = r0; r < r1; ++r) s += wave_hist[r * n_chains + c];
        const uint32_t incl = wave_incl_scan(s, lane);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t off = chains[c].len;
        for (int w = 0; w < wave; ++w) off += wtot[w];
        uint32_t run = off + incl - s;
        for (uint32_t r = r0; r < r1; ++r) {
            const uint32_t h = wave_hist[r * n_chains + c];
            wave_hist[r * n_chains + c] = run;
            run += h;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t total = chains[c].len;
            for (int w = 0; w < BK_WAVES; ++w) total += wtot[w];
            if (total > capacity) { chains[c].err |= MM_ERRF_QUEUE_OVERFLOW; total = capacity; }
            chains[c].len = total;
        }
        __syncthreads();
    }
}

// Stable scatter into the chain tails.  wave_base = scanned wave_hist.
__global__ __launch_bounds__(BK_THREADS) void k_bucket_scatter(uint32_t n, const int32_t* __restrict__ rating,
                                                               const uint32_t* __restrict__ cons,
                                                               const uint8_t* __restrict__ group, BucketCfg B,
                                                               uint32_t n_chains,
                                                               const uint32_t* __restrict__ wave_base,
                                                               uint32_t first_slot,
                                                               const uint32_t* __restrict__ slot_sel,
                                                               int32_t* __restrict__ q_rating,
                                                               uint32_t* __restrict__ q_cons,
                                                               uint32_t* __restrict__ q_slot,
                                                               uint8_t* __restrict__ state,
                                                               uint32_t* __restrict__ out_slot)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t row = (blockIdx.x * BK_WAVES + wave) * n_chains;
    LaneCounters cur = { 0u, 0u, 0u, 0u };
    if ((uint32_t)lane < n_chains) cur.v0 = wave_base[row + lane];
    if ((uint32_t)lane + 64u < n_chains) cur.v1 = wave_base[row + lane + 64u];
    if ((uint32_t)lane + 128u < n_chains) cur.v2 = wave_base[row + lane + 128u];
    if ((uint32_t)lane + 192u < n_chains) cur.v3 = wave_base[row + lane + 192u];
    const uint32_t base = blockIdx.x * BK_CHUNK + wave * BK_PER_WAVE;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t r = 0; r < BK_PER_WAVE / 64; ++r) {
        const uint32_t i = base + r * 64 + lane;
        const bool valid = i < n;
        uint32_t ch = BK_INVALID, cn = 0;
        int32_t rt = 0;
        if (valid) {
            rt = rating[i];
            cn = cons[i] & MM_CONS_USER_MASK;
            ch = dev_chain_of(B, rt, cn, group, i);
        }
        // slot_sel: the host picked the free slots itself (the ring range held a waiting player)
        const uint32_t slot = (slot_sel && valid) ? slot_sel[i]
                                                  : (uint32_t)(((unsigned long long)first_slot + i) % B.capacity);
        if (valid && out_slot) out_slot[i] = (ch == BK_INVALID) ? MM_NO_SLOT : slot;
        unsigned long long todo = __ballot(ch != BK_INVALID);
        while (todo) {
            const int l = __ffsll(todo) - 1;
            const uint32_t c0 = (uint32_t)__shfl((int)ch, l);
            const unsigned long long m = __ballot(ch == c0);
            const uint32_t start = lc_get(cur, c0);
            if (ch == c0) {
                const uint32_t pos = start + (uint32_t)__popcll(m & lt);
                if (pos < B.capacity) {
                    const size_t o = (size_t)c0 * B.capacity + pos;
                    q_rating[o] = rt;
                    q_cons[o] = cn;
                    q_slot[o] = slot;
                    state[slot] = MM_ST_LIVE;
                }
            }
            lc_add(cur, c0, (uint32_t)__popcll(m), lane);
            todo &= ~m;
        }
    }
}

// ------------------------------------------------------------------------------------
// liveness (ActiveUser) kernels
// ------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_cancel(uint32_t n, const uint32_t* __restrict__ slots,
                                                uint8_t* __restrict__ state)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) state[slots[i]] = MM_ST_CANCELLED;
}

// One block per chain of the ticked mode: records whether the head of the queue was
// cancelled (MATCH_CHECK.md §4: decides whether the first live attempt sees the stale
// lobby) and removes cancelled entries in place, appending their slots to `released`.
__global__ __launch_bounds__(WK_THREADS) void k_purge(uint32_t mode, uint32_t n_groups, uint32_t capacity,
                                                      ChainDev* __restrict__ chains,
                                                      int32_t* __restrict__ q_rating, uint32_t* __restrict__ q_cons,
                                                      uint32_t* __restrict__ q_slot, uint8_t* __restrict__ state,
                                                      uint32_t* __restrict__ released,
                                                      uint32_t* __restrict__ n_released)
{
    __shared__ uint32_t wtot[WK_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t c = mode * n_groups + blockIdx.x;
    const size_t qo = (size_t)c * capacity;
    const uint32_t m = chains[c].len;
    if (m == 0) {
        if (tid == 0) { chains[c].head_state = 0; chains[c].purged = 0; }
        return;
    }
    const uint32_t head_state = state[q_slot[qo]] != MM_ST_LIVE ? 2u : 1u;
    uint32_t wr = 0;
    for (uint32_t rd = 0; rd < m; rd += WK_THREADS) {
        const uint32_t i = rd + tid;
        int32_t rt = 0;
        uint32_t cn = 0, sl = 0;
        bool live = false;
        if (i < m) {
            rt = q_rating[qo + i];
            cn = q_cons[qo + i];
            sl = q_slot[qo + i];
            live = state[sl] == MM_ST_LIVE;
            if (!live) {
                state[sl] = MM_ST_FREE;
                released[atomicAdd(n_released, 1u)] = sl;
            }
        }
        const unsigned long long bm = __ballot(live);
        if (lane == 0) wtot[wave] = (uint32_t)__popcll(bm);
        __syncthreads();   // also orders the reads above before the in-place writes below
        uint32_t off = wr, tot = 0;
        for (int w = 0; w < WK_THREADS / 64; ++w) {
            if (w < wave) off += wtot[w];
            tot += wtot[w];
        }
        if (live) {
            const uint32_t d = off + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull));
            q_rating[qo + d] = rt;
            q_cons[qo + d] = cn;
            q_slot[qo + d] = sl;
        }
        wr += tot;
        __syncthreads();
    }
    if (tid == 0) {
        chains[c].len = wr;
        chains[c].head_state = head_state;
        chains[c].purged = m - wr;
    }
}

__global__ __launch_bounds__(256) void k_reset(uint32_t n_chains, ChainDev* __restrict__ chains)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t words = (uint32_t)(sizeof(ChainDev) / 4);
    if (i < n_chains * words) ((uint32_t*)chains)[i] = 0;
}

// ------------------------------------------------------------------------------------
// the walk kernel: Search.Worker.consume/5 to quiescence, one workgroup per chain
// ------------------------------------------------------------------------------------

struct PairChain;
static __device__ __forceinline__ bool pair_chain_is_fast(const struct PairChain* pc, uint32_t g);
struct TeamChain;
static __device__ __forceinline__ bool team_chain_is_fast(const struct TeamChain* tc, uint32_t g);

struct WalkParams {
    uint32_t mode, n_groups, capacity, purge;
    uint32_t out_cap;          // lobbies per group the out_* arrays can hold
    uint32_t out_slot_stride;  // u32 per group in out_slots
    uint32_t out_rec_stride;   // records per group in out_score / out_pass
    uint32_t max_passes;
    ModeDev M;
    ChainDev* chains;
    int32_t* q_rating;
    uint32_t* q_cons;
    uint32_t* q_slot;
    uint8_t* state;
    uint32_t* released;
    uint32_t* n_released;
    uint32_t* out_slots;       // [n_groups][out_slot_stride], team-ordered slots, L per lobby
    float* out_score;          // [n_groups][out_rec_stride]
    uint32_t* out_pass;        // [n_groups][out_rec_stride]
    const struct PairChain* pskip;   // chains the pair path (mm_pair.inc) walks this tick, or NULL
    const struct TeamChain* tskip;   // chains the team path (mm_team.inc) walked this tick, or NULL
};

// Discipline for the LDS lobby inside wave 0: every lane may READ it between two
// wave_sync() points; only lane 0 WRITES it, bracketed by wave_sync() on both sides.

// docs/MATCH_CHECK.md §2.1-2.3: the team `(r, cn)` would be seated in, or -1.  Read-only.
static __device__ int lobby_pick_team(const LobbyDev& lb, const ModeDev& M, int32_t r, uint32_t cn)
{
    int at = -1;
    for (uint32_t t = 0; t < M.teams; ++t)
        if (lb.cnt[t]) { at = (int)t; break; }
    if (at >= 0) {
        const int32_t ar = lb.rating[at][0];
        const uint32_t ac = lb.cons[at][0];
        const uint32_t d = r >= ar ? (uint32_t)r - (uint32_t)ar : (uint32_t)ar - (uint32_t)r;
        if (d > M.window) return -1;
        if ((cn ^ ac) & M.eqmask) return -1;
    }
    const uint32_t role = (cn >> 16) & 0xFu;
    int best = -1;
    long long best_sum = 0;
    for (uint32_t t = 0; t < M.teams; ++t) {
        uint32_t have = 0;
        long long sum = 0;
        for (uint32_t k = 0; k < lb.cnt[t]; ++k) {
            have += ((lb.cons[t][k] >> 16) & 0xFu) == role ? 1u : 0u;
            sum += lb.rating[t][k];
        }
        if (have >= M.quota[role]) continue;
        if (best < 0 || sum < best_sum) { best = (int)t; best_sum = sum; }
    }
    return best;
}

// docs/MATCH_CHECK.md §2.4: append to the chosen team.
static __device__ int lobby_try_seat(LobbyDev& lb, const ModeDev& M, int32_t r, uint32_t cn, uint32_t slot, int lane)
{
    const int best = lobby_pick_team(lb, M, r, cn);
    wave_sync();
    if (best >= 0 && lane == 0) {
        const uint32_t k = lb.cnt[best];
        lb.slot[best][k] = slot;
        lb.rating[best][k] = r;
        lb.cons[best][k] = cn;
        lb.cnt[best] = k + 1;
        lb.n = lb.n + 1;
    }
    wave_sync();
    return best;
}

// remove_inactive_players/1 (search/worker.ex:267-280) on the LDS lobby; the released
// slots go to the host so it can recycle them.
static __device__ void lobby_filter(LobbyDev& lb, const ModeDev& M, uint8_t* state, uint32_t* released,
                                    uint32_t* n_released, int lane)
{
    wave_sync();
    if (lane == 0) {
        for (uint32_t t = 0; t < M.teams; ++t) {
            uint32_t w = 0;
            const uint32_t n = lb.cnt[t];
            for (uint32_t k = 0; k < n; ++k) {
                const uint32_t sl = lb.slot[t][k];
                if (state[sl] == MM_ST_LIVE) {
                    const int32_t r = lb.rating[t][k];
                    const uint32_t cn = lb.cons[t][k];
                    lb.slot[t][w] = sl;
                    lb.rating[t][w] = r;
                    lb.cons[t][w] = cn;
                    ++w;
                } else {
                    state[sl] = MM_ST_FREE;
                    released[atomicAdd(n_released, 1u)] = sl;
                }
            }
            lb.n = lb.n - (n - w);
            lb.cnt[t] = w;
        }
    }
    wave_sync();
}

static __device__ bool lobby_has_cancelled(const LobbyDev& lb, const ModeDev& M, const uint8_t* state)
{
    bool any = false;
    for (uint32_t t = 0; t < M.teams; ++t)
        for (uint32_t k = 0; k < lb.cnt[t]; ++k)
            if (state[lb.slot[t][k]] != MM_ST_LIVE) any = true;
    return any;
}

// The open lobby as wave 0 sees it while it walks: a register copy (uniform across the lanes)
// of everything match_check needs — anchor, seats per team, players per (team, role), team
// rating sums, free seats per role — so that a seat costs a handful of ALU operations and one
// record written to the LDS lobby by lane 0, not a re-count of the lobby.
struct LobbySum {
    uint32_t n, cnt[MM_MAX_TEAMS], rc[MM_MAX_TEAMS];     // rc: 4 bits per role
    long long sum[MM_MAX_TEAMS];
    int32_t ar;
    uint32_t ac, at;                                      // anchor: rating, cons, its team (MM_MAX_TEAMS: none)
    uint32_t free, full;                                  // 4 bits per role, seats over all teams
};

static __device__ __forceinline__ void lsum_load(LobbySum& L, const LobbyDev& lb, const ModeDev& M)
{
    L.n = lb.n;
    L.full = 0;
    for (uint32_t rr = 0; rr < M.n_roles; ++rr) L.full |= (M.teams * M.quota[rr]) << (4u * rr);   // <= 16 only for
    uint32_t used = 0;                                                                            // a one-role mode
    int at = -1;
#pragma unroll
    for (uint32_t t = 0; t < MM_MAX_TEAMS; ++t) {
        L.cnt[t] = t < M.teams ? lb.cnt[t] : 0u;
        L.rc[t] = 0;
        L.sum[t] = 0;
        for (uint32_t k = 0; k < L.cnt[t]; ++k) {
            L.rc[t] += 1u << (4u * ((lb.cons[t][k] >> 16) & 7u));
            L.sum[t] += lb.rating[t][k];
        }
        used += L.rc[t];
        if (at < 0 && L.cnt[t]) at = (int)t;
    }
    L.free = L.full - used;                               // per nibble: no borrow, used <= full
    L.ar = at >= 0 ? lb.rating[at][0] : 0;
    L.ac = at >= 0 ? lb.cons[at][0] : 0u;
    L.at = at >= 0 ? (uint32_t)at : (uint32_t)MM_MAX_TEAMS;
    __builtin_amdgcn_wave_barrier();                      // every lane has read the lobby before lane 0 writes it again
}

// docs/MATCH_CHECK.md §2.3-2.4 for a player already known to pass §2.2 (or an empty lobby):
// the eligible team with the smallest rating sum, lowest index on a tie; -1 if no team has a
// free seat of the role.
static __device__ __forceinline__ int lsum_seat(LobbySum& L, LobbyDev& lb, const ModeDev& M, int32_t r, uint32_t cn, uint32_t slot, int lane)
{
    const uint32_t role = (cn >> 16) & 7u;
    int best = -1;
    long long best_sum = 0;
#pragma unroll
    for (uint32_t t = 0; t < MM_MAX_TEAMS; ++t) {
        if (t >= M.teams) continue;
        if (((L.rc[t] >> (4u * role)) & 15u) >= M.quota[role]) continue;
        if (best < 0 || L.sum[t] < best_sum) { best = (int)t; best_sum = L.sum[t]; }
    }
    if (best < 0) return -1;
    uint32_t k = 0;
#pragma unroll
    for (uint32_t t = 0; t < MM_MAX_TEAMS; ++t)
        if ((int)t == best) { k = L.cnt[t]; L.cnt[t] = k + 1u; L.rc[t] += 1u << (4u * role); L.sum[t] += r; }
    // the anchor is the first player of the lowest-numbered non-empty team (MATCH_CHECK.md §2.1):
    // a seat in a team below the anchor's (empty until now — a cancel emptied it, or the lobby
    // was empty) makes this player the anchor
    if ((uint32_t)best < L.at) { L.ar = r; L.ac = cn; L.at = (uint32_t)best; }
    L.n += 1u;
    L.free -= 1u << (4u * role);
    if (lane == 0) {
        lb.slot[best][k] = slot;
        lb.rating[best][k] = r;
        lb.cons[best][k] = cn;
        lb.cnt[best] = k + 1u;
        lb.n = L.n;
    }
    return best;
}

__global__ __launch_bounds__(WK_THREADS) void k_walk(WalkParams P)
{
    __shared__ int32_t t_rating[WK_TILE];
    __shared__ uint32_t t_cons[WK_TILE];
    __shared__ uint32_t t_slot[WK_TILE];
    __shared__ uint32_t t_mask[WK_TILE / 32];       // bit = element left the queue (seated)
    __shared__ uint32_t t_pref[WK_TILE / 32];       // exclusive survivor prefix per mask word
    __shared__ LobbyDev lb;
    __shared__ uint32_t s_changed, s_total, s_w0tot;
    __shared__ uint32_t s_cmd[6];                   // wave 0 -> block: {kind, from, anchor rating, anchor cons, free seats, found}

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t g = blockIdx.x;
    if (P.pskip && pair_chain_is_fast(P.pskip, g)) return;   // walked by the pair path
    if (P.tskip && team_chain_is_fast(P.tskip, g)) return;   // walked by the team path
    const uint32_t c = P.mode * P.n_groups + g;
    const ModeDev& M = P.M;
    const size_t qo = (size_t)c * P.capacity;
    int32_t* const qr = P.q_rating + qo;
    uint32_t* const qc = P.q_cons + qo;
    uint32_t* const qs = P.q_slot + qo;
    ChainDev* const ch = P.chains + c;

    // ---- load the stored lobby (LobbyState.get_state) ----
    {
        const uint32_t words = (uint32_t)(sizeof(LobbyDev) / 4);
        const uint32_t* src = (const uint32_t*)&ch->lobby;
        uint32_t* dst = (uint32_t*)&lb;
        for (uint32_t i = tid; i < words; i += WK_THREADS) dst[i] = src[i];
    }
    uint32_t m = ch->len;
    const uint32_t head_state = P.purge ? ch->head_state : 0u;
    const uint32_t before = m + ch->lobby.n + (P.purge ? ch->purged : 0u);
    __syncthreads();

    // wave-0 chain state (uniform across the wave)
    LobbySum L;                                 // register copy of the LDS lobby (see lsum_load)
    bool sum_valid = false;
    bool first_stale = false;
    uint32_t n_out = 0, err = 0;
    unsigned long long pairs = 0, scanned = 0;
    uint32_t passes = 0;

    if (wave == 0 && head_state != 0u) {
        // deferred effect of mm_cancel on the stored lobby (MATCH_CHECK.md §4): an empty
        // queue means no attempt, so the lobby stays as stored; a cancelled head filters it
        // before anyone is judged; a live head is judged against the stale lobby first.
        if (head_state == 1u && lobby_has_cancelled(lb, M, P.state)) first_stale = true;
        else lobby_filter(lb, M, P.state, P.released, P.n_released, lane);
    }

    while (m > 0) {
        bool changed = false;
        uint32_t wr = 0;
        scanned += m;
        for (uint32_t rd = 0; rd < m; rd += WK_TILE) {
            const uint32_t cnt = dev_min_u32(WK_TILE, m - rd);
            // ---- stage the tile: coalesced SoA loads HBM -> LDS ----
            for (uint32_t i = tid; i < cnt; i += WK_THREADS) {
                t_rating[i] = qr[rd + i];
                t_cons[i] = qc[rd + i];
                t_slot[i] = qs[rd + i];
            }
            if (tid < WK_TILE / 32) t_mask[tid] = 0;
            __syncthreads();

            // ---- wave 0: the first-fit chain over the tile.  When 64 candidates in a row were
            //      all rejected it asks the whole block for the next candidate the lobby would take
            //      (a starving lobby rejects thousands in a row: one block-wide step per tile then). ----
            uint32_t p = 0;
            for (;;) {
            if (wave == 0) {
                if (first_stale) {
                    // first attempt after a cancel: judged against the stale lobby, then filter
                    first_stale = false;
                    sum_valid = false;
                    if (lb.n) pairs += 1;
                    const int t = lobby_try_seat(lb, M, t_rating[0], t_cons[0], t_slot[0], lane);
                    lobby_filter(lb, M, P.state, P.released, P.n_released, lane);
                    if (t >= 0) {
                        changed = true;
                        if (lane == 0) atomicOr(&t_mask[0], 1u);
                    }
                    p = 1;
                }
                while (p < cnt) {
                    if (!sum_valid) { wave_sync(); lsum_load(L, lb, M); sum_valid = true; }
                    if (L.n == 0) {
                        // empty lobby: the popped player opens it (MATCH_CHECK.md §2.1)
                        const int t = lsum_seat(L, lb, M, t_rating[p], t_cons[p], t_slot[p], lane);
                        if (t < 0) err |= MM_ERRF_SEAT_INVARIANT;
                        if (lane == 0) atomicOr(&t_mask[p >> 5], 1u << (p & 31));
                        changed = true;
                        ++p;
                        continue;
                    }
                    const int32_t ar = L.ar;
                    const uint32_t ac = L.ac;
                    const uint32_t s_free = L.free;
                    // 64 candidates, one per lane
                    const uint32_t i = p + lane;
                    const bool valid = i < cnt;
                    const uint32_t nvalid = dev_min_u32(64u, cnt - p);
                    int32_t r = 0;
                    uint32_t cn = 0;
                    if (valid) { r = t_rating[i]; cn = t_cons[i]; }
                    const uint32_t d = r >= ar ? (uint32_t)r - (uint32_t)ar : (uint32_t)ar - (uint32_t)r;
                    const uint32_t role = (cn >> 16) & 0xFu;
                    const bool ok = valid && d <= M.window && (((cn ^ ac) & M.eqmask) == 0) &&
                                    ((s_free >> (4u * (role & 7u))) & 15u) != 0u;
                    unsigned long long S = 0;
                    if (__ballot(ok)) {
                        for (uint32_t rr = 0; rr < M.n_roles; ++rr) {
                            uint32_t fr = (s_free >> (4u * rr)) & 15u;
                            unsigned long long mk = __ballot(ok && role == rr);
                            while (fr && mk) {                 // first `fr` candidates of the role
                                S |= mk & (~mk + 1ull);
                                mk &= mk - 1ull;
                                --fr;
                            }
                        }
                    }
                    if (S == 0) {
                        pairs += nvalid;
                        p += 64;
                        if (p < cnt) break;                     // let the block look ahead
                        continue;
                    }
                    changed = true;
                    // seat the selected candidates in queue order; stop at the one that fills the
                    // lobby, or that becomes the new anchor (the rest of the chunk was judged
                    // against the old one and is looked at again)
                    unsigned long long todo = S, seated = 0;
                    uint32_t stop = 64u;
                    bool fills = false;
                    while (todo) {
                        const uint32_t b = (uint32_t)__ffsll(todo) - 1u;
                        todo &= todo - 1ull;
                        const uint32_t e = p + b;
                        const uint32_t at0 = L.at;
                        const int t = lsum_seat(L, lb, M, t_rating[e], t_cons[e], t_slot[e], lane);
                        if (t < 0) err |= MM_ERRF_SEAT_INVARIANT;
                        seated |= 1ull << b;
                        if (L.n == M.L) { fills = true; stop = b; break; }
                        if (L.at != at0) { stop = b; break; }
                    }
                    if ((seated >> lane) & 1ull) atomicOr(&t_mask[(p + lane) >> 5], 1u << ((p + lane) & 31));
                    const uint32_t last = stop;
                    if (!fills) {
                        if (stop < 64u) { pairs += stop + 1u; p += stop + 1u; }     // new anchor: judge the rest again
                        else { pairs += nvalid; p += 64; }
                        continue;
                    }
                    // ---- is_filled: emit in team order (search/worker.ex:313-319) ----
                    if (L.n != M.L) err |= MM_ERRF_SEAT_INVARIANT;
                    wave_sync();
                    if (lane == 0) {
                        if (n_out < P.out_cap) {
                            long long smin = 0, smax = 0;
                            uint32_t k = 0;
                            uint32_t* os = P.out_slots + (size_t)g * P.out_slot_stride + (size_t)n_out * M.L;
                            for (uint32_t t = 0; t < M.teams; ++t) {
                                long long sm = 0;
                                for (uint32_t j = 0; j < lb.cnt[t]; ++j) {
                                    os[k++] = lb.slot[t][j];
                                    sm += lb.rating[t][j];
                                }
                                if (t == 0 || sm < smin) smin = sm;
                                if (t == 0 || sm > smax) smax = sm;
                            }
                            P.out_score[(size_t)g * P.out_rec_stride + n_out] =
                                (float)(int32_t)(smax - smin) / (float)(int32_t)M.team_size;
                            P.out_pass[(size_t)g * P.out_rec_stride + n_out] = passes;
                        }
                        lb.n = 0;
                        for (uint32_t t = 0; t < M.teams; ++t) lb.cnt[t] = 0;
                    }
                    wave_sync();
                    L.n = 0;                                    // the register copy: an empty lobby
                    L.at = MM_MAX_TEAMS;
                    L.free = L.full;
#pragma unroll
                    for (uint32_t t = 0; t < MM_MAX_TEAMS; ++t) { L.cnt[t] = 0; L.rc[t] = 0; L.sum[t] = 0; }
                    if (n_out >= P.out_cap) err |= MM_ERRF_OUT_OVERFLOW;
                    ++n_out;
                    pairs += last + 1u;
                    p += last + 1u;
                }
                if (lane == 0) {
                    s_cmd[0] = p < cnt ? 1u : 0u;               // 1 = look ahead from p, 0 = tile done
                    s_cmd[1] = p;
                    s_cmd[2] = (uint32_t)L.ar;
                    s_cmd[3] = L.ac;
                    s_cmd[4] = L.free;
                    s_cmd[5] = 0xFFFFFFFFu;
                }
            }
            __syncthreads();
            if (s_cmd[0] == 0u) break;
            {
                // first candidate at or after s_cmd[1] that fits the anchor and whose role has a free seat
                const int32_t c_ar = (int32_t)s_cmd[2];
                const uint32_t c_ac = s_cmd[3], c_free = s_cmd[4];
                for (uint32_t i = s_cmd[1] + (uint32_t)tid; i < cnt; i += WK_THREADS) {
                    const int32_t r = t_rating[i];
                    const uint32_t cn = t_cons[i];
                    const uint32_t d = r >= c_ar ? (uint32_t)r - (uint32_t)c_ar : (uint32_t)c_ar - (uint32_t)r;
                    if (d <= M.window && (((cn ^ c_ac) & M.eqmask) == 0) && ((c_free >> (4u * ((cn >> 16) & 7u))) & 15u)) {
                        atomicMin(&s_cmd[5], i);
                        break;
                    }
                }
            }
            __syncthreads();
            if (wave == 0) {
                const uint32_t f = s_cmd[5] < cnt ? s_cmd[5] : cnt;
                pairs += f - p;                                  // everybody in between was rejected
                p = f;
            }
            }
            __syncthreads();

            // ---- survivors back to the queue, in place, order preserved (requeue) ----
            const uint32_t nwords = (cnt + 31u) >> 5;
            if (tid < WK_TILE / 32) {
                uint32_t sv = 0;
                if ((uint32_t)tid < nwords) {
                    uint32_t vm = 0xFFFFFFFFu;
                    if ((uint32_t)tid == nwords - 1u && (cnt & 31u)) vm = (1u << (cnt & 31u)) - 1u;
                    sv = (uint32_t)__popc(~t_mask[tid] & vm);
                }
                const uint32_t incl = wave_incl_scan(sv, lane);
                t_pref[tid] = incl - sv;
                if (tid == 63) s_w0tot = incl;
                if (tid == WK_TILE / 32 - 1) s_total = incl;   // wave 1's total, fixed up below
            }
            __syncthreads();
            const uint32_t w0tot = s_w0tot;
            const uint32_t total = w0tot + s_total;
            if (wr != rd || total != cnt)          // nobody left the queue so far in this pass: all in place
            for (uint32_t i = tid; i < cnt; i += WK_THREADS) {
                const uint32_t w = i >> 5, b = i & 31u;
                const uint32_t mw = t_mask[w];
                if (!((mw >> b) & 1u)) {
                    const uint32_t dpos = wr + t_pref[w] + (w >= 64u ? w0tot : 0u) +
                                          (uint32_t)__popc(~mw & ((1u << b) - 1u));
                    qr[dpos] = t_rating[i];
                    qc[dpos] = t_cons[i];
                    qs[dpos] = t_slot[i];
                }
            }
            wr += total;
            __syncthreads();
        }
        // ---- end of pass: quiescence test (MATCH_CHECK.md §4) ----
        if (tid == 0) s_changed = changed ? 1u : 0u;
        __syncthreads();
        const uint32_t chg = s_changed;
        m = wr;
        ++passes;
        if (!chg) break;
        if (passes >= P.max_passes) { err |= MM_ERRF_PASS_LIMIT; break; }
        __syncthreads();
    }

    // ---- store the lobby back (LobbyState.update_state) and the tick counters ----
    __syncthreads();
    {
        const uint32_t words = (uint32_t)(sizeof(LobbyDev) / 4);
        uint32_t* dst = (uint32_t*)&ch->lobby;
        const uint32_t* src = (const uint32_t*)&lb;
        for (uint32_t i = tid; i < words; i += WK_THREADS) dst[i] = src[i];
    }
    if (tid == 0) {
        ch->len = m;
        ch->head_state = 0;
        ch->purged = 0;
        ch->before = before;
        ch->n_out = n_out;
        ch->passes = passes;
        ch->pairs = pairs;
        ch->scanned = scanned;
        ch->err |= err;
    }
}

// The match list of a small tick (every tick of a stream) in ONE piece: the groups' emission logs gathered, emission
// order, into a staging buffer laid out [slots: total x L | score: total | pass: total] — three copies to the host
// instead of three per rating group (21 blit kernels and as many runtime calls for a tick that seats forty lobbies).
struct PackArgs {
    uint32_t pre[MM_MAX_GROUPS + 1];
    uint32_t n_groups, L, total, out_slot_stride, out_rec_stride;
};
__global__ __launch_bounds__(256) void k_pack_results(PackArgs A, const uint32_t* __restrict__ out_slots,
                                                      const float* __restrict__ out_score, const uint32_t* __restrict__ out_pass,
                                                      uint32_t* __restrict__ pk)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.total) return;
    uint32_t g = 0;
    while (g + 1u < A.n_groups && i >= A.pre[g + 1u]) ++g;
    const uint32_t j = i - A.pre[g];
    const uint32_t* const src = out_slots + (size_t)g * A.out_slot_stride + (size_t)j * A.L;
    uint32_t* const dst = pk + (size_t)i * A.L;
    for (uint32_t k = 0; k < A.L; ++k) dst[k] = src[k];
    pk[(size_t)A.total * A.L + i] = __float_as_uint(out_score[(size_t)g * A.out_rec_stride + j]);
    pk[(size_t)A.total * (A.L + 1u) + i] = out_pass[(size_t)g * A.out_rec_stride + j];
}
#define MM_PACK_MAX 8192u            // lobbies of a tick up to which its match list is packed on the device

// The TAIL of a big tick's match list — what the last kernels of the walk emitted, after the last early send — straight
// into the engine's pinned host buffers, each lobby at its final place (the buffers are device-accessible: hipHostMalloc).
// It used to leave as three copies per rating group on the copy stream: 21 copy commands of a few KB, 10-12 us apiece
// one after the other, 0.25 ms behind every tick of cfg-2 (profiles/r05_timeline_gaps_1m_1v1.txt of the round's first call).  One launch,
// coalesced stores over the link, on the engine's own stream: the caller's synchronisation covers it.
struct TailArgs {
    uint32_t pre[MM_MAX_GROUPS + 1];                 // prefix of the groups' fresh lobbies
    uint32_t from[MM_MAX_GROUPS], base[MM_MAX_GROUPS];
    uint32_t n_groups, L, total, out_slot_stride, out_rec_stride;
};
__global__ __launch_bounds__(256) void k_results_tail(TailArgs A, const uint32_t* __restrict__ out_slots,
                                                      const float* __restrict__ out_score, const uint32_t* __restrict__ out_pass,
                                                      uint32_t* __restrict__ h_slots, float* __restrict__ h_score, uint32_t* __restrict__ h_pass)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.total) return;
    uint32_t g = 0;
    while (g + 1u < A.n_groups && i >= A.pre[g + 1u]) ++g;
    const uint32_t j = A.from[g] + (i - A.pre[g]);
    const size_t at = (size_t)A.base[g] + j;
    const uint32_t* const src = out_slots + (size_t)g * A.out_slot_stride + (size_t)j * A.L;
    for (uint32_t k = 0; k < A.L; ++k) h_slots[at * A.L + k] = src[k];
    h_score[at] = out_score[(size_t)g * A.out_rec_stride + j];
    h_pass[at] = out_pass[(size_t)g * A.out_rec_stride + j];
}
#define MM_TAIL_MAX (1u << 20)       // lobbies up to which a tick's tail leaves through k_results_tail (beyond: copy commands)

// A look of the host at the chains' records in the middle of a tick (pair path: after every batch of passes; team path:
// every 16 passes) is a D2H copy of 3 KB and a stream synchronisation.  The alternative built in round 5 (VERDICT r04,
// "What's weak" 4) and kept behind MM_LOOK_POLL=1: the records leave by themselves — the last launch of a batch is this
// kernel, which stores them into the engine's PINNED host buffer (fine-grained: the stores go out over the link as they
// are performed) and then, with a release at system scope, the look's sequence number; the host polls that word in its
// own memory and goes on the moment it arrives — no copy command, no signal, no wake-up.  What it is worth: see
// mm_engine_create (look_poll).
__global__ __launch_bounds__(256) void k_look(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst_host, uint32_t nwords,
                                              uint32_t* seq_host, uint32_t seq)
{
    for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x) dst_host[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(seq_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

#include "mm_pair.inc"
#include "mm_team.inc"

static __device__ __forceinline__ bool pair_chain_is_fast(const struct PairChain* pc, uint32_t g)
{
    return pc[g].fast != 0u;
}
static __device__ __forceinline__ bool team_chain_is_fast(const struct TeamChain* tc, uint32_t g)
{
    return tc[g].fast != 0u;
}

// ------------------------------------------------------------------------------------
// host side: the C ABI
// ------------------------------------------------------------------------------------

struct mm_engine {
    mm_config cfg;
    uint32_t n_chains;
    hipStream_t stream;
    hipEvent_t ev[4];
    hipEvent_t ev_nx[2];                // around kp_nx_init (MM_CFG_TIMING): mm_path_stats.pair_nx_init_ns
    bool ev_nx_set;
    hipEvent_t ev_grp[MM_MAX_GROUPS];   // a group's match list has reached the host
    int last_hip;
    // device
    int32_t* d_q_rating;
    uint32_t* d_q_cons;
    uint32_t* d_q_slot;
    ChainDev* d_chains;
    uint8_t* d_state;
    uint32_t* d_released;
    uint32_t* d_counters;      // [0] n_released, [1] rejected
    uint32_t* d_wave_hist;
    size_t wave_hist_rows;
    int32_t* d_in_rating;      // staging for host-pointer enqueue
    uint32_t* d_in_cons;
    uint8_t* d_in_group;
    uint32_t* d_in_slot;       // also cancel staging
    uint32_t* d_in_sel;        // slots the host picked (mm_enqueue stepping over waiting players)
    size_t in_cap;
    uint32_t* d_out_slots;
    float* d_out_score;
    uint32_t* d_out_pass;
    uint32_t out_slot_stride, out_rec_stride;
    // pair path (mm_pair.inc): per rating group, the ticked mode's chains
    PairChain* d_pchains;
    uint32_t* d_pk_key[2];
    uint32_t* d_pk_oidx[2];
    uint16_t* d_pk_nx16[2];
    uint32_t* d_pk_bits[2];
    uint32_t* d_pk_scratch;
    uint16_t* d_pk_wpre;
    uint16_t* d_pk_g16;
    uint32_t* d_pk_rec1;       // second REC buffer of the fused tiled path (the first is d_pk_scratch)
    uint16_t* d_pk_exa[2];
    uint32_t* d_pk_bitsp[2];
    uint32_t* d_pk_headp[2];
    bool pair_persist;         // MM_PAIR_PERSIST=0: never several passes per launch (kp_rounds); one launch per pass as before (A/B)
    uint32_t pair_ptiles;      // MM_PAIR_PTILES: tiles of the longest chain a kp_rounds batch may have (one workgroup per CU of ONE XCD: 32)
    uint32_t pair_pcool;       // batches for which kp_rounds stays off after a launch that gave up (a time-out: somebody else holds the CUs)
    uint32_t pair_pstops;      // launches that gave up so far (diagnostics)
    uint32_t pair_pinject;     // MM_PAIR_PINJECT: PairParams.pinject (tests)
    uint32_t pair_pbatch;      // MM_PAIR_PBATCH: passes per kp_rounds launch at most (it ends earlier when the longest chain wants its compaction)
    uint32_t pair_ptimeout[2]; // MM_PAIR_PTIMEOUT_US: what a workgroup waits at the first / at a later barrier, in 100 MHz ticks
    unsigned long long* d_pk_pbar;   // [group] arrival words, then [group] XCD masks (one allocation)
    uint32_t round_ctr;
    uint32_t* d_pk_tilectl;
    uint32_t* d_pack;          // the packed match list of a small tick (k_pack_results)
    uint4* d_pk_grec;          // second level of the route (kp_group)
    uint32_t pk_gstride;
    uint32_t pair_nxseg;       // MM_PAIR_NXSEG: anchors per workgroup of kp_nx_init (0: 256)
    uint32_t pair_nxstage;     // MM_PAIR_NXSTAGE: entries it stages in LDS, anchors included (0: NXI_STAGE)
    bool pair_xcd;             // MM_PAIR_XCD=0: kp_round on the plain (tile, group) grid (A/B)
    uint32_t pair_group_min;   // MM_PAIR_GROUP: tiles of the longest chain from which a batch runs with the second level (0 = never)
    PairChain* h_pchains;      // pinned
    uint32_t* h_look_seq;      // pinned: the sequence number of the last look whose records have arrived (k_look)
    uint32_t look_seq;
    bool results_tail_kernel;  // MM_RESULTS_TAIL=0: the tail of a tick's match list as copy commands, as before (A/B)
    bool look_poll;            // MM_LOOK_POLL=0: looks as a D2H copy + stream synchronisation, as before (A/B)
    uint32_t ps_hand[4][MM_MAX_GROUPS];   // per rating group at the pair path's last look: passes, lobbies, kp_rounds passes, kp_rounds hops
    mm_path_stats ps;          // mm_path_stats_get: the launch shapes and fall-backs of the last tick (totals carried over)
    mm_tuning tn;              // the record the engine was created with (mm_tuning_get); the working copies are the fields around here
    uint32_t team_fwait, team_fix_max, team_fix_t8, team_fix_t4, team_pull_xcd, team_nowait;   // MM_TEAM_* knobs of kt_f / kt_fc, read once at create
    uint32_t pk_bits_stride, pk_max_tiles, pk_stride;
    uint32_t pair_batch;       // MM_PAIR_BATCH: tiled rounds launched per host look at the chains
    bool force_generic;        // MM_FORCE_GENERIC=1: always walk with k_walk (A/B testing)
    bool pair_debug;           // MM_PAIR_DEBUG=1: print the pair path's diagnostics per tick
    uint32_t pair_tune;        // MM_PAIR_TUNE: PairParams.tune
    bool pair_tile_fixed;      // MM_PAIR_TILE=max: every batch with the largest tile length (A/B)
    uint32_t pair_tiles_max;   // MM_PAIR_TILES: tiles of the longest chain a batch may have before the next larger tile length is taken
    unsigned long long live_upper;   // upper bound of queued players (grid sizing)
    // team path (mm_team.inc): shares the pair path's arrays, plus
    TeamChain* d_tchains;
    TeamChain* h_tchains;      // pinned
    uint32_t* d_tk_chunk;      // [group][role][tk_chunk_stride]
    uint32_t* d_tk_fv2;        // [group][pk_stride] F o F of the team walk's first passes
    uint32_t* d_tk_fpos;       // [group][pk_stride] kt_f's record of the lobby a position opens: its last member
    uint16_t* d_tk_memb;       // [group][pk_stride][tk_memb] ... and all of them (team modes only)
    uint32_t* d_tk_bitsB;      // [group][pk_bits_stride] the queue bitmap when kt_build last ran
    uint32_t* d_tk_chunkB;     // [group][role][tk_chunk_stride] chunk populations when kt_build last ran
    uint32_t* d_tk_fdone;      // [group][tk_chunk_stride] kt_fc: the launch (team_seq) whose kt_f chunk has stored its F
    uint32_t team_seq;         // kt_chase / kt_fc launches of this engine so far (never 0: it names a launch to its workgroups)
    uint32_t team_emit_max;    // MM_TEAM_EMIT_MAX
    uint32_t team_split;       // MM_TEAM_SPLIT: the stored lobby's fill in kt_f's launch (the lobby-rich passes)
    uint32_t* d_tk_sqi;        // [group][pk_stride] position -> sub-queue entry
    uint32_t team_rebuild;     // MM_TEAM_REBUILD: kt_build runs in the first two passes of a tick and every this many after
    uint32_t tk_memb;          // largest lobby of a team mode, less the anchor
    uint32_t team_f2;          // MM_TEAM_F2: passes of a tick that compose F with itself (0 = never)
    uint32_t tk_chunk_stride;
    uint32_t team_batch;       // MM_TEAM_BATCH: passes launched per host look at the chains
    uint32_t team_cap;         // MM_TEAM_CAP: TeamParams.scan_cap
    uint32_t team_late;        // MM_TEAM_LATE: lobbies per pass at or under which kt_late takes the chains over (0 = never)
    uint32_t team_late0;       // MM_TEAM_LATE0: arrivals of a mode since its last tick at or under which kt_late walks the tick from its first pass
    std::vector<uint32_t> tk_last_len;   // [chain] queue + stored lobby after the chain's last tick (what a quiescent tick left)
    uint32_t dbg_last_w, dbg_last_c;   // MM_PAIR_DEBUG + MM_TEAM_BATCH=1: per-pass deltas of the F counters
    // host
    ChainDev* h_chains;        // pinned, n_chains
    uint32_t* h_counters;      // pinned, 2
    std::vector<uint8_t> h_state;
    uint32_t next_slot;
    uint32_t cancel_pending;
    std::vector<uint32_t> r_released;
    uint32_t* h_rslots;        // pinned: the last tick's lobbies (slots, score, pass), emission order
    float* h_rscore;
    uint32_t* h_rpass;
    uint32_t r_n, r_L;
    // The match list travels while the walk is still running: at every look the host takes at the chains, the
    // lobbies emitted since the last look go out on a stream of their own (emission lists only grow).  Group g's
    // lobbies live at h_r*[r_base[g] ..) — r_base from the most a group can emit (its players at the start of the
    // tick), so a lobby's place is known before the tick is over; r_cnt / r_pre index them for mm_matches.
    hipStream_t copy_stream;
    hipEvent_t ev_copy;
    bool ev_copy_pending;
    uint32_t r_base[MM_MAX_GROUPS], r_cnt[MM_MAX_GROUPS], r_pre[MM_MAX_GROUPS + 1];
    uint32_t r_sent[MM_MAX_GROUPS];     // lobbies of the running tick already on their way / arrived
    uint32_t r_marked[MM_MAX_GROUPS];   // ... whose players' slots are FREE in h_state already
    bool r_based;
    bool results_early;        // MM_RESULTS_EARLY=0: the whole match list after the walk (A/B)
    bool poisoned;             // a tick failed half way: everything but reset / restore / destroy answers MM_ERR_STATE
    uint32_t fault_tick;       // MM_DEBUG_FAIL_TICK=k: the k-th mm_tick of this engine fails after its walk (test hook)
    uint32_t ticks_seen;
};

// Named ranges for a profiler's timeline (the reference logs nothing per attempt).  MM_ROCTX=1 makes
// every engine of the process mark its enqueue / tick phases with roctxRangePush / Pop, resolved at run time from
// librocprofiler-sdk-roctx.so or libroctx64.so — no link-time dependency, nothing at all when the variable is unset.
struct RoctxApi {
    int (*push)(const char*);
    int (*pop)(void);
    bool tried;
};
static RoctxApi g_roctx = { nullptr, nullptr, false };
static void roctx_init_once(void);
static void roctx_init(void)
{
    // engines may be created from several threads (one owner each): the table is filled exactly once
    static std::once_flag once;
    std::call_once(once, roctx_init_once);
}
static void roctx_init_once(void)
{
    g_roctx.tried = true;
    const char* on = getenv("MM_ROCTX");
    if (!on || on[0] != '1') return;
    static const char* const libs[] = { "librocprofiler-sdk-roctx.so", "libroctx64.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so.4" };
    for (unsigned i = 0; i < sizeof(libs) / sizeof(libs[0]) && !g_roctx.push; ++i) {
        void* h = dlopen(libs[i], RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        g_roctx.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        g_roctx.pop = (int (*)(void))dlsym(h, "roctxRangePop");
        if (!g_roctx.push || !g_roctx.pop) { g_roctx.push = nullptr; g_roctx.pop = nullptr; }
    }
}
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(g_roctx.push != nullptr) { if (on) (void)g_roctx.push(name); }
    ~RoctxRange() { if (on) (void)g_roctx.pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

static double host_now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

#define HIPCHK(e, call)                                   \
    do {                                                  \
        hipError_t _rc = (call);                          \
        if (_rc != hipSuccess) {                          \
            (e)->last_hip = (int)_rc;                     \
            return _rc == hipErrorOutOfMemory ? MM_ERR_OOM : MM_ERR_HIP; \
        }                                                 \
    } while (0)

// HIP's current device is per-THREAD state, and the callers this ABI is shaped for do not keep a
// thread: a dirty NIF runs on whichever dirty scheduler is free (native/mm_nif.c), the resource
// destructor on whichever thread collects the handle.  Every entry point that touches the device
// therefore selects the engine's device first and puts the caller's back on the way out, so an
// engine on device k works from any thread and leaves the host program's own device alone.
struct DeviceScope {
    int prev, dev;
    hipError_t err;
    bool ok;
    explicit DeviceScope(int d) : prev(-1), dev(d), err(hipSuccess), ok(true)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) {
            err = hipSetDevice(dev);
            ok = err == hipSuccess;
        }
    }
    ~DeviceScope()
    {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
#define ON_ENGINE_DEVICE(e)                      \
    DeviceScope _dev_scope((e)->cfg.device);     \
    if (!_dev_scope.ok) {                        \
        (e)->last_hip = (int)_dev_scope.err;      \
        return MM_ERR_HIP;                       \
    }

extern "C" uint32_t mm_abi_version(void) { return MM_ABI_VERSION; }

extern "C" const char* mm_strerror(int status)
{
    switch (status) {
    case MM_OK: return "ok";
    case MM_ERR_INVALID_ARG: return "invalid argument";
    case MM_ERR_NO_DEVICE: return "no usable HIP device";
    case MM_ERR_OOM: return "out of memory";
    case MM_ERR_FULL: return "pool capacity exhausted";
    case MM_ERR_HIP: return "HIP runtime error";
    case MM_ERR_INTERNAL: return "device-side invariant violated";
    case MM_ERR_ABI: return "ABI version mismatch";
    case MM_ERR_RANGE: return "match range out of bounds";
    case MM_ERR_STATE: return "a tick failed: mm_reset or mm_restore first";
    default: return "unknown status";
    }
}

extern "C" int mm_config_default(mm_config* cfg)
{
    if (!cfg) return MM_ERR_INVALID_ARG;
    memset(cfg, 0, sizeof(*cfg));
    static const int32_t ref[7][2] = { {0, 1499}, {1500, 1999}, {2000, 2499}, {2500, 2999},
                                       {3000, 3499}, {3500, 3999}, {4000, 5000} };
    cfg->abi_version = MM_ABI_VERSION;
    cfg->n_groups = 7;
    for (int g = 0; g < 7; ++g) { cfg->groups[g].from = ref[g][0]; cfg->groups[g].to = ref[g][1]; }
    cfg->default_group = 7 / 2 + 1;
    cfg->n_modes = 1;
    cfg->modes[0].team_size = 1;
    cfg->modes[0].teams = 2;
    cfg->modes[0].window = 50;
    cfg->modes[0].n_roles = 1;
    cfg->modes[0].role_quota[0] = 1;
    cfg->capacity = 1u << 20;
    cfg->device = 0;
    return MM_OK;
}

extern "C" int mm_find_rating_group(const mm_config* cfg, double rating, uint32_t* group)
{
    if (!cfg || !group) return MM_ERR_INVALID_ARG;
    uint32_t g = cfg->default_group;
    if (rating == rating) {
        for (uint32_t k = 0; k < cfg->n_groups && k < MM_MAX_GROUPS; ++k)
            if (rating >= (double)cfg->groups[k].from && rating <= (double)cfg->groups[k].to) { g = k; break; }
    }
    *group = g;
    return MM_OK;
}

static int cfg_validate(const mm_config* c)
{
    if (c->abi_version != MM_ABI_VERSION) return MM_ERR_ABI;
    if (c->n_groups < 1 || c->n_groups > MM_MAX_GROUPS) return MM_ERR_INVALID_ARG;
    if (c->default_group >= c->n_groups) return MM_ERR_INVALID_ARG;
    if (c->n_modes < 1 || c->n_modes > MM_MAX_MODES) return MM_ERR_INVALID_ARG;
    if (c->capacity < 1 || c->capacity > (1u << 28)) return MM_ERR_INVALID_ARG;
    for (uint32_t m = 0; m < c->n_modes; ++m) {
        const mm_mode_config* mc = &c->modes[m];
        if (mc->team_size < 1 || mc->team_size > 8) return MM_ERR_INVALID_ARG;
        if (mc->teams < 2 || mc->teams > MM_MAX_TEAMS) return MM_ERR_INVALID_ARG;
        if (mc->teams * mc->team_size > MM_MAX_LOBBY) return MM_ERR_INVALID_ARG;
        if (mc->n_roles < 1 || mc->n_roles > MM_MAX_ROLES) return MM_ERR_INVALID_ARG;
        if (mc->window > 0x3FFFFFFFu) return MM_ERR_INVALID_ARG;
        uint32_t s = 0;
        for (uint32_t r = 0; r < mc->n_roles; ++r) s += mc->role_quota[r];
        if (s != mc->team_size) return MM_ERR_INVALID_ARG;
    }
    return MM_OK;
}

static BucketCfg make_bucket_cfg(const mm_config& c)
{
    BucketCfg B;
    memset(&B, 0, sizeof(B));
    B.n_groups = c.n_groups;
    B.n_modes = c.n_modes;
    B.default_group = c.default_group;
    B.capacity = c.capacity;
    for (uint32_t g = 0; g < c.n_groups; ++g) { B.from[g] = c.groups[g].from; B.to[g] = c.groups[g].to; }
    for (uint32_t m = 0; m < c.n_modes; ++m) {
        B.n_roles[m] = c.modes[m].n_roles;
        for (uint32_t r = 0; r < c.modes[m].n_roles; ++r)
            if (c.modes[m].role_quota[r]) B.quota_mask[m] |= 1u << r;
    }
    return B;
}

static ModeDev make_mode_dev(const mm_mode_config& mc)
{
    ModeDev M;
    memset(&M, 0, sizeof(M));
    M.team_size = mc.team_size;
    M.teams = mc.teams;
    M.window = mc.window;
    M.eqmask = ((mc.flags & MM_MODE_REGION_FILTER) ? (0xFFu << 4) : 0u) |
               ((mc.flags & MM_MODE_PARTY_FILTER) ? (0xFu << 12) : 0u);
    M.n_roles = mc.n_roles;
    M.L = mc.teams * mc.team_size;
    for (uint32_t r = 0; r < mc.n_roles; ++r) M.quota[r] = mc.role_quota[r];
    return M;
}

// ONE copy stream per device for all engines of the process: the runtime maps streams onto a few hardware queues
// (four by default), and two engines with a second stream each had their main streams share a queue — two pools on
// one GPU ran one after the other (concurrent_pools 152 -> 87 M matched players/s until this was shared).
static std::mutex g_copy_mu;
static hipStream_t g_copy_stream[64];
static int g_copy_refs[64];
static int copy_stream_acquire(int dev, hipStream_t* out)
{
    std::lock_guard<std::mutex> lk(g_copy_mu);
    if (dev < 0 || dev >= 64) return 1;
    if (g_copy_refs[dev] == 0 && hipStreamCreateWithFlags(&g_copy_stream[dev], hipStreamNonBlocking) != hipSuccess) return 1;
    ++g_copy_refs[dev];
    *out = g_copy_stream[dev];
    return 0;
}
static void copy_stream_release(int dev)
{
    std::lock_guard<std::mutex> lk(g_copy_mu);
    if (dev < 0 || dev >= 64 || g_copy_refs[dev] == 0) return;
    if (--g_copy_refs[dev] == 0) {
        (void)hipStreamSynchronize(g_copy_stream[dev]);
        (void)hipStreamDestroy(g_copy_stream[dev]);
        g_copy_stream[dev] = nullptr;
    }
}

extern "C" void mm_engine_destroy(mm_engine* e)
{
    if (!e) return;
    DeviceScope dev_scope(e->cfg.device);   // frees and the stream belong to the engine's device
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    (void)hipFree(e->d_q_rating);
    (void)hipFree(e->d_q_cons);
    (void)hipFree(e->d_q_slot);
    (void)hipFree(e->d_chains);
    (void)hipFree(e->d_state);
    (void)hipFree(e->d_released);
    (void)hipFree(e->d_counters);
    (void)hipFree(e->d_wave_hist);
    (void)hipFree(e->d_in_rating);
    (void)hipFree(e->d_in_cons);
    (void)hipFree(e->d_in_group);
    (void)hipFree(e->d_in_slot);
    (void)hipFree(e->d_in_sel);
    (void)hipFree(e->d_out_slots);
    (void)hipFree(e->d_out_score);
    (void)hipFree(e->d_out_pass);
    (void)hipFree(e->d_pchains);
    (void)hipFree(e->d_tchains);
    (void)hipFree(e->d_tk_chunk);
    (void)hipFree(e->d_tk_fv2);
    (void)hipFree(e->d_tk_fpos);
    (void)hipFree(e->d_tk_memb);
    (void)hipFree(e->d_tk_bitsB);
    (void)hipFree(e->d_tk_chunkB);
    (void)hipFree(e->d_tk_fdone);
    (void)hipFree(e->d_tk_sqi);
    for (int b = 0; b < 2; ++b) {
        (void)hipFree(e->d_pk_key[b]); (void)hipFree(e->d_pk_oidx[b]);
        (void)hipFree(e->d_pk_nx16[b]); (void)hipFree(e->d_pk_bits[b]);
    }
    (void)hipFree(e->d_pk_scratch);
    (void)hipFree(e->d_pk_wpre);
    (void)hipFree(e->d_pk_g16);
    (void)hipFree(e->d_pk_rec1);
    for (int b = 0; b < 2; ++b) { (void)hipFree(e->d_pk_exa[b]); (void)hipFree(e->d_pk_bitsp[b]); (void)hipFree(e->d_pk_headp[b]); }
    (void)hipFree(e->d_pk_tilectl);
    (void)hipFree(e->d_pk_grec);
    (void)hipFree(e->d_pk_pbar);
    (void)hipFree(e->d_pack);
    if (e->h_pchains) (void)hipHostFree(e->h_pchains);
    if (e->h_look_seq) (void)hipHostFree(e->h_look_seq);
    if (e->h_tchains) (void)hipHostFree(e->h_tchains);
    if (e->h_rslots) (void)hipHostFree(e->h_rslots);
    if (e->h_rscore) (void)hipHostFree(e->h_rscore);
    if (e->h_rpass) (void)hipHostFree(e->h_rpass);
    if (e->h_chains) (void)hipHostFree(e->h_chains);
    if (e->h_counters) (void)hipHostFree(e->h_counters);
    for (int i = 0; i < 4; ++i)
        if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
    for (int i = 0; i < 2; ++i)
        if (e->ev_nx[i]) (void)hipEventDestroy(e->ev_nx[i]);
    for (uint32_t i = 0; i < MM_MAX_GROUPS; ++i)
        if (e->ev_grp[i]) (void)hipEventDestroy(e->ev_grp[i]);
    if (e->ev_copy_pending) (void)hipEventSynchronize(e->ev_copy);     // (nothing of this engine is left on the shared stream)
    if (e->copy_stream) { e->copy_stream = nullptr; copy_stream_release(e->cfg.device); }
    if (e->ev_copy) (void)hipEventDestroy(e->ev_copy);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

static int engine_reset_device(mm_engine* e)
{
    const uint32_t words = e->n_chains * (uint32_t)(sizeof(ChainDev) / 4);
    hipLaunchKernelGGL(k_reset, dim3((words + 255) / 256), dim3(256), 0, e->stream, e->n_chains, e->d_chains);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipMemsetAsync(e->d_state, 0, e->cfg.capacity, e->stream));
    HIPCHK(e, hipMemsetAsync(e->d_counters, 0, 2 * sizeof(uint32_t), e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return MM_OK;
}

// ---- mm_tuning (include/mm_engine.h): the table of fields, their environment defaults and ranges ----
struct TuneDesc { const char* name; const char* env; uint32_t off, def, lo, hi, kind; };   // kind 1: zero or a power of two in [lo, hi]
#define MM_TD(f, envname, def, lo, hi, kind) { #f, envname, (uint32_t)offsetof(mm_tuning, f), def, lo, hi, kind }
static const TuneDesc k_tune[] = {
    MM_TD(force_generic, "MM_FORCE_GENERIC", 0u, 0u, 1u, 0u),
    MM_TD(debug, "MM_PAIR_DEBUG", 0u, 0u, 1u, 0u),
    MM_TD(results_early, "MM_RESULTS_EARLY", 1u, 0u, 1u, 0u),
    MM_TD(results_tail, "MM_RESULTS_TAIL", 1u, 0u, 1u, 0u),
    MM_TD(look_poll, "MM_LOOK_POLL", 0u, 0u, 1u, 0u),
    MM_TD(fail_tick, "MM_DEBUG_FAIL_TICK", 0u, 0u, 0xFFFFFFFFu, 0u),
    MM_TD(pair_persist, "MM_PAIR_PERSIST", 1u, 0u, 1u, 0u),
    MM_TD(pair_ptiles, "MM_PAIR_PTILES", 32u, 1u, 32u, 0u),
    MM_TD(pair_pbatch, "MM_PAIR_PBATCH", 48u, 1u, 4096u, 0u),
    MM_TD(pair_batch, "MM_PAIR_BATCH", 48u, 1u, 4096u, 0u),
    MM_TD(pair_ptimeout_us, "MM_PAIR_PTIMEOUT_US", 5000u, 0u, 1000000u, 0u),
    MM_TD(pair_pinject, "MM_PAIR_PINJECT", 0u, 0u, 0xFFFFFFFFu, 0u),
    MM_TD(pair_tiles_max, "MM_PAIR_TILES", PK_TILES_MAX, 1u, 4096u, 0u),
    MM_TD(pair_tile_fixed, "MM_PAIR_TILE", 0u, 0u, 1u, 0u),
    MM_TD(pair_xcd, "MM_PAIR_XCD", 1u, 0u, 1u, 0u),
    MM_TD(pair_group_min, "MM_PAIR_GROUP", PK_GROUP_MIN, 0u, 0xFFFFFFFFu, 0u),
    MM_TD(pair_nxseg, "MM_PAIR_NXSEG", 0u, 64u, NXI_SEG, 1u),
    MM_TD(pair_nxstage, "MM_PAIR_NXSTAGE", 0u, 0u, 0xFFFFFFFFu, 0u),
    MM_TD(pair_tune, "MM_PAIR_TUNE", 0u, 0u, 0xFFFFFFFFu, 0u),
    MM_TD(team_batch, "MM_TEAM_BATCH", 16u, 1u, 4096u, 0u),
    MM_TD(team_f2, "MM_TEAM_F2", 32u, 0u, 0xFFFFFFFFu, 0u),
    MM_TD(team_rebuild, "MM_TEAM_REBUILD", 8u, 1u, 4096u, 0u),
    MM_TD(team_emit_max, "MM_TEAM_EMIT_MAX", TC_EMIT_MAX, 1u, TC_EMIT_MAX, 0u),
    MM_TD(team_split, "MM_TEAM_SPLIT", 1u, 0u, 1u, 0u),
    MM_TD(team_fwait, "MM_TEAM_FWAIT", 1u << 14, 0u, 0xFFFFFFFFu, 0u),
    MM_TD(team_fix_max, "MM_TEAM_FIXMAX", 0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u),
    MM_TD(team_fix_t8, "MM_TEAM_FIXT8", 10u, 0u, 0xFFFFFFFFu, 0u),
    MM_TD(team_fix_t4, "MM_TEAM_FIXT4", 64u, 0u, 0xFFFFFFFFu, 0u),
    MM_TD(team_pull_xcd, "MM_TEAM_PULLX", 1u, 0u, 1u, 0u),
    MM_TD(team_nowait, "MM_TEAM_NOWAIT", 0u, 0u, 0xFFFFFFFFu, 0u),
    MM_TD(team_late, "MM_TEAM_LATE", 6u, 0u, 0xFFFFFFFFu, 0u),
    MM_TD(team_late0, "MM_TEAM_LATE0", 512u, 0u, 0xFFFFFFFFu, 0u),
    MM_TD(team_cap, "MM_TEAM_CAP", TT_SCAN_CAP, 1u, 4096u, 0u),
};
#undef MM_TD
static const uint32_t k_tune_n = (uint32_t)(sizeof(k_tune) / sizeof(k_tune[0]));
static_assert(sizeof(mm_tuning) == (sizeof(k_tune) / sizeof(k_tune[0]) + 1u) * sizeof(uint32_t), "every field of mm_tuning has its row in k_tune");

static bool tune_in_range(const TuneDesc& d, uint32_t v)
{
    if (d.kind == 1u) return v == 0u || (v >= d.lo && v <= d.hi && (v & (v - 1u)) == 0u);
    return v >= d.lo && v <= d.hi;
}
static uint32_t& tune_field(mm_tuning* t, const TuneDesc& d) { return *(uint32_t*)((char*)t + d.off); }

// the defaults: built-in values, each overridden by its environment variable when that parses and is in range
static void tune_defaults(mm_tuning* full)
{
    full->size = (uint32_t)sizeof(mm_tuning);
    for (uint32_t i = 0; i < k_tune_n; ++i) {
        const TuneDesc& d = k_tune[i];
        uint32_t v = d.def;
        const char* s = getenv(d.env);
        if (s && *s) {
            char* end = NULL;
            unsigned long long x = strtoull(s, &end, 0);
            bool ok = end && *end == 0 && end != s && x <= 0xFFFFFFFFull;
            if (!ok && d.off == (uint32_t)offsetof(mm_tuning, pair_tile_fixed) && (s[0] == 'm' || s[0] == 'M')) { x = 1u; ok = true; }   // MM_PAIR_TILE=max
            if (ok && tune_in_range(d, (uint32_t)x)) v = (uint32_t)x;
            else fprintf(stderr, "[mm-engine] %s=%s is not a value of mm_tuning.%s: ignored, %u stays\n", d.env, s, d.name, v);
        }
        tune_field(full, d) = v;
    }
}

extern "C" int mm_tuning_default(mm_tuning* t)
{
    if (!t || t->size < sizeof(uint32_t)) return MM_ERR_INVALID_ARG;
    mm_tuning full;
    tune_defaults(&full);
    const uint32_t n = t->size < (uint32_t)sizeof(full) ? t->size & ~3u : (uint32_t)sizeof(full);
    full.size = n;
    memcpy(t, &full, n);
    return MM_OK;
}

extern "C" int mm_tuning_set(mm_tuning* t, const char* name, uint32_t value)
{
    if (!t || !name) return MM_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < k_tune_n; ++i) {
        const TuneDesc& d = k_tune[i];
        if (strcmp(d.name, name) != 0) continue;
        if (d.off + sizeof(uint32_t) > t->size) return MM_ERR_INVALID_ARG;
        if (!tune_in_range(d, value)) return MM_ERR_RANGE;
        tune_field(t, d) = value;
        return MM_OK;
    }
    return MM_ERR_INVALID_ARG;
}

extern "C" const char* mm_tuning_name(uint32_t index) { return index < k_tune_n ? k_tune[index].name : NULL; }

extern "C" int mm_tuning_get(const mm_engine* e, mm_tuning* t)
{
    if (!e || !t || t->size < sizeof(uint32_t)) return MM_ERR_INVALID_ARG;
    mm_tuning full = e->tn;
    const uint32_t n = t->size < (uint32_t)sizeof(full) ? t->size & ~3u : (uint32_t)sizeof(full);
    full.size = n;
    memcpy(t, &full, n);
    return MM_OK;
}

extern "C" int mm_engine_create(const mm_config* cfg, mm_engine** out) { return mm_engine_create_ex(cfg, NULL, out); }

extern "C" int mm_engine_create_ex(const mm_config* cfg, const mm_tuning* tuning, mm_engine** out)
{
    try {
        if (!cfg || !out) return MM_ERR_INVALID_ARG;
        *out = NULL;
        int rc = cfg_validate(cfg);
        if (rc) return rc;
        mm_tuning tn;
        tune_defaults(&tn);
        if (tuning) {
            if (tuning->size < sizeof(uint32_t)) return MM_ERR_INVALID_ARG;
            for (uint32_t i = 0; i < k_tune_n; ++i) {
                const TuneDesc& d = k_tune[i];
                if (d.off + sizeof(uint32_t) > tuning->size) continue;        // a shorter record of an older caller: the default stays
                const uint32_t v = *(const uint32_t*)((const char*)tuning + d.off);
                if (!tune_in_range(d, v)) return MM_ERR_RANGE;
                tune_field(&tn, d) = v;
            }
        }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
            return MM_ERR_NO_DEVICE;
        roctx_init();
        mm_engine* e = new (std::nothrow) mm_engine();
        if (!e) return MM_ERR_OOM;
        e->cfg = *cfg;
        e->tn = tn;
        e->n_chains = cfg->n_modes * cfg->n_groups;
        e->last_hip = 0;
        e->next_slot = 0;
        e->cancel_pending = 0;
        e->r_n = 0;
        e->r_L = 2;
        e->in_cap = 0;
        e->wave_hist_rows = 0;
        e->live_upper = 0;
        e->poisoned = false;
        e->ticks_seen = 0;
        {
            // the tuning record (include/mm_engine.h, mm_tuning): validated by the caller below; the engine keeps it (mm_tuning_get)
            // and its working copies — some of which change while it runs (pair_persist after a PF_XCD stop)
            const mm_tuning& t = e->tn;
            e->fault_tick = t.fail_tick;
            e->force_generic = t.force_generic != 0u;
            e->pair_debug = t.debug != 0u;
            e->pair_tune = t.pair_tune;
            e->pair_tile_fixed = t.pair_tile_fixed != 0u;
            e->pair_tiles_max = t.pair_tiles_max;
            e->pair_xcd = t.pair_xcd != 0u;
            e->pair_group_min = t.pair_group_min;
            e->round_ctr = 0;
            e->pair_persist = t.pair_persist != 0u;
            e->pair_ptiles = t.pair_ptiles;
            e->pair_pinject = t.pair_pinject;
            // the first barrier of a launch is where a workgroup that found no CU is waited for (another engine's launch in the
            // way ends within a millisecond or two); behind it everybody is on the chip and only slow, never absent
            // — 5 ms, half a tick period of the stream (20 ms until round 5: two whole periods of spinning whenever somebody
            // else held the CUs, and kp_round is a cheap way on); 40 x that behind it (200 ms), as before.  In 100 MHz ticks.
            e->pair_ptimeout[0] = t.pair_ptimeout_us * 100u;
            e->pair_ptimeout[1] = e->pair_ptimeout[0] * 40u;
            e->pair_pcool = 0;
            e->pair_pstops = 0;
            e->pair_nxseg = t.pair_nxseg;
            e->pair_nxstage = t.pair_nxstage;
            e->pair_pbatch = t.pair_pbatch;       // 48 / 64 / 96 measured: 95.7 / 94.6 / 93.5 M matched players/s (round 4)
            e->pair_batch = t.pair_batch;         // 16 / 32 / 48 / 64 measured: 48 by 1-2 %
            e->team_batch = t.team_batch;
            e->team_f2 = t.team_f2;               // 24 / 40 measured within 0.1 ms of each other
            e->team_rebuild = t.team_rebuild;     // 4 / 6 / 8 measured within 0.1 ms of each other, 16: +0.9 ms, every pass: +1.4 ms
            e->team_emit_max = t.team_emit_max;   // emitter workgroups per chain and launch at most (kt_pack resets vis[] for this many workers: TV_AHEAD)
            e->team_split = t.team_split;
            e->team_fwait = t.team_fwait;         // ~5 ms of polls, then the chaser helps itself
            e->team_fix_max = t.team_fix_max;     // (measured: mending always wins once a task is a quarter wave's)
            e->team_fix_t8 = t.team_fix_t8;
            e->team_fix_t4 = t.team_fix_t4;
            e->team_pull_xcd = t.team_pull_xcd;
            e->team_nowait = t.team_nowait;
            memset(&e->ps, 0, sizeof(e->ps));
            e->ps.mode = 0xFFFFFFFFu;
            e->team_seq = 0;
            e->team_late0 = t.team_late0;
            e->tk_last_len.assign(e->n_chains, 0u);
            e->results_early = t.results_early != 0u;
            // measured on cfg-3 (profiles/r03_ab_team_late.txt): 0: 12.99 ms, 3: 12.50, 6: 12.46, 12: 12.76, 20: 13.67, 40: 15.73 per tick,
            // and again with one launch per pass (kt_fc): 0: 10.9, 6: 9.89, 12: 10.01, 20: 10.6 — a look-up costs the chaser ~4 us
            // (dependent trips to memory at ~1.2 us each), so it only beats a pass kernel while a pass seats a handful of lobbies
            e->team_late = t.team_late;
            e->team_cap = t.team_cap;             // <= 4096: kt_f notes members as 13-bit sub-queue offsets (TF_REL_BITS)
            // look_poll OFF by default.  Measured (profiles/r05_ab_look_poll.txt, cfg-2 / cfg-3, 40 steps, twice each on one box): the
            // median step gains 0.03 ms of 10.1 (1v1) and 0.05 ms of 8.5 (5v5) — the copy + synchronisation of a look is NOT where a
            // tick's time goes — and every polled 1v1 run had ONE step of 17 ms: a host thread that spins for the whole tick is what a
            // container's CPU quota throttles first (a dirty scheduler of the BEAM would fare no better).
            e->look_poll = t.look_poll != 0u;
            e->results_tail_kernel = t.results_tail != 0u;
        }
        const size_t cap = cfg->capacity;
    #define CREATE_CHK(call)                                                 \
        do {                                                                 \
            hipError_t _rc = (call);                                         \
            if (_rc != hipSuccess) {                                         \
                int code = _rc == hipErrorOutOfMemory ? MM_ERR_OOM : MM_ERR_HIP; \
                mm_engine_destroy(e);                                        \
                return code;                                                 \
            }                                                                \
        } while (0)
        DeviceScope dev_scope(cfg->device);        // the caller's device is put back when create returns
        if (!dev_scope.ok) {
            mm_engine_destroy(e);
            return MM_ERR_HIP;
        }
        CREATE_CHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        for (int i = 0; i < 4; ++i) CREATE_CHK(hipEventCreate(&e->ev[i]));
        for (int i = 0; i < 2; ++i) CREATE_CHK(hipEventCreate(&e->ev_nx[i]));
        e->ev_nx_set = false;
        for (uint32_t i = 0; i < cfg->n_groups; ++i) CREATE_CHK(hipEventCreate(&e->ev_grp[i]));
        if (copy_stream_acquire(cfg->device, &e->copy_stream)) { e->copy_stream = nullptr; mm_engine_destroy(e); return MM_ERR_HIP; }
        CREATE_CHK(hipEventCreate(&e->ev_copy));
        e->ev_copy_pending = false;
        e->r_based = false;
        memset(e->r_cnt, 0, sizeof(e->r_cnt));
        memset(e->r_pre, 0, sizeof(e->r_pre));
        CREATE_CHK(hipMalloc((void**)&e->d_q_rating, e->n_chains * cap * sizeof(int32_t)));
        CREATE_CHK(hipMalloc((void**)&e->d_q_cons, e->n_chains * cap * sizeof(uint32_t)));
        CREATE_CHK(hipMalloc((void**)&e->d_q_slot, e->n_chains * cap * sizeof(uint32_t)));
        CREATE_CHK(hipMalloc((void**)&e->d_chains, e->n_chains * sizeof(ChainDev)));
        CREATE_CHK(hipMalloc((void**)&e->d_state, cap));
        CREATE_CHK(hipMalloc((void**)&e->d_released, (cap + 64) * sizeof(uint32_t)));
        CREATE_CHK(hipMalloc((void**)&e->d_pack, (size_t)MM_PACK_MAX * (MM_MAX_LOBBY + 2u) * sizeof(uint32_t)));
        CREATE_CHK(hipMalloc((void**)&e->d_counters, 2 * sizeof(uint32_t)));
        // emission logs: a group can emit at most (cap + lobby) / 2 lobbies of >= 2 players
        e->out_slot_stride = (uint32_t)(cap + 64);
        e->out_rec_stride = (uint32_t)(cap / 2 + 16);
        CREATE_CHK(hipMalloc((void**)&e->d_out_slots, (size_t)cfg->n_groups * e->out_slot_stride * sizeof(uint32_t)));
        CREATE_CHK(hipMalloc((void**)&e->d_out_score, (size_t)cfg->n_groups * e->out_rec_stride * sizeof(float)));
        CREATE_CHK(hipMalloc((void**)&e->d_out_pass, (size_t)cfg->n_groups * e->out_rec_stride * sizeof(uint32_t)));
        {
            e->pk_stride = (uint32_t)((cap + 63) & ~(size_t)63);
            const size_t gc = (size_t)cfg->n_groups * e->pk_stride;
            e->pk_bits_stride = (uint32_t)(cap / 32 + 4);
            CREATE_CHK(hipMalloc((void**)&e->d_pchains, cfg->n_groups * sizeof(PairChain)));
            e->pk_max_tiles = (uint32_t)(cap / (PK_TMAX / 4u) + 2);   // tiles of the smallest tile length
            for (int b = 0; b < 2; ++b) {
                CREATE_CHK(hipMalloc((void**)&e->d_pk_key[b], (gc + 64) * sizeof(uint32_t)));
                CREATE_CHK(hipMalloc((void**)&e->d_pk_oidx[b], gc * sizeof(uint32_t)));
                CREATE_CHK(hipMalloc((void**)&e->d_pk_nx16[b], (gc + 64) * sizeof(uint16_t)));
                CREATE_CHK(hipMalloc((void**)&e->d_pk_bits[b], (size_t)cfg->n_groups * e->pk_bits_stride * sizeof(uint32_t)));
            }
            CREATE_CHK(hipMalloc((void**)&e->d_pk_scratch, gc * sizeof(uint32_t)));
            CREATE_CHK(hipMalloc((void**)&e->d_pk_g16, (gc + 64) * sizeof(uint16_t)));
            CREATE_CHK(hipMalloc((void**)&e->d_pk_rec1, gc * sizeof(uint32_t)));
            for (int b = 0; b < 2; ++b) {
                CREATE_CHK(hipMalloc((void**)&e->d_pk_exa[b], (gc + 64) * sizeof(uint16_t)));
                CREATE_CHK(hipMalloc((void**)&e->d_pk_bitsp[b], (size_t)cfg->n_groups * e->pk_bits_stride * sizeof(uint32_t)));
                CREATE_CHK(hipMalloc((void**)&e->d_pk_headp[b], (size_t)cfg->n_groups * e->pk_max_tiles * sizeof(uint32_t)));
            }
            CREATE_CHK(hipMalloc((void**)&e->d_pk_wpre, (size_t)cfg->n_groups * e->pk_bits_stride * sizeof(uint16_t)));
            CREATE_CHK(hipMalloc((void**)&e->d_pk_tilectl, (size_t)TC_N * cfg->n_groups * e->pk_max_tiles * sizeof(uint32_t)));
            CREATE_CHK(hipMalloc((void**)&e->d_pk_pbar, (size_t)MM_MAX_GROUPS * 2u * sizeof(unsigned long long) + 64u));
            // two tiles' worth of entries per group of PK_GS tiles, whatever the tile length of the batch
            e->pk_gstride = (uint32_t)(2u * (e->pk_stride / PK_GS) + 4u * PK_TMAX);
            CREATE_CHK(hipMalloc((void**)&e->d_pk_grec, (size_t)cfg->n_groups * e->pk_gstride * sizeof(uint4)));
            // an entry carries the round it was made in (the walk takes one of its own round only): no stamp to start with
            CREATE_CHK(hipMemsetAsync(e->d_pk_grec, 0xFF, (size_t)cfg->n_groups * e->pk_gstride * sizeof(uint4), e->stream));
            CREATE_CHK(hipHostMalloc((void**)&e->h_pchains, cfg->n_groups * sizeof(PairChain), hipHostMallocDefault));
            // a tick emits at most (cap + lobby) / 2 lobbies per group, cap + lobbies-in-progress players overall
            // (+ one lobby of slack per group: a group's region is sized by the players it holds when the tick begins)
            CREATE_CHK(hipHostMalloc((void**)&e->h_rslots, (cap + (size_t)3 * MM_MAX_LOBBY * cfg->n_groups + 64) * sizeof(uint32_t), hipHostMallocDefault));
            CREATE_CHK(hipHostMalloc((void**)&e->h_rscore, (cap / 2 + (size_t)3 * MM_MAX_LOBBY * cfg->n_groups + 64) * sizeof(float), hipHostMallocDefault));
            CREATE_CHK(hipHostMalloc((void**)&e->h_rpass, (cap / 2 + (size_t)3 * MM_MAX_LOBBY * cfg->n_groups + 64) * sizeof(uint32_t), hipHostMallocDefault));
            CREATE_CHK(hipMemsetAsync(e->d_pchains, 0, cfg->n_groups * sizeof(PairChain), e->stream));
            e->tk_chunk_stride = (uint32_t)(e->pk_stride / TT_CH + 2);
            CREATE_CHK(hipMalloc((void**)&e->d_tchains, cfg->n_groups * sizeof(TeamChain)));
            CREATE_CHK(hipMalloc((void**)&e->d_tk_chunk, (size_t)cfg->n_groups * MM_MAX_ROLES * e->tk_chunk_stride * sizeof(uint32_t)));
            CREATE_CHK(hipMalloc((void**)&e->d_tk_fv2, gc * sizeof(uint32_t)));
            e->tk_memb = 0;
            for (uint32_t k = 0; k < cfg->n_modes; ++k) {
                const uint32_t L = cfg->modes[k].teams * cfg->modes[k].team_size;
                if (L > 2u && L - 1u > e->tk_memb) e->tk_memb = L - 1u;
            }
            if (e->tk_memb) {
                CREATE_CHK(hipMalloc((void**)&e->d_tk_fpos, gc * sizeof(uint32_t)));
                CREATE_CHK(hipMalloc((void**)&e->d_tk_memb, gc * e->tk_memb * sizeof(uint16_t)));
                CREATE_CHK(hipMalloc((void**)&e->d_tk_bitsB, (size_t)cfg->n_groups * e->pk_bits_stride * sizeof(uint32_t)));
                CREATE_CHK(hipMalloc((void**)&e->d_tk_chunkB, (size_t)cfg->n_groups * MM_MAX_ROLES * e->tk_chunk_stride * sizeof(uint32_t)));
                CREATE_CHK(hipMalloc((void**)&e->d_tk_sqi, gc * sizeof(uint32_t)));
                CREATE_CHK(hipMalloc((void**)&e->d_tk_fdone, (size_t)cfg->n_groups * e->tk_chunk_stride * sizeof(uint32_t)));
                CREATE_CHK(hipMemsetAsync(e->d_tk_fdone, 0, (size_t)cfg->n_groups * e->tk_chunk_stride * sizeof(uint32_t), e->stream));
            }
            CREATE_CHK(hipHostMalloc((void**)&e->h_tchains, cfg->n_groups * sizeof(TeamChain), hipHostMallocDefault));
            CREATE_CHK(hipMemsetAsync(e->d_tchains, 0, cfg->n_groups * sizeof(TeamChain), e->stream));
        }
        CREATE_CHK(hipHostMalloc((void**)&e->h_chains, e->n_chains * sizeof(ChainDev), hipHostMallocDefault));
        CREATE_CHK(hipHostMalloc((void**)&e->h_look_seq, 64, hipHostMallocDefault));
        *e->h_look_seq = 0;
        e->look_seq = 0;
        CREATE_CHK(hipHostMalloc((void**)&e->h_counters, 2 * sizeof(uint32_t), hipHostMallocDefault));
        e->h_state.assign(cap, MM_ST_FREE);
        if (e->tk_memb) {
            // the first launch of a kernel is slow (8 ms for kt_late in a stream's first late tick): take it here, on
            // chain records that say "not walked" (the workgroups return at once)
            TeamParams W;
            memset(&W, 0, sizeof(W));
            W.n_groups = cfg->n_groups;
            W.tchains = e->d_tchains;
            W.chains = e->d_chains;
            hipLaunchKernelGGL(kt_late, dim3(cfg->n_groups), dim3(TL_THREADS), 0, e->stream, W);
            hipLaunchKernelGGL(kt_fc, dim3(cfg->n_groups), dim3(TT_CH), 0, e->stream, W);
            CREATE_CHK(hipGetLastError());
        }
    #undef CREATE_CHK
        rc = engine_reset_device(e);
        if (rc) { mm_engine_destroy(e); return rc; }
        *out = e;
        return MM_OK;
    } catch (const std::bad_alloc&) {
        return MM_ERR_OOM;
    } catch (...) {
        return MM_ERR_INTERNAL;
    }
}

extern "C" int mm_reset(mm_engine* e)
{
    try {
        if (!e) return MM_ERR_INVALID_ARG;
        ON_ENGINE_DEVICE(e);
        std::fill(e->h_state.begin(), e->h_state.end(), (uint8_t)MM_ST_FREE);
        e->next_slot = 0;
        e->cancel_pending = 0;
        e->r_n = 0;
        e->live_upper = 0;
        std::fill(e->tk_last_len.begin(), e->tk_last_len.end(), 0u);
        const int rc = engine_reset_device(e);
        if (rc == MM_OK) e->poisoned = false;
        return rc;
    } catch (const std::bad_alloc&) {
        return MM_ERR_OOM;
    } catch (...) {
        return MM_ERR_INTERNAL;
    }
}

static int ensure_staging(mm_engine* e, size_t n)
{
    if (n <= e->in_cap) return MM_OK;
    size_t cap = e->in_cap ? e->in_cap : 4096;
    while (cap < n) cap *= 2;
    (void)hipFree(e->d_in_rating); e->d_in_rating = NULL;
    (void)hipFree(e->d_in_cons); e->d_in_cons = NULL;
    (void)hipFree(e->d_in_group); e->d_in_group = NULL;
    (void)hipFree(e->d_in_slot); e->d_in_slot = NULL;
    (void)hipFree(e->d_in_sel); e->d_in_sel = NULL;
    e->in_cap = 0;
    HIPCHK(e, hipMalloc((void**)&e->d_in_rating, cap * sizeof(int32_t)));
    HIPCHK(e, hipMalloc((void**)&e->d_in_cons, cap * sizeof(uint32_t)));
    HIPCHK(e, hipMalloc((void**)&e->d_in_group, cap));
    HIPCHK(e, hipMalloc((void**)&e->d_in_slot, cap * sizeof(uint32_t)));
    HIPCHK(e, hipMalloc((void**)&e->d_in_sel, cap * sizeof(uint32_t)));
    e->in_cap = cap;
    return MM_OK;
}

// Shared by both enqueue entry points; all pointers are device pointers.
static int enqueue_device_impl(mm_engine* e, uint32_t n, const int32_t* d_rating, const uint32_t* d_cons,
                               const uint8_t* d_group, const uint32_t* d_slot_sel, uint32_t* d_out_slot,
                               uint32_t* rejected, float* bucket_ms, uint32_t* h_out_slot = nullptr)
{
    const uint32_t blocks = (n + BK_CHUNK - 1) / BK_CHUNK;
    const size_t rows = (size_t)blocks * BK_WAVES;
    if (rows > e->wave_hist_rows) {
        (void)hipFree(e->d_wave_hist);
        e->d_wave_hist = NULL;
        e->wave_hist_rows = 0;
        HIPCHK(e, hipMalloc((void**)&e->d_wave_hist, rows * e->n_chains * sizeof(uint32_t)));
        e->wave_hist_rows = rows;
    }
    const BucketCfg B = make_bucket_cfg(e->cfg);
    const bool timing = (e->cfg.flags & MM_CFG_TIMING) != 0;
    HIPCHK(e, hipMemsetAsync(e->d_counters + 1, 0, sizeof(uint32_t), e->stream));
    if (timing) HIPCHK(e, hipEventRecord(e->ev[0], e->stream));
    hipLaunchKernelGGL(k_bucket_count, dim3(blocks), dim3(BK_THREADS), 0, e->stream, n, d_rating, d_cons, d_group, B,
                       e->n_chains, e->d_wave_hist, e->d_counters + 1);
    hipLaunchKernelGGL(k_bucket_scan, dim3(e->n_chains), dim3(BK_THREADS), 0, e->stream, (uint32_t)rows, e->n_chains,
                       e->d_wave_hist, e->d_chains, e->cfg.capacity);
    hipLaunchKernelGGL(k_bucket_scatter, dim3(blocks), dim3(BK_THREADS), 0, e->stream, n, d_rating, d_cons, d_group, B,
                       e->n_chains, e->d_wave_hist, e->next_slot, d_slot_sel, e->d_q_rating, e->d_q_cons, e->d_q_slot,
                       e->d_state, d_out_slot);
    HIPCHK(e, hipGetLastError());
    if (timing) HIPCHK(e, hipEventRecord(e->ev[1], e->stream));
    HIPCHK(e, hipMemcpyAsync(e->h_counters + 1, e->d_counters + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    // the handles travel with the counter: one round trip to the device per enqueue, not two (a stream enqueues every tick)
    if (h_out_slot) HIPCHK(e, hipMemcpyAsync(h_out_slot, d_out_slot, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    *rejected = e->h_counters[1];
    *bucket_ms = 0.f;
    if (timing) HIPCHK(e, hipEventElapsedTime(bucket_ms, e->ev[0], e->ev[1]));
    return MM_OK;
}

static int ring_range_free(const mm_engine* e, uint32_t n)
{
    const uint32_t cap = e->cfg.capacity;
    if (n > cap) return 0;
    if (e->live_upper == 0) return 1;           // nobody queued, seated or cancelled-and-not-yet-purged: every slot is free
    const uint32_t a = e->next_slot;
    const uint32_t n1 = n < cap - a ? n : cap - a;
    if (n1 && memchr(&e->h_state[a], MM_ST_LIVE, n1)) return 0;
    if (n1 && memchr(&e->h_state[a], MM_ST_CANCELLED, n1)) return 0;
    if (n > n1) {
        if (memchr(&e->h_state[0], MM_ST_LIVE, n - n1)) return 0;
        if (memchr(&e->h_state[0], MM_ST_CANCELLED, n - n1)) return 0;
    }
    return 1;
}

// Slot allocation of mm_enqueue: the next n FREE slots in ring order from next_slot, stepping over
// slots whose player is still waiting (a starving anchor may keep its slot for hours while the
// ring wraps many times).  While nobody is in the way this is the plain range next_slot..+n-1
// (`*contiguous`, no list needed).  0 = fewer than n free slots in the whole pool.
static int pick_free_slots(const mm_engine* e, uint32_t n, std::vector<uint32_t>& sel, bool* contiguous)
{
    const uint32_t cap = e->cfg.capacity;
    *contiguous = false;
    if (n > cap) return 0;
    if (ring_range_free(e, n)) { *contiguous = true; return 1; }
    sel.clear();
    sel.reserve(n);
    uint32_t s = e->next_slot;
    for (uint32_t seen = 0; seen < cap && sel.size() < n; ++seen) {
        if (e->h_state[s] == MM_ST_FREE) sel.push_back(s);
        s = s + 1u == cap ? 0u : s + 1u;
    }
    return sel.size() == n;
}

extern "C" int mm_enqueue(mm_engine* e, uint32_t n, const int32_t* rating, const uint32_t* cons,
                          const uint8_t* group, uint32_t* out_slot, mm_enqueue_stats* st)
{
    try {
        if (!e || (n && (!rating || !cons))) return MM_ERR_INVALID_ARG;
        if (e->poisoned) return MM_ERR_STATE;
        ON_ENGINE_DEVICE(e);
        RoctxRange rr("mm_enqueue");
        const double t0 = host_now_ms();
        if (st) memset(st, 0, sizeof(*st));
        if (n == 0) return MM_OK;
        std::vector<uint32_t> sel;
        bool contiguous = false;
        if (!pick_free_slots(e, n, sel, &contiguous)) return MM_ERR_FULL;
        if (group)
            for (uint32_t i = 0; i < n; ++i)
                if (group[i] >= e->cfg.n_groups) return MM_ERR_INVALID_ARG;
        int rc = ensure_staging(e, n);
        if (rc) return rc;
        if (!contiguous)
            HIPCHK(e, hipMemcpyAsync(e->d_in_sel, sel.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        HIPCHK(e, hipMemcpyAsync(e->d_in_rating, rating, n * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
        HIPCHK(e, hipMemcpyAsync(e->d_in_cons, cons, n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        if (group) HIPCHK(e, hipMemcpyAsync(e->d_in_group, group, n, hipMemcpyHostToDevice, e->stream));
        uint32_t rejected = 0;
        float bms = 0.f;
        std::vector<uint32_t> tmp;
        uint32_t* slots = out_slot;
        if (!slots) { tmp.resize(n); slots = tmp.data(); }
        rc = enqueue_device_impl(e, n, e->d_in_rating, e->d_in_cons, group ? e->d_in_group : NULL,
                                 contiguous ? NULL : e->d_in_sel, e->d_in_slot, &rejected, &bms, slots);   // syncs: `sel` outlives the copy
        if (rc) return rc;
        for (uint32_t i = 0; i < n; ++i)
            if (slots[i] != MM_NO_SLOT) e->h_state[slots[i]] = MM_ST_LIVE;
        e->next_slot = contiguous ? (uint32_t)(((unsigned long long)e->next_slot + n) % e->cfg.capacity)
                                  : (sel[n - 1] + 1u) % e->cfg.capacity;
        e->live_upper += n - rejected;
        if (st) {
            st->accepted = n - rejected;
            st->rejected = rejected;
            st->bucket_ms = bms;
            st->total_ms = (float)(host_now_ms() - t0);
        }
        return MM_OK;
    } catch (const std::bad_alloc&) {
        return MM_ERR_OOM;
    } catch (...) {
        return MM_ERR_INTERNAL;
    }
}

extern "C" int mm_enqueue_device(mm_engine* e, uint32_t n, const int32_t* d_rating, const uint32_t* d_cons,
                                 uint32_t* first_slot, mm_enqueue_stats* st)
{
    try {
        if (!e || (n && (!d_rating || !d_cons))) return MM_ERR_INVALID_ARG;
        if (e->poisoned) return MM_ERR_STATE;
        ON_ENGINE_DEVICE(e);
        const double t0 = host_now_ms();
        if (st) memset(st, 0, sizeof(*st));
        if (first_slot) *first_slot = e->next_slot;
        if (n == 0) return MM_OK;
        if (!ring_range_free(e, n)) return MM_ERR_FULL;
        uint32_t rejected = 0;
        float bms = 0.f;
        int rc = ensure_staging(e, n);                       // the slot column (d_in_slot) of the batch
        if (rc) return rc;
        rc = enqueue_device_impl(e, n, d_rating, d_cons, NULL, NULL, e->d_in_slot, &rejected, &bms);
        if (rc) return rc;
        const uint32_t cap = e->cfg.capacity;
        const uint32_t a = e->next_slot, n1 = n < cap - a ? n : cap - a;
        if (rejected == 0u) {
            memset(&e->h_state[a], MM_ST_LIVE, n1);
            if (n > n1) memset(&e->h_state[0], MM_ST_LIVE, n - n1);
        } else {
            // some players were refused on the device (mode not configured, role not seatable): only the
            // accepted ones hold their slot; a refused player's slot stays FREE (it is in no queue and no
            // lobby, nothing would ever release it)
            std::vector<uint32_t> got(n);
            HIPCHK(e, hipMemcpy(got.data(), e->d_in_slot, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < n; ++i)
                if (got[i] != MM_NO_SLOT) e->h_state[got[i]] = MM_ST_LIVE;
        }
        e->next_slot = (uint32_t)(((unsigned long long)a + n) % cap);
        e->live_upper += n - rejected;
        if (st) {
            st->accepted = n - rejected;
            st->rejected = rejected;
            st->bucket_ms = bms;
            st->total_ms = (float)(host_now_ms() - t0);
        }
        return MM_OK;
    } catch (const std::bad_alloc&) {
        return MM_ERR_OOM;
    } catch (...) {
        return MM_ERR_INTERNAL;
    }
}

extern "C" int mm_cancel(mm_engine* e, uint32_t n, const uint32_t* slot)
{
    try {
        if (!e || (n && !slot)) return MM_ERR_INVALID_ARG;
        if (e->poisoned) return MM_ERR_STATE;
        ON_ENGINE_DEVICE(e);
        std::vector<uint32_t> live;
        live.reserve(n);
        for (uint32_t i = 0; i < n; ++i) {
            if (slot[i] >= e->cfg.capacity) continue;
            if (e->h_state[slot[i]] == MM_ST_LIVE) {
                e->h_state[slot[i]] = MM_ST_CANCELLED;
                live.push_back(slot[i]);
            }
        }
        if (live.empty()) return MM_OK;
        int rc = ensure_staging(e, live.size());
        if (rc) return rc;
        const uint32_t k = (uint32_t)live.size();
        HIPCHK(e, hipMemcpyAsync(e->d_in_slot, live.data(), k * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(k_cancel, dim3((k + 255) / 256), dim3(256), 0, e->stream, k, e->d_in_slot, e->d_state);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipStreamSynchronize(e->stream));   // `live` must outlive the copy
        e->cancel_pending += k;
        return MM_OK;
    } catch (const std::bad_alloc&) {
        return MM_ERR_OOM;
    } catch (...) {
        return MM_ERR_INTERNAL;
    }
}

// ---- the match list on its way to the host while the walk runs --------------------------------------------------
#ifndef MM_RESULTS_MIN_PAIR
#define MM_RESULTS_MIN_PAIR 16384u   // new lobbies at a look at the chains that are worth a batch of copies (the logic tests build with less)
#endif
#ifndef MM_RESULTS_MIN_TEAM
#define MM_RESULTS_MIN_TEAM 4096u
#endif
// r_base: where group g's lobbies live in h_r*.  `before[g]`: the players the group holds as the tick begins (queue +
// stored lobby + cancelled entries): it cannot emit more than before / L lobbies.
static void results_set_bases(mm_engine* e, const uint32_t* before, uint32_t L)
{
    uint32_t run = 0;
    for (uint32_t g = 0; g < e->cfg.n_groups; ++g) {
        e->r_base[g] = run;
        run += before[g] / L + 1u;
    }
    e->r_based = true;
}

// the players of lobbies [from, to) of group g are matched: their slots are FREE again (ActiveUser.remove_user for
// the matched players, game-lobby/worker.ex:73-103)
static void results_mark(mm_engine* e, uint32_t g, uint32_t to, uint32_t L)
{
    const uint32_t from = e->r_marked[g];
    if (to <= from) return;
    const uint32_t* const sl = e->h_rslots + (size_t)(e->r_base[g] + from) * L;
    for (size_t i = 0, n = (size_t)(to - from) * L; i < n; ++i) e->h_state[sl[i]] = MM_ST_FREE;
    e->r_marked[g] = to;
}

// what arrived with the last batch of copies: release the slots (host work while the device walks on)
static int results_absorb(mm_engine* e, uint32_t L)
{
    if (!e->ev_copy_pending) return MM_OK;
    HIPCHK(e, hipEventSynchronize(e->ev_copy));
    e->ev_copy_pending = false;
    for (uint32_t g = 0; g < e->cfg.n_groups; ++g) results_mark(e, g, e->r_sent[g], L);
    return MM_OK;
}

// lobbies [r_sent[g], n_out[g]) of every group go out on the copy stream.  The emission lists only grow and every
// kernel that wrote entries below n_out has finished (the caller has just synchronised the engine stream).
static int results_send(mm_engine* e, const uint32_t* n_out, uint32_t L, uint32_t min_new, bool* on_main = nullptr)
{
    uint32_t fresh = 0;
    for (uint32_t g = 0; g < e->cfg.n_groups; ++g) fresh += n_out[g] > e->r_sent[g] ? n_out[g] - e->r_sent[g] : 0u;
    if (on_main) *on_main = false;
    if (fresh < min_new) return MM_OK;
    int rc = results_absorb(e, L);                       // one batch of copies in flight at a time
    if (rc) return rc;
    // the end of a tick with little left to send (every tick of a stream): on the engine's own stream, behind the
    // walk — the caller synchronises that stream anyway, and a second stream's event would be one more round trip
    if (on_main != nullptr && fresh <= MM_TAIL_MAX && e->results_tail_kernel) {
        TailArgs A;
        memset(&A, 0, sizeof(A));
        A.n_groups = e->cfg.n_groups; A.L = L; A.out_slot_stride = e->out_slot_stride; A.out_rec_stride = e->out_rec_stride;
        for (uint32_t g = 0; g < e->cfg.n_groups; ++g) {
            const uint32_t a = e->r_sent[g], b = n_out[g] > a ? n_out[g] : a;
            A.from[g] = a; A.base[g] = e->r_base[g]; A.pre[g + 1u] = A.pre[g] + (b - a);
            e->r_sent[g] = b;
        }
        A.total = A.pre[e->cfg.n_groups];
        if (A.total) {
            hipLaunchKernelGGL(k_results_tail, dim3((A.total + 255u) / 256u), dim3(256), 0, e->stream, A, e->d_out_slots, e->d_out_score,
                               e->d_out_pass, e->h_rslots, e->h_rscore, e->h_rpass);
            HIPCHK(e, hipGetLastError());
        }
        *on_main = true;                                 // the caller marks the slots after its own synchronisation
        return MM_OK;
    }
    const bool main_st = on_main != nullptr && fresh <= 8192u;
    hipStream_t const st = main_st ? e->stream : e->copy_stream;
    for (uint32_t g = 0; g < e->cfg.n_groups; ++g) {
        const uint32_t a = e->r_sent[g], b = n_out[g];
        if (b <= a) continue;
        const size_t at = (size_t)e->r_base[g] + a;
        HIPCHK(e, hipMemcpyAsync(&e->h_rslots[at * L], e->d_out_slots + (size_t)g * e->out_slot_stride + (size_t)a * L,
                                 (size_t)(b - a) * L * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(e, hipMemcpyAsync(&e->h_rscore[at], e->d_out_score + (size_t)g * e->out_rec_stride + a, (size_t)(b - a) * sizeof(float),
                                 hipMemcpyDeviceToHost, st));
        HIPCHK(e, hipMemcpyAsync(&e->h_rpass[at], e->d_out_pass + (size_t)g * e->out_rec_stride + a, (size_t)(b - a) * sizeof(uint32_t),
                                 hipMemcpyDeviceToHost, st));
        e->r_sent[g] = b;
    }
    if (main_st) { *on_main = true; return MM_OK; }      // the caller marks the slots after its own synchronisation
    HIPCHK(e, hipEventRecord(e->ev_copy, e->copy_stream));
    e->ev_copy_pending = true;
    return MM_OK;
}

// A look at device records through the pinned buffer (k_look): look_launch enqueues the hand-over behind whatever the
// stream holds, look_wait returns when the records of THAT look are in `dst_host`.  The poll is bounded by the stream
// itself: a stream that has finished (or failed) without the sequence number having arrived ends it.
static int look_launch(mm_engine* e, const void* d_src, void* h_dst, size_t bytes)
{
    if (!e->look_poll) {
        HIPCHK(e, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, e->stream));
        return MM_OK;
    }
    if (++e->look_seq == 0u) e->look_seq = 1u;
    hipLaunchKernelGGL(k_look, dim3(1), dim3(256), 0, e->stream, (const uint32_t*)d_src, (uint32_t*)h_dst, (uint32_t)(bytes / 4u),
                       e->h_look_seq, e->look_seq);
    HIPCHK(e, hipGetLastError());
    return MM_OK;
}
static int look_wait(mm_engine* e)
{
    if (!e->look_poll) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        return MM_OK;
    }
    const uint32_t want = e->look_seq;
    for (uint32_t spins = 0;; ++spins) {
        if (__atomic_load_n(e->h_look_seq, __ATOMIC_ACQUIRE) == want) return MM_OK;
        if ((spins & 0x3FFFu) == 0x3FFFu) {
            // (every 16k polls, a few hundred microseconds: is the stream still at work?)
            const hipError_t q = hipStreamQuery(e->stream);
            if (q == hipSuccess) {
                // everything enqueued has run: the number is there now, or the launch was lost
                if (__atomic_load_n(e->h_look_seq, __ATOMIC_ACQUIRE) == want) return MM_OK;
                e->last_hip = (int)hipErrorUnknown;
                return MM_ERR_HIP;
            }
            if (q != hipErrorNotReady) { e->last_hip = (int)q; return MM_ERR_HIP; }
        }
        __builtin_ia32_pause();
    }
}

// kp_round's workgroup map of a batch (PairParams.xseg): the tiles of a chain on as few XCDs as its tile count allows.
// Up to eight chains: an XCD each, the XCDs that are left go one by one to the chain with the most tiles per XCD;
// more chains than XCDs: longest first onto the emptiest XCD.  Returns the slots per XCD (0: no map, the plain grid).
static uint32_t pair_xcd_map(PairParams& P, const uint32_t* tiles_of, uint32_t G, bool one_xcd = false)
{
    memset(P.xseg, 0, sizeof(P.xseg));
    memset(P.xcnt, 0, sizeof(P.xcnt));
    P.xslots = 0;
    uint32_t order[MM_MAX_GROUPS], n = 0;
    for (uint32_t g = 0; g < G; ++g)
        if (tiles_of[g]) order[n++] = g;
    if (!n) return 0;
    std::sort(order, order + n, [&](uint32_t a, uint32_t b) { return tiles_of[a] != tiles_of[b] ? tiles_of[a] > tiles_of[b] : a < b; });
    uint32_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nseg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (n <= 8u) {
        uint32_t k[MM_MAX_GROUPS];
        for (uint32_t i = 0; i < n; ++i) k[i] = 1;
        for (uint32_t spare = one_xcd ? 0u : 8u - n; spare; --spare) {     // (kp_rounds: a chain's workgroups talk through ONE L2)
            uint32_t best = 0;
            for (uint32_t i = 1; i < n; ++i)
                if ((unsigned long long)tiles_of[order[i]] * k[best] > (unsigned long long)tiles_of[order[best]] * k[i]) best = i;
            if (tiles_of[order[best]] <= k[best]) break;            // a tile per XCD already
            ++k[best];
        }
        uint32_t x = 0;
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t j = 0; j < k[i]; ++j, ++x) {
                const uint32_t g = order[i], cnt = (tiles_of[g] - j + k[i] - 1u) / k[i];
                if (k[i] > 255u || cnt > 0xFFFFu) return 0;
                P.xseg[x][0] = g | (j << 8) | (k[i] << 16);
                P.xcnt[x][0] = (uint16_t)cnt;
                load[x] = cnt;
            }
    } else {
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t x = 0;
            for (uint32_t y = 1; y < 8u; ++y)
                if (load[y] < load[x]) x = y;
            const uint32_t g = order[i];
            if (nseg[x] >= PK_XSEG || tiles_of[g] > 0xFFFFu) return 0;
            P.xseg[x][nseg[x]] = g | (0u << 8) | (1u << 16);
            P.xcnt[x][nseg[x]] = (uint16_t)tiles_of[g];
            ++nseg[x];
            load[x] += tiles_of[g];
        }
    }
    uint32_t slots = 0;
    for (uint32_t x = 0; x < 8u; ++x) slots = load[x] > slots ? load[x] : slots;
    // a workgroup per CU (32 CUs an XCD): with more, the XCD that holds a long chain alone would run its workgroups in
    // several waves while others idle — big pools stay on the plain grid, which spreads every chain over all XCDs
    if (slots > 32u) { memset(P.xseg, 0, sizeof(P.xseg)); memset(P.xcnt, 0, sizeof(P.xcnt)); return 0; }
    P.xslots = slots;
    return slots;
}

// The pair path (mm_pair.inc) for every chain of `mode` it is eligible for; the others are
// left to k_walk (PairChain.fast == 0).  All launches are asynchronous on the engine stream.
static int pair_walk(mm_engine* e, uint32_t mode, const ModeDev& M, bool purge)
{
    const mm_config& cfg = e->cfg;
    const uint32_t G = cfg.n_groups;
    PairParams P;
    memset(&P, 0, sizeof(P));
    P.mode = mode;
    P.n_groups = G;
    P.capacity = cfg.capacity;
    P.window = M.window > 0xFFFFFu ? 0xFFFFFu : M.window;   // rating span of a fast chain < 2^20
    P.eqmask = M.eqmask;
    P.out_slot_stride = e->out_slot_stride;
    P.out_rec_stride = e->out_rec_stride;
    P.bits_stride = e->pk_bits_stride;
    P.pstride = e->pk_stride;
    P.tune = e->pair_tune;
    P.purge = purge ? 1u : 0u;
    P.state = e->d_state;
    P.released = e->d_released;
    P.n_released = e->d_counters;
    P.chains = e->d_chains;
    P.pchains = e->d_pchains;
    P.q_rating = e->d_q_rating;
    P.q_cons = e->d_q_cons;
    P.q_slot = e->d_q_slot;
    for (int b = 0; b < 2; ++b) {
        P.key[b] = e->d_pk_key[b]; P.oidx[b] = e->d_pk_oidx[b];
        P.nx16[b] = e->d_pk_nx16[b]; P.bits[b] = e->d_pk_bits[b];
    }
    P.scratch = e->d_pk_scratch;
    P.wpre = e->d_pk_wpre;
    P.g16 = e->d_pk_g16;
    P.rec2[0] = e->d_pk_scratch;
    P.rec2[1] = e->d_pk_rec1;
    for (int b = 0; b < 2; ++b) { P.exa[b] = e->d_pk_exa[b]; P.bitsp[b] = e->d_pk_bitsp[b]; P.headp[b] = e->d_pk_headp[b]; }
    P.tilectl = e->d_pk_tilectl;
    P.max_tiles = e->pk_max_tiles;
    P.grec = e->d_pk_grec;
    P.gstride = e->pk_gstride;
    P.grp = 0;
    P.pbar = e->d_pk_pbar;
    P.pxmask = (uint32_t*)(e->d_pk_pbar + MM_MAX_GROUPS);
    P.ptimeout0 = e->pair_ptimeout[0];
    P.ptimeout1 = e->pair_ptimeout[1];
    P.pinject = e->pair_pinject;
    P.pyield = (uint32_t*)(e->d_pk_pbar + MM_MAX_GROUPS) + MM_MAX_GROUPS;
    P.out_slots = e->d_out_slots;
    P.out_score = e->d_out_score;
    P.out_pass = e->d_out_pass;
    const unsigned long long bound64 = e->live_upper < cfg.capacity ? e->live_upper : cfg.capacity;
    const uint32_t bound = (uint32_t)bound64;
    if (e->pair_tune & 0x10000u) HIPCHK(e, hipMemsetAsync(e->d_pk_grec, 0, (size_t)G * e->pk_gstride * sizeof(uint4), e->stream));   // (diagnostics)
    hipLaunchKernelGGL(kp_init, dim3(G), dim3(1024), 0, e->stream, P, cfg.capacity);
    hipLaunchKernelGGL(kp_pack, dim3((bound + 32u + 1023u) / 1024u, G), dim3(1024), 0, e->stream, P);
    {
        // A workgroup walks its anchors eight at a time (one wave each).  256 anchors per workgroup (round 5; 2048 until then,
        // "to keep a big pool's staging traffic low"): the kernel is bound by instruction issue, the anchors of the wide rating
        // groups cost 2-3x those of the narrow ones, and 147 + 98 heavy workgroups of 2048 anchors landed four to a CU —
        // 299 us for the 1M pool against 148 us with 256 anchors a workgroup (157 / 173 us with 512 / 1024;
        // profiles/r05_ab_nx_seg.txt).  The staged window (NXI_STAGE entries from L2) is read 8192 / 256 times per entry: 130 MB.
        // (a 10M pool keeps its 2048: the grid is sized by the POOL for every rating group — the host does not know the chains'
        // lengths yet — and a quarter of a million workgroups that only find out they have nothing to do are not free)
        uint32_t seg = 256u, stage = NXI_STAGE;
        while (seg < NXI_SEG && (unsigned long long)bound > 8192ull * seg) seg <<= 1;
        if (e->pair_nxseg) seg = e->pair_nxseg;               // MM_PAIR_NXSEG / MM_PAIR_NXSTAGE (experiments)
        if (e->pair_nxstage) stage = e->pair_nxstage;
        if (stage < seg + 64u) stage = seg + 64u;
        if (stage > NXI_STAGE) stage = NXI_STAGE;
        const bool timing = (cfg.flags & MM_CFG_TIMING) != 0;
        if (timing) HIPCHK(e, hipEventRecord(e->ev_nx[0], e->stream));
        hipLaunchKernelGGL(kp_nx_init, dim3((bound + seg - 1u) / seg + 1u, G), dim3(NXI_THREADS), 0, e->stream, P, seg, stage);
        if (timing) { HIPCHK(e, hipEventRecord(e->ev_nx[1], e->stream)); e->ev_nx_set = true; }
    }
    HIPCHK(e, hipGetLastError());
    uint32_t tail_no[MM_MAX_GROUPS];
    bool tail_send = false;
    memset(e->ps_hand, 0, sizeof(e->ps_hand));
    // ---- tiled rounds for the chains that do not fit one workgroup's LDS ----
    if (bound >= PL_MAX) {
        uint32_t tiles = (bound + PK_T - 1u) / PK_T + 1u;
        if (tiles > e->pk_max_tiles) tiles = e->pk_max_tiles;
        bool after_persist = false;            // the look directly behind a kp_rounds launch: only there is PairChain.pfail news
        uint32_t last_tp = 0;                  // the tile length of the tick's last batch (it never grows: below)
        uint32_t idle_looks = 0;               // looks in a row at which no tiled chain had retired a pass or left the tiled stage
        unsigned long long last_progress = ~0ull;
        for (uint32_t guard = 0;; ++guard) {
            // A pass without a change ends a chain, so a tick is finite; an iteration (a look + a batch) retires at least one pass
            // of every tiled chain, or is one of the few kinds that retire none and are each followed by one that does (a
            // compaction-only look; a kp_rounds launch that stopped at its first barrier, after which kp_rounds stays off for
            // 16 batches).  The watchdog counts the looks in a row at which nothing had moved.
            // counted in looks that bound was 3 x capacity round trips (tens of minutes at 2^24 before MM_ERR_INTERNAL,
            // and 3u * capacity wraps above 1.43e9).  Counted in PASSES it is tight: see idle_looks below.
            if (idle_looks > 64u) return MM_ERR_INTERNAL;
            { int lrc = look_launch(e, e->d_pchains, e->h_pchains, G * sizeof(PairChain)); if (lrc) return lrc; }
            { int lrc = look_wait(e); if (lrc) return lrc; }
            ++e->ps.host_looks;
            if (e->pair_debug && guard >= 200u && guard % 200u == 0u) {           // (a tick that looks at its chains hundreds of times)
                fprintf(stderr, "[mm-pair] look %u: cool-down %u, persist %d, last tile length %u;", guard, e->pair_pcool, (int)e->pair_persist, last_tp);
                for (uint32_t g = 0; g < G; ++g) {
                    const PairChain& pc = e->h_pchains[g];
                    if (pc.fast) fprintf(stderr, " g%u[stage %u m %u qlen %u passes %u compact %u pfail 0x%x]", g, pc.stage, pc.m, pc.qlen, pc.passes, pc.want_compact, pc.pfail);
                }
                fprintf(stderr, "\n");
            }
            {   // the watchdog: passes retired + chains still tiled + compactions done must move; the looks that move nothing (a
                // compaction-only look, a kp_rounds launch that stopped at its first barrier) are each followed by one that does
                unsigned long long progress = 0;
                for (uint32_t g = 0; g < G; ++g) {
                    const PairChain& pc = e->h_pchains[g];
                    if (pc.fast) progress += (unsigned long long)pc.passes + ((unsigned long long)pc.m << 32) + (pc.stage == PS_TILED ? 1u : 0u) + pc.want_compact;
                }
                idle_looks = progress == last_progress ? idle_looks + 1u : 0u;
                last_progress = progress;
            }
            bool tiled = false, compact = false;
            uint32_t longest = 0;
            for (uint32_t g = 0; g < G; ++g) {
                const PairChain& pc = e->h_pchains[g];
                P.bm[g] = 0;
                P.bbuf[g] = 0;
                if (!pc.fast || pc.stage != PS_TILED) continue;
                // (pfail is written by kp_init and by kp_rounds' epilogue only and stays in the record: it is looked at once,
                // at the look directly behind the launch that wrote it — a later look of the tick would count the same stop again
                // and start the cool-down anew)
                if (after_persist && pc.pfail && (pc.pfail & 0xFFu) == PF_YIELD) ++e->ps.pair_yields;
                if (after_persist && pc.pfail && (pc.pfail & 0xFFu) != PF_YIELD) {
                    // the last kp_rounds launch gave up on this chain (its state is committed): one launch per pass for a while.
                    // A chain that found itself on two XCDs says the dispatch is not what the map assumes: never again.
                    ++e->pair_pstops;
                    e->pair_pcool = 16u;
                    const uint32_t why = pc.pfail & 0xFFu;
                    if (why == PF_XCD) { e->pair_persist = false; ++e->ps.pair_stops_xcd; }
                    else if (why == PF_INJECT) ++e->ps.pair_stops_inject;
                    else ++e->ps.pair_stops_timeout;
                    e->ps.degraded = 1;
                    if (e->pair_debug) fprintf(stderr, "[mm-pair] g%u: kp_rounds stopped (reason %u, iteration %u)\n", g, pc.pfail & 0xFFu, pc.pfail >> 8);
                }
                P.bm[g] = pc.m;                       // constant until the next compaction, i.e. for the whole batch
                P.bbuf[g] = (uint8_t)pc.buf;
                tiled = true;
                compact |= pc.want_compact != 0;
                longest = pc.m > longest ? pc.m : longest;
            }
            // the lobbies emitted so far leave for the host while the next batch runs (the first look sizes the groups'
            // regions); the copies are enqueued BEHIND the batch's launches: the device does not wait for the host's calls
            uint32_t sent_no[MM_MAX_GROUPS];
            {
                uint32_t bf[MM_MAX_GROUPS];
                for (uint32_t g = 0; g < G; ++g) { bf[g] = e->h_pchains[g].before; sent_no[g] = e->h_pchains[g].fast ? e->h_pchains[g].n_out : 0u; }
                if (!e->r_based) results_set_bases(e, bf, M.L);
            }
            after_persist = false;
            for (uint32_t g = 0; g < G; ++g) {
                const PairChain& pc = e->h_pchains[g];
                if (!pc.fast) continue;
                if (pc.ppass > e->ps.pair_rounds_passes) e->ps.pair_rounds_passes = pc.ppass;
                if (pc.rounds > e->ps.pair_tiled_passes) e->ps.pair_tiled_passes = pc.rounds;
            }
            for (uint32_t g = 0; g < G; ++g) {         // (what the tiled path had done when kp_late took over: the last look's is kept)
                const PairChain& pc = e->h_pchains[g];
                e->ps_hand[0][g] = pc.fast ? pc.passes : 0u;
                e->ps_hand[1][g] = pc.fast ? pc.n_out : 0u;
                e->ps_hand[2][g] = pc.fast ? pc.ppass : 0u;
                e->ps_hand[3][g] = pc.fast ? pc.ptm[15] : 0u;
            }
            if (!tiled) {
                // the chains' LDS-resident ends (kp_late) run next: what the tiled rounds have emitted since the last look leaves
                // meanwhile — enqueued BEHIND kp_late's launch, below (the device never waits for the host's copy calls)
                if (guard) { for (uint32_t g = 0; g < G; ++g) tail_no[g] = sent_no[g]; tail_send = true; }
                break;
            }
            // kp_rounds wants a chain within `pair_ptiles` tiles, and a pass costs in proportion to the tile length: a chain whose
            // QUEUED players would fit the tiles of a shorter length (or kp_rounds at all) while its index space does not is
            // compacted now rather than at 75 % alive.  Only the longest chain decides the tile length.
            bool want_fit = false;
            uint32_t yield_q = 0, yield_g = 0;
            if (e->pair_persist && !e->pair_pcool && !compact && !e->pair_tile_fixed) {
                uint32_t gl = 0;
                for (uint32_t g = 1; g < G; ++g)
                    if (P.bm[g] > P.bm[gl]) gl = g;
                const PairChain& pc = e->h_pchains[gl];
                if (P.bm[gl]) {
                    // the capacity the chain's index space has to get under for the next shorter tile length (or for kp_rounds)
                    uint32_t capq = e->pair_ptiles * PK_TMAX;
                    for (uint32_t tlen = PK_TMAX; tlen >= PK_TMAX / 4u && pc.m <= e->pair_ptiles * tlen; tlen >>= 1) capq = e->pair_ptiles * (tlen >> 1);
                    if (pc.m <= e->pair_ptiles * (PK_TMAX / 4u)) capq = PL_MAX - 1u;          // shortest tiles already: next is kp_late
                    if (pc.qlen <= capq && pc.m > capq && capq >= PL_MAX) {
                        hipLaunchKernelGGL(kp_ask_compact, dim3(G), dim3(64), 0, e->stream, P, 1u << gl);
                        compact = true;
                    } else {
                        if (pc.m > e->pair_ptiles * PK_TMAX) want_fit = true;          // not yet within kp_rounds' reach: a short batch
                        yield_q = capq;                                                // compact (and, in kp_rounds, end the batch) at this length
                        yield_g = gl;
                    }
                }
            }
            // tile length of this batch: the smallest that keeps the longest chain within PK_TILES_MAX tiles (the
            // walk costs one dependent load per tile, everything else is proportional to the tile)
            uint32_t tp = PK_TMAX;
            // several passes per launch (kp_rounds) when every chain's tiles fit the CUs of one XCD
            bool persist = e->pair_persist && !compact && (longest + PK_TMAX - 1u) / PK_TMAX <= e->pair_ptiles;
            if (persist && e->pair_pcool) { --e->pair_pcool; persist = false; e->ps.degraded = 1; }
            // (a batch that kp_rounds would have taken, walked launch by launch because it is off: a fall-back's timing)
            if (!e->pair_persist && !compact && (longest + PK_TMAX - 1u) / PK_TMAX <= e->pair_ptiles) e->ps.degraded = 1;
            // ONE cap for the tile count whether this batch is kp_rounds' or kp_round's (round 5): the tile length of a tick must
            // never GROW.  An entry next[i] = NX_FAR says "nobody fits inside the horizon", and the horizon is two tiles of the
            // length it was computed at; the walk resolves it by scanning from the end of the CURRENT horizon — right for an
            // equal or shorter tile length (it scans a superset), wrong for a longer one (the stretch between the two horizons
            // is never looked at: a later partner, or none).  Until round 5 a kp_round batch was sized for 40 tiles and a
            // kp_rounds batch for 32: a chain of 33-40 tiles of length T walked launch by launch — the cool-down after a stop of
            // kp_rounds, or while ANOTHER chain was still too long for kp_rounds — came back to kp_rounds at 2 T.  Found by
            // tests/stress.py with MM_PAIR_PTILES=5 (seed 130203984: 46 lobbies of one chain missing), on the device and on
            // the shim; it needs a stop of kp_rounds (or the knob) to happen, which is why no default run ever met it.
            const uint32_t cap_p = e->pair_ptiles < e->pair_tiles_max ? e->pair_ptiles : e->pair_tiles_max;
            const uint32_t tiles_max = e->pair_persist ? cap_p : e->pair_tiles_max;
            if (!e->pair_tile_fixed)
                for (uint32_t cand = PK_TMAX / 4u; cand < PK_TMAX; cand <<= 1)    // (an eighth was measured: slower, the fixed cost of a round takes over)
                    if ((longest + cand - 1u) / cand <= tiles_max) { tp = cand; break; }
            if (last_tp && tp > last_tp && !compact) {
                // (the cap changed in mid-tick — kp_rounds turned off for good by a PF_XCD stop — or a knob: every tiled chain is
                // compacted first, which turns its NX_FAR entries into "not computed")
                hipLaunchKernelGGL(kp_ask_compact, dim3(G), dim3(64), 0, e->stream, P, 0xFFFFFFFFu);
                compact = true;
                last_tp = 0;
            }
            tiles = (longest + tp - 1u) / tp;
#define TILE_LAUNCH(KERNEL, GRID, BLOCK, ...)                                                                              \
    do {                                                                                                                   \
        if (tp == PK_TMAX) hipLaunchKernelGGL(KERNEL<PK_TMAX>, GRID, BLOCK, 0, e->stream, __VA_ARGS__);                    \
        else if (tp == PK_TMAX / 2u) hipLaunchKernelGGL(KERNEL<PK_TMAX / 2u>, GRID, BLOCK, 0, e->stream, __VA_ARGS__);     \
        else hipLaunchKernelGGL(KERNEL<PK_TMAX / 4u>, GRID, BLOCK, 0, e->stream, __VA_ARGS__);                             \
    } while (0)
            // the four compaction kernels look at PairChain.want_compact themselves (a chain that does not want one costs them a few
            // microseconds): they are enqueued behind EVERY batch, so a chain that ends its batch for a compaction — the 75 % rule,
            // the yield, the fit — is compacted before the host looks again, one look per batch instead of two
#define COMPACT_LAUNCH()                                                                                                   \
    do {                                                                                                                   \
        TILE_LAUNCH(kc_words, dim3(tiles, G), dim3(256), P);                                                               \
        TILE_LAUNCH(kc_plan, dim3(G), dim3(1024), P);                                                                      \
        TILE_LAUNCH(kc_scatter, dim3(tiles, G), dim3(1024), P);                                                            \
        TILE_LAUNCH(kc_commit, dim3(G), dim3(1024), P);                                                                    \
    } while (0)
            if (compact) {
                COMPACT_LAUNCH();
                HIPCHK(e, hipGetLastError());
                continue;                       // look at the new lengths before the next batch
            }
            last_tp = tp;
            for (uint32_t g = 0; g < G; ++g) P.pyq[g] = 0;
            P.pyq[yield_g] = yield_q;
            {
                // one launch per pass; the first of a batch only prepares, the commit brings the
                // latest parity back into the chains' committed state
                // chains of many tiles: a second level of the route, rebuilt behind every round (kp_group)
                P.grp = (e->pair_group_min && tiles >= e->pair_group_min && tp == PK_TMAX) ? PK_GS : 0u;
                const uint32_t ngr = P.grp ? (tiles + P.grp - 1u) / P.grp : 0u;
                // the batch's workgroup map: a chain's tiles together on one XCD (mm_pair.inc, PairParams.xseg)
                dim3 rgrid(tiles, G);
                P.xslots = 0;
                uint32_t tof[MM_MAX_GROUPS];
                for (uint32_t g = 0; g < G; ++g) tof[g] = P.bm[g] ? (P.bm[g] + tp - 1u) / tp : 0u;
                if (persist) {
                    // every chain's workgroups on ONE XCD, a CU each: they talk through that XCD's L2 and must all be on the chip
                    const uint32_t slots = pair_xcd_map(P, tof, G, true);
                    if (slots) {
                        P.grp = 0;
                        // (the arrival words are zero: kp_init at the start of the tick, kc_commit behind every batch)
                        // (the batch ends by itself when the longest chain can be compacted into shorter tiles: it may be long)
                        const uint32_t K = e->pair_pbatch, slice = MM_PERSIST_SLICE ? MM_PERSIST_SLICE : K + 1u;
                        for (uint32_t it = 0; it <= K; it += slice)
                            TILE_LAUNCH(kp_rounds, dim3(8u * slots), dim3(PT_THREADS), P, it, it + slice < K + 1u ? it + slice : K + 1u, K);
                        ++e->ps.pair_rounds_launches;
                        after_persist = true;
                        COMPACT_LAUNCH();
                        HIPCHK(e, hipGetLastError());
                        { int arc = results_absorb(e, M.L); if (arc) return arc; }
                        { int src = results_send(e, sent_no, M.L, e->results_early ? MM_RESULTS_MIN_PAIR : 0xFFFFFFFFu); if (src) return src; }
                        continue;
                    }
                    P.xslots = 0;
                }
                if (e->pair_xcd) {
                    const uint32_t slots = pair_xcd_map(P, tof, G);
                    if (slots) rgrid = dim3(8u * slots);
                }
                uint32_t r = e->round_ctr;
                TILE_LAUNCH(kp_round, rgrid, dim3(PT_THREADS), P, r, 1u);
                ++r;
                if (P.grp) TILE_LAUNCH(kp_group, dim3(ngr, G), dim3(1024), P, r);
                for (uint32_t b = 0; b < (want_fit && e->pair_batch > 12u ? 12u : e->pair_batch); ++b) {
                    TILE_LAUNCH(kp_round, rgrid, dim3(PT_THREADS), P, r, 0u);
                    ++e->ps.pair_round_launches;
                    ++r;
                    if (P.grp) TILE_LAUNCH(kp_group, dim3(ngr, G), dim3(1024), P, r);
                }
                TILE_LAUNCH(kp_round_commit, dim3(tiles, G), dim3(256), P, r);
                hipLaunchKernelGGL(kp_round_stage, dim3(G), dim3(64), 0, e->stream, P, r);
                COMPACT_LAUNCH();
                e->round_ctr = r;
            }
#undef COMPACT_LAUNCH
#undef TILE_LAUNCH
            HIPCHK(e, hipGetLastError());
            // the device is busy with the batch: now the host takes what the last look's copies brought (slots released)
            // and sends this look's lobbies after them
            { int arc = results_absorb(e, M.L); if (arc) return arc; }
            { int src = results_send(e, sent_no, M.L, e->results_early ? MM_RESULTS_MIN_PAIR : 0xFFFFFFFFu); if (src) return src; }
        }
    }
    hipLaunchKernelGGL(kp_late, dim3(G), dim3(PL_THREADS), 0, e->stream, P);
    hipLaunchKernelGGL(kp_finish, dim3(G), dim3(1024), 0, e->stream, P);
    HIPCHK(e, hipGetLastError());
    if (tail_send) {
        { int arc = results_absorb(e, M.L); if (arc) return arc; }
        { int src = results_send(e, tail_no, M.L, e->results_early ? MM_RESULTS_MIN_PAIR : 0xFFFFFFFFu); if (src) return src; }
    }
    return MM_OK;
}

// The team path (mm_team.inc).  *any is set when at least one chain was walked by it; the others
// are left to k_walk.
static int team_walk(mm_engine* e, uint32_t mode, const ModeDev& M, bool purge, bool* any)
{
    const mm_config& cfg = e->cfg;
    const uint32_t G = cfg.n_groups;
    *any = false;
    TeamParams P;
    memset(&P, 0, sizeof(P));
    P.mode = mode;
    P.n_groups = G;
    P.capacity = cfg.capacity;
    P.window = M.window > 0xFFFFFu ? 0xFFFFFu : M.window;
    P.eqmask = M.eqmask;
    P.out_cap = (cfg.capacity + MM_MAX_LOBBY) / M.L + 1u;
    P.out_slot_stride = e->out_slot_stride;
    P.out_rec_stride = e->out_rec_stride;
    P.bits_stride = e->pk_bits_stride;
    P.pstride = e->pk_stride;
    P.chunk_stride = e->tk_chunk_stride;
    P.blk_stride = e->pk_stride / 64u;
    P.purge = purge ? 1u : 0u;
    P.state = e->d_state;
    P.released = e->d_released;
    P.n_released = e->d_counters;
    P.scan_cap = e->team_cap;
    P.late_bail = 4u * e->team_late + 32u;
    P.fwait = e->team_fwait;
    P.fix_max = e->team_fix_max;
    P.fix_t8 = e->team_fix_t8;
    P.fix_t4 = e->team_fix_t4;
    P.pull_xcd = e->team_pull_xcd;
    P.nowait = e->team_nowait;
    P.debug = e->pair_debug ? (e->team_batch == 1u ? 3u : 1u) : 0u;   // bit 1 (with MM_TEAM_BATCH=1): kt_f counts every F it writes — an atomic per thread
    P.seq = 0;
    P.n_emit = 0;
    P.M = M;
    P.chains = e->d_chains;
    P.tchains = e->d_tchains;
    P.q_rating = e->d_q_rating;
    P.q_cons = e->d_q_cons;
    P.q_slot = e->d_q_slot;
    P.tkey = e->d_pk_key[0];
    P.bits[0] = e->d_pk_bits[0];
    P.bits[1] = e->d_pk_bits[1];
    P.sqk = e->d_pk_key[1];
    P.sqp = e->d_pk_oidx[0];
    P.fv = e->d_pk_scratch;
    P.vis = e->d_pk_rec1;
    P.blkbase = e->d_pk_oidx[1];
    P.chunk = e->d_tk_chunk;
    P.fv2 = e->d_tk_fv2;
    P.fpos = e->d_tk_fpos;
    P.memb = e->d_tk_memb;
    P.bitsB = e->d_tk_bitsB;
    P.chunkB = e->d_tk_chunkB;
    P.sqi = e->d_tk_sqi;
    P.fdone = e->d_tk_fdone;
    P.use_f2 = 0;
    P.out_slots = e->d_out_slots;
    P.out_score = e->d_out_score;
    P.out_pass = e->d_out_pass;
    hipLaunchKernelGGL(kt_init, dim3(G), dim3(1024), 0, e->stream, P);
    HIPCHK(e, hipGetLastError());
    { int lrc = look_launch(e, e->d_tchains, e->h_tchains, G * sizeof(TeamChain)); if (lrc) return lrc; }
    { int lrc = look_wait(e); if (lrc) return lrc; }
    uint32_t longest = 0;
    unsigned long long arrivals = 0;
    bool late_ok = e->team_late != 0u;
    for (uint32_t g = 0; g < G; ++g) {
        const TeamChain& t = e->h_tchains[g];
        if (!t.fast) continue;
        if (t.m > longest) longest = t.m;
        const uint32_t had = e->tk_last_len[mode * G + g];
        arrivals += t.before > had ? t.before - had : 0u;
        if (t.m > TL_BITS_MAX || t.sitout) late_ok = false;      // (a head that sat out is not in the first pass's sub-queues)
    }
    if (!longest) return MM_OK;
    *any = true;
    const uint32_t nch = (longest + TT_CH - 1u) / TT_CH;
    uint32_t fo_total = 0;
    // kt_fc's order of kt_f's workgroups (TeamParams.fo_*): the chains that are still walked, by falling length, aligned at
    // their ends — made again at every look of the host at the chains, so that a chain that is done costs a pass neither
    // workgroups nor (TeamParams.act) a load
    auto team_order = [&]() {
        uint32_t ord[MM_MAX_GROUPS], len[MM_MAX_GROUPS], K = 0;
        P.act = 0;
        for (uint32_t g = 0; g < G; ++g) {
            const TeamChain& t = e->h_tchains[g];
            if (!t.fast || t.m == 0u || t.done) continue;
            P.act |= 1u << g;
            const uint32_t n = (t.m + TT_CH - 1u) / TT_CH;
            uint32_t at = K++;
            while (at > 0u && len[at - 1u] < n) { ord[at] = ord[at - 1u]; len[at] = len[at - 1u]; --at; }
            ord[at] = g;
            len[at] = n;
        }
        P.fo_n = K;
        fo_total = 0;
        for (uint32_t s = 0; s < K; ++s) {
            P.fo_chain[s] = ord[s];
            P.fo_nch[s] = len[s];
            P.fo_start[s] = fo_total;
            P.fo_dhi[s] = len[s];
            fo_total += (s + 1u) * (len[s] - (s + 1u < K ? len[s + 1u] : 0u));
        }
        P.fo_start[K] = fo_total;
        P.fo_dhi[K] = 0;
    };
    team_order();
    hipLaunchKernelGGL(kt_pack, dim3(nch, G), dim3(TT_CH), 0, e->stream, P);
    HIPCHK(e, hipGetLastError());
    // Batches of passes between two looks at the chains.  The first is short (a tick of a stream seats a handful of
    // lobbies and is over after two passes); once every chain that is still walked emits at most `team_late` lobbies
    // per pass, kt_late walks them to their end in one launch (mm_team.inc) — from the first pass on when the mode has
    // seen only a few arrivals since its last (quiescent) tick: the ticks of a stream.  kt_late hands a chain back
    // after a pass that seated many lobbies after all (`late_bail`); the sub-queues are rebuilt then (it leaves no
    // tombstones behind).
    uint32_t pass = 0, batch = e->team_batch < 2u ? e->team_batch : 2u;
    uint32_t n_emit = longest / (M.L * TC_WAVES * 8u) + 1u;          // the first passes: a worker wave per lobby if an eighth of the chain is seated
    if (n_emit > e->team_emit_max) n_emit = e->team_emit_max;
    uint32_t team_no[MM_MAX_GROUPS];
    bool team_have = false, force_build = false;
    bool late_now = late_ok && e->team_late0 != 0u && arrivals <= e->team_late0;
    ++e->ps.host_looks;
    if (late_now) { hipLaunchKernelGGL(kt_build, dim3(nch, G), dim3(TT_CH), 0, e->stream, P); ++e->ps.team_build_launches; }
    for (uint32_t guard = 0;; ++guard) {
        // a pass that changes nothing ends a chain and every other pass seats somebody
        if (guard > cfg.capacity + 64u) return MM_ERR_INTERNAL;
        if (late_now) {
            hipLaunchKernelGGL(kt_late, dim3(G), dim3(TL_THREADS), 0, e->stream, P);
            ++e->ps.team_late_launches;
            force_build = true;                       // whoever comes back from kt_late needs fresh sub-queues
        } else {
            for (uint32_t b = 0; b < batch; ++b, ++pass) {
                // the first passes of a tick emit hundreds of lobbies each: the chase takes them two at a time (kt_f2)
                P.use_f2 = pass < e->team_f2 ? 1u : 0u;
                // the role sub-queues are rebuilt in the first two passes (a head that sat out the first one is back in
                // the second) and every team_rebuild passes after; in between, players that leave are tombstones in them
                if (pass < 2u || pass % e->team_rebuild == 0u || force_build) {
                    hipLaunchKernelGGL(kt_build, dim3(nch, G), dim3(TT_CH), 0, e->stream, P);
                    ++e->ps.team_build_launches;
                }
                force_build = false;
                // the emitters ride in the chase's launch, enough waves for the lobbies the last look saw per pass (a pass
                // that emits more than that is only slower)
                if (++e->team_seq == 0u) e->team_seq = 1u;
                P.seq = e->team_seq;
                P.n_emit = n_emit;
                P.hf_x = 0xFFFFFFFFu;
                if (P.use_f2) {
                    // (the stored lobbies' fill from the heads of the queues rides in kt_f's launch: MM_TEAM_SPLIT)
                    if (e->team_split) P.hf_x = nch;
                    hipLaunchKernelGGL(kt_f, dim3(nch + (e->team_split ? 1u : 0u), G), dim3(TT_CH), 0, e->stream, P);
                    hipLaunchKernelGGL(kt_f2, dim3(nch, G), dim3(TT_CH), 0, e->stream, P);
                    hipLaunchKernelGGL(kt_chase, dim3(G * (1u + P.n_emit)), dim3(TC_THREADS), 0, e->stream, P);
                    ++e->ps.team_f_launches;
                } else {
                    ++e->ps.team_fc_launches;
                    // kt_f and the chase of the pass in one launch: the chasers take F chunk by chunk as it is written
                    hipLaunchKernelGGL(kt_fc, dim3(G * (1u + P.n_emit) + fo_total), dim3(TT_CH), 0, e->stream, P);
                }
            }
        }
        HIPCHK(e, hipGetLastError());
        uint32_t p0[MM_MAX_GROUPS];
        for (uint32_t g = 0; g < G; ++g) p0[g] = e->h_tchains[g].passes;
        { int lrc = look_launch(e, e->d_tchains, e->h_tchains, G * sizeof(TeamChain)); if (lrc) return lrc; }
        // host work while the device runs the batch: what the last look's copies brought, then this look's lobbies
        { int arc = results_absorb(e, M.L); if (arc) return arc; }
        if (team_have) { int src = results_send(e, team_no, M.L, e->results_early ? MM_RESULTS_MIN_TEAM : 0xFFFFFFFFu); if (src) return src; }
        { int lrc = look_wait(e); if (lrc) return lrc; }
        ++e->ps.host_looks;
        {   // kt_fc's chasers count the anchors whose chunk flag did not come (they looked the lobby up themselves)
            uint32_t fl = 0;
            for (uint32_t g = 0; g < G; ++g) if (e->h_tchains[g].fast) fl += e->h_tchains[g].flags_late;
            e->ps.team_flags_late = fl;
        }
        bool busy = false, late = late_ok;
        uint32_t most = 0;
        for (uint32_t g = 0; g < G; ++g) {
            const TeamChain& t = e->h_tchains[g];
            if (!t.fast || t.done) continue;
            busy = true;
            most = t.n_vis > most ? t.n_vis : most;
            if (t.n_vis > e->team_late) late = false;
        }
        if (e->pair_debug && late_now)
            for (uint32_t g = 0; g < G; ++g) {
                const TeamChain& t = e->h_tchains[g];
                if (t.fast && t.passes != p0[g])
                    fprintf(stderr, "[mm-team-late] g%u: kt_late walked passes %u..%u%s (m %u, sub-queues %u %u %u %u %u): "
                            "%u lobbies by record, %u looked up, %u left open, %u stored fills; entries scanned per role %u %u %u %u %u; "
                            "cycles/16: records %u look-ups %u fills %u ring waits %u all %u\n", g, p0[g], t.passes,
                            t.done ? "" : " and handed the chain back", t.m, t.len[0], t.len[1], t.len[2], t.len[3], t.len[4],
                            t.lt[1], t.lt[2], t.lt[3], t.lt[4], t.lt[5], t.lt[6], t.lt[7], t.lt[8], t.lt[9], t.lt[10], t.lt[11],
                            t.lt[12], t.lt[14], t.lt[13]);
            }
        if (e->pair_debug && e->team_batch == 1u && !late_now) {   // MM_TEAM_BATCH=1: one line per pass, the longest chain
            uint32_t gl = 0;
            for (uint32_t g = 1; g < G; ++g)
                if (e->h_tchains[g].m > e->h_tchains[gl].m) gl = g;
            const TeamChain& tc = e->h_tchains[gl];
            if (guard == 0) { e->dbg_last_w = 0; e->dbg_last_c = 0; }
            fprintf(stderr, "[mm-team-pass] g%u pass %u queued %u lobbies %u F written %u changed %u\n", gl, tc.passes,
                    tc.qlen, tc.n_vis, tc.dbg[5] - e->dbg_last_w, tc.dbg[4] - e->dbg_last_c);
            e->dbg_last_w = tc.dbg[5];
            e->dbg_last_c = tc.dbg[4];
        }
        {   // the lobbies emitted so far leave for the host while the next passes run (sent behind the next batch's launches)
            uint32_t bf[MM_MAX_GROUPS];
            for (uint32_t g = 0; g < G; ++g) { bf[g] = e->h_tchains[g].before; team_no[g] = e->h_tchains[g].fast ? e->h_tchains[g].n_out : 0u; }
            if (!e->r_based) results_set_bases(e, bf, M.L);
            team_have = true;
        }
        if (!busy) break;
        team_order();
        if (late_now) {
            // a chain came back from kt_late (a pass seated more than late_bail lobbies): the pass kernels take over,
            // from the chain's current pass on
            late_now = false;
            if (pass < 2u) pass = 2u;                 // (no doubled rebuild: the tick is past its opening)
            batch = e->team_batch;
            continue;
        }
        late_now = late;
        n_emit = most / (TT_WAVES - 1u) + 1u;                         // (kt_fc's emitter workgroups have seven worker waves)
        if (n_emit > e->team_emit_max) n_emit = e->team_emit_max;
        // close to the switch: look again soon (a look costs a D2H round trip, an idle pass a launch)
        batch = (e->team_late && most <= 3u * e->team_late) ? (e->team_batch < 4u ? e->team_batch : 4u) : e->team_batch;
    }
    if (e->ps.team_flags_late) { e->ps.team_flags_late_total += e->ps.team_flags_late; e->ps.degraded = 1; }
    hipLaunchKernelGGL(kt_fin_scatter, dim3(nch, G), dim3(TT_CH), 0, e->stream, P);
    hipLaunchKernelGGL(kt_fin_copy, dim3(nch, G), dim3(TT_CH), 0, e->stream, P);
    HIPCHK(e, hipGetLastError());
    if (e->pair_debug)
        for (uint32_t g = 0; g < G; ++g)
        {
            const TeamChain& t = e->h_tchains[g];
            const uint32_t np = t.passes + 1u;
            fprintf(stderr, "[mm-team] g%u fast %u m %u passes %u out %u left %u | cancel tick: head sat out %u, seated %u, lobby filtered %u, anchor moved %u x | "
                    "kt_f, the middle chunk, cycles per pass: first loads %u, probes + lists %u, replacements %u, look-ups from scratch %u, barrier %u, step C %u; anchors at work per pass %u | "
                    "F values written %u, changed after the first pass %u | kt_chase: sub-queue entries looked at by the stored lobby's fills %u, by %u look-ups %u\n",
                    g, t.fast, t.m, t.passes, t.n_out, t.qlen,
                    t.dbg[6] & 1u, (t.dbg[6] >> 1) & 1u, (t.dbg[6] >> 2) & 1u, t.dbg[7],
                    t.tmk[0] / np, t.tmk[1] / np, t.tmk[2] / np, t.tmk[3] / np, t.tmk[4] / np, t.tmk[5] / np, t.dbg[2] / np, t.dbg[5], t.dbg[4],
                    t.dbg[0], t.dbg[3], t.dbg[1]);
            if (t.lt[11])
                fprintf(stderr, "[mm-team] g%u kt_fc's chaser: %u passes, %u lobbies by F; cycles per pass: all %u, the chase loop %u, its hops by F %u, of them on kt_f's flags %u; "
                        "cycles per hop without the flags %u; %u look-ups, cycles each: starts %u, the roles' stretches %u, the members' records %u, seats + kills (the one a pass ends on) %u "
                        "(with MM_TEAM_LATE=0: kt_late keeps its own timers in the same words)\n",
                        g, t.lt[11], t.lt[12], 16u * (t.lt[9] / t.lt[11]), 16u * (t.lt[10] / t.lt[11]), 16u * (t.lt[13] / t.lt[11]), 16u * (t.lt[8] / t.lt[11]),
                        t.lt[12] ? (uint32_t)(16ull * (t.lt[13] - t.lt[8]) / t.lt[12]) : 0u,
                        t.lt[3], t.lt[3] ? 16u * (t.lt[4] / t.lt[3]) : 0u, t.lt[3] ? 16u * (t.lt[5] / t.lt[3]) : 0u, t.lt[3] ? 16u * (t.lt[6] / t.lt[3]) : 0u,
                        16u * (t.lt[7] / t.lt[11]));
        }
    return MM_OK;
}

static int tick_impl(mm_engine* e, uint32_t mode, uint32_t* n_matches, mm_stats* stats);

// The exported tick: nothing may unwind into the caller (a NIF frame), and a tick that fails half
// way must not leave kernels in flight behind the error code.
extern "C" int mm_tick(mm_engine* e, uint32_t mode, uint32_t* n_matches, mm_stats* stats)
{
    if (!e || mode >= e->cfg.n_modes) return MM_ERR_INVALID_ARG;
    if (e->poisoned) return MM_ERR_STATE;
    ON_ENGINE_DEVICE(e);
    int rc;
    try {
        rc = tick_impl(e, mode, n_matches, stats);
    } catch (const std::bad_alloc&) {
        rc = MM_ERR_OOM;
    } catch (...) {
        rc = MM_ERR_INTERNAL;
    }
    if (rc != MM_OK) {
        // whatever was enqueued on the stream finishes before the caller sees the error; the queues
        // and lobbies are then in an unspecified (mid-tick) state: mm_reset or mm_restore before
        // the engine is used again (mm_engine.h)
        (void)hipStreamSynchronize(e->stream);
        e->r_n = 0;
        e->poisoned = true;
    }
    return rc;
}

static int tick_impl(mm_engine* e, uint32_t mode, uint32_t* n_matches, mm_stats* stats)
{
    RoctxRange rr_tick("mm_tick");
    const double t0 = host_now_ms();
    const mm_config& cfg = e->cfg;
    const uint32_t G = cfg.n_groups;
    const ModeDev M = make_mode_dev(cfg.modes[mode]);
    const bool timing = (cfg.flags & MM_CFG_TIMING) != 0;
    const bool purge = e->cancel_pending > 0;
    e->r_n = 0;
    e->r_L = M.L;
    {   // mm_path_stats_get: this tick's record (the totals go on)
        const uint32_t tot = e->ps.team_flags_late_total;
        memset(&e->ps, 0, sizeof(e->ps));
        e->ps.mode = mode;
        e->ps.team_flags_late_total = tot;
    }
    if (e->ev_copy_pending) { (void)hipEventSynchronize(e->ev_copy); e->ev_copy_pending = false; }   // (a tick that failed half way)
    e->r_based = false;
    for (uint32_t g = 0; g < G; ++g) { e->r_sent[g] = 0; e->r_marked[g] = 0; e->r_cnt[g] = 0; }

    HIPCHK(e, hipMemsetAsync(e->d_counters, 0, sizeof(uint32_t), e->stream));
    if (timing) HIPCHK(e, hipEventRecord(e->ev[0], e->stream));
    if (purge) {
        hipLaunchKernelGGL(k_purge, dim3(G), dim3(WK_THREADS), 0, e->stream, mode, G, cfg.capacity, e->d_chains,
                           e->d_q_rating, e->d_q_cons, e->d_q_slot, e->d_state, e->d_released, e->d_counters);
        HIPCHK(e, hipGetLastError());
    }
    if (timing) HIPCHK(e, hipEventRecord(e->ev[1], e->stream));
    const bool use_pair = M.team_size == 1u && M.teams == 2u && !e->force_generic;
    if (use_pair) {
        RoctxRange rr("mm_tick/pair walk");
        e->ps.paths |= MM_PATH_PAIR;
        int prc = pair_walk(e, mode, M, purge);
        if (prc) return prc;
    }
    // team modes: long chains take the team path
    bool team_any = false;
    if (!use_pair && !e->force_generic && e->live_upper >= TT_MIN) {
        RoctxRange rr("mm_tick/team walk");
        int trc = team_walk(e, mode, M, purge, &team_any);
        if (trc) return trc;
        if (team_any) e->ps.paths |= MM_PATH_TEAM;
    }
    if (!e->ps.paths) e->ps.paths = MM_PATH_GENERIC;        // (k_walk also takes the short chains the other two leave: not recorded)
    WalkParams P;
    memset(&P, 0, sizeof(P));
    P.mode = mode;
    P.n_groups = G;
    P.capacity = cfg.capacity;
    P.purge = purge ? 1u : 0u;
    P.out_cap = (cfg.capacity + MM_MAX_LOBBY) / M.L + 1u;   // lobbies a group can emit at most
    P.out_slot_stride = e->out_slot_stride;
    P.out_rec_stride = e->out_rec_stride;
    P.max_passes = 0x7FFFFFFFu;
    P.M = M;
    P.chains = e->d_chains;
    P.q_rating = e->d_q_rating;
    P.q_cons = e->d_q_cons;
    P.q_slot = e->d_q_slot;
    P.state = e->d_state;
    P.released = e->d_released;
    P.n_released = e->d_counters;
    P.out_slots = e->d_out_slots;
    P.out_score = e->d_out_score;
    P.out_pass = e->d_out_pass;
    P.pskip = use_pair ? e->d_pchains : NULL;
    P.tskip = team_any ? e->d_tchains : NULL;
    hipLaunchKernelGGL(k_walk, dim3(G), dim3(WK_THREADS), 0, e->stream, P);
    HIPCHK(e, hipGetLastError());
    if (timing) HIPCHK(e, hipEventRecord(e->ev[2], e->stream));
    HIPCHK(e, hipMemcpyAsync(e->h_chains, e->d_chains, e->n_chains * sizeof(ChainDev), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(e->h_counters, e->d_counters, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    // (mm_path_stats: the pair chains' records as kp_late and kp_finish left them — the last look was taken before them)
    if (use_pair) HIPCHK(e, hipMemcpyAsync(e->h_pchains, e->d_pchains, G * sizeof(PairChain), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    const double t_copy0 = host_now_ms();
    if (use_pair && e->pair_debug) {
        std::vector<PairChain> hp(G);
        HIPCHK(e, hipMemcpy(hp.data(), e->d_pchains, G * sizeof(PairChain), hipMemcpyDeviceToHost));
        for (uint32_t g = 0; g < G; ++g)
            fprintf(stderr, "[mm-pair] g%u fast %u m %u qlen %u passes %u out %u rounds %u (group hops of tile 0: %u of %u entries looked at, %u not of the round) | stale %u inv %u slowsucc %u compact %u fixed(w1) %u | chase clk %u wall(100MHz) %u wrapscan clk %u\n",
                    g, hp[g].fast, hp[g].m, hp[g].qlen, hp[g].passes, hp[g].n_out, hp[g].rounds, hp[g].ghops, hp[g].gtry, hp[g].gstale, hp[g].dbg[0], hp[g].dbg[1],
                    hp[g].dbg[2], hp[g].dbg[3], hp[g].dbg[4], hp[g].dbg[5], hp[g].dbg[6], hp[g].dbg[7]);
        for (uint32_t g = 0; g < G; ++g)
            if (hp[g].rounds) {
                const uint32_t nr = hp[g].tm[5] ? hp[g].tm[5] : 1u;
                fprintf(stderr, "[mm-pair] g%u tile1 cycles/round over %u rounds (load %u walk+stage+sweepA %u apply %u | detect %u items %u long %u (%.1f long items) | graph %u resolve %u publish %u | cand+head %u)\n",
                        g, nr, hp[g].tm[0] / nr, hp[g].tm[6] / nr, hp[g].tm[9] / nr, hp[g].tm[1] / nr, hp[g].tm[12] / nr, hp[g].tm[2] / nr,
                        (double)hp[g].tm[11] / nr, hp[g].tm[7] / nr, hp[g].tm[8] / nr, hp[g].tm[10] / nr, hp[g].tm[4] / nr);
            }
        for (uint32_t g = 0; g < G; ++g)
            if (hp[g].ppass) {
                const uint32_t nr = hp[g].ptm[5] ? hp[g].ptm[5] : 1u;
                fprintf(stderr, "[mm-pair] g%u kp_rounds: %u passes inside persistent launches (%u launches gave up so far); tile1 cycles/pass over %u passes (barrier %u fast hops %u walk+sweepA %u apply %u | detect %u items %u long %u | graph %u resolve %u publish %u | head+arrive %u) walk loop %.0f cycles, %.1f hops\n",
                        g, hp[g].ppass, e->pair_pstops, nr, hp[g].ptm[0] / nr, hp[g].ptm[3] / nr, hp[g].ptm[6] / nr, hp[g].ptm[9] / nr, hp[g].ptm[1] / nr,
                        hp[g].ptm[12] / nr, hp[g].ptm[2] / nr, hp[g].ptm[7] / nr, hp[g].ptm[8] / nr, hp[g].ptm[10] / nr, hp[g].ptm[4] / nr,
                        16.0 * hp[g].ptm[13] / nr, (double)hp[g].ptm[15] / nr);
                fprintf(stderr, "[mm-pair] g%u kp_rounds, tile 1's walker off the fast path: %u hops whose partner had left, %.0f cycles each; %u hops without a partner inside the horizon, %.0f cycles each\n",
                        g, hp[g].pslow[0], hp[g].pslow[0] ? 16.0 * hp[g].pslow[1] / hp[g].pslow[0] : 0.0, hp[g].pslow[2],
                        hp[g].pslow[2] ? 16.0 * hp[g].pslow[3] / hp[g].pslow[2] : 0.0);
            }
        {   // where the last kp_rounds launch ran and where the tables live (run-to-run spread: two modes of the walk, 9.4 / 9.7 ms)
            fprintf(stderr, "[mm-pair] physical XCD mask of each chain's workgroups in its last kp_rounds launch:");
            for (uint32_t g = 0; g < G; ++g) fprintf(stderr, " g%u=0x%x", g, hp[g].pxm);
            fprintf(stderr, " | key %p rec %p %p bitsp %p %p pbar %p stream %p\n", (void*)e->d_pk_key[0], (void*)e->d_pk_scratch, (void*)e->d_pk_rec1,
                    (void*)e->d_pk_bitsp[0], (void*)e->d_pk_bitsp[1], (void*)e->d_pk_pbar, (void*)e->stream);
        }
        if (e->pair_tune & 0x10000u) {   // every tile's own work per pass inside kp_rounds: who does the chain's barrier wait for?
            const uint32_t nt = 40u;
            std::vector<uint4> dg(nt);
            for (uint32_t g = 0; g < G; ++g) {
                if (!hp[g].ppass) continue;
                HIPCHK(e, hipMemcpy(dg.data(), e->d_pk_grec + (size_t)g * e->pk_gstride, nt * sizeof(uint4), hipMemcpyDeviceToHost));
                fprintf(stderr, "[mm-pair] g%u kp_rounds, cycles from barrier to arrival per tile (passes):", g);
                for (uint32_t t = 0; t < nt; ++t) if (dg[t].y && dg[t].y != 0xFFFFFFFFu) fprintf(stderr, " t%u=%u(%u)", t, dg[t].x / dg[t].y, dg[t].y);
                fprintf(stderr, "\n");
            }
        }
        if (e->pair_tune & 0x2000u) {
            unsigned long long tested = 0, algo = 0;
            for (uint32_t g = 0; g < G; ++g) { tested += hp[g].tested; algo += hp[g].pairs; }
            fprintf(stderr, "[mm-pair] predicate tests the kernels physically performed (next[] build, repairs, head scans, LDS-resident walk): %llu = %.2f x the %llu pair evaluations of the reference algorithm\n",
                    tested, algo ? (double)tested / (double)algo : 0.0, algo);
            std::vector<uint32_t> xc(e->pk_max_tiles);
            HIPCHK(e, hipMemcpy(xc.data(), e->d_pk_tilectl + ((size_t)TC_DISP * G + 0) * e->pk_max_tiles,
                                e->pk_max_tiles * sizeof(uint32_t), hipMemcpyDeviceToHost));
            fprintf(stderr, "[mm-pair] g0 XCD of tile 0..47's workgroup at its last round:");
            for (uint32_t t = 0; t < 48 && t < e->pk_max_tiles; ++t) fprintf(stderr, " %u", xc[t] & 0xFu);
            fprintf(stderr, "\n");
        }
        if (e->pair_tune & 0x2000u)
            for (uint32_t g = 0; g < G; ++g)
                if (hp[g].rounds)
                    fprintf(stderr, "[mm-pair] g%u walk loop of tile 1's workgroup: %u iterations, %.0f cycles each (%.0f cycles per timed round)\n",
                            g, hp[g].tm[15], hp[g].tm[15] ? 16.0 * hp[g].tm[13] / hp[g].tm[15] : 0.0,
                            16.0 * hp[g].tm[13] / (hp[g].tm[5] ? hp[g].tm[5] : 1u));
        if (e->pair_tune & 0x2000u)
            fprintf(stderr, "[mm-pair] g0 walk: cycles in iterations 1-4 per round %.0f, in the later ones %.0f\n",
                    16.0 * hp[0].tm[12] / (hp[0].tm[5] ? hp[0].tm[5] : 1u), 16.0 * hp[0].tm[14] / (hp[0].tm[5] ? hp[0].tm[5] : 1u));
    }

    uint32_t total = 0, after = 0, before = 0, pmax = 0, errf = 0;
    unsigned long long pairs = 0, scanned = 0;
    for (uint32_t g = 0; g < G; ++g) {
        const ChainDev& c = e->h_chains[mode * G + g];
        total += c.n_out;
        after += c.len + c.lobby.n;
        before += c.before;
        pairs += c.pairs;
        scanned += c.scanned;
        errf |= c.err;
        if (c.passes > pmax) pmax = c.passes;
    }
    if (errf) return MM_ERR_INTERNAL;
    if (e->fault_tick && ++e->ticks_seen == e->fault_tick) return MM_ERR_INTERNAL;   // test hook: a tick that dies after its walk
    if ((size_t)total * M.L > (size_t)cfg.capacity + (size_t)MM_MAX_LOBBY * G) return MM_ERR_INTERNAL;
    RoctxRange rr_res("mm_tick/match list");
    bool tail_on_main = false;
    // The match list, group-major emission order.  Most of it left for the host while the walk was still running
    // (results_send at every look at the chains); what the last kernels emitted follows now, and the host does its
    // part for what has arrived meanwhile — ActiveUser.remove_user for the matched players (game-lobby/worker.ex:73-103).
    {
        uint32_t bf[MM_MAX_GROUPS], no[MM_MAX_GROUPS];
        for (uint32_t g = 0; g < G; ++g) {
            const ChainDev& c = e->h_chains[mode * G + g];
            bf[g] = c.n_out * M.L;                                   // tight: the counts are final
            no[g] = c.n_out;
            if (e->r_based && (c.n_out < e->r_sent[g] || (unsigned long long)c.n_out * M.L > (unsigned long long)c.before + M.L))
                return MM_ERR_INTERNAL;
        }
        if (!e->r_based && total > 0u && total <= MM_PACK_MAX) {
            // nothing has left yet and the tick is small: one packed piece, tight layout (group g at its prefix)
            PackArgs A;
            memset(&A, 0, sizeof(A));
            A.n_groups = G; A.L = M.L; A.total = total; A.out_slot_stride = e->out_slot_stride; A.out_rec_stride = e->out_rec_stride;
            for (uint32_t g = 0; g < G; ++g) { A.pre[g + 1u] = A.pre[g] + no[g]; e->r_base[g] = A.pre[g]; e->r_sent[g] = no[g]; }
            e->r_based = true;
            hipLaunchKernelGGL(k_pack_results, dim3((total + 255u) / 256u), dim3(256), 0, e->stream, A, e->d_out_slots, e->d_out_score,
                               e->d_out_pass, e->d_pack);
            HIPCHK(e, hipGetLastError());
            HIPCHK(e, hipMemcpyAsync(e->h_rslots, e->d_pack, (size_t)total * M.L * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
            HIPCHK(e, hipMemcpyAsync(e->h_rscore, e->d_pack + (size_t)total * M.L, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, e->stream));
            HIPCHK(e, hipMemcpyAsync(e->h_rpass, e->d_pack + (size_t)total * (M.L + 1u), (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
            tail_on_main = true;
        } else {
            if (!e->r_based) results_set_bases(e, bf, M.L);
            int src = results_send(e, no, M.L, 1u, &tail_on_main);
            if (src) return src;
        }
    }
    const uint32_t nrel = e->h_counters[0];
    e->r_released.resize(nrel);
    if (nrel)
        HIPCHK(e, hipMemcpyAsync(e->r_released.data(), e->d_released, nrel * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    {
        int src = results_absorb(e, M.L);
        if (src) return src;
    }
    e->r_pre[0] = 0;
    for (uint32_t g = 0; g < G; ++g) {
        e->tk_last_len[mode * G + g] = e->h_chains[mode * G + g].len + e->h_chains[mode * G + g].lobby.n;
        e->r_cnt[g] = e->h_chains[mode * G + g].n_out;
        e->r_pre[g + 1u] = e->r_pre[g] + e->r_cnt[g];
    }
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (tail_on_main)
        for (uint32_t g = 0; g < G; ++g) results_mark(e, g, e->r_sent[g], M.L);
    // ... and the slots the liveness filter released
    for (uint32_t i = 0; i < nrel; ++i) e->h_state[e->r_released[i]] = MM_ST_FREE;
    e->cancel_pending = e->cancel_pending >= nrel ? e->cancel_pending - nrel : 0;
    {
        const unsigned long long gone = (unsigned long long)total * M.L + nrel;
        e->live_upper = e->live_upper >= gone ? e->live_upper - gone : 0;
    }
    e->r_n = total;
    if (n_matches) *n_matches = total;
    if (e->ps.paths & MM_PATH_PAIR) {            // mm_path_stats_get: the chain with the most passes, by launch shape
        // ... among the chains the pair path walked: one k_walk took (PairChain.fast == 0: a rating span beyond the packed key)
        // has no launch shapes to report, and its passes counted as kp_late's would price them as LDS steps
        uint32_t gc = 0xFFFFFFFFu;
        for (uint32_t g = 0; g < G; ++g)
            if (e->h_pchains[g].fast && (gc == 0xFFFFFFFFu || e->h_chains[mode * G + g].passes > e->h_chains[mode * G + gc].passes)) gc = g;
        e->ps.crit_group = gc;
        if (gc == 0xFFFFFFFFu) gc = 0;            // (none: the zeros of ps_hand say so)
        const ChainDev& cd = e->h_chains[mode * G + gc];
        e->ps.crit_passes = e->ps.crit_group == 0xFFFFFFFFu ? 0u : cd.passes;
        e->ps.crit_rounds_passes = e->ps_hand[2][gc];
        e->ps.crit_rounds_hops = e->ps_hand[3][gc];
        e->ps.crit_round_passes = e->ps_hand[0][gc] >= e->ps_hand[2][gc] ? e->ps_hand[0][gc] - e->ps_hand[2][gc] : 0u;
        e->ps.crit_late_passes = cd.passes >= e->ps_hand[0][gc] ? cd.passes - e->ps_hand[0][gc] : 0u;
        e->ps.crit_late_lobbies = cd.n_out >= e->ps_hand[1][gc] ? cd.n_out - e->ps_hand[1][gc] : 0u;
        // the primitives of its serial chain as tile 1's walker timed them (PairChain.ptm), the clock (kp_late's chase in
        // shader-clock cycles and in 100 MHz ticks), kp_nx_init by HIP events, the physical predicate tests (pair_tune bit 13)
        const PairChain& pc = e->h_pchains[gc];
        if (e->ps.crit_group != 0xFFFFFFFFu) {
            e->ps.crit_timed_passes = pc.ptm[5];
            e->ps.crit_timed_hops = pc.ptm[15];
            e->ps.crit_barrier_cycles = pc.ptm[0];
            e->ps.crit_hop_cycles = pc.ptm[3];
            e->ps.clk_cycles = pc.dbg[5];
            e->ps.clk_wall_ticks = pc.dbg[6];
        }
        if (e->ev_nx_set) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e->ev_nx[0], e->ev_nx[1]) == hipSuccess) e->ps.pair_nx_init_ns = (uint32_t)(ms * 1e6f);
            e->ev_nx_set = false;
        }
        if (e->pair_tune & 0x2000u) {
            unsigned long long t = 0, tn = 0;
            for (uint32_t g = 0; g < G; ++g) if (e->h_pchains[g].fast) { t += e->h_pchains[g].tested; tn += e->h_pchains[g].tested_nx; }
            e->ps.pair_tested_lo = (uint32_t)t; e->ps.pair_tested_hi = (uint32_t)(t >> 32);
            e->ps.pair_tested_nx_lo = (uint32_t)tn; e->ps.pair_tested_nx_hi = (uint32_t)(tn >> 32);
        }
    }
    e->ps.crit_team_group = 0xFFFFFFFFu;
    if (e->ps.paths & MM_PATH_TEAM) {            // ... and the team path's: what its chaser did, by launch shape (TeamChain.cp)
        uint32_t gc = 0xFFFFFFFFu;
        for (uint32_t g = 0; g < G; ++g)
            if (e->h_tchains[g].fast && (gc == 0xFFFFFFFFu || e->h_tchains[g].passes > e->h_tchains[gc].passes)) gc = g;
        if (gc != 0xFFFFFFFFu) {
            const TeamChain& t = e->h_tchains[gc];
            e->ps.crit_team_group = gc;
            e->ps.crit_team_passes = t.passes;
            e->ps.crit_team_f_passes = t.cp[0]; e->ps.crit_team_fc_passes = t.cp[1]; e->ps.crit_team_late_passes = t.cp[2];
            e->ps.crit_team_f_lobbies = t.cp[3]; e->ps.crit_team_fc_lobbies = t.cp[4]; e->ps.crit_team_late_lobbies = t.cp[5];
            e->ps.crit_team_lookups = t.cp[6]; e->ps.crit_team_late_lookups = t.cp[7];
        }
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->pool_before = before;
        stats->pool_after = after;
        stats->matches = total;
        stats->players_matched = total * M.L;
        stats->passes_max = pmax;
        stats->chains = G;
        stats->pairs = pairs;
        stats->scanned = scanned;
        if (timing) {
            (void)hipEventElapsedTime(&stats->filter_ms, e->ev[0], e->ev[1]);
            (void)hipEventElapsedTime(&stats->walk_ms, e->ev[1], e->ev[2]);
        }
        const double t1 = host_now_ms();
        stats->copy_ms = (float)(t1 - t_copy0);
        stats->total_ms = (float)(t1 - t0);
    }
    return MM_OK;
}

extern "C" int mm_matches(mm_engine* e, uint32_t first, uint32_t count, uint32_t* slots, float* score,
                          uint32_t* group, uint32_t* pass)
{
    try {
        if (!e) return MM_ERR_INVALID_ARG;
        if (first > e->r_n || count > e->r_n - first) return MM_ERR_RANGE;
        if (count == 0) return MM_OK;
        // emission index i of the tick = lobby i - r_pre[g] of group g, stored at r_base[g] + that (tick_impl)
        const uint32_t L = e->r_L;
        // (Round 5, measured and taken out: the 10 MB of a big tick's list copied by four threads in 256 KB pieces —
        // 10.01-10.28 ms a step with one thread, 10.03-10.26 with four, profiles/r05_ab_pair_micro.txt.)
        for (uint32_t g = 0; g < e->cfg.n_groups; ++g) {
            const uint32_t lo = e->r_pre[g] > first ? e->r_pre[g] : first;
            const uint32_t hi = e->r_pre[g + 1u] < first + count ? e->r_pre[g + 1u] : first + count;
            if (hi <= lo) continue;
            const size_t src = (size_t)e->r_base[g] + (lo - e->r_pre[g]), dst = lo - first, n = hi - lo;
            if (slots) memcpy(slots + dst * L, &e->h_rslots[src * L], n * L * sizeof(uint32_t));
            if (score) memcpy(score + dst, &e->h_rscore[src], n * sizeof(float));
            if (pass) memcpy(pass + dst, &e->h_rpass[src], n * sizeof(uint32_t));
            if (group) std::fill_n(group + dst, n, g);
        }
        return MM_OK;
    } catch (const std::bad_alloc&) {
        return MM_ERR_OOM;
    } catch (...) {
        return MM_ERR_INTERNAL;
    }
}

static int fetch_chains(mm_engine* e)
{
    HIPCHK(e, hipMemcpyAsync(e->h_chains, e->d_chains, e->n_chains * sizeof(ChainDev), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return MM_OK;
}

extern "C" int mm_queue_depth(mm_engine* e, uint32_t mode, uint32_t* per_group)
{
    try {
        if (!e || !per_group || mode >= e->cfg.n_modes) return MM_ERR_INVALID_ARG;
        ON_ENGINE_DEVICE(e);
        int rc = fetch_chains(e);
        if (rc) return rc;
        for (uint32_t g = 0; g < e->cfg.n_groups; ++g) per_group[g] = e->h_chains[mode * e->cfg.n_groups + g].len;
        return MM_OK;
    } catch (const std::bad_alloc&) {
        return MM_ERR_OOM;
    } catch (...) {
        return MM_ERR_INTERNAL;
    }
}

extern "C" int mm_queue_slots(mm_engine* e, uint32_t mode, uint32_t group, uint32_t* n, uint32_t* slots)
{
    try {
        if (!e || !n || mode >= e->cfg.n_modes || group >= e->cfg.n_groups) return MM_ERR_INVALID_ARG;
        ON_ENGINE_DEVICE(e);
        int rc = fetch_chains(e);
        if (rc) return rc;
        const uint32_t c = mode * e->cfg.n_groups + group;
        const uint32_t len = e->h_chains[c].len;
        const uint32_t k = len < *n ? len : *n;
        *n = len;
        if (slots && k) {
            // the queue lives in q_slot[chain][0 .. len) in order, head at 0
            HIPCHK(e, hipMemcpyAsync(slots, e->d_q_slot + (size_t)c * e->cfg.capacity, (size_t)k * sizeof(uint32_t),
                                     hipMemcpyDeviceToHost, e->stream));
            HIPCHK(e, hipStreamSynchronize(e->stream));
        }
        return MM_OK;
    } catch (const std::bad_alloc&) {
        return MM_ERR_OOM;
    } catch (...) {
        return MM_ERR_INTERNAL;
    }
}

extern "C" int mm_lobby_state(mm_engine* e, uint32_t mode, uint32_t group, uint32_t* n, uint32_t* slots,
                              uint8_t* teams)
{
    try {
        if (!e || !n || mode >= e->cfg.n_modes || group >= e->cfg.n_groups) return MM_ERR_INVALID_ARG;
        ON_ENGINE_DEVICE(e);
        int rc = fetch_chains(e);
        if (rc) return rc;
        const LobbyDev& lb = e->h_chains[mode * e->cfg.n_groups + group].lobby;
        uint32_t k = 0;
        for (uint32_t t = 0; t < e->cfg.modes[mode].teams; ++t)
            for (uint32_t j = 0; j < lb.cnt[t] && j < 8; ++j) {
                if (slots) slots[k] = lb.slot[t][j];
                if (teams) teams[k] = (uint8_t)t;
                ++k;
            }
        *n = k;
        return MM_OK;
    } catch (const std::bad_alloc&) {
        return MM_ERR_OOM;
    } catch (...) {
        return MM_ERR_INTERNAL;
    }
}

// ------------------------------------------------------------------------------------
// pool snapshot / restore
// ------------------------------------------------------------------------------------
struct SnapHeader {
    uint32_t magic, version, abi, header_bytes;
    uint32_t capacity, n_groups, n_modes, n_chains;
    uint32_t next_slot, cancel_pending, chain_bytes, reserved;
    unsigned long long live_upper, cfg_hash, payload_hash, total_bytes;
};
#define MM_SNAP_MAGIC 0x4E534D4Du   /* "MMSN" */

static unsigned long long snap_hash(const void* p, size_t n, unsigned long long h)
{
    const unsigned char* b = (const unsigned char*)p;   // FNV-1a, 64 bit
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001B3ull; }
    return h;
}

static unsigned long long snap_cfg_hash(const mm_config& c)
{
    unsigned long long h = 0xCBF29CE484222325ull;
    h = snap_hash(&c.n_groups, sizeof(c.n_groups), h);
    h = snap_hash(c.groups, sizeof(mm_rating_group) * c.n_groups, h);
    h = snap_hash(&c.default_group, sizeof(c.default_group), h);
    h = snap_hash(&c.n_modes, sizeof(c.n_modes), h);
    for (uint32_t m = 0; m < c.n_modes; ++m) {
        const mm_mode_config& mc = c.modes[m];
        h = snap_hash(&mc.team_size, sizeof(mc.team_size), h);
        h = snap_hash(&mc.teams, sizeof(mc.teams), h);
        h = snap_hash(&mc.window, sizeof(mc.window), h);
        h = snap_hash(&mc.flags, sizeof(mc.flags), h);
        h = snap_hash(&mc.n_roles, sizeof(mc.n_roles), h);
        h = snap_hash(mc.role_quota, mc.n_roles, h);
    }
    h = snap_hash(&c.capacity, sizeof(c.capacity), h);
    return h;
}

extern "C" int mm_snapshot_size(mm_engine* e, uint64_t* bytes)
{
    try {
        if (!e || !bytes) return MM_ERR_INVALID_ARG;
        ON_ENGINE_DEVICE(e);
        int rc = fetch_chains(e);
        if (rc) return rc;
        uint64_t n = sizeof(SnapHeader) + e->cfg.capacity + (uint64_t)e->n_chains * sizeof(ChainDev);
        for (uint32_t c = 0; c < e->n_chains; ++c) n += (uint64_t)e->h_chains[c].len * 12u;
        *bytes = n;
        return MM_OK;
    } catch (const std::bad_alloc&) {
        return MM_ERR_OOM;
    } catch (...) {
        return MM_ERR_INTERNAL;
    }
}

extern "C" int mm_snapshot(mm_engine* e, void* buf, uint64_t cap, uint64_t* written)
{
    try {
        if (!e || !buf || !written) return MM_ERR_INVALID_ARG;
        if (e->poisoned) return MM_ERR_STATE;                 // a mid-tick pool is not a pool worth keeping
        ON_ENGINE_DEVICE(e);
        uint64_t need = 0;
        int rc = mm_snapshot_size(e, &need);                  // also refreshes h_chains
        if (rc) return rc;
        if (cap < need) return MM_ERR_RANGE;
        unsigned char* out = (unsigned char*)buf;
        SnapHeader h;
        memset(&h, 0, sizeof(h));
        h.magic = MM_SNAP_MAGIC;
        h.version = 1;
        h.abi = MM_ABI_VERSION;
        h.header_bytes = (uint32_t)sizeof(SnapHeader);
        h.capacity = e->cfg.capacity;
        h.n_groups = e->cfg.n_groups;
        h.n_modes = e->cfg.n_modes;
        h.n_chains = e->n_chains;
        h.next_slot = e->next_slot;
        h.cancel_pending = e->cancel_pending;
        h.chain_bytes = (uint32_t)sizeof(ChainDev);
        h.live_upper = e->live_upper;
        h.cfg_hash = snap_cfg_hash(e->cfg);
        h.total_bytes = need;
        size_t off = sizeof(SnapHeader);
        memcpy(out + off, e->h_state.data(), e->cfg.capacity);                 // ActiveUser mirror
        off += e->cfg.capacity;
        memcpy(out + off, e->h_chains, (size_t)e->n_chains * sizeof(ChainDev)); // lengths + stored lobbies
        off += (size_t)e->n_chains * sizeof(ChainDev);
        const size_t cap_q = e->cfg.capacity;
        for (uint32_t c = 0; c < e->n_chains; ++c) {                             // the queues, in order
            const size_t len = e->h_chains[c].len;
            if (!len) continue;
            HIPCHK(e, hipMemcpyAsync(out + off, e->d_q_rating + c * cap_q, len * 4u, hipMemcpyDeviceToHost, e->stream));
            HIPCHK(e, hipMemcpyAsync(out + off + len * 4u, e->d_q_cons + c * cap_q, len * 4u, hipMemcpyDeviceToHost, e->stream));
            HIPCHK(e, hipMemcpyAsync(out + off + len * 8u, e->d_q_slot + c * cap_q, len * 4u, hipMemcpyDeviceToHost, e->stream));
            off += len * 12u;
        }
        HIPCHK(e, hipStreamSynchronize(e->stream));
        h.payload_hash = snap_hash(out + sizeof(SnapHeader), (size_t)need - sizeof(SnapHeader), 0xCBF29CE484222325ull);
        memcpy(out, &h, sizeof(h));
        *written = need;
        return MM_OK;
    } catch (const std::bad_alloc&) {
        return MM_ERR_OOM;
    } catch (...) {
        return MM_ERR_INTERNAL;
    }
}

extern "C" int mm_restore(mm_engine* e, const void* buf, uint64_t bytes)
{
    try {
        if (!e || !buf || bytes < sizeof(SnapHeader)) return MM_ERR_INVALID_ARG;
        ON_ENGINE_DEVICE(e);
        const unsigned char* in = (const unsigned char*)buf;
        SnapHeader h;
        memcpy(&h, in, sizeof(h));
        if (h.magic != MM_SNAP_MAGIC || h.version != 1u || h.abi != MM_ABI_VERSION || h.header_bytes != sizeof(SnapHeader) ||
            h.chain_bytes != sizeof(ChainDev) || h.total_bytes != bytes || h.capacity != e->cfg.capacity ||
            h.n_groups != e->cfg.n_groups || h.n_modes != e->cfg.n_modes || h.n_chains != e->n_chains ||
            h.cfg_hash != snap_cfg_hash(e->cfg))
            return MM_ERR_INVALID_ARG;
        if (h.payload_hash != snap_hash(in + sizeof(SnapHeader), (size_t)bytes - sizeof(SnapHeader), 0xCBF29CE484222325ull))
            return MM_ERR_INVALID_ARG;
        // structure check before anything is touched: the queue lengths have to add up
        size_t off = sizeof(SnapHeader) + h.capacity;
        const ChainDev* chains = (const ChainDev*)(in + off);
        uint64_t need = off + (uint64_t)h.n_chains * sizeof(ChainDev);
        for (uint32_t c = 0; c < h.n_chains; ++c) {
            ChainDev cd;
            memcpy(&cd, (const unsigned char*)chains + (size_t)c * sizeof(ChainDev), sizeof(cd));
            if (cd.len > h.capacity) return MM_ERR_INVALID_ARG;
            need += (uint64_t)cd.len * 12u;
        }
        if (need != bytes) return MM_ERR_INVALID_ARG;
        int rc = mm_reset(e);
        if (rc) return rc;
        memcpy(e->h_state.data(), in + sizeof(SnapHeader), h.capacity);
        HIPCHK(e, hipMemcpyAsync(e->d_state, e->h_state.data(), h.capacity, hipMemcpyHostToDevice, e->stream));
        memcpy(e->h_chains, in + off, (size_t)h.n_chains * sizeof(ChainDev));
        for (uint32_t c = 0; c < h.n_chains; ++c) {          // per-tick counters do not survive a restart
            ChainDev& cd = e->h_chains[c];
            cd.head_state = 0; cd.n_out = 0; cd.passes = 0; cd.err = 0; cd.purged = 0; cd.before = 0;
            cd.pairs = 0; cd.scanned = 0;
        }
        HIPCHK(e, hipMemcpyAsync(e->d_chains, e->h_chains, (size_t)h.n_chains * sizeof(ChainDev), hipMemcpyHostToDevice, e->stream));
        off += (size_t)h.n_chains * sizeof(ChainDev);
        const size_t cap_q = e->cfg.capacity;
        for (uint32_t c = 0; c < h.n_chains; ++c) {
            const size_t len = e->h_chains[c].len;
            if (!len) continue;
            HIPCHK(e, hipMemcpyAsync(e->d_q_rating + c * cap_q, in + off, len * 4u, hipMemcpyHostToDevice, e->stream));
            HIPCHK(e, hipMemcpyAsync(e->d_q_cons + c * cap_q, in + off + len * 4u, len * 4u, hipMemcpyHostToDevice, e->stream));
            HIPCHK(e, hipMemcpyAsync(e->d_q_slot + c * cap_q, in + off + len * 8u, len * 4u, hipMemcpyHostToDevice, e->stream));
            off += len * 12u;
        }
        HIPCHK(e, hipStreamSynchronize(e->stream));
        e->next_slot = h.next_slot;
        e->cancel_pending = h.cancel_pending;
        e->live_upper = h.live_upper;
        return MM_OK;
    } catch (const std::bad_alloc&) {
        return MM_ERR_OOM;
    } catch (...) {
        return MM_ERR_INTERNAL;
    }
}

extern "C" int mm_last_hip_error(const mm_engine* e) { return e ? e->last_hip : 0; }

// include/mm_engine.h: the launch shapes and fall-backs of the last tick.  Host state only — no device call.
extern "C" int mm_path_stats_get(mm_engine* e, mm_path_stats* out)
{
    if (!e || !out || out->size < 2u * sizeof(uint32_t)) return MM_ERR_INVALID_ARG;
    mm_path_stats ps = e->ps;
    ps.pair_persist_off = e->pair_persist ? 0u : 1u;
    ps.pair_cooldown = e->pair_pcool;
    ps.pair_stops_total = e->pair_pstops;
    const uint32_t n = out->size < (uint32_t)sizeof(ps) ? out->size : (uint32_t)sizeof(ps);
    ps.size = n;
    memcpy(out, &ps, n);
    return MM_OK;
}

#include "mm_codec.inc"
