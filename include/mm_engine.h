/*
 * mm_engine.h — C ABI of the MI355X-native matchmaking search engine.
 *
 * This is the drop-in boundary for ONE path of OpenMatchmaking/microservice-matchmaking:
 * the search/seed loop of `Matchmaking.Search.Worker` plus the external
 * `strategist.match.check` predicate it calls.  The reference has no FFI of its own
 * (it is 100 % Elixir); these are the entry points an Erlang dirty NIF would bind
 * (native/mm_nif.c is that NIF, native/elixir/ the Elixir modules; INTEGRATION.md: how they replace
 * `Search.Worker.consume/5`).  Every export cites the reference code it replaces;
 * paths are relative to /root/reference/matchmaking/.
 *
 * Rules that hold for every function:
 *   - plain C types only, no torch/HIP types in signatures;
 *   - returns 0 (MM_OK) or a negative mm_status; never throws, aborts or exits
 *     (a NIF crash would take the whole BEAM down — contrast the per-worker
 *     `restart: :transient` isolation at lib/application.ex:8-14): every entry point that can
 *     allocate catches at the boundary (MM_ERR_OOM for std::bad_alloc, MM_ERR_INTERNAL otherwise);
 *   - after MM_ERR_HIP / MM_ERR_INTERNAL / MM_ERR_OOM from mm_tick the device work already queued has
 *     finished (the stream is synchronised before the error is returned) but queues and lobbies are
 *     in a mid-tick state: call mm_reset or mm_restore before using the engine again — until then
 *     mm_enqueue*, mm_cancel, mm_tick and mm_snapshot answer MM_ERR_STATE, so an owner that only
 *     logged the error cannot go on to publish lobbies from a half-walked pool;
 *   - one owner per engine, no internal locking: calls on one engine never overlap (one
 *     GenServer owns one engine, as one Search.Worker owns one channel:
 *     lib/search/worker.ex:220-237).  The owner need not stay on one OS thread — a dirty NIF
 *     runs on whichever dirty scheduler is free — so every entry point selects the engine's
 *     HIP device itself (the current device is per-thread state in HIP) and restores the
 *     caller's before it returns;
 *   - distinct engines are independent (own HIP stream, own device memory).
 *
 * The library has exactly one backend: hand-written HIP kernels for gfx950.
 * There is no CPU fallback; mm_engine_create() fails with MM_ERR_NO_DEVICE when no
 * GPU is usable.  The CPU restatement used for parity testing lives in oracle/ and is
 * never linked into this library.
 */
#ifndef MM_ENGINE_H
#define MM_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_ABI_VERSION 1u

#define MM_MAX_GROUPS 16u /* rating groups (reference ships 7: config/config.exs:27-36) */
#define MM_MAX_MODES  16u /* game modes (4 bits in `cons`)                              */
#define MM_MAX_ROLES  8u
#define MM_MAX_TEAMS  4u
#define MM_MAX_LOBBY  16u /* teams * team_size                                          */

/* ---- player constraint word (`cons`), one u32 per queued player -------------------
 * Restates the JSON fields of the player map the reference forwards untouched
 * (lib/search/worker.ex:292-300) as a bit-packed SoA column.  Bits above 19 must be 0. */
#define MM_CONS_MODE(c)   ((uint32_t)(c) & 0xFu)          /* "game-mode" (worker.ex:294)  */
#define MM_CONS_REGION(c) (((uint32_t)(c) >> 4) & 0xFFu)  /* region id                    */
#define MM_CONS_PARTY(c)  (((uint32_t)(c) >> 12) & 0xFu)  /* party size                   */
#define MM_CONS_ROLE(c)   (((uint32_t)(c) >> 16) & 0xFu)  /* role id                      */
#define MM_CONS_MAKE(mode, region, party, role)                                          \
    (((uint32_t)(mode) & 0xFu) | (((uint32_t)(region) & 0xFFu) << 4) |                  \
     (((uint32_t)(party) & 0xFu) << 12) | (((uint32_t)(role) & 0xFu) << 16))
#define MM_CONS_USER_MASK 0x000FFFFFu

/* mm_mode_config.flags */
#define MM_MODE_REGION_FILTER 1u /* new.region must equal anchor.region */
#define MM_MODE_PARTY_FILTER  2u /* new.party  must equal anchor.party  */

/* mm_config.flags */
#define MM_CFG_TIMING 1u /* record HIP-event timings of every kernel phase in mm_stats */

typedef enum mm_status {
    MM_OK = 0,
    MM_ERR_INVALID_ARG = -1,  /* NULL pointer, bad mode/group/role index, bad config    */
    MM_ERR_NO_DEVICE = -2,    /* no usable HIP device / library built without kernels   */
    MM_ERR_OOM = -3,          /* device or pinned-host allocation failed                */
    MM_ERR_FULL = -4,         /* pool capacity exhausted (slot ring has no free range)  */
    MM_ERR_HIP = -5,          /* a HIP runtime call failed; mm_last_hip_error() has it  */
    MM_ERR_INTERNAL = -6,     /* device-side invariant violated (reported, never abort) */
    MM_ERR_ABI = -7,          /* cfg->abi_version != MM_ABI_VERSION                     */
    MM_ERR_RANGE = -8,        /* first/count outside the last tick's match list         */
    MM_ERR_STATE = -9         /* an earlier mm_tick failed: only mm_reset / mm_restore /
                                 mm_engine_destroy are accepted until one succeeded     */
} mm_status;

/* Inclusive integer rating range — one row of `config :matchmaking, RatingGroups`
 * (config/config.exs:27-36). */
typedef struct mm_rating_group {
    int32_t from;
    int32_t to;
} mm_rating_group;

/* Per-game-mode parameters of the match-check predicate (docs/MATCH_CHECK.md).  This is
 * the state the external strategist (`strategist.match.check`, call site
 * lib/search/worker.ex:296-306) would hold for a mode. */
typedef struct mm_mode_config {
    uint32_t team_size;               /* players per team, 1..8                         */
    uint32_t teams;                   /* 2..MM_MAX_TEAMS                                */
    uint32_t window;                  /* accept iff |r_new - r_anchor| <= window        */
    uint32_t flags;                   /* MM_MODE_*                                      */
    uint32_t n_roles;                 /* 1..MM_MAX_ROLES                                */
    uint8_t  role_quota[MM_MAX_ROLES];/* seats per role per team; sums to team_size     */
} mm_mode_config;

typedef struct mm_config {
    uint32_t abi_version;             /* MM_ABI_VERSION                                 */
    uint32_t n_groups;                /* 1..MM_MAX_GROUPS                               */
    mm_rating_group groups[MM_MAX_GROUPS];
    uint32_t default_group;           /* index used when no range contains the rating
                                         (generic/worker.ex:27: Enum.at(groups, div(n,2)+1)) */
    uint32_t n_modes;                 /* 1..MM_MAX_MODES                                */
    mm_mode_config modes[MM_MAX_MODES];
    uint32_t capacity;                /* max simultaneously queued players (slot ring)  */
    int32_t  device;                  /* HIP device ordinal                             */
    uint32_t flags;                   /* MM_CFG_*                                       */
} mm_config;

/* Per-tick counters (SURVEY.md §5 "metrics": returned through the ABI because the
 * reference only exposes queue depth via AMQP.Queue.status, search/worker.ex:326-334). */
typedef struct mm_stats {
    uint32_t pool_before;     /* queued players of this mode before the tick             */
    uint32_t pool_after;      /* still queued after (includes players seated in lobbies) */
    uint32_t matches;         /* lobbies emitted                                         */
    uint32_t players_matched; /* matches * teams * team_size                             */
    uint32_t passes_max;      /* max full queue rotations over the rating groups         */
    uint32_t chains;          /* independent (group, mode) chains walked                 */
    uint64_t pairs;           /* candidate-pair evaluations = match_check calls against a
                                 non-empty lobby (the unit of SURVEY.md §8(d))           */
    uint64_t scanned;         /* queue elements streamed, summed over passes             */
    float    walk_ms;         /* HIP-event time of the walk kernel (MM_CFG_TIMING)       */
    float    filter_ms;       /* HIP-event time of the liveness filter, 0 if not run     */
    float    copy_ms;         /* D2H of counters + match list                            */
    float    total_ms;        /* host wall time of mm_tick                               */
} mm_stats;

/* Timings of the last mm_enqueue*/
typedef struct mm_enqueue_stats {
    uint32_t accepted;        /* players placed into a (group, mode) queue               */
    uint32_t rejected;        /* bad mode / role: dropped, slot left unused              */
    float    bucket_ms;       /* HIP-event time of count+scan+scatter (MM_CFG_TIMING)    */
    float    total_ms;
} mm_enqueue_stats;

typedef struct mm_engine mm_engine; /* opaque; NIF resource payload */

/* ---- library-level -------------------------------------------------------------- */

uint32_t    mm_abi_version(void);
const char* mm_strerror(int status);

/* Fills *cfg with the reference's shipped configuration: the 7 rating groups of
 * config/config.exs:27-36, default_group = 4 ("diamond", generic/worker.ex:27), and one
 * mode (index 0) = 1v1, window 50, no filters.  capacity = 1<<20, device 0. */
int mm_config_default(mm_config* cfg);

/* Replaces Matchmaking.Generic.Worker.find_rating_group_by_rating/1
 * (lib/generic/worker.ex:46-53): first group, in table order, whose inclusive range
 * contains `rating`; cfg->default_group otherwise (also for NaN, the stand-in for a
 * non-number JSON value, which fails every `<=` under Erlang term order).  Pure host
 * function; the device bucketing kernel applies the same rule to int32 ratings. */
int mm_find_rating_group(const mm_config* cfg, double rating, uint32_t* group);

/* ---- engine lifetime ------------------------------------------------------------- */

/* Replaces Search.Worker.init/1 (lib/search/worker.ex:220-237) + LobbyState.init_store/0
 * (lib/models/lobby_state.ex:15-29): one queue and one open-lobby record per
 * (rating group, game mode), all device-resident.  Fails fast (like `{:error, :noconn}`,
 * worker.ex:225-228) instead of degrading. */
int  mm_engine_create(const mm_config* cfg, mm_engine** out);
void mm_engine_destroy(mm_engine* e); /* NULL-safe; NIF resource destructor */

/* ---- per-engine tuning (round 6) ---------------------------------------------------
 * How an engine walks — batch sizes, which persistent launch shapes it may use, the bounded waits,
 * the test hooks — never WHAT it computes: every setting gives the same lobbies in the same order
 * (tests/stress.py --fuzz-knobs draws them at random and compares with the oracle).  The reference's
 * counterpart is the per-worker Confex configuration read in Search.Worker.init/1
 * (lib/search/worker.ex:54-66, :220-237; config/config.exs:11-25): a property of ONE worker, not of
 * the node.  Until round 5 these were process-wide MM_* environment variables read at create, so two
 * engines of one BEAM node could not differ; now the environment only supplies the DEFAULTS
 * (mm_tuning_default), and the owner passes the record to mm_engine_create_ex.
 *
 * Size-versioned like mm_path_stats: the caller sets `size` to its sizeof(mm_tuning); fields it does
 * not know keep their defaults.  Every field is a uint32_t so that a binding can treat the record as
 * words and address fields by NAME (mm_tuning_set / mm_tuning_name) without mirroring the layout. */
typedef struct mm_tuning {
    uint32_t size;              /* in: sizeof(mm_tuning) of the caller                                          */
    /* general */
    uint32_t force_generic;     /* 1: every chain is walked by k_walk (the pair and team paths off)  MM_FORCE_GENERIC   [0]  */
    uint32_t debug;             /* 1: per-tick diagnostics on stderr, phase timers in the kernels    MM_PAIR_DEBUG      [0]  */
    uint32_t results_early;     /* 1: the match list leaves for the host while the walk still runs   MM_RESULTS_EARLY   [1]  */
    uint32_t results_tail;      /* 1: the tail of the match list by ONE kernel into pinned memory    MM_RESULTS_TAIL    [1]  */
    uint32_t look_poll;         /* 1: looks at the chains through a polled pinned word, not a copy   MM_LOOK_POLL       [0]  */
    uint32_t fail_tick;         /* test hook: the k-th mm_tick fails after its walk (0: never)        MM_DEBUG_FAIL_TICK [0]  */
    /* pair path (mm_pair.inc) */
    uint32_t pair_persist;      /* 1: several passes per launch (kp_rounds) where a chain fits one XCD MM_PAIR_PERSIST   [1]  */
    uint32_t pair_ptiles;       /* tiles of the longest chain a kp_rounds batch may have, 1..32       MM_PAIR_PTILES     [32] */
    uint32_t pair_pbatch;       /* passes per kp_rounds launch at most, 1..4096                       MM_PAIR_PBATCH     [48] */
    uint32_t pair_batch;        /* kp_round launches per look of the host at the chains, 1..4096      MM_PAIR_BATCH      [48] */
    uint32_t pair_ptimeout_us;  /* what a workgroup waits at a launch's first barrier (40x later), us MM_PAIR_PTIMEOUT_US [5000] */
    uint32_t pair_pinject;      /* test hook: a kp_rounds launch declares a stop at this iteration + 1 (0: never) MM_PAIR_PINJECT [0] */
    uint32_t pair_tiles_max;    /* tiles of the longest chain before the next tile length is taken   MM_PAIR_TILES      [40] */
    uint32_t pair_tile_fixed;   /* 1: every batch with the longest tile length                        MM_PAIR_TILE=max   [0]  */
    uint32_t pair_xcd;          /* 1: kp_round's workgroups mapped chain by chain onto the XCDs       MM_PAIR_XCD        [1]  */
    uint32_t pair_group_min;    /* tiles from which a batch runs with the second route level (0: never) MM_PAIR_GROUP    [64] */
    uint32_t pair_nxseg;        /* anchors per workgroup of kp_nx_init, a power of two in 64..2048 (0: by pool size) MM_PAIR_NXSEG [0] */
    uint32_t pair_nxstage;      /* entries kp_nx_init stages in LDS (0: all it has room for)          MM_PAIR_NXSTAGE    [0]  */
    uint32_t pair_tune;         /* PairParams.tune: diagnostic bits (mm_pair.inc)                     MM_PAIR_TUNE       [0]  */
    /* team path (mm_team.inc) */
    uint32_t team_batch;        /* passes launched per look of the host at the chains, 1..4096        MM_TEAM_BATCH      [16] */
    uint32_t team_f2;           /* passes of a tick that compose F with itself (0: none)              MM_TEAM_F2         [32] */
    uint32_t team_rebuild;      /* the role sub-queues are rebuilt every this many passes, 1..4096    MM_TEAM_REBUILD    [8]  */
    uint32_t team_emit_max;     /* emitter workgroups per chain and launch, 1..32                     MM_TEAM_EMIT_MAX   [32] */
    uint32_t team_split;        /* 1: the stored lobby's fill rides in kt_f's launch (passes with kt_f2) MM_TEAM_SPLIT    [1]  */
    uint32_t team_fwait;        /* polls a chaser waits for a chunk's flag before it looks itself     MM_TEAM_FWAIT      [16384] */
    uint32_t team_fix_max;      /* replacements per chunk above which it looks everything up again    MM_TEAM_FIXMAX     [0xFFFFFFFF] */
    uint32_t team_fix_t8;       /* replacements per wave above which a task gets 8 lanes, not 16      MM_TEAM_FIXT8      [10] */
    uint32_t team_fix_t4;       /* ... above which it gets 4                                          MM_TEAM_FIXT4      [64] */
    uint32_t team_pull_xcd;     /* 1: emitter workgroups on the chaser's XCD pull F through its L2    MM_TEAM_PULLX      [1]  */
    uint32_t team_nowait;       /* test hook: the flag of every n-th chunk never comes (0: off)       MM_TEAM_NOWAIT     [0]  */
    uint32_t team_late;         /* lobbies per pass at or under which kt_late takes over (0: never)   MM_TEAM_LATE       [6]  */
    uint32_t team_late0;        /* arrivals since a mode's last tick at or under which kt_late walks the tick from its first pass (0: never) MM_TEAM_LATE0 [512] */
    uint32_t team_cap;          /* sub-queue entries one thread of kt_f looks at per role, 1..4096    MM_TEAM_CAP        [512] */
} mm_tuning;

/* Fills *t (t->size = the caller's sizeof on entry) with the defaults: the built-in values in [brackets] above, each
 * overridden by its MM_* environment variable when that is set to a value inside the field's range (a value outside it,
 * or one that does not parse, is reported on stderr once per create and ignored — until round 5 it was silently a no-op). */
int mm_tuning_default(mm_tuning* t);
/* Sets one field by its name as spelled above.  MM_ERR_INVALID_ARG: no such field (or it lies beyond t->size);
 * MM_ERR_RANGE: the value is outside the field's range.  *t is untouched on error. */
int mm_tuning_set(mm_tuning* t, const char* name, uint32_t value);
/* The name of field number `index` (0 = the first after `size`), NULL past the last: lets a binding enumerate. */
const char* mm_tuning_name(uint32_t index);
/* mm_engine_create with a tuning record (NULL: mm_tuning_default's).  MM_ERR_RANGE for a field outside its range. */
int mm_engine_create_ex(const mm_config* cfg, const mm_tuning* tuning, mm_engine** out);
/* What the engine runs with (t->size in: the caller's sizeof; out: bytes filled). */
int mm_tuning_get(const mm_engine* e, mm_tuning* t);

/* Drops every queued player and open lobby (a fresh Mnesia + empty broker queues). */
int mm_reset(mm_engine* e);

/* ---- ingest ---------------------------------------------------------------------- */

/* Replaces the delivery side of Search.Worker (handle_info(:basic_deliver),
 * lib/search/worker.ex:352-358) together with the upstream bucketing hop
 * Generic.Worker.consume/4 (lib/generic/worker.ex:55-69) and ActiveUser.add_user/1
 * (lib/models/active_user.ex:46-55): appends n players, in array order, to the tails of
 * their (group, mode) queues.  Array order == FIFO order == `seq`.
 *   rating[i]  integer rating; cons[i] per MM_CONS_*; group[i] optional override of the
 *   rating group (NULL = derive from rating by mm_find_rating_group; the override exists for
 *   non-integer / missing ratings, which the host routes by A1 on the exact value).
 *   out_slot[i] receives the engine handle of player i (stable until it is matched or
 *   cancelled), or 0xFFFFFFFF if the player was rejected (mode not configured, role not
 *   seatable).  Handles are the next n FREE slots of a ring of `capacity` slots, in ring
 *   order: a player that waits for hours keeps its slot and later batches step over it
 *   (MM_ERR_FULL only when fewer than n slots are free in the whole pool).
 *   Host pointers; the engine copies before returning. */
int mm_enqueue(mm_engine* e, uint32_t n, const int32_t* rating, const uint32_t* cons,
               const uint8_t* group, uint32_t* out_slot, mm_enqueue_stats* st);

/* Same, with rating/cons already resident in device memory (the benchmark path, and a
 * GPU-side codec's hand-off).  Slots are first_slot + i (mod capacity): this entry point needs
 * that whole range free and returns MM_ERR_FULL otherwise (it does not step over waiting
 * players — a service with long-waiting players ingests through mm_enqueue).  A rejected
 * player (mode not configured, role not seatable; st->rejected counts them) leaves its slot of
 * the range FREE: it is in no queue, mm_cancel ignores it, and a later batch takes the slot. */
int mm_enqueue_device(mm_engine* e, uint32_t n, const int32_t* d_rating,
                      const uint32_t* d_cons, uint32_t* first_slot, mm_enqueue_stats* st);

/* Replaces ActiveUser.remove_user/1 (lib/models/active_user.ex:57-66) as observed by the
 * search loop through ActiveUser.in_queue?/1 (active_user.ex:33-44; uses at
 * search/worker.ex:308 and :272): the players stop being "in queue".  Takes effect at
 * the start of the next tick: queued ones vanish when popped, seated ones are filtered
 * out of their open lobby (remove_inactive_players/1, search/worker.ex:267-280).
 * Unknown / already matched slots are ignored. */
int mm_cancel(mm_engine* e, uint32_t n, const uint32_t* slot);

/* ---- search ---------------------------------------------------------------------- */

/* Replaces Search.Worker.consume/5 (lib/search/worker.ex:291-324) run to quiescence under
 * the canonical schedule of SURVEY.md §3.4 ("Mode R"), for every rating group of `mode`:
 * FIFO first-fit seeding of the single open lobby per (group, mode)
 * (LobbyState.get_state/update_state, lib/models/lobby_state.ex:61-131), rejected players
 * to the tail (requeue_player/5, worker.ex:239-248 -> lib/requeue/worker.ex:51-54),
 * lobby emission in publish order (worker.ex:313-319).  Blocks until the device is done
 * (dirty-NIF: ERL_NIF_DIRTY_JOB_CPU_BOUND).  *n_matches = lobbies emitted; they stay
 * readable through mm_matches() until the next mm_tick/mm_reset on this engine. */
int mm_tick(mm_engine* e, uint32_t mode, uint32_t* n_matches, mm_stats* stats);

/* Copies matches [first, first+count) of the last tick, in emission order (rating group
 * major — groups are independent workers in the reference, lib/application.ex:26-40 —
 * then publish order within the group).  Per match: L = teams*team_size slots in team
 * order ("team 1" players in seating order, then "team 2", ... — the `"teams"` map of
 * worker.ex:315-318), score = |sum(team max) - sum(team min)| / team_size as f32 (for 1v1:
 * |r1 - r2|), the rating group index, and the pass (queue rotation) it was emitted in.
 * Any output pointer may be NULL. */
int mm_matches(mm_engine* e, uint32_t first, uint32_t count, uint32_t* slots, float* score,
               uint32_t* group, uint32_t* pass);

/* Replaces Search.Worker.status/0 (lib/search/worker.ex:115-117, :326-334 ->
 * AMQP.Queue.status message_count): per-group queue length for `mode`, not counting
 * players seated in the open lobby. per_group has cfg.n_groups entries. */
int mm_queue_depth(mm_engine* e, uint32_t mode, uint32_t* per_group);

/* The queue of (mode, group), head first: the deliveries still waiting in
 * `matchmaking.queues.<group>` in the order the broker would hand them out — arrival order,
 * rotated by the requeues of rejected players (requeue_player/5, lib/search/worker.ex:239-248 ->
 * Requeue.Worker.consume/4, lib/requeue/worker.ex:51-54: a rejected player re-enters at the TAIL).
 * The reference can only see the depth (AMQP.Queue.status); the engine exposes the order so that
 * requeue ordering can be checked directly after every tick (and so that an operator can see
 * who is waiting).  In: *n = capacity of `slots` (entries); out: *n = queue length; at most
 * min(capacity, length) handles are written.  `slots` may be NULL to ask for the length. */
int mm_queue_slots(mm_engine* e, uint32_t mode, uint32_t group, uint32_t* n, uint32_t* slots);

/* The open lobby of (mode, group) — the record LobbyState.get_state/4 would pop
 * (lib/models/lobby_state.ex:61-104).  Writes up to MM_MAX_LOBBY (slot, team) pairs in
 * team order; *n = seated players. */
int mm_lobby_state(mm_engine* e, uint32_t mode, uint32_t group, uint32_t* n,
                   uint32_t* slots, uint8_t* teams);

/* ---- pool snapshot -------------------------------------
 * The reference keeps everything this engine holds in RAM-only tables and unacked broker
 * deliveries: `LobbyState` is `ram_copies` (lib/models/lobby_state.ex:19-26), `ActiveUser`
 * likewise (lib/models/active_user.ex:15-29), the queues live in the broker.  A restart of
 * the search stage therefore rebuilds its state by redelivery.  With the engine in between,
 * a restart of the BEAM node (or a move to another GPU) dumps and reloads the pool instead:
 * queues in order, stored lobbies, the ActiveUser mirror, the slot allocator.  The results
 * of the last tick (mm_matches) are not part of a snapshot.
 *
 * mm_snapshot_size: bytes mm_snapshot would write now.
 * mm_snapshot:      writes the snapshot into buf (cap bytes); *written = its size.
 *                   MM_ERR_RANGE if cap is too small.
 * mm_restore:       replaces the engine's whole state with the snapshot's.  The engine must
 *                   have been created with the same mm_config (groups, modes, capacity):
 *                   MM_ERR_INVALID_ARG otherwise, and for a damaged or truncated buffer
 *                   (checksummed) — the engine's state is untouched then.  A HIP failure
 *                   during the reload leaves the engine reset, never half restored. */
int mm_snapshot_size(mm_engine* e, uint64_t* bytes);
int mm_snapshot(mm_engine* e, void* buf, uint64_t cap, uint64_t* written);
int mm_restore(mm_engine* e, const void* buf, uint64_t bytes);

/* ---- which launch shapes the last tick took, and the fall-backs it met ------------
 * The walk has persistent launch shapes (several passes per launch) that depend on behaviour the hardware is
 * observed, not promised, to have; when an observation fails the engine falls back to one launch per pass —
 * results identical, the tick slower.  The reference has no counterpart (its worker is one process,
 * search/worker.ex:291-324); an operator of THIS engine needs to tell a regression from a fall-back, and
 * mm_stats is frozen (ABI version 1), hence a separate, size-versioned record: the caller sets `size` to its
 * sizeof(mm_path_stats), the engine fills at most that many bytes and stores what it filled.  Diagnostics
 * only — never control flow, never results. */
#define MM_PATH_GENERIC 1u /* k_walk took chains                                       */
#define MM_PATH_PAIR    2u /* the pair path (1v1 modes)                                 */
#define MM_PATH_TEAM    4u /* the team path                                             */
typedef struct mm_path_stats {
    uint32_t size;                 /* in: sizeof(mm_path_stats) of the caller; out: bytes filled */
    uint32_t mode;                 /* the mode of the last mm_tick (0xFFFFFFFF: no tick yet)     */
    uint32_t paths;                /* MM_PATH_* of the last tick                                  */
    uint32_t host_looks;           /* synchronous looks of the host at the chains in that tick    */
    /* pair path */
    uint32_t pair_rounds_launches; /* launches of several passes each (kp_rounds)                 */
    uint32_t pair_rounds_passes;   /* passes of the longest chain walked inside them              */
    uint32_t pair_round_launches;  /* launches of one pass each (kp_round)                        */
    uint32_t pair_tiled_passes;    /* passes of the longest chain on the tiled path, both kinds   */
    uint32_t pair_stops_timeout;   /* kp_rounds launches that gave up in that tick: a barrier waited too long */
    uint32_t pair_stops_xcd;       /* ... a chain's workgroups were not on one XCD                */
    uint32_t pair_stops_inject;    /* ... the test hook                                           */
    uint32_t pair_yields;          /* batches ended early for a compaction (no fall-back)         */
    uint32_t pair_persist_off;     /* 1: kp_rounds is off in this engine for good (two XCDs seen, or MM_PAIR_PERSIST=0) */
    uint32_t pair_cooldown;        /* batches for which kp_rounds still stays off after a stop    */
    uint32_t pair_stops_total;     /* stops since mm_engine_create                                */
    /* team path */
    uint32_t team_f_launches;      /* kt_f (+ kt_f2 + kt_chase) passes                            */
    uint32_t team_fc_launches;     /* kt_fc passes (kt_f, chase and emission in one launch)       */
    uint32_t team_late_launches;   /* kt_late launches (the last passes back to back)             */
    uint32_t team_build_launches;  /* kt_build launches                                           */
    uint32_t team_flags_late;      /* kt_fc: anchors whose chunk flag did not come in time in that tick (the chaser looked the lobby up itself) */
    uint32_t team_flags_late_total;/* ... since mm_engine_create                                  */
    /* the chain with the most passes of that tick (the tick's critical path), pair path */
    uint32_t crit_group;           /* its rating group                                            */
    uint32_t crit_passes;          /* its passes                                                  */
    uint32_t crit_rounds_passes;   /* ... of them inside kp_rounds launches                       */
    uint32_t crit_rounds_hops;     /* ... dependent route hops its walks took there (tile 1's walker counts them) */
    uint32_t crit_round_passes;    /* ... of them as kp_round launches                            */
    uint32_t crit_late_passes;     /* ... of them inside kp_late                                  */
    uint32_t crit_late_lobbies;    /* lobbies it emitted inside kp_late (one dependent LDS step each) */
    uint32_t degraded;             /* 1: a fall-back was in force during that tick (a stop, the cool-down, kp_rounds off, late flags):
                                         its timing is not the headline path's                    */
    /* ---- round 6 (appended: a caller built against the shorter record gets the shorter record) ----
     * the primitives of the pair path's serial chain as tile 1's walker of the critical chain timed them in that tick */
    uint32_t crit_timed_passes;    /* passes inside kp_rounds it timed                                                     */
    uint32_t crit_timed_hops;      /* route hops it took in them                                                           */
    uint32_t crit_barrier_cycles;  /* shader-clock cycles it spent at the chain's flag barrier, summed over those passes   */
    uint32_t crit_hop_cycles;      /* ... in the scalar hop loop (L2-hit hops)                                             */
    uint32_t clk_cycles;           /* a stretch of kp_late (the critical chain's chase) in shader-clock cycles ...         */
    uint32_t clk_wall_ticks;       /* ... and in ticks of the constant 100 MHz counter: clk_cycles / clk_wall_ticks x 100 MHz = the clock */
    uint32_t pair_nx_init_ns;      /* kp_nx_init (next[] for everybody: the LDS-staged masked arg-min), HIP events (MM_CFG_TIMING; else 0) */
    uint32_t pair_tested_lo, pair_tested_hi;       /* mm_tuning.pair_tune bit 13: predicate tests the pair kernels physically performed in that tick ... */
    uint32_t pair_tested_nx_lo, pair_tested_nx_hi; /* ... of them in kp_nx_init (0 without the bit: the counters cost atomics)                         */
    /* the team path's chain with the most passes: what its chaser did, by launch shape */
    uint32_t crit_team_group;      /* its rating group (0xFFFFFFFF: the team path walked nothing)                          */
    uint32_t crit_team_passes;
    uint32_t crit_team_f_passes;   /* passes as kt_f | kt_f2 | kt_chase                                                    */
    uint32_t crit_team_fc_passes;  /* passes as ONE kt_fc launch                                                           */
    uint32_t crit_team_late_passes;/* passes inside kt_late                                                                */
    uint32_t crit_team_f_lobbies;  /* lobbies its chaser reached by F o F hops (two a trip) ...                            */
    uint32_t crit_team_fc_lobbies; /* ... by single F hops beside kt_f's chunks ...                                        */
    uint32_t crit_team_late_lobbies;/* ... by its own look at the anchor's record inside kt_late                           */
    uint32_t crit_team_lookups;    /* look-ups the chaser did itself in the pass kernels (stored lobby's fill, the lobby a pass ends on, anchors without F) */
    uint32_t crit_team_late_lookups;/* ... inside kt_late                                                                  */
} mm_path_stats;
int mm_path_stats_get(mm_engine* e, mm_path_stats* out);

/* Last HIP error code seen by this engine (0 if none) — for logs, never for control flow. */
int mm_last_hip_error(const mm_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* MM_ENGINE_H */
