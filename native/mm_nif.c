/*
 * mm_nif.c — the dirty-NIF shim between the BEAM and libmm_engine.so (include/mm_engine.h,
 * include/mm_codec.h).  This is the reference-side binding of INTEGRATION.md: the file a
 * maintainer of OpenMatchmaking/microservice-matchmaking drops into `native/` (built by
 * `elixir_make` or a three-line Makefile into `priv/mm_nif.so`) so that the Elixir module
 * `Matchmaking.Search.Engine` (native/elixir/search_engine.ex) can replace
 * `Matchmaking.Search.Worker.consume/5` (lib/search/worker.ex:291-324) and the strategist RPC
 * it makes (worker.ex:296-306).
 *
 * No logic lives here: every function unpacks binaries, calls ONE export of the C ABI and
 * packs the result.  Columns cross the boundary as little-endian binaries (`<<r::little-32>>`),
 * never as lists — a 1M-player batch is three binaries, not a million terms.
 *
 *   build:  cc -O2 -fPIC -shared -I$ERL_INCLUDE -I../include mm_nif.c -L<dir> -lmm_engine -o mm_nif.so
 *
 * The BEAM (and erl_nif.h) is absent from the image this repo is developed in; the file is
 * compiled and exercised there against tests/nif/ (a small stand-in for the enif_* calls used
 * below — test infrastructure, not shipped).  Only documented erl_nif API is used.
 *
 * Scheduling: everything that can touch the device is a dirty NIF.  mm_tick blocks on
 * hipStreamSynchronize -> ERL_NIF_DIRTY_JOB_CPU_BOUND; the ingest and
 * copy calls are ERL_NIF_DIRTY_JOB_IO_BOUND.  One owner process per engine (mm_engine.h):
 * the resource carries no lock.
 */
#include <erl_nif.h>
#include <string.h>

#include "mm_codec.h"
#include "mm_engine.h"

typedef struct {
    mm_engine* e;   /* NULL after close/1 */
    mm_config cfg;  /* copy: lobby size per mode, n_groups */
} nif_engine;

static ErlNifResourceType* ENGINE_T;
static ERL_NIF_TERM A_OK, A_ERROR, A_NIL;

static void engine_dtor(ErlNifEnv* env, void* p) {
    (void)env;
    nif_engine* r = (nif_engine*)p;
    mm_engine_destroy(r->e); /* NULL-safe */
    r->e = NULL;
}

/* {:error, {code, "text"}} — the owner logs it and decides; nothing here raises except badarg */
static ERL_NIF_TERM err(ErlNifEnv* env, int rc) {
    return enif_make_tuple2(env, A_ERROR,
                            enif_make_tuple2(env, enif_make_int(env, rc),
                                             enif_make_string(env, mm_strerror(rc), ERL_NIF_LATIN1)));
}

static int get_engine(ErlNifEnv* env, ERL_NIF_TERM t, nif_engine** r) {
    return enif_get_resource(env, t, ENGINE_T, (void**)r) && (*r)->e != NULL;
}

static int get_cstr_fwd(ErlNifEnv* env, ERL_NIF_TERM t, char* store, size_t cap, const char** out);   /* get_cstr, below */

static int get_config(ErlNifEnv* env, ERL_NIF_TERM t, const mm_config** cfg) {
    ErlNifBinary b;
    if (!enif_inspect_binary(env, t, &b) || b.size != sizeof(mm_config)) return 0;
    *cfg = (const mm_config*)b.data;
    return 1;
}

/* default_config() -> binary            mm_config_default: the shipped config/config.exs:27-36 */
static ERL_NIF_TERM nif_default_config(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc; (void)argv;
    ERL_NIF_TERM out;
    mm_config* cfg = (mm_config*)enif_make_new_binary(env, sizeof(mm_config), &out);
    int rc = mm_config_default(cfg);
    return rc ? err(env, rc) : out;
}

/* find_rating_group(config, rating :: float) -> {:ok, index}     generic/worker.ex:46-53 */
static ERL_NIF_TERM nif_find_rating_group(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    const mm_config* cfg; double r; long li; uint32_t g;
    if (!get_config(env, argv[0], &cfg)) return enif_make_badarg(env);
    if (!enif_get_double(env, argv[1], &r)) {
        if (!enif_get_long(env, argv[1], &li)) return enif_make_badarg(env);
        r = (double)li;
    }
    int rc = mm_find_rating_group(cfg, r, &g);
    return rc ? err(env, rc) : enif_make_tuple2(env, A_OK, enif_make_uint(env, g));
}

/* create(config) -> {:ok, engine}        search/worker.ex:220-237 + lobby_state.ex:15-29 */
static ERL_NIF_TERM nif_create(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    const mm_config* cfg; mm_engine* e = NULL;
    if (!get_config(env, argv[0], &cfg)) return enif_make_badarg(env);
    int rc = mm_engine_create(cfg, &e);
    if (rc) return err(env, rc);
    nif_engine* r = (nif_engine*)enif_alloc_resource(ENGINE_T, sizeof(*r));
    if (!r) { mm_engine_destroy(e); return err(env, MM_ERR_OOM); }
    r->e = e;
    r->cfg = *cfg;
    ERL_NIF_TERM t = enif_make_resource(env, r);
    enif_release_resource(r); /* the term owns it now; engine_dtor runs at GC */
    return enif_make_tuple2(env, A_OK, t);
}

/* create(config, [field_name], values) -> {:ok, engine}     mm_engine_create_ex: the engine with ITS tuning —
 * field_name = a field of include/mm_engine.h's mm_tuning as a binary ("team_late"), values = <<v::little-32, ...>>, one per
 * name; fields not named keep their defaults (built-in, or the MM_* environment variable's).  The per-worker configuration of
 * search/worker.ex:54-66.  {:error, {-1, _}}: no such field; {:error, {-8, _}}: a value outside its field's range. */
static ERL_NIF_TERM nif_create_tuned(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    const mm_config* cfg; mm_engine* e = NULL; ErlNifBinary vals; unsigned n = 0;
    if (!get_config(env, argv[0], &cfg) || !enif_get_list_length(env, argv[1], &n) ||
        !enif_inspect_binary(env, argv[2], &vals) || vals.size != (size_t)n * 4)
        return enif_make_badarg(env);
    mm_tuning tn;
    tn.size = (uint32_t)sizeof tn;
    int rc = mm_tuning_default(&tn);
    ERL_NIF_TERM list = argv[1], head;
    for (unsigned i = 0; i < n && !rc; i++) {
        char name[64]; const char* nm; uint32_t v;
        if (!enif_get_list_cell(env, list, &head, &list) || !get_cstr_fwd(env, head, name, sizeof name, &nm) || nm == NULL)
            return enif_make_badarg(env);
        memcpy(&v, vals.data + (size_t)i * 4, 4);
        rc = mm_tuning_set(&tn, nm, v);
    }
    if (!rc) rc = mm_engine_create_ex(cfg, &tn, &e);
    if (rc) return err(env, rc);
    nif_engine* r = (nif_engine*)enif_alloc_resource(ENGINE_T, sizeof(*r));
    if (!r) { mm_engine_destroy(e); return err(env, MM_ERR_OOM); }
    r->e = e;
    r->cfg = *cfg;
    ERL_NIF_TERM t = enif_make_resource(env, r);
    enif_release_resource(r);
    return enif_make_tuple2(env, A_OK, t);
}

/* close(engine) -> :ok      eager destroy (terminate/2); later calls on the handle are badarg */
static ERL_NIF_TERM nif_close(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    nif_engine* r;
    if (!enif_get_resource(env, argv[0], ENGINE_T, (void**)&r)) return enif_make_badarg(env);
    mm_engine_destroy(r->e);
    r->e = NULL;
    return A_OK;
}

/* reset(engine) -> :ok */
static ERL_NIF_TERM nif_reset(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    nif_engine* r;
    if (!get_engine(env, argv[0], &r)) return enif_make_badarg(env);
    int rc = mm_reset(r->e);
    return rc ? err(env, rc) : A_OK;
}

/* enqueue(engine, ratings, cons, groups | <<>>) -> {:ok, slots, accepted, rejected}
 * ratings = <<r::little-signed-32, ...>>, cons = <<c::little-32, ...>>, groups = <<g::8, ...>>
 * (the override column of mm_enqueue; empty = derive from the rating).  search/worker.ex:352-358 */
static ERL_NIF_TERM nif_enqueue(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    nif_engine* r; ErlNifBinary ra, co, gr;
    if (!get_engine(env, argv[0], &r) || !enif_inspect_binary(env, argv[1], &ra) ||
        !enif_inspect_binary(env, argv[2], &co) || !enif_inspect_binary(env, argv[3], &gr) ||
        (ra.size & 3) || ra.size != co.size || (gr.size != 0 && gr.size != ra.size / 4) ||
        ra.size / 4 > 0xFFFFFFFFu)
        return enif_make_badarg(env);
    uint32_t n = (uint32_t)(ra.size / 4);
    ERL_NIF_TERM out;
    uint32_t* slots = (uint32_t*)enif_make_new_binary(env, (size_t)n * 4, &out);
    mm_enqueue_stats st;
    memset(&st, 0, sizeof st);
    int rc = mm_enqueue(r->e, n, (const int32_t*)ra.data, (const uint32_t*)co.data,
                        gr.size ? (const uint8_t*)gr.data : NULL, slots, &st);
    if (rc) return err(env, rc);
    return enif_make_tuple4(env, A_OK, out, enif_make_uint(env, st.accepted), enif_make_uint(env, st.rejected));
}

/* cancel(engine, slots) -> :ok           active_user.ex:57-66 as read at worker.ex:308, :272 */
static ERL_NIF_TERM nif_cancel(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    nif_engine* r; ErlNifBinary sl;
    if (!get_engine(env, argv[0], &r) || !enif_inspect_binary(env, argv[1], &sl) || (sl.size & 3))
        return enif_make_badarg(env);
    int rc = mm_cancel(r->e, (uint32_t)(sl.size / 4), (const uint32_t*)sl.data);
    return rc ? err(env, rc) : A_OK;
}

/* tick(engine, mode) -> {:ok, n, lobby_size, slots, scores, groups, {pool_before, pool_after, pairs}}
 * slots  = n * lobby_size little-32 handles, team major ("team 1" first; worker.ex:315-318)
 * scores = n little-float-32, groups = n little-32 rating-group indices; publish order.
 * worker.ex:291-324 to quiescence. */
static ERL_NIF_TERM nif_tick(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    nif_engine* r; unsigned mode; uint32_t n = 0; mm_stats st;
    if (!get_engine(env, argv[0], &r) || !enif_get_uint(env, argv[1], &mode) || mode >= r->cfg.n_modes)
        return enif_make_badarg(env);
    memset(&st, 0, sizeof st);
    int rc = mm_tick(r->e, mode, &n, &st);
    if (rc) return err(env, rc);
    unsigned L = r->cfg.modes[mode].teams * r->cfg.modes[mode].team_size;
    ERL_NIF_TERM s, sc, g;
    uint32_t* ps = (uint32_t*)enif_make_new_binary(env, (size_t)n * L * 4, &s);
    float* pf = (float*)enif_make_new_binary(env, (size_t)n * 4, &sc);
    uint32_t* pg = (uint32_t*)enif_make_new_binary(env, (size_t)n * 4, &g);
    if (n) {
        rc = mm_matches(r->e, 0, n, ps, pf, pg, NULL);
        if (rc) return err(env, rc);
    }
    ERL_NIF_TERM stats = enif_make_tuple3(env, enif_make_uint(env, st.pool_before),
                                          enif_make_uint(env, st.pool_after), enif_make_uint64(env, st.pairs));
    return enif_make_tuple7(env, A_OK, enif_make_uint(env, n), enif_make_uint(env, L), s, sc, g, stats);
}

/* queue_depth(engine, mode) -> {:ok, <<depth::little-32, ...>>}   worker.ex:115-117, :326-334 */
static ERL_NIF_TERM nif_queue_depth(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    nif_engine* r; unsigned mode; ERL_NIF_TERM out;
    if (!get_engine(env, argv[0], &r) || !enif_get_uint(env, argv[1], &mode)) return enif_make_badarg(env);
    uint32_t* d = (uint32_t*)enif_make_new_binary(env, (size_t)r->cfg.n_groups * 4, &out);
    int rc = mm_queue_depth(r->e, mode, d);
    return rc ? err(env, rc) : enif_make_tuple2(env, A_OK, out);
}

/* path_stats(engine) -> {:ok, <<field::little-32, ...>>}   mm_path_stats_get: the launch shapes and fall-backs of the last
 * tick, the fields of include/mm_engine.h's mm_path_stats in order (its leading `size` included).  Host state only, no
 * device call: a plain NIF.  The reference has nothing to put beside it (search/worker.ex:115-117 reports the queue only). */
static ERL_NIF_TERM nif_path_stats(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    nif_engine* r; ERL_NIF_TERM out;
    if (!get_engine(env, argv[0], &r)) return enif_make_badarg(env);
    mm_path_stats* ps = (mm_path_stats*)enif_make_new_binary(env, sizeof(mm_path_stats), &out);
    ps->size = (uint32_t)sizeof(mm_path_stats);
    int rc = mm_path_stats_get(r->e, ps);
    return rc ? err(env, rc) : enif_make_tuple2(env, A_OK, out);
}

/* queue_slots(engine, mode, group) -> {:ok, <<slot::little-32, ...>>}   head first: the order the
 * broker would deliver in, requeues at the tail (worker.ex:239-248, requeue/worker.ex:51-54) */
static ERL_NIF_TERM nif_queue_slots(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    nif_engine* r; unsigned mode, group; uint32_t n = 0; ERL_NIF_TERM out;
    if (!get_engine(env, argv[0], &r) || !enif_get_uint(env, argv[1], &mode) || !enif_get_uint(env, argv[2], &group))
        return enif_make_badarg(env);
    int rc = mm_queue_slots(r->e, mode, group, &n, NULL);   /* length query */
    if (rc) return err(env, rc);
    uint32_t cap = n;
    uint32_t* d = (uint32_t*)enif_make_new_binary(env, (size_t)cap * 4, &out);
    rc = mm_queue_slots(r->e, mode, group, &n, d);          /* one owner: the queue cannot have moved */
    return rc ? err(env, rc) : enif_make_tuple2(env, A_OK, out);
}

/* lobby_state(engine, mode, group) -> {:ok, slots, teams}          lobby_state.ex:61-104 */
static ERL_NIF_TERM nif_lobby_state(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    nif_engine* r; unsigned mode, group; uint32_t n = 0;
    uint32_t slots[MM_MAX_LOBBY]; uint8_t teams[MM_MAX_LOBBY];
    if (!get_engine(env, argv[0], &r) || !enif_get_uint(env, argv[1], &mode) || !enif_get_uint(env, argv[2], &group))
        return enif_make_badarg(env);
    int rc = mm_lobby_state(r->e, mode, group, &n, slots, teams);
    if (rc) return err(env, rc);
    ERL_NIF_TERM s, t;
    memcpy(enif_make_new_binary(env, (size_t)n * 4, &s), slots, (size_t)n * 4);
    memcpy(enif_make_new_binary(env, n, &t), teams, n);
    return enif_make_tuple3(env, A_OK, s, t);
}

/* snapshot(engine) -> {:ok, binary}; restore(engine, binary) -> :ok     SURVEY.md 8(f) row 4 */
static ERL_NIF_TERM nif_snapshot(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    nif_engine* r; uint64_t bytes = 0, written = 0; ErlNifBinary b;
    if (!get_engine(env, argv[0], &r)) return enif_make_badarg(env);
    int rc = mm_snapshot_size(r->e, &bytes);
    if (rc) return err(env, rc);
    if (!enif_alloc_binary((size_t)bytes, &b)) return err(env, MM_ERR_OOM); /* off-heap: snapshots are large */
    rc = mm_snapshot(r->e, b.data, bytes, &written);
    if (rc) { enif_release_binary(&b); return err(env, rc); }
    if (written != bytes && !enif_realloc_binary(&b, (size_t)written)) { enif_release_binary(&b); return err(env, MM_ERR_OOM); }
    return enif_make_tuple2(env, A_OK, enif_make_binary(env, &b));
}

static ERL_NIF_TERM nif_restore(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    nif_engine* r; ErlNifBinary b;
    if (!get_engine(env, argv[0], &r) || !enif_inspect_binary(env, argv[1], &b)) return enif_make_badarg(env);
    int rc = mm_restore(r->e, b.data, b.size);
    return rc ? err(env, rc) : A_OK;
}

/* A key of the codec: binary -> NUL-terminated copy in `store`; the atom nil -> NULL. */
static int get_cstr(ErlNifEnv* env, ERL_NIF_TERM t, char* store, size_t cap, const char** out) {
    ErlNifBinary b;
    if (enif_is_identical(t, A_NIL)) { *out = NULL; return 1; }
    if (!enif_inspect_binary(env, t, &b) || b.size >= cap || memchr(b.data, 0, b.size)) return 0;
    memcpy(store, b.data, b.size);
    store[b.size] = 0;
    *out = store;
    return 1;
}

static int get_cstr_fwd(ErlNifEnv* env, ERL_NIF_TERM t, char* store, size_t cap, const char** out) {
    return get_cstr(env, t, store, cap, out);
}

/* decode(config, [mode_name], region_key | nil, party_key | nil, role_key | nil, payloads, offsets)
 *   -> {:ok, ratings, cons, groups, status, id_offsets, id_lengths}
 * payloads = the deliveries of a tick period back to back, offsets = n+1 little-64 byte offsets.
 * generic/worker.ex:55-57 + :46-53 and search/worker.ex:292-294 for the whole batch. */
#define NIF_KEY_MAX 128
static ERL_NIF_TERM nif_decode(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    const mm_config* cfg; mm_codec_cfg cc; ErlNifBinary buf, off;
    static const size_t KEYS = MM_MAX_MODES + 3;
    memset(&cc, 0, sizeof cc);
    if (!get_config(env, argv[0], &cfg) || !enif_inspect_binary(env, argv[5], &buf) ||
        !enif_inspect_binary(env, argv[6], &off) || off.size < 8 || (off.size & 7))
        return enif_make_badarg(env);
    char* names = (char*)enif_alloc(KEYS * NIF_KEY_MAX);
    if (!names) return err(env, MM_ERR_OOM);
    ERL_NIF_TERM list = argv[1], head, ret;
    int ok = 1;
    while (ok && enif_get_list_cell(env, list, &head, &list)) {
        ok = cc.n_modes < MM_MAX_MODES &&
             get_cstr(env, head, names + (size_t)cc.n_modes * NIF_KEY_MAX, NIF_KEY_MAX, &cc.mode_name[cc.n_modes]) &&
             cc.mode_name[cc.n_modes] != NULL;
        cc.n_modes += ok;
    }
    ok = ok && enif_is_empty_list(env, list) &&
         get_cstr(env, argv[2], names + (MM_MAX_MODES + 0) * NIF_KEY_MAX, NIF_KEY_MAX, &cc.region_key) &&
         get_cstr(env, argv[3], names + (MM_MAX_MODES + 1) * NIF_KEY_MAX, NIF_KEY_MAX, &cc.party_key) &&
         get_cstr(env, argv[4], names + (MM_MAX_MODES + 2) * NIF_KEY_MAX, NIF_KEY_MAX, &cc.role_key);
    uint64_t n64 = off.size / 8 - 1;
    const uint64_t* o = (const uint64_t*)off.data;
    ok = ok && n64 <= 0xFFFFFFFFu && o[n64] <= buf.size;
    if (!ok) {
        ret = enif_make_badarg(env);
    } else {
        uint32_t n = (uint32_t)n64;
        ERL_NIF_TERM t[6];
        int32_t* ra = (int32_t*)enif_make_new_binary(env, (size_t)n * 4, &t[0]);
        uint32_t* co = (uint32_t*)enif_make_new_binary(env, (size_t)n * 4, &t[1]);
        uint8_t* gr = (uint8_t*)enif_make_new_binary(env, n, &t[2]);
        uint8_t* st = (uint8_t*)enif_make_new_binary(env, n, &t[3]);
        uint32_t* io = (uint32_t*)enif_make_new_binary(env, (size_t)n * 4, &t[4]);
        uint32_t* il = (uint32_t*)enif_make_new_binary(env, (size_t)n * 4, &t[5]);
        int rc = mm_decode_players(cfg, &cc, (const char*)buf.data, o, n, ra, co, gr, st, io, il);
        ret = rc ? err(env, rc) : enif_make_tuple7(env, A_OK, t[0], t[1], t[2], t[3], t[4], t[5]);
    }
    enif_free(names);
    return ret;
}

/* encode_lobby(game_mode, teams, team_size, [payload]) -> {:ok, json}
 * payloads team major, as tick/2 lists the slots.  search/worker.ex:315-318 */
static ERL_NIF_TERM nif_encode_lobby(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
    (void)argc;
    char gm_store[NIF_KEY_MAX]; const char* gm; unsigned teams, team_size, len = 0;
    const char* pay[MM_MAX_LOBBY]; uint32_t plen[MM_MAX_LOBBY];
    if (!get_cstr(env, argv[0], gm_store, sizeof gm_store, &gm) || gm == NULL ||
        !enif_get_uint(env, argv[1], &teams) || !enif_get_uint(env, argv[2], &team_size) ||
        !enif_get_list_length(env, argv[3], &len) || len > MM_MAX_LOBBY || teams > MM_MAX_TEAMS ||
        (uint64_t)teams * team_size != len)
        return enif_make_badarg(env);
    ERL_NIF_TERM list = argv[3], head;
    for (unsigned i = 0; i < len; i++) {
        ErlNifBinary b;
        if (!enif_get_list_cell(env, list, &head, &list) || !enif_inspect_binary(env, head, &b) ||
            b.size > 0xFFFFFFFFu)
            return enif_make_badarg(env);
        pay[i] = (const char*)b.data;
        plen[i] = (uint32_t)b.size;
    }
    uint64_t need = 0;
    int rc = mm_encode_lobby(gm, teams, team_size, pay, plen, NULL, 0, &need); /* size query */
    if (rc && rc != MM_ERR_RANGE) return err(env, rc);
    ERL_NIF_TERM out;
    char* dst = (char*)enif_make_new_binary(env, (size_t)need, &out);
    rc = mm_encode_lobby(gm, teams, team_size, pay, plen, dst, need, &need);
    return rc ? err(env, rc) : enif_make_tuple2(env, A_OK, out);
}

static ErlNifFunc funcs[] = {
    {"default_config", 0, nif_default_config, 0},
    {"find_rating_group", 2, nif_find_rating_group, 0},
    {"create", 1, nif_create, ERL_NIF_DIRTY_JOB_IO_BOUND},       /* hipMalloc of the whole pool */
    {"create", 3, nif_create_tuned, ERL_NIF_DIRTY_JOB_IO_BOUND}, /* ... with the engine's own mm_tuning */
    {"close", 1, nif_close, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"reset", 1, nif_reset, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"enqueue", 4, nif_enqueue, ERL_NIF_DIRTY_JOB_IO_BOUND},     /* H2D + bucketing kernels */
    {"cancel", 2, nif_cancel, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"tick", 2, nif_tick, ERL_NIF_DIRTY_JOB_CPU_BOUND},          /* blocks on the stream until quiescence */
    {"queue_depth", 2, nif_queue_depth, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"path_stats", 1, nif_path_stats, 0},                        /* host state only */
    {"queue_slots", 3, nif_queue_slots, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"lobby_state", 3, nif_lobby_state, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"snapshot", 1, nif_snapshot, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"restore", 2, nif_restore, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"decode", 7, nif_decode, ERL_NIF_DIRTY_JOB_CPU_BOUND},      /* ~2 M messages/s per core */
    {"encode_lobby", 4, nif_encode_lobby, 0},                    /* a few microseconds */
};

static int load(ErlNifEnv* env, void** priv, ERL_NIF_TERM info) {
    (void)priv; (void)info;
    if (mm_abi_version() != MM_ABI_VERSION) return -1;
    ENGINE_T = enif_open_resource_type(env, NULL, "mm_engine", engine_dtor, ERL_NIF_RT_CREATE, NULL);
    A_OK = enif_make_atom(env, "ok");
    A_ERROR = enif_make_atom(env, "error");
    A_NIL = enif_make_atom(env, "nil");
    return ENGINE_T ? 0 : -1;
}

ERL_NIF_INIT(Elixir.Matchmaking.Search.Engine, funcs, load, NULL, NULL, NULL)
