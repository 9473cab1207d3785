/*
 * mode_r.c — CPU oracle ("Mode R").  TEST INFRASTRUCTURE ONLY — see mode_r.h.
 * PARITY UNPINNED by the reference (no golden vectors exist for this path); pinned by
 * tests/golden/ hand derivations and oracle/literal_ref.py.
 *
 * Reference lines each function follows (paths relative to /root/reference/matchmaking/):
 *   rating_group()   lib/generic/worker.ex:46-53, :27; config/config.exs:27-36
 *   lobby_anchor(), match_check()   docs/MATCH_CHECK.md §2 — stands in for the external
 *                    strategist called at lib/search/worker.ex:296-306
 *   chain_tick()     lib/search/worker.ex:291-324 under the schedule of SURVEY.md §3.4:
 *                    get_state/update_state lib/models/lobby_state.ex:61-131,
 *                    requeue lib/search/worker.ex:239-248 + lib/requeue/worker.ex:51-54,
 *                    remove_inactive_players lib/search/worker.ex:267-280,
 *                    emission lib/search/worker.ex:313-319
 *   mo_cancel()      lib/models/active_user.ex:57-66 as read through :33-44
 */
#include "mode_r.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define SLOT_FREE 0
#define SLOT_LIVE 1
#define SLOT_CANCELLED 2

typedef struct {
    uint32_t slot;
    int32_t rating;
    uint32_t cons;
} qrec;

typedef struct {
    uint32_t n;                                   /* seated players                      */
    uint32_t cnt[MM_MAX_TEAMS];                   /* per team                            */
    qrec seat[MM_MAX_TEAMS][8];                   /* per team, seating order             */
} lobby;

typedef struct {
    uint32_t slots[MM_MAX_LOBBY];
    float score;
    uint32_t group, pass;
} match_rec;

typedef struct {
    qrec* q;
    uint32_t len, cap;
    lobby lb;
    /* per-tick outputs */
    match_rec* out;
    uint32_t n_out, cap_out;
    uint32_t passes;
    uint64_t pairs, scanned;
} chain;

struct mo_engine {
    mm_config cfg;
    uint8_t* state;   /* per slot */
    uint32_t next_slot;
    uint32_t next_seq;
    int cancel_pending;
    chain* chains;    /* [mode * n_groups + group] */
    match_rec* last;  /* concatenated matches of the last tick */
    uint32_t n_last, last_L;
};

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* generic/worker.ex:46-53 — Enum.find(groups, default, from <= r <= to).  NaN models a
 * non-number JSON value: Erlang orders atoms after numbers, so `nil <= to` is false. */
static uint32_t rating_group(const mm_config* cfg, double rating)
{
    if (rating == rating) {
        for (uint32_t g = 0; g < cfg->n_groups; ++g)
            if (rating >= (double)cfg->groups[g].from && rating <= (double)cfg->groups[g].to)
                return g;
    }
    return cfg->default_group;
}

int mo_find_rating_group(const mm_config* cfg, double rating, uint32_t* group)
{
    if (!cfg || !group) return MM_ERR_INVALID_ARG;
    *group = rating_group(cfg, rating);
    return MM_OK;
}

static int cfg_valid(const mm_config* c)
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

int mo_engine_create(const mm_config* cfg, mo_engine** out)
{
    if (!cfg || !out) return MM_ERR_INVALID_ARG;
    int rc = cfg_valid(cfg);
    if (rc) return rc;
    mo_engine* e = (mo_engine*)calloc(1, sizeof(*e));
    if (!e) return MM_ERR_OOM;
    e->cfg = *cfg;
    e->state = (uint8_t*)calloc(cfg->capacity, 1);
    uint32_t nc = cfg->n_modes * cfg->n_groups;
    e->chains = (chain*)calloc(nc, sizeof(chain));
    if (!e->state || !e->chains) { mo_engine_destroy(e); return MM_ERR_OOM; }
    *out = e;
    return MM_OK;
}

void mo_engine_destroy(mo_engine* e)
{
    if (!e) return;
    if (e->chains) {
        uint32_t nc = e->cfg.n_modes * e->cfg.n_groups;
        for (uint32_t c = 0; c < nc; ++c) { free(e->chains[c].q); free(e->chains[c].out); }
    }
    free(e->chains);
    free(e->state);
    free(e->last);
    free(e);
}

int mo_reset(mo_engine* e)
{
    if (!e) return MM_ERR_INVALID_ARG;
    uint32_t nc = e->cfg.n_modes * e->cfg.n_groups;
    for (uint32_t c = 0; c < nc; ++c) {
        e->chains[c].len = 0;
        memset(&e->chains[c].lb, 0, sizeof(lobby));
        e->chains[c].n_out = 0;
    }
    memset(e->state, 0, e->cfg.capacity);
    e->next_slot = 0;
    e->next_seq = 0;
    e->cancel_pending = 0;
    e->n_last = 0;
    return MM_OK;
}

static int chain_push(chain* c, qrec r)
{
    if (c->len == c->cap) {
        uint32_t nc = c->cap ? c->cap * 2 : 1024;
        qrec* nq = (qrec*)realloc(c->q, (size_t)nc * sizeof(qrec));
        if (!nq) return MM_ERR_OOM;
        c->q = nq;
        c->cap = nc;
    }
    c->q[c->len++] = r;
    return MM_OK;
}

/* handle_info(:basic_deliver) search/worker.ex:352-358 after the bucketing hop
 * generic/worker.ex:55-69; ActiveUser.add_user active_user.ex:46-55. */
int mo_enqueue(mo_engine* e, uint32_t n, const int32_t* rating, const uint32_t* cons,
               const uint8_t* group, uint32_t* out_slot, mm_enqueue_stats* st)
{
    if (!e || (n && (!rating || !cons))) return MM_ERR_INVALID_ARG;
    double t0 = now_ms();
    const mm_config* cfg = &e->cfg;
    if (n > cfg->capacity) return MM_ERR_FULL;
    /* slot handles: the next n FREE slots in ring order from next_slot, stepping over slots whose
     * player is still waiting (include/mm_engine.h, mm_enqueue); MM_ERR_FULL only when the pool
     * has fewer than n free slots.  Handles are this engine's bookkeeping, not reference data. */
    uint32_t nfree = 0;
    for (uint32_t s = 0; s < cfg->capacity && nfree < n; ++s)
        if (e->state[s] == SLOT_FREE) ++nfree;
    if (nfree < n) return MM_ERR_FULL;
    if (group)
        for (uint32_t i = 0; i < n; ++i)
            if (group[i] >= cfg->n_groups) return MM_ERR_INVALID_ARG;
    uint32_t acc = 0, rej = 0, cursor = e->next_slot;
    for (uint32_t i = 0; i < n; ++i) {
        while (e->state[cursor] != SLOT_FREE) cursor = (cursor + 1u) % cfg->capacity;
        uint32_t slot = cursor;
        cursor = (cursor + 1u) % cfg->capacity;
        uint32_t c = cons[i] & MM_CONS_USER_MASK;
        uint32_t mode = MM_CONS_MODE(c), role = MM_CONS_ROLE(c);
        if (mode >= cfg->n_modes || role >= cfg->modes[mode].n_roles ||
            cfg->modes[mode].role_quota[role] == 0) {
            if (out_slot) out_slot[i] = 0xFFFFFFFFu;
            ++rej;
            continue;
        }
        uint32_t g = group ? group[i] : rating_group(cfg, (double)rating[i]);
        qrec r = { slot, rating[i], c };
        int rc = chain_push(&e->chains[mode * cfg->n_groups + g], r);
        if (rc) return rc;
        e->state[slot] = SLOT_LIVE;
        if (out_slot) out_slot[i] = slot;
        ++acc;
    }
    e->next_slot = cursor;
    e->next_seq += n;
    if (st) {
        memset(st, 0, sizeof(*st));
        st->accepted = acc;
        st->rejected = rej;
        st->total_ms = (float)(now_ms() - t0);
    }
    return MM_OK;
}

int mo_cancel(mo_engine* e, uint32_t n, const uint32_t* slot)
{
    if (!e || (n && !slot)) return MM_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n; ++i) {
        if (slot[i] >= e->cfg.capacity) continue;
        if (e->state[slot[i]] == SLOT_LIVE) {
            e->state[slot[i]] = SLOT_CANCELLED;
            e->cancel_pending++;
        }
    }
    return MM_OK;
}

/* docs/MATCH_CHECK.md §2.1 */
static const qrec* lobby_anchor(const lobby* lb, uint32_t teams)
{
    for (uint32_t t = 0; t < teams; ++t)
        if (lb->cnt[t]) return &lb->seat[t][0];
    return NULL;
}

/* docs/MATCH_CHECK.md §2.2-2.4.  Returns the team `p` was seated in, or -1 (rejected). */
static int match_check(const mm_mode_config* mc, lobby* lb, const qrec* p)
{
    const qrec* a = lobby_anchor(lb, mc->teams);
    if (a) {
        int64_t d = (int64_t)p->rating - (int64_t)a->rating;
        if (d < 0) d = -d;
        if (d > (int64_t)mc->window) return -1;
        if ((mc->flags & MM_MODE_REGION_FILTER) && MM_CONS_REGION(p->cons) != MM_CONS_REGION(a->cons))
            return -1;
        if ((mc->flags & MM_MODE_PARTY_FILTER) && MM_CONS_PARTY(p->cons) != MM_CONS_PARTY(a->cons))
            return -1;
    }
    uint32_t role = MM_CONS_ROLE(p->cons);
    int best = -1;
    int64_t best_sum = 0;
    for (uint32_t t = 0; t < mc->teams; ++t) {
        uint32_t have = 0;
        int64_t sum = 0;
        for (uint32_t k = 0; k < lb->cnt[t]; ++k) {
            have += MM_CONS_ROLE(lb->seat[t][k].cons) == role;
            sum += lb->seat[t][k].rating;
        }
        if (have >= mc->role_quota[role]) continue;
        if (best < 0 || sum < best_sum) { best = (int)t; best_sum = sum; }
    }
    if (best < 0) return -1;
    lb->seat[best][lb->cnt[best]++] = *p;
    lb->n++;
    return best;
}

static void lobby_filter(mo_engine* e, lobby* lb, uint32_t teams, uint32_t* released)
{
    for (uint32_t t = 0; t < teams; ++t) {
        uint32_t w = 0;
        for (uint32_t k = 0; k < lb->cnt[t]; ++k) {
            if (e->state[lb->seat[t][k].slot] == SLOT_LIVE) lb->seat[t][w++] = lb->seat[t][k];
            else { e->state[lb->seat[t][k].slot] = SLOT_FREE; ++*released; }
        }
        lb->n -= lb->cnt[t] - w;
        lb->cnt[t] = w;
    }
}

static int lobby_has_cancelled(const mo_engine* e, const lobby* lb, uint32_t teams)
{
    for (uint32_t t = 0; t < teams; ++t)
        for (uint32_t k = 0; k < lb->cnt[t]; ++k)
            if (e->state[lb->seat[t][k].slot] != SLOT_LIVE) return 1;
    return 0;
}

static int chain_emit(chain* c, const mm_mode_config* mc, uint32_t group, uint32_t pass)
{
    if (c->n_out == c->cap_out) {
        uint32_t nc = c->cap_out ? c->cap_out * 2 : 256;
        match_rec* no = (match_rec*)realloc(c->out, (size_t)nc * sizeof(match_rec));
        if (!no) return MM_ERR_OOM;
        c->out = no;
        c->cap_out = nc;
    }
    match_rec* m = &c->out[c->n_out++];
    int64_t smin = 0, smax = 0;
    uint32_t k = 0;
    for (uint32_t t = 0; t < mc->teams; ++t) {
        int64_t s = 0;
        for (uint32_t j = 0; j < c->lb.cnt[t]; ++j) {
            m->slots[k++] = c->lb.seat[t][j].slot;
            s += c->lb.seat[t][j].rating;
        }
        if (t == 0 || s < smin) smin = s;
        if (t == 0 || s > smax) smax = s;
    }
    m->score = (float)(int32_t)(smax - smin) / (float)(int32_t)mc->team_size;
    m->group = group;
    m->pass = pass;
    return MM_OK;
}

/* One chain run to quiescence: passes of consume/5 (search/worker.ex:291-324).
 *
 * `purge` = some mm_cancel is still pending.  Its deferred effect follows the order of
 * operations inside ONE attempt (worker.ex:295-321): match_check sees the lobby as
 * stored — the strategist cannot know who cancelled — and only afterwards
 * remove_inactive_players (worker.ex:312) filters it.  So:
 *   - queue empty: no attempt happens, the lobby stays as stored;
 *   - head of the queue cancelled: its attempt leaves no trace except the filter;
 *   - head alive and the lobby holds a cancelled seat: the head is evaluated against the
 *     STALE lobby (anchor, quotas and sums include the cancelled seats), then the lobby is
 *     filtered; the filter marks the lobby changed, so this attempt never emits (worker.ex:313).
 * Cancelled queue entries vanish when popped (worker.ex:308, :312). */
static int chain_tick(mo_engine* e, chain* c, const mm_mode_config* mc, uint32_t group, int purge,
                      uint32_t* released)
{
    const uint32_t L = mc->teams * mc->team_size;
    int first_stale = 0;
    c->n_out = 0;
    c->passes = 0;
    c->pairs = 0;
    c->scanned = 0;
    if (purge && c->len > 0) {
        if (e->state[c->q[0].slot] == SLOT_LIVE && lobby_has_cancelled(e, &c->lb, mc->teams))
            first_stale = 1;
        else
            lobby_filter(e, &c->lb, mc->teams, released);
        uint32_t w = 0;
        for (uint32_t i = 0; i < c->len; ++i) {
            if (e->state[c->q[i].slot] == SLOT_LIVE) c->q[w++] = c->q[i];
            else { e->state[c->q[i].slot] = SLOT_FREE; ++*released; }
        }
        c->len = w;
    }
    while (c->len > 0) {
        int changed = 0;
        uint32_t w = 0;
        const uint32_t pass = c->passes;
        c->scanned += c->len;
        for (uint32_t i = 0; i < c->len; ++i) {
            qrec p = c->q[i];                              /* pop the head            */
            if (c->lb.n) c->pairs++;                       /* RPC against an anchor   */
            int t = match_check(mc, &c->lb, &p);           /* worker.ex:296-306       */
            if (first_stale) {                             /* worker.ex:312 after a cancel */
                first_stale = 0;
                lobby_filter(e, &c->lb, mc->teams, released);
                if (t < 0) c->q[w++] = p; else changed = 1;
                continue;                                  /* changed lobby: saved, :320 */
            }
            if (t < 0) { c->q[w++] = p; continue; }        /* requeue, worker.ex:308  */
            changed = 1;
            if (c->lb.n == L) {                            /* is_filled, worker.ex:313 */
                int rc = chain_emit(c, mc, group, pass);
                if (rc) return rc;
                for (uint32_t tt = 0; tt < mc->teams; ++tt)
                    for (uint32_t k = 0; k < c->lb.cnt[tt]; ++k)
                        e->state[c->lb.seat[tt][k].slot] = SLOT_FREE;
                memset(&c->lb, 0, sizeof(lobby));
            }                                              /* else save, worker.ex:320 */
        }
        c->len = w;
        c->passes++;
        if (!changed) break;
    }
    return MM_OK;
}

typedef struct {
    mo_engine* e;
    uint32_t mode, g0, gstep;
    int rc, purge;
    uint32_t released;
} tick_job;

static void* tick_worker(void* arg)
{
    tick_job* j = (tick_job*)arg;
    const mm_config* cfg = &j->e->cfg;
    for (uint32_t g = j->g0; g < cfg->n_groups; g += j->gstep) {
        int rc = chain_tick(j->e, &j->e->chains[j->mode * cfg->n_groups + g], &cfg->modes[j->mode], g,
                            j->purge, &j->released);
        if (rc) j->rc = rc;
    }
    return NULL;
}

int mo_tick_threads(mo_engine* e, uint32_t mode, uint32_t n_threads, uint32_t* n_matches,
                    mm_stats* stats)
{
    if (!e || mode >= e->cfg.n_modes) return MM_ERR_INVALID_ARG;
    double t0 = now_ms();
    const mm_config* cfg = &e->cfg;
    const mm_mode_config* mc = &cfg->modes[mode];
    const uint32_t L = mc->teams * mc->team_size;
    uint32_t before = 0;
    for (uint32_t g = 0; g < cfg->n_groups; ++g) {
        chain* c = &e->chains[mode * cfg->n_groups + g];
        before += c->len + c->lb.n;
    }
    if (n_threads < 1) n_threads = 1;
    if (n_threads > cfg->n_groups) n_threads = cfg->n_groups;
    tick_job jobs[MM_MAX_GROUPS];
    pthread_t th[MM_MAX_GROUPS];
    for (uint32_t t = 0; t < n_threads; ++t) {
        jobs[t].e = e; jobs[t].mode = mode; jobs[t].g0 = t; jobs[t].gstep = n_threads; jobs[t].rc = 0;
        jobs[t].purge = e->cancel_pending > 0; jobs[t].released = 0;
    }
    if (n_threads == 1) {
        tick_worker(&jobs[0]);
    } else {
        for (uint32_t t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, tick_worker, &jobs[t]);
        for (uint32_t t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    }
    for (uint32_t t = 0; t < n_threads; ++t) {
        if (jobs[t].rc) return jobs[t].rc;
        e->cancel_pending -= (int)jobs[t].released;
    }
    uint32_t total = 0, after = 0, pmax = 0;
    uint64_t pairs = 0, scanned = 0;
    for (uint32_t g = 0; g < cfg->n_groups; ++g) {
        chain* c = &e->chains[mode * cfg->n_groups + g];
        total += c->n_out;
        after += c->len + c->lb.n;
        pairs += c->pairs;
        scanned += c->scanned;
        if (c->passes > pmax) pmax = c->passes;
    }
    free(e->last);
    e->last = (match_rec*)malloc((size_t)(total ? total : 1) * sizeof(match_rec));
    if (!e->last) return MM_ERR_OOM;
    uint32_t k = 0;
    for (uint32_t g = 0; g < cfg->n_groups; ++g) {   /* group-major emission order */
        chain* c = &e->chains[mode * cfg->n_groups + g];
        memcpy(e->last + k, c->out, (size_t)c->n_out * sizeof(match_rec));
        k += c->n_out;
    }
    e->n_last = total;
    e->last_L = L;
    if (n_matches) *n_matches = total;
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->pool_before = before;
        stats->pool_after = after;
        stats->matches = total;
        stats->players_matched = total * L;
        stats->passes_max = pmax;
        stats->chains = cfg->n_groups;
        stats->pairs = pairs;
        stats->scanned = scanned;
        stats->total_ms = (float)(now_ms() - t0);
        stats->walk_ms = stats->total_ms;
    }
    return MM_OK;
}

int mo_tick(mo_engine* e, uint32_t mode, uint32_t* n_matches, mm_stats* stats)
{
    return mo_tick_threads(e, mode, 1, n_matches, stats);
}

int mo_matches(mo_engine* e, uint32_t first, uint32_t count, uint32_t* slots, float* score,
               uint32_t* group, uint32_t* pass)
{
    if (!e) return MM_ERR_INVALID_ARG;
    if (first > e->n_last || count > e->n_last - first) return MM_ERR_RANGE;
    for (uint32_t i = 0; i < count; ++i) {
        const match_rec* m = &e->last[first + i];
        if (slots) memcpy(slots + (size_t)i * e->last_L, m->slots, e->last_L * sizeof(uint32_t));
        if (score) score[i] = m->score;
        if (group) group[i] = m->group;
        if (pass) pass[i] = m->pass;
    }
    return MM_OK;
}

int mo_queue_depth(mo_engine* e, uint32_t mode, uint32_t* per_group)
{
    if (!e || !per_group || mode >= e->cfg.n_modes) return MM_ERR_INVALID_ARG;
    for (uint32_t g = 0; g < e->cfg.n_groups; ++g) {
        /* cancelled-but-not-yet-purged entries are still in the broker queue */
        per_group[g] = e->chains[mode * e->cfg.n_groups + g].len;
    }
    return MM_OK;
}

int mo_queue_slots(mo_engine* e, uint32_t mode, uint32_t group, uint32_t* n, uint32_t* slots)
{
    if (!e || !n || mode >= e->cfg.n_modes || group >= e->cfg.n_groups) return MM_ERR_INVALID_ARG;
    chain* c = &e->chains[mode * e->cfg.n_groups + group];
    uint32_t k = c->len < *n ? c->len : *n;
    if (slots)
        for (uint32_t i = 0; i < k; ++i) slots[i] = c->q[i].slot;
    *n = c->len;
    return MM_OK;
}

int mo_lobby_state(mo_engine* e, uint32_t mode, uint32_t group, uint32_t* n, uint32_t* slots,
                   uint8_t* teams)
{
    if (!e || !n || mode >= e->cfg.n_modes || group >= e->cfg.n_groups) return MM_ERR_INVALID_ARG;
    const lobby* lb = &e->chains[mode * e->cfg.n_groups + group].lb;
    uint32_t k = 0;
    for (uint32_t t = 0; t < e->cfg.modes[mode].teams; ++t)
        for (uint32_t j = 0; j < lb->cnt[t]; ++j) {
            if (slots) slots[k] = lb->seat[t][j].slot;
            if (teams) teams[k] = (uint8_t)t;
            ++k;
        }
    *n = k;
    return MM_OK;
}
