/*
 * mode_r.h — CPU oracle of the search path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the canonical sequential schedule ("Mode R", SURVEY.md §3.4,
 * docs/MATCH_CHECK.md) of Matchmaking.Search.Worker.consume/5 and the strategist
 * predicate.  PARITY UNPINNED: the reference ships no golden vector, known-answer test
 * or fixture for this path and cannot be run here (no BEAM, predicate source absent);
 * the oracle is pinned instead by hand-derived vectors (tests/golden/) and by a second,
 * literal restatement (oracle/literal_ref.py) that follows worker.ex line by line.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product library (libmm_engine.so) never links or calls it.
 *
 * The API mirrors include/mm_engine.h one to one (mo_* for mm_*), so a parity test
 * drives both with the same calls.
 */
#ifndef MODE_R_H
#define MODE_R_H

#include "../include/mm_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mo_engine mo_engine;

int  mo_engine_create(const mm_config* cfg, mo_engine** out);
void mo_engine_destroy(mo_engine* e);
int  mo_reset(mo_engine* e);
int  mo_find_rating_group(const mm_config* cfg, double rating, uint32_t* group);
int  mo_enqueue(mo_engine* e, uint32_t n, const int32_t* rating, const uint32_t* cons,
                const uint8_t* group, uint32_t* out_slot, mm_enqueue_stats* st);
int  mo_cancel(mo_engine* e, uint32_t n, const uint32_t* slot);
int  mo_tick(mo_engine* e, uint32_t mode, uint32_t* n_matches, mm_stats* stats);
/* One thread per rating group — the reference's own parallelism (application.ex:30-39). */
int  mo_tick_threads(mo_engine* e, uint32_t mode, uint32_t n_threads, uint32_t* n_matches,
                     mm_stats* stats);
int  mo_matches(mo_engine* e, uint32_t first, uint32_t count, uint32_t* slots, float* score,
                uint32_t* group, uint32_t* pass);
int  mo_queue_depth(mo_engine* e, uint32_t mode, uint32_t* per_group);
/* Queue contents (slots, head first) of (mode, group); *n in = capacity of `slots`. */
int  mo_queue_slots(mo_engine* e, uint32_t mode, uint32_t group, uint32_t* n, uint32_t* slots);
int  mo_lobby_state(mo_engine* e, uint32_t mode, uint32_t group, uint32_t* n, uint32_t* slots,
                    uint8_t* teams);

#ifdef __cplusplus
}
#endif
#endif
