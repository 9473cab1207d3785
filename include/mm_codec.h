/* mm_codec.h — the enqueue-side codec, part of libmm_engine.so.
 *
 * Replaces, once per BATCH instead of once per message on the BEAM:
 *   Poison.decode!(payload) + data["rating"]            Generic.Worker.consume/4,
 *                                                        lib/generic/worker.ex:55-57
 *   find_rating_group_by_rating(data["rating"])          lib/generic/worker.ex:46-53
 *   Poison.decode!(payload) + Map.pop(.., "game-mode")   Search.Worker.consume/5,
 *                                                        lib/search/worker.ex:292-294
 * and hands back the SoA columns mm_enqueue takes (rating, cons, group override), so that a
 * 100k players/s stream (BASELINE cfg-5) reaches the engine without per-message work in
 * Elixir.  The payload bytes themselves stay with the host (slot -> payload table); the id is
 * returned as a span into the message.
 *
 * Host-side C, no device work: decoding is byte parsing at a few hundred MB/s per core,
 * three orders of magnitude above the stream.  Semantics checked against Python's json module
 * and oracle/literal_ref.find_rating_group_by_rating (tests/test_codec.py); like the rest of
 * the oracle it is unpinned by the reference, which has no tests for this path.
 */
#ifndef MM_CODEC_H
#define MM_CODEC_H

#include "mm_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mm_codec_cfg {
    uint32_t    n_modes;                    /* <= cfg->n_modes names */
    const char* mode_name[MM_MAX_MODES];    /* value of "game-mode" (worker.ex:294) -> mode index */
    const char* region_key;                 /* extension fields of docs/MATCH_CHECK.md section 1 (small   */
    const char* party_key;                  /* non-negative integers); NULL or absent in the message = 0  */
    const char* role_key;
} mm_codec_cfg;

/* status[i] */
#define MM_DEC_OK                0u
#define MM_DEC_BAD_JSON          1u /* Poison.decode! would raise: not a JSON object, bad syntax, bad UTF-8   */
#define MM_DEC_NO_MODE           2u /* "game-mode" missing, not a string, or not a configured mode            */
#define MM_DEC_BAD_FIELD         3u /* an extension field is present but not an integer in range             */
#define MM_DEC_RATING_INEXACT    4u /* rating is a number but not an int32: the group comes from the exact   */
                                    /* value (lines 46-53 compare numbers), rating[] holds floor(), clamped  */
#define MM_DEC_RATING_NOT_NUMBER 5u /* missing / null / string ...: Erlang term order puts every non-number  */
                                    /* above every number, so no range matches -> default group; rating 0   */

/* Decodes n messages.  Message i is buf[off[i] .. off[i+1]).  Outputs (each n entries; rating,
 * cons, group, status are required, the id spans optional):
 *   rating[i], cons[i]   as mm_enqueue takes them (cons = MM_CONS_MAKE(mode, region, party, role))
 *   group[i]             rating group of the EXACT rating — pass it as mm_enqueue's group override
 *   status[i]            MM_DEC_*; for 1..3 the other columns of the row are zero and the caller
 *                        drops the message (the reference would crash the worker on 1)
 *   id_off[i], id_len[i] the "id" member's value inside the message: the contents of a string
 *                        (escapes not resolved) or the text of a number; 0, 0 if absent
 * Duplicate keys: the last one wins (Poison builds the map with Map.put).  Values nested deeper
 * than 512 levels are refused (MM_DEC_BAD_JSON).
 * Returns MM_OK, or MM_ERR_INVALID_ARG for NULL / inconsistent arguments. */
int mm_decode_players(const mm_config* cfg, const mm_codec_cfg* cc, const char* buf, const uint64_t* off, uint32_t n,
                      int32_t* rating, uint32_t* cons, uint8_t* group, uint8_t* status,
                      uint32_t* id_off, uint32_t* id_len);

/* the lobby the search stage publishes,
 *   Poison.encode!(%{"teams" => updated_grouped_players, "game-mode" => game_mode})
 *                                                        lib/search/worker.ex:315-318
 * read back by the lobby worker with Poison.decode! (lib/game-lobby/worker.ex:119-127:
 * data["teams"], data["game-mode"], required slots = players over all teams, :37-39).
 * A player inside it is its delivery payload as decoded, minus the "game-mode" member
 * (Map.pop, worker.ex:294); the teams are "team 1", "team 2", ... (docs/MATCH_CHECK.md
 * section 1).
 *
 * Input: the L = teams * team_size payloads of one lobby in mm_matches order (team major,
 * seating order inside the team), as they were delivered.  Output: one JSON object — the bytes
 * Poison 4.0.1 (reference mix.lock:17) writes for that map.  Poison decodes and re-encodes, so
 * every value is re-written in its canonical form, on every level:
 *   objects  members in DESCENDING bytewise key order (Poison.Encoder.Map folds :maps.keys/1 —
 *            ascending for maps of <= 32 keys — with a prepend); a duplicate member keeps its last
 *            value; hence {"teams":{"team 2":[..],"team 1":[..]},"game-mode":".."}
 *   arrays   order kept (Poison.Encoder.List, foldr)
 *   strings  escapes resolved, then \" \\ \n \t \r \f \b, other bytes <= 0x1F and 0x7F as \u00XX with
 *            uppercase hex digits, everything else (UTF-8, "/") raw  (Poison.Encoder.BitString)
 *   integers the digits ("-0" is 0); a number with a fraction or an exponent is a float and is
 *            written as :io_lib_format.fwrite_g/1 writes it (2500.5, 100.0, 1.0e3, 0.001, 1.0e-5; both zeros as 0.0, as OTP < 27 does)
 *
 * Equivalence bar: byte identity with Poison.encode! for objects of up to 32 members per level
 * (beyond that the VM's hash order decides :maps.keys/1, which cannot be restated; such an object is
 * still written in descending order and decodes to the same map).  Pinned by hand-derived golden
 * strings and an independent Python restatement of the same Poison / OTP sources
 * (tests/test_codec.py); no BEAM exists in this image to run Poison itself, and the consumer
 * (lib/game-lobby/worker.ex:119-127) decodes the message anyway.
 *
 * *written = bytes needed; MM_ERR_RANGE if cap is smaller (nothing useful in out then);
 * MM_ERR_INVALID_ARG for a payload that is not a JSON object or bad arguments. */
int mm_encode_lobby(const char* game_mode, uint32_t teams, uint32_t team_size, const char* const* payload,
                    const uint32_t* payload_len, char* out, uint64_t cap, uint64_t* written);

#ifdef __cplusplus
}
#endif
#endif /* MM_CODEC_H */
