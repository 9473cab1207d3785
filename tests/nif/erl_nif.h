/*
 * erl_nif.h — TEST STAND-IN, not Erlang/OTP's header.  The image has no BEAM; this file
 * declares exactly the documented enif_* calls native/mm_nif.c uses (names, argument order
 * and meaning as in the erl_nif manual), and tests/nif/mock_enif.c implements them over a
 * toy term heap so that tests/test_nif.py can load the NIF's function table and call it the
 * way the BEAM would.  Written for this repository; nothing here ships.
 */
#ifndef MOCK_ERL_NIF_H
#define MOCK_ERL_NIF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uintptr_t ERL_NIF_TERM;
typedef struct enif_environment_t ErlNifEnv;
typedef struct enif_resource_type_t ErlNifResourceType;
typedef void ErlNifResourceDtor(ErlNifEnv*, void*);
typedef uint64_t ErlNifUInt64;

typedef struct {
    size_t size;
    unsigned char* data;
    void* ref_bin; /* mock: the off-heap block of enif_alloc_binary, else NULL */
} ErlNifBinary;

typedef struct {
    const char* name;
    unsigned arity;
    ERL_NIF_TERM (*fptr)(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]);
    unsigned flags;
} ErlNifFunc;

typedef struct {
    int major, minor;
    const char* name;
    int num_of_funcs;
    ErlNifFunc* funcs;
    int (*load)(ErlNifEnv*, void** priv_data, ERL_NIF_TERM load_info);
    int (*reload)(ErlNifEnv*, void** priv_data, ERL_NIF_TERM load_info);
    int (*upgrade)(ErlNifEnv*, void** priv_data, void** old_priv_data, ERL_NIF_TERM load_info);
    void (*unload)(ErlNifEnv*, void* priv_data);
} ErlNifEntry;

typedef enum { ERL_NIF_RT_CREATE = 1, ERL_NIF_RT_TAKEOVER = 2 } ErlNifResourceFlags;
typedef enum { ERL_NIF_LATIN1 = 1 } ErlNifCharEncoding;
#define ERL_NIF_DIRTY_JOB_CPU_BOUND 1u
#define ERL_NIF_DIRTY_JOB_IO_BOUND 2u

void* enif_alloc(size_t size);
void enif_free(void* ptr);

int enif_inspect_binary(ErlNifEnv*, ERL_NIF_TERM bin_term, ErlNifBinary* bin);
unsigned char* enif_make_new_binary(ErlNifEnv*, size_t size, ERL_NIF_TERM* termp);
int enif_alloc_binary(size_t size, ErlNifBinary* bin);
int enif_realloc_binary(ErlNifBinary* bin, size_t size);
void enif_release_binary(ErlNifBinary* bin);
ERL_NIF_TERM enif_make_binary(ErlNifEnv*, ErlNifBinary* bin);

ErlNifResourceType* enif_open_resource_type(ErlNifEnv*, const char* module_str, const char* name,
                                            ErlNifResourceDtor* dtor, ErlNifResourceFlags flags,
                                            ErlNifResourceFlags* tried);
void* enif_alloc_resource(ErlNifResourceType* type, size_t size);
void enif_release_resource(void* obj);
ERL_NIF_TERM enif_make_resource(ErlNifEnv*, void* obj);
int enif_get_resource(ErlNifEnv*, ERL_NIF_TERM term, ErlNifResourceType* type, void** objp);

ERL_NIF_TERM enif_make_atom(ErlNifEnv*, const char* name);
ERL_NIF_TERM enif_make_int(ErlNifEnv*, int i);
ERL_NIF_TERM enif_make_uint(ErlNifEnv*, unsigned i);
ERL_NIF_TERM enif_make_uint64(ErlNifEnv*, ErlNifUInt64 i);
ERL_NIF_TERM enif_make_string(ErlNifEnv*, const char* string, ErlNifCharEncoding);
ERL_NIF_TERM enif_make_badarg(ErlNifEnv*);
ERL_NIF_TERM enif_make_tuple(ErlNifEnv*, unsigned cnt, ...);
#define enif_make_tuple2(env, a, b) enif_make_tuple(env, 2, a, b)
#define enif_make_tuple3(env, a, b, c) enif_make_tuple(env, 3, a, b, c)
#define enif_make_tuple4(env, a, b, c, d) enif_make_tuple(env, 4, a, b, c, d)
#define enif_make_tuple5(env, a, b, c, d, e) enif_make_tuple(env, 5, a, b, c, d, e)
#define enif_make_tuple6(env, a, b, c, d, e, f) enif_make_tuple(env, 6, a, b, c, d, e, f)
#define enif_make_tuple7(env, a, b, c, d, e, f, g) enif_make_tuple(env, 7, a, b, c, d, e, f, g)

int enif_get_uint(ErlNifEnv*, ERL_NIF_TERM term, unsigned* ip);
int enif_get_long(ErlNifEnv*, ERL_NIF_TERM term, long* ip);
int enif_get_double(ErlNifEnv*, ERL_NIF_TERM term, double* dp);
int enif_get_list_cell(ErlNifEnv*, ERL_NIF_TERM list, ERL_NIF_TERM* head, ERL_NIF_TERM* tail);
int enif_get_list_length(ErlNifEnv*, ERL_NIF_TERM term, unsigned* len);
int enif_is_empty_list(ErlNifEnv*, ERL_NIF_TERM term);
int enif_is_identical(ERL_NIF_TERM lhs, ERL_NIF_TERM rhs);

#define ERL_NIF_INIT(MODULE, FUNCS, LOAD, RELOAD, UPGRADE, UNLOAD)                              \
    __attribute__((visibility("default"))) ErlNifEntry* nif_init(void) {                       \
        static ErlNifEntry entry = {2, 15, #MODULE, (int)(sizeof(FUNCS) / sizeof(FUNCS[0])),    \
                                    FUNCS, LOAD, RELOAD, UPGRADE, UNLOAD};                      \
        return &entry;                                                                         \
    }

#ifdef __cplusplus
}
#endif
#endif
