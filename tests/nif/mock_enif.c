/*
 * mock_enif.c — TEST INFRASTRUCTURE.  A toy implementation of the enif_* calls declared in
 * tests/nif/erl_nif.h plus a small `mock_*` API through which tests/test_nif.py builds
 * argument terms, calls the NIF table of native/mm_nif.c and reads the result terms.
 * Terms are heap records owned by an environment; atoms are interned for the process;
 * resources are reference counted and destroyed (dtor) when the last term or the creator
 * lets go — the three behaviours of the BEAM the NIF relies on.
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "erl_nif.h"

enum { T_ATOM = 1, T_INT, T_FLOAT, T_BIN, T_TUPLE, T_CONS, T_NIL, T_RES, T_STR, T_EXC };

typedef struct res_hdr {
    ErlNifResourceType* type;
    long refs;
    size_t size;
    /* payload follows */
} res_hdr;

typedef struct mterm {
    int tag;
    struct mterm* next_in_env;
    union {
        struct { char* name; } atom;
        struct { long long v; int is_unsigned; unsigned long long u; } i;
        double f;
        struct { unsigned char* data; size_t size; int owned; } bin;
        struct { unsigned n; ERL_NIF_TERM* el; } tup;
        struct { ERL_NIF_TERM head, tail; } cons;
        struct { res_hdr* h; } res;
        struct { char* s; } str;
    } u;
} mterm;

struct enif_environment_t { mterm* terms; };
struct enif_resource_type_t { char name[64]; ErlNifResourceDtor* dtor; };

static mterm* ATOMS[256];
static int N_ATOMS;
static long LIVE_RESOURCES;
static mterm NIL_TERM = {T_NIL, NULL, {{0}}};

#define TERM(t) ((mterm*)(t))

static mterm* new_term(ErlNifEnv* env, int tag) {
    mterm* t = (mterm*)calloc(1, sizeof *t);
    t->tag = tag;
    t->next_in_env = env->terms;
    env->terms = t;
    return t;
}

void* enif_alloc(size_t size) { return malloc(size); }
void enif_free(void* p) { free(p); }

/* ---- binaries ---------------------------------------------------------------------- */
int enif_inspect_binary(ErlNifEnv* env, ERL_NIF_TERM term, ErlNifBinary* bin) {
    (void)env;
    if (TERM(term)->tag != T_BIN) return 0;
    bin->size = TERM(term)->u.bin.size;
    bin->data = TERM(term)->u.bin.data;
    bin->ref_bin = NULL;
    return 1;
}
unsigned char* enif_make_new_binary(ErlNifEnv* env, size_t size, ERL_NIF_TERM* termp) {
    mterm* t = new_term(env, T_BIN);
    t->u.bin.data = (unsigned char*)malloc(size ? size : 1);
    t->u.bin.size = size;
    t->u.bin.owned = 1;
    *termp = (ERL_NIF_TERM)t;
    return t->u.bin.data;
}
int enif_alloc_binary(size_t size, ErlNifBinary* bin) {
    bin->data = (unsigned char*)malloc(size ? size : 1);
    bin->size = size;
    bin->ref_bin = bin->data;
    return bin->data != NULL;
}
int enif_realloc_binary(ErlNifBinary* bin, size_t size) {
    unsigned char* p = (unsigned char*)realloc(bin->data, size ? size : 1);
    if (!p) return 0;
    bin->data = p; bin->size = size; bin->ref_bin = p;
    return 1;
}
void enif_release_binary(ErlNifBinary* bin) { free(bin->data); bin->data = NULL; bin->size = 0; }
ERL_NIF_TERM enif_make_binary(ErlNifEnv* env, ErlNifBinary* bin) { /* ownership moves to the term */
    mterm* t = new_term(env, T_BIN);
    t->u.bin.data = bin->data; t->u.bin.size = bin->size; t->u.bin.owned = 1;
    bin->data = NULL;
    return (ERL_NIF_TERM)t;
}

/* ---- resources --------------------------------------------------------------------- */
ErlNifResourceType* enif_open_resource_type(ErlNifEnv* env, const char* module_str, const char* name,
                                            ErlNifResourceDtor* dtor, ErlNifResourceFlags flags,
                                            ErlNifResourceFlags* tried) {
    (void)env; (void)module_str; (void)flags;
    ErlNifResourceType* t = (ErlNifResourceType*)calloc(1, sizeof *t);
    snprintf(t->name, sizeof t->name, "%s", name);
    t->dtor = dtor;
    if (tried) *tried = ERL_NIF_RT_CREATE;
    return t;
}
void* enif_alloc_resource(ErlNifResourceType* type, size_t size) {
    res_hdr* h = (res_hdr*)calloc(1, sizeof *h + size);
    h->type = type; h->refs = 1; h->size = size;
    LIVE_RESOURCES++;
    return h + 1;
}
static void res_unref(res_hdr* h) {
    if (--h->refs == 0) {
        if (h->type->dtor) h->type->dtor(NULL, h + 1);
        LIVE_RESOURCES--;
        free(h);
    }
}
void enif_release_resource(void* obj) { res_unref((res_hdr*)obj - 1); }
ERL_NIF_TERM enif_make_resource(ErlNifEnv* env, void* obj) {
    mterm* t = new_term(env, T_RES);
    t->u.res.h = (res_hdr*)obj - 1;
    t->u.res.h->refs++;
    return (ERL_NIF_TERM)t;
}
int enif_get_resource(ErlNifEnv* env, ERL_NIF_TERM term, ErlNifResourceType* type, void** objp) {
    (void)env;
    if (TERM(term)->tag != T_RES || TERM(term)->u.res.h->type != type) return 0;
    *objp = TERM(term)->u.res.h + 1;
    return 1;
}

/* ---- scalars, atoms, strings -------------------------------------------------------- */
ERL_NIF_TERM enif_make_atom(ErlNifEnv* env, const char* name) {
    (void)env;
    for (int i = 0; i < N_ATOMS; i++)
        if (!strcmp(ATOMS[i]->u.atom.name, name)) return (ERL_NIF_TERM)ATOMS[i];
    if (N_ATOMS == 256) abort();
    mterm* t = (mterm*)calloc(1, sizeof *t);
    t->tag = T_ATOM;
    t->u.atom.name = strdup(name);
    ATOMS[N_ATOMS++] = t;
    return (ERL_NIF_TERM)t;
}
static ERL_NIF_TERM mk_int(ErlNifEnv* env, long long v, unsigned long long u, int is_unsigned) {
    mterm* t = new_term(env, T_INT);
    t->u.i.v = v; t->u.i.u = u; t->u.i.is_unsigned = is_unsigned;
    return (ERL_NIF_TERM)t;
}
ERL_NIF_TERM enif_make_int(ErlNifEnv* env, int i) { return mk_int(env, i, (unsigned long long)(long long)i, 0); }
ERL_NIF_TERM enif_make_uint(ErlNifEnv* env, unsigned i) { return mk_int(env, i, i, 1); }
ERL_NIF_TERM enif_make_uint64(ErlNifEnv* env, ErlNifUInt64 i) { return mk_int(env, (long long)i, i, 1); }
ERL_NIF_TERM enif_make_string(ErlNifEnv* env, const char* s, ErlNifCharEncoding enc) {
    (void)enc;
    mterm* t = new_term(env, T_STR);
    t->u.str.s = strdup(s);
    return (ERL_NIF_TERM)t;
}
ERL_NIF_TERM enif_make_badarg(ErlNifEnv* env) { return (ERL_NIF_TERM)new_term(env, T_EXC); }
ERL_NIF_TERM enif_make_tuple(ErlNifEnv* env, unsigned cnt, ...) {
    mterm* t = new_term(env, T_TUPLE);
    t->u.tup.n = cnt;
    t->u.tup.el = (ERL_NIF_TERM*)calloc(cnt ? cnt : 1, sizeof(ERL_NIF_TERM));
    va_list ap;
    va_start(ap, cnt);
    for (unsigned i = 0; i < cnt; i++) t->u.tup.el[i] = va_arg(ap, ERL_NIF_TERM);
    va_end(ap);
    return (ERL_NIF_TERM)t;
}
int enif_get_uint(ErlNifEnv* env, ERL_NIF_TERM term, unsigned* ip) {
    (void)env;
    mterm* t = TERM(term);
    if (t->tag != T_INT || t->u.i.v < 0 || t->u.i.u > 0xFFFFFFFFull) return 0;
    *ip = (unsigned)t->u.i.u;
    return 1;
}
int enif_get_long(ErlNifEnv* env, ERL_NIF_TERM term, long* ip) {
    (void)env;
    if (TERM(term)->tag != T_INT) return 0;
    *ip = (long)TERM(term)->u.i.v;
    return 1;
}
int enif_get_double(ErlNifEnv* env, ERL_NIF_TERM term, double* dp) { /* like the BEAM: floats only */
    (void)env;
    if (TERM(term)->tag != T_FLOAT) return 0;
    *dp = TERM(term)->u.f;
    return 1;
}
int enif_get_list_cell(ErlNifEnv* env, ERL_NIF_TERM list, ERL_NIF_TERM* head, ERL_NIF_TERM* tail) {
    (void)env;
    if (TERM(list)->tag != T_CONS) return 0;
    *head = TERM(list)->u.cons.head;
    *tail = TERM(list)->u.cons.tail;
    return 1;
}
int enif_get_list_length(ErlNifEnv* env, ERL_NIF_TERM term, unsigned* len) {
    (void)env;
    unsigned n = 0;
    while (TERM(term)->tag == T_CONS) { n++; term = TERM(term)->u.cons.tail; }
    if (TERM(term)->tag != T_NIL) return 0;
    *len = n;
    return 1;
}
int enif_is_empty_list(ErlNifEnv* env, ERL_NIF_TERM term) { (void)env; return TERM(term)->tag == T_NIL; }
int enif_is_identical(ERL_NIF_TERM a, ERL_NIF_TERM b) { return a == b; } /* enough for atoms */

/* ---- the test driver's side --------------------------------------------------------- */
#define API __attribute__((visibility("default")))
API ErlNifEnv* mock_env_new(void) { return (ErlNifEnv*)calloc(1, sizeof(ErlNifEnv)); }
API void mock_env_free(ErlNifEnv* env) { /* what a GC of the calling process does */
    mterm* t = env->terms;
    while (t) {
        mterm* nx = t->next_in_env;
        switch (t->tag) {
            case T_BIN: if (t->u.bin.owned) free(t->u.bin.data); break;
            case T_TUPLE: free(t->u.tup.el); break;
            case T_RES: res_unref(t->u.res.h); break;
            case T_STR: free(t->u.str.s); break;
            default: break;
        }
        free(t);
        t = nx;
    }
    free(env);
}
API long mock_live_resources(void) { return LIVE_RESOURCES; }
API int mock_load(ErlNifEntry* e, ErlNifEnv* env) {
    void* priv = NULL;
    return e->load ? e->load(env, &priv, (ERL_NIF_TERM)&NIL_TERM) : 0;
}
API int mock_nif_count(ErlNifEntry* e) { return e->num_of_funcs; }
API const char* mock_module(ErlNifEntry* e) { return e->name; }
API const char* mock_nif_name(ErlNifEntry* e, int i) { return e->funcs[i].name; }
API unsigned mock_nif_arity(ErlNifEntry* e, int i) { return e->funcs[i].arity; }
API unsigned mock_nif_flags(ErlNifEntry* e, int i) { return e->funcs[i].flags; }
API ERL_NIF_TERM mock_call(ErlNifEntry* e, int i, ErlNifEnv* env, int argc, const ERL_NIF_TERM* argv) {
    return e->funcs[i].fptr(env, argc, argv);
}
API ERL_NIF_TERM mock_mk_binary(ErlNifEnv* env, const void* data, size_t size) {
    ERL_NIF_TERM t;
    memcpy(enif_make_new_binary(env, size, &t), data, size);
    return t;
}
API ERL_NIF_TERM mock_mk_int(ErlNifEnv* env, long long v) { return mk_int(env, v, (unsigned long long)v, 0); }
API ERL_NIF_TERM mock_mk_float(ErlNifEnv* env, double f) {
    mterm* t = new_term(env, T_FLOAT);
    t->u.f = f;
    return (ERL_NIF_TERM)t;
}
API ERL_NIF_TERM mock_mk_atom(const char* name) { return enif_make_atom(NULL, name); }
API ERL_NIF_TERM mock_mk_list(ErlNifEnv* env, unsigned n, const ERL_NIF_TERM* el) {
    ERL_NIF_TERM tail = (ERL_NIF_TERM)&NIL_TERM;
    for (unsigned i = n; i-- > 0;) {
        mterm* c = new_term(env, T_CONS);
        c->u.cons.head = el[i];
        c->u.cons.tail = tail;
        tail = (ERL_NIF_TERM)c;
    }
    return tail;
}
API int mock_tag(ERL_NIF_TERM t) { return TERM(t)->tag; }
API unsigned mock_tuple_arity(ERL_NIF_TERM t) { return TERM(t)->u.tup.n; }
API ERL_NIF_TERM mock_tuple_elem(ERL_NIF_TERM t, unsigned i) { return TERM(t)->u.tup.el[i]; }
API const void* mock_bin_data(ERL_NIF_TERM t) { return TERM(t)->u.bin.data; }
API size_t mock_bin_size(ERL_NIF_TERM t) { return TERM(t)->u.bin.size; }
API long long mock_int_value(ERL_NIF_TERM t) { return TERM(t)->u.i.v; }
API unsigned long long mock_uint_value(ERL_NIF_TERM t) { return TERM(t)->u.i.u; }
API const char* mock_atom_name(ERL_NIF_TERM t) { return TERM(t)->u.atom.name; }
API const char* mock_str(ERL_NIF_TERM t) { return TERM(t)->u.str.s; }
