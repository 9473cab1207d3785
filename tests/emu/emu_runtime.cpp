/* tests/emu/emu_runtime.cpp — fiber scheduler behind tests/emu/hip/hip_runtime.h.
 * TEST INFRASTRUCTURE ONLY (see the header). */
#include "hip/hip_runtime.h"

#include <setjmp.h>
#include <time.h>
#include <ucontext.h>

#include <vector>

emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace {

enum State { RUNNABLE, AT_BARRIER, AT_WAVEOP, DONE };
enum WaveOp { OP_NONE, OP_BALLOT, OP_SHFL };

struct Fiber {
    ucontext_t ctx;           /* first entry only; afterwards _setjmp/_longjmp (no signal-mask syscalls) */
    jmp_buf jb;
    bool started = false;
    char* stack = nullptr;
    State state = RUNNABLE;
    WaveOp op = OP_NONE;
    long long operand = 0;    /* ballot: pred; shfl: value */
    int src_lane = 0;         /* shfl */
    unsigned long long result = 0;
};

const size_t kStack = 256 * 1024;
std::vector<Fiber> g_fibers;
std::vector<char*> g_stacks;  /* pooled: a block of 1024 threads would otherwise map 256 MB per launch */
ucontext_t g_sched;
jmp_buf g_sched_jb;
int g_cur = -1;
const std::function<void()>* g_body = nullptr;

void fiber_main()
{
    (*g_body)();
    g_fibers[g_cur].state = DONE;
    _longjmp(g_sched_jb, 1);
}

void yield_to_sched()
{
    if (_setjmp(g_fibers[g_cur].jb) == 0) _longjmp(g_sched_jb, 1);
}

/* run fiber t until it yields or finishes */
void resume(unsigned t)
{
    Fiber& f = g_fibers[t];
    if (_setjmp(g_sched_jb) != 0) return;
    if (!f.started) {
        f.started = true;
        swapcontext(&g_sched, &f.ctx);   /* never returns here: fibers leave through g_sched_jb */
    } else _longjmp(f.jb, 1);
}

[[noreturn]] void die(const char* msg)
{
    fprintf(stderr, "[mm-emu] %s (block %u)\n", msg, blockIdx.x);
    for (size_t i = 0; i < g_fibers.size(); ++i)
        if (g_fibers[i].state != DONE && (i % 64 == 0 || g_fibers[i].state != g_fibers[i - 1].state))
            fprintf(stderr, "  thread %zu state %d op %d\n", i, (int)g_fibers[i].state, (int)g_fibers[i].op);
    abort();
}

}  // namespace

void __syncthreads(void)
{
    g_fibers[g_cur].state = AT_BARRIER;
    yield_to_sched();
}

unsigned long long __ballot(int pred)
{
    Fiber& f = g_fibers[g_cur];
    f.state = AT_WAVEOP;
    f.op = OP_BALLOT;
    f.operand = pred != 0;
    yield_to_sched();
    return g_fibers[g_cur].result;
}

int emu_shfl_i32(int v, int src_lane)
{
    Fiber& f = g_fibers[g_cur];
    f.state = AT_WAVEOP;
    f.op = OP_SHFL;
    f.operand = v;
    f.src_lane = src_lane & 63;
    yield_to_sched();
    return (int)g_fibers[g_cur].result;
}

static void run_block(unsigned nthreads)
{
    g_fibers.assign(nthreads, Fiber());
    while (g_stacks.size() < nthreads) g_stacks.push_back((char*)malloc(kStack));
    for (unsigned t = 0; t < nthreads; ++t) {
        Fiber& f = g_fibers[t];
        f.stack = g_stacks[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &g_sched;
        makecontext(&f.ctx, fiber_main, 0);
    }
    unsigned done = 0;
    while (done < nthreads) {
        bool progress = false;
        for (unsigned t = 0; t < nthreads; ++t) {
            if (g_fibers[t].state != RUNNABLE) continue;
            g_cur = (int)t;
            threadIdx.x = t;
            resume(t);
            progress = true;
            if (g_fibers[t].state == DONE) ++done;
        }
        /* wave collectives: complete when every live lane of the wave has arrived */
        for (unsigned w0 = 0; w0 < nthreads; w0 += 64) {
            unsigned w1 = w0 + 64 < nthreads ? w0 + 64 : nthreads;
            unsigned live = 0, arrived = 0;
            WaveOp op = OP_NONE;
            for (unsigned t = w0; t < w1; ++t) {
                if (g_fibers[t].state == DONE) continue;
                ++live;
                if (g_fibers[t].state == AT_WAVEOP) {
                    ++arrived;
                    if (op == OP_NONE) op = g_fibers[t].op;
                    else if (op != g_fibers[t].op) die("wave lanes at different collectives");
                }
            }
            if (!arrived || arrived != live) continue;
            if (op == OP_BALLOT) {
                unsigned long long m = 0;
                for (unsigned t = w0; t < w1; ++t)
                    if (g_fibers[t].state == AT_WAVEOP && g_fibers[t].operand) m |= 1ull << (t - w0);
                for (unsigned t = w0; t < w1; ++t)
                    if (g_fibers[t].state == AT_WAVEOP) g_fibers[t].result = m;
            } else {
                for (unsigned t = w0; t < w1; ++t) {
                    if (g_fibers[t].state != AT_WAVEOP) continue;
                    unsigned s = w0 + (unsigned)g_fibers[t].src_lane;
                    g_fibers[t].result = (s < w1 && g_fibers[s].state == AT_WAVEOP)
                                             ? (unsigned long long)g_fibers[s].operand
                                             : (unsigned long long)g_fibers[t].operand;
                }
            }
            for (unsigned t = w0; t < w1; ++t)
                if (g_fibers[t].state == AT_WAVEOP) { g_fibers[t].state = RUNNABLE; g_fibers[t].op = OP_NONE; }
            progress = true;
        }
        /* block barrier */
        unsigned live = 0, atb = 0;
        for (unsigned t = 0; t < nthreads; ++t) {
            if (g_fibers[t].state == DONE) continue;
            ++live;
            if (g_fibers[t].state == AT_BARRIER) ++atb;
        }
        if (live && atb == live) {
            for (unsigned t = 0; t < nthreads; ++t)
                if (g_fibers[t].state == AT_BARRIER) g_fibers[t].state = RUNNABLE;
            progress = true;
        }
        if (!progress && done < nthreads) die("deadlock: divergent collective or barrier");
    }
    g_fibers.clear();
}

void emu_launch(emu_dim3 grid, emu_dim3 block, const std::function<void()>& body)
{
    g_body = &body;
    gridDim = grid;
    blockDim = block;
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned b = 0; b < grid.x; ++b) {
            blockIdx = emu_dim3(b, by, 0);
            run_block(block.x);
        }
    g_body = nullptr;
}

/* ---- host API ---- */
struct emu_stream { int dummy; };
struct emu_event { double t; };

static double now_ms()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
/* the current device is per-thread state, as in HIP: the engine's entry points must select theirs */
static thread_local int t_device = 0;
static int g_set_device_calls = 0;
hipError_t hipSetDevice(int d) { t_device = d; ++g_set_device_calls; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
extern "C" int emu_current_device(void) { return t_device; }
extern "C" void emu_set_current_device(int d) { t_device = d; }
extern "C" int emu_set_device_calls(void) { return g_set_device_calls; }
hipError_t hipMalloc(void** p, size_t n)
{
    /* poison so that reads of never-written device memory show up */
    *p = malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    memset(*p, 0xA5, n);
    return hipSuccess;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new emu_stream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }   // launches run to their end inside the launch call
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "emu"; }
