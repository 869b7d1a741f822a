// Fiber scheduler + worker pool of the CPU lane-level executor (see include/emu_rt.h). TEST INFRASTRUCTURE.
#include "emu_rt.h"

#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace emu {

thread_local Ctx g;
int check_uniform_on = [] { const char* e = getenv("Y5M_EMU_CHECK_UNIFORM"); return e ? atoi(e) : 1; }();
int defer_dma_on = [] { const char* e = getenv("Y5M_EMU_DEFER_DMA"); return e ? atoi(e) : 1; }();

// ---- context switch (x86-64 SysV): callee-saved registers + stack pointer ---------------------------------------------
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

constexpr size_t STACK_BYTES = 256 << 10;
constexpr int MAX_THREADS_PER_BLOCK = 1024;

struct Pool {                     // per OS thread: fiber stacks + lane / wave objects, reused by every workgroup
    unsigned char* stacks = nullptr;
    unsigned char* lds = nullptr;
    std::vector<Lane> lanes;
    std::vector<Wave> waves;
    const std::function<void()>* body = nullptr;
    ~Pool() {
        if (stacks) munmap(stacks, STACK_BYTES * MAX_THREADS_PER_BLOCK);
        if (lds) munmap(lds, LDS_BYTES);
    }
    static constexpr size_t LDS_BYTES = 1 << 20;
    void init() {
        if (stacks) return;
        stacks = (unsigned char*)mmap(nullptr, STACK_BYTES * MAX_THREADS_PER_BLOCK, PROT_READ | PROT_WRITE,
                                      MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        // dynamic LDS: its 32-bit "LDS address" (kernels truncate the pointer for m0 / readfirstlane) must map back: keep
        // the buffer inside one 4 GiB window
        for (int tries = 0; tries < 16; ++tries) {
            lds = (unsigned char*)mmap(nullptr, LDS_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (((uintptr_t)lds >> 32) == (((uintptr_t)lds + LDS_BYTES - 1) >> 32)) break;
        }
        if (stacks == MAP_FAILED || lds == MAP_FAILED) { fprintf(stderr, "emu: mmap failed\n"); abort(); }
        lanes.resize(MAX_THREADS_PER_BLOCK);
        waves.resize(MAX_THREADS_PER_BLOCK / 64);
    }
};
static thread_local Pool pool;

static void fiber_main() {
    (*pool.body)();
    g.cur->state = DONE;
    yield_to_scheduler();
    abort();                      // a finished fiber is never resumed
}

void yield_to_scheduler() {
    Lane* l = g.cur;
    emu_switch(&l->sp, g.sched_sp);
}

static void prepare(Lane& l) {
    // initial frame: six callee-saved registers (zero) + return address = fiber_main; at fiber_main's entry the stack must
    // look as if a `call` had just pushed a return address: (rsp + 8) % 16 == 0
    uintptr_t top = ((uintptr_t)l.stack + STACK_BYTES) & ~(uintptr_t)15;
    void** sp = (void**)(top - 8);        // slot that plays the part of the caller's return address
    *sp = nullptr;
    *--sp = (void*)&fiber_main;
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    l.sp = sp;
    l.state = RUNNABLE;
    l.op = 0;
    l.nulog = 0;
    l.ulog_overflow = false;
    l.dma_head = l.dma_cnt = 0;
}

// The wave stands still (every live lane parked at a rendezvous or a barrier, the others done): compare what its lanes passed
// through readfirstlane since the last stand-still. Entry e of a lane is the k-th execution of its source line in this interval;
// on the GPU the lanes that execute that instruction together all receive ONE value (the first active lane's), so every lane that
// logged (line, k) must have logged the same value.
static void check_uniform(Wave& W, int w, Dim3 bid) {
    struct Ref { const char* file; int line, k, lane; unsigned long long val; };
    Ref refs[ULOG_MAX];
    int nrefs = 0;
    bool any = false;
    for (int i = 0; i < W.n; ++i) any |= W.lanes[i]->nulog > 0;
    if (!any) return;
    for (int i = 0; i < W.n; ++i) {
        Lane* l = W.lanes[i];
        for (int e = 0; e < l->nulog; ++e) {
            const UEntry& u = l->ulog[e];
            int k = 0;
            for (int f = 0; f < e; ++f) k += l->ulog[f].file == u.file && l->ulog[f].line == u.line;
            int r = 0;
            for (; r < nrefs; ++r)
                if (refs[r].file == u.file && refs[r].line == u.line && refs[r].k == k) break;
            if (r == nrefs) {
                if (nrefs < ULOG_MAX) refs[nrefs++] = Ref{u.file, u.line, k, l->lane, u.val};
                continue;
            }
            if (refs[r].val != u.val) {
                fprintf(stderr, "emu: readfirstlane of a value that is NOT wave-uniform at %s:%d (execution %d since the last rendezvous), "
                        "wave %d of block (%u,%u,%u): lane %d holds 0x%llx, lane %d holds 0x%llx -- the GPU would hand every lane the "
                        "first active lane's value\n", u.file, u.line, k, w, bid.x, bid.y, bid.z, refs[r].lane, refs[r].val, l->lane, u.val);
                abort();
            }
        }
        l->nulog = 0;
    }
}

static void run_block(Dim3 bid, Dim3 grid, Dim3 block, const void* kernarg) {
    Pool& P = pool;
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads > MAX_THREADS_PER_BLOCK) { fprintf(stderr, "emu: block of %d threads\n", nthreads); abort(); }
    const int nwaves = (nthreads + 63) / 64;
    g.bid = bid; g.bdim = block; g.gdim = grid; g.dyn_lds = P.lds; g.kernarg = kernarg;
    for (int t = 0; t < nthreads; ++t) {
        Lane& l = P.lanes[t];
        l.stack = P.stacks + (size_t)t * STACK_BYTES;
        l.tid = Dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        l.lane = t & 63;
        l.wave = &P.waves[t >> 6];
        prepare(l);
    }
    for (int w = 0; w < nwaves; ++w) {
        Wave& W = P.waves[w];
        W.n = (w == nwaves - 1) ? nthreads - 64 * w : 64;
        W.parity = 0;
        for (int i = 0; i < W.n; ++i) W.lanes[i] = &P.lanes[64 * w + i];
    }
    // Y5M_EMU_WAVE_ORDER: the order in which the scheduler visits the waves of a workgroup -- 0 ascending (default), 1
    // descending, >= 2 a different pseudo-random permutation per scheduler pass (seeded by the value and the block). No order
    // between waves is guaranteed on the GPU except at barriers, so every order must give the same results: a test that passes
    // ascending and fails otherwise has found a missing barrier (read-after-write or write-after-read through LDS).
    static const int wave_order = [] { const char* e = getenv("Y5M_EMU_WAVE_ORDER"); return e ? atoi(e) : 0; }();
    unsigned rng = (unsigned)wave_order * 2654435761u + bid.x * 40503u + bid.y * 977u + 12345u;
    int order[MAX_THREADS_PER_BLOCK / 64];
    int live = nthreads;
    while (live > 0) {
        bool progress = false;
        int at_barrier = 0;
        live = 0;
        for (int w = 0; w < nwaves; ++w) order[w] = wave_order == 1 ? nwaves - 1 - w : w;
        if (wave_order >= 2)
            for (int w = nwaves - 1; w > 0; --w) {
                rng = rng * 1664525u + 1013904223u;
                const int j = (int)((rng >> 8) % (unsigned)(w + 1));
                const int t = order[w]; order[w] = order[j]; order[j] = t;
            }
        for (int wi = 0; wi < nwaves; ++wi) {
            const int w = order[wi];
            Wave& W = P.waves[w];
            for (int i = 0; i < W.n; ++i) {
                Lane* l = W.lanes[i];
                if (l->state == RUNNABLE) {
                    g.cur = l;
                    emu_switch(&g.sched_sp, l->sp);
                    progress = true;
                }
            }
            if (check_uniform_on) check_uniform(W, w, bid);      // (no lane of this wave is runnable here)
            int ncoll = 0, nlive = 0, nbar = 0, op = -1;
            bool same = true;
            for (int i = 0; i < W.n; ++i) {
                Lane* l = W.lanes[i];
                if (l->state == DONE) continue;
                ++nlive;
                if (l->state == AT_COLLECTIVE) {
                    ++ncoll;
                    if (op < 0) op = l->op;
                    else if (op != l->op) same = false;
                } else if (l->state == AT_BARRIER) ++nbar;
            }
            live += nlive;
            at_barrier += nbar;
            if (nlive > 0 && ncoll == nlive) {
                if (!same) {
                    fprintf(stderr, "emu: wave %d of block (%u,%u,%u): lanes parked at DIFFERENT collective operations "
                            "(divergent control flow around a wave-wide operation)\n", w, bid.x, bid.y, bid.z);
                    abort();
                }
                W.parity ^= 1;
                for (int i = 0; i < W.n; ++i)
                    if (W.lanes[i]->state == AT_COLLECTIVE) W.lanes[i]->state = RUNNABLE;
                progress = true;
            }
        }
        if (live > 0 && at_barrier == live) {
            for (int t = 0; t < nthreads; ++t)
                if (P.lanes[t].state == AT_BARRIER) P.lanes[t].state = RUNNABLE;
            progress = true;
        }
        if (live > 0 && !progress) {
            fprintf(stderr, "emu: deadlock in block (%u,%u,%u): %d live lanes, %d at a barrier; per wave (collective / barrier / live):",
                    bid.x, bid.y, bid.z, live, at_barrier);
            for (int w = 0; w < nwaves; ++w) {
                int c = 0, b = 0, n = 0;
                for (int i = 0; i < P.waves[w].n; ++i) {
                    const int s = P.waves[w].lanes[i]->state;
                    c += s == AT_COLLECTIVE; b += s == AT_BARRIER; n += s != DONE;
                }
                fprintf(stderr, " %d/%d/%d", c, b, n);
            }
            fprintf(stderr, "\n");
            abort();
        }
    }
}

// ---- worker pool: the blocks of ONE launch at a time, taken from an atomic counter -------------------------------------
struct Job {
    Dim3 grid, block;
    const std::function<void()>* body;
    const void* kernarg;
    std::atomic<long> next{0};
    long total;
};

// (never destroyed: the detached workers wait on these for the life of the process, and destroying a condition variable with
//  waiters blocks the exit)
static std::mutex& mu = *new std::mutex;
static std::mutex& launch_mu = *new std::mutex;
static std::condition_variable& cv_work = *new std::condition_variable;
static std::condition_variable& cv_done = *new std::condition_variable;
static Job* job = nullptr;
static int generation = 0, working = 0, n_workers = 0;
static bool stopping = false;

static void work_on(Job* j) {
    pool.init();
    pool.body = j->body;
    // Y5M_EMU_BLOCK_ORDER=1: workgroups are taken from the END of the grid (no kernel may depend on the dispatch order)
    static const int block_order = [] { const char* e = getenv("Y5M_EMU_BLOCK_ORDER"); return e ? atoi(e) : 0; }();
    for (;;) {
        long b = j->next.fetch_add(1);
        if (b >= j->total) break;
        if (block_order == 1) b = j->total - 1 - b;
        const Dim3 bid((unsigned)(b % j->grid.x), (unsigned)((b / j->grid.x) % j->grid.y), (unsigned)(b / ((long)j->grid.x * j->grid.y)));
        run_block(bid, j->grid, j->block, j->kernarg);
    }
}

static void worker() {
    int seen = 0;
    for (;;) {
        Job* j;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_work.wait(lk, [&] { return stopping || (job != nullptr && generation != seen); });
            if (stopping) return;
            seen = generation;
            j = job;
            ++working;
        }
        work_on(j);
        {
            std::unique_lock<std::mutex> lk(mu);
            if (--working == 0) cv_done.notify_all();
        }
    }
}

static int n_threads() {
    static int n = -1;
    if (n < 0) {
        const char* e = getenv("Y5M_EMU_THREADS");
        n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        if (n < 1) n = 1;
        if (n > 64) n = 64;
    }
    return n;
}

void launch_blocks(Dim3 grid, Dim3 block, size_t lds_bytes, const std::function<void()>& body, const void* kernarg) {
    if (lds_bytes > Pool::LDS_BYTES) { fprintf(stderr, "emu: %zu bytes of dynamic LDS\n", lds_bytes); abort(); }
    std::lock_guard<std::mutex> one(launch_mu);            // launches are synchronous and serialised (streams are ignored)
    Job j;
    j.grid = grid; j.block = block; j.body = &body; j.kernarg = kernarg;
    j.total = (long)grid.x * grid.y * grid.z;
    const int nt = n_threads();
    if (nt == 1 || j.total == 1) {
        work_on(&j);
        return;
    }
    {
        std::unique_lock<std::mutex> lk(mu);
        for (; n_workers < nt - 1; ++n_workers) std::thread(worker).detach();
        job = &j;
        ++generation;
    }
    cv_work.notify_all();
    work_on(&j);                                           // the calling thread takes blocks too
    {
        std::unique_lock<std::mutex> lk(mu);
        // every worker that picked this generation up must have left work_on before `j` goes out of scope; workers that never
        // woke up for it find job == nullptr
        job = nullptr;
        cv_done.wait(lk, [&] { return working == 0; });
    }
}

}  // namespace emu
