// CPU lane-level executor for the HIP kernels of liby5m.so -- TEST INFRASTRUCTURE, like oracle/: only tests/ builds and
// loads it (tests/emu/build.py -> build/emu/liby5m_emu.so). The product (yolov5m_amd/) never sees it and has no CPU path.
//
// What it is: the UNMODIFIED kernel sources of yolov5m_amd/csrc/*.hip compiled for x86-64 against a stand-in
// <hip/hip_runtime.h> (tests/emu/include/hip/hip_runtime.h). Every GPU thread is a fiber; the 64 fibers of a wavefront meet
// at each wave-collective operation (MFMA, ds_read_b64_tr_b16, DPP, readlane, shuffles, ballot) and exchange operands through
// a per-wave slot table, the fibers of a workgroup meet at barriers. Workgroups run one after the other on a pool of OS
// threads (atomics are real atomics). The point: the kernels' index arithmetic, tile logic, epilogues and the engine's
// launch lists can be checked against the oracle on this CPU-only container, on every commit.
// What it is NOT: a timing model, a memory-model checker (no LDS bank / cache semantics: a cross-workgroup race or a missing
// release fence in front of a barrier is invisible here) or a replacement for the -m gpu suite. Two program-visible hazards ARE
// modelled (round 5): an LDS-DMA load lands only when a `s_waitcnt vmcnt(n)` of its wave covers it (a missing or too-weak wait
// reads stale LDS), and a value passed through readfirstlane must be equal on all lanes that execute that instruction together.
// (Loads into registers need no model: the compiler places their waits, there is no window a program can see.)
//
// Execution model details a kernel author must know:
//   * lanes of a wave run ONE AFTER THE OTHER between two collective points; a collective point is any operation listed
//     above, __syncthreads / s_barrier, and the stand-ins of `s_waitcnt` (tests/emu/translate.py) -- so data handed from
//     lane to lane through LDS needs one of those in between (on the GPU the in-order LDS pipe gives that for free);
//   * a collective op must be reached by every live lane of the wave (uniform control flow); otherwise: "deadlock" abort;
//   * readfirstlane returns the calling lane's own value and LOGS it: the scheduler aborts when lanes disagree (check_uniform).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <functional>

namespace emu {

struct Dim3 {
    unsigned x, y, z;
    constexpr Dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

enum { RUNNABLE = 0, AT_COLLECTIVE = 1, AT_BARRIER = 2, DONE = 3 };
constexpr int SLOT_BYTES = 64;

struct Wave;
// checking mode for values the kernel takes for WAVE-UNIFORM (readfirstlane): every call logs (source line, value); when the wave
// next stands still (all live lanes parked or done) the scheduler compares the lanes' logs entry by entry -- the n-th execution of
// one source line between two rendezvous is the same dynamic instruction on the GPU, where the first active lane's value would be
// broadcast; lanes that disagree mean the kernel's "uniform" value is not (emu_rt.cpp: check_uniform)
struct UEntry { const char* file; int line; unsigned long long val; };
constexpr int ULOG_MAX = 192;
// an LDS-DMA load in flight (buffer_load ... lds): its 16 bytes land in LDS only when a `s_waitcnt vmcnt(n)` of the issuing wave
// leaves at most n of them outstanding -- the LATEST moment the hardware allows, so a missing or too-weak wait reads stale LDS here
struct PendingDma { unsigned char* dst; unsigned char data[16]; };
constexpr int DMA_MAX = 63;   // vmcnt is a 6-bit counter: the 64th outstanding load stalls until the oldest has landed
struct Lane {
    void* sp;                 // saved stack pointer while switched out
    unsigned char* stack;     // base of this fiber's stack
    int state;
    int op;                   // id of the collective it is parked at (checked: all lanes of a wave must agree)
    Dim3 tid;
    int lane;                 // 0..63
    Wave* wave;
    int nulog;
    bool ulog_overflow;
    int dma_head, dma_cnt;    // ring of LDS-DMA loads in flight
    UEntry ulog[ULOG_MAX];
    PendingDma dma[DMA_MAX + 1];
};

struct Wave {
    Lane* lanes[64];
    int n, parity;
    alignas(16) unsigned char slot[2][64][SLOT_BYTES];
};

struct Ctx {                  // per OS thread: the workgroup being executed
    Lane* cur;
    Dim3 bid, bdim, gdim;
    unsigned char* dyn_lds;   // the launch's dynamic LDS (zero-sized launches get a small buffer too)
    const void* kernarg;
    void* sched_sp;
};
extern thread_local Ctx g;

void yield_to_scheduler();    // park the current lane (state / op already set)

// ---- collective protocol: deposit -> sync -> read the other lanes' deposits of the same parity -------------------
struct Coll {
    Wave* w;
    int p, lane;
    template <typename T> T& mine() { return *reinterpret_cast<T*>(w->slot[p][lane]); }
    template <typename T> const T& of(int l) const { return *reinterpret_cast<const T*>(w->slot[p][l]); }
    bool live(int l) const { return l < w->n && w->lanes[l]->state != DONE; }
};
inline Coll coll_begin() {
    Lane* l = g.cur;
    return Coll{l->wave, l->wave->parity, l->lane};
}
inline void coll_sync(int op) {
    Lane* l = g.cur;
    l->state = AT_COLLECTIVE;
    l->op = op;
    yield_to_scheduler();
}
inline void wave_sync() {
    coll_sync(1);
}
// ---- values taken for wave-uniform ---------------------------------------------------------------------------------------
extern int check_uniform_on;      // Y5M_EMU_CHECK_UNIFORM (default 1)
inline void note_uniform(const char* file, int line, const void* v, size_t n) {
    Lane* l = g.cur;
    if (!check_uniform_on) return;
    if (l->nulog >= ULOG_MAX) { l->ulog_overflow = true; return; }
    unsigned long long x = 0;
    memcpy(&x, v, n < 8 ? n : 8);
    l->ulog[l->nulog++] = UEntry{file, line, x};
}
// ---- LDS-DMA loads in flight -----------------------------------------------------------------------------------------------
extern int defer_dma_on;          // Y5M_EMU_DEFER_DMA (default 1; 0 = loads land at once, the round-4 behaviour)
inline void dma_land_oldest(Lane* l) {
    PendingDma& d = l->dma[l->dma_head];
    memcpy(d.dst, d.data, 16);
    l->dma_head = (l->dma_head + 1) % (DMA_MAX + 1);
    --l->dma_cnt;
}
inline void dma_issue(unsigned char* dst, const void* data) {
    Lane* l = g.cur;
    if (!defer_dma_on) { memcpy(dst, data, 16); return; }
    if (l->dma_cnt == DMA_MAX) dma_land_oldest(l);
    PendingDma& d = l->dma[(l->dma_head + l->dma_cnt) % (DMA_MAX + 1)];
    d.dst = dst;
    memcpy(d.data, data, 16);
    ++l->dma_cnt;
}
inline void dma_wait(int n) {     // s_waitcnt vmcnt(n): at most n loads of this lane's wave stay in flight (oldest land first)
    Lane* l = g.cur;
    while (l->dma_cnt > (n < 0 ? 0 : n)) dma_land_oldest(l);
}
inline void barrier() {
    Lane* l = g.cur;
    l->state = AT_BARRIER;
    yield_to_scheduler();
}

// ---- launch --------------------------------------------------------------------------------------------------------
void launch_blocks(Dim3 grid, Dim3 block, size_t lds_bytes, const std::function<void()>& body, const void* kernarg);

template <typename T> struct arg_store { using type = T; };

template <typename... Ps, typename... As>
inline void launch(void (*k)(Ps...), Dim3 grid, Dim3 block, size_t lds, void* /*stream*/, As&&... as) {
    static_assert(sizeof...(Ps) == sizeof...(As), "kernel argument count");
    // the kernarg segment: the kernel's parameters one after the other at their natural alignment
    alignas(16) unsigned char seg[4096];
    size_t off = 0;
    auto put = [&](auto v) {
        using T = decltype(v);
        off = (off + alignof(T) - 1) / alignof(T) * alignof(T);
        if (off + sizeof(T) <= sizeof(seg)) memcpy(seg + off, &v, sizeof(T));
        off += sizeof(T);
    };
    std::tuple<std::remove_cv_t<std::remove_reference_t<Ps>>...> params{static_cast<Ps>(as)...};
    std::apply([&](auto&... p) { (put(p), ...); }, params);
    std::function<void()> body = [&]() { std::apply(k, params); };
    launch_blocks(grid, block, lds, body, seg);
}

}  // namespace emu
