// CPU lane-level executor for the HIP kernels of liby5m.so -- TEST INFRASTRUCTURE, like oracle/: only tests/ builds and
// loads it (tests/emu/build.py -> build/emu/liby5m_emu.so). The product (yolov5m_amd/) never sees it and has no CPU path.
//
// What it is: the UNMODIFIED kernel sources of yolov5m_amd/csrc/*.hip compiled for x86-64 against a stand-in
// <hip/hip_runtime.h> (tests/emu/include/hip/hip_runtime.h). Every GPU thread is a fiber; the 64 fibers of a wavefront meet
// at each wave-collective operation (MFMA, ds_read_b64_tr_b16, DPP, readlane, shuffles, ballot) and exchange operands through
// a per-wave slot table, the fibers of a workgroup meet at barriers. Workgroups run one after the other on a pool of OS
// threads (atomics are real atomics). The point: the kernels' index arithmetic, tile logic, epilogues and the engine's
// launch lists can be checked against the oracle on this CPU-only container, on every commit.
// What it is NOT: a timing model, a memory-model checker (no s_waitcnt / LDS bank / cache semantics: a missing wait or a
// cross-workgroup race is invisible here) or a replacement for the -m gpu suite.
//
// Execution model details a kernel author must know:
//   * lanes of a wave run ONE AFTER THE OTHER between two collective points; a collective point is any operation listed
//     above, __syncthreads / s_barrier, and the stand-ins of `s_waitcnt` (tests/emu/translate.py) -- so data handed from
//     lane to lane through LDS needs one of those in between (on the GPU the in-order LDS pipe gives that for free);
//   * a collective op must be reached by every live lane of the wave (uniform control flow); otherwise: "deadlock" abort;
//   * readfirstlane returns the calling lane's own value (its uses here make uniform values scalar).
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
struct Lane {
    void* sp;                 // saved stack pointer while switched out
    unsigned char* stack;     // base of this fiber's stack
    int state;
    int op;                   // id of the collective it is parked at (checked: all lanes of a wave must agree)
    Dim3 tid;
    int lane;                 // 0..63
    Wave* wave;
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
