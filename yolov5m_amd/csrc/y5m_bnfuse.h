// BatchNorm statistics without partial rows and without a finalise launch (round 2).
//
// Before: producer (conv epilogue / backward reduce) -> partial rows in HBM -> bn_reduce_finalize_kernel (its own
// launch: ~8 us of work, ~15 us of critical chain, 158 of them per train step) -> consumer (normalise / apply).
// Now: every producing workgroup ADDS its partial sums (f32, already reduced over the workgroup's pixels) as f64
// atomics into one of BNF_SLOTS accumulator rows acc[slot][2][ld] (slot = low bits of the tile / block index:
// same-address contention measured free at 8 slots, tools/probe_atomics.hip) and is done -- fire and forget. The
// CONSUMER launch (bn_act / bn_bwd_apply) sums the slot rows in a fixed order in every workgroup's prologue and derives
// its per-channel coefficients itself (same f64 arithmetic as bn_reduce_finalize_kernel); the workgroups with
// blockIdx.x == 0 also write the per-channel arrays later launches read (scale / shift / mean / invstd, running
// statistics; dgamma / dbeta). Visibility is the kernel boundary between producer and consumer; nobody waits on anybody
// inside a launch. The accumulators are zeroed by the caller ahead of the producer (the engine: ONE memset per pass).
// What was measured on the way (B=64 @ 640^2 step, same box): producer-side finalise by the workgroup that draws the
// last agent-scope ticket: forward +0.6 ms (every workgroup stays resident for the drain of its atomics + the ticket's
// round trip), backward -0.15 ms; atomics alone (nobody finalises): -2.1 ms. The sums arrive in f64, so the order of
// the atomic adds moves a total by ~1e-16 relative -- below the f32 rounding of everything derived from it in all but
// measure-zero cases.
#pragma once
#include "y5m_common.h"

#ifdef BNF_SLOTS_OVERRIDE            /* experiment builds (tools/bn_bench.py) */
#define BNF_SLOTS BNF_SLOTS_OVERRIDE
#else
#define BNF_SLOTS 8
#endif

// add one partial sum; which = 0 (first sum) / 1 (second sum), c = channel index inside the row
__device__ __forceinline__ void bnf_add(double* acc, int ld, int slot, int which, int c, float v) {
    __hip_atomic_fetch_add(acc + ((size_t)(slot & (BNF_SLOTS - 1)) * 2 + which) * ld + c, (double)v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// consumer: totals of channel c, slot rows summed in a fixed order
__device__ __forceinline__ void bnf_sum(const double* __restrict__ acc, int ld, int c, double& a, double& b) {
    double va[BNF_SLOTS], vb[BNF_SLOTS];
#pragma unroll
    for (int s = 0; s < BNF_SLOTS; ++s) {
        va[s] = acc[((size_t)s * 2 + 0) * ld + c];
        vb[s] = acc[((size_t)s * 2 + 1) * ld + c];
    }
    a = 0.0; b = 0.0;
#pragma unroll
    for (int s = 0; s < BNF_SLOTS; ++s) { a += va[s]; b += vb[s]; }
}
