// Arrival tickets of the kernels that merge their own pieces inside the launch (split-KV, the balanced line's cut sequences, the tile
// kernel's K splits): "the last workgroup to arrive at an item reduces it".
//
// The drop-in boundary is a STATELESS callee (csrc/src/ffi.rs:3-102).  Until round 4 the arrival words had to be zero on
// entry and were put back to zero by the last arriver, so a launch that ended inconsistently (a graph replayed beside eager calls on its
// stream, metadata changed under a running launch) left words non-zero and every LATER merge on that stream was silently wrong until
// atoma_reset_sync_counters.  Now a word carries the EPOCH of the launch that last used it and a launch never trusts a word of another
// epoch:
//
//      word (64 bit) = (epoch << 16) | arrivals,      epoch = the launch's AQL dispatch id + 1
//
// An arrival is two atomics issued back to back, the second returning:  fetch_max(word, epoch << 16)  -- a word of an older launch
// (any count, any garbage below this epoch) becomes "epoch, 0 arrivals"; a word of THIS launch is left alone --  then
// fetch_add(word, 1) -> ticket.  The dispatch id is the packet's index in its hardware queue: every launch of a stream, eager or from a
// replayed graph, gets a larger one than all launches before it on that stream, with no host involvement and nothing baked into a
// captured graph; 48 bits of it never wrap.  Both atomics go to the same address from the same lane, so the L2 performs them in issue
// order; nobody waits.  Cost against the old fetch_add: one extra request in flight, no extra round trip.
//
// The last arriver ALSO puts the word back to zero (off the critical path: it is about to merge anyway).  That is not what makes a launch
// correct -- the epoch does -- but it keeps the words clean across what the epoch cannot see: dispatch ids are per hardware QUEUE, and a
// stream handle that is destroyed and created again may come back with the same address (the same words here) on another queue whose ids
// start lower (found in round 5: eight fresh streams of a test inherited the words, and the "future" epochs, of eight dead ones).  So a
// later launch is wrong only if an inconsistent episode left words non-zero AND the stream then moved to a queue with lower ids -- before,
// the episode alone sufficed, for ever.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace atoma {

typedef unsigned long long sync_word_t;
constexpr int SYNC_COUNT_BITS = 16;   // up to 65535 arrivals per item and launch

// The dispatch id is a kernel-entry SGPR pair (ENABLE_SGPR_DISPATCH_ID in the kernel descriptor, set by the compiler when the
// intrinsic is used).  clang has no __builtin for llvm.amdgcn.dispatch.id; an asm label reaches the intrinsic directly.
extern "C" __device__ unsigned long long atoma_llvm_amdgcn_dispatch_id(void) __asm("llvm.amdgcn.dispatch.id");

// wave-uniform
__device__ __forceinline__ sync_word_t sync_epoch() { return ((sync_word_t)atoma_llvm_amdgcn_dispatch_id() + 1ull) << SYNC_COUNT_BITS; }

// ticket of this arrival at `word` in the launch whose epoch is `epoch`, n arrivals expected: 0 for the first to arrive, n - 1 for the last
// Memory order.  The pieces an arriver publishes before its ticket are agent-scope write-through stores (past the XCD's L2) drained
// with s_waitcnt vmcnt(0); the last arriver reads them with agent-scope loads: correct on gfx950 by what those instructions do, but the
// HIP / LLVM memory model only promises it for a RELEASE ticket and an ACQUIRE before the reads.  `make syncrel` builds that variant
// (ATOMA_SYNC_RELEASE: release fetch_add = L2 write-back + wait in front of the atomic, acquire fence = L2 / L1 invalidate in the last
// arriver).  Measured round 5 (profiles/r05_sync_ticket_release_vs_relaxed_ab.txt, same box, interleaved): the ragged headline shape
// 0.516-0.525 -> 0.686 ms (every wavefront of the line writes the XCD's whole L2 back once or twice), the 70B rank step 8.73 -> 9.6 ms
// (every workgroup of every K-split projection), split + combine launches level (no ticket there).  So the relaxed variant ships, with
// its ordering argument stated in ISA terms: (1) pieces leave as sc1 (write-through) stores, (2) s_waitcnt vmcnt(0) returns when the
// memory side has acknowledged them, (3) only then the lane issues the ticket atomic (performed at L2 / memory, device-coherent), (4) the
// last arriver's reads are sc1 loads issued after its own atomic returned: they miss the non-coherent caches by construction.
__device__ __forceinline__ unsigned sync_arrive(sync_word_t *word, sync_word_t epoch, unsigned n) {
    __hip_atomic_fetch_max(word, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef ATOMA_SYNC_RELEASE
    const unsigned t = (unsigned)(__hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) & ((1ull << SYNC_COUNT_BITS) - 1));
    if (t + 1 == n) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
    const unsigned t = (unsigned)(__hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ((1ull << SYNC_COUNT_BITS) - 1));
#endif
    if (t + 1 == n) __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t;
}

}  // namespace atoma
