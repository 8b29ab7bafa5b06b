// gm_pair_protocol.h -- the bounded LDS hand-shake of k_pair_score_sample (gm_pair.hip), written over a small policy so
// that the same lines run on the GPU (flags in LDS, 64 lanes voting) and on the host (tests/host/pair_protocol_test.cpp:
// atomics and threads), where the bound below is checked without a GPU.
//
// The protocol.  Eight waves of a workgroup hand each other LDS contents through monotone counters ("flags"); a wait reads
// flags until all of them have reached a value.  No wait depends on another workgroup.  A wait that is not satisfied after
// WAIT_SPINS polls gives up, raises the workgroup's `abort` word and the launch's sticky `stall` word (global memory), and from
// then on NOTHING in that workgroup waits again: every later wait sees `abort` at its first unsatisfied poll, and every
// wave leaves the tile loop at its next check.  Other workgroups stop pulling queue items once they see `stall`.
//
// Bound.  A wave polls at most WAIT_SPINS + TURN_SPINS times in unsatisfied waits before it leaves the loop (one full
// timeout of either kind; the wait that follows sees `abort`), plus one poll per wait it passes.  With ~0.1 us per poll
// (LDS read + s_sleep 1) a launch with a lost hand-shake therefore ends ~30 ms after the loss (r2: every one of the three
// wait sites of every later item timed out again at 2^22 polls, i.e. minutes of a "100 % busy" GPU), and a launch that
// starts with `stall` already set ends at once.
#pragma once

#if defined(__HIPCC__)
#define GM_HD __host__ __device__ __forceinline__
#else
#define GM_HD inline
#endif

namespace gm_pair_protocol {

constexpr int WAIT_SPINS = 1 << 18;   // polls of a data wait (guards LDS contents) before it gives up
constexpr int TURN_SPINS = 1 << 16;   // polls of the matrix-pipe turn wait (ordering only, never correctness)

// P supplies: bool satisfied()  -- all flags have reached the value (every lane / thread the same answer);
//             bool aborted()    -- the workgroup's abort word is set;
//             void raise()      -- set the abort word (and whatever reports the stall);
//             void pause()      -- back off between polls.
// Returns true when the wait was satisfied; false means: leave the loop now.
template <class P>
GM_HD bool bounded_wait(P& p, int max_spins, bool raise_on_timeout) {
    if (p.satisfied()) return true;
    for (int spin = 0; spin < max_spins; ++spin) {
        if (p.aborted()) return false;
        p.pause();
        if (p.satisfied()) return true;
    }
    if (raise_on_timeout) p.raise();
    return false;
}

// worst-case unsatisfied polls of one wave between a lost hand-shake and its exit from the tile loop
constexpr long worst_case_polls_per_wave() { return (long)WAIT_SPINS + (long)TURN_SPINS; }

}  // namespace gm_pair_protocol
