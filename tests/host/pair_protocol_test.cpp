// pair_protocol_test.cpp -- CPU check of the pair kernel's bounded hand-shake (graphmaker_amd/csrc/gm_pair_protocol.h).
// The waits are the header's own code (bounded_wait over a policy); what is modelled here is only the order in which the
// eight waves of a workgroup of k_pair_score_sample (gm_pair.hip) wait for and publish their LDS flags: stg[4] (tiles
// staged by waves 0-3), turn[8] (row-units multiplied), pub (queue items published by wave 0), abort, and the launch's
// global stall word.  One thread per wave.  Cases: (1) a clean run waits and never gives up; (2) a wave that loses ONE
// publication makes every wave leave the loop after at most worst_case_polls_per_wave() unsatisfied polls (r2: every later
// wait timed out again); (3) a launch that starts with the stall word set does nothing.
// Build + run: tests/test_host_cpu.py (g++ -O2 -pthread).  Prints "ok" and exits 0, or says what failed.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "gm_pair_protocol.h"

using namespace gm_pair_protocol;

struct Workgroup {
    std::atomic<int> stg[4], turn[8], pub{0}, abort{0};
    std::atomic<unsigned>* stall;
    std::atomic<int> counter{0};
    Workgroup() { for (auto& s : stg) s = 0; for (auto& t : turn) t = 0; }
};

struct Wait {          // the policy bounded_wait() runs over: LdsWait of gm_pair_common.h with atomics for LDS words
    std::atomic<int>* flags; int cnt, need; Workgroup* wg; long* polls; int* raised; long* polls_paused;
    bool satisfied() { for (int k = 0; k < cnt; ++k) if (flags[k].load(std::memory_order_acquire) < need) { ++*polls; return false; } return true; }
    bool aborted() { return wg->abort.load() != 0; }
    void raise() { wg->abort = 1; wg->stall->fetch_or(1u); ++*raised; }
    // a poll of the kernel costs ~0.1 us of a wave that nothing can deschedule; a host thread can lose its core for
    // milliseconds, so the model's polls are made slower (2^18 of them: >= 0.4 s) -- the COUNT is what is checked
    void pause() { if ((++*polls_paused & 63) == 0) std::this_thread::sleep_for(std::chrono::microseconds(100)); else std::this_thread::yield(); }
};

struct Result { long polls = 0, paused = 0; int raised = 0, items = 0; };

// wave `wv` of a workgroup that takes queue items until `nitems`; `lose_item` / `lose_wave`: that wave skips publishing
// its last row-unit of that item (the lost hand-shake)
static void wave(Workgroup* wg, int wv, int nitems, int lose_item, int lose_wave, Result* res) {
    const int grp = wv >> 2;
    int seq = 0, units_done = 0;
    int t = wg->stall->load() ? 0x7fffffff : 0;         // tile_s[0]: every wave reads the same first item
    while (t < nitems) {
        Wait w0{wg->stg, 4, seq, wg, &res->polls, &res->raised, &res->paused};
        if (!bounded_wait(w0, WAIT_SPINS, true)) break;
        for (int u = 0; u < 8; ++u) {
            Wait wt{&wg->turn[wv ^ 4], 1, units_done + grp, wg, &res->polls, &res->raised, &res->paused};
            bounded_wait(wt, TURN_SPINS, false);
            ++units_done;
            if (!(seq == lose_item && wv == lose_wave && u == 7)) wg->turn[wv].store(units_done, std::memory_order_release);
        }
        ++res->items;
        ++seq;
        if (wv == 0) wg->pub.store(seq, std::memory_order_release);
        Wait w1{&wg->pub, 1, seq, wg, &res->polls, &res->raised, &res->paused};
        if (!bounded_wait(w1, WAIT_SPINS, true)) break;
        t = wg->stall->load() ? 0x7fffffff : seq;
        if (grp == 0 && t < nitems) {
            Wait w2{wg->turn, 8, units_done, wg, &res->polls, &res->raised, &res->paused};
            if (!bounded_wait(w2, WAIT_SPINS, true)) break;
            wg->stg[wv].store(seq, std::memory_order_release);
        }
    }
}

static std::vector<Result> run(int nitems, int lose_item, int lose_wave, unsigned stall0, unsigned* stall_out) {
    std::atomic<unsigned> stall{stall0};
    Workgroup wg;
    wg.stall = &stall;
    std::vector<Result> res(8);
    std::vector<std::thread> th;
    for (int w = 0; w < 8; ++w) th.emplace_back(wave, &wg, w, nitems, lose_item, lose_wave, &res[w]);
    for (auto& t : th) t.join();
    *stall_out = stall.load();
    return res;
}

#define CHECK(c, ...) do { if (!(c)) { std::printf("FAILED: " __VA_ARGS__); std::printf("\n"); return 1; } } while (0)

int main() {
    unsigned stall = 0;
    {   // (1) clean run
        auto r = run(40, -1, -1, 0, &stall);
        CHECK(stall == 0, "clean run set the stall word");
        for (int w = 0; w < 8; ++w) CHECK(r[w].items == 40 && r[w].raised == 0, "clean run: wave %d did %d items, raised %d", w, r[w].items, r[w].raised);
    }
    for (int lw : {1, 6}) {   // (2) one lost publication, by a wave of either group
        auto r = run(40, 3, lw, 0, &stall);
        CHECK(stall == 1, "lost hand-shake (wave %d): stall word not set", lw);
        int raised = 0;
        for (int w = 0; w < 8; ++w) {
            raised += r[w].raised;
            // unsatisfied polls: at most one full timeout of each kind, plus the polls of the waits that were passed while
            // other waves were merely late (a clean run of this model stays below 2^14 of those per wave)
            CHECK(r[w].polls <= worst_case_polls_per_wave() + (1 << 14), "wave %d polled %ld times (bound %ld)", w, r[w].polls,
                  worst_case_polls_per_wave());
            CHECK(r[w].items <= 5, "wave %d went on for %d items after the loss at item 3", w, r[w].items);
        }
        CHECK(raised >= 1 && raised <= 8, "%d timeouts raised", raised);
    }
    {   // (3) the word is already set: nothing runs
        auto r = run(40, -1, -1, 1, &stall);
        for (int w = 0; w < 8; ++w) CHECK(r[w].items == 0 && r[w].polls == 0, "pre-set stall word: wave %d did %d items", w, r[w].items);
        CHECK(stall == 1, "stall word was cleared by the launch");
    }
    std::printf("ok\n");
    return 0;
}
