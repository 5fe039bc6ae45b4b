// hip_emu.cpp -- the scheduler of tests/emu/hip_emu.h: blocks run one after the other, the lanes of a block are ucontext fibers on one OS
// thread, switched only at rendezvous points (wave-level operations, __syncthreads) and at a lane's end.  TEST INFRASTRUCTURE.
#include <setjmp.h>
#include <ucontext.h>
#include <cstdio>
#include <mutex>
#include <vector>
#include "hip_emu.h"
#undef threadIdx
#undef blockIdx
#undef blockDim
#undef gridDim

namespace emu {
namespace {
const size_t kStack = 1 << 20;
// a lane's fiber lives for the whole process: created once (makecontext), entered the first time with swapcontext, afterwards switched
// with _setjmp / _longjmp (no signal-mask system calls); between launches it waits at the end of lane_main's loop
struct Lane { ucontext_t ctx; jmp_buf jb; unsigned char *stack = nullptr; int tid = 0; bool done = false, started = false; };
struct Wave { uint64_t slot[64], pub[64]; int site[64]; bool div[64]; uint64_t present = 0, pubMask = 0; int arrived = 0, live = 0; };
std::vector<Lane *> lanes;
jmp_buf schedJb;
std::vector<Wave> waves;
ucontext_t sched;
Lane *cur = nullptr;
Idx g_block, g_bdim, g_gdim;
int barArrived = 0, barLive = 0;
unsigned barGen = 0;
long progress = 0;  // bumped whenever a rendezvous completes or a lane ends: the deadlock detector's clock
const std::function<void()> *g_body = nullptr;

void yield() { Lane *me = cur; if (_setjmp(me->jb) == 0) _longjmp(schedJb, 1); }
void resume(Lane *l) {
    if (_setjmp(schedJb) != 0) return;  // the lane yielded
    cur = l;
    if (l->started) _longjmp(l->jb, 1);
    l->started = true;
    swapcontext(&sched, &l->ctx);
}
const char *g_kernel = "";
// Every live lane of the wave waits at a wave-level call.  All at the same one: serve them.  Otherwise the lanes at a call marked as
// sitting in divergent code go first (the hardware would be executing their branch while the rest is masked off).
void resolve(Wave &w) {
    int first = -1, chosen = -1;
    bool same = true;
    for (int l = 0; l < 64; ++l) if ((w.present >> l) & 1) {
        if (first < 0) first = w.site[l];
        else if (w.site[l] != first) same = false;
        if (w.div[l] && chosen < 0) chosen = w.site[l];
    }
    if (same) chosen = first;
    else if (chosen < 0) {
        fprintf(stderr, "hip_emu: %s: the lanes of a wave wait at different wave-level calls and none of them is marked as divergent:", g_kernel);
        for (int l = 0; l < 64; ++l) if ((w.present >> l) & 1) fprintf(stderr, " %d", w.site[l]);
        fprintf(stderr, "\n");
        abort();
    }
    uint64_t group = 0;
    for (int l = 0; l < 64; ++l) if (((w.present >> l) & 1) && w.site[l] == chosen) { group |= 1ull << l; w.pub[l] = w.slot[l]; }
    static const bool debug = getenv("HIP_EMU_DEBUG") != nullptr;
    if (debug) fprintf(stderr, "  resolve: site %d group %016llx live %d waiting %d\n", chosen, (unsigned long long)group, w.live, w.arrived);
    w.pubMask = group; w.present &= ~group; w.arrived -= __builtin_popcountll(group); ++progress;
}
void lane_main() {
    for (;;) {
        (*g_body)();
        cur->done = true;
        Wave &w = waves[cur->tid / 64];
        --w.live; --barLive; ++progress;
        if (w.arrived > 0 && w.arrived == w.live) resolve(w);          // the others were waiting for this lane only
        if (barArrived > 0 && barArrived == barLive) { barArrived = 0; ++barGen; }
        yield();  // until the next block / launch gives this lane a new body
    }
}
}  // namespace

unsigned char *dyn_shared = nullptr;
Idx thread_idx() { return Idx{(unsigned)cur->tid, 0, 0}; }
Idx block_idx() { return g_block; }
Idx block_dim() { return g_bdim; }
Idx grid_dim() { return g_gdim; }
int lane() { return cur->tid & 63; }

const uint64_t *wave_gather(uint64_t v, uint64_t *mask, int site, bool divergent) {
    Wave &w = waves[cur->tid / 64];
    const int l = cur->tid & 63;
    w.slot[l] = v; w.site[l] = site; w.div[l] = divergent; w.present |= 1ull << l; ++w.arrived;
    if (w.arrived == w.live) resolve(w);
    while ((w.present >> l) & 1) yield();
    *mask = w.pubMask;
    return w.pub;
}
void block_barrier() {
    const unsigned g = barGen;
    ++barArrived;
    if (barArrived == barLive) { barArrived = 0; ++barGen; ++progress; }
    else while (barGen == g) yield();
}

void launch(const char *name, dim3 grid, dim3 block, size_t dynShared, const std::function<void()> &body) {
    // one launch at a time: the host code renders on several "devices" from one thread each (pg_render_sharded), the scheduler's state and
    // the kernels' statics (__shared__) are per process
    static std::mutex oneLaunch;
    std::lock_guard<std::mutex> lock(oneLaunch);
    static const bool trace = getenv("HIP_EMU_TRACE") != nullptr;
    if (trace) fprintf(stderr, "hip_emu: %s <<<%u, %u, %zu>>>\n", name, grid.x, block.x, dynShared);
    g_kernel = name;
    if (grid.y != 1 || grid.z != 1 || block.y != 1 || block.z != 1) { fprintf(stderr, "hip_emu: only 1-D launches\n"); abort(); }
    const int nt = (int)block.x;
    while ((int)lanes.size() < nt) {
        Lane *L = new Lane;
        L->tid = (int)lanes.size();
        L->stack = (unsigned char *)malloc(kStack);
        getcontext(&L->ctx);
        L->ctx.uc_stack.ss_sp = L->stack; L->ctx.uc_stack.ss_size = kStack; L->ctx.uc_link = nullptr;
        makecontext(&L->ctx, (void (*)())lane_main, 0);
        lanes.push_back(L);
    }
    static std::vector<unsigned char> shared;
    if (shared.size() < dynShared + 64) shared.resize(dynShared + 64);
    dyn_shared = shared.data();
    g_body = &body;
    g_bdim = Idx{block.x, 1, 1}; g_gdim = Idx{grid.x, 1, 1};
    for (unsigned b = 0; b < grid.x; ++b) {
        g_block = Idx{b, 0, 0};
        waves.assign((nt + 63) / 64, Wave());
        for (int t = 0; t < nt; ++t) { lanes[t]->done = false; ++waves[t / 64].live; }
        barLive = nt; barArrived = 0;
        int remaining = nt;
        long lastProgress = progress; int idleRounds = 0;
        while (remaining > 0) {
            remaining = 0;
            for (int t = 0; t < nt; ++t) {
                if (lanes[t]->done) continue;
                resume(lanes[t]);
                if (!lanes[t]->done) ++remaining;
            }
            if (progress == lastProgress) {
                if (++idleRounds > 4) {
                    fprintf(stderr, "hip_emu: deadlock in block %u: %d lanes wait at a rendezvous the others never reach (a wave-level call or __syncthreads in divergent code)\n", b, remaining);
                    for (size_t w = 0; w < waves.size(); ++w) fprintf(stderr, "  wave %zu: arrived %d of %d live\n", w, waves[w].arrived, waves[w].live);
                    fprintf(stderr, "  barrier: arrived %d of %d live\n", barArrived, barLive);
                    abort();
                }
            } else { idleRounds = 0; lastProgress = progress; }
        }
    }
    cur = nullptr;
}
}  // namespace emu
