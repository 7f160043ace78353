// xw_sim.cpp -- fiber scheduler behind the simulation build of nlzm_amd/csrc/xw.h.
//
// TEST HARNESS ONLY.  Every lane of every wave is a fiber with its own stack; a wave's lanes run one after the
// other until each has reached the same cross-lane operation (or a pause, a workgroup barrier, or its end), then
// the operation is completed for all of them.  Role code therefore runs with exactly the control flow it has on the
// GPU, including divergence, and a cross-lane operation reached by only part of a wave is reported as an error.
#define NLZM_SIM 1
#include "../../nlzm_amd/csrc/xw.h"

#include <vector>

namespace xw {

static Sim g_sim;
Sim &sim() { return g_sim; }

// x86-64 SysV context switch: callee-saved registers on the old stack, swap stack pointers
asm(R"(
.text
.globl xw_switch
.type xw_switch,@function
xw_switch:
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
.size xw_switch,.-xw_switch
)");

static void fiber_main()
{
    g_sim.entry(g_sim.arg);
    for (;;) yield_to_sched(kDone);
}

static constexpr size_t kStack = 256 * 1024;

static void fiber_init(Fiber &f)
{
    f.stack = (char *)malloc(kStack);
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void **sp = (void **)(top - 16);
    sp[0] = (void *)&fiber_main;        // `ret` of the first switch lands here, with rsp = 8 (mod 16) as after a call
    sp[1] = nullptr;
    sp -= 6;                            // r15 r14 r13 r12 rbx rbp
    for (int i = 0; i < 6; i++) sp[i] = nullptr;
    f.sp = sp;
    f.state = kReady;
}

static void fail(const char *msg, Wave &w)
{
    fprintf(stderr, "xw_sim: %s (block %u wave %u): lane states", msg, w.blk->index, w.index);
    for (int l = 0; l < 64; l++) fprintf(stderr, " %d/%d", w.f[l].state, w.f[l].coll);
    fprintf(stderr, "\n");
    exit(3);
}

// run a wave until it pauses, reaches a workgroup barrier or ends; true if it made progress other than pausing
static bool run_wave(Wave &w)
{
    bool progress = false;
    for (;;) {
        for (uint32_t l = 0; l < 64; l++) {
            Fiber &f = w.f[l];
            if (f.state != kReady) continue;
            g_sim.cw = &w; g_sim.cl = l;
            xw_switch(&g_sim.sched_sp, f.sp);
        }
        int st = -1, kind = 0;
        uint32_t live = 0;
        for (uint32_t l = 0; l < 64; l++) {
            const Fiber &f = w.f[l];
            if (f.state == kDone) continue;
            live++;
            if (st < 0) { st = f.state; kind = f.coll; }
            else if (st != f.state || (st == kColl && kind != f.coll)) fail("lanes of a wave wait at different operations", w);
        }
        if (!live) { w.done = true; return true; }
        if (st == kColl) {
            w.colls++;
            progress = true;
            if (kind == cBallot) {
                unsigned long long m = 0;
                for (uint32_t l = 0; l < 64; l++) if (w.f[l].state == kColl && w.f[l].in) m |= 1ull << l;
                for (uint32_t l = 0; l < 64; l++) w.f[l].out = m;
            } else if (kind == cReadlane) {
                uint32_t src = 64;
                for (uint32_t l = 0; l < 64; l++) if (w.f[l].state == kColl) {
                    const uint32_t s = w.f[l].src == 0xFFFFFFFFu ? 64u : w.f[l].src;
                    if (src == 64 && s != 64) src = s;
                    if (s != 64 && s != src) fail("readlane with a non-uniform lane index", w);
                }
                if (src == 64) { for (uint32_t l = 0; l < 64; l++) if (w.f[l].state == kColl) { src = l; break; } }   // readfirstlane
                if (src >= 64) fail("readlane of lane >= 64", w);
                const unsigned long long v = w.f[src].in;       // (an exited lane's last value, as on the hardware: undefined there)
                for (uint32_t l = 0; l < 64; l++) w.f[l].out = v;
            } else if (kind == cShfl) {
                unsigned long long tmp[64];
                for (uint32_t l = 0; l < 64; l++) tmp[l] = w.f[l].in;
                for (uint32_t l = 0; l < 64; l++) if (w.f[l].state == kColl) w.f[l].out = tmp[w.f[l].src & 63u];
            }
            for (uint32_t l = 0; l < 64; l++) if (w.f[l].state == kColl) w.f[l].state = kReady;
            continue;
        }
        if (st == kPause) {
            for (uint32_t l = 0; l < 64; l++) if (w.f[l].state == kPause) w.f[l].state = kReady;
            return progress;
        }
        if (st == kBarrier) { w.at_barrier = true; return true; }
        fail("unexpected lane state", w);
    }
}

void launch(uint32_t nblocks, uint32_t nthreads, const unsigned long long *lds_bytes, void (*entry)(void *), void *arg)
{
    std::vector<Block> blocks(nblocks);
    const uint32_t nw = (nthreads + 63) / 64;
    for (uint32_t b = 0; b < nblocks; b++) {
        blocks[b].index = b; blocks[b].nwaves = nw;
        blocks[b].waves = new Wave[nw];
        blocks[b].lds = calloc(1, lds_bytes[b] ? lds_bytes[b] : 16);
        if (const char *e = getenv("NLZM_SIM_POISON")) {       // LDS is not cleared on the device either
            uint32_t l = (uint32_t)atoi(e) * 40503u + b + 7;
            for (unsigned long long k = 0; k + 4 <= lds_bytes[b]; k += 4) { l = l * 1664525u + 1013904223u; memcpy((char *)blocks[b].lds + k, &l, 4); }
        }
        for (uint32_t k = 0; k < nw; k++) {
            Wave &w = blocks[b].waves[k];
            w.blk = &blocks[b]; w.index = k;
            for (uint32_t l = 0; l < 64; l++) {
                if (k * 64 + l < nthreads) fiber_init(w.f[l]);
                else w.f[l].state = kDone;
            }
        }
    }
    g_sim.blocks = blocks.data(); g_sim.nblocks = nblocks; g_sim.entry = entry; g_sim.arg = arg;
    unsigned long long idle_sweeps = 0;
    const char *wd = getenv("NLZM_SIM_IDLE_SWEEPS");
    const unsigned long long idle_max = wd ? strtoull(wd, nullptr, 10) : 2000000ull;
    for (;;) {
        bool all_done = true, progress = false;
        for (uint32_t b = 0; b < nblocks; b++) {
            Block &B = blocks[b];
            uint32_t at_bar = 0, alive = 0;
            for (uint32_t k = 0; k < nw; k++) {
                Wave &w = B.waves[k];
                if (w.done) continue;
                alive++;
                if (!w.at_barrier) progress |= run_wave(w);
                if (w.at_barrier) at_bar++;
            }
            if (alive) all_done = false;
            if (alive && at_bar == alive) {         // workgroup barrier complete (waves that ended do not take part)
                for (uint32_t k = 0; k < nw; k++) {
                    Wave &w = B.waves[k];
                    if (w.done || !w.at_barrier) continue;
                    w.at_barrier = false;
                    for (uint32_t l = 0; l < 64; l++) if (w.f[l].state == kBarrier) w.f[l].state = kReady;
                }
                progress = true;
            }
        }
        g_sim.now++;
        if (all_done) break;
        idle_sweeps = progress ? 0 : idle_sweeps + 1;
        if (idle_sweeps > idle_max) {
            fprintf(stderr, "xw_sim: no progress for %llu sweeps (every wave is waiting): deadlock\n", idle_sweeps);
            exit(4);
        }
    }
    for (uint32_t b = 0; b < nblocks; b++) {
        for (uint32_t k = 0; k < nw; k++) for (uint32_t l = 0; l < 64; l++) free(blocks[b].waves[k].f[l].stack);
        delete[] blocks[b].waves;
        free(blocks[b].lds);
    }
}

}  // namespace xw
