// xw.h -- the execution layer the pipeline roles (nlzm_v2.h) are written against.
//
// Role code is plain SIMT code: one flow of control per lane, 64 lanes per wave, cross-lane
// operations (ballot, readlane, shuffles), LDS, agent-scope hand-off words.  Two builds of the
// SAME role source exist:
//   * hipcc, gfx950: every function below is one or two instructions (device build, the product);
//   * g++ -DNLZM_SIM: every lane is a fiber (tests/host_sim/sim2.cpp schedules them), cross-lane
//     operations are rendezvous points of a wave's 64 fibers.  TEST HARNESS ONLY -- it exists so
//     that the role logic can be checked against the oracle on a machine without a GPU.
//
// Contract for role code: cross-lane operations are called in wave-uniform control flow (all live
// lanes of the wave reach the same call); spin-waits call xw::pause() between polls.
#pragma once

#include <stdint.h>

#ifndef NLZM_SIM
// =====================================================================================
// device build (gfx950)
// =====================================================================================
#include <hip/hip_runtime.h>
#define XW_FN __device__ __forceinline__
#define XW_NOINLINE __device__ __noinline__

namespace xw {

XW_FN uint32_t lane() { return threadIdx.x & 63u; }
XW_FN uint32_t wave() { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }    // (a scalar: what depends on it branches, not masks)
XW_FN uint32_t thread() { return threadIdx.x; }
XW_FN unsigned long long ballot(bool p) { return __ballot(p); }
XW_FN bool any(bool p) { return __ballot(p) != 0; }
XW_FN uint32_t readlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
XW_FN uint32_t readfirst(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
XW_FN unsigned long long readlane64(unsigned long long v, uint32_t l)
{
    return ((unsigned long long)readlane((uint32_t)(v >> 32), l) << 32) | readlane((uint32_t)v, l);
}
XW_FN unsigned long long readfirst64(unsigned long long v)
{
    return ((unsigned long long)readfirst((uint32_t)(v >> 32)) << 32) | readfirst((uint32_t)v);
}
// value of lane `src` (per-lane src; ds_bpermute)
XW_FN uint32_t shfl(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v); }
XW_FN unsigned long long shfl64(unsigned long long v, uint32_t src)
{
    return ((unsigned long long)shfl((uint32_t)(v >> 32), src) << 32) | shfl((uint32_t)v, src);
}
// value of lane - d (own value for the first d lanes)
XW_FN uint32_t shfl_up(uint32_t v, uint32_t d) { const uint32_t l = lane(); return shfl(v, l >= d ? l - d : l); }
XW_FN unsigned long long shfl_up64(unsigned long long v, uint32_t d) { const uint32_t l = lane(); return shfl64(v, l >= d ? l - d : l); }
// inclusive prefix scans over the 64 lanes with DPP row shifts / row broadcasts (no LDS round trips): lanes that a
// step does not reach get the identity
#define XW_DPP(old, v, ctrl, rm) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, rm, 0xF, false))
XW_FN uint32_t scan_max(uint32_t v)
{
    uint32_t t;
    t = XW_DPP(0u, v, 0x111, 0xF); v = v > t ? v : t;
    t = XW_DPP(0u, v, 0x112, 0xF); v = v > t ? v : t;
    t = XW_DPP(0u, v, 0x114, 0xF); v = v > t ? v : t;
    t = XW_DPP(0u, v, 0x118, 0xF); v = v > t ? v : t;
    t = XW_DPP(0u, v, 0x142, 0xA); v = v > t ? v : t;       // row_bcast:15 into rows 1, 3
    t = XW_DPP(0u, v, 0x143, 0xC); v = v > t ? v : t;       // row_bcast:31 into rows 2, 3
    return v;
}
XW_FN int32_t scan_min_i32(int32_t x)
{
    const uint32_t id = 0x7FFFFFFFu;
    uint32_t v = (uint32_t)x, t;
    t = XW_DPP(id, v, 0x111, 0xF); v = (int32_t)t < (int32_t)v ? t : v;
    t = XW_DPP(id, v, 0x112, 0xF); v = (int32_t)t < (int32_t)v ? t : v;
    t = XW_DPP(id, v, 0x114, 0xF); v = (int32_t)t < (int32_t)v ? t : v;
    t = XW_DPP(id, v, 0x118, 0xF); v = (int32_t)t < (int32_t)v ? t : v;
    t = XW_DPP(id, v, 0x142, 0xA); v = (int32_t)t < (int32_t)v ? t : v;
    t = XW_DPP(id, v, 0x143, 0xC); v = (int32_t)t < (int32_t)v ? t : v;
    return (int32_t)v;
}
XW_FN uint32_t scan_add(uint32_t v)
{
    v += XW_DPP(0u, v, 0x111, 0xF);
    v += XW_DPP(0u, v, 0x112, 0xF);
    v += XW_DPP(0u, v, 0x114, 0xF);
    v += XW_DPP(0u, v, 0x118, 0xF);
    v += XW_DPP(0u, v, 0x142, 0xA);
    v += XW_DPP(0u, v, 0x143, 0xC);
    return v;
}
// value of the lane below (lane 0: fill)
XW_FN uint32_t lane_below(uint32_t v, uint32_t fill) { return XW_DPP(fill, v, 0x138, 0xF); }     // wave_shr:1
// LDS written by other lanes of this wave is visible after this point (DS operations of a wave execute in order)
XW_FN void wave_sync() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
// this wave's global stores have landed / its loads returned
XW_FN void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
XW_FN void block_sync() { __syncthreads(); }
XW_FN void pause() { __builtin_amdgcn_s_sleep(1); }
XW_FN void pause_long() { __builtin_amdgcn_s_sleep(16); }
// agent-scope (sc1: write-through / L1-bypassing) accesses for words shared between workgroups
XW_FN void st_agent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
XW_FN uint32_t ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
XW_FN void st_agent64(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
XW_FN unsigned long long ld_agent64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// sixteen bytes (16-byte aligned) in ONE write-through store: lands whole, and costs what a plain store does where four 4-byte
// sc1 stores are four fabric writes.  (s_nop: the data registers of a store of more than eight bytes may not be written in the
// cycle after it -- a hazard the compiler resolves for its own stores, not for this one.)
XW_FN void st_agent128(uint32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 v = { a, b, c, d };
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"((__attribute__((address_space(1))) uint32_t *)p), "v"(v) : "memory");
}
// sixteen bytes (16-byte aligned) in ONE sc1 load: a record quad as its writer's one store left it (the wait is part of it: the compiler
// does not count an asm load)
struct u32x4 { uint32_t x, y, z, w; };
XW_FN u32x4 ld_agent128(const uint32_t *p)
{
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"((const __attribute__((address_space(1))) uint32_t *)p) : "memory");
    return u32x4{ v.x, v.y, v.z, v.w };
}
XW_FN void atomic_or_agent(uint32_t *p, uint32_t v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
XW_FN void atomic_add64_agent(unsigned long long *p, unsigned long long v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// *p = v if *p == expect; returns what was there
XW_FN uint32_t cas_agent(uint32_t *p, uint32_t expect, uint32_t v)
{
    (void)__hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expect;
}
// after a poll saw the flag: later plain loads of this wave read what the producer stored (buffer_inv sc1)
XW_FN void acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// after a poll saw its word: nothing that follows in the program may be moved in front of the poll by the compiler (the
// payload is read with sc1 loads, which need no cache invalidate: guide, Guideline 16)
XW_FN void after_poll() { asm volatile("" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
XW_FN void lds_min64(unsigned long long *p, unsigned long long v)
{
    (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
XW_FN void lds_add64(unsigned long long *p, unsigned long long v)
{
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
XW_FN void lds_or(uint32_t *p, uint32_t v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
XW_FN uint32_t lds_inc(uint32_t *p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
XW_FN void lds_add(uint32_t *p, uint32_t v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
XW_FN void lds_max(uint32_t *p, uint32_t v) { (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
XW_FN void lds_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
XW_FN uint32_t lds_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// the value as it is, but not to be reasoned about (keeps the compiler from turning a choice between two members of a
// stage's state into ONE indexed access, which would put the whole state into scratch memory)
XW_FN uint32_t opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
XW_FN unsigned long long clock100() { return wall_clock64(); }          // 100 MHz
XW_FN unsigned long long tick() { return __builtin_readcyclecounter(); }
XW_FN void need_bt(void *, uint32_t) {}                                 // (simulation hooks: worker lanes run lazily there;
XW_FN void trace(int, uint32_t = 0, uint32_t = 0, uint32_t = 0, uint32_t = 0, uint32_t = 0, uint32_t = 0, uint32_t = 0) {}   //  stage traces;
XW_FN uint32_t test_cut(uint32_t n) { return n; }                       //  parser blocks cut at random)
template <class T> XW_FN T *lds();                                      // defined by the kernel file (one __shared__ image)

}  // namespace xw

#else
// =====================================================================================
// simulation build: lanes are fibers (tests only)
// =====================================================================================
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define XW_FN inline
#define XW_NOINLINE

namespace xw {

enum : int { kReady = 0, kColl = 1, kPause = 2, kBarrier = 3, kDone = 4 };
enum : int { cBallot = 1, cReadlane = 2, cShfl = 3, cSync = 4 };

struct Fiber {
    void *sp = nullptr;
    char *stack = nullptr;
    int state = kReady;
    int coll = 0;
    unsigned long long in = 0, out = 0;
    uint32_t src = 0;
};
struct Block;
struct Wave {
    Fiber f[64];
    Block *blk = nullptr;
    uint32_t index = 0;
    bool done = false, at_barrier = false;
    unsigned long long colls = 0;
};
struct Block {
    Wave *waves = nullptr;
    uint32_t nwaves = 0, index = 0;
    void *lds = nullptr;
};
struct Sim {
    Block *blocks = nullptr;
    uint32_t nblocks = 0;
    void (*entry)(void *) = nullptr;
    void *arg = nullptr;
    void *sched_sp = nullptr;
    Wave *cw = nullptr;
    uint32_t cl = 0;
    unsigned long long now = 0;     // fake clock: scheduler sweeps
};
Sim &sim();
extern "C" void xw_switch(void **from_sp, void *to_sp);

inline Fiber &cur() { return sim().cw->f[sim().cl]; }
inline void yield_to_sched(int st)
{
    Fiber &f = cur();
    f.state = st;
    xw_switch(&f.sp, sim().sched_sp);
}
inline unsigned long long collective(int kind, unsigned long long in, uint32_t src)
{
    Fiber &f = cur();
    f.coll = kind; f.in = in; f.src = src;
    yield_to_sched(kColl);
    return f.out;
}

inline uint32_t lane() { return sim().cl; }
inline uint32_t wave() { return sim().cw->index; }
inline uint32_t thread() { return sim().cw->index * 64 + sim().cl; }
inline uint32_t block_index() { return sim().cw->blk->index; }
inline unsigned long long ballot(bool p) { return collective(cBallot, p ? 1 : 0, 0); }
inline bool any(bool p) { return ballot(p) != 0; }
inline unsigned long long readlane64(unsigned long long v, uint32_t l) { return collective(cReadlane, v, l); }
inline uint32_t readlane(uint32_t v, uint32_t l) { return (uint32_t)collective(cReadlane, v, l); }
inline unsigned long long readfirst64(unsigned long long v) { return collective(cReadlane, v, 0xFFFFFFFFu); }
inline uint32_t readfirst(uint32_t v) { return (uint32_t)collective(cReadlane, v, 0xFFFFFFFFu); }
inline unsigned long long shfl64(unsigned long long v, uint32_t src) { return collective(cShfl, v, src & 63u); }
inline uint32_t shfl(uint32_t v, uint32_t src) { return (uint32_t)collective(cShfl, v, src & 63u); }
inline uint32_t shfl_up(uint32_t v, uint32_t d) { const uint32_t l = lane(); return shfl(v, l >= d ? l - d : l); }
inline unsigned long long shfl_up64(unsigned long long v, uint32_t d) { const uint32_t l = lane(); return shfl64(v, l >= d ? l - d : l); }
inline uint32_t scan_max(uint32_t v) { for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t o = shfl_up(v, d); if (lane() >= d && o > v) v = o; } return v; }
inline int32_t scan_min_i32(int32_t v) { for (uint32_t d = 1; d < 64; d <<= 1) { const int32_t o = (int32_t)shfl_up((uint32_t)v, d); if (lane() >= d && o < v) v = o; } return v; }
inline uint32_t scan_add(uint32_t v) { for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t o = shfl_up(v, d); if (lane() >= d) v += o; } return v; }
inline uint32_t lane_below(uint32_t v, uint32_t fill) { const uint32_t o = shfl_up(v, 1); return lane() ? o : fill; }
inline void wave_sync() { (void)collective(cSync, 0, 0); }
inline void drain() {}
inline void block_sync() { yield_to_sched(kBarrier); }
inline void pause() { yield_to_sched(kPause); }
inline void pause_long() { yield_to_sched(kPause); }
inline void st_agent(uint32_t *p, uint32_t v) { *p = v; }
inline uint32_t ld_agent(const uint32_t *p) { return *(volatile const uint32_t *)p; }
inline void st_agent64(unsigned long long *p, unsigned long long v) { *p = v; }
inline unsigned long long ld_agent64(const unsigned long long *p) { return *(volatile const unsigned long long *)p; }
inline void st_agent128(uint32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
struct u32x4 { uint32_t x, y, z, w; };
inline u32x4 ld_agent128(const uint32_t *p) { const volatile uint32_t *q = p; return u32x4{ q[0], q[1], q[2], q[3] }; }
inline void atomic_or_agent(uint32_t *p, uint32_t v) { *p |= v; }
inline void atomic_add64_agent(unsigned long long *p, unsigned long long v) { *p += v; }
inline uint32_t cas_agent(uint32_t *p, uint32_t expect, uint32_t v) { const uint32_t o = *p; if (o == expect) *p = v; return o; }
inline void acquire_agent() {}
inline void after_poll() {}
inline void lds_min64(unsigned long long *p, unsigned long long v) { if (v < *p) *p = v; }
inline void lds_add64(unsigned long long *p, unsigned long long v) { *p += v; }
inline void lds_or(uint32_t *p, uint32_t v) { *p |= v; }
inline uint32_t lds_inc(uint32_t *p) { return (*p)++; }
inline void lds_add(uint32_t *p, uint32_t v) { *p += v; }
inline void lds_max(uint32_t *p, uint32_t v) { if (v > *p) *p = v; }
inline void lds_st(uint32_t *p, uint32_t v) { *p = v; }
inline uint32_t lds_ld(const uint32_t *p) { return *(volatile const uint32_t *)p; }
inline uint32_t opaque(uint32_t v) { return v; }
inline unsigned long long clock100() { return sim().now; }
inline unsigned long long tick() { return sim().now; }
void need_bt(void *user, uint32_t a);                                   // defined by the simulator: run the worker lanes up to `a`
void trace(int what, uint32_t a = 0, uint32_t b = 0, uint32_t c = 0, uint32_t d = 0, uint32_t e = 0, uint32_t f = 0, uint32_t g = 0);   // NLZM_SIM_TRACE
uint32_t test_cut(uint32_t n);                                          // NLZM_SIM_RANDOM_BLOCKS
template <class T> inline T *lds() { return (T *)sim().cw->blk->lds; }

// run `entry(arg)` on nblocks x nthreads lanes (block b gets lds_bytes[b] bytes of zeroed LDS); returns when every lane
// has returned.  watchdog: scheduler sweeps without any lane making progress other than pausing.
void launch(uint32_t nblocks, uint32_t nthreads, const unsigned long long *lds_bytes, void (*entry)(void *), void *arg);

}  // namespace xw
#endif
