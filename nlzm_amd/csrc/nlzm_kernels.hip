// nlzm_kernels.hip -- gfx950 kernels of the NLZM compress path.
//
//   rk_hash_kernel      RK256 rolling hash of every 256-byte window, one thread per
//                       position (closed form of MatchFinderRK256's roll, NLZM.cpp:798-799,
//                       1071-1083): pure, position-parallel, input tile staged in LDS.
//   prefilter_*_kernel  marks the positions whose BT4 call cannot be decided from the input alone
//   bin_kernel          groups each chunk's positions by BT4 hash head
//   pipeline_kernel     block 0: the serial half (nlzm_core.h, seven waves): HT2/HT3/RK256 state, match-table
//                       chain, forward-graph parse, model, symbol emit; blocks 1..: BT4 worker lanes
//   pipeline_multi_kernel  the same for several independent streams in one launch (block mode)
//   rans_frames_kernel  CodeFrame::Flush (NLZM.cpp:590-640): 4 interleaved rANS states
//                       per frame, renormalisation words placed by a prefix scan.
//   gather_frames_kernel concatenates the frames into the output stream.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nlzm_core.h"
#include "nlzm_v2.h"

namespace nlzm {

// LDS image of the three-stage pipeline (nlzm_v2.h): every block of pipeline2_kernel has ONE role, so the roles share
// the bytes.  A file-scope __shared__ object: every access is a ds_* instruction.
constexpr uint32_t kLogCap = 24;                    // stores a worker lane notes down during a dry run
union V2Lds {
    v2::FLds f; v2::TLds t; v2::PLds p;
    uint32_t wlog[512 * kLogCap * 2];
};
__shared__ V2Lds g_v2_lds;
}  // namespace nlzm
namespace xw {
template <class T> XW_FN T *lds() { return reinterpret_cast<T *>(&nlzm::g_v2_lds); }
}
namespace nlzm {

// ---------------------------------------------------------------------------
// 64-lane wave policy for the master (block = one wave)
// ---------------------------------------------------------------------------
// LDS image of the master: a file-scope __shared__ object, so every access is a ds_* instruction
// (a pointer kept in a struct would degrade to flat_* accesses and their full waits)
__shared__ MasterLds g_master_lds;

struct DevWave {
    static __device__ __forceinline__ MasterLds *lds() { return &g_master_lds; }
    static __device__ __forceinline__ void cnt_add(unsigned long long *p, unsigned long long v)
    {
        if (lane() == 0) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // lane i < n adds f(i) (one exec-masked LDS atomic for all of them)
    template <class F>
    static __device__ __forceinline__ void cnt_add_fn(unsigned long long *p, uint32_t n, F f)
    {
        const uint32_t v = f(lane());
        if (lane() < n && v) (void)__hip_atomic_fetch_add(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // One symbol through a nibble CDF (NLZM.cpp:348-382, 435-438): lane i holds cell[i] and cell[i+1]; the (start, freq)
    // snapshot of symbol y is picked out of the registers, cell[i] += (mixin[y][i] - cell[i]) >> 7 for i < nsy with
    // mixin[y][i] = i <= y ? i : 16384 + i + (127 - nsy) (:284-298), and the price row follows from the new cells.
    static __device__ __forceinline__ void cdf_step(uint16_t *cell, uint16_t *price_row, const uint16_t *lut, uint32_t nsy, uint32_t y,
                                                    uint32_t &start, uint32_t &freq)
    {
        const uint32_t i = lane();
        uint32_t c0 = 0, c1 = 0;
        if (i <= 16) { c0 = cell[i]; c1 = cell[i + 1]; }
        start = (uint32_t)__builtin_amdgcn_readlane((int)c0, (int)y);
        freq = (uint32_t)__builtin_amdgcn_readlane((int)c1, (int)y) - start;
        auto upd = [=](uint32_t j, uint32_t c) {
            const int mix = (j <= y) ? (int)j : (int)(16384 + j + (127 - nsy));
            return j < nsy ? (uint32_t)(uint16_t)(c + (uint32_t)((mix - (int)c) >> 7)) : c;
        };
        const uint32_t n0 = upd(i, c0), n1 = upd(i + 1, c1);
        if (i < nsy) { cell[i] = (uint16_t)n0; price_row[i] = lut[(n1 - n0) >> 6]; }
    }
    // Up to 8 symbols of distinct contexts (list in LDS: context, symbol), four per pass on sixteen lanes each: see
    // cdf_step for one symbol.  The lane that holds the symbol's cell writes its (freq << 16) + start word.
    static __device__ __forceinline__ void cdf_multi(uint16_t *cdf, uint16_t *price, const uint16_t *lut, const uint32_t *list, uint32_t n,
                                                     uint32_t *out)
    {
        const uint32_t grp = lane() >> 4, i = lane() & 15u;
        for (uint32_t base = 0; base < n; base += 4) {
            const uint32_t k = base + grp;
            if (k < n) {
                const uint32_t ctx = list[2 * k], y = list[2 * k + 1];
                const uint32_t nsy = ctx == kCtxCmd ? 4u : ((ctx == kCtxLenDirect || ctx >= kCtxSlotHi) ? 8u : 16u);
                uint16_t *cell = cdf + ctx * kCdfStride;
                const uint32_t c0 = cell[i], c1 = cell[i + 1];
                if (i == y) out[k] = ((c1 - c0) << 16) + c0;
                auto upd = [=](uint32_t j, uint32_t c) {
                    const int mix = (j <= y) ? (int)j : (int)(16384 + j + (127 - nsy));
                    return j < nsy ? (uint32_t)(uint16_t)(c + (uint32_t)((mix - (int)c) >> 7)) : c;
                };
                const uint32_t n0 = upd(i, c0), n1 = upd(i + 1, c1);
                if (i < nsy) { cell[i] = (uint16_t)n0; price[ctx * 16 + i] = lut[(n1 - n0) >> 6]; }
            }
        }
    }
    // value of lane l (wave-uniform l)
    static __device__ __forceinline__ uint32_t pick(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
    static __device__ __forceinline__ uint32_t lane() { return threadIdx.x & 63u; }
    static __device__ __forceinline__ uint32_t width() { return 64u; }
    // per look-ahead slot state kept in the lane that evaluated the slot (slot j = lane j, kPf == 64)
    struct PfLane { uint32_t idx, rkslot, stale, v4, row1, sl, sd, cmpb, simple, wrote; };
    // idx: HT2 bucket | HT3 bucket << 16 (0xFFFFFFFF: the slot touches no HT row); sl/sd: summary of its table updates;
    // simple: everything about the slot was settled by the look-ahead (see Master::pf_fill)
    static __device__ __forceinline__ void pfl_set(PfLane &p, uint32_t, uint32_t idx, uint32_t rkslot, uint32_t v4, uint32_t row1,
                                                   uint32_t sl, uint32_t sd, uint32_t cmpb, bool simple)
    {
        p.idx = idx; p.rkslot = rkslot; p.stale = 0; p.v4 = v4; p.row1 = row1; p.sl = sl; p.sd = sd; p.cmpb = cmpb; p.simple = simple;
        p.wrote = 0;
    }
    // stale bit 1: an earlier slot of the batch writes an HT row this slot has read (bucket b owns rows b and b+1, :912)
    static __device__ __forceinline__ void pfl_conflicts(PfLane &p, uint32_t n)
    {
        const uint32_t o2 = p.idx & 0xFFFFu, o3 = p.idx >> 16;
        bool conf = false;
        for (uint32_t k = 0; k + 1 < n; k++) {
            const uint32_t ik = (uint32_t)__builtin_amdgcn_readlane((int)p.idx, (int)k);
            const uint32_t i2 = ik & 0xFFFFu, i3 = ik >> 16;
            conf = conf || (lane() > k && ik != 0xFFFFFFFFu && (o2 == i2 || o3 == i3 || o3 == i3 + 1 || o3 + 1 == i3));
        }
        p.stale |= (conf && lane() < n) ? 1u : 0u;
    }
    // slots the finder wave may pass over without looking at them one by one
    static __device__ __forceinline__ unsigned long long pfl_run_mask(const PfLane &p, uint32_t n)
    {
        return __ballot(lane() < n && p.simple && !(p.stale & 5u));
    }
    // HT2/HT3 rows of slots [s0, s0+cnt) rotate (:935-936); no two of them, nor an earlier slot of the batch, share a row
    static __device__ __forceinline__ void pfl_run_store(PfLane &p, uint32_t s0, uint32_t cnt, uint32_t *ht2, uint32_t *ht3,
                                                         uint32_t q0, uint32_t wbits, uint32_t tag_mask, uint32_t ht3_shift)
    {
        if (lane() >= s0 && lane() < s0 + cnt) {
            const uint32_t q = q0 + (lane() - s0);
            const uint32_t h2 = hash4(p.v4 & 0xFFFFu), h3 = hash4(p.v4 & 0xFFFFFFu);
            const uint32_t i2 = h2 >> 20, i3 = h3 >> ht3_shift;
            ht2[i2] = q | ((h2 & tag_mask) << wbits);
            ht3[i3] = q | ((h3 & tag_mask) << wbits);
            ht3[i3 + 1] = p.row1;
            p.wrote = 1;
        }
    }
    // (called by the lane of slot j itself) the BT4 result of the slot has arrived after the look-ahead
    static __device__ __forceinline__ void pfl_update(PfLane &p, uint32_t, uint32_t sl, uint32_t sd, bool simple)
    {
        p.sl = sl; p.sd = sd; p.simple = simple;
    }
    // slot s has rotated its HT rows; v1 is what it moved into HT3 row bucket+1
    static __device__ __forceinline__ void pfl_wrote(PfLane &p, uint32_t s, uint32_t v1)
    {
        if (lane() == s) { p.wrote = 1; p.row1 = v1; }
    }
    // HT2 row i2 and HT3 rows i3, i3+1 as the slots before j of this batch have left them (row[] comes in as the
    // look-ahead read them).  Slot k stored: HT2[i2_k] = E2_k, HT3[i3_k] = E3_k, HT3[i3_k + 1] = row1_k (:935-936).
    static __device__ __forceinline__ void pfl_rows_now(const PfLane &p, uint32_t j, uint32_t i2, uint32_t i3, uint32_t q0, uint32_t wbits,
                                                        uint32_t tag_mask, uint32_t ht3_shift, uint32_t row[3])
    {
        const bool act = p.wrote && lane() < j;
        const uint32_t o2 = p.idx & 0xFFFFu, o3 = p.idx >> 16, qk = q0 + lane();
        const uint32_t e2 = qk | ((hash4(p.v4 & 0xFFFFu) & tag_mask) << wbits), e3 = qk | ((hash4(p.v4 & 0xFFFFFFu) & tag_mask) << wbits);
        const bool mA = act && o3 == i3, mB = act && o3 + 1 == i3, mC = act && o3 == i3 + 1;
        const unsigned long long b0 = __ballot(act && o2 == i2), b1 = __ballot(mA || mB), b2 = __ballot(mC || mA);
        (void)ht3_shift;
        if (b0) row[0] = (uint32_t)__builtin_amdgcn_readlane((int)e2, 63 - __builtin_clzll(b0));
        if (b1) row[1] = (uint32_t)__builtin_amdgcn_readlane((int)(mA ? e3 : p.row1), 63 - __builtin_clzll(b1));
        if (b2) row[2] = (uint32_t)__builtin_amdgcn_readlane((int)(mC ? e3 : p.row1), 63 - __builtin_clzll(b2));
    }
    static __device__ __forceinline__ uint32_t pfl_sl(const PfLane &p, uint32_t s) { return (uint32_t)__builtin_amdgcn_readlane((int)p.sl, (int)s); }
    static __device__ __forceinline__ uint32_t pfl_sd(const PfLane &p, uint32_t s) { return (uint32_t)__builtin_amdgcn_readlane((int)p.sd, (int)s); }
    static __device__ __forceinline__ uint32_t pfl_cmpb(const PfLane &p, uint32_t s) { return (uint32_t)__builtin_amdgcn_readlane((int)p.cmpb, (int)s); }
    static __device__ __forceinline__ void pfl_mark_rk(PfLane &p, uint32_t s, uint32_t n, uint32_t slot)
    {
        p.stale |= (lane() > s && lane() < n && p.rkslot == slot) ? 4u : 0u;
    }
    static __device__ __forceinline__ uint32_t pfl_stale(const PfLane &p, uint32_t s) { return (uint32_t)__builtin_amdgcn_readlane((int)p.stale, (int)s); }
    // explicit rep probes: lane i = probe i>>4, dword i&15 of the kRepPf bytes in front of the distance and at the position
    struct RepPf { uint32_t s, c; };
    static __device__ __forceinline__ RepPf rep_prefetch(const uint8_t *in, unsigned long long n, uint32_t a, uint32_t r0, uint32_t r1,
                                                         uint32_t r2, uint32_t r3)
    {
        const uint32_t i = lane(), k = i >> 4, j = (i & 15u) * 4;
        const uint32_t d = k == 0 ? r0 : (k == 1 ? r1 : (k == 2 ? r2 : r3));
        RepPf r;
        r.c = (unsigned long long)a + j + 4 <= n + 64 ? load32u(in + a + j) : 0;        // the input is followed by >= 128 bytes
        r.s = d <= a ? load32u(in + (a - d) + j) : ~r.c;                                 // distance beyond the start: no match
        return r;
    }
    static __device__ __forceinline__ void rep_lengths(RepPf r, uint32_t len[4])
    {
        const uint32_t x = r.s ^ r.c;
        const unsigned long long bal = __ballot(x != 0);
        const uint32_t nb = x ? ((uint32_t)__builtin_ctz(x) >> 3) : 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t f = (uint32_t)(bal >> (16 * k)) & 0xFFFFu;
            if (f) {
                const uint32_t l = (uint32_t)__builtin_ctz(f);
                len[k] = 4 * l + (uint32_t)__builtin_amdgcn_readlane((int)nb, 16 * k + (int)l);
            } else len[k] = kRepPf;
        }
    }
    // the value is the same in every lane: move it to a scalar register so that the arithmetic and
    // the branches that depend on it run on the scalar unit
    // a small record read with ONE LDS instruction (lane k holds word k), its words picked by v_readlane
    struct Rec { uint32_t v; };
    static __device__ __forceinline__ Rec rec_load(const uint32_t *base) { return Rec{ base[lane() & 31u] }; }
    // lanes of a record as a vector: ballot of a predicate / sum over lanes 1, 3, 5, 7
    template <class F>
    static __device__ __forceinline__ unsigned long long rec_mask(const Rec &r, F f) { return __ballot(f(lane(), r.v)); }
    template <class F>
    static __device__ __forceinline__ uint32_t rec_sum_odd4(const Rec &r, F f)
    {
        const uint32_t c = f(lane(), r.v);
        return (uint32_t)__builtin_amdgcn_readlane((int)c, 1) + (uint32_t)__builtin_amdgcn_readlane((int)c, 3) +
               (uint32_t)__builtin_amdgcn_readlane((int)c, 5) + (uint32_t)__builtin_amdgcn_readlane((int)c, 7);
    }
    // 16 words of HBM shared with the worker lanes (agent scope), lane k takes word k
    static __device__ __forceinline__ Rec rec_load_agent(const uint32_t *base)
    {
        return Rec{ __hip_atomic_load(base + (lane() & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) };
    }
    template <class F>
    static __device__ __forceinline__ Rec rec_load_fn(F f) { return Rec{ f(lane() & 7u) }; }
    template <class F>
    static __device__ __forceinline__ Rec rec_load_fn32(F f) { return Rec{ f(lane() & 31u) }; }
    // records as vectors: a or b whole / lanes 1..4 stored to dst[1..4] / lanes 1..4 equal
    static __device__ __forceinline__ Rec rec_sel(bool c, Rec a, Rec b) { return Rec{ c ? a.v : b.v }; }
    static __device__ __forceinline__ void rec_store4(uint32_t *dst, Rec r) { if (lane() - 1u < 4u) dst[lane()] = r.v; }
    static __device__ __forceinline__ uint32_t rec_eq4(Rec a, Rec b) { return ((uint32_t)__ballot(a.v == b.v) & 0x1Eu) == 0x1Eu ? 1u : 0u; }
    // the three values are all loaded before any of them is used (the loads overlap)
    static __device__ __forceinline__ void join3(uint32_t &a, uint32_t &b, uint32_t &c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }
    // word k set (v_writelane) / lanes 0..n-1 stored to dst[0..n-1] / lane k moved to lane k + 2 (lanes 0, 1 undefined)
    static __device__ __forceinline__ Rec rec_set(Rec r, uint32_t k, uint32_t v)
    {
        const uint32_t sv = uni(v);
        asm("v_writelane_b32 %0, %1, %2" : "+v"(r.v) : "s"(sv), "n"(k));      // (no builtin for it in this compiler)
        return r;
    }
    static __device__ __forceinline__ void rec_store_n(uint32_t *dst, Rec r, uint32_t n) { if (lane() < n) dst[lane()] = r.v; }
    // lane k + 3 / k + 8 moved to lane k (within the 16 lanes a record uses)
    static __device__ __forceinline__ Rec rec_shl3(Rec r) { return Rec{ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r.v, 0x103 /* row_shl:3 */, 0xF, 0xF, false) }; }
    static __device__ __forceinline__ Rec rec_shl8(Rec r) { return Rec{ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r.v, 0x108 /* row_shl:8 */, 0xF, 0xF, false) }; }
    static __device__ __forceinline__ Rec rec_shift2(Rec r)
    {
        return Rec{ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r.v, 0x112 /* row_shr:2 */, 0xF, 0xF, false) };
    }
    static __device__ __forceinline__ uint32_t rec_get(Rec r, uint32_t k) { return (uint32_t)__builtin_amdgcn_readlane((int)r.v, (int)k); }
    static __device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    // The master is ONE wave: LDS and same-CU global accesses of a wave complete in
    // program order, so a workgroup-scope fence (the waits) plus a scheduling barrier
    // is all the cross-lane ordering it needs -- no s_barrier.
    static __device__ __forceinline__ void sync()
    {
        // LDS only: DS operations of one wave execute in order, so other lanes' earlier LDS
        // writes are visible to later reads without any wait; only the compiler must not
        // move or cache LDS accesses across this point
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    static __device__ __forceinline__ void sync_global()
    {
        // lane 0's global stores (HT rows, RK table, BT links) before every lane's later loads
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    // agent-scope (sc1, write-through / L1-bypassing) accesses for words shared with worker lanes
    static __device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v)
    {
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ uint32_t ld_agent(const uint32_t *p)
    {
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ void lds_min(uint32_t *p, uint32_t v)
    {
        // result unused: a ds_min_u32 without return, nothing to wait for
        (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    template <class F>
    static __device__ __forceinline__ unsigned long long mask64(F f) { return __ballot(f(lane())); }
    // words shared by the waves of the serial half (same CU, LDS)
    static __device__ __forceinline__ void xw_store(uint32_t *p, uint32_t v)
    {
        asm volatile("" ::: "memory");
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    static __device__ __forceinline__ uint32_t xw_load(const uint32_t *p)
    {
        const uint32_t v = uni(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        asm volatile("" ::: "memory");
        return v;
    }
    static __device__ __forceinline__ void xw_add(uint32_t *p, uint32_t v)
    {
        asm volatile("" ::: "memory");
        if (lane() == 0) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    static __device__ __forceinline__ void xw_pause() { __builtin_amdgcn_s_sleep(0); }      // (hand-offs between the waves are on the critical chain)
    static __device__ __forceinline__ void sleep() { __builtin_amdgcn_s_sleep(4); }
    static __device__ __forceinline__ unsigned long long clock() { return wall_clock64(); }      // 100 MHz
    static __device__ __forceinline__ unsigned long long timeout_ticks() { return 2000000000ull; } // 20 s
    static __device__ __forceinline__ void wait_hook(void *, uint32_t) {}
    static __device__ __forceinline__ unsigned long long tick() { return __builtin_readcyclecounter(); }
    static __device__ __forceinline__ uint32_t rmin(uint32_t v)
    {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v = umin(v, (uint32_t)__shfl_xor((int)v, m, 64));
        return v;
    }
    static __device__ __forceinline__ uint32_t ror(uint32_t v)
    {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v |= (uint32_t)__shfl_xor((int)v, m, 64);
        return v;
    }
    // Up to 8 byte compares at once: group k (8 lanes x 8 bytes) compares
    // in[sp[k] ..] with in[a ..] up to cap[k] bytes; len[k] = common prefix.
    // Up to three byte compares against the bytes at `a` at once (HT2 row, two HT3 rows): job k on lanes 8k..8k+7, eight
    // bytes per lane and round; the first mismatch of a job is the lowest set bit of its byte of one ballot.
    static __device__ __forceinline__ void cmp_multi(const uint8_t *in, const uint32_t sp[8], uint32_t a,
                                                     const uint32_t cap[8], uint32_t valid, uint32_t len[8])
    {
        const uint32_t l = lane(), grp = l >> 3, j = l & 7;
        const uint32_t mysp = grp == 0 ? sp[0] : (grp == 1 ? sp[1] : sp[2]);
        const uint32_t mycap = grp == 0 ? cap[0] : (grp == 1 ? cap[1] : cap[2]);
        uint32_t todo = valid & 7u;                     // jobs without a result yet (wave-uniform)
        len[0] = 0; len[1] = 0; len[2] = 0;
        for (uint32_t off = 0; todo; off += 64) {
            const uint32_t my = off + j * 8;
            uint32_t pos = kNone;
            if (grp < 3 && ((todo >> grp) & 1u) && my < mycap) {
                const unsigned long long d = load64u(in + mysp + my) ^ load64u(in + a + my);
                if (d) {
                    const uint32_t p = my + ((uint32_t)__builtin_ctzll(d) >> 3);
                    if (p < mycap) pos = p;
                }
            }
            const uint32_t bal = (uint32_t)__ballot(pos != kNone);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (!((todo >> k) & 1u)) continue;
                const uint32_t byte = (bal >> (8 * k)) & 0xFFu;
                if (byte) { len[k] = (uint32_t)__builtin_amdgcn_readlane((int)pos, 8 * k + (int)__builtin_ctz(byte)); todo &= ~(1u << k); }
                else if (off + 64 >= cap[k]) { len[k] = cap[k]; todo &= ~(1u << k); }       // no mismatch below the cap
            }
        }
    }
};

// ---------------------------------------------------------------------------
// RK256 hash of every window: rkhash[a] = sum_{j<256} in[a+j] * ADDH^(256-j)
// ---------------------------------------------------------------------------
constexpr uint32_t kRkAddh = 0x2F0FD693u;       // NLZM.cpp:793

__global__ __launch_bounds__(256) void rk_hash_kernel(const uint8_t *__restrict__ in, unsigned long long n,
                                                      unsigned long long pos0, unsigned long long pos1,
                                                      uint32_t *__restrict__ out)
{
    // positions [pos0, pos1); a block covers 1024 consecutive positions, 4 per thread
    __shared__ uint8_t tile[1024 + 256 + 16];
    const unsigned long long blk0 = pos0 + (unsigned long long)blockIdx.x * 1024;
    for (uint32_t i = threadIdx.x; i < 1024 + 256; i += 256) {
        const unsigned long long a = blk0 + i;
        tile[i] = a < n ? in[a] : 0;
    }
    __syncthreads();
    // thread t handles positions blk0 + t + 256*k (k<4): lanes read consecutive bytes
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t o = threadIdx.x + 256u * k;
        const unsigned long long a = blk0 + o;
        if (a >= pos1 || a + 256 > n) continue;
        uint32_t h = 0;
#pragma unroll 16
        for (int j = 0; j < 256; j++) h = (h + tile[o + j]) * kRkAddh;
        out[a] = h;
    }
}

// ---------------------------------------------------------------------------
// pre-filter: which positions can have a 65+ byte match inside the window at all?
//
// parse_table skips BT4 at position p exactly when the table carried from p-1 is
// >= 64 long (NLZM.cpp:1514), which needs a genuine match of >= 65 bytes at p-1
// with distance <= W-1.  "No earlier in-window occurrence of the 65-gram at p-1"
// is a pure function of the input, so for those positions the finder set is known
// without running anything serial; the others are marked `unc` and wait for the
// master.  The filter is conservative: slot collisions only add `unc` marks.
//   T[slot]  = 1 + latest position (from earlier launches) whose 65-gram hashes there
//   M[slot2] = earliest position of THIS launch hashing there
// ---------------------------------------------------------------------------
constexpr uint32_t kPfLen = 65;
constexpr uint32_t kPfMul = 0x9E3779B1u, kPfMul2 = 0x85EBCA77u;

__global__ __launch_bounds__(256) void prefilter_hash_kernel(const uint8_t *__restrict__ in, unsigned long long n,
                                                             uint32_t a0, uint32_t a1, uint32_t wmask, uint32_t t_bits,
                                                             uint32_t m_bits, const uint32_t *__restrict__ T,
                                                             uint32_t *__restrict__ M, uint32_t *__restrict__ hbuf,
                                                             uint8_t *__restrict__ c1)
{
    __shared__ uint8_t tile[1024 + kPfLen + 15];
    const unsigned long long blk0 = (unsigned long long)a0 + (unsigned long long)blockIdx.x * 1024;
    for (uint32_t i = threadIdx.x; i < 1024 + kPfLen; i += 256) {
        const unsigned long long a = blk0 + i;
        tile[i] = a < n ? in[a] : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t o = threadIdx.x + 256u * k;
        const unsigned long long a = blk0 + o;
        if (a >= a1) continue;
        uint32_t h = 0;
        const bool ok = a + kPfLen <= n;
        if (ok) {
#pragma unroll 13
            for (uint32_t j = 0; j < kPfLen; j++) h = (h + tile[o + j]) * 0x2F0FD693u;
        }
        hbuf[a - a0] = h;
        uint8_t f = 0;
        if (ok) {
            const uint32_t t = T[(h * kPfMul) >> (32 - t_bits)];
            f = t != 0 && (uint32_t)a - (t - 1) <= wmask;
            atomicMin(&M[(h * kPfMul2) >> (32 - m_bits)], (uint32_t)a);
        }
        c1[a - a0] = f;
    }
}

// unc[x+1] = C(x); unc[0] = 1 (the position before this launch is not examined)
__global__ __launch_bounds__(256) void prefilter_mark_kernel(unsigned long long n, uint32_t a0, uint32_t a1, uint32_t m_bits,
                                                             const uint32_t *__restrict__ M, const uint32_t *__restrict__ hbuf,
                                                             const uint8_t *__restrict__ c1, uint8_t *__restrict__ unc)
{
    const unsigned long long a = (unsigned long long)a0 + (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (a >= a1) return;
    if (a == a0) unc[0] = 1;
    uint8_t f = c1[a - a0];
    if (a + kPfLen <= n) {
        const uint32_t h = hbuf[a - a0];
        f |= M[(h * kPfMul2) >> (32 - m_bits)] < (uint32_t)a;
    }
    if (a + 1 < a1) unc[a + 1 - a0] = f;
}

__global__ __launch_bounds__(256) void prefilter_insert_kernel(unsigned long long n, uint32_t a0, uint32_t a1, uint32_t t_bits,
                                                               uint32_t m_bits, uint32_t *__restrict__ T,
                                                               uint32_t *__restrict__ M, const uint32_t *__restrict__ hbuf)
{
    const unsigned long long a = (unsigned long long)a0 + (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (a >= a1 || a + kPfLen > n) return;
    const uint32_t h = hbuf[a - a0];
    atomicMax(&T[(h * kPfMul) >> (32 - t_bits)], (uint32_t)a + 1);
    M[(h * kPfMul2) >> (32 - m_bits)] = kNone;
}

// ---------------------------------------------------------------------------
// binning: positions of one chunk grouped by BT4 head (hash of 4 bytes, :1518, :983),
// ascending inside each head.  One workgroup per chunk.
//   cnt: [nchunks][nheads+1] zeroed by the host -> becomes the exclusive offsets
//   cur: [nchunks][nheads]   scratch cursors
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bin_kernel(const uint8_t *__restrict__ in, Geom g, uint32_t c0,
                                                   uint32_t nheads /* = bins */, uint32_t *__restrict__ off_all,
                                                   uint32_t *__restrict__ cur_all, uint32_t *__restrict__ pos_all)
{
    const uint32_t ci = c0 + blockIdx.x;
    uint32_t *off = off_all + (unsigned long long)blockIdx.x * (nheads + 1);
    uint32_t *cur = cur_all + (unsigned long long)blockIdx.x * nheads;
    uint32_t *pos = pos_all + (unsigned long long)blockIdx.x * g.chunk_size;
    const unsigned long long chunk_abs = (unsigned long long)ci * g.chunk_size;
    const unsigned long long remain = g.n - chunk_abs;
    const uint32_t chunk_read = (uint32_t)(remain < g.feed ? remain : g.feed);
    const uint32_t p_end = umin(g.chunk_size, chunk_read);
    const uint32_t n_ok = chunk_read >= 4 ? umin(p_end, chunk_read - 3) : 0;    // positions with >= 4 bytes of lookahead (:1515)
    const uint8_t *base = in + chunk_abs;
    __shared__ uint32_t part[1024];
    __shared__ uint32_t heads_t[1024];

    for (uint32_t p = threadIdx.x; p < n_ok; p += 1024) atomicAdd(&off[(hash4(load32u(base + p)) >> g.bt_shift) % nheads], 1u);
    __syncthreads();
    // exclusive scan over nheads counters: each thread owns a contiguous slice
    const uint32_t per = (nheads + 1023) / 1024;
    const uint32_t lo = umin(nheads, threadIdx.x * per), hi = umin(nheads, lo + per);
    uint32_t sum = 0;
    for (uint32_t h = lo; h < hi; h++) sum += off[h];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t t = 0; t < 1024; t++) { const uint32_t v = part[t]; part[t] = run; run += v; }
        off[nheads] = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t h = lo; h < hi; h++) { const uint32_t v = off[h]; off[h] = run; cur[h] = run; run += v; }
    __syncthreads();
    // stable scatter, 1024 positions at a time
    for (uint32_t t0 = 0; t0 < n_ok; t0 += 1024) {
        const uint32_t p = t0 + threadIdx.x;
        const bool ok = p < n_ok;
        const uint32_t h = ok ? (hash4(load32u(base + p)) >> g.bt_shift) % nheads : kNone;
        heads_t[threadIdx.x] = h;
        __syncthreads();
        uint32_t at = 0, after = 1;
        if (ok) {
            uint32_t before = 0;
            after = 0;
            for (uint32_t t = 0; t < 1024; t++) {
                const uint32_t o = heads_t[t];
                before += (o == h) & (t < threadIdx.x);
                after += (o == h) & (t > threadIdx.x);
            }
            at = cur[h] + before;
            pos[at] = (uint32_t)chunk_abs + p;
        }
        __syncthreads();
        if (ok && !after) cur[h] = at + 1;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// worker lanes: BT4, one serial worker per hash head (trees of different heads are
// disjoint: slot (p & mask)*2 is written by p's own insertion and re-linked only by
// later positions of the same head, NLZM.cpp:979-1021).
// ---------------------------------------------------------------------------
struct LaneIO {
    static __device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v)
    {
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ uint32_t ld_agent(const uint32_t *p)
    {
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // every write-through store of this wave has completed (inline asm: the compiler
    // may not drop or move it, cf. guide "Compiler hazard")
    static __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// The stores of a dry run, noted in LDS (a worker block does not use the master's LDS image): kLogCap (address, value)
// pairs per lane, interleaved by lane so that a wave's k-th entries sit in different banks.
static_assert(sizeof(MasterLds) >= 512 * kLogCap * 8, "store log does not fit the block's LDS");
struct LogInMaster { static __device__ __forceinline__ uint32_t *base() { return (uint32_t *)&g_master_lds; } };
struct LogInV2 { static __device__ __forceinline__ uint32_t *base() { return (uint32_t *)&g_v2_lds; } };
template <class LB>
struct StoreLog {
    uint32_t n;
    bool full;
    // entry k of this lane: target (index into tree[], or 0x80000000 | index into heads[]) and value
    __device__ __forceinline__ static uint32_t *slot(uint32_t k) { return LB::base() + 2 * (k * 512 + threadIdx.x); }
    __device__ __forceinline__ void put(uint32_t t, uint32_t v)
    {
        if (n < kLogCap) { uint32_t *e = slot(n); e[0] = t; e[1] = v; n++; }
        else full = true;
    }
    __device__ __forceinline__ void head(uint32_t *, uint32_t i, uint32_t v) { put(0x80000000u | i, v); }
    __device__ __forceinline__ void link(uint32_t *, uint32_t i, uint32_t v) { put(i, v); }
    __device__ __forceinline__ void replay(uint32_t *heads, uint32_t *tree) const
    {
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t *e = slot(k);
            const uint32_t t = e[0];
            if (t >> 31) heads[t & 0x7FFFFFFFu] = e[1]; else tree[t] = e[1];
        }
    }
};

template <class LB>
__device__ __forceinline__ void worker_role(const Geom &g, const Globals &G, uint32_t c0, uint32_t c1, uint32_t wblocks, uint32_t wblock)
{
    const uint32_t nl = wblocks * blockDim.x;
    const uint32_t gl = wblock * blockDim.x + threadIdx.x;
    // Bin b holds, in ascending order, the positions of every BT4 head h with h % bins == b; lane b walks it,
    // so the position a lane may block on is always its smallest unprocessed one.
    bool active = gl < G.nheads;
    uint32_t c = c0;
    bool loaded = false;
    uint32_t i0 = 0, e0 = 0;
    uint32_t la_end = 0;            // absolute end of the chunk's lookahead
    uint32_t stage = 0, a = 0, max_len = 0;
    unsigned long long n_calls = 0, n_tests = 0, n_cmp = 0, n_dry = 0, n_wait = 0, dry_t = 0, dry_c = 0, n_cyc = 0, n_cyc_tests = 0;
    StoreLog<LB> slog{ 0, false };
    unsigned long long t_wait0 = 0;
    uint32_t idle = 0;
    bool fail = false;

    while (__any(active)) {
        bool waiting = false;
        if (active) {
            if (stage == 0) {
                if (!loaded) {
                    if (c >= c1) active = false;
                    else {
                        const uint32_t *off = G.bin_off + (unsigned long long)(c - c0) * (G.nheads + 1);
                        i0 = off[gl]; e0 = off[gl + 1];
                        const unsigned long long chunk_abs = (unsigned long long)c * g.chunk_size;
                        const unsigned long long remain = g.n - chunk_abs;
                        la_end = (uint32_t)(chunk_abs + (remain < g.feed ? remain : g.feed));
                        loaded = true;
                    }
                }
                if (active) {
                    const uint32_t *pos = G.bin_pos + (unsigned long long)(c - c0) * g.chunk_size;
                    if (i0 >= e0) { c++; loaded = false; }
                    else {
                        a = pos[i0++];
                        max_len = umin(la_end - a, kMatchMax);
                        if (G.unc[a - G.batch_a0]) {
                            // whether this call happens is decided by the master.  If the decision is in already (the
                            // master is ahead of this lane, as it is inside nice regions where 7 of 8 calls are
                            // skipped, :1529), act on it; otherwise its MATCHES do not depend on the decision: report
                            // them now from a dry run, insert once the decision is in
                            const uint32_t f = LaneIO::ld_agent(G.bt_flag + (a - G.batch_a0));
                            if (f == kFlagCall) {
                                worker_bt_call<LaneIO, true>(g, G, a, max_len, true, n_tests, n_cmp);
                                n_calls++;
                            } else if (f != kFlagSkip) {
                                slog.n = 0; slog.full = false; dry_t = 0; dry_c = 0;
                                worker_bt_dry<LaneIO>(g, G, a, max_len, dry_t, dry_c, slog);
                                n_dry++;
                                stage = 1; t_wait0 = 0; idle = 0;
                            }
                        } else {
                            const unsigned long long t0 = __builtin_readcyclecounter(), k0 = n_tests;
                            worker_bt_call<LaneIO, true>(g, G, a, max_len, true, n_tests, n_cmp);
                            n_cyc += __builtin_readcyclecounter() - t0; n_cyc_tests += n_tests - k0;
                            n_calls++;
#ifdef NLZM_LEAD_DIAG
                            {   // where is the master (start of its look-ahead batch) now that this call is done?
                                const uint32_t mp = LaneIO::ld_agent((const uint32_t *)&G.persist->prof[31]);
                                const int lead = (int)(a - mp);
                                const int k = lead < 0 ? 0 : (lead < 64 ? 1 : (lead < 256 ? 2 : (lead < 4096 ? 3 : (lead < 65536 ? 4 : 5))));
                                atomicAdd(&G.wcnt->lead[k], 1ull);
                            }
#endif
                        }
                    }
                }
            } else {
                const uint32_t f = LaneIO::ld_agent(G.bt_flag + (a - G.batch_a0));
                if (f == kFlagCall) {
                    // the call happens: what the dry run noted is exactly what it writes
                    if (!slog.full) { slog.replay(G.bt_heads, G.bt_tree); n_tests += dry_t; n_cmp += dry_c; }
                    else worker_bt_call<LaneIO, true>(g, G, a, max_len, false, n_tests, n_cmp);
                    n_calls++;
                    stage = 0;
                } else if (f == kFlagSkip) {
                    stage = 0;
                } else {
                    waiting = true;
                    n_wait++;
                    if ((++idle & 1023u) == 0) {
                        if (LaneIO::ld_agent(G.abort_word)) active = false;
                        const unsigned long long now = wall_clock64();
                        if (!t_wait0) t_wait0 = now;
                        else if (now - t_wait0 > 3000000000ull) { fail = true; }   // 30 s
                    }
                }
            }
            if (fail) { LaneIO::st_agent(G.abort_word, 2u); active = false; }
        }
        if (!__any(active && !waiting)) __builtin_amdgcn_s_sleep(8);
    }
    // counters: one atomic per wave
    for (int m = 32; m >= 1; m >>= 1) {
        n_calls += __shfl_xor(n_calls, m, 64); n_tests += __shfl_xor(n_tests, m, 64); n_cmp += __shfl_xor(n_cmp, m, 64);
        n_dry += __shfl_xor(n_dry, m, 64); n_wait += __shfl_xor(n_wait, m, 64);
        n_cyc += __shfl_xor(n_cyc, m, 64); n_cyc_tests += __shfl_xor(n_cyc_tests, m, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&G.wcnt->bt_calls, n_calls); atomicAdd(&G.wcnt->bt_tests, n_tests); atomicAdd(&G.wcnt->cmp_bytes, n_cmp);
        atomicAdd(&G.wcnt->dry_runs, n_dry); atomicAdd(&G.wcnt->flag_waits, n_wait);
        atomicAdd(&G.wcnt->call_cycles, n_cyc); atomicAdd(&G.wcnt->call_tests, n_cyc_tests);
    }
}

// the roles of one stream's blocks: block 0 of the stream is its serial half, the others its worker lanes
__device__ __forceinline__ void pipeline_roles(const Geom &g, const Globals &G, uint32_t c0, uint32_t c1, uint32_t local_block,
                                               uint32_t wblocks)
{
    if (local_block == 0) {
        // wave 0: finders (HT2/HT3/RK256 state, nice decision, decisions for the worker lanes)
        // wave 1: the match table (carry / extend / update), published per position
        // wave 2: forward-graph parse, model, symbol emit
        // wave 3: relaxes the listed edges of each node; waves 4 / 5, 6: list the sampled-length edges / the rep probes (even, odd positions)
        if (threadIdx.x < 64) Master<DevWave>::init_shared(G, (uint32_t)((unsigned long long)c0 * g.chunk_size));
        __syncthreads();
        if (threadIdx.x >= 448) return;
        if ((threadIdx.x & 63u) == 0 && threadIdx.x < 256)   // diagnostics: which SIMD each role's wave landed on (HW_ID bits 5:4)
            atomicOr(&G.persist->prof[30], (unsigned long long)(__builtin_amdgcn_s_getreg((15 << 11) | 4) & 0xFFFFu) << (16 * (threadIdx.x >> 6)));
        Master<DevWave> m;
        m.g = g; m.G = G;
        const uint32_t a_first = (uint32_t)((unsigned long long)c0 * g.chunk_size);
        switch (threadIdx.x >> 6) {
        case 0: m.run_finder(c0, c1); break;
        case 1: m.run_table(c0, c1); break;
        case 2: m.run_parser(c0, c1); break;
        case 3: m.run_edge_apply(a_first); break;       // the four waves of the dependency chain get a SIMD each
        case 4: m.run_edge_list(a_first); break;        // (waves 4..6 share them: list makers, mostly waiting)
        case 5: m.run_rep_list(a_first, 0); break;
        default: m.run_rep_list(a_first, 1); break;
        }
    } else {
        worker_role<LogInMaster>(g, G, c0, c1, wblocks, local_block - 1);
    }
}

// ---------------------------------------------------------------------------
// the persistent launch: block 0 = the serial half (seven waves), blocks 1.. = worker lanes.
// 512-thread blocks with 136 KB of LDS: exactly one block per CU, and the grid is kept
// below the CU count, so every block is resident at once (the roles wait on each other).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512) void pipeline_kernel(Geom g, Globals G, uint32_t c0, uint32_t c1)
{
    pipeline_roles(g, G, c0, c1, blockIdx.x, gridDim.x - 1);
}

// Several independent streams in ONE launch (block mode): stream s owns blocks [s*bps, (s+1)*bps).  One launch
// keeps them all resident by construction (separate launches run at most 8 at a time on this device).
struct StreamArgs { Geom g; Globals G; uint32_t c0, c1; };
constexpr uint32_t kMaxStreamsPerLaunch = 12;           // the pack travels in the kernel-argument segment (4 KB)
struct StreamPack { StreamArgs s[kMaxStreamsPerLaunch]; };
static_assert(sizeof(StreamPack) <= 4000, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(512) void pipeline_multi_kernel(StreamPack pack, uint32_t bps)
{
    const uint32_t s = blockIdx.x / bps, local = blockIdx.x % bps;
    pipeline_roles(pack.s[s].g, pack.s[s].G, pack.s[s].c0, pack.s[s].c1, local, bps - 1);
}

// ---------------------------------------------------------------------------
// the persistent launch of the three-stage pipeline (nlzm_v2.h): blocks 0, 1, 2 of a stream are its finder, table and
// parser stage (one wave each: the lanes are the positions of a block / the edges of two nodes), the others its BT4
// worker lanes.  512-thread blocks with > 80 KB of LDS: one block per CU; the grid is kept below the CU count, so every
// block is resident at once (the stages wait on each other through progress words in HBM).
// ---------------------------------------------------------------------------
constexpr uint32_t kV2Roles = 3;
__device__ __forceinline__ void pipeline2_roles(const Geom &g, const Globals &G, const v2::GlobalsV2 &V, uint32_t c0, uint32_t c1,
                                                uint32_t local_block, uint32_t wblocks)
{
    if (local_block < kV2Roles) {
        // finder: one wave; table: kTW waves, a block each; parser: kPW waves on the same block
        if (threadIdx.x >= (local_block == 2 ? v2::kParserThreads : (local_block == 1 ? 64u * v2::kTW : 64u))) return;
        if (local_block == 0) { v2::Finder r; r.g = g; r.G = G; r.V = V; r.run(c0, c1); }
        else if (local_block == 1) { v2::Table r; r.g = g; r.G = G; r.V = V; r.run(c0, c1); }
        else { v2::Parser r; r.g = g; r.G = G; r.V = V; r.run(c0, c1); }
    } else {
        worker_role<LogInV2>(g, G, c0, c1, wblocks, local_block - kV2Roles);
    }
}
__global__ __launch_bounds__(512) void pipeline2_kernel(Geom g, Globals G, v2::GlobalsV2 V, uint32_t c0, uint32_t c1)
{
    pipeline2_roles(g, G, V, c0, c1, blockIdx.x, gridDim.x - kV2Roles);
}
struct Stream2Args { Geom g; Globals G; v2::GlobalsV2 V; uint32_t c0, c1; };
constexpr uint32_t kMaxStreams2PerLaunch = 10;          // the pack travels in the kernel-argument segment (4 KB)
struct Stream2Pack { Stream2Args s[kMaxStreams2PerLaunch]; };
static_assert(sizeof(Stream2Pack) <= 4000, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(512) void pipeline2_multi_kernel(Stream2Pack pack, uint32_t bps)
{
    const uint32_t s = blockIdx.x / bps, local = blockIdx.x % bps;
    pipeline2_roles(pack.s[s].g, pack.s[s].G, pack.s[s].V, pack.s[s].c0, pack.s[s].c1, local, bps - kV2Roles);
}

// ---------------------------------------------------------------------------
// frame coder.  One workgroup (256 threads) per frame.
//   phase 1: lanes 0..3 of wave 0 run the four interleaved states over symbols
//            n-1..0 (state i&3 takes symbol i, NLZM.cpp:599-603) and record, per
//            symbol, the 16-bit word it pushed out (or none).
//   phase 2: block-wide exclusive scan over the "pushed" flags gives every word
//            its place: in memory order the words appear by ascending symbol index
//            after the four flushed states (NLZM.cpp:444-455, 605-612).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rans_frames_kernel(const uint32_t *__restrict__ syms, unsigned long long syms_stride,
                                                          const uint8_t *__restrict__ bits, unsigned long long bits_stride,
                                                          FrameMeta *__restrict__ fmeta, uint32_t *__restrict__ scratch,
                                                          unsigned long long scratch_stride,
                                                          uint8_t *__restrict__ out, unsigned long long out_stride,
                                                          uint32_t out_cap)
{
    const uint32_t f = blockIdx.x;
    const uint32_t n = fmeta[f].nsyms, nb = fmeta[f].nbits_bytes, ops = fmeta[f].num_ops;
    const uint32_t *s = syms + f * syms_stride;
    uint32_t *w = scratch + f * scratch_stride;         // per symbol: pushed word or kNone
    uint8_t *o = out + f * out_stride;
    __shared__ uint32_t st_final[4];
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t carry_s;

    if (threadIdx.x < 4) {
        uint32_t x = 1u << 16;                           // RANS_MID, NLZM.cpp:442,600
        const uint32_t k = threadIdx.x;
        // largest index i < n with (i & 3) == k
        if (n > k) {
            for (long long i = (long long)(((n - 1 - k) & ~3u) + k); i >= 0; i -= 4) {
                const uint32_t v = s[i];
                const uint32_t start = v & 0xFFFFu, freq = v >> 16;
                uint32_t pushed = kNone;
                if (x >= (freq << 18)) {                 // x_max = ((RANS_MID >> 14) << 16) * freq
                    pushed = x & 0xFFFFu;
                    x >>= 16;
                }
                w[i] = pushed;
                x = ((x / freq) << 14) + (x % freq) + start;
            }
        }
        st_final[k] = x;
    }
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();

    // header + bit bytes
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < nb; i += 256) if (12 + i < out_cap) o[12 + i] = bits[f * bits_stride + i];
    uint8_t *r = o + 12 + nb;
    if (threadIdx.x < 16 && 12 + nb + 16 <= out_cap) {
        const uint32_t x = st_final[threadIdx.x >> 2];  // st[0] at the lowest address, little-endian
        r[threadIdx.x] = (uint8_t)(x >> (8 * (threadIdx.x & 3)));
    }
    // words: position = 16 + 2 * (#pushed among symbols < i); bytes hi, lo (NLZM.cpp:449-450 written backwards)
    for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t word = i < n ? w[i] : kNone;
        const bool has = word != kNone;
        const unsigned long long bal = __ballot(has);
        const uint32_t before = (uint32_t)__popcll(bal & ((1ull << lane) - 1));
        if (lane == 0) wave_tot[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = carry_s;
        for (uint32_t k = 0; k < wv; k++) off += wave_tot[k];
        if (has) {
            const uint32_t at = 12 + nb + 16 + 2 * (off + before);
            if (at + 2 <= out_cap) { o[at] = (uint8_t)(word >> 8); o[at + 1] = (uint8_t)word; }
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t nrans = 16 + 2 * carry_s;
        // header: num_ops, nbits (incl. 12 header bytes and the pad), nrans -- big-endian (NLZM.cpp:614-628)
        const uint32_t hdr[3] = { ops, 12 + nb, nrans };
        for (int k = 0; k < 3; k++) {
            o[4 * k + 0] = (uint8_t)(hdr[k] >> 24); o[4 * k + 1] = (uint8_t)(hdr[k] >> 16);
            o[4 * k + 2] = (uint8_t)(hdr[k] >> 8);  o[4 * k + 3] = (uint8_t)hdr[k];
        }
        fmeta[f].out_len = 12 + nb + nrans;
    }
}

// Copy frame f (out_len bytes at frames + f*stride) to dst + dst_off[f].
__global__ __launch_bounds__(256) void gather_frames_kernel(const uint8_t *__restrict__ frames, unsigned long long stride,
                                                            const unsigned long long *__restrict__ dst_off,
                                                            const FrameMeta *__restrict__ fmeta, uint8_t *__restrict__ dst)
{
    const uint32_t f = blockIdx.x;
    const uint32_t len = fmeta[f].out_len;
    const uint8_t *s = frames + f * stride;
    uint8_t *d = dst + dst_off[f];
    for (uint32_t i = threadIdx.x; i < len; i += 256) d[i] = s[i];
}

// ---- launch wrappers (called from nlzm_hip.cpp) ------------------------------
void launch_rk_hash(const uint8_t *in, unsigned long long n, unsigned long long pos0, unsigned long long pos1,
                    uint32_t *out, hipStream_t st)
{
    if (pos1 <= pos0) return;
    const unsigned long long blocks = (pos1 - pos0 + 1023) / 1024;
    hipLaunchKernelGGL(rk_hash_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, in, n, pos0, pos1, out);
}

void launch_pipeline(const Geom &g, const Globals &G, uint32_t c0, uint32_t c1, uint32_t worker_blocks, hipStream_t st)
{
    hipLaunchKernelGGL(pipeline_kernel, dim3(1 + (G.workers ? worker_blocks : 0)), dim3(512), 0, st, g, G, c0, c1);
}

void launch_pipeline2(const Geom &g, const Globals &G, const v2::GlobalsV2 &V, uint32_t c0, uint32_t c1, uint32_t worker_blocks, hipStream_t st)
{
    hipLaunchKernelGGL(pipeline2_kernel, dim3(kV2Roles + worker_blocks), dim3(512), 0, st, g, G, V, c0, c1);
}
unsigned long long stream2_pack_size() { return sizeof(Stream2Pack); }
uint32_t stream2_pack_capacity() { return kMaxStreams2PerLaunch; }
uint32_t pipeline2_role_blocks() { return kV2Roles; }
void fill_stream2_args(void *host_pack, uint32_t i, const Geom &g, const Globals &G, const v2::GlobalsV2 &V, uint32_t c0, uint32_t c1)
{
    Stream2Args &a = ((Stream2Pack *)host_pack)->s[i];
    a.g = g; a.G = G; a.V = V; a.c0 = c0; a.c1 = c1;
}
void launch_pipeline2_multi(const void *host_pack, uint32_t nstreams, uint32_t worker_blocks, hipStream_t st)
{
    hipLaunchKernelGGL(pipeline2_multi_kernel, dim3(nstreams * (kV2Roles + worker_blocks)), dim3(512), 0, st, *(const Stream2Pack *)host_pack,
                       kV2Roles + worker_blocks);
}

// several streams in one launch: fill slot i of a host-side pack, then launch
unsigned long long stream_pack_size() { return sizeof(StreamPack); }
uint32_t stream_pack_capacity() { return kMaxStreamsPerLaunch; }
void fill_stream_args(void *host_pack, uint32_t i, const Geom &g, const Globals &G, uint32_t c0, uint32_t c1)
{
    StreamArgs &a = ((StreamPack *)host_pack)->s[i];
    a.g = g; a.G = G; a.c0 = c0; a.c1 = c1;
}
void launch_pipeline_multi(const void *host_pack, uint32_t nstreams, uint32_t worker_blocks, hipStream_t st)
{
    hipLaunchKernelGGL(pipeline_multi_kernel, dim3(nstreams * (1 + worker_blocks)), dim3(512), 0, st, *(const StreamPack *)host_pack,
                       1 + worker_blocks);
}

void launch_prefilter(const uint8_t *in, unsigned long long n, uint32_t a0, uint32_t a1, uint32_t wmask, uint32_t t_bits,
                      uint32_t m_bits, uint32_t *T, uint32_t *M, uint32_t *hbuf, uint8_t *c1, uint8_t *unc, hipStream_t st)
{
    if (a1 <= a0) return;
    const uint32_t cnt = a1 - a0;
    hipLaunchKernelGGL(prefilter_hash_kernel, dim3((cnt + 1023) / 1024), dim3(256), 0, st, in, n, a0, a1, wmask, t_bits, m_bits,
                       T, M, hbuf, c1);
    hipLaunchKernelGGL(prefilter_mark_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, n, a0, a1, m_bits, M, hbuf, c1, unc);
    hipLaunchKernelGGL(prefilter_insert_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, n, a0, a1, t_bits, m_bits, T, M, hbuf);
}

void launch_bin(const uint8_t *in, const Geom &g, uint32_t c0, uint32_t nchunks, uint32_t nheads, uint32_t *off, uint32_t *cur,
                uint32_t *pos, hipStream_t st)
{
    if (!nchunks) return;
    hipLaunchKernelGGL(bin_kernel, dim3(nchunks), dim3(1024), 0, st, in, g, c0, nheads, off, cur, pos);
}

void launch_rans(const uint32_t *syms, unsigned long long syms_stride, const uint8_t *bits, unsigned long long bits_stride,
                 FrameMeta *fmeta, uint32_t *scratch, unsigned long long scratch_stride, uint8_t *out,
                 unsigned long long out_stride, uint32_t out_cap, uint32_t nframes, hipStream_t st)
{
    if (!nframes) return;
    hipLaunchKernelGGL(rans_frames_kernel, dim3(nframes), dim3(256), 0, st, syms, syms_stride, bits, bits_stride, fmeta,
                       scratch, scratch_stride, out, out_stride, out_cap);
}

void launch_gather(const uint8_t *frames, unsigned long long stride, const unsigned long long *dst_off,
                   const FrameMeta *fmeta, uint8_t *dst, uint32_t nframes, hipStream_t st)
{
    if (!nframes) return;
    hipLaunchKernelGGL(gather_frames_kernel, dim3(nframes), dim3(256), 0, st, frames, stride, dst_off, fmeta, dst);
}

}  // namespace nlzm
