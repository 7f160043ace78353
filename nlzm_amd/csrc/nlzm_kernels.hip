// nlzm_kernels.hip -- gfx950 kernels of the NLZM compress path.
//
//   rk_hash_kernel      RK256 rolling hash of every 256-byte window, one thread per
//                       position (closed form of MatchFinderRK256's roll, NLZM.cpp:798-799,
//                       1071-1083): pure, position-parallel, input tile staged in LDS.
//   prefilter_*_kernel  marks the positions whose BT4 call cannot be decided from the input alone
//   bin_kernel          groups each chunk's positions by BT4 hash head
//   pipeline2_kernel    blocks 0, 1, 2: the finder, table and parser stage (nlzm_v2.h); blocks 3..: BT4 worker lanes
//   pipeline2_multi_kernel  the same for several independent streams in one launch (block mode)
//   rans_frames_kernel  CodeFrame::Flush (NLZM.cpp:590-640): 4 interleaved rANS states
//                       per frame, renormalisation words placed by a prefix scan.
//   gather_frames_kernel concatenates the frames into the output stream.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstddef>
#include <stdint.h>

#include "nlzm_core.h"
#include "nlzm_v2.h"

namespace nlzm {

// LDS image of the three-stage pipeline (nlzm_v2.h): every block of pipeline2_kernel has ONE role, so the roles share
// the bytes.  A file-scope __shared__ object: every access is a ds_* instruction.
#ifndef NLZM_LPW
#define NLZM_LPW 64
#endif
#ifndef NLZM_AHEAD
#define NLZM_AHEAD 3
#endif
constexpr uint32_t kAhead = NLZM_AHEAD;             // calls a worker lane may have made whose fate is not decided yet
constexpr uint32_t kEntryWords = 16;                // LDS words per such call: eight of record-setters, eight of bookkeeping
union V2Lds {
    v2::FLds f; v2::TLds t; v2::PLds p;
    uint32_t wq[512 * kAhead * kEntryWords];        // worker lanes: word w of call e of thread t at (e * 16 + w) * 512 + t
};
__shared__ V2Lds g_v2_lds;
}  // namespace nlzm
namespace xw {
template <class T> XW_FN T *lds() { return reinterpret_cast<T *>(&nlzm::g_v2_lds); }
}
namespace nlzm {

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
//              (t_bitmap: ONE BIT per slot, "some earlier position hashes there" -- a stream that is no longer than its window, as every
//               stream of a block set is, has every earlier position inside the window: the same marks from 1/32 of the memory, round 6)
//   M[slot2] = earliest position of THIS launch hashing there
// ---------------------------------------------------------------------------
constexpr uint32_t kPfLen = 65;
constexpr uint32_t kPfMul = 0x9E3779B1u, kPfMul2 = 0x85EBCA77u;
// the table's slot of a 65-gram: up to 2^32 slots from its 32-bit hash; more (a window of 2^28 positions wants 2^33: one slot in 32 taken,
// and every slot that is taken by another 65-gram of the window is a false mark) from a second hash beside it
__device__ __forceinline__ unsigned long long pf_slot(uint32_t h, uint32_t h2, uint32_t t_bits)
{
    if (t_bits <= 32) return (h * kPfMul) >> (32 - t_bits);
    return ((((unsigned long long)h2 << 32) | h) * 0x9E3779B97F4A7C15ull) >> (64 - t_bits);
}

__global__ __launch_bounds__(256) void prefilter_hash_kernel(const uint8_t *__restrict__ in, unsigned long long n,
                                                             uint32_t a0, uint32_t a1, uint32_t wmask, uint32_t t_bits, uint32_t t_bitmap,
                                                             uint32_t m_bits, const uint32_t *__restrict__ T,
                                                             uint32_t *__restrict__ M, uint32_t *__restrict__ hbuf,
                                                             uint32_t *__restrict__ hbuf2, uint8_t *__restrict__ c1)
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
        uint32_t h = 0, h2 = 0;
        const bool ok = a + kPfLen <= n;
        if (ok) {
#pragma unroll 13
            for (uint32_t j = 0; j < kPfLen; j++) { h = (h + tile[o + j]) * 0x2F0FD693u; h2 = (h2 ^ tile[o + j]) * 0x01000193u + 0x7F4A7C15u; }
        }
        hbuf[a - a0] = h; hbuf2[a - a0] = h2;
        uint8_t f = 0;
        if (ok) {
            const unsigned long long sl = pf_slot(h, h2, t_bits);
            if (t_bitmap) f = (T[sl >> 5] >> (sl & 31u)) & 1u;
            else { const uint32_t t = T[sl]; f = t != 0 && (uint32_t)a - (t - 1) <= wmask; }
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

__global__ __launch_bounds__(256) void prefilter_insert_kernel(unsigned long long n, uint32_t a0, uint32_t a1, uint32_t t_bits, uint32_t t_bitmap,
                                                               uint32_t m_bits, uint32_t *__restrict__ T,
                                                               uint32_t *__restrict__ M, const uint32_t *__restrict__ hbuf,
                                                               const uint32_t *__restrict__ hbuf2)
{
    const unsigned long long a = (unsigned long long)a0 + (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (a >= a1 || a + kPfLen > n) return;
    const uint32_t h = hbuf[a - a0];
    const unsigned long long sl = pf_slot(h, hbuf2[a - a0], t_bits);
    if (t_bitmap) atomicOr(&T[sl >> 5], 1u << (sl & 31u));
    else atomicMax(&T[sl], (uint32_t)a + 1);
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
                                                   uint32_t *__restrict__ cur_all, uint32_t *__restrict__ pos_all,
                                                   const uint8_t *__restrict__ unc, uint32_t batch_a0, uint32_t lds_bins)
{
    const uint32_t ci = c0 + blockIdx.x;
    uint32_t *off = off_all + (unsigned long long)blockIdx.x * (nheads + 1);
    uint32_t *cur = cur_all + (unsigned long long)blockIdx.x * nheads;
    // an entry: the position, and its BT4 head with the pre-filter's mark in bit 31 (what a worker lane needs to start the call:
    // one load instead of a chain of three), the mark of the position before in bit 30 and of the position behind in bit 29 (what the
    // lane assumes about a decision that is not in yet; the array has 16 bytes of slack behind the launch's last position, whose
    // "position behind" is whatever stands there: an assumption, never a result)
    uint32_t *pos = pos_all + (unsigned long long)blockIdx.x * g.chunk_size * 2;
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
    // (the cursors' LDS is sized by the launch: 4 bytes per bin, up to 36,864 bins = 144 KB -- 240 worker CUs x 128 lanes are 30,720 bins, which
    //  the 48 KB of round 4 did not hold: the comparison path below took 6 ms per launch instead of 1.3)
    extern __shared__ uint32_t lcur[];
    if (nheads <= lds_bins) {
        // The cursors live in LDS.  A wave ranks its 64 positions among those of the same bin (one round of ballots per distinct
        // bin in the wave); the waves then take their places in the bins one after the other, which keeps a bin's positions
        // ascending -- 16 short steps per 1024 positions instead of a 1024-step comparison loop per position.
        for (uint32_t h = threadIdx.x; h < nheads; h += 1024) lcur[h] = off[h];
        __syncthreads();
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        for (uint32_t t0 = 0; t0 < n_ok; t0 += 1024) {
            const uint32_t p = t0 + threadIdx.x;
            const bool ok = p < n_ok;
            const uint32_t hfull = ok ? hash4(load32u(base + p)) >> g.bt_shift : 0u;
            const uint32_t h = ok ? hfull % nheads : kNone;
            uint32_t rank = 0, cnt = 0, lead = lane;
            for (unsigned long long todo = __ballot(ok); todo;) {
                const uint32_t l = (uint32_t)__builtin_ctzll(todo);
                const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)l);
                const unsigned long long m = __ballot(ok && h == v);
                if (ok && h == v) { rank = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull)); cnt = (uint32_t)__builtin_popcountll(m); lead = l; }
                todo &= ~m;
            }
            uint32_t first = 0;
            for (uint32_t w = 0; w < 16; w++) {
                if (wave == w && ok && lead == lane) { first = lcur[h]; lcur[h] = first + cnt; }    // (one lane per bin: no two touch the same cursor)
                __syncthreads();
            }
            first = (uint32_t)__shfl((int)first, (int)lead, 64);
            if (ok) {
                const uint32_t at = first + rank;
                const uint32_t a = (uint32_t)chunk_abs + p;
                pos[2 * at] = a;
                pos[2 * at + 1] = hfull | (unc[a - batch_a0] ? 0x80000000u : 0u) | ((a > batch_a0 && unc[a - batch_a0 - 1]) ? 0x40000000u : 0u) | (unc[a - batch_a0 + 1] ? 0x20000000u : 0u);
            }
        }
        return;
    }
    // more bins than the LDS holds cursors for: the cursors in HBM, a position's rank among its 1024 by comparison
    for (uint32_t t0 = 0; t0 < n_ok; t0 += 1024) {
        const uint32_t p = t0 + threadIdx.x;
        const bool ok = p < n_ok;
        const uint32_t hfull = ok ? hash4(load32u(base + p)) >> g.bt_shift : 0u;
        const uint32_t h = ok ? hfull % nheads : kNone;
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
            const uint32_t a = (uint32_t)chunk_abs + p;
            pos[2 * at] = a;
            // (bit 30: the position before is marked as well -- what a worker lane assumes about an undecided position)
            pos[2 * at + 1] = hfull | (unc[a - batch_a0] ? 0x80000000u : 0u) | ((a > batch_a0 && unc[a - batch_a0 - 1]) ? 0x40000000u : 0u) | (unc[a - batch_a0 + 1] ? 0x20000000u : 0u);
        }
        __syncthreads();
        if (ok && !after) cur[h] = at + 1;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// hot bins of a launch: the bins with the most positions, at most `hmax` of them and none below `min_count` -- each gets a
// wave of its own (worker_role_hot).  One workgroup: totals per bin, a histogram of their binary logarithms, the smallest
// power of two as threshold that leaves no more than hmax bins, then the list.
//   hot_of_bin[b]  0, or 1 + the bin's index in the list        hot_list[0] = bins in the list, hot_list[1 + k] = the k-th
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void hot_select_kernel(const uint32_t *__restrict__ off_all, uint32_t nchunks, uint32_t nheads,
                                                          uint32_t hmax, uint32_t min_count, uint32_t *__restrict__ hot_of_bin,
                                                          uint32_t *__restrict__ hot_list, WorkerCounters *__restrict__ wcnt)
{
    __shared__ uint32_t hist[32];
    __shared__ uint32_t thr_s, n_s;
    if (threadIdx.x < 32) hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) n_s = 0;
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nheads; b += 1024) {
        uint32_t tot = 0;
        for (uint32_t c = 0; c < nchunks; c++) {
            const uint32_t *off = off_all + (unsigned long long)c * (nheads + 1);
            tot += off[b + 1] - off[b];
        }
        hot_of_bin[b] = tot;
        if (tot >= min_count && tot) atomicAdd(&hist[31 - __builtin_clz(tot)], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0, k = 32;
        for (int j = 31; j >= 0; j--) {
            if (acc + hist[j] > hmax) break;
            acc += hist[j]; k = (uint32_t)j;
        }
        thr_s = k >= 32 ? 0xFFFFFFFFu : umax(min_count, 1u << k);
    }
    __syncthreads();
    const uint32_t thr = thr_s;
    for (uint32_t b = threadIdx.x; b < nheads; b += 1024) {
        const uint32_t tot = hot_of_bin[b];
        uint32_t v = 0;
        if (tot >= thr) {
            const uint32_t idx = atomicAdd(&n_s, 1u);
            if (idx < hmax) { hot_list[1 + idx] = b; v = idx + 1; }
        }
        hot_of_bin[b] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) { hot_list[0] = umin(n_s, hmax); atomicAdd(&wcnt->hot_bins, (unsigned long long)umin(n_s, hmax)); }
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
    static __device__ __forceinline__ uint32_t atomic_inc(uint32_t *p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    // every write-through store of this wave has completed (inline asm: the compiler
    // may not drop or move it, cf. guide "Compiler hazard")
    static __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    // sixteen bytes in ONE write-through store (global_store_dwordx4 sc1: lands whole, and costs what a plain one does)
    static __device__ __forceinline__ void st_quad(uint32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
    {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4 v = { a, b, c, d };
        // (s_nop: the data registers of a store of more than eight bytes may not be written in the cycle after it -- a hazard the
        //  compiler resolves for its own stores, not for this one)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"((__attribute__((address_space(1))) uint32_t *)p), "v"(v) : "memory");
    }
};

// MatchFinderBT::FindAndUpdate (:978-1022) as the worker lanes run it: the same steps as bt_find_and_update_st
// (nlzm_core.h, which the host-side checks use), written for the lane -- the node's pair is ONE 8-byte load requested
// together with the bytes to compare, what a step changes is chosen by selects, and only a record-setter (about one test in
// four) takes a branch.  A hot head's lane is a chain of such steps, one memory round trip each; the instructions around
// the round trip were as long as the round trip itself.
//
// The record-setters of a descent on their way to the record: the first four in LDS (a register each would be carried, and
// copied, through every step of the descent), the rest in bt_pairs.
struct LaneSink {
    uint32_t *slot;         // LDS: this call's eight words (stride 512: the lanes' words interleave)
    uint32_t *pairs;        // bt_pairs of the position
    uint32_t count = 0, best = 1, best_d = 0;
    uint32_t ext_idx = 0;   // the position's extension block + 1 (pairs beyond its reservation in bt_pairs: nlzm_core.h, bt_pair_ptr)
    __device__ __forceinline__ void put(const BtView &B, uint32_t d, uint32_t l)
    {
        best = l; best_d = d;
        if (count < 4) { slot[(2 * count) * 512] = d; slot[(2 * count + 1) * 512] = l; }
        else {
            uint32_t *q = bt_pair_ptr<LaneIO>(pairs, B.pstride, B.ext, B.ext_cur, B.ext_cap, ext_idx, count);
            // (no block left in the arena: the pair is dropped and the launch goes on to its end -- the host sees the cursor beyond the arena
            //  (Hx::ext_cur) and makes the stream again with every pair reserved, nlzm_hip.cpp; nothing of this launch is used)
            if (q) { LaneIO::st_agent(q, d); LaneIO::st_agent(q + 1, l); }
        }
        count++;
    }
    __device__ __forceinline__ void publish(uint32_t *rec, uint32_t tests) const
    {
        if (count > 4) LaneIO::drain();
        const uint32_t d0 = count > 0 ? slot[0] : 0u, l0 = count > 0 ? slot[512] : 0u, d1 = count > 1 ? slot[2 * 512] : 0u, l1 = count > 1 ? slot[3 * 512] : 0u;
        const uint32_t d2 = count > 2 ? slot[4 * 512] : 0u, l2 = count > 2 ? slot[5 * 512] : 0u, d3 = count > 3 ? slot[6 * 512] : 0u, l3 = count > 3 ? slot[7 * 512] : 0u;
        LaneIO::st_quad(rec + 4, d0, l0, d1, kBtTag);
        LaneIO::st_quad(rec + 8, l1, d2, l2, kBtTag);
        LaneIO::st_quad(rec + 12, d3, l3, ext_idx, kBtTag);
        LaneIO::st_quad(rec, kBtReady | (tests << 9) | count, best_d, count ? best : 0u, 0u);
    }
};
// `keep`: the call is made before it is known whether it happens (see worker_role) -- every store also notes, in `undo`,
// the slot and the value it replaces (a slot is assigned once per call, :1006-1017, so the old value is the child the call
// went on to when it took the slot), at most 1 + 256 + 2 entries.  `dry` (with keep): the stores are NOT made and the notes
// hold the new values instead -- a descent never reads a slot it has assigned, so it is the same descent.
constexpr uint32_t kUndoCap = 260;
__device__ __forceinline__ void bt_descent(const BtView &B, uint32_t a, uint32_t hidx, uint32_t max_len, bool keep, bool dry, uint32_t *undo,
                                           uint32_t &n_undo, LaneSink &sink, uint32_t &tests_out, uint32_t &cmp_out)
{
    const uint8_t *pa = B.in + a;
    uint32_t sp = B.heads[hidx];
    uint32_t nu = 0;
    if (keep) { *(unsigned long long *)undo = (0x80000000u | hidx) | ((unsigned long long)(dry ? a : sp) << 32); nu = 1; }
    if (!dry) B.heads[hidx] = a;                                    // :983-984
    uint32_t pend_l = (a & B.tmask) << 1, pend_r = pend_l + 1, len_l = 0, len_r = 0, tests = 0, cb = 0;
    uint32_t old_l = kNone, old_r = kNone;                          // (the new node's own slots: what they held belongs to no tree)
    while (sp != kNone && a > sp && a - sp <= B.wmask && tests < 256) {    // :989 (256 tests at most, :777, :988)
        tests++;
        const uint32_t pair = (sp & B.tmask) << 1;
        const uint8_t *ps = B.in + sp;
        const uint32_t init = umin(len_l, len_r);                   // :993
        // the node's pair and the first eight bytes of both sides: requested together, one round trip
        const unsigned long long pp = *(const unsigned long long *)(B.tree + pair);     // left, right
        const unsigned long long x0 = load64u(ps + init), y0 = load64u(pa + init), d0 = x0 ^ y0;
        const unsigned long long x1 = load64u(ps + init + 8), y1 = load64u(pa + init + 8), d1 = x1 ^ y1;
        // RingDictionary::MatchLengthSigned (:854-877): common prefix from `init`, at most max_len; sign = which side is smaller.
        // Nearly always decided inside these sixteen bytes (all requested together: a second round is a dependent round trip for all 64 lanes of
        // the wave, which take their tests in lockstep); the rest goes round by round.
        const uint32_t nb0 = d0 ? (uint32_t)__builtin_ctzll(d0) >> 3 : (d1 ? 8u + ((uint32_t)__builtin_ctzll(d1) >> 3) : 16u);
        uint32_t l = init + nb0;
        bool full = l >= max_len;
        const unsigned long long xs = nb0 < 8 ? x0 : x1, ys = nb0 < 8 ? y0 : y1;
        uint32_t sign = (uint32_t)(((xs >> (8 * (nb0 & 7u))) & 0xFF) < ((ys >> (8 * (nb0 & 7u))) & 0xFF));
        if (nb0 == 16 && !full) {
            for (;;) {
                const unsigned long long x = load64u(ps + l), y = load64u(pa + l), d = x ^ y;
                if (d) {
                    const uint32_t nb = (uint32_t)__builtin_ctzll(d) >> 3;
                    l += nb;
                    sign = (uint32_t)(((x >> (8 * nb)) & 0xFF) < ((y >> (8 * nb)) & 0xFF));
                    break;
                }
                l += 8;
                if (l >= max_len) break;
            }
            full = l >= max_len;
        }
        if (full) l = max_len;
        cb += (l - init) + (full ? 0u : 1u);
        const uint32_t pl = (uint32_t)pp, pr = (uint32_t)(pp >> 32), d = a - sp;
        if (l >= match_min(d) && l > sink.best) sink.put(B, d, l);  // :996-998; only record-setters change the table
        if (full) {                                                 // :1000-1004
            if (!dry) { B.tree[pend_l] = pl; B.tree[pend_r] = pr; }
            if (keep) {
                *(unsigned long long *)(undo + 2 * nu) = pend_l | ((unsigned long long)(dry ? pl : old_l) << 32);
                *(unsigned long long *)(undo + 2 * nu + 2) = pend_r | ((unsigned long long)(dry ? pr : old_r) << 32);
                nu += 2;
            }
            tests_out = tests; cmp_out = cb; n_undo = nu;
            return;
        }
        // :1006-1017 as selects
        const bool right = sign != 0;
        const uint32_t slot = right ? pend_l : pend_r;
        if (!dry) B.tree[slot] = sp;
        if (keep) { *(unsigned long long *)(undo + 2 * nu) = slot | ((unsigned long long)(dry ? sp : (right ? old_l : old_r)) << 32); nu++; }
        pend_l = right ? pair + 1 : pend_l; pend_r = right ? pend_r : pair;
        old_l = right ? pr : old_l; old_r = right ? old_r : pl;
        len_r = right ? l : len_r; len_l = right ? len_l : l;
        sp = right ? pr : pl;
    }
    if (!dry) { B.tree[pend_r] = kNone; B.tree[pend_l] = kNone; }   // :1020-1021
    if (keep) {
        *(unsigned long long *)(undo + 2 * nu) = pend_r | ((unsigned long long)(dry ? kNone : old_r) << 32);
        *(unsigned long long *)(undo + 2 * nu + 2) = pend_l | ((unsigned long long)(dry ? kNone : old_l) << 32);
        nu += 2;
    }
    tests_out = tests; cmp_out = cb; n_undo = nu;
}

// ---------------------------------------------------------------------------
// A hot bin -- the bin of a frequent 4-byte head of the text -- is a serial chain of calls that one lane cannot keep up with
// at depth: such a bin gets a WAVE.  Its lanes run consecutive calls of the bin at the same time, each
// trailing the calls before it down the tree, with exactly the results of the serial order:
//  * During a call every slot of the tree is assigned at most once (:1006-1017: the slot a call holds as pend_l / pend_r is
//    written when the call turns that way again, or at its end), and a call reads a slot only on its way down.  A call
//    marks the slot it takes as pending (kPending in the slot itself, before the store that makes the slot's node
//    reachable); a call that needs a pending slot repeats the step.  A later call can reach a node of an earlier call's
//    path only through a slot that call has assigned, i.e. after it has left the node, and can enter only the side the
//    earlier call did not take: it never overtakes, the earliest call in flight never meets a mark, the wave always moves.
//    All lanes are lanes of ONE wave: its memory operations are issued in program order (loads at the top of a step,
//    stores at its end), which orders a mark before the store that exposes it and both before the next step's loads.
//  * Calls start one per step, in the bin's order (a call starts by reading its head and storing itself there).
//  * An undecided position does not stop the wave: as in the lanes' role its decision is assumed (skip if the position
//    before is marked too, call otherwise), the results of the calls behind it are held back until the decision, and every
//    store notes (slot, value it replaced) in the lane's list.  A decision as assumed releases what was held.  Otherwise the
//    wave stops starting calls, lets the calls in flight finish, takes the stores of every call behind the position
//    back, latest first, takes back or makes the position's own, and goes on behind it.
// ---------------------------------------------------------------------------
#ifndef NLZM_RISKY_AHEAD
#define NLZM_RISKY_AHEAD 0
#endif
constexpr uint32_t kPending = 0xFFFFFFFEu;      // (no position: stream_begin refuses inputs of 0xFFFF0000 bytes and more)
// Round 6: the bin's entries pass through a WINDOW of 64 (lane j of the wave holds entry wb + j of the chunk's list, loaded with one coalesced
// load and one gather of the marked entries' decision words), and only CALLS take one of the wave's lanes:
//   * an entry whose decision is "skip" already is dropped where it stands -- any number of them per step (they cost a step each before: half
//     the steps of the hottest bin on source code);
//   * an undecided entry that is assumed to be skipped (marked like both its neighbours) is PARKED in the window: nothing is made of it, it is
//     passed at once, and it stands there as an undecided position -- what is behind it is held -- until its decision word says "skip" (dropped)
//     or "call" (a wrong assumption: everything behind it is taken back and the bin goes on AT it);
//   * every other entry is a call: unmarked, decided "call", or undecided and assumed to be called (its stores noted).  One call starts per
//     step (a call starts by reading its head and storing itself there), on the first free lane; the entries between it and the next call are
//     passed in the same step.
// The window is loaded again behind its last entry once nothing undecided is parked in it; a wrong assumption loads it again where the bin
// goes on.  Order: an entry's number in the chunk's list + 1 (`seq`), for calls and parked entries alike.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)xw::scan_max(v), 63); }
__device__ __forceinline__ void worker_role_hot(const Geom &g, const Globals &G, uint32_t c0, uint32_t c1, uint32_t bin, uint32_t hot_index)
{
    unsigned long long p0 = (unsigned long long)G.in, p1 = (unsigned long long)G.bt_heads, p2 = (unsigned long long)G.bt_tree,
                       p3 = (unsigned long long)G.bt_ready, p4 = (unsigned long long)G.bt_pairs, p5 = (unsigned long long)G.bt_flag,
                       p6 = (unsigned long long)G.hot_undo;
    uint32_t q0 = G.batch_a0, q2 = g.wmask, q3 = g.bt_tmask;
    asm volatile("" : "+s"(p0), "+s"(p1), "+s"(p2), "+s"(p3), "+s"(p4), "+s"(p5), "+s"(p6));
    asm volatile("" : "+s"(q0), "+s"(q2), "+s"(q3));
#define NLZM_AS_GLOBAL(T, x) ((T *)(__attribute__((address_space(1))) T *)(x))
    const uint8_t *in = NLZM_AS_GLOBAL(const uint8_t, p0);
    uint32_t *heads = NLZM_AS_GLOBAL(uint32_t, p1), *tree = NLZM_AS_GLOBAL(uint32_t, p2), *ready = NLZM_AS_GLOBAL(uint32_t, p3);
    uint32_t *pairs = NLZM_AS_GLOBAL(uint32_t, p4);
    const uint32_t *flags = NLZM_AS_GLOBAL(const uint32_t, p5);
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long *const undo = NLZM_AS_GLOBAL(unsigned long long, p6) + ((unsigned long long)hot_index * 64 + lane) * kUndoCap;
#undef NLZM_AS_GLOBAL
    const uint32_t batch_a0 = q0, wmask = q2, tmask = q3;
    BtView Bx{ in, heads, tree, ready, pairs, batch_a0, 0u, wmask, tmask };         // (for LaneSink::put: where a position's later pairs go)
    Bx.pstride = G.bt_pstride; Bx.ext = G.bt_ext; Bx.ext_cur = G.bt_ext_cur; Bx.ext_cap = G.bt_ext_cap; Bx.fail_word = G.abort_word;
    enum : uint32_t { kIdle = 0, kStart = 1, kRun = 2, kHeld = 3 };
    enum : uint32_t { kwNone = 0, kwDone = 1, kwPark = 2, kwCall = 3 };             // a window entry: none / dropped or started / parked (undecided once passed) / a call to be started
    // the bin's entries of the chunk, in order (wave-uniform): the window starts at entry wb, entries below `cur` are passed
    uint32_t c = c0, wb = 0, cur = 0, e0 = 0, la_end = 0;
    const uint32_t *pos = G.bin_pos;
    bool loaded = false, more = true, wvalid = false, rec = false;
    // the lane's window entry
    unsigned long long w_pe = 0;
    uint32_t w_kind = kwNone;
    bool w_risky = false, w_wrong = false;
    // ... and its entry of the window BEFORE (entries ob + j, all passed): only what is parked there and not decided yet still matters.  With it the
    // bin is looked at up to 128 entries ahead of its oldest undecided entry; with one window the bin stood with nothing in hand whenever the finder
    // stage had decided a window's last parked entry -- and the finder met it there.
    unsigned long long o_pe = 0;
    bool o_park = false, o_risky = false, o_wrong = false, ovalid = false;
    uint32_t ob = 0;
    // the lane's call
    uint32_t st = kIdle, a = 0, hidx = 0, max_len = 0, sp = kNone, pend_l = 0, pend_r = 0, len_l = 0, len_r = 0, tests = 0, cb = 0;
    uint32_t seq = 0, nu = 0;
    uint32_t cmp_off = 0;           // bytes of the test in hand that are known to agree beyond `init` (a compare longer than a step's sixteen bytes goes on in the next step)
    bool marked = false, und = false, published = false, wrong = false;
    bool last_skip = false;         // (wave-uniform) the bin's latest decision was "skip"
    LaneSink sink{ (uint32_t *)&g_v2_lds + threadIdx.x, nullptr };
    // (counters of a launch: 32 bits each inside the loop -- a launch is a few hundred thousand steps -- but the two that add up whole calls)
    unsigned long long n_tests = 0, n_cmp = 0;
    uint32_t n_calls = 0, n_open = 0, n_back = 0, n_redo = 0;
    uint32_t n_steps = 0, n_blk_risky = 0;
    unsigned long long t_wait0 = 0;
    uint32_t prog_seen = 0, steps = 0;
    // (accounting by the bin's size: WorkerCounters::hot_class)
    uint32_t k_work = 0, k_tests = 0, k_rep = 0, k_rec = 0, k_full = 0, k_blk = 0, k_noent = 0, k_skip = 0, k_und = 0;
    const unsigned long long k_t0 = __builtin_readcyclecounter();
    uint32_t k_total = 0;
    for (uint32_t cc = c0; cc < c1; cc++) { const uint32_t *off = G.bin_off + (unsigned long long)(cc - c0) * (G.nheads + 1); k_total += off[bin + 1] - off[bin]; }
#ifdef NLZM_PROFILE
    unsigned long long t_sec[7] = {}, t_last = __builtin_readcyclecounter();
#define NLZM_HOT_STAMP(k) do { const unsigned long long t_now = __builtin_readcyclecounter(); t_sec[k] += t_now - t_last; t_last = t_now; } while (0)
#else
#define NLZM_HOT_STAMP(k) do { } while (0)
#endif

    for (;;) {
        // ---- the window before is done with when nothing parked is left in it; a window that is passed but still holds parked entries takes its place
        if (ovalid && !__any(o_park)) ovalid = false;
        if (wvalid && !ovalid && !rec && cur >= wb + umin(64u, e0 - wb) && cur < e0) {
            o_pe = w_pe; o_park = w_kind == kwPark; o_risky = w_risky; o_wrong = w_wrong; ob = wb; ovalid = true;
            wvalid = false; w_kind = kwNone; w_wrong = false; w_risky = false;
        }
        // ---- the window: loaded where the bin goes on; the chunk is left when nothing of it is in flight
        if (!wvalid && more && !rec) {
            for (;;) {
                if (!loaded) {
                    if (c >= c1) { more = false; break; }
                    const uint32_t *off = G.bin_off + (unsigned long long)(c - c0) * (G.nheads + 1);
                    // (every lane reads the same two words: said so, or the whole loop's control flow is compiled as if the lanes could part)
                    cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)off[bin]); e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)off[bin + 1]);
                    const unsigned long long chunk_abs = (unsigned long long)c * g.chunk_size;
                    const unsigned long long remain = g.n - chunk_abs;
                    la_end = (uint32_t)(chunk_abs + (remain < g.feed ? remain : g.feed));
                    pos = G.bin_pos + (unsigned long long)(c - c0) * g.chunk_size * 2;
                    loaded = true;
                }
                if (cur < e0) break;
                if (__any(st != kIdle) || ovalid) break;            // (a wrong assumption would bring the bin back into this chunk)
                c++; loaded = false;
            }
            if (more && loaded && cur < e0) {
                wb = cur;
                const uint32_t idx = wb + lane;
                w_pe = 0; w_kind = kwNone; w_risky = false; w_wrong = false;
                if (idx < e0) {
                    w_pe = *(const unsigned long long *)(pos + 2 * idx);
                    const bool mk = (w_pe >> 63) != 0;
                    uint32_t f = 0;
                    if (mk) f = LaneIO::ld_agent(flags + ((uint32_t)w_pe - batch_a0));
                    // (a decision that comes in after this look finds the entry as a parked one or as a call that polls its word)
                    w_kind = !mk ? kwCall : (f == kFlagSkip ? kwDone : (f == kFlagCall ? kwCall : ((w_pe >> 61) == 7u ? kwPark : kwCall)));
                    k_skip += mk && f == kFlagSkip;
                }
                wvalid = true;
            }
        }
        NLZM_HOT_STAMP(6);
        const uint32_t wend = wvalid ? umin(64u, e0 - wb) : 0u;     // entries in the window
        // ---- the oldest undecided position (a call's or a parked entry's); a wrong assumption
        const bool w_und = wvalid && w_kind == kwPark && wb + lane < cur;
        const bool o_und = ovalid && o_park;
        const uint32_t w_seq = wb + lane + 1, o_seq = ob + lane + 1;
        // (a window's entries stand in lane order: its oldest is its first; the calls stand on whatever lane was free.  One instruction of a
        //  lone wave is four cycles whatever it does, and a reduction over the wave is fifteen of them: made only where something can be found)
        const unsigned long long w_um = __ballot(w_und), o_um = __ballot(o_und);
        uint32_t oseq = 0xFFFFFFFFu, rseq = 0xFFFFFFFFu;
        if (__any(und)) oseq = ~wave_max_u32(und ? ~seq : 0u);
        if (w_um) oseq = umin(oseq, wb + (uint32_t)__builtin_ctzll(w_um) + 1u);
        if (o_um) oseq = umin(oseq, ob + (uint32_t)__builtin_ctzll(o_um) + 1u);
        if (__any(wrong || w_wrong || (ovalid && o_wrong))) {
            rseq = ~wave_max_u32(umax(wrong ? ~seq : 0u, umax(w_wrong ? ~w_seq : 0u, (ovalid && o_wrong) ? ~o_seq : 0u)));
            rec = true;
        }
        if (rec && !__any(st == kStart || st == kRun)) {
            // every call in flight has ended: the stores of the calls behind the position are taken back, latest first; then the position's
            // own, if it was assumed to be called; the bin goes on behind it -- or AT it, if it was parked and is called
            for (;;) {
                const uint32_t smax = wave_max_u32(st == kHeld && seq > rseq ? seq : 0u);
                if (!smax) break;
                if (st == kHeld && seq == smax) {
                    for (uint32_t j = nu; j-- > 0;) {
                        const unsigned long long w = undo[j];
                        const uint32_t t = (uint32_t)w, v = (uint32_t)(w >> 32);
                        if (t >> 31) heads[t & 0x7FFFFFFFu] = v; else tree[t] = v;
                    }
                    st = kIdle; und = false; wrong = false; n_redo++;
                }
            }
            if (st == kHeld && seq == rseq) {
                for (uint32_t j = nu; j-- > 0;) {                   // (assumed to be called, and skipped: the values its stores replaced)
                    const unsigned long long w = undo[j];
                    const uint32_t t = (uint32_t)w, v = (uint32_t)(w >> 32);
                    if (t >> 31) heads[t & 0x7FFFFFFFu] = v; else tree[t] = v;
                }
                st = kIdle; und = false; wrong = false; n_back++;
            }
            const bool parked = __any((w_wrong && w_seq == rseq) || (ovalid && o_wrong && o_seq == rseq));    // (assumed to be skipped, and called: made now)
            if (parked && lane == 0) n_back++;
            cur = parked ? rseq - 1 : rseq;
            // Where the bin goes on, the entries are looked at again (their calls are taken back, decisions may have come in).  What is parked IN
            // FRONT of that place stays as it is: it is older than the wrong assumption and may be undecided still (decisions are stored in position
            // order, but two stores of one instruction need not become visible in that order).
            if (ovalid && cur < ob + 64) {
                if (cur >= ob) {                                    // inside the window before: it is the window again
                    w_pe = o_pe; w_kind = o_park ? kwPark : kwDone; w_risky = o_risky; w_wrong = o_wrong; wb = ob; wvalid = true;
                } else wvalid = false;                              // (in front of both: nothing undecided can be parked in front of it)
                ovalid = false; o_park = false; o_wrong = false;
            } else if (wvalid && cur < wb) wvalid = false;
            if (wvalid) {
                const uint32_t idx = wb + lane;
                if (idx >= cur) {
                    w_kind = kwNone; w_risky = false; w_wrong = false;
                    if (idx < e0) {
                        w_pe = *(const unsigned long long *)(pos + 2 * idx);
                        const bool mk = (w_pe >> 63) != 0;
                        uint32_t f = 0;
                        if (mk) f = LaneIO::ld_agent(flags + ((uint32_t)w_pe - batch_a0));
                        w_kind = !mk ? kwCall : (f == kFlagSkip ? kwDone : (f == kFlagCall ? kwCall : ((w_pe >> 61) == 7u ? kwPark : kwCall)));
                    }
                }
            } else { w_wrong = false; w_kind = kwNone; }
            rec = false;
            continue;
        }
        NLZM_HOT_STAMP(0);
        // ---- entries are passed: dropped ones and parked ones as many as there are, and ONE call -- on the first free lane, not while a wrong
        // assumption is being undone, and not behind a parked entry that is assumed to be skipped although the bin's last decision was "call"
        // (wrong one time in eight on prose, and a wrong assumption costs every call behind it)
        uint32_t start_lane = 64;
        unsigned long long pe = 0;
        uint32_t pe_seq = 0;
        bool made_test = false, rep_step = false, started_now = false;
        const unsigned long long free_m = __ballot(st == kIdle);
        const bool risky_open = __any((w_und && w_risky) || (o_und && o_risky));
        bool call_blocked = false;
        n_steps++;
        if (wvalid && !rec && cur < wb + wend) {
            const uint32_t cl = cur - wb;
            const unsigned long long range = (~0ull << cl) & (wend < 64 ? (1ull << wend) - 1ull : ~0ull);
            const unsigned long long mC = __ballot(w_kind == kwCall) & range, mP = __ballot(w_kind == kwPark) & range;
            const uint32_t c1i = mC ? (uint32_t)__builtin_ctzll(mC) : 64u, p1 = mP ? (uint32_t)__builtin_ctzll(mP) : 64u;
            // (a parked entry is RISKY when the bin's last decision was not "skip": nothing is passed behind it until it is decided -- the entries
            //  behind it are then no longer risky, the decision being "skip" -- and it is passed with nothing but what stands in front of it)
            bool call_ok = c1i < 64 && free_m && !risky_open && !(!last_skip && p1 < c1i);
            call_blocked = c1i < 64 && free_m && !call_ok;
            uint32_t end = cl;
            if (!risky_open) {
                end = c1i;
                if (call_ok) {
                    const unsigned long long restC = mC & ~(1ull << c1i);
                    end = restC ? (uint32_t)__builtin_ctzll(restC) : 64u;
                }
                if (!last_skip && p1 < end) end = p1 + 1;
            }
            if (call_ok) {
                start_lane = (uint32_t)__builtin_ctzll(free_m);
                pe = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w_pe >> 32), (int)c1i) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w_pe, (int)c1i);
                pe_seq = wb + c1i + 1;
            }
            end = umin(end, wend);
            if (w_kind == kwPark && lane >= cl && lane < end) w_risky = !last_skip;
            if (call_ok && lane == c1i) w_kind = kwDone;            // (started: its lane has it)
            cur = wb + end;
        }
        if (call_blocked) n_blk_risky++;
        if (lane == start_lane) {
            a = (uint32_t)pe; hidx = (uint32_t)(pe >> 32) & 0x1FFFFFFFu;
            max_len = umin(la_end - a, kMatchMax);
            marked = (pe >> 63) != 0;
            seq = pe_seq;
            st = kStart;
        }
        // (the window is done with when every entry is passed and nothing undecided is parked in it)
        if (wvalid && cur >= wb + wend && !__any(w_kind == kwPark)) wvalid = false;
        NLZM_HOT_STAMP(1);
        // ---- loads of this step
        uint32_t v_word = 0, v_flag = 0, w_flag = 0;
        unsigned long long pp = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        bool fin_now = false;
        uint32_t fin_l = kNone, fin_r = kNone;
        if (st == kRun && !(sp != kNone && a > sp && a - sp <= wmask && tests < 256)) fin_now = true;   // :989, :1020-1021
        const uint32_t pair = (sp & tmask) << 1, init = umin(len_l, len_r);
        if ((st == kStart && marked) || und) v_flag = LaneIO::ld_agent(flags + (a - batch_a0));
        // (decisions are made in position order: only the OLDEST undecided entries' words can have changed -- eight of them are looked at, the window
        //  before first.  Every parked entry looking at its word in every step was a gather of up to 128 uncached loads per step: steps of 5,600
        //  cycles where 4,000 were the rule.)
        uint32_t o_flag = 0;
        {
            const unsigned long long om = __ballot(o_und && !o_wrong), wm = __ballot(w_und && !w_wrong);
            const uint32_t o_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u));
            const uint32_t w_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wm, 0u)) + (uint32_t)__builtin_popcountll(om);
            if (o_und && !o_wrong && o_rank < 8u) o_flag = LaneIO::ld_agent(flags + ((uint32_t)o_pe - batch_a0));
            if (w_und && !w_wrong && w_rank < 8u) w_flag = LaneIO::ld_agent(flags + ((uint32_t)w_pe - batch_a0));
        }
        if (st == kStart) v_word = heads[hidx];
        else if (st == kRun && !fin_now) {
            pp = *(const unsigned long long *)(tree + pair);
            // (sixteen bytes of both sides from where the test stands: ONE round trip per step whatever the lanes' compares do -- a compare that ran on
            //  inside the step, round by round, was a second and third dependent round trip for the whole wave, and the runs of spaces of source code,
            //  the hottest bin there is, agree over more than eight bytes in most of their tests: steps of 6,000 cycles)
            const uint8_t *ps = in + sp + init + cmp_off, *pa = in + a + init + cmp_off;
            x0 = load64u(ps); y0 = load64u(pa); x1 = load64u(ps + 8); y1 = load64u(pa + 8);
        }
        // ---- what they say
        {
            const bool decided = und && !wrong && (v_flag == kFlagCall || v_flag == kFlagSkip);
            if (decided) {
                if (v_flag == kFlagCall) und = false;               // as assumed
                else wrong = true;                                  // (und stays: nothing behind it is released)
            }
            const bool w_decided = w_und && !w_wrong && (w_flag == kFlagCall || w_flag == kFlagSkip);
            if (w_decided) {
                if (w_flag == kFlagSkip) w_kind = kwDone;           // as assumed: dropped
                else w_wrong = true;                                // (stays parked: nothing behind it is released)
            }
            const bool o_decided = o_und && !o_wrong && (o_flag == kFlagCall || o_flag == kFlagSkip);
            if (o_decided) {
                if (o_flag == kFlagSkip) o_park = false;
                else o_wrong = true;
            }
            if (__any(decided || w_decided || o_decided))
                last_skip = __any((decided && v_flag == kFlagSkip) || (w_decided && w_flag == kFlagSkip) || (o_decided && o_flag == kFlagSkip));
        }
#ifdef NLZM_PROFILE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (profile build: the round trip ends here, not at the first use of what was loaded)
#endif
        NLZM_HOT_STAMP(2);
        if (st == kStart) {
            bool go = true;
            if (marked) {
                if (v_flag == kFlagSkip) { st = kIdle; go = false; }                                    // decided by now: it does not happen
                else if (v_flag != kFlagCall) { und = true; n_open++; }          // its fate is open: assumed to be called, every store noted
            }
            if (go) {
                started_now = true;
                sp = v_word;                                        // :983
                pend_l = (a & tmask) << 1; pend_r = pend_l + 1; len_l = 0; len_r = 0; tests = 0; cb = 0; cmp_off = 0;
                sink.count = 0; sink.best = 1; sink.best_d = 0; sink.ext_idx = 0; sink.pairs = pairs + (unsigned long long)(a - batch_a0) * (2 * Bx.pstride);
                published = false;
                undo[0] = (0x80000000u | hidx) | ((unsigned long long)sp << 32);
                nu = 1;
                *(unsigned long long *)(tree + pend_l) = ((unsigned long long)kPending << 32) | kPending;   // the new node's two slots: taken
                heads[hidx] = a;                                    // :984
                st = kRun;
            }
        } else if (st == kRun && !fin_now) {
            // RingDictionary::MatchLengthSigned (:854-877) over this step's sixteen bytes
            const unsigned long long d0 = x0 ^ y0, d1 = x1 ^ y1;
            const uint32_t from = init + cmp_off;
            uint32_t nb = d0 ? (uint32_t)__builtin_ctzll(d0) >> 3 : (d1 ? 8u + ((uint32_t)__builtin_ctzll(d1) >> 3) : 16u);
            uint32_t l = from + nb;
            bool full = l >= max_len;
            const unsigned long long xs = nb < 8 ? x0 : x1, ys = nb < 8 ? y0 : y1;
            const uint32_t sign = (uint32_t)(((xs >> (8 * (nb & 7u))) & 0xFF) < ((ys >> (8 * (nb & 7u))) & 0xFF));
            const uint32_t pl = (uint32_t)pp, pr = (uint32_t)(pp >> 32);
            const bool more_cmp = nb == 16 && !full;                // all sixteen agree and the cap is not reached: the compare goes on in the next step
            if (more_cmp) cmp_off += 16;
            if (full) l = max_len;
            const bool right = sign != 0;
            // the slot(s) this step reads on: still held by an earlier call -> the step is repeated
            const bool held = more_cmp || (full ? (pl == kPending || pr == kPending) : ((right ? pr : pl) == kPending));
            rep_step = held;
            if (!held) {
                made_test = true;
                cmp_off = 0;
                tests++;
                cb += (l - init) + (full ? 0u : 1u);
                const uint32_t d = a - sp;
                if (l >= match_min(d) && l > sink.best) sink.put(Bx, d, l);  // :996-998
                if (full) { fin_now = true; fin_l = pl; fin_r = pr; }       // :1000-1004
                else {
                    // :1006-1017.  The slot taken is marked before the store that makes its node reachable for later calls.
                    const uint32_t slot = right ? pend_l : pend_r;
                    // (order matters to the later calls of this wave: the mark, then the link that makes the node reachable, both
                    //  before the next step's loads.  The hardware issues one wave's memory operations in program order and
                    //  returns loads behind earlier stores to the same address; the barriers keep the compiler from re-ordering
                    //  or sinking the two plain stores)
                    tree[right ? pair + 1 : pair] = kPending;
                    asm volatile("" ::: "memory");
                    tree[slot] = sp;
                    asm volatile("" ::: "memory");
                    // noted: the slot taken now and what it held
                    undo[nu] = (right ? pair + 1 : pair) | ((unsigned long long)(right ? pr : pl) << 32);
                    nu++;
                    pend_l = right ? pair + 1 : pend_l; pend_r = right ? pend_r : pair;
                    len_r = right ? l : len_r; len_l = right ? len_l : l;
                    sp = right ? pr : pl;
                }
            }
        }
        NLZM_HOT_STAMP(3);
        if (fin_now) {
            tree[pend_l] = fin_l; tree[pend_r] = fin_r;
            st = kHeld;
        }
        // ---- a call that has ended: its result goes out when no undecided position stands before it
        if (st == kHeld && !wrong) {
            if (!published && seq <= oseq) { sink.publish(ready + (unsigned long long)(a - batch_a0) * kBtRec, tests); published = true; }
            if (published && !und && seq <= oseq) {
                n_calls++; n_tests += tests; n_cmp += cb;
                st = kIdle;
            }
        }
        NLZM_HOT_STAMP(4);
#ifdef NLZM_PROFILE
        {   // (accounting)
            const uint32_t nt = (uint32_t)__builtin_popcountll(__ballot(made_test)), nr = (uint32_t)__builtin_popcountll(__ballot(rep_step));
            k_tests += nt; k_rep += nr;
            if (nt || __any(started_now)) k_work++;
            else {
                if (rec) k_rec++;
                else if (!free_m) k_full++;
                else if (call_blocked) k_blk++;
                else k_noent++;
                if (__any(und || w_und || o_und)) k_und++;
            }
        }
#else
        (void)made_test; (void)rep_step; (void)started_now;
#endif
        // ---- watchdogs, every 4,096 steps (a launch that failed elsewhere; a position whose decision does not come while the finder stage stands)
        if ((++steps & 4095u) == 0) {
            if (__any(LaneIO::ld_agent(G.abort_word) != 0)) break;
            if (__any(und || w_und || o_und)) {
                const unsigned long long now = wall_clock64();
                const uint32_t prog = G.progress ? (uint32_t)__builtin_amdgcn_readfirstlane((int)LaneIO::ld_agent(G.progress)) : 0u;
                if (!t_wait0 || prog != prog_seen) { t_wait0 = now; prog_seen = prog; }
                else if (now - t_wait0 > 3000000000ull) { if (lane == 0) LaneIO::st_agent(G.abort_word, 2u); break; }
            } else t_wait0 = 0;
        }
        NLZM_HOT_STAMP(5);
        if (!more && !wvalid && !ovalid && !rec && !__any(st != kIdle)) break;
    }
    if (st != kIdle) {
        atomicAdd(&G.wcnt->stuck_lanes, 1ull);
        atomicMax(&G.wcnt->stuck_pos_inv, (unsigned long long)(uint32_t)~a);
    }
    unsigned long long t_calls = n_calls, t_open = n_open, t_back = n_back, t_redo = n_redo, t_skip = k_skip;
    for (int m = 32; m >= 1; m >>= 1) {
        t_calls += __shfl_xor(t_calls, m, 64); n_tests += __shfl_xor(n_tests, m, 64); n_cmp += __shfl_xor(n_cmp, m, 64);
        t_open += __shfl_xor(t_open, m, 64); t_back += __shfl_xor(t_back, m, 64); t_redo += __shfl_xor(t_redo, m, 64);
        t_skip += __shfl_xor(t_skip, m, 64);
    }
    if (lane == 0) {
        atomicAdd(&G.wcnt->bt_calls, t_calls); atomicAdd(&G.wcnt->bt_tests, n_tests); atomicAdd(&G.wcnt->cmp_bytes, n_cmp);
        atomicAdd(&G.wcnt->dry_runs, t_open); atomicAdd(&G.wcnt->spec_calls, t_back); atomicAdd(&G.wcnt->spec_good, t_redo);
        atomicAdd(&G.wcnt->hot_calls, t_calls);
        atomicAdd(&G.wcnt->hot_steps, (unsigned long long)n_steps); atomicAdd(&G.wcnt->hot_blocked_risky, (unsigned long long)n_blk_risky);
        uint32_t kc = 0;
        while (kc < 7 && (k_total >> (14 + kc))) kc++;
        unsigned long long *hc = G.wcnt->hot_class[kc];
        atomicAdd(&hc[0], 1ull); atomicAdd(&hc[1], t_calls); atomicAdd(&hc[2], n_tests); atomicAdd(&hc[3], (unsigned long long)n_steps); atomicAdd(&hc[4], (unsigned long long)k_work);
        atomicAdd(&hc[5], (unsigned long long)k_tests); atomicAdd(&hc[6], (unsigned long long)k_rep); atomicAdd(&hc[7], (unsigned long long)k_rec); atomicAdd(&hc[8], (unsigned long long)k_full); atomicAdd(&hc[9], (unsigned long long)k_blk);
        atomicAdd(&hc[10], (unsigned long long)k_noent); atomicAdd(&hc[11], (unsigned long long)(__builtin_readcyclecounter() - k_t0)); atomicAdd(&hc[12], t_skip); atomicAdd(&hc[13], (unsigned long long)k_und);
#ifdef NLZM_PROFILE
        for (int z = 0; z < 7; z++) atomicAdd(&hc[14 + z], t_sec[z]);
#endif
    }
#undef NLZM_HOT_STAMP
}

// A worker lane.  Bin b holds, in ascending order, the positions of every BT4 head h with h % bins == b; lane b walks it.
//
// Whether BT4 runs at a position the pre-filter marked is the finder stage's decision, and the lane is usually there first.
// It does not wait.  It assumes the decision -- "skip" if the positions before AND behind it are marked as well (inside a long
// repeat), "call" otherwise -- and goes
// (Until round 5 the position before alone decided the assumption: right for 97 % of those, and the 3 % were one position per long
//  repeat -- its LAST marked one, where the carried match has dropped below 64 bytes (:1514) and BT4 runs again: 520,000 wrong
//  assumptions in 300 MB of text, each a take-back, the calls behind it made again and the finder stage waiting for them.  A marked
//  position is a 65-gram that occurred before: the run's first and last position are called, the ones between skipped --
//  tests/host_sim/sim2 NLZM_SIM_RULES=1 tabulates the four cases: 9,203 / 220 / 220 / 5 called, 0 / 5 / 5 / 28,390 skipped.)
// on with the next positions of its bin, up to kAhead calls whose fate is open.  A call assumed to happen is made, every
// store noting what it replaced; one assumed not to happen is made without its stores, which are noted instead.  The first
// open call is an undecided position; its RESULT does not depend on its own decision and goes out at once.  The results of
// the calls behind it do (their trees contain it, or do not), so they are held back: the finder stage never sees a result
// that a decision could still change.  A decision as assumed: the held results go out as they are.  Otherwise the stores
// of the calls behind the position are taken back, latest first, the position's own stores are taken back or made, and
// the lane goes on from the position behind it.
// Per call, in LDS (interleaved by thread): the first four record-setters and what the lane needs to publish or drop it.
__device__ __forceinline__ void worker_role(const Geom &g, const Globals &G, uint32_t c0, uint32_t c1, uint32_t wblocks, uint32_t wblock)
{
    // The bin-taking lanes sit on the first NLZM_LPW lanes of the block's first waves: the lanes of a wave run their calls in lockstep, so a
    // lane that gets something to do waits for the longest call its wave is busy with -- the fewer lanes share a wave, the shorter.
    constexpr uint32_t kLpw = NLZM_LPW;
    const uint32_t lane_waves = (G.wthreads + kLpw - 1) / kLpw;
    if ((threadIdx.x >> 6) >= lane_waves) {
        // the waves behind the bin-taking lanes: one hot bin each (hot bin k: wave k / wblocks of worker block k % wblocks)
        if (!G.hot_list) return;
        const uint32_t k = ((threadIdx.x >> 6) - lane_waves) * wblocks + wblock;
        if (k >= G.hot_list[0]) return;
        worker_role_hot(g, G, c0, c1, G.hot_list[1 + k], k);
        return;
    }
    const uint32_t lane_in_block = (threadIdx.x >> 6) * kLpw + (threadIdx.x & 63u);
    const bool takes_bin = (threadIdx.x & 63u) < kLpw && lane_in_block < G.wthreads;
    const uint32_t gl = wblock * G.wthreads + (takes_bin ? lane_in_block : 0u);
    // What a descent touches, as values of this role (made opaque): left as kernel arguments, the compiler re-loads them
    // from the argument segment inside the test loop -- two 64-byte scalar loads and their waits per test -- rather than keep them.
    unsigned long long p0 = (unsigned long long)G.in, p1 = (unsigned long long)G.bt_heads, p2 = (unsigned long long)G.bt_tree,
                       p3 = (unsigned long long)G.bt_ready, p4 = (unsigned long long)G.bt_pairs;
    uint32_t q0 = G.batch_a0, q1 = g.bt_shift, q2 = g.wmask, q3 = g.bt_tmask;
    asm volatile("" : "+s"(p0), "+s"(p1), "+s"(p2), "+s"(p3), "+s"(p4));
    asm volatile("" : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3));
    // (re-typed as global pointers: accesses through pointers of unknown kind would be flat ones)
#define NLZM_AS_GLOBAL(T, x) ((T *)(__attribute__((address_space(1))) T *)(x))
    BtView B{ NLZM_AS_GLOBAL(const uint8_t, p0), NLZM_AS_GLOBAL(uint32_t, p1), NLZM_AS_GLOBAL(uint32_t, p2), NLZM_AS_GLOBAL(uint32_t, p3),
              NLZM_AS_GLOBAL(uint32_t, p4), q0, q1, q2, q3 };
    B.pstride = G.bt_pstride; B.ext = G.bt_ext; B.ext_cur = G.bt_ext_cur; B.ext_cap = G.bt_ext_cap; B.fail_word = G.abort_word;   // (touched at a position's fifth pair and later: rare)
    uint32_t *const undo_base = NLZM_AS_GLOBAL(uint32_t, (unsigned long long)G.bt_undo) + (unsigned long long)gl * (kAhead * kUndoCap * 2);
#undef NLZM_AS_GLOBAL
    bool active = takes_bin && gl < G.nheads && !(G.hot_of_bin && G.hot_of_bin[gl < G.nheads ? gl : 0]);        // (a hot bin has a wave of its own)
    uint32_t c = c0;
    bool loaded = false;
    uint32_t i0 = 0, e0 = 0;
    uint32_t la_end = 0;            // absolute end of the chunk's lookahead
    unsigned long long n_calls = 0, n_tests = 0, n_cmp = 0, n_open = 0, n_wait = 0, n_cyc = 0, n_cyc_tests = 0, n_back = 0, n_redo = 0;
    // the calls whose fate is open: a ring of kAhead entries, `first` the oldest (an undecided position), `nq` of them
    uint32_t first = 0, nq = 0, head_a = 0;
    uint32_t *const my = (uint32_t *)&g_v2_lds + threadIdx.x;
    enum : uint32_t { kWa = 8, kWtests = 9, kWcmp = 10, kWcount = 11, kWbest = 12, kWbestd = 13, kWundo = 14, kWinfo = 15 };   // (info: marked | bin index << 1 | assumed to be skipped << 28)
    unsigned long long t_wait0 = 0;
    uint32_t idle = 0, prog_seen = 0;
    bool fail = false;

    while (__any(active)) {
        bool waiting = false;
        uint32_t job = 0, ja = 0, jh = 0, jlen = 0, jinfo = 0;      // job 1: a call that stands; 2: one whose fate is open
        if (active) {
            // ---- the oldest open call: is its decision in?
            if (nq) {
                const uint32_t f = LaneIO::ld_agent(G.bt_flag + (head_a - G.batch_a0));
                const bool hdry = (my[(first * kEntryWords + kWinfo) * 512] >> 28) & 1u;
                if ((f == kFlagCall && !hdry) || (f == kFlagSkip && hdry)) {
                    // as assumed; so are the calls behind it up to the next undecided position, whose results go out now
                    for (uint32_t k = 0; k < kAhead; k++) {
                        uint32_t *e = my + (first * kEntryWords) * 512;
                        if (!((e[kWinfo * 512] >> 28) & 1u)) { n_calls++; n_tests += e[kWtests * 512]; n_cmp += e[kWcmp * 512]; }
                        first = first + 1 == kAhead ? 0u : first + 1; nq--;
                        if (!nq) break;
                        e = my + (first * kEntryWords) * 512;
                        const uint32_t ea = e[kWa * 512], einfo = e[kWinfo * 512];
                        if (!((einfo >> 28) & 1u)) {             // (a position assumed to be skipped has no result)
                            LaneSink sk{ e, B.pairs + (unsigned long long)(ea - B.batch_a0) * (2 * B.pstride), e[kWcount * 512] & 0x1FFu, e[kWbest * 512], e[kWbestd * 512], e[kWcount * 512] >> 9 };
                            sk.publish(B.ready + (unsigned long long)(ea - B.batch_a0) * kBtRec, e[kWtests * 512]);
                        }
                        if (einfo & 1u) {                           // a marked position: the new oldest, unless its decision is in too, and as assumed
                            head_a = ea;
                            const uint32_t f2 = LaneIO::ld_agent(G.bt_flag + (ea - G.batch_a0));
                            const bool edry = (einfo >> 28) & 1u;
                            if (!((f2 == kFlagCall && !edry) || (f2 == kFlagSkip && edry))) break;      // (otherwise: the next round)
                        }
                    }
                } else if (f == kFlagCall || f == kFlagSkip) {
                    // not as assumed: the stores of the calls behind it are taken back, latest first, its own are taken back
                    // ("skip") or made ("call"); the lane goes on behind it
                    uint32_t resume = i0;
                    for (uint32_t k = nq; k-- > 0;) {
                        const uint32_t slot = first + k >= kAhead ? first + k - kAhead : first + k;
                        const uint32_t *e = my + (slot * kEntryWords) * 512;
                        const uint32_t *u = undo_base + slot * (kUndoCap * 2);
                        if (k == 0 || !((e[kWinfo * 512] >> 28) & 1u)) {     // (a call behind it that was assumed not to happen has stored nothing)
                            for (uint32_t j = e[kWundo * 512]; j-- > 0;) {
                                const unsigned long long w = *(const unsigned long long *)(u + 2 * j);
                                const uint32_t t = (uint32_t)w, v = (uint32_t)(w >> 32);
                                if (t >> 31) B.heads[t & 0x7FFFFFFFu] = v; else B.tree[t] = v;
                            }
                        }
                        if (k == 1) resume = (e[kWinfo * 512] & 0xFFFFFFu) >> 1;     // (the entry behind it: made again)
                        if (k == 0 && hdry) resume = (e[kWinfo * 512] & 0xFFFFFFu) >> 1;    // (assumed to be skipped and called: nothing was made of it -- the lane goes on AT it)
                        if (k) n_redo++;
                    }
                    n_back++;
                    i0 = resume; nq = 0;
                }
            }
            // ---- the next position of the bin
            if (nq < kAhead) {
                if (!loaded) {
                    if (c >= c1) { if (!nq) active = false; }
                    else {
                        const uint32_t *off = G.bin_off + (unsigned long long)(c - c0) * (G.nheads + 1);
                        i0 = off[gl]; e0 = off[gl + 1];
                        const unsigned long long chunk_abs = (unsigned long long)c * g.chunk_size;
                        const unsigned long long remain = g.n - chunk_abs;
                        la_end = (uint32_t)(chunk_abs + (remain < g.feed ? remain : g.feed));
                        loaded = true;
                    }
                }
                if (active && loaded) {
                    const uint32_t *pos = G.bin_pos + (unsigned long long)(c - c0) * g.chunk_size * 2;
                    if (i0 >= e0) { if (!nq) { c++; loaded = false; } }      // (open calls: the chunk is left when they are settled)
                    else {
                        const unsigned long long pe = *(const unsigned long long *)(pos + 2 * i0);
                        ja = (uint32_t)pe; jh = (uint32_t)(pe >> 32) & 0x1FFFFFFFu; jlen = umin(la_end - ja, kMatchMax);
                        jinfo = (i0 << 1) | (uint32_t)(pe >> 63);
                        i0++;
                        if (pe >> 63) {
                            // a marked position.  Decision in already (the finder stage is ahead of this lane, as it is inside nice
                            // regions, :1529): act on it; otherwise the call is made with its fate open
                            const uint32_t f = LaneIO::ld_agent(G.bt_flag + (ja - G.batch_a0));
                            job = f == kFlagSkip ? 0u : ((f == kFlagCall && !nq) ? 1u : 2u);
#ifdef NLZM_EXP_SKIP_ANY
                            if (f != kFlagCall && ((pe >> 61) & 3u) != 0u) jinfo |= 1u << 28;
#else
                            if (f != kFlagCall && ((pe >> 61) & 3u) == 3u) jinfo |= 1u << 28;
#endif
                              // (the positions before AND behind are marked too: assumed to be skipped)
                        } else job = nq ? 2u : 1u;
                    }
                }
            }
            if (!job && nq) {
                waiting = true;
                n_wait++;
                if ((++idle & 1023u) == 0) {
                    if (LaneIO::ld_agent(G.abort_word)) active = false;
                    // give up only when the finder stage itself has not moved for 30 s (a lane that is far ahead of it
                    // waits as long as the launch takes)
                    const unsigned long long now = wall_clock64();
                    const uint32_t prog = G.progress ? LaneIO::ld_agent(G.progress) : 0u;
                    if (!t_wait0 || prog != prog_seen) { t_wait0 = now; prog_seen = prog; }
                    else if (now - t_wait0 > 3000000000ull) { fail = true; }
                }
            }
            if (fail) { LaneIO::st_agent(G.abort_word, 2u); active = false; job = 0; }
        }
        if (job) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            const unsigned long long bi = ja - B.batch_a0;
            const uint32_t slot = first + nq >= kAhead ? first + nq - kAhead : first + nq;     // (job 1: nq = 0, any slot will do)
            uint32_t *e = my + (slot * kEntryWords) * 512;
            LaneSink sink{ e, B.pairs + bi * (2 * B.pstride) };
            uint32_t tests = 0, cmpb = 0, nu = 0;
            // An undecided position that is assumed to be skipped: NOTHING is made of it (round 6; until then a descent without its stores, whose
            // result went out in case the decision was "call").  It is an open call without a result and without notes: what is behind it is
            // held until its decision; "call" is a wrong assumption like any other, and the lane then goes on AT this position.  The finder stage
            // says "call" as soon as the positions in front are settled, without waiting for a result (nlzm_v2.h, the block's wait loop).
            const bool park = job == 2 && ((jinfo >> 28) & 1u);
            if (!park) bt_descent(B, ja, jh, jlen, job == 2, false, undo_base + slot * (kUndoCap * 2), nu, sink, tests, cmpb);
            if (job == 1) {
                sink.publish(B.ready + bi * kBtRec, tests);
                n_calls++; n_tests += tests; n_cmp += cmpb;
                n_cyc += __builtin_readcyclecounter() - t0; n_cyc_tests += tests;
            } else {
                if (!nq) { if (!park) sink.publish(B.ready + bi * kBtRec, tests); head_a = ja; t_wait0 = 0; idle = 0; }    // (the oldest: its result stands whatever is decided)
                e[kWa * 512] = ja; e[kWtests * 512] = tests; e[kWcmp * 512] = cmpb; e[kWcount * 512] = sink.count | (sink.ext_idx << 9);
                e[kWbest * 512] = sink.best; e[kWbestd * 512] = sink.best_d; e[kWundo * 512] = nu; e[kWinfo * 512] = jinfo;
                nq++; if (!park) n_open++;
            }
        }
        if (!__any(active && !waiting)) __builtin_amdgcn_s_sleep(8);
    }
    if (nq) {               // left with calls open: the launch failed; say where (error report)
        atomicAdd(&G.wcnt->stuck_lanes, 1ull);
        atomicMax(&G.wcnt->stuck_pos_inv, (unsigned long long)(uint32_t)~head_a);
    }
    // counters: one atomic per wave
    for (int m = 32; m >= 1; m >>= 1) {
        n_calls += __shfl_xor(n_calls, m, 64); n_tests += __shfl_xor(n_tests, m, 64); n_cmp += __shfl_xor(n_cmp, m, 64);
        n_open += __shfl_xor(n_open, m, 64); n_wait += __shfl_xor(n_wait, m, 64);
        n_cyc += __shfl_xor(n_cyc, m, 64); n_cyc_tests += __shfl_xor(n_cyc_tests, m, 64);
        n_back += __shfl_xor(n_back, m, 64); n_redo += __shfl_xor(n_redo, m, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&G.wcnt->bt_calls, n_calls); atomicAdd(&G.wcnt->bt_tests, n_tests); atomicAdd(&G.wcnt->cmp_bytes, n_cmp);
        atomicAdd(&G.wcnt->dry_runs, n_open); atomicAdd(&G.wcnt->flag_waits, n_wait);
        atomicAdd(&G.wcnt->call_cycles, n_cyc); atomicAdd(&G.wcnt->call_tests, n_cyc_tests);
        atomicAdd(&G.wcnt->spec_calls, n_back); atomicAdd(&G.wcnt->spec_good, n_redo);
    }
}

// ---------------------------------------------------------------------------
// the persistent launch of the three-stage pipeline (nlzm_v2.h): blocks 0, 1, 2 of a stream are its finder, table and
// parser stage (one wave each: the lanes are the positions of a block / the edges of two nodes), the others its BT4
// worker lanes.  512-thread blocks with > 80 KB of LDS: one block per CU; the grid is kept below the CU count, so every
// block is resident at once (the stages wait on each other through progress words in HBM).
// ---------------------------------------------------------------------------
constexpr uint32_t kV2Roles = 3 + v2::kHelpers; // finder, table, parser, helper parsers (these leave at once where a stream has no HelpBox)
__device__ __forceinline__ void pipeline2_roles(const Geom &g, const Globals &G, const v2::GlobalsV2 &V, uint32_t c0, uint32_t c1,
                                                uint32_t local_block, uint32_t wblocks)
{
    if (local_block < kV2Roles) {
        // finder: one wave; table: kTW or kTWWide waves (the shape the launch before chose), a block each; parser: kPW waves on the same block
        if (threadIdx.x >= (local_block >= 2 ? v2::kParserThreads : (local_block == 1 ? 64u * v2::table_waves(((const v2::StateV2 *)V.state)->tb_wide[G.launch_par & 1u]) : 64u))) return;
        if (local_block == 0) { v2::Finder r; r.g = g; r.G = G; r.V = V; r.run(c0, c1); }
        else if (local_block == 1) { v2::Table r; r.g = g; r.G = G; r.V = V; r.run(c0, c1); }
        else if (local_block == 2) { v2::Parser r; r.g = g; r.G = G; r.V = V; r.run(c0, c1); }
        else if (V.hb) { v2::Parser r; r.g = g; r.G = G; r.V = V; r.run_helper(c0, local_block - 3); }
    } else {
        worker_role(g, G, c0, c1, wblocks, local_block - kV2Roles);
    }
}
__global__ __launch_bounds__(512) void pipeline2_kernel(Geom g, Globals G, v2::GlobalsV2 V, uint32_t c0, uint32_t c1)
{
    // blocks 0, 8, 16, 24, 32 are the stages: workgroups go to the XCDs round-robin, so they share XCD 0 (and its L2)
    const uint32_t b = blockIdx.x;
    const bool spread = gridDim.x > 8 * (kV2Roles - 1);                                   // (a small grid: the first blocks are the stages)
    const uint32_t before = b ? (b + 7) / 8 < kV2Roles ? (b + 7) / 8 : kV2Roles : 0u;     // stage blocks below b
    const uint32_t local = (b % 8 == 0 && b / 8 < kV2Roles && spread) ? b / 8 : (spread ? kV2Roles + (b - before) : b);
    pipeline2_roles(g, G, V, c0, c1, local, gridDim.x - kV2Roles);
}
struct Stream2Args { Geom g; Globals G; v2::GlobalsV2 V; uint32_t c0, c1; v2::RoundSnap *snap; };
constexpr uint32_t kMaxStreams2PerLaunch = 64;          // (the pack lives in device memory: the kernel-argument segment holds 4 KB)
__global__ __launch_bounds__(512) void pipeline2_multi_kernel(const Stream2Args *__restrict__ pack, uint32_t bps)
{
    uint32_t s, local;
    multi_block_of(gridDim.x, bps, blockIdx.x, s, local);       // (a stream's blocks on one XCD: nlzm_core.h)
    Stream2Args a = pack[s];
    // pointers read from memory: tell the compiler they are global ones (flat accesses would count on both wait counters)
#define NLZM_GLOBAL_PTR(p) p = (decltype(p))(__attribute__((address_space(1))) std::remove_pointer_t<decltype(p)> *)(unsigned long long)(p)
    NLZM_GLOBAL_PTR(a.G.in); NLZM_GLOBAL_PTR(a.G.rkhash); NLZM_GLOBAL_PTR(a.G.ht2); NLZM_GLOBAL_PTR(a.G.ht3); NLZM_GLOBAL_PTR(a.G.rk_table);
    NLZM_GLOBAL_PTR(a.G.bt_heads); NLZM_GLOBAL_PTR(a.G.bt_tree); NLZM_GLOBAL_PTR(a.G.persist); NLZM_GLOBAL_PTR(a.G.syms); NLZM_GLOBAL_PTR(a.G.bits);
    NLZM_GLOBAL_PTR(a.G.fmeta); NLZM_GLOBAL_PTR(a.G.cap_words); NLZM_GLOBAL_PTR(a.G.cap_used); NLZM_GLOBAL_PTR(a.G.bt_ready); NLZM_GLOBAL_PTR(a.G.bt_pairs); NLZM_GLOBAL_PTR(a.G.bt_ext); NLZM_GLOBAL_PTR(a.G.bt_ext_cur);
    NLZM_GLOBAL_PTR(a.G.bt_flag); NLZM_GLOBAL_PTR(a.G.unc); NLZM_GLOBAL_PTR(a.G.bin_off); NLZM_GLOBAL_PTR(a.G.bin_pos); NLZM_GLOBAL_PTR(a.G.abort_word);
    NLZM_GLOBAL_PTR(a.G.progress); NLZM_GLOBAL_PTR(a.G.wcnt); NLZM_GLOBAL_PTR(a.G.bt_undo); NLZM_GLOBAL_PTR(a.G.hot_of_bin); NLZM_GLOBAL_PTR(a.G.hot_list); NLZM_GLOBAL_PTR(a.G.hot_undo);
    NLZM_GLOBAL_PTR(a.V.ft); NLZM_GLOBAL_PTR(a.V.tp); NLZM_GLOBAL_PTR(a.V.tf); NLZM_GLOBAL_PTR(a.V.hx); NLZM_GLOBAL_PTR(a.V.state); NLZM_GLOBAL_PTR(a.V.hb);
#undef NLZM_GLOBAL_PTR
    pipeline2_roles(a.g, a.G, a.V, a.c0, a.c1, local, bps - kV2Roles);
}
// Either side of a shared launch whose successor is queued before the host has looked at it (nlzm_hip.cpp, blocks_step_impl):
// round_open sets every stream's progress words to the launch's first position (what step_pre's copy does for a launch of its own),
// round_close copies aside what the host checks afterwards -- the next launch resets and advances the originals.
#define NLZM_G(T, x) ((__attribute__((address_space(1))) T *)(unsigned long long)(x))       // (pointers read from memory: global ones, not flat)
__global__ __launch_bounds__(256) void round_open_kernel(const Stream2Args *__restrict__ pack)
{
    const Stream2Args &a = pack[blockIdx.x];
    auto *w = NLZM_G(uint32_t, a.V.hx);
    const uint32_t a0 = a.G.batch_a0;
    constexpr uint32_t kWords = sizeof(v2::Hx) / 4;
    // A stream whose launch before this one failed (the sticky Persist::error: this launch was queued before the host had looked): its stages and worker
    // lanes must leave at once -- the error word is what every wait of every stage looks at, the abort word what the worker lanes look at.  Cleared, the
    // table stage stood at the finder's progress word until the 30 s bound of its wait.
    const uint32_t sticky = NLZM_G(const uint32_t, a.G.persist)[offsetof(Persist, error) / 4];
    for (uint32_t i = threadIdx.x; i < kWords; i += 256) {
        uint32_t v = 0;
        if (i == offsetof(v2::Hx, f_pos) / 4 || i == offsetof(v2::Hx, t_pos) / 4 || i == offsetof(v2::Hx, t_out) / 4 || i == offsetof(v2::Hx, p_pos) / 4 ||
            i == offsetof(v2::Hx, p_seg) / 4 || i == offsetof(v2::Hx, p_seg) / 4 + 1) v = a0;         // (p_seg = a0 << 32 | a0)
        if (i == offsetof(v2::Hx, err) / 4) v = sticky;
        w[i] = v;
    }
    if (threadIdx.x == 0 && sticky && a.G.abort_word) *NLZM_G(uint32_t, a.G.abort_word) = 1u;
}
__global__ __launch_bounds__(256) void round_close_kernel(const Stream2Args *__restrict__ pack)
{
    const Stream2Args &a = pack[blockIdx.x];
    const auto *w = NLZM_G(const uint32_t, a.V.hx);
    auto *o = NLZM_G(uint32_t, a.snap);
    constexpr uint32_t kHead = offsetof(v2::RoundSnap, hx) / 4;
    for (uint32_t i = threadIdx.x; i < sizeof(v2::Hx) / 4; i += 256) o[kHead + i] = w[i];
    if (threadIdx.x == 0) {
        const auto *ps = NLZM_G(const uint32_t, a.G.persist);
        o[0] = ps[offsetof(Persist, error) / 4]; o[1] = ps[offsetof(Persist, next_chunk) / 4];
        o[2] = a.G.abort_word ? *NLZM_G(const uint32_t, a.G.abort_word) : 0u; o[3] = 0;
    }
}
#undef NLZM_G

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

void launch_pipeline2(const Geom &g, const Globals &G, const v2::GlobalsV2 &V, uint32_t c0, uint32_t c1, uint32_t worker_blocks, hipStream_t st)
{
    hipLaunchKernelGGL(pipeline2_kernel, dim3(kV2Roles + worker_blocks), dim3(512), 0, st, g, G, V, c0, c1);
}
unsigned long long stream2_pack_size() { return sizeof(Stream2Args) * kMaxStreams2PerLaunch; }
uint32_t stream2_pack_capacity() { return kMaxStreams2PerLaunch; }
uint32_t pipeline2_role_blocks() { return kV2Roles; }
void fill_stream2_args(void *host_pack, uint32_t i, const Geom &g, const Globals &G, const v2::GlobalsV2 &V, uint32_t c0, uint32_t c1, v2::RoundSnap *snap)
{
    Stream2Args &a = ((Stream2Args *)host_pack)[i];
    a.g = g; a.G = G; a.V = V; a.c0 = c0; a.c1 = c1; a.snap = snap;
}
void launch_round_open(const void *dev_pack, uint32_t nstreams, hipStream_t st)
{
    hipLaunchKernelGGL(round_open_kernel, dim3(nstreams), dim3(256), 0, st, (const Stream2Args *)dev_pack);
}
void launch_round_close(const void *dev_pack, uint32_t nstreams, hipStream_t st)
{
    hipLaunchKernelGGL(round_close_kernel, dim3(nstreams), dim3(256), 0, st, (const Stream2Args *)dev_pack);
}
// dev_pack: the streams' arguments in device memory (stream2_pack_size() bytes, filled with fill_stream2_args on the host)
void launch_pipeline2_multi(const void *dev_pack, uint32_t nstreams, uint32_t worker_blocks, hipStream_t st)
{
    hipLaunchKernelGGL(pipeline2_multi_kernel, dim3(nstreams * (kV2Roles + worker_blocks)), dim3(512), 0, st, (const Stream2Args *)dev_pack,
                       kV2Roles + worker_blocks);
}

void launch_prefilter(const uint8_t *in, unsigned long long n, uint32_t a0, uint32_t a1, uint32_t wmask, uint32_t t_bits, uint32_t t_bitmap,
                      uint32_t m_bits, uint32_t *T, uint32_t *M, uint32_t *hbuf, uint32_t *hbuf2, uint8_t *c1, uint8_t *unc, hipStream_t st)
{
    if (a1 <= a0) return;
    const uint32_t cnt = a1 - a0;
    hipLaunchKernelGGL(prefilter_hash_kernel, dim3((cnt + 1023) / 1024), dim3(256), 0, st, in, n, a0, a1, wmask, t_bits, t_bitmap, m_bits,
                       T, M, hbuf, hbuf2, c1);
    hipLaunchKernelGGL(prefilter_mark_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, n, a0, a1, m_bits, M, hbuf, c1, unc);
    hipLaunchKernelGGL(prefilter_insert_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, n, a0, a1, t_bits, t_bitmap, m_bits, T, M, hbuf, hbuf2);
}

void launch_bin(const uint8_t *in, const Geom &g, uint32_t c0, uint32_t nchunks, uint32_t nheads, uint32_t *off, uint32_t *cur,
                uint32_t *pos, const uint8_t *unc, uint32_t batch_a0, hipStream_t st)
{
    if (!nchunks) return;
    constexpr uint32_t kBinsInLdsMax = 36864;
    const uint32_t lds_bins = nheads <= kBinsInLdsMax ? nheads : 0u;
    // (more than the default 64 KB of dynamic LDS has to be allowed for the function, on the current device)
    (void)hipFuncSetAttribute((const void *)bin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kBinsInLdsMax * 4));
    hipLaunchKernelGGL(bin_kernel, dim3(nchunks), dim3(1024), lds_bins * 4, st, in, g, c0, nheads, off, cur, pos, unc, batch_a0, lds_bins);
}

void launch_hot_select(const uint32_t *off, uint32_t nchunks, uint32_t nheads, uint32_t hmax, uint32_t min_count, uint32_t *hot_of_bin,
                       uint32_t *hot_list, WorkerCounters *wcnt, hipStream_t st)
{
    hipLaunchKernelGGL(hot_select_kernel, dim3(1), dim3(1024), 0, st, off, nchunks, nheads, hmax, min_count, hot_of_bin, hot_list, wcnt);
}

unsigned long long worker_hot_undo_bytes_per_wave() { return 64ull * kUndoCap * 8; }
unsigned long long worker_undo_bytes_per_lane() { return (unsigned long long)kAhead * kUndoCap * 8; }

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
