// nlzm_core.h -- data layout shared by the kernels, the host pipeline and the simulation: geometry, counters, the
// state that survives between launches, the hand-off record between the BT4 worker lanes and the finder stage, and
// BT4 itself (MatchFinderBT::FindAndUpdate, NLZM.cpp:978-1022) as the worker lanes run it.
// The pipeline stages (finder / table / parser) are in nlzm_v2.h.
#pragma once

#include <stdint.h>

#ifndef NLZM_HD
#define NLZM_HD __host__ __device__ __forceinline__
#endif
#ifndef NLZM_HDN
#define NLZM_HDN __host__ __device__
#endif
#define NLZM_RARE(x) __builtin_expect(!!(x), 0)     // block placement: the common path falls through

namespace nlzm {

constexpr uint32_t kMatchMin = 2;       // :733
constexpr uint32_t kMatchMax = 264;     // :737
constexpr uint32_t kNice = 64;          // :734
constexpr uint32_t kParseMax = 4096;    // :1458
constexpr uint32_t kNone = 0xFFFFFFFFu;

// ---- CDF contexts (Model, :1133-1146) flattened into one table -------------
constexpr uint32_t kCtxCmd = 0;                 // CDF2
constexpr uint32_t kCtxLitHi = 1;               // CDF4
constexpr uint32_t kCtxLitLo = 2;               // CDF4 x16
constexpr uint32_t kCtxLenDirect = 18;          // CDF3
constexpr uint32_t kCtxLenExtHi = 19;           // CDF4
constexpr uint32_t kCtxLenExtLo = 20;           // CDF4 x16
constexpr uint32_t kCtxSlotHi = 36;             // CDF3 x4
constexpr uint32_t kCtxSlotLo = 40;             // CDF3 x4x8
constexpr uint32_t kNumCtx = 72;
constexpr uint32_t kCdfStride = 18;             // 17 cells used

NLZM_HD uint32_t ctx_nsyms(uint32_t ctx)
{
    if (ctx == kCtxCmd) return 4;
    if (ctx == kCtxLenDirect || ctx >= kCtxSlotHi) return 8;
    return 16;
}

struct Geom {
    uint64_t n;             // input bytes
    uint32_t wbits, wmask;  // window (after auto-shrink, :1716-1718)
    uint32_t frame_bits, frame_size, chunk_size, feed;   // :1722-1725
    uint32_t ht3_shift;     // 32 - (12 + clamp(wbits,15,17) - 15)   :1751
    uint32_t bt_shift;      // 32 - (13 + clamp(wbits,16,20) - 16)   :1752
    uint32_t rk_shift;      // 32 - (15 + clamp(wbits,16,22) - 16)   :1753
    uint32_t tag_mask;      // (1 << (32 - wbits)) - 1               :899, :1036
    uint32_t nchunks;
    // BT4 node slots.  The reference indexes nodes by p & (W-1): position p+W reuses p's
    // slot, which is harmless there because p is out of every later window (:989 checks the
    // distance before it reads the pair).  Worker lanes of different heads run up to one
    // launch apart, so the slot space is widened to >= W + positions per launch: a lane that
    // is ahead can then never overwrite a node a lagging lane may still visit.
    uint32_t bt_tmask;
};

// Operation counters, accumulated per launch.
struct Counters {
    unsigned long long bt_calls, bt_tests, cmp_bytes, ht_rows, rk_probes, rk_inserts;
    unsigned long long positions, nice_positions, segments;
    unsigned long long n_literal, n_dict, n_rep, rans_syms, bit_ops, frames, shifts;
    unsigned long long uncertain_positions;
    unsigned long long stale_ht, stale_rk, bt_slow;     // look-ahead slots that had to take the direct path (diagnostics)
    unsigned long long guess_used, guess_late, guess_wrong;   // rep-list waves: guessed rep set right / not seen in time / wrong
};

// State that survives between launches (one per stream), in HBM.
struct Persist {
    uint16_t cdf[kNumCtx * kCdfStride];
    uint32_t rep[4];
    uint32_t mt_max;
    uint32_t mt_delta[kMatchMax + 8];
    uint32_t rk_from, rk_to, rk_len, rk_end;
    unsigned long long reb_base;
    uint32_t next_chunk;
    uint32_t error;             // 0 ok; see kErr*
    uint32_t error_info[3];
    Counters cnt;
    unsigned long long prof[128];    // cycles per master phase (diagnostic builds: NLZM_PROFILE); 16..21: wait/total cycles per wave
};

constexpr uint32_t kErrFrameOverflow = 1;
constexpr uint32_t kErrTimeout = 2;
constexpr uint32_t kErrInternal = 3;
constexpr uint32_t kErrCapture = 4;

// Block mode: which stream, and which of its blocks (0, 1, 2: finder, table, parser stage; 3..: worker CUs), workgroup b of a
// launch of `grid` workgroups is, with bps blocks per stream.  Workgroups go to the eight XCDs round-robin (blocks b and b + 8
// share one: for speed only, nothing depends on it): with a multiple of eight streams every stream's blocks are dealt to ONE
// XCD, so that its hand-off rings, BT4 records and tree are served from one L2 -- stream s = x + 8 * group on XCD x.
NLZM_HD void multi_block_of(uint32_t grid, uint32_t bps, uint32_t b, uint32_t &stream, uint32_t &local)
{
    if ((grid / bps) % 8 == 0) { const uint32_t x = b % 8, j = b / 8; stream = x + 8 * (j / bps); local = j % bps; }
    else { stream = b / bps; local = b % bps; }
}

struct FrameMeta {
    uint32_t nsyms, nbits_bytes, num_ops, out_len;
};

// BT4 result of one position, written by a worker lane, read by the finder and the table stage: ONE 64-byte record of four
// 16-byte quads, each written by one store and each valid by itself (the array is cleared before a launch):
//   quad 0   ready | tests << 9 | count,  distance and length of the longest record-setter (0, 0: none),  0
//   quad 1   d0, l0, d1, TAG         quad 2   l1, d2, l2, TAG         quad 3   d3, l3, 0, TAG
// (d, l) = the record-setters of the descent (a descent visits candidates by increasing distance, so only a longer match
// changes the table: :835-852), the first four of them.  A reader that has seen the ready bit takes the rest of quad 0 as it
// is; quads 1..3 it takes when their TAG is there.  The rare fifth and later record-setters go to
//   bt_pairs[i * 2*kBtMaxPairs + 2k ...]   (k >= 4; written, and drained, before quad 0)
// whose stride is the worst case (one pair per test, <= 256 tests, :777): 2 KiB per position, sized for HBM, so no allocator
// and no overflow path exists.
constexpr uint32_t kBtMaxPairs = 256;
// A position's reservation in bt_pairs may be shorter than that (Globals::bt_pstride pairs; block mode: 32, a single stream: all 256): pair k
// >= bt_pstride of a position then lives in an EXTENSION block of kBtMaxPairs - bt_pstride pairs taken from the launch's arena (bt_ext) by
// an atomic increment of its cursor at the position's first such pair; the block's index + 1 is word 14 of the record (0: none).  32 streams
// of a block set reserved 64 GB of pair lists for a dozen pairs at four positions in ten.
template <class IO>
NLZM_HD uint32_t *bt_pair_ptr(uint32_t *pairs, uint32_t pstride, uint32_t *ext, uint32_t *ext_cur, uint32_t ext_cap, uint32_t &ext_idx, uint32_t k)
{
    if (k < pstride) return pairs + 2 * k;
    if (!ext_idx) ext_idx = IO::atomic_inc(ext_cur) + 1;
    if (ext_idx > ext_cap) return nullptr;              // the arena is used up: the caller drops the pair, the host sees the cursor (Hx::ext_cur) beyond the arena
    return ext + (unsigned long long)(ext_idx - 1) * (2 * (kBtMaxPairs - pstride)) + 2 * (k - pstride);
}
constexpr uint32_t kBtReady = 0x80000000u;
constexpr uint32_t kBtTag = 0x80000000u;
constexpr uint32_t kFlagCall = 1, kFlagSkip = 2;
constexpr uint32_t kBtRec = 16;         // words per bt_ready record
// where pair k < 4 sits in the record: distance, length
NLZM_HD uint32_t bt_rec_d(uint32_t k) { return k == 0 ? 4u : (k == 1 ? 6u : (k == 2 ? 9u : 12u)); }
NLZM_HD uint32_t bt_rec_l(uint32_t k) { return k == 0 ? 5u : (k == 1 ? 8u : (k == 2 ? 10u : 13u)); }

struct WorkerCounters {
    unsigned long long bt_calls, bt_tests, cmp_bytes, dry_runs, flag_waits;
    unsigned long long call_cycles, call_tests;     // diagnostics: cycles / tests of the write-mode calls (per-lane clocks)
    unsigned long long stuck_lanes, stuck_pos_inv;  // lanes that left while waiting for a decision (a launch that failed); ~(smallest such position)
    unsigned long long hot_bins, hot_calls;         // hot bins over all launches; calls made by their waves
    unsigned long long spec_calls, spec_good;       // decisions "skip" that took calls back; calls behind the skipped position that were made again
    unsigned long long hot_steps, hot_blocked_dry, hot_blocked_risky;   // hot bins' waves: steps; steps in which the next call could not start (a call without stores on its way / a risky assumption open)
    // hot bins' waves by the size of the bin (class k: 8,192 << k positions of the launch and more; the last class is open), what their steps were spent on:
    //   0 waves, 1 calls, 2 tests, 3 steps, 4 steps in which some lane made a test, 5 tests made (summed over the lanes), 6 lane-steps repeated for a slot
    //   another call holds, 7 idle steps while a wrong assumption is taken back, 8 idle: every lane holds a call that waits for a decision, 9 idle: the next
    //   call may not start (dry / risky), 10 idle: no entry (the chunk's end, or its calls are held), 11 cycles, 12 entries skipped (decided "skip" when they came up),
    //   13 idle steps with an undecided position open (any cause)
    //   (4 .. 10 and 13 are counted by the profile build only, NLZM_PROFILE: counting them was a tenth of a step)
    //   14 .. 20 (profile build): cycles of a step by section -- the oldest undecided position and recovery, passing entries and the start, the step's loads
    //   until they are back (the round trip), a call's start and its test, the call's end and its result, accounting and watchdogs; 20: the end of the step before
    //   and the windows' upkeep (in front of 14)
    unsigned long long hot_class[8][21];
};

// Everything the master needs from HBM.
struct Globals {
    const uint8_t *in;          // input, followed by >= 16 padding bytes
    const uint32_t *rkhash;     // rkhash[a] = RK256 hash of in[a .. a+256)   (closed form of :798-799)
    uint32_t *ht2, *ht3, *rk_table;
    uint32_t *bt_heads, *bt_tree;   // absolute positions
    Persist *persist;
    // per-frame symbol/bit streams for the chunks of this launch (index = chunk - chunk0)
    uint32_t *syms;  unsigned long long syms_stride;   // words per frame
    uint8_t *bits;   unsigned long long bits_stride;   // bytes per frame
    FrameMeta *fmeta;
    uint32_t chunk0;
    // optional capture of match tables (stage test)
    uint32_t *cap_words; unsigned long long cap_cap, cap_lo, cap_hi; unsigned long long *cap_used;
    // ---- worker mode: BT4 on per-head worker lanes ------------------------------
    uint32_t workers;           // 0: BT4 runs inside the master workgroup
    uint32_t batch_a0;          // absolute position of the first byte of this launch
    uint32_t *bt_ready;         // [(a - batch_a0) * kBtRec]
    uint32_t *bt_pairs;         // [(a - batch_a0) * 2 * bt_pstride]
    uint32_t bt_pstride;        // pairs reserved per position in bt_pairs (kBtMaxPairs: every pair fits, no extension blocks)
    uint32_t *bt_ext;           // extension blocks of kBtMaxPairs - bt_pstride pairs
    uint32_t *bt_ext_cur;       // blocks taken in this launch (reset with the stages' progress words)
    uint32_t bt_ext_cap;        // blocks there are
    uint32_t *bt_flag;          // [a - batch_a0] master -> worker: kFlagCall / kFlagSkip
    const uint8_t *unc;         // [a - batch_a0] 1: whether BT4 runs at `a` is the master's call
    const uint32_t *bin_off;    // [chunk - chunk0][nheads + 1]
    const uint32_t *bin_pos;    // [chunk - chunk0][chunk_size][2] positions grouped by bin, ascending: position, BT4 head | unc << 31
    uint32_t nheads;            // bins = min(BT4 heads, worker lanes); head h belongs to bin h % bins
    uint32_t wthreads;          // lanes of a worker block that take bins (the first so many of its 512 threads)
    const uint32_t *hot_of_bin; // [nheads] nonzero: the bin has a wave of its own (hot_select_kernel); null: no such waves
    const uint32_t *hot_list;   // [0]: hot bins of this launch, [1 + k]: the k-th of them
    unsigned long long *hot_undo;   // [hot bin index][lane][kUndoCap]: the notes of the calls of a hot bin's wave
    uint32_t *bt_undo;          // [lane][calls with their fate open][kUndoCap]: slot and replaced value of every store of such a call
    uint32_t *abort_word;       // nonzero: every role leaves its loops
    const uint32_t *progress;   // the finder stage's position (its decisions are what a worker lane may wait for)
    WorkerCounters *wcnt;
    uint32_t table_shape;       // the table stage's shape (nlzm_v2.h): 0 by the data, 1 narrow fronts, 2 wide fronts
    uint32_t launch_par;        // the stream's launch number & 1: a launch reads the shape from slot launch_par and leaves the next one's in the other
    uint32_t test_fail;         // test only: nonzero -> the finder stage raises an error at the launch's start (the fault path of a block set's queued rounds)
    void *hook_user;            // host simulation only
};

NLZM_HD uint32_t match_min(uint32_t d)      // :813-821
{
    return 2u + (d >= (1u << 8)) + (d >= (1u << 12)) + (d >= (1u << 20));
}
NLZM_HD uint32_t hash4(uint32_t x) { return x * 987660757u; }    // :739
NLZM_HD uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
NLZM_HD uint32_t umax(uint32_t a, uint32_t b) { return a < b ? b : a; }
NLZM_HD uint32_t ilog2(uint32_t x) { return 31u - (uint32_t)__builtin_clz(x | 1u); }   // the reference's "clz32", :59-75

NLZM_HD uint32_t load32u(const uint8_t *p)
{
    uint32_t v; __builtin_memcpy(&v, p, 4); return v;
}
NLZM_HD unsigned long long load64u(const uint8_t *p)
{
    unsigned long long v; __builtin_memcpy(&v, p, 8); return v;
}

// log2 cost table, :103-124
NLZM_HD uint16_t log2_lut_entry(uint32_t i)
{
    if (i == 0) i = 1;
    uint32_t next = 1u << 16;
    uint16_t acc = 0;
    for (int s = 0; s < 32; s++) {
        const uint32_t v = (i * next) >> 8;
        const uint32_t nbits = 16u - ilog2(v);
        acc = (uint16_t)(acc + nbits - 1);
        next = v << (nbits - 1);
    }
    return acc;
}

// distance -> slot / extra bits (:1227-1243, :1299-1320)
NLZM_HD uint32_t dist_slot(uint32_t dv, uint32_t &nextra, uint32_t &extra)
{
    if (dv < 4) { nextra = 0; extra = 0; return dv; }
    const uint32_t nb = ilog2(dv) + 1, ab = nb - 2;
    nextra = ab;
    extra = dv & ((1u << ab) - 1);
    return ((nb - 1) << 1) + ((dv >> ab) & 1);
}

NLZM_HD void rep_add(uint32_t r[4], uint32_t d)     // :1160-1171
{
    if (r[0] == d || r[1] == d || r[2] == d || r[3] == d) return;
    r[3] = r[2]; r[2] = r[1]; r[1] = r[0]; r[0] = d;
}

// ---------------------------------------------------------------------------
// BT4: one call = MatchFinderBT::FindAndUpdate (:978-1022) on absolute positions.
// `Cmp` supplies the byte compare (wave-wide in the master, lane-serial in the
// worker kernel); `Sink` receives every (distance, length) the reference would
// pass to MatchTable::Update (:996-998).
// ---------------------------------------------------------------------------
// `kWrite` = false is a dry run: the same descent (the links a call rewrites are
// never read again by that call), all matches reported, nothing stored.
// What a call does to the tree: written at once (kWrite), dropped (a dry run), or noted down by a dry run so that the
// commit after the master's decision is a replay of stores instead of a second descent (the descent only reads what
// earlier calls of the same head wrote, and the head's lane makes no other call in between).
struct StoreNow {
    NLZM_HD void head(uint32_t *heads, uint32_t i, uint32_t v) const { heads[i] = v; }
    NLZM_HD void link(uint32_t *tree, uint32_t i, uint32_t v) const { tree[i] = v; }
};
struct StoreDrop {
    NLZM_HD void head(uint32_t *, uint32_t, uint32_t) const {}
    NLZM_HD void link(uint32_t *, uint32_t, uint32_t) const {}
};

template <class St, class Cmp, class Sink>
NLZM_HD void bt_find_and_update_st(uint32_t *heads, uint32_t *tree, uint32_t bt_shift, uint32_t wmask, uint32_t tmask,
                                   const uint8_t *in, uint32_t a /*abs pos*/, uint32_t h4, uint32_t max_len,
                                   Cmp &cmp, Sink &sink, uint32_t &n_tests, St &st)
{
    uint32_t pend_l = (a & tmask) << 1, pend_r = pend_l + 1;     // indices into tree[]
    uint32_t len_l = 0, len_r = 0;
    uint32_t sp = heads[h4 >> bt_shift];
    st.head(heads, h4 >> bt_shift, a);
    uint32_t tests = 256;                                       // :777, :988 (uint16 there; never wraps)
    while (sp != kNone && a > sp && a - sp <= wmask && tests-- > 0) {
        n_tests++;
        const uint32_t pair = (sp & tmask) << 1;
        const uint32_t pl = tree[pair], pr = tree[pair + 1];
        const uint32_t r = cmp(in + sp, in + a, umin(len_l, len_r), max_len);
        const uint32_t l = r & 0x7FFFFFFFu;
        if (l >= match_min(a - sp)) sink(a - sp, l);
        if (l == max_len) {                                     // :1000-1004
            st.link(tree, pend_l, pl); st.link(tree, pend_r, pr);
            return;
        }
        if (r >> 31) { st.link(tree, pend_l, sp); pend_l = pair + 1; sp = pr; len_r = l; }
        else         { st.link(tree, pend_r, sp); pend_r = pair;     sp = pl; len_l = l; }
    }
    st.link(tree, pend_r, kNone); st.link(tree, pend_l, kNone);    // :1020-1021
}

template <bool kWrite, class Cmp, class Sink>
NLZM_HD void bt_find_and_update(uint32_t *heads, uint32_t *tree, uint32_t bt_shift, uint32_t wmask, uint32_t tmask,
                                const uint8_t *in, uint32_t a /*abs pos*/, uint32_t h4, uint32_t max_len,
                                Cmp &cmp, Sink &sink, uint32_t &n_tests)
{
    if (kWrite) { StoreNow st; bt_find_and_update_st(heads, tree, bt_shift, wmask, tmask, in, a, h4, max_len, cmp, sink, n_tests, st); }
    else { StoreDrop st; bt_find_and_update_st(heads, tree, bt_shift, wmask, tmask, in, a, h4, max_len, cmp, sink, n_tests, st); }
}

// ---------------------------------------------------------------------------
// Worker lane: one BT4 call for absolute position `a`, lane-serial.
// IO supplies the agent-scope stores used to hand the result to the master.
// ---------------------------------------------------------------------------
struct LaneCmp {
    unsigned long long *cmp_bytes;
    NLZM_HD uint32_t operator()(const uint8_t *s, const uint8_t *t, uint32_t init, uint32_t cap) const
    {
        uint32_t l = init;
        while (l < cap) {
            const unsigned long long x = load64u(s + l), y = load64u(t + l);
            const unsigned long long d = x ^ y;
            if (d) {
                const uint32_t nb = (uint32_t)__builtin_ctzll(d) >> 3;
                if (l + nb < cap) {
                    *cmp_bytes += (l + nb - init) + 1;
                    return (l + nb) | ((uint32_t)(((x >> (8 * nb)) & 0xFF) < ((y >> (8 * nb)) & 0xFF)) << 31);
                }
                break;
            }
            l += 8;
        }
        *cmp_bytes += cap - init;
        return cap;
    }
};

template <class IO>
struct ResultSink {
    uint32_t *pairs;        // nullptr: do not publish
    uint32_t count, best;
    uint32_t best_d = 0;
    uint32_t d0 = 0, l0 = 0, d1 = 0, l1 = 0, d2 = 0, l2 = 0, d3 = 0, l3 = 0;      // the first four record-setters, kept until the call is over
    const struct BtView *view = nullptr;    // (where pairs beyond the position's reservation go)
    uint32_t ext_idx = 0;
    NLZM_HD void operator()(uint32_t d, uint32_t l);
    // the record: quads 1..3, then quad 0 (pairs beyond the record are in memory before it)
    NLZM_HD void publish(uint32_t *rec, uint32_t tests)
    {
        if (count > 4) IO::drain();
        IO::st_quad(rec + 4, d0, l0, d1, kBtTag);
        IO::st_quad(rec + 8, l1, d2, l2, kBtTag);
        IO::st_quad(rec + 12, d3, l3, ext_idx, kBtTag);
        IO::st_quad(rec, kBtReady | (tests << 9) | count, best_d, count ? best : 0u, 0u);
    }
};

// what a worker lane's call touches (a view of Geom / Globals: the worker role holds these as values of its own)
struct BtView {
    const uint8_t *in;
    uint32_t *heads, *tree, *ready, *pairs;
    uint32_t batch_a0, bt_shift, wmask, tmask;
    uint32_t pstride = kBtMaxPairs;
    uint32_t *ext = nullptr, *ext_cur = nullptr;
    uint32_t ext_cap = 0;
    uint32_t *fail_word = nullptr;          // set to 3 when the extension arena is used up (Globals::abort_word)
};
NLZM_HD BtView bt_view(const Geom &g, const Globals &G)
{
    return BtView{ G.in, G.bt_heads, G.bt_tree, G.bt_ready, G.bt_pairs, G.batch_a0, g.bt_shift, g.wmask, g.bt_tmask,
                   G.bt_pstride, G.bt_ext, G.bt_ext_cur, G.bt_ext_cap, G.abort_word };
}
template <class IO>
NLZM_HD void ResultSink<IO>::operator()(uint32_t d, uint32_t l)
{
    if (!pairs || l <= best) return;    // only record-setters change the table
    best = l; best_d = d;
    // (selects: a branch per slot becomes ONE indexed access to the eight, in scratch)
    const bool s0 = count == 0, s1 = count == 1, s2 = count == 2, s3 = count == 3;
    d0 = s0 ? d : d0; l0 = s0 ? l : l0; d1 = s1 ? d : d1; l1 = s1 ? l : l1;
    d2 = s2 ? d : d2; l2 = s2 ? l : l2; d3 = s3 ? d : d3; l3 = s3 ? l : l3;
    if (count >= 4) {
        uint32_t *q = bt_pair_ptr<IO>(pairs, view->pstride, view->ext, view->ext_cur, view->ext_cap, ext_idx, count);
        if (q) { IO::st_agent(q, d); IO::st_agent(q + 1, l); }       // (else: dropped -- the host sees the arena's cursor beyond its end and makes the stream again)
    }
    count++;
}

// the dry run of an `unc` position with its stores noted down by `st`
template <class IO, class St>
NLZM_HD void worker_bt_dry(const BtView &B, uint32_t a, uint32_t max_len, unsigned long long &n_tests,
                           unsigned long long &cmp_bytes, St &st, uint32_t head = kNone /* the position's BT4 head, if the caller has it */)
{
    LaneCmp cmp{ &cmp_bytes };
    const unsigned long long bi = a - B.batch_a0;
    ResultSink<IO> sink{ B.pairs + bi * (2 * B.pstride), 0, 1 };
    sink.view = &B;
    uint32_t tests = 0;
    const uint32_t h4 = head == kNone ? hash4(load32u(B.in + a)) : head << B.bt_shift;     // (only h4 >> bt_shift is used)
    bt_find_and_update_st(B.heads, B.tree, B.bt_shift, B.wmask, B.tmask, B.in, a, h4, max_len, cmp, sink, tests, st);
    n_tests += tests;
    sink.publish(B.ready + bi * kBtRec, tests);
}

template <class IO, bool kWrite>
NLZM_HD void worker_bt_call(const BtView &B, uint32_t a, uint32_t max_len, bool publish,
                            unsigned long long &n_tests, unsigned long long &cmp_bytes, uint32_t head = kNone)
{
    LaneCmp cmp{ &cmp_bytes };
    const unsigned long long bi = a - B.batch_a0;
    ResultSink<IO> sink{ publish ? B.pairs + bi * (2 * B.pstride) : nullptr, 0, 1 };
    sink.view = &B;
    uint32_t tests = 0;
    const uint32_t h4 = head == kNone ? hash4(load32u(B.in + a)) : head << B.bt_shift;
    bt_find_and_update<kWrite>(B.heads, B.tree, B.bt_shift, B.wmask, B.tmask, B.in, a, h4, max_len, cmp, sink, tests);
    n_tests += tests;
    if (publish) sink.publish(B.ready + bi * kBtRec, tests);
}

}  // namespace nlzm
