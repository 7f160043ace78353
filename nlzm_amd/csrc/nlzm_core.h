// nlzm_core.h -- data layout and the serial ("master") half of the compress path.
//
// Everything here is written once, against a wave policy `W` (lane id, wave
// width, LDS sync, cross-lane reductions, grouped byte compares).  The kernels in
// nlzm_kernels.hip instantiate it with the 64-lane gfx950 policy; tests/host_sim
// instantiates the very same code with a 1-lane policy to check the logic against
// the oracle without a GPU.  Control flow is wave-uniform: every lane carries the
// same scalar state, lanes split only inside the `for (i = W::lane(); ...)` loops.
//
// The serial half is seven roles, one wave each, that hand work to each other through LDS (MasterLds):
//   run_finder      finder block of parse_table: HT2/HT3/RK256, nice decision, worker decisions   :1501-1543
//   run_table       MatchTable as mt_carry holds it: carry / extend / update, published per position :823-852, 1543
//   run_parser      parse_table's node loop: literal edge, node finality, segment end, backtrack, emit :1464-1651
//   run_edge_list   the sampled-length edges of a node, listed                                       :1558-1596
//   run_rep_list    the explicit rep probes of a node, listed (two waves: even / odd positions)      :1598-1628
//   run_edge_apply  relaxes both lists in the reference's order (only writer of nodes >= p+2)
//
// Reference map of the pieces (all NLZM.cpp):
//   Master::finders, pf_fill, fast_run   MatchFinderHT::FindAndUpdate :910-938, MatchFinderRK256::FindAndUpdate :1055-1113
//   bt_find_and_update                   MatchFinderBT::FindAndUpdate :978-1022 (worker lanes, nlzm_kernels.hip)
//   Master::emit_*, put_sym              model_encode_* :1274-1367, 1428-1439, cdf_update :348-382
//   Master::run_chunk_*                  encode_file chunk loop :1782-1886
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
constexpr uint32_t kPf = 64;            // look-ahead depth of the master (positions)
constexpr uint32_t kWinTail = 288;
constexpr uint32_t kCq = 512;           // table-command queue between the finder wave and the table wave
constexpr uint32_t kEr = 64;            // hand-off ring between the table wave and the parser wave (positions)
constexpr uint32_t kErLong = 4;         // of them with a table longer than 63 entries
constexpr uint32_t kRepPf = 64;         // bytes fetched ahead per explicit rep probe; longer matches take the exact path

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
    unsigned long long prof[56];    // cycles per master phase (diagnostic builds: NLZM_PROFILE); 16..21: wait/total cycles per wave
};

constexpr uint32_t kErrFrameOverflow = 1;
constexpr uint32_t kErrTimeout = 2;
constexpr uint32_t kErrInternal = 3;
constexpr uint32_t kErrCapture = 4;

struct FrameMeta {
    uint32_t nsyms, nbits_bytes, num_ops, out_len;
};

// BT4 result of one position, written by a worker lane and read by the master:
//   bt_ready[i * kBtRec] = 0 (not yet) | 0x80000000 | tests<<9 | count, words 1..8 of the record: the first four pairs
//   (one 64-byte record per position: a waiting master reads word and pairs with ONE load instruction)
//   bt_pairs[i * 2*kBtMaxPairs ...] = `count` (distance, length) pairs -- the
// record-setters of the descent (a descent visits candidates by increasing distance,
// so only a longer match changes the table: :835-852).  The stride is the worst case
// (one pair per test, <= 256 tests, :777): 2 KiB per position, sized for HBM, so no
// allocator and no overflow path exists.
constexpr uint32_t kBtMaxPairs = 256;
constexpr uint32_t kBtReady = 0x80000000u;
constexpr uint32_t kFlagCall = 1, kFlagSkip = 2;
constexpr uint32_t kBtRec = 16;         // words per bt_ready record
constexpr uint32_t kBtxPairs = 16;      // longest BT4 result the look-ahead keeps in LDS

struct WorkerCounters {
    unsigned long long bt_calls, bt_tests, cmp_bytes, dry_runs, flag_waits;
    unsigned long long call_cycles, call_tests;     // diagnostics: cycles / tests of the write-mode calls (per-lane clocks)
    unsigned long long lead[6];                     // NLZM_LEAD_DIAG: calls by how far ahead of the master they ended
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
    uint32_t *bt_pairs;         // [(a - batch_a0) * 2 * kBtMaxPairs]
    uint32_t *bt_flag;          // [a - batch_a0] master -> worker: kFlagCall / kFlagSkip
    const uint8_t *unc;         // [a - batch_a0] 1: whether BT4 runs at `a` is the master's call
    const uint32_t *bin_off;    // [chunk - chunk0][nheads + 1]
    const uint32_t *bin_pos;    // [chunk - chunk0][chunk_size] positions grouped by bin, ascending
    uint32_t nheads;            // bins = min(BT4 heads, worker lanes); head h belongs to bin h % bins
    uint32_t *abort_word;       // nonzero: every role leaves its loops
    WorkerCounters *wcnt;
    void *hook_user;            // host simulation only
};

// LDS image of the master workgroup.
struct MasterLds {
    uint32_t node_cost[kParseMax + 2];
    uint32_t node_delta[kParseMax + 1];
    uint32_t node_link[kParseMax + 1];      // from:13 | len:9<<13 | cmd:2<<22 ; from==0x1FFF: none
    uint32_t reps[512 * 4];                 // CarriedState ring (:1460-1467)
    uint32_t mt[512];                       // match table as a ring: entry i at (mt_base+i)&511
    uint16_t cmdlist[kParseMax + 2];        // node indices of the chosen path
    uint16_t cdf[kNumCtx * kCdfStride];
    uint16_t price[kNumCtx * 16];           // log2_lut[freq>>6] per (context, symbol)  (:435-438)
    uint16_t lut[256];                      // log2_lut (:97-124)
    uint16_t seg_len_price[kMatchMax + 8];  // price of the length symbols by length value (:1214-1225), current model
    uint16_t seg_slot_price[4 * 64];        // price of the two distance-slot symbols by (length class, slot) (:1245-1248)
    uint32_t btpairs[2 * kBtMaxPairs];      // worker result being consumed
    uint32_t sq_sym[2 * 8];                 // symbols of the command being emitted: context, symbol (distinct contexts)
    Counters cnt;                           // operation counters of this launch (LDS adds, nothing to wait for)
    // look-ahead: kPf positions are evaluated by the lanes in parallel against the finder
    // tables as they stand, then consumed in order (Master::pf_fill / finders_fast)
    // one 32-word record per slot, fetched with ONE LDS read (lane k takes word k) when the slot is consumed:
    //   0 input bytes (4)   1 HT2 bucket | HT3 bucket << 16   2 HT3 row 0 as read   3 flags
    //   4 number of HT updates | bytes compared << 8          5..10 HT updates: distance, length | open << 31
    //   11 RK hash   12 RK slot as read   13 RK candidate length   14 bt_ready as read   15..22 first 4 BT4 pairs
    //   23..26 summaries of the updates (HT only / HT + BT4)   27, 28 HT2 row and HT3 row 1 as read   29 simple but for BT4
    // flags: bits 0..2 HT candidate valid, 3 RK candidate valid, 4 RK length inexact, 5 unc
    uint32_t pf_rec[2 * 32 * kPf];          // two batches: the table wave reads the records of a run after the finder wave has moved on
    uint32_t pf_btx[2 * kPf * 2 * kBtxPairs];   // BT4 results of 5..kBtxPairs pairs (8 % of text positions), per record buffer
    uint8_t win[kPf + kWinTail];            // input bytes from the first look-ahead position on
    // ---- hand-off run_finder -> run_table -> run_parser ----
    // Wave A (finders) keeps only the table's length and top entry in registers and sends what happens to
    // the table as commands (cq); wave T owns the table ring, applies them and publishes, per position a,
    // slot a % kEr: word 0 = table length | (long slot + 1) << 16, word 1 = the input byte,
    // words 2..63 = table entries 2..63 (longer tables go to er_long whole); the parser and edge waves read it.
    uint32_t cq[kCq * 2];                   // op | arg << 8, value
    uint32_t x_cpos;                        // A: commands < x_cpos are written
    uint32_t x_tpos;                        // T: commands < x_tpos are consumed
    uint32_t er_tab[kEr * 64];
    uint32_t er_long[kErLong * (kMatchMax + 8)];
    uint32_t x_apos;                        // T: positions < x_apos are published
    uint32_t x_bpos;                        // B: positions < x_bpos are parsed (their slots are free again)
    uint32_t x_bseg;                        // B: start of the segment that contains x_bpos
    uint32_t x_bcover;                      // B: positions < x_bcover are known to lie inside the segment that starts at x_bseg
    uint32_t x_long_free;                   // B: long slots given back (monotonic)
    uint32_t x_err;                         // either: error code, both waves leave their loops
    uint32_t x_adone;                       // A, T: finished the launch (count)
    // parser -> edge waves: the node whose match and rep edges are to be relaxed (slot a & 1), and back.
    // One 32-word, 128-byte aligned block per slot, read whole by ONE LDS instruction (lane i takes word i: 32
    // distinct banks, a single pass, so a reader that sees the count sees the request it covers):
    //   0      positions < this are posted in this slot (0xFFFFFFFF: leave)
    //   2, 3   price words of the command context (symbols 0|1, 2|3) for this segment
    //   4..19  request: a, p, cost_p, rep set (4), max_len, (stamp), hand-off header, q, rep cap
    //   21, 22 (block 0 only) rep-list counts of the even / odd positions
    //   23     positions < this have a guessed rep set in words 24..28 (position, rep set): the node is not final yet
    alignas(128) uint32_t post[2][32];
    uint32_t sq_res[2];                     // per slot: end_p after the node's edges
    uint32_t x_sdone;                       // apply wave: the edges of positions < x_sdone are relaxed
    // edge-list wave -> apply wave: the sampled-length edges of position a in slot a & 3, listed AHEAD of the node's
    // finality (they depend on the position's table, the segment's prices and its place in the segment only).
    // lane k: cost of the edge as dict edge and as rep edge (without the node's cost), distance, length | valid << 12
    uint32_t ea[4 * 64 * 4];
    uint32_t ea_tag[4 * 2];                 // per slot: a + 1, segment sequence number (written after the entries)
    alignas(32) uint32_t seginfo[8];        // parser -> edge-list wave: sequence number (written last), seg_a, max_parse
    uint32_t eb[4 * 16];                    // explicit rep probes, words 2k, 2k+1: length | relaxable << 30 | valid << 31, edge price; 14, 15: listed / listed for the guess at least (position + 1)
                                            // through it; word 8: rep indices met by a sampled edge (:1573-1584)
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
    uint32_t *rec;          // the position's bt_ready record (first four pairs go there too)
    uint32_t count, best;
    uint32_t best_d = 0;
    NLZM_HD void operator()(uint32_t d, uint32_t l)
    {
        if (!pairs || l <= best) return;    // only record-setters change the table
        best = l; best_d = d;
        IO::st_agent(pairs + 2 * count, d);
        IO::st_agent(pairs + 2 * count + 1, l);
        if (count < 4) { IO::st_agent(rec + 1 + 2 * count, d); IO::st_agent(rec + 2 + 2 * count, l); }
        count++;
    }
};

// the dry run of an `unc` position with its stores noted down by `st`
template <class IO, class St>
NLZM_HD void worker_bt_dry(const Geom &g, const Globals &G, uint32_t a, uint32_t max_len, unsigned long long &n_tests,
                           unsigned long long &cmp_bytes, St &st)
{
    LaneCmp cmp{ &cmp_bytes };
    const unsigned long long bi = a - G.batch_a0;
    ResultSink<IO> sink{ G.bt_pairs + bi * (2 * kBtMaxPairs), G.bt_ready + bi * kBtRec, 0, 1 };
    uint32_t tests = 0;
    const uint32_t h4 = hash4(load32u(G.in + a));
    bt_find_and_update_st(G.bt_heads, G.bt_tree, g.bt_shift, g.wmask, g.bt_tmask, G.in, a, h4, max_len, cmp, sink, tests, st);
    n_tests += tests;
    IO::st_agent(G.bt_ready + bi * kBtRec + 9, sink.best_d);       // the longest record-setter (the last one), for the finder stage
    IO::st_agent(G.bt_ready + bi * kBtRec + 10, sink.count ? sink.best : 0u);
    IO::drain();                            // every pair has been written through before the ready word
    IO::st_agent(G.bt_ready + bi * kBtRec, kBtReady | (tests << 9) | sink.count);
}

template <class IO, bool kWrite>
NLZM_HD void worker_bt_call(const Geom &g, const Globals &G, uint32_t a, uint32_t max_len, bool publish,
                            unsigned long long &n_tests, unsigned long long &cmp_bytes)
{
    LaneCmp cmp{ &cmp_bytes };
    const unsigned long long bi = a - G.batch_a0;
    ResultSink<IO> sink{ publish ? G.bt_pairs + bi * (2 * kBtMaxPairs) : nullptr, G.bt_ready + bi * kBtRec, 0, 1 };
    uint32_t tests = 0;
    const uint32_t h4 = hash4(load32u(G.in + a));
    bt_find_and_update<kWrite>(G.bt_heads, G.bt_tree, g.bt_shift, g.wmask, g.bt_tmask, G.in, a, h4, max_len, cmp, sink, tests);
    n_tests += tests;
    if (publish) {
        IO::st_agent(G.bt_ready + bi * kBtRec + 9, sink.best_d);
        IO::st_agent(G.bt_ready + bi * kBtRec + 10, sink.count ? sink.best : 0u);
        IO::drain();                        // every pair has been written through before the ready word
        IO::st_agent(G.bt_ready + bi * kBtRec, kBtReady | (tests << 9) | sink.count);
    }
}

// ---------------------------------------------------------------------------
// The master: walks positions in order, owns HT2/HT3/RK256 state, the match
// table chain, the parse, the model and the symbol output.
// ---------------------------------------------------------------------------
#ifdef NLZM_SIM_COUNT
static unsigned long long g_dbg[8];
#endif
template <class W>
struct Master {
    Geom g;
    Globals G;

    // match table ring + RK scalars + window base (wave-uniform registers)
    uint32_t mt_base, mt_max;
    uint32_t top_d;                 // distance stored at mt(mt_max)
    bool top_open;                  // false: that entry is known not to extend (mismatch at its end)
    bool rk_open;                   // the carried RK match ran into its length cap
    uint32_t pf_base, pf_n;         // look-ahead window [pf_base, pf_base + pf_n)
    uint32_t pf_buf, pf_mark0, pf_mark1;    // record buffer in use; command count when buffer 0 / 1 was last left
    typename W::PfLane pfl;         // per slot (= per lane): HT buckets, RK slot, stale bits (1: HT rows, 4: RK slot)
    uint32_t a_long, b_long;        // long hand-off slots taken (table wave) / given back (parser wave)
    uint32_t bpos_seen;             // table wave: x_bpos as last read
    uint32_t th_a, th_hdr, th_lit;  // parser wave: hand-off header and input byte of position th_a (read ahead)
    uint32_t seg_seq;               // parser wave: segments started in this launch
    bool seg_tab_dirty;             // parser wave: a length/distance context changed since the price tables were built
    uint32_t cq_n, cq_seen;         // finder wave: commands written / consumed count last seen
    uint32_t seg_s, seg_cut;        // finder wave, inside a nice region: segment start and its forced cut
    bool prev_nice;
    // per-chunk operation counts kept in scalar registers, flushed to the LDS counters per chunk
    uint32_t n_pos, n_nice, n_unc, n_ht, n_rkp, n_rki, n_cmp, n_sht, n_srk, n_sbt;
    NLZM_HD void counts_zero() { n_pos = n_nice = n_unc = n_ht = n_rkp = n_rki = n_cmp = n_sht = n_srk = n_sbt = 0; }
    NLZM_HD void counts_flush()
    {
        Counters &c = W::lds()->cnt;
        W::cnt_add(&c.positions, n_pos); W::cnt_add(&c.nice_positions, n_nice); W::cnt_add(&c.uncertain_positions, n_unc);
        W::cnt_add(&c.ht_rows, 3ull * n_ht); W::cnt_add(&c.rk_probes, n_rkp); W::cnt_add(&c.rk_inserts, n_rki);
        W::cnt_add(&c.cmp_bytes, n_cmp);
        W::cnt_add(&c.stale_ht, n_sht); W::cnt_add(&c.stale_rk, n_srk); W::cnt_add(&c.bt_slow, n_sbt);
        counts_zero();
    }
    uint32_t rk_from, rk_to, rk_len, rk_end;
    unsigned long long base;        // absolute offset of rebased 0
    uint32_t la_end;                // rebased end of the chunk's lookahead
    uint32_t rep[4];                // live model rep set

    // frame writer (CodeFrame, :490-513)
    uint32_t *fsyms; uint8_t *fbits;
    uint32_t nsyms, nbits, word, word_bits, num_ops, nq;

    uint32_t err, err_info0;
    unsigned long long wait_cyc, role_t0;   // cycles spent waiting for another wave / role start (diagnostics)
    unsigned long long wait_cq = 0, wait_guard = 0;     // finder wave: of them for queue space / for the record buffer
    unsigned long long wait_rep = 0;                    // apply wave: of them for the rep list
    unsigned long long cmp_acc = 0;                     // apply wave: bytes compared by the rep probes it took (Persist::prof[46])
#ifdef NLZM_PROFILE
    unsigned long long prof[16];
    unsigned long long prof_t, lat_sum = 0, lat_sum2 = 0;
    NLZM_HD void prof_start() { prof_t = W::tick(); }
    NLZM_HD void prof_mark(int k) { const unsigned long long t = W::tick(); prof[k] += t - prof_t; prof_t = t; }
#else
    NLZM_HD void prof_start() {}
    NLZM_HD void prof_mark(int) {}
#endif

    // ---- LDS accessors ----------------------------------------------------
    NLZM_HD uint32_t &mt(uint32_t i) { return W::lds()->mt[(mt_base + i) & 511]; }

    NLZM_HD uint32_t price(uint32_t ctx, uint32_t y) const { return W::lds()->price[ctx * 16 + y]; }

    // ---- byte compare, one pair, all lanes (MatchLengthSigned, :854-877) ---
    // returns length | (s-byte < t-byte at the mismatch) << 31; kCount adds the bytes looked at to cmp_bytes
    template <bool kCount>
    static NLZM_HD uint32_t wave_cmp(const uint8_t *s, const uint8_t *t, uint32_t init, uint32_t cap)
    {
        uint32_t off = init, res = cap;
        uint32_t lt = 0;
        bool hit = false;
        while (off < cap) {
            const uint32_t my = off + W::lane() * 8;
            uint32_t m = kNone, mylt = 0;
            if (my < cap) {
                const unsigned long long x = load64u(s + my), y = load64u(t + my);
                const unsigned long long d = x ^ y;
                if (d) {
                    const uint32_t nb = (uint32_t)__builtin_ctzll(d) >> 3;
                    if (my + nb < cap) {
                        m = my + nb;
                        mylt = ((x >> (8 * nb)) & 0xFF) < ((y >> (8 * nb)) & 0xFF);
                    }
                }
            }
            const unsigned long long bal = W::mask64([=](uint32_t) { return m != kNone; });
            if (bal) {                                  // lanes look at ascending offsets: the lowest lane wins
                const uint32_t fl = (uint32_t)__builtin_ctzll(bal);
                res = W::pick(m, fl); lt = W::pick(mylt, fl); hit = true;
                break;
            }
            off += W::width() * 8;
        }
        if (kCount) W::cnt_add(&W::lds()->cnt.cmp_bytes, (res - init) + (hit ? 1u : 0u));
        return res | (lt << 31);
    }
    struct WaveCmp {
        NLZM_HD uint32_t operator()(const uint8_t *s, const uint8_t *t, uint32_t init, uint32_t cap) const
        {
            return wave_cmp<true>(s, t, init, cap);
        }
    };

    // ---- table commands (finder wave) ---------------------------------------------
    // The finder wave needs only the table's length and its top entry (for the nice decision :1514 and
    // the extension :1503-1512); every change is forwarded to the table wave in program order.
    static constexpr uint32_t kOpCarry = 1, kOpExt = 2, kOpUpd = 3, kOpEnd = 4, kOpRun = 5, kOpSet = 6;
    static constexpr uint32_t kSetHt = 1, kSetBt = 2, kSetBtx = 4;     // kOpSet: which pairs of the slot's record
    NLZM_HD void cq_push(uint32_t op, uint32_t arg, uint32_t val)
    {
        if (cq_n - cq_seen >= kCq) {                    // queue full as far as we know: look again / wait
            const unsigned long long t0 = W::clock();
            const unsigned long long c0 = W::tick();
            uint32_t spins = 0;
            while (cq_n - (cq_seen = W::xw_load(&W::lds()->x_tpos)) >= kCq) {
                if (W::xw_load(&W::lds()->x_err)) { err = kErrInternal + 100; return; }
                if ((++spins & 1023u) == 0 && W::clock() - t0 > W::timeout_ticks()) { fail(kErrTimeout, cq_n); return; }
                W::xw_pause();
            }
            { const unsigned long long w = W::tick() - c0; wait_cyc += w; wait_cq += w; }
        }
        uint32_t *c = W::lds()->cq + (cq_n & (kCq - 1)) * 2;
        c[0] = op | (arg << 8); c[1] = val;
        cq_n++;
    }
    NLZM_HD void cq_flush() { W::xw_store(&W::lds()->x_cpos, cq_n); }

    // MatchTable::Update (:835-852).  `open`: the match ran into its length cap, so the carried entry may
    // extend at the next position (:1503-1512).
    NLZM_HD void mt_update(uint32_t d, uint32_t len, bool open)
    {
        cq_push(kOpUpd, len, d);
        if (len > mt_max) { mt_max = len; top_d = d; top_open = open; }
        else if (len == mt_max) { top_d = umin(top_d, d); top_open = top_open || open; }
    }
    // The HT2/HT3 (and, when they were there in time, BT4) updates of one position; pairs k < nh sit at record
    // words 5.., the next nb at words 15..; (sl, sd) is their summary as the look-ahead lane computed it
    // (longest length | open << 31, smallest distance among the longest).
    NLZM_HD void mt_apply_set(uint32_t s, const typename W::Rec &rec, uint32_t nh, uint32_t nb, uint32_t sl, uint32_t sd)
    {
        const uint32_t lm = sl & 0x1FFu;
        (void)rec;
        cq_push(kOpSet, s | (pf_buf << 6) | (((nh ? kSetHt : 0u) | (nb > 4 ? kSetBtx : (nb ? kSetBt : 0u))) << 7), 0);   // the table wave reads the record
        if (lm > mt_max) { mt_max = lm; top_d = sd; top_open = (sl >> 31) != 0; }
        else if (lm == mt_max) { top_d = umin(top_d, sd); top_open = top_open || (sl >> 31) != 0; }
    }
    // the table's length / top entry after an update that the table wave applies from a record
    NLZM_HD void mt_note(uint32_t d, uint32_t len, bool open)
    {
        if (len > mt_max) { mt_max = len; top_d = d; top_open = open; }
        else if (len == mt_max) { top_d = umin(top_d, d); top_open = top_open || open; }
    }
    // carry by one position (CarryFrom with shift 1, :823-833)
    NLZM_HD void mt_carry()
    {
        cq_push(kOpCarry, 0, 0);
        if (mt_max <= 1) { mt_max = 0; top_open = false; }
        else mt_max -= 1;
    }

    // ---- the table itself (table wave) ----------------------------------------------
    // Invariant: every ring entry above t_max holds kNone, so the element-wise min-merge of Update is ONE
    // predicated LDS atomic min per lane (no read, nothing to wait for) whether the entry existed or not.
    NLZM_HD void t_update(uint32_t d, uint32_t len)
    {
        for (uint32_t i = W::lane(); i <= len; i += W::width()) W::lds_min(&mt(i), d);
        mt_max = umax(mt_max, len);
    }
    // every update of a slot (:835-852 is an element-wise min: the order does not matter) in ONE pass over the table:
    // lane e takes entry e, the smallest distance among the record's pairs that reach it (nh HT pairs, nb BT4 pairs)
    NLZM_HD void t_apply_set(const typename W::Rec &r, uint32_t nh, uint32_t nb, uint32_t maxl)
    {
        if (!(nh + nb)) return;
        // (uniform loops and selects only: a min with 0xFFFFFFFF leaves an entry as it is, entries beyond every pair's
        // length included, so no lane needs a branch)
        for (uint32_t base = 0; base <= maxl; base += W::width()) {
            const uint32_t e = base + W::lane();
            uint32_t m = kNone;
#pragma unroll
            for (int j = 0; j < 3; j++)
                if ((uint32_t)j < nh) { const uint32_t l = W::rec_get(r, 6 + 2 * j) & 0x1FFu, d = W::rec_get(r, 5 + 2 * j); m = umin(m, e <= l ? d : kNone); }
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((uint32_t)j < nb) { const uint32_t l = W::rec_get(r, 16 + 2 * j) & 0x1FFu, d = W::rec_get(r, 15 + 2 * j); m = umin(m, e <= l ? d : kNone); }
            W::lds_min(&mt(e), m);
        }
        mt_max = umax(mt_max, maxl);
    }
    // a BT4 list of 5..kBtxPairs record-setters (look-ahead slot s of record buffer buf): lengths grow along it
    NLZM_HD void t_apply_btx(uint32_t buf, uint32_t s, uint32_t cnt)
    {
        const typename W::Rec x = W::rec_load(W::lds()->pf_btx + (buf * kPf + s) * (2 * kBtxPairs));
        const uint32_t maxl = W::rec_get(x, 2 * cnt - 1);
        for (uint32_t base = 0; base <= maxl; base += W::width()) {
            const uint32_t e = base + W::lane();
            uint32_t m = kNone;
            for (uint32_t j = 0; j < cnt; j++) { const uint32_t l = W::rec_get(x, 2 * j + 1), d = W::rec_get(x, 2 * j); m = umin(m, e <= l ? d : kNone); }
            W::lds_min(&mt(e), m);
        }
        mt_max = umax(mt_max, maxl);
    }
    NLZM_HD void t_carry()
    {
        if (mt_max <= 1) {
            for (uint32_t i = W::lane(); i < 2; i += W::width()) mt(i) = kNone;
            mt_max = 0;
        } else {
            mt(0) = kNone;                              // becomes relative index 511 after the shift
            mt_max -= 1; mt_base = (mt_base + 1) & 511;
        }
    }
    NLZM_HD void t_extend(uint32_t d, uint32_t nl)
    {
        for (uint32_t i = mt_max + 1 + W::lane(); i <= nl; i += W::width()) mt(i) = d;
        mt_max = nl;
    }

    struct MtSink {
        Master *m;
        uint32_t max_len;
        NLZM_HD void operator()(uint32_t d, uint32_t l) const { m->mt_update(d, l, l >= max_len); }
    };

    // ---- model prices ------------------------------------------------------
    NLZM_HD uint32_t price_len(uint32_t lv) const                  // :1214-1225
    {
        uint32_t cst = price(kCtxLenDirect, umin(lv, 7));
        if (lv >= 7) {
            const uint32_t e = lv - 7;
            cst += price(kCtxLenExtHi, e >> 4) + price(kCtxLenExtLo + (e >> 4), e & 15);
        }
        return cst;
    }
    NLZM_HD uint32_t price_match(uint32_t d, uint32_t len) const   // :1208-1251
    {
        const uint32_t lv = len - match_min(d), lc = umin(lv, 3);
        uint32_t nx, ex;
        const uint32_t slot = dist_slot(d - 1, nx, ex);
        return price(kCtxCmd, 1) + price_len(lv) + (nx << 5) + price(kCtxSlotHi + lc, slot >> 3) +
               price(kCtxSlotLo + lc * 8 + (slot >> 3), slot & 7);
    }
    NLZM_HD uint32_t price_rep(uint32_t d, uint32_t len) const     // :1253-1272
    {
        return price(kCtxCmd, 2) + price_len(len - match_min(d)) + (2u << 5);
    }
    // per-model price tables for the match edges: rebuilt (by all lanes) only after an emit touched a length or
    // distance context; literal-only segments leave them valid
    NLZM_HD void seg_tables()
    {
        if (!seg_tab_dirty) return;
        W::sync();
        for (uint32_t lv = W::lane(); lv <= kMatchMax; lv += W::width()) W::lds()->seg_len_price[lv] = (uint16_t)price_len(lv);
        for (uint32_t i = W::lane(); i < 4 * 64; i += W::width()) {
            const uint32_t lc = i >> 6, slot = i & 63;
            W::lds()->seg_slot_price[i] = slot < 56 ? (uint16_t)(price(kCtxSlotHi + lc, slot >> 3) + price(kCtxSlotLo + lc * 8 + (slot >> 3), slot & 7)) : 0;
        }
        W::sync();
        seg_tab_dirty = false;
    }
    static NLZM_HD typename W::Rec nrec_none() { return W::rec_load_fn([](uint32_t) { return 0u; }); }
    NLZM_HD uint32_t price_literal(uint32_t y) const               // :1418-1426
    {
        uint32_t c0 = price(kCtxCmd, 0), c1 = price(kCtxLitHi, y >> 4), c2 = price(kCtxLitLo + (y >> 4), y & 15);
        W::join3(c0, c1, c2);                                       // three loads in flight, one wait
        return W::uni(c0 + c1 + c2);
    }

    // ---- symbol output (WriteRange/WriteBits + cdf_update) -------------------
    // The symbols of ONE command go through their nibble CDFs together: their contexts are distinct (command, length
    // direct / ext hi / ext lo[hi], slot hi[lc] / slot lo[lc][hi], literal hi / lo[hi]), so the (start, freq) snapshots
    // (WriteCDF :559-572, :1278-1279), the adaptations (cdf_update :348-382) and the price rows (:435-438) are
    // independent; W::cdf_multi takes four symbols per pass, sixteen lanes each.
    NLZM_HD void put_sym(uint32_t ctx, uint32_t y)
    {
        W::lds()->sq_sym[2 * nq] = ctx; W::lds()->sq_sym[2 * nq + 1] = y;
        nq++;
    }
    NLZM_HD void flush_syms()
    {
        W::sync();
        W::cdf_multi(W::lds()->cdf, W::lds()->price, W::lds()->lut, W::lds()->sq_sym, nq, fsyms + nsyms);
        nsyms += nq; num_ops += nq; nq = 0;
        W::sync();
    }
    NLZM_HD void put_bits(uint32_t v, uint32_t nb)                  // :574-588
    {
        num_ops++;
        word |= v << (32 - word_bits - nb);
        word_bits += nb;
        while (word_bits >= 8) {
            fbits[nbits] = (uint8_t)(word >> 24);
            nbits++;
            word <<= 8;
            word_bits -= 8;
        }
    }
    NLZM_HD uint32_t emit_len(uint32_t lv)                          // :1281-1297
    {
        seg_tab_dirty = true;
        put_sym(kCtxLenDirect, umin(lv, 7));
        if (lv >= 7) {
            const uint32_t e = lv - 7;
            // both WriteCDFs precede both updates in the reference; contexts differ, so equal
            put_sym(kCtxLenExtHi, e >> 4);
            put_sym(kCtxLenExtLo + (e >> 4), e & 15);
        }
        return umin(lv, 3);
    }
    NLZM_HD void emit_literal(uint32_t y)                           // :1428-1439
    {
        put_sym(kCtxCmd, 0);
        put_sym(kCtxLitHi, y >> 4);
        put_sym(kCtxLitLo + (y >> 4), y & 15);
        flush_syms();
        W::cnt_add(&W::lds()->cnt.n_literal, 1);
    }
    NLZM_HD void emit_match(uint32_t d, uint32_t len)               // :1274-1342
    {
        put_sym(kCtxCmd, 1);
        const uint32_t lc = emit_len(len - match_min(d));
        uint32_t nx, ex;
        const uint32_t slot = dist_slot(d - 1, nx, ex);
        put_sym(kCtxSlotHi + lc, slot >> 3);
        put_sym(kCtxSlotLo + lc * 8 + (slot >> 3), slot & 7);
        flush_syms();
        if (d - 1 >= 4) {
            if (nx < 4) put_bits(ex, nx);
            else { if (nx > 4) put_bits(ex >> 4, nx - 4); put_bits(ex & 15, 4); }
        }
        rep_add(rep, d);                                            // :1819
        W::cnt_add(&W::lds()->cnt.n_dict, 1);
    }
    NLZM_HD void emit_rep(uint32_t idx, uint32_t len)               // :1344-1367
    {
        put_sym(kCtxCmd, 2);
        emit_len(len - match_min(idx == 0 ? rep[0] : (idx == 1 ? rep[1] : (idx == 2 ? rep[2] : rep[3]))));
        flush_syms();
        put_bits(idx, 2);
        W::cnt_add(&W::lds()->cnt.n_rep, 1);                                                  // rep4.Add(present delta) is a no-op (:1834)
    }

    // ---- parse graph helpers -------------------------------------------------
    NLZM_HD static uint32_t pack_link(uint32_t from, uint32_t len, uint32_t cmd) { return from | (len << 13) | (cmd << 22); }

    NLZM_HD void open_nodes(uint32_t &end_p, uint32_t upto)         // :1550-1554
    {
        if (upto > end_p) {
            for (uint32_t e = end_p + 1 + W::lane(); e <= upto; e += W::width()) {
                W::lds()->node_cost[e] = kNone;
                W::lds()->node_link[e] = 0x1FFF;
            }
            end_p = upto;
            W::sync();
        }
    }

    // relax edge p -> np by one lane (or uniformly by all lanes with equal arguments)
    NLZM_HD void relax(uint32_t p, uint32_t np, uint32_t cost_p, uint32_t cst, uint32_t cmd, uint32_t len,
                       uint32_t store_delta, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t add_delta)
    {
        if (W::lds()->node_cost[np] > cost_p + cst) {                      // strict: first candidate wins ties
            W::lds()->node_cost[np] = cost_p + cst;
            W::lds()->node_delta[np] = store_delta;
            W::lds()->node_link[np] = pack_link(p, len, cmd);
            const bool has = r0 == add_delta || r1 == add_delta || r2 == add_delta || r3 == add_delta;   // RepModel::Add (:1160-1171)
            uint32_t *dst = W::lds()->reps + (np & 511) * 4;
            dst[0] = has ? r0 : add_delta; dst[1] = has ? r1 : r0; dst[2] = has ? r2 : r1; dst[3] = has ? r3 : r2;
        }
    }

    // ---- look-ahead ------------------------------------------------------------
    // Lane j evaluates position pf_base + j against the finder tables as they stand
    // now: HT2/HT3 rows and the RK256 slot are read, their candidates compared (lane-
    // serial, 8 bytes at a time), the BT4 result of the worker lanes fetched if it is
    // there already.  Positions are consumed in order afterwards; a table store made
    // by an earlier position marks the later slots that read the same row `stale`,
    // and those take the direct path again.
    NLZM_HD void pf_fill(uint32_t a_start, uint32_t pos_end_abs, uint32_t la_end_abs)
    {
        W::sync_global();                       // every table store so far has landed
        // the other record buffer: the table wave must be done with the runs of the batch that used it
        if (pf_buf) pf_mark1 = cq_n; else pf_mark0 = cq_n;
        pf_buf ^= 1u;
        {
            const uint32_t need = pf_buf ? pf_mark1 : pf_mark0;
            if ((int32_t)(W::xw_load(&W::lds()->x_tpos) - need) < 0) {
                cq_flush();
                const unsigned long long t0 = W::clock(), c0 = W::tick();
                uint32_t spins = 0;
                while ((int32_t)(W::xw_load(&W::lds()->x_tpos) - need) < 0) {
                    if (W::xw_load(&W::lds()->x_err)) { err = kErrInternal + 100; return; }
                    if ((++spins & 1023u) == 0 && W::clock() - t0 > W::timeout_ticks()) { fail(kErrTimeout, need); return; }
                    W::xw_pause();
                }
                { const unsigned long long w = W::tick() - c0; wait_cyc += w; wait_guard += w; }
            }
        }
#ifdef NLZM_LEAD_DIAG
        if (W::lane() == 0) W::st_agent((uint32_t *)&G.persist->prof[31], a_start);
#endif
        pf_base = a_start;
        pf_n = umin(kPf, pos_end_abs - a_start);
        for (uint32_t i = W::lane(); i < kPf + kWinTail; i += W::width())
            W::lds()->win[i] = (unsigned long long)a_start + i < g.n ? G.in[a_start + i] : 0;
        for (uint32_t j = W::lane(); j < pf_n; j += W::width()) {
            const uint32_t x = a_start + j, q = (uint32_t)(x - base), avail = la_end_abs - x;
            const uint8_t *cur = G.in + x;
            uint32_t *rec = W::lds()->pf_rec + pf_buf * (32 * kPf) + 32 * j;
            uint32_t flags = 0, lens = 0, v4 = 0, idx = 0xFFFFFFFFu;
            uint32_t row[3] = { 0, 0, 0 };
            unsigned long long dummy = 0;
            LaneCmp lcmp{ &dummy };
            // Loads in rounds of independent requests (a dependent HBM access is ~2,000 cycles):
            // round 1: what the position alone addresses -- its bytes, its RK256 hash, `unc`, the BT4 result word
            const bool h4 = avail >= 4, h256 = avail >= 256;
            const unsigned long long bi = x - G.batch_a0;
            if (h4) v4 = load32u(cur);
            uint32_t rkh = h256 ? G.rkhash[x] : 0u, rkv = 0, rkl = 0, ready = 0;
            if (G.workers) {
                if (G.unc[bi]) flags |= 32u;
#ifdef NLZM_SIM_LATE
                if ((x * 2654435761u) >> 29)            // host simulation: every 8th result is "not in yet" at this point
#endif
                W::wait_hook(G.hook_user, x);
                ready = W::ld_agent(G.bt_ready + bi * kBtRec);
            }
            // round 2: the HT rows, the RK slot, the pairs of a BT4 result that is in
            const uint32_t h2 = hash4(v4 & 0xFFFFu), h3 = hash4(v4 & 0xFFFFFFu);
            const uint32_t i2 = h2 >> 20, i3 = h3 >> g.ht3_shift;
            if (h4) { idx = i2 | (i3 << 16); row[0] = G.ht2[i2]; row[1] = G.ht3[i3]; row[2] = G.ht3[i3 + 1]; }
            if (h256) rkv = G.rk_table[rkh >> g.rk_shift];
            uint32_t nb = 0, bd[4] = { 0, 0, 0, 0 }, bl[4] = { 0, 0, 0, 0 };
            if ((ready & kBtReady) && h4) {
                nb = umin(ready & 0x1FFu, 4u);
                const uint32_t *pairs = G.bt_pairs + bi * (2 * kBtMaxPairs);
#pragma unroll
                for (int k = 0; k < 4; k++) if ((uint32_t)k < nb) { bd[k] = W::ld_agent(pairs + 2 * k); bl[k] = W::ld_agent(pairs + 2 * k + 1); }
                // a longer list (every record-setter of the descent, :996-998) comes along whole, into LDS
                const uint32_t cnt = ready & 0x1FFu;
                if (cnt > 4 && cnt <= kBtxPairs) {
                    uint32_t *bx = W::lds()->pf_btx + (pf_buf * kPf + j) * (2 * kBtxPairs);
                    for (uint32_t k = 0; k < 2 * cnt; k++) bx[k] = W::ld_agent(pairs + k);
                }
            }
            // round 3: the first eight bytes of every candidate (most compares end there); longer ones go on alone
            const uint32_t tag2 = h2 & g.tag_mask, tag3 = h3 & g.tag_mask;
            uint32_t csp[4] = { 0, 0, 0, 0 };           // q - sp of the valid candidates: HT2, HT3 row 0, HT3 row 1, RK
            unsigned long long c8[4] = { 0, 0, 0, 0 }, own8 = 0;
            if (h4) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const uint32_t sp = row[k] & g.wmask;
                    if ((row[k] >> g.wbits) == (k ? tag3 : tag2) && sp < q && q - sp <= g.wmask) { flags |= 1u << k; csp[k] = q - sp; }
                }
            }
            if (h256) {
                const uint32_t sp = rkv & g.wmask;
                if ((rkv >> g.wbits) == (rkh & g.tag_mask) && sp < q && q - sp <= g.wmask) { flags |= 8u; csp[3] = q - sp; }
            }
            if (flags & 15u) own8 = load64u(cur);
#pragma unroll
            for (int k = 0; k < 4; k++) if ((flags >> k) & 1u) c8[k] = load64u(cur - csp[k]);
            // MatchLengthSigned (:854-877) with the first eight bytes in hand
            auto cmp8 = [&](int k, uint32_t cap) -> uint32_t {
                const unsigned long long d = c8[k] ^ own8;
                if (d) {
                    const uint32_t nbm = (uint32_t)__builtin_ctzll(d) >> 3;
                    if (nbm < cap) return nbm;
                    return cap;
                }
                if (cap <= 8) return cap;
                return lcmp(cur - csp[k], cur, 8, cap) & 0x7FFFFFFFu;
            };
            if (h4) {
                const uint32_t max_len = umin(avail, kMatchMax);
                // MatchFinderHT::FindAndUpdate (:917-933) for HT2 then HT3, as a list of table updates
                uint32_t np = 0, cmpb = 0, best = 1;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    if (k == 1) best = 1;                           // HT3 is its own call
                    if ((flags >> k) & 1u) {
                        if (best < max_len) {
                            const uint32_t l = cmp8(k, max_len) & 0x1FFu, d = csp[k];
                            lens |= l << (9 * k);
                            cmpb += l + (l < max_len);
                            if (l > best && l >= match_min(d)) {
                                rec[5 + 2 * np] = d;
                                rec[6 + 2 * np] = l | ((uint32_t)(l >= max_len) << 31);
                                np++; best = l;
                            }
                        }
                    }
                }
                rec[4] = np | (cmpb << 8);
            } else rec[4] = 0;
            if (flags & 8u) {
                const uint32_t cap = avail & 0xFFFFu, lim = umin(cap, kMatchMax + 8);
                rkl = cmp8(3, lim);
                if (rkl == lim && lim < cap) flags |= 16u;          // longer than we looked: exact length on demand
            }
            // summary of a set of table updates: longest length | open << 31, smallest distance among the longest
            uint32_t sl = 0, sd = kNone;
            auto summarise = [&](uint32_t d, uint32_t lo) {
                const uint32_t l = lo & 0x1FFu;
                if (l > (sl & 0x1FFu)) { sl = lo; sd = d; }
                else if (l == (sl & 0x1FFu)) { sl |= lo & 0x80000000u; sd = umin(sd, d); }
            };
            if (h4) for (uint32_t k = 0; k < (rec[4] & 0xFFu); k++) summarise(rec[5 + 2 * k], rec[6 + 2 * k]);
            rec[23] = sl; rec[24] = sd;
            {
                const uint32_t max_len = umin(avail, kMatchMax);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if ((uint32_t)k >= nb) continue;
                    rec[15 + 2 * k] = bd[k]; rec[16 + 2 * k] = bl[k] | ((uint32_t)(bl[k] >= max_len) << 31);
                    summarise(bd[k], rec[16 + 2 * k]);
                }
                // (a longer list: the rest of it from LDS, where round 2 put it)
                const uint32_t cnt = ready & 0x1FFu;
                if ((ready & kBtReady) && h4 && cnt > 4 && cnt <= kBtxPairs) {
                    const uint32_t *bx = W::lds()->pf_btx + (pf_buf * kPf + j) * (2 * kBtxPairs);
                    for (uint32_t k = 4; k < cnt; k++) summarise(bx[2 * k], bx[2 * k + 1] | ((uint32_t)(bx[2 * k + 1] >= max_len) << 31));
                }
            }
            rec[25] = sl; rec[26] = sd;
            rec[0] = v4; rec[1] = idx; rec[2] = row[1]; rec[3] = flags; rec[27] = row[0]; rec[28] = row[2];
            rec[11] = rkh; rec[12] = rkv; rec[13] = rkl; rec[14] = ready;
            // a slot is `simple` when the look-ahead has settled everything about it: the BT4 call happens (not
            // `unc`), its result is here (<= 4 pairs), no RK candidate, no RK insert falls on it, none of its table
            // updates can extend at the next position.  Whether an earlier slot rewrites its HT rows is added below.
            const bool pre = G.workers && avail >= 256 && !(flags & (32u | 8u)) && (q & 255u) != 0;
            rec[29] = pre ? 1u : 0u;                                // (for bt_refresh)
            const bool simple = pre && (ready & kBtReady) && (ready & 0x1FFu) <= kBtxPairs && !(sl >> 31);
            W::pfl_set(pfl, j, idx, rkh >> g.rk_shift, v4, row[1], sl, sd, rec[4] >> 8, simple);
        }
        W::pfl_conflicts(pfl, pf_n);
        W::sync();
    }

    // a store to HT2 row i2 / HT3 rows i3, i3+1 (bucket b reads rows b and b+1, :912) invalidates
    // what later look-ahead slots read from them
    NLZM_HD void pf_mark_rk(uint32_t s, uint32_t slot) { W::pfl_mark_rk(pfl, s, pf_n, slot); }

    // ---- finders for one position (:1501-1543) --------------------------------
    // q: rebased position, a: absolute position, p: parse-relative index.
    // rp/rep_len: the node's rep set and (output) explicit rep-probe lengths.
    NLZM_HD void finders(uint32_t q, uint32_t a, uint32_t pos_end_abs)
    {
        const uint8_t *cur = G.in + a;
        const uint32_t avail = la_end - q;
        prof_mark(0);
        const uint32_t s = a - pf_base;
        typename W::Rec rec = W::rec_load(W::lds()->pf_rec + pf_buf * (32 * kPf) + 32 * s);
        const uint32_t pflags = W::rec_get(rec, 3), pstale = W::pfl_stale(pfl, s);

        // carry by one (:1501-1502, CarryFrom :823-833)
        mt_carry();
        // extend the longest entry (:1503-1512); skipped when that entry is known to end in a mismatch
        if (mt_max > 0 && top_open) {
            const uint32_t d = top_d;
            if (q >= d) {
                const uint32_t cap = umin(kMatchMax, avail);
                if (mt_max < cap) {
                    const uint32_t nl = wave_cmp<false>(cur - d, cur, mt_max, cap) & 0x7FFFFFFFu;
                    if (nl > mt_max) { cq_push(kOpExt, nl, d); mt_max = nl; }
                    top_open = nl >= cap;
                }
            }
        }

        prof_mark(1);
        const bool nice = mt_max >= kNice;                          // :1514
        n_nice += nice ? 1 : 0;
        if (G.workers) {
            // tell the worker that owns this position's BT4 head whether the call happens
            if (pflags & 32u) {
                n_unc++;
                if (W::lane() == 0) W::st_agent(G.bt_flag + (a - G.batch_a0), nice ? kFlagSkip : kFlagCall);
            } else if (nice) {
                // the pre-filter promised that no match of 65+ bytes ends up in the table at a-1
                fail(kErrInternal, a);
            }
        }
        // inside a nice region the finders run at every 8th PARSE-RELATIVE position (:1529): the segment
        // start is the parser wave's knowledge, final for `a` once it has parsed every earlier position
        // A segment cannot end inside a nice region (the table of the previous position reaches >= 64 further,
        // :1550-1554) except at the forced cut -- 4096 positions (:1469) or the chunk end (:1802) -- so the parser is
        // asked once per region and the forced cut is followed here.
        bool call = true;
        if (nice) {
            if (!prev_nice) {
                // which segment is `a` in: known once the parser has seen an edge that reaches beyond a (the table of
                // a-1 does), or has ended a segment exactly at a
                if (W::xw_load(&W::lds()->x_bcover) <= a && !wait_ge(&W::lds()->x_bpos, a, &W::lds()->x_bcover, a + 1)) return;
                seg_s = W::xw_load(&W::lds()->x_bseg);
            } else if (a == seg_cut) seg_s = a;
            seg_cut = seg_s + umin(kParseMax, pos_end_abs - seg_s);
            call = !((a - seg_s) & 7u);
        }
        prev_nice = nice;
        const bool have4 = call && avail >= 4, have256 = call && avail >= 256;
        const uint32_t max_len = umin(avail, kMatchMax);            // :915, :987
        if (G.workers && !nice && have4 && !(W::rec_get(rec, 14) & kBtReady)) {
            bt_refresh(s, q);
            rec = W::rec_load(W::lds()->pf_rec + pf_buf * (32 * kPf) + 32 * s);
        }

        if (have4) {
            const uint32_t v4 = W::rec_get(rec, 0);
            const uint32_t h2 = hash4(v4 & 0xFFFFu), h3 = hash4(v4 & 0xFFFFFFu), h4 = hash4(v4);   // :1516-1518
            const uint32_t i2 = h2 >> 20, i3 = h3 >> g.ht3_shift;   // bucket base NOT scaled by rows (:912)
            const uint32_t tag2 = h2 & g.tag_mask, tag3 = h3 & g.tag_mask;
            uint32_t row[3], len[3], valid = 0;
            const bool fresh_ht = !(pstale & 1);
            if (fresh_ht) {
                row[1] = W::rec_get(rec, 2);                        // HT3 row 0 moves down one row
            } else {
                // an earlier position rewrote a row this slot had read: read them again, compare now
                // The rows as they stand now: what the look-ahead read, overwritten by what the earlier slots of this
                // batch have stored since (each lane knows what its own slot wrote) -- no trip to memory
                n_sht++;
                row[0] = W::rec_get(rec, 27); row[1] = W::rec_get(rec, 2); row[2] = W::rec_get(rec, 28);
                W::pfl_rows_now(pfl, s, i2, i3, q - s, g.wbits, g.tag_mask, g.ht3_shift, row);
                uint32_t job_sp[8] = { 0, 0, 0 }, job_cap[8] = { 0, 0, 0 }, job_len[8];
                for (int k = 0; k < 3; k++) {
                    const uint32_t sp = row[k] & g.wmask;
                    if ((row[k] >> g.wbits) == (k ? tag3 : tag2) && sp < q && q - sp <= g.wmask) {   // :922-925
                        valid |= 1u << k; job_sp[k] = a - (q - sp); job_cap[k] = max_len;
                    }
                }
                W::cmp_multi(G.in, job_sp, a, job_cap, valid, job_len);
                for (int k = 0; k < 3; k++) len[k] = W::uni(job_len[k]);
            }
            // rows always rotate, compare or not (:935-936); q is stored un-masked (:913)
            G.ht2[i2] = q | (tag2 << g.wbits);                  // wave-uniform stores
            G.ht3[i3] = q | (tag3 << g.wbits);
            G.ht3[i3 + 1] = row[1];
            W::pfl_wrote(pfl, s, row[1]);
            n_ht++;
            prof_mark(2);
            bool bt_done = false;
            if (fresh_ht) {
                // the lane that looked at this position already ran the HT2/HT3 logic (and fetched the BT4 result
                // of the worker lanes if it was there): all table updates of the position in one pass
                const uint32_t np = W::rec_get(rec, 4), ready = W::rec_get(rec, 14), cnt = ready & 0x1FFu;
                n_cmp += np >> 8;
                bt_done = !nice && G.workers && (ready & kBtReady) && cnt <= kBtxPairs;
                const uint32_t nh = np & 0xFFu, nb = bt_done ? cnt : 0;
                if (nh + nb) mt_apply_set(s, rec, nh, nb, W::rec_get(rec, bt_done ? 25 : 23), W::rec_get(rec, bt_done ? 26 : 24));
            } else {
            // HT2 (:917-933)
            if ((valid & 1) && 1 < max_len) {
                const uint32_t l = len[0], d = q - (row[0] & g.wmask);
                n_cmp += l + (l < max_len);
                if (l > 1 && l >= match_min(d)) mt_update(d, l, l >= max_len);
            }
            // HT3, two rows, `best` gates the second (:917-933)
            uint32_t best = 1;
            for (int k = 1; k < 3; k++) {
                if (!((valid >> k) & 1) || !(best < max_len)) continue;
                const uint32_t l = len[k], d = q - (row[k] & g.wmask);
                n_cmp += l + (l < max_len);
                if (l > best && l >= match_min(d)) { mt_update(d, l, l >= max_len); best = l; }
            }
            }
            prof_mark(4);
            if (!nice && !bt_done) {                                // BT4 (:1522)
                if (G.workers) {
                    const uint32_t ready = W::rec_get(rec, 14), cnt = ready & 0x1FFu;
#ifdef NLZM_SIM_DEBUG
                    if (a == NLZM_SIM_DEBUG) printf("dbg a=%u ready=%08x cnt=%u unc=%u nice=%d max_len=%u live_ready=%08x\n", a, ready, cnt, pflags & 32u, (int)nice, max_len, G.bt_ready[(a - G.batch_a0) * kBtRec]);
#endif
                    if ((ready & kBtReady) && cnt <= 4) {
                        if (cnt) cq_push(kOpSet, s | (pf_buf << 6) | (kSetBt << 7), 0);
                        for (uint32_t k = 0; k < cnt; k++) {
                            const uint32_t l = W::rec_get(rec, 16 + 2 * k) & 0x1FFu;
                            mt_note(W::rec_get(rec, 15 + 2 * k), l, l >= max_len);
                        }
                    } else if ((ready & kBtReady) && cnt <= kBtxPairs) {
                        const typename W::Rec xr = W::rec_load(W::lds()->pf_btx + (pf_buf * kPf + s) * (2 * kBtxPairs));
                        cq_push(kOpSet, s | (pf_buf << 6) | (kSetBtx << 7), 0);
                        for (uint32_t k = 0; k < cnt; k++) {
                            const uint32_t l = W::rec_get(xr, 2 * k + 1);
                            mt_note(W::rec_get(xr, 2 * k), l, l >= max_len);
                        }
                    } else { n_sbt++; bt_consume(a, max_len); }
                } else bt_step(a, h4, max_len);
            }
            prof_mark(5);
        }
        if (have256) {
            // carried long match (:1056-1069)
            if (rk_len > 0) {
                if (q - rk_to < rk_len) {
                    const uint32_t d = rk_to - rk_from, l = rk_len - (q - rk_to);
                    if (l >= match_min(d)) mt_update(d, umin(l, kMatchMax), rk_open || l >= kMatchMax);
                } else rk_len = 0;
            }
            const uint32_t rkh = W::rec_get(rec, 11), myslot = rkh >> g.rk_shift;
            bool fresh = !(pstale & 4);
            // window ends passed since the last call: insert with the CALLING position (:1084-1087)
            for (uint32_t e = (rk_end | 255u) + 1; e < q + 256; e += 256) {
                const uint32_t hh = W::uni(G.rkhash[(uint32_t)(base + e - 256)]);
                G.rk_table[hh >> g.rk_shift] = q | (hh << g.wbits);
                pf_mark_rk(s, hh >> g.rk_shift);
                if ((hh >> g.rk_shift) == myslot) fresh = false;
                n_rki++;
            }
            rk_end = q + 256;
            if (rk_len < 256) {                                     // :1090-1107
                n_rkp++;
                uint32_t rkv, l = 0;
                bool ok, exact = true;
                if (fresh) { rkv = W::rec_get(rec, 12); ok = (pflags & 8u) != 0; l = W::rec_get(rec, 13); exact = !(pflags & 16u); }
                else {
                    n_srk++;
                    W::sync_global();
                    rkv = W::uni(G.rk_table[myslot]);
                    const uint32_t sp = rkv & g.wmask;
                    ok = (rkv >> g.wbits) == (rkh & g.tag_mask) && sp < q && q - sp <= g.wmask;
                    exact = false;
                }
                if (ok) {
                    const uint32_t d = q - (rkv & g.wmask), cap = avail & 0xFFFFu;   // uint16 parameter (:760)
                    if (!exact) l = wave_cmp<false>(cur - d, cur, 0, cap) & 0x7FFFFFFFu;
                    n_cmp += l + (l < cap);
                    if (l >= rk_len && l >= match_min(d)) {
                        rk_open = l >= cap;
                        mt_update(d, umin(l, kMatchMax), rk_open || l >= kMatchMax);
                        rk_from = q - d; rk_to = q; rk_len = l;
                    }
                }
            }
            if (!(q & 255u)) {                                      // aligned insert after the probe (:1109-1112)
                G.rk_table[myslot] = q | (rkh << g.wbits);
                pf_mark_rk(s, myslot);
                n_rki++;
            }
        }
        prof_mark(6);
    }

    // The BT4 result of slot s was not in when the look-ahead read it.  Before waiting for it, every later slot
    // of the batch that was not in either looks again (one load round for all of them): the worker lanes have had
    // the time the finder wave spent on the slots in between.
    NLZM_HD void bt_refresh(uint32_t s, uint32_t q_s)
    {
        W::sync();
        for (uint32_t j = W::lane(); j < pf_n; j += W::width()) {
            if (j < s) continue;
            uint32_t *rec = W::lds()->pf_rec + pf_buf * (32 * kPf) + 32 * j;
            if (rec[14] & kBtReady) continue;
            const uint32_t x = pf_base + j, avail = la_end - (q_s + (j - s));
            if (avail < 4) continue;
            const unsigned long long bi = x - G.batch_a0;
            W::wait_hook(G.hook_user, x);
            const uint32_t *br = G.bt_ready + bi * kBtRec;
            const uint32_t ready = W::ld_agent(br);
            if (!(ready & kBtReady)) continue;
            // (the record's pairs were written through before its ready word)
            const uint32_t nb = umin(ready & 0x1FFu, 4u), max_len = umin(avail, kMatchMax);
            uint32_t sl = rec[23], sd = rec[24];
            for (uint32_t k = 0; k < nb; k++) {
                const uint32_t d = W::ld_agent(br + 1 + 2 * k), l = W::ld_agent(br + 2 + 2 * k);
                const uint32_t lo = l | ((uint32_t)(l >= max_len) << 31);
                rec[15 + 2 * k] = d; rec[16 + 2 * k] = lo;
                if (l > (sl & 0x1FFu)) { sl = lo; sd = d; }
                else if (l == (sl & 0x1FFu)) { sl |= lo & 0x80000000u; sd = umin(sd, d); }
            }
            const uint32_t cnt = ready & 0x1FFu;
            if (cnt > 4 && cnt <= kBtxPairs) {                      // the longer list whole, as the look-ahead would have taken it
                const uint32_t *pairs = G.bt_pairs + bi * (2 * kBtMaxPairs);
                uint32_t *bx = W::lds()->pf_btx + (pf_buf * kPf + j) * (2 * kBtxPairs);
                for (uint32_t k = 0; k < 2 * cnt; k++) bx[k] = W::ld_agent(pairs + k);
                for (uint32_t k = 4; k < cnt; k++) {
                    const uint32_t d = bx[2 * k], l = bx[2 * k + 1], lo = l | ((uint32_t)(l >= max_len) << 31);
                    if (l > (sl & 0x1FFu)) { sl = lo; sd = d; }
                    else if (l == (sl & 0x1FFu)) { sl |= lo & 0x80000000u; sd = umin(sd, d); }
                }
            }
            rec[25] = sl; rec[26] = sd; rec[14] = ready;
            W::pfl_update(pfl, j, sl, sd, rec[29] && cnt <= kBtxPairs && !(sl >> 31));
        }
        W::sync();
    }

    // BT4 result of a worker lane: wait for it, then merge its pairs (MatchTable::Update, :996-998).
    NLZM_HD void bt_consume(uint32_t a, uint32_t max_len)
    {
        const unsigned long long bi = a - G.batch_a0;
        W::wait_hook(G.hook_user, a);
        const unsigned long long t0 = W::clock();
        uint32_t spins = 0, v;
        const unsigned long long c0 = W::tick();
        // the whole record with one load instruction (one 64-byte line, one request): the ready word was stored after
        // every pair had been written through (sc1) and drained, so a record that shows it also shows the pairs
        typename W::Rec br = W::rec_load_agent(G.bt_ready + bi * kBtRec);
        while (!((v = W::rec_get(br, 0)) & kBtReady)) {
            if ((++spins & 255u) == 0 && W::clock() - t0 > W::timeout_ticks()) { fail(kErrTimeout, a); return; }
            if ((spins & 255u) == 0 && W::xw_load(&W::lds()->x_err)) { err = kErrInternal + 100; return; }
            W::sleep();
            br = W::rec_load_agent(G.bt_ready + bi * kBtRec);
        }
        wait_cyc += W::tick() - c0;
        const uint32_t count = v & 0x1FFu;
        if (count <= 4) {
            for (uint32_t k = 0; k < count; k++) {
                const uint32_t l = W::rec_get(br, 2 + 2 * k);
                mt_update(W::rec_get(br, 1 + 2 * k), l, l >= max_len);
            }
            return;
        }
        const uint32_t *pairs = G.bt_pairs + bi * (2 * kBtMaxPairs);
        W::sync();
        for (uint32_t i = W::lane(); i < 2 * count; i += W::width()) W::lds()->btpairs[i] = W::ld_agent(pairs + i);
        W::sync();
        for (uint32_t k = 0; k < count; k++) {
            const uint32_t d = W::uni(W::lds()->btpairs[2 * k]), l = W::uni(W::lds()->btpairs[2 * k + 1]);
            mt_update(d, l, l >= max_len);
        }
    }

    // BT4 inside the master (workers off): wave-wide compares, uniform descent.
    NLZM_HD void bt_step(uint32_t a, uint32_t h4, uint32_t max_len)
    {
        WaveCmp wcmp;
        MtSink sink{ this, max_len };
        W::cnt_add(&W::lds()->cnt.bt_calls, 1);
        BtMem mem{ G.bt_heads, G.bt_tree };
        bt_find_and_update_uniform(mem, a, h4, max_len, wcmp, sink);
    }

    struct BtMem { uint32_t *heads, *tree; };

    // same control flow as bt_find_and_update, but every store is done by lane 0 only
    // and followed by a global sync so the wave's later (uniform) loads see it
    template <class Cmp, class Sink>
    NLZM_HD void bt_find_and_update_uniform(BtMem &mem, uint32_t a, uint32_t h4, uint32_t max_len, Cmp &cmp, Sink &sink)
    {
        uint32_t *heads = mem.heads, *tree = mem.tree;
        const uint32_t wmask = g.wmask, tmask = g.bt_tmask;
        uint32_t pend_l = (a & tmask) << 1, pend_r = pend_l + 1;
        uint32_t len_l = 0, len_r = 0;
        uint32_t sp = heads[h4 >> g.bt_shift];
        if (W::lane() == 0) heads[h4 >> g.bt_shift] = a;
        uint32_t tests = 256;
        while (sp != kNone && a > sp && a - sp <= wmask && tests-- > 0) {
            W::cnt_add(&W::lds()->cnt.bt_tests, 1);
            const uint32_t pair = (sp & tmask) << 1;
            const uint32_t pl = tree[pair], pr = tree[pair + 1];
            const uint32_t r = cmp(G.in + sp, G.in + a, umin(len_l, len_r), max_len);
            const uint32_t l = r & 0x7FFFFFFFu;
            if (l >= match_min(a - sp)) sink(a - sp, l);
            if (l == max_len) {
                if (W::lane() == 0) { tree[pend_l] = pl; tree[pend_r] = pr; }
                W::sync_global();
                return;
            }
            if (r >> 31) { if (W::lane() == 0) tree[pend_l] = sp; pend_l = pair + 1; sp = pr; len_r = l; }
            else         { if (W::lane() == 0) tree[pend_r] = sp; pend_r = pair;     sp = pl; len_l = l; }
        }
        if (W::lane() == 0) { tree[pend_r] = kNone; tree[pend_l] = kNone; }
        W::sync_global();
    }

    // ---- cross-wave waits (both waves of the master share the CU's LDS; DS operations of a wave
    // execute in order, so a reader that sees a counter also sees what was written before it) -------
    NLZM_HD bool wait_ge(const uint32_t *w, uint32_t v, const uint32_t *w2 = nullptr, uint32_t v2 = 0)   // until *w >= v (or *w2 >= v2)
    {
        if (W::xw_load(w) >= v) return true;
        const unsigned long long t0 = W::clock();
        const unsigned long long c0 = W::tick();
        uint32_t spins = 0;
        for (;;) {
            if (W::xw_load(w) >= v || (w2 && W::xw_load(w2) >= v2)) {
                wait_cyc += W::tick() - c0;
                return true;
            }
            if (W::xw_load(&W::lds()->x_err)) { err = kErrInternal + 100; return false; }
            if ((++spins & 1023u) == 0 && W::clock() - t0 > W::timeout_ticks()) { fail(kErrTimeout, v); return false; }
            W::xw_pause();
        }
    }
    NLZM_HD bool wait_space(uint32_t a)                             // until position a - kEr has been parsed
    {
        if (a - bpos_seen < kEr) return true;                       // x_bpos only grows: the last value seen is a safe bound
        if (a - (bpos_seen = W::xw_load(&W::lds()->x_bpos)) < kEr) return true;
        const unsigned long long t0 = W::clock();
        const unsigned long long c0 = W::tick();
        uint32_t spins = 0;
        while (a - (bpos_seen = W::xw_load(&W::lds()->x_bpos)) >= kEr) {
            if (W::xw_load(&W::lds()->x_err)) { err = kErrInternal + 100; return false; }
            if ((++spins & 1023u) == 0 && W::clock() - t0 > W::timeout_ticks()) { fail(kErrTimeout, a); return false; }
            W::xw_pause();
        }
        wait_cyc += W::tick() - c0;
        return true;
    }
    NLZM_HD void fail(uint32_t code, uint32_t info)
    {
        err = code; err_info0 = info;
        W::xw_store(&W::lds()->x_err, code);
    }

    // =========================== table wave: publishing =====================================
    // publish position a: table length, input byte, table entries (MatchTable as mt_carry holds it, :1543)
    NLZM_HD void t_publish(uint32_t a, uint32_t lit)
    {
        uint32_t *e = W::lds()->er_tab + (a & (kEr - 1)) * 64;
        uint32_t hdr = mt_max;
        W::sync();
        if (mt_max > 63) {
            if (!wait_ge(&W::lds()->x_long_free, a_long + 1)) return;      // x_long_free starts at kErLong
            uint32_t *l = W::lds()->er_long + (a_long % kErLong) * (kMatchMax + 8);
            for (uint32_t i = W::lane(); i <= mt_max; i += W::width()) l[i] = mt(i);
            hdr |= ((a_long % kErLong) + 1) << 16;
            a_long++;
        }
        for (uint32_t i = 2 + W::lane(); i < 64; i += W::width()) e[i] = mt(i);
        e[0] = hdr;
        e[1] = lit;
        W::sync();
        W::xw_store(&W::lds()->x_apos, a + 1);
    }

    // slots [s, s+n) of the look-ahead are `simple`: carry, HT2/HT3/BT4 updates as the records list them, nothing from
    // RK256, no extension, not nice.  The table wave replays them from the records; here only the rows rotate (all
    // slots at once) and the table's length / top entry are followed.
    NLZM_HD void fast_run(uint32_t q, uint32_t a, uint32_t s, uint32_t n)
    {
#ifdef NLZM_SIM_COUNT
        g_dbg[4] += n; g_dbg[5]++;
#endif
        cq_push(kOpRun, s | ((n - 1) << 6) | (pf_buf << 12), a);
        cq_flush();
        W::pfl_run_store(pfl, s, n, G.ht2, G.ht3, q, g.wbits, g.tag_mask, g.ht3_shift);
        uint32_t cmpb = 0;
        for (uint32_t i = 0; i < n; i++) {
            mt_max = mt_max > 1 ? mt_max - 1 : 0;                   // carry (:823-833)
            if (mt_max >= kNice) { fail(kErrInternal, a + i); return; }   // the pre-filter promised otherwise
            const uint32_t sl = W::pfl_sl(pfl, s + i), lm = sl & 0x1FFu;
            if (lm > mt_max) { mt_max = lm; top_d = W::pfl_sd(pfl, s + i); }
            else if (lm && lm == mt_max) top_d = umin(top_d, W::pfl_sd(pfl, s + i));
            cmpb += W::pfl_cmpb(pfl, s + i);
        }
        n_cmp += cmpb; n_ht += n; n_rkp += n;
        rk_end = q + n - 1 + 256;
        prev_nice = false;
    }

    NLZM_HD void run_chunk_finder(uint32_t ci)
    {
        const unsigned long long chunk_abs = (unsigned long long)ci * g.chunk_size;
        const unsigned long long remain = g.n - chunk_abs;
        const uint32_t chunk_read = (uint32_t)(remain < g.feed ? remain : g.feed);
        const uint32_t p_end = umin(g.chunk_size, chunk_read);
        const uint32_t W2 = 2u * (g.wmask + 1);
        if (chunk_abs - base >= W2) {                               // :1786-1792
            base += g.wmask + 1;
            W::cnt_add(&W::lds()->cnt.shifts, 1);
            G.ht2[0] = kNone; G.ht3[0] = kNone;                     // MatchFinderHT::Shift (:940-957)
            if (rk_end >= g.wmask + 1) rk_end -= g.wmask + 1; else rk_end = 0;   // :1115-1123
            W::sync_global();
            // BT4 keeps absolute positions: no pass over the tree (appendix D.3)
        }
        const uint32_t chunk_q = (uint32_t)(chunk_abs - base);
        la_end = chunk_q + chunk_read;
        pf_n = 0;                           // the look-ahead never crosses a chunk (lookahead limit, rebase)
        counts_zero();
        const uint32_t a0 = (uint32_t)chunk_abs, a1 = a0 + p_end;
        uint32_t a = a0;
        while (a < a1 && !err) {
            if (a - pf_base >= pf_n) { pf_fill(a, a1, a0 + chunk_read); if (err) break; }
            const uint32_t s = a - pf_base, q = chunk_q + (a - a0);
            // a run of slots the look-ahead has settled completely: passed to the table wave as ONE command
            if (rk_len == 0 && !top_open && rk_end == q + 255) {
                const unsigned long long m = W::pfl_run_mask(pfl, pf_n) >> s;
                const uint32_t n = ~m ? (uint32_t)__builtin_ctzll(~m) : 64u;
                if (n) { fast_run(q, a, s, n); a += n; continue; }
            }
            prof_start();
            finders(q, a, a1);
            if (err) break;
            cq_push(kOpEnd, W::lds()->win[a - pf_base], a);       // the table of position a is complete (:1543)
            cq_flush();
            prof_mark(12);
            a++;
        }
        counts_flush();
    }

    NLZM_HD void run_finder(uint32_t c0, uint32_t c1)
    {
        Persist *P = G.persist;
        mt_max = W::uni(P->mt_max);
        top_d = W::uni(P->mt_delta[mt_max]); top_open = true; rk_open = true;     // conservative across launches
        pf_base = 0; pf_n = 0; cq_n = 0; cq_seen = 0; prev_nice = false; seg_s = 0; seg_cut = 0;
        pf_buf = 0; pf_mark0 = 0; pf_mark1 = 0;
        wait_cyc = 0; role_t0 = W::tick();
        rk_from = W::uni(P->rk_from); rk_to = W::uni(P->rk_to); rk_len = W::uni(P->rk_len); rk_end = W::uni(P->rk_end);
        base = ((unsigned long long)W::uni((uint32_t)(P->reb_base >> 32)) << 32) | W::uni((uint32_t)P->reb_base);
        err = W::uni(P->error); err_info0 = 0;
        counts_zero();
#ifdef NLZM_PROFILE
        for (int k = 0; k < 16; k++) prof[k] = 0;
#endif
        W::sync();
        uint32_t ci = c0;
        for (; ci < c1 && !err; ci++) run_chunk_finder(ci);
        W::sync();
        if (W::lane() == 0) {
            P->rk_from = rk_from; P->rk_to = rk_to; P->rk_len = rk_len; P->rk_end = rk_end;
            P->reb_base = base;
            P->prof[16] += wait_cyc; P->prof[17] += W::tick() - role_t0; P->prof[37] += wait_cq; P->prof[38] += wait_guard;
            if (err && err < 100) { P->error = err; P->error_info[0] = err_info0; P->error_info[1] = 1; }
#ifdef NLZM_PROFILE
            for (int k = 0; k < 7; k++) P->prof[k] += prof[k];
            P->prof[12] += prof[12]; P->prof[14] += prof[13];
#endif
        }
        W::sync_global();
        W::xw_add(&W::lds()->x_adone, 1u);
    }

    // =========================== wave T: the match table ===================================
    NLZM_HD void run_table(uint32_t c0, uint32_t c1)
    {
        Persist *P = G.persist;
        mt_base = 0; mt_max = W::uni(P->mt_max);
        for (uint32_t i = W::lane(); i < 512; i += W::width()) W::lds()->mt[i] = (i <= mt_max && i <= kMatchMax) ? P->mt_delta[i] : kNone;
        err = W::uni(P->error); err_info0 = 0;
        a_long = 0; pf_base = 0; bpos_seen = (uint32_t)((unsigned long long)c0 * g.chunk_size);
        wait_cyc = 0; role_t0 = W::tick();
#ifdef NLZM_PROFILE
        for (int k = 0; k < 16; k++) prof[k] = 0;
#endif
        W::sync();
        unsigned long long a_end = (unsigned long long)c1 * g.chunk_size;
        if (a_end > g.n) a_end = g.n;
        const uint32_t last = (uint32_t)a_end;              // one END command per position of the launch
        uint32_t done = 0, have = 0, ended = (uint32_t)((unsigned long long)c0 * g.chunk_size);
        while (ended < last && !err) {
            if (done == have) {                             // fetch what the finder wave has written meanwhile
                const unsigned long long t0 = W::clock();
                const unsigned long long c0 = W::tick();
                uint32_t spins = 0;
                while ((have = W::xw_load(&W::lds()->x_cpos)) == done) {
                    if (W::xw_load(&W::lds()->x_err)) { err = kErrInternal + 100; break; }
                    if ((++spins & 1023u) == 0 && W::clock() - t0 > W::timeout_ticks()) { fail(kErrTimeout, done); break; }
                    W::xw_pause();
                }
                wait_cyc += W::tick() - c0;
                if (err) break;
            }
            // up to 32 commands with one LDS read each for the two words, then picked by lane
            const uint32_t nb = umin(have - done, 32u);
            const uint32_t first = done;
            const typename W::Rec r0 = W::rec_load_fn32([=](uint32_t i) { return W::lds()->cq[((first + i) & (kCq - 1)) * 2]; });
            const typename W::Rec r1 = W::rec_load_fn32([=](uint32_t i) { return W::lds()->cq[((first + i) & (kCq - 1)) * 2 + 1]; });
            prof_mark(0);
            for (uint32_t k = 0; k < nb && !err; k++) {
                prof_start();
                const uint32_t c = W::rec_get(r0, k), v = W::rec_get(r1, k), op = c & 0xFFu, arg = c >> 8;
                if (op == kOpUpd) { t_update(v, arg); prof_mark(1); }
                else if (op == kOpCarry) { t_carry(); prof_mark(2); }
                else if (op == kOpExt) t_extend(v, arg);
                else if (op == kOpRun) {                    // slots [s0, s0+n) of a look-ahead batch, first position v
                    const uint32_t s0 = arg & 63u, n = ((arg >> 6) & 63u) + 1, buf = (arg >> 12) & 1u;
                    typename W::Rec r = W::rec_load(W::lds()->pf_rec + buf * (32 * kPf) + 32 * s0);
                    for (uint32_t i = 0; i < n && !err; i++) {
                        // the next slot's record is requested before this one is worked on
                        const typename W::Rec rn = W::rec_load(W::lds()->pf_rec + buf * (32 * kPf) + 32 * (s0 + (i + 1 < n ? i + 1 : i)));
                        t_carry();
                        {
                            const uint32_t nb = W::rec_get(r, 14) & 0x1FFu;
                            if (nb <= 4) t_apply_set(r, W::rec_get(r, 4) & 0xFFu, nb, W::rec_get(r, 25) & 0x1FFu);
                            else { t_apply_set(r, W::rec_get(r, 4) & 0xFFu, 0, W::rec_get(r, 23) & 0x1FFu); t_apply_btx(buf, s0 + i, nb); }
                        }
                        if (!wait_space(v + i)) break;
                        capture(v + i);
                        t_publish(v + i, W::rec_get(r, 0) & 0xFFu);
                        ended = v + i + 1;
                        r = rn;
                    }
                    prof_mark(3);
                }
                else if (op == kOpSet) {                    // updates of one slot, read from its look-ahead record
                    const uint32_t s0 = arg & 63u, buf = (arg >> 6) & 1u, mode = arg >> 7;
                    const typename W::Rec r = W::rec_load(W::lds()->pf_rec + buf * (32 * kPf) + 32 * s0);
                    if (mode & kSetBtx) {
                        if (mode & kSetHt) t_apply_set(r, W::rec_get(r, 4) & 0xFFu, 0, W::rec_get(r, 23) & 0x1FFu);
                        t_apply_btx(buf, s0, W::rec_get(r, 14) & 0x1FFu);
                    } else {
                        const uint32_t nh = (mode & kSetHt) ? W::rec_get(r, 4) & 0xFFu : 0u, nb = (mode & kSetBt) ? W::rec_get(r, 14) & 0x1FFu : 0u;
                        uint32_t maxl = (mode & kSetHt) ? W::rec_get(r, 23) & 0x1FFu : 0u;
                        if (nb) maxl = umax(maxl, W::rec_get(r, 16 + 2 * (nb - 1)) & 0x1FFu);      // (record-setters: the last is the longest)
                        t_apply_set(r, nh, nb, maxl);
                    }
                    prof_mark(4);
                }
                else {                                      // kOpEnd: publish position v, input byte arg
                    if (!wait_space(v)) break;              // slot v % kEr is free again
                    capture(v);
                    t_publish(v, arg);
                    ended = v + 1;
                    prof_mark(5);
                }
            }
            done += nb;
            W::xw_store(&W::lds()->x_tpos, done);
            prof_start();
        }
        W::sync();
        for (uint32_t i = W::lane(); i <= kMatchMax; i += W::width()) P->mt_delta[i] = i <= mt_max ? mt(i) : 0;
        if (W::lane() == 0) {
            P->mt_max = mt_max;
            P->prof[18] += wait_cyc; P->prof[19] += W::tick() - role_t0;
#ifdef NLZM_PROFILE
            for (int k = 0; k < 6; k++) P->prof[48 + k] += prof[k];
#endif
            if (err && err < 100) { P->error = err; P->error_info[0] = err_info0; P->error_info[1] = 3; }
        }
        W::sync_global();
        W::xw_add(&W::lds()->x_adone, 1u);
    }

    // =========================== waves E1a, E1b, E2: the edges that leave a node ============
    // The parser wave finalises node p (its cost and rep set are complete once the literal edge of p-1 and the
    // match edges of every node <= p-2 have been relaxed) and posts it.  Its sampled-length edges (:1558-1596)
    // and its explicit rep probes (:1598-1628) all end at nodes >= p+2:
    //   E1a lists the sampled edges (length, distance, cost as dict edge and as rep edge),
    //   E1b lists the rep probes (match length at each rep distance, cost),
    //   E2  relaxes both lists in the reference's order -- it is the only writer of nodes >= p+2.
    // The parser meanwhile relaxes the literal edge of p (after E2 is done with p-1: the reference's order at
    // node p+1) and posts node p+1, so the lists of p+1 are made while the edges of p are applied.
    //
    // The post block of a slot (lanes 0..19, 23..28) and the tag of the position's edge list (20, 21) with one LDS read
    NLZM_HD typename W::Rec edge_fetch(uint32_t slot, uint32_t lslot)
    {
        return W::rec_load_fn32([=](uint32_t i) {
            return (i == 20 || i == 21) ? W::lds()->ea_tag[lslot * 2 + (i - 20)] : W::lds()->post[(i < 20 || i >= 23) ? slot : 0][i];
        });
    }
    // wait until `next` is posted and, for the apply wave, its sampled edges are listed for this segment; false: leave
    NLZM_HD bool edge_wait(typename W::Rec &rq, uint32_t next, bool need_lists)
    {
        auto ready = [&](const typename W::Rec &r) {      // (no branch per term)
            const uint32_t miss = (W::rec_get(r, 4) ^ next) | (need_lists ? (W::rec_get(r, 20) ^ (next + 1)) | (W::rec_get(r, 21) ^ W::rec_get(r, 16)) : 0u);
            return (uint32_t)(W::rec_get(r, 0) > next) & (uint32_t)(miss == 0);
        };
        rq = edge_fetch(next & 1u, next & 3u);
        if (ready(rq)) return true;
        const unsigned long long t0 = W::clock(), c0 = W::tick();
        uint32_t spins = 0;
        for (;;) {
            rq = edge_fetch(next & 1u, next & 3u);
            if (ready(rq)) break;
            if (W::rec_get(rq, 0) == kNone) return false;
            if ((++spins & 63u) == 0) {
                if (W::xw_load(&W::lds()->x_err)) return false;
                if ((spins & 1023u) == 0 && W::clock() - t0 > 4 * W::timeout_ticks()) return false;
            }
            W::xw_pause();
        }
        wait_cyc += W::tick() - c0;
        return true;
    }
    NLZM_HD void edge_leave(int k)
    {
        if (W::lane() == 0) {
            G.persist->prof[k] += wait_cyc; G.persist->prof[k + 1] += W::tick() - role_t0;
            if (k == 22) { G.persist->prof[39] += wait_rep; G.persist->prof[46] += cmp_acc; }
#ifdef NLZM_PROFILE
            // latency from the parser's post of a node to: list written (24: sampled, 26/28: rep) / apply started, done (22)
            G.persist->prof[32 + (k == 22 ? 0 : (k == 24 ? 1 : (k == 26 ? 2 : 3)))] += lat_sum;
            if (k == 22) { G.persist->prof[36] += lat_sum2; for (int i = 0; i < 6; i++) G.persist->prof[40 + i] += prof[i]; }
#endif
        }
    }

    // E1a: sampled lengths tl_k = max_len - k*step while >= 2 (:1558-1562), one lane per length.  The list of a position
    // needs its match table, its index in the segment (for the cap max_parse - p, :1545) and the segment's prices --
    // nothing of the node -- so this wave runs ahead of the parser inside the segment, up to the list ring's depth.
    NLZM_HD void run_edge_list(uint32_t a_first)
    {
        err = 0; err_info0 = 0; wait_cyc = 0; role_t0 = W::tick();
        uint32_t next = a_first, seq = 0, seg_a = 0, maxp = 0, pc_dict = 0, pc_rep = 0;
        for (;;) {
            // segment, progress of the apply wave and of the table wave, leave order: one read
            const typename W::Rec st = W::rec_load_fn([=](uint32_t i) {
                return i < 3 ? W::lds()->seginfo[i] : (i == 3 ? W::lds()->x_sdone : (i == 4 ? W::lds()->x_apos : (i == 5 ? W::lds()->post[0][0] : W::lds()->x_err)));
            });
            if (W::rec_get(st, 5) == kNone || W::rec_get(st, 6)) { edge_leave(24); return; }
            if (W::rec_get(st, 0) != seq) {
                // a new segment: its prices are in place (the parser rebuilds them before it announces the segment)
                const typename W::Rec s2 = W::rec_load_fn([=](uint32_t i) { return W::lds()->seginfo[i & 3u]; });   // a consistent set
                if (W::rec_get(s2, 0) != W::rec_get(st, 0)) continue;
                seq = W::rec_get(s2, 0); seg_a = W::rec_get(s2, 1); maxp = W::rec_get(s2, 2);
                next = seg_a;
                pc_dict = W::uni(price(kCtxCmd, 1)); pc_rep = W::uni(price(kCtxCmd, 2));
            }
            const uint32_t sdone = W::rec_get(st, 3), apos = W::rec_get(st, 4);
            if (!seq || next - seg_a >= maxp || (int32_t)(next - sdone) > 3 || (int32_t)(apos - next) < 1) {
                // nothing to list: ahead of the apply wave by the ring's depth, at the segment's end, or the table is not
                // out yet.  This wave shares its SIMD with the finder wave: it sleeps longer while it is ahead
                const unsigned long long c0 = W::tick();
                if ((int32_t)(apos - next) < 1) W::xw_pause(); else W::sleep();
                wait_cyc += W::tick() - c0;
                continue;
            }
            const uint32_t a = next, p = a - seg_a, slot = a & 3u;
            const uint32_t *e = W::lds()->er_tab + (a & (kEr - 1)) * 64;
            const uint32_t hdr = W::uni(e[0]);
            uint32_t max_len = umin(hdr & 0xFFFFu, maxp - p);       // :1545-1548, as the parser computes it
            if (max_len < kMatchMin) max_len = 0;
            if (max_len) {
                uint32_t step = (max_len - kMatchMin) >> 4;
                step += step == 0;
                auto edge = [&](uint32_t k) {
                    uint32_t *o = W::lds()->ea + (slot * 64 + k) * 4;
                    if (k * step > max_len - kMatchMin) { o[3] = 0; return; }
                    const uint32_t tl = max_len - k * step;
                    const uint32_t d = tab(e, hdr, tl);
                    const uint32_t mm = match_min(d);
                    if (tl < mm) { o[3] = 0; return; }
                    const uint32_t lv = tl - mm, lc = umin(lv, 3);
                    uint32_t nx, ex;
                    const uint32_t ds = dist_slot(d - 1, nx, ex);
                    const uint32_t plen = W::lds()->seg_len_price[lv];
                    o[0] = pc_dict + plen + (nx << 5) + W::lds()->seg_slot_price[lc * 64 + ds];
                    o[1] = pc_rep + plen + (2u << 5);
                    o[2] = d;
                    o[3] = tl | (1u << 12);
                };
                // k*step <= max_len - 2 leaves at most 32 lengths (step is 1 up to max_len 33)
                if (W::width() == 1) { for (uint32_t k = 0; k < 64; k++) edge(k); }
                else edge(W::lane());
            }
            W::sync();
            W::lds()->ea_tag[slot * 2 + 1] = seq;
            W::xw_store(&W::lds()->ea_tag[slot * 2], a + 1);
            next++;
        }
    }

    // E1b: explicit rep probes (:1598-1628): match length at each rep distance and the node cost through it.
    // The bytes come from HBM (a rep distance reaches anywhere in the window) and a node's rep set is known only
    // when the node is posted, so one probe list costs a full memory round trip: two waves take the even and the
    // odd positions, each with its own request/list slot.
    NLZM_HD void run_rep_list(uint32_t a_first, uint32_t parity)
    {
        err = 0; err_info0 = 0; wait_cyc = 0; role_t0 = W::tick();
        uint32_t next = a_first + ((a_first ^ parity) & 1u);
        const uint32_t slot = parity;
        // The list for a guessed rep set of `next` (the parser's guess while the node was not final) is written as
        // soon as it is measured: the apply wave checks the guess against the posted node itself and takes the list
        // without waiting for this wave to see the post.  (Costs are listed without the node's cost for that reason.)
        bool have_g = false;
        uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0, gseq = 0;
        // The four probes in straight-line scalar code (selects, no branch per rep); the two rare cases -- a match
        // that runs past the prefetched bytes, a probe long enough to relax (its price needs the length tables) --
        // are found by one test each.
        auto list = [&](uint32_t a, uint32_t rp0, uint32_t rp1, uint32_t rp2, uint32_t rp3, uint32_t l0, uint32_t l1, uint32_t l2, uint32_t l3,
                        uint32_t q, uint32_t rep_cap, uint32_t pc_rep) {
            const bool v0 = rp0 < q, v1 = rp1 < q, v2 = rp2 < q, v3 = rp3 < q;
            if (rep_cap > kRepPf && ((v0 && l0 == kRepPf) || (v1 && l1 == kRepPf) || (v2 && l2 == kRepPf) || (v3 && l3 == kRepPf))) {
                if (v0 && l0 == kRepPf) l0 = wave_cmp<false>(G.in + a - rp0, G.in + a, kRepPf, rep_cap) & 0x7FFFFFFFu;
                if (v1 && l1 == kRepPf) l1 = wave_cmp<false>(G.in + a - rp1, G.in + a, kRepPf, rep_cap) & 0x7FFFFFFFu;
                if (v2 && l2 == kRepPf) l2 = wave_cmp<false>(G.in + a - rp2, G.in + a, kRepPf, rep_cap) & 0x7FFFFFFFu;
                if (v3 && l3 == kRepPf) l3 = wave_cmp<false>(G.in + a - rp3, G.in + a, kRepPf, rep_cap) & 0x7FFFFFFFu;
            }
            l0 = umin(l0, rep_cap); l1 = umin(l1, rep_cap); l2 = umin(l2, rep_cap); l3 = umin(l3, rep_cap);     // min(len, 264)
            const uint32_t m0 = match_min(rp0), m1 = match_min(rp1), m2 = match_min(rp2), m3 = match_min(rp3);
            const bool x0 = v0 && l0 >= m0, x1 = v1 && l1 >= m1, x2 = v2 && l2 >= m2, x3 = v3 && l3 >= m3;
            uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            if (x0 || x1 || x2 || x3) {
                if (x0) c0 = W::uni(pc_rep + price_len(l0 - m0) + (2u << 5));
                if (x1) c1 = W::uni(pc_rep + price_len(l1 - m1) + (2u << 5));
                if (x2) c2 = W::uni(pc_rep + price_len(l2 - m2) + (2u << 5));
                if (x3) c3 = W::uni(pc_rep + price_len(l3 - m3) + (2u << 5));
            }
            uint32_t *o = W::lds()->eb + (a & 3u) * 16;
            o[0] = v0 ? (l0 | (1u << 31) | ((uint32_t)x0 << 30)) : 0u; o[1] = c0;
            o[2] = v1 ? (l1 | (1u << 31) | ((uint32_t)x1 << 30)) : 0u; o[3] = c1;
            o[4] = v2 ? (l2 | (1u << 31) | ((uint32_t)x2 << 30)) : 0u; o[5] = c2;
            o[6] = v3 ? (l3 | (1u << 31) | ((uint32_t)x3 << 30)) : 0u; o[7] = c3;
            W::sync();
        };
        for (;;) {
            typename W::Rec rq = edge_fetch(slot, 0);
            // (a post of a later position: the apply wave took the list made for the guess and the parser went on)
            auto later = [&](const typename W::Rec &r) { return W::rec_get(r, 0) == W::rec_get(r, 4) + 1 && (int32_t)(W::rec_get(r, 4) - next) > 0; };
            bool posted = (W::rec_get(rq, 0) > next && W::rec_get(rq, 4) == next) || later(rq);
            if (!posted) {
                const unsigned long long t0 = W::clock(), c0 = W::tick();
                uint32_t spins = 0;
                for (;;) {
                    if (W::rec_get(rq, 0) == kNone) { edge_leave(parity ? 28 : 26); return; }
                    if (!have_g && W::rec_get(rq, 23) > next && W::rec_get(rq, 24) == next) {
                        // measure for the guess now: the memory round trip is over when the node is posted
                        wait_cyc += W::tick() - c0;
                        g0 = W::rec_get(rq, 25); g1 = W::rec_get(rq, 26); g2 = W::rec_get(rq, 27); g3 = W::rec_get(rq, 28);
                        gseq = W::rec_get(rq, 31);
                        const typename W::RepPf gpf = W::rep_prefetch(G.in, g.n, next, g0, g1, g2, g3);
                        uint32_t gl[4];
                        W::rep_lengths(gpf, gl);
                        list(next, g0, g1, g2, g3, gl[0], gl[1], gl[2], gl[3], W::rec_get(rq, 29), W::rec_get(rq, 30), W::rec_get(rq, 3) & 0xFFFFu);
                        W::xw_store(&W::lds()->eb[(next & 3u) * 16 + 15], next + 1);    // listed for the guess of `next`
                        have_g = true;
                    } else {
                        if ((++spins & 63u) == 0) {
                            if (W::xw_load(&W::lds()->x_err)) { edge_leave(parity ? 28 : 26); return; }
                            if ((spins & 1023u) == 0 && W::clock() - t0 > 4 * W::timeout_ticks()) { edge_leave(parity ? 28 : 26); return; }
                        }
                        W::xw_pause();
                    }
                    rq = edge_fetch(slot, 0);
                    if ((W::rec_get(rq, 0) > next && W::rec_get(rq, 4) == next) || later(rq)) break;
                }
                if (!have_g) wait_cyc += W::tick() - c0;
            }
            if (W::rec_get(rq, 4) != next) { next = W::rec_get(rq, 4); have_g = false; }
            const uint32_t a = W::rec_get(rq, 4);
            const uint32_t rp0 = W::rec_get(rq, 7), rp1 = W::rec_get(rq, 8), rp2 = W::rec_get(rq, 9), rp3 = W::rec_get(rq, 10);
            if (have_g && g0 == rp0 && g1 == rp1 && g2 == rp2 && g3 == rp3 && gseq == W::rec_get(rq, 16)) {
                W::cnt_add(&W::lds()->cnt.guess_used, 1);
            } else {
                W::cnt_add(have_g ? &W::lds()->cnt.guess_wrong : &W::lds()->cnt.guess_late, 1);
                // kRepPf bytes in front of each rep distance and at the position
                const typename W::RepPf rpf = W::rep_prefetch(G.in, g.n, a, rp0, rp1, rp2, rp3);
                uint32_t rep_len[4];
                W::rep_lengths(rpf, rep_len);
                list(a, rp0, rp1, rp2, rp3, rep_len[0], rep_len[1], rep_len[2], rep_len[3], W::rec_get(rq, 14), W::rec_get(rq, 15), W::rec_get(rq, 3) & 0xFFFFu);
#ifdef NLZM_SIM_COUNT
                __atomic_fetch_add(&g_dbg[6], 1, __ATOMIC_RELAXED);
#endif
            }
#ifdef NLZM_SIM_COUNT
            __atomic_fetch_add(&g_dbg[7], 1, __ATOMIC_RELAXED);
#endif
            have_g = false;
            next += 2;
            W::lds()->eb[(a & 3u) * 16 + 15] = next - 1;
            W::xw_store(&W::lds()->eb[(a & 3u) * 16 + 14], next - 1);    // a is listed (word 15: for the guess at least)
#ifdef NLZM_PROFILE
            lat_sum += (uint32_t)((uint32_t)W::tick() - W::rec_get(rq, 12));
#endif
        }
    }

    // E2: relaxes the listed edges of each node in the reference's order
    NLZM_HD void run_edge_apply(uint32_t a_first)
    {
        err = 0; err_info0 = 0; wait_cyc = 0; role_t0 = W::tick(); cmp_acc = 0; wait_rep = 0;
#ifdef NLZM_PROFILE
        for (int k = 0; k < 16; k++) prof[k] = 0;
#endif
        uint32_t next = a_first, end_p = 1;
        for (;;) {
            typename W::Rec rq;
            if (!edge_wait(rq, next, true)) { edge_leave(22); return; }
#ifdef NLZM_PROFILE
            lat_sum2 += (uint32_t)((uint32_t)W::tick() - W::rec_get(rq, 12));
#endif
            prof_start();
            const uint32_t slot = next & 1u;
            const uint32_t p = W::rec_get(rq, 5);
            const uint32_t r0 = W::rec_get(rq, 7), r1 = W::rec_get(rq, 8), r2 = W::rec_get(rq, 9), r3 = W::rec_get(rq, 10);
            const uint32_t max_len = W::rec_get(rq, 11), cost_p = W::rec_get(rq, 6);
            if (p == 0) end_p = 1;
            uint32_t checked = 0, myri = 4;
            prof_mark(0);
            if (max_len) {
                open_nodes(end_p, max_len + p);                     // :1550-1554
                prof_mark(1);
                // Dict edge then Rep edge to the same node (:1568 then :1586), strict '<' both times, folded into one
                // compare-and-store; the targets of different lanes are distinct nodes
                uint32_t step = (max_len - kMatchMin) >> 4;
                step += step == 0;
                auto apply = [&](uint32_t k) {
                    // the lane's target node follows from max_len alone (tl_k = max_len - k*step): its cost is read
                    // together with the listed entry
                    if (k * step > max_len - kMatchMin) return;
                    const uint32_t *o = W::lds()->ea + ((next & 3u) * 64 + k) * 4;
                    const uint32_t tl = max_len - k * step, np = p + tl;
                    // (all five words are read before the first use: one LDS round trip, not two)
                    uint32_t best = W::lds()->node_cost[np], sel = 0;
                    const uint32_t cd = o[0], cr = o[1], d = o[2], w = o[3];
                    const bool listed = (w >> 12) != 0;
                    const uint32_t ri = !listed ? 4u : (r0 == d ? 0u : (r1 == d ? 1u : (r2 == d ? 2u : (r3 == d ? 3u : 4u))));
                    const uint32_t ca = listed ? cost_p + cd : kNone, cb = ri < 4 ? cost_p + cr : kNone;
                    myri = ri;
                    if (W::width() == 1) checked |= ri < 4 ? 1u << ri : 0u;
                    if (ca < best) { best = ca; sel = 1; }
                    if (cb < best) { best = cb; sel = 2; }          // cb is kNone when the distance is no rep
                    if (sel) {
                        W::lds()->node_cost[np] = best;
                        W::lds()->node_delta[np] = sel == 1 ? d : ri;
                        W::lds()->node_link[np] = pack_link(p, tl, sel);
                        uint32_t *dst = W::lds()->reps + (np & 511) * 4;        // RepModel::Add (:1160-1171)
                        uint32_t w0 = d, w1 = r0, w2 = r1, w3 = r2;
                        if (ri < 4) { w0 = r0; w1 = r1; w2 = r2; w3 = r3; }
                        dst[0] = w0; dst[1] = w1; dst[2] = w2; dst[3] = w3;
                    }
                };
                if (W::width() == 1) { for (uint32_t k = 0; k < 64; k++) apply(k); }
                else apply(W::lane());
                if (W::width() != 1) {                              // rep indices a sampled edge has met (:1573-1584)
                    const uint32_t m = myri;
                    if (W::mask64([=](uint32_t) { return m < 4; }))     // (usually none)
                        checked = (W::mask64([=](uint32_t) { return m == 0; }) ? 1u : 0u) | (W::mask64([=](uint32_t) { return m == 1; }) ? 2u : 0u) |
                                  (W::mask64([=](uint32_t) { return m == 2; }) ? 4u : 0u) | (W::mask64([=](uint32_t) { return m == 3; }) ? 8u : 0u);
                }
                W::sync();
            }
            // the rep probes come after the sampled edges (:1598); rep indices a sampled edge has met are skipped
            // (count and list with one read; if the count was not there yet, the list is read again after the wait)
            prof_mark(2);
            // If the parser's guess of this node's rep set was right (the post says so), the list made for the guess is
            // this node's list: its count word is up before the rep-list wave has seen the post.
            const uint32_t *cw = &W::lds()->eb[(next & 3u) * 16 + 14 + (W::rec_get(rq, 17) & 1u)];
            auto rb_fetch = [=]() { return W::rec_load_fn32([=](uint32_t i) { return cw[i == 0 ? 0 : (int)((i - 1) & 15u) - 14 - (int)(W::rec_get(rq, 17) & 1u)]; }); };
            typename W::Rec rb0 = rb_fetch();
            if (W::rec_get(rb0, 0) < next + 1) {
                const unsigned long long w0 = wait_cyc;
                if (!wait_ge(cw, next + 1)) { edge_leave(22); return; }
                wait_rep += wait_cyc - w0;
                rb0 = rb_fetch();
            }
            // (word k of the rep list is in lane k + 1)
            prof_mark(3);
            if (checked != 15) {
                // rep k's word sits in lane 1 + 2k: bytes compared (:1605) summed and the probes long enough to relax
                // found without a branch per rep; relaxing one is the rare case
                auto live = [=](uint32_t i, uint32_t w) { return (i & 1u) && i < 8 && !((checked >> (i >> 1)) & 1u) && (w >> 31); };
                const uint32_t csum = W::rec_sum_odd4(rb0, [=](uint32_t i, uint32_t w) { return live(i, w) ? (w & 0xFFFFu) + 1 : 0u; });
                uint32_t rm = (uint32_t)W::rec_mask(rb0, [=](uint32_t i, uint32_t w) { return live(i, w) && ((w >> 30) & 1u); });
                while (rm) {
                    const uint32_t i = (uint32_t)__builtin_ctz(rm), k = i >> 1;
                    rm &= rm - 1;
                    const uint32_t l = W::rec_get(rb0, i) & 0xFFFFu, d = k == 0 ? r0 : (k == 1 ? r1 : (k == 2 ? r2 : r3));
                    open_nodes(end_p, l + p);
                    W::sync();
                    relax(p, p + l, cost_p + W::rec_get(rb0, i + 1), 0, 2, l, k, r0, r1, r2, r3, d);   // wave-uniform
                    W::sync();
                }
                cmp_acc += csum;                                    // (to the counters when the wave leaves)
            }
            prof_mark(4);
            W::lds()->sq_res[slot] = end_p;
            W::sync();
            next++;
            W::xw_store(&W::lds()->x_sdone, next);
            prof_mark(5);
#ifdef NLZM_PROFILE
            lat_sum += (uint32_t)((uint32_t)W::tick() - W::rec_get(rq, 12));
#endif
        }
    }

    // =========================== parser wave: parse + emit ==================================
    // table entry `tl` of the position whose hand-off header is `hdr`
    NLZM_HD uint32_t tab(const uint32_t *e, uint32_t hdr, uint32_t tl) const
    {
        return tl < 64 ? e[tl] : W::lds()->er_long[((hdr >> 16) - 1) * (kMatchMax + 8) + tl];
    }

    // one parse segment (:1464-1651); returns end_p, leaves the path in cmdlist
    NLZM_HD uint32_t parse_segment(uint32_t seg_q, uint32_t seg_a, uint32_t max_parse, uint32_t &ncmds)
    {
        max_parse = umin(max_parse, kParseMax);
        W::lds()->node_cost[0] = 0; W::lds()->node_link[0] = 0x1FFF;
        W::lds()->node_cost[1] = kNone; W::lds()->node_link[1] = pack_link(0, 0, 0);
        for (int k = 0; k < 4; k++) { W::lds()->reps[k] = rep[k]; W::lds()->reps[4 + k] = rep[k]; }
        uint32_t p = 0, end_p = 1;
        uint32_t settled = 0, pend_long = 0;            // nodes whose edges are in / long hand-off slots they still hold
        // edges of every node < upto are relaxed: take over end_p, give the hand-off slots back
        // (with nodes: the same read brings nodes upto+1 and upto+2 as they stand then -- cost and rep set in lanes 3..7
        //  and 8..12 -- which is all the parser needs of them for its next step)
        typename W::Rec sr = nrec_none();
        auto settle = [&](uint32_t upto, bool publish, bool nodes = false) -> bool {
            if (!nodes && settled >= upto) return true;
            // count and end_p words with one read (the count in the lowest lane)
            auto fetch = [=]() {
                return W::rec_load_fn32([=](uint32_t i) {
                    const uint32_t n = upto + (i >= 8 ? 2u : 1u), j = i >= 8 ? i - 8 : i - 3;
                    return i == 0 ? W::lds()->x_sdone : (i < 3 ? W::lds()->sq_res[(i - 1) & 1u] :
                           (j == 0 ? W::lds()->node_cost[n] : W::lds()->reps[(n & 511) * 4 + ((j - 1) & 3)]));
                });
            };
            sr = fetch();
            if (NLZM_RARE(W::rec_get(sr, 0) < seg_a + upto)) {
                const unsigned long long t0 = W::clock(), c0 = W::tick();
                uint32_t spins = 0;
                for (;;) {
                    sr = fetch();
                    if (W::rec_get(sr, 0) >= seg_a + upto) break;
                    if ((++spins & 63u) == 0) {
                        if (W::xw_load(&W::lds()->x_err)) { err = kErrInternal + 100; return false; }
                        if ((spins & 1023u) == 0 && W::clock() - t0 > W::timeout_ticks()) { fail(kErrTimeout, seg_a + upto); return false; }
                    }
                    W::xw_pause();
                }
                wait_cyc += W::tick() - c0;
            }
            if (settled >= upto) return true;                       // (read for the nodes only)
            const uint32_t se = W::rec_get(sr, 1 + ((seg_a + upto - 1) & 1u));
            if (se > end_p) { end_p = se; W::xw_store(&W::lds()->x_bcover, seg_a + end_p); }
            settled = upto;
            if (NLZM_RARE(pend_long != 0)) { b_long += pend_long; pend_long = 0; W::xw_store(&W::lds()->x_long_free, kErLong + b_long); }
            if (publish) W::xw_store(&W::lds()->x_bpos, seg_a + upto);
            return true;
        };
        // node p: cost and rep set with one LDS instruction
        // (read for p and, as it stands, for p+1: the guess below)
        auto node_read = [&](uint32_t n) {
            return W::rec_load_fn([=](uint32_t i) { return i == 0 ? W::lds()->node_cost[n] : W::lds()->reps[(n & 511) * 4 + ((i - 1) & 3)]; });
        };
        // hand-off header and input byte of position x (kept: the position after a segment's last node is the next
        // segment's first)
        auto table_head = [&](uint32_t x, uint32_t &hdr, uint32_t &lit) -> bool {
            if (th_a != x) {
                // the table wave's position (lane 0, read first) and the two header words with one read
                auto fetch = [=]() {
                    return W::rec_load_fn([=](uint32_t i) { return i == 0 ? W::lds()->x_apos : W::lds()->er_tab[(x & (kEr - 1)) * 64 + ((i - 1) & 1u)]; });
                };
                typename W::Rec hrec = fetch();
                if (NLZM_RARE(W::rec_get(hrec, 0) < x + 1)) {
                    if (!wait_ge(&W::lds()->x_apos, x + 1)) return false;
                    hrec = fetch();
                }
                th_a = x; th_hdr = W::rec_get(hrec, 1); th_lit = W::rec_get(hrec, 2);
            }
            hdr = th_hdr; lit = th_lit;
            return true;
        };
        // the segment for the edge-list wave, which lists ahead: prices first (:1491, :1567 use the model as it is now)
        seg_tables();
        seg_seq++;
        W::lds()->seginfo[1] = seg_a; W::lds()->seginfo[2] = max_parse;
        W::sync();
        W::xw_store(&W::lds()->seginfo[0], seg_seq);
        // the command prices of this segment, for the list waves
        for (uint32_t i = W::lane(); i < 4; i += W::width()) W::lds()->post[i >> 1][2 + (i & 1u)] = ((const uint32_t *)W::lds()->price)[kCtxCmd * 8 + (i & 1u)];
        W::sync();
        W::xw_store(&W::lds()->x_bcover, seg_a + 1);
        typename W::Rec nrec = node_read(0), nrec1 = node_read(1);
        uint32_t hdr = 0, lit = 0, cst_lit = 0;
        // the rep set guessed for the node about to be posted (lanes 1..4, as in a node record)
        typename W::Rec gs = nrec;
        uint32_t gs_valid = 0;
        if (table_head(seg_a, hdr, lit)) cst_lit = price_literal(lit);
        prof_mark(0);                                               // (segment set-up)
        while (!err) {                                              // p < end_p
            const uint32_t q = seg_q + p, a = seg_a + p;
            n_pos++;
            const uint32_t cost_p = W::rec_get(nrec, 0);
            uint32_t rp[4];
            for (int k = 0; k < 4; k++) rp[k] = W::rec_get(nrec, 1 + k);
            uint32_t max_len = umin(hdr & 0xFFFFu, max_parse - p);  // :1545-1548
            if (max_len < kMatchMin) max_len = 0;
            {   // node p is final: its match and rep edges go to the edge waves (slot a & 1 is free: node p-2 is settled)
                // (the request is put together in one register, word k in lane k, and stored with one instruction;
                //  words 2..6 are the node record's lanes 0..4 as they are)
                typename W::Rec r = W::rec_shift2(nrec);
                r = W::rec_set(r, 0, a); r = W::rec_set(r, 1, p);
                r = W::rec_set(r, 7, max_len); r = W::rec_set(r, 9, hdr); r = W::rec_set(r, 10, q);
                r = W::rec_set(r, 11, umin(max_parse - p, kMatchMax)); r = W::rec_set(r, 12, seg_seq);
                // the guess posted for this node was right: the rep list made for it is this node's list
                r = W::rec_set(r, 13, gs_valid & W::rec_eq4(gs, nrec));
#ifdef NLZM_PROFILE
                r = W::rec_set(r, 8, (uint32_t)W::tick());
#endif
                W::rec_store_n(W::lds()->post[a & 1u] + 4, r, 14);
                W::sync();
                W::xw_store(&W::lds()->post[a & 1u][0], a + 1);
            }
            if (max_len && p + max_len > end_p) { end_p = p + max_len; W::xw_store(&W::lds()->x_bcover, seg_a + end_p); }
            prof_mark(1);                                           // (post)
            const uint32_t pend_long_next = (hdr >> 16) ? 1u : 0u;
            // while the edge waves work on this node: the table header and the literal price of the next position
            uint32_t hdr_n = 0, lit_n = 0, cst_n = 0;
            gs_valid = 0;
            if (!NLZM_RARE(p + 1 >= max_parse)) {
                // A guess of node p+1's rep set for its rep-list wave, so that the bytes are in when the node is
                // final: the literal edge of p would win it (then the set is p's), or it keeps what it has.  Only
                // a length-2 edge of p-1 can still prove the guess wrong (then the wave measures again).
                const bool lit_wins = cost_p + cst_lit < W::rec_get(nrec1, 0);
                uint32_t *sp = W::lds()->post[(a + 1) & 1u] + 24;
                gs = W::rec_sel(lit_wins, nrec, nrec1);             // lanes 1..4: the set
                gs_valid = 1;
                gs = W::rec_set(gs, 0, a + 1);
                gs = W::rec_set(gs, 5, q + 1); gs = W::rec_set(gs, 6, umin(max_parse - p - 1, kMatchMax));    // (what the post of p+1 will say)
                gs = W::rec_set(gs, 7, seg_seq);
                W::rec_store_n(sp, gs, 8);
                W::sync();
                W::xw_store(&W::lds()->post[(a + 1) & 1u][23], a + 2);
                prof_mark(2);                                       // (guess)
                if (NLZM_RARE(!table_head(a + 1, hdr_n, lit_n))) break;
                cst_n = price_literal(lit_n);
            }
            prof_mark(7);
            // node p+1 takes the edges of p-1 before the literal edge of p (the reference's order, strict '<')
            if (NLZM_RARE(!settle(p, true, true))) break;
            pend_long += pend_long_next;
            prof_mark(8);
            // literal edge (:1490-1499)
            const bool lit_edge = W::rec_get(sr, 3) > cost_p + cst_lit;
            if (lit_edge) {
                W::lds()->node_cost[p + 1] = cost_p + cst_lit;
                W::lds()->node_delta[p + 1] = lit;                  // the byte itself, for the emitter
                W::lds()->node_link[p + 1] = pack_link(p, 0, 0);
                W::rec_store4(W::lds()->reps + ((p + 1) & 511) * 4 - 1, nrec);     // lanes 1..4: node p's set
            }
            // the next node: what the literal edge made of it, or what it was when the edges of its last predecessor
            // were in; the one after it as it stood then (for the guess)
            nrec = W::rec_sel(lit_edge, W::rec_set(nrec, 0, cost_p + cst_lit), W::rec_shl3(sr));
            nrec1 = W::rec_shl8(sr);
            ++p;
            W::sync();
            if (NLZM_RARE(p >= end_p)) {
                // only a rep probe of the node just posted can still extend the segment
                if (!settle(p, false)) break;
                if (p >= end_p) break;
                W::xw_store(&W::lds()->x_bpos, seg_a + p);
            }
            hdr = hdr_n; lit = lit_n; cst_lit = cst_n;
            prof_mark(9);
        }
        if (!err) settle(p, false);
        // a segment that ends here is announced BEFORE the position count
        W::xw_store(&W::lds()->x_bseg, seg_a + p);
        W::xw_store(&W::lds()->x_bpos, seg_a + p);
        // backtrack (:1633-1650): collect the node indices of the path, end first
        uint32_t n = 0, cur = p;
        while (cur != 0 && !err) {
            W::lds()->cmdlist[n] = (uint16_t)cur;
            n++;
            cur = W::uni(W::lds()->node_link[cur]) & 0x1FFF;
        }
        W::sync();
        prof_mark(10);
        ncmds = n;
        W::cnt_add(&W::lds()->cnt.segments, 1);
        return end_p;
    }

    NLZM_HD void capture(uint32_t a)
    {
        if (!G.cap_words) return;
        if (a < G.cap_lo || a >= G.cap_hi) return;
        // record {pos, max_len, delta[2..max_len]}
        unsigned long long used = *G.cap_used;
        const unsigned long long need = 2 + (mt_max >= 2 ? mt_max - 1 : 0);
        if (used + need > G.cap_cap) { fail(kErrCapture, a); return; }
        W::sync_global();
        W::sync();
        if (W::lane() == 0) { G.cap_words[used] = a; G.cap_words[used + 1] = mt_max; }
        for (uint32_t i = 2 + W::lane(); i <= mt_max; i += W::width()) G.cap_words[used + i] = mt(i);
        if (W::lane() == 0) *G.cap_used = used + need;
        W::sync_global();
    }

    // one chunk = one frame (:1782-1886)
    NLZM_HD void run_chunk_parser(uint32_t ci)
    {
        const unsigned long long chunk_abs = (unsigned long long)ci * g.chunk_size;
        const unsigned long long remain = g.n - chunk_abs;
        const uint32_t chunk_read = (uint32_t)(remain < g.feed ? remain : g.feed);
        const uint32_t p_end = umin(g.chunk_size, chunk_read);
        const uint32_t W2 = 2u * (g.wmask + 1);

        // frame.Init (:534-550)
        fsyms = G.syms + (unsigned long long)(ci - G.chunk0) * G.syms_stride;
        fbits = G.bits + (unsigned long long)(ci - G.chunk0) * G.bits_stride;
        nsyms = 0; nbits = 0; word = 0; word_bits = 0; num_ops = 0; nq = 0;
        if (chunk_abs - base >= W2) base += g.wmask + 1;            // same rebase schedule as the finder wave (:1786)
        const uint32_t chunk_q = (uint32_t)(chunk_abs - base);
        counts_zero();

        uint32_t p = 0;
        while (p < p_end && !err) {
            uint32_t ncmds = 0;
            parse_segment(chunk_q + p, (uint32_t)chunk_abs + p, p_end - p, ncmds);
            if (err) break;
            for (uint32_t k = ncmds; k-- > 0;) {                    // :1809-1843
                const uint32_t node = W::uni(W::lds()->cmdlist[k]);
                const uint32_t link = W::uni(W::lds()->node_link[node]);
                const uint32_t cmd = link >> 22, len = (link >> 13) & 0x1FF;
                if (cmd == 0) { emit_literal(W::uni(W::lds()->node_delta[node])); p += 1; }
                else if (cmd == 1) { emit_match(W::uni(W::lds()->node_delta[node]), len); p += len; }
                else { emit_rep(W::uni(W::lds()->node_delta[node]), len); p += len; }
            }
            prof_mark(11);
            if (nsyms + 16 > G.syms_stride || nbits + 64 > G.bits_stride) fail(kErrFrameOverflow, ci);
        }
        counts_flush();
        // bit pad of Flush (:591-597)
        W::cnt_add(&W::lds()->cnt.rans_syms, nsyms); W::cnt_add(&W::lds()->cnt.bit_ops, num_ops - nsyms); W::cnt_add(&W::lds()->cnt.frames, 1);
        for (int i = 0; i < 4; i++) {
            fbits[nbits] = (uint8_t)(word >> 24);
            nbits++; word <<= 8;
        }
        if (W::lane() == 0) {
            FrameMeta &fm = G.fmeta[ci - G.chunk0];
            fm.nsyms = nsyms; fm.nbits_bytes = nbits; fm.num_ops = num_ops; fm.out_len = 0;
        }
    }

    NLZM_HD void run_parser(uint32_t c0, uint32_t c1)
    {
        Persist *P = G.persist;
        for (uint32_t i = W::lane(); i < kNumCtx * kCdfStride; i += W::width()) W::lds()->cdf[i] = P->cdf[i];
        for (uint32_t i = W::lane(); i < 256; i += W::width()) W::lds()->lut[i] = log2_lut_entry(i);
        W::sync();
        for (uint32_t i = W::lane(); i < kNumCtx * 16; i += W::width()) {
            const uint32_t ctx = i >> 4, y = i & 15;
            const uint16_t *cell = W::lds()->cdf + ctx * kCdfStride;
            W::lds()->price[i] = (y < ctx_nsyms(ctx)) ? W::lds()->lut[((uint32_t)cell[y + 1] - (uint32_t)cell[y]) >> 6] : 0;
        }
        for (int k = 0; k < 4; k++) rep[k] = W::uni(P->rep[k]);
        base = ((unsigned long long)W::uni((uint32_t)(P->reb_base >> 32)) << 32) | W::uni((uint32_t)P->reb_base);
        err = W::uni(P->error); err_info0 = 0;
        b_long = 0; seg_tab_dirty = true; th_a = kNone; th_hdr = 0; th_lit = 0; seg_seq = 0;
        wait_cyc = 0; role_t0 = W::tick();
        counts_zero();
#ifdef NLZM_PROFILE
        for (int k = 0; k < 16; k++) prof[k] = 0;
        prof_start();
#endif
        W::sync();
        uint32_t ci = c0;
        for (; ci < c1 && !err; ci++) run_chunk_parser(ci);
        W::xw_store(&W::lds()->post[0][0], kNone); W::xw_store(&W::lds()->post[1][0], kNone);    // the edge waves may leave
        const unsigned long long role_t1 = W::tick();
        // the finder wave has stored its part of the state and its counters
        for (uint32_t spins = 0; W::xw_load(&W::lds()->x_adone) < 2 && spins < (1u << 28); spins++) W::xw_pause();
        W::sync();
        for (uint32_t i = W::lane(); i < kNumCtx * kCdfStride; i += W::width()) P->cdf[i] = W::lds()->cdf[i];
        if (W::lane() == 0) {
            for (int k = 0; k < 4; k++) P->rep[k] = rep[k];
            P->next_chunk = ci;
            P->prof[20] += wait_cyc; P->prof[21] += role_t1 - role_t0;
            if (err && err < 100) { P->error = err; P->error_info[0] = err_info0; P->error_info[1] = 2; }
            const uint32_t xe = W::xw_load(&W::lds()->x_err);
            if (xe && !P->error) P->error = xe;
            if ((err || xe) && G.abort_word) W::st_agent(G.abort_word, 1u);
#ifdef NLZM_PROFILE
            for (int k = 7; k < 12; k++) P->prof[k] += prof[k];
            P->prof[13] += prof[13];
            P->prof[47] += prof[0]; P->prof[54] += prof[1]; P->prof[55] += prof[2];
#endif
            unsigned long long *dst = (unsigned long long *)&P->cnt;
            const unsigned long long *src = (const unsigned long long *)&W::lds()->cnt;
            for (uint32_t i = 0; i < sizeof(Counters) / 8; i++) dst[i] += src[i];
        }
    }

    // shared start-up, run by ONE wave before any role starts
    NLZM_HD static void init_shared(const Globals &G, uint32_t a0)
    {
        for (uint32_t i = W::lane(); i < sizeof(Counters) / 8; i += W::width()) ((unsigned long long *)&W::lds()->cnt)[i] = 0;
        if (W::lane() == 0) {
            W::lds()->x_apos = a0; W::lds()->x_bpos = a0; W::lds()->x_bseg = a0; W::lds()->x_bcover = a0 + 1;
            W::lds()->x_long_free = kErLong; W::lds()->x_err = 0; W::lds()->x_adone = 0;
            W::lds()->x_cpos = 0; W::lds()->x_tpos = 0; W::lds()->post[0][0] = a0; W::lds()->post[1][0] = a0; W::lds()->x_sdone = a0;
            W::lds()->post[0][20] = a0; W::lds()->post[0][21] = a0; W::lds()->post[0][22] = a0;
            for (uint32_t i = 0; i < 8; i++) { W::lds()->ea_tag[i] = 0; W::lds()->seginfo[i] = 0; }
            W::lds()->post[0][23] = a0; W::lds()->post[1][23] = a0;
            for (uint32_t i = 0; i < 4; i++) { W::lds()->eb[i * 16 + 14] = a0; W::lds()->eb[i * 16 + 15] = a0; }
        }
        (void)G;
    }
};

}  // namespace nlzm
