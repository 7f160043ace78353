// nlzm_v2.h -- the serial half of the compress path as three pipelined stages whose lanes are POSITIONS (finder,
// table) or EDGES (parser), one workgroup (CU) per stage, hand-off through small rings in HBM/L2:
//
//   role_finder  HT2/HT3 (tables in LDS), RK256, the nice decision and the decisions for the BT4 worker lanes
//                (:1501-1543 minus BT4).  A block of up to 64 consecutive positions is evaluated by the lanes in
//                parallel under a PREDICTED set of finder calls; a cross-lane scan verifies the prediction and the
//                block is cut at the first position that the reference would have treated differently.
//   role_table   the match table as mt_carry holds it (:746-752, :823-852, :1543), kept as its Pareto front
//                (distance falls as length falls); the front of every position of a block comes out of ONE
//                parallel prefix scan over the lanes; then the sampled lengths of :1558-1562 are listed.
//   role_parser  parse_table's graph (:1464-1651) as a wavefront: two nodes per step, lanes = their out-edges,
//                relaxation by 64-bit LDS atomic min on (cost, source, rank) -- the lexicographic order IS the
//                reference's "first candidate wins" under strict '>' (:1492, :1568, :1586, :1616); backtrack,
//                model_encode_* (:1274-1367, :1428-1439), frame symbol/bit streams.
//
// Written against xw.h: the same source is the gfx950 kernel and, under -DNLZM_SIM, the CPU simulation
// that tests/host_sim/sim2.cpp checks against the oracle.  All file:line cites are NLZM.cpp.
#pragma once

#include "nlzm_core.h"
#include "xw.h"

namespace nlzm {
namespace v2 {

// ------------------------------------------------------------------------------------------------
// hand-off between the stages (HBM; payload stored sc1 and drained before the progress word, R1 form)
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kFtRing = 1u << 14;          // finder -> table: positions in flight
constexpr uint32_t kFtStride = 16;              // words per position
constexpr uint32_t kTpRing = 1u << 13;          // table -> parser
constexpr uint32_t kMaxEdges = 32;              // sampled lengths per position (:1558-1560: at most 32, at max_len 33)
constexpr uint32_t kTpStride = 2 + 2 + 2 * kMaxEdges;   // header (2), the mask of samples with a new distance (2), edges (distance, attributes)
constexpr uint32_t kTpUniq = 2, kTpEdges = 4;           // (header + mask, and every two edges, are ONE 16-byte store of the table stage)
constexpr uint32_t kFrontMax = 264;             // entries of a front (one per length at most)
constexpr uint32_t kTfStride = 2 * kFrontMax;   // words per position in the front ring (end, distance)

// finder -> table record:
//   w0        number of pairs in w2.. (bits 0..2) | BT4 ran: take the worker record's list (bit 3) | top pair present (bit 4)
//             | input byte << 8
//   w2..w13   up to six (distance, length) pairs: HT2, HT3 x2, RK256 carried / new, a former top entry that has stopped growing
//   w14, w15  the top entry (distance, length) while it is still growing with the position (:1503-1512)
constexpr uint32_t kFtBt = 8u, kFtTop = 16u;
// table -> parser record:
//   w0        edges (bits 0..5) | input byte << 8 | table length (mt.max_len) << 16
//   w1        entries of the front in the front ring
//   w2, w3    the mask of sampled lengths that bring a new distance
//   w4..      per sampled length: distance, length | length value << 9 | distance slot << 18 | extra bits << 24 | valid << 31
struct Hx {                                     // progress words, one 128-byte line each
    alignas(128) uint32_t f_pos;                // finder: records of positions < f_pos are written
    alignas(128) uint32_t t_pos;                // table: records of positions < t_pos are consumed
    alignas(128) uint32_t t_out;                // table: parser records of positions < t_out are written
    alignas(128) uint32_t p_pos;                // parser: records of positions < p_pos are consumed
    alignas(128) unsigned long long p_seg;      // parser: segment start << 32 | positions below this lie in that segment
    // helper parser (HelpBox below)
    alignas(128) uint32_t h_job;                // parser: number of the job in the HelpBox (0: none yet, kHelpExit: the launch is over); every helper takes it
    struct alignas(128) HelpWords {             // per helper, a line each:
        alignas(128) uint32_t verdict;          //   whoever drops the helper's job (the parser stage, or the helper in front of it): job << 2 | 2
        alignas(128) unsigned long long prog;   //   helper: job << 32 | nodes it knows to be inside the segment << 16 | nodes final (HelpBox arrays valid below)
        alignas(128) uint32_t state;            //   helper: job << 2 | 1 done (seg_len, end_rep valid), 2 given up
    } hw[3];
    alignas(128) uint32_t ext_cur;              // worker lanes: extension blocks of pair lists taken in this launch (Globals::bt_ext_cur)
    alignas(128) uint32_t err;                  // any stage: nonzero -> every stage leaves (the FIRST code stays: raise())
    uint32_t err_info[7];                       // of the stage that raised it: stage (11 finder, 12 table, 13 parser), wait site, position, what it saw
    alignas(128) uint32_t dbg[4][32];           // per stage (0 finder, 1 table, 2 parser): where it was when it left because of an error
};
enum : uint32_t { kStFinder = 11, kStTable = 12, kStParser = 13 };
// What the host looks at after a launch, copied aside on the device when the next launch is already queued behind it (block mode)
struct RoundSnap { uint32_t error, next_chunk, aborted, pad; Hx hx; };

// ---- helper parser: a second parser workgroup that parses the BACK of a long segment while the parser stage parses its front.
// Under the same prices two parses of a segment differ only in where they started, and the recurrence forgets its start: a parse
// begun at a node s0 (help_start) as if the segment began there (cost 0, the rep set of the segment start) has, a few hundred nodes on,
// the true state of every node up to ONE cost offset.  The parser stage checks exactly that when it gets to the first block border
// behind s0 + kHelpWarm: the 264 nodes in front of the border are all the sources an edge into a later node can have
// (:737); if on them the helper's costs minus its own are one constant and the rep sets are equal, every candidate of every later
// node is shifted by that constant in the helper's parse -- same winners under strict '>' and the (source, rank) order, same rep
// sets -- and it takes the helper's nodes over instead of computing them.  Otherwise it goes on by itself.  Measured on the CPU
// (oracle/stale_probe.c, profiles/r05_parser_go_nogo.txt): 85 - 92 % of such frontiers agree 576 nodes after the start.
// There are kHelpers of them, each starting further back; a helper is to the helper behind it what the parser stage is to the first: it
// compares that helper's frontier with its own states (true up to ITS offset, if it is itself taken over) and takes its nodes over, so that
// what the parser stage takes from the first helper is the rest of the segment.
#ifndef NLZM_HELPERS
#define NLZM_HELPERS 2
#endif
constexpr uint32_t kHelpers = NLZM_HELPERS;     // (1, 2 or 3.  Three, measured in round 6: the parser stage's busy cycles 305 -> 266 per position, the headline 6.35 -> 6.40 MB/s -- the finder stage
                                                //  bounds the stream there; not the default, one CU more for 0.7 %: DESIGN.md section 13)
static_assert(kHelpers >= 1 && kHelpers <= 3, "Hx::hw holds three helpers' words");
// helper k's first node (a segment that is cut at 4,096: one helper -- the front 2,400 nodes stay with the parser stage, the helper has
// 1,792 .. 4,096; two -- 1,900 / 1,280 .. 3,100 / 2,496 .. 4,096; the helpers further back wait for their records longer)
NLZM_HD uint32_t help_start(uint32_t k) { return kHelpers == 1 ? 1792u : (kHelpers == 2 ? (k == 0 ? 1280u : 2496u) : 880u * (k + 1)); }
constexpr uint32_t kHelpWarm = 576;             // nodes a helper has behind it where the frontiers are compared
constexpr uint32_t kHelpExit = 0xFFFFFFFFu;
struct HelpBox {
    // parser -> helper: the job (valid once Hx::h_job carries its number)
    uint32_t seg_a, chunk_left, base, pad0, rep[4];
    alignas(16) uint16_t price[kNumCtx * 16];
    alignas(16) uint16_t len_price[kMatchMax + 8];
    alignas(16) uint16_t slot_price[4 * 64];
    // helper -> parser: per node (valid below Hx::h_prog's count): the emitter's words, the cost, the rep set, the bytes the node's final probes compared
    alignas(16) uint32_t link[kParseMax + 2], delta[kParseMax + 2], cost[kParseMax + 2], rep_of[(kParseMax + 2) * 4], cmpw[4][kParseMax + 2];
    uint32_t seg_len, end_rep[4];               // (valid with Hx::h_state "done")
    // the helper's own: records it has re-listed for the segment's forced cut (position a at [(a & 511) * kTpStride]).  The ring's records
    // are the parser stage's to change -- the helper runs ahead of it on a guess, and a record cut short for a segment that then ends
    // elsewhere (or whose job is dropped) would be the next segment's wrong table.
    alignas(16) uint32_t relist[512 * kTpStride];
};
constexpr unsigned long long kStagedOwn = 0x80000000ull;   // helper: bit of a staged record's header -- its edges beyond the staged sixteen are in HelpBox::relist

struct GlobalsV2 {
    uint32_t *ft;                               // [kFtRing][kFtStride]
    uint32_t *tp;                               // [kTpRing][kTpStride]
    uint32_t *tf;                               // [kTpRing][kTfStride]
    Hx *hx;
    uint32_t *state;                            // StateV2 (survives between launches)
    HelpBox *hb;                                // [kHelpers] (the job's parameters and prices in the first); null: no helper parsers
};

// stage state that survives between launches (HBM)
struct StateV2 {
    // finder
    uint32_t reach, reach_d;                    // largest end (absolute) of any closed table entry so far, and the smallest distance of an entry that ends there
    uint32_t s_active, s_d, s_end, s_seen;      // growing top entry: distance, first mismatch (kNone: not found yet), verified up to
    uint32_t prev_nice, seg_s;
    // table: the front after the last position of the launch
    uint32_t front_n;
    uint32_t front[2 * kFrontMax];
    uint32_t tb_wide[2];                        // the table stage's shape (0: narrow fronts, many waves; 1: wide fronts, fewer waves): launch k reads slot k & 1 and
                                                // leaves the next launch's in the other (a wave that enters the kernel late must not meet the word its launch writes)
};

constexpr uint32_t kErrV2Front = 40;            // front overflow on the slow path (cannot happen: one entry per length)
constexpr uint32_t kErrV2Promise = 41;          // a position the pre-filter promised not to be nice is
constexpr uint32_t kErrV2Edges = 42;

NLZM_HD uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { return umin(a, umin(b, c)); }

// wave-wide byte compare (MatchLengthSigned without the sign, :854-877): common prefix of in[s..] and in[t..] from `init`,
// at most cap; lanes look at 8 bytes each, 512 per round.  Uniform arguments, uniform result.
XW_FN uint32_t wave_cmp(const uint8_t *in, uint32_t s, uint32_t t, uint32_t init, uint32_t cap)
{
    uint32_t off = init;
    while (off < cap) {
        const uint32_t my = off + xw::lane() * 8;
        uint32_t m = kNone;
        if (my < cap) {
            const unsigned long long d = load64u(in + s + my) ^ load64u(in + t + my);
            if (d) {
                const uint32_t p = my + ((uint32_t)__builtin_ctzll(d) >> 3);
                if (p < cap) m = p;
            }
        }
        const unsigned long long bal = xw::ballot(m != kNone);
        if (bal) return xw::readlane(m, (uint32_t)__builtin_ctzll(bal));
        off += 512;
    }
    return cap;
}

// The first error of a launch is the one reported: later ones (stages that leave because another one failed) do not
// overwrite it.  Called by ONE lane.
XW_FN void raise(Hx *hx, uint32_t code, uint32_t stage, uint32_t site, uint32_t pos, uint32_t x0 = 0, uint32_t x1 = 0)
{
    if (xw::cas_agent(&hx->err, 0u, code) != 0u) return;
    xw::st_agent(&hx->err_info[0], stage); xw::st_agent(&hx->err_info[1], site); xw::st_agent(&hx->err_info[2], pos);
    xw::st_agent(&hx->err_info[3], x0); xw::st_agent(&hx->err_info[4], x1);
}

// bounded wait for a progress word to reach v; false: another stage failed or the wait timed out (err set)
XW_FN bool wait_word_ge(const uint32_t *w, uint32_t v, Hx *hx, uint32_t code)
{
    if ((int32_t)(xw::readfirst(xw::ld_agent(w)) - v) >= 0) { xw::after_poll(); return true; }
    const unsigned long long t0 = xw::clock100();
    uint32_t spins = 0;
    for (;;) {
        if ((int32_t)(xw::readfirst(xw::ld_agent(w)) - v) >= 0) { xw::after_poll(); return true; }
        if ((++spins & 63u) == 0) {
            if (xw::readfirst(xw::ld_agent(&hx->err))) return false;
#ifndef NLZM_SIM
            if (xw::clock100() - t0 > 3000000000ull) {            // 30 s
                if (xw::lane() == 0) raise(hx, kErrTimeout * 100 + code, code == 1 ? kStFinder : (code <= 3 ? kStTable : kStParser), code, v, xw::ld_agent(w));
                return false;
            }
#else
            (void)t0; (void)code;
#endif
        }
        xw::pause();
    }
}

// =================================================================================================
// finder stage
// =================================================================================================
struct FLds {
    uint32_t ht2[4096];                         // :1750
    uint32_t ht3[16384 + 8];                    // :1751 (bucket b uses rows b and b+1, :912: 2^14 + 1 rows are ever touched)
    uint32_t stage[64 * kFtStride];             // the block's records, one per lane
    uint32_t job[64 * 3 * 4];                   // byte compares longer than the lanes' own first look: lane | cand << 8, distance, cap
    uint32_t job_len[64 * 4];
    uint32_t njobs;
    uint32_t last[8192];                        // scratch of a block: the last lane of a bucket class (HT2: bucket; HT3: 4096 + bucket % 4096), + 1
    Counters cnt;
};

struct Finder {
    Geom g;
    Globals G;
    GlobalsV2 V;
    // wave-uniform state
    uint32_t base;              // absolute offset of rebased 0 (:888-891)
    uint32_t reach;             // largest absolute end of a closed table entry
    uint32_t reach_d = kNone;   // ... and the smallest distance among the entries that end there (what a cut-short RK256 entry of the same end is up against)
    uint32_t s_active, s_d, s_end, s_seen;
    uint32_t rk_from, rk_to, rk_len, rk_end;
    uint32_t rk_cut = 0;        // the carried RK256 match is shorter than the bytes agree: the uint16 length parameter ended its compare (:760, :1096)
    uint32_t prev_nice, seg_s;
    uint32_t gap_last = 0;      // the last position that no match found in front of it reaches beyond (where a segment ends unless a rep probe carried it further, :1598-1628), or the chunk's start
    uint32_t n_seg_own = 0, n_seg_wait = 0;     // starts of nice regions whose segment this stage knew itself / had to wait for
    uint32_t t_pos_seen;
    uint32_t err;
    uint32_t dbg_a = 0;         // start of the block being evaluated (error dump)
    unsigned long long n_pos, n_nice, n_unc, n_ht, n_rkp, n_rki, n_cmp, n_blocks, n_cut0, n_cut1, n_cut2, n_cut3, n_cut4, n_cut5;
    unsigned long long n_cmp_lane = 0;          // bytes compared, per lane
    unsigned long long t_wait, t_wait_bt = 0, t_total;
    uint32_t n_so_top = 0, n_so_tie = 0, n_so_won = 0;  // cut-short RK256 entries that became the growing top entry; that ended exactly where another entry ends; and won that
    uint32_t n_late_unc = 0, n_late_other = 0, n_late_hot = 0, n_late_first = 0, n_late_blocks = 0;
    unsigned long long t_f[8] = {};             // profile build: cycles per section of block()
#ifdef NLZM_PROFILE
    XW_FN unsigned long long ptick() const { return xw::tick(); }
#else
    XW_FN unsigned long long ptick() const { return 0; }
#endif

    XW_FN void fail(uint32_t code, uint32_t pos, uint32_t site = 0, uint32_t x0 = 0, uint32_t x1 = 0)
    {
        err = code;
        if (xw::lane() == 0) raise(V.hx, code, kStFinder, site, pos, x0, x1);
    }

    // A block's records are stored at its end; the progress word that covers them may only follow when they have landed (payload, then
    // flag).  Waiting for that at the block's end cost a round trip to memory per block with nothing else to do; the word is now written
    // in the middle of the NEXT block (in front of its wait for the worker lanes' results), where the stores have long landed -- and in
    // front of anything else this stage may wait for, since the stages behind it may be what that wait depends on.
    uint32_t pend_fpos = 0;         // nonzero: records up to here are stored, f_pos does not say so yet
    // the own loads of the block that starts at pf_a0, requested when the block before it ended
    uint32_t pf_a0 = kNone, pf_rkh = 0;
    unsigned long long pf_own0 = 0, pf_own1 = 0;
    uint8_t pf_unc = 0;
    XW_FN void flush_fpos()
    {
        if (!pend_fpos) return;
        xw::drain();
        if (xw::lane() == 0) xw::st_agent(&V.hx->f_pos, pend_fpos);
        pend_fpos = 0;
    }

    // The growing top entry (distance s_d): find where it stops matching, as far as `limit` (:1507-1509).
    XW_FN void slider_look(uint32_t limit)
    {
        while (s_end == kNone && s_seen < limit) {
            const uint32_t cap = umin(limit - s_seen, 512u);
            const uint32_t l = wave_cmp(G.in, s_seen - s_d, s_seen, 0, cap);
            if (l < cap) s_end = s_seen + l; else s_seen += cap;
        }
        if (s_end == kNone && (unsigned long long)s_seen >= g.n) s_end = (uint32_t)g.n;
    }

    // One block: positions [a0, a0 + n) of a chunk whose positions end at a1 and whose lookahead ends at la_end (all
    // absolute).  Returns how many of them are final (>= 1); the records of those are in the ring.
    XW_FN uint32_t block(uint32_t a0, uint32_t n, uint32_t a1, uint32_t la_end)
    {
        FLds *L = xw::lds<FLds>();
        const uint32_t i = xw::lane();
        const bool in_blk = i < n;
        const uint32_t a = a0 + i, q = a - base;
        const uint32_t avail = in_blk ? la_end - a : 0u;
        const uint32_t cap = umin(avail, kMatchMax);                 // :915, :987
        const uint8_t *cur = G.in + a;
        const unsigned long long bi = (unsigned long long)(a - G.batch_a0);

        // ---- the growing top entry: its end must be known as far as this block can see it
        if (s_active) slider_look(umin(a0 + n + kMatchMax, la_end));
        const bool s_known = s_active && s_end != kNone;
        // end of the top entry at this position after carry + extension (:1501-1512)
        const uint32_t s_e = s_active ? (s_known ? umin(s_end, a + cap) : a + cap) : 0u;
        const bool s_sliding = s_active && (!s_known || s_end >= a + cap);   // still growing here (it is the top entry)
        // lanes from jc on see it as an ordinary closed entry that ends at s_end
        const unsigned long long closed_mask = xw::ballot(in_blk && s_active && !s_sliding);
        const uint32_t jc = closed_mask ? (uint32_t)__builtin_ctzll(closed_mask) : 64u;

        const unsigned long long f0 = ptick();
        // ---- predicted finder calls (:1514, :1529) from the state the block starts with
        const uint32_t reach_pred = umax(reach, s_e);
        const bool nice_pred = in_blk && reach_pred >= a + kNice && reach_pred > a;
        if (xw::readlane(nice_pred ? 1u : 0u, 0) && !prev_nice) {
            // a nice region starts here: which segment is a0 in?  The parser knows once it has seen an edge that
            // spans a0 (the table of a0 - 1 reaches 64 further) or has ended a segment exactly there.
            if (a0 == (uint32_t)((unsigned long long)(a0 / g.chunk_size) * g.chunk_size)) seg_s = a0;   // a chunk starts a segment (:1802)
            else {
                flush_fpos();                                       // (the parser stage may need the block before to get there)
                const unsigned long long tw = xw::tick();
                const unsigned long long t0 = xw::clock100();
                uint32_t spins = 0;
                for (;;) {
                    const unsigned long long ps = xw::readfirst64(xw::ld_agent64(&V.hx->p_seg));
                    // The parser's word: the segment that starts at `ps >> 32` holds every position below `(uint32_t)ps` -- it has got that far.
                    // Round 6: this stage may know the rest itself.  A segment goes on through every position that a match found in front of it
                    // reaches beyond (:1550-1554: end_p = max(p + max_len)), and every position from the parser's word up to a0 is such a one when
                    // the last position that is NOT lies in front of the word (gap_last: a segment can only be LONGER than the matches make it --
                    // a rep probe, :1598-1628 -- never shorter; the forced cut at 4,096 positions is the one other end, :1469).  Then a0 lies in
                    // the parser's segment whatever the parse will be: 84 - 90 % of the nice regions of source code start within 256 positions
                    // of the parser, in its segment (oracle probe) -- the stage stood here ~200,000 cycles each time.
                    const uint32_t sa = (uint32_t)(ps >> 32), cover = (uint32_t)ps;
                    const bool covered = (int32_t)(cover - (a0 + 1)) >= 0;
                    const bool own = !covered && (int32_t)(gap_last - cover) < 0 && a0 - sa < kParseMax;
                    if (covered || own) {
                        xw::after_poll();
                        seg_s = sa;
                        n_seg_own += own; n_seg_wait += spins != 0;
                        xw::trace(1, a0, seg_s, (uint32_t)ps);
                        break;
                    }
                    if ((++spins & 63u) == 0) {
                        if (xw::readfirst(xw::ld_agent(&V.hx->err))) { err = kErrInternal + 100; return 1; }
#ifndef NLZM_SIM
                        if (xw::clock100() - t0 > 3000000000ull) { fail(kErrTimeout, a0, 5, (uint32_t)ps, (uint32_t)(ps >> 32)); return 1; }
#else
                        (void)t0;
#endif
                    }
                    xw::pause();
                }
                t_wait += xw::tick() - tw;
            }
        }
        const bool call_full = in_blk && !nice_pred;
        const bool call_some = call_full || (nice_pred && ((a - seg_s) & 7u) == 0);
        const bool ht_call = call_some && avail >= 4;                // :1515, :1530
        const bool rk_call = call_some && avail >= 256;              // :1525, :1538

        // ---- RK256: window ends passed since the last call are inserted with the CALLING position (:1084-1087).
        // Lane 0 does its own now; a later lane that has any becomes lane 0 of the next block.
        uint32_t cut_rk = 64;
        {
            const unsigned long long rkm = xw::ballot(rk_call);
            if (rkm & 1ull) {
                const uint32_t q0 = a0 - base;
                for (uint32_t e = (rk_end | 255u) + 1; e < q0 + 256; e += 256) {
                    const uint32_t hh = xw::readfirst(G.rkhash[base + e - 256]);
                    if (i == 0) G.rk_table[hh >> g.rk_shift] = q0 | (hh << g.wbits);
                    n_rki++;
                }
                xw::drain();
            }
            // rk_end as lane i finds it: the previous calling lane's q + 256
            const unsigned long long below = rkm & ((1ull << i) - 1);
            const uint32_t prev_end = below ? (a0 + (63u - (uint32_t)__builtin_clzll(below))) - base + 256 : rk_end;
            const bool catch_up = rk_call && i > 0 && ((prev_end | 255u) + 1) < q + 256;
            const unsigned long long cm = xw::ballot(catch_up);
            if (cm) cut_rk = (uint32_t)__builtin_ctzll(cm);
        }

        const unsigned long long f1 = ptick();
        // ---- loads that the position alone addresses: requested at the END of the block before (its cut fixes this block's first position), so
        // that their round trip passes behind that block's commit and this block's prediction; the RK256 slot, which hangs on the hash, goes out
        // at once and is used after the HT rows
        uint32_t v4 = 0;
        unsigned long long own0 = 0, own1 = 0;
        uint32_t rkh = 0;
        bool unc = false;
        if (pf_a0 == a0) {
            if (in_blk) { own0 = pf_own0; own1 = pf_own1; v4 = (uint32_t)own0; }
            rkh = rk_call ? pf_rkh : 0u;
            unc = in_blk && G.workers && pf_unc != 0;
        } else {
            if (in_blk) { own0 = load64u(cur); own1 = load64u(cur + 8); v4 = (uint32_t)own0; }
            rkh = rk_call ? G.rkhash[a] : 0u;
            unc = in_blk && G.workers && G.unc[bi] != 0;
        }
        uint32_t rkv = rk_call ? G.rk_table[rkh >> g.rk_shift] : 0u;
        const bool bt_call = call_full && avail >= 4 && G.workers;

        const unsigned long long f2 = ptick();
        // ---- HT2 / HT3 rows as each calling lane finds them (:910-938): the tables in LDS, overwritten by what the
        // earlier calling lanes of the block store (bucket b: HT2[b] = E2; HT3[b+1] = HT3[b], HT3[b] = E3)
        const uint32_t h2 = hash4(v4 & 0xFFFFu), h3 = hash4(v4 & 0xFFFFFFu);     // :1516-1517
        const uint32_t i2 = h2 >> 20, i3 = h3 >> g.ht3_shift;
        const uint32_t tag2 = h2 & g.tag_mask, tag3 = h3 & g.tag_mask;
        const uint32_t e2 = q | (tag2 << g.wbits), e3 = q | (tag3 << g.wbits);   // q un-masked (:913)
        uint32_t row2 = 0, r0 = 0, r1 = 0;
        if (ht_call) { row2 = L->ht2[i2]; r0 = L->ht3[i3]; r1 = L->ht3[i3 + 1]; }
        const unsigned long long htm = xw::ballot(ht_call);
        // Only a lane that has a LATER calling lane on one of its rows (HT2: the same bucket; HT3: buckets b-1, b, b+1, for the
        // overlap) matters to anybody else -- a few of the 64.  Found without comparing all pairs: every calling lane marks its
        // bucket class with its number (LDS max), then looks whether a higher number stands on a class it touches.  Classes
        // are buckets modulo 4096: a false neighbour only sends a lane through the ordered path for nothing.
        bool later = false;
        {
            uint32_t *last = L->last;
            if (ht_call) { xw::lds_max(&last[i2], i + 1); xw::lds_max(&last[4096 + (i3 & 4095u)], i + 1); }
            xw::wave_sync();
            if (ht_call) later = last[i2] > i + 1 || last[4096 + (i3 & 4095u)] > i + 1 || last[4096 + ((i3 + 1) & 4095u)] > i + 1 || last[4096 + ((i3 - 1) & 4095u)] > i + 1;
            xw::wave_sync();
            if (ht_call) { last[i2] = 0; last[4096 + (i3 & 4095u)] = 0; }
        }
        const unsigned long long dep = xw::ballot(later);                       // (lanes whose rows a later lane sees or overwrites)
        for (unsigned long long mm = dep; mm; mm &= mm - 1) {
            const uint32_t j = (uint32_t)__builtin_ctzll(mm);
            const uint32_t b2 = xw::readlane(i2, j), b3 = xw::readlane(i3, j);
            const uint32_t x2 = xw::readlane(e2, j), x3 = xw::readlane(e3, j), y0 = xw::readlane(r0, j);
            if (ht_call && i > j) {
                if (i2 == b2) row2 = x2;
                if (i3 == b3) { r0 = x3; r1 = y0; }
                else if (i3 == b3 + 1) r0 = y0;
                else if (i3 + 1 == b3) r1 = x3;
            }
        }

        const unsigned long long f3 = ptick();
        // ---- candidates (:922-925) and their first 16 bytes; what is longer goes to the job list
        uint32_t cd[4] = { 0, 0, 0, 0 };            // distance of candidate k: HT2, HT3 row 0, HT3 row 1, RK256; 0: none
        if (ht_call) {
            const uint32_t s0 = row2 & g.wmask, s1 = r0 & g.wmask, s2 = r1 & g.wmask;
            if ((row2 >> g.wbits) == tag2 && s0 < q && q - s0 <= g.wmask) cd[0] = q - s0;
            if ((r0 >> g.wbits) == tag3 && s1 < q && q - s1 <= g.wmask) cd[1] = q - s1;
            if ((r1 >> g.wbits) == tag3 && s2 < q && q - s2 <= g.wmask) cd[2] = q - s2;
        }
        // RK256's probe (:1090-1098): a slot whose tag fits gives a candidate; most are tag collisions that differ within
        // a few bytes, which the lane tells itself (below) -- only what survives 16 bytes is measured by the whole wave
        const bool rk_act = rk_len > 0 && q - rk_to < rk_len;       // the carried match still covers this position (:1056-1069)
        const uint32_t rk_eff = rk_act ? rk_len : 0u;
        const bool rk_probe = rk_call && rk_eff < 256;              // :1090
        bool rk_cand = false;
        uint32_t rk_d = 0;
        if (rk_probe) {
            const uint32_t sp = rkv & g.wmask;
            if ((rkv >> g.wbits) == (rkh & g.tag_mask) && sp < q && q - sp <= g.wmask) { rk_cand = true; rk_d = q - sp; }
        }
        uint32_t rk_l16 = 16;
        if (rk_cand) {
            const unsigned long long x0 = load64u(cur - rk_d) ^ own0;
            if (x0) rk_l16 = (uint32_t)__builtin_ctzll(x0) >> 3;
            else {
                const unsigned long long x1 = load64u(cur - rk_d + 8) ^ own1;
                rk_l16 = x1 ? 8 + ((uint32_t)__builtin_ctzll(x1) >> 3) : 16;
            }
        }
        // a mismatch inside the 16 bytes and inside the (uint16) length cap: the length is final; not taken (:1099): no event
        const bool rk_short = rk_cand && rk_l16 < 16 && rk_l16 < (avail & 0xFFFFu);
        const bool rk_dropped = rk_short && !(rk_l16 >= rk_eff && rk_l16 >= match_min(rk_d));
        uint32_t cl[3] = { 0, 0, 0 };
        L->njobs = 0;
        xw::wave_sync();
        // (the sixteen bytes of all three candidates are requested before any of them is looked at: ONE round trip to memory for the block --
        //  with the second eight bytes requested only behind a first eight that matched, a block paid up to six dependent ones)
        unsigned long long cx0[3], cx1[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            cx0[k] = cd[k] ? load64u(cur - cd[k]) : 0ull;
            cx1[k] = cd[k] ? load64u(cur - cd[k] + 8) : 0ull;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (!cd[k]) continue;
            const unsigned long long x0 = cx0[k] ^ own0;
            uint32_t l;
            if (x0) l = (uint32_t)__builtin_ctzll(x0) >> 3;
            else {
                const unsigned long long x1 = cx1[k] ^ own1;
                l = x1 ? 8 + ((uint32_t)__builtin_ctzll(x1) >> 3) : 16;
            }
            if (l >= cap) l = cap;
            else if (l == 16) {                                                 // longer than 16 bytes: a job
                const uint32_t at = xw::lds_inc(&L->njobs);
                L->job[at * 4] = i | ((uint32_t)k << 8); L->job[at * 4 + 1] = cd[k]; L->job[at * 4 + 2] = cap;
                l = kNone;
            }
            cl[k] = l;
        }
        xw::wave_sync();
        {   // jobs, eight at a time: job j on lanes 8j..8j+7, eight bytes per lane and round
            const uint32_t nj = xw::readfirst(L->njobs);
            for (uint32_t jb = 0; jb < nj; jb += 8) {
                const uint32_t j = jb + (i >> 3), sub = i & 7u;
                const bool have = j < nj;
                const uint32_t jl = have ? L->job[j * 4] : 0u, jd = have ? L->job[j * 4 + 1] : 1u, jcap = have ? L->job[j * 4 + 2] : 0u;
                const uint32_t ja = a0 + (jl & 63u);
                uint32_t res = kNone;                // first mismatch found by this lane
                bool open = have;
                uint32_t off = 16;
                while (xw::any(open)) {
                    uint32_t pos = kNone;
                    const uint32_t my = off + sub * 8;
                    if (open && my < jcap) {
                        const unsigned long long d = load64u(G.in + ja - jd + my) ^ load64u(G.in + ja + my);
                        if (d) { const uint32_t p = my + ((uint32_t)__builtin_ctzll(d) >> 3); if (p < jcap) pos = p; }
                    }
                    const unsigned long long bal = xw::ballot(pos != kNone);
                    const uint32_t mine = (uint32_t)(bal >> (i & ~7u)) & 0xFFu;
                    const uint32_t got = xw::shfl(pos, (i & ~7u) + (mine ? (uint32_t)__builtin_ctz(mine) : 0u));
                    if (open && mine) { res = got; open = false; }
                    else if (open && off + 64 >= jcap) { res = jcap; open = false; }
                    off += 64;
                }
                if (have && sub == 0) L->job_len[(jl & 63u) * 4 + (jl >> 8)] = res;
            }
            xw::wave_sync();
#pragma unroll
            for (int k = 0; k < 3; k++) if (cd[k] && cl[k] == kNone) cl[k] = L->job_len[i * 4 + k];
        }

        const unsigned long long f4 = ptick();
        // ---- the record of this position: pairs of HT2 and HT3 (:917-933)
        uint32_t *st = L->stage + i * kFtStride;
        uint32_t np = 0, ec = 0, ec_d = kNone, od = kNone, cmpb = 0;   // pairs, largest closed end and the smallest distance that ends there, smallest open distance, bytes compared
        auto closed_end = [&](uint32_t e, uint32_t d) __attribute__((always_inline)) {
            if (e > ec || (e == ec && d < ec_d)) { ec = e; ec_d = d; }
        };
        auto add_pair = [&](uint32_t d, uint32_t l) __attribute__((always_inline)) {
            if (l >= cap) { od = umin(od, d); return; }     // as long as the lookahead allows: may grow at the next position
            st[2 + 2 * np] = d; st[3 + 2 * np] = l; np++;
            closed_end(a + l, d);
        };
        if (ht_call) {
            if (cd[0] && 1 < cap) {
                cmpb += cl[0] + (cl[0] < cap);
                if (cl[0] > 1 && cl[0] >= match_min(cd[0])) add_pair(cd[0], cl[0]);
            }
            uint32_t best = 1;
#pragma unroll
            for (int k = 1; k < 3; k++) {
                if (!cd[k] || !(best < cap)) continue;
                cmpb += cl[k] + (cl[k] < cap);
                if (cl[k] > best && cl[k] >= match_min(cd[k])) { add_pair(cd[k], cl[k]); best = cl[k]; }
            }
        }

        // ---- RK256 (:1055-1113)
        // An RK256 entry whose compare the uint16 length parameter ended (`la_end - p` just above a multiple of 65,536: chunk offsets 56,833 - 57,080
        // at 122,368-byte chunks) is SHORTER than the bytes agree.  Where it is the table's longest entry, the next position extends it again
        // (:1503-1512 extends delta[max_len], whatever found it): it becomes the growing top entry.  Where something longer stands in the table
        // it is an ordinary closed entry.  Decided below, when the ends of everything else at this lane are known.
        uint32_t so_d = 0, so_end = 0;                      // this lane's entry of that kind: distance, absolute end (0: none)
        auto add_soft = [&](uint32_t d, uint32_t l) __attribute__((always_inline)) {
            st[2 + 2 * np] = d; st[3 + 2 * np] = l; np++;
            if (a + l > so_end || (a + l == so_end && d < so_d)) { so_d = d; so_end = a + l; }
        };
        if (rk_call && rk_act) {
            const uint32_t d = rk_to - rk_from, l = rk_len - (q - rk_to);
            if (l >= match_min(d)) { if (rk_cut && l < cap) add_soft(d, l); else add_pair(d, umin(l, kMatchMax)); }
        }
        // an aligned insert (:1109-1112) of an earlier lane rewrites the slot this lane read: cut there
        uint32_t cut_slot = 64;
        {
            const unsigned long long im = xw::ballot(rk_call && (q & 255u) == 0);
            for (unsigned long long mm = im; mm; mm &= mm - 1) {
                const uint32_t j = (uint32_t)__builtin_ctzll(mm);
                const uint32_t sj = xw::readlane(rkh >> g.rk_shift, j);
                const unsigned long long hit = xw::ballot(rk_probe && i > j && (rkh >> g.rk_shift) == sj);
                if (hit) cut_slot = umin(cut_slot, (uint32_t)__builtin_ctzll(hit));
            }
        }
        if (rk_dropped) cmpb += rk_l16 + 1;                             // (the bytes the reference's compare looked at)
        uint32_t cut_ev = 64, ev_d = 0, ev_l = 0;
        bool ev_ok = false, ev_cut = false;
        {   // the first candidate that is left is measured by the whole wave (up to 65,535 bytes: uint16 parameter, :760, :1096)
            const unsigned long long evm = xw::ballot(rk_cand && !rk_dropped);
            if (evm) {
                const uint32_t k = (uint32_t)__builtin_ctzll(evm);
                const uint32_t kd = xw::readlane(rk_d, k), kav = xw::readlane(avail, k) & 0xFFFFu, keff = xw::readlane(rk_eff, k);
                const uint32_t ka = a0 + k;
                const uint32_t l = wave_cmp(G.in, ka - kd, ka, 0, kav);
                cut_ev = k + 1;
                ev_ok = l >= keff && l >= match_min(kd);                        // :1099-1105 (state taken over below if lane k is final)
                ev_d = kd; ev_l = l;
                ev_cut = l == kav;                                              // (no mismatch: the uint16 ended it)
                if (i == k) {
                    cmpb += l + (l < kav);
                    if (ev_ok) { if (ev_cut && l < cap) add_soft(kd, l); else add_pair(kd, umin(l, kMatchMax)); }
                }
            }
        }

        const unsigned long long f5 = ptick();
        // ---- BT4: the worker lanes' result (longest record-setter in the record's words 9, 10).
        // A worker lane walks its bin in position order.  At an `unc` position it does not wait for this stage's decision: it
        // assumes it and goes on, but the RESULTS of the calls behind such a position are held back until the decision is
        // stored (a decision the other way takes those calls back).  Until round 6 a decision was stored at the block's commit, so a later
        // position of the same bin could not be in the block (it became lane 0 of the next one: 9 - 18 % of all blocks were cut for that).
        // Now every decision is stored as soon as it is exact -- in the wait loop below, when the lanes in front are settled -- and the worker
        // lets the held results go: nothing is cut for the bins any more.
        constexpr uint32_t cut_bin = 64;
        const bool bt_wait = bt_call && i < cut_bin;
        uint32_t bt_n = 0;
        flush_fpos();                                               // (the block before: its records have landed by now)
        // ---- the worker lanes' results, and the verification: what the table's reach really is in front of every lane (:1514 sees it after carry
        // + extend), where the block is cut.  Round 6: ONE loop.  The stage used to wait for the result of every lane it had predicted a call for,
        // and verified then -- but a block is cut at the first lane whose prediction was wrong, and at the start of every long repeat that is the
        // lane right behind the position that found the match (the lanes behind it are in a nice region, :1514: their calls never happen, and their
        // results -- dry runs of lanes and waves that have better things to do -- were waited for all the same: one block in six on source code).
        // A lane's reach, and so its decision, hangs on the lanes in front of it only: as soon as every lane in front of the first missing result
        // is there, everything up to and including that lane's decision is exact, and a cut at or in front of it ends the wait.
        const uint32_t ec0 = ec, ec_d0 = ec_d, od0 = od;            // (without the worker lanes' results)
        uint32_t pm = 0, before = 0, js = 64, jo = 64, cut_nice = 64, m = 0;
        bool nice_real = false, so_top = false, so_tie = false, so_win = false;
        unsigned long long stm = 0, om = 0;
        {
            const unsigned long long tw = xw::tick();
            const unsigned long long t0 = xw::clock100();
            uint32_t spins = 0;
            const bool waits = xw::any(bt_wait);
            bool have = !bt_wait;                                   // this lane's result is in (or none is awaited)
            uint32_t r_d = 0, r_l = 0;
            bool told = false;                                      // this lane's decision is stored
            uint32_t f_seen = 65;                                   // the first lane whose result was missing when the block was last verified (65: not yet)
            for (;;) {
                // (quad 0 of the record in one load: the ready word and the longest pair behind it -- they were two dependent round trips)
                {   // (host simulation: the worker emulation is asked for the FIRST missing result only, so that results come in one by one and the
                    //  ones behind a cut never do -- the device's worker lanes need no asking)
                    const unsigned long long ask = xw::ballot(!have);
                    if (ask && i == (uint32_t)__builtin_ctzll(ask)) xw::need_bt(G.hook_user, a);
                }
                if (!have) {
                    const xw::u32x4 q0 = xw::ld_agent128(G.bt_ready + bi * kBtRec);
                    if (q0.x & kBtReady) { have = true; bt_n = q0.x & 0x1FFu; r_d = q0.y; r_l = q0.z; }
                }
                const unsigned long long missing = xw::ballot(!have);
                if (waits && spins == 0) {       // (accounting: whose results are not there when the block asks first)
                    n_late_unc += (uint32_t)__builtin_popcountll(missing & xw::ballot(unc));
                    n_late_other += (uint32_t)__builtin_popcountll(missing & ~xw::ballot(unc));
                    if (missing) {      // ... of a hot bin's wave or of a lane; lane 0 of the block (the position a block was cut at) or a later one
                        const bool hot = G.hot_of_bin && !have && G.hot_of_bin[(hash4(v4) >> g.bt_shift) % G.nheads] != 0;
                        n_late_hot += (uint32_t)__builtin_popcountll(xw::ballot(hot));
                        n_late_first += (uint32_t)(missing & 1ull);
                        n_late_blocks++;
                    }
                }
                const uint32_t F = missing ? (uint32_t)__builtin_ctzll(missing) : 64u;          // the first lane whose result is missing
                if (F != f_seen) {      // (results behind the first missing one change nothing that can be used yet: verified again when THAT one has come)
                    f_seen = F;
                    xw::after_poll();
                    ec = ec0; ec_d = ec_d0; od = od0;
                    if (bt_wait && have && bt_n) { if (r_l >= cap) od = umin(od, r_d); else closed_end(a + r_l, r_d); }
                    pm = xw::scan_max(in_blk ? umax(ec, so_end) : 0u);          // inclusive prefix max of the closed ends
                    before = umax(xw::lane_below(pm, 0u), reach);
                    if (i >= jc && s_active) before = umax(before, s_end);      // the former top entry, closed
                    nice_real = in_blk && umax(before, s_sliding ? s_e : 0u) >= a + kNice;
                    // a cut-short RK256 entry that is longer than everything else in this lane's table grows from the next position on (the lanes in front of
                    // the first such lane hold theirs as closed entries: `before` is right for it).  As long as another entry: which of the two the table's
                    // end holds hangs on their distances (:835-852 keeps the smaller): looked up then.
                    const bool so_free = in_blk && so_end != 0 && od == kNone && !s_sliding;
                    so_tie = so_free && so_end == umax(before, ec);
                    so_win = false;
                    if (xw::any(so_tie)) {      // (rare: the smallest distance among the other entries that end exactly there -- own, of the lanes in front, older ones)
                        uint32_t d2 = kNone;
                        if (ec == so_end) d2 = ec_d;
                        if (reach == so_end) d2 = umin(d2, reach_d);
                        if (i >= jc && s_active && s_end == so_end) d2 = umin(d2, s_d);
                        for (uint32_t j = 0; j + 1 < n; j++) {
                            const uint32_t ej = xw::readlane(ec, j), dj = xw::readlane(ec_d, j), sj = xw::readlane(so_end, j), sdj = xw::readlane(so_d, j);
                            if (j < i && ej == so_end) d2 = umin(d2, dj);
                            if (j < i && sj == so_end) d2 = umin(d2, sdj);
                        }
                        so_win = so_tie && so_d < d2;
                    }
                    so_top = so_free && (so_end > umax(before, ec) || so_win);
                    stm = xw::ballot(so_top);
                    js = stm ? (uint32_t)__builtin_ctzll(stm) : 64u;
                    const unsigned long long bad = xw::ballot(in_blk && nice_real != nice_pred);
                    cut_nice = bad ? (uint32_t)__builtin_ctzll(bad) : 64u;
                    // a match as long as the lookahead allows becomes the growing top entry, unless one with a smaller
                    // distance is growing already (:835-852 keeps the smaller distance at the table's end)
                    om = xw::ballot(in_blk && od != kNone && (!s_sliding || od < s_d));
                    jo = om ? (uint32_t)__builtin_ctzll(om) : 64u;
                    m = umin(n, umin(cut_nice, umin(jo + 1 > 64 ? 64u : jo + 1, umin(cut_ev, umin(cut_rk, umin(cut_slot, cut_bin))))));
                    m = umin(m, js + 1 > 64 ? 64u : js + 1);
                    // (lanes behind the first missing one contribute what they have: a cut they ask for lies behind F and is not taken; a cut at or in
                    //  front of F hangs on lanes in front of F only -- F's own decision included, which is why `<=`)
                    // every lane in front of F, and in front of the cut, is exact and will be committed as it stands: its decision is said now (the
                    // worker lanes hold the results of the same bin's later positions until they have it)
                    if (unc && avail >= 4 && !told && i < umin(F, m)) { xw::st_agent(G.bt_flag + bi, nice_real ? kFlagSkip : kFlagCall); told = true; }
                    if (m <= F) break;
                    // Lane F's decision is exact now, and it is "call" (a nice lane F would be a wrong prediction: a cut at F, and the loop were left): said
                    // at once, not at the block's commit.  The worker of an undecided position that it ASSUMES to be skipped does not make the call
                    // (round 6: no dry run -- a third of all BT4 work on source code was descents thrown away): without this word it would not
                    // make it before the commit, and the commit waits for its result.
                    if (i == F && unc) xw::st_agent(G.bt_flag + bi, kFlagCall);
                }
                if ((++spins & 63u) == 0) {
                    if (xw::readfirst(xw::ld_agent(&V.hx->err))) { err = kErrInternal + 100; return 1; }
#ifndef NLZM_SIM
                    if (xw::clock100() - t0 > 3000000000ull) {
                        // (which positions are missing, and whether the first of them is one whose call this stage decides)
                        const uint32_t fl = (uint32_t)__builtin_ctzll(missing | (1ull << 63));
                        fail(kErrTimeout, a0 + fl, 6, (uint32_t)__builtin_popcountll(missing) | (xw::readlane(unc ? 1u : 0u, fl) << 8) | (n << 16),
                             xw::readlane((hash4(v4) >> g.bt_shift) % G.nheads, fl));
                        return 1;
                    }
#else
                    (void)t0;
#endif
                }
                xw::pause();
            }
            if (waits) { t_wait += xw::tick() - tw; t_wait_bt += xw::tick() - tw; }
        }

        const unsigned long long f6 = ptick();
        if (m == 0) { fail(kErrInternal, a0, 7); return 1; }
        n_blocks++;
        if (m < n) {        // (why the block was cut; diagnostics)
            // (adds, not a chain of branches: the compiler turns the chain into ONE indexed access and the stage's state goes to scratch)
            const uint32_t w = m == cut_nice ? 0u : (m == jo + 1 ? 1u : (m == cut_ev ? 2u : (m == cut_rk ? 3u : (m == cut_bin ? 5u : 4u))));
            n_cut0 += w == 0; n_cut1 += w == 1; n_cut2 += w == 2; n_cut3 += w == 3; n_cut4 += w == 4; n_cut5 += w == 5;
        }
        const bool fin = i < m;
        {   // the last position that nothing found in front of it reaches beyond (the segment start of a later nice region, above)
            const unsigned long long gapm = xw::ballot(fin && umax(before, s_sliding ? s_e : 0u) <= a);
            if (gapm) gap_last = a0 + (63u - (uint32_t)__builtin_clzll(gapm));
        }
        if (stm || xw::any(so_tie)) {       // (accounting)
            n_so_top += (uint32_t)__builtin_popcountll(xw::ballot(fin && so_top)); n_so_tie += (uint32_t)__builtin_popcountll(xw::ballot(fin && so_tie));
            n_so_won += (uint32_t)__builtin_popcountll(xw::ballot(fin && so_win));
        }

        // a position that the pre-filter promised to be a BT4 position must not be nice
        if (xw::any(fin && nice_real && !unc && avail >= 4 && G.workers)) { fail(kErrV2Promise, a0, 8); return 1; }

        const unsigned long long f7 = ptick();
        // ---- commit
        // HT rows in position order (the last writer of a row wins): the lanes with a later lane on their rows one after the
        // other, then all the others at once (no two of them touch the same row, and nobody after them touches theirs)
        for (unsigned long long mm = dep & htm & ((m < 64 ? (1ull << m) : 0ull) - 1ull); mm; mm &= mm - 1) {
            const uint32_t j = (uint32_t)__builtin_ctzll(mm);
            if (i == j) { L->ht2[i2] = e2; L->ht3[i3] = e3; L->ht3[i3 + 1] = r0; }
            xw::wave_sync();
        }
        if (ht_call && i < m && !later) { L->ht2[i2] = e2; L->ht3[i3] = e3; L->ht3[i3 + 1] = r0; }
        xw::wave_sync();
        if (fin && unc && avail >= 4) xw::st_agent(G.bt_flag + bi, nice_real ? kFlagSkip : kFlagCall);
        if (fin && rk_call && (q & 255u) == 0) G.rk_table[rkh >> g.rk_shift] = q | (rkh << g.wbits);
        // the record
        if (fin) {
            uint32_t w0 = np | (bt_call && bt_n ? kFtBt : 0u) | ((uint32_t)(uint8_t)own0 << 8);
            // the former top entry stops growing here: from now on an ordinary entry
            if (i == jc && s_active && s_end > a + 1) { st[2 + 2 * np] = s_d; st[3 + 2 * np] = s_end - a; np++; w0++; }
            if (s_sliding && i <= jo) { st[14] = s_d; st[15] = cap; w0 |= kFtTop; }
            if (i == jo) { st[14] = od; st[15] = cap; w0 |= kFtTop; }
            st[0] = w0; st[1] = a;
            uint32_t *dst = V.ft + (unsigned long long)(a & (kFtRing - 1)) * kFtStride;
#pragma unroll
            for (int k = 0; k < 4; k++) xw::st_agent128(dst + 4 * k, st[4 * k], st[4 * k + 1], st[4 * k + 2], st[4 * k + 3]);
        }
        // ---- state after lane m - 1 (the record's stores travel meanwhile: drained before the progress word below)
        {
            const uint32_t last = m - 1;
            const uint32_t pm_last = xw::readlane(pm, last);
            if (pm_last >= reach && pm_last) {     // (whose entry ends there: nearly always one lane's)
                uint32_t dm = pm_last == reach ? reach_d : kNone;
                const uint32_t le = umax(ec, so_end), ld = umin(ec == le ? ec_d : kNone, so_end == le ? so_d : kNone);
                for (unsigned long long mm = xw::ballot(fin && le == pm_last); mm; mm &= mm - 1) dm = umin(dm, xw::readlane(ld, (uint32_t)__builtin_ctzll(mm)));
                reach = pm_last; reach_d = dm;
            }
            if (s_active && jc <= last) {
                if (s_end > reach) { reach = s_end; reach_d = s_d; } else if (s_end == reach) reach_d = umin(reach_d, s_d);
                s_active = 0;
            }
            if (jo == last) { s_active = 1; s_d = xw::readlane(od, last); s_end = kNone; s_seen = a0 + last + xw::readlane(cap, last); }
            else if (js == last) { s_active = 1; s_d = xw::readlane(so_d, last); s_end = kNone; s_seen = xw::readlane(so_end, last); }
            prev_nice = xw::readlane(nice_real ? 1u : 0u, last);
            const unsigned long long rkf = xw::ballot(fin && rk_call);
            if (rkf) rk_end = (a0 + (63u - (uint32_t)__builtin_clzll(rkf))) - base + 256;
            // a calling position beyond the carried RK match drops it for good (:1066-1068; rk_to is never rebased, so a
            // later position of the next epoch could look covered again)
            if (rk_len && xw::any(fin && rk_call && !rk_act)) rk_len = 0;
            if (ev_ok && cut_ev == m) {                                         // the RK candidate of lane m-1 was taken (:1102-1104)
                rk_from = (a0 + last - base) - ev_d; rk_to = a0 + last - base; rk_len = ev_l; rk_cut = ev_cut ? 1u : 0u;
            }
        }
        // counters
        n_pos += m;
        n_nice += (unsigned long long)__builtin_popcountll(xw::ballot(fin && nice_real));
        n_unc += (unsigned long long)__builtin_popcountll(xw::ballot(fin && unc && avail >= 4));
        n_ht += (unsigned long long)__builtin_popcountll(xw::ballot(fin && ht_call));
        n_rkp += (unsigned long long)__builtin_popcountll(xw::ballot(fin && rk_probe));
        n_rki += (unsigned long long)__builtin_popcountll(xw::ballot(fin && rk_call && (q & 255u) == 0));
        n_cmp_lane += fin ? cmpb : 0u;                              // (summed over the lanes once, when the launch ends: six dependent shuffles per block otherwise)
        pend_fpos = a0 + m;                                         // (said by flush_fpos)
        // the next block's own loads (same chunk: same lookahead end)
        pf_a0 = kNone;
        if (a0 + m < a1) {
            pf_a0 = a0 + m;
            const uint32_t na = pf_a0 + i;
            const bool nin = na < a1;
            pf_own0 = nin ? load64u(G.in + na) : 0ull; pf_own1 = nin ? load64u(G.in + na + 8) : 0ull;
            pf_rkh = (nin && la_end - na >= 256) ? G.rkhash[na] : 0u;
            pf_unc = (nin && G.workers) ? G.unc[(unsigned long long)(na - G.batch_a0)] : (uint8_t)0;
        }
        xw::trace(2, a0, n, m, reach, s_active, s_d, s_end);
        {
            const unsigned long long f8 = ptick();
            t_f[0] += f1 - f0; t_f[1] += f2 - f1; t_f[2] += f3 - f2; t_f[3] += f4 - f3; t_f[4] += f5 - f4; t_f[5] += f6 - f5; t_f[6] += f7 - f6; t_f[7] += f8 - f7;
        }
        return m;
    }

    XW_FN void run(uint32_t c0, uint32_t c1)
    {
        FLds *L = xw::lds<FLds>();
        Persist *P = G.persist;
        StateV2 *S = (StateV2 *)V.state;
        const uint32_t i = xw::lane();
        const uint32_t ht3_rows = (1u << (32 - g.ht3_shift)) + 1;
        for (uint32_t k = i; k < 8192; k += 64) L->last[k] = 0;
        for (uint32_t k = i; k < 4096; k += 64) L->ht2[k] = G.ht2[k];
        for (uint32_t k = i; k < ht3_rows; k += 64) L->ht3[k] = G.ht3[k];
        base = xw::readfirst((uint32_t)P->reb_base);
        reach = xw::readfirst(S->reach); reach_d = xw::readfirst(S->reach_d); s_active = xw::readfirst(S->s_active); s_d = xw::readfirst(S->s_d);
        s_end = xw::readfirst(S->s_end); s_seen = xw::readfirst(S->s_seen);
        prev_nice = xw::readfirst(S->prev_nice); seg_s = xw::readfirst(S->seg_s);
        rk_from = xw::readfirst(P->rk_from); rk_to = xw::readfirst(P->rk_to); rk_len = xw::readfirst(P->rk_len); rk_end = xw::readfirst(P->rk_end);
        rk_cut = rk_len >> 31; rk_len &= 0x7FFFFFFFu;                       // (the length is at most 65,535)
        err = xw::readfirst(P->error);
        if (!err && G.test_fail) fail(kErrInternal + 50, (uint32_t)((unsigned long long)c0 * g.chunk_size), 9);      // (test of the fault path: option "test_fail_stream")
        n_pos = n_nice = n_unc = n_ht = n_rkp = n_rki = n_cmp = n_blocks = 0;
        n_cut0 = n_cut1 = n_cut2 = n_cut3 = n_cut4 = n_cut5 = 0;
        t_wait = 0;
        const unsigned long long t_start = xw::tick();
        t_pos_seen = (uint32_t)((unsigned long long)c0 * g.chunk_size);
        unsigned long long shifts = 0;
        xw::wave_sync();
        for (uint32_t ci = c0; ci < c1 && !err; ci++) {
            const unsigned long long chunk_abs = (unsigned long long)ci * g.chunk_size;
            const unsigned long long remain = g.n - chunk_abs;
            const uint32_t chunk_read = (uint32_t)(remain < g.feed ? remain : g.feed);
            const uint32_t p_end = umin(g.chunk_size, chunk_read);
            if (chunk_abs - base >= 2ull * (g.wmask + 1)) {                     // :1786-1792
                base += g.wmask + 1;
                shifts++;
                if (i == 0) { L->ht2[0] = kNone; L->ht3[0] = kNone; }           // MatchFinderHT::Shift (:940-957)
                if (rk_end >= g.wmask + 1) rk_end -= g.wmask + 1; else rk_end = 0;   // :1115-1123
                xw::wave_sync();
            }
            if (prev_nice) seg_s = (uint32_t)chunk_abs;                          // a chunk starts a segment (:1802)
            gap_last = (uint32_t)chunk_abs;
            const uint32_t a1 = (uint32_t)chunk_abs + p_end, la_end = (uint32_t)chunk_abs + chunk_read;
            uint32_t a = (uint32_t)chunk_abs;
            while (a < a1 && !err) {
                const uint32_t n = umin(64u, a1 - a);
                dbg_a = a;
                // ring space: the table stage must have consumed position a + n - kFtRing
                if ((int32_t)(a + n - t_pos_seen - kFtRing) > 0) {
                    flush_fpos();
                    const unsigned long long tw = xw::tick();
                    if (!wait_word_ge(&V.hx->t_pos, a + n - kFtRing, V.hx, 1)) { err = kErrInternal + 100; break; }
                    t_pos_seen = xw::readfirst(xw::ld_agent(&V.hx->t_pos));
                    t_wait += xw::tick() - tw;
                }
                a += block(a, n, a1, la_end);
            }
        }
        flush_fpos();
        for (uint32_t d = 32; d; d >>= 1) n_cmp_lane += xw::shfl64(n_cmp_lane, i ^ d);
        n_cmp += xw::readfirst64(n_cmp_lane);
        xw::wave_sync();
        // (opaque: or the 64 store addresses are computed when the launch begins, kept through it, and one of them spilled)
        for (uint32_t k = xw::opaque(i); k < 4096; k += 64) G.ht2[k] = L->ht2[k];
        for (uint32_t k = xw::opaque(i); k < ht3_rows; k += 64) G.ht3[k] = L->ht3[k];
        if (i == 0) {
            P->reb_base = base;
            P->rk_from = rk_from; P->rk_to = rk_to; P->rk_len = rk_len | (rk_cut << 31); P->rk_end = rk_end;
            S->reach = reach; S->reach_d = reach_d; S->s_active = s_active; S->s_d = s_d; S->s_end = s_end; S->s_seen = s_seen;
            S->prev_nice = prev_nice; S->seg_s = seg_s;
            // (every stage adds the counters it owns, with agent-scope atomics: the stages run on different CUs, in
            //  block mode on different XCDs, and finish within microseconds of each other)
            Counters &c = P->cnt;
            xw::atomic_add64_agent(&c.positions, n_pos); xw::atomic_add64_agent(&c.nice_positions, n_nice);
            xw::atomic_add64_agent(&c.uncertain_positions, n_unc); xw::atomic_add64_agent(&c.ht_rows, 3 * n_ht);
            xw::atomic_add64_agent(&c.rk_probes, n_rkp); xw::atomic_add64_agent(&c.rk_inserts, n_rki);
            xw::atomic_add64_agent(&c.cmp_bytes, n_cmp); xw::atomic_add64_agent(&c.shifts, shifts);
            unsigned long long *pr = P->prof;
            xw::atomic_add64_agent(&pr[0], n_blocks); xw::atomic_add64_agent(&pr[1], n_cut0); xw::atomic_add64_agent(&pr[2], n_cut1);
            xw::atomic_add64_agent(&pr[3], n_cut2); xw::atomic_add64_agent(&pr[4], n_cut3); xw::atomic_add64_agent(&pr[5], n_cut4);
            xw::atomic_add64_agent(&pr[12], n_cut5);
            xw::atomic_add64_agent(&pr[16], t_wait); xw::atomic_add64_agent(&pr[17], xw::tick() - t_start); xw::atomic_add64_agent(&pr[25], t_wait_bt);
            xw::atomic_add64_agent(&pr[28], n_late_unc); xw::atomic_add64_agent(&pr[29], n_late_other);
            xw::atomic_add64_agent(&pr[115], n_so_top); xw::atomic_add64_agent(&pr[116], n_so_tie); xw::atomic_add64_agent(&pr[117], n_so_won);
            xw::atomic_add64_agent(&pr[118], n_seg_own); xw::atomic_add64_agent(&pr[119], n_seg_wait);
            xw::atomic_add64_agent(&pr[110], n_late_hot); xw::atomic_add64_agent(&pr[111], n_late_first); xw::atomic_add64_agent(&pr[112], n_late_blocks);
#ifdef NLZM_PROFILE
            for (int z = 0; z < 8; z++) xw::atomic_add64_agent(&pr[88 + z], t_f[z]);
#endif
            if (err || xw::ld_agent(&V.hx->err)) {                  // where this stage was when it left
                uint32_t *d = V.hx->dbg[0];
                xw::st_agent(d + 0, dbg_a); xw::st_agent(d + 1, reach); xw::st_agent(d + 2, s_active); xw::st_agent(d + 3, s_d);
                xw::st_agent(d + 4, s_end); xw::st_agent(d + 5, prev_nice); xw::st_agent(d + 6, seg_s); xw::st_agent(d + 7, rk_len);
                xw::st_agent(d + 8, t_pos_seen); xw::st_agent(d + 9, err); xw::st_agent(d + 10, base);
            }
        }
    }
};


// =================================================================================================
// table stage
// =================================================================================================
// The match table of a position (mt_carry as :1543 leaves it) is delta[l] = smallest distance of any match found at a
// position q <= p that still has >= l bytes left at p (Update is an element-wise min, :835-852; CarryFrom shifts by
// one, :823-833).  With e = q + length (the match's absolute end), delta[l] at p is the smallest distance among the
// entries with e >= p + l: only the Pareto front over (e larger, distance smaller) matters, and dominance between two
// entries does not depend on p.  So the front at p is the Pareto front of ALL pairs found at positions <= p, cut to
// e >= p + 2 -- an associative merge, computed for the 64 positions of a block by a parallel prefix scan.
// (The one entry whose end moves with p, the top entry while it keeps extending :1503-1512, arrives from the finder
// stage as a fresh pair per position.)
// (Round 5 measured the fronts at 300 MB of text at -window:28: some position of a block holds more than 8 entries at some step of
//  the scan in 42 % of the blocks, more than 12 in 0.23 %, more than 16 in 0.002 % -- with 16 entries a wave's two front buffers are
//  16 KB instead of 32 KB and seven waves fit the CU's LDS where four did.
//  REAL text is different: positions with more than 16 (and more than 24) BT4 record-setters of their own are common there -- 6.8 % of
//  its blocks went down the serial path with 16 entries, 2.9 % with 24.  Measured, same box: 24 entries x 5 waves against 16 x 7: real text 2,185 against 2,765
//  cycles per position, the stand-in 421 against 388.)
// (Round 5, real text: 24 entries x 5 waves against 16 x 7, same box: real text 2,185 against 2,765 cycles per position, the stand-in 421 against 388.  Neither
//  shape serves both, so the stage has BOTH and every launch runs in the one the launch before it asked for: the fronts live in one LDS arena that the waves
//  divide among themselves by the shape, and the last wave to leave a launch looks at the share of blocks that went down the serial path.)
#ifndef NLZM_FRCAP
#define NLZM_FRCAP 16
#endif
#ifndef NLZM_KTW
#define NLZM_KTW 7
#endif
#ifndef NLZM_FRCAP_WIDE
#define NLZM_FRCAP_WIDE 24
#endif
#ifndef NLZM_KTW_WIDE
#define NLZM_KTW_WIDE 5
#endif
constexpr uint32_t kFrCap = NLZM_FRCAP, kFrCapWide = NLZM_FRCAP_WIDE;   // entries a lane's front may have on the scan path
constexpr uint32_t kTW = NLZM_KTW, kTWWide = NLZM_KTW_WIDE;             // waves of the stage: each takes whole blocks, in turn (a block of 64 positions takes a wave ~100,000 cycles:
                                                // three waves were busy 530 of 618 cycles per position at 300 MB, four 416 of 614 with the whole tail of a
                                                // block in block order; round 5, only the carry merge in order: 4 / 5 / 6 / 7 waves 506 / 483 / 483 / 483 --
                                                // from five waves on the stage is no longer what the pipeline waits for)
constexpr uint32_t kTWMax = kTW > kTWWide ? kTW : kTWWide;
constexpr uint32_t kFrArena = (kTW * kFrCap > kTWWide * kFrCapWide ? kTW * kFrCap : kTWWide * kFrCapWide) * 2 * 64;     // 64-bit words: two buffers of 64 fronts per wave
// a launch changes to the wide shape when more than 1 block in 64 took the serial path, and back when fewer than 1 in 512 did
constexpr uint32_t kTbWidenShift = 6, kTbNarrowShift = 9;
NLZM_HD uint32_t table_waves(uint32_t wide) { return wide ? kTWWide : kTW; }
struct TWave {
    uint32_t recs[64 * kFtStride];              // the block's finder records
    uint32_t overflow;
};
struct TLds {
    unsigned long long arena[kFrArena];         // the waves' fronts: wave w, buffer b at (2w + b) * 64 * cap; key = end << 32 | ~distance, descending: end falls, distance falls
    TWave w[kTWMax];
    unsigned long long carry[kFrontMax + 8];    // front after the last finished position
    unsigned long long tmp[2 * kFrontMax + 300];
    uint32_t carry_n;
    uint32_t turn;                              // blocks are taken in turn: the wave whose sequence number this is cuts the next block
    uint32_t cursor;                            // first position not yet in a block
    uint32_t carry_seq;                         // ... and take the carry in that order: the block with this sequence number merges it into its fronts
    uint32_t emit_seq;                          // ... and publish their records in that order (t_out): the block with this sequence number, once it has written them
    uint32_t stop;
    uint32_t left, sum_blocks, sum_slow;        // waves that have left the launch; their blocks, and those of them on the serial path
};

NLZM_HD unsigned long long fr_key(uint32_t e, uint32_t d) { return ((unsigned long long)e << 32) | (0xFFFFFFFFu - d); }
NLZM_HD uint32_t fr_end(unsigned long long k) { return (uint32_t)(k >> 32); }
NLZM_HD uint32_t fr_dist(unsigned long long k) { return 0xFFFFFFFFu - (uint32_t)k; }

struct Table {
    Geom g;
    Globals G;
    GlobalsV2 V;
    uint32_t err;
    uint32_t p_pos_seen;
    uint32_t fcap = kFrCap, nwaves = kTW;       // this launch's shape
    XW_FN unsigned long long *fronts(uint32_t buf) const { return xw::lds<TLds>()->arena + (unsigned long long)(2 * xw::wave() + buf) * 64 * fcap; }
    unsigned long long n_blocks, n_slow, t_wait;
    unsigned long long n_fr8 = 0, n_fr12 = 0, n_fr16 = 0, n_fr20 = 0, n_fr24 = 0;      // blocks in which some lane's front exceeded 8 / 12 / 16 / 20 / 24 entries at some step
    unsigned long long tt0 = 0, tt1 = 0, tt2 = 0, tt3 = 0, tt4 = 0;     // profile build: gather, scan, carry merge (in block order), wait for the carry, records
#ifdef NLZM_PROFILE
    XW_FN unsigned long long ptick() const { return xw::tick(); }
#else
    XW_FN unsigned long long ptick() const { return 0; }
#endif

    // merge two fronts (each sorted by descending key, at most na / nb entries at pa / pb with stride 1) into out;
    // entries of b that end before `low` are left out.  Returns the count, or kNone if it exceeds cap.
    static XW_FN uint32_t merge(const unsigned long long *pa, uint32_t na, const unsigned long long *pb, uint32_t nb, uint32_t low,
                                unsigned long long *out, uint32_t cap)
    {
        while (nb && fr_end(pb[nb - 1]) < low) nb--;            // (ends fall along the list: the expired ones are at its tail)
        while (na && fr_end(pa[na - 1]) < low) na--;
        uint32_t ia = 0, ib = 0, no = 0, dmin = kNone;
        unsigned long long ka = na ? pa[0] : 0, kb = nb ? pb[0] : 0;
        while (ia < na || ib < nb) {
            const bool ta = ib >= nb || (ia < na && ka >= kb);
            const unsigned long long k = ta ? ka : kb;
            if (ta) { ia++; ka = ia < na ? pa[ia] : 0; } else { ib++; kb = ib < nb ? pb[ib] : 0; }
            const uint32_t d = fr_dist(k);
            if (d < dmin) {                                     // not dominated by an entry that ends at least as late
                if (no >= cap) return kNone;
                out[no++] = k; dmin = d;
            }
        }
        return no;
    }

    // pairs of position a into a list (unsorted); returns the count or kNone (more than cap)
    XW_FN uint32_t gather(uint32_t a, uint32_t cap_len, const uint32_t *rec, unsigned long long *out, uint32_t cap)
    {
        uint32_t n = 0;
        const uint32_t w0 = rec[0], np = w0 & 7u;
        for (uint32_t k = 0; k < np; k++) {
            if (n >= cap) return kNone;
            out[n++] = fr_key(a + rec[3 + 2 * k], rec[2 + 2 * k]);
        }
        if (w0 & kFtTop) { if (n >= cap) return kNone; out[n++] = fr_key(a + rec[15], rec[14]); }
        if (w0 & kFtBt) {
            // the worker lane's record: the record-setters of the descent, lengths and distances growing along the list
            // (count and first four pairs in one 64-byte record: requested together)
            const unsigned long long bi = (unsigned long long)(a - G.batch_a0);
            const uint32_t *br = G.bt_ready + bi * kBtRec;
            // (the finder stage saw quad 0 before it wrote this position's record; quads 1..3 were stored before quad 0 and carry a tag:
            //  looked at again in the rare case that one is not there yet)
            uint32_t bw[16];
            for (;;) {
#pragma unroll
                for (int k = 0; k < 16; k++) bw[k] = xw::ld_agent(br + k);
                if ((bw[7] & bw[11] & bw[15] & kBtTag) != 0) break;
                if (xw::ld_agent(&V.hx->err)) break;
                xw::pause();
            }
            const uint32_t cnt = bw[0] & 0x1FFu;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                if (k >= cnt) continue;
                const uint32_t d = bw[bt_rec_d(k)], l = bw[bt_rec_l(k)];
                if (l >= cap_len) continue;                     // as long as the lookahead allows: the finder stage's top entry covers it
                if (n >= cap) return kNone;
                out[n++] = fr_key(a + l, d);
            }
            const uint32_t *pairs = G.bt_pairs + bi * (2 * G.bt_pstride);
            // (pairs beyond the position's reservation: in the extension block whose index + 1 is the record's word 14)
            // (a launch that used its arena up has positions whose block index lies beyond it: their pairs were never stored -- the launch is reported
            //  as failed and the stream made again, nlzm_hip.cpp; nothing outside the arena is read meanwhile)
            const bool ext_ok = cnt > G.bt_pstride && bw[14] - 1u < G.bt_ext_cap;
            const uint32_t *extp = ext_ok ? G.bt_ext + (unsigned long long)(bw[14] - 1) * (2 * (kBtMaxPairs - G.bt_pstride)) : pairs;
            const uint32_t cnt_there = (cnt > G.bt_pstride && !ext_ok) ? G.bt_pstride : cnt;
            for (uint32_t k = 4; k < cnt_there; k++) {
                const uint32_t *q = k < G.bt_pstride ? pairs + 2 * k : extp + 2 * (k - G.bt_pstride);
                const uint32_t d = xw::ld_agent(q), l = xw::ld_agent(q + 1);
                if (l >= cap_len) continue;
                if (n >= cap) return kNone;
                out[n++] = fr_key(a + l, d);
            }
        }
        return n;
    }

    // sort a short list by descending key and drop dominated entries, in place; returns the count
    static XW_FN uint32_t sort_filter(unsigned long long *v, uint32_t n)
    {
        for (uint32_t x = 1; x < n; x++) {
            const unsigned long long k = v[x];
            uint32_t y = x;
            while (y > 0 && v[y - 1] < k) { v[y] = v[y - 1]; y--; }
            v[y] = k;
        }
        uint32_t no = 0, dmin = kNone;
        for (uint32_t x = 0; x < n; x++) {
            const uint32_t d = fr_dist(v[x]);
            if (d < dmin) { v[no++] = v[x]; dmin = d; }
        }
        return no;
    }

    // write the parser's record of position a from its front (fn entries at f, descending)
    XW_FN void emit(uint32_t a, uint32_t a1, uint32_t lit, const unsigned long long *f, uint32_t fn)
    {
        uint32_t *rec = V.tp + (unsigned long long)(a & (kTpRing - 1)) * kTpStride;
        uint32_t *fo = V.tf + (unsigned long long)(a & (kTpRing - 1)) * kTfStride;
        const uint32_t mt_max = fn ? fr_end(f[0]) - a : 0u;
        uint32_t max_len = umin(mt_max, a1 - a);                // :1545-1548 (the 4096 cap is the parser's: it knows the segment)
        if (max_len < kMatchMin) max_len = 0;
        uint32_t ne = 0, uniq = 0, dprev = 0;
        if (max_len) {
            uint32_t step = (max_len - kMatchMin) >> 4;         // :1558-1560
            step += step == 0;
            uint32_t j = 0;
            unsigned long long kj = f[0], kn = fn > 1 ? f[1] : 0;
            unsigned long long held = 0;                        // the edge before, until it goes out with its neighbour
            for (uint32_t tl = max_len; tl >= kMatchMin; tl -= umin(tl, step)) {
                while (j + 1 < fn && fr_end(kn) >= a + tl) { j++; kj = kn; kn = j + 1 < fn ? f[j + 1] : 0; }   // the entry with the smallest end >= a + tl
                const uint32_t d = fr_dist(kj), mm = match_min(d);
                uint32_t nx, ex;
                const uint32_t slot = dist_slot(d - 1, nx, ex);
                const uint32_t valid = tl >= mm ? 1u : 0u, lv = valid ? tl - mm : 0u;
                const uint32_t at = tl | (lv << 9) | (slot << 18) | (nx << 24) | (valid << 31);
                if (ne & 1u) xw::st_agent128(rec + kTpEdges + 2 * (ne - 1), (uint32_t)held, (uint32_t)(held >> 32), d, at);
                else held = (unsigned long long)d | ((unsigned long long)at << 32);
                if (valid && d != dprev) uniq |= 1u << ne;      // the valid samples that bring a distance the one before did not have
                if (valid) dprev = d;
                ne++;
            }
            if (ne & 1u) xw::st_agent64((unsigned long long *)(rec + kTpEdges + 2 * (ne - 1)), held);
        }
        xw::st_agent128(rec, ne | (lit << 8) | (mt_max << 16), fn, uniq, 0u);
        for (uint32_t k = 0; k + 1 < fn; k += 2)
            xw::st_agent128(fo + 2 * k, fr_end(f[k]) - a, fr_dist(f[k]), fr_end(f[k + 1]) - a, fr_dist(f[k + 1]));
        if (fn & 1u) xw::st_agent64((unsigned long long *)(fo + 2 * (fn - 1)), (unsigned long long)(fr_end(f[fn - 1]) - a) | ((unsigned long long)fr_dist(f[fn - 1]) << 32));
    }

    XW_FN void capture(uint32_t a, const unsigned long long *f, uint32_t fn);

    // positions [a0, a0 + n) of a chunk whose positions end at a1 and whose lookahead ends at la_end
    // wait for an LDS word of the stage to reach v
    XW_FN bool wait_lds(const uint32_t *p, uint32_t v)
    {
        uint32_t spins = 0;
        while (xw::readfirst(xw::lds_ld(p)) != v) {
            if ((++spins & 255u) == 0 && xw::readfirst(xw::ld_agent(&V.hx->err))) return false;
            xw::pause();
        }
        xw::after_poll();
        return true;
    }

    // positions [a0, a0 + n) of a chunk whose positions end at a1 and whose lookahead ends at la_end; seq: the block's number
    XW_FN void block(uint32_t a0, uint32_t n, uint32_t a1, uint32_t la_end, uint32_t seq)
    {
        TLds *L = xw::lds<TLds>();
        TWave *W = &L->w[xw::wave()];
        const uint32_t i = xw::lane();
        const bool in_blk = i < n;
        const uint32_t a = a0 + i;
        const uint32_t cap_len = in_blk ? umin(la_end - a, kMatchMax) : 0u;
        const uint32_t *rec = V.ft + (unsigned long long)(a & (kFtRing - 1)) * kFtStride;
        uint32_t *r = W->recs + i * kFtStride;
        if (in_blk) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const unsigned long long w = xw::ld_agent64((const unsigned long long *)(rec + 2 * k));
                r[2 * k] = (uint32_t)w; r[2 * k + 1] = (uint32_t)(w >> 32);
            }
        } else r[0] = 0;
        const uint32_t lit = (r[0] >> 8) & 0xFFu;
        W->overflow = 0;
        xw::wave_sync();
        const unsigned long long q0 = ptick();
        // the position's own pairs as a front
        unsigned long long *mine = fronts(0) + i * fcap;
        uint32_t cnt = 0;
        if (in_blk) {
            cnt = gather(a, cap_len, r, mine, fcap);
            if (cnt == kNone) { W->overflow = 1; cnt = 0; }
            else cnt = sort_filter(mine, cnt);
        }
        const unsigned long long q1 = ptick();
        // prefix scan: after the step with offset D lane i holds the front of the pairs of lanes (i - 2D, i]
        uint32_t cur = 0, cmax = cnt;                               // (cmax: the largest front this lane holds at any step -- the histogram below)
        for (uint32_t D = 1; D < 64; D <<= 1) {
            xw::wave_sync();
            const uint32_t ocnt = xw::shfl_up(cnt, D);
            unsigned long long *dst = fronts(cur ^ 1) + i * fcap;
            const unsigned long long *own = fronts(cur) + i * fcap;
            uint32_t nn;
            if (i >= D && in_blk) nn = merge(own, cnt, fronts(cur) + (i - D) * fcap, ocnt, a + 1, dst, fcap);
            else { for (uint32_t k = 0; k < cnt; k++) dst[k] = own[k]; nn = cnt; }
            if (nn == kNone) { W->overflow = 1; nn = 0; }
            cnt = nn;
            cmax = umax(cmax, cnt);
            cur ^= 1;
        }
        xw::wave_sync();
        const unsigned long long q2 = ptick();
        // ---- from here on in block order: the front carried into the block
        if (G.cap_words && !wait_lds(&L->emit_seq, seq)) { err = 1; return; }     // (stage test tap: one list, appended to in position order -- whole blocks in turn)
        if (!wait_lds(&L->carry_seq, seq)) { err = 1; return; }
        const unsigned long long q3 = ptick();
        const uint32_t cn = xw::readfirst(L->carry_n);
        unsigned long long *fin_f = fronts(cur ^ 1) + i * fcap;
        uint32_t fn = 0;
        if (in_blk) {
            fn = merge(fronts(cur) + i * fcap, cnt, L->carry, cn, a + 1, fin_f, fcap);
            if (fn == kNone) { W->overflow = 1; fn = 0; }
        }
        xw::wave_sync();
        n_blocks++;
        cmax = umax(cmax, fn == kNone ? 0u : fn);
        n_fr8 += xw::any(cmax > 8); n_fr12 += xw::any(cmax > 12); n_fr16 += xw::any(cmax > 16); n_fr20 += xw::any(cmax > 20); n_fr24 += xw::any(cmax > 24);
        // Only the merge with the carried front has to happen in block order: the next block may take its carry -- this block's last
        // front -- as soon as that exists, and the records (the bulk of what is left: the sampled lengths of 64 positions) are written
        // beside the next block's merge.  (Round 4 kept the whole tail in order: 435 of a block's ~1,500 cycles per position, profile
        // build, were serial -- the stage stood at 420 - 440 however many waves it had.)
        const bool slow = xw::readfirst(W->overflow) != 0;
        if (slow) slow_block(a0, n, a1, la_end);
        else {
            // carry out: the last position's front (every lane has finished its merge with the old one: wave_sync above)
            const uint32_t last_n = xw::readlane(fn, n - 1);
            for (uint32_t k = i; k < last_n; k += 64) L->carry[k] = fronts(cur ^ 1)[(n - 1) * fcap + k];
            if (i == 0) L->carry_n = last_n;
        }
        xw::wave_sync();
        if (i == 0) xw::lds_st(&L->carry_seq, seq + 1);
        const unsigned long long q4 = ptick();
        if (!slow) {
            if (in_blk) emit(a, a1, lit, fin_f, fn);
#ifdef NLZM_SIM
            if (in_blk) sim_on_front(G.hook_user, a, fin_f, fn);
#endif
        }
        xw::drain();
        // the records are out: said in block order (t_out covers every position below it)
        if (!wait_lds(&L->emit_seq, seq)) { err = 1; return; }
        if (!slow && G.cap_words) {                                 // (stage test tap: appends to one list, so in order too)
            for (uint32_t j = 0; j < n; j++) capture(a0 + j, fronts(cur ^ 1) + j * fcap, xw::readlane(fn, j));
        }
        xw::wave_sync();
        if (i == 0) { xw::st_agent(&V.hx->t_out, a0 + n); xw::st_agent(&V.hx->t_pos, a0 + n); xw::lds_st(&L->emit_seq, seq + 1); }
        tt0 += q1 - q0; tt1 += q2 - q1; tt3 += q3 - q2; tt2 += q4 - q3; tt4 += ptick() - q4;
    }

    // a block with a front of more than fcap entries: position by position (every lane runs the same loop; rare)
    XW_FN void slow_block(uint32_t a0, uint32_t n, uint32_t a1, uint32_t la_end)
    {
        TLds *L = xw::lds<TLds>();
        TWave *W = &L->w[xw::wave()];
        n_slow++;
        for (uint32_t j = 0; j < n; j++) {
            const uint32_t a = a0 + j;
            const uint32_t cap_len = umin(la_end - a, kMatchMax);
            const uint32_t *rec = V.ft + (unsigned long long)(a & (kFtRing - 1)) * kFtStride;
            uint32_t *r = W->recs;
            if (xw::lane() < 8) {
                const unsigned long long w = xw::ld_agent64((const unsigned long long *)(rec + 2 * xw::lane()));
                r[2 * xw::lane()] = (uint32_t)w; r[2 * xw::lane() + 1] = (uint32_t)(w >> 32);
            }
            xw::wave_sync();
            unsigned long long *t0 = L->tmp, *t1 = L->tmp + 300;
            if (xw::lane() == 0) {
                uint32_t c = gather(a, cap_len, r, t0, 300);    // <= 7 + 256 pairs
                c = sort_filter(t0, c);
                const uint32_t fn = merge(t0, c, L->carry, L->carry_n, a + 1, t1, kFrontMax + 8);
                for (uint32_t k = 0; k < fn; k++) L->carry[k] = t1[k];
                L->carry_n = fn;
                emit(a, a1, (r[0] >> 8) & 0xFFu, L->carry, fn);
#ifdef NLZM_SIM
                sim_on_front(G.hook_user, a, L->carry, fn);
#endif
            }
            xw::wave_sync();
            if (G.cap_words) capture(a, L->carry, xw::readfirst(L->carry_n));
        }
    }

#ifdef NLZM_SIM
    static void sim_on_front(void *user, uint32_t a, const unsigned long long *f, uint32_t fn);
#endif

    XW_FN void run(uint32_t c0, uint32_t c1)
    {
        TLds *L = xw::lds<TLds>();
        StateV2 *S = (StateV2 *)V.state;
        const uint32_t i = xw::lane(), w = xw::wave();
        const uint32_t a_first = (uint32_t)((unsigned long long)c0 * g.chunk_size);
        unsigned long long a_last = (unsigned long long)c1 * g.chunk_size;
        if (a_last > g.n) a_last = g.n;
        const uint32_t wide = xw::readfirst(S->tb_wide[G.launch_par & 1u]);    // (what the launch before left; the kernel let nwaves waves in by the same word)
        fcap = wide ? kFrCapWide : kFrCap; nwaves = table_waves(wide);
        if (w == 0) {
            const uint32_t cn = xw::readfirst(S->front_n);
            for (uint32_t k = i; k < cn; k += 64) L->carry[k] = fr_key(S->front[2 * k], S->front[2 * k + 1]);
            if (i == 0) { L->carry_n = cn; L->turn = 0; L->cursor = a_first; L->carry_seq = 0; L->emit_seq = 0; L->stop = 0; L->left = 0; L->sum_blocks = 0; L->sum_slow = 0; }
        }
        xw::block_sync();
        err = 0; n_blocks = n_slow = 0; t_wait = 0;
        const unsigned long long t_start = xw::tick();
        uint32_t p_seen = a_first, f_seen = a_first;
        for (uint32_t seq = w; !err; seq += nwaves) {
            // ---- take the next block (in turn)
            const unsigned long long tw = xw::tick();
            if (!wait_lds(&L->turn, seq)) { err = 1; break; }
            const uint32_t a = xw::readfirst(xw::lds_ld(&L->cursor));
            if ((unsigned long long)a >= a_last) { if (i == 0) xw::lds_st(&L->turn, seq + 1); break; }
            const uint32_t ci = a / g.chunk_size;
            const unsigned long long chunk_abs = (unsigned long long)ci * g.chunk_size;
            const unsigned long long remain = g.n - chunk_abs;
            const uint32_t chunk_read = (uint32_t)(remain < g.feed ? remain : g.feed);
            const uint32_t a1 = (uint32_t)chunk_abs + umin(g.chunk_size, chunk_read), la_end = (uint32_t)chunk_abs + chunk_read;
            if ((int32_t)(f_seen - a) <= 0) {
                if (!wait_word_ge(&V.hx->f_pos, a + 1, V.hx, 2)) { err = 1; break; }
            }
            f_seen = xw::readfirst(xw::ld_agent(&V.hx->f_pos));
            xw::after_poll();
            const uint32_t n = umin(64u, umin(a1, f_seen) - a);
            if ((int32_t)(a + n - p_seen - kTpRing) > 0) {
                if (!wait_word_ge(&V.hx->p_pos, a + n - kTpRing, V.hx, 3)) { err = 1; break; }
                p_seen = xw::readfirst(xw::ld_agent(&V.hx->p_pos));
            }
            if (i == 0) { xw::lds_st(&L->cursor, a + n); xw::lds_st(&L->turn, seq + 1); }
            t_wait += xw::tick() - tw;
            block(a, n, a1, la_end, seq);
            xw::trace(3, a, n);
        }
        if (err && i == 0) raise(V.hx, kErrInternal + 200, kStTable, 9, xw::lds_ld(&L->cursor));   // (only if no stage has raised anything: the first code stays)
        xw::block_sync();
        if (w == 0 && i == 0 && xw::ld_agent(&V.hx->err)) {     // where this stage was when it left
            uint32_t *d = V.hx->dbg[1];
            xw::st_agent(d + 0, xw::lds_ld(&L->cursor)); xw::st_agent(d + 1, xw::lds_ld(&L->turn)); xw::st_agent(d + 2, xw::lds_ld(&L->carry_seq));
            xw::st_agent(d + 3, f_seen); xw::st_agent(d + 4, p_seen); xw::st_agent(d + 5, L->carry_n);
        }
        if (i == 0) {       // accounting: summed over the waves
            unsigned long long *pr = G.persist->prof;
            xw::atomic_add64_agent(&pr[6], n_blocks); xw::atomic_add64_agent(&pr[7], n_slow);
            xw::atomic_add64_agent(&pr[105], n_fr8); xw::atomic_add64_agent(&pr[106], n_fr12); xw::atomic_add64_agent(&pr[107], n_fr16);
            xw::atomic_add64_agent(&pr[108], n_fr20); xw::atomic_add64_agent(&pr[109], n_fr24);
            xw::atomic_add64_agent(&pr[44], tt0); xw::atomic_add64_agent(&pr[45], tt1);
            xw::atomic_add64_agent(&pr[46], tt2); xw::atomic_add64_agent(&pr[47], tt3); xw::atomic_add64_agent(&pr[55], tt4);
            if (w == 0) { xw::atomic_add64_agent(&pr[18], t_wait); xw::atomic_add64_agent(&pr[19], xw::tick() - t_start); }
        }
        if (w == 0) {
            const uint32_t on = xw::readfirst(L->carry_n);
            for (uint32_t k = i; k < on; k += 64) { S->front[2 * k] = fr_end(L->carry[k]); S->front[2 * k + 1] = fr_dist(L->carry[k]); }
            if (i == 0) S->front_n = on;
        }
        // the shape of the next launch: by the share of this launch's blocks that took the serial path (the last wave to leave sees every wave's count)
        if (i == 0) { xw::lds_add(&L->sum_blocks, (uint32_t)n_blocks); xw::lds_add(&L->sum_slow, (uint32_t)n_slow); }
        xw::wave_sync();                                        // (the counts, then the word that says they are there)
        if (i == 0) {
            if (xw::lds_inc(&L->left) + 1 == nwaves) {
                const uint32_t nb = xw::lds_ld(&L->sum_blocks), ns = xw::lds_ld(&L->sum_slow);
                uint32_t nxt = wide;
                if (!wide && ns > (nb >> kTbWidenShift)) nxt = 1;
                if (wide && ns < (nb >> kTbNarrowShift)) nxt = 0;
                if (G.table_shape) nxt = G.table_shape - 1;        // (option `table_shape`: 1 narrow, 2 wide; 0: by the data)
                S->tb_wide[(G.launch_par & 1u) ^ 1u] = nxt;
                if (nxt != wide) xw::atomic_add64_agent(&G.persist->prof[113], 1ull);
                if (wide) xw::atomic_add64_agent(&G.persist->prof[114], 1ull);
            }
        }
    }
};

// stage test tap: {position, max_len, delta[2..max_len]} (what the reference copies into mt_carry, :1543)
XW_FN void Table::capture(uint32_t a, const unsigned long long *f, uint32_t fn)
{
    if (a < G.cap_lo || a >= G.cap_hi) return;
    const uint32_t mt_max = fn ? fr_end(f[0]) - a : 0u;
    const unsigned long long used = xw::readfirst64(*G.cap_used);
    const unsigned long long need = 2 + (mt_max >= 2 ? mt_max - 1 : 0);
    if (used + need > G.cap_cap) { err = kErrCapture; if (xw::lane() == 0) raise(V.hx, kErrCapture, kStTable, 10, a); return; }
    if (xw::lane() == 0) { G.cap_words[used] = a; G.cap_words[used + 1] = mt_max; }
    for (uint32_t l = 2 + xw::lane(); l <= mt_max; l += 64) {
        uint32_t d = 0;
        for (uint32_t k = 0; k < fn; k++) if (fr_end(f[k]) >= a + l) d = fr_dist(f[k]);    // the last entry that still reaches
        G.cap_words[used + l] = d;
    }
    xw::drain();
    if (xw::lane() == 0) *G.cap_used = used + need;
    xw::drain();
    xw::wave_sync();
}


// =================================================================================================
// parser stage
// =================================================================================================
// parse_table (:1464-1651) visits the nodes of a segment in order; node p takes the literal edge of p-1, then relaxes
// its own sampled-length edges (dict, then rep where the distance is in the node's rep set) and its explicit rep
// probes, all with strict '>'.  For a target node the candidates therefore arrive ordered by (source, rank inside the
// source) and the first of the cheapest wins: the winner is the minimum of key = cost << 32 | source << 8 | rank.
//
// The state of node t (cost, winning edge, rep set, inside the segment or not) is a function of the states of the
// nodes < t alone, so the segment's states are the unique fixed point of "recompute every node from the others".
// A block of up to 64 nodes is iterated to that fixed point with the lanes as NODES (Jacobi passes):
//   push    every node relaxes its edges from its state of the previous pass: 64-bit LDS atomic min per edge, the
//           sampled edges split over three waves, the explicit rep probes on a fourth;
//   update  keys -> costs through the literal edges by a (min,+) prefix scan, membership by a prefix max of the
//           edges' reach, rep sets from the winners' sources;
// until a pass changes nothing.  A node is right after pass k if its winner chain has at most k match edges inside
// the block; literal runs cost no pass.  Then the block's edges that end beyond it are merged into the keys of the
// nodes to come, and the next block starts.
constexpr unsigned long long kKeyNone = ~0ull;
constexpr uint32_t kRankLit = 255, kRankProbe = 64;
constexpr uint32_t kSrcNone = 0x1FFF;
constexpr uint32_t kPW = 8;                     // waves of the stage.  A node has up to 32 sampled edges, nearly always fewer than 16:
constexpr uint32_t kEdgesPerWave = 5;           //   waves 4..6 relax five of the first fifteen each, wave 7 the sixteenth (it also loads the records
                                                //   ahead), waves 0..3 four of the rare ones each and make the explicit probe of rep slot w.
                                                //   (Round 4 dealt the rare edges to waves 4..7 instead, the probe waves being a pass's critical
                                                //   path: no difference at 300 MB, 80.6 s both ways -- rare edges cost nothing where they are.)
NLZM_HD uint32_t edge_of(uint32_t w, uint32_t j)    // sampled edge j of wave w (kMaxEdges: none)
{
    if (w < 4) return j < 4 ? 16 + w + 4 * j : kMaxEdges;
    if (w < 7) return (w - 4) + 3 * j;
    return j == 0 ? 15u : kMaxEdges;
}
constexpr uint32_t kParserThreads = 64 * kPW;
constexpr uint32_t kStagePos = 128;             // table records kept ahead in LDS: positions ...
constexpr uint32_t kStageEdges = 16;            // ... the first sampled edges of each (the rest, rare, is read from the ring)
constexpr uint32_t kStageQ = 2 + kStageEdges;   // 8-byte words per staged record: header, edges, the mask of samples with a new distance
constexpr uint32_t kGatherNodes = 32, kGatherRounds = 3;   // a block waits this many loader steps for this many records before it runs shorter
constexpr uint32_t kPumpLoads = 8;              // loads in flight per lane of the loader wave (4: 55-node blocks at 300 MB, the stage ran out of staged records; 8: 62) ...
constexpr uint32_t kPumpRecs = 64 / kStageQ;    // ... each for the words of three records: twenty-four records a step
constexpr uint32_t kInf = 0x3FFFFFFFu;
constexpr uint32_t kSpan = 64 + kMatchMax + 2;  // nodes a block's edges can end at

struct PLds {
    unsigned long long mprev[512];              // node n at [n & 511]: best key over the edges of finished blocks
    unsigned long long mcur[3][512];            // ... over the edges of the block being iterated: pass p relaxes into [p % 3]
    uint32_t nrep[512 * 4];                     // rep set of the nodes of finished blocks (CarriedState ring, :1460-1467)
    uint32_t brep[2][64 * 4];                   // ... of the block's nodes after pass p at [p & 1]
    uint32_t edge_d[512 * kMaxEdges];           // distances of the sampled edges of position a at [(a & 511) * 32 + k]
    uint32_t reach[3][64];                      // furthest node an edge or probe of the node ends at, as pass p found it at [p % 3]
    uint32_t sh[16];                            // wave 0 -> all: 0 block size, 4 error, 5 price tables stale, 6 / 9 bytes of literal edges across
                                                //   block borders, 10 / 11 price, cost of the literal edge out of the block, 12..15 model rep set; 1: staged records end here
    uint32_t node_link[kParseMax + 2];          // final nodes: from | len << 13 | cmd << 22 | rep index << 24
    uint32_t node_delta[kParseMax + 2];         // distance (dict, rep), the byte (literal)
    uint16_t cdf[kNumCtx * kCdfStride];
    uint16_t price[kNumCtx * 16];               // log2_lut[freq >> 6] per (context, symbol) (:435-438)
    uint16_t lut[256];
    uint16_t len_price[kMatchMax + 8];          // price of the length symbols by length value (:1214-1225)
    uint16_t slot_price[4 * 64];                // price of the two distance-slot symbols by (length class, slot) (:1245-1248)
    uint32_t ncmds;
    uint32_t bitbuf[64];                        // raw bits of a batch of commands (wave 7)
    uint32_t dbgw[4];                           // error dump: the segment being parsed, its block, the node count so far
    unsigned long long acc[15];                 // cycles waited / emitting / in block set-up / in passes; blocks, passes, records re-listed, put back; helper jobs posted / taken over, nodes taken over, cycles waited for the helper
    unsigned long long fpm[4];                  // single-literal runs: per probe wave, the positions where its rep slot found a match
    uint32_t ncost[512];                        // cost of the final nodes (ring like nrep: the frontier the helper's parse is compared on)
    uint32_t hj[4];                             // helper: 0 jobs posted in this launch, 1 this segment gets a job, 2 / 3 results of the take-over steps
    uint32_t stg[5];                            // loader wave: records requested up to / written up to this position, (2 unused), records of the last step; 4: records staged up to here
    unsigned long long stage[kStagePos * kStageQ];  // the table stage's records of the positions from the current block on, loaded ahead
                                                //   by wave kPW-1 (position a at [((a - launch start) & 127) * kStageQ])
    // the path of a parsed segment (node indices, end first) lives in mcur: nothing relaxes between a parse and its emission
    XW_FN uint16_t *cmdlist() { return (uint16_t *)&mcur[0][0]; }
    Counters cnt;
};

struct Parser {
    Geom g;
    Globals G;
    GlobalsV2 V;
    uint32_t base;                  // absolute offset of rebased 0
    uint32_t rep0, rep1, rep2, rep3;   // live model rep set
    uint32_t t_out_seen;
    uint32_t nsegs;                 // segments the last parse_segment() call covered (a run of single-literal segments: several)
    bool tab_dirty;
    bool is_helper = false;         // this workgroup is a helper parser (run_helper)
    uint32_t hrole = 0;             // 0: the parser stage; k + 1: helper k.  Helper hrole is the first one this workgroup may take over from
    bool prev_cut = false;          // the last segment was cut at 4,096 positions (:1469): the next one most likely is, and gets a helper job
    uint32_t myjob = 0;             // helper: the job it works on
    uint32_t hstop = 0;             // helper: 1 the job was dropped, 2 it gives the job up
    // wave kPW-1: the loader of the record stage (8-byte words counted from the launch's first position)
    uint32_t pend_t;                            // (what the loads of the last step were for, how far the stage is complete: L()->stg)
    unsigned long long pend_v[kPumpLoads];
    // frame writer (CodeFrame, :490-513)
    uint32_t *fsyms; uint8_t *fbits;
    uint32_t nsyms, nbits, word, word_bits, num_ops;       // (every wave follows the counts; `word`, the bits of the byte not yet full, is wave 7's)
    uint32_t err;
    uint32_t n_eq_fill, n_eq_rounds;            // (per launch)
    uint32_t n_cmp;                             // (per lane and launch: well below 2^32)
    unsigned long long t_s[7] = {}, t_q[5] = {};
    // the stage's accounting lives in LDS (L()->acc: it is touched once a block or less, and scalar registers are short)
    enum { kAccWait, kAccEmit, kAccSetup, kAccPass, kAccBlocks, kAccPasses, kAccRedo, kAccUndo, kAccTotal, kAccNeed, kAccAhead, kAccJobs, kAccTaken, kAccTakenNodes, kAccHelpWait, kAccN };
    XW_FN void acc(uint32_t k, unsigned long long v) { if (xw::lane() == 0) xw::lds_add64(&L()->acc[k], v); }
    unsigned long long t_work = 0, t_bar = 0, t_upd = 0, t_fill = 0, t_fin = 0, t_dirty = 0;     // profile build: this wave's push / probe work, barrier waits, update, mask fills, block end
#ifdef NLZM_PROFILE
    XW_FN unsigned long long ptick() const { return xw::tick(); }
#else
    XW_FN unsigned long long ptick() const { return 0; }
#endif

    XW_FN PLds *L() const { return xw::lds<PLds>(); }
    XW_FN void fail(uint32_t code, uint32_t info)
    {
        err = code;
        if (xw::thread() == 0) raise(V.hx, code, kStParser, 11, info);
    }
    XW_FN uint32_t price(uint32_t ctx, uint32_t y) const { return L()->price[ctx * 16 + y]; }
    XW_FN uint32_t price_len(uint32_t lv) const                     // :1214-1225
    {
        uint32_t c = price(kCtxLenDirect, umin(lv, 7));
        if (lv >= 7) { const uint32_t e = lv - 7; c += price(kCtxLenExtHi, e >> 4) + price(kCtxLenExtLo + (e >> 4), e & 15); }
        return c;
    }
    // per-model price tables of the match edges, rebuilt (by every thread of the stage) after an emit touched a length or
    // distance context.  Ends with a workgroup barrier either way.
    XW_FN void seg_tables(bool dirty)
    {
        if (dirty) {
            for (uint32_t lv = xw::thread(); lv <= kMatchMax; lv += kParserThreads) L()->len_price[lv] = (uint16_t)price_len(lv);
            for (uint32_t k = xw::thread(); k < 4 * 64; k += kParserThreads) {
                const uint32_t lc = k >> 6, slot = k & 63;
                L()->slot_price[k] = slot < 56 ? (uint16_t)(price(kCtxSlotHi + lc, slot >> 3) + price(kCtxSlotLo + lc * 8 + (slot >> 3), slot & 7)) : 0;
            }
        }
        xw::block_sync();
    }

    // ---- symbol output: model_encode_* (:1274-1367, :1428-1439) = WriteRange / WriteBits + cdf_update --------------------
    // The commands of a segment are emitted 64 at a time with the lanes as COMMANDS.  What the reference does one symbol after
    // the other splits into independent chains: a nibble CDF only sees the symbols coded in its own context, in command
    // order (a command never uses a context twice), and a symbol's place in the frame's symbol array is a prefix sum over
    // the commands.  So every wave owns a set of contexts, holds their cells in registers (16 lanes per context: cell i on
    // lane i; four contexts side by side, four such sets) and walks the commands that use one of them:
    //   wave 0  cmd                 wave 4  lit_lo[16]
    //   wave 1  lit_hi              wave 5  len_ext_lo[16]
    //   wave 2  len_direct          wave 6  dist_slot_lo[0..1][8]
    //   wave 3  len_ext_hi,         wave 7  dist_slot_lo[2..3][8]; the raw bits of the batch (WriteBits, :574-588)
    //           dist_slot_hi[4]
    // (start, freq) of a symbol is read before its update (:559-572); adaptation cell[i] += (mixin[y][i] - cell[i]) >> 7 with
    // mixin[y][i] = i <= y ? i : 16384 + i + (127 - nsy) (:284-298, :348-382); the price rows of the touched contexts (:435-438)
    // are rebuilt when the batch is done.
    XW_FN static uint32_t em_ctx(uint32_t w, uint32_t q)        // context of local index q of wave w
    {
        if (w == 0) return kCtxCmd;
        if (w == 1) return kCtxLitHi;
        if (w == 2) return kCtxLenDirect;
        if (w == 3) return q < 4 ? kCtxLenExtHi : kCtxSlotHi + (q - 4);
        if (w == 4) return kCtxLitLo + q;
        if (w == 5) return kCtxLenExtLo + q;
        return kCtxSlotLo + (w - 6) * 16 + q;
    }
    XW_FN static bool em_has(uint32_t w, uint32_t q)            // is local index q a context of wave w?
    {
        if (w < 3) return q == 0;
        if (w == 3) return q == 0 || (q >= 4 && q < 8);
        return q < 16;
    }
    // one chain: the commands of `mask` use local context qv with symbol yv, written to symbol slot pv
    XW_FN void em_chain(unsigned long long mask, uint32_t qv, uint32_t yv, uint32_t pv, uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                        uint32_t &dirty)
    {
        const uint32_t w = xw::wave(), l = xw::lane(), i = l & 15u, gl = l >> 4;
        for (; mask; mask &= mask - 1) {
            const uint32_t j = (uint32_t)__builtin_ctzll(mask);
            const uint32_t q = xw::readlane(qv, j), y = xw::readlane(yv, j), pos = xw::readlane(pv, j);
            const uint32_t st = q >> 2, gq = q & 3u, nsy = w == 3 ? (q < 4 ? 16u : 8u) : (w < 6 ? 16u : 8u);
            // (the set is picked by arithmetic: a choice between the four registers becomes ONE indexed access, which puts them into scratch)
            const uint32_t k0 = 0u - (uint32_t)(st == 0), k1 = 0u - (uint32_t)(st == 1), k2 = 0u - (uint32_t)(st == 2), k3 = 0u - (uint32_t)(st == 3);
            const uint32_t sel = (c0 & k0) | (c1 & k1) | (c2 & k2) | (c3 & k3);
            const uint32_t start = xw::readlane(sel, 16 * gq + y);
            const uint32_t nxt = y + 1 == 16 ? 16384u : xw::readlane(sel, (16 * gq + y + 1) & 63u);
            if (l == 0) fsyms[nsyms + pos] = ((nxt - start) << 16) + start;
            const int mix = i <= y ? (int)i : (int)(16384 + i + (127 - nsy));
            const uint32_t upd = (uint32_t)((int)sel + ((mix - (int)sel) >> 7)) & 0xFFFFu;
            const bool mine = gl == gq && i < nsy;
            const uint32_t ch = mine ? (upd ^ sel) : 0u;            // (what changes in the picked set, on this lane)
            c0 ^= ch & k0; c1 ^= ch & k1; c2 ^= ch & k2; c3 ^= ch & k3;
            dirty |= 1u << q;
        }
    }
    // ... of a wave with ONE context of kNsy symbols (waves 0..2: the long chains), cells on lanes 0..16
    template <uint32_t kNsy>
    XW_FN void em_chain1(unsigned long long mask, uint32_t yv, uint32_t pv, uint32_t &c0, uint32_t &dirty)
    {
        const uint32_t l = xw::lane();
        if (mask) dirty = 1;
        for (; mask; mask &= mask - 1) {
            const uint32_t j = (uint32_t)__builtin_ctzll(mask);
            const uint32_t y = xw::readlane(yv, j), pos = xw::readlane(pv, j);
            const uint32_t start = xw::readlane(c0, y), nxt = xw::readlane(c0, y + 1);
            if (l == 0) fsyms[nsyms + pos] = ((nxt - start) << 16) + start;
            const int mix = l <= y ? (int)l : (int)(16384 + l + (127 - kNsy));
            const uint32_t upd = (uint32_t)((int)c0 + ((mix - (int)c0) >> 7)) & 0xFFFFu;
            c0 = l < kNsy ? upd : c0;
        }
    }
    XW_FN void emit_commands(uint32_t ncmds, unsigned long long &n_lit, unsigned long long &n_dict, unsigned long long &n_rep)
    {
        const uint32_t e = xw::lane(), w = xw::wave(), i = e & 15u, gl = e >> 4;
        // this wave's cells
        uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, dirty = 0;
        if (w < 3) c0 = e < kCdfStride - 1 ? L()->cdf[em_ctx(w, 0) * kCdfStride + e] : 0u;       // (one context: its 17 cells on lanes 0..16)
        else if (em_has(w, gl)) c0 = L()->cdf[em_ctx(w, gl) * kCdfStride + i];
        if (em_has(w, 4 + gl)) c1 = L()->cdf[em_ctx(w, 4 + gl) * kCdfStride + i];
        if (w > 3) { c2 = L()->cdf[em_ctx(w, 8 + gl) * kCdfStride + i]; c3 = L()->cdf[em_ctx(w, 12 + gl) * kCdfStride + i]; }
        for (uint32_t k0 = ncmds; k0 > 0;) {                        // :1809-1843
            const uint32_t cnt = umin(64u, k0);
            uint32_t cmd = 3, len = 0, idx = 0, dv = 0;             // (cmd 3: no command on this lane)
            if (e < cnt) {
                const uint32_t node = L()->cmdlist()[k0 - 1 - e];
                const uint32_t link = L()->node_link[node];
                dv = L()->node_delta[node];
                cmd = (link >> 22) & 3u; len = (link >> 13) & 0x1FFu; idx = link >> 24;
            }
            const bool islit = cmd == 0, ismatch = cmd == 1, isrep = cmd == 2, haslen = ismatch || isrep;
            const uint32_t lv = haslen ? len - match_min(dv) : 0u;  // :1281, :1350
            const bool ext = haslen && lv >= 7;
            const uint32_t lc = umin(lv, 3);
            uint32_t nx = 0, ex = 0, slot = 0;
            if (ismatch) slot = dist_slot(dv - 1, nx, ex);
            const uint32_t nsym = islit ? 3u : (ismatch ? (ext ? 6u : 4u) : (isrep ? (ext ? 4u : 2u) : 0u));
            const uint32_t nbit = ismatch ? nx : (isrep ? 2u : 0u);
            const uint32_t s_inc = xw::scan_add(nsym), b_inc = xw::scan_add(nbit);
            const uint32_t sym_off = s_inc - nsym, bit_off = b_inc - nbit;
            const uint32_t tot_sym = xw::readlane(s_inc, 63), tot_bit = xw::readlane(b_inc, 63);
            // WriteBits calls (:1328-1340, :1365): none below distance 5, one up to four bits, else two; a rep index is one
            const uint32_t n_op = (uint32_t)__builtin_popcountll(xw::ballot((ismatch && nx) || isrep)) + (uint32_t)__builtin_popcountll(xw::ballot(ismatch && nx > 4));
            if (nsyms + tot_sym + 16 > G.syms_stride || nbits + (tot_bit >> 3) + 64 > G.bits_stride) { fail(kErrFrameOverflow, nsyms); return; }
            // ---- this wave's chains (symbol slots inside a command: cmd; literal: hi, lo; else: len [, ext hi, ext lo]; match: slot hi, slot lo)
            const uint32_t p_slot = sym_off + (ext ? 4u : 2u);
            if (w == 0) em_chain1<4>(xw::ballot(cmd < 3), cmd, sym_off, c0, dirty);
            else if (w == 1) em_chain1<16>(xw::ballot(islit), dv >> 4, sym_off + 1, c0, dirty);
            else if (w == 2) em_chain1<8>(xw::ballot(haslen), umin(lv, 7), sym_off + 1, c0, dirty);
            else if (w == 3) {
                em_chain(xw::ballot(ext), 0, (lv - 7) >> 4, sym_off + 2, c0, c1, c2, c3, dirty);
                em_chain(xw::ballot(ismatch), 4 + lc, slot >> 3, p_slot, c0, c1, c2, c3, dirty);
            }
            else if (w == 4) em_chain(xw::ballot(islit), (dv >> 4) & 15u, dv & 15u, sym_off + 2, c0, c1, c2, c3, dirty);
            else if (w == 5) em_chain(xw::ballot(ext), ((lv - 7) >> 4) & 15u, (lv - 7) & 15u, sym_off + 3, c0, c1, c2, c3, dirty);
            else em_chain(xw::ballot(ismatch && (lc >> 1) == w - 6), (lc & 1u) * 8 + (slot >> 3), slot & 7u, p_slot + 1, c0, c1, c2, c3, dirty);
            if (w == kPW - 1 && tot_bit) {
                // ---- raw bits, MSB first (:574-588): every field at its bit offset in a staging buffer, whole bytes out
                uint32_t *bb = L()->bitbuf;
                bb[e] = e == 0 ? word : 0u;                         // (the bits of the byte that is not full yet)
                xw::wave_sync();
                if (nbit) {
                    const uint32_t P = word_bits + bit_off;
                    const unsigned long long v = (unsigned long long)(ismatch ? ex : idx) << (64 - nbit - (P & 31u));
                    xw::lds_or(&bb[P >> 5], (uint32_t)(v >> 32));
                    if ((uint32_t)v) xw::lds_or(&bb[(P >> 5) + 1], (uint32_t)v);
                }
                xw::wave_sync();
                const uint32_t tb = word_bits + tot_bit, nfull = tb >> 3;
                for (uint32_t t = e; t < nfull; t += 64) fbits[nbits + t] = (uint8_t)(bb[t >> 2] >> (24 - 8 * (t & 3u)));
                word = (tb & 7u) ? ((bb[nfull >> 2] << (8 * (nfull & 3u))) & 0xFF000000u) : 0u;
                xw::wave_sync();
            }
            nbits += (word_bits + tot_bit) >> 3; word_bits = (word_bits + tot_bit) & 7u;
            nsyms += tot_sym; num_ops += tot_sym + n_op;
            n_lit += (uint32_t)__builtin_popcountll(xw::ballot(islit)); n_dict += (uint32_t)__builtin_popcountll(xw::ballot(ismatch));
            n_rep += (uint32_t)__builtin_popcountll(xw::ballot(isrep));
            if (xw::any(haslen)) tab_dirty = true;
            k0 -= cnt;
        }
        // ---- cells and price rows of the contexts this wave touched
        auto put = [&](uint32_t q, uint32_t cv) __attribute__((always_inline)) {
            const uint32_t up = xw::shfl(cv, (e + 1) & 63u);
            if (em_has(w, q) && ((dirty >> q) & 1u)) {
                const uint32_t ctx = em_ctx(w, q), nsy = ctx_nsyms(ctx);
                const uint32_t nxt = i + 1 == 16 ? 16384u : up;
                if (i < nsy) { L()->cdf[ctx * kCdfStride + i] = (uint16_t)cv; L()->price[ctx * 16 + i] = L()->lut[(nxt - cv) >> 6]; }
            }
        };
        if (w < 3) {
            const uint32_t up = xw::shfl(c0, (e + 1) & 63u);
            if (dirty) {
                const uint32_t ctx = em_ctx(w, 0), nsy = ctx_nsyms(ctx);
                if (e < nsy) { L()->cdf[ctx * kCdfStride + e] = (uint16_t)c0; L()->price[ctx * 16 + e] = L()->lut[(up - c0) >> 6]; }
            }
        } else {
            put(gl, c0); put(4 + gl, c1);
            if (w > 3) { put(8 + gl, c2); put(12 + gl, c3); }
        }
    }

    // ---- the record stage (wave kPW-1).  One step: what was requested by the last step goes into LDS, the next 64 words are
    // requested (positions below the table stage's t_out as the last step saw it, and below lo + 128: lo is the first
    // position still needed), t_out is requested again.  Nothing is waited for here except the last step's loads.
    XW_FN const unsigned long long *staged(uint32_t a) const { return L()->stage + (a & (kStagePos - 1)) * kStageQ; }
    XW_FN uint32_t staged_hi() const { return xw::readfirst(L()->stg[4]); }
    XW_FN void pump(uint32_t lo)
    {
        // (counted in records: a load instruction fetches the 18 words of three records on lanes 0..53, so that a lane's
        //  place in a record is the same in every step and no step divides by 18)
        const uint32_t i = xw::lane();
        const uint32_t ri = i / kStageQ, wi = i - ri * kStageQ;
        const bool on = i < kPumpRecs * kStageQ;
        uint32_t st_req = xw::readfirst(L()->stg[0]), st_wr = xw::readfirst(L()->stg[1]);
        const uint32_t pend_n = xw::readfirst(L()->stg[3]);
        if (pend_n) {
#pragma unroll
            for (uint32_t u = 0; u < kPumpLoads; u++) {
                const uint32_t rr = kPumpRecs * u + ri;
                if (on && rr < pend_n) L()->stage[((st_wr + rr) & (kStagePos - 1)) * kStageQ + wi] = pend_v[u];
            }
            st_wr += pend_n;
        }
        t_out_seen = xw::readfirst(pend_t);
        xw::after_poll();
        uint32_t lim = lo + kStagePos;
        if ((int32_t)(t_out_seen - lim) < 0) lim = t_out_seen;
        const uint32_t n = (int32_t)(lim - st_req) > 0 ? umin(kPumpRecs * kPumpLoads, lim - st_req) : 0u;
        const uint32_t wsrc = wi == 0 ? 0u : (wi == kStageQ - 1 ? kTpUniq / 2 : kTpEdges / 2 + wi - 1);   // (staged: header, sixteen edges, mask)
#pragma unroll
        for (uint32_t u = 0; u < kPumpLoads; u++) {
            const uint32_t rr = kPumpRecs * u + ri;
            if (on && rr < n)
                pend_v[u] = xw::ld_agent64((const unsigned long long *)(V.tp + (unsigned long long)((st_req + rr) & (kTpRing - 1)) * kTpStride) + wsrc);
        }
        pend_t = xw::ld_agent(&V.hx->t_out);
        if (i == 0) { L()->stg[0] = st_req + n; L()->stg[1] = st_wr; L()->stg[3] = n; L()->stg[4] = st_wr; }
    }
    // the loader goes on at position a (what is staged or on its way is dropped): a segment's back was taken over from the helper, or a
    // helper job starts in the middle of a segment
    XW_FN void pump_seek(uint32_t a)
    {
        if (xw::lane() == 0) { L()->stg[0] = a; L()->stg[1] = a; L()->stg[3] = 0; L()->stg[4] = a; }
        xw::wave_sync();
    }
    // helper: has the parser stage dropped the job (or posted another one, or ended the launch)?
    XW_FN bool helper_cancelled() const
    {
        const uint32_t j = xw::readfirst(xw::ld_agent(&V.hx->h_job)), v = xw::readfirst(xw::ld_agent(&V.hx->hw[hrole - 1].verdict));
        xw::after_poll();
        return j != myjob || v == ((myjob << 2) | 2u);
    }
    // until the record of position a is staged (false: another stage failed, the wait timed out, or -- helper -- the job was dropped)
    XW_FN bool stage_need(uint32_t a)
    {
        if ((int32_t)(staged_hi() - (a + 1)) >= 0) return true;
        const unsigned long long tw = xw::tick(), t0 = xw::clock100();
        uint32_t spins = 0;
        bool ok = true;
        for (;;) {
            pump(a);
            if ((int32_t)(staged_hi() - (a + 1)) >= 0) break;
            if ((++spins & 63u) == 0) {
                if (xw::readfirst(xw::ld_agent(&V.hx->err))) { ok = false; break; }
                if (is_helper && helper_cancelled()) { hstop = 1; ok = false; break; }
#ifndef NLZM_SIM
                if (xw::clock100() - t0 > 3000000000ull) {            // 30 s
                    if (xw::lane() == 0) raise(V.hx, kErrTimeout * 100 + 4, kStParser, 4, a, staged_hi(), t_out_seen);
                    ok = false; break;
                }
#else
                (void)t0;
#endif
            }
            if (!xw::readfirst(L()->stg[3])) xw::pause();
        }
        acc(kAccWait, xw::tick() - tw);
        acc(kAccNeed, 1); acc(kAccAhead, t_out_seen - a);             // (how far the table stage was ahead when this stage had to wait for its loader)
        return ok;
    }

    // The segment is within 264 of its forced cut (:1469): the sampled lengths of position a change with the smaller
    // max_len (:1545, :1558-1560).  Re-listed from the position's front into its ring record (lanes = samples; the
    // record is this stage's until p_pos passes it).
    // (its parts, so that a block's re-listings overlap: the first 64 entries of the NEXT position's front are requested before
    //  this one's are used, the header comes from the staged copy, and the stores are waited for once, by the caller -- 300 MB
    //  of text runs every segment into the forced cut, and a re-listing of its own was three round trips to HBM with the other
    //  seven waves at the barrier behind it)
    XW_FN unsigned long long resample_front(uint32_t a) const
    {
        const uint32_t *fo = V.tf + (unsigned long long)(a & (kTpRing - 1)) * kTfStride;
        return xw::ld_agent64((const unsigned long long *)(fo + 2 * xw::lane()));      // (beyond the front's size: whatever the slot holds, not looked at)
    }
    XW_FN void resample(uint32_t a, uint32_t max_len, unsigned long long ev0)
    {
        acc(kAccRedo, 1);
        uint32_t *rec = is_helper ? V.hb[hrole - 1].relist + (a & 511u) * kTpStride : V.tp + (unsigned long long)(a & (kTpRing - 1)) * kTpStride;
        const unsigned long long h = xw::readfirst64(staged(a)[0]);
        const uint32_t fn = (uint32_t)(h >> 32);
        const uint32_t *fo = V.tf + (unsigned long long)(a & (kTpRing - 1)) * kTfStride;
        uint32_t step = (max_len - kMatchMin) >> 4;
        step += step == 0;
        const uint32_t k = xw::lane();
        uint32_t ne = 0;
        if (max_len >= kMatchMin) ne = (max_len - kMatchMin) / step + 1;
        uint32_t d = 0, valid = 0;
        {   // the last entry of the front (smallest end) that still has >= tl bytes: lane j holds entry j, walked by register reads
            const uint32_t tl = k < ne ? max_len - k * step : 0xFFFFFFFFu;
            for (uint32_t j0 = 0; j0 < fn; j0 += 64) {
                const uint32_t cnt = umin(64u, fn - j0);
                const unsigned long long ev = j0 == 0 ? ev0 : (k < cnt ? xw::ld_agent64((const unsigned long long *)(fo + 2 * (j0 + k))) : 0ull);
                for (uint32_t jj = 0; jj < cnt; jj++) {
                    const unsigned long long e = xw::readlane64(ev, jj);
                    if ((uint32_t)e >= tl) d = (uint32_t)(e >> 32);
                }
            }
        }
        if (k < ne) {
            const uint32_t tl = max_len - k * step;
            const uint32_t mm = match_min(d);
            uint32_t nx, ex;
            const uint32_t slot = dist_slot(d - 1, nx, ex);
            valid = tl >= mm ? 1u : 0u;
            const uint32_t lv = valid ? tl - mm : 0u;
            const unsigned long long e = (unsigned long long)d | ((unsigned long long)(tl | (lv << 9) | (slot << 18) | (nx << 24) | (valid << 31)) << 32);
            xw::st_agent64((unsigned long long *)(rec + kTpEdges + 2 * k), e);
            if (k < kStageEdges) L()->stage[(a & (kStagePos - 1)) * kStageQ + 1 + k] = e;
        }
        // the valid samples that bring a distance the valid one before did not have (as the table stage lists them)
        const unsigned long long vm = xw::ballot(valid != 0), below = vm & ((1ull << k) - 1ull);
        const uint32_t dprev = xw::shfl(d, below ? 63u - (uint32_t)__builtin_clzll(below) : k);
        const uint32_t uniq = (uint32_t)xw::ballot(valid && (!below || d != dprev));
        if (k == 0) {
            xw::st_agent64((unsigned long long *)rec, (h & ~63ull) | ne);
            xw::st_agent64((unsigned long long *)(rec + kTpUniq), uniq);
            L()->stage[(a & (kStagePos - 1)) * kStageQ] = (h & ~63ull) | ne | (is_helper ? kStagedOwn : 0ull);
            L()->stage[(a & (kStagePos - 1)) * kStageQ + kStageQ - 1] = uniq;
        }
    }
    // the positions first + j of `m`, position first + j with max_len_of lane j's value
    XW_FN void resample_all(unsigned long long m, uint32_t first, uint32_t max_len_of, bool undo)
    {
        if (!m) return;
        uint32_t j = (uint32_t)__builtin_ctzll(m);
        unsigned long long ev = resample_front(first + j);
        while (m) {
            m &= m - 1;
            const uint32_t jn = m ? (uint32_t)__builtin_ctzll(m) : j;
            const unsigned long long evn = m ? resample_front(first + jn) : 0ull;
            resample(first + j, xw::readlane(max_len_of, j), ev);
            if (undo) acc(kAccUndo, 1);
            j = jn; ev = evn;
        }
        xw::drain();
    }

    // ---- explicit rep probe (:1598-1628): match length of (position a, distance r), at most c bytes ------------------
    // Every lane compares for itself, eight bytes a round: `own` holds the eight bytes at a (loaded once per block), `first` the
    // eight bytes at a - r (requested by the caller ahead of time, so that their latency passes behind other work).
    // Nearly every probe ends inside its first eight bytes; the others go on round by round.
    // (`own2` / `first2`: the eight bytes behind them, requested with them -- a match of eight bytes and more is common enough (a
    //  third of the passes with a new distance had one) that the second round, a dependent round trip to HBM on the waves a pass
    //  waits for, is worth sixteen bytes a request)
    XW_FN uint32_t probe_finish(bool want, uint32_t a, uint32_t r, uint32_t c, unsigned long long own, unsigned long long first,
                                unsigned long long own2, unsigned long long first2)
    {
        uint32_t len = 0;
        bool open = want && c > 0;
        unsigned long long x = first, y = own;
        if (open) {                                                 // the first eight bytes; the second eight are at hand
            const unsigned long long d = x ^ y;
            const uint32_t nb = d ? (uint32_t)__builtin_ctzll(d) >> 3 : 8u;
            len += umin(nb, c - len);
            if (nb < 8 || len >= c) open = false;
        }
        x = first2; y = own2;
        for (;;) {
            if (open) {
                const unsigned long long d = x ^ y;
                const uint32_t nb = d ? (uint32_t)__builtin_ctzll(d) >> 3 : 8u;
                len += umin(nb, c - len);
                if (nb < 8 || len >= c) open = false;
            }
            if (!xw::any(open)) break;
            n_eq_rounds++;
            if (open) { x = load64u(G.in + a + len - r); y = load64u(G.in + a + len); }
        }
        return len;
    }

    // rep set of a node from its winner (RepModel::Add of a dict edge's distance, :1160-1171) and what the emitter needs of the
    // winner: source | length << 13 | cmd << 22 | rep index << 24, and the distance (dict, rep).  The source's set: of a finished block from the
    // ring, of this block as the pass before left it.
    XW_FN void winner_set(uint32_t seg_a, uint32_t b0, uint32_t pbuf, uint32_t node, unsigned long long key, uint32_t &o0, uint32_t &o1,
                          uint32_t &o2, uint32_t &o3, uint32_t &link, uint32_t &delta) const
    {
        const uint32_t src = (uint32_t)(key >> 8) & 0xFFFFFFu, rank = (uint32_t)key & 0xFFu;
        if (src == kSrcNone) { o0 = rep0; o1 = rep1; o2 = rep2; o3 = rep3; link = kSrcNone; delta = 0; return; }   // node 0 (:1476)
        const uint32_t *sr = src >= b0 ? L()->brep[pbuf] + (src - b0) * 4 : L()->nrep + (src & 511u) * 4;
        const uint32_t s0 = sr[0], s1 = sr[1], s2 = sr[2], s3 = sr[3];
        o0 = s0; o1 = s1; o2 = s2; o3 = s3;
        uint32_t cmd = 0, len = 0, idx = 0;
        delta = 0;
        if (rank != kRankLit) {
            len = node - src;
            if (rank >= kRankProbe) { cmd = 2; idx = rank - kRankProbe; delta = idx == 0 ? s0 : (idx == 1 ? s1 : (idx == 2 ? s2 : s3)); }
            else {
                const uint32_t d = L()->edge_d[((seg_a + src) & 511u) * kMaxEdges + (rank >> 1)];
                delta = d;
                if (rank & 1u) { cmd = 2; idx = d == s0 ? 0u : (d == s1 ? 1u : (d == s2 ? 2u : 3u)); }
                else {
                    cmd = 1;
                    if (!(d == s0 || d == s1 || d == s2 || d == s3)) { o0 = d; o1 = s0; o2 = s1; o3 = s2; }
                }
            }
        }
        link = src | (len << 13) | (cmd << 22) | (idx << 24);
    }

    // the loader wave: how many nodes the block at node b0 gets (sh[0]), and this stage's progress word
    XW_FN void decide_block(uint32_t seg_a, uint32_t b0, uint32_t max_parse)
    {
        const uint32_t i = xw::lane();
        const uint32_t a_first = seg_a + b0;
        if (is_helper && helper_cancelled()) { hstop = 1; err = kErrInternal + 100; }
        else if (!stage_need(a_first)) err = kErrInternal + 100;
        uint32_t nb = umin(64u, max_parse - b0);
        uint32_t sh_hi = staged_hi();
        if ((int32_t)(sh_hi - (a_first + umin(nb, kGatherNodes))) < 0) {
            // A block costs about the same whatever its size: when this stage has caught up with the table stage, it
            // gives it a moment (bounded: the finder may be waiting for this stage's word) rather than run on a few nodes
            const unsigned long long tg = xw::tick();
            for (uint32_t round = 0; round < kGatherRounds; round++) {      // (a step waits for the loads of the one before)
                pump(a_first);
                sh_hi = staged_hi();
                if ((int32_t)(sh_hi - (a_first + umin(nb, kGatherNodes))) >= 0) break;
            }
            acc(kAccWait, xw::tick() - tg);
        }
        if ((int32_t)(sh_hi - (a_first + nb)) < 0) nb = sh_hi - a_first;
        nb = xw::test_cut(nb);                              // (identity on the device; the simulation cuts blocks at random here, as
                                                            //  the device does when this stage catches up with the table stage)
        if (i == 0) { L()->sh[0] = nb; L()->sh[4] = err; L()->dbgw[1] = b0; if (!is_helper) xw::st_agent(&V.hx->p_pos, a_first); }
    }

    // ---- helper parser, the parser stage's side (every thread of the stage) ---------------------------------------------------
    // A job: the segment's start, how far the chunk goes, the rebase offset, the model's rep set and the price tables as this stage
    // has just built them (3.4 KB), then the job's number.
    XW_FN void post_job(uint32_t seg_a, uint32_t chunk_left)
    {
        HelpBox *hb = V.hb;                                         // (the first box holds the job for every helper)
        const uint32_t tid = xw::thread();
        const uint32_t *pw = (const uint32_t *)L()->price, *lw = (const uint32_t *)L()->len_price, *sw = (const uint32_t *)L()->slot_price;
        for (uint32_t k = tid; k < kNumCtx * 8; k += kParserThreads) xw::st_agent((uint32_t *)hb->price + k, pw[k]);
        if (tid < (kMatchMax + 8) / 2) xw::st_agent((uint32_t *)hb->len_price + tid, lw[tid]);
        if (tid >= 256 && tid < 256 + 128) xw::st_agent((uint32_t *)hb->slot_price + (tid - 256), sw[tid - 256]);
        if (tid == 0) {
            xw::st_agent(&hb->seg_a, seg_a); xw::st_agent(&hb->chunk_left, chunk_left); xw::st_agent(&hb->base, base);
            xw::st_agent(&hb->rep[0], rep0); xw::st_agent(&hb->rep[1], rep1); xw::st_agent(&hb->rep[2], rep2); xw::st_agent(&hb->rep[3], rep3);
        }
        xw::drain();
        xw::block_sync();
        if (tid == 0) {
            const uint32_t job = L()->hj[0] + 1;
            L()->hj[0] = job;
            xw::st_agent(&V.hx->h_job, job);
            acc(kAccJobs, 1);
        }
    }
    // This workgroup's nodes below b0 are final; helper k started at node help_start(k).  Returns the segment's length if that helper's
    // nodes from b0 on were taken over -- the parser stage: links in node_link / node_delta, the rep set after the segment in sh[12..15],
    // the loader at the segment's end; a helper: the nodes copied into its own box, for whoever takes IT over -- and 0 if this workgroup
    // has to go on by itself.
    XW_FN uint32_t take_over(uint32_t seg_a, uint32_t b0, uint32_t k)
    {
        HelpBox *hb = V.hb + k;
        Hx::HelpWords *hw = &V.hx->hw[k];
        const uint32_t tid = xw::thread(), w = xw::wave();
        const uint32_t job = is_helper ? myjob : L()->hj[0], V0 = b0 - 1;      // (hj[0]: written in front of many barriers)
        // 1. is the helper beyond node V0?  It started when this workgroup did, 576 nodes in front of V0: if not, something held it up
        // (records it had to wait for) -- a short wait, then its job is dropped.
        if (w == 0) {                                               // (the wave polls as one: readfirst)
            bool there = false;
            for (uint32_t spins = 0; spins < 256 && !there; spins++) {
                const unsigned long long pg = xw::readfirst64(xw::ld_agent64(&hw->prog));
                there = (uint32_t)(pg >> 32) == job && ((uint32_t)pg & 0xFFFFu) > V0;
                if (!there) { if ((xw::readfirst(xw::ld_agent(&hw->state)) >> 2) == job) break; xw::pause(); }    // (given up: it will not get there)
            }
            xw::after_poll();
            if (tid == 0) { L()->hj[2] = there ? 1u : 0u; L()->hj[3] = 0; }
        }
        xw::block_sync();
        bool ok = L()->hj[2] != 0;
        // 2. the frontier: nodes V0 - 263 .. V0 (all of them the helper's: V0 - 263 > help_start(k))
        if (ok) {
            if (tid < kMatchMax) {
                const uint32_t t = V0 - tid;
                const uint32_t off = xw::ld_agent(&hb->cost[V0]) - L()->ncost[V0 & 511u];
                const uint32_t hc = xw::ld_agent(&hb->cost[t]);
                const uint32_t h0 = xw::ld_agent(&hb->rep_of[t * 4]), h1 = xw::ld_agent(&hb->rep_of[t * 4 + 1]), h2 = xw::ld_agent(&hb->rep_of[t * 4 + 2]), h3 = xw::ld_agent(&hb->rep_of[t * 4 + 3]);
                const uint32_t *mr = L()->nrep + (t & 511u) * 4;
                if (hc - L()->ncost[t & 511u] != off || h0 != mr[0] || h1 != mr[1] || h2 != mr[2] || h3 != mr[3]) xw::lds_or(&L()->hj[3], 1u);
            }
            xw::block_sync();
            ok = L()->hj[3] == 0;
        }
        if (!ok) {
            if (tid == 0) xw::st_agent(&hw->verdict, (job << 2) | 2u);
            xw::block_sync();
            return 0;
        }
        // 3. until the helper is through: what it has found to be inside the segment is the parser stage's word to the finder stage now
        // (a nice region that starts there waits for it, :1529); a helper passes it on in its own progress word
        if (w == 0) {
            const unsigned long long t0 = xw::clock100(), tw0 = xw::tick();
            uint32_t res = 0, cover = 0, spins = 0;
            for (;;) {
                const uint32_t st = xw::readfirst(xw::ld_agent(&hw->state));
                const unsigned long long pg = xw::readfirst64(xw::ld_agent64(&hw->prog));
                if ((uint32_t)(pg >> 32) == job) {
                    const uint32_t cv = (uint32_t)(pg >> 16) & 0xFFFFu;
                    if (cv > cover) {
                        cover = cv;
                        if (tid == 0) {
                            if (is_helper) xw::st_agent64(&V.hx->hw[hrole - 1].prog, ((unsigned long long)myjob << 32) | (cv << 16) | b0);
                            else xw::st_agent64(&V.hx->p_seg, ((unsigned long long)seg_a << 32) | (seg_a + cv));
                        }
                    }
                }
                if ((st >> 2) == job) { res = (st & 3u) == 1u ? 1u : 0u; break; }
                if ((++spins & 63u) == 0) {
                    if (xw::readfirst(xw::ld_agent(&V.hx->err))) break;
                    if (is_helper && helper_cancelled()) { hstop = 1; break; }
#ifndef NLZM_SIM
                    if (xw::clock100() - t0 > 3000000000ull) { if (tid == 0) raise(V.hx, kErrTimeout * 100 + 6, kStParser, 6, seg_a, (uint32_t)pg, st); break; }
#else
                    (void)t0;
#endif
                }
                xw::pause();
            }
            xw::after_poll();
            if (tid == 0) L()->hj[2] = res;
            acc(kAccHelpWait, xw::tick() - tw0);
        }
        xw::block_sync();
        if (!L()->hj[2]) {                                          // (given up behind the comparison: a segment that ends inside its last 264 + 64 nodes)
            if (tid == 0) xw::st_agent(&hw->verdict, (job << 2) | 2u);
            xw::block_sync();
            return 0;
        }
        // 4. its nodes
        const uint32_t seg_len = xw::ld_agent(&hb->seg_len);
        if (is_helper) {
            HelpBox *own = V.hb + (hrole - 1);
            for (uint32_t t = b0 + tid; t <= seg_len; t += kParserThreads) {
                xw::st_agent(&own->link[t], xw::ld_agent(&hb->link[t])); xw::st_agent(&own->delta[t], xw::ld_agent(&hb->delta[t]));
                if (t < seg_len)
                    for (uint32_t q = 0; q < 4; q++) xw::st_agent(&own->cmpw[q][t], xw::ld_agent(&hb->cmpw[q][t]));
            }
            xw::drain();
        } else {
            for (uint32_t t = b0 + tid; t <= seg_len; t += kParserThreads) {
                L()->node_link[t] = xw::ld_agent(&hb->link[t]); L()->node_delta[t] = xw::ld_agent(&hb->delta[t]);
                if (t < seg_len) n_cmp += xw::ld_agent(&hb->cmpw[0][t]) + xw::ld_agent(&hb->cmpw[1][t]) + xw::ld_agent(&hb->cmpw[2][t]) + xw::ld_agent(&hb->cmpw[3][t]);
            }
            if (w == kPW - 1) pump_seek(seg_a + seg_len);
        }
        if (tid == 0) {                                             // (uniform addresses: a per-lane one is computed early, kept, and spilled)
            L()->sh[12] = xw::ld_agent(&hb->end_rep[0]); L()->sh[13] = xw::ld_agent(&hb->end_rep[1]);
            L()->sh[14] = xw::ld_agent(&hb->end_rep[2]); L()->sh[15] = xw::ld_agent(&hb->end_rep[3]);
            acc(kAccTaken, 1); acc(kAccTakenNodes, seg_len + 1 - b0);
        }
        xw::block_sync();
        return seg_len;
    }

    // ---- one parse segment: nodes 0.. of positions seg_a.. (every thread of the stage); returns its length, the path
    // in cmdlist (ncmds entries, end first)
    XW_FN uint32_t parse_segment(uint32_t seg_a, uint32_t max_parse, uint32_t &ncmds)
    {
        const uint32_t chunk_left = max_parse;                      // positions from seg_a to the end of the chunk
        nsegs = 1;
        max_parse = umin(max_parse, kParseMax);
        const uint32_t i = xw::lane(), w = xw::wave(), tid = xw::thread();
        const uint32_t seg_q = seg_a - base;
        const uint32_t hs = is_helper ? help_start(hrole - 1) : 0u; // the node this workgroup starts at (a helper: as if the segment began there)
        // A segment starts at seg_a: said BEFORE this stage asks for the position's record.  The finder stage may be waiting
        // for exactly this word at seg_a (a nice region that starts where the segment before was cut at 4,096 positions,
        // :1469: no edge spans the cut, so nothing else tells it) and the record of seg_a comes only after it.
        if (tid == 0 && !is_helper) xw::st_agent64(&V.hx->p_seg, ((unsigned long long)seg_a << 32) | (seg_a + 1));
        if (w == kPW - 1) {                                         // (meanwhile: the first node's record, staged)
            if (!stage_need(seg_a + hs)) err = kErrInternal + 100;
            // Does this segment get a helper job?  One that can run into the forced cut (:1469), after one that did, if the helper is idle.
            bool want_help = false;
            if (V.hb && !is_helper && chunk_left >= kParseMax && prev_cut) {
                const uint32_t posted = xw::readfirst(xw::lds_ld(&L()->hj[0]));
                want_help = true;                                   // (every helper has to be through with the job before)
                for (uint32_t k = 0; k < kHelpers; k++) want_help = want_help && (posted == 0 || (xw::readfirst(xw::ld_agent(&V.hx->hw[k].state)) >> 2) == posted);
                xw::after_poll();
            }
            if (i == 0) { L()->sh[4] = err; L()->hj[1] = want_help ? 1u : 0u; }
        }
        seg_tables(tab_dirty);                                      // (tab_dirty is the same in every wave: run_chunk, run)
        tab_dirty = false;
        const uint32_t pc_dict = price(kCtxCmd, 1), pc_rep = price(kCtxCmd, 2), pc_lit = price(kCtxCmd, 0);
        // node 0 (:1472-1482)
        if (tid == 0) {
            L()->dbgw[0] = seg_a; L()->dbgw[1] = 0; L()->dbgw[2] = max_parse;
            L()->mprev[hs & 511u] = ((unsigned long long)kSrcNone << 8) | kRankLit;
            L()->mprev[(hs + 1) & 511u] = kKeyNone;
            L()->node_link[hs] = kSrcNone;
        }
        uint32_t end_p = hs + 1, end_open = hs + 1, b0 = hs;
        uint32_t seg_len = 0;
        bool decided = false;
        if (L()->sh[4]) { err = kErrInternal + 100; return 0; }
        if (!is_helper) {
            // A position without any match is a segment of its own (more than half of all segments are): its node has no
            // sampled edge, and if none of the four rep probes finds anything (:1598-1628) the only command is the literal.
            // Literals leave the rep set alone and prices do not matter here, so a RUN of such positions is found at once
            // (lanes = the positions from seg_a on; about half of these segments follow another one directly).
            const uint32_t sh_hi = staged_hi();
            const uint32_t kmax = umin(64u, umin(sh_hi - seg_a, chunk_left));
            const uint32_t hj = i < kmax ? (uint32_t)staged(seg_a + i)[0] : 1u;
            const unsigned long long withm = xw::ballot((hj & 63u) != 0);        // (lanes >= kmax count as having a match)
            const uint32_t p0 = withm ? (uint32_t)__builtin_ctzll(withm) : 64u;    // the leading positions without any match
            if (p0) {
                uint32_t counted = 0;
                if (w < 4) {
                    const uint32_t m0 = xw::opaque(rep0), m1 = xw::opaque(rep1), m2 = xw::opaque(rep2), m3 = xw::opaque(rep3);
                    const uint32_t r = w == 0 ? m0 : (w == 1 ? m1 : (w == 2 ? m2 : m3));
                    const uint32_t pcap = umin(umin(chunk_left - umin(i, chunk_left - 1), kParseMax), kMatchMax);
                    const bool want = i < p0 && r < seg_q + i;                                      // :1601
                    uint32_t l = 0;
                    if (xw::any(want)) {
                        const unsigned long long yo = want ? load64u(G.in + seg_a + i) : 0ull, xo = want ? load64u(G.in + seg_a + i - r) : 0ull;
                        const unsigned long long yo2 = want ? load64u(G.in + seg_a + i + 8) : 0ull, xo2 = want ? load64u(G.in + seg_a + i + 8 - r) : 0ull;
                        l = probe_finish(want, seg_a + i, r, pcap, yo, xo, yo2, xo2);
                    }
                    if (want) { counted = l + (l < pcap); n_cmp += counted; }
                    const unsigned long long okm = xw::ballot(want && l >= match_min(r));
                    if (i == 0) L()->fpm[w] = okm;
                }
                xw::block_sync();
                const unsigned long long hit = L()->fpm[0] | L()->fpm[1] | L()->fpm[2] | L()->fpm[3];
                const uint32_t run = hit ? umin(p0, (uint32_t)__builtin_ctzll(hit)) : p0;           // positions that are segments of one literal
                // (fpm is written again only after another barrier: the one below, or the first block's)
                if (i >= run) n_cmp -= counted;     // (not part of the run: their probes are made, and counted, when their segment is parsed)
                if (run) {
                    if (w == 0 && i < run) {
                        L()->node_link[i + 1] = 0; L()->node_delta[i + 1] = (hj >> 8) & 0xFFu;       // literal: from the node before, the byte
                        L()->cmdlist()[run - 1 - i] = (uint16_t)(i + 1);
                    }
                    if (tid == 0) {
                        L()->sh[12] = rep0; L()->sh[13] = rep1; L()->sh[14] = rep2; L()->sh[15] = rep3;    // (literals leave the rep set alone)
                        L()->ncmds = run;
                        xw::st_agent(&V.hx->p_pos, seg_a);
                        if (run > 1) xw::st_agent64(&V.hx->p_seg, ((unsigned long long)(seg_a + run - 1) << 32) | (seg_a + run));
                    }
                    xw::block_sync();
                    ncmds = run;
                    nsegs = run;
                    return run;
                }
            }
        }
        // (helpers: the parser stage has posted the job to all of them; a helper may take over from the ones behind it)
        bool help = is_helper || L()->hj[1] != 0;                   // (written in front of seg_tables' barrier: the same in every thread)
        uint32_t next_help = hrole;                                 // the first helper this workgroup has not tried to take over from
        if (help && !is_helper) post_job(seg_a, chunk_left);
        while (!seg_len) {
            const unsigned long long ts = xw::tick();
            const unsigned long long q0 = ptick();
            if (b0 == end_p || b0 >= max_parse) {
                // node b0 is the segment's last node: no edges leave it; its key is complete
                if (tid == 0) {
                    uint32_t r0, r1, r2, r3, link, delta;
                    const unsigned long long k = L()->mprev[b0 & 511u];
                    winner_set(seg_a, b0, 0, b0, k, r0, r1, r2, r3, link, delta);
                    if (((uint32_t)k & 0xFFu) == kRankLit) delta = L()->sh[9];      // the byte of position b0 - 1
                    L()->node_link[b0] = link; L()->node_delta[b0] = delta;
                    L()->sh[12] = r0; L()->sh[13] = r1; L()->sh[14] = r2; L()->sh[15] = r3;    // the model's rep set after the segment
                    if (is_helper) { xw::st_agent(&V.hb[hrole - 1].link[b0], link); xw::st_agent(&V.hb[hrole - 1].delta[b0], delta); }
                }
                seg_len = b0;
                break;
            }
            // ---- the block: nodes b0 .. b0+nb-1 whose records are out (the first one is inside the segment: it will come).
            // Its size is the loader wave's decision -- made at the end of the block before, beside the merge of that block's
            // edges and in front of ITS last barrier, whenever the segment goes on (decided); here only for a segment's first block.
            if (!decided) {
                if (w == kPW - 1) decide_block(seg_a, b0, max_parse);
                xw::block_sync();
            }
            decided = false;
            const unsigned long long q1 = ptick();
            const uint32_t nb = L()->sh[0];
            if (L()->sh[4]) { err = kErrInternal + 100; return 0; }
            const bool inb = i < nb;
            const uint32_t node = b0 + i, a = seg_a + node;
            // within 264 of the forced cut the table is cut short (:1545): such records are re-listed first
            if (b0 + nb + kMatchMax > max_parse && max_parse == kParseMax) {
                if (w == kPW - 1) {
                    const uint32_t h0 = inb ? (uint32_t)staged(a)[0] : 0u;
                    const uint32_t ml = (h0 & 63u) ? ((uint32_t)(staged(a)[1] >> 32) & 0x1FFu) : 0u;
                    resample_all(xw::ballot(inb && ml > max_parse - node), seg_a + b0, max_parse - node, false);
                }
                xw::block_sync();
            }
            const unsigned long long own8 = (w < 4 && inb) ? load64u(G.in + a) : 0ull;     // (probe waves: the bytes at the node's position, on their way during the set-up)
            const unsigned long long own16 = (w < 4 && inb) ? load64u(G.in + a + 8) : 0ull;
            const unsigned long long q2 = ptick();
            // ---- set-up: the block's records, from the stage.  Every wave: the node's header, its distinct distances (for
            // the probes' "already met" test) and the wave's own edges; their distances also go into the ring the winners'
            // distances are looked up in.
            const uint32_t *rec = V.tp + (unsigned long long)(a & (kTpRing - 1)) * kTpStride;
            const unsigned long long *srec = staged(a);
            // (every LDS word a lane may need is requested before the first one is looked at -- one round trip, not one per
            //  dependent condition; a lane outside the block reads some staged record and drops it)
            const unsigned long long hd_w = srec[0], w1_w = srec[1], wu_w = srec[kStageQ - 1];
            if (is_helper && (hd_w & kStagedOwn)) rec = V.hb[hrole - 1].relist + (a & 511u) * kTpStride;      // (re-listed by this workgroup: its own copy)
            unsigned long long sv[kEdgesPerWave];
#pragma unroll
            for (uint32_t j = 0; j < kEdgesPerWave; j++) { const uint32_t k = edge_of(w, j); sv[j] = srec[1 + (k < kStageEdges ? k : 0u)]; }
            const unsigned long long hd = inb ? hd_w : 0ull;
            const uint32_t ne = (uint32_t)hd & 63u, lit = ((uint32_t)hd >> 8) & 0xFFu;
            const uint32_t uniq = ne ? (uint32_t)wu_w : 0u;
            const uint32_t nd = (uint32_t)__builtin_popcount(uniq);
            const uint32_t max_len = ne ? ((uint32_t)(w1_w >> 32) & 0x1FFu) : 0u;
            const uint32_t sreach = inb ? node + max_len : 0u;                      // :1550
            uint32_t ed[kEdgesPerWave], ea[kEdgesPerWave];          // this wave's edges of the node (distance; length | price words)
#pragma unroll
            for (uint32_t j = 0; j < kEdgesPerWave; j++) { ed[j] = 0; ea[j] = 0; }
            // (requested here, used after the literal prices below: an edge beyond the sixteen that are staged comes from the ring
            //  in HBM, and nearly every block has a node with one)
            unsigned long long er[kEdgesPerWave];
#pragma unroll
            for (uint32_t j = 0; j < kEdgesPerWave; j++) {
                const uint32_t k = edge_of(w, j);
                er[j] = k < ne ? (k < kStageEdges ? sv[j] : xw::ld_agent64((const unsigned long long *)(rec + kTpEdges + 2 * k))) : 0ull;
            }
            const unsigned long long q3 = ptick();
            uint32_t dd[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };            // the node's first eight distinct valid distances (probe waves)
            if (w < 4) {
                uint32_t um = uniq;
#pragma unroll
                for (uint32_t z = 0; z < 8; z++) {
                    const uint32_t k = um ? (uint32_t)__builtin_ctz(um) : 0u;
                    dd[z] = um ? (k < kStageEdges ? (uint32_t)srec[1 + k] : xw::ld_agent(rec + kTpEdges + 2 * k)) : 0u;
                    um &= um - 1;
                }
            }
            if (w == 4) L()->reach[1][i] = sreach;                   // (the relax waves have the time: the probe waves are what a block waits for)
            const unsigned long long q4 = ptick();
            const uint32_t litw = inb ? pc_lit + price(kCtxLitHi, lit >> 4) + price(kCtxLitLo + (lit >> 4), lit & 15) : 0u;   // :1418-1426
            const uint32_t S = xw::scan_add(litw) - litw;           // price of the literals of nodes b0 .. b0+i-1
            if (w == 0 && i == nb - 1) { L()->sh[6] = lit; L()->sh[10] = litw; }   // (the literal edge into node b0 + nb: its byte, its price)
            uint32_t lpv[kEdgesPerWave], spv[kEdgesPerWave];       // (the price words of all edges requested, then used: as above)
#pragma unroll
            for (uint32_t j = 0; j < kEdgesPerWave; j++) {
                const uint32_t at = (uint32_t)(er[j] >> 32);
                const uint32_t lv = (at >> 31) ? (at >> 9) & 0x1FFu : 0u, slot = (at >> 31) ? (at >> 18) & 63u : 0u;
                lpv[j] = L()->len_price[lv];
                spv[j] = L()->slot_price[umin(lv, 3) * 64 + slot];
            }
#pragma unroll
            for (uint32_t j = 0; j < kEdgesPerWave; j++) {
                const uint32_t k = edge_of(w, j);
                if (k < ne) L()->edge_d[(a & 511u) * kMaxEdges + k] = (uint32_t)er[j];
                const uint32_t at = (uint32_t)(er[j] >> 32);
                if (at >> 31) {
                    const uint32_t tl = at & 0x1FFu, nx = (at >> 24) & 31u;
                    const uint32_t lp = lpv[j];
                    const uint32_t wd = lp + (nx << 5) + spv[j];     // (+ pc_dict: :1208-1251)
                    ed[j] = (uint32_t)er[j];
                    ea[j] = tl | (wd << 9) | (lp << 21);        // wd < 4096, lp < 2048
                }
            }
            if (w >= 4) for (uint32_t t = tid - 256u; t < kSpan; t += 256u) L()->mcur[1][(b0 + t) & 511u] = kKeyNone;
            xw::block_sync();
            if (w == 0) acc(kAccSetup, xw::tick() - ts);
            t_q[0] += q1 - q0; t_q[1] += q2 - q1; t_q[2] += q3 - q2; t_q[3] += q4 - q3; t_q[4] += ptick() - q4;
            const unsigned long long tp0 = xw::tick();
            // ---- passes.  Every wave keeps the state of its lane's node in registers (all waves compute the same update)
            unsigned long long key = kKeyNone;
            uint32_t c = kInf, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
            bool lv = false;
            uint32_t istar = 64, blk_end = end_p, pass = 0;
            // this wave's rep slot: what was measured for which distance, and the probe edge it gives under the node's present set
            uint32_t mr = 0, ml = 0, cr = 0, pw = 0, pt = 0;
            bool want = false;
            const uint32_t pcap = umin(max_parse - node, kMatchMax);                                    // :1605-1606
            for (;;) {
                const uint32_t buf = pass % 3u, nbuf = (pass + 1) % 3u;
                const unsigned long long k0 = ptick();
                if (w == kPW - 1) pump(seg_a + b0);                 // (the records of the blocks to come)
                // the explicit probe of rep slot w, unless a sampled edge has met that distance (:1598-1628): what is to be
                // measured anew is decided first and its bytes are requested, so that they arrive behind the relaxation
                bool dirty = false, wnt = false, fresh = false;
                uint32_t r = 0;
                unsigned long long px = 0, px2 = 0;
                if (pass > 0 && w < 4) {
                    r = w == 0 ? r0 : (w == 1 ? r1 : (w == 2 ? r2 : r3));
                    dirty = lv && r != cr;
                    if (xw::any(dirty)) {
                        // (the bytes of a distance not measured yet are requested before the test whether a sampled edge has met it:
                        //  the test takes ~1,000 cycles of LDS reads and compares, the load's latency passes behind it; a few loads in vain)
                        if (dirty && mr != r && pcap > 0 && r < seg_q + node) { px = load64u(G.in + a - r); px2 = load64u(G.in + a + 8 - r); }
                        bool met = false;
#pragma unroll
                        for (uint32_t z = 0; z < 8; z++) met = met || dd[z] == r;
                        if (xw::any(dirty && nd > 8)) {
                            // the distinct distances beyond the eight in registers: from LDS, four requested at a time (one at a time is
                            // a round trip per distance, in every pass, on the waves a pass waits for)
                            uint32_t um = (dirty && nd > 8) ? uniq : 0u;
#pragma unroll
                            for (uint32_t z = 0; z < 8; z++) um &= um - 1;
                            while (xw::any(um != 0)) {
                                uint32_t v[4];
                                bool on[4];
#pragma unroll
                                for (uint32_t t = 0; t < 4; t++) {
                                    on[t] = um != 0;
                                    v[t] = L()->edge_d[(a & 511u) * kMaxEdges + (um ? (uint32_t)__builtin_ctz(um) : 0u)];
                                    um &= um - 1;
                                }
#pragma unroll
                                for (uint32_t t = 0; t < 4; t++) met = met || (on[t] && v[t] == r);
                            }
                        }
                        wnt = dirty && !met && r < seg_q + node;                                        // :1601
                        fresh = wnt && mr != r && pcap > 0;
                    }
                }
                if (pass > 0 && lv) {
                    // relax this wave's sampled edges of the node (:1566-1595)
#pragma unroll
                    for (uint32_t j = 0; j < kEdgesPerWave; j++) {
                        if (!ea[j]) continue;
                        const uint32_t k = edge_of(w, j);
                        const uint32_t tl = ea[j] & 0x1FFu, wd = (ea[j] >> 9) & 0xFFFu, lp = ea[j] >> 21;
                        unsigned long long *dst = &L()->mcur[buf][(node + tl) & 511u];
                        xw::lds_min64(dst, ((unsigned long long)(c + pc_dict + wd) << 32) | (node << 8) | (2 * k));
                        const uint32_t d = ed[j];
                        if (d == r0 || d == r1 || d == r2 || d == r3)
                            xw::lds_min64(dst, ((unsigned long long)(c + pc_rep + lp + (2u << 5)) << 32) | (node << 8) | (2 * k + 1));
                    }
                }
                const unsigned long long k0a = ptick();
                if (pass > 0 && w < 4) {
                    if (xw::any(dirty)) {
                        if (xw::any(fresh)) { const uint32_t l = probe_finish(fresh, a, r, pcap, own8, px, own16, px2); if (fresh) { mr = r; ml = l; } }
                        if (wnt && mr != r) { mr = r; ml = 0; }     // (nothing to compare: no room for a match)
                        if (dirty) {
                            cr = r; want = wnt; pw = 0; pt = 0;
                            if (wnt && ml >= match_min(r)) { pw = pc_rep + L()->len_price[ml - match_min(r)] + (2u << 5); pt = ml; }   // :1607, :1614
                        }
                    }
                    if (lv && pw) {
                        xw::lds_min64(&L()->mcur[buf][(node + pt) & 511u], ((unsigned long long)(c + pw) << 32) | (node << 8) | (kRankProbe + w));
                        xw::lds_max(&L()->reach[buf][i], node + pt);        // :1608-1612
                    }
                }
                const unsigned long long k0b = ptick();
                // the buffers of the next pass.  (Pass 0 relaxes nothing and its update reads the keys of finished blocks only:
                // the set-up has cleared mcur[1] and stored reach[1], and the set-up's barrier stands for this one.)
                if (pass > 0) {
                    if (w >= 4) for (uint32_t t = tid - 256u; t < kSpan; t += 256u) L()->mcur[nbuf][(b0 + t) & 511u] = kKeyNone;     // (the relax waves: see the set-up)
                    if (w == 4) L()->reach[nbuf][i] = sreach;
                }
                const unsigned long long k1 = ptick();
                if (pass > 0) xw::block_sync();
                const unsigned long long k2 = ptick();
                // ---- update: every node of the block from the keys (the same in every wave)
                unsigned long long kin = kKeyNone;
                if (inb) {
                    if (node <= end_open) kin = L()->mprev[node & 511u];
                    if (i >= 1 && pass > 0) { const unsigned long long kc = L()->mcur[buf][node & 511u]; if (kc < kin) kin = kc; }
                }
                const uint32_t mc = kin == kKeyNone ? kInf : (uint32_t)(kin >> 32);
                // cost through the literal edges: c[i] = min(mc[i], c[i-1] + litw[i-1]) = S[i] + min_{j<=i} (mc[j] - S[j])
                const uint32_t nc = (uint32_t)(xw::scan_min_i32((int32_t)mc - (int32_t)S) + (int32_t)S);
                const bool litwin = i >= 1 && nc < mc;                                                   // :1492 (the literal edge comes last: strict)
                const unsigned long long nkey = litwin ? (((unsigned long long)nc << 32) | ((node - 1) << 8) | kRankLit) : kin;
                const unsigned long long u1 = ptick();
                // membership: a node is inside while some edge of the nodes before it reaches it (:1486, :1550-1554)
                // (pass 0: the reach of the SAMPLED edges does not depend on the prices -- max_len is the table's (:1550) -- and the
                //  first node that no node before it reaches is where the unconditional prefix maximum stops covering, so the
                //  block's membership is known before anything is relaxed; only a probe's reach (:1608-1612) is found on the way)
                const uint32_t rlive = pass == 0 ? sreach : (lv ? L()->reach[buf][i] : 0u);     // (a node's reach counts once it was inside: it has relaxed its edges)
                const uint32_t pm = xw::scan_max(rlive);
                const uint32_t before = umax(xw::lane_below(pm, 0u), end_p);
                const bool inside = inb && node < before;
                const unsigned long long dead = xw::ballot(!inside);
                istar = dead ? (uint32_t)__builtin_ctzll(dead) : 64u;      // (lanes >= nb count as dead)
                const bool act = i <= istar && inb;                 // nodes b0 .. b0+istar: inside, or the segment's last node
                const bool nlv = i < istar && inb;
                const unsigned long long u2 = ptick();
                // rep sets from the winners (the sources' sets as of the last pass)
                uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, link = 0, delta = 0;
                if (act && nkey != kKeyNone) winner_set(seg_a, b0, (pass + 1) & 1u, node, nkey, o0, o1, o2, o3, link, delta);
                {   // a run of literal winners carries the set of the node in front of the run (:1498): taken from that lane
                    // in THIS pass, so that a literal run costs no pass
                    const uint32_t root = xw::scan_max(litwin ? 0u : i);
                    const uint32_t t0 = xw::shfl(o0, root), t1 = xw::shfl(o1, root), t2 = xw::shfl(o2, root), t3 = xw::shfl(o3, root);
                    if (litwin) { o0 = t0; o1 = t1; o2 = t2; o3 = t3; }
                }
                const unsigned long long u3 = ptick();
                const bool same = !act || (key == nkey && r0 == o0 && r1 == o1 && r2 == o2 && r3 == o3 && lv == nlv);
                const bool changed = xw::any(!same);
                if (act) { key = nkey; c = nc; r0 = o0; r1 = o1; r2 = o2; r3 = o3; }
                lv = nlv;
                if (w == 0 && act) { uint32_t *dr = L()->brep[pass & 1u] + i * 4; dr[0] = o0; dr[1] = o1; dr[2] = o2; dr[3] = o3; }
                {
                    const uint32_t done_n = umin(istar, nb);
                    blk_end = done_n ? umax(end_p, xw::readlane(pm, done_n - 1)) : end_p;
                }
                t_work += k1 - k0; t_bar += k2 - k1; t_upd += ptick() - k2;
                t_s[0] += k0a - k0; t_s[1] += k0b - k0a; t_s[2] += k1 - k0b; t_s[3] += u1 - k2; t_s[4] += u2 - u1; t_s[5] += u3 - u2; t_s[6] += ptick() - u3;
                if (!changed) break;
                pass++;
            }
            if (w == 0) { acc(kAccPass, xw::tick() - tp0); acc(kAccBlocks, 1); acc(kAccPasses, pass + 1); }
            const unsigned long long e0 = ptick();
            // ---- the block is at its fixed point: final nodes, the edges that end beyond it
            const uint32_t lbuf = pass % 3u;                         // the buffer the last update read
            const uint32_t done = umin(istar, nb);                   // nodes b0 .. b0+done-1 are inside
            if (w == 0) {
                const uint32_t lit_before = xw::lane_below(lit, 0u);   // the byte of the position before the node
                if (i <= istar && inb) {
                    uint32_t o0, o1, o2, o3, link, delta;
                    winner_set(seg_a, b0, (pass + 1) & 1u, node, key, o0, o1, o2, o3, link, delta);
                    if (((uint32_t)key & 0xFFu) == kRankLit && node > 0) delta = i ? lit_before : L()->sh[9];
                    L()->node_link[node] = link; L()->node_delta[node] = delta;
                    uint32_t *dr = L()->nrep + (node & 511u) * 4;
                    dr[0] = r0; dr[1] = r1; dr[2] = r2; dr[3] = r3;
                    L()->ncost[node & 511u] = c;
                    if (i == istar) { L()->sh[12] = r0; L()->sh[13] = r1; L()->sh[14] = r2; L()->sh[15] = r3; }   // (the segment's last node, if it ends here)
                    if (is_helper) {                                // what the parser stage compares and takes over (sc1 stores, drained in front of the barrier below)
                        HelpBox *hb = V.hb + (hrole - 1);
                        xw::st_agent(&hb->link[node], link); xw::st_agent(&hb->delta[node], delta); xw::st_agent(&hb->cost[node], c);
                        xw::st_agent128(&hb->rep_of[node * 4], r0, r1, r2, r3);
                    }
                }
                if (i == nb - 1) L()->sh[11] = c;
            }
            if (w < 4 && i < done && want) n_cmp += ml + (ml < pcap);  // bytes the final probe of this slot looked at (counter parity)
            if (is_helper) {
                if (w < 4 && inb) xw::st_agent(&V.hb[hrole - 1].cmpw[w][node], (i < done && want) ? ml + (ml < pcap) : 0u);
                if (w < 4) xw::drain();
            }
            xw::block_sync();
            if (istar < nb) {
                seg_len = b0 + istar;                               // the segment ends inside the block
                // The positions after it belong to the next segment, whose cut is elsewhere: what was re-listed for this
                // segment's forced cut goes back to the full sampling (:1545 with the new max_parse).
                // (the helper does not put records back -- its segment is only the parser stage's guess until that stage has compared
                //  the frontiers: it gives the job up, and the parser stage ends the segment itself)
                if (is_helper && b0 + nb + kMatchMax > max_parse && max_parse == kParseMax) hstop = 2;
                else if (b0 + nb + kMatchMax > max_parse && max_parse == kParseMax && w == kPW - 1) {
                    const uint32_t h0 = (inb && i >= istar) ? (uint32_t)staged(a)[0] : 0u;
                    const uint32_t ml = (h0 & 63u) ? ((uint32_t)(staged(a)[1] >> 32) & 0x1FFu) : 0u;
                    uint32_t full = umin((h0 >> 16) & 0x1FFu, chunk_left - node);
                    if (full < kMatchMin) full = 0;
                    resample_all(xw::ballot(inb && i >= istar && ml != full), seg_a + b0, full, true);
                }
            } else {
                // the nodes the block has made reachable: its edges that end beyond it become their keys (merged with what earlier
                // blocks left for the nodes that were open already); the literal edge of the block's last node goes the same way
                const uint32_t new_end = blk_end;
                for (uint32_t t = b0 + nb + tid; t <= new_end; t += kParserThreads) {
                    unsigned long long k = L()->mcur[lbuf][t & 511u];
                    if (t <= end_open) { const unsigned long long kp = L()->mprev[t & 511u]; if (kp < k) k = kp; }
                    if (t == b0 + nb) {
                        const unsigned long long kl = ((unsigned long long)(L()->sh[11] + L()->sh[10]) << 32) | ((b0 + nb - 1) << 8) | kRankLit;
                        if (kl < k) k = kl;                         // (equal cost: the smaller source wins, :1492 strict)
                        L()->sh[9] = L()->sh[6];
                    }
                    L()->mprev[t & 511u] = k;
                }
                if (tid == 0) {
                    if (is_helper) xw::st_agent64(&V.hx->hw[hrole - 1].prog, ((unsigned long long)myjob << 32) | (new_end << 16) | (b0 + nb));   // (the nodes' stores: drained in front of the barrier above)
                    else if (new_end != end_p) xw::st_agent64(&V.hx->p_seg, ((unsigned long long)seg_a << 32) | (seg_a + new_end));
                }
                if (new_end > end_open) end_open = new_end;
                end_p = new_end;
                b0 += nb;
                decided = !(b0 == end_p || b0 >= max_parse);        // (the same in every thread)
                if (decided && w == kPW - 1) decide_block(seg_a, b0, max_parse);
                xw::block_sync();
                // the first block border behind the next helper's warm-up: are its nodes the ones this workgroup would compute?
                if (help && next_help < kHelpers && !seg_len && !(b0 == end_p || b0 >= max_parse) && b0 > help_start(next_help) + kHelpWarm) {
                    seg_len = take_over(seg_a, b0, next_help);      // (0: no -- this workgroup goes on by itself, and may try the helper behind that one)
                    next_help++;
                }
            }
            t_fin += ptick() - e0;
        }
        xw::block_sync();
        if (is_helper) {                                            // (no path, no emission: the parser stage's, if it takes the nodes over)
            if (tid == 0 && !hstop) {
                HelpBox *hb = V.hb + (hrole - 1);
                xw::st_agent(&hb->seg_len, seg_len);
                xw::st_agent(&hb->end_rep[0], L()->sh[12]); xw::st_agent(&hb->end_rep[1], L()->sh[13]);
                xw::st_agent(&hb->end_rep[2], L()->sh[14]); xw::st_agent(&hb->end_rep[3], L()->sh[15]);
                xw::drain();
                xw::st_agent64(&V.hx->hw[hrole - 1].prog, ((unsigned long long)myjob << 32) | (seg_len << 16) | (seg_len + 1));
            }
            ncmds = 0;
            return seg_len;
        }
        if (help && tid == 0)                                       // (whatever a helper is still doing for this segment: dropped)
            for (uint32_t k = 0; k < kHelpers; k++) xw::st_agent(&V.hx->hw[k].verdict, (L()->hj[0] << 2) | 2u);
        prev_cut = seg_len == kParseMax;
        // backtrack (:1633-1650): node indices of the path, end first
        // (A walk through LDS is one dependent round trip per command, 130-200 cycles each with seven waves at the barrier below.
        //  Here the links of the 64 nodes from `cur` down come with ONE read -- lane l: node cur - l -- and the path through
        //  them is followed by register reads; the lanes of the nodes on it then write their places in the list together.)
        if (w == 0) {
            uint32_t n = 0, cur = xw::readfirst(seg_len);
            while (cur != 0) {
                const uint32_t top = cur;
                const uint32_t lnk = i <= top ? L()->node_link[top - i] & 0x1FFFu : 0u;
                unsigned long long on = 0;                          // bit l: node top - l is on the path
                uint32_t l = 0;
                do {
                    on |= 1ull << l;
                    cur = xw::readlane(lnk, l);                     // (the source of node top - l: always a node before it)
                    l = top - cur;
                } while (cur != 0 && l < 64);
                if ((on >> i) & 1ull) L()->cmdlist()[n + (uint32_t)__builtin_popcountll(on & ((1ull << i) - 1ull))] = (uint16_t)(top - i);
                n += (uint32_t)__builtin_popcountll(on);
            }
            if (i == 0) L()->ncmds = n;
        }
        xw::block_sync();
        ncmds = L()->ncmds;
        return seg_len;
    }

    // one chunk = one frame (:1782-1886)
    XW_FN void run_chunk(uint32_t ci)
    {
        const unsigned long long chunk_abs = (unsigned long long)ci * g.chunk_size;
        // (positions are 32-bit: stream_begin refuses inputs of 0xFFFF0000 bytes and more; the feed is longer than a chunk)
        const uint32_t p_end = umin(g.chunk_size, (uint32_t)g.n - (uint32_t)chunk_abs);
        fsyms = G.syms + (unsigned long long)(ci - G.chunk0) * G.syms_stride;
        fbits = G.bits + (unsigned long long)(ci - G.chunk0) * G.bits_stride;
        nsyms = 0; nbits = 0; word = 0; word_bits = 0; num_ops = 0;
        {   // (the window's size made opaque here: or twice it is kept in two registers, and spilled, for the whole launch)
            const unsigned long long wsize = (unsigned long long)xw::opaque(g.wmask) + 1;
            if (chunk_abs - base >= 2 * wsize) base += (uint32_t)wsize;         // :1786
        }
        unsigned long long n_lit = 0, n_dict = 0, n_rep = 0, n_seg = 0;
        uint32_t p = 0;
        while (p < p_end && !err) {
            uint32_t ncmds = 0;
            const uint32_t seg_a = (uint32_t)chunk_abs + p;
            const uint32_t len = parse_segment(seg_a, p_end - p, ncmds);
            if (err) break;
            n_seg += nsegs;
            {
                const unsigned long long te = xw::tick();
                if (xw::wave() == kPW - 1) pump(seg_a + len);                   // (records of the next segment: requested now, in LDS by the next step)
                emit_commands(ncmds, n_lit, n_dict, n_rep);
                if (xw::wave() == 0) acc(kAccEmit, xw::tick() - te);
            }
            xw::block_sync();
            rep0 = L()->sh[12]; rep1 = L()->sh[13]; rep2 = L()->sh[14]; rep3 = L()->sh[15];     // (the model's rep set after the segment: the set of its last node)
            if (xw::readfirst(xw::lds_ld(&L()->sh[4]))) err = L()->sh[4];
            p += len;
        }
        if (xw::wave() == kPW - 1 && xw::lane() < 4) fbits[nbits + xw::lane()] = xw::lane() == 0 ? (uint8_t)(word >> 24) : (uint8_t)0;   // bit pad of Flush (:591-597)
        nbits += 4;
        if (xw::wave() == 0) {
            if (xw::lane() == 0) {
                FrameMeta &fm = G.fmeta[ci - G.chunk0];
                fm.nsyms = nsyms; fm.nbits_bytes = nbits; fm.num_ops = num_ops; fm.out_len = 0;
                Counters &c = L()->cnt;
                c.n_literal += n_lit; c.n_dict += n_dict; c.n_rep += n_rep; c.segments += n_seg;
                c.rans_syms += nsyms; c.bit_ops += num_ops - nsyms; c.frames += 1;
            }
        }
    }

    XW_FN void run(uint32_t c0, uint32_t c1)
    {
        Persist *P = G.persist;
        const uint32_t tid = xw::thread();
        for (uint32_t k = tid; k < kNumCtx * kCdfStride; k += kParserThreads) L()->cdf[k] = P->cdf[k];
        for (uint32_t k = tid; k < 256; k += kParserThreads) L()->lut[k] = log2_lut_entry(k);
        for (uint32_t k = tid; k < sizeof(Counters) / 8; k += kParserThreads) ((unsigned long long *)&L()->cnt)[k] = 0;
        xw::block_sync();
        for (uint32_t k = tid; k < kNumCtx * 16; k += kParserThreads) {
            const uint32_t ctx = k >> 4, y = k & 15;
            const uint16_t *cell = L()->cdf + ctx * kCdfStride;
            L()->price[k] = (y < ctx_nsyms(ctx)) ? L()->lut[((uint32_t)cell[y + 1] - (uint32_t)cell[y]) >> 6] : 0;
        }
        rep0 = xw::readfirst(P->rep[0]); rep1 = xw::readfirst(P->rep[1]); rep2 = xw::readfirst(P->rep[2]); rep3 = xw::readfirst(P->rep[3]);
        base = xw::readfirst((uint32_t)P->reb_base);
        err = xw::readfirst(P->error);
        tab_dirty = true;
        t_out_seen = (uint32_t)((unsigned long long)c0 * g.chunk_size);
        if (tid < 5) L()->stg[tid] = (tid == 3 || tid == 2) ? 0u : t_out_seen;
        pend_t = t_out_seen;                                        // (as if t_out had been read before the table stage started)
        n_eq_fill = n_eq_rounds = 0; n_cmp = 0;
        if (tid < kAccN) L()->acc[tid] = 0;
        if (tid < 4) L()->hj[tid] = 0;
        prev_cut = false;
        xw::block_sync();
        if (tid == 0) L()->acc[kAccTotal] = 0ull - xw::tick();
        uint32_t ci = c0;
        for (; ci < c1 && !err; ci++) run_chunk(ci);
        if (V.hb && tid == 0) xw::st_agent(&V.hx->h_job, kHelpExit);      // (the helper leaves)
        xw::block_sync();
#ifdef NLZM_PROFILE
        if (xw::lane() == 0 && xw::wave() == kPW - 1) for (int z = 0; z < 5; z++) xw::atomic_add64_agent(&P->prof[56 + z], t_q[z]);
        if (xw::lane() == 0) { xw::atomic_add64_agent(&P->prof[64 + xw::wave()], t_work); xw::atomic_add64_agent(&P->prof[72 + xw::wave()], t_bar); xw::atomic_add64_agent(&P->prof[80 + xw::wave()], t_upd); }
        if (xw::lane() == 0 && xw::wave() < 4) {   // per wave: relax + probe work of a pass, barrier wait, update; mask fills
            xw::atomic_add64_agent(&P->prof[32 + xw::wave()], t_work); xw::atomic_add64_agent(&P->prof[36 + xw::wave()], t_bar);
            if (xw::wave() == 0) {
                xw::atomic_add64_agent(&P->prof[40], t_upd); xw::atomic_add64_agent(&P->prof[42], t_fin); xw::atomic_add64_agent(&P->prof[41], t_fill);
                for (int z = 0; z < 7; z++) { xw::atomic_add64_agent(&P->prof[48 + z], t_s[z]); t_s[z] = 0; }
            }
        }
#endif
        {   // bytes the probes looked at, mask fills, probe rounds: summed over the waves
            unsigned long long n_cmp64 = n_cmp;
            for (uint32_t d = 32; d; d >>= 1) n_cmp64 += xw::shfl64(n_cmp64, xw::lane() ^ d);      // (kept per lane)
            if (xw::lane() == 0) {
                xw::lds_add64(&L()->cnt.cmp_bytes, n_cmp64);
                xw::lds_add64(&L()->cnt.stale_ht, n_eq_fill); xw::lds_add64(&L()->cnt.stale_rk, n_eq_rounds);     // (counters the new stages do not use otherwise)
            }
        }
        xw::block_sync();
        if (xw::wave() == 0) {
            for (uint32_t k = xw::opaque(xw::lane()); k < kNumCtx * kCdfStride; k += 64) P->cdf[k] = L()->cdf[k];    // (opaque: or the index is kept, and spilled, from the start of the launch)
            if (xw::lane() == 0) {
                P->rep[0] = rep0; P->rep[1] = rep1; P->rep[2] = rep2; P->rep[3] = rep3;
                P->next_chunk = ci;
                // (this stage's counters only, agent-scope atomics: see the finder's)
                unsigned long long *pr = P->prof;
                auto add = [](unsigned long long *q, unsigned long long v) __attribute__((always_inline)) { xw::atomic_add64_agent(q, v); };
                add(&pr[8], L()->acc[kAccBlocks]); add(&pr[13], L()->acc[kAccPasses]); add(&pr[14], L()->acc[kAccUndo]); add(&pr[26], L()->acc[kAccNeed]); add(&pr[27], L()->acc[kAccAhead]);
                add(&pr[9], L()->cnt.stale_ht); add(&pr[10], L()->cnt.stale_rk); add(&pr[11], L()->acc[kAccRedo]);
                add(&pr[96], L()->acc[kAccJobs]); add(&pr[97], L()->acc[kAccTaken]); add(&pr[98], L()->acc[kAccTakenNodes]); add(&pr[99], L()->acc[kAccHelpWait]);
                Counters &c = P->cnt;
                const Counters &lc = L()->cnt;
                add(&c.n_literal, lc.n_literal); add(&c.n_dict, lc.n_dict); add(&c.n_rep, lc.n_rep); add(&c.segments, lc.segments);
                add(&c.rans_syms, lc.rans_syms); add(&c.bit_ops, lc.bit_ops); add(&c.frames, lc.frames); add(&c.cmp_bytes, lc.cmp_bytes);
                add(&pr[20], L()->acc[kAccWait]); add(&pr[21], L()->acc[kAccTotal] + xw::tick()); add(&pr[22], L()->acc[kAccEmit]); add(&pr[23], L()->acc[kAccSetup]); add(&pr[24], L()->acc[kAccPass]);
                const uint32_t xe = xw::ld_agent(&V.hx->err);
                if (xe) {                                           // (this stage is the only writer of the sticky error word)
                    P->error = xe;
                    uint32_t *d = V.hx->dbg[2];                     // where this stage was when it left
                    xw::st_agent(d + 0, ci); xw::st_agent(d + 1, L()->dbgw[0]); xw::st_agent(d + 2, L()->dbgw[1]); xw::st_agent(d + 3, L()->dbgw[2]);
                    xw::st_agent(d + 4, L()->stg[4]); xw::st_agent(d + 5, t_out_seen); xw::st_agent(d + 6, err);
                }
                if ((err || xe) && G.abort_word) xw::st_agent(G.abort_word, 1u);
            }
        }
    }

    // ---- the helper parser's workgroup: jobs of the parser stage until that stage ends the launch ---------------------------------
    XW_FN void run_helper(uint32_t c0, uint32_t k)
    {
        is_helper = true; hrole = k + 1;
        HelpBox *hb = V.hb;                                         // (the job's parameters and prices: the first box)
        const uint32_t tid = xw::thread();
        for (uint32_t k = tid; k < sizeof(Counters) / 8; k += kParserThreads) ((unsigned long long *)&L()->cnt)[k] = 0;
        err = 0; tab_dirty = false; myjob = 0;
        t_out_seen = (uint32_t)((unsigned long long)c0 * g.chunk_size);
        if (tid < 5) L()->stg[tid] = (tid == 3 || tid == 2) ? 0u : t_out_seen;
        pend_t = t_out_seen;
        n_eq_fill = n_eq_rounds = 0; n_cmp = 0;
        if (tid < kAccN) L()->acc[tid] = 0;
        if (tid < 16) L()->sh[tid] = 0;
        xw::block_sync();
        for (;;) {
            // the next job
            if (xw::wave() == 0) {                                  // (the wave polls as one)
                uint32_t j, spins = 0;
                for (;;) {
                    j = xw::readfirst(xw::ld_agent(&V.hx->h_job));
                    if (j != myjob) break;
                    if ((++spins & 15u) == 0 && xw::readfirst(xw::ld_agent(&V.hx->err))) { j = kHelpExit; break; }
                    xw::pause_long();
                }
                xw::after_poll();
                if (tid == 0) L()->hj[2] = j;
            }
            xw::block_sync();
            const uint32_t job = L()->hj[2];
            if (job == kHelpExit) break;
            myjob = job; hstop = 0; err = 0;
            const uint32_t seg_a = xw::ld_agent(&hb->seg_a), chunk_left = xw::ld_agent(&hb->chunk_left);
            base = xw::ld_agent(&hb->base);
            rep0 = xw::ld_agent(&hb->rep[0]); rep1 = xw::ld_agent(&hb->rep[1]); rep2 = xw::ld_agent(&hb->rep[2]); rep3 = xw::ld_agent(&hb->rep[3]);
            {
                uint32_t *pw = (uint32_t *)L()->price, *lw = (uint32_t *)L()->len_price, *sw = (uint32_t *)L()->slot_price;
                for (uint32_t k = tid; k < kNumCtx * 8; k += kParserThreads) pw[k] = xw::ld_agent((const uint32_t *)hb->price + k);
                if (tid < (kMatchMax + 8) / 2) lw[tid] = xw::ld_agent((const uint32_t *)hb->len_price + tid);
                if (tid >= 256 && tid < 256 + 128) sw[tid - 256] = xw::ld_agent((const uint32_t *)hb->slot_price + (tid - 256));
            }
            if (xw::wave() == kPW - 1) pump_seek(seg_a + help_start(k));
            if (tid == 0) { L()->sh[4] = 0; acc(kAccJobs, 1); }
            xw::block_sync();
            uint32_t ncmds = 0;
            const uint32_t len = parse_segment(seg_a, chunk_left, ncmds);
            xw::block_sync();
            if (tid == 0) {
                // done: every node up to the segment's end is in the box; given up: the job was dropped, or this workgroup cannot end the segment
                const bool done = len != 0 && !err && !hstop;
                xw::st_agent(&V.hx->hw[k].state, (job << 2) | (done ? 1u : 2u));
                if (done) acc(kAccTaken, 1);
                L()->sh[4] = 0;
                L()->hj[3] = xw::ld_agent(&V.hx->err) ? 1u : 0u;        // (another stage failed: leave)
            }
            xw::block_sync();
            if (L()->hj[3]) break;
            err = 0;
        }
        xw::block_sync();
        if (tid == 0) {
            unsigned long long *pr = G.persist->prof;
            xw::atomic_add64_agent(&pr[100], L()->acc[kAccJobs]); xw::atomic_add64_agent(&pr[101], L()->acc[kAccTaken]);
            xw::atomic_add64_agent(&pr[102], L()->acc[kAccBlocks]); xw::atomic_add64_agent(&pr[103], L()->acc[kAccPasses]);
            xw::atomic_add64_agent(&pr[104], L()->acc[kAccWait]);
        }
    }
};

}  // namespace v2
}  // namespace nlzm
