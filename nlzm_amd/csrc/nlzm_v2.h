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
constexpr uint32_t kTfStride = 5 * 64;          // words per position in the ring of dense tables: delta[l] at [l - 1]

// finder -> table record:
//   w0        number of pairs in w2.. (bits 0..2) | BT4 ran: take the worker record's list (bit 3) | top pair present (bit 4)
//             | input byte << 8
//   w2..w13   up to six (distance, length) pairs: HT2, HT3 x2, RK256 carried / new, a former top entry that has stopped growing
//   w14, w15  the top entry (distance, length) while it is still growing with the position (:1503-1512)
constexpr uint32_t kFtBt = 8u, kFtTop = 16u;
// table -> parser record:
//   w0        edges (bits 0..5) | input byte << 8 | table length (mt.max_len) << 16
//   w1        (unused)
//   w2, w3    the mask of sampled lengths that bring a new distance
//   w4..      per sampled length: distance, length | length value << 9 | distance slot << 18 | extra bits << 24 | valid << 31
struct Hx {                                     // progress words, one 128-byte line each
    alignas(128) uint32_t f_pos;                // finder: records of positions < f_pos are written
    alignas(128) uint32_t t_pos;                // table: records of positions < t_pos are consumed
    alignas(128) uint32_t t_out;                // table: parser records of positions < t_out are written
    alignas(128) uint32_t p_pos;                // parser: records of positions < p_pos are consumed
    alignas(128) unsigned long long p_seg;      // parser: segment start << 32 | positions below this lie in that segment
    alignas(128) uint32_t err;                  // any stage: nonzero -> every stage leaves (the FIRST code stays: raise())
    uint32_t err_info[7];                       // of the stage that raised it: stage (11 finder, 12 table, 13 parser), wait site, position, what it saw
    alignas(128) uint32_t dbg[4][32];           // per stage (0 finder, 1 table, 2 parser): where it was when it left because of an error
};
enum : uint32_t { kStFinder = 11, kStTable = 12, kStParser = 13 };

struct GlobalsV2 {
    uint32_t *ft;                               // [kFtRing][kFtStride]
    uint32_t *tp;                               // [kTpRing][kTpStride]
    uint32_t *tf;                               // [kTpRing][kTfStride]
    Hx *hx;
    uint32_t *state;                            // StateV2 (survives between launches)
};

// stage state that survives between launches (HBM)
struct StateV2 {
    // finder
    uint32_t reach;                             // largest end (absolute) of any closed table entry so far
    uint32_t s_active, s_d, s_end, s_seen;      // growing top entry: distance, first mismatch (kNone: not found yet), verified up to
    uint32_t prev_nice, seg_s;
    // table: the front after the last position of the launch
    uint32_t front_n;
    uint32_t front[2 * kFrontMax];
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
    uint32_t s_active, s_d, s_end, s_seen;
    uint32_t rk_from, rk_to, rk_len, rk_end;
    uint32_t prev_nice, seg_s;
    uint32_t t_pos_seen;
    uint32_t err;
    uint32_t dbg_a = 0;         // start of the block being evaluated (error dump)
    unsigned long long n_pos, n_nice, n_unc, n_ht, n_rkp, n_rki, n_cmp, n_blocks, n_cut0, n_cut1, n_cut2, n_cut3, n_cut4, n_cut5;
    unsigned long long t_wait, t_wait_bt = 0, t_total;
    uint32_t n_late_unc = 0, n_late_other = 0;
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
                const unsigned long long tw = xw::tick();
                const unsigned long long t0 = xw::clock100();
                uint32_t spins = 0;
                for (;;) {
                    const unsigned long long ps = xw::readfirst64(xw::ld_agent64(&V.hx->p_seg));
                    if ((int32_t)((uint32_t)ps - (a0 + 1)) >= 0) {
                        xw::after_poll();
                        seg_s = (uint32_t)(ps >> 32);
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
        // ---- loads that the position alone addresses
        uint32_t v4 = 0;
        unsigned long long own0 = 0, own1 = 0;
        if (in_blk) { own0 = load64u(cur); own1 = load64u(cur + 8); v4 = (uint32_t)own0; }
        const uint32_t rkh = rk_call ? G.rkhash[a] : 0u;
        uint32_t rkv = rk_call ? G.rk_table[rkh >> g.rk_shift] : 0u;
        const bool unc = in_blk && G.workers && G.unc[bi] != 0;
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
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (!cd[k]) continue;
            const unsigned long long x0 = load64u(cur - cd[k]) ^ own0;
            uint32_t l;
            if (x0) l = (uint32_t)__builtin_ctzll(x0) >> 3;
            else {
                const unsigned long long x1 = load64u(cur - cd[k] + 8) ^ own1;
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
        uint32_t np = 0, ec = 0, od = kNone, cmpb = 0;      // pairs, largest closed end, smallest open distance, bytes compared
        auto add_pair = [&](uint32_t d, uint32_t l) __attribute__((always_inline)) {
            if (l >= cap) { od = umin(od, d); return; }     // as long as the lookahead allows: may grow at the next position
            st[2 + 2 * np] = d; st[3 + 2 * np] = l; np++;
            ec = umax(ec, a + l);
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
        if (rk_call && rk_act) {
            const uint32_t d = rk_to - rk_from, l = rk_len - (q - rk_to);
            if (l >= match_min(d)) add_pair(d, umin(l, kMatchMax));
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
        bool ev_ok = false;
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
                if (i == k) {
                    cmpb += l + (l < kav);
                    if (ev_ok) add_pair(kd, umin(l, kMatchMax));
                }
            }
        }

        const unsigned long long f5 = ptick();
        // ---- BT4: the worker lanes' result (longest record-setter in the record's words 9, 10).
        // A worker lane walks its bin in position order.  At an `unc` position it does not wait for this stage's decision: it
        // assumes it and goes on, but the RESULTS of the calls behind such a position are held back until the decision is
        // stored (a decision the other way takes those calls back).  So the result of a later position of the same bin cannot
        // arrive before this block is committed: such a position becomes lane 0 of the next block.
        uint32_t cut_bin = 64;
        if (G.workers) {
            const uint32_t bin = (hash4(v4) >> g.bt_shift) % G.nheads;
            for (unsigned long long um = xw::ballot(unc && avail >= 4); um; um &= um - 1) {
                const uint32_t x = (uint32_t)__builtin_ctzll(um);
                const uint32_t bx = xw::readlane(bin, x);
                const unsigned long long later = xw::ballot(bt_call && i > x && bin == bx);
                if (later) cut_bin = umin(cut_bin, (uint32_t)__builtin_ctzll(later));
            }
        }
        const bool bt_wait = bt_call && i < cut_bin;
        uint32_t bt_n = 0;
        if (xw::any(bt_wait)) {
            const unsigned long long tw = xw::tick();
            const unsigned long long t0 = xw::clock100();
            uint32_t spins = 0;
            if (bt_wait) xw::need_bt(G.hook_user, a);
            for (;;) {
                uint32_t w0 = bt_wait ? xw::ld_agent(G.bt_ready + bi * kBtRec) : kBtReady;
                if (spins == 0) {       // (accounting: whose results are not there when the block asks first)
                    const unsigned long long late = xw::ballot(!(w0 & kBtReady));
                    n_late_unc += (uint32_t)__builtin_popcountll(late & xw::ballot(unc));
                    n_late_other += (uint32_t)__builtin_popcountll(late & ~xw::ballot(unc));
                }
                if (!xw::any(!(w0 & kBtReady))) {
                    xw::after_poll();
                    if (bt_wait) {
                        bt_n = w0 & 0x1FFu;
                        if (bt_n) {
                            const uint32_t d = xw::ld_agent(G.bt_ready + bi * kBtRec + 1), l = xw::ld_agent(G.bt_ready + bi * kBtRec + 2);   // (quad 0: with the ready word)
                            if (l >= cap) od = umin(od, d); else ec = umax(ec, a + l);
                        }
                    }
                    break;
                }
                if ((++spins & 63u) == 0) {
                    if (xw::readfirst(xw::ld_agent(&V.hx->err))) { err = kErrInternal + 100; return 1; }
#ifndef NLZM_SIM
                    if (xw::clock100() - t0 > 3000000000ull) {
                        // (which positions are missing, and whether the first of them is one whose call this stage decides)
                        const unsigned long long late = xw::ballot(!(w0 & kBtReady));
                        const uint32_t fl = (uint32_t)__builtin_ctzll(late | (1ull << 63));
                        fail(kErrTimeout, a0 + fl, 6, (uint32_t)__builtin_popcountll(late) | (xw::readlane(unc ? 1u : 0u, fl) << 8) | (n << 16),
                             xw::readlane((hash4(v4) >> g.bt_shift) % G.nheads, fl));
                        return 1;
                    }
#else
                    (void)t0;
#endif
                }
                xw::pause();
            }
            t_wait += xw::tick() - tw; t_wait_bt += xw::tick() - tw;
        }

        const unsigned long long f6 = ptick();
        // ---- verification: what the table's reach really is in front of every lane (:1514 sees it after carry + extend)
        const uint32_t pm = xw::scan_max(in_blk ? ec : 0u);                     // inclusive prefix max of the closed ends
        uint32_t before = umax(xw::lane_below(pm, 0u), reach);
        if (i >= jc && s_active) before = umax(before, s_end);                  // the former top entry, closed
        const bool nice_real = in_blk && umax(before, s_sliding ? s_e : 0u) >= a + kNice;
        const unsigned long long bad = xw::ballot(in_blk && nice_real != nice_pred);
        const uint32_t cut_nice = bad ? (uint32_t)__builtin_ctzll(bad) : 64u;
        // a match as long as the lookahead allows becomes the growing top entry, unless one with a smaller
        // distance is growing already (:835-852 keeps the smaller distance at the table's end)
        const bool s_here = s_sliding;                                          // the old one still grows at this lane
        const unsigned long long om = xw::ballot(in_blk && od != kNone && (!s_here || od < s_d));
        const uint32_t jo = om ? (uint32_t)__builtin_ctzll(om) : 64u;
        uint32_t m = umin(n, umin(cut_nice, umin(jo + 1 > 64 ? 64u : jo + 1, umin(cut_ev, umin(cut_rk, umin(cut_slot, cut_bin))))));
        if (m == 0) { fail(kErrInternal, a0, 7); return 1; }
        n_blocks++;
        if (m < n) {        // (why the block was cut; diagnostics)
            // (adds, not a chain of branches: the compiler turns the chain into ONE indexed access and the stage's state goes to scratch)
            const uint32_t w = m == cut_nice ? 0u : (m == jo + 1 ? 1u : (m == cut_ev ? 2u : (m == cut_rk ? 3u : (m == cut_bin ? 5u : 4u))));
            n_cut0 += w == 0; n_cut1 += w == 1; n_cut2 += w == 2; n_cut3 += w == 3; n_cut4 += w == 4; n_cut5 += w == 5;
        }
        const bool fin = i < m;

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
            reach = umax(reach, pm_last);
            if (s_active && jc <= last) { reach = umax(reach, s_end); s_active = 0; }
            if (jo == last) { s_active = 1; s_d = xw::readlane(od, last); s_end = kNone; s_seen = a0 + last + xw::readlane(cap, last); }
            prev_nice = xw::readlane(nice_real ? 1u : 0u, last);
            const unsigned long long rkf = xw::ballot(fin && rk_call);
            if (rkf) rk_end = (a0 + (63u - (uint32_t)__builtin_clzll(rkf))) - base + 256;
            // a calling position beyond the carried RK match drops it for good (:1066-1068; rk_to is never rebased, so a
            // later position of the next epoch could look covered again)
            if (rk_len && xw::any(fin && rk_call && !rk_act)) rk_len = 0;
            if (ev_ok && cut_ev == m) {                                         // the RK candidate of lane m-1 was taken (:1102-1104)
                rk_from = (a0 + last - base) - ev_d; rk_to = a0 + last - base; rk_len = ev_l;
            }
        }
        // counters
        n_pos += m;
        n_nice += (unsigned long long)__builtin_popcountll(xw::ballot(fin && nice_real));
        n_unc += (unsigned long long)__builtin_popcountll(xw::ballot(fin && unc && avail >= 4));
        n_ht += (unsigned long long)__builtin_popcountll(xw::ballot(fin && ht_call));
        n_rkp += (unsigned long long)__builtin_popcountll(xw::ballot(fin && rk_probe));
        n_rki += (unsigned long long)__builtin_popcountll(xw::ballot(fin && rk_call && (q & 255u) == 0));
        {
            uint32_t c = fin ? cmpb : 0u;
            for (uint32_t d = 32; d; d >>= 1) c += xw::shfl(c, i ^ d);
            n_cmp += xw::readfirst(c);
        }
        xw::drain();
        if (i == 0) xw::st_agent(&V.hx->f_pos, a0 + m);
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
        reach = xw::readfirst(S->reach); s_active = xw::readfirst(S->s_active); s_d = xw::readfirst(S->s_d);
        s_end = xw::readfirst(S->s_end); s_seen = xw::readfirst(S->s_seen);
        prev_nice = xw::readfirst(S->prev_nice); seg_s = xw::readfirst(S->seg_s);
        rk_from = xw::readfirst(P->rk_from); rk_to = xw::readfirst(P->rk_to); rk_len = xw::readfirst(P->rk_len); rk_end = xw::readfirst(P->rk_end);
        err = xw::readfirst(P->error);
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
            const uint32_t a1 = (uint32_t)chunk_abs + p_end, la_end = (uint32_t)chunk_abs + chunk_read;
            uint32_t a = (uint32_t)chunk_abs;
            while (a < a1 && !err) {
                const uint32_t n = umin(64u, a1 - a);
                dbg_a = a;
                // ring space: the table stage must have consumed position a + n - kFtRing
                if ((int32_t)(a + n - t_pos_seen - kFtRing) > 0) {
                    const unsigned long long tw = xw::tick();
                    if (!wait_word_ge(&V.hx->t_pos, a + n - kFtRing, V.hx, 1)) { err = kErrInternal + 100; break; }
                    t_pos_seen = xw::readfirst(xw::ld_agent(&V.hx->t_pos));
                    t_wait += xw::tick() - tw;
                }
                a += block(a, n, a1, la_end);
            }
        }
        xw::wave_sync();
        for (uint32_t k = i; k < 4096; k += 64) G.ht2[k] = L->ht2[k];
        for (uint32_t k = i; k < ht3_rows; k += 64) G.ht3[k] = L->ht3[k];
        if (i == 0) {
            P->reb_base = base;
            P->rk_from = rk_from; P->rk_to = rk_to; P->rk_len = rk_len; P->rk_end = rk_end;
            S->reach = reach; S->s_active = s_active; S->s_d = s_d; S->s_end = s_end; S->s_seen = s_seen;
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
// one, :823-833).  With e = q + length (the match's absolute end):
//
//     delta[l] at p  =  min { G[e'] : e' >= p + l },     G[e] = smallest distance of the matches that END exactly at e
//
// -- a match is a POINT in G (one LDS atomic min, whatever its length), moving to the next position costs nothing (G is
// indexed by the absolute end), and the table of a position is a suffix minimum over the window e in (p, p + 264].
// (The one entry whose end moves with p, the top entry while it keeps extending :1503-1512, arrives from the finder
// stage as a fresh pair per position.)  One workgroup, nothing serial in it:
//
//   loader   (wave 0)     the finder's records and the worker lanes' BT4 records of the positions to come, staged in LDS
//                         (32 positions a step, three steps in flight)
//   emitters (waves 1..7) every one keeps a G of its OWN and applies the pairs of EVERY position to it -- lanes = the position's
//                         pairs (six of the finder's, its top entry, twelve record-setters of the BT4 descent; more of them,
//                         rare, from HBM): one atomic min instruction a position -- and emits every seventh position: suffix
//                         minimum of G's window by DPP scans (lanes = lengths, 64 a chunk), the dense table to the ring the parser
//                         re-lists from near a forced cut (:1545) and the stage test reads, then the sampled lengths of
//                         :1558-1562 with the lanes as SAMPLES: distance, length value, distance slot, extra bits, the mask of
//                         samples that bring a new distance; 16-byte stores, waited for once every few positions.
constexpr uint32_t kTW = 8;                     // waves of the stage
constexpr uint32_t kTEmit = kTW - 1;            // emitters
constexpr uint32_t kTRecRing = 256;             // positions whose records are staged
constexpr uint32_t kGRing = 512;
constexpr uint32_t kTLoad = 32;                 // positions the loader requests a step
constexpr uint32_t kTDrainEvery = 4;            // positions an emitter writes between two waits for its stores
constexpr uint32_t kTInline = 12;               // record-setters of a BT4 descent staged in LDS (four in the record, eight from the pair list)
constexpr uint32_t kTRecWords = kFtStride + kBtRec + 16;    // a position's staged words: the finder's record, the BT4 record, record-setters 4 .. 11

struct TLds {
    uint32_t G[kTEmit][kGRing];                 // emitter k's G: end e at [k][e & 511]
    uint32_t rec[kTRecRing * kTRecWords];       // position p at [(p & 255) * 48]
    uint32_t staged;                            // loader: records of positions below this are in LDS
    uint32_t maxend[kTEmit];                    // emitter k: the largest end of any pair it has applied (mt.max_len = that - p)
    uint32_t e_done[kTEmit];                    // emitter k: its records of the positions below this are in memory
    uint32_t e_taken[kTEmit];                   // emitter k: the staged records of the positions below this are applied
    uint32_t stop;                              // nonzero: leave
};

NLZM_HD uint32_t tf_index(uint32_t l) { return l - 1; }     // dense table of a position in the ring: delta[l]

struct Table {
    Geom g;
    Globals G;
    GlobalsV2 V;
    uint32_t err;
    unsigned long long n_pos = 0, n_slow = 0, t_wait = 0;
    unsigned long long tt0 = 0, tt1 = 0, tt2 = 0, tt3 = 0;     // profile build, emitters: applying pairs, emitting, waiting for records, for the stores
#ifdef NLZM_PROFILE
    XW_FN unsigned long long ptick() const { return xw::tick(); }
#else
    XW_FN unsigned long long ptick() const { return 0; }
#endif
    XW_FN TLds *L() const { return xw::lds<TLds>(); }

    // wait for an LDS word of the stage to pass v (word - v > 0 as a signed difference); false: the stage is leaving
    XW_FN bool wait_lds_gt(const uint32_t *p, uint32_t v)
    {
        uint32_t spins = 0;
        while ((int32_t)(xw::readfirst(xw::lds_ld(p)) - v) <= 0) {
            if (xw::readfirst(xw::lds_ld(&L()->stop))) return false;
            if ((++spins & 255u) == 0 && xw::readfirst(xw::ld_agent(&V.hx->err))) { if (xw::lane() == 0) xw::lds_st(&L()->stop, 1u); return false; }
            xw::pause();
        }
        xw::after_poll();
        return true;
    }
    XW_FN void leave(uint32_t site, uint32_t pos)
    {
        if (xw::lane() == 0) { xw::lds_st(&L()->stop, 1u); if (!xw::ld_agent(&V.hx->err)) raise(V.hx, kErrInternal + 200, kStTable, site, pos); }
    }

    // ---- loader -----------------------------------------------------------------------------------------------------
    // Three steps in flight, 32 positions each (lane = position x 16-byte quad, two positions a lane): the records requested;
    // a step later written to LDS, and for the positions whose BT4 descent had more than four record-setters the next eight of
    // them requested (bt_pairs: in memory before the record's ready word, nlzm_core.h); a step later those written, and the
    // positions said to be staged.
    XW_FN void run_loader(uint32_t a_first, uint32_t a_last)
    {
        TLds *Lp = L();
        const uint32_t i = xw::lane(), pj = i >> 2, qd = i & 3u;
        uint32_t req = a_first;
        uint32_t f_seen = a_first, taken = a_first;
        uint32_t r_n = 0, r_a = 0, x_n = 0, x_a = 0;                // the step whose records / extra pairs are on their way
        uint32_t fq[2][4] = {}, bq[2][4] = {}, xq[2][4] = {};
        const unsigned long long t0 = xw::clock100();
        uint32_t idle = 0;
        uint32_t staged = a_first;
        while (staged < a_last) {
            // ---- extra pairs of the step before the last: into LDS; its positions are staged
            if (x_n) {
#pragma unroll
                for (uint32_t h = 0; h < 2; h++) {
                    const uint32_t a = x_a + pj + 16 * h;
                    if (pj + 16 * h < x_n) {
                        uint32_t *xd = Lp->rec + (a & (kTRecRing - 1)) * kTRecWords + kFtStride + kBtRec + 4 * qd;
                        xd[0] = xq[h][0]; xd[1] = xq[h][1]; xd[2] = xq[h][2]; xd[3] = xq[h][3];
                    }
                }
                staged = x_a + x_n;
                xw::wave_sync();
                if (i == 0) xw::lds_st(&Lp->staged, staged);
                x_n = 0;
            }
            // ---- records of the last step: into LDS (the BT4 record's quads carry a ready bit / a tag: looked at again if one is
            // not there yet); the extra pairs requested
            if (r_n) {
#pragma unroll
                for (uint32_t h = 0; h < 2; h++) {
                    const uint32_t a = r_a + pj + 16 * h;
                    const bool on = pj + 16 * h < r_n;
                    // (quad 0 of the finder's record says whether a BT4 record belongs to the position; quad 0 of that one how many pairs)
                    const uint32_t w0 = xw::shfl(fq[h][0], i & ~3u);
                    if (on && (w0 & kFtBt)) {
                        const uint32_t *br = G.bt_ready + (unsigned long long)(a - G.batch_a0) * kBtRec + 4 * qd;
                        uint32_t spins = 0;
                        while (!((qd ? bq[h][3] : bq[h][0]) & (qd ? kBtTag : kBtReady))) {
                            bq[h][0] = xw::ld_agent(br); bq[h][1] = xw::ld_agent(br + 1); bq[h][2] = xw::ld_agent(br + 2); bq[h][3] = xw::ld_agent(br + 3);
                            if ((++spins & 255u) == 0 && xw::ld_agent(&V.hx->err)) break;
                        }
                    }
                    const uint32_t bw0 = xw::shfl(bq[h][0], i & ~3u);
                    if (on) {
                        uint32_t *fd = Lp->rec + (a & (kTRecRing - 1)) * kTRecWords + 4 * qd;
                        fd[0] = fq[h][0]; fd[1] = fq[h][1]; fd[2] = fq[h][2]; fd[3] = fq[h][3];
                        uint32_t *bd = fd + kFtStride;
                        bd[0] = bq[h][0]; bd[1] = bq[h][1]; bd[2] = bq[h][2]; bd[3] = bq[h][3];
                        if ((w0 & kFtBt) && (bw0 & 0x1FFu) > 4) {
                            // pairs 4 .. 11 of the descent: words 8 .. 23 of the position's pair list
                            const uint32_t *xp = G.bt_pairs + (unsigned long long)(a - G.batch_a0) * (2 * kBtMaxPairs) + 8 + 4 * qd;
                            xq[h][0] = xw::ld_agent(xp); xq[h][1] = xw::ld_agent(xp + 1); xq[h][2] = xw::ld_agent(xp + 2); xq[h][3] = xw::ld_agent(xp + 3);
                        }
                    }
                }
                x_a = r_a; x_n = r_n; r_n = 0;
                xw::wave_sync();
                if (i == 0) xw::st_agent(&V.hx->t_pos, x_a + x_n);   // (the ring's records are copied: the finder may overwrite them)
            }
            // ---- the next positions: what the finder has written, and what the record ring has room for
            if ((int32_t)(f_seen - req) <= 0) f_seen = xw::readfirst(xw::ld_agent(&V.hx->f_pos));
            xw::after_poll();
            taken = kNone;
            for (uint32_t k = 0; k < kTEmit; k++) taken = umin(taken, xw::readfirst(xw::lds_ld(&Lp->e_taken[k])));
            const uint32_t lim = umin(f_seen, umin(taken + kTRecRing, a_last));
            const uint32_t n = (int32_t)(lim - req) > 0 ? umin(kTLoad, lim - req) : 0u;
            if (n) {
#pragma unroll
                for (uint32_t h = 0; h < 2; h++) {
                    const uint32_t a = req + pj + 16 * h;
                    if (pj + 16 * h < n) {
                        const uint32_t *fr = V.ft + (unsigned long long)(a & (kFtRing - 1)) * kFtStride + 4 * qd;
                        fq[h][0] = xw::ld_agent(fr); fq[h][1] = xw::ld_agent(fr + 1); fq[h][2] = xw::ld_agent(fr + 2); fq[h][3] = xw::ld_agent(fr + 3);
                        const uint32_t *br = G.bt_ready + (unsigned long long)(a - G.batch_a0) * kBtRec + 4 * qd;
                        bq[h][0] = xw::ld_agent(br); bq[h][1] = xw::ld_agent(br + 1); bq[h][2] = xw::ld_agent(br + 2); bq[h][3] = xw::ld_agent(br + 3);
                    }
                }
                r_a = req; r_n = n; req += n;
                idle = 0;
            } else if (!x_n) {
                if (xw::readfirst(xw::lds_ld(&Lp->stop))) return;
                if ((++idle & 63u) == 0) {
                    if (xw::readfirst(xw::ld_agent(&V.hx->err))) { if (i == 0) xw::lds_st(&Lp->stop, 1u); return; }
#ifndef NLZM_SIM
                    if (xw::clock100() - t0 > 30000000000ull) { leave(2, req); return; }    // (a launch takes less than 300 s)
#else
                    (void)t0;
#endif
                }
                xw::pause();
            }
        }
    }

    // ---- emitters -----------------------------------------------------------------------------------------------------
    XW_FN void capture(uint32_t a, uint32_t mt_max, const uint32_t *dense);
#ifdef NLZM_SIM
    static void sim_on_table(void *user, uint32_t a, uint32_t mt_max, const uint32_t *dense);
#endif
    // the chunk of position p: its last position + 1, the end of its lookahead
    struct Chunk { uint32_t lo, a1, la_end; };
    XW_FN Chunk chunk_of(uint32_t ci) const
    {
        const unsigned long long chunk_abs = (unsigned long long)ci * g.chunk_size;
        const unsigned long long remain = g.n - chunk_abs;
        const uint32_t chunk_read = (uint32_t)(remain < g.feed ? remain : g.feed);
        return Chunk{ (uint32_t)chunk_abs, (uint32_t)chunk_abs + umin(g.chunk_size, chunk_read), (uint32_t)chunk_abs + chunk_read };
    }
    // the pairs of position q into this wave's G: lanes 1..6 the finder's pairs, lane 7 its top entry, lanes 8..19 the first
    // twelve record-setters of the BT4 descent (those as long as the lookahead allows are the finder's top entry already); the
    // largest end among them into *mx (an LDS word of this wave).  Every emitter does this for every position: kept short --
    // no scalar detour (every lane reads the record's head words itself), no reduction (LDS atomic max).
    XW_FN void apply_pairs(uint32_t *Gw, uint32_t *mx, uint32_t q, uint32_t la_end)
    {
        TLds *Lp = L();
        const uint32_t i = xw::lane();
        const uint32_t cap_len = umin(la_end - q, kMatchMax);
        const uint32_t *r = Lp->rec + (q & (kTRecRing - 1)) * kTRecWords;
        const uint32_t w0 = r[0], bw0 = r[kFtStride];
        const uint32_t np = w0 & 7u, cnt = (w0 & kFtBt) ? (bw0 & 0x1FFu) : 0u;
        // (where a lane's pair sits in the staged words: the finder's at 2i, 2i + 1; the BT4 record's first four as bt_rec_d / _l say;
        //  the next eight behind the record)
        const uint32_t k = i - 8;
        const uint32_t od = i < 8 ? 2 * i : (i < 12 ? kFtStride + bt_rec_d(k & 3u) : kFtStride + kBtRec + 2 * ((i - 12) & 7u));
        const uint32_t ol = i < 8 ? 2 * i + 1 : (i < 12 ? kFtStride + bt_rec_l(k & 3u) : od + 1);
        const uint32_t d = r[od], l = r[ol];
        const bool ok = i >= 1 && i < 8 + kTInline && (i < 7 ? i - 1 < np : (i == 7 ? (w0 & kFtTop) != 0 : (k < cnt && l < cap_len)));
        // (the slot of the one end that becomes possible at this position: used 512 ends ago)
        Gw[(q + kMatchMax) & (kGRing - 1)] = kNone;
        xw::wave_sync();
        if (ok) { xw::lds_min(&Gw[(q + l) & (kGRing - 1)], d); xw::lds_max(mx, q + l); }
        if (NLZM_RARE(xw::any(cnt > kTInline))) {
            const uint32_t *pairs = G.bt_pairs + (unsigned long long)(q - G.batch_a0) * (2 * kBtMaxPairs);
            for (uint32_t j = kTInline + i; j < cnt; j += 64) {
                const uint32_t dk = xw::ld_agent(pairs + 2 * j), lk = xw::ld_agent(pairs + 2 * j + 1);
                if (lk < cap_len) { xw::lds_min(&Gw[(q + lk) & (kGRing - 1)], dk); xw::lds_max(mx, q + lk); }
            }
            n_slow++;
        }
    }
    XW_FN void emit_position(const uint32_t *Gw, uint32_t p, uint32_t mt_max, uint32_t lit, uint32_t a1)
    {
        const uint32_t k = xw::lane();
        uint32_t *dense = V.tf + (unsigned long long)(p & (kTpRing - 1)) * kTfStride;
        // ---- delta[l] = min of G from end p + l on: chunk by chunk from the top, lane k of a chunk holds length 64 c + 64 - k
        uint32_t s0 = kNone, s1 = kNone, s2 = kNone, s3 = kNone, s4 = kNone, carry = kNone;
        auto chunk = [&](uint32_t c, uint32_t &sv) __attribute__((always_inline)) {
            if (64 * c >= mt_max) return;
            const uint32_t at = 64 * c + 63 - k;                    // length - 1 of this lane
            uint32_t v = at < mt_max ? Gw[(p + 1 + at) & (kGRing - 1)] : kNone;
            v = umin(xw::scan_min_u32(v), carry);
            carry = xw::readlane(v, 63);
            sv = v;
            // the dense table for the ring: four lengths a lane, 16-byte stores (lane 4j has lengths - 1 = 64c + 60 - 4j ..+3)
            const uint32_t v1 = xw::quad_bcast<1>(v), v2 = xw::quad_bcast<2>(v), v3 = xw::quad_bcast<3>(v);
            if ((k & 3u) == 0 && at - 3 < mt_max) xw::st_agent128(dense + at - 3, v3, v2, v1, v);
        };
        chunk(4, s4); chunk(3, s3); chunk(2, s2); chunk(1, s1); chunk(0, s0);
        // ---- the sampled lengths (:1558-1560), lanes = samples
        uint32_t max_len = umin(mt_max, a1 - p);                    // :1545-1548 (the 4096 cap is the parser's: it knows the segment)
        if (max_len < kMatchMin) max_len = 0;
        uint32_t step = (max_len - kMatchMin) >> 4;
        step += step == 0;
        const bool on = max_len && k < kMaxEdges && k * step + kMatchMin <= max_len;
        const uint32_t ne = (uint32_t)__builtin_popcountll(xw::ballot(on));
        const uint32_t tl = on ? max_len - k * step : 1u;
        const uint32_t at = tl - 1, src = 63u - (at & 63u), cq = at >> 6;
        uint32_t d = xw::shfl(s0, src);
        if (max_len > 64) {
            const uint32_t d1 = xw::shfl(s1, src); d = cq == 1 ? d1 : d;
            if (max_len > 128) {
                const uint32_t d2 = xw::shfl(s2, src), d3 = xw::shfl(s3, src), d4 = xw::shfl(s4, src);
                d = cq == 2 ? d2 : (cq == 3 ? d3 : (cq == 4 ? d4 : d));
            }
        }
        if (!on) d = 1;
        const uint32_t mm = match_min(d);
        uint32_t nx, ex;
        const uint32_t slot = dist_slot(d - 1, nx, ex);
        const bool valid = on && tl >= mm;
        const uint32_t lv = valid ? tl - mm : 0u;
        const uint32_t aw = on ? (tl | (lv << 9) | (slot << 18) | (nx << 24) | ((valid ? 1u : 0u) << 31)) : 0u;
        const uint32_t dw = on ? d : 0u;
        // the valid samples that bring a distance the valid one before did not have
        const unsigned long long vm = xw::ballot(valid), below = vm & ((1ull << k) - 1ull);
        const uint32_t dprev = xw::shfl(d, below ? 63u - (uint32_t)__builtin_clzll(below) : k);
        const uint32_t uniq = (uint32_t)xw::ballot(valid && (!below || d != dprev));
        // ---- the record: header + mask, and every two samples, as 16-byte stores
        uint32_t *rec = V.tp + (unsigned long long)(p & (kTpRing - 1)) * kTpStride;
        const uint32_t dn = xw::shfl(dw, (k + 1) & 63u), an = xw::shfl(aw, (k + 1) & 63u);
        if ((k & 1u) == 0 && k < ne) xw::st_agent128(rec + kTpEdges + 2 * k, dw, aw, k + 1 < ne ? dn : 0u, k + 1 < ne ? an : 0u);
        if (k == 0) xw::st_agent128(rec, ne | (lit << 8) | (mt_max << 16), 0u, uniq, 0u);
        if (G.cap_words) { xw::drain(); capture(p, mt_max, dense); }
#ifdef NLZM_SIM
        xw::wave_sync();
        if (k == 0) sim_on_table(G.hook_user, p, mt_max, dense);
#endif
    }
    XW_FN void run_emitter(uint32_t ek, uint32_t c0, uint32_t c1, uint32_t a_first, uint32_t a_last)
    {
        TLds *Lp = L();
        StateV2 *S = (StateV2 *)V.state;
        const uint32_t i = xw::lane();
        uint32_t *Gw = Lp->G[ek];
        // the window the launch before left: ends a_first + 1 .. a_first + 264, and the largest end so far
        for (uint32_t k = i; k < kGRing; k += 64) Gw[k] = kNone;
        xw::wave_sync();
        const uint32_t had = xw::readfirst(S->front_n);
        for (uint32_t k = i; k < kFrontMax && had; k += 64) Gw[(a_first + 1 + k) & (kGRing - 1)] = S->front[k];
        if (i == 0) Lp->maxend[ek] = had ? S->front[kFrontMax] : 0u;
        xw::wave_sync();
        uint32_t p_seen = a_first, staged_seen = a_first;
        uint32_t own = a_first + ek;                                // this wave's next position
        uint32_t since = 0;                                         // positions written since this wave's stores were last waited for
        uint32_t ci = c0;
        Chunk ch = chunk_of(ci);
        for (uint32_t q = a_first; q < a_last; q++) {
            if (q >= ch.a1) { ci++; ch = chunk_of(ci); }
            const unsigned long long q0 = ptick();
            if ((int32_t)(staged_seen - q) <= 0) {
                // (while there is nothing to do: what was written goes out)
                if (since && (int32_t)(xw::readfirst(xw::lds_ld(&Lp->staged)) - q) <= 0) { publish(ek, own_next(ek, q, a_first), a_last); since = 0; }
                if (!wait_lds_gt(&Lp->staged, q)) return;
                staged_seen = xw::readfirst(xw::lds_ld(&Lp->staged));
            }
            const unsigned long long q1 = ptick();
            apply_pairs(Gw, &Lp->maxend[ek], q, ch.la_end);
            const bool mine = q == own;
            // (the position's byte, while its record is still this wave's to read: the word below frees the slot for the loader)
            const uint32_t lit = mine ? (xw::readfirst(Lp->rec[(q & (kTRecRing - 1)) * kTRecWords]) >> 8) & 0xFFu : 0u;
            if ((q & 15u) == 15u || mine) { xw::wave_sync(); if (i == 0) xw::lds_st(&Lp->e_taken[ek], q + 1); }
            const unsigned long long q2 = ptick();
            tt2 += q1 - q0; tt0 += q2 - q1;
            if (!mine) continue;
            // ---- this wave's position: room in the parser's ring, then the record
            if ((int32_t)(q + 1 - p_seen - kTpRing) > 0) {
                if (since) { publish(ek, q, a_last); since = 0; }      // (q is this wave's own position)
                if (!wait_word_ge(&V.hx->p_pos, q + 1 - kTpRing, V.hx, 3)) { if (i == 0) xw::lds_st(&Lp->stop, 1u); return; }
                p_seen = xw::readfirst(xw::ld_agent(&V.hx->p_pos));
            }
            xw::wave_sync();
            own += kTEmit;
            const uint32_t maxend = xw::readfirst(xw::lds_ld(&Lp->maxend[ek]));
            emit_position(Gw, q, (int32_t)(maxend - q) > 0 ? maxend - q : 0u, lit, ch.a1);
            n_pos++;
            // the stores of a position take microseconds to land: waited for once every few positions, then all of them are said to be out
            if (++since == kTDrainEvery) { publish(ek, q + kTEmit, a_last); since = 0; }
            tt1 += ptick() - q2;
        }
        // (nothing of this wave is left: its word no longer holds the others back)
        publish(ek, kNone, a_last);
        if (i == 0) xw::lds_st(&Lp->e_taken[ek], kNone);
        if (ek == 0) {
            // the window for the launch to come (every emitter has applied every position: this one says so)
            xw::wave_sync();
            for (uint32_t k = i; k < kFrontMax; k += 64) S->front[k] = Gw[(a_last + 1 + k) & (kGRing - 1)];
            if (i == 0) { S->front[kFrontMax] = Lp->maxend[ek]; S->front_n = 1; }
        }
    }
    // the first position from q on that is emitter ek's
    XW_FN static uint32_t own_next(uint32_t ek, uint32_t q, uint32_t a_first) { const uint32_t r = (q - a_first) % kTEmit; return q + (ek >= r ? ek - r : ek + kTEmit - r); }
    // this wave's records of its positions below `next` (one of its own, or beyond the end) are in memory: with the other emitters'
    // words, how far the parser may read
    XW_FN void publish(uint32_t ek, uint32_t next, uint32_t a_last)
    {
        TLds *Lp = L();
        const unsigned long long q0 = ptick();
        xw::drain();
        xw::wave_sync();
        if (xw::lane() == 0) xw::lds_st(&Lp->e_done[ek], next);
        xw::wave_sync();
        uint32_t m = kNone;
        for (uint32_t k = 0; k < kTEmit; k++) m = umin(m, xw::readfirst(xw::lds_ld(&Lp->e_done[k])));
        if (m > a_last) m = a_last;
        if (xw::lane() == 0) xw::st_agent(&V.hx->t_out, m);
        tt3 += ptick() - q0;
    }

    XW_FN void run(uint32_t c0, uint32_t c1)
    {
        TLds *Lp = L();
        const uint32_t i = xw::lane(), w = xw::wave();
        const uint32_t a_first = (uint32_t)((unsigned long long)c0 * g.chunk_size);
        unsigned long long a_last64 = (unsigned long long)c1 * g.chunk_size;
        if (a_last64 > g.n) a_last64 = g.n;
        const uint32_t a_last = (uint32_t)a_last64;
        if (w == 0 && i == 0) {
            Lp->staged = a_first; Lp->stop = 0;
            for (uint32_t k = 0; k < kTEmit; k++) { Lp->e_done[k] = a_first + k; Lp->e_taken[k] = a_first; }
        }
        xw::block_sync();
        err = 0;
        const unsigned long long t_start = xw::tick();
        if (w == 0) run_loader(a_first, a_last);
        else run_emitter(w - 1, c0, c1, a_first, a_last);
        xw::block_sync();
        if (w == 0 && i == 0 && xw::ld_agent(&V.hx->err)) {         // where this stage was when it left
            uint32_t *d = V.hx->dbg[1];
            xw::st_agent(d + 0, xw::lds_ld(&Lp->staged)); xw::st_agent(d + 1, xw::lds_ld(&Lp->e_taken[0])); xw::st_agent(d + 2, xw::lds_ld(&Lp->e_done[0]));
            xw::st_agent(d + 3, xw::lds_ld(&Lp->e_done[1])); xw::st_agent(d + 4, xw::lds_ld(&Lp->stop)); xw::st_agent(d + 5, 0u);
        }
        if (i == 0 && w >= 1) {     // accounting
            unsigned long long *pr = G.persist->prof;
            xw::atomic_add64_agent(&pr[6], n_pos); xw::atomic_add64_agent(&pr[7], n_slow);
            xw::atomic_add64_agent(&pr[44], tt0); xw::atomic_add64_agent(&pr[45], tt1); xw::atomic_add64_agent(&pr[46], tt2); xw::atomic_add64_agent(&pr[47], tt3);
            if (w == 1) { xw::atomic_add64_agent(&pr[18], tt2 + tt3); xw::atomic_add64_agent(&pr[19], xw::tick() - t_start); }
        }
    }
};

// stage test tap: {position, max_len, delta[2..max_len]} (what the reference copies into mt_carry, :1543).  The emitters finish
// positions out of order; the tap's words are claimed with an atomic and the host sorts by position.
XW_FN void Table::capture(uint32_t a, uint32_t mt_max, const uint32_t *dense)
{
    if (a < G.cap_lo || a >= G.cap_hi) return;
    const unsigned long long need = 2 + (mt_max >= 2 ? mt_max - 1 : 0);
    unsigned long long used = 0;
    if (xw::lane() == 0) used = xw::atomic_fetch_add64_agent(G.cap_used, need);
    used = xw::readfirst64(used);
    if (used + need > G.cap_cap) { err = kErrCapture; if (xw::lane() == 0) raise(V.hx, kErrCapture, kStTable, 10, a); return; }
    if (xw::lane() == 0) { G.cap_words[used] = a; G.cap_words[used + 1] = mt_max; }
    for (uint32_t l = 2 + xw::lane(); l <= mt_max; l += 64) G.cap_words[used + l] = xw::ld_agent(dense + tf_index(l));
    xw::drain();
}

// =================================================================================================
// parser stage
// =================================================================================================
// parse_table (:1464-1651) visits the nodes of a segment in order; node p takes the literal edge of p-1, then relaxes
// its own sampled-length edges (dict, then rep where the distance is in the node's rep set) and its explicit rep
// probes, all with strict '>'.  The stage does the same, IN ORDER, with the lanes of ONE wave as the TARGETS:
//
//   chain (wave 0)      lane L holds the state of the one node t with t % 64 == L among the 64 nodes from the next one to
//                       be processed on: cost, winning edge (source, sampled edge, dict / rep), its distance, and the rep
//                       set the node would have (RepModel::Add applied to the source's set, :1160-1171).  Processing node
//                       n: its cost and rep set are read off lane n % 64 (v_readlane), the lane is retired (final state
//                       to LDS) and becomes node n + 64; every lane reads ITS entry of node n's row -- the edge of length
//                       (L - n) mod 64, priced ahead of time -- and takes it if it is strictly cheaper than what it holds:
//                       one compare per lane and node, every edge is relaxed exactly once, no atomics, no passes.  The
//                       candidates of a target arrive in the reference's order (sources ascending; dict before rep), so
//                       "strictly cheaper" is the reference's rule.  Edges longer than 64 (rare: max_len > 64) go through
//                       a ring of keys in LDS (far[]) that a lane takes over when its node enters the window.
//   rows (waves 3,5,6,7)  the table stage's record of a node -> a row of 64 entries {distance, dict price | rep price | edge},
//                       entry l-1 = the sampled edge of length l (prices are constant inside a segment, :1491, :1567,
//                       :1585), the literal edge at entry 0; the node's distinct distances for the probes' "already met"
//                       test; within 264 of the forced cut the lengths are re-listed from the position's front (:1545).
//                       Prepared up to 128 nodes ahead of the chain.
//   probes (waves 1..2) the explicit rep probes (:1598-1628) need the node's rep set, which only the chain knows, and
//                       bytes from HBM.  The chain does NOT wait for them: it goes on as if no probe found a match (18 % of
//                       the nodes have one, 0.07 % have one that changes anything), and these waves follow it: for every
//                       retired node the four probes are made (lanes = node x slot, 16 nodes per wave step) and each match
//                       becomes an edge (key = cost << 32 | source << 8 | rank, the reference's order of candidates).
//   decide (per step)   a step = up to 64 nodes.  When the chain has stopped and the probes of the step's nodes are in: a
//                       probe edge whose target is still ahead is merged into the chain's state (far[]); one whose target
//                       was retired is compared with the target's final key -- smaller means the chain would have gone
//                       differently: the step is done again from its saved start state with the step's probe edges handed
//                       to the chain (applied at their node, after its sampled edges, as the reference does).  Rare, exact.
//   loader (wave 4)     the table stage's records staged in LDS ahead of everybody (as before).
// Then backtrack and EMISSION on all eight waves (lanes = commands), unchanged.
constexpr unsigned long long kKeyNone = ~0ull;
constexpr uint32_t kRankLit = 255, kRankProbe = 64;
constexpr uint32_t kSrcNone = 0x1FFF;
constexpr uint32_t kPW = 8;                     // waves of the stage
constexpr uint32_t kParserThreads = 64 * kPW;
constexpr uint32_t kStagePos = 128;             // table records kept ahead in LDS: positions ...
constexpr uint32_t kStageEdges = 16;            // ... the first sampled edges of each (the rest, rare, is read from the ring)
constexpr uint32_t kStageQ = 2 + kStageEdges;   // 8-byte words per staged record: header, edges, the mask of samples with a new distance
constexpr uint32_t kPumpLoads = 8;              // loads per lane and step of the loader wave ...
constexpr uint32_t kPumpRecs = 64 / kStageQ;    // ... each for the words of three records: 24 records a step, two steps in flight
constexpr uint32_t kInf = 0x3FFFFFFFu;
constexpr uint32_t kRowRing = 128;              // nodes whose rows are kept: the step being parsed and what is prepared ahead
constexpr uint32_t kFarRing = 512;              // keys of the nodes beyond the chain's window; final states of the last nodes
constexpr uint32_t kStepMax = 64;               // nodes a step covers at most
constexpr uint32_t kStepWant = 32;              // ... and the rows the chain waits for (a bounded time) before it starts one
constexpr uint32_t kLitMark = 0xFFFFFF00u;      // "distance" of the literal edge: this | the byte (no distance is that large)
constexpr uint32_t kVerifyWaves = 2, kPrepWaves = 4;    // waves 1..2; 3, 5, 6, 7
constexpr uint32_t kLoaderWave = 4;             // (a workgroup's waves go to the SIMDs in turn: wave 4 shares wave 0's, and the loader sleeps most of the time)
constexpr uint32_t kVBatch = 16;                // nodes per step of a probe wave (four lanes each)
constexpr uint32_t kVMax = 4 * kStepMax;        // probe edges a step can have
constexpr uint32_t kInfoFar = 1u << 15;         // node info: max_len (bits 0..8) | distinct distances << 9 | has edges longer than 64
// row entry: low word the distance, high word
constexpr uint32_t kRowValid = 1u << 31;        //   dict price (13 bits) | rep price << 13 | sampled edge << 26 | valid
// link of a lane / a retired node: source | rep << 13 | explicit probe << 14 | sampled edge (or rep slot of a probe) << 26
constexpr uint32_t kLinkRep = 1u << 13, kLinkProbe = 1u << 14;

struct PFin { uint32_t r[4]; uint32_t cost, link, wdist, pad; };        // a retired node: rep set, final cost, winner
struct PVEdge { unsigned long long key; uint32_t r; uint32_t info; };   // a probe's match: key, distance, node | length << 13 | slot << 22
struct PCtl {
    uint32_t n_s;                               // the step starts at this node; everything before it is final
    uint32_t stop;                              // chain: 0 while it runs, then 1 + the node it stopped at
    uint32_t done;                              // chain: nodes below this are retired (in this attempt)
    uint32_t end_p;                             // chain, with stop: the segment's end so far (:1550-1554)
    uint32_t vcount;                            // probe edges of the attempt
    uint32_t xcount;                            // ... of the attempt before, handed to the chain when the step is done again
    uint32_t redo, seg_len, end_snap;
    uint32_t far_hi, far_hi_snap;               // no node beyond this one has a key in far[]
    uint32_t prep_cur[kPrepWaves];              // rows: wave p has prepared its nodes (n % 4 == p) below this
};

struct PLds {
    unsigned long long row[kRowRing * 64];      // node n, edge of length l, at [(n & 127) * 64 + (n + l) % 64]: the target's lane
    unsigned long long far[kFarRing], far_snap[kFarRing];   // node t at [t & 511]: best key of the edges that jump over the window
    uint32_t far_d[kFarRing], far_d_snap[kFarRing];         // ... and the distance of that edge (sampled edges; a probe's is in its source's set)
    PFin fin[kFarRing];                         // node n at [n & 511]
    uint32_t snap[7 * 64];                      // the chain's registers at the start of the step
    uint32_t dd[kRowRing * 8];                  // the first eight distinct distances of a node's valid sampled edges
    uint32_t ninfo[kRowRing];
    PVEdge vlist[kVMax], extras[kVMax];
    PCtl ctl;
    uint32_t sh[16];                            // 4 error, 12..15 model rep set after the segment
    uint16_t node_link[kParseMax + 2];          // final nodes: length (0: literal) | cmd << 9 | rep index << 11
    uint32_t node_delta[kParseMax + 2];         // distance (dict, rep), the byte (literal)
    uint16_t cdf[kNumCtx * kCdfStride];
    uint16_t price[kNumCtx * 16];               // log2_lut[freq >> 6] per (context, symbol) (:435-438)
    uint16_t lut[256];
    uint16_t len_price[kMatchMax + 8];          // price of the length symbols by length value (:1214-1225)
    uint16_t slot_price[4 * 64];                // price of the two distance-slot symbols by (length class, slot) (:1245-1248)
    uint32_t ncmds;
    uint32_t bitbuf[64];                        // raw bits of a batch of commands (wave 7)
    uint32_t dbgw[4];                           // error dump: the segment being parsed, its step, the node count so far
    unsigned long long acc[11];                 // cycles waited / emitting / in set-up / in the chain; steps, attempts, ...
    unsigned long long fpm[4];                  // single-literal runs: per probe wave, the positions where its rep slot found a match
    uint32_t stg[5];                            // 4: the loader wave has staged the records below this position (0..3 unused)
    unsigned long long stage[kStagePos * kStageQ];  // the table stage's records of the positions from the current step on, loaded ahead
                                                //   by the loader wave (position a at [((a - launch start) & 127) * kStageQ])
    // the path of a parsed segment (node indices, end first) lives in the rows: nothing reads a row between a parse and its emission
    XW_FN uint16_t *cmdlist() { return (uint16_t *)&row[0]; }
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
    // wave kLoaderWave: the loader of the record stage (8-byte words counted from the launch's first position)
    uint32_t pend_t[2];                         // (t_out as read with a step's loads)
    unsigned long long pend_v[2][kPumpLoads];
    uint32_t ld_req, ld_wr, ld_n[2], ld_turn;    // records requested / written up to this position; records a set's loads are for; the set whose turn it is
    // frame writer (CodeFrame, :490-513)
    uint32_t *fsyms; uint8_t *fbits;
    uint32_t nsyms, nbits, word, word_bits, num_ops;       // (every wave follows the counts; `word`, the bits of the byte not yet full, is wave 7's)
    uint32_t err;
    uint32_t n_eq_rounds;                       // (per launch)
    uint32_t n_cmp;                             // (per lane and launch: well below 2^32)
    // the stage's accounting lives in LDS (L()->acc: it is touched once a step or less, and scalar registers are short)
    enum { kAccWait, kAccEmit, kAccSetup, kAccPass, kAccBlocks, kAccPasses, kAccRedo, kAccUndo, kAccTotal, kAccNeed, kAccAhead, kAccN };
    XW_FN void acc(uint32_t k, unsigned long long v) { if (xw::lane() == 0) xw::lds_add64(&L()->acc[k], v); }
    unsigned long long t_q[6] = {};             // profile build, wave 0: start of a step, chain, waiting for the others, decide; nodes, far / extra paths
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
                cmd = (link >> 9) & 3u; len = link & 0x1FFu; idx = link >> 11;
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

    // ---- the record stage (wave kLoaderWave).  One step: what was requested by the last step goes into LDS, the next 64 words are
    // requested (positions below the table stage's t_out as the last step saw it, and below lo + 128: lo is the first
    // position still needed), t_out is requested again.  Nothing is waited for here except the last step's loads.
    XW_FN unsigned long long *staged(uint32_t a) const { return L()->stage + (a & (kStagePos - 1)) * kStageQ; }
    XW_FN uint32_t staged_hi() const { return xw::readfirst(xw::lds_ld(&L()->stg[4])); }
    // Two such steps are in flight (two sets of load registers taken in turn: 48 records), so that a step's loads have a whole
    // other step to arrive in.
    template <uint32_t kSet> XW_FN void pump_set(uint32_t lo)
    {
        // (counted in records: a load instruction fetches the 18 words of three records on lanes 0..53, so that a lane's
        //  place in a record is the same in every step and no step divides by 18)
        const uint32_t i = xw::lane();
        const uint32_t ri = i / kStageQ, wi = i - ri * kStageQ;
        const bool on = i < kPumpRecs * kStageQ;
        const uint32_t pend_n = ld_n[kSet];
        if (pend_n) {
#pragma unroll
            for (uint32_t u = 0; u < kPumpLoads; u++) {
                const uint32_t rr = kPumpRecs * u + ri;
                if (on && rr < pend_n) L()->stage[((ld_wr + rr) & (kStagePos - 1)) * kStageQ + wi] = pend_v[kSet][u];
            }
            ld_wr += pend_n;
        }
        t_out_seen = xw::readfirst(pend_t[kSet]);                   // (as this set's last step read it)
        xw::after_poll();
        uint32_t lim = lo + kStagePos;
        if ((int32_t)(t_out_seen - lim) < 0) lim = t_out_seen;
        const uint32_t n = (int32_t)(lim - ld_req) > 0 ? umin(kPumpRecs * kPumpLoads, lim - ld_req) : 0u;
        const uint32_t wsrc = wi == 0 ? 0u : (wi == kStageQ - 1 ? kTpUniq / 2 : kTpEdges / 2 + wi - 1);   // (staged: header, sixteen edges, mask)
#pragma unroll
        for (uint32_t u = 0; u < kPumpLoads; u++) {
            const uint32_t rr = kPumpRecs * u + ri;
            if (on && rr < n)
                pend_v[kSet][u] = xw::ld_agent64((const unsigned long long *)(V.tp + (unsigned long long)((ld_req + rr) & (kTpRing - 1)) * kTpStride) + wsrc);
        }
        pend_t[kSet] = xw::ld_agent(&V.hx->t_out);
        ld_n[kSet] = n; ld_req += n;
        xw::wave_sync();                                            // (the records are in LDS before the word that says so)
        if (i == 0) xw::lds_st(&L()->stg[4], ld_wr);
    }
    XW_FN void pump(uint32_t lo)
    {
        if (ld_turn) pump_set<1>(lo); else pump_set<0>(lo);
        ld_turn ^= 1u;
    }
    XW_FN bool pump_idle() const { return ld_n[0] + ld_n[1] == 0; }
    // until the record of position a is staged (false: another stage failed, or the wait timed out)
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
#ifndef NLZM_SIM
                if (xw::clock100() - t0 > 3000000000ull) {            // 30 s
                    if (xw::lane() == 0) raise(V.hx, kErrTimeout * 100 + 4, kStParser, 4, a, staged_hi(), t_out_seen);
                    ok = false; break;
                }
#else
                (void)t0;
#endif
            }
            if (pump_idle()) xw::pause();
        }
        acc(kAccWait, xw::tick() - tw);
        acc(kAccNeed, 1); acc(kAccAhead, t_out_seen - a);             // (how far the table stage was ahead when this stage had to wait for its loader)
        return ok;
    }

    // ---- explicit rep probe (:1598-1628): match length of (position a, distance r), at most c bytes ------------------
    // Every lane compares for itself, eight bytes a round: `own` holds the eight bytes at a, `first` the eight bytes at a - r.
    // Nearly every probe ends inside its first eight bytes; the others go on round by round.
    XW_FN uint32_t probe_finish(bool want, uint32_t a, uint32_t r, uint32_t c, unsigned long long own, unsigned long long first)
    {
        uint32_t len = 0;
        bool open = want && c > 0;
        unsigned long long x = first, y = own;
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

    // =============================================================================================
    // rows (waves 3, 5, 6, 7): the record of node m of the segment at seg_a -> its row, distinct distances, info word
    // =============================================================================================
    XW_FN void prep_node(uint32_t seg_a, uint32_t m, uint32_t max_parse, uint32_t pc_dict, uint32_t pc_rep, uint32_t pc_lit)
    {
        PLds *Lp = L();
        const uint32_t k = xw::lane();
        const uint32_t a = seg_a + m;
        unsigned long long *srec = staged(a);
        const unsigned long long hd = srec[0];
        uint32_t ne = (uint32_t)hd & 63u;
        const uint32_t lit = ((uint32_t)hd >> 8) & 0xFFu, mt_max = ((uint32_t)hd >> 16) & 0x1FFu;
        const uint32_t listed = ne ? ((uint32_t)(srec[1] >> 32) & 0x1FFu) : 0u;
        uint32_t eff = umin(mt_max, max_parse - m);                 // :1545-1548
        if (eff < kMatchMin) eff = 0;
        const uint32_t *rec = V.tp + (unsigned long long)(a & (kTpRing - 1)) * kTpStride;
        unsigned long long e = 0;
        uint32_t uniq = 0;
        if (xw::readfirst(eff != listed ? 1u : 0u)) {
            // The segment is within 264 of its forced cut (:1469), or this position was listed for the cut of a segment that
            // ended before it: the sampled lengths change with max_len (:1545, :1558-1560).  Re-listed from the position's
            // table (lanes = samples); the staged copy takes the new list (the ring's record stays as the table stage wrote it).
            acc(kAccUndo, 1);
            const uint32_t *dense = V.tf + (unsigned long long)(a & (kTpRing - 1)) * kTfStride;     // the position's table: delta[l] at [l - 1]
            uint32_t step = (eff - kMatchMin) >> 4;
            step += step == 0;
            ne = eff >= kMatchMin ? (eff - kMatchMin) / step + 1 : 0u;
            uint32_t d = 0, valid = 0;
            if (k < ne) {
                const uint32_t tl = eff - k * step;
                d = xw::ld_agent(dense + tf_index(tl));
                const uint32_t mm = match_min(d);
                uint32_t nx, ex;
                const uint32_t slot = dist_slot(d - 1, nx, ex);
                valid = tl >= mm ? 1u : 0u;
                const uint32_t lv = valid ? tl - mm : 0u;
                e = (unsigned long long)d | ((unsigned long long)(tl | (lv << 9) | (slot << 18) | (nx << 24) | (valid << 31)) << 32);
                if (k < kStageEdges) srec[1 + k] = e;
            }
            // the valid samples that bring a distance the valid one before did not have (as the table stage lists them)
            const unsigned long long vm = xw::ballot(valid != 0), below = vm & ((1ull << k) - 1ull);
            const uint32_t dprev = xw::shfl(d, below ? 63u - (uint32_t)__builtin_clzll(below) : k);
            uniq = (uint32_t)xw::ballot(valid && (!below || d != dprev));
            if (k == 0) { srec[0] = (hd & ~63ull) | ne; srec[kStageQ - 1] = uniq; }
        } else {
            if (k < ne) e = k < kStageEdges ? srec[1 + k] : xw::ld_agent64((const unsigned long long *)(rec + kTpEdges + 2 * k));
            uniq = ne ? (uint32_t)srec[kStageQ - 1] : 0u;
        }
        unsigned long long *row = Lp->row + (m & (kRowRing - 1)) * 64;
        row[k] = 0;
        xw::wave_sync();
        const uint32_t d = (uint32_t)e, at = (uint32_t)(e >> 32);
        if (at >> 31) {
            const uint32_t tl = at & 0x1FFu, lv = (at >> 9) & 0x1FFu, slot = (at >> 18) & 63u, nx = (at >> 24) & 31u;
            const uint32_t lp = Lp->len_price[lv];
            const uint32_t pd = pc_dict + lp + (nx << 5) + Lp->slot_price[umin(lv, 3) * 64 + slot];     // :1208-1251
            const uint32_t pr = pc_rep + lp + (2u << 5);                                                 // :1253-1272
            if (tl <= 64) row[(m + tl) & 63u] = (unsigned long long)d | ((unsigned long long)(pd | (pr << 13) | (k << 26) | kRowValid) << 32);
        }
        if (k == 0) {
            const uint32_t plit = pc_lit + price(kCtxLitHi, lit >> 4) + price(kCtxLitLo + (lit >> 4), lit & 15);     // :1418-1426
            row[(m + 1) & 63u] = (unsigned long long)(kLitMark | lit) | ((unsigned long long)(plit | (0x1FFFu << 13) | kRowValid) << 32);
        }
        if (k < kMaxEdges && ((uniq >> k) & 1u)) {
            const uint32_t at_dd = (uint32_t)__builtin_popcount(uniq & ((1u << k) - 1u));
            if (at_dd < 8) Lp->dd[(m & (kRowRing - 1)) * 8 + at_dd] = d;
        }
#ifdef DBG_PROBE_SIM
        if (seg_a + m - base == 100 && k < 3) fprintf(stderr, "P m %u k %u eff %u listed %u ne %u uniq %x e %llx\n", m, k, eff, listed, ne, uniq, e);
#endif
        if (k == 0) Lp->ninfo[m & (kRowRing - 1)] = eff | ((uint32_t)__builtin_popcount(uniq) << 9) | (eff > 64 ? kInfoFar : 0u);
    }

    // =============================================================================================
    // chain (wave 0)
    // =============================================================================================
    struct Chain { uint32_t cost, link, wd, r0, r1, r2, r3; };
    XW_FN static uint32_t rank_of(uint32_t link, uint32_t wd)
    {
        if (wd >= kLitMark) return kRankLit;
        if (link & kLinkProbe) return kRankProbe + ((link >> 26) & 3u);
        return 2 * ((link >> 26) & 31u) + ((link >> 13) & 1u);
    }
    XW_FN static unsigned long long key_of(uint32_t cost, uint32_t link, uint32_t wd)
    {
        return ((unsigned long long)cost << 32) | ((link & 0x1FFFu) << 8) | rank_of(link, wd);
    }
    // the lanes of `take` become what the key fk of node t says (an edge that jumped over the window, or a probe's)
    XW_FN void take_far(bool take, unsigned long long fk, uint32_t t, Chain &C) const
    {
        if (!take) return;
        const PLds *Lp = L();
        const uint32_t src = (uint32_t)(fk >> 8) & 0x1FFFu, rank = (uint32_t)fk & 0xFFu;
        const PFin &fs = Lp->fin[src & (kFarRing - 1)];
        const uint32_t s0 = fs.r[0], s1 = fs.r[1], s2 = fs.r[2], s3 = fs.r[3];
        C.cost = (uint32_t)(fk >> 32);
        C.r0 = s0; C.r1 = s1; C.r2 = s2; C.r3 = s3;
        if (rank >= kRankProbe) {
            const uint32_t idx = rank - kRankProbe;
            C.wd = idx == 0 ? s0 : (idx == 1 ? s1 : (idx == 2 ? s2 : s3));
            C.link = src | kLinkRep | kLinkProbe | (idx << 26);
        } else {
            const uint32_t d = Lp->far_d[t & (kFarRing - 1)];
            C.wd = d;
            C.link = src | ((rank & 1u) ? kLinkRep : 0u) | ((rank >> 1) << 26);
            if (!(rank & 1u) && !(d == s0 || d == s1 || d == s2 || d == s3)) { C.r0 = d; C.r1 = s0; C.r2 = s1; C.r3 = s2; }
        }
    }
    // node of lane i when node n is the next to be processed
    XW_FN static uint32_t lane_node(uint32_t n, uint32_t i) { return n + ((i - n) & 63u); }
    // keys that arrived for the nodes of the window (probe edges of the step before; far edges that were pushed while the node was
    // outside): every lane looks at its node's key
    XW_FN void merge_window(uint32_t n, Chain &C) const
    {
        const uint32_t t = lane_node(n, xw::lane());
        const unsigned long long fk = L()->far[t & (kFarRing - 1)];
        const bool take = fk != kKeyNone && fk < key_of(C.cost, C.link, C.wd);
        if (xw::any(take)) take_far(take, fk, t, C);
    }

    // One step of the chain: nodes n_s .. (below n_e, and while inside the segment).  Returns the node it stopped at.
    // Every lane keeps the state of its node in LDS up to date (fin[node & 511]: once a step, whatever the node): the entry of a
    // node is final from the step the node is processed in -- no lane mask, no branch for the retirement.
    XW_FN uint32_t chain_run(Chain &C, uint32_t seg_a, uint32_t n_s, uint32_t n_e_in, uint32_t &end_p_io, uint32_t &far_hi_io, uint32_t nx_extra,
                             uint32_t pc_dict, uint32_t pc_rep)
    {
        PLds *Lp = L();
        const uint32_t i = xw::lane();
        // (loop control in scalar registers)
        const uint32_t n_e = xw::readfirst(n_e_in);
        uint32_t end_p = xw::readfirst(end_p_io), far_hi = xw::readfirst(far_hi_io);
        // (lane m % 64: what the rows' wave noted about node m of this step; bit 16: the node has probe edges handed over --
        //  the step is being done again)
        uint32_t infov = Lp->ninfo[lane_node(n_s, i) & (kRowRing - 1)];
        for (uint32_t b = 0; b < nx_extra; b += 64) {
            unsigned long long bit = b + i < nx_extra ? 1ull << ((Lp->extras[b + i].info & 0x1FFFu) & 63u) : 0ull;
            for (uint32_t sh = 32; sh; sh >>= 1) bit |= xw::shfl64(bit, i ^ sh);
            infov |= (uint32_t)((bit >> i) & 1ull) << 16;
        }
        unsigned long long fkv = far_hi >= n_s + 64 ? Lp->far[(n_s + 64) & (kFarRing - 1)] : kKeyNone;      // the key waiting for the node that enters the window (the same in every lane)
        uint32_t n = n_s;
        const unsigned long long *rows = Lp->row + i;               // (a row's entry for this lane: the edge that ends at the lane's node)
        unsigned long long rw0 = rows[(n & (kRowRing - 1)) * 64];
        unsigned long long rw1 = rows[((n + 1) & (kRowRing - 1)) * 64];
        uint32_t fslot = lane_node(n, i) & (kFarRing - 1);          // where this lane's node keeps its state
        xw::setprio_high();
        while (n < n_e && n < end_p) {
            const uint32_t sl = n & 63u;
            const bool me = i == sl;
            const uint32_t cn = xw::readlane(C.cost, sl);
            const uint32_t r0 = xw::readlane(C.r0, sl), r1 = xw::readlane(C.r1, sl), r2 = xw::readlane(C.r2, sl), r3 = xw::readlane(C.r3, sl);
            const uint32_t info = xw::readlane(infov, sl);
            // ---- the states as they stand (node n's is final), and how far the chain is
            {
                PFin &f = Lp->fin[fslot];
                f.r[0] = C.r0; f.r[1] = C.r1; f.r[2] = C.r2; f.r[3] = C.r3;
                f.cost = C.cost; f.link = C.link; f.wdist = C.wd;
                xw::lds_st(&Lp->ctl.done, n + 1);
            }
            end_p = umax(end_p, n + (info & 0x1FFu));               // :1550-1554
            const unsigned long long rw2 = rows[((n + 2) & (kRowRing - 1)) * 64];     // (two nodes ahead)
            // ---- its lane is node n + 64's now
            C.cost = me ? kInf : C.cost;
            fslot = me ? ((fslot + 64) & (kFarRing - 1)) : fslot;
            // ---- its edges (:1490-1499, :1566-1595): every lane its own
            auto relax = [&]() __attribute__((always_inline)) {
                const uint32_t d = (uint32_t)rw0, at = (uint32_t)(rw0 >> 32);
                const bool valid = (int32_t)at < 0;
                const bool inset = d == r0 || d == r1 || d == r2 || d == r3;
                const uint32_t pd = at & 0x1FFFu, pr = (at >> 13) & 0x1FFFu;
                const bool userep = inset && pr < pd;               // (dict is tried first: rep only if strictly cheaper)
                const uint32_t cand = cn + (userep ? pr : pd);
                const bool upd = valid && cand < C.cost;
                const bool push = upd && !(inset || d >= kLitMark); // RepModel::Add of a new distance
                C.cost = upd ? cand : C.cost;
                C.link = upd ? (n | (userep ? kLinkRep : 0u) | (at & 0x7C000000u)) : C.link;
                C.wd = upd ? d : C.wd;
                // (the set once more in vector registers: a select takes its mask over the scalar operand bus, so its data cannot)
                const uint32_t v0 = xw::opaque(r0), v1 = xw::opaque(r1), v2 = xw::opaque(r2), v3 = xw::opaque(r3);
                C.r3 = upd ? (push ? v2 : v3) : C.r3; C.r2 = upd ? (push ? v1 : v2) : C.r2; C.r1 = upd ? (push ? v0 : v1) : C.r1; C.r0 = upd ? (push ? d : v0) : C.r0;
            };
            if (!NLZM_RARE((info >> 15) | (far_hi >= n + 64 ? 1u : 0u))) relax();        // (no edge longer than the window, no probe edge handed over, no key in far[])
            else {
                // a key that jumped over the window may be waiting for node n + 64 (written before this node was reached; requested a node ago)
                const unsigned long long fk0 = xw::readfirst64(fkv);
                if (fk0 != kKeyNone) take_far(me, fk0, n + 64, C);
                relax();
                // ---- sampled edges longer than the window (max_len > 64: lanes = the sixteen longest samples)
                if (info & kInfoFar) {
                    t_q[5]++;
                    const unsigned long long *srec = staged(seg_a + n);
                    const uint32_t ne = (uint32_t)srec[0] & 63u;
                    const unsigned long long e = (i < kStageEdges && i < ne) ? srec[1 + i] : 0ull;
                    const uint32_t d = (uint32_t)e, at = (uint32_t)(e >> 32), tl = at & 0x1FFu;
                    uint32_t t = 0;
                    if ((at >> 31) && tl > 64) {
                        const uint32_t lv = (at >> 9) & 0x1FFu, slot = (at >> 18) & 63u, nx = (at >> 24) & 31u;
                        const uint32_t lp = Lp->len_price[lv];
                        const uint32_t pd = pc_dict + lp + (nx << 5) + Lp->slot_price[umin(lv, 3) * 64 + slot];
                        const uint32_t pr = pc_rep + lp + (2u << 5);
                        const bool inset = d == r0 || d == r1 || d == r2 || d == r3;
                        const bool userep = inset && pr < pd;
                        const unsigned long long key = ((unsigned long long)(cn + (userep ? pr : pd)) << 32) | (n << 8) | (2 * i + (userep ? 1u : 0u));
                        t = n + tl;
                        if (key < Lp->far[t & (kFarRing - 1)]) { Lp->far[t & (kFarRing - 1)] = key; Lp->far_d[t & (kFarRing - 1)] = d; }
                    }
                    (void)t;
                    far_hi = umax(far_hi, n + (info & 0x1FFu));     // (no key beyond the node's longest edge)
                    xw::wave_sync();
                }
                // ---- probe edges known from the attempt before (:1598-1628: after the sampled edges, slot by slot)
                if (info & (1u << 16)) {
                    t_q[5]++;
#pragma unroll
                    for (uint32_t pi = 0; pi < 4; pi++) {
                        const uint32_t r = pi == 0 ? r0 : (pi == 1 ? r1 : (pi == 2 ? r2 : r3));
                        uint32_t ml = 0;
                        for (uint32_t b = 0; b < nx_extra; b += 64) {
                            const bool hit = b + i < nx_extra && (Lp->extras[b + i].info & 0x1FFFu) == n && Lp->extras[b + i].r == r;
                            const unsigned long long hm = xw::ballot(hit);
                            if (hm) ml = (xw::readlane(b + i < nx_extra ? Lp->extras[b + i].info : 0u, (uint32_t)__builtin_ctzll(hm)) >> 13) & 0x1FFu;
                        }
                        if (!ml) continue;
                        const uint32_t cand = cn + pc_rep + Lp->len_price[ml - match_min(r)] + (2u << 5);        // :1607, :1614
                        const uint32_t t = n + ml;
                        end_p = umax(end_p, t);                     // :1608-1612
                        if (ml <= 64) {
                            const bool upd = i == (t & 63u) && cand < C.cost;
                            C.cost = upd ? cand : C.cost;
                            C.link = upd ? (n | kLinkRep | kLinkProbe | (pi << 26)) : C.link;
                            C.wd = upd ? r : C.wd;
                            C.r0 = upd ? r0 : C.r0; C.r1 = upd ? r1 : C.r1; C.r2 = upd ? r2 : C.r2; C.r3 = upd ? r3 : C.r3;
                        } else {
                            const unsigned long long key = ((unsigned long long)cand << 32) | (n << 8) | (kRankProbe + pi);
                            if (i == 0 && key < Lp->far[t & (kFarRing - 1)]) Lp->far[t & (kFarRing - 1)] = key;
                            far_hi = umax(far_hi, t);
                            xw::wave_sync();
                        }
                    }
                }
                // (the key for the node that enters the window next: after this node's own keys are written)
                fkv = far_hi >= n + 65 ? Lp->far[(n + 65) & (kFarRing - 1)] : kKeyNone;
            }
            rw0 = rw1; rw1 = rw2;
            n++;
        }
        xw::setprio_low();
        // (the node the chain stopped at: its state as it stands, for the step's end)
        {
            PFin &f = Lp->fin[fslot];
            f.r[0] = C.r0; f.r[1] = C.r1; f.r[2] = C.r2; f.r[3] = C.r3;
            f.cost = C.cost; f.link = C.link; f.wdist = C.wd;
        }
        end_p_io = end_p; far_hi_io = far_hi;
        return n;
    }

    // =============================================================================================
    // probes (waves 1..2): the four explicit rep probes (:1598-1628) of nodes nb0 .. hi-1, lanes = node x slot
    // =============================================================================================
    XW_FN void verify_batch(uint32_t seg_a, uint32_t seg_q, uint32_t max_parse, uint32_t nb0, uint32_t hi, uint32_t pc_rep, uint32_t &cmp_acc)
    {
        PLds *Lp = L();
        const uint32_t i = xw::lane(), pi = i & 3u;
        const uint32_t node = nb0 + (i >> 2);
        const bool on = node < hi;
        uint32_t r = 1, cost = 0;
        bool want = false;
        if (on) {
            const PFin &f = Lp->fin[node & (kFarRing - 1)];
            const uint32_t s0 = f.r[0], s1 = f.r[1], s2 = f.r[2], s3 = f.r[3];
            r = pi == 0 ? s0 : (pi == 1 ? s1 : (pi == 2 ? s2 : s3));
            cost = f.cost;
            if (pi == 0 && node > 0) {
                // what the emitter needs of the node's winner: length (0: literal) | cmd << 9 | rep index << 11; distance or byte
                const uint32_t link = f.link, wd = f.wdist;
                uint32_t out = 0, delta = wd;
                if (wd >= kLitMark) delta = wd & 0xFFu;
                else {
                    const uint32_t len = node - (link & 0x1FFFu);
                    if (link & kLinkRep) out = len | (2u << 9) | ((wd == s0 ? 0u : (wd == s1 ? 1u : (wd == s2 ? 2u : 3u))) << 11);   // (:1173-1181: the first slot that holds it)
                    else out = len | (1u << 9);
                }
                Lp->node_link[node] = (uint16_t)out; Lp->node_delta[node] = delta;
            }
            // a slot whose distance a valid sampled edge of the node has is not probed (:1580-1584, :1600)
            const uint32_t nd = (Lp->ninfo[node & (kRowRing - 1)] >> 9) & 63u;
            bool met = false;
            if (nd <= 8) {
#pragma unroll
                for (uint32_t z = 0; z < 8; z++) met = met || (z < nd && Lp->dd[(node & (kRowRing - 1)) * 8 + z] == r);
            } else {
                // (more than eight distinct distances: the row's entries and the staged samples longer than the window)
                const unsigned long long *row = Lp->row + (node & (kRowRing - 1)) * 64;
                for (uint32_t z = 0; z < 64; z++) met = met || (uint32_t)row[z] == r;
                const unsigned long long *srec = staged(seg_a + node);
                const uint32_t ne = (uint32_t)srec[0] & 63u;
                for (uint32_t z = 0; z < kStageEdges && z < ne; z++) {
                    const unsigned long long e = srec[1 + z];
                    met = met || ((uint32_t)(e >> 32) >> 31 && (uint32_t)e == r);
                }
            }
            want = !met && r < seg_q + node;                        // :1601
        }
        const uint32_t pcap = on ? umin(max_parse - node, kMatchMax) : 0u;                              // :1605-1606
        const uint32_t a = seg_a + node;
        uint32_t ml = 0;
        if (xw::any(want)) {
            const unsigned long long yo = want ? load64u(G.in + a) : 0ull, xo = want ? load64u(G.in + a - r) : 0ull;
            ml = probe_finish(want, a, r, pcap, yo, xo);
        }
        if (want) cmp_acc += ml + (ml < pcap);
#ifdef DBG_PROBE_SIM
        if (want && seg_q + node == 100) fprintf(stderr, "D node %u info %x dd0 %u uniq? ne %u\n", node, Lp->ninfo[node & (kRowRing - 1)], Lp->dd[(node & (kRowRing - 1)) * 8], (uint32_t)staged(seg_a + node)[0] & 63u);
        if (want) fprintf(stderr, "S %u %u %u %u\n", seg_q + node, pi, r, ml);
#endif
        if (want && ml >= match_min(r)) {
            const uint32_t pw = pc_rep + Lp->len_price[ml - match_min(r)] + (2u << 5);                 // :1607, :1614
            const uint32_t at = xw::lds_inc(&Lp->ctl.vcount);
            PVEdge &v = Lp->vlist[at];
            v.key = ((unsigned long long)(cost + pw) << 32) | (node << 8) | (kRankProbe + pi);
            v.r = r; v.info = node | (ml << 13) | (pi << 22);
        }
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
        PLds *Lp = L();
        const unsigned long long ts = xw::tick();
        // A segment starts at seg_a: said BEFORE this stage asks for the position's record.  The finder stage may be waiting
        // for exactly this word at seg_a (a nice region that starts where the segment before was cut at 4,096 positions,
        // :1469: no edge spans the cut, so nothing else tells it) and the record of seg_a comes only after it.
        if (tid == 0) xw::st_agent64(&V.hx->p_seg, ((unsigned long long)seg_a << 32) | (seg_a + 1));
        if (w == kLoaderWave) {                                     // (meanwhile: the first node's record, staged)
            if (!stage_need(seg_a)) err = kErrInternal + 100;
            if (i == 0) Lp->sh[4] = err;
        }
        seg_tables(tab_dirty);                                      // (tab_dirty is the same in every wave: run_chunk, run)
        tab_dirty = false;
        const uint32_t pc_dict = price(kCtxCmd, 1), pc_rep = price(kCtxCmd, 2), pc_lit = price(kCtxCmd, 0);
        if (Lp->sh[4]) { err = kErrInternal + 100; return 0; }
        {
            // A position without any match is a segment of its own (more than half of all segments are): its node has no
            // sampled edge, and if none of the four rep probes finds anything (:1598-1628) the only command is the literal.
            // Literals leave the rep set alone and prices do not matter here, so a RUN of such positions is found at once
            // (lanes = the positions from seg_a on; about half of these segments follow another one directly).
            const uint32_t sh_hi = staged_hi();
            const uint32_t kmax = umin(64u, umin(sh_hi - seg_a, chunk_left));
            const uint32_t hj = i < kmax ? (uint32_t)staged(seg_a + i)[0] : 1u;
            const unsigned long long withm = xw::ballot((hj & 63u) != 0 || ((hj >> 16) & 0x1FFu) >= kMatchMin);   // (lanes >= kmax count as having a match;
                                                                    //  a table a forced cut left without samples still has its matches)
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
                        l = probe_finish(want, seg_a + i, r, pcap, yo, xo);
                    }
                    if (want) { counted = l + (l < pcap); n_cmp += counted; }
                    const unsigned long long okm = xw::ballot(want && l >= match_min(r));
                    if (i == 0) Lp->fpm[w] = okm;
                }
                xw::block_sync();
                const unsigned long long hit = Lp->fpm[0] | Lp->fpm[1] | Lp->fpm[2] | Lp->fpm[3];
                const uint32_t run = hit ? umin(p0, (uint32_t)__builtin_ctzll(hit)) : p0;           // positions that are segments of one literal
                // (fpm is written again only after another barrier: the one below, or the chain's)
                if (i >= run) n_cmp -= counted;     // (not part of the run: their probes are made, and counted, when their segment is parsed)
                if (run) {
                    if (w == 0 && i < run) {
                        Lp->node_link[i + 1] = 0; Lp->node_delta[i + 1] = (hj >> 8) & 0xFFu;         // literal: from the node before, the byte
                        Lp->cmdlist()[run - 1 - i] = (uint16_t)(i + 1);
                    }
                    if (tid == 0) {
                        Lp->sh[12] = rep0; Lp->sh[13] = rep1; Lp->sh[14] = rep2; Lp->sh[15] = rep3;    // (literals leave the rep set alone)
                        Lp->ncmds = run;
                        xw::st_agent(&V.hx->p_pos, seg_a);
                        if (run > 1) xw::st_agent64(&V.hx->p_seg, ((unsigned long long)(seg_a + run - 1) << 32) | (seg_a + run));
                    }
                    xw::block_sync();
                    ncmds = run;
                    nsegs = run;
#ifdef NLZM_SIM
                    if (tid == 0) for (uint32_t c = 0; c < run; c++) { xw::trace(4, seg_a + c, 1, 1); xw::trace(5, 0, 0, 0); }
#endif
                    return run;
                }
            }
        }
        // ---- the chain parse.  State shared by the waves: Lp->ctl
        // The rows of the first nodes are made by all six helper waves before anything else: as many as the first node's table
        // says the segment has at least (and the records that are there)
        uint32_t n_first;
        {
            const uint32_t h0 = (uint32_t)staged(seg_a)[0];
            uint32_t eff0 = umin((h0 >> 16) & 0x1FFu, max_parse);
            n_first = umin(umin(umax(eff0 + 2, 4u), kStepWant), umin(staged_hi() - seg_a, max_parse));
        }
        if (w != 0 && w != kLoaderWave) for (uint32_t m = w < kLoaderWave ? w - 1 : w - 2; m < n_first; m += kPW - 2) prep_node(seg_a, m, max_parse, pc_dict, pc_rep, pc_lit);
        for (uint32_t t = tid; t < kFarRing; t += kParserThreads) Lp->far[t] = kKeyNone;
        if (tid == 0) {
            PCtl &c = Lp->ctl;
            c.n_s = 0; c.stop = 0; c.done = 0; c.end_p = 1; c.vcount = 0; c.xcount = 0; c.redo = 0; c.seg_len = 0; c.end_snap = 1; c.far_hi = 0; c.far_hi_snap = 0;
            for (uint32_t p = 0; p < kPrepWaves; p++) c.prep_cur[p] = n_first + ((p - n_first) & (kPrepWaves - 1));
            Lp->dbgw[0] = seg_a; Lp->dbgw[1] = 0; Lp->dbgw[2] = max_parse;
            Lp->node_link[0] = 0;
        }
        Chain C{ kInf, kSrcNone, 0, rep0, rep1, rep2, rep3 };       // (wave 0's; node 0 (:1472-1482) on lane 0)
        if (i == 0) C.cost = 0;
        uint32_t end_p = 1, seg_len = 0, far_hi = 0;
        uint32_t cmp_acc = 0;                                       // probe waves: bytes the probes of the attempt looked at
        const uint32_t prep_i = w == 1 + kVerifyWaves ? 0u : w - kLoaderWave;             // (rows' waves 3, 5, 6, 7: 0 .. 3)
        uint32_t prep_m = n_first + ((prep_i - n_first) & (kPrepWaves - 1));               // ... and the next node of theirs
        xw::block_sync();
        if (w == 0) acc(kAccSetup, xw::tick() - ts);
        for (;;) {
            const uint32_t n_s = xw::readfirst(xw::lds_ld(&Lp->ctl.n_s));
            const bool redo = xw::readfirst(xw::lds_ld(&Lp->ctl.redo)) != 0;
            if (w == 0) {
                // ================= chain =================
                const unsigned long long q0 = ptick();
                const unsigned long long tp0 = xw::tick();
                uint32_t n_e;
                {   // rows that are ready: below the smallest cursor of the rows' waves.  When this stage has caught up with the
                    // table stage it gives it a moment (bounded: the finder may be waiting for this stage's word) rather than run on a few nodes
                    uint32_t spins = 0;
                    for (;;) {
                        uint32_t e = kNone;
                        for (uint32_t p = 0; p < kPrepWaves; p++) e = umin(e, xw::readfirst(xw::lds_ld(&Lp->ctl.prep_cur[p])));
                        n_e = umin(e, umin(n_s + kStepMax, max_parse));
                        const uint32_t want_n = umin(kStepWant, max_parse - n_s);      // (the chain stops by itself where the segment ends)
                        if (n_e >= n_s + want_n || (n_e > n_s && spins >= 8)) break;
                        if ((++spins & 63u) == 0 && (xw::readfirst(xw::ld_agent(&V.hx->err)) || xw::readfirst(xw::lds_ld(&Lp->sh[4])))) { n_e = n_s; break; }
                        xw::pause();
                    }
                    xw::after_poll();
                    if (n_e > n_s) n_e = n_s + xw::readfirst(xw::test_cut(n_e - n_s));    // (identity on the device; the simulation cuts steps at random here, as
                                                                    //  the device does when this stage catches up with the table stage)
                }
                uint32_t nx_extra = 0;
                if (redo) {
                    // the step again, from its start state, with the probe edges the attempt before has found
                    C.cost = Lp->snap[i]; C.link = Lp->snap[64 + i]; C.wd = Lp->snap[128 + i];
                    C.r0 = Lp->snap[192 + i]; C.r1 = Lp->snap[256 + i]; C.r2 = Lp->snap[320 + i]; C.r3 = Lp->snap[384 + i];
                    for (uint32_t t = i; t < kFarRing; t += 64) { Lp->far[t] = Lp->far_snap[t]; Lp->far_d[t] = Lp->far_d_snap[t]; }
                    end_p = xw::readfirst(Lp->ctl.end_snap); far_hi = xw::readfirst(Lp->ctl.far_hi_snap);
                    nx_extra = xw::readfirst(Lp->ctl.xcount);
                    xw::wave_sync();
                } else {
                    merge_window(n_s, C);
                    Lp->snap[i] = C.cost; Lp->snap[64 + i] = C.link; Lp->snap[128 + i] = C.wd;
                    Lp->snap[192 + i] = C.r0; Lp->snap[256 + i] = C.r1; Lp->snap[320 + i] = C.r2; Lp->snap[384 + i] = C.r3;
                    for (uint32_t t = i; t < kFarRing; t += 64) { Lp->far_snap[t] = Lp->far[t]; Lp->far_d_snap[t] = Lp->far_d[t]; }
                    if (i == 0) { Lp->ctl.end_snap = end_p; Lp->ctl.far_hi_snap = far_hi; }
                    xw::wave_sync();
                }
                const unsigned long long q1 = ptick();
                const uint32_t n1 = chain_run(C, seg_a, n_s, n_e, end_p, far_hi, nx_extra, pc_dict, pc_rep);
                if (i == 0) {
                    // (what the finder stage may be waiting for: how far the segment reaches.  Said before anybody here waits for a record)
                    xw::st_agent64(&V.hx->p_seg, ((unsigned long long)seg_a << 32) | (seg_a + end_p));
                    xw::lds_st(&Lp->ctl.end_p, end_p);
                    xw::lds_st(&Lp->ctl.stop, n1 + 1);
                }
                acc(kAccPass, xw::tick() - tp0); acc(kAccPasses, 1);
                const unsigned long long q2 = ptick();
                t_q[0] += q1 - q0; t_q[1] += q2 - q1; t_q[4] += n1 - n_s;
                xw::block_sync();
                const unsigned long long q3 = ptick();
                // ================= decide =================
                const uint32_t cnt = xw::readfirst(xw::lds_ld(&Lp->ctl.vcount));
                bool matters = false;
                for (uint32_t b = 0; b < cnt; b += 64) {
                    bool m = false;
                    if (b + i < cnt) {
                        const PVEdge &v = Lp->vlist[b + i];
                        const uint32_t t = (v.info & 0x1FFFu) + ((v.info >> 13) & 0x1FFu);
                        if (t < n1) {                               // the target was retired without this edge
                            const PFin &f = Lp->fin[t & (kFarRing - 1)];
                            m = v.key < key_of(f.cost, f.link, f.wdist);
                        }
                    }
                    matters = matters || xw::any(m);
                }
                if (cnt && !redo && xw::readfirst(xw::test_cut(1000u)) < 20u) matters = true;     // (simulation with NLZM_SIM_RANDOM_BLOCKS: a step done again for nothing)
                if (matters) {
                    for (uint32_t b = i; b < cnt; b += 64) Lp->extras[b] = Lp->vlist[b];
                    if (i == 0) { Lp->ctl.xcount = cnt; Lp->ctl.redo = 1; Lp->ctl.vcount = 0; Lp->ctl.stop = 0; Lp->ctl.done = n_s; }
                    acc(kAccRedo, 1);
                } else {
                    // (the keys of the nodes that were retired are spent)
                    if (n_s + i < n1) Lp->far[(n_s + i) & (kFarRing - 1)] = kKeyNone;
                    xw::wave_sync();
                    // probe edges whose target is still ahead: into the keys the chain takes over (:1608-1612 opens the nodes)
                    uint32_t ext = 0;
                    for (uint32_t b = 0; b < cnt; b += 64) {
                        uint32_t t = 0;
                        if (b + i < cnt) {
                            const PVEdge &v = Lp->vlist[b + i];
                            t = (v.info & 0x1FFFu) + ((v.info >> 13) & 0x1FFu);
                            if (t >= n1) xw::lds_min64(&Lp->far[t & (kFarRing - 1)], v.key); else t = 0;
                        }
                        ext = umax(ext, xw::readlane(xw::scan_max(t), 63));
                    }
                    end_p = umax(end_p, ext); far_hi = umax(far_hi, ext);
                    xw::wave_sync();
                    acc(kAccBlocks, 1);
                    uint32_t done_len = 0;
                    if (n1 >= end_p) {
                        // the segment ends at node n1: no edges leave it; its key is complete once the keys that waited for it are in
                        merge_window(n1, C);
                        if (i == (n1 & 63u)) {
                            uint32_t out = 0, delta = C.wd;
                            if (C.wd >= kLitMark) delta = C.wd & 0xFFu;
                            else {
                                const uint32_t len = n1 - (C.link & 0x1FFFu);
                                if (C.link & kLinkRep) out = len | (2u << 9) | ((C.wd == C.r0 ? 0u : (C.wd == C.r1 ? 1u : (C.wd == C.r2 ? 2u : 3u))) << 11);
                                else out = len | (1u << 9);
                            }
                            Lp->node_link[n1] = (uint16_t)out; Lp->node_delta[n1] = delta;
                            Lp->sh[12] = C.r0; Lp->sh[13] = C.r1; Lp->sh[14] = C.r2; Lp->sh[15] = C.r3;    // the model's rep set after the segment
                        }
                        done_len = n1;
                    }
                    if (i == 0) {
                        PCtl &c = Lp->ctl;
                        c.n_s = n1; c.redo = 0; c.xcount = 0; c.vcount = 0; c.stop = 0; c.done = n1; c.seg_len = done_len;
                        Lp->dbgw[1] = n1;
                        xw::st_agent(&V.hx->p_pos, seg_a + n1);
                        xw::st_agent64(&V.hx->p_seg, ((unsigned long long)seg_a << 32) | (seg_a + end_p));
                    }
                }
                xw::wave_sync();
                t_q[2] += q3 - q2; t_q[3] += ptick() - q3;
            } else if (w <= kVerifyWaves) {
                // ================= probes =================
                if (redo) cmp_acc = 0;
                else { n_cmp += cmp_acc; cmp_acc = 0; }             // (the attempt before stood)
                for (uint32_t b = w - 1;; b += kVerifyWaves) {
                    const uint32_t nb0 = n_s + b * kVBatch;
                    uint32_t st = 0, spins = 0;
                    for (;;) {
                        st = xw::readfirst(xw::lds_ld(&Lp->ctl.stop));
                        if (st || xw::readfirst(xw::lds_ld(&Lp->ctl.done)) >= nb0 + kVBatch) break;
                        if ((++spins & 255u) == 0 && xw::readfirst(xw::ld_agent(&V.hx->err))) { st = nb0 + 1; break; }
                        xw::pause();
                    }
                    xw::after_poll();
                    const uint32_t hi = st ? umin(nb0 + kVBatch, st - 1) : nb0 + kVBatch;
                    if (nb0 >= hi) break;
                    verify_batch(seg_a, seg_q, max_parse, nb0, hi, pc_rep, cmp_acc);
                }
                xw::block_sync();
            } else if (w != kLoaderWave) {
                // ================= rows =================
                uint32_t spins = 0;
                for (;;) {
                    const uint32_t m = prep_m;
                    if (m < n_s + kRowRing && m < max_parse && (int32_t)(staged_hi() - (seg_a + m + 1)) >= 0) {
                        xw::after_poll();
                        prep_node(seg_a, m, max_parse, pc_dict, pc_rep, pc_lit);
                        prep_m = m + kPrepWaves;
                        xw::wave_sync();
                        if (i == 0) xw::lds_st(&Lp->ctl.prep_cur[prep_i], prep_m);
                        continue;
                    }
                    const uint32_t st = xw::readfirst(xw::lds_ld(&Lp->ctl.stop));
                    if (st) {
                        // (the node the chain stopped at, if the segment goes on there, is the least the next step needs)
                        const uint32_t n1 = st - 1;
                        if (m > n1 || n1 >= xw::readfirst(xw::lds_ld(&Lp->ctl.end_p)) || n1 >= max_parse) break;
                    }
                    if ((++spins & 255u) == 0 && (xw::readfirst(xw::ld_agent(&V.hx->err)) || xw::readfirst(xw::lds_ld(&Lp->sh[4])))) break;
                    xw::pause();
                }
                xw::block_sync();
            } else {
                // ================= loader =================
                uint32_t spins = 0;
                const unsigned long long t0 = xw::clock100();
                for (;;) {
                    pump(seg_a + n_s);
                    const uint32_t st = xw::readfirst(xw::lds_ld(&Lp->ctl.stop));
                    if (st) {
                        const uint32_t n1 = st - 1;
                        if (n1 >= xw::readfirst(xw::lds_ld(&Lp->ctl.end_p)) || n1 >= max_parse || (int32_t)(staged_hi() - (seg_a + n1 + 1)) >= 0) break;
                    }
                    if ((++spins & 63u) == 0) {
                        if (xw::readfirst(xw::ld_agent(&V.hx->err))) { err = kErrInternal + 100; break; }
#ifndef NLZM_SIM
                        if (st && xw::clock100() - t0 > 3000000000ull) {
                            if (i == 0) raise(V.hx, kErrTimeout * 100 + 4, kStParser, 5, seg_a + st - 1, staged_hi(), t_out_seen);
                            err = kErrInternal + 100; break;
                        }
#else
                        (void)t0;
#endif
                    }
                    if (pump_idle()) xw::pause();
                }
                if (err && i == 0) xw::lds_st(&Lp->sh[4], err);
                xw::block_sync();
            }
            xw::block_sync();                                       // (the chain's wave has decided)
            if (xw::readfirst(xw::lds_ld(&Lp->sh[4])) || xw::readfirst(xw::ld_agent(&V.hx->err))) { err = kErrInternal + 100; return 0; }
            seg_len = xw::readfirst(xw::lds_ld(&Lp->ctl.seg_len));
            if (seg_len) break;
        }
        if (w >= 1 && w <= kVerifyWaves) n_cmp += cmp_acc;         // (the last attempt stood)
        // backtrack (:1633-1650): node indices of the path, end first
        if (w == 0) {
            uint32_t n = 0, cur = seg_len;
            while (cur != 0) {
                if (i == 0) Lp->cmdlist()[n] = (uint16_t)cur;
                n++;
                const uint32_t l = xw::readfirst((uint32_t)Lp->node_link[cur]) & 0x1FFu;
                cur -= l ? l : 1u;
            }
            if (i == 0) Lp->ncmds = n;
        }
        xw::block_sync();
        ncmds = Lp->ncmds;
#ifdef NLZM_SIM
        if (tid == 0) {                                             // (simulation: the segment's commands, for a diff against the oracle's)
            xw::trace(4, seg_a, seg_len, ncmds);
            for (uint32_t c = ncmds; c > 0; c--) {
                const uint32_t nd = Lp->cmdlist()[c - 1], lk = Lp->node_link[nd], cmd = (lk >> 9) & 3u;
                xw::trace(5, cmd, lk & 0x1FFu, cmd == 2 ? lk >> 11 : (cmd == 0 ? 0u : Lp->node_delta[nd]));
            }
        }
#endif
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
                if (xw::wave() == kLoaderWave) pump(seg_a + len);               // (records of the next segment: requested now, in LDS by the next step)
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
        ld_req = ld_wr = t_out_seen; ld_n[0] = ld_n[1] = 0; ld_turn = 0;
        if (tid < 16) L()->sh[tid] = 0;
        pend_t[0] = pend_t[1] = t_out_seen;                         // (as if t_out had been read before the table stage started)
        n_eq_rounds = 0; n_cmp = 0;
        if (tid < kAccN) L()->acc[tid] = 0;
        xw::block_sync();
        if (tid == 0) L()->acc[kAccTotal] = 0ull - xw::tick();
        uint32_t ci = c0;
        for (; ci < c1 && !err; ci++) run_chunk(ci);
        xw::block_sync();
#ifdef NLZM_PROFILE
        if (xw::lane() == 0 && xw::wave() == 0) for (int z = 0; z < 6; z++) xw::atomic_add64_agent(&P->prof[56 + z], t_q[z]);
#endif
        {   // bytes the probes looked at, probe rounds: summed over the waves
            unsigned long long n_cmp64 = n_cmp;
            for (uint32_t d = 32; d; d >>= 1) n_cmp64 += xw::shfl64(n_cmp64, xw::lane() ^ d);      // (kept per lane)
            if (xw::lane() == 0) {
                xw::lds_add64(&L()->cnt.cmp_bytes, n_cmp64);
                xw::lds_add64(&L()->cnt.stale_rk, n_eq_rounds);     // (a counter the stages do not use otherwise)
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
                add(&pr[10], L()->cnt.stale_rk); add(&pr[11], L()->acc[kAccRedo]);
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
};

}  // namespace v2
}  // namespace nlzm
