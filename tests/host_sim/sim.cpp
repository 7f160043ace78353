// tests/host_sim/sim.cpp -- runs the product's master logic (nlzm_amd/csrc/nlzm_core.h)
// on the CPU with a 1-lane wave policy and checks it against the oracle.
//
// TEST HARNESS ONLY: nothing here is linked into libnlzm_hip.so.  The policy below is
// the only code that differs from the gfx950 build; every decision the kernel makes is
// the templated code in nlzm_core.h.
//
//   sim <file> <hist_bits> [check_tables]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <chrono>
#include <vector>

#define NLZM_HD inline
#define NLZM_HDN
#include "../../nlzm_amd/csrc/nlzm_core.h"
#include "../../oracle/nlzm_oracle.h"

using namespace nlzm;

struct SimWorkers;
static SimWorkers *g_workers = nullptr;
static void sim_wait_hook(uint32_t a);

struct HostIO {
    static void st_agent(uint32_t *p, uint32_t v) { *p = v; }
    static uint32_t ld_agent(const uint32_t *p) { return *p; }
    static uint32_t atomic_inc(uint32_t *p) { return (*p)++; }
    static void drain() {}
};

static MasterLds *g_lds = nullptr;
struct HostWave {
    static MasterLds *lds() { return g_lds; }
    static void cnt_add(unsigned long long *p, unsigned long long v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
    template <class F>
    static void cnt_add_fn(unsigned long long *p, uint32_t n, F f) { for (uint32_t i = 0; i < n; i++) cnt_add(p, f(i)); }
    static uint32_t pick(uint32_t v, uint32_t) { return v; }
    static void cdf_multi(uint16_t *cdf, uint16_t *price, const uint16_t *lut, const uint32_t *list, uint32_t n, uint32_t *out)
    {
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t ctx = list[2 * k], y = list[2 * k + 1];
            uint32_t start, freq;
            cdf_step(cdf + ctx * kCdfStride, price + ctx * 16, lut, ctx_nsyms(ctx), y, start, freq);
            out[k] = (freq << 16) + start;
        }
    }
    static void cdf_step(uint16_t *cell, uint16_t *price_row, const uint16_t *lut, uint32_t nsy, uint32_t y, uint32_t &start, uint32_t &freq)
    {
        start = cell[y]; freq = (uint32_t)cell[y + 1] - start;
        for (uint32_t i = 0; i < nsy; i++) {
            const int mix = (i <= y) ? (int)i : (int)(16384 + i + (127 - nsy));
            cell[i] = (uint16_t)(cell[i] + ((mix - (int)cell[i]) >> 7));
        }
        for (uint32_t i = 0; i < nsy; i++) price_row[i] = lut[((uint32_t)cell[i + 1] - (uint32_t)cell[i]) >> 6];
    }
    static void xw_store(uint32_t *p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
    static uint32_t xw_load(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
    static void xw_add(uint32_t *p, uint32_t v) { __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
    static void xw_pause() { std::this_thread::yield(); }
    static void st_agent(uint32_t *p, uint32_t v) { *p = v; }
    static uint32_t ld_agent(const uint32_t *p) { return *p; }
    static void sleep() {}
    static unsigned long long clock() { return 0; }
    static unsigned long long timeout_ticks() { return ~0ull >> 1; }
    static void wait_hook(void *, uint32_t a) { sim_wait_hook(a); }
    static unsigned long long tick() { return 0; }
    static void lds_min(uint32_t *p, uint32_t v) { if (v < *p) *p = v; }
    template <class F>
    static unsigned long long mask64(F f) { unsigned long long m = 0; for (uint32_t i = 0; i < 64; i++) if (f(i)) m |= 1ull << i; return m; }
    struct Rec { uint32_t w[32]; };
    static Rec rec_load(const uint32_t *base) { Rec r; for (int i = 0; i < 32; i++) r.w[i] = base[i]; return r; }
    static Rec rec_load_agent(const uint32_t *base) { Rec r; for (int i = 0; i < 16; i++) r.w[i] = base[i]; for (int i = 16; i < 32; i++) r.w[i] = 0; return r; }
    template <class F>
    static Rec rec_load_fn(F f)         // ascending, one load at a time (see rec_load_fn32)
    {
        Rec r{};
        for (uint32_t i = 0; i < 8; i++) { r.w[i] = f(i); __atomic_thread_fence(__ATOMIC_ACQUIRE); }
        return r;
    }
    template <class F>
    static Rec rec_load_fn32(F f)       // ascending lane order, one load at a time: a count in lane 0 is read before the data it covers
    {
        Rec r{};
        for (uint32_t i = 0; i < 32; i++) { r.w[i] = f(i); __atomic_thread_fence(__ATOMIC_ACQUIRE); }
        return r;
    }
    static uint32_t rec_get(const Rec &r, uint32_t k) { return r.w[k]; }
    static Rec rec_sel(bool c, const Rec &a, const Rec &b) { return c ? a : b; }
    static Rec rec_set(Rec r, uint32_t k, uint32_t v) { r.w[k] = v; return r; }
    static void join3(uint32_t &, uint32_t &, uint32_t &) {}
    static Rec rec_shl3(const Rec &r) { Rec o{}; for (int i = 0; i < 13; i++) o.w[i] = r.w[i + 3]; return o; }
    static Rec rec_shl8(const Rec &r) { Rec o{}; for (int i = 0; i < 8; i++) o.w[i] = r.w[i + 8]; return o; }
    static void rec_store_n(uint32_t *dst, const Rec &r, uint32_t n) { for (uint32_t i = 0; i < n; i++) dst[i] = r.w[i]; }
    static Rec rec_shift2(const Rec &r) { Rec o{}; for (int i = 2; i < 16; i++) o.w[i] = r.w[i - 2]; return o; }
    static void rec_store4(uint32_t *dst, const Rec &r) { for (int i = 1; i <= 4; i++) dst[i] = r.w[i]; }
    static uint32_t rec_eq4(const Rec &a, const Rec &b) { for (int i = 1; i <= 4; i++) if (a.w[i] != b.w[i]) return 0; return 1; }
    template <class F>
    static unsigned long long rec_mask(const Rec &r, F f) { unsigned long long m = 0; for (uint32_t i = 0; i < 32; i++) if (f(i, r.w[i])) m |= 1ull << i; return m; }
    template <class F>
    static uint32_t rec_sum_odd4(const Rec &r, F f) { return f(1, r.w[1]) + f(3, r.w[3]) + f(5, r.w[5]) + f(7, r.w[7]); }
    struct PfLane { uint32_t idx[64], rkslot[64], stale[64], v4[64], row1[64], sl[64], sd[64], cmpb[64], simple[64], wrote[64]; };
    static void pfl_set(PfLane &p, uint32_t j, uint32_t idx, uint32_t rkslot, uint32_t v4, uint32_t row1, uint32_t sl, uint32_t sd,
                        uint32_t cmpb, bool simple)
    {
        p.idx[j] = idx; p.rkslot[j] = rkslot; p.stale[j] = 0; p.v4[j] = v4; p.row1[j] = row1; p.sl[j] = sl; p.sd[j] = sd; p.cmpb[j] = cmpb;
        p.simple[j] = simple; p.wrote[j] = 0;
    }
    static void pfl_conflicts(PfLane &p, uint32_t n)
    {
        for (uint32_t j = 0; j < n; j++) {
            const uint32_t o2 = p.idx[j] & 0xFFFFu, o3 = p.idx[j] >> 16;
            for (uint32_t k = 0; k < j; k++) {
                if (p.idx[k] == 0xFFFFFFFFu) continue;
                const uint32_t i2 = p.idx[k] & 0xFFFFu, i3 = p.idx[k] >> 16;
                if (o2 == i2 || o3 == i3 || o3 == i3 + 1 || o3 + 1 == i3) p.stale[j] |= 1;
            }
        }
    }
    static unsigned long long pfl_run_mask(const PfLane &p, uint32_t n)
    {
        unsigned long long m = 0;
        for (uint32_t j = 0; j < n; j++) if (p.simple[j] && !(p.stale[j] & 5u)) m |= 1ull << j;
        return m;
    }
    static void pfl_run_store(PfLane &p, uint32_t s0, uint32_t cnt, uint32_t *ht2, uint32_t *ht3, uint32_t q0, uint32_t wbits,
                              uint32_t tag_mask, uint32_t ht3_shift)
    {
        for (uint32_t j = s0; j < s0 + cnt; j++) {
            const uint32_t q = q0 + (j - s0);
            const uint32_t h2 = hash4(p.v4[j] & 0xFFFFu), h3 = hash4(p.v4[j] & 0xFFFFFFu);
            const uint32_t i2 = h2 >> 20, i3 = h3 >> ht3_shift;
            ht2[i2] = q | ((h2 & tag_mask) << wbits);
            ht3[i3] = q | ((h3 & tag_mask) << wbits);
            ht3[i3 + 1] = p.row1[j];
            p.wrote[j] = 1;
        }
    }
    static void pfl_update(PfLane &p, uint32_t j, uint32_t sl, uint32_t sd, bool simple) { p.sl[j] = sl; p.sd[j] = sd; p.simple[j] = simple; }
    static void pfl_wrote(PfLane &p, uint32_t s, uint32_t v1) { p.wrote[s] = 1; p.row1[s] = v1; }
    static void pfl_rows_now(const PfLane &p, uint32_t j, uint32_t i2, uint32_t i3, uint32_t q0, uint32_t wbits, uint32_t tag_mask,
                             uint32_t, uint32_t row[3])
    {
        for (uint32_t k = 0; k < j; k++) {          // in order: the last writer of a row wins
            if (!p.wrote[k]) continue;
            const uint32_t o2 = p.idx[k] & 0xFFFFu, o3 = p.idx[k] >> 16, qk = q0 + k;
            const uint32_t e2 = qk | ((hash4(p.v4[k] & 0xFFFFu) & tag_mask) << wbits), e3 = qk | ((hash4(p.v4[k] & 0xFFFFFFu) & tag_mask) << wbits);
            if (o2 == i2) row[0] = e2;
            if (o3 == i3) { row[1] = e3; row[2] = p.row1[k]; }
            if (o3 + 1 == i3) row[1] = p.row1[k];
            if (o3 == i3 + 1) row[2] = e3;
        }
    }
    static uint32_t pfl_sl(const PfLane &p, uint32_t s) { return p.sl[s]; }
    static uint32_t pfl_sd(const PfLane &p, uint32_t s) { return p.sd[s]; }
    static uint32_t pfl_cmpb(const PfLane &p, uint32_t s) { return p.cmpb[s]; }
    static void pfl_mark_rk(PfLane &p, uint32_t s, uint32_t n, uint32_t slot) { for (uint32_t j = s + 1; j < n; j++) if (p.rkslot[j] == slot) p.stale[j] |= 4; }
    static uint32_t pfl_stale(const PfLane &p, uint32_t s) { return p.stale[s]; }
    struct RepPf { const uint8_t *in; unsigned long long n; uint32_t a, r[4]; };
    static RepPf rep_prefetch(const uint8_t *in, unsigned long long n, uint32_t a, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3)
    {
        return RepPf{ in, n, a, { r0, r1, r2, r3 } };
    }
    static void rep_lengths(const RepPf &r, uint32_t len[4])
    {
        for (int k = 0; k < 4; k++) {
            uint32_t l = 0;
            if (r.r[k] <= r.a) while (l < kRepPf && r.a + l < r.n + 64 && r.in[r.a - r.r[k] + l] == r.in[r.a + l]) l++;
            len[k] = l;
        }
    }
    static uint32_t uni(uint32_t v) { return v; }
    static uint32_t lane() { return 0; }
    static uint32_t width() { return 1; }
    static void sync() {}
    static void sync_global() {}
    static uint32_t rmin(uint32_t v) { return v; }
    static uint32_t ror(uint32_t v) { return v; }
    static void cmp_multi(const uint8_t *in, const uint32_t sp[8], uint32_t a, const uint32_t cap[8], uint32_t valid,
                          uint32_t len[8])
    {
        for (int k = 0; k < 8; k++) {
            len[k] = 0;
            if (!((valid >> k) & 1)) continue;
            uint32_t l = 0;
            while (l < cap[k] && in[sp[k] + l] == in[a + l]) l++;
            len[k] = l;
        }
    }
};

static void make_geom(uint64_t n, uint32_t hist_bits_req, Geom &g)
{
    auto clampu = [](uint32_t v, uint32_t lo, uint32_t hi) { return v < lo ? lo : (v > hi ? hi : v); };
    uint32_t hb = hist_bits_req;
    while (hb > 10 && n < (1ull << (hb - 1))) --hb;
    g.n = n; g.wbits = hb; g.wmask = (1u << hb) - 1;
    g.frame_bits = clampu(hb - 2, 14, 17); g.frame_size = 1u << g.frame_bits;
    g.chunk_size = ((g.frame_size * 15) / 16) - 0x200; g.feed = g.chunk_size + kMatchMax + 1;
    g.ht3_shift = 32 - (12 + clampu(hb, 15, 17) - 15);
    g.bt_shift = 32 - (13 + clampu(hb, 16, 20) - 16);
    g.rk_shift = 32 - (15 + clampu(hb, 16, 22) - 16);
    g.tag_mask = (uint32_t)((1ull << (32 - hb)) - 1);
    g.nchunks = (uint32_t)((n + g.chunk_size - 1) / g.chunk_size);
    {   // node slots >= W + positions of the larger launch (the simulation runs two)
        const unsigned long long need = (1ull << hb) + (unsigned long long)(g.nchunks - g.nchunks / 2 + 1) * g.chunk_size;
        unsigned long long slots = 1ull << hb;
        while (slots < need) slots <<= 1;
        g.bt_tmask = (uint32_t)(slots - 1);
    }
}

// ---- emulation of the device-side helpers around the master (tests only) -------------
// pre-filter: same tables and phases as prefilter_*_kernel in nlzm_kernels.hip
struct SimPrefilter {
    uint32_t t_bits, m_bits;
    std::vector<uint32_t> T, M;
    void init(uint32_t wbits, unsigned long long batch_pos)
    {
        uint32_t lg = 1; while ((1ull << lg) < batch_pos) lg++;
        t_bits = wbits + 5 > 32 ? 32 : (wbits + 5 < 16 ? 16 : wbits + 5);
        if (t_bits > 26) t_bits = 26;                    // keep the simulation small; only adds `unc` marks
        m_bits = lg + 6 > 28 ? 28 : lg + 6;
        if (m_bits > 24) m_bits = 24;
        T.assign((size_t)1 << t_bits, 0); M.assign((size_t)1 << m_bits, kNone);
    }
    void run(const uint8_t *in, unsigned long long n, uint32_t a0, uint32_t a1, uint32_t wmask, std::vector<uint8_t> &unc)
    {
        const uint32_t cnt = a1 - a0;
        std::vector<uint32_t> h(cnt, 0); std::vector<uint8_t> c1(cnt, 0);
        unc.assign(cnt + 16, 0);
        for (uint32_t a = a0; a < a1; a++) {
            if ((unsigned long long)a + 65 > n) continue;
            uint32_t hh = 0;
            for (int j = 0; j < 65; j++) hh = (hh + in[a + j]) * 0x2F0FD693u;
            h[a - a0] = hh;
            const uint32_t t = T[(hh * 0x9E3779B1u) >> (32 - t_bits)];
            c1[a - a0] = t != 0 && a - (t - 1) <= wmask;
            uint32_t &m = M[(hh * 0x85EBCA77u) >> (32 - m_bits)];
            if (a < m) m = a;
        }
        if (cnt) unc[0] = 1;
        for (uint32_t a = a0; a < a1; a++) {
            uint8_t f = c1[a - a0];
            if ((unsigned long long)a + 65 <= n) f |= M[(h[a - a0] * 0x85EBCA77u) >> (32 - m_bits)] < a;
            if (a + 1 < a1) unc[a + 1 - a0] = f;
        }
        for (uint32_t a = a0; a < a1; a++) {
            if ((unsigned long long)a + 65 > n) continue;
            uint32_t &t = T[(h[a - a0] * 0x9E3779B1u) >> (32 - t_bits)];
            if (a + 1 > t) t = a + 1;
            M[(h[a - a0] * 0x85EBCA77u) >> (32 - m_bits)] = kNone;
        }
    }
};

// worker lanes, run lazily: when the master needs the result of position a, the worker of a's
// head processes its positions up to a, exactly as worker_role does on the device
struct SimWorkers {
    Geom g; Globals *G;
    std::vector<std::vector<uint32_t>> bins;   // per head: positions of this launch, ascending
    std::vector<uint32_t> next;                // per head: next index
    std::vector<uint8_t> published;            // per position of the launch
    uint32_t c0 = 0;
    unsigned long long calls = 0, tests = 0, cmp = 0, dry = 0;
    uint32_t la_end_of(uint32_t a) const
    {
        const uint32_t ci = a / g.chunk_size;
        const unsigned long long chunk_abs = (unsigned long long)ci * g.chunk_size, remain = g.n - chunk_abs;
        return (uint32_t)(chunk_abs + (remain < g.feed ? remain : g.feed));
    }
    void build(uint32_t a0, uint32_t a1)
    {
        const uint32_t nheads = 1u << (32 - g.bt_shift);
        bins.assign(nheads, {}); next.assign(nheads, 0); published.assign(a1 - a0 + 1, 0);
        for (uint32_t a = a0; a < a1; a++) {
            if (la_end_of(a) - a < 4) continue;
            bins[hash4(load32u(G->in + a)) >> g.bt_shift].push_back(a);
        }
    }
    void step(uint32_t h, bool final_flush)
    {
        // process head h's next position; requires its flag to be known when uncertain
        const uint32_t a = bins[h][next[h]];
        const uint32_t max_len = umin(la_end_of(a) - a, kMatchMax);
        unsigned long long dt = 0, dc = 0;
        if (G->unc[a - G->batch_a0]) {
            if (!published[a - G->batch_a0]) {
                worker_bt_call<HostIO, false>(g, *G, a, max_len, true, dt, dc);
                published[a - G->batch_a0] = 1; dry++;
            }
            const uint32_t f = G->bt_flag[a - G->batch_a0];
            if (f == 0) { if (final_flush) { printf("sim: flag of uncertain position %u never published\n", a); exit(1); } return; }
            if (f == kFlagCall) { worker_bt_call<HostIO, true>(g, *G, a, max_len, false, tests, cmp); calls++; }
        } else {
            worker_bt_call<HostIO, true>(g, *G, a, max_len, true, tests, cmp);
            published[a - G->batch_a0] = 1; calls++;
        }
        next[h]++;
    }
    bool eager_mode = false;
    void need(uint32_t a)
    {
        const uint32_t h = hash4(load32u(G->in + a)) >> g.bt_shift;
        if (eager_mode) {   // after a decision arrives the lane runs on as far as it can
            while (next[h] < bins[h].size()) { const uint32_t b = next[h]; step(h, false); if (next[h] == b) break; }
            return;
        }
        while (next[h] < bins[h].size() && bins[h][next[h]] <= a) {
            const uint32_t before = next[h];
            step(h, false);
            if (next[h] == before) break;       // blocked on its own flag: only legal for a itself
        }
        // the dry run of `a` itself may have just been published while its flag is already in
    }
    // every lane infinitely fast: run each head until it blocks on a decision of the master
    void eager()
    {
        for (uint32_t h = 0; h < bins.size(); h++)
            while (next[h] < bins[h].size()) { const uint32_t b = next[h]; step(h, false); if (next[h] == b) break; }
    }
    void need_all(uint32_t) {}
    void finish()
    {
        for (uint32_t h = 0; h < bins.size(); h++) while (next[h] < bins[h].size()) step(h, true);
    }
};
static void sim_wait_hook(uint32_t a) { if (g_workers) g_workers->need(a); }

struct Check {
    const std::vector<uint32_t> *syms; const std::vector<uint8_t> *bits; const std::vector<FrameMeta> *fm;
    unsigned long long syms_stride, bits_stride;
    int bad = 0;
    // tables
    const uint32_t *cap; unsigned long long cap_used; unsigned long long cap_pos = 0; int check_tables = 0;
};

static void on_frame(void *u, uint32_t idx, uint32_t num_ops, const uint32_t *syms, uint32_t nsyms, const uint8_t *bits,
                     uint32_t nbits, const uint8_t *, uint32_t)
{
    Check *c = (Check *)u;
    if (c->bad) return;
    const FrameMeta &m = (*c->fm)[idx];
    const uint32_t *s = c->syms->data() + idx * c->syms_stride;
    const uint8_t *b = c->bits->data() + idx * c->bits_stride;
    uint32_t n = m.nsyms < nsyms ? m.nsyms : nsyms;
    for (uint32_t i = 0; i < n; i++) if (s[i] != syms[i]) { printf("frame %u: symbol %u differs (sim %08x oracle %08x)\n", idx, i, s[i], syms[i]); c->bad = 1; return; }
    if (m.nsyms != nsyms || m.nbits_bytes != nbits || m.num_ops != num_ops) {
        printf("frame %u: sizes differ sim(%u,%u,%u) oracle(%u,%u,%u)\n", idx, m.nsyms, m.nbits_bytes, m.num_ops, nsyms, nbits, num_ops);
        c->bad = 1; return;
    }
    if (memcmp(b, bits, nbits)) { printf("frame %u: bit bytes differ\n", idx); c->bad = 1; }
}

static void on_pos(void *u, uint64_t abs_pos, uint32_t max_len, const uint32_t *delta)
{
    Check *c = (Check *)u;
    if (!c->check_tables || c->bad) return;
    const uint32_t *r = c->cap + c->cap_pos;
    if (c->cap_pos >= c->cap_used) { printf("pos %llu: sim capture ended\n", (unsigned long long)abs_pos); c->bad = 1; return; }
    if (r[0] != (uint32_t)abs_pos || r[1] != max_len) {
        printf("pos %llu: sim (pos %u max_len %u) oracle max_len %u\n", (unsigned long long)abs_pos, r[0], r[1], max_len);
        c->bad = 1; return;
    }
    for (uint32_t i = 2; i <= max_len; i++) if (r[i] != delta[i]) {
        printf("pos %llu: delta[%u] sim %u oracle %u (max_len %u)\n", (unsigned long long)abs_pos, i, r[i], delta[i], max_len);
        c->bad = 1; return;
    }
    c->cap_pos += 2 + (max_len >= 2 ? max_len - 1 : 0);
}

int main(int argc, char **argv)
{
    if (argc < 3) { printf("usage: sim <file> <hist_bits> [check_tables]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { printf("cannot open %s\n", argv[1]); return 2; }
    fseek(f, 0, SEEK_END); long long n = ftello(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> in((size_t)n + 64, 0);
    if (n && fread(in.data(), 1, (size_t)n, f) != (size_t)n) return 2;
    fclose(f);
    const uint32_t hb = (uint32_t)atoi(argv[2]);
    const int check_tables = argc > 3 ? atoi(argv[3]) : 0;
    const int use_workers = argc > 4 ? atoi(argv[4]) : 0;

    Geom g; make_geom((uint64_t)n, hb, g);
    std::vector<uint32_t> rkhash((size_t)n + 1, 0);
    for (long long a = 0; a + 256 <= n; a++) {
        if (a == 0) rkhash[0] = nlzm_oracle_rk_hash256(in.data());
        else rkhash[a] = (in[a + 255] + rkhash[a - 1] - in[a - 1] * 0x0E4EA401u) * 0x2F0FD693u;
    }
    std::vector<uint32_t> ht2(4096, kNone), ht3((size_t)2 << (32 - g.ht3_shift), kNone), rkt((size_t)1 << (32 - g.rk_shift), kNone),
        heads((size_t)1 << (32 - g.bt_shift), kNone), tree(((size_t)g.bt_tmask + 1) * 2, kNone);
    Persist P; memset(&P, 0, sizeof P);
    for (uint32_t ctx = 0; ctx < kNumCtx; ctx++) {
        const uint32_t ns = ctx_nsyms(ctx);
        for (uint32_t i = 0; i <= ns; i++) P.cdf[ctx * kCdfStride + i] = (uint16_t)(i * (16384 / ns));
    }
    for (int i = 0; i < 4; i++) P.rep[i] = i + 1;
    const unsigned long long ss = 3ull * g.chunk_size + 4096, bs = 2ull * g.chunk_size + 64;
    std::vector<uint32_t> syms((size_t)(g.nchunks ? g.nchunks : 1) * ss);
    std::vector<uint8_t> bits((size_t)(g.nchunks ? g.nchunks : 1) * bs);
    std::vector<FrameMeta> fm(g.nchunks ? g.nchunks : 1);
    std::vector<uint32_t> cap; unsigned long long cap_used = 0;
    if (check_tables) cap.resize((size_t)n * 270 + 1024);

    g_lds = new MasterLds;
    Master<HostWave> m;
    m.g = g;
    memset(&m.G, 0, sizeof m.G);
    m.G.in = in.data(); m.G.rkhash = rkhash.data(); m.G.ht2 = ht2.data(); m.G.ht3 = ht3.data(); m.G.rk_table = rkt.data();
    m.G.bt_heads = heads.data(); m.G.bt_tree = tree.data(); m.G.persist = &P;
    m.G.syms = syms.data(); m.G.syms_stride = ss; m.G.bits = bits.data(); m.G.bits_stride = bs; m.G.fmeta = fm.data(); m.G.chunk0 = 0;
    if (check_tables) { m.G.cap_words = cap.data(); m.G.cap_cap = cap.size(); m.G.cap_lo = 0; m.G.cap_hi = ~0ull; m.G.cap_used = &cap_used; }
    // two launches, to exercise the state save/restore path
    const uint32_t half = g.nchunks / 2;
    SimPrefilter pf; SimWorkers wk; std::vector<uint8_t> unc; std::vector<uint32_t> ready, pairs, flag;
    uint32_t abort_word = 0; WorkerCounters wc = {};
    unsigned long long unc_total = 0;
    if (use_workers) { pf.init(g.wbits, (unsigned long long)(g.nchunks - half + 1) * g.chunk_size); wk.g = g; wk.G = &m.G; g_workers = &wk; }
    const uint32_t ranges[3] = { 0, half, g.nchunks };
    for (int r = 0; r < 2; r++) {
        const uint32_t c0 = ranges[r], c1 = ranges[r + 1];
        if (c0 == c1) continue;
        m.G.chunk0 = 0;     // frame buffers are indexed from chunk 0 in the simulation
        if (use_workers) {
            const unsigned long long a0 = (unsigned long long)c0 * g.chunk_size;
            unsigned long long a1 = (unsigned long long)c1 * g.chunk_size; if (a1 > (unsigned long long)n) a1 = n;
            pf.run(in.data(), (unsigned long long)n, (uint32_t)a0, (uint32_t)a1, g.wmask, unc);
            for (unsigned long long i = 0; i < a1 - a0; i++) unc_total += unc[i];
            ready.assign((a1 - a0 + 1) * (size_t)kBtRec, 0); pairs.resize((size_t)(a1 - a0 + 1) * 2 * kBtMaxPairs); flag.assign(a1 - a0 + 1, 0);
            m.G.workers = 1; m.G.batch_a0 = (uint32_t)a0; m.G.bt_ready = ready.data(); m.G.bt_pairs = pairs.data(); m.G.bt_flag = flag.data(); m.G.unc = unc.data();
            m.G.nheads = 1u << (32 - g.bt_shift);
            m.G.abort_word = &abort_word; m.G.wcnt = &wc;
            wk.build((uint32_t)a0, (uint32_t)a1);
            wk.eager_mode = use_workers == 2;
            if (use_workers == 2) wk.eager();
        }
        // the roles of the serial half, one host thread each
        Master<HostWave>::init_shared(m.G, (uint32_t)((unsigned long long)c0 * g.chunk_size));
        Master<HostWave> mb = m, mt = m, ms = m, ms2 = m, ms3 = m, ms4 = m;
        const uint32_t a_first = (uint32_t)((unsigned long long)c0 * g.chunk_size);
        std::thread twatch;
        if (getenv("NLZM_SIM_WATCH")) twatch = std::thread([&] {
            std::this_thread::sleep_for(std::chrono::seconds(atoi(getenv("NLZM_SIM_WATCH"))));
            MasterLds *L = g_lds;
            fprintf(stderr, "watch: sdone %u apos %u bpos %u post0 %u post1 %u cnt %u %u early %u %u\n", L->x_sdone, L->x_apos, L->x_bpos,
                    L->post[0][0], L->post[1][0], L->post[0][21], L->post[0][22], L->eb[15], L->eb[31]);
            for (int b = 0; b < 2; b++) { fprintf(stderr, "post[%d]:", b); for (int i = 0; i < 32; i++) fprintf(stderr, " %u", L->post[b][i]); fprintf(stderr, "\n"); }
            fprintf(stderr, "ea_tag:"); for (int i = 0; i < 8; i++) fprintf(stderr, " %u", L->ea_tag[i]); fprintf(stderr, "\n");
            _Exit(3);
        });
        if (twatch.joinable()) twatch.detach();
        std::thread ts([&] { ms.run_edge_list(a_first); });
        std::thread ts2([&] { ms2.run_rep_list(a_first, 0); });
        std::thread ts4([&] { ms4.run_rep_list(a_first, 1); });
        std::thread ts3([&] { ms3.run_edge_apply(a_first); });
        std::thread tb([&] { mb.run_parser(c0, c1); });
        std::thread tt([&] { mt.run_table(c0, c1); });
        m.run_finder(c0, c1);
        tt.join();
        tb.join();
        ts.join(); ts2.join(); ts3.join(); ts4.join();
        if (use_workers) wk.finish();
#ifdef NLZM_SIM_COUNT
        fprintf(stderr, "dbg: long rep compares %llu (of %llu checks), relaxed rep probes %llu (of %llu nodes)\n", g_dbg[0], g_dbg[3], g_dbg[1], g_dbg[2]);
        fprintf(stderr, "dbg: %llu positions in %llu runs; rep-set guesses missed %llu of %llu nodes\n", g_dbg[4], g_dbg[5], g_dbg[6], g_dbg[7]);
#endif
        if (P.error) { printf("sim error %u (info %u)\n", P.error, P.error_info[0]); return 1; }
    }
    if (use_workers) {
        P.cnt.bt_tests += wk.tests; P.cnt.bt_calls += wk.calls; P.cnt.cmp_bytes += wk.cmp;
        printf("workers: uncertain marks %llu (%.2f%%), master-seen uncertain %llu, dry runs %llu\n", unc_total,
               100.0 * unc_total / (n ? n : 1), P.cnt.uncertain_positions, wk.dry);
    }

    Check c; c.syms = &syms; c.bits = &bits; c.fm = &fm; c.syms_stride = ss; c.bits_stride = bs;
    c.cap = cap.data(); c.cap_used = cap_used; c.check_tables = check_tables;
    nlzm_oracle_hooks hk; memset(&hk, 0, sizeof hk);
    hk.user = &c; hk.on_frame = on_frame; hk.on_position = on_pos;
    std::vector<uint8_t> out(nlzm_oracle_bound((uint64_t)n));
    uint64_t out_n = 0; nlzm_oracle_stats st;
    int rc = nlzm_oracle_compress(in.data(), (uint64_t)n, hb, out.data(), out.size(), &out_n, &st, &hk);
    if (rc) { printf("oracle failed %d\n", rc); return 1; }
    printf("direct-path slots per 1000 positions: HT %.1f RK %.1f BT %.1f\n", 1e3 * P.cnt.stale_ht / (double)(n ? n : 1),
           1e3 * P.cnt.stale_rk / (double)(n ? n : 1), 1e3 * P.cnt.bt_slow / (double)(n ? n : 1));
    printf("%s: %s  (chunks %u, positions sim %llu oracle %llu, bt_tests sim %llu oracle %llu, cmp_bytes sim %llu oracle %llu)\n",
           argv[1], c.bad ? "MISMATCH" : "OK", g.nchunks, P.cnt.positions, (unsigned long long)st.positions, P.cnt.bt_tests,
           (unsigned long long)st.bt_tests, P.cnt.cmp_bytes + P.prof[46], (unsigned long long)st.cmp_bytes);
    return c.bad;
}
