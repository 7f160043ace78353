// tests/host_sim/sim2.cpp -- runs the product's pipeline stages (nlzm_amd/csrc/nlzm_v2.h: finder, table, parser) on the
// CPU, every GPU lane a fiber (xw_sim.cpp), and checks them against the oracle:
//   * the match table of every position (front expanded to delta[2..max_len]) as it is produced,
//   * the symbol / bit streams of every frame.
// TEST HARNESS ONLY: nothing here is linked into libnlzm_hip.so.  The BT4 worker lanes are emulated lazily (a head's
// worker runs when the finder stage asks for one of its positions), exactly as worker_role does on the device.
//
//   sim2 <file> <hist_bits> [workers: 1 lazy | 2 eager] [launches]
#define NLZM_SIM 1
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define NLZM_HD inline
#define NLZM_HDN
#include "../../nlzm_amd/csrc/nlzm_v2.h"
#include "../../oracle/nlzm_oracle.h"

using namespace nlzm;

struct HostIO {
    static void st_agent(uint32_t *p, uint32_t v) { *p = v; }
    static uint32_t ld_agent(const uint32_t *p) { return *p; }
    static void drain() {}
    static void st_quad(uint32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
    static uint32_t atomic_inc(uint32_t *p) { return (*p)++; }
};

static void make_geom(uint64_t n, uint32_t hist_bits_req, uint32_t nlaunch, Geom &g)
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
    {
        const unsigned long long need = (1ull << hb) + (unsigned long long)(g.nchunks / nlaunch + 2) * g.chunk_size;
        unsigned long long slots = 1ull << hb;
        while (slots < need) slots <<= 1;
        g.bt_tmask = (uint32_t)(slots - 1);
    }
}

// pre-filter: same tables and phases as prefilter_*_kernel in nlzm_kernels.hip
struct SimPrefilter {
    uint32_t t_bits, m_bits;
    std::vector<uint32_t> T, M;
    void init(uint32_t wbits, unsigned long long batch_pos)
    {
        uint32_t lg = 1; while ((1ull << lg) < batch_pos) lg++;
        t_bits = wbits + 5 > 32 ? 32 : (wbits + 5 < 16 ? 16 : wbits + 5);
        if (t_bits > 26) t_bits = 26;
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

// worker lanes, run lazily: when the finder stage needs the result of position a, the worker of a's head processes its
// positions up to a, exactly as worker_role does on the device
struct SimWorkers {
    Geom g; Globals *G;
    std::vector<std::vector<uint32_t>> bins;
    std::vector<uint32_t> next;
    std::vector<uint8_t> published;
    unsigned long long calls = 0, tests = 0, cmp = 0, dry = 0;
    bool eager_mode = false;
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
        const uint32_t a = bins[h][next[h]];
        const uint32_t max_len = umin(la_end_of(a) - a, kMatchMax);
        unsigned long long dt = 0, dc = 0;
        if (G->unc[a - G->batch_a0]) {
            // (as the device's worker lanes do: a marked position between two marked ones is assumed to be skipped and NOTHING is made of it until the
            //  finder stage says "call" -- which it says without a result, as soon as the positions in front are settled; any other marked position
            //  is assumed to be called: its result goes out at once, its stores wait for the decision)
            const unsigned long long bi = a - G->batch_a0;
            const bool park = bi > 0 && G->unc[bi - 1] && G->unc[bi + 1];
            if (!park && !published[bi]) {
                worker_bt_call<HostIO, false>(bt_view(g, *G), a, max_len, true, dt, dc);
                published[bi] = 1; dry++;
            }
            const uint32_t f = G->bt_flag[bi];
            if (f == 0) { if (final_flush) { printf("sim: flag of uncertain position %u never published\n", a); exit(1); } return; }
            if (f == kFlagCall) { worker_bt_call<HostIO, true>(bt_view(g, *G), a, max_len, park, tests, cmp); published[bi] = 1; calls++; }
        } else {
            worker_bt_call<HostIO, true>(bt_view(g, *G), a, max_len, true, tests, cmp);
            published[a - G->batch_a0] = 1; calls++;
        }
        next[h]++;
    }
    void need(uint32_t a)
    {
        const uint32_t h = hash4(load32u(G->in + a)) >> g.bt_shift;
        if (eager_mode) {
            while (next[h] < bins[h].size()) { const uint32_t b = next[h]; step(h, false); if (next[h] == b) break; }
            return;
        }
        while (next[h] < bins[h].size() && bins[h][next[h]] <= a) {
            const uint32_t before = next[h];
            step(h, false);
            if (next[h] == before) break;
        }
    }
    void eager()
    {
        for (uint32_t h = 0; h < bins.size(); h++)
            while (next[h] < bins[h].size()) { const uint32_t b = next[h]; step(h, false); if (next[h] == b) break; }
    }
    void finish()
    {
        for (uint32_t h = 0; h < bins.size(); h++) while (next[h] < bins[h].size()) step(h, true);
    }
};
static SimWorkers *g_workers = nullptr;
void xw::need_bt(void *, uint32_t a) { if (g_workers) g_workers->need(a); }
// stage traces (NLZM_SIM_TRACE=1: blocks of the finder and table stage; NLZM_SIM_TRACE_SEG=1: the segment a nice region is in)
void xw::trace(int what, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f, uint32_t g)
{
    static const bool all = getenv("NLZM_SIM_TRACE") != nullptr, seg = getenv("NLZM_SIM_TRACE_SEG") != nullptr;
    // (NLZM_SIM_TRACE_LO / _HI: only the blocks that start in [lo, hi) -- a large input with one place of interest)
    static const unsigned long long lo = getenv("NLZM_SIM_TRACE_LO") ? strtoull(getenv("NLZM_SIM_TRACE_LO"), nullptr, 10) : 0ull,
                                    hi = getenv("NLZM_SIM_TRACE_HI") ? strtoull(getenv("NLZM_SIM_TRACE_HI"), nullptr, 10) : ~0ull;
    if (xw::lane() != 0 || a < lo || a >= hi) return;
    if (what == 1 && seg) fprintf(stderr, "F region at %u: segment %u (cover %u)\n", a, b, c);
    if (what == 2 && all) fprintf(stderr, "F block a0 %u n %u m %u reach %u slider %u d %u end %u\n", a, b, c, d, e, f, g);
    if (what == 3 && all) fprintf(stderr, "T block a %u n %u\n", a, b);
}
// NLZM_SIM_RANDOM_BLOCKS=k: the parser's blocks are cut to 1..k nodes at random
uint32_t xw::test_cut(uint32_t nb)
{
    static const int rnd = getenv("NLZM_SIM_RANDOM_BLOCKS") ? atoi(getenv("NLZM_SIM_RANDOM_BLOCKS")) : 0;
    static uint32_t lcg = 12345;
    if (!rnd) return nb;
    lcg = lcg * 1664525u + 1013904223u;
    const uint32_t cut = 1 + (lcg >> 16) % (uint32_t)rnd;
    return cut < nb ? cut : nb;
}

// ---- the oracle's match tables, kept to check the table stage position by position ------------------------------
struct RefTables {
    std::vector<unsigned long long> off;    // per position: offset into words
    std::vector<uint32_t> words;            // max_len, delta[2..max_len]
    unsigned long long checked = 0;
    int bad = 0;
};
static RefTables g_ref;
// (NLZM_SIM_CHECK_LO / _HI: the tables of the positions in [lo, hi) only -- a large input: the oracle's tables of every position would not fit)
static unsigned long long g_check_lo = 0, g_check_hi = ~0ull;
static void ref_on_pos(void *, uint64_t abs_pos, uint32_t max_len, const uint32_t *delta)
{
    if (g_ref.off.size() != abs_pos) { printf("oracle positions out of order\n"); exit(2); }
    g_ref.off.push_back(g_ref.words.size());
    if (abs_pos < g_check_lo || abs_pos >= g_check_hi) return;
    g_ref.words.push_back(max_len);
    for (uint32_t i = 2; i <= max_len; i++) g_ref.words.push_back(delta[i]);
}
void v2::Table::sim_on_front(void *, uint32_t a, const unsigned long long *f, uint32_t fn)
{
    if (g_ref.bad) return;
    if (a < g_check_lo || a >= g_check_hi) return;
    if (a >= g_ref.off.size()) { printf("table stage: position %u beyond the oracle's\n", a); g_ref.bad = 1; return; }
    const uint32_t *r = g_ref.words.data() + g_ref.off[a];
    const uint32_t max_len = r[0];
    const uint32_t mt = fn ? v2::fr_end(f[0]) - a : 0u;
    bool ok = mt == max_len;
    for (uint32_t l = 2; ok && l <= max_len; l++) {
        uint32_t d = 0;
        for (uint32_t k = 0; k < fn; k++) if (v2::fr_end(f[k]) >= a + l) d = v2::fr_dist(f[k]);
        if (d != r[l - 1]) ok = false;
    }
    g_ref.checked++;
    if (!ok) {
        printf("position %u: table differs. oracle max_len %u:", a, max_len);
        for (uint32_t l = 2; l <= max_len && l < 40; l++) printf(" %u", r[l - 1]);
        printf("\n  sim front (%u):", fn);
        for (uint32_t k = 0; k < fn && k < 40; k++) printf(" (len %u, d %u)", v2::fr_end(f[k]) - a, v2::fr_dist(f[k]));
        printf("\n");
        g_ref.bad = 1;
    }
}

static uint32_t g_stop_chunk = 0xFFFFFFFFu;     // NLZM_SIM_MAX_LAUNCH: only the launches before it were simulated
struct Check {
    const std::vector<uint32_t> *syms; const std::vector<uint8_t> *bits; const std::vector<FrameMeta> *fm;
    unsigned long long syms_stride, bits_stride;
    int bad = 0;
};
static void on_frame(void *u, uint32_t idx, uint32_t num_ops, const uint32_t *syms, uint32_t nsyms, const uint8_t *bits,
                     uint32_t nbits, const uint8_t *, uint32_t)
{
    Check *c = (Check *)u;
    if (c->bad || idx >= g_stop_chunk) return;
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

// ---- the launch: block 0 finder, block 1 table, block 2 parser (one wave each) -----------------------------------
struct LaunchArgs { Geom g; Globals G; v2::GlobalsV2 V; uint32_t c0, c1; };
static void role_entry(void *p)
{
    const LaunchArgs &A = *(const LaunchArgs *)p;
    const uint32_t b = xw::block_index();
    if (b == 0) { if (xw::wave() == 0) { v2::Finder r; r.g = A.g; r.G = A.G; r.V = A.V; r.run(A.c0, A.c1); } }
    else if (b == 1) { if (xw::wave() < v2::table_waves(((const v2::StateV2 *)A.V.state)->tb_wide[A.G.launch_par & 1u])) { v2::Table r; r.g = A.g; r.G = A.G; r.V = A.V; r.run(A.c0, A.c1); } }
    else if (b == 2) { if (xw::wave() < v2::kPW) { v2::Parser r; r.g = A.g; r.G = A.G; r.V = A.V; r.run(A.c0, A.c1); } }
    else if (A.V.hb && xw::wave() < v2::kPW) { v2::Parser r; r.g = A.g; r.G = A.G; r.V = A.V; r.run_helper(A.c0, b - 3); }
}

int main(int argc, char **argv)
{
    if (argc < 3) { printf("usage: sim2 <file> <hist_bits> [workers 1|2] [launches]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { printf("cannot open %s\n", argv[1]); return 2; }
    fseek(f, 0, SEEK_END); long long n = ftello(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> in((size_t)n + 1024, 0);
    if (n && fread(in.data(), 1, (size_t)n, f) != (size_t)n) return 2;
    fclose(f);
    const uint32_t hb = (uint32_t)atoi(argv[2]);
    if (getenv("NLZM_SIM_CHECK_LO")) g_check_lo = strtoull(getenv("NLZM_SIM_CHECK_LO"), nullptr, 10);
    if (getenv("NLZM_SIM_CHECK_HI")) g_check_hi = strtoull(getenv("NLZM_SIM_CHECK_HI"), nullptr, 10);
    const int use_workers = argc > 3 ? atoi(argv[3]) : 1;
    const uint32_t nlaunch = argc > 4 ? (uint32_t)atoi(argv[4]) : 2;

    // the oracle first: its tables are the reference the table stage is checked against as it goes
    nlzm_oracle_hooks hk; memset(&hk, 0, sizeof hk);
    hk.on_position = ref_on_pos;
    std::vector<uint8_t> out(nlzm_oracle_bound((uint64_t)n));
    uint64_t out_n = 0; nlzm_oracle_stats st;
    if (nlzm_oracle_compress(in.data(), (uint64_t)n, hb, out.data(), out.size(), &out_n, &st, &hk)) { printf("oracle failed\n"); return 1; }

    Geom g; make_geom((uint64_t)n, hb, nlaunch, g);
    std::vector<uint32_t> rkhash((size_t)n + 1024, 0);
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

    LaunchArgs A; memset(&A, 0, sizeof A);
    A.g = g;
    Globals &G = A.G;
    G.in = in.data(); G.rkhash = rkhash.data(); G.ht2 = ht2.data(); G.ht3 = ht3.data(); G.rk_table = rkt.data();
    G.bt_heads = heads.data(); G.bt_tree = tree.data(); G.persist = &P;
    G.syms = syms.data(); G.syms_stride = ss; G.bits = bits.data(); G.bits_stride = bs; G.fmeta = fm.data(); G.chunk0 = 0;
    std::vector<uint32_t> ft((size_t)v2::kFtRing * v2::kFtStride), tp((size_t)v2::kTpRing * v2::kTpStride), tf((size_t)v2::kTpRing * v2::kTfStride);
    // NLZM_SIM_POISON=seed: everything the kernel does not initialise itself holds garbage, as on the device
    const uint32_t poison = getenv("NLZM_SIM_POISON") ? (uint32_t)atoi(getenv("NLZM_SIM_POISON")) : 0;
    uint32_t plcg = poison * 2654435761u + 1;
    auto junk = [&](std::vector<uint32_t> &v) { if (poison) for (auto &x : v) { plcg = plcg * 1664525u + 1013904223u; x = plcg; } };
    junk(ft); junk(tp); junk(tf);
    v2::Hx *hx = (v2::Hx *)aligned_alloc(128, sizeof(v2::Hx));
    v2::StateV2 S; memset(&S, 0, sizeof S);
    A.V.ft = ft.data(); A.V.tp = tp.data(); A.V.tf = tf.data(); A.V.hx = hx; A.V.state = (uint32_t *)&S;
    // the helper parser (NLZM_SIM_HELPER=0: without it)
    std::vector<uint32_t> hbmem(v2::kHelpers * sizeof(v2::HelpBox) / 4 + 4);
    junk(hbmem);
    const bool with_helper = !(getenv("NLZM_SIM_HELPER") && atoi(getenv("NLZM_SIM_HELPER")) == 0);
    A.V.hb = with_helper ? (v2::HelpBox *)(((uintptr_t)hbmem.data() + 15) & ~(uintptr_t)15) : nullptr;

    SimPrefilter pf; SimWorkers wk; std::vector<uint8_t> unc; std::vector<uint32_t> ready, pairs, flag;
    uint32_t abort_word = 0; WorkerCounters wc = {};
    unsigned long long unc_total = 0;
    pf.init(g.wbits, (unsigned long long)(g.nchunks / nlaunch + 2) * g.chunk_size); wk.g = g; wk.G = &G; g_workers = &wk;
    unsigned long long lds_bytes[3 + v2::kHelpers] = { sizeof(v2::FLds), sizeof(v2::TLds) };
    for (uint32_t b = 2; b < 3 + v2::kHelpers; b++) lds_bytes[b] = sizeof(v2::PLds);
    printf("LDS: finder %zu, table %zu, parser %zu bytes\n", sizeof(v2::FLds), sizeof(v2::TLds), sizeof(v2::PLds));
    for (uint32_t r = 0; r < nlaunch; r++) {
        const uint32_t c0 = (uint32_t)((unsigned long long)g.nchunks * r / nlaunch), c1 = (uint32_t)((unsigned long long)g.nchunks * (r + 1) / nlaunch);
        if (c0 == c1) continue;
        const unsigned long long a0 = (unsigned long long)c0 * g.chunk_size;
        unsigned long long a1 = (unsigned long long)c1 * g.chunk_size; if (a1 > (unsigned long long)n) a1 = n;
        pf.run(in.data(), (unsigned long long)n, (uint32_t)a0, (uint32_t)a1, g.wmask, unc);
        for (unsigned long long i = 0; i < a1 - a0; i++) unc_total += unc[i];
        // (NLZM_SIM_PSTRIDE=k: k pairs reserved per position, the rest in extension blocks -- the block sets' layout)
        const uint32_t pstride = getenv("NLZM_SIM_PSTRIDE") ? (uint32_t)atoi(getenv("NLZM_SIM_PSTRIDE")) : kBtMaxPairs;
        static std::vector<uint32_t> ext;
        const uint32_t ext_cap = pstride < kBtMaxPairs ? (uint32_t)((a1 - a0) + 64) : 0u;
        ext.resize((size_t)ext_cap * 2 * (kBtMaxPairs - pstride) + 2); junk(ext);
        G.bt_pstride = pstride; G.bt_ext = ext.data(); G.bt_ext_cap = ext_cap; G.bt_ext_cur = &hx->ext_cur;
        ready.assign((a1 - a0 + 1) * (size_t)kBtRec, 0); pairs.resize((size_t)(a1 - a0 + 1) * 2 * pstride); junk(pairs); flag.assign(a1 - a0 + 1, 0);
        G.workers = 1; G.batch_a0 = (uint32_t)a0; G.bt_ready = ready.data(); G.bt_pairs = pairs.data(); G.bt_flag = flag.data(); G.unc = unc.data();
        G.nheads = 1u << (32 - g.bt_shift);
        G.abort_word = &abort_word; G.wcnt = &wc;
        G.launch_par = r & 1u; G.table_shape = getenv("NLZM_SIM_TABLE_SHAPE") ? (uint32_t)atoi(getenv("NLZM_SIM_TABLE_SHAPE")) : 0u;
        wk.build((uint32_t)a0, (uint32_t)a1);
        wk.eager_mode = use_workers == 2;
        memset(hx, 0, sizeof *hx);                      // (before the eager worker lanes: the extension blocks' cursor is one of its words)
        if (use_workers == 2) wk.eager();
        hx->f_pos = hx->t_pos = hx->t_out = hx->p_pos = (uint32_t)a0;
        hx->p_seg = ((unsigned long long)(uint32_t)a0 << 32) | (uint32_t)a0;
        A.c0 = c0; A.c1 = c1;
        xw::launch(3 + v2::kHelpers, v2::kParserThreads > 64 * v2::kTWMax ? v2::kParserThreads : 64 * v2::kTWMax, lds_bytes, role_entry, &A);
        wk.finish();
        {   // how well do the marks of the neighbours predict the finder stage's decision at a marked position?  (NLZM_SIM_RULES=1)
            static unsigned long long tab[2][2][2] = {};   // [prev marked][next marked][decision skip]
            const unsigned long long cnt = a1 - a0;
            for (unsigned long long i = 0; i < cnt; i++) {
                if (!unc[i] || !flag[i]) continue;
                tab[i > 0 && unc[i - 1]][i + 1 < cnt && unc[i + 1]][flag[i] == kFlagSkip]++;
            }
            if (getenv("NLZM_SIM_RULES") && r + 1 == nlaunch)
                for (int p = 0; p < 2; p++) for (int q = 0; q < 2; q++)
                    printf("marked positions with prev %s, next %s: decided call %llu, skip %llu\n", p ? "marked" : "unmarked", q ? "marked" : "unmarked", tab[p][q][0], tab[p][q][1]);
        }
        if (P.error || hx->err) { printf("sim error %u / %u (info %u %u)\n", P.error, hx->err, P.error_info[0], P.error_info[1]); return 1; }
        if (g_ref.bad) break;
        if (getenv("NLZM_SIM_MAX_LAUNCH") && r + 1 >= (uint32_t)atoi(getenv("NLZM_SIM_MAX_LAUNCH"))) { g_stop_chunk = c1; break; }
    }
    P.cnt.bt_tests += wk.tests; P.cnt.bt_calls += wk.calls; P.cnt.cmp_bytes += wk.cmp;

    Check c; c.syms = &syms; c.bits = &bits; c.fm = &fm; c.syms_stride = ss; c.bits_stride = bs;
    if (!g_ref.bad) {
        memset(&hk, 0, sizeof hk);
        hk.user = &c; hk.on_frame = on_frame;
        if (nlzm_oracle_compress(in.data(), (uint64_t)n, hb, out.data(), out.size(), &out_n, &st, &hk)) { printf("oracle failed\n"); return 1; }
    }
    const double np = (double)(n ? n : 1);
    printf("finder: %llu blocks (%.1f positions each); cut by: nice %llu, new top entry %llu, RK candidate %llu, RK catch-up %llu, same worker bin %llu, other %llu\n",
           P.prof[0], np / (double)(P.prof[0] ? P.prof[0] : 1), P.prof[1], P.prof[2], P.prof[3], P.prof[4], P.prof[12], P.prof[5]);
    printf("cut-short RK256 entries: %llu became the growing top entry, %llu ended where another entry ends (%llu of them the nearer one)\n", P.prof[115], P.prof[116], P.prof[117]);
    printf("table shape: %llu launches wide, changed %llu times\n", P.prof[114], P.prof[113]);
    printf("table: %llu blocks, %llu on the slow path; parser: %llu blocks (%.1f nodes each), %.2f passes per block, mask fills %llu, probe rounds %llu, re-sampled %llu (put back %llu)\n",
           P.prof[6], P.prof[7], P.prof[8], np / (double)(P.prof[8] ? P.prof[8] : 1), (double)P.prof[13] / (double)(P.prof[8] ? P.prof[8] : 1), P.prof[9], P.prof[10], P.prof[11], P.prof[14]);
    printf("helper parser: %llu jobs posted, %llu taken over (%llu nodes), parser waited %llu sweeps for it; helper: %llu jobs seen, %llu done, %llu blocks\n",
           P.prof[96], P.prof[97], P.prof[98], P.prof[99], P.prof[100], P.prof[101], P.prof[102]);
    printf("workers: uncertain marks %llu (%.2f%%), dry runs %llu\n", unc_total, 100.0 * unc_total / np, wk.dry);
    const int bad = g_ref.bad || c.bad || (g_stop_chunk != 0xFFFFFFFFu ? 0 : 1) * (P.cnt.positions != st.positions || P.cnt.nice_positions != st.nice_positions ||
                    P.cnt.segments != st.segments || P.cnt.bt_tests != st.bt_tests || P.cnt.bt_calls != st.bt_calls ||
                    P.cnt.ht_rows != st.ht_rows || P.cnt.rk_probes != st.rk_probes || P.cnt.rk_inserts != st.rk_inserts ||
                    P.cnt.cmp_bytes != st.cmp_bytes_needed);
    printf("%s: %s  (chunks %u, tables checked %llu, positions %llu/%llu nice %llu/%llu segments %llu/%llu bt_tests %llu/%llu ht_rows %llu/%llu "
           "rk_probes %llu/%llu rk_inserts %llu/%llu cmp_bytes %llu/%llu)\n",
           argv[1], bad ? "MISMATCH" : "OK", g.nchunks, g_ref.checked, P.cnt.positions, (unsigned long long)st.positions,
           P.cnt.nice_positions, (unsigned long long)st.nice_positions, P.cnt.segments, (unsigned long long)st.segments,
           P.cnt.bt_tests, (unsigned long long)st.bt_tests, P.cnt.ht_rows, (unsigned long long)st.ht_rows,
           P.cnt.rk_probes, (unsigned long long)st.rk_probes, P.cnt.rk_inserts, (unsigned long long)st.rk_inserts,
           P.cnt.cmp_bytes, (unsigned long long)st.cmp_bytes_needed);
    return bad;
}
