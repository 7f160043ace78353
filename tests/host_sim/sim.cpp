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

#include <vector>

#define NLZM_HD inline
#define NLZM_HDN
#include "../../nlzm_amd/csrc/nlzm_core.h"
#include "../../oracle/nlzm_oracle.h"

using namespace nlzm;

struct HostWave {
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
}

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

    Geom g; make_geom((uint64_t)n, hb, g);
    std::vector<uint32_t> rkhash((size_t)n + 1, 0);
    for (long long a = 0; a + 256 <= n; a++) {
        if (a == 0) rkhash[0] = nlzm_oracle_rk_hash256(in.data());
        else rkhash[a] = (in[a + 255] + rkhash[a - 1] - in[a - 1] * 0x0E4EA401u) * 0x2F0FD693u;
    }
    std::vector<uint32_t> ht2(4096, kNone), ht3((size_t)2 << (32 - g.ht3_shift), kNone), rkt((size_t)1 << (32 - g.rk_shift), kNone),
        heads((size_t)1 << (32 - g.bt_shift), kNone), tree((size_t)2 << g.wbits, kNone);
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

    MasterLds *lds = new MasterLds;
    Master<HostWave> m;
    m.g = g; m.L = lds;
    memset(&m.G, 0, sizeof m.G);
    m.G.in = in.data(); m.G.rkhash = rkhash.data(); m.G.ht2 = ht2.data(); m.G.ht3 = ht3.data(); m.G.rk_table = rkt.data();
    m.G.bt_heads = heads.data(); m.G.bt_tree = tree.data(); m.G.persist = &P;
    m.G.syms = syms.data(); m.G.syms_stride = ss; m.G.bits = bits.data(); m.G.bits_stride = bs; m.G.fmeta = fm.data(); m.G.chunk0 = 0;
    if (check_tables) { m.G.cap_words = cap.data(); m.G.cap_cap = cap.size(); m.G.cap_lo = 0; m.G.cap_hi = ~0ull; m.G.cap_used = &cap_used; }
    // two launches, to exercise the state save/restore path
    const uint32_t half = g.nchunks / 2;
    m.run(0, half);
    m.run(half, g.nchunks);
    if (P.error) { printf("sim error %u\n", P.error); return 1; }

    Check c; c.syms = &syms; c.bits = &bits; c.fm = &fm; c.syms_stride = ss; c.bits_stride = bs;
    c.cap = cap.data(); c.cap_used = cap_used; c.check_tables = check_tables;
    nlzm_oracle_hooks hk; memset(&hk, 0, sizeof hk);
    hk.user = &c; hk.on_frame = on_frame; hk.on_position = on_pos;
    std::vector<uint8_t> out(nlzm_oracle_bound((uint64_t)n));
    uint64_t out_n = 0; nlzm_oracle_stats st;
    int rc = nlzm_oracle_compress(in.data(), (uint64_t)n, hb, out.data(), out.size(), &out_n, &st, &hk);
    if (rc) { printf("oracle failed %d\n", rc); return 1; }
    printf("%s: %s  (chunks %u, positions sim %llu oracle %llu, bt_tests sim %llu oracle %llu, cmp_bytes sim %llu oracle %llu)\n",
           argv[1], c.bad ? "MISMATCH" : "OK", g.nchunks, P.cnt.positions, (unsigned long long)st.positions, P.cnt.bt_tests,
           (unsigned long long)st.bt_tests, P.cnt.cmp_bytes, (unsigned long long)st.cmp_bytes);
    return c.bad;
}
