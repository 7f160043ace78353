/*
 * stale_probe.c -- go/no-go measurement for a "guess and certify" parser (round 5).
 *
 * TEST INFRASTRUCTURE ONLY (oracle/): a CPU experiment on the restatement, never part of the product.
 *
 * Question: if a scout parses a segment with the prices of D positions EARLIER (parse_table only reads the model,
 * NLZM.cpp:1491,1567,1585,1614), how much of what it finds -- per node the winning edge (source, kind, length,
 * distance / rep slot) and the rep set (NLZM.cpp:1460-1467,1575-1576) -- is what the true prices give, and how many
 * Jacobi passes (the GPU parser's full recompute of a block of nodes from the states of the pass before) does a
 * block need when it STARTS from the guess instead of from nothing?
 *
 * Method.  The driver loop below is nlzm_oracle_compress's (NLZM.cpp:1711-1910) with three additions:
 *   - the sampled edges of every position (NLZM.cpp:1558-1562; they depend on the match table only, never on the
 *     model) are recorded per segment through the on_position hook,
 *   - model snapshots are kept per segment start (a ring), so that "the model D positions ago" can be looked up,
 *   - in the sampled chunks every segment is parsed again from the recorded edges: with the true model (must
 *     reproduce the command list: the self-check), and with each stale model; then the GPU's block-Jacobi scheme is
 *     emulated on blocks of NB nodes, from nothing and from each guess.
 *
 * Output: per depth window, the fraction of blocks whose guess is the fixed point (one certifying pass), the mean
 * number of passes from the guess, the mean number of passes from nothing (to be compared with the GPU's own counter:
 * 5.81 passes per 62-node block at 300 MB, profiles/r04_wave_accounting.txt).
 *
 *   gcc -O2 -o stale_probe stale_probe.c && ./stale_probe <file> <window bits> [every_nth_chunk=8] [NB=62]
 */
#define _POSIX_C_SOURCE 200809L
#include "nlzm_oracle.c"
#include <stdio.h>

#define MAXS 40                      /* sampled lengths per position: at most 32 (max_len 33: step 1) */
typedef struct { uint16_t n, tl[MAXS]; uint32_t d[MAXS]; } pedges_t;

typedef struct {
    uint32_t cost, rep[4], delta;    /* delta: distance (dict) or rep slot (rep), as NLZM.cpp:1573,1591 */
    uint16_t from, len;
    uint8_t cmd, rank;
} sn_t;

static pedges_t g_edges[NLZM_PARSE_MAX + 1];
static uint64_t g_seg_abs;           /* absolute position of the running segment's node 0 */
static uint32_t g_seg_cap;           /* max_parse of the running segment */

static void hook_pos(void *u, uint64_t abs_pos, uint32_t mt_max, const uint32_t *delta)
{
    (void)u;
    const uint32_t p = (uint32_t)(abs_pos - g_seg_abs);
    pedges_t *pe = &g_edges[p];
    pe->n = 0;
    uint32_t max_len = u32min(mt_max, g_seg_cap - p);
    if (max_len < NLZM_MATCH_MIN) max_len = 0;
    uint16_t step = (uint16_t)((uint16_t)(max_len - NLZM_MATCH_MIN) >> 4);
    step = (uint16_t)(step + (step == 0));
    for (uint16_t tl = (uint16_t)max_len; tl >= NLZM_MATCH_MIN; tl = (uint16_t)(tl - (tl < step ? tl : step))) {
        /* every sample is kept, also the ones below the distance's minimum length: they open nodes (:1550) */
        pe->tl[pe->n] = tl; pe->d[pe->n] = delta[tl]; pe->n++;
    }
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* in-order parse of a recorded segment under a given model and rep set (parse_table, NLZM.cpp:1464-1651)            */
/* ---------------------------------------------------------------------------------------------------------------- */
typedef struct { int ok; uint32_t end_p; } sres_t;

static inline void s_relax(sn_t *nd, uint32_t p, uint32_t np, uint32_t c, uint8_t cmd, uint8_t rank, uint32_t len,
                           uint32_t store, uint32_t add)
{
    if (nd[np].cost > nd[p].cost + c) {
        sn_t *t = &nd[np];
        t->cost = nd[p].cost + c; t->cmd = cmd; t->rank = rank; t->from = (uint16_t)p; t->len = (uint16_t)len; t->delta = store;
        memcpy(t->rep, nd[p].rep, sizeof t->rep);
        if (cmd) rep_add(t->rep, add);
    }
}

static sres_t s_parse(enc_t *e, const model_t *m, const uint32_t rep0[4], uint32_t seg_reb, uint32_t cap,
                      uint32_t end_true, sn_t *nd)
{
    sres_t r = { 1, 0 };
    for (uint32_t i = 0; i <= end_true + NLZM_MATCH_MAX + 1 && i <= NLZM_PARSE_MAX; i++) { nd[i].cost = 0xFFFFFFFFu; nd[i].from = 0xFFFF; nd[i].cmd = 0xFF; nd[i].rank = 0; nd[i].len = 0; nd[i].delta = 0; memset(nd[i].rep, 0, 16); }
    nd[0].cost = 0; memcpy(nd[0].rep, rep0, 16);
    uint32_t p = 0, end_p = 1;
    while (p < end_p) {
        if (p >= end_true) { r.ok = 0; break; }          /* the guess wants a longer segment: no tables there */
        const uint32_t q = seg_reb + p;
        const uint8_t *cur = e->in + e->base + q;
        s_relax(nd, p, p + 1, price_literal(m, cur[0]), 0, 0, 0, 0, 0);
        const pedges_t *pe = &g_edges[p];
        if (pe->n && end_p < p + pe->tl[0]) end_p = p + pe->tl[0];
        uint32_t checked = 0;
        for (uint32_t k = 0; k < pe->n; k++) {
            const uint32_t tl = pe->tl[k], d = pe->d[k];
            if (tl < nlzm_oracle_match_min(d)) continue;
            s_relax(nd, p, p + tl, price_match(m, d, tl), 1, (uint8_t)(2 * k + 1), tl, d, d);
            const int ri = rep_find(nd[p].rep, d);
            if (ri < 0) continue;
            checked |= 1u << ri;
            s_relax(nd, p, p + tl, price_rep(m, d, tl), 2, (uint8_t)(2 * k + 2), tl, (uint32_t)ri, d);
        }
        if (checked != 15) {
            uint32_t rp[4]; memcpy(rp, nd[p].rep, 16);
            for (uint32_t ri = 0; ri < 4; ri++) {
                if ((checked >> ri) & 1 || rp[ri] >= q) continue;
                const uint32_t pcap = (uint16_t)(cap - p);
                uint32_t l = common_len_signed(e, q - rp[ri], q, u32min(pcap, NLZM_MATCH_MAX), 0) & 0x7FFFFFFFu;
                l = u32min(l, NLZM_MATCH_MAX);
                if (l >= nlzm_oracle_match_min(rp[ri])) {
                    if (end_p < p + l) end_p = p + l;
                    s_relax(nd, p, p + l, price_rep(m, rp[ri], l), 2, (uint8_t)(100 + ri), l, ri, rp[ri]);
                }
            }
        }
        ++p;
    }
    r.end_p = end_p;
    if (r.ok && end_p != end_true) r.ok = 0;             /* the guess ends the segment earlier */
    return r;
}

/* The same parse started at node s0 > 0 as if the segment began there (node s0: cost 0, the given rep set; no edge from an
 * earlier node exists): what a HELPER parser would compute that starts in the middle of a segment, under the segment's own
 * prices.  Used to measure how many nodes it takes until its states agree with the true ones up to a constant cost
 * offset ("rank convergence" of the shortest-path recurrence).  Runs to node `upto` (<= the segment's true end). */
static void s_parse_from(enc_t *e, const model_t *m, const uint32_t rep0[4], uint32_t seg_reb, uint32_t cap, uint32_t s0,
                         uint32_t upto, sn_t *nd)
{
    for (uint32_t i = s0; i <= upto + NLZM_MATCH_MAX + 1 && i <= NLZM_PARSE_MAX; i++) { nd[i].cost = 0xFFFFFFFFu; nd[i].from = 0xFFFF; nd[i].cmd = 0xFF; nd[i].rank = 0; nd[i].len = 0; nd[i].delta = 0; memset(nd[i].rep, 0, 16); }
    nd[s0].cost = 0; memcpy(nd[s0].rep, rep0, 16);
    for (uint32_t p = s0; p < upto; p++) {
        const uint32_t q = seg_reb + p;
        const uint8_t *cur = e->in + e->base + q;
        s_relax(nd, p, p + 1, price_literal(m, cur[0]), 0, 0, 0, 0, 0);
        const pedges_t *pe = &g_edges[p];
        uint32_t checked = 0;
        for (uint32_t k = 0; k < pe->n; k++) {
            const uint32_t tl = pe->tl[k], d = pe->d[k];
            if (tl < nlzm_oracle_match_min(d)) continue;
            s_relax(nd, p, p + tl, price_match(m, d, tl), 1, (uint8_t)(2 * k + 1), tl, d, d);
            const int ri = rep_find(nd[p].rep, d);
            if (ri < 0) continue;
            checked |= 1u << ri;
            s_relax(nd, p, p + tl, price_rep(m, d, tl), 2, (uint8_t)(2 * k + 2), tl, (uint32_t)ri, d);
        }
        if (checked != 15) {
            uint32_t rp[4]; memcpy(rp, nd[p].rep, 16);
            for (uint32_t ri = 0; ri < 4; ri++) {
                if ((checked >> ri) & 1 || rp[ri] >= q) continue;
                const uint32_t pcap = (uint16_t)(cap - p);
                uint32_t l = common_len_signed(e, q - rp[ri], q, u32min(pcap, NLZM_MATCH_MAX), 0) & 0x7FFFFFFFu;
                l = u32min(l, NLZM_MATCH_MAX);
                if (l >= nlzm_oracle_match_min(rp[ri])) s_relax(nd, p, p + l, price_rep(m, rp[ri], l), 2, (uint8_t)(100 + ri), l, ri, rp[ri]);
            }
        }
    }
}

/* Do the states of nodes (M - 264, M] agree up to ONE cost offset (and exactly in winner-independent terms: the rep sets)?
 * Then every candidate of every node behind M is shifted by that offset in the helper's run: same winners, same rep sets. */
static uint64_t g_fs_why[4];
static int frontier_same(const sn_t *a, const sn_t *b, uint32_t s0, uint32_t M)
{
    const uint32_t lo = M >= NLZM_MATCH_MAX ? M - NLZM_MATCH_MAX + 1 : 0;
    if (lo < s0) return 0;
    const uint32_t off = b[M].cost - a[M].cost;
    int bad = 0;
    for (uint32_t t = lo; t <= M; t++) {
        if (b[t].cost - a[t].cost != off) bad |= 1;
        if (memcmp(a[t].rep, b[t].rep, 16)) bad |= 2;
    }
    g_fs_why[bad]++;
    return !bad;
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* the GPU parser's pass on a block of nodes [b0, b1): every node recomputed from the states of the pass before     */
/* (PARSER: push = every sampled edge and probe of every node of the block relaxed from the    */
/* node's OLD cost and rep set into a key min(cost << 32 | source << 8 | rank); update = costs through the literal  */
/* edges by a prefix scan, rep sets from the winners' sources, literal runs inside the pass)                        */
/* ---------------------------------------------------------------------------------------------------------------- */
typedef struct { uint64_t key; uint32_t cost, len, store, add; uint8_t cmd; } cand_t;

static inline uint64_t mk_key(uint32_t cost, uint32_t src, uint32_t rank) { return ((uint64_t)cost << 32) | ((uint64_t)src << 8) | rank; }

static void push_node(enc_t *e, const model_t *m, uint32_t seg_reb, uint32_t cap, uint32_t s, uint32_t cost_s,
                      const uint32_t rep_s[4], uint32_t b0, uint32_t b1, cand_t *best /* indexed t - b0 */)
{
    if (cost_s == 0xFFFFFFFFu) return;
    const uint32_t q = seg_reb + s;
    const pedges_t *pe = &g_edges[s];
    uint32_t checked = 0;
    for (uint32_t k = 0; k < pe->n; k++) {
        const uint32_t tl = pe->tl[k], d = pe->d[k];
        if (tl < nlzm_oracle_match_min(d)) continue;
        const uint32_t t = s + tl;
        const int ri = rep_find(rep_s, d);
        if (ri >= 0) checked |= 1u << ri;
        if (t < b0 || t >= b1) continue;
        cand_t c = { mk_key(cost_s + price_match(m, d, tl), s, 2 * k + 1), 0, tl, d, d, 1 };
        if (c.key < best[t - b0].key) best[t - b0] = c;
        if (ri >= 0) {
            cand_t c2 = { mk_key(cost_s + price_rep(m, d, tl), s, 2 * k + 2), 0, tl, (uint32_t)ri, d, 2 };
            if (c2.key < best[t - b0].key) best[t - b0] = c2;
        }
    }
    if (checked != 15) {
        for (uint32_t ri = 0; ri < 4; ri++) {
            if ((checked >> ri) & 1 || rep_s[ri] >= q) continue;
            const uint32_t pcap = (uint16_t)(cap - s);
            uint32_t l = common_len_signed(e, q - rep_s[ri], q, u32min(pcap, NLZM_MATCH_MAX), 0) & 0x7FFFFFFFu;
            l = u32min(l, NLZM_MATCH_MAX);
            if (l >= nlzm_oracle_match_min(rep_s[ri]) && s + l >= b0 && s + l < b1) {
                cand_t c = { mk_key(cost_s + price_rep(m, rep_s[ri], l), s, 100 + ri), 0, l, ri, rep_s[ri], 2 };
                if (c.key < best[s + l - b0].key) best[s + l - b0] = c;
            }
        }
    }
}

static int same_state(const sn_t *a, const sn_t *b)
{
    return a->cost == b->cost && a->from == b->from && a->cmd == b->cmd && a->len == b->len && a->delta == b->delta &&
           !memcmp(a->rep, b->rep, 16);
}

/* one pass; `fin` = final (true) states of the segment (used for nodes < b0 only), cur/nxt = the block's states */
static void jacobi_pass(enc_t *e, const model_t *m, uint32_t seg_reb, uint32_t cap, const sn_t *fin, uint32_t b0, uint32_t b1,
                        const cand_t *ext, const sn_t *cur, sn_t *nxt)
{
    cand_t best[64 + 8];
    for (uint32_t i = 0; i < b1 - b0; i++) best[i] = ext[i];
    for (uint32_t s = b0; s < b1; s++) push_node(e, m, seg_reb, cap, s, cur[s - b0].cost, cur[s - b0].rep, b0, b1, best);
    for (uint32_t t = b0; t < b1; t++) {
        sn_t *o = &nxt[t - b0];
        const sn_t *prev = (t == b0) ? &fin[t - 1] : &nxt[t - 1 - b0];
        uint64_t lk = ~0ull;
        if (prev->cost != 0xFFFFFFFFu)
            lk = mk_key(prev->cost + price_literal(m, e->in[e->base + seg_reb + t - 1]), t - 1, 0);
        const cand_t *c = &best[t - b0];
        if (lk < c->key) {
            o->cost = (uint32_t)(lk >> 32); o->from = (uint16_t)(t - 1); o->cmd = 0; o->rank = 0; o->len = 0; o->delta = 0;
            memcpy(o->rep, prev->rep, 16);
        } else if (c->key != ~0ull) {
            const uint32_t s = (uint32_t)((c->key >> 8) & 0xFFFFFF);
            const sn_t *src = (s < b0) ? &fin[s] : &cur[s - b0];
            o->cost = (uint32_t)(c->key >> 32); o->from = (uint16_t)s; o->cmd = c->cmd; o->rank = (uint8_t)c->key; o->len = (uint16_t)c->len; o->delta = c->store;
            memcpy(o->rep, src->rep, 16);
            rep_add(o->rep, c->add);
        } else {
            o->cost = 0xFFFFFFFFu; o->from = 0xFFFF; o->cmd = 0xFF; o->rank = 0; o->len = 0; o->delta = 0; memset(o->rep, 0, 16);
        }
    }
}

/* passes until one changes nothing (that one counted), from the block states in `st` (overwritten); asserts the result */
static uint32_t jacobi_block(enc_t *e, const model_t *m, uint32_t seg_reb, uint32_t cap, const sn_t *fin, uint32_t b0, uint32_t b1,
                             sn_t *st, int *exact)
{
    cand_t ext[64 + 8];
    for (uint32_t i = 0; i < b1 - b0; i++) { ext[i].key = ~0ull; }
    const uint32_t lo = b0 > NLZM_MATCH_MAX ? b0 - NLZM_MATCH_MAX : 0;
    for (uint32_t s = lo; s < b0; s++) push_node(e, m, seg_reb, cap, s, fin[s].cost, fin[s].rep, b0, b1, ext);
    sn_t a[64 + 8], b[64 + 8];
    memcpy(a, st, (b1 - b0) * sizeof(sn_t));
    uint32_t passes = 0;
    for (;;) {
        jacobi_pass(e, m, seg_reb, cap, fin, b0, b1, ext, a, b);
        passes++;
        int same = 1;
        for (uint32_t i = 0; i < b1 - b0; i++) if (!same_state(&a[i], &b[i])) { same = 0; break; }
        memcpy(a, b, (b1 - b0) * sizeof(sn_t));
        if (same || passes > 200) break;
    }
    *exact = 1;
    for (uint32_t i = 0; i < b1 - b0; i++) if (!same_state(&a[i], &fin[b0 + i])) *exact = 0;
    return passes;
}

/* ---------------------------------------------------------------------------------------------------------------- */
#define NSTALE 4
static const uint32_t k_stale[NSTALE] = { 4096, 8192, 16384, 32768 };   /* positions of staleness */
#define SNAP_RING 4096
typedef struct { uint64_t abs; model_t m; } snap_t;
static snap_t *g_snaps; static uint64_t g_nsnap;

typedef struct {
    uint64_t blocks, nodes, scratch_passes;
    uint64_t g_blocks[NSTALE], g_same[NSTALE], g_accept[NSTALE], g_passes[NSTALE], g_hist[NSTALE][8];
    uint64_t g_seg[NSTALE], g_seg_bad[NSTALE], g_nodes_same[NSTALE], g_rep0_wrong[NSTALE];
    uint64_t segs, seg_cut, inexact;
    uint64_t t_blocks, t_accept, t_passes;   /* guess = stale prices but the TRUE rep set at the segment start (8192 stale) */
    /* convergence mode: helpers started at node s0 of a long segment; [k]: agree with the truth on the frontier at s0 + kWarm[k] */
    uint64_t c_starts, c_ok[8], c_first[9], c_after_bad[8];
} acc_t;
static const uint32_t kWarm[8] = { 264, 320, 384, 512, 640, 768, 1024, 1536 };

static void report(const char *tag, const acc_t *a, uint32_t NB)
{
    printf("%s: %llu segments (%llu cut at 4096), %llu blocks of <= %u nodes (%.1f nodes each); from nothing: %.2f passes per block; emulation inexact: %llu\n",
           tag, (unsigned long long)a->segs, (unsigned long long)a->seg_cut, (unsigned long long)a->blocks, NB,
           a->blocks ? (double)a->nodes / a->blocks : 0.0, a->blocks ? (double)a->scratch_passes / a->blocks : 0.0, (unsigned long long)a->inexact);
    for (int k = 0; k < NSTALE; k++) {
        if (!a->g_blocks[k]) continue;
        printf("  prices %5u positions stale, rep set chained from the guess before: segments whose end the guess misses %.2f %%, rep set at segment start wrong %.2f %%;"
               " nodes with the true (winner, rep set) %.2f %%; blocks identical %.1f %%, accepted by ONE pass %.1f %%, passes from the guess %.2f  [1:%.1f 2:%.1f 3:%.1f 4:%.1f 5:%.1f 6:%.1f 7+:%.1f %%]\n",
               k_stale[k], 100.0 * a->g_seg_bad[k] / (a->g_seg[k] ? a->g_seg[k] : 1), 100.0 * a->g_rep0_wrong[k] / (a->g_seg[k] ? a->g_seg[k] : 1),
               100.0 * a->g_nodes_same[k] / (a->nodes ? a->nodes : 1),
               100.0 * a->g_same[k] / a->g_blocks[k], 100.0 * a->g_accept[k] / a->g_blocks[k], (double)a->g_passes[k] / a->g_blocks[k],
               100.0 * a->g_hist[k][1] / a->g_blocks[k], 100.0 * a->g_hist[k][2] / a->g_blocks[k], 100.0 * a->g_hist[k][3] / a->g_blocks[k],
               100.0 * a->g_hist[k][4] / a->g_blocks[k], 100.0 * a->g_hist[k][5] / a->g_blocks[k], 100.0 * a->g_hist[k][6] / a->g_blocks[k], 100.0 * a->g_hist[k][7] / a->g_blocks[k]);
    }
    if (a->c_starts) {
        printf("  helpers started inside a segment (same prices, rep set of the segment start): %llu starts; frontier (264 nodes) equal to the truth up to one cost offset after",
               (unsigned long long)a->c_starts);
        for (int k = 0; k < 8; k++) printf(" %u nodes: %.2f %%%s", kWarm[k], 100.0 * a->c_ok[k] / a->c_starts, k < 7 ? "," : "\n");
        printf("    first agreement at"); for (int k = 0; k < 8; k++) printf(" %u: %.2f %%,", kWarm[k], 100.0 * a->c_first[k] / a->c_starts);
        printf(" never within 1536: %.2f %%; agreement lost again after it was reached (must be 0):", 100.0 * a->c_first[8] / a->c_starts);
        for (int k = 0; k < 8; k++) printf(" %llu", (unsigned long long)a->c_after_bad[k]);
        printf("\n");
    }
    if (a->t_blocks)
        printf("  prices  8192 positions stale, TRUE rep set at the segment start: accepted by one pass %.1f %%, passes from the guess %.2f\n",
               100.0 * a->t_accept / a->t_blocks, (double)a->t_passes / a->t_blocks);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: stale_probe <file> <window bits> [every_nth_chunk=8] [NB=62] [report every MB=50]\n"); return 2; }
    const uint32_t every = argc > 3 ? (uint32_t)atoi(argv[3]) : 8;
    const uint32_t NB = argc > 4 ? (uint32_t)atoi(argv[4]) : 62;
    const uint64_t rep_mb = argc > 5 ? (uint64_t)atoll(argv[5]) : 50;
    const int conv_mode = argc > 6 && !strcmp(argv[6], "conv");
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END); const uint64_t n = (uint64_t)ftello(f); fseek(f, 0, SEEK_SET);
    uint8_t *src = (uint8_t *)malloc(n + 1);
    if (fread(src, 1, n, f) != n) return 1;
    fclose(f);

    nlzm_oracle_init();
    nlzm_oracle_stats stats; memset(&stats, 0, sizeof stats);
    uint32_t hb, fb, chunk_size, feed;
    nlzm_oracle_geometry(n, (uint32_t)atoi(argv[2]), &hb, &fb, &chunk_size, &feed);
    const uint32_t W = 1u << hb, frame_size = 1u << fb;
    enc_t e; memset(&e, 0, sizeof e);
    e.in = src; e.n = n; e.wbits = hb; e.wmask = W - 1; e.base = 0; e.st = &stats;
    ht_init(&e.ht2, 12, 1, hb);
    ht_init(&e.ht3, 12 + clampu(hb, 15, 17) - 15, 2, hb);
    { const uint32_t bits = 13 + clampu(hb, 16, 20) - 16; e.bt.shift = 32 - bits;
      e.bt.heads = (uint32_t *)malloc((size_t)4 << bits); e.bt.tree = (uint32_t *)malloc((size_t)8 << hb);
      memset(e.bt.heads, 0xFF, (size_t)4 << bits); memset(e.bt.tree, 0xFF, (size_t)8 << hb); }
    { const uint32_t bits = 15 + clampu(hb, 16, 22) - 16; e.rk.shift = 32 - bits; e.rk.tag_mask = (uint32_t)((1ull << (32 - hb)) - 1);
      e.rk.table = (uint32_t *)malloc((size_t)4 << bits); memset(e.rk.table, 0xFF, (size_t)4 << bits); }
    model_t *model = (model_t *)malloc(sizeof *model);
    parse_t *ps = (parse_t *)malloc(sizeof *ps);
    model_reset(model);
    mtab_t carry; carry.max_len = 0;
    frame_t fr; fr.cap_syms = 4 * frame_size; fr.cap_bits = frame_size;
    fr.syms = (uint32_t *)malloc((size_t)fr.cap_syms * 4); fr.bits = (uint8_t *)malloc(fr.cap_bits);
    g_snaps = (snap_t *)calloc(SNAP_RING, sizeof(snap_t));
    sn_t *truth = (sn_t *)malloc((NLZM_PARSE_MAX + 2) * sizeof(sn_t));
    sn_t *guess = (sn_t *)malloc((NLZM_PARSE_MAX + 2) * sizeof(sn_t));
    nlzm_oracle_hooks hk; memset(&hk, 0, sizeof hk); hk.on_position = hook_pos;

    acc_t acc, tot; memset(&acc, 0, sizeof acc); memset(&tot, 0, sizeof tot);
    uint32_t chain_rep[NSTALE][4]; int chain_ok[NSTALE] = { 0 };
    uint64_t chunk_abs = 0, next_report = rep_mb * 1000000ull; uint32_t chunk_idx = 0;
    while (chunk_abs < n) {
        const uint32_t chunk_read = (uint32_t)((n - chunk_abs < feed) ? (n - chunk_abs) : feed);
        const uint32_t p_end = u32min(chunk_size, chunk_read);
        fr.nsyms = 0; fr.nbits_bytes = 0; fr.word = 0; fr.word_bits = 0; fr.num_ops = 0;
        if (chunk_abs - e.base >= 2ull * W) rebase(&e);
        const uint32_t chunk_reb = (uint32_t)(chunk_abs - e.base);
        e.la_end = chunk_reb + chunk_read;
        const int probe = (chunk_idx % every) == every - 1;
        if (probe) for (int k = 0; k < NSTALE; k++) chain_ok[k] = 0;
        uint32_t p = 0;
        while (p < p_end) {
            /* snapshot ring: one per segment start, thinned to one per 256 positions */
            if (!g_nsnap || chunk_abs + p >= g_snaps[(g_nsnap - 1) % SNAP_RING].abs + 256) {
                snap_t *s = &g_snaps[g_nsnap % SNAP_RING]; s->abs = chunk_abs + p; s->m = *model; g_nsnap++;
            }
            g_seg_abs = chunk_abs + p; g_seg_cap = u32min(p_end - p, NLZM_PARSE_MAX);
            const uint32_t seg_len = parse_segment(&e, ps, model, &carry, chunk_reb + p, p_end - p, &hk);
            if (probe && seg_len > 1) {
                const nlzm_oracle_stats keep = stats;
                const uint32_t seg_reb = chunk_reb + p, cap = g_seg_cap;
                sres_t r = s_parse(&e, model, model->rep, seg_reb, cap, seg_len, truth);
                /* self-check: the recorded edges + the true model give the reference's command list */
                { uint32_t cur = seg_len, nn = 0; int ok = r.ok;
                  while (cur != 0 && ok) { const sn_t *t = &truth[cur]; const nlzm_oracle_cmd *c = &ps->cmds[ps->ncmds - 1 - nn];
                      if (nn >= ps->ncmds || c->cmd != t->cmd || c->len != t->len || (c->cmd && c->delta != t->delta)) { ok = 0; fprintf(stderr, "node %u: ref cmd %u len %u delta %u | mine cmd %u len %u delta %u from %u\n", cur, c->cmd, c->len, c->delta, t->cmd, t->len, t->delta, t->from); } nn++; cur = t->from; }
                  if (!ok || nn != ps->ncmds) { fprintf(stderr, "self-check failed at %llu: seg_len %u r.ok %d r.end %u nn %u ncmds %u\n", (unsigned long long)g_seg_abs, seg_len, r.ok, r.end_p, nn, ps->ncmds); return 3; } }
                acc.segs++; if (seg_len == NLZM_PARSE_MAX) acc.seg_cut++;
                if (conv_mode) {
                    /* helpers at every 512th node of a long segment */
                    for (uint32_t s0 = 512; s0 + 264 < seg_len; s0 += 512) {
                        const uint32_t upto = u32min(seg_len, s0 + 1536);
                        s_parse_from(&e, model, model->rep, seg_reb, cap, s0, upto, guess);
                        acc.c_starts++;
                        int first = 8, was = 0;
                        for (int k = 0; k < 8; k++) {
                            const uint32_t M = s0 + kWarm[k];
                            if (M > upto) { if (was) acc.c_ok[k]++; continue; }      /* (beyond the segment: counts as the last state) */
                            const int ok = frontier_same(truth, guess, s0, M);
                            if (ok) { acc.c_ok[k]++; if (first == 8) first = k; was = 1; } else { if (was) acc.c_after_bad[k]++; was = 0; }
                        }
                        acc.c_first[first]++;
                    }
                    stats = keep;
                    goto emit_it;
                }
                /* from nothing */
                for (uint32_t b0 = 1; b0 <= seg_len; b0 += NB) {
                    const uint32_t b1 = u32min(b0 + NB, seg_len + 1);
                    sn_t st[64 + 8];
                    for (uint32_t i = 0; i < b1 - b0; i++) { st[i].cost = 0xFFFFFFFFu; st[i].from = 0xFFFF; st[i].cmd = 0xFF; st[i].rank = 0; st[i].len = 0; st[i].delta = 0; memset(st[i].rep, 0, 16); }
                    int exact; const uint32_t np = jacobi_block(&e, model, seg_reb, cap, truth, b0, b1, st, &exact);
                    acc.blocks++; acc.nodes += b1 - b0; acc.scratch_passes += np; if (!exact) acc.inexact++;
                }
                for (int k = 0; k <= NSTALE; k++) {
                    const int kk = k < NSTALE ? k : 1;                       /* the extra round: 8192 stale, true rep set */
                    const uint64_t want = g_seg_abs >= k_stale[kk] ? g_seg_abs - k_stale[kk] : 0;
                    const snap_t *sm = 0;
                    for (uint64_t j = g_nsnap; j-- > 0 && j + SNAP_RING > g_nsnap;) { const snap_t *s = &g_snaps[j % SNAP_RING]; if (s->abs <= want) { sm = s; break; } }
                    if (!sm) continue;
                    const uint32_t *rep0 = (k < NSTALE && chain_ok[k]) ? chain_rep[k] : model->rep;
                    const int rep0_wrong = memcmp(rep0, model->rep, 16) != 0;
                    sres_t g = s_parse(&e, &sm->m, rep0, seg_reb, cap, seg_len, guess);
                    if (k < NSTALE) {
                        acc.g_seg[k]++; if (!g.ok) acc.g_seg_bad[k]++; if (rep0_wrong) acc.g_rep0_wrong[k]++;
                        /* the scout goes on from ITS final node of the segment (if it ended where the segment ends) */
                        if (g.ok) { memcpy(chain_rep[k], guess[seg_len].rep, 16); chain_ok[k] = 1; } else chain_ok[k] = 0;
                    }
                    for (uint32_t b0 = 1; b0 <= seg_len; b0 += NB) {
                        const uint32_t b1 = u32min(b0 + NB, seg_len + 1);
                        sn_t st[64 + 8]; int same = 1;
                        for (uint32_t t = b0; t < b1; t++) {
                            sn_t *o = &st[t - b0]; *o = guess[t];
                            /* the cost of the guessed winner under the TRUE prices, along the guessed tree */
                            if (o->from == 0xFFFF || o->cmd == 0xFF) { o->cost = 0xFFFFFFFFu; }
                            else {
                                const uint32_t s = o->from; const uint32_t cs = s < b0 ? truth[s].cost : st[s - b0].cost;
                                uint32_t pr;
                                if (o->cmd == 0) pr = price_literal(model, e.in[e.base + seg_reb + s]);
                                else if (o->cmd == 1) pr = price_match(model, o->delta, o->len);
                                else { const uint32_t *rs = guess[s].rep; pr = price_rep(model, rs[o->delta & 3], o->len); }
                                o->cost = cs == 0xFFFFFFFFu ? cs : cs + pr;
                            }
                            const sn_t *tr = &truth[t];
                            const int node_same = o->from == tr->from && o->cmd == tr->cmd && o->len == tr->len && o->delta == tr->delta && !memcmp(o->rep, tr->rep, 16);
                            if (k < NSTALE && node_same) acc.g_nodes_same[k]++;
                            if (!node_same) same = 0;
                        }
                        int exact; const uint32_t np = jacobi_block(&e, model, seg_reb, cap, truth, b0, b1, st, &exact);
                        if (!exact) acc.inexact++;
                        if (k < NSTALE) { acc.g_blocks[k]++; acc.g_same[k] += same; acc.g_accept[k] += np == 1; acc.g_passes[k] += np; acc.g_hist[k][np > 7 ? 7 : np]++; }
                        else { acc.t_blocks++; acc.t_accept += np == 1; acc.t_passes += np; }
                    }
                }
                stats = keep;
            }
emit_it:
            for (uint32_t i = 0; i < ps->ncmds; i++) {
                const nlzm_oracle_cmd c = ps->cmds[i];
                if (c.cmd == 0) { emit_literal(&fr, model, src[chunk_abs + p]); p += 1; }
                else if (c.cmd == 1) { emit_match(&fr, model, c.delta, c.len); rep_add(model->rep, c.delta); p += c.len; }
                else { emit_rep(&fr, model, c.delta, c.len); rep_add(model->rep, model->rep[c.delta]); p += c.len; }
            }
        }
        chunk_abs += p_end; chunk_idx++;
        if (chunk_abs >= next_report || chunk_abs >= n) {
            char tag[64]; snprintf(tag, sizeof tag, "[%llu - %llu MB]", (unsigned long long)((next_report / 1000000ull) - rep_mb), (unsigned long long)(chunk_abs / 1000000ull));
            report(tag, &acc, NB);
            uint64_t *a = (uint64_t *)&acc, *t = (uint64_t *)&tot;
            for (size_t i = 0; i < sizeof acc / 8; i++) { t[i] += a[i]; a[i] = 0; }
            next_report += rep_mb * 1000000ull;
        }
    }
    report("[whole input]", &tot, NB);
    if (conv_mode) printf("frontier tests: equal %llu, cost offsets differ %llu, rep sets differ %llu, both %llu\n", (unsigned long long)g_fs_why[0], (unsigned long long)g_fs_why[1], (unsigned long long)g_fs_why[2], (unsigned long long)g_fs_why[3]);
    return 0;
}
