/*
 * skip_probe.c -- what a BT4 worker could KNOW about the finder's decisions (round 6).
 *
 * TEST INFRASTRUCTURE ONLY (oracle/): a CPU experiment on the restatement, never part of the product.
 *
 * Whether BT4 runs at a position is the finder's decision: it is skipped where the table carried from the position before is
 * 64 or longer (NLZM.cpp:1514).  On the GPU the worker of a BT4 head is ahead of that decision and ASSUMES it for the positions
 * the pre-filter marks (a 65-gram that occurred before): "skip" when the positions before and behind are marked too, "call"
 * otherwise.  A wave that owns a hot head does not start anything behind an assumed "skip" that follows a "call" decision of its
 * bin (wrong one time in eight on prose) -- on source code it stands there a quarter of its steps, and the finder waits a whole
 * call for it at every such position.  But a call of the head itself says more than the marks: a call at q that found a match of
 * L >= 65 bytes puts an entry into the table that is still >= 64 long at every p <= q + L - 64 of the same chunk -- BT4 is skipped
 * there whatever else happens (:1514), provided the call at q stands.  This probe counts, per decision at a marked position,
 * what the marks assume, what the head's own last calls make CERTAIN, and what the decision was.
 *
 *   gcc -O2 -o skip_probe skip_probe.c nlzm_oracle.c && ./skip_probe <file> <window bits>
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nlzm_oracle.h"

static const uint8_t *g_in;
static uint64_t g_n, g_calls0;
static uint32_t g_wmask, g_chunk;
static nlzm_oracle_stats g_st;
static uint8_t *g_mark;                 /* the pre-filter's mark, exact: the 65-gram at p - 1 occurred before inside the window */
static uint8_t *g_called;               /* 1: BT4 ran, 2: it did not (positions with >= 4 bytes ahead) */
static uint16_t *g_best;                /* longest match of the call */

/* exact marks: hash of every 65-gram -> last position */
typedef struct { uint64_t key; uint32_t pos1; } slot_t;
static void make_marks(void)
{
    uint64_t cap = 1; while (cap < 2 * g_n + 16) cap <<= 1;
    slot_t *t = calloc(cap, sizeof *t);
    const uint64_t B = 0x9E3779B97F4A7C15ull;
    uint64_t pw = 1; for (int i = 0; i < 64; i++) pw *= B;
    uint64_t h = 0;
    for (uint64_t a = 0; a + 65 <= g_n; a++) {
        if (a == 0) { for (int j = 0; j < 65; j++) h = h * B + g_in[j] + 1; }
        else h = (h - (g_in[a - 1] + 1) * pw) * B + g_in[a + 64] + 1;
        uint64_t s = (h * 0xD6E8FEB86659FD93ull) >> 20 & (cap - 1);
        while (t[s].pos1 && t[s].key != h) s = (s + 1) & (cap - 1);
        if (t[s].pos1 && a - (t[s].pos1 - 1) <= g_wmask && a + 1 < g_n) g_mark[a + 1] = 1;
        t[s].key = h; t[s].pos1 = (uint32_t)a + 1;
    }
    free(t);
}

static void on_position(void *u, uint64_t a, uint32_t max_len, const uint32_t *delta)
{
    (void)u; (void)max_len; (void)delta;
    const uint64_t calls = g_st.bt_calls - g_calls0;
    g_calls0 = g_st.bt_calls;
    if (a + 4 > g_n) return;
    g_called[a] = calls ? 1 : 2;
    if (calls) g_best[a] = (uint16_t)nlzm_oracle_debug_bt_last_best();
}

typedef struct { uint64_t n, cert, skip, wrong; } cls_t;
static void line(const char *what, const cls_t *c)
{
    printf("  %-58s %10llu | certain skip %9llu (%5.1f %%) | of the rest: skipped %9llu, called (assumption wrong) %8llu (%5.2f %% of all)\n", what,
           (unsigned long long)c->n, (unsigned long long)c->cert, 100.0 * c->cert / (c->n ? c->n : 1), (unsigned long long)c->skip, (unsigned long long)c->wrong,
           100.0 * c->wrong / (c->n ? c->n : 1));
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <file> <window bits>\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END); g_n = (uint64_t)ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *in = malloc(g_n + 64);
    if (fread(in, 1, g_n, f) != g_n) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    g_in = in;
    uint32_t wb = (uint32_t)atoi(argv[2]);
    while (wb > 10 && g_n < (1ull << (wb - 1))) wb--;
    g_wmask = (1u << wb) - 1;
    { uint32_t fb = wb - 2; if (fb < 14) fb = 14; if (fb > 17) fb = 17; g_chunk = ((1u << fb) * 15) / 16 - 0x200; }
    g_mark = calloc(g_n + 2, 1); g_called = calloc(g_n + 2, 1); g_best = calloc(g_n + 2, 2);
    make_marks();
    const uint64_t cap = nlzm_oracle_bound(g_n);
    uint8_t *out = malloc(cap);
    uint64_t on = 0;
    nlzm_oracle_hooks hk; memset(&hk, 0, sizeof hk);
    hk.on_position = on_position;
    if (nlzm_oracle_compress(in, g_n, (uint32_t)atoi(argv[2]), out, cap, &on, &g_st, &hk)) { fprintf(stderr, "oracle failed\n"); return 1; }

    /* heads by their number of positions: the hot ones are what a wave owns */
    const uint32_t shift = 32 - (13 + (wb < 16 ? 16 : wb > 20 ? 20 : wb) - 16), nheads = 1u << (32 - shift);
    uint64_t *cnt = calloc(nheads, 8);
    for (uint64_t a = 0; a + 4 <= g_n; a++) { uint32_t v; memcpy(&v, g_in + a, 4); cnt[nlzm_oracle_hash4(v) >> shift]++; }
    uint64_t thr = 0;
    {   /* a head is "hot" with 8,192 positions per 979 K (a launch of 8 chunks) and more */
        thr = (uint64_t)((double)g_n * 8192.0 / (8.0 * g_chunk));
    }
    uint8_t *last = calloc(nheads, 1);              /* the head's latest decision at a marked position: 1 call, 2 skip */
    uint32_t *cert = calloc(nheads, 4);             /* positions <= this are skipped for certain (same chunk): q + L - 64 of the head's calls */
    uint32_t *cert_chunk = calloc(nheads, 4);
    uint32_t any_cert = 0, any_chunk = 0xFFFFFFFFu;
    cls_t risky[2] = { { 0 } }, calm[2] = { { 0 } }, acall[2] = { { 0 } }, risky_any[2] = { { 0 } };
    uint64_t marked = 0, decided_call = 0, decided_skip = 0, pos = 0;
    for (uint64_t a = 0; a + 4 <= g_n; a++) {
        if (!g_called[a]) continue;
        pos++;
        uint32_t v; memcpy(&v, g_in + a, 4);
        const uint32_t h = nlzm_oracle_hash4(v) >> shift, ck = (uint32_t)(a / g_chunk);
        const int hot = cnt[h] >= thr, called = g_called[a] == 1;
        if (cert_chunk[h] != ck) { cert_chunk[h] = ck; cert[h] = 0; }
        if (any_chunk != ck) { any_chunk = ck; any_cert = 0; }
        if (g_mark[a]) {
            marked++; decided_call += called; decided_skip += !called;
            const int askip = a > 0 && g_mark[a - 1] && g_mark[a + 1];
            const int certain = a <= cert[h], certain_any = a <= any_cert;
            if (certain && called) { fprintf(stderr, "BUG: position %llu certain skip but called\n", (unsigned long long)a); return 1; }
            cls_t *c = askip ? (last[h] == 1 ? &risky[hot] : &calm[hot]) : &acall[hot];
            c->n++;
            if (certain) c->cert++; else if (askip ? !called : called) c->skip++; else c->wrong++;      /* (for `acall`: skip = as assumed) */
            if (askip && last[h] == 1) { cls_t *d = &risky_any[hot]; d->n++; if (certain_any) d->cert++; else if (!called) d->skip++; else d->wrong++; }
            last[h] = called ? 1 : 2;
        }
        if (called && g_best[a] >= 65) {
            const uint32_t e = (uint32_t)a + g_best[a] - 64;
            if (e > cert[h]) cert[h] = e;
            if (e > any_cert) any_cert = e;
        }
    }
    printf("%s: %llu bytes at -window:%u, %llu positions, %llu marked by an exact pre-filter (%.1f %%): %llu called, %llu skipped; hot heads: %llu positions and more\n", argv[1],
           (unsigned long long)g_n, wb, (unsigned long long)pos, (unsigned long long)marked, 100.0 * marked / pos, (unsigned long long)decided_call, (unsigned long long)decided_skip, (unsigned long long)thr);
    for (int hot = 1; hot >= 0; hot--) {
        printf("%s heads, decisions at marked positions:\n", hot ? "HOT" : "other");
        line("assumed skip, the head's last decision was CALL (\"risky\")", &risky[hot]);
        line("  the same, certain by ANY head's last calls", &risky_any[hot]);
        line("assumed skip, the head's last decision was skip / none", &calm[hot]);
        line("assumed call (a neighbour is unmarked) [skip = as assumed]", &acall[hot]);
    }
    return 0;
}
