/*
 * seg_probe.c -- go/no-go measurement for "the finder knows where a segment starts" (round 5).
 *
 * TEST INFRASTRUCTURE ONLY (oracle/): a CPU experiment on the restatement, never part of the product.
 *
 * Why.  Inside a nice region (carried match >= 64, NLZM.cpp:1514) the finders are called at every eighth position counted
 * from the SEGMENT's start, and where a segment starts is the parser's knowledge: the GPU's finder stage stands at the
 * parser's word at every start of a nice region (16 % of its blocks on real text).
 *
 * Claim to be priced.  A segment (parse_table, NLZM.cpp:1464-1651) ends at the first position that no sampled edge
 * reaches -- open_nodes(max_len + p) with max_len = min(table's longest, max_parse - p), NLZM.cpp:1545-1550 -- or at
 * 4,096 positions or at the chunk's end.  All of that is the match tables' doing, which the finder stage has; the one
 * exception is a rep probe (NLZM.cpp:1598-1628) that reaches beyond every sampled edge, which hangs on the parse.
 *
 * Method.  The oracle runs with two hooks: on_position records the table's longest entry per position, on_segment the
 * true segment starts.  Then the segmentation is PREDICTED from the recorded lengths alone, chunk by chunk, and compared:
 *   - segments whose predicted start is not a true start,
 *   - starts of nice regions (the carried longest entry reaches 64 for the first time) at which the predicted segment
 *     start differs from the true one -- each of these is a finder block that would have to be re-run,
 *   - how long a wrong prediction lasts (positions until prediction and truth share a segment start again).
 *
 *   gcc -O2 -o seg_probe seg_probe.c nlzm_oracle.c && ./seg_probe <file> <window bits>
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nlzm_oracle.h"

static uint16_t *g_len;         /* the table's longest entry after the finders, per position (NLZM.cpp:1543) */
static uint8_t *g_start;        /* 1: a true segment starts here */
static uint64_t g_n;

static void on_position(void *u, uint64_t a, uint32_t max_len, const uint32_t *delta)
{
    (void)u; (void)delta;
    if (a < g_n) g_len[a] = (uint16_t)max_len;
}
static void on_segment(void *u, uint64_t a, uint32_t seg_len, const nlzm_oracle_cmd *c, uint32_t nc)
{
    (void)u; (void)seg_len; (void)c; (void)nc;
    if (a < g_n) g_start[a] = 1;
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
    const uint32_t wb = (uint32_t)atoi(argv[2]);
    g_len = calloc(g_n + 1, sizeof *g_len); g_start = calloc(g_n + 1, 1);
    const uint64_t cap = nlzm_oracle_bound(g_n);
    uint8_t *out = malloc(cap);
    uint64_t on = 0;
    nlzm_oracle_stats st;
    nlzm_oracle_hooks hk; memset(&hk, 0, sizeof hk);
    hk.on_position = on_position; hk.on_segment = on_segment;
    if (nlzm_oracle_compress(in, g_n, wb, out, cap, &on, &st, &hk)) { fprintf(stderr, "oracle failed\n"); return 1; }
    uint32_t hb, fb, cs, feed;
    nlzm_oracle_geometry(g_n, wb, &hb, &fb, &cs, &feed);

    uint64_t segs = 0, pred_segs = 0, pred_wrong = 0, nice_starts = 0, nice_wrong = 0, wrong_positions = 0, rep_grown = st.seg_rep_grow;
    for (uint64_t c0 = 0; c0 < g_n; c0 += cs) {
        const uint64_t c1 = c0 + cs < g_n ? c0 + cs : g_n;
        uint64_t s = c0, end = c0;          /* the predicted segment: its start, the farthest node a sampled edge has opened */
        uint64_t ts = c0;                   /* the true segment's start */
        int was_nice = 0;
        for (uint64_t p = c0; p < c1; p++) {
            if (g_start[p]) { ts = p; segs++; }
            if (p == c0 || p == end || p - s == NLZM_PARSE_MAX) {      /* nothing reaches p (or the cut): a segment starts */
                s = p; end = p + 1; pred_segs++;
                if (!g_start[p]) pred_wrong++;
            }
            if (s != ts) wrong_positions++;
            /* nice at p: the entry carried from p - 1 is still >= 64 long (its extension at p can only add to that) */
            const int nice = p > c0 && g_len[p - 1] >= 1 + 64;
            if (nice && !was_nice) { nice_starts++; if (s != ts) nice_wrong++; }
            was_nice = nice;
            uint64_t room = c1 - s; if (room > NLZM_PARSE_MAX) room = NLZM_PARSE_MAX;     /* max_parse (NLZM.cpp:1469, :1802) */
            uint64_t l = g_len[p];
            if (l > room - (p - s)) l = room - (p - s);
            if (l < NLZM_MATCH_MIN) l = 0;
            if (l && p + l > end) end = p + l;
        }
    }
    printf("%s: %llu bytes, window %u, %llu positions, stream %llu bytes\n", argv[1], (unsigned long long)g_n, hb, (unsigned long long)st.positions, (unsigned long long)on);
    printf("segments: %llu true, %llu predicted from the tables' longest entries; predicted starts that are not true starts: %llu (%.4f %%)\n",
           (unsigned long long)segs, (unsigned long long)pred_segs, (unsigned long long)pred_wrong, 100.0 * (double)pred_wrong / (double)(pred_segs ? pred_segs : 1));
    printf("rep probes that opened nodes beyond what was open (the oracle's seg_rep_grow: events, several per segment; most stay below what a later sampled edge opens anyway): %llu\n",
           (unsigned long long)rep_grown);
    printf("starts of nice regions: %llu; with the predicted segment start different from the true one: %llu (%.4f %%)\n",
           (unsigned long long)nice_starts, (unsigned long long)nice_wrong, 100.0 * (double)nice_wrong / (double)(nice_starts ? nice_starts : 1));
    printf("positions at which prediction and truth are in different segments: %llu (%.4f %%)\n",
           (unsigned long long)wrong_positions, 100.0 * (double)wrong_positions / (double)(g_n ? g_n : 1));
    return 0;
}
