/*
 * head_probe.c -- what the BT4 calls of the hottest heads look like (round 5).
 *
 * TEST INFRASTRUCTURE ONLY (oracle/): a CPU experiment on the restatement, never part of the product.
 *
 * On the GPU the calls of one BT4 head are one serial chain (a call walks the tree the call before it left), so a head
 * that holds a large share of the positions bounds the stream's rate by (its calls) x (the time of one call).  This
 * probe says how many calls the hottest heads get, how many tests a call makes (MatchFinderBT::FindAndUpdate,
 * NLZM.cpp:977-1022, at most 256: :988) and how many bytes it compares -- through the statistics the oracle keeps
 * anyway, read before and after every position (the on_position hook); heads are told apart by their four bytes.
 *
 *   gcc -O2 -o head_probe head_probe.c nlzm_oracle.c && ./head_probe <file> <window bits>
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nlzm_oracle.h"

#define SLOTS (1u << 22)
typedef struct { uint32_t key, used; uint64_t calls, tests, cmp, capped; uint32_t hist[9]; } head_t;   /* hist: tests per call 0, 1-2, 3-4, 5-8, ..., 129-256 */
static head_t *g_tab;
static const uint8_t *g_in;
static uint64_t g_n, g_tests0, g_cmp0, g_calls0;
static nlzm_oracle_stats g_st;

static void on_position(void *u, uint64_t a, uint32_t max_len, const uint32_t *delta)
{
    (void)u; (void)max_len; (void)delta;
    const uint64_t calls = g_st.bt_calls - g_calls0, tests = g_st.bt_tests - g_tests0, cmp = g_st.cmp_bytes - g_cmp0;
    g_calls0 = g_st.bt_calls; g_tests0 = g_st.bt_tests; g_cmp0 = g_st.cmp_bytes;
    if (!calls || a + 4 > g_n) return;
    uint32_t v; memcpy(&v, g_in + a, 4);
    uint32_t s = (v * 2654435761u) >> 10;
    while (g_tab[s].used && g_tab[s].key != v) s = (s + 1) & (SLOTS - 1);
    head_t *h = &g_tab[s];
    h->used = 1; h->key = v; h->calls++; h->tests += tests; h->cmp += cmp; h->capped += tests >= 256;
    static const uint32_t edge[8] = { 0, 1, 2, 4, 8, 16, 32, 64 };                             /* 0 | 1 | 2 | 3-4 | 5-8 | ... | 65+ */
    int b = 8; for (int k = 0; k < 8; k++) if (tests <= edge[k]) { b = k; break; }
    h->hist[b]++;
}

static int by_calls(const void *x, const void *y)
{
    const head_t *a = x, *b = y;
    return a->calls < b->calls ? 1 : a->calls > b->calls ? -1 : 0;
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
    g_tab = calloc(SLOTS, sizeof *g_tab);
    const uint64_t cap = nlzm_oracle_bound(g_n);
    uint8_t *out = malloc(cap);
    uint64_t on = 0;
    nlzm_oracle_hooks hk; memset(&hk, 0, sizeof hk);
    hk.on_position = on_position;
    if (nlzm_oracle_compress(in, g_n, (uint32_t)atoi(argv[2]), out, cap, &on, &g_st, &hk)) { fprintf(stderr, "oracle failed\n"); return 1; }
    qsort(g_tab, SLOTS, sizeof *g_tab, by_calls);
    printf("%s: %llu bytes, %llu positions, %llu BT4 calls (%.1f %% of the positions), %.1f tests per call\n", argv[1], (unsigned long long)g_n,
           (unsigned long long)g_st.positions, (unsigned long long)g_st.bt_calls, 100.0 * (double)g_st.bt_calls / (double)g_st.positions, (double)g_st.bt_tests / (double)g_st.bt_calls);
    printf("%-14s %10s %8s %8s %10s %9s   tests per call: 0 | 1 | 2 | 3-4 | 5-8 | 9-16 | 17-32 | 33-64 | 65+ (%% of the head's calls)\n", "head (4 bytes)", "calls", "% calls", "tests/c", "bytes/test", "% at 256");
    uint64_t cum = 0;
    for (int i = 0; i < 12 && g_tab[i].used; i++) {
        const head_t *h = &g_tab[i];
        char txt[32]; int o = 0;
        for (int k = 0; k < 4; k++) { const unsigned c = (h->key >> (8 * k)) & 255u; o += snprintf(txt + o, sizeof txt - (size_t)o, c >= 33 && c < 127 ? "%c" : (c == 32 ? "_" : "\\x%02x"), c); }
        cum += h->calls;
        printf("%-14s %10llu %8.2f %8.1f %10.1f %9.2f  ", txt, (unsigned long long)h->calls, 100.0 * (double)h->calls / (double)g_st.bt_calls, (double)h->tests / (double)h->calls,
               (double)h->cmp / (double)(h->tests ? h->tests : 1), 100.0 * (double)h->capped / (double)h->calls);
        for (int b = 0; b < 9; b++) printf(" %5.1f", 100.0 * h->hist[b] / (double)h->calls);
        printf("\n");
    }
    printf("the twelve together: %.1f %% of the calls\n", 100.0 * (double)cum / (double)g_st.bt_calls);
    return 0;
}
