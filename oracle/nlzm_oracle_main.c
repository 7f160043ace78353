/*
 * nlzm_oracle_main.c -- command line for the CPU restatement.
 *
 * TEST INFRASTRUCTURE ONLY.  Same c/d/t/h commands and -window flag as the
 * reference CLI (NLZM.cpp:2050-2178) so the two binaries can be run side by
 * side; extra: `-stats` prints the operation counters, `-digest:<file>` writes
 * the F2/F3/F4 digests as JSON (same format as the instrumented reference).
 */
#define _POSIX_C_SOURCE 200809L
#include "nlzm_oracle.h"
#include "digest.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

int nlzm_oracle_compress_digest(const uint8_t *, uint64_t, uint32_t, uint8_t *, uint64_t, uint64_t *,
                                nlzm_oracle_stats *, uint64_t *, uint64_t *, uint32_t);

static uint8_t *slurp(const char *path, uint64_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    fseek(f, 0, SEEK_END);
    long long sz = ftello(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *b = (uint8_t *)malloc((size_t)sz + 1);
    if (sz && fread(b, 1, (size_t)sz, f) != (size_t)sz) { free(b); fclose(f); return 0; }
    fclose(f);
    *n = (uint64_t)sz;
    return b;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv)
{
    uint32_t hist_bits = 22;
    int want_stats = 0;
    const char *digest_path = 0;
    printf("NLZM 1.03 oracle (CPU restatement, test infrastructure)\n");
    while (argc >= 2 && argv[1][0] == '-') {
        char *a = argv[1];
        argv++; argc--;
        while (*a == '-') a++;
        if (!strncmp(a, "window:", 7)) {
            int v = atoi(a + 7);
            hist_bits = (uint32_t)(v < 15 ? 15 : (v > 28 ? 28 : v));     /* NLZM.cpp:2085 */
            printf("Window bits: %u\n", hist_bits);
        } else if (!strcmp(a, "stats")) {
            want_stats = 1;
        } else if (!strncmp(a, "digest:", 7)) {
            digest_path = a + 7;
        } else {
            printf("Unrecognized flag %s\n", a);
            return -1;
        }
    }
    const int cmd = argc >= 2 ? (argv[1][0] | 0x20) : 0;
    if (argc == 4 && cmd == 'c') {
        FILE *probe = fopen(argv[3], "rb");
        if (probe) { printf("Error: %s already exists\n", argv[3]); fclose(probe); return -1; }
        uint64_t n = 0;
        uint8_t *in = slurp(argv[2], &n);
        if (!in) { printf("Error: %s file does not exist\n", argv[2]); return -1; }
        const uint64_t cap = nlzm_oracle_bound(n);
        uint8_t *out = (uint8_t *)malloc(cap);
        uint64_t out_n = 0;
        nlzm_oracle_stats st;
        uint64_t dg[6];
        const double t0 = now_s();
        int rc;
        if (digest_path) {
            /* snapshots: one per frame */
            const uint32_t max_frames = (uint32_t)(n / 14848 + 2);
            uint64_t *snaps = (uint64_t *)malloc((size_t)max_frames * 3 * 8);
            rc = nlzm_oracle_compress_digest(in, n, hist_bits, out, cap, &out_n, &st, dg, snaps, max_frames);
            if (!rc) {
                nlzm_digest d; nlzm_digest_init(&d);
                d.mt = dg[0]; d.seg = dg[1]; d.frm = dg[2]; d.n_pos = dg[3]; d.n_seg = dg[4]; d.n_frames = dg[5];
                d.snap = snaps;
                FILE *df = fopen(digest_path, "w");
                if (df) { nlzm_digest_write_json(&d, df, 1); fclose(df); }
            }
            free(snaps);
        } else {
            rc = nlzm_oracle_compress(in, n, hist_bits, out, cap, &out_n, &st, 0);
        }
        const double t1 = now_s();
        if (rc) { printf("Error: compress failed (%d)\n", rc); return -1; }
        FILE *fo = fopen(argv[3], "wb");
        if (!fo) { printf("Error: %s file does not exist\n", argv[3]); return -1; }
        fwrite(out, 1, (size_t)out_n, fo);
        fclose(fo);
        printf("%llu -> %llu\nDone (input CRC32 %X, %.2f sec)\n", (unsigned long long)n,
               (unsigned long long)out_n, nlzm_oracle_crc32(in, n, 0), t1 - t0);
        if (want_stats) {
            printf("{\"in_bytes\": %llu, \"out_bytes\": %llu, \"bt_calls\": %llu, \"bt_tests\": %llu, "
                   "\"cmp_bytes\": %llu, \"ht_rows\": %llu, \"rk_probes\": %llu, \"rk_inserts\": %llu, "
                   "\"positions\": %llu, \"nice_positions\": %llu, \"segments\": %llu, \"seg_rep_grow\": %llu, "
                   "\"n_literal\": %llu, \"n_dict\": %llu, \"n_rep\": %llu, \"rans_syms\": %llu, "
                   "\"bit_ops\": %llu, \"frames\": %llu, \"shifts\": %llu, \"cmp_bytes_needed\": %llu, \"seconds\": %.3f}\n",
                   (unsigned long long)st.in_bytes, (unsigned long long)st.out_bytes,
                   (unsigned long long)st.bt_calls, (unsigned long long)st.bt_tests,
                   (unsigned long long)st.cmp_bytes, (unsigned long long)st.ht_rows,
                   (unsigned long long)st.rk_probes, (unsigned long long)st.rk_inserts,
                   (unsigned long long)st.positions, (unsigned long long)st.nice_positions,
                   (unsigned long long)st.segments, (unsigned long long)st.seg_rep_grow,
                   (unsigned long long)st.n_literal, (unsigned long long)st.n_dict,
                   (unsigned long long)st.n_rep, (unsigned long long)st.rans_syms,
                   (unsigned long long)st.bit_ops, (unsigned long long)st.frames,
                   (unsigned long long)st.shifts, (unsigned long long)st.cmp_bytes_needed, t1 - t0);
        }
        free(out); free(in);
        return 0;
    }
    if ((argc == 4 && cmd == 'd') || (argc == 3 && cmd == 't')) {
        if (cmd == 'd') {
            FILE *probe = fopen(argv[3], "rb");
            if (probe) { printf("Error: %s already exists\n", argv[3]); fclose(probe); return -1; }
        }
        uint64_t n = 0;
        uint8_t *in = slurp(argv[2], &n);
        if (!in) { printf("Error: %s file does not exist\n", argv[2]); return -1; }
        uint64_t out_n = 0;
        int rc = nlzm_oracle_decompress(in, n, 0, 0, &out_n);    /* size pass */
        if (rc) { printf("Error: malformed stream (%d)\n", rc); return -1; }
        uint8_t *out = (uint8_t *)malloc((size_t)out_n + 1);
        const double t0 = now_s();
        rc = nlzm_oracle_decompress(in, n, out, out_n, &out_n);
        const double t1 = now_s();
        if (rc) { printf("Error: malformed stream (%d)\n", rc); return -1; }
        if (cmd == 'd') {
            FILE *fo = fopen(argv[3], "wb");
            if (!fo) { printf("Error: %s file does not exist\n", argv[3]); return -1; }
            fwrite(out, 1, (size_t)out_n, fo);
            fclose(fo);
        }
        printf("%llu -> %llu\nDone (output CRC32 %X, %.2f sec)\n", (unsigned long long)n,
               (unsigned long long)out_n, nlzm_oracle_crc32(out, out_n, 0), t1 - t0);
        free(out); free(in);
        return 0;
    }
    if (argc == 3 && cmd == 'h') {
        uint64_t n = 0;
        uint8_t *in = slurp(argv[2], &n);
        if (!in) { printf("Error: %s file does not exist\n", argv[2]); return -1; }
        printf("%X\n", nlzm_oracle_crc32(in, n, 0));
        free(in);
        return 0;
    }
    printf("Commands:\n\t[flags] c [input] [output]\n\td [input] [output]\n\tt [input]\n\th [input]\n"
           "Flags:\n\t-window:bits (15..28, default 22)\n\t-stats\n\t-digest:<json file>\n");
    return 0;
}
