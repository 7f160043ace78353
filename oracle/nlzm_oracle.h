/*
 * nlzm_oracle.h -- CPU restatement of NLZM 1.03's compress/decompress path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may build, link or call it, and only as the checker.  The product
 * (nlzm_amd/csrc, include/nlzm_hip.h) never includes or links this file.
 *
 * Parity pinning: this restatement is checked byte-for-byte against the
 * reference itself (oracle/_ref/nlzm_ref, built by oracle/Makefile from
 * /root/reference/NLZM.cpp where it lies) and against golden vectors that
 * were generated from that build (tests/golden/, scripts in oracle/).
 */
#ifndef NLZM_ORACLE_H
#define NLZM_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLZM_MATCH_MIN 2u
#define NLZM_MATCH_MAX 264u      /* NLZM.cpp:737  MATCH_MIN + 255 + 7 */
#define NLZM_NICE_LEN 64u        /* NLZM.cpp:734 */
#define NLZM_PARSE_MAX 4096u     /* NLZM.cpp:1458 */

/* Deterministic operation counters. Identical for any
 * bit-exact implementation of the path, so they define "algorithmic bytes". */
typedef struct nlzm_oracle_stats {
    uint64_t in_bytes, out_bytes;
    uint64_t bt_calls;      /* MatchFinderBT::FindAndUpdate invocations (NLZM.cpp:978) */
    uint64_t bt_tests;      /* loop iterations at NLZM.cpp:989 */
    uint64_t cmp_bytes;     /* byte pairs compared at NLZM.cpp:863 */
    uint64_t ht_rows;       /* rows visited at NLZM.cpp:919 */
    uint64_t rk_probes;     /* executions of NLZM.cpp:1091 */
    uint64_t rk_inserts;    /* executions of NLZM.cpp:1086 and :1111 */
    uint64_t positions;     /* loop iterations of NLZM.cpp:1486 */
    uint64_t nice_positions;/* positions with carried max_len >= 64 (NLZM.cpp:1514 false) */
    uint64_t segments;      /* parse_table calls */
    uint64_t seg_rep_grow;  /* explicit rep probes that grew end_p (NLZM.cpp:1608) */
    uint64_t n_literal, n_dict, n_rep;
    uint64_t rans_syms, bit_ops, frames;
    uint64_t shifts;        /* window rebases (NLZM.cpp:1786) */
    uint64_t cmp_bytes_needed; /* cmp_bytes without the bytes an explicit rep probe compares beyond 264: the reference
                              * measures up to max_parse_len - p <= 4096 bytes and then caps the length at 264
                              * (NLZM.cpp:1605-1606), so those bytes cannot change the result */
} nlzm_oracle_stats;

/* One parsed command as emitted by the driver loop (NLZM.cpp:1809-1843). */
typedef struct nlzm_oracle_cmd {
    uint8_t cmd;        /* 0 literal, 1 dict, 2 rep */
    uint16_t len;       /* 0 for a literal */
    uint32_t delta;     /* distance (dict) or rep index (rep) */
} nlzm_oracle_cmd;

/* Optional observation hooks -- used to produce/compare golden intermediates. */
typedef struct nlzm_oracle_hooks {
    void *user;
    /* after the finders ran for absolute input position `abs_pos`
     * (the state the reference copies into mt_carry at NLZM.cpp:1543) */
    void (*on_position)(void *user, uint64_t abs_pos, uint32_t max_len, const uint32_t *delta);
    /* one call per parse_table() result */
    void (*on_segment)(void *user, uint64_t abs_start, uint32_t seg_len,
                       const nlzm_oracle_cmd *cmds, uint32_t ncmds);
    /* immediately before/after CodeFrame::Flush (NLZM.cpp:1850) */
    void (*on_frame)(void *user, uint32_t frame_idx, uint32_t num_ops,
                     const uint32_t *syms, uint32_t nsyms,
                     const uint8_t *bits, uint32_t nbits_bytes,
                     const uint8_t *frame_bytes, uint32_t frame_len);
} nlzm_oracle_hooks;

void nlzm_oracle_init(void);

uint64_t nlzm_oracle_bound(uint64_t n);

/* Whole stream: header + frames + terminator, exactly what encode_file writes
 * (NLZM.cpp:1711-1910).  hist_bits_req is the value *after* the CLI clamp. */
int nlzm_oracle_compress(const uint8_t *src, uint64_t n, uint32_t hist_bits_req,
                         uint8_t *dst, uint64_t dst_cap, uint64_t *dst_len,
                         nlzm_oracle_stats *stats, const nlzm_oracle_hooks *hooks);

/* decode_file (NLZM.cpp:1912-2039).  dst may be NULL to only count bytes.
 * Returns 0 on success, negative on malformed input. */
int nlzm_oracle_decompress(const uint8_t *src, uint64_t n,
                           uint8_t *dst, uint64_t dst_cap, uint64_t *dst_len);

uint32_t nlzm_oracle_crc32(const uint8_t *p, uint64_t n, uint32_t crc);

/* Test diagnostics: matches RK256 took whose compare the uint16 length parameter (NLZM.cpp:760) ended and that were the
 * table's longest entry, summed over the process's compress calls so far (such an entry grows again at the next position,
 * NLZM.cpp:1503-1512). */
uint64_t nlzm_oracle_debug_rk_u16_cuts(void);
uint32_t nlzm_oracle_debug_bt_last_best(void);   /* the longest match of the last BT4 call (oracle/skip_probe.c) */

/* Small pure functions exposed for known-answer tests. */
const uint16_t *nlzm_oracle_log2_lut(void);                 /* 256 entries */
uint32_t nlzm_oracle_match_min(uint32_t dist);
uint32_t nlzm_oracle_hash4(uint32_t x);
uint32_t nlzm_oracle_rk_hash256(const uint8_t *win);        /* hash of win[0..256) */
/* run `n` updates with symbols syms[] on a fresh CDF of `nbits` (2,3,4); writes cells */
void nlzm_oracle_cdf_run(int nbits, const uint8_t *syms, uint32_t n, uint16_t *cells_out);
/* frame geometry for a (post-shrink) hist_bits */
void nlzm_oracle_geometry(uint64_t flen, uint32_t hist_bits_req, uint32_t *hist_bits,
                          uint32_t *frame_bits, uint32_t *chunk_size, uint32_t *feed_size);
/* rANS-encode one frame from its symbol/bit streams (CodeFrame::Flush, NLZM.cpp:590) */
uint32_t nlzm_oracle_flush_frame(const uint32_t *syms, uint32_t nsyms,
                                 const uint8_t *bits, uint32_t nbits_bytes_payload,
                                 uint32_t num_ops, uint8_t *out, uint32_t out_cap);

#ifdef __cplusplus
}
#endif
#endif
