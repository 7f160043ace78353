/*
 * nlzm_oracle_dump.c -- digest hooks for the CPU restatement.
 *
 * TEST INFRASTRUCTURE ONLY.  Hashes the restatement's intermediate results with
 * the same code (digest.h) that oracle/_ref/nlzm_ref_instr uses on the
 * reference's own, so tests can compare them value for value.
 */
#include "nlzm_oracle.h"
#include "digest.h"

#include <string.h>

static void dump_pos(void *u, uint64_t abs_pos, uint32_t max_len, const uint32_t *delta)
{
    nlzm_digest_pos((nlzm_digest *)u, abs_pos, max_len, delta);
}

static void dump_seg(void *u, uint64_t abs_start, uint32_t seg_len, const nlzm_oracle_cmd *c, uint32_t n)
{
    nlzm_digest *d = (nlzm_digest *)u;
    nlzm_digest_seg_begin(d, abs_start, seg_len);
    for (uint32_t i = 0; i < n; i++) nlzm_digest_seg_cmd(d, c[i].cmd, c[i].len, c[i].delta);
}

static void dump_frame(void *u, uint32_t idx, uint32_t num_ops, const uint32_t *syms, uint32_t nsyms,
                       const uint8_t *bits, uint32_t nbits, const uint8_t *fb, uint32_t flen)
{
    (void)idx;
    /* bits[] already carries the 4 pad bytes: split them back into a pending word */
    const uint32_t payload = nbits - 4;
    const uint32_t word = ((uint32_t)bits[payload] << 24) | ((uint32_t)bits[payload + 1] << 16) |
                          ((uint32_t)bits[payload + 2] << 8) | bits[payload + 3];
    nlzm_digest_frame((nlzm_digest *)u, num_ops, syms, nsyms, bits, payload, word, fb, flen);
}

/* Compress with all three digests; out64 receives {mt, seg, frm, n_pos, n_seg, n_frames};
 * if snaps != NULL it receives up to snap_cap frames x 3 values. */
int nlzm_oracle_compress_digest(const uint8_t *src, uint64_t n, uint32_t hist_bits_req,
                                uint8_t *dst, uint64_t dst_cap, uint64_t *dst_len,
                                nlzm_oracle_stats *stats, uint64_t *out64,
                                uint64_t *snaps, uint32_t snap_cap)
{
    nlzm_digest d;
    nlzm_digest_init(&d);
    nlzm_oracle_hooks hk;
    hk.user = &d; hk.on_position = dump_pos; hk.on_segment = dump_seg; hk.on_frame = dump_frame;
    const int rc = nlzm_oracle_compress(src, n, hist_bits_req, dst, dst_cap, dst_len, stats, &hk);
    out64[0] = d.mt; out64[1] = d.seg; out64[2] = d.frm;
    out64[3] = d.n_pos; out64[4] = d.n_seg; out64[5] = d.n_frames;
    if (snaps) {
        const uint64_t k = d.n_frames < snap_cap ? d.n_frames : snap_cap;
        memcpy(snaps, d.snap, (size_t)k * 3 * sizeof(uint64_t));
    }
    free(d.snap);
    return rc;
}

/* ---- capture of one frame's symbol/bit streams (fixture F4, rANS kernel parity) ---- */

typedef struct {
    uint32_t want;
    uint32_t *syms; uint32_t nsyms, cap_syms;
    uint8_t *bits; uint32_t nbits, cap_bits;
    uint8_t *frame; uint32_t flen, cap_frame;
    uint32_t num_ops;
    int found;
} cap_t;

static void capture_frame_cb(void *u, uint32_t idx, uint32_t num_ops, const uint32_t *syms, uint32_t nsyms,
                      const uint8_t *bits, uint32_t nbits, const uint8_t *fb, uint32_t flen)
{
    cap_t *c = (cap_t *)u;
    if (idx != c->want) return;
    if (nsyms > c->cap_syms || nbits > c->cap_bits || flen > c->cap_frame) { c->found = -1; return; }
    memcpy(c->syms, syms, 4ull * nsyms); c->nsyms = nsyms;
    memcpy(c->bits, bits, nbits); c->nbits = nbits;
    memcpy(c->frame, fb, flen); c->flen = flen;
    c->num_ops = num_ops;
    c->found = 1;
}

/* sizes_out = {nsyms, nbits_bytes (incl. pad), frame_len, num_ops}; returns 0 when captured */
int nlzm_oracle_capture_frame(const uint8_t *src, uint64_t n, uint32_t hist_bits_req, uint32_t frame_idx,
                              uint32_t *syms, uint32_t cap_syms, uint8_t *bits, uint32_t cap_bits,
                              uint8_t *frame, uint32_t cap_frame, uint32_t *sizes_out)
{
    cap_t c; memset(&c, 0, sizeof c);
    c.want = frame_idx; c.syms = syms; c.cap_syms = cap_syms; c.bits = bits; c.cap_bits = cap_bits;
    c.frame = frame; c.cap_frame = cap_frame;
    nlzm_oracle_hooks hk; memset(&hk, 0, sizeof hk);
    hk.user = &c; hk.on_frame = capture_frame_cb;
    const uint64_t cap = nlzm_oracle_bound(n);
    uint8_t *dst = (uint8_t *)malloc(cap);
    uint64_t dl = 0;
    const int rc = nlzm_oracle_compress(src, n, hist_bits_req, dst, cap, &dl, 0, &hk);
    free(dst);
    if (rc) return rc;
    if (c.found != 1) return -10;
    sizes_out[0] = c.nsyms; sizes_out[1] = c.nbits; sizes_out[2] = c.flen; sizes_out[3] = c.num_ops;
    return 0;
}

/* ---- per-position match tables for a position range (fixture F2, finder parity) ---- */

typedef struct {
    uint64_t lo, hi;
    uint32_t *out; uint64_t cap_words, used;    /* records: pos, max_len, delta[2..max_len] */
    int overflow;
} mtcap_t;

static void cap_pos(void *u, uint64_t abs_pos, uint32_t max_len, const uint32_t *delta)
{
    mtcap_t *c = (mtcap_t *)u;
    if (abs_pos < c->lo || abs_pos >= c->hi) return;
    const uint64_t need = 2 + (max_len >= 2 ? max_len - 1 : 0);
    if (c->used + need > c->cap_words) { c->overflow = 1; return; }
    c->out[c->used++] = (uint32_t)abs_pos;
    c->out[c->used++] = max_len;
    for (uint32_t i = 2; i <= max_len; i++) c->out[c->used++] = delta[i];
}

int nlzm_oracle_capture_tables(const uint8_t *src, uint64_t n, uint32_t hist_bits_req,
                               uint64_t pos_lo, uint64_t pos_hi,
                               uint32_t *out_words, uint64_t cap_words, uint64_t *used_words)
{
    mtcap_t c; memset(&c, 0, sizeof c);
    c.lo = pos_lo; c.hi = pos_hi; c.out = out_words; c.cap_words = cap_words;
    nlzm_oracle_hooks hk; memset(&hk, 0, sizeof hk);
    hk.user = &c; hk.on_position = cap_pos;
    const uint64_t cap = nlzm_oracle_bound(n);
    uint8_t *dst = (uint8_t *)malloc(cap);
    uint64_t dl = 0;
    const int rc = nlzm_oracle_compress(src, n, hist_bits_req, dst, cap, &dl, 0, &hk);
    free(dst);
    if (rc) return rc;
    if (c.overflow) return -11;
    *used_words = c.used;
    return 0;
}
