/*
 * digest.h -- FNV-1a digests over the path's intermediate results.
 *
 * TEST INFRASTRUCTURE ONLY.  Shared by (a) the hook prelude that is piped in
 * front of the reference source when oracle/Makefile builds the instrumented
 * oracle/_ref/nlzm_ref_instr, and (b) oracle/nlzm_oracle_dump.c, so that the
 * reference and the restatement are hashed by the very same code:
 *
 *   F2  match tables : (pos, max_len, delta[2..max_len]) per input position,
 *                       state copied into mt_carry at NLZM.cpp:1543
 *   F3  segments     : (start, length, commands...) per parse_table() call
 *   F4  frames       : (num_ops, symbols, bit bytes incl. the 4 pad bytes,
 *                       frame bytes) around CodeFrame::Flush, NLZM.cpp:1850
 *
 * One running 64-bit value per family; a snapshot of all three is recorded at
 * every frame so a divergence can be located to a chunk.
 */
#ifndef NLZM_DIGEST_H
#define NLZM_DIGEST_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct nlzm_digest {
    uint64_t mt, seg, frm;
    uint64_t n_pos, n_seg, n_frames;
    /* per-frame snapshots */
    uint64_t *snap;       /* 3 values per frame */
    uint32_t snap_cap;
} nlzm_digest;

#define NLZM_FNV_INIT 0xcbf29ce484222325ull
#define NLZM_FNV_PRIME 0x100000001b3ull

static inline uint64_t nlzm_fnv_u32(uint64_t h, uint32_t w)
{
    for (int i = 0; i < 4; i++) { h ^= (w >> (8 * i)) & 0xFF; h *= NLZM_FNV_PRIME; }
    return h;
}

static inline uint64_t nlzm_fnv_bytes(uint64_t h, const uint8_t *p, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++) { h ^= p[i]; h *= NLZM_FNV_PRIME; }
    return h;
}

static inline void nlzm_digest_init(nlzm_digest *d)
{
    d->mt = d->seg = d->frm = NLZM_FNV_INIT;
    d->n_pos = d->n_seg = d->n_frames = 0;
    d->snap = 0; d->snap_cap = 0;
}

static inline void nlzm_digest_pos(nlzm_digest *d, uint64_t abs_pos, uint32_t max_len, const uint32_t *delta)
{
    uint64_t h = d->mt;
    h = nlzm_fnv_u32(h, (uint32_t)abs_pos);
    h = nlzm_fnv_u32(h, max_len);
    for (uint32_t i = 2; i <= max_len; i++) h = nlzm_fnv_u32(h, delta[i]);
    d->mt = h;
    d->n_pos++;
}

static inline void nlzm_digest_seg_begin(nlzm_digest *d, uint64_t abs_start, uint32_t seg_len)
{
    d->seg = nlzm_fnv_u32(nlzm_fnv_u32(d->seg, (uint32_t)abs_start), seg_len);
    d->n_seg++;
}

static inline void nlzm_digest_seg_cmd(nlzm_digest *d, uint32_t cmd, uint32_t len, uint32_t delta)
{
    /* a literal's len/delta fields are don't-care in the reference's node */
    d->seg = nlzm_fnv_u32(d->seg, cmd);
    if (cmd != 0) d->seg = nlzm_fnv_u32(nlzm_fnv_u32(d->seg, len), delta);
}

/* bits: payload bytes written so far (without pad); word: pending bit word */
static inline void nlzm_digest_frame(nlzm_digest *d, uint32_t num_ops, const uint32_t *syms, uint32_t nsyms,
                                     const uint8_t *bits, uint32_t nbits_payload, uint32_t word,
                                     const uint8_t *frame_bytes, uint32_t frame_len)
{
    uint64_t h = d->frm;
    h = nlzm_fnv_u32(nlzm_fnv_u32(h, num_ops), nsyms);
    for (uint32_t i = 0; i < nsyms; i++) h = nlzm_fnv_u32(h, syms[i]);
    h = nlzm_fnv_u32(h, nbits_payload + 4);
    h = nlzm_fnv_bytes(h, bits, nbits_payload);
    for (int i = 0; i < 4; i++) { uint8_t b = (uint8_t)(word >> 24); h = nlzm_fnv_bytes(h, &b, 1); word <<= 8; }
    h = nlzm_fnv_u32(h, frame_len);
    h = nlzm_fnv_bytes(h, frame_bytes, frame_len);
    d->frm = h;
    if (d->n_frames >= d->snap_cap) {
        d->snap_cap = d->snap_cap ? d->snap_cap * 2 : 64;
        d->snap = (uint64_t *)realloc(d->snap, (size_t)d->snap_cap * 3 * sizeof(uint64_t));
    }
    d->snap[3 * d->n_frames + 0] = d->mt;
    d->snap[3 * d->n_frames + 1] = d->seg;
    d->snap[3 * d->n_frames + 2] = d->frm;
    d->n_frames++;
}

/* JSON: {"n_pos":..,"n_seg":..,"n_frames":..,"mt":"hex","seg":"hex","frm":"hex","snap":[["..","..",".."],...]} */
static inline void nlzm_digest_write_json(const nlzm_digest *d, FILE *f, int with_snaps)
{
    fprintf(f, "{\"n_pos\": %llu, \"n_seg\": %llu, \"n_frames\": %llu, \"mt\": \"%016llx\", \"seg\": \"%016llx\", \"frm\": \"%016llx\"",
            (unsigned long long)d->n_pos, (unsigned long long)d->n_seg, (unsigned long long)d->n_frames,
            (unsigned long long)d->mt, (unsigned long long)d->seg, (unsigned long long)d->frm);
    if (with_snaps) {
        fprintf(f, ", \"snap\": [");
        for (uint64_t i = 0; i < d->n_frames; i++)
            fprintf(f, "%s[\"%016llx\", \"%016llx\", \"%016llx\"]", i ? ", " : "",
                    (unsigned long long)d->snap[3 * i], (unsigned long long)d->snap[3 * i + 1],
                    (unsigned long long)d->snap[3 * i + 2]);
        fprintf(f, "]");
    }
    fprintf(f, "}\n");
}

#endif
