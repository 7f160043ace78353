/*
 * nlzm_oracle.c -- plain-C CPU restatement of NLZM 1.03 (compress + decompress).
 *
 * TEST INFRASTRUCTURE ONLY (see nlzm_oracle.h).  Every function cites the
 * reference lines (file NLZM.cpp under /root/reference) whose behaviour it
 * restates.  Layout differs from the reference on purpose: the input is one
 * flat buffer addressed by absolute offset instead of a
 * ring + lookahead pointer; "rebased" 32-bit coordinates are kept only where
 * the reference's stored values depend on them (HT/RK tables, BT nodes).
 */
#include "nlzm_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* small helpers                                                             */
/* ------------------------------------------------------------------------- */

static inline uint32_t u32min(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t u32max(uint32_t a, uint32_t b) { return a < b ? b : a; }

/* NLZM.cpp:59-75 -- despite its name the reference's clz32 is floor(log2 x). */
static inline uint32_t ilog2(uint32_t x) { return 31u - (uint32_t)__builtin_clz(x | 1u); }

/* NLZM.cpp:813-821 */
uint32_t nlzm_oracle_match_min(uint32_t dist)
{
    return 2u + (dist >= (1u << 8)) + (dist >= (1u << 12)) + (dist >= (1u << 20));
}

/* NLZM.cpp:739 */
uint32_t nlzm_oracle_hash4(uint32_t x) { return x * 987660757u; }

/* ------------------------------------------------------------------------- */
/* tables: log2 cost LUT (NLZM.cpp:97-124), CDF mixins (NLZM.cpp:284-303)    */
/* ------------------------------------------------------------------------- */

enum { CDF_ADAPT = 7, CDF_BITS = 14, CDF_TOTAL = 1 << CDF_BITS };

static uint16_t g_lut[256];
static int16_t g_mix2[4][4], g_mix3[8][8], g_mix4[16][16];
static uint32_t g_crc[256];
static int g_ready;

static void build_mixin(int16_t *m, int n)
{
    /* NLZM.cpp:284-298: target CDF when symbol y is seen. */
    const int bias = (1 << CDF_ADAPT) - 1 - n;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            m[y * n + x] = (int16_t)(x <= y ? x : CDF_TOTAL + x + bias);
}

void nlzm_oracle_init(void)
{
    if (g_ready) return;
    /* NLZM.cpp:103-124: 32 squarings of i/256 in 16-bit fixed point; the sum of
     * the per-step exponents is -32*log2(i/256). */
    for (uint32_t i = 1; i < 256; i++) {
        uint32_t next = 1u << 16;
        uint16_t acc = 0;
        for (int s = 0; s < 32; s++) {
            const uint32_t v = (i * next) >> 8;
            const uint32_t nbits = 16u - ilog2(v);
            acc = (uint16_t)(acc + nbits - 1);
            next = v << (nbits - 1);
        }
        g_lut[i] = acc;
    }
    g_lut[0] = g_lut[1];
    build_mixin(&g_mix2[0][0], 4);
    build_mixin(&g_mix3[0][0], 8);
    build_mixin(&g_mix4[0][0], 16);
    /* NLZM.cpp:128-145: reflected CRC-32, poly 0xEDB88320 (byte-wise table only) */
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        g_crc[n] = c;
    }
    g_ready = 1;
}

const uint16_t *nlzm_oracle_log2_lut(void) { nlzm_oracle_init(); return g_lut; }

/* NLZM.cpp:157-199 (slicing-by-16 there; same function, byte at a time here) */
uint32_t nlzm_oracle_crc32(const uint8_t *p, uint64_t n, uint32_t crc)
{
    nlzm_oracle_init();
    uint32_t c = crc ^ 0xFFFFFFFFu;
    while (n--) c = g_crc[(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

/* ------------------------------------------------------------------------- */
/* adaptive nibble CDFs (NLZM.cpp:212-438)                                   */
/* A CDF of 2^k symbols is k+... cells: cell[0]=0, cell[2^k]=16384 fixed.    */
/* ------------------------------------------------------------------------- */

typedef struct { uint16_t c[17]; } cdf_t;   /* used for 2-, 3- and 4-bit alphabets */

static void cdf_reset(cdf_t *d, int nsyms)
{
    /* NLZM.cpp:265-282, 330-346: uniform start */
    for (int i = 0; i <= nsyms; i++) d->c[i] = (uint16_t)(i * (CDF_TOTAL / nsyms));
}

static inline void cdf_adapt(cdf_t *d, const int16_t *mix_row, int nsyms)
{
    /* NLZM.cpp:352-382: cell += (mixin - cell) >> 7, arithmetic shift, i < nsyms */
    for (int i = 0; i < nsyms; i++)
        d->c[i] = (uint16_t)(d->c[i] + ((mix_row[i] - (int)d->c[i]) >> CDF_ADAPT));
}

static inline void cdf_adapt2(cdf_t *d, int y) { cdf_adapt(d, g_mix2[y], 4); }
static inline void cdf_adapt3(cdf_t *d, int y) { cdf_adapt(d, g_mix3[y], 8); }
static inline void cdf_adapt4(cdf_t *d, int y) { cdf_adapt(d, g_mix4[y], 16); }

/* NLZM.cpp:435-438 */
static inline uint32_t cdf_price(const cdf_t *d, int y)
{
    return g_lut[(uint32_t)(d->c[y + 1] - d->c[y]) >> (CDF_BITS - 8)];
}

/* NLZM.cpp:388-433 (scalar forms) */
static inline int cdf_find(const cdf_t *d, int nbits, uint32_t f)
{
    int r = 0;
    for (int step = 1 << (nbits - 1); step; step >>= 1)
        r += step * (f >= d->c[r + step]);
    return r;
}

void nlzm_oracle_cdf_run(int nbits, const uint8_t *syms, uint32_t n, uint16_t *cells_out)
{
    nlzm_oracle_init();
    cdf_t d;
    const int ns = 1 << nbits;
    cdf_reset(&d, ns);
    for (uint32_t i = 0; i < n; i++) {
        if (nbits == 2) cdf_adapt2(&d, syms[i]);
        else if (nbits == 3) cdf_adapt3(&d, syms[i]);
        else cdf_adapt4(&d, syms[i]);
    }
    for (int i = 0; i <= ns; i++) cells_out[i] = d.c[i];
}

/* ------------------------------------------------------------------------- */
/* model (NLZM.cpp:1125-1206)                                                */
/* ------------------------------------------------------------------------- */

typedef struct {
    uint32_t rep[4];
    cdf_t cmd;                    /* CDF2 */
    cdf_t lit_hi, lit_lo[16];     /* CDF4 */
    cdf_t len_direct;             /* CDF3 */
    cdf_t len_ext_hi, len_ext_lo[16]; /* CDF4 */
    cdf_t slot_hi[4];             /* CDF3 */
    cdf_t slot_lo[4][8];          /* CDF3 */
} model_t;

static void model_reset(model_t *m)
{
    for (int i = 0; i < 4; i++) m->rep[i] = (uint32_t)i + 1;   /* NLZM.cpp:1154-1158 */
    cdf_reset(&m->cmd, 4);
    cdf_reset(&m->lit_hi, 16);
    cdf_reset(&m->len_direct, 8);
    cdf_reset(&m->len_ext_hi, 16);
    for (int i = 0; i < 16; i++) { cdf_reset(&m->lit_lo[i], 16); cdf_reset(&m->len_ext_lo[i], 16); }
    for (int c = 0; c < 4; c++) {
        cdf_reset(&m->slot_hi[c], 8);
        for (int i = 0; i < 8; i++) cdf_reset(&m->slot_lo[c][i], 8);
    }
}

/* NLZM.cpp:1160-1171: push front only when absent, no move-to-front */
static inline void rep_add(uint32_t r[4], uint32_t d)
{
    if (r[0] == d || r[1] == d || r[2] == d || r[3] == d) return;
    r[3] = r[2]; r[2] = r[1]; r[1] = r[0]; r[0] = d;
}

/* NLZM.cpp:1173-1181 */
static inline int rep_find(const uint32_t r[4], uint32_t d)
{
    for (int i = 0; i < 4; i++) if (r[i] == d) return i;
    return -1;
}

/* distance -> (slot, extra bit count, extra bits)   NLZM.cpp:1227-1243, 1299-1320 */
static inline uint32_t dist_slot(uint32_t dv, uint32_t *nextra, uint32_t *extra)
{
    if (dv < 4) { *nextra = 0; *extra = 0; return dv; }
    const uint32_t nb = ilog2(dv) + 1;
    const uint32_t ab = nb - 2;
    *nextra = ab;
    *extra = dv & ((1u << ab) - 1);
    return ((nb - 1) << 1) + ((dv >> ab) & 1);
}

/* shared length part of match/rep price  NLZM.cpp:1214-1225 / 1256-1267 */
static inline uint32_t price_len(const model_t *m, uint32_t lv)
{
    uint32_t c = cdf_price(&m->len_direct, (int)u32min(lv, 7));
    if (lv >= 7) {
        lv -= 7;
        c += cdf_price(&m->len_ext_hi, (int)(lv >> 4));
        c += cdf_price(&m->len_ext_lo[lv >> 4], (int)(lv & 15));
    }
    return c;
}

/* NLZM.cpp:1208-1251 */
static uint32_t price_match(const model_t *m, uint32_t delta, uint32_t len)
{
    const uint32_t lv = len - nlzm_oracle_match_min(delta);
    const uint32_t lc = u32min(lv, 3);
    uint32_t nx, ex;
    const uint32_t slot = dist_slot(delta - 1, &nx, &ex);
    return cdf_price(&m->cmd, 1) + price_len(m, lv) + (nx << 5)
         + cdf_price(&m->slot_hi[lc], (int)(slot >> 3))
         + cdf_price(&m->slot_lo[lc][slot >> 3], (int)(slot & 7));
}

/* NLZM.cpp:1253-1272 */
static uint32_t price_rep(const model_t *m, uint32_t delta, uint32_t len)
{
    return cdf_price(&m->cmd, 2) + price_len(m, len - nlzm_oracle_match_min(delta)) + (2u << 5);
}

/* NLZM.cpp:1418-1426 */
static uint32_t price_literal(const model_t *m, int y)
{
    return cdf_price(&m->cmd, 0) + cdf_price(&m->lit_hi, y >> 4) + cdf_price(&m->lit_lo[y >> 4], y & 15);
}

/* ------------------------------------------------------------------------- */
/* frame writer: symbol + raw-bit buffers, 4-way rANS flush                  */
/* (CodeFrame, NLZM.cpp:490-513, 534-640; rANS NLZM.cpp:440-469)             */
/* ------------------------------------------------------------------------- */

typedef struct {
    uint32_t *syms;  uint32_t nsyms, cap_syms;      /* (freq<<16)+start  NLZM.cpp:565 */
    uint8_t *bits;   uint32_t nbits_bytes, cap_bits;
    uint32_t word, word_bits, num_ops;
} frame_t;

static inline void frame_sym(frame_t *f, const cdf_t *d, int y)
{
    /* NLZM.cpp:559-572 */
    const uint32_t start = d->c[y], freq = (uint32_t)(d->c[y + 1] - d->c[y]);
    f->num_ops++;
    f->syms[f->nsyms++] = (freq << 16) + start;
}

static inline void frame_bits(frame_t *f, uint32_t v, uint32_t nb)
{
    /* NLZM.cpp:574-588: MSB-first; one op per call regardless of nb */
    f->num_ops++;
    f->word |= v << (32 - f->word_bits - nb);
    f->word_bits += nb;
    while (f->word_bits >= 8) {
        f->bits[f->nbits_bytes++] = (uint8_t)(f->word >> 24);
        f->word <<= 8;
        f->word_bits -= 8;
    }
}

static inline void frame_pad_bits(frame_t *f)
{
    /* NLZM.cpp:591-597: always four bytes */
    for (int i = 0; i < 4; i++) {
        f->bits[f->nbits_bytes++] = (uint8_t)(f->word >> 24);
        f->word <<= 8;
    }
    f->word_bits = 0;
}

static inline void put_be32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}

/* CodeFrame::Flush, NLZM.cpp:590-640.  `bits` already holds the 4 pad bytes.
 * Layout: [num_ops][12+nbits][nrans] | bits | st0..st3 LE | renorm words.   */
uint32_t nlzm_oracle_flush_frame(const uint32_t *syms, uint32_t nsyms,
                                 const uint8_t *bits, uint32_t nbits_bytes,
                                 uint32_t num_ops, uint8_t *out, uint32_t out_cap)
{
    /* worst case 2 renorm bytes per symbol + 16 state bytes */
    const uint64_t need = 12ull + nbits_bytes + 16ull + 2ull * nsyms;
    uint8_t *tmp = (uint8_t *)malloc((size_t)(2ull * nsyms + 16));
    if (!tmp) return 0;
    uint8_t *w = tmp + 2ull * nsyms + 16;     /* grows downwards */
    uint32_t st[4] = { 1u << 16, 1u << 16, 1u << 16, 1u << 16 };
    for (uint32_t i = nsyms; i-- > 0;) {
        /* rans_enc_put, NLZM.cpp:444-455 */
        const uint32_t start = syms[i] & 0xFFFF, freq = syms[i] >> 16;
        uint32_t x = st[i & 3];
        const uint32_t x_max = ((1u << 16 >> CDF_BITS) << 16) * freq;
        if (x >= x_max) {
            *--w = (uint8_t)x;
            *--w = (uint8_t)(x >> 8);
            x >>= 16;
        }
        st[i & 3] = ((x / freq) << CDF_BITS) + (x % freq) + start;
    }
    for (int k = 3; k >= 0; k--) {           /* NLZM.cpp:461-469, 605-608 */
        w -= 4;
        w[0] = (uint8_t)st[k]; w[1] = (uint8_t)(st[k] >> 8);
        w[2] = (uint8_t)(st[k] >> 16); w[3] = (uint8_t)(st[k] >> 24);
    }
    const uint32_t nrans = (uint32_t)((tmp + 2ull * nsyms + 16) - w);
    uint32_t total = 0;
    if (need <= out_cap || 12ull + nbits_bytes + nrans <= out_cap) {
        put_be32(out, num_ops);
        put_be32(out + 4, 12 + nbits_bytes);
        put_be32(out + 8, nrans);
        memcpy(out + 12, bits, nbits_bytes);
        memcpy(out + 12 + nbits_bytes, w, nrans);
        total = 12 + nbits_bytes + nrans;
    }
    free(tmp);
    return total;
}

/* symbol emitters ------------------------------------------------------- */

/* NLZM.cpp:1428-1439 */
static void emit_literal(frame_t *f, model_t *m, int y)
{
    const int hi = y >> 4, lo = y & 15;
    frame_sym(f, &m->cmd, 0);
    frame_sym(f, &m->lit_hi, hi);
    frame_sym(f, &m->lit_lo[hi], lo);
    cdf_adapt2(&m->cmd, 0);
    cdf_adapt4(&m->lit_hi, hi);
    cdf_adapt4(&m->lit_lo[hi], lo);
}

/* length symbols common to match and rep: NLZM.cpp:1281-1297 / 1348-1364 */
static uint32_t emit_len(frame_t *f, model_t *m, uint32_t lv)
{
    const int d = (int)u32min(lv, 7);
    frame_sym(f, &m->len_direct, d);
    cdf_adapt3(&m->len_direct, d);
    if (lv >= 7) {
        const uint32_t e = lv - 7;
        const int hi = (int)(e >> 4), lo = (int)(e & 15);
        frame_sym(f, &m->len_ext_hi, hi);
        frame_sym(f, &m->len_ext_lo[hi], lo);
        cdf_adapt4(&m->len_ext_hi, hi);
        cdf_adapt4(&m->len_ext_lo[hi], lo);
    }
    return u32min(lv, 3);
}

/* NLZM.cpp:1274-1342 */
static void emit_match(frame_t *f, model_t *m, uint32_t delta, uint32_t len)
{
    frame_sym(f, &m->cmd, 1);
    cdf_adapt2(&m->cmd, 1);
    const uint32_t lc = emit_len(f, m, len - nlzm_oracle_match_min(delta));
    uint32_t nx, ex;
    const uint32_t slot = dist_slot(delta - 1, &nx, &ex);
    const int hi = (int)(slot >> 3), lo = (int)(slot & 7);
    frame_sym(f, &m->slot_hi[lc], hi);
    frame_sym(f, &m->slot_lo[lc][hi], lo);
    cdf_adapt3(&m->slot_hi[lc], hi);
    cdf_adapt3(&m->slot_lo[lc][hi], lo);
    if (delta - 1 >= 4) {
        if (nx < 4) {
            frame_bits(f, ex, nx);           /* NLZM.cpp:1328-1330 (nx may be 0: still one op) */
        } else {
            if (nx > 4) frame_bits(f, ex >> 4, nx - 4);
            frame_bits(f, ex & 15, 4);
        }
    }
}

/* NLZM.cpp:1344-1367 */
static void emit_rep(frame_t *f, model_t *m, uint32_t idx, uint32_t len)
{
    frame_sym(f, &m->cmd, 2);
    cdf_adapt2(&m->cmd, 2);
    emit_len(f, m, len - nlzm_oracle_match_min(m->rep[idx]));
    frame_bits(f, idx, 2);
}

/* ------------------------------------------------------------------------- */
/* encoder context: flat input + rebased coordinates                         */
/* ------------------------------------------------------------------------- */

typedef struct {
    uint32_t max_len;
    uint32_t delta[NLZM_MATCH_MAX + 1];
} mtab_t;

typedef struct {
    uint32_t hash_bits, nrows, shift, tag_mask;
    uint32_t *rows;
} ht_t;

typedef struct {
    uint32_t shift;
    uint32_t *heads, *tree;
} bt_t;

typedef struct {
    uint32_t shift, tag_mask;
    uint32_t *table;
    uint32_t carry_from, carry_to, carry_len;
    uint32_t rh, rh_end;
} rk_t;

typedef struct {
    const uint8_t *in;          /* whole input */
    uint64_t n;
    uint32_t wbits, wmask;      /* window */
    uint64_t base;              /* absolute offset of rebased position 0 (k*W) */
    uint32_t la_end;            /* rebased end of the current chunk's lookahead */
    ht_t ht2, ht3;
    bt_t bt;
    rk_t rk;
    nlzm_oracle_stats *st;
} enc_t;

static inline uint8_t at(const enc_t *e, uint32_t reb) { return e->in[e->base + reb]; }

/* RingDictionary::MatchLengthSigned, NLZM.cpp:854-877.  max_len and the
 * running length are uint16 there; callers pass already-truncated values. */
static uint32_t common_len_signed(const enc_t *e, uint32_t p0, uint32_t p1, uint32_t max_len, uint32_t init)
{
    const uint8_t *a = e->in + e->base + p0, *b = e->in + e->base + p1;
    uint32_t l = init;
    while (l < max_len) {
        e->st->cmp_bytes++;
        if (a[l] != b[l]) return l | ((uint32_t)(a[l] < b[l]) << 31);
        l++;
    }
    return l;
}

/* MatchTable::Update, NLZM.cpp:835-852 (element-wise min-merge) */
static void mt_update(mtab_t *t, uint32_t d, uint32_t len)
{
    uint32_t i = 0;
    for (; i <= len && i <= t->max_len; i++) t->delta[i] = u32min(t->delta[i], d);
    for (; i <= len; i++) t->delta[i] = d;
    t->max_len = u32max(t->max_len, len);
}

/* ---- HT2 / HT3 : MatchFinderHT, NLZM.cpp:893-957 ------------------------ */

static void ht_init(ht_t *h, uint32_t hash_bits, uint32_t nrows, uint32_t wbits)
{
    h->hash_bits = hash_bits; h->nrows = nrows;
    h->shift = 32 - hash_bits;
    h->tag_mask = (uint32_t)((1ull << (32 - wbits)) - 1);
    const size_t cnt = (size_t)nrows << hash_bits;
    h->rows = (uint32_t *)malloc(cnt * 4);
    memset(h->rows, 0xFF, cnt * 4);
}

static void ht_find(enc_t *e, ht_t *h, mtab_t *mt, uint32_t hv, uint32_t p)
{
    const uint32_t tag = hv & h->tag_mask;
    uint32_t *row = h->rows + (hv >> h->shift);        /* NLZM.cpp:912: not scaled by nrows */
    uint32_t carry = p | (tag << e->wbits);            /* p is NOT masked (NLZM.cpp:913) */
    const uint32_t max_len = u32min(e->la_end - p, NLZM_MATCH_MAX);
    uint32_t best = NLZM_MATCH_MIN - 1;
    for (uint32_t i = 0; i < h->nrows; i++) {
        const uint32_t v = row[i];
        e->st->ht_rows++;
        if (best < max_len && (v >> e->wbits) == tag) {
            const uint32_t sp = v & e->wmask;
            if (sp < p && p - sp <= e->wmask) {
                const uint32_t l = common_len_signed(e, sp, p, max_len, 0) & 0x7FFFFFFFu;
                if (l > best && l >= nlzm_oracle_match_min(p - sp)) {
                    mt_update(mt, p - sp, l);
                    best = l;
                }
            }
        }
        row[i] = carry;                                 /* NLZM.cpp:935-936 */
        carry = v;
    }
}

/* ---- BT4 : MatchFinderBT, NLZM.cpp:959-1031 ----------------------------- */

/* (debug, for oracle/skip_probe.c: the longest match the last bt_find call handed to the table, 0: none) */
static uint32_t g_bt_last_best;
uint32_t nlzm_oracle_debug_bt_last_best(void) { return g_bt_last_best; }

static void bt_find(enc_t *e, mtab_t *mt, uint32_t h4, uint32_t p)
{
    bt_t *b = &e->bt;
    g_bt_last_best = 0;
    uint32_t *pend_l = b->tree + ((size_t)(p & e->wmask) << 1);
    uint32_t *pend_r = pend_l + 1;
    uint32_t len_l = 0, len_r = 0;
    uint32_t sp = b->heads[h4 >> b->shift];
    b->heads[h4 >> b->shift] = p;
    const uint32_t max_len = u32min(e->la_end - p, NLZM_MATCH_MAX);
    uint16_t tests = 256;                               /* NLZM.cpp:777, 988 */
    e->st->bt_calls++;
    while (sp != 0xFFFFFFFFu && p > sp && p - sp <= e->wmask && tests-- > 0) {
        e->st->bt_tests++;
        uint32_t *pair = b->tree + ((size_t)(sp & e->wmask) << 1);
        const uint32_t r = common_len_signed(e, sp, p, max_len, u32min(len_l, len_r));
        const uint32_t l = r & 0x7FFFFFFFu;
        if (l >= nlzm_oracle_match_min(p - sp)) { mt_update(mt, p - sp, l); if (l > g_bt_last_best) g_bt_last_best = l; }
        if (l == max_len) {                             /* NLZM.cpp:1000-1004 */
            *pend_l = pair[0];
            *pend_r = pair[1];
            return;
        }
        if (r >> 31) {                                  /* candidate sorts below p */
            *pend_l = sp; pend_l = pair + 1; sp = *pend_l; len_r = l;
        } else {
            *pend_r = sp; pend_r = pair; sp = *pend_r; len_l = l;
        }
    }
    *pend_r = 0xFFFFFFFFu;                              /* NLZM.cpp:1020-1021 */
    *pend_l = 0xFFFFFFFFu;
}

/* ---- RK256 : MatchFinderRK256, NLZM.cpp:788-811, 1033-1123 --------------- */

#define RK_ADDH 0x2F0FD693u
#define RK_REMH 0x0E4EA401u     /* ADDH^256 mod 2^32 */

uint32_t nlzm_oracle_rk_hash256(const uint8_t *w)
{
    uint32_t h = 0;
    for (int i = 0; i < 256; i++) h = (h + w[i]) * RK_ADDH;   /* NLZM.cpp:798 */
    return h;
}

/* (test diagnostics, not part of the algorithm: how often the uint16 cap of NLZM.cpp:760 -- not a mismatch, not the
 *  lookahead, not the maximum length -- ended the compare of a match that was taken and is the table's longest entry.  Such
 *  an entry grows again at the next position, :1503-1512; the generator corpus.u16_cut exists to make it happen.) */
static uint64_t g_rk_u16_cuts;
uint64_t nlzm_oracle_debug_rk_u16_cuts(void) { return g_rk_u16_cuts; }

static void rk_find(enc_t *e, mtab_t *mt, uint32_t p)
{
    rk_t *r = &e->rk;
    if (r->carry_len > 0) {                             /* NLZM.cpp:1056-1069 */
        if (p - r->carry_to < r->carry_len) {
            const uint32_t d = r->carry_to - r->carry_from;
            const uint32_t l = r->carry_len - (p - r->carry_to);
            if (l >= nlzm_oracle_match_min(d)) mt_update(mt, d, u32min(l, NLZM_MATCH_MAX));
        } else {
            r->carry_len = 0;
        }
    }
    /* NLZM.cpp:1071-1088: roll the window end up to p+256; a table insert at
     * every 256-aligned end passed on the way stores the CALLING p. */
    while (e->la_end >= p + 256 && r->rh_end < p + 256) {
        const uint32_t c0 = at(e, r->rh_end);
        if (r->rh_end >= 256) r->rh = (c0 + r->rh - at(e, r->rh_end - 256) * RK_REMH) * RK_ADDH;
        else r->rh = (c0 + r->rh) * RK_ADDH;
        r->rh_end++;
        if (!(r->rh_end & 255) && r->rh_end < p + 256) {
            r->table[r->rh >> r->shift] = p | (r->rh << e->wbits);
            e->st->rk_inserts++;
        }
    }
    if (r->carry_len < 256) {                           /* NLZM.cpp:1090-1107 */
        const uint32_t v = r->table[r->rh >> r->shift];
        const uint32_t sp = v & e->wmask;
        e->st->rk_probes++;
        if ((v >> e->wbits) == (r->rh & r->tag_mask) && sp < p && p - sp <= e->wmask) {
            const uint32_t cap = (uint16_t)(e->la_end - p);    /* uint16 parameter, NLZM.cpp:760 */
            const uint32_t l = common_len_signed(e, sp, p, cap, 0) & 0x7FFFFFFFu;
            if (l >= r->carry_len && l >= nlzm_oracle_match_min(p - sp)) {
                mt_update(mt, p - sp, u32min(l, NLZM_MATCH_MAX));
                r->carry_from = sp; r->carry_to = p; r->carry_len = l;
                if (l == cap && l < u32min(e->la_end - p, NLZM_MATCH_MAX) && mt->max_len == l) g_rk_u16_cuts++;
            }
        }
    }
    if (!(r->rh_end & 255) && r->rh_end == p + 256) {   /* NLZM.cpp:1109-1112 */
        r->table[r->rh >> r->shift] = p | (r->rh << e->wbits);
        e->st->rk_inserts++;
    }
}

/* window rebase, NLZM.cpp:1786-1792 with each finder's Shift */
static void rebase(enc_t *e)
{
    const uint32_t W = e->wmask + 1;
    e->base += W;
    e->st->shifts++;
    /* MatchFinderHT::Shift (NLZM.cpp:940-957) iterates with `row` but reads and
     * writes `*rows`; `*rows & window_mask` is always < shift (= W), so every
     * iteration takes the else branch: the only effect is rows[0] = 0xFFFFFFFF. */
    e->ht2.rows[0] = 0xFFFFFFFFu;
    e->ht3.rows[0] = 0xFFFFFFFFu;
    /* MatchFinderBT::Shift, NLZM.cpp:1024-1031 */
    {
        const size_t nh = (size_t)1 << (32 - e->bt.shift);
        for (size_t i = 0; i < nh; i++) {
            uint32_t v = e->bt.heads[i];
            e->bt.heads[i] = (v >= W && v != 0xFFFFFFFFu) ? v - W : 0xFFFFFFFFu;
        }
        const size_t nt = (size_t)2 << e->wbits;
        for (size_t i = 0; i < nt; i++) {
            uint32_t v = e->bt.tree[i];
            e->bt.tree[i] = (v >= W && v != 0xFFFFFFFFu) ? v - W : 0xFFFFFFFFu;
        }
    }
    /* MatchFinderRK256::Shift, NLZM.cpp:1115-1123 */
    if (e->rk.rh_end >= W) e->rk.rh_end -= W; else { e->rk.rh = 0; e->rk.rh_end = 0; }
}

/* ------------------------------------------------------------------------- */
/* forward-graph optimal parse of one segment (parse_table, NLZM.cpp:1464-1651) */
/* ------------------------------------------------------------------------- */

typedef struct { uint32_t cost, delta; uint16_t from, len; uint8_t cmd; } node_t;

typedef struct {
    node_t node[NLZM_PARSE_MAX + 1];
    uint32_t reps[512][4];          /* CarriedState ring, NLZM.cpp:1460-1467 */
    nlzm_oracle_cmd cmds[NLZM_PARSE_MAX];
    uint32_t ncmds;
} parse_t;

static inline void open_nodes(parse_t *ps, uint32_t *end_p, uint32_t upto)
{
    while (*end_p < upto) {                             /* NLZM.cpp:1550-1554 */
        ++*end_p;
        ps->node[*end_p].cost = 0xFFFFFFFFu;
        ps->node[*end_p].from = 0xFFFF;
    }
}

static inline void relax(parse_t *ps, uint32_t p, uint32_t np, uint32_t cost, uint8_t cmd,
                         uint32_t len, uint32_t store_delta, uint32_t add_delta)
{
    /* strict > : the first candidate in program order wins ties */
    if (ps->node[np].cost > ps->node[p].cost + cost) {
        node_t *t = &ps->node[np];
        t->cost = ps->node[p].cost + cost;
        t->cmd = cmd; t->from = (uint16_t)p; t->len = (uint16_t)len; t->delta = store_delta;
        memcpy(ps->reps[np & 511], ps->reps[p & 511], sizeof ps->reps[0]);
        rep_add(ps->reps[np & 511], add_delta);
    }
}

/* seg = rebased position of parse-relative 0.  Returns end_p. */
static uint32_t parse_segment(enc_t *e, parse_t *ps, const model_t *m, mtab_t *carry,
                              uint32_t seg, uint32_t max_parse, const nlzm_oracle_hooks *hk)
{
    max_parse = u32min(max_parse, NLZM_PARSE_MAX);
    ps->node[0].cost = 0; ps->node[0].from = 0xFFFF; ps->node[0].len = 0; ps->node[0].cmd = 0xFF;
    memcpy(ps->reps[0], m->rep, sizeof ps->reps[0]);
    ps->node[1].cost = 0xFFFFFFFFu; ps->node[1].cmd = 0; ps->node[1].len = 0; ps->node[1].from = 0;
    memcpy(ps->reps[1], ps->reps[0], sizeof ps->reps[0]);

    mtab_t mt;
    uint32_t p = 0, end_p = 1;
    while (p < end_p) {
        const uint32_t q = seg + p;                     /* rebased position */
        const uint8_t *cur = e->in + e->base + q;
        e->st->positions++;

        /* literal edge, NLZM.cpp:1490-1499 */
        {
            const uint32_t c = price_literal(m, cur[0]);
            if (ps->node[p + 1].cost > ps->node[p].cost + c) {
                node_t *t = &ps->node[p + 1];
                t->cost = ps->node[p].cost + c; t->cmd = 0; t->from = (uint16_t)p; t->len = 0;
                memcpy(ps->reps[(p + 1) & 511], ps->reps[p & 511], sizeof ps->reps[0]);
            }
        }

        /* carry previous table by one and extend its longest entry, NLZM.cpp:1501-1512 */
        if (carry->max_len <= 1) {
            mt.max_len = 0;
        } else {
            mt.max_len = carry->max_len - 1;
            memcpy(mt.delta, carry->delta + 1, (mt.max_len + 1) * sizeof(uint32_t));
        }
        if (mt.max_len > 0 && q >= mt.delta[mt.max_len]) {
            const uint32_t d = mt.delta[mt.max_len];
            while (mt.max_len < NLZM_MATCH_MAX && e->la_end - q > mt.max_len &&
                   cur[mt.max_len] == cur[(int64_t)mt.max_len - (int64_t)d]) {
                ++mt.max_len;
                mt.delta[mt.max_len] = d;
            }
        }

        /* finders, NLZM.cpp:1514-1541 */
        const int nice = mt.max_len >= NLZM_NICE_LEN;
        if (nice) e->st->nice_positions++;
        if (!nice || !(p & 7)) {
            if (e->la_end - q >= 4) {
                uint32_t v4; memcpy(&v4, cur, 4);       /* little-endian load, NLZM.cpp:740-742 */
                ht_find(e, &e->ht2, &mt, nlzm_oracle_hash4(v4 & 0xFFFFu), q);
                ht_find(e, &e->ht3, &mt, nlzm_oracle_hash4(v4 & 0xFFFFFFu), q);
                if (!nice) bt_find(e, &mt, nlzm_oracle_hash4(v4), q);
            }
            if (e->la_end - q >= 256) rk_find(e, &mt, q);
        }
        *carry = mt;                                    /* NLZM.cpp:1543 */
        if (hk && hk->on_position) hk->on_position(hk->user, e->base + q, mt.max_len, mt.delta);

        uint32_t max_len = u32min(mt.max_len, max_parse - p);
        if (max_len < NLZM_MATCH_MIN) max_len = 0;
        open_nodes(ps, &end_p, max_len + p);

        uint32_t checked = 0;
        /* sampled lengths: uint16 arithmetic in the reference, NLZM.cpp:1558-1560 */
        uint16_t step = (uint16_t)((uint16_t)(max_len - NLZM_MATCH_MIN) >> 4);
        step = (uint16_t)(step + (step == 0));
        for (uint16_t tl = (uint16_t)max_len; tl >= NLZM_MATCH_MIN; tl = (uint16_t)(tl - (tl < step ? tl : step))) {
            const uint32_t d = mt.delta[tl];
            if (tl < nlzm_oracle_match_min(d)) continue;
            const uint32_t np = p + tl;
            relax(ps, p, np, price_match(m, d, tl), 1, tl, d, d);          /* NLZM.cpp:1567-1577 */
            const int ri = rep_find(ps->reps[p & 511], d);
            if (ri < 0) continue;
            checked |= 1u << ri;
            relax(ps, p, np, price_rep(m, d, tl), 2, tl, (uint32_t)ri, d); /* NLZM.cpp:1585-1595 */
        }

        if (checked != 15) {                            /* NLZM.cpp:1598-1628 */
            uint32_t rp[4];
            memcpy(rp, ps->reps[p & 511], sizeof rp);   /* the ring entry itself may be rewritten below */
            for (uint32_t ri = 0; ri < 4; ri++) {
                /* the reference reads rep4.table[ri] through a reference into the
                 * ring; entry p is never a relax target (np > p), so a copy is equal */
                if ((checked >> ri) & 1 || rp[ri] >= q) continue;
                const uint32_t pcap = (uint16_t)(max_parse - p);
                uint32_t l = common_len_signed(e, q - rp[ri], q, pcap, 0) & 0x7FFFFFFFu;
                {   /* bytes compared beyond the 264 that can matter */
                    const uint32_t ncap = u32min(pcap, NLZM_MATCH_MAX), nl = u32min(l, NLZM_MATCH_MAX);
                    e->st->cmp_bytes_needed -= (uint64_t)(l + (l < pcap)) - (uint64_t)(nl + (nl < ncap));   /* (cmp_bytes is added at the end) */
                }
                l = u32min(l, NLZM_MATCH_MAX);
                if (l >= nlzm_oracle_match_min(rp[ri])) {
                    if (end_p < l + p) e->st->seg_rep_grow++;
                    open_nodes(ps, &end_p, l + p);
                    relax(ps, p, p + l, price_rep(m, rp[ri], l), 2, l, ri, rp[ri]);
                }
            }
        }
        ++p;
    }

    /* backtrack (NLZM.cpp:1633-1650 reverses links in place; a list is equivalent) */
    uint32_t n = 0, cur = p;
    while (cur != 0) {
        const node_t *t = &ps->node[cur];
        ps->cmds[n].cmd = t->cmd; ps->cmds[n].len = t->len; ps->cmds[n].delta = t->delta;
        n++;
        cur = t->from;
    }
    for (uint32_t i = 0; i < n / 2; i++) {
        nlzm_oracle_cmd tmp = ps->cmds[i]; ps->cmds[i] = ps->cmds[n - 1 - i]; ps->cmds[n - 1 - i] = tmp;
    }
    ps->ncmds = n;
    e->st->segments++;
    return end_p;
}

/* ------------------------------------------------------------------------- */
/* whole-stream encoder (encode_file, NLZM.cpp:1711-1910)                    */
/* ------------------------------------------------------------------------- */

static inline uint32_t clampu(uint32_t v, uint32_t lo, uint32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

void nlzm_oracle_geometry(uint64_t flen, uint32_t hist_bits_req, uint32_t *hist_bits,
                          uint32_t *frame_bits, uint32_t *chunk_size, uint32_t *feed_size)
{
    uint32_t hb = hist_bits_req;
    while (hb > 10 && flen < (1ull << (hb - 1))) --hb;              /* NLZM.cpp:1716-1718 */
    const uint32_t fb = clampu(hb - 2, 14, 17);                      /* NLZM.cpp:1722 */
    const uint32_t fs = 1u << fb;
    const uint32_t cs = ((fs * 15) / 16) - 0x200;                    /* NLZM.cpp:1724 */
    if (hist_bits) *hist_bits = hb;
    if (frame_bits) *frame_bits = fb;
    if (chunk_size) *chunk_size = cs;
    if (feed_size) *feed_size = cs + NLZM_MATCH_MAX + 1;             /* NLZM.cpp:1725 */
}

uint64_t nlzm_oracle_bound(uint64_t n)
{
    /* each frame holds <= 122,368 input bytes in a 128 KiB buffer */
    return 16 + (n / 14848 + 2) * 131072ull;
}

int nlzm_oracle_compress(const uint8_t *src, uint64_t n, uint32_t hist_bits_req,
                         uint8_t *dst, uint64_t dst_cap, uint64_t *dst_len,
                         nlzm_oracle_stats *stats, const nlzm_oracle_hooks *hooks)
{
    nlzm_oracle_init();
    nlzm_oracle_stats local;
    if (!stats) stats = &local;
    memset(stats, 0, sizeof *stats);
    if (n >= 0xFFFF0000ull) return -2;      /* 32-bit absolute positions only */

    uint32_t hb, fb, chunk_size, feed;
    nlzm_oracle_geometry(n, hist_bits_req, &hb, &fb, &chunk_size, &feed);
    const uint32_t W = 1u << hb, frame_size = 1u << fb;

    enc_t e;
    memset(&e, 0, sizeof e);
    e.in = src; e.n = n; e.wbits = hb; e.wmask = W - 1; e.base = 0; e.st = stats;
    ht_init(&e.ht2, 12, 1, hb);                                      /* NLZM.cpp:1750 */
    ht_init(&e.ht3, 12 + clampu(hb, 15, 17) - 15, 2, hb);            /* NLZM.cpp:1751 */
    {
        const uint32_t bits = 13 + clampu(hb, 16, 20) - 16;          /* NLZM.cpp:1752 */
        e.bt.shift = 32 - bits;
        e.bt.heads = (uint32_t *)malloc((size_t)4 << bits);
        e.bt.tree = (uint32_t *)malloc((size_t)8 << hb);
        memset(e.bt.heads, 0xFF, (size_t)4 << bits);
        memset(e.bt.tree, 0xFF, (size_t)8 << hb);
    }
    {
        const uint32_t bits = 15 + clampu(hb, 16, 22) - 16;          /* NLZM.cpp:1753 */
        e.rk.shift = 32 - bits;
        e.rk.tag_mask = (uint32_t)((1ull << (32 - hb)) - 1);
        e.rk.table = (uint32_t *)malloc((size_t)4 << bits);
        memset(e.rk.table, 0xFF, (size_t)4 << bits);
    }

    model_t *model = (model_t *)malloc(sizeof *model);
    parse_t *ps = (parse_t *)malloc(sizeof *ps);
    model_reset(model);
    mtab_t carry; carry.max_len = 0;

    frame_t fr;
    fr.cap_syms = 4 * frame_size; fr.cap_bits = frame_size;
    fr.syms = (uint32_t *)malloc((size_t)fr.cap_syms * 4);
    fr.bits = (uint8_t *)malloc(fr.cap_bits);
    uint8_t *fbuf = (uint8_t *)malloc(frame_size + 64);

    int rc = 0;
    uint64_t out = 0;
    if (dst_cap < 8) { rc = -1; goto done; }
    dst[out++] = (uint8_t)(hb >> 8); dst[out++] = (uint8_t)hb;       /* NLZM.cpp:1762-1766 */
    dst[out++] = (uint8_t)(fb >> 8); dst[out++] = (uint8_t)fb;

    uint64_t chunk_abs = 0;
    uint32_t frame_idx = 0;
    while (chunk_abs < n) {
        const uint32_t chunk_read = (uint32_t)((n - chunk_abs < feed) ? (n - chunk_abs) : feed);
        const uint32_t p_end = u32min(chunk_size, chunk_read);
        fr.nsyms = 0; fr.nbits_bytes = 0; fr.word = 0; fr.word_bits = 0; fr.num_ops = 0;

        if (chunk_abs - e.base >= 2ull * W) rebase(&e);              /* NLZM.cpp:1786 */
        const uint32_t chunk_reb = (uint32_t)(chunk_abs - e.base);
        e.la_end = chunk_reb + chunk_read;

        uint32_t p = 0;
        while (p < p_end) {
            const uint32_t seg_len = parse_segment(&e, ps, model, &carry, chunk_reb + p, p_end - p, hooks);
            if (hooks && hooks->on_segment)
                hooks->on_segment(hooks->user, chunk_abs + p, seg_len, ps->cmds, ps->ncmds);
            for (uint32_t i = 0; i < ps->ncmds; i++) {
                const nlzm_oracle_cmd c = ps->cmds[i];
                if (c.cmd == 0) {                                    /* NLZM.cpp:1810-1814 */
                    emit_literal(&fr, model, src[chunk_abs + p]);
                    p += 1; stats->n_literal++;
                } else if (c.cmd == 1) {                             /* NLZM.cpp:1815-1828 */
                    emit_match(&fr, model, c.delta, c.len);
                    rep_add(model->rep, c.delta);
                    p += c.len; stats->n_dict++;
                } else {                                             /* NLZM.cpp:1829-1843 */
                    emit_rep(&fr, model, c.delta, c.len);
                    rep_add(model->rep, model->rep[c.delta]);        /* present already: no change */
                    p += c.len; stats->n_rep++;
                }
            }
        }

        stats->rans_syms += fr.nsyms;
        stats->bit_ops += fr.num_ops - fr.nsyms;
        frame_pad_bits(&fr);
        const uint32_t flen = nlzm_oracle_flush_frame(fr.syms, fr.nsyms, fr.bits, fr.nbits_bytes,
                                                      fr.num_ops, fbuf, frame_size + 64);
        if (!flen) { rc = -3; goto done; }
        if (hooks && hooks->on_frame)
            hooks->on_frame(hooks->user, frame_idx, fr.num_ops, fr.syms, fr.nsyms, fr.bits, fr.nbits_bytes, fbuf, flen);
        if (out + flen + 4 > dst_cap) { rc = -1; goto done; }
        memcpy(dst + out, fbuf, flen);
        out += flen;
        frame_idx++;
        chunk_abs += p_end;
    }
    if (out + 4 > dst_cap) { rc = -1; goto done; }
    dst[out++] = 0; dst[out++] = 0; dst[out++] = 0; dst[out++] = 0;  /* NLZM.cpp:1891-1895 */
    stats->frames = frame_idx;
    stats->in_bytes = n; stats->out_bytes = out;
    stats->cmp_bytes_needed += stats->cmp_bytes;      /* (held minus the probes' excess until here) */
    *dst_len = out;
done:
    free(fbuf); free(fr.bits); free(fr.syms); free(ps); free(model);
    free(e.rk.table); free(e.bt.tree); free(e.bt.heads); free(e.ht3.rows); free(e.ht2.rows);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* decoder (DecodeFrame NLZM.cpp:642-731, model_decode_* :1369-1416,1441-1456, */
/*          decode_file :1912-2039)                                          */
/* ------------------------------------------------------------------------- */

typedef struct {
    const uint8_t *bits, *rans, *end;
    uint32_t word, word_bits, num_ops;
    uint32_t st[4];
    uint32_t idx;
    int bad;
} dframe_t;

static inline uint32_t get_be32(const uint8_t *p)
{
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

static int dsym(dframe_t *f, const cdf_t *d, int nbits)
{
    /* NLZM.cpp:666-712 */
    f->num_ops--;
    uint32_t *rs = &f->st[f->idx++ & 3];
    const uint32_t fr = *rs & (CDF_TOTAL - 1);
    const int y = cdf_find(d, nbits, fr);
    const uint32_t start = d->c[y], freq = (uint32_t)(d->c[y + 1] - d->c[y]);
    uint32_t x = freq * (*rs >> CDF_BITS) + fr - start;              /* NLZM.cpp:457-459 */
    if (x < (1u << 16)) {                                            /* NLZM.cpp:481-488 */
        if (f->rans + 2 > f->end) { f->bad = 1; return 0; }
        x = (x << 16) + ((uint32_t)f->rans[0] << 8) + f->rans[1];
        f->rans += 2;
    }
    *rs = x;
    return y;
}

static uint32_t dbits(dframe_t *f, uint32_t nb)
{
    /* NLZM.cpp:714-731 */
    f->num_ops--;
    while (f->word_bits < 24) {
        if (f->bits >= f->end) { f->bad = 1; return 0; }
        f->word |= (uint32_t)*f->bits++ << (24 - f->word_bits);
        f->word_bits += 8;
    }
    if (nb == 0) return 0;              /* reference: word >> 32 is UB; value unused there */
    const uint32_t y = f->word >> (32 - nb);
    f->word <<= nb;
    f->word_bits -= nb;
    return y;
}

static uint32_t dec_len(dframe_t *f, model_t *m)
{
    /* model_decode_lv, NLZM.cpp:1369-1383 */
    uint32_t lv = (uint32_t)dsym(f, &m->len_direct, 3);
    cdf_adapt3(&m->len_direct, (int)lv);
    if (lv == 7) {
        const int hi = dsym(f, &m->len_ext_hi, 4);
        const int lo = dsym(f, &m->len_ext_lo[hi], 4);
        cdf_adapt4(&m->len_ext_hi, hi);
        cdf_adapt4(&m->len_ext_lo[hi], lo);
        lv += ((uint32_t)hi << 4) + (uint32_t)lo;
    }
    return lv;
}

int nlzm_oracle_decompress(const uint8_t *src, uint64_t n,
                           uint8_t *dst, uint64_t dst_cap, uint64_t *dst_len)
{
    nlzm_oracle_init();
    if (n < 8) return -1;
    const uint32_t hb = ((uint32_t)src[0] << 8) + src[1];
    const uint32_t fb = ((uint32_t)src[2] << 8) + src[3];
    if (hb < 10 || hb > 28 || fb < 12 || fb > 20) return -2;  /* reference asserts hb>=12 (NLZM.cpp:1918); see README of oracle */
    model_t *m = (model_t *)malloc(sizeof *m);
    model_reset(m);
    uint64_t in = 4, out = 0;
    int rc = 0;
    for (;;) {
        if (in + 4 > n) { rc = -3; break; }
        dframe_t f; memset(&f, 0, sizeof f);
        f.num_ops = get_be32(src + in);
        if (!f.num_ops) break;                                       /* terminator */
        if (in + 12 > n) { rc = -3; break; }
        const uint32_t nb = get_be32(src + in + 4), nr = get_be32(src + in + 8);
        if (nb < 12 || in + (uint64_t)nb + nr > n || nr < 16) { rc = -3; break; }
        f.bits = src + in + 12;
        f.rans = src + in + nb;
        f.end = src + in + nb + nr;
        for (int i = 0; i < 4; i++) {                                /* NLZM.cpp:658-660 */
            f.st[i] = (uint32_t)f.rans[0] | ((uint32_t)f.rans[1] << 8) | ((uint32_t)f.rans[2] << 16) | ((uint32_t)f.rans[3] << 24);
            f.rans += 4;
        }
        while (f.num_ops > 0 && !f.bad) {
            const int cmd = dsym(&f, &m->cmd, 2);
            cdf_adapt2(&m->cmd, cmd);
            if (cmd == 0) {
                const int hi = dsym(&f, &m->lit_hi, 4);
                const int lo = dsym(&f, &m->lit_lo[hi], 4);
                cdf_adapt4(&m->lit_hi, hi);
                cdf_adapt4(&m->lit_lo[hi], lo);
                if (dst) { if (out >= dst_cap) { rc = -4; break; } dst[out] = (uint8_t)((hi << 4) + lo); }
                out++;
                continue;
            }
            uint32_t dv, lv;
            if (cmd == 1) {
                lv = dec_len(&f, m);
                const uint32_t lc = u32min(lv, 3);                   /* model_decode_dv, NLZM.cpp:1385-1416 */
                const int shi = dsym(&f, &m->slot_hi[lc], 3);
                const int slo = dsym(&f, &m->slot_lo[lc][shi], 3);
                cdf_adapt3(&m->slot_hi[lc], shi);
                cdf_adapt3(&m->slot_lo[lc][shi], slo);
                dv = ((uint32_t)shi << 3) + (uint32_t)slo;
                if (dv >= 4) {
                    uint32_t ab = (dv >> 1) - 1;
                    dv = (2 + (dv & 1)) << ab;
                    if (ab < 4) dv += dbits(&f, ab);
                    else { ab -= 4; if (ab > 0) dv += dbits(&f, ab) << 4; dv += dbits(&f, 4); }
                }
                dv += 1;
            } else if (cmd == 2) {
                const uint32_t ri = dbits(&f, 2);                    /* NLZM.cpp:1999-2001 */
                lv = dec_len(&f, m);
                dv = m->rep[ri];
            } else { rc = -5; break; }
            lv += nlzm_oracle_match_min(dv);
            rep_add(m->rep, dv);
            if (dv > out) { rc = -6; break; }
            if (dst) {
                if (out + lv > dst_cap) { rc = -4; break; }
                for (uint32_t i = 0; i < lv; i++) dst[out + i] = dst[out + i - dv];
            }
            out += lv;
        }
        if (rc) break;
        if (f.bad) { rc = -7; break; }
        in += (uint64_t)nb + nr;
    }
    free(m);
    if (!rc && dst_len) *dst_len = out;
    return rc;
}
