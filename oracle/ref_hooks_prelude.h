/*
 * ref_hooks_prelude.h -- observation hooks for the REFERENCE build.
 *
 * TEST INFRASTRUCTURE ONLY.  oracle/Makefile pipes this file, followed by the
 * reference source read from /root/reference/NLZM.cpp through a handful of
 * `sed` insertions, into g++ (nothing of the reference is written to disk; the
 * only output is the binary oracle/_ref/nlzm_ref_instr).  The insertions call
 * the macros below at the sites SURVEY.md section 8c lists:
 *
 *   NLZM_HOOK_POS    after `mt_carry = mt;`                     (NLZM.cpp:1543)
 *   NLZM_HOOK_SEG    after the parse_table() call               (NLZM.cpp:1802)
 *   NLZM_HOOK_FRAME  around `frame.Flush()`                     (NLZM.cpp:1850)
 *   NLZM_HOOK_SHIFT  inside the window-rebase branch            (NLZM.cpp:1786)
 *
 * The compressed output of the instrumented binary is byte-identical to the
 * plain build (checked by oracle/make_golden.py).
 * Digests are written as JSON to the file named by $NLZM_DIGEST_OUT at exit.
 */
#include "digest.h"
#include <string.h>

static nlzm_digest g_dg;
static int g_dg_on = 0;
static unsigned long long g_seg_abs = 0;      /* absolute offset of parse-relative 0 */
static unsigned long long g_reb_base = 0;     /* k * window_size */
static unsigned char *g_frame_bits_copy = 0;
static unsigned g_frame_nbits = 0, g_frame_word = 0, g_frame_ops = 0, g_frame_nsyms = 0;
static unsigned *g_frame_syms_copy = 0;

static void nlzm_hooks_atexit(void)
{
    const char *path = getenv("NLZM_DIGEST_OUT");
    if (!g_dg_on || !path) return;
    FILE *f = fopen(path, "w");
    if (!f) return;
    nlzm_digest_write_json(&g_dg, f, 1);
    fclose(f);
}

static inline void nlzm_hooks_start(void)
{
    if (g_dg_on) return;
    g_dg_on = 1;
    nlzm_digest_init(&g_dg);
    atexit(nlzm_hooks_atexit);
}

#define NLZM_HOOK_POS(rel_p, mt) do { nlzm_hooks_start(); \
    nlzm_digest_pos(&g_dg, g_seg_abs + (rel_p), (mt).max_len, (mt).delta); } while (0)

#define NLZM_HOOK_SEG(nodes, seg_len, abs_start) do { nlzm_hooks_start(); \
    nlzm_digest_seg_begin(&g_dg, (abs_start), (seg_len)); \
    unsigned _q = 0; \
    while (_q < (seg_len)) { \
        nlzm_digest_seg_cmd(&g_dg, (nodes)[_q].cmd, (nodes)[_q].len, (nodes)[_q].delta); \
        _q += (nodes)[_q].cmd == 0 ? 1u : (nodes)[_q].len; \
    } } while (0)

#define NLZM_HOOK_FRAME_PRE(fr) do { nlzm_hooks_start(); \
    g_frame_ops = (fr).num_ops; g_frame_nsyms = (fr).num_rans_syms; \
    g_frame_nbits = (unsigned)((fr).ptr_bits - ((fr).start + 12)); g_frame_word = (fr).word; \
    g_frame_syms_copy = (unsigned *)realloc(g_frame_syms_copy, 4u * (g_frame_nsyms + 1)); \
    memcpy(g_frame_syms_copy, (fr).buf_rans, 4u * g_frame_nsyms); \
    g_frame_bits_copy = (unsigned char *)realloc(g_frame_bits_copy, g_frame_nbits + 1); \
    memcpy(g_frame_bits_copy, (fr).start + 12, g_frame_nbits); } while (0)

#define NLZM_HOOK_FRAME_POST(fr, written) do { \
    nlzm_digest_frame(&g_dg, g_frame_ops, g_frame_syms_copy, g_frame_nsyms, g_frame_bits_copy, \
                      g_frame_nbits, g_frame_word, (fr).start, (written)); } while (0)

#define NLZM_HOOK_SHIFT(w) do { g_reb_base += (w); } while (0)
