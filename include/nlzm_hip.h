/*
 * nlzm_hip.h -- C ABI of the MI355X (gfx950) implementation of NLZM 1.03's
 * compress-side hot path.
 *
 * The reference has no library/FFI surface: its only seam for this path is
 *
 *     void encode_file(FILE *fin, FILE *fout, uint32 hist_bits)      NLZM.cpp:1711
 *
 * called from main() (NLZM.cpp:2114).  nlzm_hip_compress() replaces the body of
 * that function -- everything between the first fread (NLZM.cpp:1774) and the
 * terminator fwrite (NLZM.cpp:1895) -- and produces the same bytes.  The
 * stage-level entry points below expose the same work split at the reference's
 * internal function boundaries so each stage can be checked on its own.
 *
 * Conventions: plain C types only; `const uint8_t *` host buffers are owned by
 * the caller; `void *d_*` arguments are device (HBM) pointers owned by the
 * caller; every function returns 0 on success or a negative NLZM_HIP_E_* code
 * (never exit()s, unlike the reference's ASSERT, NLZM.cpp:25);
 * nlzm_hip_last_error() gives a message.  One context per process and device;
 * calls are not re-entrant (the reference is single-threaded too).
 */
#ifndef NLZM_HIP_H
#define NLZM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLZM_HIP_E_ARG        (-1)   /* bad argument                                   */
#define NLZM_HIP_E_NODEVICE   (-2)   /* no usable gfx950 device / HIP runtime error    */
#define NLZM_HIP_E_NOMEM      (-3)   /* device allocation failed                       */
#define NLZM_HIP_E_CAPACITY   (-4)   /* dst_cap too small                              */
#define NLZM_HIP_E_KERNEL     (-5)   /* kernel reported an internal error / timeout    */
#define NLZM_HIP_E_TOOBIG     (-6)   /* input >= 2^32-2^16 bytes (32-bit positions)    */

/* Operation counters: same definitions as SURVEY.md section 8d (algorithmic bytes). */
typedef struct nlzm_hip_stats {
    uint64_t in_bytes, out_bytes;
    uint64_t bt_calls, bt_tests, cmp_bytes, ht_rows, rk_probes, rk_inserts;
    uint64_t positions, nice_positions, segments;
    uint64_t n_literal, n_dict, n_rep, rans_syms, bit_ops, frames, shifts;
    uint64_t uncertain_positions;     /* positions whose finder set needed the master's decision */
} nlzm_hip_stats;

/* Device-side timings of the last nlzm_hip_compress*() call (HIP events on the
 * library's own stream), in milliseconds. */
typedef struct nlzm_hip_timing {
    double total_ms;        /* first launch .. last launch complete                 */
    double h2d_ms, d2h_ms;  /* copies (0 for the *_dev entry point)                 */
    double prep_ms;         /* position-parallel pre-pass kernels                   */
    double match_parse_ms;  /* the persistent match-find + parse + emit kernel(s)   */
    double rans_ms;         /* frame rANS kernel(s)                                 */
    uint32_t match_parse_launches, rans_launches, prep_launches;
} nlzm_hip_timing;

/* ---- lifetime ------------------------------------------------------------ */

/* Select the device and create the library's stream.  Fails (NLZM_HIP_E_NODEVICE)
 * when there is no GPU: there is no CPU fallback behind this ABI. */
/* The process's FIRST nlzm_hip_init sets GPU_MAX_HW_QUEUES=16 in the process environment -- once, and only if the caller has not set it and the
 * process does not have the GPU open yet (block mode runs the small kernels of 32 streams beside one persistent launch; with the HIP runtime's
 * default of 4 hardware queues they serialise: ~70 instead of ~100 MB/s).  The runtime reads the variable at the process's FIRST HIP call: a host
 * program that uses HIP before nlzm_hip_init must export it itself.  nlzm_hip_get_counter("gpu_max_hw_queues_effective") says what the runtime
 * has read: the caller's value, 16, or 4 when it was too late. */
int nlzm_hip_init(int device);
void nlzm_hip_shutdown(void);
const char *nlzm_hip_last_error(void);

/* Upper bound on the stream size for n input bytes (every frame fits a 128 KiB
 * buffer in the reference: NLZM.cpp:1722-1724, 1738). */
uint64_t nlzm_hip_compress_bound(uint64_t n);

/* Window/frame geometry the stream will use for a file of flen bytes when
 * `-window:hist_bits_req` was asked (auto-shrink NLZM.cpp:1716-1718, frame sizes
 * NLZM.cpp:1722-1725). */
void nlzm_hip_geometry(uint64_t flen, uint32_t hist_bits_req, uint32_t *hist_bits,
                       uint32_t *frame_bits, uint32_t *chunk_size, uint32_t *feed_size);

/* ---- whole path: replaces encode_file (NLZM.cpp:1711-1910) ---------------- */

/* src/dst are host buffers.  hist_bits_req is the value after the CLI clamp to
 * [15,28] (NLZM.cpp:2085).  *dst_len receives header + frames + terminator. */
int nlzm_hip_compress(const uint8_t *src, uint64_t n, uint32_t hist_bits_req,
                      uint8_t *dst, uint64_t dst_cap, uint64_t *dst_len);

/* Same, but input and output already live in HBM (d_src must be followed by at
 * least 128 readable padding bytes).  Used by bench.py so the timed region starts
 * with the input resident. */
int nlzm_hip_compress_dev(const void *d_src, uint64_t n, uint32_t hist_bits_req,
                          void *d_dst, uint64_t dst_cap, uint64_t *dst_len);

/* Incremental form of the same call, for time-boxed runs: begin binds the input,
 * each step compresses the next `max_chunks` chunks (one chunk = one frame's
 * worth of input, NLZM.cpp:1724), finish appends the terminator.  The bytes
 * produced are identical to the one-shot call. */
int nlzm_hip_stream_begin(const void *d_src, uint64_t n, uint32_t hist_bits_req,
                          void *d_dst, uint64_t dst_cap);
int nlzm_hip_stream_step(uint32_t max_chunks, uint64_t *in_done, uint64_t *out_done, int *finished);
int nlzm_hip_stream_finish(uint64_t *dst_len);

int nlzm_hip_get_stats(nlzm_hip_stats *out);
int nlzm_hip_get_timing(nlzm_hip_timing *out);
/* Diagnostic counters of the pipeline stages for the last stream (what "stage_report" prints, by name; also "block_pool_bytes", "block_redo_streams",
 * "gpu_max_hw_queues_effective"): cycles are summed over the
 * stream's launches, e.g. "parser_total_cycles", "parser_wait_cycles", "parser_pass_cycles", "parser_passes", "parser_blocks",
 * "finder_total_cycles", "finder_wait_cycles", "finder_bt_wait_cycles", "table_total_cycles", "table_wait_cycles", "helper_jobs",
 * "helper_taken", "helper_taken_nodes", "helper_wait_cycles", "worker_call_cycles", "worker_call_tests", "worker_calls",
 * "hot_bin_calls", "positions".  No reference counterpart (the reference prints a progress line, NLZM.cpp:1861-1864); bench.py's
 * latency bound and the tests read them. */
int nlzm_hip_get_counter(const char *key, uint64_t *value);

/* ---- stage: frame coder, replaces CodeFrame::Flush (NLZM.cpp:590-640) ------ */

/* For each frame f: syms[sym_off[f] .. sym_off[f+1]) are the (freq<<16)+start
 * words buffered by WriteRange (NLZM.cpp:565); bits[bits_off[f] .. bits_off[f+1])
 * are the raw-bit bytes INCLUDING the four pad bytes Flush appends
 * (NLZM.cpp:591-597); num_ops[f] is the op count.  Frame f is written to
 * out + f*out_stride and its length to out_len[f].  Host pointers. */
int nlzm_hip_rans_frames(const uint32_t *syms, const uint64_t *sym_off,
                         const uint8_t *bits, const uint64_t *bits_off,
                         const uint32_t *num_ops, uint32_t nframes,
                         uint8_t *out, uint64_t out_stride, uint32_t *out_len);

/* ---- stage: match finding, replaces the finder block of parse_table -------- */
/* (NLZM.cpp:1501-1543: carry + HT2/HT3/BT4/RK256 -> the table copied to mt_carry)
 *
 * Runs the whole path on src but returns, for positions [pos_lo, pos_hi), the
 * per-position match tables as records {pos, max_len, delta[2..max_len]} of
 * uint32 words (same record format as the oracle's capture).  Host pointers. */
int nlzm_hip_find_matches(const uint8_t *src, uint64_t n, uint32_t hist_bits_req,
                          uint64_t pos_lo, uint64_t pos_hi,
                          uint32_t *out_words, uint64_t cap_words, uint64_t *used_words);

/* ---- stage: parse + emit, replaces parse_table's relaxations and the -------- */
/* model_encode_* calls of the driver loop (NLZM.cpp:1545-1650, 1809-1843)
 *
 * Returns the symbol/bit streams of frame `frame_idx` before rANS coding:
 * sizes_out = {nsyms, nbits_bytes (incl. pad), num_ops}.  Host pointers. */
int nlzm_hip_parse_emit(const uint8_t *src, uint64_t n, uint32_t hist_bits_req,
                        uint32_t frame_idx, uint32_t *syms, uint32_t cap_syms,
                        uint8_t *bits, uint32_t cap_bits, uint32_t *sizes_out);

/* ---- independent blocks (the path's only shard axis) ---- */
/* The reference has no block mode: k blocks = k runs of encode_file (NLZM.cpp:1711) on the byte ranges
 * [i*ceil(n/k), min(n, (i+1)*ceil(n/k))), each with its own header, window, model and terminator; the streams are
 * self-delimiting (frame headers carry sizes, :645-663; terminator :646-648) and are written back to back in block
 * order.  On one GPU all k streams are in flight at once, in one persistent launch per round: a stream needs three CUs for
 * the stages of its serial half and at least one CU of BT4 worker lanes, so k <= CUs / 4 (64 on an MI355X; the CUs left
 * are divided among the streams); an MI355X does best with 32 to 40.  d_src needs 128 readable bytes behind it (the host-buffer entry points allocate 512).
 * begin binds the input and allocates every stream's state, each step advances every stream by max_chunks chunks
 * (0: to its end), finish appends the terminators and concatenates into d_dst.  The launches of consecutive rounds are
 * queued back to back (the pre-pass kernels of the next round and the frame coder of the round before run beside a launch,
 * out of two sets of per-launch buffers), and a step that has delivered its max_chunks leaves the NEXT round of the same
 * size queued on the device for the next step to collect, so that the device does not idle between the calls: in_done_total
 * counts what has been collected (a step whose max_chunks is SMALLER than the step's before it still collects the round that one
 * left queued, i.e. more than it asked for: max_chunks bounds what a step starts, not what it finds started).  d_src and the set's
 * device buffers are therefore read and written between the calls too: the input must not change until finish or abandon.
 * A begin or step that fails has closed
 * the set (every stream's buffers freed, nothing left queued that reads d_src): there is nothing to abandon then;
 * nlzm_hip_blocks_abandon() drops a set the caller does not want to finish.  A stream whose launch fails hands its error on to the launch
 * that is already queued behind it, whose stages leave at once: the failing step returns within the time of a launch, not after a wait's bound.
 * Memory: the streams of a set reserve 32 BT4 (distance, length) pairs per position of a launch and take the rest from a per-launch arena of
 * positions / 64 + 1024 extension blocks (a single stream reserves all 256 pairs a position can have, NLZM.cpp:777).  An input with more such
 * positions than the arena serves (contrived: nlzm_amd/corpus.py `spines`) does not fail: the stream concerned is made again from its first
 * byte as a single stream, with buffers of its own, when the set is finished -- the same bytes, later; nlzm_hip_get_counter("block_redo_streams")
 * counts them. */
int nlzm_hip_blocks_begin(const void *d_src, uint64_t n, uint32_t nblocks, uint32_t hist_bits_req);
int nlzm_hip_blocks_step(uint32_t max_chunks_per_block, uint64_t *in_done_total, int *finished, double *device_ms);
int nlzm_hip_blocks_finish(void *d_dst, uint64_t dst_cap, uint64_t *block_len, uint64_t *dst_len);
void nlzm_hip_blocks_abandon(void);
/* one-shot form */
int nlzm_hip_compress_blocks_dev(const void *d_src, uint64_t n, uint32_t nblocks, uint32_t hist_bits_req,
                                 void *d_dst, uint64_t dst_cap, uint64_t *block_len, uint64_t *dst_len);

/* same, host buffers */
int nlzm_hip_compress_blocks(const uint8_t *src, uint64_t n, uint32_t nblocks, uint32_t hist_bits_req,
                             uint8_t *dst, uint64_t dst_cap, uint64_t *block_len, uint64_t *dst_len);

/* ---- streaming host input and output ---------------------- */
/* The reference reads its input and writes its frames as it goes (NLZM.cpp:1774-1778, :1853, :1870-1885).  Here the
 * caller announces the input's length, hands the bytes over in pieces, in order (any piece size; a piece travels through
 * pinned staging buffers on a copy stream while the chunks whose input has arrived are being compressed), and takes the
 * stream back in pieces (whole frames, as they are finished).  The host side never holds more than a piece; the input stays
 * whole in HBM (matches reach back a window).  feed_finish after the last piece, feed_output until it gives 0 bytes,
 * feed_end to release the buffers.  A call that fails has ended the feed. */
int nlzm_hip_feed_begin(uint64_t n, uint32_t hist_bits_req);
int nlzm_hip_feed(const uint8_t *piece, uint64_t len);
int nlzm_hip_feed_output(uint8_t *dst, uint64_t cap, uint64_t *len);
int nlzm_hip_feed_finish(void);
void nlzm_hip_feed_end(void);

/* ---- independent blocks on several GPUs of one node ------------------------ */
/* devices[0..ndev): HIP device ordinals, each listed once.  The input is cut into ndev * blocks_per_dev blocks exactly as
 * nlzm_hip_compress_blocks cuts it for that many blocks (the reference side: one encode_file call per byte range,
 * NLZM.cpp:1711); device i compresses blocks [i*blocks_per_dev, (i+1)*blocks_per_dev) in block mode on a host thread and a
 * device context of its own (nlzm_hip_init is not needed for it, and the process-wide context is left as it was).  No GPU
 * talks to another while compressing; the streams are then gathered onto devices[0], GPU to GPU (over xGMI where the
 * devices are linked), and copied out in block order.  block_len: ndev * blocks_per_dev entries (may be NULL). */
int nlzm_hip_compress_blocks_multi(const int *devices, uint32_t ndev, uint32_t blocks_per_dev,
                                   const uint8_t *src, uint64_t n, uint32_t hist_bits_req,
                                   uint8_t *dst, uint64_t dst_cap, uint64_t *block_len, uint64_t *dst_len);

/* Block mode's placement rule, for inspection (no device needed): workgroup `workgroup` of the shared persistent launch of
 * nstreams streams with blocks_per_stream workgroups each (three stage workgroups + the worker CUs) belongs to *stream and is
 * its block *local (0 finder, 1 table, 2 parser, 3.. workers).  With a multiple of eight streams all blocks of a stream have the
 * same workgroup index modulo 8, i.e. lie on one XCD under the round-robin dispatch (speed only). */
void nlzm_hip_block_placement(uint32_t nstreams, uint32_t blocks_per_stream, uint32_t workgroup, uint32_t *stream, uint32_t *local);

/* ---- tuning knobs (defaults are what bench.py measures) -------------------- */
/* key: "workers" (only 1: BT4 runs on per-head worker lanes), "batch_chunks" (chunks per persistent launch), "worker_blocks" (worker CUs of a stream, default 240: the stage CUs and these fill the device),
 * "worker_threads" (bin-taking lanes per worker CU, 64..512, default 128), "hot_waves" (waves per worker CU that take a hot
 * BT4 bin each, 0..6, default 2), "hot_min" (positions per launch from which a bin counts as hot, default 0: by the stream's pace -- 24 positions per millisecond of the launch before);
 * block mode: "block_worker_threads" (default 320) and "block_hot_waves" (default 3), the same two for the streams of a block set
 * (their worker CUs are few: the count follows from the number of streams), "block_batch_chunks" (chunks of every stream per
 * shared launch, default 8); "prefilter_bits_per_position" (log2 of the pre-filter table's entries per input position, default 4);
 * "keep_block_pool" (default 1: the one device allocation of a block set is kept when the set is closed and used again by the
 * next set that fits -- the driver clears freed device memory, and an allocation made soon after a large one was freed waits for it;
 * 0 releases it, as nlzm_hip_shutdown does; the kept allocation is what the set needed -- up to 0.85 of the device's free memory --
 * and is given back by itself when any other allocation of the library, e.g. a single stream's, fails for lack of memory);
 * "parser_helper" (default 1: a stream gets a helper parser workgroup -- one CU more -- that parses the back of every segment that is cut
 * at 4,096 positions while the parser stage parses its front; "block_parser_helper", default 0, the same for the streams of a block set);
 * "table_shape" (default 0: the table stage runs every launch with 16-entry fronts on seven waves or with 24-entry fronts on five, whichever the
 * launch before it asked for -- the share of its blocks in which some position had more entries than the fronts hold decides; 1 / 2 fix the
 * shape: source code runs ~20 % faster in the wide one, prose ~8 % faster in the narrow one);
 * test only: "test_fail_launch" (default -1) / "test_fail_stream": the finder stage of that launch of that stream of a block set (of the single stream)
 * raises an error at once; "block_ext_blocks" (default -1: by the launch's size): extension blocks of a block-set stream's pair-list arena;
 * "stage_report" (1: the stages' cycle accounting of
 * every finished stream, and of a block set per stream, on stderr).  There are no environment knobs.
 * None of them changes a byte of the output.  The options are read when a stream or a block set is opened
 * (nlzm_hip_stream_begin, nlzm_hip_blocks_begin, nlzm_hip_feed_begin, the one-call entries): what is open keeps what it was opened with.
 * nlzm_hip_compress_blocks_multi is EXPERIMENTAL until a run on more than one device has been recorded (it has only been run with one). */
int nlzm_hip_set_option(const char *key, int64_t value);

#ifdef __cplusplus
}
#endif
#endif
