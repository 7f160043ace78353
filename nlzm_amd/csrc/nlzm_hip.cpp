// nlzm_hip.cpp -- host pipeline behind the C ABI of include/nlzm_hip.h.
//
// Replaces encode_file (NLZM.cpp:1711-1910): the whole input lives in HBM as one
// flat buffer; chunks (= frames, NLZM.cpp:1724) are processed in batches by the
// persistent match-find/parse/emit launch, then every frame of the batch is
// rANS-coded in parallel and gathered into the output stream.
// There is no CPU implementation of any stage in this library.
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <unistd.h>

#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <array>
#include <mutex>
#include <thread>
#include <chrono>
#include <algorithm>
#include <vector>

#include "../../include/nlzm_hip.h"
#include "nlzm_core.h"
#include "nlzm_v2.h"

namespace nlzm {
void launch_rk_hash(const uint8_t *in, unsigned long long n, unsigned long long pos0, unsigned long long pos1,
                    uint32_t *out, hipStream_t st);
void launch_pipeline2(const Geom &g, const Globals &G, const v2::GlobalsV2 &V, uint32_t c0, uint32_t c1, uint32_t worker_blocks, hipStream_t st);
unsigned long long stream2_pack_size();
uint32_t stream2_pack_capacity();
uint32_t pipeline2_role_blocks();
void fill_stream2_args(void *host_pack, uint32_t i, const Geom &g, const Globals &G, const v2::GlobalsV2 &V, uint32_t c0, uint32_t c1, v2::RoundSnap *snap);
void launch_pipeline2_multi(const void *dev_pack, uint32_t nstreams, uint32_t worker_blocks, hipStream_t st);
void launch_round_open(const void *dev_pack, uint32_t nstreams, hipStream_t st);
void launch_round_close(const void *dev_pack, uint32_t nstreams, hipStream_t st);
void launch_prefilter(const uint8_t *in, unsigned long long n, uint32_t a0, uint32_t a1, uint32_t wmask, uint32_t t_bits, uint32_t t_bitmap,
                      uint32_t m_bits, uint32_t *T, uint32_t *M, uint32_t *hbuf, uint32_t *hbuf2, uint8_t *c1, uint8_t *unc, hipStream_t st);
unsigned long long worker_undo_bytes_per_lane();
unsigned long long worker_hot_undo_bytes_per_wave();
void launch_hot_select(const uint32_t *off, uint32_t nchunks, uint32_t nheads, uint32_t hmax, uint32_t min_count, uint32_t *hot_of_bin,
                       uint32_t *hot_list, WorkerCounters *wcnt, hipStream_t st);
void launch_bin(const uint8_t *in, const Geom &g, uint32_t c0, uint32_t nchunks, uint32_t nheads, uint32_t *off, uint32_t *cur,
                uint32_t *pos, const uint8_t *unc, uint32_t batch_a0, hipStream_t st);
void launch_rans(const uint32_t *syms, unsigned long long syms_stride, const uint8_t *bits, unsigned long long bits_stride,
                 FrameMeta *fmeta, uint32_t *scratch, unsigned long long scratch_stride, uint8_t *out,
                 unsigned long long out_stride, uint32_t out_cap, uint32_t nframes, hipStream_t st);
void launch_gather(const uint8_t *frames, unsigned long long stride, const unsigned long long *dst_off,
                   const FrameMeta *fmeta, uint8_t *dst, uint32_t nframes, hipStream_t st);
}  // namespace nlzm

using namespace nlzm;

namespace {

char g_err[2048] = "";
std::mutex g_err_mu;            // block streams run on host threads
char *thread_err();             // the error text of the device state this thread works for (a multi-device call: one per device)
int set_err(int code, const char *fmt, ...)
{
    std::lock_guard<std::mutex> lk(g_err_mu);
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    if (char *te = thread_err()) memcpy(te, g_err, sizeof g_err);
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return set_err(e_ == hipErrorOutOfMemory ? NLZM_HIP_E_NOMEM : NLZM_HIP_E_NODEVICE,    \
                           "%s failed: %s", #expr, hipGetErrorString(e_));                        \
    } while (0)

int g_hwq_effective = 0;                           // what the HIP runtime reads as GPU_MAX_HW_QUEUES (dev_init)
void blocks_close(bool drop_pool = false);         // defined with the block-set entry points
bool idle_block_pool_dropped();                    // no block set open and its kept allocation still there: frees it, true (defined there too)

inline uint32_t clampu(uint32_t v, uint32_t lo, uint32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One allocation for everything a stream keeps on the device (block mode: one for the whole block set).  A context without a
// pool takes every buffer from hipMalloc by itself.
struct Pool {
    uint8_t *base = nullptr;
    size_t size = 0, used = 0;
    bool measuring = false;             // only add up what the stream would take
};

struct Ctx {
    bool inited = false;
    int device = 0;
    hipStream_t st = nullptr;
    hipEvent_t ev[8] = {};
    hipEvent_t ev_b[8] = {};                // (block mode) the events of the second launch set

    // options
    int64_t opt_workers = 1;
    int64_t opt_batch = 32;
    int64_t opt_worker_blocks = 240;        // + the stage blocks, one 512-thread block per CU.  Round 5: with the serial half at 400 - 460 cycles per position the
                                            // number of worker CUs matters again at depth -- a lane's bin holds the heads h with h % lanes equal, a long call of one
                                            // head holds up the positions of the others, and the finder stage waits: the whole 1e9-byte stream with 60 / 240
                                            // worker CUs 459 / 412 cycles per position (BT4 results waited for: 173 / 127), 300 MB 399 / 396.  Round 3 had measured (profiles/
                                            // r03_worker_cu_sweep.log): 240 / 120 / 60 / 30 / 16 / 8 worker CUs give 3.71 / 3.72 / 3.72 / 3.70 / 3.70 / 3.63 MB/s
                                            // at 150 MB depth and 240 / 60 / 32 the same at 20 MB and 300 MB -- the lanes are there for latency, and the hot
                                            // bins have waves of their own; 60 leaves a margin and three quarters of the device to other streams
    int64_t opt_worker_threads = 128;       // lanes of a worker block that take bins (with two of a CU's eight waves walking trees a test takes less
                                            // time than with all eight -- measured at 60 MB: 512 lanes per CU 2.48 MB/s, 256 2.58, 128 2.62, 64 2.62)
    int64_t opt_hot_waves = 2;              // waves of a worker block behind its bin-taking lanes that take a hot bin each (0: none)
    int64_t opt_hot_min = 0;                // positions per launch from which a bin may count as hot; 0 (default): by the stream's pace.  A bin needs a wave when its calls
                                            // come faster than a lane serves them -- a lane's call costs ~42 us with its lockstep partners' --, and how fast they come hangs on
                                            // how fast the FINDER moves: bins of 24 positions per millisecond of the launch before and more (the first launch: positions / 240).
                                            // Measured at launches of 8 chunks (profiles/r06_ab_runs.txt): the stand-in (166 ms a launch) 401 / 390 / 382 / 371 cycles per position
                                            // at 8,192 / 6,144 / 4,096 / 2,048; markup (362 ms) 872 / 906 / 1,005 / 1,007 -- a wave's call takes twice a lane's (its steps are
                                            // heavier), which is lost where a lane would have kept up.  Rounds 3 - 5 had 8,192 fixed.
    double last_launch_ms = 0;              // duration of the stream's last persistent launch (0: none yet)
    int64_t opt_tbits_max = 34;             // log2 of the pre-filter table's entries at most (block mode shrinks it to fit)
    int64_t opt_block_threads = 320;        // block mode: lanes of a worker block that take bins, and the waves behind them that take a hot bin each.  Measured
    int64_t opt_block_hot_waves = 3;        // with 32 streams of 17 MB (4 worker CUs each): 512 lanes and no such waves 9.6 s, 256 + 4 waves 8.2 s, 128 + 6 waves 8.6 s
                                            // (profiles/r04_block_mode.txt): under load a stream waits for the serial chains of its busiest heads
    int64_t opt_block_batch = 8;            // block mode: chunks of every stream per shared launch (the rounds overlap, so their length matters little --
                                            // 6 / 8 / 12 / 16 chunks: 88.9 / 89.8 / 88.7 / 89.6 MB/s; the pool holds 2.3 KB per position of a launch and stream)
    int64_t opt_tbits_per = 4;              // log2 of the pre-filter table's entries per input position (capped by window + 5 and 32 bits)
    int64_t opt_keep_pool = 1;              // block mode keeps its one allocation when a set is closed: the driver clears freed device memory, and an allocation
                                            // made soon after a large one was freed waits for that -- opening 32 streams took 0.12 s or 4.5 s (tests/gpu_begin_probe.py)
    int64_t opt_helper = 1;                 // a helper parser workgroup (nlzm_v2.h, HelpBox): 1 CU more per stream.  The streams of a block
    int64_t opt_block_helper = 0;           // set run without one unless "block_parser_helper" says otherwise (a stream of a full device waits for its BT4 results)
    int64_t opt_multi_same = 0;             // test only ("multi_allow_same_device"): nlzm_hip_compress_blocks_multi accepts a device twice, so that its
                                            // threads, device states and gather loop run with two parts on a box with one GPU
    int64_t opt_table_shape = 0;            // the table stage's shape ("table_shape"): 0 every launch in the shape the launch before it asked for (nlzm_v2.h, TLds), 1 always 16-entry
                                            // fronts on seven waves, 2 always 24 entries on five
    int64_t opt_test_fail_launch = -1;      // test only ("test_fail_launch", with "test_fail_stream" = index of the stream of a block set): the finder stage of that launch raises
    int64_t opt_test_fail_stream = 0;       // an error at once -- the fault path of a round that is queued behind a failing one
    int64_t opt_block_ext_blocks = -1;      // test only ("block_ext_blocks"): extension blocks of a block-set stream's pair-list arena per launch (default: positions / 64 + 1024)
    bool arena_out = false;                 // the last step_post_check failed because the launch used its pair-list arena up (block mode makes the stream again by itself)
    int64_t opt_report = 0;                 // 1: the stages' cycle accounting of every finished stream on stderr (nlzm_hip_set_option "stage_report")
    int cu_count = 0;
    // what the open stream runs with: the options as they were at stream_begin (its buffers are sized for them)
    uint32_t run_worker_blocks = 0, run_worker_threads = 0;
    unsigned long long prof_last[128] = {}; // of the last finished stream: Persist::prof and the worker lanes' counters (nlzm_hip_get_counter)
    WorkerCounters wc_last{};
    double acct[8] = {};                    // of the last finished stream, cycles per position: finder total / wait / of it for BT4, table total / wait, parser total / wait / passes


    // stream state
    bool open = false;
    Geom g{};
    const uint8_t *d_in = nullptr;
    uint8_t *d_dst = nullptr;
    uint64_t dst_cap = 0, out_pos = 0;
    uint32_t next_chunk = 0;                // chunks below this are coded and gathered
    uint32_t pre_chunk = 0;                 // chunks below this have had their pre-pass queued (block mode runs it a launch ahead)

    // device buffers
    unsigned long long rkhash_len = 0;      // entries of rkhash (a launch's positions and their lookahead)
    uint32_t *rkhash = nullptr, *ht2 = nullptr, *ht3 = nullptr, *rk_table = nullptr, *bt_heads = nullptr, *bt_tree = nullptr;
    Persist *persist = nullptr;
    uint32_t *syms = nullptr, *scratch = nullptr;
    uint8_t *bits = nullptr, *frames = nullptr;
    FrameMeta *fmeta = nullptr;
    unsigned long long *dst_off = nullptr;
    unsigned long long syms_stride = 0, bits_stride = 0, frame_stride = 0;
    uint32_t batch = 0;
    // worker mode
    bool workers = false;
    uint32_t *pf_T = nullptr, *pf_M = nullptr, *pf_h = nullptr, *pf_h2 = nullptr; uint8_t *pf_c1 = nullptr, *unc = nullptr;
    uint32_t t_bits = 0, m_bits = 0, nheads = 0;
    uint32_t t_bitmap = 0;                  // the pre-filter table holds one bit per slot (a stream no longer than its window: every earlier position is inside it)
    uint32_t v2_launch_no = 0;              // persistent launches of the open stream so far (its parity picks the table stage's shape slot)
    uint32_t *bt_ready = nullptr, *bt_pairs = nullptr, *bt_flag = nullptr, *abort_word = nullptr;
    uint32_t *bt_ext = nullptr; uint32_t pstride = kBtMaxPairs, ext_cap = 0;       // pairs reserved per position; extension blocks for the rest
    uint32_t *bin_off = nullptr, *bin_cur = nullptr, *bin_pos = nullptr, *bt_undo = nullptr, *hot_of_bin = nullptr, *hot_list = nullptr;
    unsigned long long *hot_undo = nullptr;
    uint32_t hot_max = 0;
    WorkerCounters *wcnt = nullptr;
    // three-stage pipeline (nlzm_v2.h): hand-off rings, progress words, stage state
    uint32_t *v2_ft = nullptr, *v2_tp = nullptr, *v2_tf = nullptr, *v2_state = nullptr;
    v2::Hx *v2_hx = nullptr;
    v2::HelpBox *v2_hb = nullptr;
    v2::Hx hx_host;

    // capture (stage tests)
    uint32_t *cap_words = nullptr; unsigned long long cap_cap = 0, cap_lo = 0, cap_hi = 0; unsigned long long *cap_used = nullptr;
    // frame capture for parse_emit
    int64_t want_frame = -1;
    std::vector<uint32_t> got_syms; std::vector<uint8_t> got_bits; FrameMeta got_meta{};
    bool got = false;

    // owned copies for the host-buffer entry point
    uint8_t *own_in = nullptr, *own_dst = nullptr;

    nlzm_hip_stats stats{};
    nlzm_hip_timing tm{};
    unsigned long long last_dry_runs = 0, last_flag_waits = 0;
    // what the step's frame coder has reported (step_post_issue .. _done)
    std::vector<FrameMeta> post_hm; std::vector<unsigned long long> post_hoff; Persist post_P; uint32_t post_aborted = 0; unsigned long long post_pos = 0;
    // Block mode queues launch k + 1 (and runs its pre-pass) while launch k is on the device and codes the frames of launch k
    // while launch k + 1 is: everything a pre-pass writes or a frame coder reads exists twice, and swap_sets() makes the other
    // set the current one (the members above) before a pre-pass.
    bool double_sets = false;
    uint32_t set_idx = 0;
    struct LaunchSet {
        uint32_t *rkhash = nullptr, *bt_ready = nullptr, *bt_flag = nullptr, *abort_word = nullptr, *bin_off = nullptr, *bin_pos = nullptr,
                 *hot_of_bin = nullptr, *hot_list = nullptr, *syms = nullptr;
        uint8_t *unc = nullptr, *bits = nullptr;
        FrameMeta *fmeta = nullptr;
        v2::RoundSnap *snap = nullptr;
    } alt;
    v2::RoundSnap *snap = nullptr;          // (current set) what round_close_kernel copies aside for the host
    v2::RoundSnap post_snap;
    Pool *pool = nullptr;                   // (block mode) where the stream's buffers come from
    bool pooled = false;                    // the open stream's buffers are the pool's: not freed one by one
};

void swap_sets(Ctx &C)
{
    Ctx::LaunchSet &A = C.alt;
    std::swap(C.rkhash, A.rkhash); std::swap(C.bt_ready, A.bt_ready); std::swap(C.bt_flag, A.bt_flag); std::swap(C.abort_word, A.abort_word);
    std::swap(C.bin_off, A.bin_off); std::swap(C.bin_pos, A.bin_pos); std::swap(C.hot_of_bin, A.hot_of_bin); std::swap(C.hot_list, A.hot_list);
    std::swap(C.syms, A.syms); std::swap(C.unc, A.unc); std::swap(C.bits, A.bits); std::swap(C.fmeta, A.fmeta); std::swap(C.snap, A.snap);
    C.set_idx ^= 1;
}

// a device buffer of the stream: from its pool, or from hipMalloc
template <class T> int dev_alloc(Ctx &C, T **p, size_t bytes)
{
    if (!C.pool) {
        hipError_t e = hipMalloc((void **)p, bytes);
        if (e == hipErrorOutOfMemory && idle_block_pool_dropped()) {    // (the allocation a closed block set left behind: given back, once, for this one)
            (void)hipGetLastError();
            e = hipMalloc((void **)p, bytes);
        }
        if (e != hipSuccess) return set_err(e == hipErrorOutOfMemory ? NLZM_HIP_E_NOMEM : NLZM_HIP_E_NODEVICE, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return 0;
    }
    Pool &P = *C.pool;
    const size_t at = (P.used + 255) & ~(size_t)255;
    if (!P.measuring && at + bytes > P.size) return set_err(NLZM_HIP_E_NOMEM, "stream pool of %zu bytes is too small", P.size);
    *p = P.measuring ? nullptr : (T *)(P.base + at);
    P.used = at + bytes;
    return 0;
}
#define DEVALLOC(ptr, bytes) do { const int rc_ = dev_alloc(C, &(ptr), (bytes)); if (rc_) return rc_; } while (0)
#define DEVFILL(expr) do { if (!(C.pool && C.pool->measuring)) HIPCHK(expr); } while (0)

//   (the plan keeps the launch set and the events the step was queued with: block mode has two steps of a stream open at a time)
struct StepPlan {
    uint32_t c0 = 0, c1 = 0, nb = 0; Globals G; v2::GlobalsV2 V;
    hipEvent_t ev[8] = {};
    uint32_t *syms = nullptr, *abort_word = nullptr; uint8_t *bits = nullptr; FrameMeta *fmeta = nullptr;
    v2::RoundSnap *snap = nullptr;              // non-null: the launch is followed by round_close_kernel, the host reads this
};

struct BlockJob {
    Ctx c;
    uint64_t lo = 0, n = 0, len = 0, bound = 0;
    uint8_t *d_out = nullptr;
    int rc = 0;
    bool redo = false;                              // a launch of this stream used its pair-list arena up: the stream is made again, by itself, when the set is finished
    Pool pool;                                      // this stream's slice of the block set's one allocation
};

// Everything the entry points keep per device: the single-stream context and the open block set.  The process-wide one serves
// the one-device API (nlzm_hip_init picks its device); a multi-device call gives each of its per-device host threads one
// of its own and points `t_dev` at it, so that the same code runs on every device at once.
struct DevState {
    char err[sizeof g_err] = "";                    // the last error raised by a thread that works for this state
    Ctx ctx;                                        // the context behind the single-stream entry points
    std::vector<BlockJob> jobs;                     // the open block set (nlzm_hip_blocks_begin .. _finish)
    uint8_t *blocks_pool = nullptr;                 // ... and the one allocation all its streams' buffers lie in: kept between block sets
    size_t blocks_pool_size = 0;                    //     (option keep_block_pool) and used again by the next set that fits
    std::vector<hipStream_t> group_st;              // one HIP stream and an event pair per shared launch of a round
    std::vector<std::array<hipEvent_t, 3>> group_ev;   // per launch set: launch begins / ends / its results are copied aside
    void *pack_host = nullptr, *pack_dev = nullptr; // the streams' launch arguments of a round: pinned host copy, device copy
    uint64_t blocks_n = 0;
    const uint8_t *blocks_src = nullptr;
    uint32_t blocks_hist = 0;
    int64_t blocks_wb = 0;
    uint64_t blocks_per = 0;                        // bytes per block when the caller fixes the partition (0: ceil(n / nblocks))
    uint64_t redo_streams = 0;                      // streams of the last block set that were made again as single streams (their pair-list arena had run out)
    // the rounds of the block set (blocks_step_impl): two are open at a time, and one may stay queued when a step returns
    struct Rounds {
        bool have = false;                          // round `q` is queued (pre-passes and launch) and not collected yet
        uint32_t q = 0;
        std::vector<StepPlan> plan[2];
        std::vector<uint32_t> act[2];
    } rounds;
    // streaming host input (nlzm_hip_feed_*): two pinned staging buffers on a copy stream of their own
    struct Feed {
        bool open = false;
        uint64_t n = 0, fed = 0, arrived = 0, taken = 0;    // input bytes handed over / known to be in HBM; output bytes handed back
        uint8_t *pin[2] = { nullptr, nullptr };
        hipEvent_t ev[2] = { nullptr, nullptr };
        uint64_t end_of[2] = { 0, 0 };                      // input offset a staging buffer's last upload ends at
        hipStream_t st = nullptr;
        uint32_t next = 0;
    } feed;
};
constexpr uint64_t kFeedPiece = 32ull << 20;        // bytes per staging buffer
DevState g_dev0;
thread_local DevState *t_dev = nullptr;
inline DevState &cur() { return t_dev ? *t_dev : g_dev0; }
char *thread_err() { return t_dev ? t_dev->err : nullptr; }
#define g_ctx (cur().ctx)
#define g_jobs (cur().jobs)
#define g_group_st (cur().group_st)
#define g_group_ev (cur().group_ev)
#define g_pack_host (cur().pack_host)
#define g_pack_dev (cur().pack_dev)
#define g_blocks_n (cur().blocks_n)
#define g_blocks_src (cur().blocks_src)
#define g_blocks_hist (cur().blocks_hist)
#define g_blocks_wb (cur().blocks_wb)

void free_stream_buffers(Ctx &C)
{
    void *ptrs[] = { C.rkhash, C.ht2, C.ht3, C.rk_table, C.bt_heads, C.bt_tree, C.persist, C.syms, C.scratch,
                     C.bits, C.frames, C.fmeta, C.dst_off, C.own_in, C.own_dst, C.pf_T, C.pf_M, C.pf_h, C.pf_h2, C.pf_c1, C.unc,
                     C.bt_ready, C.bt_pairs, C.bt_flag, C.abort_word, C.bin_off, C.bin_cur, C.bin_pos, C.wcnt, C.bt_undo, C.hot_of_bin, C.hot_list, C.hot_undo,
                     C.v2_ft, C.v2_tp, C.v2_tf, C.v2_state, C.v2_hx, C.v2_hb, C.bt_ext };
    if (!C.pooled) for (void *p : ptrs) if (p) (void)hipFree(p);
    C.pooled = false;
    C.alt = Ctx::LaunchSet{}; C.snap = nullptr; C.set_idx = 0;      // (a second launch set only ever comes from a pool)
    C.v2_ft = C.v2_tp = C.v2_tf = C.v2_state = nullptr; C.v2_hx = nullptr; C.v2_hb = nullptr;
    C.pf_T = C.pf_M = C.pf_h = C.pf_h2 = nullptr; C.pf_c1 = C.unc = nullptr; C.bt_ready = C.bt_pairs = nullptr; C.bt_ext = nullptr;
    C.bt_flag = C.abort_word = C.bin_off = C.bin_cur = C.bin_pos = C.bt_undo = C.hot_of_bin = C.hot_list = nullptr; C.hot_undo = nullptr; C.wcnt = nullptr;
    C.rkhash = C.ht2 = C.ht3 = C.rk_table = C.bt_heads = C.bt_tree = nullptr;
    C.persist = nullptr; C.syms = C.scratch = nullptr; C.bits = C.frames = nullptr; C.fmeta = nullptr;
    C.dst_off = nullptr; C.own_in = C.own_dst = nullptr;
    C.open = false;
}

void make_geom(uint64_t n, uint32_t hist_bits_req, Geom &g)
{
    uint32_t hb = hist_bits_req;
    while (hb > 10 && n < (1ull << (hb - 1))) --hb;                 // NLZM.cpp:1716-1718
    g.n = n;
    g.wbits = hb; g.wmask = (1u << hb) - 1;
    g.frame_bits = clampu(hb - 2, 14, 17);                          // :1722
    g.frame_size = 1u << g.frame_bits;
    g.chunk_size = ((g.frame_size * 15) / 16) - 0x200;              // :1724
    g.feed = g.chunk_size + kMatchMax + 1;                          // :1725
    g.ht3_shift = 32 - (12 + clampu(hb, 15, 17) - 15);              // :1751
    g.bt_shift = 32 - (13 + clampu(hb, 16, 20) - 16);               // :1752
    g.rk_shift = 32 - (15 + clampu(hb, 16, 22) - 16);               // :1753
    g.tag_mask = (uint32_t)((1ull << (32 - hb)) - 1);
    g.nchunks = (uint32_t)((n + g.chunk_size - 1) / g.chunk_size);
    g.bt_tmask = g.wmask;       // widened by stream_begin once the launch size is known
}

int stream_begin(Ctx &C, const void *d_src, uint64_t n, uint32_t hist_bits_req, void *d_dst, uint64_t dst_cap)
{
    if (!C.inited) return set_err(NLZM_HIP_E_NODEVICE, "nlzm_hip_init() has not succeeded");
    if (n >= 0xFFFF0000ull) return set_err(NLZM_HIP_E_TOOBIG, "input of %llu bytes needs >32-bit positions", (unsigned long long)n);
    if (hist_bits_req < 10 || hist_bits_req > 28) return set_err(NLZM_HIP_E_ARG, "hist_bits %u outside [10,28]", hist_bits_req);
    if (dst_cap < 8) return set_err(NLZM_HIP_E_CAPACITY, "dst_cap < 8");
    {   // keep caller-visible buffers of a host-entry call alive across the reset
        uint8_t *oi = C.own_in, *od = C.own_dst;
        C.own_in = C.own_dst = nullptr;
        free_stream_buffers(C);
        C.own_in = oi; C.own_dst = od;
    }
    make_geom(n, hist_bits_req, C.g);
    const Geom &g = C.g;
    C.d_in = (const uint8_t *)d_src; C.d_dst = (uint8_t *)d_dst; C.dst_cap = dst_cap;
    C.out_pos = 0; C.next_chunk = 0; C.pre_chunk = 0; C.last_launch_ms = 0;
    memset(&C.stats, 0, sizeof C.stats);
    memset(&C.tm, 0, sizeof C.tm);
    C.stats.in_bytes = n;

    const size_t ht3_rows = (size_t)2 << (32 - g.ht3_shift);
    // (RK256 hashes of ONE launch's positions -- the array is indexed by absolute position through a base pointer moved back by
    //  the launch's first hashed position, so only a launch's worth is ever resident: 4 B x (launch + lookahead) instead of 4 B
    //  per input byte, which was 4 GB at 1e9 bytes)
    C.rkhash_len = 0;
    DEVALLOC(C.ht2, 4096 * 4);
    DEVALLOC(C.ht3, ht3_rows * 4);
    DEVALLOC(C.rk_table, (size_t)4 << (32 - g.rk_shift));
    DEVALLOC(C.bt_heads, (size_t)4 << (32 - g.bt_shift));
    {
        // node slots >= W + positions per launch (see Geom::bt_tmask)
        uint32_t b = (uint32_t)(C.opt_batch < 1 ? 1 : C.opt_batch);
        if (g.nchunks && b > g.nchunks) b = g.nchunks;
        const unsigned long long need = (1ull << g.wbits) + (unsigned long long)b * g.chunk_size;
        unsigned long long slots = 1ull << g.wbits;
        while (slots < need) slots <<= 1;
        C.g.bt_tmask = (uint32_t)(slots - 1);
    }
    DEVALLOC(C.bt_tree, ((size_t)C.g.bt_tmask + 1) * 8);
    DEVALLOC(C.persist, sizeof(Persist));
    DEVFILL(hipMemsetAsync(C.ht2, 0xFF, 4096 * 4, C.st));                         // :902
    DEVFILL(hipMemsetAsync(C.ht3, 0xFF, ht3_rows * 4, C.st));
    DEVFILL(hipMemsetAsync(C.rk_table, 0xFF, (size_t)4 << (32 - g.rk_shift), C.st));   // :1040
    DEVFILL(hipMemsetAsync(C.bt_heads, 0xFF, (size_t)4 << (32 - g.bt_shift), C.st));   // :968
    DEVFILL(hipMemsetAsync(C.bt_tree, 0xFF, ((size_t)C.g.bt_tmask + 1) * 8, C.st));     // :969

    Persist P;
    memset(&P, 0, sizeof P);
    for (uint32_t ctx = 0; ctx < kNumCtx; ctx++) {                                // model_init :1183-1206, cdf_init :324-346
        const uint32_t ns = (ctx == kCtxCmd) ? 4 : ((ctx == kCtxLenDirect || ctx >= kCtxSlotHi) ? 8 : 16);
        for (uint32_t i = 0; i <= ns; i++) P.cdf[ctx * kCdfStride + i] = (uint16_t)(i * (16384 / ns));
    }
    for (int i = 0; i < 4; i++) P.rep[i] = (uint32_t)i + 1;                       // :1154-1158
    DEVFILL(hipMemcpyAsync(C.persist, &P, sizeof P, hipMemcpyHostToDevice, C.st));

    C.batch = (uint32_t)(C.opt_batch < 1 ? 1 : C.opt_batch);
    if (C.batch > g.nchunks && g.nchunks) C.batch = g.nchunks;
    if (!C.batch) C.batch = 1;
    C.pooled = C.pool != nullptr;       // (from here on: a stream_begin that fails midway must not hipFree pointers into the set's pool)
    if (!C.pool) {  // the per-launch arrays take about 2.3 KB per position of a launch: a launch that does not fit is cut down
        // (not for the streams of a block set: blocks_begin fitted their batch to the memory, and the free memory differs between the pass that
        //  measures what a stream takes and the pass that takes it -- the pool itself is allocated in between)
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        // (the pre-filter table as it will be sized below: up to 2^33 entries)
        uint32_t tb = g.wbits + 5 > 33 ? 33 : (g.wbits + 5 < 16 ? 16 : g.wbits + 5);
        { uint32_t lgn = 1; while ((1ull << lgn) < g.n) lgn++; const uint32_t want = lgn + (uint32_t)C.opt_tbits_per; if (want < tb) tb = want < 16 ? 16 : want; }
        if ((int64_t)tb > C.opt_tbits_max) tb = (uint32_t)(C.opt_tbits_max < 16 ? 16 : C.opt_tbits_max);
        const double room = 0.6 * (double)free_b - 4.0 * (double)(1ull << tb);
        const double per_chunk = 2300.0 * g.chunk_size + 8.0 * g.chunk_size * 4;
        if (room > per_chunk && (double)C.batch * per_chunk > room) C.batch = (uint32_t)(room / per_chunk);
    }
    C.rkhash_len = (unsigned long long)C.batch * g.chunk_size + g.feed + 256 + 1024 + 2048;
    if (C.rkhash_len > n + 2048) C.rkhash_len = n + 2048;
    DEVALLOC(C.rkhash, C.rkhash_len * 4 + 16);
    C.syms_stride = 3ull * g.chunk_size + 4096;          // <= 3 symbols per input byte
    C.bits_stride = 2ull * g.chunk_size + 64;            // <= 13 raw bits per input byte
    C.frame_stride = 12 + C.bits_stride + 16 + 2 * C.syms_stride;
    DEVALLOC(C.syms, C.batch * C.syms_stride * 4);
    DEVALLOC(C.scratch, C.batch * C.syms_stride * 4);
    DEVALLOC(C.bits, C.batch * C.bits_stride);
    DEVALLOC(C.frames, C.batch * C.frame_stride);
    DEVALLOC(C.fmeta, C.batch * sizeof(FrameMeta));
    DEVALLOC(C.dst_off, C.batch * sizeof(unsigned long long));

    // worker mode: pre-filter tables, per-launch hand-off arrays, bins
    C.workers = true;
    C.nheads = 1u << (32 - g.bt_shift);
    {   // The persistent launch needs every block resident at once (the stages and the worker lanes wait on each other):
        // 512-thread blocks with > 80 KB of LDS, one per CU.  Fewer CUs than blocks (a partitioned or masked device)
        // would spin until the timeouts fire, so the worker blocks are clamped to what the device holds.
        int cus = C.cu_count;
        if (!cus) { hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, C.device)); cus = prop.multiProcessorCount; C.cu_count = cus; }
        const int64_t room = (int64_t)cus - (int64_t)pipeline2_role_blocks();
        if (room < 1) return set_err(NLZM_HIP_E_ARG, "device has %d CUs: the pipeline needs at least %u", cus, pipeline2_role_blocks() + 1);
        if (C.opt_worker_blocks > room) C.opt_worker_blocks = room;
    }
    {   // one bin per worker lane (or per head when there are fewer heads than lanes)
        const unsigned long long lanes = (unsigned long long)(C.opt_worker_blocks < 1 ? 1 : C.opt_worker_blocks) * (unsigned long long)C.opt_worker_threads;
        if (lanes < C.nheads) C.nheads = (uint32_t)lanes;
    }
    C.run_worker_blocks = (uint32_t)(C.opt_worker_blocks < 1 ? 1 : C.opt_worker_blocks);
    C.run_worker_threads = (uint32_t)C.opt_worker_threads;
    if (C.workers) {
        const unsigned long long bpos = (unsigned long long)C.batch * g.chunk_size;
        uint32_t lg = 1; while ((1ull << lg) < bpos) lg++;
        // (32 slots per window position: a slot taken by another 65-gram of the window is a false mark.  Up to 2^33 slots -- 32 GiB for the 1e9-byte
        //  stream at -window:28, which had the 32-bit hash's 2^32 until round 5: 6 % false marks at depth instead of 3 %)
        C.t_bits = g.wbits + 5 > 33 ? 33 : (g.wbits + 5 < 16 ? 16 : g.wbits + 5);
        {   // (... and by the input: 2^tbits_per entries per position (default 16; the 1e9-byte stream at -window:28 has four, the cap): a denser table marks more
            //  positions as undecided -- 300 MB with four instead of eight entries per position waited twice as long for BT4 results;
            //  every entry is cleared when a stream begins, which is what opening a set of 32 blocks spent most of its time on)
            uint32_t lgn = 1; while ((1ull << lgn) < g.n) lgn++;
            const uint32_t want = lgn + (uint32_t)C.opt_tbits_per;
            if (want < C.t_bits) C.t_bits = want < 16 ? 16 : want;
        }
        if ((int64_t)C.t_bits > C.opt_tbits_max) C.t_bits = (uint32_t)(C.opt_tbits_max < 16 ? 16 : C.opt_tbits_max);   // (smaller: only more `unc` marks)
        C.m_bits = lg + 6 > 28 ? 28 : lg + 6;
        // (a stream that is no longer than its window -- every stream of a block set, whose window the reference shrinks to the block, :1716-1718 --
        //  never meets an earlier position outside the window: one bit per slot says all a 32-bit position would; 2 GB -> 64 MB per stream of the bench's set)
        C.t_bitmap = g.n <= (unsigned long long)g.wmask + 1 ? 1u : 0u;
        const auto t_bytes = [&]() { return C.t_bitmap ? (((size_t)1 << C.t_bits) / 8 < 4 ? (size_t)4 : ((size_t)1 << C.t_bits) / 8) : (size_t)4 << C.t_bits; };
        if (C.pool) DEVALLOC(C.pf_T, t_bytes());
        else {
            // (a device with less free memory than the table wants: a smaller table only marks more positions as undecided)
            for (;;) {
                const int rc_t = dev_alloc(C, &C.pf_T, t_bytes());
                if (!rc_t) break;
                if (rc_t != NLZM_HIP_E_NOMEM || C.t_bits <= 28) return rc_t;
                (void)hipGetLastError();
                C.t_bits--;
            }
        }
        DEVALLOC(C.pf_M, (size_t)4 << C.m_bits);
        DEVFILL(hipMemsetAsync(C.pf_T, 0, t_bytes(), C.st));
        DEVFILL(hipMemsetAsync(C.pf_M, 0xFF, (size_t)4 << C.m_bits, C.st));
        DEVALLOC(C.pf_h, bpos * 4);
        DEVALLOC(C.pf_h2, bpos * 4);
        DEVALLOC(C.pf_c1, bpos);
        DEVALLOC(C.unc, bpos + 16);
        // hand-off arrays between workgroups (written with sc1 stores, read with sc1 loads: plain device memory)
        DEVALLOC(C.bt_ready, bpos * 4 * kBtRec);
        // a position's pairs beyond the four in its record: the worst case (256 pairs, 2 KiB per position) reserved for a single stream; the
        // streams of a block set reserve 32 pairs (256 bytes) and take extension blocks from an arena for the positions that have more
        // (nlzm_core.h, bt_pair_ptr; a launch that uses the arena up fails with an error, it never drops a pair)
        C.pstride = C.pool ? 32u : kBtMaxPairs;
        C.ext_cap = C.pstride < kBtMaxPairs ? (uint32_t)(bpos / 64 + 1024) : 0u;
        if (C.ext_cap && C.opt_block_ext_blocks >= 0) C.ext_cap = (uint32_t)(C.opt_block_ext_blocks < 1 ? 1 : C.opt_block_ext_blocks);
        DEVALLOC(C.bt_pairs, bpos * (2ull * C.pstride * 4));
        if (C.ext_cap) DEVALLOC(C.bt_ext, (size_t)C.ext_cap * (2ull * (kBtMaxPairs - C.pstride) * 4));
        DEVALLOC(C.bt_flag, bpos * 4);
        DEVALLOC(C.abort_word, 4);
        DEVALLOC(C.bin_off, (size_t)C.batch * (C.nheads + 1) * 4);
        DEVALLOC(C.bin_cur, (size_t)C.batch * C.nheads * 4);
        DEVALLOC(C.bin_pos, bpos * 8);
        DEVALLOC(C.wcnt, sizeof(WorkerCounters));
        DEVFILL(hipMemsetAsync(C.wcnt, 0, sizeof(WorkerCounters), C.st));
        DEVALLOC(C.bt_undo, (size_t)C.nheads * worker_undo_bytes_per_lane());     // (6 KB per lane)
        {   // hot bins: the waves of a worker block behind its bin-taking lanes (none in block mode, where every lane takes bins)
            int64_t hw = C.opt_hot_waves;
            const int64_t spare = (512 - C.opt_worker_threads) / 64;
            if (hw > spare) hw = spare;
            C.hot_max = hw > 0 ? (uint32_t)(hw * (C.opt_worker_blocks < 1 ? 1 : C.opt_worker_blocks)) : 0u;
            if (C.hot_max) {
                DEVALLOC(C.hot_of_bin, (size_t)C.nheads * 4);
                DEVALLOC(C.hot_list, ((size_t)C.hot_max + 1) * 4);
                DEVALLOC(C.hot_undo, (size_t)C.hot_max * worker_hot_undo_bytes_per_wave());
            }
        }
    }

    // hand-off between the finder, table and parser stages
    DEVALLOC(C.v2_ft, (size_t)v2::kFtRing * v2::kFtStride * 4);
    DEVALLOC(C.v2_tp, (size_t)v2::kTpRing * v2::kTpStride * 4);
    DEVALLOC(C.v2_tf, (size_t)v2::kTpRing * v2::kTfStride * 4);
    DEVALLOC(C.v2_state, sizeof(v2::StateV2));
    DEVALLOC(C.v2_hx, sizeof(v2::Hx));
    if (C.opt_helper) DEVALLOC(C.v2_hb, v2::kHelpers * sizeof(v2::HelpBox));
    DEVFILL(hipMemsetAsync(C.v2_state, 0, sizeof(v2::StateV2), C.st));
    C.v2_launch_no = 0;

    if (C.double_sets) {
        if (!C.pool) return set_err(NLZM_HIP_E_ARG, "a second launch set needs a pool");
        const unsigned long long bpos = (unsigned long long)C.batch * g.chunk_size;
        Ctx::LaunchSet &A = C.alt;
        DEVALLOC(A.rkhash, C.rkhash_len * 4 + 16);
        DEVALLOC(A.syms, C.batch * C.syms_stride * 4);
        DEVALLOC(A.bits, C.batch * C.bits_stride);
        DEVALLOC(A.fmeta, C.batch * sizeof(FrameMeta));
        DEVALLOC(A.unc, bpos + 16);
        DEVALLOC(A.bt_ready, bpos * 4 * kBtRec);
        DEVALLOC(A.bt_flag, bpos * 4);
        DEVALLOC(A.abort_word, 4);
        DEVALLOC(A.bin_off, (size_t)C.batch * (C.nheads + 1) * 4);
        DEVALLOC(A.bin_pos, bpos * 8);
        if (C.hot_max) {
            DEVALLOC(A.hot_of_bin, (size_t)C.nheads * 4);
            DEVALLOC(A.hot_list, ((size_t)C.hot_max + 1) * 4);
        }
        DEVALLOC(A.snap, sizeof(v2::RoundSnap));
        DEVALLOC(C.snap, sizeof(v2::RoundSnap));
    }

    // stream header (:1762-1766)
    const uint8_t hdr[4] = { (uint8_t)(g.wbits >> 8), (uint8_t)g.wbits, (uint8_t)(g.frame_bits >> 8), (uint8_t)g.frame_bits };
    DEVFILL(hipMemcpyAsync(C.d_dst, hdr, 4, hipMemcpyHostToDevice, C.st));
    C.out_pos = 4;

    C.pooled = C.pool != nullptr;
    if (C.pool && C.pool->measuring) { C.pooled = false; return 0; }
    HIPCHK(hipStreamSynchronize(C.st));
    C.open = true;
    return 0;
}

// One launch's worth of a stream, in three parts so that several streams can share ONE persistent launch:
//   step_pre   pre-pass kernels and the hand-off arrays of chunks [c0, c1) on the stream's own HIP stream
//   (launch)   pipeline_kernel for this stream alone, or pipeline_multi_kernel for a group of streams
//   step_post  frame coder, frame lengths back to the host, checks, gather into the output
// ahead: the launch before this one may still be on the device (block mode; the other launch set is made the current one, and
// the progress words are set by round_open_kernel on the launch's own HIP stream instead of a copy here)
int step_pre(Ctx &C, uint32_t todo, StepPlan &P, bool ahead = false)
{
    const Geom &g = C.g;
    if (ahead) { if (!C.double_sets) return set_err(NLZM_HIP_E_ARG, "no second launch set"); swap_sets(C); }
    const uint32_t c0 = C.pre_chunk, nb = todo < C.batch ? todo : C.batch, c1 = c0 + nb;
    C.pre_chunk = c1;
    P.c0 = c0; P.c1 = c1; P.nb = nb;
    memcpy(P.ev, C.set_idx ? C.ev_b : C.ev, sizeof P.ev);
    P.syms = C.syms; P.bits = C.bits; P.fmeta = C.fmeta; P.abort_word = C.abort_word; P.snap = ahead ? C.snap : nullptr;
    Globals &G = P.G;
    memset(&G, 0, sizeof G);
    G.in = C.d_in; G.rkhash = nullptr; G.ht2 = C.ht2; G.ht3 = C.ht3; G.rk_table = C.rk_table;
    G.bt_heads = C.bt_heads; G.bt_tree = C.bt_tree; G.persist = C.persist;
    G.syms = C.syms; G.syms_stride = C.syms_stride; G.bits = C.bits; G.bits_stride = C.bits_stride;
    G.fmeta = C.fmeta; G.chunk0 = c0;
    G.cap_words = C.cap_words; G.cap_cap = C.cap_cap; G.cap_lo = C.cap_lo; G.cap_hi = C.cap_hi; G.cap_used = C.cap_used;
    G.workers = C.workers ? 1 : 0;
    const unsigned long long a0 = (unsigned long long)c0 * g.chunk_size;
    unsigned long long a1 = (unsigned long long)c1 * g.chunk_size;
    if (a1 > g.n) a1 = g.n;
    G.batch_a0 = (uint32_t)a0;
    {   // pre-pass: RK256 hash of every window the launch can touch (catch-up inserts reach back < 512 bytes).
        // Feed mode: the hashes of windows beyond the launch's last position (up to a1 + feed + 511) may be computed from bytes
        // whose upload is still in flight (feed_chunks_ready guarantees a1 + (feed - chunk) + 1024 only).  They are never used:
        // the finder reads rkhash[p] for p < a1 only, and the next launch hashes again from its own a0 - 1024 on.
        const unsigned long long lo = a0 > 1024 ? a0 - 1024 : 0;
        unsigned long long hi = a1 + g.feed + 256;
        if (hi + 255 > g.n) hi = g.n >= 255 ? g.n - 255 : 0;
        HIPCHK(hipEventRecord(P.ev[7], C.st));
        if (hi > lo && hi - lo > C.rkhash_len) return set_err(NLZM_HIP_E_ARG, "launch of %u chunks is larger than the stream was opened for", nb);
        G.rkhash = C.rkhash - lo;                       // rkhash[a] for a in [lo, hi): the launch's own array
        if (g.n >= 256 && hi > lo) launch_rk_hash(C.d_in, g.n, lo, hi, C.rkhash - lo, C.st);
    }
    HIPCHK(hipEventRecord(P.ev[5], C.st));
    if (C.workers) {
        const unsigned long long cnt = a1 - a0;
        G.bt_ready = C.bt_ready; G.bt_pairs = C.bt_pairs; G.bt_flag = C.bt_flag; G.unc = C.unc;
        G.bt_pstride = C.pstride; G.bt_ext = C.bt_ext; G.bt_ext_cap = C.ext_cap; G.bt_ext_cur = &C.v2_hx->ext_cur;
        G.bin_off = C.bin_off; G.bin_pos = C.bin_pos; G.nheads = C.nheads; G.wthreads = C.run_worker_threads;
        G.abort_word = C.abort_word; G.wcnt = C.wcnt; G.bt_undo = C.bt_undo;
        HIPCHK(hipMemsetAsync(C.bt_ready, 0, cnt * 4 * kBtRec, C.st));
        HIPCHK(hipMemsetAsync(C.bt_flag, 0, cnt * 4, C.st));
        HIPCHK(hipMemsetAsync(C.abort_word, 0, 4, C.st));
        HIPCHK(hipMemsetAsync(C.bin_off, 0, (size_t)nb * (C.nheads + 1) * 4, C.st));
        launch_prefilter(C.d_in, g.n, (uint32_t)a0, (uint32_t)a1, g.wmask, C.t_bits, C.t_bitmap, C.m_bits, C.pf_T, C.pf_M, C.pf_h, C.pf_h2,
                         C.pf_c1, C.unc, C.st);
        launch_bin(C.d_in, g, c0, nb, C.nheads, C.bin_off, C.bin_cur, C.bin_pos, C.unc, (uint32_t)a0, C.st);
        if (C.hot_max) {
            const unsigned long long lpos = (unsigned long long)C.batch * g.chunk_size;
            // (the streams of a block set have a hundred heads to a lane: there a wave pays from positions / 480 on -- 112.6 -> 115.6 MB/s against 8,192)
            const double by_pace = C.pool ? (double)lpos / 480.0 : (C.last_launch_ms > 0 ? 24.0 * C.last_launch_ms : (double)lpos / 240.0);
            const uint32_t hot_min = C.opt_hot_min > 0 ? (uint32_t)C.opt_hot_min : (uint32_t)(by_pace < 512 ? 512 : (by_pace > 1e9 ? 1e9 : by_pace));
            launch_hot_select(C.bin_off, nb, C.nheads, C.hot_max, hot_min, C.hot_of_bin, C.hot_list, C.wcnt, C.st);
            G.hot_of_bin = C.hot_of_bin; G.hot_list = C.hot_list; G.hot_undo = C.hot_undo;
        }
    }
    {   // progress words of the stages: everything before the launch's first position is done
        v2::Hx &h = C.hx_host;          // (lives until the copy has been made)
        memset(&h, 0, sizeof h);
        h.f_pos = h.t_pos = h.t_out = h.p_pos = (uint32_t)a0;
        h.p_seg = ((unsigned long long)(uint32_t)a0 << 32) | (uint32_t)a0;
        if (!ahead) HIPCHK(hipMemcpyAsync(C.v2_hx, &h, sizeof h, hipMemcpyHostToDevice, C.st));
        P.V.ft = C.v2_ft; P.V.tp = C.v2_tp; P.V.tf = C.v2_tf; P.V.hx = C.v2_hx; P.V.state = C.v2_state; P.V.hb = C.v2_hb;
        G.progress = &C.v2_hx->f_pos;
        G.test_fail = C.opt_test_fail_launch >= 0 && (int64_t)C.v2_launch_no == C.opt_test_fail_launch ? 1u : 0u;
        G.table_shape = (uint32_t)C.opt_table_shape; G.launch_par = (uint32_t)(C.v2_launch_no++ & 1u);
    }
    HIPCHK(hipEventRecord(P.ev[6], C.st));
    return 0;
}

// pipe_ms: time of the persistent launch when it was not this stream's own (group launch), else < 0
// (in three parts, so that the streams of a block set have their frame coders and gathers in flight together: every part of
//  every stream is queued before the next part waits for any of them)
int step_post_issue(Ctx &C, const StepPlan &P)
{
    const uint32_t nb = P.nb;
    C.post_hm.resize(nb); C.post_hoff.resize(nb);
    std::vector<FrameMeta> &hm = C.post_hm;
    Persist &Pst = C.post_P;
    uint32_t &aborted = C.post_aborted;
    aborted = 0;
    HIPCHK(hipEventRecord(P.ev[1], C.st));
    launch_rans(P.syms, C.syms_stride, P.bits, C.bits_stride, P.fmeta, C.scratch, C.syms_stride, C.frames,
                C.frame_stride, (uint32_t)C.frame_stride, nb, C.st);
    HIPCHK(hipEventRecord(P.ev[2], C.st));
    HIPCHK(hipMemcpyAsync(hm.data(), P.fmeta, nb * sizeof(FrameMeta), hipMemcpyDeviceToHost, C.st));
    if (P.snap) {           // (the stream's next launch may be running: the copy round_close_kernel made)
        HIPCHK(hipMemcpyAsync(&C.post_snap, P.snap, sizeof(v2::RoundSnap), hipMemcpyDeviceToHost, C.st));
        return 0;
    }
    HIPCHK(hipMemcpyAsync(&Pst, C.persist, sizeof Pst, hipMemcpyDeviceToHost, C.st));
    if (C.workers) HIPCHK(hipMemcpyAsync(&aborted, P.abort_word, 4, hipMemcpyDeviceToHost, C.st));
    HIPCHK(hipMemcpyAsync(&C.hx_host, C.v2_hx, sizeof(v2::Hx), hipMemcpyDeviceToHost, C.st));
    return 0;
}
int step_post_check(Ctx &C, const StepPlan &P)
{
    const Geom &g = C.g;
    const uint32_t c0 = P.c0, c1 = P.c1, nb = P.nb;
    std::vector<FrameMeta> &hm = C.post_hm;
    std::vector<unsigned long long> &hoff = C.post_hoff;
    HIPCHK(hipStreamSynchronize(C.st));
    HIPCHK(hipGetLastError());
    if (P.snap) { C.post_P.error = C.post_snap.error; C.post_P.next_chunk = C.post_snap.next_chunk; C.post_aborted = C.post_snap.aborted; C.hx_host = C.post_snap.hx; }
    const Persist &Pst = C.post_P;
    const uint32_t aborted = C.post_aborted;
    float rk_ms = 0, pre_ms = 0;
    HIPCHK(hipEventElapsedTime(&rk_ms, P.ev[7], P.ev[5]));
    HIPCHK(hipEventElapsedTime(&pre_ms, P.ev[5], P.ev[6]));
    C.tm.prep_ms += rk_ms + pre_ms; C.tm.prep_launches += C.workers ? 5 : 1; C.tm.total_ms += rk_ms + pre_ms;
    C.arena_out = false;
    if (Pst.error || C.hx_host.err) {
        // the first error any stage raised, and where every stage was when it left (nlzm_v2.h: raise(), Hx::dbg)
        const v2::Hx &h = C.hx_host;
        WorkerCounters wc{};
        if (C.workers) (void)hipMemcpy(&wc, C.wcnt, sizeof wc, hipMemcpyDeviceToHost);
        return set_err(NLZM_HIP_E_KERNEL,
                       "device error %u in chunks [%u,%u) (parser stopped at chunk %u): raised by stage %u at wait site %u, position %u, saw %u %u | "
                       "progress: finder %u, table in %u out %u, parser %u, segment %u covered to %u | "
                       "finder: block at %u reach %u top entry %u d %u end %u prev_nice %u seg_s %u rk_len %u t_pos_seen %u err %u base %u | "
                       "table: cursor %u turn %u carry_seq %u f_seen %u p_seen %u carry_n %u | "
                       "parser: chunk %u segment %u block node %u max_parse %u staged to %u t_out_seen %u err %u | "
                       "worker lanes left waiting %llu, first of them at position %u",
                       Pst.error ? Pst.error : h.err, c0, c1, Pst.next_chunk, h.err_info[0], h.err_info[1], h.err_info[2], h.err_info[3], h.err_info[4],
                       h.f_pos, h.t_pos, h.t_out, h.p_pos, (uint32_t)(h.p_seg >> 32), (uint32_t)h.p_seg,
                       h.dbg[0][0], h.dbg[0][1], h.dbg[0][2], h.dbg[0][3], h.dbg[0][4], h.dbg[0][5], h.dbg[0][6], h.dbg[0][7], h.dbg[0][8], h.dbg[0][9], h.dbg[0][10],
                       h.dbg[1][0], h.dbg[1][1], h.dbg[1][2], h.dbg[1][3], h.dbg[1][4], h.dbg[1][5],
                       h.dbg[2][0], h.dbg[2][1], h.dbg[2][2], h.dbg[2][3], h.dbg[2][4], h.dbg[2][5], h.dbg[2][6],
                       wc.stuck_lanes, wc.stuck_lanes ? (uint32_t)~(uint32_t)wc.stuck_pos_inv : 0u);
    }
    // (the worker lanes drop a pair that finds no extension block and go on: the cursor says how many blocks were asked for)
    C.arena_out = C.ext_cap && C.hx_host.ext_cur > C.ext_cap;
    if (C.arena_out) return set_err(NLZM_HIP_E_KERNEL, "the extension arena of the BT4 pair lists (%u blocks per launch) was used up in chunks [%u,%u): more positions with over %u "
                                     "record-setters than a block set reserves for", C.ext_cap, c0, c1, C.pstride);
    if (aborted) return set_err(NLZM_HIP_E_KERNEL, "worker lanes aborted (code %u) in chunks [%u,%u)", aborted, c0, c1);
    if (Pst.next_chunk != c1) return set_err(NLZM_HIP_E_KERNEL, "master stopped at chunk %u, expected %u", Pst.next_chunk, c1);
    unsigned long long pos = C.out_pos;
    for (uint32_t f = 0; f < nb; f++) {
        // the reference asserts that the frame fits its buffer (:592, :610); the first one is 4 bytes shorter (:1784)
        const uint32_t room = g.frame_size - ((c0 + f) == 0 ? 4 : 0);
        if (hm[f].out_len > room)
            return set_err(NLZM_HIP_E_KERNEL, "frame %u is %u bytes: the reference would assert (:610)", c0 + f, hm[f].out_len);
        hoff[f] = pos; pos += hm[f].out_len;
    }
    if (pos + 4 > C.dst_cap) return set_err(NLZM_HIP_E_CAPACITY, "dst_cap %llu too small", (unsigned long long)C.dst_cap);
    if (C.want_frame >= (int64_t)c0 && C.want_frame < (int64_t)c1) {
        const uint32_t f = (uint32_t)(C.want_frame - c0);
        C.got_meta = hm[f];
        C.got_syms.resize(hm[f].nsyms); C.got_bits.resize(hm[f].nbits_bytes);
        HIPCHK(hipMemcpy(C.got_syms.data(), P.syms + f * C.syms_stride, hm[f].nsyms * 4ull, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(C.got_bits.data(), P.bits + f * C.bits_stride, hm[f].nbits_bytes, hipMemcpyDeviceToHost));
        C.got = true;
    }
    HIPCHK(hipMemcpyAsync(C.dst_off, hoff.data(), nb * sizeof(unsigned long long), hipMemcpyHostToDevice, C.st));
    HIPCHK(hipEventRecord(P.ev[3], C.st));
    launch_gather(C.frames, C.frame_stride, C.dst_off, P.fmeta, C.d_dst, nb, C.st);
    HIPCHK(hipEventRecord(P.ev[4], C.st));
    C.post_pos = pos;
    return 0;
}
int step_post_done(Ctx &C, const StepPlan &P, float pipe_ms)
{
    const unsigned long long pos = C.post_pos;
    const uint32_t c1 = P.c1;
    HIPCHK(hipStreamSynchronize(C.st));
    float a = pipe_ms, b = 0, c = 0;
    if (pipe_ms < 0) HIPCHK(hipEventElapsedTime(&a, P.ev[0], P.ev[1]));
    HIPCHK(hipEventElapsedTime(&b, P.ev[1], P.ev[2]));
    HIPCHK(hipEventElapsedTime(&c, P.ev[3], P.ev[4]));
    C.tm.match_parse_ms += a; C.tm.match_parse_launches++;
    if (a > 0) C.last_launch_ms = a;
    C.tm.rans_ms += b + c; C.tm.rans_launches++;
    C.tm.total_ms += a + b + c;
    C.out_pos = pos;
    C.next_chunk = c1;
    return 0;
}
int step_post(Ctx &C, const StepPlan &P, float pipe_ms)
{
    int rc = step_post_issue(C, P);
    if (!rc) rc = step_post_check(C, P);
    if (!rc) rc = step_post_done(C, P, pipe_ms);
    return rc;
}

int stream_step(Ctx &C, uint32_t max_chunks, uint64_t *in_done, uint64_t *out_done, int *finished)
{
    if (!C.open) return set_err(NLZM_HIP_E_ARG, "no open stream");
    const Geom &g = C.g;
    uint32_t todo = g.nchunks - C.next_chunk;
    if (max_chunks && todo > max_chunks) todo = max_chunks;
    while (todo) {
        StepPlan P;
        int rc = step_pre(C, todo, P);
        if (rc) return rc;
        HIPCHK(hipEventRecord(C.ev[0], C.st));
        launch_pipeline2(g, P.G, P.V, P.c0, P.c1, C.run_worker_blocks, C.st);
        rc = step_post(C, P, -1.0f);
        if (rc) return rc;
        todo -= P.nb;
    }
    if (in_done) {
        const unsigned long long d = (unsigned long long)C.next_chunk * g.chunk_size;
        *in_done = d < g.n ? d : g.n;
    }
    if (out_done) *out_done = C.out_pos;
    if (finished) *finished = C.next_chunk >= g.nchunks;
    return 0;
}

int refresh_stats(Ctx &C)
{
    Persist P;
    HIPCHK(hipMemcpy(&P, C.persist, sizeof P, hipMemcpyDeviceToHost));
    nlzm_hip_stats &s = C.stats;
    s.out_bytes = C.out_pos;
    s.bt_calls = P.cnt.bt_calls; s.bt_tests = P.cnt.bt_tests; s.cmp_bytes = P.cnt.cmp_bytes; s.ht_rows = P.cnt.ht_rows;
    s.rk_probes = P.cnt.rk_probes; s.rk_inserts = P.cnt.rk_inserts; s.positions = P.cnt.positions;
    s.nice_positions = P.cnt.nice_positions; s.segments = P.cnt.segments; s.n_literal = P.cnt.n_literal;
    s.n_dict = P.cnt.n_dict; s.n_rep = P.cnt.n_rep; s.rans_syms = P.cnt.rans_syms; s.bit_ops = P.cnt.bit_ops;
    s.frames = P.cnt.frames; s.shifts = P.cnt.shifts; s.uncertain_positions = P.cnt.uncertain_positions;
    memcpy(C.prof_last, P.prof, sizeof C.prof_last);
    {
        const double n = (double)(P.cnt.positions ? P.cnt.positions : 1);
        C.acct[0] = P.prof[17] / n; C.acct[1] = P.prof[16] / n; C.acct[2] = P.prof[25] / n; C.acct[3] = P.prof[19] / n; C.acct[4] = P.prof[18] / n;
        C.acct[5] = P.prof[21] / n; C.acct[6] = P.prof[20] / n; C.acct[7] = P.prof[24] / n;
    }
    if (C.opt_report) {
        // per-stage accounting of the three-stage pipeline (Persist::prof, filled by nlzm_v2.h)
        const double n = (double)(P.cnt.positions ? P.cnt.positions : 1);
        fprintf(stderr, "cycles/position  finder: total %.0f wait %.0f (%.0f of it for worker results) | table: total %.0f wait %.0f | parser: total %.0f wait %.0f (block set-up %.0f, passes %.0f, emit %.0f)\n",
                P.prof[17] / n, P.prof[16] / n, P.prof[25] / n, P.prof[19] / n, P.prof[18] / n, P.prof[21] / n, P.prof[20] / n, P.prof[23] / n, P.prof[24] / n, P.prof[22] / n);
        fprintf(stderr, "finder: %llu blocks (%.1f positions each); cut by: nice %llu, new top entry %llu, RK candidate %llu, RK catch-up %llu, same worker bin %llu, other %llu\n",
                P.prof[0], n / (double)(P.prof[0] ? P.prof[0] : 1), P.prof[1], P.prof[2], P.prof[3], P.prof[4], P.prof[12], P.prof[5]);
        fprintf(stderr, "table: %llu blocks, %llu on the slow path; parser: %llu blocks (%.1f nodes each), %.2f passes per block (%.0f cycles per pass), mask fills %llu, probe rounds %llu, re-sampled %llu\n",
                P.prof[6], P.prof[7], P.prof[8], n / (double)(P.prof[8] ? P.prof[8] : 1), (double)P.prof[13] / (double)(P.prof[8] ? P.prof[8] : 1),
                (double)P.prof[24] / (double)(P.prof[13] ? P.prof[13] : 1), P.prof[9], P.prof[10], P.prof[11]);
        fprintf(stderr, "table: %llu launches with %u-entry fronts on %u waves (the others: %u on %u), the shape changed %llu times\n", P.prof[114], v2::kFrCapWide, v2::kTWWide, v2::kFrCap, v2::kTW, P.prof[113]);
        fprintf(stderr, "table: blocks in which some position's front had more than 8 / 12 / 16 / 20 / 24 / the launch's capacity of entries at some step of the scan: %.2f / %.2f / %.2f / %.3f / %.3f / %.3f %%\n",
                100.0 * P.prof[105] / (P.prof[6] ? P.prof[6] : 1), 100.0 * P.prof[106] / (P.prof[6] ? P.prof[6] : 1), 100.0 * P.prof[107] / (P.prof[6] ? P.prof[6] : 1),
                100.0 * P.prof[108] / (P.prof[6] ? P.prof[6] : 1), 100.0 * P.prof[109] / (P.prof[6] ? P.prof[6] : 1), 100.0 * P.prof[7] / (P.prof[6] ? P.prof[6] : 1));
        fprintf(stderr, "finder: RK256 entries cut short by the uint16 length parameter that became the growing top entry: %llu; that ended exactly where another entry ends: %llu (%llu of them the nearer one)\n",
                P.prof[115], P.prof[116], P.prof[117]);
        fprintf(stderr, "finder: starts of nice regions whose segment the stage knew itself, ahead of the parser's word: %llu; that it had to wait for: %llu\n", P.prof[118], P.prof[119]);
        fprintf(stderr, "finder: worker results not there at the first look: %llu of positions whose call is the finder's decision (unc), %llu of others\n", P.prof[28], P.prof[29]);
        fprintf(stderr, "finder: blocks that had to wait for a worker result: %llu (%.0f cycles each); late results of hot bins' waves %llu, late results at lane 0 (the position the block before was cut at) %llu\n",
                P.prof[112], (double)P.prof[25] / (double)(P.prof[112] ? P.prof[112] : 1), P.prof[110], P.prof[111]);
        fprintf(stderr, "parser: waited for its record loader %llu times, the table stage %.0f positions ahead on average then\n", P.prof[26], (double)P.prof[27] / (double)(P.prof[26] ? P.prof[26] : 1));
        if (P.prof[96] || P.prof[100])
            fprintf(stderr, "helper parser: %llu jobs posted, %llu taken over (%llu nodes = %.1f %% of the positions), the parser stage waited %.0f cycles per position for it; "
                            "helper: %llu jobs seen, %llu done, %llu blocks (%.2f passes each), waited %.0f cycles per position for records\n",
                    P.prof[96], P.prof[97], P.prof[98], 100.0 * P.prof[98] / n, P.prof[99] / n, P.prof[100], P.prof[101], P.prof[102],
                    (double)P.prof[103] / (double)(P.prof[102] ? P.prof[102] : 1), P.prof[104] / n);
        if (P.prof[88]) fprintf(stderr, "finder sections (cycles/position, profile build): predict %.0f, own loads %.0f, HT rows %.0f, candidates + jobs %.0f, record + RK256 %.0f, "
                                "BT4 results (wait included) %.0f, verify %.0f, commit %.0f\n", P.prof[88] / n, P.prof[89] / n, P.prof[90] / n, P.prof[91] / n, P.prof[92] / n,
                                P.prof[93] / n, P.prof[94] / n, P.prof[95] / n);
        if (P.prof[44]) fprintf(stderr, "table stage sections (cycles/position summed over the waves, profile build): gather %.0f, scan %.0f, waiting for the carry %.0f, carry merge (the part in block order) %.0f, records %.0f\n",
                                P.prof[44] / n, P.prof[45] / n, P.prof[47] / n, P.prof[46] / n, P.prof[55] / n);
        if (P.prof[32]) {
            const double np = (double)(P.prof[13] ? P.prof[13] : 1);
            fprintf(stderr, "parser, cycles per pass (profile build): relax waves %.0f %.0f %.0f, probe wave %.0f (of it: sets that changed %.0f, mask fills %.0f), update %.0f, "
                            "barrier waits per wave %.0f %.0f %.0f %.0f; block end %.0f cycles/position\n",
                    P.prof[32] / np, P.prof[33] / np, P.prof[34] / np, P.prof[35] / np, P.prof[43] / np, P.prof[41] / np, P.prof[40] / np,
                    P.prof[36] / np, P.prof[37] / np, P.prof[38] / np, P.prof[39] / np, P.prof[42] / n);
            fprintf(stderr, "parser, cycles per pass by wave 0..7 (profile build): work");
            for (int w = 0; w < 8; w++) fprintf(stderr, " %.0f", P.prof[64 + w] / np);
            fprintf(stderr, " | barrier wait");
            for (int w = 0; w < 8; w++) fprintf(stderr, " %.0f", P.prof[72 + w] / np);
            fprintf(stderr, " | update");
            for (int w = 0; w < 8; w++) fprintf(stderr, " %.0f", P.prof[80 + w] / np);
            fprintf(stderr, "\n");
            const double nbk = (double)(P.prof[8] ? P.prof[8] : 1);
            fprintf(stderr, "parser loader wave, cycles per block set-up: block size + barrier %.0f, re-list %.0f, own edges %.0f, all edges %.0f, literal scan + clear + barrier %.0f\n",
                    P.prof[56] / nbk, P.prof[57] / nbk, P.prof[58] / nbk, P.prof[59] / nbk, P.prof[60] / nbk);
            fprintf(stderr, "parser wave 0, cycles per pass: relax %.0f, probe %.0f, clear %.0f | update: keys + cost scan %.0f, membership %.0f, winner sets %.0f, rest %.0f\n",
                    P.prof[48] / np, P.prof[49] / np, P.prof[50] / np, P.prof[51] / np, P.prof[52] / np, P.prof[53] / np, P.prof[54] / np);
        }
    }
    if (C.workers) {
        WorkerCounters wc;
        HIPCHK(hipMemcpy(&wc, C.wcnt, sizeof wc, hipMemcpyDeviceToHost));
        s.bt_calls += wc.bt_calls; s.bt_tests += wc.bt_tests; s.cmp_bytes += wc.cmp_bytes;
        C.last_dry_runs = wc.dry_runs; C.last_flag_waits = wc.flag_waits;
        C.wc_last = wc;
        if (C.opt_report)
            fprintf(stderr, "worker lanes: %llu calls made with their fate open (at and behind a position not decided yet), %llu decisions that took calls back, %llu calls made again for it\n",
                    wc.dry_runs, wc.spec_calls, wc.spec_good);
        if (C.opt_report && C.hot_max)
            fprintf(stderr, "hot bins (a wave each): %llu over all launches, %llu of %llu calls made by their waves\n", wc.hot_bins, wc.hot_calls, wc.bt_calls);
        if (C.opt_report && C.hot_max && wc.hot_steps)
            fprintf(stderr, "hot bins' waves: %llu steps (%.1f per call); the next call could not start in %.1f %% of them (a call without its stores on its way) + %.1f %% (an assumed \"skip\" behind a \"call\" still open); lanes: %llu turns spent waiting for a decision\n",
                    wc.hot_steps, (double)wc.hot_steps / (wc.hot_calls ? wc.hot_calls : 1), 100.0 * wc.hot_blocked_dry / wc.hot_steps, 100.0 * wc.hot_blocked_risky / wc.hot_steps, wc.flag_waits);
        if (C.opt_report && C.hot_max && wc.hot_steps) {
            // (what a step was spent on is counted by the profile build only: the counting was a tenth of the step)
            fprintf(stderr, "hot bins' waves by the bin's positions in the launch (class: waves | calls, tests/call, entries skipped | steps, cycles/step; the profile build adds | %% of the steps: some lane tests "
                            "(tests per such step; lane-steps repeated for a held slot per step), taking back, every lane holds a call, next call may not start, no entry | idle steps with an undecided position open)\n");
            for (int k = 0; k < 8; k++) {
                const unsigned long long *h = wc.hot_class[k];
                if (!h[0]) continue;
                const double st = (double)(h[3] ? h[3] : 1);
                fprintf(stderr, "  %s %7u: %5llu | %10llu calls, %5.1f, %10llu | %12llu steps, %5.0f", k ? ">=" : "< ", k ? 8192u << k : 16384u, h[0], h[1], (double)h[2] / (h[1] ? h[1] : 1), h[12], h[3], (double)h[11] / st);
                if (h[14] + h[16]) {
                    fprintf(stderr, " | %4.1f (%.2f; %.2f), %4.1f, %4.1f, %4.1f, %4.1f | %4.1f\n", 100.0 * h[4] / st, (double)h[5] / (h[4] ? h[4] : 1), (double)h[6] / st, 100.0 * h[7] / st, 100.0 * h[8] / st, 100.0 * h[9] / st,
                            100.0 * h[10] / st, 100.0 * h[13] / st);
                    fprintf(stderr, "              cycles of a step by section: end of the step before + windows %.0f, oldest undecided + recovery %.0f, entries passed + start %.0f, loads until they are back %.0f, "
                                    "call start / test %.0f, call end + result %.0f, accounting + watchdogs %.0f\n", h[20] / st, h[14] / st, h[15] / st, h[16] / st, h[17] / st, h[18] / st, h[19] / st);
                } else fprintf(stderr, "\n");
            }
        }
        if (C.opt_report && wc.call_tests)
            fprintf(stderr, "worker lanes: %.0f cycles per BT4 test, %.1f tests per timed call (lane clocks, divergence included)\n",
                    (double)wc.call_cycles / wc.call_tests, (double)wc.call_tests / (wc.bt_calls ? wc.bt_calls : 1));
    }
    return 0;
}

int stream_finish(Ctx &C, uint64_t *dst_len)
{
    if (!C.open) return set_err(NLZM_HIP_E_ARG, "no open stream");
    if (C.next_chunk < C.g.nchunks) return set_err(NLZM_HIP_E_ARG, "stream not finished (%u of %u chunks)", C.next_chunk, C.g.nchunks);
    if (C.out_pos + 4 > C.dst_cap) return set_err(NLZM_HIP_E_CAPACITY, "dst_cap too small");
    HIPCHK(hipMemsetAsync(C.d_dst + C.out_pos, 0, 4, C.st));        // terminator (:1891-1895)
    C.out_pos += 4;
    HIPCHK(hipStreamSynchronize(C.st));
    const int rc = refresh_stats(C);
    if (rc) return rc;
    if (dst_len) *dst_len = C.out_pos;
    return 0;
}

}  // namespace

extern "C" {

static void dev_shutdown(DevState &D);
static int dev_init(DevState &D, int device)
{
    Ctx &C = D.ctx;
    // Block mode queues the pre-pass kernels and frame coders of 32 streams beside a persistent launch, each stream on a HIP stream of its
    // own: with the runtime's default of 4 hardware queues they would line up behind one another.  The runtime reads the variable when it
    // starts, i.e. at this process's first HIP call -- ours, unless the host program has made one already (then it has to export
    // GPU_MAX_HW_QUEUES=16 itself: include/nlzm_hip.h).  A value the caller has set is left alone.  Set, never read: the library has no
    // environment knobs of its own.
    // (once per process, and only here -- the per-device threads of nlzm_hip_compress_blocks_multi come in with the runtime long started.  What the
    //  runtime will have read is kept for nlzm_hip_get_counter("gpu_max_hw_queues_effective"): the caller's value; 16 if this call set it in time;
    //  the runtime's default of 4 if the process had the GPU open already -- /dev/kfd among its files -- when this library was first called.)
    static std::once_flag hwq_once;
    std::call_once(hwq_once, [] {
        const char *have = getenv("GPU_MAX_HW_QUEUES");
        if (have && *have) { g_hwq_effective = atoi(have); return; }
        bool kfd = false;
        if (DIR *d = opendir("/proc/self/fd")) {
            while (struct dirent *e = readdir(d)) {
                char path[64], to[64];
                snprintf(path, sizeof path, "/proc/self/fd/%s", e->d_name);
                const ssize_t k = readlink(path, to, sizeof to - 1);
                if (k > 0) { to[k] = 0; if (!strcmp(to, "/dev/kfd")) kfd = true; }
            }
            closedir(d);
        }
        if (kfd) { g_hwq_effective = 4; return; }       // (too late to matter: left alone)
        (void)setenv("GPU_MAX_HW_QUEUES", "16", 0);
        g_hwq_effective = 16;
    });
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return set_err(NLZM_HIP_E_NODEVICE, "no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return set_err(NLZM_HIP_E_ARG, "device %d out of range (%d present)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(NLZM_HIP_E_NODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    if (C.inited && C.device == device) return 0;
    if (C.inited) dev_shutdown(D);
    C.device = device;
    C.cu_count = prop.multiProcessorCount;
    HIPCHK(hipStreamCreateWithFlags(&C.st, hipStreamNonBlocking));
    for (auto &ev : C.ev) HIPCHK(hipEventCreate(&ev));
    C.inited = true;
    return 0;
}

int nlzm_hip_init(int device) { return dev_init(cur(), device); }

static void dev_shutdown(DevState &D)
{
    Ctx &C = D.ctx;
    if (!C.inited) return;
    DevState *keep = t_dev;
    t_dev = &D;
    blocks_close(true);
    t_dev = keep;
    if (D.feed.pin[0] || D.feed.st) { for (int k = 0; k < 2; k++) { if (D.feed.pin[k]) (void)hipHostFree(D.feed.pin[k]); if (D.feed.ev[k]) (void)hipEventDestroy(D.feed.ev[k]); }
                                      if (D.feed.st) (void)hipStreamDestroy(D.feed.st); D.feed = DevState::Feed{}; }
    free_stream_buffers(C);
    if (C.cap_words) { (void)hipFree(C.cap_words); C.cap_words = nullptr; }
    if (C.cap_used) { (void)hipFree(C.cap_used); C.cap_used = nullptr; }
    for (auto &ev : C.ev) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    if (C.st) { (void)hipStreamDestroy(C.st); C.st = nullptr; }
    C.inited = false;
}
void nlzm_hip_shutdown(void) { dev_shutdown(cur()); }

const char *nlzm_hip_last_error(void) { return g_err; }

uint64_t nlzm_hip_compress_bound(uint64_t n)
{
    // worst ratio frame_size/chunk_size is 16384/14848 (hist_bits <= 16)
    return 16 + 131072 + (n / 14848 + 1) * 16384;
}

void nlzm_hip_geometry(uint64_t flen, uint32_t hist_bits_req, uint32_t *hist_bits, uint32_t *frame_bits,
                       uint32_t *chunk_size, uint32_t *feed_size)
{
    Geom g;
    make_geom(flen, hist_bits_req, g);
    if (hist_bits) *hist_bits = g.wbits;
    if (frame_bits) *frame_bits = g.frame_bits;
    if (chunk_size) *chunk_size = g.chunk_size;
    if (feed_size) *feed_size = g.feed;
}

int nlzm_hip_stream_begin(const void *d_src, uint64_t n, uint32_t hist_bits_req, void *d_dst, uint64_t dst_cap)
{
    Ctx &C = g_ctx;
    if (C.own_in) { (void)hipFree(C.own_in); C.own_in = nullptr; }
    if (C.own_dst) { (void)hipFree(C.own_dst); C.own_dst = nullptr; }
    return stream_begin(C, d_src, n, hist_bits_req, d_dst, dst_cap);
}
int nlzm_hip_stream_step(uint32_t max_chunks, uint64_t *in_done, uint64_t *out_done, int *finished)
{
    return stream_step(g_ctx, max_chunks, in_done, out_done, finished);
}
int nlzm_hip_stream_finish(uint64_t *dst_len) { return stream_finish(g_ctx, dst_len); }

int nlzm_hip_compress_dev(const void *d_src, uint64_t n, uint32_t hist_bits_req, void *d_dst, uint64_t dst_cap,
                          uint64_t *dst_len)
{
    int rc = nlzm_hip_stream_begin(d_src, n, hist_bits_req, d_dst, dst_cap);
    if (rc) return rc;
    rc = stream_step(g_ctx, 0, nullptr, nullptr, nullptr);
    if (rc) return rc;
    return stream_finish(g_ctx, dst_len);
}

int nlzm_hip_compress(const uint8_t *src, uint64_t n, uint32_t hist_bits_req, uint8_t *dst, uint64_t dst_cap,
                      uint64_t *dst_len)
{
    Ctx &C = g_ctx;
    if (!C.inited) return set_err(NLZM_HIP_E_NODEVICE, "nlzm_hip_init() has not succeeded");
    if ((!src && n) || !dst || !dst_len) return set_err(NLZM_HIP_E_ARG, "null argument");
    if (n >= 0xFFFF0000ull) return set_err(NLZM_HIP_E_TOOBIG, "input too large");
    if (C.own_in) { (void)hipFree(C.own_in); C.own_in = nullptr; }
    if (C.own_dst) { (void)hipFree(C.own_dst); C.own_dst = nullptr; }
    const uint64_t bound = nlzm_hip_compress_bound(n);
    HIPCHK(hipMalloc(&C.own_in, n + 512));
    HIPCHK(hipMalloc(&C.own_dst, bound));
    HIPCHK(hipEventRecord(C.ev[5], C.st));
    HIPCHK(hipMemsetAsync(C.own_in + n, 0, 512, C.st));
    if (n) HIPCHK(hipMemcpyAsync(C.own_in, src, n, hipMemcpyHostToDevice, C.st));
    HIPCHK(hipEventRecord(C.ev[6], C.st));
    HIPCHK(hipStreamSynchronize(C.st));
    float h2d = 0;
    HIPCHK(hipEventElapsedTime(&h2d, C.ev[5], C.ev[6]));
    int rc = stream_begin(C, C.own_in, n, hist_bits_req, C.own_dst, bound);
    if (rc) return rc;
    rc = stream_step(C, 0, nullptr, nullptr, nullptr);
    if (rc) return rc;
    uint64_t len = 0;
    rc = stream_finish(C, &len);
    if (rc) return rc;
    if (len > dst_cap) return set_err(NLZM_HIP_E_CAPACITY, "stream is %llu bytes, dst_cap %llu", (unsigned long long)len, (unsigned long long)dst_cap);
    HIPCHK(hipEventRecord(C.ev[5], C.st));
    HIPCHK(hipMemcpyAsync(dst, C.own_dst, len, hipMemcpyDeviceToHost, C.st));
    HIPCHK(hipEventRecord(C.ev[6], C.st));
    HIPCHK(hipStreamSynchronize(C.st));
    float d2h = 0;
    HIPCHK(hipEventElapsedTime(&d2h, C.ev[5], C.ev[6]));
    C.tm.h2d_ms = h2d; C.tm.d2h_ms = d2h;
    *dst_len = len;
    return 0;
}

int nlzm_hip_get_stats(nlzm_hip_stats *out)
{
    Ctx &C = g_ctx;
    if (!out) return set_err(NLZM_HIP_E_ARG, "null argument");
    if (C.open) { const int rc = refresh_stats(C); if (rc) return rc; }
    *out = C.stats;
    return 0;
}

int nlzm_hip_get_counter(const char *key, uint64_t *value)
{
    Ctx &C = g_ctx;
    if (!key || !value) return set_err(NLZM_HIP_E_ARG, "null argument");
    if (C.open) { const int rc = refresh_stats(C); if (rc) return rc; }
    static const struct { const char *name; int idx; } kProf[] = {
        { "finder_blocks", 0 }, { "table_blocks", 6 }, { "parser_blocks", 8 }, { "parser_passes", 13 },
        { "finder_wait_cycles", 16 }, { "finder_total_cycles", 17 }, { "table_wait_cycles", 18 }, { "table_total_cycles", 19 },
        { "parser_wait_cycles", 20 }, { "parser_total_cycles", 21 }, { "parser_emit_cycles", 22 }, { "parser_setup_cycles", 23 }, { "parser_pass_cycles", 24 },
        { "finder_bt_wait_cycles", 25 }, { "table_slow_blocks", 7 }, { "rk_cut_short_grown", 115 }, { "rk_cut_short_ties", 116 }, { "rk_cut_short_ties_won", 117 }, { "table_shape_changes", 113 }, { "table_wide_launches", 114 },
        { "finder_seg_own", 118 }, { "finder_seg_waited", 119 }, { "helper_jobs", 96 }, { "helper_taken", 97 }, { "helper_taken_nodes", 98 }, { "helper_wait_cycles", 99 }, { "helper_jobs_done", 101 }, { "helper_blocks", 102 }, { "helper_passes", 103 },
    };
    for (const auto &e : kProf) if (!strcmp(key, e.name)) { *value = C.prof_last[e.idx]; return 0; }
    if (!strcmp(key, "worker_call_cycles")) { *value = C.wc_last.call_cycles; return 0; }
    if (!strcmp(key, "worker_call_tests")) { *value = C.wc_last.call_tests; return 0; }
    if (!strcmp(key, "worker_calls")) { *value = C.wc_last.bt_calls; return 0; }
    if (!strcmp(key, "hot_bin_calls")) { *value = C.wc_last.hot_calls; return 0; }
    if (!strcmp(key, "positions")) { *value = C.stats.positions; return 0; }
    if (!strcmp(key, "block_pool_bytes")) { *value = cur().blocks_pool_size; return 0; }
    if (!strcmp(key, "block_redo_streams")) { *value = cur().redo_streams; return 0; }
    if (!strcmp(key, "gpu_max_hw_queues_effective")) { *value = (uint64_t)g_hwq_effective; return 0; }
    return set_err(NLZM_HIP_E_ARG, "unknown counter %s", key);
}

int nlzm_hip_get_timing(nlzm_hip_timing *out)
{
    Ctx &C = g_ctx;
    if (!out) return set_err(NLZM_HIP_E_ARG, "null argument");
    *out = C.tm;
    return 0;
}

void nlzm_hip_block_placement(uint32_t nstreams, uint32_t blocks_per_stream, uint32_t workgroup, uint32_t *stream, uint32_t *local)
{
    uint32_t s = 0, l = 0;
    multi_block_of(nstreams * blocks_per_stream, blocks_per_stream, workgroup, s, l);
    if (stream) *stream = s;
    if (local) *local = l;
}

int nlzm_hip_set_option(const char *key, int64_t value)
{
    Ctx &C = g_ctx;
    if (!key) return set_err(NLZM_HIP_E_ARG, "null key");
    if (!strcmp(key, "workers")) {      // BT4 always runs on the worker lanes (the three-stage pipeline has no other place for it)
        if (value != 1) return set_err(NLZM_HIP_E_ARG, "workers: only 1 is supported");
        return 0;
    }
    if (!strcmp(key, "worker_blocks")) { if (value < 1 || value > 255) return set_err(NLZM_HIP_E_ARG, "worker_blocks out of range"); C.opt_worker_blocks = value; return 0; }
    if (!strcmp(key, "hot_waves")) { if (value < 0 || value > 6) return set_err(NLZM_HIP_E_ARG, "hot_waves out of range"); C.opt_hot_waves = value; return 0; }
    if (!strcmp(key, "hot_min")) { if (value < 0 || value > (1 << 30)) return set_err(NLZM_HIP_E_ARG, "hot_min out of range"); C.opt_hot_min = value; return 0; }
    if (!strcmp(key, "worker_threads")) { if (value < 64 || value > 512 || value % 64) return set_err(NLZM_HIP_E_ARG, "worker_threads out of range"); C.opt_worker_threads = value; return 0; }
    if (!strcmp(key, "block_worker_threads")) { if (value < 64 || value > 512 || value % 64) return set_err(NLZM_HIP_E_ARG, "block_worker_threads out of range"); C.opt_block_threads = value; return 0; }
    if (!strcmp(key, "block_hot_waves")) { if (value < 0 || value > 6) return set_err(NLZM_HIP_E_ARG, "block_hot_waves out of range"); C.opt_block_hot_waves = value; return 0; }
    if (!strcmp(key, "prefilter_bits_per_position")) { if (value < 0 || value > 8) return set_err(NLZM_HIP_E_ARG, "prefilter_bits_per_position out of range"); C.opt_tbits_per = value; return 0; }
    if (!strcmp(key, "stage_report")) { C.opt_report = value != 0; return 0; }
    if (!strcmp(key, "parser_helper")) { C.opt_helper = value != 0; return 0; }
    if (!strcmp(key, "table_shape")) { if (value < 0 || value > 2) return set_err(NLZM_HIP_E_ARG, "table_shape out of range"); C.opt_table_shape = value; return 0; }
    if (!strcmp(key, "multi_allow_same_device")) { C.opt_multi_same = value != 0; return 0; }
    if (!strcmp(key, "test_fail_launch")) { C.opt_test_fail_launch = value; return 0; }
    if (!strcmp(key, "test_fail_stream")) { if (value < 0 || value > 63) return set_err(NLZM_HIP_E_ARG, "test_fail_stream out of range"); C.opt_test_fail_stream = value; return 0; }
    if (!strcmp(key, "block_ext_blocks")) { C.opt_block_ext_blocks = value; return 0; }
    if (!strcmp(key, "block_parser_helper")) { C.opt_block_helper = value != 0; return 0; }
    if (!strcmp(key, "keep_block_pool")) { C.opt_keep_pool = value != 0; if (!value && g_jobs.empty()) blocks_close(true); return 0; }
    if (!strcmp(key, "block_batch_chunks")) { if (value < 1 || value > 4096) return set_err(NLZM_HIP_E_ARG, "block_batch_chunks out of range"); C.opt_block_batch = value; return 0; }
    if (!strcmp(key, "batch_chunks")) { if (value < 1 || value > 4096) return set_err(NLZM_HIP_E_ARG, "batch_chunks out of range"); C.opt_batch = value; return 0; }
    return set_err(NLZM_HIP_E_ARG, "unknown option %s", key);
}

int nlzm_hip_rans_frames(const uint32_t *syms, const uint64_t *sym_off, const uint8_t *bits, const uint64_t *bits_off,
                         const uint32_t *num_ops, uint32_t nframes, uint8_t *out, uint64_t out_stride, uint32_t *out_len)
{
    Ctx &C = g_ctx;
    if (!C.inited) return set_err(NLZM_HIP_E_NODEVICE, "nlzm_hip_init() has not succeeded");
    if (!nframes) return 0;
    if (!syms || !sym_off || !bits || !bits_off || !num_ops || !out || !out_len) return set_err(NLZM_HIP_E_ARG, "null argument");
    uint64_t max_syms = 1, max_bits = 4;
    for (uint32_t f = 0; f < nframes; f++) {
        if (sym_off[f + 1] - sym_off[f] > max_syms) max_syms = sym_off[f + 1] - sym_off[f];
        if (bits_off[f + 1] - bits_off[f] > max_bits) max_bits = bits_off[f + 1] - bits_off[f];
    }
    uint32_t *d_syms = nullptr, *d_scr = nullptr; uint8_t *d_bits = nullptr, *d_out = nullptr; FrameMeta *d_fm = nullptr;
    struct Guard {      // the device buffers go with the call, on every path out
        uint32_t *&a, *&b; uint8_t *&c, *&d; FrameMeta *&e;
        ~Guard() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); if (c) (void)hipFree(c); if (d) (void)hipFree(d); if (e) (void)hipFree(e); }
    } guard{ d_syms, d_scr, d_bits, d_out, d_fm };
    const unsigned long long fstride = 12 + max_bits + 16 + 2 * max_syms;
    HIPCHK(hipMalloc(&d_syms, nframes * max_syms * 4));
    HIPCHK(hipMalloc(&d_scr, nframes * max_syms * 4));
    HIPCHK(hipMalloc(&d_bits, nframes * max_bits));
    HIPCHK(hipMalloc(&d_out, nframes * fstride));
    HIPCHK(hipMalloc(&d_fm, nframes * sizeof(FrameMeta)));
    std::vector<FrameMeta> hm(nframes);
    for (uint32_t f = 0; f < nframes; f++) {
        const uint64_t ns = sym_off[f + 1] - sym_off[f], nb = bits_off[f + 1] - bits_off[f];
        hm[f].nsyms = (uint32_t)ns; hm[f].nbits_bytes = (uint32_t)nb; hm[f].num_ops = num_ops[f]; hm[f].out_len = 0;
        if (ns) HIPCHK(hipMemcpyAsync(d_syms + f * max_syms, syms + sym_off[f], ns * 4, hipMemcpyHostToDevice, C.st));
        if (nb) HIPCHK(hipMemcpyAsync(d_bits + f * max_bits, bits + bits_off[f], nb, hipMemcpyHostToDevice, C.st));
    }
    HIPCHK(hipMemcpyAsync(d_fm, hm.data(), nframes * sizeof(FrameMeta), hipMemcpyHostToDevice, C.st));
    launch_rans(d_syms, max_syms, d_bits, max_bits, d_fm, d_scr, max_syms, d_out, fstride, (uint32_t)fstride, nframes, C.st);
    HIPCHK(hipMemcpyAsync(hm.data(), d_fm, nframes * sizeof(FrameMeta), hipMemcpyDeviceToHost, C.st));
    HIPCHK(hipStreamSynchronize(C.st));
    HIPCHK(hipGetLastError());
    int rc = 0;
    for (uint32_t f = 0; f < nframes && !rc; f++) {
        out_len[f] = hm[f].out_len;
        if (hm[f].out_len > out_stride) { rc = set_err(NLZM_HIP_E_CAPACITY, "frame %u needs %u bytes", f, hm[f].out_len); break; }
        HIPCHK(hipMemcpy(out + f * out_stride, d_out + f * fstride, hm[f].out_len, hipMemcpyDeviceToHost));
    }
    return rc;
}

int nlzm_hip_find_matches(const uint8_t *src, uint64_t n, uint32_t hist_bits_req, uint64_t pos_lo, uint64_t pos_hi,
                          uint32_t *out_words, uint64_t cap_words, uint64_t *used_words)
{
    Ctx &C = g_ctx;
    if (!C.inited) return set_err(NLZM_HIP_E_NODEVICE, "nlzm_hip_init() has not succeeded");
    if (!out_words || !used_words) return set_err(NLZM_HIP_E_ARG, "null argument");
    if (C.cap_words) { (void)hipFree(C.cap_words); C.cap_words = nullptr; }
    if (C.cap_used) { (void)hipFree(C.cap_used); C.cap_used = nullptr; }
    HIPCHK(hipMalloc(&C.cap_words, (cap_words + 1) * 4));
    HIPCHK(hipMalloc(&C.cap_used, 8));
    HIPCHK(hipMemset(C.cap_used, 0, 8));
    C.cap_cap = cap_words; C.cap_lo = pos_lo; C.cap_hi = pos_hi;
    const uint64_t bound = nlzm_hip_compress_bound(n);
    std::vector<uint8_t> tmp(bound);
    uint64_t len = 0;
    int rc = nlzm_hip_compress(src, n, hist_bits_req, tmp.data(), bound, &len);
    unsigned long long used = 0;
    if (!rc) {
        HIPCHK(hipMemcpy(&used, C.cap_used, 8, hipMemcpyDeviceToHost));
        // (the table stage's waves finish positions out of order: the records {position, max_len, delta[2..max_len]} are put
        //  into position order here)
        std::vector<uint32_t> raw(used);
        HIPCHK(hipMemcpy(raw.data(), C.cap_words, used * 4, hipMemcpyDeviceToHost));
        std::vector<std::pair<uint32_t, unsigned long long>> recs;      // position, offset
        for (unsigned long long at = 0; at + 2 <= used;) {
            recs.emplace_back(raw[at], at);
            at += 2 + (raw[at + 1] >= 2 ? raw[at + 1] - 1 : 0);
        }
        std::sort(recs.begin(), recs.end());
        unsigned long long o = 0;
        for (const auto &r : recs) {
            const unsigned long long len = 2 + (raw[r.second + 1] >= 2 ? raw[r.second + 1] - 1 : 0);
            memcpy(out_words + o, raw.data() + r.second, len * 4);
            o += len;
        }
        *used_words = used;
    }
    (void)hipFree(C.cap_words); (void)hipFree(C.cap_used);
    C.cap_words = nullptr; C.cap_used = nullptr; C.cap_cap = 0;
    return rc;
}

int nlzm_hip_parse_emit(const uint8_t *src, uint64_t n, uint32_t hist_bits_req, uint32_t frame_idx, uint32_t *syms,
                        uint32_t cap_syms, uint8_t *bits, uint32_t cap_bits, uint32_t *sizes_out)
{
    Ctx &C = g_ctx;
    if (!C.inited) return set_err(NLZM_HIP_E_NODEVICE, "nlzm_hip_init() has not succeeded");
    if (!syms || !bits || !sizes_out) return set_err(NLZM_HIP_E_ARG, "null argument");
    C.want_frame = frame_idx; C.got = false;
    const uint64_t bound = nlzm_hip_compress_bound(n);
    std::vector<uint8_t> tmp(bound);
    uint64_t len = 0;
    int rc = nlzm_hip_compress(src, n, hist_bits_req, tmp.data(), bound, &len);
    C.want_frame = -1;
    if (rc) return rc;
    if (!C.got) return set_err(NLZM_HIP_E_ARG, "frame %u does not exist", frame_idx);
    if (C.got_meta.nsyms > cap_syms || C.got_meta.nbits_bytes > cap_bits) return set_err(NLZM_HIP_E_CAPACITY, "capture buffers too small");
    memcpy(syms, C.got_syms.data(), C.got_meta.nsyms * 4ull);
    memcpy(bits, C.got_bits.data(), C.got_meta.nbits_bytes);
    sizes_out[0] = C.got_meta.nsyms; sizes_out[1] = C.got_meta.nbits_bytes; sizes_out[2] = C.got_meta.num_ops;
    return 0;
}

}  // extern "C"

// ---- independent blocks --------------------------------------------------------------
namespace {

int block_ctx_init(Ctx &c, int device, int64_t worker_blocks, int64_t batch)
{
    c.device = device;
    c.opt_report = 0;
    c.opt_workers = 1; c.opt_worker_blocks = worker_blocks; c.opt_batch = batch; c.opt_worker_threads = g_ctx.opt_block_threads;        // (block mode: a stream has few worker CUs, every lane of them takes bins)
    HIPCHK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
    for (auto &ev : c.ev) HIPCHK(hipEventCreate(&ev));
    for (auto &ev : c.ev_b) HIPCHK(hipEventCreate(&ev));
    c.double_sets = true;
    c.inited = true;
    return 0;
}
void block_ctx_destroy(Ctx &c)
{
    free_stream_buffers(c);
    for (auto &ev : c.ev) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    for (auto &ev : c.ev_b) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    if (c.st) { (void)hipStreamDestroy(c.st); c.st = nullptr; }
    c.inited = false;
}

}  // namespace

namespace {
// "keep_block_pool" keeps a closed set's one allocation (most of the device's memory for the bench's set) for the next set; anything
// else that then cannot allocate -- a single stream, a feed, find_matches -- takes it back here instead of failing with NOMEM.
bool idle_block_pool_dropped()
{
    if (!g_jobs.empty() || !cur().blocks_pool) return false;
    (void)hipFree(cur().blocks_pool);
    cur().blocks_pool = nullptr; cur().blocks_pool_size = 0;
    return true;
}

void blocks_close(bool drop_pool)
{
    // (a round may still be queued or on the device -- an abandoned set, a failed step: every device wait is bounded)
    for (auto &st : g_group_st) (void)hipStreamSynchronize(st);
    for (auto &j : g_jobs) if (j.c.st) (void)hipStreamSynchronize(j.c.st);
    (void)hipGetLastError();
    cur().rounds = DevState::Rounds{};
    for (auto &j : g_jobs) { j.d_out = nullptr; if (j.c.inited) block_ctx_destroy(j.c); }
    g_jobs.clear();
    if (cur().blocks_pool && (drop_pool || !g_ctx.opt_keep_pool)) { (void)hipFree(cur().blocks_pool); cur().blocks_pool = nullptr; cur().blocks_pool_size = 0; }
    for (auto &st : g_group_st) (void)hipStreamDestroy(st);
    for (auto &ev : g_group_ev) for (auto &e : ev) (void)hipEventDestroy(e);
    g_group_st.clear(); g_group_ev.clear();
    if (g_pack_host) (void)hipHostFree(g_pack_host);
    if (g_pack_dev) (void)hipFree(g_pack_dev);
    g_pack_host = g_pack_dev = nullptr;
}

// run f(block) for every open block, `conc` at a time, each on a host thread of its own
template <class F>
void for_blocks(uint32_t conc, F f)
{
    DevState &D = cur();
    const int device = D.ctx.device;
    std::mutex mu;
    uint32_t next_block = 0;
    auto worker = [&]() {
        t_dev = &D;
        (void)hipSetDevice(device);
        for (;;) {
            uint32_t i;
            { std::lock_guard<std::mutex> lk(mu); if (next_block >= g_jobs.size()) return; i = next_block++; }
            f(i, g_jobs[i]);
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < conc; t++) th.emplace_back(worker);
    for (auto &t : th) t.join();
}
}  // namespace

extern "C" {

int nlzm_hip_blocks_begin(const void *d_src, uint64_t n, uint32_t nblocks, uint32_t hist_bits_req)
{
    Ctx &C = g_ctx;
    const uint64_t per_fixed = cur().blocks_per;    // (a multi-device call fixes the partition; cleared here)
    cur().blocks_per = 0;
    if (!C.inited) return set_err(NLZM_HIP_E_NODEVICE, "nlzm_hip_init() has not succeeded");
    if (!nblocks || nblocks > 64) return set_err(NLZM_HIP_E_ARG, "nblocks out of range");
    blocks_close();
    // every block is in flight at once: one master CU + its worker CUs per stream, all resident together
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, C.device));
    // (a spare CU per stream while there is room for it; every workgroup of the launch has a CU of its own either way:
    //  at most CUs / 4 streams -- three stage CUs and one worker CU each -- which is 64 on an MI355X)
    // (the helper parser's workgroup leaves at once where the streams run without one: it takes no CU then)
    const int64_t roles_live = (int64_t)pipeline2_role_blocks() - (C.opt_block_helper ? 0 : (int64_t)v2::kHelpers);
    int64_t wb = prop.multiProcessorCount / (int64_t)nblocks - roles_live;
    if (wb > 1 && nblocks > 1) wb--;
    if (wb > C.opt_worker_blocks) wb = C.opt_worker_blocks;
    if (wb < 1) return set_err(NLZM_HIP_E_ARG, "%u streams do not fit %d CUs (at most %d)", nblocks, prop.multiProcessorCount,
                               prop.multiProcessorCount / (int)(roles_live + 1));
    g_blocks_wb = wb; g_blocks_n = n; g_blocks_src = (const uint8_t *)d_src; g_blocks_hist = hist_bits_req;
    // Every stream holds its own tables and hand-off arrays: the pre-filter table (4 << t_bits bytes) and the per-launch
    // arrays (about 2.2 KB per position of a launch) are sized so that all streams fit the free memory.
    int64_t tbits_max = 32, batch = C.opt_block_batch;
    {
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        free_b += cur().blocks_pool_size;               // (the allocation kept from the set before is this set's to use)
        const double per_stream = 0.85 * (double)free_b / nblocks;
        Geom g0;
        make_geom(per_fixed ? per_fixed : (n + nblocks - 1) / nblocks, hist_bits_req, g0);
        const double fixed = 8.0 * ((double)g0.wmask + 1) * 2 + 5e7;     // BT4 tree (widened), the rest
        double left = per_stream - fixed;
        if (left < 2e8) return set_err(NLZM_HIP_E_NOMEM, "%u streams of %llu bytes at -window:%u do not fit %.1f GB of free memory", nblocks,
                                       (unsigned long long)g0.n, g0.wbits, free_b / 1e9);
        const double slot_bytes = g0.n <= (unsigned long long)g0.wmask + 1 ? 0.125 : 4.0;        // (one bit per slot where the block is no longer than its window: stream_begin)
        while (tbits_max > 16 && slot_bytes * (double)(1ull << tbits_max) > 0.4 * left) tbits_max--;
        left -= slot_bytes * (double)(1ull << (tbits_max < (int64_t)g0.wbits + 5 ? tbits_max : (int64_t)g0.wbits + 5));
        const int64_t fit = (int64_t)(left / (2300.0 * g0.chunk_size + 8.0 * g0.chunk_size * 4));
        if (fit < 1) return set_err(NLZM_HIP_E_NOMEM, "%u streams do not fit the device memory", nblocks);
        if (batch > fit) batch = fit;
    }
    const uint64_t per = per_fixed ? per_fixed : (n + nblocks - 1) / nblocks;     // block i = [i*per, min(n, (i+1)*per))
    g_jobs.resize(nblocks);
    for (uint32_t i = 0; i < nblocks; i++) {
        g_jobs[i].lo = (uint64_t)i * per < n ? (uint64_t)i * per : n;
        const uint64_t hi = (uint64_t)(i + 1) * per < n ? (uint64_t)(i + 1) * per : n;
        g_jobs[i].n = hi - g_jobs[i].lo;
        g_jobs[i].bound = nlzm_hip_compress_bound(g_jobs[i].n);
    }
    const int device = C.device;
    {   // ONE allocation for the whole block set: what a stream takes is added up first (the same code path, nothing touched on
        // the device), then every stream gets its slice -- some thirty-five hipMalloc calls per stream otherwise
        std::vector<size_t> need(nblocks);
        for (uint32_t i = 0; i < nblocks; i++) {
            Ctx m;                                  // (a scratch context: options as the streams will have them)
            m.inited = true; m.device = device; m.st = C.st;
            m.opt_workers = 1; m.opt_worker_blocks = wb; m.opt_batch = batch; m.opt_worker_threads = C.opt_block_threads; m.opt_tbits_max = tbits_max; m.cu_count = C.cu_count;
            m.opt_hot_waves = C.opt_block_hot_waves; m.opt_hot_min = C.opt_hot_min; m.opt_tbits_per = C.opt_tbits_per; m.opt_helper = C.opt_block_helper;
            m.opt_block_ext_blocks = C.opt_block_ext_blocks;
            Pool mp; mp.measuring = true;
            m.pool = &mp; m.double_sets = true;
            uint8_t *fake_out = nullptr;
            const int rc = stream_begin(m, g_blocks_src + g_jobs[i].lo, g_jobs[i].n, hist_bits_req, (void *)(uintptr_t)16, g_jobs[i].bound);
            (void)fake_out;
            m.st = nullptr;
            if (rc) { blocks_close(); return rc; }
            need[i] = ((mp.used + 255) & ~(size_t)255) + ((g_jobs[i].bound + 255) & ~(size_t)255) + 4096;
        }
        size_t total = 0;
        for (size_t v : need) total += v;
        if (cur().blocks_pool && cur().blocks_pool_size < total) { (void)hipFree(cur().blocks_pool); cur().blocks_pool = nullptr; cur().blocks_pool_size = 0; }
        if (!cur().blocks_pool) {
            if (hipMalloc(&cur().blocks_pool, total) != hipSuccess) { cur().blocks_pool = nullptr; blocks_close(); return set_err(NLZM_HIP_E_NOMEM, "block set: %zu bytes for %u streams", total, nblocks); }
            cur().blocks_pool_size = total;
        }
        size_t at = 0;
        for (uint32_t i = 0; i < nblocks; i++) {
            g_jobs[i].pool.base = cur().blocks_pool + at; g_jobs[i].pool.size = need[i]; g_jobs[i].pool.used = 0; g_jobs[i].pool.measuring = false;
            at += need[i];
        }
    }
    for_blocks(nblocks, [&](uint32_t i, BlockJob &j) {
        j.rc = block_ctx_init(j.c, device, wb, batch);
        j.c.opt_tbits_max = tbits_max; j.c.cu_count = C.cu_count;
        j.c.opt_hot_waves = C.opt_block_hot_waves; j.c.opt_hot_min = C.opt_hot_min; j.c.opt_tbits_per = C.opt_tbits_per; j.c.opt_helper = C.opt_block_helper;
        j.c.opt_table_shape = C.opt_table_shape; j.c.opt_block_ext_blocks = C.opt_block_ext_blocks;
        j.c.opt_test_fail_launch = (int64_t)i == C.opt_test_fail_stream ? C.opt_test_fail_launch : -1;
        j.c.pool = &j.pool;
        if (!j.rc) j.rc = dev_alloc(j.c, &j.d_out, j.bound);
        if (!j.rc) j.rc = stream_begin(j.c, g_blocks_src + j.lo, j.n, hist_bits_req, j.d_out, j.bound);
        (void)i;
    });
    for (auto &j : g_jobs) if (j.rc) { const int rc = j.rc; blocks_close(); return rc; }
    const int rc = [&]() -> int {
        HIPCHK(hipHostMalloc(&g_pack_host, 2 * stream2_pack_size(), hipHostMallocDefault));      // (one per launch set)
        HIPCHK(hipMalloc(&g_pack_dev, 2 * stream2_pack_size()));
        {   // the HIP stream of the shared launches: of higher priority than the streams' own, i.e. on a hardware queue apart
            int lo_p = 0, hi_p = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
            hipStream_t st;
            HIPCHK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi_p));
            g_group_st.push_back(st);
        }
        for (uint32_t qi = 0; qi < 2; qi++) {
            std::array<hipEvent_t, 3> ev;
            for (auto &e : ev) HIPCHK(hipEventCreate(&e));
            g_group_ev.push_back(ev);
        }
        return 0;
    }();
    if (rc) blocks_close();         // (nothing of a block set that failed to open stays allocated)
    return rc;
}

static int blocks_step_impl(uint32_t max_chunks_per_block, uint64_t *in_done_total, int *finished, double *device_ms);
int nlzm_hip_blocks_step(uint32_t max_chunks_per_block, uint64_t *in_done_total, int *finished, double *device_ms)
{
    if (g_jobs.empty()) return set_err(NLZM_HIP_E_ARG, "no open block set");
    const int rc = blocks_step_impl(max_chunks_per_block, in_done_total, finished, device_ms);
    if (rc) {
        // A failed round ends the block set: wait for whatever is still queued (every device wait is bounded), then free
        // every stream's buffers -- the caller's source buffer is not read after this returns.
        char keep[sizeof g_err];
        { std::lock_guard<std::mutex> lk(g_err_mu); memcpy(keep, g_err, sizeof keep); }
        for (auto &st : g_group_st) (void)hipStreamSynchronize(st);
        for (auto &j : g_jobs) if (j.c.st) (void)hipStreamSynchronize(j.c.st);
        (void)hipGetLastError();
        blocks_close();
        { std::lock_guard<std::mutex> lk(g_err_mu); memcpy(g_err, keep, sizeof keep); }
    }
    return rc;
}
static int blocks_step_impl(uint32_t max_chunks_per_block, uint64_t *in_done_total, int *finished, double *device_ms)
{
    Ctx &C = g_ctx;
    const size_t nj = g_jobs.size();
    hipEvent_t e0 = C.ev[5], e1 = C.ev[6];
    HIPCHK(hipEventRecord(e0, C.st));
    HIPCHK(hipStreamSynchronize(C.st));
    // Rounds: every unfinished stream advances by one launch's worth, and the streams of a round share ONE persistent launch.
    // The rounds overlap: while launch r is on the device (on the 224 CUs its workgroups hold), the pre-pass kernels of round
    // r + 1 run on the CUs that are left, launch r + 1 is queued behind launch r, and the host waits for launch r only to code
    // and gather its frames -- which then runs beside launch r + 1.  (Measured before, 32 streams: 115-140 ms of pre-pass
    // and frame coding between two launches of 580 ms.)  A round's launch is bracketed by round_open_kernel (progress words)
    // and round_close_kernel (what the host checks, copied aside): both on the launch's HIP stream, which has a hardware queue
    // of its own (a stream of higher priority), so that nothing of the other streams queues behind a persistent launch.
    // A call that has collected its share leaves the next round QUEUED (sized like its own rounds) for the next call to collect:
    // the device does not idle between the calls of a caller that steps through the set.
    if (nj > stream2_pack_capacity()) return set_err(NLZM_HIP_E_ARG, "too many streams for one launch");
    DevState::Rounds &R = cur().rounds;
    hipStream_t gs = g_group_st[0];
    for (auto &p : R.plan) if (p.size() != nj) p.assign(nj, StepPlan{});
    const uint32_t kAll = 0xFFFFFFFFu;
    std::vector<uint32_t> quota(nj, max_chunks_per_block ? max_chunks_per_block : kAll);       // chunks this call still collects, per stream
    auto queue_round = [&](uint32_t q, bool in_call) -> int {      // pre-passes of the round's streams, then its launch
        std::vector<uint32_t> &act = R.act[q];
        act.clear();
        std::vector<uint32_t> todo(nj, 0);
        for (size_t i = 0; i < nj; i++) {
            const Ctx &c = g_jobs[i].c;
            const uint32_t rem = c.g.nchunks - c.pre_chunk;
            const uint32_t want = in_call ? quota[i] : (max_chunks_per_block ? max_chunks_per_block : kAll);
            todo[i] = rem < want ? rem : want;
            if (todo[i]) act.push_back((uint32_t)i);
        }
        if (act.empty()) return 0;
        for (uint32_t i : act) { const int rc = step_pre(g_jobs[i].c, todo[i], R.plan[q][i], true); if (rc) return rc; }
        uint8_t *ph = (uint8_t *)g_pack_host + (size_t)q * stream2_pack_size(), *pd = (uint8_t *)g_pack_dev + (size_t)q * stream2_pack_size();
        for (uint32_t k = 0; k < act.size(); k++) {
            Ctx &c = g_jobs[act[k]].c;
            const StepPlan &P = R.plan[q][act[k]];
            HIPCHK(hipStreamWaitEvent(gs, P.ev[6], 0));                 // its pre-pass is done
            fill_stream2_args(ph, k, c.g, P.G, P.V, P.c0, P.c1, P.snap);
        }
        HIPCHK(hipMemcpyAsync(pd, ph, stream2_pack_size(), hipMemcpyHostToDevice, gs));
        launch_round_open(pd, (uint32_t)act.size(), gs);
        HIPCHK(hipEventRecord(g_group_ev[q][0], gs));
        launch_pipeline2_multi(pd, (uint32_t)act.size(), (uint32_t)g_blocks_wb, gs);
        HIPCHK(hipEventRecord(g_group_ev[q][1], gs));
        launch_round_close(pd, (uint32_t)act.size(), gs);
        HIPCHK(hipEventRecord(g_group_ev[q][2], gs));
        return 0;
    };
    if (!R.have) {
        const int rc = queue_round(R.q, true);
        if (rc) return rc;
        R.have = !R.act[R.q].empty();
    }
    while (R.have) {
        const uint32_t q = R.q;
        bool more = false;              // does this call collect another round after this one?
        for (uint32_t i : R.act[q]) { const uint32_t nb = R.plan[q][i].nb; if (quota[i] != kAll) quota[i] -= nb < quota[i] ? nb : quota[i]; }
        for (size_t i = 0; i < nj; i++) more |= quota[i] && g_jobs[i].c.pre_chunk < g_jobs[i].c.g.nchunks;
        { const int rc = queue_round(q ^ 1, more); if (rc) return rc; }
        {   // (every stream of the round is looked at, so that the first failure is reported with its own diagnostics)
            int first_rc = 0;
            char first_msg[sizeof g_err] = "";
            std::vector<int> rcs(nj, 0);
            auto note = [&](uint32_t i, int rc) {
                if (rc && !rcs[i]) rcs[i] = rc;
                if (rc && !first_rc) {
                    first_rc = rc;
                    std::lock_guard<std::mutex> lk(g_err_mu);
                    snprintf(first_msg, sizeof first_msg, "block %u: %.*s", i, (int)sizeof first_msg - 32, g_err);
                }
            };
            // (a stream marked `redo` -- a launch of it used its pair-list arena up -- is out of the set's rounds: what is still queued of it runs on
            //  a state that is valid but not the reference's, and nothing of it is looked at; nlzm_hip_blocks_finish makes the stream again)
            for (uint32_t i : R.act[q]) {
                if (g_jobs[i].redo) continue;
                const hipError_t e = hipStreamWaitEvent(g_jobs[i].c.st, g_group_ev[q][2], 0);
                note(i, e == hipSuccess ? step_post_issue(g_jobs[i].c, R.plan[q][i]) : set_err(NLZM_HIP_E_NODEVICE, "hipStreamWaitEvent failed: %s", hipGetErrorString(e)));
            }
            for (uint32_t i : R.act[q]) if (!rcs[i] && !g_jobs[i].redo) {
                Ctx &c = g_jobs[i].c;
                const int rc = step_post_check(c, R.plan[q][i]);
                if (rc && c.arena_out) { g_jobs[i].redo = true; c.next_chunk = c.pre_chunk = c.g.nchunks; continue; }
                note(i, rc);
            }
            for (uint32_t i : R.act[q]) if (!rcs[i] && !g_jobs[i].redo) note(i, step_post_done(g_jobs[i].c, R.plan[q][i], 0.0f));
            if (first_rc) { std::lock_guard<std::mutex> lk(g_err_mu); memcpy(g_err, first_msg, sizeof g_err); return first_rc; }
        }
        {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, g_group_ev[q][0], g_group_ev[q][1]));
            for (uint32_t i : R.act[q]) { Ctx &c = g_jobs[i].c; c.tm.match_parse_ms += ms; c.tm.total_ms += ms; }
        }
        R.q = q ^ 1;
        R.have = !R.act[q ^ 1].empty();
        if (!more) break;               // (what is queued now is the next call's first round)
    }
    HIPCHK(hipEventRecord(e1, C.st));
    HIPCHK(hipStreamSynchronize(C.st));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    if (device_ms) *device_ms = ms;
    uint64_t tot = 0; int all = 1;
    for (size_t i = 0; i < nj; i++) {
        const Ctx &c = g_jobs[i].c;
        const unsigned long long d = (unsigned long long)c.next_chunk * c.g.chunk_size;
        tot += d < c.g.n ? d : c.g.n;
        all &= c.next_chunk >= c.g.nchunks;
    }
    if (in_done_total) *in_done_total = tot;
    if (finished) *finished = all;
    return 0;
}

int nlzm_hip_blocks_finish(void *d_dst, uint64_t dst_cap, uint64_t *block_len, uint64_t *dst_len)
{
    Ctx &C = g_ctx;
    if (g_jobs.empty()) return set_err(NLZM_HIP_E_ARG, "no open block set");
    if (!d_dst || !dst_len) return set_err(NLZM_HIP_E_ARG, "null argument");
    for_blocks((uint32_t)g_jobs.size(), [&](uint32_t, BlockJob &j) { j.rc = stream_finish(j.c, &j.len); });
    // A stream whose launch ran out of extension blocks for its BT4 pair lists (a block set reserves 32 pairs per position and an arena for the
    // positions that have more: an input with such positions all over it compresses as a single stream, which reserves all 256, but not here)
    // is made again now, from its first byte, as a single stream with buffers of its own, into its place in the set: the bytes are the same
    // either way (the reference run on the block), only the time differs.
    cur().redo_streams = 0;
    for (auto &j : g_jobs) {
        if (!j.redo) continue;
        cur().redo_streams++;
        Ctx &c = j.c;
        (void)hipStreamSynchronize(c.st);
        for (auto &st : g_group_st) (void)hipStreamSynchronize(st);
        c.pool = nullptr; c.double_sets = false;
        c.opt_worker_blocks = C.opt_worker_blocks; c.opt_worker_threads = C.opt_worker_threads; c.opt_hot_waves = C.opt_hot_waves; c.opt_batch = C.opt_batch;
        c.opt_helper = C.opt_helper; c.opt_tbits_max = C.opt_tbits_max; c.opt_test_fail_launch = -1;
        j.rc = stream_begin(c, g_blocks_src + j.lo, j.n, g_blocks_hist, j.d_out, j.bound);
        if (!j.rc) j.rc = stream_step(c, 0, nullptr, nullptr, nullptr);
        if (!j.rc) j.rc = stream_finish(c, &j.len);
        if (C.opt_report) fprintf(stderr, "block set: the stream of block %zu was made again as a single stream (its pair-list arena of %u blocks per launch had run out)%s\n",
                                  (size_t)(&j - &g_jobs[0]), c.ext_cap, j.rc ? ": FAILED" : "");
    }
    if (C.opt_report) {
        // which stage limits a stream under load: smallest / median / largest over the streams, cycles per position
        static const char *const what[8] = { "finder total", "finder waiting", "  of it for BT4 results", "table stage total", "table stage waiting",
                                             "parser total", "parser waiting (records)", "parser passes" };
        fprintf(stderr, "block set of %zu streams, %lld worker CUs each -- per stream, cycles per position (min / median / max over the streams):\n", g_jobs.size(), (long long)g_blocks_wb);
        for (int k = 0; k < 8; k++) {
            std::vector<double> v;
            for (auto &j : g_jobs) if (!j.rc) v.push_back(j.c.acct[k]);
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            fprintf(stderr, "  %-26s %8.0f %8.0f %8.0f\n", what[k], v.front(), v[v.size() / 2], v.back());
        }
    }
    int rc = 0;
    uint64_t pos = 0;
    memset(&C.stats, 0, sizeof C.stats);
    for (size_t i = 0; i < g_jobs.size() && !rc; i++) {
        BlockJob &j = g_jobs[i];
        if (j.rc) { rc = j.rc; break; }
        if (pos + j.len > dst_cap) { rc = set_err(NLZM_HIP_E_CAPACITY, "dst_cap %llu too small", (unsigned long long)dst_cap); break; }
        if (hipMemcpyAsync((uint8_t *)d_dst + pos, j.d_out, j.len, hipMemcpyDeviceToDevice, C.st) != hipSuccess)
            rc = set_err(NLZM_HIP_E_NODEVICE, "gathering block %zu failed", i);
        if (block_len) block_len[i] = j.len;
        pos += j.len;
        uint64_t *dst = (uint64_t *)&C.stats; const uint64_t *src = (const uint64_t *)&j.c.stats;   // counters of the whole job
        for (size_t k = 0; k < sizeof(C.stats) / 8; k++) dst[k] += src[k];
    }
    (void)hipStreamSynchronize(C.st);
    blocks_close();
    if (!rc) *dst_len = pos;
    return rc;
}

void nlzm_hip_blocks_abandon(void) { blocks_close(); }

int nlzm_hip_compress_blocks_dev(const void *d_src, uint64_t n, uint32_t nblocks, uint32_t hist_bits_req, void *d_dst,
                                 uint64_t dst_cap, uint64_t *block_len, uint64_t *dst_len)
{
    const auto t0 = std::chrono::steady_clock::now();
    int rc = nlzm_hip_blocks_begin(d_src, n, nblocks, hist_bits_req);
    if (rc) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    double dev_ms = 0;
    rc = nlzm_hip_blocks_step(0, nullptr, nullptr, &dev_ms);
    if (rc) return rc;              // (the failed step has closed the set)
    const auto t2 = std::chrono::steady_clock::now();
    rc = nlzm_hip_blocks_finish(d_dst, dst_cap, block_len, dst_len);
    if (g_ctx.opt_report) {
        const auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "block set of %u: begin (tables, pre-filter) %.0f ms, steps %.0f ms (device %.0f ms), finish (gather) %.0f ms\n", nblocks,
                ms(t0, t1), ms(t1, t2), dev_ms, ms(t2, std::chrono::steady_clock::now()));
    }
    return rc;
}

int nlzm_hip_compress_blocks(const uint8_t *src, uint64_t n, uint32_t nblocks, uint32_t hist_bits_req, uint8_t *dst,
                             uint64_t dst_cap, uint64_t *block_len, uint64_t *dst_len)
{
    Ctx &C = g_ctx;
    if (!C.inited) return set_err(NLZM_HIP_E_NODEVICE, "nlzm_hip_init() has not succeeded");
    if ((!src && n) || !dst || !dst_len || !nblocks) return set_err(NLZM_HIP_E_ARG, "null argument");
    uint8_t *d_in = nullptr, *d_out = nullptr;
    const uint64_t bound = nlzm_hip_compress_bound(n) + (uint64_t)nblocks * (16 + 131072);
    HIPCHK(hipMalloc(&d_in, n + 512));
    if (hipMalloc(&d_out, bound) != hipSuccess) { (void)hipFree(d_in); return set_err(NLZM_HIP_E_NOMEM, "output buffer"); }
    if (hipMemset(d_in + n, 0, 512) != hipSuccess || (n && hipMemcpy(d_in, src, n, hipMemcpyHostToDevice) != hipSuccess)) {
        (void)hipFree(d_in); (void)hipFree(d_out);
        return set_err(NLZM_HIP_E_NODEVICE, "copying the input to the device failed");
    }
    uint64_t len = 0;
    int rc = nlzm_hip_compress_blocks_dev(d_in, n, nblocks, hist_bits_req, d_out, bound, block_len, &len);
    if (!rc && len > dst_cap) rc = set_err(NLZM_HIP_E_CAPACITY, "streams are %llu bytes, dst_cap %llu", (unsigned long long)len, (unsigned long long)dst_cap);
    if (!rc && hipMemcpy(dst, d_out, len, hipMemcpyDeviceToHost) != hipSuccess) rc = set_err(NLZM_HIP_E_NODEVICE, "copy back failed");
    (void)hipFree(d_in); (void)hipFree(d_out);
    if (!rc) *dst_len = len;
    return rc;
}

// ---- streaming host input (the reference reads and writes as it goes: NLZM.cpp:1774-1778, :1853, :1870-1885) ----
// The caller hands the input over in pieces, in order, and takes the stream back in pieces.  A piece goes through one of
// two pinned staging buffers onto a copy stream; while it travels, the chunks whose input has arrived are compressed, so
// host reads, uploads and kernels overlap and the host never holds more than a piece (the input stays whole in HBM:
// matches reach back a window).
static void feed_close(DevState &D)
{
    DevState::Feed &F = D.feed;
    for (int k = 0; k < 2; k++) { if (F.pin[k]) (void)hipHostFree(F.pin[k]); if (F.ev[k]) (void)hipEventDestroy(F.ev[k]); F.pin[k] = nullptr; F.ev[k] = nullptr; }
    if (F.st) (void)hipStreamDestroy(F.st);
    F = DevState::Feed{};
}
// chunks whose input (with the lookahead the launch's kernels read) lies below `arrived`
static uint32_t feed_chunks_ready(const Geom &g, uint64_t arrived)
{
    if (arrived >= g.n) return g.nchunks;
    const uint64_t slack = (uint64_t)g.feed - g.chunk_size + 1024;      // lookahead of the last chunk + RK256 / pre-filter windows
    // (what a launch READS AND USES lies below its last position + slack; its RK256 pre-pass also hashes a little further, into bytes
    //  that may still be arriving -- those hashes are recomputed by the next launch before anything looks at them: step_pre)
    if (arrived < slack + g.chunk_size) return 0;
    return (uint32_t)((arrived - slack) / g.chunk_size);
}
static int feed_run(DevState &D)
{
    Ctx &C = D.ctx;
    const uint32_t ready = feed_chunks_ready(C.g, D.feed.arrived);
    if (ready > C.next_chunk) return stream_step(C, ready - C.next_chunk, nullptr, nullptr, nullptr);
    return 0;
}

int nlzm_hip_feed_begin(uint64_t n, uint32_t hist_bits_req)
{
    DevState &D = cur();
    Ctx &C = D.ctx;
    if (!C.inited) return set_err(NLZM_HIP_E_NODEVICE, "nlzm_hip_init() has not succeeded");
    if (n >= 0xFFFF0000ull) return set_err(NLZM_HIP_E_TOOBIG, "input too large");
    feed_close(D);
    if (C.own_in) { (void)hipFree(C.own_in); C.own_in = nullptr; }
    if (C.own_dst) { (void)hipFree(C.own_dst); C.own_dst = nullptr; }
    const uint64_t bound = nlzm_hip_compress_bound(n);
    HIPCHK(hipMalloc(&C.own_in, n + 512));
    HIPCHK(hipMalloc(&C.own_dst, bound));
    HIPCHK(hipMemsetAsync(C.own_in + n, 0, 512, C.st));
    int rc = stream_begin(C, C.own_in, n, hist_bits_req, C.own_dst, bound);
    if (rc) return rc;
    DevState::Feed &F = D.feed;
    rc = [&]() -> int {
        HIPCHK(hipStreamCreateWithFlags(&F.st, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) { HIPCHK(hipHostMalloc((void **)&F.pin[k], kFeedPiece, hipHostMallocDefault)); HIPCHK(hipEventCreate(&F.ev[k])); }
        return 0;
    }();
    if (rc) { feed_close(D); return rc; }
    F.open = true; F.n = n;
    return 0;
}

int nlzm_hip_feed(const uint8_t *piece, uint64_t len)
{
    DevState &D = cur();
    DevState::Feed &F = D.feed;
    if (!F.open) return set_err(NLZM_HIP_E_ARG, "no open feed");
    if ((!piece && len) || F.fed + len > F.n) return set_err(NLZM_HIP_E_ARG, "feed of %llu bytes at %llu exceeds the %llu announced", (unsigned long long)len,
                                                             (unsigned long long)F.fed, (unsigned long long)F.n);
    while (len) {
        const uint32_t k = F.next;
        const uint64_t m = len < kFeedPiece ? len : kFeedPiece;
        // the staging buffer is free once its last upload has landed; what landed is input the kernels may read
        HIPCHK(hipEventSynchronize(F.ev[k]));
        if (F.end_of[k] > F.arrived) F.arrived = F.end_of[k];
        memcpy(F.pin[k], piece, m);
        HIPCHK(hipMemcpyAsync(D.ctx.own_in + F.fed, F.pin[k], m, hipMemcpyHostToDevice, F.st));
        HIPCHK(hipEventRecord(F.ev[k], F.st));
        F.fed += m; F.end_of[k] = F.fed; F.next = k ^ 1u;
        piece += m; len -= m;
        // (this piece is on its way: meanwhile, the chunks whose input is there)
        const int rc = feed_run(D);
        if (rc) { feed_close(D); return rc; }
    }
    return 0;
}

// the bytes of the stream produced since the last call (whole frames); *len = 0: nothing new
int nlzm_hip_feed_output(uint8_t *dst, uint64_t cap, uint64_t *len)
{
    DevState &D = cur();
    DevState::Feed &F = D.feed;
    if (!F.open) return set_err(NLZM_HIP_E_ARG, "no open feed");
    if (!dst || !len) return set_err(NLZM_HIP_E_ARG, "null argument");
    const uint64_t have = D.ctx.out_pos - F.taken, m = have < cap ? have : cap;
    if (m) HIPCHK(hipMemcpy(dst, D.ctx.own_dst + F.taken, m, hipMemcpyDeviceToHost));
    F.taken += m;
    *len = m;
    return 0;
}

// after the last piece: the remaining chunks and the terminator; then nlzm_hip_feed_output until it returns 0 bytes, then
// nlzm_hip_feed_end
int nlzm_hip_feed_finish(void)
{
    DevState &D = cur();
    DevState::Feed &F = D.feed;
    if (!F.open) return set_err(NLZM_HIP_E_ARG, "no open feed");
    if (F.fed != F.n) { const unsigned long long a = F.fed, b = F.n; feed_close(D); return set_err(NLZM_HIP_E_ARG, "%llu of %llu input bytes were fed", a, b); }
    HIPCHK(hipStreamSynchronize(F.st));
    F.arrived = F.n;
    int rc = feed_run(D);
    if (!rc) { uint64_t total = 0; rc = stream_finish(D.ctx, &total); }
    if (rc) feed_close(D);
    return rc;
}
void nlzm_hip_feed_end(void) { feed_close(cur()); }

// ---- independent blocks on several GPUs of one node -------------------------------------------------
// One host thread and one device state per GPU; device i compresses blocks [i*m, (i+1)*m) of the n-byte input's partition into
// ndev*m blocks (the same byte ranges nlzm_hip_compress_blocks uses for that many blocks) in block mode; there is no traffic
// between the GPUs while they compress.  The only exchange is the final gather of the streams onto the first device of the
// list, GPU to GPU (hipMemcpyPeerAsync: over xGMI where the devices are linked), from where the artifact goes to the host.
int nlzm_hip_compress_blocks_multi(const int *devices, uint32_t ndev, uint32_t blocks_per_dev, const uint8_t *src, uint64_t n,
                                   uint32_t hist_bits_req, uint8_t *dst, uint64_t dst_cap, uint64_t *block_len, uint64_t *dst_len)
{
    if (!devices || !ndev || !blocks_per_dev || (!src && n) || !dst || !dst_len) return set_err(NLZM_HIP_E_ARG, "null argument");
    if (ndev > 64) return set_err(NLZM_HIP_E_ARG, "more than 64 devices");
    if (!g_dev0.ctx.opt_multi_same)
        for (uint32_t i = 0; i < ndev; i++)
            for (uint32_t k = 0; k < i; k++)
                if (devices[i] == devices[k]) return set_err(NLZM_HIP_E_ARG, "device %d is listed twice", devices[i]);
    const uint64_t nb_total = (uint64_t)ndev * blocks_per_dev;
    const uint64_t per = n ? (n + nb_total - 1) / nb_total : 1;
    struct Part {
        DevState D;
        int device = 0, rc = 0;
        uint64_t lo = 0, n = 0, bound = 0, len = 0;
        uint8_t *d_in = nullptr, *d_out = nullptr;
        std::vector<uint64_t> blens;
        char msg[sizeof g_err] = "";
        bool pinned = false, direct = false;
        double h2d_ms = 0, run_ms = 0, gather_ms = 0;
    };
    std::vector<Part> parts(ndev);
    int dev_before = -1;
    (void)hipGetDevice(&dev_before);                // (the caller's current device is put back on the way out)
    // the caller's pages pinned for the uploads when the driver allows it (a pageable copy goes through a bounce buffer): the whole
    // range once, page-aligned, before the threads start -- their parts share pages
    bool pinned_all = false;
    uint8_t *pin_lo = nullptr; size_t pin_len = 0;
    if (n) {
        const uintptr_t pg = 4096, lo = (uintptr_t)src & ~(pg - 1), hi = ((uintptr_t)src + n + pg - 1) & ~(pg - 1);
        pin_lo = (uint8_t *)lo; pin_len = (size_t)(hi - lo);
        pinned_all = hipHostRegister(pin_lo, pin_len, hipHostRegisterPortable) == hipSuccess;
        if (!pinned_all) (void)hipGetLastError();
    }
    auto run_part = [&](uint32_t i) {
        Part &P = parts[i];
        t_dev = &P.D;
        P.device = devices[i];
        P.lo = (uint64_t)i * blocks_per_dev * per < n ? (uint64_t)i * blocks_per_dev * per : n;
        const uint64_t hi = (uint64_t)(i + 1) * blocks_per_dev * per < n ? (uint64_t)(i + 1) * blocks_per_dev * per : n;
        P.n = hi - P.lo;
        P.bound = nlzm_hip_compress_bound(P.n) + (uint64_t)blocks_per_dev * (16 + 131072);
        P.blens.assign(blocks_per_dev, 0);
        P.rc = [&]() -> int {
            int rc = dev_init(P.D, P.device);
            if (rc) return rc;
            {   // the options set through nlzm_hip_set_option hold for every device of the call
                const Ctx &o = g_dev0.ctx; Ctx &c = P.D.ctx;
                c.opt_batch = o.opt_batch; c.opt_worker_blocks = o.opt_worker_blocks; c.opt_worker_threads = o.opt_worker_threads;
                c.opt_hot_waves = o.opt_hot_waves; c.opt_hot_min = o.opt_hot_min; c.opt_report = o.opt_report;
                c.opt_block_threads = o.opt_block_threads; c.opt_block_hot_waves = o.opt_block_hot_waves; c.opt_block_batch = o.opt_block_batch;
                c.opt_tbits_per = o.opt_tbits_per; c.opt_helper = o.opt_helper; c.opt_block_helper = o.opt_block_helper; c.opt_table_shape = o.opt_table_shape;
            }
            HIPCHK(hipMalloc(&P.d_in, P.n + 512));
            HIPCHK(hipMalloc(&P.d_out, P.bound));
            HIPCHK(hipMemset(P.d_in + P.n, 0, 512));
            const auto t0 = std::chrono::steady_clock::now();
            P.pinned = pinned_all;
            if (P.n) HIPCHK(hipMemcpy(P.d_in, src + P.lo, P.n, hipMemcpyHostToDevice));
            const auto t1 = std::chrono::steady_clock::now();
            P.D.blocks_per = per;
            rc = nlzm_hip_compress_blocks_dev(P.d_in, P.n, blocks_per_dev, hist_bits_req, P.d_out, P.bound, P.blens.data(), &P.len);
            P.h2d_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
            P.run_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
            return rc;
        }();
        if (P.rc) snprintf(P.msg, sizeof P.msg, "device %d: %.*s", P.device, (int)sizeof P.msg - 32, P.D.err);      // (this thread's own text)
        t_dev = nullptr;
    };
    {
        std::vector<std::thread> th;
        for (uint32_t i = 0; i < ndev; i++) th.emplace_back(run_part, i);
        for (auto &t : th) t.join();
    }
    if (pinned_all) (void)hipHostUnregister(pin_lo);
    int rc = 0;
    uint64_t total = 0;
    for (auto &P : parts) { if (P.rc && !rc) { rc = P.rc; std::lock_guard<std::mutex> lk(g_err_mu); memcpy(g_err, P.msg, sizeof g_err); } total += P.len; }
    if (!rc && total > dst_cap) rc = set_err(NLZM_HIP_E_CAPACITY, "streams are %llu bytes, dst_cap %llu", (unsigned long long)total, (unsigned long long)dst_cap);
    if (!rc) {
        // the gather: every device's streams onto the first one, in block order, then one copy to the host
        const int root = parts[0].device;
        uint8_t *d_all = nullptr;
        rc = [&]() -> int {
            HIPCHK(hipSetDevice(root));
            if (ndev == 1) { HIPCHK(hipMemcpy(dst, parts[0].d_out, total, hipMemcpyDeviceToHost)); return 0; }
            HIPCHK(hipMalloc(&d_all, total ? total : 1));
            // GPU to GPU: directly over the link where the root may address the device's memory (xGMI inside a node), else staged by the
            // runtime; which it was is reported.  Every copy is queued before any is waited for; events on the root's stream time them.
            std::vector<hipEvent_t> ev(2 * parts.size(), nullptr);
            std::vector<int> enabled_here;
            uint64_t off = 0;
            int grc = 0;
            for (size_t k = 0; k < parts.size() && !grc; k++) {
                Part &P = parts[k];
                if (P.device != root) {
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, root, P.device) == hipSuccess && can) {
                        const hipError_t e = hipDeviceEnablePeerAccess(P.device, 0);
                        P.direct = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
                        if (e == hipSuccess) enabled_here.push_back(P.device);
                        (void)hipGetLastError();
                    }
                } else P.direct = true;
                if (hipEventCreate(&ev[2 * k]) != hipSuccess || hipEventCreate(&ev[2 * k + 1]) != hipSuccess) { grc = set_err(NLZM_HIP_E_NODEVICE, "hipEventCreate failed"); break; }
                (void)hipEventRecord(ev[2 * k], nullptr);
                if (P.len && hipMemcpyPeerAsync(d_all + off, root, P.d_out, P.device, P.len, nullptr) != hipSuccess)
                    grc = set_err(NLZM_HIP_E_NODEVICE, "gather from device %d failed: %s", P.device, hipGetErrorString(hipGetLastError()));
                (void)hipEventRecord(ev[2 * k + 1], nullptr);
                off += P.len;
            }
            if (!grc && hipDeviceSynchronize() != hipSuccess) grc = set_err(NLZM_HIP_E_NODEVICE, "gather failed: %s", hipGetErrorString(hipGetLastError()));
            for (size_t k = 0; k < parts.size(); k++) {
                float ms = 0;
                if (!grc && ev[2 * k] && ev[2 * k + 1] && hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]) == hipSuccess) parts[k].gather_ms = ms;
                if (ev[2 * k]) (void)hipEventDestroy(ev[2 * k]);
                if (ev[2 * k + 1]) (void)hipEventDestroy(ev[2 * k + 1]);
            }
            for (int d : enabled_here) (void)hipDeviceDisablePeerAccess(d);     // (only what this call enabled: the caller's settings stay)
            if (grc) return grc;
            HIPCHK(hipMemcpy(dst, d_all, total, hipMemcpyDeviceToHost));
            return 0;
        }();
        if (d_all) (void)hipFree(d_all);
        if (!rc) {
            if (block_len) for (uint32_t i = 0; i < ndev; i++) for (uint32_t k = 0; k < blocks_per_dev; k++) block_len[(uint64_t)i * blocks_per_dev + k] = parts[i].blens[k];
            *dst_len = total;
        }
    }
    // the job's counters (nlzm_hip_get_stats of the process-wide context reports them) and the clean-up, device by device
    memset(&g_dev0.ctx.stats, 0, sizeof g_dev0.ctx.stats);
    for (auto &P : parts) {
        uint64_t *d = (uint64_t *)&g_dev0.ctx.stats; const uint64_t *q = (const uint64_t *)&P.D.ctx.stats;
        for (size_t k = 0; k < sizeof(nlzm_hip_stats) / 8; k++) d[k] += q[k];
        (void)hipSetDevice(P.device);
        if (P.d_in) (void)hipFree(P.d_in);
        if (P.d_out) (void)hipFree(P.d_out);
        dev_shutdown(P.D);
    }
    if (g_dev0.ctx.opt_report)
        for (auto &P : parts)
            fprintf(stderr, "device %d: %llu bytes in %u blocks -- upload %.1f ms (%s), compress %.1f ms, gather %.1f ms (%s), %llu bytes out\n", P.device,
                    (unsigned long long)P.n, blocks_per_dev, P.h2d_ms, P.pinned ? "pinned" : "pageable", P.run_ms, P.gather_ms,
                    P.device == parts[0].device ? "local" : (P.direct ? "peer access" : "staged by the runtime"), (unsigned long long)P.len);
    if (dev_before >= 0) (void)hipSetDevice(dev_before);
    return rc;
}

}  // extern "C"
