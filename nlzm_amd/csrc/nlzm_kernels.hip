// nlzm_kernels.hip -- gfx950 kernels of the NLZM compress path.
//
//   rk_hash_kernel      RK256 rolling hash of every 256-byte window, one thread per
//                       position (closed form of MatchFinderRK256's roll, NLZM.cpp:798-799,
//                       1071-1083): pure, position-parallel, input tile staged in LDS.
//   master_kernel       the serial half (nlzm_core.h): HT2/HT3/RK256 state, match-table
//                       chain, forward-graph parse, model, symbol emit.  One workgroup.
//   rans_frames_kernel  CodeFrame::Flush (NLZM.cpp:590-640): 4 interleaved rANS states
//                       per frame, renormalisation words placed by a prefix scan.
//   gather_frames_kernel concatenates the frames into the output stream.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nlzm_core.h"

namespace nlzm {

// ---------------------------------------------------------------------------
// 64-lane wave policy for the master (block = one wave)
// ---------------------------------------------------------------------------
struct DevWave {
    static __device__ __forceinline__ uint32_t lane() { return threadIdx.x & 63u; }
    static __device__ __forceinline__ uint32_t width() { return 64u; }
    static __device__ __forceinline__ void sync() { __syncthreads(); }
    static __device__ __forceinline__ void sync_global()
    {
        // same CU, same L1: a workgroup-scope fence orders lane 0's stores before
        // every lane's later loads
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
    }
    static __device__ __forceinline__ uint32_t rmin(uint32_t v)
    {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v = umin(v, (uint32_t)__shfl_xor((int)v, m, 64));
        return v;
    }
    static __device__ __forceinline__ uint32_t ror(uint32_t v)
    {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v |= (uint32_t)__shfl_xor((int)v, m, 64);
        return v;
    }
    // Up to 8 byte compares at once: group k (8 lanes x 8 bytes) compares
    // in[sp[k] ..] with in[a ..] up to cap[k] bytes; len[k] = common prefix.
    static __device__ __forceinline__ void cmp_multi(const uint8_t *in, const uint32_t sp[8], uint32_t a,
                                                     const uint32_t cap[8], uint32_t valid, uint32_t len[8])
    {
        const uint32_t l = lane(), grp = l >> 3, j = l & 7;
        uint32_t mysp = 0, mycap = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) if (grp == (uint32_t)k) { mysp = sp[k]; mycap = cap[k]; }
        bool active = (valid >> grp) & 1u;
        if (!active) mycap = 0;
        uint32_t res = mycap;          // no mismatch below cap => cap
        uint32_t off = 0;
        while (__any(active && off < mycap)) {
            uint32_t m = kNone;
            const uint32_t my = off + j * 8;
            if (active && my < mycap) {
                const unsigned long long d = load64u(in + mysp + my) ^ load64u(in + a + my);
                if (d) {
                    const uint32_t pos = my + ((uint32_t)__builtin_ctzll(d) >> 3);
                    if (pos < mycap) m = pos;
                }
            }
            m = umin(m, (uint32_t)__shfl_xor((int)m, 1, 64));
            m = umin(m, (uint32_t)__shfl_xor((int)m, 2, 64));
            m = umin(m, (uint32_t)__shfl_xor((int)m, 4, 64));
            if (active && m != kNone) { res = m; active = false; }
            off += 64;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) len[k] = (uint32_t)__shfl((int)res, k * 8, 64);
    }
};

// ---------------------------------------------------------------------------
// RK256 hash of every window: rkhash[a] = sum_{j<256} in[a+j] * ADDH^(256-j)
// ---------------------------------------------------------------------------
constexpr uint32_t kRkAddh = 0x2F0FD693u;       // NLZM.cpp:793

__global__ __launch_bounds__(256) void rk_hash_kernel(const uint8_t *__restrict__ in, unsigned long long n,
                                                      unsigned long long pos0, unsigned long long pos1,
                                                      uint32_t *__restrict__ out)
{
    // positions [pos0, pos1); a block covers 1024 consecutive positions, 4 per thread
    __shared__ uint8_t tile[1024 + 256 + 16];
    const unsigned long long blk0 = pos0 + (unsigned long long)blockIdx.x * 1024;
    for (uint32_t i = threadIdx.x; i < 1024 + 256; i += 256) {
        const unsigned long long a = blk0 + i;
        tile[i] = a < n ? in[a] : 0;
    }
    __syncthreads();
    // thread t handles positions blk0 + t + 256*k (k<4): lanes read consecutive bytes
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t o = threadIdx.x + 256u * k;
        const unsigned long long a = blk0 + o;
        if (a >= pos1 || a + 256 > n) continue;
        uint32_t h = 0;
#pragma unroll 16
        for (int j = 0; j < 256; j++) h = (h + tile[o + j]) * kRkAddh;
        out[a] = h;
    }
}

// ---------------------------------------------------------------------------
// master
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void master_kernel(Geom g, Globals G, uint32_t c0, uint32_t c1)
{
    __shared__ MasterLds lds;
    Master<DevWave> m;
    m.g = g; m.G = G; m.L = &lds;
    m.run(c0, c1);
}

// ---------------------------------------------------------------------------
// frame coder.  One workgroup (256 threads) per frame.
//   phase 1: lanes 0..3 of wave 0 run the four interleaved states over symbols
//            n-1..0 (state i&3 takes symbol i, NLZM.cpp:599-603) and record, per
//            symbol, the 16-bit word it pushed out (or none).
//   phase 2: block-wide exclusive scan over the "pushed" flags gives every word
//            its place: in memory order the words appear by ascending symbol index
//            after the four flushed states (NLZM.cpp:444-455, 605-612).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rans_frames_kernel(const uint32_t *__restrict__ syms, unsigned long long syms_stride,
                                                          const uint8_t *__restrict__ bits, unsigned long long bits_stride,
                                                          FrameMeta *__restrict__ fmeta, uint32_t *__restrict__ scratch,
                                                          unsigned long long scratch_stride,
                                                          uint8_t *__restrict__ out, unsigned long long out_stride,
                                                          uint32_t out_cap)
{
    const uint32_t f = blockIdx.x;
    const uint32_t n = fmeta[f].nsyms, nb = fmeta[f].nbits_bytes, ops = fmeta[f].num_ops;
    const uint32_t *s = syms + f * syms_stride;
    uint32_t *w = scratch + f * scratch_stride;         // per symbol: pushed word or kNone
    uint8_t *o = out + f * out_stride;
    __shared__ uint32_t st_final[4];
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t carry_s;

    if (threadIdx.x < 4) {
        uint32_t x = 1u << 16;                           // RANS_MID, NLZM.cpp:442,600
        const uint32_t k = threadIdx.x;
        // largest index i < n with (i & 3) == k
        if (n > k) {
            for (long long i = (long long)(((n - 1 - k) & ~3u) + k); i >= 0; i -= 4) {
                const uint32_t v = s[i];
                const uint32_t start = v & 0xFFFFu, freq = v >> 16;
                uint32_t pushed = kNone;
                if (x >= (freq << 18)) {                 // x_max = ((RANS_MID >> 14) << 16) * freq
                    pushed = x & 0xFFFFu;
                    x >>= 16;
                }
                w[i] = pushed;
                x = ((x / freq) << 14) + (x % freq) + start;
            }
        }
        st_final[k] = x;
    }
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();

    // header + bit bytes
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < nb; i += 256) if (12 + i < out_cap) o[12 + i] = bits[f * bits_stride + i];
    uint8_t *r = o + 12 + nb;
    if (threadIdx.x < 16 && 12 + nb + 16 <= out_cap) {
        const uint32_t x = st_final[threadIdx.x >> 2];  // st[0] at the lowest address, little-endian
        r[threadIdx.x] = (uint8_t)(x >> (8 * (threadIdx.x & 3)));
    }
    // words: position = 16 + 2 * (#pushed among symbols < i); bytes hi, lo (NLZM.cpp:449-450 written backwards)
    for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t word = i < n ? w[i] : kNone;
        const bool has = word != kNone;
        const unsigned long long bal = __ballot(has);
        const uint32_t before = (uint32_t)__popcll(bal & ((1ull << lane) - 1));
        if (lane == 0) wave_tot[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = carry_s;
        for (uint32_t k = 0; k < wv; k++) off += wave_tot[k];
        if (has) {
            const uint32_t at = 12 + nb + 16 + 2 * (off + before);
            if (at + 2 <= out_cap) { o[at] = (uint8_t)(word >> 8); o[at + 1] = (uint8_t)word; }
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t nrans = 16 + 2 * carry_s;
        // header: num_ops, nbits (incl. 12 header bytes and the pad), nrans -- big-endian (NLZM.cpp:614-628)
        const uint32_t hdr[3] = { ops, 12 + nb, nrans };
        for (int k = 0; k < 3; k++) {
            o[4 * k + 0] = (uint8_t)(hdr[k] >> 24); o[4 * k + 1] = (uint8_t)(hdr[k] >> 16);
            o[4 * k + 2] = (uint8_t)(hdr[k] >> 8);  o[4 * k + 3] = (uint8_t)hdr[k];
        }
        fmeta[f].out_len = 12 + nb + nrans;
    }
}

// Copy frame f (out_len bytes at frames + f*stride) to dst + dst_off[f].
__global__ __launch_bounds__(256) void gather_frames_kernel(const uint8_t *__restrict__ frames, unsigned long long stride,
                                                            const unsigned long long *__restrict__ dst_off,
                                                            const FrameMeta *__restrict__ fmeta, uint8_t *__restrict__ dst)
{
    const uint32_t f = blockIdx.x;
    const uint32_t len = fmeta[f].out_len;
    const uint8_t *s = frames + f * stride;
    uint8_t *d = dst + dst_off[f];
    for (uint32_t i = threadIdx.x; i < len; i += 256) d[i] = s[i];
}

// ---- launch wrappers (called from nlzm_hip.cpp) ------------------------------
void launch_rk_hash(const uint8_t *in, unsigned long long n, unsigned long long pos0, unsigned long long pos1,
                    uint32_t *out, hipStream_t st)
{
    if (pos1 <= pos0) return;
    const unsigned long long blocks = (pos1 - pos0 + 1023) / 1024;
    hipLaunchKernelGGL(rk_hash_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, in, n, pos0, pos1, out);
}

void launch_master(const Geom &g, const Globals &G, uint32_t c0, uint32_t c1, hipStream_t st)
{
    hipLaunchKernelGGL(master_kernel, dim3(1), dim3(64), 0, st, g, G, c0, c1);
}

void launch_rans(const uint32_t *syms, unsigned long long syms_stride, const uint8_t *bits, unsigned long long bits_stride,
                 FrameMeta *fmeta, uint32_t *scratch, unsigned long long scratch_stride, uint8_t *out,
                 unsigned long long out_stride, uint32_t out_cap, uint32_t nframes, hipStream_t st)
{
    if (!nframes) return;
    hipLaunchKernelGGL(rans_frames_kernel, dim3(nframes), dim3(256), 0, st, syms, syms_stride, bits, bits_stride, fmeta,
                       scratch, scratch_stride, out, out_stride, out_cap);
}

void launch_gather(const uint8_t *frames, unsigned long long stride, const unsigned long long *dst_off,
                   const FrameMeta *fmeta, uint8_t *dst, uint32_t nframes, hipStream_t st)
{
    if (!nframes) return;
    hipLaunchKernelGGL(gather_frames_kernel, dim3(nframes), dim3(256), 0, st, frames, stride, dst_off, fmeta, dst);
}

}  // namespace nlzm
