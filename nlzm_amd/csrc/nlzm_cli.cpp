// nlzm_cli.cpp -- host side of the MI355X build: NLZM's own command line
// (NLZM.cpp:2050-2178) in one file.  `c` runs the compress path on the GPU through
// the C ABI of include/nlzm_hip.h and fails if no gfx950 device is present;
// `d`/`t` decode on the host (the decoder is a serial byte-copy machine and stays
// on the CPU: SURVEY.md 8f-1); `h` prints the CRC32.
//
// Messages, flag handling and exit codes follow the reference:
//   flags lower-cased, leading '-' stripped, -window:N clamped to [15,28]   :2074-2092
//   "Error: %s already exists" / "Error: %s file does not exist", return -1   :2095-2112
//   banner and usage text                                                     :2051, :2165-2171
#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <thread>
#include <string>
#include <vector>

#include "../../include/nlzm_hip.h"

namespace {

// ---- CRC32 (crc32_calc, NLZM.cpp:126-199; display only, never stored) ---------
uint32_t crc_table[256];
void crc_init()
{
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        crc_table[n] = c;
    }
}
uint32_t crc_calc(const uint8_t *p, uint64_t n, uint32_t crc)
{
    uint32_t c = crc ^ 0xFFFFFFFFu;
    while (n--) c = crc_table[(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

bool slurp(const char *path, std::vector<uint8_t> &buf)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseeko(f, 0, SEEK_END);
    const long long sz = ftello(f);
    fseeko(f, 0, SEEK_SET);
    buf.resize((size_t)sz);
    const bool ok = !sz || fread(buf.data(), 1, (size_t)sz, f) == (size_t)sz;
    fclose(f);
    return ok;
}

// ---- host decoder: decode_file (NLZM.cpp:1912-2039) -----------------------------
struct Cdf { uint16_t c[17]; };
struct Model {
    uint32_t rep[4];
    Cdf cmd, lit_hi, lit_lo[16], len_direct, len_ext_hi, len_ext_lo[16], slot_hi[4], slot_lo[4][8];
};
void cdf_set(Cdf &d, int ns) { for (int i = 0; i <= ns; i++) d.c[i] = (uint16_t)(i * (16384 / ns)); }
void model_init(Model &m)
{
    for (int i = 0; i < 4; i++) m.rep[i] = (uint32_t)i + 1;
    cdf_set(m.cmd, 4); cdf_set(m.lit_hi, 16); cdf_set(m.len_direct, 8); cdf_set(m.len_ext_hi, 16);
    for (int i = 0; i < 16; i++) { cdf_set(m.lit_lo[i], 16); cdf_set(m.len_ext_lo[i], 16); }
    for (int c = 0; c < 4; c++) { cdf_set(m.slot_hi[c], 8); for (int i = 0; i < 8; i++) cdf_set(m.slot_lo[c][i], 8); }
}
inline void cdf_adapt(Cdf &d, int ns, int y)
{
    for (int i = 0; i < ns; i++) {
        const int mix = i <= y ? i : 16384 + i + (127 - ns);                       // :284-298
        d.c[i] = (uint16_t)(d.c[i] + ((mix - (int)d.c[i]) >> 7));                 // :348-382
    }
}
inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct Frame {
    const uint8_t *bits, *rans, *end;
    uint32_t word = 0, word_bits = 0, num_ops = 0, st[4], idx = 0;
    bool bad = false;
    int sym(Cdf &d, int nbits)                                                    // ReadCDF, :666-712
    {
        num_ops--;
        uint32_t &rs = st[idx++ & 3];
        const uint32_t f = rs & 16383u;
        int y = 0;
        for (int step = 1 << (nbits - 1); step; step >>= 1) y += step * (f >= d.c[y + step]);   // :388-433
        const uint32_t start = d.c[y], freq = (uint32_t)d.c[y + 1] - start;
        uint32_t x = freq * (rs >> 14) + f - start;                               // :457-459
        if (x < 65536u) {                                                         // :481-488
            if (rans + 2 > end) { bad = true; return 0; }
            x = (x << 16) + ((uint32_t)rans[0] << 8) + rans[1];
            rans += 2;
        }
        rs = x;
        cdf_adapt(d, 1 << nbits, y);
        return y;
    }
    uint32_t raw(uint32_t nb)                                                     // ReadBits, :714-731
    {
        num_ops--;
        while (word_bits < 24) {
            if (bits >= end) { bad = true; return 0; }
            word |= (uint32_t)*bits++ << (24 - word_bits);
            word_bits += 8;
        }
        const uint32_t y = word >> (32 - nb);
        word <<= nb; word_bits -= nb;
        return y;
    }
};
inline uint32_t match_min(uint32_t d) { return 2u + (d >= 256u) + (d >= 4096u) + (d >= (1u << 20)); }   // :813-821
inline void rep_add(uint32_t r[4], uint32_t d)
{
    if (r[0] == d || r[1] == d || r[2] == d || r[3] == d) return;
    r[3] = r[2]; r[2] = r[1]; r[1] = r[0]; r[0] = d;
}
uint32_t dec_len(Frame &f, Model &m)                                              // model_decode_lv, :1369-1383
{
    uint32_t lv = (uint32_t)f.sym(m.len_direct, 3);
    if (lv == 7) {
        const int hi = f.sym(m.len_ext_hi, 4);
        const int lo = f.sym(m.len_ext_lo[hi], 4);
        lv += ((uint32_t)hi << 4) + (uint32_t)lo;
    }
    return lv;
}

// A byte range of the file as the decoder sees it
struct Span {
    const uint8_t *p; size_t n;
    size_t size() const { return n; }
    const uint8_t &operator[](size_t i) const { return p[i]; }
    const uint8_t *data() const { return p; }
};

// Length of the stream that starts at in[0]: header, frames hopped over by the sizes their headers carry (:645-663),
// terminator (:646-648).  0: malformed.  Block mode (k independent streams back to back) is split with this.
size_t stream_length(const Span &in)
{
    if (in.size() < 8) return 0;
    size_t pos = 4;
    for (;;) {
        if (pos + 4 > in.size()) return 0;
        if (!be32(&in[pos])) return pos + 4;
        if (pos + 12 > in.size()) return 0;
        const uint32_t nb = be32(&in[pos + 4]), nr = be32(&in[pos + 8]);
        if (nb < 12 || nr < 16 || pos + (size_t)nb + nr > in.size()) return 0;
        pos += (size_t)nb + nr;
    }
}

// returns 0 or a negative code; out receives the decoded bytes
int decode_stream(const Span &in, std::vector<uint8_t> &out, uint32_t *hist_bits, uint32_t *frame_bits)
{
    if (in.size() < 8) return -1;
    const uint32_t hb = ((uint32_t)in[0] << 8) + in[1], fb = ((uint32_t)in[2] << 8) + in[3];
    *hist_bits = hb; *frame_bits = fb;
    // the reference asserts 12 <= hist_bits (:1918) although its encoder can write 10 or 11 for
    // inputs under 2 KiB (:1716); this decoder accepts those streams too
    if (hb < 10 || hb > 28 || fb < 12 || fb > 20) return -2;
    Model *m = new Model;
    model_init(*m);
    size_t pos = 4;
    int rc = 0;
    for (;;) {
        if (pos + 4 > in.size()) { rc = -3; break; }
        Frame f;
        f.num_ops = be32(&in[pos]);
        if (!f.num_ops) break;
        if (pos + 12 > in.size()) { rc = -3; break; }
        const uint32_t nb = be32(&in[pos + 4]), nr = be32(&in[pos + 8]);
        if (nb < 12 || nr < 16 || pos + (size_t)nb + nr > in.size()) { rc = -3; break; }
        f.bits = &in[pos + 12]; f.rans = &in[pos + nb]; f.end = in.data() + pos + nb + nr;
        for (int i = 0; i < 4; i++) { f.st[i] = f.rans[0] | (f.rans[1] << 8) | (f.rans[2] << 16) | ((uint32_t)f.rans[3] << 24); f.rans += 4; }
        while (f.num_ops > 0 && !f.bad) {
            const int cmd = f.sym(m->cmd, 2);
            if (cmd == 0) {
                const int hi = f.sym(m->lit_hi, 4);
                const int lo = f.sym(m->lit_lo[hi], 4);
                out.push_back((uint8_t)((hi << 4) + lo));
                continue;
            }
            uint32_t lv, dv;
            if (cmd == 1) {
                lv = dec_len(f, *m);
                const uint32_t lc = lv < 3 ? lv : 3;
                const int shi = f.sym(m->slot_hi[lc], 3);
                const int slo = f.sym(m->slot_lo[lc][shi], 3);
                dv = ((uint32_t)shi << 3) + (uint32_t)slo;
                if (dv >= 4) {                                                    // :1395-1413
                    uint32_t ab = (dv >> 1) - 1;
                    dv = (2 + (dv & 1)) << ab;
                    if (ab < 4) dv += f.raw(ab);
                    else { ab -= 4; if (ab > 0) dv += f.raw(ab) << 4; dv += f.raw(4); }
                }
                dv += 1;
            } else if (cmd == 2) {
                const uint32_t ri = f.raw(2);
                lv = dec_len(f, *m);
                dv = m->rep[ri];
            } else { rc = -5; break; }
            lv += match_min(dv);
            rep_add(m->rep, dv);
            if (dv > out.size()) { rc = -6; break; }
            const size_t from = out.size() - dv;
            for (uint32_t i = 0; i < lv; i++) out.push_back(out[from + i]);
        }
        if (rc) break;
        if (f.bad) { rc = -7; break; }
        pos += (size_t)nb + nr;
    }
    delete m;
    return rc;
}

void lower(char *v) { for (; *v; v++) *v = (char)(*v | 0x20); }

}  // namespace

int main(int argc, char **argv)
{
    printf("NLZM 1.03 - Written by Nauful (MI355X/gfx950 build)\n");
    // (-blocks:k runs k streams' kernels side by side: the HIP runtime needs more than its default of 4 hardware queues for that, and reads
    //  the variable at the first HIP call -- nlzm_hip_init sets it too, this is for a main that touches HIP before it)
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    crc_init();
    uint32_t hist_bits = 22;                                                      // :2071
    uint32_t nblocks = 1;                           // -blocks:k (not in the reference): k independent streams, back to back
    uint32_t ngpus = 0;                             // -gpus:g (not in the reference): the blocks on g GPUs of this node, -blocks:k on each
    while (argc >= 2 && *argv[1] == '-') {
        char *arg = argv[1];
        argv++; argc--;
        lower(arg);
        while (*arg == '-') ++arg;
        if (!strncmp(arg, "window:", 7)) {
            const int v = atoi(arg + 7);
            hist_bits = (uint32_t)(v < 15 ? 15 : (v > 28 ? 28 : v));
            printf("Window bits: %d\n", hist_bits);
        } else if (!strncmp(arg, "blocks:", 7)) {
            const int v = atoi(arg + 7);
            nblocks = (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
            printf("Blocks: %d\n", nblocks);
        } else if (!strncmp(arg, "gpus:", 5)) {
            const int v = atoi(arg + 5);
            ngpus = (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
            printf("GPUs: %d\n", ngpus);
        } else {
            printf("Unrecognized flag %s\n", arg);
            return -1;
        }
    }
    const int cmd = argc >= 2 ? (argv[1][0] | 0x20) : 0;
    if (argc == 4 && cmd == 'c') {
        if (FILE *probe = fopen(argv[3], "rb")) { printf("Error: %s already exists\n", argv[3]); fclose(probe); return -1; }
        const bool one_stream = !ngpus && nblocks == 1;
        std::vector<uint8_t> in;
        FILE *fin = nullptr;
        uint64_t in_size = 0;
        if (one_stream) {       // read, uploaded and compressed piece by piece (nlzm_hip_feed_*): the file is never whole in host memory
            fin = fopen(argv[2], "rb");
            if (!fin) { printf("Error: %s file does not exist\n", argv[2]); return -1; }
            fseeko(fin, 0, SEEK_END);
            in_size = (uint64_t)ftello(fin);
            fseeko(fin, 0, SEEK_SET);
        } else {
            if (!slurp(argv[2], in)) { printf("Error: %s file does not exist\n", argv[2]); return -1; }
            in_size = in.size();
        }
        FILE *fout = fopen(argv[3], "wb");
        if (!fout) { printf("Error: %s file does not exist\n", argv[3]); if (fin) fclose(fin); return -1; }
        if (!ngpus && nlzm_hip_init(0)) { printf("Error: %s\n", nlzm_hip_last_error()); fclose(fout); remove(argv[3]); return -1; }
        const uint32_t nstreams = ngpus ? ngpus * nblocks : nblocks;
        uint32_t hb, fb, cs, feed;
        nlzm_hip_geometry(in_size, hist_bits, &hb, &fb, &cs, &feed);
        {   // the reference's summary (:1755-1759): sizes of its own structures for this window
            const uint32_t c1 = hb < 15 ? 15 : (hb > 17 ? 17 : hb), c2 = hb < 16 ? 16 : (hb > 20 ? 20 : hb), c3 = hb < 16 ? 16 : (hb > 22 ? 22 : hb);
            const uint64_t mf = (4ull << 12) + (8ull << (12 + c1 - 15)) + (4ull << (13 + c2 - 16)) + (8ull << hb) + (4ull << (15 + c3 - 16));
            printf("Model: %d KB\n", 2);
            printf("Parser: %d KB\n", 65);
            printf("Dictionary: %d KB\n", (int)(((1ull << hb) + 1023) >> 10));
            printf("Frame: %d KB\n", (int)(((1u << fb) + 1023) >> 10));
            printf("Dictionary search: %d KB\n", (int)((mf + 1023) >> 10));
        }
        if (nstreams > 1)
            printf("Note: %u independent streams are written back to back; this program's d/t read them, the reference's d "
                   "stops after the first\n", nstreams);
        printf("Working...\r");
        uint64_t out_n = 0;
        const clock_t t0 = clock();
        struct timespec w0, w1;
        clock_gettime(CLOCK_MONOTONIC, &w0);
        if (one_stream) {
            // the reference's loop in big steps: read a piece, hand it over, write the frames that are finished (:1774-1778, :1853, :1870-1885)
            std::vector<uint8_t> piece(16u << 20), obuf(16u << 20);
            uint32_t crc = 0;
            int rc = nlzm_hip_feed_begin(in_size, hist_bits);
            auto drain = [&]() {
                for (uint64_t got = 1; !rc && got;) {
                    rc = nlzm_hip_feed_output(obuf.data(), obuf.size(), &got);
                    if (!rc && got) { fwrite(obuf.data(), 1, (size_t)got, fout); out_n += got; }
                }
            };
            for (uint64_t done = 0; !rc && done < in_size;) {
                const size_t m = (size_t)(in_size - done < piece.size() ? in_size - done : piece.size());
                if (fread(piece.data(), 1, m, fin) != m) { printf("Error: %s could not be read\n", argv[2]); rc = -1; break; }
                crc = crc_calc(piece.data(), m, crc);
                rc = nlzm_hip_feed(piece.data(), m);
                done += m;
                drain();
                printf("Working... %" PRIu64 " -> %" PRIu64 "\r", done, out_n);
                fflush(stdout);
            }
            if (!rc) rc = nlzm_hip_feed_finish();
            drain();
            nlzm_hip_feed_end();
            fclose(fin);
            clock_gettime(CLOCK_MONOTONIC, &w1);
            if (rc) { if (rc != -1) printf("Error: %s\n", nlzm_hip_last_error()); fclose(fout); remove(argv[3]); return -1; }
            fclose(fout);
            printf("Working... %" PRIu64 " -> %" PRIu64 "\n", in_size, out_n);
            printf("Done (input CRC32 %X, %.2f sec)\n", crc, (double)(w1.tv_sec - w0.tv_sec) + 1e-9 * (double)(w1.tv_nsec - w0.tv_nsec));
            nlzm_hip_shutdown();
            return 0;
        }
        std::vector<uint8_t> out(nlzm_hip_compress_bound(in.size()) + (size_t)nstreams * (16 + 131072));
        std::vector<uint64_t> blen(nstreams);
        std::vector<int> devs;
        for (uint32_t d = 0; d < ngpus; d++) devs.push_back((int)d);
        const int rc = ngpus ? nlzm_hip_compress_blocks_multi(devs.data(), ngpus, nblocks, in.data(), in.size(), hist_bits, out.data(), out.size(), blen.data(), &out_n)
                     : nblocks > 1 ? nlzm_hip_compress_blocks(in.data(), in.size(), nblocks, hist_bits, out.data(), out.size(), blen.data(), &out_n)
                                   : nlzm_hip_compress(in.data(), in.size(), hist_bits, out.data(), out.size(), &out_n);
        clock_gettime(CLOCK_MONOTONIC, &w1);
        (void)t0;
        if (rc) { printf("Error: %s\n", nlzm_hip_last_error()); fclose(fout); remove(argv[3]); return -1; }
        fwrite(out.data(), 1, (size_t)out_n, fout);
        fclose(fout);
        if (nstreams > 1) {
            // the block index, a sidecar: where every block's stream starts, how long it is and how many input bytes it holds.  The
            // streams stay self-delimiting -- the index only saves d/t the hop over every frame header of every block before the parallel decode can
            // start, and keeps the later blocks' boundaries when the container is damaged inside an earlier one.
            const std::string ip = std::string(argv[3]) + ".idx";
            if (FILE *fi = fopen(ip.c_str(), "wb")) {
                fprintf(fi, "NLZMIDX 1 %u %" PRIu64 " %" PRIu64 "\n", nstreams, (uint64_t)in.size(), out_n);
                const uint64_t per = (in.size() + nstreams - 1) / nstreams;
                uint64_t off = 0;
                for (uint32_t i = 0; i < nstreams; i++) {
                    const uint64_t lo = (uint64_t)i * per < in.size() ? (uint64_t)i * per : in.size(), hi = lo + per < in.size() ? lo + per : in.size();
                    fprintf(fi, "%" PRIu64 " %" PRIu64 " %" PRIu64 "\n", off, blen[i], hi - lo);
                    off += blen[i];
                }
                fclose(fi);
                printf("Block index: %s\n", ip.c_str());
            }
        }
        printf("Working... %" PRIu64 " -> %" PRIu64 "\n", (uint64_t)in.size(), out_n);
        printf("Done (input CRC32 %X, %.2f sec)\n", crc_calc(in.data(), in.size(), 0),
               (double)(w1.tv_sec - w0.tv_sec) + 1e-9 * (double)(w1.tv_nsec - w0.tv_nsec));
        nlzm_hip_shutdown();
    } else if ((argc == 4 && cmd == 'd') || (argc == 3 && cmd == 't')) {
        if (cmd == 'd') {
            if (FILE *probe = fopen(argv[3], "rb")) { printf("Error: %s already exists\n", argv[3]); fclose(probe); return -1; }
        }
        std::vector<uint8_t> in, out;
        if (!slurp(argv[2], in)) { printf("Error: %s file does not exist\n", argv[2]); return -1; }
        FILE *fout = nullptr;
        if (cmd == 'd') {
            fout = fopen(argv[3], "wb");
            if (!fout) { printf("Error: %s file does not exist\n", argv[3]); return -1; }
        }
        uint32_t hb = 0, fb = 0;
        const clock_t t0 = clock();
        // one stream (the reference's format), or several back to back (block mode): found by hopping over the frames,
        // decoded on a host thread each, written in order
        std::vector<Span> parts;
        size_t cut_tail = 0;
        bool by_index = false;
        {   // the sidecar index of a block container, if it is there and fits the file: the blocks' boundaries without hopping over their frames
            const std::string ip = std::string(argv[2]) + ".idx";
            if (FILE *fi = fopen(ip.c_str(), "rb")) {
                unsigned ver = 0, k = 0;
                unsigned long long n_in = 0, n_out = 0;
                std::vector<Span> idx;
                bool ok = fscanf(fi, "NLZMIDX %u %u %llu %llu", &ver, &k, &n_in, &n_out) == 4 && ver == 1 && k >= 1 && k <= 4096 && n_out >= in.size();
                unsigned long long expect = 0;
                bool cut = false;
                for (unsigned i = 0; ok && i < k && !cut; i++) {
                    unsigned long long off = 0, len = 0, raw = 0;
                    ok = fscanf(fi, "%llu %llu %llu", &off, &len, &raw) == 3 && off == expect && len >= 8;
                    if (ok && off + len > in.size()) { cut = true; break; }      // (the file ends inside this block: the ones in front of it are whole)
                    // (a block's stream starts with its header and ends with its terminator, :1915-1921, :646-648)
                    if (ok) ok = in[off] == 0 && in[off + 1] >= 10 && in[off + 1] <= 28 && be32(&in[off + len - 4]) == 0;
                    if (ok) { idx.push_back(Span{ in.data() + off, (size_t)len }); expect = off + len; }
                }
                fclose(fi);
                if (ok && !idx.empty() && (cut || (expect == in.size() && n_out == in.size()))) {
                    parts = idx; by_index = true;
                    if (cut) cut_tail = in.size() - (size_t)expect;
                } else printf("Note: %s does not fit this file; the blocks are found by their frame headers\n", ip.c_str());
            }
        }
        for (size_t pos = 0; !by_index && pos < in.size();) {
            const Span rest{ in.data() + pos, in.size() - pos };
            const size_t len = stream_length(rest);
            if (!len) {
                // what follows is not a stream: the reference stops at the first terminator (:646-648) and so do we;
                // a container that is cut off inside its first stream is malformed
                if (parts.empty()) break;
                // a further stream header (:1915-1921: hist_bits, frame_bits, both big-endian 16-bit) with a frame that is cut off:
                // the complete streams are decoded and written, and the exit status says that the container was cut
                const bool header = rest.n >= 4 && rest.p[0] == 0 && rest.p[1] >= 10 && rest.p[1] <= 28 && rest.p[2] == 0 && rest.p[3] >= 12 && rest.p[3] <= 20;
                if (header) cut_tail = rest.n;
                else printf("Note: %zu bytes after the last stream ignored\n", rest.n);
                break;
            }
            parts.push_back(Span{ rest.p, len });
            pos += len;
        }
        int rc = parts.empty() ? -3 : 0;
        if (parts.size() == 1) rc = decode_stream(parts[0], out, &hb, &fb);
        else if (!rc) {
            std::vector<std::vector<uint8_t>> outs(parts.size());
            std::vector<int> rcs(parts.size(), 0);
            std::vector<uint32_t> hbs(parts.size(), 0), fbs(parts.size(), 0);
            std::vector<std::thread> th;
            for (size_t i = 0; i < parts.size(); i++)
                th.emplace_back([&, i] { rcs[i] = decode_stream(parts[i], outs[i], &hbs[i], &fbs[i]); });
            for (auto &t : th) t.join();
            for (size_t i = 0; i < parts.size() && !rc; i++) { rc = rcs[i]; out.insert(out.end(), outs[i].begin(), outs[i].end()); }
            hb = hbs[0]; fb = fbs[0];
            printf("Blocks: %d\n", (int)parts.size());
        }
        if (rc) { printf("Assert failed: malformed stream (%d)\n", rc); if (fout) fclose(fout); return -1; }
        printf("Dictionary: %d KB\n", (int)(((1ull << hb) + 1023) >> 10));
        printf("Frame: %d KB\n", (int)(((1u << fb) + 1023) >> 10));
        if (fout) { fwrite(out.data(), 1, out.size(), fout); fclose(fout); }
        printf("Working... %" PRIu64 " -> %" PRIu64 "\n", (uint64_t)in.size(), (uint64_t)out.size());
        printf("Done (output CRC32 %X, %.2f sec)\n", crc_calc(out.data(), out.size(), 0), (clock() - t0) / (double)CLOCKS_PER_SEC);
        if (cut_tail) { printf("Error: the container is cut off inside block %zu (%zu bytes of it present); %zu complete blocks decoded\n", parts.size() + 1, cut_tail, parts.size()); return -2; }
    } else if (argc == 3 && cmd == 'h') {
        std::vector<uint8_t> in;
        if (!slurp(argv[2], in)) { printf("Error: %s file does not exist\n", argv[2]); return -1; }
        printf("%X\n", crc_calc(in.data(), in.size(), 0));
    } else {
        printf("Commands:\n"
               "\t[flags] c [input] [output] - Compress input file to output file (best parser)\n"
               "\td [input] [output] - Decompress input file to output file\n"
               "\tt [input] - Decompress input file in memory\n"
               "\th [input] - Calculate CRC32 for input file\n"
               "Flags:\n"
               "\t-window:bits = Maximum window size in bits, default 22 (4 MB), min 15, max 28 (32 KB to 256 MB)\n"
               "\t-blocks:k = (this build) compress k independent blocks at once; d/t read the streams back to back\n"
               "\t-gpus:g = (this build) the blocks on GPUs 0..g-1 of this node, -blocks:k of them on each\n");
    }
    return 0;
}
