/*
 * ref_hip_binding.h -- the reference-side binding of INTEGRATION.md, as the recipe `make ref_hip` applies it.
 *
 * TEST INFRASTRUCTURE ONLY.  oracle/Makefile compiles the reference's single source file where it lies
 * (/root/reference/NLZM.cpp) with this header in front of it and ONE line added by sed after the opening brace of
 * encode_file (NLZM.cpp:1711):
 *
 *     void encode_file(FILE *fin, FILE *fout, uint32 hist_bits) {
 *     NLZM_HIP_ENCODE_FILE(fin, fout, hist_bits)            <- added
 *
 * so that the reference's own main (NLZM.cpp:2050-2178: flags, messages, file handling) drives the MI355X path
 * through the C ABI of include/nlzm_hip.h.  The macro body is the patch INTEGRATION.md shows; it uses the reference's own
 * types and helpers (byte, uint64, ASSERT, crc32_calc, _fpos64), which are defined by the time it is expanded.
 * Nothing of the reference is copied: the output binary lands in oracle/_ref/ only.
 */
#ifndef NLZM_REF_HIP_BINDING_H
#define NLZM_REF_HIP_BINDING_H

#include "../include/nlzm_hip.h"

#define NLZM_HIP_ENCODE_FILE(fin, fout, hist_bits)                                                       \
    {                                                                                                     \
        fseek(fin, 0, SEEK_END);                                                                          \
        uint64 flen_ = _fpos64(fin);                                                                      \
        fseek(fin, 0, SEEK_SET);                                                                          \
        byte *src_ = new byte[flen_ + 1];                                                                 \
        ASSERT(fread(src_, 1, flen_, fin) == flen_);                                                      \
        uint32 crc_ = crc32_calc(src_, flen_, 0);             /* display only, as at :1778 / :1899 */      \
        uint64_t cap_ = nlzm_hip_compress_bound(flen_), out_len_ = 0;                                     \
        byte *dst_ = new byte[cap_];                                                                      \
        clock_t t0_ = clock();                                                                            \
        if (nlzm_hip_init(0) || nlzm_hip_compress(src_, flen_, hist_bits, dst_, cap_, &out_len_)) {       \
            printf("Assert failed %s\n", nlzm_hip_last_error());  /* like any other ASSERT (:25) */       \
            exit(-1);                                                                                     \
        }                                                                                                 \
        fwrite(dst_, 1, out_len_, fout);                      /* header + frames + terminator (:1853, :1895) */ \
        printf("Working... %" PRIu64 " -> %" PRIu64 "\n", (uint64_t)flen_, out_len_);                     \
        printf("Done (input CRC32 %X, %.2f sec)\n", crc_, (clock() - t0_) / double(CLOCKS_PER_SEC));      \
        delete[] dst_;                                                                                    \
        delete[] src_;                                                                                    \
        return;                                                                                           \
    }

#endif
