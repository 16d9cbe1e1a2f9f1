// gunzip.h -- whole-buffer gzip decoder of the ingest (gunzip.cpp): every member of a .gz file held in memory -> its text
#pragma once
#include <cstddef>
#include <cstdint>

enum GunzipStatus {
    GUNZIP_OK = 0,
    GUNZIP_NOT_GZIP,      // the buffer does not begin with a gzip member
    GUNZIP_CORRUPT,       // invalid header, block, code or distance; CRC-32 or length mismatch
    GUNZIP_TRUNCATED,     // the input ends inside a member
    GUNZIP_OUTPUT_FULL    // the text is longer than the output buffer
};

// Decodes the gzip members of in[0 .. in_len) into out[0 .. out_cap); *out_len = bytes of text.  Members that follow one
// another decode to the concatenation of their texts (bgzip writes thousands); bytes behind the last member that do not
// begin another one are ignored (zlib's gzread does the same).  The CRC-32 and the length in every trailer are checked.
// Reads only inside in[], writes only inside out[].
GunzipStatus gunzip_buffer(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len);

// CRC-32 of the gzip trailer (carry-less multiplication where the CPU has it, slicing tables elsewhere)
uint32_t gunzip_crc32(uint32_t crc, const uint8_t *p, size_t n);
const char *gunzip_status_text(GunzipStatus s);
