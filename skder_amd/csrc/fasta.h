// fasta.h -- FASTA text -> device layout on the device (fasta.hip)
#pragma once
#include "common.h"

struct FastaFile {          // one file of a batch (host fills, the kernel reads)
    uint64_t text_off;      // offset of the file's text in the text buffer: the byte in front of it is '\n', at least 80 bytes of
                            // '\n' follow its end
    uint64_t out_off;       // offset of the file's region in the bases buffer (a multiple of 32)
    uint32_t text_len;
    uint32_t out_cap;       // bytes of the region: bound(text_len)
    uint32_t rec_cap;       // entries of each of its three tables
    uint32_t table_off;     // offset of its tables in the batch's table arrays
    uint32_t tile_off;      // first of its 4 KB tiles among the batch's tiles (tiled parser; ceil(text_len / 4096) tiles per file)
    uint32_t pad;
};
struct FastaResult {        // one per file (the kernel writes)
    uint32_t n_kept;        // records of >= ANI_MIN_CONTIG bases: kept_rel / kept_len hold their starts (relative to out_off) and lengths
    uint32_t n_lens;        // lengths of ALL non-empty records (text in front of the first header counts as one): for N50
    uint32_t flags;         // FASTA_HOST_* : the host parses this file instead
    uint32_t packed_size;   // bytes of the region in use
    uint32_t first_hdr;     // text offset (inside the file) of the '>' of the first kept record's header line
    uint32_t pad[3];
};
#define FASTA_HOST_BLANKS 1u     /* blanks inside sequence lines, or '\r' without '\n' */
#define FASTA_HOST_HUGE 2u       /* a record of 2^31 bases or more */
#define FASTA_HOST_TABLE 4u      /* more records than the table holds */
#define FASTA_HOST_REGION 8u     /* output region too small (cannot happen with out_cap = bound(text_len)) */

// The tiled parser (three kernels: tile summaries, a scan over every file's tiles, the writes): total_tiles = sum of the files'
// tiles; d_work: fasta_tiles_work_bytes() bytes of device memory, 16-byte aligned, the caller's until the stream has passed the
// kernels.  d_files must carry tile_off.
size_t fasta_tiles_work_bytes(uint32_t total_tiles, uint32_t n_files);
void fasta_parse_tiles_launch(const uint8_t *d_text, const FastaFile *d_files, uint32_t n_files, uint32_t total_tiles, void *d_work, uint8_t *d_bases,
                              uint32_t *d_kept_rel, uint32_t *d_kept_len, uint32_t *d_all_len, FastaResult *d_results, hipStream_t st);
// one wavefront per file (round 3's first device parser; SKDER_AMD_FASTA_WAVE=1)
void fasta_parse_launch(const uint8_t *d_text, const FastaFile *d_files, uint32_t n_files, uint8_t *d_bases, uint32_t *d_kept_rel,
                        uint32_t *d_kept_len, uint32_t *d_all_len, FastaResult *d_results, hipStream_t st);
