// host_io.h -- FASTA ingest, listing files, skani-format TSV output (host side of the drop-in)
#pragma once
#include <cstdlib>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "engine.h"

// output of the FASTA parser: the genome's kept records already in DEVICE LAYOUT -- every record starts on a
// 32-byte boundary and is padded with 'A' to the next one -- either in memory owned by the genome (grown
// as needed: gzip input, whose inflated size is not known beforehand) or in a caller-supplied region of
// fixed capacity (plain files: straight into the pinned staging buffer, no second copy, no page faults)
struct HostGenome {
    std::string path;        // as written in the listing (byte for byte)
    std::string first_name;  // header line of the first record >= 500 bp
    std::vector<uint32_t> rec_len;   // kept records only
    std::vector<uint64_t> rec_rel;   // offset of every kept record inside the packed layout (multiples of 32)
    uint64_t packed_size = 0;        // bytes of the packed layout
    uint8_t *own = nullptr;          // malloc'ed packed layout when no region was supplied
    uint64_t n50 = 0;                // over ALL records (util.py:686-724)
    HostGenome() = default;
    HostGenome(const HostGenome &) = delete;
    HostGenome &operator=(const HostGenome &) = delete;
    HostGenome(HostGenome &&o) noexcept { *this = std::move(o); }
    HostGenome &operator=(HostGenome &&o) noexcept
    {
        if (this != &o) {
            free(own);
            path = std::move(o.path); first_name = std::move(o.first_name); rec_len = std::move(o.rec_len);
            rec_rel = std::move(o.rec_rel); packed_size = o.packed_size; own = o.own; n50 = o.n50;
            o.own = nullptr; o.packed_size = 0;
        }
        return *this;
    }
    ~HostGenome() { free(own); }
};

// plain or gzip FASTA -> kept records (>= ANI_MIN_CONTIG) in device layout.  region == nullptr: into g.own;
// else into region[0 .. region_cap) (SkError "region" if it does not fit).  Throws SkError.
// scratch: a reader thread's reusable working memory (text buffer, inflate state); nullptr: one per calling thread
struct IoScratch;
void read_fasta(const std::string &path, HostGenome &g, uint8_t *region = nullptr, size_t region_cap = 0, IoScratch *scratch = nullptr);
std::vector<std::string> read_listing(const std::string &path);

// sketch a list of genomes read from disk into `s` (batched H2D copies); names/paths returned
struct GenomeNames { std::vector<std::string> path, first_name; std::vector<uint64_t> n50; };
// threads: reader threads of this call (0: SKDER_AMD_IO_THREADS, or one per core up to 128)
void sketch_files(skder_sketches *s, const std::vector<std::string> &paths, GenomeNames &names, unsigned threads = 0);
unsigned ingest_threads();

// TSV writers. Atomic: written to a temporary name, then renamed.
void write_triangle_tsv(const std::string &out, const std::vector<skder_edge_t> &edges, const GenomeNames &names,
                        double min_af_pct);
void write_rect_tsv(const std::string &out, const std::vector<skder_edge_t> &edges, const GenomeNames &ref_names,
                    const GenomeNames &query_names, double min_af_pct);

// the same tables as row arrays (the selection step reads these, no text round trip)
std::vector<skder_edge_t> triangle_rows_ordered(const std::vector<skder_edge_t> &edges, double min_af_pct);
std::vector<skder_edge_t> rect_rows_ordered(const std::vector<skder_edge_t> &edges, double min_af_pct);
// the same orders established in place (no copy of the edge list; the callers hand the list over by swap)
void triangle_rows_order_inplace(std::vector<skder_edge_t> &edges, double min_af_pct);      // the parallel form when the host has room for it, else the serial one
void triangle_rows_order_serial(std::vector<skder_edge_t> &edges, double min_af_pct);       // one thread, strictly in place
// false: not run (memory, size), nothing changed.  small_table_bytes: up to this size the ordered list is built in a second list that is swapped in
// (and kept for the next call); above it in a raw block that is copied back and released (the harness passes 0 to take that branch on small tables)
bool triangle_rows_order_parallel(std::vector<skder_edge_t> &edges, double min_af_pct, unsigned threads, size_t small_table_bytes = 256ull << 20);
void rect_rows_order_inplace(std::vector<skder_edge_t> &edges, double min_af_pct);          // likewise
void rect_rows_order_serial(std::vector<skder_edge_t> &edges, double min_af_pct);           // pieces sorted on the host threads, merged pairwise, in place
bool rect_rows_order_parallel(std::vector<skder_edge_t> &edges, double min_af_pct, unsigned threads, size_t small_table_bytes = 256ull << 20);
void host_parallel_chunks(size_t n, const std::function<void(size_t, size_t)> &fn);          // fn(lo, hi) over pieces of [0, n) on the host threads
void write_rows_tsv(const std::string &out, const skder_edge_t *rows, size_t n, const GenomeNames &ref_names,
                    const GenomeNames &query_names);
// Concatenated_N50.txt (util.py:476-501)
void write_n50_tsv(const std::string &out, const GenomeNames &names);
// sketch store
void store_save(const std::string &out, skder_sketches *s, const GenomeNames &names);
void store_load(const std::string &path, skder_sketches *s, GenomeNames &names);
void rows_order_release_spare();       // the <= 256 MB list the row orders keep between calls goes back to the allocator
bool staging_release(int device);      // free the device's cached ingest buffers (false: in use)
