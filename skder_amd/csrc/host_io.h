// host_io.h -- FASTA ingest, listing files, skani-format TSV output (host side of the drop-in)
#pragma once
#include <string>
#include <vector>

#include "engine.h"

struct HostGenome {
    std::string path;        // as written in the listing (byte for byte)
    std::string first_name;  // header line of the first record >= 500 bp
    std::vector<uint32_t> rec_len;   // kept records only
    std::vector<uint8_t> bases;      // kept records, back to back
    uint64_t n50 = 0;                // over ALL records (util.py:686-724)
};

// plain or gzip FASTA -> kept records (>= ANI_MIN_CONTIG). Throws SkError.
void read_fasta(const std::string &path, HostGenome &g);
std::vector<std::string> read_listing(const std::string &path);

// sketch a list of genomes read from disk into `s` (batched H2D copies); names/paths returned
struct GenomeNames { std::vector<std::string> path, first_name; std::vector<uint64_t> n50; };
void sketch_files(skder_sketches *s, const std::vector<std::string> &paths, GenomeNames &names);

// TSV writers. Atomic: written to a temporary name, then renamed.
void write_triangle_tsv(const std::string &out, const std::vector<skder_edge_t> &edges, const GenomeNames &names,
                        double min_af_pct);
void write_rect_tsv(const std::string &out, const std::vector<skder_edge_t> &edges, const GenomeNames &ref_names,
                    const GenomeNames &query_names, double min_af_pct);

// the same tables as row arrays (the selection step reads these, no text round trip)
std::vector<skder_edge_t> triangle_rows_ordered(const std::vector<skder_edge_t> &edges, double min_af_pct);
std::vector<skder_edge_t> rect_rows_ordered(const std::vector<skder_edge_t> &edges, double min_af_pct);
void write_rows_tsv(const std::string &out, const skder_edge_t *rows, size_t n, const GenomeNames &ref_names,
                    const GenomeNames &query_names);
// Concatenated_N50.txt (util.py:476-501)
void write_n50_tsv(const std::string &out, const GenomeNames &names);
// sketch store
void store_save(const std::string &out, skder_sketches *s, const GenomeNames &names);
void store_load(const std::string &path, skder_sketches *s, GenomeNames &names);
