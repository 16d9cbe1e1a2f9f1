"""ctypes loader for libskder_amd.so (the C ABI declared in include/skder_amd.h).

The library is built in-tree by `make -C skder_amd/csrc` (or `__graft_entry__.build()`).  There is no
Python or CPU fallback: a missing library is an ImportError-like RuntimeError, and every compute
entry point fails on a machine without a gfx950 device."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libskder_amd.so")
SKDER_TILE = 8192
ERRLEN = 2048


class Batch(C.Structure):
    _fields_ = [("n_genomes", C.c_uint32), ("n_records", C.c_uint32), ("rec_off", C.c_void_p),
                ("rec_len", C.c_void_p), ("genome_rec_begin", C.c_void_p)]


class RawView(C.Structure):
    _fields_ = [("n_genomes", C.c_uint32), ("n_seeds", C.c_uint64), ("n_markers", C.c_uint64),
                ("n_rec_goff", C.c_uint64), ("d_seed_kmer", C.c_void_p), ("d_seed_gpos", C.c_void_p),
                ("d_seed_ctg", C.c_void_p), ("d_markers", C.c_void_p), ("h_seed_off", C.c_void_p),
                ("h_marker_off", C.c_void_p), ("h_genome_len", C.c_void_p), ("h_genome_nrec", C.c_void_p),
                ("h_rec_goff", C.c_void_p)]


class Edge(C.Structure):
    _fields_ = [("ref", C.c_uint32), ("query", C.c_uint32), ("ani", C.c_double), ("af_ref", C.c_double),
                ("af_query", C.c_double), ("n_chains", C.c_uint32), ("n_anchors", C.c_uint32),
                ("aligned_bases", C.c_uint64), ("sum_anchors", C.c_uint64), ("sum_seeds", C.c_uint64),
                ("cell_seeds", C.c_uint64), ("ani_raw", C.c_double)]


# every symbol include/skder_amd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "skder_amd_triangle": (C.c_int, [C.c_char_p, C.c_double, C.c_double, C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]),
    "skder_amd_dist": (C.c_int, [C.c_char_p, C.c_char_p, C.c_double, C.c_double, C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]),
    "skder_amd_sketch": (C.c_void_p, [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]),
    "skder_amd_search": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double, C.c_double, C.c_char_p, C.c_char_p, C.c_size_t]),
    "skder_amd_db_free": (None, [C.c_void_p]),
    "skder_amd_parse_skani_params": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.c_char_p, C.c_size_t]),
    "skder_amd_ctx_create": (C.c_void_p, [C.c_int, C.c_char_p, C.c_size_t]),
    "skder_amd_ctx_destroy": (None, [C.c_void_p]),
    "skder_amd_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "skder_amd_last_error": (C.c_char_p, [C.c_void_p]),
    "skder_amd_sketches_new": (C.c_void_p, [C.c_void_p]),
    "skder_amd_sketches_free": (None, [C.c_void_p]),
    "skder_amd_sketch_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Batch)]),
    "skder_amd_sketches_reserve": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "skder_amd_sketches_view": (C.c_int, [C.c_void_p, C.POINTER(RawView)]),
    "skder_amd_sketches_append_raw": (C.c_int, [C.c_void_p, C.POINTER(RawView)]),
    "skder_amd_sketches_index": (C.c_int, [C.c_void_p]),
    "skder_amd_sketches_index_part": (C.c_int, [C.c_void_p, C.c_void_p]),
    "skder_amd_sketches_rep_cuts": (C.c_int, [C.c_void_p, C.c_void_p]),
    "skder_amd_sketches_set_rep_cuts": (C.c_int, [C.c_void_p, C.c_void_p]),
    "skder_amd_screen_rows": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.POINTER(C.POINTER(C.c_uint32)),
                                        C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint64)]),
    "skder_amd_pairs_probed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "skder_amd_chain_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                        C.POINTER(C.POINTER(Edge)), C.POINTER(C.c_uint64)]),
    "skder_amd_triangle_rows": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double,
                                          C.POINTER(C.POINTER(Edge)), C.POINTER(C.c_uint64)]),
    "skder_amd_rectangle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.POINTER(Edge)),
                                      C.POINTER(C.c_uint64)]),
    "skder_amd_copy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "skder_amd_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "skder_amd_select_greedy": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_char_p),
                                         C.c_double, C.c_double, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                         C.c_char_p, C.c_size_t]),
    "skder_amd_select_dynamic": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_char_p),
                                          C.c_double, C.c_double, C.c_double, C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                          C.c_char_p, C.c_size_t]),
    "skder_amd_select_clusters": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32),
                                           C.c_uint32, C.c_double, C.c_double, C.c_char_p, C.c_char_p, C.c_size_t]),
    "skder_amd_rows_pass": (C.c_int, [C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "skder_amd_pct2_cents": (C.c_int64, [C.c_float]),
    "skder_amd_descend_lengths": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "skder_amd_descend_fill": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]),
    "skder_amd_peer_fallbacks": (C.c_uint32, []),
    "skder_amd_set_ani_output": (C.c_int, [C.c_int]),
    "skder_amd_release_cached_buffers": (C.c_int, [C.c_int]),
    "skder_amd_last_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "skder_amd_last_index_ms": (C.c_double, [C.c_void_p]),
    "skder_amd_last_runs_ms": (C.c_double, [C.c_void_p]),
    "skder_amd_synth_fill": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p]),
    "skder_amd_triangle_n50": (C.c_int, [C.c_char_p, C.c_double, C.c_double, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p,
                                         C.c_size_t]),
    "skder_amd_sketch_n50": (C.c_void_p, [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]),
    "skder_amd_triangle_multi": (C.c_int, [C.c_char_p, C.c_double, C.c_double, C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_char_p,
                                           C.c_char_p, C.c_size_t]),
    "skder_amd_sketch_multi": (C.c_void_p, [C.c_char_p, C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]),
    "skder_amd_db_from_sketches": (C.c_void_p, [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p, C.c_char_p,
                                               C.c_size_t]),
    "skder_amd_db_size": (C.c_uint32, [C.c_void_p]),
    "skder_amd_db_path": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "skder_amd_db_n50": (C.c_uint64, [C.c_void_p, C.c_uint32]),
    "skder_amd_db_triangle": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_char_p, C.POINTER(C.POINTER(Edge)),
                                        C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]),
    "skder_amd_search_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, C.c_double, C.c_double,
                                         C.POINTER(C.c_char_p), C.POINTER(C.POINTER(Edge)), C.POINTER(C.c_uint64),
                                         C.c_char_p, C.c_size_t]),
    "skder_amd_search_batch_live": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, C.c_double, C.c_double,
                                              C.POINTER(C.c_char_p), C.c_void_p, C.POINTER(C.POINTER(Edge)), C.POINTER(C.c_uint64),
                                              C.c_char_p, C.c_size_t]),
    "skder_amd_db_save": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "skder_amd_db_load": (C.c_void_p, [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]),
    "skder_amd_debug_genome": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_LIB = None


def _preload_pytorch_hip_runtime():
    """PyTorch wheels ship their own copy of libamdhip64; libskder_amd.so links the system one.  Two
    copies of the HIP runtime in one process cannot both own the GPU (the second sees no device), and
    which one came first would depend on import order.  When PyTorch is installed its copy is loaded
    first and globally, so that this library's dependency resolves to the same runtime PyTorch uses."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.isfile(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libskder_amd.so; raises RuntimeError if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError("libskder_amd.so is missing (run `make -C skder_amd/csrc`); "
                               "skder_amd has no CPU fallback")
        _preload_pytorch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)      # AttributeError if a declared symbol is not exported
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB
