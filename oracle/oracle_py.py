"""ctypes binding of the CPU oracle (oracle/libani_oracle.so). TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by skder_amd."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "k", "c", "marker_k", "marker_c", "min_contig", "chunk_len", "band", "bp_band", "max_gap",
        "max_lin", "anchor_score", "min_anchors", "pad", "small_pass", "rep_floor", "learned",
        "sample_window", "hash_first_step", "quarters")] + [("rule", C.c_int32 * 8)]


class Chain(C.Structure):
    _fields_ = [("score", C.c_int32), ("n_anchors", C.c_uint32), ("n_seeds", C.c_uint32),
                ("q0", C.c_uint32), ("q1", C.c_uint32), ("r0", C.c_uint32), ("r1", C.c_uint32),
                ("rctg", C.c_uint32), ("kept", C.c_uint32), ("chunk", C.c_uint32)]


class Pair(C.Structure):
    _fields_ = [("chunked_query", C.c_int32), ("n_anchors", C.c_uint32), ("n_chunks", C.c_uint32),
                ("n_chains_all", C.c_uint32), ("n_chains", C.c_uint32),
                ("sum_anchors", C.c_uint64), ("sum_seeds", C.c_uint64), ("sum_span", C.c_uint64),
                ("aligned_bases", C.c_uint64), ("cell_seeds", C.c_uint64),
                ("ani_raw", C.c_double), ("ani_span", C.c_double), ("ani", C.c_double),
                ("af_ref", C.c_double), ("af_query", C.c_double)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libani_oracle.so")
        if not os.path.isfile(path):
            raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.oracle_default_params.argtypes = [C.POINTER(Params)]
        L.oracle_genome_load.restype = C.c_void_p
        L.oracle_genome_load.argtypes = [C.c_char_p, C.POINTER(Params), C.c_char_p, C.c_size_t]
        L.oracle_genome_from_bases.restype = C.c_void_p
        L.oracle_genome_from_bases.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p,
                                               C.POINTER(Params)]
        L.oracle_genome_free.argtypes = [C.c_void_p]
        for fn in ("n_seeds", "n_markers", "n_contigs", "rep_cut"):
            getattr(L, "oracle_genome_" + fn).restype = C.c_uint32
            getattr(L, "oracle_genome_" + fn).argtypes = [C.c_void_p]
        for fn in ("total_len", "n50"):
            getattr(L, "oracle_genome_" + fn).restype = C.c_uint64
            getattr(L, "oracle_genome_" + fn).argtypes = [C.c_void_p]
        L.oracle_genome_name.restype = C.c_char_p
        L.oracle_genome_name.argtypes = [C.c_void_p]
        L.oracle_genome_seeds.argtypes = [C.c_void_p] * 5
        L.oracle_genome_markers.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_genome_contig_offsets.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_screen.restype = C.c_int
        L.oracle_screen.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(Params), C.POINTER(C.c_uint32)]
        L.oracle_pair.restype = C.c_int
        L.oracle_pair.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Params), C.POINTER(Pair),
                                  C.c_void_p, C.c_uint32]
        L.oracle_root.restype = C.c_double
        L.oracle_root.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
        L.oracle_model_ani.restype = C.c_double
        L.oracle_model_ani.argtypes = [C.c_double, C.c_double]
        L.oracle_mm_hash64.restype = C.c_uint64
        L.oracle_mm_hash64.argtypes = [C.c_uint64]
        for fn, nstr in (("oracle_triangle", 1), ("oracle_dist", 2), ("oracle_search", 2)):
            if hasattr(L, fn):
                f = getattr(L, fn)
                f.restype = C.c_int
                f.argtypes = [C.c_char_p] * nstr + [C.c_double, C.c_double, C.c_int, C.c_char_p,
                                                    C.POINTER(Params), C.c_char_p, C.c_size_t]
        _LIB = L
    return _LIB


def default_params(**kw) -> Params:
    p = Params()
    lib().oracle_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Genome:
    def __init__(self, handle):
        self.h = handle

    @classmethod
    def load(cls, path: str, p: Params) -> "Genome":
        err = C.create_string_buffer(512)
        h = lib().oracle_genome_load(path.encode(), C.byref(p), err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return cls(h)

    @classmethod
    def from_bases(cls, bases: np.ndarray, lens: np.ndarray, p: Params, file_name="", first_name="") -> "Genome":
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        h = lib().oracle_genome_from_bases(bases.ctypes.data, lens.ctypes.data, len(lens),
                                           file_name.encode(), first_name.encode(), C.byref(p))
        return cls(h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_genome_free(self.h)
            self.h = None

    n_seeds = property(lambda s: lib().oracle_genome_n_seeds(s.h))
    n_markers = property(lambda s: lib().oracle_genome_n_markers(s.h))
    n_contigs = property(lambda s: lib().oracle_genome_n_contigs(s.h))
    total_len = property(lambda s: lib().oracle_genome_total_len(s.h))
    n50 = property(lambda s: lib().oracle_genome_n50(s.h))
    rep_cut = property(lambda s: lib().oracle_genome_rep_cut(s.h))
    name = property(lambda s: lib().oracle_genome_name(s.h).decode())

    def seeds(self):
        n = self.n_seeds
        kmer = np.empty(n, np.uint64); gpos = np.empty(n, np.uint32)
        ctg = np.empty(n, np.uint32); fwd = np.empty(n, np.uint8)
        lib().oracle_genome_seeds(self.h, kmer.ctypes.data, gpos.ctypes.data, ctg.ctypes.data, fwd.ctypes.data)
        return kmer, gpos, ctg, fwd

    def markers(self):
        m = np.empty(self.n_markers, np.uint64)
        lib().oracle_genome_markers(self.h, m.ctypes.data)
        return m

    def contig_offsets(self):
        o = np.empty(self.n_contigs + 1, np.uint32)
        lib().oracle_genome_contig_offsets(self.h, o.ctypes.data)
        return o


def screen(a: Genome, b: Genome, screen_pct: float, p: Params):
    sh = C.c_uint32(0)
    ok = lib().oracle_screen(a.h, b.h, screen_pct, C.byref(p), C.byref(sh))
    return bool(ok), sh.value


def pair(ref: Genome, query: Genome, p: Params, chains: bool = False):
    out = Pair()
    if not chains:
        lib().oracle_pair(ref.h, query.h, C.byref(p), C.byref(out), None, 0)
        return out
    cap = 1 << 16
    buf = (Chain * cap)()
    lib().oracle_pair(ref.h, query.h, C.byref(p), C.byref(out), buf, cap)
    n = min(out.n_chains_all, cap)
    arr = np.frombuffer(buf, dtype=np.uint32, count=10 * n).reshape(n, 10).copy()
    return out, arr


def _drv(fn, strs, min_af, screen_pct, threads, out, p):
    err = C.create_string_buffer(1024)
    rc = getattr(lib(), fn)(*[s.encode() for s in strs], min_af, screen_pct, threads, out.encode(),
                            C.byref(p), err, 1024)
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (fn, err.value.decode()))


def triangle(listing, min_af, screen_pct, threads, out, p):
    _drv("oracle_triangle", [listing], min_af, screen_pct, threads, out, p)


def dist(ref_listing, query_listing, min_af, screen_pct, threads, out, p):
    _drv("oracle_dist", [ref_listing, query_listing], min_af, screen_pct, threads, out, p)


def search(listing_db, query_path, min_af, screen_pct, threads, out, p):
    _drv("oracle_search", [listing_db, query_path], min_af, screen_pct, threads, out, p)
