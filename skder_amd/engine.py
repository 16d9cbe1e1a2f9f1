"""Device-level Python view of libskder_amd.so (section B of include/skder_amd.h).

torch is used here ONLY to own device memory and (in multigpu.py) to drive RCCL; every kernel is
launched by the C library on its own HIP stream."""
import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib
from ._lib import Batch, Edge, RawView, SKDER_TILE

EDGE_DTYPE = np.dtype([("ref", "<u4"), ("query", "<u4"), ("ani", "<f8"), ("af_ref", "<f8"), ("af_query", "<f8"),
                       ("n_chains", "<u4"), ("n_anchors", "<u4"), ("aligned_bases", "<u8"),
                       ("sum_anchors", "<u8"), ("sum_seeds", "<u8"), ("cell_seeds", "<u8"), ("ani_raw", "<f8")])
assert EDGE_DTYPE.itemsize == C.sizeof(Edge)
# skder_descendant_t (include/skder_amd.h): a descendant of a real assembly for the device generator
DESCENDANT_DTYPE = np.dtype([("parent", "<u4"), ("sub_ppm", "<u4"), ("indel_ppm", "<u4"), ("n_events", "<u4"), ("seed", "<u8"),
                             ("ev", [("rec", "<u4"), ("type", "<u4"), ("s", "<u4"), ("n", "<u4"), ("b", "<u4")], (3,)), ("pad", "<u4")])
assert DESCENDANT_DTYPE.itemsize == 88


class BatchLayout:
    """Device layout of a batch of genomes: every kept record starts on a 32-byte boundary, 32 readable
    bytes precede the first record and SKDER_TILE+64 follow the last (include/skder_amd.h)."""

    def __init__(self, rec_lens: Sequence[np.ndarray]):
        self.n_genomes = len(rec_lens)
        self.genome_rec_begin = np.zeros(self.n_genomes + 1, np.uint32)
        lens = []
        for g, rl in enumerate(rec_lens):
            rl = np.asarray(rl, np.uint32)
            if (rl < 500).any():
                raise ValueError("records shorter than 500 bp must be dropped before layout")
            lens.append(rl)
            self.genome_rec_begin[g + 1] = self.genome_rec_begin[g] + len(rl)
        self.rec_len = np.concatenate(lens) if lens else np.zeros(0, np.uint32)
        padded = (self.rec_len.astype(np.uint64) + np.uint64(31)) & ~np.uint64(31)
        self.rec_off = np.zeros(len(self.rec_len), np.uint64)
        if len(padded):
            self.rec_off[0] = 32
            self.rec_off[1:] = 32 + np.cumsum(padded)[:-1]
        self.payload_end = int(32 + padded.sum())
        self.total_bytes = self.payload_end + SKDER_TILE + 64
        self.total_bases = int(self.rec_len.astype(np.uint64).sum())

    def c_batch(self) -> Batch:
        b = Batch()
        b.n_genomes = self.n_genomes
        b.n_records = len(self.rec_len)
        b.rec_off = self.rec_off.ctypes.data
        b.rec_len = self.rec_len.ctypes.data
        b.genome_rec_begin = self.genome_rec_begin.ctypes.data
        return b

    def pack_host(self, genomes_bases: Sequence[np.ndarray]) -> np.ndarray:
        """host buffer in device layout from per-genome ASCII bases (records back to back)"""
        buf = np.full(self.total_bytes, ord("A"), np.uint8)
        r = 0
        for g, bases in enumerate(genomes_bases):
            src = 0
            for _ in range(self.genome_rec_begin[g], self.genome_rec_begin[g + 1]):
                l = int(self.rec_len[r])
                o = int(self.rec_off[r])
                buf[o:o + l] = bases[src:src + l]
                src += l
                r += 1
        return buf


class Context:
    def __init__(self, device: int = 0):
        err = C.create_string_buffer(_lib.ERRLEN)
        self.h = _lib.lib().skder_amd_ctx_create(device, err, _lib.ERRLEN)
        if not self.h:
            raise RuntimeError("skder_amd: " + err.value.decode())
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            try:
                _lib.lib().skder_amd_ctx_destroy(self.h)
            except Exception:      # interpreter shutdown: the module globals may already be gone
                pass
            self.h = None

    def __del__(self):
        self.close()

    def check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError("skder_amd %s failed: %s" % (what, _lib.lib().skder_amd_last_error(self.h).decode()))

    @property
    def stream(self) -> int:
        return _lib.lib().skder_amd_ctx_stream(self.h)

    def timing(self) -> np.ndarray:
        out = (C.c_double * 8)()
        _lib.lib().skder_amd_last_timing(self.h, out)
        return np.array(out[:])

    def index_ms(self) -> float:
        """device milliseconds of the last seed-index build"""
        return float(_lib.lib().skder_amd_last_index_ms(self.h))

    def runs_ms(self) -> float:
        """device milliseconds of the run-extraction kernel in the last triangle_rows / rectangle call"""
        return float(_lib.lib().skder_amd_last_runs_ms(self.h))

    def counters(self) -> np.ndarray:
        out = (C.c_uint64 * 4)()
        _lib.lib().skder_amd_last_counters(self.h, out)
        return np.array(out[:], np.uint64)

    def synth_fill(self, d_bases_ptr: int, layout: BatchLayout, lineage: np.ndarray, params: np.ndarray):
        lineage = np.ascontiguousarray(lineage, np.uint64)
        params = np.ascontiguousarray(params, np.uint32)
        b = layout.c_batch()
        self.check(_lib.lib().skder_amd_synth_fill(self.h, d_bases_ptr, C.byref(b), lineage.ctypes.data,
                                                   params.ctypes.data), "synth_fill")


    # ---- descendants of real assemblies, generated on the device (skder_amd/csrc/descend.hip)
    def descend_lengths(self, d_anc_ptr: int, anc_layout: BatchLayout, desc: np.ndarray) -> np.ndarray:
        """length of every record of every descendant (one entry per record of its parent, in order)"""
        desc = np.ascontiguousarray(desc, DESCENDANT_DTYPE)
        nrec = int(sum(int(anc_layout.genome_rec_begin[p + 1] - anc_layout.genome_rec_begin[p]) for p in desc["parent"]))
        out = np.zeros(max(nrec, 1), np.uint32)
        b = anc_layout.c_batch()
        self.check(_lib.lib().skder_amd_descend_lengths(self.h, d_anc_ptr, C.byref(b), desc.ctypes.data, len(desc), out.ctypes.data, nrec), "descend_lengths")
        return out[:nrec]

    def descend_fill(self, d_anc_ptr: int, anc_layout: BatchLayout, desc: np.ndarray, d_out_ptr: int, rec_out_off: np.ndarray):
        """the bases; rec_out_off: where each record goes in the output buffer (2^64 - 1: dropped)"""
        desc = np.ascontiguousarray(desc, DESCENDANT_DTYPE)
        off = np.ascontiguousarray(rec_out_off, np.uint64)
        b = anc_layout.c_batch()
        self.check(_lib.lib().skder_amd_descend_fill(self.h, d_anc_ptr, C.byref(b), desc.ctypes.data, len(desc), d_out_ptr, off.ctypes.data, len(off)), "descend_fill")

    def descendants(self, d_anc_ptr: int, anc_layout: BatchLayout, desc: np.ndarray, torch):
        """lengths, layout (records of 500 bases and more), bases: -> (device tensor, BatchLayout of the descendants)"""
        lens = self.descend_lengths(d_anc_ptr, anc_layout, desc)
        per, at = [], 0
        for p in desc["parent"]:
            k = int(anc_layout.genome_rec_begin[p + 1] - anc_layout.genome_rec_begin[p])
            per.append(lens[at:at + k])
            at += k
        layout = BatchLayout([l[l >= 500] for l in per])
        off = np.full(len(lens), np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64)
        off[lens >= 500] = layout.rec_off
        d = torch.full((layout.total_bytes,), ord("A"), dtype=torch.uint8, device="cuda:%d" % self.device)
        self.descend_fill(d_anc_ptr, anc_layout, desc, d.data_ptr(), off)
        return d, layout


def _np_from(ptr, n, dtype):
    if not n:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dtype).itemsize,)).view(dtype).copy()


class Sketches:
    """A set of sketched genomes resident in HBM."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.h = _lib.lib().skder_amd_sketches_new(ctx.h)
        if not self.h:
            raise RuntimeError("skder_amd_sketches_new failed")

    def close(self):
        if getattr(self, "h", None):
            try:
                if getattr(self.ctx, "h", None):
                    _lib.lib().skder_amd_sketches_free(self.h)
            except Exception:
                pass
            self.h = None

    def __del__(self):
        self.close()

    def reserve(self, n_seeds: int, n_markers: int):
        """capacity hint for the totals over all batches (avoids re-allocation while batches are appended)"""
        self.ctx.check(_lib.lib().skder_amd_sketches_reserve(self.h, int(n_seeds), int(n_markers)), "sketches_reserve")

    def sketch_batch(self, d_bases_ptr: int, layout: BatchLayout):
        b = layout.c_batch()
        self.ctx.check(_lib.lib().skder_amd_sketch_batch(self.h, d_bases_ptr, C.byref(b)), "sketch_batch")

    def view(self) -> dict:
        """raw sketch arrays: device pointers for seeds/markers, numpy copies of the host metadata"""
        v = RawView()
        self.ctx.check(_lib.lib().skder_amd_sketches_view(self.h, C.byref(v)), "sketches_view")
        G = v.n_genomes
        return dict(n_genomes=G, n_seeds=v.n_seeds, n_markers=v.n_markers,
                    d_seed_kmer=v.d_seed_kmer, d_seed_gpos=v.d_seed_gpos, d_seed_ctg=v.d_seed_ctg, d_markers=v.d_markers,
                    seed_off=_np_from(v.h_seed_off, G + 1, np.uint64), marker_off=_np_from(v.h_marker_off, G + 1, np.uint64),
                    genome_len=_np_from(v.h_genome_len, G, np.uint64), genome_nrec=_np_from(v.h_genome_nrec, G, np.uint32),
                    rec_goff=_np_from(v.h_rec_goff, v.n_rec_goff, np.uint32))

    def append_raw(self, n_genomes, d_seed_kmer, d_seed_gpos, d_seed_ctg, d_markers, seed_off, marker_off, genome_len,
                   genome_nrec, rec_goff):
        seed_off = np.ascontiguousarray(seed_off, np.uint64)
        marker_off = np.ascontiguousarray(marker_off, np.uint64)
        genome_len = np.ascontiguousarray(genome_len, np.uint64)
        genome_nrec = np.ascontiguousarray(genome_nrec, np.uint32)
        rec_goff = np.ascontiguousarray(rec_goff, np.uint32)
        v = RawView()
        v.n_genomes = n_genomes
        v.n_seeds = int(seed_off[-1] - seed_off[0])
        v.n_markers = int(marker_off[-1] - marker_off[0])
        v.n_rec_goff = len(rec_goff)
        v.d_seed_kmer, v.d_seed_gpos, v.d_seed_ctg, v.d_markers = d_seed_kmer, d_seed_gpos, d_seed_ctg, d_markers
        v.h_seed_off = seed_off.ctypes.data
        v.h_marker_off = marker_off.ctypes.data
        v.h_genome_len = genome_len.ctypes.data
        v.h_genome_nrec = genome_nrec.ctypes.data
        v.h_rec_goff = rec_goff.ctypes.data
        self.ctx.check(_lib.lib().skder_amd_sketches_append_raw(self.h, C.byref(v)), "sketches_append_raw")

    def index(self):
        self.ctx.check(_lib.lib().skder_amd_sketches_index(self.h), "sketches_index")

    # ---- one pair matrix over several GPUs (include/skder_amd.h, "one pair matrix over several GPUs")
    def index_part(self, full: np.ndarray):
        """bucket index for the genomes with full[g] != 0 (the ones this GPU owns), chunk tables for the others"""
        full = np.ascontiguousarray(full, np.uint8)
        self.ctx.check(_lib.lib().skder_amd_sketches_index_part(self.h, full.ctypes.data), "sketches_index_part")

    def rep_cuts(self, n_genomes: int) -> np.ndarray:
        out = np.empty(n_genomes, np.uint32)
        self.ctx.check(_lib.lib().skder_amd_sketches_rep_cuts(self.h, out.ctypes.data), "sketches_rep_cuts")
        return out

    def set_rep_cuts(self, values: np.ndarray):
        values = np.ascontiguousarray(values, np.uint32)
        self.ctx.check(_lib.lib().skder_amd_sketches_set_rep_cuts(self.h, values.ctypes.data), "sketches_set_rep_cuts")

    def screen_rows(self, row_begin: int, row_stride: int, screen_pct: float):
        """candidate pairs (ref, query) of triangle rows row_begin, row_begin + row_stride, ..."""
        pr, pq, n = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.c_uint64(0)
        self.ctx.check(_lib.lib().skder_amd_screen_rows(self.h, row_begin, row_stride, screen_pct, C.byref(pr), C.byref(pq), C.byref(n)),
                       "screen_rows")
        return _np_from(pr, n.value, np.uint32), _np_from(pq, n.value, np.uint32)

    def pairs_probed(self, ref: np.ndarray, query: np.ndarray, queries: "Sketches" = None) -> np.ndarray:
        """index of the genome each pair probes (the pair is chained on the GPU that owns it)"""
        ref = np.ascontiguousarray(ref, np.uint32)
        query = np.ascontiguousarray(query, np.uint32)
        out = np.empty(len(ref), np.uint32)
        self.ctx.check(_lib.lib().skder_amd_pairs_probed(self.h, (queries or self).h, ref.ctypes.data, query.ctypes.data, len(ref),
                                                         out.ctypes.data, None), "pairs_probed")
        return out

    def chain_pairs(self, ref: np.ndarray, query: np.ndarray, queries: "Sketches" = None, copy: bool = True) -> np.ndarray:
        ref = np.ascontiguousarray(ref, np.uint32)
        query = np.ascontiguousarray(query, np.uint32)
        p, n = C.POINTER(Edge)(), C.c_uint64(0)
        self.ctx.check(_lib.lib().skder_amd_chain_pairs(self.h, (queries or self).h, ref.ctypes.data, query.ctypes.data, len(ref),
                                                        C.byref(p), C.byref(n)), "chain_pairs")
        return self._edges(p, n, copy)

    def _edges(self, p, n, copy=True) -> np.ndarray:
        if not n.value:
            return np.zeros(0, EDGE_DTYPE)
        raw = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value * EDGE_DTYPE.itemsize,))
        return raw.view(EDGE_DTYPE).copy() if copy else raw.view(EDGE_DTYPE)

    def triangle_rows(self, row_begin: int = 0, row_stride: int = 1, screen_pct: float = 80.0, copy: bool = True) -> np.ndarray:
        """copy=False: a view of the library's edge buffer, valid until the next call on this context"""
        p = C.POINTER(Edge)()
        n = C.c_uint64(0)
        self.ctx.check(_lib.lib().skder_amd_triangle_rows(self.h, row_begin, row_stride, screen_pct, C.byref(p), C.byref(n)),
                       "triangle_rows")
        return self._edges(p, n, copy)

    def rectangle(self, queries: "Sketches", screen_pct: float = 80.0, copy: bool = True) -> np.ndarray:
        p = C.POINTER(Edge)()
        n = C.c_uint64(0)
        self.ctx.check(_lib.lib().skder_amd_rectangle(self.h, queries.h, screen_pct, C.byref(p), C.byref(n)), "rectangle")
        return self._edges(p, n, copy)

    def debug_genome(self, g: int, n_seeds: int) -> dict:
        nch, rep, bits = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        sk = np.zeros(n_seeds, np.uint32); sg = np.zeros(n_seeds, np.uint32)
        sc = np.zeros(n_seeds, np.uint32); pc = np.zeros(n_seeds, np.uint32)
        self.ctx.check(_lib.lib().skder_amd_debug_genome(self.h, g, C.byref(nch), C.byref(rep), C.byref(bits),
                                                         sk.ctypes.data, sg.ctypes.data, sc.ctypes.data, pc.ctypes.data),
                       "debug_genome")
        return dict(n_chunks=nch.value, rep_cut=rep.value, bucket_bits=bits.value, skmer=sk, sgpos=sg, sctg=sc, pchunk=pc)


def _copy_d2d(dst_ptr: int, src_ptr: int, nbytes: int, ctx=None):
    """device-to-device copy, complete on return: through the library (its own stream) when a context
    is given, else through the HIP runtime this process already has loaded"""
    if ctx is not None:
        ctx.check(_lib.lib().skder_amd_copy_d2d(ctx.h, dst_ptr, src_ptr, nbytes), "copy_d2d")
        return
    hip = C.CDLL(None)                      # the runtime loaded globally by _lib (never a second copy)
    if not hasattr(hip, "hipMemcpy"):
        hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rc = hip.hipMemcpy(dst_ptr, src_ptr, nbytes, 3)
    if rc != 0:
        raise RuntimeError("hipMemcpy failed: %d" % rc)


def download(ptr: int, n: int, dtype, ctx=None) -> np.ndarray:
    """copy n elements from a device pointer (torch as the allocator/copier)"""
    import torch
    nbytes = n * np.dtype(dtype).itemsize
    if not nbytes:
        return np.zeros(0, dtype)
    t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    _copy_d2d(t.data_ptr(), ptr, nbytes, ctx)
    return t.cpu().numpy().view(dtype)


class _DeviceArray:
    """a raw device pointer dressed for torch.as_tensor (the CUDA array interface, which torch's HIP build reads too)"""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}


def device_view(ptr: int, n: int, torch_dtype, device: int = None):
    """n elements at a raw device pointer as a torch tensor WITHOUT a copy; the memory stays the owner's (keep it alive while the view is
    used).  device: the GPU the pointer lives on (default: torch's current device)"""
    import torch
    dev = "cuda" if device is None else "cuda:%d" % device
    if n <= 0 or not ptr:
        return torch.empty(0, dtype=torch_dtype, device=dev)
    typestr = {torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1"}[torch_dtype]
    return torch.as_tensor(_DeviceArray(ptr, n, typestr), device=dev)


def download_tensor(ptr: int, n: int, torch_dtype, ctx=None):
    """copy n elements from a raw device pointer into a new torch tensor on the current device"""
    import torch
    t = torch.empty(max(n, 0), dtype=torch_dtype, device="cuda")
    if n:
        torch.cuda.synchronize()
        _copy_d2d(t.data_ptr(), ptr, n * t.element_size(), ctx)
    return t
