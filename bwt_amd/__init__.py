"""ctypes binding of libbwt_hip.so (the C-ABI declared in include/bwt_hip.h).

This is the Python-side view of the product for tests, the smoke test and the
bench.  It contains no algorithm: every call goes to the HIP library, and
importing it fails loudly when the library has not been built.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BWT_HIP_LIB") or os.path.join(_HERE, "lib", "libbwt_hip.so")  # the variable is for A/B experiments with a variant build

_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


class BwtHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bwt_hip error %d: %s" % (code, msg))
        self.code = code


def build(verbose=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], check=True,
                   stdout=None if verbose else subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `make -C bwt_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.bwt_hip_last_error.restype = C.c_char_p
    sig = {
        "bwt_hip_create": [i32, C.POINTER(vp)],
        "bwt_hip_set_text": [vp, _u8p, i64],
        "bwt_hip_set_text_device": [vp, vp, i64],
        "bwt_hip_gen_text": [vp, i64, C.c_uint64, i32],
        "bwt_hip_get_text": [vp, _u8p],
        "bwt_hip_leaf": [vp, i64, i64, i64, C.POINTER(vp)],
        "bwt_hip_merge": [vp, vp, vp, i64, i64, C.POINTER(vp)],
        "bwt_hip_rec": [vp, i64, i64, i64, i64, i64, C.POINTER(vp)],
        "bwt_hip_pmbwt": [vp, i64, i64, i64, C.POINTER(vp)],
        "bwt_hip_pmbwt_host": [vp, _u8p, i64, i64, i64, i64, _u8p, C.POINTER(vp)],
        "bwt_hip_block_upload": [vp, i64, i64, _u8p, i64, i64, _i64p, _i64p, i64, C.POINTER(vp)],
        "bwt_hip_block_export_device": [vp, vp, vp, vp, vp],
        "bwt_hip_block_import_device": [vp, i64, i64, vp, i64, i64, vp, vp, i64, C.POINTER(vp)],
        "bwt_hip_block_info": [vp, _i64p, _i64p, _i64p, _i64p, C.POINTER(i32), _i64p],
        "bwt_hip_block_get_L": [vp, vp, _u8p],
        "bwt_hip_block_L_device": [vp, C.POINTER(vp)],
        "bwt_hip_block_get_samples": [vp, vp, _i64p, _i64p, i64],
        "bwt_hip_leaf_sa": [vp, i64, i64, _i64p],
        "bwt_hip_block_init_extra": [vp, vp],
        "bwt_hip_block_get_C": [vp, vp, i32, _i64p],
        "bwt_hip_block_rank": [vp, vp, i64, _u8p, _i64p, _i64p],
        "bwt_hip_block_LF": [vp, vp, i64, _u8p, _i64p, _i64p],
        "bwt_hip_gap": [vp, vp, vp, i64, _i64p, _i64p],
        "bwt_hip_dc_sample": [vp, i64, _i64p, _i64p, _i64p, C.POINTER(i32)],
        "bwt_hip_leaf_capacity": [vp, i64, _i64p],
        "bwt_hip_symbol_codes": [vp, C.POINTER(i32), _u8p, _u8p],
        "bwt_hip_dist_leaf": [vp, i64, i64, i64, i32, i32, vp, i64, vp, _i64p, _i64p, _i64p, _i64p, _i64p, C.POINTER(i32)],
        "bwt_hip_block_ibwt": [vp, vp, _u8p, _i64p],
        "bwt_hip_block_sa": [vp, vp, i64, _i64p, _i64p],
        "bwt_hip_dist_bounds": [i64, i32, _i64p],
        "bwt_hip_dist_chains": [vp, vp, i64, i64, i64, vp],
        "bwt_hip_dist_partition": [vp, vp, i64, i64, i32, vp, _i64p],
        "bwt_hip_dist_wire_packable": [vp, C.POINTER(i32)],
        "bwt_hip_dist_wire_pack": [vp, vp, i64, vp],
        "bwt_hip_dist_wire_unpack": [vp, vp, i64, vp],
        "bwt_hip_dist_merge_slice": [vp, vp, i64, vp, i64, vp, vp, vp],
        "bwt_hip_dist_merge_slice_noalloc": [vp, vp, i64, vp, i64, vp, vp, vp],
        "bwt_hip_dist_samples_route": [vp, i32, vp, vp, i64, i64, i64, i64, i64, i32, i32, vp, vp, _i64p],
        "bwt_hip_dist_layout": [vp, i64, i32, _i64p, C.POINTER(i32)],
        "bwt_hip_dist_chains_grouped": [vp, vp, i64, i64, i64, vp, vp, i64, vp, _i64p, _i64p, _i64p],
        "bwt_hip_dist_merge_slice_grouped": [vp, vp, i64, vp, i64, vp, _i64p, vp, i32, i64, vp, vp],
        "bwt_hip_dist_samples_answer": [vp, vp, i64, i64, vp, i64, i64, i64, vp, vp],
        "bwt_hip_dist_samples_place": [vp, i32, vp, vp, i64, vp, vp, i64, i64, i64, i64, i64, vp, vp, _i64p],
        "bwt_hip_set_timing": [vp, i32],
        "bwt_hip_reset_timers": [vp],
        "bwt_hip_timer_count": [vp],
        "bwt_hip_timer_get": [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_double), _i64p],
        "bwt_hip_mem_stats": [vp, _i64p, _i64p, _i64p],
        "bwt_hip_trim": [vp],
        "bwt_hip_sync": [vp],
        "bwt_hip_test_sort_u64": [vp, _u64p, _u32p, i64, i32, i32],
        "bwt_hip_test_sort_u32": [vp, _u32p, _u32p, i64, i32, i32],
        "bwt_hip_test_scan_u32": [vp, _u32p, i64, i32],
    }
    sig.update({
        "bwt_hip_rccl_unique_id": [_u8p],
        "bwt_hip_transport_rccl": [vp, _u8p, i32, i32, C.POINTER(vp)],
        "bwt_hip_transport_local": [i32, C.POINTER(vp)],
        "bwt_hip_dist_pmbwt": [vp, vp, i64, i64, i64, C.POINTER(vp)],
        "bwt_hip_slice_info": [vp, _i64p, _i64p, _i64p, _i64p, _i64p],
        "bwt_hip_slice_L_device": [vp, C.POINTER(vp)],
        "bwt_hip_slice_get_L": [vp, vp, _u8p],
        "bwt_hip_slice_get_samples": [vp, vp, _i64p, _i64p, i64],
        "bwt_hip_pmbwt_multi": [C.POINTER(i32), i32, _u8p, i64, i64, i64, i64, _u8p, _i64p, _i64p, _i64p, _i64p, i64, _i64p,
                                C.POINTER(i32)],
    })
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = i32
    L.bwt_hip_destroy.argtypes = [vp]
    L.bwt_hip_destroy.restype = None
    L.bwt_hip_block_free.argtypes = [vp, vp]
    L.bwt_hip_block_free.restype = None
    L.bwt_hip_host_alloc.argtypes = [i64, C.POINTER(vp)]
    L.bwt_hip_host_alloc.restype = i32
    L.bwt_hip_host_free.argtypes = [vp]
    L.bwt_hip_host_free.restype = None
    L.bwt_hip_transport_free.argtypes = [vp]
    L.bwt_hip_transport_free.restype = None
    L.bwt_hip_slice_free.argtypes = [vp]
    L.bwt_hip_slice_free.restype = None
    L.bwt_hip_stream.argtypes = [vp]
    L.bwt_hip_stream.restype = vp
    _lib = L
    return L


def _chk(rc):
    if rc != 0:
        raise BwtHipError(rc, lib().bwt_hip_last_error().decode("utf-8", "replace"))


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class HostBuffer:
    """Page-locked host memory as a numpy uint8 array (`.a`); freed by close() or with the object."""

    def __init__(self, nbytes):
        self._p = C.c_void_p()
        _chk(lib().bwt_hip_host_alloc(nbytes, C.byref(self._p)))
        self.a = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self._p.value))

    def close(self):
        if self._p:
            self.a = None
            lib().bwt_hip_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pmbwt_multi(devices, T, bwt_rec_threshold, ssa_n_samples, build_gap_segment_sz=0):
    """bwt::pmbwt of the host text T over the listed GPUs (one thread and context per entry; an entry may repeat).
    Returns (L, s_rank, e_rank, sample ranks, sample positions, GPUs used)."""
    T = np.ascontiguousarray(T, dtype=np.uint8)
    n = len(T)
    dev = (C.c_int * len(devices))(*devices)
    L = np.empty(n, dtype=np.uint8)
    cap = min(n, max(1, ssa_n_samples))
    sr, sp = np.empty(cap, dtype=np.int64), np.empty(cap, dtype=np.int64)
    s_rank, e_rank, ns, used = C.c_int64(-1), C.c_int64(-1), C.c_int64(0), C.c_int(0)
    _chk(lib().bwt_hip_pmbwt_multi(dev, len(devices), _p(T, _u8p), n, bwt_rec_threshold, ssa_n_samples, build_gap_segment_sz,
                                   _p(L, _u8p), C.byref(s_rank), C.byref(e_rank), _p(sr, _i64p), _p(sp, _i64p), cap,
                                   C.byref(ns), C.byref(used)))
    k = min(ns.value, cap)
    return L, s_rank.value, e_rank.value, sr[:k], sp[:k], used.value


class Transport:
    """A bwt_hip_transport handle (one rank's end)."""

    def __init__(self, h):
        self._h = h

    def close(self):
        if self._h:
            lib().bwt_hip_transport_free(self._h)
            self._h = None

    @staticmethod
    def rccl_unique_id():
        buf = np.zeros(128, dtype=np.uint8)
        _chk(lib().bwt_hip_rccl_unique_id(_p(buf, _u8p)))
        return buf

    @staticmethod
    def rccl(ctx, unique_id, rank, world):
        h = C.c_void_p()
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        _chk(lib().bwt_hip_transport_rccl(ctx._h, _p(uid, _u8p), rank, world, C.byref(h)))
        return Transport(h)


class Slice:
    """This rank's slice of a block spread over the ranks of a multi-GPU build."""

    def __init__(self, ctx, h):
        self.ctx, self._h = ctx, h
        v = [C.c_int64() for _ in range(5)]
        _chk(lib().bwt_hip_slice_info(h, *[C.byref(x) for x in v]))
        self.first, self.size, self.s_rank, self.e_rank, self.n_samples = [x.value for x in v]

    @property
    def L(self):
        out = np.empty(self.size, dtype=np.uint8)
        _chk(lib().bwt_hip_slice_get_L(self.ctx._h, self._h, _p(out, _u8p)))
        return out

    def samples(self):
        r = np.empty(self.n_samples, dtype=np.int64)
        p = np.empty(self.n_samples, dtype=np.int64)
        _chk(lib().bwt_hip_slice_get_samples(self.ctx._h, self._h, _p(r, _i64p), _p(p, _i64p), self.n_samples))
        return r, p

    def read_into(self, L_out, rank_out, pos_out):
        """The slice's symbols and samples into the caller's arrays (for example page-locked memory)."""
        assert L_out.dtype == np.uint8 and len(L_out) >= self.size and len(rank_out) >= self.n_samples and len(pos_out) >= self.n_samples
        _chk(lib().bwt_hip_slice_get_L(self.ctx._h, self._h, _p(L_out, _u8p)))
        _chk(lib().bwt_hip_slice_get_samples(self.ctx._h, self._h, _p(rank_out, _i64p), _p(pos_out, _i64p), self.n_samples))

    def free(self):
        if self._h:
            lib().bwt_hip_slice_free(self._h)
            self._h = None


def dist_bounds(m, parts):
    """First left rank of each of the `parts` parts a merge with a left block of m suffixes is cut into; [parts] = m+1."""
    lo = np.zeros(parts + 1, dtype=np.int64)
    _chk(lib().bwt_hip_dist_bounds(m, parts, _p(lo, _i64p)))
    return lo


class Context:
    """One GPU context; mirrors the lifetime of the reference's `mallocator mem` (src/pmbwt.cc:317-332)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _chk(lib().bwt_hip_create(device, C.byref(self._h)))
        self.n = 0

    def close(self):
        if self._h:
            lib().bwt_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- text -------------------------------------------------------------
    def set_text(self, T):
        T = np.ascontiguousarray(T, dtype=np.uint8)
        _chk(lib().bwt_hip_set_text(self._h, _p(T, _u8p), len(T)))
        self.n = len(T)

    def set_text_device(self, ptr, n):
        _chk(lib().bwt_hip_set_text_device(self._h, C.c_void_p(ptr), n))
        self.n = n

    def gen_text(self, n, seed=42, sigma=4):
        _chk(lib().bwt_hip_gen_text(self._h, n, seed, sigma))
        self.n = n

    def get_text(self):
        T = np.empty(self.n, dtype=np.uint8)
        _chk(lib().bwt_hip_get_text(self._h, _p(T, _u8p)))
        return T

    def read_text(self, out):
        """The context's text into a caller's uint8 array (for example page-locked memory of HostBuffer)."""
        assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and len(out) >= self.n
        _chk(lib().bwt_hip_get_text(self._h, _p(out, _u8p)))

    # ---- pipeline ---------------------------------------------------------
    def _blk(self, h):
        return Block(self, h)

    def leaf(self, a, b, ssa_n_samples):
        h = C.c_void_p()
        _chk(lib().bwt_hip_leaf(self._h, a, b, ssa_n_samples, C.byref(h)))
        return self._blk(h)

    def merge(self, l, r, ssa_n_samples, build_gap_segment_sz=0):
        h = C.c_void_p()
        lh, rh = l._take(), r._take()
        _chk(lib().bwt_hip_merge(self._h, lh, rh, ssa_n_samples, build_gap_segment_sz, C.byref(h)))
        return self._blk(h)

    def rec(self, a, b, bwt_rec_threshold, ssa_n_samples, build_gap_segment_sz=0):
        h = C.c_void_p()
        _chk(lib().bwt_hip_rec(self._h, a, b, bwt_rec_threshold, ssa_n_samples, build_gap_segment_sz, C.byref(h)))
        return self._blk(h)

    def pmbwt(self, bwt_rec_threshold, ssa_n_samples, build_gap_segment_sz=0):
        h = C.c_void_p()
        _chk(lib().bwt_hip_pmbwt(self._h, bwt_rec_threshold, ssa_n_samples, build_gap_segment_sz, C.byref(h)))
        return self._blk(h)

    def dist_pmbwt(self, transport, bwt_rec_threshold, ssa_n_samples, build_gap_segment_sz=0):
        """bwt::pmbwt over the transport's ranks; every rank calls it.  Returns this rank's Slice."""
        h = C.c_void_p()
        _chk(lib().bwt_hip_dist_pmbwt(self._h, transport._h, bwt_rec_threshold, ssa_n_samples, build_gap_segment_sz, C.byref(h)))
        return Slice(self, h)

    def pmbwt_host(self, T, bwt_rec_threshold, ssa_n_samples, build_gap_segment_sz=0, L_out=None):
        """bwt::pmbwt from a host text (uint8 array) with the upload overlapped; L_out (uint8 array of len(T)) receives L."""
        assert T.dtype == np.uint8 and T.flags["C_CONTIGUOUS"]
        assert L_out is None or (L_out.dtype == np.uint8 and L_out.flags["C_CONTIGUOUS"] and len(L_out) >= len(T))
        h = C.c_void_p()
        _chk(lib().bwt_hip_pmbwt_host(self._h, _p(T, _u8p), len(T), bwt_rec_threshold, ssa_n_samples, build_gap_segment_sz,
                                      _p(L_out, _u8p), C.byref(h)))
        self.n = len(T)
        return self._blk(h)

    def upload_block(self, a, b, L, s_rank, e_rank, samp_rank, samp_pos):
        L = np.ascontiguousarray(L, dtype=np.uint8)
        sr = np.ascontiguousarray(samp_rank, dtype=np.int64)
        sp = np.ascontiguousarray(samp_pos, dtype=np.int64)
        assert len(L) == b - a
        h = C.c_void_p()
        _chk(lib().bwt_hip_block_upload(self._h, a, b, _p(L, _u8p), s_rank, e_rank, _p(sr, _i64p), _p(sp, _i64p),
                                        len(sr), C.byref(h)))
        return self._blk(h)

    def import_block_device(self, a, b, L_ptr, s_rank, e_rank, rank_ptr, pos_ptr, n_samples):
        """Block from device buffers (raw addresses): uint8 L[b-a], int64 ranks / positions."""
        h = C.c_void_p()
        _chk(lib().bwt_hip_block_import_device(self._h, a, b, C.c_void_p(L_ptr), s_rank, e_rank, C.c_void_p(rank_ptr),
                                               C.c_void_p(pos_ptr), n_samples, C.byref(h)))
        return self._blk(h)

    def leaf_sa(self, a, b):
        SA = np.empty(b - a, dtype=np.int64)
        _chk(lib().bwt_hip_leaf_sa(self._h, a, b, _p(SA, _i64p)))
        return SA

    def symbol_codes(self):
        """(compact, code_of, symbol_of): whether the text is held with its symbols renumbered 1 .. sigma (scattered
        alphabets), and the two tables (uint8[256]) between the text's symbols and the codes device buffers hold."""
        compact = C.c_int32(0)
        code_of, symbol_of = np.zeros(256, dtype=np.uint8), np.zeros(256, dtype=np.uint8)
        _chk(lib().bwt_hip_symbol_codes(self._h, C.byref(compact), _p(code_of, _u8p), _p(symbol_of, _u8p)))
        return bool(compact.value), code_of, symbol_of

    def dc_sample(self):
        """(period, positions, ranks) of the text's difference-cover sample: ranks[k] = rank of T[positions[k]:] among
        the sampled suffixes (built if it does not exist yet)."""
        n_out, period = C.c_int64(0), C.c_int32(0)
        _chk(lib().bwt_hip_dc_sample(self._h, 0, None, None, C.byref(n_out), C.byref(period)))
        pos = np.empty(n_out.value, dtype=np.int64)
        rank = np.empty(n_out.value, dtype=np.int64)
        _chk(lib().bwt_hip_dc_sample(self._h, n_out.value, _p(pos, _i64p), _p(rank, _i64p), None, None))
        return period.value, pos, rank

    def gap(self, l, r, build_gap_segment_sz=0, want_gap=True, want_g=True):
        gap = np.empty(l.b - l.a + 1, dtype=np.int64) if want_gap else None
        g = np.empty(r.b - r.a, dtype=np.int64) if want_g else None
        _chk(lib().bwt_hip_gap(self._h, l._h, r._h, build_gap_segment_sz, _p(gap, _i64p), _p(g, _i64p)))
        return gap, g

    # ---- one merge spread over several GPUs (raw device addresses) ----------
    def dist_leaf(self, a, b, ssa_n_samples, part, parts, L_ptr=None, cap=0, slots_ptr=None):
        """One GPU's share of a leaf cut by key range over `parts` GPUs: None if the leaf does not take the route, else
        dict(first_rank, count, s_rank, e_rank, bounds)."""
        fr, cnt, sr, er = C.c_int64(0), C.c_int64(0), C.c_int64(-1), C.c_int64(-1)
        bounds = np.zeros(parts + 1, dtype=np.int64)
        ok = C.c_int32(0)
        _chk(lib().bwt_hip_dist_leaf(self._h, a, b, ssa_n_samples, part, parts, L_ptr, cap, slots_ptr, C.byref(fr), C.byref(cnt),
                                     C.byref(sr), C.byref(er), _p(bounds, _i64p), C.byref(ok)))
        if not ok.value:
            return None
        return {"first_rank": fr.value, "count": cnt.value, "s_rank": sr.value, "e_rank": er.value, "bounds": bounds}

    def dist_chains(self, l, ra, rb, build_gap_segment_sz, g_ptr):
        _chk(lib().bwt_hip_dist_chains(self._h, l._h, ra, rb, build_gap_segment_sz, C.c_void_p(g_ptr)))

    def dist_partition(self, g_ptr, count, m, parts, out_ptr):
        counts = np.zeros(parts, dtype=np.int64)
        _chk(lib().bwt_hip_dist_partition(self._h, C.c_void_p(g_ptr), count, m, parts, C.c_void_p(out_ptr), _p(counts, _i64p)))
        return counts

    def dist_wire_packable(self):
        ok = C.c_int(0)
        _chk(lib().bwt_hip_dist_wire_packable(self._h, C.byref(ok)))
        return bool(ok.value)

    def dist_wire_pack(self, src_ptr, count, packed_ptr):
        _chk(lib().bwt_hip_dist_wire_pack(self._h, C.c_void_p(src_ptr), count, C.c_void_p(packed_ptr)))

    def dist_wire_unpack(self, packed_ptr, count, dst_ptr):
        _chk(lib().bwt_hip_dist_wire_unpack(self._h, C.c_void_p(packed_ptr), count, C.c_void_p(dst_ptr)))

    def dist_merge_slice(self, lL_ptr, mL, rL_ptr, rn, g_ptr, out_ptr, S_ptr):
        _chk(lib().bwt_hip_dist_merge_slice(self._h, C.c_void_p(lL_ptr), mL, C.c_void_p(rL_ptr), rn, C.c_void_p(g_ptr),
                                            C.c_void_p(out_ptr), C.c_void_p(S_ptr)))

    def dist_merge_slice_noalloc(self, lL_ptr, mL, rL_ptr, rn, g_ptr, out_ptr, S_ptr):
        _chk(lib().bwt_hip_dist_merge_slice_noalloc(self._h, C.c_void_p(lL_ptr), mL, C.c_void_p(rL_ptr), rn, C.c_void_p(g_ptr),
                                                    C.c_void_p(out_ptr), C.c_void_p(S_ptr)))

    def dist_samples_route(self, is_left, rank_ptr, pos_ptr, n, a, c, m, r, members, shift, queries_ptr, perm_ptr):
        vp = C.c_void_p
        counts = np.zeros(members, dtype=np.int64)
        _chk(lib().bwt_hip_dist_samples_route(self._h, int(is_left), vp(rank_ptr), vp(pos_ptr), n, a, c, m, r, members, shift,
                                              vp(queries_ptr), vp(perm_ptr), _p(counts, _i64p)))
        return counts

    def dist_layout(self, m, parts):
        """(first left rank of every part [parts + 1 entries], shift): shift 13 = the merge's g values travel grouped"""
        lo = np.zeros(parts + 1, dtype=np.int64)
        shift = C.c_int(0)
        _chk(lib().bwt_hip_dist_layout(self._h, m, parts, _p(lo, _i64p), C.byref(shift)))
        return lo, shift.value

    def dist_chains_grouped(self, l, ra, rb, build_gap_segment_sz, g_text_ptr, words_ptr, words_cap, starts_ptr):
        vp = C.c_void_p
        out = [C.c_int64(0) for _ in range(3)]
        _chk(lib().bwt_hip_dist_chains_grouped(self._h, l._h, ra, rb, build_gap_segment_sz, vp(g_text_ptr), vp(words_ptr), words_cap,
                                               vp(starts_ptr), *[C.byref(x) for x in out]))
        return [x.value for x in out]  # words, buckets, fillers

    def dist_merge_slice_grouped(self, lL_ptr, mL, rL_ptr, rn, words_ptr, word_off, starts_ptr, nbk, out_ptr, S_ptr):
        vp = C.c_void_p
        off = np.ascontiguousarray(word_off, dtype=np.int64)
        _chk(lib().bwt_hip_dist_merge_slice_grouped(self._h, vp(lL_ptr), mL, vp(rL_ptr), rn, vp(words_ptr), _p(off, _i64p), vp(starts_ptr),
                                                    len(off), nbk, vp(out_ptr), vp(S_ptr)))

    def dist_samples_answer(self, queries_ptr, n_left, n_right, S_ptr, i_lo, right_before, a, g_ptr, answers_ptr):
        vp = C.c_void_p
        _chk(lib().bwt_hip_dist_samples_answer(self._h, vp(queries_ptr), n_left, n_right, vp(S_ptr), i_lo, right_before, a,
                                               vp(g_ptr), vp(answers_ptr)))

    def dist_samples_place(self, is_left, rank_ptr, pos_ptr, n, answers_ptr, perm_ptr, a, c, first_index, n_all, ns_out,
                           o_rank_ptr, o_pos_ptr):
        vp = C.c_void_p
        kept = C.c_int64(0)
        _chk(lib().bwt_hip_dist_samples_place(self._h, int(is_left), vp(rank_ptr), vp(pos_ptr), n, vp(answers_ptr), vp(perm_ptr),
                                              a, c, first_index, n_all, ns_out, vp(o_rank_ptr), vp(o_pos_ptr), C.byref(kept)))
        return kept.value

    def set_timing(self, level):
        _chk(lib().bwt_hip_set_timing(self._h, level))

    def reset_timers(self):
        _chk(lib().bwt_hip_reset_timers(self._h))

    def timers(self):
        """{phase: (milliseconds, algorithmic work, calls)}"""
        out = {}
        for i in range(lib().bwt_hip_timer_count(self._h)):
            name, ms, work, calls = C.c_char_p(), C.c_double(), C.c_double(), C.c_int64()
            _chk(lib().bwt_hip_timer_get(self._h, i, C.byref(name), C.byref(ms), C.byref(work), C.byref(calls)))
            out[name.value.decode()] = (ms.value, work.value, calls.value)
        return out

    def mem_stats(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _chk(lib().bwt_hip_mem_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"live": a.value, "peak": b.value, "reserved": c.value}

    def trim(self):
        _chk(lib().bwt_hip_trim(self._h))

    def sync(self):
        _chk(lib().bwt_hip_sync(self._h))

    @property
    def stream(self):
        return lib().bwt_hip_stream(self._h)

    # ---- primitives for unit tests ------------------------------------------
    def test_sort(self, keys, vals=None, begin_bit=0, end_bit=None):
        keys = np.ascontiguousarray(keys).copy()
        vals = None if vals is None else np.ascontiguousarray(vals, dtype=np.uint32).copy()
        if keys.dtype == np.uint64:
            _chk(lib().bwt_hip_test_sort_u64(self._h, _p(keys, _u64p), _p(vals, _u32p), len(keys), begin_bit,
                                             64 if end_bit is None else end_bit))
        else:
            assert keys.dtype == np.uint32
            _chk(lib().bwt_hip_test_sort_u32(self._h, _p(keys, _u32p), _p(vals, _u32p), len(keys), begin_bit,
                                             32 if end_bit is None else end_bit))
        return keys, vals

    def test_scan(self, data, op):
        d = np.ascontiguousarray(data, dtype=np.uint32).copy()
        _chk(lib().bwt_hip_test_scan_u32(self._h, _p(d, _u32p), len(d), op))
        return d


class Block:
    """A partial BWT of T[a:b) in HBM: the fields of the reference's `struct bwt` (include/bwt.h:135-145)."""

    def __init__(self, ctx, h):
        self.ctx, self._h = ctx, h
        a, b, s, e, n = (C.c_int64() for _ in range(5))
        ec = C.c_int()
        _chk(lib().bwt_hip_block_info(h, C.byref(a), C.byref(b), C.byref(s), C.byref(e), C.byref(ec), C.byref(n)))
        self.a, self.b, self.s_rank, self.e_rank, self.e_char, self.n_samples = (
            a.value, b.value, s.value, e.value, ec.value, n.value)

    def _take(self):
        h, self._h = self._h, None
        return h

    def free(self):
        if self._h and self.ctx._h:
            lib().bwt_hip_block_free(self.ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def L(self):
        """L[a:b) as a numpy array (index 0 is rank a)."""
        out = np.empty(self.b - self.a, dtype=np.uint8)
        _chk(lib().bwt_hip_block_get_L(self.ctx._h, self._h, _p(out, _u8p)))
        return out

    def read_L(self, out):
        """L[a:b) into a caller's uint8 array (for example page-locked memory of HostBuffer)."""
        assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and len(out) >= self.b - self.a
        _chk(lib().bwt_hip_block_get_L(self.ctx._h, self._h, _p(out, _u8p)))

    @property
    def L_device(self):
        p = C.c_void_p()
        _chk(lib().bwt_hip_block_L_device(self._h, C.byref(p)))
        return p.value

    def export_device(self, L_ptr, rank_ptr, pos_ptr):
        """Copies L and the samples into device buffers given by raw addresses."""
        _chk(lib().bwt_hip_block_export_device(self.ctx._h, self._h, C.c_void_p(L_ptr), C.c_void_p(rank_ptr),
                                               C.c_void_p(pos_ptr)))

    def samples(self):
        r = np.empty(self.n_samples, dtype=np.int64)
        p = np.empty(self.n_samples, dtype=np.int64)
        _chk(lib().bwt_hip_block_get_samples(self.ctx._h, self._h, _p(r, _i64p), _p(p, _i64p), self.n_samples))
        return r, p

    def read_samples(self, rank_out, pos_out):
        """The samples into a caller's int64 arrays; returns how many were written."""
        k = min(self.n_samples, len(rank_out), len(pos_out))
        assert rank_out.dtype == np.int64 and pos_out.dtype == np.int64
        _chk(lib().bwt_hip_block_get_samples(self.ctx._h, self._h, _p(rank_out, _i64p), _p(pos_out, _i64p), k))
        return k

    def init_extra(self):
        _chk(lib().bwt_hip_block_init_extra(self.ctx._h, self._h))

    def ibwt(self):
        """T[a:b) decoded from L on the device (bwt::ibwt)."""
        out = np.empty(self.b - self.a, dtype=np.uint8)
        _chk(lib().bwt_hip_block_ibwt(self.ctx._h, self._h, _p(out, _u8p), None))
        return out

    def ibwt_mismatches(self):
        """Positions where the decoded block differs from the context's text (0 = round trip ok); nothing is downloaded."""
        bad = C.c_int64(-1)
        _chk(lib().bwt_hip_block_ibwt(self.ctx._h, self._h, None, C.byref(bad)))
        return bad.value

    def sa(self, rank):
        """SA[r] for absolute ranks r (bwt::sa)."""
        rank = np.ascontiguousarray(rank, dtype=np.int64)
        out = np.empty(len(rank), dtype=np.int64)
        _chk(lib().bwt_hip_block_sa(self.ctx._h, self._h, len(rank), _p(rank, _i64p), _p(out, _i64p)))
        return out

    def C(self, alpha_max):
        out = np.empty(alpha_max + 1, dtype=np.int64)
        _chk(lib().bwt_hip_block_get_C(self.ctx._h, self._h, alpha_max, _p(out, _i64p)))
        return out

    def rank(self, sym, pos):
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        pos = np.ascontiguousarray(pos, dtype=np.int64)
        out = np.empty(len(sym), dtype=np.int64)
        _chk(lib().bwt_hip_block_rank(self.ctx._h, self._h, len(sym), _p(sym, _u8p), _p(pos, _i64p), _p(out, _i64p)))
        return out

    def LF(self, sym, rank):
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        rank = np.ascontiguousarray(rank, dtype=np.int64)
        out = np.empty(len(sym), dtype=np.int64)
        _chk(lib().bwt_hip_block_LF(self.ctx._h, self._h, len(sym), _p(sym, _u8p), _p(rank, _i64p), _p(out, _i64p)))
        return out
