"""ctypes front ends of the two CPU checkers.

TEST INFRASTRUCTURE ONLY.  ``Oracle`` wraps ``liboracle.so`` (the C restatement
in ``pmbwt_oracle.c``); ``Reference`` wraps ``_ref/libbwt_ref.so`` (the real
reference compiled from its own headers by ``oracle/Makefile``).  Only the
tests, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
import this package; the product path never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libbwt_ref.so")

_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)


def build(quiet=True):
    """(Re)build the checkers with oracle/Makefile (the reference one only when its tree is present)."""
    subprocess.run(["make", "-C", _HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _p8(a):
    return a.ctypes.data_as(_u8p)


def _p64(a):
    return a.ctypes.data_as(_i64p) if a is not None else None


def gen_text(n, seed=42, sigma=4):
    """Synthetic text: splitmix64 stream from `seed`, T[i] = 1 + x % sigma, T[n-1] = 0 (numpy, vectorised)."""
    i = np.arange(1, n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    T = np.empty(n, dtype=np.uint8)
    T[: n - 1] = (np.uint64(1) + z % np.uint64(sigma)).astype(np.uint8)
    T[n - 1] = 0
    return T


def gen_repetitive(unit_len, copies, rate, seed=43, sigma=4):
    """Long-LCP text: `copies` of one random unit, each with i.i.d. point substitutions at `rate`; 0-terminated."""
    base = gen_text(unit_len + 1, seed, sigma)[:unit_len]
    rng = np.random.RandomState(seed + 1)
    parts = []
    for _ in range(copies):
        c = base.copy()
        k = rng.binomial(unit_len, rate)
        if k:
            where = rng.randint(0, unit_len, size=k)
            c[where] = 1 + (c[where] - 1 + rng.randint(1, sigma, size=k)) % sigma
        parts.append(c)
    return np.concatenate(parts + [np.zeros(1, dtype=np.uint8)])


def fnv1a64(a):
    h = 0xCBF29CE484222325
    lib = Oracle.lib()
    return int(lib.orc_fnv1a64(_p8(np.ascontiguousarray(a, dtype=np.uint8)), len(a)))


class _Opt(C.Structure):
    _fields_ = [("mem_budget", C.c_double), ("alpha_min", C.c_int), ("alpha_max", C.c_int),
                ("ssa_n_samples", C.c_int64), ("bwt_rec_threshold", C.c_int64),
                ("build_gap_segment_sz", C.c_int64), ("gap_sum_gran", C.c_int64)]


class Oracle:
    """The C restatement.  Blocks are opaque handles; arrays are numpy."""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            if not os.path.exists(ORACLE_SO):
                build()
            L = C.CDLL(ORACLE_SO)
            L.orc_fnv1a64.restype = C.c_uint64
            L.orc_fnv1a64.argtypes = [_u8p, C.c_int64]
            L.orc_set_defaults.argtypes = [C.POINTER(_Opt), C.c_int64]
            L.orc_sa_range.argtypes = [_u8p, C.c_int64, C.c_int64, C.c_int64, _i64p]
            L.orc_leaf.restype = C.c_void_p
            L.orc_leaf.argtypes = [_u8p, C.c_int64, C.c_int64, C.c_int64, _u8p, C.POINTER(_Opt), _i64p]
            L.orc_merge.restype = C.c_void_p
            L.orc_merge.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_Opt), _i64p, _i64p]
            L.orc_pmbwt.restype = C.c_void_p
            L.orc_pmbwt.argtypes = [_u8p, C.c_int64, _u8p, C.POINTER(_Opt)]
            L.orc_block_free.argtypes = [C.c_void_p]
            for f in ("a", "b", "s_rank", "e_rank", "n_samples"):
                getattr(L, "orc_block_" + f).restype = C.c_int64
                getattr(L, "orc_block_" + f).argtypes = [C.c_void_p]
            L.orc_block_e_char.argtypes = [C.c_void_p]
            L.orc_block_samples.restype = C.c_int64
            L.orc_block_samples.argtypes = [C.c_void_p, _i64p, _i64p, C.c_int64]
            L.orc_block_init_extra.argtypes = [C.c_void_p]
            L.orc_block_C.argtypes = [C.c_void_p, _i64p]
            L.orc_block_rank.restype = C.c_int64
            L.orc_block_rank.argtypes = [C.c_void_p, C.c_int, C.c_int64]
            L.orc_block_LF.restype = C.c_int64
            L.orc_block_LF.argtypes = [C.c_void_p, C.c_int, C.c_int64]
            L.orc_block_sa.restype = C.c_int64
            L.orc_block_sa.argtypes = [C.c_void_p, C.c_int64]
            L.orc_block_bin_search.restype = C.c_int64
            L.orc_block_bin_search.argtypes = [C.c_void_p, C.c_int64]
            L.orc_block_ibwt.argtypes = [C.c_void_p, _u8p]
            L.orc_bwt_direct.argtypes = [_u8p, C.c_int64, _u8p, _i64p]
            L.orc_gen_text.argtypes = [_u8p, C.c_int64, C.c_uint64, C.c_int]
            L.orc_int_log2.argtypes = [C.c_int]
            cls._lib = L
        return cls._lib

    @classmethod
    def opt(cls, n, mem_budget=0.0, alpha_min=0, alpha_max=0, bwt_rec_threshold=0, ssa_n_samples=0,
            build_gap_segment_sz=0):
        """set_defaults, then the caller's explicit threshold / sample count on top (as the reference's tests do)."""
        o = _Opt(mem_budget, alpha_min, alpha_max, 0, 0, build_gap_segment_sz, 0)
        cls.lib().orc_set_defaults(C.byref(o), n)
        if bwt_rec_threshold:
            o.bwt_rec_threshold = bwt_rec_threshold
        if ssa_n_samples:
            o.ssa_n_samples = ssa_n_samples
        return o

    @classmethod
    def gen_text(cls, n, seed=42, sigma=4):
        T = np.empty(n, dtype=np.uint8)
        cls.lib().orc_gen_text(_p8(T), n, seed, sigma)
        return T

    @classmethod
    def sa_range(cls, T, a, b):
        SA = np.empty(b - a, dtype=np.int64)
        cls.lib().orc_sa_range(_p8(T), len(T), a, b, _p64(SA))
        return SA

    @classmethod
    def bwt_direct(cls, T, want_sa=False):
        L = np.empty(len(T), dtype=np.uint8)
        SA = np.empty(len(T), dtype=np.int64) if want_sa else None
        cls.lib().orc_bwt_direct(_p8(T), len(T), _p8(L), _p64(SA))
        return (L, SA) if want_sa else L

    @classmethod
    def leaf(cls, T, a, b, L, opt, want_sa=False):
        SA = np.empty(b - a, dtype=np.int64) if want_sa else None
        h = cls.lib().orc_leaf(_p8(T), len(T), a, b, _p8(L), C.byref(opt), _p64(SA))
        blk = OracleBlock(h, T, L, opt)
        return (blk, SA) if want_sa else blk

    @classmethod
    def merge(cls, l, r, opt, want_taps=False):
        gap = np.empty(l.b - l.a + 1, dtype=np.int64) if want_taps else None
        g = np.empty(r.b - r.a, dtype=np.int64) if want_taps else None
        h = cls.lib().orc_merge(l._take(), r._take(), C.byref(opt), _p64(gap), _p64(g))
        m = OracleBlock(h, l.T, l.L, opt)
        return (m, gap, g) if want_taps else m

    @classmethod
    def pmbwt(cls, T, opt):
        L = np.zeros(len(T), dtype=np.uint8)
        h = cls.lib().orc_pmbwt(_p8(T), len(T), _p8(L), C.byref(opt))
        return OracleBlock(h, T, L, opt)


class OracleBlock:
    def __init__(self, h, T, L, opt):
        self._h, self.T, self.L, self.opt = h, T, L, opt
        self._lib = Oracle.lib()

    def _take(self):
        h, self._h = self._h, None
        return h

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.orc_block_free(self._h)
            self._h = None

    a = property(lambda s: s._lib.orc_block_a(s._h))
    b = property(lambda s: s._lib.orc_block_b(s._h))
    s_rank = property(lambda s: s._lib.orc_block_s_rank(s._h))
    e_rank = property(lambda s: s._lib.orc_block_e_rank(s._h))
    e_char = property(lambda s: s._lib.orc_block_e_char(s._h))
    n_samples = property(lambda s: s._lib.orc_block_n_samples(s._h))

    def samples(self):
        """(rank, pos) arrays ordered by pos."""
        cap = int(self.n_samples) + 8
        r = np.empty(cap, dtype=np.int64)
        p = np.empty(cap, dtype=np.int64)
        k = self._lib.orc_block_samples(self._h, _p64(r), _p64(p), cap)
        assert k <= cap
        return r[:k].copy(), p[:k].copy()

    def init_extra(self):
        self._lib.orc_block_init_extra(self._h)

    def C(self):
        out = np.empty(self.opt.alpha_max + 1, dtype=np.int64)
        self._lib.orc_block_C(self._h, _p64(out))
        return out

    def rank(self, c, i):
        return self._lib.orc_block_rank(self._h, c, i)

    def LF(self, c, i):
        return self._lib.orc_block_LF(self._h, c, i)

    def sa(self, r):
        return self._lib.orc_block_sa(self._h, r)

    def bin_search(self, x):
        return self._lib.orc_block_bin_search(self._h, x)

    def ibwt(self):
        I = np.zeros(len(self.T), dtype=np.uint8)
        self._lib.orc_block_ibwt(self._h, _p8(I))
        return I


class Reference:
    """The real reference (serial build), when oracle/_ref/libbwt_ref.so exists."""
    _lib = None

    @classmethod
    def available(cls):
        return os.path.exists(REF_SO)

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(REF_SO)
            lng = C.c_long
            lp = C.POINTER(C.c_long)
            L.ref_set_defaults.argtypes = [C.c_double, lng, lp, lp]
            L.ref_sa_range.argtypes = [_u8p, lng, lng, lng, lp]
            L.ref_leaf.restype = C.c_void_p
            L.ref_leaf.argtypes = [_u8p, lng, lng, lng, _u8p, C.c_int, C.c_int, lng, lng]
            L.ref_gap.argtypes = [C.c_void_p, C.c_void_p, lp]
            L.ref_merge.restype = C.c_void_p
            L.ref_merge.argtypes = [C.c_void_p, C.c_void_p]
            L.ref_pmbwt.restype = C.c_void_p
            L.ref_pmbwt.argtypes = [_u8p, lng, _u8p, C.c_double, C.c_int, C.c_int, lng, lng, lng]
            for f in ("threshold", "ssa_n_samples", "a", "b", "s_rank", "e_rank", "n_samples"):
                getattr(L, "ref_" + f).restype = lng
                getattr(L, "ref_" + f).argtypes = [C.c_void_p]
            L.ref_e_char.argtypes = [C.c_void_p]
            L.ref_samples.restype = lng
            L.ref_samples.argtypes = [C.c_void_p, lp, lp, lng]
            L.ref_init_extra.argtypes = [C.c_void_p, _u8p]
            L.ref_C.argtypes = [C.c_void_p, lp]
            for f in ("rank", "LF"):
                getattr(L, "ref_" + f).restype = lng
                getattr(L, "ref_" + f).argtypes = [C.c_void_p, C.c_int, lng]
            for f in ("sa", "bin_search"):
                getattr(L, "ref_" + f).restype = lng
                getattr(L, "ref_" + f).argtypes = [C.c_void_p, lng]
            L.ref_ibwt.argtypes = [C.c_void_p, _u8p]
            L.ref_free.argtypes = [C.c_void_p]
            cls._lib = L
        return cls._lib

    @classmethod
    def set_defaults(cls, n, mem_budget=0.0):
        th, ns = C.c_long(), C.c_long()
        cls.lib().ref_set_defaults(mem_budget, n, C.byref(th), C.byref(ns))
        return th.value, ns.value

    @classmethod
    def sa_range(cls, T, a, b):
        SA = np.empty(b - a, dtype=np.int64)
        cls.lib().ref_sa_range(_p8(T), len(T), a, b, SA.ctypes.data_as(C.POINTER(C.c_long)))
        return SA

    @classmethod
    def leaf(cls, T, a, b, L, alpha_min=0, alpha_max=0, ssa_n_samples=0, seg=0):
        h = cls.lib().ref_leaf(_p8(T), len(T), a, b, _p8(L), alpha_min, alpha_max, ssa_n_samples, seg)
        return RefBlock(h, T, L, alpha_max or 255)

    @classmethod
    def gap(cls, l, r):
        out = np.empty(l.b - l.a + 1, dtype=np.int64)
        cls.lib().ref_gap(l._h, r._h, out.ctypes.data_as(C.POINTER(C.c_long)))
        return out

    @classmethod
    def merge(cls, l, r):
        h = cls.lib().ref_merge(l._take(), r._take())
        return RefBlock(h, l.T, l.L, l.alpha_max)

    @classmethod
    def pmbwt(cls, T, mem_budget=0.0, alpha_min=0, alpha_max=0, bwt_rec_threshold=0, ssa_n_samples=0, seg=0):
        L = np.zeros(len(T), dtype=np.uint8)
        h = cls.lib().ref_pmbwt(_p8(T), len(T), _p8(L), mem_budget, alpha_min, alpha_max,
                                bwt_rec_threshold, ssa_n_samples, seg)
        return RefBlock(h, T, L, alpha_max or 255)


class RefBlock:
    def __init__(self, h, T, L, alpha_max):
        self._h, self.T, self.L, self.alpha_max = h, T, L, alpha_max
        self._lib = Reference.lib()
        self._W = None

    def _take(self):
        h, self._h = self._h, None
        return h

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ref_free(self._h)
            self._h = None

    a = property(lambda s: s._lib.ref_a(s._h))
    b = property(lambda s: s._lib.ref_b(s._h))
    s_rank = property(lambda s: s._lib.ref_s_rank(s._h))
    e_rank = property(lambda s: s._lib.ref_e_rank(s._h))
    e_char = property(lambda s: s._lib.ref_e_char(s._h))
    n_samples = property(lambda s: s._lib.ref_n_samples(s._h))
    threshold = property(lambda s: s._lib.ref_threshold(s._h))
    ssa_n_samples = property(lambda s: s._lib.ref_ssa_n_samples(s._h))

    def samples(self):
        cap = int(self.n_samples) + 8
        r = np.empty(cap, dtype=np.int64)
        p = np.empty(cap, dtype=np.int64)
        lp = C.POINTER(C.c_long)
        k = self._lib.ref_samples(self._h, r.ctypes.data_as(lp), p.ctypes.data_as(lp), cap)
        assert k <= cap
        return r[:k].copy(), p[:k].copy()

    def init_extra(self):
        self._W = np.zeros(len(self.T), dtype=np.uint8)
        self._lib.ref_init_extra(self._h, _p8(self._W))

    def C(self):
        out = np.empty(self.alpha_max + 1, dtype=np.int64)
        self._lib.ref_C(self._h, out.ctypes.data_as(C.POINTER(C.c_long)))
        return out

    def rank(self, c, i):
        return self._lib.ref_rank(self._h, c, i)

    def LF(self, c, i):
        return self._lib.ref_LF(self._h, c, i)

    def sa(self, r):
        return self._lib.ref_sa(self._h, r)

    def bin_search(self, x):
        return self._lib.ref_bin_search(self._h, x)

    def ibwt(self):
        I = np.zeros(len(self.T), dtype=np.uint8)
        self._lib.ref_ibwt(self._h, _p8(I))
        return I
