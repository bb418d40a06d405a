"""TEST INFRASTRUCTURE: pbc_amd/csrc/*.cuh compiled for the host (see hostsim.cpp)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libhostsim.so")
_CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build():
    srcs = [os.path.join(_HERE, f) for f in ("hostsim.cpp", "hostsim_shim.h")]
    csrc = os.path.join(_HERE, "..", "..", "pbc_amd", "csrc")
    srcs += [os.path.join(csrc, f) for f in os.listdir(csrc)]
    if (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call([_CLANG, "-O1", "-Wno-psabi", "-std=c++17", "-fPIC", "-shared", "-I", _HERE,
                               "-o", _LIB, os.path.join(_HERE, "hostsim.cpp")])
    return _LIB


class HostSim:
    def __init__(self, param_text):
        L = ctypes.CDLL(build())
        L.hostsim_init.restype = ctypes.c_void_p
        L.hostsim_init.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.hostsim_error.restype = ctypes.c_char_p
        vp = ctypes.c_void_p
        L.hostsim_prod_pairing.argtypes = [vp, vp, vp, vp, ctypes.c_size_t, ctypes.c_int]
        L.hostsim_fq_op.argtypes = [vp, ctypes.c_int, vp, vp, vp, ctypes.c_size_t]
        L.hostsim_lens.argtypes = [vp] + [ctypes.POINTER(ctypes.c_int)] * 3
        self.L = L
        b = param_text.encode() if isinstance(param_text, str) else param_text
        self.h = L.hostsim_init(b, len(b))
        if not self.h:
            raise ValueError(L.hostsim_error().decode())
        a, b2, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L.hostsim_lens(self.h, ctypes.byref(a), ctypes.byref(b2), ctypes.byref(c))
        self.len1, self.len2, self.lenT = a.value, b2.value, c.value

    def macs(self, reset=False):
        """multiply-adds (32 x 32 + 64 bit) the kernel source has executed since the last reset"""
        self.L.hostsim_macs_read.restype = ctypes.c_uint64
        self.L.hostsim_macs_read.argtypes = [ctypes.c_int]
        return int(self.L.hostsim_macs_read(1 if reset else 0))

    def prod_pairing(self, g1, g2, k=1):
        g1 = np.ascontiguousarray(g1, np.uint8)
        g2 = np.ascontiguousarray(g2, np.uint8)
        n = g1.size // (self.len1 * k)
        out = np.empty((n, self.lenT), np.uint8)
        self.L.hostsim_prod_pairing(self.h, out.ctypes.data, g1.ctypes.data, g2.ctypes.data, n, k)
        return out

    def pairing_wave(self, g1, g2):
        """element_pairing through the one-pairing-per-wavefront routine (pairing_aw.cuh), type a"""
        g1 = np.ascontiguousarray(g1, np.uint8)
        g2 = np.ascontiguousarray(g2, np.uint8)
        n = g1.size // self.len1
        out = np.empty((n, self.lenT), np.uint8)
        self.L.hostsim_pairing_wave.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t]
        if self.L.hostsim_pairing_wave(self.h, out.ctypes.data, g1.ctypes.data, g2.ctypes.data, n):
            raise RuntimeError("no wave routine for this pairing")
        return out

    def pp_wave(self, g1, g2):
        """pairing_pp_apply on the wave routine (pairing_aw.cuh pp_apply_wave): one first argument, n second arguments"""
        g1 = np.ascontiguousarray(g1, np.uint8)
        g2 = np.ascontiguousarray(g2, np.uint8)
        n = g2.size // self.len2
        out = np.empty((n, self.lenT), np.uint8)
        self.L.hostsim_pp_wave.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t]
        if self.L.hostsim_pp_wave(self.h, out.ctypes.data, g1.ctypes.data, g2.ctypes.data, n):
            raise RuntimeError("no wave routine for this pairing")
        return out

    def prod_wave(self, g1, g2, k):
        """element_prod_pairing on the wave routines: a Miller record per term, one finish per product"""
        g1 = np.ascontiguousarray(g1, np.uint8)
        g2 = np.ascontiguousarray(g2, np.uint8)
        n = g1.size // self.len1 // k
        out = np.empty((n, self.lenT), np.uint8)
        self.L.hostsim_prod_wave.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_int]
        if self.L.hostsim_prod_wave(self.h, out.ctypes.data, g1.ctypes.data, g2.ctypes.data, n, k):
            raise RuntimeError("no wave routine for this pairing")
        return out

    def pp(self, g1, g2):
        g1 = np.ascontiguousarray(g1, np.uint8)
        g2 = np.ascontiguousarray(g2, np.uint8)
        n = g2.size // self.len2
        out = np.empty((n, self.lenT), np.uint8)
        self.L.hostsim_pp.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t]
        if self.L.hostsim_pp(self.h, out.ctypes.data, g1.ctypes.data, g2.ctypes.data, n):
            raise RuntimeError("hostsim_pp failed")
        return out

    def group(self, what, a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        la = self.len1 if what == 0 else self.lenT
        n = a.size // la
        out = np.empty((n, la), np.uint8)
        self.L.hostsim_group.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_size_t]
        self.L.hostsim_group(self.h, what, out.ctypes.data, a.ctypes.data, b.ctypes.data, n)
        return out

    def finalpow(self, a):
        return self.group(3, a, a)

    def group_mode(self, slow):
        """True: only the complete word-form routines (what the library runs for the lanes its fast routines report)"""
        self.L.hostsim_group_mode.argtypes = [ctypes.c_int]
        self.L.hostsim_group_mode(1 if slow else 0)

    def fallbacks(self, reset=True):
        """lanes the fast group routines have reported since the last reset"""
        self.L.hostsim_group_fallbacks.restype = ctypes.c_uint64
        self.L.hostsim_group_fallbacks.argtypes = [ctypes.c_int]
        return int(self.L.hostsim_group_fallbacks(1 if reset else 0))

    def element_pp(self, group, base, zr, zlen):
        """element_pp_init(base) + element_pp_pow_zn over the scalars zr (group 1, 2: points; 3: GT)"""
        base = np.ascontiguousarray(base, np.uint8).reshape(-1)
        zr = np.ascontiguousarray(zr, np.uint8)
        n = zr.size // zlen
        out = np.empty((n, base.size), np.uint8)
        self.L.hostsim_element_pp.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_size_t]
        self.L.hostsim_element_pp(self.h, group, out.ctypes.data, base.ctypes.data, zr.ctypes.data, n)
        return out

    def compress(self, direction, recs):
        recs = np.ascontiguousarray(recs, np.uint8)
        lp, lc = self.len1, self.len1 // 2 + (1 if direction < 2 else 0)
        li, lo = (lp, lc) if direction % 2 == 0 else (lc, lp)
        n = recs.size // li
        out = np.empty((n, lo), np.uint8)
        self.L.hostsim_compress.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        self.L.hostsim_compress(self.h, direction, out.ctypes.data, recs.ctypes.data, n)
        return out

    def g2_mul(self, a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        n = a.size // self.len2
        out = np.empty((n, self.len2), np.uint8)
        self.L.hostsim_g2_mul.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t]
        self.L.hostsim_g2_mul(self.h, out.ctypes.data, a.ctypes.data, b.ctypes.data, n)
        return out

    def g2_points(self, what, recs, hlen=0):
        """twists: what 0 element_from_hash (recs: digests of hlen bytes), 1 compress, 2 decompress, 3 / 4 to / from x-only"""
        recs = np.ascontiguousarray(recs, np.uint8)
        lp, lc, lx = self.len2, self.len2 // 2 + 1, self.len2 // 2
        li, lo = {0: (hlen, lp), 1: (lp, lc), 2: (lc, lp), 3: (lp, lx), 4: (lx, lp)}[what]
        n = recs.size // li
        out = np.empty((n, lo), np.uint8)
        self.L.hostsim_g2_points.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        if self.L.hostsim_g2_points(self.h, what, out.ctypes.data, recs.ctypes.data, hlen, n):
            raise RuntimeError("hostsim_g2_points failed")
        return out

    def from_hash(self, data, hlen):
        data = np.ascontiguousarray(data, np.uint8)
        n = data.size // hlen
        out = np.empty((n, self.len1), np.uint8)
        self.L.hostsim_from_hash.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        if self.L.hostsim_from_hash(self.h, out.ctypes.data, data.ctypes.data, hlen, n):
            raise RuntimeError("hostsim_from_hash failed")
        return out

    def stage(self, stage, g1=None, g2=None, n=0, out_len=4096):
        out = np.zeros(max(out_len, n * self.lenT), np.uint8)
        self.L.hostsim_stage.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        g1p = np.ascontiguousarray(g1, np.uint8).ctypes.data if g1 is not None else None
        g2p = np.ascontiguousarray(g2, np.uint8).ctypes.data if g2 is not None else None
        rc = self.L.hostsim_stage(self.h, stage, out.ctypes.data, out.size, g1p, g2p, n)
        return out, rc

    def fq_op(self, op, a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        L = self.len1 // 2
        n = a.size // L
        out = np.empty((n, L), np.uint8)
        self.L.hostsim_fq_op(self.h, op, out.ctypes.data, a.ctypes.data, b.ctypes.data, n)
        return out

    # ---- round 5: group law, multi-exponentiations, Z_r (group_more.cuh lane bodies) ----
    def len_zr(self):
        self.L.hostsim_len_zr.argtypes = [ctypes.c_void_p]
        return self.L.hostsim_len_zr(self.h)

    def affine_op(self, op, group, a, b=None):
        a = np.ascontiguousarray(a, np.uint8)
        b = None if b is None else np.ascontiguousarray(b, np.uint8)
        out = np.zeros_like(a)
        self.L.hostsim_affine_op.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_size_t]
        rc = self.L.hostsim_affine_op(self.h, op, group, out.ctypes.data, a.ctypes.data, None if b is None else b.ctypes.data, len(a))
        assert rc == 0
        return out

    def multi(self, group, bases, scalars):
        k = len(bases)
        bases = [np.ascontiguousarray(x, np.uint8) for x in bases]
        scalars = [np.ascontiguousarray(x, np.uint8) for x in scalars]
        out = np.zeros_like(bases[0])
        self.L.hostsim_multi.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 7 + [ctypes.c_size_t]
        ptr = []
        for j in range(3):
            ptr += [bases[j].ctypes.data if j < k else None, scalars[j].ctypes.data if j < k else None]
        rc = self.L.hostsim_multi(self.h, group, k, out.ctypes.data, *ptr, len(out))
        assert rc == 0
        return out

    def fx(self, sqr, op, words):
        """one fused product (fp.cuh fp_mulx / fp_sqrx / fp_sopx for sqr = 2): words = 7 lists of N little-endian words
        (a, a2, b, b2, c1, c2, d2)"""
        ops = np.ascontiguousarray(np.array(words, np.uint32).reshape(-1))
        out = np.zeros(len(ops) // 7, np.uint32)
        self.L.hostsim_fx.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        n = self.L.hostsim_fx(self.h, sqr, op, ops.ctypes.data, out.ctypes.data)
        assert n == len(out)
        return out

    def e_rxs(self):
        self.L.hostsim_e_rxs.argtypes = [ctypes.c_void_p]
        return self.L.hostsim_e_rxs(self.h)

    def zr_op(self, op, a, b=None, hlen=0):
        a = np.ascontiguousarray(a, np.uint8)
        b = None if b is None else np.ascontiguousarray(b, np.uint8)
        out = np.zeros((len(a), self.len_zr()), np.uint8)
        self.L.hostsim_zr_op.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_size_t]
        rc = self.L.hostsim_zr_op(self.h, op, out.ctypes.data, a.ctypes.data, None if b is None else b.ctypes.data, hlen, len(a))
        assert rc == 0
        return out
