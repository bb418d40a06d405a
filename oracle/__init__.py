"""oracle -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so (the plain-C CPU restatement of the reference's
pairing path, oracle/pbc_oracle.c) plus helpers for the golden vector files written by
oracle/_ref/ref_tool (the unmodified reference compiled by oracle/Makefile).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  pbc_amd/ (the product) never does.
"""
import ctypes
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
REF_TOOL = os.path.join(_HERE, "_ref", "ref_tool")
GLUE_TEST = os.path.join(_HERE, "_ref", "glue_test")
PRELOAD_LIB = os.path.join(_HERE, "_ref", "libpbc_hip_preload.so")
BLS_EXAMPLE = os.path.join(_HERE, "_ref", "bls_example")                # example/bls.c, unchanged, dynamic libpbc
BLS_EXAMPLE_WRAPPED = os.path.join(_HERE, "_ref", "bls_example_wrapped")  # the same source linked with -Wl,--wrap


def build(force=False):
    """Compile the C restatement (and, where /root/reference exists, oracle/_ref)."""
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "pbc_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if os.path.isdir("/root/reference/arith"):
        if force or not os.path.exists(REF_TOOL):
            subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "ref"])
        subprocess.check_call(["make", "-s", "-C", _HERE, "glue", "preload"])      # reference + integration glue, drop-in demo


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        vp, cp, sz, ci = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int
        L.oracle_pairing_init.argtypes = [ctypes.POINTER(vp), cp, sz]
        L.oracle_pairing_clear.argtypes = [vp]
        for f in ("oracle_type", "oracle_len_G1", "oracle_len_G2", "oracle_len_GT"):
            getattr(L, f).argtypes = [vp]
        L.oracle_pairing_batch.argtypes = [vp, vp, vp, vp, sz]
        L.oracle_prod_pairing_batch.argtypes = [vp, vp, vp, vp, sz, ci]
        L.oracle_fq_op.argtypes = [vp, ci, vp, vp, vp, sz]
        L.oracle_gt_mul.argtypes = [vp, vp, vp, vp, sz]
        L.oracle_gt_pow.argtypes = [vp, vp, vp, sz, vp, sz]
        L.oracle_g_mul.argtypes = [vp, ci, vp, vp, sz, vp, sz]
        L.oracle_from_hash.argtypes = [vp, vp, ci, vp, sz]
        L.oracle_point_format.argtypes = [vp, ci, vp, vp, sz]
        L.oracle_from_hash_g2.argtypes = [vp, vp, ci, vp, sz]
        L.oracle_point_format_g2.argtypes = [vp, ci, vp, vp, sz]
        L.oracle_finalpow.argtypes = [vp, vp, vp, sz]
        L.oracle_g1_op.argtypes = [vp, ci, vp, vp, vp, sz]
        L.oracle_zr_op.argtypes = [vp, ci, vp, vp, ci, vp, sz]
        L.oracle_pow_multi.argtypes = [vp, ci, ci, vp, vp, sz, vp, sz]
        L.oracle_counters.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ci]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


class OraclePairing:
    """CPU oracle for one parameter set (mirrors pairing_t)."""

    def __init__(self, param_text):
        if isinstance(param_text, str):
            param_text = param_text.encode()
        self._h = ctypes.c_void_p()
        if lib().oracle_pairing_init(ctypes.byref(self._h), param_text, len(param_text)):
            raise ValueError("oracle_pairing_init failed")
        self.type = chr(lib().oracle_type(self._h))
        self.len_G1 = lib().oracle_len_G1(self._h)
        self.len_G2 = lib().oracle_len_G2(self._h)
        self.len_GT = lib().oracle_len_GT(self._h)

    def __del__(self):
        try:
            if self._h:
                lib().oracle_pairing_clear(self._h)
        except Exception:
            pass

    def pairing_batch(self, g1, g2):
        g1, g2 = _u8(g1), _u8(g2)
        n = g1.size // self.len_G1
        out = np.empty((n, self.len_GT), np.uint8)
        if lib().oracle_pairing_batch(self._h, _ptr(g1), _ptr(g2), _ptr(out), n):
            raise RuntimeError("oracle_pairing_batch failed")
        return out

    def prod_pairing_batch(self, g1, g2, k):
        g1, g2 = _u8(g1), _u8(g2)
        n = g1.size // (self.len_G1 * k)
        out = np.empty((n, self.len_GT), np.uint8)
        if lib().oracle_prod_pairing_batch(self._h, _ptr(g1), _ptr(g2), _ptr(out), n, k):
            raise RuntimeError("oracle_prod_pairing_batch failed")
        return out

    def fq_op(self, op, a, b=None):
        a = _u8(a)
        b = _u8(b) if b is not None else None
        L = self.len_G1 // 2           # G1 = E(Fq): x || y
        n = a.size // L
        out = np.empty((n, L), np.uint8)
        if lib().oracle_fq_op(self._h, op, _ptr(a), _ptr(b), _ptr(out), n):
            raise RuntimeError("oracle_fq_op failed")
        return out

    def gt_mul(self, a, b):
        a, b = _u8(a), _u8(b)
        n = a.size // self.len_GT
        out = np.empty((n, self.len_GT), np.uint8)
        if lib().oracle_gt_mul(self._h, _ptr(a), _ptr(b), _ptr(out), n):
            raise RuntimeError("oracle_gt_mul failed")
        return out

    def gt_pow(self, a, e):
        """a: (n, lenGT) bytes; e: (n, elen) big-endian exponent bytes."""
        a, e = _u8(a), _u8(e)
        n = a.size // self.len_GT
        elen = e.size // n
        out = np.empty((n, self.len_GT), np.uint8)
        if lib().oracle_gt_pow(self._h, _ptr(a), _ptr(e), elen, _ptr(out), n):
            raise RuntimeError("oracle_gt_pow failed")
        return out

    def from_hash(self, digests):
        """element_from_hash on G1: (n, hlen) digests -> (n, len1) points"""
        d = np.ascontiguousarray(digests, np.uint8)
        n, hlen = d.shape
        out = np.empty((n, self.len_G1), np.uint8)
        if lib().oracle_from_hash(self._h, _ptr(d), hlen, _ptr(out), n):
            raise RuntimeError("oracle_from_hash failed")
        return out

    def point_format(self, what, recs):
        """G1 point formats: 0 compress, 1 decompress, 2 to x-only, 3 from x-only"""
        recs = np.ascontiguousarray(recs, np.uint8)
        fb = self.len_G1 // 2
        li, lo = {0: (2 * fb, fb + 1), 1: (fb + 1, 2 * fb), 2: (2 * fb, fb), 3: (fb, 2 * fb)}[what]
        n = recs.size // li
        out = np.empty((n, lo), np.uint8)
        if lib().oracle_point_format(self._h, what, _ptr(recs), _ptr(out), n):
            raise RuntimeError("oracle_point_format failed")
        return out

    def from_hash_g2(self, digests):
        """element_from_hash on the G2 twist (types d, g, f)"""
        d = np.ascontiguousarray(digests, np.uint8)
        n, hlen = d.shape
        out = np.empty((n, self.len_G2), np.uint8)
        if lib().oracle_from_hash_g2(self._h, _ptr(d), hlen, _ptr(out), n):
            raise RuntimeError("oracle_from_hash_g2 failed")
        return out

    def point_format_g2(self, what, recs):
        """G2 twist point formats: 0 compress, 1 decompress, 2 to x-only, 3 from x-only"""
        recs = np.ascontiguousarray(recs, np.uint8)
        fb = self.len_G2 // 2
        li, lo = {0: (2 * fb, fb + 1), 1: (fb + 1, 2 * fb), 2: (2 * fb, fb), 3: (fb, 2 * fb)}[what]
        n = recs.size // li
        out = np.empty((n, lo), np.uint8)
        if lib().oracle_point_format_g2(self._h, what, _ptr(recs), _ptr(out), n):
            raise RuntimeError("oracle_point_format_g2 failed")
        return out

    def finalpow(self, a):
        """pairing->finalpow on (n, lenGT) records of GT's underlying field"""
        a = _u8(a)
        n = a.size // self.len_GT
        out = np.empty((n, self.len_GT), np.uint8)
        if lib().oracle_finalpow(self._h, _ptr(a), _ptr(out), n):
            raise RuntimeError("oracle_finalpow failed")
        return out

    def g_mul(self, group, pts, e):
        pts, e = _u8(pts), _u8(e)
        L = self.len_G1 if group == 1 else self.len_G2
        n = pts.size // L
        elen = e.size // n
        out = np.empty((n, L), np.uint8)
        if lib().oracle_g_mul(self._h, group, _ptr(pts), _ptr(e), elen, _ptr(out), n):
            raise RuntimeError("oracle_g_mul failed")
        return out


    # ---- round 5: group law on G1, Z_r, multi-exponentiations ----
    def g1_op(self, op, a, b=None):
        """op 0 a+b, 1 a-b, 2 -a, 3 2a on G1 records (O = zeros)"""
        a = _u8(a)
        b = None if b is None else _u8(b)
        n = a.size // self.len_G1
        out = np.empty((n, self.len_G1), np.uint8)
        if lib().oracle_g1_op(self._h, op, _ptr(a), None if b is None else _ptr(b), _ptr(out), n):
            raise RuntimeError("oracle_g1_op failed")
        return out

    def zr_op(self, op, a, b=None, hlen=0):
        """op 0 mul, 1 add, 2 sub, 3 invert, 4 neg, 5 halve, 6 double, 7 div, 8 from_hash (a: n x hlen digests)"""
        a = _u8(a)
        b = None if b is None else _u8(b)
        n = a.shape[0]
        lz = a.shape[1] if op != 8 else (b.shape[1] if b is not None else None)
        if op == 8:
            raise ValueError("use zr_from_hash")
        out = np.empty((n, lz), np.uint8)
        if lib().oracle_zr_op(self._h, op, _ptr(a), None if b is None else _ptr(b), 0, _ptr(out), n):
            raise RuntimeError("oracle_zr_op failed")
        return out

    def zr_from_hash(self, digests, len_zr):
        d = _u8(digests)
        n, hlen = d.shape
        out = np.empty((n, len_zr), np.uint8)
        if lib().oracle_zr_op(self._h, 8, _ptr(d), None, hlen, _ptr(out), n):
            raise RuntimeError("oracle_zr_op failed")
        return out

    def pow_multi(self, group, bases, scalars):
        """a1^n1 a2^n2 (a3^n3) on G1 (group 1, additive) or GT (group 3)"""
        k = len(bases)
        A = np.ascontiguousarray(np.stack([_u8(x) for x in bases], axis=1))
        E = np.ascontiguousarray(np.stack([_u8(x) for x in scalars], axis=1))
        n, elen = A.shape[0], E.shape[2]
        out = np.empty((n, A.shape[2]), np.uint8)
        if lib().oracle_pow_multi(self._h, group, k, _ptr(A), _ptr(E), elen, _ptr(out), n):
            raise RuntimeError("oracle_pow_multi failed")
        return out


def counters(reset=False):
    m, i = ctypes.c_uint64(), ctypes.c_uint64()
    lib().oracle_counters(ctypes.byref(m), ctypes.byref(i), int(reset))
    return m.value, i.value


class Vec:
    """A golden vector file written by ref_tool gen/kat (layout: oracle/ref_harness.c)."""

    def __init__(self, path):
        raw = open(path, "rb").read()
        assert raw[:8] == b"PBCVEC01", path
        t, n, k, l1, l2, lt = struct.unpack("<6I", raw[8:32])
        self.type, self.n, self.k, self.len1, self.len2, self.lenT = chr(t), n, k, l1, l2, lt
        off = 32
        a = np.frombuffer(raw, np.uint8)
        self.g1 = a[off:off + n * k * l1].reshape(n * k, l1).copy(); off += n * k * l1
        self.g2 = a[off:off + n * k * l2].reshape(n * k, l2).copy(); off += n * k * l2
        self.gt = a[off:off + n * lt].reshape(n, lt).copy(); off += n * lt
        assert off == len(raw), path


class Rec:
    """Record container of the reference harness's round-5 modes (ref_harness.c rec_write: gops, zrops, pow23):
    .arrays = list of uint8 matrices in file order, .type = the parameter type letter."""

    def __init__(self, path):
        raw = open(path, "rb").read()
        assert raw[:8] == b"PBCREC01", path
        t, count = struct.unpack("<2I", raw[8:16])
        self.type = chr(t)
        dims = struct.unpack("<%dI" % (2 * count), raw[16:16 + 8 * count])
        off = 16 + 8 * count
        a = np.frombuffer(raw, np.uint8)
        self.arrays = []
        for i in range(count):
            rows, width = dims[2 * i], dims[2 * i + 1]
            self.arrays.append(a[off:off + rows * width].reshape(rows, width).copy())
            off += rows * width
        assert off == len(raw), path


def ref_gen(param_path, mode, n, k, seed, out_path):
    """Run the compiled reference to produce a vector file (needs oracle/_ref/ref_tool)."""
    subprocess.check_call([REF_TOOL, "gen", param_path, mode, str(n), str(k), str(seed), out_path],
                          stderr=subprocess.DEVNULL)
    return Vec(out_path)


def usable_cores():
    """Host cores this process may really use: the smaller of the logical CPU count (affinity mask respected) and the
    cgroup CPU quota (a GPU box shows 256 logical CPUs under a 16-core quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def ref_soak(param_path, n_random, k, seed, out_path, rbits, workers=None):
    """SURVEY 8d's full-parity distribution from the compiled reference on every usable core (ref_harness.c `soak`:
    n_random uniformly random units of k terms; k == 1 adds crafted points -- limb patterns of the device library's
    Montgomery form, x = q - 1, ... -- and records with coordinates >= q).  Returns (Vec, the tool's JSON summary)."""
    import json
    out = subprocess.run([REF_TOOL, "soak", param_path, str(n_random), str(k), str(seed), out_path,
                          str(workers or usable_cores()), str(rbits)], check=True, capture_output=True, text=True)
    return Vec(out_path), json.loads(out.stdout.strip().splitlines()[-1])
