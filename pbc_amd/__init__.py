"""pbc_amd -- MI355X-native batched bilinear pairings behind PBC's pairing API.

The product is ``libpbc_hip.so`` (hand-written HIP for gfx950, C-ABI in
``include/pbc_hip.h``).  This package is only the thin Python host mirror used by the
tests and the benchmark: it loads the shared object with ctypes and exposes a ``Pairing``
object shaped like the reference's ``pairing_t`` (``pairing_init_set_buf``,
``pairing_length_in_bytes_*``, ``element_pairing`` / ``element_prod_pairing`` over
batches of ``element_to_bytes`` records).

There is no CPU fallback: if the HIP library is missing or no GPU is present, calls fail
loudly.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# PBC_HIP_LIB selects an experimental build variant (tools/ only); the default is the product.
LIB_PATH = os.path.join(_HERE, os.environ.get("PBC_HIP_LIB", "libpbc_hip.so"))
PARAM_DIR = os.path.join(_HERE, "param")

_lib = None


class PbcHipError(RuntimeError):
    pass


def build(force=False):
    """Compile libpbc_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs.append(os.path.join(_HERE, "..", "include", "pbc_hip.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-j8", "-C", _HERE])
    return LIB_PATH


def lib():
    """The loaded C-ABI library (raises if it has not been built: no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PbcHipError("%s is missing: run `make -C pbc_amd` (or __graft_entry__.build())" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        vp, cp, sz, ci = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int
        L.pbc_hip_pairing_init_set_buf.argtypes = [ctypes.POINTER(vp), cp, sz]
        L.pbc_hip_pairing_clear.argtypes = [vp]
        L.pbc_hip_pairing_clear.restype = None
        for f in ("pbc_hip_pairing_type", "pbc_hip_pairing_length_in_bytes_G1",
                  "pbc_hip_pairing_length_in_bytes_G2", "pbc_hip_pairing_length_in_bytes_GT",
                  "pbc_hip_length_in_bytes_Fq"):
            getattr(L, f).argtypes = [vp]
        L.pbc_hip_element_pairing_batch.argtypes = [vp, vp, vp, vp, sz]
        L.pbc_hip_element_pairing_batch_dev.argtypes = [vp, vp, vp, vp, sz, vp]
        L.pbc_hip_element_prod_pairing_batch.argtypes = [vp, vp, vp, vp, sz, ci]
        L.pbc_hip_element_prod_pairing_batch_dev.argtypes = [vp, vp, vp, vp, sz, ci, vp]
        L.pbc_hip_fq_op_batch.argtypes = [vp, ci, vp, vp, vp, sz]
        L.pbc_hip_int_mac_peak.argtypes = [ci, ci, ctypes.POINTER(ctypes.c_double),
                                           ctypes.POINTER(ctypes.c_double)]
        L.pbc_hip_diag_mul_bench.argtypes = [ci, ci, ci, ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_double)]
        L.pbc_hip_diag_stage.argtypes = [vp, ci, vp, sz, vp, vp, sz]
        L.pbc_hip_pairing_length_in_bytes_Zr.argtypes = [vp]
        L.pbc_hip_element_mul_zn_batch.argtypes = [vp, ci, vp, vp, vp, sz]
        L.pbc_hip_pairing_length_in_bytes_compressed_G1.argtypes = [vp]
        L.pbc_hip_pairing_length_in_bytes_compressed_G2.argtypes = [vp]
        L.pbc_hip_pairing_use_devices.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ci]
        L.pbc_hip_element_to_bytes_compressed_batch.argtypes = [vp, ci, vp, vp, sz]
        L.pbc_hip_element_from_bytes_compressed_batch.argtypes = [vp, ci, vp, vp, sz]
        L.pbc_hip_pairing_length_in_bytes_x_only_G1.argtypes = [vp]
        L.pbc_hip_pairing_length_in_bytes_x_only_G2.argtypes = [vp]
        L.pbc_hip_element_to_bytes_x_only_batch.argtypes = [vp, ci, vp, vp, sz]
        L.pbc_hip_element_from_bytes_x_only_batch.argtypes = [vp, ci, vp, vp, sz]
        L.pbc_hip_element_from_hash_batch.argtypes = [vp, ci, vp, vp, ci, sz]
        L.pbc_hip_element_mul_GT_batch.argtypes = [vp, vp, vp, vp, sz]
        L.pbc_hip_element_pow_zn_GT_batch.argtypes = [vp, vp, vp, vp, sz]
        L.pbc_hip_pairing_pp_init.argtypes = [ctypes.POINTER(vp), vp, vp]
        L.pbc_hip_pairing_pp_clear.argtypes = [vp]
        L.pbc_hip_pairing_pp_clear.restype = None
        L.pbc_hip_pairing_pp_apply_batch.argtypes = [vp, vp, vp, sz]
        L.pbc_hip_pairing_pp_apply_batch_dev.argtypes = [vp, vp, vp, sz, vp]
        L.pbc_hip_algorithmic_macs_per_unit.argtypes = [vp, ci]
        L.pbc_hip_algorithmic_macs_per_unit.restype = ctypes.c_double
        L.pbc_hip_last_error.restype = cp
        L.pbc_hip_finalpow_batch.argtypes = [vp, vp, vp, sz]
        L.pbc_hip_host_alloc.argtypes = [ctypes.POINTER(vp), sz]
        L.pbc_hip_host_free.argtypes = [vp]
        L.pbc_hip_host_free.restype = None
        L.pbc_hip_pairing_release_workspaces.argtypes = [vp]
        for f in ("add", "sub"):
            getattr(L, "pbc_hip_element_%s_batch" % f).argtypes = [vp, ci, vp, vp, vp, sz]
            getattr(L, "pbc_hip_element_%s_batch_dev" % f).argtypes = [vp, ci, vp, vp, vp, sz, vp]
        for f in ("neg", "double"):
            getattr(L, "pbc_hip_element_%s_batch" % f).argtypes = [vp, ci, vp, vp, sz]
            getattr(L, "pbc_hip_element_%s_batch_dev" % f).argtypes = [vp, ci, vp, vp, sz, vp]
        L.pbc_hip_zr_op_batch.argtypes = [vp, ci, vp, vp, vp, sz]
        L.pbc_hip_zr_op_batch_dev.argtypes = [vp, ci, vp, vp, vp, sz, vp]
        L.pbc_hip_zr_from_hash_batch.argtypes = [vp, vp, vp, ci, sz]
        L.pbc_hip_zr_from_hash_batch_dev.argtypes = [vp, vp, vp, ci, sz, vp]
        L.pbc_hip_element_pow2_zn_batch.argtypes = [vp, ci, vp] + [vp] * 4 + [sz]
        L.pbc_hip_element_pow3_zn_batch.argtypes = [vp, ci, vp] + [vp] * 6 + [sz]
        L.pbc_hip_element_pow2_zn_batch_dev.argtypes = [vp, ci, vp] + [vp] * 4 + [sz, vp]
        L.pbc_hip_element_pow3_zn_batch_dev.argtypes = [vp, ci, vp] + [vp] * 6 + [sz, vp]
        L.pbc_hip_element_snprint.argtypes = [vp, ci, ctypes.c_char_p, sz, vp]
        L.pbc_hip_element_mul_zn_batch_dev.argtypes = [vp, ci, vp, vp, vp, sz, vp]
        L.pbc_hip_element_mul_GT_batch_dev.argtypes = [vp, vp, vp, vp, sz, vp]
        L.pbc_hip_element_pow_zn_GT_batch_dev.argtypes = [vp, vp, vp, vp, sz, vp]
        L.pbc_hip_finalpow_batch_dev.argtypes = [vp, vp, vp, sz, vp]
        L.pbc_hip_element_from_hash_batch_dev.argtypes = [vp, ci, vp, vp, ci, sz, vp]
        for f in ("to_bytes_compressed", "from_bytes_compressed", "to_bytes_x_only", "from_bytes_x_only"):
            getattr(L, "pbc_hip_element_%s_batch_dev" % f).argtypes = [vp, ci, vp, vp, sz, vp]
        L.pbc_hip_element_pp_init.argtypes = [ctypes.POINTER(vp), vp, ci, vp]
        L.pbc_hip_element_pp_clear.argtypes = [vp]
        L.pbc_hip_element_pp_clear.restype = None
        L.pbc_hip_element_pp_pow_zn_batch.argtypes = [vp, vp, vp, sz]
        L.pbc_hip_element_pp_pow_zn_batch_dev.argtypes = [vp, vp, vp, sz, vp]
        L.pbc_hip_element_set_str.argtypes = [vp, ci, vp, cp, ci]
        L.pbc_hip_param_snprint.argtypes = [vp, ctypes.c_char_p, sz]
        L.pbc_hip_diag_dw_schedule.argtypes = [vp, ci, vp, sz]
        L.pbc_hip_diag_dw_schedule.restype = sz
        L.pbc_hip_diag_fw_schedule.argtypes = [vp, ci, vp, sz]
        L.pbc_hip_diag_fw_schedule.restype = sz
        L.pbc_hip_diag_gw_schedule.argtypes = [vp, ci, vp, sz]
        L.pbc_hip_diag_gw_schedule.restype = sz
        L.pbc_hip_diag_ag_table.argtypes = [vp, vp, sz]
        L.pbc_hip_diag_ag_table.restype = sz
        L.pbc_hip_fq_limb_image_bytes.argtypes = [vp]
        L.pbc_hip_element_pairing_batch_limbs.argtypes = [vp, vp, vp, vp, sz]
        L.pbc_hip_element_prod_pairing_batch_limbs.argtypes = [vp, vp, vp, vp, sz, ci]
        L.pbc_hip_element_prod_pairing_batch_limbs_dev.argtypes = [vp, vp, vp, vp, sz, ci, vp]
        _lib = L
    return _lib


#: every symbol include/pbc_hip.h declares (checked by the CPU test-suite)
EXPORTS = (
    "pbc_hip_pairing_init_set_buf", "pbc_hip_pairing_clear", "pbc_hip_pairing_type",
    "pbc_hip_pairing_length_in_bytes_G1", "pbc_hip_pairing_length_in_bytes_G2",
    "pbc_hip_pairing_length_in_bytes_GT", "pbc_hip_element_pairing_batch",
    "pbc_hip_element_pairing_batch_dev", "pbc_hip_element_prod_pairing_batch",
    "pbc_hip_element_prod_pairing_batch_dev", "pbc_hip_fq_op_batch",
    "pbc_hip_length_in_bytes_Fq", "pbc_hip_int_mac_peak",
    "pbc_hip_algorithmic_macs_per_unit", "pbc_hip_last_error", "pbc_hip_diag_mul_bench", "pbc_hip_diag_stage",
    "pbc_hip_pairing_pp_init", "pbc_hip_pairing_pp_clear", "pbc_hip_pairing_pp_apply_batch",
    "pbc_hip_pairing_pp_apply_batch_dev", "pbc_hip_pairing_length_in_bytes_Zr",
    "pbc_hip_element_from_hash_batch", "pbc_hip_element_mul_zn_batch", "pbc_hip_element_mul_GT_batch", "pbc_hip_element_pow_zn_GT_batch",
    "pbc_hip_pairing_length_in_bytes_compressed_G1", "pbc_hip_element_to_bytes_compressed_batch",
    "pbc_hip_element_from_bytes_compressed_batch", "pbc_hip_pairing_use_devices", "pbc_hip_device_count",
    "pbc_hip_pairing_length_in_bytes_x_only_G1", "pbc_hip_element_to_bytes_x_only_batch",
    "pbc_hip_pairing_length_in_bytes_compressed_G2", "pbc_hip_pairing_length_in_bytes_x_only_G2",
    "pbc_hip_element_from_bytes_x_only_batch", "pbc_hip_host_alloc", "pbc_hip_host_free", "pbc_hip_finalpow_batch",
    "pbc_hip_pairing_release_workspaces", "pbc_hip_element_snprint", "pbc_hip_element_set_str", "pbc_hip_param_snprint",
    "pbc_hip_element_mul_zn_batch_dev", "pbc_hip_element_mul_GT_batch_dev", "pbc_hip_element_pow_zn_GT_batch_dev",
    "pbc_hip_finalpow_batch_dev", "pbc_hip_element_from_hash_batch_dev", "pbc_hip_element_to_bytes_compressed_batch_dev",
    "pbc_hip_element_from_bytes_compressed_batch_dev", "pbc_hip_element_to_bytes_x_only_batch_dev",
    "pbc_hip_element_from_bytes_x_only_batch_dev", "pbc_hip_element_pp_init", "pbc_hip_element_pp_clear",
    "pbc_hip_element_pp_pow_zn_batch", "pbc_hip_element_pp_pow_zn_batch_dev",
    "pbc_hip_element_add_batch", "pbc_hip_element_sub_batch", "pbc_hip_element_neg_batch", "pbc_hip_element_double_batch",
    "pbc_hip_element_add_batch_dev", "pbc_hip_element_sub_batch_dev", "pbc_hip_element_neg_batch_dev", "pbc_hip_element_double_batch_dev",
    "pbc_hip_zr_op_batch", "pbc_hip_zr_op_batch_dev", "pbc_hip_zr_from_hash_batch", "pbc_hip_zr_from_hash_batch_dev",
    "pbc_hip_element_pow2_zn_batch", "pbc_hip_element_pow3_zn_batch", "pbc_hip_element_pow2_zn_batch_dev", "pbc_hip_element_pow3_zn_batch_dev",
    "pbc_hip_diag_dw_schedule", "pbc_hip_diag_fw_schedule", "pbc_hip_diag_gw_schedule", "pbc_hip_diag_ag_table", "pbc_hip_fq_limb_image_bytes", "pbc_hip_element_pairing_batch_limbs", "pbc_hip_element_prod_pairing_batch_limbs",
    "pbc_hip_element_prod_pairing_batch_limbs_dev",
)


def _err():
    m = lib().pbc_hip_last_error()
    return m.decode() if m else "unknown error"


def param_text(name):
    """Stock parameter sets shipped with the reference (param/a.param, d159.param, f.param)."""
    with open(os.path.join(PARAM_DIR, name + ".param")) as fh:
        return fh.read()


ZR_OPS = {"mul": 0, "add": 1, "sub": 2, "invert": 3, "neg": 4, "halve": 5, "double": 6, "div": 7}   # pbc_hip_zr_op_batch


def _np_ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


class Pairing:
    """Mirror of ``pairing_t``: built from a PBC parameter text, applied to batches.

    Host-side calls take/return numpy uint8 arrays of ``element_to_bytes`` records;
    ``*_dev`` calls take raw device pointers (e.g. ``torch.Tensor.data_ptr()``) and a
    HIP stream handle and only enqueue work.
    """

    def __init__(self, param):
        if isinstance(param, str):
            param = param.encode()
        self._h = ctypes.c_void_p()
        if lib().pbc_hip_pairing_init_set_buf(ctypes.byref(self._h), param, len(param)):
            self._h = None
            raise PbcHipError("pairing_init_set_buf: " + _err())
        L = lib()
        self.type = chr(L.pbc_hip_pairing_type(self._h))
        self.length_in_bytes_G1 = L.pbc_hip_pairing_length_in_bytes_G1(self._h)
        self.length_in_bytes_G2 = L.pbc_hip_pairing_length_in_bytes_G2(self._h)
        self.length_in_bytes_GT = L.pbc_hip_pairing_length_in_bytes_GT(self._h)
        self.length_in_bytes_Fq = L.pbc_hip_length_in_bytes_Fq(self._h)
        self.length_in_bytes_Zr = L.pbc_hip_pairing_length_in_bytes_Zr(self._h)
        self._param_text = param.decode(errors="replace")

    @property
    def field_order(self):
        """q (type a1: p), the order of the base field, from the parameter text"""
        for line in self._param_text.splitlines():
            f = line.split()
            if len(f) == 2 and f[0] in ("q", "p"):
                return int(f[1])
        raise PbcHipError("no q / p in the parameter text")

    def clear(self):
        if getattr(self, "_h", None):
            try:
                lib().pbc_hip_pairing_clear(self._h)
            except TypeError:          # interpreter shutdown: the module's globals are gone
                pass
            self._h = None

    __del__ = clear

    # ---- element_pairing over a batch -------------------------------------------------
    def element_pairing(self, g1, g2):
        import numpy as np
        g1 = np.ascontiguousarray(g1, dtype=np.uint8)
        g2 = np.ascontiguousarray(g2, dtype=np.uint8)
        n = g1.size // self.length_in_bytes_G1
        if g1.size != n * self.length_in_bytes_G1 or g2.size != n * self.length_in_bytes_G2:
            raise ValueError("g1/g2 sizes do not describe the same number of elements")
        gt = np.empty((n, self.length_in_bytes_GT), np.uint8)
        if lib().pbc_hip_element_pairing_batch(self._h, _np_ptr(gt), _np_ptr(g1), _np_ptr(g2), n):
            raise PbcHipError("element_pairing: " + _err())
        return gt

    def element_pairing_dev(self, d_gt, d_g1, d_g2, n, stream=0):
        if lib().pbc_hip_element_pairing_batch_dev(self._h, d_gt, d_g1, d_g2, n, stream):
            raise PbcHipError("element_pairing_dev: " + _err())

    # ---- element_prod_pairing over a batch of k-term products -------------------------
    def element_prod_pairing(self, g1, g2, k):
        import numpy as np
        g1 = np.ascontiguousarray(g1, dtype=np.uint8)
        g2 = np.ascontiguousarray(g2, dtype=np.uint8)
        n = g1.size // (self.length_in_bytes_G1 * k)
        if g1.size != n * k * self.length_in_bytes_G1 or g2.size != n * k * self.length_in_bytes_G2:
            raise ValueError("g1/g2 sizes do not describe n*k elements")
        gt = np.empty((n, self.length_in_bytes_GT), np.uint8)
        if lib().pbc_hip_element_prod_pairing_batch(self._h, _np_ptr(gt), _np_ptr(g1), _np_ptr(g2), n, k):
            raise PbcHipError("element_prod_pairing: " + _err())
        return gt

    # ---- the same on the reference's montfp limb images (include/pbc_hip.h; what integration/pbc_hip_glue.c exchanges) ----
    @property
    def limb_image_bytes(self):
        """8 t: bytes of one F_q coordinate as the reference's montfp element holds it"""
        w = lib().pbc_hip_fq_limb_image_bytes(self._h)
        if w <= 0:
            raise PbcHipError("limb image: " + _err())
        return w

    def to_limb_images(self, rec):
        """wire records (n, L) -> limb-image records: every F_q coordinate x as t little-endian 64-bit limbs of
        x 2^(64 t) mod q (host arithmetic; tests and tools)"""
        import numpy as np
        rec = np.ascontiguousarray(rec, dtype=np.uint8)
        fb, w = self.length_in_bytes_Fq, self.limb_image_bytes
        n, L = rec.shape
        out = np.zeros((n, L // fb * w), np.uint8)
        q, R = self.field_order, 1 << (8 * w)
        for i in range(n):
            for c in range(L // fb):
                x = int.from_bytes(rec[i, c * fb:(c + 1) * fb].tobytes(), "big") % q
                out[i, c * w:(c + 1) * w] = np.frombuffer((x * R % q).to_bytes(w, "little"), np.uint8)
        return out

    def from_limb_images(self, img):
        import numpy as np
        img = np.ascontiguousarray(img, dtype=np.uint8)
        fb, w = self.length_in_bytes_Fq, self.limb_image_bytes
        n, L = img.shape
        out = np.zeros((n, L // w * fb), np.uint8)
        q = self.field_order
        rinv = pow(1 << (8 * w), -1, q)
        for i in range(n):
            for c in range(L // w):
                x = int.from_bytes(img[i, c * w:(c + 1) * w].tobytes(), "little") * rinv % q
                out[i, c * fb:(c + 1) * fb] = np.frombuffer(x.to_bytes(fb, "big"), np.uint8)
        return out

    def element_prod_pairing_limbs(self, g1_img, g2_img, k=1):
        import numpy as np
        g1_img = np.ascontiguousarray(g1_img, dtype=np.uint8)
        g2_img = np.ascontiguousarray(g2_img, dtype=np.uint8)
        fb, w = self.length_in_bytes_Fq, self.limb_image_bytes
        n = g1_img.size // (self.length_in_bytes_G1 // fb * w * k)
        gt = np.empty((n, self.length_in_bytes_GT // fb * w), np.uint8)
        if lib().pbc_hip_element_prod_pairing_batch_limbs(self._h, _np_ptr(gt), _np_ptr(g1_img), _np_ptr(g2_img), n, k):
            raise PbcHipError("element_prod_pairing_limbs: " + _err())
        return gt

    def element_prod_pairing_dev(self, d_gt, d_g1, d_g2, n, k, stream=0):
        if lib().pbc_hip_element_prod_pairing_batch_dev(self._h, d_gt, d_g1, d_g2, n, k, stream):
            raise PbcHipError("element_prod_pairing_dev: " + _err())

    # ---- text formats (host side; include/pbc_hip.h) ------------------------------------------
    def _group_len(self, group):
        lens = {0: self.length_in_bytes_Zr, 1: self.length_in_bytes_G1, 2: self.length_in_bytes_G2, 3: self.length_in_bytes_GT}
        if group not in lens:
            raise PbcHipError("group must be 0 (Zr), 1, 2 or 3 (GT)")
        return lens[group]

    def element_snprint(self, group, rec, size=None):
        """element_snprint of the element a record deserialises to (group 0 Zr, 1 G1, 2 G2, 3 GT).  ``size``: the
        buffer size to pass (default: large enough); returns (text stored, length the full text has)."""
        import numpy as np
        rec = np.ascontiguousarray(rec, dtype=np.uint8).reshape(-1)
        if rec.size != self._group_len(group):
            raise ValueError("one record of the group expected")
        n = lib().pbc_hip_element_snprint(self._h, group, None, 0, _np_ptr(rec)) if size is None else None
        if n is not None and n < 0:
            raise PbcHipError("element_snprint: " + _err())
        buf = ctypes.create_string_buffer((n + 1) if size is None else max(size, 1))
        full = lib().pbc_hip_element_snprint(self._h, group, buf, len(buf) if size is None else size, _np_ptr(rec))
        if full < 0:
            raise PbcHipError("element_snprint: " + _err())
        return (buf.value.decode() if (size is None or size > 0) else ""), full

    def element_set_str(self, group, text, base=10):
        """element_set_str: returns (record, characters consumed); 0 consumed = the reference's failure value."""
        import numpy as np
        rec = np.zeros(self._group_len(group), np.uint8)
        used = lib().pbc_hip_element_set_str(self._h, group, _np_ptr(rec), text.encode() if isinstance(text, str) else text, base)
        return rec, used

    def param_snprint(self):
        """pbc_param_out_str of the object's parameters"""
        n = lib().pbc_hip_param_snprint(self._h, None, 0)
        if n < 0:
            raise PbcHipError("param_snprint: " + _err())
        buf = ctypes.create_string_buffer(n + 1)
        lib().pbc_hip_param_snprint(self._h, buf, n + 1)
        return buf.value.decode()

    def release_workspaces(self):
        """free the per-stream workspaces of the product kernels now (include/pbc_hip.h)"""
        if lib().pbc_hip_pairing_release_workspaces(self._h):
            raise PbcHipError("release_workspaces: " + _err())

    # ---- group operations ------------------------------------------------------------------
    def _scalars(self, zr, n):
        import numpy as np
        zr = np.ascontiguousarray(zr, dtype=np.uint8)
        if zr.size != n * self.length_in_bytes_Zr:
            raise ValueError("scalars must be n records of length_in_bytes_Zr big-endian bytes")
        return zr

    def element_mul_zn(self, group, pts, zr):
        """out[i] = [zr[i]] pts[i] in G1 (group=1) or G2 (group=2)."""
        import numpy as np
        pts = np.ascontiguousarray(pts, dtype=np.uint8)
        lp = self.length_in_bytes_G1 if group == 1 else self.length_in_bytes_G2
        n = pts.size // lp
        zr = self._scalars(zr, n)
        out = np.empty((n, lp), np.uint8)
        if lib().pbc_hip_element_mul_zn_batch(self._h, group, _np_ptr(out), _np_ptr(pts), _np_ptr(zr), n):
            raise PbcHipError("element_mul_zn: " + _err())
        return out

    def _point_len(self, group):
        return self.length_in_bytes_G2 if group == 2 else self.length_in_bytes_G1

    # ---- the group law, Z_r, multi-exponentiations (round 5) -----------------------------------------
    def element_group_op(self, what, group, a, b=None):
        """what: "add", "sub" (two operands), "neg", "double" on records of G1 / G2 (element_add / element_sub /
        element_neg / element_double; O is the all-zero record)."""
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.uint8)
        lp = self._point_len(group)
        n = a.size // lp
        out = np.empty((n, lp), np.uint8)
        f = getattr(lib(), "pbc_hip_element_%s_batch" % what)
        if what in ("add", "sub"):
            b = np.ascontiguousarray(b, dtype=np.uint8)
            rc = f(self._h, group, _np_ptr(out), _np_ptr(a), _np_ptr(b), n)
        else:
            rc = f(self._h, group, _np_ptr(out), _np_ptr(a), n)
        if rc:
            raise PbcHipError("element_%s: " % what + _err())
        return out


    def zr_op(self, what, a, b=None):
        """Z_r arithmetic on element_to_bytes records of Zr: what in pbc_amd.ZR_OPS."""
        import numpy as np
        lz = self.length_in_bytes_Zr
        a = np.ascontiguousarray(a, dtype=np.uint8)
        n = a.size // lz
        out = np.empty((n, lz), np.uint8)
        b = None if b is None else np.ascontiguousarray(b, dtype=np.uint8)
        if lib().pbc_hip_zr_op_batch(self._h, ZR_OPS[what], _np_ptr(out), _np_ptr(a), _np_ptr(b), n):
            raise PbcHipError("zr_%s: " % what + _err())
        return out

    def zr_from_hash(self, digests):
        import numpy as np
        d = np.ascontiguousarray(digests, dtype=np.uint8)
        n, hlen = d.shape
        out = np.empty((n, self.length_in_bytes_Zr), np.uint8)
        if lib().pbc_hip_zr_from_hash_batch(self._h, _np_ptr(out), _np_ptr(d), hlen, n):
            raise PbcHipError("zr_from_hash: " + _err())
        return out

    def element_pow_multi(self, group, bases, scalars):
        """element_pow2_zn / element_pow3_zn: bases = [a1, a2(, a3)] records of group 1 / 2 / 3 (GT), scalars = [n1, n2(, n3)]."""
        import numpy as np
        k = len(bases)
        lp = self._group_len(group)
        bases = [np.ascontiguousarray(x, dtype=np.uint8) for x in bases]
        n = bases[0].size // lp
        scalars = [self._scalars(z, n) for z in scalars]
        out = np.empty((n, lp), np.uint8)
        args = []
        for j in range(k):
            args += [_np_ptr(bases[j]), _np_ptr(scalars[j])]
        f = lib().pbc_hip_element_pow2_zn_batch if k == 2 else lib().pbc_hip_element_pow3_zn_batch
        if f(self._h, group, _np_ptr(out), *args, n):
            raise PbcHipError("element_pow%d_zn: " % k + _err())
        return out

    def element_group_op_dev(self, what, group, d_out, d_a, d_b, n, stream=0):
        f = getattr(lib(), "pbc_hip_element_%s_batch_dev" % what)
        rc = f(self._h, group, d_out, d_a, d_b, n, stream) if what in ("add", "sub") else f(self._h, group, d_out, d_a, n, stream)
        if rc:
            raise PbcHipError("element_%s_dev: " % what + _err())

    def zr_op_dev(self, what, d_out, d_a, d_b, n, stream=0):
        if lib().pbc_hip_zr_op_batch_dev(self._h, ZR_OPS[what], d_out, d_a, d_b, n, stream):
            raise PbcHipError("zr_%s_dev: " % what + _err())

    def zr_from_hash_dev(self, d_out, d_data, hlen, n, stream=0):
        if lib().pbc_hip_zr_from_hash_batch_dev(self._h, d_out, d_data, hlen, n, stream):
            raise PbcHipError("zr_from_hash_dev: " + _err())

    def element_pow_multi_dev(self, group, d_out, d_bases, d_scalars, n, stream=0):
        k = len(d_bases)
        args = []
        for j in range(k):
            args += [d_bases[j], d_scalars[j]]
        f = lib().pbc_hip_element_pow2_zn_batch_dev if k == 2 else lib().pbc_hip_element_pow3_zn_batch_dev
        if f(self._h, group, d_out, *args, n, stream):
            raise PbcHipError("element_pow%d_zn_dev: " % k + _err())

    def _compressed_len(self, group):
        f = lib().pbc_hip_pairing_length_in_bytes_compressed_G2 if group == 2 else lib().pbc_hip_pairing_length_in_bytes_compressed_G1
        return f(self._h)

    def element_to_bytes_compressed(self, group, pts):
        """x||y records -> x||sign records (element_to_bytes_compressed)."""
        import numpy as np
        pts = np.ascontiguousarray(pts, dtype=np.uint8)
        n = pts.size // self._point_len(group)
        out = np.empty((n, self._compressed_len(group)), np.uint8)
        if lib().pbc_hip_element_to_bytes_compressed_batch(self._h, group, _np_ptr(out), _np_ptr(pts), n):
            raise PbcHipError("element_to_bytes_compressed: " + _err())
        return out

    def element_from_bytes_compressed(self, group, recs):
        """x||sign records -> x||y records (element_from_bytes_compressed)."""
        import numpy as np
        recs = np.ascontiguousarray(recs, dtype=np.uint8)
        n = recs.size // self._compressed_len(group)
        out = np.empty((n, self._point_len(group)), np.uint8)
        if lib().pbc_hip_element_from_bytes_compressed_batch(self._h, group, _np_ptr(out), _np_ptr(recs), n):
            raise PbcHipError("element_from_bytes_compressed: " + _err())
        return out

    def _x_only_len(self, group):
        f = lib().pbc_hip_pairing_length_in_bytes_x_only_G2 if group == 2 else lib().pbc_hip_pairing_length_in_bytes_x_only_G1
        return f(self._h)

    def element_to_bytes_x_only(self, group, pts):
        """x||y records -> x records (element_to_bytes_x_only)."""
        import numpy as np
        pts = np.ascontiguousarray(pts, dtype=np.uint8)
        n = pts.size // self._point_len(group)
        out = np.empty((n, self._x_only_len(group)), np.uint8)
        if lib().pbc_hip_element_to_bytes_x_only_batch(self._h, group, _np_ptr(out), _np_ptr(pts), n):
            raise PbcHipError("element_to_bytes_x_only: " + _err())
        return out

    def element_from_bytes_x_only(self, group, recs):
        """x records -> x||y records (element_from_bytes_x_only; y is the root element_sqrt picks)."""
        import numpy as np
        recs = np.ascontiguousarray(recs, dtype=np.uint8)
        n = recs.size // self._x_only_len(group)
        out = np.empty((n, self._point_len(group)), np.uint8)
        if lib().pbc_hip_element_from_bytes_x_only_batch(self._h, group, _np_ptr(out), _np_ptr(recs), n):
            raise PbcHipError("element_from_bytes_x_only: " + _err())
        return out

    def element_from_hash(self, group, digests):
        """digests: (n, hlen) uint8 -> n points of G1 or G2 (element_from_hash)."""
        import numpy as np
        d = np.ascontiguousarray(digests, dtype=np.uint8)
        n, hlen = d.shape
        out = np.empty((n, self._point_len(group)), np.uint8)
        if lib().pbc_hip_element_from_hash_batch(self._h, group, _np_ptr(out), _np_ptr(d), hlen, n):
            raise PbcHipError("element_from_hash: " + _err())
        return out

    def element_mul_GT(self, a, b):
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8)
        n = a.size // self.length_in_bytes_GT
        out = np.empty((n, self.length_in_bytes_GT), np.uint8)
        if lib().pbc_hip_element_mul_GT_batch(self._h, _np_ptr(out), _np_ptr(a), _np_ptr(b), n):
            raise PbcHipError("element_mul_GT: " + _err())
        return out

    def element_pow_zn_GT(self, a, zr):
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.uint8)
        n = a.size // self.length_in_bytes_GT
        zr = self._scalars(zr, n)
        out = np.empty((n, self.length_in_bytes_GT), np.uint8)
        if lib().pbc_hip_element_pow_zn_GT_batch(self._h, _np_ptr(out), _np_ptr(a), _np_ptr(zr), n):
            raise PbcHipError("element_pow_zn_GT: " + _err())
        return out

    # ---- the same on device-resident buffers (raw pointers, a HIP stream; asynchronous) ------------------
    def element_mul_zn_dev(self, group, d_out, d_in, d_zr, n, stream=0):
        if lib().pbc_hip_element_mul_zn_batch_dev(self._h, group, d_out, d_in, d_zr, n, stream):
            raise PbcHipError("element_mul_zn_dev: " + _err())

    def element_mul_GT_dev(self, d_out, d_a, d_b, n, stream=0):
        if lib().pbc_hip_element_mul_GT_batch_dev(self._h, d_out, d_a, d_b, n, stream):
            raise PbcHipError("element_mul_GT_dev: " + _err())

    def element_pow_zn_GT_dev(self, d_out, d_a, d_zr, n, stream=0):
        if lib().pbc_hip_element_pow_zn_GT_batch_dev(self._h, d_out, d_a, d_zr, n, stream):
            raise PbcHipError("element_pow_zn_GT_dev: " + _err())

    def finalpow_dev(self, d_out, d_in, n, stream=0):
        if lib().pbc_hip_finalpow_batch_dev(self._h, d_out, d_in, n, stream):
            raise PbcHipError("finalpow_dev: " + _err())

    def element_from_hash_dev(self, group, d_out, d_data, hlen, n, stream=0):
        if lib().pbc_hip_element_from_hash_batch_dev(self._h, group, d_out, d_data, hlen, n, stream):
            raise PbcHipError("element_from_hash_dev: " + _err())

    def point_format_dev(self, what, group, d_out, d_in, n, stream=0):
        """what: to_bytes_compressed, from_bytes_compressed, to_bytes_x_only, from_bytes_x_only"""
        if getattr(lib(), "pbc_hip_element_%s_batch_dev" % what)(self._h, group, d_out, d_in, n, stream):
            raise PbcHipError(what + "_dev: " + _err())

    def element_pp_init(self, group, rec):
        """Mirror of element_pp_init: fixed-base powers of one element of G1 / G2 (group 1, 2) or GT (group 3)."""
        return ElementPP(self, group, rec)

    # ---- preprocessed pairings (pairing_pp_init / pairing_pp_apply) ----------------------
    def finalpow(self, a):
        """pairing->finalpow over (n, lenGT) records of GT's underlying field"""
        import numpy as np
        a = np.ascontiguousarray(a, np.uint8)
        n = a.size // self.length_in_bytes_GT
        out = np.empty((n, self.length_in_bytes_GT), np.uint8)
        if lib().pbc_hip_finalpow_batch(self._h, _np_ptr(out), _np_ptr(a), n):
            raise PbcHipError(_err())
        return out

    def pp_init(self, g1):
        """Mirror of pairing_pp_init: returns a PairingPP bound to the fixed first argument."""
        return PairingPP(self, g1)

    # ---- diagnostics ------------------------------------------------------------------
    def fq_op(self, op, a, b=None):
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8) if b is not None else None
        n = a.size // self.length_in_bytes_Fq
        c = np.empty((n, self.length_in_bytes_Fq), np.uint8)
        if lib().pbc_hip_fq_op_batch(self._h, op, _np_ptr(c), _np_ptr(a), _np_ptr(b), n):
            raise PbcHipError("fq_op: " + _err())
        return c

    def use_devices(self, devices):
        """range-split host-buffer batches over these HIP ordinals ([] = back to the creation device)"""
        arr = (ctypes.c_int * len(devices))(*devices)
        if lib().pbc_hip_pairing_use_devices(self._h, arr, len(devices)):
            raise PbcHipError("use_devices: " + _err())

    def algorithmic_macs_per_unit(self, k=1):
        return lib().pbc_hip_algorithmic_macs_per_unit(self._h, k)


class PairingPP:
    """Mirror of pairing_pp_t (include/pbc_pairing.h:10-15, 54-89)."""

    def __init__(self, pairing, g1):
        import numpy as np
        self.pairing = pairing
        g1 = np.ascontiguousarray(g1, dtype=np.uint8).reshape(-1)
        if g1.size != pairing.length_in_bytes_G1:
            raise ValueError("g1 must be one G1 record")
        self._h = ctypes.c_void_p()
        if lib().pbc_hip_pairing_pp_init(ctypes.byref(self._h), pairing._h, _np_ptr(g1)):
            self._h = None
            raise PbcHipError("pairing_pp_init: " + _err())

    def apply(self, g2):
        """pairing_pp_apply over a batch of second arguments."""
        import numpy as np
        g2 = np.ascontiguousarray(g2, dtype=np.uint8)
        n = g2.size // self.pairing.length_in_bytes_G2
        gt = np.empty((n, self.pairing.length_in_bytes_GT), np.uint8)
        if lib().pbc_hip_pairing_pp_apply_batch(self._h, _np_ptr(gt), _np_ptr(g2), n):
            raise PbcHipError("pairing_pp_apply: " + _err())
        return gt

    def apply_dev(self, d_gt, d_g2, n, stream=0):
        if lib().pbc_hip_pairing_pp_apply_batch_dev(self._h, d_gt, d_g2, n, stream):
            raise PbcHipError("pairing_pp_apply_dev: " + _err())

    def clear(self):
        if getattr(self, "_h", None):
            lib().pbc_hip_pairing_pp_clear(self._h)
            self._h = None

    __del__ = clear


class ElementPP:
    """Mirror of element_pp_t (include/pbc_field.h:591-625): element_pp_init / element_pp_pow_zn / element_pp_clear."""

    def __init__(self, pairing, group, rec):
        import numpy as np
        self.pairing, self.group = pairing, group
        rec = np.ascontiguousarray(rec, dtype=np.uint8).reshape(-1)
        self.length = pairing._group_len(group)
        if rec.size != self.length:
            raise ValueError("one record of the group expected")
        self._h = ctypes.c_void_p()
        if lib().pbc_hip_element_pp_init(ctypes.byref(self._h), pairing._h, group, _np_ptr(rec)):
            self._h = None
            raise PbcHipError("element_pp_init: " + _err())

    def pow_zn(self, zr):
        import numpy as np
        zr = np.ascontiguousarray(zr, dtype=np.uint8)
        n = zr.size // self.pairing.length_in_bytes_Zr
        out = np.empty((n, self.length), np.uint8)
        if lib().pbc_hip_element_pp_pow_zn_batch(self._h, _np_ptr(out), _np_ptr(zr), n):
            raise PbcHipError("element_pp_pow_zn: " + _err())
        return out

    def pow_zn_dev(self, d_out, d_zr, n, stream=0):
        if lib().pbc_hip_element_pp_pow_zn_batch_dev(self._h, d_out, d_zr, n, stream):
            raise PbcHipError("element_pp_pow_zn_dev: " + _err())

    def clear(self):
        if getattr(self, "_h", None):
            try:
                lib().pbc_hip_element_pp_clear(self._h)
            except TypeError:          # interpreter shutdown: the module's globals are gone
                pass
            self._h = None

    __del__ = clear


def int_mac_peak(variant=0, iters=2000):
    """Register-only instruction-throughput probe; returns (lane-ops per second, ms)."""
    r, ms = ctypes.c_double(), ctypes.c_double()
    if lib().pbc_hip_int_mac_peak(variant, iters, ctypes.byref(r), ctypes.byref(ms)):
        raise PbcHipError("int_mac_peak: " + _err())
    return r.value, ms.value


def mul_bench(variant, iters=400, waves_per_simd=1):
    """512-bit multiplier micro-benchmark; returns (F_q products per second, ms)."""
    r, ms = ctypes.c_double(), ctypes.c_double()
    if lib().pbc_hip_diag_mul_bench(variant, iters, waves_per_simd, ctypes.byref(r), ctypes.byref(ms)):
        raise PbcHipError("mul_bench: " + _err())
    return r.value, ms.value
