"""GPU tests (-m gpu) of the group operations as a first-class path (round 4): the fast ladders (limb-form type a, regular
signed windows over every field policy, Lucas-ladder GT powers) against the complete ladders, the oracle and the
reference's vectors; the _dev / stream forms; fixed-base tables (element_pp_*) against the reference's
element_pp_pow_zn; the flow of example/bls.c as a device-resident batch against the reference's own run of it."""
import ctypes
import os
import struct

import numpy as np
import pytest

from conftest import golden, _param, PARAM_OF, param_value

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _be(x, n):
    return np.frombuffer(int(x).to_bytes(n, "big"), np.uint8)


def _order(key):
    p = PARAM_OF.get(key, key)
    try:
        return param_value(p, "r")
    except KeyError:
        return param_value(p, "n")


def _scalars(key, n, seed, edge=True):
    r = _order(key)
    zl = (r.bit_length() + 7) // 8
    rng = np.random.default_rng(seed)
    ks = [int.from_bytes(rng.bytes(zl), "big") % r for _ in range(n)]
    if edge:
        for i, k in enumerate([0, 1, 2, 3, r - 1, r - 2, r % (1 << (8 * zl)), (1 << (8 * zl)) - 1, (1 << (8 * zl)) - 2, 15, 16, 17]):
            if i < n:
                ks[i] = k
    return np.stack([_be(k, zl) for k in ks]), zl


FAMILIES = [("a", "a_rand32.vec"), ("d", "d_rand32.vec"), ("f", "f_rand16.vec"), ("g149", "g149_rand16.vec"), ("d201", "d201_rand12.vec"),
            ("f_256", "f_256_rand4.vec"), ("e", "e_rand6.vec"), ("a1", "a1_rand6.vec"), ("a_160_256", "a_160_256_rand6.vec")]


@pytest.mark.parametrize("key,name", FAMILIES)
def test_fast_and_complete_ladders_agree(hips, oracles, key, name):
    """element_mul_zn (G1 and G2) and element_pow_zn on GT through the default route -- fast pass, flagged lanes through
    the complete pass -- against the same library restricted to the complete ladders ("hip_group_slow 1") and, on a
    sample, the oracle: random scalars below r, the exceptional ones (0, 1, r - 1, r, all-ones: the lanes the fast pass
    must report), ragged n."""
    import pbc_amd
    v = golden(name)
    n = 700 if key in ("a1", "e") else 2500
    Z, zl = _scalars(key, n, 3)
    H = hips[key]
    S = pbc_amd.Pairing(_param(PARAM_OF.get(key, key)) + "hip_group_slow 1\n")
    i = np.arange(n) % v.n
    for group, src in ((1, v.g1), (2, v.g2)):
        pts = np.ascontiguousarray(src[i])
        got = H.element_mul_zn(group, pts, Z)
        assert np.array_equal(got, S.element_mul_zn(group, pts, Z)), (key, group)
        if group == 1 or key in ("a", "e", "a1", "a_160_256"):
            m = 24
            assert np.array_equal(got[:m], oracles[key].g_mul(group, pts[:m], Z[:m]))
    g = np.ascontiguousarray(v.gt[i])
    got = H.element_pow_zn_GT(g, Z)
    assert np.array_equal(got, S.element_pow_zn_GT(g, Z))
    assert np.array_equal(got[:16], oracles[key].gt_pow(g[:16], Z[:16]))
    S.clear()


def test_type_a_group_ops_on_points_of_small_order_and_other_norms(hip_a, oracle_a):
    """What only the complete pass can serve, mixed into a batch of ordinary lanes: points of order 2 and 4 and points
    outside the order-r subgroup (reference vectors a_g1mulfull6), off-curve records (O), GT elements that are not of
    norm 1 (random field elements: the Lucas ladder does not apply)."""
    v = golden("a_rand32.vec")
    w = golden("a_g1mulfull6.vec")
    assert np.array_equal(hip_a.element_mul_zn(1, w.g1, w.g2), w.gt)
    Z, zl = _scalars("a", 64, 9)
    pts = np.ascontiguousarray(np.tile(v.g1, (2, 1)))
    pts[5] = 0                                                 # (0, 0): the point of order 2
    pts[9, 127] ^= 1                                           # off the curve: O
    pts[11] = w.g1[2]
    assert np.array_equal(hip_a.element_mul_zn(1, pts, Z), oracle_a.g_mul(1, pts, Z))
    rng = np.random.default_rng(2)
    A = np.ascontiguousarray(np.tile(v.gt, (2, 1)))
    A[3::7] = rng.integers(0, 256, A[3::7].shape, dtype=np.uint8)
    A[3::7, 0] = A[3::7, 64] = 0
    assert np.array_equal(hip_a.element_pow_zn_GT(A, Z), oracle_a.gt_pow(A, Z))


@pytest.mark.parametrize("extra", ["", "hip_no_xs 1\n", "hip_no_bm1 1\n"])
def test_type_f_gt_powers_outside_the_cyclotomic_subgroup(oracles, extra):
    """f.param element_pow_zn on GT: the fast pass (cyclotomic squarings in the pairing kernels' basis) serves pairing
    values; random F_q^12 elements and 0 fail its membership test and take the generic ladder; both in one ragged batch."""
    import pbc_amd
    H = pbc_amd.Pairing(_param("f") + extra)
    v = golden("f_rand16.vec")
    n = 389
    Z, zl = _scalars("f", n, 21)
    rng = np.random.default_rng(4)
    A = np.ascontiguousarray(v.gt[np.arange(n) % v.n])
    A[5::9] = rng.integers(0, 256, A[5::9].shape, dtype=np.uint8)
    A[5::9, ::20] = 0
    A[14] = 0
    m = 64
    got = H.element_pow_zn_GT(A, Z)
    assert np.array_equal(got[:m], oracles["f"].gt_pow(A[:m], Z[:m]))
    S = pbc_amd.Pairing(_param("f") + "hip_group_slow 1\n")
    assert np.array_equal(got, S.element_pow_zn_GT(A, Z))
    S.clear()
    H.clear()


@pytest.mark.parametrize("key,name", [("a", "a_chain1024.vec"), ("d", "d_chain256.vec"), ("f", "f_chain128.vec")])
def test_group_dev_entry_points_streams_and_in_place(hips, key, name):
    """The _dev forms on torch buffers: two streams in flight, out == in (element_mul_zn(x, x, k) is ordinary PBC usage),
    a batch above one residency of the resident kernels with a ragged tail; bytes equal to the host forms."""
    import torch
    H = hips[key]
    v = golden(name)
    n = 1024 * 128 + 77 if key == "a" else 9000
    Z, zl = _scalars(key, n, 5)
    i = (np.arange(n) * 3) % v.n
    want1 = H.element_mul_zn(1, v.g1[i], Z)
    want2 = H.element_mul_zn(2, v.g2[i], Z)
    wantT = H.element_pow_zn_GT(v.gt[i], Z)
    dZ = torch.from_numpy(Z).cuda()
    x1, x2, xt = torch.from_numpy(v.g1[i]).cuda(), torch.from_numpy(v.g2[i]).cuda(), torch.from_numpy(v.gt[i]).cuda()
    o2 = torch.full((n + 8, v.len2), 0xA5, dtype=torch.uint8, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    H.element_mul_zn_dev(1, x1.data_ptr(), x1.data_ptr(), dZ.data_ptr(), n, s1.cuda_stream)      # in place
    H.element_mul_zn_dev(2, o2.data_ptr(), x2.data_ptr(), dZ.data_ptr(), n, s2.cuda_stream)
    H.element_pow_zn_GT_dev(xt.data_ptr(), xt.data_ptr(), dZ.data_ptr(), n, s1.cuda_stream)     # in place, same stream as the first
    torch.cuda.synchronize()
    assert np.array_equal(x1.cpu().numpy(), want1)
    got2 = o2.cpu().numpy()
    assert np.array_equal(got2[:n], want2) and (got2[n:] == 0xA5).all()
    assert np.array_equal(xt.cpu().numpy(), wantT)
    # GT product, final power, hashing, point formats on device buffers
    a, b = torch.from_numpy(v.gt[i]).cuda(), torch.from_numpy(v.gt[(i + 1) % v.n]).cuda()
    o = torch.empty_like(a)
    H.element_mul_GT_dev(o.data_ptr(), a.data_ptr(), b.data_ptr(), n, 0)
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy(), H.element_mul_GT(v.gt[i], v.gt[(i + 1) % v.n]))
    m = 600
    H.finalpow_dev(o.data_ptr(), a.data_ptr(), m, 0)
    torch.cuda.synchronize()
    assert np.array_equal(o[:m].cpu().numpy(), H.finalpow(v.gt[i[:m]]))
    dig = np.random.default_rng(1).integers(0, 256, (m, 32), dtype=np.uint8)
    dd = torch.from_numpy(dig).cuda()
    for group, lp in ((1, v.len1), (2, v.len2)):
        pts = torch.empty(m, lp, dtype=torch.uint8, device="cuda")
        H.element_from_hash_dev(group, pts.data_ptr(), dd.data_ptr(), 32, m, s1.cuda_stream)
        s1.synchronize()
        hp = H.element_from_hash(group, dig)
        assert np.array_equal(pts.cpu().numpy(), hp)
        lc = lp // 2 + 1
        c = torch.empty(m, lc, dtype=torch.uint8, device="cuda")
        back = torch.empty_like(pts)
        H.point_format_dev("to_bytes_compressed", group, c.data_ptr(), pts.data_ptr(), m, s2.cuda_stream)
        H.point_format_dev("from_bytes_compressed", group, back.data_ptr(), c.data_ptr(), m, s2.cuda_stream)
        s2.synchronize()
        assert np.array_equal(c.cpu().numpy(), H.element_to_bytes_compressed(group, hp))
        assert torch.equal(back, pts)
        xo = torch.empty(m, lp // 2, dtype=torch.uint8, device="cuda")
        H.point_format_dev("to_bytes_x_only", group, xo.data_ptr(), pts.data_ptr(), m, 0)
        H.point_format_dev("from_bytes_x_only", group, back.data_ptr(), xo.data_ptr(), m, 0)
        torch.cuda.synchronize()
        assert np.array_equal(back.cpu().numpy(), H.element_from_bytes_x_only(group, H.element_to_bytes_x_only(group, hp)))


def test_type_a_group_kernels_with_a_forced_small_grid(hips):
    """al_gmul_kernel / al_gtpow_kernel wrap their bodies in the resident loop: three workgroups over 1000 units."""
    import pbc_amd
    v = golden("a_chain1024.vec")
    P = pbc_amd.Pairing(_param("a") + "hip_resident_slots 3\n")
    Z, zl = _scalars("a", 1000, 8)
    assert np.array_equal(P.element_mul_zn(1, v.g1[:1000], Z), hips["a"].element_mul_zn(1, v.g1[:1000], Z))
    assert np.array_equal(P.element_pow_zn_GT(v.gt[:1000], Z), hips["a"].element_pow_zn_GT(v.gt[:1000], Z))
    P.clear()


PP_VECTORS = [("a", 1), ("a", 3), ("d159", 1), ("d159", 2), ("d159", 3), ("f", 1), ("f", 2), ("f", 3), ("g149", 1), ("e", 1), ("d201", 2)]


@pytest.mark.parametrize("pname,group", PP_VECTORS)
def test_element_pp_matches_reference(hips, pname, group):
    """element_pp_init + element_pp_pow_zn against what the reference's base table gives (ref_tool ppow: one base, random
    scalars and 0, 1, r - 1), then against element_mul_zn / element_pow_zn of this library on 3000 fresh scalars, host and
    _dev forms."""
    import torch
    key = {"a": "a", "d159": "d", "f": "f"}.get(pname, pname)
    H = hips[key]
    v = golden("%s_pp%dpow12.vec" % (pname, group))
    pp = H.element_pp_init(group, v.g1[0])
    assert np.array_equal(pp.pow_zn(v.g2), v.gt)
    n = 3000
    Z, zl = _scalars(key, n, 21)
    B = np.tile(v.g1[0], (n, 1))
    want = H.element_pow_zn_GT(B, Z) if group == 3 else H.element_mul_zn(group, B, Z)
    assert np.array_equal(pp.pow_zn(Z), want)
    dZ = torch.from_numpy(Z).cuda()
    o = torch.full((n + 4, v.len1), 0xA5, dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    pp.pow_zn_dev(o.data_ptr(), dZ.data_ptr(), n, st.cuda_stream)
    st.synchronize()
    got = o.cpu().numpy()
    assert np.array_equal(got[:n], want) and (got[n:] == 0xA5).all()
    pp.clear()


def test_element_pp_bases_of_small_order(hip_a, oracle_a):
    """A base whose multiples run into O inside the table (the point of order 2, a point of order 4 from the whole-curve
    vectors times (q + 1) / 4 ...) or that is off the curve: element_pp serves it through the complete ladder."""
    w = golden("a_g1mulfull6.vec")
    Z, zl = _scalars("a", 40, 4)
    for base in (np.zeros(128, np.uint8), w.g1[0], np.full(128, 7, np.uint8)):
        pp = hip_a.element_pp_init(1, base)
        B = np.tile(base, (len(Z), 1))
        assert np.array_equal(pp.pow_zn(Z), oracle_a.g_mul(1, B, Z))
        pp.clear()


def _load_bls(name):
    raw = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    assert raw[:8] == b"PBCBLS01"
    t, n, hlen, l1, l2, lz = struct.unpack("<6I", raw[8:32])
    a = np.frombuffer(raw, np.uint8)
    off, out = 32, {}
    for nm, cnt, ln in (("digests", n, hlen), ("h", n, l1), ("sig", n, l1), ("g", 1, l2), ("pk", 1, l2), ("sk", 1, lz)):
        out[nm] = a[off:off + cnt * ln].reshape(cnt, ln).copy()
        off += cnt * ln
    return out


@pytest.mark.parametrize("key,name", [("a", "a_bls64.bin"), ("d", "d159_bls32.bin"), ("f", "f_bls32.bin")])
def test_bls_flow_device_resident_matches_the_reference_run(hips, key, name):
    """example/bls.c:41-117 as a batch, every step a _dev call on buffers that never leave the device: hash the digests
    (== the reference's h_i), sign with the secret key (== its sig_i), derive the public key from g with a fixed-base table
    (== its pk), verify all signatures at once by random linear combination in 16-term products, and see a forged
    signature sink exactly its batch.  The fixture is the reference's own run of the flow (ref_tool bls)."""
    import torch
    H = hips[key]
    b = _load_bls(name)
    n, hlen = b["digests"].shape
    L1, L2, LT, LZ = H.length_in_bytes_G1, H.length_in_bytes_G2, H.length_in_bytes_GT, H.length_in_bytes_Zr
    r = _order(key)
    s = torch.cuda.current_stream().cuda_stream
    dig = torch.from_numpy(b["digests"]).cuda()
    sk = torch.from_numpy(np.tile(b["sk"], (n, 1))).cuda()
    h = torch.empty(n, L1, dtype=torch.uint8, device="cuda")
    sig = torch.empty(n, L1, dtype=torch.uint8, device="cuda")
    H.element_from_hash_dev(1, h.data_ptr(), dig.data_ptr(), hlen, n, s)
    H.element_mul_zn_dev(1, sig.data_ptr(), h.data_ptr(), sk.data_ptr(), n, s)
    gpp = H.element_pp_init(2, b["g"][0])                       # key generation: pk = g^sk with the generator preprocessed
    pk = torch.empty(1, L2, dtype=torch.uint8, device="cuda")
    gpp.pow_zn_dev(pk.data_ptr(), sk.data_ptr(), 1, s)
    torch.cuda.synchronize()
    gpp.clear()
    assert np.array_equal(h.cpu().numpy(), b["h"]) and np.array_equal(sig.cpu().numpy(), b["sig"])
    assert np.array_equal(pk.cpu().numpy(), b["pk"])
    rng = np.random.default_rng(6)
    rj = [int.from_bytes(rng.bytes(8), "big") | 1 for _ in range(n)]
    zp = torch.from_numpy(np.stack([_be(x, LZ) for x in rj])).cuda()
    zn = torch.from_numpy(np.stack([_be(r - x, LZ) for x in rj])).cuda()
    m = 8
    T2 = torch.stack([pk[0], torch.from_numpy(b["g"][0]).cuda()]).unsqueeze(0).expand(n, 2, L2).contiguous()
    one = np.zeros(LT, np.uint8)
    one[H.length_in_bytes_Fq - 1] = 1                           # GT's 1: the first coordinate is 1, the rest 0

    def verify(sigs):
        A = torch.empty(n, L1, dtype=torch.uint8, device="cuda")
        B = torch.empty(n, L1, dtype=torch.uint8, device="cuda")
        H.element_mul_zn_dev(1, A.data_ptr(), h.data_ptr(), zp.data_ptr(), n, s)
        H.element_mul_zn_dev(1, B.data_ptr(), sigs.data_ptr(), zn.data_ptr(), n, s)
        T1 = torch.stack([A, B], dim=1).contiguous()
        out = torch.empty(n // m, LT, dtype=torch.uint8, device="cuda")
        H.element_prod_pairing_dev(out.data_ptr(), T1.data_ptr(), T2.data_ptr(), n // m, 2 * m, s)
        torch.cuda.synchronize()
        return (out.cpu().numpy() == one[None]).all(axis=1)
    assert verify(sig).all()
    forged = sig.clone()
    forged[m + 2] = sig[m + 3]
    ok = verify(forged)
    assert not ok[1] and ok[0] and ok[2:].all()


def test_group_host_forms_pinned_in_place_and_over_a_device_set(hips):
    """Host forms of element_mul_zn / GT powers: page-locked buffers (in place for type a), pageable ones (staged chunk
    buffers the object keeps), a device set of three positions with more chunks than positions, out == in."""
    import pbc_amd
    L = pbc_amd.lib()
    v = golden("a_chain1024.vec")
    n = 5000
    Z, zl = _scalars("a", n, 12)
    pts = np.ascontiguousarray(v.g1[(np.arange(n) * 5) % v.n])
    gts = np.ascontiguousarray(v.gt[(np.arange(n) * 5) % v.n])
    want, wantT = hips["a"].element_mul_zn(1, pts, Z), hips["a"].element_pow_zn_GT(gts, Z)
    H = pbc_amd.Pairing(_param("a") + "hip_host_chunk 700\n")
    H.use_devices([0, 0, 0])
    assert np.array_equal(H.element_mul_zn(1, pts, Z), want)
    bufs = []

    def pinned(a):
        p = ctypes.c_void_p()
        assert L.pbc_hip_host_alloc(ctypes.byref(p), a.size) == 0
        bufs.append(p)
        arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(a.size,))
        arr[:] = a.reshape(-1)
        return p, arr
    pp, ap = pinned(pts)
    pz, az = pinned(Z)
    pt, at = pinned(gts)
    po, ao = pinned(np.zeros_like(pts))
    for _ in range(2):
        ao[:] = 0
        assert L.pbc_hip_element_mul_zn_batch(H._h, 1, po, pp, pz, n) == 0, pbc_amd._err()
        assert np.array_equal(ao.reshape(n, -1), want)
    assert L.pbc_hip_element_mul_zn_batch(H._h, 1, pp, pp, pz, n) == 0          # out == in
    assert np.array_equal(ap.reshape(n, -1), want)
    assert L.pbc_hip_element_pow_zn_GT_batch(H._h, pt, pt, pz, n) == 0
    assert np.array_equal(at.reshape(n, -1), wantT)
    for p in bufs:
        L.pbc_hip_host_free(p)
    H.clear()


@pytest.mark.parametrize("key,name", [("d", "d_rand32.vec"), ("f", "f_rand16.vec"), ("g149", "g149_rand16.vec")])
def test_text_forms_check_the_twist_equation(hips, key, name):
    """element_set_str / element_snprint on G2 of types d, f, g (curves over F_q^d / F_q^2): a text or record off the twist
    becomes / prints as O and set_str returns 0, as curve_set_str and curve_from_bytes do (ecc/curve.c:555-623); points
    of the twist round-trip."""
    H = hips[key]
    v = golden(name)
    text, full = H.element_snprint(2, v.g2[0])
    assert text.startswith("[[") and full == len(text)
    rec, used = H.element_set_str(2, text)
    assert used == len(text) and np.array_equal(rec, v.g2[0])
    bad = v.g2[0].copy()
    bad[-1] ^= 1
    assert H.element_snprint(2, bad)[0] == "O"
    i = text.rindex("]]")
    last = text[:i].rsplit(" ", 1)
    off = last[0] + " " + str(int(last[1]) + 1) + text[i:]
    rec, used = H.element_set_str(2, off)
    assert used == 0 and not rec.any()
