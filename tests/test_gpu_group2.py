"""GPU tests (-m gpu) of round 5's group surface through the C-ABI: element_add / sub / neg / double on G1 / G2, Z_r
arithmetic and element_from_hash on Zr, element_pow2_zn / element_pow3_zn on G1 / G2 / GT -- against the reference's
fixtures (tests/golden/*.rec), the oracle on larger fresh batches, the algebra (Shamir's trick = the sum of the separate
ladders), the _dev / stream forms, and the flows of example/zss.c and example/hess.c as device-resident batches."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, golden, _param, PARAM_OF, param_value

pytestmark = pytest.mark.gpu


def rec(name):
    return oracle.Rec(os.path.join(GOLDEN, name)).arrays


def _be(x, n):
    return np.frombuffer(int(x).to_bytes(n, "big"), np.uint8)


def _order(key):
    p = PARAM_OF.get(key, key)
    try:
        return param_value(p, "r")
    except KeyError:
        return param_value(p, "n")


GOPS = [("a", 1), ("a1", 1), ("e", 1), ("d159", 1), ("d159", 2), ("f", 1), ("f", 2), ("g149", 2), ("d201", 2), ("f_256", 2)]
ZROPS = ["a", "a1", "d159", "d224", "f", "f_256", "g149"]
POW23 = [("a", 1), ("a", 3), ("e", 1), ("d159", 1), ("d159", 2), ("d159", 3), ("f", 1), ("f", 2), ("f", 3), ("g149", 3), ("a1", 3)]
KEY = {"d159": "d"}


@pytest.mark.parametrize("pname,group", GOPS)
def test_group_law_matches_reference(hips, pname, group):
    """the reference's element_add / element_sub / element_neg / element_double, including B = A, B = -A, O operands and
    points of the whole curve; off-curve records are O"""
    H = hips[KEY.get(pname, pname)]
    A, B, ADD, SUB, NEG, DBL = rec("%s_gops%d.rec" % (pname, group))
    assert np.array_equal(H.element_group_op("add", group, A, B), ADD)
    assert np.array_equal(H.element_group_op("sub", group, A, B), SUB)
    assert np.array_equal(H.element_group_op("neg", group, A), NEG)
    assert np.array_equal(H.element_group_op("double", group, A), DBL)
    bad = A.copy()
    bad[:, -1] ^= 1
    assert np.array_equal(H.element_group_op("add", group, bad[:6], B[:6]), B[:6])
    assert not H.element_group_op("double", group, bad[:6]).any()


@pytest.mark.parametrize("pname", ZROPS)
def test_zr_arithmetic_matches_reference(hips, pname):
    H = hips[KEY.get(pname, pname)]
    R = rec(pname + "_zrops.rec")
    A, B = R[0], R[1]
    for what, idx in (("add", 2), ("sub", 3), ("mul", 4), ("invert", 5), ("neg", 6), ("double", 7), ("halve", 8), ("div", 9)):
        binary = what in ("add", "sub", "mul", "div")
        assert np.array_equal(H.zr_op(what, A, B if binary else None), R[idx]), what
    assert np.array_equal(H.zr_from_hash(R[10]), R[11])


@pytest.mark.parametrize("pname,group", POW23)
def test_multi_exponentiation_matches_reference(hips, pname, group):
    import pbc_amd
    H = hips[KEY.get(pname, pname)]
    A1, A2, A3, N1, N2, N3, P2, P3 = rec("%s_pow23g%d.rec" % (pname, group))
    assert np.array_equal(H.element_pow_multi(group, [A1, A2], [N1, N2]), P2)
    assert np.array_equal(H.element_pow_multi(group, [A1, A2, A3], [N1, N2, N3]), P3)
    # the other route: the joint ladders (Shamir's trick: fast pass + the complete routine for reported lanes; GT: the
    # generic table ladder), selected by "hip_group_slow 1"
    S = pbc_amd.Pairing(_param(PARAM_OF.get(KEY.get(pname, pname), pname)) + "hip_group_slow 1\n")
    assert np.array_equal(S.element_pow_multi(group, [A1, A2], [N1, N2]), P2)
    assert np.array_equal(S.element_pow_multi(group, [A1, A2, A3], [N1, N2, N3]), P3)
    S.clear()


@pytest.mark.parametrize("key,name", [("a", "a_chain1024.vec"), ("d", "d_chain256.vec"), ("f", "f_chain128.vec")])
def test_group_law_and_zr_on_fresh_batches_vs_oracle(hips, oracles, key, name):
    """3000 units: G1 sums / differences of multiples of the fixture's points against the oracle's affine law, Z_r against
    the oracle's F_p routines on r; every group: a + b - b = a, 2a = a + a, -(-a) = a; ragged sizes"""
    H, O = hips[key], oracles[key]
    v = golden(name)
    n = 3000
    rng = np.random.default_rng(17)
    i, j = rng.integers(0, v.n, n), rng.integers(0, v.n, n)
    j[::7] = i[::7]                                       # equal operands: the tangent case inside a batch
    A, B = v.g1[i], v.g1[j]
    m = 256
    assert np.array_equal(H.element_group_op("add", 1, A, B)[:m], O.g1_op(0, A[:m], B[:m]))
    assert np.array_equal(H.element_group_op("sub", 1, A, B)[:m], O.g1_op(1, A[:m], B[:m]))
    for group, X, Y in ((1, A, B), (2, v.g2[i], v.g2[j])):
        s = H.element_group_op("add", group, X, Y)
        assert np.array_equal(H.element_group_op("sub", group, s, Y), X)
        assert np.array_equal(H.element_group_op("double", group, X), H.element_group_op("add", group, X, X))
        assert np.array_equal(H.element_group_op("neg", group, H.element_group_op("neg", group, X)), X)
        d = H.element_group_op("sub", group, X, X)
        assert not d.any()                               # O is the zero record
    r = _order(key)
    zl = H.length_in_bytes_Zr
    a = np.stack([_be(int.from_bytes(rng.bytes(zl), "big") % r or 1, zl) for _ in range(n)])
    b = np.stack([_be(int.from_bytes(rng.bytes(zl), "big") % r or 1, zl) for _ in range(n)])
    for what, op in (("mul", 0), ("add", 1), ("sub", 2), ("invert", 3), ("neg", 4), ("halve", 5), ("double", 6), ("div", 7)):
        binary = what in ("add", "sub", "mul", "div")
        got = H.zr_op(what, a, b if binary else None)
        assert np.array_equal(got[:m], O.zr_op(op, a[:m], b[:m] if binary else None)), what
    assert np.array_equal(H.zr_op("mul", H.zr_op("invert", a), a), np.tile(_be(1, zl), (n, 1)))
    dig = rng.integers(0, 256, (n, 37), dtype=np.uint8)
    assert np.array_equal(H.zr_from_hash(dig)[:m], O.zr_from_hash(dig[:m], zl))


@pytest.mark.parametrize("key,name", [("a", "a_chain1024.vec"), ("d", "d_chain256.vec"), ("f", "f_chain128.vec"), ("g149", "g149_chain64.vec")])
def test_shamir_trick_equals_the_separate_ladders(hips, key, name):
    """[n1] a1 + [n2] a2 (+ [n3] a3) from one joint table ladder ("hip_group_slow 1": the fast pass, and the complete
    routine for the lanes it reports -- equal bases, scalars 0 and >= r are in the batch) = the sum of element_mul_zn
    results = the default route (composition); GT likewise with products of element_pow_zn"""
    import pbc_amd
    D = hips[key]
    H = pbc_amd.Pairing(_param(PARAM_OF.get(key, key)) + "hip_group_slow 1\n")
    v = golden(name)
    n = 700
    rng = np.random.default_rng(23)
    r = _order(key)
    zl = H.length_in_bytes_Zr
    idx = [rng.integers(0, v.n, n) for _ in range(3)]
    idx[1][::5] = idx[0][::5]                            # equal bases
    Z = []
    for t in range(3):
        ks = [int.from_bytes(rng.bytes(zl), "big") % r for _ in range(n)]
        for q, k in enumerate([0, 1, r - 1, (1 << (8 * zl)) - 1, 2]):
            ks[(q * 3 + t) % n] = k
        Z.append(np.stack([_be(k, zl) for k in ks]))
    for group, src in ((1, v.g1), (2, v.g2)):
        X = [src[idx[t]] for t in range(3)]
        M = [H.element_mul_zn(group, X[t], Z[t]) for t in range(3)]
        s2 = H.element_group_op("add", group, M[0], M[1])
        assert np.array_equal(H.element_pow_multi(group, X[:2], Z[:2]), s2)
        s3 = H.element_group_op("add", group, s2, M[2])
        assert np.array_equal(H.element_pow_multi(group, X, Z), s3)
        assert np.array_equal(D.element_pow_multi(group, X[:2], Z[:2]), s2)
        assert np.array_equal(D.element_pow_multi(group, X, Z), s3)
    m = 200
    X = [v.gt[idx[t][:m]] for t in range(3)]
    Pw = [H.element_pow_zn_GT(X[t], Z[t][:m]) for t in range(3)]
    p2 = H.element_mul_GT(Pw[0], Pw[1])
    assert np.array_equal(H.element_pow_multi(3, X[:2], [z[:m] for z in Z[:2]]), p2)
    p3 = H.element_mul_GT(p2, Pw[2])
    assert np.array_equal(H.element_pow_multi(3, X, [z[:m] for z in Z]), p3)
    assert np.array_equal(D.element_pow_multi(3, X[:2], [z[:m] for z in Z[:2]]), p2)
    assert np.array_equal(D.element_pow_multi(3, X, [z[:m] for z in Z]), p3)
    H.clear()


def test_zss_signatures_as_a_device_resident_batch(hips):
    """example/zss.c:28-58 for n messages at once, every step a _dev call on torch buffers (one stream, nothing leaves the
    device until the comparison): S = [1 / (H(m) + x)] P,  e([H(m)] P + Ppub, S) = e(P, P).  A forged signature fails."""
    import torch
    H = hips["a"]
    n = 2500
    rng = np.random.default_rng(31)
    r = _order("a")
    zl, l1, lt = H.length_in_bytes_Zr, H.length_in_bytes_G1, H.length_in_bytes_GT
    P = H.element_from_hash(1, rng.integers(0, 256, (1, 32), dtype=np.uint8))
    x = _be(int.from_bytes(rng.bytes(zl), "big") % r, zl)[None, :]
    Ppub = H.element_mul_zn(1, P, x)
    msgs = rng.integers(0, 256, (n, 24), dtype=np.uint8)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dP, dPpub, dx, dm = dev(np.tile(P, (n, 1))), dev(np.tile(Ppub, (n, 1))), dev(np.tile(x, (n, 1))), dev(msgs)
    h = torch.empty(n, zl, dtype=torch.uint8, device="cuda")
    t1 = torch.empty_like(h)
    S = torch.empty(n, l1, dtype=torch.uint8, device="cuda")
    t2 = torch.empty_like(S)
    t3 = torch.empty(n, lt, dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    s = st.cuda_stream
    H.zr_from_hash_dev(h.data_ptr(), dm.data_ptr(), 24, n, s)                    # SIGN
    H.zr_op_dev("add", t1.data_ptr(), h.data_ptr(), dx.data_ptr(), n, s)
    H.zr_op_dev("invert", t1.data_ptr(), t1.data_ptr(), 0, n, s)
    H.element_mul_zn_dev(1, S.data_ptr(), dP.data_ptr(), t1.data_ptr(), n, s)
    H.element_mul_zn_dev(1, t2.data_ptr(), dP.data_ptr(), h.data_ptr(), n, s)    # VERIFY
    H.element_group_op_dev("add", 1, t2.data_ptr(), t2.data_ptr(), dPpub.data_ptr(), n, s)
    H.element_pairing_dev(t3.data_ptr(), t2.data_ptr(), S.data_ptr(), n, s)
    st.synchronize()
    want = H.element_pairing(P, P)
    got = t3.cpu().numpy()
    assert (got == want).all()
    # the host forms give the same signatures; a signature under another key does not verify
    hh = H.zr_from_hash(msgs)
    assert np.array_equal(hh, h.cpu().numpy())
    sig = H.element_mul_zn(1, np.tile(P, (n, 1)), H.zr_op("invert", H.zr_op("add", hh, np.tile(x, (n, 1)))))
    assert np.array_equal(sig, S.cpu().numpy())
    forged = H.element_group_op("double", 1, sig[:8])
    assert not (H.element_pairing(t2.cpu().numpy()[:8], forged) == want).all(axis=1).any()


def test_hess_signatures_as_a_device_resident_batch(hips):
    """example/hess.c:44-86 for n messages: r = e(P1, P)^k, v = H(m) to_mpz(r) in Z_r, u = [v] Did + [k] P1 (ONE
    element_pow2_zn); verify e(u, P) e(Qid, -Ppub)^v = r.  The one host step is the reference's own (element_to_mpz of a GT
    element: the first coordinate as an integer, arith/fieldquadratic.c fq_to_mpz)."""
    import torch
    H = hips["a"]
    n = 1200
    rng = np.random.default_rng(37)
    rr = _order("a")
    zl, l1, lt = H.length_in_bytes_Zr, H.length_in_bytes_G1, H.length_in_bytes_GT
    hp = H.element_from_hash(1, rng.integers(0, 256, (3, 32), dtype=np.uint8))
    P, Qid, P1 = hp[0:1], hp[1:2], hp[2:3]
    s = _be(int.from_bytes(rng.bytes(zl), "big") % rr, zl)[None, :]
    Ppub, Did = H.element_mul_zn(1, P, s), H.element_mul_zn(1, Qid, s)
    ks = np.stack([_be(int.from_bytes(rng.bytes(zl), "big") % rr, zl) for _ in range(n)])
    msgs = rng.integers(0, 256, (n, 19), dtype=np.uint8)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    til = lambda a: np.tile(a, (n, 1))
    dP, dP1, dDid, dQid, dk, dm = dev(til(P)), dev(til(P1)), dev(til(Did)), dev(til(Qid)), dev(ks), dev(msgs)
    t1 = torch.empty(n, lt, dtype=torch.uint8, device="cuda")
    r = torch.empty_like(t1)
    H.element_pairing_dev(t1.data_ptr(), dP1.data_ptr(), dP.data_ptr(), n, 0)
    H.element_pow_zn_GT_dev(r.data_ptr(), t1.data_ptr(), dk.data_ptr(), n, 0)
    t3 = torch.empty(n, zl, dtype=torch.uint8, device="cuda")
    H.zr_from_hash_dev(t3.data_ptr(), dm.data_ptr(), 19, n, 0)
    torch.cuda.synchronize()
    to_zr = lambda gt: np.stack([_be(int.from_bytes(row[:lt // 2].tobytes(), "big") % rr, zl) for row in gt])
    dt2 = dev(to_zr(r.cpu().numpy()))
    v = torch.empty_like(t3)
    H.zr_op_dev("mul", v.data_ptr(), t3.data_ptr(), dt2.data_ptr(), n, 0)
    u = torch.empty(n, l1, dtype=torch.uint8, device="cuda")
    H.element_pow_multi_dev(1, u.data_ptr(), [dDid.data_ptr(), dP1.data_ptr()], [v.data_ptr(), dk.data_ptr()], n, 0)
    # VERIFY
    nPpub = dev(til(H.element_group_op("neg", 1, Ppub)))
    t6, t7 = torch.empty_like(t1), torch.empty_like(t1)
    H.element_pairing_dev(t6.data_ptr(), u.data_ptr(), dP.data_ptr(), n, 0)
    H.element_pairing_dev(t7.data_ptr(), dQid.data_ptr(), nPpub.data_ptr(), n, 0)
    H.element_pow_zn_GT_dev(t7.data_ptr(), t7.data_ptr(), v.data_ptr(), n, 0)
    H.element_mul_GT_dev(t6.data_ptr(), t6.data_ptr(), t7.data_ptr(), n, 0)
    torch.cuda.synchronize()
    assert torch.equal(t6, r)
    dt8 = dev(to_zr(t6.cpu().numpy()))
    t8 = torch.empty_like(v)
    H.zr_op_dev("mul", t8.data_ptr(), t3.data_ptr(), dt8.data_ptr(), n, 0)
    torch.cuda.synchronize()
    assert torch.equal(t8, v)
    # the pow2 form of u equals the two ladders and the sum of example/hess.c:63-65
    t4 = H.element_mul_zn(1, til(Did), v.cpu().numpy())
    t5 = H.element_mul_zn(1, til(P1), ks)
    assert np.array_equal(H.element_group_op("add", 1, t4, t5), u.cpu().numpy())


def test_new_entry_points_on_a_device_set(hips):
    """host-buffer forms over a device set (the one GPU listed three times: three workers, ragged shares): same bytes as
    the single-device call"""
    import pbc_amd
    H = hips["d"]
    v = golden("d_chain256.vec")
    n = 4099
    rng = np.random.default_rng(41)
    i, j = rng.integers(0, v.n, n), rng.integers(0, v.n, n)
    r = _order("d")
    zl = H.length_in_bytes_Zr
    Z1 = np.stack([_be(int.from_bytes(rng.bytes(zl), "big") % r, zl) for _ in range(n)])
    Z2 = np.stack([_be(int.from_bytes(rng.bytes(zl), "big") % r, zl) for _ in range(n)])
    want_add = H.element_group_op("add", 2, v.g2[i], v.g2[j])
    want_pow = H.element_pow_multi(1, [v.g1[i], v.g1[j]], [Z1, Z2])
    want_zr = H.zr_op("mul", Z1, Z2)
    P = pbc_amd.Pairing(_param("d159"))
    P.use_devices([0, 0, 0])
    assert np.array_equal(P.element_group_op("add", 2, v.g2[i], v.g2[j]), want_add)
    assert np.array_equal(P.element_pow_multi(1, [v.g1[i], v.g1[j]], [Z1, Z2]), want_pow)
    assert np.array_equal(P.zr_op("mul", Z1, Z2), want_zr)
    P.clear()


def test_fixed_base_table_with_a_device_set_and_twist_records_with_unreduced_coordinates(hips):
    """(1) element_pp_pow_zn on an object that has a device set runs on the table's device instead of failing;
    (2) a G2 record of a twist whose coordinate is written as x + q (still below 2^(8 len)) is the same point: it prints
    as the canonical record does, and parses back to it."""
    import pbc_amd
    v = golden("d_chain256.vec")
    P = pbc_amd.Pairing(_param("d159"))
    pp = P.element_pp_init(2, v.g2[3])
    r = _order("d")
    zl = P.length_in_bytes_Zr
    rng = np.random.default_rng(43)
    Z = np.stack([_be(int.from_bytes(rng.bytes(zl), "big") % r, zl) for _ in range(300)])
    want = pp.pow_zn(Z)
    assert np.array_equal(want, P.element_mul_zn(2, np.tile(v.g2[3], (300, 1)), Z))
    P.use_devices([0, 0])
    assert np.array_equal(pp.pow_zn(Z), want)
    P.use_devices([])
    q = param_value("d159", "q")
    fb = P.length_in_bytes_Fq
    rec = v.g2[5].copy()
    x0 = int.from_bytes(rec[:fb].tobytes(), "big")
    assert x0 + q < 1 << (8 * fb)
    big = rec.copy()
    big[:fb] = _be(x0 + q, fb)
    text, _ = P.element_snprint(2, rec)
    assert text != "O" and P.element_snprint(2, big)[0] == text
    back, used = P.element_set_str(2, text)
    assert used == len(text) and np.array_equal(back, rec)
    off = rec.copy()
    off[-1] ^= 1
    assert P.element_snprint(2, off)[0] == "O"
    pp.clear()
    P.clear()


@pytest.mark.parametrize("key,name", [("a", "a_chain1024.vec"), ("d", "d_chain256.vec"), ("f", "f_chain128.vec")])
def test_multi_exponentiation_with_the_result_over_an_input(hips, key, name):
    """element_pow2_zn / element_pow3_zn with x aliased to each base in turn (the reference lets x be any of them,
    arith/field.c:153-241; ADVICE r5: the composition route wrote [n1] a1 into out before reading a2): the _dev forms on
    both routes, G1 and GT, against the result computed into a separate buffer"""
    import pbc_amd
    import torch
    v = golden(name)
    n = 300
    rng = np.random.default_rng(61)
    r = _order(key)
    routes = [hips[key], pbc_amd.Pairing(_param(PARAM_OF.get(key, key)) + "hip_group_slow 1\n")]
    for H in routes:
        zl = H.length_in_bytes_Zr
        Z = [torch.from_numpy(np.stack([_be(int.from_bytes(rng.bytes(zl), "big") % r, zl) for _ in range(n)])).cuda() for _ in range(3)]
        for group, src in ((1, v.g1), (3, v.gt)):
            X = [np.ascontiguousarray(src[rng.integers(0, v.n, n)]) for _ in range(3)]
            for k in (2, 3):
                want = H.element_pow_multi(group, X[:k], [z.cpu().numpy() for z in Z[:k]])
                for alias in range(k):
                    D = [torch.from_numpy(x).cuda() for x in X[:k]]
                    H.element_pow_multi_dev(group, D[alias].data_ptr(), [d.data_ptr() for d in D], [z.data_ptr() for z in Z[:k]], n, 0)
                    torch.cuda.synchronize()
                    assert np.array_equal(D[alias].cpu().numpy(), want), (group, k, alias)
    routes[1].clear()


def test_joint_ladder_of_type_a_equals_the_composition(hips):
    """element_pow2_zn / element_pow3_zn on a.param G1 / G2 run ONE limb-form ladder for all bases (group_al.cuh gmulk_lane:
    four doublings + k additions per window); "hip_multi_compose 1" keeps the single-base ladders + additions.  Same bytes
    on 5000 units with the rows the joint ladder must hand to the complete routine: equal and opposite bases, a base off
    the curve (= O), scalars 0, 1, 2, r - 1, r, >= r, all ones; and with the result overlapping a base at an OFFSET (built in
    a temporary)."""
    import pbc_amd
    import torch
    H = hips["a"]
    C = pbc_amd.Pairing(_param("a") + "hip_multi_compose 1\n")
    v = golden("a_chain1024.vec")
    n = 5000
    rng = np.random.default_rng(71)
    r = _order("a")
    zl = H.length_in_bytes_Zr
    idx = [rng.integers(0, v.n, n) for _ in range(3)]
    idx[1][::9] = idx[0][::9]                                   # equal bases
    X = [np.ascontiguousarray(v.g1[idx[t]]) for t in range(3)]
    neg = H.element_group_op("neg", 1, X[0][3::9])
    X[2][3::9] = neg                                            # opposite bases (third and first)
    X[1][5::50] = neg[:len(X[1][5::50])]                        # ... and second and (another row's) first
    X[1][7::100, 5] ^= 1                                        # off the curve: O
    Z = []
    for t in range(3):
        ks = [int.from_bytes(rng.bytes(zl), "big") % r for _ in range(n)]
        for q, k in enumerate([0, 1, 2, r - 1, r, r + 1, (1 << (8 * zl)) - 1, 1 << (8 * zl - 1)]):
            for rep in range(6):
                ks[(q * 7 + t * 3 + rep * 131) % n] = k
        Z.append(np.stack([_be(k, zl) for k in ks]))
    Z[2][3::9] = Z[0][3::9]                                     # [k] P + ... + [k] (-P): the two cancel
    for group in (1, 2):
        for k in (2, 3):
            got = H.element_pow_multi(group, X[:k], Z[:k])
            assert np.array_equal(got, C.element_pow_multi(group, X[:k], Z[:k])), (group, k)
    # overlap at an offset: out = the first base's buffer shifted by one record
    m = 400
    lp = H.length_in_bytes_G1
    buf = torch.zeros((m + 1) * lp, dtype=torch.uint8, device="cuda")
    buf[lp:] = torch.from_numpy(X[0][:m].reshape(-1)).cuda()
    b2 = torch.from_numpy(X[1][:m]).cuda()
    z1, z2 = torch.from_numpy(Z[0][:m]).cuda(), torch.from_numpy(Z[1][:m]).cuda()
    want = C.element_pow_multi(1, [X[0][:m], X[1][:m]], [Z[0][:m], Z[1][:m]])
    H.element_pow_multi_dev(1, buf.data_ptr(), [buf.data_ptr() + lp, b2.data_ptr()], [z1.data_ptr(), z2.data_ptr()], m, 0)
    torch.cuda.synchronize()
    assert np.array_equal(buf[:m * lp].cpu().numpy().reshape(m, lp), want)
    C.clear()


@pytest.mark.parametrize("key,name", [("d", "d_chain256.vec"), ("f", "f_chain128.vec")])
def test_joint_ladder_of_the_five_word_fields_equals_the_composition(hips, key, name):
    """the same on G1 of d159.param / f.param (group_l5.cuh gmulk_lane): 3000 units with equal / opposite bases, a base off
    the curve, special scalars -- against the single-base ladders + additions ("hip_multi_compose 1")"""
    import pbc_amd
    H = hips[key]
    C = pbc_amd.Pairing(_param(PARAM_OF.get(key, key)) + "hip_multi_compose 1\n")
    v = golden(name)
    n = 3000
    rng = np.random.default_rng(73)
    r = _order(key)
    zl = H.length_in_bytes_Zr
    idx = [rng.integers(0, v.n, n) for _ in range(3)]
    idx[1][::9] = idx[0][::9]
    X = [np.ascontiguousarray(v.g1[idx[t]]) for t in range(3)]
    X[2][3::9] = H.element_group_op("neg", 1, X[0][3::9])
    X[1][7::100, 5] ^= 1
    Z = []
    for t in range(3):
        ks = [int.from_bytes(rng.bytes(zl), "big") % r for _ in range(n)]
        for q, k in enumerate([0, 1, 2, r - 1, r, r + 1, (1 << (8 * zl)) - 1, 1 << (8 * zl - 1)]):
            for rep in range(6):
                ks[(q * 7 + t * 3 + rep * 131) % n] = k
        Z.append(np.stack([_be(k, zl) for k in ks]))
    Z[2][3::9] = Z[0][3::9]
    for k in (2, 3):
        assert np.array_equal(H.element_pow_multi(1, X[:k], Z[:k]), C.element_pow_multi(1, X[:k], Z[:k])), k
    C.clear()


def test_two_threads_issue_on_one_stream(hips):
    """two host threads enqueue two-pass operations (element_mul_zn on a.param G1: fast kernel + the complete kernel for the
    lanes it flags, sharing a flags workspace keyed by (device, stream)) on the SAME stream of one object: each call's
    kernels are enqueued as a unit (pbc_hip.hip WsEnt::issue), so every result equals the single-threaded one.  The
    scalars 0 / r - 1 / >= r and equal-looking inputs make sure both passes have flagged lanes."""
    import threading
    import torch
    H = hips["a"]
    v = golden("a_chain1024.vec")
    n, rounds = 3000, 12
    r = _order("a")
    zl = H.length_in_bytes_Zr
    rng = np.random.default_rng(5)
    jobs = []
    for t in range(2):
        ks = [int.from_bytes(rng.bytes(zl), "big") % r for _ in range(n)]
        for q, k in enumerate([0, 1, r - 1, r, r + 1, (1 << (8 * zl)) - 1]):
            ks[(q * 7 + t) % n] = k
        z = np.stack([_be(k, zl) for k in ks])
        x = np.ascontiguousarray(v.g1[rng.integers(0, v.n, n)])
        x[5 + t::97] ^= 1                                # off-curve records: O
        jobs.append((x, z, H.element_mul_zn(1, x, z)))
    outs = [[torch.empty(n, H.length_in_bytes_G1, dtype=torch.uint8, device="cuda") for _ in range(rounds)] for _ in range(2)]
    dev = [(torch.from_numpy(x).cuda(), torch.from_numpy(z).cuda()) for x, z, _ in jobs]
    torch.cuda.synchronize()
    errs = []

    def worker(t):
        try:
            for i in range(rounds):
                H.element_mul_zn_dev(1, outs[t][i].data_ptr(), dev[t][0].data_ptr(), dev[t][1].data_ptr(), n, 0)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for t in range(2):
        for i in range(rounds):
            assert np.array_equal(outs[t][i].cpu().numpy(), jobs[t][2]), (t, i)
