"""CPU tests of the KERNEL SOURCE itself: pbc_amd/csrc/*.cuh compiled for the host
(tests/hostsim: same limb arithmetic, towers, Miller loops and final exponentiations as the
HIP kernels, one lane at a time) against the reference's golden vectors.  This is a debug
mirror for GPU-less bring-up, not a product path: libpbc_hip.so never contains it."""
import ctypes

import numpy as np
import pytest

from conftest import G2_HASH, G2_COMPRESS, G2_XONLY, XONLY, check_x_only, check_x_only_g2, golden, _param, PARAM_OF, OTHER, GENERIC_A, GENERIC_OTHER, GENERIC_F, FILES_OF, key_of, param_value

hostsim = pytest.importorskip("hostsim")


@pytest.fixture(scope="module")
def sims():
    class Lazy(dict):
        def __missing__(self, t):
            self[t] = hostsim.HostSim(_param(PARAM_OF.get(t, t)))
            return self[t]
    return Lazy()


@pytest.mark.parametrize("name,count", [
    ("a_kat.vec", 1), ("a_rand32.vec", 6), ("a_edge20.vec", 20), ("a_prod2x8.vec", 4), ("a_prod3x10_edge.vec", 10),
    ("d_rand32.vec", 6), ("d_edge20.vec", 20), ("d_prod16x4.vec", 2), ("d_prod3x10_edge.vec", 10),
    ("f_rand16.vec", 3), ("f_edge10.vec", 10), ("f_prod3x5_edge.vec", 5),
] + [(name, 2 if d in ("a1", "e") else 4) for d in OTHER for name in FILES_OF[d]] + [("g149_prod4x3.vec", 1)]
  + [(name, 2) for g in GENERIC_A + GENERIC_OTHER + GENERIC_F for name in (FILES_OF[g][0], FILES_OF[g][2])]
  # points of the whole curve, outside the order-r subgroup (ref_tool gen ... fullorder)
  + [("a_full12.vec", 3), ("a_prodfull3x4.vec", 2), ("d159_full12.vec", 4), ("d159_prodfull3x4.vec", 2), ("g149_full12.vec", 2),
     ("g149_prodfull3x4.vec", 1), ("f_full12.vec", 2), ("f_prodfull3x4.vec", 1), ("a1_full4.vec", 1), ("d224_full8.vec", 2)])
def test_kernel_source_on_host_matches_reference(sims, name, count):
    v = golden(name)
    n = min(count, v.n)
    out = sims[key_of(name)].prod_pairing(v.g1[:n * v.k], v.g2[:n * v.k], v.k)
    assert np.array_equal(out, v.gt[:n])


@pytest.mark.parametrize("t,q", [("a", None), ("d", 625852803282871856053922297323874661378036491717)]
                         + [(d, None) for d in OTHER + ["a_160_500", "a_224_768", "a1_200", "e_160_400", "f_256", "f_200", "f_r256"]])
def test_kernel_fq_ops_on_host(sims, oracles, t, q):
    if q is None:
        q = param_value(t, "p" if t.startswith("a1") else "q")
    nb = sims[t].len1 // 2
    rng = np.random.default_rng(2)
    xs = [int.from_bytes(rng.bytes(nb), "big") % q for _ in range(40)] + [1, q - 1, 2 ** (8 * nb) - 1]
    ys = [int.from_bytes(rng.bytes(nb), "big") % q for _ in range(40)] + [q - 1, q - 1, 2 ** (8 * nb) - 1]
    A = np.stack([np.frombuffer(x.to_bytes(nb, "big"), np.uint8) for x in xs])
    B = np.stack([np.frombuffer(y.to_bytes(nb, "big"), np.uint8) for y in ys])
    for op in range(7):
        assert np.array_equal(sims[t].fq_op(op, A, B), oracles[t].fq_op(op, A, B)), op


def test_type_a_products_with_one_product_per_lane_on_host():
    """Type a products run one TERM per lane by default (pairing_al.cuh: miller_record_lane + prod_finish_lane);
    "hip_prod_shared 1" keeps the kernel that shares the accumulator's squaring between the terms of a lane
    (a_prod_pairing_lane, the shape of a_pairings_affine) -- same bytes from both."""
    sim = hostsim.HostSim(_param("a") + "hip_prod_shared 1\n")
    for name, n in (("a_prod2x8.vec", 2), ("a_prod3x10_edge.vec", 10), ("a_prodfull3x4.vec", 2)):
        v = golden(name)
        n = min(n, v.n)
        assert np.array_equal(sim.prod_pairing(v.g1[:n * v.k], v.g2[:n * v.k], v.k), v.gt[:n]), name


def test_type_f_generic_hard_part_on_host():
    """f.param is a BN curve and takes the x-chain; the generic fixed-window power over
    (q^4-q^2+1)/r (any Type-F parameters; plain square-and-multiply) must give the same bytes."""
    v = golden("f_rand16.vec")
    sim = hostsim.HostSim(_param("f") + "hip_no_bn 1\n")
    assert np.array_equal(sim.prod_pairing(v.g1[:2], v.g2[:2], 1), v.gt[:2])


def test_type_f_reference_basis_path_on_host():
    """q = 3 mod 4 (f.param): the pairing runs in the i-basis of F_q^2 (beta -> -1); "hip_no_bm1 1" forces the parameter
    file's beta, the path of every q = 1 mod 4 parameter set.  The default object must really be on the i-basis path: it
    executes fewer multiply-adds."""
    v = golden("f_rand16.vec")
    sim = hostsim.HostSim(_param("f") + "hip_no_bm1 1\n")
    sim.macs(reset=True)
    assert np.array_equal(sim.prod_pairing(v.g1[:2], v.g2[:2], 1), v.gt[:2])
    general = sim.macs(reset=True)
    sim = hostsim.HostSim(_param("f"))
    sim.macs(reset=True)
    assert np.array_equal(sim.prod_pairing(v.g1[:2], v.g2[:2], 1), v.gt[:2])
    assert sim.macs(reset=True) < 0.95 * general


def test_type_d_word_form_point_arithmetic_on_host():
    """d159.param keeps the running point of its Miller loop in limb form; parameters whose q leaves the top limb
    nearly empty take the word-form step routines inside the same kernels: force them on d159.param
    ("hip_no_limb 1") -- single pairings, a product with edge cases, a preprocessed first argument."""
    sim = hostsim.HostSim(_param("d159") + "hip_no_limb 1\n")
    v = golden("d_rand32.vec")
    assert np.array_equal(sim.prod_pairing(v.g1[:3], v.g2[:3], 1), v.gt[:3])
    w = golden("d_prod3x10_edge.vec")
    assert np.array_equal(sim.prod_pairing(w.g1[:4 * w.k], w.g2[:4 * w.k], w.k), w.gt[:4])
    assert np.array_equal(sim.pp(v.g1[0], v.g2[:1]), v.gt[:1])


def test_pairing_pp_on_host(sims, oracles):
    """pairing_pp_init + pairing_pp_apply give element_pairing's bytes (a_param.c:149-220, :317-360)."""
    v = golden("a_chain1024.vec")
    P = v.g1[5]
    Q = v.g2[:6].copy()
    Q[2, 127] ^= 1                                         # off-curve second argument -> 1
    want = oracles["a"].pairing_batch(np.tile(P, (6, 1)), Q)
    assert np.array_equal(sims["a"].pp(P, Q), want)
    bad = P.copy(); bad[100] ^= 4                          # off-curve first argument -> all 1
    one = np.zeros(128, np.uint8); one[63] = 1
    assert np.array_equal(sims["a"].pp(bad, Q), np.tile(one, (6, 1)))


@pytest.mark.parametrize("t,name", [("d", "d_rand32.vec"), ("d201", "d201_rand12.vec"), ("g149", "g149_rand16.vec"),
                                    ("a1", "a1_rand6.vec"), ("a_160_256", "a_160_256_rand6.vec"),
                                    ("a_150_300_mm", "a_150_300_mm_rand6.vec")])
def test_pairing_pp_types_d_g_on_host(sims, oracles, t, name):
    """d_pairing_pp_init/apply (d_param.c:794-966), g_pairing_pp_init/apply (g_param.c:619-787): same bytes as
    element_pairing; off-curve arguments give the identity."""
    v = golden(name)
    P = v.g1[1]
    m = 2 if t == "a1" else 4
    Q = v.g2[:m].copy()
    Q[m - 1, -1] ^= 1
    want = oracles[t].pairing_batch(np.tile(P, (m, 1)), Q)
    assert np.array_equal(sims[t].pp(P, Q), want)
    bad = P.copy(); bad[3] ^= 4
    one = np.zeros(sims[t].lenT, np.uint8); one[sims[t].len1 // 2 - 1] = 1
    assert np.array_equal(sims[t].pp(bad, Q), np.tile(one, (m, 1)))


@pytest.mark.parametrize("t,name", [("a", "a_rand32.vec"), ("d", "d_rand32.vec"), ("f", "f_rand16.vec")]
                         + [(d, FILES_OF[d][0]) for d in OTHER + ["a_150_300_mm", "a1_200", "e_160_400", "f_256", "f_200", "f_r256"]])
def test_group_ops_on_host(sims, oracles, t, name):
    """element_mul_zn on G1, element_mul / element_pow_zn on GT vs the oracle."""
    v = golden(name)
    rng = np.random.default_rng(21)
    n = 2 if v.n < 6 else 3
    r = param_value(t, "n" if t.startswith("a1") else "r")
    zl = (r.bit_length() + 7) // 8          # pairing_length_in_bytes_Zr
    ks = [int.from_bytes(rng.bytes(zl), "big") % r for _ in range(n - 1)] + [1]
    Z = np.stack([np.frombuffer(k.to_bytes(zl, "big"), np.uint8) for k in ks])
    assert np.array_equal(sims[t].group(0, v.g1[:n], Z), oracles[t].g_mul(1, v.g1[:n], Z))
    assert np.array_equal(sims[t].group(1, v.gt[:n], v.gt[n:2 * n]), oracles[t].gt_mul(v.gt[:n], v.gt[n:2 * n]))
    assert np.array_equal(sims[t].group(2, v.gt[:n], Z), oracles[t].gt_pow(v.gt[:n], Z))


@pytest.mark.parametrize("name", ["a_hash32.vec", "a_hash13.vec", "a_hash100.vec", "d159_hash32.vec", "d201_hash32.vec",
                                  "d278027-190-181_hash32.vec", "f_hash32.vec", "g149_hash32.vec", "e_hash20.vec",
                                  "a1_hash20.vec"])
def test_from_hash_on_host(sims, name):
    """element_from_hash(G1) (ecc/curve.c:455-482): digest expansion, retry loop, square root,
    sign normalisation and cofactor multiplication vs the reference's outputs."""
    v = golden(name)
    n = min(v.n, 6)
    key = {"d159": "d", "d201": "d201", "d278027-190-181": "d278027-190-181"}.get(name.split("_")[0], key_of(name))
    got = sims[key].from_hash(v.g1[:n], v.len1)
    assert np.array_equal(got, v.gt[:n])


@pytest.mark.parametrize("key,name", [("d", "d159_g2mul6.vec"), ("d201", "d201_g2mul6.vec"), ("g149", "g149_g2mul6.vec"),
                                      ("f", "f_g2mul6.vec")])
def test_g2_scalar_multiplication_on_host(sims, key, name):
    """element_mul_zn on G2 of the asymmetric types: the twists over F_q^3 / F_q^5 (d, g) and F_q^2 (f)."""
    v = golden(name)
    assert np.array_equal(sims[key].g2_mul(v.g1, v.g2), v.gt)


@pytest.mark.parametrize("key,name", [("e", "e_g1mul3.vec"), ("d224", "d224_g1mul6.vec")])
def test_g1_scalar_multiplication_wide_fields_on_host(sims, key, name):
    v = golden(name)
    assert np.array_equal(sims[key].group(0, v.g1, v.g2), v.gt)


@pytest.mark.parametrize("key,name", [("a", "a_g1mulfull6.vec"), ("a", "a_g2mulfull6.vec"), ("d", "d159_g1mulfull6.vec"),
                                      ("g149", "g149_g1mulfull6.vec"), ("e", "e_g1mulfull6.vec")])
def test_scalar_multiplication_on_whole_curve_points_on_host(sims, key, name):
    """element_mul_zn on points outside the order-r subgroup (curve_from_bytes accepts them) vs the reference"""
    v = golden(name)
    n = 2 if key == "e" else v.n
    assert np.array_equal(sims[key].group(0, v.g1[:n], v.g2[:n]), v.gt[:n])


def test_type_a_with_a_1024_bit_field_on_host(sims):
    """pbc_param_init_a_gen(160, 1024), the reference's standard "a" generator call: 1024-bit q, cofactor of 864 bits
    (round 1 refused every cofactor above 768 bits at init).  Pairing, product and element_from_hash vs the reference."""
    S = sims["a_160_1024"]
    v, pr, h = golden("a_160_1024_rand4.vec"), golden("a_160_1024_prod3x3_edge.vec"), golden("a_160_1024_hash20.vec")
    assert np.array_equal(S.prod_pairing(v.g1[:1], v.g2[:1], 1), v.gt[:1])
    assert np.array_equal(S.prod_pairing(pr.g1[:3], pr.g2[:3], 3), pr.gt[:1])
    assert np.array_equal(S.from_hash(h.g1[:1], h.len1), h.gt[:1])


@pytest.mark.parametrize("key,name", [("a", "a_finalpow6.vec"), ("d", "d159_finalpow6.vec"), ("f", "f_finalpow6.vec"),
                                      ("g149", "g149_finalpow6.vec"), ("e", "e_finalpow3.vec"), ("a1", "a1_finalpow3.vec")])
def test_finalpow_on_host(sims, key, name):
    """pairing->finalpow through the kernels' own final-exponentiation routines vs the reference"""
    v = golden(name)
    n = 2 if key in ("e", "a1") else 3
    assert np.array_equal(sims[key].finalpow(v.g1[:n]), v.gt[:n])


@pytest.mark.parametrize("key,name", [("a", "a_finalpow6.vec"), ("d", "d159_finalpow6.vec"), ("g149", "g149_finalpow6.vec")])
def test_finalpow_when_the_easy_part_gives_one_on_host(sims, oracles, key, name):
    """elements of proper subfields (what element_from_hash produces on GT for the polymod towers): the first stage of
    the final exponentiation gives +-1, where the single-inversion formulas must not divide by the vanishing part"""
    v = golden(name)
    x = v.g1[:2].copy()
    lt = x.shape[1]
    if key == "a":
        x[:, lt // 2:] = 0
    else:
        d = 3 if key == "d" else 5
        fb = lt // (2 * d)
        for i in range(1, d):
            x[:, i * fb:(i + 1) * fb] = x[:, :fb]
            x[:, (d + i) * fb:(d + i + 1) * fb] = x[:, d * fb:(d + 1) * fb]
    assert np.array_equal(sims[key].finalpow(x), oracles[key].finalpow(x))


def test_group_law_is_complete_on_host(sims, oracles):
    """small-order points and scalars >= r: the double-and-add meets R = P (needs a doubling), R = -P and R = O.
    Type a: #E = q + 1 = h r with 12 | h, so the curve has points of order 2, 3, 4, 6."""
    S, O = sims["a"], oracles["a"]
    q, r = param_value("a", "q"), param_value("a", "r")
    v = golden("a_full12.vec")
    be = lambda x, n: np.frombuffer(int(x).to_bytes(n, "big"), np.uint8)
    pts = []
    for m in (2, 3, 4, 6, 12):                             # T -> [(q+1)/m] T has order dividing m
        e = np.tile(be((q + 1) // m, 64), (4, 1))
        pts.append(O.g_mul(1, v.g1[:4], e))
    P = np.concatenate(pts)                                # 20 points of small order (a few may be O)
    assert any(p.any() for p in P)
    for k in (0, 1, 2, 3, 4, 5, 6, 7, 11, 12, 13, r - 1, r, r + 1, r + 5, 2 ** 160 - 1):
        Z = np.tile(be(k, 20), (len(P), 1))
        assert np.array_equal(S.group(0, P, Z), O.g_mul(1, P, Z)), k
    # whole-curve points with scalars in [r, 2^160)
    Z = np.stack([be(r + 1000 * i + 7, 20) for i in range(6)])
    assert np.array_equal(S.group(0, v.g1[:6], Z), O.g_mul(1, v.g1[:6], Z))


COMPRESS = [("a", "a_compress12.vec"), ("d", "d159_compress12.vec"), ("d278027-190-181", "d278027-190-181_compress12.vec"),
            ("f", "f_compress12.vec"), ("g149", "g149_compress12.vec"), ("e", "e_compress4.vec")]


@pytest.mark.parametrize("key,name", COMPRESS)
def test_compressed_points_on_host(sims, key, name):
    """element_to_bytes_compressed / element_from_bytes_compressed (ecc/curve.c:762-815) vs the reference."""
    v = golden(name)                                       # file: x||y records, (no second input), x||s records
    n = min(v.n, 6)
    assert np.array_equal(sims[key].compress(0, v.g1[:n]), v.gt[:n])
    assert np.array_equal(sims[key].compress(1, v.gt[:n]), v.g1[:n])


def test_dense_wide_modulus_products_on_host():
    """worst case for the column accumulator of the 33-word product: a 2^1033 - c shaped odd modulus (every
    limb of q full) and operands with every limb full -- products and squares against Python integers."""
    n = ((1 << 1031) - 12345) | 1
    p = 4 * n - 1
    S = hostsim.HostSim("type a1\np %d\nn %d\nl 4\n" % (p, n))
    nb = (p.bit_length() + 7) // 8
    vals = [p - 1, p - 2, (1 << 1032) - 1, (1 << 1033) - 1 - (1 << 500), 1, 2]
    xs = [v % p for v in vals for _ in vals]
    ys = [v % p for _ in vals for v in vals]
    A = np.stack([np.frombuffer(x.to_bytes(nb, "big"), np.uint8) for x in xs])
    B = np.stack([np.frombuffer(y.to_bytes(nb, "big"), np.uint8) for y in ys])
    got = S.fq_op(0, A, B)
    want = np.stack([np.frombuffer((x * y % p).to_bytes(nb, "big"), np.uint8) for x, y in zip(xs, ys)])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("key,name", [("a", "a_rand32.vec"), ("d", "d_rand32.vec"), ("f", "f_rand16.vec"), ("g149", "g149_rand16.vec"),
                                      ("e_160_400", "e_160_400_rand6.vec"), ("a_150_300_mm", "a_150_300_mm_rand6.vec"),
                                      ("f_256", "f_256_rand4.vec"), ("d224", "d224_rand12.vec")])
def test_fresh_points_kernel_source_vs_oracle(sims, oracles, key, name):
    """inputs no fixture holds: random multiples [k]P of the fixture's G1 points (computed by the oracle)
    paired with its G2 points, through the kernel source on the host and through the oracle; plus
    e([k]P, Q) = e(P, Q)^k with the power taken by the kernel source."""
    v = golden(name)
    S, O = sims[key], oracles[key]
    n = 3
    r = param_value(key, "r")
    zl = (r.bit_length() + 7) // 8
    rng = np.random.default_rng(53)
    ks = [int.from_bytes(rng.bytes(zl), "big") % (r - 1) + 1 for _ in range(n)]
    Z = np.stack([np.frombuffer(k.to_bytes(zl, "big"), np.uint8) for k in ks])
    kP = O.g_mul(1, v.g1[:n], Z)
    got = S.prod_pairing(kP, v.g2[1:n + 1], 1)
    assert np.array_equal(got, O.pairing_batch(kP, v.g2[1:n + 1]))
    same = S.prod_pairing(kP, v.g2[:n], 1)
    assert np.array_equal(same, S.group(2, v.gt[:n], Z))


@pytest.mark.parametrize("key,name,exact", XONLY)
def test_x_only_points_on_host(sims, key, name, exact):
    """element_to_bytes_x_only / element_from_bytes_x_only (ecc/curve.c:821-836) vs the reference"""
    v = golden(name)
    S = sims[key]
    check_x_only(lambda p: S.compress(2, p), lambda x: S.compress(3, x), v, exact, 0 if exact else param_value(key, "q"))


@pytest.mark.parametrize("key,name", G2_HASH)
def test_from_hash_on_the_twists_on_host(sims, key, name):
    """element_from_hash on G2 of the asymmetric types (curve_from_hash over F_q^d / F_q^2, ecc/curve.c:455-482;
    polymod_from_hash poly.c:341-348, fq_from_hash fieldquadratic.c:311-316) vs the reference"""
    v = golden(name)                                       # file: digests, (nothing), points
    assert np.array_equal(sims[key].g2_points(0, v.g1, v.len1), v.gt)


@pytest.mark.parametrize("key,name", G2_COMPRESS)
def test_compressed_points_on_the_twists_on_host(sims, key, name):
    v = golden(name)                                       # file: x||y records, (nothing), x||s records
    n = v.n
    S = sims[key]
    assert np.array_equal(S.g2_points(1, v.g1[:n]), v.gt[:n])
    assert np.array_equal(S.g2_points(2, v.gt[:n]), v.g1[:n])
    flipped = v.gt[:n].copy()
    flipped[:, -1] ^= 1                                    # the other root
    neg = S.g2_points(2, flipped)
    assert np.array_equal(neg[:, :v.len1 // 2], v.g1[:n, :v.len1 // 2]) and not np.array_equal(neg, v.g1[:n])
    assert np.array_equal(S.g2_points(1, neg), flipped)


@pytest.mark.parametrize("key,name,exact", G2_XONLY)
def test_x_only_points_on_the_twists_on_host(sims, key, name, exact):
    """element_to_bytes_x_only / element_from_bytes_x_only on G2 of types f, d, g vs the reference"""
    v = golden(name)
    S = sims[key]
    check_x_only_g2(lambda p: S.g2_points(3, p), lambda x: S.g2_points(4, x), v, exact, param_value(key, "q"), S.len1 // 2)


@pytest.mark.parametrize("key,hlen", [("a", 20), ("a", 70), ("d", 32), ("f", 24), ("g149", 32), ("d224", 28), ("e_160_400", 20)])
def test_from_hash_and_point_formats_on_fresh_digests_vs_oracle(sims, oracles, key, hlen):
    """digests no fixture holds: element_from_hash, then compressed and x-only round trips of the hashed points,
    through the kernel source on the host and through the oracle (itself pinned on the reference's vectors)"""
    rng = np.random.default_rng(hlen * 131 + len(key))
    n = 16
    D = rng.integers(0, 256, (n, hlen), dtype=np.uint8)
    S, O = sims[key], oracles[key]
    pts = S.from_hash(D, hlen)
    assert np.array_equal(pts, O.from_hash(D))
    c = S.compress(0, pts)
    assert np.array_equal(c, O.point_format(0, pts))
    assert np.array_equal(S.compress(1, c), pts)
    x = S.compress(2, pts)
    back = S.compress(3, x)
    fb = S.len1 // 2
    assert np.array_equal(back[:, :fb], pts[:, :fb])
    # (x, y') lies on the curve with y' = +-y: its compressed form differs from c at most in the sign byte
    assert np.array_equal(S.compress(0, back)[:, :fb], c[:, :fb])


@pytest.mark.parametrize("key,hlen", [("d", 32), ("d224", 20), ("f", 32), ("f_256", 21), ("g149", 32)])
def test_twist_hashing_and_compression_on_fresh_digests_vs_oracle(sims, oracles, key, hlen):
    """element_from_hash and the compressed form on the G2 twists for digests no fixture holds"""
    rng = np.random.default_rng(hlen * 17 + len(key))
    D = rng.integers(0, 256, (12, hlen), dtype=np.uint8)
    S, O = sims[key], oracles[key]
    pts = S.g2_points(0, D, hlen)
    assert np.array_equal(pts, O.from_hash_g2(D))
    c = S.g2_points(1, pts)
    assert np.array_equal(c, O.point_format_g2(0, pts))
    assert np.array_equal(S.g2_points(2, c), pts)


def test_limb_form_type_a_subtraction_constants(sims):
    """pairing_al.cuh forms a - b as a + (K - b) limb by limb: the host-built K must be multiples of q whose limbs
    dominate every subtrahend the kernel pairs them with (AConst::ksub).  The kernel's own bound tracker -- worst-case
    limb sizes and values, asserted at every operation of the host mirror -- runs inside every type a test above."""
    S = sims["a"]
    S.L.hostsim_check_ksub.argtypes = [ctypes.c_void_p]
    assert S.L.hostsim_check_ksub(S.h) == 0


# ---- round 4: the fast ladders of the group operations (group_al.cuh, group_ops.cuh ec_mul_win_lane / ec_pp_* / gt_pp_*) ----
def _be(x, n):
    return np.frombuffer(int(x).to_bytes(n, "big"), np.uint8)


def _order(pname):
    from conftest import param_value
    try:
        return param_value(pname, "r")
    except KeyError:
        return param_value(pname, "n")


def test_limb_form_scalar_multiplication_and_gt_powers_type_a_on_host(sims, oracles):
    """GAL<16>::gmul_lane / gt_pow_lane (the kernels' source, with the worst-case bound tracker of the limb form armed):
    random and exceptional scalars, whole-curve points; lanes the ladder reports (result O, even scalar meeting -P) take
    the complete routine, as in the library.  Bytes against the oracle and the reference's vectors."""
    S, O = sims["a"], oracles["a"]
    v = golden("a_rand32.vec")
    r = _order("a")
    rng = np.random.default_rng(5)
    ks = [int.from_bytes(rng.bytes(20), "big") % r for _ in range(8)] + [0, 1, 2, 3, r - 1, r - 2, r, r + 1, 2**160 - 1, 2**160 - 2, 15, 16, 17, 2**159]
    Z = np.stack([_be(k, 20) for k in ks])
    P = v.g1[:len(ks)]
    S.fallbacks()
    assert np.array_equal(S.group(0, P, Z), O.g_mul(1, P, Z))
    assert S.fallbacks() == 3                                  # 0 P, r P and (r - 1) P = r P - P
    w = golden("a_g1mulfull6.vec")                             # points outside the order-r subgroup (reference vectors)
    assert np.array_equal(S.group(0, w.g1, w.g2), w.gt)
    g = v.gt[:len(ks)]
    assert np.array_equal(S.group(2, g, Z), O.gt_pow(g, Z))
    assert S.fallbacks() == 0                                  # pairing values have norm 1: the Lucas ladder serves them all
    A = rng.integers(0, 256, (4, 128), dtype=np.uint8)
    A[:, 0] = A[:, 64] = 0
    assert np.array_equal(S.group(2, A, Z[:4]), O.gt_pow(A, Z[:4]))
    assert S.fallbacks() == 4                                  # elements of other norm: the generic routine
    S.group_mode(True)
    try:
        assert np.array_equal(S.group(0, P, Z), O.g_mul(1, P, Z)) and S.fallbacks() == 0
    finally:
        S.group_mode(False)


@pytest.mark.parametrize("key,name", [("d", "d_rand32.vec"), ("f", "f_rand16.vec"), ("g149", "g149_rand16.vec"), ("e", "e_rand6.vec"),
                                      ("d201", "d201_rand12.vec"), ("f_256", "f_256_rand4.vec"), ("a1", "a1_rand6.vec")])
def test_windowed_ladders_on_host(sims, oracles, key, name):
    """ec_mul_win_lane over every field policy: G1 against the oracle, G2 on the twists against the complete ladder (which
    the reference's vectors pin, test_g2_scalar_multiplication_on_host)."""
    from conftest import PARAM_OF
    S, O = sims[key], oracles[key]
    v = golden(name)
    r = _order(PARAM_OF.get(key, key))
    zl = (r.bit_length() + 7) // 8
    rng = np.random.default_rng(11)
    ks = ([int.from_bytes(rng.bytes(zl), "big") % r for _ in range(3)] + [0, 1, 2, r - 1, r, (1 << (8 * zl)) - 1])[:v.n]
    Z = np.stack([_be(k, zl) for k in ks])
    n = len(ks)
    S.fallbacks()
    assert np.array_equal(S.group(0, v.g1[:n], Z), O.g_mul(1, v.g1[:n], Z))
    assert 1 <= S.fallbacks() <= 3
    if key not in ("e", "a1"):
        got = S.g2_mul(v.g2[:n], Z)
        S.group_mode(True)
        try:
            want = S.g2_mul(v.g2[:n], Z)
        finally:
            S.group_mode(False)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("key,name,group", [("d", "d_rand32.vec", 1), ("d", "d_rand32.vec", 3), ("f", "f_rand16.vec", 1), ("f", "f_rand16.vec", 2),
                                            ("f", "f_rand16.vec", 3), ("a", "a_rand32.vec", 3)])
def test_fixed_base_tables_on_host(sims, oracles, key, name, group):
    """element_pp_init + element_pp_pow_zn (ec_pp_entry_lane / ec_pp_pow_lane, gt_pp_*): the table as the library's
    kernels build it, powers against element_mul_zn / element_pow_zn of the same base."""
    from conftest import PARAM_OF
    S, O = sims[key], oracles[key]
    v = golden(name)
    r = _order(PARAM_OF.get(key, key))
    zl = (r.bit_length() + 7) // 8
    rng = np.random.default_rng(13)
    ks = [int.from_bytes(rng.bytes(zl), "big") % r for _ in range(3)] + [0, 1, 255, 256, r - 1]
    Z = np.stack([_be(k, zl) for k in ks])
    base = {1: v.g1[1], 2: v.g2[1], 3: v.gt[1]}[group]
    B = np.tile(base, (len(ks), 1))
    got = S.element_pp(group, base, Z, zl)
    if group == 2:
        S.group_mode(True)
        try:
            want = S.g2_mul(B, Z)
        finally:
            S.group_mode(False)
    else:
        want = O.gt_pow(B, Z) if group == 3 else O.g_mul(1, B, Z)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("key,pname,group", [("d", "d159", 1), ("f", "f", 1), ("f", "f", 3), ("a", "a", 3)])
def test_element_pp_reference_vectors_on_host(sims, key, pname, group):
    """the reference's element_pp_pow_zn outputs (ref_tool ppow) through the table routines of the kernel source"""
    v = golden("%s_pp%dpow12.vec" % (pname, group))
    assert np.array_equal(sims[key].element_pp(group, v.g1[0], v.g2, v.len2), v.gt)


@pytest.mark.parametrize("extra", ["", "hip_no_xs 1\n", "hip_no_limb 1\n", "hip_no_xs 1\nhip_no_limb 1\n", "hip_no_cyc 1\n"])
def test_type_f_sparse_xi_and_limb_form_steps_on_host(extra):
    """Round 4, f.param: the pairing kernels take F_q^12 in the basis X' = X / c with X'^6 = xi' = 4 + 2 i sparse (init_stage4:
    the 6th root c on the device, products by xi' as shifts and adds) and run the steps on E(F_q) in limb form under the
    worst-case bound tracker.  Every switch combination gives the reference's bytes; the default executes fewer
    multiply-adds than the parameter file's xi."""
    import hostsim
    from conftest import _param
    S = hostsim.HostSim(_param("f") + extra)
    v, w, e = golden("f_rand16.vec"), golden("f_prod3x5_edge.vec"), golden("f_edge10.vec")
    S.macs(reset=True)
    assert np.array_equal(S.prod_pairing(v.g1[:4], v.g2[:4], 1), v.gt[:4])
    macs = S.macs(reset=True) // 4
    assert np.array_equal(S.prod_pairing(w.g1, w.g2, w.k), w.gt)
    assert np.array_equal(S.prod_pairing(e.g1, e.g2, 1), e.gt)
    if "no_xs" in extra:
        assert macs > 1_950_000
    elif not extra:
        assert macs < 1_750_000


@pytest.mark.parametrize("extra", ["", "hip_no_xs 1\n", "hip_no_bm1 1\n"])
def test_type_f_gt_powers_by_cyclotomic_squarings_on_host(oracles, extra):
    """element_pow_zn on GT, f.param (group_ops.cuh f_gt_pow_cyc_lane): the element taken into the pairing kernels' basis,
    membership in the cyclotomic subgroup checked (a^(q^4) a = a^(q^2)), Granger-Scott squarings in the LDS area with
    regular signed 4-bit windows (inverse = conjugate); elements outside the subgroup go to the generic ladder, as in
    the library.  Random and exceptional scalars; bytes against the oracle."""
    import hostsim
    from conftest import _param
    S, O = hostsim.HostSim(_param("f") + extra), oracles["f"]
    v = golden("f_rand16.vec")
    r = _order("f")
    rng = np.random.default_rng(2)
    ks = [int.from_bytes(rng.bytes(20), "big") % r for _ in range(3)] + [0, 1, 2, r - 1, r, 2**160 - 1, 16]
    Z = np.stack([_be(k, 20) for k in ks])
    g = v.gt[:len(ks)]
    S.fallbacks()
    S.macs(reset=True)
    assert np.array_equal(S.group(2, g, Z), O.gt_pow(g, Z))
    fast = S.macs(reset=True) // len(ks)
    assert S.fallbacks() == 0
    A = rng.integers(0, 256, (2, 240), dtype=np.uint8)
    A[:, ::20] = 0
    A = np.concatenate([A, np.zeros((1, 240), np.uint8)])      # and 0, which has no inverse
    assert np.array_equal(S.group(2, A, Z[:3]), O.gt_pow(A, Z[:3]))
    assert S.fallbacks() == 3
    S.group_mode(True)
    try:
        S.macs(reset=True)
        assert np.array_equal(S.group(2, g, Z), O.gt_pow(g, Z))
        slow = S.macs(reset=True) // len(ks)
    finally:
        S.group_mode(False)
    if not extra:
        assert fast < 0.5 * slow, (fast, slow)


def test_type_a_one_pairing_per_wavefront_on_host(sims):
    """pairing_aw.cuh (the low-latency path for small batches): the lane recurrence of the wave-wide Montgomery product
    run lane by lane on the host, under AL's worst-case bound tracker plus the recurrence's own checks (column cleared
    by m, accumulators below 2^60, what a lane carries to the next step within 32 bits); the reference's vectors, the
    edge cases (invalid arguments, O, points of order two), a second 512-bit parameter set whose add step negates P."""
    import hostsim
    from conftest import _param, PARAM_OF
    S = sims["a"]
    for name in ("a_rand32.vec", "a_edge20.vec"):
        v = golden(name)
        m = min(v.n, 20)
        assert np.array_equal(S.pairing_wave(v.g1[:m], v.g2[:m]), v.gt[:m]), name
    S2 = hostsim.HostSim(_param(PARAM_OF.get("a_160_512_mm", "a_160_512_mm")))
    v = golden("a_160_512_mm_rand6.vec")
    assert np.array_equal(S2.pairing_wave(v.g1, v.g2), v.gt)
    with pytest.raises(RuntimeError):
        sims["d"].pairing_wave(v.g1, v.g2)


@pytest.mark.parametrize("key,pname,name", [("d", "d159", "d_rand32.vec"), ("f", "f", "f_rand16.vec")])
def test_limb_form_g1_ladder_of_the_5_word_fields_on_host(oracles, key, pname, name):
    """group_l5.cuh (round 4): element_mul_zn on G1 of d159.param / f.param in limb form -- the Jacobian doubling and mixed
    addition of the pairing kernels' limb-form steps without their line coefficients, under the same worst-case tracker;
    random and exceptional scalars against the oracle, the lanes the ladder reports through the complete routine;
    "hip_no_limb 1" keeps the word-form policy and gives the same bytes."""
    import hostsim
    from conftest import _param
    S, O = hostsim.HostSim(_param(pname)), oracles[key]
    v = golden(name)
    r = _order(pname)
    zl = (r.bit_length() + 7) // 8
    rng = np.random.default_rng(3)
    ks = [int.from_bytes(rng.bytes(zl), "big") % r for _ in range(6)] + [0, 1, 2, 3, r - 1, r - 2, r, r + 1, 2**(8 * zl) - 1, 2**(8 * zl) - 2, 15, 16, 17]
    Z = np.stack([_be(k, zl) for k in ks])
    P = np.tile(v.g1, (2, 1))[:len(ks)]
    S.fallbacks()
    got = S.group(0, P, Z)
    assert np.array_equal(got, O.g_mul(1, P, Z))
    assert S.fallbacks() == 3                                  # 0 P, r P and (r - 1) P = r P - P
    W = hostsim.HostSim(_param(pname) + "hip_no_limb 1\n")
    assert np.array_equal(W.group(0, P, Z), got)
    bad = P.copy()
    bad[2, -1] ^= 1                                            # off the curve: O
    assert np.array_equal(S.group(0, bad, Z), O.g_mul(1, bad, Z))


def test_pp_apply_and_products_on_the_wave_routines_on_host(sims, oracles):
    """round 5 (pairing_aw.cuh pp_apply_wave / miller_record_wave / prod_finish_wave): pairing_pp_apply and few-term
    products on the one-unit-per-wavefront routines give the reference's bytes -- same values as element_pairing /
    element_prod_pairing, edge cases (off-curve arguments, O) included"""
    S = sims["a"]
    v = golden("a_rand32.vec")
    assert np.array_equal(S.pp_wave(v.g1[3], v.g2[:5]), oracles["a"].pairing_batch(np.tile(v.g1[3], (5, 1)), v.g2[:5]))
    e = golden("a_edge20.vec")
    for i in (0, 7, 13):                                      # whatever the fixture holds there: valid or not
        got = S.pp_wave(e.g1[i], e.g2[i:i + 2])
        assert np.array_equal(got, oracles["a"].pairing_batch(np.tile(e.g1[i], (2, 1)), e.g2[i:i + 2]))
    p = golden("a_prod2x8.vec")
    assert np.array_equal(S.prod_wave(p.g1[:6], p.g2[:6], 2), p.gt[:3])
    pe = golden("a_prod3x10_edge.vec")
    assert np.array_equal(S.prod_wave(pe.g1, pe.g2, 3), pe.gt)
    m = golden("a_160_512_mm_rand6.vec")                       # sign1 = -1: the addition step negates P
    Sm = sims["a_160_512_mm"]
    assert np.array_equal(Sm.pp_wave(m.g1[1], m.g2[1:3]), np.concatenate([m.gt[1:2], Sm.prod_pairing(m.g1[1:2], m.g2[2:3], 1)]))


@pytest.mark.parametrize("key,qkey,nw", [("e", "q", 33), ("a1", "p", 33), ("e_160_400", "q", 16)])
def test_fused_products_on_host(sims, key, qkey, nw):
    """fp.cuh fp_mulx / fp_sqrx (round 5: the steps of types a1 / e): every selector -- the sums and differences before
    the product, the small multiple k b, the shifted terms after it, the closing doublings -- against plain integers
    mod q, on the memory-operand path (33 words) and on the register path (16 words); operands in [0, q) including 0 and
    q - 1.  The constant the type e steps multiply by (x of the auxiliary point) is checked to be a small integer."""
    sim = sims[key]
    q = param_value(key, qkey)
    W = 28 if nw >= 32 else 29                              # fp.cuh Limbs29: the Montgomery radix is 2^(W L), L limbs of W bits
    R = 1 << (W * ((32 * nw + W - 1) // W))
    Rinv = pow(R, -1, q)
    rng = np.random.default_rng(20250922)
    words = lambda x: [(x >> (32 * i)) & 0xffffffff for i in range(nw)]
    value = lambda w: sum(int(x) << (32 * i) for i, x in enumerate(w))
    rnd = lambda: int.from_bytes(rng.bytes(4 * nw + 8), "big") % q
    A_ADD, A_SUB, B_ADD, B_SUB, C1_ADD, C1_SUB, C2_ADD, C2_SUB = 1, 2, 4, 8, 16, 32, 256, 512
    cases = [(0, 0), (1, 0)]
    for _ in range(60):
        op = int(rng.choice([0, A_ADD, A_SUB])) | int(rng.choice([0, B_ADD, B_SUB])) | int(rng.choice([0, C1_ADD, C1_SUB])) | int(rng.choice([0, C2_ADD, C2_SUB]))
        op |= int(rng.integers(0, 4)) << 6 | int(rng.integers(0, 4)) << 10 | int(rng.integers(0, 4)) << 12
        if rng.integers(0, 2):
            op |= int(rng.choice([2, 3, 5, 7, 8, 100, 255])) << 16
        cases.append((int(rng.integers(0, 3)), op))
    cases += [(2, 0), (2, 1 << 14), (2, (1 << 14) | 8 | (2 << 16) | 32 | (1 << 6))]      # the two-product sums as the steps use them
    for t, (sqr, op) in enumerate(cases):
        e = [rnd() for _ in range(7)]
        if t % 7 == 3:
            e[t % 7] = 0
        if t % 7 == 5:
            e[(t + 1) % 7] = q - 1
        a, a2, b, b2, c1, c2, d2 = e
        k = (op >> 16) & 255
        if sqr == 2:                     # fp_sopx: a b +- a2 (k b2 +- d2), one reduction
            v = (k if k > 1 else 1) * b2 + (d2 if op & 4 else -d2 if op & 8 else 0)
            z = (a * b + (-1 if op & (1 << 14) else 1) * a2 * v) * Rinv
        else:
            x = a + (a2 if op & 1 else -a2 if op & 2 else 0)
            y = x if sqr else (k if k > 1 else 1) * b + (b2 if op & 4 else -b2 if op & 8 else 0)
            z = x * y * Rinv
        z += (c1 << ((op >> 6) & 3)) * (1 if op & 16 else -1 if op & 32 else 0)
        z += (c2 << ((op >> 10) & 3)) * (1 if op & 256 else -1 if op & 512 else 0)
        z = (z << ((op >> 12) & 3)) % q
        got = value(sim.fx(sqr, op, [words(v) for v in e]))
        assert got == z, (key, sqr, hex(op))
    if key.startswith("e"):
        assert 1 <= sim.e_rxs() <= 255


@pytest.mark.parametrize("pname,rand,prod", [("a1", "a1_rand6.vec", "a1_prod3x3_edge.vec"), ("a_160_1024", "a_160_1024_rand4.vec", "a_160_1024_prod3x3_edge.vec"),
                                             ("a_160_256", "a_160_256_rand6.vec", "a_160_256_prod3x4_edge.vec")])
def test_wave_kernels_of_type_a1_and_generic_type_a_on_host(pname, rand, prod):
    """pairing_aw.cuh with AG<N> (round 6: type a1, type a outside the fast path; 38 limbs of 28 bits on the 33-word fields): the same
    source lane by lane on the host under the bound tracker of AG's host mirror -- limbs within what was derived for them, every
    subtrahend dominated by its borrowed constant (limbs AND value, with the object's own top-limb fill), no negative limb, column
    capacity, products' values below 2 q with the object's own radix slack, nothing above the limbs q fills -- plus the
    recurrence's checks; the reference's vectors for pairings (edge cases included), products and pairing_pp_apply"""
    import hostsim
    from conftest import _param
    S = hostsim.HostSim(_param(pname))
    v = golden(rand)
    assert np.array_equal(S.pairing_wave(v.g1, v.g2), v.gt)
    if pname == "a1":
        e = golden("a1_edge6.vec")
        assert np.array_equal(S.pairing_wave(e.g1, e.g2), e.gt)
    w = golden(prod)
    n = min(w.n, 2)
    assert np.array_equal(S.prod_wave(w.g1[:n * w.k], w.g2[:n * w.k], w.k), w.gt[:n])
    assert np.array_equal(S.pp_wave(v.g1[0], v.g2[:2]), S.pairing_wave(np.tile(v.g1[0], (2, 1)), v.g2[:2]))


@pytest.mark.parametrize("pname,rand,prod", [("e", "e_rand6.vec", "e_prod3x3_edge.vec"), ("e_160_400", "e_160_400_rand6.vec", "e_160_400_prod3x4_edge.vec")])
def test_wave_kernels_of_type_e_on_host(pname, rand, prod, oracles):
    """pairing_ew.cuh (round 6: type e on the limb-per-lane routines): its doubling and addition steps in rounds of four products,
    numerator and denominator apart, the sliding-window power -- the same source on the host under the tracker of AG's mirror, which
    checks every bound the file notes (subtrahends below the constants that dominate them with e.param's twelve-bit top limb,
    columns within 2.55 units, limbs within 32 bits); the reference's vectors (edge cases and products included) and cross pairs
    against the C restatement"""
    import hostsim
    from conftest import _param
    S = hostsim.HostSim(_param(pname))
    v = golden(rand)
    assert np.array_equal(S.pairing_wave(v.g1, v.g2), v.gt)
    if pname == "e":
        e = golden("e_edge6.vec")
        assert np.array_equal(S.pairing_wave(e.g1, e.g2), e.gt)
    w = golden(prod)
    assert np.array_equal(S.prod_wave(w.g1, w.g2, w.k), w.gt)
    i, j = np.arange(v.n), (np.arange(v.n) * 5 + 2) % v.n
    assert np.array_equal(S.pairing_wave(v.g1[i], v.g2[j]), oracles[pname].pairing_batch(v.g1[i], v.g2[j]))
