"""CPU tests: the plain-C oracle (oracle/pbc_oracle.c) against the reference's golden
vectors (tests/golden/*.vec, written by the unmodified reference via oracle/_ref/ref_tool)
and against the algebraic properties the reference's own tests check
(pbc/bilinear.test:17-33, pbc/pairing_test.pbc:16-21, guru/prodpairing_test.c:12-31)."""
import glob
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, G2_HASH, G2_COMPRESS, G2_XONLY, golden, key_of, param_value


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "*.vec"))
                                        if not any(t in os.path.basename(p) for t in ("_hash", "_g1mul", "_g2mul", "_compress", "_xonly", "_g2hash", "_g2compress", "_g2xonly", "_finalpow", "pow12"))))
def test_oracle_matches_reference_vectors(oracles, name):
    """Types A, D (d159 and the five other shipped type d files) and F: the D and F fixtures are the only pins for those curves
    (the reference ships no D/F known-answer test)."""
    v = golden(name)
    oracle_a = oracles[key_of(name)]
    if v.n > 64:                       # keep the CPU suite fast: head, tail and a stride
        idx = np.r_[0:16, v.n - 16:v.n, 16:v.n - 16:61]
    else:
        idx = np.arange(v.n)
    if v.k == 1:
        out = oracle_a.pairing_batch(v.g1[idx], v.g2[idx])
    else:
        sel = (idx[:, None] * v.k + np.arange(v.k)[None, :]).ravel()
        out = oracle_a.prod_pairing_batch(v.g1[sel], v.g2[sel], v.k)
    assert np.array_equal(out, v.gt[idx])


def test_kat_pairing_test_pbc(oracle_a):
    """pbc/pairing_test.pbc:3-10 -- the reference's only pairing known-answer test."""
    v = golden("a_kat.vec")
    g = int("2382389466570123849673299401984867521337122094157231907755149435707124249269394670242462497382963719723036281844079382411446883273020125104982896098602669")
    assert int.from_bytes(v.g1[0, :64].tobytes(), "big") == g
    want_re = int("1352478452661998164151215014828915385601138645645403926287105573769451214277485326392786454433874957123922454604362337349978217917242114505658729401276644")
    out = oracle_a.pairing_batch(v.g1, v.g2)
    assert int.from_bytes(out[0, :64].tobytes(), "big") == want_re
    assert np.array_equal(out, v.gt)


def test_fq_mul_count_matches_survey(oracle_a):
    """one Type-A pairing = 4392 Fq mul + 4 inversions in the reference
    algorithm (+6 mul for the two on-curve checks of from_bytes)."""
    v = golden("a_kat.vec")
    oracle.counters(reset=True)
    oracle_a.pairing_batch(v.g1, v.g2)
    mul, inv = oracle.counters()
    assert (mul, inv) == (4392 + 6, 4)


def _be(x, n):
    return np.frombuffer(int(x).to_bytes(n, "big"), np.uint8)


def test_bilinearity(oracle_a):
    """pbc/pairing_test.pbc:16-21 / pbc/bilinear.test:24-33: e(aP,bQ) = e(P,Q)^(ab)."""
    v = golden("a_rand32.vec")
    a, b = 171583727262251826931173602797951212789946235851, 233634857565210859330459959563397971304462340857
    r = 730750818665451621361119245571504901405976559617
    P, Q = v.g1[:2], v.g2[:2]
    aP = oracle_a.g_mul(1, P, np.tile(_be(a, 20), (2, 1)))
    bQ = oracle_a.g_mul(2, Q, np.tile(_be(b, 20), (2, 1)))
    lhs = oracle_a.pairing_batch(aP, bQ)
    rhs = oracle_a.gt_pow(v.gt[:2], np.tile(_be(a * b % r, 20), (2, 1)))
    assert np.array_equal(lhs, rhs)


def test_identity_and_offcurve(oracle_a):
    """pairing_apply short-circuit (include/pbc_pairing.h:123-130) + curve_from_bytes
    mapping off-curve bytes to O (ecc/curve.c:618-621)."""
    v = golden("a_rand32.vec")
    g1 = v.g1[:1].copy()
    g1[0, -1] ^= 1
    out = oracle_a.pairing_batch(g1, v.g2[:1])
    one = np.zeros(128, np.uint8)
    one[63] = 1
    assert np.array_equal(out[0], one)


def test_prod_equals_product_of_pairings(oracle_a):
    """guru/prodpairing_test.c:12-31, benchmark/multipairing.c:49."""
    v = golden("a_prod2x8.vec")
    singles = oracle_a.pairing_batch(v.g1, v.g2)
    prod = oracle_a.gt_mul(singles[0::2], singles[1::2])
    assert np.array_equal(prod, v.gt)


def test_fq_ops_against_python_ints(oracle_a, a_param_text):
    """guru/fp_test.c:18-84 restated: Fq add/sub/mul/invert/neg/halve/double vs big ints."""
    q = int([l.split()[1] for l in a_param_text.splitlines() if l.startswith("q ")][0])
    rng = np.random.default_rng(1)
    xs = [int.from_bytes(rng.bytes(64), "big") % q for _ in range(16)] + [0, 1, q - 1]
    ys = [int.from_bytes(rng.bytes(64), "big") % q for _ in range(16)] + [q - 1, 0, q - 1]
    A = np.stack([_be(x, 64) for x in xs])
    B = np.stack([_be(y, 64) for y in ys])
    exp = {0: lambda x, y: x * y % q, 1: lambda x, y: (x + y) % q, 2: lambda x, y: (x - y) % q,
           3: lambda x, y: pow(x, q - 2, q), 4: lambda x, y: (-x) % q,
           5: lambda x, y: x * pow(2, q - 2, q) % q, 6: lambda x, y: 2 * x % q}
    for op, fn in exp.items():
        got = oracle_a.fq_op(op, A, B)
        want = np.stack([_be(fn(x, y), 64) for x, y in zip(xs, ys)])
        assert np.array_equal(got, want), op


@pytest.mark.parametrize("t,name,a", [("d", "d_rand32.vec", 42), ("f", "f_rand16.vec", 17 * 2**80 + 5)])
def test_bilinearity_d_f(oracles, t, name, a):
    """pbc/bilinear.test:24-33 for the asymmetric curves: e(aP, Q) = e(P, Q)^a."""
    O = oracles[t]
    v = golden(name)
    P, Q = v.g1[:2], v.g2[:2]
    aP = O.g_mul(1, P, np.tile(_be(a, 20), (2, 1)))
    lhs = O.pairing_batch(aP, Q)
    rhs = O.gt_pow(v.gt[:2], np.tile(_be(a, 20), (2, 1)))
    assert np.array_equal(lhs, rhs)


def test_op_counts_d_f(oracles):
    """Reference-algorithm work per pairing (SURVEY.md 3.4/3.5 order of magnitude; the counts
    here are of the oracle's schoolbook towers and feed nothing but this sanity check)."""
    for t, name in (("d", "d_rand32.vec"), ("f", "f_rand16.vec")):
        v = golden(name)
        oracle.counters(reset=True)
        oracles[t].pairing_batch(v.g1[:1], v.g2[:1])
        mul, inv = oracle.counters()
        assert 10_000 < mul < 400_000 and 100 < inv < 400, (t, mul, inv)


HASH_FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "*_hash*.vec")))
FORMAT_FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "*_compress*.vec")) if "_g2" not in p)
XONLY_FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "*_xonly*.vec")) if "_g2" not in p)


@pytest.mark.parametrize("name", HASH_FILES)
def test_oracle_from_hash_matches_reference_vectors(oracles, name):
    """oracle_from_hash (curve_from_hash, ecc/curve.c:455-482) pinned on every element_from_hash vector of the
    reference: types a (three digest lengths), a1, d (three files), e, f, g"""
    v = golden(name)
    assert np.array_equal(oracles[key_of(name)].from_hash(v.g1.reshape(v.n, v.len1)), v.gt)


@pytest.mark.parametrize("name", FORMAT_FILES)
def test_oracle_compressed_points_match_reference_vectors(oracles, name):
    v = golden(name)
    O = oracles[key_of(name)]
    assert np.array_equal(O.point_format(0, v.g1), v.gt)
    assert np.array_equal(O.point_format(1, v.gt), v.g1)


@pytest.mark.parametrize("name", XONLY_FILES)
def test_oracle_x_only_points_match_reference_vectors(oracles, name):
    """exact where q = 3 mod 4; elsewhere the reference's root depends on its random non-residue (see the x-only
    tests of the product), so only x and the curve equation are pinned"""
    v = golden(name)
    O = oracles[key_of(name)]
    assert np.array_equal(O.point_format(2, v.g1), v.g2)
    got = O.point_format(3, v.g2)
    fb = v.len1 // 2
    assert np.array_equal(got[:, :fb], v.gt[:, :fb])
    q = None
    try:
        from conftest import param_value
        q = param_value(key_of(name), "q")
    except KeyError:
        pass
    if q is None or q % 4 == 3:
        assert np.array_equal(got, v.gt)
    else:
        for g, w in zip(got, v.gt):
            y, yr = int.from_bytes(g[fb:].tobytes(), "big"), int.from_bytes(w[fb:].tobytes(), "big")
            assert y == yr or y == q - yr


@pytest.mark.parametrize("key,name", G2_HASH)
def test_oracle_from_hash_on_the_twists_matches_reference_vectors(oracles, key, name):
    v = golden(name)
    assert np.array_equal(oracles[key].from_hash_g2(v.g1.reshape(v.n, v.len1)), v.gt)


@pytest.mark.parametrize("key,name", G2_COMPRESS)
def test_oracle_compressed_points_on_the_twists_match_reference_vectors(oracles, key, name):
    v = golden(name)
    O = oracles[key]
    assert np.array_equal(O.point_format_g2(0, v.g1), v.gt)
    assert np.array_equal(O.point_format_g2(1, v.gt), v.g1)


@pytest.mark.parametrize("key,name,exact", G2_XONLY)
def test_oracle_x_only_points_on_the_twists(oracles, key, name, exact):
    """the oracle takes its extension-field roots by Tonelli-Shanks, the reference by fq_sqrt / polymod_sqrt: x is
    pinned, y up to sign"""
    v = golden(name)
    O = oracles[key]
    q = param_value(key, "q")
    fb = (q.bit_length() + 7) // 8
    half = v.len1 // 2
    assert np.array_equal(O.point_format_g2(2, v.g1), v.g2)
    got = O.point_format_g2(3, v.g2)
    assert np.array_equal(got[:, :half], v.gt[:, :half])
    for g, w in zip(got, v.gt):
        cg = [int.from_bytes(g[half + i:half + i + fb].tobytes(), "big") for i in range(0, half, fb)]
        cw = [int.from_bytes(w[half + i:half + i + fb].tobytes(), "big") for i in range(0, half, fb)]
        assert cg == cw or cg == [(q - c) % q for c in cw]


FINALPOW = [("a", "a_finalpow6.vec"), ("d", "d159_finalpow6.vec"), ("f", "f_finalpow6.vec"), ("g149", "g149_finalpow6.vec"),
            ("e", "e_finalpow3.vec"), ("a1", "a1_finalpow3.vec")]


@pytest.mark.parametrize("key,name", FINALPOW)
def test_oracle_finalpow_matches_reference(oracles, key, name):
    """pairing->finalpow (include/pbc_pairing.h:41) on random elements of GT's underlying field, written by the reference"""
    v = golden(name)
    assert np.array_equal(oracles[key].finalpow(v.g1), v.gt)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference sources (build container only)")
def test_golden_recipe_reproduces_every_fixture(tmp_path):
    """tests/golden/make_golden.sh is the COMPLETE recipe: run against the unmodified reference (compiled by
    oracle/Makefile from the sources where they lie) it rewrites every committed fixture byte for byte, and there is no
    committed fixture it does not write (25 fixtures had no line in the script)."""
    import subprocess
    root = os.path.dirname(GOLDEN.rstrip("/"))
    root = os.path.dirname(root)
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, G=str(tmp_path))
    subprocess.check_call(["sh", os.path.join(GOLDEN, "make_golden.sh")], cwd=root, env=env,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith((".vec", ".txt", ".bin", ".rec")))
    written = sorted(os.listdir(tmp_path))
    assert written == committed
    for f in committed:
        with open(os.path.join(GOLDEN, f), "rb") as a, open(os.path.join(tmp_path, f), "rb") as b:
            assert a.read() == b.read(), f
