"""GPU tests (-m gpu) of the one-pairing-per-wavefront kernel of the five-word type d fields (pairing_dw.cuh, round 6): small
batches of d159 element_pairing calls run level programs generated -- and checked against the reference's vectors on
Python integers -- by tools/dw_gen.py; the bytes are those of the one-pairing-per-lane kernel and of the reference."""
import numpy as np
import pytest

from conftest import golden, _param

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lane():
    import pbc_amd
    P = pbc_amd.Pairing(_param("d159") + "hip_dwave_max 0\n")          # never the wave kernel
    yield P
    P.clear()


@pytest.mark.parametrize("name", ["d_rand32.vec", "d_edge20.vec", "d_chain256.vec", "d159_full12.vec"])
def test_wave_kernel_matches_the_reference_vectors(hips, name):
    """batches up to hip_dwave_max (default 4096) take the wave kernel: random, edge (off-curve -> identity), chain and
    whole-curve inputs"""
    v = golden(name)
    assert np.array_equal(hips["d"].element_pairing(v.g1, v.g2), v.gt)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 4095, 4096, 4097])
def test_wave_kernel_equals_the_lane_kernel_around_the_cut_over(hips, lane, n):
    v = golden("d_chain256.vec")
    i = np.arange(n)
    g1, g2 = np.ascontiguousarray(v.g1[(i * 3 + 1) % v.n]), np.ascontiguousarray(v.g2[(i * 7 + 2) % v.n])
    g1[::41] ^= 1                                               # off-curve first arguments: the identity of GT
    assert np.array_equal(hips["d"].element_pairing(g1, g2), lane.element_pairing(g1, g2))


def test_wave_kernel_on_fresh_random_inputs(hips, oracles):
    """inputs no fixture holds: cross pairs of the chain, against the C restatement"""
    v = golden("d_chain256.vec")
    rng = np.random.default_rng(11)
    i, j = rng.integers(0, v.n, 200), rng.integers(0, v.n, 200)
    assert np.array_equal(hips["d"].element_pairing(v.g1[i], v.g2[j]), oracles["d"].pairing_batch(v.g1[i], v.g2[j]))


def test_other_five_word_parameters_keep_the_lane_kernel(hips):
    """the wave kernel is built for d = 3: type g (d = 5, five words) is not routed to it"""
    for key, name in (("g149", "g149_rand16.vec"),):
        v = golden(name)
        assert np.array_equal(hips[key].element_pairing(v.g1, v.g2), v.gt)


# ---- element_prod_pairing and pairing_pp_apply on wavefronts (products: one wavefront per TERM, then one per product) ----
@pytest.mark.parametrize("name", ["d_prod16x4.vec", "d_prod3x10_edge.vec"])
def test_products_on_wavefronts_match_the_reference_vectors(hips, lane, name):
    v = golden(name)
    got = hips["d"].element_prod_pairing(v.g1, v.g2, v.k)
    assert np.array_equal(got, v.gt)
    assert np.array_equal(got, lane.element_prod_pairing(v.g1, v.g2, v.k))


@pytest.mark.parametrize("n,k", [(1, 2), (1, 3), (7, 5), (64, 16), (3, 64), (1, 700), (300, 2)])
def test_products_on_wavefronts_equal_the_c_restatement(hips, oracles, n, k):
    """term counts from two to hundreds (the workspace holds one record per term), an off-curve term in some products (those give
    the identity, as cc_pairings_affine's callers see it through the lane kernel)"""
    v = golden("d_chain256.vec")
    i = np.arange(n * k)
    g1, g2 = np.ascontiguousarray(v.g1[(i * 5 + 3) % v.n]), np.ascontiguousarray(v.g2[(i * 11 + 1) % v.n])
    if n > 2:
        g1[k + 1, 3] ^= 4                                         # product 1: a term off the curve
        g2[2 * k, 7] ^= 1                                         # product 2: its first term's second argument
    got = hips["d"].element_prod_pairing(g1, g2, k)
    m = min(n, 12 if k <= 16 else 2)
    assert np.array_equal(got[:m], oracles["d"].prod_pairing_batch(g1[:m * k], g2[:m * k], k))
    if n > 2:
        one = np.zeros(120, np.uint8)
        one[19] = 1
        assert np.array_equal(got[1], one) and np.array_equal(got[2], one)


@pytest.mark.parametrize("n,k", [(5119, 2), (5120, 2), (5121, 2), (600, 16)])
def test_products_equal_the_lane_kernel_around_the_cut_over(hips, lane, n, k):
    v = golden("d_chain256.vec")
    i = np.arange(n * k)
    g1, g2 = np.ascontiguousarray(v.g1[(i * 3 + 1) % v.n]), np.ascontiguousarray(v.g2[(i * 7 + 2) % v.n])
    g1[::97] ^= 1
    assert np.array_equal(hips["d"].element_prod_pairing(g1, g2, k), lane.element_prod_pairing(g1, g2, k))


def test_product_is_the_product_of_the_pairings(hips):
    """a size-independent property: prod_t e(P_t, Q_t) through the product route = the GT product of the single pairings"""
    v = golden("d_chain256.vec")
    H = hips["d"]
    k = 4
    g1, g2 = v.g1[:32 * k], v.g2[32 * k - 1::-1][:32 * k]
    singles = H.element_pairing(g1, np.ascontiguousarray(g2)).reshape(32, k, -1)
    acc = singles[:, 0]
    for t in range(1, k):
        acc = H.element_mul_GT(np.ascontiguousarray(acc), np.ascontiguousarray(singles[:, t]))
    assert np.array_equal(H.element_prod_pairing(g1, np.ascontiguousarray(g2), k), acc)


@pytest.mark.parametrize("n", [1, 3, 200, 5120, 5121])
def test_pairing_pp_apply_on_wavefronts(hips, lane, oracles, n):
    """pairing_pp_apply for up to hip_dwave_max second arguments reads the lines of the pairing_pp_init table inside the
    wave kernel: same bytes as element_pairing, as the lane kernel's apply on the same table, and as the C restatement"""
    v = golden("d_chain256.vec")
    H = hips["d"]
    i = np.arange(n)
    Q = np.ascontiguousarray(v.g2[(i * 7 + 2) % v.n])
    if n > 2:
        Q[2, -1] ^= 1                                             # off the twist: the identity
    for pi in (0, 9):
        pp, pl = H.pp_init(v.g1[pi]), lane.pp_init(v.g1[pi])
        got = pp.apply(Q)
        assert np.array_equal(got, pl.apply(Q))
        m = min(n, 8)
        assert np.array_equal(got[:m], oracles["d"].pairing_batch(np.tile(v.g1[pi], (m, 1)), Q[:m]))
        pp.clear()
        pl.clear()
    bad = v.g1[2].copy()
    bad[1] ^= 8                                                   # a first argument off the curve: every result is the identity
    one = np.zeros(120, np.uint8)
    one[19] = 1
    assert np.array_equal(H.pp_init(bad).apply(Q[:min(n, 4)]), np.tile(one, (min(n, 4), 1)))


# ---- the six- and seven-word type d fields (seven / eight limbs): the same level programs, other curves and loop lengths ----
@pytest.mark.parametrize("pname", ["d278027-190-181", "d277699-175-167", "d105171-196-185", "d201", "d224"])
def test_wider_fields_on_wavefronts(oracles, pname):
    """d = 3 on six and seven words (175 ... 224-bit q; 22- ... 28-byte coordinates; eight limbs: the four lanes' joined columns of an
    eight-term sum are relieved before the reduction): pairings, products and pairing_pp_apply of small batches against the
    reference's vectors, the lane kernels ("hip_dwave_max 0") and the C restatement"""
    import pbc_amd
    H, Ln = pbc_amd.Pairing(_param(pname)), pbc_amd.Pairing(_param(pname) + "hip_dwave_max 0\n")
    for name in ("_rand12.vec", "_edge8.vec"):
        v = golden(pname + name)
        assert np.array_equal(H.element_pairing(v.g1, v.g2), v.gt)
    w = golden(pname + "_prod3x4_edge.vec")
    assert np.array_equal(H.element_prod_pairing(w.g1, w.g2, w.k), w.gt)
    v = golden(pname + "_rand12.vec")
    rng = np.random.default_rng(5)
    n = 700
    i, j = rng.integers(0, v.n, n), rng.integers(0, v.n, n)
    g1, g2 = np.ascontiguousarray(v.g1[i]), np.ascontiguousarray(v.g2[j])
    g1[::53, 2] ^= 1                                            # off the curve: the identity
    got = H.element_pairing(g1, g2)
    assert np.array_equal(got, Ln.element_pairing(g1, g2))
    assert np.array_equal(got[:6], oracles[pname].pairing_batch(g1[:6], g2[:6]))
    for k in (2, 7):
        m = n // k
        assert np.array_equal(H.element_prod_pairing(g1[:m * k], g2[:m * k], k), Ln.element_prod_pairing(g1[:m * k], g2[:m * k], k)), k
    pp, pl = H.pp_init(v.g1[1]), Ln.pp_init(v.g1[1])
    assert np.array_equal(pp.apply(g2), pl.apply(g2))
    assert np.array_equal(pp.apply(g2[:4]), H.element_pairing(np.tile(v.g1[1], (4, 1)), g2[:4]))
    pp.clear(); pl.clear(); H.clear(); Ln.clear()
