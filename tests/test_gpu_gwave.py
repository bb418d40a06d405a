"""GPU tests (-m gpu) of the one-pairing-per-wavefront kernel of type g on the five-word field (pairing_gw.cuh, round 6): small
batches of g149.param element_pairing / element_prod_pairing / pairing_pp_apply calls run level programs generated -- and checked against the reference's vectors on Python
integers -- by tools/gw_gen.py; the bytes are those of the one-pairing-per-lane kernel and of the reference."""
import numpy as np
import pytest

from conftest import golden, _param

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lane():
    import pbc_amd
    P = pbc_amd.Pairing(_param("g149") + "hip_dwave_max 0\n")          # never the wave kernel
    yield P
    P.clear()


@pytest.mark.parametrize("name", ["g149_rand16.vec", "g149_edge10.vec", "g149_chain64.vec", "g149_full12.vec"])
def test_wave_kernel_matches_the_reference_vectors(hips, name):
    v = golden(name)
    assert np.array_equal(hips["g149"].element_pairing(v.g1, v.g2), v.gt)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 5119, 5120, 5121])
def test_wave_kernel_equals_the_lane_kernel_around_the_cut_over(hips, lane, n):
    v = golden("g149_chain64.vec")
    i = np.arange(n)
    g1, g2 = np.ascontiguousarray(v.g1[(i * 3 + 1) % v.n]), np.ascontiguousarray(v.g2[(i * 7 + 2) % v.n])
    g1[::41] ^= 1                                               # off-curve first arguments: the identity of GT
    g2[5::97, 3] ^= 2                                           # ... second arguments off the twist
    assert np.array_equal(hips["g149"].element_pairing(g1, g2), lane.element_pairing(g1, g2))


def test_wave_kernel_on_fresh_random_inputs(hips, oracles):
    v = golden("g149_chain64.vec")
    rng = np.random.default_rng(11)
    i, j = rng.integers(0, v.n, 40), rng.integers(0, v.n, 40)
    assert np.array_equal(hips["g149"].element_pairing(v.g1[i], v.g2[j]), oracles["g149"].pairing_batch(v.g1[i], v.g2[j]))


# ---- element_prod_pairing and pairing_pp_apply on wavefronts (products: one wavefront per TERM, then one per product) ----
@pytest.mark.parametrize("name", ["g149_prod4x3.vec", "g149_prod3x4_edge.vec"])
def test_products_on_wavefronts_match_the_reference_vectors(hips, lane, name):
    v = golden(name)
    got = hips["g149"].element_prod_pairing(v.g1, v.g2, v.k)
    assert np.array_equal(got, v.gt)
    assert np.array_equal(got, lane.element_prod_pairing(v.g1, v.g2, v.k))


@pytest.mark.parametrize("n,k", [(1, 2), (1, 3), (7, 5), (33, 16), (2, 64), (1, 300), (300, 2)])
def test_products_on_wavefronts_equal_the_c_restatement(hips, oracles, n, k):
    """term counts from two to hundreds (one workspace record per term), an off-curve term in some products (the identity)"""
    v = golden("g149_chain64.vec")
    i = np.arange(n * k)
    g1, g2 = np.ascontiguousarray(v.g1[(i * 5 + 3) % v.n]), np.ascontiguousarray(v.g2[(i * 11 + 1) % v.n])
    if n > 2:
        g1[k + 1, 3] ^= 4                                         # product 1: a term off the curve
        g2[2 * k, 7] ^= 1                                         # product 2: its first term's second argument
    got = hips["g149"].element_prod_pairing(g1, g2, k)
    m = min(n, 6 if k <= 16 else 1)
    assert np.array_equal(got[:m], oracles["g149"].prod_pairing_batch(g1[:m * k], g2[:m * k], k))
    if n > 2:
        one = np.zeros(got.shape[1], np.uint8)
        one[got.shape[1] // 10 - 1] = 1
        assert np.array_equal(got[1], one) and np.array_equal(got[2], one)


@pytest.mark.parametrize("n,k", [(5119, 2), (5120, 2), (5121, 2), (300, 16)])
def test_products_equal_the_lane_kernel_around_the_cut_over(hips, lane, n, k):
    v = golden("g149_chain64.vec")
    i = np.arange(n * k)
    g1, g2 = np.ascontiguousarray(v.g1[(i * 3 + 1) % v.n]), np.ascontiguousarray(v.g2[(i * 7 + 2) % v.n])
    g1[::97] ^= 1
    assert np.array_equal(hips["g149"].element_prod_pairing(g1, g2, k), lane.element_prod_pairing(g1, g2, k))


def test_product_is_the_product_of_the_pairings(hips):
    """a size-independent property: prod_t e(P_t, Q_t) through the product route = the GT product of the single pairings"""
    v = golden("g149_chain64.vec")
    H = hips["g149"]
    k = 4
    g1, g2 = v.g1[:16 * k], np.ascontiguousarray(v.g2[16 * k - 1::-1][:16 * k])
    singles = H.element_pairing(g1, g2).reshape(16, k, -1)
    acc = singles[:, 0]
    for t in range(1, k):
        acc = H.element_mul_GT(np.ascontiguousarray(acc), np.ascontiguousarray(singles[:, t]))
    assert np.array_equal(H.element_prod_pairing(g1, g2, k), acc)


@pytest.mark.parametrize("n", [1, 3, 200, 5120, 5121])
def test_pairing_pp_apply_on_wavefronts(hips, lane, oracles, n):
    """pairing_pp_apply for up to hip_dwave_max second arguments reads the lines of the pairing_pp_init table inside the wave
    kernel: same bytes as the lane kernel's apply on the same table and as the C restatement's element_pairing"""
    v = golden("g149_chain64.vec")
    H = hips["g149"]
    i = np.arange(n)
    Q = np.ascontiguousarray(v.g2[(i * 7 + 2) % v.n])
    if n > 2:
        Q[2, -1] ^= 1                                             # off the twist: the identity
    for pi in (0, 9):
        pp, pl = H.pp_init(v.g1[pi]), lane.pp_init(v.g1[pi])
        got = pp.apply(Q)
        assert np.array_equal(got, pl.apply(Q))
        m = min(n, 6)
        assert np.array_equal(got[:m], oracles["g149"].pairing_batch(np.tile(v.g1[pi], (m, 1)), Q[:m]))
        pp.clear()
        pl.clear()
    bad = v.g1[2].copy()
    bad[1] ^= 8                                                   # a first argument off the curve: every result is the identity
    got = H.pp_init(bad).apply(Q[:min(n, 4)])
    one = np.zeros(got.shape[1], np.uint8)
    one[got.shape[1] // 10 - 1] = 1
    assert np.array_equal(got, np.tile(one, (min(n, 4), 1)))
