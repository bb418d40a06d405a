"""GPU tests (-m gpu) of the one-pairing-per-wavefront kernel of type f on the five-word BN fields (pairing_fw.cuh, round 6): small
batches of f.param element_pairing calls run level programs generated -- and checked against the reference's vectors on Python
integers -- by tools/fw_gen.py; the bytes are those of the one-pairing-per-lane kernel and of the reference."""
import numpy as np
import pytest

from conftest import golden, _param

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lane():
    import pbc_amd
    P = pbc_amd.Pairing(_param("f") + "hip_fwave_max 0\n")             # never the wave kernel
    yield P
    P.clear()


@pytest.mark.parametrize("name", ["f_rand16.vec", "f_edge10.vec", "f_chain128.vec", "f_full12.vec"])
def test_wave_kernel_matches_the_reference_vectors(hips, name):
    """batches up to hip_fwave_max take the wave kernel: random, edge (off-curve -> identity), chain and whole-curve inputs"""
    v = golden(name)
    assert np.array_equal(hips["f"].element_pairing(v.g1, v.g2), v.gt)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 4095, 4096, 4097])
def test_wave_kernel_equals_the_lane_kernel_around_the_cut_over(hips, lane, n):
    v = golden("f_chain128.vec")
    i = np.arange(n)
    g1, g2 = np.ascontiguousarray(v.g1[(i * 3 + 1) % v.n]), np.ascontiguousarray(v.g2[(i * 7 + 2) % v.n])
    g1[::41] ^= 1                                               # off-curve first arguments: the identity of GT
    g2[5::97, 3] ^= 2                                           # ... second arguments off the twist
    assert np.array_equal(hips["f"].element_pairing(g1, g2), lane.element_pairing(g1, g2))


def test_wave_kernel_on_fresh_random_inputs(hips, oracles):
    """inputs no fixture holds: cross pairs of the chain, against the C restatement"""
    v = golden("f_chain128.vec")
    rng = np.random.default_rng(11)
    i, j = rng.integers(0, v.n, 60), rng.integers(0, v.n, 60)
    assert np.array_equal(hips["f"].element_pairing(v.g1[i], v.g2[j]), oracles["f"].pairing_batch(v.g1[i], v.g2[j]))


def test_other_type_f_parameters_keep_the_lane_kernel(hips):
    """the wave kernel is built for five-word BN fields: the 200- and 256-bit parameter sets are not routed to it"""
    for key, name in (("f_200", "f_200_rand4.vec"), ("f_256", "f_256_rand4.vec")):
        v = golden(name)
        assert np.array_equal(hips[key].element_pairing(v.g1, v.g2), v.gt)


# ---- element_prod_pairing: one wavefront per TERM, then one per product ----
@pytest.mark.parametrize("name", ["f_prod3x5_edge.vec", "f_prod4x3.vec", "f_prodfull3x4.vec"])
def test_products_on_wavefronts_match_the_reference_vectors(hips, lane, name):
    v = golden(name)
    got = hips["f"].element_prod_pairing(v.g1, v.g2, v.k)
    assert np.array_equal(got, v.gt)
    assert np.array_equal(got, lane.element_prod_pairing(v.g1, v.g2, v.k))


@pytest.mark.parametrize("n,k", [(1, 2), (5, 3), (40, 16), (2, 70), (300, 2), (4096, 2), (4097, 2)])
def test_products_on_wavefronts_equal_the_lane_kernel_and_the_c_restatement(hips, lane, oracles, n, k):
    v = golden("f_chain128.vec")
    i = np.arange(n * k)
    g1, g2 = np.ascontiguousarray(v.g1[(i * 5 + 3) % v.n]), np.ascontiguousarray(v.g2[(i * 11 + 1) % v.n])
    if n > 2:
        g1[k + 1, 3] ^= 4                                         # product 1: a term off the curve -> the identity
        g2[2 * k, 7] ^= 1                                         # product 2: its first term's second argument
    got = hips["f"].element_prod_pairing(g1, g2, k)
    m = min(n, 3 if k <= 16 else 1)
    assert np.array_equal(got[:m], oracles["f"].prod_pairing_batch(g1[:m * k], g2[:m * k], k))
    if n <= 300:
        assert np.array_equal(got, lane.element_prod_pairing(g1, g2, k))
    if n > 2:
        one = np.zeros(240, np.uint8)
        one[19] = 1
        assert np.array_equal(got[1], one) and np.array_equal(got[2], one)


def test_product_is_the_product_of_the_pairings(hips):
    v = golden("f_chain128.vec")
    H = hips["f"]
    k = 3
    g1, g2 = np.ascontiguousarray(v.g1[:20 * k]), np.ascontiguousarray(v.g2[20 * k - 1::-1][:20 * k])
    singles = H.element_pairing(g1, g2).reshape(20, k, -1)
    acc = np.ascontiguousarray(singles[:, 0])
    for t in range(1, k):
        acc = H.element_mul_GT(acc, np.ascontiguousarray(singles[:, t]))
    assert np.array_equal(H.element_prod_pairing(g1, g2, k), acc)
