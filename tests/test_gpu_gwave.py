"""GPU tests (-m gpu) of the one-pairing-per-wavefront kernel of type g on the five-word field (pairing_gw.cuh, round 6): small
batches of g149.param element_pairing calls run level programs generated -- and checked against the reference's vectors on Python
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


def test_products_and_pairing_pp_keep_their_kernels(hips, lane):
    """products and pairing_pp_apply of type g stay on the lane kernels (same bytes as before)"""
    w = golden("g149_prod4x3.vec")
    assert np.array_equal(hips["g149"].element_prod_pairing(w.g1, w.g2, w.k), w.gt)
    v = golden("g149_chain64.vec")
    pp = hips["g149"].pp_init(v.g1[2])
    assert np.array_equal(pp.apply(v.g2[:9]), lane.element_pairing(np.tile(v.g1[2], (9, 1)), v.g2[:9]))
    pp.clear()
