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
