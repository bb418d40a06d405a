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
    """the wave kernel is built for d = 3 on five words: type g (d = 5) and the wider type d fields are not routed to it"""
    for key, name in (("g149", "g149_rand16.vec"), ("d201", "d201_rand12.vec")):
        v = golden(name)
        assert np.array_equal(hips[key].element_pairing(v.g1, v.g2), v.gt)
