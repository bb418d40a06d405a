"""GPU tests (-m gpu) of the low-latency path for small batches (round 4, pairing_aw.cuh): element_pairing on a.param with
one WAVEFRONT per pairing -- an F_q element is one register across 18 lanes, the Montgomery product runs over the lanes
(v_readlane / DPP wave shifts); up to "hip_wave4_max" units (default 768) FOUR wavefronts -- up to "hip_wave2_max" (default 1280) two -- share the independent products
of every step of a pairing through LDS.  Batches up to "hip_wave_max" (default 5120) take these kernels; the bytes are
those of the throughput kernel and of the reference's vectors, invalid arguments included."""
import numpy as np
import pytest

from conftest import golden, _param, PARAM_OF

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 63, 64, 768, 769, 1024, 1280, 1281, 5120, 5121])
def test_wave_pairings_match_the_throughput_kernel_and_the_reference(hip_a, n):
    import pbc_amd
    v = golden("a_chain1024.vec")
    T = pbc_amd.Pairing(_param("a") + "hip_wave_max 0\n")       # never the wave path
    i = (np.arange(n) * 7 + 3) % v.n
    j = (np.arange(n) * 13 + n) % v.n
    g1, g2 = np.ascontiguousarray(v.g1[i]), np.ascontiguousarray(v.g2[j])
    got = hip_a.element_pairing(g1, g2)
    assert np.array_equal(got, T.element_pairing(g1, g2))
    d = i == j
    if d.any():
        assert np.array_equal(got[d], v.gt[i[d]])
    T.clear()


@pytest.mark.parametrize("extra", ["", "hip_wave4_max 0\nhip_wave2_max 0\n", "hip_wave4_max 100000\n", "hip_wave4_max 0\nhip_wave2_max 100000\n"])
def test_wave_pairings_on_the_reference_vectors_and_edge_cases(extra):
    """all three forms (one, two and four wavefronts per pairing) on the reference's vectors, invalid arguments included"""
    import pbc_amd
    H = pbc_amd.Pairing(_param("a") + extra)
    for name in ("a_rand32.vec", "a_edge20.vec", "a_chain1024.vec"):
        v = golden(name)
        assert np.array_equal(H.element_pairing(v.g1, v.g2), v.gt), name
    H.clear()


def test_wave_pairings_other_512_bit_parameters(hips):
    """r = 2^a - 2^b - 1 (the add step negates P), a second 512-bit field"""
    v = golden("a_160_512_mm_rand6.vec")
    assert np.array_equal(hips["a_160_512_mm"].element_pairing(v.g1, v.g2), v.gt)


def test_wave_pairings_device_buffers_and_streams(hip_a):
    import torch
    v = golden("a_chain1024.vec")
    n = 777
    g1 = torch.from_numpy(np.ascontiguousarray(v.g1[:n])).cuda()
    g2 = torch.from_numpy(np.ascontiguousarray(v.g2[:n])).cuda()
    out = torch.empty((n + 1, 128), dtype=torch.uint8, device="cuda")
    out[n] = 0xA5
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hip_a.element_pairing_dev(out.data_ptr(), g1.data_ptr(), g2.data_ptr(), n, s.cuda_stream)
    s.synchronize()
    assert np.array_equal(out[:n].cpu().numpy(), v.gt[:n])
    assert (out[n] == 0xA5).all()


@pytest.mark.parametrize("n", [1, 2, 63, 300, 768, 769, 1280, 1281, 5120, 5121])
def test_wave_pp_apply_matches_the_throughput_kernel(hip_a, n):
    """pairing_pp_apply on small batches (round 5: pp_apply_wave, four wavefronts per unit up to hip_wave4_max, one up to
    hip_wave_max) = the lane kernel's bytes = element_pairing's; first arguments that deserialise to O included"""
    import pbc_amd
    v = golden("a_chain1024.vec")
    T = pbc_amd.Pairing(_param("a") + "hip_wave_max 0\n")
    j = (np.arange(n) * 13 + n) % v.n
    g2 = np.ascontiguousarray(v.g2[j])
    for base in (v.g1[5], np.zeros_like(v.g1[5])):
        pp, ppT = hip_a.pp_init(base), T.pp_init(base)
        got = pp.apply(g2)
        assert np.array_equal(got, ppT.apply(g2))
        m = min(n, 64)
        assert np.array_equal(got[:m], hip_a.element_pairing(np.tile(base, (m, 1)), g2[:m]))
        pp.clear()
        ppT.clear()
    T.clear()


@pytest.mark.parametrize("n,k", [(1, 2), (1, 5), (3, 8), (64, 16), (200, 5), (341, 3), (1024, 5), (1025, 5)])
def test_wave_products_match_the_throughput_kernels(hip_a, n, k):
    """few-term products with n k <= hip_wave_max (round 5: a wave -- four up to hip_wave4_max terms -- per term, then one
    per product): the bytes of the one-term-per-lane kernels and of the reference's product vectors"""
    import pbc_amd
    v = golden("a_chain1024.vec")
    T = pbc_amd.Pairing(_param("a") + "hip_wave_max 0\n")
    t = np.arange(n * k)
    g1, g2 = np.ascontiguousarray(v.g1[(t * 5 + 2) % v.n]), np.ascontiguousarray(v.g2[(t * 3 + k) % v.n])
    g1[::37] ^= 1                                             # off-curve first arguments: their products are 1
    assert np.array_equal(hip_a.element_prod_pairing(g1, g2, k), T.element_prod_pairing(g1, g2, k))
    T.clear()


def test_wave_products_on_the_reference_vectors(hip_a):
    for name in ("a_prod2x8.vec", "a_prod3x10_edge.vec", "a_prod16x4.vec", "a_prodfull3x4.vec"):
        v = golden(name)
        assert np.array_equal(hip_a.element_prod_pairing(v.g1, v.g2, v.k), v.gt), name
