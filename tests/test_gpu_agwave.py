"""GPU tests (-m gpu) of the wave kernels of type a1, of type a parameters outside the 512-bit Solinas fast path (pairing_aw.cuh
with AG<N>, round 6) and of type e (pairing_ew.cuh, on the same routines): small batches of a1.param / a_160_1024 / a_160_512_mm / a_160_256 calls -- element_pairing, element_prod_pairing,
pairing_pp_apply -- run one wavefront (four for the smallest batches) per pairing / per TERM; the bytes are those of the
one-pairing-per-lane kernels, of the reference's vectors and of the C restatement.  Parameter sets whose q does not fill its top
limb (a_160_500, a_224_768, a1_200 ...) keep the lane kernels: the object reports which route it has."""
import os

import numpy as np
import pytest

from conftest import golden, _param

pytestmark = pytest.mark.gpu

WAVE_SETS = ["a1", "a_160_1024", "a_160_256", "a_160_512_mm"]   # (a_160_512_mm: a 512-bit q, i.e. the fast path's own wave kernels -- the same checks)
E_SETS = ["e", "e_160_400"]                                 # type e (pairing_ew.cuh): no pairing_pp (e_param.c installs none)
FILES = {"a1": ("a1_rand6.vec", "a1_edge6.vec", "a1_prod3x3_edge.vec", "a1_chain8.vec"),
         "a_160_1024": ("a_160_1024_rand4.vec", None, "a_160_1024_prod3x3_edge.vec", "a_160_1024_rand4.vec"),
         "a_160_512_mm": ("a_160_512_mm_rand6.vec", None, "a_160_512_mm_prod3x4_edge.vec", "a_160_512_mm_rand6.vec"),
         "a_160_256": ("a_160_256_rand6.vec", None, "a_160_256_prod3x4_edge.vec", "a_160_256_rand6.vec"),
         "e": ("e_rand6.vec", "e_edge6.vec", "e_prod3x3_edge.vec", "e_chain8.vec"),
         "e_160_400": ("e_160_400_rand6.vec", None, "e_160_400_prod3x4_edge.vec", "e_160_400_rand6.vec")}


def _lane(pname):
    import pbc_amd
    return pbc_amd.Pairing(_param(pname) + "hip_wave_max 0\n")          # never the wave kernels


@pytest.mark.parametrize("pname", WAVE_SETS + E_SETS)
def test_wave_kernels_match_the_reference_vectors(hips, pname):
    rand, edge, prod, _ = FILES[pname]
    for name in (rand, edge):
        if name:
            v = golden(name)
            assert np.array_equal(hips[pname].element_pairing(v.g1, v.g2), v.gt), name
    w = golden(prod)
    assert np.array_equal(hips[pname].element_prod_pairing(w.g1, w.g2, w.k), w.gt)


@pytest.mark.parametrize("pname", WAVE_SETS + E_SETS)
@pytest.mark.parametrize("n", [1, 5, 70, 1025])
def test_wave_kernel_equals_the_lane_kernel(hips, pname, n):
    """both shapes (16-word fields: four wavefronts per pairing up to hip_wave4_max = 1024 units, one above), off-curve arguments included"""
    v = golden(FILES[pname][3])
    i = np.arange(n)
    g1, g2 = np.ascontiguousarray(v.g1[(i * 3 + 1) % v.n]), np.ascontiguousarray(v.g2[(i * 7 + 2) % v.n])
    g1[::41, 5] ^= 1                                            # off-curve first arguments: the identity of GT
    g2[3::17, 9] ^= 2
    lane = _lane(pname)
    m = min(n, 96)                                              # (a lane needs 0.2 s for an a1.param pairing whatever the batch)
    got = hips[pname].element_pairing(g1, g2)
    assert np.array_equal(got[:m], lane.element_pairing(g1[:m], g2[:m]))
    if n > m:                                                   # (one wavefront per unit above hip_wave4_max:) the same units in a four-wavefront launch
        assert np.array_equal(got[m:m + 64], hips[pname].element_pairing(g1[m:m + 64], g2[m:m + 64]))
    lane.clear()


@pytest.mark.parametrize("pname", WAVE_SETS + E_SETS)
def test_wave_kernel_on_cross_pairs_against_the_c_restatement(hips, oracles, pname):
    v = golden(FILES[pname][3])
    rng = np.random.default_rng(5)
    i, j = rng.integers(0, v.n, 6), rng.integers(0, v.n, 6)
    assert np.array_equal(hips[pname].element_pairing(v.g1[i], v.g2[j]), oracles[pname].pairing_batch(v.g1[i], v.g2[j]))


@pytest.mark.parametrize("pname", WAVE_SETS + E_SETS)
@pytest.mark.parametrize("n,k", [(1, 2), (3, 5), (2, 16), (300, 4)])
def test_products_on_wavefronts(hips, oracles, pname, n, k):
    v = golden(FILES[pname][3])
    i = np.arange(n * k)
    g1, g2 = np.ascontiguousarray(v.g1[(i * 5 + 3) % v.n]), np.ascontiguousarray(v.g2[(i * 11 + 1) % v.n])
    if n > 2:
        g1[k + 1, 3] ^= 4                                         # product 1: a term off the curve -> the identity
    got = hips[pname].element_prod_pairing(g1, g2, k)
    m = min(n, 3 if k <= 5 else 1)
    assert np.array_equal(got[:m], oracles[pname].prod_pairing_batch(g1[:m * k], g2[:m * k], k))
    if n > 2:
        fb = got.shape[1] // (1 if pname in E_SETS else 2)     # (GT of type e is F_q itself)
        one = np.zeros(got.shape[1], np.uint8)
        one[fb - 1] = 1
        assert np.array_equal(got[1], one)
    # the product of the single pairings (GT products on the device); product 1 has the off-curve term
    H = hips[pname]
    mm = min(n, 8)
    singles = H.element_pairing(g1[:mm * k], g2[:mm * k]).reshape(mm, k, -1)
    acc = singles[:, 0]
    for t in range(1, k):
        acc = H.element_mul_GT(np.ascontiguousarray(acc), np.ascontiguousarray(singles[:, t]))
    keep = [u for u in range(mm) if not (n > 2 and u == 1)]
    assert np.array_equal(got[:mm][keep], acc[keep])


@pytest.mark.parametrize("pname", WAVE_SETS)
@pytest.mark.parametrize("n", [1, 40, 1100])
def test_pairing_pp_apply_on_wavefronts(hips, pname, n):
    v = golden(FILES[pname][3])
    H = hips[pname]
    i = np.arange(n)
    Q = np.ascontiguousarray(v.g2[(i * 7 + 2) % v.n])
    if n > 2:
        Q[2, -1] ^= 1                                             # off the curve: the identity
    pp = H.pp_init(v.g1[1])
    got = pp.apply(Q)
    assert np.array_equal(got, H.element_pairing(np.tile(v.g1[1], (n, 1)), Q))
    lane = _lane(pname)
    pl = lane.pp_init(v.g1[1])
    m = min(n, 40)
    assert np.array_equal(got[:m], pl.apply(Q[:m]))
    pp.clear()
    pl.clear()
    lane.clear()
    bad = v.g1[2].copy()
    bad[1] ^= 8
    g = H.pp_init(bad).apply(Q[:2])
    one = np.zeros(g.shape[1], np.uint8)
    one[g.shape[1] // 2 - 1] = 1
    assert np.array_equal(g, np.tile(one, (len(g), 1)))


@pytest.mark.parametrize("pname", ["a_160_500", "a_224_768", "a1_200", "a_150_300_mm"])
def test_other_sizes_keep_the_lane_kernels(hips, pname):
    """q does not fill twelve bits of its top limb (or leaves less than ten bits of the radix): no table, the lane kernels
    answer -- the same vectors as ever"""
    from conftest import FILES_OF
    rand = FILES_OF[pname][0]
    v = golden(rand)
    assert np.array_equal(hips[pname].element_pairing(v.g1, v.g2), v.gt)


@pytest.mark.parametrize("pname", ["a1", "a_160_256", "e"])
def test_one_wavefront_per_unit_shape(hips, pname):
    """ "hip_wave4_max 0": every unit on ONE wavefront (two products at a time in one instruction stream) -- the default on the
    33-word fields is four wavefronts at every size, so this shape is named explicitly"""
    import pbc_amd
    v = golden(FILES[pname][3])
    H1 = pbc_amd.Pairing(_param(pname) + "hip_wave4_max 0\nhip_wave8_max 0\n")
    i = np.arange(5)
    g1, g2 = np.ascontiguousarray(v.g1[(i * 3 + 1) % v.n]), np.ascontiguousarray(v.g2[(i * 7 + 2) % v.n])
    g1[4, 5] ^= 1
    assert np.array_equal(H1.element_pairing(g1, g2), hips[pname].element_pairing(g1, g2))
    assert np.array_equal(H1.element_prod_pairing(g1[:4], g2[:4], 2), hips[pname].element_prod_pairing(g1[:4], g2[:4], 2))
    if pname not in E_SETS:
        p1, p4 = H1.pp_init(v.g1[1]), hips[pname].pp_init(v.g1[1])
        assert np.array_equal(p1.apply(g2), p4.apply(g2))
        p1.clear()
        p4.clear()
    H1.clear()


def test_cut_over_to_the_lane_kernels(hips):
    """batches above hip_wave_max (6144 on the 16-word fields) take the lane kernels: the same bytes on either side"""
    v = golden(FILES["a_160_256"][3])
    n = 6146
    i = np.arange(n)
    g1, g2 = np.ascontiguousarray(v.g1[(i * 3 + 1) % v.n]), np.ascontiguousarray(v.g2[(i * 7 + 2) % v.n])
    H = hips["a_160_256"]
    whole = H.element_pairing(g1, g2)                          # lanes
    assert np.array_equal(whole[:6144], H.element_pairing(g1[:6144], g2[:6144]))   # wavefronts
    assert np.array_equal(whole[6144:], H.element_pairing(g1[6144:], g2[6144:]))


@pytest.mark.skipif(not os.environ.get("PBC_TEST_WAVE8"), reason="the eight-wavefront route is opt-in (hip_wave8_max N; PBC_TEST_WAVE8=1 runs this test): see host_params.h ag_aux_build")
@pytest.mark.parametrize("pname", ["a1", "a_160_1024", "a_160_256"])
def test_eight_wavefronts_per_unit_shape(hips, pname):
    """ "hip_wave8_max 256": up to that many terms a unit gets EIGHT wavefronts and the three-round Miller loop (miller_loop_p: the line
    of a step multiplied in beside the next step's first products); the default keeps four and the five-round loop -- the same bytes,
    off-curve arguments included, for pairings and for products (whose terms take the same loop).  The CPU suite runs the same
    schedule on the host mirror (tests/test_hostsim.py)"""
    import pbc_amd
    v = golden(FILES[pname][3])
    H4 = pbc_amd.Pairing(_param(pname) + "hip_wave8_max 256\n")
    i = np.arange(9)
    g1, g2 = np.ascontiguousarray(v.g1[(i * 3 + 1) % v.n]), np.ascontiguousarray(v.g2[(i * 7 + 2) % v.n])
    g1[4, 5] ^= 1
    assert np.array_equal(H4.element_pairing(g1, g2), hips[pname].element_pairing(g1, g2))
    assert np.array_equal(H4.element_prod_pairing(g1[:8], g2[:8], 4), hips[pname].element_prod_pairing(g1[:8], g2[:8], 4))
    H4.clear()
