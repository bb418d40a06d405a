"""CPU tests of round 5's group surface (element_add / sub / neg / double on G1 / G2, Z_r arithmetic, element_pow2_zn /
element_pow3_zn): the reference's fixtures (tests/golden/*.rec, written by oracle/_ref/ref_tool gops / zrops / pow23)
against (a) the plain-C oracle and (b) the lane bodies of the HIP kernels compiled for the host (tests/hostsim) -- the
same source pbc_hip_group2.hip launches."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, _param, PARAM_OF

hostsim = pytest.importorskip("hostsim")


def rec(name):
    return oracle.Rec(os.path.join(GOLDEN, name)).arrays


@pytest.fixture(scope="module")
def sims():
    class Lazy(dict):
        def __missing__(self, t):
            self[t] = hostsim.HostSim(_param(PARAM_OF.get(t, t)))
            return self[t]
    return Lazy()


GOPS = [("a", 1), ("a1", 1), ("e", 1), ("d159", 1), ("d159", 2), ("f", 1), ("f", 2), ("g149", 2), ("d201", 2), ("f_256", 2)]
ZROPS = ["a", "a1", "d159", "d224", "f", "f_256", "g149"]
POW23 = [("a", 1), ("a", 3), ("e", 1), ("d159", 1), ("d159", 2), ("d159", 3), ("f", 1), ("f", 2), ("f", 3), ("g149", 3), ("a1", 3)]
ZR_FILE_OPS = {1: 2, 2: 3, 0: 4, 3: 5, 4: 6, 6: 7, 5: 8, 7: 9}      # op code -> array index of its result


@pytest.mark.parametrize("key,group", [g for g in GOPS if g[1] == 1])
def test_oracle_group_law_matches_reference(oracles, key, group):
    A, B, ADD, SUB, NEG, DBL = rec("%s_gops%d.rec" % (key, group))
    O = oracles[key]
    assert np.array_equal(O.g1_op(0, A, B), ADD)
    assert np.array_equal(O.g1_op(1, A, B), SUB)
    assert np.array_equal(O.g1_op(2, A), NEG)
    assert np.array_equal(O.g1_op(3, A), DBL)


@pytest.mark.parametrize("key", ZROPS)
def test_oracle_zr_arithmetic_matches_reference(oracles, key):
    R = rec(key + "_zrops.rec")
    O = oracles[key]
    for op, idx in ZR_FILE_OPS.items():
        binary = op in (0, 1, 2, 7)
        assert np.array_equal(O.zr_op(op, R[0], R[1] if binary else None), R[idx]), op
    assert np.array_equal(O.zr_from_hash(R[10], R[0].shape[1]), R[11])


@pytest.mark.parametrize("key,group", [p for p in POW23 if p[1] != 2 and p[0] != "g149"])
def test_oracle_multi_exponentiation_matches_reference(oracles, key, group):
    A1, A2, A3, N1, N2, N3, P2, P3 = rec("%s_pow23g%d.rec" % (key, group))
    O = oracles[key]
    assert np.array_equal(O.pow_multi(group, [A1, A2], [N1, N2]), P2)
    assert np.array_equal(O.pow_multi(group, [A1, A2, A3], [N1, N2, N3]), P3)


@pytest.mark.parametrize("key,group", GOPS)
def test_group_law_kernel_source_on_host(sims, key, group):
    A, B, ADD, SUB, NEG, DBL = rec("%s_gops%d.rec" % (key, group))
    S = sims[key]
    assert np.array_equal(S.affine_op(0, group, A, B), ADD)
    assert np.array_equal(S.affine_op(1, group, A, B), SUB)
    assert np.array_equal(S.affine_op(2, group, A), NEG)
    assert np.array_equal(S.affine_op(3, group, A), DBL)
    # off-curve records are O (curve_from_bytes): O + B = B, A - O' = A
    bad = A.copy()
    bad[:, -1] ^= 1
    assert np.array_equal(S.affine_op(0, group, bad[:4], B[:4]), B[:4])


@pytest.mark.parametrize("key", ZROPS)
def test_zr_kernel_source_on_host(sims, key):
    R = rec(key + "_zrops.rec")
    S = sims[key]
    for op, idx in ZR_FILE_OPS.items():
        binary = op in (0, 1, 2, 7)
        assert np.array_equal(S.zr_op(op, R[0], R[1] if binary else None), R[idx]), op
    assert np.array_equal(S.zr_op(8, R[10], None, R[10].shape[1]), R[11])


@pytest.mark.parametrize("key,group", POW23)
def test_multi_exponentiation_kernel_source_on_host(sims, key, group):
    A1, A2, A3, N1, N2, N3, P2, P3 = rec("%s_pow23g%d.rec" % (key, group))
    S = sims[key]
    S.fallbacks()
    assert np.array_equal(S.multi(group, [A1, A2], [N1, N2]), P2)
    assert np.array_equal(S.multi(group, [A1, A2, A3], [N1, N2, N3]), P3)
    if group != 3:
        # the fast pass (one doubling + one incomplete addition per bit; G1 of type a and of the five-word fields: the joint
        # limb-form window ladder) serves the ordinary rows; the rows with a2 = a1 or a2 = -a1 (and, on the window ladder,
        # with a zero scalar) are reported and take the complete routine -- which alone gives the same bytes
        assert 0 < S.fallbacks() <= 8
        S.group_mode(1)
        assert np.array_equal(S.multi(group, [A1, A2], [N1, N2]), P2)
        assert np.array_equal(S.multi(group, [A1, A2, A3], [N1, N2, N3]), P3)
        S.group_mode(0)
