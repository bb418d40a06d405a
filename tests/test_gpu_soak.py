"""SURVEY 8d's full parity run: 2^16 uniformly random (P, Q) per BASELINE configuration and 2^12 random 16-term products,
computed by the UNMODIFIED reference (oracle/_ref/ref_tool soak, every usable host core) and compared with the GPU's
bytes in full -- plus crafted inputs the chain fixtures cannot contain (ref_harness.c cmd_soak): points whose x
coordinate is q - 1, 1, (q +- 1) / 2, ..., or whose Montgomery residue (the radix of the kernels' 29-bit limb form) has
all-ones / all-zero / alternating limbs or a single bit on a limb boundary, on the curve and on the twists (every
coefficient of x, or only the first), and records whose coordinates are written as v + t q >= q.

The seed is PBC_SOAK_SEED (default fixed, so that a failure reproduces); sizes PBC_SOAK_LOG2 / PBC_SOAK_LOG2_PROD.  Each
case leaves a summary in gpurun_out/soak_<case>.json when that directory exists (-> profiles/r06_soak.json).
A CPU-only case pins the generator itself: its crafted and random units against the C restatement (oracle/)."""
import json
import os
import time

import numpy as np
import pytest

import oracle
from conftest import ROOT, PARAM_OF

SEED = int(os.environ.get("PBC_SOAK_SEED", "20260930"))
LOG2 = int(os.environ.get("PBC_SOAK_LOG2", "16"))
LOG2_PROD = int(os.environ.get("PBC_SOAK_LOG2_PROD", "12"))
needs_ref = pytest.mark.skipif(not os.path.exists(oracle.REF_TOOL), reason="oracle/_ref/ref_tool (the compiled reference) is not built")


def _param_path(name):
    return os.path.join(ROOT, "pbc_amd", "param", PARAM_OF.get(name, name) + ".param")


def _rbits(P):
    """bits of the Montgomery radix of the library's limb form: 29-bit limbs, L = ceil(32 N / 29) (fp.cuh Limbs29)"""
    n_words = -(-(P.length_in_bytes_G1 // 2 * 8) // 32)
    if n_words > 16:                                  # (above 512 bits the library runs 33 words: 38 limbs of 28 bits, fp.cuh Limbs29)
        return 28 * 38
    return 29 * (-(-32 * n_words // 29))


def _rbits_of(name):
    return {"a": 522, "d": 174, "f": 174, "g149": 174}[name]


def test_radix_table_matches_the_library_layout():
    # fixed_length_in_bytes 64 -> 16 words -> 18 limbs; 20 bytes -> 5 words -> 6 limbs
    for fb, want in ((64, 522), (20, 174)):
        n_words = -(-fb * 8 // 32)
        assert 29 * (-(-32 * n_words // 29)) == want


@needs_ref
@pytest.mark.parametrize("name", ["a", "d", "f", "g149"])
def test_soak_generator_agrees_with_the_c_restatement(name, oracles, tmp_path):
    """no GPU: a small soak file (random + every crafted unit) from the reference against oracle/pbc_oracle.c"""
    v, info = oracle.ref_soak(_param_path(name), 24, 1, 77, str(tmp_path / "s.vec"), _rbits_of(name), workers=4)
    assert info["crafted_units"] == v.n - 24 and info["noncanonical_coordinates"] > 0
    assert np.array_equal(oracles[name].pairing_batch(v.g1, v.g2), v.gt)
    w, _ = oracle.ref_soak(_param_path(name), 24, 1, 77, str(tmp_path / "t.vec"), _rbits_of(name), workers=3)
    assert np.array_equal(w.g1[24:], v.g1[24:]) and np.array_equal(w.gt[24:], v.gt[24:])     # the crafted block does not depend on the split


def _record(case, info, n, mismatches, t_ref, t_gpu):
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        info = dict(info, case=case, units_compared=int(n), mismatches=int(mismatches), gpu_s=round(t_gpu, 2), total_s=round(t_ref + t_gpu, 2),
                    usable_cores=oracle.usable_cores())
        with open(os.path.join(out, "soak_%s_seed%d.json" % (case, SEED)), "w") as fh:
            json.dump(info, fh)


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("name", ["a", "d", "f"])
def test_full_parity_on_uniformly_random_and_crafted_inputs(name, hips, tmp_path):
    P = hips[name]
    assert _rbits(P) == _rbits_of(name)
    t0 = time.time()
    v, info = oracle.ref_soak(_param_path(name), 1 << LOG2, 1, SEED, str(tmp_path / "soak.vec"), _rbits_of(name))
    t1 = time.time()
    got = P.element_pairing(v.g1, v.g2)
    bad = np.nonzero((got != v.gt).any(axis=1))[0]
    _record(name, info, v.n, len(bad), t1 - t0, time.time() - t1)
    assert len(bad) == 0, "units that differ from the reference: %s (first crafted unit is %d)" % (bad[:16], 1 << LOG2)
    # the crafted block also through the small-batch kernels (type a: one pairing per wavefront / workgroup)
    c0 = 1 << LOG2
    assert np.array_equal(P.element_pairing(v.g1[c0:c0 + 60], v.g2[c0:c0 + 60]), v.gt[c0:c0 + 60])


@pytest.mark.gpu
@needs_ref
def test_full_parity_on_uniformly_random_products(hips, tmp_path):
    P = hips["a"]
    t0 = time.time()
    v, info = oracle.ref_soak(_param_path("a"), 1 << LOG2_PROD, 16, SEED + 1, str(tmp_path / "soak.vec"), 522)
    t1 = time.time()
    got = P.element_prod_pairing(v.g1, v.g2, 16)
    bad = np.nonzero((got != v.gt).any(axis=1))[0]
    _record("a-prod16", info, v.n, len(bad), t1 - t0, time.time() - t1)
    assert len(bad) == 0, "products that differ from the reference: %s" % bad[:16]


WAVE_LOG2 = int(os.environ.get("PBC_SOAK_LOG2_WAVE", "12"))


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("name", ["d", "d278027-190-181", "d201", "d224", "f", "g149", "a1", "e"])
def test_wave_kernels_on_uniformly_random_and_crafted_inputs(name, hips, tmp_path):
    """the small-batch route of types d, f, g, a1 and e (one pairing / one TERM per wavefront -- per workgroup of four on a1.param --, pairing_dw.cuh,
    pairing_fw.cuh, pairing_gw.cuh, pairing_aw.cuh) on inputs no fixture holds: 2^12
    uniformly random pairs and the crafted block (limb patterns in the field's own radix: six, seven and eight limbs) in calls
    of at most 4096 units, 2^10 random four-term products, and pairing_pp_apply on the random second arguments -- against the
    unmodified reference"""
    import pbc_amd
    P = hips[name]
    rbits = _rbits(P)
    t0 = time.time()
    log2 = WAVE_LOG2 - (2 if name == "a1" else 0)     # (a1.param: 9 ms a pairing on a CPU core)
    v, info = oracle.ref_soak(_param_path(name), 1 << log2, 1, SEED + 2, str(tmp_path / "soak.vec"), rbits)
    w, _ = oracle.ref_soak(_param_path(name), 1 << (log2 - 2), 4, SEED + 3, str(tmp_path / "soakp.vec"), rbits)
    t1 = time.time()
    assert info["crafted_units"] > 100
    bad = 0
    # (type e: e_pairing's value f(Q + R) / f(R) depends on the auxiliary point R -- which the reference draws at random, e_param.c:869-870 --
    # as soon as an argument lies outside E[r]; the crafted points, taken from the whole curve, do: the random block and the records
    # written as v + t q are what has a reference value)
    judged = np.ones(v.n, bool)
    if name == "e":
        judged[(1 << log2):(1 << log2) + 6 * info["crafted_patterns"]] = False
    for a in range(0, v.n, 4096):
        got = P.element_pairing(v.g1[a:a + 4096], v.g2[a:a + 4096])
        bad += int(((got != v.gt[a:a + 4096]).any(axis=1) & judged[a:a + 4096]).sum())
    gotp = P.element_prod_pairing(w.g1, w.g2, 4)
    badp = int((gotp != w.gt).any(axis=1).sum())
    # pairing_pp_apply: e(P_0, Q_i) for the first 512 random Q_i = what the lane kernels give for the same pairs
    m, badq = (512 if name != "a1" else 128), 0
    if name not in ("f", "e"):                        # (types f and e have no pairing_pp routines: f_param.c / e_param.c install none)
        lane = pbc_amd.Pairing(open(_param_path(name)).read() + "hip_dwave_max 0\nhip_wave_max 0\n")
        pp = P.pp_init(v.g1[0])
        badq = int((pp.apply(v.g2[:m]) != lane.element_pairing(np.tile(v.g1[0], (m, 1)), v.g2[:m])).any(axis=1).sum())
        pp.clear()
        lane.clear()
    _record("wave-" + name, dict(info, products_of_4=int(w.n), pp_apply_units=m), v.n + w.n + m, bad + badp + badq, t1 - t0, time.time() - t1)
    assert bad == 0 and badp == 0 and badq == 0, (bad, badp, badq)
