"""CPU tests of the KERNEL SOURCE itself: pbc_amd/csrc/*.cuh compiled for the host
(tests/hostsim: same limb arithmetic, towers, Miller loops and final exponentiations as the
HIP kernels, one lane at a time) against the reference's golden vectors.  This is a debug
mirror for GPU-less bring-up, not a product path: libpbc_hip.so never contains it."""
import numpy as np
import pytest

from conftest import golden, _param, PARAM_OF

hostsim = pytest.importorskip("hostsim")


@pytest.fixture(scope="module")
def sims():
    class Lazy(dict):
        def __missing__(self, t):
            self[t] = hostsim.HostSim(_param(PARAM_OF[t]))
            return self[t]
    return Lazy()


@pytest.mark.parametrize("name,count", [
    ("a_kat.vec", 1), ("a_rand32.vec", 6), ("a_edge20.vec", 20), ("a_prod2x8.vec", 4), ("a_prod3x10_edge.vec", 10),
    ("d_rand32.vec", 6), ("d_edge20.vec", 20), ("d_prod16x4.vec", 2), ("d_prod3x10_edge.vec", 10),
    ("f_rand16.vec", 3), ("f_edge10.vec", 10), ("f_prod3x5_edge.vec", 5),
])
def test_kernel_source_on_host_matches_reference(sims, name, count):
    v = golden(name)
    n = min(count, v.n)
    out = sims[v.type].prod_pairing(v.g1[:n * v.k], v.g2[:n * v.k], v.k)
    assert np.array_equal(out, v.gt[:n])


@pytest.mark.parametrize("t,q", [("a", None), ("d", 625852803282871856053922297323874661378036491717)])
def test_kernel_fq_ops_on_host(sims, oracles, t, q):
    if q is None:
        q = int([l.split()[1] for l in _param("a").splitlines() if l.startswith("q ")][0])
    nb = sims[t].len1 // 2
    rng = np.random.default_rng(2)
    xs = [int.from_bytes(rng.bytes(nb), "big") % q for _ in range(40)] + [1, q - 1, 2 ** (8 * nb) - 1]
    ys = [int.from_bytes(rng.bytes(nb), "big") % q for _ in range(40)] + [q - 1, q - 1, 2 ** (8 * nb) - 1]
    A = np.stack([np.frombuffer(x.to_bytes(nb, "big"), np.uint8) for x in xs])
    B = np.stack([np.frombuffer(y.to_bytes(nb, "big"), np.uint8) for y in ys])
    for op in range(7):
        assert np.array_equal(sims[t].fq_op(op, A, B), oracles[t].fq_op(op, A, B)), op


def test_type_f_generic_hard_part_on_host():
    """f.param is a BN curve and takes the x-chain; the generic fixed-window power over
    (q^4-q^2+1)/r (any Type-F parameters) must give the same bytes."""
    v = golden("f_rand16.vec")
    sim = hostsim.HostSim(_param("f") + "hip_no_bn 1\n")
    assert np.array_equal(sim.prod_pairing(v.g1[:2], v.g2[:2], 1), v.gt[:2])


def test_pairing_pp_on_host(sims, oracles):
    """pairing_pp_init + pairing_pp_apply give element_pairing's bytes (a_param.c:149-220, :317-360)."""
    v = golden("a_chain1024.vec")
    P = v.g1[5]
    Q = v.g2[:6].copy()
    Q[2, 127] ^= 1                                         # off-curve second argument -> 1
    want = oracles["a"].pairing_batch(np.tile(P, (6, 1)), Q)
    assert np.array_equal(sims["a"].pp(P, Q), want)
    bad = P.copy(); bad[100] ^= 4                          # off-curve first argument -> all 1
    one = np.zeros(128, np.uint8); one[63] = 1
    assert np.array_equal(sims["a"].pp(bad, Q), np.tile(one, (6, 1)))
