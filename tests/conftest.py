import os
import sys

import pytest

# torch bundles its own HIP runtime: it must be the first one loaded into the process,
# otherwise (libpbc_hip.so pulling /opt/rocm's libamdhip64 first) torch later reports
# "No HIP GPUs are available".  The product itself never needs torch.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the HIP library and the C oracle normally travel with the tree; (re)build them when stale
    import pbc_amd
    import oracle
    pbc_amd.build()
    oracle.build()


@pytest.fixture(scope="session")
def a_param_text():
    with open(os.path.join(ROOT, "pbc_amd", "param", "a.param")) as fh:
        return fh.read()


@pytest.fixture(scope="session")
def oracle_a(a_param_text):
    import oracle
    return oracle.OraclePairing(a_param_text)


@pytest.fixture(scope="session")
def hip_a(a_param_text):
    """The product: libpbc_hip.so through its C-ABI.  No fallback of any kind."""
    import pbc_amd
    return pbc_amd.Pairing(a_param_text)


PARAM_OF = {"a": "a", "d": "d159", "f": "f"}
# the other type d parameter files the reference ships (param/): 175..224-bit q, 6 / 7 word fields
OTHER_D = ["d277699-175-167", "d278027-190-181", "d105171-196-185", "d201", "d224"]
# ... the type g (Freeman, k = 10) file: 149-bit q, F_q^5 / F_q^10 towers, and type a1
OTHER = OTHER_D + ["g149", "a1", "e"]
# param -> (random single pairings, edge cases incl. off-curve inputs, product with edge cases)
FILES_OF = {d: (d + "_rand12.vec", d + "_edge8.vec", d + "_prod3x4_edge.vec") for d in OTHER_D}
FILES_OF["g149"] = ("g149_rand16.vec", "g149_edge10.vec", "g149_prod3x4_edge.vec")
# type a1: 1033-bit p, composite group order n (a1.param)
FILES_OF["a1"] = ("a1_rand6.vec", "a1_edge6.vec", "a1_prod3x3_edge.vec")
# type e: k = 1, 1020-bit q, GT = F_q (e.param)
FILES_OF["e"] = ("e_rand6.vec", "e_edge6.vec", "e_prod3x3_edge.vec")


def key_of(vec_name):
    """fixture file name -> key for the oracles / hips / sims fixtures: "<param>_*.vec" for the
    extra parameter files, else the type letter (a_*, d_*, f_* = a.param, d159.param, f.param)"""
    prefix = vec_name.split("_")[0]
    return prefix if prefix in OTHER else prefix[0]


def _param(name):
    with open(os.path.join(ROOT, "pbc_amd", "param", name + ".param")) as fh:
        return fh.read()


def param_value(name, key):
    """integer value of a "key value" line of pbc_amd/param/<name>.param"""
    for line in _param(PARAM_OF.get(name, name)).splitlines():
        f = line.split()
        if len(f) == 2 and f[0] == key:
            return int(f[1])
    raise KeyError(key)



@pytest.fixture(scope="session")
def oracles():
    """type letter or parameter file name -> CPU oracle"""
    import oracle

    class Lazy(dict):
        def __missing__(self, t):
            self[t] = oracle.OraclePairing(_param(PARAM_OF.get(t, t)))
            return self[t]
    return Lazy()


@pytest.fixture(scope="session")
def hips():
    """type letter or parameter file name -> the product (libpbc_hip.so)"""
    import pbc_amd

    class Lazy(dict):
        def __missing__(self, t):
            self[t] = pbc_amd.Pairing(_param(PARAM_OF.get(t, t)))   # raises if the type is not built in
            return self[t]
    return Lazy()


def golden(name):
    import oracle
    return oracle.Vec(os.path.join(GOLDEN, name))
