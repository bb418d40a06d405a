"""CPU tests of the drop-in boundary: libpbc_hip.so loads, exports exactly what
include/pbc_hip.h declares, and its host logic (parameter parsing, lengths, error
behaviour of pairing_init_set_buf, ecc/pairing.c:88-98) works without a GPU.
No compute entry point is exercised here."""
import ctypes
import os
import re

import pytest

import pbc_amd
from conftest import ROOT


def test_library_builds_and_loads():
    pbc_amd.build()
    assert os.path.exists(pbc_amd.LIB_PATH)
    pbc_amd.lib()


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "pbc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pbc_hip_\w+)\s*\(", hdr))
    assert declared == set(pbc_amd.EXPORTS)
    L = ctypes.CDLL(pbc_amd.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name


def test_every_entry_point_cites_the_reference():
    hdr = open(os.path.join(ROOT, "include", "pbc_hip.h")).read()
    for ref in ("include/pbc_pairing.h:141", "include/pbc_pairing.h:153", "ecc/pairing.c:88",
                "ecc/a_param.c:1053", "ecc/a_param.c:1283", "arith/montfp.c"):
        assert ref in hdr, ref


def test_init_parses_type_a_lengths(a_param_text):
    p = pbc_amd.Pairing(a_param_text)
    assert p.type == "a"
    # pairing_length_in_bytes_* for a.param (128/128/128, Fq 64)
    assert (p.length_in_bytes_G1, p.length_in_bytes_G2, p.length_in_bytes_GT) == (128, 128, 128)
    assert p.length_in_bytes_Fq == 64
    assert p.algorithmic_macs_per_unit(1) == 4392 * (2 * 16 * 16 + 16)
    p.clear()


@pytest.mark.parametrize("bad", [
    "",                                   # no type
    "type z\nq 7\n",                      # unknown type
    "type a\nq 11\n",                     # missing keys
    "type a\nq 12\nh 1\nr 1\nexp2 3\nexp1 1\nsign1 1\nsign0 1\n",   # even / tiny q
])
def test_init_failure_returns_nonzero_with_message(bad):
    with pytest.raises(pbc_amd.PbcHipError) as e:
        pbc_amd.Pairing(bad or "\n")
    assert str(e.value)


def test_no_cpu_fallback_in_product():
    """The product path must never import or link the oracle."""
    for root, _, files in os.walk(os.path.join(ROOT, "pbc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".c", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "liboracle" not in txt and "pbc_oracle" not in txt, f
