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


# every type a / a1 / d / e / f / g parameter file the reference ships under param/: (type, G1, G2, GT, Zr bytes)
SHIPPED = {
    "a": ("a", 128, 128, 128, 20), "a1": ("1", 260, 260, 260, 128), "e": ("e", 256, 256, 128, 20),
    "f": ("f", 40, 80, 240, 20), "g149": ("g", 38, 190, 190, 19),
    "d159": ("d", 40, 120, 120, 20), "d201": ("d", 52, 156, 156, 23), "d224": ("d", 56, 168, 168, 28),
    "d105171-196-185": ("d", 50, 150, 150, 24), "d277699-175-167": ("d", 44, 132, 132, 21),
    "d278027-190-181": ("d", 48, 144, 144, 23),
}


@pytest.mark.parametrize("name", sorted(SHIPPED))
def test_init_parses_every_shipped_parameter_file(name):
    """pairing_init_set_buf on the host: type dispatch and pairing_length_in_bytes_* (no GPU needed)."""
    import os
    text = open(os.path.join(pbc_amd.PARAM_DIR, name + ".param")).read()
    p = pbc_amd.Pairing(text)
    t, l1, l2, lt, lz = SHIPPED[name]
    assert p.type == t
    assert (p.length_in_bytes_G1, p.length_in_bytes_G2, p.length_in_bytes_GT, p.length_in_bytes_Zr) == (l1, l2, lt, lz)
    assert p.algorithmic_macs_per_unit(1) > 0 and p.algorithmic_macs_per_unit(4) >= p.algorithmic_macs_per_unit(1)
    p.clear()


@pytest.mark.parametrize("bad", [
    "",                                   # no type
    "type z\nq 7\n",                      # unknown type
    "type a\nq 11\n",                     # missing keys
    "type a\nq 12\nh 1\nr 1\nexp2 3\nexp1 1\nsign1 1\nsign0 1\n",   # even / tiny q
    "type i\nm 97\nt 12\nn 1\nn2 1\n",  # eta_T pairing over GF(3^m): not built in
    "type d\nq 7\nr 3\na 1\nb 1\nk 4\nnqr 3\ncoeff0 1\ncoeff1 1\ncoeff2 1\nh 1\n",   # wrong embedding degree
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
