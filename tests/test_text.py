"""CPU tests of the text formats (include/pbc_hip.h pbc_hip_element_snprint / _set_str /
pbc_hip_param_snprint) against what the reference itself prints: tests/golden/*_text8.txt hold element_to_bytes records
with their element_snprint text and the lines of pbc_param_out_str, written by `ref_tool text` (oracle/ref_harness.c,
tests/golden/make_golden.sh).  Host-side string handling: no GPU involved."""
import os

import numpy as np
import pytest

import pbc_amd
from conftest import ROOT, _param

SETS = ["a", "a1", "d159", "e", "f", "g149", "d201"]


def _load(name):
    params, elems = [], []
    for line in open(os.path.join(ROOT, "tests", "golden", name + "_text8.txt")):
        line = line.rstrip("\n")
        if line.startswith("P "):
            params.append(line[2:])
        else:
            g, hx, txt = line.split(" ", 2)
            elems.append((int(g), np.frombuffer(bytes.fromhex(hx), np.uint8), txt))
    return params, elems


@pytest.fixture(scope="module")
def objs():
    return {n: pbc_amd.Pairing(_param(n)) for n in SETS}


@pytest.mark.parametrize("name", SETS)
def test_param_snprint_is_pbc_param_out_str(objs, name):
    params, _ = _load(name)
    assert objs[name].param_snprint() == "\n".join(params) + "\n"


@pytest.mark.parametrize("name", SETS)
def test_element_snprint_matches_the_reference(objs, name):
    P = objs[name]
    _, elems = _load(name)
    assert {g for g, _, _ in elems} == {0, 1, 2, 3} and any(t == "O" for _, _, t in elems)
    for g, rec, txt in elems:
        if txt == "O" and name in ("a", "a1"):
            # O has no wire format of its own: the fixture writes zeros, and on y^2 = x^3 + x the zero record is the
            # point (0, 0), which element_from_bytes accepts and the reference prints as such
            txt = "[0, 0]"
        got, full = P.element_snprint(g, rec)
        assert (got, full) == (txt, len(txt)), (g, txt[:40])
    # snprintf semantics: truncation keeps the full length as the return value
    g, rec, txt = max(elems, key=lambda e: len(e[2]))
    assert P.element_snprint(g, rec, size=10) == (txt[:9], len(txt))
    assert P.element_snprint(g, rec, size=0) == ("", len(txt))
    assert P.element_snprint(g, rec, size=len(txt) + 1) == (txt, len(txt))


@pytest.mark.parametrize("name", SETS)
def test_element_set_str_round_trips_the_reference_text(objs, name):
    P = objs[name]
    _, elems = _load(name)
    for g, rec, txt in elems:
        got, used = P.element_set_str(g, txt)
        assert used == len(txt) and np.array_equal(got, rec), (g, txt[:40])
        got, used = P.element_set_str(g, "  " + txt.replace(",", " ,  ") + "tail")     # blanks as the reference's parsers skip them
        assert used == len("  " + txt.replace(",", " ,  ")) and np.array_equal(got, rec), (g, txt[:40])


def test_element_set_str_bases_reduction_and_errors(objs):
    P = objs["a"]
    q = int([l for l in _load("a")[0] if l.startswith("q ")][0][2:])
    r = int([l for l in _load("a")[0] if l.startswith("r ")][0][2:])
    nb = P.length_in_bytes_Zr
    for text, base, want, used in (("ff", 16, 255, 2), ("FFzz", 16, 255, 2), ("1 0 1", 2, 5, 5), ("12x", 0, 12, 2), ("z", 36, 35, 1),
                                   (str(r + 7), 10, 7, len(str(r + 7))), ("", 10, 0, 0), ("9", 8, 0, 0)):
        rec, n = P.element_set_str(0, text, base)
        assert (int.from_bytes(rec.tobytes(), "big"), n) == (want, used), text
    assert P.element_set_str(0, "5", 1)[1] == 0 and P.element_set_str(0, "5", 37)[1] == 0      # pbc_mpz_set_str: bad base
    # GT of type a is F_q^2: "[x, y]", coordinates reduced mod q
    rec, n = P.element_set_str(3, "[%d, 3]" % (q + 2))
    assert n == len("[%d, 3]" % (q + 2)) and int.from_bytes(rec[:64].tobytes(), "big") == 2 and int.from_bytes(rec[64:].tobytes(), "big") == 3
    for bad in ("1, 2]", "[1 2]", "[1, 2", "[1, 2, 3]"):
        rec, n = P.element_set_str(3, bad)
        assert n == 0 and not rec.any(), bad
    # points: "O", a point of the curve, a pair off the curve (curve_set_str: O, returns 0)
    rec, n = P.element_set_str(1, " O")
    assert n == 2 and not rec.any()
    _, elems = _load("a")
    g1 = [e for e in elems if e[0] == 1 and e[2] != "O"][0]
    x, y = g1[2][1:-1].split(", ")
    rec, n = P.element_set_str(1, "[%s, %d]" % (x, (int(y) + 1) % q))
    assert n == 0 and not rec.any()
    rec, n = P.element_set_str(1, "[%s, %d]" % (x, q - int(y)))                  # -P is on the curve
    assert n > 0 and int.from_bytes(rec[64:].tobytes(), "big") == q - int(y)
    # a record off the curve prints as O (curve_from_bytes), (0, 0) on y^2 = x^3 + x prints as a point
    off = g1[1].copy()
    off[-1] ^= 1
    assert P.element_snprint(1, off) == ("O", 1)
    assert P.element_snprint(1, np.zeros(128, np.uint8)) == ("[0, 0]", 6)
    assert objs["d159"].element_snprint(1, np.zeros(40, np.uint8)) == ("O", 1)
    with pytest.raises(pbc_amd.PbcHipError):
        P.element_snprint(4, np.zeros(128, np.uint8))
