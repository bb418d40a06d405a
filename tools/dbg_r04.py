import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pbc_amd
from conftest import golden, _param
for pname, group in (("a", 1), ("d159", 1), ("d159", 2)):
    H = pbc_amd.Pairing(_param(pname))
    v = golden("%s_pp%dpow12.vec" % (pname, group))
    pp = H.element_pp_init(group, v.g1[0])
    got = pp.pow_zn(v.g2)
    bad = [i for i in range(len(got)) if not np.array_equal(got[i], v.gt[i])]
    print(pname, group, "bad rows", bad, "of", len(got))
    want2 = H.element_mul_zn(group, v.g1, v.g2)
    print("  mul_zn vs vector:", np.array_equal(want2, v.gt))
    for i in bad[:3]:
        print("  z", v.g2[i].tobytes().hex())
        print("  got ", got[i][:24].tobytes().hex(), " want", v.gt[i][:24].tobytes().hex())
    S = pbc_amd.Pairing(_param(pname) + "hip_group_slow 1\n")
    pp2 = S.element_pp_init(group, v.g1[0])
    print("  slow-object pp:", np.array_equal(pp2.pow_zn(v.g2), v.gt))
    # single-byte scalars: which table entries are wrong
    zl = H.length_in_bytes_Zr
    Z = np.zeros((zl * 4, zl), np.uint8)
    for row in range(zl):
        for j, w in enumerate((1, 2, 129, 255)):
            Z[row * 4 + j, zl - 1 - row] = w
    B = np.tile(v.g1[0], (len(Z), 1))
    g = pp.pow_zn(Z); w_ = H.element_mul_zn(group, B, Z)
    badr = [(i // 4, (1, 2, 129, 255)[i % 4]) for i in range(len(Z)) if not np.array_equal(g[i], w_[i])]
    print("  single-digit scalars wrong (row, digit):", badr[:20], len(badr))
