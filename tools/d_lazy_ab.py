#!/usr/bin/env python3
"""A/B of the type d signed-limb experiment (pbc_amd/csrc/pairing_d_lazy.cuh) on a GPU: parity against the
reference vectors and kernel throughput (element_pairing and pairing_pp_apply) with PBC_HIP_D_LAZY=0 / 1, each in a fresh process (the switch is read
once per process).  Prints one JSON line per variant.

  python tools/d_lazy_ab.py [log2 n]           # default 2^18 pairings per launch
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, pbc_amd
from conftest import golden, _param
P = pbc_amd.Pairing(_param("d159"))
ok = True
for name in ("d_rand32.vec", "d_edge20.vec", "d_prod16x4.vec", "d_prod3x10_edge.vec"):
    v = golden(name)
    got = P.element_prod_pairing(v.g1, v.g2, v.k) if v.k > 1 else P.element_pairing(v.g1, v.g2)
    ok &= bool(np.array_equal(got, v.gt))
n = 1 << {lg}
v = golden("d_rand32.vec")
reps = (n + v.n - 1) // v.n
g1 = torch.from_numpy(np.tile(v.g1, (reps, 1))[:n].copy()).cuda()
g2 = torch.from_numpy(np.tile(v.g2, (reps, 1))[:n].copy()).cuda()
gt = torch.empty((n, v.gt.shape[1]), dtype=torch.uint8, device="cuda")
for _ in range(2):
    P.element_pairing_dev(gt.data_ptr(), g1.data_ptr(), g2.data_ptr(), n)
torch.cuda.synchronize()
t = time.perf_counter()
K = 5
for _ in range(K):
    P.element_pairing_dev(gt.data_ptr(), g1.data_ptr(), g2.data_ptr(), n)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / K
same = bool(np.array_equal(gt[: v.n].cpu().numpy(), v.gt))
# pairing_pp_init / pairing_pp_apply with the first fixture point as the fixed argument
pp = P.pp_init(v.g1[0])
want = P.element_pairing(np.tile(v.g1[0], (v.n, 1)), v.g2)
pp_ok = bool(np.array_equal(pp.apply(v.g2), want))
for _ in range(2):
    pp.apply_dev(gt.data_ptr(), g2.data_ptr(), n)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(K):
    pp.apply_dev(gt.data_ptr(), g2.data_ptr(), n)
torch.cuda.synchronize()
dtp = (time.perf_counter() - t) / K
print(json.dumps({{"lazy": os.environ.get("PBC_HIP_D_LAZY", "0"), "parity": ok and same and pp_ok, "n": n,
                  "ms_per_launch": round(dt * 1e3, 3), "pairings_per_s": round(n / dt),
                  "pp_ms_per_launch": round(dtp * 1e3, 3), "pp_applies_per_s": round(n / dtp)}}))
"""


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    rc = 0
    for lazy in ("0", "1"):
        env = dict(os.environ, PBC_HIP_D_LAZY=lazy)
        out = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, lg=lg)], env=env, capture_output=True, text=True, timeout=900)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""
        print(line or json.dumps({"lazy": lazy, "error": out.stderr[-400:]}))
        if not line or not json.loads(line).get("parity"):
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
