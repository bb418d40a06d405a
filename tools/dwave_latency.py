"""Latency of small batches, d159.param (or DW_PARAM=<another type d parameter file>, DW_PARAM=f: the type f wave kernel): one pairing per wavefront (pairing_dw.cuh) against the one-pairing-per-lane kernel;
device buffers, events around the call, median of 7 after 2 warm-ups.
   python tools/dwave_latency.py [sizes...]              element_pairing
   python tools/dwave_latency.py prod K [sizes...]       element_prod_pairing, K terms (sizes: products)
   python tools/dwave_latency.py pp [sizes...]           pairing_pp_apply"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pbc_amd  # noqa: E402
from conftest import golden, _param  # noqa: E402

PNAME = os.environ.get("DW_PARAM", "d159")                          # DW_PARAM=d278027-190-181: a six-word field
v = golden("d_chain256.vec" if PNAME == "d159" else "f_chain128.vec" if PNAME == "f" else "g149_chain64.vec" if PNAME == "g149" else PNAME + "_rand12.vec")
args = sys.argv[1:]
mode, k = "pairing", 1
if args and args[0] == "prod":
    mode, k, args = "prod", int(args[1]), args[2:]
elif args and args[0] == "pp":
    mode, args = "pp", args[1:]
sizes = [int(x) for x in args] or [1, 16, 256, 1024, 2048, 4096, 8192, 16384]
P = {"wave": pbc_amd.Pairing(_param(PNAME) + "hip_dwave_max 100000000\nhip_fwave_max 100000000\n"), "lane": pbc_amd.Pairing(_param(PNAME) + "hip_dwave_max 0\nhip_fwave_max 0\n")}
pps = {name: H.pp_init(v.g1[3]) for name, H in P.items()} if mode == "pp" else {}
what = {"pairing": "wavefront per pairing", "prod": "wavefront per term (%d terms)" % k, "pp": "pairing_pp_apply, wavefront each"}[mode]
for n in sizes:
    i = np.arange(n * k) % v.n
    g1 = torch.from_numpy(np.ascontiguousarray(v.g1[i])).cuda()
    g2 = torch.from_numpy(np.ascontiguousarray(v.g2[(i * 5 + 1) % v.n])).cuda()
    row, outs = {}, {}
    for name, H in P.items():
        out = torch.empty((n, H.length_in_bytes_GT), dtype=torch.uint8, device="cuda")
        ts = []
        for rep in range(9):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s = torch.cuda.current_stream().cuda_stream
            a.record()
            if mode == "pairing":
                H.element_pairing_dev(out.data_ptr(), g1.data_ptr(), g2.data_ptr(), n, s)
            elif mode == "prod":
                H.element_prod_pairing_dev(out.data_ptr(), g1.data_ptr(), g2.data_ptr(), n, k, s)
            else:
                pps[name].apply_dev(out.data_ptr(), g2.data_ptr(), n, s)
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        row[name] = float(np.median(ts[2:]))
        outs[name] = out.cpu().numpy()
    print("n = %6d   %s %8.3f ms  (%9.0f /s)    lane %8.3f ms  (%9.0f /s)    same bytes: %s" %
          (n, what, row["wave"], n / row["wave"] * 1e3, row["lane"], n / row["lane"] * 1e3, np.array_equal(outs["wave"], outs["lane"])), flush=True)
for pp in pps.values():
    pp.clear()
