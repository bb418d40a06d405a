"""Latency of small batches on type a1 / generic type a parameter files (AG_PARAM=a1 | a_160_1024 | a_160_512_mm | a_160_256): the wave
kernels of pairing_aw.cuh with AG<N> (four wavefronts per unit up to hip_wave4_max, one above) against the one-pairing-per-lane
kernels; device buffers, events around the call, median of 5 after 2 warm-ups.
   python tools/agwave_latency.py [sizes...]              element_pairing
   python tools/agwave_latency.py prod K [sizes...]       element_prod_pairing, K terms (sizes: products)
   python tools/agwave_latency.py pp [sizes...]           pairing_pp_apply
LANE_MAX (default 4096): the lane kernels are timed up to this size only (0.2 s a launch on a1.param)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pbc_amd  # noqa: E402
from conftest import golden, _param  # noqa: E402

PNAME = os.environ.get("AG_PARAM", "a1")
W4 = os.environ.get("AG_WAVE4_MAX")
W8 = os.environ.get("AG_WAVE8_MAX")
v = golden({"a1": "a1_chain8.vec", "e": "e_chain8.vec", "a_160_1024": "a_160_1024_rand4.vec"}.get(PNAME, PNAME + "_rand6.vec"))
args = sys.argv[1:]
mode, k = "pairing", 1
if args and args[0] == "prod":
    mode, k, args = "prod", int(args[1]), args[2:]
elif args and args[0] == "pp":
    mode, args = "pp", args[1:]
sizes = [int(x) for x in args] or [1, 64, 1024, 4096]
lane_max = int(os.environ.get("LANE_MAX", "4096"))
P = {"wave": pbc_amd.Pairing(_param(PNAME) + "hip_wave_max 100000000\n" + ("hip_wave4_max %s\n" % W4 if W4 else "") + ("hip_wave8_max %s\n" % W8 if W8 else "")), "lane": pbc_amd.Pairing(_param(PNAME) + "hip_wave_max 0\n")}
pps = {name: H.pp_init(v.g1[1]) for name, H in P.items()} if mode == "pp" else {}
for n in sizes:
    i = np.arange(n * k) % v.n
    g1 = torch.from_numpy(np.ascontiguousarray(v.g1[i])).cuda()
    g2 = torch.from_numpy(np.ascontiguousarray(v.g2[(i * 5 + 1) % v.n])).cuda()
    row, outs = {}, {}
    for name, H in P.items():
        if name == "lane" and n > lane_max:
            continue
        out = torch.empty((n, H.length_in_bytes_GT), dtype=torch.uint8, device="cuda")
        ts = []
        for rep in range(5 if name == "lane" else 7):
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
    lane = "lane %9.3f ms  (%9.0f /s)    same bytes: %s" % (row["lane"], n / row["lane"] * 1e3, np.array_equal(outs["wave"], outs["lane"])) if "lane" in row else "lane      --"
    print("%s %s%s n = %6d   wavefronts %9.3f ms  (%9.0f /s)    %s" % (PNAME, mode, " k = %d" % k if k > 1 else "", n, row["wave"], n / row["wave"] * 1e3, lane), flush=True)
for pp in pps.values():
    pp.clear()
