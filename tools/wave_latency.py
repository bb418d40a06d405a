"""Latency of small batches, a.param: the one-pairing-per-wavefront kernel and its four-wavefront form (pairing_aw.cuh) against the throughput kernel,
device buffers, events around the call; median of 7 after 2 warm-ups.   python tools/wave_latency.py [sizes...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pbc_amd  # noqa: E402
from conftest import golden, _param  # noqa: E402

v = golden("a_chain1024.vec")
sizes = [int(x) for x in sys.argv[1:]] or [1, 16, 256, 1024, 2048, 3072, 4096, 6144, 8192, 16384]
P = {"wave4": pbc_amd.Pairing(_param("a") + "hip_wave_max 1000000\nhip_wave4_max 1000000\n"),
     "wave2": pbc_amd.Pairing(_param("a") + "hip_wave_max 1000000\nhip_wave4_max 0\nhip_wave2_max 1000000\n"),
     "wave": pbc_amd.Pairing(_param("a") + "hip_wave_max 1000000\nhip_wave4_max 0\nhip_wave2_max 0\n"), "lane": pbc_amd.Pairing(_param("a") + "hip_wave_max 0\n")}
for n in sizes:
    i = np.arange(n) % v.n
    g1 = torch.from_numpy(np.ascontiguousarray(v.g1[i])).cuda()
    g2 = torch.from_numpy(np.ascontiguousarray(v.g2[(i * 5 + 1) % v.n])).cuda()
    row = {}
    outs = {}
    for name, H in P.items():
        out = torch.empty((n, 128), dtype=torch.uint8, device="cuda")
        ts = []
        for rep in range(9):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            H.element_pairing_dev(out.data_ptr(), g1.data_ptr(), g2.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        row[name] = float(np.median(ts[2:]))
        outs[name] = out.cpu().numpy()
    same = all(np.array_equal(outs[k], outs["lane"]) for k in ("wave", "wave2", "wave4"))
    print("n = %6d   4 waves %8.3f ms  (%9.0f /s)   2 waves %8.3f ms  (%9.0f /s)   1 wave %8.3f ms  (%9.0f /s)    lane %8.3f ms  (%9.0f /s)    same bytes: %s" %
          (n, row["wave4"], n / row["wave4"] * 1e3, row["wave2"], n / row["wave2"] * 1e3, row["wave"], n / row["wave"] * 1e3, row["lane"], n / row["lane"] * 1e3, same), flush=True)

# round 5: pairing_pp_apply and k-term products on the wave routines against the lane kernels
def timed(call):
    ts = []
    for rep in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        call()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts[2:]))


st = torch.cuda.current_stream().cuda_stream
for n in [s for s in sizes if s <= 5120]:
    i = np.arange(n) % v.n
    g2 = torch.from_numpy(np.ascontiguousarray(v.g2[(i * 5 + 1) % v.n])).cuda()
    row, outs = {}, {}
    for name, H in P.items():
        pp = H.pp_init(v.g1[7])
        out = torch.empty((n, 128), dtype=torch.uint8, device="cuda")
        row[name] = timed(lambda: pp.apply_dev(out.data_ptr(), g2.data_ptr(), n, st))
        outs[name] = out.cpu().numpy()
        pp.clear()
    same = all(np.array_equal(outs[k], outs["lane"]) for k in ("wave", "wave2", "wave4"))
    print("pp_apply  n = %6d   4 waves %8.3f ms   2 waves %8.3f ms   1 wave %8.3f ms   lane %8.3f ms   same bytes: %s" % (n, row["wave4"], row["wave2"], row["wave"], row["lane"], same), flush=True)
for n, k in ((1, 2), (1, 5), (1, 8), (1, 16), (16, 5), (64, 16), (200, 5)):
    t = np.arange(n * k)
    g1 = torch.from_numpy(np.ascontiguousarray(v.g1[(t * 3 + 1) % v.n])).cuda()
    g2 = torch.from_numpy(np.ascontiguousarray(v.g2[(t * 5 + 2) % v.n])).cuda()
    row, outs = {}, {}
    for name, H in P.items():
        out = torch.empty((n, 128), dtype=torch.uint8, device="cuda")
        row[name] = timed(lambda: H.element_prod_pairing_dev(out.data_ptr(), g1.data_ptr(), g2.data_ptr(), n, k, st))
        outs[name] = out.cpu().numpy()
    same = np.array_equal(outs["wave"], outs["lane"]) and np.array_equal(outs["wave4"], outs["lane"])
    print("products  n = %4d k = %2d   4 waves %8.3f ms   1 wave %8.3f ms   lane %8.3f ms   same bytes: %s" % (n, k, row["wave4"], row["wave"], row["lane"], same), flush=True)
