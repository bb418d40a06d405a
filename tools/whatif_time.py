#!/usr/bin/env python3
"""Kernel time of a library VARIANT whose results may be wrong on purpose (what-if builds: "how fast would it be
without X").  No correctness gate, no JSON line: this is not a benchmark and its numbers never go into bench.py output.

  PBC_HIP_LIB=libpbc_hip_variant.so python tools/whatif_time.py <workload> [log2n] [steps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pbc_amd
import bench

w = sys.argv[1]
pname, fixture, k, dlog, desc = bench.WORKLOADS[w]
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else dlog
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
P = pbc_amd.Pairing(open(os.path.join(ROOT, "pbc_amd", "param", pname + ".param")).read())
g1, g2, _ = bench.load_vec(os.path.join(ROOT, "tests", "golden", fixture))
D, n = g1.shape[0], 1 << log2n
d1, d2 = torch.from_numpy(g1).cuda(), torch.from_numpy(g2).cuda()
t = torch.arange(0, n * k, device="cuda")
G1, G2 = d1[(t // D) % D].contiguous(), d2[t % D].contiguous()
TRACE = os.environ.get("PBC_WHATIF_TRACE") == "1"      # library built with -DPBC_F_WHATIF_TRACE: per-wave timestamps behind the results
GT = torch.empty(n * P.length_in_bytes_GT + (256 + 4096 * 32 if TRACE else 0), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream()


def step():
    if k == 1:
        P.element_pairing_dev(GT.data_ptr(), G1.data_ptr(), G2.data_ptr(), n, st.cuda_stream)
    else:
        P.element_prod_pairing_dev(GT.data_ptr(), G1.data_ptr(), G2.data_ptr(), n, k, st.cuda_stream)


step()
torch.cuda.synchronize()
ms = []
for _ in range(steps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); step(); b.record(st)
    torch.cuda.synchronize()
    ms.append(a.elapsed_time(b))
print("WHAT-IF %s %s 2^%d: %.3f ms per launch (min %.3f) = %.3f M units/s -- results unchecked" %
      (os.environ.get("PBC_HIP_LIB", "libpbc_hip.so"), w, log2n, sum(ms) / len(ms), min(ms), n / (sum(ms) / len(ms)) / 1e3))

if TRACE:
    off = (n * P.length_in_bytes_GT + 255) & ~255
    t = GT[off:off + 2048 * 32].cpu().numpy().view(np.uint64).reshape(-1, 4)
    t = t[t[:, 1] > 0]
    t0, t1 = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64)
    span = t1.max() - t0.min()
    print("waves %d  span %.0f ticks (100 MHz: %.2f ms)  start skew max %.2f ms  wave duration min/mean/max %.2f / %.2f / %.2f ms" %
          (len(t), span, span / 1e5, (t0.max() - t0.min()) / 1e5, (t1 - t0).min() / 1e5, (t1 - t0).mean() / 1e5, (t1 - t0).max() / 1e5))
    xcc = t[:, 3] & 0xf
    for x in sorted(set(xcc)):
        m = xcc == x
        print("  xcc %d: %4d waves, duration mean %.2f max %.2f ms, last end %.2f ms" % (x, m.sum(), (t1 - t0)[m].mean() / 1e5, (t1 - t0)[m].max() / 1e5, (t1[m].max() - t0.min()) / 1e5))
    cu = (t[:, 2] >> 8) & 0xf
    se = (t[:, 2] >> 13) & 0x7
    d = (t1 - t0) / 1e5
    key = xcc * 1000 + se * 16 + cu
    per = {k: d[key == k] for k in sorted(set(key))}
    cnt = np.array([len(v) for v in per.values()])
    print("  distinct (xcc, se, cu): %d, waves per CU min/max %d / %d; per-CU mean duration min %.2f max %.2f ms" %
          (len(per), cnt.min(), cnt.max(), min(v.mean() for v in per.values()), max(v.mean() for v in per.values())))
