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
GT = torch.empty(n, P.length_in_bytes_GT, dtype=torch.uint8, device="cuda")
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
