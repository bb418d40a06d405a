import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, pbc_amd
from conftest import golden, _param
v = golden("a_chain1024.vec")
for extra in ("", "hip_wave_max 0\n"):
    P = pbc_amd.Pairing(_param("a") + extra)
    for n in (131072, 131073, 131072 + 1024, 131072 + 5120, 131072 + 5121, 262144 + 300):
        i = np.arange(n) % v.n
        g1 = torch.from_numpy(np.ascontiguousarray(v.g1[i])).cuda(); g2 = torch.from_numpy(np.ascontiguousarray(v.g2[i])).cuda()
        out = torch.empty((n, 128), dtype=torch.uint8, device="cuda")
        ts = []
        for rep in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); P.element_pairing_dev(out.data_ptr(), g1.data_ptr(), g2.data_ptr(), n, torch.cuda.current_stream().cuda_stream); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        ok = np.array_equal(out.cpu().numpy()[-2000:], v.gt[i[-2000:]])
        print(repr(extra), n, "%.2f ms" % np.median(ts[1:]), ok, flush=True)
