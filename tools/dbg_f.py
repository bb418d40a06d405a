import sys, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, oracle
txt = open("pbc_amd/param/f.param").read()
v = oracle.Vec("tests/golden/f_rand16.vec")
n = 4
A = v.gt[:n].copy(); B = v.gt[n:2*n].copy()
if len(sys.argv) > 1 and sys.argv[1] == "gpu":
    import pbc_amd
    P = pbc_amd.Pairing(txt)
    res = {}
    for op in (10, 11, 12, 13, 14, 15, 16):
        out = np.zeros((n, 240), np.uint8)
        rc = pbc_amd.lib().pbc_hip_diag_stage(P._h, op, out.ctypes.data, out.size, A.ctypes.data, B.ctypes.data, n)
        res[str(op)] = out
    gm = np.zeros((n, 240), np.uint8)
    pbc_amd.lib().pbc_hip_diag_stage(P._h, 1, gm.ctypes.data, gm.size, v.g1[:n].ctypes.data, v.g2[:n].ctypes.data, n)
    res["1"] = gm
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez("gpurun_out/dbg_f.npz", **res)
    print("saved")
else:
    from hostsim import HostSim
    H = HostSim(txt)
    g = np.load("gpurun_out/dbg_f.npz")
    for op in (10, 11, 12, 13, 14, 15, 16):
        hm, _ = H.stage(op, A, B, n)
        print(op, "equal" if np.array_equal(hm[:n*240].reshape(n,240), g[str(op)]) else "DIFF")
    hm, _ = H.stage(1, v.g1[:n], v.g2[:n], n)
    print(1, "equal" if np.array_equal(hm[:n*240].reshape(n,240), g["1"]) else "DIFF")
