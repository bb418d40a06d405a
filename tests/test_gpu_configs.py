"""GPU tests (-m gpu) for launch configurations that earlier rounds built but never ran ("weak" 1):
bench.py --strong (BASELINE config 5's sharding), resident-workgroup launches with ragged tails on every kernel that
loops over the batch, and the host-buffer entry points with a device set + page-locked buffers + products + more chunks
than devices."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import golden, _param

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tile(a, n):
    return np.ascontiguousarray(np.tile(a, (-(-n // len(a)), 1))[:n])


# ---- (a) bench.py --strong: one job range-split over the ranks -------------------------------------------------------
@pytest.mark.parametrize("workload,log2n,k", [("a-prod16", 14, 16), ("a", 16, 1)])
def test_bench_strong_scaling_two_ranks_share_the_gpu(workload, log2n, k):
    """BASELINE config 5 is `bench.py --workload a-prod16 --strong --gpus 8`: ONE job of 2^log2n units cut into
    contiguous ranges, one per rank.  Two ranks on the one GPU of the test box (gloo for the barrier and the clock): the
    job size is the global batch, both ranks report a kernel time, and BOTH shards -- rank 1's starts at unit n/2 --
    pass the pre-timing gate against the reference fixture / the product of single pairings."""
    env = dict(os.environ, PBC_BENCH_SAME_DEVICE="1", PBC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", workload, "--strong", "--log2n", str(log2n), "--no-cpu-baseline", "--no-host-path"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    n = 1 << log2n
    assert j["n_gpus"] == 2 and j["scaling"] == "strong"
    assert j["config"]["global_batch"] == n and j["config"]["units_per_gpu"] == n // 2 and j["config"]["terms_per_unit"] == k
    assert len(j["per_rank_kernel_ms"]) == 2 and all(x > 0 for x in j["per_rank_kernel_ms"])
    gate = j["per_rank_gate"]
    assert [g["first_unit"] for g in gate] == [0, n // 2] and [g["units"] for g in gate] == [n // 2, n // 2]
    assert all(g["checked"] > 0 for g in gate), gate
    assert j["value"] > 0


def test_bench_strong_scaling_eight_ranks_ragged_job():
    """what the driver's 8-GPU node runs for BASELINE config 5 -- `bench.py --workload a-prod16 --strong --gpus 8` -- with
    EIGHT ranks (on the one GPU of the test box, gloo for the barrier and the clock) and a job of 2^11 + 5 products, which
    no rank count divides: the shares are 256 or 257 units, every rank's first unit follows from the floor split, every
    shard passes its gate, and the shares add up to the job."""
    env = dict(os.environ, PBC_BENCH_SAME_DEVICE="1", PBC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--workload", "a-prod16", "--strong", "--log2n", "11", "--extra-units", "5", "--no-cpu-baseline", "--no-host-path"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    n = (1 << 11) + 5
    assert j["n_gpus"] == 8 and j["scaling"] == "strong" and j["config"]["global_batch"] == n
    gate = j["per_rank_gate"]
    assert [g["first_unit"] for g in gate] == [r_ * n // 8 for r_ in range(8)]
    assert [g["units"] for g in gate] == [(r_ + 1) * n // 8 - r_ * n // 8 for r_ in range(8)]
    assert sum(g["units"] for g in gate) == n and sorted(set(g["units"] for g in gate)) == [256, 257]
    assert all(g["checked"] > 0 for g in gate), gate
    assert len(j["per_rank_kernel_ms"]) == 8 and all(x > 0 for x in j["per_rank_kernel_ms"])


# ---- (b) resident workgroups: several units per lane, ragged tails ---------------------------------------------------
def _check_tail(got, want, n):
    assert np.array_equal(got[:n], want), "a unit differs"
    assert (got[n:] == 0xA5).all(), "bytes past the last unit were written"


RESIDENCY = 1024 * 128       # workgroups an MI355X holds at once for the type a / f / 7-word type d kernels, times their lanes


@pytest.mark.parametrize("extra", [1, 77, 128, 300])
@pytest.mark.parametrize("case", ["a", "a-lane", "a-pp", "d201", "d190"])
def test_resident_grid_ragged_sizes_singles(hips, case, extra):
    """al_pairing_kernel, al_pp_apply_kernel and the 7-word d_prod_pairing_kernel walk the batch in strides of the chip's
    residency (pbc_hip.hip PBC_RESIDENT_LOOP): batches just above ONE residency with ragged tails -- every unit bit-exact
    against the reference's vectors (tiled), none left out, nothing written past the end."""
    import torch
    # "a": the tail of wave-kernel size runs on the one-pairing-per-wavefront kernel behind the resident launch;
    # "a-lane" ("hip_wave_max 0"): the resident kernel walks its own ragged tail
    key, name = {"a": ("a", "a_chain1024.vec"), "a-lane": ("a", "a_chain1024.vec"), "a-pp": ("a", "a_chain1024.vec"), "d201": ("d201", "d201_rand12.vec"),
                 "d190": ("d278027-190-181", "d278027-190-181_rand12.vec")}[case]
    v = golden(name)
    P = hips[key]
    if case == "a-lane":
        import pbc_amd
        P = pbc_amd.Pairing(_param("a") + "hip_wave_max 0\n")
    n = RESIDENCY + extra
    s = torch.cuda.current_stream().cuda_stream
    GT = torch.full((n + 64, v.lenT), 0xA5, dtype=torch.uint8, device="cuda")
    g2 = torch.from_numpy(_tile(v.g2, n)).cuda()
    if case == "a-pp":
        pp = P.pp_init(v.g1[3])
        pp.apply_dev(GT.data_ptr(), g2.data_ptr(), n, s)
        torch.cuda.synchronize()
        pp.clear()
        g1 = np.ascontiguousarray(np.tile(v.g1[3], (v.n, 1)))
        want = _tile(P.element_pairing(g1, v.g2), n)          # e(P_3, Q_j): 1024 single pairings
        assert np.array_equal(want[3], v.gt[3])
    else:
        g1 = torch.from_numpy(_tile(v.g1, n)).cuda()
        P.element_pairing_dev(GT.data_ptr(), g1.data_ptr(), g2.data_ptr(), n, s)
        torch.cuda.synchronize()
        want = _tile(v.gt, n)
    _check_tail(GT.cpu().numpy(), want, n)


@pytest.mark.parametrize("k,extra", [(3, 1), (3, 300), (16, 77), (16, 128)])
def test_resident_grid_ragged_sizes_type_a_products(hip_a, k, extra):
    """One term per lane (al_miller_kernel over n k terms) and one product per lane (al_prod_finish_kernel over n
    products): BOTH kernels loop, so n itself is taken just above one residency.  Expected bytes: the products of the
    reference's single-pairing values (the fixture), formed with element_mul on GT for the distinct units."""
    import torch
    v = golden("a_chain1024.vec")
    n = RESIDENCY + extra
    D = v.n
    t = np.arange(n * k)
    i1, i2 = (t * 5 + t // D) % D, (t * 11 + 3) % D
    g1 = torch.from_numpy(v.g1).cuda()[torch.from_numpy(i1).cuda()].contiguous()
    g2 = torch.from_numpy(v.g2).cuda()[torch.from_numpy(i2).cuda()].contiguous()
    GT = torch.full((n + 64, v.lenT), 0xA5, dtype=torch.uint8, device="cuda")
    hip_a.element_prod_pairing_dev(GT.data_ptr(), g1.data_ptr(), g2.data_ptr(), n, k, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = GT.cpu().numpy()
    assert (got[n:] == 0xA5).all()
    # check the first units, the units around the residency boundary and the ragged tail against single pairings
    sel = np.unique(np.concatenate([np.arange(0, 40), np.arange(RESIDENCY - 130, n)]))
    sel = sel[sel < n]
    terms = (sel[:, None] * k + np.arange(k)[None, :]).reshape(-1)
    singles = hip_a.element_pairing(v.g1[i1[terms]], v.g2[i2[terms]]).reshape(len(sel), k, -1)
    want = singles[:, 0]
    for j in range(1, k):
        want = hip_a.element_mul_GT(want, singles[:, j])
    assert np.array_equal(got[sel], want)
    # ... and a strided sample of the rest through a second, independent launch shape (small batches, no loop)
    rest = np.arange(41, RESIDENCY - 131, 4099)
    terms = (rest[:, None] * k + np.arange(k)[None, :]).reshape(-1)
    assert np.array_equal(got[rest], hip_a.element_prod_pairing(v.g1[i1[terms]], v.g2[i2[terms]], k))


SMALL_GRID = [("a", "a_chain1024.vec", 1), ("a", "a_chain1024.vec", 3), ("a-pp", "a_chain1024.vec", 1), ("f", "f_chain128.vec", 1),
              ("f", "f_chain128.vec", 2), ("f_256", "f_256_rand4.vec", 1), ("d201", "d201_rand12.vec", 1), ("d201", "d201_rand12.vec", 3),
              ("d224", "d224_rand12.vec", 1), ("d201-pp", "d201_rand12.vec", 1),
              ("d278027-190-181", "d278027-190-181_rand12.vec", 1), ("d278027-190-181", "d278027-190-181_rand12.vec", 3),
              ("d278027-190-181-pp", "d278027-190-181_rand12.vec", 1)]


@pytest.mark.parametrize("case,name,k", SMALL_GRID)
def test_every_resident_kernel_with_a_forced_small_grid(hips, case, name, k):
    """every kernel that wraps its body in the resident loop, launched with THREE workgroups
    ("hip_resident_slots 3" in the parameter text) over 1000 units: each lane runs its body three times, the last time
    with a ragged workgroup -- state carried from one iteration into the next (LDS slots, accumulators, workspace records)
    would show as wrong bytes.  Compared with the default launch of the same library."""
    import pbc_amd
    key = case.replace("-pp", "")
    v = golden(name)
    # ("hip_wave_max 0": the throughput kernel also at this size, not the one-pairing-per-wavefront kernel of small batches)
    P = pbc_amd.Pairing(_param({"a": "a", "f": "f"}.get(key, key)) + "hip_resident_slots 3\nhip_wave_max 0\n")
    ref = hips[key]
    n = 1000
    t = np.arange(n * k)
    M = v.n * v.k
    i1 = (t * 3 + 1) % M
    i2 = np.where(t % 4 == 0, i1, (t * 7) % M)              # every fourth unit is a pair of the fixture itself
    g1, g2 = np.ascontiguousarray(v.g1.reshape(-1, v.len1)[i1]), np.ascontiguousarray(v.g2.reshape(-1, v.len2)[i2])
    if case.endswith("-pp"):
        a, b = P.pp_init(g1[0]), ref.pp_init(g1[0])
        got, want = a.apply(g2), b.apply(g2)
        a.clear(); b.clear()
    elif k == 1:
        got, want = P.element_pairing(g1, g2), ref.element_pairing(g1, g2)
    else:
        got, want = P.element_prod_pairing(g1, g2, k), ref.element_prod_pairing(g1, g2, k)
    assert np.array_equal(got, want)
    if k == 1 and not case.endswith("-pp"):
        d = np.nonzero(i1 == i2)[0]
        assert len(d) >= n // 4 and np.array_equal(got[d], v.gt[i1[d]])
    P.clear()


# ---- (c) host buffers: device set + page-locked memory + products + more chunks than devices ---------------------------
@pytest.mark.parametrize("key,name,pname", [("a", "a_chain1024.vec", "a"), ("d", "d_chain256.vec", "d159")])
def test_device_set_with_pinned_buffers_products_and_many_chunks_one_physical_device(hips, key, name, pname):
    """pbc_hip_pairing_use_devices + pbc_hip_host_alloc (hipHostMallocPortable) + k = 3 + "hip_host_chunk 37": 14 chunks
    dealt to a device set of three positions, worked on IN PLACE by three host threads with their own streams and product
    workspaces.  The test box has ONE GPU, so the set lists device 0 three times -- the threading, the chunk arithmetic
    and the per-position contexts are the multi-device ones; two DISTINCT devices sharing portable pinned memory can only
    run on the driver's multi-GPU node.  Also: the staged route of the same call (pageable memory), an output buffer that
    overlaps an input (must be staged, not corrupted), and a buffer that is page-locked only in its first half."""
    import pbc_amd
    L = pbc_amd.lib()
    v = golden(name)
    k, n = 3, 37 * 13 + 5
    t = np.arange(n * k)
    g1, g2 = np.ascontiguousarray(v.g1[(t * 5 + 2) % v.n]), np.ascontiguousarray(v.g2[(t * 3) % v.n])
    want = hips[key].element_prod_pairing(g1, g2, k)
    H = pbc_amd.Pairing(_param(pname) + "hip_host_chunk 37\n")
    H.use_devices([0, 0, 0])
    assert np.array_equal(H.element_prod_pairing(g1, g2, k), want)             # pageable numpy memory: staged, 3 workers
    bufs = []

    def pinned(a):
        p = ctypes.c_void_p()
        assert L.pbc_hip_host_alloc(ctypes.byref(p), a.size) == 0
        bufs.append(p)
        arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(a.size,))
        arr[:] = a.reshape(-1)
        return p, arr
    p1, a1 = pinned(g1)
    p2, a2 = pinned(g2)
    pt, at = pinned(np.zeros(n * v.lenT, np.uint8))
    for rep in range(3):                                                         # steady state: nothing is reallocated
        at[:] = 0
        assert L.pbc_hip_element_prod_pairing_batch(H._h, pt, p1, p2, n, k) == 0, pbc_amd._err()
        assert np.array_equal(at.reshape(n, -1), want)
    # singles through the same object (k = 1 takes the pairing kernels, no workspace)
    m = n * k
    pt1, at1 = pinned(np.zeros(m * v.lenT, np.uint8))
    assert L.pbc_hip_element_pairing_batch(H._h, pt1, p1, p2, m) == 0
    assert np.array_equal(at1.reshape(m, -1), hips[key].element_pairing(g1, g2))
    # output overlapping the first input: results must still be those of the untouched inputs
    assert v.lenT <= k * v.len1
    assert L.pbc_hip_element_prod_pairing_batch(H._h, p1, p1, p2, n, k) == 0
    assert np.array_equal(a1[:n * v.lenT].reshape(n, -1), want)
    a1[:] = g1.reshape(-1)
    # a range that is page-locked only at its start (hipHostRegister over the first half of a pageable array): the library
    # must not take it for page-locked memory (its last byte is not); it goes the staged route, where the HIP runtime's own
    # copy may refuse a half-registered range -- either the right bytes or a clean error, never a fault in a kernel
    import torch
    big = np.zeros(g2.size + 8192, np.uint8)
    off = (-big.ctypes.data) % 4096
    half = (g2.size // 2) // 4096 * 4096
    rt = torch.cuda.cudart()
    if half >= 4096 and hasattr(rt, "cudaHostRegister"):
        view = big[off:off + g2.size]
        view[:] = g2.reshape(-1)
        assert int(rt.cudaHostRegister(view.ctypes.data, half, 0)) == 0
        try:
            at[:] = 0
            rc = L.pbc_hip_element_prod_pairing_batch(H._h, pt, p1, ctypes.c_void_p(view.ctypes.data), n, k)
            assert rc in (0, 1)
            if rc == 0:
                assert np.array_equal(at.reshape(n, -1), want)
            else:
                assert "copy failed" in pbc_amd._err()
        finally:
            rt.cudaHostUnregister(view.ctypes.data)
        assert L.pbc_hip_element_prod_pairing_batch(H._h, pt, p1, p2, n, k) == 0      # the object is intact afterwards
        assert np.array_equal(at.reshape(n, -1), want)
    # the same object with EIGHT positions in its set (the driver's node has eight devices; here device 0 eight times):
    # eight workers, 14 chunks, in place on the pinned buffers and staged from pageable memory
    H.use_devices([0] * 8)
    at[:] = 0
    assert L.pbc_hip_element_prod_pairing_batch(H._h, pt, p1, p2, n, k) == 0, pbc_amd._err()
    assert np.array_equal(at.reshape(n, -1), want)
    assert np.array_equal(H.element_prod_pairing(g1, g2, k), want)
    assert np.array_equal(H.element_pairing(g1, g2), hips[key].element_pairing(g1, g2))
    H.use_devices([0, 0, 0])
    H.release_workspaces()                                                       # frees chunk buffers and workspaces; the next call rebuilds them
    assert np.array_equal(H.element_prod_pairing(g1, g2, k), want)
    for p in bufs:
        L.pbc_hip_host_free(p)
    H.clear()
