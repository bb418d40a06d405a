"""N>1 path on CPU: world-size-2 gloo processes run the range split + host-side gather of
pbc_amd/multi.py (the code a multi-GPU caller uses) with the CPU oracle standing in for the
per-rank GPU worker -- the sharding logic is identical, only the worker differs."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, k, n_units, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle
    from pbc_amd import multi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O = oracle.OraclePairing(open(os.path.join(ROOT, "pbc_amd", "param", "a.param")).read())
    v = oracle.Vec(os.path.join(ROOT, "tests", "golden", "a_chain1024.vec"))
    g1, g2 = v.g1[:n_units * k], v.g2[:n_units * k]
    compute = (lambda a, b: O.pairing_batch(a, b)) if k == 1 else (lambda a, b: O.prod_pairing_batch(a, b, k))
    out = multi.element_pairing_sharded(compute, g1, g2, k)
    t = multi.max_over_ranks(1.0 + rank)
    if rank == 0:
        q.put((out, t))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k,n_units", [(1, 9), (3, 5), (1, 1)])
def test_range_split_gather_world2(k, n_units):
    import oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + k
    procs = [ctx.Process(target=_worker, args=(r, 2, port, k, n_units, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    O = oracle.OraclePairing(open(os.path.join(ROOT, "pbc_amd", "param", "a.param")).read())
    v = oracle.Vec(os.path.join(ROOT, "tests", "golden", "a_chain1024.vec"))
    g1, g2 = v.g1[:n_units * k], v.g2[:n_units * k]
    want = O.pairing_batch(g1, g2) if k == 1 else O.prod_pairing_batch(g1, g2, k)
    assert np.array_equal(out, want)
    assert t == 2.0                       # max over ranks of (1.0, 2.0)


def test_range_split_properties():
    from pbc_amd.multi import range_split
    for n in (0, 1, 7, 8, 9, 1 << 20):
        for w in (1, 2, 3, 4, 8):
            rs = range_split(n, w)
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert max(hi - lo for lo, hi in rs) - min(hi - lo for lo, hi in rs) <= -(-n // w)
