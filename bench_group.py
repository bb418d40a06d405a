"""bench_group.py -- the group operations next to the pairing as bench.py workloads (SURVEY 8f row 2 measured the way 8d
measures the pairings): `python bench.py --workload a-g1-mul` etc.  Same contract as bench.py (one JSON line from rank 0,
inputs resident in HBM, events around the launches, range split over ranks with no data-path collective); bench.py
dispatches here for the names in GROUP_WORKLOADS.

A step:
  *-g1-mul / *-g2-mul   one element_mul_zn launch: n points x n random scalars below r
  *-gt-pow              one element_pow_zn launch on n pairing values
  *-hash-g1             one element_from_hash launch on n 32-byte digests
  *-g1-pp / *-gt-pp     one element_pp_pow_zn launch: n scalars against ONE preprocessed base (element_pp_init outside the clock)
  a-bls-verify          the flow of example/bls.c:64-117 for n signatures as a batch, device-resident throughout: hash the n
                        digests, scale hash and signature of message j by a random r_j (resp. -r_j), and check the products
                        prod_j e(r_j H_j, pk) e(-r_j sigma_j, g) == 1 over groups of eight signatures (16-term products).
Before timing, results are compared with the CPU oracle on a sample (and, for bls-verify, with the reference's own run of
the flow, tests/golden/a_bls64.bin) -- a wrong kernel cannot post a number.
"""
import json
import os
import struct
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
import pbc_amd  # noqa: E402

# name: (param, fixture, op, default log2 units)
GROUP_WORKLOADS = {}
for _t, _p, _fx in (("a", "a", "a_chain1024.vec"), ("d", "d159", "d_chain256.vec"), ("f", "f", "f_chain128.vec")):
    GROUP_WORKLOADS[_t + "-g1-mul"] = (_p, _fx, "g1mul", 20 if _t == "a" else 18)
    if _t != "a":                      # type a is symmetric: G2 is the same curve and the same kernel as G1
        GROUP_WORKLOADS[_t + "-g2-mul"] = (_p, _fx, "g2mul", 17)
    GROUP_WORKLOADS[_t + "-gt-pow"] = (_p, _fx, "gtpow", 20 if _t == "a" else 17)
    GROUP_WORKLOADS[_t + "-hash-g1"] = (_p, _fx, "hashg1", 18 if _t == "a" else 22)     # (5-word fields: 2^18 hashes are a 1 ms launch)
    GROUP_WORKLOADS[_t + "-g1-pp"] = (_p, _fx, "g1pp", 20)
    GROUP_WORKLOADS[_t + "-gt-pp"] = (_p, _fx, "gtpp", 20 if _t != "f" else 18)
    GROUP_WORKLOADS[_t + "-decompress"] = (_p, _fx, "decompress", 18)
    GROUP_WORKLOADS[_t + "-compress"] = (_p, _fx, "compress", 20)
    # round 5: the group law, Z_r inversion, element_pow2_zn on G1 and GT
    GROUP_WORKLOADS[_t + "-g1-add"] = (_p, _fx, "g1add", 20)
    GROUP_WORKLOADS[_t + "-zr-inv"] = (_p, _fx, "zrinv", 20)
    GROUP_WORKLOADS[_t + "-g1-pow2"] = (_p, _fx, "g1pow2", 18 if _t == "a" else 17)
    GROUP_WORKLOADS[_t + "-gt-pow2"] = (_p, _fx, "gtpow2", 18 if _t == "a" else 16)
GROUP_WORKLOADS["a-bls-verify"] = ("a", "a_chain1024.vec", "blsverify", 17)

DESC = {"g1mul": "element_mul_zn on G1", "g2mul": "element_mul_zn on G2 (the twist)", "gtpow": "element_pow_zn on GT",
        "hashg1": "element_from_hash on G1 (32-byte digests)", "g1pp": "element_pp_pow_zn on G1 (fixed base)",
        "gtpp": "element_pp_pow_zn on GT (fixed base)", "decompress": "element_from_bytes_compressed on G1",
        "compress": "element_to_bytes_compressed on G1", "g1add": "element_add on G1", "zrinv": "element_invert on Zr",
        "g1pow2": "element_pow2_zn on G1", "gtpow2": "element_pow2_zn on GT", "blsverify": "BLS batch verification (hash, 2 x mul_zn, 16-term products)"}
UNIT = {"g1mul": "scalar multiplications/s", "g2mul": "scalar multiplications/s", "gtpow": "powers/s", "hashg1": "hashes/s",
        "g1pp": "scalar multiplications/s", "gtpp": "powers/s", "decompress": "points/s", "compress": "points/s", "blsverify": "signatures/s",
        "g1add": "additions/s", "zrinv": "inversions/s", "g1pow2": "double scalar multiplications/s", "gtpow2": "double powers/s"}


def param_int(text, key):
    for line in text.splitlines():
        f = line.split()
        if len(f) == 2 and f[0] == key:
            return int(f[1])
    raise KeyError(key)


def reference_model(text, op, nlimb):
    """F_q products of the REFERENCE's algorithm per unit (the work model SURVEY 8d prescribes: products x (2 N^2 + N)
    word multiply-adds, inversions excluded) -- and, separately, the inversions it performs, which is where the reference
    spends most of its time on the affine group law (one mpz_invert per group operation, ecc/curve.c:102-207).
    generic_pow_mpz (arith/field.c:14-126): b squarings + b / (k + 1) + 2^k - 2 multiplications for a b-bit exponent,
    k = 4 for 158..474 bits, 5 up to 1324."""
    r = param_int(text, "r")
    q = param_int(text, "q")
    typ = [ln.split()[1] for ln in text.splitlines() if ln.startswith("type ")][0]
    b = r.bit_length()

    def win(bits):
        return 8 if bits > 9065 else 7 if bits > 3529 else 6 if bits > 1324 else 5 if bits > 474 else 4 if bits > 157 else 3 if bits > 47 else 2

    def pow_ops(bits):                 # (squarings, multiplications) of generic_pow_mpz
        k = win(bits)
        return bits, bits / (k + 1) + 2 ** k - 2
    ext = {"a": (1, 1, 2), "d": (1, 3, 6), "f": (1, 2, 12)}[typ]      # degrees of G1's, G2's and GT's fields over F_q
    # F_q products of one product / square in an extension of degree d as the reference forms them
    fmul = {1: 1, 2: 3, 3: 6, 6: 18, 12: 54}                           # fi/fq_mul Karatsuba 3; polymod degree 3: 6; F_q^6 = 3 x 6; F_q^12: kar_poly_2 on 6 x F_q^2 = 18 x 3
    sq, mu = pow_ops(b)
    if op in ("g1mul", "g2mul"):
        d = ext[0] if op == "g1mul" else ext[1]
        dbl, add = 4 * fmul[d], 3 * fmul[d]                            # double_no_check: x^2, lambda / 2y, lambda^2, (x - x1) lambda; curve_mul: 3
        return sq * dbl + mu * add, sq + mu
    if op == "gtpow":
        d = ext[2]
        return (sq + mu) * fmul[d], 0
    if op == "hashg1":
        # curve_from_hash (ecc/curve.c:455-482): two tries on average, each x^3 + a x + b (2 products) and a square root
        # by one power of about bits(q) bits, then the cofactor multiplication (element_mul_mpz by h = #E / r)
        qs, qm = pow_ops(q.bit_length())
        try:
            h = param_int(text, "h")
        except KeyError:
            h = 0
        prods, invs = 2 * (2 + qs + qm), 0
        if h > 1:
            hs, hm = pow_ops(h.bit_length())
            prods += hs * 4 + hm * 3
            invs += hs + hm
        return prods, invs
    if op == "decompress":              # curve_from_x (ecc/curve.c:770-815): x^3 + a x + b (2 products) and one square root (a power of about bits(q) bits)
        qs, qm = pow_ops(q.bit_length())
        return 2 + qs + qm, 0
    if op == "compress":                # the parity of the canonical y: no arithmetic
        return 0, 0
    if op == "g1add":                   # curve_mul (ecc/curve.c:153-207): lambda = dy / dx, lambda^2, (x1 - x3) lambda: 3 products, 1 inversion
        return 3, 1
    if op == "zrinv":                   # mpz_invert on r-sized integers: no F_q product at all
        return 0, 1
    if op in ("g1pow2", "gtpow2"):      # element_pow2_zn (arith/field.c:153-196): b squarings, 3 b / 4 products on average + one for the table
        d = ext[0] if op == "g1pow2" else ext[2]
        if op == "g1pow2":
            return b * 4 * fmul[d] + (0.75 * b + 1) * 3 * fmul[d], b + 0.75 * b + 1
        return (b + 0.75 * b + 1) * fmul[d], 0
    rows = b // 5 + 1                  # element_pow_base_table (arith/field.c:286-323): one multiplication per 5-bit row
    if op == "g1pp":
        return rows * 3, rows
    if op == "gtpp":
        return rows * fmul[ext[2]], 0
    raise KeyError(op)


def executed_group_macs(workload):
    path = os.path.join(ROOT, "profiles", "executed_macs.json")
    if os.path.exists(path):
        e = json.load(open(path)).get(workload)
        if e:
            return float(e["executed_macs_per_unit"])
    return None


def cpu_baseline(param_path, op):
    import oracle
    tool = oracle.REF_TOOL
    cores = oracle.usable_cores()      # the cgroup quota, not the logical CPU count of the box
    if not os.path.exists(tool):
        return None
    per = {"g1mul": 400, "g2mul": 200, "gtpow": 2000, "hashg1": 200, "g1pp": 4000, "gtpp": 20000, "blsverify": 150,
           "compress": 400000, "decompress": 4000, "g1add": 100000, "zrinv": 400000, "g1pow2": 300, "gtpow2": 1500}[op]
    if not param_path.endswith("a.param"):
        per *= 4 if op in ("g1mul", "hashg1", "g1pp", "decompress", "g1pow2") else 1
    per *= int(os.environ.get("PBC_CPU_SAMPLE_SCALE", "1"))

    def run(n, workers):
        out = subprocess.run([tool, "benchg", param_path, op, str(n), str(workers)], capture_output=True, text=True, timeout=300)
        return json.loads(out.stdout.strip().splitlines()[-1])
    one = run(per, 1)
    allc = run(per, cores)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(p)
    except Exception:  # noqa: BLE001
        pass
    return {"value": round(allc["units_per_s"], 1), "unit": UNIT[op], "cores": cores, "kind": "reference",
            "sample": "%d %s per worker x %d forked workers (ref_tool benchg, %s), %.1f s wall" % (per, op, cores, os.path.basename(param_path), allc["wall_s"]),
            "single_core": round(one["units_per_s"], 1), "per_core_when_all_busy": round(allc["per_core"], 1), "logical_cpus": os.cpu_count(), "cgroup_cpu_quota_cores": quota}


def load_bls(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"PBCBLS01"
    t, n, hlen, l1, l2, lz = struct.unpack("<6I", raw[8:32])
    a = np.frombuffer(raw, np.uint8)
    off = 32
    out = {}
    for name, cnt, ln in (("digests", n, hlen), ("h", n, l1), ("sig", n, l1), ("g", 1, l2), ("pk", 1, l2), ("sk", 1, lz)):
        out[name] = a[off:off + cnt * ln].reshape(cnt, ln).copy()
        off += cnt * ln
    return out


def main(args, load_vec, ensure_built, MAC_PEAK):
    import bench                                     # pmc_traffic / evidence_commit (bench.py calls this with itself as __main__)
    pname, fixture, op, dlog = GROUP_WORKLOADS[args.workload]
    if args.log2n is None:
        args.log2n = dlog
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("PBC_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("PBC_BENCH_SAME_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    ensure_built(dist, local_rank)
    import oracle  # the checker of the pre-timing gate and the CPU baseline; never on the measured path

    param_path = os.path.join(ROOT, "pbc_amd", "param", pname + ".param")
    text = open(param_path).read()
    extra = ("\n" + args.param_extra.replace("=", " ").replace(",", "\n") + "\n") if args.param_extra else ""
    P = pbc_amd.Pairing(text + extra)
    O = oracle.OraclePairing(text)
    L1, L2, LT, LZ = P.length_in_bytes_G1, P.length_in_bytes_G2, P.length_in_bytes_GT, P.length_in_bytes_Zr
    n_job = (1 << args.log2n) + getattr(args, "extra_units", 0)
    if args.strong:
        first = rank * n_job // world
        n = (rank + 1) * n_job // world - first
    else:
        first, n = 0, n_job
    g1, g2, gt = load_vec(os.path.join(ROOT, "tests", "golden", fixture))
    D = g1.shape[0]
    rng = np.random.default_rng(1234 + rank)
    r = param_int(text, "r")
    # scalars below r: random bytes with the top byte cut below r's
    Z = rng.integers(0, 256, (n, LZ), dtype=np.uint8)
    top = r >> (8 * (LZ - 1))
    Z[:, 0] &= (1 << (top.bit_length() - 1)) - 1
    dZ = torch.from_numpy(Z).cuda()
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream
    idx = (torch.arange(first, first + n, device="cuda") * 7 + rank) % D
    pp = None
    sample = np.unique(np.concatenate([np.arange(min(n, 48)), np.arange(max(0, n - 48), n)]))
    bls = None

    if op in ("g1mul", "g2mul"):
        group = 1 if op == "g1mul" else 2
        src = g1 if group == 1 else g2
        IN = torch.from_numpy(src).cuda()[idx].contiguous()
        OUT = torch.empty_like(IN)
        unit_bytes = 2 * IN.shape[1] + LZ

        def step():
            P.element_mul_zn_dev(group, OUT.data_ptr(), IN.data_ptr(), dZ.data_ptr(), n, s)

        def gate():
            if group == 1 or pname == "a":
                want = O.g_mul(group, IN[sample].cpu().numpy(), Z[sample])
            else:                      # the twists: the complete ladder of the same library on another route (hip_group_slow), itself pinned by the reference's vectors in the tests
                Q = pbc_amd.Pairing(text + "hip_group_slow 1\n")
                want = Q.element_mul_zn(2, IN[sample].cpu().numpy(), Z[sample])
                Q.clear()
            return np.array_equal(OUT[sample].cpu().numpy(), want)
    elif op == "gtpow":
        IN = torch.from_numpy(gt).cuda()[idx].contiguous()
        OUT = torch.empty_like(IN)
        unit_bytes = 2 * LT + LZ

        def step():
            P.element_pow_zn_GT_dev(OUT.data_ptr(), IN.data_ptr(), dZ.data_ptr(), n, s)

        def gate():
            return np.array_equal(OUT[sample].cpu().numpy(), O.gt_pow(IN[sample].cpu().numpy(), Z[sample]))
    elif op == "hashg1":
        hlen = 32
        Hn = rng.integers(0, 256, (n, hlen), dtype=np.uint8)
        IN = torch.from_numpy(Hn).cuda()
        OUT = torch.empty(n, L1, dtype=torch.uint8, device="cuda")
        unit_bytes = hlen + L1

        def step():
            P.element_from_hash_dev(1, OUT.data_ptr(), IN.data_ptr(), hlen, n, s)

        def gate():
            return np.array_equal(OUT[sample].cpu().numpy(), O.from_hash(Hn[sample]))
    elif op in ("compress", "decompress"):
        pts = torch.from_numpy(g1).cuda()[idx].contiguous()
        comp = torch.from_numpy(O.point_format(0, g1)).cuda()[idx].contiguous()
        IN, OUT = (pts, torch.empty_like(comp)) if op == "compress" else (comp, torch.empty_like(pts))
        unit_bytes = IN.shape[1] + OUT.shape[1]
        what = "to_bytes_compressed" if op == "compress" else "from_bytes_compressed"

        def step():
            P.point_format_dev(what, 1, OUT.data_ptr(), IN.data_ptr(), n, s)

        def gate():
            return np.array_equal(OUT[sample].cpu().numpy(), O.point_format(0 if op == "compress" else 1, IN[sample].cpu().numpy()))
    elif op == "g1add":
        IN = torch.from_numpy(g1).cuda()[idx].contiguous()
        IN2 = torch.from_numpy(g1).cuda()[(idx * 3 + 1) % D].contiguous()
        OUT = torch.empty_like(IN)
        unit_bytes = 3 * L1

        def step():
            P.element_group_op_dev("add", 1, OUT.data_ptr(), IN.data_ptr(), IN2.data_ptr(), n, s)

        def gate():
            return np.array_equal(OUT[sample].cpu().numpy(), O.g1_op(0, IN[sample].cpu().numpy(), IN2[sample].cpu().numpy()))
    elif op == "zrinv":
        Z[:, -1] |= 1                    # never zero
        dZ = torch.from_numpy(Z).cuda()
        OUT = torch.empty_like(dZ)
        unit_bytes = 2 * LZ

        def step():
            P.zr_op_dev("invert", OUT.data_ptr(), dZ.data_ptr(), 0, n, s)

        def gate():
            return np.array_equal(OUT[sample].cpu().numpy(), O.zr_op(3, Z[sample]))
    elif op in ("g1pow2", "gtpow2"):
        group = 1 if op == "g1pow2" else 3
        src = g1 if group == 1 else gt
        IN = torch.from_numpy(src).cuda()[idx].contiguous()
        IN2 = torch.from_numpy(src).cuda()[(idx * 3 + 1) % D].contiguous()
        Z2 = rng.integers(0, 256, (n, LZ), dtype=np.uint8)
        Z2[:, 0] &= (1 << (top.bit_length() - 1)) - 1
        dZ2 = torch.from_numpy(Z2).cuda()
        OUT = torch.empty_like(IN)
        unit_bytes = 3 * IN.shape[1] + 2 * LZ

        def step():
            P.element_pow_multi_dev(group, OUT.data_ptr(), [IN.data_ptr(), IN2.data_ptr()], [dZ.data_ptr(), dZ2.data_ptr()], n, s)

        def gate():
            want = O.pow_multi(group, [IN[sample].cpu().numpy(), IN2[sample].cpu().numpy()], [Z[sample], Z2[sample]])
            return np.array_equal(OUT[sample].cpu().numpy(), want)
    elif op in ("g1pp", "gtpp"):
        group = 1 if op == "g1pp" else 3
        base = g1[5] if group == 1 else gt[5]
        pp = P.element_pp_init(group, base)
        OUT = torch.empty(n, len(base), dtype=torch.uint8, device="cuda")
        unit_bytes = LZ + len(base)

        def step():
            pp.pow_zn_dev(OUT.data_ptr(), dZ.data_ptr(), n, s)

        def gate():
            B = np.tile(base, (len(sample), 1))
            want = O.g_mul(1, B, Z[sample]) if group == 1 else O.gt_pow(B, Z[sample])
            return np.array_equal(OUT[sample].cpu().numpy(), want)
    else:                               # blsverify
        bls = load_bls(os.path.join(ROOT, "tests", "golden", "a_bls64.bin"))
        hlen, m = 32, 8                                                     # m signatures per product: 2 m = 16 terms
        n = n // m * m
        Hn = rng.integers(0, 256, (n, hlen), dtype=np.uint8)
        nf = len(bls["digests"])
        Hn[:nf] = bls["digests"]                                            # the first messages are the reference's own
        dDig = torch.from_numpy(Hn).cuda()
        dSK = torch.from_numpy(np.tile(bls["sk"], (n, 1))).cuda()
        H = torch.empty(n, L1, dtype=torch.uint8, device="cuda")
        SIG = torch.empty(n, L1, dtype=torch.uint8, device="cuda")
        # setup (not timed): the signer's side -- sigma_j = sk H(m_j)
        P.element_from_hash_dev(1, H.data_ptr(), dDig.data_ptr(), hlen, n, s)
        P.element_mul_zn_dev(1, SIG.data_ptr(), H.data_ptr(), dSK.data_ptr(), n, s)
        torch.cuda.synchronize()
        if not (np.array_equal(H[:nf].cpu().numpy(), bls["h"]) and np.array_equal(SIG[:nf].cpu().numpy(), bls["sig"])):
            sys.exit("bench_group.py: hashes / signatures differ from the reference's run of example/bls.c's flow -- refusing to time")
        # verifier's randomisers r_j (64-bit) and their negatives mod r, as Z_r records
        rj = [int.from_bytes(rng.bytes(8), "big") | 1 for _ in range(n)]
        Zp = np.stack([np.frombuffer(int(x).to_bytes(LZ, "big"), np.uint8) for x in rj])
        Zn = np.stack([np.frombuffer(int(r - x).to_bytes(LZ, "big"), np.uint8) for x in rj])
        dZp, dZn = torch.from_numpy(Zp).cuda(), torch.from_numpy(Zn).cuda()
        PK = torch.from_numpy(bls["pk"]).cuda()
        G = torch.from_numpy(bls["g"]).cuda()
        T1 = torch.empty(n, 2, L1, dtype=torch.uint8, device="cuda")       # per signature: (r_j H_j, -r_j sigma_j)
        T2 = torch.stack([PK[0], G[0]]).unsqueeze(0).expand(n, 2, L2).contiguous()   # ... paired with (pk, g)
        Hv = torch.empty(n, L1, dtype=torch.uint8, device="cuda")
        A = torch.empty(n, L1, dtype=torch.uint8, device="cuda")
        B = torch.empty(n, L1, dtype=torch.uint8, device="cuda")
        OUT = torch.empty(n // m, LT, dtype=torch.uint8, device="cuda")
        one = np.zeros(LT, np.uint8)
        one[LT // 2 - 1] = 1
        dOne = torch.from_numpy(one).cuda()
        unit_bytes = hlen + L1 + 2 * LZ + LT // m
        verdict = {}

        def step(sig=SIG):
            P.element_from_hash_dev(1, Hv.data_ptr(), dDig.data_ptr(), hlen, n, s)        # the verifier hashes the messages itself
            P.element_mul_zn_dev(1, A.data_ptr(), Hv.data_ptr(), dZp.data_ptr(), n, s)
            P.element_mul_zn_dev(1, B.data_ptr(), sig.data_ptr(), dZn.data_ptr(), n, s)
            T1[:, 0] = A
            T1[:, 1] = B
            P.element_prod_pairing_dev(OUT.data_ptr(), T1.data_ptr(), T2.data_ptr(), n // m, 2 * m, s)
            verdict["ok"] = (OUT == dOne).all(dim=1)

        def gate():
            ok = bool(verdict["ok"].all().item())
            # every single signature of the reference's run also verifies one by one: e(sigma, g) == e(h, pk)
            e1 = P.element_pairing(bls["sig"], np.tile(bls["g"], (nf, 1)))
            e2 = P.element_pairing(bls["h"], np.tile(bls["pk"], (nf, 1)))
            ok = ok and np.array_equal(e1, e2)
            # a forged signature must sink its batch -- and only its batch
            forged = SIG.clone()
            forged[m + 3] = SIG[m + 4]
            step(forged)
            torch.cuda.synchronize()
            v = verdict["ok"].cpu().numpy()
            ok = ok and (not v[1]) and v[0] and v[2:].all()
            step()
            torch.cuda.synchronize()
            return ok

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    if not gate():
        sys.exit("bench_group.py: %s results of rank %d differ from the oracle -- refusing to time" % (args.workload, rank))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    # the warm-up steps directly before the timed region, topped up to 0.3 s of launches (the shader clock falls back during
    # the host-side gate and needs tens of milliseconds of load to climb again: bench.py, profiles/r06_notes.md)
    spin_t0, spun = time.perf_counter(), 0
    while spun < max(1, args.warmup) or (time.perf_counter() - spin_t0 < 0.3 and spun < 256):
        step()
        spun += 1
        if spun >= max(1, args.warmup):
            torch.cuda.synchronize()
    sync_all()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    sync_all()
    dt = time.perf_counter() - t0
    my_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
    per_rank_ms = [my_ms]
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        per_rank_ms = [None] * world
        dist.all_gather_object(per_rank_ms, my_ms)
    if rank == 0:
        total = (n_job if args.strong else n * world) * args.steps
        value = total / dt
        sec = my_ms * 1e-3
        nl = {"a": 16, "d159": 5, "f": 5}[pname]
        per_mul = 2 * nl * nl + nl
        exe = executed_group_macs(args.workload)
        if op == "blsverify":
            alg_prod, alg_inv = None, None
            alg_rate = None
        else:
            alg_prod, alg_inv = reference_model(text, op, nl)
            alg_rate = n * alg_prod * per_mul / sec
        exe_rate = n * exe / sec if exe else None
        peak_measured = max(pbc_amd.int_mac_peak(13, 4000)[0], pbc_amd.int_mac_peak(14, 4000)[0])
        rate = exe_rate if exe_rate is not None else alg_rate
        out = {
            "metric": "%s (%s.param) per second, 2^%d batch" % (DESC[op], pname, args.log2n),
            "value": round(value, 1), "unit": UNIT[op], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "u32 (multi-word Montgomery F_q, bit-exact integer)",
            "data": "synthetic: points / pairing values of tests/golden/%s, random scalars below r, random 32-byte digests; resident in HBM" % fixture,
            "config": {"workload": "%s, %s.param, 2^%d units %s per step" % (DESC[op], pname, args.log2n, "in the whole job" if args.strong else "per GPU"),
                       "units_per_gpu": n, "global_batch": n_job if args.strong else n * world,
                       "parallelism": "range-split x%d, no collectives" % world,
                       **({"param_extra": args.param_extra} if args.param_extra else {})},
            "per_rank_kernel_ms": [round(float(x), 3) for x in per_rank_ms],
            "kernel_only": {"value": round(n / sec, 1), "unit": UNIT[op] + " per GPU, events around the launches on rank 0"},
            "roofline": {
                "bound": "valu-int32-mac", "achieved": round(rate / 1e12, 4) if rate else None, "peak": round(MAC_PEAK / 1e12, 4),
                "unit": "TMAC/s (32x32->64 bit)", "frac": round(rate / MAC_PEAK, 4) if rate else None,
                # executed basis: SURVEY 8d's model prices F_q products only, and the reference's affine group law spends its
                # time in the mpz_invert of every group operation (reported below as reference_inversions_per_unit)
                "frac_basis": "executed" if exe_rate is not None else "algorithmic",
                "peak_measured": round(peak_measured / 1e12, 4),
                "algorithmic": None if alg_rate is None else {
                    "macs_per_unit": alg_prod * per_mul, "fq_products_per_unit": round(alg_prod, 1), "reference_inversions_per_unit": round(alg_inv, 1),
                    "achieved": round(alg_rate / 1e12, 4), "frac": round(alg_rate / MAC_PEAK, 4),
                    "note": "F_q products of the reference's algorithm (generic_pow_mpz over the affine group law / the towers) x (2 N^2 + N); its inversions are not priced"},
                "executed": None if exe_rate is None else {"macs_per_unit": exe, "achieved": round(exe_rate / 1e12, 4), "frac": round(exe_rate / MAC_PEAK, 4),
                                                            "note": "multiply-adds the kernel source executes (profiles/executed_macs.json)"},
                "traffic": bench.pmc_traffic(args.workload, n * unit_bytes, n), "kernel_ms": round(my_ms, 3),
                "hbm": {"achieved": round(n * unit_bytes / sec / 1e9, 3), "peak": 8000.0, "unit": "GB/s", "algorithmic_bytes_per_unit": unit_bytes},
            },
        }
        if world == 1:
            cb = None if args.no_cpu_baseline else cpu_baseline(param_path, op)
            out["cpu_baseline"] = cb if cb is not None else bench.container_cpu_baseline(args.workload)
        if bench.evidence_commit():
            out["commit"] = bench.evidence_commit()
        print(json.dumps(out), flush=True)
    if pp is not None:
        pp.clear()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
