#!/usr/bin/env python3
"""bench.py -- pairings/sec on a 2^20-pair Type-A (a.param) batch per MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N>1 it
is launched under torch.distributed.run, one rank per GPU.  One "step" = one pass of the
hot path (one batched element_pairing launch) over one 2^20-pair batch that is already
resident in HBM.  Weak scaling: every rank owns its own 2^20-pair shard (range split, no
data-path collective; torch.distributed is used only for the barrier and the max-over-ranks
clock).  Rank 0 prints ONE JSON line.

Inputs: synthetic but valid group elements -- all (P_i, Q_j), i,j < 1024, from the
committed fixture tests/golden/a_chain1024.vec (P_i=(i+1)P0, Q_j=(j+1)Q0);
2^20 distinct pairs.  Before timing, the diagonal of one result batch is compared with the
reference's own outputs stored in the fixture (bit-exact) -- a wrong kernel cannot post a
number.
"""
import argparse
import json
import os
import struct
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import pbc_amd  # noqa: E402  (the product; raises if libpbc_hip.so is missing)
import bench_group  # noqa: E402  (the group operations as workloads: same contract, dispatched from main())


def ensure_built(dist, local_rank):
    """libpbc_hip.so normally travels with the tree; if it does not, local rank 0 compiles it (hipcc, ~3 min) and the
    other ranks wait at a barrier of the process group (not by polling the file's age)."""
    err = None
    if local_rank == 0:
        try:
            pbc_amd.build()
        except Exception as e:  # noqa: BLE001  -- reach the barrier first, then fail on every rank
            err = e
    if dist is not None:
        flags = [None] * dist.get_world_size()
        dist.all_gather_object(flags, repr(err) if err else "")
        bad = [f for f in flags if f]
        if bad:
            sys.exit("bench.py: building libpbc_hip.so failed: %s" % bad[0])
    elif err:
        raise err
    pbc_amd.lib()                      # every rank loads the same file; raises loudly when it is missing


HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Integer multiply-add roofline: v_mad_u64_u32 issues one wave64 instruction per 4 cycles per SIMD = 16 lanes per clock:
# 256 CU x 4 SIMD x 16 lanes x 2.4 GHz (MI355X_MICROARCH.md: 157.3 TFLOP/s fp32 vector = 78.6 T lane-FMA/s is the
# dual-issue / packed figure; the 32 x 32 + 64-bit multiply-add runs at half of it).  A constant, so that `frac` is
# reproducible; the live probe of the same pipe is reported beside it as peak_measured.
MAC_PEAK = 256 * 4 * 16 * 2.4e9


def load_vec(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"PBCVEC01"
    t, n, k, l1, l2, lt = struct.unpack("<6I", raw[8:32])
    a = np.frombuffer(raw, np.uint8)
    off = 32
    g1 = a[off:off + n * k * l1].reshape(n * k, l1); off += n * k * l1
    g2 = a[off:off + n * k * l2].reshape(n * k, l2); off += n * k * l2
    gt = a[off:off + n * lt].reshape(n, lt)
    return g1.copy(), g2.copy(), gt.copy()


class ClockSampler:
    """Shader clock and board power WHILE the timed steps run, read from the amdgpu sysfs files of this rank's card
    (pp_dpm_sclk marks the current level with '*', hwmon power1_average / power1_input in microwatts) by a polling
    thread; nothing is launched on the GPU.  The same kernel's time differs between boxes of the pool ("box
    spread"): with the clock printed beside `value` a reader can normalise.  Everything is optional: a box without the
    files reports null."""

    def __init__(self, index):
        import glob
        cards = sorted(p for p in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(os.path.join(p, "pp_dpm_sclk")))
        # a box shows the cards of the whole node in sysfs, HIP only its own: match by PCI address
        self.dev = None
        try:
            pr = torch.cuda.get_device_properties(index)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for c in cards:
                if os.path.basename(os.path.realpath(c)).lower() == bdf:
                    self.dev = c
        except Exception:  # noqa: BLE001
            pass
        if self.dev is None and len(cards) == 1:
            self.dev = cards[0]
        self.sclk, self.power, self.cap = [], [], None
        self._stop = False
        self._thread = None
        self.pfile = None
        if self.dev:
            import glob as g
            for name in ("power1_average", "power1_input"):
                f = g.glob(os.path.join(self.dev, "hwmon", "hwmon*", name))
                if f:
                    self.pfile = f[0]
                    break
            capf = g.glob(os.path.join(self.dev, "hwmon", "hwmon*", "power1_cap"))
            try:
                self.cap = int(open(capf[0]).read()) / 1e6 if capf else None
            except (OSError, ValueError):
                pass

    def _read(self):
        try:
            for line in open(os.path.join(self.dev, "pp_dpm_sclk")):
                if "*" in line:
                    self.sclk.append(float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()))
        except (OSError, ValueError, IndexError):
            pass
        if self.pfile:
            try:
                self.power.append(int(open(self.pfile).read()) / 1e6)
            except (OSError, ValueError):
                pass

    def _run(self):
        while not self._stop:
            self._read()
            time.sleep(0.002)

    def __enter__(self):
        if self.dev:
            import threading
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread:
            self._thread.join()

    def summary(self):
        if not self.dev or not self.sclk:
            return None
        out = {"sclk_mhz": {"mean": round(sum(self.sclk) / len(self.sclk), 1), "min": min(self.sclk), "max": max(self.sclk), "samples": len(self.sclk)},
               "source": self.dev + "/pp_dpm_sclk (+ hwmon power), polled every 2 ms during the timed steps"}
        if self.power:
            out["power_w"] = {"mean": round(sum(self.power) / len(self.power), 1), "max": round(max(self.power), 1), "cap": self.cap}
        return out


def kernel_source_sha():
    """hash of the kernel sources the library is built from (pbc_amd/csrc/* and the Makefile): PMC summaries carry the
    hash they were collected at (tools/summarise.py), and `roofline.traffic` quotes one only when it matches the tree"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "pbc_amd", "csrc", "*")) + [os.path.join(ROOT, "pbc_amd", "Makefile")]):
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


CPU_BASELINE_FILE = os.path.join(ROOT, "profiles", "r05_cpu_baselines.json")


def container_cpu_baseline(workload):
    """The reference's CPU rate for `workload` as measured once in the build container (tools/cpu_baselines.py ->
    profiles/r05_cpu_baselines.json; the reference needs no GPU).  Attached, marked "where": "build container", when the
    live CPU leg is switched off (--no-cpu-baseline: metered GPU-box minutes) -- the headline workloads run it live."""
    try:
        e = json.load(open(CPU_BASELINE_FILE))
        b = dict(e["workloads"][workload])
        b["where"] = "build container (%s, %d vCPU)" % (e["cpu_model"], e["cores"])
        return b
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(param_path, k=1, fixture=None, pp=False):
    """PBC+GMP (the unmodified reference compiled into oracle/_ref) on the host cores, on a
    bounded sample of the same workload; falls back to the single-core C port.  pp: pairing_pp_apply with a fixed
    first argument (benchmark/benchmark.c:75-81) instead of element_pairing."""
    import oracle  # checker/baseline only -- never on the measured GPU path
    cores = oracle.usable_cores()      # the cgroup quota, not the 256 logical CPUs a GPU box shows (VERDICT r5 "weak" 8)
    tool = oracle.REF_TOOL
    if os.path.exists(tool):
        try:
            def run(per_worker, workers):
                out = subprocess.run([tool, "bench", param_path, str(per_worker), str(-1 if pp else k), str(workers)],
                                     capture_output=True, text=True, timeout=300)
                return json.loads(out.stdout.strip().splitlines()[-1])
            # relative cost of one reference pairing (keeps the CPU sample at roughly 10-30 s)
            base = os.path.basename(param_path)
            scale = {"a.param": 1, "a1.param": 48, "e.param": 8, "f.param": 16, "f_256.param": 48, "g149.param": 12}.get(
                base, 8 if base.startswith("d") and base != "d159.param" else 4 if base.startswith("d") else 1) * k
            mult = int(os.environ.get("PBC_CPU_SAMPLE_SCALE", "1"))   # (tools/cpu_baselines.py: longer samples on a few-core host)
            one = run(max(16, 2048 // scale), 1)     # one core alone (~2 s)
            per_worker = max(8, 1024 // scale) * mult
            allc = run(per_worker, cores)            # one forked worker per usable core
            quota = None
            try:
                q, p = open("/sys/fs/cgroup/cpu.max").read().split()
                quota = None if q == "max" else float(q) / float(p)
            except Exception:  # noqa: BLE001
                pass
            return {"value": round(allc["units_per_s"], 1), "unit": "pairings/s" if k == 1 else "products/s", "cores": cores,
                    "kind": "reference",
                    "sample": "%d %s calls per worker x %d forked workers (%s), %.1f s wall"
                              % (per_worker, "pairing_pp_apply" if pp else "element_pairing" if k == 1 else "element_prod_pairing(k=%d)" % k, cores,
                                 os.path.basename(param_path), allc["wall_s"]),
                    "single_core": round(one["units_per_s"], 1),
                    "per_core_when_all_busy": round(allc["per_core"], 1),
                    "logical_cpus": os.cpu_count(), "cgroup_cpu_quota_cores": quota}
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("cpu_baseline: ref_tool failed (%r), using the C port\n" % (e,))
    O = oracle.OraclePairing(open(param_path).read())
    fx = fixture or {"a": "a_chain1024.vec", "d": "d_chain256.vec", "f": "f_chain128.vec"}[os.path.basename(param_path)[0]]
    g1, g2, _ = load_vec(os.path.join(ROOT, "tests", "golden", fx))
    reps = -(-128 // g1.shape[0])
    g1, g2 = np.tile(g1, (reps, 1)), np.tile(g2, (reps, 1))
    m = 128 // k * k
    t0 = time.time()
    if k == 1:
        O.pairing_batch(g1[:m], g2[:m])
    else:
        O.prod_pairing_batch(g1[:m], g2[:m], k)
    dt = time.time() - t0
    return {"value": round(m / k / dt, 1), "unit": "pairings/s" if k == 1 else "products/s", "cores": 1, "kind": "port",
            "sample": "%d units, single thread, oracle/pbc_oracle.c" % (m // k)}


def pmc_traffic(workload, alg_bytes=None, n=None, root=None, sha=None):
    """HBM bytes per launch from the rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this
    same command, taken by tools/collect.sh at the commit the summary names): the guide's gfx950 correction applied
    -- FETCH_SIZE counts 32-byte requests as if they were 64-byte ones on wide coalesced reads, so the read bytes are
    2 x FETCH_SIZE KB -- and `ratio_vs_algorithmic` = corrected bytes / the records the launch has to read and write.
    A ratio well above 1 is scratch (register spill / private array) traffic.  None when no summary is committed."""
    sha = sha or kernel_source_sha()
    stale = None
    for rel in ("profiles/r06_pmc_%s.json" % workload, "profiles/r05_pmc_%s.json" % workload, "profiles/r04_pmc_%s.json" % workload):
        path = os.path.join(root or ROOT, rel)
        if not os.path.exists(path):
            continue
        j = json.load(open(path))
        if "FETCH_SIZE" not in j or "WRITE_SIZE" not in j:
            continue
        if j.get("kernel_src_sha") != sha:
            # counters of OTHER kernel sources say nothing about this build: reported as stale, never as this run's traffic
            stale = stale or {"file": rel, "commit": j.get("commit", "")[:12], "kernel_src_sha": j.get("kernel_src_sha"),
                              "bytes_per_launch_then": int(2 * j["FETCH_SIZE"]["avg_per_launch"] * 1024 + j["WRITE_SIZE"]["avg_per_launch"] * 1024)}
            continue
        rd, wr = j["FETCH_SIZE"]["avg_per_launch"] * 1024, j["WRITE_SIZE"]["avg_per_launch"] * 1024
        out = {"bytes_per_launch": int(2 * rd + wr), "raw": {"FETCH_SIZE_bytes": int(rd), "WRITE_SIZE_bytes": int(wr)},
               "correction": "read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, gfx950); WRITE_SIZE as reported",
               "source": "%s (rocprofv3 --pmc passes of this command%s; not this run)" % (rel, ", commit " + j["commit"][:12] if "commit" in j else ""),
               "units_per_launch": j.get("units_per_launch"), "kernel_src_sha": sha}
        units = j.get("units_per_launch") or n
        if alg_bytes is not None and n and units:
            out["ratio_vs_algorithmic"] = round((2 * rd + wr) / (alg_bytes / n * units), 3)
        return out
    return {"bytes_per_launch": None, "kernel_src_sha": sha, "stale": stale,
            "note": "no PMC summary for these kernel sources under profiles/ (tools/collect.sh takes them)"} if stale else None


def evidence_commit():
    """the commit tools/collect.sh took this snapshot from (it writes .evidence_head next to this file and refuses a
    dirty tree); None outside an evidence run"""
    try:
        return open(os.path.join(ROOT, ".evidence_head")).read().strip() or None
    except OSError:
        return None


def executed_macs(workload):
    """multiply-adds the kernel source executes per unit (tools/executed_macs.py: counted on the host-compiled mirror of
    the kernel source, control flow is data-independent); None when the table has no entry"""
    path = os.path.join(ROOT, "profiles", "executed_macs.json")
    if os.path.exists(path):
        e = json.load(open(path)).get(workload)
        if e:
            return float(e["executed_macs_per_unit"])
    return None


WORKLOADS = {
    # name: (param, fixture, k, default log2 units, description)
    "a": ("a", "a_chain1024.vec", 1, 20, "Type A (a.param) element_pairing"),
    "d": ("d159", "d_chain256.vec", 1, 18, "Type D (d159.param) element_pairing"),
    "f": ("f", "f_chain128.vec", 1, 18, "Type F (f.param) element_pairing"),
    "d201": ("d201", "d201_rand12.vec", 1, 18, "Type D (d201.param, 7-word field) element_pairing"),
    "d224": ("d224", "d224_rand12.vec", 1, 18, "Type D (d224.param, 7-word field) element_pairing"),
    "a1": ("a1", "a1_chain8.vec", 1, 17, "Type A1 (a1.param, 1033-bit p) element_pairing"),
    "e": ("e", "e_chain8.vec", 1, 17, "Type E (e.param, k = 1, 1020-bit q) element_pairing"),
    "f256": ("f_256", "f_256_rand4.vec", 1, 16, "Type F (254-bit BN field from pbc_param_init_f_gen(256)) element_pairing"),
    "g": ("g149", "g149_chain64.vec", 1, 17, "Type G (g149.param, k = 10) element_pairing"),
    "d190": ("d278027-190-181", "d278027-190-181_rand12.vec", 1, 18,
             "Type D (d278027-190-181.param, 6-word field) element_pairing"),
    "a-prod16": ("a", "a_chain1024.vec", 16, 18, "Type A (a.param) element_prod_pairing, 16 terms"),
    "d-prod16": ("d159", "d_chain256.vec", 16, 18, "Type D (d159.param) element_prod_pairing, 16 terms"),
    "a-pp": ("a", "a_chain1024.vec", 1, 20, "Type A (a.param) pairing_pp_apply, fixed first argument"),
    "d-pp": ("d159", "d_chain256.vec", 1, 18, "Type D (d159.param) pairing_pp_apply, fixed first argument"),
    "a1-pp": ("a1", "a1_chain8.vec", 1, 17, "Type A1 (a1.param) pairing_pp_apply, fixed first argument"),
    "g-pp": ("g149", "g149_chain64.vec", 1, 17, "Type G (g149.param) pairing_pp_apply, fixed first argument"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="a", choices=sorted(WORKLOADS) + sorted(bench_group.GROUP_WORKLOADS),
                    help="default: the BASELINE.json metric config (2^20 Type-A pairings)")
    ap.add_argument("--log2n", type=int, default=None, help="units per GPU per step (with --strong: units of the whole job)")
    ap.add_argument("--extra-units", type=int, default=0,
                    help="units beyond 2^log2n (a job size no rank count divides: the range split gets ragged shares)")
    ap.add_argument("--strong", action="store_true",
                    help="the 2^log2n units are the whole job, range-split over the ranks (BASELINE config 5: "
                         "--workload a-prod16 --strong --gpus 8 = 2^18 products sharded across 8 GPUs)")
    ap.add_argument("--sweep", action="store_true",
                    help="instead of the bench line: launch time against batch size, n = 2^10 ... 2^log2n in powers of two plus one chip "
                         "residency (1024 workgroups x 128 lanes) -1 / +0 / +1, on one GPU; one JSON line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--param-extra", default="", help="lines appended to the parameter text (the library's A/B switches, "
                    "e.g. 'hip_prod_shared=1', comma-separated); reported in config.param_extra")
    ap.add_argument("--no-host-path", action="store_true",
                    help="skip the pinned-host -> host timing of the host-buffer entry point (reported beside `value`, never as it)")
    ap.add_argument("--host-path", action="store_true", help=argparse.SUPPRESS)     # round-1 spelling: now the default
    args = ap.parse_args()
    if args.workload in bench_group.GROUP_WORKLOADS:
        return bench_group.main(args, load_vec, ensure_built, MAC_PEAK)
    pname, fixture, k, dlog, desc = WORKLOADS[args.workload]
    if args.log2n is None:
        args.log2n = dlog

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run" % args.gpus)
    # test hooks (single-GPU boxes): PBC_BENCH_SAME_DEVICE=1 puts every rank on cuda:0 and
    # PBC_BENCH_BACKEND=gloo replaces RCCL for the barrier/clock -- the data path has no collective
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

    param_path = os.path.join(ROOT, "pbc_amd", "param", pname + ".param")
    pairing = pbc_amd.Pairing(open(param_path).read() + ("\n" + args.param_extra.replace("=", " ").replace(",", "\n") + "\n" if args.param_extra else ""))
    L1, L2, LT = pairing.length_in_bytes_G1, pairing.length_in_bytes_G2, pairing.length_in_bytes_GT
    n_job = (1 << args.log2n) + args.extra_units
    if args.strong:                      # range split of one job: rank r owns units [r n_job / world, (r + 1) n_job / world)
        first = rank * n_job // world
        n = (rank + 1) * n_job // world - first
    else:                                # weak scaling: every rank owns its own 2^log2n-unit shard
        first, n = 0, n_job
    g1, g2, gt_ref = load_vec(os.path.join(ROOT, "tests", "golden", fixture))
    D = g1.shape[0]
    rot = 0 if args.strong else (rank * 131) % D   # weak: rank r uses a rotated set of Q's so that shards differ
    d1 = torch.from_numpy(g1).cuda()
    d2 = torch.from_numpy(np.roll(g2, -rot, axis=0)).cuda()
    # term t of the job pairs P_(t // D mod D) with Q_(t mod D): D*D distinct pairs, tiled beyond
    t = torch.arange(first * k, (first + n) * k, device="cuda")
    G1 = d1[(t // D) % D].contiguous()
    G2 = d2[t % D].contiguous()
    GT = torch.empty(n, LT, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()

    pp = pairing.pp_init(g1[0]) if args.workload.endswith("-pp") else None

    def step():
        if pp is not None:
            pp.apply_dev(GT.data_ptr(), G2.data_ptr(), n, stream.cuda_stream)
        elif k == 1:
            pairing.element_pairing_dev(GT.data_ptr(), G1.data_ptr(), G2.data_ptr(), n, stream.cuda_stream)
        else:
            pairing.element_prod_pairing_dev(GT.data_ptr(), G1.data_ptr(), G2.data_ptr(), n, k, stream.cuda_stream)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    # correctness gate against the reference's own outputs stored in the fixture, on EVERY rank and shard: unit u of this
    # rank's shard is term t = first + u, which pairs P_a with Q_b, a = (t // D) % D, b = (t % D + rot) % D; wherever a == b
    # the result must be the fixture's e(P_a, Q_a).  k > 1: the term-order check below (also on every rank).
    gated_units = 0
    if pp is not None:
        if rot == 0 and first == 0 and not np.array_equal(GT[0].cpu().numpy(), gt_ref[0]):     # unit 0 is e(P_0, Q_0)
            sys.exit("bench.py: pp_apply differs from the reference fixture -- refusing to time")
        chk = torch.empty(256, LT, dtype=torch.uint8, device="cuda")
        P0 = d1[:1].expand(256, L1).contiguous()
        pairing.element_pairing_dev(chk.data_ptr(), P0.data_ptr(), G2[:256].contiguous().data_ptr(), 256, stream.cuda_stream)
        torch.cuda.synchronize()
        if not torch.equal(chk, GT[:256]):
            sys.exit("bench.py: pp_apply differs from element_pairing -- refusing to time")
        gated_units = 256
    elif k == 1:
        a_idx, b_idx = (t // D) % D, (t % D + rot) % D
        diag = torch.nonzero(a_idx == b_idx).flatten()[:D]
        if len(diag) and not np.array_equal(GT[diag].cpu().numpy(), gt_ref[a_idx[diag].cpu().numpy()]):
            sys.exit("bench.py: GPU results of rank %d differ from the reference fixture -- refusing to time" % rank)
        gated_units = int(len(diag))
    if k > 1:
        # a k-term product must equal the product of its k single pairings, taken from a
        # separate single-pairing launch: compare through a second product with permuted terms
        perm = torch.arange(k - 1, -1, -1, device="cuda")
        m = min(64, n)
        sel = (torch.arange(m, device="cuda")[:, None] * k + perm[None, :]).reshape(-1)
        GT2 = torch.empty(m, LT, dtype=torch.uint8, device="cuda")
        A1, A2 = G1[sel].contiguous(), G2[sel].contiguous()     # keep alive until the launch has run
        pairing.element_prod_pairing_dev(GT2.data_ptr(), A1.data_ptr(), A2.data_ptr(), m, k, stream.cuda_stream)
        torch.cuda.synchronize()
        if not torch.equal(GT2, GT[:m]):
            sys.exit("bench.py: product of pairings is not invariant under term order (rank %d) -- refusing to time" % rank)
        # ... and must equal the product of its k single pairings (element_pairing launches + GT products on the device)
        S = torch.empty(m * k, LT, dtype=torch.uint8, device="cuda")
        B1, B2 = G1[:m * k].contiguous(), G2[:m * k].contiguous()
        pairing.element_pairing_dev(S.data_ptr(), B1.data_ptr(), B2.data_ptr(), m * k, stream.cuda_stream)
        torch.cuda.synchronize()
        acc = S.cpu().numpy().reshape(m, k, LT)
        prod = acc[:, 0]
        for j in range(1, k):
            prod = pairing.element_mul_GT(prod, acc[:, j])
        if not np.array_equal(prod, GT[:m].cpu().numpy()):
            sys.exit("bench.py: product of pairings differs from the product of single pairings (rank %d) -- refusing to time" % rank)
        gated_units = m

    if args.sweep:
        # launch time against batch size (the resident kernels' time grows in steps of one chip residency; below it a
        # launch runs at partial occupancy): events around `steps` launches of the first m units
        res = 1024 * 128
        sizes = sorted(set([1 << e for e in range(10, args.log2n + 1)] + [m for m in (res - 1, res, res + 1, 2 * res, 2 * res + 1) if m <= n]))
        rows = []
        for m in sizes:
            def step_m(m=m):
                if pp is not None:
                    pp.apply_dev(GT.data_ptr(), G2.data_ptr(), m, stream.cuda_stream)
                elif k == 1:
                    pairing.element_pairing_dev(GT.data_ptr(), G1.data_ptr(), G2.data_ptr(), m, stream.cuda_stream)
                else:
                    pairing.element_prod_pairing_dev(GT.data_ptr(), G1.data_ptr(), G2.data_ptr(), m, k, stream.cuda_stream)
            step_m()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(args.steps):
                step_m()
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.steps
            rows.append({"n": m, "ms": round(ms, 4), "units_per_s": round(m / ms * 1e3, 1)})
        if rank == 0:
            print(json.dumps({"sweep": args.workload, "param_extra": args.param_extra, "unit": "pairings/s" if k == 1 else "products/s",
                              "steps": args.steps, "rows": rows}), flush=True)
        return
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    # The W untimed warm-up steps run HERE, directly before the timed region, and are topped up to at least 0.3 s of
    # launches: the shader clock falls back while the gate above compares on the host and needs tens of milliseconds of
    # load to climb again -- round 6 measured the same 31.6 M cycles per d159 launch take 15.8, 14.5, 13.5, 13.2, 13.2 ms
    # over the first five launches after an idle gap (GRBM_GUI_ACTIVE: 2.01 -> 2.38 GHz; profiles/r06_notes.md), which a
    # 14 ms kernel timed over a handful of steps shows and an 80 ms one does not.
    spin_t0, spun = time.perf_counter(), 0
    while spun < max(1, args.warmup) or (time.perf_counter() - spin_t0 < 0.3 and spun < 256):
        step()
        spun += 1
        if spun >= max(1, args.warmup):
            torch.cuda.synchronize()
    sync_all()
    sampler = ClockSampler(dev_index)
    with sampler:
        t0 = time.perf_counter()
        for a, b in evs:
            a.record(stream)
            step()
            b.record(stream)
        sync_all()
        dt = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    my_kern_ms = sum(kern_ms) / len(kern_ms)
    per_rank_ms = [my_kern_ms]
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        per_rank_ms = [None] * world
        dist.all_gather_object(per_rank_ms, my_kern_ms)        # a slow rank must be visible in the line rank 0 prints
    per_rank_gate = [{"first_unit": first, "units": n, "checked": gated_units}]
    if dist is not None:
        per_rank_gate = [None] * world
        dist.all_gather_object(per_rank_gate, {"first_unit": first, "units": n, "checked": gated_units})

    # the host-buffer entry point (what the PBC glue calls): pinned host memory in, host memory out, PCIe included
    host_path = None
    if rank == 0 and pp is None and not args.no_host_path:
        import ctypes
        h1, h2 = G1.cpu().pin_memory(), G2.cpu().pin_memory()
        hout = torch.empty(n, LT, dtype=torch.uint8).pin_memory()
        L = pbc_amd.lib()
        ts = []
        for _ in range(3):
            t1 = time.perf_counter()
            if k == 1:
                rc = L.pbc_hip_element_pairing_batch(pairing._h, ctypes.c_void_p(hout.data_ptr()), ctypes.c_void_p(h1.data_ptr()),
                                                     ctypes.c_void_p(h2.data_ptr()), n)
            else:
                rc = L.pbc_hip_element_prod_pairing_batch(pairing._h, ctypes.c_void_p(hout.data_ptr()), ctypes.c_void_p(h1.data_ptr()),
                                                          ctypes.c_void_p(h2.data_ptr()), n, k)
            ts.append(time.perf_counter() - t1)
            if rc:
                sys.exit("bench.py: host-buffer call failed: %s" % pbc_amd._err())
        if not torch.equal(hout, GT.cpu()):
            sys.exit("bench.py: host-buffer path differs from the device-pointer path")
        host_path = {"value": round(n / min(ts), 1), "ms": round(min(ts) * 1e3, 2), "bytes_over_pcie": n * (k * (L1 + L2) + LT),
                     "note": "pinned host buffers in, pinned host buffers out through the host-buffer entry point (the kernels read and "
                             "write page-locked caller memory in place over PCIe; pageable memory would be staged), best of 3 calls on "
                             "rank 0; PCIe-inclusive (SURVEY 8d's wall-clock form of the metric); never the reported `value`"}
        del h1, h2, hout

    if rank == 0:
        total_units = (n_job if args.strong else n * world) * args.steps
        value = total_units / dt
        avg_kern_s = my_kern_ms * 1e-3
        macs_per_unit = pairing.algorithmic_macs_per_unit(k)
        if pp is not None:
            macs_per_unit = pairing.algorithmic_macs_per_unit(-1)   # the reference's pp_apply algorithm
        exe_per_unit = executed_macs(args.workload)
        # live probes of the multiply-add pipe (register-only v_mad_u64_u32 chains with an SGPR factor)
        peak_measured = max(pbc_amd.int_mac_peak(13, 4000)[0], pbc_amd.int_mac_peak(14, 4000)[0])
        alg_rate = n * macs_per_unit / avg_kern_s
        exe_rate = n * exe_per_unit / avg_kern_s if exe_per_unit else None
        basis = "executed" if exe_rate is not None and exe_rate < alg_rate else "algorithmic"
        rate = exe_rate if basis == "executed" else alg_rate
        unit_bytes = k * (L1 + L2) + LT
        alg_bytes = n * unit_bytes
        unit_name = "pairings/s" if k == 1 else "products/s"
        out = {
            "metric": "pairings/sec on 2^20-batch Type-A (a.param)" if args.workload == "a"
                      else "%s per second, 2^%d%s batch" % (desc, args.log2n, " + %d" % args.extra_units if args.extra_units else ""),
            "value": round(value, 1),
            "unit": unit_name,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "warmup_launches": spun,             # W, topped up to 0.3 s of launches directly before the timed steps (clock ramp)
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "u32 (multi-word Montgomery F_q, bit-exact integer)",
            "data": "synthetic: (P_i,Q_j) cross pairs of tests/golden/%s (%d x %d distinct), resident in HBM" % (fixture, D, D),
            "config": {"workload": "%s, 2^%d%s units %s per step" % (desc, args.log2n, " + %d" % args.extra_units if args.extra_units else "",
                                                                     "in the whole job" if args.strong else "per GPU"),
                       "units_per_gpu": n, "terms_per_unit": k, "global_batch": n_job if args.strong else n * world,
                       "parallelism": "range-split x%d, no collectives" % world,
                       **({"param_extra": args.param_extra} if args.param_extra else {})},
            "per_rank_kernel_ms": [round(float(x), 3) for x in per_rank_ms],
            "per_rank_gate": per_rank_gate,      # every rank's shard is checked before timing (units compared bit for bit)
            "kernel_only": {"value": round(n / avg_kern_s, 1), "unit": unit_name + " per GPU, events around the launch on rank 0"},
            "host_path": host_path,
            "clocks": sampler.summary(),         # rank 0's card while the timed steps ran (null where sysfs has no amdgpu files)
            "roofline": {
                "bound": "valu-int32-mac",   # integer VALU throughput bounds this path, not HBM/MFMA
                "achieved": round(rate / 1e12, 4),
                "peak": round(MAC_PEAK / 1e12, 4),
                "unit": "TMAC/s (32x32->64 bit)",
                "frac": round(rate / MAC_PEAK, 4),
                # the smaller of the two work models: never credits multiply-adds the kernel did not execute, nor
                # executed ones beyond what the reference's algorithm needs
                "frac_basis": basis,
                "peak_measured": round(peak_measured / 1e12, 4),
                "algorithmic": {"macs_per_unit": macs_per_unit, "achieved": round(alg_rate / 1e12, 4),
                                "frac": round(alg_rate / MAC_PEAK, 4),
                                "note": "the reference algorithm's F_q products x (2 N^2 + N) word multiply-adds (SURVEY 8d)"},
                "executed": None if exe_rate is None else {
                    "macs_per_unit": exe_per_unit, "achieved": round(exe_rate / 1e12, 4), "frac": round(exe_rate / MAC_PEAK, 4),
                    "note": "multiply-adds the kernel source executes (profiles/executed_macs.json, tools/executed_macs.py)"},
                "traffic": pmc_traffic(args.workload, alg_bytes, n),
                "kernel_ms": round(avg_kern_s * 1e3, 3),
                "algorithmic_macs_per_unit": macs_per_unit,
                "executed_macs_per_unit": exe_per_unit,
                "hbm": {"achieved": round(alg_bytes / avg_kern_s / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg_bytes / avg_kern_s / 1e9 / HBM_PEAK_GBS, 6),
                        "algorithmic_bytes_per_unit": unit_bytes},
            },
        }
        if world == 1:
            cb = None if args.no_cpu_baseline else cpu_baseline(param_path, k, fixture, pp=args.workload.endswith("-pp"))
            out["cpu_baseline"] = cb if cb is not None else container_cpu_baseline(args.workload)
        if evidence_commit():
            out["commit"] = evidence_commit()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
