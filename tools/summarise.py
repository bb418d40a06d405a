"""gpurun_out/ev_$ROUND (tools/evidence.sh on the GPU box, started by tools/collect.sh; ROUND defaults to r05) -> committed
summaries profiles/$ROUND_*: bench lines, kernel-trace stats of the dominant kernels, PMC counters per launch (averaged over the
full-size launches of the workload's main kernel; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them -- bench.py
applies the gfx950 correction when it quotes them), sweeps, the suite's tail.  Refuses to write when the tree is dirty
or HEAD is not the commit the evidence run was stamped with: one commit per evidence set."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
RND = os.environ.get("ROUND", "r06")
EV = "gpurun_out/ev_" + RND
head = subprocess.check_output(["git", "rev-parse", "HEAD"], text=True).strip()
dirty = subprocess.check_output(["git", "status", "--porcelain"], text=True).strip()
stamp = open(EV + "/HEAD").read().strip() if os.path.exists(EV + "/HEAD") else None
if dirty or stamp != head:
    sys.exit("refusing to write profiles/: %s" % ("work tree is dirty" if dirty else "evidence is from %s, HEAD is %s" % (stamp, head)))
written = []
MANIFEST = "profiles/%s_MANIFEST.json" % RND


def put(src, dst):
    shutil.copy(src, dst)
    written.append(dst)


for f in sorted(glob.glob(EV + "/bench_*.json")):
    if os.path.getsize(f):
        line = open(f).readline()
        assert json.loads(line).get("commit") == head, f
        put(f, "profiles/%s_" % RND + os.path.basename(f))
for f in sorted(glob.glob(EV + "/sweep_*.json")):
    if os.path.getsize(f):
        put(f, "profiles/%s_" % RND + os.path.basename(f))
for src, dst, what in (("wave_latency.txt", "profiles/%s_closing_wave_latency.txt" % RND, "python tools/wave_latency.py 1 256 512 1024 2048 4096 5120"),
                       ("tail.txt", "profiles/%s_closing_tail.txt" % RND, "python tools/tail_latency.py"),
                       ("dwave_latency.txt", "profiles/%s_dwave_latency.txt" % RND, "python tools/dwave_latency.py 1 16 256 1024 2048 3072 4096 8192"),
                       ("small_batches.txt", "profiles/%s_small_batches.txt" % RND, "python tools/dwave_latency.py {prod K, pp} ...; DW_PARAM={d190, d201, d224, f, g149} python tools/dwave_latency.py ...; AG_PARAM={a1, a_160_1024, e} python tools/agwave_latency.py ... (tools/evidence.sh)"),
                       ("glue.txt", "profiles/%s_closing_glue.txt" % RND, "oracle/_ref/glue_test {a,d159,f,d201,d224,g149,a1,e}.param 100 latency (a1: 30); {a,d159}.param 1048576 bench")):
    if os.path.exists(EV + "/" + src) and os.path.getsize(EV + "/" + src):
        with open(dst, "w") as fh:
            fh.write("commit %s: %s\n" % (head, what))
            fh.write("".join(l for l in open(EV + "/" + src) if "amdgpu.ids" not in l))
        written.append(dst)
if os.path.exists(EV + "/pytest.log"):
    with open("profiles/%s_gputest_tail.txt" % RND, "w") as fh:
        fh.write("commit %s: python -m pytest tests -m gpu -q\n" % head)
        fh.writelines(open(EV + "/pytest.log").readlines()[-6:])
    written.append("profiles/%s_gputest_tail.txt" % RND)
MAIN = {"a": "al_pairing_kernel", "d": "d_prod_pairing_kernel", "f": "f_prod_pairing_kernel", "a-prod16": "al_miller_kernel",
        "a-g1-mul": "al_gmul_kernel", "f-gt-pow": "f_gtpow_kernel", "d-prod16": "d_prod_pairing_kernel", "d190": "d_prod_pairing_kernel",
        "a-pp": "al_pp_apply_kernel", "e": "e_prod_pairing_kernel", "a1": "a1_prod_pairing_kernel", "a1-pp": "a1_pp_apply_kernel"}
for w, kern in MAIN.items():
    ks = glob.glob("%s/kt_%s/**/*kernel_stats.csv" % (EV, w), recursive=True)
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        dst = "profiles/%s_kernel_stats_%s.csv" % (RND, w)
        # the timed steps alone: bench.py warms the clock up with extra launches right before them (0.3 s), and the average
        # over ALL launches of a short kernel is dominated by the cold ones (profiles/r06_notes.md, "clock ramp")
        timed = ""
        tr = glob.glob("%s/kt_%s/**/*kernel_trace.csv" % (EV, w), recursive=True)
        if tr:
            L = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0))
                 for r in csv.DictReader(open(tr[0])) if kern in r["Kernel_Name"]]
            full = sorted(x for x in L if x[2] == max(y[2] for y in L))
            last = [x[1] for x in full[-6:]]
            if last:
                timed = "# the last %d full-size launches (bench.py's timed steps): average %.0f ns, min %d, max %d; all %d full-size launches: %s\n" % (
                    len(last), sum(last) / len(last), min(last), max(last), len(full), " ".join("%.2f" % (x[1] / 1e6) for x in full) + " ms")
        with open(dst, "w") as fh:
            fh.write("# commit %s: rocprofv3 --kernel-trace --stats -- python bench.py --workload %s --steps 6 --warmup 2 --no-cpu-baseline --no-host-path\n" % (head, w))
            fh.write(timed)
            fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
            for r in rows[:6]:
                fh.write('"%s",%s,%s,%s,%s,%s,%s\n' % (r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                      r["Percentage"], r["MinNs"], r["MaxNs"]))
        written.append(dst)
    out = {}
    for f in glob.glob("%s/pmc?_%s/**/*counter_collection.csv" % (EV, w), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            v = [x for x in v if x >= 0.5 * max(v)]       # the timed full-size launches, not bench.py's small gate launches
            out[k] = {"launches": len(v), "avg_per_launch": sum(v) / len(v)}
    if out:
        out["commit"] = head
        out["kernel"] = kern
        sys.path.insert(0, ROOT)
        import bench                                     # the hash bench.py's roofline.traffic checks before it quotes this file
        out["kernel_src_sha"] = bench.kernel_source_sha()
        b = "profiles/%s_bench_%s.json" % (RND, w)
        if os.path.exists(b):
            out["units_per_launch"] = json.loads(open(b).readline())["config"].get("units_per_gpu")
        dst = "profiles/%s_pmc_%s.json" % (RND, w)
        json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
        written.append(dst)
        print(w, {k: "%.4g" % v["avg_per_launch"] for k, v in sorted(out.items()) if isinstance(v, dict)})
if os.environ.get("ADDENDUM"):
    man = json.load(open(MANIFEST))
    man["files"] = sorted(set(man["files"]) - set(written))
    for a in man.get("addenda", []):
        a["files"] = sorted(set(a["files"]) - set(written))
    man.setdefault("addenda", []).append({"commit": head, "written": time.strftime("%Y-%m-%d %H:%M:%S"), "files": sorted(written),
                                          "why": os.environ.get("ADDENDUM_WHY", "partial re-collection after a change to the kernels of these workloads")})
    man["addenda"] = [a for a in man["addenda"] if a["files"]]
    json.dump(man, open(MANIFEST, "w"), indent=1)
else:
    json.dump({"commit": head, "written": time.strftime("%Y-%m-%d %H:%M:%S"), "files": sorted(written),
               "how": "tools/collect.sh (clean tree at this commit) -> gpurun tools/evidence.sh -> tools/summarise.py"},
              open(MANIFEST, "w"), indent=1)
print("%d files under profiles/ from commit %s" % (len(written) + 1, head[:12]))
