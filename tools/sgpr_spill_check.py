#!/usr/bin/env python3
"""Static check over gfx950 assembly for the defect behind the two GPU memory faults of rounds 3 / 4
(tools/gpu_faults.md): an SGPR spilled into a lane of a VGPR (v_writelane_b32) whose carrier VGPR is later itself saved
to / restored from scratch while EXEC is NOT all ones.  v_writelane / v_readlane ignore EXEC, scratch_store / scratch_load
do not: the lanes of inactive threads -- which hold spilled scalars such as return addresses, the scratch offset or loop
bounds -- are then not saved, and a later v_readlane returns garbage.

    tools/sgpr_spill_check.py FILE.s [name-pattern]      exit status 1 when a hazard is found

Per function: carriers = destination VGPRs of v_writelane_b32.  A hazard is a scratch store / load of a carrier that is
not inside a whole-wave region, i.e. between `s_or_saveexec_b64 sX, -1` (or s_xor_saveexec / `s_mov_b64 exec, -1`) and
the `s_mov_b64 exec, sX` that ends it.  The compiler's own prologue / epilogue saves of the carriers are whole-wave and
pass; what fails is a carrier that the register allocator spills again in the middle of the function."""
import re
import sys


def functions(path):
    cur, name = None, None
    for line in open(path):
        m = re.match(r"^(_Z[\w$.]+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            yield name, cur
            cur = None
            continue
        s = line.split(";")[0].strip()
        if s and not s.startswith("."):
            cur.append(s)
        elif s.startswith(".LBB"):
            cur.append(s)


def vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(name, ins):
    carriers = set()
    for s in ins:
        if s.startswith("v_writelane_b32"):
            carriers |= vregs(s.split()[1].rstrip(","))
    if not carriers:
        return []
    hazards, wwm = [], False
    for i, s in enumerate(ins):
        op = s.split()[0]
        if op in ("s_or_saveexec_b64", "s_xor_saveexec_b64") and s.rstrip().endswith("-1"):
            wwm = True
        elif op == "s_mov_b64" and s.split()[1].rstrip(",") == "exec":
            wwm = s.rstrip().endswith("-1")
        elif s.startswith(".LBB"):
            wwm = False                       # conservative: a whole-wave region does not span basic blocks
        elif op.startswith(("scratch_store", "scratch_load", "buffer_store", "buffer_load")):
            toks = [t.rstrip(",") for t in s.split()[1:]]
            data = toks[0] if "load" in op else (toks[1] if len(toks) > 1 else toks[0])
            regs = vregs(data) if "load" in op else set().union(*[vregs(t) for t in toks[:2]])
            hit = regs & carriers
            if hit and not wwm:
                hazards.append((i, s, sorted(hit)))
    return hazards


def main():
    path = sys.argv[1]
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    bad = 0
    for name, ins in functions(path):
        if pat and not pat.search(name):
            continue
        hz = check(name, ins)
        if hz:
            bad += 1
            print("%s: %d scratch accesses of SGPR-spill carrier VGPRs outside a whole-wave region" % (name[:110], len(hz)))
            for i, s, regs in hz[:4]:
                print("    [%d] %s    (carrier v%s)" % (i, s, regs))
    print("%d function(s) with hazards" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
