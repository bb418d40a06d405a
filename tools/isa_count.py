#!/usr/bin/env python3
"""Static instruction counts per function / basic block of the gfx950 assembly that `make -C pbc_amd`
leaves in /tmp/pbc_hip_build (-save-temps).  Used with the call counts of the host mirror to model the
dynamic VALU instruction count of a kernel without a GPU (profiles/r01_notes.md, "static model").

  tools/isa_count.py FILE.s PATTERN [--blocks]
"""
import re
import sys
from collections import OrderedDict


def classify(mn):
    if mn.startswith("v_mad_u64_u32") or mn.startswith("v_mad_i64_i32"):
        return "mad64"
    if mn.startswith("v_"):
        return "valu"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.split("_")[0] in ("scratch", "buffer", "global", "flat"):
        return "vmem"
    return "other"


def parse(path):
    funcs = OrderedDict()
    cur = None
    blk = None
    with open(path) as fh:
        for line in fh:
            m = re.match(r"^(_Z[\w$.]+):", line)
            if m:
                cur = funcs.setdefault(m.group(1), OrderedDict())
                blk = cur.setdefault("entry", {"ins": [], "calls": []})
                continue
            if cur is None:
                continue
            if line.startswith(".Lfunc_end"):
                cur = None
                continue
            m = re.match(r"^(\.LBB[\w]+):", line)
            if m:
                blk = cur.setdefault(m.group(1), {"ins": [], "calls": []})
                continue
            s = line.strip()
            if not s or s.startswith(";") or s.startswith("."):
                continue
            mn = s.split()[0]
            blk["ins"].append((mn, s))
    return funcs


def totals(blocks):
    t = {}
    for b in blocks.values():
        for mn, _ in b["ins"]:
            c = classify(mn)
            t[c] = t.get(c, 0) + 1
    return t


def main():
    path, pat = sys.argv[1], sys.argv[2]
    funcs = parse(path)
    for name, blocks in funcs.items():
        if not re.search(pat, name):
            continue
        t = totals(blocks)
        v = t.get("valu", 0) + t.get("mad64", 0)
        print(f"{name[:90]:90s} blocks={len(blocks):4d} VALU={v:6d} mad64={t.get('mad64', 0):5d} salu={t.get('salu', 0):5d} "
              f"lds={t.get('lds', 0):4d} vmem={t.get('vmem', 0):4d}")
        if "--blocks" in sys.argv:
            for bn, b in blocks.items():
                tt = totals({bn: b})
                vv = tt.get("valu", 0) + tt.get("mad64", 0)
                tail = [s for mn, s in b["ins"] if mn.startswith("s_cbranch") or mn.startswith("s_branch")]
                calls = [re.search(r"(_Z\w+)@rel32@lo", s2).group(1) for mn, s2 in b["ins"] if "@rel32@lo" in s2 and "_Z" in s2]
                short = [re.sub(r"^_ZN3pbc\w*?(\d+)([a-z_0-9]+?)E.*$", r"\2", c)[:14] for c in calls]
                if "--hot" in sys.argv and vv < 40 and not calls:
                    continue
                tail = tail + ["CALLS " + ",".join(short)] if calls else tail
                print(f"   {bn:14s} VALU={vv:5d} mad64={tt.get('mad64', 0):4d} vmem={tt.get('vmem', 0):3d} lds={tt.get('lds', 0):3d}  {' | '.join(x.split(';')[0].strip() for x in tail)[:100]}")


if __name__ == "__main__":
    main()
