#!/usr/bin/env python3
"""Static check of gfx950 assembly for registers that a kernel keeps live across a call although the callee's code
changes them (tools/gpu_faults.md).  With inter-procedural register allocation (the backend's default for functions with
internal linkage) a caller keeps values in registers the callee's recorded clobber mask does not name -- also in
caller-saved ones (s0-s29, s40-s47, ...; v0-v39, v48-v55, ...).  If the mask is narrower than what the callee's body
really writes, the caller reads garbage after the call: the type g resident-loop fault of rounds 3-5, which
`-mllvm -enable-ipra=0` removes.

    tools/ipra_check.py FILE.s KERNEL_MANGLED_NAME [--vgpr]      exit status 1 when a finding is reported

Method (conservative in the listing, exact in nothing -- this reads assembly, not MIR):
  * every s_swappc is resolved to its callee by tracking function addresses through s_getpc / s_add_u32 f@rel32, SGPR
    copies and SGPR spill lanes (v_writelane / v_readlane);
  * a callee's NET clobber set = registers its body (and, transitively, its callees) writes, minus registers it saves to a
    spill lane or a scratch slot before the first change and restores by the last one;
  * forward data flow over the kernel's control-flow graph (basic blocks from labels and branches): a register is
    "poisoned" from a call whose callee's net clobber set holds it until the kernel writes it again, along every path
    including loop back edges; a READ of a poisoned register is a finding."""
import re
import sys


def parse(path):
    funcs, name = {}, None
    for line in open(path):
        m = re.match(r"^(_Z[\w$.]+):", line)
        if m:
            name = m.group(1)
            funcs[name] = []
            continue
        if line.startswith(".Lfunc_end"):
            name = None
            continue
        if name is not None:
            s = line.split(";")[0].strip()
            if s:
                funcs[name].append(s)
    return funcs


def regs(tok, kind):
    tok = tok.strip().rstrip(",")
    m = re.match(kind + r"\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(kind + r"(\d+)$", tok)
    return {int(m.group(1))} if m else set()


NODEF = ("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_setpc", "s_barrier", "s_endpgm", "s_bitcmp", "s_sleep",
         "s_setprio", "scratch_store", "global_store", "ds_write", "buffer_store", "flat_store", "s_sendmsg", "s_setreg",
         "global_atomic", "s_dcache", "s_icache", "s_trap", "s_code_end", "ds_add", "ds_gws")


def defs_uses(s, kind):
    """(registers of `kind` written, read) by one instruction -- `kind` is "s" or "v\""""
    p = s.split(None, 1)
    if len(p) < 2 or s.endswith(":") or s.startswith("."):
        return set(), set()
    op, ops = p[0], [o.strip() for o in p[1].split(",")]
    d, start = set(), 0
    if not op.startswith(NODEF):
        if kind == "s" or not op.startswith(("s_", "v_cmp", "v_readlane", "v_readfirstlane")):
            d |= regs(ops[0], kind)
        start = 1
        if kind == "s" and op.startswith("v_") and len(ops) > 1 and ("_co_" in op or op.startswith(("v_mad_u64", "v_mad_i64", "v_div_scale"))):
            d |= regs(ops[1], kind)
            start = 2
    u = set()
    for o in ops[start:]:
        for t in re.findall(kind + r"\[\d+:\d+\]|\b" + kind + r"\d+\b", o):
            u |= regs(t, kind)
    if kind == "v" and op.startswith(("v_writelane", "v_mac", "v_fmac")):
        u |= regs(ops[0], kind)
    return d, u


class Analysis:
    def __init__(self, funcs, kind):
        self.funcs, self.kind, self.memo = funcs, kind, {}

    def preserved(self, f):
        """registers the body saves (lane or s32-relative scratch slot) before changing and restores last"""
        body, kind = self.funcs.get(f, []), self.kind
        first, last = {}, {}
        for s in body:
            if kind == "s":
                m = re.match(r"v_writelane_b32 (v\d+), s(\d+), (\d+)", s)
                if m:
                    first.setdefault(int(m.group(2)), ("save", (m.group(1), int(m.group(3)))))
                    continue
                m = re.match(r"v_readlane_b32 s(\d+), (v\d+), (\d+)", s)
                if m:
                    r = int(m.group(1))
                    last[r] = ("restore", (m.group(2), int(m.group(3))))
                    first.setdefault(r, ("def", None))
                    continue
            else:
                m = re.match(r"scratch_store_dword(?:x\d)? off, (v\[\d+:\d+\]|v\d+), s3[23](?: offset:(\d+))?", s)
                if m:
                    for i, r in enumerate(sorted(regs(m.group(1), "v"))):
                        first.setdefault(r, ("save", int(m.group(2) or 0) + 4 * i))
                    continue
                m = re.match(r"scratch_load_dword(?:x\d)? (v\[\d+:\d+\]|v\d+), off, s3[23](?: offset:(\d+))?", s)
                if m:
                    for i, r in enumerate(sorted(regs(m.group(1), "v"))):
                        last[r] = ("restore", int(m.group(2) or 0) + 4 * i)
                        first.setdefault(r, ("def", None))
                    continue
            d, u = defs_uses(s, kind)
            for r in u:
                first.setdefault(r, ("use", None))
            for r in d:
                first.setdefault(r, ("def", None))
                last[r] = ("def", None)
        return {r for r, (k, slot) in first.items() if k == "save" and last.get(r) == ("restore", slot)}

    def clob(self, f, stack=()):
        if f in self.memo:
            return self.memo[f]
        w = set()
        for s in self.funcs.get(f, []):
            w |= defs_uses(s, self.kind)[0]
            m = re.search(r"(_Z[\w$.]+)@rel32@lo", s)
            if m and m.group(1) not in stack and m.group(1) != f:
                w |= self.clob(m.group(1), stack + (f,))
        w -= self.preserved(f)
        self.memo[f] = w
        return w


def callees_of(ins):
    """index of every s_swappc -> callee name (function addresses followed through copies and spill lanes)"""
    sreg, slot, out = {}, {}, {}
    for i, s in enumerate(ins):
        m = re.match(r"s_add_u32 s(\d+), s\d+, (_Z[\w$.]+)@rel32@lo", s)
        if m:
            sreg[int(m.group(1))] = m.group(2)
            continue
        m = re.match(r"v_writelane_b32 (v\d+), s(\d+), (\d+)", s)
        if m:
            k = (m.group(1), int(m.group(3)))
            if int(m.group(2)) in sreg:
                slot[k] = sreg[int(m.group(2))]
            else:
                slot.pop(k, None)
            continue
        m = re.match(r"v_readlane_b32 s(\d+), (v\d+), (\d+)", s)
        if m:
            k = (m.group(2), int(m.group(3)))
            if k in slot:
                sreg[int(m.group(1))] = slot[k]
            else:
                sreg.pop(int(m.group(1)), None)
            continue
        m = re.match(r"s_mov_b64 s\[(\d+):\d+\], s\[(\d+):\d+\]", s)
        if m:
            a, c = int(m.group(1)), int(m.group(2))
            if c in sreg:
                sreg[a] = sreg[c]
            else:
                sreg.pop(a, None)
            continue
        m = re.match(r"s_swappc_b64 s\[30:31\], s\[(\d+):\d+\]", s)
        if m:
            out[i] = sreg.get(int(m.group(1)))
            continue
        for r in defs_uses(s, "s")[0]:
            sreg.pop(r, None)
    return out


def check(funcs, kernel, kind):
    """forward data flow over the kernel's control-flow graph: a register is POISONED after a call whose callee's net
    clobber set holds it, until the kernel writes it again; a read of a poisoned register is a finding"""
    ins = funcs[kernel]
    A = Analysis(funcs, kind)
    calls = callees_of(ins)
    top = 106 if kind == "s" else 512
    skip = set(range(0, int(__import__("os").environ.get("IPRA_RET", "32")))) if kind == "v" else {30, 31}   # return values (v0 .. IPRA_RET - 1) / the return address itself
    # basic blocks
    leaders = {0}
    labels = {}
    for i, s in enumerate(ins):
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = i
            leaders.add(i)
        if s.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")) and i + 1 < len(ins):
            leaders.add(i + 1)
    order = sorted(leaders)
    end_of = {b: (order[k + 1] if k + 1 < len(order) else len(ins)) for k, b in enumerate(order)}
    succ = {}
    for b in order:
        e = end_of[b]
        last = ins[e - 1]
        out = []
        m = re.match(r"s_c?branch\S* (\.LBB\d+_\d+)", last)
        if m and m.group(1) in labels:
            out.append(labels[m.group(1)])
        if not last.startswith(("s_branch", "s_endpgm", "s_setpc")) and e < len(ins):
            out.append(e)
        succ[b] = out
    du = [defs_uses(s, kind) for s in ins]
    cl = {i: ({r for r in A.clob(f) if r < top} - skip) for i, f in calls.items() if f}
    state_in = {b: None for b in order}
    state_in[0] = frozenset()
    work = [0]
    findings = {}
    while work:
        b = work.pop()
        st = set(state_in[b])
        for i in range(b, end_of[b]):
            d, u = du[i]
            hit = u & st
            if hit and not ins[i].startswith("s_swappc"):
                for r in hit:
                    findings.setdefault((i, r), None)
            if i in cl:
                st |= cl[i]
                st -= d                                               # (s_swappc writes s[30:31])
            else:
                st -= d
        fs = frozenset(st)
        for n in succ[b]:
            old = state_in[n]
            new = fs if old is None else (old | fs)
            if new != old:
                state_in[n] = new
                work.append(n)
    # name the call that poisoned each finding (nearest preceding call in layout whose clobber set holds the register)
    out = []
    for (i, r) in sorted(findings):
        src = max((j for j in cl if j < i and r in cl[j]), default=None)
        if src is None:
            src = max((j for j in cl if r in cl[j]), default=None)
        out.append("%s%d read @%d by `%s` after the callee of call@%s (%s) changed it and nothing rewrote it"
                   % (kind, r, i, ins[i], src, calls[src][-44:] if src is not None else "?"))
    return out, len(calls), sum(1 for f in calls.values() if not f), A


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    kind = "v" if "--vgpr" in sys.argv else "s"
    funcs = parse(path)
    findings, ncalls, unknown, A = check(funcs, kernel, kind)
    seen = set()
    for f in findings:
        key = re.sub(r"@\d+", "", f)
        if key in seen:
            continue
        seen.add(key)
        if len(seen) <= 40:
            print(f)
    print("%s: %d calls (%d unresolved), %d finding(s) (%d distinct)" % (kernel[:60], ncalls, unknown, len(findings), len(seen)))
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
