#!/usr/bin/env python3
"""Generator of the LEVEL PROGRAMS of the wave-per-pairing type g kernel (pbc_amd/csrc/pairing_gw.cuh, round 6).

The machine is tools/dw_gen.py's (one pairing per wavefront, every F_q element a slot of an LDS slot file, a LEVEL = every lane
computes one lazily reduced sum of F_q products from the slots its table row names).  Type g is the MNT family with k = 10:
F_q^10 = F_q^5[sqrt v], F_q^5 = F_q[x] / (x^5 + c4 x^4 + ... + c0) (ecc/g_param.c; the lane kernel: pairing_d.cuh's TypeMNT<5, 5>).
The formulas are those of tools/dw_gen.py with d = 5 -- cc_miller_no_denom_affine with the lines (a Qx + c) + (b Qy) sqrt v,
cc_tatepower as u = conj(m)^2 / N, ^(q + 1), one inversion and the Lucas ladder over Phi_10(q) / r -- but a coefficient of an F_q^5
product has up to nine terms (five direct, four folded high coefficients) and one of an F_q^10 product fourteen, so sums longer
than eight terms are CHAINED (a partial sum comes back in as `partial * 1`), and a packer moves sums to later levels where a level
would exceed the machine's capacity (16 sums when a sum has more than four terms: four lanes each; 31 otherwise).  ONE track: the
levels of a pairing in program order (gw_sched.h builds the straight line from the signed digits of r and the bits of
Phi_10(q) / r).  The script runs the tables on Python integers against the reference's vectors (tests/golden/g149_*.vec) and
writes pbc_amd/csrc/gw_tables.h.

  python tools/gw_gen.py            check and (re)write the header
  python tools/gw_gen.py --check    exit 1 if the committed header differs
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dw_gen import Slots, Prog, Node, ROOT, load_vec, naf_digits, param

D = 5
SLOTS = Slots()
CONSTS = ["ZERO", "ONE", "M1", "TWO", "M2", "THREE", "FOUR", "M8", "SIXTEEN", "HALF", "QVI", "A", "V", "NV"]
XP = [["XP%d_%d" % (D + j, k) for k in range(D)] for j in range(D - 1)]          # x^(5 + j) reduced: coefficient k
XQ = [["XQ%d_%d" % (i, k) for k in range(D)] for i in range(1, D)]               # (x^i)^q: coefficient k
POINT = ["X", "Y", "Z", "nZ", "W", "ZZ", "ZZZ", "Px", "Py", "nPy"]


def vec(name):
    return ["%s%d" % (name, i) for i in range(D)]


QX, QY, VQY = vec("Qx"), vec("Qy"), vec("VQy")
LX, LY, VLY = vec("Lx"), vec("Ly"), vec("VLy")
FX, FY = vec("f.x"), vec("f.y")
UX, UY = vec("u.x"), vec("u.y")                   # a second F_q^10 register: a term's Miller value (products)
LCO = ["La", "Lb", "Lc"]                          # the line's coefficients in F_q (pairing_pp_apply loads them from the table)
REGS5 = ["ux", "uy", "N", "gx", "gy", "wA", "wB", "Dn", "Bn", "tt", "iD", "iB", "P", "v0_", "v1_", "acc"]


# ---------------------------------------------------------------------------------------------------------------------
# sums of any length, products of polynomials
# ---------------------------------------------------------------------------------------------------------------------
def long_sum(p, terms, out=None, floor=1, first=8):
    """a sum of any number of terms as a chain of sums of at most eight (the partial sum returns as `partial * 1`); first: the
    length of the chain's first sum when there is more than one (4: that level stays one of plain four-term sums)"""
    terms = list(terms)
    assert terms
    node = None
    while terms:
        take = (8 if len(terms) <= 8 else first) if node is None else 7
        head, terms = terms[:take], terms[take:]
        if node is not None:
            head = [(node, "ONE")] + head
        node = p.sop(head, out=out if not terms else None, floor=floor)
    return node


def poly_sums(p, pairs, outs, extra=None, floor=1, first=8):
    """outs[k] = coefficient k of sum over (A, B) in pairs of A * B in F_q^5 (+ extra[k] terms): the high coefficients first (they
    are folded in as H_j * x^(5+j)[k]), then the low ones"""
    H = []
    for j in range(D - 1):
        t = [(A[i], B[D + j - i]) for A, B in pairs for i in range(D) if 0 <= D + j - i < D]
        H.append(long_sum(p, t, floor=floor))
    res = []
    for k in range(D):
        t = [(A[i], B[k - i]) for A, B in pairs for i in range(D) if 0 <= k - i < D]
        t += [(H[j], XP[j][k]) for j in range(D - 1)]
        if extra:
            t += extra[k]
        res.append(long_sum(p, t, out=outs[k] if outs else None, floor=floor, first=first))
    return res


def frob5(p, a, outs, neg=False):
    """(sum a_i x^i)^q = a_0 + sum_{i >= 1} a_i (x^i)^q; neg: the negative of it"""
    res = []
    for k in range(D):
        t = ([(a[0], "M1" if neg else "ONE")] if k == 0 else []) + [(a[i], ("N" if neg else "") + XQ[i - 1][k]) for i in range(1, D)]
        res.append(long_sum(p, t, out=outs[k] if outs else None))
    return res


def finish_packed(p):
    """levels by dependency depth, then sums of over-full levels moved down (16 sums where one has more than four terms, 31
    otherwise) until every level fits the machine"""
    for _ in range(400):
        p.finish()
        bad = None
        for lev, row in enumerate(p.levels):
            cap = 16 if max(len(n.terms) for n in row) > 4 else 31
            if len(row) > cap:
                bad = (lev, row, cap)
                break
        if bad is None:
            return p
        lev, row, cap = bad
        users = {n: 0 for n in row}
        for m in p.nodes:
            for t in m.terms:
                for op in t:
                    if isinstance(op, Node) and op in users:
                        users[op] += 1
        order = sorted(row, key=lambda n: (users[n], -len(n.terms)))
        for n in order[:len(row) - cap]:
            n.floor = lev + 2                         # (levels count from 1)
    raise AssertionError((p.name, "no packing"))


# ---------------------------------------------------------------------------------------------------------------------
# programs
# ---------------------------------------------------------------------------------------------------------------------
def line_values(p, la, lb, lc):
    """the line's value at Q: (a Qx + c) + (b Qy) sqrt v, and v times its second part"""
    for i in range(D):
        p.sop([(la, QX[i])] + ([(lc, "ONE")] if i == 0 else []), out=LX[i])
        p.mul(lb, QY[i], out=LY[i])
        p.mul(lb, VQY[i], out=VLY[i])


def prog_point_dbl():
    """V <- 2V and the tangent (tools/dw_gen.py prog_point_dbl: modified Jacobian coordinates, W = a Z^4)"""
    p = Prog("pt_dbl", SLOTS, "pt")
    emit_point_dbl(p)
    return finish_packed(p)


def emit_point_dbl(p):
    X, Y, Z, nZ, W = "X", "Y", "Z", "nZ", "W"
    XX = p.mul(X, X)
    YY = p.mul(Y, Y)
    Z3 = p.sop([(Y, Z), (Y, Z)], out="Z")
    nZ3 = p.sop([(Y, nZ), (Y, nZ)], out="nZ")
    W16 = p.mul(W, "SIXTEEN")
    M = p.sop([(XX, "THREE"), (W, "ONE")])
    S1 = p.mul(X, YY)
    Y4 = p.mul(YY, YY)
    lb = p.mul(nZ3, "ZZ", out="Lb")
    ZZn = p.mul(Z3, Z3, out="ZZ")
    X3 = p.sop([(M, M), (S1, "M8")], out="X")
    la = p.mul(M, "ZZ", out="La")
    MX = p.mul(M, X)
    nM = p.mul(M, "M1")
    S4 = p.mul(S1, "FOUR")
    p.mul(Y4, W16, out="W")
    p.mul(ZZn, Z3, out="ZZZ")
    p.sop([(M, S4), (nM, X3), (Y4, "M8")], out="Y")
    lc = p.sop([(YY, "TWO"), (MX, "M1")], out="Lc")
    line_values(p, la, lb, lc)


def prog_point_add(neg):
    """V <- V +- P and the chord (tools/dw_gen.py prog_point_add)"""
    p = Prog("pt_add%s" % ("m" if neg else "p"), SLOTS, "pt")
    X, Y, Z, nZ = "X", "Y", "Z", "nZ"
    Py = "nPy" if neg else "Py"
    H = p.sop([("Px", "ZZ"), (X, "M1")])
    R = p.sop([(Py, "ZZZ"), (Y, "M1")])
    nY = p.mul(Y, "M1")
    Z3 = p.mul(Z, H, out="Z")
    nZ3 = p.mul(nZ, H, out="nZ")
    HH = p.mul(H, H)
    nR = p.mul(R, "M1")
    lc = p.sop([(Z3, Py), (nR, "Px")], out="Lc")
    HHH = p.mul(HH, H)
    XHH = p.mul(X, HH)
    ZZ3 = p.mul(Z3, Z3, out="ZZ")
    X3 = p.sop([(R, R), (HHH, "M1"), (XHH, "M2")], out="X")
    RX = p.mul(R, XHH)
    nYH = p.mul(nY, HHH)
    Z43 = p.mul(ZZ3, ZZ3)
    p.mul(ZZ3, Z3, out="ZZZ")
    p.sop([(RX, "ONE"), (nR, X3), (nYH, "ONE")], out="Y")
    p.mul(Z43, "A", out="W")
    la = p.mul(R, "ONE", out="La")
    lb = p.mul(nZ3, "ONE", out="Lb")
    line_values(p, la, lb, lc)
    return finish_packed(p)


def prog_line_mul():
    """f <- f * l: x = fx lx + fy (v ly), y = fx ly + fy lx"""
    p = Prog("line_mul", SLOTS, "ft")
    poly_sums(p, [(FX, LX), (FY, VLY)], FX)
    poly_sums(p, [(FX, LY), (FY, LX)], FY)
    return finish_packed(p)


def prog_f_sqr():
    """f <- f^2: x = fx^2 + fy (v fy), y = (2 fx) fy"""
    p = Prog("f_sqr", SLOTS, "ft")
    emit_f_sqr(p)
    return finish_packed(p)


def emit_f_sqr(p):
    vfy = [p.mul(FY[i], "V") for i in range(D)]
    dfx = [p.mul(FX[i], "TWO") for i in range(D)]
    poly_sums(p, [(FX, FX), (FY, vfy)], FX)
    poly_sums(p, [(dfx, FY)], FY)


def prog_sqr_dbl():
    """f <- f^2 and V <- 2V side by side (they share nothing): the Miller loop's square with the NEXT step's doubling; the packer
    fits the doubling's sums into the square's levels where the machine has lanes left"""
    p = Prog("sqrdbl", SLOTS, "ft")
    emit_f_sqr(p)
    emit_point_dbl(p)
    return finish_packed(p)


def prog_ln_eval():
    """pairing_pp_apply: the line's value from coefficients a table supplied"""
    p = Prog("ln_eval", SLOTS, "pt")
    line_values(p, "La", "Lb", "Lc")
    return finish_packed(p)


def prog_mul_f_u():
    """f <- f * u (products: u takes a term's Miller value)"""
    p = Prog("mul_f_u", SLOTS, "ft")
    vuy = [p.mul(UY[i], "V") for i in range(D)]
    poly_sums(p, [(FX, UX), (FY, vuy)], FX)
    poly_sums(p, [(FX, UY), (FY, UX)], FY)
    return finish_packed(p)


def prog_fe1():
    """u = conj(m)^2 = (a^2 + v b^2) - 2ab s, N = a^2 - v b^2;  g = u^q (conjugated);  w = g u = A + B s;  Dn = N^q N"""
    p = Prog("fe1", SLOTS, "ft")
    ux, uy, N, gx, gy = vec("ux"), vec("uy"), vec("N"), vec("gx"), vec("gy")
    vb = [p.mul(FY[i], "V") for i in range(D)]
    nvb = [p.mul(FY[i], "NV") for i in range(D)]
    m2b = [p.mul(FY[i], "M2") for i in range(D)]
    U = poly_sums(p, [(FX, FX), (FY, vb)], ux)
    Nn = poly_sums(p, [(FX, FX), (FY, nvb)], N)
    Uy = poly_sums(p, [(FX, m2b)], uy)
    G = frob5(p, U, gx)
    Gy = frob5(p, Uy, gy, neg=True)                    # (x0 + x1 s)^q = x0^q - x1^q s
    vuy = [p.mul(Uy[i], "V") for i in range(D)]
    poly_sums(p, [(G, U), (Gy, vuy)], vec("wA"))
    poly_sums(p, [(G, Uy), (Gy, U)], vec("wB"))
    Nq = frob5(p, Nn, None)
    poly_sums(p, [(Nq, Nn)], vec("Dn"))
    return finish_packed(p)


def prog_fe2():
    """tt = Dn Bn (Bn = B, or 1 when B = 0: the lane-code test before this program);  its conjugates and their product acc;
    nrm = (tt acc)_0"""
    p = Prog("fe2", SLOTS, "ft")
    T = poly_sums(p, [(vec("Dn"), vec("Bn"))], vec("tt"))
    c1 = frob5(p, T, None)
    c2 = frob5(p, c1, None)
    c3 = frob5(p, c2, None)
    c4 = frob5(p, c3, None)
    a12 = poly_sums(p, [(c1, c2)], None)
    a34 = poly_sums(p, [(c3, c4)], None)
    acc = poly_sums(p, [(a12, a34)], vec("acc"))
    n = poly_sums(p, [(T, acc)], None)
    p.mul(n[0], "ONE", out="nrm")
    return finish_packed(p)


def prog_fe3():
    """1 / tt = acc ninv;  iD = Bn / tt = 1 / Dn, iB = Dn / tt = 1 / Bn;  P = 2 A / Dn;  v0 = 2, v1 = P"""
    p = Prog("fe3", SLOTS, "ft")
    ti = [p.mul("acc%d" % i, "ninv") for i in range(D)]
    iD = poly_sums(p, [(ti, vec("Bn"))], vec("iD"))
    poly_sums(p, [(ti, vec("Dn"))], vec("iB"))
    h0 = poly_sums(p, [(vec("wA"), iD)], None)
    for i in range(D):
        p.mul(h0[i], "TWO", out="P%d" % i)
        p.mul(h0[i], "TWO", out="v1_%d" % i)
        p.mul("TWO" if i == 0 else "ZERO", "ONE", out="v0_%d" % i)
    return finish_packed(p)


def prog_lucas(bit):
    """lucas_even (d_param.c:462-482): mm = v0 v1 - P;  bit: v1 <- v1^2 - 2, v0 <- mm;  else v0 <- v0^2 - 2, v1 <- mm"""
    p = Prog("lucas%d" % bit, SLOTS, "ft")
    v0, v1, P = vec("v0_"), vec("v1_"), vec("P")
    sq, keep = (v1, v0) if bit else (v0, v1)
    poly_sums(p, [(v0, v1)], keep, extra=[[(P[k], "M1")] for k in range(D)], first=4)
    poly_sums(p, [(sq, sq)], sq, extra=[[("ONE", "M2")] if k == 0 else [] for k in range(D)], first=4)
    return finish_packed(p)


def prog_fe4():
    """out = V_k / 2 + (P V_k - 2 V_{k-1}) Dn / (4 v B) s"""
    p = Prog("fe4", SLOTS, "ft")
    v0, v1, P = vec("v0_"), vec("v1_"), vec("P")
    t = poly_sums(p, [(P, v1)], None, extra=[[(v0[k], "M2")] for k in range(D)])
    t = poly_sums(p, [(t, vec("Dn"))], None)
    t = poly_sums(p, [(t, vec("iB"))], None)
    for i in range(D):
        p.mul(t[i], "QVI", out=FY[i])
        p.mul(v1[i], "HALF", out=FX[i])
    return finish_packed(p)


def build():
    for c in CONSTS:
        SLOTS.add(c)
    for row in XP + XQ:
        for n in row:
            SLOTS.add(n)
    for row in XQ:
        for n in row:
            SLOTS.add("N" + n)                        # the negated Frobenius constants
    for n in POINT + QX + QY + VQY + LX + LY + VLY + FX + FY + ["nrm", "ninv"] + LCO + UX + UY:
        SLOTS.add(n)
    for r in REGS5:
        for n in vec(r):
            SLOTS.add(n)
    progs = [prog_point_dbl(), prog_point_add(False), prog_point_add(True), prog_line_mul(), prog_f_sqr(), prog_sqr_dbl(), prog_ln_eval(), prog_mul_f_u(), prog_fe1(), prog_fe2(), prog_fe3(),
             prog_lucas(0), prog_lucas(1), prog_fe4()]
    return {p.name: p for p in progs}


# ---------------------------------------------------------------------------------------------------------------------
# the pairing as a sequence of program names (gw_sched.h is this function)
# ---------------------------------------------------------------------------------------------------------------------
def steps(plus, minus, rbits):
    """the Miller loop's steps in order: "dbl" / "addp" / "addm" / "sqr" """
    dig = lambda m: ((plus >> m) & 1) - ((minus >> m) & 1)
    st = []
    for m in range(rbits - 2, -1, -1):
        st.append("dbl")
        if m > 0 and dig(m):
            st.append("addm" if dig(m) < 0 else "addp")
        if m > 0:
            st.append("sqr")
    return st


def final_sequence(phik):
    seq = ["fe1", "OP_BZERO", "fe2", "OP_INV", "fe3"]
    nb = phik.bit_length()
    for j in range(nb - 1, -1, -1):
        seq.append("lucas%d" % ((phik >> j) & 1 if j else 0))
    seq.append("fe4")
    return seq


def pp_sequence(plus, minus, rbits):
    """pairing_pp_apply: line i of the table -> La, Lb, Lc (the first by an entry of its own, line i + 1 beside the first level of
    the product with line i), its value, the product; squares where the Miller loop has them"""
    seq, i = [("OP_LOADLINE", 0)], 0
    st = steps(plus, minus, rbits)
    nl = sum(1 for x in st if x != "sqr")
    for x in st:
        if x == "sqr":
            seq.append("f_sqr")
        else:
            seq += ["ln_eval", ("line_mul", i + 1 if i + 1 < nl else None)]
            i += 1
    return seq


def sequence(plus, minus, rbits, phik):
    return miller_sequence(plus, minus, rbits) + final_sequence(phik)


def miller_sequence(plus, minus, rbits):
    seq = []
    dig = lambda m: ((plus >> m) & 1) - ((minus >> m) & 1)
    for m in range(rbits - 2, -1, -1):
        seq += ["pt_dbl", "line_mul"]
        if m > 0 and dig(m):
            seq += ["pt_addm" if dig(m) < 0 else "pt_addp", "line_mul"]
        if m > 0:
            seq.append("f_sqr")
    fused = []                                       # a square and the doubling after it: one program
    for n in seq:
        if n == "pt_dbl" and fused and fused[-1] == "f_sqr":
            fused[-1] = "sqrdbl"
        else:
            fused.append(n)
    return fused


# ---------------------------------------------------------------------------------------------------------------------
# the model
# ---------------------------------------------------------------------------------------------------------------------
class Model:
    def __init__(self, pname, progs):
        P = param(pname)
        self.P, self.q, self.progs = P, P["q"], progs
        q, v = self.q, P["nqr"]
        self.fb = (q.bit_length() + 7) // 8
        self.env = {n: 0 for n in SLOTS.order}
        e = self.env
        inv = lambda x: pow(x, -1, q)
        c = [P["coeff%d" % i] for i in range(D)]
        xp = [[(-x) % q for x in c]]
        for j in range(1, D - 1):
            prev, top = xp[-1], xp[-1][D - 1]
            xp.append([(top * xp[0][0]) % q] + [(prev[i - 1] + top * xp[0][i]) % q for i in range(1, D)])
        self.xp = xp
        xq = [self.fpow([0, 1] + [0] * (D - 2), q)]
        for j in range(2, D):
            xq.append(self.fmul(xq[-1], xq[0]))
        e.update(ZERO=0, ONE=1, M1=q - 1, TWO=2, M2=q - 2, THREE=3, FOUR=4, M8=q - 8, SIXTEEN=16, HALF=inv(2), QVI=inv(4 * v % q), A=P["a"] % q, V=v,
                 NV=q - v)
        for j in range(D - 1):
            for k in range(D):
                e[XP[j][k]] = xp[j][k]
                e[XQ[j][k]] = xq[j][k]
                e["N" + XQ[j][k]] = (-xq[j][k]) % q
        self.v, self.vinv = v, inv(v)
        self.plus, self.minus, self.rbits = naf_digits(P["r"])
        self.phik = (q ** 4 - q ** 3 + q * q - q + 1) // P["r"]
        self.levels = 0
        self.flat = []

    def fmul(self, a, b):
        q = self.q
        d = [0] * (2 * D - 1)
        for i in range(D):
            for j in range(D):
                d[i + j] = (d[i + j] + a[i] * b[j]) % q
        return [(d[k] + sum(d[D + j] * self.xp[j][k] for j in range(D - 1))) % q for k in range(D)]

    def fpow(self, a, n):
        r = [1] + [0] * (D - 1)
        for bit in bin(n)[2:]:
            r = self.fmul(r, r)
            if bit == "1":
                r = self.fmul(r, a)
        return r

    def run(self, name):
        e, q = self.env, self.q
        if name == "OP_INV":
            e["ninv"] = pow(e["nrm"], -1, q) if e["nrm"] else 0
            self.flat.append(("op", "inv"))
            return
        if name == "OP_BZERO":
            b0 = all(e["wB%d" % i] == 0 for i in range(D))
            for i in range(D):
                e["Bn%d" % i] = (1 if i == 0 else 0) if b0 else e["wB%d" % i]
            self.flat.append(("op", "bzero"))
            return
        load = None
        if isinstance(name, tuple):                   # ("OP_LOADLINE", i) / (program, line to load beside its first level)
            name, load = name
            if name == "OP_LOADLINE":
                for c, n in enumerate(LCO):
                    e[n] = self.table[load][c]
                self.flat.append(("op", "loadline", load))
                return
        p = self.progs[name]
        for lev, row in enumerate(p.levels):
            T = max(len(n.terms) for n in row)
            assert len(row) <= (16 if T > 4 else 31), (name, lev, len(row), T)
            writes = []
            for n in row:
                acc = 0
                for a, b in n.terms:
                    acc += e[p.ref(a)] * e[p.ref(b)]
                writes.append((n.slot, acc % q))
            assert len({w[0] for w in writes}) == len(writes), (name, lev)
            for s, v in writes:
                e[s] = v
            self.levels += 1
            ld = load if lev == 0 else None
            if ld is not None:
                for c, n in enumerate(LCO):
                    e[n] = self.table[ld][c]
            self.flat.append(("level", name, lev, ld))

    def set_inputs(self, g1, g2):
        q, e, fb, P = self.q, self.env, self.fb, self.P
        gi = lambda b, i: int.from_bytes(b[fb * i:fb * (i + 1)], "big") % q
        Px, Py = gi(g1, 0), gi(g1, 1)
        Qx, Qy = [gi(g2, i) for i in range(D)], [gi(g2, D + i) for i in range(D)]
        a, b, v = P["a"], P["b"], self.v
        ok = ((Px * Px + a) * Px + b - Py * Py) % q == 0
        ta, tb = a * v * v % q, b * v * v * v % q
        x2 = self.fmul(Qx, Qx)
        x2[0] = (x2[0] + ta) % q
        x3 = self.fmul(x2, Qx)
        x3[0] = (x3[0] + tb) % q
        ok = ok and x3 == self.fmul(Qy, Qy)
        e.update(X=Px, Y=Py, Z=1, nZ=q - 1, W=a % q, ZZ=1, ZZZ=1, Px=Px, Py=Py, nPy=(q - Py) % q)
        for i in range(D):
            e[QX[i]] = Qx[i] * self.vinv % q
            e[QY[i]] = Qy[i] * self.vinv * self.vinv % q
            e[VQY[i]] = Qy[i] * self.vinv % q
            e[FX[i]], e[FY[i]] = (1 if i == 0 else 0), 0
        return ok

    def miller(self, g1, g2):
        ok = self.set_inputs(g1, g2)
        self.flat = []
        for n in miller_sequence(self.plus, self.minus, self.rbits):
            self.run(n)
        return ok

    def finish(self):
        for n in final_sequence(self.phik):
            self.run(n)
        self.flat.append(("op", "end"))
        return [self.env[n] for n in FX + FY]

    def pairing(self, g1, g2):
        ok = self.miller(g1, g2)
        r = self.finish()
        return r if ok else None

    def product(self, terms):
        """element_prod_pairing: one wavefront per TERM for the Miller values, then one per product -- f <- the first record, u <-
        each further one and mul_f_u, ONE final exponentiation"""
        vals, ok = [], True
        for g1, g2 in terms:
            ok = self.miller(g1, g2) and ok
            vals.append([self.env[n] for n in FX + FY])
        self.flat = []
        for n, v in zip(FX + FY, vals[0]):
            self.env[n] = v
        for val in vals[1:]:
            for n, v in zip(UX + UY, val):
                self.env[n] = v
            self.run("mul_f_u")
            self.flat.append(("op", "mark"))
        if len(vals) == 1:
            self.flat.append(("op", "mark"))
        r = self.finish()
        return r if ok else None

    def pp_table(self, g1, g2dummy):
        """what d_pp_init_lane leaves (in this script's scaling of the lines)"""
        ok = self.set_inputs(g1, g2dummy)
        keep, tab = self.flat, []
        for x in steps(self.plus, self.minus, self.rbits):
            if x == "sqr":
                continue
            self.run("pt_dbl" if x == "dbl" else "pt_" + x)
            tab.append([self.env[n] for n in LCO])
        self.flat = keep
        return tab, ok

    def pp_apply(self, table, g1, g2):
        ok = self.set_inputs(g1, g2)
        self.table = table
        self.flat = []
        for n in pp_sequence(self.plus, self.minus, self.rbits):
            self.run(n)
        r = self.finish()
        return r if ok else None


def check(progs, count=3):
    M = Model("g149", progs)
    bad = 0
    ident = [1] + [0] * (2 * D - 1)
    for name in ("g149_rand16.vec", "g149_edge10.vec"):
        g1, g2, gt = load_vec(os.path.join(ROOT, "tests", "golden", name))
        for i in range(min(count, len(gt))):
            M.levels = 0
            r = M.pairing(g1[i], g2[i])
            levels = M.levels
            want = [int.from_bytes(gt[i][M.fb * c:M.fb * c + M.fb], "big") for c in range(2 * D)]
            if (r or ident) != want:
                bad += 1
                print("MISMATCH", name, i)
            if i < 2:
                tab, ok = M.pp_table(g1[i], g2[i])
                r = M.pp_apply(tab, g1[i], g2[i])
                if (r or ident) != want:
                    bad += 1
                    print("MISMATCH (pp)", name, i)
    for name in ("g149_prod4x3.vec", "g149_prod3x4_edge.vec"):
        g1, g2, gt = load_vec(os.path.join(ROOT, "tests", "golden", name))
        k = len(g1) // len(gt)
        for i in range(len(gt)):
            r = M.product([(g1[i * k + t], g2[i * k + t]) for t in range(k)])
            if (r or ident) != [int.from_bytes(gt[i][M.fb * c:M.fb * c + M.fb], "big") for c in range(2 * D)]:
                bad += 1
                print("MISMATCH (product)", name, i)
    return bad, levels


def tables(progs):
    rows, index, pidx = [], [], []
    z = SLOTS["ZERO"]
    for name in sorted(progs):
        tab = progs[name].table()
        pidx.append((name, len(index), len(tab)))
        for T, lanes in tab:
            assert len(lanes) <= 31
            index.append((len(rows), T, len(lanes)))
            for o, xs, ys in lanes:
                b = [o] + xs + [z] * (8 - len(xs)) + ys + [z] * (8 - len(ys)) + [0, 0, 0]
                rows.append([b[4 * i] | b[4 * i + 1] << 8 | b[4 * i + 2] << 16 | b[4 * i + 3] << 24 for i in range(5)])
    return rows, index, pidx


_PROGS = None


def flat_schedule(kind="pairing", pname="g149"):
    """the packed schedules as the model executes them (gw_sched.h must build the same): "pairing"; "miller" (a term of a product);
    "finish" (the product with a term's value, a mark, the final exponentiation); "pp" (pairing_pp_apply)"""
    global _PROGS
    if _PROGS is None:
        _PROGS = build()
    progs = _PROGS
    rows, index, pidx = tables(progs)
    first = {name: f for name, f, c in pidx}
    M = Model(pname, progs)
    g1, g2, gt = load_vec(os.path.join(ROOT, "tests", "golden", "g149_rand16.vec"))
    if kind == "pairing":
        M.pairing(g1[0], g2[0])
    elif kind == "miller":
        M.miller(g1[0], g2[0])
        M.flat.append(("op", "end"))
    elif kind == "finish":
        M.product([(g1[0], g2[0]), (g1[1], g2[1])])
    else:
        tab, ok = M.pp_table(g1[0], g2[0])
        M.pp_apply(tab, g1[0], g2[0])
    out = []
    for e in M.flat:
        if e[0] == "level":
            r, T, lanes = index[first[e[1]] + e[2]]
            out.append(r | lanes << 12 | T << 34 | ((e[3] << 42 | 1 << 55) if e[3] is not None else 0))
        else:
            out.append({"bzero": 1, "inv": 2, "end": 3, "loadline": 4, "mark": 5}[e[1]] << 38 | ((e[2] << 42 | 1 << 55) if len(e) > 2 else 0))
    return out


def emit(progs):
    rows, index, pidx = tables(progs)
    keep = lambda n: not (n.split(".")[0] in ("ft", "pt") and ".t" in n)
    out = ["// gw_tables.h -- GENERATED by tools/gw_gen.py (do not edit): the level programs of the wave-per-pairing type g kernel",
           "// (pairing_gw.cuh) for the machine of pairing_dw.cuh.  One ROW of five dwords per sum: the slot it writes, eight x operands, eight",
           "// y operands, one byte each (terms a sum does not have name the ZERO slot); a LEVEL is (first row, terms per sum, working lanes);",
           "// a PROGRAM is a run of levels, found by name by the host (gw_sched.h).",
           "#pragma once", "#include <stdint.h>", "namespace pbc { namespace gw {",
           "constexpr int kSlots = %d;" % len(SLOTS.order)]
    out.append("enum Slot : int { " + ", ".join("S_%s = %d" % (n.replace(".", "_"), i) for i, n in enumerate(SLOTS.order) if keep(n)) + " };")
    out.append("enum { OP_LEVEL = 0, OP_BZERO = 1, OP_INV = 2, OP_END = 3, OP_LOADLINE = 4, OP_MARK = 5 };      // schedule entries (gw_sched.h; dw_tables.h's values)")
    out.append("struct LevelRef { uint16_t row; uint8_t T, lanes; };")
    out.append("struct ProgRef { const char *name; uint16_t first, count; };")
    out.append("constexpr int kProgs = %d, kLevels = %d, kRows = %d;" % (len(pidx), len(index), len(rows)))
    out.append("static const ProgRef h_prog[kProgs] = {" + ", ".join('{"%s", %d, %d}' % x for x in pidx) + "};")
    out.append("static const LevelRef h_level[kLevels] = {" + ", ".join("{%d, %d, %d}" % x for x in index) + "};")
    out.append("__device__ const uint32_t g_rows[kRows * 5] = {" + ",".join("0x%xu" % w for r in rows for w in r) + "};")
    out.append("} }  // namespace pbc::gw")
    return "\n".join(out) + "\n"


def main():
    global _PROGS
    progs = build()
    _PROGS = progs
    print("slots", len(SLOTS.order), "programs", len(progs), "rows", sum(len(r) for p in progs.values() for r in p.levels))
    for n in sorted(progs):
        p = progs[n]
        print("%-9s levels %2d  sums per level %s  terms %s" % (n, len(p.levels), [len(r) for r in p.levels], [max(len(x.terms) for x in r) for r in p.levels]))
    assert len(SLOTS.order) <= 255, len(SLOTS.order)
    bad, levels = check(progs)
    print("levels per pairing %d; vectors: %s" % (levels, "MISMATCH" if bad else "ok"))
    if bad:
        sys.exit(1)
    text = emit(progs)
    path = os.path.join(ROOT, "pbc_amd", "csrc", "gw_tables.h")
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(path) and open(path).read() == text else 1)
    open(path, "w").write(text)
    print("wrote", path, len(text), "bytes")


if __name__ == "__main__":
    main()
