#!/usr/bin/env python3
"""Generator of the LEVEL PROGRAMS of the wave-per-pairing type f kernel (pbc_amd/csrc/pairing_fw.cuh, round 6).

The machine is tools/dw_gen.py's (one pairing per wavefront, every F_q element a slot of an LDS slot file, a LEVEL = every lane
computes one lazily reduced sum of F_q products from the slots its table row names): this script describes the type f
pairing -- BN curve, F_q^12 = F_q^2[X] / (X^6 + alpha), F_q^2 = F_q[s] / (s^2 - beta) -- as such sums:
  * the Miller loop of cc_miller_no_denom (ecc/f_param.c:216-233) on E(F_q) in Jacobian coordinates with the lines
    c + (a Qx') X^4 + (b Qy') X^3 of f_miller_evalfn (:109-149) multiplied into the accumulator in ONE level (five terms per
    F_q coefficient), squarings in two (the doubled / beta- / xi-scaled copies first, then eight-term sums);
  * f_tateexp (:250-283): the easy part with polymod_invert's norm trick (one F_q inversion), the hard part as the BN vector
    chain of pairing_f.cuh f_hard_bn (three powers by |x|, Frobenius maps, thirteen products) -- general products take three
    levels (a coefficient has twelve terms: six + the first half as a thirteenth... seven);
  * ONE track: the levels of a pairing in program order (the schedule is a straight line; fw_sched.h builds it from the signed
    digits of r and the bits of x).
It runs the tables on Python integers against the reference's vectors (tests/golden/f_*.vec) and writes
pbc_amd/csrc/fw_tables.h.

  python tools/fw_gen.py            check and (re)write the header
  python tools/fw_gen.py --check    exit 1 if the committed header differs
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dw_gen
from dw_gen import Slots, Prog, ROOT, load_vec, naf_digits, param

SLOTS = Slots()
CONSTS = ["ZERO", "ONE", "M1", "TWO", "M2", "THREE", "FOUR", "M8", "BETA", "NBETA", "NAX", "NAY", "BNAX", "BNAY"]
E2 = [None] + [("E2_%d.x" % i, "E2_%d.y" % i, "E2_%d.by" % i) for i in range(1, 6)]          # X^(q^2) = e2 X: e2^i, beta e2^i.y
GM = [None] + [("G_%d.x" % i, "G_%d.y" % i, "G_%d.nby" % i, "G_%d.nx" % i) for i in range(1, 6)]   # gamma^i: x, y, -beta y, -x
QD = ["QA.x", "QA.y", "QA.by", "QAN.x", "QAN.y", "QAN.by", "QB.x", "QB.y", "QB.by", "QBN.x", "QBN.y", "QBN.by"]   # Qx', Qx' xi, Qy', Qy' xi
POINT = ["X", "Y", "Z", "nZ", "ZZ", "ZZZ", "Px", "Py", "nPy", "L.c"]
LINE = ["FA.x", "FA.y", "FA.by", "FAN.x", "FAN.y", "FAN.by", "FB.x", "FB.y", "FB.by", "FBN.x", "FBN.y", "FBN.by"]
REGS = ["F", "FX", "FX2", "FX3", "T0", "T1", "Y", "U"]
SCAL = ["nrm", "ninv"]


def reg(r):
    """the twelve slots of an F_q^12 register: [(x, y)] * 6"""
    return [("%s.%d.x" % (r, i), "%s.%d.y" % (r, i)) for i in range(6)]


# ---------------------------------------------------------------------------------------------------------------------
# programs
# ---------------------------------------------------------------------------------------------------------------------
def g2_terms(ax, ay, bx, by, bby):
    """(a.x + a.y s)(b.x + b.y s): terms of the real and of the imaginary part; bby = beta b.y"""
    return [(ax, bx), (ay, bby)], [(ax, by), (ay, bx)]


def prog_sqr(dst, a):
    p = Prog("sqr_%s_%s" % (dst, a), SLOTS, "ft")
    emit_sqr(p, dst, a)
    return p.finish()


def emit_sqr(p, dst, a):
    A, D = reg(a), reg(dst)
    Bj = [p.mul(A[j][1], "BETA") for j in range(6)]
    W = {}
    for j in (3, 4, 5):
        W[j] = (p.sop([(A[j][0], "NAX"), (A[j][1], "BNAY")]), p.sop([(A[j][0], "NAY"), (A[j][1], "NAX")]), p.sop([(A[j][0], "BNAY"), (A[j][1], "BNAX")]))
    Dbl = [(p.mul(A[i][0], "TWO"), p.mul(A[i][1], "TWO")) for i in range(5)]
    for k in range(6):
        tx, ty = [], []
        for i in range(6):
            for j in range(i, 6):
                if (i + j) % 6 != k:
                    continue
                sx, sy, sby = W[j] if i + j >= 6 else (A[j][0], A[j][1], Bj[j])
                fx, fy = (A[i][0], A[i][1]) if i == j else Dbl[i]
                rx, ry = g2_terms(fx, fy, sx, sy, sby)
                tx += rx
                ty += ry
        p.sop(tx, out=D[k][0], floor=2)
        p.sop(ty, out=D[k][1], floor=2)


def prog_mul(dst, a, b):
    p = Prog("mul_%s_%s_%s" % (dst, a, b), SLOTS, "ft")
    A, B, D = reg(a), reg(b), reg(dst)
    Bj = [p.mul(B[j][1], "BETA") for j in range(6)]
    W = {}
    for j in range(1, 6):
        W[j] = (p.sop([(B[j][0], "NAX"), (B[j][1], "BNAY")]), p.sop([(B[j][0], "NAY"), (B[j][1], "NAX")]), p.sop([(B[j][0], "BNAY"), (B[j][1], "BNAX")]))
    for k in range(6):
        halves = [([], []), ([], [])]
        for i in range(6):
            j = (k - i) % 6
            sx, sy, sby = W[j] if i + j >= 6 else (B[j][0], B[j][1], Bj[j])
            rx, ry = g2_terms(A[i][0], A[i][1], sx, sy, sby)
            halves[i // 3][0].extend(rx)
            halves[i // 3][1].extend(ry)
        hx = p.sop(halves[0][0], floor=2)
        hy = p.sop(halves[0][1], floor=2)
        p.sop([(hx, "ONE")] + halves[1][0], out=D[k][0], floor=3)
        p.sop([(hy, "ONE")] + halves[1][1], out=D[k][1], floor=3)
    return p.finish()


def prog_line_mul():
    """F <- F * (c + fa X^4 + fb X^3): out_i = c v_i + fa v_{i-4} + fb v_{i-3}, a factor xi on wrap (folded into FAN / FBN)"""
    p = Prog("line_mul", SLOTS, "ft")
    V = reg("F")
    for i in range(6):
        j, k = (i + 2) % 6, (i + 3) % 6
        fa = ("FAN.x", "FAN.y", "FAN.by") if i + 2 < 6 else ("FA.x", "FA.y", "FA.by")
        fb = ("FBN.x", "FBN.y", "FBN.by") if i + 3 < 6 else ("FB.x", "FB.y", "FB.by")
        p.sop([("L.c", V[i][0]), (fa[0], V[j][0]), (fa[2], V[j][1]), (fb[0], V[k][0]), (fb[2], V[k][1])], out=V[i][0])
        p.sop([("L.c", V[i][1]), (fa[0], V[j][1]), (fa[1], V[j][0]), (fb[0], V[k][1]), (fb[1], V[k][0])], out=V[i][1])
    return p.finish()


def line_values(p, la, lb):
    """a Qx', a Qx' xi, b Qy', b Qy' xi (x, y, beta y each) from the line's coefficients"""
    for pre, q, l in (("FA", "QA", la), ("FAN", "QAN", la), ("FB", "QB", lb), ("FBN", "QBN", lb)):
        for c in ("x", "y", "by"):
            p.mul(l, "%s.%s" % (q, c), out="%s.%s" % (pre, c))


def prog_point_dbl():
    """V <- 2V on y^2 = x^3 + b and the tangent (do_tangent, f_param.c:171-184, scaled by -Z^6 as tools/dw_gen.py's):
         M = 3X^2;  a' = M Z^2, b' = -(2YZ) Z^2, c' = 2Y^2 - M X;  X3 = M^2 - 8XY^2, Y3 = M (4XY^2 - X3) - 8Y^4, Z3 = 2YZ"""
    p = Prog("pt_dbl", SLOTS, "pt")
    emit_point_dbl(p)
    return p.finish()


def prog_sqr_dbl():
    """F <- F^2 and V <- 2V side by side (they share nothing): the Miller loop's square with the NEXT step's doubling -- the first
    two levels of the doubling (four sums each) ride in the square's two levels, four levels instead of six"""
    p = Prog("sqrdbl", SLOTS, "ft")
    emit_sqr(p, "F", "F")
    emit_point_dbl(p)
    return p.finish()


def emit_point_dbl(p):
    X, Y, Z, nZ = "X", "Y", "Z", "nZ"
    XX = p.mul(X, X)
    YY = p.mul(Y, Y)
    Z3 = p.sop([(Y, Z), (Y, Z)], out="Z")
    nZ3 = p.sop([(Y, nZ), (Y, nZ)], out="nZ")
    M = p.mul(XX, "THREE")
    S1 = p.mul(X, YY)
    Y4 = p.mul(YY, YY)
    lb = p.mul(nZ3, "ZZ")
    ZZn = p.mul(Z3, Z3, out="ZZ")
    X3 = p.sop([(M, M), (S1, "M8")], out="X")
    la = p.mul(M, "ZZ")
    MX = p.mul(M, X)
    nM = p.mul(M, "M1")
    S4 = p.mul(S1, "FOUR")
    p.mul(ZZn, Z3, out="ZZZ")
    p.sop([(M, S4), (nM, X3), (Y4, "M8")], out="Y")
    p.sop([(YY, "TWO"), (MX, "M1")], out="L.c")
    line_values(p, la, lb)


def prog_point_add(neg):
    """V <- V +- P and the chord (do_line, f_param.c:190-199, scaled by -Z3): H = Px Z^2 - X, R = Py' Z^3 - Y;
         a' = R, b' = -Z3, c' = Z3 Py' - R Px, Z3 = Z H"""
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
    p.sop([(Z3, Py), (nR, "Px")], out="L.c")
    HHH = p.mul(HH, H)
    XHH = p.mul(X, HH)
    ZZn = p.mul(Z3, Z3, out="ZZ")
    X3 = p.sop([(R, R), (HHH, "M1"), (XHH, "M2")], out="X")
    RX = p.mul(R, XHH)
    nYH = p.mul(nY, HHH)
    p.mul(ZZn, Z3, out="ZZZ")
    p.sop([(RX, "ONE"), (nR, X3), (nYH, "ONE")], out="Y")
    line_values(p, R, nZ3)
    return p.finish()


def prog_qp2(dst, a):
    """coefficient-wise q^2-power Frobenius (qpower, f_param.c:257-268): X^(q^2) = e2 X"""
    p = Prog("qp2_%s_%s" % (dst, a), SLOTS, "ft")
    A, D = reg(a), reg(dst)
    p.mul(A[0][0], "ONE", out=D[0][0])
    p.mul(A[0][1], "ONE", out=D[0][1])
    for i in range(1, 6):
        ex, ey, eby = E2[i]
        p.sop([(A[i][0], ex), (A[i][1], eby)], out=D[i][0])
        p.sop([(A[i][0], ey), (A[i][1], ex)], out=D[i][1])
    return p.finish()


def prog_frob(dst, a):
    """a^q: conjugate the F_q^2 coefficients, scale by gamma^i (X^q = gamma X)"""
    p = Prog("frob_%s_%s" % (dst, a), SLOTS, "ft")
    A, D = reg(a), reg(dst)
    p.mul(A[0][0], "ONE", out=D[0][0])
    p.mul(A[0][1], "M1", out=D[0][1])
    for i in range(1, 6):
        gx, gy, gnby, gnx = GM[i]
        p.sop([(A[i][0], gx), (A[i][1], gnby)], out=D[i][0])          # (x - y s)(gx + gy s) = x gx - beta y gy + (x gy - y gx) s
        p.sop([(A[i][0], gy), (A[i][1], gnx)], out=D[i][1])
    return p.finish()


def prog_conj(dst, a):
    """the q^6-power Frobenius: negate the odd coefficients (pairing_f.cuh f12_conj)"""
    p = Prog("conj_%s_%s" % (dst, a), SLOTS, "ft")
    A, D = reg(a), reg(dst)
    for i in range(6):
        for c in (0, 1):
            p.mul(A[i][c], "M1" if i & 1 else "ONE", out=D[i][c])
    return p.finish()


def prog_copy(dst, a):
    p = Prog("copy_%s_%s" % (dst, a), SLOTS, "ft")
    A, D = reg(a), reg(dst)
    for i in range(6):
        for c in (0, 1):
            p.mul(A[i][c], "ONE", out=D[i][c])
    return p.finish()


def prog_norm(n):
    """N(n_0) = n_0.x^2 - beta n_0.y^2 of the F_q^2 coefficient 0 of register n -> slot nrm"""
    p = Prog("norm_%s" % n, SLOTS, "ft")
    N = reg(n)
    nby = p.mul(N[0][1], "NBETA")
    p.sop([(N[0][0], N[0][0]), (N[0][1], nby)], out="nrm")
    return p.finish()


def prog_scale_inv(dst, t, n):
    """dst_i = t_i / n_0 with ninv = 1 / N(n_0):  1 / n_0 = (n_0.x - n_0.y s) ninv"""
    p = Prog("scinv_%s_%s_%s" % (dst, t, n), SLOTS, "ft")
    T, N, D = reg(t), reg(n), reg(dst)
    ix = p.mul(N[0][0], "ninv")
    nn = p.mul("ninv", "M1")
    iy = p.mul(N[0][1], nn)
    iby = p.mul(iy, "BETA")
    for i in range(6):
        p.sop([(T[i][0], ix), (T[i][1], iby)], out=D[i][0])
        p.sop([(T[i][0], iy), (T[i][1], ix)], out=D[i][1])
    return p.finish()


MAKERS = {"sqr": prog_sqr, "mul": prog_mul, "qp2": prog_qp2, "frob": prog_frob, "conj": prog_conj, "copy": prog_copy, "norm": prog_norm,
          "scinv": prog_scale_inv}


# ---------------------------------------------------------------------------------------------------------------------
# the pairing as a sequence of program names (fw_sched.h is this function)
# ---------------------------------------------------------------------------------------------------------------------
def miller_sequence(plus, minus, rbits):
    seq = []
    dig = lambda m: ((plus >> m) & 1) - ((minus >> m) & 1)
    for m in range(rbits - 2, -1, -1):
        seq += ["pt_dbl", "line_mul"]
        if m > 0 and dig(m):
            seq += ["pt_addm" if dig(m) < 0 else "pt_addp", "line_mul"]
        if m > 0:
            seq.append("sqr_F_F")
    out = []                                         # a square and the doubling after it: one program
    for n in seq:
        if n == "pt_dbl" and out and out[-1] == "sqr_F_F":
            out[-1] = "sqrdbl"
        else:
            out.append(n)
    return out


def pow_x(dst, a, x, xneg):
    """dst <- a^x by square-and-multiply on dst (f12_pow_x), conjugated for a negative x"""
    seq = ["copy_%s_%s" % (dst, a)]
    for bit in bin(x)[3:]:
        seq.append("sqr_%s_%s" % (dst, dst))
        if bit == "1":
            seq.append("mul_%s_%s_%s" % (dst, dst, a))
    if xneg:
        seq.append("conj_%s_%s" % (dst, dst))
    return seq


def final_sequence(x, xneg):
    """f_tateexp (f_param.c:250-283) as pairing_f.cuh f_final_exp / f_hard_bn run it; registers F (in and out), FX, FX2, FX3, T0,
    T1, Y, U"""
    s = []
    # easy part: F <- F^(q^8) F^(q^6) / (F^(q^2) F)
    s += ["qp2_Y_F", "qp2_Y_Y", "qp2_Y_Y", "qp2_Y_Y", "conj_U_F", "mul_Y_Y_U"]          # Y = F^(q^8) conj(F)
    s += ["qp2_U_F", "mul_U_U_F"]                                                          # U = F^(q^2) F
    # 1 / U (polymod_invert, poly.c:521-536): T0 = prod_{i=1..5} sigma^i(U), N = U T0 in F_q^2
    s += ["qp2_T1_U", "copy_T0_T1"]
    for _ in range(4):
        s += ["qp2_T1_T1", "mul_T0_T0_T1"]
    s += ["mul_T1_U_T0", "norm_T1", "OP_INV", "scinv_U_T0_T1"]
    s += ["mul_F_Y_U"]
    # hard part (f_hard_bn)
    s += pow_x("FX", "F", x, xneg) + pow_x("FX2", "FX", x, xneg) + pow_x("FX3", "FX2", x, xneg)
    s += ["frob_U_FX3", "mul_Y_U_FX3", "conj_Y_Y", "sqr_T0_Y"]                             # y6; T0 = y6^2
    s += ["frob_U_FX2", "mul_Y_U_FX", "conj_Y_Y", "mul_T0_T0_Y"]                           # y4
    s += ["conj_Y_FX2", "mul_T0_T0_Y"]                                                     # y5
    s += ["frob_U_FX", "conj_U_U", "mul_T1_U_Y", "mul_T1_T1_T0"]                           # y3
    s += ["qp2_Y_FX2", "mul_T0_T0_Y", "sqr_T1_T1", "mul_T1_T1_T0", "sqr_T1_T1"]            # y2
    s += ["conj_Y_F", "mul_T0_T1_Y"]                                                       # y1
    s += ["frob_U_F", "qp2_Y_F", "mul_FX_U_Y", "frob_U_Y", "mul_FX_FX_U", "mul_T1_T1_FX"]  # y0
    s += ["sqr_T0_T0", "mul_F_T0_T1"]
    return s


def bn_x(q):
    x = int(round((q / 36) ** 0.25))
    for d in range(-64, 65):
        for sgn in (1, -1):
            y = sgn * (x + d)
            if 36 * y ** 4 + 36 * y ** 3 + 24 * y ** 2 + 6 * y + 1 == q:
                return y
    return None


def build(pname="f"):
    for c in CONSTS:
        SLOTS.add(c)
    for t in E2[1:] + GM[1:]:
        for n in t:
            SLOTS.add(n)
    for n in QD + POINT + LINE + SCAL:
        SLOTS.add(n)
    for r in REGS:
        for x, y in reg(r):
            SLOTS.add(x)
            SLOTS.add(y)
    P = param(pname)
    x = bn_x(P["q"])
    names = ["pt_dbl", "pt_addp", "pt_addm", "line_mul", "mul_F_F_U", "sqrdbl", "sqr_F_F"]
    plus, minus, rbits = naf_digits(P["r"])
    for n in miller_sequence(plus, minus, rbits) + final_sequence(abs(x), x < 0) + final_sequence(abs(x), not (x < 0)):
        if n not in names and n != "OP_INV":
            names.append(n)
    progs = {}
    for n in names:
        if n == "pt_dbl":
            progs[n] = prog_point_dbl()
        elif n == "sqrdbl":
            progs[n] = prog_sqr_dbl()
        elif n in ("pt_addp", "pt_addm"):
            progs[n] = prog_point_add(n == "pt_addm")
        elif n == "line_mul":
            progs[n] = prog_line_mul()
        else:
            kind, *regs = n.split("_")
            progs[n] = MAKERS[kind](*regs)
    return progs


# ---------------------------------------------------------------------------------------------------------------------
# the model
# ---------------------------------------------------------------------------------------------------------------------
class Model:
    def __init__(self, pname, progs):
        P = param(pname)
        self.P, self.q, self.progs = P, P["q"], progs
        q, beta = self.q, P["beta"]
        self.fb = (q.bit_length() + 7) // 8
        self.env = {n: 0 for n in SLOTS.order}
        e = self.env
        na = ((-P["alpha0"]) % q, (-P["alpha1"]) % q)
        self.na, self.nai = na, self.inv2(na)
        e.update(ZERO=0, ONE=1, M1=q - 1, TWO=2, M2=q - 2, THREE=3, FOUR=4, M8=q - 8, BETA=beta, NBETA=q - beta, NAX=na[0], NAY=na[1],
                 BNAX=beta * na[0] % q, BNAY=beta * na[1] % q)
        e2, gam = self.pow2(na, (q * q - 1) // 6), self.pow2(na, (q - 1) // 6)
        ep, gp = e2, gam
        for i in range(1, 6):
            e[E2[i][0]], e[E2[i][1]], e[E2[i][2]] = ep[0], ep[1], beta * ep[1] % q
            e[GM[i][0]], e[GM[i][1]], e[GM[i][2]], e[GM[i][3]] = gp[0], gp[1], (-beta * gp[1]) % q, (-gp[0]) % q
            ep, gp = self.m2(ep, e2), self.m2(gp, gam)
        self.plus, self.minus, self.rbits = naf_digits(P["r"])
        self.x = bn_x(q)
        self.levels = 0
        self.flat = []

    def m2(self, a, b):
        q, beta = self.q, self.P["beta"]
        return ((a[0] * b[0] + beta * a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)

    def inv2(self, a):
        q = self.q
        n = pow((a[0] * a[0] - self.P["beta"] * a[1] * a[1]) % q, -1, q)
        return (a[0] * n % q, (-a[1] * n) % q)

    def pow2(self, a, e):
        r = (1, 0)
        for bit in bin(e)[2:]:
            r = self.m2(r, r)
            if bit == "1":
                r = self.m2(r, a)
        return r

    def run(self, name):
        if name == "OP_INV":
            self.env["ninv"] = pow(self.env["nrm"], -1, self.q) if self.env["nrm"] else 0
            self.flat.append(("op", "inv"))
            return
        p = self.progs[name]
        for lev, row in enumerate(p.levels):
            T = max(len(n.terms) for n in row)
            assert len(row) <= (16 if T > 4 else 31), (name, lev, len(row), T)
            writes = []
            for n in row:
                acc = 0
                for a, b in n.terms:
                    acc += self.env[p.ref(a)] * self.env[p.ref(b)]
                writes.append((n.slot, acc % self.q))
            assert len({w[0] for w in writes}) == len(writes), (name, lev)
            for s, v in writes:
                self.env[s] = v
            self.levels += 1
            self.flat.append(("level", name, lev))

    def set_inputs(self, g1, g2):
        q, e, fb, beta = self.q, self.env, self.fb, self.P["beta"]
        gi = lambda b, i: int.from_bytes(b[fb * i:fb * (i + 1)], "big") % q
        Px, Py = gi(g1, 0), gi(g1, 1)
        Qx, Qy = (gi(g2, 0), gi(g2, 1)), (gi(g2, 2), gi(g2, 3))
        ok = (Px * Px * Px + self.P["b"] - Py * Py) % q == 0
        tb = self.m2(((-self.P["alpha0"]) % q, (-self.P["alpha1"]) % q), (self.P["b"], 0))       # y^2 = x^3 - alpha b
        x3 = self.m2(self.m2(Qx, Qx), Qx)
        ok = ok and ((x3[0] + tb[0]) % q, (x3[1] + tb[1]) % q) == self.m2(Qy, Qy)
        e.update(X=Px, Y=Py, Z=1, nZ=q - 1, ZZ=1, ZZZ=1, Px=Px, Py=Py, nPy=(q - Py) % q)
        qa, qb = self.m2(Qx, self.nai), self.m2(Qy, self.nai)                                    # the untwisting map (f_pairing, :296-303)
        for pre, v in (("QA", qa), ("QAN", Qx), ("QB", qb), ("QBN", Qy)):
            e[pre + ".x"], e[pre + ".y"], e[pre + ".by"] = v[0], v[1], beta * v[1] % q
        for i, (x, y) in enumerate(reg("F")):
            e[x], e[y] = (1 if i == 0 else 0), 0
        return ok

    def miller(self, g1, g2):
        ok = self.set_inputs(g1, g2)
        self.flat = []
        for n in miller_sequence(self.plus, self.minus, self.rbits):
            self.run(n)
        return ok

    def finish(self):
        for n in final_sequence(abs(self.x), self.x < 0):
            self.run(n)
        self.flat.append(("op", "end"))
        return [self.env[c] for x, y in reg("F") for c in (x, y)]

    def pairing(self, g1, g2):
        ok = self.miller(g1, g2)
        r = self.finish()
        return r if ok else None

    def product(self, terms):
        """element_prod_pairing (type f installs no product routine: generic_prod_pairings, ecc/pairing.c:35-46, multiplies k reduced
        pairings; the reduced product of the Miller values is the same element): one wavefront per TERM for the Miller values,
        then one per product -- F <- the first record, U <- each further one and mul_F_F_U, ONE final exponentiation"""
        vals, ok = [], True
        for g1, g2 in terms:
            ok = self.miller(g1, g2) and ok
            vals.append([self.env[c] for x, y in reg("F") for c in (x, y)])
        self.flat = []
        names = [c for x, y in reg("F") for c in (x, y)]
        for n, v in zip(names, vals[0]):
            self.env[n] = v
        for val in vals[1:]:
            for n, v in zip([c for x, y in reg("U") for c in (x, y)], val):
                self.env[n] = v
            self.run("mul_F_F_U")
        r = self.finish()
        return r if ok else None


def check(progs, count=4):
    M = Model("f", progs)
    bad = 0
    ident = [1] + [0] * 11
    for name in ("f_rand16.vec", "f_edge10.vec"):
        g1, g2, gt = load_vec(os.path.join(ROOT, "tests", "golden", name))
        for i in range(min(count, len(gt))):
            M.levels = 0
            r = M.pairing(g1[i], g2[i])
            want = [int.from_bytes(gt[i][20 * c:20 * c + 20], "big") for c in range(12)]
            if (r or ident) != want:
                bad += 1
                print("MISMATCH", name, i)
    levels = M.levels
    for name in ("f_prod3x5_edge.vec", "f_prod4x3.vec"):
        g1, g2, gt = load_vec(os.path.join(ROOT, "tests", "golden", name))
        k = len(g1) // len(gt)
        for i in range(len(gt)):
            r = M.product([(g1[i * k + t], g2[i * k + t]) for t in range(k)])
            if (r or ident) != [int.from_bytes(gt[i][20 * c:20 * c + 20], "big") for c in range(12)]:
                bad += 1
                print("MISMATCH (product)", name, i)
    return bad, levels


def tables(progs):
    """rows, level index and program index in emission order"""
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


def flat_schedule(kind="pairing", pname="f"):
    """the packed schedules as the model executes them (fw_sched.h must build the same): "pairing"; "miller" (a term of a product);
    "finish" (the product with a term's value, then the final exponentiation)"""
    progs = _PROGS if _PROGS is not None else build(pname)
    rows, index, pidx = tables(progs)
    first = {name: f for name, f, c in pidx}
    M = Model(pname, progs)
    g1, g2, gt = load_vec(os.path.join(ROOT, "tests", "golden", "f_rand16.vec"))
    if kind == "pairing":
        M.pairing(g1[0], g2[0])
    elif kind == "miller":
        M.miller(g1[0], g2[0])
        M.flat.append(("op", "end"))
    else:
        M.product([(g1[0], g2[0]), (g1[1], g2[1])])
    out = []
    for e in M.flat:
        if e[0] == "level":
            r, T, lanes = index[first[e[1]] + e[2]]
            out.append(r | lanes << 12 | T << 34)
        else:
            out.append({"inv": 2, "end": 3}[e[1]] << 38)
    return out


def emit(progs):
    rows, index, pidx = tables(progs)
    keep = lambda n: ".t" not in n or not n.split(".")[0] in ("ft", "pt")
    out = ["// fw_tables.h -- GENERATED by tools/fw_gen.py (do not edit): the level programs of the wave-per-pairing type f kernel",
           "// (pairing_fw.cuh) for the machine of pairing_dw.cuh.  One ROW of five dwords per sum: the slot it writes, eight x operands, eight",
           "// y operands, one byte each (terms a sum does not have name the ZERO slot); a LEVEL is (first row, terms per sum, working lanes);",
           "// a PROGRAM is a run of levels, found by name by the host (fw_sched.h).",
           "#pragma once", "#include <stdint.h>", "namespace pbc { namespace fw {",
           "constexpr int kSlots = %d;" % len(SLOTS.order)]
    out.append("enum Slot : int { " + ", ".join("S_%s = %d" % (n.replace(".", "_"), i) for i, n in enumerate(SLOTS.order) if keep(n)) + " };")
    out.append("enum { OP_LEVEL = 0, OP_INV = 2, OP_END = 3 };      // schedule entries (fw_sched.h; the values of dw_tables.h)")
    out.append("struct LevelRef { uint16_t row; uint8_t T, lanes; };")
    out.append("struct ProgRef { const char *name; uint16_t first, count; };")
    out.append("constexpr int kProgs = %d, kLevels = %d, kRows = %d;" % (len(pidx), len(index), len(rows)))
    out.append("static const ProgRef h_prog[kProgs] = {" + ", ".join('{"%s", %d, %d}' % x for x in pidx) + "};")
    out.append("static const LevelRef h_level[kLevels] = {" + ", ".join("{%d, %d, %d}" % x for x in index) + "};")
    out.append("__device__ const uint32_t g_rows[kRows * 5] = {" + ",".join("0x%xu" % w for r in rows for w in r) + "};")
    out.append("} }  // namespace pbc::fw")
    return "\n".join(out) + "\n"


_PROGS = None


def main():
    global _PROGS
    progs = build()
    _PROGS = progs
    print("slots", len(SLOTS.order), "programs", len(progs), "rows", sum(len(r) for p in progs.values() for r in p.levels))
    assert len(SLOTS.order) <= 255, len(SLOTS.order)
    bad, levels = check(progs)
    print("levels per pairing %d; vectors: %s" % (levels, "MISMATCH" if bad else "ok"))
    if bad:
        sys.exit(1)
    text = emit(progs)
    path = os.path.join(ROOT, "pbc_amd", "csrc", "fw_tables.h")
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(path) and open(path).read() == text else 1)
    open(path, "w").write(text)
    print("wrote", path, len(text), "bytes")


if __name__ == "__main__":
    main()
