#!/usr/bin/env python3
"""Generator of the LEVEL PROGRAMS of the wave-per-pairing type d kernel (pbc_amd/csrc/pairing_dw.cuh, round 6).

One d159 pairing as a lane's serial instruction stream takes 3.9 ms whatever the batch size (2.0 M vector instructions);
every tower operation of pairing_d.cuh, though, is a set of independent lazily reduced sums of F_q products.  The wave
kernel runs ONE pairing per wavefront: every F_q element is a 6-limb slot of an LDS slot file, a LEVEL is "lane l computes
out[l] = sum_t x[l][t] * y[l][t] / R (Montgomery) from the slots its table row names and writes its slot", and a PROGRAM
is a list of levels.  This script
  * describes the operations of the pairing (Miller steps on E(F_q) with the line's evaluation, F_q^6 = F_q^3[sqrt v]
    products and squares, the pieces of cc_tatepower) as sums of products over named slots -- sums, differences and small
    multiples are products with the constants 1, -1, 2, ... so that NOTHING but sums of products exists;
  * levels them (a node's level = 1 + the deepest producer among its operands), assigns lanes and slots (a level's lanes read
    all their operands before any lane writes: a result may overwrite a slot whose last reader sits in the same level);
  * runs the emitted tables on Python integers -- the same driver sequence as the kernel -- and compares the pairing with
    the reference's vectors (tests/golden/d_*.vec): the tables are checked without a GPU;
  * writes pbc_amd/csrc/dw_tables.h.

The formulas are those of pairing_d.cuh (d_dbl_core, d_add_core, d_evalfn_pack, f6l_mul, f6l_sqr, d_final_exp), i.e. of
ecc/d_param.c:321-422, :441-564 up to factors of F_q^* in the lines, which the final exponentiation removes.

  python tools/dw_gen.py            check against the vectors and (re)write pbc_amd/csrc/dw_tables.h
  python tools/dw_gen.py --check    exit 1 if the committed header differs from what the script would write
"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TMAX = 8                     # terms per sum: (8 + 1) * 6 * 2^58 < 2^64 (fp.cuh sop_limbs)
LANES = 64


# ---------------------------------------------------------------------------------------------------------------------
# programs
# ---------------------------------------------------------------------------------------------------------------------
class Slots:
    """the slot file: named state / constant slots first, the temporaries of each program region behind them"""

    def __init__(self):
        self.index = {}
        self.order = []

    def add(self, name):
        if name not in self.index:
            self.index[name] = len(self.order)
            self.order.append(name)
        return name

    def __getitem__(self, name):
        return self.index[name]


class Node:
    def __init__(self, terms, out, floor=1):
        self.terms, self.out, self.level, self.lane, self.slot, self.floor = terms, out, 0, 0, None, floor


class Prog:
    """a list of sums of products; operands are slot names (state, constants: the value BEFORE the program unless a node of
    this program has written the slot at a lower level) or nodes of this program"""

    def __init__(self, name, slots, region):
        self.name, self.slots, self.region, self.nodes = name, slots, region, []

    def sop(self, terms, out=None, floor=1):
        assert 1 <= len(terms) <= TMAX, (self.name, len(terms))
        n = Node(list(terms), out, floor)
        self.nodes.append(n)
        return n

    # -- helpers: everything is a sum of products --
    def mul(self, a, b, out=None):
        return self.sop([(a, b)], out)

    def lin(self, pairs, out=None):
        """sum of c_i * a_i for constant slot names c_i"""
        return self.sop([(a, c) for a, c in pairs], out)

    def finish(self):
        slots = self.slots
        # levels; a node that overwrites a state slot waits for the last reader of the slot's old value (same level is fine:
        # a level's lanes read before any lane writes)
        writers = {}
        for n in self.nodes:
            if n.out is not None:
                assert n.out not in writers, (self.name, n.out)
                writers[n.out] = n
        floor = {n: n.floor for n in self.nodes}
        for _ in range(64):
            for n in self.nodes:
                n.level = max([floor[n]] + [1 + op.level for t in n.terms for op in t if isinstance(op, Node)])
            moved = False
            for n in self.nodes:
                for t in n.terms:
                    for op in t:
                        if isinstance(op, str) and op in writers and n.level > writers[op].level:
                            floor[writers[op]] = n.level
                            moved = True
            if not moved:
                break
        else:
            raise AssertionError((self.name, "no consistent levels"))
        nlev = max(n.level for n in self.nodes)
        for n in self.nodes:
            for t in n.terms:
                for op in t:
                    if isinstance(op, str) and op in writers:
                        assert n.level <= writers[op].level, (self.name, "old value of", op, "read above its writer")
        # temporaries: linear scan over levels; a slot is free again for writers at levels >= its last reader's level
        last_read = {}
        for n in self.nodes:
            for t in n.terms:
                for op in t:
                    if isinstance(op, Node):
                        last_read[op] = max(last_read.get(op, 0), n.level)
        free_at = []                       # (level from which the slot may be written again, slot name)
        ntemp = 0
        for lev in range(1, nlev + 1):
            for n in [m for m in self.nodes if m.level == lev]:
                if n.out is not None:
                    n.slot = n.out
                    continue
                cand = [i for i, (fl, _) in enumerate(free_at) if fl <= lev]
                if cand:
                    _, name = free_at.pop(cand[0])
                else:
                    name = slots.add("%s.t%d" % (self.region, ntemp))
                    ntemp += 1
                n.slot = name
                free_at.append((last_read.get(n, lev) if n in last_read else lev, name))
        # lanes
        self.levels = []
        for lev in range(1, nlev + 1):
            row = [n for n in self.nodes if n.level == lev]
            assert len(row) <= LANES, (self.name, lev, len(row))
            for i, n in enumerate(row):
                n.lane = i
            self.levels.append(row)
        return self

    def ref(self, op):
        return op.slot if isinstance(op, Node) else op

    def table(self):
        """[(T, [(out, [x...], [y...]) per lane])] with slot indices; sums are padded with 0 * 0 (slot ZERO)"""
        out = []
        z = self.slots["ZERO"]
        for row in self.levels:
            T = max(len(n.terms) for n in row)
            T = 1 if T == 1 else 2 if T == 2 else 4 if T <= 4 else 8
            lanes = []
            for n in row:
                xs = [self.slots[self.ref(a)] for a, _ in n.terms] + [z] * (T - len(n.terms))
                ys = [self.slots[self.ref(b)] for _, b in n.terms] + [z] * (T - len(n.terms))
                lanes.append((self.slots[n.slot], xs, ys))
            out.append((T, lanes))
        return out


# ---------------------------------------------------------------------------------------------------------------------
# the tower on slots
# ---------------------------------------------------------------------------------------------------------------------
def f3_mul_terms(a, b, k):
    """terms of coefficient k of the plain product of two degree-2 polynomials"""
    return [(a[i], b[k - i]) for i in range(3) if 0 <= k - i <= 2]


def f3_mul2(p, A1, B1, A2, B2, outs=(None, None, None), extra=((), (), ())):
    """A1 B1 + A2 B2 in F_q^3 (pairing_d.cuh f3l_mul2): the high coefficients first, then every output one lazy sum over
    both products and the table terms x^3 = sum XP3[k] x^k, x^4 = sum XP4[k] x^k; extra[k]: further terms of output k"""
    H0 = p.sop(f3_mul_terms(A1, B1, 3) + (f3_mul_terms(A2, B2, 3) if A2 else []))
    H1 = p.sop(f3_mul_terms(A1, B1, 4) + (f3_mul_terms(A2, B2, 4) if A2 else []))
    r = []
    for k in range(3):
        t = f3_mul_terms(A1, B1, k) + (f3_mul_terms(A2, B2, k) if A2 else []) + [(H0, "XP3_%d" % k), (H1, "XP4_%d" % k)] + list(extra[k])
        r.append(p.sop(t, outs[k]))
    return r


def f3_mul(p, A, B, outs=(None, None, None), extra=((), (), ())):
    return f3_mul2(p, A, B, None, None, outs, extra)


def f3_scale(p, A, c):
    return [p.mul(a, c) for a in A]


def poly_coeff(A, B, k):
    return f3_mul_terms(A, B, k)


def f6_mul(p, a, b, outs=None):
    """(ax + ay s)(bx + by s) = (ax bx + v ay by) + (ay bx + ax by) s, s^2 = v, in TWO levels: first the five coefficients
    E_k of the plain product ay by and the high coefficients of ax bx and of ay bx + ax by; then every output one sum of at
    most eight terms -- v E_k, and E_3, E_4 through the table constants v x^3, v x^4 (VXP3, VXP4)"""
    ax, ay = a
    bx, by = b
    E = [p.sop(poly_coeff(ay, by, k)) for k in range(5)]
    Hx = [p.sop(poly_coeff(ax, bx, k)) for k in (3, 4)]
    Hy = [p.sop(poly_coeff(ay, bx, k) + poly_coeff(ax, by, k)) for k in (3, 4)]
    ox, oy = [], []
    for k in range(3):
        ox.append(p.sop(poly_coeff(ax, bx, k) + [(Hx[0], "XP3_%d" % k), (Hx[1], "XP4_%d" % k), (E[k], "V"), (E[3], "VXP3_%d" % k), (E[4], "VXP4_%d" % k)],
                        outs[0][k] if outs else None))
        oy.append(p.sop(poly_coeff(ay, bx, k) + poly_coeff(ax, by, k) + [(Hy[0], "XP3_%d" % k), (Hy[1], "XP4_%d" % k)], outs[1][k] if outs else None))
    return ox, oy


def f6_sqr(p, a, outs=None):
    """(ax^2 + v ay^2) + 2 ax ay s in two levels (the doubled cross terms as repeated terms)"""
    ax, ay = a
    E = [p.sop(poly_coeff(ay, ay, k)) for k in range(5)]
    Hx = [p.sop(poly_coeff(ax, ax, k)) for k in (3, 4)]
    Hy = [p.sop(poly_coeff(ax, ay, k) + poly_coeff(ax, ay, k)) for k in (3, 4)]
    ox, oy = [], []
    for k in range(3):
        ox.append(p.sop(poly_coeff(ax, ax, k) + [(Hx[0], "XP3_%d" % k), (Hx[1], "XP4_%d" % k), (E[k], "V"), (E[3], "VXP3_%d" % k), (E[4], "VXP4_%d" % k)],
                        outs[0][k] if outs else None))
        oy.append(p.sop(poly_coeff(ax, ay, k) + poly_coeff(ax, ay, k) + [(Hy[0], "XP3_%d" % k), (Hy[1], "XP4_%d" % k)], outs[1][k] if outs else None))
    return ox, oy


def f3_frob(p, A, outs=(None, None, None)):
    """a^q = a0 + a1 x^q + a2 x^2q  (f3_frob)"""
    return [p.sop(([(A[0], "ONE")] if i == 0 else []) + [(A[1], "XQ1_%d" % i), (A[2], "XQ2_%d" % i)], outs[i]) for i in range(3)]


F = [["f.x0", "f.x1", "f.x2"], ["f.y0", "f.y1", "f.y2"]]
QX, QY = ["Qx0", "Qx1", "Qx2"], ["Qy0", "Qy1", "Qy2"]


def line_names(bank):
    """the line coefficients a', b', c' in F_q of a step (bank 0 / 1: the point track runs ahead of the accumulator)"""
    return ["L%d.a" % bank, "L%d.b" % bank, "L%d.c" % bank]


def value_names(bank):
    """the line's VALUE at Q, l = (a' Qx + c') + (b' Qy) s: six F_q slots"""
    return [["V%d.x%d" % (bank, i) for i in range(3)], ["V%d.y%d" % (bank, i) for i in range(3)]]


def eval_previous_line(p, bank):
    """d_evalfn_pack for the line whose coefficients the PREVIOUS point program left in the other bank: it rides in the first
    level of this one (the coefficients are complete, this program's own go to `bank` from its second level on)"""
    la, lb, lc = line_names(1 - bank)
    V = value_names(1 - bank)
    for i in range(3):
        p.sop([(la, QX[i])] + ([(lc, "ONE")] if i == 0 else []), out=V[0][i])
        p.sop([(lb, QY[i])], out=V[1][i], floor=2)      # (half of it one level later: an eight-term level deals four lanes to a sum)


def prog_point_dbl(slots, bank):
    """V <- 2V and the tangent's coefficients into bank `bank` (d_dbl_core, scaled by -1); state X, Y, Z, nZ = -Z,
    W = a Z^4 (modified Jacobian coordinates: four levels instead of five), and Z^2, Z^3 of the NEW point for a chord step:
         M = 3X^2 + W;  a' = M Z^2, b' = -(2YZ) Z^2, c' = 2Y^2 - M X;  X3 = M^2 - 8XY^2, Y3 = M (4XY^2 - X3) - 8Y^4,
         Z3 = 2YZ, W3 = 16 Y^4 W"""
    p = Prog("pt_dbl%d" % bank, slots, "pt")
    la, lb, lc = line_names(bank)
    X, Y, Z, nZ, W = "X", "Y", "Z", "nZ", "W"
    eval_previous_line(p, bank)
    XX = p.mul(X, X)
    YY = p.mul(Y, Y)
    ZZ = p.mul(Z, Z)
    Z3 = p.sop([(Y, Z), (Y, Z)], out="Z")
    nZ3 = p.sop([(Y, nZ), (Y, nZ)], out="nZ")
    W16 = p.mul(W, "SIXTEEN")
    M = p.sop([(XX, "THREE"), (W, "ONE")])
    S1 = p.mul(X, YY)
    Y4 = p.mul(YY, YY)
    p.mul(nZ3, ZZ, out=lb)
    ZZn = p.mul(Z3, Z3, out="ZZ")
    X3 = p.sop([(M, M), (S1, "M8")], out="X")
    p.mul(M, ZZ, out=la)
    MX = p.mul(M, X)
    nM = p.mul(M, "M1")
    S4 = p.mul(S1, "FOUR")
    p.mul(Y4, W16, out="W")
    p.mul(ZZn, Z3, out="ZZZ")
    p.sop([(M, S4), (nM, X3), (Y4, "M8")], out="Y")
    p.sop([(YY, "TWO"), (MX, "M1")], out=lc)
    return p.finish()


def prog_point_add(slots, bank, neg):
    """V <- V +- P and the chord's coefficients (d_add_core, scaled by -1), five levels on Z^2, Z^3 from the doubling before it:
         H = Px Z^2 - X, R = Py' Z^3 - Y;  a' = R, b' = -Z3, c' = Z3 Py' - R Px,  Z3 = Z H;  then W = a Z3^4"""
    p = Prog("pt_add%s%d" % ("m" if neg else "p", bank), slots, "pt")
    la, lb, lc = line_names(bank)
    X, Y, Z, nZ = "X", "Y", "Z", "nZ"
    Py = "nPy" if neg else "Py"
    eval_previous_line(p, bank)
    H = p.sop([("Px", "ZZ"), (X, "M1")])
    R = p.sop([(Py, "ZZZ"), (Y, "M1")])
    nY = p.mul(Y, "M1")
    Z3 = p.mul(Z, H, out="Z")
    nZ3 = p.mul(nZ, H, out="nZ")
    HH = p.mul(H, H)
    nR = p.mul(R, "M1")
    p.lin([(R, "ONE")], out=la)
    p.sop([(Z3, Py), (nR, "Px")], out=lc)
    p.lin([(nZ3, "ONE")], out=lb)
    HHH = p.mul(HH, H)
    XHH = p.mul(X, HH)
    ZZ3 = p.mul(Z3, Z3)
    X3 = p.sop([(R, R), (HHH, "M1"), (XHH, "M2")], out="X")
    RX = p.mul(R, XHH)
    nYH = p.mul(nY, HHH)
    Z43 = p.mul(ZZ3, ZZ3)
    p.sop([(RX, "ONE"), (nR, X3), (nYH, "ONE")], out="Y")
    p.mul(Z43, "A", out="W")
    return p.finish()


def prog_point_eval(slots, bank):
    """the value of the LAST line (no point program follows it); `bank`: the bank a following program would have used"""
    p = Prog("pt_eval%d" % bank, slots, "pt")
    eval_previous_line(p, bank)
    return p.finish()


def prog_f_mul_line(slots, bank):
    """f <- f * l, l's value in bank `bank`"""
    p = Prog("f_mul%d" % bank, slots, "ft")
    f6_mul(p, F, value_names(bank), outs=F)
    return p.finish()


def prog_f_sqr(slots):
    p = Prog("f_sqr", slots, "ft")
    f6_sqr(p, F, outs=F)
    return p.finish()


# -- final exponentiation (d_final_exp = cc_tatepower with one inversion) ------------------------------------------------
def names3(prefix):
    return [slots_add("%s%d" % (prefix, i)) for i in range(3)]


def slots_add(name):
    SLOTS.add(name)
    return name


def prog_fe1(slots):
    """u = conj(m)^2 = (a^2 + v b^2) - 2ab s, N = a^2 - v b^2;  w = u^q u (A + B s);  D = N^q N;  t = D Bn is formed by the
    driver after the zero test of B"""
    p = Prog("fe1", slots, "fe")
    a, b = F
    vb = f3_scale(p, b, "V")
    nb = f3_scale(p, b, "M1")
    nvb = f3_scale(p, vb, "M1")
    ux = f3_mul2(p, a, a, vb, b)                       # a^2 + v b^2
    N = f3_mul2(p, a, a, nvb, b)                       # a^2 - v b^2
    uy = f3_mul2(p, a, nb, a, nb)                      # -2ab
    uqx = f3_frob(p, ux)
    uqy_pos = f3_frob(p, uy)
    uqy = f3_scale(p, uqy_pos, "M1")                   # (x0 + x1 s)^q = x0^q - x1^q s
    f6_mul(p, (uqx, uqy), (ux, uy), outs=(names3("wA"), names3("wB")))
    Nq = f3_frob(p, N)
    f3_mul(p, Nq, N, outs=names3("D"))
    return p.finish()


def prog_fe2(slots):
    """t = D Bn, then the norm route of f3_inv: w3 = t^q t^(q^2), m = t w3 (only m0 is non-zero): the driver inverts m0"""
    p = Prog("fe2", slots, "fe")
    D, Bn = names3("D"), names3("Bn")
    t = f3_mul(p, D, Bn, outs=names3("tDB"))
    t1 = f3_frob(p, t)
    t2 = f3_frob(p, t1)
    w3 = f3_mul(p, t1, t2, outs=names3("w3"))
    f3_mul(p, t, w3, outs=names3("nrm"))
    return p.finish()


def prog_fe3(slots):
    """1/(D B) = w3 * (1/m0);  invD = that * Bn, invB = that * D;  h0 = A invD, P = 2 h0;  v0 = 2, v1 = P"""
    p = Prog("fe3", slots, "fe")
    inv = f3_scale(p, names3("w3"), "ninv")
    invD = f3_mul(p, inv, names3("Bn"))
    f3_mul(p, inv, names3("D"), outs=names3("invB"))
    h0 = f3_mul(p, names3("wA"), invD)
    for i in range(3):
        p.lin([(h0[i], "TWO")], out="P%d" % i)
        p.lin([(h0[i], "TWO")], out="v1_%d" % i)
    p.lin([("TWO", "ONE")], out="v0_0")
    p.lin([("ZERO", "ZERO")], out="v0_1")
    p.lin([("ZERO", "ZERO")], out="v0_2")
    return p.finish()


def prog_lucas(slots, bit):
    """one step of lucas_even (d_param.c:462-482):  mm = v0 v1 - P;  bit ? (v1 <- v1^2 - 2, v0 <- mm) : (v0 <- v0^2 - 2, v1 <- mm)"""
    p = Prog("lucas%d" % bit, slots, "fe")
    v0, v1, P = names3("v0_"), names3("v1_"), names3("P")
    sq, other = (v1, v0) if bit else (v0, v1)
    f3_mul(p, v0, v1, outs=other, extra=tuple([(P[k], "M1")] for k in range(3)))
    f3_mul(p, sq, sq, outs=sq, extra=([("M2", "ONE")], [], []))
    return p.finish()


def prog_fe4(slots):
    """out.x = V_k / 2;  out.y = (P V_k - 2 V_{k-1}) D invB / (4 v)   (v1 = V_k, v0 = V_{k-1})"""
    p = Prog("fe4", slots, "fe")
    v0, v1, P = names3("v0_"), names3("v1_"), names3("P")
    nv0 = f3_scale(p, v0, "M2")
    t = f3_mul(p, P, v1, extra=tuple([(nv0[k], "ONE")] for k in range(3)))
    t = f3_mul(p, t, names3("D"))
    t = f3_mul(p, t, names3("invB"))
    for i in range(3):
        p.sop([(t[i], "QVI")], out=F[1][i])            # / (4 v)
        p.sop([(v1[i], "HALF")], out=F[0][i])
    return p.finish()


SLOTS = Slots()
CONSTS = ["ZERO", "ONE", "M1", "TWO", "M2", "THREE", "FOUR", "M8", "SIXTEEN", "HALF", "QVI", "A", "V"] + \
         ["XP3_%d" % k for k in range(3)] + ["XP4_%d" % k for k in range(3)] + ["XQ1_%d" % k for k in range(3)] + ["XQ2_%d" % k for k in range(3)] + \
         ["VXP3_%d" % k for k in range(3)] + ["VXP4_%d" % k for k in range(3)]
STATE = ["X", "Y", "Z", "nZ", "W", "ZZ", "ZZZ", "Px", "Py", "nPy"] + QX + QY + F[0] + F[1] + line_names(0) + line_names(1) + \
        sum(value_names(0), []) + sum(value_names(1), []) + ["ninv"]


def build():
    for c in CONSTS + STATE:
        SLOTS.add(c)
    for pre in ("wA", "wB", "D", "Bn", "tDB", "w3", "nrm", "invB", "P", "v0_", "v1_"):
        names3(pre)
    progs = []
    for bank in (0, 1):
        progs += [prog_point_dbl(SLOTS, bank), prog_point_add(SLOTS, bank, False), prog_point_add(SLOTS, bank, True), prog_point_eval(SLOTS, bank),
                  prog_f_mul_line(SLOTS, bank)]
    progs += [prog_f_sqr(SLOTS), prog_fe1(SLOTS), prog_fe2(SLOTS), prog_fe3(SLOTS), prog_lucas(SLOTS, 0), prog_lucas(SLOTS, 1), prog_fe4(SLOTS)]
    return {p.name: p for p in progs}


# ---------------------------------------------------------------------------------------------------------------------
# the model: the tables on Python integers, driven like the kernel
# ---------------------------------------------------------------------------------------------------------------------
def param(name):
    d = {}
    for line in open(os.path.join(ROOT, "pbc_amd", "param", name + ".param")):
        f = line.split()
        if len(f) == 2:
            d[f[0]] = int(f[1]) if f[1].isdigit() else f[1]
    return d


def naf_digits(r):
    """hostbn.h naf_of_half: digit i of n = r >> 1 at position i + 1; returns (plus, minus, rbits)"""
    n, pos, top, plus, minus = r >> 1, 1, 0, 0, 0
    while n:
        if n & 1:
            if n & 3 == 1:
                plus |= 1 << pos
                n -= 1
            else:
                minus |= 1 << pos
                n += 1
            top = pos
        n >>= 1
        pos += 1
    return plus, minus, top + 1


class Model:
    def __init__(self, pname, progs):
        P = param(pname)
        self.q, self.P, self.progs = P["q"], P, progs
        self.fb = (P["q"].bit_length() + 7) // 8     # length_in_bytes of an F_q coordinate
        q = self.q
        c = [P["coeff0"], P["coeff1"], P["coeff2"]]
        xp3 = [(-x) % q for x in c]
        xp4 = [(xp3[2] * xp3[0]) % q] + [(xp3[i - 1] + xp3[2] * xp3[i]) % q for i in (1, 2)]
        self.xp = [xp3, xp4]
        v = P["nqr"]
        xq = self.f3_pow([0, 1, 0], q)
        xq2 = self.f3_mul_int(xq, xq)
        inv = lambda x: pow(x, -1, q)
        self.env = {n: 0 for n in SLOTS.order}
        e = self.env
        e.update(ZERO=0, ONE=1, M1=q - 1, TWO=2, M2=q - 2, THREE=3, FOUR=4, M8=q - 8, SIXTEEN=16, HALF=inv(2), QVI=inv(4 * v % q), A=P["a"], V=v)
        for k in range(3):
            e["XP3_%d" % k], e["XP4_%d" % k], e["XQ1_%d" % k], e["XQ2_%d" % k] = xp3[k], xp4[k], xq[k], xq2[k]
            e["VXP3_%d" % k], e["VXP4_%d" % k] = v * xp3[k] % q, v * xp4[k] % q
        self.vinv, self.v = inv(v), v
        self.plus, self.minus, self.rbits = naf_digits(P["r"])
        self.phik = (q * q - q + 1) // P["r"]
        self.stats = {"levels": 0}
        self.flat = []                               # the schedule as executed: ("level", (program, level) | None, ...) / ("op", name)

    def f3_mul_int(self, a, b):
        q = self.q
        d = [0] * 5
        for i in range(3):
            for j in range(3):
                d[i + j] = (d[i + j] + a[i] * b[j]) % q
        return [(d[k] + d[3] * self.xp[0][k] + d[4] * self.xp[1][k]) % q for k in range(3)]

    def f3_pow(self, a, n):
        r = [1, 0, 0]
        for bit in bin(n)[2:]:
            r = self.f3_mul_int(r, r)
            if bit == "1":
                r = self.f3_mul_int(r, a)
        return r

    def run_level(self, *tracks, load=None):
        """one VM level: the given (program, level) pairs side by side -- every lane reads before any lane writes; `load`
        (pairing_pp_apply): line `load` of the table arrives in coefficient bank load % 2 when the level is over"""
        writes = []
        for tr in tracks:
            for p, lev in tr:
                assert len(p.levels[lev]) <= 32, (p.name, lev)
                for n in p.levels[lev]:
                    acc = 0
                    for a, b in n.terms:
                        acc += self.env[p.ref(a)] * self.env[p.ref(b)]
                    writes.append((n.slot, acc % self.q))
        names_written = [w[0] for w in writes]
        assert len(set(names_written)) == len(names_written), names_written
        # an eight-term level deals FOUR lanes to a sum (pairing_dw.cuh exec_split8): sixteen sums in all, both tracks together
        if any(len(n.terms) > 4 for tr in tracks for p, lev in tr for n in p.levels[lev]):
            assert sum(len(p.levels[lev]) for tr in tracks for p, lev in tr) <= 16, [(p.name, lev) for tr in tracks for p, lev in tr]
        for s, v in writes:
            self.env[s] = v
        self.stats["levels"] += 1
        a = tracks[0][0] if len(tracks) > 0 and tracks[0] else None
        b = tracks[1][0] if len(tracks) > 1 and tracks[1] else None
        if load is not None:
            for c, n in enumerate(line_names(load % 2)):
                self.env[n] = self.table[load][c]
        self.flat.append(("level", (a[0].name, a[1]) if a else None, (b[0].name, b[1]) if b else None, load))

    def run(self, name):
        p = self.progs[name]
        for lev in range(len(p.levels)):
            self.run_level([(p, lev)])

    def set_point(self, g1):
        """the first argument: curve_is_valid_point (the driver's lane code), then the point track's state"""
        q, e, fb = self.q, self.env, self.fb
        gi = lambda b, i: int.from_bytes(b[fb * i:fb * (i + 1)], "big") % q
        Px, Py = gi(g1, 0), gi(g1, 1)
        a, b = self.P["a"], self.P["b"]
        e.update(X=Px, Y=Py, Z=1, nZ=q - 1, W=a % q, ZZ=1, ZZZ=1, Px=Px, Py=Py, nPy=(q - Py) % q)
        return ((Px * Px + a) * Px + b - Py * Py) % q == 0

    def set_twist(self, g2):
        """the second argument: the check on the twist, the twist map, and f = 1"""
        q, e, fb = self.q, self.env, self.fb
        gi = lambda b, i: int.from_bytes(b[fb * i:fb * (i + 1)], "big") % q
        Qx, Qy = [gi(g2, i) for i in range(3)], [gi(g2, 3 + i) for i in range(3)]
        a, b, v = self.P["a"], self.P["b"], self.v
        ta, tb = a * v * v % q, b * v * v * v % q
        x2 = self.f3_mul_int(Qx, Qx)
        x2[0] = (x2[0] + ta) % q
        x3 = self.f3_mul_int(x2, Qx)
        x3[0] = (x3[0] + tb) % q
        for i in range(3):
            e[QX[i]], e[QY[i]] = Qx[i] * self.vinv % q, Qy[i] * self.vinv * self.vinv % q
            e[F[0][i]], e[F[1][i]] = (1 if i == 0 else 0), 0
        return x3 == self.f3_mul_int(Qy, Qy)

    def steps(self):
        """the Miller loop's steps in order: ("dbl",) / ("add", negative digit) / ("sqr",)"""
        dig = lambda m: ((self.plus >> m) & 1) - ((self.minus >> m) & 1)
        steps = []
        for m in range(self.rbits - 2, -1, -1):
            steps.append(("dbl",))
            if m > 0 and dig(m):
                steps.append(("add", dig(m) < 0))
            if m > 0:
                steps.append(("sqr",))
        return steps

    def pairing(self, g1, g2):
        if not self.miller(g1, g2):
            return None
        return self.final_exp()

    def miller(self, g1, g2):
        """the Miller value into f; False: an argument is not on its curve (the driver stores the identity then)"""
        ok = self.set_point(g1)
        ok = self.set_twist(g2) and ok
        # the Miller loop (d_miller_lane), software-pipelined: the point work of a step runs beside the accumulator's work
        # of the step before
        steps = self.steps()
        pt = [s for s in steps if s[0] != "sqr"]
        fs = []                                      # accumulator track: ("mul", index of the line) / ("sqr",)
        li = 0
        for s in steps:
            if s[0] == "sqr":
                fs.append(("sqr",))
            else:
                fs.append(("mul", li))
                li += 1

        def pt_name(s, bank):
            return "pt_dbl%d" % bank if s[0] == "dbl" else "pt_add%s%d" % ("m" if s[1] else "p", bank)
        # Two tracks advance one level per VM level (the accumulator track on the first lanes, the point track on the last):
        #   * point program j leaves the COEFFICIENTS of line j in bank j % 2 and, in its first level, evaluates line j - 1
        #     (the other coefficient bank) into value bank (j - 1) % 2; after the last one a one-level program evaluates the last line;
        #   * the accumulator's product with line i (two levels, both read value bank i % 2) starts when that value is there:
        #     point program i + 1 has run its first TWO levels;
        #   * point program j starts when the product with line j - 3 is COMPLETE (its first levels overwrite that value bank).
        # dw_sched.h (the host) is this loop.
        P = pt + [("eval",)]
        fi = pi = 0
        fprog = pprog = None
        flev = plev = 0
        line_ready = 0                               # lines whose value is complete
        fmul_done = 0                                # products with a line completed
        f_is_mul = False
        while fi < len(fs) or fprog is not None:
            if fprog is None:
                if fs[fi][0] == "sqr":
                    fprog, flev, f_is_mul = self.progs["f_sqr"], 0, False
                    fi += 1
                elif fs[fi][1] < line_ready:
                    fprog, flev, f_is_mul = self.progs["f_mul%d" % (fs[fi][1] % 2)], 0, True
                    fi += 1
            if pprog is None and pi < len(P) and (pi < 3 or fmul_done >= pi - 2):
                pprog, plev = self.progs["pt_eval%d" % (pi % 2) if P[pi][0] == "eval" else pt_name(P[pi], pi % 2)], 0
                pi += 1
            assert fprog is not None or pprog is not None
            self.run_level([(fprog, flev)] if fprog else [], [(pprog, plev)] if pprog else [])
            if fprog is not None:
                flev += 1
                if flev == len(fprog.levels):
                    fmul_done += 1 if f_is_mul else 0
                    fprog = None
            if pprog is not None:
                plev += 1
                if plev == 2 and pi >= 2:
                    line_ready = pi - 1
                if plev == len(pprog.levels):
                    pprog = None
        assert pi == len(P) and pprog is None
        return ok

    # ---- element_prod_pairing: one wavefront per TERM for the Miller values, then one per product ----
    def product(self, terms):
        """prod e(g1_t, g2_t): the Miller values of the terms (kernel 1, independent wavefronts), their product by f_mul0 with
        term t's value loaded into value bank 0, ONE final exponentiation (kernel 2); any invalid term: the identity"""
        vals, ok = [], True
        for g1, g2 in terms:
            ok = self.miller(g1, g2) and ok
            vals.append([self.env[n] for n in F[0] + F[1]])
        self.flat = []
        for n, x in zip(F[0] + F[1], vals[0]):
            self.env[n] = x
        for val in vals[1:]:
            for n, x in zip(sum(value_names(0), []), val):
                self.env[n] = x
            self.run("f_mul0")
        r = self.final_exp()
        return r if ok else None

    # ---- pairing_pp_init / pairing_pp_apply: the lines' coefficients come from a table ----
    def pp_table(self, g1):
        """what d_pp_init_lane leaves (here in this script's scaling of the lines -- any factor of F_q^* is as good)"""
        ok = self.set_point(g1)
        keep, tab = self.flat, []
        for j, s in enumerate(st for st in self.steps() if st[0] != "sqr"):
            name = "pt_dbl%d" % (j % 2) if s[0] == "dbl" else "pt_add%s%d" % ("m" if s[1] else "p", j % 2)
            self.run(name)
            tab.append([self.env[n] for n in line_names(j % 2)])
        self.flat = keep
        return tab, ok

    def pp_apply(self, table, p_valid, g2):
        """the accumulator track as in `miller`; the point track only EVALUATES: line i (coefficient bank i % 2, loaded beside the
        first level of the evaluation of line i - 1) -> value bank i % 2 by pt_eval{(i + 1) % 2}, two levels;
          * the product with line i starts when its evaluation is complete;
          * the evaluation of line i starts when the product with line i - 2 is complete (it overwrites that value bank).
        dw_sched.h build_pp is this loop."""
        ok = self.set_twist(g2) and p_valid
        self.table = table
        fs, li = [], 0
        for s in self.steps():
            if s[0] == "sqr":
                fs.append(("sqr",))
            else:
                fs.append(("mul", li))
                li += 1
        nl = li
        assert nl == len(table)
        for c, n in enumerate(line_names(0)):
            self.env[n] = table[0][c]
        self.flat.append(("op", "loadline", 0))
        fi = pi = evals_done = fmul_done = 0
        fprog = pprog = None
        flev = plev = 0
        f_is_mul = False
        while fi < len(fs) or fprog is not None:
            if fprog is None:
                if fs[fi][0] == "sqr":
                    fprog, flev, f_is_mul = self.progs["f_sqr"], 0, False
                    fi += 1
                elif fs[fi][1] < evals_done:
                    fprog, flev, f_is_mul = self.progs["f_mul%d" % (fs[fi][1] % 2)], 0, True
                    fi += 1
            if pprog is None and pi < nl and (pi < 2 or fmul_done >= pi - 1):
                pprog, plev = self.progs["pt_eval%d" % ((pi + 1) % 2)], 0
                pi += 1
            assert fprog is not None or pprog is not None
            load = pi if pprog is not None and plev == 0 and pi < nl else None
            self.run_level([(fprog, flev)] if fprog else [], [(pprog, plev)] if pprog else [], load=load)
            if fprog is not None:
                flev += 1
                if flev == len(fprog.levels):
                    fmul_done += 1 if f_is_mul else 0
                    fprog = None
            if pprog is not None:
                plev += 1
                if plev == len(pprog.levels):
                    evals_done += 1
                    pprog = None
        assert pi == nl and pprog is None and evals_done == nl
        r = self.final_exp()
        return r if ok else None

    def final_exp(self):
        q, e = self.q, self.env
        self.run("fe1")
        self.flat.append(("op", "bzero"))
        b0 = all(e["wB%d" % i] == 0 for i in range(3))
        for i in range(3):
            e["Bn%d" % i] = (1 if i == 0 else 0) if b0 else e["wB%d" % i]
        self.run("fe2")
        assert e["nrm1"] == 0 and e["nrm2"] == 0
        self.flat.append(("op", "inv"))
        e["ninv"] = pow(e["nrm0"], -1, q) if e["nrm0"] else 0
        self.run("fe3")
        nb = self.phik.bit_length()
        for j in range(nb - 1, -1, -1):
            bit = (self.phik >> j) & 1 if j else 0
            self.run("lucas%d" % bit)
        self.run("fe4")
        self.flat.append(("op", "end"))
        return [e[F[0][i]] for i in range(3)] + [e[F[1][i]] for i in range(3)]


OPS = {"level": 0, "bzero": 1, "inv": 2, "end": 3, "loadline": 4}


def pack_entry(progs_index, a, b, op=0, load=None):
    """the 64-bit schedule entry of dw_sched.h: row a | lanes a | row b | lanes b | terms | op | table line | line flag"""
    ra, la, ta = progs_index[a] if a else (0, 0, 0)
    rb, lb, tb = progs_index[b] if b else (0, 0, 0)
    e = ra | la << 12 | rb << 17 | lb << 29 | max(ta if la else 0, tb if lb else 0) << 34 | op << 38
    if load is not None:
        assert load < 4096
        e |= load << 42 | 1 << 55
    return e


def flat_schedule(kind="pairing", pname="d159"):
    """the packed schedules as the model executes them (compared with the host's dw_sched.h by the tests): "pairing";
    "miller" (a term of a product: the Miller value only); "finish" (the two levels of a product with a term's value, then the
    final exponentiation); "pp" (pairing_pp_apply)"""
    progs = build()
    idx, rows = {}, 0
    for name in sorted(progs):
        for lev, (T, lanes) in enumerate(progs[name].table()):
            idx[(name, lev)] = (rows, len(lanes), T)
            rows += len(lanes)
    M = Model(pname, progs)
    g1, g2, gt = load_vec(os.path.join(ROOT, "tests", "golden", "d_rand32.vec"))
    if kind == "pairing":
        M.pairing(g1[0], g2[0])
    elif kind == "miller":
        M.miller(g1[0], g2[0])
        M.flat.append(("op", "end"))
    elif kind == "finish":
        M.product([(g1[0], g2[0]), (g1[1], g2[1])])
    else:
        tab, ok = M.pp_table(g1[0])
        M.flat = []
        M.pp_apply(tab, ok, g2[0])
    out = []
    for e in M.flat:
        if e[0] == "level":
            out.append(pack_entry(idx, e[1], e[2], 0, e[3]))
        else:
            out.append(pack_entry(idx, None, None, OPS[e[1]], e[2] if len(e) > 2 else None))
    return out


def load_vec(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"PBCVEC01"
    t, n, k, l1, l2, lt = struct.unpack("<6I", raw[8:32])
    off = 32
    g1 = [raw[off + i * l1: off + (i + 1) * l1] for i in range(n * k)]; off += n * k * l1
    g2 = [raw[off + i * l2: off + (i + 1) * l2] for i in range(n * k)]; off += n * k * l2
    gt = [raw[off + i * lt: off + (i + 1) * lt] for i in range(n)]
    return g1, g2, gt


OTHER_FIELDS = ["d278027-190-181", "d277699-175-167", "d105171-196-185", "d201", "d224"]      # six- and seven-word fields: the same tables


def check(progs, count=6, others=True):
    bad = 0
    levels = {}
    ident = [1, 0, 0, 0, 0, 0]
    M = Model("d159", progs)
    want_of = lambda b, fb=20: [int.from_bytes(b[fb * c:fb * c + fb], "big") for c in range(6)]
    for name in ("d_rand32.vec", "d_edge20.vec"):
        g1, g2, gt = load_vec(os.path.join(ROOT, "tests", "golden", name))
        for i in range(min(count, len(gt))):
            M.stats["levels"] = 0
            r = M.pairing(g1[i], g2[i])
            levels["pairing"] = M.stats["levels"]
            if (r or ident) != want_of(gt[i]):
                bad += 1
                print("MISMATCH", name, i)
            if i < 3:                                 # the same pairing from the table of pairing_pp_init
                tab, ok = M.pp_table(g1[i])
                M.stats["levels"] = 0
                r = M.pp_apply(tab, ok, g2[i])
                levels["pp_apply"] = M.stats["levels"]
                if (r or ident) != want_of(gt[i]):
                    bad += 1
                    print("MISMATCH (pp)", name, i)
    for name, units in (("d_prod16x4.vec", 2), ("d_prod3x10_edge.vec", 10)):
        g1, g2, gt = load_vec(os.path.join(ROOT, "tests", "golden", name))
        k = len(g1) // len(gt)
        for i in range(min(units, len(gt))):
            r = M.product([(g1[i * k + t], g2[i * k + t]) for t in range(k)])
            if (r or ident) != want_of(gt[i]):
                bad += 1
                print("MISMATCH (product)", name, i)
    for pname in (OTHER_FIELDS if others else []):    # the programs do not know the field: other curves, other loop lengths
        M = Model(pname, progs)
        for name in (pname + "_rand12.vec", pname + "_edge8.vec"):
            g1, g2, gt = load_vec(os.path.join(ROOT, "tests", "golden", name))
            for i in range(min(3, len(gt))):
                r = M.pairing(g1[i], g2[i])
                if (r or ident) != want_of(gt[i], M.fb):
                    bad += 1
                    print("MISMATCH", name, i)
    return bad, levels


# ---------------------------------------------------------------------------------------------------------------------
# the header
# ---------------------------------------------------------------------------------------------------------------------
def emit(progs):
    """rows of five dwords per sum: out | x0..x7 | y0..y7 as bytes (unused terms: the ZERO slot); a level = (first row, T, lanes)"""
    out = ["// dw_tables.h -- GENERATED by tools/dw_gen.py (do not edit): the level programs of the wave-per-pairing type d kernel",
           "// (pairing_dw.cuh).  One ROW of five dwords per sum: the slot it writes, eight x operands, eight y operands, one byte each",
           "// (terms a sum does not have name the ZERO slot); a LEVEL is (first row, terms per sum, working lanes).",
           "#pragma once", "#include <stdint.h>", "namespace pbc { namespace dw {",
           "constexpr int kSlots = %d;" % len(SLOTS.order)]
    keep = lambda n: "." not in n or n.split(".")[0] in ("f", "L0", "L1", "V0", "V1")
    out.append("enum Slot : int { " + ", ".join("S_%s = %d" % (n.replace(".", "_"), i) for i, n in enumerate(SLOTS.order) if keep(n)) + " };")
    rows, index = [], []
    z = SLOTS["ZERO"]
    for name in sorted(progs):
        tab = progs[name].table()
        first = len(index)
        for T, lanes in tab:
            index.append((len(rows), T, len(lanes)))
            for o, xs, ys in lanes:
                b = [o] + xs + [z] * (8 - len(xs)) + ys + [z] * (8 - len(ys)) + [0, 0, 0]
                rows.append([b[4 * i] | b[4 * i + 1] << 8 | b[4 * i + 2] << 16 | b[4 * i + 3] << 24 for i in range(5)])
        out.append("constexpr int P_%s = %d, N_%s = %d;" % (name, first, name, len(tab)))
    out.append("enum { OP_LEVEL = 0, OP_BZERO = 1, OP_INV = 2, OP_END = 3, OP_LOADLINE = 4 };      // schedule entries (dw_sched.h)")
    out.append("struct LevelRef { uint16_t row; uint8_t T, lanes; };")
    out.append("constexpr int kLevels = %d, kRows = %d;" % (len(index), len(rows)))
    out.append("// (the level table is read by the HOST: it flattens a pairing into a schedule of packed entries, dw_sched.h)")
    out.append("static const LevelRef h_level[kLevels] = {" + ", ".join("{%d, %d, %d}" % x for x in index) + "};")
    out.append("__device__ const uint32_t g_rows[kRows * 5] = {" + ",".join("0x%xu" % w for r in rows for w in r) + "};")
    out.append("} }  // namespace pbc::dw")
    return "\n".join(out) + "\n"


def main():
    progs = build()
    assert len(SLOTS.order) <= 255, len(SLOTS.order)
    bad, levels = check(progs)
    for name in sorted(progs):
        p = progs[name]
        print("%-10s levels %2d  sums %3d  widest level %2d lanes  terms per level %s" % (name, len(p.levels), len(p.nodes), max(len(r) for r in p.levels),
              [max(len(n.terms) for n in r) for r in p.levels]))
    print("slots %d; levels executed: %s; vectors: %s" % (len(SLOTS.order), levels, "MISMATCH" if bad else "ok"))
    if bad:
        sys.exit(1)
    text = emit(progs)
    path = os.path.join(ROOT, "pbc_amd", "csrc", "dw_tables.h")
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(path) and open(path).read() == text else 1)
    open(path, "w").write(text)
    print("wrote", path, len(text), "bytes")


if __name__ == "__main__":
    main()
