#!/usr/bin/env python3
"""Writes pbc_amd/param/f_r256.param: a type f (Barreto-Naehrig) parameter set whose group order r has 256 bits and lies
above (2/3) 2^256, so that the signed-digit (NAF) form of r >> 1 has its leading digit at position 256 -- the case
ADVICE r2 found unsupported by 8-word digit arrays.  pbc_param_init_f_gen (ecc/f_param.c:481-595) cannot produce it: it
starts its search at x = 2^((bits - 6) / 4), which gives 254 bits for bits = 256 and 258 for bits = 260.  The family is
the reference's: q = 36x^4 + 36x^3 + 24x^2 + 6x + 1, r = 36x^4 + 36x^3 + 18x^2 + 6x + 1 (tryplusx, f_param.c:83-95);
b, beta, alpha are found the way f_gen finds them (curve of order r, a non-residue, x^6 + alpha irreducible over F_q^2
with the sextic twist y^2 = x^3 - alpha b of order divisible by r) with a deterministic search instead of random draws.
The UNMODIFIED reference then reads the file like any other (tests/golden/make_golden.sh) and its outputs pin the oracle
and the kernels.  Pure Python, no dependencies:  python3 tests/golden/gen_bn_param.py > pbc_amd/param/f_r256.param
"""
import sys


def is_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def find_x():
    x = int(2 ** 62.66) | 1                     # 36 x^4 ~ 0.76 * 2^256
    while True:
        q = 36 * x ** 4 + 36 * x ** 3 + 24 * x ** 2 + 6 * x + 1
        r = q - 6 * x * x
        if q.bit_length() == 256 and r.bit_length() == 256 and 3 * r > 2 ** 257 and is_prime(q) and is_prime(r):
            return x, q, r
        x += 2


class Fq2:
    """F_q[s] / (s^2 - beta), elements as pairs"""

    def __init__(self, q, beta):
        self.q, self.beta = q, beta

    def add(self, a, b):
        return ((a[0] + b[0]) % self.q, (a[1] + b[1]) % self.q)

    def sub(self, a, b):
        return ((a[0] - b[0]) % self.q, (a[1] - b[1]) % self.q)

    def mul(self, a, b):
        q = self.q
        return ((a[0] * b[0] + self.beta * a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)

    def inv(self, a):
        q = self.q
        n = pow((a[0] * a[0] - self.beta * a[1] * a[1]) % q, -1, q)
        return (a[0] * n % q, -a[1] * n % q)

    def pow(self, a, e):
        r = (1, 0)
        while e:
            if e & 1:
                r = self.mul(r, a)
            a = self.mul(a, a)
            e >>= 1
        return r

    def sqrt(self, c):                          # q = 3 mod 4; None if c is not a square
        q = self.q
        if c == (0, 0):
            return c
        n = (c[0] * c[0] - self.beta * c[1] * c[1]) % q
        if pow(n, (q - 1) // 2, q) != 1:
            return None
        n = pow(n, (q + 1) // 4, q)
        for sn in (n, -n):
            t = (c[0] + sn) * pow(2, -1, q) % q
            if t and pow(t, (q - 1) // 2, q) == 1:
                x0 = pow(t, (q + 1) // 4, q)
                x1 = c[1] * pow(2 * x0, -1, q) % q
                if self.mul((x0, x1), (x0, x1)) == c:
                    return (x0, x1)
        if c[1] == 0:                           # c in F_q, a non-residue there: sqrt = s sqrt(c / beta)
            t = c[0] * pow(self.beta, -1, q) % q
            x1 = pow(t, (q + 1) // 4, q)
            if self.mul((0, x1), (0, x1)) == c:
                return (0, x1)
        return None


def ec_mul(F, P, k, zero, one_inv):
    """k P on y^2 = x^3 + b (a = 0) in affine coordinates over the field object F (None = O)"""
    def add(A, B):
        if A is None:
            return B
        if B is None:
            return A
        if A[0] == B[0]:
            if F.add(A[1], B[1]) == zero:
                return None
            lam = F.mul(F.mul((3, 0) if isinstance(zero, tuple) else 3, F.mul(A[0], A[0])), F.inv(F.add(A[1], A[1])))
        else:
            lam = F.mul(F.sub(B[1], A[1]), F.inv(F.sub(B[0], A[0])))
        x3 = F.sub(F.sub(F.mul(lam, lam), A[0]), B[0])
        return (x3, F.sub(F.mul(lam, F.sub(A[0], x3)), A[1]))
    R = None
    while k:
        if k & 1:
            R = add(R, P)
        P = add(P, P)
        k >>= 1
    return R


class Fq:
    def __init__(self, q):
        self.q = q

    def add(self, a, b):
        return (a + b) % self.q

    def sub(self, a, b):
        return (a - b) % self.q

    def mul(self, a, b):
        return a * b % self.q

    def inv(self, a):
        return pow(a, -1, self.q)


def main():
    x, q, r = find_x()
    assert q % 4 == 3 and q % 6 == 1
    K = Fq(q)
    # b: the first value for which a point of y^2 = x^3 + b is killed by r (the curve then has order r: Hasse)
    b = 1
    while True:
        b += 1
        P = None
        for x0 in range(1, 50):
            y2 = (x0 ** 3 + b) % q
            if pow(y2, (q - 1) // 2, q) == 1:
                P = (x0, pow(y2, (q + 1) // 4, q))
                break
        if P is not None and ec_mul(K, P, r, 0, None) is None:
            break
    # beta: a non-residue that is not just -1 (so that the kernels' beta products are exercised on a full-size value)
    beta = 0x9E3779B97F4A7C15F39CC0605CEDC8341082276BF3A27251F86C6A11D0C18E95 % q
    while pow(beta, (q - 1) // 2, q) != q - 1:
        beta += 1
    F = Fq2(q, beta)
    Q1 = q * q - 1
    a = [0x243F6A8885A308D313198A2E03707344A4093822299F31D0082EFA98EC4E6C89 % q, 0x452821E638D01377BE5466CF34E90C6CC0AC29B7C97C50DD3F84D5B5B5470917 % q]
    while True:
        c = ((-a[0]) % q, (-a[1]) % q)          # X^6 = -alpha
        if F.pow(c, Q1 // 2) != (1, 0) and F.pow(c, Q1 // 3) != (1, 0):
            break
        a[0] += 1
    alpha = (a[0], a[1])
    n2 = r * (2 * q - r)                        # order of the sextic twist that carries the order-r subgroup
    for attempt in range(2):
        tb = F.mul(((-alpha[0]) % q, (-alpha[1]) % q), (b, 0))
        ok = True
        found = 0
        xx = 1
        while found < 2:
            X = (xx, 1)
            xx += 1
            y = F.sqrt(F.add(F.mul(F.mul(X, X), X), tb))
            if y is None:
                continue
            found += 1
            if ec_mul(F, (X, y), n2, (0, 0), None) is not None:
                ok = False
                break
        if ok:
            break
        alpha = F.pow(alpha, 5)                 # the wrong twist: alpha^5 (f_param.c:578-583)
    else:
        raise SystemExit("no twist of order divisible by r")
    sys.stdout.write("type f\nq %d\nr %d\nb %d\nbeta %d\nalpha0 %d\nalpha1 %d\n" % (q, r, b, beta, alpha[0], alpha[1]))
    sys.stderr.write("x = %d (%d bits); q, r: %d, %d bits; r / 2^256 = %.4f\n" % (x, x.bit_length(), q.bit_length(), r.bit_length(), r / 2.0 ** 256))


if __name__ == "__main__":
    main()
