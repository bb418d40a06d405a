/*
 * pbc_oracle.c -- TEST INFRASTRUCTURE ONLY (see pbc_oracle.h).
 *
 * CPU restatement, in plain C (64-bit limbs + unsigned __int128, no GMP), of the
 * reference's pairing hot path.  Every routine cites the reference file:line whose
 * algorithm it follows.  All citations are relative to /root/reference.
 *
 * Only the *values* matter for parity: an Fq element is an exact residue, so the
 * radix/limb choices here (R = 2^(64 n), same as arith/montfp.c:571-587) are private.
 *
 * Parity status: PINNED (pbc/pairing_test.pbc KAT + tests/golden vectors produced by
 * the unmodified reference; see tests/test_oracle.py).
 */
#include "pbc_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
#define MAXL 8                       /* 512-bit moduli at most (Type A a.param) */

static uint64_t g_mul_count, g_inv_count;
void oracle_counters(uint64_t *mul, uint64_t *inv, int reset) {
  if (mul) *mul = g_mul_count;
  if (inv) *inv = g_inv_count;
  if (reset) g_mul_count = g_inv_count = 0;
}

/* ------------------------------------------------------------------ */
/* plain multi-limb integers                                           */
/* ------------------------------------------------------------------ */
typedef struct { uint64_t v[MAXL]; } fe;          /* Fq element, Montgomery form, reduced */
#define BIGL 40
typedef struct { uint64_t v[BIGL]; } big;         /* scratch integer for parsing/exponents */

static int bn_cmp(const uint64_t *a, const uint64_t *b, int n) {
  for (int i = n - 1; i >= 0; i--) {
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  }
  return 0;
}
static uint64_t bn_add(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
  u128 c = 0;
  for (int i = 0; i < n; i++) { c += (u128) a[i] + b[i]; r[i] = (uint64_t) c; c >>= 64; }
  return (uint64_t) c;
}
static uint64_t bn_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
  uint64_t bw = 0;
  for (int i = 0; i < n; i++) {
    u128 d = (u128) a[i] - b[i] - bw;
    r[i] = (uint64_t) d; bw = (uint64_t) (d >> 64) & 1;
  }
  return bw;
}
static int bn_is0(const uint64_t *a, int n) {
  uint64_t x = 0;
  for (int i = 0; i < n; i++) x |= a[i];
  return x == 0;
}
static int big_bits(const big *a) {
  for (int i = BIGL - 1; i >= 0; i--) if (a->v[i]) {
    int b = 63; while (!((a->v[i] >> b) & 1)) b--;
    return i * 64 + b + 1;
  }
  return 0;
}
static int big_bit(const big *a, int i) { return (int) ((a->v[i / 64] >> (i % 64)) & 1); }
static int big_from_dec(big *r, const char *s) {
  memset(r, 0, sizeof *r);
  if (!*s) return 1;
  for (; *s >= '0' && *s <= '9'; s++) {
    u128 c = (u128) (*s - '0');
    for (int i = 0; i < BIGL; i++) { c += (u128) r->v[i] * 10; r->v[i] = (uint64_t) c; c >>= 64; }
    if (c) return 1;
  }
  return 0;
}
static void big_from_be(big *r, const uint8_t *b, size_t len) {
  memset(r, 0, sizeof *r);
  for (size_t i = 0; i < len && i < BIGL * 8; i++) r->v[i / 8] |= (uint64_t) b[len - 1 - i] << (8 * (i % 8));
}

/* ------------------------------------------------------------------ */
/* Fq: Montgomery arithmetic, arith/montfp.c                           */
/* ------------------------------------------------------------------ */
typedef struct {
  int n;                 /* limbs (montfp.c:571) */
  int nbytes;            /* fixed_length_in_bytes = ceil(bits/8) (montfp.c:577) */
  uint64_t p[MAXL];
  uint64_t ninv;         /* -p^-1 mod 2^64 (montfp.c:592-598) */
  fe R, R2, zero;        /* R mod p (=1), R^2 mod p */
  big pm2;               /* p-2, Fermat exponent */
} fpctx;

/* c = a*b*R^-1 mod p : interleaved Montgomery product, same value as mont_mul
 * (arith/montfp.c:334-364).  Accepts a < R (not nec. reduced) as long as b < p. */
static void fp_mul(const fpctx *F, fe *c, const fe *a, const fe *b) {
  int n = F->n;
  uint64_t t[MAXL + 2] = {0};
  g_mul_count++;
  for (int i = 0; i < n; i++) {
    u128 uv; uint64_t cy = 0;
    for (int j = 0; j < n; j++) {
      uv = (u128) a->v[j] * b->v[i] + t[j] + cy;
      t[j] = (uint64_t) uv; cy = (uint64_t) (uv >> 64);
    }
    uv = (u128) t[n] + cy; t[n] = (uint64_t) uv; t[n + 1] = (uint64_t) (uv >> 64);
    uint64_t m = t[0] * F->ninv;
    uv = (u128) m * F->p[0] + t[0]; cy = (uint64_t) (uv >> 64);
    for (int j = 1; j < n; j++) {
      uv = (u128) m * F->p[j] + t[j] + cy;
      t[j - 1] = (uint64_t) uv; cy = (uint64_t) (uv >> 64);
    }
    uv = (u128) t[n] + cy; t[n - 1] = (uint64_t) uv;
    t[n] = t[n + 1] + (uint64_t) (uv >> 64);
  }
  if (t[n] || bn_cmp(t, F->p, n) >= 0) bn_sub(t, t, F->p, n);   /* montfp.c:114-119 */
  memset(c, 0, sizeof *c);
  memcpy(c->v, t, 8 * n);
}
static void fp_sqr(const fpctx *F, fe *c, const fe *a) { fp_mul(F, c, a, a); } /* field.c:383 generic_square */
/* fp_add (montfp.c:220-250): add, then subtract p on carry or >= p */
static void fp_add(const fpctx *F, fe *c, const fe *a, const fe *b) {
  uint64_t cy = bn_add(c->v, a->v, b->v, F->n);
  if (cy || bn_cmp(c->v, F->p, F->n) >= 0) bn_sub(c->v, c->v, F->p, F->n);
}
static void fp_dbl(const fpctx *F, fe *c, const fe *a) { fe t = *a; fp_add(F, c, &t, &t); } /* montfp.c:252-270 */
/* fp_sub (montfp.c:282-316) */
static void fp_sub(const fpctx *F, fe *c, const fe *a, const fe *b) {
  if (bn_sub(c->v, a->v, b->v, F->n)) bn_add(c->v, c->v, F->p, F->n);
}
/* fp_neg (montfp.c:318-330) */
static void fp_neg(const fpctx *F, fe *c, const fe *a) {
  if (bn_is0(a->v, F->n)) { *c = F->zero; return; }
  bn_sub(c->v, F->p, a->v, F->n);
}
/* fp_halve (montfp.c:272-280 -> generic via 1/2): a/2 = a>>1 if even else (a+p)>>1 */
static void fp_halve(const fpctx *F, fe *c, const fe *a) {
  uint64_t t[MAXL + 1]; int n = F->n;
  memcpy(t, a->v, 8 * n); t[n] = 0;
  if (t[0] & 1) t[n] = bn_add(t, t, F->p, n);
  for (int i = 0; i < n; i++) t[i] = (t[i] >> 1) | (t[i + 1] << 63);
  memcpy(c->v, t, 8 * n);
}
static int fp_is0(const fpctx *F, const fe *a) { return bn_is0(a->v, F->n); }
static int fp_eq(const fpctx *F, const fe *a, const fe *b) { return !bn_cmp(a->v, b->v, F->n); }
/* a^e, plain square-and-multiply (value identical to fp_pow_mpz, montfp.c:379-396) */
static void fp_pow(const fpctx *F, fe *c, const fe *a, const big *e) {
  fe r = F->R, base = *a;
  int nb = big_bits(e);
  for (int i = nb - 1; i >= 0; i--) {
    fp_sqr(F, &r, &r);
    if (big_bit(e, i)) fp_mul(F, &r, &r, &base);
  }
  *c = r;
}
/* fp_invert (montfp.c:401-422) uses mpz_invert; the inverse is unique so Fermat
 * a^(p-2) gives the identical residue.  0 -> 0 (reference: undefined, "requires nonzero"). */
static void fp_inv(const fpctx *F, fe *c, const fe *a) {
  uint64_t save = g_mul_count;
  g_inv_count++;
  fp_pow(F, c, a, &F->pm2);
  g_mul_count = save;                /* count an inversion as an inversion, not as muls */
}
static void fp_set_ui(const fpctx *F, fe *c, uint64_t x) {
  fe t; memset(&t, 0, sizeof t); t.v[0] = x;
  fp_mul(F, c, &t, &F->R2); g_mul_count--;
}
/* fp_from_bytes (montfp.c:498-517): big-endian, reduced mod p, to Montgomery form */
static void fp_from_bytes(const fpctx *F, fe *c, const uint8_t *b) {
  fe t; memset(&t, 0, sizeof t);
  for (int i = 0; i < F->nbytes; i++) t.v[i / 8] |= (uint64_t) b[F->nbytes - 1 - i] << (8 * (i % 8));
  fp_mul(F, c, &t, &F->R2); g_mul_count--;
}
/* fp_to_bytes (montfp.c:487-496) + pbc_mpz_out_raw_n (field.c:629-638) */
static void fp_to_bytes(const fpctx *F, uint8_t *b, const fe *a) {
  fe one, t; memset(&one, 0, sizeof one); one.v[0] = 1;
  fp_mul(F, &t, a, &one); g_mul_count--;
  for (int i = 0; i < F->nbytes; i++) b[F->nbytes - 1 - i] = (uint8_t) (t.v[i / 8] >> (8 * (i % 8)));
}
/* field_init_mont_fp (montfp.c:533-600) */
static int fp_init(fpctx *F, const big *p) {
  memset(F, 0, sizeof *F);
  int bits = big_bits(p);
  if (bits < 65 || bits > 64 * MAXL || !(p->v[0] & 1)) return 1;
  F->n = (bits + 63) / 64;
  F->nbytes = (bits + 7) / 8;
  memcpy(F->p, p->v, 8 * F->n);
  uint64_t x = 1;                                /* Newton: x = p^-1 mod 2^64 */
  for (int i = 0; i < 6; i++) x *= 2 - F->p[0] * x;
  F->ninv = (uint64_t) 0 - x;
  /* R mod p and R^2 mod p by repeated doubling of 1 */
  fe r; memset(&r, 0, sizeof r); r.v[0] = 1;
  for (int i = 0; i < 64 * F->n; i++) fp_dbl(F, &r, &r);
  F->R = r;
  for (int i = 0; i < 64 * F->n; i++) fp_dbl(F, &r, &r);
  F->R2 = r;
  F->pm2 = *p;
  big two; memset(&two, 0, sizeof two); two.v[0] = 2;
  bn_sub(F->pm2.v, F->pm2.v, two.v, BIGL);
  return 0;
}

/* ------------------------------------------------------------------ */
/* Fq2 = Fq[i]/(i^2+1), arith/fieldquadratic.c fi_*                    */
/* ------------------------------------------------------------------ */
typedef struct { fe x, y; } fe2;

/* fi_mul (fieldquadratic.c:425-457): Karatsuba, 3 M */
static void fi_mul(const fpctx *F, fe2 *n, const fe2 *a, const fe2 *b) {
  fe e0, e1, e2;
  fp_add(F, &e0, &a->x, &a->y);
  fp_add(F, &e1, &b->x, &b->y);
  fp_mul(F, &e2, &e0, &e1);
  fp_mul(F, &e0, &a->x, &b->x);
  fp_mul(F, &e1, &a->y, &b->y);
  fp_sub(F, &e2, &e2, &e0);
  fp_sub(F, &n->x, &e0, &e1);
  fp_sub(F, &n->y, &e2, &e1);
}
/* fi_square (fieldquadratic.c:459-477): (x+y)(x-y), 2xy */
static void fi_sqr(const fpctx *F, fe2 *n, const fe2 *a) {
  fe e0, e1;
  fp_add(F, &e0, &a->x, &a->y);
  fp_sub(F, &e1, &a->x, &a->y);
  fp_mul(F, &e0, &e0, &e1);
  fp_mul(F, &e1, &a->x, &a->y);
  fp_dbl(F, &e1, &e1);
  n->x = e0; n->y = e1;
}
/* fi_invert (fieldquadratic.c:479-496): conj / norm */
static void fi_inv(const fpctx *F, fe2 *n, const fe2 *a) {
  fe e0, e1;
  fp_sqr(F, &e0, &a->x);
  fp_sqr(F, &e1, &a->y);
  fp_add(F, &e0, &e0, &e1);
  fp_inv(F, &e0, &e0);
  fp_mul(F, &n->x, &a->x, &e0);
  fp_neg(F, &e0, &e0);
  fp_mul(F, &n->y, &a->y, &e0);
}

/* ------------------------------------------------------------------ */
/* E(Fq): y^2 = x^3 + a x + b, affine, ecc/curve.c                     */
/* ------------------------------------------------------------------ */
typedef struct { int inf; fe x, y; } pt;

/* curve_is_valid_point (curve.c:57-77) */
static int pt_on_curve(const fpctx *F, const fe *ca, const fe *cb, const pt *P) {
  fe t0, t1;
  if (P->inf) return 1;
  fp_sqr(F, &t0, &P->x);
  fp_add(F, &t0, &t0, ca);
  fp_mul(F, &t0, &t0, &P->x);
  fp_add(F, &t0, &t0, cb);
  fp_sqr(F, &t1, &P->y);
  return fp_eq(F, &t0, &t1);
}
/* curve_from_bytes (curve.c:609-623): x||y, off-curve -> O */
static void pt_from_bytes(const fpctx *F, const fe *ca, const fe *cb, pt *P, const uint8_t *b) {
  P->inf = 0;
  fp_from_bytes(F, &P->x, b);
  fp_from_bytes(F, &P->y, b + F->nbytes);
  if (!pt_on_curve(F, ca, cb, P)) P->inf = 1;
}
static void pt_to_bytes(const fpctx *F, uint8_t *b, const pt *P) {
  if (P->inf) { memset(b, 0, 2 * F->nbytes); return; }   /* x,y of O are 0 after set0 (curve.c:79-85 keeps old; we emit 0) */
  fp_to_bytes(F, b, &P->x);
  fp_to_bytes(F, b + F->nbytes, &P->y);
}
/* double_no_check / curve_double (curve.c:102-151) */
static void pt_dbl(const fpctx *F, const fe *ca, pt *R, const pt *P) {
  fe l, e0, e1, x3, y3;
  if (P->inf || fp_is0(F, &P->y)) { R->inf = 1; return; }
  fp_sqr(F, &l, &P->x);
  fp_dbl(F, &e0, &l); fp_add(F, &l, &l, &e0);       /* 3x^2 */
  fp_add(F, &l, &l, ca);
  fp_dbl(F, &e0, &P->y);
  fp_inv(F, &e0, &e0);
  fp_mul(F, &l, &l, &e0);                            /* lambda */
  fp_dbl(F, &e1, &P->x);
  fp_sqr(F, &x3, &l);
  fp_sub(F, &x3, &x3, &e1);
  fp_sub(F, &e1, &P->x, &x3);
  fp_mul(F, &y3, &e1, &l);
  fp_sub(F, &y3, &y3, &P->y);
  R->inf = 0; R->x = x3; R->y = y3;
}
/* curve_mul = point addition (curve.c:153-207) */
static void pt_add(const fpctx *F, const fe *ca, pt *R, const pt *P, const pt *Q) {
  fe l, e0, x3, y3;
  if (P->inf) { *R = *Q; return; }
  if (Q->inf) { *R = *P; return; }
  if (fp_eq(F, &P->x, &Q->x)) {
    if (fp_eq(F, &P->y, &Q->y) && !fp_is0(F, &P->y)) { pt_dbl(F, ca, R, P); return; }
    R->inf = 1; return;
  }
  fp_sub(F, &e0, &Q->x, &P->x);
  fp_inv(F, &e0, &e0);
  fp_sub(F, &l, &Q->y, &P->y);
  fp_mul(F, &l, &l, &e0);
  fp_sqr(F, &x3, &l);
  fp_sub(F, &x3, &x3, &P->x);
  fp_sub(F, &x3, &x3, &Q->x);
  fp_sub(F, &e0, &P->x, &x3);
  fp_mul(F, &y3, &e0, &l);
  fp_sub(F, &y3, &y3, &P->y);
  R->inf = 0; R->x = x3; R->y = y3;
}
static void pt_mul(const fpctx *F, const fe *ca, pt *R, const pt *P, const big *e) {
  pt acc; acc.inf = 1; memset(&acc.x, 0, sizeof acc.x); memset(&acc.y, 0, sizeof acc.y);
  for (int i = big_bits(e) - 1; i >= 0; i--) {
    pt_dbl(F, ca, &acc, &acc);
    if (big_bit(e, i)) pt_add(F, ca, &acc, &acc, P);
  }
  *R = acc;
}

/* ------------------------------------------------------------------ */
/* pairing object                                                      */
/* ------------------------------------------------------------------ */
struct oracle_pairing {
  int type;
  fpctx Fq;
  big q, r, h;
  int len1, len2, lenT;
  /* type A (ecc/a_param.c:30-34) */
  int exp2, exp1, sign1;
  fe ca, cb;             /* curve a=1, b=0 (a_param.c:1450-1452) */
};

/* ---------------- param text -> key/value (ecc/param.c:100-170) ---------------- */
static const char *kv_find(const char *txt, size_t len, const char *key, char *buf, size_t buflen) {
  size_t klen = strlen(key), i = 0;
  while (i < len) {
    size_t ls = i;
    while (i < len && txt[i] != '\n') i++;
    size_t le = i; if (i < len) i++;
    while (ls < le && (txt[ls] == ' ' || txt[ls] == '\t')) ls++;
    if (le - ls > klen && !memcmp(txt + ls, key, klen) && (txt[ls + klen] == ' ' || txt[ls + klen] == '\t')) {
      size_t vs = ls + klen;
      while (vs < le && (txt[vs] == ' ' || txt[vs] == '\t')) vs++;
      size_t ve = le;
      while (ve > vs && (txt[ve - 1] == ' ' || txt[ve - 1] == '\r' || txt[ve - 1] == '\t')) ve--;
      if (ve - vs >= buflen) return NULL;
      memcpy(buf, txt + vs, ve - vs); buf[ve - vs] = 0;
      return buf;
    }
  }
  return NULL;
}
static int kv_big(const char *txt, size_t len, const char *key, big *out) {
  char buf[1024];
  if (!kv_find(txt, len, key, buf, sizeof buf)) return 1;
  return big_from_dec(out, buf);
}
static int kv_int(const char *txt, size_t len, const char *key, int *out) {
  char buf[64];
  if (!kv_find(txt, len, key, buf, sizeof buf)) return 1;
  *out = atoi(buf);
  return 0;
}

/* a_init_pairing (ecc/a_param.c:1431-1472) + pbc_param_init_a (:1489-1502) */
static int init_a(oracle_pairing *P, const char *txt, size_t len) {
  int sign0;
  if (kv_big(txt, len, "q", &P->q) || kv_big(txt, len, "r", &P->r) || kv_big(txt, len, "h", &P->h)) return 1;
  if (kv_int(txt, len, "exp2", &P->exp2) || kv_int(txt, len, "exp1", &P->exp1) ||
      kv_int(txt, len, "sign1", &P->sign1) || kv_int(txt, len, "sign0", &sign0)) return 1;
  if (fp_init(&P->Fq, &P->q)) return 1;
  P->ca = P->Fq.R;                    /* a = 1 */
  P->cb = P->Fq.zero;                 /* b = 0 */
  P->len1 = P->len2 = P->lenT = 2 * P->Fq.nbytes;
  return 0;
}

int oracle_pairing_init(oracle_pairing **out, const char *txt, size_t len) {
  char tb[16];
  if (!len) len = strlen(txt);
  oracle_pairing *P = calloc(1, sizeof *P);
  if (!P) return 1;
  if (!kv_find(txt, len, "type", tb, sizeof tb)) { free(P); return 1; }
  P->type = tb[0];
  int rc = 1;
  if (!strcmp(tb, "a")) rc = init_a(P, txt, len);
  if (rc) { free(P); return 1; }
  *out = P;
  return 0;
}
void oracle_pairing_clear(oracle_pairing *p) { free(p); }
int oracle_type(const oracle_pairing *p) { return p->type; }
int oracle_len_G1(const oracle_pairing *p) { return p->len1; }
int oracle_len_G2(const oracle_pairing *p) { return p->len2; }
int oracle_len_GT(const oracle_pairing *p) { return p->lenT; }

/* ------------------------------------------------------------------ */
/* Type A pairing, ecc/a_param.c                                       */
/* ------------------------------------------------------------------ */
/* compute_abc_tangent (a_param.c:61-84): affine tangent scaled by -2Vy */
static void a_abc_tangent(const fpctx *F, fe *a, fe *b, fe *c, const fe *Vx, const fe *Vy) {
  fe e0;
  fp_sqr(F, a, Vx);
  fp_add(F, &e0, a, a);
  fp_add(F, a, &e0, a);
  fp_add(F, a, a, &F->R);             /* + cc->a = 1 */
  fp_neg(F, a, a);
  fp_dbl(F, b, Vy);
  fp_mul(F, &e0, b, Vy);
  fp_mul(F, c, a, Vx);
  fp_add(F, c, c, &e0);
  fp_neg(F, c, c);
}
/* compute_abc_tangent_proj (a_param.c:86-112) */
static void a_abc_tangent_proj(const fpctx *F, fe *a, fe *b, fe *c, const fe *Vx, const fe *Vy,
                               const fe *z, const fe *z2) {
  fe e0;
  fp_sqr(F, a, z2);
  fp_sqr(F, b, Vx);
  fp_dbl(F, &e0, b);
  fp_add(F, b, &e0, b);
  fp_add(F, a, a, b);
  fp_neg(F, a, a);
  fp_dbl(F, &e0, Vy);
  fp_mul(F, b, &e0, z2);
  fp_mul(F, b, b, z);
  fp_mul(F, c, Vx, a);
  fp_mul(F, a, a, z2);
  fp_mul(F, &e0, &e0, Vy);
  fp_add(F, c, c, &e0);
  fp_neg(F, c, c);
}
/* compute_abc_line (a_param.c:114-130) */
static void a_abc_line(const fpctx *F, fe *a, fe *b, fe *c, const fe *Vx, const fe *Vy,
                       const fe *V1x, const fe *V1y) {
  fe e0;
  fp_sub(F, a, Vy, V1y);
  fp_sub(F, b, V1x, Vx);
  fp_mul(F, c, Vx, V1y);
  fp_mul(F, &e0, Vy, V1x);
  fp_sub(F, c, c, &e0);
}
/* a_miller_evalfn (a_param.c:306-315): (c - a Qx) + i (b Qy) */
static void a_evalfn(const fpctx *F, fe2 *out, const fe *a, const fe *b, const fe *c,
                     const fe *Qx, const fe *Qy) {
  fp_mul(F, &out->y, a, Qx);
  fp_sub(F, &out->x, c, &out->y);
  fp_mul(F, &out->y, b, Qy);
}
/* lucas_odd (a_param.c:226-283) */
static void a_lucas_odd(const fpctx *F, fe2 *out, fe2 *in, const big *cofactor) {
  fe t0, t1, v0, v1;
  fe *in0 = &in->x, *in1 = &in->y;
  fp_set_ui(F, &t0, 2);
  fp_dbl(F, &t1, in0);
  v0 = t0; v1 = t1;
  int j = big_bits(cofactor) - 1;
  for (;;) {
    if (!j) {
      fp_mul(F, &v1, &v0, &v1); fp_sub(F, &v1, &v1, &t1);
      fp_sqr(F, &v0, &v0);      fp_sub(F, &v0, &v0, &t0);
      break;
    }
    if (big_bit(cofactor, j)) {
      fp_mul(F, &v0, &v0, &v1); fp_sub(F, &v0, &v0, &t1);
      fp_sqr(F, &v1, &v1);      fp_sub(F, &v1, &v1, &t0);
    } else {
      fp_mul(F, &v1, &v0, &v1); fp_sub(F, &v1, &v1, &t1);
      fp_sqr(F, &v0, &v0);      fp_sub(F, &v0, &v0, &t0);
    }
    j--;
  }
  fp_mul(F, in0, &v0, &t1);
  fp_dbl(F, &v1, &v1);
  fp_sub(F, &v1, &v1, in0);
  fp_sqr(F, &t1, &t1);
  fp_sub(F, &t1, &t1, &t0);
  fp_sub(F, &t1, &t1, &t0);
  { fe ti; fp_inv(F, &ti, &t1); fp_mul(F, &v1, &v1, &ti); }   /* element_div (field.c:459-469) */
  fp_halve(F, &v0, &v0);
  fp_mul(F, &v1, &v1, in1);
  out->x = v0; out->y = v1;
}
/* a_tateexp (a_param.c:285-303) */
static void a_tateexp(const fpctx *F, fe2 *out, fe2 *in, const big *cofactor) {
  fe2 temp;
  fi_inv(F, &temp, in);
  fp_neg(F, &in->y, &in->y);
  fi_mul(F, in, in, &temp);
  a_lucas_odd(F, out, in, cofactor);
}

/* a_pairing_proj (a_param.c:1053-1198): the default Type-A map (a_param.c:1443) */
static void a_pairing_proj(const oracle_pairing *P, fe2 *out, const pt *in1, const pt *in2) {
  const fpctx *F = &P->Fq;
  fe Vx = in1->x, Vy = in1->y, V1x, V1y, z = F->R, z2 = F->R;
  fe a, b, c, e0;
  fe2 f, f0, f1;
  const fe *Qx = &in2->x, *Qy = &in2->y;
  f.x = F->R; f.y = F->zero;
  int i, n;
#define POINT_TO_AFFINE() do { fp_inv(F, &z, &z); fp_sqr(F, &e0, &z); fp_mul(F, &Vx, &Vx, &e0); \
    fp_mul(F, &e0, &e0, &z); fp_mul(F, &Vy, &Vy, &e0); z = F->R; z2 = F->R; } while (0)
#define PROJ_DOUBLE() do { fe *e1 = &a, *e2 = &b, *e3 = &c; \
    fp_sqr(F, &e0, &Vx); fp_dbl(F, e1, &e0); fp_add(F, &e0, e1, &e0); fp_sqr(F, e1, &z2); fp_add(F, &e0, &e0, e1); \
    fp_mul(F, &z, &Vy, &z); fp_dbl(F, &z, &z); fp_sqr(F, &z2, &z); \
    fp_sqr(F, e2, &Vy); fp_mul(F, e1, &Vx, e2); fp_dbl(F, e1, e1); fp_dbl(F, e1, e1); \
    fp_dbl(F, e3, e1); fp_sqr(F, &Vx, &e0); fp_sub(F, &Vx, &Vx, e3); \
    fp_sqr(F, e2, e2); fp_dbl(F, e2, e2); fp_dbl(F, e2, e2); fp_dbl(F, e2, e2); \
    fp_sub(F, e1, e1, &Vx); fp_mul(F, &e0, &e0, e1); fp_sub(F, &Vy, &e0, e2); } while (0)
#define DO_TANGENT() do { a_abc_tangent_proj(F, &a, &b, &c, &Vx, &Vy, &z, &z2); \
    a_evalfn(F, &f0, &a, &b, &c, Qx, Qy); fi_mul(F, &f, &f, &f0); } while (0)
  n = P->exp1;
  for (i = 0; i < n; i++) { fi_sqr(F, &f, &f); DO_TANGENT(); PROJ_DOUBLE(); }
  POINT_TO_AFFINE();
  if (P->sign1 < 0) { V1x = Vx; fp_neg(F, &V1y, &Vy); fi_inv(F, &f1, &f); }
  else { V1x = Vx; V1y = Vy; f1 = f; }
  n = P->exp2;
  for (; i < n; i++) { fi_sqr(F, &f, &f); DO_TANGENT(); PROJ_DOUBLE(); }
  fi_mul(F, &f, &f, &f1);
  POINT_TO_AFFINE();
  a_abc_line(F, &a, &b, &c, &Vx, &Vy, &V1x, &V1y);
  a_evalfn(F, &f0, &a, &b, &c, Qx, Qy);
  fi_mul(F, &f, &f, &f0);
  a_tateexp(F, out, &f, &P->h);
#undef POINT_TO_AFFINE
#undef PROJ_DOUBLE
#undef DO_TANGENT
}

/* multi_double (ecc/curve.c:210-281): simultaneous affine doubling, ONE inversion */
static void a_multi_double(const fpctx *F, const fe *ca, pt *V, int n, fe *table) {
  fe e0, e1, e2;
  for (int i = 0; i < n; i++) {
    fp_dbl(F, &table[i], &V[i].y);
    if (i > 0) fp_mul(F, &table[i], &table[i], &table[i - 1]);
  }
  fp_inv(F, &e2, &table[n - 1]);
  for (int i = n - 1; i > 0; i--) {
    fp_mul(F, &table[i], &table[i - 1], &e2);
    fp_mul(F, &e2, &e2, &V[i].y);
    fp_dbl(F, &e2, &e2);
  }
  table[0] = e2;
  for (int i = 0; i < n; i++) {
    fp_sqr(F, &e2, &V[i].x);
    fp_dbl(F, &e1, &e2); fp_add(F, &e2, &e2, &e1);   /* element_mul_si(e2,e2,3) */
    fp_add(F, &e2, &e2, ca);
    fp_mul(F, &e2, &e2, &table[i]);
    fp_dbl(F, &e1, &V[i].x);
    fp_sqr(F, &e0, &e2);
    fp_sub(F, &e0, &e0, &e1);
    fp_sub(F, &e1, &V[i].x, &e0);
    fp_mul(F, &e1, &e1, &e2);
    fp_sub(F, &e1, &e1, &V[i].y);
    V[i].x = e0; V[i].y = e1;
  }
}

/* a_pairings_affine (a_param.c:1283-1383): default Type-A prod_pairings (:1444) */
static void a_pairings_affine(const oracle_pairing *P, fe2 *out, const pt *in1, const pt *in2, int k) {
  const fpctx *F = &P->Fq;
  pt *V = malloc(sizeof(pt) * k), *V1 = malloc(sizeof(pt) * k);
  fe *table = malloc(sizeof(fe) * k);
  fe a, b, c;
  fe2 f, f0, f1;
  int i, j, n;
  for (j = 0; j < k; j++) V[j] = in1[j];
  f.x = F->R; f.y = F->zero;
#define DO_TANGENTS() for (j = 0; j < k; j++) { a_abc_tangent(F, &a, &b, &c, &V[j].x, &V[j].y); \
    a_evalfn(F, &f0, &a, &b, &c, &in2[j].x, &in2[j].y); fi_mul(F, &f, &f, &f0); }
  n = P->exp1;
  for (i = 0; i < n; i++) { fi_sqr(F, &f, &f); DO_TANGENTS(); a_multi_double(F, &P->ca, V, k, table); }
  if (P->sign1 < 0) {
    for (j = 0; j < k; j++) { V1[j] = V[j]; fp_neg(F, &V1[j].y, &V[j].y); }
    fi_inv(F, &f1, &f);
  } else {
    for (j = 0; j < k; j++) V1[j] = V[j];
    f1 = f;
  }
  n = P->exp2;
  for (; i < n; i++) { fi_sqr(F, &f, &f); DO_TANGENTS(); a_multi_double(F, &P->ca, V, k, table); }
  fi_mul(F, &f, &f, &f1);
  for (j = 0; j < k; j++) {
    a_abc_line(F, &a, &b, &c, &V[j].x, &V[j].y, &V1[j].x, &V1[j].y);
    a_evalfn(F, &f0, &a, &b, &c, &in2[j].x, &in2[j].y);
    fi_mul(F, &f, &f, &f0);
  }
  a_tateexp(F, out, &f, &P->h);
#undef DO_TANGENTS
  free(V); free(V1); free(table);
}

/* GT serialisation: fq_to_bytes x||y (fieldquadratic.c:323-329); GT identity "0" == 1
 * (ecc/pairing.c:135-283 mulg wrapper). */
static void gt_one_bytes(const oracle_pairing *P, uint8_t *out) {
  memset(out, 0, P->lenT);
  if (P->type == 'a') out[P->Fq.nbytes - 1] = 1;
}

int oracle_pairing_batch(const oracle_pairing *P, const uint8_t *g1, const uint8_t *g2,
                         uint8_t *gt, size_t n) {
  const fpctx *F = &P->Fq;
  if (P->type != 'a') return 1;
  for (size_t u = 0; u < n; u++) {
    pt A, B; fe2 o;
    pt_from_bytes(F, &P->ca, &P->cb, &A, g1 + u * P->len1);
    pt_from_bytes(F, &P->ca, &P->cb, &B, g2 + u * P->len2);
    uint8_t *ob = gt + u * P->lenT;
    /* pairing_apply identity short-circuit (include/pbc_pairing.h:123-130) */
    if (A.inf || B.inf) { gt_one_bytes(P, ob); continue; }
    a_pairing_proj(P, &o, &A, &B);
    fp_to_bytes(F, ob, &o.x);
    fp_to_bytes(F, ob + F->nbytes, &o.y);
  }
  return 0;
}

int oracle_prod_pairing_batch(const oracle_pairing *P, const uint8_t *g1, const uint8_t *g2,
                              uint8_t *gt, size_t n, int k) {
  const fpctx *F = &P->Fq;
  if (P->type != 'a' || k < 1) return 1;
  pt *A = malloc(sizeof(pt) * k), *B = malloc(sizeof(pt) * k);
  for (size_t u = 0; u < n; u++) {
    int ident = 0;
    for (int j = 0; j < k; j++) {
      pt_from_bytes(F, &P->ca, &P->cb, &A[j], g1 + (u * k + j) * P->len1);
      pt_from_bytes(F, &P->ca, &P->cb, &B[j], g2 + (u * k + j) * P->len2);
      if (A[j].inf || B[j].inf) ident = 1;
    }
    uint8_t *ob = gt + u * P->lenT;
    /* element_prod_pairing: ANY identity input -> whole product = 1 (pbc_pairing.h:161-168) */
    if (ident) { gt_one_bytes(P, ob); continue; }
    fe2 o;
    a_pairings_affine(P, &o, A, B, k);
    fp_to_bytes(F, ob, &o.x);
    fp_to_bytes(F, ob + F->nbytes, &o.y);
  }
  free(A); free(B);
  return 0;
}

int oracle_fq_op(const oracle_pairing *P, int op, const uint8_t *a, const uint8_t *b,
                 uint8_t *c, size_t n) {
  const fpctx *F = &P->Fq;
  int L = F->nbytes;
  for (size_t i = 0; i < n; i++) {
    fe x, y, z;
    fp_from_bytes(F, &x, a + i * L);
    if (b) fp_from_bytes(F, &y, b + i * L); else y = F->zero;
    switch (op) {
      case 0: fp_mul(F, &z, &x, &y); break;
      case 1: fp_add(F, &z, &x, &y); break;
      case 2: fp_sub(F, &z, &x, &y); break;
      case 3: fp_inv(F, &z, &x); break;
      case 4: fp_neg(F, &z, &x); break;
      case 5: fp_halve(F, &z, &x); break;
      case 6: fp_dbl(F, &z, &x); break;
      default: return 1;
    }
    fp_to_bytes(F, c + i * L, &z);
  }
  return 0;
}

int oracle_gt_mul(const oracle_pairing *P, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n) {
  const fpctx *F = &P->Fq;
  if (P->type != 'a') return 1;
  int L = F->nbytes;
  for (size_t i = 0; i < n; i++) {
    fe2 x, y, z;
    fp_from_bytes(F, &x.x, a + i * 2 * L); fp_from_bytes(F, &x.y, a + i * 2 * L + L);
    fp_from_bytes(F, &y.x, b + i * 2 * L); fp_from_bytes(F, &y.y, b + i * 2 * L + L);
    fi_mul(F, &z, &x, &y);
    fp_to_bytes(F, out + i * 2 * L, &z.x); fp_to_bytes(F, out + i * 2 * L + L, &z.y);
  }
  return 0;
}

int oracle_gt_pow(const oracle_pairing *P, const uint8_t *a, const uint8_t *e, size_t elen,
                  uint8_t *out, size_t n) {
  const fpctx *F = &P->Fq;
  if (P->type != 'a') return 1;
  int L = F->nbytes;
  for (size_t i = 0; i < n; i++) {
    fe2 x, r; big ex;
    big_from_be(&ex, e + i * elen, elen);
    fp_from_bytes(F, &x.x, a + i * 2 * L); fp_from_bytes(F, &x.y, a + i * 2 * L + L);
    r.x = F->R; r.y = F->zero;
    for (int bi = big_bits(&ex) - 1; bi >= 0; bi--) {
      fi_sqr(F, &r, &r);
      if (big_bit(&ex, bi)) fi_mul(F, &r, &r, &x);
    }
    fp_to_bytes(F, out + i * 2 * L, &r.x); fp_to_bytes(F, out + i * 2 * L + L, &r.y);
  }
  return 0;
}

int oracle_g_mul(const oracle_pairing *P, int group, const uint8_t *ptb, const uint8_t *e,
                 size_t elen, uint8_t *out, size_t n) {
  const fpctx *F = &P->Fq;
  if (P->type != 'a') return 1;
  (void) group;                        /* Type A: G1 == G2 == E(Fq) */
  for (size_t i = 0; i < n; i++) {
    pt A, R; big ex;
    big_from_be(&ex, e + i * elen, elen);
    pt_from_bytes(F, &P->ca, &P->cb, &A, ptb + i * P->len1);
    pt_mul(F, &P->ca, &R, &A, &ex);
    pt_to_bytes(F, out + i * P->len1, &R);
  }
  return 0;
}
