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
#define MAXL 17                      /* 1088-bit moduli at most (Type A1 a1.param: 1033-bit p) */

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
  pt eR;                 /* type E: the auxiliary point R of e_pairing (e_param.c:472-483) */
  big phikonr;           /* type E: (q - 1)/r (e_param.c:857-860) */
  int sign0;             /* type E: r = 2^exp2 + sign1 2^exp1 + sign0 (e_param.c:23-26) */
  int a1;                /* type A1 (ecc/a_param.c:1564-2321): type 'a' with r = n composite, h = l */
  fe ca, cb;             /* curve a, b in Fq (A: a=1, b=0, a_param.c:1450-1452) */
  struct dctx *D;        /* type D (ecc/d_param.c:40-51) */
  struct fctx *Fx;       /* type F (ecc/f_param.c:35-45) */
};
static int init_d(oracle_pairing *P, const char *txt, size_t len);
static int init_e(oracle_pairing *P, const char *txt, size_t len);
static int big_divexact(big *quo, const big *z, const big *d);
static int e_pairing_bytes(const oracle_pairing *P, const uint8_t *g1, const uint8_t *g2, uint8_t *gt, int k);
static int init_f(oracle_pairing *P, const char *txt, size_t len);
static int d_pairing_bytes(const oracle_pairing *P, const uint8_t *g1, const uint8_t *g2, uint8_t *gt, int k);
static int f_pairing_bytes(const oracle_pairing *P, const uint8_t *g1, const uint8_t *g2, uint8_t *gt, int k);
static int df_gt_mul(const oracle_pairing *P, const uint8_t *a, const uint8_t *b, uint8_t *out);
static int df_gt_pow(const oracle_pairing *P, const uint8_t *a, const big *e, uint8_t *out);

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

/* a1_init_pairing (ecc/a_param.c:2230-2273) + pbc_param_init_a1 (:2289-2298): y^2 = x^3 + x over F_p,
 * group order r = n (composite), cofactor phikonr = l = (p + 1)/n */
static int init_a1(oracle_pairing *P, const char *txt, size_t len) {
  if (kv_big(txt, len, "p", &P->q) || kv_big(txt, len, "n", &P->r) || kv_big(txt, len, "l", &P->h)) return 1;
  if (fp_init(&P->Fq, &P->q)) return 1;
  P->a1 = 1;
  P->ca = P->Fq.R;
  P->cb = P->Fq.zero;
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
  else if (!strcmp(tb, "a1")) rc = init_a1(P, txt, len);
  else if (!strcmp(tb, "e")) rc = init_e(P, txt, len);
  else if (!strcmp(tb, "d") || !strcmp(tb, "g")) rc = init_d(P, txt, len);
  else if (!strcmp(tb, "f")) rc = init_f(P, txt, len);
  if (rc) { free(P); return 1; }
  *out = P;
  return 0;
}
void oracle_pairing_clear(oracle_pairing *p) { if (p) { free(p->D); free(p->Fx); } free(p); }
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

/* compute_abc_line_proj (a_param.c:1820-1837): chord through the Jacobian V and the affine V1 */
static void a1_abc_line_proj(const fpctx *F, fe *a, fe *b, fe *c, const fe *Vx, const fe *Vy,
                             const fe *z, const fe *z2, const fe *V1x, const fe *V1y) {
  fe e0;
  fp_mul(F, c, z, z2);
  fp_mul(F, &e0, V1y, c);
  fp_sub(F, a, Vy, &e0);
  fp_mul(F, b, c, V1x);
  fp_mul(F, &e0, Vx, z);
  fp_sub(F, b, b, &e0);
  fp_mul(F, c, b, V1y);
  fp_mul(F, &e0, a, V1x);
  fp_add(F, c, c, &e0);
  fp_neg(F, c, c);
}
/* the closing lines of a1_pairing_proj / a1_pairings_affine (a_param.c:1986-1994, :2167-2175):
 * f^(p-1) = conj(f)/f, then element_pow_mpz by phikonr = l (generic_pow_mpz field.c:113-126) */
static void a1_tateexp(const fpctx *F, fe2 *out, fe2 *f, const big *l) {
  fe2 f0, acc;
  fi_inv(F, &f0, f);
  fp_neg(F, &f->y, &f->y);
  fi_mul(F, f, f, &f0);
  acc.x = F->R; acc.y = F->zero;
  for (int i = big_bits(l) - 1; i >= 0; i--) {
    fi_sqr(F, &acc, &acc);
    if (big_bit(l, i)) fi_mul(F, &acc, &acc, f);
  }
  *out = acc;
}
/* a1_pairing_proj (a_param.c:1840-2015): the default Type-A1 map (:2261): plain double-and-add
 * over the bits of n in Jacobian coordinates, mixed addition of the affine in1 */
static void a1_pairing_proj(const oracle_pairing *P, fe2 *out, const pt *in1, const pt *in2) {
  const fpctx *F = &P->Fq;
  fe Vx = in1->x, Vy = in1->y, z = F->R, z2 = F->R;
  const fe *Px = &in1->x, *Py = &in1->y, *Qx = &in2->x, *Qy = &in2->y;
  fe a, b, c, e0;
  fe2 f, f0;
  f.x = F->R; f.y = F->zero;
  int m = big_bits(&P->r);
  m = m > 2 ? m - 2 : 0;
  for (;;) {
    a_abc_tangent_proj(F, &a, &b, &c, &Vx, &Vy, &z, &z2);
    a_evalfn(F, &f0, &a, &b, &c, Qx, Qy);
    fi_mul(F, &f, &f, &f0);
    if (!m) break;
    { /* proj_double (a_param.c:1900-1938) */
      fe *e1 = &a, *e2 = &b, *e3 = &c;
      fp_sqr(F, &e0, &Vx); fp_dbl(F, e1, &e0); fp_add(F, &e0, e1, &e0); fp_sqr(F, e1, &z2); fp_add(F, &e0, &e0, e1);
      fp_mul(F, &z, &Vy, &z); fp_dbl(F, &z, &z); fp_sqr(F, &z2, &z);
      fp_sqr(F, e2, &Vy); fp_mul(F, e1, &Vx, e2); fp_dbl(F, e1, e1); fp_dbl(F, e1, e1);
      fp_dbl(F, e3, e1); fp_sqr(F, &Vx, &e0); fp_sub(F, &Vx, &Vx, e3);
      fp_sqr(F, e2, e2); fp_dbl(F, e2, e2); fp_dbl(F, e2, e2); fp_dbl(F, e2, e2);
      fp_sub(F, e1, e1, &Vx); fp_mul(F, &e0, &e0, e1); fp_sub(F, &Vy, &e0, e2);
    }
    if (big_bit(&P->r, m)) {
      a1_abc_line_proj(F, &a, &b, &c, &Vx, &Vy, &z, &z2, Px, Py);
      a_evalfn(F, &f0, &a, &b, &c, Qx, Qy);
      fi_mul(F, &f, &f, &f0);
      { /* proj_add (a_param.c:1868-1898) */
        fe *e1 = &a, *e2 = &b, *e3 = &c;
        fp_mul(F, &e0, Px, &z2); fp_sub(F, &e0, &e0, &Vx);
        fp_sqr(F, e1, &e0);
        fp_mul(F, e2, &z, &z2); fp_mul(F, e2, e2, Py); fp_sub(F, e2, e2, &Vy);
        z2 = Vx;
        fp_sqr(F, &Vx, e2);
        fp_mul(F, e3, &e0, e1);
        fp_sub(F, &Vx, &Vx, e3);
        fp_dbl(F, e3, &z2); fp_mul(F, e3, e3, e1);
        fp_sub(F, &Vx, &Vx, e3);
        fp_mul(F, e3, &z2, e1); fp_sub(F, e3, e3, &Vx); fp_mul(F, e3, e3, e2);
        fp_mul(F, e2, &e0, e1); fp_mul(F, e2, e2, &Vy);
        fp_sub(F, &Vy, e3, e2);
        fp_mul(F, &z, &z, &e0);
        fp_sqr(F, &z2, &z);
      }
    }
    m--;
    fi_sqr(F, &f, &f);
  }
  a1_tateexp(F, out, &f, &P->h);
}
/* a1_pairings_affine (a_param.c:2100-2193): the Type-A1 prod_pairings (:2263): shared f, affine
 * tangents and chords per term; element_multi_double / element_multi_add (curve.c:210-379) batch
 * the inversions of the k affine doublings / additions -- the points are the same */
static void a1_pairings_affine(const oracle_pairing *P, fe2 *out, const pt *in1, const pt *in2, int k) {
  const fpctx *F = &P->Fq;
  pt *V = malloc(sizeof(pt) * k);
  fe a, b, c;
  fe2 f, f0;
  for (int j = 0; j < k; j++) V[j] = in1[j];
  f.x = F->R; f.y = F->zero;
  int m = big_bits(&P->r);
  m = m > 2 ? m - 2 : 0;
  for (;;) {
    for (int j = 0; j < k; j++) {
      a_abc_tangent(F, &a, &b, &c, &V[j].x, &V[j].y);
      a_evalfn(F, &f0, &a, &b, &c, &in2[j].x, &in2[j].y);
      fi_mul(F, &f, &f, &f0);
    }
    if (!m) break;
    for (int j = 0; j < k; j++) pt_dbl(F, &P->ca, &V[j], &V[j]);
    if (big_bit(&P->r, m)) {
      for (int j = 0; j < k; j++) {
        a_abc_line(F, &a, &b, &c, &V[j].x, &V[j].y, &in1[j].x, &in1[j].y);
        a_evalfn(F, &f0, &a, &b, &c, &in2[j].x, &in2[j].y);
        fi_mul(F, &f, &f, &f0);
      }
      for (int j = 0; j < k; j++) pt_add(F, &P->ca, &V[j], &V[j], &in1[j]);
    }
    m--;
    fi_sqr(F, &f, &f);
  }
  a1_tateexp(F, out, &f, &P->h);
  free(V);
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
  out[P->Fq.nbytes - 1] = 1;          /* the first F_q coordinate is the constant term for every type */
}

int oracle_pairing_batch(const oracle_pairing *P, const uint8_t *g1, const uint8_t *g2,
                         uint8_t *gt, size_t n) {
  const fpctx *F = &P->Fq;
  if (P->type == 'd' || P->type == 'g' || P->type == 'f') {
    for (size_t u = 0; u < n; u++) {
      int rc = P->type != 'f' ? d_pairing_bytes(P, g1 + u * P->len1, g2 + u * P->len2, gt + u * P->lenT, 1)
                              : f_pairing_bytes(P, g1 + u * P->len1, g2 + u * P->len2, gt + u * P->lenT, 1);
      if (rc) return rc;
    }
    return 0;
  }
  if (P->type == 'e') {
    for (size_t u = 0; u < n; u++)
      if (e_pairing_bytes(P, g1 + u * P->len1, g2 + u * P->len2, gt + u * P->lenT, 1)) return 1;
    return 0;
  }
  if (P->type != 'a') return 1;
  for (size_t u = 0; u < n; u++) {
    pt A, B; fe2 o;
    pt_from_bytes(F, &P->ca, &P->cb, &A, g1 + u * P->len1);
    pt_from_bytes(F, &P->ca, &P->cb, &B, g2 + u * P->len2);
    uint8_t *ob = gt + u * P->lenT;
    /* pairing_apply identity short-circuit (include/pbc_pairing.h:123-130).  A first argument of order 2 (y = 0: the
     * zero-filled record (0, 0)) makes the reference divide by zero (point_to_affine, a_param.c:1073-1080); its
     * pairing value is 1 (2 is coprime to the group order), which is what is returned here. */
    if (A.inf || B.inf || fp_is0(F, &A.y)) { gt_one_bytes(P, ob); continue; }
    if (P->a1) a1_pairing_proj(P, &o, &A, &B); else a_pairing_proj(P, &o, &A, &B);
    fp_to_bytes(F, ob, &o.x);
    fp_to_bytes(F, ob + F->nbytes, &o.y);
  }
  return 0;
}

int oracle_prod_pairing_batch(const oracle_pairing *P, const uint8_t *g1, const uint8_t *g2,
                              uint8_t *gt, size_t n, int k) {
  const fpctx *F = &P->Fq;
  if (k < 1) return 1;
  if (P->type == 'd' || P->type == 'g' || P->type == 'f') {
    for (size_t u = 0; u < n; u++) {
      int rc = P->type != 'f' ? d_pairing_bytes(P, g1 + u * k * P->len1, g2 + u * k * P->len2, gt + u * P->lenT, k)
                              : f_pairing_bytes(P, g1 + u * k * P->len1, g2 + u * k * P->len2, gt + u * P->lenT, k);
      if (rc) return rc;
    }
    return 0;
  }
  if (P->type == 'e') {
    for (size_t u = 0; u < n; u++)
      if (e_pairing_bytes(P, g1 + u * k * P->len1, g2 + u * k * P->len2, gt + u * P->lenT, k)) return 1;
    return 0;
  }
  if (P->type != 'a') return 1;
  pt *A = malloc(sizeof(pt) * k), *B = malloc(sizeof(pt) * k);
  for (size_t u = 0; u < n; u++) {
    int ident = 0;
    for (int j = 0; j < k; j++) {
      pt_from_bytes(F, &P->ca, &P->cb, &A[j], g1 + (u * k + j) * P->len1);
      pt_from_bytes(F, &P->ca, &P->cb, &B[j], g2 + (u * k + j) * P->len2);
      if (A[j].inf || B[j].inf || fp_is0(F, &A[j].y)) ident = 1;     /* y = 0: see oracle_pairing_batch */
    }
    uint8_t *ob = gt + u * P->lenT;
    /* element_prod_pairing: ANY identity input -> whole product = 1 (pbc_pairing.h:161-168) */
    if (ident) { gt_one_bytes(P, ob); continue; }
    fe2 o;
    if (P->a1) a1_pairings_affine(P, &o, A, B, k); else a_pairings_affine(P, &o, A, B, k);
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
  if (P->type == 'd' || P->type == 'g' || P->type == 'f') {
    for (size_t i = 0; i < n; i++)
      if (df_gt_mul(P, a + i * P->lenT, b + i * P->lenT, out + i * P->lenT)) return 1;
    return 0;
  }
  if (P->type == 'e') {                 /* GT = F_q (pairing_GT_init(pairing, p->Fq), e_param.c:863) */
    for (size_t i = 0; i < n; i++) {
      fe x, y;
      fp_from_bytes(F, &x, a + i * P->lenT); fp_from_bytes(F, &y, b + i * P->lenT);
      fp_mul(F, &x, &x, &y);
      fp_to_bytes(F, out + i * P->lenT, &x);
    }
    return 0;
  }
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
  if (P->type == 'd' || P->type == 'g' || P->type == 'f') {
    for (size_t i = 0; i < n; i++) {
      big ex; big_from_be(&ex, e + i * elen, elen);
      if (df_gt_pow(P, a + i * P->lenT, &ex, out + i * P->lenT)) return 1;
    }
    return 0;
  }
  if (P->type == 'e') {
    for (size_t i = 0; i < n; i++) {
      fe x; big ex;
      big_from_be(&ex, e + i * elen, elen);
      fp_from_bytes(F, &x, a + i * P->lenT);
      fp_pow(F, &x, &x, &ex);
      fp_to_bytes(F, out + i * P->lenT, &x);
    }
    return 0;
  }
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
  if (P->type != 'a' && P->type != 'e' && group != 1) return 1;   /* D/F/G: only G1 = E(Fq) scalar mult is provided */
  for (size_t i = 0; i < n; i++) {
    pt A, R; big ex;
    big_from_be(&ex, e + i * elen, elen);
    pt_from_bytes(F, &P->ca, &P->cb, &A, ptb + i * P->len1);
    pt_mul(F, &P->ca, &R, &A, &ex);
    pt_to_bytes(F, out + i * P->len1, &R);
  }
  return 0;
}

static int fp_sqrt_ts(const fpctx *F, fe *out, const fe *a, const big *q);

/* ---- element_from_hash and the point formats on G1 = E(F_q) ---------------------------------------- */
/* fp_from_hash (arith/montfp.c:441-449) over pbc_mpz_from_hash (arith/field.c:643-668): the digest is laid
 * out as H || 0 || H || 1 || ... up to the byte length of q, read big-endian, halved while it exceeds q. */
static void fp_from_hash(const oracle_pairing *P, fe *x, const uint8_t *data, int len) {
  const fpctx *F = &P->Fq;
  uint8_t buf[8 * MAXL];
  int count = F->nbytes, i = 0, done = 0;
  uint8_t counter = 0;
  for (;;) {
    int n;
    if (len >= count - i) { n = count - i; done = 1; } else n = len;
    memcpy(buf + i, data, (size_t) n);
    i += n;
    if (done) break;
    buf[i++] = counter++;
    if (i == count) break;
  }
  big z;
  big_from_be(&z, buf, (size_t) count);
  while (bn_cmp(z.v, P->q.v, BIGL) > 0) {
    for (int w = 0; w < BIGL - 1; w++) z.v[w] = (z.v[w] >> 1) | (z.v[w + 1] << 63);
    z.v[BIGL - 1] >>= 1;
  }
  fe t; memset(&t, 0, sizeof t);
  memcpy(t.v, z.v, 8 * (size_t) F->n);
  fp_mul(F, x, &t, &F->R2);            /* fp_set_mpz: z = q becomes 0 */
}
/* fp_sgn_odd (montfp.c:460-472): 0 for 0, +1 when the canonical residue is odd, else -1 */
static int fp_sgn(const fpctx *F, const fe *a) {
  uint8_t b[8 * MAXL];
  if (fp_is0(F, a)) return 0;
  fp_to_bytes(F, b, a);
  return (b[F->nbytes - 1] & 1) ? 1 : -1;
}
/* point_from_x (ecc/curve.c:778-791): y = sqrt(x^3 + a x + b), "requires a solution to exist"; returns 0 if none.
 * element_tonelli's root (arith/field.c:672-720) is x^((q+1)/4) when q = 3 mod 4; otherwise it depends on the
 * reference's random non-residue and callers that do not fix the sign get either root. */
static int pt_from_x(const oracle_pairing *P, pt *R, const fe *x) {
  const fpctx *F = &P->Fq;
  fe t;
  fp_sqr(F, &t, x);
  fp_add(F, &t, &t, &P->ca);
  fp_mul(F, &t, &t, x);
  fp_add(F, &t, &t, &P->cb);
  R->inf = 0;
  R->x = *x;
  if (fp_is0(F, &t)) { R->y = F->zero; return 1; }
  return fp_sqrt_ts(F, &R->y, &t, &P->q);
}
/* curve_from_hash (ecc/curve.c:455-482); G1 of every type, G2 of the symmetric ones */
int oracle_from_hash(const oracle_pairing *P, const uint8_t *data, int hlen, uint8_t *out, size_t n) {
  const fpctx *F = &P->Fq;
  for (size_t i = 0; i < n; i++) {
    fe x;
    pt A, R;
    fp_from_hash(P, &x, data + i * (size_t) hlen, hlen);
    while (!pt_from_x(P, &A, &x)) {     /* x <- x^2 + 1 until the right-hand side is a square */
      fp_sqr(F, &x, &x);
      fp_add(F, &x, &x, &F->R);
    }
    if (fp_sgn(F, &A.y) < 0) fp_neg(F, &A.y, &A.y);
    if (big_bits(&P->h)) pt_mul(F, &P->ca, &R, &A, &P->h); else R = A;   /* cofactor (curve.c:477) */
    pt_to_bytes(F, out + i * (size_t) P->len1, &R);
  }
  return 0;
}
/* what 0: element_to_bytes_compressed (curve.c:762-773), 1: element_from_bytes_compressed (:799-813),
 * 2: element_to_bytes_x_only (:821-827), 3: element_from_bytes_x_only (:829-836; the root as point_from_x
 * leaves it -- compare up to sign when q = 1 mod 4).  An x without a point gives the zero record. */
int oracle_point_format(const oracle_pairing *P, int what, const uint8_t *in, uint8_t *out, size_t n) {
  const fpctx *F = &P->Fq;
  const size_t fb = (size_t) F->nbytes, lp = 2 * fb;
  for (size_t i = 0; i < n; i++) {
    if (what == 0 || what == 2) {
      const uint8_t *pnt = in + i * lp;
      uint8_t *o = out + i * (fb + (what == 0));
      memcpy(o, pnt, fb);
      if (what == 0) { fe y; fp_from_bytes(F, &y, pnt + fb); o[fb] = fp_sgn(F, &y) > 0; }
    } else {
      const size_t li = fb + (what == 1);
      fe x;
      pt A;
      fp_from_bytes(F, &x, in + i * li);
      if (!pt_from_x(P, &A, &x)) { memset(out + i * lp, 0, lp); continue; }
      if (what == 1) {
        const int sg = fp_sgn(F, &A.y);
        if (in[i * li + fb] ? sg < 0 : sg > 0) fp_neg(F, &A.y, &A.y);
      }
      pt_to_bytes(F, out + i * lp, &A);
    }
  }
  return 0;
}

/* ================================================================== */
/* Type E (k = 1, ordinary curve over a 1020-bit F_q), ecc/e_param.c   */
/* ================================================================== */
/* square root in F_q by Tonelli-Shanks (the reference draws its auxiliary point with
 * curve_set_gen_no_cofac, e_param.c:869-870: a RANDOM point; the pairing value does not depend on
 * it -- the Tate pairing f_P((Q+R) - (R)) is independent of R -- so the oracle takes the curve point
 * with the smallest x >= 1).  Returns 0 when a is not a square. */
static int fp_sqrt_ts(const fpctx *F, fe *out, const fe *a, const big *q) {
  big t = *q, e;
  t.v[0] &= ~1ull;                                   /* q - 1 */
  int s = 0;
  while (!big_bit(&t, 0)) { for (int w = 0; w < BIGL - 1; w++) t.v[w] = (t.v[w] >> 1) | (t.v[w + 1] << 63); t.v[BIGL - 1] >>= 1; s++; }
  big half = *q; half.v[0] &= ~1ull;
  for (int w = 0; w < BIGL - 1; w++) half.v[w] = (half.v[w] >> 1) | (half.v[w + 1] << 63);
  half.v[BIGL - 1] >>= 1;                            /* (q - 1)/2 */
  fe chk; fp_pow(F, &chk, a, &half);
  if (!fp_eq(F, &chk, &F->R)) return 0;
  fe z, c, x, b, tt;                                 /* non-residue z */
  for (uint64_t k = 2;; k++) { fp_set_ui(F, &z, k); fp_pow(F, &chk, &z, &half); if (!fp_eq(F, &chk, &F->R)) break; }
  fp_pow(F, &c, &z, &t);
  e = t; { big one; memset(&one, 0, sizeof one); one.v[0] = 1; bn_add(e.v, e.v, one.v, BIGL); }
  for (int w = 0; w < BIGL - 1; w++) e.v[w] = (e.v[w] >> 1) | (e.v[w + 1] << 63);
  e.v[BIGL - 1] >>= 1;                               /* (t + 1)/2 */
  fp_pow(F, &x, a, &e);
  fp_pow(F, &b, a, &t);
  int m = s;
  while (!fp_eq(F, &b, &F->R)) {
    int i = 0; tt = b;
    while (!fp_eq(F, &tt, &F->R)) { fp_sqr(F, &tt, &tt); i++; }
    fe g = c;
    for (int j = 0; j < m - i - 1; j++) fp_sqr(F, &g, &g);
    fp_mul(F, &x, &x, &g);
    fp_sqr(F, &c, &g);
    fp_mul(F, &b, &b, &c);
    m = i;
  }
  *out = x;
  return 1;
}
/* e_miller_affine (e_param.c:302-466): numerator v and denominator vd of f_P(Q+R)/f_P(R), Solinas
 * r = 2^exp2 + sign1 2^exp1 + sign0.  (The default e_miller_proj, :64-300, is the same function in
 * Jacobian coordinates with the same collated divisions.) */
static void e_miller(const oracle_pairing *P, fe *res, const pt *Pp, const pt *QR, const pt *R) {
  const fpctx *F = &P->Fq;
  fe v = F->R, vd = F->R, v1, vd1, a, b, c, e0, e1;
  pt Z = *Pp, Z1;
  const fe *numx = &QR->x, *numy = &QR->y, *denx = &R->x, *deny = &R->y;
#define DO_VERTICAL(e, ed, Ax) do { fp_sub(F, &e0, numx, (Ax)); fp_mul(F, &(e), &(e), &e0); \
    fp_sub(F, &e0, denx, (Ax)); fp_mul(F, &(ed), &(ed), &e0); } while (0)
#define EVAL(e, ed) do { fp_mul(F, &e0, &a, numx); fp_mul(F, &e1, &b, numy); fp_add(F, &e0, &e0, &e1); fp_add(F, &e0, &e0, &c); \
    fp_mul(F, &(e), &(e), &e0); fp_mul(F, &e0, &a, denx); fp_mul(F, &e1, &b, deny); fp_add(F, &e0, &e0, &e1); \
    fp_add(F, &e0, &e0, &c); fp_mul(F, &(ed), &(ed), &e0); } while (0)
#define DO_TANGENT(e, ed) do { fp_sqr(F, &a, &Z.x); fp_dbl(F, &e0, &a); fp_add(F, &a, &a, &e0); fp_add(F, &a, &a, &P->ca); \
    fp_neg(F, &a, &a); fp_add(F, &b, &Z.y, &Z.y); fp_mul(F, &e0, &b, &Z.y); fp_mul(F, &c, &a, &Z.x); \
    fp_add(F, &c, &c, &e0); fp_neg(F, &c, &c); EVAL(e, ed); } while (0)
  int i, n = P->exp1;
  for (i = 0; i < n; i++) {
    fp_sqr(F, &v, &v); fp_sqr(F, &vd, &vd);
    DO_TANGENT(v, vd);
    pt_dbl(F, &P->ca, &Z, &Z);
    DO_VERTICAL(vd, v, &Z.x);
  }
  if (P->sign1 < 0) {
    v1 = vd; vd1 = v;
    DO_VERTICAL(vd1, v1, &Z.x);
    Z1 = Z; fp_neg(F, &Z1.y, &Z.y);
  } else { v1 = v; vd1 = vd; Z1 = Z; }
  n = P->exp2;
  for (; i < n; i++) {
    fp_sqr(F, &v, &v); fp_sqr(F, &vd, &vd);
    DO_TANGENT(v, vd);
    pt_dbl(F, &P->ca, &Z, &Z);
    DO_VERTICAL(vd, v, &Z.x);
  }
  fp_mul(F, &v, &v, &v1);
  fp_mul(F, &vd, &vd, &vd1);
  /* do_line(v, vd, Z, Z1) */
  fp_sub(F, &b, &Z1.x, &Z.x);
  fp_sub(F, &a, &Z.y, &Z1.y);
  fp_mul(F, &c, &Z.x, &Z1.y);
  fp_mul(F, &e0, &Z.y, &Z1.x);
  fp_sub(F, &c, &c, &e0);
  EVAL(v, vd);
  pt_add(F, &P->ca, &Z, &Z, &Z1);
  DO_VERTICAL(vd, v, &Z.x);
  if (P->sign0 > 0) DO_VERTICAL(v, vd, &Pp->x);
  fp_inv(F, &vd, &vd);
  fp_mul(F, res, &v, &vd);
#undef DO_VERTICAL
#undef EVAL
#undef DO_TANGENT
}
/* e_pairing (e_param.c:472-483): QR = Q + R, Miller, power by phikonr = (q-1)/r.  Type e has no
 * product routine: generic_prod_pairings (pairing.c:35-46) multiplies the k pairings. */
static int e_pairing_bytes(const oracle_pairing *P, const uint8_t *g1, const uint8_t *g2, uint8_t *gt, int k) {
  const fpctx *F = &P->Fq;
  fe acc = F->R;
  int ident = 0;
  for (int j = 0; j < k; j++) {
    pt A, B, QR; fe m;
    pt_from_bytes(F, &P->ca, &P->cb, &A, g1 + (size_t) j * P->len1);
    pt_from_bytes(F, &P->ca, &P->cb, &B, g2 + (size_t) j * P->len2);
    if (A.inf || B.inf) { ident = 1; continue; }
    pt_add(F, &P->ca, &QR, &B, &P->eR);
    e_miller(P, &m, &A, &QR, &P->eR);
    fp_pow(F, &m, &m, &P->phikonr);
    fp_mul(F, &acc, &acc, &m);
  }
  if (ident) { gt_one_bytes(P, gt); return 0; }
  fp_to_bytes(F, gt, &acc);
  return 0;
}
/* e_init_pairing (e_param.c:832-872) + pbc_param_init_e (:891-906) */
static int init_e(oracle_pairing *P, const char *txt, size_t len) {
  big a, b;
  if (kv_big(txt, len, "q", &P->q) || kv_big(txt, len, "r", &P->r) || kv_big(txt, len, "h", &P->h) ||
      kv_big(txt, len, "a", &a) || kv_big(txt, len, "b", &b)) return 1;
  if (kv_int(txt, len, "exp2", &P->exp2) || kv_int(txt, len, "exp1", &P->exp1) ||
      kv_int(txt, len, "sign1", &P->sign1) || kv_int(txt, len, "sign0", &P->sign0)) return 1;
  if (fp_init(&P->Fq, &P->q)) return 1;
  const fpctx *F = &P->Fq;
#define SETBIG(dst, src) do { fe t_; memset(&t_, 0, sizeof t_); memcpy(t_.v, (src).v, 8 * F->n); fp_mul(F, &(dst), &t_, &F->R2); } while (0)
  SETBIG(P->ca, a); SETBIG(P->cb, b);
#undef SETBIG
  /* phikonr = (q - 1)/r (k = 1) */
  { big qm1 = P->q; qm1.v[0] &= ~1ull; if (big_divexact(&P->phikonr, &qm1, &P->r)) return 1; }
  /* auxiliary point: smallest x >= 1 with x^3 + a x + b a square */
  for (uint64_t x = 1;; x++) {
    fe fx, rhs, y;
    fp_set_ui(F, &fx, x);
    fp_sqr(F, &rhs, &fx); fp_add(F, &rhs, &rhs, &P->ca); fp_mul(F, &rhs, &rhs, &fx); fp_add(F, &rhs, &rhs, &P->cb);
    if (fp_is0(F, &rhs)) continue;
    if (fp_sqrt_ts(F, &y, &rhs, &P->q)) { P->eR.inf = 0; P->eR.x = fx; P->eR.y = y; break; }
  }
  P->len1 = P->len2 = 2 * F->nbytes;
  P->lenT = F->nbytes;
  return 0;
}

/* ================================================================== */
/* Type D (MNT, k = 6), ecc/d_param.c, and Type G (Freeman, k = 10),   */
/* ecc/g_param.c: the same construction with d = k/2 = 3 or 5          */
/* ================================================================== */
/* Fq^d = Fq[x]/(x^d + c_(d-1) x^(d-1) + ... + c0): polymod ring, arith/poly.c (n = 3 or 5) */
#define MAXD 5
typedef struct { fe c[MAXD]; } fd;
struct dctx {
  int deg;               /* d: 3 for type d (d_param.c:1016-1026), 5 for type g (g_param.c:1269-1281) */
  fd xpwr[MAXD - 1];     /* x^d .. x^(2d-2) mod f: compute_x_powers (poly.c:1302-1333) */
  fe nqr;                /* v: Fq^k = Fq^d[sqrt(v)], v in Fq (d_param.c:1028-1032, g_param.c:1283-1285) */
  fe nqrinv, nqrinv2;    /* v^-1, v^-2 (d_param.c:1072-1075, g_param.c:1322-1325; constants of Fq inside Fq^d) */
  fd xpowq[MAXD - 1];    /* x^q, x^2q, (x^3q, x^4q) (d_param.c:1044-1050, g_param.c:1307-1316) */
  fe ta, tb;             /* twist y^2 = x^3 + a v^2 x + b v^3 (curve.c:885-901) */
  big phikonr;           /* Phi_k(q)/r: (q^2 - q + 1)/r (d_param.c:1036-1042), (q^4 - q^3 + q^2 - q + 1)/r (g_param.c:1288-1305) */
};
#define DEG (P->D->deg)

static void fd_add(const oracle_pairing *P, fd *r, const fd *a, const fd *b) { for (int i = 0; i < DEG; i++) fp_add(&P->Fq, &r->c[i], &a->c[i], &b->c[i]); }
static void fd_sub(const oracle_pairing *P, fd *r, const fd *a, const fd *b) { for (int i = 0; i < DEG; i++) fp_sub(&P->Fq, &r->c[i], &a->c[i], &b->c[i]); }
static void fd_dbl(const oracle_pairing *P, fd *r, const fd *a) { for (int i = 0; i < DEG; i++) fp_dbl(&P->Fq, &r->c[i], &a->c[i]); }
static void fd_neg(const oracle_pairing *P, fd *r, const fd *a) { for (int i = 0; i < DEG; i++) fp_neg(&P->Fq, &r->c[i], &a->c[i]); }
static void fd_halve(const oracle_pairing *P, fd *r, const fd *a) { for (int i = 0; i < DEG; i++) fp_halve(&P->Fq, &r->c[i], &a->c[i]); }
/* polymod_const_mul (poly.c:1550-1558) */
static void fd_mul_fq(const oracle_pairing *P, fd *r, const fd *a, const fe *s) { for (int i = 0; i < DEG; i++) fp_mul(&P->Fq, &r->c[i], &a->c[i], s); }
static int fd_is0(const oracle_pairing *P, const fd *a) { for (int i = 0; i < DEG; i++) if (!fp_is0(&P->Fq, &a->c[i])) return 0; return 1; }
static int fd_eq(const oracle_pairing *P, const fd *a, const fd *b) { for (int i = 0; i < DEG; i++) if (!fp_eq(&P->Fq, &a->c[i], &b->c[i])) return 0; return 1; }
static void fd_set_fq(const oracle_pairing *P, fd *r, const fe *s) { for (int i = 0; i < MAXD; i++) r->c[i] = P->Fq.zero; r->c[0] = *s; }
/* polymod_mul_degree3 (poly.c:910-930) / polymod_mul (poly.c:880-908, the x^i table of
 * compute_x_powers): product mod f; the Karatsuba grouping the reference uses for d = 3
 * gives the same ring element as this schoolbook form. */
static void fd_mul(const oracle_pairing *P, fd *r, const fd *a, const fd *b) {
  const fpctx *F = &P->Fq;
  const int d = DEG;
  fe hi[2 * MAXD - 1], t;
  for (int i = 0; i < 2 * d - 1; i++) hi[i] = F->zero;
  for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) {
    fp_mul(F, &t, &a->c[i], &b->c[j]);
    fp_add(F, &hi[i + j], &hi[i + j], &t);
  }
  fd res, p0;
  fd_set_fq(P, &res, &F->zero);
  for (int i = 0; i < d; i++) res.c[i] = hi[i];
  for (int i = d; i < 2 * d - 1; i++) { fd_mul_fq(P, &p0, &P->D->xpwr[i - d], &hi[i]); fd_add(P, &res, &res, &p0); }
  *r = res;
}
static void fd_sqr(const oracle_pairing *P, fd *r, const fd *a) { fd_mul(P, r, a, a); }  /* poly.c:1049-1089, :1091-1143 */
/* a^q for a in Fq^d: a0 + sum a_i x^(iq) (the qpower macros, d_param.c:507-527, g_param.c:486-518) */
static void fd_frob(const oracle_pairing *P, fd *r, const fd *a) {
  const fpctx *F = &P->Fq;
  fd e2, res;
  fd_mul_fq(P, &res, &P->D->xpowq[0], &a->c[1]);
  for (int i = 2; i < DEG; i++) { fd_mul_fq(P, &e2, &P->D->xpowq[i - 1], &a->c[i]); fd_add(P, &res, &res, &e2); }
  fp_add(F, &res.c[0], &res.c[0], &a->c[0]);
  *r = res;
}
/* polymod_invert (poly.c:521-536) is a polynomial extended Euclid; the inverse is unique,
 * here: a^-1 = (a^q a^(q^2) ... a^(q^(d-1))) / N(a), N(a) in Fq. */
static void fd_inv(const oracle_pairing *P, fd *r, const fd *a) {
  const fpctx *F = &P->Fq;
  fd t, w, n;
  fd_frob(P, &t, a);
  w = t;
  for (int i = 2; i < DEG; i++) { fd_frob(P, &t, &t); fd_mul(P, &w, &w, &t); }
  fd_mul(P, &n, a, &w);
  fe ni; fp_inv(F, &ni, &n.c[0]);
  fd_mul_fq(P, r, &w, &ni);
}
static void fd_pow(const oracle_pairing *P, fd *r, const fd *a, const big *e) {
  const fpctx *F = &P->Fq;
  fd acc, base = *a; fd_set_fq(P, &acc, &F->R);
  for (int i = big_bits(e) - 1; i >= 0; i--) {
    fd_sqr(P, &acc, &acc);
    if (big_bit(e, i)) fd_mul(P, &acc, &acc, &base);
  }
  *r = acc;
}
static void fd_from_bytes(const oracle_pairing *P, fd *r, const uint8_t *b) { fd_set_fq(P, r, &P->Fq.zero); for (int i = 0; i < DEG; i++) fp_from_bytes(&P->Fq, &r->c[i], b + i * P->Fq.nbytes); }  /* poly.c:735-752 */
static void fd_to_bytes(const oracle_pairing *P, uint8_t *b, const fd *a) { for (int i = 0; i < DEG; i++) fp_to_bytes(&P->Fq, b + i * P->Fq.nbytes, &a->c[i]); }    /* poly.c:718-733 */

/* Fq^k = Fq^d[sqrt(v)]: generic quadratic extension, arith/fieldquadratic.c fq_* */
typedef struct { fd x, y; } fk;
/* fq_mul (fieldquadratic.c:197-233) */
static void fk_mul(const oracle_pairing *P, fk *r, const fk *a, const fk *b) {
  fd e0 = a->x, e1 = b->x, e2, rx, ry;
  fd_add(P, &e0, &a->x, &a->y);
  fd_add(P, &e1, &b->x, &b->y);
  fd_mul(P, &e2, &e0, &e1);
  fd_mul(P, &e0, &a->x, &b->x);
  fd_mul(P, &e1, &a->y, &b->y);
  fd_mul_fq(P, &rx, &e1, &P->D->nqr);
  fd_add(P, &rx, &rx, &e0);
  fd_sub(P, &e2, &e2, &e0);
  fd_sub(P, &ry, &e2, &e1);
  r->x = rx; r->y = ry;
}
/* fq_square (fieldquadratic.c:249-269) */
static void fk_sqr(const oracle_pairing *P, fk *r, const fk *a) {
  fd e0, e1;
  fd_sqr(P, &e0, &a->x);
  fd_sqr(P, &e1, &a->y);
  fd_mul_fq(P, &e1, &e1, &P->D->nqr);
  fd_add(P, &e0, &e0, &e1);
  fd_mul(P, &e1, &a->x, &a->y);
  fd_dbl(P, &e1, &e1);
  r->x = e0; r->y = e1;
}
/* fq_invert (fieldquadratic.c:290-309) */
static void fk_inv(const oracle_pairing *P, fk *r, const fk *a) {
  fd e0, e1;
  fd_sqr(P, &e0, &a->x);
  fd_sqr(P, &e1, &a->y);
  fd_mul_fq(P, &e1, &e1, &P->D->nqr);
  fd_sub(P, &e0, &e0, &e1);
  fd_inv(P, &e0, &e0);
  fd_mul(P, &r->x, &a->x, &e0);
  fd_neg(P, &e0, &e0);
  fd_mul(P, &r->y, &a->y, &e0);
}
static void fk_one(const oracle_pairing *P, fk *r) { fd_set_fq(P, &r->x, &P->Fq.R); fd_set_fq(P, &r->y, &P->Fq.zero); }
static int fk_is1(const oracle_pairing *P, const fk *a) { fk o; fk_one(P, &o); return fd_eq(P, &a->x, &o.x) && fd_is0(P, &a->y); }

/* d_miller_evalfn (d_param.c:99-111, g_param.c:86-102) */
static void d_evalfn(const oracle_pairing *P, fk *e0, const fe *a, const fe *b, const fe *c, const fd *Qx, const fd *Qy) {
  const fpctx *F = &P->Fq;
  fk_one(P, e0);
  for (int i = 0; i < DEG; i++) {
    fp_mul(F, &e0->x.c[i], &Qx->c[i], a);
    fp_mul(F, &e0->y.c[i], &Qy->c[i], b);
  }
  fp_add(F, &e0->x.c[0], &e0->x.c[0], c);
}
/* cc_miller_no_denom_affine (d_param.c:321-422, the default :1085; g_param.c:308-408, the default :1342) */
static void d_miller(const oracle_pairing *P, fk *res, const pt *Pp, const fd *Qx, const fd *Qy) {
  const fpctx *F = &P->Fq;
  fk v, e0;
  pt Z = *Pp;
  fe a, b, c, t0;
  fk_one(P, &v);
  int m = big_bits(&P->r);
  m = m > 2 ? m - 2 : 0;
  for (;;) {
    /* do_tangent (d_param.c:344-362) */
    fp_sqr(F, &a, &Z.x);
    fp_dbl(F, &t0, &a); fp_add(F, &a, &a, &t0);       /* element_mul_si(a, a, 3) */
    fp_add(F, &a, &a, &P->ca);
    fp_neg(F, &a, &a);
    fp_add(F, &b, &Z.y, &Z.y);
    fp_mul(F, &t0, &b, &Z.y);
    fp_mul(F, &c, &a, &Z.x);
    fp_add(F, &c, &c, &t0);
    fp_neg(F, &c, &c);
    d_evalfn(P, &e0, &a, &b, &c, Qx, Qy);
    fk_mul(P, &v, &v, &e0);
    if (!m) break;
    pt_dbl(F, &P->ca, &Z, &Z);
    if (big_bit(&P->r, m)) {
      /* do_line (d_param.c:364-379) */
      fp_sub(F, &b, &Pp->x, &Z.x);
      fp_sub(F, &a, &Z.y, &Pp->y);
      fp_mul(F, &t0, &b, &Z.y);
      fp_mul(F, &c, &a, &Z.x);
      fp_add(F, &c, &c, &t0);
      fp_neg(F, &c, &c);
      d_evalfn(P, &e0, &a, &b, &c, Qx, Qy);
      fk_mul(P, &v, &v, &e0);
      pt_add(F, &P->ca, &Z, &Z, Pp);
    }
    m--;
    fk_sqr(P, &v, &v);
  }
  *res = v;
}
/* lucas_even (d_param.c:441-502, g_param.c:413-469), over Fq^d */
static void d_lucas_even(const oracle_pairing *P, fk *out, fk *in, const big *cofactor) {
  const fpctx *F = &P->Fq;
  if (fk_is1(P, in)) { *out = *in; return; }
  fd t0, t1, v0, v1;
  fd *in0 = &in->x, *in1 = &in->y;
  { fe two; fp_set_ui(F, &two, 2); fd_set_fq(P, &t0, &two); }
  fd_dbl(P, &t1, in0);
  v0 = t0; v1 = t1;
  int j = big_bits(cofactor) - 1;
  for (;;) {
    if (!j) {
      fd_mul(P, &v1, &v0, &v1); fd_sub(P, &v1, &v1, &t1);
      fd_sqr(P, &v0, &v0);      fd_sub(P, &v0, &v0, &t0);
      break;
    }
    if (big_bit(cofactor, j)) {
      fd_mul(P, &v0, &v0, &v1); fd_sub(P, &v0, &v0, &t1);
      fd_sqr(P, &v1, &v1);      fd_sub(P, &v1, &v1, &t0);
    } else {
      fd_mul(P, &v1, &v0, &v1); fd_sub(P, &v1, &v1, &t1);
      fd_sqr(P, &v0, &v0);      fd_sub(P, &v0, &v0, &t0);
    }
    j--;
  }
  fd_dbl(P, &v0, &v0);
  fd_mul(P, in0, &t1, &v1);
  fd_sub(P, in0, in0, &v0);
  fd_sqr(P, &t1, &t1);
  fd_sub(P, &t1, &t1, &t0);
  fd_sub(P, &t1, &t1, &t0);
  fd_halve(P, &v0, &v1);
  { fd ti; fd_inv(P, &ti, &t1); fd_mul(P, &v1, in0, &ti); }   /* element_div */
  fd_mul(P, &v1, &v1, in1);
  out->x = v0; out->y = v1;
}
/* cc_tatepower, k == 6 branch (d_param.c:505-564) and tatepower10 (g_param.c:471-536): with d = k/2,
 * in^((q^d - 1)(q + 1)) = in^(q^(d+1)) in^(q^d) / (in^q in), then the Lucas ladder over Phi_k(q)/r */
static void d_tatepower(const oracle_pairing *P, fk *out, fk *in) {
  fk e0, e3;
  /* qpower(1): e0 = in.x^q + in.y^q sqrt(v) */
  fd_frob(P, &e0.x, &in->x); fd_frob(P, &e0.y, &in->y);
  e3 = e0;
  e0.x = in->x; fd_neg(P, &e0.y, &in->y);
  fk_mul(P, &e3, &e3, &e0);
  /* qpower(-1) */
  fd_frob(P, &e0.x, &in->x); fd_frob(P, &e0.y, &in->y); fd_neg(P, &e0.y, &e0.y);
  fk_mul(P, &e0, &e0, in);
  fk_inv(P, &e0, &e0);
  fk_mul(P, in, &e3, &e0);
  e0 = *in;
  d_lucas_even(P, out, &e0, &P->D->phikonr);
}
/* curve_is_valid_point / curve_from_bytes over Fq^d for the twist (curve.c:57-77, 609-623) */
typedef struct { int inf; fd x, y; } ptd;
static void d_twist_from_bytes(const oracle_pairing *P, ptd *Q, const uint8_t *b) {
  const fpctx *F = &P->Fq;
  fd t0, t1;
  Q->inf = 0;
  fd_from_bytes(P, &Q->x, b);
  fd_from_bytes(P, &Q->y, b + DEG * F->nbytes);
  fd_sqr(P, &t0, &Q->x);
  fp_add(F, &t0.c[0], &t0.c[0], &P->D->ta);
  fd_mul(P, &t0, &t0, &Q->x);
  fp_add(F, &t0.c[0], &t0.c[0], &P->D->tb);
  fd_sqr(P, &t1, &Q->y);
  if (!fd_eq(P, &t0, &t1)) Q->inf = 1;
}
static void fk_to_bytes(const oracle_pairing *P, uint8_t *b, const fk *a) { fd_to_bytes(P, b, &a->x); fd_to_bytes(P, b + DEG * P->Fq.nbytes, &a->y); }
static void fk_from_bytes(const oracle_pairing *P, fk *a, const uint8_t *b) { fd_from_bytes(P, &a->x, b); fd_from_bytes(P, &a->y, b + DEG * P->Fq.nbytes); }

/* cc_pairing (d_param.c:570-587, g_param.c:541-558) / cc_pairings_affine (d_param.c:710-736): the
 * product routine interleaves the k Miller loops (shared squaring, simultaneous inversions); the
 * value is the product of the k Miller functions, one tate power.  Type g has no product routine
 * (generic_prod_pairings, pairing.c:35-46: the product of the full pairings -- the same element). */
static int d_pairing_bytes(const oracle_pairing *P, const uint8_t *g1, const uint8_t *g2, uint8_t *gt, int k) {
  const fpctx *F = &P->Fq;
  fk acc, m, out;
  int ident = 0;
  fk_one(P, &acc);
  for (int j = 0; j < k; j++) {
    pt A; ptd B; fd Qx, Qy;
    pt_from_bytes(F, &P->ca, &P->cb, &A, g1 + (size_t) j * P->len1);
    d_twist_from_bytes(P, &B, g2 + (size_t) j * P->len2);
    if (A.inf || B.inf) { ident = 1; continue; }
    fd_mul_fq(P, &Qx, &B.x, &P->D->nqrinv);      /* twist map (x,y) -> (v^-1 x, v^-2 y sqrt(v)) */
    fd_mul_fq(P, &Qy, &B.y, &P->D->nqrinv2);
    d_miller(P, &m, &A, &Qx, &Qy);
    fk_mul(P, &acc, &acc, &m);
  }
  if (ident) { gt_one_bytes(P, gt); return 0; }
  d_tatepower(P, &out, &acc);
  fk_to_bytes(P, gt, &out);
  return 0;
}

/* big helpers for Phi_k(q)/r */
static void big_mul(big *r, const big *a, const big *b) {
  big z; memset(&z, 0, sizeof z);
  for (int i = 0; i < BIGL; i++) {
    if (!a->v[i]) continue;
    u128 c = 0;
    for (int j = 0; i + j < BIGL; j++) { c += (u128) a->v[i] * b->v[j] + z.v[i + j]; z.v[i + j] = (uint64_t) c; c >>= 64; }
  }
  *r = z;
}
static int big_divexact(big *quo, const big *z, const big *d) {
  big rem; memset(quo, 0, sizeof *quo); memset(&rem, 0, sizeof rem);
  for (int i = big_bits(z) - 1; i >= 0; i--) {
    for (int w = BIGL - 1; w > 0; w--) rem.v[w] = (rem.v[w] << 1) | (rem.v[w - 1] >> 63);
    rem.v[0] = (rem.v[0] << 1) | (uint64_t) big_bit(z, i);
    if (bn_cmp(rem.v, d->v, BIGL) >= 0) { bn_sub(rem.v, rem.v, d->v, BIGL); quo->v[i / 64] |= 1ull << (i % 64); }
  }
  return bn_is0(rem.v, BIGL) ? 0 : 1;
}

/* d_init_pairing (d_param.c:993-1095) + pbc_param_init_d;  g_init_pairing (g_param.c:1248-1354) +
 * pbc_param_init_g (:1378-1402) */
static int init_d(oracle_pairing *P, const char *txt, size_t len) {
  big a, b, nqr, co[MAXD]; int k;
  const int d = P->type == 'g' ? 5 : 3;
  if (kv_big(txt, len, "q", &P->q) || kv_big(txt, len, "r", &P->r) || kv_big(txt, len, "a", &a) ||
      kv_big(txt, len, "b", &b) || kv_big(txt, len, "nqr", &nqr))
    return 1;
  if (kv_big(txt, len, "h", &P->h)) return 1;        /* cofactor of E(F_q) (d_param.c:1016, g_param.c:1267) */
  if (d == 3 && (kv_int(txt, len, "k", &k) || k != 6)) return 1;      /* type g files carry no k that matters: k = 10 */
  for (int i = 0; i < d; i++) {
    char key[8];
    snprintf(key, sizeof key, "coeff%d", i);
    if (kv_big(txt, len, key, &co[i])) return 1;
  }
  if (fp_init(&P->Fq, &P->q)) return 1;
  const fpctx *F = &P->Fq;
  struct dctx *D = P->D = calloc(1, sizeof *D);
  D->deg = d;
  fe cf[MAXD];
#define SETBIG(dst, src) do { fe t_; memset(&t_, 0, sizeof t_); memcpy(t_.v, (src).v, 8 * F->n); fp_mul(F, &(dst), &t_, &F->R2); } while (0)
  SETBIG(P->ca, a); SETBIG(P->cb, b); SETBIG(D->nqr, nqr);
  for (int i = 0; i < d; i++) SETBIG(cf[i], co[i]);
  /* x^d = -(c0 + c1 x + ...); x^(d+j) = x * x^(d+j-1) reduced (compute_x_powers, poly.c:1302-1333) */
  fd_set_fq(P, &D->xpwr[0], &F->zero);
  for (int i = 0; i < d; i++) fp_neg(F, &D->xpwr[0].c[i], &cf[i]);
  for (int j = 1; j < d - 1; j++) {
    fd *prev = &D->xpwr[j - 1], *cur = &D->xpwr[j], t;
    fd_set_fq(P, cur, &F->zero);
    for (int i = 1; i < d; i++) cur->c[i] = prev->c[i - 1];
    fd_mul_fq(P, &t, &D->xpwr[0], &prev->c[d - 1]);
    fd_add(P, cur, cur, &t);
  }
  fp_inv(F, &D->nqrinv, &D->nqr);
  fp_sqr(F, &D->nqrinv2, &D->nqrinv);
  /* twist coefficients a v^2, b v^3 (field_reinit_curve_twist, curve.c:885-901) */
  { fe v2; fp_sqr(F, &v2, &D->nqr); fp_mul(F, &D->ta, &P->ca, &v2); fp_mul(F, &v2, &v2, &D->nqr); fp_mul(F, &D->tb, &P->cb, &v2); }
  /* xpowq[i-1] = x^(iq) */
  {
    fd x; fd_set_fq(P, &x, &F->zero); x.c[1] = F->R;
    fd_pow(P, &D->xpowq[0], &x, &P->q);
    for (int i = 1; i < d - 1; i++) fd_mul(P, &D->xpowq[i], &D->xpowq[i - 1], &D->xpowq[0]);
  }
  /* phikonr = Phi_k(q) / r: q^2 - q + 1 (k = 6), q^4 - q^3 + q^2 - q + 1 (k = 10) */
  {
    big one, q2, z; memset(&one, 0, sizeof one); one.v[0] = 1;
    big_mul(&q2, &P->q, &P->q);
    z = q2;
    bn_sub(z.v, z.v, P->q.v, BIGL);
    bn_add(z.v, z.v, one.v, BIGL);                       /* q^2 - q + 1 */
    if (d == 5) {
      big q3, q4;
      big_mul(&q3, &q2, &P->q);
      big_mul(&q4, &q3, &P->q);
      bn_add(z.v, z.v, q4.v, BIGL);
      bn_sub(z.v, z.v, q3.v, BIGL);
    }
    if (big_divexact(&D->phikonr, &z, &P->r)) return 1;
  }
  P->len1 = 2 * F->nbytes; P->len2 = 2 * d * F->nbytes; P->lenT = 2 * d * F->nbytes;
  return 0;
}
static int df_gt_mul(const oracle_pairing *P, const uint8_t *a, const uint8_t *b, uint8_t *out);
static int df_gt_pow(const oracle_pairing *P, const uint8_t *a, const big *e, uint8_t *out);


/* ================================================================== */
/* Type F (BN, k = 12), ecc/f_param.c                                  */
/* ================================================================== */
/* Fq2 = Fq[sqrt(beta)] (generic fq_*, fieldquadratic.c); Fq12 = Fq2[x]/(x^6 + alpha) (polymod) */
typedef struct { fe x, y; } g2;
typedef struct { g2 c[6]; } f12;
struct fctx {
  fe beta;               /* nqr of Fq (f_param.c:345-348) */
  g2 negalpha;           /* x^6 = negalpha = -(alpha0 + alpha1 sqrt(beta)) (f_param.c:355-361) */
  g2 negalphainv;
  g2 xpowq2, xpowq6, xpowq8;   /* x^(q^k) = (this) * x (f_param.c:431-444) */
  g2 tb;                 /* twist curve y^2 = x^3 + tb, tb = -alpha b (f_param.c:372-381) */
  big tateexp;           /* (q^4 - q^2 + 1)/r (f_param.c:414-420) */
};

static void g2_add(const fpctx *F, g2 *r, const g2 *a, const g2 *b) { fp_add(F, &r->x, &a->x, &b->x); fp_add(F, &r->y, &a->y, &b->y); }
static void __attribute__((unused)) g2_sub(const fpctx *F, g2 *r, const g2 *a, const g2 *b) { fp_sub(F, &r->x, &a->x, &b->x); fp_sub(F, &r->y, &a->y, &b->y); }
static void __attribute__((unused)) g2_neg(const fpctx *F, g2 *r, const g2 *a) { fp_neg(F, &r->x, &a->x); fp_neg(F, &r->y, &a->y); }
static int g2_eq(const fpctx *F, const g2 *a, const g2 *b) { return fp_eq(F, &a->x, &b->x) && fp_eq(F, &a->y, &b->y); }
/* fq_mul (fieldquadratic.c:197-233) */
static void g2_mul(const oracle_pairing *P, g2 *r, const g2 *a, const g2 *b) {
  const fpctx *F = &P->Fq;
  fe e0, e1, e2, rx;
  fp_add(F, &e0, &a->x, &a->y);
  fp_add(F, &e1, &b->x, &b->y);
  fp_mul(F, &e2, &e0, &e1);
  fp_mul(F, &e0, &a->x, &b->x);
  fp_mul(F, &e1, &a->y, &b->y);
  fp_mul(F, &rx, &e1, &P->Fx->beta);
  fp_add(F, &rx, &rx, &e0);
  fp_sub(F, &e2, &e2, &e0);
  fp_sub(F, &r->y, &e2, &e1);
  r->x = rx;
}
/* fq_square (fieldquadratic.c:249-269) */
static void g2_sqr(const oracle_pairing *P, g2 *r, const g2 *a) {
  const fpctx *F = &P->Fq;
  fe e0, e1;
  fp_sqr(F, &e0, &a->x);
  fp_sqr(F, &e1, &a->y);
  fp_mul(F, &e1, &e1, &P->Fx->beta);
  fp_add(F, &e0, &e0, &e1);
  fp_mul(F, &e1, &a->x, &a->y);
  fp_dbl(F, &e1, &e1);
  r->x = e0; r->y = e1;
}
/* fq_invert (fieldquadratic.c:290-309) */
static void g2_inv(const oracle_pairing *P, g2 *r, const g2 *a) {
  const fpctx *F = &P->Fq;
  fe e0, e1;
  fp_sqr(F, &e0, &a->x);
  fp_sqr(F, &e1, &a->y);
  fp_mul(F, &e1, &e1, &P->Fx->beta);
  fp_sub(F, &e0, &e0, &e1);
  fp_inv(F, &e0, &e0);
  fp_mul(F, &r->x, &a->x, &e0);
  fp_neg(F, &e0, &e0);
  fp_mul(F, &r->y, &a->y, &e0);
}
static void g2_mul_fq(const fpctx *F, g2 *r, const g2 *a, const fe *s) { fp_mul(F, &r->x, &a->x, s); fp_mul(F, &r->y, &a->y, s); }
static void g2_from_bytes(const fpctx *F, g2 *r, const uint8_t *b) { fp_from_bytes(F, &r->x, b); fp_from_bytes(F, &r->y, b + F->nbytes); }
static void g2_to_bytes(const fpctx *F, uint8_t *b, const g2 *a) { fp_to_bytes(F, b, &a->x); fp_to_bytes(F, b + F->nbytes, &a->y); }
static void g2_zero(const fpctx *F, g2 *r) { r->x = F->zero; r->y = F->zero; }

static void f12_one(const fpctx *F, f12 *r) { for (int i = 0; i < 6; i++) g2_zero(F, &r->c[i]); r->c[0].x = F->R; }
/* polymod_mul (poly.c:1005-1047), n = 6, x^(6+i) = negalpha x^i */
static void f12_mul(const oracle_pairing *P, f12 *r, const f12 *a, const f12 *b) {
  const fpctx *F = &P->Fq;
  g2 d[11], t;
  for (int i = 0; i < 11; i++) g2_zero(F, &d[i]);
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) {
    g2_mul(P, &t, &a->c[i], &b->c[j]);
    g2_add(F, &d[i + j], &d[i + j], &t);
  }
  for (int i = 0; i < 5; i++) {
    g2_mul(P, &t, &d[6 + i], &P->Fx->negalpha);
    g2_add(F, &d[i], &d[i], &t);
  }
  for (int i = 0; i < 6; i++) r->c[i] = d[i];
}
static void f12_sqr(const oracle_pairing *P, f12 *r, const f12 *a) { f12_mul(P, r, a, a); }   /* poly.c:1091-1143 */
/* coefficient-wise Frobenius power: out^(q^k) for even k, x^(q^k) = e x (qpower, f_param.c:257-268) */
static void f12_qpower(const oracle_pairing *P, f12 *r, const f12 *a, const g2 *e) {
  g2 epow = *e;
  f12 res;
  res.c[0] = a->c[0];
  g2_mul(P, &res.c[1], &a->c[1], e);
  for (int i = 2; i < 6; i++) {
    g2_mul(P, &epow, &epow, e);
    g2_mul(P, &res.c[i], &a->c[i], &epow);
  }
  *r = res;
}
/* polymod_invert (poly.c:521-536): unique inverse; here via the norm to Fq2:
 * sigma = (q^2)-power Frobenius, a^-1 = prod_{i=1..5} sigma^i(a) / N, N = prod_{i=0..5} sigma^i(a) in Fq2 */
static void f12_inv(const oracle_pairing *P, f12 *r, const f12 *a) {
  f12 s = *a, t, n;
  f12_qpower(P, &s, &s, &P->Fx->xpowq2);
  t = s;
  for (int i = 2; i <= 5; i++) { f12_qpower(P, &s, &s, &P->Fx->xpowq2); f12_mul(P, &t, &t, &s); }
  f12_mul(P, &n, a, &t);
  g2 ni; g2_inv(P, &ni, &n.c[0]);
  for (int i = 0; i < 6; i++) g2_mul(P, &r->c[i], &t.c[i], &ni);
}
static void f12_pow(const oracle_pairing *P, f12 *r, const f12 *a, const big *e) {
  f12 acc, base = *a; f12_one(&P->Fq, &acc);
  for (int i = big_bits(e) - 1; i >= 0; i--) {
    f12_sqr(P, &acc, &acc);
    if (big_bit(e, i)) f12_mul(P, &acc, &acc, &base);
  }
  *r = acc;
}
static void f12_to_bytes(const fpctx *F, uint8_t *b, const f12 *a) { for (int i = 0; i < 6; i++) g2_to_bytes(F, b + 2 * i * F->nbytes, &a->c[i]); }
static void f12_from_bytes(const fpctx *F, f12 *a, const uint8_t *b) { for (int i = 0; i < 6; i++) g2_from_bytes(F, &a->c[i], b + 2 * i * F->nbytes); }

/* f_miller_evalfn (f_param.c:109-149): v <- v * (a Qx x^4 + b Qy x^3 + c) */
static void f_evalfn(const oracle_pairing *P, f12 *v, const fe *a, const fe *b, const fe *c, const g2 *Qx, const g2 *Qy) {
  const fpctx *F = &P->Fq;
  static const int term[6][4] = { {0, 2, 3, 2}, {1, 3, 4, 2}, {2, 4, 5, 2}, {3, 5, 0, 1}, {4, 0, 1, 0}, {5, 1, 2, 0} };
  f12 e0;
  for (int t = 0; t < 6; t++) {
    int i = term[t][0], j = term[t][1], k = term[t][2], flag = term[t][3];
    g2 e1, e2;
    g2_mul(P, &e1, &v->c[j], Qx);
    if (flag == 1) g2_mul(P, &e1, &e1, &P->Fx->negalpha);
    g2_mul_fq(F, &e1, &e1, a);
    g2_mul(P, &e2, &v->c[k], Qy);
    g2_mul_fq(F, &e2, &e2, b);
    g2_add(F, &e2, &e2, &e1);
    if (flag == 2) g2_mul(P, &e2, &e2, &P->Fx->negalpha);
    g2_mul_fq(F, &e1, &v->c[i], c);
    g2_add(F, &e2, &e2, &e1);
    e0.c[i] = e2;
  }
  *v = e0;
}
/* cc_miller_no_denom (f_param.c:97-248) */
static void f_miller(const oracle_pairing *P, f12 *res, const pt *Pp, const g2 *Qx, const g2 *Qy) {
  const fpctx *F = &P->Fq;
  f12 v;
  pt Z = *Pp;
  fe a, b, c, t0;
  f12_one(F, &v);
  int m = big_bits(&P->r);
  m = m > 2 ? m - 2 : 0;
  for (;;) {
    /* do_tangent (f_param.c:171-184): curve a coefficient is 0 */
    fp_sqr(F, &a, &Z.x);
    fp_dbl(F, &t0, &a); fp_add(F, &a, &a, &t0);
    fp_neg(F, &a, &a);
    fp_add(F, &b, &Z.y, &Z.y);
    fp_mul(F, &t0, &b, &Z.y);
    fp_mul(F, &c, &a, &Z.x);
    fp_add(F, &c, &c, &t0);
    fp_neg(F, &c, &c);
    f_evalfn(P, &v, &a, &b, &c, Qx, Qy);
    if (!m) break;
    pt_dbl(F, &P->ca, &Z, &Z);
    if (big_bit(&P->r, m)) {
      /* do_line (f_param.c:190-199) */
      fp_sub(F, &b, &Pp->x, &Z.x);
      fp_sub(F, &a, &Z.y, &Pp->y);
      fp_mul(F, &t0, &b, &Z.y);
      fp_mul(F, &c, &a, &Z.x);
      fp_add(F, &c, &c, &t0);
      fp_neg(F, &c, &c);
      f_evalfn(P, &v, &a, &b, &c, Qx, Qy);
      pt_add(F, &P->ca, &Z, &Z, Pp);
    }
    m--;
    f12_sqr(P, &v, &v);
  }
  *res = v;
}
/* f_tateexp (f_param.c:250-283) */
static void f_tateexp(const oracle_pairing *P, f12 *out) {
  f12 x, y;
  f12_qpower(P, &y, out, &P->Fx->xpowq8);
  f12_qpower(P, &x, out, &P->Fx->xpowq6);
  f12_mul(P, &y, &y, &x);
  f12_qpower(P, &x, out, &P->Fx->xpowq2);
  f12_mul(P, &x, &x, out);
  f12_inv(P, &x, &x);
  f12_mul(P, out, &y, &x);
  f12_pow(P, out, out, &P->Fx->tateexp);
}
typedef struct { int inf; g2 x, y; } pt2;
static void f_twist_from_bytes(const oracle_pairing *P, pt2 *Q, const uint8_t *b) {
  const fpctx *F = &P->Fq;
  g2 t0, t1;
  Q->inf = 0;
  g2_from_bytes(F, &Q->x, b);
  g2_from_bytes(F, &Q->y, b + 2 * F->nbytes);
  g2_sqr(P, &t0, &Q->x);               /* a = 0 */
  g2_mul(P, &t0, &t0, &Q->x);
  g2_add(F, &t0, &t0, &P->Fx->tb);
  g2_sqr(P, &t1, &Q->y);
  if (!g2_eq(F, &t0, &t1)) Q->inf = 1;
}
/* f_pairing (f_param.c:289-311); products: generic_prod_pairings (ecc/pairing.c:35-46) =
 * product of k full pairings (Type F installs no dedicated prod_pairings). */
static int f_pairing_bytes(const oracle_pairing *P, const uint8_t *g1, const uint8_t *g2b, uint8_t *gt, int k) {
  const fpctx *F = &P->Fq;
  f12 acc, m;
  int ident = 0;
  f12_one(F, &acc);
  for (int j = 0; j < k; j++) {
    pt A; pt2 B; g2 x, y;
    pt_from_bytes(F, &P->ca, &P->cb, &A, g1 + (size_t) j * P->len1);
    f_twist_from_bytes(P, &B, g2b + (size_t) j * P->len2);
    if (A.inf || B.inf) { ident = 1; continue; }
    g2_mul(P, &x, &B.x, &P->Fx->negalphainv);
    g2_mul(P, &y, &B.y, &P->Fx->negalphainv);
    f_miller(P, &m, &A, &x, &y);
    f_tateexp(P, &m);
    f12_mul(P, &acc, &acc, &m);
  }
  if (ident) { gt_one_bytes(P, gt); return 0; }
  f12_to_bytes(F, gt, &acc);
  return 0;
}

/* f_init_pairing (f_param.c:335-447) */
static int init_f(oracle_pairing *P, const char *txt, size_t len) {
  big b, beta, a0, a1;
  if (kv_big(txt, len, "q", &P->q) || kv_big(txt, len, "r", &P->r) || kv_big(txt, len, "b", &b) ||
      kv_big(txt, len, "beta", &beta) || kv_big(txt, len, "alpha0", &a0) || kv_big(txt, len, "alpha1", &a1))
    return 1;
  if (fp_init(&P->Fq, &P->q)) return 1;
  const fpctx *F = &P->Fq;
  struct fctx *X = P->Fx = calloc(1, sizeof *X);
  fe fa0, fa1;
  P->ca = F->zero;
  SETBIG(P->cb, b); SETBIG(X->beta, beta); SETBIG(fa0, a0); SETBIG(fa1, a1);
  fp_neg(F, &X->negalpha.x, &fa0);
  fp_neg(F, &X->negalpha.y, &fa1);
  g2_inv(P, &X->negalphainv, &X->negalpha);
  g2_mul_fq(F, &X->tb, &X->negalpha, &P->cb);       /* -alpha0 b - alpha1 b sqrt(beta) */
  /* tateexp = ((q^2 - 1) q^2 + 1) / r */
  {
    int n = F->n;
    big q2, z; memset(&q2, 0, sizeof q2); memset(&z, 0, sizeof z);
    for (int i = 0; i < n; i++) { u128 c = 0; for (int j = 0; j < n; j++) { c += (u128) P->q.v[i] * P->q.v[j] + q2.v[i + j]; q2.v[i + j] = (uint64_t) c; c >>= 64; } q2.v[i + n] += (uint64_t) c; }
    big one; memset(&one, 0, sizeof one); one.v[0] = 1;
    big q2m1 = q2; bn_sub(q2m1.v, q2m1.v, one.v, BIGL);
    for (int i = 0; i < 2 * n; i++) { u128 c = 0; for (int j = 0; j < 2 * n; j++) { c += (u128) q2m1.v[i] * q2.v[j] + z.v[i + j]; z.v[i + j] = (uint64_t) c; c >>= 64; } z.v[i + 2 * n] += (uint64_t) c; }
    bn_add(z.v, z.v, one.v, BIGL);
    big quo, rem; memset(&quo, 0, sizeof quo); memset(&rem, 0, sizeof rem);
    for (int i = big_bits(&z) - 1; i >= 0; i--) {
      for (int w = BIGL - 1; w > 0; w--) rem.v[w] = (rem.v[w] << 1) | (rem.v[w - 1] >> 63);
      rem.v[0] = (rem.v[0] << 1) | (uint64_t) big_bit(&z, i);
      if (bn_cmp(rem.v, P->r.v, BIGL) >= 0) { bn_sub(rem.v, rem.v, P->r.v, BIGL); quo.v[i / 64] |= 1ull << (i % 64); }
    }
    if (!bn_is0(rem.v, BIGL)) return 1;
    X->tateexp = quo;
  }
  /* xpowq2/6/8: coefficient of x in x^(q^k) (f_param.c:431-444) */
  {
    f12 xp; f12_one(F, &xp); xp.c[0].x = F->zero; xp.c[1].x = F->R;
    for (int i = 1; i <= 8; i++) {
      f12_pow(P, &xp, &xp, &P->q);
      if (i == 2) X->xpowq2 = xp.c[1];
      if (i == 6) X->xpowq6 = xp.c[1];
      if (i == 8) X->xpowq8 = xp.c[1];
    }
  }
  P->len1 = 2 * F->nbytes; P->len2 = 4 * F->nbytes; P->lenT = 12 * F->nbytes;
  return 0;
}
#undef SETBIG

/* ================================================================== */
/* G2 twists of types d, g (E'(F_q^d)) and f (E'(F_q^2)): hashing and compressed points                 */
/* ================================================================== */
/* One element of the twist's field, whichever it is. */
typedef struct { fd d; g2 q; } xe;
#define ISF (P->type == 'f')
static void xe_mul(const oracle_pairing *P, xe *r, const xe *a, const xe *b) { if (ISF) g2_mul(P, &r->q, &a->q, &b->q); else fd_mul(P, &r->d, &a->d, &b->d); }
static void xe_sqr(const oracle_pairing *P, xe *r, const xe *a) { if (ISF) g2_sqr(P, &r->q, &a->q); else fd_sqr(P, &r->d, &a->d); }
static void xe_add(const oracle_pairing *P, xe *r, const xe *a, const xe *b) { if (ISF) g2_add(&P->Fq, &r->q, &a->q, &b->q); else fd_add(P, &r->d, &a->d, &b->d); }
static void xe_neg(const oracle_pairing *P, xe *r, const xe *a) { if (ISF) g2_neg(&P->Fq, &r->q, &a->q); else fd_neg(P, &r->d, &a->d); }
static int xe_eq(const oracle_pairing *P, const xe *a, const xe *b) { return ISF ? g2_eq(&P->Fq, &a->q, &b->q) : fd_eq(P, &a->d, &b->d); }
static void xe_set_fq(const oracle_pairing *P, xe *r, const fe *s) {
  memset(r, 0, sizeof *r);
  if (ISF) { r->q.x = *s; r->q.y = P->Fq.zero; } else fd_set_fq(P, &r->d, s);
}
static int xe_is0(const oracle_pairing *P, const xe *a) { xe z; xe_set_fq(P, &z, &P->Fq.zero); return xe_eq(P, a, &z); }
static int xe_ncoef(const oracle_pairing *P) { return ISF ? 2 : DEG; }
static fe *xe_coef(const oracle_pairing *P, xe *a, int i) { return ISF ? (i ? &a->q.y : &a->q.x) : &a->d.c[i]; }
static void xe_from_bytes(const oracle_pairing *P, xe *r, const uint8_t *b) {
  xe_set_fq(P, r, &P->Fq.zero);
  for (int i = 0; i < xe_ncoef(P); i++) fp_from_bytes(&P->Fq, xe_coef(P, r, i), b + i * P->Fq.nbytes);
}
static void xe_to_bytes(const oracle_pairing *P, uint8_t *b, xe *a) {
  for (int i = 0; i < xe_ncoef(P); i++) fp_to_bytes(&P->Fq, b + i * P->Fq.nbytes, xe_coef(P, a, i));
}
/* polymod_sgn (arith/poly.c:1189-1199) / fq_sign (arith/fieldquadratic.c:159-165): the first non-zero coefficient's */
static int xe_sgn(const oracle_pairing *P, xe *a) {
  for (int i = 0; i < xe_ncoef(P); i++) { int sg = fp_sgn(&P->Fq, xe_coef(P, a, i)); if (sg) return sg; }
  return 0;
}
/* polymod_from_hash (poly.c:341-348): every coefficient the same value; fq_from_hash (fieldquadratic.c:311-316):
 * the two halves of the digest */
static void xe_from_hash(const oracle_pairing *P, xe *r, const uint8_t *data, int len) {
  xe_set_fq(P, r, &P->Fq.zero);
  if (ISF) {
    const int k = len / 2;
    fp_from_hash(P, &r->q.x, data, k);
    fp_from_hash(P, &r->q.y, data + k, len - k);
  } else {
    fe h;
    fp_from_hash(P, &h, data, len);
    for (int i = 0; i < DEG; i++) r->d.c[i] = h;
  }
}
static void xe_pow(const oracle_pairing *P, xe *r, const xe *a, const big *e) {
  xe acc, base = *a;
  xe_set_fq(P, &acc, &P->Fq.R);
  for (int i = big_bits(e) - 1; i >= 0; i--) {
    xe_sqr(P, &acc, &acc);
    if (big_bit(e, i)) xe_mul(P, &acc, &acc, &base);
  }
  *r = acc;
}
static void big_shr1(big *a) {
  for (int w = 0; w < BIGL - 1; w++) a->v[w] = (a->v[w] >> 1) | (a->v[w + 1] << 63);
  a->v[BIGL - 1] >>= 1;
}
/* Square root in the twist's field; 0 when `a` is not a square.  The reference gets there by other means --
 * polymod_sqrt (poly.c:634-700) factors y^2 - a with random trial polynomials, fq_sqrt (fieldquadratic.c:357-392)
 * uses the norm -- and its callers fix the sign afterwards (curve_from_hash, element_from_bytes_compressed) or
 * keep whichever root came out (x-only), so the restatement is Tonelli-Shanks over |K*| = q^m - 1 = 2^s t from the
 * non-residue the tower is built on.  is_sqr: polymod_is_sqr (:617-632) is Euler's criterion (0 is no square);
 * fq_is_sqr (fieldquadratic.c:339-355) goes through the norm (0 is one). */
static int xe_sqrt(const oracle_pairing *P, xe *out, const xe *a) {
  xe one, z, c, x, b, tt, g;
  xe_set_fq(P, &one, &P->Fq.R);
  if (xe_is0(P, a)) { if (!ISF) return 0; *out = *a; return 1; }
  big n = P->q, t, e, half;
  for (int i = 1; i < xe_ncoef(P); i++) big_mul(&n, &n, &P->q);
  n.v[0] &= ~1ull;                                   /* q^m - 1 (q^m is odd) */
  half = n; big_shr1(&half);
  xe_pow(P, &tt, a, &half);
  if (!xe_eq(P, &tt, &one)) return 0;
  t = n;
  int s = 0;
  while (!big_bit(&t, 0)) { big_shr1(&t); s++; }
  if (ISF) { memset(&z, 0, sizeof z); z.q = P->Fx->negalpha; } else xe_set_fq(P, &z, &P->D->nqr);
  xe_pow(P, &c, &z, &t);
  e = t; e.v[0] += 1;                                /* t odd: no carry */
  big_shr1(&e);                                      /* (t + 1)/2 */
  xe_pow(P, &x, a, &e);
  xe_pow(P, &b, a, &t);
  int m = s;
  while (!xe_eq(P, &b, &one)) {
    int i = 0; tt = b;
    while (!xe_eq(P, &tt, &one)) { xe_sqr(P, &tt, &tt); i++; }
    g = c;
    for (int j = 0; j < m - i - 1; j++) xe_sqr(P, &g, &g);
    xe_mul(P, &x, &x, &g);
    xe_sqr(P, &c, &g);
    xe_mul(P, &b, &b, &c);
    m = i;
  }
  *out = x;
  return 1;
}
/* right-hand side of the twist: x^3 + a v^2 x + b v^3 (curve.c:885-901) / x^3 + tb (f_param.c:372-383) */
static void xe_rhs(const oracle_pairing *P, xe *t, const xe *x) {
  xe ca, cb;
  if (ISF) { xe_set_fq(P, &ca, &P->Fq.zero); memset(&cb, 0, sizeof cb); cb.q = P->Fx->tb; }
  else { xe_set_fq(P, &ca, &P->D->ta); xe_set_fq(P, &cb, &P->D->tb); }
  xe_sqr(P, t, x);
  xe_add(P, t, t, &ca);
  xe_mul(P, t, t, x);
  xe_add(P, t, t, &cb);
}
static int twist_types(const oracle_pairing *P) { return P->type == 'd' || P->type == 'g' || P->type == 'f'; }
/* curve_from_hash (ecc/curve.c:455-482) on the twist; no cofactor there (d_param.c:1057, f_param.c:383, g_param.c:1319) */
int oracle_from_hash_g2(const oracle_pairing *P, const uint8_t *data, int hlen, uint8_t *out, size_t n) {
  if (!twist_types(P)) return 1;
  const size_t fb = (size_t) xe_ncoef(P) * (size_t) P->Fq.nbytes;
  for (size_t i = 0; i < n; i++) {
    xe x, y, t, one;
    xe_set_fq(P, &one, &P->Fq.R);
    xe_from_hash(P, &x, data + i * (size_t) hlen, hlen);
    for (;;) {
      xe_rhs(P, &t, &x);
      if (xe_sqrt(P, &y, &t)) break;
      xe_sqr(P, &x, &x);
      xe_add(P, &x, &x, &one);
    }
    if (xe_sgn(P, &y) < 0) xe_neg(P, &y, &y);
    xe_to_bytes(P, out + i * 2 * fb, &x);
    xe_to_bytes(P, out + i * 2 * fb + fb, &y);
  }
  return 0;
}
/* what 0 / 1: element_to_bytes_compressed / element_from_bytes_compressed (ecc/curve.c:762-813) on the twist;
 * 2 / 3: the x-only pair (:821-836; the root as it comes: compare up to sign) */
int oracle_point_format_g2(const oracle_pairing *P, int what, const uint8_t *in, uint8_t *out, size_t n) {
  if (!twist_types(P)) return 1;
  const size_t fb = (size_t) xe_ncoef(P) * (size_t) P->Fq.nbytes, lp = 2 * fb;
  for (size_t i = 0; i < n; i++) {
    if (what == 0 || what == 2) {
      uint8_t *o = out + i * (fb + (what == 0));
      memcpy(o, in + i * lp, fb);
      if (what == 0) { xe y; xe_from_bytes(P, &y, in + i * lp + fb); o[fb] = xe_sgn(P, &y) > 0; }
    } else {
      const size_t li = fb + (what == 1);
      xe x, y, t;
      xe_from_bytes(P, &x, in + i * li);
      xe_rhs(P, &t, &x);
      if (!xe_sqrt(P, &y, &t)) { memset(out + i * lp, 0, lp); continue; }
      if (what == 1) {
        const int sg = xe_sgn(P, &y);
        if (in[i * li + fb] ? sg < 0 : sg > 0) xe_neg(P, &y, &y);
      }
      xe_to_bytes(P, out + i * lp, &x);
      xe_to_bytes(P, out + i * lp + fb, &y);
    }
  }
  return 0;
}

static int df_gt_mul(const oracle_pairing *P, const uint8_t *a, const uint8_t *b, uint8_t *out) {
  const fpctx *F = &P->Fq;
  if (P->type == 'd' || P->type == 'g') { fk x, y; fk_from_bytes(P, &x, a); fk_from_bytes(P, &y, b); fk_mul(P, &x, &x, &y); fk_to_bytes(P, out, &x); return 0; }
  if (P->type == 'f') { f12 x, y; f12_from_bytes(F, &x, a); f12_from_bytes(F, &y, b); f12_mul(P, &x, &x, &y); f12_to_bytes(F, out, &x); return 0; }
  return 1;
}
static int df_gt_pow(const oracle_pairing *P, const uint8_t *a, const big *e, uint8_t *out) {
  const fpctx *F = &P->Fq;
  if (P->type == 'd' || P->type == 'g') {
    fk x, acc; fk_from_bytes(P, &x, a); fk_one(P, &acc);
    for (int i = big_bits(e) - 1; i >= 0; i--) { fk_sqr(P, &acc, &acc); if (big_bit(e, i)) fk_mul(P, &acc, &acc, &x); }
    fk_to_bytes(P, out, &acc); return 0;
  }
  if (P->type == 'f') { f12 x; f12_from_bytes(F, &x, a); f12_pow(P, &x, &x, e); f12_to_bytes(F, out, &x); return 0; }
  return 1;
}

/* pairing->finalpow (a_finalpow a_param.c:1420-1429, cc_finalpow d_param.c:566-568, f_finalpow f_param.c:285-287,
 * g_finalpow g_param.c:1162-1164, e_finalpow e_param.c:828-830): the final exponentiation alone, on an element of GT's
 * underlying field given in GT's wire format.  (The consumers are gt_random / gt_from_hash, ecc/pairing.c:121,127.) */
int oracle_finalpow(const oracle_pairing *P, const uint8_t *in, uint8_t *out, size_t n) {
  const fpctx *F = &P->Fq;
  for (size_t i = 0; i < n; i++) {
    const uint8_t *a = in + i * P->lenT;
    uint8_t *o = out + i * P->lenT;
    if (P->type == 'a') {
      fe2 x, r;
      fp_from_bytes(F, &x.x, a); fp_from_bytes(F, &x.y, a + F->nbytes);
      if (P->a1) a1_tateexp(F, &r, &x, &P->h); else a_tateexp(F, &r, &x, &P->h);
      fp_to_bytes(F, o, &r.x); fp_to_bytes(F, o + F->nbytes, &r.y);
    } else if (P->type == 'd' || P->type == 'g') {
      fk x, r; fk_from_bytes(P, &x, a); d_tatepower(P, &r, &x); fk_to_bytes(P, o, &r);
    } else if (P->type == 'f') {
      f12 x; f12_from_bytes(F, &x, a); f_tateexp(P, &x); f12_to_bytes(F, o, &x);
    } else if (P->type == 'e') {
      fe x; fp_from_bytes(F, &x, a); fp_pow(F, &x, &x, &P->phikonr); fp_to_bytes(F, o, &x);
    } else return 1;
  }
  return 0;
}

/* ---- round 5: group law on G1, Z_r arithmetic, multi-exponentiations --------------------------------------------------- */
/* element_add / element_sub / element_neg / element_double on G1 = E(F_q) (curve_mul ecc/curve.c:153-207, curve_invert
 * :79-100, curve_double :102-151) -- op 0 a+b, 1 a-b, 2 -a, 3 2a; O is the all-zero record (and off-curve records). */
int oracle_g1_op(const oracle_pairing *P, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n) {
  const fpctx *F = &P->Fq;
  for (size_t i = 0; i < n; i++) {
    pt A, B, R;
    pt_from_bytes(F, &P->ca, &P->cb, &A, a + i * P->len1);
    if (!A.inf && fp_is0(F, &A.x) && fp_is0(F, &A.y)) A.inf = 1;
    B = A;
    if (op < 2) {
      pt_from_bytes(F, &P->ca, &P->cb, &B, b + i * P->len1);
      if (!B.inf && fp_is0(F, &B.x) && fp_is0(F, &B.y)) B.inf = 1;
    }
    if ((op == 1 || op == 2) && !B.inf) fp_neg(F, &B.y, &B.y);
    if (op == 2) R = B; else if (op == 3) pt_dbl(F, &P->ca, &R, &A); else pt_add(F, &P->ca, &R, &A, &B);
    pt_to_bytes(F, out + i * P->len1, &R);
  }
  return 0;
}
/* Z_r (the reference's F_p back end on the modulus r; pairing->Zr): op 0 a*b, 1 a+b, 2 a-b, 3 1/a, 4 -a, 5 a/2, 6 2a,
 * 7 a/b, 8 element_from_hash (a: digests of hlen bytes -- fp_from_hash arith/montfp.c:440-448: H || 0 || H || 1 ...
 * up to the byte length of r, halved while above r) on big-endian records of ceil(bits(r)/8) bytes */
/* 1/x mod m for an odd modulus m that need not be prime (type a1: r = n is composite, so Fermat's power is not the
 * inverse): binary extended Euclid, as mpz_invert would return it -- x, result in Montgomery form of Z; 0 when
 * gcd(x, m) != 1 */
static void zr_inv(const fpctx *Z, fe *c, const fe *x) {
  const int n = Z->n;
  fe one, t;
  memset(&one, 0, sizeof one); one.v[0] = 1;
  fp_mul(Z, &t, x, &one);                                  /* canonical integer */
  uint64_t u[MAXL + 1], v[MAXL + 1], x1[MAXL + 1], x2[MAXL + 1], m[MAXL + 1];
  memset(u, 0, sizeof u); memset(v, 0, sizeof v); memset(x1, 0, sizeof x1); memset(x2, 0, sizeof x2); memset(m, 0, sizeof m);
  memcpy(u, t.v, 8 * (size_t) n); memcpy(v, Z->p, 8 * (size_t) n); memcpy(m, Z->p, 8 * (size_t) n);
  x1[0] = 1;
  const int L = n + 1;
  if (bn_is0(u, L)) { memset(c, 0, sizeof *c); return; }
  for (int guard = 0; guard < 4 * 64 * L; guard++) {
    uint64_t oneL[MAXL + 1]; memset(oneL, 0, sizeof oneL); oneL[0] = 1;
    if (!bn_cmp(u, oneL, L) || !bn_cmp(v, oneL, L)) break;
    while (!(u[0] & 1)) {
      for (int i = 0; i < L - 1; i++) u[i] = (u[i] >> 1) | (u[i + 1] << 63);
      u[L - 1] >>= 1;
      if (x1[0] & 1) bn_add(x1, x1, m, L);
      for (int i = 0; i < L - 1; i++) x1[i] = (x1[i] >> 1) | (x1[i + 1] << 63);
      x1[L - 1] >>= 1;
    }
    while (!(v[0] & 1)) {
      for (int i = 0; i < L - 1; i++) v[i] = (v[i] >> 1) | (v[i + 1] << 63);
      v[L - 1] >>= 1;
      if (x2[0] & 1) bn_add(x2, x2, m, L);
      for (int i = 0; i < L - 1; i++) x2[i] = (x2[i] >> 1) | (x2[i + 1] << 63);
      x2[L - 1] >>= 1;
    }
    if (bn_cmp(u, v, L) >= 0) {
      bn_sub(u, u, v, L);
      if (bn_sub(x1, x1, x2, L)) bn_add(x1, x1, m, L);
      if (bn_is0(u, L)) break;                             /* gcd = v != 1 */
    } else {
      bn_sub(v, v, u, L);
      if (bn_sub(x2, x2, x1, L)) bn_add(x2, x2, m, L);
    }
  }
  uint64_t oneL[MAXL + 1]; memset(oneL, 0, sizeof oneL); oneL[0] = 1;
  fe r; memset(&r, 0, sizeof r);
  if (!bn_cmp(u, oneL, L)) memcpy(r.v, x1, 8 * (size_t) n);
  else if (!bn_cmp(v, oneL, L)) memcpy(r.v, x2, 8 * (size_t) n);
  fp_mul(Z, c, &r, &Z->R2);
}
int oracle_zr_op(const oracle_pairing *P, int op, const uint8_t *a, const uint8_t *b, int hlen, uint8_t *out, size_t n) {
  fpctx Z;
  if (fp_init(&Z, &P->r)) return 1;
  const int L = Z.nbytes;
  for (size_t i = 0; i < n; i++) {
    fe x, y, z;
    if (op == 8) {
      uint8_t buf[8 * MAXL];
      const uint8_t *data = a + i * (size_t) hlen;
      int count = L, k = 0, done = 0;
      uint8_t counter = 0;
      for (;;) {
        int m;
        if (hlen >= count - k) { m = count - k; done = 1; } else m = hlen;
        memcpy(buf + k, data, (size_t) m);
        k += m;
        if (done) break;
        buf[k++] = counter++;
        if (k == count) break;
      }
      big v;
      big_from_be(&v, buf, (size_t) count);
      while (bn_cmp(v.v, P->r.v, BIGL) > 0) {
        for (int w = 0; w < BIGL - 1; w++) v.v[w] = (v.v[w] >> 1) | (v.v[w + 1] << 63);
        v.v[BIGL - 1] >>= 1;
      }
      fe t; memset(&t, 0, sizeof t);
      memcpy(t.v, v.v, 8 * (size_t) Z.n);
      fp_mul(&Z, &z, &t, &Z.R2);
      fp_to_bytes(&Z, out + i * L, &z);
      continue;
    }
    fp_from_bytes(&Z, &x, a + i * L);
    if (b) fp_from_bytes(&Z, &y, b + i * L); else y = x;
    switch (op) {
      case 0: fp_mul(&Z, &z, &x, &y); break;
      case 1: fp_add(&Z, &z, &x, &y); break;
      case 2: fp_sub(&Z, &z, &x, &y); break;
      case 3: zr_inv(&Z, &z, &x); break;
      case 4: fp_neg(&Z, &z, &x); break;
      case 5: fp_halve(&Z, &z, &x); break;
      case 6: fp_dbl(&Z, &z, &x); break;
      default: zr_inv(&Z, &z, &y); fp_mul(&Z, &z, &z, &x); break;
    }
    fp_to_bytes(&Z, out + i * L, &z);
  }
  return 0;
}
/* element_pow2_zn / element_pow3_zn (include/pbc_field.h:496-531; arith/field.c:153-241) as the plain composition
 * a1^n1 a2^n2 (a3^n3) -- group 1: G1 (additive), group 3: GT; k = 2 or 3 bases, records and scalars side by side in
 * `a` (k records per unit) and `e` (k scalars of elen bytes per unit) */
int oracle_pow_multi(const oracle_pairing *P, int group, int k, const uint8_t *a, const uint8_t *e, size_t elen, uint8_t *out, size_t n) {
  const fpctx *F = &P->Fq;
  if (group == 1) {
    for (size_t i = 0; i < n; i++) {
      pt acc; acc.inf = 1; memset(&acc.x, 0, sizeof acc.x); memset(&acc.y, 0, sizeof acc.y);
      for (int j = 0; j < k; j++) {
        pt A, R; big ex;
        big_from_be(&ex, e + (i * k + j) * elen, elen);
        pt_from_bytes(F, &P->ca, &P->cb, &A, a + (i * k + j) * P->len1);
        if (!A.inf && fp_is0(F, &A.x) && fp_is0(F, &A.y)) A.inf = 1;
        pt_mul(F, &P->ca, &R, &A, &ex);
        pt_add(F, &P->ca, &acc, &acc, &R);
      }
      pt_to_bytes(F, out + i * P->len1, &acc);
    }
    return 0;
  }
  if (group != 3) return 1;
  uint8_t t0[16 * 8 * MAXL], t1[16 * 8 * MAXL], t2[16 * 8 * MAXL];
  if ((size_t) P->lenT > sizeof t0) return 1;
  for (size_t i = 0; i < n; i++) {
    for (int j = 0; j < k; j++) {
      if (oracle_gt_pow(P, a + (i * k + j) * P->lenT, e + (i * k + j) * elen, elen, j ? t1 : t0, 1)) return 1;
      if (j) {
        if (oracle_gt_mul(P, t0, t1, t2, 1)) return 1;
        memcpy(t0, t2, (size_t) P->lenT);
      }
    }
    memcpy(out + i * P->lenT, t0, (size_t) P->lenT);
  }
  return 0;
}
