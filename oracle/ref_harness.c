/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * A small driver that links against the *unmodified* reference PBC library
 * (compiled from /root/reference by oracle/Makefile into oracle/_ref/) and
 *   gen   : emits binary vector files  (inputs in element_to_bytes format +
 *           GT outputs of element_pairing / element_prod_pairing)
 *   kat   : re-checks the reference's only pairing known-answer test
 *           (pbc/pairing_test.pbc:3-10) through the C API
 *   bench : times element_pairing / element_prod_pairing on the host cores
 *           (loop shaped like benchmark/benchmark.c:70-99), one forked
 *           worker per requested core (PBC is not thread safe).
 *
 * Vector file layout (little endian):
 *   char magic[8] = "PBCVEC01"; u32 type_char; u32 n; u32 k; u32 len1; u32 len2; u32 lenT;
 *   u8 in1[n*k*len1]; u8 in2[n*k*len2]; u8 out[n*lenT];
 * Unit u (0..n-1) uses terms u*k .. u*k+k-1; k==1 -> element_pairing,
 * k>1 -> element_prod_pairing.
 *
 * Input distributions (mode):
 *   chain  : P0=from_hash(G1,"pbc-mi355x/P0"), Q0=from_hash(G2,"pbc-mi355x/Q0"),
 *            P_i=(i+1)P0, Q_i=(i+1)Q0 by repeated element_add
 *   random : pbc_random_set_deterministic(seed) BEFORE pairing_init, element_random
 *   edge   : random, but a few units get an identity (off-curve bytes) input
 *   fullorder : points of the whole curve, NOT multiplied by the cofactor (curve_random_no_cofac_solvefory,
 *            ecc/curve.c:405-422, restated through the public accessors): curve_from_bytes (:609-623) accepts
 *            them, and they lie outside the order-r subgroup whenever the curve has a cofactor
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include <sys/wait.h>
#include <sys/time.h>
#include <gmp.h>
#include "pbc.h"

static double now(void) {
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return tv.tv_sec + 1e-6 * tv.tv_usec;
}

static char *slurp(const char *path, size_t *len) {
  FILE *fp = fopen(path, "rb");
  if (!fp) { perror(path); exit(2); }
  char *buf = malloc(1 << 16);
  *len = fread(buf, 1, (1 << 16) - 1, fp);
  buf[*len] = 0;
  fclose(fp);
  return buf;
}

static void init_pairing(pairing_t pairing, const char *path, char *type_out) {
  size_t len;
  char *s = slurp(path, &len);
  if (pairing_init_set_buf(pairing, s, len)) { fprintf(stderr, "pairing init failed\n"); exit(2); }
  const char *t = strstr(s, "type ");
  *type_out = t ? t[5] : '?';
  free(s);
}

static void w32(FILE *fp, uint32_t v) { fwrite(&v, 4, 1, fp); }

/* a random point of the whole curve group: random x until x^3 + ax + b is a square, y = its root, random sign;
 * no cofactor multiplication (ecc/curve.c:405-422 without the element_mul_mpz of :427) */
static void full_order_point(element_t P) {
  element_ptr x = curve_x_coord(P);
  element_t t;
  mpz_t coin;
  element_init_same_as(t, x);
  mpz_init(coin);
  do {
    element_random(x);
    element_square(t, x);
    element_add(t, t, curve_a_coeff(P));
    element_mul(t, t, x);
    element_add(t, t, curve_b_coeff(P));
  } while (!element_is_sqr(t));
  curve_from_x(P, x);
  pbc_mpz_randomb(coin, 1);
  if (mpz_odd_p(coin)) element_neg(P, P);
  element_clear(t);
  mpz_clear(coin);
}

static int cmd_gen(int argc, char **argv) {
  if (argc < 7) { fprintf(stderr, "gen <param> <chain|random|edge> <n> <k> <seed> <out>\n"); return 2; }
  const char *param = argv[1], *mode = argv[2];
  int n = atoi(argv[3]), k = atoi(argv[4]);
  unsigned seed = (unsigned) atoi(argv[5]);
  const char *outp = argv[6];
  pairing_t pairing;
  char type;
  pbc_random_set_deterministic(seed);
  init_pairing(pairing, param, &type);
  int l1 = pairing_length_in_bytes_G1(pairing);
  int l2 = pairing_length_in_bytes_G2(pairing);
  int lt = pairing_length_in_bytes_GT(pairing);
  size_t tot = (size_t) n * k;
  unsigned char *b1 = calloc(tot, l1), *b2 = calloc(tot, l2), *bo = calloc(n, lt);
  element_t *P = malloc(sizeof(element_t) * k), *Q = malloc(sizeof(element_t) * k);
  element_t P0, Q0, out;
  element_init_G1(P0, pairing); element_init_G2(Q0, pairing); element_init_GT(out, pairing);
  for (int j = 0; j < k; j++) { element_init_G1(P[j], pairing); element_init_G2(Q[j], pairing); }
  int chain = !strcmp(mode, "chain"), edge = !strcmp(mode, "edge"), full = !strcmp(mode, "fullorder");
  element_t Pc, Qc;
  element_init_G1(Pc, pairing); element_init_G2(Qc, pairing);
  if (chain) {
    element_from_hash(P0, "pbc-mi355x/P0", 13);
    element_from_hash(Q0, "pbc-mi355x/Q0", 13);
    element_set(Pc, P0); element_set(Qc, Q0);
  }
  for (int u = 0; u < n; u++) {
    for (int j = 0; j < k; j++) {
      size_t idx = (size_t) u * k + j;
      if (chain) {
        element_set(P[j], Pc); element_set(Q[j], Qc);
        element_add(Pc, Pc, P0); element_add(Qc, Qc, Q0);
      } else if (full) {
        full_order_point(P[j]); full_order_point(Q[j]);
      } else {
        element_random(P[j]); element_random(Q[j]);
      }
      element_to_bytes(b1 + idx * l1, P[j]);
      element_to_bytes(b2 + idx * l2, Q[j]);
      if (edge && (u % 5 == 1) && j == (u / 5) % k) {
        /* an off-curve encoding: deserialises to O (ecc/curve.c:618-621) */
        unsigned char *t = (u % 2) ? b1 + idx * l1 : b2 + idx * l2;
        int L = (u % 2) ? l1 : l2;
        t[L - 1] ^= 1;
        if (u % 2) element_from_bytes(P[j], t); else element_from_bytes(Q[j], t);
      }
    }
    if (k == 1) element_pairing(out, P[0], Q[0]);
    else element_prod_pairing(out, P, Q, k);
    element_to_bytes(bo + (size_t) u * lt, out);
  }
  FILE *fp = fopen(outp, "wb");
  if (!fp) { perror(outp); return 2; }
  fwrite("PBCVEC01", 1, 8, fp);
  w32(fp, (uint32_t) type); w32(fp, n); w32(fp, k); w32(fp, l1); w32(fp, l2); w32(fp, lt);
  fwrite(b1, l1, tot, fp); fwrite(b2, l2, tot, fp); fwrite(bo, lt, n, fp);
  fclose(fp);
  fprintf(stderr, "wrote %s: type %c n=%d k=%d len=%d/%d/%d\n", outp, type, n, k, l1, l2, lt);
  return 0;
}

/* pbc/pairing_test.pbc:3-10, restated through the C API */
static int cmd_kat(int argc, char **argv) {
  if (argc < 2) return 2;
  pairing_t pairing; char type;
  init_pairing(pairing, argv[1], &type);
  element_t g, h, e, want;
  element_init_G1(g, pairing); element_init_G2(h, pairing);
  element_init_GT(e, pairing); element_init_GT(want, pairing);
  element_set_str(g, "[2382389466570123849673299401984867521337122094157231907755149435707124249269394670242462497382963719723036281844079382411446883273020125104982896098602669, 2152768906589770702756591740710760107878949212304343787392475836859241438597588807103470081101790991563152395601123682809718038151417122294066319979967168]", 10);
  element_set_str(h, "[5832612417453786541700129157230442590988122495898645678468800815872828277169950107203266157735206975228912899931278160262081308603240860553459187732968543, 5825590786822892934138376868455818413990615826926356662470129700411774690868351658310187202553513693344017463065909279569624651155563430675084173630054336]", 10);
  element_set_str(want, "[1352478452661998164151215014828915385601138645645403926287105573769451214277485326392786454433874957123922454604362337349978217917242114505658729401276644, 2809858014072341042857607405424304552357466023841122154308055820747972163307396014445308786731013691659356362568425895483877936945589613445697089590886519]", 10);
  element_pairing(e, g, h);
  int ok = !element_cmp(e, want);
  printf("KAT %s\n", ok ? "PASS" : "FAIL");
  if (argc >= 3) { /* dump g,h,e bytes as a 1-unit vector file */
    FILE *fp = fopen(argv[2], "wb");
    int l1 = pairing_length_in_bytes_G1(pairing), l2 = pairing_length_in_bytes_G2(pairing), lt = pairing_length_in_bytes_GT(pairing);
    unsigned char buf[1024];
    fwrite("PBCVEC01", 1, 8, fp);
    w32(fp, type); w32(fp, 1); w32(fp, 1); w32(fp, l1); w32(fp, l2); w32(fp, lt);
    element_to_bytes(buf, g); fwrite(buf, 1, l1, fp);
    element_to_bytes(buf, h); fwrite(buf, 1, l2, fp);
    element_to_bytes(buf, want); fwrite(buf, 1, lt, fp);
    fclose(fp);
  }
  return ok ? 0 : 1;
}

/* bench <param> <n_per_worker> <k> <workers>  -> prints JSON */
static int cmd_bench(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "bench <param> <n> <k> <workers>\n"); return 2; }
  int n = atoi(argv[2]), k = atoi(argv[3]), workers = atoi(argv[4]);
  const int pp_mode = k == -1;
  if (pp_mode) k = 1;
  int fds[256][2];
  if (workers > 256) workers = 256;
  double t_all0 = now();
  for (int w = 0; w < workers; w++) {
    if (pipe(fds[w])) return 2;
    pid_t pid = fork();
    if (pid == 0) {
      close(fds[w][0]);
      pairing_t pairing; char type;
      init_pairing(pairing, argv[1], &type);
      element_t *P = malloc(sizeof(element_t) * k), *Q = malloc(sizeof(element_t) * k);
      element_t P0, Q0, out;
      element_init_G1(P0, pairing); element_init_G2(Q0, pairing); element_init_GT(out, pairing);
      element_from_hash(P0, "pbc-mi355x/P0", 13);
      element_from_hash(Q0, "pbc-mi355x/Q0", 13);
      for (int j = 0; j < k; j++) {
        element_init_G1(P[j], pairing); element_init_G2(Q[j], pairing);
        element_mul_si(P[j], P0, j + 1 + w); element_mul_si(Q[j], Q0, j + 1 + w);
      }
      double t0 = now();
      if (pp_mode) {
        /* k = -1: pairing_pp_apply with a fixed first argument, as benchmark/benchmark.c:75-81 times it */
        pairing_pp_t pp;
        pairing_pp_init(pp, P[0], pairing);
        t0 = now();
        for (int i = 0; i < n; i++) {
          pairing_pp_apply(out, Q[0], pp);
          element_add(Q[0], Q[0], Q0);
        }
      } else
      for (int i = 0; i < n; i++) {
        if (k == 1) element_pairing(out, P[0], Q[0]);
        else element_prod_pairing(out, P, Q, k);
        /* step inputs like benchmark/benchmark.c does (fresh inputs every iteration) */
        element_add(P[i % k], P[i % k], P0);
        element_add(Q[i % k], Q[i % k], Q0);
      }
      double dt = now() - t0;
      if (write(fds[w][1], &dt, sizeof dt) != sizeof dt) _exit(3);
      _exit(0);
    }
    close(fds[w][1]);
  }
  double sum_rate = 0, max_dt = 0;
  for (int w = 0; w < workers; w++) {
    double dt = 0;
    if (read(fds[w][0], &dt, sizeof dt) != sizeof dt) { fprintf(stderr, "worker %d failed\n", w); return 3; }
    sum_rate += n / dt;
    if (dt > max_dt) max_dt = dt;
  }
  while (wait(NULL) > 0) {}
  double wall = now() - t_all0;
  printf("{\"units_per_s\": %.3f, \"per_core\": %.3f, \"workers\": %d, \"n_per_worker\": %d, \"k\": %d, \"max_worker_s\": %.4f, \"wall_s\": %.4f}\n",
         sum_rate, sum_rate / workers, workers, n, pp_mode ? -1 : k, max_dt, wall);
  return 0;
}

/* benchg <param> <op> <n> <workers>: the group operations next to the pairing on the host cores, one forked worker per core
 * (the loop of example/bls.c's steps, fresh inputs every iteration).  op: g1mul / g2mul (element_mul_zn with a random
 * scalar), gtpow (element_pow_zn on a pairing value), hashg1 (element_from_hash of a 32-byte digest), g1pp / gtpp
 * (element_pp_pow_zn after one element_pp_init outside the clock), blsverify (one signature check as example/bls.c:64-117
 * does it: a hash and two pairings). */
static int cmd_benchg(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "benchg <param> <op> <n> <workers>\n"); return 2; }
  const char *op = argv[2];
  int n = atoi(argv[3]), workers = atoi(argv[4]);
  int fds[256][2];
  if (workers > 256) workers = 256;
  double t_all0 = now();
  for (int w = 0; w < workers; w++) {
    if (pipe(fds[w])) return 2;
    pid_t pid = fork();
    if (pid == 0) {
      close(fds[w][0]);
      pairing_t pairing; char type;
      pbc_random_set_deterministic(1000u + (unsigned) w);
      init_pairing(pairing, argv[1], &type);
      element_t P, Q, R1, R1b, R2, k, k2, gt, gto, sk, pk, sig, t1, t2;
      element_pp_t pp;
      element_init_G1(P, pairing); element_init_G1(R1, pairing); element_init_G1(sig, pairing); element_init_G1(R1b, pairing);
      element_init_Zr(k2, pairing);
      element_init_G2(Q, pairing); element_init_G2(R2, pairing); element_init_G2(pk, pairing);
      element_init_Zr(k, pairing); element_init_Zr(sk, pairing);
      element_init_GT(gt, pairing); element_init_GT(gto, pairing); element_init_GT(t1, pairing); element_init_GT(t2, pairing);
      element_random(P); element_random(Q); element_random(sk);
      element_random(R1);
      element_pairing(gt, P, Q);
      element_pow_zn(pk, Q, sk);
      element_random(k); element_pow_zn(t1, gt, sk);
      unsigned char digest[32];
      memset(digest, 0x5a, sizeof digest);
      const int is_g1pp = !strcmp(op, "g1pp"), is_gtpp = !strcmp(op, "gtpp");
      if (is_g1pp) element_pp_init(pp, P);
      if (is_gtpp) element_pp_init(pp, gt);
      unsigned bad = 0;
      unsigned char cbuf[2][1024];
      const int is_comp = !strcmp(op, "compress"), is_decomp = !strcmp(op, "decompress");
      element_to_bytes_compressed(cbuf[0], P);
      element_to_bytes_compressed(cbuf[1], R1);
      double t0 = now();
      for (int i = 0; i < n; i++) {
        digest[i & 31] = (unsigned char) (digest[i & 31] * 5 + i + w);
        /* point formats (ecc/curve.c:762-815) alone, alternating between two points / records:
         * compress = element_to_bytes_compressed, decompress = element_from_bytes_compressed (a square root in F_q) */
        if (is_comp) { element_to_bytes_compressed(cbuf[i & 1], (i & 1) ? R1 : P); continue; }
        if (is_decomp) { element_from_bytes_compressed((i & 1) ? R1 : P, cbuf[i & 1]); continue; }
        if (!strcmp(op, "g1add")) { element_add(R1, R1, P); continue; }                      /* curve_mul: one inversion + 3 products */
        if (!strcmp(op, "zrinv")) { element_invert(k2, k); element_add(k, k2, sk); if (element_is0(k)) element_set1(k); continue; }
        if (!strcmp(op, "g1pow2")) { element_random(k); element_random(k2); element_pow2_zn(R1b, P, k, R1, k2); element_add(P, P, R1b); continue; }
        if (!strcmp(op, "gtpow2")) { element_random(k); element_random(k2); element_pow2_zn(gto, gt, k, t1, k2); element_mul(gt, gt, gto); continue; }
        if (!strcmp(op, "g1mul")) { element_random(k); element_mul_zn(R1, P, k); element_add(P, P, R1); }
        else if (!strcmp(op, "g2mul")) { element_random(k); element_mul_zn(R2, Q, k); element_add(Q, Q, R2); }
        else if (!strcmp(op, "gtpow")) { element_random(k); element_pow_zn(gto, gt, k); element_mul(gt, gt, gto); }
        else if (!strcmp(op, "hashg1")) { element_from_hash(R1, digest, 32); }
        else if (is_g1pp) { element_random(k); element_pp_pow_zn(R1, k, pp); }
        else if (is_gtpp) { element_random(k); element_pp_pow_zn(gto, k, pp); }
        else if (!strcmp(op, "blsverify")) {
          element_from_hash(R1, digest, 32);
          if (i == 0) element_pow_zn(sig, R1, sk);       /* (signing is not what this loop times: one signature, re-made per digest below) */
          element_pow_zn(sig, R1, sk);
          element_pairing(t1, sig, Q);
          element_pairing(t2, R1, pk);
          bad += element_cmp(t1, t2) != 0;
        } else _exit(4);
      }
      double dt = now() - t0;
      if (bad) _exit(5);
      if (write(fds[w][1], &dt, sizeof dt) != sizeof dt) _exit(3);
      _exit(0);
    }
    close(fds[w][1]);
  }
  double sum_rate = 0, max_dt = 0;
  for (int w = 0; w < workers; w++) {
    double dt = 0;
    if (read(fds[w][0], &dt, sizeof dt) != sizeof dt) { fprintf(stderr, "worker %d failed\n", w); return 3; }
    sum_rate += n / dt;
    if (dt > max_dt) max_dt = dt;
  }
  while (wait(NULL) > 0) {}
  double wall = now() - t_all0;
  printf("{\"units_per_s\": %.3f, \"per_core\": %.3f, \"workers\": %d, \"n_per_worker\": %d, \"op\": \"%s\", \"max_worker_s\": %.4f, \"wall_s\": %.4f}\n",
         sum_rate, sum_rate / workers, workers, n, op, max_dt, wall);
  return 0;
}

/* ppow <param> <group 1|2|3> <n> <seed> <out>: element_pp_init on one random element of G1 / G2 / GT (a pairing value) and
 * element_pp_pow_zn for n random scalars (the last ones 0 -- GT only; 2 for the curve groups --, 1, r - 1).  File: in1 = the base (one record, repeated n times),
 * in2 = scalars, out = powers. */
static int cmd_ppow(int argc, char **argv) {
  if (argc < 6) { fprintf(stderr, "ppow <param> <group> <n> <seed> <out>\n"); return 2; }
  int group = atoi(argv[2]), n = atoi(argv[3]);
  pairing_t pairing; char type;
  pbc_random_set_deterministic((unsigned) atoi(argv[4]));
  init_pairing(pairing, argv[1], &type);
  int lp = group == 1 ? pairing_length_in_bytes_G1(pairing) : group == 2 ? pairing_length_in_bytes_G2(pairing) : pairing_length_in_bytes_GT(pairing);
  int lz = pairing_length_in_bytes_Zr(pairing);
  unsigned char *in = malloc((size_t) n * lp), *zs = malloc((size_t) n * lz), *out = malloc((size_t) n * lp);
  element_t B, R, k, P, Q;
  element_pp_t pp;
  element_init_G1(P, pairing); element_init_G2(Q, pairing);
  if (group == 1) { element_init_G1(B, pairing); element_init_G1(R, pairing); element_random(B); }
  else if (group == 2) { element_init_G2(B, pairing); element_init_G2(R, pairing); element_random(B); }
  else { element_init_GT(B, pairing); element_init_GT(R, pairing); element_random(P); element_random(Q); element_pairing(B, P, Q); }
  element_init_Zr(k, pairing);
  element_pp_init(pp, B);
  for (int i = 0; i < n; i++) {
    element_random(k);
    /* (power 0 only in GT: the identity of a curve group has no wire format -- curve_to_bytes, ecc/curve.c:595-601, writes
     * whatever coordinates the element held before) */
    if (i == n - 3) { if (group == 3) element_set0(k); else element_set_si(k, 2); }
    if (i == n - 2) element_set1(k);
    if (i == n - 1) { element_set1(k); element_neg(k, k); }
    element_pp_pow_zn(R, k, pp);
    element_to_bytes(in + (size_t) i * lp, B);
    element_to_bytes(zs + (size_t) i * lz, k);
    element_to_bytes(out + (size_t) i * lp, R);
  }
  element_pp_clear(pp);
  FILE *fp = fopen(argv[5], "wb");
  fwrite("PBCVEC01", 1, 8, fp);
  w32(fp, (uint32_t) type); w32(fp, n); w32(fp, 1); w32(fp, lp); w32(fp, lz); w32(fp, lp);
  fwrite(in, lp, n, fp); fwrite(zs, lz, n, fp); fwrite(out, lp, n, fp);
  fclose(fp);
  fprintf(stderr, "wrote %s: type %c group %d n=%d\n", argv[5], type, group, n);
  return 0;
}

/* bls <param> <n> <seed> <out>: the flow of example/bls.c (:41-117) for n messages under one key: g random in G2, secret key
 * in Zr, public key g^sk, h_i = element_from_hash(digest_i), sig_i = h_i^sk, each checked e(sig_i, g) == e(h_i, pk) here.
 * File "PBCBLS01": u32 type, n, hlen, len G1, len G2, len Zr; digests; h_i; sig_i; g; pk; sk. */
static int cmd_bls(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "bls <param> <n> <seed> <out>\n"); return 2; }
  int n = atoi(argv[2]);
  const int hlen = 32;
  unsigned seed = (unsigned) atoi(argv[3]);
  pairing_t pairing; char type;
  pbc_random_set_deterministic(seed);
  init_pairing(pairing, argv[1], &type);
  int l1 = pairing_length_in_bytes_G1(pairing), l2 = pairing_length_in_bytes_G2(pairing), lz = pairing_length_in_bytes_Zr(pairing);
  unsigned char *dg = malloc((size_t) n * hlen), *hs = malloc((size_t) n * l1), *sg = malloc((size_t) n * l1);
  unsigned char *gb = malloc(l2), *pkb = malloc(l2), *skb = malloc(lz);
  uint64_t st = 0x9e3779b97f4a7c15ull * (seed + 7);
  for (size_t i = 0; i < (size_t) n * hlen; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; dg[i] = (unsigned char) (st >> 24); }
  element_t g, pk, sk, h, sig, t1, t2;
  element_init_G2(g, pairing); element_init_G2(pk, pairing); element_init_Zr(sk, pairing);
  element_init_G1(h, pairing); element_init_G1(sig, pairing); element_init_GT(t1, pairing); element_init_GT(t2, pairing);
  element_random(g); element_random(sk);
  element_pow_zn(pk, g, sk);
  for (int i = 0; i < n; i++) {
    element_from_hash(h, dg + (size_t) i * hlen, hlen);
    element_pow_zn(sig, h, sk);
    element_pairing(t1, sig, g);
    element_pairing(t2, h, pk);
    if (element_cmp(t1, t2)) { fprintf(stderr, "signature %d does not verify\n", i); return 3; }
    element_to_bytes(hs + (size_t) i * l1, h);
    element_to_bytes(sg + (size_t) i * l1, sig);
  }
  element_to_bytes(gb, g); element_to_bytes(pkb, pk); element_to_bytes(skb, sk);
  FILE *fp = fopen(argv[4], "wb");
  fwrite("PBCBLS01", 1, 8, fp);
  w32(fp, (uint32_t) type); w32(fp, n); w32(fp, hlen); w32(fp, l1); w32(fp, l2); w32(fp, lz);
  fwrite(dg, hlen, n, fp); fwrite(hs, l1, n, fp); fwrite(sg, l1, n, fp);
  fwrite(gb, l2, 1, fp); fwrite(pkb, l2, 1, fp); fwrite(skb, lz, 1, fp);
  fclose(fp);
  fprintf(stderr, "wrote %s: type %c n=%d\n", argv[4], type, n);
  return 0;
}

/* hash <param> <n> <hlen> <seed> <out>: element_from_hash(G1) on n pseudo-random hlen-byte digests.
 * Vector file: in1 = digests (len1 = hlen), in2 empty (len2 = 0), out = G1 bytes (lenT = len G1). */
/* gmul <param> <group 1|2> <n> <seed> <out>: out_i = [k_i] P_i (element_mul_zn) for random points of
 * G1 / G2 and random scalars; the last two scalars are 1 and r - 1.  File: points, scalars, results. */
static int cmd_gmul(int argc, char **argv) {
  if (argc < 6) { fprintf(stderr, "gmul <param> <group> <n> <seed> <out> [full]\n"); return 2; }
  int group = atoi(argv[2]), n = atoi(argv[3]);
  const int full = argc > 6 && !strcmp(argv[6], "full");   /* points of the whole curve (no cofactor multiplication) */
  unsigned seed = (unsigned) atoi(argv[4]);
  pairing_t pairing; char type;
  pbc_random_set_deterministic(seed);
  init_pairing(pairing, argv[1], &type);
  int lp = group == 1 ? pairing_length_in_bytes_G1(pairing) : pairing_length_in_bytes_G2(pairing);
  int lz = pairing_length_in_bytes_Zr(pairing);
  unsigned char *in = malloc((size_t) n * lp), *zs = malloc((size_t) n * lz), *out = malloc((size_t) n * lp);
  element_t P, R, k;
  if (group == 1) { element_init_G1(P, pairing); element_init_G1(R, pairing); }
  else { element_init_G2(P, pairing); element_init_G2(R, pairing); }
  element_init_Zr(k, pairing);
  for (int i = 0; i < n; i++) {
    if (full) full_order_point(P); else element_random(P);
    element_random(k);
    if (i == n - 2) element_set1(k);
    if (i == n - 1) { element_set1(k); element_neg(k, k); }
    element_mul_zn(R, P, k);
    element_to_bytes(in + (size_t) i * lp, P);
    element_to_bytes(zs + (size_t) i * lz, k);
    element_to_bytes(out + (size_t) i * lp, R);
  }
  FILE *fp = fopen(argv[5], "wb");
  fwrite("PBCVEC01", 1, 8, fp);
  w32(fp, (uint32_t) type); w32(fp, n); w32(fp, 1); w32(fp, lp); w32(fp, lz); w32(fp, lp);
  fwrite(in, lp, n, fp); fwrite(zs, lz, n, fp); fwrite(out, lp, n, fp);
  fclose(fp);
  fprintf(stderr, "wrote %s: type %c group %d n=%d\n", argv[5], type, group, n);
  return 0;
}

/* compress <param> <n> <seed> <out>: random G1 points in element_to_bytes form and their
 * element_to_bytes_compressed form (x || sign byte, ecc/curve.c:762-773); each compressed record is
 * fed back through element_from_bytes_compressed (:800-815) and must reproduce the point. */
static int cmd_compress(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "compress <param> <n> <seed> <out> [group]\n"); return 2; }
  int n = atoi(argv[2]);
  unsigned seed = (unsigned) atoi(argv[3]);
  const int group = argc > 5 ? atoi(argv[5]) : 1;
  pairing_t pairing; char type;
  pbc_random_set_deterministic(seed);
  init_pairing(pairing, argv[1], &type);
  int lp = group == 2 ? pairing_length_in_bytes_G2(pairing) : pairing_length_in_bytes_G1(pairing);
  element_t P, R;
  if (group == 2) { element_init_G2(P, pairing); element_init_G2(R, pairing); }
  else { element_init_G1(P, pairing); element_init_G1(R, pairing); }
  int lc = group == 2 ? pairing_length_in_bytes_compressed_G2(pairing) : pairing_length_in_bytes_compressed_G1(pairing);
  unsigned char *in = malloc((size_t) n * lp), *out = malloc((size_t) n * lc);
  for (int i = 0; i < n; i++) {
    element_random(P);
    element_to_bytes(in + (size_t) i * lp, P);
    element_to_bytes_compressed(out + (size_t) i * lc, P);
    element_from_bytes_compressed(R, out + (size_t) i * lc);
    if (element_cmp(P, R)) { fprintf(stderr, "round trip failed at %d\n", i); return 1; }
  }
  FILE *fp = fopen(argv[4], "wb");
  fwrite("PBCVEC01", 1, 8, fp);
  w32(fp, (uint32_t) type); w32(fp, n); w32(fp, 1); w32(fp, lp); w32(fp, 0); w32(fp, lc);
  fwrite(in, lp, n, fp); fwrite(out, lc, n, fp);
  fclose(fp);
  fprintf(stderr, "wrote %s: type %c n=%d %d -> %d bytes\n", argv[4], type, n, lp, lc);
  return 0;
}

/* xonly <param> <n> <seed> <out>: points hashed from a counter (independent of the generator state), their
 * element_to_bytes_x_only form (x alone, ecc/curve.c:821-827) and what element_from_bytes_x_only (:829-836)
 * rebuilds from it: y is whichever root element_sqrt returns.  For q = 3 mod 4 that is x^((q+1)/4); for
 * q = 1 mod 4 element_tonelli (arith/field.c:672-720) works from a randomly drawn non-residue
 * (field_gen_nqr), so <seed> (set before the pairing is initialised) lets a caller see whether the root moves. */
static int cmd_xonly(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "xonly <param> <n> <seed> <out> [group]\n"); return 2; }
  int n = atoi(argv[2]);
  const int group = argc > 5 ? atoi(argv[5]) : 1;
  pairing_t pairing; char type;
  pbc_random_set_deterministic((unsigned) atoi(argv[3]));
  init_pairing(pairing, argv[1], &type);
  int lp = group == 2 ? pairing_length_in_bytes_G2(pairing) : pairing_length_in_bytes_G1(pairing);
  element_t P, R;
  if (group == 2) { element_init_G2(P, pairing); element_init_G2(R, pairing); }
  else { element_init_G1(P, pairing); element_init_G1(R, pairing); }
  int lx = group == 2 ? pairing_length_in_bytes_x_only_G2(pairing) : pairing_length_in_bytes_x_only_G1(pairing);
  unsigned char *in = malloc((size_t) n * lp), *xs = malloc((size_t) n * lx), *out = malloc((size_t) n * lp);
  for (int i = 0; i < n; i++) {
    char msg[32];
    int ml = snprintf(msg, sizeof msg, "x-only/%d", i);
    element_from_hash(P, msg, ml);
    element_to_bytes(in + (size_t) i * lp, P);
    if (element_to_bytes_x_only(xs + (size_t) i * lx, P) != lx) return 1;
    element_from_bytes_x_only(R, xs + (size_t) i * lx);
    element_to_bytes(out + (size_t) i * lp, R);
  }
  FILE *fp = fopen(argv[4], "wb");
  fwrite("PBCVEC01", 1, 8, fp);
  w32(fp, (uint32_t) type); w32(fp, n); w32(fp, 1); w32(fp, lp); w32(fp, lx); w32(fp, lp);
  fwrite(in, lp, n, fp); fwrite(xs, lx, n, fp); fwrite(out, lp, n, fp);
  fclose(fp);
  fprintf(stderr, "wrote %s: type %c n=%d x-only %d bytes\n", argv[4], type, n, lx);
  return 0;
}

/* gena <rbits> <qbits> <seed> <out.param>: a fresh type a parameter set (pbc_param_init_a_gen,
 * ecc/a_param.c:1504-1562) written with pbc_param_out_str */
static int cmd_gena(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "gena <rbits> <qbits> <seed> <out.param>\n"); return 2; }
  pbc_param_t par;
  pbc_random_set_deterministic((unsigned) atoi(argv[3]));
  pbc_param_init_a_gen(par, atoi(argv[1]), atoi(argv[2]));
  FILE *fp = fopen(argv[4], "w");
  if (!fp) { perror(argv[4]); return 2; }
  pbc_param_out_str(fp, par);
  fclose(fp);
  pbc_param_clear(par);
  return 0;
}

/* gena1 <prime bits> <seed> <out.param>: type a1 parameters for n = p1 p2, two random primes of the
 * given size (pbc_param_init_a1_gen, ecc/a_param.c:2300-2321) */
static int cmd_gena1(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "gena1 <prime bits> <seed> <out.param>\n"); return 2; }
  int bits = atoi(argv[1]);
  pbc_random_set_deterministic((unsigned) atoi(argv[2]));
  mpz_t p1, p2, n;
  mpz_init(p1); mpz_init(p2); mpz_init(n);
  pbc_mpz_randomb(p1, bits); mpz_setbit(p1, bits - 1); mpz_nextprime(p1, p1);
  pbc_mpz_randomb(p2, bits); mpz_setbit(p2, bits - 1); mpz_nextprime(p2, p2);
  mpz_mul(n, p1, p2);
  pbc_param_t par;
  pbc_param_init_a1_gen(par, n);
  FILE *fp = fopen(argv[3], "w");
  if (!fp) { perror(argv[3]); return 2; }
  pbc_param_out_str(fp, par);
  fclose(fp);
  pbc_param_clear(par);
  mpz_clear(p1); mpz_clear(p2); mpz_clear(n);
  return 0;
}
/* gene <rbits> <qbits> <seed> <out.param>: type e parameters (pbc_param_init_e_gen, ecc/e_param.c:908-1005) */
static int cmd_gene(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "gene <rbits> <qbits> <seed> <out.param>\n"); return 2; }
  pbc_param_t par;
  pbc_random_set_deterministic((unsigned) atoi(argv[3]));
  pbc_param_init_e_gen(par, atoi(argv[1]), atoi(argv[2]));
  FILE *fp = fopen(argv[4], "w");
  if (!fp) { perror(argv[4]); return 2; }
  pbc_param_out_str(fp, par);
  fclose(fp);
  pbc_param_clear(par);
  return 0;
}

/* genf <bits> <seed> <out.param>: type f (BN) parameters (pbc_param_init_f_gen, ecc/f_param.c:459-577) */
static int cmd_genf(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "genf <bits> <seed> <out.param>\n"); return 2; }
  pbc_param_t par;
  pbc_random_set_deterministic((unsigned) atoi(argv[2]));
  pbc_param_init_f_gen(par, atoi(argv[1]));
  FILE *fp = fopen(argv[3], "w");
  if (!fp) { perror(argv[3]); return 2; }
  pbc_param_out_str(fp, par);
  fclose(fp);
  pbc_param_clear(par);
  return 0;
}

static int cmd_hash(int argc, char **argv) {
  if (argc < 6) { fprintf(stderr, "hash <param> <n> <hlen> <seed> <out> [group]\n"); return 2; }
  int n = atoi(argv[2]), hlen = atoi(argv[3]);
  const int group = argc > 6 ? atoi(argv[6]) : 1;
  unsigned seed = (unsigned) atoi(argv[4]);
  pairing_t pairing; char type;
  pbc_random_set_deterministic(seed);
  init_pairing(pairing, argv[1], &type);
  int l1 = group == 2 ? pairing_length_in_bytes_G2(pairing) : pairing_length_in_bytes_G1(pairing);
  unsigned char *in = malloc((size_t) n * hlen), *out = malloc((size_t) n * l1);
  uint64_t st = 0x9e3779b97f4a7c15ull * (seed + 1);
  for (size_t i = 0; i < (size_t) n * hlen; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; in[i] = (unsigned char) (st >> 24); }
  element_t h;
  if (group == 2) element_init_G2(h, pairing); else element_init_G1(h, pairing);
  for (int i = 0; i < n; i++) {
    element_from_hash(h, in + (size_t) i * hlen, hlen);
    element_to_bytes(out + (size_t) i * l1, h);
  }
  FILE *fp = fopen(argv[5], "wb");
  fwrite("PBCVEC01", 1, 8, fp);
  w32(fp, (uint32_t) type); w32(fp, n); w32(fp, 1); w32(fp, hlen); w32(fp, 0); w32(fp, l1);
  fwrite(in, hlen, n, fp); fwrite(out, l1, n, fp);
  fclose(fp);
  fprintf(stderr, "wrote %s: type %c n=%d hlen=%d\n", argv[5], type, n, hlen);
  return 0;
}

/* finalpow <param> <n> <seed> <out>: n random elements of GT's underlying field (element_random on the wrapped
 * element, ecc/pairing.c:135-283) and what pairing->finalpow (include/pbc_pairing.h:41) makes of them.
 * File: inputs (GT format), nothing, outputs. */
static int cmd_finalpow(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "finalpow <param> <n> <seed> <out>\n"); return 2; }
  int n = atoi(argv[2]);
  pairing_t pairing; char type;
  pbc_random_set_deterministic((unsigned) atoi(argv[3]));
  init_pairing(pairing, argv[1], &type);
  int lt = pairing_length_in_bytes_GT(pairing);
  unsigned char *in = malloc((size_t) n * lt), *out = malloc((size_t) n * lt);
  element_t e;
  element_init_GT(e, pairing);
  for (int i = 0; i < n; i++) {
    element_random((element_ptr) e->data);               /* the element inside the GT wrapper */
    element_to_bytes(in + (size_t) i * lt, e);
    pairing->finalpow(e);
    element_to_bytes(out + (size_t) i * lt, e);
  }
  FILE *fp = fopen(argv[4], "wb");
  fwrite("PBCVEC01", 1, 8, fp);
  w32(fp, (uint32_t) type); w32(fp, n); w32(fp, 1); w32(fp, lt); w32(fp, 0); w32(fp, lt);
  fwrite(in, lt, n, fp); fwrite(out, lt, n, fp);
  fclose(fp);
  fprintf(stderr, "wrote %s: type %c n=%d finalpow\n", argv[4], type, n);
  return 0;
}

/* rdep <param> <seed>: two pairing objects from the same parameter text (type e draws its auxiliary point R from
 * the generator at init, e_param.c:866-870, so the two objects hold different R), the same input bytes: prints
 * whether element_pairing agrees for a point pair of the order-r subgroup and for a pair of the whole curve.  For
 * k = 1 the value f_P(Q+R)/f_P(R) is independent of R only when r P = O. */
static int cmd_rdep(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "rdep <param> <seed>\n"); return 2; }
  pairing_t pa, pb; char type;
  pbc_random_set_deterministic((unsigned) atoi(argv[2]));
  init_pairing(pa, argv[1], &type);
  init_pairing(pb, argv[1], &type);
  for (int full = 0; full < 2; full++) {
    element_t P, Q, P2, Q2, ea, eb;
    unsigned char b1[2048], b2[2048];
    element_init_G1(P, pa); element_init_G2(Q, pa); element_init_GT(ea, pa);
    element_init_G1(P2, pb); element_init_G2(Q2, pb); element_init_GT(eb, pb);
    if (full) { full_order_point(P); full_order_point(Q); } else { element_random(P); element_random(Q); }
    element_to_bytes(b1, P); element_to_bytes(b2, Q);
    element_from_bytes(P2, b1); element_from_bytes(Q2, b2);
    element_pairing(ea, P, Q);
    element_pairing(eb, P2, Q2);
    element_to_bytes(b1, ea); element_to_bytes(b2, eb);
    printf("%s points: two objects %s\n", full ? "whole-curve" : "subgroup",
           memcmp(b1, b2, pairing_length_in_bytes_GT(pa)) ? "DISAGREE" : "agree");
  }
  return 0;
}

/* text <param-file> <count> <seed>: golden vectors of the text formats.  One line per element:
 *   <group> <hex of element_to_bytes> <element_snprint text>      group: 0 Zr, 1 G1, 2 G2, 3 GT
 * preceded by the lines of pbc_param_out_str (each prefixed "P "); GT elements are pairings of the G1 / G2 elements,
 * every fourth G1 / G2 element is O (element_set0). */
static int cmd_text(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "text <param> <count> <seed>\n"); return 2; }
  int count = atoi(argv[2]);
  pbc_random_set_deterministic((unsigned) atoi(argv[3]));
  static char buf[1 << 16], txt[1 << 16];
  FILE *fp = fopen(argv[1], "r");
  if (!fp) { perror(argv[1]); return 1; }
  size_t len = fread(buf, 1, sizeof buf - 1, fp);
  fclose(fp);
  buf[len] = 0;
  pbc_param_t par;
  if (pbc_param_init_set_buf(par, buf, len)) return 1;
  {
    char *pt = NULL;
    size_t pl = 0;
    FILE *ms = open_memstream(&pt, &pl);
    pbc_param_out_str(ms, par);
    fclose(ms);
    for (char *line = strtok(pt, "\n"); line; line = strtok(NULL, "\n")) printf("P %s\n", line);
    free(pt);
  }
  pairing_t pairing;
  pairing_init_pbc_param(pairing, par);
  element_t z, g1, g2, gt;
  element_init_Zr(z, pairing);
  element_init_G1(g1, pairing);
  element_init_G2(g2, pairing);
  element_init_GT(gt, pairing);
  static unsigned char bytes[4096];
  for (int i = 0; i < count; i++) {
    element_random(z);
    element_random(g1);
    element_random(g2);
    element_pairing(gt, g1, g2);
    if (i % 4 == 3) { element_set0(g1); element_set0(g2); }
    element_t *e[4] = {&z, &g1, &g2, &gt};
    for (int g = 0; g < 4; g++) {
      int n = element_to_bytes(bytes, *e[g]);
      if ((g == 1 || g == 2) && element_is0(*e[g])) memset(bytes, 0, (size_t) n);   /* O has no defined coordinates */
      int t = element_snprint(txt, sizeof txt, *e[g]);
      if (t < 0 || (size_t) t >= sizeof txt) return 1;
      printf("%d ", g);
      for (int b = 0; b < n; b++) printf("%02x", bytes[b]);
      printf(" %s\n", txt);
    }
  }
  return 0;
}


/* ---- round 5: the group law, Z_r arithmetic and multi-exponentiations as fixtures ----------------------------------
 * Record container "PBCREC01": u32 type, u32 count, then per array u32 rows, u32 width, then the arrays' bytes in order
 * (oracle/__init__.py Rec). */
typedef struct { unsigned char *p; uint32_t rows, width; } recarr;
static recarr rec_new(uint32_t rows, uint32_t width) { recarr a = {calloc((size_t) rows * width + 1, 1), rows, width}; return a; }
static int rec_write(const char *path, char type, recarr *arr, int count) {
  FILE *fp = fopen(path, "wb");
  if (!fp) { perror(path); return 1; }
  fwrite("PBCREC01", 1, 8, fp);
  w32(fp, (uint32_t) type); w32(fp, (uint32_t) count);
  for (int i = 0; i < count; i++) { w32(fp, arr[i].rows); w32(fp, arr[i].width); }
  for (int i = 0; i < count; i++) fwrite(arr[i].p, arr[i].width, arr[i].rows, fp);
  fclose(fp);
  return 0;
}
/* a record of G1 / G2 / GT.  O of a curve group is written as zero bytes -- the convention of include/pbc_hip.h; the
 * reference's curve_to_bytes (ecc/curve.c:603-609) ignores inf_flag and writes whatever coordinates the element last held */
static void put_rec(unsigned char *dst, element_t e, int group, int len) {
  if (group != 3 && element_is0(e)) memset(dst, 0, (size_t) len); else element_to_bytes(dst, e);
}
static void init_group(element_t e, pairing_t pairing, int group) {
  if (group == 1) element_init_G1(e, pairing); else if (group == 2) element_init_G2(e, pairing); else element_init_GT(e, pairing);
}
/* gops <param> <group 1|2> <n> <seed> <out>: element_add / element_sub / element_neg / element_double on G1 / G2
 * (curve_mul ecc/curve.c:153-207, curve_invert :79-100, curve_double :102-151).  Arrays: A, B, A+B, A-B, -A, 2A.
 * The last rows are the special cases of the group law: B = A, B = -A, A = O, B = O, A = B = O; the row before them
 * takes points of the whole curve (no cofactor multiplication). */
static int cmd_gops(int argc, char **argv) {
  if (argc < 6) { fprintf(stderr, "gops <param> <group> <n> <seed> <out>\n"); return 2; }
  int group = atoi(argv[2]), n = atoi(argv[3]);
  pairing_t pairing; char type;
  pbc_random_set_deterministic((unsigned) atoi(argv[4]));
  init_pairing(pairing, argv[1], &type);
  int lp = group == 1 ? pairing_length_in_bytes_G1(pairing) : pairing_length_in_bytes_G2(pairing);
  recarr arr[6];
  for (int i = 0; i < 6; i++) arr[i] = rec_new(n, lp);
  element_t A, B, R;
  init_group(A, pairing, group); init_group(B, pairing, group); init_group(R, pairing, group);
  for (int i = 0; i < n; i++) {
    element_random(A); element_random(B);
    if (i == n - 6) { full_order_point(A); full_order_point(B); }
    if (i == n - 5) element_set(B, A);
    if (i == n - 4) element_neg(B, A);
    if (i == n - 3 || i == n - 1) element_set0(A);
    if (i == n - 2 || i == n - 1) element_set0(B);
    put_rec(arr[0].p + (size_t) i * lp, A, group, lp);
    put_rec(arr[1].p + (size_t) i * lp, B, group, lp);
    element_add(R, A, B); put_rec(arr[2].p + (size_t) i * lp, R, group, lp);
    element_sub(R, A, B); put_rec(arr[3].p + (size_t) i * lp, R, group, lp);
    element_neg(R, A); put_rec(arr[4].p + (size_t) i * lp, R, group, lp);
    element_double(R, A); put_rec(arr[5].p + (size_t) i * lp, R, group, lp);
  }
  if (rec_write(argv[5], type, arr, 6)) return 1;
  fprintf(stderr, "wrote %s: type %c group %d n=%d gops\n", argv[5], type, group, n);
  return 0;
}
/* zrops <param> <n> <hlen> <seed> <out>: Z_r arithmetic (the F_p back end on the modulus r; example/zss.c:40-41 adds and
 * inverts there, example/hess.c:63 multiplies) -- arrays: A, B, A+B, A-B, A*B, 1/A, -A, 2A, A/2, A/B, digests,
 * element_from_hash(digest).  Rows n-3 .. n-1: A = 1, A = r - 1, B = A.  A, B are never 0 (1/0 is outside the contract). */
static int cmd_zrops(int argc, char **argv) {
  if (argc < 6) { fprintf(stderr, "zrops <param> <n> <hlen> <seed> <out>\n"); return 2; }
  int n = atoi(argv[2]), hlen = atoi(argv[3]);
  pairing_t pairing; char type;
  pbc_random_set_deterministic((unsigned) atoi(argv[4]));
  init_pairing(pairing, argv[1], &type);
  int lz = pairing_length_in_bytes_Zr(pairing);
  recarr arr[12];
  for (int i = 0; i < 12; i++) arr[i] = rec_new(n, i == 10 ? hlen : lz);
  element_t a, b, r;
  element_init_Zr(a, pairing); element_init_Zr(b, pairing); element_init_Zr(r, pairing);
  mpz_t bytes;
  mpz_init(bytes);
  for (int i = 0; i < n; i++) {
    do element_random(a); while (element_is0(a));
    do element_random(b); while (element_is0(b));
    if (i == n - 3) element_set1(a);
    if (i == n - 2) { element_set1(a); element_neg(a, a); }
    if (i == n - 1) element_set(b, a);
    size_t o = (size_t) i * lz;
    element_to_bytes(arr[0].p + o, a);
    element_to_bytes(arr[1].p + o, b);
    element_add(r, a, b); element_to_bytes(arr[2].p + o, r);
    element_sub(r, a, b); element_to_bytes(arr[3].p + o, r);
    element_mul(r, a, b); element_to_bytes(arr[4].p + o, r);
    element_invert(r, a); element_to_bytes(arr[5].p + o, r);
    element_neg(r, a); element_to_bytes(arr[6].p + o, r);
    element_double(r, a); element_to_bytes(arr[7].p + o, r);
    element_halve(r, a); element_to_bytes(arr[8].p + o, r);
    element_div(r, a, b); element_to_bytes(arr[9].p + o, r);
    pbc_mpz_randomb(bytes, 8 * hlen);
    unsigned char *d = arr[10].p + (size_t) i * hlen;
    for (int j = 0; j < hlen; j++) { d[j] = (unsigned char) mpz_fdiv_ui(bytes, 256); mpz_fdiv_q_2exp(bytes, bytes, 8); }
    element_from_hash(r, d, hlen); element_to_bytes(arr[11].p + o, r);
  }
  if (rec_write(argv[5], type, arr, 12)) return 1;
  fprintf(stderr, "wrote %s: type %c n=%d zrops\n", argv[5], type, n);
  return 0;
}
/* pow23 <param> <group 1|2|3> <n> <seed> <out>: element_pow2_zn / element_pow3_zn (include/pbc_field.h:496-531,
 * arith/field.c:153-241) -- arrays: A1, A2, A3, N1, N2, N3, A1^N1 A2^N2, A1^N1 A2^N2 A3^N3 (additive notation on the
 * curves).  Rows n-4 .. n-1: N2 = 0; A2 = A1; A2 = 1 / A1 with N2 = N1 (the pow2 value is the identity); N1 = N2 = N3 = r - 1. */
static int cmd_pow23(int argc, char **argv) {
  if (argc < 6) { fprintf(stderr, "pow23 <param> <group> <n> <seed> <out>\n"); return 2; }
  int group = atoi(argv[2]), n = atoi(argv[3]);
  pairing_t pairing; char type;
  pbc_random_set_deterministic((unsigned) atoi(argv[4]));
  init_pairing(pairing, argv[1], &type);
  int lp = group == 1 ? pairing_length_in_bytes_G1(pairing) : group == 2 ? pairing_length_in_bytes_G2(pairing) : pairing_length_in_bytes_GT(pairing);
  int lz = pairing_length_in_bytes_Zr(pairing);
  recarr arr[8];
  for (int i = 0; i < 8; i++) arr[i] = rec_new(n, (i >= 3 && i < 6) ? lz : lp);
  element_t A[3], N[3], R;
  for (int j = 0; j < 3; j++) { init_group(A[j], pairing, group); element_init_Zr(N[j], pairing); }
  init_group(R, pairing, group);
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < 3; j++) { element_random(A[j]); element_random(N[j]); }
    if (i == n - 4) element_set0(N[1]);
    if (i == n - 3) element_set(A[1], A[0]);
    if (i == n - 2) { element_invert(A[1], A[0]); element_set(N[1], N[0]); }
    if (i == n - 1) for (int j = 0; j < 3; j++) { element_set1(N[j]); element_neg(N[j], N[j]); }
    for (int j = 0; j < 3; j++) {
      put_rec(arr[j].p + (size_t) i * lp, A[j], group, lp);
      element_to_bytes(arr[3 + j].p + (size_t) i * lz, N[j]);
    }
    element_pow2_zn(R, A[0], N[0], A[1], N[1]); put_rec(arr[6].p + (size_t) i * lp, R, group, lp);
    element_pow3_zn(R, A[0], N[0], A[1], N[1], A[2], N[2]); put_rec(arr[7].p + (size_t) i * lp, R, group, lp);
  }
  if (rec_write(argv[5], type, arr, 8)) return 1;
  fprintf(stderr, "wrote %s: type %c group %d n=%d pow23\n", argv[5], type, group, n);
  return 0;
}

/* ---- soak: SURVEY 8d's full-parity distribution, on every host core -------------------------------------------------
 * soak <param> <n_random> <k> <seed> <out> <workers> <rbits>
 *   n_random units of k terms with uniformly random inputs (pbc_random_set_deterministic BEFORE pairing_init, then
 *   element_random: arith/random.c:80-83, ecc/curve.c:430-438), split over `workers` forked processes (worker w draws
 *   from seed + 7919 (w + 1)); and -- k == 1 only -- a block of CRAFTED units appended by the parent:
 *     * points of the curve / the twist whose x coordinate is q - 1, q - 2, 1, 2, (q +- 1) / 2, or whose MONTGOMERY
 *       residue x R mod q (R = 2^rbits, the radix of the device library's 29-bit limb form) has limbs that are all ones,
 *       all zero but one, alternating, or a single bit on a limb boundary -- nudged upwards until x^3 + a x + b is a square
 *       (curve_from_x; such points lie on the whole curve, not in the order-r subgroup, which curve_from_bytes accepts);
 *       on a twist over F_q^d either every coefficient of x carries the pattern or only the first (the rest zero);
 *     * records whose coordinates are written as v + t q >= q (fp_from_bytes reduces them, arith/montfp.c:498-517).
 *   Every crafted point meets a random partner, and crafted G1 points meet crafted G2 points.
 * File: the vector layout above with n = n_random + crafted; stdout: one JSON line with the counts. */
#include <sys/mman.h>
static void *shared_alloc(size_t n) {
  void *p = mmap(NULL, n ? n : 1, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) { perror("mmap"); exit(2); }
  return p;
}
/* pattern number t -> an integer below q; *mont says whether it is meant as the Montgomery residue */
#define SOAK_PATTERNS 22
static void soak_pattern(mpz_t m, int t, const mpz_t q, int rbits, int *mont) {
  const int W = 29, L = rbits / W;
  mpz_t one; mpz_init_set_ui(one, 1);
  mpz_set_ui(m, 0);
  *mont = t >= 6;
  switch (t) {
    case 0: mpz_sub_ui(m, q, 1); break;
    case 1: mpz_sub_ui(m, q, 2); break;
    case 2: mpz_set_ui(m, 1); break;
    case 3: mpz_set_ui(m, 2); break;
    case 4: mpz_sub_ui(m, q, 1); mpz_fdiv_q_2exp(m, m, 1); break;
    case 5: mpz_add_ui(m, q, 1); mpz_fdiv_q_2exp(m, m, 1); break;
    case 6: mpz_sub_ui(m, q, 1); break;                                  /* residue q - 1: x = -1 / R */
    case 7: mpz_set_ui(m, 1); break;                                     /* residue 1: x = 1 / R */
    case 8: mpz_mul_2exp(m, one, W * L); mpz_sub_ui(m, m, 1); break;     /* every limb all ones (cut below q further down) */
    case 9: case 10: case 11: {                                           /* one limb all ones, the rest zero */
      int i = t == 9 ? 0 : t == 10 ? L / 2 : L - 2;
      mpz_mul_2exp(m, one, W); mpz_sub_ui(m, m, 1); mpz_mul_2exp(m, m, W * i); break;
    }
    case 12: case 13:                                                     /* alternating all-ones / all-zero limbs */
      for (int i = t - 12; i < L; i += 2) { mpz_t u; mpz_init(u); mpz_mul_2exp(u, one, W); mpz_sub_ui(u, u, 1); mpz_mul_2exp(u, u, W * i); mpz_add(m, m, u); mpz_clear(u); }
      break;
    case 14: mpz_mul_2exp(m, one, W); break;                              /* a single bit on a limb boundary */
    case 15: mpz_mul_2exp(m, one, W * (L / 2)); break;
    case 16: mpz_mul_2exp(m, one, W * (L - 2)); break;
    case 17:                                                               /* every limb 2^28 (the top bit of each limb) */
      for (int i = 0; i < L; i++) { mpz_t u; mpz_init(u); mpz_mul_2exp(u, one, W * i + W - 1); mpz_add(m, m, u); mpz_clear(u); }
      break;
    case 18:                                                               /* every limb 1 */
      for (int i = 0; i < L; i++) { mpz_t u; mpz_init(u); mpz_mul_2exp(u, one, W * i); mpz_add(m, m, u); mpz_clear(u); }
      break;
    case 19: mpz_mul_2exp(m, one, 32 * ((W * L) / 64)); mpz_sub_ui(m, m, 1); break;     /* low half of the words all ones */
    case 20: mpz_mul_2exp(m, one, W * (L - 1)); mpz_sub_ui(m, m, 1); break;            /* all limbs but the top one all ones */
    default: mpz_mul_2exp(m, one, W * L); mpz_sub_ui(m, m, 1); mpz_fdiv_q_2exp(m, m, 1); mpz_mul_2exp(m, m, 1); break;   /* ...1110 */
  }
  if (mpz_cmp(m, q) >= 0) mpz_fdiv_r_2exp(m, m, mpz_sizeinbase(q, 2) - 1);
  mpz_clear(one);
}
/* an F_q element (the base field of the coordinates) from pattern t, nudged `bump` upwards in the pattern's domain */
static void soak_set_fq(element_ptr c, int t, int bump, int rbits) {
  mpz_t m, r; int mont;
  mpz_init(m); mpz_init(r);
  soak_pattern(m, t, c->field->order, rbits, &mont);
  mpz_add_ui(m, m, bump);
  mpz_mod(m, m, c->field->order);
  if (mont) {                                                            /* x = m / R mod q */
    mpz_set_ui(r, 1); mpz_mul_2exp(r, r, rbits); mpz_invert(r, r, c->field->order);
    mpz_mul(m, m, r); mpz_mod(m, m, c->field->order);
  }
  element_set_mpz(c, m);
  mpz_clear(m); mpz_clear(r);
}
/* a point of P's curve whose x follows pattern t (sparse: only the first coefficient of x, the others zero) */
static void soak_point(element_t P, int t, int sparse, int rbits) {
  element_ptr x = curve_x_coord(P);
  element_t xx, u;
  element_init_same_as(xx, x);
  element_init_same_as(u, x);
  const int d = element_item_count(xx);
  for (int bump = 0;; bump++) {
    if (!d) soak_set_fq(xx, t, bump, rbits);
    else {
      element_set0(xx);
      for (int i = 0; i < (sparse ? 1 : d); i++) soak_set_fq(element_item(xx, i), t, i ? 0 : bump, rbits);
    }
    element_square(u, xx);
    element_add(u, u, curve_a_coeff(P));
    element_mul(u, u, xx);
    element_add(u, u, curve_b_coeff(P));
    if (element_is0(u) || !element_is_sqr(u)) continue;                  /* (u == 0: a point of order two) */
    /* points of tiny order are left out: Miller's algorithm then passes through O and 2-torsion (a vertical tangent,
     * f = 0), where the reference's value hangs on what mpz_invert does with 0 -- x = 1 on y^2 = x^3 + x has order 4 */
    curve_from_x(P, xx);
    element_t T;
    int tiny = 0;
    element_init_same_as(T, P);
    element_set(T, P);
    for (int i = 2; i <= 12 && !tiny; i++) { element_add(T, T, P); tiny = element_is0(T); }
    element_clear(T);
    if (!tiny) break;
  }
  if (t & 1) element_neg(P, P);
  element_clear(xx); element_clear(u);
}
/* every coordinate v of a record -> v + t q with the largest t that fits its fb bytes (which = bit mask of coordinates) */
static int soak_noncanonical(unsigned char *rec, int len, int fb, const mpz_t q, unsigned which) {
  int changed = 0;
  mpz_t v, lim;
  mpz_init(v); mpz_init(lim);
  mpz_set_ui(lim, 1); mpz_mul_2exp(lim, lim, 8 * fb);
  for (int c = 0; c < len / fb; c++) {
    if (!((which >> (c % 8)) & 1)) continue;
    mpz_import(v, fb, 1, 1, 1, 0, rec + c * fb);
    if (!mpz_sgn(v)) continue;                                           /* (zero stays zero: the reference keys on the bytes being zero) */
    mpz_add(v, v, q);
    if (mpz_cmp(v, lim) >= 0) continue;
    for (;;) { mpz_add(v, v, q); if (mpz_cmp(v, lim) >= 0) { mpz_sub(v, v, q); break; } }
    memset(rec + c * fb, 0, fb);
    size_t cnt = (mpz_sizeinbase(v, 2) + 7) / 8;
    mpz_export(rec + c * fb + (fb - cnt), NULL, 1, 1, 1, 0, v);
    changed++;
  }
  mpz_clear(v); mpz_clear(lim);
  return changed;
}
static int cmd_soak(int argc, char **argv) {
  if (argc < 8) { fprintf(stderr, "soak <param> <n_random> <k> <seed> <out> <workers> <rbits>\n"); return 2; }
  const char *param = argv[1], *outp = argv[5];
  const int nr = atoi(argv[2]), k = atoi(argv[3]), rbits = atoi(argv[7]);
  const unsigned seed = (unsigned) strtoul(argv[4], NULL, 10);
  int workers = atoi(argv[6]);
  if (workers < 1) workers = 1;
  if (workers > 256) workers = 256;
  const double t0 = now();
  pairing_t pairing; char type;
  pbc_random_set_deterministic(seed);
  init_pairing(pairing, param, &type);
  const int l1 = pairing_length_in_bytes_G1(pairing), l2 = pairing_length_in_bytes_G2(pairing), lt = pairing_length_in_bytes_GT(pairing);
  /* crafted block: per pattern and sparseness one G1 and one G2 point; units (cP, rQ), (rP, cQ), (cP, cQ); then non-canonical records */
  const int npat = k == 1 ? SOAK_PATTERNS : 0, nnc = k == 1 ? 24 : 0;
  const int ncraft = 3 * 2 * npat + nnc;
  const size_t n = (size_t) nr + ncraft, tot = (size_t) nr * k + ncraft;
  unsigned char *b1 = shared_alloc(tot * l1), *b2 = shared_alloc(tot * l2), *bo = shared_alloc(n * lt);
  for (int w = 0; w < workers; w++) {
    pid_t pid = fork();
    if (pid < 0) { perror("fork"); return 2; }
    if (pid == 0) {
      pairing_t pw; char tw;
      pbc_random_set_deterministic(seed + 7919u * (unsigned) (w + 1));
      init_pairing(pw, param, &tw);
      element_t *P = malloc(sizeof(element_t) * k), *Q = malloc(sizeof(element_t) * k), out;
      for (int j = 0; j < k; j++) { element_init_G1(P[j], pw); element_init_G2(Q[j], pw); }
      element_init_GT(out, pw);
      const size_t u0 = (size_t) nr * w / workers, u1 = (size_t) nr * (w + 1) / workers;
      for (size_t u = u0; u < u1; u++) {
        for (int j = 0; j < k; j++) {
          element_random(P[j]); element_random(Q[j]);
          element_to_bytes(b1 + (u * k + j) * l1, P[j]);
          element_to_bytes(b2 + (u * k + j) * l2, Q[j]);
        }
        if (k == 1) element_pairing(out, P[0], Q[0]); else element_prod_pairing(out, P, Q, k);
        element_to_bytes(bo + u * lt, out);
      }
      _exit(0);
    }
  }
  int changed = 0;
  if (ncraft) {
    element_t P, Q, out;
    element_init_G1(P, pairing); element_init_G2(Q, pairing); element_init_GT(out, pairing);
    const int fb = l1 / 2;                                                /* bytes of one F_q coordinate */
    size_t u = nr;
    for (int t = 0; t < npat; t++)
      for (int sparse = 0; sparse < 2; sparse++)
        for (int kind = 0; kind < 3; kind++, u++) {                       /* 0: (cP, rQ)  1: (rP, cQ)  2: (cP, cQ) */
          if (kind == 1) element_random(P); else soak_point(P, t, sparse, rbits);
          /* (type g: a twist point with both coordinates in F_q gives a Miller value in F_q^2, which the easy part of the final
           * exponentiation sends to +-1 -- and the reference's own cc_tatepower of g_param.c then inverts zero ("division by
           * zero", poly.c:416): no reference value exists, so its sparse patterns fill every coefficient too) */
          if (kind == 0) element_random(Q); else soak_point(Q, (t + kind) % npat, type == 'g' ? 0 : sparse, rbits);
          element_to_bytes(b1 + u * l1, P);
          element_to_bytes(b2 + u * l2, Q);
          element_pairing(out, P, Q);
          element_to_bytes(bo + u * lt, out);
        }
    for (int c = 0; c < nnc; c++, u++) {
      element_random(P); element_random(Q);
      element_to_bytes(b1 + u * l1, P);
      element_to_bytes(b2 + u * l2, Q);
      changed += soak_noncanonical(b1 + u * l1, l1, fb, curve_x_coord(P)->field->order, c % 3 == 1 ? 0 : 1u + (unsigned) c % 3);
      changed += soak_noncanonical(b2 + u * l2, l2, fb, curve_x_coord(P)->field->order, c % 3 == 0 ? 0 : 0xffu >> (c % 5));
      element_from_bytes(P, b1 + u * l1);
      element_from_bytes(Q, b2 + u * l2);
      element_pairing(out, P, Q);
      element_to_bytes(bo + u * lt, out);
    }
  }
  int bad = 0, st;
  while (wait(&st) > 0) bad += !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
  if (bad) { fprintf(stderr, "soak: %d workers failed\n", bad); return 3; }
  FILE *fp = fopen(outp, "wb");
  if (!fp) { perror(outp); return 2; }
  fwrite("PBCVEC01", 1, 8, fp);
  w32(fp, (uint32_t) type); w32(fp, (uint32_t) n); w32(fp, k); w32(fp, l1); w32(fp, l2); w32(fp, lt);
  /* k > 1 has no crafted block, so the term arrays are exactly n k records */
  fwrite(b1, l1, tot, fp); fwrite(b2, l2, tot, fp); fwrite(bo, lt, n, fp);
  fclose(fp);
  printf("{\"param\": \"%s\", \"type\": \"%c\", \"random_units\": %d, \"terms_per_unit\": %d, \"crafted_units\": %d, \"crafted_patterns\": %d, "
         "\"noncanonical_units\": %d, \"noncanonical_coordinates\": %d, \"seed\": %u, \"workers\": %d, \"rbits\": %d, \"reference_wall_s\": %.2f}\n",
         param, type, nr, k, ncraft, npat, nnc, changed, seed, workers, rbits, now() - t0);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: ref_tool gen|kat|bench ...\n"); return 2; }
  if (!strcmp(argv[1], "gen")) return cmd_gen(argc - 1, argv + 1);
  if (!strcmp(argv[1], "kat")) return cmd_kat(argc - 1, argv + 1);
  if (!strcmp(argv[1], "bench")) return cmd_bench(argc - 1, argv + 1);
  if (!strcmp(argv[1], "benchg")) return cmd_benchg(argc - 1, argv + 1);
  if (!strcmp(argv[1], "ppow")) return cmd_ppow(argc - 1, argv + 1);
  if (!strcmp(argv[1], "bls")) return cmd_bls(argc - 1, argv + 1);
  if (!strcmp(argv[1], "hash")) return cmd_hash(argc - 1, argv + 1);
  if (!strcmp(argv[1], "gmul")) return cmd_gmul(argc - 1, argv + 1);
  if (!strcmp(argv[1], "compress")) return cmd_compress(argc - 1, argv + 1);
  if (!strcmp(argv[1], "xonly")) return cmd_xonly(argc - 1, argv + 1);
  if (!strcmp(argv[1], "gena")) return cmd_gena(argc - 1, argv + 1);
  if (!strcmp(argv[1], "gena1")) return cmd_gena1(argc - 1, argv + 1);
  if (!strcmp(argv[1], "gene")) return cmd_gene(argc - 1, argv + 1);
  if (!strcmp(argv[1], "genf")) return cmd_genf(argc - 1, argv + 1);
  if (!strcmp(argv[1], "rdep")) return cmd_rdep(argc - 1, argv + 1);
  if (!strcmp(argv[1], "finalpow")) return cmd_finalpow(argc - 1, argv + 1);
  if (!strcmp(argv[1], "text")) return cmd_text(argc - 1, argv + 1);
  if (!strcmp(argv[1], "gops")) return cmd_gops(argc - 1, argv + 1);
  if (!strcmp(argv[1], "zrops")) return cmd_zrops(argc - 1, argv + 1);
  if (!strcmp(argv[1], "pow23")) return cmd_pow23(argc - 1, argv + 1);
  if (!strcmp(argv[1], "soak")) return cmd_soak(argc - 1, argv + 1);
  return 2;
}
