/* glue_test.c -- TEST DRIVER: unmodified reference PBC + pbc_hip_glue.c.  Every check compares
 * the GPU result with the reference's own CPU result through element_cmp.
 *   usage: glue_test <param-file> [n]        (PBC_HIP_LIB = path of libpbc_hip.so) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "pbc_hip_glue.h"

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  size_t n = argc > 2 ? (size_t) atoi(argv[2]) : 200;
  char text[8192];
  FILE *fp = fopen(argv[1], "rb");
  if (!fp) { perror(argv[1]); return 2; }
  size_t len = fread(text, 1, sizeof text - 1, fp);
  text[len] = 0;
  fclose(fp);
  pbc_random_set_deterministic(4242);
  pairing_t pairing;
  if (pairing_init_set_buf(pairing, text, len)) return 2;
  if (argc > 3 && !strcmp(argv[3], "bench")) {
    /* throughput of element_pairing_batch THROUGH the PBC element API: n element_t pairs (64 distinct
     * points, shallow copies) -> n GT elements; dominated by element_to_bytes / element_from_bytes */
    enum { D = 64 };
    element_t p0[D], q0[D];
    for (int i = 0; i < D; i++) {
      element_init_G1(p0[i], pairing); element_init_G2(q0[i], pairing);
      element_random(p0[i]); element_random(q0[i]);
    }
    element_t *Pb = malloc(sizeof(element_t) * n), *Qb = malloc(sizeof(element_t) * n), *Ob = malloc(sizeof(element_t) * n);
    for (size_t i = 0; i < n; i++) { Pb[i][0] = p0[i % D][0]; Qb[i][0] = q0[(i / D) % D][0]; element_init_GT(Ob[i], pairing); }
    if (pbc_hip_attach(pairing, text, len)) { printf("ATTACH FAILED\n"); return 1; }
    if (element_pairing_batch(Ob, Pb, Qb, n < 4096 ? n : 4096)) { printf("batch call failed\n"); return 1; }   /* warm-up */
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (element_pairing_batch(Ob, Pb, Qb, n)) { printf("batch call failed\n"); return 1; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    element_t chk;
    element_init_GT(chk, pairing);
    pbc_hip_detach(pairing);
    int bad = 0;
    for (int t = 0; t <= 16; t++) {                                 /* 17 units spread over the chunks of the batch, on the CPU */
      const size_t u = t == 16 ? n - 1 : n / 16 * (size_t) t;
      element_pairing(chk, Pb[u], Qb[u]);
      if (element_cmp(chk, Ob[u])) { if (!bad) fprintf(stderr, "first mismatch at unit %zu\n", u); bad++; }
    }
    printf("%s: element_pairing_batch of %zu element_t pairs: %.3f s = %.0f pairs/s (%s)\n", argv[1], n, dt, n / dt,
           bad ? "MISMATCH" : "last result equals the CPU pairing");
    return bad ? 1 : 0;
  }
  if (argc > 3 && !strcmp(argv[3], "latency")) {
    /* what a program that calls element_pairing ONE pair at a time pays through the hooks (example/bls.c's shape):
     * n single calls on the GPU (pairing->map -> a one-unit batch), then the same calls on the CPU */
    element_t p, q, o, c;
    element_init_G1(p, pairing); element_init_G2(q, pairing); element_init_GT(o, pairing); element_init_GT(c, pairing);
    element_random(p); element_random(q);
    if (pbc_hip_attach(pairing, text, len)) { printf("ATTACH FAILED\n"); return 1; }
    element_pairing(o, p, q);                                          /* warm-up: context, constants, buffers */
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t i = 0; i < n; i++) element_pairing(o, p, q);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double gpu = ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / n;
    pbc_hip_detach(pairing);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t i = 0; i < n; i++) element_pairing(c, p, q);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double cpu = ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / n;
    printf("%s: single element_pairing calls through the hooks: %.3f ms each on the GPU, %.3f ms each on the CPU (%s)\n", argv[1],
           gpu * 1e3, cpu * 1e3, element_cmp(c, o) ? "MISMATCH" : "same value");
    int bad = element_cmp(c, o) ? 1 : 0;
    /* the same for ONE element_prod_pairing of four terms (a Groth16-style check) and ONE pairing_pp_apply per call */
    enum { KT = 4 };
    element_t ps[KT], qs[KT];
    for (int i = 0; i < KT; i++) { element_init_G1(ps[i], pairing); element_init_G2(qs[i], pairing); element_random(ps[i]); element_random(qs[i]); }
    const size_t m = n < 50 ? n : 50;
    double tg[2], tc[2];
    for (int pass = 0; pass < 2; pass++) {                              /* pass 0: the hooks; pass 1: stock PBC */
      if (pass == 0 && pbc_hip_attach(pairing, text, len)) { printf("ATTACH FAILED\n"); return 1; }
      element_ptr r = pass == 0 ? o : c;
      double *t = pass == 0 ? tg : tc;
      element_prod_pairing(r, ps, qs, KT);
      clock_gettime(CLOCK_MONOTONIC, &t0);
      for (size_t i = 0; i < m; i++) element_prod_pairing(r, ps, qs, KT);
      clock_gettime(CLOCK_MONOTONIC, &t1);
      t[0] = ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / m;
      if (pass == 1 && element_cmp(c, o)) { bad = 1; printf("element_prod_pairing MISMATCH\n"); }
      if (pass == 0) pbc_hip_detach(pairing);
    }
    element_t o2, c2;
    element_init_GT(o2, pairing); element_init_GT(c2, pairing);
    for (int pass = 0; pass < 2; pass++) {
      if (pass == 0 && pbc_hip_attach(pairing, text, len)) { printf("ATTACH FAILED\n"); return 1; }
      element_ptr r = pass == 0 ? o2 : c2;
      double *t = pass == 0 ? tg : tc;
      pairing_pp_t pp;
      pairing_pp_init(pp, p, pairing);
      pairing_pp_apply(r, q, pp);
      clock_gettime(CLOCK_MONOTONIC, &t0);
      for (size_t i = 0; i < m; i++) pairing_pp_apply(r, q, pp);
      clock_gettime(CLOCK_MONOTONIC, &t1);
      t[1] = ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / m;
      pairing_pp_clear(pp);
      if (pass == 0) pbc_hip_detach(pairing);
    }
    if (element_cmp(c2, o2)) { bad = 1; printf("pairing_pp_apply MISMATCH\n"); }
    printf("%s: one element_prod_pairing of %d terms per call: %.3f ms on the GPU, %.3f ms on the CPU; one pairing_pp_apply per call: %.3f ms on the GPU, %.3f ms on the CPU (%s)\n",
           argv[1], KT, tg[0] * 1e3, tc[0] * 1e3, tg[1] * 1e3, tc[1] * 1e3, bad ? "MISMATCH" : "same values");
    return bad;
  }
  if (argc > 3 && !strcmp(argv[3], "hash")) {   /* element_from_hash_batch on G1 and G2 vs the CPU */
    int fails = 0;
    enum { HN = 8, HL = 32 };
    if (pbc_hip_attach(pairing, text, len)) { printf("ATTACH FAILED\n"); return 1; }
    unsigned char dig[HN * HL];
    for (int i = 0; i < HN * HL; i++) dig[i] = (unsigned char) (i * 37 + 11);
    for (int group = 1; group <= 2; group++) {
      element_t h[HN], c;
      for (int i = 0; i < HN; i++) { if (group == 1) element_init_G1(h[i], pairing); else element_init_G2(h[i], pairing); }
      if (group == 1) element_init_G1(c, pairing); else element_init_G2(c, pairing);
      if (element_from_hash_batch(h, dig, HL, HN)) { printf("from_hash batch failed (G%d)\n", group); fails++; }
      for (int i = 0; i < HN; i++) {
        element_from_hash(c, dig + i * HL, HL);
        if (element_cmp(c, h[i])) { printf("G%d from_hash mismatch at %d\n", group, i); fails++; }
        element_clear(h[i]);
      }
      element_clear(c);
    }
    pbc_hip_detach(pairing);
    printf("%s: element_from_hash_batch on G1 and G2: %s\n", argv[1], fails ? "FAIL" : "PASS");
    return fails ? 1 : 0;
  }
  if (argc > 3 && !strcmp(argv[3], "lifetime")) {
    /* orders of pairing_pp_clear / pairing_clear / detach that stock PBC tolerates, and the empty product in a batch */
    int fails = 0;
    element_t P, Q, want, got;
    element_init_G1(P, pairing); element_init_G2(Q, pairing); element_init_GT(want, pairing); element_init_GT(got, pairing);
    element_random(P); element_random(Q);
    element_pairing(want, P, Q);                                             /* CPU */
    if (pbc_hip_attach(pairing, text, len)) { printf("ATTACH FAILED\n"); return 1; }
    pairing_pp_t pp1, pp2, pp3;
    pairing_pp_init(pp1, P, pairing);                                        /* GPU */
    pairing_pp_apply(got, Q, pp1);
    if (element_cmp(got, want)) { printf("pp_apply mismatch (attached)\n"); fails++; }
    pbc_hip_detach(pairing);                                                 /* pp1 still alive */
    pairing_pp_init(pp2, P, pairing);                                        /* CPU again */
    pairing_pp_apply(got, Q, pp2);
    if (element_cmp(got, want)) { printf("pp_apply mismatch (detached)\n"); fails++; }
    pairing_pp_clear(pp1);                                                   /* a GPU-made object cleared after the detach */
    pairing_pp_clear(pp2);
    if (pbc_hip_attach(pairing, text, len)) { printf("RE-ATTACH FAILED\n"); return 1; }
    pairing_pp_init(pp3, P, pairing);
    pairing_pp_apply(got, Q, pp3);
    if (element_cmp(got, want)) { printf("pp_apply mismatch (re-attached)\n"); fails++; }
    {
      element_t o[3];
      for (int i = 0; i < 3; i++) { element_init_GT(o[i], pairing); element_set(o[i], want); }
      if (element_prod_pairing_batch(o, &P, &Q, 3, 0)) { printf("empty products failed\n"); fails++; }
      for (int i = 0; i < 3; i++) { if (!element_is1(o[i])) { printf("empty product is not 1\n"); fails++; } element_clear(o[i]); }
    }
    element_clear(P); element_clear(Q); element_clear(want); element_clear(got);
    pairing_clear(pairing);                                                  /* detaches; pp3 still alive */
    pairing_pp_clear(pp3);                                                   /* stock PBC allows this order */
    printf("%s: pairing_pp_t lifetimes across detach / pairing_clear: %s\n", argv[1], fails ? "FAIL" : "PASS");
    return fails ? 1 : 0;
  }
  int fails = 0, K = 4;
  if (n < 24) n = 24;                  /* the fixed-index sections below use up to 24 elements */
  element_t *P = malloc(sizeof(element_t) * n), *Q = malloc(sizeof(element_t) * n);
  element_t *cpu = malloc(sizeof(element_t) * n), *gpu = malloc(sizeof(element_t) * n);
  for (size_t i = 0; i < n; i++) {
    element_init_G1(P[i], pairing); element_init_G2(Q[i], pairing);
    element_init_GT(cpu[i], pairing); element_init_GT(gpu[i], pairing);
    element_random(P[i]); element_random(Q[i]);
  }
  element_set0(P[3]);                                   /* identity inputs */
  if (n > 7) element_set0(Q[7]);
  for (size_t i = 0; i < n; i++) element_pairing(cpu[i], P[i], Q[i]);          /* CPU reference */
  element_t cprod, gprod;
  element_init_GT(cprod, pairing); element_init_GT(gprod, pairing);
  element_prod_pairing(cprod, P + 8, Q + 8, K);
  element_t cpp0[12], cpp1[12];                          /* CPU references for the pp checks */
  for (size_t i = 0; i < 12 && i < n; i++) {
    element_init_GT(cpp0[i], pairing); element_init_GT(cpp1[i], pairing);
    element_pairing(cpp0[i], P[0], Q[i]);
    element_pairing(cpp1[i], P[1], Q[i]);
  }

  if (pbc_hip_attach(pairing, text, len)) { printf("ATTACH FAILED\n"); return 1; }
  /* 1. unchanged call sites: element_pairing / element_prod_pairing now run on the GPU */
  for (size_t i = 0; i < 12 && i < n; i++) {
    element_pairing(gpu[i], P[i], Q[i]);
    if (element_cmp(gpu[i], cpu[i])) { printf("element_pairing mismatch at %zu\n", i); fails++; }
  }
  element_prod_pairing(gprod, P + 8, Q + 8, K);
  if (element_cmp(gprod, cprod)) { printf("element_prod_pairing mismatch\n"); fails++; }
  /* 1b. preprocessed pairings through pairing->pp_init/pp_apply (on the GPU for types a, d, g;
   *     the reference's own CPU routines otherwise) */
  {
    pairing_pp_t pp;
    pairing_pp_init(pp, P[0], pairing);
    for (size_t i = 0; i < 6 && i < n; i++) {
      pairing_pp_apply(gpu[i], Q[i], pp);
      if (element_cmp(gpu[i], cpp0[i])) { printf("pairing_pp_apply mismatch at %zu\n", i); fails++; }
    }
    pairing_pp_clear(pp);
    pairing_pp_init(pp, P[1], pairing);
    if (pairing_pp_apply_batch(gpu, Q, 12 < n ? 12 : n, pp)) { printf("pp batch failed\n"); fails++; }
    pairing_pp_clear(pp);
    for (size_t i = 0; i < 12 && i < n; i++)
      if (element_cmp(gpu[i], cpp1[i])) { printf("pp batch mismatch at %zu\n", i); fails++; }
  }
  /* 1c. group operations: G1 scalar multiplication and GT powers vs the CPU */
  {
    size_t m = n < 24 ? n : 24;
    element_t *zr = malloc(sizeof(element_t) * m), *o1 = malloc(sizeof(element_t) * m), *ot = malloc(sizeof(element_t) * m);
    for (size_t i = 0; i < m; i++) {
      element_init_Zr(zr[i], pairing); element_random(zr[i]);
      element_init_G1(o1[i], pairing); element_init_GT(ot[i], pairing);
    }
    if (element_pow_zn_batch(o1, P, zr, m) || element_pow_zn_batch(ot, cpu, zr, m)) { printf("pow_zn batch failed\n"); fails++; }
    for (size_t i = 0; i < m; i++) {
      element_t c1, ct;
      element_init_G1(c1, pairing); element_init_GT(ct, pairing);
      element_pow_zn(c1, P[i], zr[i]);
      element_pow_zn(ct, cpu[i], zr[i]);
      if (element_cmp(c1, o1[i])) { printf("G1 pow_zn mismatch at %zu\n", i); fails++; }
      if (element_cmp(ct, ot[i])) { printf("GT pow_zn mismatch at %zu\n", i); fails++; }
      element_clear(c1); element_clear(ct);
    }
  }
  /* 1d. pairing->finalpow behind element_from_hash on GT (gt_from_hash, ecc/pairing.c:117-122): GPU now, CPU after detach */
  /* (not for type g: the reference's own tatepower10 divides by zero on these subfield elements -- pbc_die) */
  const int hash_gt = strstr(text, "type g") == NULL;
  element_t gth[4];
  for (int i = 0; i < 4 && hash_gt; i++) { element_init_GT(gth[i], pairing); element_from_hash(gth[i], "finalpow/glue" + i, 9); }
  /* 2. the batch entry points */
  if (element_pairing_batch(gpu, P, Q, n)) { printf("batch call failed\n"); fails++; }
  for (size_t i = 0; i < n; i++) if (element_cmp(gpu[i], cpu[i])) { printf("batch mismatch at %zu\n", i); fails++; break; }
  size_t np = n / K;
  if (element_prod_pairing_batch(gpu, P, Q, np, K)) { printf("prod batch call failed\n"); fails++; }
  pbc_hip_detach(pairing);
  for (int i = 0; i < 4 && hash_gt; i++) {
    element_t c;
    element_init_GT(c, pairing);
    element_from_hash(c, "finalpow/glue" + i, 9);                              /* CPU */
    if (element_cmp(c, gth[i])) { printf("finalpow (GT from_hash) mismatch at %d\n", i); fails++; }
    element_clear(c);
  }
  for (size_t u = 0; u < np && u < 16; u++) {
    element_prod_pairing(cprod, P + u * K, Q + u * K, K);                     /* CPU again after detach */
    if (element_cmp(gpu[u], cprod)) { printf("prod batch mismatch at %zu\n", u); fails++; }
  }
  printf("%s: %zu pairings, %zu products of %d: %s\n", argv[1], n, np, K, fails ? "FAIL" : "PASS");
  return fails ? 1 : 0;
}
