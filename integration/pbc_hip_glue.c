/* pbc_hip_glue.c -- see pbc_hip_glue.h.  Plain C, PBC's public API only. */
#include "pbc_hip_glue.h"

#include <dlfcn.h>
#include <pthread.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct pbc_hip_pairing_s pbc_hip_pairing_t;
static struct {
  void *dl;
  int (*init)(pbc_hip_pairing_t **, const char *, size_t);
  void (*clear)(pbc_hip_pairing_t *);
  int (*len1)(const pbc_hip_pairing_t *), (*len2)(const pbc_hip_pairing_t *), (*lenT)(const pbc_hip_pairing_t *);
  int (*type)(const pbc_hip_pairing_t *);
  int (*use_devices)(pbc_hip_pairing_t *, const int *, int);
  int (*device_count)(void);
  int (*pair)(pbc_hip_pairing_t *, unsigned char *, const unsigned char *, const unsigned char *, size_t);
  int (*prod)(pbc_hip_pairing_t *, unsigned char *, const unsigned char *, const unsigned char *, size_t, int);
  const char *(*err)(void);
  int (*pp_init)(void **, pbc_hip_pairing_t *, const unsigned char *);
  void (*pp_clear)(void *);
  int (*pp_apply)(void *, unsigned char *, const unsigned char *, size_t);
  int (*lenZr)(const pbc_hip_pairing_t *);
  int (*mul_zn)(pbc_hip_pairing_t *, int, unsigned char *, const unsigned char *, const unsigned char *, size_t);
  int (*gt_mul)(pbc_hip_pairing_t *, unsigned char *, const unsigned char *, const unsigned char *, size_t);
  int (*gt_pow)(pbc_hip_pairing_t *, unsigned char *, const unsigned char *, const unsigned char *, size_t);
  int (*from_hash)(pbc_hip_pairing_t *, int, unsigned char *, const unsigned char *, int, size_t);
  int (*host_alloc)(void **, size_t);
  void (*host_free)(void *);
  int (*finalpow)(pbc_hip_pairing_t *, unsigned char *, const unsigned char *, size_t);
} L;

/* one attachment per pairing_s (kept in a table keyed by the pairing pointer so that struct pairing_s itself needs no
 * new field).  The table grows on demand and is guarded by a lock; an entry is removed by pbc_hip_detach or by
 * pairing_clear itself (the pairing's clear_func is hooked), so a pairing_t reused at the same address attaches anew. */
typedef struct {
  struct pairing_s *pairing;
  pbc_hip_pairing_t *gpu;
  void (*cpu_map)(element_ptr, element_ptr, element_ptr, struct pairing_s *);
  void (*cpu_prod)(element_ptr, element_t[], element_t[], int, struct pairing_s *);
  void (*cpu_pp_init)(pairing_pp_t, element_t, struct pairing_s *);
  void (*cpu_pp_clear)(pairing_pp_t);
  void (*cpu_pp_apply)(element_t, element_t, pairing_pp_t);
  void (*cpu_clear)(struct pairing_s *);
  void (*cpu_finalpow)(element_t);
  /* page-locked staging buffers of the batch calls (pbc_hip_host_alloc): kept and grown, never per call */
  unsigned char *pin[3];
  size_t pin_cap[3];
} attach_t;
static attach_t **g_att;
static int g_natt, g_catt;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
/* calls that ran on the GPU (reported at exit when PBC_HIP_VERBOSE=1; the preload test reads them) */
static struct { unsigned long map, prod, pp_init, pp_apply, batch_units, finalpow; } g_stat;

/* pairing_pp_t objects made by the hooks below, and the pairings that were detached while some were alive.
 * pairing_pp_clear / pairing_pp_apply dispatch through pairing->pp_clear / pp_apply (include/pbc_pairing.h:79-98), and
 * stock PBC lets a program clear a pairing_pp_t AFTER pairing_clear: the CPU routines restored by a detach must never
 * see a GPU handle.  So detach releases the GPU side of every live object (marking it dead), and -- only while such
 * objects exist -- leaves the three pp hooks installed with the pairing's CPU routines parked in a "retired" record:
 * a dead object is freed by the hook, applying it dies with a message, objects made on the CPU pass through. */
typedef struct { void *pp, *data; struct pairing_s *pairing; int kind /* 1 GPU handle, 2 element copy */, dead; } ppreg_t;
static ppreg_t *g_pp;
static int g_npp, g_cpp;
typedef struct {
  struct pairing_s *pairing;
  void (*cpu_pp_init)(pairing_pp_t, element_t, struct pairing_s *);
  void (*cpu_pp_clear)(pairing_pp_t);
  void (*cpu_pp_apply)(element_t, element_t, pairing_pp_t);
} retired_t;
static retired_t *g_ret;
static int g_nret, g_cret;

static attach_t *find(struct pairing_s *p) {
  attach_t *r = NULL;
  pthread_mutex_lock(&g_lock);
  for (int i = 0; i < g_natt; i++) if (g_att[i]->pairing == p) { r = g_att[i]; break; }
  pthread_mutex_unlock(&g_lock);
  return r;
}
static attach_t *add_entry(void) {
  attach_t *a = calloc(1, sizeof *a);
  if (!a) return NULL;
  pthread_mutex_lock(&g_lock);
  if (g_natt == g_catt) {
    int nc = g_catt ? 2 * g_catt : 8;
    attach_t **t = realloc(g_att, (size_t) nc * sizeof *t);
    if (!t) { pthread_mutex_unlock(&g_lock); free(a); return NULL; }
    g_att = t; g_catt = nc;
  }
  g_att[g_natt++] = a;
  pthread_mutex_unlock(&g_lock);
  return a;
}
static void drop_entry(attach_t *a) {
  pthread_mutex_lock(&g_lock);
  for (int i = 0; i < g_natt; i++) if (g_att[i] == a) { g_att[i] = g_att[--g_natt]; break; }
  pthread_mutex_unlock(&g_lock);
  for (int i = 0; i < 3; i++) if (a->pin[i]) L.host_free(a->pin[i]);
  free(a);
}
/* at least `bytes` of page-locked memory in staging slot i of the attachment */
static unsigned char *pinned(attach_t *a, int i, size_t bytes) {
  if (a->pin_cap[i] < bytes) {
    if (a->pin[i]) L.host_free(a->pin[i]);
    a->pin[i] = NULL; a->pin_cap[i] = 0;
    void *p = NULL;
    if (L.host_alloc(&p, bytes + (bytes >> 2) + 64)) return NULL;
    a->pin[i] = p; a->pin_cap[i] = bytes + (bytes >> 2) + 64;
  }
  return a->pin[i];
}
static int load_lib(void) {
  if (L.dl) return 0;
  const char *path = getenv("PBC_HIP_LIB");
  L.dl = dlopen(path ? path : "libpbc_hip.so", RTLD_NOW | RTLD_GLOBAL);
  if (!L.dl) { fprintf(stderr, "pbc_hip: %s\n", dlerror()); return 1; }
#define SYM(f, n) do { *(void **) &L.f = dlsym(L.dl, n); if (!L.f) { fprintf(stderr, "pbc_hip: missing %s\n", n); return 1; } } while (0)
  SYM(init, "pbc_hip_pairing_init_set_buf"); SYM(clear, "pbc_hip_pairing_clear");
  SYM(len1, "pbc_hip_pairing_length_in_bytes_G1"); SYM(len2, "pbc_hip_pairing_length_in_bytes_G2");
  SYM(lenT, "pbc_hip_pairing_length_in_bytes_GT"); SYM(type, "pbc_hip_pairing_type");
  SYM(use_devices, "pbc_hip_pairing_use_devices"); SYM(device_count, "pbc_hip_device_count");
  SYM(pair, "pbc_hip_element_pairing_batch"); SYM(prod, "pbc_hip_element_prod_pairing_batch");
  SYM(err, "pbc_hip_last_error");
  SYM(pp_init, "pbc_hip_pairing_pp_init"); SYM(pp_clear, "pbc_hip_pairing_pp_clear");
  SYM(pp_apply, "pbc_hip_pairing_pp_apply_batch");
  SYM(lenZr, "pbc_hip_pairing_length_in_bytes_Zr"); SYM(mul_zn, "pbc_hip_element_mul_zn_batch");
  SYM(gt_mul, "pbc_hip_element_mul_GT_batch"); SYM(gt_pow, "pbc_hip_element_pow_zn_GT_batch");
  SYM(from_hash, "pbc_hip_element_from_hash_batch");
  SYM(host_alloc, "pbc_hip_host_alloc"); SYM(host_free, "pbc_hip_host_free");
  SYM(finalpow, "pbc_hip_finalpow_batch");
#undef SYM
  return 0;
}

/* Worker threads for the element <-> bytes conversions of the batch calls (each costs a Montgomery
 * reduction and a GMP export per coordinate; for 2^20 pairs that is seconds on one core, far more than
 * the GPU needs).  PBC_HIP_GLUE_THREADS overrides the default of min(online CPUs, 16). */
typedef struct { void (*fn)(size_t lo, size_t hi, void *ctx); void *ctx; size_t lo, hi; } job_t;
static void *job_main(void *arg) { job_t *j = arg; j->fn(j->lo, j->hi, j->ctx); return NULL; }
static void parallel_range(size_t lo0, size_t hi0, void (*fn)(size_t, size_t, void *), void *ctx) {
  const size_t n = hi0 - lo0;
  long nt = sysconf(_SC_NPROCESSORS_ONLN);
  const char *e = getenv("PBC_HIP_GLUE_THREADS");
  if (e) nt = atol(e);
  if (nt > 16) nt = 16;
  if (nt < 1 || n < 256) nt = 1;
  if (nt == 1) { fn(lo0, hi0, ctx); return; }
  pthread_t th[16];
  job_t job[16];
  int started = 0;
  for (long t = 0; t < nt; t++) {
    job[t].fn = fn; job[t].ctx = ctx;
    job[t].lo = lo0 + n * (size_t) t / (size_t) nt; job[t].hi = lo0 + n * (size_t) (t + 1) / (size_t) nt;
    if (t + 1 == nt || pthread_create(&th[t], NULL, job_main, &job[t])) { fn(job[t].lo, t + 1 == nt ? job[t].hi : hi0, ctx); break; }
    started++;
  }
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}
typedef struct {
  element_t *in1, *in2, *out;
  unsigned char *b1, *b2, *bt;
  size_t *slot;
  int k, l1, l2, lt;
} conv_t;
static void to_bytes_range(size_t lo, size_t hi, void *ctx) {
  conv_t *c = ctx;
  for (size_t i = lo; i < hi; i++) {
    const size_t u = c->slot[i];
    for (int j = 0; j < c->k; j++) {
      element_to_bytes(c->b1 + (i * c->k + j) * c->l1, c->in1[u * c->k + j]);
      element_to_bytes(c->b2 + (i * c->k + j) * c->l2, c->in2[u * c->k + j]);
    }
  }
}
static void from_bytes_range(size_t lo, size_t hi, void *ctx) {
  conv_t *c = ctx;
  for (size_t i = lo; i < hi; i++) element_from_bytes(c->out[c->slot[i]], c->bt + i * c->lt);
}

/* one GPU call of the batch pipeline, on its own thread while the CPU threads convert the neighbouring chunks */
typedef struct {
  void *att;
  int k;
  unsigned char *b1, *b2, *bt;
  size_t m;
  int rc;
  char err[256];
} gpu_job_t;
static void *gpu_job_main(void *arg);

/* n*k (in1, in2) terms -> n GT results.  `out` are GT elements (the mulg wrapper, ecc/pairing.c:135-283). */
static int run_batch(attach_t *a, element_t out[], element_t in1[], element_t in2[], size_t n, int k) {
  if (k < 1) { for (size_t u = 0; u < n; u++) element_set1(out[u]); return 0; }   /* empty products, as hip_prod */
  int l1 = L.len1(a->gpu), l2 = L.len2(a->gpu), lt = L.lenT(a->gpu);
  size_t terms = n * (size_t) k, m = 0;
  unsigned char *b1 = pinned(a, 0, terms * l1), *b2 = pinned(a, 1, terms * l2), *bt = pinned(a, 2, n * lt);
  size_t *slot = malloc(n * sizeof *slot);
  if (!b1 || !b2 || !bt || !slot) { free(slot); pbc_error("pbc_hip: out of memory for a batch of %lu units", (unsigned long) n); return 1; }
  /* host pre-filter: identity inputs never reach the device (pairing_apply :123-130,
   * element_prod_pairing :161-168); PBC's wire format cannot express O */
  for (size_t u = 0; u < n; u++) {
    int ident = 0;
    for (int j = 0; j < k; j++) if (element_is0(in1[u * k + j]) || element_is0(in2[u * k + j])) ident = 1;
    if (ident) { element_set0(out[u]); continue; }
    slot[m++] = u;
  }
  conv_t c = {in1, in2, out, b1, b2, bt, slot, k, l1, l2, lt};
  int rc = 0;
  /* Three stages per chunk -- element_to_bytes (CPU threads), the GPU call, element_from_bytes (CPU threads) -- run as a
   * pipeline: while the GPU works on chunk i the CPU threads convert the results of chunk i - 1 and the inputs of
   * chunk i + 1.  (The conversions cost more CPU time than the GPU needs for the pairings.) */
  const size_t CH = 131072 / (size_t) k > 32768 ? 131072 / (size_t) k : 32768;   /* at least one chip residency of lanes per GPU call (products: of terms, and 256 workgroups of products) */
  const size_t nc = (m + CH - 1) / CH;
  if (m) parallel_range(0, m < CH ? m : CH, to_bytes_range, &c);
  for (size_t ci = 0; ci < nc && !rc; ci++) {
    const size_t lo = ci * CH, hi = lo + CH < m ? lo + CH : m;
    gpu_job_t job = {a, k, b1 + lo * k * l1, b2 + lo * k * l2, bt + lo * lt, hi - lo, 0, {0}};
    pthread_t th;
    const int threaded = nc > 1 && !pthread_create(&th, NULL, gpu_job_main, &job);
    if (!threaded) gpu_job_main(&job);
    if (ci > 0) parallel_range(lo - CH, lo, from_bytes_range, &c);
    if (hi < m) parallel_range(hi, hi + CH < m ? hi + CH : m, to_bytes_range, &c);
    if (threaded) pthread_join(th, NULL);
    if (job.rc) { rc = 1; pbc_error("pbc_hip: %s", job.err); }
  }
  if (m && !rc) { parallel_range((nc - 1) * CH, m, from_bytes_range, &c); g_stat.batch_units += m; }
  free(slot);
  return rc;
}

static void *gpu_job_main(void *arg) {
  gpu_job_t *j = arg;
  attach_t *a = j->att;
  j->rc = j->k == 1 ? L.pair(a->gpu, j->bt, j->b1, j->b2, j->m) : L.prod(a->gpu, j->bt, j->b1, j->b2, j->m, j->k);
  if (j->rc) { strncpy(j->err, L.err(), sizeof j->err - 1); j->err[sizeof j->err - 1] = 0; }   /* the message is per thread */
  return NULL;
}

/* A GPU call behind one of PBC's void-returning hooks failed.  PBC's convention for an unrecoverable condition is
 * pbc_die (misc/utils.c:75-82: message, exit(128)); there is no other path: a pairing silently computed somewhere else
 * would hide the fault. */
static void gpu_failed(const char *what) { pbc_die("pbc_hip: %s: %s", what, L.err()); }
static void *xmalloc(size_t n) {
  void *p = malloc(n ? n : 1);
  if (!p) pbc_die("pbc_hip: out of memory");
  return p;
}

/* pairing->map replacement: `out` is the element INSIDE the GT wrapper (out->data of the GT
 * element, include/pbc_pairing.h:131-134), so it is filled through its own field. */
static void hip_map(element_ptr out, element_ptr in1, element_ptr in2, struct pairing_s *p) {
  attach_t *a = find(p);
  int l1 = L.len1(a->gpu), l2 = L.len2(a->gpu), lt = L.lenT(a->gpu);
  unsigned char *buf = xmalloc((size_t) l1 + l2 + lt);
  element_to_bytes(buf, in1);
  element_to_bytes(buf + l1, in2);
  if (L.pair(a->gpu, buf + l1 + l2, buf, buf + l1, 1)) gpu_failed("element_pairing");
  element_from_bytes(out, buf + l1 + l2);
  g_stat.map++;
  free(buf);
}
static void hip_prod(element_ptr out, element_t in1[], element_t in2[], int n_prod, struct pairing_s *p) {
  attach_t *a = find(p);
  int l1 = L.len1(a->gpu), l2 = L.len2(a->gpu), lt = L.lenT(a->gpu);
  if (n_prod <= 0) { element_set1(out); return; }         /* the empty product (the reference's loops leave out = 1) */
  unsigned char *b1 = xmalloc((size_t) n_prod * l1), *b2 = xmalloc((size_t) n_prod * l2), *bt = xmalloc(lt);
  for (int j = 0; j < n_prod; j++) { element_to_bytes(b1 + (size_t) j * l1, in1[j]); element_to_bytes(b2 + (size_t) j * l2, in2[j]); }
  if (L.prod(a->gpu, bt, b1, b2, 1, n_prod)) gpu_failed("element_prod_pairing");
  element_from_bytes(out, bt);
  g_stat.prod++;
  free(b1); free(b2); free(bt);
}

/* pairing->finalpow replacement (include/pbc_pairing.h:41; called by gt_random / gt_from_hash, ecc/pairing.c:121,127):
 * `e` is the GT element, its bytes are those of the element inside the wrapper */
static void hip_finalpow(element_t e) {
  attach_t *a = find(e->field->pairing);
  int lt = L.lenT(a->gpu);
  unsigned char *buf = xmalloc(2 * (size_t) lt);
  element_to_bytes(buf, e);
  if (L.finalpow(a->gpu, buf + lt, buf, 1)) gpu_failed("pairing->finalpow");
  element_from_bytes(e, buf + lt);
  g_stat.finalpow++;
  free(buf);
}

/* pairing->pp_init / pp_apply / pp_clear replacements (include/pbc_pairing.h:39-41, 54-89).
 * p->data holds the GPU handle (types a, a1, d, g) or a copy of the first argument (types e, f: no preprocessed form
 * on the GPU, every apply is an ordinary GPU pairing).  Every object is entered in g_pp (see there). */
static void pp_forget(void *pp) {                          /* a pairing_pp_t reused at the same address */
  pthread_mutex_lock(&g_lock);
  for (int i = 0; i < g_npp; ) if (g_pp[i].pp == pp) g_pp[i] = g_pp[--g_npp]; else i++;
  pthread_mutex_unlock(&g_lock);
}
static void pp_register(void *pp, void *data, struct pairing_s *pairing, int kind) {
  pp_forget(pp);
  pthread_mutex_lock(&g_lock);
  if (g_npp == g_cpp) {
    int nc = g_cpp ? 2 * g_cpp : 16;
    ppreg_t *t = realloc(g_pp, (size_t) nc * sizeof *t);
    if (!t) { pthread_mutex_unlock(&g_lock); pbc_die("pbc_hip: out of memory"); }
    g_pp = t; g_cpp = nc;
  }
  g_pp[g_npp++] = (ppreg_t) {pp, data, pairing, kind, 0};
  pthread_mutex_unlock(&g_lock);
}
/* 1 and *e filled when `p` was made by these hooks (take: remove the entry) */
static int pp_lookup(pairing_pp_t p, ppreg_t *e, int take) {
  int found = 0;
  pthread_mutex_lock(&g_lock);
  for (int i = 0; i < g_npp; i++)
    if (g_pp[i].pp == (void *) p && g_pp[i].data == p->data) {
      *e = g_pp[i];
      if (take) g_pp[i] = g_pp[--g_npp];
      found = 1;
      break;
    }
  pthread_mutex_unlock(&g_lock);
  return found;
}
static int retired_find(struct pairing_s *pairing, retired_t *out) {
  int found = 0;
  pthread_mutex_lock(&g_lock);
  for (int i = 0; i < g_nret; i++) if (g_ret[i].pairing == pairing) { *out = g_ret[i]; found = 1; break; }
  pthread_mutex_unlock(&g_lock);
  return found;
}
static void retired_drop(struct pairing_s *pairing) {
  pthread_mutex_lock(&g_lock);
  for (int i = 0; i < g_nret; ) if (g_ret[i].pairing == pairing) g_ret[i] = g_ret[--g_nret]; else i++;
  pthread_mutex_unlock(&g_lock);
}
static void pp_release(const ppreg_t *e) {                  /* the resources behind a live entry */
  if (e->kind == 1) L.pp_clear(e->data);
  else { element_clear(e->data); free(e->data); }
}

static void hip_pp_init(pairing_pp_t p, element_t in1, struct pairing_s *pairing) {
  attach_t *a = find(pairing);
  if (!a) {                                                /* detached, hooks still installed for older objects */
    retired_t r;
    if (!retired_find(pairing, &r)) pbc_die("pbc_hip: pairing_pp_init on a pairing that is not attached");
    pp_forget(p);
    r.cpu_pp_init(p, in1, pairing);
    return;
  }
  int t = L.type(a->gpu);
  if (t == 'a' || t == '1' || t == 'd' || t == 'g') {
    unsigned char *buf = xmalloc(L.len1(a->gpu));
    void *h = NULL;
    element_to_bytes(buf, in1);
    if (L.pp_init(&h, a->gpu, buf)) pbc_die("pbc_hip: pairing_pp_init: %s", L.err());
    p->data = h;
    g_stat.pp_init++;
    free(buf);
    pp_register(p, h, pairing, 1);
  } else {
    element_ptr c = xmalloc(sizeof(*c));
    element_init_same_as(c, in1);
    element_set(c, in1);
    p->data = c;
    pp_register(p, c, pairing, 2);
  }
}
static void hip_pp_clear(pairing_pp_t p) {
  ppreg_t e;
  if (pp_lookup(p, &e, 1)) { if (!e.dead) pp_release(&e); return; }
  attach_t *a = find(p->pairing);                          /* made on the CPU, before the attach or after a detach */
  retired_t r;
  if (a) a->cpu_pp_clear(p); else if (retired_find(p->pairing, &r)) r.cpu_pp_clear(p);
}
static void hip_pp_apply(element_t out, element_t in2, pairing_pp_t p) {
  ppreg_t e;
  attach_t *a = find(p->pairing);
  if (!pp_lookup(p, &e, 0)) {
    /* not made by these hooks.  On an attached pairing that is an object initialised before the attach: there is no
     * GPU form of it and no CPU path here (initialise it after pbc_hip_attach).  On a detached pairing the hook is only
     * still installed for older objects (see g_pp): the pairing is a stock CPU pairing again, pass through. */
    retired_t r;
    if (a || !retired_find(p->pairing, &r)) pbc_die("pbc_hip: pairing_pp_apply: this pairing_pp_t was not initialised on the GPU");
    r.cpu_pp_apply(out, in2, p);
    return;
  }
  if (e.dead || !a) pbc_die("pbc_hip: pairing_pp_apply on a pairing_pp_t whose pairing was cleared or detached from the GPU");
  if (e.kind == 2) { hip_map(out, e.data, in2, p->pairing); return; }
  int l2 = L.len2(a->gpu), lt = L.lenT(a->gpu);
  unsigned char *buf = xmalloc((size_t) l2 + lt);
  element_to_bytes(buf, in2);
  if (L.pp_apply(e.data, buf + l2, buf, 1)) pbc_die("pbc_hip: pairing_pp_apply: %s", L.err());
  element_from_bytes(out, buf + l2);
  g_stat.pp_apply++;
  free(buf);
}

/* outs[i] = e(P, in2[i]) for the P given to pairing_pp_init: pairing_pp_apply over a batch */
int pairing_pp_apply_batch(element_t out[], element_t in2[], size_t n, pairing_pp_t p) {
  if (!n) return 0;
  if (!p->pairing) { for (size_t i = 0; i < n; i++) element_set0(out[i]); return 0; }   /* P was O */
  attach_t *a = find(p->pairing);
  ppreg_t e;
  if (!a || !p->data || !pp_lookup(p, &e, 0) || e.dead) return 1;
  if (e.kind == 2) {
    /* no preprocessed form on the GPU for this type (e, f): p->data is a copy of the first
     * argument -- run the batch as n ordinary pairings */
    element_t *ps = malloc(n * sizeof(element_t));
    if (!ps) return 1;
    for (size_t i = 0; i < n; i++) ps[i][0] = *(element_ptr) p->data;     /* shallow, read-only */
    int rc = run_batch(a, out, ps, in2, n, 1);
    free(ps);
    return rc;
  }
  int l2 = L.len2(a->gpu), lt = L.lenT(a->gpu);
  unsigned char *b2 = malloc(n * l2 + 1), *bt = malloc(n * lt + 1);
  size_t *slot = malloc(n * sizeof *slot), m = 0;
  if (!b2 || !bt || !slot) { free(b2); free(bt); free(slot); return 1; }
  for (size_t i = 0; i < n; i++) {
    if (element_is0(in2[i])) { element_set0(out[i]); continue; }
    element_to_bytes(b2 + m * l2, in2[i]);
    slot[m++] = i;
  }
  int rc = m ? L.pp_apply(p->data, bt, b2, m) : 0;
  if (!rc) for (size_t i = 0; i < m; i++) element_from_bytes(out[slot[i]], bt + i * lt);
  free(b2); free(bt); free(slot);
  return rc;
}

/* out[i] = in[i]^zr[i] (element_pow_zn / element_mul_zn, include/pbc_field.h:311,374) for
 * elements of G1, of G2 when the pairing is symmetric, or of GT; identity inputs stay identity. */
int element_pow_zn_batch(element_t out[], element_t in[], element_t zr[], size_t n) {
  if (!n) return 0;
  struct pairing_s *p = in[0]->field->pairing;
  attach_t *a = find(p);
  if (!a) return 1;
  int is_gt = in[0]->field == p->GT;
  int group = in[0]->field == p->G1 ? 1 : (in[0]->field == p->G2 ? 2 : 0);
  if (!is_gt && !group) return 1;
  int le = element_length_in_bytes(in[0]), lz = L.lenZr(a->gpu);
  unsigned char *be = malloc(n * le + 1), *bz = malloc(n * lz + 1), *bo = malloc(n * le + 1);
  size_t *slot = malloc(n * sizeof *slot), m = 0;
  if (!be || !bz || !bo || !slot) { free(be); free(bz); free(bo); free(slot); return 1; }
  for (size_t i = 0; i < n; i++) {
    if (element_is0(in[i]) || element_is0(zr[i])) { element_set0(out[i]); continue; }   /* x^0 = O^k = identity */
    element_to_bytes(be + m * le, in[i]);
    element_to_bytes(bz + m * lz, zr[i]);
    slot[m++] = i;
  }
  int rc = !m ? 0 : is_gt ? L.gt_pow(a->gpu, bo, be, bz, m) : L.mul_zn(a->gpu, group, bo, be, bz, m);
  if (rc) pbc_error("pbc_hip: %s", L.err());
  else for (size_t i = 0; i < m; i++) element_from_bytes(out[slot[i]], bo + i * le);
  free(be); free(bz); free(bo); free(slot);
  return rc;
}

/* out[i] = element_from_hash(data + i * hlen, hlen) (include/pbc_field.h:257 -> curve_from_hash,
 * ecc/curve.c:455-482) for elements of G1 or G2: hashing, square roots and the cofactor multiplication run on
 * the device. */
int element_from_hash_batch(element_t out[], const void *data, int hlen, size_t n) {
  if (!n) return 0;
  struct pairing_s *p = out[0]->field->pairing;
  attach_t *a = find(p);
  if (!a) return 1;
  int group = out[0]->field == p->G1 ? 1 : (out[0]->field == p->G2 ? 2 : 0);
  if (!group) return 1;
  int le = element_length_in_bytes(out[0]);
  unsigned char *bo = malloc(n * (size_t) le + 1);
  if (!bo) return 1;
  int rc = L.from_hash(a->gpu, group, bo, data, hlen, n);
  if (rc) pbc_error("pbc_hip: %s", L.err());
  else for (size_t i = 0; i < n; i++) element_from_bytes(out[i], bo + i * (size_t) le);
  free(bo);
  return rc;
}

static void report_at_exit(void) {
  const char *e = getenv("PBC_HIP_VERBOSE");
  if (e && *e == '1')
    fprintf(stderr, "pbc_hip: on the GPU: %lu element_pairing, %lu element_prod_pairing, %lu pairing_pp_init, %lu pairing_pp_apply, "
            "%lu units in batch calls, %lu finalpow\n", g_stat.map, g_stat.prod, g_stat.pp_init, g_stat.pp_apply, g_stat.batch_units,
            g_stat.finalpow);
}
/* pairing->clear_func replacement: drop the GPU object, then run the pairing's own clean-up (pairing_clear,
 * ecc/pairing.c:104-106) */
static void hip_clear(struct pairing_s *p) {
  attach_t *a = find(p);
  if (!a) return;
  void (*cpu_clear)(struct pairing_s *) = a->cpu_clear;
  pbc_hip_detach(p);
  if (cpu_clear) cpu_clear(p);
}
int pbc_hip_attach(pairing_t pairing, const char *param, size_t len) {
  static int registered;
  if (load_lib()) return 1;
  if (find(pairing)) return 1;                            /* attached already (detach first to re-attach) */
  pbc_hip_pairing_t *g;
  if (L.init(&g, param, len)) { pbc_error("pbc_hip: %s", L.err()); return 1; }
  if (L.len1(g) != pairing_length_in_bytes_G1(pairing) || L.len2(g) != pairing_length_in_bytes_G2(pairing) ||
      L.lenT(g) != pairing_length_in_bytes_GT(pairing)) { L.clear(g); pbc_error("pbc_hip: record lengths differ from the pairing's"); return 1; }
  {
    /* PBC_HIP_DEVICES=all (or a comma list of ordinals): batches are range-split over these GPUs */
    const char *e = getenv("PBC_HIP_DEVICES");
    int devs[16], nd = 0;
    if (e && !strcmp(e, "all")) { int c = L.device_count(); for (nd = 0; nd < c && nd < 16; nd++) devs[nd] = nd; }
    else if (e) { for (const char *q = e; *q && nd < 16; ) { devs[nd++] = atoi(q); q = strchr(q, ','); if (!q) break; q++; } }
    if (nd > 0 && L.use_devices(g, devs, nd)) { pbc_error("pbc_hip: %s", L.err()); L.clear(g); return 1; }
  }
  attach_t *a = add_entry();
  if (!a) { L.clear(g); return 1; }
  if (!registered) { registered = 1; atexit(report_at_exit); }
  a->pairing = pairing; a->gpu = g; a->cpu_map = pairing->map; a->cpu_prod = pairing->prod_pairings;
  a->cpu_pp_init = pairing->pp_init; a->cpu_pp_clear = pairing->pp_clear; a->cpu_pp_apply = pairing->pp_apply;
  {
    /* detached earlier with pairing_pp_t objects alive: the pp hooks are still installed, the CPU routines are in the
     * retired record; a record left by another pairing that lived at this address is stale */
    retired_t r;
    if (retired_find(pairing, &r) && pairing->pp_clear == hip_pp_clear) { a->cpu_pp_init = r.cpu_pp_init; a->cpu_pp_clear = r.cpu_pp_clear; a->cpu_pp_apply = r.cpu_pp_apply; }
    retired_drop(pairing);
  }
  a->cpu_clear = pairing->clear_func;
  a->cpu_finalpow = pairing->finalpow;
  pairing->finalpow = hip_finalpow;
  pairing->map = hip_map;
  pairing->prod_pairings = hip_prod;
  pairing->clear_func = hip_clear;
  /* preprocessed pairings exist on the GPU for types a, a1 ('1'), d and g; for the others pairing_pp_t keeps a copy of
   * the first argument (hip_pp_init) */
  pairing->pp_init = hip_pp_init; pairing->pp_clear = hip_pp_clear; pairing->pp_apply = hip_pp_apply;
  return 0;
}
void pbc_hip_detach(pairing_t pairing) {
  attach_t *a = find(pairing);
  if (!a) return;
  pairing->map = a->cpu_map; pairing->prod_pairings = a->cpu_prod;
  pairing->clear_func = a->cpu_clear;
  pairing->finalpow = a->cpu_finalpow;
  /* pairing_pp_t objects of this pairing that are still alive: release what they hold now (the GPU object goes away, the
   * fields may be cleared next), keep their entries so that a later pairing_pp_clear is recognised */
  int live = 0;
  for (;;) {
    ppreg_t e = {0};
    int have = 0;
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < g_npp && !have; i++)
      if (g_pp[i].pairing == pairing && !g_pp[i].dead) { g_pp[i].dead = 1; e = g_pp[i]; have = 1; }
    pthread_mutex_unlock(&g_lock);
    if (!have) break;
    pp_release(&e);
  }
  pthread_mutex_lock(&g_lock);
  for (int i = 0; i < g_npp; i++) live += g_pp[i].pairing == pairing;
  pthread_mutex_unlock(&g_lock);
  if (live) {
    pthread_mutex_lock(&g_lock);
    if (g_nret == g_cret) {
      int nc = g_cret ? 2 * g_cret : 8;
      retired_t *t = realloc(g_ret, (size_t) nc * sizeof *t);
      if (!t) { pthread_mutex_unlock(&g_lock); pbc_die("pbc_hip: out of memory"); }
      g_ret = t; g_cret = nc;
    }
    g_ret[g_nret++] = (retired_t) {pairing, a->cpu_pp_init, a->cpu_pp_clear, a->cpu_pp_apply};
    pthread_mutex_unlock(&g_lock);
  } else {
    pairing->pp_init = a->cpu_pp_init; pairing->pp_clear = a->cpu_pp_clear; pairing->pp_apply = a->cpu_pp_apply;
  }
  L.clear(a->gpu);
  drop_entry(a);
}
int element_pairing_batch(element_t out[], element_t in1[], element_t in2[], size_t n) {
  if (!n) return 0;
  attach_t *a = find(out[0]->field->pairing);
  return a ? run_batch(a, out, in1, in2, n, 1) : 1;
}
int element_prod_pairing_batch(element_t out[], element_t in1[], element_t in2[], size_t n, int k) {
  if (!n) return 0;
  attach_t *a = find(out[0]->field->pairing);
  return a ? run_batch(a, out, in1, in2, n, k) : 1;
}
