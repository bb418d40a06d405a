/*
 * pbc_hip.h -- C-ABI of libpbc_hip.so: the MI355X (gfx950) batched bilinear-pairing engine.
 *
 * This is the drop-in boundary for PBC's pairing hot path.  Every entry point takes plain
 * pointers and sizes; group elements cross the boundary in PBC's own element_to_bytes /
 * element_from_bytes wire format:
 *     F_q      fixed-width big-endian canonical residue      (arith/montfp.c:487-517)
 *     F_q^2    re || im                                       (arith/fieldquadratic.c:323-337)
 *     polymod  c0 || ... || c(n-1)                            (arith/poly.c:718-752)
 *     point    x || y, off-curve bytes deserialise to O       (ecc/curve.c:603-623)
 *     GT       the underlying extension-field element          (ecc/pairing.c:175-185)
 * so a PBC maintainer binds it with element_to_bytes/element_from_bytes only
 * (see INTEGRATION.md for the stub that installs these behind pairing->map /
 * pairing->prod_pairings).
 *
 * All functions return 0 on success, non-zero on failure (pairing_init_set_buf
 * convention, ecc/pairing.c:88-98); pbc_hip_last_error() gives the message.  A
 * pbc_hip_pairing_t is not re-entrant for the host-buffer forms: one host-buffer batch call at a time per object.  The
 * _dev forms may be enqueued on several streams of one object at once and from several host threads, also on the SAME
 * (object, stream) pair: the scratch of a call (the flags of the two-pass group operations -- fast kernel + complete kernel
 * for the lanes it flagged --, the term records of the products) lives in a workspace keyed by (device, stream) whose
 * issue lock the call holds while it enqueues its kernels, so the calls of two threads on one stream are enqueued one
 * after the other, never interleaved (round 6; tests/test_gpu_group2.py::test_two_threads_issue_on_one_stream).  The
 * per-launch unit counters ("hip_dynamic 1") are slots of a ring, one per launch.  Device memory: every host-buffer form
 * works on buffers the object keeps (hipMalloc; pbc_hip_pairing_release_workspaces / pbc_hip_pairing_clear free them); only
 * the _dev forms of element_pow2/3_zn on the composition route and of the limb-image calls take a stream-ordered temporary (hipMallocAsync on the
 * caller's stream) -- observed to misbehave under the system HIP runtime when called from short-lived threads (round 6), so
 * nothing the reference-side glue reaches uses it.
 *
 * Input classes (every one is covered by a GPU test, tests/test_gpu_parity.py):
 *   - Points of the whole curve.  curve_from_bytes (ecc/curve.c:609-623) checks the curve equation only, so a G1 / G2
 *     record may hold a point outside the order-r subgroup.  Pairings, products and element_mul_zn give the
 *     reference's bytes for such points as well (the group law is complete, scalars may exceed r).  The one
 *     exception is inherited: for type e (k = 1) the reference's own value for a point with r P != O depends on the
 *     auxiliary point it draws at random at init (e_param.c:866-870), so there is nothing to agree with.
 *   - Off-curve records deserialise to O: the pairing is the identity of GT, a product containing one is the
 *     identity, [k] O = O.
 *   - Zero-filled records.  All-zero bytes are what element_to_bytes writes for O, and they are treated as O by
 *     every entry point: pairing = identity, product = identity, [k] 0 = 0.  Where the curve has b != 0 they are
 *     off the curve anyway.  On y^2 = x^3 + x (types a, a1) they are the 2-torsion point (0, 0); its pairing value is
 *     the identity too (2 is coprime to r), while the reference divides by zero there (mpz_invert of 0 in
 *     point_to_affine, a_param.c:1073-1080) and returns an arbitrary value.
 *     Pairings of other points of tiny order (3, 4, 6, 12 on type a) are unspecified here as in the reference: the
 *     Miller loop runs through V = O or V = +-P, where the reference divides by zero.  element_mul_zn is exact
 *     for them.
 *   - Coordinates >= q are reduced mod q on load, as fp_from_bytes does (montfp.c:498-517).
 */
#ifndef PBC_HIP_H
#define PBC_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pbc_hip_pairing_s pbc_hip_pairing_t;

/* Replaces pairing_init_set_buf / pairing_init_set_str (include/pbc_pairing.h:95-102,
 * ecc/pairing.c:88-106): parse a PBC parameter text (types a, a1, d, e, f, g; field sizes: a / a1 / e
 * up to 1056 bits, d up to 224, f up to 256, g 160), derive the per-curve constants (a_init_pairing
 * ecc/a_param.c:1431-1472, a1_init_pairing :2230-2273, d_init_pairing ecc/d_param.c:993-1095,
 * e_init_pairing ecc/e_param.c:832-872, f_init_pairing ecc/f_param.c:335-447, g_init_pairing
 * ecc/g_param.c:1248-1354) and bind the
 * object to the calling thread's current HIP device.  len == 0 means strlen(param).
 *
 * Keys beyond PBC's own in the parameter text select implementation variants of THIS library (PBC ignores unknown keys,
 * so the same text still initialises the reference).  Every variant gives the same bytes; they exist for same-box A/B
 * measurements and for the tests, and the defaults are the measured winners:
 *   hip_wave_max N       type a: batches up to N units take the one-pairing-per-wavefront kernels (default 5120, 0 = never)
 *   hip_wave4_max N      ... and up to N units four wavefronts per pairing (default 768)
 *   hip_wave2_max N      ... and above that, up to N units, two (default 1280)
 *   hip_dynamic 1        resident kernels fetch their units from a per-launch counter instead of a fixed stride
 *   hip_no_fair 1        resident kernels without the time-sliced wave priorities
 *   hip_resident_slots N workgroups of a resident launch instead of the occupancy query (tests: forces many strides)
 *   hip_zero_copy 0      host-buffer calls stage page-locked buffers through device memory instead of working in place
 *   hip_host_chunk N     host-buffer calls: units per staged chunk
 *   hip_prod_shared 1    type a products: one product per lane with shared squarings (the reference's shape) instead of
 *                        one term per lane;  hip_prod_chunk N: terms per launch of the one-term-per-lane kernels
 *   hip_group_slow 1     group operations: only the complete ladders / generic powers (no fast pass)
 *   hip_no_limb 1        types d, f: word-form steps on E(F_q) in the pairing kernels, word-form ladders on G1
 *   hip_no_xs 1, hip_no_bm1 1, hip_no_cyc 1, hip_no_bn 1
 *                        type f: the parameter file's xi instead of the sparse one / its basis of F_q^2 instead of the
 *                        i-basis / plain squarings instead of Granger-Scott ones / the generic hard part instead of the
 *                        BN chain */
int pbc_hip_pairing_init_set_buf(pbc_hip_pairing_t **out, const char *param, size_t len);
/* Replaces pairing_clear (include/pbc_pairing.h:109-116). */
void pbc_hip_pairing_clear(pbc_hip_pairing_t *p);

/* Multi-GPU for the host-buffer entry points (element_pairing_batch, element_prod_pairing_batch):
 * the batch is range-split over `devices` (HIP ordinals; a device may be listed more than once),
 * every unit is independent, so there is no exchange between devices and each result is copied
 * straight into the caller's buffer.  n = 0 returns to the single device the object was created
 * on.  The *_dev entry points always run on the device that owns the caller's pointers. */
int pbc_hip_pairing_use_devices(pbc_hip_pairing_t *p, const int *devices, int n);
/* Number of visible HIP devices (hipGetDeviceCount; 0 when there is none). */
int pbc_hip_device_count(void);
/* Page-locked host memory for the host-buffer entry points.  When all three buffers of a call (gt, g1, g2) are page-locked
 * and 16-byte aligned -- from here, hipHostMalloc, hipHostRegister or a framework's pinned allocator; for an object with a device set:
 * allocated with hipHostMallocPortable, as this call does -- and the field's coordinates are a multiple of four bytes
 * long (every shipped parameter set but a1.param, g149.param and d201.param), the kernels read the records and write the results IN PLACE
 * over PCIe: no staging copies, the call costs the kernel's time (measured: 2^20 type a pairings pinned host -> pinned
 * host in 81.6 ms against 81.7 ms for HBM-resident data and 90.1 ms with staged copies).  "hip_zero_copy 0" in the
 * parameter text turns it off.  Any other memory is staged through device buffers (pageable memory: synchronously, by the
 * runtime).  The buffers must stay allocated and unmodified until the call returns (it returns after the kernels).  PBC
 * has no counterpart; the glue (integration/pbc_hip_glue.c) marshals element_t arrays into such buffers. */
int pbc_hip_host_alloc(void **out, size_t bytes);
void pbc_hip_host_free(void *p);
/* 'a', 'd', 'f', 'g', or '1' for a1 (the "type" key, ecc/param.c:172-205). */
int pbc_hip_pairing_type(const pbc_hip_pairing_t *p);
/* Replace pairing_length_in_bytes_{G1,G2,GT} (include/pbc_pairing.h:183-238). */
int pbc_hip_pairing_length_in_bytes_G1(const pbc_hip_pairing_t *p);
int pbc_hip_pairing_length_in_bytes_G2(const pbc_hip_pairing_t *p);
int pbc_hip_pairing_length_in_bytes_GT(const pbc_hip_pairing_t *p);

/* Batched element_pairing (include/pbc_pairing.h:141-145 -> pairing_apply :118-135 ->
 * pairing->map = a_pairing_proj, ecc/a_param.c:1053-1198):
 *     gt[i] = to_bytes( e( from_bytes_G1(g1[i]), from_bytes_G2(g2[i]) ) ),  i < n
 * including the identity short-circuit (an input that deserialises to O gives GT's 1).
 * Host buffers, AoS: g1 is n*len_G1 bytes, g2 n*len_G2, gt n*len_GT. */
int pbc_hip_element_pairing_batch(pbc_hip_pairing_t *p, uint8_t *gt, const uint8_t *g1,
                                  const uint8_t *g2, size_t n);
/* Same with device-resident buffers, enqueued on `stream` (a hipStream_t; NULL = default
 * stream).  Asynchronous: returns after the launch.
 * Batch sizes (MI355X, a.param; profiles/r04_sweep_*.json, tools/wave_latency.py).  The throughput kernels give every LANE a
 * whole pairing: a launch costs one lane's 2.1 M dependent multiply-adds however small it is -- 6.4 ms for any
 * n <= 32768 (one wave per SIMD), 10.3 ms up to 131072 (two waves per SIMD: one chip residency of 1024 workgroups x 128
 * lanes), and from there the resident workgroups walk the batch in strides of the residency, so the time grows in
 * steps of ceil(n / 131072) x 10 ms (2^20: 79.5 ms, 13.1 M pairings/s).  A tail of up to 5120 units beyond a whole number
 * of strides goes to the small-batch kernels below instead of paying a whole stride (131073 units: 11.9 ms, not 17.0);
 * beyond that, feed multiples of 131072 where you can.  Type f: 7.2-8.3 ms / 11-12 ms / steps of 11-12 ms, no tail kernel.
 * Cut-over for small batches (type a, 512-bit q): up to 5120 units ("hip_wave_max N" in the parameter text moves it,
 * 0 disables it) a launch gives every pairing a WAVEFRONT (csrc/pairing_aw.cuh: one limb per lane, products across
 * the lanes), up to 768 units ("hip_wave4_max N") a workgroup of FOUR wavefronts that share the independent products
 * of every step, up to 1280 ("hip_wave2_max N") a workgroup of two: 1.0 ms for one pairing (n <= 256), 1.3 ms at 512,
 * 1.7 ms at 1024, 2.6 ms at 2048, 4.4 ms at 4096, 5.3 ms at 5120 (the throughput kernel: 5.8-6.5 ms depending on the
 * box) -- same bytes as the throughput kernel.  pairing_pp_apply takes the same three forms (0.68 ms for one second
 * argument, 1.04 ms at 1024, 3.0 ms at 5120; lane kernel 3.1-3.5 ms), and products of k terms with n k <= hip_wave_max
 * give every TERM a workgroup and then every product one: 1.0 ms for one product of 2 .. 16 terms (lane kernels: 5.7-6.4).
 *
 * Batch sizes for the 33-word fields (a1.param, e.param, type a above 512 bits): the kernels hold TWO waves per SIMD in
 * 256-lane workgroups, i.e. 2^17 units fill an MI355X once (e.param: 40 ms for any batch up to 2^16 units, 68 ms for
 * 2^17, 131 ms for 2^18); batches of 2^18 and more amortise the tail of a launch.
 * Small batches of type a1 and of type a parameter sets outside the 512-bit fast path (round 6): where q leaves ten bits of the
 * limb radix free and fills twelve bits of its top limb (a1.param, the 1024-bit and 253-bit generated type a sets; the object decides
 * at init) a launch of up to 12 288 units on the 33-word fields / 6144 on the 16-word ones ("hip_wave_max N", 0 = never) gives
 * every pairing -- every TERM of a product, then every product; every second argument of pairing_pp_apply -- a workgroup of four
 * wavefronts ("hip_wave4_max N": above it one) on the same limb-per-lane routines with the Miller loop over the signed digits of
 * the order: a1.param 9.2 ms a pairing or a product of any few terms and 4.1 ms a pairing_pp_apply, where one lane needs 0.2 s /
 * 0.2 s a term / 70 ms (a1_pairing, a1_pairings_affine, a1_pairing_pp_apply: ecc/a_param.c:1840-2015, :2100-2193, :1728-1818).
 * ("hip_wave8_max N", off by default: eight wavefronts per unit and three rounds of products a doubling step instead of five for up to
 * N terms -- a1.param 7.7 ms; opt-in because of an unexplained abort later in the same process: DESIGN.md 4.5.)
 * Type e under the same conditions on q (csrc/pairing_ew.cuh; e_pairing, ecc/e_param.c:472-483, and generic_prod_pairings,
 * ecc/pairing.c:35-46): four wavefronts per unit up to 512 units, one up to 10 240 -- e.param 3.2 ms a pairing or a product of
 * few terms where one lane needs 34.5 ms a term. */
int pbc_hip_element_pairing_batch_dev(pbc_hip_pairing_t *p, void *d_gt, const void *d_g1,
                                      const void *d_g2, size_t n, void *stream);

/* Batched element_prod_pairing (include/pbc_pairing.h:153-171 -> pairing->prod_pairings =
 * a_pairings_affine, ecc/a_param.c:1283-1383): n products of k terms each,
 *     gt[u] = to_bytes( prod_{j<k} e(g1[u*k+j], g2[u*k+j]) ),
 * with the reference's rule that ANY identity input makes the whole product 1 (:161-168). */
int pbc_hip_element_prod_pairing_batch(pbc_hip_pairing_t *p, uint8_t *gt, const uint8_t *g1,
                                       const uint8_t *g2, size_t n, int k);
int pbc_hip_element_prod_pairing_batch_dev(pbc_hip_pairing_t *p, void *d_gt, const void *d_g1,
                                           const void *d_g2, size_t n, int k, void *stream);
/* The product kernels of types a, d and g use a device workspace: one buffer per (device, stream) a *_dev product call was
 * enqueued on, grown on demand and kept by the object.  Types d / g keep the Miller state of every term there:
 * ceil(n / 128) * 128 * k * R  bytes with R = 4 (2 d N + 5 L) (N words and L 29-bit limbs per F_q element, d = 3 or 5 --
 * 240 for d159), at most one chip residency of workgroups where the kernel runs resident.  Type a with a 512-bit q runs
 * one TERM per lane and leaves each term's Miller value in a 160-byte record for the kernel that multiplies them:
 * min(n k, 2^22) * 160 bytes (640 MB at most; longer batches are cut into launches of 2^22 terms).  At most 8 buffers are
 * kept (least recently used first out, after a device synchronisation); the host-buffer entry points use up to three per
 * device.  Everything is freed by pbc_hip_pairing_clear; this call frees the buffers now (it synchronises the devices
 * they live on). */
int pbc_hip_pairing_release_workspaces(pbc_hip_pairing_t *p);

/* Preprocessed pairings with a fixed first argument (the BLS shape: one public key or
 * generator, many signatures).  Replace pairing_pp_init / pairing_pp_clear / pairing_pp_apply
 * (include/pbc_pairing.h:54-89 -> a_pairing_pp_init ecc/a_param.c:149-220, a_pairing_pp_apply
 * :317-360).  pp_init derives the per-step line coefficients of g1 on the device once;
 * pp_apply_batch computes gt[i] = e(g1, g2[i]) for a batch of second arguments with 7 instead of
 * 18 F_q products per Miller step.  Results are identical to element_pairing (identity rules
 * included: a g1 or g2[i] that deserialises to O gives 1).  Type A. */
typedef struct pbc_hip_pp_s pbc_hip_pp_t;
int pbc_hip_pairing_pp_init(pbc_hip_pp_t **pp, pbc_hip_pairing_t *p, const uint8_t *g1);
void pbc_hip_pairing_pp_clear(pbc_hip_pp_t *pp);
int pbc_hip_pairing_pp_apply_batch(pbc_hip_pp_t *pp, uint8_t *gt, const uint8_t *g2, size_t n);
int pbc_hip_pairing_pp_apply_batch_dev(pbc_hip_pp_t *pp, void *d_gt, const void *d_g2, size_t n, void *stream);

/* Batched group operations next to the pairing (the callers' other hot loops: example/bls.c
 * signs with element_pow_zn and checks with GT products).  Scalars are Z_r elements in
 * element_to_bytes form: big-endian, pbc_hip_pairing_length_in_bytes_Zr() bytes
 * (pairing_length_in_bytes_Zr, include/pbc_pairing.h:235-238), values < r.
 *   element_mul_zn / element_pow_zn on G1, G2 (include/pbc_field.h:311, :374 ->
 *     generic_pow_mpz arith/field.c:113-126 over curve_mul ecc/curve.c:153-207):
 *     out[i] = [zr[i]] in[i]; group = 1 or 2 (G2 of types d / g / f is the twist over F_q^d /
 *     F_q^2: field_reinit_curve_twist ecc/curve.c:885-901, f_param.c:372-383).  The point at
 *     infinity (only reachable from off-curve input or a zero scalar) is written as zero bytes.
 *   element_mul on GT (include/pbc_field.h:280 -> mulg wrapper ecc/pairing.c:135-283);
 *   element_pow_zn on GT. */
int pbc_hip_pairing_length_in_bytes_Zr(const pbc_hip_pairing_t *p);
/* element_from_hash on G1 / G2 (include/pbc_field.h:257 -> curve_from_hash ecc/curve.c:455-482,
 * fp_from_hash arith/montfp.c:440-448, pbc_mpz_from_hash arith/field.c:643-668): n digests of hlen
 * bytes each -> n points, including the cofactor multiplication.  group = 1 for every type (any
 * odd q: Tonelli-Shanks where q = 1 mod 4); group = 2 for the symmetric types a, a1, e and for the twists of
 * types d, g, f (x from polymod_from_hash, arith/poly.c:341-348, or fq_from_hash, fieldquadratic.c:311-316;
 * square roots in F_q^d / F_q^2 on the device; those curves carry no cofactor: d_param.c:1057, f_param.c:383,
 * g_param.c:1319). */
int pbc_hip_element_from_hash_batch(pbc_hip_pairing_t *p, int group, uint8_t *out, const uint8_t *data,
                                    int hlen, size_t n);
/* element_to_bytes_compressed / element_from_bytes_compressed (ecc/curve.c:762-773, :800-815;
 * pairing_length_in_bytes_compressed_G1, include/pbc_pairing.h:199-204) on G1 and G2: records of
 * length_in_bytes(x) + 1 bytes, x || s with s = 1 when element_sign(y) > 0 (F_q: the canonical y is odd;
 * F_q^d / F_q^2 of the twists: the first non-zero coefficient of y is, poly.c:1189-1199, fieldquadratic.c:159-165).
 * Decompression takes the square root on the device; an x with no point above it gives zeros. */
int pbc_hip_pairing_length_in_bytes_compressed_G1(const pbc_hip_pairing_t *p);
int pbc_hip_pairing_length_in_bytes_compressed_G2(const pbc_hip_pairing_t *p);   /* include/pbc_pairing.h:210-216 */
int pbc_hip_element_to_bytes_compressed_batch(pbc_hip_pairing_t *p, int group, uint8_t *out, const uint8_t *in,
                                              size_t n);
int pbc_hip_element_from_bytes_compressed_batch(pbc_hip_pairing_t *p, int group, uint8_t *out, const uint8_t *in,
                                                size_t n);
/* element_to_bytes_x_only / element_from_bytes_x_only (ecc/curve.c:821-836; pairing_length_in_bytes_x_only_G1,
 * include/pbc_pairing.h:187-193): records of length_in_bytes_Fq bytes holding x alone.  from_bytes rebuilds the
 * point with the square root the reference's element_sqrt returns when that is defined (q = 3 mod 4: types a, a1,
 * f.param, g149.param); for q = 1 mod 4 the reference's own root depends on the random non-residue it drew at
 * start-up, and the result agrees with it up to the sign of y.  An x with no point above it gives zeros.
 * group = 2 on the twists of types d, g, f as well: type f with q = 3 mod 4 reproduces the reference's root
 * (fq_sqrt, arith/fieldquadratic.c:357-392, is a formula over roots in F_q); types d and g take their roots with
 * a randomised algorithm in the reference (polymod_sqrt, arith/poly.c:634-700): up to the sign of y. */
int pbc_hip_pairing_length_in_bytes_x_only_G1(const pbc_hip_pairing_t *p);
int pbc_hip_pairing_length_in_bytes_x_only_G2(const pbc_hip_pairing_t *p);   /* include/pbc_pairing.h:218-224 */
int pbc_hip_element_to_bytes_x_only_batch(pbc_hip_pairing_t *p, int group, uint8_t *out, const uint8_t *in,
                                          size_t n);
int pbc_hip_element_from_bytes_x_only_batch(pbc_hip_pairing_t *p, int group, uint8_t *out, const uint8_t *in,
                                            size_t n);
int pbc_hip_element_mul_zn_batch(pbc_hip_pairing_t *p, int group, uint8_t *out, const uint8_t *in,
                                 const uint8_t *zr, size_t n);
int pbc_hip_element_mul_GT_batch(pbc_hip_pairing_t *p, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n);
int pbc_hip_element_pow_zn_GT_batch(pbc_hip_pairing_t *p, uint8_t *out, const uint8_t *a, const uint8_t *zr,
                                    size_t n);
/* How the scalar multiplications run (the reference: generic_pow_mpz, a sliding window over the affine group law with one
 * mpz_invert per group operation, arith/field.c:14-126).  Here a regular signed fixed-window ladder (w = 4: per-lane table
 * of the odd multiples P .. 15P, four doublings and one mixed addition per window, two inversions in all), for the
 * 512-bit type a field on the limb-form arithmetic of the pairing kernel; powers in GT of type a by the Lucas ladder for
 * elements of norm 1 (every pairing value).  These ladders use the plain Jacobian formulas; a lane whose result is O,
 * whose point has small order or whose scalar is exceptional is detected (Z = 0) and handed to a second, complete pass
 * (two group operations per bit, any point of the curve, any scalar) that runs for those lanes only.  "hip_group_slow 1"
 * in the parameter text selects the complete pass for every lane.
 *
 * Host-buffer forms: as the pairing entry points -- range split over the object's device set, page-locked buffers worked
 * on in place where a lane reads its records once with word loads (GT products, final powers, the type a ladders with
 * scalars of a whole number of words), anything else staged through chunk buffers the object keeps (no allocation in the
 * steady state).  out may be the same buffer as an input.
 * _dev forms: device-resident buffers, enqueued on `stream` (NULL = default stream), asynchronous; a hash -> mul_zn ->
 * prod_pairing pipeline (example/bls.c:41-117 as a batch) stays on the device throughout.  out == in (exactly) is allowed
 * for element_mul_zn and the GT operations. */
int pbc_hip_element_mul_zn_batch_dev(pbc_hip_pairing_t *p, int group, void *d_out, const void *d_in, const void *d_zr,
                                     size_t n, void *stream);
int pbc_hip_element_mul_GT_batch_dev(pbc_hip_pairing_t *p, void *d_out, const void *d_a, const void *d_b, size_t n, void *stream);
int pbc_hip_element_pow_zn_GT_batch_dev(pbc_hip_pairing_t *p, void *d_out, const void *d_a, const void *d_zr, size_t n,
                                        void *stream);
int pbc_hip_finalpow_batch_dev(pbc_hip_pairing_t *p, void *d_out, const void *d_in, size_t n, void *stream);
int pbc_hip_element_from_hash_batch_dev(pbc_hip_pairing_t *p, int group, void *d_out, const void *d_data, int hlen, size_t n,
                                        void *stream);
int pbc_hip_element_to_bytes_compressed_batch_dev(pbc_hip_pairing_t *p, int group, void *d_out, const void *d_in, size_t n, void *stream);
int pbc_hip_element_from_bytes_compressed_batch_dev(pbc_hip_pairing_t *p, int group, void *d_out, const void *d_in, size_t n, void *stream);
int pbc_hip_element_to_bytes_x_only_batch_dev(pbc_hip_pairing_t *p, int group, void *d_out, const void *d_in, size_t n, void *stream);
int pbc_hip_element_from_bytes_x_only_batch_dev(pbc_hip_pairing_t *p, int group, void *d_out, const void *d_in, size_t n, void *stream);

/* Fixed-base powers: replace element_pp_init / element_pp_pow_zn / element_pp_clear (include/pbc_field.h:591-625 ->
 * element_build_base_table / element_pow_base_table, arith/field.c:243-323: a table of in^(w 2^(5 i)), a power = a product
 * of table entries) for a base in G1, G2 (group 1, 2) or GT (group 3) -- the BLS shape: one generator, many secret keys or
 * signatures' scalars (example/bls.c:41-62).  pp_init builds [w 2^(8 i)] B, w = 1 .. 255, i < length_in_bytes_Zr, on the
 * device (one entry per lane); a power is then one mixed addition per scalar BYTE and one inversion (GT: one product per
 * byte), no doublings.  Results are those of element_pow_zn / element_mul_zn on the same base.  A base of small order
 * (some table entry is O) or off the curve (= O) is served by the complete ladder.  The table is built on, and bound to,
 * the device the pairing object was created on (whatever the calling thread's current device is); the host-buffer form
 * runs there also when the object has a device set (staged); the _dev form needs pointers of that device. */
typedef struct pbc_hip_element_pp_s pbc_hip_element_pp_t;
int pbc_hip_element_pp_init(pbc_hip_element_pp_t **pp, pbc_hip_pairing_t *p, int group, const uint8_t *in);
void pbc_hip_element_pp_clear(pbc_hip_element_pp_t *pp);
int pbc_hip_element_pp_pow_zn_batch(pbc_hip_element_pp_t *pp, uint8_t *out, const uint8_t *zr, size_t n);
int pbc_hip_element_pp_pow_zn_batch_dev(pbc_hip_element_pp_t *pp, void *d_out, const void *d_zr, size_t n, void *stream);
/* pairing->finalpow (include/pbc_pairing.h:41; a_finalpow ecc/a_param.c:1420-1429, cc_finalpow ecc/d_param.c:566-568,
 * f_finalpow ecc/f_param.c:285-287, g_finalpow ecc/g_param.c:1162-1164, e_finalpow ecc/e_param.c:828-830; its callers
 * are gt_random / gt_from_hash, ecc/pairing.c:121,127): out[i] = in[i]^((q^k - 1)/r), the final exponentiation alone,
 * for n records of GT's underlying field in GT's wire format.  in[i] = 0 is outside the reference's contract too. */
int pbc_hip_finalpow_batch(pbc_hip_pairing_t *p, uint8_t *out, const uint8_t *in, size_t n);   /* (_dev form: above) */

/* The reference's own limb image as the exchange format (round 6).  element_to_bytes / element_from_bytes cost the host
 * a Montgomery reduction, a GMP export and two allocations per F_q coordinate (arith/montfp.c:487-517, :122-139); an
 * integration that sits INSIDE PBC can hand over what a montfp element holds instead (the per-element data of
 * arith/montfp.c:36-39: t = ceil(bits(q) / 64) little-endian 64-bit limbs of x 2^(64 t) mod q, fully reduced; a zero
 * element -- flag 0, :93-100 -- as t zero limbs) and take the results back the same way.  A limb-image record is the wire
 * record with every length_in_bytes(F_q)-byte coordinate replaced by its 8 t bytes, in the same order (x || y,
 * coefficient 0 first: fq_to_bytes fieldquadratic.c:323-329, polymod_to_bytes poly.c:718-733, curve_to_bytes
 * curve.c:603-609).  The change of Montgomery radix is one F_q product per coordinate on the device; everything else
 * (off-curve inputs -> O, identity outputs, products) is the wire-format path.  Replaces, for the batch calls of
 * integration/pbc_hip_glue.c, fp_to_bytes / fp_from_bytes on every coordinate. */
/* Diagnostic (tests): a schedule the wavefront kernels of type d interpret for this object (csrc/dw_sched.h: which level of
 * which program runs when, flattened by the host from the signed digits of r and the bits of Phi_6(q) / r -- the walk of
 * cc_miller_no_denom_affine, ecc/d_param.c:321-422, and lucas_even, :462-482).  which = 0: element_pairing; 1: the Miller
 * value of one term of element_prod_pairing; 2: the product with a term's value + the final exponentiation; 3:
 * pairing_pp_apply (d_pairing_pp_apply, :908-966, on the table of pairing_pp_init).  Returns the number of 64-bit entries
 * (0: the object is not a five-word type d pairing, or which is out of range), writes at most cap of them. */
size_t pbc_hip_diag_dw_schedule(pbc_hip_pairing_t *p, int which, uint64_t *out, size_t cap);
/* The same for the wavefront kernel of type f (csrc/fw_sched.h: cc_miller_no_denom, ecc/f_param.c:216-233, then f_tateexp, :250-283,
 * with the BN vector chain for the hard part; 0: the object is not a five-word BN type f pairing). */
size_t pbc_hip_diag_fw_schedule(pbc_hip_pairing_t *p, int which, uint64_t *out, size_t cap);    /* which: 0 pairing, 1 a term's Miller value, 2 product + final exponentiation */
/* ... and of type g on the five-word field (csrc/gw_sched.h: the loop of ecc/d_param.c:321-422 that ecc/g_param.c shares, cc_tatepower
 * and lucas_even over Phi_10(q) / r, g_param.c:471-558; pairing_pp_apply: g_param.c's copy of d_pairing_pp_apply); which as for
 * type d; 0: not a five-word type g pairing. */
size_t pbc_hip_diag_gw_schedule(pbc_hip_pairing_t *p, int which, uint64_t *out, size_t cap);
/* Diagnostic (tests): the table the wave kernels of type a1 / generic type a / type e read for this object (csrc/pairing_aw.cuh AG<N>,
 * host_params.h ag_aux_build): [0] = LEFF, the limbs q fills; from word 4 on five constants of L limbs each, c q in borrowed form for
 * (c, D) = (2, 1) (4, 2) (8, 4) (12, 2) (16, 2).  Returns the number of words (0: this q keeps the lane kernels -- the kernels that
 * stand for a1_pairing, ecc/a_param.c:1840-2015, and e_pairing, ecc/e_param.c:472-483, at every batch size). */
size_t pbc_hip_diag_ag_table(pbc_hip_pairing_t *p, uint32_t *out, size_t cap);
int pbc_hip_fq_limb_image_bytes(pbc_hip_pairing_t *p);          /* 8 t (0 on failure) */
int pbc_hip_element_pairing_batch_limbs(pbc_hip_pairing_t *p, uint8_t *gt, const uint8_t *g1, const uint8_t *g2, size_t n);
int pbc_hip_element_prod_pairing_batch_limbs(pbc_hip_pairing_t *p, uint8_t *gt, const uint8_t *g1, const uint8_t *g2, size_t n, int k);
int pbc_hip_element_prod_pairing_batch_limbs_dev(pbc_hip_pairing_t *p, void *d_gt, const void *d_g1, const void *d_g2, size_t n, int k, void *stream);

/* The group law, Z_r arithmetic and multi-exponentiations the reference's examples use around the pairing (round 5;
 * csrc/group_more.cuh, csrc/pbc_hip_group2.hip).  Host-buffer and _dev + stream forms as above; out may be exactly one of
 * the inputs.
 *   element_add / element_sub / element_neg (= element_invert) / element_double on G1, G2 (group 1, 2; include/pbc_field.h:
 *     element_add :261, element_sub :267, element_neg :287, element_double :293 -> curve_mul ecc/curve.c:153-207,
 *     curve_invert :79-100, curve_double :102-151; callers example/zss.c:49, example/hess.c:65,72): the affine law with
 *     the reference's case analysis (O neutral, equal points -> tangent, opposite points or a 2-torsion tangent -> O),
 *     one inversion per unit, any point of the curve.  O is read and written as zero bytes (off-curve records are O);
 *     the reference's curve_to_bytes ignores inf_flag and writes the element's stale coordinates for O (ecc/curve.c:603-609).
 *   pbc_hip_zr_op_batch: Z_r arithmetic on element_to_bytes records of Zr (big-endian, length_in_bytes_Zr, values < r;
 *     the reference runs its F_p back end on r: example/zss.c:40-41 adds and inverts, example/hess.c:63 multiplies).
 *     op 0 a*b, 1 a+b, 2 a-b, 3 1/a, 4 -a, 5 a/2, 6 2a, 7 a/b (b may be NULL for the unary ops).  1/0 is outside the
 *     reference's contract (mpz_invert fails); it gives 0 here.  pbc_hip_zr_from_hash_batch: element_from_hash on Zr
 *     (fp_from_hash, arith/montfp.c:440-448 over pbc_mpz_from_hash arith/field.c:643-668) for n digests of hlen bytes.
 *   element_pow2_zn / element_pow3_zn (include/pbc_field.h:496-531 -> arith/field.c:153-241) on G1, G2 (group 1, 2:
 *     [n1] a1 + [n2] a2 (+ [n3] a3)) and GT (group 3: a1^n1 a2^n2 (a3^n3)).  Default route on G1 / G2 of type a (512-bit
 *     field) and G1 of the five-word type d / f fields: ONE limb-form signed-window ladder for all bases -- a table of odd
 *     multiples per base, four doublings and k additions per window (round 6: 1.45-1.6 times the composition); the lanes
 *     it cannot finish (equal or opposite bases meeting in the accumulator, a base off the curve, a zero scalar, the
 *     result O) take the complete multi-scalar kernel.  Elsewhere (and with "hip_multi_compose 1"): the composition of the
 *     tuned single-base ladders (element_mul_zn / element_pow_zn per base, then the additions / products).
 *     "hip_group_slow 1" in the parameter text: Shamir's trick on the word-form arithmetic -- a per-lane table of the
 *     subset sums, ONE doubling (squaring) per scalar bit for all bases.  All routes are complete (any point of the curve,
 *     any scalar of length_in_bytes_Zr bytes) and give the same bytes; out may overlap any base.
 *   The all-zero record on types a / a1 (curve y^2 = x^3 + x): the group law above reads and writes it as O, although
 *     (0, 0) is a finite point of order two there and pbc_hip_element_snprint prints the record as "[0, 0]" (what the
 *     reference prints for that point).  P + (-P) therefore prints as "[0, 0]", not "O", and adding the genuine
 *     2-torsion point is outside this library's group law (it lies outside G1: the cofactor is even). */
int pbc_hip_element_add_batch(pbc_hip_pairing_t *p, int group, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n);
int pbc_hip_element_sub_batch(pbc_hip_pairing_t *p, int group, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n);
int pbc_hip_element_neg_batch(pbc_hip_pairing_t *p, int group, uint8_t *out, const uint8_t *a, size_t n);
int pbc_hip_element_double_batch(pbc_hip_pairing_t *p, int group, uint8_t *out, const uint8_t *a, size_t n);
int pbc_hip_element_add_batch_dev(pbc_hip_pairing_t *p, int group, void *d_out, const void *d_a, const void *d_b, size_t n, void *stream);
int pbc_hip_element_sub_batch_dev(pbc_hip_pairing_t *p, int group, void *d_out, const void *d_a, const void *d_b, size_t n, void *stream);
int pbc_hip_element_neg_batch_dev(pbc_hip_pairing_t *p, int group, void *d_out, const void *d_a, size_t n, void *stream);
int pbc_hip_element_double_batch_dev(pbc_hip_pairing_t *p, int group, void *d_out, const void *d_a, size_t n, void *stream);
int pbc_hip_zr_op_batch(pbc_hip_pairing_t *p, int op, uint8_t *out, const uint8_t *a, const uint8_t *b, size_t n);
int pbc_hip_zr_op_batch_dev(pbc_hip_pairing_t *p, int op, void *d_out, const void *d_a, const void *d_b, size_t n, void *stream);
int pbc_hip_zr_from_hash_batch(pbc_hip_pairing_t *p, uint8_t *out, const uint8_t *data, int hlen, size_t n);
int pbc_hip_zr_from_hash_batch_dev(pbc_hip_pairing_t *p, void *d_out, const void *d_data, int hlen, size_t n, void *stream);
int pbc_hip_element_pow2_zn_batch(pbc_hip_pairing_t *p, int group, uint8_t *out, const uint8_t *a1, const uint8_t *n1,
                                  const uint8_t *a2, const uint8_t *n2, size_t n);
int pbc_hip_element_pow3_zn_batch(pbc_hip_pairing_t *p, int group, uint8_t *out, const uint8_t *a1, const uint8_t *n1,
                                  const uint8_t *a2, const uint8_t *n2, const uint8_t *a3, const uint8_t *n3, size_t n);
int pbc_hip_element_pow2_zn_batch_dev(pbc_hip_pairing_t *p, int group, void *d_out, const void *d_a1, const void *d_n1,
                                      const void *d_a2, const void *d_n2, size_t n, void *stream);
int pbc_hip_element_pow3_zn_batch_dev(pbc_hip_pairing_t *p, int group, void *d_out, const void *d_a1, const void *d_n1,
                                      const void *d_a2, const void *d_n2, const void *d_a3, const void *d_n3, size_t n, void *stream);

/* Text formats on element_to_bytes records -- host-side string handling, no device involved, usable
 * without libpbc.  group: 0 = Zr, 1 = G1, 2 = G2, 3 = GT.
 *   pbc_hip_element_snprint   what element_snprint / element_printf("%B") print for the element the record deserialises
 *                             to (include/pbc_field.h:175; fp_snprint arith/montfp.c:177-185: decimal; fq_snprint
 *                             arith/fieldquadratic.c:105-130: "[x, y]"; polymod_snprint arith/poly.c:1221-1250:
 *                             "[c0, ..., c(n-1)]"; curve_snprint ecc/curve.c:501-533: "[x, y]" or "O"; GT prints its
 *                             underlying field element, ecc/pairing.c:187-190).  snprintf semantics: at most n - 1
 *                             characters and a NUL are stored, the full length is returned (-1 on error).  A record
 *                             of a curve over F_q that is not on the curve prints as "O", as curve_from_bytes
 *                             (ecc/curve.c:609-623) would make it; on the twists (G2 of types d, f, g) the curve
 *                             equation is checked on the device (one [1] R through element_mul_zn) -- with no usable
 *                             device only the all-zero record prints as "O".  Types a, a1: the all-zero record is
 *                             the point (0, 0) of y^2 = x^3 + x and prints as "[0, 0]"; "O" parses to that record,
 *                             which every batch entry point treats as O (see "zero-filled records" above), so O does
 *                             not round-trip through text there.
 *   pbc_hip_element_set_str   element_set_str (fp_set_str :187-195 with pbc_mpz_set_str arith/field.c:725-755: base 2..36,
 *                             0 = 10, blanks skipped, reduced mod q; fq_set_str :145-157; polymod_set_str :1269-1283;
 *                             curve_set_str ecc/curve.c:555-578: "O" or "[x, y]", off-curve points become O and return 0
 *                             -- on the twists through the same device check as above): writes the record, returns the
 *                             characters consumed (0: syntax error).
 *   pbc_hip_param_snprint     pbc_param_out_str (include/pbc_param.h:38; a_out_str ecc/a_param.c:36-46 and its
 *                             siblings): "type t" and the keys of the type in the reference's order, integers in decimal. */
int pbc_hip_element_snprint(const pbc_hip_pairing_t *p, int group, char *s, size_t n, const uint8_t *rec);
int pbc_hip_element_set_str(const pbc_hip_pairing_t *p, int group, uint8_t *rec, const char *s, int base);
int pbc_hip_param_snprint(const pbc_hip_pairing_t *p, char *s, size_t n);

/* Batched base-field operations on canonical bytes: the arith/montfp.c semantics the
 * kernels are built on (mont_mul :334-377, fp_add/sub/double/halve/neg :220-330,
 * fp_invert :401-422), exposed so they can be checked differentially the way
 * guru/checkfp.c does.  op: 0 mul, 1 add, 2 sub, 3 invert, 4 neg, 5 halve, 6 double.
 * Host buffers of n*len_Fq bytes each (b may be NULL for unary ops). */
int pbc_hip_fq_op_batch(pbc_hip_pairing_t *p, int op, uint8_t *c, const uint8_t *a,
                        const uint8_t *b, size_t n);
int pbc_hip_length_in_bytes_Fq(const pbc_hip_pairing_t *p);

/* Register-only integer multiply-add micro-benchmark (the measured int-MAC roofline,
 * SURVEY.md 8d): runs `iters` dependent-chain-free v_mad_u64_u32 per lane on the whole
 * chip and returns MAC/s.  variant selects the instruction mix (see csrc/pbc_hip.hip). */
int pbc_hip_int_mac_peak(int variant, int iters, double *mac_per_s, double *ms);

/* 512-bit Montgomery multiplier micro-benchmark: `iters` dependent products per lane with
 * `waves_per_simd` resident waves; returns F_q products per second.  variant 0: saturated
 * 32-bit limbs (mad+addc), 1/2: unsaturated 29-bit limbs (one/two accumulators), 3/4: the
 * dedicated squaring. */
int pbc_hip_diag_mul_bench(int variant, int iters, int waves_per_simd, double *mul_per_s, double *ms);

/* Diagnostics for bring-up: stage 0 copies the object's derived constant block (as uploaded to
 * __constant__ memory) into out; stage 1 (type F) returns the Miller values before the final
 * exponentiation for n input pairs. */
int pbc_hip_diag_stage(pbc_hip_pairing_t *p, int stage, uint8_t *out, size_t out_len,
                       const uint8_t *g1, const uint8_t *g2, size_t n);

/* Algorithmic work model used for the roofline: reference F_q multiplications
 * per unit x (2N^2+N) 32-bit MACs.  k >= 1: a k-term product (k = 1: one pairing); k = -1: one
 * pairing_pp_apply (the reference's pp algorithm: no arithmetic on the first argument's curve). */
double pbc_hip_algorithmic_macs_per_unit(const pbc_hip_pairing_t *p, int k);

const char *pbc_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
