/*
 * pbc_hip_glue.h -- the reference-side binding: what a PBC maintainer adds to route the
 * pairing hot path through libpbc_hip.so (include/pbc_hip.h).  Compiles against the
 * UNMODIFIED PBC headers; nothing in PBC itself changes.
 *
 *   pairing_t pairing;  pairing_init_set_buf(pairing, text, len);      // stock PBC
 *   pbc_hip_attach(pairing, text, len);                                // + this line
 *   element_pairing(out, in1, in2);              // unchanged call sites now run on the GPU
 *   element_prod_pairing(out, in1, in2, n);      //   (pairing->map / pairing->prod_pairings,
 *                                                //    include/pbc_pairing.h:27-30)
 *   element_pairing_batch(out, in1, in2, n);     // new: n pairings, one launch
 *   element_prod_pairing_batch(out, in1, in2, n, k);
 */
#ifndef PBC_HIP_GLUE_H
#define PBC_HIP_GLUE_H
#include <pbc.h>

/* Load libpbc_hip.so (path: $PBC_HIP_LIB, else the default search path), build the GPU
 * pairing object from the same parameter text, and install GPU-backed map / prod_pairings
 * function pointers in `pairing` (the same seam pairing_option_set uses, ecc/a_param.c:1399-1418).
 * Returns 0 on success, 1 on failure (pairing_init convention); on failure `pairing` is untouched. */
int pbc_hip_attach(pairing_t pairing, const char *param, size_t len);
/* Restore the CPU function pointers and free the GPU object. */
void pbc_hip_detach(pairing_t pairing);

/* out[i] = e(in1[i], in2[i]) for i < n -- element_pairing semantics per item, including the
 * identity short-circuit of pairing_apply (include/pbc_pairing.h:118-135). */
int element_pairing_batch(element_t out[], element_t in1[], element_t in2[], size_t n);
/* out[u] = prod_{j<k} e(in1[u*k+j], in2[u*k+j]) -- element_prod_pairing semantics per product
 * (any identity input => 1, include/pbc_pairing.h:153-171). */
int element_prod_pairing_batch(element_t out[], element_t in1[], element_t in2[], size_t n, int k);
/* out[i] = e(P, in2[i]) for the P a pairing_pp_t was initialised with (pairing_pp_init after
 * pbc_hip_attach): pairing_pp_apply (include/pbc_pairing.h:79-89) over a batch.  Type A. */
int pairing_pp_apply_batch(element_t out[], element_t in2[], size_t n, pairing_pp_t p);
/* out[i] = in[i]^zr[i]: element_pow_zn / element_mul_zn (include/pbc_field.h:311, :374) over a
 * batch of G1 elements (G2 too for symmetric pairings) or GT elements. */
int element_pow_zn_batch(element_t out[], element_t in[], element_t zr[], size_t n);
/* out[i] = element_from_hash of the i-th hlen-byte digest (include/pbc_field.h:257) for elements of G1 or G2
 * of any type. */
int element_from_hash_batch(element_t out[], const void *data, int hlen, size_t n);
#endif
