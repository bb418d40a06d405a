/*
 * pbc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's pairing hot path,
 * used exclusively as the parity checker by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.  The product path (pbc_amd/csrc, libpbc_hip.so)
 * never includes, links or calls anything in this directory.
 *
 * Parity status: PINNED -- checked against the reference's own Type-A known
 * answer test (pbc/pairing_test.pbc:3-10) and against outputs of the unmodified
 * reference compiled here (oracle/_ref/ref_tool gen -> tests/golden/ *.vec).
 */
#ifndef PBC_ORACLE_H
#define PBC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_pairing oracle_pairing;

/* Parse a PBC .param text (same grammar as ecc/param.c:100-170: "key value" lines)
 * and build the pairing constants.  Returns 0 on success (pairing_init_set_buf
 * convention, ecc/pairing.c:88-98). */
int oracle_pairing_init(oracle_pairing **out, const char *param_text, size_t len);
void oracle_pairing_clear(oracle_pairing *p);

int oracle_type(const oracle_pairing *p);        /* 'a', 'd', 'f' */
int oracle_len_G1(const oracle_pairing *p);      /* pairing_length_in_bytes_G1 */
int oracle_len_G2(const oracle_pairing *p);
int oracle_len_GT(const oracle_pairing *p);

/* element_from_bytes(G1), element_from_bytes(G2), element_pairing, element_to_bytes(GT)
 * on n independent pairs (AoS, element_to_bytes layout). */
int oracle_pairing_batch(const oracle_pairing *p, const uint8_t *g1, const uint8_t *g2,
                         uint8_t *gt, size_t n);
/* element_prod_pairing over n products of k terms each. */
int oracle_prod_pairing_batch(const oracle_pairing *p, const uint8_t *g1, const uint8_t *g2,
                              uint8_t *gt, size_t n, int k);

/* Fq micro-oracle (arith/montfp.c semantics on canonical big-endian bytes):
 * op 0: a*b  1: a+b  2: a-b  3: 1/a  4: -a  5: a/2  6: 2a */
int oracle_fq_op(const oracle_pairing *p, int op, const uint8_t *a, const uint8_t *b,
                 uint8_t *c, size_t n);
/* GT helpers for property tests: out = a*b ; out = a^e (e big-endian bytes, elen long) */
int oracle_gt_mul(const oracle_pairing *p, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
int oracle_gt_pow(const oracle_pairing *p, const uint8_t *a, const uint8_t *e, size_t elen,
                  uint8_t *out, size_t n);
/* G1/G2 scalar multiplication on wire bytes (group: 1 or 2), for bilinearity tests. */
/* element_from_hash on G1 (G2 of the symmetric types): curve_from_hash, ecc/curve.c:455-482 */
int oracle_from_hash(const oracle_pairing *p, const uint8_t *data, int hlen, uint8_t *out, size_t n);
/* G1 point formats (ecc/curve.c:762-836): what 0 to_bytes_compressed, 1 from_bytes_compressed, 2 to_bytes_x_only,
 * 3 from_bytes_x_only */
int oracle_point_format(const oracle_pairing *p, int what, const uint8_t *in, uint8_t *out, size_t n);
/* the same on the G2 twists of types d, g, f (element_from_hash without a cofactor; compressed and x-only points) */
int oracle_from_hash_g2(const oracle_pairing *p, const uint8_t *data, int hlen, uint8_t *out, size_t n);
int oracle_point_format_g2(const oracle_pairing *p, int what, const uint8_t *in, uint8_t *out, size_t n);
int oracle_g_mul(const oracle_pairing *p, int group, const uint8_t *pt, const uint8_t *e,
                 size_t elen, uint8_t *out, size_t n);

/* pairing->finalpow on GT-format records (the final exponentiation alone) */
int oracle_finalpow(const oracle_pairing *p, const uint8_t *in, uint8_t *out, size_t n);

/* round 5: the group law on G1 (op 0 a+b, 1 a-b, 2 -a, 3 2a; O = zero record), Z_r arithmetic (op 0 mul, 1 add, 2 sub,
 * 3 invert, 4 neg, 5 halve, 6 double, 7 div, 8 from_hash of hlen-byte digests), element_pow2_zn / element_pow3_zn on G1
 * (group 1) and GT (group 3) with the k records / scalars of a unit side by side */
int oracle_g1_op(const oracle_pairing *p, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
int oracle_zr_op(const oracle_pairing *p, int op, const uint8_t *a, const uint8_t *b, int hlen, uint8_t *out, size_t n);
int oracle_pow_multi(const oracle_pairing *p, int group, int k, const uint8_t *a, const uint8_t *e, size_t elen, uint8_t *out, size_t n);

/* counts of Fq multiplications / inversions since last reset (for the work model) */
void oracle_counters(uint64_t *mul, uint64_t *inv, int reset);

#ifdef __cplusplus
}
#endif
#endif
