/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, link, load or call anything under oracle/.
 *
 * fq.h : CPU restatement of the ristretto255 scalar field F_q used by the
 * reference prover, q = 2^252 + 27742317777372353535851937790883648493.
 * Follows Spartan/src/scalar/ristretto255.rs: 4 x u64 little-endian limbs,
 * always in Montgomery form with R = 2^256 (ristretto255.rs:199-200).
 *
 * Pinning: the reference's own known-answer tests (ristretto255.rs:789-1213)
 * are transcribed as data in tests/golden/fq_kat.json and checked by
 * tests/test_oracle_fq.py together with a Python big-integer model.
 */
#ifndef VPIN_ORACLE_FQ_H
#define VPIN_ORACLE_FQ_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } fq_t; /* Montgomery form, R = 2^256 */

extern const fq_t FQ_MODULUS; /* ristretto255.rs:249-254 (raw, not Montgomery) */
extern const fq_t FQ_R;       /* ristretto255.rs:301-306 : one()  */
extern const fq_t FQ_R2;      /* ristretto255.rs:309-314 */
extern const fq_t FQ_R3;      /* ristretto255.rs:317-322 */
#define FQ_INV 0xd2b51da312547e1bULL /* ristretto255.rs:298 */

fq_t fq_zero(void);
fq_t fq_one(void);
fq_t fq_add(const fq_t *a, const fq_t *b);  /* ristretto255.rs:746-757 */
fq_t fq_sub(const fq_t *a, const fq_t *b);  /* ristretto255.rs:729-743 */
fq_t fq_neg(const fq_t *a);                 /* ristretto255.rs:760-775 */
fq_t fq_mul(const fq_t *a, const fq_t *b);  /* ristretto255.rs:701-726 */
fq_t fq_square(const fq_t *a);              /* ristretto255.rs:482-511 */
fq_t fq_montgomery_reduce(const uint64_t r[8]); /* ristretto255.rs:653-698 */
fq_t fq_from_u64(uint64_t v);               /* ristretto255.rs:214-218 */
fq_t fq_from_raw(const uint64_t v[4]);      /* ristretto255.rs:476-478 */
int  fq_from_bytes(fq_t *out, const uint8_t b[32]); /* 1 = canonical; :398-422 */
void fq_to_bytes(uint8_t out[32], const fq_t *a);   /* ristretto255.rs:426-438 */
fq_t fq_from_bytes_wide(const uint8_t b[64]);       /* ristretto255.rs:442-473 */
fq_t fq_from_bytes_mod_order(const uint8_t b[32]);  /* dalek Scalar::from_bytes_mod_order */
fq_t fq_pow_vartime(const fq_t *a, const uint64_t by[4]); /* :531-545 */
fq_t fq_invert(const fq_t *a);              /* ristretto255.rs:548-602; 0 -> 0 */
fq_t fq_batch_invert(fq_t *inputs, size_t n); /* ristretto255.rs:604-651 */
int  fq_eq(const fq_t *a, const fq_t *b);
int  fq_is_zero(const fq_t *a);

#ifdef __cplusplus
}
#endif
#endif
