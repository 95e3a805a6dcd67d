/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see fq.h).
 * group.h : CPU restatement of the ristretto255 group the reference reaches through
 * Spartan/src/group.rs (type aliases onto curve25519-dalek 3.2.0, which is NOT vendored
 * under /root/reference: Cargo.lock pins it).  Restated from the published algorithm
 * (RFC 9496 "The ristretto255 and decaf448 Groups", sections 4.1-4.3; Edwards25519 extended
 * coordinates per Hisil-Wong-Carter-Dawson 2008) and anchored on the reference call
 * sites: group.rs:6-8,26-27,103-122; commitments.rs:20-38,85-98.
 *
 * Pinning: RFC 9496 Appendix A vectors (multiples of the generator, hash-to-group) in
 * tests/golden/ristretto_kat.json, plus a Python big-int model (tests/pymodel_group.py).
 */
#ifndef VPIN_ORACLE_GROUP_H
#define VPIN_ORACLE_GROUP_H
#include "fq.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[5]; } fe_t; /* GF(2^255-19), 5 x 51-bit limbs */

typedef struct { fe_t X, Y, Z, T; } ge_t; /* extended twisted Edwards, a=-1 */

/* field */
void fe_from_bytes(fe_t *r, const uint8_t b[32]); /* ignores bit 255 */
void fe_to_bytes(uint8_t b[32], const fe_t *a);   /* canonical */

/* group */
void ge_identity(ge_t *r);
void ge_basepoint(ge_t *r);
void ge_add(ge_t *r, const ge_t *p, const ge_t *q);
void ge_sub(ge_t *r, const ge_t *p, const ge_t *q);
void ge_double(ge_t *r, const ge_t *p);
void ge_neg(ge_t *r, const ge_t *p);
int  ge_eq(const ge_t *p, const ge_t *q); /* ristretto equality */
/* RistrettoPoint::compress / CompressedRistretto::decompress (RFC 9496 4.3.2 / 4.3.1) */
void ge_compress(uint8_t out[32], const ge_t *p);
int  ge_decompress(ge_t *r, const uint8_t in[32]); /* 1 on success */
/* RistrettoPoint::from_uniform_bytes (dalek 3.2.0 = RFC 9496 4.3.4 one-way map) */
void ge_from_uniform_bytes(ge_t *r, const uint8_t in[64]);
/* 128-byte affine-free transport form used by the C ABI: X|Y|Z|T canonical LE */
void ge_to_xyzt(uint8_t out[128], const ge_t *p);
void ge_from_xyzt(ge_t *r, const uint8_t in[128]);

/* scalar * point, scalar given as 32-byte canonical little-endian integer */
void ge_scalarmul_bytes(ge_t *r, const uint8_t s[32], const ge_t *p);
/* &Scalar * &GroupElement (group.rs:39-55): Montgomery scalar -> to_bytes -> mul */
void ge_scalarmul(ge_t *r, const fq_t *s, const ge_t *p);
/* GroupElement::vartime_multiscalar_mul (group.rs:103-122): sum s_i * P_i.
 * Pippenger with signed radix-2^w digits and zero-digit skipping (dalek's vartime
 * strategy); any correct evaluation order gives the same group element. */
void ge_msm(ge_t *r, const fq_t *scalars, const ge_t *points, size_t n);

/* MultiCommitGens::new (commitments.rs:20-38): SHAKE256(label || B_compressed) ->
 * (n+1) x 64 bytes -> from_uniform_bytes.  gens has n+1 entries: G[0..n) then h. */
void oracle_gens_new(ge_t *gens, size_t n, const uint8_t *label, size_t label_len);
/* Commitments for [Scalar] / Scalar (commitments.rs:85-98): MSM(v, G) + blind * h */
void oracle_commit(ge_t *r, const fq_t *v, size_t n, const fq_t *blind, const ge_t *G, const ge_t *h);

/* DensePolynomial::commit_inner (dense_mlpoly.rs:160-175), rows in parallel (OpenMP
 * mirrors rayon's into_par_iter).  out = L x 32 B compressed. */
void oracle_hyrax_commit(uint8_t *out, const fq_t *Z, size_t L_size, size_t R_size,
                         const fq_t *blinds, const ge_t *G, const ge_t *h, int threads);

#ifdef __cplusplus
}
#endif
#endif
