/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see fq.h).
 * poly.h : CPU restatement of the dense multilinear-polynomial table ops and
 * the per-round sum-check reductions of the reference prover.
 *   Spartan/src/dense_mlpoly.rs  (EqPolynomial, DensePolynomial)
 *   Spartan/src/sumcheck.rs      (round evaluation loops)
 *   Spartan/src/unipoly.rs       (interpolation)
 * Tables are flat arrays of fq_t (32-byte Montgomery elements), exactly the
 * reference's Vec<Scalar> memory image.
 */
#ifndef VPIN_ORACLE_POLY_H
#define VPIN_ORACLE_POLY_H
#include "fq.h"

#ifdef __cplusplus
extern "C" {
#endif

/* EqPolynomial::evals, dense_mlpoly.rs:78-94. out has 2^ell entries. */
void oracle_eq_evals(const fq_t *r, int ell, fq_t *out);

/* DensePolynomial::bound_poly_var_top, dense_mlpoly.rs:229-236.
 * In place; caller halves its own length afterwards. */
void oracle_bound_poly_var_top(fq_t *Z, size_t len, const fq_t *r);

/* One round of prove_cubic_with_additive_term, sumcheck.rs:624-652, with the
 * phase-1 combiner A*(B*C - D) (r1csproof.rs:104-108). len = current table
 * length (pairs are (i, i+len/2)). out = {e0, e2, e3}. */
void oracle_sc_cubic_round(const fq_t *A, const fq_t *B, const fq_t *C, const fq_t *D,
                           size_t len, fq_t out[3]);

/* One round of prove_quad, sumcheck.rs:460-469, combiner A*B
 * (r1csproof.rs:139-140). out = {e0, e2}. */
void oracle_sc_quad_round(const fq_t *A, const fq_t *B, size_t len, fq_t out[2]);

/* DensePolynomial::bound, dense_mlpoly.rs:220-227: LZ[i] = sum_j L[j]*Z[j*R+i]. */
void oracle_poly_bound(const fq_t *Z, const fq_t *L, size_t L_size, size_t R_size, fq_t *LZ);

/* DotProductProofLog::compute_dotproduct, nizk/mod.rs:442-445. */
fq_t oracle_dotproduct(const fq_t *a, const fq_t *b, size_t n);

/* DensePolynomial::evaluate, dense_mlpoly.rs:249-255. */
fq_t oracle_poly_evaluate(const fq_t *Z, const fq_t *r, int ell);

/* UniPoly::from_evals, unipoly.rs:23-54. n_evals = 3 or 4; coeffs gets n_evals
 * entries, constant term first. */
void oracle_unipoly_from_evals(const fq_t *evals, int n_evals, fq_t *coeffs);
/* UniPoly::evaluate, unipoly.rs:72-80. */
fq_t oracle_unipoly_evaluate(const fq_t *coeffs, int n, const fq_t *r);

#ifdef __cplusplus
}
#endif
#endif
