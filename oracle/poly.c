/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see fq.h / poly.h).
 */
#include "poly.h"
#include <stdlib.h>

void oracle_eq_evals(const fq_t *r, int ell, fq_t *evals) {
  /* dense_mlpoly.rs:78-94: in-place doubling, highest index first */
  size_t n = (size_t)1 << ell;
  for (size_t i = 0; i < n; i++) evals[i] = fq_one();
  size_t size = 1;
  for (int j = 0; j < ell; j++) {
    size *= 2;
    for (size_t i = size - 1;; i -= 2) { /* i = size-1, size-3, ..., 1 */
      fq_t scalar = evals[i / 2];
      evals[i] = fq_mul(&scalar, &r[j]);
      evals[i - 1] = fq_sub(&scalar, &evals[i]);
      if (i == 1) break;
    }
  }
}

void oracle_bound_poly_var_top(fq_t *Z, size_t len, const fq_t *r) {
  size_t n = len / 2;
  for (size_t i = 0; i < n; i++) {
    fq_t d = fq_sub(&Z[i + n], &Z[i]);
    fq_t t = fq_mul(r, &d);
    Z[i] = fq_add(&Z[i], &t);
  }
}

static inline fq_t comb_cubic(const fq_t *a, const fq_t *b, const fq_t *c, const fq_t *d) {
  /* r1csproof.rs:104-108 : A * (B * C - D) */
  fq_t bc = fq_mul(b, c);
  fq_t t = fq_sub(&bc, d);
  return fq_mul(a, &t);
}

void oracle_sc_cubic_round(const fq_t *A, const fq_t *B, const fq_t *C, const fq_t *D,
                           size_t plen, fq_t out[3]) {
  fq_t e0 = fq_zero(), e2 = fq_zero(), e3 = fq_zero();
  size_t len = plen / 2;
  for (size_t i = 0; i < len; i++) {
    /* eval 0 */
    fq_t t = comb_cubic(&A[i], &B[i], &C[i], &D[i]);
    e0 = fq_add(&e0, &t);
    /* eval 2: -low + 2*high */
    fq_t a2 = fq_add(&A[len + i], &A[len + i]); a2 = fq_sub(&a2, &A[i]);
    fq_t b2 = fq_add(&B[len + i], &B[len + i]); b2 = fq_sub(&b2, &B[i]);
    fq_t c2 = fq_add(&C[len + i], &C[len + i]); c2 = fq_sub(&c2, &C[i]);
    fq_t d2 = fq_add(&D[len + i], &D[len + i]); d2 = fq_sub(&d2, &D[i]);
    t = comb_cubic(&a2, &b2, &c2, &d2);
    e2 = fq_add(&e2, &t);
    /* eval 3: eval2 point + high - low */
    fq_t a3 = fq_add(&a2, &A[len + i]); a3 = fq_sub(&a3, &A[i]);
    fq_t b3 = fq_add(&b2, &B[len + i]); b3 = fq_sub(&b3, &B[i]);
    fq_t c3 = fq_add(&c2, &C[len + i]); c3 = fq_sub(&c3, &C[i]);
    fq_t d3 = fq_add(&d2, &D[len + i]); d3 = fq_sub(&d3, &D[i]);
    t = comb_cubic(&a3, &b3, &c3, &d3);
    e3 = fq_add(&e3, &t);
  }
  out[0] = e0; out[1] = e2; out[2] = e3;
}

void oracle_sc_quad_round(const fq_t *A, const fq_t *B, size_t plen, fq_t out[2]) {
  fq_t e0 = fq_zero(), e2 = fq_zero();
  size_t len = plen / 2;
  for (size_t i = 0; i < len; i++) {
    fq_t t = fq_mul(&A[i], &B[i]);
    e0 = fq_add(&e0, &t);
    fq_t a2 = fq_add(&A[len + i], &A[len + i]); a2 = fq_sub(&a2, &A[i]);
    fq_t b2 = fq_add(&B[len + i], &B[len + i]); b2 = fq_sub(&b2, &B[i]);
    t = fq_mul(&a2, &b2);
    e2 = fq_add(&e2, &t);
  }
  out[0] = e0; out[1] = e2;
}

void oracle_poly_bound(const fq_t *Z, const fq_t *L, size_t L_size, size_t R_size, fq_t *LZ) {
  for (size_t i = 0; i < R_size; i++) {
    fq_t acc = fq_zero();
    for (size_t j = 0; j < L_size; j++) {
      fq_t t = fq_mul(&L[j], &Z[j * R_size + i]);
      acc = fq_add(&acc, &t);
    }
    LZ[i] = acc;
  }
}

fq_t oracle_dotproduct(const fq_t *a, const fq_t *b, size_t n) {
  fq_t acc = fq_zero();
  for (size_t i = 0; i < n; i++) {
    fq_t t = fq_mul(&a[i], &b[i]);
    acc = fq_add(&acc, &t);
  }
  return acc;
}

fq_t oracle_poly_evaluate(const fq_t *Z, const fq_t *r, int ell) {
  size_t n = (size_t)1 << ell;
  fq_t *chis = (fq_t *)malloc(sizeof(fq_t) * n);
  oracle_eq_evals(r, ell, chis);
  fq_t v = oracle_dotproduct(Z, chis, n);
  free(chis);
  return v;
}

void oracle_unipoly_from_evals(const fq_t *e, int n_evals, fq_t *coeffs) {
  fq_t two = fq_from_u64(2), six = fq_from_u64(6);
  fq_t two_inv = fq_invert(&two), six_inv = fq_invert(&six);
  if (n_evals == 3) {
    /* unipoly.rs:26-33 */
    fq_t c = e[0];
    fq_t t = fq_sub(&e[2], &e[1]); t = fq_sub(&t, &e[1]); t = fq_add(&t, &c);
    fq_t a = fq_mul(&two_inv, &t);
    fq_t b = fq_sub(&e[1], &c); b = fq_sub(&b, &a);
    coeffs[0] = c; coeffs[1] = b; coeffs[2] = a;
  } else {
    /* unipoly.rs:35-50 */
    fq_t d = e[0];
    fq_t t = fq_sub(&e[3], &e[2]); t = fq_sub(&t, &e[2]); t = fq_sub(&t, &e[2]);
    t = fq_add(&t, &e[1]); t = fq_add(&t, &e[1]); t = fq_add(&t, &e[1]); t = fq_sub(&t, &e[0]);
    fq_t a = fq_mul(&six_inv, &t);
    fq_t u = fq_add(&e[0], &e[0]);
    for (int k = 0; k < 5; k++) u = fq_sub(&u, &e[1]);
    for (int k = 0; k < 4; k++) u = fq_add(&u, &e[2]);
    u = fq_sub(&u, &e[3]);
    fq_t b = fq_mul(&two_inv, &u);
    fq_t c = fq_sub(&e[1], &d); c = fq_sub(&c, &a); c = fq_sub(&c, &b);
    coeffs[0] = d; coeffs[1] = c; coeffs[2] = b; coeffs[3] = a;
  }
}

fq_t oracle_unipoly_evaluate(const fq_t *coeffs, int n, const fq_t *r) {
  fq_t eval = coeffs[0], power = *r;
  for (int i = 1; i < n; i++) {
    fq_t t = fq_mul(&power, &coeffs[i]);
    eval = fq_add(&eval, &t);
    power = fq_mul(&power, r);
  }
  return eval;
}
