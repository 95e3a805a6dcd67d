/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see fq.h / sat.h).
 * proto_common.h : protocol pieces shared by sat.c (R1CSProof) and spark.c (R1CSEvalProof):
 * bincode reader/writer, Pedersen commit helpers, generator views, the Sigma protocols,
 * the ZK sum-check and DotProductProofLog.  Everything is `static` on purpose: each .c file
 * gets a private copy and no generic name (compress, w_u64 ...) leaks out of liboracle.so.
 */
#ifndef VPIN_ORACLE_PROTO_COMMON_H
#define VPIN_ORACLE_PROTO_COMMON_H
#include "sat.h"
#include <stdlib.h>
#include <string.h>
#if defined(__GNUC__)
#pragma GCC diagnostic ignored "-Wunused-function"
#endif

/* ------------------------------------------------------------------ small helpers */

static size_t log2z(size_t n) { size_t l = 0; while (((size_t)1 << l) < n) l++; return l; }

static fq_t F_add(fq_t a, fq_t b) { return fq_add(&a, &b); }
static fq_t F_sub(fq_t a, fq_t b) { return fq_sub(&a, &b); }
static fq_t F_mul(fq_t a, fq_t b) { return fq_mul(&a, &b); }

typedef struct { size_t n; const ge_t *G; ge_t h; } mcg_t; /* MultiCommitGens view */

typedef struct { uint8_t b[32]; } cg_t; /* CompressedGroup */

static cg_t compress(const ge_t *p) { cg_t c; ge_compress(c.b, p); return c; }

/* Commitments for Scalar (commitments.rs:85-90) */
static ge_t commit_scalar(const fq_t *x, const fq_t *blind, const mcg_t *g1) {
  ge_t a, b, r;
  ge_scalarmul(&a, x, &g1->G[0]);
  ge_scalarmul(&b, blind, &g1->h);
  ge_add(&r, &a, &b);
  return r;
}
/* Commitments for [Scalar] (commitments.rs:93-98) */
static ge_t commit_vec(const fq_t *v, size_t n, const fq_t *blind, const mcg_t *gn) {
  ge_t r;
  oracle_commit(&r, v, n, blind, gn->G, &gn->h);
  return r;
}

/* bincode writer / reader */
typedef struct { uint8_t *p; size_t len, cap; int bad; } wbuf;
static void w_bytes(wbuf *w, const void *src, size_t n) {
  if (w->len + n > w->cap) { w->bad = 1; return; }
  memcpy(w->p + w->len, src, n);
  w->len += n;
}
static void w_u64(wbuf *w, uint64_t v) { uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i)); w_bytes(w, b, 8); }
static void w_scalar(wbuf *w, const fq_t *s) { for (int i = 0; i < 4; i++) w_u64(w, s->l[i]); } /* Montgomery limbs */
static void w_point(wbuf *w, const cg_t *c) { w_bytes(w, c->b, 32); }

typedef struct { const uint8_t *p; size_t len, pos; int bad; } rbuf;
static void r_bytes(rbuf *r, void *dst, size_t n) {
  if (r->pos + n > r->len) { r->bad = 1; memset(dst, 0, n); return; }
  memcpy(dst, r->p + r->pos, n);
  r->pos += n;
}
static uint64_t r_u64(rbuf *r) { uint8_t b[8]; r_bytes(r, b, 8); uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | b[i]; return v; }
static fq_t r_scalar(rbuf *r) { fq_t s; for (int i = 0; i < 4; i++) s.l[i] = r_u64(r); return s; }
static cg_t r_point(rbuf *r) { cg_t c; r_bytes(r, c.b, 32); return c; }

/* ------------------------------------------------------------------ generators */

typedef struct {
  ge_t *g;      /* stream g[0..R+2) of MultiCommitGens::new under b"gens_r1cs_sat" */
  size_t R, L, ell;
  mcg_t gens_1, gens_3, gens_4; /* R1CSSumcheckGens (r1csproof.rs:49-74) */
  mcg_t pc_n, pc_1;             /* PolyCommitmentGens.gens.{gens_n,gens_1} (nizk/mod.rs:411-425) */
} satgens_t;

static void satgens_new(satgens_t *sg, size_t num_vars) {
  /* R1CSGens::new (r1csproof.rs:84-89) with label b"gens_r1cs_sat" (lib.rs:314) */
  size_t ell = log2z(num_vars), left = ell / 2, right = ell - left;
  sg->ell = ell; sg->L = (size_t)1 << left; sg->R = (size_t)1 << right;
  size_t need = sg->R + 2 < 5 ? 5 : sg->R + 2;
  sg->g = (ge_t *)malloc(sizeof(ge_t) * need);
  oracle_gens_new(sg->g, need - 1, (const uint8_t *)"gens_r1cs_sat", 13);
  /* DotProductProofGens::new(R): MultiCommitGens::new(R+1).split_at(R): h = g[R+1] for both */
  sg->pc_n.n = sg->R; sg->pc_n.G = sg->g; sg->pc_n.h = sg->g[sg->R + 1];
  sg->pc_1.n = 1; sg->pc_1.G = sg->g + sg->R; sg->pc_1.h = sg->g[sg->R + 1];
  sg->gens_1 = sg->pc_1; /* R1CSSumcheckGens::new clones gens_pc.gens.gens_1 */
  sg->gens_3.n = 3; sg->gens_3.G = sg->g; sg->gens_3.h = sg->g[3];
  sg->gens_4.n = 4; sg->gens_4.G = sg->g; sg->gens_4.h = sg->g[4];
}
static void satgens_free(satgens_t *sg) { free(sg->g); }

/* ------------------------------------------------------------------ Sigma protocols */

typedef struct { cg_t delta, beta; fq_t z[4]; int n; fq_t z_delta, z_beta; } dotproof_t;

/* DotProductProof::prove (nizk/mod.rs:315-374) */
static void dotproduct_prove(dotproof_t *pf, const mcg_t *g1, const mcg_t *gn, merlin_t *tr, merlin_t *tape,
                             const fq_t *x, const fq_t *blind_x, const fq_t *a, const fq_t *y, const fq_t *blind_y, int n) {
  tr_append_protocol_name(tr, "dot product proof");
  fq_t d[4];
  tr_challenge_vector(tape, "d_vec", d, (size_t)n);
  fq_t r_delta = tr_challenge_scalar(tape, "r_delta");
  fq_t r_beta = tr_challenge_scalar(tape, "r_beta");
  ge_t t = commit_vec(x, (size_t)n, blind_x, gn);
  cg_t Cx = compress(&t);
  tr_append_point(tr, "Cx", Cx.b);
  t = commit_scalar(y, blind_y, g1);
  cg_t Cy = compress(&t);
  tr_append_point(tr, "Cy", Cy.b);
  tr_append_scalars(tr, "a", a, (size_t)n);
  t = commit_vec(d, (size_t)n, &r_delta, gn);
  pf->delta = compress(&t);
  tr_append_point(tr, "delta", pf->delta.b);
  fq_t ad = oracle_dotproduct(a, d, (size_t)n);
  t = commit_scalar(&ad, &r_beta, g1);
  pf->beta = compress(&t);
  tr_append_point(tr, "beta", pf->beta.b);
  fq_t c = tr_challenge_scalar(tr, "c");
  pf->n = n;
  for (int i = 0; i < n; i++) pf->z[i] = F_add(F_mul(c, x[i]), d[i]);
  pf->z_delta = F_add(F_mul(c, *blind_x), r_delta);
  pf->z_beta = F_add(F_mul(c, *blind_y), r_beta);
}

/* DotProductProof::verify (nizk/mod.rs:376-409) */
static int dotproduct_verify(const dotproof_t *pf, const mcg_t *g1, const mcg_t *gn, merlin_t *tr,
                             const fq_t *a, const cg_t *Cx, const cg_t *Cy) {
  int n = pf->n;
  tr_append_protocol_name(tr, "dot product proof");
  tr_append_point(tr, "Cx", Cx->b);
  tr_append_point(tr, "Cy", Cy->b);
  tr_append_scalars(tr, "a", a, (size_t)n);
  tr_append_point(tr, "delta", pf->delta.b);
  tr_append_point(tr, "beta", pf->beta.b);
  fq_t c = tr_challenge_scalar(tr, "c");
  ge_t pCx, pCy, pd, pb, lhs, rhs, t;
  if (!ge_decompress(&pCx, Cx->b) || !ge_decompress(&pCy, Cy->b) || !ge_decompress(&pd, pf->delta.b) ||
      !ge_decompress(&pb, pf->beta.b)) return 0;
  ge_scalarmul(&t, &c, &pCx); ge_add(&lhs, &t, &pd);
  rhs = commit_vec(pf->z, (size_t)n, &pf->z_delta, gn);
  int ok = ge_eq(&lhs, &rhs);
  fq_t za = oracle_dotproduct(pf->z, a, (size_t)n);
  ge_scalarmul(&t, &c, &pCy); ge_add(&lhs, &t, &pb);
  rhs = commit_scalar(&za, &pf->z_beta, g1);
  ok &= ge_eq(&lhs, &rhs);
  return ok;
}

typedef struct { cg_t alpha; fq_t z1, z2; } knowproof_t;
/* KnowledgeProof::prove (nizk/mod.rs:27-53) */
static cg_t knowledge_prove(knowproof_t *pf, const mcg_t *g, merlin_t *tr, merlin_t *tape, const fq_t *x, const fq_t *r) {
  tr_append_protocol_name(tr, "knowledge proof");
  fq_t t1 = tr_challenge_scalar(tape, "t1"), t2 = tr_challenge_scalar(tape, "t2");
  ge_t p = commit_scalar(x, r, g);
  cg_t C = compress(&p);
  tr_append_point(tr, "C", C.b);
  p = commit_scalar(&t1, &t2, g);
  pf->alpha = compress(&p);
  tr_append_point(tr, "alpha", pf->alpha.b);
  fq_t c = tr_challenge_scalar(tr, "c");
  pf->z1 = F_add(F_mul(*x, c), t1);
  pf->z2 = F_add(F_mul(*r, c), t2);
  return C;
}
static int knowledge_verify(const knowproof_t *pf, const mcg_t *g, merlin_t *tr, const cg_t *C) {
  tr_append_protocol_name(tr, "knowledge proof");
  tr_append_point(tr, "C", C->b);
  tr_append_point(tr, "alpha", pf->alpha.b);
  fq_t c = tr_challenge_scalar(tr, "c");
  ge_t lhs = commit_scalar(&pf->z1, &pf->z2, g), pC, pa, t, rhs;
  if (!ge_decompress(&pC, C->b) || !ge_decompress(&pa, pf->alpha.b)) return 0;
  ge_scalarmul(&t, &c, &pC); ge_add(&rhs, &t, &pa);
  return ge_eq(&lhs, &rhs);
}

typedef struct { cg_t alpha; fq_t z; } eqproof_t;
/* EqualityProof::prove (nizk/mod.rs:89-118) */
static void equality_prove(eqproof_t *pf, const mcg_t *g, merlin_t *tr, merlin_t *tape,
                           const fq_t *v1, const fq_t *s1, const fq_t *v2, const fq_t *s2) {
  tr_append_protocol_name(tr, "equality proof");
  fq_t r = tr_challenge_scalar(tape, "r");
  ge_t p = commit_scalar(v1, s1, g);
  cg_t C1 = compress(&p);
  tr_append_point(tr, "C1", C1.b);
  p = commit_scalar(v2, s2, g);
  cg_t C2 = compress(&p);
  tr_append_point(tr, "C2", C2.b);
  ge_scalarmul(&p, &r, &g->h);
  pf->alpha = compress(&p);
  tr_append_point(tr, "alpha", pf->alpha.b);
  fq_t c = tr_challenge_scalar(tr, "c");
  pf->z = F_add(F_mul(c, F_sub(*s1, *s2)), r);
}
static int equality_verify(const eqproof_t *pf, const mcg_t *g, merlin_t *tr, const cg_t *C1, const cg_t *C2) {
  tr_append_protocol_name(tr, "equality proof");
  tr_append_point(tr, "C1", C1->b);
  tr_append_point(tr, "C2", C2->b);
  tr_append_point(tr, "alpha", pf->alpha.b);
  fq_t c = tr_challenge_scalar(tr, "c");
  ge_t p1, p2, pa, C, t, rhs, lhs;
  if (!ge_decompress(&p1, C1->b) || !ge_decompress(&p2, C2->b) || !ge_decompress(&pa, pf->alpha.b)) return 0;
  ge_sub(&C, &p1, &p2);
  ge_scalarmul(&t, &c, &C); ge_add(&rhs, &t, &pa);
  ge_scalarmul(&lhs, &pf->z, &g->h);
  return ge_eq(&lhs, &rhs);
}

typedef struct { cg_t alpha, beta, delta; fq_t z[5]; } prodproof_t;
/* ProductProof::prove (nizk/mod.rs:161-232) */
static void product_prove(prodproof_t *pf, const mcg_t *g, merlin_t *tr, merlin_t *tape,
                          const fq_t *x, const fq_t *rX, const fq_t *y, const fq_t *rY, const fq_t *z, const fq_t *rZ,
                          cg_t *X, cg_t *Y, cg_t *Z) {
  tr_append_protocol_name(tr, "product proof");
  fq_t b1 = tr_challenge_scalar(tape, "b1"), b2 = tr_challenge_scalar(tape, "b2"), b3 = tr_challenge_scalar(tape, "b3"),
       b4 = tr_challenge_scalar(tape, "b4"), b5 = tr_challenge_scalar(tape, "b5");
  ge_t p = commit_scalar(x, rX, g); *X = compress(&p); tr_append_point(tr, "X", X->b);
  p = commit_scalar(y, rY, g); *Y = compress(&p); tr_append_point(tr, "Y", Y->b);
  p = commit_scalar(z, rZ, g); *Z = compress(&p); tr_append_point(tr, "Z", Z->b);
  p = commit_scalar(&b1, &b2, g); pf->alpha = compress(&p); tr_append_point(tr, "alpha", pf->alpha.b);
  p = commit_scalar(&b3, &b4, g); pf->beta = compress(&p); tr_append_point(tr, "beta", pf->beta.b);
  ge_t Xp;
  ge_decompress(&Xp, X->b);
  mcg_t gX = {1, &Xp, g->h};
  p = commit_scalar(&b3, &b5, &gX); pf->delta = compress(&p); tr_append_point(tr, "delta", pf->delta.b);
  fq_t c = tr_challenge_scalar(tr, "c");
  pf->z[0] = F_add(b1, F_mul(c, *x));
  pf->z[1] = F_add(b2, F_mul(c, *rX));
  pf->z[2] = F_add(b3, F_mul(c, *y));
  pf->z[3] = F_add(b4, F_mul(c, *rY));
  pf->z[4] = F_add(b5, F_mul(c, F_sub(*rZ, F_mul(*rX, *y))));
}
static int product_check(const cg_t *P, const cg_t *X, const fq_t *c, const mcg_t *g, const fq_t *z1, const fq_t *z2) {
  ge_t pP, pX, t, lhs;
  if (!ge_decompress(&pP, P->b) || !ge_decompress(&pX, X->b)) return 0;
  ge_scalarmul(&t, c, &pX); ge_add(&lhs, &pP, &t);
  ge_t rhs = commit_scalar(z1, z2, g);
  return ge_eq(&lhs, &rhs);
}
static int product_verify(const prodproof_t *pf, const mcg_t *g, merlin_t *tr, const cg_t *X, const cg_t *Y, const cg_t *Z) {
  tr_append_protocol_name(tr, "product proof");
  tr_append_point(tr, "X", X->b); tr_append_point(tr, "Y", Y->b); tr_append_point(tr, "Z", Z->b);
  tr_append_point(tr, "alpha", pf->alpha.b); tr_append_point(tr, "beta", pf->beta.b); tr_append_point(tr, "delta", pf->delta.b);
  fq_t c = tr_challenge_scalar(tr, "c");
  ge_t Xp;
  if (!ge_decompress(&Xp, X->b)) return 0;
  mcg_t gX = {1, &Xp, g->h};
  return product_check(&pf->alpha, X, &c, g, &pf->z[0], &pf->z[1]) && product_check(&pf->beta, Y, &c, g, &pf->z[2], &pf->z[3]) &&
         product_check(&pf->delta, Z, &c, &gX, &pf->z[2], &pf->z[4]);
}

/* ------------------------------------------------------------------ ZK sum-check */

typedef struct {
  int rounds, deg;
  cg_t *comm_polys, *comm_evals;
  dotproof_t *proofs;
} zksc_t;

static void zksc_alloc(zksc_t *p, int rounds, int deg) {
  p->rounds = rounds; p->deg = deg;
  p->comm_polys = (cg_t *)calloc((size_t)rounds, sizeof(cg_t));
  p->comm_evals = (cg_t *)calloc((size_t)rounds, sizeof(cg_t));
  p->proofs = (dotproof_t *)calloc((size_t)rounds, sizeof(dotproof_t));
}
static void zksc_free(zksc_t *p) { free(p->comm_polys); free(p->comm_evals); free(p->proofs); }

static void zksc_write(wbuf *w, const zksc_t *p) {
  /* ZKSumcheckInstanceProof { comm_polys, comm_evals, proofs } (sumcheck.rs:64-69) */
  w_u64(w, (uint64_t)p->rounds);
  for (int i = 0; i < p->rounds; i++) w_point(w, &p->comm_polys[i]);
  w_u64(w, (uint64_t)p->rounds);
  for (int i = 0; i < p->rounds; i++) w_point(w, &p->comm_evals[i]);
  w_u64(w, (uint64_t)p->rounds);
  for (int i = 0; i < p->rounds; i++) {
    const dotproof_t *d = &p->proofs[i];
    w_point(w, &d->delta); w_point(w, &d->beta);
    w_u64(w, (uint64_t)d->n);
    for (int k = 0; k < d->n; k++) w_scalar(w, &d->z[k]);
    w_scalar(w, &d->z_delta); w_scalar(w, &d->z_beta);
  }
}
static int zksc_read(rbuf *r, zksc_t *p, int deg) {
  uint64_t n = r_u64(r);
  if (r->bad || n > 64) return 0;
  zksc_alloc(p, (int)n, deg);
  for (uint64_t i = 0; i < n; i++) p->comm_polys[i] = r_point(r);
  if (r_u64(r) != n) return 0;
  for (uint64_t i = 0; i < n; i++) p->comm_evals[i] = r_point(r);
  if (r_u64(r) != n) return 0;
  for (uint64_t i = 0; i < n; i++) {
    dotproof_t *d = &p->proofs[i];
    d->delta = r_point(r); d->beta = r_point(r);
    uint64_t zn = r_u64(r);
    if (zn != (uint64_t)deg + 1) return 0;
    d->n = (int)zn;
    for (uint64_t k = 0; k < zn; k++) d->z[k] = r_scalar(r);
    d->z_delta = r_scalar(r); d->z_beta = r_scalar(r);
  }
  return !r->bad;
}

/* ZKSumcheckInstanceProof::prove_{cubic_with_additive_term,quad} (sumcheck.rs:428-776).
 * K = 4 tables (deg 3) or 2 tables (deg 2).  Tables are folded in place. Returns 0 if the
 * reference's internal assert (sumcheck.rs:531/722) would fire. */
static int zksc_prove(zksc_t *pf, int K, const fq_t *claim, const fq_t *blind_claim, int num_rounds,
                      fq_t **tabs, size_t len, const mcg_t *g1, const mcg_t *gn, merlin_t *tr, merlin_t *tape,
                      fq_t *r_out, fq_t *final_claims, fq_t *blind_last) {
  int deg = (K == 4) ? 3 : 2, nc = deg + 1;
  zksc_alloc(pf, num_rounds, deg);
  fq_t *blinds_poly = (fq_t *)malloc(sizeof(fq_t) * (size_t)num_rounds), *blinds_evals = (fq_t *)malloc(sizeof(fq_t) * (size_t)num_rounds);
  tr_challenge_vector(tape, "blinds_poly", blinds_poly, (size_t)num_rounds);
  tr_challenge_vector(tape, "blinds_evals", blinds_evals, (size_t)num_rounds);
  fq_t claim_pr = *claim;
  ge_t cp = commit_scalar(&claim_pr, blind_claim, g1);
  cg_t comm_claim = compress(&cp);
  int ok = 1;
  for (int j = 0; j < num_rounds; j++) {
    fq_t evals[4], coeffs[4];
    if (K == 4) {
      fq_t e[3];
      oracle_sc_cubic_round(tabs[0], tabs[1], tabs[2], tabs[3], len, e);
      evals[0] = e[0]; evals[1] = F_sub(claim_pr, e[0]); evals[2] = e[1]; evals[3] = e[2];
    } else {
      fq_t e[2];
      oracle_sc_quad_round(tabs[0], tabs[1], len, e);
      evals[0] = e[0]; evals[1] = F_sub(claim_pr, e[0]); evals[2] = e[1];
    }
    oracle_unipoly_from_evals(evals, nc, coeffs);
    ge_t t = commit_vec(coeffs, (size_t)nc, &blinds_poly[j], gn);
    pf->comm_polys[j] = compress(&t);
    tr_append_point(tr, "comm_poly", pf->comm_polys[j].b);
    fq_t r_j = tr_challenge_scalar(tr, "challenge_nextround");
    for (int k = 0; k < K; k++) oracle_bound_poly_var_top(tabs[k], len, &r_j);
    len /= 2;
    fq_t eval = oracle_unipoly_evaluate(coeffs, nc, &r_j);
    t = commit_scalar(&eval, &blinds_evals[j], g1);
    cg_t comm_eval = compress(&t);
    tr_append_point(tr, "comm_claim_per_round", comm_claim.b);
    tr_append_point(tr, "comm_eval", comm_eval.b);
    fq_t w[2];
    tr_challenge_vector(tr, "combine_two_claims_to_one", w, 2);
    fq_t target = F_add(F_mul(w[0], claim_pr), F_mul(w[1], eval));
    const fq_t *blind_sc = (j == 0) ? blind_claim : &blinds_evals[j - 1];
    fq_t blind = F_add(F_mul(w[0], *blind_sc), F_mul(w[1], blinds_evals[j]));
    { /* assert_eq!(target.commit(&blind, gens_1).compress(), comm_target) */
      ge_t a, b, s1, s2, ct;
      if (!ge_decompress(&a, comm_claim.b) || !ge_decompress(&b, comm_eval.b)) ok = 0;
      ge_scalarmul(&s1, &w[0], &a); ge_scalarmul(&s2, &w[1], &b); ge_add(&ct, &s1, &s2);
      ge_t tc = commit_scalar(&target, &blind, g1);
      cg_t c1 = compress(&ct), c2 = compress(&tc);
      if (memcmp(c1.b, c2.b, 32) != 0) ok = 0;
    }
    fq_t a[4], pw = fq_one();
    for (int i = 0; i < nc; i++) {
      fq_t a_sc = (i == 0) ? fq_from_u64(2) : fq_one();
      a[i] = F_add(F_mul(w[0], a_sc), F_mul(w[1], pw));
      pw = F_mul(pw, r_j);
    }
    dotproduct_prove(&pf->proofs[j], g1, gn, tr, tape, coeffs, &blinds_poly[j], a, &target, &blind, nc);
    claim_pr = eval;
    comm_claim = comm_eval;
    r_out[j] = r_j;
    pf->comm_evals[j] = comm_eval;
  }
  for (int k = 0; k < K; k++) final_claims[k] = tabs[k][0];
  *blind_last = blinds_evals[num_rounds - 1];
  free(blinds_poly); free(blinds_evals);
  return ok;
}

/* ZKSumcheckInstanceProof::verify (sumcheck.rs:84-172) */
static int zksc_verify(const zksc_t *pf, const cg_t *comm_claim, int num_rounds, int deg, const mcg_t *g1, const mcg_t *gn,
                       merlin_t *tr, cg_t *comm_out, fq_t *r_out) {
  if (pf->rounds != num_rounds || (int)gn->n != deg + 1) return 0;
  int nc = deg + 1;
  for (int i = 0; i < num_rounds; i++) {
    tr_append_point(tr, "comm_poly", pf->comm_polys[i].b);
    fq_t r_i = tr_challenge_scalar(tr, "challenge_nextround");
    const cg_t *ccl = (i == 0) ? comm_claim : &pf->comm_evals[i - 1];
    const cg_t *cev = &pf->comm_evals[i];
    tr_append_point(tr, "comm_claim_per_round", ccl->b);
    tr_append_point(tr, "comm_eval", cev->b);
    fq_t w[2];
    tr_challenge_vector(tr, "combine_two_claims_to_one", w, 2);
    ge_t a, b, s1, s2, ct;
    if (!ge_decompress(&a, ccl->b) || !ge_decompress(&b, cev->b)) return 0;
    ge_scalarmul(&s1, &w[0], &a); ge_scalarmul(&s2, &w[1], &b); ge_add(&ct, &s1, &s2);
    cg_t comm_target = compress(&ct);
    fq_t av[4], pw = fq_one();
    for (int k = 0; k < nc; k++) {
      fq_t a_sc = (k == 0) ? fq_from_u64(2) : fq_one();
      av[k] = F_add(F_mul(w[0], a_sc), F_mul(w[1], pw));
      pw = F_mul(pw, r_i);
    }
    if (!dotproduct_verify(&pf->proofs[i], g1, gn, tr, av, &pf->comm_polys[i], &comm_target)) return 0;
    r_out[i] = r_i;
  }
  *comm_out = pf->comm_evals[num_rounds - 1];
  return 1;
}

/* ------------------------------------------------------------------ PolyEvalProof (log) */

typedef struct { int lg; cg_t *Lv, *Rv; cg_t delta, beta; fq_t z1, z2; } dplog_t;
static void dplog_free(dplog_t *p) { free(p->Lv); free(p->Rv); }

/* BulletReductionProof::prove (nizk/bullet.rs:32-132) */
static void bullet_prove(dplog_t *pf, merlin_t *tr, const ge_t *Q, const ge_t *G_in, const ge_t *H,
                         const fq_t *a_in, const fq_t *b_in, size_t n, const fq_t *blind,
                         const fq_t *blinds1, const fq_t *blinds2,
                         fq_t *a_hat, fq_t *b_hat, ge_t *g_hat, fq_t *blind_fin_out) {
  ge_t *G = (ge_t *)malloc(sizeof(ge_t) * n);
  fq_t *a = (fq_t *)malloc(sizeof(fq_t) * n), *b = (fq_t *)malloc(sizeof(fq_t) * n);
  memcpy(G, G_in, sizeof(ge_t) * n); memcpy(a, a_in, sizeof(fq_t) * n); memcpy(b, b_in, sizeof(fq_t) * n);
  int lg = (int)log2z(n);
  pf->lg = lg;
  pf->Lv = (cg_t *)calloc((size_t)(lg ? lg : 1), sizeof(cg_t));
  pf->Rv = (cg_t *)calloc((size_t)(lg ? lg : 1), sizeof(cg_t));
  fq_t blind_fin = *blind;
  fq_t *sc = (fq_t *)malloc(sizeof(fq_t) * (n / 2 + 3));
  ge_t *pt = (ge_t *)malloc(sizeof(ge_t) * (n / 2 + 3));
  int round = 0;
  while (n != 1) {
    n /= 2;
    fq_t *aL = a, *aR = a + n, *bL = b, *bR = b + n;
    ge_t *GL = G, *GR = G + n;
    fq_t cL = oracle_dotproduct(aL, bR, n), cR = oracle_dotproduct(aR, bL, n);
    const fq_t *blind_L = &blinds1[round], *blind_R = &blinds2[round];
    ge_t Lp, Rp;
    memcpy(sc, aL, sizeof(fq_t) * n); sc[n] = cL; sc[n + 1] = *blind_L;
    memcpy(pt, GR, sizeof(ge_t) * n); pt[n] = *Q; pt[n + 1] = *H;
    ge_msm(&Lp, sc, pt, n + 2);
    memcpy(sc, aR, sizeof(fq_t) * n); sc[n] = cR; sc[n + 1] = *blind_R;
    memcpy(pt, GL, sizeof(ge_t) * n); pt[n] = *Q; pt[n + 1] = *H;
    ge_msm(&Rp, sc, pt, n + 2);
    pf->Lv[round] = compress(&Lp); pf->Rv[round] = compress(&Rp);
    tr_append_point(tr, "L", pf->Lv[round].b);
    tr_append_point(tr, "R", pf->Rv[round].b);
    fq_t u = tr_challenge_scalar(tr, "u"), u_inv = fq_invert(&u);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (long i = 0; i < (long)n; i++) {
      aL[i] = F_add(F_mul(aL[i], u), F_mul(u_inv, aR[i]));
      bL[i] = F_add(F_mul(bL[i], u_inv), F_mul(u, bR[i]));
      ge_t t1, t2;
      ge_scalarmul(&t1, &u_inv, &GL[i]); ge_scalarmul(&t2, &u, &GR[i]); ge_add(&GL[i], &t1, &t2);
    }
    blind_fin = F_add(F_add(blind_fin, F_mul(F_mul(*blind_L, u), u)), F_mul(F_mul(*blind_R, u_inv), u_inv));
    round++;
  }
  *a_hat = a[0]; *b_hat = b[0]; *g_hat = G[0]; *blind_fin_out = blind_fin;
  free(G); free(a); free(b); free(sc); free(pt);
}

/* DotProductProofLog::prove (nizk/mod.rs:447-531); returns C_y (the commitment to y) */
static cg_t dplog_prove(dplog_t *pf, const mcg_t *gn, const mcg_t *g1, merlin_t *tr, merlin_t *tape,
                        const fq_t *x, const fq_t *blind_x, const fq_t *a, const fq_t *y, const fq_t *blind_y, size_t n) {
  tr_append_protocol_name(tr, "dot product proof (log)");
  fq_t d = tr_challenge_scalar(tape, "d");
  fq_t r_delta = tr_challenge_scalar(tape, "r_delta");
  fq_t r_beta = tr_challenge_scalar(tape, "r_delta"); /* sic: the reference reuses the label */
  size_t lg = log2z(n);
  fq_t *bv1 = (fq_t *)malloc(sizeof(fq_t) * (2 * lg + 1)), *bv2 = (fq_t *)malloc(sizeof(fq_t) * (2 * lg + 1));
  tr_challenge_vector(tape, "blinds_vec_1", bv1, 2 * lg);
  tr_challenge_vector(tape, "blinds_vec_2", bv2, 2 * lg);
  ge_t t = commit_vec(x, n, blind_x, gn);
  cg_t Cx = compress(&t);
  tr_append_point(tr, "Cx", Cx.b);
  t = commit_scalar(y, blind_y, g1);
  cg_t Cy = compress(&t);
  tr_append_point(tr, "Cy", Cy.b);
  tr_append_scalars(tr, "a", a, n);
  fq_t r = tr_challenge_scalar(tr, "r");
  ge_t Gs; /* gens_1.scale(&r): G[0] scaled, h unchanged */
  ge_scalarmul(&Gs, &r, &g1->G[0]);
  fq_t blind_Gamma = F_add(*blind_x, F_mul(r, *blind_y));
  fq_t x_hat, a_hat, rhat_Gamma;
  ge_t g_hat;
  bullet_prove(pf, tr, &Gs, gn->G, &gn->h, x, a, n, &blind_Gamma, bv1, bv2, &x_hat, &a_hat, &g_hat, &rhat_Gamma);
  fq_t y_hat = F_mul(x_hat, a_hat);
  mcg_t ghat = {1, &g_hat, g1->h};
  t = commit_scalar(&d, &r_delta, &ghat);
  pf->delta = compress(&t);
  tr_append_point(tr, "delta", pf->delta.b);
  mcg_t g1s = {1, &Gs, g1->h};
  t = commit_scalar(&d, &r_beta, &g1s);
  pf->beta = compress(&t);
  tr_append_point(tr, "beta", pf->beta.b);
  fq_t c = tr_challenge_scalar(tr, "c");
  pf->z1 = F_add(d, F_mul(c, y_hat));
  pf->z2 = F_add(F_mul(a_hat, F_add(F_mul(c, rhat_Gamma), r_beta)), r_delta);
  free(bv1); free(bv2);
  return Cy;
}

/* BulletReductionProof::verify + DotProductProofLog::verify (bullet.rs:134-231, mod.rs:533-588) */
static int dplog_verify(const dplog_t *pf, size_t n, const mcg_t *gn, const mcg_t *g1, merlin_t *tr, const fq_t *a,
                        const cg_t *Cx, const cg_t *Cy) {
  tr_append_protocol_name(tr, "dot product proof (log)");
  tr_append_point(tr, "Cx", Cx->b);
  tr_append_point(tr, "Cy", Cy->b);
  tr_append_scalars(tr, "a", a, n);
  fq_t r = tr_challenge_scalar(tr, "r");
  ge_t Gs, pCx, pCy, Gamma, t;
  ge_scalarmul(&Gs, &r, &g1->G[0]);
  if (!ge_decompress(&pCx, Cx->b) || !ge_decompress(&pCy, Cy->b)) return 0;
  ge_scalarmul(&t, &r, &pCy); ge_add(&Gamma, &pCx, &t);
  int lg = pf->lg;
  if (((size_t)1 << lg) != n) return 0;
  fq_t *u = (fq_t *)malloc(sizeof(fq_t) * (size_t)(lg + 1)), *ui = (fq_t *)malloc(sizeof(fq_t) * (size_t)(lg + 1));
  for (int i = 0; i < lg; i++) {
    tr_append_point(tr, "L", pf->Lv[i].b);
    tr_append_point(tr, "R", pf->Rv[i].b);
    u[i] = tr_challenge_scalar(tr, "u");
    ui[i] = u[i];
  }
  fq_t allinv = fq_batch_invert(ui, (size_t)lg);
  for (int i = 0; i < lg; i++) { u[i] = fq_square(&u[i]); ui[i] = fq_square(&ui[i]); }
  fq_t *s = (fq_t *)malloc(sizeof(fq_t) * n);
  s[0] = allinv;
  for (size_t i = 1; i < n; i++) {
    int lg_i = 0;
    while (((size_t)2 << lg_i) <= i) lg_i++;
    size_t k = (size_t)1 << lg_i;
    s[i] = F_mul(s[i - k], u[(lg - 1) - lg_i]);
  }
  ge_t G_hat, Gamma_hat;
  ge_msm(&G_hat, s, gn->G, n);
  fq_t a_hat = oracle_dotproduct(a, s, n);
  size_t m = 2 * (size_t)lg + 1;
  fq_t *sc = (fq_t *)malloc(sizeof(fq_t) * m);
  ge_t *pt = (ge_t *)malloc(sizeof(ge_t) * m);
  int ok = 1;
  for (int i = 0; i < lg; i++) {
    sc[i] = u[i]; sc[lg + i] = ui[i];
    ok &= ge_decompress(&pt[i], pf->Lv[i].b);
    ok &= ge_decompress(&pt[lg + i], pf->Rv[i].b);
  }
  sc[2 * lg] = fq_one(); pt[2 * lg] = Gamma;
  if (ok) {
    ge_t acc; ge_identity(&acc);
    for (size_t i = 0; i < m; i++) { ge_scalarmul(&t, &sc[i], &pt[i]); ge_add(&acc, &acc, &t); }
    Gamma_hat = acc;
  }
  tr_append_point(tr, "delta", pf->delta.b);
  tr_append_point(tr, "beta", pf->beta.b);
  fq_t c = tr_challenge_scalar(tr, "c");
  ge_t pb, pd, lhs, rhs, t2;
  ok = ok && ge_decompress(&pb, pf->beta.b) && ge_decompress(&pd, pf->delta.b);
  if (ok) {
    /* lhs = ((Gamma_hat*c + beta)*a_hat + delta); rhs = (g_hat + Gs*a_hat)*z1 + h*z2 */
    ge_scalarmul(&t, &c, &Gamma_hat); ge_add(&t, &t, &pb); ge_scalarmul(&t, &a_hat, &t); ge_add(&lhs, &t, &pd);
    ge_scalarmul(&t, &a_hat, &Gs); ge_add(&t, &G_hat, &t); ge_scalarmul(&t, &pf->z1, &t);
    ge_scalarmul(&t2, &pf->z2, &g1->h); ge_add(&rhs, &t, &t2);
    ok = ge_eq(&lhs, &rhs);
  }
  free(u); free(ui); free(s); free(sc); free(pt);
  return ok;
}

#endif
