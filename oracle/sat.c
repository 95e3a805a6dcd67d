/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see fq.h / sat.h).
 * CPU restatement of vPIN's sat-proof prover and verifier.
 */
#define _POSIX_C_SOURCE 199309L
#include "sat.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static double g_timings[5];
void oracle_sat_last_timings(double out[5]) { memcpy(out, g_timings, sizeof g_timings); }

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

/* ------------------------------------------------------------------ R1CS instance */

void oracle_r1cs_multiply_vec(const r1cs_t *inst, const fq_t *z, fq_t *Az, fq_t *Bz, fq_t *Cz) {
  fq_t *out[3] = {Az, Bz, Cz};
  for (int m = 0; m < 3; m++) {
    for (size_t i = 0; i < inst->num_cons; i++) out[m][i] = fq_zero();
    for (size_t k = 0; k < inst->nnz[m]; k++) {
      fq_t t = fq_mul(&inst->val[m][k], &z[inst->col[m][k]]);
      out[m][inst->row[m][k]] = fq_add(&out[m][inst->row[m][k]], &t);
    }
  }
}

void oracle_r1cs_eval_table_sparse(const r1cs_t *inst, const fq_t *rx, fq_t *eA, fq_t *eB, fq_t *eC) {
  fq_t *out[3] = {eA, eB, eC};
  size_t ncols = 2 * inst->num_vars;
  for (int m = 0; m < 3; m++) {
    for (size_t i = 0; i < ncols; i++) out[m][i] = fq_zero();
    for (size_t k = 0; k < inst->nnz[m]; k++) {
      fq_t t = fq_mul(&rx[inst->row[m][k]], &inst->val[m][k]);
      out[m][inst->col[m][k]] = fq_add(&out[m][inst->col[m][k]], &t);
    }
  }
}

static fq_t *build_z(const r1cs_t *inst, const fq_t *vars, const fq_t *inputs) {
  /* commit_test.rs:162-170: z = [vars, 1, inputs, 0...] of length 2*num_vars */
  size_t nv = inst->num_vars;
  fq_t *z = (fq_t *)calloc(2 * nv, sizeof(fq_t));
  memcpy(z, vars, nv * sizeof(fq_t));
  z[nv] = fq_one();
  for (size_t i = 0; i < inst->num_inputs; i++) z[nv + 1 + i] = inputs[i];
  return z;
}

int oracle_r1cs_is_sat(const r1cs_t *inst, const fq_t *vars, const fq_t *inputs) {
  fq_t *z = build_z(inst, vars, inputs);
  size_t n = inst->num_cons;
  fq_t *Az = (fq_t *)malloc(3 * n * sizeof(fq_t)), *Bz = Az + n, *Cz = Bz + n;
  oracle_r1cs_multiply_vec(inst, z, Az, Bz, Cz);
  int ok = 1;
  for (size_t i = 0; i < n && ok; i++) {
    fq_t p = fq_mul(&Az[i], &Bz[i]);
    if (!fq_eq(&p, &Cz[i])) ok = 0;
  }
  free(Az); free(z);
  return ok;
}

void oracle_r1cs_evaluate(const r1cs_t *inst, const fq_t *rx, const fq_t *ry, fq_t out[3]) {
  size_t lx = log2z(inst->num_cons), ly = log2z(2 * inst->num_vars);
  fq_t *tx = (fq_t *)malloc(sizeof(fq_t) * inst->num_cons), *ty = (fq_t *)malloc(sizeof(fq_t) * 2 * inst->num_vars);
  oracle_eq_evals(rx, (int)lx, tx);
  oracle_eq_evals(ry, (int)ly, ty);
  for (int m = 0; m < 3; m++) {
    fq_t acc = fq_zero();
    for (size_t k = 0; k < inst->nnz[m]; k++) {
      fq_t t = fq_mul(&tx[inst->row[m][k]], &ty[inst->col[m][k]]);
      t = fq_mul(&t, &inst->val[m][k]);
      acc = fq_add(&acc, &t);
    }
    out[m] = acc;
  }
  free(tx); free(ty);
}

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

/* ------------------------------------------------------------------ the sat proof */

size_t oracle_sat_proof_max_bytes(size_t num_cons, size_t num_vars) {
  size_t lx = log2z(num_cons), ly = log2z(2 * num_vars), ell = log2z(num_vars);
  size_t L = (size_t)1 << (ell / 2), lgR = ell - ell / 2;
  return 8 + 32 * L + (lx + ly) * (64 + 64 + 8 + 4 * 32 + 64) + 48 + 4 * 32 + 96 + 3 * 32 + 5 * 32 + 2 * 64 +
         32 + 16 + 64 * lgR + 64 + 64 + 1024;
}

/* DensePolynomial::commit (dense_mlpoly.rs:193-218) */
static void dense_commit(cg_t *C, fq_t *blinds, const fq_t *Z, const satgens_t *sg, merlin_t *tape, int threads) {
  tr_challenge_vector(tape, "poly_blinds", blinds, sg->L);
  oracle_hyrax_commit((uint8_t *)C, Z, sg->L, sg->R, blinds, sg->pc_n.G, &sg->pc_n.h, threads);
}

size_t oracle_vpin_sat_prove(const r1cs_t *inst, const fq_t *vars_para, const fq_t *vars_input,
                             const fq_t *vars, const fq_t *inputs,
                             const uint8_t seed_commit64[64], const uint8_t seed_proof64[64], int threads,
                             uint8_t *proof_out, size_t proof_cap, uint8_t *comm_para_out, uint8_t *comm_input_out,
                             fq_t inst_evals[3], fq_t *rx, fq_t *ry) {
  double t_start = now_s();
  size_t nv = inst->num_vars, ncons = inst->num_cons;
  satgens_t sg;
  satgens_new(&sg, nv);
  size_t L = sg.L, R = sg.R;
  int ok = 1;

  /* proof_point_mult.rs:44-52: two commits under RandomTape::new(&[2u8]) */
  merlin_t tape1;
  const uint8_t two = 2;
  tape_init(&tape1, &two, 1, seed_commit64);
  cg_t *comm_para = (cg_t *)malloc(sizeof(cg_t) * L), *comm_input = (cg_t *)malloc(sizeof(cg_t) * L), *comm_vars = (cg_t *)malloc(sizeof(cg_t) * L);
  fq_t *blind_para = (fq_t *)malloc(sizeof(fq_t) * L), *blind_input = (fq_t *)malloc(sizeof(fq_t) * L), *blind_vars = (fq_t *)malloc(sizeof(fq_t) * L);
  double t0 = now_s();
  dense_commit(comm_para, blind_para, vars_para, &sg, &tape1, threads);
  dense_commit(comm_input, blind_input, vars_input, &sg, &tape1, threads);
  /* my_dense_mlpoly_commit's blinds = element-wise sum (commit_test.rs:42-54); the commitment the
   * proof carries is the row-wise sum of the two commitments (proof_point_mult.rs:75-80) */
  for (size_t i = 0; i < L; i++) {
    blind_vars[i] = F_add(blind_para[i], blind_input[i]);
    ge_t a, b, s;
    if (!ge_decompress(&a, comm_para[i].b) || !ge_decompress(&b, comm_input[i].b)) ok = 0;
    ge_add(&s, &a, &b);
    comm_vars[i] = compress(&s);
  }
  g_timings[0] = now_s() - t0;
  memcpy(comm_para_out, comm_para, 32 * L);
  memcpy(comm_input_out, comm_input, 32 * L);

  /* proof_point_mult.rs:83 + commit_test.rs:74-75 */
  merlin_t tr, tape;
  merlin_init(&tr, (const uint8_t *)"snark_example", 13);
  tape_init(&tape, (const uint8_t *)"proof", 5, seed_proof64);
  tr_append_protocol_name(&tr, "Spartan SNARK proof");

  /* my_R1CSProof_prove (commit_test.rs:136-334) */
  tr_append_protocol_name(&tr, "R1CS proof");
  /* PolyCommitment::append_to_transcript (dense_mlpoly.rs:305-313) */
  merlin_append_message(&tr, "poly_commitment", (const uint8_t *)"poly_commitment_begin", 21);
  for (size_t i = 0; i < L; i++) tr_append_point(&tr, "poly_commitment_share", comm_vars[i].b);
  merlin_append_message(&tr, "poly_commitment", (const uint8_t *)"poly_commitment_end", 19);

  t0 = now_s();
  fq_t *z = build_z(inst, vars, inputs);
  int nrx = (int)log2z(ncons), nry = (int)log2z(2 * nv);
  fq_t *tau = (fq_t *)malloc(sizeof(fq_t) * (size_t)nrx);
  tr_challenge_vector(&tr, "challenge_tau", tau, (size_t)nrx);
  fq_t *ptau = (fq_t *)malloc(sizeof(fq_t) * ncons * 4), *pAz = ptau + ncons, *pBz = pAz + ncons, *pCz = pBz + ncons;
  oracle_eq_evals(tau, nrx, ptau);
  oracle_r1cs_multiply_vec(inst, z, pAz, pBz, pCz);

  zksc_t sc1, sc2;
  fq_t claims1[4], blind_post1, zero = fq_zero();
  fq_t *tabs1[4] = {ptau, pAz, pBz, pCz};
  ok &= zksc_prove(&sc1, 4, &zero, &zero, nrx, tabs1, ncons, &sg.gens_1, &sg.gens_4, &tr, &tape, rx, claims1, &blind_post1);
  g_timings[1] = now_s() - t0;

  fq_t tau_claim = claims1[0], Az_claim = claims1[1], Bz_claim = claims1[2], Cz_claim = claims1[3];
  fq_t Az_blind = tr_challenge_scalar(&tape, "Az_blind"), Bz_blind = tr_challenge_scalar(&tape, "Bz_blind"),
       Cz_blind = tr_challenge_scalar(&tape, "Cz_blind"), prod_blind = tr_challenge_scalar(&tape, "prod_Az_Bz_blind");
  knowproof_t pok_Cz;
  cg_t comm_Cz = knowledge_prove(&pok_Cz, &sg.gens_1, &tr, &tape, &Cz_claim, &Cz_blind);
  prodproof_t pprod;
  cg_t comm_Az, comm_Bz, comm_prod;
  fq_t prod = F_mul(Az_claim, Bz_claim);
  product_prove(&pprod, &sg.gens_1, &tr, &tape, &Az_claim, &Az_blind, &Bz_claim, &Bz_blind, &prod, &prod_blind, &comm_Az, &comm_Bz, &comm_prod);
  tr_append_point(&tr, "comm_Az_claim", comm_Az.b);
  tr_append_point(&tr, "comm_Bz_claim", comm_Bz.b);
  tr_append_point(&tr, "comm_Cz_claim", comm_Cz.b);
  tr_append_point(&tr, "comm_prod_Az_Bz_claims", comm_prod.b);
  fq_t blind_expected1 = F_mul(tau_claim, F_sub(prod_blind, Cz_blind));
  fq_t claim_post1 = F_mul(F_sub(F_mul(Az_claim, Bz_claim), Cz_claim), tau_claim);
  eqproof_t eq1;
  equality_prove(&eq1, &sg.gens_1, &tr, &tape, &claim_post1, &blind_expected1, &claim_post1, &blind_post1);

  t0 = now_s();
  fq_t r_A = tr_challenge_scalar(&tr, "challenege_Az"), r_B = tr_challenge_scalar(&tr, "challenege_Bz"),
       r_C = tr_challenge_scalar(&tr, "challenege_Cz");
  fq_t claim2 = F_add(F_add(F_mul(r_A, Az_claim), F_mul(r_B, Bz_claim)), F_mul(r_C, Cz_claim));
  fq_t blind_claim2 = F_add(F_add(F_mul(r_A, Az_blind), F_mul(r_B, Bz_blind)), F_mul(r_C, Cz_blind));
  size_t zl = 2 * nv;
  fq_t *evals_rx = (fq_t *)malloc(sizeof(fq_t) * ncons);
  oracle_eq_evals(rx, nrx, evals_rx);
  fq_t *eA = (fq_t *)malloc(sizeof(fq_t) * zl * 4), *eB = eA + zl, *eC = eB + zl, *eABC = eC + zl;
  oracle_r1cs_eval_table_sparse(inst, evals_rx, eA, eB, eC);
  for (size_t i = 0; i < zl; i++) eABC[i] = F_add(F_add(F_mul(r_A, eA[i]), F_mul(r_B, eB[i])), F_mul(r_C, eC[i]));
  fq_t *zcopy = (fq_t *)malloc(sizeof(fq_t) * zl);
  memcpy(zcopy, z, sizeof(fq_t) * zl);
  fq_t *tabs2[2] = {zcopy, eABC};
  fq_t claims2[2], blind_post2;
  ok &= zksc_prove(&sc2, 2, &claim2, &blind_claim2, nry, tabs2, zl, &sg.gens_1, &sg.gens_3, &tr, &tape, ry, claims2, &blind_post2);
  g_timings[2] = now_s() - t0;

  t0 = now_s();
  fq_t eval_vars_at_ry = oracle_poly_evaluate(vars, ry + 1, nry - 1);
  fq_t blind_eval = tr_challenge_scalar(&tape, "blind_eval");
  /* PolyEvalProof::prove (dense_mlpoly.rs:326-379) */
  tr_append_protocol_name(&tr, "polynomial evaluation proof");
  size_t left = sg.ell / 2, right = sg.ell - left;
  fq_t *Lv = (fq_t *)malloc(sizeof(fq_t) * L), *Rv = (fq_t *)malloc(sizeof(fq_t) * R), *LZ = (fq_t *)malloc(sizeof(fq_t) * R);
  if (left) oracle_eq_evals(ry + 1, (int)left, Lv); else Lv[0] = fq_one();
  oracle_eq_evals(ry + 1 + left, (int)right, Rv);
  oracle_poly_bound(vars, Lv, L, R, LZ);
  fq_t LZ_blind = oracle_dotproduct(blind_vars, Lv, L);
  dplog_t pe;
  cg_t comm_vars_at_ry = dplog_prove(&pe, &sg.pc_n, &sg.pc_1, &tr, &tape, LZ, &LZ_blind, Rv, &eval_vars_at_ry, &blind_eval, R);
  g_timings[3] = now_s() - t0;

  fq_t one = fq_one();
  fq_t blind_eval_Z = F_mul(F_sub(one, ry[0]), blind_eval);
  fq_t blind_expected2 = F_mul(claims2[1], blind_eval_Z);
  fq_t claim_post2 = F_mul(claims2[0], claims2[1]);
  eqproof_t eq2;
  equality_prove(&eq2, &sg.pc_1, &tr, &tape, &claim_post2, &blind_expected2, &claim_post2, &blind_post2);

  /* my_lib_prove: inst.evaluate + claims appended (commit_test.rs:100-109) */
  oracle_r1cs_evaluate(inst, rx, ry, inst_evals);
  tr_append_scalar(&tr, "Ar_claim", &inst_evals[0]);
  tr_append_scalar(&tr, "Br_claim", &inst_evals[1]);
  tr_append_scalar(&tr, "Cr_claim", &inst_evals[2]);

  /* serialise R1CSProof (r1csproof.rs:21-47) */
  wbuf w = {proof_out, 0, proof_cap, 0};
  w_u64(&w, (uint64_t)L);
  for (size_t i = 0; i < L; i++) w_point(&w, &comm_vars[i]);
  zksc_write(&w, &sc1);
  w_point(&w, &comm_Az); w_point(&w, &comm_Bz); w_point(&w, &comm_Cz); w_point(&w, &comm_prod);
  w_point(&w, &pok_Cz.alpha); w_scalar(&w, &pok_Cz.z1); w_scalar(&w, &pok_Cz.z2);
  w_point(&w, &pprod.alpha); w_point(&w, &pprod.beta); w_point(&w, &pprod.delta);
  for (int i = 0; i < 5; i++) w_scalar(&w, &pprod.z[i]);
  w_point(&w, &eq1.alpha); w_scalar(&w, &eq1.z);
  zksc_write(&w, &sc2);
  w_point(&w, &comm_vars_at_ry);
  w_u64(&w, (uint64_t)pe.lg);
  for (int i = 0; i < pe.lg; i++) w_point(&w, &pe.Lv[i]);
  w_u64(&w, (uint64_t)pe.lg);
  for (int i = 0; i < pe.lg; i++) w_point(&w, &pe.Rv[i]);
  w_point(&w, &pe.delta); w_point(&w, &pe.beta); w_scalar(&w, &pe.z1); w_scalar(&w, &pe.z2);
  w_point(&w, &eq2.alpha); w_scalar(&w, &eq2.z);

  zksc_free(&sc1); zksc_free(&sc2); dplog_free(&pe);
  free(comm_para); free(comm_input); free(comm_vars); free(blind_para); free(blind_input); free(blind_vars);
  free(z); free(tau); free(ptau); free(evals_rx); free(eA); free(zcopy); free(Lv); free(Rv); free(LZ);
  satgens_free(&sg);
  g_timings[4] = now_s() - t_start;
  if (!ok || w.bad) return 0;
  return w.len;
}

int oracle_vpin_sat_verify(const uint8_t *proof, size_t proof_len, size_t num_cons, size_t num_vars,
                           const fq_t *inputs, size_t num_inputs, const fq_t inst_evals[3],
                           const uint8_t *comm_para, const uint8_t *comm_input, fq_t *rx, fq_t *ry) {
  satgens_t sg;
  satgens_new(&sg, num_vars);
  size_t L = sg.L, R = sg.R;
  rbuf r = {proof, proof_len, 0, 0};
  int ok = 1;
  zksc_t sc1 = {0}, sc2 = {0};
  dplog_t pe = {0};
  cg_t *comm_vars = NULL, *combined = NULL;

  if (r_u64(&r) != L) { ok = 0; goto done; }
  comm_vars = (cg_t *)malloc(sizeof(cg_t) * L);
  combined = (cg_t *)malloc(sizeof(cg_t) * L);
  for (size_t i = 0; i < L; i++) comm_vars[i] = r_point(&r);
  int nrx = (int)log2z(num_cons), nry = (int)log2z(2 * num_vars);
  if (!zksc_read(&r, &sc1, 3)) { ok = 0; goto done; }
  cg_t comm_Az = r_point(&r), comm_Bz = r_point(&r), comm_Cz = r_point(&r), comm_prod = r_point(&r);
  knowproof_t pok; pok.alpha = r_point(&r); pok.z1 = r_scalar(&r); pok.z2 = r_scalar(&r);
  prodproof_t pp; pp.alpha = r_point(&r); pp.beta = r_point(&r); pp.delta = r_point(&r);
  for (int i = 0; i < 5; i++) pp.z[i] = r_scalar(&r);
  eqproof_t eq1; eq1.alpha = r_point(&r); eq1.z = r_scalar(&r);
  if (!zksc_read(&r, &sc2, 2)) { ok = 0; goto done; }
  cg_t comm_vars_at_ry = r_point(&r);
  uint64_t lg = r_u64(&r);
  if (r.bad || lg > 40) { ok = 0; goto done; }
  pe.lg = (int)lg;
  pe.Lv = (cg_t *)calloc(lg ? lg : 1, sizeof(cg_t)); pe.Rv = (cg_t *)calloc(lg ? lg : 1, sizeof(cg_t));
  for (uint64_t i = 0; i < lg; i++) pe.Lv[i] = r_point(&r);
  if (r_u64(&r) != lg) { ok = 0; goto done; }
  for (uint64_t i = 0; i < lg; i++) pe.Rv[i] = r_point(&r);
  pe.delta = r_point(&r); pe.beta = r_point(&r); pe.z1 = r_scalar(&r); pe.z2 = r_scalar(&r);
  eqproof_t eq2; eq2.alpha = r_point(&r); eq2.z = r_scalar(&r);
  if (r.bad || r.pos != r.len) { ok = 0; goto done; }

  /* my_lib_verify / my_r1csproof_verify (commit_test.rs:340-530) */
  merlin_t tr;
  merlin_init(&tr, (const uint8_t *)"snark_example", 13);
  tr_append_protocol_name(&tr, "Spartan SNARK proof");
  tr_append_protocol_name(&tr, "R1CS proof");
  for (size_t i = 0; i < L; i++) {
    ge_t a, b, s;
    if (!ge_decompress(&a, comm_para + 32 * i) || !ge_decompress(&b, comm_input + 32 * i)) { ok = 0; goto done; }
    ge_add(&s, &a, &b);
    combined[i] = compress(&s);
  }
  merlin_append_message(&tr, "poly_commitment", (const uint8_t *)"poly_commitment_begin", 21);
  for (size_t i = 0; i < L; i++) tr_append_point(&tr, "poly_commitment_share", combined[i].b);
  merlin_append_message(&tr, "poly_commitment", (const uint8_t *)"poly_commitment_end", 19);
  fq_t tau[64];
  tr_challenge_vector(&tr, "challenge_tau", tau, (size_t)nrx);
  fq_t zero = fq_zero();
  ge_t c0 = commit_scalar(&zero, &zero, &sg.gens_1);
  cg_t claim_phase1 = compress(&c0), comm_post1;
  if (!zksc_verify(&sc1, &claim_phase1, nrx, 3, &sg.gens_1, &sg.gens_4, &tr, &comm_post1, rx)) { ok = 0; goto done; }
  if (!knowledge_verify(&pok, &sg.gens_1, &tr, &comm_Cz)) { ok = 0; goto done; }
  if (!product_verify(&pp, &sg.gens_1, &tr, &comm_Az, &comm_Bz, &comm_prod)) { ok = 0; goto done; }
  tr_append_point(&tr, "comm_Az_claim", comm_Az.b);
  tr_append_point(&tr, "comm_Bz_claim", comm_Bz.b);
  tr_append_point(&tr, "comm_Cz_claim", comm_Cz.b);
  tr_append_point(&tr, "comm_prod_Az_Bz_claims", comm_prod.b);
  fq_t taus_bound_rx = fq_one(), one = fq_one();
  for (int i = 0; i < nrx; i++)
    taus_bound_rx = F_mul(taus_bound_rx, F_add(F_mul(rx[i], tau[i]), F_mul(F_sub(one, rx[i]), F_sub(one, tau[i]))));
  {
    ge_t pprod, pCz, d, e;
    if (!ge_decompress(&pprod, comm_prod.b) || !ge_decompress(&pCz, comm_Cz.b)) { ok = 0; goto done; }
    ge_sub(&d, &pprod, &pCz);
    ge_scalarmul(&e, &taus_bound_rx, &d);
    cg_t expected1 = compress(&e);
    if (!equality_verify(&eq1, &sg.gens_1, &tr, &expected1, &comm_post1)) { ok = 0; goto done; }
  }
  fq_t r_A = tr_challenge_scalar(&tr, "challenege_Az"), r_B = tr_challenge_scalar(&tr, "challenege_Bz"),
       r_C = tr_challenge_scalar(&tr, "challenege_Cz");
  cg_t comm_claim2, comm_post2;
  {
    ge_t a, b, c, t, acc;
    if (!ge_decompress(&a, comm_Az.b) || !ge_decompress(&b, comm_Bz.b) || !ge_decompress(&c, comm_Cz.b)) { ok = 0; goto done; }
    ge_scalarmul(&acc, &r_A, &a); ge_scalarmul(&t, &r_B, &b); ge_add(&acc, &acc, &t);
    ge_scalarmul(&t, &r_C, &c); ge_add(&acc, &acc, &t);
    comm_claim2 = compress(&acc);
  }
  if (!zksc_verify(&sc2, &comm_claim2, nry, 2, &sg.gens_1, &sg.gens_3, &tr, &comm_post2, ry)) { ok = 0; goto done; }
  /* PolyEvalProof::verify (dense_mlpoly.rs:381-404) */
  {
    tr_append_protocol_name(&tr, "polynomial evaluation proof");
    size_t left = sg.ell / 2, right = sg.ell - left;
    fq_t *Lv = (fq_t *)malloc(sizeof(fq_t) * L), *Rv = (fq_t *)malloc(sizeof(fq_t) * R);
    if (left) oracle_eq_evals(ry + 1, (int)left, Lv); else Lv[0] = fq_one();
    oracle_eq_evals(ry + 1 + left, (int)right, Rv);
    ge_t *Cd = (ge_t *)malloc(sizeof(ge_t) * L);
    for (size_t i = 0; i < L; i++) ok &= ge_decompress(&Cd[i], comm_vars[i].b);
    ge_t C_LZ;
    ge_msm(&C_LZ, Lv, Cd, L);
    cg_t cC_LZ = compress(&C_LZ);
    if (ok) ok = dplog_verify(&pe, R, &sg.pc_n, &sg.pc_1, &tr, Rv, &cC_LZ, &comm_vars_at_ry);
    free(Lv); free(Rv); free(Cd);
    if (!ok) goto done;
  }
  {
    /* poly_input_eval: SparsePolynomial over [1, inputs...] at ry[1..] (commit_test.rs:457-468) */
    int nv_bits = (int)log2z(num_vars);
    fq_t pie = fq_zero();
    for (size_t e = 0; e < num_inputs + 1; e++) {
      fq_t chi = fq_one();
      for (int j = 0; j < nv_bits; j++) {
        int bit = (int)((e >> (nv_bits - j - 1)) & 1);
        chi = F_mul(chi, bit ? ry[1 + j] : F_sub(one, ry[1 + j]));
      }
      fq_t val = (e == 0) ? fq_one() : inputs[e - 1];
      pie = F_add(pie, F_mul(chi, val));
    }
    ge_t pv, t1, t2, cz;
    if (!ge_decompress(&pv, comm_vars_at_ry.b)) { ok = 0; goto done; }
    fq_t omr = F_sub(one, ry[0]);
    ge_scalarmul(&t1, &omr, &pv);
    ge_t cpie = commit_scalar(&pie, &zero, &sg.pc_1);
    ge_scalarmul(&t2, &ry[0], &cpie);
    ge_add(&cz, &t1, &t2);
    fq_t comb = F_add(F_add(F_mul(r_A, inst_evals[0]), F_mul(r_B, inst_evals[1])), F_mul(r_C, inst_evals[2]));
    ge_t e;
    ge_scalarmul(&e, &comb, &cz);
    cg_t expected2 = compress(&e);
    if (!equality_verify(&eq2, &sg.gens_1, &tr, &expected2, &comm_post2)) { ok = 0; goto done; }
  }
  /* the proof's comm_vars must be the combined commitment the verifier recomputed */
  for (size_t i = 0; i < L; i++)
    if (memcmp(comm_vars[i].b, combined[i].b, 32) != 0) ok = 0;
done:
  if (sc1.comm_polys) zksc_free(&sc1);
  if (sc2.comm_polys) zksc_free(&sc2);
  if (pe.Lv) dplog_free(&pe);
  free(comm_vars); free(combined);
  satgens_free(&sg);
  return ok;
}
