/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see fq.h / spark.h).
 * CPU restatement of SNARK::encode and R1CSEvalProof::{prove,verify} (SPARK).
 */
#define _POSIX_C_SOURCE 199309L
#include "spark.h"
#include "proto_common.h"
#include <time.h>

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static double g_spark_timings[7];
void oracle_spark_last_timings(double out[7]) { memcpy(out, g_spark_timings, sizeof g_spark_timings); }

size_t oracle_sat_prove_core(const r1cs_t *inst, const fq_t *vars_para, const fq_t *vars_input,
                             const fq_t *vars, const fq_t *inputs,
                             const uint8_t seed_commit64[64], const uint8_t seed_proof64[64], int threads,
                             uint8_t *proof_out, size_t proof_cap, uint8_t *comm_para_out, uint8_t *comm_input_out,
                             fq_t inst_evals[3], fq_t *rx, fq_t *ry, merlin_t *tr_out, merlin_t *tape_out);
int oracle_sat_verify_core(const uint8_t *proof, size_t proof_len, size_t num_cons, size_t num_vars,
                           const fq_t *inputs, size_t num_inputs, const fq_t *inst_evals_in, fq_t inst_evals_out[3],
                           const uint8_t *comm_para, const uint8_t *comm_input, fq_t *rx, fq_t *ry,
                           size_t *consumed, merlin_t *tr_out);

static size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }
static fq_t *fq_alloc(size_t n) { return (fq_t *)malloc(sizeof(fq_t) * (n ? n : 1)); }

/* ------------------------------------------------------------------ generators */

/* PolyCommitmentGens::new(num_vars, label) (dense_mlpoly.rs:26-33) as a view on the label's
 * SHAKE stream g[]: DotProductProofGens::new(R) = MultiCommitGens::new(R+1).split_at(R)
 * (nizk/mod.rs:411-425): gens_n = g[0..R), gens_1 = g[R], both with h = g[R+1]. */
typedef struct { size_t ell, L, R; mcg_t pc_n, pc_1; } pcgens_t;

static void pcgens_view(pcgens_t *p, const ge_t *g, size_t ell) {
  size_t left = ell / 2, right = ell - left;
  p->ell = ell; p->L = (size_t)1 << left; p->R = (size_t)1 << right;
  p->pc_n.n = p->R; p->pc_n.G = g; p->pc_n.h = g[p->R + 1];
  p->pc_1.n = 1; p->pc_1.G = g + p->R; p->pc_1.h = g[p->R + 1];
}

/* SparseMatPolyCommitmentGens::new (sparse_mlpoly.rs:300-329), batch_size = 3 */
typedef struct { ge_t *g; pcgens_t ops, mem, derefs; } sparkgens_t;

static void sparkgens_new(sparkgens_t *sg, size_t nx, size_t ny, size_t N) {
  size_t lgN = log2z(N);
  size_t v_ops = lgN + 4;                  /* (3*5).next_power_of_two().log_2() */
  size_t v_mem = (nx > ny ? nx : ny) + 1;
  size_t v_derefs = lgN + 3;               /* (3*2).next_power_of_two().log_2() */
  size_t vmax = v_ops > v_mem ? v_ops : v_mem;
  size_t Rmax = (size_t)1 << (vmax - vmax / 2);
  sg->g = (ge_t *)malloc(sizeof(ge_t) * (Rmax + 2));
  oracle_gens_new(sg->g, Rmax + 1, (const uint8_t *)"gens_r1cs_eval", 14);
  pcgens_view(&sg->ops, sg->g, v_ops);
  pcgens_view(&sg->mem, sg->g, v_mem);
  pcgens_view(&sg->derefs, sg->g, v_derefs);
}
static void sparkgens_free(sparkgens_t *sg) { free(sg->g); }

/* ------------------------------------------------------------------ dense representation */

struct spark_decomm {
  size_t nx, ny, N, M;
  uint32_t *row[3], *col[3];         /* ops_addr_usize, N each (sparse_to_dense_vecs :368-380) */
  uint32_t *row_ts[3], *col_ts[3];   /* read_ts, N each */
  uint32_t *row_audit, *col_audit;   /* audit_ts, M each */
  fq_t *val[3];                      /* N each */
  fq_t *comb_ops;                    /* 16N */
  fq_t *comb_mem;                    /* 2M */
};

void oracle_spark_decomm_free(spark_decomm_t *d) {
  if (!d) return;
  for (int m = 0; m < 3; m++) { free(d->row[m]); free(d->col[m]); free(d->row_ts[m]); free(d->col_ts[m]); free(d->val[m]); }
  free(d->row_audit); free(d->col_audit); free(d->comb_ops); free(d->comb_mem);
  free(d);
}

/* AddrTimestamps::new (sparse_mlpoly.rs:232-265): audit_ts runs on across the three matrices */
static void addr_timestamps(uint32_t *const addr[3], size_t N, size_t M, uint32_t *read_ts[3], uint32_t *audit) {
  memset(audit, 0, sizeof(uint32_t) * M);
  for (int m = 0; m < 3; m++)
    for (size_t i = 0; i < N; i++) {
      uint32_t a = addr[m][i];
      read_ts[m][i] = audit[a];
      audit[a] += 1;
    }
}

static void shape(const r1cs_t *inst, size_t *nx, size_t *ny, size_t *N, size_t *M) {
  *nx = log2z(inst->num_cons);
  *ny = log2z(2 * inst->num_vars);
  size_t mx = 1;
  for (int m = 0; m < 3; m++) { size_t p = next_pow2(inst->nnz[m]); if (p > mx) mx = p; } /* get_num_nz_entries :364 */
  *N = mx;
  *M = (size_t)1 << (*nx > *ny ? *nx : *ny);
}

size_t oracle_spark_comm_bytes(const r1cs_t *inst) {
  size_t nx, ny, N, M;
  shape(inst, &nx, &ny, &N, &M);
  size_t v_ops = log2z(N) + 4, v_mem = (nx > ny ? nx : ny) + 1;
  return 8 * 6 + 8 + 32 * ((size_t)1 << (v_ops / 2)) + 8 + 32 * ((size_t)1 << (v_mem / 2));
}

size_t oracle_snark_proof_max_bytes(const r1cs_t *inst) {
  size_t nx, ny, N, M;
  shape(inst, &nx, &ny, &N, &M);
  size_t lgN = log2z(N), lgM = log2z(M);
  size_t b = oracle_sat_proof_max_bytes(inst->num_cons, inst->num_vars) + 96;
  b += 8 + 32 * ((size_t)1 << ((lgN + 3) / 2));                 /* comm_derefs */
  b += 64 * 32;                                                   /* eval_* scalars */
  /* batched product proofs: per layer l rounds (l = 0..lg-1) of 3 scalars + vec headers + claims */
  b += lgN * (lgN * (8 + 96) + 64 + 24 * 32 + 64) + 18 * 32 + 64;
  b += lgM * (lgM * (8 + 96) + 64 + 8 * 32 + 64) + 64;
  b += 3 * (16 + 64 * 40 + 128 + 64) + 40 * 32 + 1024;          /* three PolyEvalProofs + hash-layer evals */
  return b;
}

static spark_decomm_t *dense_rep(const r1cs_t *inst) {
  spark_decomm_t *d = (spark_decomm_t *)calloc(1, sizeof *d);
  shape(inst, &d->nx, &d->ny, &d->N, &d->M);
  size_t N = d->N, M = d->M;
  for (int m = 0; m < 3; m++) {
    d->row[m] = (uint32_t *)calloc(N, 4); d->col[m] = (uint32_t *)calloc(N, 4);
    d->row_ts[m] = (uint32_t *)calloc(N, 4); d->col_ts[m] = (uint32_t *)calloc(N, 4);
    d->val[m] = (fq_t *)calloc(N, sizeof(fq_t));
    memcpy(d->row[m], inst->row[m], 4 * inst->nnz[m]);
    memcpy(d->col[m], inst->col[m], 4 * inst->nnz[m]);
    memcpy(d->val[m], inst->val[m], sizeof(fq_t) * inst->nnz[m]);
  }
  d->row_audit = (uint32_t *)calloc(M, 4); d->col_audit = (uint32_t *)calloc(M, 4);
  addr_timestamps(d->row, N, M, d->row_ts, d->row_audit);
  addr_timestamps(d->col, N, M, d->col_ts, d->col_audit);
  /* comb_ops = merge(row.ops_addr, row.read_ts, col.ops_addr, col.read_ts, val) (:418-426),
   * padded 15N -> 16N with zeros (dense_mlpoly.rs:272-285) */
  d->comb_ops = (fq_t *)calloc(16 * N, sizeof(fq_t));
  uint32_t *const *grp[4] = {d->row, d->row_ts, d->col, d->col_ts};
  for (int g = 0; g < 4; g++)
    for (int m = 0; m < 3; m++) {
      fq_t *dst = d->comb_ops + (size_t)(3 * g + m) * N;
      const uint32_t *src = grp[g][m];
      for (size_t i = 0; i < N; i++) dst[i] = fq_from_u64(src[i]);
    }
  for (int m = 0; m < 3; m++) memcpy(d->comb_ops + (size_t)(12 + m) * N, d->val[m], sizeof(fq_t) * N);
  /* comb_mem = row.audit_ts ++ col.audit_ts (:427-428) */
  d->comb_mem = (fq_t *)calloc(2 * M, sizeof(fq_t));
  for (size_t i = 0; i < M; i++) { d->comb_mem[i] = fq_from_u64(d->row_audit[i]); d->comb_mem[M + i] = fq_from_u64(d->col_audit[i]); }
  return d;
}

/* DensePolynomial::commit(gens, None) (dense_mlpoly.rs:193-218): zero blinds */
static cg_t *commit_noblind(const fq_t *Z, const pcgens_t *g, int threads) {
  cg_t *C = (cg_t *)malloc(sizeof(cg_t) * g->L);
  fq_t *blinds = (fq_t *)calloc(g->L, sizeof(fq_t));
  oracle_hyrax_commit((uint8_t *)C, Z, g->L, g->R, blinds, g->pc_n.G, &g->pc_n.h, threads);
  free(blinds);
  return C;
}

spark_decomm_t *oracle_spark_encode(const r1cs_t *inst, int threads, uint8_t *comm_out, size_t comm_cap, size_t *comm_len) {
  double t0 = now_s();
  spark_decomm_t *d = dense_rep(inst);
  sparkgens_t sg;
  sparkgens_new(&sg, d->nx, d->ny, d->N);
  cg_t *c_ops = commit_noblind(d->comb_ops, &sg.ops, threads);
  cg_t *c_mem = commit_noblind(d->comb_mem, &sg.mem, threads);
  /* R1CSCommitment { num_cons, num_vars, num_inputs, comm: SparseMatPolyCommitment { batch_size,
   * num_ops, num_mem_cells, comm_comb_ops, comm_comb_mem } } (r1csinstance.rs:53-58, sparse_mlpoly.rs:332-338) */
  wbuf w = {comm_out, 0, comm_cap, 0};
  w_u64(&w, inst->num_cons); w_u64(&w, inst->num_vars); w_u64(&w, inst->num_inputs);
  w_u64(&w, 3); w_u64(&w, d->N); w_u64(&w, d->M);
  w_u64(&w, sg.ops.L); for (size_t i = 0; i < sg.ops.L; i++) w_point(&w, &c_ops[i]);
  w_u64(&w, sg.mem.L); for (size_t i = 0; i < sg.mem.L; i++) w_point(&w, &c_mem[i]);
  free(c_ops); free(c_mem);
  sparkgens_free(&sg);
  g_spark_timings[0] = now_s() - t0;
  if (w.bad) { oracle_spark_decomm_free(d); return NULL; }
  *comm_len = w.len;
  return d;
}

/* ------------------------------------------------------------------ small protocol helpers */

/* AppendToTranscript for UniPoly (unipoly.rs:112-120) */
static void tr_append_unipoly(merlin_t *tr, const fq_t *coeffs, int n) {
  merlin_append_message(tr, "poly", (const uint8_t *)"UniPoly_begin", 13);
  for (int i = 0; i < n; i++) tr_append_scalar(tr, "coeff", &coeffs[i]);
  merlin_append_message(tr, "poly", (const uint8_t *)"UniPoly_end", 11);
}

/* PolyCommitment::append_to_transcript (dense_mlpoly.rs:305-313) */
static void tr_append_polycomm(merlin_t *tr, const char *label, const cg_t *C, size_t L) {
  merlin_append_message(tr, label, (const uint8_t *)"poly_commitment_begin", 21);
  for (size_t i = 0; i < L; i++) tr_append_point(tr, "poly_commitment_share", C[i].b);
  merlin_append_message(tr, label, (const uint8_t *)"poly_commitment_end", 19);
}

static void eq_evals_any(const fq_t *r, size_t ell, fq_t *out) {
  if (ell) oracle_eq_evals(r, (int)ell, out); else out[0] = fq_one();
}

/* fold `evals` (2^k entries) with challenges via bound_poly_var_bot in reverse order
 * (sparse_mlpoly.rs:104-109): returns the joint claim */
static fq_t combine_bot(fq_t *evals, size_t n, const fq_t *ch, size_t k) {
  for (size_t ii = k; ii-- > 0;) {
    n /= 2;
    for (size_t i = 0; i < n; i++) evals[i] = F_add(evals[2 * i], F_mul(ch[ii], F_sub(evals[2 * i + 1], evals[2 * i])));
  }
  return evals[0];
}

static void w_scalars(wbuf *w, const fq_t *v, size_t n) { w_u64(w, n); for (size_t i = 0; i < n; i++) w_scalar(w, &v[i]); }
static int r_scalars(rbuf *r, fq_t *v, size_t n) {
  if (r_u64(r) != n) return 0;
  for (size_t i = 0; i < n; i++) v[i] = r_scalar(r);
  return !r->bad;
}

static void dplog_write(wbuf *w, const dplog_t *p) {
  w_u64(w, (uint64_t)p->lg); for (int i = 0; i < p->lg; i++) w_point(w, &p->Lv[i]);
  w_u64(w, (uint64_t)p->lg); for (int i = 0; i < p->lg; i++) w_point(w, &p->Rv[i]);
  w_point(w, &p->delta); w_point(w, &p->beta); w_scalar(w, &p->z1); w_scalar(w, &p->z2);
}
static int dplog_read(rbuf *r, dplog_t *p) {
  uint64_t lg = r_u64(r);
  if (r->bad || lg > 40) return 0;
  p->lg = (int)lg;
  p->Lv = (cg_t *)calloc(lg ? lg : 1, sizeof(cg_t)); p->Rv = (cg_t *)calloc(lg ? lg : 1, sizeof(cg_t));
  for (uint64_t i = 0; i < lg; i++) p->Lv[i] = r_point(r);
  if (r_u64(r) != lg) return 0;
  for (uint64_t i = 0; i < lg; i++) p->Rv[i] = r_point(r);
  p->delta = r_point(r); p->beta = r_point(r); p->z1 = r_scalar(r); p->z2 = r_scalar(r);
  return !r->bad;
}

/* PolyEvalProof::prove with blinds_opt = None, blind_Zr_opt = None (dense_mlpoly.rs:326-379) */
static void polyeval_prove_plain(dplog_t *pf, const fq_t *Z, const fq_t *r, const fq_t *Zr, const pcgens_t *g,
                                 merlin_t *tr, merlin_t *tape) {
  tr_append_protocol_name(tr, "polynomial evaluation proof");
  size_t left = g->ell / 2, right = g->ell - left;
  fq_t *Lv = fq_alloc(g->L), *Rv = fq_alloc(g->R), *LZ = fq_alloc(g->R);
  eq_evals_any(r, left, Lv);
  eq_evals_any(r + left, right, Rv);
  oracle_poly_bound(Z, Lv, g->L, g->R, LZ);
  fq_t zero = fq_zero();
  (void)dplog_prove(pf, &g->pc_n, &g->pc_1, tr, tape, LZ, &zero, Rv, Zr, &zero, g->R);
  free(Lv); free(Rv); free(LZ);
}

/* PolyEvalProof::verify_plain (dense_mlpoly.rs:381-419) */
static int polyeval_verify_plain(const dplog_t *pf, const fq_t *r, const fq_t *Zr, const cg_t *C, size_t nC,
                                 const pcgens_t *g, merlin_t *tr) {
  if (nC != g->L) return 0;
  fq_t zero = fq_zero();
  ge_t czr = commit_scalar(Zr, &zero, &g->pc_1);
  cg_t C_Zr = compress(&czr);
  tr_append_protocol_name(tr, "polynomial evaluation proof");
  size_t left = g->ell / 2, right = g->ell - left;
  fq_t *Lv = fq_alloc(g->L), *Rv = fq_alloc(g->R);
  eq_evals_any(r, left, Lv);
  eq_evals_any(r + left, right, Rv);
  ge_t *Cd = (ge_t *)malloc(sizeof(ge_t) * g->L);
  int ok = 1;
  for (size_t i = 0; i < g->L; i++) ok &= ge_decompress(&Cd[i], C[i].b);
  if (ok) {
    ge_t C_LZ;
    ge_msm(&C_LZ, Lv, Cd, g->L);
    cg_t cC_LZ = compress(&C_LZ);
    ok = dplog_verify(pf, g->R, &g->pc_n, &g->pc_1, tr, Rv, &cC_LZ, &C_Zr);
  }
  free(Lv); free(Rv); free(Cd);
  return ok;
}

/* ------------------------------------------------------------------ product circuits */

/* ProductCircuit (product_tree.rs:12-66): left[l] / right[l] hold (n/2)>>l entries */
typedef struct { int layers; size_t n; fq_t **left, **right; } pcirc_t;

static void pcirc_new(pcirc_t *c, const fq_t *poly, size_t n) {
  int layers = (int)log2z(n);
  c->layers = layers; c->n = n;
  c->left = (fq_t **)calloc((size_t)layers, sizeof(fq_t *));
  c->right = (fq_t **)calloc((size_t)layers, sizeof(fq_t *));
  size_t h = n / 2;
  c->left[0] = fq_alloc(h); c->right[0] = fq_alloc(h);
  memcpy(c->left[0], poly, sizeof(fq_t) * h);
  memcpy(c->right[0], poly + h, sizeof(fq_t) * h);
  for (int l = 0; l + 1 < layers; l++) {
    size_t q = h / 2; /* len/4 with len = 2h */
    c->left[l + 1] = fq_alloc(q); c->right[l + 1] = fq_alloc(q);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (h > 4096)
#endif
    for (long i = 0; i < (long)h; i++) {
      fq_t p = F_mul(c->left[l][i], c->right[l][i]);
      if ((size_t)i < q) c->left[l + 1][i] = p; else c->right[l + 1][(size_t)i - q] = p;
    }
    h = q;
  }
}
static fq_t pcirc_eval(const pcirc_t *c) { return F_mul(c->left[c->layers - 1][0], c->right[c->layers - 1][0]); }
static void pcirc_free(pcirc_t *c) {
  for (int l = 0; l < c->layers; l++) { free(c->left[l]); free(c->right[l]); }
  free(c->left); free(c->right);
}

/* (e0, e2, e3) of one instance in prove_cubic_batched (sumcheck.rs:273-302), comb = A*B*C */
static void cubic_evals(const fq_t *A, const fq_t *B, const fq_t *C, size_t len, fq_t out[3]) {
  size_t h = len / 2;
  fq_t e0 = fq_zero(), e2 = fq_zero(), e3 = fq_zero();
#ifdef _OPENMP
#pragma omp parallel if (h > 2048)
#endif
  {
    fq_t l0 = fq_zero(), l2 = fq_zero(), l3 = fq_zero();
#ifdef _OPENMP
#pragma omp for schedule(static) nowait
#endif
    for (long i = 0; i < (long)h; i++) {
      fq_t a0 = A[i], a1 = A[h + i], b0 = B[i], b1 = B[h + i], c0 = C[i], c1 = C[h + i];
      l0 = F_add(l0, F_mul(F_mul(a0, b0), c0));
      fq_t a2 = F_sub(F_add(a1, a1), a0), b2 = F_sub(F_add(b1, b1), b0), c2 = F_sub(F_add(c1, c1), c0);
      l2 = F_add(l2, F_mul(F_mul(a2, b2), c2));
      fq_t a3 = F_sub(F_add(a2, a1), a0), b3 = F_sub(F_add(b2, b1), b0), c3 = F_sub(F_add(c2, c1), c0);
      l3 = F_add(l3, F_mul(F_mul(a3, b3), c3));
    }
#ifdef _OPENMP
#pragma omp critical
#endif
    { e0 = F_add(e0, l0); e2 = F_add(e2, l2); e3 = F_add(e3, l3); }
  }
  out[0] = e0; out[1] = e2; out[2] = e3;
}

static void bind_top(fq_t *Z, size_t len, const fq_t *r) { oracle_bound_poly_var_top(Z, len, r); }

/* one serialised LayerProofBatched + claims, kept until the struct order is known */
typedef struct {
  int num_layers, npc, ndotp;
  int *rounds;          /* per layer */
  fq_t **polys;         /* per layer: rounds*3 compressed coeffs */
  fq_t **claims_left, **claims_right; /* per layer: npc each */
  fq_t *dotp[3];        /* ndotp each (left, right, weight) */
} batched_t;

static void batched_free(batched_t *b) {
  for (int l = 0; l < b->num_layers; l++) { free(b->polys[l]); free(b->claims_left[l]); free(b->claims_right[l]); }
  free(b->rounds); free(b->polys); free(b->claims_left); free(b->claims_right);
  for (int k = 0; k < 3; k++) free(b->dotp[k]);
}

static void batched_write(wbuf *w, const batched_t *b) {
  /* ProductCircuitEvalProofBatched { proof: Vec<LayerProofBatched>, claims_dotp } (product_tree.rs:141-167) */
  w_u64(w, (uint64_t)b->num_layers);
  for (int l = 0; l < b->num_layers; l++) {
    w_u64(w, (uint64_t)b->rounds[l]);
    for (int j = 0; j < b->rounds[l]; j++) w_scalars(w, b->polys[l] + 3 * j, 3);
    w_scalars(w, b->claims_left[l], (size_t)b->npc);
    w_scalars(w, b->claims_right[l], (size_t)b->npc);
  }
  for (int k = 0; k < 3; k++) w_scalars(w, b->dotp[k], (size_t)b->ndotp);
}

static int batched_read(rbuf *r, batched_t *b, int num_layers, int npc, int ndotp) {
  memset(b, 0, sizeof *b);
  if (r_u64(r) != (uint64_t)num_layers) return 0;
  b->num_layers = num_layers; b->npc = npc; b->ndotp = ndotp;
  b->rounds = (int *)calloc((size_t)num_layers, sizeof(int));
  b->polys = (fq_t **)calloc((size_t)num_layers, sizeof(fq_t *));
  b->claims_left = (fq_t **)calloc((size_t)num_layers, sizeof(fq_t *));
  b->claims_right = (fq_t **)calloc((size_t)num_layers, sizeof(fq_t *));
  for (int k = 0; k < 3; k++) b->dotp[k] = fq_alloc((size_t)ndotp);
  for (int l = 0; l < num_layers; l++) {
    uint64_t nr = r_u64(r);
    if (r->bad || nr != (uint64_t)l) return 0; /* layer l (from the top) has l rounds */
    b->rounds[l] = (int)nr;
    b->polys[l] = fq_alloc(3 * (size_t)nr);
    for (uint64_t j = 0; j < nr; j++) if (!r_scalars(r, b->polys[l] + 3 * j, 3)) return 0; /* degree bound 3 */
    b->claims_left[l] = fq_alloc((size_t)npc); b->claims_right[l] = fq_alloc((size_t)npc);
    if (!r_scalars(r, b->claims_left[l], (size_t)npc)) return 0;
    if (!r_scalars(r, b->claims_right[l], (size_t)npc)) return 0;
  }
  for (int k = 0; k < 3; k++) if (!r_scalars(r, b->dotp[k], (size_t)ndotp)) return 0;
  return !r->bad;
}

/* ProductCircuitEvalProofBatched::prove (product_tree.rs:258-385).  dl/dr/dw: the DotProductCircuit
 * tables (dlen = n/2 entries each), consumed.  rand_out gets log2(n) entries. */
static void batched_prove(batched_t *b, pcirc_t **pc, int npc, fq_t **dl, fq_t **dr, fq_t **dw, int ndotp,
                          merlin_t *tr, fq_t *rand_out) {
  int num_layers = pc[0]->layers;
  size_t n = pc[0]->n;
  memset(b, 0, sizeof *b);
  b->num_layers = num_layers; b->npc = npc; b->ndotp = ndotp;
  b->rounds = (int *)calloc((size_t)num_layers, sizeof(int));
  b->polys = (fq_t **)calloc((size_t)num_layers, sizeof(fq_t *));
  b->claims_left = (fq_t **)calloc((size_t)num_layers, sizeof(fq_t *));
  b->claims_right = (fq_t **)calloc((size_t)num_layers, sizeof(fq_t *));
  for (int k = 0; k < 3; k++) b->dotp[k] = fq_alloc((size_t)ndotp);

  int maxc = npc + ndotp;
  fq_t *claims = fq_alloc((size_t)maxc), *coeffs = fq_alloc((size_t)maxc);
  fq_t *rand = fq_alloc((size_t)num_layers + 1), *rand_prod = fq_alloc((size_t)num_layers + 1);
  fq_t *polyC = fq_alloc(n / 2);
  int nrand = 0;
  for (int i = 0; i < npc; i++) claims[i] = pcirc_eval(pc[i]);

  for (int layer_id = num_layers - 1, out = 0; layer_id >= 0; layer_id--, out++) {
    size_t h = (n / 2) >> layer_id; /* entries in left[layer_id] = len/2 */
    eq_evals_any(rand, (size_t)nrand, polyC);
    int num_rounds = (int)log2z(h);
    int nclaims = npc;
    int with_dotp = (layer_id == 0 && ndotp > 0);
    if (with_dotp) {
      for (int k = 0; k < ndotp; k++) {
        fq_t acc = fq_zero();
        for (size_t i = 0; i < h; i++) acc = F_add(acc, F_mul(F_mul(dl[k][i], dr[k][i]), dw[k][i]));
        claims[npc + k] = acc;
      }
      nclaims += ndotp;
    }
    tr_challenge_vector(tr, "rand_coeffs_next_layer", coeffs, (size_t)nclaims);
    fq_t e = fq_zero();
    for (int i = 0; i < nclaims; i++) e = F_add(e, F_mul(claims[i], coeffs[i]));

    b->rounds[out] = num_rounds;
    b->polys[out] = fq_alloc(3 * (size_t)num_rounds);
    size_t len = h;
    for (int j = 0; j < num_rounds; j++) {
      fq_t c0 = fq_zero(), c2 = fq_zero(), c3 = fq_zero(), ev[3];
      for (int i = 0; i < nclaims; i++) {
        if (i < npc) cubic_evals(pc[i]->left[layer_id], pc[i]->right[layer_id], polyC, len, ev);
        else cubic_evals(dl[i - npc], dr[i - npc], dw[i - npc], len, ev);
        c0 = F_add(c0, F_mul(ev[0], coeffs[i]));
        c2 = F_add(c2, F_mul(ev[1], coeffs[i]));
        c3 = F_add(c3, F_mul(ev[2], coeffs[i]));
      }
      fq_t evals[4] = {c0, F_sub(e, c0), c2, c3}, cf[4];
      oracle_unipoly_from_evals(evals, 4, cf);
      tr_append_unipoly(tr, cf, 4);
      fq_t r_j = tr_challenge_scalar(tr, "challenge_nextround");
      rand_prod[j] = r_j;
      for (int i = 0; i < npc; i++) { bind_top(pc[i]->left[layer_id], len, &r_j); bind_top(pc[i]->right[layer_id], len, &r_j); }
      bind_top(polyC, len, &r_j);
      if (with_dotp)
        for (int k = 0; k < ndotp; k++) { bind_top(dl[k], len, &r_j); bind_top(dr[k], len, &r_j); bind_top(dw[k], len, &r_j); }
      e = oracle_unipoly_evaluate(cf, 4, &r_j);
      b->polys[out][3 * j] = cf[0]; b->polys[out][3 * j + 1] = cf[2]; b->polys[out][3 * j + 2] = cf[3];
      len /= 2;
    }
    b->claims_left[out] = fq_alloc((size_t)npc); b->claims_right[out] = fq_alloc((size_t)npc);
    for (int i = 0; i < npc; i++) {
      b->claims_left[out][i] = pc[i]->left[layer_id][0];
      b->claims_right[out][i] = pc[i]->right[layer_id][0];
      tr_append_scalar(tr, "claim_prod_left", &b->claims_left[out][i]);
      tr_append_scalar(tr, "claim_prod_right", &b->claims_right[out][i]);
    }
    if (with_dotp)
      for (int k = 0; k < ndotp; k++) {
        b->dotp[0][k] = dl[k][0]; b->dotp[1][k] = dr[k][0]; b->dotp[2][k] = dw[k][0];
        tr_append_scalar(tr, "claim_dotp_left", &b->dotp[0][k]);
        tr_append_scalar(tr, "claim_dotp_right", &b->dotp[1][k]);
        tr_append_scalar(tr, "claim_dotp_weight", &b->dotp[2][k]);
      }
    fq_t r_layer = tr_challenge_scalar(tr, "challenge_r_layer");
    for (int i = 0; i < npc; i++)
      claims[i] = F_add(b->claims_left[out][i], F_mul(r_layer, F_sub(b->claims_right[out][i], b->claims_left[out][i])));
    rand[0] = r_layer;
    memcpy(rand + 1, rand_prod, sizeof(fq_t) * (size_t)num_rounds);
    nrand = num_rounds + 1;
  }
  memcpy(rand_out, rand, sizeof(fq_t) * (size_t)nrand);
  free(claims); free(coeffs); free(rand); free(rand_prod); free(polyC);
}

/* SumcheckInstanceProof::verify (sumcheck.rs:27-61), degree bound 3 */
static int sc_verify(const fq_t *polys, int num_rounds, fq_t claim, merlin_t *tr, fq_t *e_out, fq_t *r_out) {
  fq_t e = claim;
  for (int i = 0; i < num_rounds; i++) {
    const fq_t *c = polys + 3 * i;
    /* CompressedUniPoly::decompress (unipoly.rs:98-109) */
    fq_t lin = F_sub(F_sub(F_sub(F_sub(e, c[0]), c[0]), c[1]), c[2]);
    fq_t cf[4] = {c[0], lin, c[1], c[2]};
    /* eval(0) + eval(1) == e holds by construction of lin; the reference asserts it all the same */
    fq_t s01 = F_add(cf[0], F_add(F_add(cf[0], cf[1]), F_add(cf[2], cf[3])));
    if (!fq_eq(&s01, &e)) return 0;
    tr_append_unipoly(tr, cf, 4);
    fq_t r_i = tr_challenge_scalar(tr, "challenge_nextround");
    r_out[i] = r_i;
    e = oracle_unipoly_evaluate(cf, 4, &r_i);
  }
  *e_out = e;
  return 1;
}

/* ProductCircuitEvalProofBatched::verify (product_tree.rs:387-485).
 * claims_prod: npc; claims_dotp: ndotp (may be 0).  Outputs: claims_out (npc), claims_dotp_out
 * (3*ndotp/2), rand_out (num_layers). */
static int batched_verify(const batched_t *b, const fq_t *claims_prod, const fq_t *claims_dotp, merlin_t *tr,
                          fq_t *claims_out, fq_t *claims_dotp_out, fq_t *rand_out) {
  int num_layers = b->num_layers, npc = b->npc, ndotp = b->ndotp;
  int maxc = npc + ndotp;
  fq_t *claims = fq_alloc((size_t)maxc), *coeffs = fq_alloc((size_t)maxc);
  fq_t *rand = fq_alloc((size_t)num_layers + 1), *rand_prod = fq_alloc((size_t)num_layers + 1);
  int nrand = 0, ok = 1;
  memcpy(claims, claims_prod, sizeof(fq_t) * (size_t)npc);
  fq_t one = fq_one();
  for (int i = 0; i < num_layers && ok; i++) {
    int nclaims = npc;
    if (i == num_layers - 1) { memcpy(claims + npc, claims_dotp, sizeof(fq_t) * (size_t)ndotp); nclaims += ndotp; }
    tr_challenge_vector(tr, "rand_coeffs_next_layer", coeffs, (size_t)nclaims);
    fq_t claim = fq_zero();
    for (int k = 0; k < nclaims; k++) claim = F_add(claim, F_mul(claims[k], coeffs[k]));
    fq_t claim_last;
    if (b->rounds[i] != i || !sc_verify(b->polys[i], i, claim, tr, &claim_last, rand_prod)) { ok = 0; break; }
    const fq_t *cl = b->claims_left[i], *cr = b->claims_right[i];
    for (int k = 0; k < npc; k++) { tr_append_scalar(tr, "claim_prod_left", &cl[k]); tr_append_scalar(tr, "claim_prod_right", &cr[k]); }
    if (nrand != i) { ok = 0; break; }
    fq_t eq = fq_one();
    for (int k = 0; k < nrand; k++)
      eq = F_mul(eq, F_add(F_mul(rand[k], rand_prod[k]), F_mul(F_sub(one, rand[k]), F_sub(one, rand_prod[k]))));
    fq_t expected = fq_zero();
    for (int k = 0; k < npc; k++) expected = F_add(expected, F_mul(coeffs[k], F_mul(F_mul(cl[k], cr[k]), eq)));
    if (i == num_layers - 1)
      for (int k = 0; k < ndotp; k++) {
        tr_append_scalar(tr, "claim_dotp_left", &b->dotp[0][k]);
        tr_append_scalar(tr, "claim_dotp_right", &b->dotp[1][k]);
        tr_append_scalar(tr, "claim_dotp_weight", &b->dotp[2][k]);
        expected = F_add(expected, F_mul(F_mul(F_mul(coeffs[npc + k], b->dotp[0][k]), b->dotp[1][k]), b->dotp[2][k]));
      }
    if (!fq_eq(&expected, &claim_last)) { ok = 0; break; }
    fq_t r_layer = tr_challenge_scalar(tr, "challenge_r_layer");
    for (int k = 0; k < npc; k++) claims[k] = F_add(cl[k], F_mul(r_layer, F_sub(cr[k], cl[k])));
    if (i == num_layers - 1)
      for (int k = 0; k < ndotp / 2; k++)
        for (int t = 0; t < 3; t++) {
          const fq_t *v = b->dotp[t];
          claims_dotp_out[3 * k + t] = F_add(v[2 * k], F_mul(r_layer, F_sub(v[2 * k + 1], v[2 * k])));
        }
    rand[0] = r_layer;
    memcpy(rand + 1, rand_prod, sizeof(fq_t) * (size_t)i);
    nrand = i + 1;
  }
  if (ok) { memcpy(claims_out, claims, sizeof(fq_t) * (size_t)npc); memcpy(rand_out, rand, sizeof(fq_t) * (size_t)nrand); }
  free(claims); free(coeffs); free(rand); free(rand_prod);
  return ok;
}

/* ------------------------------------------------------------------ hash layer / network */

/* Layers::build_hash_layer + Layers::new (sparse_mlpoly.rs:547-672) for one side */
typedef struct { pcirc_t init, audit, read[3], write[3]; } layers_t;

static void layers_new(layers_t *ly, const fq_t *mem, size_t M, uint32_t *const addr[3], fq_t *const derefs[3],
                       uint32_t *const read_ts[3], const uint32_t *audit_ts, size_t N, const fq_t *r_hash, const fq_t *gamma) {
  fq_t r2 = F_mul(*r_hash, *r_hash);
  fq_t *tmp = fq_alloc(M > N ? M : N);
  /* hash_func(addr, val, ts) = ts*r^2 + val*r + addr, minus r_multiset_check */
  for (size_t i = 0; i < M; i++) tmp[i] = F_sub(F_add(F_mul(mem[i], *r_hash), fq_from_u64(i)), *gamma);
  pcirc_new(&ly->init, tmp, M);
  for (size_t i = 0; i < M; i++)
    tmp[i] = F_sub(F_add(F_add(F_mul(fq_from_u64(audit_ts[i]), r2), F_mul(mem[i], *r_hash)), fq_from_u64(i)), *gamma);
  pcirc_new(&ly->audit, tmp, M);
  for (int m = 0; m < 3; m++) {
    for (size_t i = 0; i < N; i++)
      tmp[i] = F_sub(F_add(F_add(F_mul(fq_from_u64(read_ts[m][i]), r2), F_mul(derefs[m][i], *r_hash)), fq_from_u64(addr[m][i])), *gamma);
    pcirc_new(&ly->read[m], tmp, N);
    for (size_t i = 0; i < N; i++)
      tmp[i] = F_sub(F_add(F_add(F_mul(fq_from_u64((uint64_t)read_ts[m][i] + 1), r2), F_mul(derefs[m][i], *r_hash)), fq_from_u64(addr[m][i])), *gamma);
    pcirc_new(&ly->write[m], tmp, N);
  }
  free(tmp);
}
static void layers_free(layers_t *ly) {
  pcirc_free(&ly->init); pcirc_free(&ly->audit);
  for (int m = 0; m < 3; m++) { pcirc_free(&ly->read[m]); pcirc_free(&ly->write[m]); }
}

/* SparseMatPolyEvalProof::equalize (sparse_mlpoly.rs:1448-1465) */
static void equalize(const fq_t *rx, size_t nx, const fq_t *ry, size_t ny, fq_t *rx_ext, fq_t *ry_ext, size_t *n) {
  size_t m = nx > ny ? nx : ny;
  for (size_t i = 0; i < m - nx; i++) rx_ext[i] = fq_zero();
  memcpy(rx_ext + (m - nx), rx, sizeof(fq_t) * nx);
  for (size_t i = 0; i < m - ny; i++) ry_ext[i] = fq_zero();
  memcpy(ry_ext + (m - ny), ry, sizeof(fq_t) * ny);
  *n = m;
}

static fq_t eval_u32(const uint32_t *v, size_t n, const fq_t *eq) {
  fq_t acc = fq_zero();
  for (size_t i = 0; i < n; i++) if (v[i]) acc = F_add(acc, F_mul(fq_from_u64(v[i]), eq[i]));
  return acc;
}

/* the evaluation values a proof carries (ProductLayerProof / HashLayerProof scalars) */
typedef struct {
  /* ProductLayerProof (sparse_mlpoly.rs:1036-1042) */
  fq_t pl_row[8], pl_col[8]; /* init, read[3], write[3], audit */
  fq_t dotp_left[3], dotp_right[3];
  /* HashLayerProof (sparse_mlpoly.rs:698-707) */
  fq_t hl_row_addr[3], hl_row_ts[3], hl_row_audit;
  fq_t hl_col_addr[3], hl_col_ts[3], hl_col_audit;
  fq_t hl_val[3], hl_deref_row[3], hl_deref_col[3];
} spark_evals_t;

/* SparseMatPolyEvalProof::prove (sparse_mlpoly.rs:1466-1533) -> bincode(R1CSEvalProof) appended to w */
static int spark_prove(wbuf *w, const spark_decomm_t *d, const fq_t *rx, const fq_t *ry, const fq_t evals[3],
                       const sparkgens_t *sg, merlin_t *tr, merlin_t *tape, int threads) {
  size_t N = d->N, M = d->M, lgN = log2z(N), lgM = log2z(M);
  double t0 = now_s();
  tr_append_protocol_name(tr, "Sparse polynomial evaluation proof");
  fq_t rx_ext[64], ry_ext[64];
  size_t nm;
  equalize(rx, d->nx, ry, d->ny, rx_ext, ry_ext, &nm);
  fq_t *mem_rx = fq_alloc(M), *mem_ry = fq_alloc(M);
  eq_evals_any(rx_ext, nm, mem_rx);
  eq_evals_any(ry_ext, nm, mem_ry);

  /* dense.deref (:525-531, :267-283) and Derefs::new's comb (:56-71) */
  fq_t *comb = (fq_t *)calloc(8 * N, sizeof(fq_t));
  fq_t *drow[3], *dcol[3];
  for (int m = 0; m < 3; m++) {
    drow[m] = comb + (size_t)m * N; dcol[m] = comb + (size_t)(3 + m) * N;
    for (size_t i = 0; i < N; i++) { drow[m][i] = mem_rx[d->row[m][i]]; dcol[m][i] = mem_ry[d->col[m][i]]; }
  }
  cg_t *comm_derefs = commit_noblind(comb, &sg->derefs, threads);
  /* DerefsCommitment::append_to_transcript (:216-222) */
  merlin_append_message(tr, "derefs_commitment", (const uint8_t *)"begin_derefs_commitment", 23);
  tr_append_polycomm(tr, "comm_poly_row_col_ops_val", comm_derefs, sg->derefs.L);
  merlin_append_message(tr, "derefs_commitment", (const uint8_t *)"end_derefs_commitment", 21);
  g_spark_timings[2] = now_s() - t0;

  t0 = now_s();
  fq_t r_mem_check[2];
  tr_challenge_vector(tr, "challenge_r_hash", r_mem_check, 2);
  layers_t row_l, col_l;
  layers_new(&row_l, mem_rx, M, d->row, drow, d->row_ts, d->row_audit, N, &r_mem_check[0], &r_mem_check[1]);
  layers_new(&col_l, mem_ry, M, d->col, dcol, d->col_ts, d->col_audit, N, &r_mem_check[0], &r_mem_check[1]);
  g_spark_timings[3] = now_s() - t0;

  t0 = now_s();
  /* PolyEvalNetworkProof::prove (:1336-1370) */
  tr_append_protocol_name(tr, "Sparse polynomial evaluation proof");
  /* ProductLayerProof::prove (:1049-1227) */
  tr_append_protocol_name(tr, "Sparse polynomial product layer proof");
  spark_evals_t ev;
  int ok = 1;
  layers_t *sides[2] = {&row_l, &col_l};
  fq_t *plv[2] = {ev.pl_row, ev.pl_col};
  static const char *lab[2][4] = {{"claim_row_eval_init", "claim_row_eval_read", "claim_row_eval_write", "claim_row_eval_audit"},
                                  {"claim_col_eval_init", "claim_col_eval_read", "claim_col_eval_write", "claim_col_eval_audit"}};
  for (int s = 0; s < 2; s++) {
    fq_t *p = plv[s];
    p[0] = pcirc_eval(&sides[s]->init);
    p[7] = pcirc_eval(&sides[s]->audit);
    fq_t ws = fq_one(), rs = fq_one();
    for (int m = 0; m < 3; m++) {
      p[1 + m] = pcirc_eval(&sides[s]->read[m]);
      p[4 + m] = pcirc_eval(&sides[s]->write[m]);
      rs = F_mul(rs, p[1 + m]); ws = F_mul(ws, p[4 + m]);
    }
    fq_t lhs = F_mul(p[0], ws), rhs = F_mul(rs, p[7]);
    if (!fq_eq(&lhs, &rhs)) ok = 0; /* assert_eq!(row_eval_init * ws, rs * row_eval_audit) */
    tr_append_scalar(tr, lab[s][0], &p[0]);
    tr_append_scalars(tr, lab[s][1], p + 1, 3);
    tr_append_scalars(tr, lab[s][2], p + 4, 3);
    tr_append_scalar(tr, lab[s][3], &p[7]);
  }
  /* DotProductCircuit halves (:1103-1125): order left_A, right_A, left_B, right_B, left_C, right_C */
  size_t h = N / 2;
  fq_t *dl[6], *dr[6], *dw[6];
  for (int m = 0; m < 3; m++) {
    for (int half = 0; half < 2; half++) {
      int k = 2 * m + half;
      dl[k] = fq_alloc(h); dr[k] = fq_alloc(h); dw[k] = fq_alloc(h);
      memcpy(dl[k], drow[m] + (size_t)half * h, sizeof(fq_t) * h);
      memcpy(dr[k], dcol[m] + (size_t)half * h, sizeof(fq_t) * h);
      memcpy(dw[k], d->val[m] + (size_t)half * h, sizeof(fq_t) * h);
      fq_t acc = fq_zero();
      for (size_t i = 0; i < h; i++) acc = F_add(acc, F_mul(F_mul(dl[k][i], dr[k][i]), dw[k][i]));
      if (half == 0) ev.dotp_left[m] = acc; else ev.dotp_right[m] = acc;
    }
    tr_append_scalar(tr, "claim_eval_dotp_left", &ev.dotp_left[m]);
    tr_append_scalar(tr, "claim_eval_dotp_right", &ev.dotp_right[m]);
    fq_t sum = F_add(ev.dotp_left[m], ev.dotp_right[m]);
    if (!fq_eq(&sum, &evals[m])) ok = 0; /* assert_eq!(eval_dotp_left + eval_dotp_right, eval[i]) */
  }
  pcirc_t *ops_c[12] = {&row_l.read[0], &row_l.read[1], &row_l.read[2], &row_l.write[0], &row_l.write[1], &row_l.write[2],
                        &col_l.read[0], &col_l.read[1], &col_l.read[2], &col_l.write[0], &col_l.write[1], &col_l.write[2]};
  batched_t pf_ops, pf_mem;
  fq_t rand_ops[64], rand_mem[64];
  batched_prove(&pf_ops, ops_c, 12, dl, dr, dw, 6, tr, rand_ops);
  pcirc_t *mem_c[4] = {&row_l.init, &row_l.audit, &col_l.init, &col_l.audit};
  batched_prove(&pf_mem, mem_c, 4, NULL, NULL, NULL, 0, tr, rand_mem);
  for (int k = 0; k < 6; k++) { free(dl[k]); free(dr[k]); free(dw[k]); }
  layers_free(&row_l); layers_free(&col_l);
  g_spark_timings[4] = now_s() - t0;

  t0 = now_s();
  /* HashLayerProof::prove (:740-849) */
  tr_append_protocol_name(tr, "Sparse polynomial hash layer proof");
  fq_t *eq_ops = fq_alloc(N), *eq_mem = fq_alloc(M);
  eq_evals_any(rand_ops, lgN, eq_ops);
  eq_evals_any(rand_mem, lgM, eq_mem);
  for (int m = 0; m < 3; m++) {
    ev.hl_deref_row[m] = oracle_dotproduct(drow[m], eq_ops, N);
    ev.hl_deref_col[m] = oracle_dotproduct(dcol[m], eq_ops, N);
  }
  dplog_t pe_derefs, pe_ops, pe_mem;
  {
    /* DerefsEvalProof::prove (:137-158, :90-135) */
    tr_append_protocol_name(tr, "Derefs evaluation proof");
    fq_t e8[8], ch[3], rj[64];
    for (int m = 0; m < 3; m++) { e8[m] = ev.hl_deref_row[m]; e8[3 + m] = ev.hl_deref_col[m]; }
    e8[6] = e8[7] = fq_zero();
    tr_append_scalars(tr, "evals_ops_val", e8, 8);
    tr_challenge_vector(tr, "challenge_combine_n_to_one", ch, 3);
    fq_t joint = combine_bot(e8, 8, ch, 3);
    memcpy(rj, ch, sizeof ch); memcpy(rj + 3, rand_ops, sizeof(fq_t) * lgN);
    tr_append_scalar(tr, "joint_claim_eval", &joint);
    polyeval_prove_plain(&pe_derefs, comb, rj, &joint, &sg->derefs, tr, tape);
  }
  for (int m = 0; m < 3; m++) {
    ev.hl_row_addr[m] = eval_u32(d->row[m], N, eq_ops);
    ev.hl_row_ts[m] = eval_u32(d->row_ts[m], N, eq_ops);
    ev.hl_col_addr[m] = eval_u32(d->col[m], N, eq_ops);
    ev.hl_col_ts[m] = eval_u32(d->col_ts[m], N, eq_ops);
    ev.hl_val[m] = oracle_dotproduct(d->val[m], eq_ops, N);
  }
  ev.hl_row_audit = eval_u32(d->row_audit, M, eq_mem);
  ev.hl_col_audit = eval_u32(d->col_audit, M, eq_mem);
  {
    fq_t e16[16], ch[4], rj[64];
    for (int m = 0; m < 3; m++) {
      e16[m] = ev.hl_row_addr[m]; e16[3 + m] = ev.hl_row_ts[m]; e16[6 + m] = ev.hl_col_addr[m];
      e16[9 + m] = ev.hl_col_ts[m]; e16[12 + m] = ev.hl_val[m];
    }
    e16[15] = fq_zero();
    tr_append_scalars(tr, "claim_evals_ops", e16, 16);
    tr_challenge_vector(tr, "challenge_combine_n_to_one", ch, 4);
    fq_t joint = combine_bot(e16, 16, ch, 4);
    memcpy(rj, ch, sizeof ch); memcpy(rj + 4, rand_ops, sizeof(fq_t) * lgN);
    tr_append_scalar(tr, "joint_claim_eval_ops", &joint);
    polyeval_prove_plain(&pe_ops, d->comb_ops, rj, &joint, &sg->ops, tr, tape);
  }
  {
    fq_t e2[2] = {ev.hl_row_audit, ev.hl_col_audit}, ch[1], rj[64];
    tr_append_scalars(tr, "claim_evals_mem", e2, 2);
    tr_challenge_vector(tr, "challenge_combine_two_to_one", ch, 1);
    fq_t joint = combine_bot(e2, 2, ch, 1);
    rj[0] = ch[0]; memcpy(rj + 1, rand_mem, sizeof(fq_t) * lgM);
    tr_append_scalar(tr, "joint_claim_eval_mem", &joint);
    polyeval_prove_plain(&pe_mem, d->comb_mem, rj, &joint, &sg->mem, tr, tape);
  }
  g_spark_timings[5] = now_s() - t0;

  /* bincode(R1CSEvalProof { proof: SparseMatPolyEvalProof { comm_derefs, poly_eval_network_proof {
   *   proof_prod_layer, proof_hash_layer } } }) */
  w_u64(w, sg->derefs.L);
  for (size_t i = 0; i < sg->derefs.L; i++) w_point(w, &comm_derefs[i]);
  const fq_t *plc[2] = {ev.pl_row, ev.pl_col};
  for (int s = 0; s < 2; s++) { w_scalar(w, &plc[s][0]); w_scalars(w, plc[s] + 1, 3); w_scalars(w, plc[s] + 4, 3); w_scalar(w, &plc[s][7]); }
  w_scalars(w, ev.dotp_left, 3); w_scalars(w, ev.dotp_right, 3);
  batched_write(w, &pf_mem);
  batched_write(w, &pf_ops);
  w_scalars(w, ev.hl_row_addr, 3); w_scalars(w, ev.hl_row_ts, 3); w_scalar(w, &ev.hl_row_audit);
  w_scalars(w, ev.hl_col_addr, 3); w_scalars(w, ev.hl_col_ts, 3); w_scalar(w, &ev.hl_col_audit);
  w_scalars(w, ev.hl_val, 3);
  w_scalars(w, ev.hl_deref_row, 3); w_scalars(w, ev.hl_deref_col, 3);
  dplog_write(w, &pe_ops); dplog_write(w, &pe_mem); dplog_write(w, &pe_derefs);

  batched_free(&pf_ops); batched_free(&pf_mem);
  dplog_free(&pe_ops); dplog_free(&pe_mem); dplog_free(&pe_derefs);
  free(comm_derefs); free(comb); free(mem_rx); free(mem_ry); free(eq_ops); free(eq_mem);
  return ok;
}

size_t oracle_vpin_snark_prove(const r1cs_t *inst, const spark_decomm_t *decomm, const fq_t *vars_para,
                               const fq_t *vars_input, const fq_t *vars, const fq_t *inputs,
                               const uint8_t seed_commit64[64], const uint8_t seed_proof64[64], int threads,
                               uint8_t *proof_out, size_t proof_cap, uint8_t *comm_para, uint8_t *comm_input) {
  double t_start = now_s();
  merlin_t tr, tape;
  fq_t inst_evals[3], rx[64], ry[64];
  size_t sat_len = oracle_sat_prove_core(inst, vars_para, vars_input, vars, inputs, seed_commit64, seed_proof64, threads,
                                         proof_out, proof_cap, comm_para, comm_input, inst_evals, rx, ry, &tr, &tape);
  g_spark_timings[1] = now_s() - t_start;
  if (!sat_len) return 0;
  wbuf w = {proof_out, sat_len, proof_cap, 0};
  for (int i = 0; i < 3; i++) w_scalar(&w, &inst_evals[i]); /* SNARK.inst_evals (lib.rs:336) */
  sparkgens_t sg;
  sparkgens_new(&sg, decomm->nx, decomm->ny, decomm->N);
  int ok = spark_prove(&w, decomm, rx, ry, inst_evals, &sg, &tr, &tape, threads);
  sparkgens_free(&sg);
  g_spark_timings[6] = now_s() - t_start;
  if (!ok || w.bad) return 0;
  return w.len;
}

/* ------------------------------------------------------------------ verifier */

/* HashLayerProof::verify_helper (sparse_mlpoly.rs:851-900) */
static int hash_verify_helper(const fq_t *rand_mem, size_t lgM, const fq_t *claim_init, const fq_t *claim_read,
                              const fq_t *claim_write, const fq_t *claim_audit, const fq_t *eval_ops_val,
                              const fq_t *eval_ops_addr, const fq_t *eval_read_ts, const fq_t *eval_audit_ts,
                              const fq_t *r, const fq_t *r_hash, const fq_t *gamma) {
  fq_t r2 = F_mul(*r_hash, *r_hash), one = fq_one();
  /* IdentityPolynomial::evaluate (dense_mlpoly.rs:121-127) */
  fq_t addr = fq_zero();
  for (size_t i = 0; i < lgM; i++) addr = F_add(addr, F_mul(fq_from_u64((uint64_t)1 << (lgM - i - 1)), rand_mem[i]));
  /* EqPolynomial::new(r).evaluate(rand_mem) (dense_mlpoly.rs:58-66) */
  fq_t val = fq_one();
  for (size_t i = 0; i < lgM; i++)
    val = F_mul(val, F_add(F_mul(r[i], rand_mem[i]), F_mul(F_sub(one, r[i]), F_sub(one, rand_mem[i]))));
  fq_t hinit = F_sub(F_add(F_mul(val, *r_hash), addr), *gamma);
  if (!fq_eq(&hinit, claim_init)) return 0;
  for (int i = 0; i < 3; i++) {
    fq_t hr = F_sub(F_add(F_add(F_mul(eval_read_ts[i], r2), F_mul(eval_ops_val[i], *r_hash)), eval_ops_addr[i]), *gamma);
    if (!fq_eq(&hr, &claim_read[i])) return 0;
  }
  for (int i = 0; i < 3; i++) {
    fq_t wts = F_add(eval_read_ts[i], one);
    fq_t hw = F_sub(F_add(F_add(F_mul(wts, r2), F_mul(eval_ops_val[i], *r_hash)), eval_ops_addr[i]), *gamma);
    if (!fq_eq(&hw, &claim_write[i])) return 0;
  }
  fq_t haudit = F_sub(F_add(F_add(F_mul(*eval_audit_ts, r2), F_mul(val, *r_hash)), addr), *gamma);
  return fq_eq(&haudit, claim_audit);
}

int oracle_vpin_snark_verify(const uint8_t *proof, size_t proof_len, const uint8_t *comm, size_t comm_len,
                             const fq_t *inputs, size_t num_inputs, const uint8_t *comm_para, const uint8_t *comm_input) {
  /* R1CSCommitment */
  rbuf rc = {comm, comm_len, 0, 0};
  size_t num_cons = r_u64(&rc), num_vars = r_u64(&rc), n_in = r_u64(&rc);
  size_t batch = r_u64(&rc), N = r_u64(&rc), M = r_u64(&rc);
  if (rc.bad || batch != 3 || n_in != num_inputs || !N || !M || (N & (N - 1)) || (M & (M - 1)) || N > ((size_t)1 << 40)) return 0;
  size_t nx = log2z(num_cons), ny = log2z(2 * num_vars);
  if (M != ((size_t)1 << (nx > ny ? nx : ny))) return 0;
  sparkgens_t sg;
  sparkgens_new(&sg, nx, ny, N);
  size_t lgN = log2z(N), lgM = log2z(M);
  int ok = 1;
  cg_t *c_ops = NULL, *c_mem = NULL, *c_derefs = NULL;
  batched_t pf_ops, pf_mem;
  dplog_t pe_ops = {0}, pe_mem = {0}, pe_derefs = {0};
  memset(&pf_ops, 0, sizeof pf_ops); memset(&pf_mem, 0, sizeof pf_mem);
  if (r_u64(&rc) != sg.ops.L) { ok = 0; goto done; }
  c_ops = (cg_t *)malloc(sizeof(cg_t) * sg.ops.L);
  for (size_t i = 0; i < sg.ops.L; i++) c_ops[i] = r_point(&rc);
  if (r_u64(&rc) != sg.mem.L) { ok = 0; goto done; }
  c_mem = (cg_t *)malloc(sizeof(cg_t) * sg.mem.L);
  for (size_t i = 0; i < sg.mem.L; i++) c_mem[i] = r_point(&rc);
  if (rc.bad || rc.pos != rc.len) { ok = 0; goto done; }

  /* sat part + inst_evals (commit_test.rs:498-524) */
  merlin_t tr;
  fq_t inst_evals[3], rx[64], ry[64];
  size_t used = 0;
  if (!oracle_sat_verify_core(proof, proof_len, num_cons, num_vars, inputs, num_inputs, NULL, inst_evals, comm_para,
                              comm_input, rx, ry, &used, &tr)) { ok = 0; goto done; }
  rbuf r = {proof, proof_len, used, 0};

  /* parse R1CSEvalProof */
  if (r_u64(&r) != sg.derefs.L) { ok = 0; goto done; }
  c_derefs = (cg_t *)malloc(sizeof(cg_t) * sg.derefs.L);
  for (size_t i = 0; i < sg.derefs.L; i++) c_derefs[i] = r_point(&r);
  spark_evals_t ev;
  fq_t *plv[2] = {ev.pl_row, ev.pl_col};
  for (int s = 0; s < 2; s++) {
    plv[s][0] = r_scalar(&r);
    if (!r_scalars(&r, plv[s] + 1, 3) || !r_scalars(&r, plv[s] + 4, 3)) { ok = 0; goto done; }
    plv[s][7] = r_scalar(&r);
  }
  if (!r_scalars(&r, ev.dotp_left, 3) || !r_scalars(&r, ev.dotp_right, 3)) { ok = 0; goto done; }
  if (!batched_read(&r, &pf_mem, (int)lgM, 4, 0)) { ok = 0; goto done; }
  if (!batched_read(&r, &pf_ops, (int)lgN, 12, 6)) { ok = 0; goto done; }
  if (!r_scalars(&r, ev.hl_row_addr, 3) || !r_scalars(&r, ev.hl_row_ts, 3)) { ok = 0; goto done; }
  ev.hl_row_audit = r_scalar(&r);
  if (!r_scalars(&r, ev.hl_col_addr, 3) || !r_scalars(&r, ev.hl_col_ts, 3)) { ok = 0; goto done; }
  ev.hl_col_audit = r_scalar(&r);
  if (!r_scalars(&r, ev.hl_val, 3) || !r_scalars(&r, ev.hl_deref_row, 3) || !r_scalars(&r, ev.hl_deref_col, 3)) { ok = 0; goto done; }
  if (!dplog_read(&r, &pe_ops) || !dplog_read(&r, &pe_mem) || !dplog_read(&r, &pe_derefs)) { ok = 0; goto done; }
  if (r.bad || r.pos != r.len) { ok = 0; goto done; }

  /* SparseMatPolyEvalProof::verify (:1535-1571) */
  tr_append_protocol_name(&tr, "Sparse polynomial evaluation proof");
  fq_t rx_ext[64], ry_ext[64];
  size_t nm;
  equalize(rx, nx, ry, ny, rx_ext, ry_ext, &nm);
  if (((size_t)1 << nm) != M) { ok = 0; goto done; }
  merlin_append_message(&tr, "derefs_commitment", (const uint8_t *)"begin_derefs_commitment", 23);
  tr_append_polycomm(&tr, "comm_poly_row_col_ops_val", c_derefs, sg.derefs.L);
  merlin_append_message(&tr, "derefs_commitment", (const uint8_t *)"end_derefs_commitment", 21);
  fq_t r_mem_check[2];
  tr_challenge_vector(&tr, "challenge_r_hash", r_mem_check, 2);
  /* PolyEvalNetworkProof::verify (:1372-1434) */
  tr_append_protocol_name(&tr, "Sparse polynomial evaluation proof");
  /* ProductLayerProof::verify (:1229-1322) */
  tr_append_protocol_name(&tr, "Sparse polynomial product layer proof");
  static const char *lab[2][4] = {{"claim_row_eval_init", "claim_row_eval_read", "claim_row_eval_write", "claim_row_eval_audit"},
                                  {"claim_col_eval_init", "claim_col_eval_read", "claim_col_eval_write", "claim_col_eval_audit"}};
  for (int s = 0; s < 2; s++) {
    fq_t *p = plv[s], ws = fq_one(), rs = fq_one();
    for (int m = 0; m < 3; m++) { rs = F_mul(rs, p[1 + m]); ws = F_mul(ws, p[4 + m]); }
    fq_t lhs = F_mul(p[0], ws), rhs = F_mul(rs, p[7]);
    if (!fq_eq(&lhs, &rhs)) { ok = 0; goto done; }
    tr_append_scalar(&tr, lab[s][0], &p[0]);
    tr_append_scalars(&tr, lab[s][1], p + 1, 3);
    tr_append_scalars(&tr, lab[s][2], p + 4, 3);
    tr_append_scalar(&tr, lab[s][3], &p[7]);
  }
  fq_t claims_dotp_circuit[6];
  for (int m = 0; m < 3; m++) {
    fq_t sum = F_add(ev.dotp_left[m], ev.dotp_right[m]);
    if (!fq_eq(&sum, &inst_evals[m])) { ok = 0; goto done; }
    tr_append_scalar(&tr, "claim_eval_dotp_left", &ev.dotp_left[m]);
    tr_append_scalar(&tr, "claim_eval_dotp_right", &ev.dotp_right[m]);
    claims_dotp_circuit[2 * m] = ev.dotp_left[m]; claims_dotp_circuit[2 * m + 1] = ev.dotp_right[m];
  }
  fq_t claims_prod_circuit[12];
  memcpy(claims_prod_circuit, ev.pl_row + 1, sizeof(fq_t) * 6);
  memcpy(claims_prod_circuit + 6, ev.pl_col + 1, sizeof(fq_t) * 6);
  fq_t claims_ops[12], claims_dotp[9], rand_ops[64], claims_mem[4], rand_mem[64];
  if (!batched_verify(&pf_ops, claims_prod_circuit, claims_dotp_circuit, &tr, claims_ops, claims_dotp, rand_ops)) { ok = 0; goto done; }
  fq_t mem_claims_in[4] = {ev.pl_row[0], ev.pl_row[7], ev.pl_col[0], ev.pl_col[7]};
  if (!batched_verify(&pf_mem, mem_claims_in, NULL, &tr, claims_mem, NULL, rand_mem)) { ok = 0; goto done; }

  /* HashLayerProof::verify (:902-1032) */
  tr_append_protocol_name(&tr, "Sparse polynomial hash layer proof");
  {
    tr_append_protocol_name(&tr, "Derefs evaluation proof");
    fq_t e8[8], ch[3], rj[64];
    for (int m = 0; m < 3; m++) { e8[m] = ev.hl_deref_row[m]; e8[3 + m] = ev.hl_deref_col[m]; }
    e8[6] = e8[7] = fq_zero();
    tr_append_scalars(&tr, "evals_ops_val", e8, 8);
    tr_challenge_vector(&tr, "challenge_combine_n_to_one", ch, 3);
    fq_t joint = combine_bot(e8, 8, ch, 3);
    memcpy(rj, ch, sizeof ch); memcpy(rj + 3, rand_ops, sizeof(fq_t) * lgN);
    tr_append_scalar(&tr, "joint_claim_eval", &joint);
    if (!polyeval_verify_plain(&pe_derefs, rj, &joint, c_derefs, sg.derefs.L, &sg.derefs, &tr)) { ok = 0; goto done; }
  }
  for (int m = 0; m < 3; m++) {
    if (!fq_eq(&claims_dotp[3 * m], &ev.hl_deref_row[m]) || !fq_eq(&claims_dotp[3 * m + 1], &ev.hl_deref_col[m]) ||
        !fq_eq(&claims_dotp[3 * m + 2], &ev.hl_val[m])) { ok = 0; goto done; }
  }
  {
    fq_t e16[16], ch[4], rj[64];
    for (int m = 0; m < 3; m++) {
      e16[m] = ev.hl_row_addr[m]; e16[3 + m] = ev.hl_row_ts[m]; e16[6 + m] = ev.hl_col_addr[m];
      e16[9 + m] = ev.hl_col_ts[m]; e16[12 + m] = ev.hl_val[m];
    }
    e16[15] = fq_zero();
    tr_append_scalars(&tr, "claim_evals_ops", e16, 16);
    tr_challenge_vector(&tr, "challenge_combine_n_to_one", ch, 4);
    fq_t joint = combine_bot(e16, 16, ch, 4);
    memcpy(rj, ch, sizeof ch); memcpy(rj + 4, rand_ops, sizeof(fq_t) * lgN);
    tr_append_scalar(&tr, "joint_claim_eval_ops", &joint);
    if (!polyeval_verify_plain(&pe_ops, rj, &joint, c_ops, sg.ops.L, &sg.ops, &tr)) { ok = 0; goto done; }
  }
  {
    fq_t e2[2] = {ev.hl_row_audit, ev.hl_col_audit}, ch[1], rj[64];
    tr_append_scalars(&tr, "claim_evals_mem", e2, 2);
    tr_challenge_vector(&tr, "challenge_combine_two_to_one", ch, 1);
    fq_t joint = combine_bot(e2, 2, ch, 1);
    rj[0] = ch[0]; memcpy(rj + 1, rand_mem, sizeof(fq_t) * lgM);
    tr_append_scalar(&tr, "joint_claim_eval_mem", &joint);
    if (!polyeval_verify_plain(&pe_mem, rj, &joint, c_mem, sg.mem.L, &sg.mem, &tr)) { ok = 0; goto done; }
  }
  /* claims_ops = row_read(3) row_write(3) col_read(3) col_write(3); claims_mem = row init/audit, col init/audit */
  if (!hash_verify_helper(rand_mem, lgM, &claims_mem[0], claims_ops, claims_ops + 3, &claims_mem[1], ev.hl_deref_row,
                          ev.hl_row_addr, ev.hl_row_ts, &ev.hl_row_audit, rx_ext, &r_mem_check[0], &r_mem_check[1])) { ok = 0; goto done; }
  if (!hash_verify_helper(rand_mem, lgM, &claims_mem[2], claims_ops + 6, claims_ops + 9, &claims_mem[3], ev.hl_deref_col,
                          ev.hl_col_addr, ev.hl_col_ts, &ev.hl_col_audit, ry_ext, &r_mem_check[0], &r_mem_check[1])) { ok = 0; goto done; }
done:
  if (pf_ops.rounds) batched_free(&pf_ops);
  if (pf_mem.rounds) batched_free(&pf_mem);
  if (pe_ops.Lv) dplog_free(&pe_ops);
  if (pe_mem.Lv) dplog_free(&pe_mem);
  if (pe_derefs.Lv) dplog_free(&pe_derefs);
  free(c_ops); free(c_mem); free(c_derefs);
  sparkgens_free(&sg);
  return ok;
}
