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

#include "proto_common.h"

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

size_t oracle_sat_prove_core(const r1cs_t *inst, const fq_t *vars_para, const fq_t *vars_input,
                             const fq_t *vars, const fq_t *inputs,
                             const uint8_t seed_commit64[64], const uint8_t seed_proof64[64], int threads,
                             uint8_t *proof_out, size_t proof_cap, uint8_t *comm_para_out, uint8_t *comm_input_out,
                             fq_t inst_evals[3], fq_t *rx, fq_t *ry, merlin_t *tr_out, merlin_t *tape_out) {
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
  if (tr_out) *tr_out = tr;
  if (tape_out) *tape_out = tape;
  if (!ok || w.bad) return 0;
  return w.len;
}

size_t oracle_vpin_sat_prove(const r1cs_t *inst, const fq_t *vars_para, const fq_t *vars_input,
                             const fq_t *vars, const fq_t *inputs,
                             const uint8_t seed_commit64[64], const uint8_t seed_proof64[64], int threads,
                             uint8_t *proof_out, size_t proof_cap, uint8_t *comm_para_out, uint8_t *comm_input_out,
                             fq_t inst_evals[3], fq_t *rx, fq_t *ry) {
  return oracle_sat_prove_core(inst, vars_para, vars_input, vars, inputs, seed_commit64, seed_proof64, threads,
                               proof_out, proof_cap, comm_para_out, comm_input_out, inst_evals, rx, ry, NULL, NULL);
}

int oracle_sat_verify_core(const uint8_t *proof, size_t proof_len, size_t num_cons, size_t num_vars,
                           const fq_t *inputs, size_t num_inputs, const fq_t *inst_evals_in, fq_t inst_evals_out[3],
                           const uint8_t *comm_para, const uint8_t *comm_input, fq_t *rx, fq_t *ry,
                           size_t *consumed, merlin_t *tr_out) {
  /* inst_evals_in == NULL: the bytes are a whole SNARK (lib.rs:334-338) and inst_evals follow the
   * R1CSProof; *consumed then covers both.  Otherwise the bytes are the R1CSProof alone. */
  fq_t inst_evals[3];
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
  if (inst_evals_in) memcpy(inst_evals, inst_evals_in, sizeof inst_evals);
  else for (int i = 0; i < 3; i++) inst_evals[i] = r_scalar(&r);
  if (inst_evals_out) memcpy(inst_evals_out, inst_evals, sizeof inst_evals);
  if (r.bad || (!consumed && r.pos != r.len)) { ok = 0; goto done; }
  if (consumed) *consumed = r.pos;

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
  /* my_lib_verify (commit_test.rs:521-524) */
  tr_append_scalar(&tr, "Ar_claim", &inst_evals[0]);
  tr_append_scalar(&tr, "Br_claim", &inst_evals[1]);
  tr_append_scalar(&tr, "Cr_claim", &inst_evals[2]);
  if (tr_out) *tr_out = tr;
done:
  if (sc1.comm_polys) zksc_free(&sc1);
  if (sc2.comm_polys) zksc_free(&sc2);
  if (pe.Lv) dplog_free(&pe);
  free(comm_vars); free(combined);
  satgens_free(&sg);
  return ok;
}

int oracle_vpin_sat_verify(const uint8_t *proof, size_t proof_len, size_t num_cons, size_t num_vars,
                           const fq_t *inputs, size_t num_inputs, const fq_t inst_evals[3],
                           const uint8_t *comm_para, const uint8_t *comm_input, fq_t *rx, fq_t *ry) {
  return oracle_sat_verify_core(proof, proof_len, num_cons, num_vars, inputs, num_inputs, inst_evals, NULL,
                                comm_para, comm_input, rx, ry, NULL, NULL);
}
