/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see fq.h).
 * sat.h : CPU restatement of vPIN's R1CS satisfiability proof (the Spartan "sat proof")
 * exactly as vPIN's forked prover glue drives it:
 *   vPIN_proof_generation/src/commit_test.rs:27-57   my_dense_mlpoly_commit
 *   vPIN_proof_generation/src/commit_test.rs:59-133  my_lib_prove   (sat part + inst_evals)
 *   vPIN_proof_generation/src/commit_test.rs:136-334 my_R1CSProof_prove
 *   vPIN_proof_generation/src/commit_test.rs:340-496 my_r1csproof_verify
 *   vPIN_proof_generation/src/proof_point_mult.rs:23-101 (commit para / input, combine)
 * on top of Spartan/src/{sumcheck.rs:428-776, nizk/mod.rs, nizk/bullet.rs,
 * dense_mlpoly.rs:193-218,326-379, r1csproof.rs:49-155, r1csinstance.rs:240-302,
 * sparse_mlpoly.rs:440-498}.
 *
 * The reference draws its prover randomness from OsRng (Spartan/src/random.rs:14-22), so
 * proofs are only reproducible given the two 64-byte draws; both are explicit inputs here.
 * Proof bytes follow bincode 1.3.3 defaults.
 *
 * Parity status: the reference holds no golden proof and its Rust toolchain is absent, so
 * the proof BYTES are "parity unpinned"; what is pinned: every primitive underneath
 * (F_q, ristretto255, SHAKE256, Merlin) against reference/public vectors, and the
 * protocol by the restated verifier accepting the restated prover.
 */
#ifndef VPIN_ORACLE_SAT_H
#define VPIN_ORACLE_SAT_H
#include "group.h"
#include "keccak.h"
#include "poly.h"

#ifdef __cplusplus
extern "C" {
#endif

/* R1CSInstance after Instance::new's padding and column remap (lib.rs:138-244):
 * num_cons, num_vars are powers of two; cols index z = [vars | 1 | inputs | 0...] of
 * length 2*num_vars. */
typedef struct {
  size_t num_cons, num_vars, num_inputs;
  size_t nnz[3];           /* A, B, C */
  const uint32_t *row[3];
  const uint32_t *col[3];
  const fq_t *val[3];
} r1cs_t;

/* SparseMatPolynomial::multiply_vec x3 (sparse_mlpoly.rs:467-481) */
void oracle_r1cs_multiply_vec(const r1cs_t *inst, const fq_t *z, fq_t *Az, fq_t *Bz, fq_t *Cz);
/* compute_eval_table_sparse x3 (sparse_mlpoly.rs:483-498); outputs have 2*num_vars entries */
void oracle_r1cs_eval_table_sparse(const r1cs_t *inst, const fq_t *evals_rx, fq_t *eA, fq_t *eB, fq_t *eC);
/* R1CSInstance::is_sat (r1csinstance.rs:240-270). vars has num_vars entries. */
int oracle_r1cs_is_sat(const r1cs_t *inst, const fq_t *vars, const fq_t *inputs);
/* R1CSInstance::evaluate (r1csinstance.rs:297-302) */
void oracle_r1cs_evaluate(const r1cs_t *inst, const fq_t *rx, const fq_t *ry, fq_t out[3]);

/* Byte budget for a sat proof of this shape */
size_t oracle_sat_proof_max_bytes(size_t num_cons, size_t num_vars);

/*
 * The whole vPIN sat-proof flow for one gadget instance
 * (proof_point_mult.rs:38-94 minus SNARK::encode / R1CSEvalProof):
 *   tape1 = RandomTape::new([2]) seeded by seed_commit64  -> blinds of the para and input commits
 *   comm_para, comm_input = DensePolynomial::commit; combined = row-wise sum
 *   transcript = Transcript::new("snark_example"); my_lib_prove with tape2 = RandomTape::new("proof")
 *   seeded by seed_proof64; stops after appending Ar/Br/Cr claims.
 * vars_para / vars_input / vars are the three padded assignments (num_vars entries each).
 * Outputs: proof (bincode bytes of R1CSProof), comm_para / comm_input (L x 32 B each),
 * inst_evals (Ar,Br,Cr), rx (log2 num_cons), ry (log2 num_vars + 1).
 * Returns proof length, or 0 on an internal consistency failure (the reference would panic).
 */
size_t oracle_vpin_sat_prove(const r1cs_t *inst, const fq_t *vars_para, const fq_t *vars_input,
                             const fq_t *vars, const fq_t *inputs,
                             const uint8_t seed_commit64[64], const uint8_t seed_proof64[64],
                             int threads,
                             uint8_t *proof_out, size_t proof_cap,
                             uint8_t *comm_para, uint8_t *comm_input,
                             fq_t inst_evals[3], fq_t *rx, fq_t *ry);

/* my_lib_verify's sat part (commit_test.rs:340-496, 498-530): 1 = accept */
int oracle_vpin_sat_verify(const uint8_t *proof, size_t proof_len,
                           size_t num_cons, size_t num_vars,
                           const fq_t *inputs, size_t num_inputs, const fq_t inst_evals[3],
                           const uint8_t *comm_para, const uint8_t *comm_input,
                           fq_t *rx_out, fq_t *ry_out);

/* timing spans of the last oracle_vpin_sat_prove call, seconds:
 * [0] polycommit (para+input), [1] prove_sc_phase_one (incl. eq table + SpMV),
 * [2] prove_sc_phase_two (incl. sparse eval table), [3] polyeval, [4] total */
void oracle_sat_last_timings(double out[5]);

#ifdef __cplusplus
}
#endif
#endif
