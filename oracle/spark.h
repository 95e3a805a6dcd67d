/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see fq.h).
 * spark.h : CPU restatement of the SPARK half of vPIN's Spartan SNARK -- the computation
 * commitment (`SNARK::encode`) and the sparse-polynomial evaluation proof
 * (`R1CSEvalProof::prove` / `::verify`) -- as my_lib_prove / my_lib_verify drive them:
 *   Spartan/src/lib.rs:294-359                 SNARKGens::new, SNARK::encode
 *   Spartan/src/r1csinstance.rs:29-49,309-372  R1CSCommitmentGens, commit, R1CSEvalProof
 *   Spartan/src/sparse_mlpoly.rs:42-1572       Derefs, AddrTimestamps, multi_commit, Layers,
 *                                              HashLayerProof, ProductLayerProof,
 *                                              PolyEvalNetworkProof, SparseMatPolyEvalProof
 *   Spartan/src/product_tree.rs:18-486         ProductCircuit, DotProductCircuit,
 *                                              ProductCircuitEvalProofBatched
 *   Spartan/src/sumcheck.rs:22-61,182-425      SumcheckInstanceProof::verify, prove_cubic_batched
 *   Spartan/src/dense_mlpoly.rs:193-218,326-419 commit (no tape), PolyEvalProof prove/verify_plain
 *   vPIN_proof_generation/src/commit_test.rs:59-133,498-548  my_lib_prove / my_lib_verify
 *
 * The SNARK bytes are bincode 1.3.3 of `SNARK { r1cs_sat_proof, inst_evals, r1cs_eval_proof }`
 * (lib.rs:334-338).  Parity status: as for sat.h -- bytes "parity unpinned" (no Rust toolchain,
 * no golden proof in the reference); pinned by primitives + restated verifier.
 */
#ifndef VPIN_ORACLE_SPARK_H
#define VPIN_ORACLE_SPARK_H
#include "sat.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ComputationDecommitment: MultiSparseMatPolynomialAsDense (sparse_mlpoly.rs:285-292) */
typedef struct spark_decomm spark_decomm_t;

/* bincode size of R1CSCommitment (r1csinstance.rs:53-58) for this instance */
size_t oracle_spark_comm_bytes(const r1cs_t *inst);
/* serialised SNARK upper bound */
size_t oracle_snark_proof_max_bytes(const r1cs_t *inst);

/* SNARK::encode (lib.rs:347-359): builds the dense representation and commits comb_ops /
 * comb_mem.  comm_out receives bincode(R1CSCommitment).  Generators are sized from the
 * instance's own max nnz (the reference sizes them from a hand-tuned num_non_zero_entries that
 * must round to the same power of two or its commit asserts). */
spark_decomm_t *oracle_spark_encode(const r1cs_t *inst, int threads, uint8_t *comm_out, size_t comm_cap,
                                    size_t *comm_len);
void oracle_spark_decomm_free(spark_decomm_t *d);

/* proof_point_mult.rs:38-94 in full: sat proof (sat.h) continued by inst_evals and
 * R1CSEvalProof::prove on the same transcript and random tape.  Returns the SNARK length or 0. */
size_t oracle_vpin_snark_prove(const r1cs_t *inst, const spark_decomm_t *decomm,
                               const fq_t *vars_para, const fq_t *vars_input, const fq_t *vars,
                               const fq_t *inputs, const uint8_t seed_commit64[64],
                               const uint8_t seed_proof64[64], int threads,
                               uint8_t *proof_out, size_t proof_cap,
                               uint8_t *comm_para, uint8_t *comm_input);

/* my_lib_verify (commit_test.rs:498-548): 1 = accept.  comm = bincode(R1CSCommitment). */
int oracle_vpin_snark_verify(const uint8_t *proof, size_t proof_len, const uint8_t *comm, size_t comm_len,
                             const fq_t *inputs, size_t num_inputs,
                             const uint8_t *comm_para, const uint8_t *comm_input);

/* timing spans of the last encode / snark_prove, seconds:
 * [0] encode, [1] sat part, [2] derefs+commit, [3] network build, [4] product-layer proofs,
 * [5] hash-layer proofs, [6] total prove */
void oracle_spark_last_timings(double out[7]);

#ifdef __cplusplus
}
#endif
#endif
