// spark_dev.h -- device-side state and launchers of the SPARK half (spark.hip), driven by spark.cpp
#pragma once
#include <mutex>

#include "ctx.h"

// ComputationDecommitment (lib.rs:70-73): MultiSparseMatPolynomialAsDense (sparse_mlpoly.rs:285-292)
// resident in HBM.  Index / timestamp vectors stay as u32 next to their field-element images
// inside comb_ops / comb_mem.
struct vpin_spark_decomm {
  size_t num_cons = 0, num_vars = 0, num_inputs = 0;
  size_t nx = 0, ny = 0, N = 0, M = 0;
  // u32 [row A,B,C | row_read_ts A,B,C | col A,B,C | col_read_ts A,B,C] (12 x N, the order of
  // comb_ops), then row_audit_ts (M), col_audit_ts (M)
  uint32_t* idx = nullptr;
  vpin::fq* vals = nullptr;        // 3N: val A,B,C (slices 12..14 of comb_ops), the only part of the two combined polynomials
                                   // that is not a u32
  // The combined polynomials as field elements (16N: idx as Scalar::from, vals, a zero slice; 2M: the audit timestamps) exist
  // only while something needs them whole: SNARK::encode's commitment, the two-pass and the multi-GPU hash layer
  // (spark_comb_tables below).  Round 5: they used to stay resident next to idx -- 17.2 + 2.1 GB for the 2^25 instance.
  vpin_table* comb_ops = nullptr;
  vpin_table* comb_mem = nullptr;
  std::mutex comb_mu;  // a proof split over several GPUs builds them on first use and leaves them (spark_comb_tables)
  bool comb_unpooled = false;  // the cached copies come straight from the driver (they outlive whichever context built them)
  // the column that carries a large share of matrix m's entries (the constant 1 in B and C of vPIN's gadgets), found once by
  // SNARK::encode; 0xffffffff: none.  The derefs commitment of every proof takes its entries out of the table walks.
  uint32_t hot_col[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
};

namespace vpin {

constexpr size_t kSparkPinned = 8192;  // fq elements of pinned staging (ctx->h_spark)
constexpr int kSparkMaxInst = 18;      // 12 product circuits + 6 dot-product circuits
constexpr size_t kSparkHostTop = 32;   // tree levels of at most this many entries are proven on the host

int spark_pinned(vpin_ctx* c);  // allocate ctx->h_spark on first use

// trace.hip: read / audit time stamps of one side of SNARK::encode's memory trace (AddrTimestamps::new) from its n = 3N addresses
int spark_trace_timestamps(vpin_ctx* c, const uint32_t* addr, size_t n, size_t M, uint32_t* ts, uint32_t* audit);
int spark_check_bounds(vpin_ctx* c, const uint32_t* a, size_t nnz, uint32_t limit, uint32_t* d_bad);

// dst[i] = Scalar::from(src[i]) (Montgomery form)
int spark_u32_to_fq(vpin_ctx* c, const uint32_t* src, fq* dst, size_t n);
// d->comb_ops / d->comb_mem built from idx and vals (no-op when they exist; serialised on d->comb_mu and complete on return, so
// several contexts may call it for one decommitment); spark_comb_release frees them again
int spark_comb_tables(vpin_ctx* c, vpin_spark_decomm* d);
// the same as two tables of the caller's (a proof that needs them whole must not touch the shared decommitment: other
// contexts may be proving from it)
int spark_comb_make(vpin_ctx* c, const vpin_spark_decomm* d, vpin_table** ops, vpin_table** mem, bool pooled = true);
void spark_comb_release(vpin_ctx* c, vpin_spark_decomm* d, bool to_driver = false);

// Derefs (sparse_mlpoly.rs:267-283,525-531): comb[m*N+i] = mem_rx[row_m[i]], comb[(3+m)*N+i] =
// mem_ry[col_m[i]], comb[6N..8N) = 0
int spark_gather_derefs(vpin_ctx* c, const vpin_spark_decomm* d, const fq* mem_rx, const fq* mem_ry, fq* comb);
// fills d->hot_col: candidates are the constant-1 column (num_vars) and the first input (num_vars + 1)
int spark_find_hot_cols(vpin_ctx* c, vpin_spark_decomm* d);

// Product-circuit forest: `ncirc` trees of `n` leaves; tree t lives at base + t*2n, level l (n>>l
// entries) at offset 2n - (2n>>l) inside it.  ProductCircuit.left_vec[l] / right_vec[l] are the two
// halves of level l (product_tree.rs:18-56).
struct SparkForest {
  fq* base = nullptr;
  size_t n = 0;
  int ncirc = 0;
  size_t stride() const { return 2 * n; }
  size_t level_off(int l) const { return 2 * n - ((2 * n) >> l); }
};

// Layers::build_hash_layer (sparse_mlpoly.rs:547-622) for both sides, written as level 0 of the
// forests: ops = [row read A,B,C | row write A,B,C | col read A,B,C | col write A,B,C] (n = N),
// mem = [row init, row audit, col init, col audit] (n = M); then every upper level.
// r_hash_sqr = r_hash^2; r_hash_sqr_boost = r_hash^2 * R (its Montgomery image taken to Montgomery form
// once more), which lets the kernel multiply RAW u32 timestamps straight into Montgomery form.
int spark_build_forests(vpin_ctx* c, const vpin_spark_decomm* d, const fq* comb_derefs, const fq* mem_rx, const fq* mem_ry,
                        const uint8_t r_hash[32], const uint8_t r_hash_sqr[32], const uint8_t r_hash_sqr_boost[32],
                        const uint8_t gamma[32], SparkForest* ops, SparkForest* mem);

// Round 5: the ops forest alone, the mem circuits' roots without their trees (-> c->h_spark[2 * circuit], the layout of
// spark_fetch_tops(&mem, 2)), and the mem forest alone -- built after the ops forest is proven, into the memory it frees
int spark_build_forest_ops(vpin_ctx* c, const vpin_spark_decomm* d, const fq* comb_derefs, const uint8_t r_hash[32],
                           const uint8_t r_hash_sqr[32], const uint8_t r_hash_sqr_boost[32], const uint8_t gamma[32], SparkForest* ops);
int spark_mem_roots(vpin_ctx* c, const vpin_spark_decomm* d, const fq* mem_rx, const fq* mem_ry, const uint8_t r_hash[32],
                    const uint8_t r_hash_sqr[32], const uint8_t r_hash_sqr_boost[32], const uint8_t gamma[32]);
int spark_build_forest_mem(vpin_ctx* c, const vpin_spark_decomm* d, const fq* mem_rx, const fq* mem_ry, const uint8_t r_hash[32],
                           const uint8_t r_hash_sqr[32], const uint8_t r_hash_sqr_boost[32], const uint8_t gamma[32], SparkForest* mem);

// The same for a subset of the circuits: f holds ncirc trees, ids[j] = global circuit of tree j (ops: side*6 + kind*3 + m,
// kind 0 read / 1 write; mem: side*2 + kind, kind 0 init / 1 audit).  One proof over several GPUs: a rank's own circuits.
int spark_build_forest_sub(vpin_ctx* c, const vpin_spark_decomm* d, const fq* comb_derefs, const fq* mem_rx, const fq* mem_ry,
                           const uint8_t r_hash[32], const uint8_t r_hash_sqr[32], const uint8_t r_hash_sqr_boost[32],
                           const uint8_t gamma[32], SparkForest* f, const int* ids, bool is_mem);

// the last `cnt` entries of every tree (the levels of <= cnt/2 entries), to ctx->h_spark
// [tree][cnt]; synchronises
int spark_fetch_tops(vpin_ctx* c, const SparkForest* f, size_t cnt);

// One round of prove_cubic_batched (sumcheck.rs:273-302) for the `ncirc` product circuits of forest
// level `level`, eq-factored: returns per circuit the three sums  sum_i E[i] * (A_x B_x)[i], x = 0,2,3
// with A/B = left/right halves (live length len) and E the suffix table of this round (len/2 or,
// when r != nullptr and the tables are first folded with r, len/4 entries).  Results land in
// ctx->h_spark[3*t + k] once spark_wait_flag returns.  `len` is the live length BEFORE this call.
// Starts a new launch group; with_dotp announces that spark_dotp_round follows in the same group.
// lead = true: h_spark[3*t] = sum_i E[i]*(A_0 B_0)[i] and h_spark[3*t + 1] = sum_i E[i]*(dA dB)[i] (value at 0 and x^2
// coefficient of the quadratic; the host derives the values at 2 and 3 from the circuit's claim), third slot zero.
// ndotp: number of dot-product halves spark_dotp_round adds to the same launch group (0 = none; 6 on one GPU).
int spark_prod_round(vpin_ctx* c, const SparkForest* f, int level, size_t len, const fq* E, const uint8_t* r, int ndotp,
                     bool lead = false);

// Same for the 6 DotProductCircuit halves of layer 0 (comb = A*B*C, three foldable tables each,
// sumcheck.rs:304-330).  src tables: left = comb_derefs row slices, right = col slices, weight =
// comb_ops val slices, each cut in two halves of N/2 (sparse_mlpoly.rs:1103-1125).  The first fold
// writes into `scratch` (18 x N/4 entries) so the committed polynomials stay intact.
// round 0: r == nullptr, len = N/2.  Results at ctx->h_spark[3*(12+k) + x].  Call right after
// spark_prod_round(..., with_dotp = true).
// vals = the three val slices of comb_ops (3 x N), N = leaves of the ops forest.
// halves / ndotp: the subset of the six halves this call proves (one proof over several GPUs: a rank's own halves), results
// at slots 12, 13, .. in that order; nullptr / 6 = all of them.
int spark_dotp_round(vpin_ctx* c, size_t N, const fq* vals, const fq* comb_derefs, fq* scratch, size_t len, bool first_fold,
                     const uint8_t* r, const int* halves = nullptr, int ndotp = 6);

// completion of the current launch group: spins on the pinned flag word the last block publishes
int spark_wait_flag(vpin_ctx* c);

// after the last round: the two live entries of every table, for the host to bind with the final
// challenge: h_spark[4*t + {0,1,2,3}] = A[0], A[1], B[0], B[1] per product circuit and, with_dotp,
// h_spark[64 + 6*k + {0..5}] = L[0], L[1], R[0], R[1], W[0], W[1] per dot-product circuit (from scratch when
// `folded`, else from the source tables: N/2 == 2, no fold has happened); waits for completion
int spark_collect(vpin_ctx* c, const SparkForest* f, int level, const vpin_spark_decomm* d, const fq* comb_derefs,
                  const fq* scratch, bool with_dotp, bool folded);

// DotProductCircuit::evaluate (product_tree.rs:87-91) of the six halves: h_spark[3*k] = sum_i L[i]*R[i]*W[i]
// over the N/2 entries of half k; synchronises
// halves / nh: a subset of the halves (results at h_spark[3*i] in that order); nullptr / 6 = all
int spark_triple_sums(vpin_ctx* c, const vpin_spark_decomm* d, const fq* comb_derefs, const int* halves = nullptr, int nh = 6);
// the same over explicit arrays: derefs slices (6 x N) and val slices (3 x N)
int spark_triple_sums_raw(vpin_ctx* c, const fq* comb_derefs, const fq* vals, size_t N, const int* halves = nullptr, int nh = 6);

// Persistent tail (spark.hip): rounds j0..k-1 of forest level `level` (h = 2^k entries per half) in ONE launch, one workgroup per
// circuit (+ the six dot-product halves when vals != nullptr), leading-coefficient form.  The host keeps the transcript:
// per round spark_tail_wait(idx) -> sums at spark_tail_sums()[3*inst + x] -> spark_tail_reply(idx, r_j) (not after the last
// round, which also leaves the two live entries of every table at spark_tail_final()[6*inst + 2*table + e]); spark_tail_end
// retires the launch's sequence numbers.  len0 = live length before round j0; r_prev = r_{j0-1} when j0 > 0.
size_t spark_tail_pairs();  // rounds with at most this many pairs per circuit go to the tail (0 = never)
// ... and from this many pairs on the host may ASK for it (layers without dot-product halves: up to 8 workgroups per circuit,
// 1024 pairs each); spark_tail_launch then returns 1 when the grid does not fit beside the device's other resident tails --
// the caller proves that round with a launch and asks again at the next one
size_t spark_tail_first_pairs(bool with_dotp);
int spark_tail_launch(vpin_ctx* c, const SparkForest* f, int level, int k, int j0, size_t len0, const fq* pyr, const uint8_t* r_prev,
                      size_t N, const fq* vals, const fq* comb_derefs, fq* scratch, const int* halves = nullptr, int ndotp = 6);
int spark_tail_wait(vpin_ctx* c, int idx, int ninst, int ncirc);
void spark_tail_reply(vpin_ctx* c, int idx, const uint8_t r[32]);
void spark_tail_end(vpin_ctx* c);
void spark_tail_abort(vpin_ctx* c);  // error paths: drain a resident tail and retire its sequence numbers
const fq* spark_tail_sums(vpin_ctx* c);
const fq* spark_tail_final(vpin_ctx* c);

int spark_wait(vpin_ctx* c);  // stream sync

// out[s] = sum_i table[s*len + i] * eq[i], s < nslices (DensePolynomial::evaluate of every slice at
// the point whose eq table is `eq`); results in ctx->h_spark[3*s]; synchronises
// r0 / step: only the entries r0, r0 + step, .. of every slice, against eq[0 .. len/step) (split by residue class)
int spark_slice_evals(vpin_ctx* c, const fq* table, size_t len, int nslices, const fq* eq, size_t r0 = 0, size_t step = 1);

// ---- split by residue class (spark.hip): the local views of rank r0 of `step` ----
// comb_loc: 6 x N/step (the derefs slices at the entries r0 + k*step); comb_rows: nrows_loc x R (rows r0, r0 + step, .. of the
// derefs polynomial's commitment matrix, dense)
int spark_gather_derefs_strided(vpin_ctx* c, const vpin_spark_decomm* d, const fq* mem_rx, const fq* mem_ry, size_t r0, size_t step,
                                fq* comb_loc, fq* comb_rows, size_t R, size_t nrows_loc);
int spark_take_strided(vpin_ctx* c, const fq* src, size_t nloc, size_t r0, size_t step, fq* dst);
int spark_build_forests_strided(vpin_ctx* c, const vpin_spark_decomm* d, const fq* comb_loc, const fq* mem_rx, const fq* mem_ry,
                                const uint8_t r_hash[32], const uint8_t r_hash_sqr[32], const uint8_t r_hash_sqr_boost[32],
                                const uint8_t gamma[32], SparkForest* ops, SparkForest* mem, size_t r0, size_t step);

}  // namespace vpin
