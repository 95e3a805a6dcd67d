/*
 * vpin_hip.h -- C ABI of the MI355X (gfx950) Spartan sat-proof hot path.
 *
 * This is the drop-in boundary for vt-asaplab/vPIN's prover.  The reference has no
 * FFI seam of its own; the seam is the set of plain Rust functions that
 * vPIN_proof_generation/src/commit_test.rs calls into libspartan (SURVEY.md 8(b),
 * rows B1-B5).  Each entry point below names the reference function it replaces
 * (paths relative to src/proof_generation/).  INTEGRATION.md shows the Rust `extern "C"`
 * block a vPIN maintainer would add.
 *
 * Conventions
 *  - A scalar is 32 bytes: the reference's in-memory `Scalar([u64;4])`, little-endian
 *    limbs, Montgomery form with R = 2^256 (Spartan/src/scalar/ristretto255.rs:199-200).
 *    A table is a flat array of scalars == the memory image of a Vec<Scalar>.
 *  - A group element crosses as 128 bytes of extended twisted-Edwards coordinates
 *    (X,Y,Z,T; each 32-byte canonical little-endian mod 2^255-19) or as a 32-byte
 *    compressed ristretto255 encoding (curve25519-dalek CompressedRistretto).
 *  - Every call returns 0 on success or a negative VPIN_E* code; nothing aborts across
 *    the boundary.  The caller owns host buffers; the library owns device memory behind
 *    opaque handles.  One host thread drives one ctx (the reference's protocol loop is
 *    single-threaded); calls are synchronous from the caller's point of view.
 *  - There is NO CPU fallback: every entry point fails with VPIN_ENODEV when no gfx950
 *    device is usable.
 */
#ifndef VPIN_HIP_H
#define VPIN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPIN_OK 0
#define VPIN_EINVAL (-1)  /* bad argument (null handle, length not a power of two, ...) */
#define VPIN_ENODEV (-2)  /* no usable HIP device */
#define VPIN_ENOMEM (-3)  /* device or host allocation failed */
#define VPIN_EHIP (-4)    /* a HIP runtime call failed; see vpin_last_error() */
#define VPIN_ESHAPE (-5)  /* operand shapes do not match (the reference would assert) */
#define VPIN_EVERIFY (-6) /* a verifier rejected the proof (ProofVerifyError / a failed assert in the reference) */

typedef struct vpin_ctx vpin_ctx;
typedef struct vpin_table vpin_table;

const char* vpin_strerror(int code);
/* text of the last HIP failure seen by this thread ("" if none) */
const char* vpin_last_error(void);
/* A host that is about to run something that may take the PROCESS down (a first outing of RCCL at world > 1 ...) leaves the line
 * it has already earned here: on SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGTERM the handler writes these bytes to stdout and ends
 * the process with _exit(0) -- async-signal-safe calls only.  n = 0 disarms (the previous handlers come back).  bench.py arms it
 * with the weak line before its strong sub-record (n <= 8192). */
int vpin_crash_line_set(const char* line, size_t n);
/* ABI version of this library (bumped on any signature change) */
int vpin_abi_version(void);

/* ---- context ------------------------------------------------------------------- */
int vpin_ctx_create(int device, vpin_ctx** out);
/* Same with a stream priority: < 0 high (its kernels are dispatched ahead of other streams' queued
 * workgroups), 0 normal, > 0 low.  For running latency-bound small proofs beside a large one. */
int vpin_ctx_create_prio(int device, int priority, vpin_ctx** out);
/* Same with the context's stream confined to a subset of the device's compute units (hipExtStreamCreateWithCUMask): bit i
 * of cu_mask (n_words x 32 bits) enables CU i in the runtime's numbering -- on an 8-XCD part consecutive bits go round-robin
 * over the XCDs, so the first 8k bits are k CUs of every XCD.  A service that proves one large instance beside many small
 * ones gives each side its own CUs (a spatial split) instead of letting their workgroups share every CU; the row-commitment
 * kernels size their grids to the enabled CUs.  A masked stream has normal priority. */
int vpin_ctx_create_cumask(int device, const uint32_t* cu_mask, uint32_t n_words, vpin_ctx** out);
/* A second stream for an (unmasked) context, confined to the CUs of cu_mask: every proof on the context moves to it when
 * its phase-1 sum-check is over -- the point where the progress word (vpin_ctx_set_progress_flag) becomes 1 -- and returns
 * to the first stream when the call returns.  For a scheduler that holds other contexts back during phase 1 and gives the
 * chip a spatial split afterwards: the large proof's commitments then own their CUs (row-per-lane strips stay in step).
 * cu_mask = NULL removes the second stream. */
int vpin_ctx_set_cumask_after_phase1(vpin_ctx* ctx, const uint32_t* cu_mask, uint32_t n_words);
void vpin_ctx_destroy(vpin_ctx* ctx);
/* hipStream_t the ctx launches on (as void*), for callers that time with HIP events */
void* vpin_ctx_stream(vpin_ctx* ctx);
/* Optional progress word in host memory: a proof stores 1 to it when its phase-1 sum-check is over (and again
 * when the whole sat part is) and 2 when its derefs commitment -- the largest MSM of a SNARK -- is done, so that a
 * caller running other proofs on other contexts can hold them back while the phase-1 kernels of this one are being
 * timed, or until its power-bound MSM has had the chip to itself.  NULL clears it. */
int vpin_ctx_set_progress_flag(vpin_ctx* ctx, int* flag);
/* Several contexts prove on this device at the same time (a service, bench.py's lanes): the long VALU-bound row-commitment
 * kernels of this context then run ONE workgroup per CU instead of three (a wave per SIMD), so that a large instance's
 * commitment no longer stalls every other stream for its whole duration and the latency-bound proofs of the other lanes
 * keep most of every CU (LeNet step on four lanes: 451 -> 435 ms against two workgroups per CU, 513 before any limit). */
int vpin_ctx_set_shared_device(vpin_ctx* ctx, int on);
/* commitment rows this context has handed to the row-per-lane kernel (msm_strip_kernel) since it was created: lets a test
 * or a scheduler see that the path it asked for is the one that ran */
unsigned long long vpin_ctx_strip_rows_taken(vpin_ctx* ctx);
/* row chunks the bucket method (vpin_hyrax_commit_pippenger, VPIN_MSM_PIPPENGER) has launched on this context: a commitment
 * whose digit buffer would pass 2 GiB takes its rows in several (the 2^14 x 2^15 polynomials of a 2^25-constraint instance: 15) */
unsigned long long vpin_ctx_pip_row_chunks(vpin_ctx* ctx);
/* the context's device memory pool: out = {bytes obtained from the driver and still held (handles + temporaries + cached
 * blocks), bytes of those sitting in the free lists (reusable by the next proof), blocks}.  What bench.py's hbm_breakdown and a
 * service's admission control are made of. */
int vpin_ctx_pool_stats(vpin_ctx* ctx, size_t out[3]);
/* out = {calls, bytes}: allocations this library has taken from the driver (hipMalloc) since the process started, all contexts
 * (the contexts' pools, the unpooled tables of a split proof; generator tables not counted).  A proof of a shape its context
 * has already proven must leave both unchanged: tests/test_gpu_pool.py (round 5's SNARK::encode took 20 GB from the driver
 * per call -- 0.3 ms most of the time, 0.2-2.8 s every few calls). */
int vpin_driver_alloc_stats(unsigned long long out[2]);
/* hand the pool's cached (free) blocks back to the driver: after set-up work whose temporaries no proof will reuse (gadget
 * synthesis, SNARK::encode), or when a service changes workload.  Synchronises the context's stream. */
int vpin_ctx_pool_trim(vpin_ctx* ctx);
/* Low-memory mode: proofs on this context keep a smaller working set at a small cost in time.  Today: the SPARK mem forest
 * (4 product circuits over M leaves) is built after the ops forest has been proven, into the memory it frees, its roots coming
 * first from a product reduction -- 16 GiB less for a 2^25-constraint instance, ~1 % more time.  Same bytes. */
int vpin_ctx_set_low_memory(vpin_ctx* ctx, int on);
/* How many proofs the generator window tables built through this context will serve: 0 (default) = many -- the widest
 * windows the table budget allows (fewest additions per scalar; the table costs ~0.2 s to build for the largest
 * instance); n > 0 = a process that proves n times and exits (vpin_prove: a process per label, like `cargo run -- <label>`):
 * the window width then minimises table construction + n proofs' commitments (10 instead of 12 bits for the 2^25
 * instance, 8 for the small ones). */
int vpin_ctx_set_expected_proofs(vpin_ctx* ctx, int n);
/* compute units and shader clock (kHz) of the context's device: the VALU-issue ceiling bench.py prices the MSM against */
int vpin_ctx_device_props(vpin_ctx* ctx, int* num_cus, int* clock_khz);
/* free / total HBM of the context's device, bytes (the table budgets are chosen from `total`) */
int vpin_ctx_mem_info(vpin_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
int vpin_ctx_sync(vpin_ctx* ctx);

/* Host buffers that cross the bus often (a service's witness buffers, the triplets of an instance it proves again and again)
 * can be page-locked once: the host-buffer entry points (vpin_r1cs_upload, vpin_table_upload, vpin_sat_prove, vpin_snark_prove)
 * then copy at the bus rate instead of through the driver's staging buffers (CNN A's sat proof from host buffers: 18.9 -> see
 * DESIGN.md section 5).  hipHostRegister / hipHostUnregister: the range must stay valid until unregistered. */
int vpin_host_register(void* p, size_t bytes);
int vpin_host_unregister(void* p);

/* ---- tables: device-resident Vec<Scalar> ---------------------------------------- */
/* DensePolynomial::new (Spartan/src/dense_mlpoly.rs:132-138): len must be a power of 2 */
int vpin_table_upload(vpin_ctx* ctx, const uint8_t* mont32, size_t len, vpin_table** out);
int vpin_table_alloc(vpin_ctx* ctx, size_t len, vpin_table** out); /* zero-filled */
/* adopt caller-owned device memory (e.g. a framework tensor) without copying */
int vpin_table_wrap(vpin_ctx* ctx, void* device_ptr, size_t len, vpin_table** out);
int vpin_table_clone(vpin_ctx* ctx, const vpin_table* src, vpin_table** out);
void vpin_table_free(vpin_ctx* ctx, vpin_table* t);
size_t vpin_table_len(const vpin_table* t); /* DensePolynomial::len(), halves on every bind */
void* vpin_table_device_ptr(const vpin_table* t);
/* Index<usize> for DensePolynomial (dense_mlpoly.rs:295-302): read n scalars from off */
int vpin_table_read(vpin_ctx* ctx, const vpin_table* t, size_t off, size_t n, uint8_t* out);
/* the reverse: n scalars from host memory into the table's allocation at element offset `off` (len is unchanged) */
int vpin_table_write(vpin_ctx* ctx, vpin_table* t, size_t off, size_t n, const uint8_t* src);

/* ---- sum-check round reductions -------------------------------------------------- */
/* Round evaluation loop of ZKSumcheckInstanceProof::prove_cubic_with_additive_term
 * (Spartan/src/sumcheck.rs:624-652) with R1CSProof::prove_phase_one's combiner
 * A*(B*C-D) (Spartan/src/r1csproof.rs:104-108).  out = e0|e2|e3, 3 x 32 B Montgomery. */
int vpin_sc_cubic_round(vpin_ctx* ctx, const vpin_table* tau, const vpin_table* Az,
                        const vpin_table* Bz, const vpin_table* Cz, uint8_t out_e0_e2_e3[96]);
/* Round evaluation loop of ZKSumcheckInstanceProof::prove_quad (sumcheck.rs:460-469)
 * with prove_phase_two's combiner A*B (r1csproof.rs:139-140).  out = e0|e2. */
int vpin_sc_quad_round(vpin_ctx* ctx, const vpin_table* A, const vpin_table* B,
                       uint8_t out_e0_e2[64]);
/* DensePolynomial::bound_poly_var_top (dense_mlpoly.rs:229-236) on k tables at once
 * (sumcheck.rs:673-676 / :485-486): Z[i] += r*(Z[i+n]-Z[i]); len /= 2. In place. */
int vpin_sc_bind(vpin_ctx* ctx, vpin_table* const* tables, int k, const uint8_t r[32]);
/* Fused single pass: bind the four (two) tables with r, then evaluate the NEXT round on
 * the folded tables -- one read of every live element and one write of every folded
 * element per round (SURVEY.md 8(d) "algorithmic bytes").  Same results as
 * vpin_sc_bind followed by vpin_sc_*_round.  Requires len >= 4. */
int vpin_sc_cubic_bind_round(vpin_ctx* ctx, vpin_table* tau, vpin_table* Az, vpin_table* Bz,
                             vpin_table* Cz, const uint8_t r[32], uint8_t out_e0_e2_e3[96]);
int vpin_sc_quad_bind_round(vpin_ctx* ctx, vpin_table* A, vpin_table* B, const uint8_t r[32],
                            uint8_t out_e0_e2[64]);

/* EqPolynomial::evals (dense_mlpoly.rs:78-94): r = ell x 32 B, result has 2^ell entries */
int vpin_eq_table(vpin_ctx* ctx, const uint8_t* r, int ell, vpin_table** out);

/* Eq-factored phase 1 (same e0,e2,e3 as vpin_sc_cubic_*, cheaper): the folded eq(tau,.) table of round j
 * equals a scalar times the suffix table eq(tau_{j+1..}, .), so it is neither stored per round nor folded.
 * vpin_eq_suffix_tables builds all suffix tables once (one table of 2^ell elements); the round calls take
 * (Az,Bz,Cz) and level = j+1 and return the UNSCALED sums S_x = sum_i E[i]*(Az_x[i]*Bz_x[i] - Cz_x[i]) for
 * x = 0, 2, 3 (the reference's comb is tau*(Az*Bz - Cz), r1csproof.rs:104-108); the caller multiplies by
 * c_{j,x} = s_j*((1-tau_j) + x*(2*tau_j-1)) with s_j = prod_{i<j} eq1(tau_i, r_i) to get e0, e2, e3. */
int vpin_eq_suffix_tables(vpin_ctx* ctx, const uint8_t* tau, int ell, vpin_table** out);
int vpin_sc_cubic3_round(vpin_ctx* ctx, const vpin_table* pyramid, int ell, int level, vpin_table* Az,
                         vpin_table* Bz, vpin_table* Cz, uint8_t out_S0_S2_S3[96]);
int vpin_sc_cubic3_bind_round(vpin_ctx* ctx, const vpin_table* pyramid, int ell, int level, vpin_table* Az,
                              vpin_table* Bz, vpin_table* Cz, const uint8_t r[32], uint8_t out_S0_S2_S3[96]);
/* Leading-coefficient form of the same round, the one vpin_sat_prove runs while phase 1's claim is consistent: with
 * the eq factor split off the round polynomial is l(x) * t(x), t(x) = sum_i E[i]*(Az_x*Bz_x - Cz_x)[i] quadratic, and
 * the kernel returns t(0), the x^2 coefficient sum_i E[i]*(dAz*dBz)[i] and -- only when r == NULL (first round) --
 * t(1).  The same e0, e2, e3 of Spartan/src/sumcheck.rs:624-652 follow by exact field identities.  r != NULL binds
 * the tables with r first (dense_mlpoly.rs:229-236). */
int vpin_sc_cubic3_lead_round(vpin_ctx* ctx, const vpin_table* pyramid, int ell, int level, vpin_table* Az,
                              vpin_table* Bz, vpin_table* Cz, const uint8_t* r, uint8_t out[96]);
/* One round of SumcheckInstanceProof::prove_cubic_batched (Spartan/src/sumcheck.rs:273-330,346-370) as
 * ProductCircuitEvalProofBatched::prove issues it (Spartan/src/product_tree.rs:258-385): `ncirc` product circuits of
 * `n` leaves each (tree t = forest[t*2n ..), level l at offset 2n - (2n >> l), left/right halves = poly_A/poly_B) proven
 * in lock-step, eq-factored (E = eq suffix table of this round at E[e_off..], len/2 entries, or len/4 when r != NULL and
 * the tables are first bound with r).  out: per circuit 3 scalars -- lead == 0: sum_i E[i]*(A_x*B_x)[i] at x = 0, 2, 3;
 * lead == 1: the value at 0, the x^2 coefficient sum_i E[i]*(dA*dB)[i], and zero.  derefs != NULL (ncirc == 12,
 * level == 0) adds the six DotProductCircuit halves of Spartan/src/sparse_mlpoly.rs:1103-1125 (A*B*C over derefs row |
 * col slices (6 x n) and the val slices (3 x n); first fold into scratch, 18 x n/4): out[12..18) = sums at 0, 2, 3. */
int vpin_spark_batched_round(vpin_ctx* ctx, vpin_table* forest, size_t n, int ncirc, int level, size_t len,
                             const vpin_table* E, size_t e_off, const uint8_t* r, int lead, const vpin_table* derefs,
                             const vpin_table* vals, vpin_table* scratch, int first_fold, uint8_t* out);

/* ---- Pedersen generators and fixed-base MSM ----------------------------------------- */
/* The generator stream of MultiCommitGens::new (Spartan/src/commitments.rs:20-38):
 * nb points g[0..nb) as 128-byte X|Y|Z|T each.  Builds the device window table
 * T[w][j][k] = (k+1) * 2^(8w) * g_j once; every MSM below is a gather-and-add over it.
 * For R1CSGens (Spartan/src/r1csproof.rs:84-89) the stream is g[0..R+2): G = g[0..R),
 * gens_1.G[0] = g[R], h = g[R+1]; gens_3 / gens_4 are prefixes with h = g[3] / g[4]. */
typedef struct vpin_gens vpin_gens;
int vpin_gens_create(vpin_ctx* ctx, const uint8_t* gens_xyzt, size_t nb, vpin_gens** out);
/* RistrettoPoint::from_uniform_bytes for nb generators at once on the device: stream64 = 64 x nb bytes of the label's SHAKE256
 * stream (Spartan/src/commitments.rs:20-38), out_xyzt = nb x 128 bytes X|Y|Z|T like vpin_host_gens_derive (same points;
 * the projective representative may differ) */
int vpin_gens_map_stream(vpin_ctx* ctx, const uint8_t* stream64, size_t nb, uint8_t* out_xyzt);
/* Shared form: one immutable table per (device, label), holding the longest prefix of the label's
 * generator stream requested so far -- MultiCommitGens::new(n, label) for every n is a prefix of the same
 * SHAKE256 stream (Spartan/src/commitments.rs:20-38), so all contexts / streams / polynomial sizes of a
 * label share it.  gens_xyzt (nb points) is only read when a new table has to be built.  budget_gb = 0:
 * default table budget.  The returned handle is owned by the registry (never pass it to vpin_gens_free). */
int vpin_gens_shared(vpin_ctx* ctx, const char* label, const uint8_t* gens_xyzt, size_t nb, size_t budget_gb,
                     const vpin_gens** out);
/* frees every shared table; only when no context uses them any more */
void vpin_gens_shared_clear(void);
/* bytes of the process-wide window tables on `device` */
size_t vpin_gens_shared_bytes(int device);
void vpin_gens_free(vpin_ctx* ctx, vpin_gens* g);
size_t vpin_gens_count(const vpin_gens* g);
/* Window layout the table budget chose: out = {c, W, split, c_hi, W_hi, bases}: generators [0, split) have c-bit signed
 * windows (W per scalar), [split, bases) c_hi-bit ones (0 when the table has a single segment); `bases` counts the
 * nb generators plus the prefix-sum bases S_k = g_0 + ... + g_{2^k - 1} behind them. */
int vpin_gens_layout(const vpin_gens* g, size_t out[6]);
/* bytes per window-table entry (affine Niels point, possibly padded to a cache line) */
size_t vpin_gens_entry_bytes(void);
/* DensePolynomial::commit_inner (Spartan/src/dense_mlpoly.rs:160-175): Z viewed as L rows
 * of R = len/L scalars; out[i] = compress( sum_j Z[iR+j]*g[j] + blinds[i]*g[blind_base] ).
 * blinds = L x 32 B Montgomery scalars on the host. */
int vpin_hyrax_commit(vpin_ctx* ctx, const vpin_gens* g, const vpin_table* Z, const uint8_t* blinds,
                      size_t L, size_t blind_base, uint8_t* out_compressed /* L*32 */);
/* Rows [row0, row0 + nrows) of the same commitment (Z viewed as L rows; blinds = nrows x 32 B for those rows, or NULL
 * for DensePolynomial::commit(gens, None)).  The L row commitments are independent MSMs over shared generators
 * (rayon's into_par_iter at dense_mlpoly.rs:166-173), so a commitment splits across GPUs by rows: every rank commits
 * a contiguous block and the 32-byte results are all-gathered -- no point crosses a link. */
int vpin_hyrax_commit_rows(vpin_ctx* ctx, const vpin_gens* g, const vpin_table* Z, size_t L, size_t row0, size_t nrows,
                           const uint8_t* blinds, size_t blind_base, uint8_t* out_compressed);
/* The same commitment (DensePolynomial::commit_inner, Spartan/src/dense_mlpoly.rs:160-175) computed the way
 * GroupElement::vartime_multiscalar_mul does above 190 terms (Spartan/src/group.rs:103-122, dalek's Pippenger): signed
 * c_bits-bit digits, bucket accumulation staged in LDS, running-sum reduction -- from the generators alone, no window table
 * walked.  A measured alternative, NOT what the provers call (it is slower on this part: profiles/r05_pippenger.txt).
 * blinds may be NULL (commit(gens, None)); c_bits in 9..12, or 0 = the measured best (9).  Same bytes as
 * vpin_hyrax_commit. */
int vpin_hyrax_commit_pippenger(vpin_ctx* ctx, const vpin_gens* g, const vpin_table* Z, const uint8_t* blinds, size_t L,
                                size_t blind_base, int c_bits, uint8_t* out_compressed);
/* The two commitments of proof_point_{add,mult}.rs:44-52 plus their row-wise sum
 * (:75-80) in one call: out_sum[i] = compress(decompress(a[i]) + decompress(b[i])). */
int vpin_hyrax_commit_pair(vpin_ctx* ctx, const vpin_gens* g, const vpin_table* Za, const vpin_table* Zb,
                           const uint8_t* blinds_a, const uint8_t* blinds_b, size_t L, size_t blind_base,
                           uint8_t* out_a, uint8_t* out_b, uint8_t* out_sum);
/* GroupElement::vartime_multiscalar_mul (Spartan/src/group.rs:103-122) over a prefix of the
 * generator stream: out[r] = sum_{j<ncols} s[r][j] * g[j] for `rows` independent rows of
 * host scalars (Montgomery).  Either output may be NULL. */
int vpin_gens_msm(vpin_ctx* ctx, const vpin_gens* g, const uint8_t* scalars_mont, size_t rows, size_t ncols,
                  uint8_t* out_compressed /* rows*32 */, uint8_t* out_xyzt /* rows*128 */);

/* GroupElement::vartime_multiscalar_mul (Spartan/src/group.rs:103-122) over ARBITRARY points, given as compressed
 * ristretto255 encodings (n x 32 B) with n Montgomery scalars: out = sum_i s_i * decompress(P_i).  One lane per term
 * (decompression, double-and-add, block tree): sized for the row counts of Hyrax commitments (the verifier's <L, C> of
 * dense_mlpoly.rs:381-404), not for millions of points.  VPIN_EVERIFY when an encoding does not decode.  Either output
 * may be NULL. */
int vpin_msm(vpin_ctx* ctx, const uint8_t* scalars_mont, const uint8_t* points_compressed, size_t n, uint8_t* out_compressed,
             uint8_t* out_xyzt);
/* out[i] = compress(decompress(a[i]) + decompress(b[i])), n points: the row-wise sum of two commitment vectors
 * (vPIN_proof_generation/src/commit_test.rs:340-361).  VPIN_EVERIFY when an encoding does not decode. */
int vpin_points_add(vpin_ctx* ctx, const uint8_t* a_compressed, const uint8_t* b_compressed, size_t n, uint8_t* out_compressed);

/* Same MSM for the few-row, latency-bound case: returns the per-workgroup partial points
 * (X|Y|Z|T, 128 B each, rows x vpin_gens_msm_parts_count(ncols)); the caller adds them. */
size_t vpin_gens_msm_parts_count(size_t ncols);
int vpin_gens_msm_parts(vpin_ctx* ctx, const vpin_gens* g, const uint8_t* scalars_mont, size_t rows, size_t ncols,
                        uint8_t* parts_xyzt);

/* DensePolynomial::bound (Spartan/src/dense_mlpoly.rs:220-227): LZ[i] = sum_j L[j]*Z[j*R+i],
 * Lvec = L_size host scalars, out = R = len/L_size host scalars. */
int vpin_poly_bound(vpin_ctx* ctx, const vpin_table* Z, const uint8_t* Lvec, size_t L_size, uint8_t* out_LZ);
/* The hash layer's slice pass (sparse_mlpoly.rs:740-849; single GPU): Z holds 2^nbits slices of N = 2^r_len scalars each.
 * One pass over the first `used` slices yields (i) their evaluations at r -- DensePolynomial::evaluate, dense_mlpoly.rs:
 * 239-253 -- in evals_out (used x 32 bytes) and (ii), for combining challenges ch (nbits scalars; NULL: skip), what
 * PolyEvalProof::prove's first step computes for the whole Z at the point (ch, r): LZ = L^T Z with L = eq((ch, r)[..left])
 * (dense_mlpoly.rs:340-349; R = 2^(ell - ell/2) scalars in LZ_out), without reading Z again.  The slices from `used` on must be
 * zero (the reference pads the combined polynomials with zero slices).  Needs ell / 2 >= nbits (ell = r_len + nbits). */
int vpin_poly_slices_bound(vpin_ctx* ctx, const vpin_table* Z, int nbits, int used, const uint8_t* r, size_t r_len,
                           const uint8_t* ch, uint8_t* evals_out, uint8_t* LZ_out);
/* The same with the first n32 slices given as u32 values (host memory, n32 x 2^r_len: addresses and timestamps of the
 * computation decommitment, sparse_mlpoly.rs:418-428, whose field images Scalar::from(v) the proof never forms: 4 bytes and
 * eight 32-bit multiply-adds per entry instead of 32 bytes and a product mod q) and the remaining used - n32 slices as a table
 * of field elements (Zfq, NULL when used == n32).  Same elements as vpin_poly_slices_bound over the field images. */
int vpin_poly_slices_bound_u32(vpin_ctx* ctx, const uint32_t* slices_u32, int n32, const vpin_table* Zfq, int nbits, int used,
                               const uint8_t* r, size_t r_len, const uint8_t* ch, uint8_t* evals_out, uint8_t* LZ_out);

/* ---- the sat proof (host orchestration over the kernels above) ------------------------ */
/* R1CSInstance after Instance::new's padding and column remap (Spartan/src/lib.rs:138-244):
 * num_cons / num_vars are powers of two; col indexes z = [vars | 1 | inputs | 0...] of length
 * 2*num_vars; val = 32-byte Montgomery scalars. */
typedef struct {
  size_t num_cons, num_vars, num_inputs;
  size_t nnz[3];          /* A, B, C */
  const uint32_t* row[3];
  const uint32_t* col[3];
  const uint8_t* val[3];
} vpin_r1cs;
/* Device-resident instance: CSR + CSC copies of (A,B,C) in HBM (the reference's
 * SparseMatPolynomial, Spartan/src/sparse_mlpoly.rs:330-380), built once per instance. */
typedef struct vpin_r1cs_dev vpin_r1cs_dev;
int vpin_r1cs_upload(vpin_ctx* ctx, const vpin_r1cs* inst, vpin_r1cs_dev** out);
void vpin_r1cs_free(vpin_ctx* ctx, vpin_r1cs_dev* d);
void vpin_r1cs_dims(const vpin_r1cs_dev* d, size_t* num_cons, size_t* num_vars, size_t* num_inputs);
/* z = [vars | 1 | inputs | 0...] (vPIN_proof_generation/src/commit_test.rs:162-170); inputs on the host */
int vpin_r1cs_build_z(vpin_ctx* ctx, const vpin_r1cs_dev* d, const vpin_table* vars, const uint8_t* inputs,
                      vpin_table** out_z);
/* R1CSInstance::multiply_vec (Spartan/src/r1csinstance.rs:272-285 -> sparse_mlpoly.rs:467-481) */
int vpin_r1cs_multiply_vec(vpin_ctx* ctx, const vpin_r1cs_dev* d, const vpin_table* z, vpin_table** Az,
                           vpin_table** Bz, vpin_table** Cz);
/* R1CSInstance::compute_eval_table_sparse (r1csinstance.rs:287-295 -> sparse_mlpoly.rs:483-498) folded with
 * the three challenges r_A|r_B|r_C as at commit_test.rs:257-268: out[i] = r_A*A(rx,i)+r_B*B(rx,i)+r_C*C(rx,i) */
int vpin_r1cs_eval_table(vpin_ctx* ctx, const vpin_r1cs_dev* d, const vpin_table* evals_rx, const uint8_t r_abc[96],
                         vpin_table** out);
/* R1CSInstance::evaluate (r1csinstance.rs:297-302) from the two eq tables; out = Ar|Br|Cr */
int vpin_r1cs_evaluate(vpin_ctx* ctx, const vpin_r1cs_dev* d, const vpin_table* evals_rx, const vpin_table* evals_ry,
                       uint8_t out[96]);

/* One gadget instance's satisfiability proof exactly as vPIN drives it:
 * proof_point_{add,mult}.rs:38-94 (commit para / input under RandomTape::new(&[2]), combine)
 * + commit_test.rs:59-109 my_lib_prove up to the Ar/Br/Cr claims (= my_R1CSProof_prove,
 * commit_test.rs:136-334, + inst.evaluate).  The reference seeds its two RandomTapes from
 * OsRng (Spartan/src/random.rs:14-22); the two 64-byte draws are explicit inputs here, which
 * is what makes proofs reproducible.  Outputs: bincode bytes of R1CSProof, the two L x 32 B
 * commitments the verifier needs, inst_evals = Ar|Br|Cr, rx (log2 num_cons scalars),
 * ry (log2 num_vars + 1 scalars).  rx_out / ry_out may be NULL. */
int vpin_sat_prove(vpin_ctx* ctx, const vpin_r1cs* inst, const uint8_t* vars_para, const uint8_t* vars_input,
                   const uint8_t* vars, const uint8_t* inputs, const uint8_t seed_commit64[64],
                   const uint8_t seed_proof64[64], uint8_t* proof_out, size_t proof_cap, size_t* proof_len,
                   uint8_t* comm_para_out, uint8_t* comm_input_out, uint8_t inst_evals_out[96],
                   uint8_t* rx_out, uint8_t* ry_out);
/* Same proof with the instance and the three assignments already resident in HBM (what bench.py times). */
int vpin_sat_prove_resident(vpin_ctx* ctx, const vpin_r1cs_dev* inst, const vpin_table* vars_para,
                            const vpin_table* vars_input, const vpin_table* vars, const uint8_t* inputs,
                            const uint8_t seed_commit64[64], const uint8_t seed_proof64[64], uint8_t* proof_out,
                            size_t proof_cap, size_t* proof_len, uint8_t* comm_para_out, uint8_t* comm_input_out,
                            uint8_t inst_evals_out[96], uint8_t* rx_out, uint8_t* ry_out);
size_t vpin_sat_proof_max_bytes(size_t num_cons, size_t num_vars);
/* my_dense_mlpoly_commit (vPIN_proof_generation/src/commit_test.rs:27-57) as proof_point_{add,mult}.rs:58-59 call it: the
 * Hyrax commitment of the WHOLE assignment under the element-wise sum of the blinds of the two commitments made before it
 * (RandomTape::new(&[2]), labels b"poly_blinds" x L twice).  The reference computes it inside its timed span and then reads
 * row 0 only (assert_eq!(c, c_prime), proof_point_mult.rs:69-73): the proof carries the row-wise sum of the two partial
 * commitments, which is the same vector.  vpin_sat_prove* / vpin_snark_prove* therefore never compute it (SURVEY.md 8(f) row
 * N4); this entry point exists so that a host which wants the reference's assert, or the reference's span "with" the dead
 * work, can have it: out = L x 32 B, L = 2^(log2(len) / 2), equal to comm_para + comm_input row by row. */
int vpin_dense_mlpoly_commit_sum(vpin_ctx* ctx, const vpin_table* vars, const uint8_t seed_commit64[64], uint8_t* out_compressed);
/* R1CSGens::new (Spartan/src/r1csproof.rs:84-89) for this polynomial size ahead of the first proof: host
 * fixed-base tables + the shared device window table (built on demand by the prove calls otherwise). */
int vpin_sat_prepare(vpin_ctx* ctx, size_t num_vars);
/* R1CSCommitmentGens::new (Spartan/src/r1csinstance.rs:29-49; SparseMatPolyCommitmentGens::new,
 * sparse_mlpoly.rs:300-329) for an instance of this shape ahead of the first encode / proof: derives the
 * b"gens_r1cs_eval" stream and builds (or finds) the shared device window table.  Call it for the LARGEST instance
 * first: smaller ones then share its table instead of each leaving a superseded one behind. */
int vpin_spark_prepare(vpin_ctx* ctx, size_t num_cons, size_t num_vars, size_t max_nnz);

/* wall-clock spans of the last vpin_sat_prove call on this thread's process, seconds:
 * [0] polycommit (uploads + 2 commits + combine)  [1] prove_sc_phase_one (eq table, SpMV, 4 uploads, rounds)
 * [2] prove_sc_phase_two  [3] polyeval  [4] total  [5] generators (0 when cached)
 * [6] host SpMV share of [1]+[2]  [7] inst.evaluate */
void vpin_sat_last_timings(double out[8]);

/* ------------------------------------------------------------------------------------------------
 * SPARK: the computation commitment and the sparse-polynomial evaluation proof that complete the
 * reference's SNARK.
 * ---------------------------------------------------------------------------------------------- */

/* ComputationDecommitment (Spartan/src/lib.rs:70-73): MultiSparseMatPolynomialAsDense
 * (Spartan/src/sparse_mlpoly.rs:285-292) resident in HBM. */
typedef struct vpin_spark_decomm vpin_spark_decomm;

/* ---- one proof over the GPUs of one node -----------------------------------------------------------
 * SPMD: every rank (one process per GPU, or one thread per rank in a rehearsal) holds the same instance, creates a
 * vpin_comm, attaches it to its context with vpin_ctx_set_comm and calls the SAME vpin_snark_prove_* / vpin_sat_prove_*
 * with the same inputs and seeds.  Inside the call the heavy steps are sharded and their small results all-gathered:
 *   - every Hyrax row commitment by interleaved rows r, r + world, .. (the witness pair of proof_point_mult.rs:44-52 and
 *     the derefs commitment of sparse_mlpoly.rs:525-531; rows are independent MSMs over shared generators,
 *     dense_mlpoly.rs:160-175) -- 32 bytes per row cross the links, no point and no reduction;
 *   - power-of-two worlds: BY RESIDUE CLASS.  The folds pair (i, i + len/2) (dense_mlpoly.rs:229-236), so rank r keeps the
 *     entries = r (mod world) of every table -- Az, Bz, Cz, z and the eval table of the sat proof's two sum-checks
 *     (sumcheck.rs:428-776), the leaves, trees and dot-product vectors of all 12 + 4 product circuits
 *     (product_tree.rs:259-385) -- and folds them with the single-GPU kernels; no table entry ever moves.  Per round a rank
 *     publishes its partial sums (3 scalars per circuit, 96 bytes in the sat rounds); the last log2(world) rounds of every
 *     sum-check run on the host from the world gathered entries per table;
 *   - other worlds (3, 5, 6, ..): the 12 + 4 product circuits and the 6 dot-product halves by circuit index
 *     (vpin_dist_plan below); the sat sum-checks stay replicated;
 *   - the 23 slice evaluations of the hash layer (sparse_mlpoly.rs:740-849) by residue class or by slice, and
 *     DensePolynomial::bound of the evaluation proofs (dense_mlpoly.rs:220-227) by rows (partial vectors all-gathered on
 *     the device: RCCL ncclAllGather when enabled, staged through the host transport otherwise).
 * The bullet reductions, the sigma protocols and the transcript stay replicated.  Every rank returns the same bytes, equal
 * to the single-GPU proof.  Without a comm (or with world == 1) the calls are the single-GPU ones.
 *
 * Transports for the small host-side exchanges (results are already on the host, where the transcript lives):
 *   vpin_comm_create_shm        ranks = processes of one node; a POSIX shared-memory segment `name`, which MUST be unique per
 *                               job (two live jobs under one name attach to each other; only the segment of a job that
 *                               has died is recognised and replaced).  Rank 0 creates it and unlinks it once everyone is
 *                               attached.  A flat all-gather of a few
 *                               hundred bytes costs ~1-3 us, against ~20-30 us for a RCCL kernel launch.
 *   vpin_comm_create_local      `world` handles for the threads of one process (rehearsals, tests)
 *   vpin_comm_create_callbacks  the caller's own all-gather (gloo in the CPU tests; a host's existing fabric)
 * Every wait is bounded (VPIN_COMM_TIMEOUT_S, default 120 s) and watches a shared abort word: a rank that fails or
 * disappears makes the others return VPIN_ECOMM instead of hanging. */
#define VPIN_ECOMM (-7) /* a peer failed or timed out, or the ranks disagree on a collective */
typedef struct vpin_comm vpin_comm;
typedef int (*vpin_allgather_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank);
/* slot_bytes: largest message sent in one piece (larger ones are chunked); 0 = default (1 MiB) */
int vpin_comm_create_shm(const char* name, int rank, int world, size_t slot_bytes, vpin_comm** out);
int vpin_comm_create_local(int world, size_t slot_bytes, vpin_comm** out_handles /* world entries */);
int vpin_comm_create_callbacks(int rank, int world, vpin_allgather_fn fn, void* user, vpin_comm** out);
void vpin_comm_destroy(vpin_comm* cm);
int vpin_comm_rank(const vpin_comm* cm);
int vpin_comm_world(const vpin_comm* cm);
/* host buffers: recv = world x bytes */
int vpin_comm_allgather(vpin_comm* cm, const void* send, void* recv, size_t bytes);
/* RCCL for device buffers: rank 0 draws the ncclUniqueId, the host transport carries it, every rank runs
 * ncclCommInitRank on the context's device (librccl is dlopen'ed; VPIN_ENODEV when it is absent).  Collective. */
int vpin_comm_enable_rccl(vpin_comm* cm, vpin_ctx* ctx);
/* device buffers on the context's stream: ncclAllGather when RCCL is enabled, else staged through the host transport */
int vpin_comm_allgather_dev(vpin_comm* cm, vpin_ctx* ctx, const void* d_send, void* d_recv, size_t bytes);
/* Rehearsal of N ranks on fewer GPUs: on != 0 makes the ranks compute ONE AT A TIME (a token is held between
 * collectives and passed on inside them), so that the time a rank spends between two collectives is its own work and
 * nothing else; the statistics then give the critical path of the N-GPU run: sum over the collectives of the slowest
 * rank's section.  Collective on first use. */
int vpin_comm_set_serialize(vpin_comm* cm, int on);
typedef struct {
  uint64_t collectives;
  double bytes;   /* payload this rank sent */
  double wait_s;  /* time inside collectives (waiting for peers + copying) */
  double busy_s;  /* time between collectives (this rank's own sections) */
  double crit_s;  /* sum over collectives of max over ranks of the section before it */
} vpin_comm_stats;
int vpin_comm_stats_read(vpin_comm* cm, vpin_comm_stats* out, int reset);
/* the same per step of the protocol (the section before a collective is booked on the call site's tag; under
 * vpin_comm_set_serialize the proofs add empty marker collectives between their steps): text, one line per tag
 * "<tag> <collectives> <busy_s> <crit_s>"; returns the buffer size needed */
size_t vpin_comm_stats_tags(vpin_comm* cm, char* buf, size_t cap);
/* seconds per all-gather of `bytes` per rank over `iters` back-to-back collectives issued inside the library. Collective. */
int vpin_comm_latency(vpin_comm* cm, size_t bytes, int iters, double* seconds_per_collective);
/* Marks the group dead: every peer's current or next wait returns VPIN_ECOMM at once instead of running into the timeout
 * (shared-memory and local transports; a callbacks transport is the caller's to tear down).  The proving entry points do
 * this themselves when a rank leaves a collective proof with VPIN_ENOMEM / VPIN_EHIP / VPIN_ECOMM; a host calls it when a
 * rank fails outside the library (a witness that could not be read, a signal). */
void vpin_comm_abort(vpin_comm* cm);
/* attach (or detach with NULL): proofs on this context become collective calls over `cm`'s ranks */
int vpin_ctx_set_comm(vpin_ctx* ctx, vpin_comm* cm);
/* the b"gens_r1cs_eval" view a polynomial of 2^ell scalars is committed under (PolyCommitmentGens::new(ell, ..)) */
int vpin_spark_gens_view(vpin_ctx* ctx, size_t ell, const vpin_gens** out, size_t* L, size_t* R);
/* How the circuits of one proof are dealt to the ranks (deterministic; every rank computes the same plan):
 * owner_ops[12] / owner_dotp[6] / owner_mem[4] = owning rank of each "ops" circuit (row read A,B,C | row write A,B,C |
 * col read | col write), dot-product half (matrix, half) and "mem" circuit (row init, row audit, col init, col audit). */
int vpin_dist_plan(int world, int owner_ops[12], int owner_dotp[6], int owner_mem[4]);

/* bincode size of R1CSCommitment (Spartan/src/r1csinstance.rs:53-58) for this instance. */
size_t vpin_spark_comm_bytes(const vpin_r1cs* inst);
/* upper bound on bincode(SNARK) (Spartan/src/lib.rs:334-338) for this instance. */
size_t vpin_snark_proof_max_bytes(const vpin_r1cs* inst);

/* SNARK::encode (Spartan/src/lib.rs:347-359) = R1CSInstance::commit (r1csinstance.rs:309-322) =
 * SparseMatPolynomial::multi_commit (sparse_mlpoly.rs:382-438,500-520): dense representation
 * (ops_addr / read_ts / audit_ts / val), comb_ops and comb_mem committed row-wise under the
 * b"gens_r1cs_eval" generators.  comm_out receives bincode(R1CSCommitment): num_cons, num_vars,
 * num_inputs, batch_size = 3, num_ops, num_mem_cells, comm_comb_ops, comm_comb_mem.
 * The generators are sized from the instance's own max nnz (the reference passes a hand-tuned
 * num_non_zero_entries, point_mult.rs:67 / point_addition.rs:70, which must round to the same
 * power of two or its commit asserts, commitments.rs:95). */
int vpin_spark_encode(vpin_ctx* ctx, const vpin_r1cs* inst, vpin_spark_decomm** out, uint8_t* comm_out,
                      size_t comm_cap, size_t* comm_len);
void vpin_spark_decomm_free(vpin_ctx* ctx, vpin_spark_decomm* d);
/* The column SNARK::encode found to carry a large share of matrix A / B / C's entries (the constant 1 of an R1CS, or the
 * first input), 0xffffffff where there is none or the instance is too small for it to matter.  Derefs::new
 * (Spartan/src/sparse_mlpoly.rs:56-71) puts the SAME scalar eq(ry)[col] at every one of those entries, so the derefs
 * commitment of each proof (sparse_mlpoly.rs:525-531) adds v * g_j for them instead of walking a window table. */
void vpin_spark_decomm_hot_cols(const vpin_spark_decomm* d, uint32_t out[3]);

/* my_lib_prove in full (vPIN_proof_generation/src/commit_test.rs:59-133): the sat proof of
 * vpin_sat_prove_resident, then inst_evals, then R1CSEvalProof::prove (r1csinstance.rs:330-354 ->
 * sparse_mlpoly.rs:1466-1533) on the same transcript and RandomTape.  proof_out receives
 * bincode(SNARK { r1cs_sat_proof, inst_evals, r1cs_eval_proof }) -- what proof_point_mult.rs:96
 * measures as "Proof size". */
int vpin_snark_prove_resident(vpin_ctx* ctx, const vpin_r1cs_dev* inst, const vpin_spark_decomm* decomm,
                              const vpin_table* vars_para, const vpin_table* vars_input, const vpin_table* vars,
                              const uint8_t* inputs, const uint8_t seed_commit64[64], const uint8_t seed_proof64[64],
                              uint8_t* proof_out, size_t proof_cap, size_t* proof_len, uint8_t* comm_para_out,
                              uint8_t* comm_input_out);
/* proof_point_mult.rs:38-94 from host buffers (instance triplets + the three assignments in host memory):
 * SNARK::encode, then my_lib_prove in full.  comm_out receives bincode(R1CSCommitment). */
int vpin_snark_prove(vpin_ctx* ctx, const vpin_r1cs* inst, const uint8_t* vars_para, const uint8_t* vars_input,
                     const uint8_t* vars, const uint8_t* inputs, const uint8_t seed_commit64[64],
                     const uint8_t seed_proof64[64], uint8_t* proof_out, size_t proof_cap, size_t* proof_len,
                     uint8_t* comm_out, size_t comm_cap, size_t* comm_len, uint8_t* comm_para_out,
                     uint8_t* comm_input_out);
/* Verifiers (host C++; the evaluation proofs' fixed-base MSMs run on the device tables).  VPIN_OK = accept,
 * VPIN_EVERIFY = reject.
 * vpin_sat_verify   = my_r1csproof_verify (vPIN_proof_generation/src/commit_test.rs:340-496) with claimed inst_evals;
 * vpin_snark_verify = my_lib_verify (commit_test.rs:498-548): sat part, inst_evals, R1CSEvalProof::verify
 *                     (Spartan/src/r1csinstance.rs:356-372 -> sparse_mlpoly.rs:1535-1571) against
 *                     comm = bincode(R1CSCommitment) from vpin_spark_encode. */
int vpin_sat_verify(vpin_ctx* ctx, const uint8_t* proof, size_t proof_len, size_t num_cons, size_t num_vars,
                    const uint8_t* inputs, size_t num_inputs, const uint8_t inst_evals[96], const uint8_t* comm_para,
                    const uint8_t* comm_input);
int vpin_snark_verify(vpin_ctx* ctx, const uint8_t* proof, size_t proof_len, const uint8_t* comm, size_t comm_len,
                      const uint8_t* inputs, size_t num_inputs, const uint8_t* comm_para, const uint8_t* comm_input);
/* wall-clock spans of the last encode / snark prove on this thread, seconds: [0] encode
 * [1] derefs + commit  [2] network build  [3] product-layer proofs  [4] hash-layer proofs
 * [5] sat part  [6] whole prove  [7] unused */
void vpin_spark_last_timings(double out[8]);

/* ---- gadgets: the R1CS instances vPIN proves (host, no GPU needed) ----------------------- */
/* Owns the (A,B,C) triplets after Instance::new padding (Spartan/src/lib.rs:138-244) and the
 * three padded assignments vPIN commits to (para / input / all). */
typedef struct vpin_instance vpin_instance;
/* vPIN_proof_generation/src/point_addition.rs:67-327.  px,py,rx,ry: N x 32 LE bytes as in
 * rust_files/<label>/pointAdd/point_add_{px,py,rx,ry}_byte.json; rz: N bytes 0/1 (load_data_add.rs) */
int vpin_gadget_point_add(const uint8_t* px, const uint8_t* py, const uint8_t* rx, const uint8_t* ry,
                          const uint8_t* rz, size_t N, vpin_instance** out);
/* vPIN_proof_generation/src/point_mult.rs:61-704 (n = 128 bits, load_data.rs:62).
 * weights: N x 16 bytes (u128 little-endian, weight.json parsed as u128); px,py: N x 32 LE bytes */
int vpin_gadget_point_mult(const uint8_t* weights_le16, const uint8_t* px, const uint8_t* py, size_t N,
                           vpin_instance** out);
void vpin_instance_free(vpin_instance* g);
const vpin_r1cs* vpin_instance_r1cs(const vpin_instance* g);
size_t vpin_instance_num_cons_unpadded(const vpin_instance* g);
size_t vpin_instance_num_vars_unpadded(const vpin_instance* g);
const uint8_t* vpin_instance_vars_para(const vpin_instance* g);  /* num_vars (padded) x 32 B */
const uint8_t* vpin_instance_vars_input(const vpin_instance* g);
const uint8_t* vpin_instance_vars(const vpin_instance* g);
const uint8_t* vpin_instance_inputs(const vpin_instance* g);     /* num_inputs x 32 B or NULL */
/* Instance::is_sat (lib.rs:246-275): 1 satisfied, 0 not */
int vpin_instance_is_sat(const vpin_instance* g);
/* synthetic stand-in for the Python inference service's witness dump: count points k*G on the
 * curve E2 (src/convolution/Client.py:134-143), k from SplitMix64(seed); 32-byte LE x and y */
int vpin_synthetic_points(uint64_t seed, size_t count, uint8_t* out_x, uint8_t* out_y);

/* ---- gadgets built on the device ---------------------------------------------------------------
 * The same two gadgets with the instance never existing on the host: per-operation template replicated by
 * kernels into CSR/CSC, witness synthesis (one thread per operation), Instance::new's padding and column
 * remap, is_sat, and SNARK::encode's dense representation (addresses, read/audit timestamps) in closed form.
 * Replaces, inside the reference's timed span (proof_point_mult.rs:24-101): point_mult.rs:61-704,
 * point_addition.rs:67-327, Spartan/src/lib.rs:138-244, r1csinstance.rs:240-270, sparse_mlpoly.rs:232-265,368-438.
 * Results are bit-identical to vpin_gadget_point_* + vpin_r1cs_upload + vpin_spark_encode on the same inputs. */
typedef struct vpin_dev_instance vpin_dev_instance;
int vpin_gadget_point_add_dev(vpin_ctx* ctx, const uint8_t* px, const uint8_t* py, const uint8_t* rx, const uint8_t* ry,
                              const uint8_t* rz, size_t N, vpin_dev_instance** out);
int vpin_gadget_point_mult_dev(vpin_ctx* ctx, const uint8_t* weights_le16, const uint8_t* px, const uint8_t* py, size_t N,
                               vpin_dev_instance** out);
void vpin_dev_instance_free(vpin_ctx* ctx, vpin_dev_instance* g);
const vpin_r1cs_dev* vpin_dev_instance_r1cs(const vpin_dev_instance* g);
const vpin_table* vpin_dev_instance_vars_para(const vpin_dev_instance* g);
const vpin_table* vpin_dev_instance_vars_input(const vpin_dev_instance* g);
const vpin_table* vpin_dev_instance_vars(const vpin_dev_instance* g);
const uint8_t* vpin_dev_instance_inputs(const vpin_dev_instance* g); /* host bytes, num_inputs x 32 B or NULL */
size_t vpin_dev_instance_num_cons_unpadded(const vpin_dev_instance* g);
size_t vpin_dev_instance_num_vars_unpadded(const vpin_dev_instance* g);
size_t vpin_dev_instance_nnz(const vpin_dev_instance* g, int m);
/* push-order triplets of matrix m (0 = A, 1 = B, 2 = C) after Instance::new's column remap, copied to the host */
int vpin_dev_instance_triplets(vpin_ctx* ctx, const vpin_dev_instance* g, int m, uint32_t* row_out, uint32_t* col_out,
                               uint8_t* val_out /* nnz x 32 B */);
/* R1CSInstance::is_sat on the device: 1 satisfied, 0 not, < 0 error */
int vpin_dev_instance_is_sat(vpin_ctx* ctx, const vpin_dev_instance* g);
/* SNARK::encode for a device-built instance (same outputs as vpin_spark_encode) */
int vpin_spark_encode_dev(vpin_ctx* ctx, const vpin_dev_instance* g, vpin_spark_decomm** out, uint8_t* comm_out,
                          size_t comm_cap, size_t* comm_len);
size_t vpin_dev_instance_comm_bytes(const vpin_dev_instance* g);
size_t vpin_dev_instance_proof_max_bytes(const vpin_dev_instance* g);
/* proof_point_mult.rs:38-94 for a device-built instance: SNARK::encode + my_lib_prove in full */
int vpin_snark_prove_dev(vpin_ctx* ctx, const vpin_dev_instance* g, const uint8_t seed_commit64[64],
                         const uint8_t seed_proof64[64], uint8_t* proof_out, size_t proof_cap, size_t* proof_len,
                         uint8_t* comm_out, size_t comm_cap, size_t* comm_len, uint8_t* comm_para_out,
                         uint8_t* comm_input_out);

/* BulletReductionProof::prove, Spartan/src/nizk/bullet.rs:32-132, with the round challenges GIVEN (u_mont: log2(R)
 * Montgomery scalars) instead of drawn from a transcript, over the R stream generators of `g` only (the caller's
 * c*Q and blind*H terms are host work: nizk/mod.rs:447-531).  x = the vector being reduced (a in bullet.rs), a = the
 * public vector (b in bullet.rs), R scalars each.  Per round k: cLR_out[64k..] = <a_L,b_R> | <a_R,b_L>,
 * LR_out[64k..] = compressed <a_L,G_R> | <a_R,G_L>; at the end x_hat | a_hat and the compressed g_hat.  The device never
 * folds G (bullet.hip); classic != 0 forces the three-launch rounds that rows longer than 4096 scalars use, classic == 0
 * the fused one-launch rounds (VPIN_ESHAPE where R has none).  Kernel-level parity handle. */
int vpin_bullet_reduce(vpin_ctx* c, const vpin_gens* g, const uint8_t* x_mont, const uint8_t* a_mont, size_t R, const uint8_t* u_mont,
                       int classic, uint8_t* cLR_out, uint8_t* LR_out, uint8_t xhat_ahat_out[64], uint8_t ghat_out[32]);

/* ---- host-only entry points (no GPU needed) -------------------------------------------- */
/* shape of the instance vpin_gadget_point_mult* (is_mult != 0) / vpin_gadget_point_add* will build for n_ops operations:
 * padded num_cons and num_vars and the non-zero entries of A, B, C (point_mult.rs:61-67, point_addition.rs:67-70) -- enough
 * for vpin_sat_prepare / vpin_spark_prepare before the witness has been read */
int vpin_gadget_shape(int is_mult, size_t n_ops, size_t* num_cons, size_t* num_vars, size_t nnz[3]);
/* MultiCommitGens::new (Spartan/src/commitments.rs:20-38): first nb points of the stream */
int vpin_host_gens_derive(const char* label, size_t nb, uint8_t* out_xyzt /* nb*128 */);
/* Merlin: Transcript::new(proto); append_message(label,msg); challenge_bytes(clabel,out) */
int vpin_host_merlin_kat(const char* proto, const char* label, const uint8_t* msg, size_t n, const char* clabel,
                         uint8_t* out, size_t out_n);
/* Commitments::commit (commitments.rs:85-98) under MultiCommitGens::new(n,label), n <= 4 */
int vpin_host_commit(const char* label, const uint8_t* v_mont, size_t n, const uint8_t* blind_mont, uint8_t out[32]);

/* a*P + b*Q on the host (the verifier's variable-base multiplications, host/curve.h: width-5 NAF, shared doublings):
 * scalars in Montgomery form, points compressed; VPIN_EVERIFY when a point does not decode */
int vpin_host_scalar_mul2(const uint8_t a_mont[32], const uint8_t P[32], const uint8_t b_mont[32], const uint8_t Q[32], uint8_t out[32]);

/* self-test of the pinned-memory mailbox framing used between resident kernels and the host (sequence number + checksum per
 * scalar; a torn or mixed publication is rejected and read again): 0 = as expected */
int vpin_host_mailbox_selftest(void);

/* ---- built-in kernel timing (HIP events on the ctx stream) ------------------------ */
/* kernel classes */
#define VPIN_K_SC_CUBIC 0
#define VPIN_K_SC_QUAD 1
#define VPIN_K_SC_BIND 2
#define VPIN_K_SC_CUBIC_FUSED 3
#define VPIN_K_SC_QUAD_FUSED 4
#define VPIN_K_EQ 5
#define VPIN_K_MSM 6
#define VPIN_K_SC_TAIL 7 /* single-workgroup tail rounds (<= 512 pairs), latency bound */
#define VPIN_K_SPARK_ROUND 8 /* batched cubic rounds of the product / dot-product circuits (SPARK) */
#define VPIN_K_SPARK_BUILD 9 /* SPARK gathers, hash layer, product-tree levels, slice evaluations */
#define VPIN_K_SPARK_ROUND_BIG 10 /* the subset of class 8 with >= 2^20 pairs per circuit: the streaming regime (also counted in 8) */
#define VPIN_K_MSM_ROWS 11 /* the subset of class 6 that is a row commitment of >= 128 rows (msm_rows_kernel; also counted in 6) */
#define VPIN_K_SPARK_TAIL 12 /* persistent tail launches (all the small rounds of one layer; time includes the host's turn-arounds) */
#define VPIN_K_COUNT 16
typedef struct {
  uint64_t launches;
  double ms;         /* sum of event-measured durations */
  double alg_bytes;  /* sum of algorithmic bytes moved by those launches */
  double units;      /* classes 8 / 10: pair evaluations (circuits x pairs); class 11 under vpin_prof_enable(ctx, 2): affine
                      * table additions those launches performed */
} vpin_kstat;
/* on = 1: HIP-event bracketing of the kernel classes; on = 2: additionally count the table additions of the row
 * commitments (an extra counting kernel per commitment: for roofline passes, not for timed regions) */
int vpin_prof_enable(vpin_ctx* ctx, int on);
int vpin_prof_reset(vpin_ctx* ctx);
/* resolves outstanding events; stats must have VPIN_K_COUNT entries */
int vpin_prof_read(vpin_ctx* ctx, vpin_kstat* stats);

#ifdef __cplusplus
}
#endif
#endif
