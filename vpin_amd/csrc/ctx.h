// ctx.h -- internal definitions behind the opaque handles of include/vpin_hip.h
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/vpin_hip.h"
#include "fq_dev.h"

namespace vpin {

void set_last_error(const char* what, hipError_t e);

#define VPIN_HIP_TRY(expr)                         \
  do {                                             \
    hipError_t e_ = (expr);                        \
    if (e_ != hipSuccess) {                        \
      ::vpin::set_last_error(#expr, e_);           \
      return VPIN_EHIP;                            \
    }                                              \
  } while (0)

struct ProfRec {
  int kclass;
  double bytes;
  hipEvent_t start, stop;
  int also = -1;  // a second class the same launch is counted in (a subset class)
  double units = 0.0;
};

}  // namespace vpin

struct vpin_ctx;

struct vpin_table {
  vpin::fq* d = nullptr;  // device pointer
  size_t len = 0;         // live length (halves on bind)
  size_t cap = 0;         // allocated length
  bool owned = true;
  vpin_ctx* owner = nullptr;  // context whose pool `d` came from (vpin_table_free(NULL, t) returns it there)
};

struct vpin_gens;

struct vpin_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cus = 256;        // compute units the stream may use (the enabled ones of a CU-masked context)
  bool cu_masked = false;   // created by vpin_ctx_create_cumask, or on the second stream of vpin_ctx_set_cumask_after_phase1
  // vpin_ctx_set_cumask_after_phase1: a second, CU-masked stream a proof moves to after its phase-1 sum-check (capi.hip
  // ctx_enter_alt / ctx_leave_alt; `stream` is always the stream in use)
  hipStream_t stream_main = nullptr, stream_alt = nullptr;
  hipEvent_t stream_switch_ev = nullptr;
  int cus_main = 256, cus_alt = 0;
  // scratch for block partials of the round reductions and the final 3 scalars
  vpin::fq* d_partials = nullptr;
  size_t partials_cap = 0;  // in fq elements
  vpin::fq* d_out = nullptr;  // 8 fq
  vpin::fq* h_out = nullptr;  // pinned, 8 fq
  // profiling
  bool prof = false;
  bool prof_count_adds = false;            // vpin_prof_enable(ctx, 2)
  unsigned long long* d_add_count = nullptr;  // device counter of msm_count_adds_kernel
  std::vector<vpin::ProfRec> recs;
  std::vector<hipEvent_t> free_events;
  vpin_kstat stats[VPIN_K_COUNT] = {};
  // device memory pool: freed blocks are kept by size class and reused (hipMalloc/hipFree cost
  // hundreds of microseconds and hipFree synchronises the device; one proof makes ~100 of them).
  // All work is issued on `stream`, so a recycled block is only touched by later-ordered work.
  // pool_mu guards both maps: another context's allocation failure may hand this context's cached blocks back
  // to the driver (dev_alloc's out-of-memory path)
  std::mutex pool_mu;
  std::atomic<int> pins{0};  // threads other than the owner working on this context (capi.hip): destroy waits for them
  std::map<size_t, std::vector<void*>> pool_free_lists;
  std::unordered_map<void*, size_t> pool_sizes;
  // host-side prover state (generator sets per polynomial size), owned by prover.cpp
  void* prover_cache = nullptr;
  void (*prover_cache_free)(vpin_ctx*) = nullptr;
  // SPARK (spark.cpp / spark.hip): generator views per label, pinned staging for per-round results
  void* spark_cache = nullptr;
  void (*spark_cache_free)(vpin_ctx*) = nullptr;
  // verifier (verify.cpp): its own generator sets per polynomial size (derived once per context, not once per proof)
  void* verify_cache = nullptr;
  void (*verify_cache_free)(vpin_ctx*) = nullptr;
  vpin::fq* h_spark = nullptr;  // pinned, kSparkPinned fq; the last element's first word is the completion flag
  uint32_t* d_spark_cnt = nullptr;  // device: per-instance and global "blocks done" counters (self-resetting)
  uint32_t spark_seq = 0;           // sequence number of the last flagged launch group
  int round_split = 0, round_split_grid = 0, round_split_ncirc = 0;  // the current group sums its partials in round_finish_kernel
  int round_group_ndotp = 0;  // dot-product halves announced for the current launch group (spark_prod_round)
  uint32_t tail_seq = 0;            // persistent tail kernel (spark.hip): sequence base of the current / next launch
  // tail rounds on several workgroups per circuit (round 6): per-instance arrival counters (self-resetting), the workgroups'
  // partial sums, this context's share of the device's resident-workgroup budget
  uint32_t* d_tail_cnt = nullptr;
  vpin::fq* d_tail_red = nullptr;
  int tail_reserved = 0;
  std::atomic<int> tail_rounds{0};  // > 0 while a persistent tail kernel is resident (read by other threads: dev_alloc's reclaim)
  vpin::fq tail_sums[3 * 18];   // host copies of the current tail round's results (assembled from the mailbox pieces)
  vpin::fq tail_final[6 * 18];
  unsigned long long pip_row_chunks = 0;    // row chunks launched by the bucket method so far (vpin_ctx_pip_row_chunks: tests)
  unsigned long long strip_rows_taken = 0;  // rows handed to msm_strip_kernel so far (vpin_ctx_strip_rows_taken: tests)
  bool low_memory = false;     // vpin_ctx_set_low_memory: trade ~1 % of a large proof's time for a third less working set
  bool shared_device = false;  // other contexts prove on this device at the same time (vpin_ctx_set_shared_device)
  int expected_proofs = 0;     // proofs the window tables built through this context will serve; 0 = many (vpin_ctx_set_expected_proofs)
  double gens_scalars_per_proof = 0.0;  // set by the caller of vpin_gens_shared: full-size scalars one proof commits under the table
  void* h_bullet = nullptr;  // pinned staging of the bullet reduction's per-round results (bullet.hip), 64 KiB
  uint32_t bullet_seq = 0;   // sequence number of the last fused bullet round (mailbox_dev.h)
  // one proof over several GPUs (include/vpin_hip.h, vpin_ctx_set_comm): proofs on this context are collective calls
  std::atomic<vpin_comm*> comm_pub{nullptr};  // == comm, for other threads (dev_alloc's reclaim skips the peers of a collective proof)
  vpin_comm* comm = nullptr;
  volatile int* progress_flag = nullptr;  // optional host word: 1 when a SNARK's sat part is done, 2 after its derefs commitment
};

namespace vpin {

// RAII-ish helper used by launchers: brackets a launch with events when profiling is on.
struct ProfScope {
  vpin_ctx* ctx;
  int rec = -1;
  ProfScope(vpin_ctx* c, int kclass, double bytes, int also = -1, double units = 0.0);
  ~ProfScope();
};

// Move the context's work to its CU-masked second stream / back (no-ops without one).  Work issued afterwards is ordered
// behind everything issued before (an event), so pooled blocks and tables carry over; the caller must not have a
// persistent tail kernel resident.  AltStreamGuard: a proof entry point leaves the second stream however it returns.
void ctx_enter_alt(vpin_ctx* c);
void ctx_leave_alt(vpin_ctx* c);
struct AltStreamGuard {
  vpin_ctx* c;
  explicit AltStreamGuard(vpin_ctx* ctx) : c(ctx) {}
  ~AltStreamGuard() { ctx_leave_alt(c); }
};

// pooled device allocation (see vpin_ctx::pool_free_lists)
int dev_alloc(vpin_ctx* c, size_t bytes, void** out);
hipError_t driver_malloc(void** p, size_t bytes);  // hipMalloc behind a free-memory check (large blocks): no out-of-memory call into the runtime
void note_driver_alloc(size_t bytes);  // every hipMalloc the library makes outside dev_alloc reports here (vpin_driver_alloc_stats)
void dev_free(vpin_ctx* c, void* p);
int live_ctx_count();      // contexts of this process that exist right now
double host_cpu_quota();   // CPUs the process may use at once (cgroup CFS quota, else the hardware's)
void dev_free_owned(vpin_ctx* owner, vpin_ctx* fallback, void* p);  // to the owner's pool, whichever context frees the handle
void dev_pool_release(vpin_ctx* c);
// one block the caller holds (from dev_alloc) straight back to the driver, not to the pool; the caller has synchronised the
// work that used it
void dev_release_block(vpin_ctx* c, void* p);

// scoped pooled buffer
struct DevBuf {
  vpin_ctx* c;
  void* p = nullptr;
  explicit DevBuf(vpin_ctx* ctx) : c(ctx) {}
  ~DevBuf() { if (p) dev_free(c, p); }
  int alloc(size_t bytes) { return dev_alloc(c, bytes, &p); }
  void release() { if (p) { dev_free(c, p); p = nullptr; } }  // back to the pool before the scope ends
};

int table_alloc_uninit(vpin_ctx* c, size_t len, vpin_table** out);

// split-phase pair commitment (msm.hip)
struct CommitPairState;
// row0 / nrows / row_step: the rows row0, row0 + row_step, .. (nrows of them) of the L rows (one commitment split across
// ranks); finish then takes those rows' blinds, in that order, and writes nrows results per output
int commit_pair_begin(vpin_ctx* c, const vpin_gens* g, const vpin_table* Za, const vpin_table* Zb, size_t L,
                      CommitPairState** out, size_t row0 = 0, size_t nrows = (size_t)-1, size_t row_step = 1);
int commit_pair_finish(vpin_ctx* c, const vpin_gens* g, CommitPairState* st, const uint8_t* blinds_a, const uint8_t* blinds_b,
                       size_t blind_base, uint8_t* out_a, uint8_t* out_b, uint8_t* out_sum);

// asynchronous round launch / collect (sumcheck.hip), for the host prover's overlap
int sc_round_launch(vpin_ctx* c, int K, vpin_table* const* tabs, const uint8_t* r);
int sc_round_wait(vpin_ctx* c, int K, uint8_t* out);
int sc_final_claims(vpin_ctx* c, vpin_table* const* tables, int k, const uint8_t r[32], uint8_t* out);
int sc_cubic3_launch(vpin_ctx* c, vpin_table* const* t, const vpin_table* pyramid, int ell, int level, const uint8_t* r,
                     bool lead = false);

// strided views for one sum-check over several GPUs (r1cs.hip): rows / columns / entries = r0 (mod step)
int r1cs_multiply_vec_strided(vpin_ctx* c, const vpin_r1cs_dev* d, const vpin_table* z, size_t r0, size_t step, vpin_table** Az,
                              vpin_table** Bz, vpin_table** Cz);
int table_take_strided(vpin_ctx* c, const vpin_table* src, size_t r0, size_t step, vpin_table** out);
int r1cs_eval_table_strided(vpin_ctx* c, const vpin_r1cs_dev* d, const vpin_table* evals_rx, const uint8_t r_abc[96], size_t r0,
                            size_t step, vpin_table** out);

// DensePolynomial::bound split by row blocks over the ranks of c->comm (poly.hip)
// z_rows != nullptr: the rows rank, rank + world, .. stored densely (otherwise a contiguous block of Z's rows per rank)
int poly_bound_dist(vpin_ctx* c, const vpin_table* Z, const uint8_t* Lvec, size_t L_size, uint8_t* out_LZ, const fq* z_rows = nullptr);

// the hash layer's slice evaluations and DensePolynomial::bound from one pass over a combined polynomial (poly.hip):
// Z = S slices of N = T * Rs scalars; Ltop = eq(r_top, .) (T scalars), Rv = eq(r_bot, .) (Rs scalars), host memory, Montgomery.
// d_LZs (S * Rs scalars, device) <- LZ_s; ev_out (S scalars, host) <- Z_s(r).  Then, with the combining challenges known,
// out_LZ (Rs scalars, host) <- sum_s coef[s] LZ_s.
// Z32 / n32: the first n32 of the S slices held as u32 (addresses / timestamps: their field images are never formed); Z: the others
int slices_bound(vpin_ctx* c, const uint32_t* Z32, int n32, const fq* Z, size_t N, int S, size_t Rs, const uint8_t* Ltop, size_t T,
                 const uint8_t* Rv, fq* d_LZs, uint8_t* ev_out);
int slices_combine(vpin_ctx* c, const fq* d_LZs, int S, size_t Rs, const uint8_t* coef, uint8_t* out_LZ);

inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }

// few-row fixed-base MSM over device-resident scalars (msm.hip): rows x ncols Montgomery scalars ->
// rows x vpin_gens_msm_parts_count(ncols) partial points (canonical X|Y|Z|T) in host memory; synchronises
int gens_msm_parts_dev(vpin_ctx* c, const vpin_gens* g, const fq* d_scalars, size_t rows, size_t ncols, uint8_t* parts_xyzt);
// the same without the synchronisation (the caller provides the device scratch and waits on the stream itself)
// the derefs commitment with each matrix's hot column taken out of the table walks (msm.hip msm_rows_hot_kernel)
// row0 / nrows / row_step: the rows row0, row0 + row_step, .. (out_compressed then holds nrows results); default all rows
int hyrax_commit_derefs_hot(vpin_ctx* c, const vpin_gens* g, const vpin_table* Z, size_t L, size_t N, const uint32_t* const col_idx[3],
                            const uint32_t hot[3], const fq* e_ry, uint8_t* out_compressed, size_t row0 = 0, size_t nrows = (size_t)-1,
                            size_t row_step = 1, const fq* z_rows = nullptr);
// the same rows of a commitment without blinds (DensePolynomial::commit(gens, None)) through the plain row kernel
// z_rows != nullptr (both functions): the nrows rows stored densely instead of being read from Z (which then only gives the shape)
int hyrax_commit_rows_strided(vpin_ctx* c, const vpin_gens* g, const vpin_table* Z, size_t L, size_t row0, size_t nrows, size_t row_step,
                              uint8_t* out_compressed, const fq* z_rows = nullptr);
size_t gens_msm_parts_scratch_bytes(size_t rows, size_t ncols);
int gens_msm_parts_launch(vpin_ctx* c, const vpin_gens* g, const fq* d_scalars, size_t rows, size_t ncols, void* scratch,
                          uint8_t* parts_xyzt, bool host_mapped = false);

// row commitments by Pippenger's bucket method from a compact array of the generators (msm_pip.hip); cbits 0: by row length
struct ge_niels;
struct ge_ext;
int pip_rows(vpin_ctx* c, const ge_niels* d_gn, const fq* dZ, size_t rows, size_t stride, size_t ncols, const fq* d_extra, int n_extra,
             int cbits, ge_ext* d_points, unsigned long long* d_adds);
int pip_default_bits(size_t n);

// device side of the bullet reduction (bullet.hip)
struct BulletState;
int bullet_begin(vpin_ctx* c, const uint8_t* x_mont, const uint8_t* a_mont, size_t R, BulletState** out);
int bullet_round_begin(vpin_ctx* c, const vpin_gens* g, BulletState* st, size_t n, uint8_t* parts_xyzt, uint8_t cLR[64]);
int bullet_round_end(vpin_ctx* c);
uint8_t* bullet_pinned(vpin_ctx* c);  // 64 KiB of pinned host memory owned by the context (nullptr on failure)
int bullet_fold(vpin_ctx* c, BulletState* st, size_t n, const uint8_t u[32], const uint8_t u_inv[32]);
int bullet_finish(vpin_ctx* c, const vpin_gens* g, BulletState* st, uint8_t xhat_ahat[64], uint8_t* parts_xyzt);
void bullet_free(vpin_ctx* c, BulletState* st);
// one launch per round (msm.hip bullet_step_kernel) when the state allows it: R a multiple of 32, at most 32768
bool bullet_fused(const BulletState* st);
int bullet_step(vpin_ctx* c, const vpin_gens* g, BulletState* st, size_t n, const uint8_t* u_prev, const uint8_t* u_inv_prev,
                uint8_t cLR[64]);
// the partial points (128 B each, X|Y|Z|T) of the last bullet_step (half length n) / bullet_finish_fused (n = 0) the host has to
// add for row 0 = L or 1 = R, valid after bullet_round_end; out has room for 256 pointers; returns their number
size_t bullet_part_ptrs(vpin_ctx* c, const BulletState* st, size_t n, int row, const uint8_t** out);
int bullet_finish_fused(vpin_ctx* c, const vpin_gens* g, BulletState* st, const uint8_t u[32], const uint8_t u_inv[32],
                        uint8_t xhat_ahat[64]);
int bullet_step_launch(vpin_ctx* c, const vpin_gens* g, const fq* a_prev, const fq* b_prev, fq* a_next, fq* b_next, fq* sj, size_t n,
                       size_t R, bool fold, bool finish, const uint8_t* u, const uint8_t* u_inv, uint8_t* parts_pinned,
                       uint32_t* up_pinned, uint32_t seq, void* dev_parts);

// VPIN_CLI_TRACE=1: wall-clock laps of the cold (one-shot CLI) path on stderr; each lap drains the stream
struct TraceLap {
  vpin_ctx* c;
  const char* who;
  bool on;
  std::chrono::steady_clock::time_point t;
  TraceLap(vpin_ctx* ctx, const char* w) : c(ctx), who(w), on(getenv("VPIN_CLI_TRACE") != nullptr), t(std::chrono::steady_clock::now()) {}
  void operator()(const char* what) {
    if (!on) return;
    if (c) (void)hipStreamSynchronize(c->stream);
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[%s] %-22s %9.1f ms\n", who, what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

}  // namespace vpin
