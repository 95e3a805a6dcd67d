// spark.cpp -- host orchestration of the SPARK half of vPIN's Spartan SNARK on one MI355X:
// the computation commitment and the sparse-polynomial evaluation proof that follow the sat proof
// on the same transcript.
//
// C++ counterpart of (names mirror the reference):
//   Spartan/src/lib.rs:294-359                    SNARKGens::new, SNARK::encode
//   Spartan/src/r1csinstance.rs:29-49,309-372     R1CSCommitmentGens, R1CSInstance::commit, R1CSEvalProof
//   Spartan/src/sparse_mlpoly.rs:42-1572          Derefs, AddrTimestamps, multi_commit, Layers,
//                                                 ProductLayerProof, HashLayerProof, PolyEvalNetworkProof,
//                                                 SparseMatPolyEvalProof
//   Spartan/src/product_tree.rs:258-385           ProductCircuitEvalProofBatched::prove
//   Spartan/src/sumcheck.rs:248-425               SumcheckInstanceProof::prove_cubic_batched
//   vPIN_proof_generation/src/commit_test.rs:59-133  my_lib_prove (whole)
// The gathers, the hash layer, the product trees, every sum-check round, the slice evaluations and the
// commitments run in spark.hip / msm.hip / poly.hip; this file owns the transcript, the per-round
// scalars and the bincode image.  It shares no code with the test-side checker.
#include <cstdio>

#include "host/prover_common.h"
#include "gadget_dev.h"
#include "spark_dev.h"

namespace {

using namespace vpin_host;
using namespace vpin_prover;

// VPIN_SPARK_TRACE=1: finer spans of one proof on stderr (development aid)
static bool spark_trace() { static const bool on = getenv("VPIN_SPARK_TRACE") != nullptr; return on; }
struct TraceSpan {
  const char* name; Clock::time_point t0;
  explicit TraceSpan(const char* n) : name(n), t0(Clock::now()) {}
  ~TraceSpan() { if (spark_trace()) fprintf(stderr, "[spark] %-28s %8.3f ms\n", name, secs(t0, Clock::now()) * 1e3); }
};

// ---- generators: one stream under b"gens_r1cs_eval", three PolyCommitmentGens views -----------

struct SparkGens {
  std::vector<Point> g;                              // host copy of the stream prefix derived so far
  std::map<size_t, std::unique_ptr<PcGens>> views;  // by num_vars of the committed polynomial
};

static void spark_cache_free(vpin_ctx* c) {
  auto* sg = static_cast<SparkGens*>(c->spark_cache);
  if (!sg) return;
  delete sg;  // the device tables belong to the shared registry
  c->spark_cache = nullptr;
}

// table budget for the b"gens_r1cs_eval" stream: its largest user (the derefs commitment, 6N full-width
// scalars per proof) is the single most expensive step of a SNARK, and every window bit saves ~8 % of
// it, so on a 288 GB part the table gets up to 80 GB (11-bit windows for 32 770 generators)
static size_t spark_budget_gb(vpin_ctx* c) {
  // a process that proves once and exits sizes its tables by use, far below either budget (msm.hip gens_build), and the
  // first hipMemGetInfo of a process costs 35-50 ms: not asked then (a failed allocation still halves the table)
  if (c->expected_proofs > 0) return (size_t)24;
  static const size_t gb = [c] {
    const char* e = getenv("VPIN_SPARK_GENS_BUDGET_GB");
    if (e && atoi(e) > 0) return (size_t)atoi(e);
    // from the memory that is FREE when the first table of the stream is built (another tenant's allocations, a smaller
    // part); msm.hip's gens_build additionally caps every table at a third of the free memory of its moment
    size_t free_b = 0, total_b = 0;
    (void)hipSetDevice(c->device);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (size_t)24;
    return free_b >= ((size_t)200 << 30) ? (size_t)80 : (size_t)24;
  }();
  return gb;
}

// PolyCommitmentGens::new(ell, b"gens_r1cs_eval") (dense_mlpoly.rs:26-33; lib.rs:315-321)
static int get_view(vpin_ctx* c, size_t ell, const PcGens** out) {
  if (!c->spark_cache) { c->spark_cache = new SparkGens(); c->spark_cache_free = spark_cache_free; }
  auto* sg = static_cast<SparkGens*>(c->spark_cache);
  auto it = sg->views.find(ell);
  if (it == sg->views.end()) {
    const size_t left = ell / 2, R = (size_t)1 << (ell - left), nb = R + 2;
    // every MultiCommitGens::new(n, label) is a prefix of the same SHAKE stream
    vpin::TraceLap lap(nullptr, "spark get_view");
    if (sg->g.size() < nb) derive_gens(sg->g, nb, "gens_r1cs_eval", c);
    lap("derive_gens (host)");
    std::unique_ptr<PcGens> v(new PcGens());
    v->ell = ell; v->L = (size_t)1 << left; v->R = R;
    int rc = vpin_gens_shared(c, "gens_r1cs_eval", nullptr, nb, 0, &v->dev);
    if (rc == VPIN_EINVAL) {
      std::vector<uint8_t> xyzt(128 * nb);
#pragma omp parallel for schedule(static) num_threads(host_threads())
      for (long i = 0; i < (long)nb; i++) sg->g[i].to_xyzt(xyzt.data() + 128 * (size_t)i);
      // full-size scalars one proof commits under this table: the derefs polynomial's six non-zero slices, 6N of the 16N
      // entries of the largest view (the computation commitment's entries are addresses, counters and small constants)
      c->gens_scalars_per_proof = (double)((size_t)1 << ell) * 0.375;
      rc = vpin_gens_shared(c, "gens_r1cs_eval", xyzt.data(), nb, spark_budget_gb(c), &v->dev);
    }
    if (rc) return rc;
    lap("device table");
#pragma omp parallel for schedule(static, 1) num_threads(2)
    for (int k = 0; k < 2; k++) {
      if (k == 0) v->fb_gR = FixedBase(sg->g[R]);
      else v->fb_h = FixedBase(sg->g[R + 1]);
    }
    lap("fixed bases (host)");
    v->bind_views();
    it = sg->views.emplace(ell, std::move(v)).first;
  }
  *out = it->second.get();
  return VPIN_OK;
}

struct Shape { size_t nx, ny, N, M, v_ops, v_mem, v_derefs; };

static size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }

static Shape shape_of(size_t num_cons, size_t num_vars, const size_t nnz[3]) {
  Shape s;
  s.nx = log2z(num_cons);
  s.ny = log2z(2 * num_vars);
  s.N = 1;
  for (int m = 0; m < 3; m++) s.N = std::max(s.N, next_pow2(nnz[m]));  // get_num_nz_entries (sparse_mlpoly.rs:364)
  s.M = (size_t)1 << std::max(s.nx, s.ny);
  // SparseMatPolyCommitmentGens::new (sparse_mlpoly.rs:300-329), batch_size = 3
  s.v_ops = log2z(s.N) + 4;
  s.v_mem = std::max(s.nx, s.ny) + 1;
  s.v_derefs = log2z(s.N) + 3;
  return s;
}

// DensePolynomial::commit(gens, None) (dense_mlpoly.rs:193-218): zero blinds
static int commit_noblind(vpin_ctx* c, const PcGens* pc, const vpin_table* Z, std::vector<CG>& out) {
  std::vector<uint8_t> zeros(pc->L * 32, 0);
  out.resize(pc->L);
  return vpin_hyrax_commit(c, pc->dev, Z, zeros.data(), pc->L, pc->R + 1, out[0].b);
}

static void append_polycomm(Transcript& tr, const char* label, const std::vector<CG>& C) {
  // PolyCommitment::append_to_transcript (dense_mlpoly.rs:305-313)
  tr.append_message(label, "poly_commitment_begin");
  for (auto& p : C) tr.append_point("poly_commitment_share", p.b);
  tr.append_message(label, "poly_commitment_end");
}

static void append_unipoly(Transcript& tr, const Fq* coeffs, int n) {
  // AppendToTranscript for UniPoly (unipoly.rs:112-120)
  tr.append_message("poly", "UniPoly_begin");
  for (int i = 0; i < n; i++) tr.append_scalar("coeff", coeffs[i]);
  tr.append_message("poly", "UniPoly_end");
}

static void w_scalars(Writer& w, const Fq* v, size_t n) { w.u64(n); for (size_t i = 0; i < n; i++) w.scalar(v[i]); }

// fold 2^k evaluations with bound_poly_var_bot in reverse challenge order (sparse_mlpoly.rs:104-109)
static Fq combine_bot(std::vector<Fq> e, const std::vector<Fq>& ch) {
  size_t n = e.size();
  for (size_t ii = ch.size(); ii-- > 0;) {
    n /= 2;
    for (size_t i = 0; i < n; i++) e[i] = e[2 * i] + ch[ii] * (e[2 * i + 1] - e[2 * i]);
  }
  return e[0];
}

// PolyEvalProof::prove with blinds None, blind_Zr None (dense_mlpoly.rs:326-379)
// z_rows (one proof over several GPUs, split by residue class): this rank's rows rank, rank + world, .. of Z stored densely; Z then
// only gives the shape
// lz_pre / rv_pre (single GPU, SlicePass below): LZ = L^T Z and R = eq(r[left..], .) already known from the pass that evaluated
// the slices -- the table is not read again
static int polyeval_prove_plain(vpin_ctx* c, const PcGens& pc, const vpin_table* Z, const std::vector<Fq>& r, const Fq& Zr,
                                Transcript& tr, Transcript& tape, DpLog& out, const vpin::fq* z_rows = nullptr,
                                const std::vector<Fq>* lz_pre = nullptr, const std::vector<Fq>* rv_pre = nullptr) {
  if (r.size() != pc.ell || (!(lz_pre && rv_pre) && (!Z || Z->len != ((size_t)1 << pc.ell)))) return VPIN_ESHAPE;
  tr.append_protocol_name("polynomial evaluation proof");
  const size_t left = pc.ell / 2, right = pc.ell - left;
  if (lz_pre && rv_pre) {  // (Z may be null: the table is not needed)
    if (lz_pre->size() != pc.R || rv_pre->size() != pc.R) return VPIN_ESHAPE;
    TraceSpan ts("  dplog");
    return dplog_prove(c, pc, tr, tape, *lz_pre, Fq::zero(), *rv_pre, Zr, Fq::zero(), out);
  }
  std::vector<Fq> Lv(pc.L), Rv(pc.R), LZ(pc.R);
  host_eq(r.data(), left, Lv.data());
  host_eq(r.data() + left, right, Rv.data());
  int rc;
  {
    TraceSpan ts("  poly_bound");
    // one proof over several GPUs: every rank sums its block of rows, the partial vectors are all-gathered on the device
    if (c->comm && c->comm->world > 1) rc = vpin::poly_bound_dist(c, Z, B(Lv.data()), pc.L, B(LZ.data()), z_rows);
    else rc = vpin_poly_bound(c, Z, B(Lv.data()), pc.L, B(LZ.data()));
  }
  if (rc) return rc;
  if ((rc = vpin::comm_mark(c, "hash_poly_bound"))) return rc;
  TraceSpan ts("  dplog");
  return dplog_prove(c, pc, tr, tape, LZ, Fq::zero(), Rv, Zr, Fq::zero(), out);
}

// Single GPU: the slice evaluations of a combined polynomial and, later, the LZ vector of its evaluation proof from ONE pass
// over the table (poly.hip slices_bound / slices_combine).  nbits = log2 of the slice count, used = slices that are not
// identically zero, rand = the point the slices are evaluated at (log2 of the slice length challenges).
struct SlicePass {
  vpin::DevBuf lzs;
  std::vector<Fq> Rv;
  int used = 0, nbits = 0;
  bool on = false;
  explicit SlicePass(vpin_ctx* c) : lzs(c) {}
  static bool fits(const PcGens& pc, int nbits, const std::vector<Fq>& rand) {
    const bool off = getenv("VPIN_HASH_TWO_PASS") != nullptr;  // A/B and tests: the separate evaluation and bound passes (read per proof)
    return !off && pc.ell / 2 >= (size_t)nbits && rand.size() + (size_t)nbits == pc.ell;
  }
  // table32 / n32: the first n32 slices as u32 (the decommitment's addresses and timestamps); table: the other used - n32
  int run(vpin_ctx* c, const PcGens& pc, const uint32_t* table32, int n32, const vpin::fq* table, int nbits_, int used_,
          const std::vector<Fq>& rand, Fq* ev) {
    nbits = nbits_; used = used_;
    const size_t left = pc.ell / 2, T = pc.L >> nbits, N = (size_t)1 << rand.size();
    std::vector<Fq> Ltop(T);
    host_eq(rand.data(), left - (size_t)nbits, Ltop.data());
    Rv.resize(pc.R);
    host_eq(rand.data() + (left - (size_t)nbits), pc.ell - left, Rv.data());
    if (lzs.alloc((size_t)used * pc.R * 32)) return VPIN_ENOMEM;
    int rc = vpin::slices_bound(c, table32, n32, table, N, used, pc.R, B(Ltop.data()), T, B(Rv.data()), (vpin::fq*)lzs.p, B(ev));
    on = rc == VPIN_OK;
    return rc;
  }
  // LZ of the combined polynomial at (ch, rand): sum_s eq(ch, s) LZ_s
  int combine(vpin_ctx* c, const PcGens& pc, const std::vector<Fq>& ch, std::vector<Fq>& LZ) {
    std::vector<Fq> coef((size_t)1 << nbits);
    host_eq(ch.data(), (size_t)nbits, coef.data());
    LZ.resize(pc.R);
    return vpin::slices_combine(c, (const vpin::fq*)lzs.p, used, pc.R, B(coef.data()), B(LZ.data()));
  }
};

// ---- one proof over several GPUs: who owns which circuit, and the per-round exchange -------------------------------
// The 12 "ops" circuits and the 6 dot-product halves are proven in the same rounds, so they are dealt together (longest
// processing time first; an ops circuit folds 2 tables over all layers = 4 units, a dot-product half 3 tables on layer 0 = 3
// units); the 4 "mem" circuits form their own phase and go to the ranks with the least ops work.  Every rank computes
// the same plan.  Ranks own whole circuits: leaves, tree, every round and the persistent tails of a circuit stay on one
// GPU, and per round only its three scalars leave it.
struct DistPlan {
  vpin_comm* cm = nullptr;
  int rank = 0, world = 1;
  int owner_ops[12], owner_dotp[6], owner_mem[4];
  std::vector<std::vector<int>> ops_of, dotp_of, mem_of;  // per rank, ascending
  int max_ops_inst = 0, max_mem_inst = 0;                 // most instances (circuits + halves) any rank owns per phase
  int lw = 0;                                             // log2(world) when the proof is split by residue class, else 0
};

static void make_plan(int world, int owner_ops[12], int owner_dotp[6], int owner_mem[4]) {
  std::vector<long> load(world, 0);
  auto least = [&](const std::vector<long>& key2) {
    int best = 0;
    for (int r = 1; r < world; r++)
      if (load[r] < load[best] || (load[r] == load[best] && key2[r] < key2[best])) best = r;
    return best;
  };
  const std::vector<long> none(world, 0);
  for (int t = 0; t < 12; t++) { int r = least(none); owner_ops[t] = r; load[r] += 4; }
  for (int k = 0; k < 6; k++) { int r = least(none); owner_dotp[k] = r; load[r] += 3; }
  std::vector<long> ops_load = load;
  std::fill(load.begin(), load.end(), 0);
  for (int t = 0; t < 4; t++) { int r = least(ops_load); owner_mem[t] = r; load[r] += 1; }
}

static void plan_init(DistPlan& pl, vpin_comm* cm) {
  pl.cm = cm; pl.rank = cm->rank; pl.world = cm->world;
  make_plan(pl.world, pl.owner_ops, pl.owner_dotp, pl.owner_mem);
  pl.ops_of.assign(pl.world, {}); pl.dotp_of.assign(pl.world, {}); pl.mem_of.assign(pl.world, {});
  for (int t = 0; t < 12; t++) pl.ops_of[pl.owner_ops[t]].push_back(t);
  for (int k = 0; k < 6; k++) pl.dotp_of[pl.owner_dotp[k]].push_back(k);
  for (int t = 0; t < 4; t++) pl.mem_of[pl.owner_mem[t]].push_back(t);
  for (int r = 0; r < pl.world; r++) {
    pl.max_ops_inst = std::max(pl.max_ops_inst, (int)(pl.ops_of[r].size() + pl.dotp_of[r].size()));
    pl.max_mem_inst = std::max(pl.max_mem_inst, (int)pl.mem_of[r].size());
  }
}

// `per` scalars of each of this rank's instances (its circuits in ascending order, then -- with_dotp -- its dot-product
// halves) -> the same for ALL instances in the single-GPU slot order: circuit g at global[per*g], half k at global[per*(12+k)]
static int dist_exchange(vpin_ctx* c, const DistPlan& pl, bool mem, bool with_dotp, const Fq* local, int per, Fq* global,
                         const char* tag = nullptr) {
  const int maxi = mem ? pl.max_mem_inst : pl.max_ops_inst;
  const size_t chunk = (size_t)maxi * per;
  std::vector<Fq> sb(chunk, Fq::zero()), rb(chunk * pl.world);
  const auto& mine = mem ? pl.mem_of[pl.rank] : pl.ops_of[pl.rank];
  const size_t nloc = mine.size() + (with_dotp ? pl.dotp_of[pl.rank].size() : 0);
  if (nloc) memcpy(sb.data(), local, nloc * per * 32);
  int rc = vpin::comm_allgather_ctx(c, sb.data(), rb.data(), chunk * 32, tag);
  if (rc) return rc;
  for (int r = 0; r < pl.world; r++) {
    const Fq* src = rb.data() + (size_t)r * chunk;
    for (int g : (mem ? pl.mem_of[r] : pl.ops_of[r])) { memcpy(global + (size_t)per * g, src, (size_t)per * 32); src += per; }
    if (with_dotp)
      for (int k : pl.dotp_of[r]) { memcpy(global + (size_t)per * (12 + k), src, (size_t)per * 32); src += per; }
  }
  return VPIN_OK;
}

// ---- ProductCircuitEvalProofBatched::prove (product_tree.rs:258-385) ------------------------------

struct Batched {
  std::vector<std::vector<Fq>> polys, claims_left, claims_right;  // per layer, top layer first
  std::vector<Fq> dotp[3];
};

static void write_batched(Writer& w, const Batched& b) {
  w.u64(b.polys.size());
  for (size_t l = 0; l < b.polys.size(); l++) {
    w.u64(b.polys[l].size() / 3);  // SumcheckInstanceProof.compressed_polys
    for (size_t j = 0; j < b.polys[l].size() / 3; j++) w_scalars(w, &b.polys[l][3 * j], 3);
    w_scalars(w, b.claims_left[l].data(), b.claims_left[l].size());
    w_scalars(w, b.claims_right[l].data(), b.claims_right[l].size());
  }
  for (int k = 0; k < 3; k++) w_scalars(w, b.dotp[k].data(), b.dotp[k].size());
}

static inline size_t pyramid_offset(int ell, int k) { return ((size_t)1 << ell) - ((size_t)2 << (ell - k)); }

struct DotpCtx {  // the six DotProductCircuit halves ride along on layer 0 of the ops forest
  const vpin_spark_decomm* d;
  const vpin::fq* comb_derefs;
  vpin::fq* scratch;
  Fq claims[6];  // their evaluations (claim_eval_dotp_left/right per matrix)
};

// pl != nullptr (one proof over several GPUs): `f` holds this rank's circuits only (possibly none: f.ncirc == 0), npc_all is
// the number of circuits of the whole forest (12 ops / 4 mem); the rounds run on the owned circuits and halves and every
// per-round result is exchanged (dist_exchange), after which the transcript work below is the same on every rank.
static int batched_prove(vpin_ctx* c, vpin::SparkForest& f, DotpCtx* dotp, Transcript& tr, Batched& out, std::vector<Fq>& rand,
                         const DistPlan* pl = nullptr, int npc_all = 0) {
  const int npc = pl ? npc_all : f.ncirc, ndotp = dotp ? 6 : 0;
  const bool is_mem = pl && npc_all == 4;
  const int nl = f.ncirc;                                             // circuits this rank runs
  const std::vector<int> no_halves;
  const std::vector<int>& my_halves = pl ? pl->dotp_of[pl->rank] : no_halves;
  const int ndl = !dotp ? 0 : (pl ? (int)my_halves.size() : 6);       // dot-product halves this rank runs
  const int* halves = pl ? my_halves.data() : nullptr;
  static const bool fine = getenv("VPIN_SPARK_TRACE") && atoi(getenv("VPIN_SPARK_TRACE")) >= 2;
  double t_setup = 0, t_first = 0, t_rounds = 0, t_epi = 0, t_host = 0;
  auto tl0 = Clock::now();
  const int num_layers = (int)log2z(f.n);
  int rc;
  // host copies of the small top levels
  const size_t cnt = std::min<size_t>(2 * vpin::kSparkHostTop, f.stride());
  if (nl && (rc = vpin::spark_fetch_tops(c, &f, cnt))) return rc;
  std::vector<Fq> tops((size_t)npc * cnt);
  if (!pl) memcpy(tops.data(), c->h_spark, tops.size() * 32);
  else if ((rc = dist_exchange(c, *pl, is_mem, false, reinterpret_cast<const Fq*>(c->h_spark), (int)cnt, tops.data(), is_mem ? "mem_tops" : "ops_tops"))) return rc;
  std::vector<Fq> res_all(3 * (size_t)vpin::kSparkMaxInst), fin_all(6 * (size_t)vpin::kSparkMaxInst), pack(6 * (size_t)vpin::kSparkMaxInst);

  struct TailGuard { vpin_ctx* c; ~TailGuard() { vpin::spark_tail_abort(c); } } tail_guard{c};  // a no-op unless an error return leaves a tail resident
  out.polys.assign(num_layers, {});
  out.claims_left.assign(num_layers, {});
  out.claims_right.assign(num_layers, {});
  std::vector<Fq> claims(npc + ndotp), coeffs;
  for (int t = 0; t < npc; t++) claims[t] = tops[(size_t)t * cnt + cnt - 2];  // the root: ProductCircuit::evaluate
  rand.clear();
  TableGuard tg(c);
  const Fq one = Fq::one();

  for (int layer_id = num_layers - 1, o = 0; layer_id >= 0; layer_id--, o++) {
    const size_t h = f.n >> (layer_id + 1);  // entries of left_vec[layer_id]
    const int k = (int)log2z(h);             // rounds; rand.size() == k
    const bool with_dotp = (layer_id == 0 && ndotp > 0);
    int nclaims = npc;
    if (with_dotp) {
      for (int i = 0; i < 6; i++) claims[npc + i] = dotp->claims[i];
      nclaims += 6;
    }
    if (fine) tl0 = Clock::now();
    coeffs = tr.challenge_vector("rand_coeffs_next_layer", nclaims);
    Fq e = Fq::zero();
    for (int i = 0; i < nclaims; i++) e = e + claims[i] * coeffs[i];
    std::vector<Fq> r(k);
    std::vector<Fq>& polys = out.polys[o];
    polys.resize(3 * (size_t)k);
    std::vector<Fq> cl(npc), cr(npc);
    const bool on_host = (2 * h <= vpin::kSparkHostTop) && layer_id != 0;

    if (on_host) {
      // prove_cubic_batched (sumcheck.rs:248-425) as written, on <= 16-entry tables
      std::vector<Fq> C(h), A((size_t)npc * h), Bv((size_t)npc * h);
      host_eq(rand.data(), (size_t)k, C.data());
      for (int t = 0; t < npc; t++) {
        const Fq* lvl = &tops[(size_t)t * cnt + cnt - 4 * h];  // level of 2h entries
        memcpy(&A[(size_t)t * h], lvl, h * 32);
        memcpy(&Bv[(size_t)t * h], lvl + h, h * 32);
      }
      size_t len = h;
      for (int j = 0; j < k; j++) {
        const size_t half = len / 2;
        Fq c0 = Fq::zero(), c2 = Fq::zero(), c3 = Fq::zero();
        for (int t = 0; t < npc; t++) {
          const Fq* a = &A[(size_t)t * h];
          const Fq* b = &Bv[(size_t)t * h];
          Fq e0 = Fq::zero(), e2 = Fq::zero(), e3 = Fq::zero();
          for (size_t i = 0; i < half; i++) {
            e0 = e0 + a[i] * b[i] * C[i];
            Fq a2 = a[half + i] + a[half + i] - a[i], b2 = b[half + i] + b[half + i] - b[i], c2p = C[half + i] + C[half + i] - C[i];
            e2 = e2 + a2 * b2 * c2p;
            Fq a3 = a2 + a[half + i] - a[i], b3 = b2 + b[half + i] - b[i], c3p = c2p + C[half + i] - C[i];
            e3 = e3 + a3 * b3 * c3p;
          }
          c0 = c0 + e0 * coeffs[t]; c2 = c2 + e2 * coeffs[t]; c3 = c3 + e3 * coeffs[t];
        }
        Fq evals[4] = {c0, e - c0, c2, c3}, cf[4];
        unipoly_from_evals(evals, 4, cf);
        append_unipoly(tr, cf, 4);
        Fq rj = tr.challenge_scalar("challenge_nextround");
        r[j] = rj;
        for (int t = 0; t < npc; t++) {
          Fq* a = &A[(size_t)t * h];
          Fq* b = &Bv[(size_t)t * h];
          for (size_t i = 0; i < half; i++) { a[i] = a[i] + rj * (a[half + i] - a[i]); b[i] = b[i] + rj * (b[half + i] - b[i]); }
        }
        for (size_t i = 0; i < half; i++) C[i] = C[i] + rj * (C[half + i] - C[i]);
        e = unipoly_eval(cf, 4, rj);
        polys[3 * j] = cf[0]; polys[3 * j + 1] = cf[2]; polys[3 * j + 2] = cf[3];
        len = half;
      }
      for (int t = 0; t < npc; t++) { cl[t] = A[(size_t)t * h]; cr[t] = Bv[(size_t)t * h]; }
    } else {
      // eq-factored rounds on the device: poly_C = eq(rand, .) folded with r_0..r_{j-1} equals
      // s_j * eq(rand_{j..}, .), s_j = prod_{i<j} eq1(rand_i, r_i); at the round's evaluation point x it is
      // s_j*((1-rand_j) + x*(2 rand_j - 1)) * E_{j+1}[i].  The kernel returns sum_i E_{j+1}[i]*(A_x B_x)[i].
      if (k < 1) return VPIN_ESHAPE;
      const int ndl_here = with_dotp ? ndl : 0;          // halves this rank runs on this layer
      const bool runs = nl > 0 || ndl_here > 0;          // a rank without circuits of this forest only follows the transcript
      vpin_table* pyr = nullptr;
      if (runs) {
        if ((rc = vpin_eq_suffix_tables(c, B(rand.data()), k, &pyr))) return rc;
        tg.add(pyr);
      }
      Fq s = one;
      // Leading-coefficient rounds: per circuit the kernel returns t(0) and the x^2 coefficient of
      // t(x) = sum_i E[i] (A_x B_x)[i]; t(1) follows from the circuit's claim, which the prover knows exactly
      // (claims[] are evaluations of the trees it built): with cn = (sum_t coeff_t claim_t) / s the combined
      // quadratic T satisfies cn = (1-rho) T(0) + rho T(1), and cn becomes T(r_j) after the round.  Exact field
      // identities: same c0, c2, c3 as summing at x = 0, 2, 3.  A zero rho_j (never, for a transcript
      // challenge) takes the three-sum kernel for that round.
      std::vector<Fq> rho_inv(rand.begin(), rand.begin() + k);
      bool lead_ok = true;
      for (auto& x : rho_inv) lead_ok = lead_ok && !x.is_zero();
      if (lead_ok) {  // Montgomery's trick: one inversion per layer
        std::vector<Fq> pre(k);
        Fq acc = one;
        for (int j = 0; j < k; j++) { pre[j] = acc; acc = acc * rho_inv[j]; }
        acc = acc.invert();
        for (int j = k - 1; j >= 0; j--) { Fq t = acc * rho_inv[j]; rho_inv[j] = acc * pre[j]; acc = t; }
      }
      Fq cn = Fq::zero();
      for (int t = 0; t < npc; t++) cn = cn + claims[t] * coeffs[t];
      // Rounds with at most spark_tail_pairs() pairs per circuit are proven by ONE resident launch (spark.hip, persistent
      // tail): the kernel publishes a round's sums to pinned memory and polls a pinned mailbox for the challenge this
      // loop derives from the transcript.  Larger rounds take one launch each.
      // (round 6: a layer without dot-product halves on a single GPU may start it at up to 8192 pairs, on several workgroups per
      // circuit, when the device has room for them -- spark_tail_launch says so)
      const size_t tail_pairs = lead_ok ? (pl ? vpin::spark_tail_pairs() : vpin::spark_tail_first_pairs(ndl_here > 0)) : 0;
      if (pl && tail_pairs == 0) {  // the split rounds end in the persistent tail; a zero challenge (never) or VPIN_SPARK_TAIL_PAIRS=0 rules it out
        vpin::set_last_error("one proof over several GPUs needs the persistent tail rounds", hipErrorUnknown);
        return VPIN_ESHAPE;
      }
      bool tail_on = false;
      int tail_j0 = 0;
      const int ninst = nl + ndl_here;  // instances this rank's launches carry
      if (fine) { auto t = Clock::now(); t_setup += secs(tl0, t); tl0 = t; }
      for (int j = 0; j < k; j++) {
        const size_t len = j == 0 ? h : (h >> (j - 1));  // live length before this round's launch
        const vpin::fq* E = runs ? pyr->d + pyramid_offset(k, j + 1) : nullptr;
        const uint8_t* rprev = j ? B(&r[j - 1]) : nullptr;
        if (!tail_on && (h >> (j + 1)) <= tail_pairs) {
          rc = runs ? vpin::spark_tail_launch(c, &f, layer_id, k, j, len, pyr->d, rprev, ndl_here ? dotp->d->N : 0,
                                              ndl_here ? dotp->d->vals : nullptr,
                                              ndl_here ? dotp->comb_derefs : nullptr, ndl_here ? dotp->scratch : nullptr,
                                              halves, ndl_here) : VPIN_OK;
          if (rc < 0) return rc;
          if (rc == 0) {   // (1: no room for the workgroups of an early start -- this round by a launch, asked again at the next)
            tail_on = true;
            tail_j0 = j;
          }
        }
        const Fq* res = nullptr;
        if (tail_on) {
          if (runs) {
            if ((rc = vpin::spark_tail_wait(c, j - tail_j0, ninst, nl))) return rc;
            res = reinterpret_cast<const Fq*>(vpin::spark_tail_sums(c));
          }
        } else if (runs) {
          if ((rc = vpin::spark_prod_round(c, &f, layer_id, len, E, rprev, ndl_here, lead_ok))) return rc;
          if (ndl_here && (rc = vpin::spark_dotp_round(c, dotp->d->N, dotp->d->vals, dotp->comb_derefs, dotp->scratch, len, j == 1, rprev, halves, ndl_here))) return rc;
          if ((rc = vpin::spark_wait_flag(c))) return rc;
          res = reinterpret_cast<const Fq*>(c->h_spark);
        }
        if (pl) {
          // this rank's sums (circuits at slots 0.., halves at slots 12..) -> everyone's, in the single-GPU slot order
          for (int t = 0; t < nl; t++) memcpy(&pack[3 * (size_t)t], res + 3 * (size_t)t, 96);
          // (a launch group puts its halves at slots 12.., the persistent tail numbers its instances consecutively)
          for (int i = 0; i < ndl_here; i++) memcpy(&pack[3 * (size_t)(nl + i)], res + 3 * (size_t)((tail_on ? nl : 12) + i), 96);
          if ((rc = dist_exchange(c, *pl, is_mem, with_dotp, pack.data(), 3, res_all.data(),
                                  is_mem ? (tail_on ? "mem_tail_round" : "mem_round") : (tail_on ? "ops_tail_round" : "ops_round"))))
            return rc;
          res = res_all.data();
        }
        if (fine) { auto t = Clock::now(); (j == 0 ? t_first : t_rounds) += secs(tl0, t); if (k >= 11) fprintf(stderr, " w%.1f", secs(tl0, t) * 1e6); tl0 = t; }
        const Fq rho = rand[j], omr = one - rho;
        Fq S0 = Fq::zero(), S2 = Fq::zero(), S3 = Fq::zero(), T1 = Fq::zero(), Sinf = Fq::zero();
        if (lead_ok) {
          for (int t = 0; t < npc; t++) { S0 = S0 + res[3 * t] * coeffs[t]; Sinf = Sinf + res[3 * t + 1] * coeffs[t]; }
          T1 = (cn - omr * S0) * rho_inv[j];
          const Fq two_inf = Sinf + Sinf, d10 = T1 - S0;
          S2 = T1 + d10 + two_inf;                               // T(2) = 2 T(1) - T(0) + 2 Tinf
          S3 = S2 + d10 + two_inf + two_inf;                     // T(3) = 3 T(1) - 2 T(0) + 6 Tinf
        } else {
          for (int t = 0; t < npc; t++) { S0 = S0 + res[3 * t] * coeffs[t]; S2 = S2 + res[3 * t + 1] * coeffs[t]; S3 = S3 + res[3 * t + 2] * coeffs[t]; }
        }
        const Fq two_rho = rho + rho;
        Fq c0 = s * omr * S0;
        Fq c2 = s * (two_rho + rho - one) * S2;                       // (1-rho) + 2(2rho-1) = 3rho - 1
        Fq c3 = s * (two_rho + two_rho + rho - one - one) * S3;       // (1-rho) + 3(2rho-1) = 5rho - 2
        if (with_dotp)
          for (int i = 0; i < 6; i++) {
            const Fq* q = res + 3 * (12 + i);
            c0 = c0 + q[0] * coeffs[npc + i]; c2 = c2 + q[1] * coeffs[npc + i]; c3 = c3 + q[2] * coeffs[npc + i];
          }
        Fq evals[4] = {c0, e - c0, c2, c3}, cf[4];
        unipoly_from_evals(evals, 4, cf);
        append_unipoly(tr, cf, 4);
        Fq rj = tr.challenge_scalar("challenge_nextround");
        if (tail_on && runs && j + 1 < k) vpin::spark_tail_reply(c, j - tail_j0, B(&rj));  // the kernel folds while the host finishes the round
        r[j] = rj;
        e = unipoly_eval(cf, 4, rj);
        if (lead_ok) cn = S0 + rj * ((T1 - S0 - Sinf) + rj * Sinf);   // T(r_j)
        s = s * (rho * rj + omr * (one - rj));
        polys[3 * j] = cf[0]; polys[3 * j + 1] = cf[2]; polys[3 * j + 2] = cf[3];
        if (fine) { auto t = Clock::now(); t_host += secs(tl0, t); if (k >= 11) fprintf(stderr, " m%.1f", secs(tl0, t) * 1e6); tl0 = t; }
      }
      if (fine && k >= 11) fprintf(stderr, "\n");
      // final fold of the two live entries per table with r_{k-1}
      const Fq rl = r[k - 1];
      if (tail_on) {
        const Fq* fin = runs ? reinterpret_cast<const Fq*>(vpin::spark_tail_final(c)) : nullptr;
        if (pl) {
          for (int t = 0; t < ninst; t++) memcpy(&pack[6 * (size_t)t], fin + 6 * (size_t)t, 192);  // the tail numbers its instances 0..ninst-1
          if ((rc = dist_exchange(c, *pl, is_mem, with_dotp, pack.data(), 6, fin_all.data(), is_mem ? "mem_finals" : "ops_finals"))) return rc;
          fin = fin_all.data();  // halves at slots 12.. = npc + i (only the ops forest, npc == 12, carries them)
        }
        for (int t = 0; t < npc; t++) {
          cl[t] = fin[6 * t] + rl * (fin[6 * t + 1] - fin[6 * t]);
          cr[t] = fin[6 * t + 2] + rl * (fin[6 * t + 3] - fin[6 * t + 2]);
        }
        if (with_dotp) {
          for (int t = 0; t < 3; t++) out.dotp[t].resize(6);
          for (int i = 0; i < 6; i++)
            for (int t = 0; t < 3; t++) {
              const Fq* q = fin + 6 * (npc + i) + 2 * t;
              out.dotp[t][i] = q[0] + rl * (q[1] - q[0]);
            }
        }
        if (runs) vpin::spark_tail_end(c);
      } else {
        if (pl) return VPIN_ESHAPE;
        if ((rc = vpin::spark_collect(c, &f, layer_id, with_dotp ? dotp->d : nullptr, with_dotp ? dotp->comb_derefs : nullptr,
                                      with_dotp ? dotp->scratch : nullptr, with_dotp, k >= 2)))
          return rc;
        const Fq* res = reinterpret_cast<const Fq*>(c->h_spark);
        for (int t = 0; t < npc; t++) {
          cl[t] = res[4 * t] + rl * (res[4 * t + 1] - res[4 * t]);
          cr[t] = res[4 * t + 2] + rl * (res[4 * t + 3] - res[4 * t + 2]);
        }
        if (with_dotp) {
          const Fq* q = reinterpret_cast<const Fq*>(c->h_spark) + 64;
          for (int t = 0; t < 3; t++) out.dotp[t].resize(6);
          for (int i = 0; i < 6; i++)
            for (int t = 0; t < 3; t++) out.dotp[t][i] = q[6 * i + 2 * t] + rl * (q[6 * i + 2 * t + 1] - q[6 * i + 2 * t]);
        }
      }
    }

    for (int t = 0; t < npc; t++) {
      tr.append_scalar("claim_prod_left", cl[t]);
      tr.append_scalar("claim_prod_right", cr[t]);
    }
    if (with_dotp)
      for (int i = 0; i < 6; i++) {
        tr.append_scalar("claim_dotp_left", out.dotp[0][i]);
        tr.append_scalar("claim_dotp_right", out.dotp[1][i]);
        tr.append_scalar("claim_dotp_weight", out.dotp[2][i]);
      }
    Fq r_layer = tr.challenge_scalar("challenge_r_layer");
    for (int t = 0; t < npc; t++) claims[t] = cl[t] + r_layer * (cr[t] - cl[t]);
    out.claims_left[o] = cl;
    out.claims_right[o] = cr;
    std::vector<Fq> ext;
    ext.reserve(k + 1);
    ext.push_back(r_layer);
    ext.insert(ext.end(), r.begin(), r.end());
    rand.swap(ext);
    if (fine) { auto t = Clock::now(); t_epi += secs(tl0, t); tl0 = t; }
  }
  if (fine)
    fprintf(stderr, "[spark]   forest of %d x 2^%d: setup %.3f  first result %.3f  later results %.3f  host per-round math %.3f  epilogue %.3f ms\n",
            npc, num_layers, t_setup * 1e3, t_first * 1e3, t_rounds * 1e3, t_host * 1e3, t_epi * 1e3);
  return VPIN_OK;
}

// ---- the same proof with every circuit split by RESIDUE CLASS over a power-of-two world (spark.hip) -----------------
// `f` is this rank's local forest: all npc circuits, n / W leaves each (local index k <-> global index rank + k W).  A layer
// of h = 2^k entries per half is h / W = 2^(k - lw) entries locally: the first k - lw rounds run on the device exactly as on
// one GPU (same launches, same persistent tail, the suffix pyramid of the first k - lw challenges; the rank's eq-factored
// sums are scaled by eq(rand_lo, rank)), each rank contributes its partial sums and all ranks derive the same challenge;
// then every local table is one entry, the W entries of each table are gathered and the last lw rounds run on the host.
// Layers of at most 32 entries are proven on the host from the gathered tree tops, as on one GPU.
struct StridedDotp {
  size_t Nloc;                    // N / W
  const vpin::fq* vals_loc;       // 3 x Nloc: the val slices at the local entries
  const vpin::fq* comb_loc;       // 6 x Nloc: the derefs slices at the local entries
  vpin::fq* scratch;              // 18 x Nloc / 4
  Fq claims[6];
};

// sum over ranks of `cnt` scalars per rank
static int dist_sum(vpin_ctx* c, const DistPlan& pl, const Fq* mine, size_t cnt, Fq* out, const char* tag) {
  std::vector<Fq> all(cnt * (size_t)pl.world);
  int rc = vpin::comm_allgather_ctx(c, mine, all.data(), cnt * 32, tag);
  if (rc) return rc;
  for (size_t i = 0; i < cnt; i++) {
    Fq a = Fq::zero();
    for (int r = 0; r < pl.world; r++) a = a + all[(size_t)r * cnt + i];
    out[i] = a;
  }
  return VPIN_OK;
}

static int batched_prove_strided(vpin_ctx* c, vpin::SparkForest& f, size_t n_global, StridedDotp* dotp, Transcript& tr, Batched& out,
                                 std::vector<Fq>& rand, const DistPlan& pl, bool is_mem) {
  const int npc = f.ncirc, ndotp = dotp ? 6 : 0, W = pl.world, lw = pl.lw;
  const int num_layers = (int)log2z(n_global);
  int rc;
  const Fq one = Fq::one();
  // ---- tree tops: the last cnt entries of every GLOBAL tree (levels of <= cnt / 2 entries) from the ranks' local tops ----
  const size_t cnt = std::min<size_t>(2 * vpin::kSparkHostTop, 2 * n_global), cntl = cnt / (size_t)W;
  if (cntl < 2 || cntl > f.stride()) return VPIN_ESHAPE;
  if ((rc = vpin::spark_fetch_tops(c, &f, cntl))) return rc;
  std::vector<Fq> tops((size_t)npc * cnt, Fq::zero());
  {
    std::vector<Fq> all((size_t)npc * cntl * W);
    if ((rc = vpin::comm_allgather_ctx(c, c->h_spark, all.data(), (size_t)npc * cntl * 32, is_mem ? "mem_tops" : "ops_tops"))) return rc;
    for (int t = 0; t < npc; t++) {
      Fq* g = &tops[(size_t)t * cnt];
      // levels of m >= W entries: global[i] = local_{i mod W}[i / W]; a level of m entries sits at offset cnt - 2m
      for (size_t m = cnt / 2; m >= (size_t)W; m >>= 1) {
        const size_t ml = m / W;
        for (size_t i = 0; i < m; i++) g[cnt - 2 * m + i] = all[((size_t)(i % W) * npc + t) * cntl + (cntl - 2 * ml) + i / W];
      }
      // the levels above: products of the level below (product_tree.rs:18-35)
      for (size_t m = (size_t)W / 2; m >= 1; m >>= 1)
        for (size_t i = 0; i < m; i++) g[cnt - 2 * m + i] = g[cnt - 4 * m + i] * g[cnt - 4 * m + m + i];
    }
  }
  out.polys.assign(num_layers, {});
  out.claims_left.assign(num_layers, {});
  out.claims_right.assign(num_layers, {});
  std::vector<Fq> claims(npc + ndotp), coeffs;
  for (int t = 0; t < npc; t++) claims[t] = tops[(size_t)t * cnt + cnt - 2];
  rand.clear();
  TableGuard tg(c);
  struct TailGuard { vpin_ctx* c; ~TailGuard() { vpin::spark_tail_abort(c); } } tail_guard{c};
  const int ninst = npc + ndotp;
  std::vector<Fq> res((size_t)3 * vpin::kSparkMaxInst), mine((size_t)6 * vpin::kSparkMaxInst), fin((size_t)6 * vpin::kSparkMaxInst);

  for (int layer_id = num_layers - 1, o = 0; layer_id >= 0; layer_id--, o++) {
    const size_t h = n_global >> (layer_id + 1);
    const int k = (int)log2z(h);
    const bool with_dotp = (layer_id == 0 && ndotp > 0);
    int nclaims = npc;
    if (with_dotp) {
      for (int i = 0; i < 6; i++) claims[npc + i] = dotp->claims[i];
      nclaims += 6;
    }
    coeffs = tr.challenge_vector("rand_coeffs_next_layer", nclaims);
    Fq e = Fq::zero();
    for (int i = 0; i < nclaims; i++) e = e + claims[i] * coeffs[i];
    std::vector<Fq> r(k);
    std::vector<Fq>& polys = out.polys[o];
    polys.resize(3 * (size_t)k);
    std::vector<Fq> cl(npc), cr(npc);
    const bool on_host = (2 * h <= vpin::kSparkHostTop) && layer_id != 0;
    // host tables of the layer: product circuits A, B; dot-product halves L, R, W (filled from the tops or from the gather)
    std::vector<std::vector<Fq>> HA(npc), HB(npc), HD[3];
    int j_host = 0;  // first round that runs on the host tables
    Fq s = one, cn = Fq::zero();
    std::vector<Fq> rho_inv(rand.begin(), rand.begin() + k);
    bool lead_ok = !on_host;
    for (auto& x : rho_inv) lead_ok = lead_ok && !x.is_zero();
    if (lead_ok && k > 0) {
      std::vector<Fq> pre(k);
      Fq acc = one;
      for (int j = 0; j < k; j++) { pre[j] = acc; acc = acc * rho_inv[j]; }
      acc = acc.invert();
      for (int j = k - 1; j >= 0; j--) { Fq t = acc * rho_inv[j]; rho_inv[j] = acc * pre[j]; acc = t; }
    }
    for (int t = 0; t < npc; t++) cn = cn + claims[t] * coeffs[t];

    vpin_table* pyr = nullptr;
    int kl = 0;           // device rounds
    bool tail_on = false;
    int tail_j0 = 0;
    Fq c_rank = one;
    if (on_host) {
      for (int t = 0; t < npc; t++) {
        const Fq* lvl = &tops[(size_t)t * cnt + cnt - 4 * h];
        HA[t].assign(lvl, lvl + h);
        HB[t].assign(lvl + h, lvl + 2 * h);
      }
    } else {
      kl = k - lw;
      if (kl < 1 || !lead_ok) {
        vpin::set_last_error("split by residue class: layer too small for the world, or a zero challenge", hipErrorUnknown);
        return VPIN_ESHAPE;
      }
      j_host = kl;
      if ((rc = vpin_eq_suffix_tables(c, B(rand.data()), kl, &pyr))) return rc;
      tg.add(pyr);
      std::vector<Fq> eq_lo((size_t)W);
      host_eq(rand.data() + kl, (size_t)lw, eq_lo.data());
      c_rank = eq_lo[pl.rank];
    }

    for (int j = 0; j < k; j++) {
      const Fq* rs = nullptr;
      if (j < j_host) {
        // ---- a device round on the local tables ----
        const size_t hl = h >> lw;
        const size_t len = j == 0 ? hl : (hl >> (j - 1));
        const vpin::fq* E = pyr->d + pyramid_offset(kl, j + 1);
        const uint8_t* rprev = j ? B(&r[j - 1]) : nullptr;
        const size_t tail_pairs = vpin::spark_tail_pairs();
        if (tail_pairs == 0) return VPIN_ESHAPE;
        if (!tail_on && (hl >> (j + 1)) <= tail_pairs) {
          if ((rc = vpin::spark_tail_launch(c, &f, layer_id, kl, j, len, pyr->d, rprev, with_dotp ? dotp->Nloc : 0,
                                            with_dotp ? dotp->vals_loc : nullptr, with_dotp ? dotp->comb_loc : nullptr,
                                            with_dotp ? dotp->scratch : nullptr, nullptr, with_dotp ? 6 : 0)))
            return rc;
          tail_on = true;
          tail_j0 = j;
        }
        const Fq* loc;
        if (tail_on) {
          if ((rc = vpin::spark_tail_wait(c, j - tail_j0, with_dotp ? ninst : npc, npc))) return rc;
          loc = reinterpret_cast<const Fq*>(vpin::spark_tail_sums(c));
          for (int t = 0; t < npc; t++) for (int x = 0; x < 3; x++) mine[3 * (size_t)t + x] = loc[3 * (size_t)t + x] * c_rank;
          if (with_dotp) memcpy(&mine[3 * (size_t)12], loc + 3 * (size_t)npc, 6 * 96);
        } else {
          if ((rc = vpin::spark_prod_round(c, &f, layer_id, len, E, rprev, with_dotp ? 6 : 0, true))) return rc;
          if (with_dotp && (rc = vpin::spark_dotp_round(c, dotp->Nloc, dotp->vals_loc, dotp->comb_loc, dotp->scratch, len, j == 1, rprev)))
            return rc;
          if ((rc = vpin::spark_wait_flag(c))) return rc;
          loc = reinterpret_cast<const Fq*>(c->h_spark);
          for (int t = 0; t < npc; t++) for (int x = 0; x < 3; x++) mine[3 * (size_t)t + x] = loc[3 * (size_t)t + x] * c_rank;
          if (with_dotp) memcpy(&mine[3 * (size_t)12], loc + 3 * (size_t)12, 6 * 96);
        }
        // slots: circuits 0.., halves 12.. (npc <= 12)
        if (!with_dotp) { if ((rc = dist_sum(c, pl, mine.data(), 3 * (size_t)npc, res.data(), is_mem ? "mem_round" : "ops_round"))) return rc; }
        else if ((rc = dist_sum(c, pl, mine.data(), 3 * (size_t)18, res.data(), "ops_round"))) return rc;
        rs = res.data();
      } else {
        // ---- a host round on the gathered (or top-level) tables: the kernels' conventions ----
        const size_t len = HA[0].size(), half = len / 2;
        std::vector<Fq> E(half);
        host_eq(rand.data() + j + 1, (size_t)(k - j - 1), E.data());
        for (int t = 0; t < npc; t++) {
          Fq a0 = Fq::zero(), a1 = Fq::zero(), a2 = Fq::zero();
          for (size_t i = 0; i < half; i++) {
            const Fq A0 = HA[t][i], dA = HA[t][i + half] - A0, B0 = HB[t][i], dB = HB[t][i + half] - B0;
            if (lead_ok) {
              a0 = a0 + E[i] * (A0 * B0);
              a1 = a1 + E[i] * (dA * dB);
            } else {
              const Fq A2 = A0 + dA + dA, B2 = B0 + dB + dB, A3 = A2 + dA, B3 = B2 + dB;
              a0 = a0 + E[i] * (A0 * B0); a1 = a1 + E[i] * (A2 * B2); a2 = a2 + E[i] * (A3 * B3);
            }
          }
          res[3 * (size_t)t] = a0; res[3 * (size_t)t + 1] = a1; res[3 * (size_t)t + 2] = a2;
        }
        if (with_dotp)
          for (int i = 0; i < 6; i++) {
            Fq q0 = Fq::zero(), q2 = Fq::zero(), q3 = Fq::zero();
            for (size_t x = 0; x < half; x++) {
              Fq v0[3], v2[3], v3[3];
              for (int tb = 0; tb < 3; tb++) {
                const Fq p = HD[tb][i][x], d = HD[tb][i][x + half] - p;
                v0[tb] = p; v2[tb] = p + d + d; v3[tb] = v2[tb] + d;
              }
              q0 = q0 + v0[0] * v0[1] * v0[2]; q2 = q2 + v2[0] * v2[1] * v2[2]; q3 = q3 + v3[0] * v3[1] * v3[2];
            }
            res[3 * (size_t)(12 + i)] = q0; res[3 * (size_t)(12 + i) + 1] = q2; res[3 * (size_t)(12 + i) + 2] = q3;
          }
        rs = res.data();
      }
      // ---- the round's polynomial and challenge (the same on every rank; identical to batched_prove) ----
      Fq c0, c2, c3, T1 = Fq::zero(), S0 = Fq::zero(), Sinf = Fq::zero();
      if (on_host) {
        // prove_cubic_batched as written (sumcheck.rs:248-425): the host tables carry no eq factoring
        const size_t len = HA[0].size(), half = len / 2;
        std::vector<Fq> C(len);
        // poly_C = eq(rand, .) folded with r_0..r_{j-1}: s * eq(rand_{j..}, .)
        host_eq(rand.data() + j, (size_t)(k - j), C.data());
        c0 = c2 = c3 = Fq::zero();
        for (int t = 0; t < npc; t++) {
          Fq e0 = Fq::zero(), e2 = Fq::zero(), e3 = Fq::zero();
          for (size_t i = 0; i < half; i++) {
            const Fq a = HA[t][i], a1 = HA[t][i + half], b = HB[t][i], b1 = HB[t][i + half], cc = s * C[i], cc1 = s * C[i + half];
            e0 = e0 + a * b * cc;
            const Fq a2 = a1 + a1 - a, b2 = b1 + b1 - b, c2p = cc1 + cc1 - cc;
            e2 = e2 + a2 * b2 * c2p;
            const Fq a3 = a2 + a1 - a, b3 = b2 + b1 - b, c3p = c2p + cc1 - cc;
            e3 = e3 + a3 * b3 * c3p;
          }
          c0 = c0 + e0 * coeffs[t]; c2 = c2 + e2 * coeffs[t]; c3 = c3 + e3 * coeffs[t];
        }
      } else {
        const Fq rho = rand[j], omr = one - rho;
        Fq S2 = Fq::zero(), S3 = Fq::zero();
        if (lead_ok) {
          for (int t = 0; t < npc; t++) { S0 = S0 + rs[3 * t] * coeffs[t]; Sinf = Sinf + rs[3 * t + 1] * coeffs[t]; }
          T1 = (cn - omr * S0) * rho_inv[j];
          const Fq two_inf = Sinf + Sinf, d10 = T1 - S0;
          S2 = T1 + d10 + two_inf;
          S3 = S2 + d10 + two_inf + two_inf;
        } else {
          for (int t = 0; t < npc; t++) { S0 = S0 + rs[3 * t] * coeffs[t]; S2 = S2 + rs[3 * t + 1] * coeffs[t]; S3 = S3 + rs[3 * t + 2] * coeffs[t]; }
        }
        const Fq two_rho = rho + rho;
        c0 = s * omr * S0;
        c2 = s * (two_rho + rho - one) * S2;
        c3 = s * (two_rho + two_rho + rho - one - one) * S3;
        if (with_dotp)
          for (int i = 0; i < 6; i++) {
            const Fq* q = rs + 3 * (12 + i);
            c0 = c0 + q[0] * coeffs[npc + i]; c2 = c2 + q[1] * coeffs[npc + i]; c3 = c3 + q[2] * coeffs[npc + i];
          }
      }
      Fq evals[4] = {c0, e - c0, c2, c3}, cf[4];
      unipoly_from_evals(evals, 4, cf);
      append_unipoly(tr, cf, 4);
      const Fq rj = tr.challenge_scalar("challenge_nextround");
      if (j < j_host && tail_on && j + 1 < j_host) vpin::spark_tail_reply(c, j - tail_j0, B(&rj));
      r[j] = rj;
      e = unipoly_eval(cf, 4, rj);
      if (!on_host) {
        const Fq rho = rand[j], omr = one - rho;
        if (lead_ok) cn = S0 + rj * ((T1 - S0 - Sinf) + rj * Sinf);
        s = s * (rho * rj + omr * (one - rj));
      } else {
        s = s * (rand[j] * rj + (one - rand[j]) * (one - rj));
      }
      polys[3 * j] = cf[0]; polys[3 * j + 1] = cf[2]; polys[3 * j + 2] = cf[3];
      if (j >= j_host) {
        // bound_poly_var_top on the host tables
        auto fold = [&](std::vector<Fq>& T) {
          const size_t half = T.size() / 2;
          for (size_t i = 0; i < half; i++) T[i] = T[i] + rj * (T[i + half] - T[i]);
          T.resize(half);
        };
        for (int t = 0; t < npc; t++) { fold(HA[t]); fold(HB[t]); }
        if (with_dotp) for (int tb = 0; tb < 3; tb++) for (int i = 0; i < 6; i++) fold(HD[tb][i]);
      } else if (j + 1 == j_host) {
        // the device rounds are over: the two live entries of every local table, bound with r_j, are this rank's entry of
        // the W-entry tables the remaining rounds run on
        const Fq* fl = reinterpret_cast<const Fq*>(vpin::spark_tail_final(c));
        for (int t = 0; t < npc; t++) {
          mine[2 * (size_t)t] = fl[6 * t] + rj * (fl[6 * t + 1] - fl[6 * t]);
          mine[2 * (size_t)t + 1] = fl[6 * t + 2] + rj * (fl[6 * t + 3] - fl[6 * t + 2]);
        }
        size_t per_rank = 2 * (size_t)npc;
        if (with_dotp) {
          for (int i = 0; i < 6; i++)
            for (int tb = 0; tb < 3; tb++) {
              const Fq* q = fl + 6 * (npc + i) + 2 * tb;
              mine[per_rank + 3 * (size_t)i + tb] = q[0] + rj * (q[1] - q[0]);
            }
          per_rank += 18;
        }
        vpin::spark_tail_end(c);
        tail_on = false;
        std::vector<Fq> all(per_rank * (size_t)W);
        if ((rc = vpin::comm_allgather_ctx(c, mine.data(), all.data(), per_rank * 32, is_mem ? "mem_gather_tables" : "ops_gather_tables")))
          return rc;
        for (int t = 0; t < npc; t++) {
          HA[t].resize(W); HB[t].resize(W);
          for (int rk = 0; rk < W; rk++) { HA[t][rk] = all[(size_t)rk * per_rank + 2 * t]; HB[t][rk] = all[(size_t)rk * per_rank + 2 * t + 1]; }
        }
        if (with_dotp)
          for (int tb = 0; tb < 3; tb++) {
            HD[tb].assign(6, std::vector<Fq>(W));
            for (int i = 0; i < 6; i++)
              for (int rk = 0; rk < W; rk++) HD[tb][i][rk] = all[(size_t)rk * per_rank + 2 * (size_t)npc + 3 * (size_t)i + tb];
          }
      }
    }
    for (int t = 0; t < npc; t++) { cl[t] = HA[t][0]; cr[t] = HB[t][0]; }
    if (with_dotp) {
      for (int tb = 0; tb < 3; tb++) out.dotp[tb].resize(6);
      for (int i = 0; i < 6; i++)
        for (int tb = 0; tb < 3; tb++) out.dotp[tb][i] = HD[tb][i][0];
    }
    for (int t = 0; t < npc; t++) {
      tr.append_scalar("claim_prod_left", cl[t]);
      tr.append_scalar("claim_prod_right", cr[t]);
    }
    if (with_dotp)
      for (int i = 0; i < 6; i++) {
        tr.append_scalar("claim_dotp_left", out.dotp[0][i]);
        tr.append_scalar("claim_dotp_right", out.dotp[1][i]);
        tr.append_scalar("claim_dotp_weight", out.dotp[2][i]);
      }
    const Fq r_layer = tr.challenge_scalar("challenge_r_layer");
    for (int t = 0; t < npc; t++) claims[t] = cl[t] + r_layer * (cr[t] - cl[t]);
    out.claims_left[o] = cl;
    out.claims_right[o] = cr;
    std::vector<Fq> ext;
    ext.reserve(k + 1);
    ext.push_back(r_layer);
    ext.insert(ext.end(), r.begin(), r.end());
    rand.swap(ext);
  }
  return VPIN_OK;
}

static thread_local double g_spark_timings[8];

// SparseMatPolyEvalProof::prove (sparse_mlpoly.rs:1466-1533) -> bincode(R1CSEvalProof) appended to w
static int spark_prove(vpin_ctx* c, const vpin_spark_decomm* d, const std::vector<Fq>& rx, const std::vector<Fq>& ry,
                       const Fq evals[3], Transcript& tr, Transcript& tape, Writer& w) {
  const size_t N = d->N, M = d->M, lgN = log2z(N), lgM = log2z(M);
  if (N < 4 || M < 4 || rx.size() != d->nx || ry.size() != d->ny) return VPIN_ESHAPE;
  int rc;
  const Shape sh{d->nx, d->ny, N, M, lgN + 4, std::max(d->nx, d->ny) + 1, lgN + 3};
  const PcGens *g_ops = nullptr, *g_mem = nullptr, *g_derefs = nullptr;
  // the longest stream first: a later, longer request would rebuild the table and drop earlier views
  if ((rc = get_view(c, std::max(sh.v_ops, sh.v_mem), &g_ops)) || (rc = get_view(c, sh.v_ops, &g_ops)) ||
      (rc = get_view(c, sh.v_mem, &g_mem)) || (rc = get_view(c, sh.v_derefs, &g_derefs)))
    return rc;
  TableGuard tg(c);
  auto t0 = Clock::now();

  // one proof over several GPUs (vpin_ctx_set_comm): every rank is here with the same transcript state
  DistPlan plan;
  const DistPlan* dz = nullptr;
  if (c->comm && c->comm->world > 1) {
    if (c->comm->world > 12) return VPIN_EINVAL;  // every rank owns at least one of the 12 ops circuits
    plan_init(plan, c->comm);
    dz = &plan;
    // a power-of-two world splits every circuit by residue class (batched_prove_strided: perfectly balanced, and the derefs
    // gather, the leaves and the slices shrink with the world too); otherwise the circuits are dealt out whole
    static const bool by_circuit = getenv("VPIN_DIST_BY_CIRCUIT") != nullptr;
    const size_t Wz = (size_t)plan.world;
    if (!by_circuit && (Wz & (Wz - 1)) == 0 && Wz <= 16 && N / Wz >= 4096 && M / Wz >= 4096 && g_derefs->L >= Wz)
      plan.lw = (int)log2z(Wz);
  }
  const bool st = dz && dz->lw > 0;                       // split by residue class
  const size_t Wz = dz ? (size_t)dz->world : 1, rk = dz ? (size_t)dz->rank : 0;
  tr.append_protocol_name("Sparse polynomial evaluation proof");
  // equalize (sparse_mlpoly.rs:1448-1465) + the two memories eq(rx_ext, .), eq(ry_ext, .)
  const size_t nm = std::max(d->nx, d->ny);
  std::vector<Fq> rx_ext(nm, Fq::zero()), ry_ext(nm, Fq::zero());
  std::copy(rx.begin(), rx.end(), rx_ext.begin() + (nm - d->nx));
  std::copy(ry.begin(), ry.end(), ry_ext.begin() + (nm - d->ny));
  vpin_table *mem_rx = nullptr, *mem_ry = nullptr, *comb = nullptr;
  if ((rc = vpin_eq_table(c, B(rx_ext.data()), (int)nm, &mem_rx))) return rc;
  tg.add(mem_rx);
  if ((rc = vpin_eq_table(c, B(ry_ext.data()), (int)nm, &mem_ry))) return rc;
  tg.add(mem_ry);
  // Derefs (sparse_mlpoly.rs:525-531,56-71) and their commitment
  vpin::DevBuf b_cloc(c), b_crows(c), b_vloc(c);          // split by residue class: the local views (spark.hip)
  vpin::fq *comb_loc = nullptr, *comb_rows = nullptr, *vals_loc = nullptr;
  vpin_table comb_shape;                                  // shape-only handle of the (never materialised) derefs polynomial
  comb_shape.d = nullptr; comb_shape.len = comb_shape.cap = 8 * N; comb_shape.owned = false;
  const size_t nrows_loc = st ? vpin::comm_strided_count(g_derefs->L, dz->rank, dz->world) : 0;
  if (st) {
    const size_t Nl = N / Wz;
    if (b_cloc.alloc(6 * Nl * 32) || b_crows.alloc(std::max<size_t>(1, nrows_loc) * g_derefs->R * 32) || b_vloc.alloc(3 * Nl * 32)) return VPIN_ENOMEM;
    comb_loc = (vpin::fq*)b_cloc.p; comb_rows = (vpin::fq*)b_crows.p; vals_loc = (vpin::fq*)b_vloc.p;
    if ((rc = vpin::spark_gather_derefs_strided(c, d, mem_rx->d, mem_ry->d, rk, Wz, comb_loc, comb_rows, g_derefs->R, nrows_loc))) return rc;
    for (int m = 0; m < 3; m++)
      if ((rc = vpin::spark_take_strided(c, d->vals + (size_t)m * N, Nl, rk, Wz, vals_loc + (size_t)m * Nl))) return rc;
    comb = &comb_shape;
  } else {
    if ((rc = vpin::table_alloc_uninit(c, 8 * N, &comb))) return rc;
    tg.add(comb);
    if ((rc = vpin::spark_gather_derefs(c, d, mem_rx->d, mem_ry->d, comb->d))) return rc;
  }
  if ((rc = vpin::comm_mark(c, "derefs_gather"))) return rc;
  std::vector<CG> comm_derefs;
  const bool hot = d->hot_col[0] != 0xffffffffu || d->hot_col[1] != 0xffffffffu || d->hot_col[2] != 0xffffffffu;
  if (dz) {
    // the L row commitments are independent MSMs over shared generators: rank r commits rows r, r + world, .. (the rows
    // differ in cost: two of the polynomial's eight slices are zero padding, the col slices carry the hot columns), 32
    // bytes per row are all-gathered (no point crosses a link, nothing is reduced)
    const size_t L = g_derefs->L, pmax = vpin::comm_block_max(L, dz->world);
    const size_t nrows = vpin::comm_strided_count(L, dz->rank, dz->world);  // rows rank, rank + world, ..
    std::vector<uint8_t> mine(pmax * 32, 0), all(pmax * 32 * (size_t)dz->world);
    if (nrows) {
      if (hot) {
        const uint32_t* col_idx[3] = {d->idx + 6 * d->N, d->idx + 7 * d->N, d->idx + 8 * d->N};
        rc = vpin::hyrax_commit_derefs_hot(c, g_derefs->dev, comb, L, d->N, col_idx, d->hot_col, mem_ry->d, mine.data(), (size_t)dz->rank,
                                           nrows, (size_t)dz->world, st ? comb_rows : nullptr);
      } else {
        rc = vpin::hyrax_commit_rows_strided(c, g_derefs->dev, comb, L, (size_t)dz->rank, nrows, (size_t)dz->world, mine.data(),
                                             st ? comb_rows : nullptr);
      }
      if (rc) return rc;
    }
    if ((rc = vpin::comm_allgather_ctx(c, mine.data(), all.data(), mine.size(), "derefs_commit"))) return rc;
    comm_derefs.resize(L);
    for (int r = 0; r < dz->world; r++) {
      const size_t n = vpin::comm_strided_count(L, r, dz->world);
      for (size_t k = 0; k < n; k++) memcpy(comm_derefs[(size_t)r + k * (size_t)dz->world].b, all.data() + ((size_t)r * pmax + k) * 32, 32);
    }
  } else if (d->hot_col[0] != 0xffffffffu || d->hot_col[1] != 0xffffffffu || d->hot_col[2] != 0xffffffffu) {
    // the entries of each matrix's hot column hold one scalar, E_ry[hot]: one addition each instead of a table walk
    const uint32_t* col_idx[3] = {d->idx + 6 * d->N, d->idx + 7 * d->N, d->idx + 8 * d->N};
    comm_derefs.resize(g_derefs->L);
    if ((rc = vpin::hyrax_commit_derefs_hot(c, g_derefs->dev, comb, g_derefs->L, d->N, col_idx, d->hot_col, mem_ry->d,
                                            comm_derefs[0].b)))
      return rc;
  } else if ((rc = commit_noblind(c, g_derefs, comb, comm_derefs))) {
    return rc;
  }
  tr.append_message("derefs_commitment", "begin_derefs_commitment");  // DerefsCommitment::append_to_transcript (:216-222)
  append_polycomm(tr, "comm_poly_row_col_ops_val", comm_derefs);
  tr.append_message("derefs_commitment", "end_derefs_commitment");
  g_spark_timings[1] = secs(t0, Clock::now());
  if (c->progress_flag) *c->progress_flag = 2;  // the proof's largest MSM is done (a scheduler may let other streams in now)

  // ---- PolyEvalNetwork::new (sparse_mlpoly.rs:681-696) ----
  t0 = Clock::now();
  std::vector<Fq> r_mem_check = tr.challenge_vector("challenge_r_hash", 2);
  const Fq r_hash = r_mem_check[0], gamma = r_mem_check[1], r_hash_sqr = r_hash * r_hash, r2_boost = r_hash_sqr * Fq::r2();
  vpin::SparkForest f_ops, f_mem;
  vpin::DevBuf b_ops(c), b_mem(c), b_scr(c);
  const int nl_ops = (dz && !st) ? (int)dz->ops_of[dz->rank].size() : 12, nl_mem = (dz && !st) ? (int)dz->mem_of[dz->rank].size() : 4;
  const size_t nl_dotp = (dz && !st) ? dz->dotp_of[dz->rank].size() : 6;  // the first-fold scratch is numbered by the local half
  const size_t Nf = st ? N / Wz : N, Mf = st ? M / Wz : M;               // leaves per tree on this rank
  // Single GPU, a context in low-memory mode (vpin_ctx_set_low_memory, or VPIN_MEM_FOREST_LATE=1; round 5): the mem forest is
  // built AFTER the ops forest has been proven, into the memory it frees (-16 GiB of working set for the 2^25 instance, the
  // LeNet step's HBM 197 -> 181 GiB); the four mem roots the transcript wants first come from a product reduction over the
  // leaves (spark_mem_roots).  It costs 1 % of the 2^25 proof and 1.6 % of the four-lane step, so it is not the default.
  const bool defer_mem = !dz && (c->low_memory || getenv("VPIN_MEM_FOREST_LATE") != nullptr);
  if (b_ops.alloc((size_t)nl_ops * 2 * Nf * 32) || (nl_mem && !defer_mem && b_mem.alloc((size_t)nl_mem * 2 * Mf * 32)) ||
      b_scr.alloc(std::max<size_t>(256, 3 * nl_dotp * (Nf / 4) * 32)))
    return VPIN_ENOMEM;
  if ((rc = vpin::comm_mark(c, "network_alloc"))) return rc;
  f_ops.base = (vpin::fq*)b_ops.p; f_ops.n = Nf; f_ops.ncirc = nl_ops;
  f_mem.base = (vpin::fq*)b_mem.p; f_mem.n = Mf; f_mem.ncirc = nl_mem;
  if (st) {
    if ((rc = vpin::spark_build_forests_strided(c, d, comb_loc, mem_rx->d, mem_ry->d, B(&r_hash), B(&r_hash_sqr), B(&r2_boost), B(&gamma),
                                                &f_ops, &f_mem, rk, Wz)))
      return rc;
  } else if (defer_mem) {
    if ((rc = vpin::spark_build_forest_ops(c, d, comb->d, B(&r_hash), B(&r_hash_sqr), B(&r2_boost), B(&gamma), &f_ops))) return rc;
  } else if (!dz) {
    if ((rc = vpin::spark_build_forests(c, d, comb->d, mem_rx->d, mem_ry->d, B(&r_hash), B(&r_hash_sqr), B(&r2_boost), B(&gamma),
                                        &f_ops, &f_mem)))
      return rc;
  } else {
    // leaves and trees of this rank's circuits only
    if ((rc = vpin::spark_build_forest_sub(c, d, comb->d, mem_rx->d, mem_ry->d, B(&r_hash), B(&r_hash_sqr), B(&r2_boost), B(&gamma),
                                           &f_ops, dz->ops_of[dz->rank].data(), false)))
      return rc;
    if (nl_mem && (rc = vpin::spark_build_forest_sub(c, d, comb->d, mem_rx->d, mem_ry->d, B(&r_hash), B(&r_hash_sqr), B(&r2_boost),
                                                     B(&gamma), &f_mem, dz->mem_of[dz->rank].data(), true)))
      return rc;
  }
  if ((rc = vpin::spark_wait(c))) return rc;
  if ((rc = vpin::comm_mark(c, "network_build"))) return rc;
  g_spark_timings[2] = secs(t0, Clock::now());

  // ---- PolyEvalNetworkProof::prove / ProductLayerProof::prove (:1336-1370, :1049-1227) ----
  t0 = Clock::now();
  tr.append_protocol_name("Sparse polynomial evaluation proof");
  tr.append_protocol_name("Sparse polynomial product layer proof");
  // roots of the 16 circuits
  Fq pl[2][8];  // per side: init, read[3], write[3], audit
  {
    std::vector<Fq> roots(2 * (size_t)vpin::kSparkMaxInst);
    if ((rc = vpin::spark_fetch_tops(c, &f_ops, 2))) return rc;
    const Fq* t = reinterpret_cast<const Fq*>(c->h_spark);
    std::vector<Fq> st_roots(2 * 16);
    if (st) {
      // a circuit's root is the product of its leaves = the product over the ranks of their local roots
      std::vector<Fq> mine16(16), all16(16 * Wz);
      for (int i = 0; i < 12; i++) mine16[i] = t[2 * i];
      if ((rc = vpin::spark_fetch_tops(c, &f_mem, 2))) return rc;
      for (int i = 0; i < 4; i++) mine16[12 + i] = t[2 * i];
      if ((rc = vpin::comm_allgather_ctx(c, mine16.data(), all16.data(), 16 * 32, "roots"))) return rc;
      for (int i = 0; i < 16; i++) {
        Fq p = Fq::one();
        for (size_t r = 0; r < Wz; r++) p = p * all16[16 * r + i];
        st_roots[2 * i] = p;
      }
      for (int i = 0; i < 12; i++) roots[2 * i] = st_roots[2 * i];
      t = roots.data();
    } else if (dz) {
      if ((rc = dist_exchange(c, *dz, false, false, t, 2, roots.data(), "roots"))) return rc;
      t = roots.data();
    }
    for (int s = 0; s < 2; s++)
      for (int m = 0; m < 3; m++) { pl[s][1 + m] = t[2 * (s * 6 + m)]; pl[s][4 + m] = t[2 * (s * 6 + 3 + m)]; }
    if (st) {
      for (int i = 0; i < 4; i++) roots[2 * i] = st_roots[2 * (12 + i)];
      t = roots.data();
    } else {
      if (defer_mem) {
        if ((rc = vpin::spark_mem_roots(c, d, mem_rx->d, mem_ry->d, B(&r_hash), B(&r_hash_sqr), B(&r2_boost), B(&gamma)))) return rc;
      } else if (nl_mem && (rc = vpin::spark_fetch_tops(c, &f_mem, 2))) {
        return rc;
      }
      t = reinterpret_cast<const Fq*>(c->h_spark);
      if (dz) {
        if ((rc = dist_exchange(c, *dz, true, false, t, 2, roots.data(), "roots"))) return rc;
        t = roots.data();
      }
    }
    for (int s = 0; s < 2; s++) { pl[s][0] = t[2 * (2 * s)]; pl[s][7] = t[2 * (2 * s + 1)]; }
  }
  static const char* lab[2][4] = {{"claim_row_eval_init", "claim_row_eval_read", "claim_row_eval_write", "claim_row_eval_audit"},
                                  {"claim_col_eval_init", "claim_col_eval_read", "claim_col_eval_write", "claim_col_eval_audit"}};
  for (int s = 0; s < 2; s++) {
    Fq ws = Fq::one(), rs = Fq::one();
    for (int m = 0; m < 3; m++) { rs = rs * pl[s][1 + m]; ws = ws * pl[s][4 + m]; }
    if (!(pl[s][0] * ws == rs * pl[s][7])) return VPIN_ESHAPE;  // assert_eq!(row_eval_init * ws, rs * row_eval_audit)
    tr.append_scalar(lab[s][0], pl[s][0]);
    tr.append_scalars(lab[s][1], &pl[s][1], 3);
    tr.append_scalars(lab[s][2], &pl[s][4], 3);
    tr.append_scalar(lab[s][3], pl[s][7]);
  }
  DotpCtx dotp{d, comb->d, (vpin::fq*)b_scr.p, {}};
  StridedDotp sdotp{Nf, vals_loc, comb_loc, (vpin::fq*)b_scr.p, {}};
  Fq dotp_left[3], dotp_right[3];
  {
    // DotProductCircuit::evaluate (product_tree.rs:87-91) of the six halves
    if (st) {
      // every rank sums its residue class of all six halves; the partial sums are added up
      if ((rc = vpin::spark_triple_sums_raw(c, comb_loc, vals_loc, Nf))) return rc;
      Fq mine6[6];
      for (int i = 0; i < 6; i++) mine6[i] = reinterpret_cast<const Fq*>(c->h_spark)[3 * i];
      if ((rc = dist_sum(c, *dz, mine6, 6, dotp.claims, "triple_sums"))) return rc;
      for (int i = 0; i < 6; i++) sdotp.claims[i] = dotp.claims[i];
    } else if (!dz) {
      if ((rc = vpin::spark_triple_sums(c, d, comb->d))) return rc;
      for (int i = 0; i < 6; i++) dotp.claims[i] = reinterpret_cast<const Fq*>(c->h_spark)[3 * i];
    } else {
      // every rank sums the halves it will prove; one scalar per half is exchanged
      const auto& hv = dz->dotp_of[dz->rank];
      if (!hv.empty() && (rc = vpin::spark_triple_sums(c, d, comb->d, hv.data(), (int)hv.size()))) return rc;
      std::vector<Fq> mine6(6, Fq::zero()), all6(6 * (size_t)dz->world);
      for (size_t i = 0; i < hv.size(); i++) mine6[hv[i]] = reinterpret_cast<const Fq*>(c->h_spark)[3 * i];
      if ((rc = vpin::comm_allgather_ctx(c, mine6.data(), all6.data(), 6 * 32, "triple_sums"))) return rc;
      for (int k = 0; k < 6; k++) dotp.claims[k] = all6[6 * (size_t)dz->owner_dotp[k] + k];
    }
    for (int m = 0; m < 3; m++) {
      dotp_left[m] = dotp.claims[2 * m];
      dotp_right[m] = dotp.claims[2 * m + 1];
      tr.append_scalar("claim_eval_dotp_left", dotp_left[m]);
      tr.append_scalar("claim_eval_dotp_right", dotp_right[m]);
      if (!(dotp_left[m] + dotp_right[m] == evals[m])) return VPIN_ESHAPE;  // assert_eq!(left + right, eval[i])
    }
  }
  Batched pf_ops, pf_mem;
  std::vector<Fq> rand_ops, rand_mem;
  {
    TraceSpan ts("product: ops forest");
    if (st) rc = batched_prove_strided(c, f_ops, N, &sdotp, tr, pf_ops, rand_ops, *dz, false);
    else rc = batched_prove(c, f_ops, &dotp, tr, pf_ops, rand_ops, dz, 12);
    if (rc) return rc;
  }
  if (defer_mem) {
    TraceSpan ts("network: mem forest (deferred)");
    b_ops.release();   // every level of the ops forest has been folded away: its memory takes the mem forest
    b_scr.release();
    if (b_mem.alloc((size_t)4 * 2 * M * 32)) return VPIN_ENOMEM;
    f_mem.base = (vpin::fq*)b_mem.p;
    if ((rc = vpin::spark_build_forest_mem(c, d, mem_rx->d, mem_ry->d, B(&r_hash), B(&r_hash_sqr), B(&r2_boost), B(&gamma), &f_mem))) return rc;
    if ((rc = vpin::spark_wait(c))) return rc;
  }
  {
    TraceSpan ts("product: mem forest");
    if (st) rc = batched_prove_strided(c, f_mem, M, nullptr, tr, pf_mem, rand_mem, *dz, true);
    else rc = batched_prove(c, f_mem, nullptr, tr, pf_mem, rand_mem, dz, 4);
    if (rc) return rc;
  }
  g_spark_timings[3] = secs(t0, Clock::now());

  // ---- HashLayerProof::prove (:740-849) ----
  t0 = Clock::now();
  tr.append_protocol_name("Sparse polynomial hash layer proof");
  vpin_table *eq_ops = nullptr, *eq_mem = nullptr;
  // single GPU: slice evaluations and the evaluation proofs' LZ vectors from one pass per combined polynomial (SlicePass)
  const bool one_pass = !dz && SlicePass::fits(*g_derefs, 3, rand_ops) && SlicePass::fits(*g_ops, 4, rand_ops) &&
                        SlicePass::fits(*g_mem, 1, rand_mem);
  SlicePass sp_derefs(c), sp_ops(c), sp_mem(c);
  // the combined polynomials as whole field tables: only the two-pass and the multi-GPU paths read them (this proof's own
  // copies: the decommitment is shared with other contexts)
  vpin_table *comb_ops = nullptr, *comb_mem = nullptr;
  if (!one_pass && dz) {
    // a proof split over several GPUs: built once per decommitment on first use and kept (the ranks of a rehearsal share one
    // decommitment and one copy; on a real node every GPU holds its own)
    auto* dm = const_cast<vpin_spark_decomm*>(d);
    dm->comb_unpooled = true;
    if ((rc = vpin::spark_comb_tables(c, dm))) return rc;
    comb_ops = dm->comb_ops; comb_mem = dm->comb_mem;
  } else if (!one_pass) {
    if ((rc = vpin::spark_comb_make(c, d, &comb_ops, &comb_mem))) return rc;
    tg.add(comb_ops); tg.add(comb_mem);
  }
  if (!one_pass) {
    TraceSpan ts("hash: eq tables");
    // split by residue class: eq(rand, rank + k W) = eq(rand_hi, k) * eq(rand_lo, rank), so the local table is the table of
    // the first lg - lw challenges and the rank's sums are scaled by one constant
    const int lwz = st ? dz->lw : 0;
    if ((rc = vpin_eq_table(c, B(rand_ops.data()), (int)lgN - lwz, &eq_ops))) return rc;
    tg.add(eq_ops);
    if ((rc = vpin_eq_table(c, B(rand_mem.data()), (int)lgM - lwz, &eq_mem))) return rc;
    tg.add(eq_mem);
  }
  if ((rc = vpin::comm_mark(c, "hash_eq_tables"))) return rc;
  Fq ev_derefs[6], ev_ops[15], ev_mem[2];
  const Fq* hs = reinterpret_cast<const Fq*>(c->h_spark);
  if (st) {
    // every rank evaluates its residue class of all 23 slices; the partial sums (scaled by eq(rand_lo, rank)) are added up
    std::vector<Fq> lo_ops(Wz), lo_mem(Wz);
    host_eq(rand_ops.data() + (lgN - dz->lw), (size_t)dz->lw, lo_ops.data());
    host_eq(rand_mem.data() + (lgM - dz->lw), (size_t)dz->lw, lo_mem.data());
    Fq mine23[23], all23[23];
    if ((rc = vpin::spark_slice_evals(c, comb_loc, Nf, 6, eq_ops->d))) return rc;
    for (int i = 0; i < 6; i++) mine23[i] = hs[3 * i] * lo_ops[rk];
    if ((rc = vpin::spark_slice_evals(c, comb_ops->d, N, 15, eq_ops->d, rk, Wz))) return rc;
    for (int i = 0; i < 15; i++) mine23[6 + i] = hs[3 * i] * lo_ops[rk];
    if ((rc = vpin::spark_slice_evals(c, comb_mem->d, M, 2, eq_mem->d, rk, Wz))) return rc;
    for (int i = 0; i < 2; i++) mine23[21 + i] = hs[3 * i] * lo_mem[rk];
    if ((rc = dist_sum(c, *dz, mine23, 23, all23, "hash_slice_evals"))) return rc;
    for (int i = 0; i < 6; i++) ev_derefs[i] = all23[i];
    for (int i = 0; i < 15; i++) ev_ops[i] = all23[6 + i];
    for (int i = 0; i < 2; i++) ev_mem[i] = all23[21 + i];
  } else if (dz) {
    // The 23 DensePolynomial::evaluate of the hash layer (6 derefs + 15 ops slices against eq(rand_ops), 2 mem slices
    // against eq(rand_mem)) depend on nothing the transcript produces in between: deal them out in contiguous runs of
    // equal cost (a slice costs its length), evaluate, exchange 23 scalars.
    const size_t cost[3] = {N, N, M};
    const int cnts[3] = {6, 15, 2};
    const vpin::fq* tabs[3] = {comb->d, comb_ops->d, comb_mem->d};
    const vpin::fq* eqs[3] = {eq_ops->d, eq_ops->d, eq_mem->d};
    size_t total = 0;
    for (int g = 0; g < 3; g++) total += cost[g] * (size_t)cnts[g];
    int owner[23];
    {
      size_t cum = 0;
      int sidx = 0;
      for (int g = 0; g < 3; g++)
        for (int i = 0; i < cnts[g]; i++, sidx++) {
          owner[sidx] = (int)std::min<size_t>((size_t)dz->world - 1, (cum + cost[g] / 2) * (size_t)dz->world / total);
          cum += cost[g];
        }
    }
    std::vector<Fq> mine23(23, Fq::zero()), all23(23 * (size_t)dz->world);
    int base = 0;
    for (int g = 0; g < 3; g++) {
      int first = -1, cnt = 0;
      for (int i = 0; i < cnts[g]; i++)
        if (owner[base + i] == dz->rank) { if (first < 0) first = i; cnt++; }
      if (cnt) {  // a rank's slices of one table are contiguous
        if ((rc = vpin::spark_slice_evals(c, tabs[g] + (size_t)first * cost[g], cost[g], cnt, eqs[g]))) return rc;
        for (int i = 0; i < cnt; i++) mine23[base + first + i] = hs[3 * i];
      }
      base += cnts[g];
    }
    if ((rc = vpin::comm_allgather_ctx(c, mine23.data(), all23.data(), 23 * 32, "hash_slice_evals"))) return rc;
    auto pick = [&](int sidx) { return all23[23 * (size_t)owner[sidx] + sidx]; };
    for (int i = 0; i < 6; i++) ev_derefs[i] = pick(i);
    for (int i = 0; i < 15; i++) ev_ops[i] = pick(6 + i);
    for (int i = 0; i < 2; i++) ev_mem[i] = pick(21 + i);
  } else if (one_pass) {
    TraceSpan ts("hash: derefs slices (one pass)");
    if ((rc = sp_derefs.run(c, *g_derefs, nullptr, 0, comb->d, 3, 6, rand_ops, ev_derefs))) return rc;
  } else {
    TraceSpan ts("hash: derefs slice evals");
    if ((rc = vpin::spark_slice_evals(c, comb->d, N, 6, eq_ops->d))) return rc;
    for (int i = 0; i < 6; i++) ev_derefs[i] = hs[3 * i];
  }
  DpLog pe_derefs, pe_ops, pe_mem;
  {
    // DerefsEvalProof::prove (:137-158, :90-135)
    tr.append_protocol_name("Derefs evaluation proof");
    std::vector<Fq> e8(8, Fq::zero());
    for (int i = 0; i < 6; i++) e8[i] = ev_derefs[i];
    tr.append_scalars("evals_ops_val", e8.data(), 8);
    std::vector<Fq> ch = tr.challenge_vector("challenge_combine_n_to_one", 3);
    Fq joint = combine_bot(e8, ch);
    std::vector<Fq> rj(ch);
    rj.insert(rj.end(), rand_ops.begin(), rand_ops.end());
    tr.append_scalar("joint_claim_eval", joint);
    TraceSpan ts("hash: polyeval derefs");
    std::vector<Fq> lz;
    if (sp_derefs.on && (rc = sp_derefs.combine(c, *g_derefs, ch, lz))) return rc;
    if ((rc = polyeval_prove_plain(c, *g_derefs, comb, rj, joint, tr, tape, pe_derefs, st ? comb_rows : nullptr,
                                   sp_derefs.on ? &lz : nullptr, sp_derefs.on ? &sp_derefs.Rv : nullptr))) return rc;
    if ((rc = vpin::comm_mark(c, "hash_bullet"))) return rc;
  }
  if (one_pass) {
    TraceSpan ts("hash: ops+mem slices (one pass)");
    if ((rc = sp_ops.run(c, *g_ops, d->idx, 12, d->vals, 4, 15, rand_ops, ev_ops))) return rc;
    if ((rc = sp_mem.run(c, *g_mem, d->idx + 12 * N, 2, nullptr, 1, 2, rand_mem, ev_mem))) return rc;
  } else if (!dz) {
    TraceSpan ts("hash: ops+mem slice evals");
    if ((rc = vpin::spark_slice_evals(c, comb_ops->d, N, 15, eq_ops->d))) return rc;
    for (int i = 0; i < 15; i++) ev_ops[i] = hs[3 * i];
    if ((rc = vpin::spark_slice_evals(c, comb_mem->d, M, 2, eq_mem->d))) return rc;
    for (int i = 0; i < 2; i++) ev_mem[i] = hs[3 * i];
  }
  {
    std::vector<Fq> e16(16, Fq::zero());
    for (int i = 0; i < 15; i++) e16[i] = ev_ops[i];  // row addr, row read_ts, col addr, col read_ts, val (comb_ops order)
    tr.append_scalars("claim_evals_ops", e16.data(), 16);
    std::vector<Fq> ch = tr.challenge_vector("challenge_combine_n_to_one", 4);
    Fq joint = combine_bot(e16, ch);
    std::vector<Fq> rj(ch);
    rj.insert(rj.end(), rand_ops.begin(), rand_ops.end());
    tr.append_scalar("joint_claim_eval_ops", joint);
    TraceSpan ts("hash: polyeval ops");
    std::vector<Fq> lz;
    if (sp_ops.on && (rc = sp_ops.combine(c, *g_ops, ch, lz))) return rc;
    if ((rc = polyeval_prove_plain(c, *g_ops, comb_ops, rj, joint, tr, tape, pe_ops, nullptr, sp_ops.on ? &lz : nullptr,
                                   sp_ops.on ? &sp_ops.Rv : nullptr))) return rc;
    if ((rc = vpin::comm_mark(c, "hash_bullet"))) return rc;
  }
  {
    std::vector<Fq> e2 = {ev_mem[0], ev_mem[1]};
    tr.append_scalars("claim_evals_mem", e2.data(), 2);
    std::vector<Fq> ch = tr.challenge_vector("challenge_combine_two_to_one", 1);
    Fq joint = combine_bot(e2, ch);
    std::vector<Fq> rj(ch);
    rj.insert(rj.end(), rand_mem.begin(), rand_mem.end());
    tr.append_scalar("joint_claim_eval_mem", joint);
    TraceSpan ts("hash: polyeval mem");
    std::vector<Fq> lz;
    if (sp_mem.on && (rc = sp_mem.combine(c, *g_mem, ch, lz))) return rc;
    if ((rc = polyeval_prove_plain(c, *g_mem, comb_mem, rj, joint, tr, tape, pe_mem, nullptr, sp_mem.on ? &lz : nullptr,
                                   sp_mem.on ? &sp_mem.Rv : nullptr))) return rc;
    if ((rc = vpin::comm_mark(c, "hash_bullet"))) return rc;
  }
  g_spark_timings[4] = secs(t0, Clock::now());

  // ---- bincode(R1CSEvalProof) (r1csinstance.rs:326-328; sparse_mlpoly.rs:1438-1441,1326-1329,1036-1042,698-707) ----
  w.u64(comm_derefs.size());
  for (auto& p : comm_derefs) w.point(p);
  for (int s = 0; s < 2; s++) { w.scalar(pl[s][0]); w_scalars(w, &pl[s][1], 3); w_scalars(w, &pl[s][4], 3); w.scalar(pl[s][7]); }
  w_scalars(w, dotp_left, 3);
  w_scalars(w, dotp_right, 3);
  write_batched(w, pf_mem);
  write_batched(w, pf_ops);
  w_scalars(w, &ev_ops[0], 3); w_scalars(w, &ev_ops[3], 3); w.scalar(ev_mem[0]);  // eval_row: addr, read_ts, audit_ts
  w_scalars(w, &ev_ops[6], 3); w_scalars(w, &ev_ops[9], 3); w.scalar(ev_mem[1]);  // eval_col
  w_scalars(w, &ev_ops[12], 3);                                                   // eval_val
  w_scalars(w, &ev_derefs[0], 3); w_scalars(w, &ev_derefs[3], 3);                 // eval_derefs
  write_dplog(w, pe_ops);
  write_dplog(w, pe_mem);
  write_dplog(w, pe_derefs);
  return VPIN_OK;
}

}  // namespace

extern "C" {

size_t vpin_spark_comm_bytes(const vpin_r1cs* inst) {
  if (!inst) return 0;
  Shape s = shape_of(inst->num_cons, inst->num_vars, inst->nnz);
  return 8 * 6 + 8 + 32 * ((size_t)1 << (s.v_ops / 2)) + 8 + 32 * ((size_t)1 << (s.v_mem / 2));
}

size_t vpin_snark_proof_max_bytes(const vpin_r1cs* inst) {
  if (!inst) return 0;
  Shape s = shape_of(inst->num_cons, inst->num_vars, inst->nnz);
  const size_t lgN = log2z(s.N), lgM = log2z(s.M);
  size_t b = vpin_sat_proof_max_bytes(inst->num_cons, inst->num_vars) + 96;
  b += 8 + 32 * ((size_t)1 << (s.v_derefs / 2)) + 64 * 32;
  b += lgN * (lgN * 104 + 64 + 24 * 32 + 64) + 18 * 32 + 64;
  b += lgM * (lgM * 104 + 64 + 8 * 32 + 64) + 64;
  b += 3 * (16 + 64 * 40 + 128 + 64) + 40 * 32 + 1024;
  return b;
}

int vpin_spark_prepare(vpin_ctx* c, size_t num_cons, size_t num_vars, size_t max_nnz) {
  if (!c || !vpin::is_pow2(num_cons) || !vpin::is_pow2(num_vars) || max_nnz == 0) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  const size_t nnz[3] = {max_nnz, max_nnz, max_nnz};
  const Shape s = shape_of(num_cons, num_vars, nnz);
  const PcGens* g = nullptr;
  int rc = get_view(c, std::max(s.v_ops, s.v_mem), &g);
  if (!rc) {
    // the host fixed-base tables of the other views' two blind generators (~3 ms each, kept per process and point): all at once
    auto* sg = static_cast<SparkGens*>(c->spark_cache);
    std::vector<size_t> idx;
    for (size_t ell : {s.v_ops, s.v_mem, s.v_derefs}) {
      const size_t R = (size_t)1 << (ell - ell / 2);
      if (R + 1 < sg->g.size()) { idx.push_back(R); idx.push_back(R + 1); }
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(host_threads())
    for (long k = 0; k < (long)idx.size(); k++) { FixedBase warm(sg->g[idx[(size_t)k]]); (void)warm; }
  }
  if (!rc) rc = get_view(c, s.v_ops, &g);
  if (!rc) rc = get_view(c, s.v_mem, &g);
  if (!rc) rc = get_view(c, s.v_derefs, &g);
  return rc;
}

int vpin_spark_gens_view(vpin_ctx* c, size_t ell, const vpin_gens** out, size_t* L, size_t* R) {
  if (!c || !out || !L || !R || ell < 2 || ell > 40) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  const PcGens* g = nullptr;
  int rc = get_view(c, ell, &g);
  if (rc) return rc;
  *out = g->dev; *L = g->L; *R = g->R;
  return VPIN_OK;
}

int vpin_dist_plan(int world, int owner_ops[12], int owner_dotp[6], int owner_mem[4]) {
  if (world < 1 || world > 12 || !owner_ops || !owner_dotp || !owner_mem) return VPIN_EINVAL;
  make_plan(world, owner_ops, owner_dotp, owner_mem);
  return VPIN_OK;
}

void vpin_spark_decomm_hot_cols(const vpin_spark_decomm* d, uint32_t out[3]) {
  for (int m = 0; m < 3; m++) out[m] = d ? d->hot_col[m] : 0xffffffffu;
}

void vpin_spark_decomm_free(vpin_ctx* c, vpin_spark_decomm* d) {
  if (!d) return;
  if (c) {
    (void)hipSetDevice(c->device);
    if (d->idx) vpin::dev_free(c, d->idx);
    if (d->vals) vpin::dev_free(c, d->vals);
    if (d->comb_ops) vpin_table_free(c, d->comb_ops);
    if (d->comb_mem) vpin_table_free(c, d->comb_mem);
  }
  delete d;
}

// second half of SNARK::encode: commit comb_ops / comb_mem (dense_mlpoly.rs:193-218 under b"gens_r1cs_eval") and
// serialise bincode(R1CSCommitment) (r1csinstance.rs:53-58, sparse_mlpoly.rs:332-338)
static int encode_commit(vpin_ctx* c, const Shape& s, std::unique_ptr<vpin_spark_decomm>& d, vpin::TraceLap& lap, size_t num_cons,
                         size_t num_vars, size_t num_inputs, vpin_spark_decomm** out, uint8_t* comm_out, size_t comm_cap,
                         size_t* comm_len, Clock::time_point t0) {
  auto fail = [&](int rc) { vpin_spark_decomm_free(c, d.release()); return rc; };
  int rc;
  const PcGens *g_ops = nullptr, *g_mem = nullptr;
  if ((rc = get_view(c, std::max(s.v_ops, s.v_mem), &g_ops)) || (rc = get_view(c, s.v_ops, &g_ops)) ||
      (rc = get_view(c, s.v_mem, &g_mem)))
    return fail(rc);
  lap("generators (views)");
  std::vector<CG> c_ops, c_mem;
  if ((rc = vpin::spark_comb_tables(c, d.get()))) return fail(rc);
  lap("comb tables");
  if ((rc = commit_noblind(c, g_ops, d->comb_ops, c_ops))) return fail(rc);
  lap("commit ops");
  if ((rc = commit_noblind(c, g_mem, d->comb_mem, c_mem))) return fail(rc);
  lap("commit mem");
  // (commit_noblind has synchronised the stream.)  The blocks go back to the context's POOL, never straight to the driver
  // (round 6): the proof that follows takes its 16N-scalar temporaries out of them, and a host that encodes per proof
  // (vpin_snark_prove from host buffers) would otherwise pay a 17 GB hipMalloc / hipFree pair every time -- 0.3 ms usually,
  // seconds every few calls (bench.py --trace E --host-buffers: 106 -> 975 ms/step when it struck).  A service that encodes
  // once and keeps proving trims the pool after its set-up (vpin_ctx_pool_trim), as bench.py does.
  if (!getenv("VPIN_KEEP_COMB")) vpin::spark_comb_release(c, d.get(), false);
  lap("comb release");
  if ((rc = vpin::spark_find_hot_cols(c, d.get()))) return fail(rc);
  lap("hot columns");
  Writer w;
  w.u64(num_cons); w.u64(num_vars); w.u64(num_inputs);
  w.u64(3); w.u64(s.N); w.u64(s.M);
  w.u64(c_ops.size()); for (auto& p : c_ops) w.point(p);
  w.u64(c_mem.size()); for (auto& p : c_mem) w.point(p);
  if (w.buf.size() > comm_cap) return fail(VPIN_ESHAPE);
  memcpy(comm_out, w.buf.data(), w.buf.size());
  *comm_len = w.buf.size();
  *out = d.release();
  g_spark_timings[0] = secs(t0, Clock::now());
  return VPIN_OK;
}

// SNARK::encode (lib.rs:347-359) = R1CSInstance::commit (r1csinstance.rs:309-322) =
// SparseMatPolynomial::multi_commit (sparse_mlpoly.rs:500-520)
int vpin_spark_encode(vpin_ctx* c, const vpin_r1cs* inst, vpin_spark_decomm** out, uint8_t* comm_out, size_t comm_cap,
                      size_t* comm_len) {
  if (!c || !inst || !out || !comm_out || !comm_len) return VPIN_EINVAL;
  if (!vpin::is_pow2(inst->num_cons) || !vpin::is_pow2(inst->num_vars) || inst->num_inputs >= inst->num_vars) return VPIN_ESHAPE;
  auto t0 = Clock::now();
  (void)hipSetDevice(c->device);
  const Shape s = shape_of(inst->num_cons, inst->num_vars, inst->nnz);
  const size_t N = s.N, M = s.M;
  if (N < 4 || M < 4 || M > ((size_t)1 << 32) || N > ((size_t)1 << 31)) return VPIN_ESHAPE;
  if (comm_cap < vpin_spark_comm_bytes(inst)) return VPIN_ESHAPE;
  for (int m = 0; m < 3; m++)
    if (inst->nnz[m] && (!inst->row[m] || !inst->col[m] || !inst->val[m])) return VPIN_EINVAL;
  std::unique_ptr<vpin_spark_decomm> d(new vpin_spark_decomm());
  d->num_cons = inst->num_cons; d->num_vars = inst->num_vars; d->num_inputs = inst->num_inputs;
  d->nx = s.nx; d->ny = s.ny; d->N = N; d->M = M;
  auto fail = [&](int rc) { vpin_spark_decomm_free(c, d.release()); return rc; };

  // sparse_to_dense_vecs (:368-380) + AddrTimestamps::new (:232-265).  Round 5: on the device -- the addresses go up as they
  // are (padded entries read address 0), the time stamps come from a stable sort of each side's 3N accesses (trace.hip); the
  // host used to walk both traces sequentially (60 % of this call for CNN A).  VPIN_ENCODE_HOST_TRACE=1: the old host walk (A/B).
  vpin::TraceLap lap(c, "spark_encode");
  int rc;
  if ((rc = vpin::dev_alloc(c, (12 * N + 2 * M) * 4, (void**)&d->idx))) return fail(rc);
  if (getenv("VPIN_ENCODE_HOST_TRACE")) {
    std::vector<uint32_t> idx(12 * N + 2 * M, 0);
    for (int m = 0; m < 3; m++) {
      for (size_t k = 0; k < inst->nnz[m]; k++) {
        if (inst->row[m][k] >= inst->num_cons || inst->col[m][k] >= 2 * inst->num_vars) return fail(VPIN_ESHAPE);
        idx[(size_t)m * N + k] = inst->row[m][k];
        idx[(size_t)(6 + m) * N + k] = inst->col[m][k];
      }
    }
#pragma omp parallel for schedule(static) num_threads(2)
    for (int side = 0; side < 2; side++) {
      uint32_t* audit = idx.data() + 12 * N + (size_t)side * M;
      for (int m = 0; m < 3; m++) {
        const uint32_t* addr = idx.data() + (size_t)(side * 6 + m) * N;
        uint32_t* ts = idx.data() + (size_t)(side * 6 + 3 + m) * N;
        for (size_t i = 0; i < N; i++) ts[i] = audit[addr[i]]++;
      }
    }
    lap("host idx + timestamps");
    if (hipMemcpyAsync(d->idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
      return fail(VPIN_EHIP);
  } else {
    vpin::DevBuf b_bad(c);
    if (b_bad.alloc(4)) return fail(VPIN_ENOMEM);
    if (hipMemsetAsync(b_bad.p, 0, 4, c->stream) != hipSuccess) return fail(VPIN_EHIP);
    for (int side = 0; side < 2; side++) {
      uint32_t* addr = d->idx + (size_t)side * 6 * N;
      if (hipMemsetAsync(addr, 0, 3 * N * 4, c->stream) != hipSuccess) return fail(VPIN_EHIP);
      for (int m = 0; m < 3; m++) {
        const uint32_t* src = side ? inst->col[m] : inst->row[m];
        if (inst->nnz[m] && hipMemcpyAsync(addr + (size_t)m * N, src, inst->nnz[m] * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess)
          return fail(VPIN_EHIP);
        if ((rc = vpin::spark_check_bounds(c, addr + (size_t)m * N, inst->nnz[m], (uint32_t)(side ? 2 * inst->num_vars : inst->num_cons),
                                           (uint32_t*)b_bad.p)))
          return fail(rc);
      }
      if ((rc = vpin::spark_trace_timestamps(c, addr, 3 * N, M, addr + 3 * N, d->idx + 12 * N + (size_t)side * M))) return fail(rc);
    }
    uint32_t bad = 0;
    if (hipMemcpyAsync(&bad, b_bad.p, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
      return fail(VPIN_EHIP);
    if (bad) return fail(VPIN_ESHAPE);  // lib.rs:171-178 InvalidIndex
    lap("device idx + timestamps");
  }
  if ((rc = vpin::dev_alloc(c, 3 * N * 32, (void**)&d->vals))) return fail(rc);
  if (hipMemsetAsync(d->vals, 0, 3 * N * 32, c->stream) != hipSuccess) return fail(VPIN_EHIP);
  for (int m = 0; m < 3; m++)
    if (inst->nnz[m] && hipMemcpyAsync(d->vals + (size_t)m * N, inst->val[m], inst->nnz[m] * 32, hipMemcpyHostToDevice,
                                       c->stream) != hipSuccess)
      return fail(VPIN_EHIP);
  if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(VPIN_EHIP);
  lap("upload + comb tables");

  return encode_commit(c, s, d, lap, inst->num_cons, inst->num_vars, inst->num_inputs, out, comm_out, comm_cap, comm_len, t0);
}

// SNARK::encode for a device-built gadget instance: the dense representation comes from gadget_dev.hip's
// closed forms instead of the host's sequential memory trace
int vpin_spark_encode_dev(vpin_ctx* c, const vpin_dev_instance* g, vpin_spark_decomm** out, uint8_t* comm_out, size_t comm_cap,
                          size_t* comm_len) {
  if (!c || !g || !g->r1cs || !out || !comm_out || !comm_len) return VPIN_EINVAL;
  auto t0 = Clock::now();
  (void)hipSetDevice(c->device);
  const size_t num_cons = g->r1cs->num_cons, num_vars = g->r1cs->num_vars;
  const Shape s = shape_of(num_cons, num_vars, g->nnz);
  const size_t N = s.N, M = s.M;
  if (N < 4 || M < 4 || M > ((size_t)1 << 32) || N > ((size_t)1 << 31)) return VPIN_ESHAPE;
  if (comm_cap < vpin_dev_instance_comm_bytes(g)) return VPIN_ESHAPE;
  std::unique_ptr<vpin_spark_decomm> d(new vpin_spark_decomm());
  d->num_cons = num_cons; d->num_vars = num_vars; d->num_inputs = g->num_inputs;
  d->nx = s.nx; d->ny = s.ny; d->N = N; d->M = M;
  auto fail = [&](int rc) { vpin_spark_decomm_free(c, d.release()); return rc; };
  vpin::TraceLap lap(c, "spark_encode_dev");
  int rc;
  if ((rc = vpin::dev_alloc(c, (12 * N + 2 * M) * 4, (void**)&d->idx))) return fail(rc);
  if ((rc = vpin::dev_alloc(c, 3 * N * 32, (void**)&d->vals))) return fail(rc);
  if ((rc = vpin::gadget_fill_decomm(c, g, d.get()))) return fail(rc);
  lap("trace + comb tables");
  return encode_commit(c, s, d, lap, num_cons, num_vars, g->num_inputs, out, comm_out, comm_cap, comm_len, t0);
}

size_t vpin_dev_instance_comm_bytes(const vpin_dev_instance* g) {
  if (!g || !g->r1cs) return 0;
  vpin_r1cs r{};
  r.num_cons = g->r1cs->num_cons; r.num_vars = g->r1cs->num_vars; r.num_inputs = g->num_inputs;
  for (int m = 0; m < 3; m++) r.nnz[m] = g->nnz[m];
  return vpin_spark_comm_bytes(&r);
}

size_t vpin_dev_instance_proof_max_bytes(const vpin_dev_instance* g) {
  if (!g || !g->r1cs) return 0;
  vpin_r1cs r{};
  r.num_cons = g->r1cs->num_cons; r.num_vars = g->r1cs->num_vars; r.num_inputs = g->num_inputs;
  for (int m = 0; m < 3; m++) r.nnz[m] = g->nnz[m];
  return vpin_snark_proof_max_bytes(&r);
}

// Kernel-level entry of the hash layer's one-pass slice evaluation (include/vpin_hip.h)
int vpin_poly_slices_bound(vpin_ctx* c, const vpin_table* Z, int nbits, int used, const uint8_t* r, size_t r_len, const uint8_t* ch,
                           uint8_t* evals_out, uint8_t* LZ_out) {
  if (!c || !Z || !Z->d || nbits < 0 || nbits > 4 || used < 1 || used > (1 << nbits) || !r || !evals_out) return VPIN_EINVAL;
  const size_t ell = r_len + (size_t)nbits;
  if (Z->len != ((size_t)1 << ell) || ell / 2 < (size_t)nbits) return VPIN_ESHAPE;
  PcGens pc;  // shape only
  pc.ell = ell; pc.L = (size_t)1 << (ell / 2); pc.R = (size_t)1 << (ell - ell / 2);
  std::vector<Fq> rand(r_len);
  memcpy(rand.data(), r, r_len * 32);
  SlicePass sp(c);
  std::vector<Fq> ev((size_t)used);
  int rc = sp.run(c, pc, nullptr, 0, Z->d, nbits, used, rand, ev.data());
  if (rc) return rc;
  memcpy(evals_out, ev.data(), (size_t)used * 32);
  if (ch && LZ_out) {
    std::vector<Fq> chv((size_t)nbits), LZ;
    memcpy(chv.data(), ch, (size_t)nbits * 32);
    if ((rc = sp.combine(c, pc, chv, LZ))) return rc;
    memcpy(LZ_out, LZ.data(), pc.R * 32);
  }
  return VPIN_OK;
}

// the same with the first n32 slices given as u32 (host memory, n32 x N values: the decommitment's addresses and timestamps,
// whose field images Scalar::from(v) are never formed) and the other used - n32 slices as a table of field elements
int vpin_poly_slices_bound_u32(vpin_ctx* c, const uint32_t* slices_u32, int n32, const vpin_table* Zfq, int nbits, int used,
                               const uint8_t* r, size_t r_len, const uint8_t* ch, uint8_t* evals_out, uint8_t* LZ_out) {
  if (!c || !slices_u32 || n32 < 1 || nbits < 0 || nbits > 4 || used < n32 || used > (1 << nbits) || !r || !evals_out) return VPIN_EINVAL;
  const size_t ell = r_len + (size_t)nbits, N = (size_t)1 << r_len;
  if (ell / 2 < (size_t)nbits || (used > n32 && (!Zfq || !Zfq->d || Zfq->len < (size_t)(used - n32) * N))) return VPIN_ESHAPE;
  PcGens pc;  // shape only
  pc.ell = ell; pc.L = (size_t)1 << (ell / 2); pc.R = (size_t)1 << (ell - ell / 2);
  (void)hipSetDevice(c->device);
  vpin::DevBuf b32(c);
  if (b32.alloc((size_t)n32 * N * 4)) return VPIN_ENOMEM;
  if (hipMemcpyAsync(b32.p, slices_u32, (size_t)n32 * N * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) return VPIN_EHIP;
  std::vector<Fq> rand(r_len);
  memcpy(rand.data(), r, r_len * 32);
  SlicePass sp(c);
  std::vector<Fq> ev((size_t)used);
  int rc = sp.run(c, pc, (const uint32_t*)b32.p, n32, used > n32 ? Zfq->d : nullptr, nbits, used, rand, ev.data());
  if (rc) return rc;
  memcpy(evals_out, ev.data(), (size_t)used * 32);
  if (ch && LZ_out) {
    std::vector<Fq> chv((size_t)nbits), LZ;
    memcpy(chv.data(), ch, (size_t)nbits * 32);
    if ((rc = sp.combine(c, pc, chv, LZ))) return rc;
    memcpy(LZ_out, LZ.data(), pc.R * 32);
  }
  return VPIN_OK;
}

// my_lib_prove (commit_test.rs:59-133) in full: R1CSProof, inst_evals, R1CSEvalProof -> bincode(SNARK)
int vpin_snark_prove_resident(vpin_ctx* c, const vpin_r1cs_dev* dinst, const vpin_spark_decomm* decomm,
                              const vpin_table* vars_para, const vpin_table* vars_input, const vpin_table* vars,
                              const uint8_t* inputs, const uint8_t seed_commit64[64], const uint8_t seed_proof64[64],
                              uint8_t* proof_out, size_t proof_cap, size_t* proof_len, uint8_t* comm_para_out,
                              uint8_t* comm_input_out) {
  if (!c || !dinst || !decomm || !vars_para || !vars_input || !vars || !seed_commit64 || !seed_proof64 || !proof_out ||
      !proof_len || !comm_para_out || !comm_input_out)
    return VPIN_EINVAL;
  size_t nv, ncons, ni;
  vpin_r1cs_dims(dinst, &ncons, &nv, &ni);
  if (vars_para->len != nv || vars_input->len != nv || vars->len != nv || (ni && !inputs)) return VPIN_ESHAPE;
  if (decomm->num_cons != ncons || decomm->num_vars != nv || decomm->num_inputs != ni) return VPIN_ESHAPE;
  auto t0 = Clock::now();
  vpin::AltStreamGuard alt_guard(c);  // vpin_ctx_set_cumask_after_phase1: back on the first stream however this returns
  Transcript tr("snark_example"), tape("snark_example");
  uint8_t ie[96];
  std::vector<Fq> rx(log2z(ncons)), ry(log2z(2 * nv));
  size_t sat_len = 0;
  // collective over c->comm: a failure of this rank alone (memory, a HIP error) fails the peers' next wait (comm_leave)
  int rc = sat_prove_core(c, dinst, nv, ncons, ni, vars_para, vars_input, vars, inputs, seed_commit64, seed_proof64, proof_out,
                          proof_cap, &sat_len, comm_para_out, comm_input_out, ie, B(rx.data()), B(ry.data()), &tr, &tape);
  if (rc) return vpin::comm_leave(c->comm, rc);
  if (c->progress_flag) *c->progress_flag = 1;
  g_spark_timings[5] = secs(t0, Clock::now());
  Writer w;
  w.bytes(ie, 96);  // SNARK.inst_evals (lib.rs:334-338)
  Fq evals[3];
  memcpy(evals, ie, 96);
  if ((rc = spark_prove(c, decomm, rx, ry, evals, tr, tape, w))) return vpin::comm_leave(c->comm, rc);
  if (sat_len + w.buf.size() > proof_cap) return VPIN_ESHAPE;
  memcpy(proof_out + sat_len, w.buf.data(), w.buf.size());
  *proof_len = sat_len + w.buf.size();
  g_spark_timings[6] = secs(t0, Clock::now());
  return VPIN_OK;
}

// proof_point_mult.rs:38-94 from host buffers: SNARK::encode, then my_lib_prove in full
int vpin_snark_prove(vpin_ctx* c, const vpin_r1cs* inst, const uint8_t* vars_para, const uint8_t* vars_input,
                     const uint8_t* vars, const uint8_t* inputs, const uint8_t seed_commit64[64], const uint8_t seed_proof64[64],
                     uint8_t* proof_out, size_t proof_cap, size_t* proof_len, uint8_t* comm_out, size_t comm_cap, size_t* comm_len,
                     uint8_t* comm_para_out, uint8_t* comm_input_out) {
  if (!c || !inst || !vars_para || !vars_input || !vars) return VPIN_EINVAL;
  const size_t nv = inst->num_vars;
  if (!vpin::is_pow2(nv) || !vpin::is_pow2(inst->num_cons) || inst->num_inputs >= nv) return VPIN_ESHAPE;
  vpin_r1cs_dev* dinst = nullptr;
  vpin_spark_decomm* decomm = nullptr;
  const bool trace = getenv("VPIN_CLI_TRACE") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    (void)hipStreamSynchronize(c->stream);
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[vpin_snark_prove] %-18s %9.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  };
  int rc = vpin_r1cs_upload(c, inst, &dinst);
  if (rc) return rc;
  lap("r1cs_upload");
  TableGuard tg(c);
  vpin_table *d_para = nullptr, *d_input = nullptr, *d_vars = nullptr;
  rc = vpin_spark_encode(c, inst, &decomm, comm_out, comm_cap, comm_len);
  lap("spark_encode");
  if (!rc) rc = vpin_table_upload(c, vars_para, nv, &d_para);
  if (!rc) { tg.add(d_para); rc = vpin_table_upload(c, vars_input, nv, &d_input); }
  if (!rc) { tg.add(d_input); rc = vpin_table_upload(c, vars, nv, &d_vars); }
  lap("witness_upload");
  if (!rc) {
    tg.add(d_vars);
    rc = vpin_snark_prove_resident(c, dinst, decomm, d_para, d_input, d_vars, inputs, seed_commit64, seed_proof64, proof_out,
                                   proof_cap, proof_len, comm_para_out, comm_input_out);
  }
  lap("prove_resident");
  vpin_spark_decomm_free(c, decomm);
  vpin_r1cs_free(c, dinst);
  return rc;
}

// the same span for a device-built instance (vpin_gadget_point_*_dev): nothing crosses PCIe but the proof
int vpin_snark_prove_dev(vpin_ctx* c, const vpin_dev_instance* g, const uint8_t seed_commit64[64], const uint8_t seed_proof64[64],
                         uint8_t* proof_out, size_t proof_cap, size_t* proof_len, uint8_t* comm_out, size_t comm_cap,
                         size_t* comm_len, uint8_t* comm_para_out, uint8_t* comm_input_out) {
  if (!c || !g || !g->r1cs) return VPIN_EINVAL;
  vpin_spark_decomm* decomm = nullptr;
  vpin::TraceLap lap(c, "vpin_snark_prove_dev");
  int rc = vpin_spark_encode_dev(c, g, &decomm, comm_out, comm_cap, comm_len);
  lap("spark_encode_dev");
  if (!rc)
    rc = vpin_snark_prove_resident(c, g->r1cs, decomm, g->vars_para, g->vars_input, g->vars, g->num_inputs ? g->inputs : nullptr,
                                   seed_commit64, seed_proof64, proof_out, proof_cap, proof_len, comm_para_out, comm_input_out);
  lap("prove_resident");
  vpin_spark_decomm_free(c, decomm);
  return rc;
}

// [0] encode, [1] derefs + commit, [2] network build, [3] product-layer proofs, [4] hash-layer proofs,
// [5] sat part, [6] whole prove
void vpin_spark_last_timings(double out[8]) { memcpy(out, g_spark_timings, sizeof g_spark_timings); }

}  // extern "C"
