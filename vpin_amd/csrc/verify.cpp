// verify.cpp -- the verifier side of vPIN's Spartan SNARK (host C++; the two large fixed-base MSMs of
// each evaluation proof run on the device window tables).
//
// C++ counterpart of
//   vPIN_proof_generation/src/commit_test.rs:340-496  my_r1csproof_verify
//   vPIN_proof_generation/src/commit_test.rs:498-548  my_lib_verify
//   Spartan/src/sumcheck.rs:27-61,84-183              SumcheckInstanceProof::verify, ZKSumcheckInstanceProof::verify
//   Spartan/src/nizk/mod.rs:68-104,157-190,246-292,376-407,533-588   Sigma-protocol verifiers, DotProductProofLog::verify
//   Spartan/src/nizk/bullet.rs:134-231                BulletReductionProof::verify
//   Spartan/src/dense_mlpoly.rs:381-419               PolyEvalProof::verify / verify_plain
//   Spartan/src/sparse_mlpoly.rs:160-214,851-1032,1229-1322,1372-1434,1535-1571   SPARK verifiers
//   Spartan/src/product_tree.rs:387-485               ProductCircuitEvalProofBatched::verify
// It exists so that `vpin_prove` can close the reference binary's loop ("Proof verification
// successful!"); it shares no code with the test-side checker.
#include <functional>

#include "host/prover_common.h"

namespace {

using namespace vpin_host;
using namespace vpin_prover;

// VPIN_VERIFY_TRACE=1: spans of one verification on stderr (development aid)
struct VSpan {
  const char* name;
  std::chrono::steady_clock::time_point t0;
  static bool on() { static const bool v = getenv("VPIN_VERIFY_TRACE") != nullptr; return v; }
  explicit VSpan(const char* n) : name(n), t0(std::chrono::steady_clock::now()) {}
  ~VSpan() {
    if (on()) fprintf(stderr, "[verify] %-28s %8.3f ms\n", name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
};

// The transcript is sequential, the group equations it leads to are not: the sigma-protocol checks of the ZK sum-check
// rounds (two variable-base and up to seven fixed-base multiplications each, ~50 rounds per proof) are collected while
// the transcript runs and evaluated across the host cores afterwards.
struct Deferred {
  std::vector<std::function<bool()>> jobs;
  void add(std::function<bool()> f) { jobs.push_back(std::move(f)); }
  bool run() {
    bool ok = true;
    const int nt = jobs.size() >= 4 ? host_threads() : 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt) reduction(&& : ok)
    for (long i = 0; i < (long)jobs.size(); i++) ok = ok && jobs[(size_t)i]();
    jobs.clear();
    return ok;
  }
};

struct Reader {
  const uint8_t* p;
  size_t len, pos = 0;
  bool bad = false;
  Reader(const uint8_t* p_, size_t n) : p(p_), len(n) {}
  void bytes(void* dst, size_t n) {
    if (pos + n > len) { bad = true; memset(dst, 0, n); return; }
    memcpy(dst, p + pos, n);
    pos += n;
  }
  uint64_t u64() { uint64_t v = 0; bytes(&v, 8); return v; }
  Fq scalar() {
    Fq s;
    bytes(s.l, 32);
    // Scalar deserialises as raw Montgomery limbs (derive(Deserialize) on Scalar([u64;4])); a
    // non-reduced image would not be a field element: reject
    static const uint64_t Q[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL};
    for (int i = 3; i >= 0; i--) {
      if (s.l[i] < Q[i]) break;
      if (s.l[i] > Q[i] || i == 0) { bad = true; break; }
    }
    return s;
  }
  CG point() { CG c; bytes(c.b, 32); return c; }
  bool scalars(Fq* v, size_t n) {
    if (u64() != n) { bad = true; return false; }
    for (size_t i = 0; i < n; i++) v[i] = scalar();
    return !bad;
  }
  bool points(std::vector<CG>& v, size_t n) {
    if (u64() != n) { bad = true; return false; }
    v.resize(n);
    for (auto& c : v) c = point();
    return !bad;
  }
};

static bool decompress(Point& out, const CG& c) { return Point::decompress(out, c.b); }
static bool same(const CG& a, const CG& b) { return memcmp(a.b, b.b, 32) == 0; }

// variable-base MSM on the host cores (the verifier's C_LZ = <L, C> over the row commitments)
static Point msm_var(const Fq* s, const Point* P, size_t n) {
  const int nt = n >= 64 ? host_threads() : 1;
  std::vector<Point> part(nt, Point::identity());
#pragma omp parallel for schedule(static) num_threads(nt)
  for (long i = 0; i < (long)n; i++) {
    Point& acc = part[omp_get_thread_num()];
    if (!s[i].is_zero()) acc = acc + P[i].mul(s[i]);
  }
  Point r = Point::identity();
  for (auto& p : part) r = r + p;
  return r;
}

// ---- Sigma protocols -----------------------------------------------------------------------

struct KnowP { CG alpha; Fq z1, z2; };
static bool knowledge_verify(const KnowP& pf, const Mcg& g, Transcript& tr, const CG& C) {
  tr.append_protocol_name("knowledge proof");
  tr.append_point("C", C.b);
  tr.append_point("alpha", pf.alpha.b);
  Fq c = tr.challenge_scalar("c");
  Point pC, pa;
  if (!decompress(pC, C) || !decompress(pa, pf.alpha)) return false;
  return commit1(pf.z1, pf.z2, g).equals(pC.mul(c) + pa);
}

struct EqP { CG alpha; Fq z; };
static bool equality_verify(const EqP& pf, const Mcg& g, Transcript& tr, const CG& C1, const CG& C2) {
  tr.append_protocol_name("equality proof");
  tr.append_point("C1", C1.b);
  tr.append_point("C2", C2.b);
  tr.append_point("alpha", pf.alpha.b);
  Fq c = tr.challenge_scalar("c");
  Point p1, p2, pa;
  if (!decompress(p1, C1) || !decompress(p2, C2) || !decompress(pa, pf.alpha)) return false;
  return g.h->mul(pf.z).equals((p1 - p2).mul(c) + pa);
}

struct ProdP { CG alpha, beta, delta; Fq z[5]; };
static bool product_check(const Point& P, const Point& X, const Fq& c, const Point& lhs_g, const Point& h, const Fq& z1, const Fq& z2) {
  return (P + X.mul(c)).equals(lhs_g.mul(z1) + h.mul(z2));
}
static bool product_verify(const ProdP& pf, const Mcg& g, Transcript& tr, const CG& X, const CG& Y, const CG& Z) {
  tr.append_protocol_name("product proof");
  tr.append_point("X", X.b);
  tr.append_point("Y", Y.b);
  tr.append_point("Z", Z.b);
  tr.append_point("alpha", pf.alpha.b);
  tr.append_point("beta", pf.beta.b);
  tr.append_point("delta", pf.delta.b);
  Fq c = tr.challenge_scalar("c");
  Point pX, pY, pZ, pa, pb, pd;
  if (!decompress(pX, X) || !decompress(pY, Y) || !decompress(pZ, Z) || !decompress(pa, pf.alpha) ||
      !decompress(pb, pf.beta) || !decompress(pd, pf.delta))
    return false;
  const Point G = g.G[0]->base, H = g.h->base;
  return product_check(pa, pX, c, G, H, pf.z[0], pf.z[1]) && product_check(pb, pY, c, G, H, pf.z[2], pf.z[3]) &&
         product_check(pd, pZ, c, pX, H, pf.z[2], pf.z[4]);
}

struct DotP { CG delta, beta; std::vector<Fq> z; Fq z_delta, z_beta; };
// DotProductProof::verify (nizk/mod.rs:376-407) over the <= 4 sum-check generators: the transcript part now, the two group
// equations into `later` (pf, g1, gn outlive it; pCx, pCy are the decoded Cx, Cy)
static bool dotproduct_verify(const DotP& pf, const Mcg& g1, const Mcg& gn, Transcript& tr, const Fq* a, int n, const CG& Cx,
                              const Point& pCx, const CG& Cy, const Point& pCy, Deferred& later) {
  if ((int)pf.z.size() != n || gn.n != n) return false;
  tr.append_protocol_name("dot product proof");
  tr.append_point("Cx", Cx.b);
  tr.append_point("Cy", Cy.b);
  tr.append_scalars("a", a, n);
  tr.append_point("delta", pf.delta.b);
  tr.append_point("beta", pf.beta.b);
  const Fq c = tr.challenge_scalar("c");
  Fq dotp = Fq::zero();
  for (int i = 0; i < n; i++) dotp = dotp + pf.z[i] * a[i];
  const DotP* pp = &pf;
  const Mcg *p1 = &g1, *pn = &gn;
  later.add([=]() {
    Point pd, pb;
    if (!decompress(pd, pp->delta) || !decompress(pb, pp->beta)) return false;
    return (pCx.mul(c) + pd).equals(commit(pp->z.data(), n, pp->z_delta, *pn)) && (pCy.mul(c) + pb).equals(commit1(dotp, pp->z_beta, *p1));
  });
  return true;
}

// ---- ZK sum-check (sumcheck.rs:84-183) -------------------------------------------------------

struct ZkScP { std::vector<CG> comm_polys, comm_evals; std::vector<DotP> proofs; };

static bool read_zksc(Reader& r, ZkScP& p, int rounds, int deg) {
  if (!r.points(p.comm_polys, (size_t)rounds) || !r.points(p.comm_evals, (size_t)rounds)) return false;
  if (r.u64() != (uint64_t)rounds) return false;
  p.proofs.resize(rounds);
  for (auto& d : p.proofs) {
    d.delta = r.point(); d.beta = r.point();
    d.z.resize(deg + 1);
    if (!r.scalars(d.z.data(), (size_t)deg + 1)) return false;
    d.z_delta = r.scalar(); d.z_beta = r.scalar();
  }
  return !r.bad;
}

static bool zksc_verify(const ZkScP& pf, const CG& comm_claim, int rounds, int deg, const Mcg& g1, const Mcg& gn, Transcript& tr,
                        CG& comm_out, std::vector<Fq>& r_out, Deferred& later) {
  if (gn.n != deg + 1 || (int)pf.comm_polys.size() != rounds || rounds < 1) return false;
  VSpan vs(" zk sum-check (transcript)");
  r_out.clear();
  // every commitment of the proof is decoded once, across the cores (the round loop needs them in transcript order)
  std::vector<Point> p_polys(rounds), p_evals(rounds);
  Point p_claim;
  bool dec_ok = decompress(p_claim, comm_claim);
#pragma omp parallel for schedule(static) num_threads(host_threads()) reduction(&& : dec_ok)
  for (int i = 0; i < rounds; i++) dec_ok = dec_ok && decompress(p_polys[i], pf.comm_polys[i]) && decompress(p_evals[i], pf.comm_evals[i]);
  if (!dec_ok) return false;
  for (int i = 0; i < rounds; i++) {
    tr.append_point("comm_poly", pf.comm_polys[i].b);
    Fq r_i = tr.challenge_scalar("challenge_nextround");
    const CG& claim = i == 0 ? comm_claim : pf.comm_evals[i - 1];
    const CG& ev = pf.comm_evals[i];
    tr.append_point("comm_claim_per_round", claim.b);
    tr.append_point("comm_eval", ev.b);
    std::vector<Fq> w = tr.challenge_vector("combine_two_claims_to_one", 2);
    // the round's target commitment goes into the transcript: the one group operation that stays in the sequence
    const Point p_target = Point::mul2(w[0], i == 0 ? p_claim : p_evals[i - 1], w[1], p_evals[i]);
    CG target = compress(p_target);
    std::vector<Fq> a(deg + 1);
    Fq pw = Fq::one();
    for (int j = 0; j <= deg; j++) {
      Fq a_sc = j == 0 ? Fq::one() + Fq::one() : Fq::one();
      a[j] = w[0] * a_sc + w[1] * pw;
      pw = pw * r_i;
    }
    if (!dotproduct_verify(pf.proofs[i], g1, gn, tr, a.data(), deg + 1, pf.comm_polys[i], p_polys[i], target, p_target, later)) return false;
    r_out.push_back(r_i);
  }
  comm_out = pf.comm_evals[rounds - 1];
  return true;
}

// ---- DotProductProofLog / PolyEvalProof -------------------------------------------------------

struct DpLogP { std::vector<CG> Lv, Rv; CG delta, beta; Fq z1, z2; };

static bool read_dplog(Reader& r, DpLogP& p, size_t lgR) {
  if (!r.points(p.Lv, lgR) || !r.points(p.Rv, lgR)) return false;
  p.delta = r.point(); p.beta = r.point(); p.z1 = r.scalar(); p.z2 = r.scalar();
  return !r.bad;
}

// DotProductProofLog::verify (nizk/mod.rs:533-588) + BulletReductionProof::verify (bullet.rs:134-231);
// G_hat = <s, G> over the R generators is a fixed-base MSM on the device table
static bool dplog_verify(vpin_ctx* c, const DpLogP& pf, const PcGens& pc, Transcript& tr, const std::vector<Fq>& a, const CG& Cx,
                         const CG& Cy) {
  const size_t n = pc.R, lg = log2z(n);
  if (pf.Lv.size() != lg || a.size() != n) return false;
  VSpan vs("  dplog");
  tr.append_protocol_name("dot product proof (log)");
  tr.append_point("Cx", Cx.b);
  tr.append_point("Cy", Cy.b);
  tr.append_scalars("a", a.data(), n);
  Fq r = tr.challenge_scalar("r");
  Point pCx, pCy;
  if (!decompress(pCx, Cx) || !decompress(pCy, Cy)) return false;
  const Point Gs = pc.fb_gR.mul(r);  // gens_1.scale(&r)
  const Point Gamma = pCx + pCy.mul(r);
  std::vector<Fq> u(lg), ui(lg);
  for (size_t i = 0; i < lg; i++) {
    tr.append_point("L", pf.Lv[i].b);
    tr.append_point("R", pf.Rv[i].b);
    u[i] = tr.challenge_scalar("u");
    if (u[i].is_zero()) return false;
    ui[i] = u[i].invert();
  }
  Fq allinv = Fq::one();
  for (size_t i = 0; i < lg; i++) allinv = allinv * ui[i];
  std::vector<Fq> usq(lg), uisq(lg);
  for (size_t i = 0; i < lg; i++) { usq[i] = u[i] * u[i]; uisq[i] = ui[i] * ui[i]; }
  std::vector<Fq> s(n + 2, Fq::zero());
  s[0] = allinv;
  for (size_t i = 1; i < n; i++) {
    size_t lg_i = 0;
    while (((size_t)2 << lg_i) <= i) lg_i++;
    const size_t k = (size_t)1 << lg_i;
    s[i] = s[i - k] * usq[(lg - 1) - lg_i];
  }
  Point G_hat;
  if (msm_rows_host_sum(c, pc.dev, s.data(), 1, n + 2, &G_hat)) return false;
  Fq a_hat = Fq::zero();
  for (size_t i = 0; i < n; i++) a_hat = a_hat + a[i] * s[i];
  Point Gamma_hat = Gamma;
  {
    std::vector<Point> term(lg);
    bool dec_ok = true;
#pragma omp parallel for schedule(dynamic, 1) num_threads(host_threads()) reduction(&& : dec_ok)
    for (long i = 0; i < (long)lg; i++) {
      Point pl, pr;
      const bool ok_i = decompress(pl, pf.Lv[(size_t)i]) && decompress(pr, pf.Rv[(size_t)i]);
      if (ok_i) term[(size_t)i] = Point::mul2(usq[(size_t)i], pl, uisq[(size_t)i], pr);
      dec_ok = dec_ok && ok_i;
    }
    if (!dec_ok) return false;
    for (size_t i = 0; i < lg; i++) Gamma_hat = Gamma_hat + term[i];
  }
  tr.append_point("delta", pf.delta.b);
  tr.append_point("beta", pf.beta.b);
  Fq cc = tr.challenge_scalar("c");
  Point pb, pd;
  if (!decompress(pb, pf.beta) || !decompress(pd, pf.delta)) return false;
  Point lhs = (Gamma_hat.mul(cc) + pb).mul(a_hat) + pd;
  Point rhs = (G_hat + Gs.mul(a_hat)).mul(pf.z1) + pc.fb_h.mul(pf.z2);
  return lhs.equals(rhs);
}

// PolyEvalProof::verify (dense_mlpoly.rs:381-404): C_Zr given
static bool polyeval_verify(vpin_ctx* c, const DpLogP& pf, const PcGens& pc, Transcript& tr, const Fq* r, const CG& C_Zr,
                            const std::vector<CG>& comm) {
  if (comm.size() != pc.L) return false;
  VSpan vs(" polyeval");
  tr.append_protocol_name("polynomial evaluation proof");
  const size_t left = pc.ell / 2, right = pc.ell - left;
  std::vector<Fq> Lv(pc.L), Rv(pc.R);
  host_eq(r, left, Lv.data());
  host_eq(r + left, right, Rv.data());
  CG C_LZ;
  static const bool host_only = getenv("VPIN_VERIFY_HOST_MSM") != nullptr;
  bool on_device = false;
  if (pc.L >= 128 && !host_only) {
    // <L, C> over the row commitments on the device (msm_var.hip): decompression and the scalar multiplications, one lane each
    const int rc = vpin_msm(c, B(Lv.data()), comm[0].b, pc.L, C_LZ.b, nullptr);
    if (rc == VPIN_EVERIFY) return false;  // a commitment that does not decode: the proof's fault
    on_device = rc == VPIN_OK;             // VPIN_ENOMEM / VPIN_EHIP are the machine's fault: the host path decides
  }
  if (!on_device) {
    std::vector<Point> Cd(pc.L);
    bool ok = true;
#pragma omp parallel for schedule(static) num_threads(pc.L >= 64 ? host_threads() : 1) reduction(&& : ok)
    for (long i = 0; i < (long)pc.L; i++) ok = ok && decompress(Cd[i], comm[i]);
    if (!ok) return false;
    C_LZ = compress(msm_var(Lv.data(), Cd.data(), pc.L));
  }
  return dplog_verify(c, pf, pc, tr, Rv, C_LZ, C_Zr);
}
// PolyEvalProof::verify_plain (dense_mlpoly.rs:406-419)
static bool polyeval_verify_plain(vpin_ctx* c, const DpLogP& pf, const PcGens& pc, Transcript& tr, const Fq* r, const Fq& Zr,
                                  const std::vector<CG>& comm) {
  CG C_Zr = compress(commit1(Zr, Fq::zero(), pc.gens_1));
  return polyeval_verify(c, pf, pc, tr, r, C_Zr, comm);
}

// ---- sat proof ---------------------------------------------------------------------------------

struct SatGensV {  // R1CSGens (r1csproof.rs:49-89) for the verifier
  size_t ell, L, R;
  std::vector<Point> g;
  FixedBase fb[5];
  PcGens pc;
  Mcg gens_1, gens_3, gens_4;
};

static int make_pc(vpin_ctx* c, const char* label, const std::vector<Point>& g, size_t ell, size_t budget_gb, PcGens& pc) {
  const size_t left = ell / 2, R = (size_t)1 << (ell - left), nb = R + 2;
  pc.ell = ell; pc.L = (size_t)1 << left; pc.R = R;
  int rc = vpin_gens_shared(c, label, nullptr, nb, 0, &pc.dev);
  if (rc == VPIN_EINVAL) {
    std::vector<uint8_t> xyzt(128 * nb);
    for (size_t i = 0; i < nb; i++) g[i].to_xyzt(xyzt.data() + 128 * i);
    rc = vpin_gens_shared(c, label, xyzt.data(), nb, budget_gb, &pc.dev);
  }
  if (rc) return rc;
  pc.fb_gR = FixedBase(g[R]);
  pc.fb_h = FixedBase(g[R + 1]);
  pc.bind_views();
  return VPIN_OK;
}

// The verifier's generator sets live in the context, like the prover's (R1CSGens::new / R1CSCommitmentGens::new are
// functions of the sizes only): deriving R + 2 hash-to-group points and seven fixed-base tables per VERIFICATION was most
// of a small proof's verification time.
struct VCache {
  std::map<size_t, std::unique_ptr<SatGensV>> sat;       // by num_vars
  std::vector<Point> g_eval;                              // b"gens_r1cs_eval" stream prefix derived so far
  std::map<size_t, std::unique_ptr<PcGens>> eval_views;   // by ell
};
static void vcache_free(vpin_ctx* c) {
  delete static_cast<VCache*>(c->verify_cache);
  c->verify_cache = nullptr;
}
static VCache* vcache(vpin_ctx* c) {
  if (!c->verify_cache) { c->verify_cache = new VCache(); c->verify_cache_free = vcache_free; }
  return static_cast<VCache*>(c->verify_cache);
}

static int sat_gens_v(vpin_ctx* c, size_t num_vars, SatGensV** out) {
  VCache* vc = vcache(c);
  auto it = vc->sat.find(num_vars);
  if (it != vc->sat.end()) { *out = it->second.get(); return VPIN_OK; }
  VSpan vs("setup: sat generators");
  std::unique_ptr<SatGensV> sgp(new SatGensV());
  SatGensV& sg = *sgp;
  sg.ell = log2z(num_vars);
  const size_t left = sg.ell / 2;
  sg.L = (size_t)1 << left;
  sg.R = (size_t)1 << (sg.ell - left);
  const size_t nb = sg.R + 2 < 5 ? 5 : sg.R + 2;
  derive_gens(sg.g, nb, "gens_r1cs_sat", c);
  for (int i = 0; i < 5; i++) sg.fb[i] = FixedBase(sg.g[i]);
  int rc = make_pc(c, "gens_r1cs_sat", sg.g, sg.ell, 0, sg.pc);
  if (rc) return rc;
  sg.gens_1 = sg.pc.gens_1;
  sg.gens_3 = Mcg{3, {&sg.fb[0], &sg.fb[1], &sg.fb[2], nullptr}, &sg.fb[3]};
  sg.gens_4 = Mcg{4, {&sg.fb[0], &sg.fb[1], &sg.fb[2], &sg.fb[3]}, &sg.fb[4]};
  *out = sgp.get();
  vc->sat[num_vars] = std::move(sgp);
  return VPIN_OK;
}

// PolyCommitmentGens::new(ell, b"gens_r1cs_eval") for the verifier
static int eval_view_v(vpin_ctx* c, size_t ell, const PcGens** out) {
  VCache* vc = vcache(c);
  auto it = vc->eval_views.find(ell);
  if (it != vc->eval_views.end()) { *out = it->second.get(); return VPIN_OK; }
  VSpan vs("setup: eval generators");
  const size_t nb = ((size_t)1 << (ell - ell / 2)) + 2;
  if (vc->g_eval.size() < nb) derive_gens(vc->g_eval, nb, "gens_r1cs_eval", c);
  std::unique_ptr<PcGens> v(new PcGens());
  int rc = make_pc(c, "gens_r1cs_eval", vc->g_eval, ell, 0, *v);
  if (rc) return rc;
  *out = v.get();
  vc->eval_views[ell] = std::move(v);
  return VPIN_OK;
}

// my_r1csproof_verify (commit_test.rs:340-496) + the claims my_lib_verify appends (:521-524).
// whole_snark: inst_evals are read from the bytes after the R1CSProof.  Returns false on rejection.
static bool sat_verify(vpin_ctx* c, Reader& r, size_t num_cons, size_t num_vars, const Fq* inputs, size_t num_inputs,
                       const Fq* inst_evals_in, Fq inst_evals[3], const uint8_t* comm_para, const uint8_t* comm_input,
                       Transcript& tr, std::vector<Fq>& rx, std::vector<Fq>& ry, int* err) {
  SatGensV* sgp = nullptr;
  if ((*err = sat_gens_v(c, num_vars, &sgp))) return false;
  SatGensV& sg = *sgp;
  const size_t L = sg.L;
  const int nrx = (int)log2z(num_cons), nry = (int)log2z(2 * num_vars);
  std::vector<CG> comm_vars;
  if (!r.points(comm_vars, L)) return false;
  ZkScP sc1, sc2;
  if (!read_zksc(r, sc1, nrx, 3)) return false;
  CG comm_Az = r.point(), comm_Bz = r.point(), comm_Cz = r.point(), comm_prod = r.point();
  KnowP pok; pok.alpha = r.point(); pok.z1 = r.scalar(); pok.z2 = r.scalar();
  ProdP pp; pp.alpha = r.point(); pp.beta = r.point(); pp.delta = r.point();
  for (int i = 0; i < 5; i++) pp.z[i] = r.scalar();
  EqP eq1; eq1.alpha = r.point(); eq1.z = r.scalar();
  if (!read_zksc(r, sc2, nry, 2)) return false;
  CG comm_vars_at_ry = r.point();
  DpLogP pe;
  if (!read_dplog(r, pe, log2z(sg.R))) return false;
  EqP eq2; eq2.alpha = r.point(); eq2.z = r.scalar();
  if (inst_evals_in) memcpy(inst_evals, inst_evals_in, 96);
  else for (int i = 0; i < 3; i++) inst_evals[i] = r.scalar();
  if (r.bad) return false;

  tr.append_protocol_name("Spartan SNARK proof");
  tr.append_protocol_name("R1CS proof");
  // the commitment the proof carries must be the row-wise sum of the two witness commitments
  // (proof_point_mult.rs:75-80; my_lib_verify recombines com_1 + com_2, commit_test.rs:369-375)
  std::vector<CG> combined(L);
  static const bool host_only = getenv("VPIN_VERIFY_HOST_MSM") != nullptr;
  bool on_device = false;
  if (L >= 128 && !host_only) {
    // 2L decompressions, L additions and L compressions: one lane per row on the device (msm_var.hip)
    const int rc = vpin_points_add(c, comm_para, comm_input, L, combined[0].b);
    if (rc == VPIN_EVERIFY) return false;
    on_device = rc == VPIN_OK;  // any other failure is an infrastructure error, not a rejection: recombine on the host
    if (on_device)
      for (size_t i = 0; i < L; i++)
        if (!same(combined[i], comm_vars[i])) return false;
  }
  if (!on_device) {
    for (size_t i = 0; i < L; i++) {
      CG a, b;
      memcpy(a.b, comm_para + 32 * i, 32);
      memcpy(b.b, comm_input + 32 * i, 32);
      Point pa, pb;
      if (!decompress(pa, a) || !decompress(pb, b)) return false;
      combined[i] = compress(pa + pb);
      if (!same(combined[i], comm_vars[i])) return false;
    }
  }
  tr.append_message("poly_commitment", "poly_commitment_begin");
  for (size_t i = 0; i < L; i++) tr.append_point("poly_commitment_share", combined[i].b);
  tr.append_message("poly_commitment", "poly_commitment_end");
  std::vector<Fq> tau = tr.challenge_vector("challenge_tau", nrx);
  const Fq zero = Fq::zero(), one = Fq::one();
  CG claim1 = compress(commit1(zero, zero, sg.gens_1)), post1, post2;
  Deferred later;  // sc1, sc2 and the generator sets outlive it
  if (!zksc_verify(sc1, claim1, nrx, 3, sg.gens_1, sg.gens_4, tr, post1, rx, later)) return false;
  if (!knowledge_verify(pok, sg.gens_1, tr, comm_Cz)) return false;
  if (!product_verify(pp, sg.gens_1, tr, comm_Az, comm_Bz, comm_prod)) return false;
  tr.append_point("comm_Az_claim", comm_Az.b);
  tr.append_point("comm_Bz_claim", comm_Bz.b);
  tr.append_point("comm_Cz_claim", comm_Cz.b);
  tr.append_point("comm_prod_Az_Bz_claims", comm_prod.b);
  Fq taus_bound_rx = one;
  for (int i = 0; i < nrx; i++) taus_bound_rx = taus_bound_rx * (rx[i] * tau[i] + (one - rx[i]) * (one - tau[i]));
  Point pAz, pBz, pCz, pProd;
  if (!decompress(pAz, comm_Az) || !decompress(pBz, comm_Bz) || !decompress(pCz, comm_Cz) || !decompress(pProd, comm_prod)) return false;
  if (!equality_verify(eq1, sg.gens_1, tr, compress((pProd - pCz).mul(taus_bound_rx)), post1)) return false;
  Fq r_A = tr.challenge_scalar("challenege_Az"), r_B = tr.challenge_scalar("challenege_Bz"), r_C = tr.challenge_scalar("challenege_Cz");
  CG claim2 = compress(Point::mul2(r_A, pAz, r_B, pBz) + pCz.mul(r_C));
  if (!zksc_verify(sc2, claim2, nry, 2, sg.gens_1, sg.gens_3, tr, post2, ry, later)) return false;
  {
    VSpan vs(" zk sum-check (equations)");
    if (!later.run()) return false;
  }
  if (!polyeval_verify(c, pe, sg.pc, tr, ry.data() + 1, comm_vars_at_ry, comm_vars)) return false;
  // poly_input_eval: SparsePolynomial over [1, inputs...] at ry[1..] (commit_test.rs:457-468)
  const int nvb = (int)log2z(num_vars);
  Fq pie = Fq::zero();
  for (size_t e = 0; e < num_inputs + 1; e++) {
    Fq chi = one;
    for (int j = 0; j < nvb; j++) chi = chi * (((e >> (nvb - j - 1)) & 1) ? ry[1 + j] : one - ry[1 + j]);
    pie = pie + chi * (e == 0 ? one : inputs[e - 1]);
  }
  Point pv;
  if (!decompress(pv, comm_vars_at_ry)) return false;
  Point cz = pv.mul(one - ry[0]) + commit1(pie, zero, sg.pc.gens_1).mul(ry[0]);
  Fq comb = r_A * inst_evals[0] + r_B * inst_evals[1] + r_C * inst_evals[2];
  if (!equality_verify(eq2, sg.gens_1, tr, compress(cz.mul(comb)), post2)) return false;
  tr.append_scalar("Ar_claim", inst_evals[0]);
  tr.append_scalar("Br_claim", inst_evals[1]);
  tr.append_scalar("Cr_claim", inst_evals[2]);
  return true;
}

// ---- SPARK -----------------------------------------------------------------------------------------

static void append_unipoly(Transcript& tr, const Fq* coeffs, int n) {
  tr.append_message("poly", "UniPoly_begin");
  for (int i = 0; i < n; i++) tr.append_scalar("coeff", coeffs[i]);
  tr.append_message("poly", "UniPoly_end");
}

struct BatchedP {
  int num_layers = 0, npc = 0, ndotp = 0;
  std::vector<std::vector<Fq>> polys, cl, cr;
  std::vector<Fq> dotp[3];
};

static bool read_batched(Reader& r, BatchedP& b, int num_layers, int npc, int ndotp) {
  if (r.u64() != (uint64_t)num_layers) return false;
  b.num_layers = num_layers; b.npc = npc; b.ndotp = ndotp;
  b.polys.resize(num_layers); b.cl.resize(num_layers); b.cr.resize(num_layers);
  for (int l = 0; l < num_layers; l++) {
    if (r.u64() != (uint64_t)l) return false;  // layer l from the top has l rounds
    b.polys[l].resize(3 * (size_t)l);
    for (int j = 0; j < l; j++) if (!r.scalars(&b.polys[l][3 * j], 3)) return false;  // degree bound 3
    b.cl[l].resize(npc); b.cr[l].resize(npc);
    if (!r.scalars(b.cl[l].data(), npc) || !r.scalars(b.cr[l].data(), npc)) return false;
  }
  for (int k = 0; k < 3; k++) { b.dotp[k].resize(ndotp); if (!r.scalars(b.dotp[k].data(), ndotp)) return false; }
  return !r.bad;
}

// SumcheckInstanceProof::verify (sumcheck.rs:27-61), degree bound 3
static bool sc_verify(const std::vector<Fq>& polys, int rounds, Fq claim, Transcript& tr, Fq& e_out, std::vector<Fq>& r_out) {
  Fq e = claim;
  r_out.resize(rounds);
  for (int i = 0; i < rounds; i++) {
    const Fq* c = &polys[3 * i];
    Fq cf[4] = {c[0], e - c[0] - c[0] - c[1] - c[2], c[1], c[2]};  // CompressedUniPoly::decompress (unipoly.rs:98-109)
    if (!(cf[0] + (cf[0] + cf[1] + cf[2] + cf[3]) == e)) return false;
    append_unipoly(tr, cf, 4);
    Fq ri = tr.challenge_scalar("challenge_nextround");
    r_out[i] = ri;
    e = unipoly_eval(cf, 4, ri);
  }
  e_out = e;
  return true;
}

// ProductCircuitEvalProofBatched::verify (product_tree.rs:387-485)
static bool batched_verify(const BatchedP& b, const Fq* claims_prod, const Fq* claims_dotp, Transcript& tr, std::vector<Fq>& claims_out,
                           std::vector<Fq>& dotp_out, std::vector<Fq>& rand) {
  const int npc = b.npc, ndotp = b.ndotp, nl = b.num_layers;
  std::vector<Fq> claims(claims_prod, claims_prod + npc), rprod;
  rand.clear();
  const Fq one = Fq::one();
  for (int i = 0; i < nl; i++) {
    if (i == nl - 1) claims.insert(claims.end(), claims_dotp, claims_dotp + ndotp);
    std::vector<Fq> coeffs = tr.challenge_vector("rand_coeffs_next_layer", claims.size());
    Fq claim = Fq::zero();
    for (size_t k = 0; k < claims.size(); k++) claim = claim + claims[k] * coeffs[k];
    Fq claim_last;
    if (!sc_verify(b.polys[i], i, claim, tr, claim_last, rprod)) return false;
    const std::vector<Fq>&cl = b.cl[i], &cr = b.cr[i];
    for (int k = 0; k < npc; k++) { tr.append_scalar("claim_prod_left", cl[k]); tr.append_scalar("claim_prod_right", cr[k]); }
    if ((int)rand.size() != i) return false;
    Fq eq = one;
    for (int k = 0; k < i; k++) eq = eq * (rand[k] * rprod[k] + (one - rand[k]) * (one - rprod[k]));
    Fq expected = Fq::zero();
    for (int k = 0; k < npc; k++) expected = expected + coeffs[k] * (cl[k] * cr[k] * eq);
    if (i == nl - 1)
      for (int k = 0; k < ndotp; k++) {
        tr.append_scalar("claim_dotp_left", b.dotp[0][k]);
        tr.append_scalar("claim_dotp_right", b.dotp[1][k]);
        tr.append_scalar("claim_dotp_weight", b.dotp[2][k]);
        expected = expected + coeffs[npc + k] * b.dotp[0][k] * b.dotp[1][k] * b.dotp[2][k];
      }
    if (!(expected == claim_last)) return false;
    Fq r_layer = tr.challenge_scalar("challenge_r_layer");
    claims.assign(npc, Fq::zero());
    for (int k = 0; k < npc; k++) claims[k] = cl[k] + r_layer * (cr[k] - cl[k]);
    if (i == nl - 1) {
      dotp_out.assign(3 * (ndotp / 2), Fq::zero());
      for (int k = 0; k < ndotp / 2; k++)
        for (int t = 0; t < 3; t++) {
          const std::vector<Fq>& v = b.dotp[t];
          dotp_out[3 * k + t] = v[2 * k] + r_layer * (v[2 * k + 1] - v[2 * k]);
        }
    }
    std::vector<Fq> ext;
    ext.push_back(r_layer);
    ext.insert(ext.end(), rprod.begin(), rprod.end());
    rand.swap(ext);
  }
  claims_out = claims;
  return true;
}

static Fq combine_bot(std::vector<Fq> e, const std::vector<Fq>& ch) {
  size_t n = e.size();
  for (size_t ii = ch.size(); ii-- > 0;) {
    n /= 2;
    for (size_t i = 0; i < n; i++) e[i] = e[2 * i] + ch[ii] * (e[2 * i + 1] - e[2 * i]);
  }
  return e[0];
}

// HashLayerProof::verify_helper (sparse_mlpoly.rs:851-900)
static bool hash_helper(const std::vector<Fq>& rand_mem, const Fq& claim_init, const Fq* claim_read, const Fq* claim_write,
                        const Fq& claim_audit, const Fq* ops_val, const Fq* ops_addr, const Fq* read_ts, const Fq& audit_ts,
                        const std::vector<Fq>& r, const Fq& r_hash, const Fq& gamma) {
  const Fq one = Fq::one(), r2 = r_hash * r_hash;
  const size_t n = rand_mem.size();
  Fq addr = Fq::zero(), val = one;  // IdentityPolynomial / EqPolynomial evaluations (dense_mlpoly.rs:121-127,58-66)
  for (size_t i = 0; i < n; i++) {
    addr = addr + Fq::from_u64((uint64_t)1 << (n - i - 1)) * rand_mem[i];
    val = val * (r[i] * rand_mem[i] + (one - r[i]) * (one - rand_mem[i]));
  }
  if (!(val * r_hash + addr - gamma == claim_init)) return false;
  for (int i = 0; i < 3; i++) {
    if (!(read_ts[i] * r2 + ops_val[i] * r_hash + ops_addr[i] - gamma == claim_read[i])) return false;
    if (!((read_ts[i] + one) * r2 + ops_val[i] * r_hash + ops_addr[i] - gamma == claim_write[i])) return false;
  }
  return audit_ts * r2 + val * r_hash + addr - gamma == claim_audit;
}

// SparseMatPolyEvalProof::verify (sparse_mlpoly.rs:1535-1571) and everything below it
static bool spark_verify(vpin_ctx* c, Reader& r, size_t nx, size_t ny, size_t N, size_t M, const std::vector<CG>& c_ops,
                         const std::vector<CG>& c_mem, const std::vector<Fq>& rx, const std::vector<Fq>& ry, const Fq evals[3],
                         Transcript& tr, int* err) {
  const size_t lgN = log2z(N), lgM = log2z(M), nm = std::max(nx, ny);
  if (((size_t)1 << nm) != M) return false;
  const size_t v_ops = lgN + 4, v_mem = nm + 1, v_derefs = lgN + 3, vmax = std::max(v_ops, v_mem);
  const PcGens *p_ops = nullptr, *p_mem = nullptr, *p_derefs = nullptr;
  // the longest stream first, so the shared device table is built once
  if ((*err = eval_view_v(c, vmax, &p_ops)) || (*err = eval_view_v(c, v_ops, &p_ops)) || (*err = eval_view_v(c, v_mem, &p_mem)) ||
      (*err = eval_view_v(c, v_derefs, &p_derefs)))
    return false;
  const PcGens &g_ops = *p_ops, &g_mem = *p_mem, &g_derefs = *p_derefs;
  if (c_ops.size() != g_ops.L || c_mem.size() != g_mem.L) return false;

  // parse R1CSEvalProof
  std::vector<CG> c_derefs;
  if (!r.points(c_derefs, g_derefs.L)) return false;
  Fq pl[2][8], dotp_left[3], dotp_right[3];
  for (int s = 0; s < 2; s++) {
    pl[s][0] = r.scalar();
    if (!r.scalars(&pl[s][1], 3) || !r.scalars(&pl[s][4], 3)) return false;
    pl[s][7] = r.scalar();
  }
  if (!r.scalars(dotp_left, 3) || !r.scalars(dotp_right, 3)) return false;
  BatchedP pf_mem, pf_ops;
  if (!read_batched(r, pf_mem, (int)lgM, 4, 0) || !read_batched(r, pf_ops, (int)lgN, 12, 6)) return false;
  Fq row_addr[3], row_ts[3], row_audit, col_addr[3], col_ts[3], col_audit, vals[3], d_row[3], d_col[3];
  if (!r.scalars(row_addr, 3) || !r.scalars(row_ts, 3)) return false;
  row_audit = r.scalar();
  if (!r.scalars(col_addr, 3) || !r.scalars(col_ts, 3)) return false;
  col_audit = r.scalar();
  if (!r.scalars(vals, 3) || !r.scalars(d_row, 3) || !r.scalars(d_col, 3)) return false;
  DpLogP pe_ops, pe_mem, pe_derefs;
  if (!read_dplog(r, pe_ops, log2z(g_ops.R)) || !read_dplog(r, pe_mem, log2z(g_mem.R)) || !read_dplog(r, pe_derefs, log2z(g_derefs.R)))
    return false;
  if (r.bad || r.pos != r.len) return false;

  tr.append_protocol_name("Sparse polynomial evaluation proof");
  std::vector<Fq> rx_ext(nm, Fq::zero()), ry_ext(nm, Fq::zero());
  std::copy(rx.begin(), rx.end(), rx_ext.begin() + (nm - nx));
  std::copy(ry.begin(), ry.end(), ry_ext.begin() + (nm - ny));
  tr.append_message("derefs_commitment", "begin_derefs_commitment");
  tr.append_message("comm_poly_row_col_ops_val", "poly_commitment_begin");
  for (auto& p : c_derefs) tr.append_point("poly_commitment_share", p.b);
  tr.append_message("comm_poly_row_col_ops_val", "poly_commitment_end");
  tr.append_message("derefs_commitment", "end_derefs_commitment");
  std::vector<Fq> r_mem_check = tr.challenge_vector("challenge_r_hash", 2);
  tr.append_protocol_name("Sparse polynomial evaluation proof");     // PolyEvalNetworkProof::verify (:1372-1434)
  tr.append_protocol_name("Sparse polynomial product layer proof");  // ProductLayerProof::verify (:1229-1322)
  static const char* lab[2][4] = {{"claim_row_eval_init", "claim_row_eval_read", "claim_row_eval_write", "claim_row_eval_audit"},
                                  {"claim_col_eval_init", "claim_col_eval_read", "claim_col_eval_write", "claim_col_eval_audit"}};
  for (int s = 0; s < 2; s++) {
    Fq ws = Fq::one(), rs = Fq::one();
    for (int m = 0; m < 3; m++) { rs = rs * pl[s][1 + m]; ws = ws * pl[s][4 + m]; }
    if (!(pl[s][0] * ws == rs * pl[s][7])) return false;
    tr.append_scalar(lab[s][0], pl[s][0]);
    tr.append_scalars(lab[s][1], &pl[s][1], 3);
    tr.append_scalars(lab[s][2], &pl[s][4], 3);
    tr.append_scalar(lab[s][3], pl[s][7]);
  }
  Fq claims_dotp_circuit[6], claims_prod_circuit[12];
  for (int m = 0; m < 3; m++) {
    if (!(dotp_left[m] + dotp_right[m] == evals[m])) return false;
    tr.append_scalar("claim_eval_dotp_left", dotp_left[m]);
    tr.append_scalar("claim_eval_dotp_right", dotp_right[m]);
    claims_dotp_circuit[2 * m] = dotp_left[m];
    claims_dotp_circuit[2 * m + 1] = dotp_right[m];
  }
  for (int k = 0; k < 6; k++) { claims_prod_circuit[k] = pl[0][1 + k]; claims_prod_circuit[6 + k] = pl[1][1 + k]; }
  std::vector<Fq> claims_ops, claims_dotp, rand_ops, claims_mem, none, rand_mem;
  Fq mem_in[4] = {pl[0][0], pl[0][7], pl[1][0], pl[1][7]};
  {
    VSpan vs(" product layer");
    if (!batched_verify(pf_ops, claims_prod_circuit, claims_dotp_circuit, tr, claims_ops, claims_dotp, rand_ops)) return false;
    if (!batched_verify(pf_mem, mem_in, nullptr, tr, claims_mem, none, rand_mem)) return false;
  }
  if (claims_dotp.size() != 9 || rand_ops.size() != lgN || rand_mem.size() != lgM) return false;

  tr.append_protocol_name("Sparse polynomial hash layer proof");  // HashLayerProof::verify (:902-1032)
  {
    tr.append_protocol_name("Derefs evaluation proof");
    std::vector<Fq> e8(8, Fq::zero());
    for (int m = 0; m < 3; m++) { e8[m] = d_row[m]; e8[3 + m] = d_col[m]; }
    tr.append_scalars("evals_ops_val", e8.data(), 8);
    std::vector<Fq> ch = tr.challenge_vector("challenge_combine_n_to_one", 3);
    Fq joint = combine_bot(e8, ch);
    std::vector<Fq> rj(ch);
    rj.insert(rj.end(), rand_ops.begin(), rand_ops.end());
    tr.append_scalar("joint_claim_eval", joint);
    if (!polyeval_verify_plain(c, pe_derefs, g_derefs, tr, rj.data(), joint, c_derefs)) return false;
  }
  for (int m = 0; m < 3; m++)
    if (!(claims_dotp[3 * m] == d_row[m]) || !(claims_dotp[3 * m + 1] == d_col[m]) || !(claims_dotp[3 * m + 2] == vals[m])) return false;
  {
    std::vector<Fq> e16(16, Fq::zero());
    for (int m = 0; m < 3; m++) { e16[m] = row_addr[m]; e16[3 + m] = row_ts[m]; e16[6 + m] = col_addr[m]; e16[9 + m] = col_ts[m]; e16[12 + m] = vals[m]; }
    tr.append_scalars("claim_evals_ops", e16.data(), 16);
    std::vector<Fq> ch = tr.challenge_vector("challenge_combine_n_to_one", 4);
    Fq joint = combine_bot(e16, ch);
    std::vector<Fq> rj(ch);
    rj.insert(rj.end(), rand_ops.begin(), rand_ops.end());
    tr.append_scalar("joint_claim_eval_ops", joint);
    if (!polyeval_verify_plain(c, pe_ops, g_ops, tr, rj.data(), joint, c_ops)) return false;
  }
  {
    std::vector<Fq> e2 = {row_audit, col_audit};
    tr.append_scalars("claim_evals_mem", e2.data(), 2);
    std::vector<Fq> ch = tr.challenge_vector("challenge_combine_two_to_one", 1);
    Fq joint = combine_bot(e2, ch);
    std::vector<Fq> rj(ch);
    rj.insert(rj.end(), rand_mem.begin(), rand_mem.end());
    tr.append_scalar("joint_claim_eval_mem", joint);
    if (!polyeval_verify_plain(c, pe_mem, g_mem, tr, rj.data(), joint, c_mem)) return false;
  }
  // claims_ops = row read(3) row write(3) col read(3) col write(3); claims_mem = row init, row audit, col init, col audit
  if (!hash_helper(rand_mem, claims_mem[0], &claims_ops[0], &claims_ops[3], claims_mem[1], d_row, row_addr, row_ts, row_audit, rx_ext,
                   r_mem_check[0], r_mem_check[1]))
    return false;
  return hash_helper(rand_mem, claims_mem[2], &claims_ops[6], &claims_ops[9], claims_mem[3], d_col, col_addr, col_ts, col_audit, ry_ext,
                     r_mem_check[0], r_mem_check[1]);
}

}  // namespace

extern "C" {

// my_r1csproof_verify with the claimed inst_evals (what a caller that keeps SPARK elsewhere checks)
int vpin_sat_verify(vpin_ctx* c, const uint8_t* proof, size_t proof_len, size_t num_cons, size_t num_vars, const uint8_t* inputs,
                    size_t num_inputs, const uint8_t inst_evals[96], const uint8_t* comm_para, const uint8_t* comm_input) {
  if (!c || !proof || !inst_evals || !comm_para || !comm_input || (num_inputs && !inputs)) return VPIN_EINVAL;
  if (!vpin::is_pow2(num_cons) || !vpin::is_pow2(num_vars) || num_inputs >= num_vars) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  Reader r(proof, proof_len);
  Transcript tr("snark_example");
  std::vector<Fq> rx, ry;
  Fq ie[3];
  int err = 0;
  bool ok = sat_verify(c, r, num_cons, num_vars, reinterpret_cast<const Fq*>(inputs), num_inputs, reinterpret_cast<const Fq*>(inst_evals),
                       ie, comm_para, comm_input, tr, rx, ry, &err);
  if (err) return err;
  return ok && !r.bad && r.pos == r.len ? VPIN_OK : VPIN_EVERIFY;
}

// my_lib_verify (commit_test.rs:498-548): comm = bincode(R1CSCommitment) from vpin_spark_encode
int vpin_snark_verify(vpin_ctx* c, const uint8_t* proof, size_t proof_len, const uint8_t* comm, size_t comm_len, const uint8_t* inputs,
                      size_t num_inputs, const uint8_t* comm_para, const uint8_t* comm_input) {
  if (!c || !proof || !comm || !comm_para || !comm_input || (num_inputs && !inputs)) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  Reader rc(comm, comm_len);
  const size_t num_cons = rc.u64(), num_vars = rc.u64(), n_in = rc.u64(), batch = rc.u64(), N = rc.u64(), M = rc.u64();
  if (rc.bad || batch != 3 || n_in != num_inputs || !vpin::is_pow2(num_cons) || !vpin::is_pow2(num_vars) || !vpin::is_pow2(N) ||
      !vpin::is_pow2(M) || N < 4 || M < 4 || N > ((size_t)1 << 40) || num_inputs >= num_vars)
    return VPIN_EVERIFY;
  const size_t nx = log2z(num_cons), ny = log2z(2 * num_vars);
  std::vector<CG> c_ops, c_mem;
  const size_t L_ops = (size_t)1 << ((log2z(N) + 4) / 2), L_mem = (size_t)1 << ((std::max(nx, ny) + 1) / 2);
  if (!rc.points(c_ops, L_ops) || !rc.points(c_mem, L_mem) || rc.bad || rc.pos != rc.len) return VPIN_EVERIFY;
  Reader r(proof, proof_len);
  Transcript tr("snark_example");
  std::vector<Fq> rx, ry;
  Fq ie[3];
  int err = 0;
  bool ok;
  {
    VSpan vs("sat_verify");
    ok = sat_verify(c, r, num_cons, num_vars, reinterpret_cast<const Fq*>(inputs), num_inputs, nullptr, ie, comm_para, comm_input, tr, rx, ry,
                    &err);
  }
  if (err) return err;
  if (!ok) return VPIN_EVERIFY;
  VSpan vs("spark_verify");
  ok = spark_verify(c, r, nx, ny, N, M, c_ops, c_mem, rx, ry, ie, tr, &err);
  if (err) return err;
  return ok ? VPIN_OK : VPIN_EVERIFY;
}

}  // extern "C"
