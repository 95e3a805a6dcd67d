// prover_common.h -- host-side protocol pieces shared by prover.cpp (R1CSProof) and spark.cpp
// (R1CSEvalProof): bincode writer, Pedersen commitments over fixed-base tables, the Sigma
// protocols, UniPoly helpers and DotProductProofLog over the device generator table.
// Internal to the library (everything `static`, one copy per translation unit).
#pragma once
#include <omp.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../ctx.h"
#include "../comm.h"
#include "curve.h"
#include "transcript.h"

extern "C" {
size_t vpin_gens_msm_parts_count(size_t ncols);
int vpin_gens_msm_parts(vpin_ctx* ctx, const vpin_gens* g, const uint8_t* scalars_mont, size_t rows, size_t ncols,
                        uint8_t* parts_xyzt);
int vpin_poly_bound(vpin_ctx* c, const vpin_table* Z, const uint8_t* Lvec, size_t L_size, uint8_t* out_LZ);
int vpin_gens_map_stream(vpin_ctx* ctx, const uint8_t* stream64, size_t nb, uint8_t* out_xyzt);
void vpin_r1cs_dims(const vpin_r1cs_dev* d, size_t* num_cons, size_t* num_vars, size_t* num_inputs);
}

namespace vpin_prover {

#if defined(__clang__)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wunused-function"
#endif


using namespace vpin_host;
using Clock = std::chrono::steady_clock;
static double secs(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

// Host worker threads for the few batched host loops (generator derivation, per-round blind
// commitments).  OpenMP workers that spin after a parallel region burn a container's CPU quota and
// get the whole process throttled, so waiting is made passive and the team is kept small.
// VPIN_HOST_THREADS = n fixes the team size.  Unset (round 6): at most 8 and at most the process's CPU quota divided by twice
// the contexts that exist (every context's proving thread plus its team may be busy at once, and the runtime's own threads
// need room): 16 CPUs / (2 x 8 rank contexts) = 1, / (2 x 4 lanes) = 2, a single context 8.
static int host_threads() {
  static const int fixed = [] {
    setenv("KMP_BLOCKTIME", "0", 0);
    setenv("OMP_WAIT_POLICY", "PASSIVE", 0);
    const char* e = getenv("VPIN_HOST_THREADS");
    int want = e ? atoi(e) : 0;
    const int hw = omp_get_num_procs();
    return want > hw ? hw : want;
  }();
  if (fixed >= 1) return fixed;
  const int ctxs = vpin::live_ctx_count();
  int n = (int)(vpin::host_cpu_quota() / (2.0 * (double)(ctxs > 0 ? ctxs : 1)));
  return n < 1 ? 1 : n > 8 ? 8 : n;
}

static size_t log2z(size_t n) { size_t l = 0; while (((size_t)1 << l) < n) l++; return l; }
static const uint8_t* B(const Fq* p) { return reinterpret_cast<const uint8_t*>(p); }
static uint8_t* B(Fq* p) { return reinterpret_cast<uint8_t*>(p); }

struct CG { uint8_t b[32]; };
static CG compress(const Point& p) { CG c; p.compress(c.b); return c; }

// ---- generators -------------------------------------------------------------------------

struct Mcg {  // MultiCommitGens view over fixed-base tables
  int n;
  const FixedBase* G[4];
  const FixedBase* h;
};

// MultiCommitGens::new (commitments.rs:20-38) stream under `label`.  Every set of a label is a prefix of one SHAKE stream, so
// the longest prefix derived so far is kept per process: the prover's and the verifier's generator sets of one CLI run (and
// every context of a service) hash each point to the group once.
inline void derive_gens(std::vector<Point>& g, size_t nb, const char* label, vpin_ctx* dev = nullptr) {
  static std::mutex mu;
  static std::map<std::string, std::shared_ptr<const std::vector<Point>>> cache;
  std::shared_ptr<const std::vector<Point>> have;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(label);
    if (it != cache.end()) have = it->second;
  }
  if (have && have->size() >= nb) {
    g.assign(have->begin(), have->begin() + (long)nb);
    return;
  }
  uint8_t bc[32];
  Point::basepoint().compress(bc);  // GROUP_BASEPOINT_COMPRESSED (group.rs:26-27)
  Shake256 sh;
  sh.absorb(reinterpret_cast<const uint8_t*>(label), strlen(label));
  sh.absorb(bc, 32);
  sh.finalize();
  std::vector<uint8_t> stream(64 * nb);
  sh.squeeze(stream.data(), stream.size());
  auto fresh = std::make_shared<std::vector<Point>>(nb);
  const size_t n0 = have ? have->size() : 0;
  for (size_t i = 0; i < n0; i++) (*fresh)[i] = (*have)[i];
  // the map to the group (two exponentiations per point): on the device when the caller has one and the set is large
  bool mapped = false;
  if (dev && nb - n0 >= 256) {
    std::vector<uint8_t> xyzt(128 * (nb - n0));
    if (vpin_gens_map_stream(dev, stream.data() + 64 * n0, nb - n0, xyzt.data()) == VPIN_OK) {
      for (size_t i = n0; i < nb; i++) (*fresh)[i] = Point::from_xyzt(xyzt.data() + 128 * (i - n0));
      mapped = true;
    }
  }
  if (!mapped) {
#pragma omp parallel for schedule(static) num_threads(host_threads())
    for (long i = (long)n0; i < (long)nb; i++) (*fresh)[i] = Point::from_uniform_bytes(stream.data() + 64 * i);
  }
  g = *fresh;
  std::lock_guard<std::mutex> lock(mu);
  auto& slot = cache[label];
  if (!slot || slot->size() < nb) slot = fresh;
}

// Commitments for Scalar / [Scalar] over at most 4 generators (commitments.rs:85-98)
static Point commit(const Fq* v, int n, const Fq& blind, const Mcg& g) {
  Point acc = Point::identity();
  for (int i = 0; i < n; i++) g.G[i]->mul_acc(acc, v[i]);
  g.h->mul_acc(acc, blind);
  return acc;
}
static Point commit1(const Fq& x, const Fq& blind, const Mcg& g) { return commit(&x, 1, blind, g); }

// few-row fixed-base MSM: GPU partial points, summed on the host (rows <= 2 here)
static int msm_rows_host_sum(vpin_ctx* c, const vpin_gens* dev, const Fq* scalars, size_t rows, size_t ncols, Point* out) {
  const size_t np = vpin_gens_msm_parts_count(ncols);
  std::vector<uint8_t> parts(rows * np * 128);
  int rc = vpin_gens_msm_parts(c, dev, B(scalars), rows, ncols, parts.data());
  if (rc) return rc;
  for (size_t r = 0; r < rows; r++) {
    Point acc = Point::from_xyzt(parts.data() + (r * np) * 128);
    for (size_t k = 1; k < np; k++) acc = acc + Point::from_xyzt(parts.data() + (r * np + k) * 128);
    out[r] = acc;
  }
  return VPIN_OK;
}

// ---- bincode writer ------------------------------------------------------------------------

struct Writer {
  std::vector<uint8_t> buf;
  void bytes(const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; buf.insert(buf.end(), b, b + n); }
  void u64(uint64_t v) { bytes(&v, 8); }
  void scalar(const Fq& s) { bytes(s.l, 32); }  // Montgomery limbs, as derive(Serialize) on Scalar([u64;4])
  void point(const CG& c) { bytes(c.b, 32); }
};

// ---- sigma protocols -----------------------------------------------------------------------

struct DotProof { CG delta, beta; std::vector<Fq> z; Fq z_delta, z_beta; };

// DotProductProof::prove (nizk/mod.rs:315-374).  Cx is known to the caller (it is the round's
// comm_poly), so it is passed in instead of being recomputed.
[[maybe_unused]] static void dotproduct_prove(DotProof& pf, const Mcg& g1, const Mcg& gn, Transcript& tr, Transcript& tape, const Fq* x,
                             const Fq& blind_x, const Fq* a, const Fq& y, const Fq& blind_y, int n, const CG& Cx) {
  tr.append_protocol_name("dot product proof");
  std::vector<Fq> d = tape.challenge_vector("d_vec", n);
  Fq r_delta = tape.challenge_scalar("r_delta"), r_beta = tape.challenge_scalar("r_beta");
  (void)blind_x;
  tr.append_point("Cx", Cx.b);
  CG Cy = compress(commit1(y, blind_y, g1));
  tr.append_point("Cy", Cy.b);
  tr.append_scalars("a", a, n);
  pf.delta = compress(commit(d.data(), n, r_delta, gn));
  tr.append_point("delta", pf.delta.b);
  Fq ad = Fq::zero();
  for (int i = 0; i < n; i++) ad = ad + a[i] * d[i];
  pf.beta = compress(commit1(ad, r_beta, g1));
  tr.append_point("beta", pf.beta.b);
  Fq c = tr.challenge_scalar("c");
  pf.z.resize(n);
  for (int i = 0; i < n; i++) pf.z[i] = c * x[i] + d[i];
  pf.z_delta = c * blind_x + r_delta;
  pf.z_beta = c * blind_y + r_beta;
}

struct KnowProof { CG alpha; Fq z1, z2; };
static CG knowledge_prove(KnowProof& pf, const Mcg& g, Transcript& tr, Transcript& tape, const Fq& x, const Fq& r) {
  tr.append_protocol_name("knowledge proof");
  Fq t1 = tape.challenge_scalar("t1"), t2 = tape.challenge_scalar("t2");
  CG C = compress(commit1(x, r, g));
  tr.append_point("C", C.b);
  pf.alpha = compress(commit1(t1, t2, g));
  tr.append_point("alpha", pf.alpha.b);
  Fq c = tr.challenge_scalar("c");
  pf.z1 = x * c + t1;
  pf.z2 = r * c + t2;
  return C;
}

struct EqProof { CG alpha; Fq z; };
static void equality_prove(EqProof& pf, const Mcg& g, Transcript& tr, Transcript& tape, const Fq& v1, const Fq& s1,
                           const Fq& v2, const Fq& s2) {
  tr.append_protocol_name("equality proof");
  Fq r = tape.challenge_scalar("r");
  CG C1 = compress(commit1(v1, s1, g));
  tr.append_point("C1", C1.b);
  CG C2 = compress(commit1(v2, s2, g));
  tr.append_point("C2", C2.b);
  pf.alpha = compress(g.h->mul(r));
  tr.append_point("alpha", pf.alpha.b);
  Fq c = tr.challenge_scalar("c");
  pf.z = c * (s1 - s2) + r;
}

struct ProdProof { CG alpha, beta, delta; Fq z[5]; };
static void product_prove(ProdProof& pf, const Mcg& g, Transcript& tr, Transcript& tape, const Fq& x, const Fq& rX,
                          const Fq& y, const Fq& rY, const Fq& z, const Fq& rZ, CG& X, CG& Y, CG& Z) {
  tr.append_protocol_name("product proof");
  Fq b1 = tape.challenge_scalar("b1"), b2 = tape.challenge_scalar("b2"), b3 = tape.challenge_scalar("b3"),
     b4 = tape.challenge_scalar("b4"), b5 = tape.challenge_scalar("b5");
  Point Xp = commit1(x, rX, g);
  X = compress(Xp); tr.append_point("X", X.b);
  Y = compress(commit1(y, rY, g)); tr.append_point("Y", Y.b);
  Z = compress(commit1(z, rZ, g)); tr.append_point("Z", Z.b);
  pf.alpha = compress(commit1(b1, b2, g)); tr.append_point("alpha", pf.alpha.b);
  pf.beta = compress(commit1(b3, b4, g)); tr.append_point("beta", pf.beta.b);
  // delta = b3 * X + b5 * h  (gens_X = {G: [X], h}); X re-derived from its encoding as the reference does
  Point Xd;
  Point::decompress(Xd, X.b);
  Point dl = Xd.mul(b3);
  g.h->mul_acc(dl, b5);
  pf.delta = compress(dl); tr.append_point("delta", pf.delta.b);
  Fq c = tr.challenge_scalar("c");
  pf.z[0] = b1 + c * x;
  pf.z[1] = b2 + c * rX;
  pf.z[2] = b3 + c * y;
  pf.z[3] = b4 + c * rY;
  pf.z[4] = b5 + c * (rZ - rX * y);
}

// UniPoly::from_evals / evaluate (unipoly.rs:23-54,72-80)
static void unipoly_from_evals(const Fq* e, int n, Fq* coeffs) {
  static const Fq two_inv = Fq::from_u64(2).invert(), six_inv = Fq::from_u64(6).invert();
  if (n == 3) {
    Fq c = e[0];
    Fq a = two_inv * (e[2] - e[1] - e[1] + c);
    Fq b = e[1] - c - a;
    coeffs[0] = c; coeffs[1] = b; coeffs[2] = a;
  } else {
    Fq d = e[0];
    Fq a = six_inv * (e[3] - e[2] - e[2] - e[2] + e[1] + e[1] + e[1] - e[0]);
    Fq b = two_inv * (e[0] + e[0] - e[1] - e[1] - e[1] - e[1] - e[1] + e[2] + e[2] + e[2] + e[2] - e[3]);
    Fq c = e[1] - d - a - b;
    coeffs[0] = d; coeffs[1] = c; coeffs[2] = b; coeffs[3] = a;
  }
}
static Fq unipoly_eval(const Fq* coeffs, int n, const Fq& r) {
  Fq eval = coeffs[0], power = r;
  for (int i = 1; i < n; i++) { eval = eval + power * coeffs[i]; power = power * r; }
  return eval;
}

// EqPolynomial::evals (dense_mlpoly.rs:78-94) on the host, for the short L / R vectors
static void host_eq(const Fq* r, size_t ell, Fq* out) {
  out[0] = Fq::one();
  size_t size = 1;
  for (size_t j = 0; j < ell; j++) {
    for (size_t i = size; i-- > 0;) {
      Fq s = out[i];
      out[2 * i + 1] = s * r[j];
      out[2 * i] = s - out[2 * i + 1];
    }
    size *= 2;
  }
}

// PolyCommitmentGens (dense_mlpoly.rs:20-33) as a view on one label's generator stream:
// gens_n = g[0..R), gens_1.G = g[R], h = g[R+1] (nizk/mod.rs:411-425); `dev` is the window table
// over a prefix of at least R+2 generators of that stream.
struct PcGens {
  size_t ell = 0, L = 0, R = 0;
  const vpin_gens* dev = nullptr;
  FixedBase fb_gR, fb_h;
  Mcg gens_1;
  void bind_views() { gens_1 = Mcg{1, {&fb_gR, nullptr, nullptr, nullptr}, &fb_h}; }
};

struct DpLog { CG Cy; std::vector<CG> Lvec, Rvec; CG delta, beta; Fq z1, z2; };

static void write_dplog(Writer& w, const DpLog& p) {
  w.u64(p.Lvec.size()); for (auto& q : p.Lvec) w.point(q);
  w.u64(p.Rvec.size()); for (auto& q : p.Rvec) w.point(q);
  w.point(p.delta); w.point(p.beta); w.scalar(p.z1); w.scalar(p.z2);
}

// DotProductProofLog::prove (nizk/mod.rs:447-531) with BulletReductionProof::prove
// (nizk/bullet.rs:32-132) for x = LZ (R entries, blind LZ_blind), a = Rv, y (blind blind_y).
// G is never folded explicitly: the folded generator G_k[i] = sum_{j = i mod n} s_j g_j for known
// coefficients s_j, so every L/R of the reduction (and g_hat) is a fixed-base MSM over the
// original stream -> one GPU call per round.
static int dplog_prove(vpin_ctx* c, const PcGens& pc, Transcript& tr, Transcript& tape, const std::vector<Fq>& LZ,
                       const Fq& LZ_blind, const std::vector<Fq>& Rv, const Fq& y, const Fq& blind_y, DpLog& out) {
  const size_t R = pc.R;
  int rc;
  tr.append_protocol_name("dot product proof (log)");
  Fq d_ = tape.challenge_scalar("d");
  Fq r_delta = tape.challenge_scalar("r_delta");
  Fq r_beta = tape.challenge_scalar("r_delta");  // sic: the reference draws r_beta under the label "r_delta"
  const size_t lgR = log2z(R);
  std::vector<Fq> bv1 = tape.challenge_vector("blinds_vec_1", 2 * lgR);
  std::vector<Fq> bv2 = tape.challenge_vector("blinds_vec_2", 2 * lgR);
  const size_t ncols = R + 2;  // scalars over g[0..R) | g[R] | g[R+1]=h
  std::vector<Fq> srow(2 * ncols, Fq::zero());
  // The vectors a, b and the generator coefficients s_j stay on the device (bullet.hip); per round the GPU
  // returns the partial points of  a_L . G_R  and  a_R . G_L  over the stream generators and the two cross
  // inner products; the c*Q + blind*H terms (Q = r * g[R]) are two fixed-base multiplications each here.
  // Uploaded first (no wait): the copies run while the host hashes the R scalars of `a` into the transcript.
  vpin::BulletState* bs = nullptr;
  if ((rc = vpin::bullet_begin(c, B(LZ.data()), B(Rv.data()), R, &bs))) return rc;
  struct BsGuard {  // an early return may leave the uploads in flight: wait before the buffers go back to the pool
    vpin_ctx* c; vpin::BulletState* s; bool done;
    ~BsGuard() { if (!done) (void)hipStreamSynchronize(c->stream); vpin::bullet_free(c, s); }
  } bs_guard{c, bs, false};
  CG Cx;
  {
    memcpy(srow.data(), LZ.data(), R * 32);
    srow[R] = Fq::zero();
    srow[R + 1] = LZ_blind;
    Point p;
    if ((rc = msm_rows_host_sum(c, pc.dev, srow.data(), 1, ncols, &p))) return rc;
    Cx = compress(p);
  }
  tr.append_point("Cx", Cx.b);
  out.Cy = compress(commit1(y, blind_y, pc.gens_1));
  tr.append_point("Cy", out.Cy.b);
  if (R >= 8192) {
    // (the evaluation proofs of a 2^25-constraint SNARK absorb 74 k scalars here: their conversion out of Montgomery form runs
    // on the context's host team, the hashing stays sequential)
    std::vector<uint8_t> ab(R * 32);
#pragma omp parallel for schedule(static) num_threads(host_threads())
    for (size_t i = 0; i < R; i++) Rv[i].to_bytes(ab.data() + 32 * i);
    tr.append_scalars_bytes("a", ab.data(), R);
  } else {
    tr.append_scalars("a", Rv.data(), R);
  }
  Fq r_ = tr.challenge_scalar("r");
  // gens_1_scaled.G[0] = r * g[R]; every use below multiplies g[R] by r times something
  Fq blind_Gamma = LZ_blind + r_ * blind_y;
  out.Lvec.resize(lgR); out.Rvec.resize(lgR);
  Fq blind_fin = blind_Gamma;
  const size_t np = vpin_gens_msm_parts_count(R);
  std::vector<uint8_t> parts_pageable;
  uint8_t* parts_buf = vpin::bullet_pinned(c);  // pinned: the per-round device-to-host copies complete without staging
  if (!parts_buf || 2 * np * 128 > 32 * 1024) { parts_pageable.resize(2 * np * 128); parts_buf = parts_pageable.data(); }
  struct { uint8_t* p; uint8_t* data() const { return p; } } parts{parts_buf};
  auto sum_parts = [&](const uint8_t* p) {
    Point acc = Point::from_xyzt(p);
    for (size_t k = 1; k < np; k++) acc = acc + Point::from_xyzt(p + k * 128);
    return acc;
  };
  size_t n = R;
  static const bool fine = getenv("VPIN_SPARK_TRACE") && atoi(getenv("VPIN_SPARK_TRACE")) >= 3;
  double tb[6] = {0, 0, 0, 0, 0, 0};
  // Fused rounds: one launch folds the previous challenge in, runs the round's MSM and publishes the inner products; its
  // partial points arrive per side (bullet_part_ptrs: at most 128 for L and R together).
  const bool fused = vpin::bullet_fused(bs);
  auto sum_side = [&](int row, size_t nn) {  // row 0 = L, 1 = R; nn = 0: g_hat
    const uint8_t* ptrs[256];
    const size_t cnt = vpin::bullet_part_ptrs(c, bs, nn, row, ptrs);
    Point acc = Point::identity();
    for (size_t k = 0; k < cnt; k++) acc = acc + Point::from_xyzt(ptrs[k]);
    return acc;
  };
  Fq u_prev = Fq::zero(), u_inv_prev = Fq::zero();
  for (size_t round = 0; round < lgR; round++) {
    n /= 2;
    Fq cLR[2];
    auto t0 = Clock::now();
    if (fused) rc = vpin::bullet_step(c, pc.dev, bs, n, round ? B(&u_prev) : nullptr, round ? B(&u_inv_prev) : nullptr, B(cLR));
    else rc = vpin::bullet_round_begin(c, pc.dev, bs, n, parts.data(), B(cLR));
    if (rc) return rc;
    auto t1 = Clock::now();
    // c_L * Q + blind_L * H and c_R * Q + blind_R * H on the host while the device is still summing the stream terms
    Point qh[2] = {Point::identity(), Point::identity()};
    pc.fb_gR.mul_acc(qh[0], cLR[0] * r_); pc.fb_h.mul_acc(qh[0], bv1[round]);
    pc.fb_gR.mul_acc(qh[1], cLR[1] * r_); pc.fb_h.mul_acc(qh[1], bv2[round]);
    auto t2 = Clock::now();
    if ((rc = vpin::bullet_round_end(c))) return rc;
    auto t3 = Clock::now();
    Point lr[2] = {(fused ? sum_side(0, n) : sum_parts(parts.data())) + qh[0],
                   (fused ? sum_side(1, n) : sum_parts(parts.data() + np * 128)) + qh[1]};
    auto t4 = Clock::now();
    out.Lvec[round] = compress(lr[0]);
    out.Rvec[round] = compress(lr[1]);
    tr.append_point("L", out.Lvec[round].b);
    tr.append_point("R", out.Rvec[round].b);
    Fq u = tr.challenge_scalar("u"), u_inv = u.invert();
    auto t5 = Clock::now();
    if (fused) { u_prev = u; u_inv_prev = u_inv; }  // folded in by the next round's launch
    else if ((rc = vpin::bullet_fold(c, bs, n, B(&u), B(&u_inv)))) return rc;
    blind_fin = blind_fin + bv1[round] * u * u + bv2[round] * u_inv * u_inv;
    if (fine) {
      auto t6 = Clock::now();
      tb[0] += secs(t0, t1); tb[1] += secs(t1, t2); tb[2] += secs(t2, t3); tb[3] += secs(t3, t4); tb[4] += secs(t4, t5); tb[5] += secs(t5, t6);
    }
  }
  if (fine)
    fprintf(stderr, "[dplog] R=%zu, %zu rounds, us per round: rows+inner products %.1f | host Q,H terms %.1f | wait for the MSM %.1f | "
            "sum %zu parts x2 %.1f | compress x2 + transcript + invert %.1f | fold launch %.1f\n", R, lgR, tb[0] / lgR * 1e6,
            tb[1] / lgR * 1e6, tb[2] / lgR * 1e6, np, tb[3] / lgR * 1e6, tb[4] / lgR * 1e6, tb[5] / lgR * 1e6);
  Fq xa[2];
  // g_hat = sum_j s_j g_j
  if (fused) rc = vpin::bullet_finish_fused(c, pc.dev, bs, B(&u_prev), B(&u_inv_prev), B(xa));
  else rc = vpin::bullet_finish(c, pc.dev, bs, B(xa), parts.data());
  if (rc) return rc;
  const Fq x_hat = xa[0], a_hat = xa[1], y_hat = x_hat * a_hat;
  Point g_hat = fused ? sum_side(0, 0) : sum_parts(parts.data());
  {
    Point p = g_hat.mul(d_);  // d.commit(&r_delta, {G:[g_hat], h})
    pc.fb_h.mul_acc(p, r_delta);
    out.delta = compress(p);
    tr.append_point("delta", out.delta.b);
    Point q = pc.fb_gR.mul(d_ * r_);  // d.commit(&r_beta, gens_1_scaled)
    pc.fb_h.mul_acc(q, r_beta);
    out.beta = compress(q);
    tr.append_point("beta", out.beta.b);
  }
  Fq c_ = tr.challenge_scalar("c");
  out.z1 = d_ + c_ * y_hat;
  out.z2 = a_hat * (c_ * blind_fin + r_beta) + r_delta;
  bs_guard.done = true;  // bullet_finish* waited for the stream
  return VPIN_OK;
}

struct TableGuard {
  vpin_ctx* c;
  std::vector<vpin_table*> t;
  explicit TableGuard(vpin_ctx* c_) : c(c_) {}
  ~TableGuard() { for (auto* p : t) vpin_table_free(c, p); }
  vpin_table* add(vpin_table* p) { t.push_back(p); return p; }
};

#if defined(__clang__)
#pragma clang diagnostic pop
#endif

// prover.cpp
int sat_prove_core(vpin_ctx* c, const vpin_r1cs_dev* dinst, size_t nv, size_t ncons, size_t ni,
                   const vpin_table* d_para, const vpin_table* d_input, const vpin_table* d_vars, const uint8_t* inputs,
                   const uint8_t seed_commit64[64], const uint8_t seed_proof64[64], uint8_t* proof_out, size_t proof_cap,
                   size_t* proof_len, uint8_t* comm_para_out, uint8_t* comm_input_out, uint8_t inst_evals_out[96],
                   uint8_t* rx_out, uint8_t* ry_out, vpin_host::Transcript* tr_out, vpin_host::Transcript* tape_out);

}  // namespace vpin_prover
