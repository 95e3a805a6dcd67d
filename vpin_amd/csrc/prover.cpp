// prover.cpp -- host orchestration of vPIN's R1CS satisfiability proof on one MI355X.
//
// C++ counterpart of the reference's Rust glue (the Rust toolchain is absent here, so the
// host side above the kernel ABI is written in C++ and mirrors the reference functions
// name for name):
//   vPIN_proof_generation/src/proof_point_mult.rs:38-94   commit para/input, combine
//   vPIN_proof_generation/src/commit_test.rs:59-133       my_lib_prove   (sat part)
//   vPIN_proof_generation/src/commit_test.rs:136-334      my_R1CSProof_prove
//   Spartan/src/sumcheck.rs:428-776                       ZK sum-checks (round loop)
//   Spartan/src/nizk/mod.rs, nizk/bullet.rs               sigma protocols, log-size dot product
//   Spartan/src/dense_mlpoly.rs:326-379                   PolyEvalProof::prove
// The data-parallel work runs in the HIP kernels (sumcheck.hip, msm.hip, poly.hip); this file
// owns the strictly sequential part: the Merlin transcript, the per-round <=5-term Pedersen
// commitments, and proof serialisation (bincode layout).  It shares no code with the test-side checker.
#include "host/prover_common.h"

namespace {

using namespace vpin_host;
using namespace vpin_prover;

// ---- generators -------------------------------------------------------------------------

struct SatGens {
  size_t ell = 0, L = 0, R = 0, nb = 0;
  std::vector<Point> g;  // generator stream g[0..nb)
  FixedBase fb[5];
  const vpin_gens* dev = nullptr;  // shared per (device, label): vpin_gens_shared
  PcGens pc;                   // gens_pc (r1csproof.rs:76-89): fb_gR, fb_h, device table
  Mcg gens_1, gens_3, gens_4;  // R1CSSumcheckGens (r1csproof.rs:49-74); gens_1 also = gens_pc.gens.gens_1
};

struct ProverCache {
  std::map<size_t, std::unique_ptr<SatGens>> by_nv;
};

static void cache_free(vpin_ctx* c) {
  auto* pc = static_cast<ProverCache*>(c->prover_cache);
  if (!pc) return;
  delete pc;  // the device tables belong to the shared registry
  c->prover_cache = nullptr;
}

// R1CSGens::new(b"gens_r1cs_sat", _, num_vars) (r1csproof.rs:84-89, lib.rs:314)
static int get_gens(vpin_ctx* c, size_t num_vars, SatGens** out) {
  if (!c->prover_cache) { c->prover_cache = new ProverCache(); c->prover_cache_free = cache_free; }
  auto* pc = static_cast<ProverCache*>(c->prover_cache);
  auto it = pc->by_nv.find(num_vars);
  if (it != pc->by_nv.end()) { *out = it->second.get(); return VPIN_OK; }
  std::unique_ptr<SatGens> sg(new SatGens());
  sg->ell = log2z(num_vars);
  size_t left = sg->ell / 2;
  sg->L = (size_t)1 << left;
  sg->R = (size_t)1 << (sg->ell - left);
  sg->nb = sg->R + 2 < 5 ? 5 : sg->R + 2;
  vpin::TraceLap lap(nullptr, "sat gens");
  derive_gens(sg->g, sg->nb, "gens_r1cs_sat", c);
  lap("derive_gens (host)");
#pragma omp parallel for schedule(dynamic, 1) num_threads(host_threads())
  for (int i = 0; i < 7; i++) {
    if (i < 5) sg->fb[i] = FixedBase(sg->g[i]);
    else if (i == 5) sg->pc.fb_gR = FixedBase(sg->g[sg->R]);
    else sg->pc.fb_h = FixedBase(sg->g[sg->R + 1]);
  }
  sg->pc.ell = sg->ell; sg->pc.L = sg->L; sg->pc.R = sg->R;
  sg->pc.bind_views();
  sg->gens_1 = sg->pc.gens_1;
  sg->gens_3 = Mcg{3, {&sg->fb[0], &sg->fb[1], &sg->fb[2], nullptr}, &sg->fb[3]};
  sg->gens_4 = Mcg{4, {&sg->fb[0], &sg->fb[1], &sg->fb[2], &sg->fb[3]}, &sg->fb[4]};
  int rc = vpin_gens_shared(c, "gens_r1cs_sat", nullptr, sg->nb, 0, &sg->dev);
  if (rc == VPIN_EINVAL) {  // no table of this label covers nb generators yet: build one
    std::vector<uint8_t> xyzt(128 * sg->nb);
    for (size_t i = 0; i < sg->nb; i++) sg->g[i].to_xyzt(xyzt.data() + 128 * i);
    c->gens_scalars_per_proof = (double)num_vars;  // the two halves of the assignment, about half of them full-size scalars
    rc = vpin_gens_shared(c, "gens_r1cs_sat", xyzt.data(), sg->nb, 0, &sg->dev);
  }
  if (rc) return rc;
  lap("fixed bases + table");
  sg->pc.dev = sg->dev;
  *out = sg.get();
  pc->by_nv[num_vars] = std::move(sg);
  return VPIN_OK;
}

// ---- ZK sum-check over device tables -------------------------------------------------------

struct ZkSc { std::vector<CG> comm_polys, comm_evals; std::vector<DotProof> proofs; };

static void write_zksc(Writer& w, const ZkSc& p) {
  w.u64(p.comm_polys.size()); for (auto& c : p.comm_polys) w.point(c);
  w.u64(p.comm_evals.size()); for (auto& c : p.comm_evals) w.point(c);
  w.u64(p.proofs.size());
  for (auto& d : p.proofs) {
    w.point(d.delta); w.point(d.beta);
    w.u64(d.z.size()); for (auto& s : d.z) w.scalar(s);
    w.scalar(d.z_delta); w.scalar(d.z_beta);
  }
}

// ZKSumcheckInstanceProof::prove_cubic_with_additive_term (K=4) / prove_quad (K=2)
// (sumcheck.rs:428-776): the per-round evaluation loop and the table folds run on the GPU
// (one fused kernel per round); everything between them is the reference's host sequence.
// Two schedule changes that leave every transcript byte unchanged:
//  * the RandomTape is a private transcript, so the per-round DotProductProof randomness
//    (d_vec, r_delta, r_beta: nizk/mod.rs:333-335) is drawn up front in the reference's order and
//    every commitment term that depends only on it (delta_j, blind*h parts) is computed for all
//    rounds at once across the host cores;
//  * round j+1's kernel is launched as soon as r_j is known and runs while the host finishes
//    round j's dot-product proof.
//
// Phase 1 (K = 4) runs eq-factored when `tau`/`pyramid` are given: tabs = {Az,Bz,Cz}; the eq(tau,.)
// table is never materialised per round.  After binding r_0..r_{j-1} it equals s_j * eq(tau_{j..}, .)
// with s_j = prod_{i<j} eq1(tau_i, r_i), so its value at the round's evaluation point x is
// s_j*((1-tau_j) + x*(2*tau_j-1)) * E_{j+1}[i]; the GPU returns sum_i E_{j+1}[i]*(Az_x Bz_x - Cz_x)[i] and
// the three scale factors are applied here.  Exact field arithmetic: same e0,e2,e3, same proof bytes.
// One sum-check over several GPUs (lw > 0: the world is 2^lw ranks).  The fold pairs (i, i + len/2), so both
// members of a pair are = r (mod 2^lw) while len/2 >= 2^lw: rank r holds the entries = r (mod world) of every table
// (`tabs` are those LOCAL tables, 2^(rounds - lw) entries) and folds them with the unchanged kernels; no table entry ever
// moves.  Per round every rank contributes its partial sums (three scalars) and all ranks derive the same challenge.  The
// eq factor of phase 1 splits the same way: eq(tau_{j+1..}, r + k*world) = eq(tau_{j+1..rounds-lw-1}, k) * eq(tau_lo, r)
// (tau_lo = the last lw challenges), so `pyramid` is the suffix pyramid of the first rounds - lw challenges and the
// rank's sums are scaled by eq(tau_lo, r).  After rounds - lw rounds every local table is down to one entry: the world
// entries of each table are all-gathered and the last lw rounds run on the host (the reference's loops on <= 8 entries).
static int zk_sumcheck(vpin_ctx* c, int K, vpin_table** tabs, const Fq& claim, const Fq& blind_claim, int rounds,
                       const Mcg& g1, const Mcg& gn, Transcript& tr, Transcript& tape, ZkSc& pf, std::vector<Fq>& r_out,
                       Fq* final_claims, Fq& blind_last, const std::vector<Fq>* tau = nullptr,
                       const vpin_table* pyramid = nullptr, int lw = 0) {
  const int nc = (K == 4) ? 4 : 3;
  const bool factored = (K == 4 && tau != nullptr && pyramid != nullptr);
  const int ntab = factored ? 3 : K;
  vpin_comm* cm = lw > 0 ? c->comm : nullptr;
  const int world = cm ? cm->world : 1;
  const int loc_rounds = rounds - lw;  // rounds the device runs on the local tables
  if (lw > 0 && (!cm || (1 << lw) != world || loc_rounds < 1)) return VPIN_EINVAL;
  Fq c_rank = Fq::one();               // eq(tau_lo, rank): scale of this rank's eq-factored sums
  std::vector<Fq> eq_lo;               // eq(tau_lo, .), world entries
  if (cm && factored) {
    eq_lo.resize(world);
    host_eq(tau->data() + loc_rounds, (size_t)lw, eq_lo.data());
    c_rank = eq_lo[cm->rank];
  }
  std::vector<std::vector<Fq>> htab;   // the tables once they are down to `world` entries: htab[t][i]
  Fq s_eq = Fq::one();
  const Fq f_one = Fq::one(), f_two = Fq::from_u64(2), f_three = Fq::from_u64(3), f_five = Fq::from_u64(5);
  std::vector<Fq> blinds_poly = tape.challenge_vector("blinds_poly", rounds);
  std::vector<Fq> blinds_evals = tape.challenge_vector("blinds_evals", rounds);
  struct RoundRand { Fq d[4], r_delta, r_beta; };
  std::vector<RoundRand> rr(rounds);
  for (int j = 0; j < rounds; j++) {
    for (int i = 0; i < nc; i++) rr[j].d[i] = tape.challenge_scalar("d_vec");
    rr[j].r_delta = tape.challenge_scalar("r_delta");
    rr[j].r_beta = tape.challenge_scalar("r_beta");
  }
  std::vector<CG> delta(rounds);
  std::vector<Point> P_bp(rounds), P_be(rounds), P_rb(rounds);
#pragma omp parallel for schedule(dynamic, 1) num_threads(host_threads())
  for (int j = 0; j < rounds; j++) {
    delta[j] = compress(commit(rr[j].d, nc, rr[j].r_delta, gn));
    P_bp[j] = gn.h->mul(blinds_poly[j]);
    P_be[j] = g1.h->mul(blinds_evals[j]);
    P_rb[j] = g1.h->mul(rr[j].r_beta);
  }
  Fq claim_pr = claim;
  CG comm_claim = compress(commit1(claim_pr, blind_claim, g1));
  r_out.clear();
  // Leading-coefficient rounds (sc_dev.h lead_bcd): the kernel returns t(0) and the x^2 coefficient of the
  // quadratic t(x) = sum_i E[i] (Az_x Bz_x - Cz_x)[i]; t(1) follows from the round's claim.  Round 0 returns
  // t(0), t(1) and the coefficient directly; when its claim does not check (a witness that does not satisfy the
  // instance: claim 0 is then not the true sum) or a tau_j is zero, every later round uses the three-sum kernel,
  // so the bytes are those of the reference in that case too.
  bool lead = factored;
  std::vector<Fq> tau_inv;
  if (factored) {
    tau_inv.assign(tau->begin(), tau->begin() + rounds);
    for (auto& x : tau_inv) lead = lead && !x.is_zero();
    if (lead) {
      std::vector<Fq> pre(rounds);
      Fq acc = f_one;
      for (int j = 0; j < rounds; j++) { pre[j] = acc; acc = acc * tau_inv[j]; }
      acc = acc.invert();
      for (int j = rounds - 1; j >= 0; j--) { Fq t = acc * tau_inv[j]; tau_inv[j] = acc * pre[j]; acc = t; }
    }
  }
  Fq cn = claim;  // claim_pr / s_eq: the claim on the quadratic t
  int rc = factored ? vpin::sc_cubic3_launch(c, tabs, pyramid, loc_rounds, 1, nullptr, lead) : vpin::sc_round_launch(c, K, tabs, nullptr);
  if (rc) return rc;
  // host rounds (the last lw): sums over the pairs (i, i + len/2) of the gathered tables, in the kernels' conventions
  auto host_round = [&](int j, bool lead_now, Fq e[3]) {
    const size_t len = htab[0].size(), half = len / 2;
    e[0] = e[1] = e[2] = Fq::zero();
    if (factored) {
      std::vector<Fq> E(half);  // eq(tau_{j+1..}, .) over the variables still unbound after this round's
      host_eq(tau->data() + j + 1, (size_t)(rounds - j - 1), E.data());
      for (size_t i = 0; i < half; i++) {
        const Fq a0 = htab[0][i], a1 = htab[0][i + half], b0 = htab[1][i], b1 = htab[1][i + half], c0 = htab[2][i], c1 = htab[2][i + half];
        const Fq da = a1 - a0, db = b1 - b0, dc = c1 - c0;
        if (lead_now) {
          e[0] = e[0] + E[i] * (a0 * b0 - c0);
          e[1] = e[1] + E[i] * (da * db);
          if (j == 0) e[2] = e[2] + E[i] * (a1 * b1 - c1);
        } else {
          const Fq a2 = a1 + da, b2 = b1 + db, c2 = c1 + dc, a3 = a2 + da, b3 = b2 + db, c3 = c2 + dc;
          e[0] = e[0] + E[i] * (a0 * b0 - c0);
          e[1] = e[1] + E[i] * (a2 * b2 - c2);
          e[2] = e[2] + E[i] * (a3 * b3 - c3);
        }
      }
    } else {
      for (size_t i = 0; i < half; i++) {
        const Fq a0 = htab[0][i], a1 = htab[0][i + half], b0 = htab[1][i], b1 = htab[1][i + half];
        e[0] = e[0] + a0 * b0;
        e[1] = e[1] + (a1 + a1 - a0) * (b1 + b1 - b0);
      }
    }
  };
  Fq r_j = Fq::zero();
  static const bool fine = getenv("VPIN_SPARK_TRACE") && atoi(getenv("VPIN_SPARK_TRACE")) >= 2;
  Clock::time_point tp0 = Clock::now();
  for (int j = 0; j < rounds; j++) {
    Fq e[3];
    if (j < loc_rounds) {
      if ((rc = vpin::sc_round_wait(c, K, B(e)))) return rc;
      if (cm) {  // this rank's partial sums -> everyone's total
        if (factored) for (int i = 0; i < 3; i++) e[i] = e[i] * c_rank;
        std::vector<Fq> all(3 * (size_t)world);
        if ((rc = vpin::comm_allgather_ctx(c, e, all.data(), 96, K == 4 ? "sat_phase1_round" : "sat_phase2_round"))) return rc;
        for (int i = 0; i < 3; i++) {
          e[i] = Fq::zero();
          for (int r = 0; r < world; r++) e[i] = e[i] + all[3 * (size_t)r + i];
        }
      }
    } else {
      host_round(j, lead, e);
    }
    Clock::time_point tp1 = Clock::now();
    bool lead_next = lead;
    Fq t0, t1, tinf;
    if (factored) {
      const Fq& t = (*tau)[j];
      if (lead) {
        t0 = e[0]; tinf = e[1];
        if (j == 0) {
          t1 = e[2];
          lead_next = (cn == (f_one - t) * t0 + t * t1);  // s_eq = 1
        } else {
          t1 = (cn - (f_one - t) * t0) * tau_inv[j];
        }
        const Fq d10 = t1 - t0, two_inf = tinf + tinf;
        e[1] = t1 + d10 + two_inf;                    // t(2) = 2 t(1) - t(0) + 2 tinf
        e[2] = e[1] + d10 + two_inf + two_inf;        // t(3) = 3 t(1) - 2 t(0) + 6 tinf
      }
      e[0] = e[0] * (s_eq * (f_one - t));
      e[1] = e[1] * (s_eq * (f_three * t - f_one));
      e[2] = e[2] * (s_eq * (f_five * t - f_two));
    }
    Fq evals[4], coeffs[4];
    evals[0] = e[0]; evals[1] = claim_pr - e[0]; evals[2] = e[1];
    if (K == 4) evals[3] = e[2];
    unipoly_from_evals(evals, nc, coeffs);
    Point cp = P_bp[j];
    for (int i = 0; i < nc; i++) gn.G[i]->mul_acc(cp, coeffs[i]);
    CG comm_poly = compress(cp);
    tr.append_point("comm_poly", comm_poly.b);
    pf.comm_polys.push_back(comm_poly);
    r_j = tr.challenge_scalar("challenge_nextround");
    // fold with r_j and evaluate the next round while the host finishes this one
    if (j + 1 < loc_rounds) {
      rc = factored ? vpin::sc_cubic3_launch(c, tabs, pyramid, loc_rounds, j + 2, B(&r_j), lead_next) : vpin::sc_round_launch(c, K, tabs, B(&r_j));
      if (rc) return rc;
    } else if (cm && j + 1 == loc_rounds) {
      // the local tables are down to two entries: fold them with r_j and gather the world entries of every table
      Fq mine[4] = {Fq::zero(), Fq::zero(), Fq::zero(), Fq::zero()};
      if (tabs[0]->len != 2) return VPIN_ESHAPE;
      if ((rc = vpin::sc_final_claims(c, tabs, ntab, B(&r_j), B(mine)))) return rc;
      std::vector<Fq> all(4 * (size_t)world);
      if ((rc = vpin::comm_allgather_ctx(c, mine, all.data(), 128, "sat_gather_tables"))) return rc;
      htab.assign(ntab, std::vector<Fq>(world));
      for (int t = 0; t < ntab; t++)
        for (int r = 0; r < world; r++) htab[t][r] = all[4 * (size_t)r + t];
    } else if (cm && j + 1 < rounds) {
      for (auto& T : htab) {  // bound_poly_var_top on the host tables
        const size_t half = T.size() / 2;
        for (size_t i = 0; i < half; i++) T[i] = T[i] + r_j * (T[i + half] - T[i]);
        T.resize(half);
      }
    }
    Clock::time_point tp2 = Clock::now();
    if (lead) cn = t0 + r_j * ((t1 - t0 - tinf) + r_j * tinf);  // t(r_j)
    lead = lead_next;
    if (factored) {  // s_{j+1} = s_j * eq1(tau_j, r_j)
      const Fq& t = (*tau)[j];
      s_eq = s_eq * (t * r_j + (f_one - t) * (f_one - r_j));
    }
    Fq eval = unipoly_eval(coeffs, nc, r_j);
    Point ce = P_be[j];
    g1.G[0]->mul_acc(ce, eval);
    CG comm_eval = compress(ce);
    tr.append_point("comm_claim_per_round", comm_claim.b);
    tr.append_point("comm_eval", comm_eval.b);
    std::vector<Fq> w = tr.challenge_vector("combine_two_claims_to_one", 2);
    Fq target = w[0] * claim_pr + w[1] * eval;
    const Fq& blind_sc = (j == 0) ? blind_claim : blinds_evals[j - 1];
    Fq blind = w[0] * blind_sc + w[1] * blinds_evals[j];
    Fq a[4], pw = Fq::one();
    for (int i = 0; i < nc; i++) {
      Fq a_sc = (i == 0) ? Fq::from_u64(2) : Fq::one();
      a[i] = w[0] * a_sc + w[1] * pw;
      pw = pw * r_j;
    }
    // DotProductProof::prove (nizk/mod.rs:315-374); Cx is this round's comm_poly
    pf.proofs.emplace_back();
    DotProof& dp = pf.proofs.back();
    tr.append_protocol_name("dot product proof");
    tr.append_point("Cx", comm_poly.b);
    CG Cy = compress(commit1(target, blind, g1));
    tr.append_point("Cy", Cy.b);
    tr.append_scalars("a", a, nc);
    dp.delta = delta[j];
    tr.append_point("delta", dp.delta.b);
    Fq ad = Fq::zero();
    for (int i = 0; i < nc; i++) ad = ad + a[i] * rr[j].d[i];
    Point pb = P_rb[j];
    g1.G[0]->mul_acc(pb, ad);
    dp.beta = compress(pb);
    tr.append_point("beta", dp.beta.b);
    Fq cc = tr.challenge_scalar("c");
    dp.z.resize(nc);
    for (int i = 0; i < nc; i++) dp.z[i] = cc * coeffs[i] + rr[j].d[i];
    dp.z_delta = cc * blinds_poly[j] + rr[j].r_delta;
    dp.z_beta = cc * blind + rr[j].r_beta;
    claim_pr = eval;
    comm_claim = comm_eval;
    r_out.push_back(r_j);
    pf.comm_evals.push_back(comm_eval);
    if (fine) {  // w = wait for the round's sums, c = comm_poly .. next launch (critical path), h = the rest of the host's round
      Clock::time_point tp3 = Clock::now();
      fprintf(stderr, " w%.1f c%.1f h%.1f", 1e6 * secs(tp0, tp1), 1e6 * secs(tp1, tp2), 1e6 * secs(tp2, tp3));
      if (j == rounds - 1) fprintf(stderr, "\n");
      tp0 = tp3;
    }
  }
  // last fold (sumcheck.rs:673-676 of the final round), then the final claims P[0]
  if (factored) final_claims[0] = s_eq;  // tau(rx) = prod_i eq1(tau_i, r_i)
  if (cm) {
    for (int t = 0; t < ntab; t++) {
      if (htab[t].size() != 2) return VPIN_ESHAPE;
      final_claims[(factored ? 1 : 0) + t] = htab[t][0] + r_j * (htab[t][1] - htab[t][0]);
    }
  } else if (tabs[0]->len == 2) {
    rc = vpin::sc_final_claims(c, tabs, ntab, B(&r_j), B(&final_claims[factored ? 1 : 0]));
    if (rc) return rc;
  } else {
    rc = vpin_sc_bind(c, tabs, ntab, B(&r_j));
    if (rc) return rc;
    for (int k = 0; k < ntab; k++) {
      rc = vpin_table_read(c, tabs[k], 0, 1, B(&final_claims[factored ? k + 1 : k]));
      if (rc) return rc;
    }
  }
  blind_last = blinds_evals[rounds - 1];
  return VPIN_OK;
}

// ---- R1CS helpers (host, O(nnz)) -------------------------------------------------------------

[[maybe_unused]] static inline void acc_mul(Fq& dst, const Fq& val, const Fq& x, const Fq& one, const Fq& m1) {
  if (x.is_zero()) return;
  if (val == one) dst = dst + x;
  else if (val == m1) dst = dst - x;
  else dst = dst + val * x;
}

static thread_local double g_timings[8];  // per host thread: concurrent proofs on separate contexts

}  // namespace

extern "C" {

// MultiCommitGens::new on the host (no GPU needed): nb points as X|Y|Z|T
int vpin_host_gens_derive(const char* label, size_t nb, uint8_t* out_xyzt) {
  if (!label || !out_xyzt || nb == 0) return VPIN_EINVAL;
  std::vector<Point> g;
  derive_gens(g, nb, label);
  for (size_t i = 0; i < nb; i++) g[i].to_xyzt(out_xyzt + 128 * i);
  return VPIN_OK;
}

// host transcript self-test hook: Transcript::new(label); append_message; challenge_bytes
int vpin_host_merlin_kat(const char* proto, const char* label, const uint8_t* msg, size_t n, const char* clabel,
                         uint8_t* out, size_t out_n) {
  if (!proto || !label || !clabel || !out) return VPIN_EINVAL;
  Transcript t(proto);
  t.append_message(label, msg, n);
  t.challenge_bytes(clabel, out, out_n);
  return VPIN_OK;
}

// a*P + b*Q through the host's variable-base paths: Straus for the pair, and the single multiplications added up must agree
int vpin_host_scalar_mul2(const uint8_t a_mont[32], const uint8_t P[32], const uint8_t b_mont[32], const uint8_t Q[32], uint8_t out[32]) {
  if (!a_mont || !P || !b_mont || !Q || !out) return VPIN_EINVAL;
  Point p, q;
  if (!Point::decompress(p, P) || !Point::decompress(q, Q)) return VPIN_EVERIFY;
  Fq a, b;
  memcpy(a.l, a_mont, 32);
  memcpy(b.l, b_mont, 32);
  const Point joint = Point::mul2(a, p, b, q), apart = p.mul(a) + q.mul(b);
  if (!joint.equals(apart)) return VPIN_EHIP;  // an arithmetic fault of this library, not of the input
  joint.compress(out);
  return VPIN_OK;
}

// host Pedersen commitment self-test hook: sum v[i]*g[i] + blind*g[n] under `label`, compressed
int vpin_host_commit(const char* label, const uint8_t* v_mont, size_t n, const uint8_t* blind_mont, uint8_t out[32]) {
  if (!label || !v_mont || !blind_mont || !out || n == 0 || n > 4) return VPIN_EINVAL;
  std::vector<Point> g;
  derive_gens(g, n + 1, label);
  std::vector<FixedBase> fb;
  for (auto& p : g) fb.emplace_back(p);
  Mcg m{(int)n, {&fb[0], n > 1 ? &fb[1] : nullptr, n > 2 ? &fb[2] : nullptr, n > 3 ? &fb[3] : nullptr}, &fb[n]};
  Fq blind;
  memcpy(blind.l, blind_mont, 32);
  commit(reinterpret_cast<const Fq*>(v_mont), (int)n, blind, m).compress(out);
  return VPIN_OK;
}

// BulletReductionProof::prove (nizk/bullet.rs:32-132) with the challenges GIVEN (u[k], Montgomery) instead of drawn from a
// transcript, over the stream generators of `g` only (no Q, no H): per round the cross inner products c_L | c_R, the
// compressed points <a_L, G_R> | <a_R, G_L>, and at the end x_hat | a_hat and the compressed g_hat.  The parity handle
// of bullet.hip / msm.hip's bullet_step_kernel: classic != 0 forces the three-launch rounds.
int vpin_bullet_reduce(vpin_ctx* c, const vpin_gens* g, const uint8_t* x_mont, const uint8_t* a_mont, size_t R, const uint8_t* u_mont,
                       int classic, uint8_t* cLR_out, uint8_t* LR_out, uint8_t xhat_ahat_out[64], uint8_t ghat_out[32]) {
  if (!c || !g || !x_mont || !a_mont || !u_mont || !cLR_out || !LR_out || !xhat_ahat_out || !ghat_out || !vpin::is_pow2(R) || R < 2)
    return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  vpin::BulletState* bs = nullptr;
  int rc = vpin::bullet_begin(c, x_mont, a_mont, R, &bs);
  if (rc) return rc;
  struct Guard { vpin_ctx* c; vpin::BulletState* s; ~Guard() { (void)hipStreamSynchronize(c->stream); vpin::bullet_free(c, s); } } guard{c, bs};
  const bool fused = !classic && vpin::bullet_fused(bs);
  if (!classic && !fused) return VPIN_ESHAPE;  // the caller asked for the fused rounds and this R does not have them
  const size_t lgR = log2z(R), np = vpin_gens_msm_parts_count(R);
  auto sum_side = [&](int row, size_t nn) {
    const uint8_t* ptrs[256];
    const size_t cnt = vpin::bullet_part_ptrs(c, bs, nn, row, ptrs);
    Point acc = Point::identity();
    for (size_t k = 0; k < cnt; k++) acc = acc + Point::from_xyzt(ptrs[k]);
    return acc;
  };
  std::vector<uint8_t> parts(2 * np * 128);
  const Fq* u = reinterpret_cast<const Fq*>(u_mont);
  Fq u_prev = Fq::zero(), ui_prev = Fq::zero();
  size_t n = R;
  for (size_t k = 0; k < lgR; k++) {
    n /= 2;
    if (fused) rc = vpin::bullet_step(c, g, bs, n, k ? B(&u_prev) : nullptr, k ? B(&ui_prev) : nullptr, cLR_out + 64 * k);
    else rc = vpin::bullet_round_begin(c, g, bs, n, parts.data(), cLR_out + 64 * k);
    if (rc) return rc;
    if ((rc = vpin::bullet_round_end(c))) return rc;
    for (int row = 0; row < 2; row++) {
      Point acc = Point::identity();
      if (fused) acc = sum_side(row, n);
      else
        for (size_t p = 0; p < np; p++) acc = acc + Point::from_xyzt(parts.data() + ((size_t)row * np + p) * 128);
      acc.compress(LR_out + 64 * k + 32 * row);
    }
    u_prev = u[k];
    ui_prev = u[k].invert();
    if (!fused && (rc = vpin::bullet_fold(c, bs, n, B(&u_prev), B(&ui_prev)))) return rc;
  }
  Point ghat = Point::identity();
  if (fused) {
    if ((rc = vpin::bullet_finish_fused(c, g, bs, B(&u_prev), B(&ui_prev), xhat_ahat_out))) return rc;
    ghat = sum_side(0, 0);
  } else {
    if ((rc = vpin::bullet_finish(c, g, bs, xhat_ahat_out, parts.data()))) return rc;
    for (size_t p = 0; p < np; p++) ghat = ghat + Point::from_xyzt(parts.data() + p * 128);
  }
  ghat.compress(ghat_out);
  return VPIN_OK;
}

void vpin_sat_last_timings(double out[8]) { memcpy(out, g_timings, sizeof g_timings); }

size_t vpin_sat_proof_max_bytes(size_t num_cons, size_t num_vars) {
  size_t lx = log2z(num_cons), ly = log2z(2 * num_vars), ell = log2z(num_vars);
  size_t L = (size_t)1 << (ell / 2), lgR = ell - ell / 2;
  return 8 + 32 * L + (lx + ly) * 400 + 2048 + 64 * lgR;
}

}  // extern "C"

namespace vpin_prover {

// my_lib_prove up to and including the Ar/Br/Cr claims.  tr_out / tape_out (optional) receive the
// transcript and the RandomTape as they stand afterwards, for R1CSEvalProof::prove (spark.cpp).
// what a rank is about to prove and which per-process switches shape its collective sequence
static int dist_fingerprint_check(vpin_ctx* c, size_t nv, size_t ncons, size_t ni, const uint8_t seed_commit64[64],
                                  const uint8_t seed_proof64[64]) {
  vpin_comm* cm = c->comm;
  auto fnv = [](const uint8_t* p, size_t n) { uint64_t h = 0xcbf29ce484222325ull; for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; } return h; };
  auto env_num = [](const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; };
  uint64_t fp[10] = {
      (uint64_t)nv, (uint64_t)ncons, (uint64_t)ni, fnv(seed_commit64, 64), fnv(seed_proof64, 64),
      (uint64_t)(getenv("VPIN_DIST_NO_SAT_SPLIT") != nullptr), (uint64_t)(getenv("VPIN_DIST_BY_CIRCUIT") != nullptr),
      (uint64_t)env_num("VPIN_SPARK_TAIL_PAIRS", 1024), (uint64_t)(getenv("VPIN_NO_HOT_COLS") != nullptr),
      (uint64_t)(getenv("VPIN_BULLET_CLASSIC") != nullptr)};
  std::vector<uint64_t> all((size_t)cm->world * 10);
  int rc = vpin::comm_allgather_ctx(c, fp, all.data(), sizeof fp, "fingerprint");
  if (rc) return rc;
  static const char* what[10] = {"num_vars", "num_cons", "num_inputs", "seed_commit", "seed_proof", "VPIN_DIST_NO_SAT_SPLIT",
                                 "VPIN_DIST_BY_CIRCUIT", "VPIN_SPARK_TAIL_PAIRS", "VPIN_NO_HOT_COLS", "VPIN_BULLET_CLASSIC"};
  for (int r = 0; r < cm->world; r++)
    for (int k = 0; k < 10; k++)
      if (all[(size_t)r * 10 + k] != fp[k]) {
        char msg[200];
        snprintf(msg, sizeof msg, "collective proof: rank %d and rank %d differ in %s (%llu vs %llu)", cm->rank, r, what[k],
                 (unsigned long long)fp[k], (unsigned long long)all[(size_t)r * 10 + k]);
        vpin::set_last_error(msg, hipErrorUnknown);
        return VPIN_ECOMM;
      }
  return VPIN_OK;
}

int sat_prove_core(vpin_ctx* c, const vpin_r1cs_dev* dinst, size_t nv, size_t ncons, size_t ni,
                   const vpin_table* d_para, const vpin_table* d_input, const vpin_table* d_vars, const uint8_t* inputs,
                   const uint8_t seed_commit64[64], const uint8_t seed_proof64[64], uint8_t* proof_out, size_t proof_cap,
                   size_t* proof_len, uint8_t* comm_para_out, uint8_t* comm_input_out, uint8_t inst_evals_out[96],
                   uint8_t* rx_out, uint8_t* ry_out, vpin_host::Transcript* tr_out, vpin_host::Transcript* tape_out) {
  auto t_begin = Clock::now();
  memset(g_timings, 0, sizeof g_timings);
  (void)hipSetDevice(c->device);

  SatGens* sg = nullptr;
  auto t0 = Clock::now();
  int rc = get_gens(c, nv, &sg);
  if (rc) return rc;
  g_timings[5] = secs(t0, Clock::now());
  const size_t L = sg->L, R = sg->R;
  TableGuard tg(c);

  // ---- polycommit: proof_point_mult.rs:44-80 ----
  t0 = Clock::now();
  // the two row-MSMs do not depend on the blinds: enqueue them, then draw the 2L blinds while they run
  // One proof over several GPUs (vpin_ctx_set_comm): the L row commitments are independent MSMs (rayon's into_par_iter,
  // dense_mlpoly.rs:166-173), so every rank commits a contiguous block of rows of both polynomials and the 32-byte results
  // are all-gathered; the blinds are drawn in full by everyone (the tape is part of the deterministic protocol state).
  vpin_comm* cm = (c->comm && c->comm->world > 1) ? c->comm : nullptr;
  // Before anything is split: the ranks must be about to run the SAME protocol.  Several branches below and in spark_prove
  // are chosen per process (environment knobs, the instance's shape, the seeds): ranks that differ would issue different
  // collective sequences and, at best, stall until the timeout.  One all-gather of a fingerprint makes that a prompt
  // VPIN_ECOMM with a message instead.
  if (cm && (rc = dist_fingerprint_check(c, nv, ncons, ni, seed_commit64, seed_proof64))) return rc;
  // rank r commits rows r, r + world, ..: the rows differ in cost (the padding tail of the assignment is all zeros)
  const size_t nrows = cm ? vpin::comm_strided_count(L, cm->rank, cm->world) : L;
  vpin::CommitPairState* cps = nullptr;
  if (nrows && (rc = vpin::commit_pair_begin(c, sg->dev, d_para, d_input, L, &cps, cm ? (size_t)cm->rank : 0, nrows, cm ? (size_t)cm->world : 1)))
    return rc;
  const uint8_t two = 2;
  Transcript tape1 = make_tape(&two, 1, seed_commit64);
  std::vector<Fq> blind_para = tape1.challenge_vector("poly_blinds", L);
  std::vector<Fq> blind_input = tape1.challenge_vector("poly_blinds", L);
  std::vector<Fq> blind_vars(L);
  for (size_t i = 0; i < L; i++) blind_vars[i] = blind_para[i] + blind_input[i];  // commit_test.rs:42-54
  std::vector<CG> comm_vars(L);
  if (!cm) {
    rc = vpin::commit_pair_finish(c, sg->dev, cps, B(blind_para.data()), B(blind_input.data()), R + 1, comm_para_out,
                                  comm_input_out, comm_vars[0].b);
    if (rc) return rc;
  } else {
    const size_t pmax = vpin::comm_block_max(L, cm->world);
    std::vector<uint8_t> mine(3 * pmax * 32, 0), all((size_t)cm->world * 3 * pmax * 32);
    if (nrows) {
      std::vector<Fq> bp(nrows), bi(nrows);  // this rank's rows' blinds, in its row order
      for (size_t k = 0; k < nrows; k++) { bp[k] = blind_para[cm->rank + k * cm->world]; bi[k] = blind_input[cm->rank + k * cm->world]; }
      if ((rc = vpin::commit_pair_finish(c, sg->dev, cps, B(bp.data()), B(bi.data()), R + 1, mine.data(), mine.data() + pmax * 32,
                                         mine.data() + 2 * pmax * 32)))
        return rc;
    }
    if ((rc = vpin::comm_allgather_ctx(c, mine.data(), all.data(), mine.size(), "witness_commit"))) return rc;
    for (int r = 0; r < cm->world; r++) {
      const size_t n = vpin::comm_strided_count(L, r, cm->world);
      const uint8_t* blk = all.data() + (size_t)r * 3 * pmax * 32;
      for (size_t k = 0; k < n; k++) {
        const size_t row = (size_t)r + k * (size_t)cm->world;
        memcpy(comm_para_out + row * 32, blk + k * 32, 32);
        memcpy(comm_input_out + row * 32, blk + pmax * 32 + k * 32, 32);
        memcpy(comm_vars[row].b, blk + 2 * pmax * 32 + k * 32, 32);
      }
    }
  }
  g_timings[0] = secs(t0, Clock::now());

  // ---- transcripts: proof_point_mult.rs:83, commit_test.rs:74-75,148,155 ----
  Transcript tr("snark_example");
  Transcript tape = make_tape(reinterpret_cast<const uint8_t*>("proof"), 5, seed_proof64);
  tr.append_protocol_name("Spartan SNARK proof");
  tr.append_protocol_name("R1CS proof");
  tr.append_message("poly_commitment", "poly_commitment_begin");
  for (size_t i = 0; i < L; i++) tr.append_point("poly_commitment_share", comm_vars[i].b);
  tr.append_message("poly_commitment", "poly_commitment_end");

  // ---- phase 1 ----
  t0 = Clock::now();
  const size_t zl = 2 * nv;
  vpin_table* d_z = nullptr;
  if ((rc = vpin_r1cs_build_z(c, dinst, d_vars, inputs, &d_z))) return rc;
  tg.add(d_z);
  const int nrx = (int)log2z(ncons), nry = (int)log2z(zl);
  std::vector<Fq> tau = tr.challenge_vector("challenge_tau", nrx);
  // Both sum-checks over the ranks of c->comm (zk_sumcheck): a power-of-two world, and local tables long enough to be
  // worth a kernel; otherwise every rank runs them in full.
  int lw = 0;
  if (cm && (cm->world & (cm->world - 1)) == 0) {
    const int l = (int)log2z((size_t)cm->world);
    static const bool off = getenv("VPIN_DIST_NO_SAT_SPLIT") != nullptr;
    if (!off && nrx - l >= 12 && nry - l >= 12) lw = l;
  }
  const size_t wstep = (size_t)1 << lw, wrank = lw ? (size_t)cm->rank : 0;
  vpin_table* d_pyr = nullptr;  // suffix tables eq(tau_{k..}, .), k = 1..nrx (of the first nrx - lw challenges when split)
  if ((rc = vpin_eq_suffix_tables(c, B(tau.data()), nrx - lw, &d_pyr))) return rc;
  tg.add(d_pyr);
  const Fq one = Fq::one();
  vpin_table* d_abc[3] = {nullptr, nullptr, nullptr};
  auto t_spmv = Clock::now();
  if (lw) rc = vpin::r1cs_multiply_vec_strided(c, dinst, d_z, wrank, wstep, &d_abc[0], &d_abc[1], &d_abc[2]);  // rows = rank (mod world)
  else rc = vpin_r1cs_multiply_vec(c, dinst, d_z, &d_abc[0], &d_abc[1], &d_abc[2]);
  if (rc) return rc;
  for (int m = 0; m < 3; m++) tg.add(d_abc[m]);
  g_timings[6] += secs(t_spmv, Clock::now());
  ZkSc sc1, sc2;
  std::vector<Fq> rx, ry;
  Fq claims1[4], blind_post1;
  vpin_table* tabs1[3] = {d_abc[0], d_abc[1], d_abc[2]};
  rc = zk_sumcheck(c, 4, tabs1, Fq::zero(), Fq::zero(), nrx, sg->gens_1, sg->gens_4, tr, tape, sc1, rx, claims1, blind_post1,
                   &tau, d_pyr, lw);
  if (rc) return rc;
  if ((rc = vpin::comm_mark(c, "sat_phase1_rest"))) return rc;
  vpin::ctx_enter_alt(c);  // vpin_ctx_set_cumask_after_phase1: the rest of the proof runs on the context's own CUs
  if (c->progress_flag) *c->progress_flag = 1;  // phase 1 (the roofline kernel's launches) is over: other streams may start
  g_timings[1] = secs(t0, Clock::now());

  const Fq tau_claim = claims1[0], Az_claim = claims1[1], Bz_claim = claims1[2], Cz_claim = claims1[3];
  Fq Az_blind = tape.challenge_scalar("Az_blind"), Bz_blind = tape.challenge_scalar("Bz_blind"),
     Cz_blind = tape.challenge_scalar("Cz_blind"), prod_blind = tape.challenge_scalar("prod_Az_Bz_blind");
  KnowProof pok_Cz;
  CG comm_Cz = knowledge_prove(pok_Cz, sg->gens_1, tr, tape, Cz_claim, Cz_blind);
  ProdProof pprod;
  CG comm_Az, comm_Bz, comm_prod;
  product_prove(pprod, sg->gens_1, tr, tape, Az_claim, Az_blind, Bz_claim, Bz_blind, Az_claim * Bz_claim, prod_blind, comm_Az,
                comm_Bz, comm_prod);
  tr.append_point("comm_Az_claim", comm_Az.b);
  tr.append_point("comm_Bz_claim", comm_Bz.b);
  tr.append_point("comm_Cz_claim", comm_Cz.b);
  tr.append_point("comm_prod_Az_Bz_claims", comm_prod.b);
  Fq blind_expected1 = tau_claim * (prod_blind - Cz_blind);
  Fq claim_post1 = (Az_claim * Bz_claim - Cz_claim) * tau_claim;
  EqProof eq1;
  equality_prove(eq1, sg->gens_1, tr, tape, claim_post1, blind_expected1, claim_post1, blind_post1);

  // ---- phase 2 ----
  t0 = Clock::now();
  Fq r_A = tr.challenge_scalar("challenege_Az"), r_B = tr.challenge_scalar("challenege_Bz"),
     r_C = tr.challenge_scalar("challenege_Cz");
  Fq claim2 = r_A * Az_claim + r_B * Bz_claim + r_C * Cz_claim;
  Fq blind_claim2 = r_A * Az_blind + r_B * Bz_blind + r_C * Cz_blind;
  vpin_table *d_eq_rx = nullptr, *d_eabc = nullptr;
  if ((rc = vpin_eq_table(c, B(rx.data()), nrx, &d_eq_rx))) return rc;
  tg.add(d_eq_rx);
  t_spmv = Clock::now();
  {
    // r_A*A(rx,.) + r_B*B(rx,.) + r_C*C(rx,.) (compute_eval_table_sparse x3, commit_test.rs:257-268)
    Fq rabc[3] = {r_A, r_B, r_C};
    if (lw) rc = vpin::r1cs_eval_table_strided(c, dinst, d_eq_rx, B(rabc), wrank, wstep, &d_eabc);  // columns = rank (mod world)
    else rc = vpin_r1cs_eval_table(c, dinst, d_eq_rx, B(rabc), &d_eabc);
    if (rc) return rc;
    tg.add(d_eabc);
  }
  g_timings[6] += secs(t_spmv, Clock::now());
  Fq claims2[2], blind_post2;
  vpin_table* d_z2 = d_z;
  if (lw) {  // z at the same columns
    if ((rc = vpin::table_take_strided(c, d_z, wrank, wstep, &d_z2))) return rc;
    tg.add(d_z2);
  }
  vpin_table* tabs2[2] = {d_z2, d_eabc};
  rc = zk_sumcheck(c, 2, tabs2, claim2, blind_claim2, nry, sg->gens_1, sg->gens_3, tr, tape, sc2, ry, claims2, blind_post2, nullptr,
                   nullptr, lw);
  if (rc) return rc;
  if ((rc = vpin::comm_mark(c, "sat_phase2_rest"))) return rc;
  g_timings[2] = secs(t0, Clock::now());

  // ---- polyeval: commit_test.rs:283-296, dense_mlpoly.rs:326-379 ----
  t0 = Clock::now();
  const size_t left = sg->ell / 2, right = sg->ell - left;
  std::vector<Fq> Lv(L), Rv(R), LZ(R);
  host_eq(ry.data() + 1, left, Lv.data());
  host_eq(ry.data() + 1 + left, right, Rv.data());
  if (cm) rc = vpin::poly_bound_dist(c, d_vars, B(Lv.data()), L, B(LZ.data()));  // row blocks, partial vectors all-gathered
  else rc = vpin_poly_bound(c, d_vars, B(Lv.data()), L, B(LZ.data()));
  if (rc) return rc;
  // poly_vars.evaluate(ry[1..]) = <L*Z, R>
  Fq eval_vars_at_ry = Fq::zero();
  for (size_t i = 0; i < R; i++) eval_vars_at_ry = eval_vars_at_ry + LZ[i] * Rv[i];
  Fq blind_eval = tape.challenge_scalar("blind_eval");
  tr.append_protocol_name("polynomial evaluation proof");
  Fq LZ_blind = Fq::zero();
  for (size_t i = 0; i < L; i++) LZ_blind = LZ_blind + blind_vars[i] * Lv[i];

  DpLog pe;
  if ((rc = dplog_prove(c, sg->pc, tr, tape, LZ, LZ_blind, Rv, eval_vars_at_ry, blind_eval, pe))) return rc;
  g_timings[3] = secs(t0, Clock::now());

  Fq blind_eval_Z = (one - ry[0]) * blind_eval;
  Fq blind_expected2 = claims2[1] * blind_eval_Z;
  Fq claim_post2 = claims2[0] * claims2[1];
  EqProof eq2;
  equality_prove(eq2, sg->gens_1, tr, tape, claim_post2, blind_expected2, claim_post2, blind_post2);

  // ---- inst.evaluate(rx, ry) and the claims my_lib_prove appends (commit_test.rs:100-109) ----
  t0 = Clock::now();
  Fq ie[3];
  {
    vpin_table* d_eq_ry = nullptr;
    if ((rc = vpin_eq_table(c, B(ry.data()), nry, &d_eq_ry))) return rc;
    tg.add(d_eq_ry);
    if ((rc = vpin_r1cs_evaluate(c, dinst, d_eq_rx, d_eq_ry, B(ie)))) return rc;
  }
  tr.append_scalar("Ar_claim", ie[0]);
  tr.append_scalar("Br_claim", ie[1]);
  tr.append_scalar("Cr_claim", ie[2]);
  memcpy(inst_evals_out, ie, 96);
  g_timings[7] = secs(t0, Clock::now());

  // ---- serialise R1CSProof (r1csproof.rs:21-47; bincode defaults) ----
  Writer w;
  w.u64(L);
  for (auto& cg : comm_vars) w.point(cg);
  write_zksc(w, sc1);
  w.point(comm_Az); w.point(comm_Bz); w.point(comm_Cz); w.point(comm_prod);
  w.point(pok_Cz.alpha); w.scalar(pok_Cz.z1); w.scalar(pok_Cz.z2);
  w.point(pprod.alpha); w.point(pprod.beta); w.point(pprod.delta);
  for (int i = 0; i < 5; i++) w.scalar(pprod.z[i]);
  w.point(eq1.alpha); w.scalar(eq1.z);
  write_zksc(w, sc2);
  w.point(pe.Cy);  // comm_vars_at_ry
  write_dplog(w, pe);
  w.point(eq2.alpha); w.scalar(eq2.z);
  if (w.buf.size() > proof_cap) return VPIN_ESHAPE;
  memcpy(proof_out, w.buf.data(), w.buf.size());
  *proof_len = w.buf.size();
  if (rx_out) memcpy(rx_out, rx.data(), rx.size() * 32);
  if (ry_out) memcpy(ry_out, ry.data(), ry.size() * 32);
  if ((rc = vpin::comm_mark(c, "sat_replicated"))) return rc;
  if (tr_out) *tr_out = tr;
  if (tape_out) *tape_out = tape;
  g_timings[4] = secs(t_begin, Clock::now());
  static const bool trace_on = getenv("VPIN_SPARK_TRACE") && atoi(getenv("VPIN_SPARK_TRACE")) >= 2;
  if (trace_on)
    fprintf(stderr, "[sat] 2^%d: polycommit %.3f  phase 1 %.3f  phase 2 %.3f  polyeval %.3f  inst.evaluate %.3f  (sparse products %.3f)  total %.3f ms\n",
            nrx, 1e3 * g_timings[0], 1e3 * g_timings[1], 1e3 * g_timings[2], 1e3 * g_timings[3], 1e3 * g_timings[7],
            1e3 * g_timings[6], 1e3 * g_timings[4]);
  return VPIN_OK;
}

}  // namespace vpin_prover

extern "C" {

// R1CSGens::new for this polynomial size, ahead of the first proof (host fixed-base tables + the
// shared device window table); proofs build them on demand otherwise
int vpin_sat_prepare(vpin_ctx* c, size_t num_vars) {
  if (!c || !vpin::is_pow2(num_vars)) return VPIN_EINVAL;
  SatGens* sg = nullptr;
  return get_gens(c, num_vars, &sg);
}

int vpin_sat_prove_resident(vpin_ctx* c, const vpin_r1cs_dev* dinst, const vpin_table* vars_para, const vpin_table* vars_input,
                            const vpin_table* vars, const uint8_t* inputs, const uint8_t seed_commit64[64],
                            const uint8_t seed_proof64[64], uint8_t* proof_out, size_t proof_cap, size_t* proof_len,
                            uint8_t* comm_para_out, uint8_t* comm_input_out, uint8_t inst_evals_out[96], uint8_t* rx_out,
                            uint8_t* ry_out) {
  if (!c || !dinst || !vars_para || !vars_input || !vars || !seed_commit64 || !seed_proof64 || !proof_out || !proof_len ||
      !comm_para_out || !comm_input_out || !inst_evals_out)
    return VPIN_EINVAL;
  size_t nv, ncons, ni;
  vpin_r1cs_dims(dinst, &ncons, &nv, &ni);
  if (vars_para->len != nv || vars_input->len != nv || vars->len != nv || (ni && !inputs)) return VPIN_ESHAPE;
  vpin::AltStreamGuard alt_guard(c);
  return vpin::comm_leave(c->comm, vpin_prover::sat_prove_core(c, dinst, nv, ncons, ni, vars_para, vars_input, vars, inputs, seed_commit64,
                                                               seed_proof64, proof_out, proof_cap, proof_len, comm_para_out,
                                                               comm_input_out, inst_evals_out, rx_out, ry_out, nullptr, nullptr));
}

// my_dense_mlpoly_commit (vPIN_proof_generation/src/commit_test.rs:27-57) as proof_point_{add,mult}.rs:58-59 call it
int vpin_dense_mlpoly_commit_sum(vpin_ctx* c, const vpin_table* vars, const uint8_t seed_commit64[64], uint8_t* out_compressed) {
  if (!c || !vars || !vars->d || !seed_commit64 || !out_compressed) return VPIN_EINVAL;
  const size_t nv = vars->len;
  if (!vpin::is_pow2(nv) || nv < 2) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  SatGens* sg = nullptr;
  int rc = get_gens(c, nv, &sg);
  if (rc) return rc;
  const size_t L = sg->L, R = sg->R;
  const uint8_t two = 2;
  Transcript tape1 = make_tape(&two, 1, seed_commit64);                       // RandomTape::new(&[2u8]), proof_point_mult.rs:44
  std::vector<Fq> blind_para = tape1.challenge_vector("poly_blinds", L);     // the blinds of the two commitments before it
  std::vector<Fq> blind_input = tape1.challenge_vector("poly_blinds", L);
  std::vector<Fq> blind_sum(L);
  for (size_t i = 0; i < L; i++) blind_sum[i] = blind_para[i] + blind_input[i];  // commit_test.rs:42-46
  return vpin_hyrax_commit(c, sg->dev, vars, B(blind_sum.data()), L, R + 1, out_compressed);  // commit_inner(&blinds, &gens.gens.gens_n)
}

int vpin_sat_prove(vpin_ctx* c, const vpin_r1cs* inst, const uint8_t* vars_para, const uint8_t* vars_input,
                   const uint8_t* vars, const uint8_t* inputs, const uint8_t seed_commit64[64],
                   const uint8_t seed_proof64[64], uint8_t* proof_out, size_t proof_cap, size_t* proof_len,
                   uint8_t* comm_para_out, uint8_t* comm_input_out, uint8_t inst_evals_out[96], uint8_t* rx_out,
                   uint8_t* ry_out) {
  if (!c || !inst || !vars_para || !vars_input || !vars) return VPIN_EINVAL;
  const size_t nv = inst->num_vars;
  if (!vpin::is_pow2(nv) || !vpin::is_pow2(inst->num_cons) || inst->num_inputs >= nv) return VPIN_ESHAPE;
  vpin_r1cs_dev* dinst = nullptr;
  vpin::TraceLap lap(c, "vpin_sat_prove");  // VPIN_CLI_TRACE=1
  int rc = vpin_r1cs_upload(c, inst, &dinst);
  if (rc) return rc;
  lap("r1cs_upload");
  TableGuard tg(c);
  vpin_table *d_para = nullptr, *d_input = nullptr, *d_vars = nullptr;
  rc = vpin_table_upload(c, vars_para, nv, &d_para);
  if (!rc) { tg.add(d_para); rc = vpin_table_upload(c, vars_input, nv, &d_input); }
  if (!rc) { tg.add(d_input); rc = vpin_table_upload(c, vars, nv, &d_vars); }
  lap("witness_upload");
  if (!rc) {
    tg.add(d_vars);
    rc = vpin_sat_prove_resident(c, dinst, d_para, d_input, d_vars, inputs, seed_commit64, seed_proof64, proof_out, proof_cap,
                                 proof_len, comm_para_out, comm_input_out, inst_evals_out, rx_out, ry_out);
  }
  vpin_r1cs_free(c, dinst);
  return rc;
}

}  // extern "C"
