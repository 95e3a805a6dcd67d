// sumcheck.hip -- multilinear sum-check round reductions and table folds for gfx950.
//
// Replaces the two hot loops of the reference's ZK sum-checks
//   Spartan/src/sumcheck.rs:624-652  (cubic, 4 tables, comb A*(B*C-D))
//   Spartan/src/sumcheck.rs:460-469  (quadratic, 2 tables, comb A*B)
// and DensePolynomial::bound_poly_var_top (Spartan/src/dense_mlpoly.rs:229-236).
//
// Data layout: a table is a flat array of 32-byte Montgomery elements in HBM; element i
// of the low half pairs with element i+len/2 (the TOP variable is bound first), so both
// streams are unit-stride and every lane issues 2 x global_load_dwordx4 per element.
// The fused kernel reads each live element once and writes each folded element once per
// round: 32*(len + len/2) bytes per table -- the algorithmic minimum of SURVEY.md 8(d).
// HBM-streaming integer work: no LDS tiling is useful (no reuse), no MFMA (255-bit
// modular arithmetic).  Per-thread partial sums are reduced with wavefront shuffles
// (64 lanes), then across the 4 waves of a block through LDS, then by a 1-block finisher.
#include <cstring>

#include "ctx.h"
#include "host/field.h"
#include "sc_dev.h"

namespace vpin {

fq_const make_fq_const(const uint8_t* p) {
  vpin_host::Fq r;
  memcpy(r.l, p, 32);
  fq_const out;
  for (int i = 0; i < 8; i++) {
    vpin_host::Fq pw = vpin_host::Fq::zero();
    pw.l[i / 2] = (uint64_t)1 << (32 * (i & 1));
    const vpin_host::Fq t = r * pw;  // r~ * 2^(32 i) * R^-1 mod q, canonical
    uint32_t limbs[8];
    memcpy(limbs, t.l, 32);
    for (int k = 0; k < 8; k++) out.tt[k][i] = limbs[k];
  }
  return out;
}

// Round evaluation on tables of live length 2*half: pairs (i, i+half).
template <int K>
__global__ __launch_bounds__(kBlock, kMinWaves) void sc_eval_kernel(Tabs<K> tabs, size_t half,
                                                         fq* __restrict__ partials) {
  Acc<K> acc;
  acc.init();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < half; i += (size_t)gridDim.x * kBlock) {
    if constexpr (K == 4) {
      fq u[3], p1, d1, p2, d2;
      load_pd(tabs.t[1], i, half, p1, d1);
      load_pd(tabs.t[2], i, half, p2, d2);
      acc.stage_bc(u, p1, d1, p2, d2);
      load_pd(tabs.t[3], i, half, p1, d1);
      acc.stage_d(u, p1, d1);
      load_pd(tabs.t[0], i, half, p1, d1);
      acc.stage_a(u, p1, d1);
    } else {
      fq p[K], d[K];
#pragma unroll
      for (int k = 0; k < K; k++) load_pd(tabs.t[k], i, half, p[k], d[k]);
      acc.add_pair(p, d);
    }
  }
  block_reduce_store<Acc<K>::NE>(acc.e, partials);
}

// Fused: fold the tables (live length 4*quarter) with r, store the folded halves, and
// evaluate the next round on the folded pair (i, i+quarter).
template <int K>
__global__ __launch_bounds__(kBlock, kMinWaves) void sc_bind_eval_kernel(Tabs<K> tabs, size_t quarter, fq r,
                                                              fq* __restrict__ partials) {
  Acc<K> acc;
  acc.init();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < quarter; i += (size_t)gridDim.x * kBlock) {
    if constexpr (K == 4) {
      fq u[3], p1, d1, p2, d2;
      fold_pd(tabs.t[1], i, quarter, r, p1, d1);
      fold_pd(tabs.t[2], i, quarter, r, p2, d2);
      acc.stage_bc(u, p1, d1, p2, d2);
      fold_pd(tabs.t[3], i, quarter, r, p1, d1);
      acc.stage_d(u, p1, d1);
      fold_pd(tabs.t[0], i, quarter, r, p1, d1);
      acc.stage_a(u, p1, d1);
    } else {
      fq p[K], d[K];
#pragma unroll
      for (int k = 0; k < K; k++) fold_pd(tabs.t[k], i, quarter, r, p[k], d[k]);
      acc.add_pair(p, d);
    }
  }
  block_reduce_store<Acc<K>::NE>(acc.e, partials);
}

// Phase-1 kernel without the eq(tau,.) table: three foldable tables (Az,Bz,Cz) and the suffix table
// E = eq(tau_{j+1..}, .) of this round (read-only, one element per pair).  12 Montgomery products
// and 13 loads / 6 stores per pair instead of 14 / 16 / 8; the three sums are scaled by the host.
// LEAD (sc_dev.h lead_bcd): return t(0) = sum E*(B_0 C_0 - D_0) and the x^2 coefficient sum E*dB*dC of the round's
// quadratic (10 products per pair); without BIND (the first round) also t(1), so the host can check the claim.
template <bool BIND, bool LEAD>
__global__ __launch_bounds__(kBlock, kMinWaves) void sc_cubic3_kernel(Tabs<3> tabs, const fq* __restrict__ E, size_t pairs,
                                                                      fq r, fq_const rc, fq* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) uint32_t tt[8][8];
  if (BIND && LEAD) {
    if (threadIdx.x < 64) tt[threadIdx.x >> 3][threadIdx.x & 7] = rc.tt[threadIdx.x >> 3][threadIdx.x & 7];
    __syncthreads();
  }
  Acc<4> acc;
  acc.init();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < pairs; i += (size_t)gridDim.x * kBlock) {
    fq u[3], p1, d1, p2, d2;
    if (BIND && LEAD) { fold_pd_c(tabs.t[0], i, pairs, tt, p1, d1); fold_pd_c(tabs.t[1], i, pairs, tt, p2, d2); }
    else if (BIND) { fold_pd(tabs.t[0], i, pairs, r, p1, d1); fold_pd(tabs.t[1], i, pairs, r, p2, d2); }
    else { load_pd(tabs.t[0], i, pairs, p1, d1); load_pd(tabs.t[1], i, pairs, p2, d2); }
    if (LEAD) {
      fq p3, d3;
      if (BIND) fold_pd_c(tabs.t[2], i, pairs, tt, p3, d3);
      else load_pd(tabs.t[2], i, pairs, p3, d3);
      const fq e = fq_load(E + i);
      acc.lead_bcd(p1, d1, p2, d2, p3, e);  // (the lazily reduced sums of spark.hip's product rounds do not pay here: same-box
                                            //  A/B 162.3 against 151.9-162.6 us per launch, profiles/r04_ab_lazy.txt)
      if (!BIND) acc.lead_one(p1, d1, p2, d2, p3, d3, e);
    } else {
      acc.stage_bc(u, p1, d1, p2, d2);
      if (BIND) fold_pd(tabs.t[2], i, pairs, r, p1, d1);
      else load_pd(tabs.t[2], i, pairs, p1, d1);
      acc.stage_d(u, p1, d1);
      acc.stage_e(u, fq_load(E + i));
    }
  }
  block_reduce_store<3>(acc.e, partials);
}

// Tail rounds (tables of at most kSmallPairs pairs): one workgroup does the fold, the evaluation
// and the whole reduction, and writes the 2-3 scalars straight to the pinned result buffer -- no
// partials, no finisher launch.  These rounds are launch-latency bound, not bandwidth bound.
constexpr int kSmallBlock = 512;
constexpr size_t kSmallPairs = 512;

template <int K, bool BIND>
__global__ __launch_bounds__(kSmallBlock) void sc_tail_kernel(Tabs<K> tabs, size_t pairs, fq r, fq* __restrict__ out) {
  Acc<K> acc;
  acc.init();
  const size_t i = threadIdx.x;
  if (i < pairs) {
    fq p[K], d[K];
    if (BIND) {
      const size_t quarter = pairs, half = 2 * pairs;
#pragma unroll
      for (int k = 0; k < K; k++) {
        fq a0 = fq_load(tabs.t[k] + i), a1 = fq_load(tabs.t[k] + half + i);
        fq b0 = fq_load(tabs.t[k] + quarter + i), b1 = fq_load(tabs.t[k] + half + quarter + i);
        p[k] = fq_add(a0, fq_mul(r, fq_sub(a1, a0)));
        fq hi = fq_add(b0, fq_mul(r, fq_sub(b1, b0)));
        fq_store(tabs.t[k] + i, p[k]);
        fq_store(tabs.t[k] + quarter + i, hi);
        d[k] = fq_sub(hi, p[k]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < K; k++) {
        p[k] = fq_load(tabs.t[k] + i);
        d[k] = fq_sub(fq_load(tabs.t[k] + pairs + i), p[k]);
      }
    }
    acc.add_pair(p, d);
  }
  constexpr int NE = Acc<K>::NE;
  __shared__ fq sh[kSmallBlock / 64][NE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NE; k++) {
    fq s = fq_wave_sum(acc.e[k]);
    if (lane == 0) sh[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NE) {
    fq s = sh[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kSmallBlock / 64; w++) s = fq_add(s, sh[w][threadIdx.x]);
    fq_store(&out[threadIdx.x], s);
  }
}

template <bool BIND, bool LEAD>
__global__ __launch_bounds__(kSmallBlock) void sc_tail3_kernel(Tabs<3> tabs, const fq* __restrict__ E, size_t pairs, fq r,
                                                               fq* __restrict__ out) {
  Acc<4> acc;
  acc.init();
  const size_t i = threadIdx.x;
  if (i < pairs) {
    fq u[3], p1, d1, p2, d2;
    if (BIND) { fold_pd(tabs.t[0], i, pairs, r, p1, d1); fold_pd(tabs.t[1], i, pairs, r, p2, d2); }
    else { load_pd(tabs.t[0], i, pairs, p1, d1); load_pd(tabs.t[1], i, pairs, p2, d2); }
    if (LEAD) {
      fq p3, d3;
      if (BIND) fold_pd(tabs.t[2], i, pairs, r, p3, d3);
      else load_pd(tabs.t[2], i, pairs, p3, d3);
      const fq e = fq_load(E + i);
      acc.lead_bcd(p1, d1, p2, d2, p3, e);
      if (!BIND) acc.lead_one(p1, d1, p2, d2, p3, d3, e);
    } else {
      acc.stage_bc(u, p1, d1, p2, d2);
      if (BIND) fold_pd(tabs.t[2], i, pairs, r, p1, d1);
      else load_pd(tabs.t[2], i, pairs, p1, d1);
      acc.stage_d(u, p1, d1);
      acc.stage_e(u, fq_load(E + i));
    }
  }
  __shared__ fq sh[kSmallBlock / 64][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    fq s = fq_wave_sum(acc.e[k]);
    if (lane == 0) sh[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    fq s = sh[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kSmallBlock / 64; w++) s = fq_add(s, sh[w][threadIdx.x]);
    fq_store(&out[threadIdx.x], s);
  }
}

// suffix-eq pyramid step: level k from level k+1 (m elements): dst[i] = (1-tau_k)*src[i], dst[m+i] = tau_k*src[i]
__global__ __launch_bounds__(kBlock) void eq_pyramid_step_kernel(const fq* __restrict__ src, fq* __restrict__ dst, size_t m,
                                                                 fq tau_k) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (size_t)gridDim.x * kBlock) {
    fq v = fq_load(src + i);
    fq hi = fq_mul(v, tau_k);
    fq_store(dst + i, fq_sub(v, hi));
    fq_store(dst + m + i, hi);
  }
}

// The small end of the pyramid in one workgroup: level ell (the single 1) down to level k_lo, every level
// of at most 512 elements.  taus.t[k - k_lo] = tau_k.
struct TauPack { fq t[10]; };
__global__ __launch_bounds__(512) void eq_pyramid_top_kernel(fq* __restrict__ base, int ell, int k_lo, TauPack taus) {
  const size_t n = (size_t)1 << ell;
  if (threadIdx.x == 0) fq_store(base + (n - 2), fq_one());  // level ell
  __syncthreads();
  for (int k = ell - 1; k >= k_lo; k--) {  // level k from level k+1
    const size_t m = (size_t)1 << (ell - k - 1);
    const fq* src = base + (n - ((size_t)2 << (ell - k - 1)));
    fq* dst = base + (n - ((size_t)2 << (ell - k)));
    if (threadIdx.x < m) {
      fq v = fq_load(src + threadIdx.x);
      fq hi = fq_mul(v, taus.t[k - k_lo]);
      fq_store(dst + threadIdx.x, fq_sub(v, hi));
      fq_store(dst + m + threadIdx.x, hi);
    }
    __syncthreads();
  }
}

// Plain fold of K tables (live length 2*half).
// last fold of a sum-check (live length 2): the k bound values go to the tables AND to pinned host memory, so the final
// claims cost one launch and one wait instead of a bind, its sync and k synchronous 32-byte copies (~25 us each)
__global__ __launch_bounds__(64) void sc_final_kernel(Tabs<4> tabs, int k, fq r, fq* __restrict__ out) {
  const int i = threadIdx.x;
  if (i >= k) return;
  const fq a0 = fq_load(tabs.t[i]), a1 = fq_load(tabs.t[i] + 1);
  const fq v = fq_add(a0, fq_mul(r, fq_sub(a1, a0)));
  fq_store(tabs.t[i], v);
  fq_store(out + i, v);
}

template <int K>
__global__ __launch_bounds__(kBlock) void sc_bind_kernel(Tabs<K> tabs, size_t half, fq r) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < half; i += (size_t)gridDim.x * kBlock) {
#pragma unroll
    for (int k = 0; k < K; k++) {
      fq a0 = fq_load(tabs.t[k] + i);
      fq a1 = fq_load(tabs.t[k] + half + i);
      fq_store(tabs.t[k] + i, fq_add(a0, fq_mul(r, fq_sub(a1, a0))));
    }
  }
}

// EqPolynomial::evals (dense_mlpoly.rs:78-94), one doubling step j: for the current size
// `size` (already doubled), evals[i] = evals[i/2]*r_j (i odd), evals[i-1] = evals[i/2]-evals[i].
// Out-of-place ping-pong so every thread reads src[i] and writes dst[2i], dst[2i+1].
__global__ __launch_bounds__(kBlock) void eq_step_kernel(const fq* __restrict__ src, fq* __restrict__ dst,
                                                         size_t prev, fq rj) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < prev; i += (size_t)gridDim.x * kBlock) {
    fq s = fq_load(src + i);
    fq hi = fq_mul(s, rj);
    fq_store(dst + 2 * i + 1, hi);
    fq_store(dst + 2 * i, fq_sub(s, hi));
  }
}

// The first k0 <= 9 doubling steps of EqPolynomial::evals in one workgroup (two LDS buffers of 512 elements), so a
// table of 2^ell entries costs ell - k0 + 1 launches instead of ell: out[0 .. 2^k0) = eq(r_0..r_{k0-1}, .)
struct EqHead { fq r[9]; };
__global__ __launch_bounds__(512) void eq_head_kernel(fq* __restrict__ out, int k0, EqHead rs) {
  __shared__ fq buf[2][512];
  if (threadIdx.x == 0) buf[0][0] = fq_one();
  __syncthreads();
  int cur = 0;
  for (int j = 0; j < k0; j++) {
    const size_t prev = (size_t)1 << j;
    if (threadIdx.x < prev) {
      const fq s = buf[cur][threadIdx.x];
      const fq hi = fq_mul(s, rs.r[j]);
      buf[cur ^ 1][2 * threadIdx.x + 1] = hi;
      buf[cur ^ 1][2 * threadIdx.x] = fq_sub(s, hi);
    }
    cur ^= 1;
    __syncthreads();
  }
  if (threadIdx.x < ((size_t)1 << k0)) fq_store(out + threadIdx.x, buf[cur][threadIdx.x]);
}

// ---- one-launch eq tables -----------------------------------------------------------------------------------------
// A doubling step per launch makes an eq table of 2^ell entries cost ell - 8 dependent launches and three times its size
// in HBM traffic; the factorisation  E[h * 512 + i] = H[h] * Lo[i]  (Lo = the table of the last nine variables, built by
// doubling in LDS by every workgroup; H[h] = the product of the h-bits' factors, ell - 9 products by one thread) writes
// every entry once from ONE launch.  Field values are canonical, so the bytes equal the doubling construction's.
struct TauAll { fq t[32]; };
constexpr int kEqLoVars = 9, kEqChunk = 1 << kEqLoVars;

// lo[cur][0..512) = eq(t[first..first+9), .) with t[first] the most significant index bit (new variable = LSB: EqPolynomial::evals)
__device__ __forceinline__ int eq_lo_build_lsb(fq (*buf)[kEqChunk], const TauAll& ts, int first) {
  if (threadIdx.x == 0) buf[0][0] = fq_one();
  __syncthreads();
  int cur = 0;
  for (int j = 0; j < kEqLoVars; j++) {
    const int prev = 1 << j;
    if ((int)threadIdx.x < prev) {
      const fq s = buf[cur][threadIdx.x];
      const fq hi = fq_mul(s, ts.t[first + j]);
      buf[cur ^ 1][2 * threadIdx.x + 1] = hi;
      buf[cur ^ 1][2 * threadIdx.x] = fq_sub(s, hi);
    }
    cur ^= 1;
    __syncthreads();
  }
  return cur;
}

// EqPolynomial::evals (dense_mlpoly.rs:78-94) of ell >= 10 variables in one launch.  A workgroup writes up to 32
// consecutive chunks of 512 entries: E[((b*32 + m) * 512) + i] = Hhi[b] * Hmid[m] * Lo[i]  (Hmid over the five variables
// above Lo's nine, Hhi over the rest: index MSB <-> r_0), so the per-workgroup set-up is shared by 16384 outputs.
constexpr int kEqMidVars = 5, kEqMid = 1 << kEqMidVars;
__global__ __launch_bounds__(kEqChunk) void eq_table_fused_kernel(fq* __restrict__ out, int ell, int nmid, TauAll rs) {
  __shared__ fq buf[2][kEqChunk];
  __shared__ fq hm[kEqMid];
  const int cur = eq_lo_build_lsb(buf, rs, ell - kEqLoVars);
  // nmid: variables r_{ell-9-nmid .. ell-10} (chunks per workgroup = 2^nmid, chosen by the launcher)
  const int nhi = ell - kEqLoVars - nmid;                                              // variables r_0 .. r_{nhi-1}
  const fq one = fq_one();
  if ((int)threadIdx.x < (1 << nmid)) {
    // Hhi[b] * Hmid[m]: m's bit (nmid-1-j) pairs with r_{nhi+j}; b's bit (nhi-1-j) with r_j
    fq run = one;
    const size_t bidx = blockIdx.x;
    for (int j = 0; j < nhi; j++) run = fq_mul(run, ((bidx >> (nhi - 1 - j)) & 1) ? rs.t[j] : fq_sub(one, rs.t[j]));
    const int m = threadIdx.x;
    for (int j = 0; j < nmid; j++) run = fq_mul(run, ((m >> (nmid - 1 - j)) & 1) ? rs.t[nhi + j] : fq_sub(one, rs.t[nhi + j]));
    hm[m] = run;
  }
  __syncthreads();
  const fq lo = buf[cur][threadIdx.x];
  fq* dst = out + ((size_t)blockIdx.x << (nmid + kEqLoVars)) + threadIdx.x;
  for (int m = 0; m < (1 << nmid); m++) fq_store(dst + (size_t)m * kEqChunk, fq_mul(hm[m], lo));
}

// All suffix tables of ell >= 11 variables (vpin_eq_suffix_tables' layout) in one launch.  Level k = eq(tau_k.., .) has
// 2^(ell-k) entries, tau_k on the most significant index bit; its chunk c (512 entries) is P_k(c) * Lo with Lo = level
// ell-9 and P_k(c) the product of the factors of c's low (ell-k-9) bits (bit j <-> tau_{ell-10-j}).  A workgroup covers
// 32 consecutive chunk indices c = 32 b + m and writes chunk c of every level that has one; workgroup 0 also writes the
// levels of fewer than 512 entries.  Prefix products: pm[m][j] over m's bits j < 5, pb[j] over b's bits above them.
__global__ __launch_bounds__(kEqChunk) void eq_pyramid_fused_kernel(fq* __restrict__ base, int ell, int nlow, TauAll ts) {
  __shared__ fq buf[2][kEqChunk];
  __shared__ fq pm[kEqMid][kEqMidVars];
  __shared__ fq pb[32];
  const size_t n = (size_t)1 << ell;
  if (threadIdx.x == 0) {
    buf[0][0] = fq_one();
    if (blockIdx.x == 0) fq_store(base + (n - 2), fq_one());  // level ell
  }
  __syncthreads();
  int cur = 0;
  for (int s = 1; s <= kEqLoVars; s++) {  // Lo by doubling from the last variable, newest variable on top
    const int k = ell - s, m = 1 << (s - 1);
    if ((int)threadIdx.x < m) {
      const fq v = buf[cur][threadIdx.x];
      const fq hi = fq_mul(v, ts.t[k]);
      buf[cur ^ 1][threadIdx.x] = fq_sub(v, hi);
      buf[cur ^ 1][m + threadIdx.x] = hi;
    }
    cur ^= 1;
    __syncthreads();
    if (blockIdx.x == 0 && (int)threadIdx.x < 2 * m)
      fq_store(base + (n - ((size_t)2 << (ell - k))) + threadIdx.x, buf[cur][threadIdx.x]);  // level k, 2m entries
  }
  const int nbits = ell - kEqLoVars - 1;  // bits of a chunk index of level 1
  const fq one = fq_one();
  if ((int)threadIdx.x < (1 << nlow)) {
    fq run = one;
    for (int j = 0; j < nlow; j++) {
      const fq& t = ts.t[ell - kEqLoVars - 1 - j];
      run = fq_mul(run, ((threadIdx.x >> j) & 1) ? t : fq_sub(one, t));
      pm[threadIdx.x][j] = run;
    }
  } else if (threadIdx.x == 64) {
    fq run = one;
    const size_t bidx = blockIdx.x;
    for (int j = nlow; j < nbits; j++) {
      const fq& t = ts.t[ell - kEqLoVars - 1 - j];
      run = fq_mul(run, ((bidx >> (j - nlow)) & 1) ? t : fq_sub(one, t));
      pb[j] = run;
    }
  }
  __syncthreads();
  const fq lo = buf[cur][threadIdx.x];
  for (int k = 1; k <= ell - kEqLoVars - 1; k++) {
    const int nb = ell - k - kEqLoVars;  // level k has 2^nb chunks
    fq* lvl = base + (n - ((size_t)2 << (ell - k))) + threadIdx.x;
    if (nb <= nlow) {
      if (blockIdx.x != 0) continue;
      for (int m = 0; m < (1 << nb); m++) fq_store(lvl + (size_t)m * kEqChunk, fq_mul(pm[m][nb - 1], lo));
    } else {
      if (((size_t)blockIdx.x >> (nb - nlow)) != 0) continue;
      const fq hb = fq_mul(pb[nb - 1], lo);
      if (nlow == 0) { fq_store(lvl + (size_t)blockIdx.x * kEqChunk, hb); continue; }
      for (int m = 0; m < (1 << nlow); m++)
        fq_store(lvl + (((size_t)blockIdx.x << nlow) + m) * kEqChunk, fq_mul(pm[m][nlow - 1], hb));
    }
  }
}

template <int K>
static int check_tabs(vpin_ctx* c, const vpin_table* const* t, size_t min_len) {
  if (!c) return VPIN_EINVAL;
  for (int k = 0; k < K; k++)
    if (!t[k] || !t[k]->d) return VPIN_EINVAL;
  for (int k = 1; k < K; k++)
    if (t[k]->len != t[0]->len) return VPIN_ESHAPE;
  if (!is_pow2(t[0]->len) || t[0]->len < min_len) return VPIN_ESHAPE;
  return VPIN_OK;
}

template <int NE>
static int finish_launch(vpin_ctx* c, int nblocks) {
  // h_out is pinned host memory mapped into the device address space: the finisher writes the
  // scalars where the host reads them after the stream sync (no memcpy node per round)
  hipLaunchKernelGGL((sc_finish_kernel<NE>), dim3(1), dim3(kBlock), 0, c->stream, c->d_partials, nblocks, c->h_out);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}

template <int K>
static int launch_eval(vpin_ctx* c, const vpin_table* const* t, int kclass) {
  int rc = check_tabs<K>(c, t, 2);
  if (rc) return rc;
  (void)hipSetDevice(c->device);
  Tabs<K> tabs;
  for (int k = 0; k < K; k++) tabs.t[k] = t[k]->d;
  size_t half = t[0]->len / 2;
  if (half <= kSmallPairs) {
    ProfScope ps(c, VPIN_K_SC_TAIL, (double)K * 32.0 * (double)t[0]->len);
    hipLaunchKernelGGL((sc_tail_kernel<K, false>), dim3(1), dim3(kSmallBlock), 0, c->stream, tabs, half, fq{}, c->h_out);
    VPIN_HIP_TRY(hipGetLastError());
    return VPIN_OK;
  }
  int grid = grid_for(half);
  {
    ProfScope ps(c, kclass, (double)K * 32.0 * (double)t[0]->len);
    hipLaunchKernelGGL((sc_eval_kernel<K>), dim3(grid), dim3(kBlock), 0, c->stream, tabs, half, c->d_partials);
  }
  VPIN_HIP_TRY(hipGetLastError());
  return finish_launch<Acc<K>::NE>(c, grid);
}

template <int K>
static int launch_bind_eval(vpin_ctx* c, vpin_table* const* t, const uint8_t* r, int kclass) {
  int rc = check_tabs<K>(c, t, 4);
  if (rc) return rc;
  if (!r) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  Tabs<K> tabs;
  for (int k = 0; k < K; k++) tabs.t[k] = t[k]->d;
  size_t len = t[0]->len, quarter = len / 4;
  if (quarter <= kSmallPairs) {
    {
      ProfScope ps(c, VPIN_K_SC_TAIL, (double)K * 32.0 * ((double)len + (double)len / 2));
      hipLaunchKernelGGL((sc_tail_kernel<K, true>), dim3(1), dim3(kSmallBlock), 0, c->stream, tabs, quarter, load_host_fq(r),
                         c->h_out);
    }
    VPIN_HIP_TRY(hipGetLastError());
    for (int k = 0; k < K; k++) t[k]->len = len / 2;
    return VPIN_OK;
  }
  int grid = grid_for(quarter);
  {
    // one read of every live element + one write of every folded element
    ProfScope ps(c, kclass, (double)K * 32.0 * ((double)len + (double)len / 2));
    hipLaunchKernelGGL((sc_bind_eval_kernel<K>), dim3(grid), dim3(kBlock), 0, c->stream, tabs, quarter,
                       load_host_fq(r), c->d_partials);
  }
  VPIN_HIP_TRY(hipGetLastError());
  for (int k = 0; k < K; k++) t[k]->len = len / 2;
  return finish_launch<Acc<K>::NE>(c, grid);
}

static int fetch(vpin_ctx* c, int ne, uint8_t* out) {
  // spin on the stream's status: a blocking wait costs ~10 us even when the round's kernel has long finished, which is
  // the usual case below ~2^18 pairs (the host's share of a ZK round is longer than the kernel)
  for (long spins = 0;; spins++) {
    const hipError_t q = hipStreamQuery(c->stream);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) { set_last_error("sum-check round: hipStreamQuery", q); return VPIN_EHIP; }
    if (spins > 4000000) { VPIN_HIP_TRY(hipStreamSynchronize(c->stream)); break; }
  }
  memcpy(out, c->h_out, (size_t)ne * sizeof(fq));
  return VPIN_OK;
}

template <int K>
static int run_eval(vpin_ctx* c, const vpin_table* const* t, uint8_t* out, int kclass) {
  int rc = launch_eval<K>(c, t, kclass);
  return rc ? rc : fetch(c, Acc<K>::NE, out);
}

template <int K>
static int run_bind_eval(vpin_ctx* c, vpin_table* const* t, const uint8_t* r, uint8_t* out, int kclass) {
  if (!out) return VPIN_EINVAL;
  int rc = launch_bind_eval<K>(c, t, r, kclass);
  return rc ? rc : fetch(c, Acc<K>::NE, out);
}

// Asynchronous halves used by the host prover to overlap a round's kernel with the previous
// round's transcript work: launch now, collect the 2-3 scalars later.  K = 4 (cubic) or 2 (quad);
// r == nullptr evaluates the tables as they are, otherwise binds with r first (fused kernel).
int sc_round_launch(vpin_ctx* c, int K, vpin_table* const* tabs, const uint8_t* r) {
  if (K == 4) return r ? launch_bind_eval<4>(c, tabs, r, VPIN_K_SC_CUBIC_FUSED) : launch_eval<4>(c, tabs, VPIN_K_SC_CUBIC);
  if (K == 2) return r ? launch_bind_eval<2>(c, tabs, r, VPIN_K_SC_QUAD_FUSED) : launch_eval<2>(c, tabs, VPIN_K_SC_QUAD);
  return VPIN_EINVAL;
}
int sc_round_wait(vpin_ctx* c, int K, uint8_t* out) { return fetch(c, K == 4 ? 3 : 2, out); }

// level k (k = 1..ell) of the suffix pyramid inside one table of 2^ell elements
static inline size_t pyramid_offset(int ell, int k) { return ((size_t)1 << ell) - ((size_t)2 << (ell - k)); }

// Phase-1 round on (Az,Bz,Cz) + the suffix table of level `level` (= round index + 1); r == nullptr
// evaluates the tables as they are, otherwise binds them with r first.  Results: the three UNSCALED
// sums  sum_i E[i]*(B_x C_x - D_x)[i]  for x = 0, 2, 3; with `lead`: the sum at x = 0, the x^2 coefficient
// sum_i E[i]*(dB dC)[i] and (first round only) the sum at x = 1.
int sc_cubic3_launch(vpin_ctx* c, vpin_table* const* t, const vpin_table* pyramid, int ell, int level, const uint8_t* r, bool lead) {
  int rc = check_tabs<3>(c, t, r ? 4 : 2);
  if (rc) return rc;
  if (!pyramid || level < 1 || level > ell || pyramid->len != ((size_t)1 << ell)) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  Tabs<3> tabs;
  for (int k = 0; k < 3; k++) tabs.t[k] = t[k]->d;
  const size_t len = t[0]->len, pairs = r ? len / 4 : len / 2;
  if (pairs != ((size_t)1 << (ell - level))) return VPIN_ESHAPE;
  const fq* E = pyramid->d + pyramid_offset(ell, level);
  const fq rr = r ? load_host_fq(r) : fq{};
  const fq_const rconst = (r && lead) ? make_fq_const(r) : fq_const{};
  // algorithmic bytes stay those of the reference formulation (4 tables): SURVEY.md 8(d)
  const double bytes = r ? 4 * 32.0 * ((double)len + (double)len / 2) : 4 * 32.0 * (double)len;
  if (pairs <= kSmallPairs) {
    {
      ProfScope ps(c, VPIN_K_SC_TAIL, bytes);
#define VPIN_T3(B_, L_) hipLaunchKernelGGL((sc_tail3_kernel<B_, L_>), dim3(1), dim3(kSmallBlock), 0, c->stream, tabs, E, pairs, rr, c->h_out)
      if (r) { if (lead) VPIN_T3(true, true); else VPIN_T3(true, false); }
      else { if (lead) VPIN_T3(false, true); else VPIN_T3(false, false); }
#undef VPIN_T3
    }
    VPIN_HIP_TRY(hipGetLastError());
  } else {
    int grid = grid_for(pairs);
    {
      ProfScope ps(c, r ? VPIN_K_SC_CUBIC_FUSED : VPIN_K_SC_CUBIC, bytes);
#define VPIN_C3(B_, L_) hipLaunchKernelGGL((sc_cubic3_kernel<B_, L_>), dim3(grid), dim3(kBlock), 0, c->stream, tabs, E, pairs, rr, rconst, c->d_partials)
      if (r) { if (lead) VPIN_C3(true, true); else VPIN_C3(true, false); }
      else { if (lead) VPIN_C3(false, true); else VPIN_C3(false, false); }
#undef VPIN_C3
    }
    VPIN_HIP_TRY(hipGetLastError());
    rc = finish_launch<3>(c, grid);
    if (rc) return rc;
  }
  if (r)
    for (int k = 0; k < 3; k++) t[k]->len = len / 2;
  return VPIN_OK;
}


// tables of live length 2 -> length 1 with r; the k (<= 4) bound values in `out` (k x 32 bytes)
int sc_final_claims(vpin_ctx* c, vpin_table* const* tables, int k, const uint8_t r[32], uint8_t* out) {
  if (!c || !tables || !r || !out || k < 1 || k > 4) return VPIN_EINVAL;
  Tabs<4> t4{};
  for (int i = 0; i < k; i++) {
    if (!tables[i] || !tables[i]->d || tables[i]->len != 2) return VPIN_ESHAPE;
    t4.t[i] = tables[i]->d;
  }
  (void)hipSetDevice(c->device);
  {
    ProfScope ps(c, VPIN_K_SC_BIND, (double)k * 96.0);
    hipLaunchKernelGGL(sc_final_kernel, dim3(1), dim3(64), 0, c->stream, t4, k, load_host_fq(r), c->h_out);
  }
  VPIN_HIP_TRY(hipGetLastError());
  for (int i = 0; i < k; i++) tables[i]->len = 1;
  return fetch(c, k, out);
}

}  // namespace vpin

using namespace vpin;

extern "C" {

int vpin_sc_cubic_round(vpin_ctx* c, const vpin_table* tau, const vpin_table* Az, const vpin_table* Bz,
                        const vpin_table* Cz, uint8_t out[96]) {
  if (!out) return VPIN_EINVAL;
  const vpin_table* t[4] = {tau, Az, Bz, Cz};
  return run_eval<4>(c, t, out, VPIN_K_SC_CUBIC);
}

int vpin_sc_quad_round(vpin_ctx* c, const vpin_table* A, const vpin_table* B, uint8_t out[64]) {
  if (!out) return VPIN_EINVAL;
  const vpin_table* t[2] = {A, B};
  return run_eval<2>(c, t, out, VPIN_K_SC_QUAD);
}

int vpin_sc_bind(vpin_ctx* c, vpin_table* const* tables, int k, const uint8_t r[32]) {
  if (!c || !tables || !r || k < 1 || k > 4) return VPIN_EINVAL;
  for (int i = 0; i < k; i++)
    if (!tables[i] || !tables[i]->d) return VPIN_EINVAL;
  for (int i = 1; i < k; i++)
    if (tables[i]->len != tables[0]->len) return VPIN_ESHAPE;
  size_t len = tables[0]->len;
  if (!is_pow2(len) || len < 2) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  size_t half = len / 2;
  int grid = grid_for(half);
  fq rr = load_host_fq(r);
  {
    ProfScope ps(c, VPIN_K_SC_BIND, (double)k * 32.0 * ((double)len + (double)half));
    // fold in groups so one launch covers up to 4 tables
    Tabs<4> t4;
    Tabs<2> t2;
    Tabs<1> t1;
    switch (k) {
      case 4:
        for (int i = 0; i < 4; i++) t4.t[i] = tables[i]->d;
        hipLaunchKernelGGL((sc_bind_kernel<4>), dim3(grid), dim3(kBlock), 0, c->stream, t4, half, rr);
        break;
      case 3:
        for (int i = 0; i < 2; i++) t2.t[i] = tables[i]->d;
        hipLaunchKernelGGL((sc_bind_kernel<2>), dim3(grid), dim3(kBlock), 0, c->stream, t2, half, rr);
        t1.t[0] = tables[2]->d;
        hipLaunchKernelGGL((sc_bind_kernel<1>), dim3(grid), dim3(kBlock), 0, c->stream, t1, half, rr);
        break;
      case 2:
        for (int i = 0; i < 2; i++) t2.t[i] = tables[i]->d;
        hipLaunchKernelGGL((sc_bind_kernel<2>), dim3(grid), dim3(kBlock), 0, c->stream, t2, half, rr);
        break;
      default:
        t1.t[0] = tables[0]->d;
        hipLaunchKernelGGL((sc_bind_kernel<1>), dim3(grid), dim3(kBlock), 0, c->stream, t1, half, rr);
    }
  }
  VPIN_HIP_TRY(hipGetLastError());
  for (int i = 0; i < k; i++) tables[i]->len = half;
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

int vpin_sc_cubic_bind_round(vpin_ctx* c, vpin_table* tau, vpin_table* Az, vpin_table* Bz, vpin_table* Cz,
                             const uint8_t r[32], uint8_t out[96]) {
  vpin_table* t[4] = {tau, Az, Bz, Cz};
  return run_bind_eval<4>(c, t, r, out, VPIN_K_SC_CUBIC_FUSED);
}

int vpin_sc_quad_bind_round(vpin_ctx* c, vpin_table* A, vpin_table* B, const uint8_t r[32], uint8_t out[64]) {
  vpin_table* t[2] = {A, B};
  return run_bind_eval<2>(c, t, r, out, VPIN_K_SC_QUAD_FUSED);
}

// All suffix tables eq(tau_{k..ell-1}, .), k = 1..ell, in one table of 2^ell elements (level k holds
// 2^(ell-k) elements at offset 2^ell - 2^(ell-k+1); the last element is unused).  Same doubling as
// EqPolynomial::evals (dense_mlpoly.rs:78-94) run from the last variable, every intermediate kept.
int vpin_eq_suffix_tables(vpin_ctx* c, const uint8_t* tau, int ell, vpin_table** out) {
  if (!c || !out || !tau || ell < 1 || ell > 40) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  vpin_table* t = nullptr;
  int rc = table_alloc_uninit(c, (size_t)1 << ell, &t);
  if (rc) return rc;
  hipError_t e = hipSuccess;
  {
    ProfScope ps(c, VPIN_K_EQ, 32.0 * 3.0 * (double)(((size_t)1 << (ell - 1)) - 1));
    static const bool fused = getenv("VPIN_EQ_STEPWISE") == nullptr;
    if (fused && ell >= kEqLoVars + 2 && ell <= 32) {
      TauAll ta;
      for (int k = 0; k < ell; k++) ta.t[k] = load_host_fq(tau + 32 * (size_t)k);
      const int nbits = ell - kEqLoVars - 1, nlow = std::min(kEqMidVars, std::max(0, nbits - 8));
      hipLaunchKernelGGL(eq_pyramid_fused_kernel, dim3((unsigned)((size_t)1 << (nbits - nlow))), dim3(kEqChunk), 0, c->stream, t->d, ell, nlow, ta);
    } else {
      // levels of up to 512 elements (k >= ell-9) in one launch, the larger ones one launch each
      const int k_lo = ell - 9 > 1 ? ell - 9 : 1;
      TauPack tp;
      for (int k = k_lo; k <= ell - 1; k++) tp.t[k - k_lo] = load_host_fq(tau + 32 * (size_t)k);
      hipLaunchKernelGGL(eq_pyramid_top_kernel, dim3(1), dim3(512), 0, c->stream, t->d, ell, k_lo, tp);
      for (int k = k_lo - 1; k >= 1; k--) {  // level k from level k+1
        size_t m = (size_t)1 << (ell - k - 1);
        hipLaunchKernelGGL(eq_pyramid_step_kernel, dim3(grid_for(m)), dim3(kBlock), 0, c->stream,
                           (const fq*)(t->d + pyramid_offset(ell, k + 1)), t->d + pyramid_offset(ell, k), m,
                           load_host_fq(tau + 32 * (size_t)k));
      }
    }
    e = hipGetLastError();
  }
  if (e != hipSuccess) { set_last_error("vpin_eq_suffix_tables", e); vpin_table_free(c, t); return VPIN_EHIP; }
  *out = t;
  return VPIN_OK;
}

// synchronous C-ABI forms of the eq-factored phase-1 round (unscaled sums, see sc_cubic3_launch)
int vpin_sc_cubic3_round(vpin_ctx* c, const vpin_table* pyramid, int ell, int level, vpin_table* Az, vpin_table* Bz,
                         vpin_table* Cz, uint8_t out[96]) {
  if (!out) return VPIN_EINVAL;
  vpin_table* t[3] = {Az, Bz, Cz};
  int rc = sc_cubic3_launch(c, t, pyramid, ell, level, nullptr, false);
  return rc ? rc : sc_round_wait(c, 4, out);
}
int vpin_sc_cubic3_bind_round(vpin_ctx* c, const vpin_table* pyramid, int ell, int level, vpin_table* Az, vpin_table* Bz,
                              vpin_table* Cz, const uint8_t r[32], uint8_t out[96]) {
  if (!out || !r) return VPIN_EINVAL;
  vpin_table* t[3] = {Az, Bz, Cz};
  int rc = sc_cubic3_launch(c, t, pyramid, ell, level, r, false);
  return rc ? rc : sc_round_wait(c, 4, out);
}

// leading-coefficient form of the same round (the kernels the prover runs when phase 1's claim is consistent):
// out = t(0) | x^2 coefficient of t | t(1) (the last only when r == NULL, i.e. the first round)
int vpin_sc_cubic3_lead_round(vpin_ctx* c, const vpin_table* pyramid, int ell, int level, vpin_table* Az, vpin_table* Bz,
                              vpin_table* Cz, const uint8_t* r, uint8_t out[96]) {
  if (!out) return VPIN_EINVAL;
  vpin_table* t[3] = {Az, Bz, Cz};
  int rc = sc_cubic3_launch(c, t, pyramid, ell, level, r, true);
  return rc ? rc : sc_round_wait(c, 4, out);
}

int vpin_eq_table(vpin_ctx* c, const uint8_t* r, int ell, vpin_table** out) {
  if (!c || !out || ell < 0 || ell > 40 || (ell > 0 && !r)) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  size_t n = (size_t)1 << ell;
  static const bool fused = getenv("VPIN_EQ_STEPWISE") == nullptr;
  if (fused && ell >= kEqLoVars + 1 && ell <= 32) {
    // one launch, every entry written once (eq_table_fused_kernel)
    vpin_table* t = nullptr;
    int rc1 = table_alloc_uninit(c, n, &t);
    if (rc1) return rc1;
    TauAll ra;
    for (int j = 0; j < ell; j++) ra.t[j] = load_host_fq(r + 32 * (size_t)j);
    {
      ProfScope ps(c, VPIN_K_EQ, 32.0 * (double)n);
      // chunks per workgroup: one while the table has at most 256 chunks (latency), up to 32 for the large ones (set-up amortised)
      const int nmid = std::min(kEqMidVars, std::max(0, ell - kEqLoVars - 8));
      hipLaunchKernelGGL(eq_table_fused_kernel, dim3((unsigned)(n >> (kEqLoVars + nmid))), dim3(kEqChunk), 0, c->stream, t->d, ell, nmid, ra);
    }
    hipError_t e1 = hipGetLastError();  // no temporaries: nothing to wait for, later work is ordered on the stream
    if (e1 != hipSuccess) { set_last_error("eq_table", e1); vpin_table_free(c, t); return VPIN_EHIP; }
    *out = t;
    return VPIN_OK;
  }
  vpin_table *a = nullptr, *b = nullptr;
  int rc = table_alloc_uninit(c, n, &a);  // every element is written by the doubling steps
  if (rc) return rc;
  if (ell > 0) {
    rc = table_alloc_uninit(c, n, &b);
    if (rc) { vpin_table_free(c, a); return rc; }
  }
  // the first k0 = min(ell, 9) steps in one launch (eq_head_kernel), then one launch per step, ping-ponging a <-> b
  const int k0 = ell < 9 ? ell : 9;
  vpin_table* src = ((ell - k0) % 2 == 0) ? a : b;  // so that the final result lands in `a`
  hipError_t e = hipSuccess;
  {
    ProfScope ps(c, VPIN_K_EQ, 32.0 * 3.0 * (double)(n - 1));
    EqHead rs;
    for (int j = 0; j < k0; j++) rs.r[j] = load_host_fq(r + 32 * (size_t)j);
    hipLaunchKernelGGL(eq_head_kernel, dim3(1), dim3(512), 0, c->stream, src->d, k0, rs);
    vpin_table* dst = (src == a) ? b : a;
    size_t prev = (size_t)1 << k0;
    for (int j = k0; j < ell; j++) {
      hipLaunchKernelGGL(eq_step_kernel, dim3(grid_for(prev)), dim3(kBlock), 0, c->stream, src->d, dst->d, prev,
                         load_host_fq(r + 32 * (size_t)j));
      prev *= 2;
      vpin_table* tmp = src; src = dst; dst = tmp;
    }
  }
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (b) vpin_table_free(c, b);
  if (e != hipSuccess) { set_last_error("eq_table", e); vpin_table_free(c, a); return VPIN_EHIP; }
  *out = a;  // src == a after an even number of swaps from the chosen start
  return VPIN_OK;
}

}  // extern "C"
