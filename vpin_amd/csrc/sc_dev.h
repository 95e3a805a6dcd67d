// sc_dev.h -- device pieces shared by the sum-check kernels (sumcheck.hip) and the SPARK kernels
// (spark.hip): per-pair evaluation stages, table folds, block/grid reductions.
#pragma once
#include <cstring>

#include "ctx.h"

namespace vpin {

#ifndef VPIN_SC_MIN_WAVES
#define VPIN_SC_MIN_WAVES 3  // 164 VGPRs, no scratch (4 would need 128 VGPRs and spill ~150 B)
#endif
constexpr int kBlock = 256;
constexpr int kMinWaves = VPIN_SC_MIN_WAVES;  // waves per SIMD asked of the register allocator
constexpr int kMaxBlocksCap = 16384;
inline int max_blocks() {
  static const int n = [] { const char* e = getenv("VPIN_SC_BLOCKS"); int v = e ? atoi(e) : 1024; return v < 1 ? 1 : v > kMaxBlocksCap ? kMaxBlocksCap : v; }();
  return n;
}

// ---- per-pair evaluation -------------------------------------------------------------

// phase-1 combiner (r1csproof.rs:104-108): A * (B*C - D)
__device__ __forceinline__ fq comb_cubic(const fq& a, const fq& b, const fq& c, const fq& d) {
  return fq_mul(a, fq_sub(fq_mul(b, c), d));
}

template <int K>
struct Acc;

template <>
struct Acc<4> {
  static constexpr int NE = 3;
  fq e[3];
  __device__ __forceinline__ void init() { e[0] = e[1] = e[2] = fq_zero(); }
  // p = low element, d = high - low of each of the 4 tables (sumcheck.rs:631-650).
  // The evaluation points are low (x=0), 2*high-low = low+2d (x=2), low+3d (x=3); p walks
  // along the line so only p[4], d[4] stay live.
  __device__ __forceinline__ void add_pair(fq* p, const fq* d) {
    e[0] = fq_add(e[0], comb_cubic(p[0], p[1], p[2], p[3]));
#pragma unroll
    for (int k = 0; k < 4; k++) p[k] = fq_add(fq_add(p[k], d[k]), d[k]);
    e[1] = fq_add(e[1], comb_cubic(p[0], p[1], p[2], p[3]));
#pragma unroll
    for (int k = 0; k < 4; k++) p[k] = fq_add(p[k], d[k]);
    e[2] = fq_add(e[2], comb_cubic(p[0], p[1], p[2], p[3]));
  }
  // Staged form with a smaller live set: u[x] = B_x*C_x - D_x is built table by table, A is
  // loaded last.  Each stage takes (p,d) of ONE table; the caller loads/folds that table just
  // before the call, so at most two tables' values are live next to u[3] and e[3].
  __device__ __forceinline__ void stage_bc(fq* u, fq pb, const fq& db, fq pc, const fq& dc) {
    u[0] = fq_mul(pb, pc);
    pb = fq_add(fq_add(pb, db), db); pc = fq_add(fq_add(pc, dc), dc);
    u[1] = fq_mul(pb, pc);
    pb = fq_add(pb, db); pc = fq_add(pc, dc);
    u[2] = fq_mul(pb, pc);
  }
  __device__ __forceinline__ void stage_d(fq* u, fq pd, const fq& dd) {
    u[0] = fq_sub(u[0], pd);
    pd = fq_add(fq_add(pd, dd), dd);
    u[1] = fq_sub(u[1], pd);
    pd = fq_add(pd, dd);
    u[2] = fq_sub(u[2], pd);
  }
  // eq-factored form: the folded eq(tau,.) table is a per-round scalar times the suffix table E, so
  // A_x = c_x * E[i]; the kernel accumulates sum_i E[i]*u_x[i] and the host applies c_x
  __device__ __forceinline__ void stage_e(const fq* u, const fq& E) {
    e[0] = fq_add(e[0], fq_mul(E, u[0]));
    e[1] = fq_add(e[1], fq_mul(E, u[1]));
    e[2] = fq_add(e[2], fq_mul(E, u[2]));
  }
  // Leading-coefficient form.  With the eq factor split off, the round polynomial is l(x) * t(x) with l linear and
  // t(x) = sum_i E[i] u_x[i] quadratic, so t(0), the x^2 coefficient of t and the round's claim (which fixes t(1))
  // determine it: e[0] += E * u(0), e[1] += E * dB*dC -- two products and no p+2d / p+3d chains per pair instead
  // of three.  `first` rounds (claim not yet known to be consistent) also return e[2] += E * u(1).
  __device__ __forceinline__ void lead_bc(const fq& pb, const fq& db, const fq& pc, const fq& dc, const fq& E) {
    e[0] = fq_add(e[0], fq_mul(E, fq_mul(pb, pc)));
    e[1] = fq_add(e[1], fq_mul(E, fq_mul(db, dc)));
  }
  __device__ __forceinline__ void lead_bcd(const fq& pb, const fq& db, const fq& pc, const fq& dc, const fq& pd, const fq& E) {
    e[0] = fq_add(e[0], fq_mul(E, fq_sub(fq_mul(pb, pc), pd)));
    e[1] = fq_add(e[1], fq_mul(E, fq_mul(db, dc)));
  }
  __device__ __forceinline__ void lead_one(const fq& pb, const fq& db, const fq& pc, const fq& dc, const fq& pd, const fq& dd,
                                           const fq& E) {
    e[2] = fq_add(e[2], fq_mul(E, fq_sub(fq_mul(fq_add(pb, db), fq_add(pc, dc)), fq_add(pd, dd))));
  }
  __device__ __forceinline__ void stage_a(const fq* u, fq pa, const fq& da) {
    e[0] = fq_add(e[0], fq_mul(pa, u[0]));
    pa = fq_add(fq_add(pa, da), da);
    e[1] = fq_add(e[1], fq_mul(pa, u[1]));
    pa = fq_add(pa, da);
    e[2] = fq_add(e[2], fq_mul(pa, u[2]));
  }
};

// The two sums of the leading-coefficient rounds (lead_bc / lead_bcd above) with the products accumulated unreduced
// (fq_dev.h fqw_mac): e[0] += E * X, e[1] += E * Y, reduced every seven pairs and once at the end.
struct LeadAcc {
  fq_wide w[2];
  int n;
  __device__ __forceinline__ void init() { fqw_zero(w[0]); fqw_zero(w[1]); n = 0; }
  __device__ __forceinline__ void add(fq* e, const fq& E, const fq& X, const fq& Y) {
#ifdef VPIN_NO_LAZY_ACC  // A/B builds: every product reduced at once, as before round 4
    e[0] = fq_add(e[0], fq_mul(E, X));
    e[1] = fq_add(e[1], fq_mul(E, Y));
#else
    fqw_mac(w[0], E, X);
    fqw_mac(w[1], E, Y);
#ifndef VPIN_LAZY_NO_FLUSH  // (tools/isa_counts.py compiles the loop once without the flush to price the two apart)
    if (++n == 7) flush(e);
#endif
#endif
  }
  __device__ __forceinline__ void flush(fq* e) {
    if (n == 0) return;
    e[0] = fq_add(e[0], fqw_reduce(w[0]));
    e[1] = fq_add(e[1], fqw_reduce(w[1]));
    fqw_zero(w[0]); fqw_zero(w[1]);
    n = 0;
  }
};

template <>
struct Acc<2> {
  static constexpr int NE = 2;
  fq e[2];
  __device__ __forceinline__ void init() { e[0] = e[1] = fq_zero(); }
  // sumcheck.rs:460-469, comb A*B (r1csproof.rs:139-140)
  __device__ __forceinline__ void add_pair(fq* p, const fq* d) {
    e[0] = fq_add(e[0], fq_mul(p[0], p[1]));
#pragma unroll
    for (int k = 0; k < 2; k++) p[k] = fq_add(fq_add(p[k], d[k]), d[k]);
    e[1] = fq_add(e[1], fq_mul(p[0], p[1]));
  }
};

template <int K>
struct Tabs {
  fq* t[K];
};

// block-level reduction of NE accumulators; thread e < NE of the block writes partial e
template <int NE>
__device__ __forceinline__ void block_reduce_store(fq* e, fq* __restrict__ partials) {
  __shared__ fq sh[kBlock / 64][NE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NE; k++) {
    fq s = fq_wave_sum(e[k]);
    if (lane == 0) sh[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NE) {
    fq s = sh[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kBlock / 64; w++) s = fq_add(s, sh[w][threadIdx.x]);
    fq_store(&partials[(size_t)blockIdx.x * NE + threadIdx.x], s);
  }
}

// (p, d) of one table's pair (i, i+half), unfolded
__device__ __forceinline__ void load_pd(const fq* __restrict__ t, size_t i, size_t half, fq& p, fq& d) {
  p = fq_load(t + i);
  d = fq_sub(fq_load(t + half + i), p);
}
// fold one table's two pairs with r (dense_mlpoly.rs:232), store the folded values, return (p, d)
__device__ __forceinline__ void fold_pd(fq* __restrict__ t, size_t i, size_t quarter, const fq& r, fq& p, fq& d) {
  const size_t half = 2 * quarter;
  fq a0 = fq_load(t + i), a1 = fq_load(t + half + i);
  fq b0 = fq_load(t + quarter + i), b1 = fq_load(t + half + quarter + i);
  p = fq_add(a0, fq_mul(r, fq_sub(a1, a0)));
  fq hi = fq_add(b0, fq_mul(r, fq_sub(b1, b0)));
  fq_store(t + i, p);
  fq_store(t + quarter + i, hi);
  d = fq_sub(hi, p);
}

// fold with the launch-wide constant form of r (fq_dev.h fq_mul_const; tt = the constants in LDS)
__device__ __forceinline__ void fold_pd_c(fq* __restrict__ t, size_t i, size_t quarter, const uint32_t (*tt)[8], fq& p, fq& d) {
  const size_t half = 2 * quarter;
  fq a0 = fq_load(t + i), a1 = fq_load(t + half + i);
  fq b0 = fq_load(t + quarter + i), b1 = fq_load(t + half + quarter + i);
  p = fq_add(a0, fq_mul_const(fq_sub(a1, a0), tt));
  fq hi = fq_add(b0, fq_mul_const(fq_sub(b1, b0), tt));
  fq_store(t + i, p);
  fq_store(t + quarter + i, hi);
  d = fq_sub(hi, p);
}

// Sum nblocks x NE block partials into out[NE].
template <int NE>
__global__ __launch_bounds__(kBlock) void sc_finish_kernel(const fq* __restrict__ partials, int nblocks,
                                                           fq* __restrict__ out) {
  fq e[NE];
#pragma unroll
  for (int k = 0; k < NE; k++) e[k] = fq_zero();
  for (int b = threadIdx.x; b < nblocks; b += kBlock)
#pragma unroll
    for (int k = 0; k < NE; k++) e[k] = fq_add(e[k], fq_load(&partials[(size_t)b * NE + k]));
  __shared__ fq sh[kBlock / 64][NE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NE; k++) {
    fq s = fq_wave_sum(e[k]);
    if (lane == 0) sh[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NE) {
    fq s = sh[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kBlock / 64; w++) s = fq_add(s, sh[w][threadIdx.x]);
    fq_store(&out[threadIdx.x], s);
  }
}

static inline int grid_for(size_t work) {
  size_t b = (work + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > (size_t)max_blocks()) b = max_blocks();
  return (int)b;
}

// T_i = r~ * 2^(32 i) * 2^-256 mod q for the Montgomery image r~ in `p` (host arithmetic, once per launch), transposed
fq_const make_fq_const(const uint8_t* p);

static inline fq load_host_fq(const uint8_t* p) {
  fq r;
  memcpy(r.v, p, 32);
  return r;
}

}  // namespace vpin
