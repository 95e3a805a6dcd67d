// msm.hip -- fixed-base multi-scalar multiplication over the Pedersen generators, gfx950.
//
// Replaces the reference's polynomial-commitment MSMs:
//   DensePolynomial::commit_inner   Spartan/src/dense_mlpoly.rs:160-175  (L row commitments)
//   Commitments for [Scalar]        Spartan/src/commitments.rs:93-98     (MSM + blind*h)
//   vartime_multiscalar_mul         Spartan/src/group.rs:103-122         (dalek Straus/Pippenger)
//
// MI355X-first design: every MSM of the sat proof is over the SAME generator stream
// g[0..R+2) (MultiCommitGens::new, commitments.rs:20-38), so instead of per-row Pippenger
// buckets we spend HBM (288 GB) on a window table  T[w][j][k] = (k+1) * 2^(c*w) * g_j  of
// AFFINE points (y+x, y-x, 2dxy; 96 B) with c = 11 or 12 bit signed windows (W = 24 / 22
// windows, 1024 / 2048 multiples each: 2.4-4.3 MB per generator, 9 GB for R = 2048, 19 GB for
// R = 8192) and turn each row commitment into a pure gather-and-add: a non-zero scalar costs
// at most W table additions of 7 field multiplies, no doublings, no bucket reduction, no atomics.  Zero scalars and zero
// digits are skipped (the witness is ~40% zero padding and full of 0/1 bits), like dalek's
// vartime MSM.  One 256-thread workgroup per row; per-thread partial sums are combined by
// an LDS tree.  Integer-ALU bound (~8 field multiplies per table add); no MFMA.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "ctx.h"
#include "fp_dev.h"
#include "fp10_dev.h"
#include "ge_tree_dev.h"
#include "mailbox_dev.h"

#ifndef VPIN_NIELS_SLOT
// bytes per table entry: 96 (packed; the default) or 128 (one entry per 128-byte line: a gather is one aligned line, never two
// half-used ones).  Round 6 re-measured the pair with the spatial split and the strip kernel in place, both libraries built on the box
// and the 16386 generators of the derefs commitments at 12-bit windows in both (VPIN_SPARK_GENS_BUDGET_GB=100,
// VPIN_GENS_FREE_FRACTION=0.45): LeNet step 358.4 against 366.3 ms (-2.2 %, 12 W less) for 109 instead of 95 GiB of tables
// (profiles/r06_ab_slot128.txt).  NOT the default: with 14 GiB less headroom a 2^25 proof split over two rank-threads of ONE GPU
// (tests, rehearsals) runs out of memory once anything else is resident, and that out-of-memory hipMalloc took the process down
// inside the HIP runtime instead of returning an error.  Build with -DVPIN_NIELS_SLOT=128 (and the two budgets above) to trade
// the memory for the time on a part that proves single-GPU traces only.
#define VPIN_NIELS_SLOT 96
#endif
static_assert(VPIN_NIELS_SLOT == 96 || VPIN_NIELS_SLOT == 128, "table entry size");
// Round 6: the slot size is a property of a TABLE (decided when it is built: VPIN_TABLE_SLOT = 96 | 128, default VPIN_NIELS_SLOT), not
// of the build: an entry is three field elements (y+x, y-x, 2dxy) at the start of a slot of s32 = 3 or 4 32-byte units.

struct vpin_gens {
  // Two segments: bases [0, split) in `table` with wide windows (the generators every large commitment of a proof
  // walks), bases [split, nbt) in `table_hi` with narrow ones (the upper half of the 32k-generator stream, touched by
  // SNARK::encode and one evaluation proof, and the prefix sums).  split == nbt: a single segment.
  vpin::fp* table = nullptr;        // [W][split][E] slots of s32 field elements
  vpin::fp* table_hi = nullptr;     // [W_hi][nbt - split][E_hi]
  int s32 = VPIN_NIELS_SLOT / 32;   // 32-byte units per slot: 3 (96 B, packed) or 4 (128 B, a line per entry)
  size_t nb = 0;                    // number of bases in the stream
  size_t nbt = 0;                   // bases in the table: the stream, then the prefix sums S_k = g_0 + ... + g_{2^k - 1}
  size_t split = 0;
  int c = 12, W = 22, E = 2048;     // window bits, windows, entries per window (= 2^(c-1)) of the first segment
  int c_hi = 8, W_hi = 32, E_hi = 128;
};

namespace vpin {

struct TableView {
  const fp* t;     // first segment
  const fp* t_hi;  // second segment
  size_t split, nbt;       // bases [0, split) | [split, nbt)
  int c, W, E, c_hi, W_hi, E_hi;
  size_t sum0;  // table index of S_0; S_k = g_0 + ... + g_{2^k - 1} sits at sum0 + k
  int s32;      // field elements per slot (3 or 4)
};

// the segment of base j: table, index inside it, bases per window, window parameters
struct TableSeg {
  const fp* t;
  size_t j, nb;
  int c, W, E, s32;
  // entry d (multiple d + 1) of window w of this base
  __device__ __forceinline__ const fp* entry(int w, uint32_t d) const { return t + (((size_t)w * nb + j) * (size_t)E + d) * (size_t)s32; }
};
__device__ __forceinline__ TableSeg table_seg(const TableView& tv, size_t j) {
  if (j < tv.split) return TableSeg{tv.t, j, tv.split, tv.c, tv.W, tv.E, tv.s32};
  return TableSeg{tv.t_hi, j - tv.split, tv.nbt - tv.split, tv.c_hi, tv.W_hi, tv.E_hi, tv.s32};
}

// ---- table construction ---------------------------------------------------------------

// Prefix sums of the stream over power-of-two lengths, kept in the table as extra bases: a commitment row whose
// scalars are all one value s (the padding tails of the SPARK polynomials are runs of a single eq value, a
// quarter of the derefs polynomial for vPIN's instances) is s * S_k, one scalar multiplication instead of 2^k.
// Block k sums g_j over [2^(k-1), 2^k) (block 0: g_0) into xyzt[nb + k]; gens_sum_scan_kernel then accumulates.
__global__ __launch_bounds__(256) void gens_sum_kernel(fp* __restrict__ xyzt, size_t nb) {
  const int k = blockIdx.x;
  const size_t lo = k ? ((size_t)1 << (k - 1)) : 0, hi = (size_t)1 << k;
  ge_ext acc = ge_identity();
  for (size_t j = lo + threadIdx.x; j < hi && j < nb; j += 256) {
    ge_ext p;
    p.X = fp_load(xyzt + 4 * j); p.Y = fp_load(xyzt + 4 * j + 1); p.Z = fp_load(xyzt + 4 * j + 2); p.T = fp_load(xyzt + 4 * j + 3);
    acc = ge_add(acc, p);
  }
  __shared__ ge_ext sh[256];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) {
      acc = ge_add(acc, sh[threadIdx.x + st]);
      sh[threadIdx.x] = acc;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    fp* o = xyzt + 4 * (nb + k);
    fp_store(o, acc.X); fp_store(o + 1, acc.Y); fp_store(o + 2, acc.Z); fp_store(o + 3, acc.T);
  }
}
__global__ void gens_sum_scan_kernel(fp* __restrict__ xyzt, size_t nb, int nsum) {
  if (threadIdx.x || blockIdx.x) return;
  ge_ext acc;
  fp* o = xyzt + 4 * nb;
  acc.X = fp_load(o); acc.Y = fp_load(o + 1); acc.Z = fp_load(o + 2); acc.T = fp_load(o + 3);
  for (int k = 1; k < nsum; k++) {
    fp* q = xyzt + 4 * (nb + k);
    ge_ext d;
    d.X = fp_load(q); d.Y = fp_load(q + 1); d.Z = fp_load(q + 2); d.T = fp_load(q + 3);
    acc = ge_add(acc, d);
    fp_store(q, acc.X); fp_store(q + 1, acc.Y); fp_store(q + 2, acc.Z); fp_store(q + 3, acc.T);
  }
}

// one thread per base: shifts[j][w] = 2^(c*w) * g_j
__global__ __launch_bounds__(64) void gens_shift_kernel(const fp* __restrict__ xyzt, size_t nb, int W, int c,
                                                        ge_ext* __restrict__ shifts) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nb) return;
  ge_ext p;
  p.X = fp_load(xyzt + 4 * j); p.Y = fp_load(xyzt + 4 * j + 1); p.Z = fp_load(xyzt + 4 * j + 2); p.T = fp_load(xyzt + 4 * j + 3);
  for (int w = 0; w < W; w++) {
    ge_ext* o = shifts + j * W + w;
    fp_store(&o->X, p.X); fp_store(&o->Y, p.Y); fp_store(&o->Z, p.Z); fp_store(&o->T, p.T);
    for (int k = 0; k < c; k++) p = ge_double(p);
  }
}

__device__ __forceinline__ ge_niels niels_load(const fp* o) {
  ge_niels e;
  e.ypx = fp_load(o); e.ymx = fp_load(o + 1); e.xy2d = fp_load(o + 2);
  return e;
}

// one thread per (base, window): the E multiples of the shifted base, normalised to affine with
// one inversion per thread (Montgomery's trick): forward pass stores X,Y,Z in the entry slots and
// the running product of the Z's in `prefix`; the backward pass peels the inverses off.
__global__ __launch_bounds__(64) void gens_table_kernel(const ge_ext* __restrict__ shifts, size_t nb, int W, int E,
                                                        fp* __restrict__ table, int s32, fp* __restrict__ prefix) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nb * (size_t)W) return;
  size_t j = idx / W;
  int w = (int)(idx % W);
  const ge_ext* s = shifts + j * W + w;
  ge_ext p;
  p.X = fp_load(&s->X); p.Y = fp_load(&s->Y); p.Z = fp_load(&s->Z); p.T = fp_load(&s->T);
  ge_cached pc = ge_to_cached(p);
  fp* out = table + (((size_t)w * nb + j) * E) * (size_t)s32;  // slot k at out + k * s32: {ypx, ymx, xy2d}
  fp* pre = prefix + ((size_t)w * nb + j) * E;
  ge_ext q = p;
  fp run = fp_one();
  for (int k = 0; k < E; k++) {
    fp_store(out + (size_t)k * s32, q.X); fp_store(out + (size_t)k * s32 + 1, q.Y); fp_store(out + (size_t)k * s32 + 2, q.Z);
    run = fp_mul(run, q.Z);
    fp_store(pre + k, run);
    if (k + 1 < E) q = ge_add_cached(q, pc);
  }
  fp acc = fp_invert(run);
  const fp d2 = FP_D2();
  for (int k = E - 1; k >= 0; k--) {
    fp X = fp_load(out + (size_t)k * s32), Y = fp_load(out + (size_t)k * s32 + 1), Z = fp_load(out + (size_t)k * s32 + 2);
    fp zinv = (k > 0) ? fp_mul(acc, fp_load(pre + k - 1)) : acc;
    acc = fp_mul(acc, Z);
    fp x = fp_mul(X, zinv), y = fp_mul(Y, zinv);
    fp_store(out + (size_t)k * s32, fp_add(y, x));
    fp_store(out + (size_t)k * s32 + 1, fp_sub(y, x));
    fp_store(out + (size_t)k * s32 + 2, fp_mul(fp_mul(x, y), d2));
  }
}

// ---- scalar handling --------------------------------------------------------------------

__device__ __forceinline__ bool fq_same(const fq& a, const fq& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

// accumulate s * g_j into acc through the window table; s canonical, non-zero
// s -> q - s when s is in the upper half (bit 251 or 252 set): s*g = -((q - s)*g).  Nothing for a random scalar,
// but the R1CS values -1, -2 (a fifth of comb_ops' val slices) become one table add instead of W.
__device__ __forceinline__ bool fq_fold_sign(fq& s) {
  if (s.v[7] < 0x08000000u) return false;
  fq d;
  unsigned bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(fq_modulus_limb(i), s.v[i], bw, &bw);
  s = d;
  return true;
}

__device__ __forceinline__ void table_mul_acc(ge_ext& acc, fq s, const TableView& tv, size_t j) {
  const bool flip = fq_fold_sign(s);
  const TableSeg sg = table_seg(tv, j);
  uint32_t carry = 0;
  const uint32_t mask = (1u << sg.c) - 1u, half = 1u << (sg.c - 1);
  // software pipeline: the (random, 96-byte) table entry of window w+1 is requested before the
  // 7-multiply add of window w, so the gather latency hides behind arithmetic
  ge_niels e_cur;
  bool have_cur = false, neg_cur = false;
#pragma unroll 1
  for (int w = 0; w <= sg.W; w++) {
    ge_niels e_next;
    bool have_next = false, neg_next = false;
    if (w < sg.W) {
      uint32_t v = (s.v[0] & mask) + carry;
      // shift the 256-bit scalar right by c bits (static register indices only)
#pragma unroll
      for (int i = 0; i < 7; i++) s.v[i] = __builtin_amdgcn_alignbit(s.v[i + 1], s.v[i], sg.c);
      s.v[7] >>= sg.c;
      neg_next = v > half;
      uint32_t mag = neg_next ? (mask + 1u) - v : v;
      carry = neg_next ? 1u : 0u;
      neg_next ^= flip;
      if (mag != 0) {
        e_next = niels_load(sg.entry(w, mag - 1));
        have_next = true;
      }
    }
    if (have_cur) acc = ge_add_niels(acc, e_cur, neg_cur);
    e_cur = e_next;
    have_cur = have_next;
    neg_cur = neg_next;
  }
}

// The same walk with the accumulator in the ten-limb form (fp10_dev.h): the product of two field elements is 142 VALU
// instructions instead of 197 and the additions between products carry nothing.  Used by the row-commitment kernels, where
// a lane adds hundreds of table entries to one accumulator; the entries stay in the packed 96-byte form in HBM.
__device__ __forceinline__ void table_mul_acc10(ge10& acc, fq s, const TableView& tv, size_t j) {
  const bool flip = fq_fold_sign(s);
  const TableSeg sg = table_seg(tv, j);
  uint32_t carry = 0;
  const uint32_t mask = (1u << sg.c) - 1u, half = 1u << (sg.c - 1);
  ge_niels e_cur;
  bool have_cur = false, neg_cur = false;
#pragma unroll 1
  for (int w = 0; w <= sg.W; w++) {
    ge_niels e_next;
    bool have_next = false, neg_next = false;
    if (w < sg.W) {
      uint32_t v = (s.v[0] & mask) + carry;
#pragma unroll
      for (int i = 0; i < 7; i++) s.v[i] = __builtin_amdgcn_alignbit(s.v[i + 1], s.v[i], sg.c);
      s.v[7] >>= sg.c;
      neg_next = v > half;
      uint32_t mag = neg_next ? (mask + 1u) - v : v;
      carry = neg_next ? 1u : 0u;
      neg_next ^= flip;
      if (mag != 0) {
        e_next = niels_load(sg.entry(w, mag - 1));
        have_next = true;
      }
    }
    if (have_cur) acc = ge10_add_niels(acc, e_cur, neg_cur);
    e_cur = e_next;
    have_cur = have_next;
    neg_cur = neg_next;
  }
}

// accumulator type of the row kernels: TEN = ten-limb form (default), false = the eight-limb form (A/B runs: VPIN_MSM_FP8)
template <bool TEN> struct RowAcc;
template <> struct RowAcc<false> {
  ge_ext a = ge_identity();
  __device__ __forceinline__ void mul_acc(const fq& s, const TableView& tv, size_t j) { table_mul_acc(a, s, tv, j); }
  __device__ __forceinline__ void add_cached(const ge_cached& q) { a = ge_add_cached(a, q); }
  __device__ __forceinline__ ge_ext ext() const { return a; }
};
template <> struct RowAcc<true> {
  ge10 a = ge10_identity();
  __device__ __forceinline__ void mul_acc(const fq& s, const TableView& tv, size_t j) { table_mul_acc10(a, s, tv, j); }
  __device__ __forceinline__ void add_cached(const ge_cached& q) { a = ge10_add_cached(a, q); }
  __device__ __forceinline__ ge_ext ext() const { return ge10_to_ext(a); }
};

// digit w (c bits) of a canonical scalar, for the kernels that split one scalar over several lanes
__device__ __forceinline__ uint32_t scalar_digit(const fq& s, int w, int c) {
  int off = w * c, word = off >> 5, sh = off & 31;
  uint32_t lo = s.v[word], hi = (word + 1 < 8) ? s.v[word + 1] : 0u;
  uint64_t both = ((uint64_t)hi << 32) | lo;
  return (uint32_t)(both >> sh) & ((1u << c) - 1u);
}

// signed-digit carry into window w0: decided by the nearest lower digit that is not exactly 2^(c-1)
__device__ __forceinline__ uint32_t carry_into(const fq& s, int w0, int c) {
  const uint32_t half = 1u << (c - 1);
  for (int w = w0 - 1; w >= 0; w--) {
    uint32_t d = scalar_digit(s, w, c);
    if (d != half) return d > half ? 1u : 0u;
  }
  return 0u;
}

// windows [w0, w1) of s * g_j (sg = table_seg(tv, j))
__device__ __forceinline__ void table_mul_acc_range(ge_ext& acc, const fq& s_in, const TableSeg& sg, int w0, int w1) {
  fq s = s_in;
  const bool flip = fq_fold_sign(s);
  uint32_t carry = carry_into(s, w0, sg.c);
  const uint32_t full = 1u << sg.c, half = 1u << (sg.c - 1);
#pragma unroll 1
  for (int w = w0; w < w1; w++) {
    uint32_t v = scalar_digit(s, w, sg.c) + carry;
    bool neg = v > half;
    uint32_t mag = neg ? full - v : v;
    carry = neg ? 1u : 0u;
    if (mag != 0) acc = ge_add_niels(acc, niels_load(sg.entry(w, mag - 1)), neg != flip);
  }
}

// ---- row-batched commitment -------------------------------------------------------------

constexpr int kMsmBlock = 256;
// extra dynamic LDS of the row-commitment kernels when other contexts prove on the device (vpin_ctx_set_shared_device): with
// it one workgroup per CU instead of three (a wave per SIMD), the rest of every CU stays with the other lanes' kernels
constexpr unsigned kSharedPad = 40000u;
// rows x (ncols scalars from Z with row stride `stride`) + optional extra scalars on bases
// [extra_base0, extra_base0 + n_extra).  out[row] = sum_j s[row][j] * g_j  (extended coords)
constexpr int kSeg = 8192;  // scalars per compaction segment (uint16 indices, 16 KiB of LDS)

template <bool TEN>
__global__ __launch_bounds__(kMsmBlock, 3) void msm_rows_kernel(const fq* __restrict__ Z, size_t stride, size_t ncols,
                                                             const fq* __restrict__ extra, int n_extra, size_t extra_base0,
                                                             TableView tv, ge_ext* __restrict__ out) {
  const size_t row = blockIdx.x;
  RowAcc<TEN> acc;
  const size_t total = ncols + (size_t)n_extra;
  // A row of one repeated scalar s: s * (g_0 + ... + g_{ncols-1}) from the prefix-sum base (see gens_sum_kernel).
  // Three probes keep ordinary rows from paying for the full comparison pass.
  if (ncols >= 256 && (ncols & (ncols - 1)) == 0 && ((size_t)1 << (tv.nbt - tv.sum0 - 1)) >= ncols) {
    const fq* zr = Z + row * stride;
    const fq first = fq_load(zr);
    if (fq_same(first, fq_load(zr + 1)) && fq_same(first, fq_load(zr + ncols / 2)) && fq_same(first, fq_load(zr + ncols - 1))) {
      int same = 1;
      for (size_t j = threadIdx.x; j < ncols; j += kMsmBlock) same &= fq_same(first, fq_load(zr + j)) ? 1 : 0;
      if (__syncthreads_and(same)) {
        if (threadIdx.x == 0) {
          if (blockIdx.y == 0) {  // with column chunks every chunk sees the same row: the first one owns it, the others add nothing
            if (!fq_is_zero(first)) acc.mul_acc(fq_from_mont(first), tv, tv.sum0 + (size_t)(63 - __builtin_clzll((unsigned long long)ncols)));
            for (int e = 0; e < n_extra; e++) {
              const fq x = fq_load(extra + row * (size_t)n_extra + e);
              if (!fq_is_zero(x)) acc.mul_acc(fq_from_mont(x), tv, extra_base0 + e);
            }
          }
          ge_ext* o = out + row * gridDim.y + blockIdx.y;
          const ge_ext av = acc.ext();
          fp_store(&o->X, av.X); fp_store(&o->Y, av.Y); fp_store(&o->Z, av.Z); fp_store(&o->T, av.T);
        }
        return;
      }
    }
  }
  // gridDim.y column chunks per row (few-row MSMs need more than `rows` workgroups)
  const size_t per = (total + gridDim.y - 1) / gridDim.y;
  const size_t j0 = (size_t)blockIdx.y * per, j1 = (j0 + per < total) ? j0 + per : total;
  // The witness is ~40% zeros in irregular positions; lanes that met a zero scalar would idle
  // while their wave-mates do 32 table adds.  So each segment is first compacted: the indices
  // of its non-zero scalars go to an LDS list (order is irrelevant: the group is commutative),
  // then the threads stride over the dense list.
  __shared__ uint16_t nz_list[kSeg];
  __shared__ uint32_t nz_count;
  for (size_t seg = j0; seg < j1; seg += kSeg) {
    const size_t seg_end = (seg + kSeg < j1) ? seg + kSeg : j1;
    if (threadIdx.x == 0) nz_count = 0;
    __syncthreads();
    for (size_t j = seg + threadIdx.x; j < seg_end; j += kMsmBlock) {
      const fq* sp = (j < ncols) ? (Z + row * stride + j) : (extra + row * (size_t)n_extra + (j - ncols));
      if (!fq_is_zero(fq_load(sp))) nz_list[atomicAdd(&nz_count, 1u)] = (uint16_t)(j - seg);
    }
    __syncthreads();
    const uint32_t cnt = nz_count;
    for (uint32_t k = threadIdx.x; k < cnt; k += kMsmBlock) {
      const size_t j = seg + nz_list[k];
      fq s;
      size_t base;
      if (j < ncols) { s = fq_load(Z + row * stride + j); base = j; }
      else { s = fq_load(extra + row * (size_t)n_extra + (j - ncols)); base = extra_base0 + (j - ncols); }
      acc.mul_acc(fq_from_mont(s), tv, base);
    }
    __syncthreads();
  }
  __shared__ ge_ext sh[kMsmBlock];
  sh[threadIdx.x] = acc.ext();
  __syncthreads();
  ge_tree_quad(sh, kMsmBlock);
  if (threadIdx.x == 0) {
    const ge_ext av = sh[0];
    ge_ext* o = out + row * gridDim.y + blockIdx.y;
    fp_store(&o->X, av.X); fp_store(&o->Y, av.Y); fp_store(&o->Z, av.Z); fp_store(&o->T, av.T);
  }
}

// ---- row commitment with one repeated scalar per row taken out (the derefs commitment) ---------------------------------
// Derefs (sparse_mlpoly.rs:267-283): the col-derefs vector of matrix m is E_ry[col_m[i]], and in an R1CS the column of the
// constant 1 carries a large share of a matrix's entries (37 % of B, 8 % of C for vPIN's point-mult gadget: one in twelve
// scalars of the whole derefs polynomial).  All those entries hold the SAME scalar v = E_ry[hot]: with T_j = v * g_j
// computed once per proof (R table walks), an entry at column j of a row costs one point addition instead of a table walk
// of W.  The entries are recognised by their column INDEX (the decommitment's index slices), never by value.
struct HotRows {
  const uint32_t* idx[3];   // col index slice of matrix m (N entries, the order of the polynomial)
  uint32_t hot[3];          // the hot column of matrix m, 0xffffffff: none
  const ge_cached* T[3];    // T[m][j] = E_ry[hot[m]] * g_j, j < ncols
  size_t row0, rows_per_vec;  // matrix m's vector occupies rows [row0 + m*rows_per_vec, row0 + (m+1)*rows_per_vec)
};

// T[j] = v * g_j for j < n (*vp = v in Montgomery form, read on the device: no round trip), as cached points
__global__ __launch_bounds__(64) void scalar_times_bases_kernel(const fq* __restrict__ vp, size_t n, TableView tv,
                                                                ge_cached* __restrict__ T) {
  const size_t j = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (j >= n) return;
  const fq v = fq_load(vp);
  ge_ext acc = ge_identity();
  if (!fq_is_zero(v)) table_mul_acc(acc, fq_from_mont(v), tv, j);
  const ge_cached cch = ge_to_cached(acc);
  fp_store(&T[j].YpX, cch.YpX); fp_store(&T[j].YmX, cch.YmX); fp_store(&T[j].Z, cch.Z); fp_store(&T[j].T2d, cch.T2d);
}

template <bool TEN>
__global__ __launch_bounds__(kMsmBlock, 3) void msm_rows_hot_kernel(const fq* __restrict__ Z, size_t stride, size_t ncols, TableView tv,
                                                                 HotRows hr, ge_ext* __restrict__ out, size_t row_base, size_t row_step,
                                                                 int z_compact, const uint8_t* __restrict__ strip_flag) {
  const size_t row = row_base + (size_t)blockIdx.x * row_step;  // row of the polynomial; out[] is indexed by blockIdx.x
  // z_compact: Z holds only this launch's rows, densely (a rank's rows of a split commitment gathered on their own)
  const fq* zr = Z + (z_compact ? (size_t)blockIdx.x : row) * stride;
  RowAcc<TEN> acc;
  // strip_flag[row of this launch] != 0: the table walks of this row are msm_strip_kernel's; only its hot entries are added here
  const bool hot_only = strip_flag && strip_flag[blockIdx.x];
  // a row of one repeated scalar (padding tails): s * (g_0 + ... + g_{ncols-1}), as in msm_rows_kernel
  if (!hot_only && ncols >= 256 && (ncols & (ncols - 1)) == 0 && ((size_t)1 << (tv.nbt - tv.sum0 - 1)) >= ncols) {
    const fq first = fq_load(zr);
    if (fq_same(first, fq_load(zr + 1)) && fq_same(first, fq_load(zr + ncols / 2)) && fq_same(first, fq_load(zr + ncols - 1))) {
      int same = 1;
      for (size_t j = threadIdx.x; j < ncols; j += kMsmBlock) same &= fq_same(first, fq_load(zr + j)) ? 1 : 0;
      if (__syncthreads_and(same)) {
        if (threadIdx.x == 0) {
          // with column chunks every chunk sees the same row: the first one owns it, the others add nothing
          if (blockIdx.y == 0 && !fq_is_zero(first)) acc.mul_acc(fq_from_mont(first), tv, tv.sum0 + (size_t)(63 - __builtin_clzll((unsigned long long)ncols)));
          ge_ext* o = out + (size_t)blockIdx.x * gridDim.y + blockIdx.y;
          const ge_ext av = acc.ext();
          fp_store(&o->X, av.X); fp_store(&o->Y, av.Y); fp_store(&o->Z, av.Z); fp_store(&o->T, av.T);
        }
        return;
      }
    }
  }
  const uint32_t* idx = nullptr;
  const ge_cached* T = nullptr;
  uint32_t hot = 0xffffffffu;
  if (row >= hr.row0 && row < hr.row0 + 3 * hr.rows_per_vec) {
    const size_t m = (row - hr.row0) / hr.rows_per_vec;
    if (hr.hot[m] != 0xffffffffu) {
      hot = hr.hot[m];
      T = hr.T[m];
      idx = hr.idx[m] + (row - hr.row0 - m * hr.rows_per_vec) * ncols;
    }
  }
  // one LDS list, two ends: ordinary non-zero scalars from the front, hot entries from the back
  __shared__ uint16_t list[kSeg];
  __shared__ uint32_t n_front, n_back;
  // gridDim.y column chunks per row: a row of 16384 full-width scalars is ~11 ms of one workgroup, and a commitment (or a
  // rank's share of one) of a few waves of such workgroups ends in a long, mostly idle tail
  const size_t per = (ncols + gridDim.y - 1) / gridDim.y;
  const size_t c0 = (size_t)blockIdx.y * per, c1 = (c0 + per < ncols) ? c0 + per : ncols;
  for (size_t seg = c0; seg < c1; seg += kSeg) {
    const size_t seg_end = (seg + kSeg < c1) ? seg + kSeg : c1;
    if (threadIdx.x == 0) { n_front = 0; n_back = 0; }
    __syncthreads();
    for (size_t j = seg + threadIdx.x; j < seg_end; j += kMsmBlock) {
      if (fq_is_zero(fq_load(zr + j))) continue;
      if (idx && idx[j] == hot) list[kSeg - 1 - atomicAdd(&n_back, 1u)] = (uint16_t)(j - seg);
      else if (!hot_only) list[atomicAdd(&n_front, 1u)] = (uint16_t)(j - seg);
    }
    __syncthreads();
    const uint32_t nf = n_front, nb = n_back;
    for (uint32_t k = threadIdx.x; k < nf; k += kMsmBlock) {
      const size_t j = seg + list[k];
      acc.mul_acc(fq_from_mont(fq_load(zr + j)), tv, j);
    }
    for (uint32_t k = threadIdx.x; k < nb; k += kMsmBlock) {
      const ge_cached* t = T + seg + list[kSeg - 1 - k];
      ge_cached q;
      q.YpX = fp_load(&t->YpX); q.YmX = fp_load(&t->YmX); q.Z = fp_load(&t->Z); q.T2d = fp_load(&t->T2d);
      acc.add_cached(q);
    }
    __syncthreads();
  }
  __shared__ ge_ext sh[kMsmBlock];
  sh[threadIdx.x] = acc.ext();
  __syncthreads();
  ge_tree_quad(sh, kMsmBlock);
  if (threadIdx.x == 0) {
    const ge_ext av = sh[0];
    ge_ext* o = out + (size_t)blockIdx.x * gridDim.y + blockIdx.y;
    fp_store(&o->X, av.X); fp_store(&o->Y, av.Y); fp_store(&o->Z, av.Z); fp_store(&o->T, av.T);
  }
}


// ---- row per lane: the table walks of a LARGE commitment out of L2 instead of HBM -------------------------------------------
// Under the row kernels above the chip sits at its power limit: tools/ubench_msm_variants (profiles/r04_ubench_msm_variants.txt)
// measured the point addition alone at 31.3 G/s and 2.39 GHz, the same loop with the table walk's gathers at 22.6 G/s with
// the clock down to 1.79 GHz at 1390 W -- a 96-byte entry out of a 71 GB table costs about as much energy as a third of the
// addition it feeds.  Every row of a commitment walks the SAME generators, so the fix is to make thousands of lanes want the
// same 196 KB block (one window of one generator: 2048 multiples) at the same moment: here a LANE is a row, a workgroup is
// 256 rows, and all workgroups of a "strip" (a run of ncols / S generators) step through (generator, window) together, each
// lane adding the multiple its own digit selects.  A block is then read ~rows / 2048 times within microseconds, i.e. once
// from HBM and otherwise from L2 (workgroup b works on strip b mod S; workgroups are dealt round-robin to the 8 XCDs and S is a
// multiple of 8, so a strip stays on one XCD's L2).  Measured on the synthetic walk: 26.4 G additions/s at 2.06 GHz.
// Rows whose walks are irregular stay with msm_rows_hot_kernel (constant rows, rows with many zero scalars or hot entries:
// a lane that skips idles while its wave-mates walk); the hot entries of the rows taken here are added by that kernel too
// (hot_only), and strip_combine_kernel sums the S partial points of a row into it.

// flag[i] = 1: row i of the launch goes to msm_strip_kernel
__global__ __launch_bounds__(kMsmBlock) void strip_classify_kernel(const fq* __restrict__ Z, size_t stride, size_t ncols, HotRows hr,
                                                                    size_t row_base, size_t row_step, int z_compact,
                                                                    uint8_t* __restrict__ flag) {
  const size_t row = row_base + (size_t)blockIdx.x * row_step;
  const fq* zr = Z + (z_compact ? (size_t)blockIdx.x : row) * stride;
  const uint32_t* idx = nullptr;
  uint32_t hot = 0xffffffffu;
  if (hr.rows_per_vec && row >= hr.row0 && row < hr.row0 + 3 * hr.rows_per_vec) {
    const size_t m = (row - hr.row0) / hr.rows_per_vec;
    if (hr.hot[m] != 0xffffffffu) { hot = hr.hot[m]; idx = hr.idx[m] + (row - hr.row0 - m * hr.rows_per_vec) * ncols; }
  }
  const fq first = fq_load(zr);
  int same = 1;
  uint32_t skipped = 0;
  for (size_t j = threadIdx.x; j < ncols; j += kMsmBlock) {
    const fq v = fq_load(zr + j);
    same &= fq_same(first, v) ? 1 : 0;
    skipped += (fq_is_zero(v) || (idx && idx[j] == hot)) ? 1u : 0u;
  }
  __shared__ uint32_t sh[kMsmBlock];
  sh[threadIdx.x] = skipped;
  const int all_same = __syncthreads_and(same);
  for (int st = kMsmBlock / 2; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) flag[blockIdx.x] = (!all_same && sh[0] * 8 <= ncols) ? 1 : 0;
}

// list = the flagged rows in ascending order, *count = how many (one workgroup of 1024 threads; nrows <= 2^20)
__global__ __launch_bounds__(1024) void strip_list_kernel(const uint8_t* __restrict__ flag, size_t nrows, uint32_t* __restrict__ list,
                                                           uint32_t* __restrict__ count) {
  __shared__ uint32_t base[1024];
  const size_t per = (nrows + 1023) / 1024, r0 = (size_t)threadIdx.x * per, r1 = r0 + per < nrows ? r0 + per : nrows;
  uint32_t n = 0;
  for (size_t r = r0; r < r1; r++) n += flag[r] ? 1u : 0u;
  base[threadIdx.x] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int t = 0; t < 1024; t++) { const uint32_t v = base[t]; base[t] = run; run += v; }
    *count = run;
  }
  __syncthreads();
  uint32_t o = base[threadIdx.x];
  for (size_t r = r0; r < r1; r++)
    if (flag[r]) list[o++] = (uint32_t)r;
}

// the listed rows from n_use on go back to the row kernel: flags cleared, *count = n_use (launched after strip_list_kernel)
__global__ __launch_bounds__(256) void strip_trim_kernel(const uint32_t* __restrict__ list, uint32_t* __restrict__ count, uint32_t n_use,
                                                          uint32_t n_all, uint8_t* __restrict__ flag) {
  const uint32_t li = n_use + blockIdx.x * 256 + threadIdx.x;
  if (li < n_all) flag[list[li]] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *count = n_use;
}

// parts[li * S + s] = sum over strip s of the table walks of listed row li (ten-limb accumulator)
__global__ __launch_bounds__(kMsmBlock, 3) void msm_strip_kernel(const fq* __restrict__ Z, size_t stride, size_t ncols, TableView tv, HotRows hr,
                                                                  const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count,
                                                                  int S, size_t row_base, size_t row_step, int z_compact,
                                                                  ge_ext* __restrict__ parts, uint32_t* __restrict__ sync, int lag) {
  const uint32_t n = *list_count;
  const uint32_t s = blockIdx.x % (uint32_t)S, grp = blockIdx.x / (uint32_t)S;
  if (grp * kMsmBlock >= n) return;
  const uint32_t li = grp * kMsmBlock + threadIdx.x;
  const bool live = li < n;
  const uint32_t i = list[live ? li : n - 1];
  const size_t row = row_base + (size_t)i * row_step;
  const fq* zr = Z + (z_compact ? (size_t)i : row) * stride;
  const uint32_t* idx = nullptr;
  uint32_t hot = 0xffffffffu;
  if (hr.rows_per_vec && row >= hr.row0 && row < hr.row0 + 3 * hr.rows_per_vec) {
    const size_t m = (row - hr.row0) / hr.rows_per_vec;
    if (hr.hot[m] != 0xffffffffu) { hot = hr.hot[m]; idx = hr.idx[m] + (row - hr.row0 - m * hr.rows_per_vec) * ncols; }
  }
  const size_t per = (ncols + (size_t)S - 1) / (size_t)S, j0 = (size_t)s * per, j1 = j0 + per < ncols ? j0 + per : ncols;
  ge10 acc = ge10_identity();
  // Keeping a strip's workgroups in step (sync != nullptr): a workgroup counts every generator it has finished, and starts
  // generator j only when all `groups` workgroups of its strip have finished generator j - lag -- the L2 then holds the few
  // blocks between the slowest and the fastest workgroup instead of losing them.  All workgroups of the launch are resident
  // (2 or 3 per CU by construction); should they not be (another stream on the device), the bounded wait gives up and the
  // workgroup runs on unsynchronised -- the result does not depend on the order.
  const uint32_t groups = (n + kMsmBlock - 1) / kMsmBlock;
  uint32_t* my_sync = sync ? sync + (size_t)s * (per + 1) : nullptr;
  __shared__ int sh_nosync;
  if (threadIdx.x == 0) sh_nosync = 0;
  __syncthreads();
  for (size_t j = j0; j < j1; j++) {
    if (my_sync && j - j0 >= (size_t)lag) {
      if (threadIdx.x == 0 && !sh_nosync) {
        long spin = 0;
        while (__hip_atomic_load(my_sync + (j - j0 - (size_t)lag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < groups)
          if (++spin > (1L << 16)) { sh_nosync = 1; break; }  // ~0.1 s; a legitimate wait is one generator's walk (~150 us)
      }
      __syncthreads();
    }
    fq v = fq_load(zr + j);
    const bool act = live && !fq_is_zero(v) && !(idx && idx[j] == hot);
    if (__builtin_amdgcn_ballot_w64(act) != 0) {  // somebody in the wave walks this generator
      fq sc = fq_from_mont(v);
      if (!act) sc = fq_zero();  // all digits zero: the walk below adds nothing for this lane, in step with its wave-mates
      table_mul_acc10(acc, sc, tv, j);
    }
    if (my_sync) {
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(my_sync + (j - j0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (live) {
    const ge_ext av = ge10_to_ext(acc);
    ge_ext* o = parts + (size_t)li * (size_t)S + s;
    fp_store(&o->X, av.X); fp_store(&o->Y, av.Y); fp_store(&o->Z, av.Z); fp_store(&o->T, av.T);
  }
}

// out[list[li]] += sum_s parts[li * S + s]
__global__ __launch_bounds__(64) void strip_combine_kernel(const ge_ext* __restrict__ parts, const uint32_t* __restrict__ list,
                                                            const uint32_t* __restrict__ list_count, int S, ge_ext* __restrict__ out) {
  const uint32_t li = blockIdx.x * 64 + threadIdx.x;
  if (li >= *list_count) return;
  ge_ext* o = out + list[li];
  ge_ext a;
  a.X = fp_load(&o->X); a.Y = fp_load(&o->Y); a.Z = fp_load(&o->Z); a.T = fp_load(&o->T);
  for (int s = 0; s < S; s++) {
    const ge_ext* p = parts + (size_t)li * (size_t)S + s;
    ge_ext q;
    q.X = fp_load(&p->X); q.Y = fp_load(&p->Y); q.Z = fp_load(&p->Z); q.T = fp_load(&p->T);
    a = ge_add(a, q);
  }
  fp_store(&o->X, a.X); fp_store(&o->Y, a.Y); fp_store(&o->Z, a.Z); fp_store(&o->T, a.T);
}

// Profiling aid (vpin_prof_enable(ctx, 2)): the affine table additions msm_rows_kernel performs for the same
// arguments -- non-zero signed digits of every non-zero scalar, one scalar for a constant row -- summed into *count.
__device__ __forceinline__ uint32_t count_digits(fq s, const TableView& tv, size_t j) {
  (void)fq_fold_sign(s);
  const TableSeg sg = table_seg(tv, j);
  uint32_t carry = 0, n = 0;
  const uint32_t full = 1u << sg.c, half = 1u << (sg.c - 1);
  for (int w = 0; w < sg.W; w++) {
    uint32_t v = scalar_digit(s, w, sg.c) + carry;
    const bool neg = v > half;
    const uint32_t mag = neg ? full - v : v;
    carry = neg ? 1u : 0u;
    n += mag != 0;
  }
  return n;
}
__global__ __launch_bounds__(kMsmBlock) void msm_count_adds_kernel(const fq* __restrict__ Z, size_t stride, size_t ncols,
                                                                   const fq* __restrict__ extra, int n_extra, size_t extra_base0,
                                                                   TableView tv, unsigned long long* __restrict__ count,
                                                                   HotRows hr = HotRows{}, size_t row_base = 0, size_t row_step = 1,
                                                                   int z_compact = 0) {
  const size_t row = row_base + (size_t)blockIdx.x * row_step;
  const fq* zr = Z + (z_compact ? (size_t)blockIdx.x : row) * stride;
  // msm_rows_hot_kernel: the hot-column entries are not table additions
  const uint32_t* hidx = nullptr;
  uint32_t hot = 0xffffffffu;
  if (hr.rows_per_vec && row >= hr.row0 && row < hr.row0 + 3 * hr.rows_per_vec) {
    const size_t m = (row - hr.row0) / hr.rows_per_vec;
    if (hr.hot[m] != 0xffffffffu) { hot = hr.hot[m]; hidx = hr.idx[m] + (row - hr.row0 - m * hr.rows_per_vec) * ncols; }
  }
  uint32_t n = 0;
  bool constant = false;
  if (ncols >= 256 && (ncols & (ncols - 1)) == 0 && ((size_t)1 << (tv.nbt - tv.sum0 - 1)) >= ncols) {
    const fq first = fq_load(zr);
    int same = 1;
    for (size_t j = threadIdx.x; j < ncols; j += kMsmBlock) same &= fq_same(first, fq_load(zr + j)) ? 1 : 0;
    constant = __syncthreads_and(same) != 0;
    if (constant && threadIdx.x == 0 && !fq_is_zero(first))
      n += count_digits(fq_from_mont(first), tv, tv.sum0 + (size_t)(63 - __builtin_clzll((unsigned long long)ncols)));
  }
  if (!constant)
    for (size_t j = threadIdx.x; j < ncols; j += kMsmBlock) {
      const fq s = fq_load(zr + j);
      if (!fq_is_zero(s) && !(hidx && hidx[j] == hot)) n += count_digits(fq_from_mont(s), tv, j);
    }
  for (int e = threadIdx.x; e < n_extra; e += kMsmBlock) {
    const fq x = fq_load(extra + row * (size_t)n_extra + e);
    if (!fq_is_zero(x)) n += count_digits(fq_from_mont(x), tv, extra_base0 + e);
  }
  __shared__ uint32_t sh[kMsmBlock];
  sh[threadIdx.x] = n;
  __syncthreads();
  for (int st = kMsmBlock / 2; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0 && sh[0]) atomicAdd(count, (unsigned long long)sh[0]);
}

// Few-row MSMs (the evaluation proof's bullet rounds: 1-2 rows of R+2 scalars) are latency bound,
// so the 32 windows of a scalar are spread over 8 threads (4 table adds each) and a 256-thread
// workgroup covers 32 scalars; the per-workgroup partial points go back to the host, which adds
// the few dozen partials and compresses (microseconds on a CPU core, ~0.4 ms as a GPU tail).
constexpr int kWideScalars = 32;  // scalars per workgroup and pass
// Q passes per workgroup: 32 Q scalars each.  Long rows (>= 8192 scalars) take Q = 4: the 255 tree additions of a
// workgroup are then spread over four times as many table additions, and a row needs a quarter of the workgroups.
template <int Q>
__global__ __launch_bounds__(kMsmBlock) void msm_wide_kernel(const fq* __restrict__ S, size_t ncols, TableView tv,
                                                             fp* __restrict__ parts_xyzt) {
  const size_t row = blockIdx.y;
  const int grp = threadIdx.x & 7;
  ge_ext acc = ge_identity();
#pragma unroll 1
  for (int q = 0; q < Q; q++) {
    const size_t j = ((size_t)blockIdx.x * Q + q) * kWideScalars + (threadIdx.x >> 3);
    if (j < ncols) {
      fq s = fq_load(S + row * ncols + j);
      if (!fq_is_zero(s)) {
        const TableSeg sg = table_seg(tv, j);
        const int wpg = (sg.W + 7) / 8;  // this lane's windows [grp*wpg, (grp+1)*wpg)
        int w0 = grp * wpg, w1 = (w0 + wpg < sg.W) ? w0 + wpg : sg.W;
        if (w0 < w1) table_mul_acc_range(acc, fq_from_mont(s), sg, w0, w1);
      }
    }
  }
  __shared__ ge_ext sh[kMsmBlock];
  sh[threadIdx.x] = acc;
  __syncthreads();
  ge_tree_quad(sh, kMsmBlock);
  if (threadIdx.x == 0) {
    acc = sh[0];
    fp* o = parts_xyzt + 4 * (row * gridDim.x + blockIdx.x);
    fp_store(o, fp_freeze(acc.X)); fp_store(o + 1, fp_freeze(acc.Y)); fp_store(o + 2, fp_freeze(acc.Z)); fp_store(o + 3, fp_freeze(acc.T));
  }
}


// ---- one kernel per round of the bullet reduction (host side: bullet.hip) ------------------------------------------
// BulletReductionProof::prove (nizk/bullet.rs:32-132), round with half length n, everything the device does for it:
//   * fold the previous round's challenge into a, b and the generator coefficients s_j (bullet.rs:99-109);
//   * L = <a_L, G_R>, R = <a_R, G_L> over the stream generators as fixed-base MSMs with scalars a'[.] * s'_j
//     (bullet.hip: G is never folded) -- the last nmsm workgroups, 32 generators each, eight lanes per scalar;
//   * the cross inner products c_L = <a_L, b_R>, c_R = <a_R, b_L> and the folded a, b for the next round -- the first
//     workgroups of the grid, published through the pinned mailbox so the host forms c*Q + blind*H while the MSM runs.
// No workgroup reads what another one writes (a, b are double-buffered; s_j belongs to its eight lanes), so one launch
// replaces the fold, rows and MSM launches and both copies of a round.  finish = 1: the last fold only --
// g_hat = sum_j s_j g_j, x_hat, a_hat.
struct BulletStep {
  const fq* a_prev; const fq* b_prev;  // live length 4n when fold (2n otherwise)
  fq* a_next; fq* b_next;              // live length 2n after this launch
  fq* sj;                              // [R]
  size_t n, R;
  int fold, finish, nmsm;
  fq u, u_inv;                         // the previous round's challenge (fold / finish)
  fp* parts;                           // pinned: [2][nmsm] points X|Y|Z|T, row 0 = L, row 1 = R
  uint32_t* up;                        // pinned: two scalars per inner-product workgroup, 12 words each
  uint32_t seq;
};

__global__ __launch_bounds__(kMsmBlock) void bullet_step_kernel(BulletStep a, TableView tv) {
  const size_t n = a.n;
  // the inner-product workgroups come first in the grid: they are dispatched first, so the host has c_L, c_R (and forms
  // c*Q + blind*H) while the MSM workgroups, more than fit the device at once for the longest rows, are still running
  const int nip = (int)gridDim.x - a.nmsm;
  if ((int)blockIdx.x < nip) {
    // ---- inner products + the folded vectors ----
    const int ib = (int)blockIdx.x;
    const size_t i = (size_t)ib * kMsmBlock + threadIdx.x;
    fq pl = fq_zero(), pr = fq_zero();
    if (a.finish) {
      if (ib == 0 && threadIdx.x == 0) {
        pl = fq_add(fq_mul(fq_load(a.a_prev), a.u), fq_mul(a.u_inv, fq_load(a.a_prev + 1)));  // x_hat
        pr = fq_add(fq_mul(fq_load(a.b_prev), a.u_inv), fq_mul(a.u, fq_load(a.b_prev + 1)));  // a_hat
      }
    } else if (i < n) {
      fq a0, a1, b0, b1;
      if (a.fold) {
        a0 = fq_add(fq_mul(fq_load(a.a_prev + i), a.u), fq_mul(a.u_inv, fq_load(a.a_prev + 2 * n + i)));
        a1 = fq_add(fq_mul(fq_load(a.a_prev + n + i), a.u), fq_mul(a.u_inv, fq_load(a.a_prev + 3 * n + i)));
        b0 = fq_add(fq_mul(fq_load(a.b_prev + i), a.u_inv), fq_mul(a.u, fq_load(a.b_prev + 2 * n + i)));
        b1 = fq_add(fq_mul(fq_load(a.b_prev + n + i), a.u_inv), fq_mul(a.u, fq_load(a.b_prev + 3 * n + i)));
      } else {
        a0 = fq_load(a.a_prev + i); a1 = fq_load(a.a_prev + n + i);
        b0 = fq_load(a.b_prev + i); b1 = fq_load(a.b_prev + n + i);
      }
      fq_store(a.a_next + i, a0); fq_store(a.a_next + n + i, a1);
      fq_store(a.b_next + i, b0); fq_store(a.b_next + n + i, b1);
      pl = fq_mul(a0, b1);
      pr = fq_mul(a1, b0);
    }
    __shared__ fq shp[kMsmBlock / 64][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    pl = fq_wave_sum(pl);
    pr = fq_wave_sum(pr);
    if (lane == 0) { shp[wave][0] = pl; shp[wave][1] = pr; }
    __syncthreads();
    if (threadIdx.x < 2) {
      fq t = shp[0][threadIdx.x];
#pragma unroll
      for (int w = 1; w < kMsmBlock / 64; w++) t = fq_add(t, shp[w][threadIdx.x]);
      publish_scalar(a.up + (2 * (size_t)ib + threadIdx.x) * 12, t, a.seq);
    }
    return;
  }
  // ---- the round's MSM: 32 generators per workgroup, the W windows of a scalar over eight lanes ----
  const int g = threadIdx.x >> 3, grp = threadIdx.x & 7;
  const size_t blk = (size_t)blockIdx.x - nip;  // MSM workgroup index
  const size_t j = blk * kWideScalars + g;
  ge_ext acc = ge_identity();
  if (j < a.R) {
    fq sc = fq_load(a.sj + j);
    if (a.fold || a.finish) {
      const size_t hp = a.finish ? 1 : 2 * n;  // half length of the vectors the challenge folded
      sc = fq_mul(sc, ((j & (2 * hp - 1)) < hp) ? a.u_inv : a.u);
      if (grp == 0) fq_store(a.sj + j, sc);  // the eight lanes of j read s_j in the same instruction, before this store
    }
    if (!a.finish) {
      const size_t pos = j & (2 * n - 1);
      const size_t idx = pos >= n ? pos - n : n + pos;  // L: a_L[pos - n] on G_R; R: a_R[pos] on G_L
      fq av = a.fold ? fq_add(fq_mul(fq_load(a.a_prev + idx), a.u), fq_mul(a.u_inv, fq_load(a.a_prev + 2 * n + idx)))
                     : fq_load(a.a_prev + idx);
      sc = fq_mul(av, sc);
    }
    if (!fq_is_zero(sc)) {
      const TableSeg sg = table_seg(tv, j);
      const int wpg = (sg.W + 7) / 8;
      const int w0 = grp * wpg, w1 = (w0 + wpg < sg.W) ? w0 + wpg : sg.W;
      if (w0 < w1) table_mul_acc_range(acc, fq_from_mont(sc), sg, w0, w1);
    }
  }
  // entry grp*32 + g: the first three levels add up the eight lanes of a generator, the rest runs over the generators
  __shared__ ge_ext sh[kMsmBlock];
  sh[grp * kWideScalars + g] = acc;
  __syncthreads();
  // generators alternate between R (pos < n) and L (pos >= n) in runs of n: one side per workgroup while n >= 32
  const int split = (a.finish || n >= (size_t)kWideScalars) ? 0 : (int)n;
  ge_tree_quad(sh, kMsmBlock, split);
  if (threadIdx.x < 2) {
    // thread 0 stores sh[0]; with a split, thread 1 stores sh[split] (the L run)
    const bool first_is_L = a.finish || ((blk * kWideScalars) & (2 * n - 1)) >= n;
    if (threadIdx.x == 0 || split) {
      const ge_ext p = sh[threadIdx.x ? split : 0];
      const int row = threadIdx.x ? 0 : ((split || !first_is_L) ? 1 : 0);
      fp* o = a.parts + 4 * ((size_t)row * a.nmsm + blk);
      fp_store(o, fp_freeze(p.X)); fp_store(o + 1, fp_freeze(p.Y)); fp_store(o + 2, fp_freeze(p.Z)); fp_store(o + 3, fp_freeze(p.T));
    }
  }
}

// More than 128 MSM workgroups (R > 4096): their partial points stay on the device and this second launch sums them per side
// in groups of up to 128 (one tree each), so the host still adds at most 8 points per side.  in: [2][nmsm] as
// bullet_step_kernel left them (a workgroup's slot of the side it does not hold is stale), out: [rows][G] in pinned memory.
__global__ __launch_bounds__(kMsmBlock) void bullet_parts_reduce_kernel(const fp* __restrict__ in, int nmsm, size_t n, int finish,
                                                                        fp* __restrict__ out) {
  const int G = gridDim.x, g = blockIdx.x, row = blockIdx.y, per = nmsm / G;
  __shared__ ge_ext sh[kMsmBlock / 2];
  if ((int)threadIdx.x < kMsmBlock / 2) {
    ge_ext p = ge_identity();
    const int b = g * per + (int)threadIdx.x;
    if ((int)threadIdx.x < per) {
      const bool is_L = finish || (((size_t)b * kWideScalars) & (2 * n - 1)) >= n;
      if (finish || n < (size_t)kWideScalars || is_L == (row == 0)) {
        const fp* q = in + 4 * ((size_t)row * nmsm + b);
        p.X = fp_load(q); p.Y = fp_load(q + 1); p.Z = fp_load(q + 2); p.T = fp_load(q + 3);
      }
    }
    sh[threadIdx.x] = p;
  }
  __syncthreads();
  ge_tree_quad(sh, kMsmBlock / 2);
  if (threadIdx.x == 0) {
    const ge_ext p = sh[0];
    fp* o = out + 4 * ((size_t)row * G + g);
    fp_store(o, fp_freeze(p.X)); fp_store(o + 1, fp_freeze(p.Y)); fp_store(o + 2, fp_freeze(p.Z)); fp_store(o + 3, fp_freeze(p.T));
  }
}

// second stage for long rows: out[row][p] = sum of parts p, p+P, p+2P, ... of that row (frozen X|Y|Z|T)
__global__ __launch_bounds__(64) void parts_reduce_kernel(const fp* __restrict__ in, size_t rows, size_t nraw, int P,
                                                          fp* __restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * (size_t)P) return;
  const size_t row = t / P, p = t % P;
  ge_ext acc = ge_identity();
  for (size_t k = p; k < nraw; k += P) {
    const fp* q = in + 4 * (row * nraw + k);
    ge_ext e;
    e.X = fp_load(q); e.Y = fp_load(q + 1); e.Z = fp_load(q + 2); e.T = fp_load(q + 3);
    acc = ge_add(acc, e);
  }
  fp* o = out + 4 * t;
  fp_store(o, fp_freeze(acc.X)); fp_store(o + 1, fp_freeze(acc.Y)); fp_store(o + 2, fp_freeze(acc.Z)); fp_store(o + 3, fp_freeze(acc.T));
}

// out[i] = s_i * g[base] for n scalars (the blind terms of the row commitments): 8 threads per
// scalar, ceil(W/8) windows each, 3-level LDS tree -- one wave handles 8 scalars
__global__ __launch_bounds__(64) void single_base_mul_kernel(const fq* __restrict__ S, size_t n, TableView tv, size_t base,
                                                             ge_ext* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 8 + (threadIdx.x >> 3);
  const int grp = threadIdx.x & 7;
  ge_ext acc = ge_identity();
  if (i < n) {
    fq s = fq_load(S + i);
    if (!fq_is_zero(s)) {
      const TableSeg sg = table_seg(tv, base);
      const int wpg = (sg.W + 7) / 8;
      int w0 = grp * wpg, w1 = (w0 + wpg < sg.W) ? w0 + wpg : sg.W;
      if (w0 < w1) table_mul_acc_range(acc, fq_from_mont(s), sg, w0, w1);
    }
  }
  __shared__ ge_ext sh[64];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 4; st >= 1; st >>= 1) {
    if (grp < st) {
      acc = ge_add(acc, sh[threadIdx.x + st]);
      sh[threadIdx.x] = acc;
    }
    __syncthreads();
  }
  if (grp == 0 && i < n) {
    ge_ext* o = out + i;
    fp_store(&o->X, acc.X); fp_store(&o->Y, acc.Y); fp_store(&o->Z, acc.Z); fp_store(&o->T, acc.T);
  }
}

// out[row] = sum of the `chunks` partial points of that row
__global__ __launch_bounds__(64) void ge_sum_chunks_kernel(const ge_ext* __restrict__ parts, size_t rows, int chunks,
                                                           ge_ext* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  ge_ext acc;
  const ge_ext* p = parts + i * chunks;
  acc.X = fp_load(&p->X); acc.Y = fp_load(&p->Y); acc.Z = fp_load(&p->Z); acc.T = fp_load(&p->T);
  for (int k = 1; k < chunks; k++) {
    ge_ext q;
    q.X = fp_load(&p[k].X); q.Y = fp_load(&p[k].Y); q.Z = fp_load(&p[k].Z); q.T = fp_load(&p[k].T);
    acc = ge_add(acc, q);
  }
  fp_store(&out[i].X, acc.X); fp_store(&out[i].Y, acc.Y); fp_store(&out[i].Z, acc.Z); fp_store(&out[i].T, acc.T);
}

// out[i] = a[i] + b[i]  (row-wise commitment combination, proof_point_mult.rs:75-80)
__global__ __launch_bounds__(64) void ge_add_rows_kernel(const ge_ext* __restrict__ a, const ge_ext* __restrict__ b, size_t n,
                                                         ge_ext* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ge_ext p, q;
  p.X = fp_load(&a[i].X); p.Y = fp_load(&a[i].Y); p.Z = fp_load(&a[i].Z); p.T = fp_load(&a[i].T);
  q.X = fp_load(&b[i].X); q.Y = fp_load(&b[i].Y); q.Z = fp_load(&b[i].Z); q.T = fp_load(&b[i].T);
  ge_ext r = ge_add(p, q);
  fp_store(&out[i].X, r.X); fp_store(&out[i].Y, r.Y); fp_store(&out[i].Z, r.Z); fp_store(&out[i].T, r.T);
}

// out[j] = generator j (j < n), out[n] = generator blind_base (n_extra = 1): multiple 1 of window 0 in the base's table row
__global__ __launch_bounds__(256) void pip_gens_kernel(TableView tv, size_t n, size_t blind_base, int n_extra, ge_niels* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n + (size_t)n_extra) return;
  const TableSeg sg = table_seg(tv, i < n ? i : blind_base);
  const ge_niels e = niels_load(sg.entry(0, 0));
  fp_store(&out[i].ypx, e.ypx); fp_store(&out[i].ymx, e.ymx); fp_store(&out[i].xy2d, e.xy2d);
}

// RistrettoPoint::compress for n points, one thread each; also emits canonical X|Y|Z|T
__global__ __launch_bounds__(64) void ge_compress_kernel(const ge_ext* __restrict__ pts, size_t n, fp* __restrict__ out32,
                                                         fp* __restrict__ out_xyzt) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ge_ext p;
  p.X = fp_load(&pts[i].X); p.Y = fp_load(&pts[i].Y); p.Z = fp_load(&pts[i].Z); p.T = fp_load(&pts[i].T);
  if (out32) fp_store(out32 + i, ge_compress(p));
  if (out_xyzt) {
    fp_store(out_xyzt + 4 * i, fp_freeze(p.X)); fp_store(out_xyzt + 4 * i + 1, fp_freeze(p.Y));
    fp_store(out_xyzt + 4 * i + 2, fp_freeze(p.Z)); fp_store(out_xyzt + 4 * i + 3, fp_freeze(p.T));
  }
}

}  // namespace vpin

namespace vpin {

// ---- hash-to-group on the device: RistrettoPoint::from_uniform_bytes (RFC 9496 4.3.4 MAP, twice, added) ------------------
// MultiCommitGens::new (commitments.rs:20-38) maps 64 bytes of a SHAKE256 stream per generator; the host does the (sequential)
// stream and, for a 32 k-generator set, would spend 25-60 ms of eight cores on the two exponentiations per point.  One lane
// per generator here: the whole set in one pass of the chip.
__device__ __forceinline__ fp FP_ONE_MINUS_D_SQ() { return fp_const(0x945fc176u, 0xe27c09c1u, 0xcd5e350fu, 0x2c81a138u, 0xbe70dfe4u, 0x9994abddu, 0xb2b3e0d7u, 0x029072a8u); }
__device__ __forceinline__ fp FP_D_MINUS_ONE_SQ() { return fp_const(0x44ed4d20u, 0x31ad5aaau, 0xb01e1999u, 0xd29e4a2cu, 0x529b4eebu, 0x4cdcd32fu, 0xf66c2241u, 0x5968b37au); }
__device__ __forceinline__ fp FP_SQRT_AD_MINUS_ONE() { return fp_const(0x497b2e1bu, 0x7e97f6a0u, 0x1b7854bdu, 0xaf9d8e0cu, 0x31f5d1fdu, 0x0f3cfcc9u, 0x2b8348acu, 0x376931bfu); }
__device__ __forceinline__ fp FP_D_EDWARDS() { return fp_const(0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu); }

// RFC 9496 4.2 SQRT_RATIO_M1(u, v)
__device__ __forceinline__ fp fp_sqrt_ratio_m1(const fp& u, const fp& v, bool* was_square) {
  const fp v3 = fp_mul(fp_sqr(v), v), v7 = fp_mul(fp_sqr(v3), v);
  fp r = fp_mul(fp_mul(u, v3), fp_pow_p58(fp_mul(u, v7)));
  const fp check = fp_mul(v, fp_sqr(r));
  const fp nu = fp_neg(u), nu_i = fp_mul(nu, FP_SQRT_M1());
  const bool correct = fp_eq(check, u), flipped = fp_eq(check, nu), flipped_i = fp_eq(check, nu_i);
  r = fp_select(flipped || flipped_i, fp_mul(r, FP_SQRT_M1()), r);
  *was_square = correct || flipped;
  return fp_abs(r);
}

__device__ __noinline__ ge_ext ge_elligator(const fp& t) {
  const fp one = fp_one(), m1 = fp_neg(one), d = FP_D_EDWARDS();
  const fp r = fp_mul(FP_SQRT_M1(), fp_sqr(t));
  const fp u = fp_mul(fp_add(r, one), FP_ONE_MINUS_D_SQ());
  const fp v = fp_mul(fp_sub(m1, fp_mul(r, d)), fp_add(r, d));
  bool sq;
  fp s = fp_sqrt_ratio_m1(u, v, &sq);
  const fp sp = fp_neg(fp_abs(fp_mul(s, t)));
  s = fp_select(sq, s, sp);
  const fp c = fp_select(sq, m1, r);
  const fp N = fp_sub(fp_mul(fp_mul(c, fp_sub(r, one)), FP_D_MINUS_ONE_SQ()), v);
  const fp ss = fp_sqr(s);
  const fp w0 = fp_mul(fp_add(s, s), v), w1 = fp_mul(N, FP_SQRT_AD_MINUS_ONE()), w2 = fp_sub(one, ss), w3 = fp_add(one, ss);
  ge_ext p;
  p.X = fp_mul(w0, w3); p.Y = fp_mul(w2, w1); p.Z = fp_mul(w1, w3); p.T = fp_mul(w0, w2);
  return p;
}

// stream: 64 bytes per generator; out: X | Y | Z | T, canonical 32-byte integers (the form Point::to_xyzt has on the host)
__global__ __launch_bounds__(64) void gens_derive_kernel(const uint32_t* __restrict__ stream, size_t nb, fp* __restrict__ xyzt) {
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= nb) return;
  fp t0, t1;
#pragma unroll
  for (int k = 0; k < 8; k++) { t0.v[k] = stream[16 * i + k]; t1.v[k] = stream[16 * i + 8 + k]; }
  t0.v[7] &= 0x7fffffffu;  // FieldElement::from_bytes ignores bit 255
  t1.v[7] &= 0x7fffffffu;
  const ge_ext p = ge_add(ge_elligator(t0), ge_elligator(t1));
  fp_store(xyzt + 4 * i + 0, fp_freeze(p.X));
  fp_store(xyzt + 4 * i + 1, fp_freeze(p.Y));
  fp_store(xyzt + 4 * i + 2, fp_freeze(p.Z));
  fp_store(xyzt + 4 * i + 3, fp_freeze(p.T));
}

}  // namespace vpin

using namespace vpin;

extern "C" {

static int gens_build_once(vpin_ctx* c, const uint8_t* gens_xyzt, size_t nb, size_t budget, int cmax, vpin_gens** out);

// The table budget is a speed knob (every window bit saves ~8 % of a commitment), so it yields to what the device really has
// at this moment: at most a third of the FREE memory (a co-tenant, another process's tables, a smaller part), and a failed
// allocation halves it instead of failing the proof.  The instance, its decommitment and the proof temporaries need the rest.
static int gens_build(vpin_ctx* c, const uint8_t* gens_xyzt, size_t nb, size_t budget, vpin_gens** out) {
  (void)hipSetDevice(c->device);
  size_t free_b = 0, total_b = 0;
  // (not asked in a one-shot process: its tables are sized by use, and the first hipMemGetInfo of a process costs 35-50 ms)
  const hipError_t e_info = c->expected_proofs > 0 ? hipErrorUnknown : hipMemGetInfo(&free_b, &total_b);
  if (e_info == hipSuccess && free_b) {
    static const double frac = [] { const char* e = getenv("VPIN_GENS_FREE_FRACTION"); double v = e ? atof(e) : 0.0; return v > 0.0 && v < 1.0 ? v : 1.0 / 3.0; }();
    const size_t cap = (size_t)((double)free_b * frac);
    if (budget > cap) budget = cap;
  }
  const size_t floor_b = (size_t)1 << 28;
  if (budget < floor_b) budget = floor_b;
  // A table that serves a known, small number of proofs (vpin_ctx_set_expected_proofs: a CLI process) is not worth its widest
  // windows: an entry costs ~4x a table addition to construct (projective chain, batched inversion, 96-byte scattered
  // store), so the width minimises entries x 4 + scalars x windows (measured: 142 ms to build the 12-bit table of the 2^25
  // instance, whose one proof then saves 48 ms of additions against the 10-bit one, built in 36 ms).
  // VPIN_GENS_CMAX (experiments): 13-bit windows are 20 instead of 22 additions per scalar for twice the table (129 GB for the
  // first 16386 generators): measured in round 5, see DESIGN.md section 4
  static const int env_cmax = [] { const char* e = getenv("VPIN_GENS_CMAX"); int v = e ? atoi(e) : 12; return v < 8 ? 8 : v > 14 ? 14 : v; }();
  int cmax = env_cmax;
  if (c->expected_proofs > 0 && c->gens_scalars_per_proof > 0.0) {
    const double S = c->gens_scalars_per_proof * (double)c->expected_proofs, n_lo = (double)std::min<size_t>(nb, 16386);
    double best = 0.0;
    for (int cc = 8; cc <= 12; cc++) {  // 8 bits = the narrowest windows the second table segment already uses
      const double W = (double)((254 + cc - 1) / cc), cost = n_lo * W * (double)(1u << (cc - 1)) * 4.0 + S * W;
      if (cc == 8 || cost < best) { best = cost; cmax = cc; }
    }
  }
  for (;;) {
    int rc = gens_build_once(c, gens_xyzt, nb, budget, cmax, out);
    if (rc != VPIN_ENOMEM || budget <= floor_b) return rc;
    (void)hipGetLastError();
    budget /= 2;
    if (budget < floor_b) budget = floor_b;
  }
}

// bytes of a table slot for the NEXT table built: VPIN_TABLE_SLOT = 96 | 128 (read per build), default VPIN_NIELS_SLOT.  128 puts every
// entry on a 128-byte line of its own: -2.2 % on the four-lane LeNet step for a third more table memory at the same window width
// (profiles/r06_ab_slot128.txt) -- bench.py asks for it on a single-GPU run, the library's default stays 96 (headroom of split proofs)
static size_t table_slot_bytes() {
  const char* e = getenv("VPIN_TABLE_SLOT");
  const int v = e ? atoi(e) : VPIN_NIELS_SLOT;
  return v == 128 ? 128 : v == 96 ? 96 : (size_t)VPIN_NIELS_SLOT;
}

static int gens_build_once(vpin_ctx* c, const uint8_t* gens_xyzt, size_t nb, size_t budget, int cmax, vpin_gens** out) {
  if (!c || !gens_xyzt || !out || nb == 0) return VPIN_EINVAL;
  const size_t slot = table_slot_bytes();
  (void)hipSetDevice(c->device);
  vpin_gens* g = new (std::nothrow) vpin_gens();
  if (!g) return VPIN_ENOMEM;
  g->nb = nb;
  g->s32 = (int)(slot / 32);
  int nsum = 1;
  while (((size_t)2 << (nsum - 1)) <= nb) nsum++;  // S_0 .. S_floor(log2 nb)
  const size_t nbt = nb + (size_t)nsum;
  g->nbt = nbt;
  // Window widths by table budget.  Up to kSplitBases generators: one segment, the widest windows that fit (12 bits
  // up to ~5k generators under 24 GB, ...).  Longer streams: the first kSplitBases generators -- all that the
  // per-proof commitments of the largest instance walk (R = 2^14 columns + the two blind bases) -- get 12-bit
  // windows when the budget allows, the rest 8-bit ones: 71 + 6.4 GB for 32 786 bases, the 77 GB a uniform 11-bit
  // table took, and 22 instead of 24 table adds per scalar where the time goes.
  constexpr size_t kSplitBases = 16386;
  // the widest windows (<= cmax bits) whose table fits the budget; 6 bits (43 windows of 32 entries: 2.2 GB for 16386
  // generators) at the least, whatever the budget -- that allocation then succeeds or the build fails with VPIN_ENOMEM.
  // W and E always belong to the width returned (round 5: a budget below the 7-bit table used to leave 7-bit W and E beside
  // c = 6 -- a table too short for its scalars and WRONG sums; found by tools/ubench_pippenger.py's 1 GB walk, see
  // test_narrowest_windows_under_a_tiny_budget)
  auto fit = [&](size_t n, size_t bud, int cmax, int* c_, int* W_, int* E_) {
    int cc = cmax;
    for (; cc > 6; cc--)
      if (n * (size_t)((254 + cc - 1) / cc) * ((size_t)1 << (cc - 1)) * slot <= bud) break;
    *c_ = cc;
    *W_ = (254 + cc - 1) / cc;
    *E_ = 1 << (cc - 1);
  };
  g->split = nbt;
  fit(nb, budget, cmax, &g->c, &g->W, &g->E);
  if (nb > kSplitBases && getenv("VPIN_GENS_UNIFORM") == nullptr) {
    int c2, W2, E2;
    fit(kSplitBases, budget, cmax, &c2, &W2, &E2);
    const size_t lo_bytes = kSplitBases * (size_t)W2 * E2 * slot;
    if (c2 > g->c && lo_bytes < budget) {
      g->split = kSplitBases;
      g->c = c2; g->W = W2; g->E = E2;
      fit(nbt - kSplitBases, budget - lo_bytes, g->c, &g->c_hi, &g->W_hi, &g->E_hi);
    }
  }
  const size_t n_hi = nbt - g->split;
  const size_t entries = g->split * (size_t)g->W * g->E, entries_hi = n_hi * (size_t)g->W_hi * g->E_hi;
  const size_t max_entries = entries > entries_hi ? entries : entries_hi;
  const size_t max_shifts = std::max(g->split * (size_t)g->W, n_hi * (size_t)g->W_hi);
  TraceLap lap(c, "gens_build");
  // temporaries come from the context pool, so the blocks go on to serve proof temporaries instead of being
  // returned to the driver (freed VRAM is wiped before it can be handed out again: tools/ubench_malloc2.hip)
  DevBuf raw(c), b_shifts(c), b_prefix(c);
  if (raw.alloc(nbt * 128) != VPIN_OK || b_shifts.alloc(max_shifts * sizeof(ge_ext)) != VPIN_OK ||
      b_prefix.alloc(max_entries * sizeof(fp)) != VPIN_OK || driver_malloc((void**)&g->table, entries * slot) != hipSuccess ||
      (n_hi && driver_malloc((void**)&g->table_hi, entries_hi * slot) != hipSuccess)) {
    if (g->table) (void)hipFree(g->table);
    if (g->table_hi) (void)hipFree(g->table_hi);
    delete g;
    return VPIN_ENOMEM;
  }
  ge_ext* shifts = (ge_ext*)b_shifts.p;
  fp* prefix = (fp*)b_prefix.p;
  lap("hipMalloc");
  hipError_t e = hipMemcpyAsync(raw.p, gens_xyzt, nb * 128, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(gens_sum_kernel, dim3((unsigned)nsum), dim3(256), 0, c->stream, (fp*)raw.p, nb);
    hipLaunchKernelGGL(gens_sum_scan_kernel, dim3(1), dim3(64), 0, c->stream, (fp*)raw.p, nb, nsum);
    hipLaunchKernelGGL(gens_shift_kernel, dim3((unsigned)((g->split + 63) / 64)), dim3(64), 0, c->stream, (const fp*)raw.p, g->split,
                       g->W, g->c, shifts);
    hipLaunchKernelGGL(gens_table_kernel, dim3((unsigned)((g->split * g->W + 63) / 64)), dim3(64), 0, c->stream, shifts, g->split, g->W,
                       g->E, g->table, g->s32, prefix);
    if (n_hi) {  // same temporaries, stream-ordered after the first segment
      hipLaunchKernelGGL(gens_shift_kernel, dim3((unsigned)((n_hi + 63) / 64)), dim3(64), 0, c->stream,
                         (const fp*)raw.p + 4 * g->split, n_hi, g->W_hi, g->c_hi, shifts);
      hipLaunchKernelGGL(gens_table_kernel, dim3((unsigned)((n_hi * g->W_hi + 63) / 64)), dim3(64), 0, c->stream, shifts, n_hi,
                         g->W_hi, g->E_hi, g->table_hi, g->s32, prefix);
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  lap("table kernels");
  // A service's table (no expected proof count): the largest build temporary -- a third of the table's size -- goes back to the
  // driver instead of staying in the context's pool, where nothing of its size comes to reuse it: 227.6 -> 219.5 GiB in use after
  // the LeNet step, the step unchanged, 0.8 s more set-up (the freed memory is wiped when it is next handed out).  Not in a one-shot
  // process, whose table build is inside the timed span.  VPIN_GENS_TMP_KEEP restores the old behaviour.
  // Only THIS block leaves the pool (the stream has been synchronised above, successfully): the context's other cached blocks
  // stay where the next proofs find them, and no other lane's work is synchronised with.
  if (e == hipSuccess && c->expected_proofs == 0 && !getenv("VPIN_GENS_TMP_KEEP") && b_prefix.p) {
    void* pp = b_prefix.p;
    b_prefix.p = nullptr;
    dev_release_block(c, pp);
  }
  if (e != hipSuccess) {
    set_last_error("vpin_gens_create", e);
    (void)hipFree(g->table);
    if (g->table_hi) (void)hipFree(g->table_hi);
    delete g;
    return VPIN_EHIP;
  }
  *out = g;
  return VPIN_OK;
}

static size_t default_budget() {
  static const size_t budget = [] {
    const char* e = getenv("VPIN_GENS_BUDGET_GB");
    size_t gb = e ? (size_t)atoi(e) : 24;
    return (gb ? gb : 24) << 30;
  }();
  return budget;
}

int vpin_gens_create(vpin_ctx* c, const uint8_t* gens_xyzt, size_t nb, vpin_gens** out) {
  return gens_build(c, gens_xyzt, nb, default_budget(), out);
}

// The first nb generators of a label's stream from 64 x nb bytes of SHAKE256 output (the caller's: host/transcript.h), mapped
// to the group on the device; out_xyzt: nb x 128 bytes on the host, the layout of vpin_host_gens_derive
int vpin_gens_map_stream(vpin_ctx* c, const uint8_t* stream64, size_t nb, uint8_t* out_xyzt) {
  if (!c || !stream64 || !out_xyzt || nb == 0) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  DevBuf bs(c), bo(c);
  if (bs.alloc(nb * 64) || bo.alloc(nb * 128)) return VPIN_ENOMEM;
  VPIN_HIP_TRY(hipMemcpyAsync(bs.p, stream64, nb * 64, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(gens_derive_kernel, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, c->stream, (const uint32_t*)bs.p, nb, (fp*)bo.p);
  VPIN_HIP_TRY(hipGetLastError());
  VPIN_HIP_TRY(hipMemcpyAsync(out_xyzt, bo.p, nb * 128, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

// Process-wide registry of window tables by (device, label): every MultiCommitGens::new(n, label) is a
// prefix of one SHAKE stream (commitments.rs:20-38), so one table of the longest prefix requested so
// far serves every context, stream and polynomial size of that label.  Tables are immutable once
// built; superseded (shorter) ones stay alive because other contexts may still hold them.
namespace {
struct RegEntry { int device; std::string label; vpin_gens* g; };
struct Building { int device; std::string label; size_t nb; };
std::mutex g_reg_mu;
std::condition_variable g_reg_cv;
std::vector<RegEntry> g_reg;
std::vector<Building> g_building;  // tables under construction (the registry is unlocked meanwhile)
}  // namespace

int vpin_gens_shared(vpin_ctx* c, const char* label, const uint8_t* gens_xyzt, size_t nb, size_t budget_gb,
                     const vpin_gens** out) {
  if (!c || !label || !out || nb == 0) return VPIN_EINVAL;
  auto lookup = [&]() -> const vpin_gens* {
    const vpin_gens* best = nullptr;
    for (auto& e : g_reg)
      if (e.device == c->device && e.label == label && e.g->nb >= nb && (!best || e.g->nb < best->nb)) best = e.g;
    return best;
  };
  {
    std::unique_lock<std::mutex> lock(g_reg_mu);
    for (;;) {
      if (const vpin_gens* best = lookup()) { *out = best; return VPIN_OK; }
      // another thread is building a table that will cover this request: wait for it instead of building a second one
      // (tens of GB each); a shorter or different one under way does not help
      bool covered = false;
      for (auto& b : g_building) covered = covered || (b.device == c->device && b.label == label && b.nb >= nb);
      if (!covered) break;
      g_reg_cv.wait(lock);
    }
    if (!gens_xyzt) return VPIN_EINVAL;
    // A longer prefix of a stream that already has a table: the shorter table stays (views of other contexts point into it)
    // and the new one is built beside it.  Said once: a service should ask for its largest shape first (vpin_spark_prepare).
    static std::atomic<bool> said{false};
    for (auto& e : g_reg)
      if (e.device == c->device && e.label == label && !said.exchange(true))
        fprintf(stderr, "vpin_gens_shared: a table of \"%s\" for %zu generators is built beside the one for %zu (both stay allocated): "
                        "call vpin_spark_prepare / vpin_sat_prepare with the largest shape first\n", label, nb, e.g->nb);
    g_building.push_back(Building{c->device, label, nb});
  }
  // Built with the registry unlocked: other contexts of the process keep finding their tables, and building tables of other
  // labels, meanwhile.
  vpin_gens* g = nullptr;
  const int rc = gens_build(c, gens_xyzt, nb, budget_gb ? (budget_gb << 30) : default_budget(), &g);
  std::lock_guard<std::mutex> lock(g_reg_mu);
  for (size_t i = 0; i < g_building.size(); i++)
    if (g_building[i].device == c->device && g_building[i].label == label && g_building[i].nb == nb) { g_building.erase(g_building.begin() + (long)i); break; }
  if (!rc) {
    g_reg.push_back(RegEntry{c->device, label, g});
    *out = g;
  }
  g_reg_cv.notify_all();  // also on failure: the waiters then build for themselves
  return rc;
}

size_t vpin_gens_shared_bytes(int device) {
  std::lock_guard<std::mutex> lock(g_reg_mu);
  size_t total = 0;
  for (auto& e : g_reg)
    if (e.device == device)
      total += (size_t)32 * (size_t)e.g->s32 * ((size_t)e.g->W * e.g->split * (size_t)e.g->E +
                                     (e.g->split < e.g->nbt ? (size_t)e.g->W_hi * (e.g->nbt - e.g->split) * (size_t)e.g->E_hi : 0));
  return total;
}

void vpin_gens_shared_clear(void) {
  std::lock_guard<std::mutex> lock(g_reg_mu);
  for (auto& e : g_reg) {
    (void)hipSetDevice(e.device);
    (void)hipDeviceSynchronize();
    if (e.g->table) (void)hipFree(e.g->table);
    if (e.g->table_hi) (void)hipFree(e.g->table_hi);
    delete e.g;
  }
  g_reg.clear();
}

void vpin_gens_free(vpin_ctx* c, vpin_gens* g) {
  if (!g) return;
  if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
  if (g->table) (void)hipFree(g->table);
  if (g->table_hi) (void)hipFree(g->table_hi);
  delete g;
}

size_t vpin_gens_count(const vpin_gens* g) { return g ? g->nb : 0; }
// window layout chosen by the table budget: out = {c, W, split, c_hi, W_hi, nb + prefix-sum bases}
int vpin_gens_layout(const vpin_gens* g, size_t out[6]) {
  if (!g || !out) return VPIN_EINVAL;
  out[0] = (size_t)g->c; out[1] = (size_t)g->W; out[2] = g->split;
  out[3] = g->split < g->nbt ? (size_t)g->c_hi : 0; out[4] = g->split < g->nbt ? (size_t)g->W_hi : 0; out[5] = g->nbt;
  return VPIN_OK;
}
size_t vpin_gens_entry_bytes(void) { return table_slot_bytes(); }  // of the next table built (VPIN_TABLE_SLOT)

// shared implementation: rows of scalars -> points (kept on device), then optional outputs
// the row kernels' field arithmetic: ten 26/25-bit limbs (fp10_dev.h) unless VPIN_MSM_FP8 asks for the eight-limb form
static inline bool msm_ten_limbs() {
  static const bool ten = getenv("VPIN_MSM_FP8") == nullptr;
  return ten;
}

static inline TableView view(const vpin_gens* g) {
  return TableView{g->table, g->table_hi, g->split, g->nbt, g->c, g->W, g->E, g->c_hi, g->W_hi, g->E_hi, g->nb, g->s32};
}

// VPIN_MSM_PIPPENGER = c (9..12, or 1 = the default width): every row commitment of the provers by the bucket method of
// msm_pip.hip instead of the table walk (few-row MSMs, the bullet reduction and single-base terms stay on the table).  For
// parity runs of whole proofs (tests/test_gpu_msm_pippenger.py) and A/B timing; read per call.
static int pip_mode() {
  const char* e = getenv("VPIN_MSM_PIPPENGER");
  const int v = e ? atoi(e) : 0;
  return v <= 0 ? 0 : (v >= 9 && v <= 12 ? v : -1);  // -1: the default width
}

static int msm_rows_pip(vpin_ctx* c, const vpin_gens* g, const fq* dZ, size_t rows, size_t stride, size_t ncols, const fq* d_extra,
                        int n_extra, size_t extra_base0, int cbits, ge_ext* d_points) {
  DevBuf dgn(c);
  if (dgn.alloc((ncols + 1) * sizeof(ge_niels))) return VPIN_ENOMEM;
  hipLaunchKernelGGL(pip_gens_kernel, dim3((unsigned)((ncols + 1 + 255) / 256)), dim3(256), 0, c->stream, view(g), ncols, extra_base0,
                     n_extra, (ge_niels*)dgn.p);
  return pip_rows(c, (const ge_niels*)dgn.p, dZ, rows, stride, ncols, d_extra, n_extra, cbits, d_points,
                  c->prof_count_adds ? c->d_add_count : nullptr);
}

static int msm_rows(vpin_ctx* c, const vpin_gens* g, const fq* dZ, size_t rows, size_t stride, size_t ncols,
                    const fq* d_extra, int n_extra, size_t extra_base0, ge_ext* d_points) {
  if (const int pm = pip_mode())
    if (n_extra <= 1 && ncols + (size_t)n_extra <= 32768)
      return msm_rows_pip(c, g, dZ, rows, stride, ncols, d_extra, n_extra, extra_base0, pm < 0 ? 0 : pm, d_points);
  double nz_est = (double)rows * ((double)ncols + n_extra);
  size_t total = ncols + (size_t)n_extra;
  int chunks = 1;
  if (rows < 128) {  // spread a few rows over the chip: ~one scalar per thread, <= 64 chunks per row
    size_t want = (total + kMsmBlock - 1) / kMsmBlock * 4;
    chunks = (int)(want < 1 ? 1 : want > 64 ? 64 : want);
  } else if (rows >= 768 && total > (size_t)kSeg) {
    // long rows (more than one compaction segment): one workgroup per segment, so that a launch of a few waves of
    // workgroups (a rank's share of a split commitment) does not end in a long, mostly idle tail
    static const int env_chunks = [] { const char* e = getenv("VPIN_MSM_ROW_CHUNKS"); return e ? atoi(e) : 0; }();
    chunks = env_chunks > 0 ? env_chunks : (int)std::min<size_t>(8, (total + kSeg - 1) / kSeg);
  } else if (rows < 768 && total >= 2 * (size_t)kMsmBlock) {
    // Fewer than three workgroups per CU: a wave's 22-addition chains run at ~13 us per addition instead of ~9.5 us
    // with the SIMDs full, and every thread walks several of them.  Split the rows into column chunks until the chip is
    // full or a thread is down to one scalar (the 2^15..2^17-constraint instances' derefs commitment: 573 -> ~300 us).
    size_t want = (768 + rows - 1) / rows, most = total / kMsmBlock;
    chunks = (int)(want < most ? want : most);
    if (chunks < 1) chunks = 1;
  }
  DevBuf parts(c);
  ge_ext* dst = d_points;
  if (chunks > 1) {
    if (parts.alloc(rows * (size_t)chunks * sizeof(ge_ext))) return VPIN_ENOMEM;
    dst = (ge_ext*)parts.p;
  }
  if (c->prof_count_adds && c->d_add_count && rows >= 128)
    hipLaunchKernelGGL(msm_count_adds_kernel, dim3((unsigned)rows), dim3(kMsmBlock), 0, c->stream, dZ, stride, ncols, d_extra,
                       n_extra, extra_base0, view(g), c->d_add_count);
  {
    ProfScope ps(c, VPIN_K_MSM, 32.0 * nz_est, rows >= 128 ? VPIN_K_MSM_ROWS : -1);
    // unused dynamic LDS lowers the workgroups per CU from 3 to 2 on a shared device (vpin_ctx_set_shared_device);
    // VPIN_MSM_LDS_PAD overrides (experiments)
    static const int env_pad = [] { const char* e = getenv("VPIN_MSM_LDS_PAD"); return e ? atoi(e) : -1; }();
    const unsigned pad = env_pad >= 0 ? (unsigned)env_pad : (c->shared_device ? kSharedPad : 0u);
    if (msm_ten_limbs())
      hipLaunchKernelGGL(msm_rows_kernel<true>, dim3((unsigned)rows, (unsigned)chunks), dim3(kMsmBlock), pad, c->stream, dZ, stride,
                         ncols, d_extra, n_extra, extra_base0, view(g), dst);
    else
      hipLaunchKernelGGL(msm_rows_kernel<false>, dim3((unsigned)rows, (unsigned)chunks), dim3(kMsmBlock), pad, c->stream, dZ, stride,
                         ncols, d_extra, n_extra, extra_base0, view(g), dst);
  }
  if (chunks > 1)
    hipLaunchKernelGGL(ge_sum_chunks_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, c->stream, (const ge_ext*)dst,
                       rows, chunks, d_points);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}

int vpin_hyrax_commit(vpin_ctx* c, const vpin_gens* g, const vpin_table* Z, const uint8_t* blinds, size_t L,
                      size_t blind_base, uint8_t* out_compressed) {
  if (!c || !g || !Z || !Z->d || !blinds || !out_compressed || L == 0) return VPIN_EINVAL;
  if (Z->len % L != 0) return VPIN_ESHAPE;  // assert_eq!(L_size * R_size, self.Z.len())
  size_t R = Z->len / L;
  if (R > g->nb || blind_base >= g->nb) return VPIN_ESHAPE;  // assert_eq!(gens_n.n, self.len())
  (void)hipSetDevice(c->device);
  DevBuf dbl(c), dpts(c), dout(c);
  if (dbl.alloc(L * 32) || dpts.alloc(L * sizeof(ge_ext)) || dout.alloc(L * 32)) return VPIN_ENOMEM;
  VPIN_HIP_TRY(hipMemcpyAsync(dbl.p, blinds, L * 32, hipMemcpyHostToDevice, c->stream));
  int rc = msm_rows(c, g, Z->d, L, R, R, (const fq*)dbl.p, 1, blind_base, (ge_ext*)dpts.p);
  if (rc) return rc;
  hipLaunchKernelGGL(ge_compress_kernel, dim3((unsigned)((L + 63) / 64)), dim3(64), 0, c->stream, (const ge_ext*)dpts.p, L,
                     (fp*)dout.p, (fp*)nullptr);
  VPIN_HIP_TRY(hipGetLastError());
  VPIN_HIP_TRY(hipMemcpyAsync(out_compressed, dout.p, L * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}


// the same commitment by Pippenger's bucket method from the generators alone (msm_pip.hip): same bytes, no table walked
int vpin_hyrax_commit_pippenger(vpin_ctx* c, const vpin_gens* g, const vpin_table* Z, const uint8_t* blinds, size_t L,
                                size_t blind_base, int c_bits, uint8_t* out_compressed) {
  if (!c || !g || !Z || !Z->d || !out_compressed || L == 0) return VPIN_EINVAL;
  if (Z->len % L != 0) return VPIN_ESHAPE;
  const size_t R = Z->len / L;
  if (R > g->nb || (blinds && blind_base >= g->nb)) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  const int n_extra = blinds ? 1 : 0;
  DevBuf dbl(c), dpts(c), dout(c);
  if (dbl.alloc(L * 32) || dpts.alloc(L * sizeof(ge_ext)) || dout.alloc(L * 32)) return VPIN_ENOMEM;
  if (blinds) VPIN_HIP_TRY(hipMemcpyAsync(dbl.p, blinds, L * 32, hipMemcpyHostToDevice, c->stream));
  int rc = msm_rows_pip(c, g, Z->d, L, R, R, (const fq*)dbl.p, n_extra, blind_base, c_bits, (ge_ext*)dpts.p);
  if (rc) return rc;
  hipLaunchKernelGGL(ge_compress_kernel, dim3((unsigned)((L + 63) / 64)), dim3(64), 0, c->stream, (const ge_ext*)dpts.p, L,
                     (fp*)dout.p, (fp*)nullptr);
  VPIN_HIP_TRY(hipGetLastError());
  VPIN_HIP_TRY(hipMemcpyAsync(out_compressed, dout.p, L * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

}  // extern "C"

namespace vpin {

// DensePolynomial::commit(gens, None) for the derefs polynomial (8N scalars as L rows of R; sparse_mlpoly.rs:525-531) with the
// hot column of each matrix's col-derefs vector taken out (msm_rows_hot_kernel).  col_idx[m]: the N column indices of matrix
// m on the device, hot[m]: its hot column or 0xffffffff, e_ry: the table the col-derefs were gathered from.
int hyrax_commit_derefs_hot(vpin_ctx* c, const vpin_gens* g, const vpin_table* Z, size_t L, size_t N, const uint32_t* const col_idx[3],
                            const uint32_t hot[3], const fq* e_ry, uint8_t* out_compressed, size_t row0, size_t nrows, size_t row_step,
                            const fq* z_rows) {
  // z_rows != nullptr: the nrows rows of this call stored densely (Z may then be a shape-only handle: d unused)
  if (!c || !g || !Z || (!Z->d && !z_rows) || !col_idx || !hot || !e_ry || !out_compressed || L == 0 || Z->len % L) return VPIN_EINVAL;
  const size_t R = Z->len / L;
  if (R > g->nb || N % R || Z->len < 6 * N) return VPIN_ESHAPE;
  const fq* zbase = z_rows ? z_rows : (const fq*)Z->d;
  const int z_compact = z_rows ? 1 : 0;
  if (nrows == (size_t)-1) { row0 = 0; nrows = L; row_step = 1; }  // all rows
  if (row_step == 0 || (nrows && row0 + (nrows - 1) * row_step >= L)) return VPIN_ESHAPE;
  if (nrows == 0) return VPIN_OK;
  if (pip_mode()) return hyrax_commit_rows_strided(c, g, Z, L, row0, nrows, row_step, out_compressed, z_rows);  // plain rows, buckets
  (void)hipSetDevice(c->device);
  DevBuf dpts(c), dout(c), dT(c);
  if (dpts.alloc(nrows * sizeof(ge_ext)) || dout.alloc(nrows * 32)) return VPIN_ENOMEM;
  HotRows hr{};
  hr.row0 = 3 * (N / R);
  hr.rows_per_vec = N / R;
  // one T table per distinct hot column
  uint32_t distinct[3];
  int nd = 0;
  for (int m = 0; m < 3; m++) {
    hr.idx[m] = col_idx[m];
    hr.hot[m] = hot[m];
    if (hot[m] == 0xffffffffu) continue;
    bool seen = false;
    for (int k = 0; k < nd; k++) seen = seen || distinct[k] == hot[m];
    if (!seen) distinct[nd++] = hot[m];
  }
  if (nd && dT.alloc((size_t)nd * R * sizeof(ge_cached))) return VPIN_ENOMEM;
  for (int k = 0; k < nd; k++) {
    ge_cached* Tk = (ge_cached*)dT.p + (size_t)k * R;
    hipLaunchKernelGGL(scalar_times_bases_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, c->stream, e_ry + distinct[k], R, view(g), Tk);
    for (int m = 0; m < 3; m++)
      if (hot[m] == distinct[k]) hr.T[m] = Tk;
  }
  if (c->prof_count_adds && c->d_add_count)
    hipLaunchKernelGGL(msm_count_adds_kernel, dim3((unsigned)nrows), dim3(kMsmBlock), 0, c->stream, zbase, R, R, (const fq*)nullptr, 0,
                       (size_t)0, view(g), c->d_add_count, hr, row0, row_step, z_compact);
  static const int env_chunks = [] { const char* e = getenv("VPIN_MSM_HOT_CHUNKS"); return e ? atoi(e) : 0; }();
  const int chunks = env_chunks > 0 ? env_chunks : (int)std::max<size_t>(1, std::min<size_t>(8, R / kSeg));
  DevBuf dparts(c);
  if (chunks > 1 && dparts.alloc(nrows * (size_t)chunks * sizeof(ge_ext))) return VPIN_ENOMEM;
  // Row per lane (msm_strip_kernel) for a commitment of enough regular rows to keep every strip's block hot in L2 and the
  // chip full: the 2^25-constraint instance's 16384 x 16384 derefs polynomial (9.5 k of its rows qualify).  Smaller ones,
  // and a rank's share of a split commitment, stay with one workgroup per row.
  // VPIN_MSM_STRIP = 0: off, S: number of strips; VPIN_MSM_STRIP_MIN = n: take the path from n rows on (tests, A/B runs; read per
  // call so that a test can switch it)
  const char* es = getenv("VPIN_MSM_STRIP");
  const char* em = getenv("VPIN_MSM_STRIP_MIN");
  const int env_strip = es ? atoi(es) : -1;
  const size_t min_rows = em ? (size_t)atol(em) : 8192, min_list = em ? (size_t)1 : 4096, min_R = em ? (size_t)64 : 4096;
  int S = env_strip > 0 ? env_strip : 16;
  DevBuf dflag(c), dlist(c), dsparts(c), dsync(c);
  uint32_t n_strip = 0;
  bool uniform_shape = false;  // every strip workgroup of the launch is resident at once: they may wait for each other
  // Not on a shared device (vpin_ctx_set_shared_device: other contexts prove at the same time and the row kernels run one
  // workgroup per CU): with a single wave per SIMD nothing hides the strip kernel's per-generator chain scalar load ->
  // Montgomery conversion -> first gather, and its workgroups no longer step together: the default LeNet step of bench.py
  // measured 469-473 ms with it against 438-442 ms without (profiles/r04_ab_strip.txt), while the 2^25 instance proven alone
  // gains (313 -> 303 ms).  VPIN_MSM_STRIP > 0 forces it (A/B runs).
  const bool strip_shared = getenv("VPIN_MSM_STRIP_SHARED") != nullptr;  // A/B: also on a shared device, one workgroup per CU
  // Nor for a rank's share of a split commitment (c->comm set: at W = 2 the share of 16384 rows is exactly min_rows, and
  // rehearsal ranks share one GPU without saying so): its workgroups would not all be resident and the bounded wait below
  // would be sat out per generator.
  if (env_strip != 0 && (env_strip > 0 || ((!c->shared_device || strip_shared) && c->comm == nullptr)) && msm_ten_limbs() && nrows >= min_rows && R >= min_R && R <= g->split &&
      S % 8 == 0 && (size_t)S <= R / 8) {
    if (dflag.alloc(nrows) || dlist.alloc((nrows + 1) * sizeof(uint32_t))) return VPIN_ENOMEM;
    uint32_t* d_list = (uint32_t*)dlist.p;
    uint32_t* d_count = d_list + nrows;
    hipLaunchKernelGGL(strip_classify_kernel, dim3((unsigned)nrows), dim3(kMsmBlock), 0, c->stream, zbase, R, R, hr, row0, row_step, z_compact,
                       (uint8_t*)dflag.p);
    hipLaunchKernelGGL(strip_list_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint8_t*)dflag.p, nrows, d_list, d_count);
    VPIN_HIP_TRY(hipMemcpyAsync(&n_strip, d_count, sizeof n_strip, hipMemcpyDeviceToHost, c->stream));
    VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
    if (n_strip < min_list) n_strip = 0;  // too few rows to share a block's fetch: the row kernel takes everything
    if (n_strip && env_strip < 0) {
      // Every CU must run the SAME number of strip workgroups, or the workgroups of a strip drift apart (a CU with two of them
      // runs each 1.5x faster than a CU with three) and a table block is long evicted from L2 when the slow ones want it: the
      // PMC pass of the first version (608 workgroups on 768 slots) showed 95 B of L2 misses per addition, more than the row
      // kernel's 34 B.  The eligible rows beyond the chosen shape go back to the row kernel.
      // Choose the strips: every eligible row (G = ceil(n / 256) groups), S the largest multiple of 8 (a strip stays on one
      // XCD) with G x S workgroups all RESIDENT (3 per CU) -- they wait for each other (msm_strip_kernel's sync) -- and strips of
      // at least 128 generators.  Measured on the 2^25 instance (7844 eligible rows; profiles/r04_ab_strip.txt): free-running
      // workgroups 85.8 ms and 139 M FETCH_SIZE units per dispatch, one generator of slack 81.6 ms and 91 M, two 82.6 / 103, four
      // 85.0 / 111.
      const uint32_t found = n_strip;
      const size_t slots = (size_t)c->num_cus * (c->shared_device ? 1 : 3), G = ((size_t)n_strip + kMsmBlock - 1) / kMsmBlock;
      size_t bestG = 0;
      int bestS = 0;
      for (int cand = 8; cand <= 64 && (size_t)cand * 128 <= R; cand += 8)
        if (G * (size_t)cand <= slots) { bestG = G; bestS = cand; }
      // On a context confined to some of the CUs (vpin_ctx_create_cumask) "every eligible row" can leave the CUs unevenly
      // loaded (31 groups x 16 strips on 576 slots): take fewer groups and more strips when that fills more slots; the rows
      // beyond the shape go back to the row kernel.  VPIN_MSM_STRIP_SHAPE=G,S forces a shape (A/B runs).
      if (c->cu_masked)
        for (int cand = 8; cand <= 64 && (size_t)cand * 128 <= R; cand += 8) {
          const size_t g2 = std::min(G, slots / (size_t)cand);
          if (g2 * kMsmBlock >= min_list && g2 * (size_t)cand > bestG * (size_t)bestS) { bestG = g2; bestS = cand; }
        }
      if (const char* eshape = getenv("VPIN_MSM_STRIP_SHAPE")) {
        int eg = 0, esn = 0;
        if (sscanf(eshape, "%d,%d", &eg, &esn) == 2 && eg > 0 && esn > 0 && esn % 8 == 0 && (size_t)esn <= R / 8) {
          // the strip's workgroups wait for each other: a forced shape must be resident as a whole like the chosen ones
          if (std::min(G, (size_t)eg) * (size_t)esn <= slots) { bestG = std::min(G, (size_t)eg); bestS = esn; }
          else if (getenv("VPIN_MSM_STRIP_TRACE")) fprintf(stderr, "[strip] VPIN_MSM_STRIP_SHAPE=%d,%d refused: %zu workgroups on %zu slots\n", eg, esn, std::min(G, (size_t)eg) * (size_t)esn, slots);
        }
      }
      const uint32_t n_use = bestG ? (uint32_t)std::min<size_t>(n_strip, bestG * kMsmBlock) : 0;
      if (n_use == 0) {
        n_strip = 0;
      } else {
        if (n_use < n_strip)
          hipLaunchKernelGGL(strip_trim_kernel, dim3((n_strip - n_use + 255) / 256), dim3(256), 0, c->stream, (const uint32_t*)d_list, d_count,
                             n_use, n_strip, (uint8_t*)dflag.p);
        n_strip = n_use;
        S = bestS;
        uniform_shape = true;
      }
      if (getenv("VPIN_MSM_STRIP_TRACE")) fprintf(stderr, "[strip] %zu rows: %u eligible, %u taken as %zu groups x %d strips on %zu slots\n", nrows, found, n_use, bestG, bestS, slots);
    }
    if (n_strip && dsparts.alloc((size_t)n_strip * (size_t)S * sizeof(ge_ext))) return VPIN_ENOMEM;
  }
  const uint8_t* d_flag = n_strip ? (const uint8_t*)dflag.p : nullptr;
  {
    ProfScope ps(c, VPIN_K_MSM, 32.0 * (double)(nrows * R), VPIN_K_MSM_ROWS);
    static const int env_pad = [] { const char* e = getenv("VPIN_MSM_LDS_PAD"); return e ? atoi(e) : -1; }();
    const unsigned pad = env_pad >= 0 ? (unsigned)env_pad : (c->shared_device ? kSharedPad : 0u);
    if (n_strip) {
      c->strip_rows_taken += n_strip;
      const unsigned groups = (unsigned)((n_strip + kMsmBlock - 1) / kMsmBlock);
      const size_t per = (R + (size_t)S - 1) / (size_t)S;
      const char* el = getenv("VPIN_MSM_STRIP_LAG");  // generators a workgroup may run ahead of its strip's slowest; 0: no sync
      const int lag = el ? atoi(el) : 1;
      uint32_t* d_sync = nullptr;
      if (lag > 0 && uniform_shape) {
        if (dsync.alloc((size_t)S * (per + 1) * sizeof(uint32_t))) return VPIN_ENOMEM;
        d_sync = (uint32_t*)dsync.p;
        VPIN_HIP_TRY(hipMemsetAsync(d_sync, 0, (size_t)S * (per + 1) * sizeof(uint32_t), c->stream));
      }
      hipLaunchKernelGGL(msm_strip_kernel, dim3(groups * (unsigned)S), dim3(kMsmBlock), pad, c->stream, zbase, R, R, view(g), hr,
                         (const uint32_t*)dlist.p, (const uint32_t*)dlist.p + nrows, S, row0, row_step, z_compact, (ge_ext*)dsparts.p,
                         d_sync, lag);
    }
    if (msm_ten_limbs())
      hipLaunchKernelGGL(msm_rows_hot_kernel<true>, dim3((unsigned)nrows, (unsigned)chunks), dim3(kMsmBlock), pad, c->stream, zbase,
                         R, R, view(g), hr, chunks > 1 ? (ge_ext*)dparts.p : (ge_ext*)dpts.p, row0, row_step, z_compact, d_flag);
    else
      hipLaunchKernelGGL(msm_rows_hot_kernel<false>, dim3((unsigned)nrows, (unsigned)chunks), dim3(kMsmBlock), pad, c->stream, zbase,
                         R, R, view(g), hr, chunks > 1 ? (ge_ext*)dparts.p : (ge_ext*)dpts.p, row0, row_step, z_compact, d_flag);
    if (chunks > 1)
      hipLaunchKernelGGL(ge_sum_chunks_kernel, dim3((unsigned)((nrows + 63) / 64)), dim3(64), 0, c->stream, (const ge_ext*)dparts.p, nrows,
                         chunks, (ge_ext*)dpts.p);
    if (n_strip)
      hipLaunchKernelGGL(strip_combine_kernel, dim3((n_strip + 63) / 64), dim3(64), 0, c->stream, (const ge_ext*)dsparts.p,
                         (const uint32_t*)dlist.p, (const uint32_t*)dlist.p + nrows, S, (ge_ext*)dpts.p);
  }
  hipLaunchKernelGGL(ge_compress_kernel, dim3((unsigned)((nrows + 63) / 64)), dim3(64), 0, c->stream, (const ge_ext*)dpts.p, nrows,
                     (fp*)dout.p, (fp*)nullptr);
  VPIN_HIP_TRY(hipGetLastError());
  VPIN_HIP_TRY(hipMemcpyAsync(out_compressed, dout.p, nrows * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

int hyrax_commit_rows_strided(vpin_ctx* c, const vpin_gens* g, const vpin_table* Z, size_t L, size_t row0, size_t nrows, size_t row_step,
                              uint8_t* out_compressed, const fq* z_rows) {
  if (!c || !g || !Z || (!Z->d && !z_rows) || !out_compressed || L == 0 || row_step == 0) return VPIN_EINVAL;
  if (nrows == 0) return VPIN_OK;
  if (Z->len % L != 0 || row0 + (nrows - 1) * row_step >= L) return VPIN_ESHAPE;
  const size_t R = Z->len / L;
  if (R > g->nb) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  DevBuf dpts(c), dout(c);
  if (dpts.alloc(nrows * sizeof(ge_ext)) || dout.alloc(nrows * 32)) return VPIN_ENOMEM;
  int rc = z_rows ? msm_rows(c, g, z_rows, nrows, R, R, nullptr, 0, 0, (ge_ext*)dpts.p)
                  : msm_rows(c, g, Z->d + row0 * R, nrows, row_step * R, R, nullptr, 0, 0, (ge_ext*)dpts.p);
  if (rc) return rc;
  hipLaunchKernelGGL(ge_compress_kernel, dim3((unsigned)((nrows + 63) / 64)), dim3(64), 0, c->stream, (const ge_ext*)dpts.p, nrows,
                     (fp*)dout.p, (fp*)nullptr);
  VPIN_HIP_TRY(hipGetLastError());
  VPIN_HIP_TRY(hipMemcpyAsync(out_compressed, dout.p, nrows * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

}  // namespace vpin

extern "C" {

int vpin_hyrax_commit_rows(vpin_ctx* c, const vpin_gens* g, const vpin_table* Z, size_t L, size_t row0, size_t nrows,
                           const uint8_t* blinds, size_t blind_base, uint8_t* out_compressed) {
  if (!c || !g || !Z || !Z->d || !out_compressed || L == 0 || nrows == 0) return VPIN_EINVAL;
  if (Z->len % L != 0 || row0 + nrows > L) return VPIN_ESHAPE;
  const size_t R = Z->len / L;
  if (R > g->nb || (blinds && blind_base >= g->nb)) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  DevBuf dbl(c), dpts(c), dout(c);
  if ((blinds && dbl.alloc(nrows * 32)) || dpts.alloc(nrows * sizeof(ge_ext)) || dout.alloc(nrows * 32)) return VPIN_ENOMEM;
  if (blinds) VPIN_HIP_TRY(hipMemcpyAsync(dbl.p, blinds, nrows * 32, hipMemcpyHostToDevice, c->stream));
  int rc = msm_rows(c, g, Z->d + row0 * R, nrows, R, R, blinds ? (const fq*)dbl.p : nullptr, blinds ? 1 : 0, blind_base, (ge_ext*)dpts.p);
  if (rc) return rc;
  hipLaunchKernelGGL(ge_compress_kernel, dim3((unsigned)((nrows + 63) / 64)), dim3(64), 0, c->stream, (const ge_ext*)dpts.p, nrows,
                     (fp*)dout.p, (fp*)nullptr);
  VPIN_HIP_TRY(hipGetLastError());
  VPIN_HIP_TRY(hipMemcpyAsync(out_compressed, dout.p, nrows * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

int vpin_hyrax_commit_pair(vpin_ctx* c, const vpin_gens* g, const vpin_table* Za, const vpin_table* Zb,
                           const uint8_t* blinds_a, const uint8_t* blinds_b, size_t L, size_t blind_base,
                           uint8_t* out_a, uint8_t* out_b, uint8_t* out_sum) {
  if (!c || !g || !Za || !Zb || !blinds_a || !blinds_b || !out_a || !out_b || !out_sum || L == 0) return VPIN_EINVAL;
  if (Za->len != Zb->len || Za->len % L != 0) return VPIN_ESHAPE;
  size_t R = Za->len / L;
  if (R > g->nb || blind_base >= g->nb) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  DevBuf dbl(c), dpts(c), dout(c);
  if (dbl.alloc(2 * L * 32) || dpts.alloc(3 * L * sizeof(ge_ext)) || dout.alloc(3 * L * 32)) return VPIN_ENOMEM;
  VPIN_HIP_TRY(hipMemcpyAsync(dbl.p, blinds_a, L * 32, hipMemcpyHostToDevice, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync((uint8_t*)dbl.p + L * 32, blinds_b, L * 32, hipMemcpyHostToDevice, c->stream));
  ge_ext* pts = (ge_ext*)dpts.p;
  int rc = msm_rows(c, g, Za->d, L, R, R, (const fq*)dbl.p, 1, blind_base, pts);
  if (rc) return rc;
  rc = msm_rows(c, g, Zb->d, L, R, R, (const fq*)dbl.p + L, 1, blind_base, pts + L);
  if (rc) return rc;
  hipLaunchKernelGGL(ge_add_rows_kernel, dim3((unsigned)((L + 63) / 64)), dim3(64), 0, c->stream, pts, pts + L, L, pts + 2 * L);
  hipLaunchKernelGGL(ge_compress_kernel, dim3((unsigned)((3 * L + 63) / 64)), dim3(64), 0, c->stream, (const ge_ext*)pts, 3 * L,
                     (fp*)dout.p, (fp*)nullptr);
  VPIN_HIP_TRY(hipGetLastError());
  VPIN_HIP_TRY(hipMemcpyAsync(out_a, dout.p, L * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync(out_b, (uint8_t*)dout.p + L * 32, L * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync(out_sum, (uint8_t*)dout.p + 2 * L * 32, L * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

// partial points of few-row MSMs: parts_xyzt must hold rows * vpin_gens_msm_parts_count(ncols) * 128 bytes
// at most kMaxParts per row: longer rows get a second, on-device summation stage (the host would
// otherwise add a thousand points per row for the 32k-generator evaluation proofs)
constexpr size_t kMaxParts = 128;  // the host adds up to 128 partial points per row (~20 us) rather than wait for a second kernel
static inline int wide_q(size_t ncols) { return ncols >= 8192 ? 4 : 1; }
static inline size_t raw_parts(size_t ncols) { const size_t per = (size_t)kWideScalars * wide_q(ncols); return (ncols + per - 1) / per; }
size_t vpin_gens_msm_parts_count(size_t ncols) { size_t n = raw_parts(ncols); return n < kMaxParts ? n : kMaxParts; }

int vpin_gens_msm_parts(vpin_ctx* c, const vpin_gens* g, const uint8_t* scalars_mont, size_t rows, size_t ncols,
                        uint8_t* parts_xyzt) {
  if (!c || !g || !scalars_mont || !parts_xyzt || rows == 0 || ncols == 0) return VPIN_EINVAL;
  if (ncols > g->nb) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  DevBuf ds(c);
  if (ds.alloc(rows * ncols * 32)) return VPIN_ENOMEM;
  VPIN_HIP_TRY(hipMemcpyAsync(ds.p, scalars_mont, rows * ncols * 32, hipMemcpyHostToDevice, c->stream));
  return vpin::gens_msm_parts_dev(c, g, (const fq*)ds.p, rows, ncols, parts_xyzt);
}

}  // extern "C"

// Split-phase form of vpin_hyrax_commit_pair for the host prover: `begin` enqueues the two big
// row MSMs (which do not depend on the blinds), the host draws the 2L blinds from its RandomTape
// while they run, `finish` adds blind_i * g[blind_base] per row, combines and compresses.
namespace vpin {
struct CommitPairState {
  DevBuf pts;  // [5L] : M_a | M_b | blind_a*h | blind_b*h (reused for sums) | a+b
  size_t L = 0;  // rows this state commits (a block of the polynomial's rows when the commitment is split across ranks)
  explicit CommitPairState(vpin_ctx* c) : pts(c) {}
};

int commit_pair_begin(vpin_ctx* c, const vpin_gens* g, const vpin_table* Za, const vpin_table* Zb, size_t L_all,
                      CommitPairState** out, size_t row0, size_t nrows, size_t row_step) {
  if (!c || !g || !Za || !Zb || !out || L_all == 0) return VPIN_EINVAL;
  if (Za->len != Zb->len || Za->len % L_all != 0) return VPIN_ESHAPE;
  size_t R = Za->len / L_all;
  if (R > g->nb) return VPIN_ESHAPE;
  if (nrows == (size_t)-1) { row0 = 0; nrows = L_all; row_step = 1; }
  if (nrows == 0 || row_step == 0 || row0 + (nrows - 1) * row_step >= L_all) return VPIN_ESHAPE;
  const size_t L = nrows;
  (void)hipSetDevice(c->device);
  CommitPairState* st = new (std::nothrow) CommitPairState(c);
  if (!st) return VPIN_ENOMEM;
  st->L = L;
  if (st->pts.alloc(5 * L * sizeof(ge_ext))) { delete st; return VPIN_ENOMEM; }
  ge_ext* pts = (ge_ext*)st->pts.p;
  int rc = msm_rows(c, g, Za->d + row0 * R, L, row_step * R, R, nullptr, 0, 0, pts);
  if (!rc) rc = msm_rows(c, g, Zb->d + row0 * R, L, row_step * R, R, nullptr, 0, 0, pts + L);
  if (rc) { delete st; return rc; }
  *out = st;
  return VPIN_OK;
}

int commit_pair_finish(vpin_ctx* c, const vpin_gens* g, CommitPairState* st, const uint8_t* blinds_a, const uint8_t* blinds_b,
                       size_t blind_base, uint8_t* out_a, uint8_t* out_b, uint8_t* out_sum) {
  if (!st) return VPIN_EINVAL;
  std::unique_ptr<CommitPairState> guard(st);
  if (!c || !g || !blinds_a || !blinds_b || !out_a || !out_b || !out_sum) return VPIN_EINVAL;
  if (blind_base >= g->nb) return VPIN_ESHAPE;
  const size_t L = st->L;
  DevBuf dbl(c), dout(c);
  if (dbl.alloc(2 * L * 32) || dout.alloc(3 * L * 32)) return VPIN_ENOMEM;
  VPIN_HIP_TRY(hipMemcpyAsync(dbl.p, blinds_a, L * 32, hipMemcpyHostToDevice, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync((uint8_t*)dbl.p + L * 32, blinds_b, L * 32, hipMemcpyHostToDevice, c->stream));
  ge_ext* pts = (ge_ext*)st->pts.p;
  // 2L single-scalar multiples of the blind generator
  hipLaunchKernelGGL(single_base_mul_kernel, dim3((unsigned)((2 * L + 7) / 8)), dim3(64), 0, c->stream, (const fq*)dbl.p, 2 * L,
                     view(g), blind_base, pts + 2 * L);
  const unsigned gb = (unsigned)((2 * L + 63) / 64);
  hipLaunchKernelGGL(ge_add_rows_kernel, dim3(gb), dim3(64), 0, c->stream, pts, pts + 2 * L, 2 * L, pts + 2 * L);  // a, b
  hipLaunchKernelGGL(ge_add_rows_kernel, dim3((unsigned)((L + 63) / 64)), dim3(64), 0, c->stream, pts + 2 * L, pts + 3 * L, L,
                     pts + 4 * L);  // a + b
  hipLaunchKernelGGL(ge_compress_kernel, dim3((unsigned)((3 * L + 63) / 64)), dim3(64), 0, c->stream,
                     (const ge_ext*)(pts + 2 * L), 3 * L, (fp*)dout.p, (fp*)nullptr);
  VPIN_HIP_TRY(hipGetLastError());
  VPIN_HIP_TRY(hipMemcpyAsync(out_a, dout.p, L * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync(out_b, (uint8_t*)dout.p + L * 32, L * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync(out_sum, (uint8_t*)dout.p + 2 * L * 32, L * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}
}  // namespace vpin

extern "C" {

int vpin_gens_msm(vpin_ctx* c, const vpin_gens* g, const uint8_t* scalars_mont, size_t rows, size_t ncols,
                  uint8_t* out_compressed, uint8_t* out_xyzt) {
  if (!c || !g || !scalars_mont || rows == 0 || ncols == 0 || (!out_compressed && !out_xyzt)) return VPIN_EINVAL;
  if (ncols > g->nb) return VPIN_ESHAPE;  // dalek asserts equal lengths (group.rs:105)
  (void)hipSetDevice(c->device);
  DevBuf ds(c), dpts(c), dout(c), dx(c);
  if (ds.alloc(rows * ncols * 32) || dpts.alloc(rows * sizeof(ge_ext)) || dout.alloc(rows * 32) || dx.alloc(rows * 128))
    return VPIN_ENOMEM;
  VPIN_HIP_TRY(hipMemcpyAsync(ds.p, scalars_mont, rows * ncols * 32, hipMemcpyHostToDevice, c->stream));
  int rc = msm_rows(c, g, (const fq*)ds.p, rows, ncols, ncols, nullptr, 0, 0, (ge_ext*)dpts.p);
  if (rc) return rc;
  hipLaunchKernelGGL(ge_compress_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, c->stream, (const ge_ext*)dpts.p,
                     rows, out_compressed ? (fp*)dout.p : (fp*)nullptr, out_xyzt ? (fp*)dx.p : (fp*)nullptr);
  VPIN_HIP_TRY(hipGetLastError());
  if (out_compressed) VPIN_HIP_TRY(hipMemcpyAsync(out_compressed, dout.p, rows * 32, hipMemcpyDeviceToHost, c->stream));
  if (out_xyzt) VPIN_HIP_TRY(hipMemcpyAsync(out_xyzt, dx.p, rows * 128, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

}  // extern "C"

namespace vpin {

// asynchronous form: kernels and the device-to-host copy are enqueued on the context's stream; `scratch` must hold
// rows * (raw_parts(ncols) + nparts) * 128 bytes and outlive them (gens_msm_parts_scratch_bytes)
size_t gens_msm_parts_scratch_bytes(size_t rows, size_t ncols) { return rows * (raw_parts(ncols) + vpin_gens_msm_parts_count(ncols)) * 128; }
int gens_msm_parts_launch(vpin_ctx* c, const vpin_gens* g, const fq* d_scalars, size_t rows, size_t ncols, void* scratch,
                          uint8_t* parts_xyzt, bool host_mapped) {
  if (!c || !g || !d_scalars || !parts_xyzt || !scratch || rows == 0 || ncols == 0) return VPIN_EINVAL;
  if (ncols > g->nb) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  const size_t nraw = raw_parts(ncols), nparts = vpin_gens_msm_parts_count(ncols);
  // host_mapped: parts_xyzt is pinned host memory, which the device addresses directly -- the last stage stores its
  // partial points there and the ~10 us copy command per call goes away (the caller waits for the stream instead)
  fp* dp = (fp*)scratch;
  fp* dr = (fp*)((uint8_t*)scratch + rows * nraw * 128);
  fp* first_out = (host_mapped && nraw == nparts) ? (fp*)parts_xyzt : dp;
  {
    ProfScope ps(c, VPIN_K_MSM, 32.0 * (double)rows * (double)ncols);
    if (wide_q(ncols) == 4)
      hipLaunchKernelGGL((msm_wide_kernel<4>), dim3((unsigned)nraw, (unsigned)rows), dim3(kMsmBlock), 0, c->stream, d_scalars, ncols,
                         view(g), first_out);
    else
      hipLaunchKernelGGL((msm_wide_kernel<1>), dim3((unsigned)nraw, (unsigned)rows), dim3(kMsmBlock), 0, c->stream, d_scalars, ncols,
                         view(g), first_out);
  }
  const void* src = first_out;
  if (nraw > nparts) {
    fp* second_out = host_mapped ? (fp*)parts_xyzt : dr;
    hipLaunchKernelGGL(parts_reduce_kernel, dim3((unsigned)((rows * nparts + 63) / 64)), dim3(64), 0, c->stream, (const fp*)dp, rows,
                       nraw, (int)nparts, second_out);
    src = second_out;
  }
  VPIN_HIP_TRY(hipGetLastError());
  if (src != (const void*)parts_xyzt) VPIN_HIP_TRY(hipMemcpyAsync(parts_xyzt, src, rows * nparts * 128, hipMemcpyDeviceToHost, c->stream));
  return VPIN_OK;
}

// one launch per bullet round (bullet_step_kernel); parts_pinned / up_pinned are pinned host memory.  R must be a multiple
// of 32.  R > 4096 (more than 128 MSM workgroups): dev_parts (2 x R/32 x 128 bytes of device memory) takes the workgroups'
// points and a second launch leaves R/4096 sums per side in parts_pinned (bullet_parts_reduce_kernel).
int bullet_step_launch(vpin_ctx* c, const vpin_gens* g, const fq* a_prev, const fq* b_prev, fq* a_next, fq* b_next, fq* sj, size_t n,
                       size_t R, bool fold, bool finish, const uint8_t* u, const uint8_t* u_inv, uint8_t* parts_pinned,
                       uint32_t* up_pinned, uint32_t seq, void* dev_parts) {
  if (!c || !g || !a_prev || !b_prev || !sj || !parts_pinned || !up_pinned || R % kWideScalars || R > g->nb) return VPIN_EINVAL;
  if (!finish && (n == 0 || 2 * n > R || !a_next || !b_next)) return VPIN_EINVAL;
  if ((fold || finish) && (!u || !u_inv)) return VPIN_EINVAL;
  const int nmsm = (int)(R / kWideScalars);
  const bool reduce = nmsm > 128;
  if (reduce && (!dev_parts || (nmsm & (nmsm - 1)))) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  BulletStep a{};
  a.a_prev = a_prev; a.b_prev = b_prev; a.a_next = a_next; a.b_next = b_next; a.sj = sj;
  a.n = finish ? 0 : n; a.R = R; a.fold = fold ? 1 : 0; a.finish = finish ? 1 : 0;
  a.nmsm = nmsm;
  if (u) { memcpy(a.u.v, u, 32); memcpy(a.u_inv.v, u_inv, 32); }
  a.parts = reduce ? (fp*)dev_parts : (fp*)parts_pinned; a.up = up_pinned; a.seq = seq;
  const int nip = finish ? 1 : (int)((n + kMsmBlock - 1) / kMsmBlock);
  {
    ProfScope ps(c, VPIN_K_MSM, 32.0 * 2.0 * (double)R);
    hipLaunchKernelGGL(bullet_step_kernel, dim3((unsigned)(a.nmsm + nip)), dim3(kMsmBlock), 0, c->stream, a, view(g));
  }
  if (reduce)
    hipLaunchKernelGGL(bullet_parts_reduce_kernel, dim3((unsigned)(nmsm / 128), finish ? 1u : 2u), dim3(kMsmBlock), 0, c->stream,
                       (const fp*)dev_parts, nmsm, a.n, a.finish, (fp*)parts_pinned);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}

int gens_msm_parts_dev(vpin_ctx* c, const vpin_gens* g, const fq* d_scalars, size_t rows, size_t ncols, uint8_t* parts_xyzt) {
  if (!c || !g || !d_scalars || !parts_xyzt || rows == 0 || ncols == 0) return VPIN_EINVAL;
  if (ncols > g->nb) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  const size_t nraw = raw_parts(ncols), nparts = vpin_gens_msm_parts_count(ncols);
  DevBuf dp(c), dr(c);
  if (dp.alloc(rows * nraw * 128) || (nraw > nparts && dr.alloc(rows * nparts * 128))) return VPIN_ENOMEM;
  {
    ProfScope ps(c, VPIN_K_MSM, 32.0 * (double)rows * (double)ncols);
    if (wide_q(ncols) == 4)
      hipLaunchKernelGGL((msm_wide_kernel<4>), dim3((unsigned)nraw, (unsigned)rows), dim3(kMsmBlock), 0, c->stream, d_scalars, ncols,
                         view(g), (fp*)dp.p);
    else
      hipLaunchKernelGGL((msm_wide_kernel<1>), dim3((unsigned)nraw, (unsigned)rows), dim3(kMsmBlock), 0, c->stream, d_scalars, ncols,
                         view(g), (fp*)dp.p);
  }
  const void* src = dp.p;
  if (nraw > nparts) {
    hipLaunchKernelGGL(parts_reduce_kernel, dim3((unsigned)((rows * nparts + 63) / 64)), dim3(64), 0, c->stream, (const fp*)dp.p, rows,
                       nraw, (int)nparts, (fp*)dr.p);
    src = dr.p;
  }
  VPIN_HIP_TRY(hipGetLastError());
  VPIN_HIP_TRY(hipMemcpyAsync(parts_xyzt, src, rows * nparts * 128, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

}  // namespace vpin
