// msm_pip.hip -- row commitments by Pippenger's bucket method staged in LDS: the MSM BASELINE.json's north_star names, built
// as a measured alternative to the window-table walk of msm.hip (vpin_hyrax_commit_pippenger; NOT on the default path: see
// profiles/r05_pippenger.txt and DESIGN.md section 4 for why the table walk stays).
//
// What it computes: DensePolynomial::commit_inner's rows (Spartan/src/dense_mlpoly.rs:160-175), each
// GroupElement::vartime_multiscalar_mul (Spartan/src/group.rs:103-122: dalek's Pippenger above 190 terms) of R scalars over
// the same R generators (Spartan/src/commitments.rs:93-98), from the generators ALONE -- no window table.
//
// Shape on gfx950 (one workgroup of 256 lanes per (row, window) pair, a persistent grid of three workgroups per CU):
//   pip_recode_kernel   canonical scalars -> W = ceil(253 / c) signed c-bit digits, u16 each (sign in bit 15), laid out
//                       [row][window][column] so that a (row, window) pair reads 2 R contiguous bytes
//   pip_window_kernel   1. counting sort of the pair's non-zero digits in LDS: histogram over the 2^(c-1) buckets with LDS
//                          atomics, block scan, scatter of (column, sign) as u16 into an LDS list ordered by bucket
//                       2. lane t owns the K = 2^(c-1) / 256 consecutive buckets [tK, tK + K): it walks its part of the list
//                          from the top bucket down and adds the generator of every entry (96-byte affine entry, gathered
//                          from a compact array that stays in L2) to ONE accumulator in registers -- never reset, so after
//                          bucket i it holds the running sum B_{K-1} + .. + B_i of the running-sum trick; those K - 1
//                          intermediate values go to a per-lane scratch slot, their sum is the lane's weighted sum
//                          W_t = sum_i (i + 1) B_i, the last one its plain sum S_t.  A HEAVY bucket (more than max(32, n / 128)
//                          entries, twice a lane's share of a long row: the narrow top window's, a witness row's ones) is summed by all lanes together
//                          first and enters its owner's walk as one point
//                       3. across lanes: window sum = sum_t (W_t + tK S_t) = sum_t W_t + K sum_{t >= 1} Suf_t with
//                          Suf_t = S_t + S_{t+1} + ..: a suffix scan of the S_t through LDS (8 steps), log2 K doublings,
//                          one addition, and the block tree of ge_tree_dev.h
//   pip_finish_kernel   row = sum_w 2^(cw) (window sum w): Horner with c doublings per window, one lane per row
// Bucket sums live in registers; LDS stages the sort and the cross-lane reduction (40 KB at R = 16384, c = 11: three
// workgroups per CU).  2^(c-1) extended points of 160 B in LDS per pair would allow one workgroup per CU at c = 11.
//
// The order in which a bucket's points are added depends on the LDS atomics' order: the SUM is the same group element, its
// projective representation is not -- the compressed encoding (what a commitment is) is unique.
#include <algorithm>
#include <cstdlib>

#include "ctx.h"
#include "fp_dev.h"
#include "fp10_dev.h"
#include "ge_tree_dev.h"

namespace vpin {

namespace {
constexpr int kPipBlock = 256;
constexpr int kLimbs = 40;  // a ge10: four coordinates of ten limbs

template <int C>
struct PipShape {
  static constexpr int B = 1 << (C - 1);   // buckets: digits 1 .. 2^(c-1)
  static constexpr int K = B / kPipBlock;  // buckets per lane
  static constexpr int W = (253 + C - 1) / C;
  static_assert(K >= 1, "at least one bucket per lane");
};

// ---- digits -------------------------------------------------------------------------------------------------------------
// dig[(row W + w) n + j] = signed digit w of scalar (row, j): |d| in bits 0..14, sign in bit 15, 0 for a zero digit.
// Scalars j < ncols come from Z (row stride `stride`), j == ncols (n_extra = 1) from extra[row] (the blind).
// A canonical scalar is < 2^253, so the top window takes no carry out; *bad is set if a top digit exceeds 2^(c-1).
template <int C>
__global__ __launch_bounds__(kPipBlock) void pip_recode_kernel(const fq* __restrict__ Z, size_t stride, size_t ncols,
                                                               const fq* __restrict__ extra, int n_extra, size_t rows,
                                                               uint16_t* __restrict__ dig, uint32_t* __restrict__ bad) {
  constexpr int W = PipShape<C>::W;
  constexpr uint32_t B = PipShape<C>::B, mask = (1u << C) - 1u;
  const size_t n = ncols + (size_t)n_extra;
  const size_t idx = (size_t)blockIdx.x * kPipBlock + threadIdx.x;
  if (idx >= rows * n) return;
  const size_t row = idx / n, j = idx - row * n;
  fq s = fq_from_mont(fq_load(j < ncols ? Z + row * stride + j : extra + row));
  uint16_t* out = dig + (row * W) * n + j;
  uint32_t carry = 0;
#pragma unroll
  for (int w = 0; w < W; w++) {
    uint32_t v = (s.v[0] & mask) + carry;
#pragma unroll
    for (int i = 0; i < 7; i++) s.v[i] = __builtin_amdgcn_alignbit(s.v[i + 1], s.v[i], C);
    s.v[7] >>= C;
    bool neg = false;
    if (w + 1 < W) {
      neg = v > B;
      if (neg) v = (mask + 1u) - v;
      carry = neg ? 1u : 0u;
    } else if (v > B) {
      *bad = 1u;
      v = 0;
    }
    out[(size_t)w * n] = (uint16_t)(v | (neg && v ? 0x8000u : 0u));
  }
}

// ---- buckets ------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ ge_niels gn_load(const ge_niels* __restrict__ p) {
  ge_niels e;
  e.ypx = fp_load(&p->ypx); e.ymx = fp_load(&p->ymx); e.xy2d = fp_load(&p->xy2d);
  return e;
}
__device__ __forceinline__ void ge10_store_strided(uint32_t* __restrict__ p, const ge10& a) {  // limb l at p[256 l]
#pragma unroll
  for (int l = 0; l < 10; l++) {
    p[(size_t)l * kPipBlock] = a.X.v[l]; p[(size_t)(10 + l) * kPipBlock] = a.Y.v[l];
    p[(size_t)(20 + l) * kPipBlock] = a.Z.v[l]; p[(size_t)(30 + l) * kPipBlock] = a.T.v[l];
  }
}
__device__ __forceinline__ ge10 ge10_load_strided(const uint32_t* __restrict__ p) {
  ge10 a;
#pragma unroll
  for (int l = 0; l < 10; l++) {
    a.X.v[l] = p[(size_t)l * kPipBlock]; a.Y.v[l] = p[(size_t)(10 + l) * kPipBlock];
    a.Z.v[l] = p[(size_t)(20 + l) * kPipBlock]; a.T.v[l] = p[(size_t)(30 + l) * kPipBlock];
  }
  return a;
}

__device__ __forceinline__ ge10 ge10_load_dense(const uint32_t* __restrict__ p) {  // 40 consecutive words
  ge10 a;
#pragma unroll
  for (int l = 0; l < 10; l++) { a.X.v[l] = p[l]; a.Y.v[l] = p[10 + l]; a.Z.v[l] = p[20 + l]; a.T.v[l] = p[30 + l]; }
  return a;
}
__device__ __forceinline__ void ge10_store_dense(uint32_t* __restrict__ p, const ge10& a) {
#pragma unroll
  for (int l = 0; l < 10; l++) { p[l] = a.X.v[l]; p[10 + l] = a.Y.v[l]; p[20 + l] = a.Z.v[l]; p[30 + l] = a.T.v[l]; }
}
__device__ __forceinline__ ge10 ge10_shfl_xor(const ge10& a, int m) {
  ge10 r;
#pragma unroll
  for (int l = 0; l < 10; l++) {
    r.X.v[l] = __shfl_xor(a.X.v[l], m, 64); r.Y.v[l] = __shfl_xor(a.Y.v[l], m, 64);
    r.Z.v[l] = __shfl_xor(a.Z.v[l], m, 64); r.T.v[l] = __shfl_xor(a.T.v[l], m, 64);
  }
  return r;
}

constexpr int kHeavyMax = 128;  // a heavy bucket holds more than n / 128 entries: fewer than 128 of them

// dynamic LDS: max(2 B u32 counters + n u16 list entries, 40 KB for the reduction)
template <int C>
size_t pip_lds_bytes(size_t n) {
  const size_t sort = (size_t)2 * PipShape<C>::B * 4 + ((n * 2 + 15) & ~(size_t)15);
  return std::max(sort, (size_t)kLimbs * kPipBlock * 4);
}
// words of scratch per workgroup: the lanes' K - 1 running sums (limb-strided) + the heavy buckets' sums (dense)
template <int C>
constexpr size_t pip_scratch_words() { return (size_t)(PipShape<C>::K - 1) * kLimbs * kPipBlock + (size_t)kHeavyMax * kLimbs; }

// items = rows * W (row, window) pairs, pair `item` reads dig + item * n; wsum[item] <- its window sum;
// scratch: gridDim.x * pip_scratch_words<C>() u32; adds: optional counter of bucket additions (non-zero digits)
//
// HEAVY buckets (more than max(32, n / 128) entries -- twice a lane's average share): a lane that owned one alone would hold
// its whole wave back.  The narrow top window (253 - c (W - 1) bits: two buckets at c = 9 or 12) puts a whole row into a few
// buckets, and a witness row puts its ones into bucket 1.  Their entries go behind the light buckets' in the list; all 256
// lanes add a stride of each heavy run, a shuffle tree and four wave partials give its sum, and the owner adds that ONE point
// where its walk passes the bucket.
template <int C>
__global__ __launch_bounds__(kPipBlock, 3) void pip_window_kernel(const uint16_t* __restrict__ dig, const ge_niels* __restrict__ gn,
                                                                uint32_t n, size_t items, uint32_t* __restrict__ scratch,
                                                                ge_ext* __restrict__ wsum, unsigned long long* __restrict__ adds) {
  constexpr int B = PipShape<C>::B, K = PipShape<C>::K;
  extern __shared__ __align__(16) uint32_t lds[];
  uint32_t* start = lds;       // [B] counts, then the first list position of every bucket (light layout)
  uint32_t* cursor = lds + B;  // [B] scatter cursors; after the scatter: heavy slot + 1 of the bucket, 0 for a light one
  uint16_t* sorted = reinterpret_cast<uint16_t*>(lds + 2 * B);  // [n] (column | sign << 15), ordered by bucket
  __shared__ uint32_t wave_tot[kPipBlock / 64];
  __shared__ uint32_t hcount, hfill;
  __shared__ uint32_t hoff[kHeavyMax], hlen[kHeavyMax];
  __shared__ uint32_t wave_pt[(kPipBlock / 64) * kLimbs];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  uint32_t* wg_scr = scratch + (size_t)blockIdx.x * pip_scratch_words<C>();
  uint32_t* my_scr = wg_scr + t;                                          // running sum i at my_scr + (i - 1) 40 * 256, limb-strided
  uint32_t* hsum = wg_scr + (size_t)(K - 1) * kLimbs * kPipBlock;         // heavy sum h at hsum + 40 h
  const uint32_t thr = n / 128 > 32 ? n / 128 : 32;

  auto fetch = [&](uint32_t e, bool& neg) {
    const uint32_t v = sorted[e];
    neg = (v >> 15) != 0;
    return gn_load(gn + (v & 0x7fffu));
  };

  for (size_t item = blockIdx.x; item < items; item += gridDim.x) {
    const uint16_t* d = dig + item * n;
    // 1. counting sort
    for (int i = t; i < B; i += kPipBlock) start[i] = 0;
    if (t == 0) { hcount = 0; hfill = 0; }
    __syncthreads();
    for (uint32_t j = t; j < n; j += kPipBlock) {
      const uint32_t m = d[j] & 0x7fffu;
      if (m) atomicAdd(&start[m - 1], 1u);
    }
    __syncthreads();
    uint32_t loc[K], hpos[K], mine = 0;
#pragma unroll
    for (int i = 0; i < K; i++) {
      loc[i] = start[t * K + i];
      hpos[i] = 0xffffffffu;
      if (loc[i] > thr) {
        const uint32_t slot = atomicAdd(&hcount, 1u);
        const uint32_t off = atomicAdd(&hfill, loc[i]);
        hoff[slot] = off;
        hlen[slot] = loc[i];
        hpos[i] = slot;
      } else {
        mine += loc[i];
      }
    }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    uint32_t beg = incl - mine, total_light = 0;
#pragma unroll
    for (int k = 0; k < kPipBlock / 64; k++) {
      if (k < wv) beg += wave_tot[k];
      total_light += wave_tot[k];
    }
    const uint32_t nh = hcount, total = total_light + hfill;
    if (total == 0) {  // an empty window (the upper windows of a row of bits): the identity, uniform for the workgroup
      if (t == 0) {
        ge_ext* o = wsum + item;
        fp_store(&o->X, fp_zero()); fp_store(&o->Y, fp_one()); fp_store(&o->Z, fp_one()); fp_store(&o->T, fp_zero());
      }
      __syncthreads();
      continue;
    }
    {
      uint32_t pos = beg;
#pragma unroll
      for (int i = 0; i < K; i++) {
        start[t * K + i] = pos;
        if (hpos[i] != 0xffffffffu) {
          cursor[t * K + i] = total_light + hoff[hpos[i]];
        } else {
          cursor[t * K + i] = pos;
          pos += loc[i];
        }
      }
    }
    const uint32_t end = beg + mine;
    __syncthreads();
    for (uint32_t j = t; j < n; j += kPipBlock) {
      const uint32_t v = d[j], m = v & 0x7fffu;
      if (m) sorted[atomicAdd(&cursor[m - 1], 1u)] = (uint16_t)(j | (v & 0x8000u));
    }
    __syncthreads();
    if (adds && t == 0) atomicAdd(adds, (unsigned long long)total);
#pragma unroll
    for (int i = 0; i < K; i++) cursor[t * K + i] = hpos[i] != 0xffffffffu ? hpos[i] + 1u : 0u;  // the lane's own buckets only

    // 2a. the heavy buckets, all lanes together
    for (uint32_t h = 0; h < nh; h++) {
      const uint32_t lo = total_light + hoff[h], hi = lo + hlen[h];
      ge10 a = ge10_identity();
#pragma unroll 1
      for (uint32_t e = lo + t; e < hi; e += kPipBlock) {
        bool neg;
        const ge_niels q = fetch(e, neg);
        a = ge10_add_niels(a, q, neg);
      }
#pragma unroll 1
      for (int m = 32; m >= 1; m >>= 1) a = ge10_add_ge10(a, ge10_shfl_xor(a, m));
      if (lane == 0) ge10_store_dense(wave_pt + wv * kLimbs, a);
      __syncthreads();
      if (t == 0) {
        ge10 s = a;
#pragma unroll 1
        for (int k = 1; k < kPipBlock / 64; k++) s = ge10_add_ge10(s, ge10_load_dense(wave_pt + k * kLimbs));
        ge10_store_dense(hsum + (size_t)h * kLimbs, s);
      }
      __syncthreads();  // the sum is visible to its owner; wave_pt may be rewritten
    }

    // 2b. the lane's light buckets, top down, into one accumulator
    ge10 acc = ge10_identity();
    {
      int e = (int)end - 1, b = K - 1;
      uint32_t lo = K > 1 ? start[t * K + b] : beg;
      ge_niels cur;
      bool neg_cur = false;
      if (e >= (int)beg) cur = fetch((uint32_t)e, neg_cur);
#pragma unroll 1
      while (e >= (int)beg) {
        if (K > 1) {
          while ((uint32_t)e < lo) {  // bucket b is complete: acc = B_{K-1} + .. + B_b (b >= 1 here: lo of bucket 0 is beg)
            if (const uint32_t s = cursor[t * K + b]) acc = ge10_add_ge10(acc, ge10_load_dense(hsum + (size_t)(s - 1) * kLimbs));
            ge10_store_strided(my_scr + (size_t)(b - 1) * kLimbs * kPipBlock, acc);
            b--;
            lo = start[t * K + b];
          }
        }
        ge_niels nxt = cur;
        bool neg_nxt = false;
        if (e - 1 >= (int)beg) nxt = fetch((uint32_t)(e - 1), neg_nxt);  // requested before this addition
        acc = ge10_add_niels(acc, cur, neg_cur);
        cur = nxt;
        neg_cur = neg_nxt;
        e--;
      }
      if (K > 1) {
#pragma unroll 1
        for (; b >= 1; b--) {  // the buckets below hold no light entries
          if (const uint32_t s = cursor[t * K + b]) acc = ge10_add_ge10(acc, ge10_load_dense(hsum + (size_t)(s - 1) * kLimbs));
          ge10_store_strided(my_scr + (size_t)(b - 1) * kLimbs * kPipBlock, acc);
        }
      }
      if (const uint32_t s = cursor[t * K]) acc = ge10_add_ge10(acc, ge10_load_dense(hsum + (size_t)(s - 1) * kLimbs));
    }
    ge10 weighted = acc;
    if (K > 1) {
#pragma unroll 1
      for (int i = 1; i < K; i++) weighted = ge10_add_ge10(weighted, ge10_load_strided(my_scr + (size_t)(i - 1) * kLimbs * kPipBlock));
    }
    __syncthreads();  // the list and the counters are dead: LDS becomes the reduction's

    // 3. across lanes: Suf_t = S_t + S_{t+1} + .. (Hillis-Steele over the ten-limb form, limb l of lane t at lds[256 l + t])
    ge10 suf = acc;
    ge10_store_strided(lds + t, suf);
    __syncthreads();
#pragma unroll 1
    for (int dlt = 1; dlt < kPipBlock; dlt <<= 1) {
      const bool has = t + dlt < kPipBlock;
      ge10 other = suf;
      if (has) other = ge10_load_strided(lds + t + dlt);
      __syncthreads();
      if (has) {
        suf = ge10_add_ge10(suf, other);
        ge10_store_strided(lds + t, suf);
      }
      __syncthreads();
    }
    ge10 u = suf;
    for (int k = 1; k < K; k <<= 1) u = ge10_double(u);  // K Suf_t
    u = ge10_add_ge10(weighted, u);
    const ge_ext ue = ge10_to_ext(t == 0 ? weighted : u);  // lane 0's own buckets carry no offset
    ge_ext* sh = reinterpret_cast<ge_ext*>(lds);
    __syncthreads();
    sh[t] = ue;
    __syncthreads();
    ge_tree_quad(sh, kPipBlock);
    if (t == 0) {
      const ge_ext r = sh[0];
      ge_ext* o = wsum + item;
      fp_store(&o->X, r.X); fp_store(&o->Y, r.Y); fp_store(&o->Z, r.Z); fp_store(&o->T, r.T);
    }
    __syncthreads();
  }
}

// out[row] = sum_w 2^(C w) wsum[row W + w]
template <int C>
__global__ __launch_bounds__(64) void pip_finish_kernel(const ge_ext* __restrict__ wsum, size_t rows, ge_ext* __restrict__ out) {
  constexpr int W = PipShape<C>::W;
  const size_t row = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (row >= rows) return;
  auto load = [&](int w) {
    const ge_ext* p = wsum + row * W + w;
    ge_ext e;
    e.X = fp_load(&p->X); e.Y = fp_load(&p->Y); e.Z = fp_load(&p->Z); e.T = fp_load(&p->T);
    return ge10_from_ext(e);
  };
  ge10 acc = load(W - 1);
#pragma unroll 1
  for (int w = W - 2; w >= 0; w--) {
#pragma unroll 1
    for (int k = 0; k < C; k++) acc = ge10_double(acc);
    acc = ge10_add_ge10(acc, load(w));
  }
  const ge_ext r = ge10_to_ext(acc);
  ge_ext* o = out + row;
  fp_store(&o->X, r.X); fp_store(&o->Y, r.Y); fp_store(&o->Z, r.Z); fp_store(&o->T, r.T);
}

template <int C>
int pip_rows_c(vpin_ctx* c, const ge_niels* d_gn, const fq* dZ, size_t rows, size_t stride, size_t ncols, const fq* d_extra, int n_extra,
               ge_ext* d_points, unsigned long long* d_adds) {
  constexpr int W = PipShape<C>::W;
  const size_t n = ncols + (size_t)n_extra;
  const size_t lds = pip_lds_bytes<C>(n);
  // The digits are 2 W bytes per scalar (58 at c = 9): rows are taken in chunks of at most ~2 GiB of digits -- still thousands
  // of (row, window) pairs per launch -- so that the 2^14 x 2^14 derefs polynomial of the 2^25 instance needs 2, not 16 GB
  // (VPIN_PIP_DIGIT_BYTES: another cap, read per call -- the tests force several chunks with it)
  const char* e_cap = getenv("VPIN_PIP_DIGIT_BYTES");
  const size_t cap = e_cap && atol(e_cap) > 0 ? (size_t)atol(e_cap) : (size_t)2 << 30;
  const size_t chunk_rows = std::min(rows, std::max<size_t>(1, cap / (W * n * 2)));
  const size_t chunk_items = chunk_rows * W;
  const unsigned grid = (unsigned)std::min<size_t>(chunk_items, (size_t)c->num_cus * 3);
  DevBuf b_dig(c), b_wsum(c), b_scr(c), b_bad(c);
  if (b_dig.alloc(chunk_items * n * 2) || b_wsum.alloc(chunk_items * sizeof(ge_ext)) || b_bad.alloc(4) ||
      b_scr.alloc((size_t)grid * pip_scratch_words<C>() * 4))
    return VPIN_ENOMEM;
  VPIN_HIP_TRY(hipMemsetAsync(b_bad.p, 0, 4, c->stream));
  // more than 64 KB of dynamic LDS needs the attribute (c = 12 with long rows)
  VPIN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&pip_window_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));
  {
    ProfScope ps(c, VPIN_K_MSM, 32.0 * (double)rows * (double)n, VPIN_K_MSM_ROWS);
    for (size_t r0 = 0; r0 < rows; r0 += chunk_rows) {
      const size_t nr = std::min(chunk_rows, rows - r0), items = nr * W;
      c->pip_row_chunks++;
      hipLaunchKernelGGL(pip_recode_kernel<C>, dim3((unsigned)((nr * n + kPipBlock - 1) / kPipBlock)), dim3(kPipBlock), 0, c->stream,
                         dZ + r0 * stride, stride, ncols, d_extra ? d_extra + r0 : d_extra, n_extra, nr, (uint16_t*)b_dig.p, (uint32_t*)b_bad.p);
      hipLaunchKernelGGL(pip_window_kernel<C>, dim3((unsigned)std::min<size_t>(items, grid)), dim3(kPipBlock), lds, c->stream,
                         (const uint16_t*)b_dig.p, d_gn, (uint32_t)n, items, (uint32_t*)b_scr.p, (ge_ext*)b_wsum.p, d_adds);
      hipLaunchKernelGGL(pip_finish_kernel<C>, dim3((unsigned)((nr + 63) / 64)), dim3(64), 0, c->stream, (const ge_ext*)b_wsum.p, nr,
                         d_points + r0);
    }
  }
  VPIN_HIP_TRY(hipGetLastError());
  uint32_t bad = 0;
  VPIN_HIP_TRY(hipMemcpyAsync(&bad, b_bad.p, 4, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return bad ? VPIN_EINVAL : VPIN_OK;  // a scalar that was not canonical Montgomery
}
}  // namespace

// window bits when the caller does not choose: measured (profiles/r05_pippenger.txt), c = 9 -- one bucket per lane, no
// intermediate running sums -- is the fastest at R = 4096 and R = 16384 although it makes the most additions (28.4 per scalar
// against 23.0 at c = 11); the paper count W (n + 2^c) / n would pick 10-11 (profiles/r04_msm_cost_model.md)
int pip_default_bits(size_t) { return 9; }

// rows x (ncols scalars of Z, row stride `stride`, + n_extra in {0, 1} scalars extra[row] on generator ncols) over the compact
// generator array d_gn (ncols + n_extra affine entries); d_points[row] <- the row's sum.  Synchronises.
int pip_rows(vpin_ctx* c, const ge_niels* d_gn, const fq* dZ, size_t rows, size_t stride, size_t ncols, const fq* d_extra, int n_extra,
             int cbits, ge_ext* d_points, unsigned long long* d_adds) {
  if (!c || !d_gn || !dZ || !d_points || rows == 0 || ncols == 0 || n_extra < 0 || n_extra > 1 || (n_extra && !d_extra)) return VPIN_EINVAL;
  const size_t n = ncols + (size_t)n_extra;
  if (n > 32768) return VPIN_ESHAPE;  // a list entry is a 15-bit column and a sign
  if (cbits == 0) cbits = pip_default_bits(n);
  switch (cbits) {
    case 9: return pip_rows_c<9>(c, d_gn, dZ, rows, stride, ncols, d_extra, n_extra, d_points, d_adds);
    case 10: return pip_rows_c<10>(c, d_gn, dZ, rows, stride, ncols, d_extra, n_extra, d_points, d_adds);
    case 11: return pip_rows_c<11>(c, d_gn, dZ, rows, stride, ncols, d_extra, n_extra, d_points, d_adds);
    case 12: return pip_rows_c<12>(c, d_gn, dZ, rows, stride, ncols, d_extra, n_extra, d_points, d_adds);
    default: return VPIN_EINVAL;
  }
}

}  // namespace vpin
