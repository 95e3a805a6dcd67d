// r1cs.hip -- the sparse R1CS matrices (A,B,C) on the device, gfx950.
//
// Replaces the three O(nnz) host loops that sit between the kernels of the sat proof:
//   SparseMatPolynomial::multiply_vec              Spartan/src/sparse_mlpoly.rs:467-481   (Az,Bz,Cz)
//   SparseMatPolynomial::compute_eval_table_sparse Spartan/src/sparse_mlpoly.rs:483-498   (A(rx,.),...)
//   SparseMatPolynomial::evaluate_with_tables      Spartan/src/sparse_mlpoly.rs:440-452   (A(rx,ry),...)
// The reference scatters (`Mz[row] += val*z[col]`); 32-byte field elements have no atomics, so
// the instance is kept twice in HBM -- row-sorted (CSR) for the SpMV, column-sorted (CSC) for
// the eval table -- and every output element is produced by exactly one thread (short
// rows/columns) or one workgroup (the few very long columns: the constant-1 column is touched
// by every gadget copy).  Results are field sums, so any order gives the same bits.
#include <algorithm>
#include <cstring>
#include <vector>

#include "ctx.h"
#include "r1cs_dev.h"

namespace vpin {

constexpr int kRB = 256;

__device__ __forceinline__ fq mul_special(const fq& val, const fq& x) {
  return fq_mul(val, x);
}

// out[row] = sum_k val[k] * z[col[k]]  over the row's CSR segment, for the 3 matrices
__global__ __launch_bounds__(kRB) void spmv_kernel(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ col,
                                                   const fq* __restrict__ val, const fq* __restrict__ z, size_t nrows,
                                                   fq* __restrict__ out) {
  size_t r = (size_t)blockIdx.x * kRB + threadIdx.x;
  if (r >= nrows) return;
  uint32_t k0 = rowptr[r], k1 = rowptr[r + 1];
  fq acc = fq_zero();
  for (uint32_t k = k0; k < k1; k++) {
    fq x = fq_load(z + col[k]);
    if (fq_is_zero(x)) continue;
    acc = fq_add(acc, fq_mul(fq_load(val + k), x));
  }
  fq_store(out + r, acc);
}

// The same for the rows r0, r0 + step, ..: out[k] = (M z)[r0 + k*step].  One sum-check over several GPUs (prover.cpp): the
// fold pairs (i, i + len/2) of DensePolynomial::bound_poly_var_top keep both members in one residue class mod a power of
// two, so rank r0 owns the rows = r0 (mod world) and never exchanges a table entry.
__global__ __launch_bounds__(kRB) void spmv_strided_kernel(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ col,
                                                           const fq* __restrict__ val, const fq* __restrict__ z, size_t nloc,
                                                           size_t r0, size_t step, fq* __restrict__ out) {
  size_t k = (size_t)blockIdx.x * kRB + threadIdx.x;
  if (k >= nloc) return;
  const size_t r = r0 + k * step;
  uint32_t k0 = rowptr[r], k1 = rowptr[r + 1];
  fq acc = fq_zero();
  for (uint32_t q = k0; q < k1; q++) {
    fq x = fq_load(z + col[q]);
    if (fq_is_zero(x)) continue;
    acc = fq_add(acc, fq_mul(fq_load(val + q), x));
  }
  fq_store(out + k, acc);
}

// dst[k] = src[r0 + k*step]
__global__ __launch_bounds__(kRB) void strided_take_kernel(const fq* __restrict__ src, size_t nloc, size_t r0, size_t step,
                                                           fq* __restrict__ dst) {
  size_t k = (size_t)blockIdx.x * kRB + threadIdx.x;
  if (k < nloc) fq_store(dst + k, fq_load(src + r0 + k * step));
}

// eval table at the columns r0, r0 + step, ..: out[k] (+)= rc * sum over column (r0 + k*step)
__global__ __launch_bounds__(kRB) void eval_table_strided_kernel(const uint32_t* __restrict__ colptr, const uint32_t* __restrict__ row,
                                                                 const fq* __restrict__ val, const fq* __restrict__ rx, size_t nloc,
                                                                 size_t r0, size_t step, fq rc, int accumulate, fq* __restrict__ out) {
  size_t k = (size_t)blockIdx.x * kRB + threadIdx.x;
  if (k >= nloc) return;
  const size_t c = r0 + k * step;
  uint32_t k0 = colptr[c], k1 = colptr[c + 1];
  fq acc = fq_zero();
  if (k1 - k0 <= kLongCol) {  // long columns: eval_table_long_finish_strided_kernel
    for (uint32_t q = k0; q < k1; q++) acc = fq_add(acc, fq_mul(fq_load(val + q), fq_load(rx + row[q])));
    if (k1 > k0) acc = fq_mul(acc, rc);
  }
  if (accumulate) acc = fq_add(acc, fq_load(out + k));
  fq_store(out + k, acc);
}

// partial eval table of one matrix scaled by its challenge: out[c] (+)= rc * sum_k val[k]*rx[row[k]]
// accumulate != 0 adds onto the existing out[c] (used to fold A, B, C into one table)
__global__ __launch_bounds__(kRB) void eval_table_kernel(const uint32_t* __restrict__ colptr, const uint32_t* __restrict__ row,
                                                         const fq* __restrict__ val, const fq* __restrict__ rx, size_t ncols,
                                                         fq rc, int accumulate, fq* __restrict__ out) {
  size_t c = (size_t)blockIdx.x * kRB + threadIdx.x;
  if (c >= ncols) return;
  uint32_t k0 = colptr[c], k1 = colptr[c + 1];
  if (k1 - k0 > kLongCol) return;  // long columns: eval_table_long_kernel
  fq acc = fq_zero();
  for (uint32_t k = k0; k < k1; k++) acc = fq_add(acc, fq_mul(fq_load(val + k), fq_load(rx + row[k])));
  if (k1 > k0) acc = fq_mul(acc, rc);
  if (accumulate) acc = fq_add(acc, fq_load(out + c));
  fq_store(out + c, acc);
}

// chunk partial: part[ch] = sum_{k in chunk} val[k] * rx[row[k]]
__global__ __launch_bounds__(kRB) void long_chunk_kernel(const uint32_t* __restrict__ chunk_k0, const uint32_t* __restrict__ chunk_k1,
                                                         const uint32_t* __restrict__ row, const fq* __restrict__ val,
                                                         const fq* __restrict__ rx, fq* __restrict__ part) {
  uint32_t k0 = chunk_k0[blockIdx.x], k1 = chunk_k1[blockIdx.x];
  fq acc = fq_zero();
  for (uint32_t k = k0 + threadIdx.x; k < k1; k += kRB) acc = fq_add(acc, fq_mul(fq_load(val + k), fq_load(rx + row[k])));
  __shared__ fq sh[kRB / 64];
  fq s = fq_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    fq t = sh[0];
    for (int w = 1; w < kRB / 64; w++) t = fq_add(t, sh[w]);
    fq_store(part + blockIdx.x, t);
  }
}

// one thread per long column: out[c] (+)= rc * sum of its chunk partials
// one WAVE per long column (the constant-1 column of a 2^25-constraint instance has thousands of chunks: one thread adding
// them up in a row took ~450 us per matrix)
__device__ __forceinline__ fq long_column_sum(const uint32_t* __restrict__ long_first, size_t i, const fq* __restrict__ part) {
  fq t = fq_zero();
  for (uint32_t ch = long_first[i] + threadIdx.x; ch < long_first[i + 1]; ch += 64) t = fq_add(t, fq_load(part + ch));
  return fq_wave_sum(t);
}

__global__ __launch_bounds__(64) void eval_table_long_finish_kernel(const uint32_t* __restrict__ long_cols,
                                                                    const uint32_t* __restrict__ long_first, size_t n_long,
                                                                    const fq* __restrict__ part, fq rc, int accumulate,
                                                                    fq* __restrict__ out) {
  const size_t i = blockIdx.x;
  if (i >= n_long) return;
  fq t = long_column_sum(long_first, i, part);
  if (threadIdx.x != 0) return;
  t = fq_mul(t, rc);
  uint32_t c = long_cols[i];
  if (accumulate) t = fq_add(t, fq_load(out + c));
  fq_store(out + c, t);
}

__global__ __launch_bounds__(64) void eval_table_long_finish_strided_kernel(const uint32_t* __restrict__ long_cols,
                                                                            const uint32_t* __restrict__ long_first, size_t n_long,
                                                                            const fq* __restrict__ part, fq rc, size_t r0, size_t step,
                                                                            fq* __restrict__ out) {
  const size_t i = blockIdx.x;
  if (i >= n_long) return;
  const uint32_t c = long_cols[i];
  if (c % step != r0) return;  // another rank's column
  fq t = long_column_sum(long_first, i, part);
  if (threadIdx.x != 0) return;
  t = fq_mul(t, rc);
  const size_t k = (c - r0) / step;
  fq_store(out + k, fq_add(t, fq_load(out + k)));  // the short-column kernel has written (or accumulated into) out[k] already
}

// one wave per long column: partials[i] = ry[c] * sum of its chunk partials
__global__ __launch_bounds__(64) void evaluate_long_finish_kernel(const uint32_t* __restrict__ long_cols,
                                                                  const uint32_t* __restrict__ long_first, size_t n_long,
                                                                  const fq* __restrict__ part, const fq* __restrict__ ry,
                                                                  fq* __restrict__ partials) {
  const size_t i = blockIdx.x;
  if (i >= n_long) return;
  const fq t = long_column_sum(long_first, i, part);
  if (threadIdx.x == 0) fq_store(partials + i, fq_mul(t, fq_load(ry + long_cols[i])));
}

// partials[block] = sum_k rx[row[k]] * ry[col[k]] * val[k]   (CSR order: row index recovered by search)
__global__ __launch_bounds__(kRB) void evaluate_kernel(const uint32_t* __restrict__ colptr, const uint32_t* __restrict__ row,
                                                       const fq* __restrict__ val, const fq* __restrict__ rx,
                                                       const fq* __restrict__ ry, size_t ncols, fq* __restrict__ partials) {
  // one thread per column of the CSC form: sum_k val*rx[row] then * ry[col]
  fq acc = fq_zero();
  for (size_t c = (size_t)blockIdx.x * kRB + threadIdx.x; c < ncols; c += (size_t)gridDim.x * kRB) {
    uint32_t k0 = colptr[c], k1 = colptr[c + 1];
    if (k1 == k0 || k1 - k0 > kLongCol) continue;
    fq s = fq_zero();
    for (uint32_t k = k0; k < k1; k++) s = fq_add(s, fq_mul(fq_load(val + k), fq_load(rx + row[k])));
    acc = fq_add(acc, fq_mul(s, fq_load(ry + c)));
  }
  __shared__ fq sh[kRB / 64];
  fq s = fq_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    fq t = sh[0];
    for (int w = 1; w < kRB / 64; w++) t = fq_add(t, sh[w]);
    fq_store(partials + blockIdx.x, t);
  }
}

__global__ __launch_bounds__(kRB) void sum_partials_kernel(const fq* __restrict__ partials, int n, fq* __restrict__ out) {
  fq acc = fq_zero();
  for (int i = threadIdx.x; i < n; i += kRB) acc = fq_add(acc, fq_load(partials + i));
  __shared__ fq sh[kRB / 64];
  fq s = fq_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    fq t = sh[0];
    for (int w = 1; w < kRB / 64; w++) t = fq_add(t, sh[w]);
    fq_store(out, t);
  }
}

// z = [vars | 1 | inputs | 0...] (commit_test.rs:162-170): fill the upper half
__global__ __launch_bounds__(kRB) void build_z_hi_kernel(fq* __restrict__ z, size_t nv, const fq* __restrict__ inputs, size_t ni) {
  size_t i = (size_t)blockIdx.x * kRB + threadIdx.x;
  if (i >= nv) return;
  fq v = fq_zero();
  if (i == 0) v = fq_one();
  else if (i <= ni) v = fq_load(inputs + (i - 1));
  fq_store(z + nv + i, v);
}

// ---- triplets -> CSR / CSC on the device -------------------------------------------------------------------------------------
// cnt_r[row] / cnt_c[col] += 1 (the callers pass rowptr + 1 / colptr + 1, so the inclusive scan leaves the pointers in place);
// *bad = 1 on an index out of range
// atomicAdd(base + key, 1) for every active lane, returning the value before the lane's own increment.  An R1CS has columns that
// carry a large share of a matrix's entries (the constant 1: 37 % of B in vPIN's point-mult gadget): a wave's lanes then hit ONE
// counter, and 64 serialised atomics per wave made this pass cost more than the proof's sum-checks.  The lanes that share the
// first active lane's key are served by a single atomic (count = their number, rank = position among them); the others take
// their own.
__device__ __forceinline__ uint32_t wave_counted_inc(uint32_t* __restrict__ base, uint32_t key, bool active) {
  const int lane = (int)(threadIdx.x & 63);
  uint32_t res = 0u;
  bool todo = active;
  // peel off up to four groups of lanes with equal keys (the hot key is rarely the FIRST lane's, but it is the largest group, so
  // it goes within a few rounds), then one atomic per remaining lane
#pragma unroll 1
  for (int round = 0; round < 4; round++) {
    const unsigned long long act = __ballot(todo);
    if (!act) return res;
    const int leader = __ffsll((long long)act) - 1;
    const uint32_t k0 = (uint32_t)__shfl((int)key, leader, 64);
    const bool same = todo && key == k0;
    const unsigned long long m = __ballot(same);
    uint32_t first = 0u;
    if (lane == leader) first = atomicAdd(base + k0, (uint32_t)__popcll(m));
    first = (uint32_t)__shfl((int)first, leader, 64);
    if (same) { res = first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); todo = false; }
  }
  if (todo) res = atomicAdd(base + key, 1u);
  return res;
}

__global__ __launch_bounds__(kRB) void triplet_hist_kernel(const uint32_t* __restrict__ row, const uint32_t* __restrict__ col, size_t nnz,
                                                           uint32_t nrows, uint32_t ncols, uint32_t* __restrict__ cnt_r,
                                                           uint32_t* __restrict__ cnt_c, uint32_t* __restrict__ bad) {
  // (whole waves step through the loop together: the wave-level helper wants every lane of a wave to call it)
  const size_t stride = (size_t)gridDim.x * kRB, rounds = (nnz + stride - 1) / stride;
  for (size_t it = 0; it < rounds; it++) {
    const size_t k = it * stride + (size_t)blockIdx.x * kRB + threadIdx.x;
    const bool in = k < nnz;
    const uint32_t r = in ? row[k] : 0u, cc = in ? col[k] : 0u;
    const bool ok = in && r < nrows && cc < ncols;
    if (in && !ok) *bad = 1u;
    (void)wave_counted_inc(cnt_r, r, ok);
    (void)wave_counted_inc(cnt_c, cc, ok);
  }
}

__global__ __launch_bounds__(kRB) void triplet_scatter_kernel(const uint32_t* __restrict__ row, const uint32_t* __restrict__ col,
                                                              const fq* __restrict__ val, size_t nnz, uint32_t nrows, uint32_t ncols,
                                                              const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ colptr,
                                                              uint32_t* __restrict__ rcur, uint32_t* __restrict__ ccur,
                                                              uint32_t* __restrict__ csr_col, fq* __restrict__ csr_val,
                                                              uint32_t* __restrict__ csc_row, fq* __restrict__ csc_val) {
  const size_t stride = (size_t)gridDim.x * kRB, rounds = (nnz + stride - 1) / stride;
  for (size_t it = 0; it < rounds; it++) {
    const size_t k = it * stride + (size_t)blockIdx.x * kRB + threadIdx.x;
    const bool in = k < nnz;
    const uint32_t r = in ? row[k] : 0u, cc = in ? col[k] : 0u;
    const bool ok = in && r < nrows && cc < ncols;  // (a bad index was reported by the histogram pass)
    const uint32_t ir = wave_counted_inc(rcur, r, ok), ic = wave_counted_inc(ccur, cc, ok);
    if (!ok) continue;
    const fq v = fq_load(val + k);
    const uint32_t pr = rowptr[r] + ir, pc = colptr[cc] + ic;
    csr_col[pr] = cc; fq_store(csr_val + pr, v);
    csc_row[pc] = r; fq_store(csc_val + pc, v);
  }
}

// (column, first entry, end) of every column with more than kLongCol entries -> out[3 * (*count)++]; entries beyond cap are
// counted but not written (the caller then walks colptr itself)
__global__ __launch_bounds__(kRB) void long_cols_kernel(const uint32_t* __restrict__ colptr, size_t ncols, uint32_t* __restrict__ out,
                                                        uint32_t* __restrict__ count, uint32_t cap) {
  for (size_t i = (size_t)blockIdx.x * kRB + threadIdx.x; i < ncols; i += (size_t)gridDim.x * kRB) {
    const uint32_t a = colptr[i], b = colptr[i + 1];
    if (b - a > kLongCol) {
      const uint32_t at = atomicAdd(count, 1u);
      if (at < cap) { out[3 * at] = (uint32_t)i; out[3 * at + 1] = a; out[3 * at + 2] = b; }
    }
  }
}

// in-place inclusive prefix sum of n u32 (three launches: per-block scans of kScanElems, a one-workgroup scan of the block
// totals, the offsets added back)
constexpr int kScanThreads = 256, kScanPer = 8, kScanElems = kScanThreads * kScanPer;
__global__ __launch_bounds__(kScanThreads) void scan_blocks_kernel(uint32_t* __restrict__ a, size_t n, uint32_t* __restrict__ sums) {
  __shared__ uint32_t sh[kScanThreads];
  const size_t base = (size_t)blockIdx.x * kScanElems + (size_t)threadIdx.x * kScanPer;
  uint32_t v[kScanPer], run = 0;
#pragma unroll
  for (int i = 0; i < kScanPer; i++) { v[i] = base + i < n ? a[base + i] : 0u; run += v[i]; v[i] = run; }
  sh[threadIdx.x] = run;
  __syncthreads();
  for (int st = 1; st < kScanThreads; st <<= 1) {  // Hillis-Steele over the per-thread totals
    const uint32_t add = (int)threadIdx.x >= st ? sh[threadIdx.x - st] : 0u;
    __syncthreads();
    sh[threadIdx.x] += add;
    __syncthreads();
  }
  const uint32_t before = threadIdx.x ? sh[threadIdx.x - 1] : 0u;
#pragma unroll
  for (int i = 0; i < kScanPer; i++)
    if (base + i < n) a[base + i] = v[i] + before;
  if (threadIdx.x == kScanThreads - 1) sums[blockIdx.x] = sh[kScanThreads - 1];
}
__global__ __launch_bounds__(1024) void scan_sums_kernel(uint32_t* __restrict__ sums, size_t nb) {
  __shared__ uint32_t sh[1024];
  const size_t per = (nb + 1023) / 1024, b0 = (size_t)threadIdx.x * per, b1 = b0 + per < nb ? b0 + per : nb;
  uint32_t run = 0;
  for (size_t b = b0; b < b1; b++) run += sums[b];
  sh[threadIdx.x] = run;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int t = 0; t < 1024; t++) { const uint32_t x = sh[t]; sh[t] = acc; acc += x; }
  }
  __syncthreads();
  uint32_t acc = sh[threadIdx.x];
  for (size_t b = b0; b < b1; b++) { const uint32_t x = sums[b]; sums[b] = acc; acc += x; }  // exclusive block offsets
}
__global__ __launch_bounds__(kScanThreads) void scan_add_kernel(uint32_t* __restrict__ a, size_t n, const uint32_t* __restrict__ sums) {
  const uint32_t off = sums[blockIdx.x];
  const size_t base = (size_t)blockIdx.x * kScanElems + (size_t)threadIdx.x * kScanPer;
#pragma unroll
  for (int i = 0; i < kScanPer; i++)
    if (base + i < n) a[base + i] += off;
}
static size_t scan_scratch_bytes(size_t n) { return ((n + kScanElems - 1) / kScanElems + 1) * sizeof(uint32_t); }
static int scan_inclusive_u32(vpin_ctx* c, uint32_t* a, size_t n, uint32_t* sums) {
  if (n == 0) return VPIN_OK;
  const size_t nb = (n + kScanElems - 1) / kScanElems;
  hipLaunchKernelGGL(scan_blocks_kernel, dim3((unsigned)nb), dim3(kScanThreads), 0, c->stream, a, n, sums);
  if (nb > 1) {
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, c->stream, sums, nb);
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(kScanThreads), 0, c->stream, a, n, (const uint32_t*)sums);
  }
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}

template <typename T>
static int up(vpin_ctx* c, T** dst, const T* src, size_t n) {
  if (hipMalloc((void**)dst, (n ? n : 1) * sizeof(T)) != hipSuccess) return VPIN_ENOMEM;
  if (n) VPIN_HIP_TRY(hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
  return VPIN_OK;
}

}  // namespace vpin

using namespace vpin;

extern "C" {

void vpin_r1cs_dims(const vpin_r1cs_dev* d, size_t* num_cons, size_t* num_vars, size_t* num_inputs) {
  if (num_cons) *num_cons = d ? d->num_cons : 0;
  if (num_vars) *num_vars = d ? d->num_vars : 0;
  if (num_inputs) *num_inputs = d ? d->num_inputs : 0;
}

void vpin_r1cs_free(vpin_ctx* c, vpin_r1cs_dev* d) {
  if (!d) return;
  if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
  for (int m = 0; m < 3; m++) {
    void* ps[] = {d->rowptr[m], d->csr_col[m], d->csr_val[m], d->colptr[m], d->csc_row[m], d->csc_val[m], d->long_cols[m],
                  d->long_first[m], d->chunk_k0[m], d->chunk_k1[m]};
    for (void* p : ps)
      if (p) { if (d->pooled) dev_free_owned(d->owner, c, p); else (void)hipFree(p); }
  }
  delete d;
}

// ---- CSR / CSC from the host's triplets, built on the device (round 5: the host used to counting-sort every matrix on one
// thread: 80 ms of an 83 ms sat proof from host buffers for CNN A) ----------------------------------------------------------
// histogram -> exclusive scan -> scatter with one atomic cursor per row / column.  The order of a row's (column's) entries is
// whatever the atomics give: the kernels above add field elements, so any order yields the same bits.

int vpin_r1cs_upload(vpin_ctx* c, const vpin_r1cs* inst, vpin_r1cs_dev** out) {
  if (!c || !inst || !out) return VPIN_EINVAL;
  if (!is_pow2(inst->num_cons) || !is_pow2(inst->num_vars) || inst->num_inputs >= inst->num_vars) return VPIN_ESHAPE;
  const size_t nrows = inst->num_cons, ncols = 2 * inst->num_vars;
  if (nrows >= ((size_t)1 << 32) || ncols >= ((size_t)1 << 32)) return VPIN_ESHAPE;
  for (int m = 0; m < 3; m++)
    if (inst->nnz[m] >= ((size_t)1 << 32) || (inst->nnz[m] && (!inst->row[m] || !inst->col[m] || !inst->val[m]))) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  vpin_r1cs_dev* d = new (std::nothrow) vpin_r1cs_dev();
  if (!d) return VPIN_ENOMEM;
  d->num_cons = inst->num_cons; d->num_vars = inst->num_vars; d->num_inputs = inst->num_inputs;
  d->pooled = true;
  d->owner = c;
  int rc = VPIN_OK;
  TraceLap lap(c, "vpin_r1cs_upload");  // VPIN_CLI_TRACE=1
  auto alloc = [&](auto** p, size_t n) { return dev_alloc(c, (n ? n : 1) * sizeof(**p), (void**)p); };
  DevBuf b_err(c), b_long(c);
  constexpr uint32_t kLongCap = 1u << 16;  // long columns found on the device; more than this: the host walks colptr
  if (b_err.alloc(2 * sizeof(uint32_t)) || b_long.alloc((size_t)kLongCap * 3 * sizeof(uint32_t))) { vpin_r1cs_free(c, d); return VPIN_ENOMEM; }
  uint32_t* d_flags = (uint32_t*)b_err.p;  // [0] bad index seen, [1] long columns found
  for (int m = 0; m < 3 && rc == VPIN_OK; m++) {
    const size_t nnz = inst->nnz[m];
    d->nnz[m] = nnz;
    DevBuf b_row(c), b_col(c), b_val(c), b_rcur(c), b_ccur(c), b_sums(c);
    if ((rc = alloc(&d->rowptr[m], nrows + 1)) || (rc = alloc(&d->colptr[m], ncols + 1)) || (rc = alloc(&d->csr_col[m], nnz)) ||
        (rc = alloc(&d->csr_val[m], nnz)) || (rc = alloc(&d->csc_row[m], nnz)) || (rc = alloc(&d->csc_val[m], nnz)))
      break;
    if (b_row.alloc((nnz ? nnz : 1) * 4) || b_col.alloc((nnz ? nnz : 1) * 4) || b_val.alloc((nnz ? nnz : 1) * 32) || b_rcur.alloc(nrows * 4) ||
        b_ccur.alloc(ncols * 4) || b_sums.alloc(scan_scratch_bytes(std::max(nrows, ncols) + 1))) { rc = VPIN_ENOMEM; break; }
    hipError_t e = hipSuccess;
    if (nnz) {
      e = hipMemcpyAsync(b_row.p, inst->row[m], nnz * 4, hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(b_col.p, inst->col[m], nnz * 4, hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(b_val.p, inst->val[m], nnz * 32, hipMemcpyHostToDevice, c->stream);
    }
    if (e == hipSuccess) e = hipMemsetAsync(d->rowptr[m], 0, (nrows + 1) * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d->colptr[m], 0, (ncols + 1) * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b_rcur.p, 0, nrows * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b_ccur.p, 0, ncols * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, 2 * sizeof(uint32_t), c->stream);
    if (e != hipSuccess) { set_last_error("vpin_r1cs_upload", e); rc = VPIN_EHIP; break; }
    lap("alloc + H2D + memsets");
    const unsigned gb = (unsigned)std::min<size_t>(4096, (nnz + kRB - 1) / kRB + 1);
    hipLaunchKernelGGL(triplet_hist_kernel, dim3(gb), dim3(kRB), 0, c->stream, (const uint32_t*)b_row.p, (const uint32_t*)b_col.p, nnz,
                       (uint32_t)nrows, (uint32_t)ncols, d->rowptr[m] + 1, d->colptr[m] + 1, d_flags);
    if ((rc = scan_inclusive_u32(c, d->rowptr[m] + 1, nrows, (uint32_t*)b_sums.p)) ||
        (rc = scan_inclusive_u32(c, d->colptr[m] + 1, ncols, (uint32_t*)b_sums.p)))
      break;
    hipLaunchKernelGGL(triplet_scatter_kernel, dim3(gb), dim3(kRB), 0, c->stream, (const uint32_t*)b_row.p, (const uint32_t*)b_col.p,
                       (const fq*)b_val.p, nnz, (uint32_t)nrows, (uint32_t)ncols, (const uint32_t*)d->rowptr[m], (const uint32_t*)d->colptr[m],
                       (uint32_t*)b_rcur.p, (uint32_t*)b_ccur.p, d->csr_col[m], d->csr_val[m], d->csc_row[m], d->csc_val[m]);
    hipLaunchKernelGGL(long_cols_kernel, dim3((unsigned)std::min<size_t>(4096, (ncols + kRB - 1) / kRB)), dim3(kRB), 0, c->stream,
                       (const uint32_t*)d->colptr[m], ncols, (uint32_t*)b_long.p, d_flags + 1, kLongCap);
    uint32_t flags[2] = {0, 0};
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { set_last_error("vpin_r1cs_upload", e); rc = VPIN_EHIP; break; }
    lap("hist + scan + scatter + long cols");
    if (flags[0]) { rc = VPIN_ESHAPE; break; }  // lib.rs:171-178 InvalidIndex
    // the few long columns (the constant 1, the inputs): (column, first entry, end) triples, sorted by column on the host
    std::vector<uint32_t> longs, lfirst, ck0, ck1;
    std::vector<uint32_t> trip;
    if (flags[1] <= kLongCap) {
      trip.resize((size_t)flags[1] * 3);
      if (flags[1]) {
        // (on the context's stream, never the legacy NULL stream: CU-masked streams are blocking streams and would wait for it)
        e = hipMemcpyAsync(trip.data(), b_long.p, trip.size() * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { set_last_error("vpin_r1cs_upload", e); rc = VPIN_EHIP; break; }
      }
    } else {  // more long columns than the device list holds: walk the whole colptr on the host
      std::vector<uint32_t> colptr(ncols + 1);
      e = hipMemcpyAsync(colptr.data(), d->colptr[m], (ncols + 1) * 4, hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) { set_last_error("vpin_r1cs_upload", e); rc = VPIN_EHIP; break; }
      for (size_t i = 0; i < ncols; i++)
        if (colptr[i + 1] - colptr[i] > kLongCol) { trip.push_back((uint32_t)i); trip.push_back(colptr[i]); trip.push_back(colptr[i + 1]); }
    }
    std::vector<size_t> order(trip.size() / 3);
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return trip[3 * a] < trip[3 * b]; });
    for (size_t oi : order) {
      longs.push_back(trip[3 * oi]);
      lfirst.push_back((uint32_t)ck0.size());
      for (uint32_t k = trip[3 * oi + 1]; k < trip[3 * oi + 2]; k += kChunk) {
        ck0.push_back(k);
        ck1.push_back(std::min<uint32_t>(k + kChunk, trip[3 * oi + 2]));
      }
    }
    lfirst.push_back((uint32_t)ck0.size());
    d->n_long[m] = longs.size();
    d->n_chunks[m] = ck0.size();
    auto upv = [&](uint32_t** dst, const std::vector<uint32_t>& v) {
      int r2 = alloc(dst, v.size());
      if (!r2 && !v.empty() && hipMemcpyAsync(*dst, v.data(), v.size() * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) r2 = VPIN_EHIP;
      return r2;
    };
    if ((rc = upv(&d->long_cols[m], longs)) || (rc = upv(&d->long_first[m], lfirst)) || (rc = upv(&d->chunk_k0[m], ck0)) ||
        (rc = upv(&d->chunk_k1[m], ck1)))
      break;
    if (hipStreamSynchronize(c->stream) != hipSuccess) rc = VPIN_EHIP;  // host vectors and the DevBufs die at scope end
    lap("long-column lists");
  }
  if (rc) { vpin_r1cs_free(c, d); return rc; }
  *out = d;
  return VPIN_OK;
}

// z table from the assignment: z = [vars | 1 | inputs | 0..] (commit_test.rs:162-170)
int vpin_r1cs_build_z(vpin_ctx* c, const vpin_r1cs_dev* d, const vpin_table* vars, const uint8_t* inputs, vpin_table** out_z) {
  if (!c || !d || !vars || !out_z || (d->num_inputs && !inputs)) return VPIN_EINVAL;
  if (vars->len != d->num_vars) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  vpin_table* z = nullptr;
  int rc = table_alloc_uninit(c, 2 * d->num_vars, &z);  // lower half copied, upper half written by the kernel
  if (rc) return rc;
  DevBuf bin(c);
  fq* d_in = nullptr;
  if (d->num_inputs) {
    if (bin.alloc(d->num_inputs * 32)) { vpin_table_free(c, z); return VPIN_ENOMEM; }
    d_in = (fq*)bin.p;
    (void)hipMemcpyAsync(d_in, inputs, d->num_inputs * 32, hipMemcpyHostToDevice, c->stream);
  }
  (void)hipMemcpyAsync(z->d, vars->d, d->num_vars * 32, hipMemcpyDeviceToDevice, c->stream);
  hipLaunchKernelGGL(build_z_hi_kernel, dim3((unsigned)((d->num_vars + kRB - 1) / kRB)), dim3(kRB), 0, c->stream, z->d,
                     d->num_vars, d_in, d->num_inputs);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // `inputs` is a caller buffer
  if (e != hipSuccess) { set_last_error("vpin_r1cs_build_z", e); vpin_table_free(c, z); return VPIN_EHIP; }
  *out_z = z;
  return VPIN_OK;
}

// R1CSInstance::multiply_vec (r1csinstance.rs:272-285): three fresh tables of num_cons entries
int vpin_r1cs_multiply_vec(vpin_ctx* c, const vpin_r1cs_dev* d, const vpin_table* z, vpin_table** Az, vpin_table** Bz,
                           vpin_table** Cz) {
  if (!c || !d || !z || !Az || !Bz || !Cz) return VPIN_EINVAL;
  if (z->len != 2 * d->num_vars) return VPIN_ESHAPE;  // assert_eq!(z.len(), num_cols)
  (void)hipSetDevice(c->device);
  vpin_table* t[3] = {nullptr, nullptr, nullptr};
  for (int m = 0; m < 3; m++) {
    int rc = table_alloc_uninit(c, d->num_cons, &t[m]);  // spmv_kernel writes every row
    if (rc) { for (int k = 0; k < m; k++) vpin_table_free(c, t[k]); return rc; }
    hipLaunchKernelGGL(spmv_kernel, dim3((unsigned)((d->num_cons + kRB - 1) / kRB)), dim3(kRB), 0, c->stream, d->rowptr[m],
                       d->csr_col[m], d->csr_val[m], z->d, d->num_cons, t[m]->d);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error("spmv", e); for (auto* p : t) vpin_table_free(c, p); return VPIN_EHIP; }
  *Az = t[0]; *Bz = t[1]; *Cz = t[2];
  return VPIN_OK;
}

// r_A*A(rx,.) + r_B*B(rx,.) + r_C*C(rx,.) over all 2*num_vars columns
// (compute_eval_table_sparse x3 + the combination at commit_test.rs:257-268)
int vpin_r1cs_eval_table(vpin_ctx* c, const vpin_r1cs_dev* d, const vpin_table* evals_rx, const uint8_t r_abc[96],
                         vpin_table** out) {
  if (!c || !d || !evals_rx || !r_abc || !out) return VPIN_EINVAL;
  if (evals_rx->len != d->num_cons) return VPIN_ESHAPE;  // assert_eq!(rx.len(), num_rows)
  (void)hipSetDevice(c->device);
  const size_t ncols = 2 * d->num_vars;
  vpin_table* t = nullptr;
  int rc = table_alloc_uninit(c, ncols, &t);  // matrix A's pass writes every column, B and C accumulate
  if (rc) return rc;
  // long-column chunk partials: scratch of THIS call from the calling context's pool (the instance is shared,
  // read-only, by every context / stream that proves it)
  DevBuf chunk_scratch(c);
  const size_t max_chunks = std::max(d->n_chunks[0], std::max(d->n_chunks[1], d->n_chunks[2]));
  if (max_chunks && chunk_scratch.alloc(max_chunks * sizeof(fq))) { vpin_table_free(c, t); return VPIN_ENOMEM; }
  fq* chunk_partials = (fq*)chunk_scratch.p;
  for (int m = 0; m < 3; m++) {
    fq rc_m;
    memcpy(rc_m.v, r_abc + 32 * m, 32);
    const int accum = m > 0 ? 1 : 0;
    hipLaunchKernelGGL(eval_table_kernel, dim3((unsigned)((ncols + kRB - 1) / kRB)), dim3(kRB), 0, c->stream, d->colptr[m],
                       d->csc_row[m], d->csc_val[m], evals_rx->d, ncols, rc_m, accum, t->d);
    if (d->n_long[m]) {
      hipLaunchKernelGGL(long_chunk_kernel, dim3((unsigned)d->n_chunks[m]), dim3(kRB), 0, c->stream, d->chunk_k0[m],
                         d->chunk_k1[m], d->csc_row[m], d->csc_val[m], evals_rx->d, chunk_partials);
      hipLaunchKernelGGL(eval_table_long_finish_kernel, dim3((unsigned)d->n_long[m]), dim3(64), 0, c->stream,
                         d->long_cols[m], d->long_first[m], d->n_long[m], chunk_partials, rc_m, accum, t->d);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error("eval_table", e); vpin_table_free(c, t); return VPIN_EHIP; }
  *out = t;
  return VPIN_OK;
}

}  // extern "C"

namespace vpin {

// multiply_vec for the rows = r0 (mod step): three tables of num_cons / step entries
int r1cs_multiply_vec_strided(vpin_ctx* c, const vpin_r1cs_dev* d, const vpin_table* z, size_t r0, size_t step, vpin_table** Az,
                              vpin_table** Bz, vpin_table** Cz) {
  if (!c || !d || !z || !Az || !Bz || !Cz || step == 0 || r0 >= step || d->num_cons % step) return VPIN_EINVAL;
  if (z->len != 2 * d->num_vars) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  const size_t nloc = d->num_cons / step;
  vpin_table* t[3] = {nullptr, nullptr, nullptr};
  for (int m = 0; m < 3; m++) {
    int rc = table_alloc_uninit(c, nloc, &t[m]);
    if (rc) { for (int k = 0; k < m; k++) vpin_table_free(c, t[k]); return rc; }
    hipLaunchKernelGGL(spmv_strided_kernel, dim3((unsigned)((nloc + kRB - 1) / kRB)), dim3(kRB), 0, c->stream, d->rowptr[m],
                       d->csr_col[m], d->csr_val[m], z->d, nloc, r0, step, t[m]->d);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error("spmv_strided", e); for (auto* p : t) vpin_table_free(c, p); return VPIN_EHIP; }
  *Az = t[0]; *Bz = t[1]; *Cz = t[2];
  return VPIN_OK;
}

// out[k] = src[r0 + k*step], a fresh table of src->len / step entries
int table_take_strided(vpin_ctx* c, const vpin_table* src, size_t r0, size_t step, vpin_table** out) {
  if (!c || !src || !out || step == 0 || r0 >= step || src->len % step) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  const size_t nloc = src->len / step;
  vpin_table* t = nullptr;
  int rc = table_alloc_uninit(c, nloc, &t);
  if (rc) return rc;
  hipLaunchKernelGGL(strided_take_kernel, dim3((unsigned)((nloc + kRB - 1) / kRB)), dim3(kRB), 0, c->stream, (const fq*)src->d, nloc, r0,
                     step, t->d);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error("strided_take", e); vpin_table_free(c, t); return VPIN_EHIP; }
  *out = t;
  return VPIN_OK;
}

// the eval table of vpin_r1cs_eval_table at the columns = r0 (mod step): 2*num_vars / step entries
int r1cs_eval_table_strided(vpin_ctx* c, const vpin_r1cs_dev* d, const vpin_table* evals_rx, const uint8_t r_abc[96], size_t r0,
                            size_t step, vpin_table** out) {
  if (!c || !d || !evals_rx || !r_abc || !out || step == 0 || r0 >= step || (2 * d->num_vars) % step) return VPIN_EINVAL;
  if (evals_rx->len != d->num_cons) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  const size_t nloc = 2 * d->num_vars / step;
  vpin_table* t = nullptr;
  int rc = table_alloc_uninit(c, nloc, &t);
  if (rc) return rc;
  DevBuf chunk_scratch(c);
  const size_t max_chunks = std::max(d->n_chunks[0], std::max(d->n_chunks[1], d->n_chunks[2]));
  if (max_chunks && chunk_scratch.alloc(max_chunks * sizeof(fq))) { vpin_table_free(c, t); return VPIN_ENOMEM; }
  fq* chunk_partials = (fq*)chunk_scratch.p;
  for (int m = 0; m < 3; m++) {
    fq rc_m;
    memcpy(rc_m.v, r_abc + 32 * m, 32);
    hipLaunchKernelGGL(eval_table_strided_kernel, dim3((unsigned)((nloc + kRB - 1) / kRB)), dim3(kRB), 0, c->stream, d->colptr[m],
                       d->csc_row[m], d->csc_val[m], evals_rx->d, nloc, r0, step, rc_m, m > 0 ? 1 : 0, t->d);
    if (d->n_long[m]) {
      hipLaunchKernelGGL(long_chunk_kernel, dim3((unsigned)d->n_chunks[m]), dim3(kRB), 0, c->stream, d->chunk_k0[m],
                         d->chunk_k1[m], d->csc_row[m], d->csc_val[m], evals_rx->d, chunk_partials);
      hipLaunchKernelGGL(eval_table_long_finish_strided_kernel, dim3((unsigned)d->n_long[m]), dim3(64), 0, c->stream,
                         d->long_cols[m], d->long_first[m], d->n_long[m], chunk_partials, rc_m, r0, step, t->d);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error("eval_table_strided", e); vpin_table_free(c, t); return VPIN_EHIP; }
  *out = t;
  return VPIN_OK;
}

}  // namespace vpin

extern "C" {

// R1CSInstance::evaluate (r1csinstance.rs:297-302) given the two eq tables: out = Ar|Br|Cr
int vpin_r1cs_evaluate(vpin_ctx* c, const vpin_r1cs_dev* d, const vpin_table* evals_rx, const vpin_table* evals_ry,
                       uint8_t out[96]) {
  if (!c || !d || !evals_rx || !evals_ry || !out) return VPIN_EINVAL;
  if (evals_rx->len != d->num_cons || evals_ry->len != 2 * d->num_vars) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  const size_t ncols = 2 * d->num_vars;
  DevBuf bo(c), chunk_scratch(c);
  if (bo.alloc(96)) return VPIN_ENOMEM;
  fq* d_out = (fq*)bo.p;
  const size_t max_chunks = std::max(d->n_chunks[0], std::max(d->n_chunks[1], d->n_chunks[2]));
  if (max_chunks && chunk_scratch.alloc(max_chunks * sizeof(fq))) return VPIN_ENOMEM;
  fq* chunk_partials = (fq*)chunk_scratch.p;
  int grid = (int)std::min<size_t>((ncols + kRB - 1) / kRB, 1024);
  for (int m = 0; m < 3; m++) {
    int nparts = grid + (int)d->n_long[m];
    if ((size_t)nparts > c->partials_cap) return VPIN_ESHAPE;
    hipLaunchKernelGGL(evaluate_kernel, dim3(grid), dim3(kRB), 0, c->stream, d->colptr[m], d->csc_row[m], d->csc_val[m],
                       evals_rx->d, evals_ry->d, ncols, c->d_partials);
    if (d->n_long[m]) {
      hipLaunchKernelGGL(long_chunk_kernel, dim3((unsigned)d->n_chunks[m]), dim3(kRB), 0, c->stream, d->chunk_k0[m],
                         d->chunk_k1[m], d->csc_row[m], d->csc_val[m], evals_rx->d, chunk_partials);
      hipLaunchKernelGGL(evaluate_long_finish_kernel, dim3((unsigned)d->n_long[m]), dim3(64), 0, c->stream,
                         d->long_cols[m], d->long_first[m], d->n_long[m], chunk_partials, evals_ry->d, c->d_partials + grid);
    }
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kRB), 0, c->stream, c->d_partials, nparts, d_out + m);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, 96, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) { set_last_error("vpin_r1cs_evaluate", e); return VPIN_EHIP; }
  return VPIN_OK;
}

}  // extern "C"
