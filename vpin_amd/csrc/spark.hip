// spark.hip -- device kernels of the SPARK half (computation commitment + sparse-polynomial
// evaluation proof) for gfx950.
//
// Replaces the data-parallel loops of
//   Spartan/src/sparse_mlpoly.rs:267-283,525-531  AddrTimestamps::deref (gather)
//   Spartan/src/sparse_mlpoly.rs:547-622          Layers::build_hash_layer
//   Spartan/src/product_tree.rs:18-56             ProductCircuit::new (pairwise-product tree)
//   Spartan/src/sumcheck.rs:273-330,346-370       prove_cubic_batched round evaluation + folds
//   Spartan/src/dense_mlpoly.rs:249-255           DensePolynomial::evaluate of the dense slices
//
// Layout.  A product circuit over n leaves is ONE array of 2n field elements: level l (n>>l entries)
// at offset 2n - (2n>>l); the reference's left_vec[l] / right_vec[l] are its two halves, and level
// l+1 is the element-wise product of those halves, so building a level is a unit-stride stream
// (2 loads, 1 store, 1 Montgomery product per output).  The 12 "ops" circuits (and the 4 "mem"
// circuits) of a proof have equal size and are proven in lock-step, so they sit in one allocation
// and every kernel takes the circuit index from blockIdx.y: one launch per round for all of them.
//
// Round kernel.  The third factor of every product-circuit sum-check is eq(rand, .), so the rounds
// run eq-factored exactly like phase 1 of the sat proof (sumcheck.hip): the kernel returns
// sum_i E[i]*(A_x*B_x)[i] for x = 0,2,3 with E the read-only suffix table of the round, and the
// host applies the per-round scalar.  Per pair: fold 2 tables (4 products), 3 products A_x*B_x,
// 3 products by E: 10 Montgomery products, 9 loads, 4 stores.  HBM-streaming integer work.
#include <atomic>
#include <chrono>
#include <cstdlib>

#include "mailbox_dev.h"
#include "sc_dev.h"
#include "spark_dev.h"

namespace vpin {

// R^2 mod q (ristretto255.rs:309-314): fq_mul(raw, kR2) = raw in Montgomery form
__device__ __forceinline__ fq fq_r2() {
  fq r;
  r.v[0] = 0x449c0f01u; r.v[1] = 0xa40611e3u; r.v[2] = 0x68859347u; r.v[3] = 0xd00e1ba7u;
  r.v[4] = 0x17f5be65u; r.v[5] = 0xceec73d2u; r.v[6] = 0x7c309a3du; r.v[7] = 0x0399411bu;
  return r;
}
__device__ __forceinline__ fq fq_raw_u32(uint32_t x) {
  fq r = fq_zero();
  r.v[0] = x;
  return r;
}

__global__ __launch_bounds__(kBlock) void u32_to_fq_kernel(const uint32_t* __restrict__ src, fq* __restrict__ dst, size_t n) {
  const fq r2 = fq_r2();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    uint32_t x = src[i];
    fq_store(dst + i, x ? fq_mul(fq_raw_u32(x), r2) : fq_zero());
  }
}

// blockIdx.y = slice 0..7: 0..2 row derefs of A,B,C, 3..5 col derefs, 6..7 zero padding
__global__ __launch_bounds__(kBlock) void gather_derefs_kernel(const uint32_t* __restrict__ idx, size_t N,
                                                               const fq* __restrict__ mem_rx, const fq* __restrict__ mem_ry,
                                                               fq* __restrict__ comb) {
  const int s = blockIdx.y;
  fq* dst = comb + (size_t)s * N;
  if (s >= 6) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (size_t)gridDim.x * kBlock) fq_store(dst + i, fq_zero());
    return;
  }
  const uint32_t* a = idx + (size_t)(s < 3 ? s : 3 + s) * N;  // idx slices 0..2 = row, 6..8 = col
  const fq* mem = s < 3 ? mem_rx : mem_ry;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (size_t)gridDim.x * kBlock)
    fq_store(dst + i, fq_load(mem + a[i]));
}

// hash_func(addr, val, ts) - r_multiset_check = ts*r^2 + val*r + addr - gamma  (sparse_mlpoly.rs:557-560).
// r2_boost = r^2 * R (Montgomery form of the Montgomery image), so fq_mul(raw ts, r2_boost) is ts*r^2
// in Montgomery form without a separate conversion of ts.
struct HashParams { fq r, r2, r2_boost, gamma; };
struct CircIds { int v[12]; };  // global circuit (or dot-product half) of each local one

// blockIdx.y = side*3 + m.  Writes level 0 of read circuit (side*6 + m) and write circuit (side*6 + 3 + m).
__global__ __launch_bounds__(kBlock) void hash_ops_kernel(const uint32_t* __restrict__ idx, const fq* __restrict__ derefs, size_t N,
                                                          HashParams hp, fq* __restrict__ forest) {
  const int side = blockIdx.y / 3, m = blockIdx.y % 3;
  const uint32_t* addr = idx + (size_t)(side * 6 + m) * N;
  const uint32_t* ts = idx + (size_t)(side * 6 + 3 + m) * N;
  const fq* val = derefs + (size_t)(side * 3 + m) * N;
  fq* rd = forest + (size_t)(side * 6 + m) * 2 * N;
  fq* wr = forest + (size_t)(side * 6 + 3 + m) * 2 * N;
  const fq r2c = fq_r2();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (size_t)gridDim.x * kBlock) {
    fq h = fq_sub(fq_mul(fq_load(val + i), hp.r), hp.gamma);
    uint32_t a = addr[i], t = ts[i];
    if (a) h = fq_add(h, fq_mul(fq_raw_u32(a), r2c));
    if (t) h = fq_add(h, fq_mul(fq_raw_u32(t), hp.r2_boost));
    fq_store(rd + i, h);
    fq_store(wr + i, fq_add(h, hp.r2));  // write timestamp = read timestamp + 1
  }
}

// blockIdx.y = side.  Level 0 of init circuit (2*side) and audit circuit (2*side + 1).
__global__ __launch_bounds__(kBlock) void hash_mem_kernel(const uint32_t* __restrict__ audit_ts, const fq* __restrict__ mem_rx,
                                                          const fq* __restrict__ mem_ry, size_t M, HashParams hp,
                                                          fq* __restrict__ forest) {
  const int side = blockIdx.y;
  const fq* mem = side ? mem_ry : mem_rx;
  const uint32_t* ts = audit_ts + (size_t)side * M;
  fq* init = forest + (size_t)(2 * side) * 2 * M;
  fq* audit = forest + (size_t)(2 * side + 1) * 2 * M;
  const fq r2c = fq_r2();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < M; i += (size_t)gridDim.x * kBlock) {
    fq h = fq_sub(fq_mul(fq_load(mem + i), hp.r), hp.gamma);
    if (i) h = fq_add(h, fq_mul(fq_raw_u32((uint32_t)i), r2c));
    fq_store(init + i, h);
    uint32_t t = ts[i];
    if (t) h = fq_add(h, fq_mul(fq_raw_u32(t), hp.r2_boost));
    fq_store(audit + i, h);
  }
}

// The same leaves for a SUBSET of the circuits (one proof over several GPUs: a rank builds the trees of the circuits it
// owns and nothing else).  ids.v[j] = global circuit of local tree j, in the numbering of the two kernels above:
// ops: side*6 + kind*3 + m (kind 0 = read, 1 = write); mem: side*2 + kind (kind 0 = init, 1 = audit).
__global__ __launch_bounds__(kBlock) void hash_ops_sub_kernel(const uint32_t* __restrict__ idx, const fq* __restrict__ derefs, size_t N,
                                                              HashParams hp, fq* __restrict__ forest, CircIds ids) {
  const int g = ids.v[blockIdx.y], side = g / 6, kind = (g % 6) / 3, m = g % 3;
  const uint32_t* addr = idx + (size_t)(side * 6 + m) * N;
  const uint32_t* ts = idx + (size_t)(side * 6 + 3 + m) * N;
  const fq* val = derefs + (size_t)(side * 3 + m) * N;
  fq* leaf = forest + (size_t)blockIdx.y * 2 * N;
  const fq r2c = fq_r2();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (size_t)gridDim.x * kBlock) {
    fq h = fq_sub(fq_mul(fq_load(val + i), hp.r), hp.gamma);
    uint32_t a = addr[i], t = ts[i];
    if (a) h = fq_add(h, fq_mul(fq_raw_u32(a), r2c));
    if (t) h = fq_add(h, fq_mul(fq_raw_u32(t), hp.r2_boost));
    fq_store(leaf + i, kind ? fq_add(h, hp.r2) : h);
  }
}

__global__ __launch_bounds__(kBlock) void hash_mem_sub_kernel(const uint32_t* __restrict__ audit_ts, const fq* __restrict__ mem_rx,
                                                              const fq* __restrict__ mem_ry, size_t M, HashParams hp,
                                                              fq* __restrict__ forest, CircIds ids) {
  const int g = ids.v[blockIdx.y], side = g / 2, kind = g % 2;
  const fq* mem = side ? mem_ry : mem_rx;
  const uint32_t* ts = audit_ts + (size_t)side * M;
  fq* leaf = forest + (size_t)blockIdx.y * 2 * M;
  const fq r2c = fq_r2();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < M; i += (size_t)gridDim.x * kBlock) {
    fq h = fq_sub(fq_mul(fq_load(mem + i), hp.r), hp.gamma);
    if (i) h = fq_add(h, fq_mul(fq_raw_u32((uint32_t)i), r2c));
    if (kind) {
      uint32_t t = ts[i];
      if (t) h = fq_add(h, fq_mul(fq_raw_u32(t), hp.r2_boost));
    }
    fq_store(leaf + i, h);
  }
}

// ---- one proof over several GPUs, split by RESIDUE CLASS (world = a power of two) -------------------------------------
// Every level of a product tree pairs entry i with entry i + len/2 (product_tree.rs:18-35), and so does every fold of the
// sum-checks over it (dense_mlpoly.rs:229-236): while len/2 is a multiple of the world size both members of a pair lie in
// the same residue class.  Rank r0 therefore owns the leaves i = r0 (mod step) of EVERY circuit; its local arrays (local
// index k <-> global index r0 + k*step) are product trees over n/step leaves in their own right, built and proven by the
// unchanged kernels, and no table entry ever moves.  The kernels below produce those local leaves and the local views of
// the committed polynomials directly from the decommitment.

// comb_loc[s*(N/step) + k] = Derefs slice s (0..2 row A,B,C; 3..5 col A,B,C) at entry r0 + k*step
__global__ __launch_bounds__(kBlock) void gather_derefs_loc_kernel(const uint32_t* __restrict__ idx, size_t N, const fq* __restrict__ mem_rx,
                                                                   const fq* __restrict__ mem_ry, fq* __restrict__ comb_loc, size_t r0,
                                                                   size_t step) {
  const int s = blockIdx.y;
  const size_t nloc = N / step;
  const uint32_t* a = idx + (size_t)(s < 3 ? s : 3 + s) * N;
  const fq* mem = s < 3 ? mem_rx : mem_ry;
  fq* dst = comb_loc + (size_t)s * nloc;
  for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < nloc; k += (size_t)gridDim.x * kBlock)
    fq_store(dst + k, fq_load(mem + a[r0 + k * step]));
}

// rows_out[jl*R + c] = entry (r0 + jl*step)*R + c of the derefs polynomial (8 slices of N, the last two zero): the rows
// r0, r0 + step, .. of its commitment matrix, stored densely
__global__ __launch_bounds__(kBlock) void gather_derefs_rows_kernel(const uint32_t* __restrict__ idx, size_t N, size_t R,
                                                                    const fq* __restrict__ mem_rx, const fq* __restrict__ mem_ry,
                                                                    fq* __restrict__ rows_out, size_t r0, size_t step, size_t nrows_loc) {
  const size_t total = nrows_loc * R;
  for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
    const size_t jl = t / R, cidx = t - jl * R;
    const size_t g = (r0 + jl * step) * R + cidx;
    const size_t sl = g / N, i = g - sl * N;
    fq v = fq_zero();
    if (sl < 3) v = fq_load(mem_rx + idx[sl * N + i]);
    else if (sl < 6) v = fq_load(mem_ry + idx[(3 + sl) * N + i]);
    fq_store(rows_out + t, v);
  }
}

// local leaves of the 12 ops circuits: blockIdx.y = side*3 + m writes the read circuit (side*6 + m) and the write circuit
// (side*6 + 3 + m) of the LOCAL forest (n/step leaves each)
__global__ __launch_bounds__(kBlock) void hash_ops_strided_kernel(const uint32_t* __restrict__ idx, const fq* __restrict__ comb_loc, size_t N,
                                                                  HashParams hp, fq* __restrict__ forest, size_t r0, size_t step) {
  const int side = blockIdx.y / 3, m = blockIdx.y % 3;
  const size_t nloc = N / step;
  const uint32_t* addr = idx + (size_t)(side * 6 + m) * N;
  const uint32_t* ts = idx + (size_t)(side * 6 + 3 + m) * N;
  const fq* val = comb_loc + (size_t)(side * 3 + m) * nloc;
  fq* rd = forest + (size_t)(side * 6 + m) * 2 * nloc;
  fq* wr = forest + (size_t)(side * 6 + 3 + m) * 2 * nloc;
  const fq r2c = fq_r2();
  for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < nloc; k += (size_t)gridDim.x * kBlock) {
    const size_t i = r0 + k * step;
    fq h = fq_sub(fq_mul(fq_load(val + k), hp.r), hp.gamma);
    uint32_t a = addr[i], t = ts[i];
    if (a) h = fq_add(h, fq_mul(fq_raw_u32(a), r2c));
    if (t) h = fq_add(h, fq_mul(fq_raw_u32(t), hp.r2_boost));
    fq_store(rd + k, h);
    fq_store(wr + k, fq_add(h, hp.r2));
  }
}

// local leaves of the 4 mem circuits: blockIdx.y = side
__global__ __launch_bounds__(kBlock) void hash_mem_strided_kernel(const uint32_t* __restrict__ audit_ts, const fq* __restrict__ mem_rx,
                                                                  const fq* __restrict__ mem_ry, size_t M, HashParams hp,
                                                                  fq* __restrict__ forest, size_t r0, size_t step) {
  const int side = blockIdx.y;
  const size_t nloc = M / step;
  const fq* mem = side ? mem_ry : mem_rx;
  const uint32_t* ts = audit_ts + (size_t)side * M;
  fq* init = forest + (size_t)(2 * side) * 2 * nloc;
  fq* audit = forest + (size_t)(2 * side + 1) * 2 * nloc;
  const fq r2c = fq_r2();
  for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < nloc; k += (size_t)gridDim.x * kBlock) {
    const size_t i = r0 + k * step;
    fq h = fq_sub(fq_mul(fq_load(mem + i), hp.r), hp.gamma);
    if (i) h = fq_add(h, fq_mul(fq_raw_u32((uint32_t)i), r2c));
    fq_store(init + k, h);
    uint32_t t = ts[i];
    if (t) h = fq_add(h, fq_mul(fq_raw_u32(t), hp.r2_boost));
    fq_store(audit + k, h);
  }
}

// dst[k] = src[r0 + k*step]
__global__ __launch_bounds__(kBlock) void take_strided_kernel(const fq* __restrict__ src, size_t nloc, size_t r0, size_t step,
                                                              fq* __restrict__ dst) {
  for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < nloc; k += (size_t)gridDim.x * kBlock)
    fq_store(dst + k, fq_load(src + r0 + k * step));
}

// next level of every tree: dst[i] = src[i] * src[i + h], i < h  (product_tree.rs:18-35)
__global__ __launch_bounds__(kBlock) void tree_level_kernel(fq* __restrict__ forest, size_t stride, size_t src_off, size_t dst_off,
                                                            size_t h) {
  fq* t = forest + (size_t)blockIdx.y * stride;
  const fq* src = t + src_off;
  fq* dst = t + dst_off;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < h; i += (size_t)gridDim.x * kBlock)
    fq_store(dst + i, fq_mul(fq_load(src + i), fq_load(src + h + i)));
}

// Three levels per pass.  The source level has 2h entries and a node's children sit half a level apart (product_tree.rs:18-35),
// so with q = h/4 the eight entries i + j*q (j = 0..7) give four nodes of the next level (i + j*q, j < 4), two of the one
// after (i, i + q) and one of the third (i): the tree is read once per three levels (2h entries in, 1.75 h out instead of
// 2h + h + h/2 in, 1.75 h out) -- the tree build is bound by HBM, 65 GB per proof of the 2^25 instance level by level.
__global__ __launch_bounds__(kBlock) void tree_level3_kernel(fq* __restrict__ forest, size_t stride, size_t src_off, size_t d1_off,
                                                             size_t d2_off, size_t d3_off, size_t h) {
  fq* t = forest + (size_t)blockIdx.y * stride;
  const fq* src = t + src_off;
  fq *d1 = t + d1_off, *d2 = t + d2_off, *d3 = t + d3_off;
  const size_t q = h / 4;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < q; i += (size_t)gridDim.x * kBlock) {
    const fq a0 = fq_mul(fq_load(src + i), fq_load(src + h + i));
    const fq a1 = fq_mul(fq_load(src + q + i), fq_load(src + h + q + i));
    const fq a2 = fq_mul(fq_load(src + 2 * q + i), fq_load(src + h + 2 * q + i));
    const fq a3 = fq_mul(fq_load(src + 3 * q + i), fq_load(src + h + 3 * q + i));
    fq_store(d1 + i, a0); fq_store(d1 + q + i, a1); fq_store(d1 + 2 * q + i, a2); fq_store(d1 + 3 * q + i, a3);
    const fq b0 = fq_mul(a0, a2), b1 = fq_mul(a1, a3);
    fq_store(d2 + i, b0); fq_store(d2 + q + i, b1);
    fq_store(d3 + i, fq_mul(b0, b1));
  }
}

// the top of every tree in one workgroup: levels from `h0` outputs down to 1 (h0 <= 1024)
constexpr int kTopBlock = 1024;
__global__ __launch_bounds__(kTopBlock) void tree_top_kernel(fq* __restrict__ forest, size_t stride, size_t n2, size_t h0) {
  fq* t = forest + (size_t)blockIdx.x * stride;
  for (size_t h = h0; h >= 1; h >>= 1) {
    // level with h entries is built from the level with 2h entries
    const fq* src = t + (n2 - 4 * h);
    fq* dst = t + (n2 - 2 * h);
    if (threadIdx.x < h) fq_store(dst + threadIdx.x, fq_mul(fq_load(src + threadIdx.x), fq_load(src + h + threadIdx.x)));
    __syncthreads();
  }
}


// ---- fused finisher -----------------------------------------------------------------------
// Every round kernel ends with the "last block done" pattern instead of a second launch: a block
// stores its 3 partial sums, bumps its instance's counter, and the block that finds it was the last
// one sums the instance's partials and writes the three scalars to pinned host memory; the last
// instance to finish (global counter) then publishes the launch group's sequence number in the
// pinned flag word the host is spinning on (no hipStreamSynchronize on the round path: ~7 us
// instead of ~14 us per round on MI355X, tools/ubench_sync.hip).  Counters reset themselves.
struct Finisher {
  fq* partials;        // [inst][nblocks][3]
  uint32_t* counters;  // [kSparkMaxInst] per instance, [kSparkMaxInst] global
  fq* out;             // pinned: out[3*(inst0 + y) + k]
  uint32_t* flag;      // pinned
  uint32_t seq;
  int inst0;           // first instance index of this launch
  int total_inst;      // instances in the launch group (all kernels flagged with the same seq)
  int fused;           // 0: the workgroups only store their partials, round_finish_kernel sums and publishes (large grids)
};

// block-wide sums of e[0..2]: valid in threads 0..2 (thread k holds sum k)
__device__ __forceinline__ fq block_sum3(fq* e) {
  __shared__ fq sh[kBlock / 64][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    fq t = fq_wave_sum(e[k]);
    if (lane == 0) sh[wave][k] = t;
  }
  __syncthreads();
  fq t = fq_zero();
  if (threadIdx.x < 3) {
    t = sh[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kBlock / 64; w++) t = fq_add(t, sh[w][threadIdx.x]);
  }
  __syncthreads();  // sh is reused by a second call
  return t;
}

// called by every thread of the block after the per-thread accumulators e[3] are final
__device__ __forceinline__ void finish_block(fq* e, const Finisher& f) {
  fq t = block_sum3(e);
  if (!f.fused) {
    // Large grids: the "last block done" pattern below costs every workgroup an agent-scope release fence, and on a part
    // with one L2 per XCD that is an L2 write-back per workgroup, serialised per XCD -- measured 0.14 us x workgroups
    // (tools/ubench_rounds.py: 768 workgroups of one pair per thread took 104 us, 384 took 50).  The kernel boundary
    // does the one write-back instead, and round_finish_kernel sums and publishes.
    if (threadIdx.x < 3) fq_store(&f.partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x], t);
    return;
  }
  if (gridDim.x > 1) {
    fq* mine = f.partials + (size_t)blockIdx.y * gridDim.x * 3;
    if (threadIdx.x < 3) fq_store(&mine[(size_t)blockIdx.x * 3 + threadIdx.x], t);
    __shared__ bool is_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) is_last = (atomicAdd(&f.counters[f.inst0 + blockIdx.y], 1u) == gridDim.x - 1);
    __syncthreads();
    if (!is_last) return;
    __threadfence();  // acquire: the other blocks' partials (never read before by this CU in this kernel)
    fq s[3] = {fq_zero(), fq_zero(), fq_zero()};
    for (int b = threadIdx.x; b < (int)gridDim.x; b += kBlock)
#pragma unroll
      for (int k = 0; k < 3; k++) s[k] = fq_add(s[k], fq_load(&mine[(size_t)b * 3 + k]));
    t = block_sum3(s);
  }
  if (threadIdx.x < 3) {
    fq_store(&f.out[3 * (size_t)(f.inst0 + blockIdx.y) + threadIdx.x], t);
    __threadfence_system();
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (gridDim.x > 1) f.counters[f.inst0 + blockIdx.y] = 0;
    if (atomicAdd(&f.counters[kSparkMaxInst], 1u) == (uint32_t)f.total_inst - 1) {
      f.counters[kSparkMaxInst] = 0;
      __threadfence_system();
      *(volatile uint32_t*)f.flag = f.seq;
    }
  }
}

// ---- batched cubic rounds ------------------------------------------------------------------

// unit-stride fold with separate source and destination (the first fold of the dot-product tables
// must not overwrite the committed polynomials)
__device__ __forceinline__ void fold_pd2(const fq* src, fq* dst, size_t i, size_t q, const fq& r, fq& p, fq& d) {
  fq a0 = fq_load(src + i), a1 = fq_load(src + 2 * q + i);
  fq b0 = fq_load(src + q + i), b1 = fq_load(src + 3 * q + i);
  p = fq_add(a0, fq_mul(r, fq_sub(a1, a0)));
  fq hi = fq_add(b0, fq_mul(r, fq_sub(b1, b0)));
  fq_store(dst + i, p);
  fq_store(dst + q + i, hi);
  d = fq_sub(hi, p);
}

// LEAD: return sum E*A_0*B_0 and sum E*dA*dB (sc_dev.h lead_bc) instead of the sums at x = 0, 2, 3; its folds use the
// launch-wide constant form of r (fq_dev.h fq_mul_const, constants in LDS)
// BIG: the same code under its own name for launches of >= 2^20 pairs per circuit (the streaming regime), so profilers
// report that class separately (bench.py roofline.secondary, tools/pmc_summary.py)
template <bool BIND, bool LEAD, bool BIG = false>
__global__ __launch_bounds__(kBlock, kMinWaves) void prod_round_kernel(fq* __restrict__ forest, size_t stride, size_t off, size_t h,
                                                                       const fq* __restrict__ E, size_t pairs, fq r, fq_const rc,
                                                                       Finisher fin) {
  __shared__ __attribute__((aligned(16))) uint32_t tt[8][8];
  if (BIND && LEAD) {
    if (threadIdx.x < 64) tt[threadIdx.x >> 3][threadIdx.x & 7] = rc.tt[threadIdx.x >> 3][threadIdx.x & 7];
    __syncthreads();
  }
  fq* A = forest + (size_t)blockIdx.y * stride + off;
  fq* B = A + h;
  Acc<4> acc;
  acc.init();
  LeadAcc lacc;
  if (LEAD) lacc.init();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < pairs; i += (size_t)gridDim.x * kBlock) {
    fq u[3], p1, d1, p2, d2;
    if (BIND && LEAD) { fold_pd_c(A, i, pairs, tt, p1, d1); fold_pd_c(B, i, pairs, tt, p2, d2); }
    else if (BIND) { fold_pd(A, i, pairs, r, p1, d1); fold_pd(B, i, pairs, r, p2, d2); }
    else { load_pd(A, i, pairs, p1, d1); load_pd(B, i, pairs, p2, d2); }
    if (LEAD) {
      lacc.add(acc.e, fq_load(E + i), fq_mul(p1, p2), fq_mul(d1, d2));  // lead_bc, the two products by E unreduced
    } else {
      acc.stage_bc(u, p1, d1, p2, d2);
      acc.stage_e(u, fq_load(E + i));
    }
  }
  if (LEAD) lacc.flush(acc.e);
  finish_block(acc.e, fin);
}

struct DotpPtrs { const fq* src[3]; fq* dst[3]; size_t src_stride[3]; };

// MODE 0: evaluate src as is; 1: fold src -> dst, evaluate; (dst == src for the in-place rounds)
template <bool BIND>
__global__ __launch_bounds__(kBlock, kMinWaves) void dotp_round_kernel(const fq* __restrict__ derefs, const fq* __restrict__ vals,
                                                                       size_t N, fq* __restrict__ scratch, bool from_scratch,
                                                                       size_t pairs, fq r, Finisher fin, CircIds kmap) {
  const int k = kmap.v[blockIdx.y], m = k >> 1, half = k & 1;  // the half this workgroup row proves (identity on one GPU)
  const size_t hN = N / 2, q4 = N / 4;
  const fq* src[3];
  fq* dst[3];
#pragma unroll
  for (int t = 0; t < 3; t++) dst[t] = scratch + (size_t)(3 * blockIdx.y + t) * q4;  // scratch is numbered by the LOCAL half
  if (from_scratch) {
#pragma unroll
    for (int t = 0; t < 3; t++) src[t] = dst[t];
  } else {
    src[0] = derefs + (size_t)m * N + (size_t)half * hN;
    src[1] = derefs + (size_t)(3 + m) * N + (size_t)half * hN;
    src[2] = vals + (size_t)m * N + (size_t)half * hN;
  }
  Acc<4> acc;
  acc.init();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < pairs; i += (size_t)gridDim.x * kBlock) {
    fq u[3], p1, d1, p2, d2;
    if (BIND) { fold_pd2(src[0], dst[0], i, pairs, r, p1, d1); fold_pd2(src[1], dst[1], i, pairs, r, p2, d2); }
    else { load_pd(src[0], i, pairs, p1, d1); load_pd(src[1], i, pairs, p2, d2); }
    acc.stage_bc(u, p1, d1, p2, d2);
    if (BIND) fold_pd2(src[2], dst[2], i, pairs, r, p1, d1);
    else load_pd(src[2], i, pairs, p1, d1);
    acc.stage_a(u, p1, d1);
  }
  finish_block(acc.e, fin);
}

// Second half of a launch group whose kernels ran with Finisher::fused == 0: one wave per instance sums the instance's
// workgroup partials (product circuits: np workgroups each at `partials`; dot-product halves: nd each at `partials_d`),
// writes the three scalars to pinned host memory, and the workgroup publishes the group's sequence number.
constexpr int kFinishBlock = 256;  // one wave per SIMD: fits beside the other lanes' resident workgroups (a 1024-thread
                                   // workgroup waited 155 us on average for a CU during the 2^25 instance's row commitments)
__global__ __launch_bounds__(kFinishBlock) void round_finish_kernel(const fq* __restrict__ partials, int ncirc, int np,
                                                                    const fq* __restrict__ partials_d, int ndotp, int nd,
                                                                    fq* __restrict__ out, uint32_t* __restrict__ flag, uint32_t seq) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int inst = wave; inst < ncirc + ndotp; inst += kFinishBlock / 64) {
    const bool dp = inst >= ncirc;
    const fq* p = dp ? partials_d + (size_t)(inst - ncirc) * nd * 3 : partials + (size_t)inst * np * 3;
    const int nb = dp ? nd : np;
    fq e[3] = {fq_zero(), fq_zero(), fq_zero()};
    for (int b = lane; b < nb; b += 64)
#pragma unroll
      for (int k = 0; k < 3; k++) e[k] = fq_add(e[k], fq_load(&p[(size_t)b * 3 + k]));
#pragma unroll
    for (int k = 0; k < 3; k++) e[k] = fq_wave_sum(e[k]);
    if (lane == 0) {
      const int slot = dp ? 12 + (inst - ncirc) : inst;
#pragma unroll
      for (int k = 0; k < 3; k++) fq_store(&out[3 * (size_t)slot + k], e[k]);
      __threadfence_system();
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    *(volatile uint32_t*)flag = seq;
  }
}

// per-instance finisher: out[3*(inst0 + y) + k] = sum over the instance's block partials
__global__ __launch_bounds__(kBlock) void inst_finish_kernel(const fq* __restrict__ partials, int nblocks, int inst0,
                                                             fq* __restrict__ out) {
  const fq* p = partials + (size_t)blockIdx.x * nblocks * 3;
  fq e[3] = {fq_zero(), fq_zero(), fq_zero()};
  for (int b = threadIdx.x; b < nblocks; b += kBlock)
#pragma unroll
    for (int k = 0; k < 3; k++) e[k] = fq_add(e[k], fq_load(&p[(size_t)b * 3 + k]));
  __shared__ fq sh[kBlock / 64][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    fq s = fq_wave_sum(e[k]);
    if (lane == 0) sh[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    fq s = sh[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kBlock / 64; w++) s = fq_add(s, sh[w][threadIdx.x]);
    fq_store(&out[3 * (size_t)(inst0 + blockIdx.x) + threadIdx.x], s);
  }
}

// The collectors gather both groups' live entries in ONE single-workgroup launch and flag completion.
__global__ __launch_bounds__(128) void collect_kernel(const fq* __restrict__ forest, size_t stride, size_t off, size_t h, int ncirc,
                                                      const fq* __restrict__ derefs, const fq* __restrict__ vals, size_t N,
                                                      const fq* __restrict__ scratch, bool with_dotp, bool from_scratch,
                                                      fq* __restrict__ out, uint32_t* flag, uint32_t seq) {
  if (threadIdx.x < 64) {
    int t = threadIdx.x >> 2, w = threadIdx.x & 3;  // w: A[0], A[1], B[0], B[1]
    if (t < ncirc) {
      const fq* A = forest + (size_t)t * stride + off;
      fq_store(out + 4 * t + w, fq_load(A + (w >> 1) * h + (w & 1)));
    }
  } else if (with_dotp) {
    int idx = threadIdx.x - 64;
    if (idx < 36) {
      int k = idx / 6, t = (idx % 6) >> 1, e = idx & 1, m = k >> 1, half = k & 1;
      const size_t hN = N / 2;
      const fq* src;
      if (from_scratch) src = scratch + (size_t)(3 * k + t) * (N / 4);
      else src = (t == 0 ? derefs + (size_t)m * N : t == 1 ? derefs + (size_t)(3 + m) * N : vals + (size_t)m * N) + (size_t)half * hN;
      fq_store(out + 64 + idx, fq_load(src + e));
    }
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) *(volatile uint32_t*)flag = seq;
}

__global__ __launch_bounds__(kBlock) void fetch_tops_kernel(const fq* __restrict__ forest, size_t stride, size_t cnt, int ncirc,
                                                            fq* __restrict__ out) {
  size_t total = cnt * (size_t)ncirc;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    size_t t = i / cnt, j = i % cnt;
    fq_store(out + i, fq_load(forest + t * stride + (stride - cnt) + j));
  }
}

// Three slices per workgroup row: the eq table is read once per three slices instead of once per slice (the hash layer
// evaluates 6 + 15 + 2 slices against two eq tables of N and M entries: 23 -> 9 reads of an eq table).
// blockIdx.y = group g: slices 3g .. 3g+2; partial k of the group is slice 3g + k.
// r0 / step: the entries r0, r0 + step, .. of every slice against eq[0 .. len/step) (one proof over several GPUs, split by
// residue class: eq is then the table of the shortened point, the caller scales the sums); 0 / 1: all entries
__global__ __launch_bounds__(kBlock) void slice_dot3_kernel(const fq* __restrict__ table, size_t len, int nslices,
                                                            const fq* __restrict__ eq, fq* __restrict__ partials, size_t r0, size_t step) {
  const int s0 = 3 * blockIdx.y;
  const fq* t0 = table + (size_t)s0 * len;
  const bool has1 = s0 + 1 < nslices, has2 = s0 + 2 < nslices;
  const size_t nloc = len / step;
  fq e[3] = {fq_zero(), fq_zero(), fq_zero()};
  for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < nloc; k += (size_t)gridDim.x * kBlock) {
    const size_t i = r0 + k * step;
    const fq q = fq_load(eq + k);
    fq v = fq_load(t0 + i);
    if (!fq_is_zero(v)) e[0] = fq_add(e[0], fq_mul(v, q));
    if (has1) { v = fq_load(t0 + len + i); if (!fq_is_zero(v)) e[1] = fq_add(e[1], fq_mul(v, q)); }
    if (has2) { v = fq_load(t0 + 2 * len + i); if (!fq_is_zero(v)) e[2] = fq_add(e[2], fq_mul(v, q)); }
  }
  block_reduce_store<3>(e, partials + (size_t)blockIdx.y * gridDim.x * 3);
}

// blockIdx.y = dot-product circuit half k: partial sums of L*R*W
__global__ __launch_bounds__(kBlock) void triple_sum_kernel(const fq* __restrict__ derefs, const fq* __restrict__ vals, size_t N,
                                                            fq* __restrict__ partials, CircIds kmap) {
  const int k = kmap.v[blockIdx.y], m = k >> 1, half = k & 1;
  const size_t hN = N / 2;
  const fq* L = derefs + (size_t)m * N + (size_t)half * hN;
  const fq* R = derefs + (size_t)(3 + m) * N + (size_t)half * hN;
  const fq* W = vals + (size_t)m * N + (size_t)half * hN;
  fq acc = fq_zero();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < hN; i += (size_t)gridDim.x * kBlock) {
    fq w = fq_load(W + i);
    if (!fq_is_zero(w)) acc = fq_add(acc, fq_mul(fq_mul(fq_load(L + i), fq_load(R + i)), w));
  }
  fq e[3] = {acc, fq_zero(), fq_zero()};
  block_reduce_store<3>(e, partials + (size_t)blockIdx.y * gridDim.x * 3);
}

// ---- launchers -------------------------------------------------------------------------------

int spark_pinned(vpin_ctx* c) {
  if (c->h_spark) return VPIN_OK;
  (void)hipSetDevice(c->device);
  VPIN_HIP_TRY(hipHostMalloc((void**)&c->h_spark, kSparkPinned * sizeof(fq), hipHostMallocDefault));
  memset(c->h_spark, 0, kSparkPinned * sizeof(fq));
  VPIN_HIP_TRY(hipMalloc((void**)&c->d_spark_cnt, 2 * kSparkMaxInst * sizeof(uint32_t)));
  VPIN_HIP_TRY(hipMemsetAsync(c->d_spark_cnt, 0, 2 * kSparkMaxInst * sizeof(uint32_t), c->stream));
  VPIN_HIP_TRY(hipMalloc((void**)&c->d_tail_cnt, kSparkMaxInst * sizeof(uint32_t)));
  VPIN_HIP_TRY(hipMalloc((void**)&c->d_tail_red, (size_t)kSparkMaxInst * 8 * 2 * sizeof(fq)));  // kTailMaxWgs = 8
  VPIN_HIP_TRY(hipMemsetAsync(c->d_tail_cnt, 0, kSparkMaxInst * sizeof(uint32_t), c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  c->spark_seq = 0;
  return VPIN_OK;
}

static inline uint32_t* flag_ptr(vpin_ctx* c) { return reinterpret_cast<uint32_t*>(c->h_spark + (kSparkPinned - 1)); }

// Wait for the launch group that carries the current sequence number: spin briefly on the pinned flag
// word (the kernels' results are fenced before it), then fall back to an ordinary stream sync --
// tools that intercept dispatches (rocprofv3) may hold a launch back until the host synchronises.
int spark_wait_flag(vpin_ctx* c) {
  volatile uint32_t* f = flag_ptr(c);
  const uint32_t want = c->spark_seq;
  for (int spins = 0; spins < 20000; spins++) {  // ~100-200 us
    if (*f == want) return VPIN_OK;
    __builtin_ia32_pause();
  }
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  if (*f != want) { set_last_error("spark_wait_flag: launch group did not publish its flag", hipErrorUnknown); return VPIN_EHIP; }
  return VPIN_OK;
}

int spark_wait(vpin_ctx* c) {
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

int spark_u32_to_fq(vpin_ctx* c, const uint32_t* src, fq* dst, size_t n) {
  ProfScope ps(c, VPIN_K_SPARK_BUILD, 36.0 * (double)n);
  hipLaunchKernelGGL(u32_to_fq_kernel, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, src, dst, n);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}

// a table of `len` scalars straight from the driver (no pool, no owner: whoever frees the decommitment frees it, whichever
// contexts are still alive)
static int table_alloc_unpooled(vpin_ctx* c, size_t len, vpin_table** out) {
  vpin_table* t = new (std::nothrow) vpin_table();
  if (!t) return VPIN_ENOMEM;
  if (driver_malloc((void**)&t->d, len * sizeof(fq)) != hipSuccess) {
    (void)hipGetLastError();
    dev_pool_release(c);  // this context's cached blocks back to the driver, then once more (as dev_alloc does)
    if (driver_malloc((void**)&t->d, len * sizeof(fq)) != hipSuccess) { (void)hipGetLastError(); delete t; return VPIN_ENOMEM; }
  }
  note_driver_alloc(len * sizeof(fq));
  t->len = t->cap = len;
  t->owner = nullptr;
  *out = t;
  return VPIN_OK;
}

int spark_comb_make(vpin_ctx* c, const vpin_spark_decomm* d, vpin_table** ops, vpin_table** mem, bool pooled) {
  if (!c || !d || !d->idx || !d->vals || !ops || !mem) return VPIN_EINVAL;
  const size_t N = d->N, M = d->M;
  int rc;
  *ops = *mem = nullptr;
  // VPIN_ENCODE_LAPS=1: host time of every call below on stderr when the whole takes > 100 ms (round 6: SNARK::encode of the 2^25
  // instance took 68 ms or, every few proofs, 0.5-2 s with the GPU idle -- this is where)
  static const bool laps_on = getenv("VPIN_ENCODE_LAPS") != nullptr;
  using clk = std::chrono::steady_clock;
  clk::time_point tp[8];
  int np = 0;
  auto mark = [&] { if (laps_on && np < 8) tp[np++] = clk::now(); };
  mark();
  if ((rc = pooled ? table_alloc_uninit(c, 16 * N, ops) : table_alloc_unpooled(c, 16 * N, ops))) return rc;
  if ((rc = pooled ? table_alloc_uninit(c, 2 * M, mem) : table_alloc_unpooled(c, 2 * M, mem))) { vpin_table_free(c, *ops); *ops = nullptr; return rc; }
  mark();
  if ((rc = spark_u32_to_fq(c, d->idx, (*ops)->d, 12 * N)) || (rc = spark_u32_to_fq(c, d->idx + 12 * N, (*mem)->d, 2 * M))) return rc;
  mark();
  VPIN_HIP_TRY(hipMemcpyAsync((*ops)->d + 12 * N, d->vals, 3 * N * sizeof(fq), hipMemcpyDeviceToDevice, c->stream));
  mark();
  VPIN_HIP_TRY(hipMemsetAsync((*ops)->d + 15 * N, 0, N * sizeof(fq), c->stream));
  mark();
  if (laps_on) {
    (void)hipStreamSynchronize(c->stream);
    mark();
    auto ms = [&](int i) { return std::chrono::duration<double, std::milli>(tp[i + 1] - tp[i]).count(); };
    if (std::chrono::duration<double, std::milli>(tp[np - 1] - tp[0]).count() > 100.0)
      fprintf(stderr, "[comb_make] N 2^%d: alloc %.2f | u32_to_fq launches %.2f | hipMemcpyAsync D2D call %.2f | hipMemsetAsync call %.2f | "
                      "stream sync %.2f ms\n", (int)__builtin_ctzll(N), ms(0), ms(1), ms(2), ms(3), ms(4));
  }
  return VPIN_OK;
}

int spark_comb_tables(vpin_ctx* c, vpin_spark_decomm* d) {
  if (!d) return VPIN_EINVAL;
  std::lock_guard<std::mutex> g(d->comb_mu);
  if (d->comb_ops && d->comb_mem) return VPIN_OK;
  spark_comb_release(c, d, false);
  // (round 6: this passed comb_unpooled as `pooled` -- every SNARK::encode took its 16N + 2M scalars straight from the driver and
  // gave them back, a 20 GB hipMalloc / hipFree pair that costs 0.3 ms most of the time and 0.2-2.8 s every few proofs: the
  // unstable encode_ms and the doubled reference span of round 5)
  int rc = spark_comb_make(c, d, &d->comb_ops, &d->comb_mem, /*pooled=*/!d->comb_unpooled);
  if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = VPIN_EHIP;  // other contexts' streams read them next
  if (rc) spark_comb_release(c, d, false);
  return rc;
}

// to_driver: the blocks leave the context's pool too (SNARK::encode: nothing of a proof reuses 16N scalars; cached, they would
// only inflate the pool -- the caller has synchronised the stream)
void spark_comb_release(vpin_ctx* c, vpin_spark_decomm* d, bool to_driver) {
  if (!d) return;
  for (vpin_table** t : {&d->comb_ops, &d->comb_mem}) {
    if (!*t) continue;
    if (to_driver && (*t)->owned && (*t)->owner == c) {
      dev_release_block(c, (*t)->d);
      delete *t;
    } else {
      vpin_table_free(c, *t);
    }
    *t = nullptr;
  }
}

// cnt[2m + k] += entries of matrix m's col slice equal to candidate k
__global__ __launch_bounds__(kBlock) void count_cols_kernel(const uint32_t* __restrict__ idx, size_t N, uint32_t v0, uint32_t v1,
                                                            unsigned long long* __restrict__ cnt) {
  const int m = blockIdx.y;
  const uint32_t* a = idx + (size_t)(6 + m) * N;
  unsigned n0 = 0, n1 = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (size_t)gridDim.x * kBlock) {
    const uint32_t x = a[i];
    n0 += x == v0;
    n1 += x == v1;
  }
  __shared__ unsigned sh[2][kBlock];
  sh[0][threadIdx.x] = n0;
  sh[1][threadIdx.x] = n1;
  __syncthreads();
  for (int st = kBlock / 2; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) { sh[0][threadIdx.x] += sh[0][threadIdx.x + st]; sh[1][threadIdx.x] += sh[1][threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (sh[0][0]) atomicAdd(&cnt[2 * m], (unsigned long long)sh[0][0]);
    if (sh[1][0]) atomicAdd(&cnt[2 * m + 1], (unsigned long long)sh[1][0]);
  }
}

int spark_find_hot_cols(vpin_ctx* c, vpin_spark_decomm* d) {
  if (!c || !d || !d->idx || d->N == 0) return VPIN_EINVAL;
  for (int m = 0; m < 3; m++) d->hot_col[m] = 0xffffffffu;
  static const bool off = getenv("VPIN_NO_HOT_COLS") != nullptr;
  if (off || d->N < ((size_t)1 << 20)) return VPIN_OK;  // below ~2^20 entries the per-proof T tables cost more than they save
  (void)hipSetDevice(c->device);
  DevBuf cnt(c);
  if (cnt.alloc(6 * sizeof(unsigned long long))) return VPIN_ENOMEM;
  VPIN_HIP_TRY(hipMemsetAsync(cnt.p, 0, 6 * sizeof(unsigned long long), c->stream));
  const uint32_t v0 = (uint32_t)d->num_vars, v1 = (uint32_t)d->num_vars + 1;
  hipLaunchKernelGGL(count_cols_kernel, dim3(grid_for(d->N), 3), dim3(kBlock), 0, c->stream, (const uint32_t*)d->idx, d->N, v0, v1,
                     (unsigned long long*)cnt.p);
  VPIN_HIP_TRY(hipGetLastError());
  unsigned long long h[6];
  VPIN_HIP_TRY(hipMemcpyAsync(h, cnt.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  for (int m = 0; m < 3; m++) {
    const int k = h[2 * m + 1] > h[2 * m] ? 1 : 0;
    if (h[2 * m + k] >= d->N / 64) d->hot_col[m] = k ? v1 : v0;  // at least 1.5 % of the slice
  }
  return VPIN_OK;
}

int spark_gather_derefs(vpin_ctx* c, const vpin_spark_decomm* d, const fq* mem_rx, const fq* mem_ry, fq* comb) {
  ProfScope ps(c, VPIN_K_SPARK_BUILD, (double)d->N * (6 * 68.0 + 2 * 32.0));
  hipLaunchKernelGGL(gather_derefs_kernel, dim3(grid_for(d->N), 8), dim3(kBlock), 0, c->stream, (const uint32_t*)d->idx, d->N,
                     mem_rx, mem_ry, comb);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}

static int build_levels(vpin_ctx* c, SparkForest* f) {
  const size_t n2 = 2 * f->n;
  size_t h = f->n / 2;  // entries of the level being built
  int l = 0;
  static const bool one_by_one = getenv("VPIN_TREE_LEVEL_BY_LEVEL") != nullptr;
  while (h > (size_t)kTopBlock) {
    if (!one_by_one && h / 4 > (size_t)kTopBlock) {  // three levels per pass while the third is still a grid's worth
      hipLaunchKernelGGL(tree_level3_kernel, dim3(grid_for(h / 4), f->ncirc), dim3(kBlock), 0, c->stream, f->base, f->stride(),
                         f->level_off(l), f->level_off(l + 1), f->level_off(l + 2), f->level_off(l + 3), h);
      h >>= 3;
      l += 3;
      continue;
    }
    hipLaunchKernelGGL(tree_level_kernel, dim3(grid_for(h), f->ncirc), dim3(kBlock), 0, c->stream, f->base, f->stride(),
                       f->level_off(l), f->level_off(l + 1), h);
    h >>= 1;
    l++;
  }
  if (h >= 1) hipLaunchKernelGGL(tree_top_kernel, dim3(f->ncirc), dim3(kTopBlock), 0, c->stream, f->base, f->stride(), n2, h);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}

int spark_build_forests(vpin_ctx* c, const vpin_spark_decomm* d, const fq* comb_derefs, const fq* mem_rx, const fq* mem_ry,
                        const uint8_t r_hash[32], const uint8_t r_hash_sqr[32], const uint8_t r_hash_sqr_boost[32],
                        const uint8_t gamma[32], SparkForest* ops, SparkForest* mem) {
  if (!c || !d || !ops || !mem || !ops->base || !mem->base) return VPIN_EINVAL;
  if (ops->n != d->N || mem->n != d->M || ops->ncirc != 12 || mem->ncirc != 4 || d->N < 2 || d->M < 2) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  HashParams hp;
  hp.r = load_host_fq(r_hash);
  hp.r2 = load_host_fq(r_hash_sqr);
  hp.r2_boost = load_host_fq(r_hash_sqr_boost);
  hp.gamma = load_host_fq(gamma);
  {
    // leaves: 6 gathers of (addr, ts, val) -> 12 N leaves; 2 x (ts, mem) -> 4 M leaves; then 1 product per inner node
    ProfScope ps(c, VPIN_K_SPARK_BUILD, (double)d->N * (6 * (8.0 + 32.0) + 12 * 32.0 + 12 * 64.0) + (double)d->M * (2 * 36.0 + 4 * 32.0 + 4 * 64.0));
    // (leaves and the first three levels in ONE pass -- eight strided leaves per thread -- were measured and dropped: 21.2 ms
    // against 18.7 ms for the 2^25 instance with the leaf kernels below and three tree levels per pass after them)
    hipLaunchKernelGGL(hash_ops_kernel, dim3(grid_for(d->N), 6), dim3(kBlock), 0, c->stream, (const uint32_t*)d->idx, comb_derefs,
                       d->N, hp, ops->base);
    hipLaunchKernelGGL(hash_mem_kernel, dim3(grid_for(d->M), 2), dim3(kBlock), 0, c->stream,
                       (const uint32_t*)(d->idx + 12 * d->N), mem_rx, mem_ry, d->M, hp, mem->base);
    VPIN_HIP_TRY(hipGetLastError());
    int rc = build_levels(c, ops);
    if (!rc) rc = build_levels(c, mem);
    if (rc) return rc;
  }
  return VPIN_OK;
}

// ---- the mem circuits' roots without their trees (round 5) --------------------------------------------------------------
// ProductLayerProof::prove sends the 16 circuits' roots before the first sum-check; the mem forest itself (4 circuits over M
// leaves: 16 GiB for the 2^25 instance) is only proven AFTER the ops forest (24 GiB).  Building it then, into the memory the
// ops forest frees, takes a third off the proof's working set; what is needed early is only the product of each circuit's
// leaves -- the same field element as the tree's root (multiplication mod q is exact and commutative).
__device__ __forceinline__ fq fq_wave_prod(fq a) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) a = fq_mul(a, fq_shfl_xor(a, off));
  return a;
}
// blockIdx.y = side; parts[(2 * side + kind) * gridDim.x + blockIdx.x] = product of this block's leaves of circuit 2*side + kind
__global__ __launch_bounds__(kBlock) void hash_mem_roots_kernel(const uint32_t* __restrict__ audit_ts, const fq* __restrict__ mem_rx,
                                                                const fq* __restrict__ mem_ry, size_t M, HashParams hp,
                                                                fq* __restrict__ parts) {
  const int side = blockIdx.y;
  const fq* mem = side ? mem_ry : mem_rx;
  const uint32_t* ts = audit_ts + (size_t)side * M;
  const fq r2c = fq_r2();
  fq pi = fq_one(), pa = fq_one();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < M; i += (size_t)gridDim.x * kBlock) {
    fq h = fq_sub(fq_mul(fq_load(mem + i), hp.r), hp.gamma);
    if (i) h = fq_add(h, fq_mul(fq_raw_u32((uint32_t)i), r2c));
    pi = fq_mul(pi, h);
    const uint32_t t = ts[i];
    if (t) h = fq_add(h, fq_mul(fq_raw_u32(t), hp.r2_boost));
    pa = fq_mul(pa, h);
  }
  pi = fq_wave_prod(pi);
  pa = fq_wave_prod(pa);
  __shared__ fq sh[kBlock / 64][2];
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = pi; sh[threadIdx.x >> 6][1] = pa; }
  __syncthreads();
  if (threadIdx.x < 2) {
    fq t = sh[0][threadIdx.x];
    for (int w = 1; w < kBlock / 64; w++) t = fq_mul(t, sh[w][threadIdx.x]);
    fq_store(parts + (size_t)(2 * side + (int)threadIdx.x) * gridDim.x + blockIdx.x, t);
  }
}
// out[2 * circ] = product of the np partial products of circuit circ (blockIdx.x); the h_spark layout of spark_fetch_tops(f, 2)
__global__ __launch_bounds__(kBlock) void roots_finish_kernel(const fq* __restrict__ parts, int np, fq* __restrict__ out) {
  fq p = fq_one();
  for (int k = (int)threadIdx.x; k < np; k += kBlock) p = fq_mul(p, fq_load(parts + (size_t)blockIdx.x * np + k));
  p = fq_wave_prod(p);
  __shared__ fq sh[kBlock / 64];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = p;
  __syncthreads();
  if (threadIdx.x == 0) {
    fq t = sh[0];
    for (int w = 1; w < kBlock / 64; w++) t = fq_mul(t, sh[w]);
    fq_store(out + 2 * blockIdx.x, t);
    fq_store(out + 2 * blockIdx.x + 1, fq_zero());
  }
}

static HashParams hash_params(const uint8_t r_hash[32], const uint8_t r_hash_sqr[32], const uint8_t r_hash_sqr_boost[32],
                              const uint8_t gamma[32]) {
  HashParams hp;
  hp.r = load_host_fq(r_hash);
  hp.r2 = load_host_fq(r_hash_sqr);
  hp.r2_boost = load_host_fq(r_hash_sqr_boost);
  hp.gamma = load_host_fq(gamma);
  return hp;
}

// roots of the four mem circuits (init / audit per side) -> c->h_spark[2 * circuit], as spark_fetch_tops(&mem_forest, 2) leaves them
int spark_mem_roots(vpin_ctx* c, const vpin_spark_decomm* d, const fq* mem_rx, const fq* mem_ry, const uint8_t r_hash[32],
                    const uint8_t r_hash_sqr[32], const uint8_t r_hash_sqr_boost[32], const uint8_t gamma[32]) {
  if (!c || !d || !mem_rx || !mem_ry || d->M < 2) return VPIN_EINVAL;
  int rc = spark_pinned(c);
  if (rc) return rc;
  (void)hipSetDevice(c->device);
  const int np = (int)std::min<size_t>(1024, (d->M + kBlock - 1) / kBlock);
  DevBuf parts(c);
  if (parts.alloc((size_t)4 * np * sizeof(fq))) return VPIN_ENOMEM;
  {
    ProfScope ps(c, VPIN_K_SPARK_BUILD, (double)d->M * 2 * 36.0);
    hipLaunchKernelGGL(hash_mem_roots_kernel, dim3((unsigned)np, 2), dim3(kBlock), 0, c->stream, (const uint32_t*)(d->idx + 12 * d->N), mem_rx,
                       mem_ry, d->M, hash_params(r_hash, r_hash_sqr, r_hash_sqr_boost, gamma), (fq*)parts.p);
  }
  hipLaunchKernelGGL(roots_finish_kernel, dim3(4), dim3(kBlock), 0, c->stream, (const fq*)parts.p, np, c->h_spark);
  VPIN_HIP_TRY(hipGetLastError());
  return spark_wait(c);
}

// the ops forest alone / the mem forest alone (spark_build_forests below does both)
int spark_build_forest_ops(vpin_ctx* c, const vpin_spark_decomm* d, const fq* comb_derefs, const uint8_t r_hash[32],
                           const uint8_t r_hash_sqr[32], const uint8_t r_hash_sqr_boost[32], const uint8_t gamma[32], SparkForest* ops) {
  if (!c || !d || !ops || !ops->base || !comb_derefs) return VPIN_EINVAL;
  if (ops->n != d->N || ops->ncirc != 12 || d->N < 2) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  ProfScope ps(c, VPIN_K_SPARK_BUILD, (double)d->N * (6 * (8.0 + 32.0) + 12 * 32.0 + 12 * 64.0));
  hipLaunchKernelGGL(hash_ops_kernel, dim3(grid_for(d->N), 6), dim3(kBlock), 0, c->stream, (const uint32_t*)d->idx, comb_derefs, d->N,
                     hash_params(r_hash, r_hash_sqr, r_hash_sqr_boost, gamma), ops->base);
  VPIN_HIP_TRY(hipGetLastError());
  return build_levels(c, ops);
}
int spark_build_forest_mem(vpin_ctx* c, const vpin_spark_decomm* d, const fq* mem_rx, const fq* mem_ry, const uint8_t r_hash[32],
                           const uint8_t r_hash_sqr[32], const uint8_t r_hash_sqr_boost[32], const uint8_t gamma[32], SparkForest* mem) {
  if (!c || !d || !mem || !mem->base || !mem_rx || !mem_ry) return VPIN_EINVAL;
  if (mem->n != d->M || mem->ncirc != 4 || d->M < 2) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  ProfScope ps(c, VPIN_K_SPARK_BUILD, (double)d->M * (2 * 36.0 + 4 * 32.0 + 4 * 64.0));
  hipLaunchKernelGGL(hash_mem_kernel, dim3(grid_for(d->M), 2), dim3(kBlock), 0, c->stream, (const uint32_t*)(d->idx + 12 * d->N), mem_rx,
                     mem_ry, d->M, hash_params(r_hash, r_hash_sqr, r_hash_sqr_boost, gamma), mem->base);
  VPIN_HIP_TRY(hipGetLastError());
  return build_levels(c, mem);
}

int spark_build_forest_sub(vpin_ctx* c, const vpin_spark_decomm* d, const fq* comb_derefs, const fq* mem_rx, const fq* mem_ry,
                           const uint8_t r_hash[32], const uint8_t r_hash_sqr[32], const uint8_t r_hash_sqr_boost[32],
                           const uint8_t gamma[32], SparkForest* f, const int* ids, bool is_mem) {
  if (!c || !d || !f || !f->base || !ids || f->ncirc < 1 || f->ncirc > (is_mem ? 4 : 12)) return VPIN_EINVAL;
  if (f->n != (is_mem ? d->M : d->N) || f->n < 2) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  HashParams hp;
  hp.r = load_host_fq(r_hash);
  hp.r2 = load_host_fq(r_hash_sqr);
  hp.r2_boost = load_host_fq(r_hash_sqr_boost);
  hp.gamma = load_host_fq(gamma);
  CircIds cid{};
  for (int j = 0; j < f->ncirc; j++) {
    if (ids[j] < 0 || ids[j] >= (is_mem ? 4 : 12)) return VPIN_EINVAL;
    cid.v[j] = ids[j];
  }
  const double per = is_mem ? (36.0 + 32.0 + 64.0) : (8.0 + 32.0 + 32.0 + 64.0);
  ProfScope ps(c, VPIN_K_SPARK_BUILD, (double)f->n * f->ncirc * per);
  if (is_mem)
    hipLaunchKernelGGL(hash_mem_sub_kernel, dim3(grid_for(d->M), f->ncirc), dim3(kBlock), 0, c->stream,
                       (const uint32_t*)(d->idx + 12 * d->N), mem_rx, mem_ry, d->M, hp, f->base, cid);
  else
    hipLaunchKernelGGL(hash_ops_sub_kernel, dim3(grid_for(d->N), f->ncirc), dim3(kBlock), 0, c->stream, (const uint32_t*)d->idx,
                       comb_derefs, d->N, hp, f->base, cid);
  VPIN_HIP_TRY(hipGetLastError());
  return build_levels(c, f);
}

int spark_gather_derefs_strided(vpin_ctx* c, const vpin_spark_decomm* d, const fq* mem_rx, const fq* mem_ry, size_t r0, size_t step,
                                fq* comb_loc, fq* comb_rows, size_t R, size_t nrows_loc) {
  if (!c || !d || !mem_rx || !mem_ry || !comb_loc || !comb_rows || step == 0 || r0 >= step || d->N % step || R == 0 || d->N % R)
    return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  ProfScope ps(c, VPIN_K_SPARK_BUILD, ((double)d->N * 6 * 68.0 + (double)nrows_loc * R * 68.0) / 1.0);
  hipLaunchKernelGGL(gather_derefs_loc_kernel, dim3(grid_for(d->N / step), 6), dim3(kBlock), 0, c->stream, (const uint32_t*)d->idx, d->N,
                     mem_rx, mem_ry, comb_loc, r0, step);
  if (nrows_loc)
    hipLaunchKernelGGL(gather_derefs_rows_kernel, dim3(grid_for(nrows_loc * R)), dim3(kBlock), 0, c->stream, (const uint32_t*)d->idx, d->N,
                       R, mem_rx, mem_ry, comb_rows, r0, step, nrows_loc);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}

int spark_take_strided(vpin_ctx* c, const fq* src, size_t nloc, size_t r0, size_t step, fq* dst) {
  if (!c || !src || !dst || step == 0 || r0 >= step) return VPIN_EINVAL;
  if (nloc == 0) return VPIN_OK;
  hipLaunchKernelGGL(take_strided_kernel, dim3(grid_for(nloc)), dim3(kBlock), 0, c->stream, src, nloc, r0, step, dst);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}

// local forests of rank r0 of `step`: ops over N/step leaves, mem over M/step leaves, all 12 / 4 circuits
int spark_build_forests_strided(vpin_ctx* c, const vpin_spark_decomm* d, const fq* comb_loc, const fq* mem_rx, const fq* mem_ry,
                                const uint8_t r_hash[32], const uint8_t r_hash_sqr[32], const uint8_t r_hash_sqr_boost[32],
                                const uint8_t gamma[32], SparkForest* ops, SparkForest* mem, size_t r0, size_t step) {
  if (!c || !d || !ops || !mem || !ops->base || !mem->base || step == 0 || r0 >= step) return VPIN_EINVAL;
  if (ops->n * step != d->N || mem->n * step != d->M || ops->ncirc != 12 || mem->ncirc != 4 || ops->n < 2 || mem->n < 2) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  HashParams hp;
  hp.r = load_host_fq(r_hash);
  hp.r2 = load_host_fq(r_hash_sqr);
  hp.r2_boost = load_host_fq(r_hash_sqr_boost);
  hp.gamma = load_host_fq(gamma);
  ProfScope ps(c, VPIN_K_SPARK_BUILD, ((double)d->N * (6 * (8.0 + 32.0) + 12 * 32.0 + 12 * 64.0) + (double)d->M * (2 * 36.0 + 4 * 32.0 + 4 * 64.0)) / (double)step);
  hipLaunchKernelGGL(hash_ops_strided_kernel, dim3(grid_for(ops->n), 6), dim3(kBlock), 0, c->stream, (const uint32_t*)d->idx, comb_loc, d->N, hp,
                     ops->base, r0, step);
  hipLaunchKernelGGL(hash_mem_strided_kernel, dim3(grid_for(mem->n), 2), dim3(kBlock), 0, c->stream,
                     (const uint32_t*)(d->idx + 12 * d->N), mem_rx, mem_ry, d->M, hp, mem->base, r0, step);
  VPIN_HIP_TRY(hipGetLastError());
  int rc = build_levels(c, ops);
  if (!rc) rc = build_levels(c, mem);
  return rc;
}

int spark_fetch_tops(vpin_ctx* c, const SparkForest* f, size_t cnt) {
  if (cnt * (size_t)f->ncirc > kSparkPinned || cnt > f->stride()) return VPIN_ESHAPE;
  int rc = spark_pinned(c);
  if (rc) return rc;
  hipLaunchKernelGGL(fetch_tops_kernel, dim3(grid_for(cnt * f->ncirc)), dim3(kBlock), 0, c->stream, (const fq*)f->base, f->stride(),
                     cnt, f->ncirc, c->h_spark);
  VPIN_HIP_TRY(hipGetLastError());
  return spark_wait(c);
}

// block partial scratch for up to kSparkMaxInst instances x kRoundBlocks blocks
constexpr int kRoundBlocks = 4096;  // layout stride of the partial buffer; the launch cap is round_blocks()

// Workgroups per circuit of a round kernel (and pairs per thread below which a round gets fewer).  Round 1 found 64 x 12
// workgroups with >= 8 pairs per thread 10 % faster than 512 x 12 with 2 and blamed the per-circuit atomics; the cost was
// the agent-scope release fence of the fused finisher, an L2 write-back per workgroup (finish_block).  With the partials
// summed by a second launch a round of 2^16 pairs went 173 -> 71 us (tools/ubench_rounds.py) and more, shorter workgroups
// help: the LeNet step 490 / 476 / 473 / 476 / 465 ms at 64 / 128 / 256 / 512 / 1024, the CNN E trace 73.9 / 70.5 /
// 69.8 / 68.9 / 69.9 ms, the 2^25 instance alone flat (same box, round 2).
static inline int round_blocks() {
  static const int n = [] { const char* e = getenv("VPIN_ROUND_BLOCKS"); int v = e ? atoi(e) : 256; return v < 1 ? 1 : v > kRoundBlocks ? kRoundBlocks : v; }();
  return n;
}

static int round_partials(vpin_ctx* c, fq** out) {
  if (c->partials_cap < (size_t)kSparkMaxInst * kRoundBlocks * 3) return VPIN_ENOMEM;
  *out = c->d_partials;
  return VPIN_OK;
}

// Pairs per thread: at least VPIN_SPARK_PAIRS_PER_THREAD (default 1) until the launch reaches round_blocks() workgroups per
// circuit, grid-stride beyond.  A thread's pairs are serial work (~7 us each in a lone wave), so mid-size rounds (2^11..2^16
// pairs), which cannot fill the device anyway, get one pair per thread: 75 -> ~25 us per round measured on the 2^16 instance;
// the streaming rounds (>= 2^17 pairs per circuit) sit at the cap either way and keep their 8+ pairs per thread.
static inline int round_grid(size_t pairs, int ncirc = 12, bool shared_device = false) {
  static const size_t per_thread = [] { const char* e = getenv("VPIN_SPARK_PAIRS_PER_THREAD"); size_t v = e ? (size_t)atoi(e) : 1; return v ? v : 1; }();
  size_t b = (pairs + kBlock * per_thread - 1) / (kBlock * per_thread);
  if (b < 1) b = 1;
  // A proof that has the device to itself caps the LAUNCH, not the circuit, so the 4 "mem" circuits and the 6 dot-product
  // halves get proportionally more workgroups each; on a shared device (vpin_ctx_set_shared_device: other contexts prove at
  // the same time) the per-circuit cap stays and the leftover CUs are the other streams'.
  const size_t cap0 = (size_t)round_blocks();
  size_t cap = shared_device ? cap0 : cap0 * 12 / (size_t)(ncirc < 1 ? 1 : ncirc > 12 ? 12 : ncirc);
  if (cap > (size_t)kRoundBlocks) cap = kRoundBlocks;
  if (b > cap) b = cap;
  return (int)b;
}

static Finisher make_finisher(vpin_ctx* c, fq* partials, int inst0, int total_inst) {
  Finisher f;
  f.partials = partials;
  f.counters = c->d_spark_cnt;
  f.out = c->h_spark;
  f.flag = flag_ptr(c);
  f.seq = c->spark_seq;
  f.inst0 = inst0;
  f.total_inst = total_inst;
  f.fused = 1;
  return f;
}

// launch groups of more workgroups than this sum their partials in a second, one-workgroup launch (finish_block)
static inline size_t fused_finish_max() {
  static const size_t n = [] { const char* e = getenv("VPIN_SPARK_FUSED_FINISH_MAX"); long v = e ? atol(e) : 64; return (size_t)(v < 0 ? 0 : v); }();
  return n;
}

static int round_finish_launch(vpin_ctx* c, const fq* partials, int ncirc, int np, int ndotp, int nd) {
  hipLaunchKernelGGL(round_finish_kernel, dim3(1), dim3(kFinishBlock), 0, c->stream, partials, ncirc, np,
                     partials + (size_t)12 * kRoundBlocks * 3, ndotp, nd, c->h_spark, flag_ptr(c), c->spark_seq);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}

int spark_prod_round(vpin_ctx* c, const SparkForest* f, int level, size_t len, const fq* E, const uint8_t* r, int ndotp,
                     bool lead) {
  const bool with_dotp = ndotp > 0;
  if (!c || !f || !f->base || !E) return VPIN_EINVAL;
  const size_t h = f->n >> (level + 1);
  if (h == 0 || len > h || !is_pow2(len) || len < (r ? 4u : 2u) || f->ncirc > 12 || f->ncirc < 1 || ndotp < 0 || ndotp > 6) return VPIN_ESHAPE;
  int rc = spark_pinned(c);
  if (rc) return rc;
  (void)hipSetDevice(c->device);
  fq* partials = nullptr;
  if ((rc = round_partials(c, &partials))) return rc;
  const size_t pairs = r ? len / 4 : len / 2;
  const int grid = round_grid(pairs, f->ncirc, c->shared_device);
  const fq rr = r ? load_host_fq(r) : fq{};
  const fq_const rconst = (r && lead) ? make_fq_const(r) : fq_const{};
  c->spark_seq++;  // a new launch group: this kernel (+ the dot-product kernel that follows when with_dotp)
  Finisher fin = make_finisher(c, partials, 0, f->ncirc + ndotp);
  // the dot-product kernel of the same group (spark_dotp_round, next call) gets round_grid(pairs, 6) workgroups per half
  const size_t group_blocks = (size_t)grid * f->ncirc + (with_dotp ? (size_t)round_grid(pairs, 6, c->shared_device) * ndotp : 0);
  fin.fused = group_blocks <= fused_finish_max() ? 1 : 0;
  c->round_split = fin.fused ? 0 : 1;
  c->round_split_grid = grid;
  c->round_split_ncirc = f->ncirc;
  c->round_group_ndotp = ndotp;
  {
    // algorithmic bytes of the reference formulation: A and B of every circuit and the shared eq table read,
    // folded halves written
    const double bytes = (double)f->ncirc * 2 * 32.0 * (r ? (double)len * 1.5 : (double)len) + 32.0 * (r ? (double)len * 1.5 : (double)len);
    ProfScope ps(c, VPIN_K_SPARK_ROUND, bytes, pairs >= ((size_t)1 << 20) ? VPIN_K_SPARK_ROUND_BIG : -1, (double)f->ncirc * (double)pairs);
#define VPIN_PROD_LAUNCH(B_, L_, G_)                                                                                              \
  hipLaunchKernelGGL((prod_round_kernel<B_, L_, G_>), dim3(grid, f->ncirc), dim3(kBlock), 0, c->stream, f->base, f->stride(), \
                     f->level_off(level), h, E, pairs, rr, rconst, fin)
    const bool big = lead && pairs >= ((size_t)1 << 20);
    if (r) { if (big) VPIN_PROD_LAUNCH(true, true, true); else if (lead) VPIN_PROD_LAUNCH(true, true, false); else VPIN_PROD_LAUNCH(true, false, false); }
    else { if (big) VPIN_PROD_LAUNCH(false, true, true); else if (lead) VPIN_PROD_LAUNCH(false, true, false); else VPIN_PROD_LAUNCH(false, false, false); }
#undef VPIN_PROD_LAUNCH
  }
  VPIN_HIP_TRY(hipGetLastError());
  if (!fin.fused && !with_dotp) return round_finish_launch(c, partials, f->ncirc, grid, 0, 0);
  return VPIN_OK;
}

int spark_dotp_round(vpin_ctx* c, size_t N, const fq* vals, const fq* comb_derefs, fq* scratch, size_t len, bool first_fold,
                     const uint8_t* r, const int* halves, int ndotp) {
  if (!c || !vals || !comb_derefs || !scratch || ndotp < 1 || ndotp > 6) return VPIN_EINVAL;
  if (ndotp != c->round_group_ndotp) return VPIN_ESHAPE;  // the launch group was announced with another count (spark_prod_round)
  CircIds kmap{};
  for (int i = 0; i < ndotp; i++) {
    kmap.v[i] = halves ? halves[i] : i;
    if (kmap.v[i] < 0 || kmap.v[i] > 5) return VPIN_EINVAL;
  }
  if (len > N / 2 || !is_pow2(len) || len < (r ? 4u : 2u) || (first_fold && (!r || len != N / 2))) return VPIN_ESHAPE;
  int rc = spark_pinned(c);
  if (rc) return rc;
  (void)hipSetDevice(c->device);
  fq* partials = nullptr;
  if ((rc = round_partials(c, &partials))) return rc;
  const size_t pairs = r ? len / 4 : len / 2;
  const int grid = round_grid(pairs, 6, c->shared_device);
  const fq rr = r ? load_host_fq(r) : fq{};
  const bool from_scratch = r && !first_fold;
  // same launch group as the product circuits' kernel just issued: slots 12.. of the group's instances
  Finisher fin = make_finisher(c, partials + (size_t)12 * kRoundBlocks * 3, 12, c->round_split_ncirc + ndotp);
  fin.fused = c->round_split ? 0 : 1;  // decided for the whole group by spark_prod_round
  {
    const double bytes = ndotp * 3 * 32.0 * (r ? (double)len * 1.5 : (double)len);
    ProfScope ps(c, VPIN_K_SPARK_ROUND, bytes);
    if (r)
      hipLaunchKernelGGL((dotp_round_kernel<true>), dim3(grid, ndotp), dim3(kBlock), 0, c->stream, comb_derefs, vals, N, scratch,
                         from_scratch, pairs, rr, fin, kmap);
    else
      hipLaunchKernelGGL((dotp_round_kernel<false>), dim3(grid, ndotp), dim3(kBlock), 0, c->stream, comb_derefs, vals, N, scratch,
                         false, pairs, rr, fin, kmap);
  }
  VPIN_HIP_TRY(hipGetLastError());
  if (!fin.fused) return round_finish_launch(c, partials, c->round_split_ncirc, c->round_split_grid, ndotp, grid);
  return VPIN_OK;
}

int spark_collect(vpin_ctx* c, const SparkForest* f, int level, const vpin_spark_decomm* d, const fq* comb_derefs,
                  const fq* scratch, bool with_dotp, bool folded) {
  if (!c || !f || !f->base || f->ncirc > 12 || (with_dotp && (!d || !comb_derefs || !scratch))) return VPIN_EINVAL;
  int rc = spark_pinned(c);
  if (rc) return rc;
  const size_t h = f->n >> (level + 1);
  if (h < 2) return VPIN_ESHAPE;  // a 1-entry half has no second element; the host handles those layers
  c->spark_seq++;
  hipLaunchKernelGGL(collect_kernel, dim3(1), dim3(128), 0, c->stream, (const fq*)f->base, f->stride(), f->level_off(level), h,
                     f->ncirc, comb_derefs, with_dotp ? (const fq*)d->vals : (const fq*)nullptr,
                     with_dotp ? d->N : (size_t)0, scratch, with_dotp, folded, c->h_spark, flag_ptr(c), c->spark_seq);
  VPIN_HIP_TRY(hipGetLastError());
  return spark_wait_flag(c);
}

int spark_triple_sums(vpin_ctx* c, const vpin_spark_decomm* d, const fq* comb_derefs, const int* halves, int nh) {
  if (!d) return VPIN_EINVAL;
  return spark_triple_sums_raw(c, comb_derefs, d->vals, d->N, halves, nh);
}

int spark_triple_sums_raw(vpin_ctx* c, const fq* comb_derefs, const fq* vals, size_t N, const int* halves, int nh) {
  if (!c || !comb_derefs || !vals || N < 2 || nh < 1 || nh > 6) return VPIN_EINVAL;
  CircIds kmap{};
  for (int i = 0; i < nh; i++) kmap.v[i] = halves ? halves[i] : i;
  int rc = spark_pinned(c);
  if (rc) return rc;
  (void)hipSetDevice(c->device);
  fq* partials = nullptr;
  if ((rc = round_partials(c, &partials))) return rc;
  const int grid = round_grid(N / 2, 6, c->shared_device);
  {
    ProfScope ps(c, VPIN_K_SPARK_BUILD, 1.5 * nh * 32.0 * (double)N);
    hipLaunchKernelGGL(triple_sum_kernel, dim3(grid, nh), dim3(kBlock), 0, c->stream, comb_derefs, vals, N, partials, kmap);
  }
  hipLaunchKernelGGL(inst_finish_kernel, dim3(nh), dim3(kBlock), 0, c->stream, (const fq*)partials, grid, 0, c->h_spark);
  VPIN_HIP_TRY(hipGetLastError());
  return spark_wait(c);
}

int spark_slice_evals(vpin_ctx* c, const fq* table, size_t len, int nslices, const fq* eq, size_t r0, size_t step) {
  if (!c || !table || !eq || nslices < 1 || nslices > kSparkMaxInst || step == 0 || r0 >= step || len % step) return VPIN_EINVAL;
  int rc = spark_pinned(c);
  if (rc) return rc;
  (void)hipSetDevice(c->device);
  fq* partials = nullptr;
  if ((rc = round_partials(c, &partials))) return rc;
  // a streaming read: enough workgroups to keep HBM busy whatever the number of slice groups (the round kernels' cap of
  // 64 per circuit is sized for their finisher, not for this)
  // (2048 against 512 workgroups per group: 9.9 -> 9.3 ms over the 2^25 instance's three launches; the kernel is bound by its
  // one product mod q per 64 bytes, not by HBM)
  const int grid = (int)std::min<size_t>(2048, std::max<size_t>(1, (len / step + kBlock * 8 - 1) / (kBlock * 8)));
  const int groups = (nslices + 2) / 3;
  {
    ProfScope ps(c, VPIN_K_SPARK_BUILD, ((double)nslices * 32.0 * (double)len + 32.0 * (double)len) / (double)step);
    hipLaunchKernelGGL(slice_dot3_kernel, dim3(grid, groups), dim3(kBlock), 0, c->stream, table, len, nslices, eq, partials, r0, step);
  }
  hipLaunchKernelGGL(inst_finish_kernel, dim3(groups), dim3(kBlock), 0, c->stream, (const fq*)partials, grid, 0, c->h_spark);
  VPIN_HIP_TRY(hipGetLastError());
  if ((rc = spark_wait(c))) return rc;
  // group g's three sums sit at h_spark[3g + k] = slice 3g + k: spread them to the h_spark[3 * slice] layout of the callers
  fq tmp[kSparkMaxInst];
  for (int i = 0; i < nslices; i++) tmp[i] = c->h_spark[i];
  for (int i = 0; i < nslices; i++) c->h_spark[3 * i] = tmp[i];
  return VPIN_OK;
}


// ---- persistent tail: all the small rounds of a layer in ONE launch ---------------------------------------------
// Once a layer's tables are down to kTailPairs pairs per circuit every further round is latency, not bandwidth:
// the classic path pays a launch, a finisher and a pinned-flag round trip per round (25-35 us measured).  The tail
// kernel stays resident for the rest of the layer instead: one workgroup per circuit (plus one per dot-product
// half on layer 0) folds and evaluates its own tables, publishes its three scalars and a sequence word to pinned
// host memory, then polls a pinned host mailbox for the round's challenge.  The transcript stays on the host:
// Keccak-f[1600] costs 5.8 us on 25 lanes of a wave and 8.8-11 us on one (two permutations per round), the
// mailbox round trip 2.9 us (tools/ubench_fs.hip, profiles/r02_ubench_fs.txt).  No workgroup depends on another one
// (the circuits only share the challenge), so there is no grid barrier and nothing to deadlock on; every poll loop is
// bounded and watches an abort word, so the grid always drains.
struct TailArgs {
  fq* forest; size_t stride, off, h; int ncirc;  // product circuits: blockIdx.x < ncirc
  const fq* pyr; int k;                           // suffix pyramid of the layer's k rounds
  int j0; size_t len0;                            // first round of the tail, live length before it
  fq r_prev;                                      // r_{j0-1} (j0 > 0)
  const fq* derefs; const fq* vals; fq* scratch; size_t N;  // dot-product halves: blockIdx.x - ncirc = index into kmap
  CircIds kmap;                                             // the half (matrix, half) each of those workgroups proves
  // Mailbox in pinned host memory.  Every 32-byte scalar travels as three 16-byte pieces {seq, w, w, w}: a 16-byte store
  // (GPU -> host) or load (host -> GPU) is one bus transaction, so a piece that carries the expected sequence number is
  // whole and current -- no fence, no flag, no second round trip.
  uint32_t* up;            // [inst][kTailUpChunks][4]: sums 0..2, then the 6 final entries
  const uint32_t* down;    // 3 pieces: the challenge
  const uint32_t* abort_flag;
  uint32_t seq0;           // round j publishes / waits for seq0 + (j - j0) + 1
  unsigned long long* trace;  // VPIN_TAIL_TRACE: pinned, per round {start, published, reply seen, -} in 100 MHz ticks (instance 0)
  uint32_t poll_sleep;        // VPIN_TAIL_SLEEP=1: s_sleep between two polls of the mailbox (A/B of what the polling wave costs its CU)
  // Round 6: the rounds between kTailPairs1 and kTailMaxWgs x kTailPairs1 pairs on `wgs` workgroups per circuit (blockIdx.x =
  // circuit x wgs + g).  Workgroup g owns the pairs whose index / 4 is congruent to g mod wgs -- a fold maps (i, i + pairs,
  // i + 2 pairs, i + 3 pairs) to (i, i + pairs), all of one class while 4 x wgs divides pairs, so no workgroup ever reads what
  // another one wrote (128-byte lines are not shared either).  A round's sums: every workgroup stores its partial pair to `red`,
  // the one that arrives last at the circuit's counter adds them and publishes -- nobody waits for anybody on the device.  When a
  // round is down to kTailPairs1 pairs, workgroup 0 goes on alone (the host's reply, which needs every partial, is the barrier).
  int wgs;
  fq* red;               // [inst][wgs][2]
  uint32_t* cnt;         // [inst]: arrivals of the current round, 0 between rounds (atomicInc wraps it)
};

constexpr int kTailBlock = 512;
constexpr int kTailUpChunks = 32;             // 16-byte pieces per instance: 9 for the sums, 18 for the final entries
constexpr long kTailSpinLimit = 4000000;      // ~10 s of polling: a lost host ends the kernel instead of hanging the GPU
constexpr size_t kTailQuadPairs = kTailBlock / 4;  // rounds with at most this many pairs run four lanes per pair
constexpr size_t kTailPairs1 = 1024;          // rounds with at most this many pairs per circuit: one workgroup per circuit
constexpr int kTailMaxWgs = 8;                // larger rounds (up to 8192 pairs): up to this many workgroups per circuit

__device__ __forceinline__ fq fq_shfl_from(const fq& a, int src) {
  fq r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl(a.v[i], src, 64);
  return r;
}

// sums of e[0..2] over the first `nw` waves of the block; valid in threads 0..2
__device__ __forceinline__ fq tail_block_sum3(fq* e, int nw) {
  __shared__ fq sh[kTailBlock / 64][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < nw) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      fq t = e[k];
      for (int off = 32; off >= 1; off >>= 1) t = fq_add(t, fq_shfl_xor(t, off));
      if (lane == 0) sh[wave][k] = t;
    }
  }
  __syncthreads();
  fq t = fq_zero();
  if (threadIdx.x < 3) {
    t = sh[0][threadIdx.x];
    for (int w = 1; w < nw; w++) t = fq_add(t, sh[w][threadIdx.x]);
  }
  __syncthreads();
  return t;
}

// MULTI = false: the kernel as it was (one workgroup per circuit; 161 VGPRs) -- the default.  MULTI = true: launched only when
// VPIN_SPARK_TAIL_WGS asks for several workgroups per circuit (199 VGPRs; measured slower, profiles/r06_ab_tail_wgs.txt).
template <bool MULTI>
__global__ __launch_bounds__(kTailBlock) void spark_tail_kernel(TailArgs a) {
  const int W = MULTI ? a.wgs : 1;
  const int inst = (int)blockIdx.x / W, g = (int)blockIdx.x % W;
  const bool is_dotp = inst >= a.ncirc;
  __shared__ fq sh_r;
  __shared__ int sh_stop;
  __shared__ int sh_last;
  fq r = a.r_prev;
  size_t len = a.len0;
  fq* A = nullptr;
  fq* Bt = nullptr;
  const fq* src[3] = {nullptr, nullptr, nullptr};
  fq* dst[3] = {nullptr, nullptr, nullptr};
  if (!is_dotp) {
    A = a.forest + (size_t)inst * a.stride + a.off;
    Bt = A + a.h;
  } else {
    const int kl = inst - a.ncirc, kk = a.kmap.v[kl], m = kk >> 1, half = kk & 1;
    const size_t hN = a.N / 2, q4 = a.N / 4;
    src[0] = a.derefs + (size_t)m * a.N + (size_t)half * hN;
    src[1] = a.derefs + (size_t)(3 + m) * a.N + (size_t)half * hN;
    src[2] = a.vals + (size_t)m * a.N + (size_t)half * hN;
#pragma unroll
    for (int t = 0; t < 3; t++) dst[t] = a.scratch + (size_t)(3 * kl + t) * q4;  // scratch is numbered by the LOCAL half
  }
  uint32_t* up = a.up + (size_t)inst * kTailUpChunks * 4;
  for (int j = a.j0; j < a.k; j++) {
    const bool bind = j > 0, last = j == a.k - 1;
    const size_t pairs = bind ? len / 4 : len / 2;
    const uint32_t seq = a.seq0 + (uint32_t)(j - a.j0) + 1u;
    const fq* E = a.pyr + ((((size_t)1) << a.k) - (((size_t)2) << (a.k - (j + 1))));  // pyramid level j + 1
    const bool multi = W > 1 && pairs > kTailPairs1;   // this round runs on all W workgroups of the circuit
    if (W > 1 && !multi && g != 0) return;             // from here on workgroup 0 alone (the others' folds are visible: see below)
    if (a.trace && inst == 0 && g == 0 && threadIdx.x == 0) a.trace[4 * (j - a.j0)] = wall_clock64();
    if (multi) {
      // product circuits only (the launcher keeps layers with dot-product halves on one workgroup), leading-coefficient form
      Acc<4> acc;
      acc.init();
      const size_t ngroups = pairs >> 2;  // runs of four pairs = one 128-byte line per table half
      for (size_t q = (size_t)g + (size_t)W * (threadIdx.x >> 2); q < ngroups; q += (size_t)W * (kTailBlock >> 2)) {
        const size_t i = 4 * q + (threadIdx.x & 3);
        fq p1, d1, p2, d2;
        if (bind) { fold_pd(A, i, pairs, r, p1, d1); fold_pd(Bt, i, pairs, r, p2, d2); }
        else { load_pd(A, i, pairs, p1, d1); load_pd(Bt, i, pairs, p2, d2); }
        acc.lead_bc(p1, d1, p2, d2, fq_load(E + i));
      }
      if (bind) len /= 2;
      const size_t mine = pairs / (size_t)W;  // pairs of this workgroup: its first `mine` threads carry them
      const int nw = mine >= (size_t)kTailBlock ? kTailBlock / 64 : (int)((mine + 63) / 64);
      const fq t = tail_block_sum3(acc.e, nw);
      if (threadIdx.x < 2) fq_store(a.red + ((size_t)inst * W + g) * 2 + threadIdx.x, t);
      __threadfence();  // this workgroup's folds and its partial sums: visible device-wide before it is counted
      __syncthreads();
      if (threadIdx.x == 0) {
        sh_last = atomicInc(a.cnt + inst, (unsigned)(W - 1)) == (unsigned)(W - 1);  // wraps to 0 with the last arrival: self-resetting
      }
      __syncthreads();
      if (sh_last) {  // everybody else of this circuit has stored and fenced: add up and publish
        __threadfence();
        if (threadIdx.x < 2) {
          fq sum = fq_load(a.red + ((size_t)inst * W) * 2 + threadIdx.x);
          for (int w = 1; w < W; w++) sum = fq_add(sum, fq_load(a.red + ((size_t)inst * W + w) * 2 + threadIdx.x));
          publish_scalar(up + 12 * threadIdx.x, sum, seq);
        }
      }
    } else {
    Acc<4> acc;
    acc.init();
    int nw = kTailBlock / 64, lo_off = 1;
    fq mine = fq_zero();  // quad mode: this lane's folded entry (A[i], A[q+i], B[i], B[q+i] by role)
    if (!is_dotp && pairs <= kTailQuadPairs) {
      // Four lanes per pair: each folds ONE of the pair's four entries (a lone wave issues one modular product in
      // ~0.7 us whatever its lanes do, so eight products per lane are eight times that), then lane 0 of the quad
      // forms E*A_0*B_0 and lane 1 E*dA*dB.
      const int role = threadIdx.x & 3;
      const size_t pi = threadIdx.x >> 2;
      if (pi < pairs) {
        fq* T = (role & 2) ? Bt : A;
        const size_t idx = (role & 1) ? pairs + pi : pi;
        if (bind) {
          const fq x0 = fq_load(T + idx), x1 = fq_load(T + 2 * pairs + idx);
          mine = fq_add(x0, fq_mul(r, fq_sub(x1, x0)));
          fq_store(T + idx, mine);
        } else {
          mine = fq_load(T + idx);
        }
      }
      const int lbase = (threadIdx.x & 63) & ~3;  // lane of the quad's role 0
      const fq pA = fq_shfl_from(mine, lbase), hA = fq_shfl_from(mine, lbase + 1), pB = fq_shfl_from(mine, lbase + 2),
               hB = fq_shfl_from(mine, lbase + 3);
      if (pi < pairs && role < 2) {
        const fq u = role ? fq_mul(fq_sub(hA, pA), fq_sub(hB, pB)) : fq_mul(pA, pB);
        acc.e[role] = fq_mul(fq_load(E + pi), u);
      }
      nw = (int)((4 * pairs + 63) / 64);
      lo_off = 4;  // quad layout: e[0] lives in the role-0 lanes, e[1] in the role-1 lanes (reduced over strides 32..4 below)
    } else if (!is_dotp) {
      for (size_t i = threadIdx.x; i < pairs; i += kTailBlock) {
        fq p1, d1, p2, d2;
        if (bind) { fold_pd(A, i, pairs, r, p1, d1); fold_pd(Bt, i, pairs, r, p2, d2); }
        else { load_pd(A, i, pairs, p1, d1); load_pd(Bt, i, pairs, p2, d2); }
        acc.lead_bc(p1, d1, p2, d2, fq_load(E + i));
      }
      if (pairs < (size_t)kTailBlock) nw = (int)((pairs + 63) / 64);
    } else {
      // round 0 reads the committed polynomials as they are, round 1 folds them into scratch, later rounds fold scratch in place
      const bool from_src = j <= 1;
      for (size_t i = threadIdx.x; i < pairs; i += kTailBlock) {
        fq u[3], p1, d1, p2, d2;
        if (bind) {
          fold_pd2(from_src ? src[0] : dst[0], dst[0], i, pairs, r, p1, d1);
          fold_pd2(from_src ? src[1] : dst[1], dst[1], i, pairs, r, p2, d2);
        } else { load_pd(src[0], i, pairs, p1, d1); load_pd(src[1], i, pairs, p2, d2); }
        acc.stage_bc(u, p1, d1, p2, d2);
        if (bind) fold_pd2(from_src ? src[2] : dst[2], dst[2], i, pairs, r, p1, d1);
        else load_pd(src[2], i, pairs, p1, d1);
        acc.stage_a(u, p1, d1);
      }
      if (pairs < (size_t)kTailBlock) nw = (int)((pairs + 63) / 64);
    }
    if (bind) len /= 2;
    fq t;
    if (lo_off == 4) {
      // quad layout: every lane reduces the sum of its own role over the lanes of that role (strides 32..4)
      fq v = (threadIdx.x & 1) ? acc.e[1] : acc.e[0];
      __shared__ fq shq[kTailBlock / 64][2];
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      if (wave < nw) {
        for (int off = 32; off >= 4; off >>= 1) v = fq_add(v, fq_shfl_xor(v, off));
        if (lane < 2) shq[wave][lane] = v;
      }
      __syncthreads();  // also orders this round's folds before the next round's loads
      t = fq_zero();
      if (threadIdx.x < 2) {
        t = shq[0][threadIdx.x];
        for (int w = 1; w < nw; w++) t = fq_add(t, shq[w][threadIdx.x]);
      }
    } else {
      t = tail_block_sum3(acc.e, nw);  // its barriers also order this round's folds before the next round's loads
    }
    if ((int)threadIdx.x < (is_dotp ? 3 : 2)) publish_scalar(up + 12 * threadIdx.x, t, seq);  // a product circuit has two sums
    if (last) {
      // live length is 2: both entries of every table, for the host to bind with the last challenge
      if (!is_dotp && lo_off == 4) {
        if (threadIdx.x < 4) publish_scalar(up + 36 + 12 * threadIdx.x, mine, seq);  // A[0], A[1], B[0], B[1]
      } else {
        const int ne = is_dotp ? 6 : 4;
        if ((int)threadIdx.x < ne) {
          const int tt = threadIdx.x >> 1, e = threadIdx.x & 1;
          const fq* tab;
          if (!is_dotp) tab = tt ? Bt : A;
          else tab = (a.k >= 2) ? dst[tt] : src[tt];  // no fold has happened in a one-round layer
          publish_scalar(up + 36 + 12 * threadIdx.x, fq_load(tab + e), seq);
        }
      }
    }
    }  // (single-workgroup round)
    if (a.trace && inst == 0 && g == 0 && threadIdx.x == 0) a.trace[4 * (j - a.j0) + 1] = wall_clock64();
    if (last) break;
    if (threadIdx.x == 0) {
      int stop = 1;
      for (long spin = 0; spin < kTailSpinLimit; spin++) {
        // one 16-byte read per poll (12-18 workgroups poll the same three pieces: a third of the PCIe reads); the other two
        // pieces only once the first carries the round's sequence number
        u32x4 c0 = load16_system(a.down), c1, c2;
        if (c0.x != seq) {
          if (a.poll_sleep || g != 0) __builtin_amdgcn_s_sleep(16);  // ~1024 cycles off the SIMD's arbiter; the extra workgroups of a circuit always
          if ((spin & 15) == 15 && __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
          continue;
        }
        load48_system(a.down, c0, c1, c2);
        fq rr;
        if (take_reply(c0, c1, c2, seq, rr)) {  // whole (checksum) and current
          sh_r = rr;
          stop = 0;
          break;
        }
        if (__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
      }
      sh_stop = stop;
      if (a.trace && inst == 0 && g == 0) a.trace[4 * (j - a.j0) + 2] = wall_clock64();
    }
    __syncthreads();
    if (sh_stop) return;
    r = sh_r;
    // The reply exists only because every workgroup of every circuit had published (fenced) its round: an acquire here and
    // the next round may read any class's folded entries -- workgroup 0 does when it goes on alone
    if (multi) __threadfence();
  }
}

// pinned mailbox carved out of ctx->h_spark (kSparkPinned fq = 256 KiB), offsets in fq elements (32 B)
constexpr size_t kTailUp = 4096;                     // kSparkMaxInst x kTailUpChunks x 16 B = 18 x 512 B = 288 fq
constexpr size_t kTailWords = 4096 + 320;            // uint32 words: abort at [0], the reply's three pieces at [16..28)
constexpr size_t kTailTrace = 4096 + 384;            // 64 rounds x 4 x u64
static inline uint32_t* tail_words(vpin_ctx* c) { return reinterpret_cast<uint32_t*>(c->h_spark + kTailWords); }
static inline uint32_t* tail_up(vpin_ctx* c) { return reinterpret_cast<uint32_t*>(c->h_spark + kTailUp); }

size_t spark_tail_pairs() {
  static const size_t n = [] { const char* e = getenv("VPIN_SPARK_TAIL_PAIRS"); long v = e ? atol(e) : 1024; return (size_t)(v < 0 ? 0 : v); }();
  return n;  // 0 disables the persistent tail (classic per-round launches only)
}

// Resident workgroups of persistent tails per device, all contexts of the process: a tail's workgroups wait for the host, and the
// host for ALL of them, so a launch whose workgroups do not all fit beside the other resident ones would sit out its poll bound.
// The tail kernel takes a CU per workgroup (161 VGPRs x 8 waves).  One workgroup per circuit (<= 18 per context) always fit; the
// multi-workgroup rounds reserve their grid against HALF the CUs the context's stream may use (other processes on the device,
// transient kernels) or run classically for another round (spark_tail_launch returns 1).
static std::atomic<int> g_tail_resident[16];
static int tail_max_wgs() {
  // default 1: measured in round 6, 2 / 4 / 8 workgroups per circuit do not beat a launch per round (a resident workgroup needs
  // ~30 us for 1024 pairs on its one CU, the chip-wide launch ~25 us for 8192) -- profiles/r06_ab_tail_wgs.txt
  const char* e = getenv("VPIN_SPARK_TAIL_WGS");  // read per call: tests and A/B runs switch it inside one process
  const int v = e ? atoi(e) : 1;
  return v < 1 ? 1 : v > kTailMaxWgs ? kTailMaxWgs : v;
}
// pairs per circuit from which the host may ask for the tail (layers without dot-product halves: the multi-workgroup rounds)
size_t spark_tail_first_pairs(bool with_dotp) {
  const size_t one = spark_tail_pairs();
  if (with_dotp || one != kTailPairs1) return one;   // (a non-default VPIN_SPARK_TAIL_PAIRS: experiments keep one workgroup)
  return one * (size_t)tail_max_wgs();
}
static void tail_release(vpin_ctx* c) {
  if (c->tail_reserved) { g_tail_resident[c->device & 15].fetch_sub(c->tail_reserved, std::memory_order_acq_rel); c->tail_reserved = 0; }
}

int spark_tail_launch(vpin_ctx* c, const SparkForest* f, int level, int k, int j0, size_t len0, const fq* pyr, const uint8_t* r_prev,
                      size_t N, const fq* vals, const fq* comb_derefs, fq* scratch, const int* halves, int ndotp) {
  if (!c || !f || !f->base || !pyr || k < 1 || j0 < 0 || j0 >= k || (j0 > 0 && !r_prev) || f->ncirc > 12 || f->ncirc < 1) return VPIN_EINVAL;
  const bool with_dotp = vals != nullptr && ndotp > 0;
  if (with_dotp && (!comb_derefs || !scratch || ndotp > 6 || level != 0 || N != f->n)) return VPIN_EINVAL;
  const size_t h = f->n >> (level + 1);
  if (h != ((size_t)1 << k) || len0 != (j0 == 0 ? h : (h >> (j0 - 1)))) return VPIN_ESHAPE;
  int rc = spark_pinned(c);
  if (rc) return rc;
  (void)hipSetDevice(c->device);
  TailArgs a{};
  a.forest = f->base; a.stride = f->stride(); a.off = f->level_off(level); a.h = h; a.ncirc = f->ncirc;
  a.pyr = pyr; a.k = k; a.j0 = j0; a.len0 = len0;
  if (r_prev) a.r_prev = load_host_fq(r_prev);
  a.derefs = comb_derefs; a.vals = vals; a.scratch = scratch; a.N = N;
  for (int i = 0; i < 6; i++) a.kmap.v[i] = (with_dotp && halves && i < ndotp) ? halves[i] : i;
  a.up = tail_up(c);
  uint32_t* w = tail_words(c);
  a.abort_flag = w; a.down = w + 16;
  w[0] = 0;
  a.seq0 = c->tail_seq;
  static const uint32_t poll_sleep = getenv("VPIN_TAIL_SLEEP") ? 1u : 0u;
  a.poll_sleep = poll_sleep;
  static const bool trace_on = getenv("VPIN_TAIL_TRACE") != nullptr;
  a.trace = trace_on ? reinterpret_cast<unsigned long long*>(c->h_spark + kTailTrace) : nullptr;
  if (trace_on) memset(c->h_spark + kTailTrace, 0, 64 * 32);
  const int ninst = f->ncirc + (with_dotp ? ndotp : 0);
  // workgroups per circuit: the first round's pairs / 1024 (2, 4 or 8) when the host starts the tail early
  const size_t pairs0 = j0 == 0 ? len0 / 2 : len0 / 4;
  int wgs = 1;
  if (pairs0 > kTailPairs1) {
    if (with_dotp || pairs0 % kTailPairs1 || pairs0 / kTailPairs1 > (size_t)kTailMaxWgs) return VPIN_ESHAPE;
    wgs = (int)(pairs0 / kTailPairs1);
  }
  const int want = ninst * wgs;
  {
    std::atomic<int>& res = g_tail_resident[c->device & 15];
    const int cap = wgs > 1 ? c->num_cus / 2 : (1 << 30);
    int cur = res.load(std::memory_order_acquire);
    for (;;) {
      if (cur + want > cap) return 1;   // declined: the caller proves this round with a launch and may ask again
      if (res.compare_exchange_weak(cur, cur + want, std::memory_order_acq_rel)) break;
    }
    c->tail_reserved = want;
  }
  a.wgs = wgs; a.red = c->d_tail_red; a.cnt = c->d_tail_cnt;
  c->tail_rounds = k - j0;
  {
    ProfScope ps(c, VPIN_K_SPARK_TAIL, 0.0);
    if (wgs > 1) hipLaunchKernelGGL(spark_tail_kernel<true>, dim3((unsigned)want), dim3(kTailBlock), 0, c->stream, a);
    else hipLaunchKernelGGL(spark_tail_kernel<false>, dim3((unsigned)want), dim3(kTailBlock), 0, c->stream, a);
  }
  if (hipGetLastError() != hipSuccess) { tail_release(c); c->tail_rounds = 0; return VPIN_EHIP; }
  return VPIN_OK;
}

// results of tail round `idx` (0-based within the tail) of all `ninst` instances: sums -> spark_tail_sums()[3*inst + x] and,
// after the tail's last round, the live entries -> spark_tail_final()[6*inst + e]
int spark_tail_wait(vpin_ctx* c, int idx, int ninst, int ncirc) {
  const uint32_t want = c->tail_seq + (uint32_t)idx + 1u;
  const bool last = idx == c->tail_rounds - 1;
  const uint32_t* up = tail_up(c);
  const auto t0 = std::chrono::steady_clock::now();
  unsigned done_mask = 0;
  const unsigned all = (1u << ninst) - 1u;
  double next_nudge = 1.0;
  for (long spins = 0;; spins++) {
    for (int i = 0; i < ninst; i++) {
      if (done_mask >> i & 1u) continue;
      const uint32_t* u = up + (size_t)i * kTailUpChunks * 4;
      bool ok = true;
      const int nsum = i < ncirc ? 2 : 3;  // leading-coefficient form: two sums per product circuit, three per dot-product half
      for (int k = 0; k < nsum && ok; k++) ok = tail_take(u + 12 * k, want, &c->tail_sums[3 * i + k]);
      if (nsum == 2) c->tail_sums[3 * i + 2] = fq_zero_host();
      const int ne = i < ncirc ? 4 : 6;
      for (int e = 0; last && e < ne && ok; e++) ok = tail_take(u + 36 + 12 * e, want, &c->tail_final[6 * i + e]);
      if (ok) done_mask |= 1u << i;
    }
    if (done_mask == all) return VPIN_OK;
    __builtin_ia32_pause();
    if ((spins & 0xfff) == 0xfff) {
      // Round 6: no HIP call in the fast path of this loop.  hipStreamQuery used to be called every 4096 spins (to nudge runtimes /
      // profilers that submit lazily); a HIP entry point can block behind another host thread's device-wide wait (hipFree, a
      // NULL-stream copy) which in turn waits for THIS context's resident kernel -- the kernel then polls out its bound for a reply
      // this thread cannot send.  Now: once per second, and only after a second without an answer.
      const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      static const double limit_s = [] { const char* e = getenv("VPIN_TAIL_TIMEOUT_S"); double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 20.0; }();
      if (waited > next_nudge) {
        const double tq = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const hipError_t q = hipStreamQuery(c->stream);
        const double tq1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "[tail] ctx %p waits %.2f s for round %d of %d (instances done %#x of %#x, seq %u, stream %s, query took %.3f s, "
                        "masked %d)\n", (void*)c, waited, idx, c->tail_rounds.load(), done_mask, all, want,
                q == hipSuccess ? "IDLE: the kernel has left" : q == hipErrorNotReady ? "busy" : hipGetErrorName(q), tq1 - tq, (int)c->cu_masked);
        next_nudge += 1.0;
      }
      if (waited > limit_s) {
        for (int i = 0; i < ninst; i++)
          if (!(done_mask >> i & 1u)) {
            const uint32_t* u = up + (size_t)i * kTailUpChunks * 4;
            fprintf(stderr, "[tail]   instance %d: sequence words of its first sum %u %u %u (want %u); reply slot holds %u\n", i, u[0], u[4], u[8],
                    want, tail_words(c)[16]);
          }
        tail_words(c)[0] = 1;  // abort: every workgroup leaves its poll loop
        (void)hipStreamSynchronize(c->stream);
        // retire this launch's sequence numbers: the next proof on this context must not accept the pieces the aborted
        // kernel has already published, nor the kernel of the next proof the stale challenge
        c->tail_seq += (uint32_t)c->tail_rounds + 1u;
        c->tail_rounds = 0;
        tail_release(c);
        // an aborted multi-workgroup round leaves the arrival counters anywhere: start them over
        (void)hipMemsetAsync(c->d_tail_cnt, 0, kSparkMaxInst * sizeof(uint32_t), c->stream);
        (void)hipStreamSynchronize(c->stream);
        set_last_error("spark_tail_wait: the persistent round kernel did not answer", hipErrorUnknown);
        return VPIN_EHIP;
      }
    }
  }
}

// challenge of tail round `idx` to the kernel (mailbox_dev.h host_reply: three 16-byte pieces, sequence number + checksum)
void spark_tail_reply(vpin_ctx* c, int idx, const uint8_t r[32]) {
  host_reply(tail_words(c) + 16, c->tail_seq + (uint32_t)idx + 1u, r);
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
}

// an error path leaves a tail resident: make every workgroup leave its poll loop, wait for the grid to drain and retire the
// launch's sequence numbers (the next proof must accept neither its published pieces nor its challenge)
void spark_tail_abort(vpin_ctx* c) {
  if (!c || c->tail_rounds == 0) return;
  tail_words(c)[0] = 1;
  (void)hipStreamSynchronize(c->stream);
  c->tail_seq += (uint32_t)c->tail_rounds + 1u;
  c->tail_rounds = 0;
  tail_release(c);
  (void)hipMemsetAsync(c->d_tail_cnt, 0, kSparkMaxInst * sizeof(uint32_t), c->stream);
  (void)hipStreamSynchronize(c->stream);
}

// after the last round's results were taken: retire the sequence numbers of this tail
void spark_tail_end(vpin_ctx* c) {
  static const bool trace_on = getenv("VPIN_TAIL_TRACE") != nullptr;
  if (trace_on) {
    (void)hipStreamSynchronize(c->stream);
    const unsigned long long* t = reinterpret_cast<const unsigned long long*>(c->h_spark + kTailTrace);
    fprintf(stderr, "[tail] %d rounds:", c->tail_rounds.load());
    for (int i = 0; i < c->tail_rounds && i < 64; i++) {
      const double comp = (double)(t[4 * i + 1] - t[4 * i]) * 0.01, wait = t[4 * i + 2] ? (double)(t[4 * i + 2] - t[4 * i + 1]) * 0.01 : 0.0;
      fprintf(stderr, " %.1f+%.1f", comp, wait);
    }
    fprintf(stderr, " us (compute+publish, wait for reply)\n");
  }
  c->tail_seq += (uint32_t)c->tail_rounds + 1u;
  c->tail_rounds = 0;
  tail_release(c);
}

const fq* spark_tail_sums(vpin_ctx* c) { return c->tail_sums; }
const fq* spark_tail_final(vpin_ctx* c) { return c->tail_final; }

}  // namespace vpin

// Host-only self-test of the mailbox framing (mailbox_dev.h): a whole publication is accepted; a torn piece (new sequence
// word over stale payload), pieces of two publications and a stale sequence number are all rejected.  0 = as expected.
extern "C" int vpin_host_mailbox_selftest(void) {
  using namespace vpin;
  alignas(16) uint32_t slot[12];
  uint8_t r[32];
  for (int i = 0; i < 32; i++) r[i] = (uint8_t)(17 * i + 3);
  host_reply(slot, 7u, r);
  fq got;
  if (!tail_take(slot, 7u, &got) || memcmp(got.v, r, 32) != 0) return 1;   // whole and current
  if (tail_take(slot, 8u, &got)) return 2;                                   // stale sequence number
  alignas(16) uint32_t torn[12];
  memcpy(torn, slot, sizeof slot);
  torn[2] ^= 0x10u;                                                          // 8 + 8 tear: second half of piece 0 from an older value
  if (tail_take(torn, 7u, &got)) return 3;
  uint8_t r2[32];
  for (int i = 0; i < 32; i++) r2[i] = (uint8_t)(29 * i + 1);
  alignas(16) uint32_t other[12];
  host_reply(other, 7u, r2);                                                 // same sequence number, another scalar (never happens; the
  memcpy(torn, slot, sizeof slot);                                           // checksum still tells the pieces apart)
  memcpy(torn + 4, other + 4, 16);
  if (tail_take(torn, 7u, &got)) return 4;
  return 0;
}

// Kernel-level C-ABI entry (include/vpin_hip.h): one round of prove_cubic_batched for the product circuits of one forest
// level and, optionally, the six dot-product circuit halves -- exactly the launch group spark.cpp issues per round.
extern "C" int vpin_spark_batched_round(vpin_ctx* c, vpin_table* forest, size_t n, int ncirc, int level, size_t len,
                                        const vpin_table* E, size_t e_off, const uint8_t* r, int lead, const vpin_table* derefs,
                                        const vpin_table* vals, vpin_table* scratch, int first_fold, uint8_t* out) {
  using namespace vpin;
  if (!c || !forest || !E || !out || ncirc < 1 || ncirc > 12 || !is_pow2(n) || n < 4 || level < 0) return VPIN_EINVAL;
  const bool with_dotp = derefs != nullptr;
  if (with_dotp && (!vals || !scratch || ncirc != 12 || level != 0)) return VPIN_EINVAL;
  if (forest->len < (size_t)ncirc * 2 * n) return VPIN_ESHAPE;
  const size_t pairs = r ? len / 4 : len / 2;
  if (e_off + pairs > E->len) return VPIN_ESHAPE;
  if (with_dotp && (derefs->len < 6 * n || vals->len < 3 * n || scratch->len < 18 * (n / 4))) return VPIN_ESHAPE;
  SparkForest f;
  f.base = forest->d; f.n = n; f.ncirc = ncirc;
  int rc = spark_prod_round(c, &f, level, len, E->d + e_off, r, with_dotp ? 6 : 0, lead != 0);
  if (!rc && with_dotp) rc = spark_dotp_round(c, n, vals->d, derefs->d, scratch->d, len, first_fold != 0, r);
  if (!rc) rc = spark_wait_flag(c);
  if (rc) return rc;
  memcpy(out, c->h_spark, (size_t)(ncirc + (with_dotp ? 6 : 0)) * 96);
  return VPIN_OK;
}

