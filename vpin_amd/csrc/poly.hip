// poly.hip -- dense-polynomial helpers of the evaluation proof, gfx950.
//   DensePolynomial::bound   Spartan/src/dense_mlpoly.rs:220-227   LZ[i] = sum_j L[j]*Z[j*R+i]
// Z is read exactly once, row-major, so every wavefront streams 2 KiB contiguous per row.
#include <cstring>
#include <vector>

#include "comm.h"
#include "ctx.h"

namespace vpin {

constexpr int kPB = 256;

// partial[chunk][i] = sum_{j in chunk} L[j] * Z[j*R + i]
__global__ __launch_bounds__(kPB) void poly_bound_kernel(const fq* __restrict__ Z, const fq* __restrict__ Lv, size_t Ls,
                                                         size_t Rs, size_t rows_per_chunk, fq* __restrict__ partial) {
  size_t i = (size_t)blockIdx.x * kPB + threadIdx.x;
  if (i >= Rs) return;
  size_t j0 = (size_t)blockIdx.y * rows_per_chunk, j1 = j0 + rows_per_chunk;
  if (j1 > Ls) j1 = Ls;
  fq acc = fq_zero();
  for (size_t j = j0; j < j1; j++) {
    fq z = fq_load(Z + j * Rs + i);
    if (fq_is_zero(z)) continue;
    acc = fq_add(acc, fq_mul(fq_load(Lv + j), z));
  }
  fq_store(partial + (size_t)blockIdx.y * Rs + i, acc);
}

__global__ __launch_bounds__(kPB) void poly_bound_reduce_kernel(const fq* __restrict__ partial, size_t Rs, int chunks,
                                                                fq* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * kPB + threadIdx.x;
  if (i >= Rs) return;
  fq acc = fq_load(partial + i);
  for (int k = 1; k < chunks; k++) acc = fq_add(acc, fq_load(partial + (size_t)k * Rs + i));
  fq_store(out + i, acc);
}

// ---- the hash layer's slice evaluations and its DensePolynomial::bound in ONE pass over the table ----------------------
// HashLayerProof::prove (sparse_mlpoly.rs:740-849) evaluates every slice Z_s (length N) of a combined polynomial at the
// same point r (DensePolynomial::evaluate), draws the combining challenges c from the transcript and then proves the
// combined polynomial's evaluation at (c, r): PolyEvalProof::prove's first step is LZ = L^T Z with L = eq((c, r)[..left])
// (dense_mlpoly.rs:340-349) -- a second pass over the same 8N / 16N / 2M scalars.  The slice index is the top of the row
// index, so with r = (r_top, r_bot) split where the rows end,
//     LZ_s[i] = sum_t eq(r_top, t) Z_s[t * R + i]                 (needs no challenge: computed first, one pass)
//     Z_s(r)  = sum_i eq(r_bot, i) LZ_s[i]                        (the slice evaluations)
//     LZ[i]   = sum_s eq(c, s) LZ_s[i]                            (after the challenges: S x R scalars, no table pass)
// exact field arithmetic, so the same elements as the two passes (and the big eq(r, .) tables are never built).
// partial[(chunk * S + s) * Rs + i] = sum_{t in chunk} Ltop[t] * Z[s * N + t * Rs + i], products accumulated unreduced
// (fq_wide: one Montgomery reduction per seven terms)
__global__ __launch_bounds__(kPB) void slices_bound_kernel(const fq* __restrict__ Z, size_t N, const fq* __restrict__ Ltop, size_t T,
                                                           size_t Rs, size_t rows_per_chunk, int S_total, int s0,
                                                           fq* __restrict__ partial) {
  const size_t i = (size_t)blockIdx.x * kPB + threadIdx.x;
  if (i >= Rs) return;
  const fq* zs = Z + (size_t)blockIdx.z * N;
  size_t j0 = (size_t)blockIdx.y * rows_per_chunk, j1 = j0 + rows_per_chunk;
  if (j1 > T) j1 = T;
  fq acc = fq_zero();
  fq_wide w;
  fqw_zero(w);
  int n = 0;
  for (size_t j = j0; j < j1; j++) {
    const fq z = fq_load(zs + j * Rs + i);
    if (fq_is_zero(z)) continue;
    fqw_mac(w, fq_load(Ltop + j), z);
    if (++n == 7) { acc = fq_add(acc, fqw_reduce(w)); fqw_zero(w); n = 0; }
  }
  if (n) acc = fq_add(acc, fqw_reduce(w));
  fq_store(partial + ((size_t)blockIdx.y * S_total + s0 + blockIdx.z) * Rs + i, acc);
}

// The same for slices held as u32 (addresses and timestamps of the computation decommitment; their field images are
// Scalar::from(v) = v R mod q): Ltop[t] * (v R) R^-1 = Ltop[t] * v as an integer multiple -- eight 32 x 32 multiply-adds into a
// 320-bit accumulator instead of a product mod q, and 4 bytes from HBM instead of 32.  The accumulator is brought back to
// [0, q) once per (thread, chunk): lo + hi 2^256 = fq_mul(lo, R) + fq_mul(hi, R^2) (Montgomery products: x R R^-1 = x,
// hi R^2 R^-1 = hi R), canonical, so the element is bit for bit the one the field-image pass computes.
__device__ __forceinline__ fq fq_r2_poly() {  // R^2 mod q (ristretto255.rs:309-314)
  fq r;
  r.v[0] = 0x449c0f01u; r.v[1] = 0xa40611e3u; r.v[2] = 0x68859347u; r.v[3] = 0xd00e1ba7u;
  r.v[4] = 0x17f5be65u; r.v[5] = 0xceec73d2u; r.v[6] = 0x7c309a3du; r.v[7] = 0x0399411bu;
  return r;
}
__global__ __launch_bounds__(kPB) void slices_bound_u32_kernel(const uint32_t* __restrict__ Z32, size_t N, const fq* __restrict__ Ltop,
                                                               size_t T, size_t Rs, size_t rows_per_chunk, int S_total,
                                                               fq* __restrict__ partial) {
  const size_t i = (size_t)blockIdx.x * kPB + threadIdx.x;
  if (i >= Rs) return;
  const uint32_t* zs = Z32 + (size_t)blockIdx.z * N;
  size_t j0 = (size_t)blockIdx.y * rows_per_chunk, j1 = j0 + rows_per_chunk;
  if (j1 > T) j1 = T;
  uint32_t a[10];
#pragma unroll
  for (int k = 0; k < 10; k++) a[k] = 0;
  for (size_t j = j0; j < j1; j++) {
    const uint32_t v = zs[j * Rs + i];
    if (v == 0) continue;
    const fq l = fq_load(Ltop + j);
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint64_t t = (uint64_t)l.v[k] * v + a[k] + carry;   // < 2^64: (2^32-1)^2 + 2 (2^32-1)
      a[k] = (uint32_t)t;
      carry = t >> 32;
    }
    const uint64_t t8 = (uint64_t)a[8] + carry;
    a[8] = (uint32_t)t8;
    a[9] += (uint32_t)(t8 >> 32);   // T <= 2^28 rows of values below 2^32 q: the sum stays below 2^320
  }
  fq lo, hi = fq_zero();
#pragma unroll
  for (int k = 0; k < 8; k++) lo.v[k] = a[k];
  hi.v[0] = a[8]; hi.v[1] = a[9];
  const fq acc = fq_add(fq_mul(lo, fq_one()), fq_mul(hi, fq_r2_poly()));
  fq_store(partial + ((size_t)blockIdx.y * S_total + blockIdx.z) * Rs + i, acc);
}

// ev[s] = sum_i Rv[i] * LZs[s * Rs + i]; one workgroup per slice
__global__ __launch_bounds__(kPB) void slices_eval_kernel(const fq* __restrict__ LZs, const fq* __restrict__ Rv, size_t Rs,
                                                          fq* __restrict__ ev) {
  const fq* lz = LZs + (size_t)blockIdx.x * Rs;
  fq acc = fq_zero();
  fq_wide w;
  fqw_zero(w);
  int n = 0;
  for (size_t i = threadIdx.x; i < Rs; i += kPB) {
    fqw_mac(w, fq_load(Rv + i), fq_load(lz + i));
    if (++n == 7) { acc = fq_add(acc, fqw_reduce(w)); fqw_zero(w); n = 0; }
  }
  if (n) acc = fq_add(acc, fqw_reduce(w));
  acc = fq_wave_sum(acc);
  __shared__ fq sh[kPB / 64];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    fq t = sh[0];
    for (int k = 1; k < kPB / 64; k++) t = fq_add(t, sh[k]);
    fq_store(ev + blockIdx.x, t);
  }
}

// out[i] = sum_s coef[s] * LZs[s * Rs + i]
__global__ __launch_bounds__(kPB) void slices_combine_kernel(const fq* __restrict__ LZs, const fq* __restrict__ coef, int S, size_t Rs,
                                                             fq* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * kPB + threadIdx.x;
  if (i >= Rs) return;
  fq acc = fq_zero();
  for (int s = 0; s < S; s++) acc = fq_add(acc, fq_mul(fq_load(coef + s), fq_load(LZs + (size_t)s * Rs + i)));
  fq_store(out + i, acc);
}

// Z32 / n32: the first n32 slices as u32 (nullptr / 0: none); Z: the following S - n32 slices as field elements
int slices_bound(vpin_ctx* c, const uint32_t* Z32, int n32, const fq* Z, size_t N, int S, size_t Rs, const uint8_t* Ltop, size_t T,
                 const uint8_t* Rv, fq* d_LZs, uint8_t* ev_out) {
  if (!c || S < 1 || n32 < 0 || n32 > S || (n32 && !Z32) || (n32 < S && !Z) || !Ltop || !Rv || !d_LZs || !ev_out || T == 0 || Rs == 0 ||
      T * Rs != N)
    return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  // Row chunks: only as many as it takes to fill the chip (~4096 workgroups with the column blocks and the slices): every chunk
  // writes S x Rs partial sums that the reduction reads back -- 64 chunks made that 1 GB for the 2^25 instance's ops polynomial,
  // more than the 1.6 GB of u32 slices the pass is there to read
  const size_t col_blocks = (Rs + kPB - 1) / kPB;
  size_t want_chunks = (4096 + col_blocks * (size_t)S - 1) / (col_blocks * (size_t)S);
  if (want_chunks > 64) want_chunks = 64;
  if (want_chunks > T) want_chunks = T;
  if (want_chunks < 1) want_chunks = 1;
  const size_t rows_per_chunk = (T + want_chunks - 1) / want_chunks;
  const int chunks = (int)((T + rows_per_chunk - 1) / rows_per_chunk);
  DevBuf bL(c), bR(c), bpart(c), bev(c);
  if (bL.alloc(T * 32) || bR.alloc(Rs * 32) || bpart.alloc((size_t)chunks * S * Rs * 32) || bev.alloc((size_t)S * 32)) return VPIN_ENOMEM;
  VPIN_HIP_TRY(hipMemcpyAsync(bL.p, Ltop, T * 32, hipMemcpyHostToDevice, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync(bR.p, Rv, Rs * 32, hipMemcpyHostToDevice, c->stream));
  {
    ProfScope ps(c, VPIN_K_SPARK_BUILD, (32.0 * (double)(S - n32) + 4.0 * (double)n32) * (double)N);
    const unsigned gx = (unsigned)((Rs + kPB - 1) / kPB);
    if (n32)
      hipLaunchKernelGGL(slices_bound_u32_kernel, dim3(gx, (unsigned)chunks, (unsigned)n32), dim3(kPB), 0, c->stream, Z32, N,
                         (const fq*)bL.p, T, Rs, rows_per_chunk, S, (fq*)bpart.p);
    if (n32 < S)
      hipLaunchKernelGGL(slices_bound_kernel, dim3(gx, (unsigned)chunks, (unsigned)(S - n32)), dim3(kPB), 0, c->stream, Z, N,
                         (const fq*)bL.p, T, Rs, rows_per_chunk, S, n32, (fq*)bpart.p);
  }
  hipLaunchKernelGGL(poly_bound_reduce_kernel, dim3((unsigned)(((size_t)S * Rs + kPB - 1) / kPB)), dim3(kPB), 0, c->stream,
                     (const fq*)bpart.p, (size_t)S * Rs, chunks, d_LZs);
  hipLaunchKernelGGL(slices_eval_kernel, dim3((unsigned)S), dim3(kPB), 0, c->stream, (const fq*)d_LZs, (const fq*)bR.p, Rs, (fq*)bev.p);
  VPIN_HIP_TRY(hipGetLastError());
  VPIN_HIP_TRY(hipMemcpyAsync(ev_out, bev.p, (size_t)S * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));  // Ltop / Rv / ev_out are the caller's
  return VPIN_OK;
}

int slices_combine(vpin_ctx* c, const fq* d_LZs, int S, size_t Rs, const uint8_t* coef, uint8_t* out_LZ) {
  if (!c || !d_LZs || S < 1 || !coef || !out_LZ || Rs == 0) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  DevBuf bc(c), bout(c);
  if (bc.alloc((size_t)S * 32) || bout.alloc(Rs * 32)) return VPIN_ENOMEM;
  VPIN_HIP_TRY(hipMemcpyAsync(bc.p, coef, (size_t)S * 32, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(slices_combine_kernel, dim3((unsigned)((Rs + kPB - 1) / kPB)), dim3(kPB), 0, c->stream, d_LZs, (const fq*)bc.p, S, Rs,
                     (fq*)bout.p);
  VPIN_HIP_TRY(hipGetLastError());
  VPIN_HIP_TRY(hipMemcpyAsync(out_LZ, bout.p, Rs * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

// rows [row0, row0 + nrows) of the same sum, as a device vector of Rs scalars (zeros when nrows == 0): one polynomial bound
// split across ranks by row blocks; dL = the L_size coefficients on the device
static int poly_bound_rows(vpin_ctx* c, const fq* Z, const fq* dL, size_t Rs, size_t row0, size_t nrows, fq* d_out) {
  if (nrows == 0) {
    VPIN_HIP_TRY(hipMemsetAsync(d_out, 0, Rs * 32, c->stream));
    return VPIN_OK;
  }
  size_t rows_per_chunk = nrows / 64 ? nrows / 64 : 1;
  int chunks = (int)((nrows + rows_per_chunk - 1) / rows_per_chunk);
  DevBuf bpart(c);
  if (bpart.alloc((size_t)chunks * Rs * 32)) return VPIN_ENOMEM;
  dim3 grid((unsigned)((Rs + kPB - 1) / kPB), (unsigned)chunks);
  hipLaunchKernelGGL(poly_bound_kernel, grid, dim3(kPB), 0, c->stream, Z + row0 * Rs, dL + row0, nrows, Rs, rows_per_chunk, (fq*)bpart.p);
  hipLaunchKernelGGL(poly_bound_reduce_kernel, dim3(grid.x), dim3(kPB), 0, c->stream, (const fq*)bpart.p, Rs, chunks, d_out);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;  // bpart returns to the pool; later work on this stream is ordered behind the kernels
}

// DensePolynomial::bound over the ranks of c->comm: every rank sums its block of rows, the partial vectors are all-gathered
// on the device (RCCL ncclAllGather when enabled, staged through the host transport otherwise) and added up by everyone.
// Field addition is exact, so the result does not depend on the split.
int poly_bound_dist(vpin_ctx* c, const vpin_table* Z, const uint8_t* Lvec, size_t L_size, uint8_t* out_LZ, const fq* z_rows) {
  if (!c || !c->comm || !Z || (!Z->d && !z_rows) || !Lvec || !out_LZ || L_size == 0) return VPIN_EINVAL;
  if (Z->len % L_size != 0) return VPIN_ESHAPE;
  vpin_comm* cm = c->comm;
  const size_t Rs = Z->len / L_size;
  (void)hipSetDevice(c->device);
  DevBuf bL(c), bmine(c), ball(c), bout(c);
  if (bL.alloc(L_size * 32) || bmine.alloc(Rs * 32) || ball.alloc((size_t)cm->world * Rs * 32) || bout.alloc(Rs * 32)) return VPIN_ENOMEM;
  int rc;
  if (z_rows) {
    // this rank's rows rank, rank + world, .. stored densely: their coefficients in the same order
    const size_t nloc = comm_strided_count(L_size, cm->rank, cm->world);
    std::vector<uint8_t> Lloc((nloc ? nloc : 1) * 32);
    for (size_t k = 0; k < nloc; k++) memcpy(Lloc.data() + 32 * k, Lvec + 32 * ((size_t)cm->rank + k * (size_t)cm->world), 32);
    VPIN_HIP_TRY(hipMemcpyAsync(bL.p, Lloc.data(), (nloc ? nloc : 1) * 32, hipMemcpyHostToDevice, c->stream));
    VPIN_HIP_TRY(hipStreamSynchronize(c->stream));  // Lloc is a local
    rc = poly_bound_rows(c, z_rows, (const fq*)bL.p, Rs, 0, nloc, (fq*)bmine.p);
  } else {
    size_t row0, nrows;
    comm_block(L_size, cm->rank, cm->world, &row0, &nrows);
    VPIN_HIP_TRY(hipMemcpyAsync(bL.p, Lvec, L_size * 32, hipMemcpyHostToDevice, c->stream));
    rc = poly_bound_rows(c, Z->d, (const fq*)bL.p, Rs, row0, nrows, (fq*)bmine.p);
  }
  if (rc) return rc;
  if (cm->serialize) VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  if ((rc = comm_allgather_dev(cm, c, bmine.p, ball.p, Rs * 32))) return rc;
  hipLaunchKernelGGL(poly_bound_reduce_kernel, dim3((unsigned)((Rs + kPB - 1) / kPB)), dim3(kPB), 0, c->stream, (const fq*)ball.p, Rs,
                     cm->world, (fq*)bout.p);
  VPIN_HIP_TRY(hipGetLastError());
  VPIN_HIP_TRY(hipMemcpyAsync(out_LZ, bout.p, Rs * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

}  // namespace vpin

using namespace vpin;

extern "C" int vpin_poly_bound(vpin_ctx* c, const vpin_table* Z, const uint8_t* Lvec, size_t L_size, uint8_t* out_LZ) {
  if (!c || !Z || !Z->d || !Lvec || !out_LZ || L_size == 0) return VPIN_EINVAL;
  if (Z->len % L_size != 0) return VPIN_ESHAPE;
  size_t Rs = Z->len / L_size;
  (void)hipSetDevice(c->device);
  size_t rows_per_chunk = L_size / 64 ? L_size / 64 : 1;
  int chunks = (int)((L_size + rows_per_chunk - 1) / rows_per_chunk);
  DevBuf bL(c), bpart(c), bout(c);
  int rc = VPIN_OK;
  if (bL.alloc(L_size * 32) || bpart.alloc((size_t)chunks * Rs * 32) || bout.alloc(Rs * 32)) {
    rc = VPIN_ENOMEM;
  } else {
    fq *dL = (fq*)bL.p, *dpart = (fq*)bpart.p, *dout = (fq*)bout.p;
    hipError_t e = hipMemcpyAsync(dL, Lvec, L_size * 32, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      dim3 grid((unsigned)((Rs + kPB - 1) / kPB), (unsigned)chunks);
      hipLaunchKernelGGL(poly_bound_kernel, grid, dim3(kPB), 0, c->stream, Z->d, dL, L_size, Rs, rows_per_chunk, dpart);
      hipLaunchKernelGGL(poly_bound_reduce_kernel, dim3(grid.x), dim3(kPB), 0, c->stream, dpart, Rs, chunks, dout);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_LZ, dout, Rs * 32, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { set_last_error("vpin_poly_bound", e); rc = VPIN_EHIP; }
  }
  return rc;
}
