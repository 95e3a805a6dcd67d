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
