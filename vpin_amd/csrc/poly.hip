// poly.hip -- dense-polynomial helpers of the evaluation proof, gfx950.
//   DensePolynomial::bound   Spartan/src/dense_mlpoly.rs:220-227   LZ[i] = sum_j L[j]*Z[j*R+i]
// Z is read exactly once, row-major, so every wavefront streams 2 KiB contiguous per row.
#include <cstring>

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
