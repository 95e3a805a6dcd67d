// msm_var.hip -- variable-base multiscalar multiplication over ristretto255 on gfx950.
//
// Replaces GroupElement::vartime_multiscalar_mul (Spartan/src/group.rs:103-122) where the bases are NOT a generator
// stream with a window table: the verifier's C_LZ = <L, C> over the L decompressed row commitments
// (Spartan/src/dense_mlpoly.rs:381-404), and the row-wise sum of two commitment vectors
// (vPIN_proof_generation/src/commit_test.rs:340-361: comm_para[i] + comm_input[i]).
//
// The sizes are the row counts of Hyrax commitments (2^4 .. 2^14 points), not millions: a bucket method would spend its
// time in the bucket reduction.  One lane per (scalar, point): decompress (RFC 9496 4.3.1: one exponentiation), then a
// left-to-right double-and-add over the scalar's bits in the ten-limb field form (fp10_dev.h), then the block's points are
// summed by the four-lanes-per-addition tree of msm.hip and one partial point per workgroup goes back.  16384 points are
// 256 one-wave workgroups: the whole chip, ~3300 dependent products deep.
#include <cstring>
#include <vector>

#include "ctx.h"
#include "fp10_dev.h"

namespace vpin {

// d = -121665/121666 (RFC 9496 section 4)
__device__ __forceinline__ fp FP_D() { return fp_const(0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu); }

// CompressedRistretto::decompress (RFC 9496 4.3.1); false for a non-canonical, negative or off-group encoding
__device__ __noinline__ bool ge_decompress(const fp& s_in, ge_ext& out) {
  const fp s = fp_freeze(s_in);
  bool canonical = true;
#pragma unroll
  for (int i = 0; i < 8; i++) canonical = canonical && (s.v[i] == s_in.v[i]);
  if (!canonical || (s.v[0] & 1u)) return false;
  const fp one = fp_one();
  const fp ss = fp_sqr(s), u1 = fp_sub(one, ss), u2 = fp_add(one, ss), u2s = fp_sqr(u2);
  const fp v = fp_sub(fp_neg(fp_mul(FP_D(), fp_sqr(u1))), u2s);
  bool sq;
  const fp invsqrt = fp_invsqrt(fp_mul(v, u2s), &sq);
  const fp den_x = fp_mul(invsqrt, u2), den_y = fp_mul(fp_mul(invsqrt, den_x), v);
  const fp x = fp_abs(fp_mul(fp_add(s, s), den_x)), y = fp_mul(u1, den_y), t = fp_mul(x, y);
  if (!sq || fp_is_negative(t) || fp_is_zero(y)) return false;
  out.X = x; out.Y = y; out.Z = one; out.T = t;
  return true;
}

constexpr int kVarBlock = 64;  // one wave per workgroup: 16384 points fill the chip

// sh[0] = sum of sh[0..n) with four lanes per addition (the tree of msm.hip, for a 64-lane block)
__device__ __forceinline__ fp fpv_shfl_from(const fp& a, int src) {
  fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl(a.v[i], src, 64);
  return r;
}
__device__ __forceinline__ fp fpv_pick(bool c, const fp& a, const fp& b) {
  fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
__device__ __forceinline__ void ge_tree_quad64(ge_ext* sh, int n) {
  const int role = threadIdx.x & 3, qbase = (threadIdx.x & 63) & ~3;
  const fp* shf = reinterpret_cast<const fp*>(sh);
  fp* shw = reinterpret_cast<fp*>(sh);
  for (int s = n / 2; s >= 1; s >>= 1) {
    for (int w = threadIdx.x >> 2; w < s; w += (int)(blockDim.x >> 2)) {
      const int i = w;
      const int f0 = role < 2 ? 1 : (role == 2 ? 3 : 2);
      const fp p0 = shf[4 * i + f0], q0 = shf[4 * (i + s) + f0];
      fp u = p0, v = q0;
      if (role < 2) {
        const fp p1 = shf[4 * i], q1 = shf[4 * (i + s)];
        u = fpv_pick(role == 0, fp_sub(p0, p1), fp_add(p0, p1));
        v = fpv_pick(role == 0, fp_sub(q0, q1), fp_add(q0, q1));
      }
      fp m = fp_mul(u, v);
      m = fp_mul(m, fpv_pick(role == 2, FP_D2(), fp_one()));
      m = fpv_pick(role == 3, fp_add(m, m), m);
      const fp a = fpv_shfl_from(m, qbase), b = fpv_shfl_from(m, qbase + 1), c = fpv_shfl_from(m, qbase + 2),
               d = fpv_shfl_from(m, qbase + 3);
      const fp E = fp_sub(b, a), H = fp_add(b, a), F = fp_sub(d, c), G = fp_add(d, c);
      u = fpv_pick(role == 0 || role == 3, E, fpv_pick(role == 1, G, F));
      v = fpv_pick(role == 0, F, fpv_pick(role == 2, G, H));
      shw[4 * i + role] = fp_mul(u, v);
    }
    __syncthreads();
  }
}

// partial[b] = sum over the block's lanes of s_i * P_i; bad[0] != 0 when a point does not decode.
// scalars: Montgomery form (mont != 0) or canonical integers.
__global__ __launch_bounds__(kVarBlock) void msm_var_kernel(const fq* __restrict__ scalars, const fp* __restrict__ points, size_t n, int mont,
                                                            ge_ext* __restrict__ partial, uint32_t* __restrict__ bad) {
  const size_t i = (size_t)blockIdx.x * kVarBlock + threadIdx.x;
  ge10 acc = ge10_identity();
  if (i < n) {
    ge_ext P;
    fq s = fq_load(scalars + i);
    if (mont) s = fq_from_mont(s);
    const bool ok = ge_decompress(fp_load(points + i), P);
    if (!ok) atomicOr(bad, 1u);
    if (ok && !fq_is_zero(s)) {
      // the point as an affine table entry (Z = 1 after decompression): (y + x, y - x, 2 d x y)
      ge_niels q;
      q.ypx = fp_add(P.Y, P.X); q.ymx = fp_sub(P.Y, P.X); q.xy2d = fp_mul(P.T, FP_D2());
      int top = 255;
      while (top > 0 && !((s.v[top >> 5] >> (top & 31)) & 1u)) top--;
      acc = ge10_from_ext(P);
      for (int b = top - 1; b >= 0; b--) {
        acc = ge10_double(acc);
        if ((s.v[b >> 5] >> (b & 31)) & 1u) acc = ge10_add_niels(acc, q, false);
      }
    }
  }
  __shared__ ge_ext sh[kVarBlock];
  sh[threadIdx.x] = ge10_to_ext(acc);
  __syncthreads();
  ge_tree_quad64(sh, kVarBlock);
  if (threadIdx.x == 0) {
    ge_ext* o = partial + blockIdx.x;
    fp_store(&o->X, sh[0].X); fp_store(&o->Y, sh[0].Y); fp_store(&o->Z, sh[0].Z); fp_store(&o->T, sh[0].T);
  }
}

// out[i] = compress(decompress(a[i]) + decompress(b[i]))
__global__ __launch_bounds__(kVarBlock) void points_add_kernel(const fp* __restrict__ a, const fp* __restrict__ b, size_t n,
                                                               fp* __restrict__ out, uint32_t* __restrict__ bad) {
  const size_t i = (size_t)blockIdx.x * kVarBlock + threadIdx.x;
  if (i >= n) return;
  ge_ext P, Q;
  const bool ok = ge_decompress(fp_load(a + i), P) && ge_decompress(fp_load(b + i), Q);
  if (!ok) { atomicOr(bad, 1u); return; }
  fp_store(out + i, ge_compress(ge_add(P, Q)));
}

// one block: sum of m partial points -> compressed and canonical X|Y|Z|T
__global__ __launch_bounds__(kVarBlock) void msm_var_finish_kernel(const ge_ext* __restrict__ partial, size_t m, fp* __restrict__ out32,
                                                                   fp* __restrict__ out_xyzt) {
  __shared__ ge_ext sh[kVarBlock];
  ge_ext acc = ge_identity();
  for (size_t k = threadIdx.x; k < m; k += kVarBlock) {
    ge_ext p;
    p.X = fp_load(&partial[k].X); p.Y = fp_load(&partial[k].Y); p.Z = fp_load(&partial[k].Z); p.T = fp_load(&partial[k].T);
    acc = ge_add(acc, p);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  ge_tree_quad64(sh, kVarBlock);
  if (threadIdx.x == 0) {
    const ge_ext r = sh[0];
    fp_store(out32, ge_compress(r));
    fp_store(out_xyzt, fp_freeze(r.X)); fp_store(out_xyzt + 1, fp_freeze(r.Y)); fp_store(out_xyzt + 2, fp_freeze(r.Z));
    fp_store(out_xyzt + 3, fp_freeze(r.T));
  }
}

}  // namespace vpin

using namespace vpin;

extern "C" {

// GroupElement::vartime_multiscalar_mul (Spartan/src/group.rs:103-122) over arbitrary points given in their compressed
// ristretto255 encodings: out = sum_i s_i * decompress(P_i).  VPIN_EVERIFY when an encoding does not decode (the reference
// unwraps CompressedGroup::decompress and panics).
int vpin_msm(vpin_ctx* c, const uint8_t* scalars_mont, const uint8_t* points_compressed, size_t n, uint8_t* out_compressed,
             uint8_t* out_xyzt) {
  if (!c || !scalars_mont || !points_compressed || n == 0 || (!out_compressed && !out_xyzt)) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  const size_t nb = (n + kVarBlock - 1) / kVarBlock;
  DevBuf ds(c), dp(c), dpart(c), dout(c), dbad(c);
  if (ds.alloc(n * 32) || dp.alloc(n * 32) || dpart.alloc(nb * sizeof(ge_ext)) || dout.alloc(32 + 128) || dbad.alloc(4)) return VPIN_ENOMEM;
  VPIN_HIP_TRY(hipMemcpyAsync(ds.p, scalars_mont, n * 32, hipMemcpyHostToDevice, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync(dp.p, points_compressed, n * 32, hipMemcpyHostToDevice, c->stream));
  VPIN_HIP_TRY(hipMemsetAsync(dbad.p, 0, 4, c->stream));
  {
    ProfScope ps(c, VPIN_K_MSM, 64.0 * (double)n);
    hipLaunchKernelGGL(msm_var_kernel, dim3((unsigned)nb), dim3(kVarBlock), 0, c->stream, (const fq*)ds.p, (const fp*)dp.p, n, 1,
                       (ge_ext*)dpart.p, (uint32_t*)dbad.p);
    hipLaunchKernelGGL(msm_var_finish_kernel, dim3(1), dim3(kVarBlock), 0, c->stream, (const ge_ext*)dpart.p, nb, (fp*)dout.p,
                       (fp*)((uint8_t*)dout.p + 32));
  }
  VPIN_HIP_TRY(hipGetLastError());
  uint8_t host[160];
  uint32_t bad = 0;
  VPIN_HIP_TRY(hipMemcpyAsync(host, dout.p, 160, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync(&bad, dbad.p, 4, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  if (bad) return VPIN_EVERIFY;
  if (out_compressed) memcpy(out_compressed, host, 32);
  if (out_xyzt) memcpy(out_xyzt, host + 32, 128);
  return VPIN_OK;
}

// out[i] = compress(decompress(a[i]) + decompress(b[i])): the row-wise sum of two Hyrax commitments
// (vPIN_proof_generation/src/commit_test.rs:340-361 on the verifier's side, proof_point_mult.rs:75-80 on the prover's)
int vpin_points_add(vpin_ctx* c, const uint8_t* a_compressed, const uint8_t* b_compressed, size_t n, uint8_t* out_compressed) {
  if (!c || !a_compressed || !b_compressed || !out_compressed || n == 0) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  DevBuf da(c), db(c), dout(c), dbad(c);
  if (da.alloc(n * 32) || db.alloc(n * 32) || dout.alloc(n * 32) || dbad.alloc(4)) return VPIN_ENOMEM;
  VPIN_HIP_TRY(hipMemcpyAsync(da.p, a_compressed, n * 32, hipMemcpyHostToDevice, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync(db.p, b_compressed, n * 32, hipMemcpyHostToDevice, c->stream));
  VPIN_HIP_TRY(hipMemsetAsync(dbad.p, 0, 4, c->stream));
  hipLaunchKernelGGL(points_add_kernel, dim3((unsigned)((n + kVarBlock - 1) / kVarBlock)), dim3(kVarBlock), 0, c->stream, (const fp*)da.p,
                     (const fp*)db.p, n, (fp*)dout.p, (uint32_t*)dbad.p);
  VPIN_HIP_TRY(hipGetLastError());
  uint32_t bad = 0;
  VPIN_HIP_TRY(hipMemcpyAsync(out_compressed, dout.p, n * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync(&bad, dbad.p, 4, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return bad ? VPIN_EVERIFY : VPIN_OK;
}

}  // extern "C"
