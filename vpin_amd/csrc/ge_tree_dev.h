// ge_tree_dev.h -- the block tree that sums the points of a workgroup with four lanes per addition (msm.hip, msm_pip.hip)
#pragma once
#include "fp_dev.h"

namespace vpin {

// ---- block tree with four lanes per addition -----------------------------------------------------------------
// The few-row MSMs are latency bound: one wave per SIMD issues a modular product in ~0.5 us, and an addition of two
// extended points is nine of them in a row on one lane.  Here the four products of each half of the addition
// (add-2008-hwcd-3: A,B,C,D then X3,Y3,Z3,T3) run on the four lanes of a quad -- same code path, operands picked by the
// lane's role, the halves exchanged with lane shuffles -- so a level costs three products instead of nine.
__device__ __forceinline__ fp fp_shfl_from(const fp& a, int src) {
  fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl(a.v[i], src, 64);
  return r;
}
__device__ __forceinline__ fp fp_pick(bool c, const fp& a, const fp& b) {
  fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
// sh[0] = sum of sh[0..n) (n a power of two, n <= blockDim.x); every thread of the block calls it.
// split > 0 (a power of two below n): the entries alternate in runs of `split` between two sums (index & split); the
// level that would mix them is skipped and the levels below it reduce both runs: sh[0] = sum of the entries with
// (index & split) == 0, sh[split] = sum of the others.
__device__ __forceinline__ void ge_tree_quad(ge_ext* sh, int n, int split = 0) {
  const int role = threadIdx.x & 3, qbase = (threadIdx.x & 63) & ~3;
  const fp* shf = reinterpret_cast<const fp*>(sh);
  fp* shw = reinterpret_cast<fp*>(sh);
  for (int s = n / 2; s >= 1; s >>= 1) {
    if (s == split) continue;
    const int items = s < split ? 2 * s : s;
    for (int w = threadIdx.x >> 2; w < items; w += (int)(blockDim.x >> 2)) {
      const int i = w < s ? w : split + (w - s);
      // role 0: (Y1-X1)(Y2-X2)   role 1: (Y1+X1)(Y2+X2)   role 2: 2d T1 T2   role 3: 2 Z1 Z2
      const int f0 = role < 2 ? 1 : (role == 2 ? 3 : 2);  // Y | T | Z
      const fp p0 = shf[4 * i + f0], q0 = shf[4 * (i + s) + f0];
      fp u = p0, v = q0;
      if (role < 2) {  // uniform per quad pair: both take the same instructions, the select below is per lane
        const fp p1 = shf[4 * i], q1 = shf[4 * (i + s)];
        u = fp_pick(role == 0, fp_sub(p0, p1), fp_add(p0, p1));
        v = fp_pick(role == 0, fp_sub(q0, q1), fp_add(q0, q1));
      }
      fp m = fp_mul(u, v);
      m = fp_mul(m, fp_pick(role == 2, FP_D2(), fp_one()));
      m = fp_pick(role == 3, fp_add(m, m), m);
      const fp a = fp_shfl_from(m, qbase), b = fp_shfl_from(m, qbase + 1), c = fp_shfl_from(m, qbase + 2),
               d = fp_shfl_from(m, qbase + 3);
      const fp E = fp_sub(b, a), H = fp_add(b, a), F = fp_sub(d, c), G = fp_add(d, c);
      // X3 = E F, Y3 = G H, Z3 = F G, T3 = E H
      u = fp_pick(role == 0 || role == 3, E, fp_pick(role == 1, G, F));
      v = fp_pick(role == 0, F, fp_pick(role == 2, G, H));
      shw[4 * i + role] = fp_mul(u, v);
    }
    __syncthreads();
  }
}

}  // namespace vpin
