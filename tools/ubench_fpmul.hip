// ubench_fpmul.hip -- A/B of the GF(2^255-19) product the row-commitment MSM spends its time in:
//   A  fp_mul of vpin_amd/csrc/fp_dev.h: eight 32-bit limbs, 64 multiply-adds v_mad_u64_u32 each followed by a
//      v_addc_co_u32 into a 96-bit column accumulator, fold of the high half with 2^256 = 38
//   B  carry-free radix: ten limbs of 26/25 bits (the ref10 layout), 100 v_mad_u64_u32 into plain 64-bit column
//      accumulators (ten products of < 2^57.3 each: no carry can leave 64 bits), wrapped terms pre-multiplied by 19, one
//      carry pass.  Values STAY in the ten-limb form from one product to the next -- the best case for B: no packing to the
//      96-byte table entries, no unpacking of loaded operands.
// Both run the same dependent chain x <- x * y per lane, 12 waves per CU as in msm_rows_kernel (3 workgroups of 256), and
// the final values are compared mod p.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vpin_amd/csrc tools/ubench_fpmul.hip -o tools/ubench_fpmul
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fp10_dev.h"

using namespace vpin;

__global__ __launch_bounds__(256, 3) void chain_a(const fp* __restrict__ in, fp* __restrict__ out, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  fp x = fp_load(in + 2 * t), y = fp_load(in + 2 * t + 1);
  for (int i = 0; i < iters; i++) x = fp_mul(x, y);
  fp_store(out + t, fp_freeze(x));
}

__global__ __launch_bounds__(256, 3) void chain_b(const fp* __restrict__ in, fp* __restrict__ out, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  fe10 x = fe10_from_fp(fp_load(in + 2 * t)), y = fe10_from_fp(fp_load(in + 2 * t + 1));
  for (int i = 0; i < iters; i++) x = fe10_mul(x, y);
  fp_store(out + t, fp_freeze(fe10_to_fp(x)));
}

// the point addition the MSM's inner loop performs, on a chain: acc <- acc + (table entry), 7 products + 8 add/sub
__global__ __launch_bounds__(256, 3) void chain_add_a(const fp* __restrict__ in, fp* __restrict__ out, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  ge_ext acc = ge_identity();
  ge_niels q;
  q.ypx = fp_load(in + 2 * t); q.ymx = fp_load(in + 2 * t + 1); q.xy2d = fp_add(q.ypx, q.ymx);
  for (int i = 0; i < iters; i++) acc = ge_add_niels(acc, q, (i & 1) != 0);
  fp_store(out + t, fp_freeze(fp_add(fp_add(acc.X, acc.Y), fp_add(acc.Z, acc.T))));
}

__global__ __launch_bounds__(256, 3) void chain_add_b(const fp* __restrict__ in, fp* __restrict__ out, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  ge10 acc = ge10_identity();
  ge_niels q;
  q.ypx = fp_load(in + 2 * t); q.ymx = fp_load(in + 2 * t + 1); q.xy2d = fp_add(q.ypx, q.ymx);
  for (int i = 0; i < iters; i++) acc = ge10_add_niels(acc, q, (i & 1) != 0);
  const ge_ext e = ge10_to_ext(acc);
  fp_store(out + t, fp_freeze(fp_add(fp_add(e.X, e.Y), fp_add(e.Z, e.T))));
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 3, threads = blocks * 256;
  std::vector<uint32_t> h((size_t)threads * 16);
  uint64_t s = 0x9e3779b97f4a7c15ull;
  for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
  for (size_t i = 7; i < h.size(); i += 8) h[i] &= 0x7fffffffu;  // < 2^255
  fp *din, *da, *db;
  CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&da, (size_t)threads * 32)); CK(hipMalloc(&db, (size_t)threads * 32));
  CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms[4] = {0, 0, 0, 0};
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(chain_a, dim3(blocks), dim3(256), 0, 0, din, da, iters); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[0], e0, e1));
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(chain_b, dim3(blocks), dim3(256), 0, 0, din, db, iters); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[1], e0, e1));
  }
  std::vector<uint32_t> ra((size_t)threads * 8), rb((size_t)threads * 8);
  CK(hipMemcpy(ra.data(), da, ra.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(rb.data(), db, rb.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < ra.size(); i++) bad += ra[i] != rb[i];
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(chain_add_a, dim3(blocks), dim3(256), 0, 0, din, da, iters / 8); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[2], e0, e1));
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(chain_add_b, dim3(blocks), dim3(256), 0, 0, din, db, iters / 8); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[3], e0, e1));
  }
  CK(hipMemcpy(ra.data(), da, ra.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(rb.data(), db, rb.size() * 4, hipMemcpyDeviceToHost));
  size_t bad_add = 0;
  for (size_t i = 0; i < ra.size(); i++) bad_add += ra[i] != rb[i];
  const double n = (double)threads * iters;
  printf("device %s, %d CUs, %d lanes x %d dependent products\n", prop.name, prop.multiProcessorCount, threads, iters);
  printf("A  8 x 32-bit limbs, mad + addc (fp_dev.h fp_mul)      : %8.3f ms  %7.2f G products/s\n", ms[0], n / ms[0] / 1e6);
  printf("B  10 x 25.5-bit limbs, carry-free accumulation        : %8.3f ms  %7.2f G products/s  (B/A time %.3f)\n", ms[1], n / ms[1] / 1e6, ms[1] / ms[0]);
  printf("results equal mod p: %s (%zu words differ)\n", bad ? "NO" : "yes", bad);
  const double na = (double)threads * (iters / 8);
  printf("A  point + table entry (7 products, 8 add/sub), chain  : %8.3f ms  %7.2f G additions/s\n", ms[2], na / ms[2] / 1e6);
  printf("B  the same in ten limbs (entry unpacked per addition) : %8.3f ms  %7.2f G additions/s  (B/A time %.3f)\n", ms[3], na / ms[3] / 1e6, ms[3] / ms[2]);
  printf("point chains equal: %s\n", bad_add ? "NO" : "yes");
  printf("JSON {\"products_Gps_8x32\": %.2f, \"products_Gps_10x25\": %.2f, \"point_adds_Gps_8x32\": %.2f, \"point_adds_Gps_10x25\": %.2f, \"equal\": %s}\n",
         n / ms[0] / 1e6, n / ms[1] / 1e6, na / ms[2] / 1e6, na / ms[3] / 1e6, (bad || bad_add) ? "false" : "true");
  return (bad || bad_add) ? 2 : 0;
}
