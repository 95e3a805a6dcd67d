// trace.hip -- the memory trace of SNARK::encode for an instance given as host triplets, on the device.
//
// AddrTimestamps::new (Spartan/src/sparse_mlpoly.rs:232-265) walks the accesses of one side (rows, then columns) in order --
// matrix A, B, C, entry 0 .. N-1, padded entries reading address 0 -- with a counter per address:
//     read_ts[i] = audit[addr[i]]; audit[addr[i]] += 1
// i.e. read_ts = how many EARLIER accesses went to the same address, audit_ts = how many accesses an address got in all.  The
// host loop is sequential (round 4: one thread per side, 60 % of vpin_spark_encode's time for CNN A).  Here: a STABLE radix sort
// of (address, position) pairs (rocPRIM's radix_sort_pairs is stable) groups every address's accesses in their original order;
// the rank inside the group is the time stamp, the group's size the audit value.  Exact integers either way: the same u32s.
// (Device-built gadget instances do not come here: gadget_dev.hip writes their trace in closed form.)
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "ctx.h"
#include "spark_dev.h"

namespace vpin {

namespace {
constexpr int kTB = 256;

__global__ __launch_bounds__(kTB) void iota_kernel(uint32_t* __restrict__ v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kTB + threadIdx.x; i < n; i += (size_t)gridDim.x * kTB) v[i] = (uint32_t)i;
}
// head[p] = p where a new address starts in the sorted order, else 0 (a running maximum then gives every position its group's start)
__global__ __launch_bounds__(kTB) void heads_kernel(const uint32_t* __restrict__ keys, size_t n, uint32_t* __restrict__ head) {
  for (size_t p = (size_t)blockIdx.x * kTB + threadIdx.x; p < n; p += (size_t)gridDim.x * kTB)
    head[p] = (p > 0 && keys[p] != keys[p - 1]) ? (uint32_t)p : 0u;
}
__global__ __launch_bounds__(kTB) void ranks_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ pos,
                                                    const uint32_t* __restrict__ head, size_t n, size_t M, uint32_t* __restrict__ ts,
                                                    uint32_t* __restrict__ audit) {
  for (size_t p = (size_t)blockIdx.x * kTB + threadIdx.x; p < n; p += (size_t)gridDim.x * kTB) {
    const uint32_t rank = (uint32_t)p - head[p];
    ts[pos[p]] = rank;
    // the last access of the group: its size (an address out of range -- reported by spark_check_bounds -- must not be followed)
    if ((p + 1 == n || keys[p + 1] != keys[p]) && keys[p] < M) audit[keys[p]] = rank + 1u;
  }
}
// *bad = 1 when one of the first nnz entries is >= limit
__global__ __launch_bounds__(kTB) void bounds_kernel(const uint32_t* __restrict__ a, size_t nnz, uint32_t limit, uint32_t* __restrict__ bad) {
  for (size_t i = (size_t)blockIdx.x * kTB + threadIdx.x; i < nnz; i += (size_t)gridDim.x * kTB)
    if (a[i] >= limit) *bad = 1u;
}
struct MaxOp {
  __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};
inline unsigned grid_of(size_t n) { return (unsigned)std::min<size_t>(4096, (n + kTB - 1) / kTB + 1); }
}  // namespace

// addr: n = 3N addresses of one side (every one < M, M a power of two); ts: n read time stamps; audit: M audit time stamps
int spark_trace_timestamps(vpin_ctx* c, const uint32_t* addr, size_t n, size_t M, uint32_t* ts, uint32_t* audit) {
  if (!c || !addr || !ts || !audit || n == 0 || !is_pow2(M)) return VPIN_EINVAL;
  if (n > (size_t)0x7fffffff) return VPIN_ESHAPE;  // positions are u32 and the library calls take 32-bit sizes
  (void)hipSetDevice(c->device);
  DevBuf b_keys(c), b_pos_in(c), b_pos(c), b_head(c), b_tmp(c);
  if (b_keys.alloc(n * 4) || b_pos_in.alloc(n * 4) || b_pos.alloc(n * 4) || b_head.alloc(n * 4)) return VPIN_ENOMEM;
  uint32_t *keys = (uint32_t*)b_keys.p, *pos_in = (uint32_t*)b_pos_in.p, *pos = (uint32_t*)b_pos.p, *head = (uint32_t*)b_head.p;
  // all 32 bits: an address out of range (the caller reports it as VPIN_ESHAPE) must still sort as a group of its own
  const int end_bit = 32;
  (void)M;
  size_t tmp_sort = 0, tmp_scan = 0;
  VPIN_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_sort, addr, keys, (const uint32_t*)pos_in, pos, n, 0u, (unsigned)end_bit, c->stream));
  VPIN_HIP_TRY(rocprim::inclusive_scan(nullptr, tmp_scan, head, head, n, MaxOp(), c->stream));
  size_t tmp_bytes = std::max(tmp_sort, tmp_scan);
  if (b_tmp.alloc(tmp_bytes ? tmp_bytes : 256)) return VPIN_ENOMEM;
  hipLaunchKernelGGL(iota_kernel, dim3(grid_of(n)), dim3(kTB), 0, c->stream, pos_in, n);
  VPIN_HIP_TRY(rocprim::radix_sort_pairs(b_tmp.p, tmp_bytes, addr, keys, (const uint32_t*)pos_in, pos, n, 0u, (unsigned)end_bit, c->stream));
  hipLaunchKernelGGL(heads_kernel, dim3(grid_of(n)), dim3(kTB), 0, c->stream, (const uint32_t*)keys, n, head);
  tmp_bytes = std::max(tmp_sort, tmp_scan);
  VPIN_HIP_TRY(rocprim::inclusive_scan(b_tmp.p, tmp_bytes, head, head, n, MaxOp(), c->stream));
  VPIN_HIP_TRY(hipMemsetAsync(audit, 0, M * 4, c->stream));
  hipLaunchKernelGGL(ranks_kernel, dim3(grid_of(n)), dim3(kTB), 0, c->stream, (const uint32_t*)keys, (const uint32_t*)pos, (const uint32_t*)head, n,
                     M, ts, audit);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}

int spark_check_bounds(vpin_ctx* c, const uint32_t* a, size_t nnz, uint32_t limit, uint32_t* d_bad) {
  if (nnz) hipLaunchKernelGGL(bounds_kernel, dim3(grid_of(nnz)), dim3(kTB), 0, c->stream, a, nnz, limit, d_bad);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}

}  // namespace vpin
