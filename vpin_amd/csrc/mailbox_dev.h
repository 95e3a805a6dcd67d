// mailbox_dev.h -- scalars between a running kernel and the host through pinned host memory (spark.hip's persistent
// round kernel, bullet.hip's fused round kernel).  Every 32-byte scalar travels as three 16-byte pieces {seq, w, w, w}: a
// 16-byte store (GPU -> host) or load (host -> GPU) is one bus transaction, so a piece that carries the expected sequence
// number is whole and current -- no fence, no flag, no second round trip (tools/ubench_fs.hip measures the round trip).
#pragma once
#include <cstdint>
#include <cstring>

#include "fq_dev.h"

namespace vpin {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// the three 16-byte pieces of the host's reply in one go: three uncached (system-scope) loads in flight, one wait
__device__ __forceinline__ void load48_system(const uint32_t* p, u32x4& c0, u32x4& c1, u32x4& c2) {
  asm volatile(
      "global_load_dwordx4 %0, %3, off sc0 sc1\n\t"
      "global_load_dwordx4 %1, %3, off offset:16 sc0 sc1\n\t"
      "global_load_dwordx4 %2, %3, off offset:32 sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(c0), "=&v"(c1), "=&v"(c2)
      : "v"(p)
      : "memory");
}
__device__ __forceinline__ u32x4 load16_system(const uint32_t* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store16_system(uint32_t* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void publish_scalar(uint32_t* slot, const fq& v, uint32_t seq) {
  store16_system(slot, u32x4{seq, v.v[0], v.v[1], v.v[2]});
  store16_system(slot + 4, u32x4{seq, v.v[3], v.v[4], v.v[5]});
  store16_system(slot + 8, u32x4{seq, v.v[6], v.v[7], 0u});
}
// ---- host side ----
typedef uint32_t tail_v4 __attribute__((vector_size(16), aligned(16)));
static inline fq fq_zero_host() { fq z; memset(z.v, 0, sizeof z.v); return z; }

// one scalar out of its three pieces; false while a piece still carries an older sequence number
static inline bool tail_take(const uint32_t* slot, uint32_t want, fq* out) {
  const tail_v4 c0 = *reinterpret_cast<const volatile tail_v4*>(slot), c1 = *reinterpret_cast<const volatile tail_v4*>(slot + 4),
                c2 = *reinterpret_cast<const volatile tail_v4*>(slot + 8);
  if (c0[0] != want || c1[0] != want || c2[0] != want) return false;
  out->v[0] = c0[1]; out->v[1] = c0[2]; out->v[2] = c0[3]; out->v[3] = c1[1]; out->v[4] = c1[2]; out->v[5] = c1[3];
  out->v[6] = c2[1]; out->v[7] = c2[2];
  return true;
}

}  // namespace vpin
