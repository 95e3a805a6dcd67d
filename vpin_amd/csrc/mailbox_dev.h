// mailbox_dev.h -- scalars between a running kernel and the host through pinned host memory (spark.hip's persistent
// round kernel, bullet.hip's fused round kernel).  Every 32-byte scalar travels as three 16-byte pieces {seq, w, w, w}; the
// spare word of the third piece carries a checksum (seq xor the eight payload words).  A reader accepts a scalar when all
// three pieces carry the expected sequence number AND the checksum matches, and polls again otherwise -- no fence, no flag,
// no second round trip (tools/ubench_fs.hip measures the round trip).
// Platform note: on gfx950 + x86-64 a 16-byte aligned global_store_dwordx4 / movdqa crosses PCIe as one transaction, so a
// piece with the right sequence number is in practice whole; neither HIP nor the C++ memory model promises that, which is
// what the checksum is for: a piece torn 8 + 8 (new sequence word, stale payload words), or pieces of two different
// publications, fail it (up to a 2^-32 coincidence) and are simply read again.  Correctness does not rest on the
// 16-byte atomicity, only the absence of retries does.
#pragma once
#include <emmintrin.h>

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
__device__ __host__ __forceinline__ uint32_t mailbox_check(uint32_t seq, const uint32_t w[8]) {
  return seq ^ w[0] ^ w[1] ^ w[2] ^ w[3] ^ w[4] ^ w[5] ^ w[6] ^ w[7] ^ 0x5a5a5a5au;
}
__device__ __forceinline__ void publish_scalar(uint32_t* slot, const fq& v, uint32_t seq) {
  store16_system(slot, u32x4{seq, v.v[0], v.v[1], v.v[2]});
  store16_system(slot + 4, u32x4{seq, v.v[3], v.v[4], v.v[5]});
  store16_system(slot + 8, u32x4{seq, v.v[6], v.v[7], mailbox_check(seq, v.v)});
}
// the host's reply (three pieces at p) if it is whole and carries `seq`
__device__ __forceinline__ bool take_reply(const u32x4& c0, const u32x4& c1, const u32x4& c2, uint32_t seq, fq& out) {
  if (c0.x != seq || c1.x != seq || c2.x != seq) return false;
  out.v[0] = c0.y; out.v[1] = c0.z; out.v[2] = c0.w; out.v[3] = c1.y; out.v[4] = c1.z; out.v[5] = c1.w; out.v[6] = c2.y; out.v[7] = c2.z;
  return c2.w == mailbox_check(seq, out.v);
}
// ---- host side ----
static inline fq fq_zero_host() { fq z; memset(z.v, 0, sizeof z.v); return z; }

// 16-byte accesses as explicit SSE2 moves (a volatile vector access may be split by the compiler)
static inline void host_load16(const uint32_t* p, uint32_t out[4]) {
  _mm_storeu_si128(reinterpret_cast<__m128i*>(out), _mm_load_si128(reinterpret_cast<const __m128i*>(p)));
}
static inline void host_store16(uint32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  _mm_store_si128(reinterpret_cast<__m128i*>(p), _mm_set_epi32((int)d, (int)c, (int)b, (int)a));
}

// one scalar out of its three pieces; false while a piece still carries an older sequence number or the checksum fails
static inline bool tail_take(const uint32_t* slot, uint32_t want, fq* out) {
  uint32_t c0[4], c1[4], c2[4];
  host_load16(slot, c0);
  host_load16(slot + 4, c1);
  host_load16(slot + 8, c2);
  if (c0[0] != want || c1[0] != want || c2[0] != want) return false;
  fq v;
  v.v[0] = c0[1]; v.v[1] = c0[2]; v.v[2] = c0[3]; v.v[3] = c1[1]; v.v[4] = c1[2]; v.v[5] = c1[3]; v.v[6] = c2[1]; v.v[7] = c2[2];
  if (c2[3] != mailbox_check(want, v.v)) return false;
  *out = v;
  return true;
}
// the host's reply to a kernel: three pieces at `down`
static inline void host_reply(uint32_t* down, uint32_t seq, const uint8_t r[32]) {
  uint32_t wv[8];
  memcpy(wv, r, 32);
  host_store16(down, seq, wv[0], wv[1], wv[2]);
  host_store16(down + 4, seq, wv[3], wv[4], wv[5]);
  host_store16(down + 8, seq, wv[6], wv[7], mailbox_check(seq, wv));
}

}  // namespace vpin
