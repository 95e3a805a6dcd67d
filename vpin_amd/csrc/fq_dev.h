// fq_dev.h -- device-side F_q arithmetic for gfx950 (CDNA4).
//
// q = 2^252 + 27742317777372353535851937790883648493 (ristretto255 scalar field).
// Elements are kept in Montgomery form with R = 2^256, eight 32-bit little-endian
// limbs: byte-for-byte the reference's in-memory `Scalar([u64;4])`
// (Spartan/src/scalar/ristretto255.rs:199-200), so tables cross the C ABI by memcpy.
//
// 255-bit modular integer work: no MFMA.  The multiplier is v_mad_u64_u32
// (32x32+64 -> 64); the Montgomery reduction exploits q's shape: limbs 4..6 of q are
// zero and limb 7 is 2^28, so a reduction step costs 4 multiplies and one shift.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef VPIN_MUL_INLINE
#define VPIN_MUL_INLINE __forceinline__
#endif

namespace vpin {

struct alignas(16) fq {
  uint32_t v[8];
};

// q, little-endian 32-bit limbs
#define VPIN_Q0 0x5cf5d3edu
#define VPIN_Q1 0x5812631au
#define VPIN_Q2 0xa2f79cd6u
#define VPIN_Q3 0x14def9deu
#define VPIN_Q7 0x10000000u
// -q^{-1} mod 2^32 (low word of the reference's INV, ristretto255.rs:298)
#define VPIN_QINV32 0x12547e1bu

__host__ __device__ __forceinline__ constexpr uint32_t fq_modulus_limb(int i) {
  return i == 0 ? VPIN_Q0 : i == 1 ? VPIN_Q1 : i == 2 ? VPIN_Q2 : i == 3 ? VPIN_Q3 : i == 7 ? VPIN_Q7 : 0u;
}

__device__ __forceinline__ fq fq_zero() {
  fq r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = 0;
  return r;
}

// R mod q = Montgomery one (ristretto255.rs:301-306)
__device__ __forceinline__ fq fq_one() {
  fq r;
  r.v[0] = 0x8d98951du; r.v[1] = 0xd6ec3174u; r.v[2] = 0x737dcf70u; r.v[3] = 0xc6ef5bf4u;
  r.v[4] = 0xfffffffeu; r.v[5] = 0xffffffffu; r.v[6] = 0xffffffffu; r.v[7] = 0x0fffffffu;
  return r;
}

__device__ __forceinline__ fq fq_load(const fq* __restrict__ p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 lo = q[0], hi = q[1];
  fq r;
  r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
  r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
  return r;
}

__device__ __forceinline__ void fq_store(fq* __restrict__ p, const fq& a) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
  q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

__device__ __forceinline__ bool fq_is_zero(const fq& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i];
  return o == 0;
}

// Carry chains use clang's add/sub-with-carry builtins: they lower to v_add_co_u32 / v_addc_co_u32
// links (one VALU instruction per limb), a quarter of what the 64-bit-per-limb formulation cost.

// r = (t >= q) ? t - q : t      (t < 2q)
__device__ __forceinline__ fq fq_cond_sub_q(const fq& t) {
  fq d;
  unsigned bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(t.v[i], fq_modulus_limb(i), bw, &bw);
  fq r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = bw ? t.v[i] : d.v[i];
  return r;
}

// ristretto255.rs:746-757
__device__ __forceinline__ fq fq_add(const fq& a, const fq& b) {
  fq t;
  unsigned cy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) t.v[i] = __builtin_addc(a.v[i], b.v[i], cy, &cy);
  return fq_cond_sub_q(t);  // a+b < 2q < 2^254: no carry out of limb 7
}

// ristretto255.rs:729-743
__device__ __forceinline__ fq fq_sub(const fq& a, const fq& b) {
  fq d;
  unsigned bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d.v[i] = __builtin_subc(a.v[i], b.v[i], bw, &bw);
  // add q back on underflow
  fq r;
  unsigned cy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __builtin_addc(d.v[i], bw ? fq_modulus_limb(i) : 0u, cy, &cy);
  return r;
}

__device__ __forceinline__ fq fq_neg(const fq& a) { return fq_sub(fq_zero(), a); }

__device__ __forceinline__ fq fq_dbl(const fq& a) { return fq_add(a, a); }

// One Montgomery reduction step on a 9-word window t[0..8] (+ overflow t9):
// adds m*q with m = t0 * (-q^{-1}) so the low word cancels, then shifts down 32 bits.
// q's limbs 4..6 are zero and limb 7 is 2^28, so only 4 multiplies are needed.
#define VPIN_MONT_STEP(t, t9)                                              \
  {                                                                        \
    uint32_t m_ = (t)[0] * VPIN_QINV32;                                    \
    uint64_t c_ = (uint64_t)m_ * VPIN_Q0 + (t)[0];                         \
    c_ >>= 32;                                                             \
    c_ += (uint64_t)m_ * VPIN_Q1 + (t)[1]; (t)[0] = (uint32_t)c_; c_ >>= 32; \
    c_ += (uint64_t)m_ * VPIN_Q2 + (t)[2]; (t)[1] = (uint32_t)c_; c_ >>= 32; \
    c_ += (uint64_t)m_ * VPIN_Q3 + (t)[3]; (t)[2] = (uint32_t)c_; c_ >>= 32; \
    c_ += (t)[4]; (t)[3] = (uint32_t)c_; c_ >>= 32;                         \
    c_ += (t)[5]; (t)[4] = (uint32_t)c_; c_ >>= 32;                         \
    c_ += (t)[6]; (t)[5] = (uint32_t)c_; c_ >>= 32;                         \
    c_ += ((uint64_t)m_ << 28) + (t)[7]; (t)[6] = (uint32_t)c_; c_ >>= 32;  \
    c_ += (t)[8]; (t)[7] = (uint32_t)c_; c_ >>= 32;                         \
    (t)[8] = (t9) + (uint32_t)c_;                                          \
  }

// ---- multiply-accumulate building block ------------------------------------------------------
// A column accumulator is 96 bits: acc (64) + ovf (32).  One MAC is v_mad_u64_u32 (32x32+64,
// carry-out to an SGPR pair) followed by v_addc_co_u32 folding that carry into ovf -- two
// instructions per limb product, against ~6 for what hipcc derives from portable C (measured on
// MI355X: profiles/r01_ubench_valu.txt).  MACs are issued in pairs (mad, mad, addc, addc) so the
// VALU-writes-SGPR -> VALU-reads-SGPR wait state is filled with the second multiply instead of an
// s_nop; each instruction is its own asm statement, so hipcc's hazard recognizer still sees every
// operand and pads whatever this ordering leaves open.
struct acc96 {
  uint64_t lo;
  uint32_t hi;
};

__device__ __forceinline__ void mac1(acc96& c, uint32_t a, uint32_t b) {
  uint64_t cy;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c.lo), "=s"(cy) : "v"(a), "v"(b));
  asm("v_addc_co_u32 %0, vcc, %0, 0, %1" : "+v"(c.hi) : "s"(cy) : "vcc");
}

__device__ __forceinline__ void mac2(acc96& c, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
  uint64_t cy0, cy1;
  // volatile: keeps the (mad, mad, addc, addc) order, which is what fills the wait state
  asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c.lo), "=s"(cy0) : "v"(a0), "v"(b0));
  asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c.lo), "=s"(cy1) : "v"(a1), "v"(b1));
  asm volatile("v_addc_co_u32 %0, vcc, %0, 0, %1" : "+v"(c.hi) : "s"(cy0) : "vcc");
  asm volatile("v_addc_co_u32 %0, vcc, %0, 0, %1" : "+v"(c.hi) : "s"(cy1) : "vcc");
}

// c += x (64-bit) with carry into hi
__device__ __forceinline__ void acc_add64(acc96& c, uint64_t x) {
  unsigned cy = 0;
  uint32_t l = __builtin_addc((uint32_t)c.lo, (uint32_t)x, 0u, &cy);
  uint32_t h = __builtin_addc((uint32_t)(c.lo >> 32), (uint32_t)(x >> 32), cy, &cy);
  c.hi = __builtin_addc(c.hi, 0u, cy, &cy);
  c.lo = (uint64_t)l | ((uint64_t)h << 32);
}

// next column: drop the low 32 bits
__device__ __forceinline__ void acc_shift(acc96& c) {
  c.lo = (c.lo >> 32) | ((uint64_t)c.hi << 32);
  c.hi = 0;
}

// sum_{i+j=k} a_i*b_j for the limb range [lo_i, hi_i], paired
template <int K, int I0, int I1>
__device__ __forceinline__ void mac_column(acc96& c, const uint32_t* a, const uint32_t* b) {
  constexpr int n = I1 - I0 + 1;
#pragma unroll
  for (int t = 0; t + 1 < n; t += 2) mac2(c, a[I0 + t], b[K - I0 - t], a[I0 + t + 1], b[K - I0 - t - 1]);
  if (n & 1) mac1(c, a[I1], b[K - I1]);
}

// Montgomery product a*b*R^{-1} mod q (ristretto255.rs:701-726 + :653-698), product-scanning
// (FIPS) form: column k collects a_i*b_j (i+j=k) and m_i*q_j (i+j=k); q_4..q_6 = 0 and
// q_7 = 2^28, so the reduction costs 4 multiplies and one shift-add per m_i.
__device__ VPIN_MUL_INLINE fq fq_mul(fq a, fq b) {
  acc96 c{0, 0};
  uint32_t m[8];
  fq r;
  const uint32_t q0 = VPIN_Q0, q1 = VPIN_Q1, q2 = VPIN_Q2, q3 = VPIN_Q3;
#define VPIN_M(i) m[(i) < 0 ? 0 : (i) > 7 ? 7 : (i)]
  // columns 0..7: a*b terms, m_i*q_{k-i} for the earlier m's, then m_k cancels the low word
#define VPIN_FQ_LOW_COL(k)                                                                  \
  mac_column<k, 0, k>(c, a.v, b.v);                                                          \
  if (k >= 2) mac2(c, VPIN_M(k - 1), q1, VPIN_M(k - 2), q2);                                  \
  else if (k == 1) mac1(c, VPIN_M(0), q1);                                                   \
  if (k >= 3) mac1(c, VPIN_M(k - 3), q3);                                                    \
  if (k >= 7) acc_add64(c, (uint64_t)VPIN_M(k - 7) << 28);                                   \
  m[k] = (uint32_t)c.lo * VPIN_QINV32;                                                       \
  mac1(c, m[k], q0);                                                                         \
  acc_shift(c);
  VPIN_FQ_LOW_COL(0) VPIN_FQ_LOW_COL(1) VPIN_FQ_LOW_COL(2) VPIN_FQ_LOW_COL(3)
  VPIN_FQ_LOW_COL(4) VPIN_FQ_LOW_COL(5) VPIN_FQ_LOW_COL(6) VPIN_FQ_LOW_COL(7)
#undef VPIN_FQ_LOW_COL
  // columns 8..14: remaining a*b terms and m_i*q_j with j = k-i in {1,2,3,7}, i <= 7
#define VPIN_FQ_HIGH_COL(k)                                                                 \
  mac_column<k, k - 7, 7>(c, a.v, b.v);                                                      \
  if (k == 8) mac2(c, VPIN_M(7), q1, VPIN_M(6), q2);                                          \
  if (k == 9) mac1(c, VPIN_M(7), q2);                                                        \
  if (k <= 10) mac1(c, VPIN_M(k - 3), q3);                                                   \
  acc_add64(c, (uint64_t)VPIN_M(k - 7) << 28);                                               \
  r.v[k - 8] = (uint32_t)c.lo;                                                               \
  acc_shift(c);
  VPIN_FQ_HIGH_COL(8) VPIN_FQ_HIGH_COL(9) VPIN_FQ_HIGH_COL(10) VPIN_FQ_HIGH_COL(11)
  VPIN_FQ_HIGH_COL(12) VPIN_FQ_HIGH_COL(13) VPIN_FQ_HIGH_COL(14)
#undef VPIN_FQ_HIGH_COL
#undef VPIN_M
  r.v[7] = (uint32_t)c.lo;  // column 15: only the carry
  return fq_cond_sub_q(r);  // result < 2q
}

__device__ __forceinline__ fq fq_sqr(const fq& a) { return fq_mul(a, a); }

// ---- lazy accumulation: sum of products, reduced once ---------------------------------------------------------------
// A round kernel's sums  e += E * x  (x already reduced) do not need the Montgomery reduction of every product: the
// 512-bit products are added up in sixteen limbs and reduced once per few terms.  fqw_reduce(sum_k a_k b_k) is
// (sum_k a_k b_k) R^-1 mod q = sum_k fq_mul(a_k, b_k) -- the same field element, and canonical, so the bytes do not change --
// while a term costs 64 multiply-adds instead of 96 and no conditional subtractions (the round kernels are VALU-issue bound).
// Bound: Montgomery reduction wants its argument below R q = 2^256 q; q^2 < 2^505, so at most SEVEN products of reduced
// operands may sit in an accumulator (7 q^2 < 2^508 < R q).
struct fq_wide {
  uint32_t v[16];
};
__device__ __forceinline__ void fqw_zero(fq_wide& w) {
#pragma unroll
  for (int i = 0; i < 16; i++) w.v[i] = 0;
}
// c += x (32 bits), carries into the upper words
__device__ __forceinline__ void acc_add32(acc96& c, uint32_t x) {
  unsigned cy = 0;
  uint32_t l = __builtin_addc((uint32_t)c.lo, x, 0u, &cy);
  uint32_t h = __builtin_addc((uint32_t)(c.lo >> 32), 0u, cy, &cy);
  c.hi = __builtin_addc(c.hi, 0u, cy, &cy);
  c.lo = (uint64_t)l | ((uint64_t)h << 32);
}
// w += a * b, product scanning: column k takes its limb products, the running carry and w's own limb
__device__ VPIN_MUL_INLINE void fqw_mac(fq_wide& w, const fq& a, const fq& b) {
  acc96 c{0, 0};
#define VPIN_FQW_COL(k, I0, I1)            \
  mac_column<k, I0, I1>(c, a.v, b.v);      \
  acc_add32(c, w.v[k]);                    \
  w.v[k] = (uint32_t)c.lo;                 \
  acc_shift(c);
  VPIN_FQW_COL(0, 0, 0) VPIN_FQW_COL(1, 0, 1) VPIN_FQW_COL(2, 0, 2) VPIN_FQW_COL(3, 0, 3)
  VPIN_FQW_COL(4, 0, 4) VPIN_FQW_COL(5, 0, 5) VPIN_FQW_COL(6, 0, 6) VPIN_FQW_COL(7, 0, 7)
  VPIN_FQW_COL(8, 1, 7) VPIN_FQW_COL(9, 2, 7) VPIN_FQW_COL(10, 3, 7) VPIN_FQW_COL(11, 4, 7)
  VPIN_FQW_COL(12, 5, 7) VPIN_FQW_COL(13, 6, 7) VPIN_FQW_COL(14, 7, 7)
#undef VPIN_FQW_COL
  acc_add32(c, w.v[15]);  // column 15: carries only; the sum stays below 2^512 (see the bound above)
  w.v[15] = (uint32_t)c.lo;
}
// w R^-1 mod q, canonical (w < R q): fq_mul's interleaved reduction with w's limbs in place of the a*b columns
__device__ VPIN_MUL_INLINE fq fqw_reduce(const fq_wide& w) {
  acc96 c{0, 0};
  uint32_t m[8];
  fq r;
  const uint32_t q0 = VPIN_Q0, q1 = VPIN_Q1, q2 = VPIN_Q2, q3 = VPIN_Q3;
#define VPIN_M(i) m[(i) < 0 ? 0 : (i) > 7 ? 7 : (i)]
#define VPIN_FQW_LOW(k)                                                                     \
  acc_add32(c, w.v[k]);                                                                      \
  if (k >= 2) mac2(c, VPIN_M(k - 1), q1, VPIN_M(k - 2), q2);                                  \
  else if (k == 1) mac1(c, VPIN_M(0), q1);                                                   \
  if (k >= 3) mac1(c, VPIN_M(k - 3), q3);                                                    \
  if (k >= 7) acc_add64(c, (uint64_t)VPIN_M(k - 7) << 28);                                   \
  m[k] = (uint32_t)c.lo * VPIN_QINV32;                                                       \
  mac1(c, m[k], q0);                                                                         \
  acc_shift(c);
  VPIN_FQW_LOW(0) VPIN_FQW_LOW(1) VPIN_FQW_LOW(2) VPIN_FQW_LOW(3) VPIN_FQW_LOW(4) VPIN_FQW_LOW(5) VPIN_FQW_LOW(6) VPIN_FQW_LOW(7)
#undef VPIN_FQW_LOW
#define VPIN_FQW_HIGH(k)                                                                    \
  acc_add32(c, w.v[k]);                                                                      \
  if (k == 8) mac2(c, VPIN_M(7), q1, VPIN_M(6), q2);                                          \
  if (k == 9) mac1(c, VPIN_M(7), q2);                                                        \
  if (k <= 10) mac1(c, VPIN_M(k - 3), q3);                                                   \
  acc_add64(c, (uint64_t)VPIN_M(k - 7) << 28);                                               \
  r.v[k - 8] = (uint32_t)c.lo;                                                               \
  acc_shift(c);
  VPIN_FQW_HIGH(8) VPIN_FQW_HIGH(9) VPIN_FQW_HIGH(10) VPIN_FQW_HIGH(11) VPIN_FQW_HIGH(12) VPIN_FQW_HIGH(13) VPIN_FQW_HIGH(14)
#undef VPIN_FQW_HIGH
#undef VPIN_M
  acc_add32(c, w.v[15]);
  r.v[7] = (uint32_t)c.lo;
  return fq_cond_sub_q(r);  // result < 2q
}

// ---- product with a launch-wide constant ------------------------------------------------------------------------
// The fold of a sum-check round multiplies every table difference by the same challenge r.  With
// T_i = r~ * 2^(32 i) * 2^-256 mod q (eight canonical constants, computed once per launch on the host), the
// Montgomery product is  r~ * d * 2^-256 = sum_i d_i * T_i  (mod q): 64 limb products with no interleaved
// reduction; the 288-bit sum S is brought into [0, q) with 2^252 = -(q - 2^252): S_lo - (S >> 252) * c, plus q on
// borrow.  The same element of F_q as fq_mul(r~, d), in canonical form: identical bits.  The constants sit in LDS,
// transposed (tt[k][i] = limb k of T_i): column k of the sum needs the eight words tt[k][0..7], two 128-bit
// broadcast reads, so they cost no registers between columns.
struct fq_const {
  uint32_t tt[8][8];
};

__device__ __forceinline__ fq fq_mul_const(const fq& d, const uint32_t (*tt)[8]) {
  acc96 c{0, 0};
  uint32_t s[9];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint4 lo = *reinterpret_cast<const uint4*>(&tt[k][0]);
    const uint4 hi = *reinterpret_cast<const uint4*>(&tt[k][4]);
    mac2(c, d.v[0], lo.x, d.v[1], lo.y);
    mac2(c, d.v[2], lo.z, d.v[3], lo.w);
    mac2(c, d.v[4], hi.x, d.v[5], hi.y);
    mac2(c, d.v[6], hi.z, d.v[7], hi.w);
    s[k] = (uint32_t)c.lo;
    acc_shift(c);
  }
  s[8] = (uint32_t)c.lo;  // S < 8 * 2^32 * q < 2^288
  const uint32_t h0 = (s[8] << 4) | (s[7] >> 28), h1 = s[8] >> 28;  // S >> 252 = h0 + h1 * 2^32, h1 < 16
  s[7] &= 0x0fffffffu;
  // P = (S >> 252) * c, c = q - 2^252 = {Q0, Q1, Q2, Q3}: below 2^161
  uint32_t P[6];
  uint64_t t = (uint64_t)h0 * VPIN_Q0;
  P[0] = (uint32_t)t;
  t = (uint64_t)h0 * VPIN_Q1 + (t >> 32); P[1] = (uint32_t)t;
  t = (uint64_t)h0 * VPIN_Q2 + (t >> 32); P[2] = (uint32_t)t;
  t = (uint64_t)h0 * VPIN_Q3 + (t >> 32); P[3] = (uint32_t)t;
  P[4] = (uint32_t)(t >> 32);
  t = (uint64_t)h1 * VPIN_Q0 + P[1]; P[1] = (uint32_t)t;
  t = (uint64_t)h1 * VPIN_Q1 + P[2] + (t >> 32); P[2] = (uint32_t)t;
  t = (uint64_t)h1 * VPIN_Q2 + P[3] + (t >> 32); P[3] = (uint32_t)t;
  t = (uint64_t)h1 * VPIN_Q3 + P[4] + (t >> 32); P[4] = (uint32_t)t;
  P[5] = (uint32_t)(t >> 32);
  fq r;
  unsigned bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __builtin_subc(s[i], i < 6 ? P[i] : 0u, bw, &bw);
  fq o;
  unsigned cy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o.v[i] = __builtin_addc(r.v[i], bw ? fq_modulus_limb(i) : 0u, cy, &cy);
  return o;
}

// Montgomery form -> canonical integer (Scalar::to_bytes, ristretto255.rs:426-438):
// montgomery_reduce(a, 0) = a * R^-1 mod q
__device__ __forceinline__ fq fq_from_mont(const fq& a) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = a.v[i];
  t[8] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    VPIN_MONT_STEP(t, 0u);
  }
  fq r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return fq_cond_sub_q(r);
}

// ---- wave / block reductions (64-wide wavefront) ----------------------------------

__device__ __forceinline__ fq fq_shfl_xor(const fq& a, int mask) {
  fq r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl_xor(a.v[i], mask, 64);
  return r;
}

// sum over the 64 lanes of a wavefront; every lane gets the total
__device__ __forceinline__ fq fq_wave_sum(fq a) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) a = fq_add(a, fq_shfl_xor(a, off));
  return a;
}

}  // namespace vpin
