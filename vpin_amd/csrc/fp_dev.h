// fp_dev.h -- device-side arithmetic in GF(2^255-19) and Edwards25519 / ristretto255 point
// operations for gfx950.  This is the group the reference reaches through
// Spartan/src/group.rs (curve25519-dalek RistrettoPoint); results cross the C ABI as
// canonical encodings, so the internal representation is free:
//   field element : eight 32-bit limbs, value in [0, 2^256) ("weakly reduced": any
//                   representative of the class mod p; canonicalised only when encoded)
//   point         : extended twisted Edwards (X:Y:Z:T), a = -1
//   table entry   : "cached" form (Y+X, Y-X, Z, 2dT) so a table add costs 8 multiplies
// 2^256 = 38 (mod p) makes reduction a multiply-by-38 fold; no Montgomery form needed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fq_dev.h"  // acc96 / mac_column: the shared multiply-accumulate building block

namespace vpin {

struct alignas(16) fp {
  uint32_t v[8];
};

__device__ __forceinline__ fp fp_zero() {
  fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = 0;
  return r;
}
__device__ __forceinline__ fp fp_one() {
  fp r = fp_zero();
  r.v[0] = 1;
  return r;
}

__device__ __forceinline__ fp fp_load(const fp* __restrict__ p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 lo = q[0], hi = q[1];
  fp r;
  r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
  r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
  return r;
}
__device__ __forceinline__ void fp_store(fp* __restrict__ p, const fp& a) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
  q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

// Carry chains are written with clang's add/sub-with-carry builtins: they lower to
// v_add_co_u32 / v_addc_co_u32 links (one VALU instruction per limb, the compiler fills the
// VCC wait states), where the earlier 64-bit-per-limb formulation cost ~100 instructions per add.

// fold a small carry c (value c * 2^256 = 38c) back in; result again in [0, 2^256)
__device__ __forceinline__ fp fp_fold(fp t, uint32_t c) {
  unsigned cy = 0;
  t.v[0] = __builtin_addc(t.v[0], c * 38u, 0u, &cy);
#pragma unroll
  for (int i = 1; i < 8; i++) t.v[i] = __builtin_addc(t.v[i], 0u, cy, &cy);
  // a second wrap leaves a value below 38*c in limb 0 and zeros above it: no chain needed
  t.v[0] += cy ? 38u : 0u;
  return t;
}

__device__ __forceinline__ fp fp_add(const fp& a, const fp& b) {
  fp t;
  unsigned cy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) t.v[i] = __builtin_addc(a.v[i], b.v[i], cy, &cy);
  return fp_fold(t, cy);
}

__device__ __forceinline__ fp fp_sub(const fp& a, const fp& b) {
  // a - b + 2^256*borrow, and -2^256 = -38: subtract 38 per borrow (twice at most; after a second
  // wrap the limbs above limb 0 are all ones and limb 0 >= 2^32 - 38, so it ends there)
  fp t;
  unsigned bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) t.v[i] = __builtin_subc(a.v[i], b.v[i], bw, &bw);
  unsigned b2 = 0;
  t.v[0] = __builtin_subc(t.v[0], bw ? 38u : 0u, 0u, &b2);
#pragma unroll
  for (int i = 1; i < 8; i++) t.v[i] = __builtin_subc(t.v[i], 0u, b2, &b2);
  t.v[0] -= b2 ? 38u : 0u;
  return t;
}

__device__ __forceinline__ fp fp_neg(const fp& a) { return fp_sub(fp_zero(), a); }

#ifndef VPIN_FPMUL_INLINE
#define VPIN_FPMUL_INLINE __forceinline__
#endif

// a*b mod p (weakly reduced): 8x8 product-scanning multiply on the asm MAC block (fq_dev.h),
// then the high half folds in with 2^256 = 38 (mod p)
__device__ VPIN_FPMUL_INLINE fp fp_mul(fp a, fp b) {
  acc96 c{0, 0};
  uint32_t t[16];
#define VPIN_FP_COL(k)                                                            \
  mac_column<k, (k > 7 ? k - 7 : 0), (k < 7 ? k : 7)>(c, a.v, b.v);                \
  t[k] = (uint32_t)c.lo;                                                           \
  acc_shift(c);
  VPIN_FP_COL(0) VPIN_FP_COL(1) VPIN_FP_COL(2) VPIN_FP_COL(3) VPIN_FP_COL(4) VPIN_FP_COL(5) VPIN_FP_COL(6) VPIN_FP_COL(7)
  VPIN_FP_COL(8) VPIN_FP_COL(9) VPIN_FP_COL(10) VPIN_FP_COL(11) VPIN_FP_COL(12) VPIN_FP_COL(13) VPIN_FP_COL(14)
#undef VPIN_FP_COL
  t[15] = (uint32_t)c.lo;
  fp r;
  uint64_t k = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    k += (uint64_t)t[8 + i] * 38u + t[i];
    r.v[i] = (uint32_t)k;
    k >>= 32;
  }
  return fp_fold(r, (uint32_t)k);
}

__device__ __forceinline__ fp fp_sqr(const fp& a) { return fp_mul(a, a); }

// multiply by a small constant (< 2^31)
__device__ __forceinline__ fp fp_mul_small(const fp& a, uint32_t s) {
  fp r;
  uint64_t k = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    k += (uint64_t)a.v[i] * s;
    r.v[i] = (uint32_t)k;
    k >>= 32;
  }
  return fp_fold(r, (uint32_t)k);
}

// canonical representative in [0, p)
__device__ __forceinline__ fp fp_freeze(fp a) {
  // a < 2^256 = 2p + 38: subtract p while a >= p (at most twice, plus the 19-gap)
#pragma unroll
  for (int round = 0; round < 3; round++) {
    // d = a - p = a + 19 - 2^255
    fp d;
    uint64_t k = 19;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      k += a.v[i];
      d.v[i] = (uint32_t)k;
      k >>= 32;
    }
    // (a + 19) as a 257-bit number: carry k and limbs d; a >= p  <=>  a + 19 >= 2^255
    bool ge = (k != 0) || (d.v[7] >> 31);
    // subtract 2^255: clear via arithmetic on the top
    uint32_t top = d.v[7] - 0x80000000u;  // valid when bit 255 set or carry set
    fp e = d;
    e.v[7] = top;
#pragma unroll
    for (int i = 0; i < 8; i++) a.v[i] = ge ? e.v[i] : a.v[i];
  }
  return a;
}

__device__ __forceinline__ bool fp_is_negative(const fp& a) { return fp_freeze(a).v[0] & 1; }
__device__ __forceinline__ bool fp_eq(const fp& a, const fp& b) {
  fp x = fp_freeze(a), y = fp_freeze(b);
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= x.v[i] ^ y.v[i];
  return o == 0;
}
__device__ __forceinline__ bool fp_is_zero(const fp& a) { return fp_eq(a, fp_zero()); }
__device__ __forceinline__ fp fp_select(bool c, const fp& a, const fp& b) {
  fp r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
__device__ __forceinline__ fp fp_abs(const fp& a) { return fp_select(fp_is_negative(a), fp_neg(a), a); }

// a^(2^n) by repeated squaring
__device__ __noinline__ fp fp_sqr_n(fp a, int n) {
  for (int i = 0; i < n; i++) a = fp_mul(a, a);
  return a;
}

// a^((p-5)/8) = a^(2^252 - 3), standard curve25519 addition chain
__device__ __noinline__ fp fp_pow_p58(const fp& z) {
  fp t0 = fp_sqr_n(z, 1);                 // 2
  fp t1 = fp_mul(z, fp_sqr_n(t0, 2));     // 9
  t0 = fp_mul(t0, t1);                    // 11
  t0 = fp_mul(t1, fp_sqr_n(t0, 1));       // 31 = 2^5 - 1
  t0 = fp_mul(fp_sqr_n(t0, 5), t0);       // 2^10 - 1
  t1 = fp_mul(fp_sqr_n(t0, 10), t0);      // 2^20 - 1
  fp t2 = fp_mul(fp_sqr_n(t1, 20), t1);   // 2^40 - 1
  t1 = fp_mul(fp_sqr_n(t2, 10), t0);      // 2^50 - 1
  t2 = fp_mul(fp_sqr_n(t1, 50), t1);      // 2^100 - 1
  fp t3 = fp_mul(fp_sqr_n(t2, 100), t2);  // 2^200 - 1
  t1 = fp_mul(fp_sqr_n(t3, 50), t1);      // 2^250 - 1
  return fp_mul(fp_sqr_n(t1, 2), z);      // 2^252 - 3
}

// a^(p-2) = a^(2^255-21) = (a^(2^252-3))^8 * a^3
__device__ __noinline__ fp fp_invert(const fp& a) {
  fp t = fp_sqr_n(fp_pow_p58(a), 3);
  return fp_mul(t, fp_mul(fp_sqr(a), a));
}

// ---- curve constants (little-endian 32-bit limbs), RFC 9496 section 4 ------------------------
__device__ __forceinline__ fp fp_const(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5,
                                       uint32_t a6, uint32_t a7) {
  fp r;
  r.v[0] = a0; r.v[1] = a1; r.v[2] = a2; r.v[3] = a3; r.v[4] = a4; r.v[5] = a5; r.v[6] = a6; r.v[7] = a7;
  return r;
}
// 2*d
__device__ __forceinline__ fp FP_D2() { return fp_const(0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u, 0x56dffce7u, 0x2406d9dcu); }
__device__ __forceinline__ fp FP_SQRT_M1() { return fp_const(0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u); }
__device__ __forceinline__ fp FP_INVSQRT_A_MINUS_D() { return fp_const(0x805d40eau, 0x99c8fdaau, 0x5a4172beu, 0x9d2f1617u, 0xfe01d840u, 0x16c27b91u, 0xcfaffca2u, 0x786c8905u); }

// ---- points ---------------------------------------------------------------------------------

struct ge_ext { fp X, Y, Z, T; };          // extended
struct ge_cached { fp YpX, YmX, Z, T2d; };  // (Y+X, Y-X, Z, 2dT)

__device__ __forceinline__ ge_ext ge_identity() {
  ge_ext r;
  r.X = fp_zero(); r.Y = fp_one(); r.Z = fp_one(); r.T = fp_zero();
  return r;
}

__device__ __forceinline__ ge_cached ge_to_cached(const ge_ext& p) {
  ge_cached c;
  c.YpX = fp_add(p.Y, p.X);
  c.YmX = fp_sub(p.Y, p.X);
  c.Z = p.Z;
  c.T2d = fp_mul(p.T, FP_D2());
  return c;
}

// extended + cached (add-2008-hwcd-3 with the second operand pre-arranged): 8 multiplies
__device__ __forceinline__ ge_ext ge_add_cached(const ge_ext& p, const ge_cached& q, bool negate_q = false) {
  fp qa = negate_q ? q.YmX : q.YpX, qb = negate_q ? q.YpX : q.YmX;
  fp PP = fp_mul(fp_add(p.Y, p.X), qa);
  fp MM = fp_mul(fp_sub(p.Y, p.X), qb);
  fp TT = fp_mul(p.T, q.T2d);
  fp ZZ = fp_mul(p.Z, q.Z);
  fp ZZ2 = fp_add(ZZ, ZZ);
  fp E = fp_sub(PP, MM), H = fp_add(PP, MM);
  fp G = negate_q ? fp_sub(ZZ2, TT) : fp_add(ZZ2, TT);
  fp F = negate_q ? fp_add(ZZ2, TT) : fp_sub(ZZ2, TT);
  ge_ext r;
  r.X = fp_mul(E, F); r.Y = fp_mul(G, H); r.Z = fp_mul(F, G); r.T = fp_mul(E, H);
  return r;
}

__device__ __forceinline__ ge_ext ge_add(const ge_ext& p, const ge_ext& q) { return ge_add_cached(p, ge_to_cached(q)); }

// Affine table entry (y+x, y-x, 2dxy): 96 bytes, and a table add costs 7 multiplies (Z2 = 1)
struct ge_niels { fp ypx, ymx, xy2d; };

__device__ __forceinline__ ge_ext ge_add_niels(const ge_ext& p, const ge_niels& q, bool negate_q) {
  fp qa = negate_q ? q.ymx : q.ypx, qb = negate_q ? q.ypx : q.ymx;
  fp PP = fp_mul(fp_add(p.Y, p.X), qa);
  fp MM = fp_mul(fp_sub(p.Y, p.X), qb);
  fp TT = fp_mul(p.T, q.xy2d);
  fp ZZ2 = fp_add(p.Z, p.Z);
  fp E = fp_sub(PP, MM), H = fp_add(PP, MM);
  fp G = negate_q ? fp_sub(ZZ2, TT) : fp_add(ZZ2, TT);
  fp F = negate_q ? fp_add(ZZ2, TT) : fp_sub(ZZ2, TT);
  ge_ext r;
  r.X = fp_mul(E, F); r.Y = fp_mul(G, H); r.Z = fp_mul(F, G); r.T = fp_mul(E, H);
  return r;
}

// dbl-2008-hwcd
__device__ __forceinline__ ge_ext ge_double(const ge_ext& p) {
  fp A = fp_sqr(p.X), B = fp_sqr(p.Y), C = fp_sqr(p.Z);
  C = fp_add(C, C);
  fp D = fp_neg(A);
  fp xy = fp_add(p.X, p.Y);
  fp E = fp_sub(fp_sub(fp_sqr(xy), A), B);
  fp G = fp_add(D, B), F = fp_sub(G, C), H = fp_sub(D, B);
  ge_ext r;
  r.X = fp_mul(E, F); r.Y = fp_mul(G, H); r.Z = fp_mul(F, G); r.T = fp_mul(E, H);
  return r;
}

// RFC 9496 4.2 SQRT_RATIO_M1 specialised to u = 1 (all the encoder needs)
__device__ __forceinline__ fp fp_invsqrt(const fp& v, bool* was_square) {
  fp v3 = fp_mul(fp_sqr(v), v);
  fp v7 = fp_mul(fp_sqr(v3), v);
  fp r = fp_mul(v3, fp_pow_p58(v7));
  fp check = fp_mul(v, fp_sqr(r));
  fp one = fp_one(), m1 = fp_neg(one);
  fp m_i = fp_mul(m1, FP_SQRT_M1());
  bool correct = fp_eq(check, one), flipped = fp_eq(check, m1), flipped_i = fp_eq(check, m_i);
  fp r_prime = fp_mul(r, FP_SQRT_M1());
  r = fp_select(flipped || flipped_i, r_prime, r);
  *was_square = correct || flipped;
  return fp_abs(r);
}

// RistrettoPoint::compress (RFC 9496 4.3.2) -> canonical little-endian limbs of s
__device__ __noinline__ fp ge_compress(const ge_ext& p) {
  fp u1 = fp_mul(fp_add(p.Z, p.Y), fp_sub(p.Z, p.Y));
  fp u2 = fp_mul(p.X, p.Y);
  bool sq;
  fp invsqrt = fp_invsqrt(fp_mul(u1, fp_sqr(u2)), &sq);
  fp den1 = fp_mul(invsqrt, u1), den2 = fp_mul(invsqrt, u2);
  fp z_inv = fp_mul(fp_mul(den1, den2), p.T);
  fp ix0 = fp_mul(p.X, FP_SQRT_M1()), iy0 = fp_mul(p.Y, FP_SQRT_M1());
  fp ench = fp_mul(den1, FP_INVSQRT_A_MINUS_D());
  bool rotate = fp_is_negative(fp_mul(p.T, z_inv));
  fp x = fp_select(rotate, iy0, p.X), y = fp_select(rotate, ix0, p.Y);
  fp den_inv = fp_select(rotate, ench, den2);
  y = fp_select(fp_is_negative(fp_mul(x, z_inv)), fp_neg(y), y);
  fp s = fp_abs(fp_mul(den_inv, fp_sub(p.Z, y)));
  return fp_freeze(s);
}

}  // namespace vpin
