// fp10_dev.h -- GF(2^255-19) in ten limbs of 26/25 bits for the inner loop of the row-commitment MSM (gfx950).
//
// fp_dev.h keeps a field element in eight 32-bit limbs; its product is 64 v_mad_u64_u32, each followed by a v_addc_co_u32
// that folds the carry-out into the 96-bit column accumulator, plus the column shifts: 197 VALU instructions, of which
// only 72 multiply.  Here a column of the schoolbook product is at most ten products of < 2^60 each, so it fits a plain
// 64-bit accumulator and NO carry instruction follows a multiply-add: 101 v_mad_u64_u32 and 41 other instructions, and the
// additions / subtractions between products are ten independent 32-bit adds with no carry chain at all (reduction is
// deferred to the next product).  Measured on MI355X (tools/ubench_fpmul.hip, profiles/r03_ubench_fpmul.txt): 245.8 against
// 183.0 G dependent products/s.  Same field, same values: everything that leaves a kernel is converted back to the
// eight-limb form and canonicalised by the same encoders as before, so the bytes are unchanged.
//
// Layout (the one of the curve25519 reference implementations): value = sum v[i] * 2^ceil(25.5 i), even limbs nominally 26
// bits, odd limbs 25.  Bounds are tracked in units of "1x" = (even < 2^26, odd < 2^25 + 2^17):
//   fe10_mul(f, g): f <= 4x, g <= 3x  ->  1x.   Column bound: 10 * 2^28 * (19 * 3 * 2^26) = 2^63.15 < 2^64.
//   fe10_add: bounds add.   fe10_sub(a, b): b <= 1x (it is subtracted from the bias 2p, which is 2x)  ->  bound(a) + 2x.
#pragma once
#include "fp_dev.h"

namespace vpin {

struct fe10 {
  uint32_t v[10];
};

__device__ __forceinline__ fe10 fe10_zero() {
  fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = 0;
  return r;
}
__device__ __forceinline__ fe10 fe10_one() {
  fe10 r = fe10_zero();
  r.v[0] = 1;
  return r;
}

// any representative a < 2^256 (table entries are weakly reduced): limb 9 takes bits 230..255, up to 26 bits
__device__ __forceinline__ fe10 fe10_from_fp(const fp& a) {
  fe10 r;
  r.v[0] = a.v[0] & 0x3ffffffu;                                              // bits   0.. 25
  r.v[1] = __builtin_amdgcn_alignbit(a.v[1], a.v[0], 26) & 0x1ffffffu;       //       26.. 50
  r.v[2] = __builtin_amdgcn_alignbit(a.v[2], a.v[1], 19) & 0x3ffffffu;       //       51.. 76
  r.v[3] = __builtin_amdgcn_alignbit(a.v[3], a.v[2], 13) & 0x1ffffffu;       //       77..101
  r.v[4] = (a.v[3] >> 6);                                                    //      102..127 (26 bits: the rest of word 3)
  r.v[5] = a.v[4] & 0x1ffffffu;                                              //      128..152
  r.v[6] = __builtin_amdgcn_alignbit(a.v[5], a.v[4], 25) & 0x3ffffffu;       //      153..178
  r.v[7] = __builtin_amdgcn_alignbit(a.v[6], a.v[5], 19) & 0x1ffffffu;       //      179..203
  r.v[8] = __builtin_amdgcn_alignbit(a.v[7], a.v[6], 12) & 0x3ffffffu;       //      204..229
  r.v[9] = a.v[7] >> 6;                                                      //      230..255
  return r;
}

// limbs <= 4x -> eight 32-bit limbs, weakly reduced (< 2^256); the value mod p is kept
__device__ __forceinline__ fp fe10_to_fp(const fe10& a) {
  // carry to nominal widths first (values up to 2^28 per limb)
  uint32_t h[10];
#pragma unroll
  for (int i = 0; i < 10; i++) h[i] = a.v[i];
  uint32_t c;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bits = (i & 1) ? 25 : 26;
    c = h[i] >> bits;
    h[i] &= (1u << bits) - 1u;
    h[i + 1] += c;
  }
  c = h[9] >> 25;
  h[9] &= 0x1ffffffu;
  h[0] += 19u * c;  // c < 2^4: h[0] < 2^26 + 2^9, which the packing below absorbs (it adds with carries)
  fp r;
  uint64_t k;
  k = (uint64_t)h[0] + ((uint64_t)h[1] << 26);                                   // bits 0..51(+)
  r.v[0] = (uint32_t)k; k >>= 32;
  k += (uint64_t)h[2] << 19;                                                    // 51 - 32
  r.v[1] = (uint32_t)k; k >>= 32;
  k += (uint64_t)h[3] << 13;                                                    // 77 - 64
  r.v[2] = (uint32_t)k; k >>= 32;
  k += (uint64_t)h[4] << 6;                                                     // 102 - 96
  r.v[3] = (uint32_t)k; k >>= 32;
  k += (uint64_t)h[5] + ((uint64_t)h[6] << 25);                                 // 128 - 128, 153 - 128
  r.v[4] = (uint32_t)k; k >>= 32;
  k += (uint64_t)h[7] << 19;                                                    // 179 - 160
  r.v[5] = (uint32_t)k; k >>= 32;
  k += (uint64_t)h[8] << 12;                                                    // 204 - 192
  r.v[6] = (uint32_t)k; k >>= 32;
  k += (uint64_t)h[9] << 6;                                                     // 230 - 224
  r.v[7] = (uint32_t)k; k >>= 32;
  return fp_fold(r, (uint32_t)k);
}

__device__ __forceinline__ fe10 fe10_add(const fe10& a, const fe10& b) {
  fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}

// a - b + 2p, b <= 1x
__device__ __forceinline__ fe10 fe10_sub(const fe10& a, const fe10& b) {
  fe10 r;
  r.v[0] = a.v[0] + 0x7ffffdau - b.v[0];  // 2^27 - 38
#pragma unroll
  for (int i = 1; i < 10; i++) r.v[i] = a.v[i] + ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - b.v[i];  // 2^26 - 2, 2^27 - 2
  return r;
}

// f <= 4x, g <= 3x -> 1x
__device__ __forceinline__ fe10 fe10_mul(const fe10& f, const fe10& g) {
  uint32_t g19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    g19[i] = (g.v[i] << 4) + (g.v[i] << 1) + g.v[i];  // 19 g_i < 2^32
    f2[i] = f.v[i] << 1;
  }
  uint64_t h[10];
#pragma unroll
  for (int k = 0; k < 10; k++) {
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = (k - i + 10) % 10;
      const bool wrap = i + j >= 10;             // 2^255 = 19
      const bool dbl = (i & 1) && (j & 1);       // two odd limbs: their weights' product is twice the target limb's weight
      s += (uint64_t)(dbl ? f2[i] : f.v[i]) * (wrap ? g19[j] : g.v[j]);  // v_mad_u64_u32; the column stays below 2^64
    }
    h[k] = s;
  }
  uint64_t c;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bits = (i & 1) ? 25 : 26;
    c = h[i] >> bits;
    h[i] &= (1u << bits) - 1u;
    h[i + 1] += c;
  }
  c = h[9] >> 25;
  h[9] &= 0x1ffffffu;
  h[0] += c * 19u;
  c = h[0] >> 26;
  h[0] &= 0x3ffffffu;
  h[1] += c;
  fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = (uint32_t)h[i];
  return r;
}

// ---- points: extended coordinates in the ten-limb form, every coordinate 1x ---------------------------------------------

struct ge10 {
  fe10 X, Y, Z, T;
};

__device__ __forceinline__ ge10 ge10_identity() {
  ge10 r;
  r.X = fe10_zero(); r.Y = fe10_one(); r.Z = fe10_one(); r.T = fe10_zero();
  return r;
}

__device__ __forceinline__ ge_ext ge10_to_ext(const ge10& p) {
  ge_ext r;
  r.X = fe10_to_fp(p.X); r.Y = fe10_to_fp(p.Y); r.Z = fe10_to_fp(p.Z); r.T = fe10_to_fp(p.T);
  return r;
}

// second half of add-2008-hwcd-3, shared by the two additions below:
//   PP, MM, TT: 1x products; ZZ2: 2x.   E = PP - MM (3x), H = PP + MM (2x), S = ZZ2 + TT (3x), D = ZZ2 - TT (4x);
//   G, F = S, D (or D, S when the second operand is negated);  X3 = E F, Y3 = G H, Z3 = F G = S D, T3 = E H.
// Every product takes its wider factor first (<= 4x) and the narrower one second (<= 3x).
__device__ __forceinline__ ge10 ge10_add_tail(const fe10& PP, const fe10& MM, const fe10& TT, const fe10& ZZ2, bool negate_q) {
  const fe10 E = fe10_sub(PP, MM), H = fe10_add(PP, MM);
  const fe10 S = fe10_add(ZZ2, TT), D = fe10_sub(ZZ2, TT);
  ge10 r;
  fe10 G, F;
#pragma unroll
  for (int i = 0; i < 10; i++) { G.v[i] = negate_q ? D.v[i] : S.v[i]; F.v[i] = negate_q ? S.v[i] : D.v[i]; }
  r.X = fe10_mul(F, E);
  r.Y = fe10_mul(G, H);
  r.Z = fe10_mul(D, S);
  r.T = fe10_mul(E, H);
  return r;
}

// p + (affine table entry) or p - it: 7 products (ge_add_niels of fp_dev.h in the ten-limb form; the entry arrives in the
// 96-byte packed form and is unpacked here)
__device__ __forceinline__ ge10 ge10_add_niels(const ge10& p, const ge_niels& q, bool negate_q) {
  const fe10 qa = fe10_from_fp(negate_q ? q.ymx : q.ypx), qb = fe10_from_fp(negate_q ? q.ypx : q.ymx);
  const fe10 PP = fe10_mul(fe10_add(p.Y, p.X), qa);   // 2x * 1x
  const fe10 MM = fe10_mul(fe10_sub(p.Y, p.X), qb);   // 3x * 1x
  const fe10 TT = fe10_mul(p.T, fe10_from_fp(q.xy2d));
  return ge10_add_tail(PP, MM, TT, fe10_add(p.Z, p.Z), negate_q);
}

// p + (cached projective point): 8 products
__device__ __forceinline__ ge10 ge10_add_cached(const ge10& p, const ge_cached& q) {
  const fe10 PP = fe10_mul(fe10_add(p.Y, p.X), fe10_from_fp(q.YpX));
  const fe10 MM = fe10_mul(fe10_sub(p.Y, p.X), fe10_from_fp(q.YmX));
  const fe10 TT = fe10_mul(p.T, fe10_from_fp(q.T2d));
  const fe10 ZZ = fe10_mul(p.Z, fe10_from_fp(q.Z));
  return ge10_add_tail(PP, MM, TT, fe10_add(ZZ, ZZ), false);
}

// dbl-2008-hwcd in the ten-limb form: 4 squarings + 4 products; input 1x, output 1x
__device__ __forceinline__ ge10 ge10_double(const ge10& p) {
  const fe10 A = fe10_mul(p.X, p.X), B = fe10_mul(p.Y, p.Y), ZZ = fe10_mul(p.Z, p.Z);
  const fe10 C = fe10_add(ZZ, ZZ);                       // 2x
  const fe10 xy = fe10_add(p.X, p.Y);                    // 2x
  const fe10 S = fe10_mul(xy, xy);                       // (X+Y)^2, 1x
  const fe10 H = fe10_add(A, B);                         // 2x      (a = -1: H = -(A+B) up to the sign folded in below)
  const fe10 E = fe10_sub(H, S);                         // A + B - (X+Y)^2 = -2XY          (4x as fe10_sub's bound; S is 1x)
  const fe10 G = fe10_sub(A, B);                         // A - B                           (3x)
  const fe10 F = fe10_add(C, G);                         // C + G                           (5x: too wide as a first factor ...)
  // ... so bring F down before it multiplies: one product by one costs the same as any, a carry pass is cheaper
  fe10 Fr = F;
  {
    uint32_t c;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int bits = (i & 1) ? 25 : 26;
      c = Fr.v[i] >> bits;
      Fr.v[i] &= (1u << bits) - 1u;
      Fr.v[i + 1] += c;
    }
    c = Fr.v[9] >> 25;
    Fr.v[9] &= 0x1ffffffu;
    Fr.v[0] += 19u * c;                                  // Fr: 1x (+ a few bits on limb 0)
  }
  // with e = -E = 2XY, g = -G = B - A, f = -F... the signs: X3 = E F, Y3 = G H', Z3 = F G, T3 = E H' where H' = -(A+B)
  // for a = -1 (D = -A): E = (X+Y)^2 - A - B, G = D + B = B - A, F = G - C, H' = D - B = -(A + B).
  // Above: E_ = -E, G_ = -G, F_ = C + G_ = -F, H = -H'.  Products of two negated factors keep their sign:
  //   X3 = E F = E_ F_,  Y3 = G H' = G_ H,  Z3 = F G = F_ G_,  T3 = E H' = E_ H.
  ge10 r;
  r.X = fe10_mul(E, Fr);   // E 4x first, Fr 1x second
  r.Y = fe10_mul(G, H);    // 3x, 2x
  r.Z = fe10_mul(G, Fr);   // 3x, 1x
  r.T = fe10_mul(E, H);    // 4x, 2x
  return r;
}

__device__ __forceinline__ ge10 ge10_from_ext(const ge_ext& p) {
  ge10 r;
  r.X = fe10_from_fp(p.X); r.Y = fe10_from_fp(p.Y); r.Z = fe10_from_fp(p.Z); r.T = fe10_from_fp(p.T);
  return r;
}

// p + q, both extended in the ten-limb form, every coordinate 1x: 9 products (add-2008-hwcd-3 with 2d T2 computed here)
__device__ __forceinline__ ge10 ge10_add_ge10(const ge10& p, const ge10& q) {
  const fe10 PP = fe10_mul(fe10_add(p.Y, p.X), fe10_add(q.Y, q.X));   // 2x * 2x
  const fe10 MM = fe10_mul(fe10_sub(p.Y, p.X), fe10_sub(q.Y, q.X));   // 3x * 3x
  const fe10 TT = fe10_mul(p.T, fe10_mul(q.T, fe10_from_fp(FP_D2())));
  const fe10 ZZ = fe10_mul(p.Z, q.Z);
  return ge10_add_tail(PP, MM, TT, fe10_add(ZZ, ZZ), false);
}

}  // namespace vpin
