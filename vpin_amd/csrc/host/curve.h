// host/curve.h -- host-side ristretto255 (RFC 9496) for the per-round Pedersen commitments
// the protocol glue computes between GPU kernels (<= 5 terms each: Spartan/src/sumcheck.rs:
// 655-758, Spartan/src/nizk/mod.rs) and for generator derivation
// (MultiCommitGens::new, Spartan/src/commitments.rs:20-38).  The reference gets all of this
// from curve25519-dalek through Spartan/src/group.rs.
#pragma once
#include <vector>

#include "field.h"

namespace vpin_host {

struct Consts {
  Fe d, d2, sqrt_m1, invsqrt_a_minus_d, sqrt_ad_minus_one, one_minus_d_sq, d_minus_one_sq, bx, by;
  Consts() {
    auto H = [](std::initializer_list<uint8_t> b) { uint8_t t[32]; int i = 0; for (auto x : b) t[i++] = x; return Fe::from_bytes(t); };
    d = H({163,120,89,19,202,77,235,117,171,216,65,65,77,10,112,0,152,232,121,119,121,64,199,140,115,254,111,43,238,108,3,82});
    d2 = d + d;
    sqrt_m1 = H({176,160,14,74,39,27,238,196,120,228,47,173,6,24,67,47,167,215,251,61,153,0,77,43,11,223,193,79,128,36,131,43});
    invsqrt_a_minus_d = H({234,64,93,128,170,253,200,153,190,114,65,90,23,22,47,157,64,216,1,254,145,123,194,22,162,252,175,207,5,137,108,120});
    sqrt_ad_minus_one = H({27,46,123,73,160,246,151,126,189,84,120,27,12,142,157,175,253,209,245,49,201,252,60,15,172,72,131,43,191,49,105,55});
    one_minus_d_sq = Fe::one() - d.square();
    d_minus_one_sq = (d - Fe::one()).square();
    bx = H({26,213,37,143,96,45,86,201,178,167,37,149,96,199,44,105,92,220,214,253,49,226,164,192,254,83,110,205,211,54,105,33});
    by = H({88,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102});
  }
};
inline const Consts& K() { static const Consts k; return k; }

// SQRT_RATIO_M1 (RFC 9496 4.2)
inline bool sqrt_ratio_m1(Fe& out, const Fe& u, const Fe& v) {
  Fe v3 = v.square() * v, v7 = v3.square() * v;
  Fe r = (u * v3) * (u * v7).pow_p58();
  Fe check = v * r.square();
  Fe nu = u.neg(), nu_i = nu * K().sqrt_m1;
  bool correct = check.equals(u), flipped = check.equals(nu), flipped_i = check.equals(nu_i);
  if (flipped || flipped_i) r = r * K().sqrt_m1;
  out = r.abs();
  return correct || flipped;
}

struct Cached { Fe ypx, ymx, z, t2d; };

struct Point {
  Fe X, Y, Z, T;

  static Point identity() { return Point{Fe::zero(), Fe::one(), Fe::one(), Fe::zero()}; }
  static Point basepoint() { return Point{K().bx, K().by, Fe::one(), K().bx * K().by}; }

  Cached cached() const { return Cached{Y + X, Y - X, Z, T * K().d2}; }

  Point add(const Cached& q, bool negq = false) const {
    Fe PP = (Y + X) * (negq ? q.ymx : q.ypx), MM = (Y - X) * (negq ? q.ypx : q.ymx);
    Fe TT = T * q.t2d, ZZ = Z * q.z, ZZ2 = ZZ + ZZ;
    Fe E = PP - MM, H = PP + MM;
    Fe G = negq ? ZZ2 - TT : ZZ2 + TT, F = negq ? ZZ2 + TT : ZZ2 - TT;
    return Point{E * F, G * H, F * G, E * H};
  }
  Point operator+(const Point& o) const { return add(o.cached()); }
  Point operator-(const Point& o) const { return add(o.cached(), true); }
  Point dbl() const {
    Fe A = X.square(), B = Y.square(), C = Z.square();
    C = C + C;
    Fe D = A.neg(), E = (X + Y).square() - A - B, G = D + B, F = G - C, H = D - B;
    return Point{E * F, G * H, F * G, E * H};
  }
  bool equals(const Point& o) const { return (X * o.Y).equals(Y * o.X) || (Y * o.Y).equals(X * o.X); }

  // RistrettoPoint::compress (RFC 9496 4.3.2)
  void compress(uint8_t out[32]) const {
    Fe u1 = (Z + Y) * (Z - Y), u2 = X * Y, invsqrt;
    sqrt_ratio_m1(invsqrt, Fe::one(), u1 * u2.square());
    Fe den1 = invsqrt * u1, den2 = invsqrt * u2, z_inv = den1 * den2 * T;
    Fe ix0 = X * K().sqrt_m1, iy0 = Y * K().sqrt_m1, ench = den1 * K().invsqrt_a_minus_d;
    bool rotate = (T * z_inv).is_negative();
    Fe x = rotate ? iy0 : X, y = rotate ? ix0 : Y, den_inv = rotate ? ench : den2;
    if ((x * z_inv).is_negative()) y = y.neg();
    (den_inv * (Z - y)).abs().to_bytes(out);
  }
  // CompressedRistretto::decompress (RFC 9496 4.3.1)
  static bool decompress(Point& out, const uint8_t in[32]) {
    Fe s = Fe::from_bytes(in);
    uint8_t chk[32];
    s.to_bytes(chk);
    if (memcmp(chk, in, 32) != 0 || (in[0] & 1)) return false;
    Fe one = Fe::one(), ss = s.square(), u1 = one - ss, u2 = one + ss, u2s = u2.square();
    Fe v = (K().d * u1.square()).neg() - u2s, invsqrt;
    bool sq = sqrt_ratio_m1(invsqrt, one, v * u2s);
    Fe den_x = invsqrt * u2, den_y = invsqrt * den_x * v;
    Fe x = ((s + s) * den_x).abs(), y = u1 * den_y, t = x * y;
    if (!sq || t.is_negative() || y.is_zero()) return false;
    out = Point{x, y, one, t};
    return true;
  }
  // RFC 9496 4.3.4 MAP
  static Point elligator(const Fe& t) {
    const Consts& k = K();
    Fe one = Fe::one(), m1 = one.neg();
    Fe r = k.sqrt_m1 * t.square();
    Fe u = (r + one) * k.one_minus_d_sq;
    Fe v = (m1 - r * k.d) * (r + k.d);
    Fe s;
    bool sq = sqrt_ratio_m1(s, u, v);
    Fe sp = (s * t).abs().neg();
    Fe c = m1;
    if (!sq) { s = sp; c = r; }
    Fe N = c * (r - one) * k.d_minus_one_sq - v;
    Fe w0 = (s + s) * v, w1 = N * k.sqrt_ad_minus_one, w2 = one - s.square(), w3 = one + s.square();
    return Point{w0 * w3, w2 * w1, w1 * w3, w0 * w2};
  }
  // RistrettoPoint::from_uniform_bytes (dalek 3.2.0)
  static Point from_uniform_bytes(const uint8_t b[64]) {
    return elligator(Fe::from_bytes(b)) + elligator(Fe::from_bytes(b + 32));
  }
  void to_xyzt(uint8_t out[128]) const { X.to_bytes(out); Y.to_bytes(out + 32); Z.to_bytes(out + 64); T.to_bytes(out + 96); }
  static Point from_xyzt(const uint8_t in[128]) {
    return Point{Fe::from_bytes(in), Fe::from_bytes(in + 32), Fe::from_bytes(in + 64), Fe::from_bytes(in + 96)};
  }

  // variable-base scalar multiplication, scalar as canonical little-endian bytes
  Point mul_bytes(const uint8_t s[32]) const {
    Point acc = identity();
    Cached me = cached();
    bool started = false;
    for (int i = 255; i >= 0; i--) {
      if (started) acc = acc.dbl();
      if ((s[i >> 3] >> (i & 7)) & 1) { acc = acc.add(me); started = true; }
    }
    return acc;
  }
  Point mul(const Fq& s) const { uint8_t b[32]; s.to_bytes(b); return mul_bytes(b); }
};

// Fixed-base 8-bit signed-window table for one generator: 32 x 128 cached multiples.
struct FixedBase {
  std::vector<Cached> t;  // [32][128]
  Point base;
  FixedBase() {}
  explicit FixedBase(const Point& p) : t(32 * 128), base(p) {
    Point s = p;
    for (int w = 0; w < 32; w++) {
      Cached sc = s.cached();
      Point q = s;
      t[w * 128] = sc;
      for (int k = 1; k < 128; k++) { q = q.add(sc); t[w * 128 + k] = q.cached(); }
      for (int k = 0; k < 8; k++) s = s.dbl();
    }
  }
  // acc += s * base
  void mul_acc(Point& acc, const Fq& s) const {
    if (s.is_zero()) return;
    uint8_t b[32];
    s.to_bytes(b);
    unsigned carry = 0;
    for (int w = 0; w < 32; w++) {
      unsigned v = b[w] + carry;
      bool neg = v > 128;
      unsigned mag = neg ? 256 - v : v;
      carry = neg ? 1 : 0;
      if (mag) acc = acc.add(t[w * 128 + mag - 1], neg);
    }
  }
  Point mul(const Fq& s) const { Point a = Point::identity(); mul_acc(a, s); return a; }
};

}  // namespace vpin_host
