// host/curve.h -- host-side ristretto255 (RFC 9496) for the per-round Pedersen commitments
// the protocol glue computes between GPU kernels (<= 5 terms each: Spartan/src/sumcheck.rs:
// 655-758, Spartan/src/nizk/mod.rs) and for generator derivation
// (MultiCommitGens::new, Spartan/src/commitments.rs:20-38).  The reference gets all of this
// from curve25519-dalek through Spartan/src/group.rs.
#pragma once
#include <array>
#include <map>
#include <memory>
#include <mutex>
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

  // width-5 non-adjacent form of a canonical scalar (< 2^253): digits odd in [-15, 15] or 0, at most one non-zero among
  // five consecutive positions; returns the index of the top non-zero digit, -1 for zero
  static int wnaf5(const uint8_t s[32], int8_t naf[257]) {
    uint64_t x[5] = {0, 0, 0, 0, 0};
    memcpy(x, s, 32);
    int top = -1;
    memset(naf, 0, 257);
    for (int pos = 0; pos < 257;) {
      const int w = pos >> 6, b = pos & 63;
      if (!(x[0] | x[1] | x[2] | x[3] | x[4])) break;
      if (!((x[w] >> b) & 1)) { pos++; continue; }
      // the 5-bit window at pos (it may straddle two words)
      uint64_t bits = x[w] >> b;
      if (b > 59 && w < 4) bits |= x[w + 1] << (64 - b);
      int d = (int)(bits & 31);
      if (d > 16) d -= 32;
      naf[pos] = (int8_t)d;
      top = pos;
      // subtract d * 2^pos: clear the window, and carry 2^(pos + 5) in when d was negative
      x[w] &= ~((uint64_t)31 << b);
      if (b > 59 && w < 4) x[w + 1] &= ~(((uint64_t)31) >> (64 - b));
      if (d < 0) {
        int cp = pos + 5;
        for (int cw = cp >> 6; cw < 5; cw++) {
          const uint64_t add = cw == (cp >> 6) ? (uint64_t)1 << (cp & 63) : 1;
          const uint64_t before = x[cw];
          x[cw] += add;
          if (x[cw] >= before) break;  // no carry out of this word
        }
      }
      pos += 5;
    }
    return top;
  }
  // odd multiples P, 3P, .., 15P as cached points
  void odd_multiples(Cached out[8]) const {
    out[0] = cached();
    const Point p2 = dbl();
    Point q = *this;
    for (int i = 1; i < 8; i++) { q = q.add(p2.cached()); out[i] = q.cached(); }
  }
  // variable-base (and variable-time) scalar multiplication, scalar as canonical little-endian bytes: 255 doublings and
  // ~43 additions (the verifier's sigma-protocol checks are a few hundred of these per proof)
  Point mul_bytes(const uint8_t s[32]) const {
    int8_t naf[257];
    const int top = wnaf5(s, naf);
    if (top < 0) return identity();
    Cached odd[8];
    odd_multiples(odd);
    Point acc = identity();
    for (int i = top; i >= 0; i--) {
      if (i != top) acc = acc.dbl();
      const int d = naf[i];
      if (d > 0) acc = acc.add(odd[d >> 1]);
      else if (d < 0) acc = acc.add(odd[(-d) >> 1], true);
    }
    return acc;
  }
  // a * P + b * Q with shared doublings (Straus)
  static Point mul2_bytes(const uint8_t a[32], const Point& P, const uint8_t b[32], const Point& Q) {
    int8_t na[257], nb[257];
    const int ta = wnaf5(a, na), tb = wnaf5(b, nb);
    const int top = ta > tb ? ta : tb;
    if (top < 0) return identity();
    Cached oa[8], ob[8];
    P.odd_multiples(oa);
    Q.odd_multiples(ob);
    Point acc = identity();
    for (int i = top; i >= 0; i--) {
      if (i != top) acc = acc.dbl();
      const int da = na[i], db = nb[i];
      if (da > 0) acc = acc.add(oa[da >> 1]);
      else if (da < 0) acc = acc.add(oa[(-da) >> 1], true);
      if (db > 0) acc = acc.add(ob[db >> 1]);
      else if (db < 0) acc = acc.add(ob[(-db) >> 1], true);
    }
    return acc;
  }
  static Point mul2(const Fq& a, const Point& P, const Fq& b, const Point& Q) {
    uint8_t ab[32], bb[32];
    a.to_bytes(ab);
    b.to_bytes(bb);
    return mul2_bytes(ab, P, bb, Q);
  }
  Point mul(const Fq& s) const { uint8_t b[32]; s.to_bytes(b); return mul_bytes(b); }
};

// Fixed-base signed-window table for one generator: 26 windows of 10 bits, 512 AFFINE multiples each as
// (y+x, y-x, 2dxy) -- 1.6 MB per generator, 26 mixed additions of 7 products per scalar.  The per-round commitments of
// the ZK sum-checks (8 scalar multiplications per round, sumcheck.rs:655-758) are the host's share of a round and, for
// instances below ~2^20 constraints, longer than the round's kernel.
struct Niels { Fe ypx, ymx, xy2d; };

struct FixedBase {
  static constexpr int kBits = 10, kWindows = 26, kEntries = 1 << (kBits - 1);
  std::shared_ptr<const std::vector<Niels>> tab;  // [kWindows][kEntries]; shared by every FixedBase of the same point
  const Niels* t = nullptr;
  Point base;
  FixedBase() {}
  // The table is a function of the point alone and costs ~3 ms to build: one per process and point (prover and verifier
  // of one CLI run, every context of a service use the same few generators of gens_r1cs_sat / gens_r1cs_eval).
  explicit FixedBase(const Point& p) : base(p) {
    static std::mutex mu;
    static std::map<std::array<uint8_t, 32>, std::shared_ptr<const std::vector<Niels>>> cache;
    std::array<uint8_t, 32> key;
    p.compress(key.data());
    {
      std::lock_guard<std::mutex> lock(mu);
      auto it = cache.find(key);
      if (it != cache.end()) tab = it->second;
    }
    if (!tab) {
      auto fresh = build(p);
      std::lock_guard<std::mutex> lock(mu);
      auto& slot = cache[key];
      if (!slot) slot = fresh;
      tab = slot;
    }
    t = tab->data();
  }
  static std::shared_ptr<const std::vector<Niels>> build(const Point& p) {
    auto out = std::make_shared<std::vector<Niels>>((size_t)kWindows * kEntries);
    std::vector<Niels>& tt = *out;
    std::vector<Point> m((size_t)kWindows * kEntries);
    Point s = p;
    for (int w = 0; w < kWindows; w++) {
      Cached sc = s.cached();
      Point q = s;
      m[(size_t)w * kEntries] = s;
      for (int k = 1; k < kEntries; k++) { q = q.add(sc); m[(size_t)w * kEntries + k] = q; }
      for (int k = 0; k < kBits; k++) s = s.dbl();
    }
    // one inversion for all Z (Montgomery's trick)
    std::vector<Fe> pre(m.size());
    Fe acc = Fe::one();
    for (size_t i = 0; i < m.size(); i++) { pre[i] = acc; acc = acc * m[i].Z; }
    acc = acc.invert();
    for (size_t i = m.size(); i-- > 0;) {
      Fe zi = acc * pre[i];
      acc = acc * m[i].Z;
      Fe x = m[i].X * zi, y = m[i].Y * zi;
      tt[i] = Niels{y + x, y - x, x * y * K().d2};
    }
    return out;
  }
  static Point add_niels(const Point& p, const Niels& q, bool negq) {
    Fe PP = p.Y.add_lazy(p.X) * (negq ? q.ymx : q.ypx), MM = p.Y.sub_lazy(p.X) * (negq ? q.ypx : q.ymx);
    Fe TT = p.T * q.xy2d, ZZ2 = p.Z.add_lazy(p.Z);
    Fe E = PP.sub_lazy(MM), H = PP.add_lazy(MM);
    Fe G = negq ? ZZ2.sub_lazy(TT) : ZZ2.add_lazy(TT), F = negq ? ZZ2.add_lazy(TT) : ZZ2.sub_lazy(TT);
    return Point{E * F, G * H, F * G, E * H};
  }
  // acc += s * base
  void mul_acc(Point& acc, const Fq& s) const {
    if (s.is_zero()) return;
    uint8_t b[40] = {0};
    s.to_bytes(b);
    const Niels* e[kWindows + 1];
    bool neg[kWindows + 1];
    int n = 0;
    unsigned carry = 0;
    for (int w = 0; w < kWindows; w++) {
      const int off = w * kBits;
      uint32_t word;
      memcpy(&word, b + (off >> 3), 4);
      unsigned v = ((word >> (off & 7)) & ((1u << kBits) - 1u)) + carry;
      const bool ng = v > (unsigned)kEntries;
      const unsigned mag = ng ? (1u << kBits) - v : v;
      carry = ng ? 1 : 0;
      if (mag) {
        e[n] = &t[(size_t)w * kEntries + mag - 1];
        neg[n] = ng;
        __builtin_prefetch(e[n]);
        __builtin_prefetch(reinterpret_cast<const char*>(e[n]) + 64);
        n++;
      }
    }
    // a canonical scalar is below 2^253 < 2^(10*26 - 1): no carry leaves the top window
    for (int i = 0; i < n; i++) acc = add_niels(acc, *e[i], neg[i]);
  }
  Point mul(const Fq& s) const { Point a = Point::identity(); mul_acc(a, s); return a; }
};

}  // namespace vpin_host
