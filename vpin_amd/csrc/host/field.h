// host/field.h -- host-side scalar field F_q and base field GF(2^255-19) for the protocol
// glue that stays on the CPU between GPU rounds (transcript challenges, univariate
// polynomials, the <=5-term Pedersen commitments of each sum-check round).
//
// Fq mirrors the reference's Scalar (Spartan/src/scalar/ristretto255.rs): 4 x u64 limbs,
// Montgomery form R = 2^256 -- the same 32 bytes that live in the device tables.
#pragma once
#include <cstdint>
#include <cstring>

namespace vpin_host {

typedef unsigned __int128 u128;

struct Fq {
  uint64_t l[4];

  static constexpr uint64_t M0 = 0x5812631a5cf5d3edULL, M1 = 0x14def9dea2f79cd6ULL, M2 = 0, M3 = 0x1000000000000000ULL;
  static constexpr uint64_t INV = 0xd2b51da312547e1bULL;

  static Fq zero() { return Fq{{0, 0, 0, 0}}; }
  static Fq one() { return Fq{{0xd6ec31748d98951dULL, 0xc6ef5bf4737dcf70ULL, 0xfffffffffffffffeULL, 0x0fffffffffffffffULL}}; }
  static Fq r2() { return Fq{{0xa40611e3449c0f01ULL, 0xd00e1ba768859347ULL, 0xceec73d217f5be65ULL, 0x0399411b7c309a3dULL}}; }
  static Fq r3() { return Fq{{0x2a9e49687b83a2dbULL, 0x278324e6aef7f3ecULL, 0x8065dc6c04ec5b65ULL, 0x0e530b773599cec7ULL}}; }

  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const Fq& o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }

  // conditional subtraction of q from a value < 2q
  static Fq reduce_once(uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3) {
    u128 d = (u128)a0 - M0;
    uint64_t r0 = (uint64_t)d;
    d = (u128)a1 - M1 - (uint64_t)((d >> 64) & 1);
    uint64_t r1 = (uint64_t)d;
    d = (u128)a2 - M2 - (uint64_t)((d >> 64) & 1);
    uint64_t r2_ = (uint64_t)d;
    d = (u128)a3 - M3 - (uint64_t)((d >> 64) & 1);
    uint64_t r3_ = (uint64_t)d;
    bool borrow = (d >> 64) & 1;
    return borrow ? Fq{{a0, a1, a2, a3}} : Fq{{r0, r1, r2_, r3_}};
  }

  Fq operator+(const Fq& o) const {
    u128 c = (u128)l[0] + o.l[0];
    uint64_t a0 = (uint64_t)c;
    c = (u128)l[1] + o.l[1] + (uint64_t)(c >> 64);
    uint64_t a1 = (uint64_t)c;
    c = (u128)l[2] + o.l[2] + (uint64_t)(c >> 64);
    uint64_t a2 = (uint64_t)c;
    c = (u128)l[3] + o.l[3] + (uint64_t)(c >> 64);
    return reduce_once(a0, a1, a2, (uint64_t)c);
  }

  Fq operator-(const Fq& o) const {
    u128 d = (u128)l[0] - o.l[0];
    uint64_t a0 = (uint64_t)d;
    d = (u128)l[1] - o.l[1] - (uint64_t)((d >> 64) & 1);
    uint64_t a1 = (uint64_t)d;
    d = (u128)l[2] - o.l[2] - (uint64_t)((d >> 64) & 1);
    uint64_t a2 = (uint64_t)d;
    d = (u128)l[3] - o.l[3] - (uint64_t)((d >> 64) & 1);
    uint64_t a3 = (uint64_t)d;
    if ((d >> 64) & 1) {
      u128 c = (u128)a0 + M0;
      a0 = (uint64_t)c;
      c = (u128)a1 + M1 + (uint64_t)(c >> 64);
      a1 = (uint64_t)c;
      c = (u128)a2 + M2 + (uint64_t)(c >> 64);
      a2 = (uint64_t)c;
      a3 = a3 + M3 + (uint64_t)(c >> 64);
    }
    return Fq{{a0, a1, a2, a3}};
  }

  Fq neg() const { return zero() - *this; }

  static Fq mont_reduce(uint64_t r[8]) {
    static const uint64_t M[4] = {M0, M1, M2, M3};
    uint64_t carry2 = 0;
    for (int i = 0; i < 4; i++) {
      uint64_t k = r[i] * INV;
      u128 c = (u128)k * M[0] + r[i];
      for (int j = 1; j < 4; j++) {
        c = (u128)k * M[j] + r[i + j] + (uint64_t)(c >> 64);
        r[i + j] = (uint64_t)c;
      }
      c = (u128)r[i + 4] + carry2 + (uint64_t)(c >> 64);
      r[i + 4] = (uint64_t)c;
      carry2 = (uint64_t)(c >> 64);
    }
    return reduce_once(r[4], r[5], r[6], r[7]);
  }

  Fq operator*(const Fq& o) const {
    uint64_t r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      uint64_t carry = 0;
      for (int j = 0; j < 4; j++) {
        u128 c = (u128)l[i] * o.l[j] + r[i + j] + carry;
        r[i + j] = (uint64_t)c;
        carry = (uint64_t)(c >> 64);
      }
      r[i + 4] = carry;
    }
    return mont_reduce(r);
  }

  Fq square() const { return *this * *this; }

  static Fq from_u64(uint64_t v) { return Fq{{v, 0, 0, 0}} * r2(); }

  // Scalar::to_bytes (ristretto255.rs:426-438): canonical little-endian
  void to_bytes(uint8_t out[32]) const {
    uint64_t r[8] = {l[0], l[1], l[2], l[3], 0, 0, 0, 0};
    Fq t = mont_reduce(r);
    memcpy(out, t.l, 32);
  }
  // Scalar::from_bytes_wide (ristretto255.rs:442-473)
  static Fq from_bytes_wide(const uint8_t b[64]) {
    Fq d0, d1;
    memcpy(d0.l, b, 32);
    memcpy(d1.l, b + 32, 32);
    return d0 * r2() + d1 * r3();
  }
  // a^(q-2); zero maps to zero like dalek's Scalar::invert
  Fq invert() const {
    static const uint64_t e[4] = {0x5812631a5cf5d3ebULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL};
    Fq res = one();
    for (int w = 3; w >= 0; w--)
      for (int i = 63; i >= 0; i--) {
        res = res.square();
        if ((e[w] >> i) & 1) res = res * *this;
      }
    return res;
  }
};

static_assert(sizeof(Fq) == 32, "Fq must be the 32-byte table element");

// ---- GF(2^255 - 19), five 51-bit limbs ------------------------------------------------------

struct Fe {
  uint64_t l[5];
  static constexpr uint64_t MASK = (1ULL << 51) - 1;

  static Fe zero() { return Fe{{0, 0, 0, 0, 0}}; }
  static Fe one() { return Fe{{1, 0, 0, 0, 0}}; }

  void carry() {
    for (int k = 0; k < 2; k++) {
      uint64_t c;
      c = l[0] >> 51; l[0] &= MASK; l[1] += c;
      c = l[1] >> 51; l[1] &= MASK; l[2] += c;
      c = l[2] >> 51; l[2] &= MASK; l[3] += c;
      c = l[3] >> 51; l[3] &= MASK; l[4] += c;
      c = l[4] >> 51; l[4] &= MASK; l[0] += 19 * c;
    }
  }
  Fe operator+(const Fe& o) const {
    Fe r;
    for (int i = 0; i < 5; i++) r.l[i] = l[i] + o.l[i];
    r.carry();
    return r;
  }
  Fe operator-(const Fe& o) const {
    // + 4p keeps every limb non-negative for carried inputs
    Fe r;
    r.l[0] = l[0] + 4 * (MASK - 18) - o.l[0];
    for (int i = 1; i < 5; i++) r.l[i] = l[i] + 4 * MASK - o.l[i];
    r.carry();
    return r;
  }
  Fe neg() const { return zero() - *this; }
  // Uncarried sum / difference for operands that go straight into a product: limbs stay below 2^54 when both inputs are
  // carried values or single lazy sums, and operator* takes limbs up to 2^54 (5 x 2^54 x 19 x 2^54 < 2^128).
  Fe add_lazy(const Fe& o) const {
    Fe r;
    for (int i = 0; i < 5; i++) r.l[i] = l[i] + o.l[i];
    return r;
  }
  Fe sub_lazy(const Fe& o) const {  // + 4p; o's limbs must be below 2^53
    Fe r;
    r.l[0] = l[0] + 4 * (MASK - 18) - o.l[0];
    for (int i = 1; i < 5; i++) r.l[i] = l[i] + 4 * MASK - o.l[i];
    return r;
  }
  Fe operator*(const Fe& o) const {
    const uint64_t* x = l;
    const uint64_t* y = o.l;
    uint64_t y1 = 19 * y[1], y2 = 19 * y[2], y3 = 19 * y[3], y4 = 19 * y[4];
    u128 t0 = (u128)x[0] * y[0] + (u128)x[1] * y4 + (u128)x[2] * y3 + (u128)x[3] * y2 + (u128)x[4] * y1;
    u128 t1 = (u128)x[0] * y[1] + (u128)x[1] * y[0] + (u128)x[2] * y4 + (u128)x[3] * y3 + (u128)x[4] * y2;
    u128 t2 = (u128)x[0] * y[2] + (u128)x[1] * y[1] + (u128)x[2] * y[0] + (u128)x[3] * y4 + (u128)x[4] * y3;
    u128 t3 = (u128)x[0] * y[3] + (u128)x[1] * y[2] + (u128)x[2] * y[1] + (u128)x[3] * y[0] + (u128)x[4] * y4;
    u128 t4 = (u128)x[0] * y[4] + (u128)x[1] * y[3] + (u128)x[2] * y[2] + (u128)x[3] * y[1] + (u128)x[4] * y[0];
    Fe r;
    t1 += (uint64_t)(t0 >> 51); r.l[0] = (uint64_t)t0 & MASK;
    t2 += (uint64_t)(t1 >> 51); r.l[1] = (uint64_t)t1 & MASK;
    t3 += (uint64_t)(t2 >> 51); r.l[2] = (uint64_t)t2 & MASK;
    t4 += (uint64_t)(t3 >> 51); r.l[3] = (uint64_t)t3 & MASK;
    uint64_t c = (uint64_t)(t4 >> 51);
    r.l[4] = (uint64_t)t4 & MASK;
    r.l[0] += 19 * c;
    c = r.l[0] >> 51; r.l[0] &= MASK; r.l[1] += c;
    return r;
  }
  Fe square() const {  // 15 products instead of 25
    const uint64_t* x = l;
    const uint64_t d0 = 2 * x[0], d1 = 2 * x[1], x3_19 = 19 * x[3], x4_19 = 19 * x[4];
    u128 t0 = (u128)x[0] * x[0] + (u128)d1 * x4_19 + (u128)(2 * x[2]) * x3_19;
    u128 t1 = (u128)d0 * x[1] + (u128)(2 * x[2]) * x4_19 + (u128)x[3] * x3_19;
    u128 t2 = (u128)d0 * x[2] + (u128)x[1] * x[1] + (u128)(2 * x[3]) * x4_19;
    u128 t3 = (u128)d0 * x[3] + (u128)d1 * x[2] + (u128)x[4] * x4_19;
    u128 t4 = (u128)d0 * x[4] + (u128)d1 * x[3] + (u128)x[2] * x[2];
    Fe r;
    t1 += (uint64_t)(t0 >> 51); r.l[0] = (uint64_t)t0 & MASK;
    t2 += (uint64_t)(t1 >> 51); r.l[1] = (uint64_t)t1 & MASK;
    t3 += (uint64_t)(t2 >> 51); r.l[2] = (uint64_t)t2 & MASK;
    t4 += (uint64_t)(t3 >> 51); r.l[3] = (uint64_t)t3 & MASK;
    uint64_t c = (uint64_t)(t4 >> 51);
    r.l[4] = (uint64_t)t4 & MASK;
    r.l[0] += 19 * c;
    c = r.l[0] >> 51; r.l[0] &= MASK; r.l[1] += c;
    return r;
  }
  Fe sqn(int n) const {
    Fe r = *this;
    for (int i = 0; i < n; i++) r = r.square();
    return r;
  }

  static Fe from_bytes(const uint8_t b[32]) {
    uint64_t w[4];
    memcpy(w, b, 32);
    Fe r;
    r.l[0] = w[0] & MASK;
    r.l[1] = ((w[0] >> 51) | (w[1] << 13)) & MASK;
    r.l[2] = ((w[1] >> 38) | (w[2] << 26)) & MASK;
    r.l[3] = ((w[2] >> 25) | (w[3] << 39)) & MASK;
    r.l[4] = (w[3] >> 12) & MASK;
    return r;
  }
  void to_bytes(uint8_t b[32]) const {
    Fe t = *this;
    t.carry();
    uint64_t q = (t.l[0] + 19) >> 51;
    q = (t.l[1] + q) >> 51; q = (t.l[2] + q) >> 51; q = (t.l[3] + q) >> 51; q = (t.l[4] + q) >> 51;
    t.l[0] += 19 * q;
    uint64_t c;
    c = t.l[0] >> 51; t.l[0] &= MASK; t.l[1] += c;
    c = t.l[1] >> 51; t.l[1] &= MASK; t.l[2] += c;
    c = t.l[2] >> 51; t.l[2] &= MASK; t.l[3] += c;
    c = t.l[3] >> 51; t.l[3] &= MASK; t.l[4] += c;
    t.l[4] &= MASK;
    uint64_t w[4];
    w[0] = t.l[0] | (t.l[1] << 51);
    w[1] = (t.l[1] >> 13) | (t.l[2] << 38);
    w[2] = (t.l[2] >> 26) | (t.l[3] << 25);
    w[3] = (t.l[3] >> 39) | (t.l[4] << 12);
    memcpy(b, w, 32);
  }
  bool is_negative() const { uint8_t b[32]; to_bytes(b); return b[0] & 1; }
  bool equals(const Fe& o) const { uint8_t a[32], b[32]; to_bytes(a); o.to_bytes(b); return memcmp(a, b, 32) == 0; }
  bool is_zero() const { return equals(zero()); }
  Fe abs() const { return is_negative() ? neg() : *this; }

  // z^(p-2)
  Fe invert() const {
    Fe t = pow_p58().sqn(3);  // z^(2^255 - 24)
    return t * (square() * *this);  // * z^3 -> z^(2^255 - 21)
  }
  // z^(2^252 - 3)
  Fe pow_p58() const {
    const Fe& z = *this;
    Fe t0 = z.square();
    Fe t1 = z * t0.sqn(2);
    t0 = t0 * t1;
    t0 = t1 * t0.square();
    t0 = t0.sqn(5) * t0;
    t1 = t0.sqn(10) * t0;
    Fe t2 = t1.sqn(20) * t1;
    t1 = t2.sqn(10) * t0;
    t2 = t1.sqn(50) * t1;
    Fe t3 = t2.sqn(100) * t2;
    t1 = t3.sqn(50) * t1;
    return t1.sqn(2) * z;
  }
};

}  // namespace vpin_host
