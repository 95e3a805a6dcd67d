/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see fq.h).
 * CPU restatement of Spartan/src/scalar/ristretto255.rs (F_q, Montgomery 4xu64).
 */
#include "fq.h"
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

const fq_t FQ_MODULUS = {{0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0x0ULL, 0x1000000000000000ULL}};
const fq_t FQ_R = {{0xd6ec31748d98951dULL, 0xc6ef5bf4737dcf70ULL, 0xfffffffffffffffeULL, 0x0fffffffffffffffULL}};
const fq_t FQ_R2 = {{0xa40611e3449c0f01ULL, 0xd00e1ba768859347ULL, 0xceec73d217f5be65ULL, 0x0399411b7c309a3dULL}};
const fq_t FQ_R3 = {{0x2a9e49687b83a2dbULL, 0x278324e6aef7f3ecULL, 0x8065dc6c04ec5b65ULL, 0x0e530b773599cec7ULL}};

/* ristretto255.rs:21-38 */
static inline uint64_t adc(uint64_t a, uint64_t b, uint64_t carry, uint64_t *co) {
  u128 r = (u128)a + b + carry;
  *co = (uint64_t)(r >> 64);
  return (uint64_t)r;
}
static inline uint64_t sbb(uint64_t a, uint64_t b, uint64_t borrow, uint64_t *bo) {
  u128 r = (u128)a - ((u128)b + (borrow >> 63));
  *bo = (uint64_t)(r >> 64);
  return (uint64_t)r;
}
static inline uint64_t mac(uint64_t a, uint64_t b, uint64_t c, uint64_t carry, uint64_t *co) {
  u128 r = (u128)a + (u128)b * c + carry;
  *co = (uint64_t)(r >> 64);
  return (uint64_t)r;
}

fq_t fq_zero(void) { fq_t z = {{0, 0, 0, 0}}; return z; }
fq_t fq_one(void) { return FQ_R; }

fq_t fq_sub(const fq_t *a, const fq_t *b) {
  uint64_t bw = 0, c = 0;
  fq_t d;
  d.l[0] = sbb(a->l[0], b->l[0], 0, &bw);
  d.l[1] = sbb(a->l[1], b->l[1], bw, &bw);
  d.l[2] = sbb(a->l[2], b->l[2], bw, &bw);
  d.l[3] = sbb(a->l[3], b->l[3], bw, &bw);
  /* bw is all-ones on underflow: conditionally add the modulus back */
  d.l[0] = adc(d.l[0], FQ_MODULUS.l[0] & bw, 0, &c);
  d.l[1] = adc(d.l[1], FQ_MODULUS.l[1] & bw, c, &c);
  d.l[2] = adc(d.l[2], FQ_MODULUS.l[2] & bw, c, &c);
  d.l[3] = adc(d.l[3], FQ_MODULUS.l[3] & bw, c, &c);
  return d;
}

fq_t fq_add(const fq_t *a, const fq_t *b) {
  uint64_t c = 0;
  fq_t d;
  d.l[0] = adc(a->l[0], b->l[0], 0, &c);
  d.l[1] = adc(a->l[1], b->l[1], c, &c);
  d.l[2] = adc(a->l[2], b->l[2], c, &c);
  d.l[3] = adc(a->l[3], b->l[3], c, &c);
  return fq_sub(&d, &FQ_MODULUS);
}

fq_t fq_neg(const fq_t *a) {
  uint64_t bw = 0;
  fq_t d;
  d.l[0] = sbb(FQ_MODULUS.l[0], a->l[0], 0, &bw);
  d.l[1] = sbb(FQ_MODULUS.l[1], a->l[1], bw, &bw);
  d.l[2] = sbb(FQ_MODULUS.l[2], a->l[2], bw, &bw);
  d.l[3] = sbb(FQ_MODULUS.l[3], a->l[3], bw, &bw);
  uint64_t mask = (uint64_t)((a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0) - 1;
  for (int i = 0; i < 4; i++) d.l[i] &= mask;
  return d;
}

fq_t fq_montgomery_reduce(const uint64_t rin[8]) {
  uint64_t r[8];
  memcpy(r, rin, sizeof r);
  uint64_t carry2 = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t k = r[i] * FQ_INV, carry = 0;
    (void)mac(r[i], k, FQ_MODULUS.l[0], 0, &carry);
    r[i + 1] = mac(r[i + 1], k, FQ_MODULUS.l[1], carry, &carry);
    r[i + 2] = mac(r[i + 2], k, FQ_MODULUS.l[2], carry, &carry);
    r[i + 3] = mac(r[i + 3], k, FQ_MODULUS.l[3], carry, &carry);
    r[i + 4] = adc(r[i + 4], carry2, carry, &carry2);
  }
  fq_t t = {{r[4], r[5], r[6], r[7]}};
  return fq_sub(&t, &FQ_MODULUS);
}

fq_t fq_mul(const fq_t *a, const fq_t *b) {
  uint64_t r[8] = {0};
  for (int i = 0; i < 4; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < 4; j++) r[i + j] = mac(r[i + j], a->l[i], b->l[j], carry, &carry);
    r[i + 4] = carry;
  }
  return fq_montgomery_reduce(r);
}

fq_t fq_square(const fq_t *a) { return fq_mul(a, a); }

fq_t fq_from_u64(uint64_t v) {
  fq_t t = {{v, 0, 0, 0}};
  return fq_mul(&t, &FQ_R2);
}

fq_t fq_from_raw(const uint64_t v[4]) {
  fq_t t = {{v[0], v[1], v[2], v[3]}};
  return fq_mul(&t, &FQ_R2);
}

static uint64_t load64(const uint8_t *b) {
  uint64_t v = 0;
  for (int i = 7; i >= 0; i--) v = (v << 8) | b[i];
  return v;
}
static void store64(uint8_t *b, uint64_t v) {
  for (int i = 0; i < 8; i++) { b[i] = (uint8_t)v; v >>= 8; }
}

int fq_from_bytes(fq_t *out, const uint8_t b[32]) {
  fq_t t;
  for (int i = 0; i < 4; i++) t.l[i] = load64(b + 8 * i);
  uint64_t bw = 0;
  (void)sbb(t.l[0], FQ_MODULUS.l[0], 0, &bw);
  (void)sbb(t.l[1], FQ_MODULUS.l[1], bw, &bw);
  (void)sbb(t.l[2], FQ_MODULUS.l[2], bw, &bw);
  (void)sbb(t.l[3], FQ_MODULUS.l[3], bw, &bw);
  int is_some = (int)(bw & 1);
  *out = fq_mul(&t, &FQ_R2);
  return is_some;
}

void fq_to_bytes(uint8_t out[32], const fq_t *a) {
  uint64_t r[8] = {a->l[0], a->l[1], a->l[2], a->l[3], 0, 0, 0, 0};
  fq_t t = fq_montgomery_reduce(r);
  for (int i = 0; i < 4; i++) store64(out + 8 * i, t.l[i]);
}

fq_t fq_from_bytes_wide(const uint8_t b[64]) {
  fq_t d0, d1;
  for (int i = 0; i < 4; i++) { d0.l[i] = load64(b + 8 * i); d1.l[i] = load64(b + 32 + 8 * i); }
  fq_t x = fq_mul(&d0, &FQ_R2), y = fq_mul(&d1, &FQ_R3);
  return fq_add(&x, &y);
}

/* curve25519-dalek Scalar::from_bytes_mod_order: reduce a 256-bit LE integer
 * mod q (used by the gadgets, VP/point_mult.rs:374-375). d0*R2 Montgomery
 * reduction is valid for any 256-bit d0 (ristretto255.rs:455-467). */
fq_t fq_from_bytes_mod_order(const uint8_t b[32]) {
  fq_t d0;
  for (int i = 0; i < 4; i++) d0.l[i] = load64(b + 8 * i);
  return fq_mul(&d0, &FQ_R2);
}

fq_t fq_pow_vartime(const fq_t *a, const uint64_t by[4]) {
  fq_t res = fq_one();
  for (int e = 3; e >= 0; e--)
    for (int i = 63; i >= 0; i--) {
      res = fq_square(&res);
      if ((by[e] >> i) & 1) res = fq_mul(&res, a);
    }
  return res;
}

/* The reference uses an addition chain for a^(q-2) (ristretto255.rs:548-602);
 * the value is the unique field inverse (0 -> 0 under pow), so plain
 * square-and-multiply restates it exactly (ristretto255.rs:1166-1184 checks
 * invert == pow(q-2)). */
fq_t fq_invert(const fq_t *a) {
  static const uint64_t qm2[4] = {0x5812631a5cf5d3ebULL, 0x14def9dea2f79cd6ULL, 0x0ULL, 0x1000000000000000ULL};
  return fq_pow_vartime(a, qm2);
}

fq_t fq_batch_invert(fq_t *inputs, size_t n) {
  fq_t *scratch = (fq_t *)malloc(sizeof(fq_t) * (n ? n : 1));
  fq_t acc = fq_one();
  for (size_t i = 0; i < n; i++) { scratch[i] = acc; acc = fq_mul(&acc, &inputs[i]); }
  acc = fq_invert(&acc);
  fq_t ret = acc;
  for (size_t i = n; i-- > 0;) {
    fq_t tmp = fq_mul(&acc, &inputs[i]);
    inputs[i] = fq_mul(&acc, &scratch[i]);
    acc = tmp;
  }
  free(scratch);
  return ret;
}

int fq_eq(const fq_t *a, const fq_t *b) {
  return a->l[0] == b->l[0] && a->l[1] == b->l[1] && a->l[2] == b->l[2] && a->l[3] == b->l[3];
}
int fq_is_zero(const fq_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
