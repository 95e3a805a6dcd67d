/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see fq.h / group.h).
 * ristretto255 over edwards25519, restated from RFC 9496 and the Edwards extended
 * coordinate formulas; the reference binds it via Spartan/src/group.rs.
 */
#include "group.h"
#include "keccak.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
#define MASK51 ((1ULL << 51) - 1)

/* ------------------------------------------------------------------ GF(2^255-19) */

static const uint8_t K_D[32] = {163,120,89,19,202,77,235,117,171,216,65,65,77,10,112,0,152,232,121,119,121,64,199,140,115,254,111,43,238,108,3,82};
static const uint8_t K_SQRT_M1[32] = {176,160,14,74,39,27,238,196,120,228,47,173,6,24,67,47,167,215,251,61,153,0,77,43,11,223,193,79,128,36,131,43};
static const uint8_t K_INVSQRT_A_MINUS_D[32] = {234,64,93,128,170,253,200,153,190,114,65,90,23,22,47,157,64,216,1,254,145,123,194,22,162,252,175,207,5,137,108,120};
static const uint8_t K_SQRT_AD_MINUS_ONE[32] = {27,46,123,73,160,246,151,126,189,84,120,27,12,142,157,175,253,209,245,49,201,252,60,15,172,72,131,43,191,49,105,55};
static const uint8_t K_ONE_MINUS_D_SQ[32] = {118,193,95,148,193,9,124,226,15,53,94,205,56,161,129,44,228,223,112,190,221,171,148,153,215,224,179,178,168,114,144,2};
static const uint8_t K_D_MINUS_ONE_SQ[32] = {32,77,237,68,170,90,173,49,153,25,30,176,44,74,158,210,235,78,155,82,47,211,220,76,65,34,108,246,122,179,104,89};
static const uint8_t K_D2[32] = {89,241,178,38,148,155,214,235,86,177,131,130,154,20,224,0,48,209,243,238,242,128,142,25,231,252,223,86,220,217,6,36};
static const uint8_t K_BX[32] = {26,213,37,143,96,45,86,201,178,167,37,149,96,199,44,105,92,220,214,253,49,226,164,192,254,83,110,205,211,54,105,33};
static const uint8_t K_BY[32] = {88,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102,102};

static fe_t C_D, C_D2, C_SQRT_M1, C_INVSQRT_A_MINUS_D, C_SQRT_AD_MINUS_ONE, C_ONE_MINUS_D_SQ, C_D_MINUS_ONE_SQ;
static ge_t C_B;
static int consts_ready = 0;

static void fe_0(fe_t *r) { memset(r, 0, sizeof *r); }
static void fe_1(fe_t *r) { fe_0(r); r->l[0] = 1; }

/* one pass: limbs 1..4 below 2^51, limb 0 below 2^51 + 19 * 2^13 (inputs with limbs below 2^64 / 19) */
static inline void fe_carry1(fe_t *r) {
  uint64_t c;
  c = r->l[0] >> 51; r->l[0] &= MASK51; r->l[1] += c;
  c = r->l[1] >> 51; r->l[1] &= MASK51; r->l[2] += c;
  c = r->l[2] >> 51; r->l[2] &= MASK51; r->l[3] += c;
  c = r->l[3] >> 51; r->l[3] &= MASK51; r->l[4] += c;
  c = r->l[4] >> 51; r->l[4] &= MASK51; r->l[0] += 19 * c;
}
static void fe_carry(fe_t *r) { fe_carry1(r); fe_carry1(r); }

/* sums and differences are carried once: every limb stays below 2^52, which fe_mul / fe_sq / fe_sub accept */
static void fe_add(fe_t *r, const fe_t *a, const fe_t *b) {
  for (int i = 0; i < 5; i++) r->l[i] = a->l[i] + b->l[i];
  fe_carry1(r);
}

static void fe_sub(fe_t *r, const fe_t *a, const fe_t *b) {
  /* add 4p (limbwise) before subtracting so nothing underflows: inputs have limbs < 2^52 */
  static const uint64_t fourp0 = 4 * ((1ULL << 51) - 19), fourp = 4 * ((1ULL << 51) - 1);
  r->l[0] = a->l[0] + fourp0 - b->l[0];
  for (int i = 1; i < 5; i++) r->l[i] = a->l[i] + fourp - b->l[i];
  fe_carry1(r);
}

static void fe_neg(fe_t *r, const fe_t *a) { fe_t z; fe_0(&z); fe_sub(r, &z, a); }

static void fe_mul(fe_t *r, const fe_t *a, const fe_t *b) {
  const uint64_t *x = a->l, *y = b->l;
  uint64_t y1_19 = 19 * y[1], y2_19 = 19 * y[2], y3_19 = 19 * y[3], y4_19 = 19 * y[4];
  u128 t0 = (u128)x[0] * y[0] + (u128)x[1] * y4_19 + (u128)x[2] * y3_19 + (u128)x[3] * y2_19 + (u128)x[4] * y1_19;
  u128 t1 = (u128)x[0] * y[1] + (u128)x[1] * y[0] + (u128)x[2] * y4_19 + (u128)x[3] * y3_19 + (u128)x[4] * y2_19;
  u128 t2 = (u128)x[0] * y[2] + (u128)x[1] * y[1] + (u128)x[2] * y[0] + (u128)x[3] * y4_19 + (u128)x[4] * y3_19;
  u128 t3 = (u128)x[0] * y[3] + (u128)x[1] * y[2] + (u128)x[2] * y[1] + (u128)x[3] * y[0] + (u128)x[4] * y4_19;
  u128 t4 = (u128)x[0] * y[4] + (u128)x[1] * y[3] + (u128)x[2] * y[2] + (u128)x[3] * y[1] + (u128)x[4] * y[0];
  uint64_t c;
  t1 += (uint64_t)(t0 >> 51); r->l[0] = (uint64_t)t0 & MASK51;
  t2 += (uint64_t)(t1 >> 51); r->l[1] = (uint64_t)t1 & MASK51;
  t3 += (uint64_t)(t2 >> 51); r->l[2] = (uint64_t)t2 & MASK51;
  t4 += (uint64_t)(t3 >> 51); r->l[3] = (uint64_t)t3 & MASK51;
  c = (uint64_t)(t4 >> 51); r->l[4] = (uint64_t)t4 & MASK51;
  r->l[0] += 19 * c;
  c = r->l[0] >> 51; r->l[0] &= MASK51; r->l[1] += c;
}

/* the same product with the 10 off-diagonal terms taken once, doubled */
static void fe_sq(fe_t *r, const fe_t *a) {
  const uint64_t *x = a->l;
  const uint64_t x0_2 = 2 * x[0], x1_2 = 2 * x[1], x1_38 = 38 * x[1], x2_38 = 38 * x[2], x3_38 = 38 * x[3], x3_19 = 19 * x[3], x4_19 = 19 * x[4];
  u128 t0 = (u128)x[0] * x[0] + (u128)x1_38 * x[4] + (u128)x2_38 * x[3];
  u128 t1 = (u128)x0_2 * x[1] + (u128)x2_38 * x[4] + (u128)x3_19 * x[3];
  u128 t2 = (u128)x0_2 * x[2] + (u128)x[1] * x[1] + (u128)x3_38 * x[4];
  u128 t3 = (u128)x0_2 * x[3] + (u128)x1_2 * x[2] + (u128)x4_19 * x[4];
  u128 t4 = (u128)x0_2 * x[4] + (u128)x1_2 * x[3] + (u128)x[2] * x[2];
  uint64_t c;
  t1 += (uint64_t)(t0 >> 51); r->l[0] = (uint64_t)t0 & MASK51;
  t2 += (uint64_t)(t1 >> 51); r->l[1] = (uint64_t)t1 & MASK51;
  t3 += (uint64_t)(t2 >> 51); r->l[2] = (uint64_t)t2 & MASK51;
  t4 += (uint64_t)(t3 >> 51); r->l[3] = (uint64_t)t3 & MASK51;
  c = (uint64_t)(t4 >> 51); r->l[4] = (uint64_t)t4 & MASK51;
  r->l[0] += 19 * c;
  c = r->l[0] >> 51; r->l[0] &= MASK51; r->l[1] += c;
}

void fe_from_bytes(fe_t *r, const uint8_t b[32]) {
  uint64_t w[4];
  for (int i = 0; i < 4; i++) {
    w[i] = 0;
    for (int j = 7; j >= 0; j--) w[i] = (w[i] << 8) | b[8 * i + j];
  }
  r->l[0] = w[0] & MASK51;
  r->l[1] = ((w[0] >> 51) | (w[1] << 13)) & MASK51;
  r->l[2] = ((w[1] >> 38) | (w[2] << 26)) & MASK51;
  r->l[3] = ((w[2] >> 25) | (w[3] << 39)) & MASK51;
  r->l[4] = (w[3] >> 12) & MASK51; /* drops bit 255 */
}

void fe_to_bytes(uint8_t b[32], const fe_t *a) {
  fe_t t = *a;
  fe_carry(&t);
  /* canonical reduction: compute q = floor((t + 19) / 2^255), then t += 19*q, drop bit 255 */
  uint64_t q = (t.l[0] + 19) >> 51;
  q = (t.l[1] + q) >> 51; q = (t.l[2] + q) >> 51; q = (t.l[3] + q) >> 51; q = (t.l[4] + q) >> 51;
  t.l[0] += 19 * q;
  uint64_t c;
  c = t.l[0] >> 51; t.l[0] &= MASK51; t.l[1] += c;
  c = t.l[1] >> 51; t.l[1] &= MASK51; t.l[2] += c;
  c = t.l[2] >> 51; t.l[2] &= MASK51; t.l[3] += c;
  c = t.l[3] >> 51; t.l[3] &= MASK51; t.l[4] += c;
  t.l[4] &= MASK51;
  uint64_t w[4];
  w[0] = t.l[0] | (t.l[1] << 51);
  w[1] = (t.l[1] >> 13) | (t.l[2] << 38);
  w[2] = (t.l[2] >> 26) | (t.l[3] << 25);
  w[3] = (t.l[3] >> 39) | (t.l[4] << 12);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 8; j++) b[8 * i + j] = (uint8_t)(w[i] >> (8 * j));
}

static int fe_is_negative(const fe_t *a) { uint8_t b[32]; fe_to_bytes(b, a); return b[0] & 1; }
static int fe_eq(const fe_t *a, const fe_t *b) { uint8_t x[32], y[32]; fe_to_bytes(x, a); fe_to_bytes(y, b); return memcmp(x, y, 32) == 0; }
static int fe_is_zero(const fe_t *a) { fe_t z; fe_0(&z); return fe_eq(a, &z); }
static void fe_abs(fe_t *r, const fe_t *a) { if (fe_is_negative(a)) fe_neg(r, a); else *r = *a; }

static void fe_pow_bits(fe_t *r, const fe_t *a, const uint8_t e[32]) {
  /* generic square-and-multiply over a 256-bit little-endian exponent */
  fe_t acc; fe_1(&acc);
  for (int i = 255; i >= 0; i--) {
    fe_sq(&acc, &acc);
    if ((e[i / 8] >> (i % 8)) & 1) fe_mul(&acc, &acc, a);
  }
  *r = acc;
}

static __attribute__((unused)) void fe_invert(fe_t *r, const fe_t *a) {
  /* a^(p-2), p-2 = 2^255 - 21 */
  uint8_t e[32]; memset(e, 0xff, 32); e[0] = 0xeb; e[31] = 0x7f;
  fe_pow_bits(r, a, e);
}

static void fe_pow_p58(fe_t *r, const fe_t *a) {
  /* a^((p-5)/8), (p-5)/8 = 2^252 - 3 */
  uint8_t e[32]; memset(e, 0xff, 32); e[0] = 0xfd; e[31] = 0x0f;
  fe_pow_bits(r, a, e);
}

/* RFC 9496 4.2 SQRT_RATIO_M1 */
static int sqrt_ratio_m1(fe_t *out, const fe_t *u, const fe_t *v) {
  fe_t v3, v7, r, check, t, nu, nu_i;
  fe_sq(&v3, v); fe_mul(&v3, &v3, v);
  fe_sq(&v7, &v3); fe_mul(&v7, &v7, v);
  fe_mul(&t, u, &v7); fe_pow_p58(&t, &t);
  fe_mul(&r, u, &v3); fe_mul(&r, &r, &t);
  fe_sq(&check, &r); fe_mul(&check, &check, v);
  fe_neg(&nu, u); fe_mul(&nu_i, &nu, &C_SQRT_M1);
  int correct = fe_eq(&check, u), flipped = fe_eq(&check, &nu), flipped_i = fe_eq(&check, &nu_i);
  if (flipped || flipped_i) fe_mul(&r, &r, &C_SQRT_M1);
  fe_abs(out, &r);
  return correct || flipped;
}

static void init_consts(void) {
  if (consts_ready) return;
  fe_from_bytes(&C_D, K_D); fe_from_bytes(&C_D2, K_D2); fe_from_bytes(&C_SQRT_M1, K_SQRT_M1);
  fe_from_bytes(&C_INVSQRT_A_MINUS_D, K_INVSQRT_A_MINUS_D);
  fe_from_bytes(&C_SQRT_AD_MINUS_ONE, K_SQRT_AD_MINUS_ONE);
  fe_from_bytes(&C_ONE_MINUS_D_SQ, K_ONE_MINUS_D_SQ);
  fe_from_bytes(&C_D_MINUS_ONE_SQ, K_D_MINUS_ONE_SQ);
  fe_from_bytes(&C_B.X, K_BX); fe_from_bytes(&C_B.Y, K_BY); fe_1(&C_B.Z); fe_mul(&C_B.T, &C_B.X, &C_B.Y);
  consts_ready = 1;
}

/* ------------------------------------------------------------------ Edwards group */

void ge_identity(ge_t *r) { fe_0(&r->X); fe_1(&r->Y); fe_1(&r->Z); fe_0(&r->T); }
void ge_basepoint(ge_t *r) { init_consts(); *r = C_B; }

void ge_add(ge_t *r, const ge_t *p, const ge_t *q) {
  init_consts();
  fe_t A, B, C, D, E, F, G, H, t;
  fe_sub(&A, &p->Y, &p->X); fe_sub(&t, &q->Y, &q->X); fe_mul(&A, &A, &t);
  fe_add(&B, &p->Y, &p->X); fe_add(&t, &q->Y, &q->X); fe_mul(&B, &B, &t);
  fe_mul(&C, &p->T, &C_D2); fe_mul(&C, &C, &q->T);
  fe_mul(&D, &p->Z, &q->Z); fe_add(&D, &D, &D);
  fe_sub(&E, &B, &A); fe_sub(&F, &D, &C); fe_add(&G, &D, &C); fe_add(&H, &B, &A);
  fe_mul(&r->X, &E, &F); fe_mul(&r->Y, &G, &H); fe_mul(&r->T, &E, &H); fe_mul(&r->Z, &F, &G);
}

void ge_neg(ge_t *r, const ge_t *p) { fe_neg(&r->X, &p->X); r->Y = p->Y; r->Z = p->Z; fe_neg(&r->T, &p->T); }
void ge_sub(ge_t *r, const ge_t *p, const ge_t *q) { ge_t n; ge_neg(&n, q); ge_add(r, p, &n); }

void ge_double(ge_t *r, const ge_t *p) {
  fe_t A, B, C, D, E, F, G, H, t;
  fe_sq(&A, &p->X); fe_sq(&B, &p->Y); fe_sq(&C, &p->Z); fe_add(&C, &C, &C);
  fe_neg(&D, &A);
  fe_add(&t, &p->X, &p->Y); fe_sq(&E, &t); fe_sub(&E, &E, &A); fe_sub(&E, &E, &B);
  fe_add(&G, &D, &B); fe_sub(&F, &G, &C); fe_sub(&H, &D, &B);
  fe_mul(&r->X, &E, &F); fe_mul(&r->Y, &G, &H); fe_mul(&r->T, &E, &H); fe_mul(&r->Z, &F, &G);
}

int ge_eq(const ge_t *p, const ge_t *q) {
  fe_t a, b;
  fe_mul(&a, &p->X, &q->Y); fe_mul(&b, &p->Y, &q->X);
  if (fe_eq(&a, &b)) return 1;
  fe_mul(&a, &p->Y, &q->Y); fe_mul(&b, &p->X, &q->X);
  return fe_eq(&a, &b);
}

/* RFC 9496 4.3.2 Encode */
void ge_compress(uint8_t out[32], const ge_t *p) {
  init_consts();
  fe_t u1, u2, t, invsqrt, den1, den2, z_inv, ix0, iy0, ench, x, y, den_inv, s, one;
  fe_1(&one);
  fe_add(&u1, &p->Z, &p->Y); fe_sub(&t, &p->Z, &p->Y); fe_mul(&u1, &u1, &t);
  fe_mul(&u2, &p->X, &p->Y);
  fe_sq(&t, &u2); fe_mul(&t, &t, &u1);
  (void)sqrt_ratio_m1(&invsqrt, &one, &t);
  fe_mul(&den1, &invsqrt, &u1); fe_mul(&den2, &invsqrt, &u2);
  fe_mul(&z_inv, &den1, &den2); fe_mul(&z_inv, &z_inv, &p->T);
  fe_mul(&ix0, &p->X, &C_SQRT_M1); fe_mul(&iy0, &p->Y, &C_SQRT_M1);
  fe_mul(&ench, &den1, &C_INVSQRT_A_MINUS_D);
  fe_mul(&t, &p->T, &z_inv);
  int rotate = fe_is_negative(&t);
  if (rotate) { x = iy0; y = ix0; den_inv = ench; } else { x = p->X; y = p->Y; den_inv = den2; }
  fe_mul(&t, &x, &z_inv);
  if (fe_is_negative(&t)) fe_neg(&y, &y);
  fe_sub(&t, &p->Z, &y); fe_mul(&s, &den_inv, &t); fe_abs(&s, &s);
  fe_to_bytes(out, &s);
}

/* RFC 9496 4.3.1 Decode */
int ge_decompress(ge_t *r, const uint8_t in[32]) {
  init_consts();
  fe_t s, ss, u1, u2, u2s, v, t, invsqrt, den_x, den_y, x, y, one;
  uint8_t chk[32];
  fe_from_bytes(&s, in); fe_to_bytes(chk, &s);
  if (memcmp(chk, in, 32) != 0 || (in[0] & 1)) return 0; /* non-canonical or negative */
  fe_1(&one);
  fe_sq(&ss, &s); fe_sub(&u1, &one, &ss); fe_add(&u2, &one, &ss); fe_sq(&u2s, &u2);
  fe_sq(&t, &u1); fe_mul(&t, &t, &C_D); fe_neg(&t, &t); fe_sub(&v, &t, &u2s);
  fe_mul(&t, &v, &u2s);
  int was_square = sqrt_ratio_m1(&invsqrt, &one, &t);
  fe_mul(&den_x, &invsqrt, &u2); fe_mul(&den_y, &invsqrt, &den_x); fe_mul(&den_y, &den_y, &v);
  fe_mul(&x, &s, &den_x); fe_add(&x, &x, &x); fe_abs(&x, &x);
  fe_mul(&y, &u1, &den_y);
  fe_mul(&t, &x, &y);
  if (!was_square || fe_is_negative(&t) || fe_is_zero(&y)) return 0;
  r->X = x; r->Y = y; fe_1(&r->Z); r->T = t;
  return 1;
}

/* RFC 9496 4.3.4 MAP (Elligator) */
static void elligator_map(ge_t *r, const fe_t *t) {
  fe_t rr, u, v, s, sp, c, N, w0, w1, w2, w3, one, tmp, m1;
  fe_1(&one); fe_neg(&m1, &one);
  fe_sq(&rr, t); fe_mul(&rr, &rr, &C_SQRT_M1);
  fe_add(&u, &rr, &one); fe_mul(&u, &u, &C_ONE_MINUS_D_SQ);
  fe_mul(&tmp, &rr, &C_D); fe_sub(&tmp, &m1, &tmp); /* -1 - r*d */
  fe_add(&v, &rr, &C_D); fe_mul(&v, &tmp, &v);
  int was_square = sqrt_ratio_m1(&s, &u, &v);
  fe_mul(&sp, &s, t); fe_abs(&sp, &sp); fe_neg(&sp, &sp);
  if (!was_square) { s = sp; c = rr; } else { c = m1; }
  fe_sub(&tmp, &rr, &one); fe_mul(&N, &c, &tmp); fe_mul(&N, &N, &C_D_MINUS_ONE_SQ); fe_sub(&N, &N, &v);
  fe_mul(&w0, &s, &v); fe_add(&w0, &w0, &w0);
  fe_mul(&w1, &N, &C_SQRT_AD_MINUS_ONE);
  fe_sq(&tmp, &s); fe_sub(&w2, &one, &tmp); fe_add(&w3, &one, &tmp);
  fe_mul(&r->X, &w0, &w3); fe_mul(&r->Y, &w2, &w1); fe_mul(&r->Z, &w1, &w3); fe_mul(&r->T, &w0, &w2);
}

void ge_from_uniform_bytes(ge_t *r, const uint8_t in[64]) {
  init_consts();
  fe_t t0, t1; ge_t p0, p1;
  fe_from_bytes(&t0, in); fe_from_bytes(&t1, in + 32); /* bit 255 of each half is masked */
  elligator_map(&p0, &t0); elligator_map(&p1, &t1);
  ge_add(r, &p0, &p1);
}

void ge_to_xyzt(uint8_t out[128], const ge_t *p) {
  fe_to_bytes(out, &p->X); fe_to_bytes(out + 32, &p->Y); fe_to_bytes(out + 64, &p->Z); fe_to_bytes(out + 96, &p->T);
}
void ge_from_xyzt(ge_t *r, const uint8_t in[128]) {
  fe_from_bytes(&r->X, in); fe_from_bytes(&r->Y, in + 32); fe_from_bytes(&r->Z, in + 64); fe_from_bytes(&r->T, in + 96);
}

/* ------------------------------------------------------------------ scalar mul / MSM */

void ge_scalarmul_bytes(ge_t *r, const uint8_t s[32], const ge_t *p) {
  ge_t acc; ge_identity(&acc);
  int started = 0;
  for (int i = 255; i >= 0; i--) {
    if (started) ge_double(&acc, &acc);
    if ((s[i / 8] >> (i % 8)) & 1) { ge_add(&acc, &acc, p); started = 1; }
  }
  *r = acc;
}

void ge_scalarmul(ge_t *r, const fq_t *s, const ge_t *p) {
  uint8_t b[32]; fq_to_bytes(b, s); ge_scalarmul_bytes(r, b, p);
}

/* signed radix-2^w digits of a canonical scalar (< 2^253): digits in [-2^(w-1), 2^(w-1)) */
static int to_radix_2w(int8_t *digits_or_null, int16_t *digits, const uint8_t s[32], int w) {
  (void)digits_or_null;
  int ndig = (256 + w - 1) / w + 1;
  uint64_t sc[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++)
    for (int j = 7; j >= 0; j--) sc[i] = (sc[i] << 8) | s[8 * i + j];
  int64_t carry = 0;
  const int64_t radix = 1LL << w, mask = radix - 1;
  for (int i = 0; i < ndig; i++) {
    int bit = i * w, word = bit / 64, off = bit % 64;
    uint64_t v = word < 4 ? sc[word] >> off : 0;
    if (off + w > 64 && word + 1 < 4) v |= sc[word + 1] << (64 - off);
    int64_t coef = carry + (int64_t)(v & (uint64_t)mask);
    carry = (coef + radix / 2) >> w;
    digits[i] = (int16_t)(coef - (carry << w));
  }
  return ndig;
}

void ge_msm(ge_t *r, const fq_t *scalars, const ge_t *points, size_t n) {
  if (n == 0) { ge_identity(r); return; }
  if (n < 8) {
    ge_t acc, t; ge_identity(&acc);
    for (size_t i = 0; i < n; i++) {
      if (fq_is_zero(&scalars[i])) continue;
      ge_scalarmul(&t, &scalars[i], &points[i]); ge_add(&acc, &acc, &t);
    }
    *r = acc; return;
  }
  int w = n < 500 ? 6 : n < 800 ? 7 : 8; /* dalek's pippenger window choice */
  int ndig = (256 + w - 1) / w + 1;
  int16_t *digits = (int16_t *)malloc(sizeof(int16_t) * n * (size_t)ndig);
  uint8_t *nz = (uint8_t *)calloc(n, 1);
  for (size_t i = 0; i < n; i++) {
    if (fq_is_zero(&scalars[i])) continue;
    nz[i] = 1;
    uint8_t b[32]; fq_to_bytes(b, &scalars[i]);
    to_radix_2w(NULL, digits + i * (size_t)ndig, b, w);
  }
  int nb = 1 << (w - 1);
  ge_t *buckets = (ge_t *)malloc(sizeof(ge_t) * (size_t)nb);
  uint8_t *used = (uint8_t *)malloc((size_t)nb);
  ge_t total; ge_identity(&total);
  for (int d = ndig - 1; d >= 0; d--) {
    for (int k = 0; k < w; k++) ge_double(&total, &total);
    memset(used, 0, (size_t)nb);
    int any = 0;
    for (size_t i = 0; i < n; i++) {
      if (!nz[i]) continue;
      int dg = digits[i * (size_t)ndig + d];
      if (dg == 0) continue; /* zero-digit skipping (vartime) */
      int b = (dg > 0 ? dg : -dg) - 1;
      ge_t pt;
      if (dg > 0) pt = points[i]; else ge_neg(&pt, &points[i]);
      if (!used[b]) { buckets[b] = pt; used[b] = 1; } else ge_add(&buckets[b], &buckets[b], &pt);
      any = 1;
    }
    if (!any) continue;
    /* sum_b (b+1)*bucket[b] by running sums */
    ge_t run, colsum; int run_set = 0, col_set = 0;
    for (int b = nb - 1; b >= 0; b--) {
      if (used[b]) { if (run_set) ge_add(&run, &run, &buckets[b]); else { run = buckets[b]; run_set = 1; } }
      if (run_set) { if (col_set) ge_add(&colsum, &colsum, &run); else { colsum = run; col_set = 1; } }
    }
    if (col_set) ge_add(&total, &total, &colsum);
  }
  free(digits); free(nz); free(buckets); free(used);
  *r = total;
}

void oracle_gens_new(ge_t *gens, size_t n, const uint8_t *label, size_t label_len) {
  init_consts();
  uint8_t bc[32];
  ge_compress(bc, &C_B); /* GROUP_BASEPOINT_COMPRESSED, group.rs:26-27 */
  shake256_ctx sh;
  shake256_init(&sh);
  shake256_absorb(&sh, label, label_len);
  shake256_absorb(&sh, bc, 32);
  shake256_finalize(&sh);
  for (size_t i = 0; i < n + 1; i++) {
    uint8_t u[64];
    shake256_squeeze(&sh, u, 64);
    ge_from_uniform_bytes(&gens[i], u);
  }
}

void oracle_commit(ge_t *r, const fq_t *v, size_t n, const fq_t *blind, const ge_t *G, const ge_t *h) {
  ge_t m, b;
  ge_msm(&m, v, G, n);
  ge_scalarmul(&b, blind, h);
  ge_add(r, &m, &b);
}

void oracle_hyrax_commit(uint8_t *out, const fq_t *Z, size_t L_size, size_t R_size,
                         const fq_t *blinds, const ge_t *G, const ge_t *h, int threads) {
  init_consts();
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
  for (long i = 0; i < (long)L_size; i++) {
    ge_t c;
    oracle_commit(&c, Z + (size_t)i * R_size, R_size, &blinds[i], G, h);
    ge_compress(out + 32 * (size_t)i, &c);
  }
}
