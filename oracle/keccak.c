/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see fq.h / keccak.h).
 */
#include "keccak.h"
#include <string.h>

static const uint64_t RC[24] = {
  0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
  0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
  0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
  0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
  0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
  0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};

static inline uint64_t rol(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

void keccak_f1600(uint64_t st[25]) {
  uint64_t bc[5], t;
  for (int round = 0; round < 24; round++) {
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      t = bc[(i + 4) % 5] ^ rol(bc[(i + 1) % 5], 1);
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = PILN[i];
      bc[0] = st[j];
      st[j] = rol(t, ROTC[i]);
      t = bc[0];
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= RC[round];
  }
}

/* byte-addressed access to the little-endian lane array */
static inline void st_xor_byte(uint64_t *st, size_t i, uint8_t b) { st[i / 8] ^= (uint64_t)b << (8 * (i % 8)); }
static inline uint8_t st_get_byte(const uint64_t *st, size_t i) { return (uint8_t)(st[i / 8] >> (8 * (i % 8))); }

#define SHAKE256_RATE 136

void shake256_init(shake256_ctx *c) { memset(c, 0, sizeof *c); }

void shake256_absorb(shake256_ctx *c, const uint8_t *in, size_t n) {
  for (size_t i = 0; i < n; i++) {
    st_xor_byte(c->st, c->pos++, in[i]);
    if (c->pos == SHAKE256_RATE) { keccak_f1600(c->st); c->pos = 0; }
  }
}

void shake256_finalize(shake256_ctx *c) {
  st_xor_byte(c->st, c->pos, 0x1f);
  st_xor_byte(c->st, SHAKE256_RATE - 1, 0x80);
  keccak_f1600(c->st);
  c->pos = 0;
  c->squeezing = 1;
}

void shake256_squeeze(shake256_ctx *c, uint8_t *out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    if (c->pos == SHAKE256_RATE) { keccak_f1600(c->st); c->pos = 0; }
    out[i] = st_get_byte(c->st, c->pos++);
  }
}

/* ------------------------------------------------------------------ STROBE-128 / Merlin */

#define STROBE_R 166
#define FLAG_I 1
#define FLAG_A 2
#define FLAG_C 4
#define FLAG_T 8
#define FLAG_M 16
#define FLAG_K 32

static void strobe_run_f(merlin_t *s) {
  s->st[s->pos] ^= s->pos_begin;
  s->st[s->pos + 1] ^= 0x04;
  s->st[STROBE_R + 1] ^= 0x80;
  uint64_t lanes[25];
  for (int i = 0; i < 25; i++) {
    lanes[i] = 0;
    for (int j = 7; j >= 0; j--) lanes[i] = (lanes[i] << 8) | s->st[8 * i + j];
  }
  keccak_f1600(lanes);
  for (int i = 0; i < 25; i++)
    for (int j = 0; j < 8; j++) s->st[8 * i + j] = (uint8_t)(lanes[i] >> (8 * j));
  s->pos = 0;
  s->pos_begin = 0;
}

static void strobe_absorb(merlin_t *s, const uint8_t *data, size_t n) {
  for (size_t i = 0; i < n; i++) {
    s->st[s->pos] ^= data[i];
    s->pos++;
    if (s->pos == STROBE_R) strobe_run_f(s);
  }
}

static void strobe_squeeze(merlin_t *s, uint8_t *data, size_t n) {
  for (size_t i = 0; i < n; i++) {
    data[i] = s->st[s->pos];
    s->st[s->pos] = 0;
    s->pos++;
    if (s->pos == STROBE_R) strobe_run_f(s);
  }
}

static void strobe_begin_op(merlin_t *s, uint8_t flags, int more) {
  if (more) return; /* continuing the previous operation (flags must match) */
  uint8_t old_begin = s->pos_begin;
  s->pos_begin = (uint8_t)(s->pos + 1);
  s->cur_flags = flags;
  uint8_t hdr[2] = {old_begin, flags};
  strobe_absorb(s, hdr, 2);
  int force_f = (flags & (FLAG_C | FLAG_K)) != 0;
  if (force_f && s->pos != 0) strobe_run_f(s);
}

static void strobe_meta_ad(merlin_t *s, const uint8_t *d, size_t n, int more) { strobe_begin_op(s, FLAG_M | FLAG_A, more); strobe_absorb(s, d, n); }
static void strobe_ad(merlin_t *s, const uint8_t *d, size_t n, int more) { strobe_begin_op(s, FLAG_A, more); strobe_absorb(s, d, n); }
static void strobe_prf(merlin_t *s, uint8_t *d, size_t n, int more) { strobe_begin_op(s, FLAG_I | FLAG_A | FLAG_C, more); strobe_squeeze(s, d, n); }

static void strobe_init(merlin_t *s, const uint8_t *proto, size_t n) {
  memset(s, 0, sizeof *s);
  static const uint8_t hdr[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
  memcpy(s->st, hdr, 6);
  memcpy(s->st + 6, "STROBEv1.0.2", 12);
  uint64_t lanes[25];
  for (int i = 0; i < 25; i++) {
    lanes[i] = 0;
    for (int j = 7; j >= 0; j--) lanes[i] = (lanes[i] << 8) | s->st[8 * i + j];
  }
  keccak_f1600(lanes);
  for (int i = 0; i < 25; i++)
    for (int j = 0; j < 8; j++) s->st[8 * i + j] = (uint8_t)(lanes[i] >> (8 * j));
  strobe_meta_ad(s, proto, n, 0);
}

static void le32(uint8_t b[4], uint32_t v) { for (int i = 0; i < 4; i++) b[i] = (uint8_t)(v >> (8 * i)); }

void merlin_append_message(merlin_t *t, const char *label, const uint8_t *msg, size_t n) {
  uint8_t len[4];
  le32(len, (uint32_t)n);
  strobe_meta_ad(t, (const uint8_t *)label, strlen(label), 0);
  strobe_meta_ad(t, len, 4, 1);
  strobe_ad(t, msg, n, 0);
}

void merlin_init(merlin_t *t, const uint8_t *label, size_t label_len) {
  strobe_init(t, (const uint8_t *)"Merlin v1.0", 11);
  merlin_append_message(t, "dom-sep", label, label_len);
}

void merlin_append_u64(merlin_t *t, const char *label, uint64_t v) {
  uint8_t b[8];
  for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
  merlin_append_message(t, label, b, 8);
}

void merlin_challenge_bytes(merlin_t *t, const char *label, uint8_t *out, size_t n) {
  uint8_t len[4];
  le32(len, (uint32_t)n);
  strobe_meta_ad(t, (const uint8_t *)label, strlen(label), 0);
  strobe_meta_ad(t, len, 4, 1);
  strobe_prf(t, out, n, 0);
}

void tr_append_protocol_name(merlin_t *t, const char *name) {
  merlin_append_message(t, "protocol-name", (const uint8_t *)name, strlen(name));
}

void tr_append_scalar(merlin_t *t, const char *label, const fq_t *s) {
  uint8_t b[32];
  fq_to_bytes(b, s);
  merlin_append_message(t, label, b, 32);
}

void tr_append_point(merlin_t *t, const char *label, const uint8_t c[32]) { merlin_append_message(t, label, c, 32); }

fq_t tr_challenge_scalar(merlin_t *t, const char *label) {
  uint8_t buf[64];
  merlin_challenge_bytes(t, label, buf, 64);
  return fq_from_bytes_wide(buf);
}

void tr_challenge_vector(merlin_t *t, const char *label, fq_t *out, size_t n) {
  for (size_t i = 0; i < n; i++) out[i] = tr_challenge_scalar(t, label);
}

void tr_append_scalars(merlin_t *t, const char *label, const fq_t *v, size_t n) {
  merlin_append_message(t, label, (const uint8_t *)"begin_append_vector", 19);
  for (size_t i = 0; i < n; i++) tr_append_scalar(t, label, &v[i]);
  merlin_append_message(t, label, (const uint8_t *)"end_append_vector", 17);
}

void tape_init(merlin_t *t, const uint8_t *name, size_t name_len, const uint8_t seed64[64]) {
  merlin_init(t, name, name_len);
  /* Scalar::random = from_u512 of 8 x next_u64 (ristretto255.rs:381-387) */
  fq_t s = fq_from_bytes_wide(seed64);
  tr_append_scalar(t, "init_randomness", &s);
}
