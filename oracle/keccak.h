/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see fq.h).
 * keccak.h : Keccak-f[1600], SHAKE256 (sha3 0.8.2 as used by
 * Spartan/src/commitments.rs:21-25) and the Merlin 3.0.0 transcript over STROBE-128
 * (Spartan/src/transcript.rs:19-43, Spartan/src/random.rs:12-31).  Neither crate is
 * vendored under /root/reference (Cargo.lock pins them); restated from FIPS 202 and the
 * published STROBE / Merlin specifications.
 * Pinning: hashlib.shake_256 (tests) and Merlin's published conformance vector.
 */
#ifndef VPIN_ORACLE_KECCAK_H
#define VPIN_ORACLE_KECCAK_H
#include "fq.h"

#ifdef __cplusplus
extern "C" {
#endif

void keccak_f1600(uint64_t st[25]);

typedef struct { uint64_t st[25]; size_t pos; int squeezing; } shake256_ctx;
void shake256_init(shake256_ctx *c);
void shake256_absorb(shake256_ctx *c, const uint8_t *in, size_t n);
void shake256_finalize(shake256_ctx *c);
void shake256_squeeze(shake256_ctx *c, uint8_t *out, size_t n);

/* Merlin transcript */
typedef struct { uint8_t st[200]; uint8_t pos, pos_begin, cur_flags; } merlin_t;
void merlin_init(merlin_t *t, const uint8_t *label, size_t label_len);        /* Transcript::new */
void merlin_append_message(merlin_t *t, const char *label, const uint8_t *msg, size_t n);
void merlin_append_u64(merlin_t *t, const char *label, uint64_t v);
void merlin_challenge_bytes(merlin_t *t, const char *label, uint8_t *out, size_t n);

/* ProofTranscript (Spartan/src/transcript.rs:19-43) */
void tr_append_protocol_name(merlin_t *t, const char *name);
void tr_append_scalar(merlin_t *t, const char *label, const fq_t *s);
void tr_append_point(merlin_t *t, const char *label, const uint8_t compressed[32]);
fq_t tr_challenge_scalar(merlin_t *t, const char *label);
void tr_challenge_vector(merlin_t *t, const char *label, fq_t *out, size_t n);
/* AppendToTranscript for [Scalar] (transcript.rs:56-64) */
void tr_append_scalars(merlin_t *t, const char *label, const fq_t *v, size_t n);

/* RandomTape (Spartan/src/random.rs:12-31) with the OsRng draw made injectable:
 * seed64 replaces the 64 bytes `Scalar::random` reads from OsRng. */
void tape_init(merlin_t *t, const uint8_t *name, size_t name_len, const uint8_t seed64[64]);

#ifdef __cplusplus
}
#endif
#endif
