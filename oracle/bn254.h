/*
 * oracle/bn254.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the BN254 arithmetic that zero_g's proving path uses through
 * halo2curves 0.3.3 (`halo2_proofs::halo2curves::bn256::{Fr, Fq, G1, G1Affine}`, imported at
 * /root/reference/src/wnn.rs:18).  That crate is a git dependency (Cargo.toml:14-28 of the
 * reference) whose source is NOT in /root/reference, so this file restates the published
 * algorithms (4x64-bit Montgomery, R = 2^256; Jacobian G1 on y^2 = x^3 + 3) and is pinned by
 * mathematical known-answer tests (tests/test_oracle_kat.py): moduli, R, R^2, INV, 2G, r*G = inf,
 * ROOT_OF_UNITY, DELTA, ZETA re-derived with Python big integers.
 *
 * PARITY STATUS: "parity unpinned" against real halo2 bytes -- the reference holds no golden
 * vector for any field/curve/MSM/NTT value.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 */
#ifndef ZG_ORACLE_BN254_H
#define ZG_ORACLE_BN254_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 256-bit field element, 4 x u64 little-endian limbs, Montgomery form (a*R mod p). */
typedef struct { uint64_t l[4]; } orc_fe;
typedef orc_fe orc_fr; /* scalar field  */
typedef orc_fe orc_fq; /* base field    */

typedef struct { orc_fq x, y; } orc_g1a;    /* affine, (0,0) = identity (halo2curves G1Affine) */
typedef struct { orc_fq x, y, z; } orc_g1;  /* Jacobian, z = 0 = identity                      */

/* ---- Fr ---- */
extern const orc_fr ORC_FR_ZERO, ORC_FR_ONE, ORC_FR_ROOT_OF_UNITY, ORC_FR_DELTA, ORC_FR_ZETA;
extern const uint64_t ORC_FR_MODULUS[4];
extern const uint64_t ORC_FR_ROOT_OF_UNITY_RAW[4], ORC_FR_DELTA_RAW[4], ORC_FR_ZETA_RAW[4];
void orc_fr_add(orc_fr *o, const orc_fr *a, const orc_fr *b);
void orc_fr_sub(orc_fr *o, const orc_fr *a, const orc_fr *b);
void orc_fr_neg(orc_fr *o, const orc_fr *a);
void orc_fr_mul(orc_fr *o, const orc_fr *a, const orc_fr *b);
void orc_fr_sqr(orc_fr *o, const orc_fr *a);
void orc_fr_inv(orc_fr *o, const orc_fr *a);              /* 0 -> 0 */
void orc_fr_pow(orc_fr *o, const orc_fr *a, const uint64_t e[4]);
void orc_fr_pow_u64(orc_fr *o, const orc_fr *a, uint64_t e);
void orc_fr_from_u64(orc_fr *o, uint64_t v);
void orc_fr_from_raw(orc_fr *o, const uint64_t v[4]);     /* canonical integer -> Montgomery */
void orc_fr_to_raw(uint64_t v[4], const orc_fr *a);       /* Montgomery -> canonical integer */
int  orc_fr_eq(const orc_fr *a, const orc_fr *b);
int  orc_fr_is_zero(const orc_fr *a);
int  orc_fr_cmp(const orc_fr *a, const orc_fr *b);        /* by canonical integer, like Fr: Ord */
void orc_fr_batch_inv(orc_fr *a, size_t n);               /* in place, zeros stay zero */
void orc_fr_from_be_bytes_reduce(orc_fr *o, const uint8_t b[32]); /* 256-bit BE integer mod r */
void orc_fr_to_be_bytes(uint8_t b[32], const orc_fr *a);

/* ---- Fq ---- */
extern const orc_fq ORC_FQ_ZERO, ORC_FQ_ONE;
extern const uint64_t ORC_FQ_MODULUS[4];
void orc_fq_add(orc_fq *o, const orc_fq *a, const orc_fq *b);
void orc_fq_sub(orc_fq *o, const orc_fq *a, const orc_fq *b);
void orc_fq_neg(orc_fq *o, const orc_fq *a);
void orc_fq_mul(orc_fq *o, const orc_fq *a, const orc_fq *b);
void orc_fq_sqr(orc_fq *o, const orc_fq *a);
void orc_fq_inv(orc_fq *o, const orc_fq *a);
void orc_fq_from_u64(orc_fq *o, uint64_t v);
void orc_fq_from_raw(orc_fq *o, const uint64_t v[4]);
void orc_fq_to_raw(uint64_t v[4], const orc_fq *a);
int  orc_fq_eq(const orc_fq *a, const orc_fq *b);
int  orc_fq_is_zero(const orc_fq *a);
void orc_fq_to_be_bytes(uint8_t b[32], const orc_fq *a);

/* ---- G1 ---- */
void orc_g1_identity(orc_g1 *o);
void orc_g1_generator(orc_g1 *o);                         /* (1, 2, 1) */
int  orc_g1_is_identity(const orc_g1 *p);
void orc_g1_double(orc_g1 *o, const orc_g1 *p);
void orc_g1_add(orc_g1 *o, const orc_g1 *p, const orc_g1 *q);
void orc_g1_add_mixed(orc_g1 *o, const orc_g1 *p, const orc_g1a *q);
void orc_g1_neg(orc_g1 *o, const orc_g1 *p);
void orc_g1_from_affine(orc_g1 *o, const orc_g1a *p);
void orc_g1_to_affine(orc_g1a *o, const orc_g1 *p);
void orc_g1_batch_to_affine(orc_g1a *o, const orc_g1 *p, size_t n);
void orc_g1_mul(orc_g1 *o, const orc_g1 *p, const orc_fr *k);   /* double-and-add, canonical k */
int  orc_g1_eq(const orc_g1 *p, const orc_g1 *q);               /* projective equality */
int  orc_g1a_on_curve(const orc_g1a *p);

/* Deterministic PRNG used for every synthetic input (SplitMix64). */
typedef struct { uint64_t s; } orc_rng;
uint64_t orc_rng_next(orc_rng *g);
void orc_rng_fr(orc_rng *g, orc_fr *o);                   /* uniform Fr, Montgomery form */
void orc_fill_fr(uint64_t seed, orc_fr *o, size_t n);
/* "advice-like" scalars: 70% zero, 20% in {0,1}, 8% < 2^8, 2% uniform */
void orc_fill_fr_sparse(uint64_t seed, orc_fr *o, size_t n);

#ifdef __cplusplus
}
#endif
#endif
