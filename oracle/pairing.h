/*
 * oracle/pairing.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * BN254 G2 and a reduced Tate pairing e: G1 x G2 -> Fq12, for the oracle's KZG/GWC verifier: with it
 * a proof is checked against the PUBLIC verification equation e(W, [s]_2) = e(zW + C - eG, [1]_2)
 * (halo2_proofs poly/kzg/multiopen/gwc/verifier.rs + poly/kzg/strategy.rs, the path
 * /root/reference/src/wnn.rs:265-280 `Wnn::verify_proof` takes), using only g2 and s_g2 -- not the
 * toxic scalar.  Any bilinear non-degenerate pairing decides that equation identically, so the plain
 * Tate pairing (Miller loop over r, exponent (q^12-1)/r by square-and-multiply) is used instead of
 * halo2curves' optimal-ate: simplest to get right, speed is irrelevant here (~20 ms).
 * Pinned by bilinearity / non-degeneracy tests and a Python big-integer re-computation
 * (tests/test_oracle_pairing.py).
 */
#ifndef ZG_ORACLE_PAIRING_H
#define ZG_ORACLE_PAIRING_H

#include "bn254.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { orc_fq c0, c1; } orc_fq2;          /* c0 + c1*i, i^2 = -1 */
typedef struct { orc_fq2 x, y; } orc_g2a;           /* affine on y^2 = x^3 + 3/(9+i); (0,0) = identity */
typedef struct { orc_fq2 c[6]; } orc_fq12;          /* sum c[j] w^j, w^6 = 9 + i */

void orc_g2_generator(orc_g2a *o);
int  orc_g2a_on_curve(const orc_g2a *p);
int  orc_g2a_is_identity(const orc_g2a *p);
void orc_g2a_add(orc_g2a *o, const orc_g2a *p, const orc_g2a *q);
void orc_g2a_mul(orc_g2a *o, const orc_g2a *p, const orc_fr *k);
int  orc_g2a_eq(const orc_g2a *p, const orc_g2a *q);

/* reduced Tate pairing; identity inputs give 1 */
void orc_pairing(orc_fq12 *o, const orc_g1a *p, const orc_g2a *q);
int  orc_fq12_eq(const orc_fq12 *a, const orc_fq12 *b);
int  orc_fq12_is_one(const orc_fq12 *a);
void orc_fq12_mul(orc_fq12 *o, const orc_fq12 *a, const orc_fq12 *b);
void orc_fq12_pow_fr(orc_fq12 *o, const orc_fq12 *a, const orc_fr *k);
/* prod_i e(p[i], q[i]) == 1 ?  (one shared final exponentiation) */
int  orc_pairing_check(const orc_g1a *p, const orc_g2a *q, size_t n);

#ifdef __cplusplus
}
#endif
#endif
