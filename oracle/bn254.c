/*
 * oracle/bn254.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see bn254.h header).
 *
 * Restates halo2curves 0.3.3 bn256 {fr.rs, fq.rs, curve.rs} (upstream git dependency of the
 * reference, Cargo.toml:14-28; source absent from /root/reference).  Constants were re-derived
 * with Python big integers (tests/test_oracle_kat.py repeats the derivation).
 */
#include "bn254.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Fr */
const uint64_t ORC_FR_MODULUS[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL,
                                    0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const orc_fe FR_R = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL,
                             0x0e0a77c19a07df2fULL}};
static const orc_fe FR_R2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL,
                              0x0216d0b17f4e44a5ULL}};
#define FE(n) orc_fr_##n
#define FE_P ORC_FR_MODULUS
#define FE_INV 0xc2e1f593efffffffULL
#define FE_R FR_R
#define FE_R2 FR_R2
#include "fe_impl.inc"
#undef FE
#undef FE_P
#undef FE_INV
#undef FE_R
#undef FE_R2

/* ------------------------------------------------------------------ Fq */
const uint64_t ORC_FQ_MODULUS[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL,
                                    0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const orc_fe FQ_R = {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL,
                             0x0e0a77c19a07df2fULL}};
static const orc_fe FQ_R2 = {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL,
                              0x06d89f71cab8351fULL}};
#define FE(n) orc_fq_##n
#define FE_P ORC_FQ_MODULUS
#define FE_INV 0x87d20782e4866389ULL
#define FE_R FQ_R
#define FE_R2 FQ_R2
#include "fe_impl.inc"
#undef FE
#undef FE_P
#undef FE_INV
#undef FE_R
#undef FE_R2

const orc_fr ORC_FR_ZERO = {{0, 0, 0, 0}};
const orc_fr ORC_FR_ONE = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL,
                            0x0e0a77c19a07df2fULL}};
const orc_fq ORC_FQ_ZERO = {{0, 0, 0, 0}};
const orc_fq ORC_FQ_ONE = {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL,
                            0x0e0a77c19a07df2fULL}};

/* Canonical (non-Montgomery) values of the halo2curves Fr constants (exported for the KAT tests). */
const uint64_t ORC_FR_ROOT_OF_UNITY_RAW[4] = {0xd34f1ed960c37c9cULL, 0x3215cf6dd39329c8ULL,
                                                 0x98865ea93dd31f74ULL, 0x03ddb9f5166d18b7ULL};
const uint64_t ORC_FR_DELTA_RAW[4] = {0x870e56bbe533e9a2ULL, 0x5b5f898e5e963f25ULL,
                                         0x64ec26aad4c86e71ULL, 0x09226b6e22c6f0caULL};
const uint64_t ORC_FR_ZETA_RAW[4] = {0x8b17ea66b99c90ddULL, 0x5bfc41088d8daaa7ULL,
                                        0xb3c4d79d41a91758ULL, 0x0ULL};
/* Montgomery forms: raw * R mod r, computed offline with Python (tests re-check them). */
const orc_fr ORC_FR_ROOT_OF_UNITY = {{0x9632c7c5b639feb8ULL, 0x985ce3400d0ff299ULL,
                                      0xb2dd880001b0ecd8ULL, 0x1d69070d6d98ce29ULL}};
const orc_fr ORC_FR_DELTA = {{0x9a0c322befd78855ULL, 0x46e82d14249b563cULL, 0x5983a663e0b0b7a7ULL,
                              0x22ab452baaa111adULL}};
const orc_fr ORC_FR_ZETA = {{0x93e7cede4a0329b3ULL, 0x7d4fdca77a96c167ULL, 0x8be4ba08b19a750aULL,
                             0x1cbd5653a5661c25ULL}};

int orc_fr_cmp(const orc_fr *a, const orc_fr *b) {
    uint64_t x[4], y[4];
    orc_fr_to_raw(x, a);
    orc_fr_to_raw(y, b);
    for (int i = 3; i >= 0; i--) {
        if (x[i] < y[i]) return -1;
        if (x[i] > y[i]) return 1;
    }
    return 0;
}

/* Montgomery's trick; zeros are skipped and stay zero (halo2 `BatchInvert`). */
void orc_fr_batch_inv(orc_fr *a, size_t n) {
    if (n == 0) return;
    orc_fr *pre = (orc_fr *)malloc(n * sizeof(orc_fr));
    orc_fr acc = ORC_FR_ONE;
    for (size_t i = 0; i < n; i++) {
        pre[i] = acc;
        if (!orc_fr_is_zero(&a[i])) orc_fr_mul(&acc, &acc, &a[i]);
    }
    orc_fr_inv(&acc, &acc);
    for (size_t i = n; i-- > 0;) {
        if (orc_fr_is_zero(&a[i])) continue;
        orc_fr t;
        orc_fr_mul(&t, &acc, &pre[i]);
        orc_fr_mul(&acc, &acc, &a[i]);
        a[i] = t;
    }
    free(pre);
}

void orc_fr_from_be_bytes_reduce(orc_fr *o, const uint8_t b[32]) {
    uint64_t v[4] = {0, 0, 0, 0};
    for (int i = 0; i < 32; i++) v[(31 - i) / 8] |= (uint64_t)b[i] << (8 * ((31 - i) % 8));
    /* CIOS with a 256-bit left operand and R2 < r yields a value < 2r that the final
       conditional subtraction canonicalises, i.e. (v mod r) in Montgomery form. */
    orc_fr_from_raw(o, v);
}

/* ------------------------------------------------------------------ G1 */
void orc_g1_identity(orc_g1 *o) {
    o->x = ORC_FQ_ZERO;
    o->y = ORC_FQ_ONE;
    o->z = ORC_FQ_ZERO;
}

void orc_g1_generator(orc_g1 *o) {
    orc_fq_from_u64(&o->x, 1);
    orc_fq_from_u64(&o->y, 2);
    o->z = ORC_FQ_ONE;
}

int orc_g1_is_identity(const orc_g1 *p) { return orc_fq_is_zero(&p->z); }

void orc_g1_neg(orc_g1 *o, const orc_g1 *p) {
    o->x = p->x;
    orc_fq_neg(&o->y, &p->y);
    o->z = p->z;
}

/* dbl-2009-l (a = 0) */
void orc_g1_double(orc_g1 *o, const orc_g1 *p) {
    if (orc_g1_is_identity(p)) { orc_g1_identity(o); return; }
    orc_fq a, b, c, d, e, f, t, x3, y3, z3;
    orc_fq_sqr(&a, &p->x);
    orc_fq_sqr(&b, &p->y);
    orc_fq_sqr(&c, &b);
    orc_fq_add(&t, &p->x, &b);
    orc_fq_sqr(&t, &t);
    orc_fq_sub(&t, &t, &a);
    orc_fq_sub(&t, &t, &c);
    orc_fq_add(&d, &t, &t);
    orc_fq_add(&e, &a, &a);
    orc_fq_add(&e, &e, &a);
    orc_fq_sqr(&f, &e);
    orc_fq_add(&t, &d, &d);
    orc_fq_sub(&x3, &f, &t);
    orc_fq_sub(&t, &d, &x3);
    orc_fq_mul(&y3, &e, &t);
    orc_fq_add(&t, &c, &c);
    orc_fq_add(&t, &t, &t);
    orc_fq_add(&t, &t, &t);
    orc_fq_sub(&y3, &y3, &t);
    orc_fq_mul(&z3, &p->y, &p->z);
    orc_fq_add(&z3, &z3, &z3);
    o->x = x3;
    o->y = y3;
    o->z = z3;
}

/* add-2007-bl */
void orc_g1_add(orc_g1 *o, const orc_g1 *p, const orc_g1 *q) {
    if (orc_g1_is_identity(p)) { *o = *q; return; }
    if (orc_g1_is_identity(q)) { *o = *p; return; }
    orc_fq z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
    orc_fq_sqr(&z1z1, &p->z);
    orc_fq_sqr(&z2z2, &q->z);
    orc_fq_mul(&u1, &p->x, &z2z2);
    orc_fq_mul(&u2, &q->x, &z1z1);
    orc_fq_mul(&s1, &p->y, &q->z);
    orc_fq_mul(&s1, &s1, &z2z2);
    orc_fq_mul(&s2, &q->y, &p->z);
    orc_fq_mul(&s2, &s2, &z1z1);
    if (orc_fq_eq(&u1, &u2)) {
        if (orc_fq_eq(&s1, &s2)) orc_g1_double(o, p);
        else orc_g1_identity(o);
        return;
    }
    orc_fq_sub(&h, &u2, &u1);
    orc_fq_add(&i, &h, &h);
    orc_fq_sqr(&i, &i);
    orc_fq_mul(&j, &h, &i);
    orc_fq_sub(&r, &s2, &s1);
    orc_fq_add(&r, &r, &r);
    orc_fq_mul(&v, &u1, &i);
    orc_fq_sqr(&x3, &r);
    orc_fq_sub(&x3, &x3, &j);
    orc_fq_sub(&x3, &x3, &v);
    orc_fq_sub(&x3, &x3, &v);
    orc_fq_sub(&t, &v, &x3);
    orc_fq_mul(&y3, &r, &t);
    orc_fq_mul(&t, &s1, &j);
    orc_fq_add(&t, &t, &t);
    orc_fq_sub(&y3, &y3, &t);
    orc_fq_add(&z3, &p->z, &q->z);
    orc_fq_sqr(&z3, &z3);
    orc_fq_sub(&z3, &z3, &z1z1);
    orc_fq_sub(&z3, &z3, &z2z2);
    orc_fq_mul(&z3, &z3, &h);
    o->x = x3;
    o->y = y3;
    o->z = z3;
}

void orc_g1_from_affine(orc_g1 *o, const orc_g1a *p) {
    if (orc_fq_is_zero(&p->x) && orc_fq_is_zero(&p->y)) { orc_g1_identity(o); return; }
    o->x = p->x;
    o->y = p->y;
    o->z = ORC_FQ_ONE;
}

/* madd-2007-bl */
void orc_g1_add_mixed(orc_g1 *o, const orc_g1 *p, const orc_g1a *q) {
    if (orc_fq_is_zero(&q->x) && orc_fq_is_zero(&q->y)) { *o = *p; return; }
    if (orc_g1_is_identity(p)) { orc_g1_from_affine(o, q); return; }
    orc_fq z1z1, u2, s2, h, hh, i, j, r, v, t, x3, y3, z3;
    orc_fq_sqr(&z1z1, &p->z);
    orc_fq_mul(&u2, &q->x, &z1z1);
    orc_fq_mul(&s2, &q->y, &p->z);
    orc_fq_mul(&s2, &s2, &z1z1);
    if (orc_fq_eq(&u2, &p->x)) {
        if (orc_fq_eq(&s2, &p->y)) orc_g1_double(o, p);
        else orc_g1_identity(o);
        return;
    }
    orc_fq_sub(&h, &u2, &p->x);
    orc_fq_sqr(&hh, &h);
    orc_fq_add(&i, &hh, &hh);
    orc_fq_add(&i, &i, &i);
    orc_fq_mul(&j, &h, &i);
    orc_fq_sub(&r, &s2, &p->y);
    orc_fq_add(&r, &r, &r);
    orc_fq_mul(&v, &p->x, &i);
    orc_fq_sqr(&x3, &r);
    orc_fq_sub(&x3, &x3, &j);
    orc_fq_sub(&x3, &x3, &v);
    orc_fq_sub(&x3, &x3, &v);
    orc_fq_sub(&t, &v, &x3);
    orc_fq_mul(&y3, &r, &t);
    orc_fq_mul(&t, &p->y, &j);
    orc_fq_add(&t, &t, &t);
    orc_fq_sub(&y3, &y3, &t);
    orc_fq_add(&z3, &p->z, &h);
    orc_fq_sqr(&z3, &z3);
    orc_fq_sub(&z3, &z3, &z1z1);
    orc_fq_sub(&z3, &z3, &hh);
    o->x = x3;
    o->y = y3;
    o->z = z3;
}

void orc_g1_to_affine(orc_g1a *o, const orc_g1 *p) {
    if (orc_g1_is_identity(p)) { o->x = ORC_FQ_ZERO; o->y = ORC_FQ_ZERO; return; }
    orc_fq zi, zi2, zi3;
    orc_fq_inv(&zi, &p->z);
    orc_fq_sqr(&zi2, &zi);
    orc_fq_mul(&zi3, &zi2, &zi);
    orc_fq_mul(&o->x, &p->x, &zi2);
    orc_fq_mul(&o->y, &p->y, &zi3);
}

void orc_g1_batch_to_affine(orc_g1a *o, const orc_g1 *p, size_t n) {
    if (n == 0) return;
    orc_fq *pre = (orc_fq *)malloc(n * sizeof(orc_fq));
    orc_fq acc = ORC_FQ_ONE;
    for (size_t i = 0; i < n; i++) {
        pre[i] = acc;
        if (!orc_g1_is_identity(&p[i])) orc_fq_mul(&acc, &acc, &p[i].z);
    }
    orc_fq_inv(&acc, &acc);
    for (size_t i = n; i-- > 0;) {
        if (orc_g1_is_identity(&p[i])) { o[i].x = ORC_FQ_ZERO; o[i].y = ORC_FQ_ZERO; continue; }
        orc_fq zi, zi2, zi3;
        orc_fq_mul(&zi, &acc, &pre[i]);
        orc_fq_mul(&acc, &acc, &p[i].z);
        orc_fq_sqr(&zi2, &zi);
        orc_fq_mul(&zi3, &zi2, &zi);
        orc_fq_mul(&o[i].x, &p[i].x, &zi2);
        orc_fq_mul(&o[i].y, &p[i].y, &zi3);
    }
    free(pre);
}

void orc_g1_mul(orc_g1 *o, const orc_g1 *p, const orc_fr *k) {
    uint64_t e[4];
    orc_fr_to_raw(e, k);
    orc_g1 acc;
    orc_g1_identity(&acc);
    for (int i = 3; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            orc_g1_double(&acc, &acc);
            if ((e[i] >> b) & 1) orc_g1_add(&acc, &acc, p);
        }
    *o = acc;
}

int orc_g1_eq(const orc_g1 *p, const orc_g1 *q) {
    int pi = orc_g1_is_identity(p), qi = orc_g1_is_identity(q);
    if (pi || qi) return pi && qi;
    orc_fq z1z1, z2z2, a, b;
    orc_fq_sqr(&z1z1, &p->z);
    orc_fq_sqr(&z2z2, &q->z);
    orc_fq_mul(&a, &p->x, &z2z2);
    orc_fq_mul(&b, &q->x, &z1z1);
    if (!orc_fq_eq(&a, &b)) return 0;
    orc_fq_mul(&a, &p->y, &q->z);
    orc_fq_mul(&a, &a, &z2z2);
    orc_fq_mul(&b, &q->y, &p->z);
    orc_fq_mul(&b, &b, &z1z1);
    return orc_fq_eq(&a, &b);
}

int orc_g1a_on_curve(const orc_g1a *p) {
    if (orc_fq_is_zero(&p->x) && orc_fq_is_zero(&p->y)) return 1;
    orc_fq l, r, three;
    orc_fq_sqr(&l, &p->y);
    orc_fq_sqr(&r, &p->x);
    orc_fq_mul(&r, &r, &p->x);
    orc_fq_from_u64(&three, 3);
    orc_fq_add(&r, &r, &three);
    return orc_fq_eq(&l, &r);
}

/* ------------------------------------------------------------------ PRNG */
uint64_t orc_rng_next(orc_rng *g) {
    uint64_t z = (g->s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

/* Rejection-sample a canonical integer < r, then move it to Montgomery form. */
void orc_rng_fr(orc_rng *g, orc_fr *o) {
    uint64_t v[4];
    for (;;) {
        for (int i = 0; i < 4; i++) v[i] = orc_rng_next(g);
        v[3] &= 0x3fffffffffffffffULL;
        int lt = 0;
        for (int i = 3; i >= 0; i--) {
            if (v[i] < ORC_FR_MODULUS[i]) { lt = 1; break; }
            if (v[i] > ORC_FR_MODULUS[i]) break;
        }
        if (lt) break;
    }
    orc_fr_from_raw(o, v);
}

void orc_fill_fr(uint64_t seed, orc_fr *o, size_t n) {
    orc_rng g = {seed};
    for (size_t i = 0; i < n; i++) orc_rng_fr(&g, &o[i]);
}

void orc_fill_fr_sparse(uint64_t seed, orc_fr *o, size_t n) {
    orc_rng g = {seed};
    for (size_t i = 0; i < n; i++) {
        uint64_t c = orc_rng_next(&g) % 100;
        if (c < 70) o[i] = ORC_FR_ZERO;
        else if (c < 90) orc_fr_from_u64(&o[i], orc_rng_next(&g) & 1);
        else if (c < 98) orc_fr_from_u64(&o[i], orc_rng_next(&g) & 0xff);
        else orc_rng_fr(&g, &o[i]);
    }
}
