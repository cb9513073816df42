/* oracle/pairing.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See pairing.h. */
#include "pairing.h"

#include <string.h>

/* ------------------------------------------------------------------ Fq2 = Fq[i]/(i^2 + 1) */
static void fq2_add(orc_fq2 *o, const orc_fq2 *a, const orc_fq2 *b) {
    orc_fq_add(&o->c0, &a->c0, &b->c0);
    orc_fq_add(&o->c1, &a->c1, &b->c1);
}
static void fq2_sub(orc_fq2 *o, const orc_fq2 *a, const orc_fq2 *b) {
    orc_fq_sub(&o->c0, &a->c0, &b->c0);
    orc_fq_sub(&o->c1, &a->c1, &b->c1);
}
static void fq2_mul(orc_fq2 *o, const orc_fq2 *a, const orc_fq2 *b) {
    orc_fq t0, t1, t2, t3;
    orc_fq_mul(&t0, &a->c0, &b->c0);
    orc_fq_mul(&t1, &a->c1, &b->c1);
    orc_fq_mul(&t2, &a->c0, &b->c1);
    orc_fq_mul(&t3, &a->c1, &b->c0);
    orc_fq_sub(&o->c0, &t0, &t1);
    orc_fq_add(&o->c1, &t2, &t3);
}
static void fq2_mul_fq(orc_fq2 *o, const orc_fq2 *a, const orc_fq *b) {
    orc_fq_mul(&o->c0, &a->c0, b);
    orc_fq_mul(&o->c1, &a->c1, b);
}
static void fq2_inv(orc_fq2 *o, const orc_fq2 *a) { /* (c0 - c1 i) / (c0^2 + c1^2) */
    orc_fq n, t;
    orc_fq_sqr(&n, &a->c0);
    orc_fq_sqr(&t, &a->c1);
    orc_fq_add(&n, &n, &t);
    orc_fq_inv(&n, &n);
    orc_fq_mul(&o->c0, &a->c0, &n);
    orc_fq_mul(&t, &a->c1, &n);
    orc_fq_neg(&o->c1, &t);
}
static int fq2_is_zero(const orc_fq2 *a) { return orc_fq_is_zero(&a->c0) && orc_fq_is_zero(&a->c1); }
static int fq2_eq(const orc_fq2 *a, const orc_fq2 *b) { return orc_fq_eq(&a->c0, &b->c0) && orc_fq_eq(&a->c1, &b->c1); }
static void fq2_from_u64(orc_fq2 *o, uint64_t c0, uint64_t c1) {
    orc_fq_from_u64(&o->c0, c0);
    orc_fq_from_u64(&o->c1, c1);
}
static void fq2_mul_xi(orc_fq2 *o, const orc_fq2 *a) { /* times 9 + i */
    orc_fq2 xi;
    fq2_from_u64(&xi, 9, 1);
    fq2_mul(o, a, &xi);
}

/* ------------------------------------------------------------------ G2 (affine; inversions are fine here) */
static const uint64_t G2_X0[4] = {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL};
static const uint64_t G2_X1[4] = {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL};
static const uint64_t G2_Y0[4] = {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL};
static const uint64_t G2_Y1[4] = {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL};

void orc_g2_generator(orc_g2a *o) {
    orc_fq_from_raw(&o->x.c0, G2_X0);
    orc_fq_from_raw(&o->x.c1, G2_X1);
    orc_fq_from_raw(&o->y.c0, G2_Y0);
    orc_fq_from_raw(&o->y.c1, G2_Y1);
}
int orc_g2a_is_identity(const orc_g2a *p) { return fq2_is_zero(&p->x) && fq2_is_zero(&p->y); }
int orc_g2a_eq(const orc_g2a *p, const orc_g2a *q) { return fq2_eq(&p->x, &q->x) && fq2_eq(&p->y, &q->y); }
int orc_g2a_on_curve(const orc_g2a *p) {
    if (orc_g2a_is_identity(p)) return 1;
    orc_fq2 l, r, b, xi, three;
    fq2_mul(&l, &p->y, &p->y);
    fq2_mul(&r, &p->x, &p->x);
    fq2_mul(&r, &r, &p->x);
    fq2_from_u64(&xi, 9, 1);
    fq2_inv(&xi, &xi);
    fq2_from_u64(&three, 3, 0);
    fq2_mul(&b, &three, &xi);
    fq2_add(&r, &r, &b);
    return fq2_eq(&l, &r);
}
void orc_g2a_add(orc_g2a *o, const orc_g2a *p, const orc_g2a *q) {
    if (orc_g2a_is_identity(p)) { *o = *q; return; }
    if (orc_g2a_is_identity(q)) { *o = *p; return; }
    orc_fq2 lam, t, x3, y3;
    if (fq2_eq(&p->x, &q->x)) {
        fq2_add(&t, &p->y, &q->y);
        if (fq2_is_zero(&t)) { memset(o, 0, sizeof *o); return; }
        orc_fq2 three;
        fq2_from_u64(&three, 3, 0);
        fq2_mul(&lam, &p->x, &p->x);
        fq2_mul(&lam, &lam, &three);
        fq2_inv(&t, &t);
        fq2_mul(&lam, &lam, &t);
    } else {
        fq2_sub(&lam, &q->y, &p->y);
        fq2_sub(&t, &q->x, &p->x);
        fq2_inv(&t, &t);
        fq2_mul(&lam, &lam, &t);
    }
    fq2_mul(&x3, &lam, &lam);
    fq2_sub(&x3, &x3, &p->x);
    fq2_sub(&x3, &x3, &q->x);
    fq2_sub(&t, &p->x, &x3);
    fq2_mul(&y3, &lam, &t);
    fq2_sub(&y3, &y3, &p->y);
    o->x = x3;
    o->y = y3;
}
void orc_g2a_mul(orc_g2a *o, const orc_g2a *p, const orc_fr *k) {
    uint64_t e[4];
    orc_fr_to_raw(e, k);
    orc_g2a acc, base = *p;
    memset(&acc, 0, sizeof acc);
    for (int i = 0; i < 256; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) orc_g2a_add(&acc, &acc, &base);
        orc_g2a_add(&base, &base, &base);
    }
    *o = acc;
}

/* ------------------------------------------------------------------ Fq12 = Fq2[w]/(w^6 - (9+i)) */
static void fq12_one(orc_fq12 *o) {
    memset(o, 0, sizeof *o);
    o->c[0].c0 = ORC_FQ_ONE;
}
void orc_fq12_mul(orc_fq12 *o, const orc_fq12 *a, const orc_fq12 *b) {
    orc_fq2 acc[11], t;
    memset(acc, 0, sizeof acc);
    for (int i = 0; i < 6; i++) {
        if (fq2_is_zero(&a->c[i])) continue; /* line functions are sparse */
        for (int j = 0; j < 6; j++) {
            fq2_mul(&t, &a->c[i], &b->c[j]);
            fq2_add(&acc[i + j], &acc[i + j], &t);
        }
    }
    for (int d = 10; d >= 6; d--) {
        fq2_mul_xi(&t, &acc[d]);
        fq2_add(&acc[d - 6], &acc[d - 6], &t);
    }
    memcpy(o->c, acc, sizeof o->c);
}
int orc_fq12_eq(const orc_fq12 *a, const orc_fq12 *b) {
    for (int i = 0; i < 6; i++)
        if (!fq2_eq(&a->c[i], &b->c[i])) return 0;
    return 1;
}
int orc_fq12_is_one(const orc_fq12 *a) {
    orc_fq12 one;
    fq12_one(&one);
    return orc_fq12_eq(a, &one);
}
static void fq12_pow_words(orc_fq12 *o, const orc_fq12 *a, const uint64_t *e, int nwords) {
    orc_fq12 acc, base = *a;
    fq12_one(&acc);
    int top = nwords * 64 - 1;
    while (top >= 0 && !((e[top / 64] >> (top % 64)) & 1)) top--;
    for (int i = top; i >= 0; i--) {
        orc_fq12_mul(&acc, &acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) orc_fq12_mul(&acc, &acc, &base);
    }
    *o = acc;
}
void orc_fq12_pow_fr(orc_fq12 *o, const orc_fq12 *a, const orc_fr *k) {
    uint64_t e[4];
    orc_fr_to_raw(e, k);
    fq12_pow_words(o, a, e, 4);
}

/* (q^12 - 1) / r, little-endian words (Python: (Q**12 - 1) // R) */
static const uint64_t FINAL_EXP[44] = {
    0x86964b64ca86f120ULL, 0x40a4efb7e54523a4ULL, 0x837fa97896e84abbULL, 0x361102b6b9b2b918ULL,
    0xc0de81def35692daULL, 0xbe04c7e8a6c3c760ULL, 0xd766f9c9d570bb7fULL, 0xc230974d83561841ULL,
    0x5bba1668c3be69a3ULL, 0x7f3811c410526294ULL, 0x29baee7ddadda71cULL, 0xbf813b8d145da900ULL,
    0x641bbadf423f9a2cULL, 0xa80bb4ea44eacc5eULL, 0xcd65664814fde37cULL, 0x4a0364b9580291d2ULL,
    0xee93dfb10826f0ddULL, 0x6b42db8dc5514724ULL, 0xbb10cf430b0f3785ULL, 0x40494e406f804216ULL,
    0x55cfe107acf3aafbULL, 0x2088ec80e0ebae87ULL, 0x846a3ed011a337a0ULL, 0x48a45a4a1e3a5195ULL,
    0xe5664568dfc50e16ULL, 0xab6a41294c0cc4ebULL, 0x82d0d602d268c7daULL, 0x6668449aed3cc48aULL,
    0x5062cd0fb2015dfcULL, 0x7f2940a8b1ddb3d1ULL, 0x77f5b63a2a226448ULL, 0xfef0781361e443aeULL,
    0xf977870e88d5c6c8ULL, 0x790364a61f676baaULL, 0x5887e72eceaddea3ULL, 0x1377e563a09a1b70ULL,
    0x0c54efee1bd8c3b2ULL, 0x3ec3d15ad524d8f7ULL, 0xdaf15466b2383a5dULL, 0xe1e30a73bb94fec0ULL,
    0x6a1c71015f3f7be2ULL, 0x842d43bf6369b1ffULL, 0x20fddadf107d20bcULL, 0x0000002f4b6dc970ULL,
};

/* ------------------------------------------------------------------ Miller loop f_{r,P}(Q)
 * P = (xP, yP) in E(Fq); Q = (x' w^2, y' w^3) is the untwisted image of (x', y') in E'(Fq2).
 * The line through T with slope lam, evaluated at Q:  (yQ - yT) - lam (xQ - xT)
 *   = (lam xT - yT)  +  (-lam x') w^2  +  y' w^3.
 * Vertical lines take values in the subfield Fq2[w^2] and die in the final exponentiation, so they are
 * skipped (denominator elimination); that includes the last addition T + P = O at T = -P. */
static void line_eval(orc_fq12 *l, const orc_fq *lam, const orc_fq *xt, const orc_fq *yt, const orc_g2a *q) {
    orc_fq t, nlam;
    memset(l, 0, sizeof *l);
    orc_fq_mul(&t, lam, xt);
    orc_fq_sub(&l->c[0].c0, &t, yt);
    orc_fq_neg(&nlam, lam);
    fq2_mul_fq(&l->c[2], &q->x, &nlam);
    l->c[3] = q->y;
}

static void miller(orc_fq12 *f, const orc_g1a *p, const orc_g2a *q) {
    fq12_one(f);
    if ((orc_fq_is_zero(&p->x) && orc_fq_is_zero(&p->y)) || orc_g2a_is_identity(q)) return;
    orc_fq xt = p->x, yt = p->y, lam, t, x3, y3, three;
    orc_fq12 l;
    orc_fq_from_u64(&three, 3);
    int inf = 0; /* T = O */
    int top = 253;
    while (!((ORC_FR_MODULUS[top / 64] >> (top % 64)) & 1)) top--;
    for (int i = top - 1; i >= 0; i--) {
        orc_fq12_mul(f, f, f);
        if (!inf) {
            if (orc_fq_is_zero(&yt)) { /* 2-torsion cannot occur in a prime-order group; kept for safety */
                inf = 1;
            } else {
                orc_fq_sqr(&lam, &xt);
                orc_fq_mul(&lam, &lam, &three);
                orc_fq_add(&t, &yt, &yt);
                orc_fq_inv(&t, &t);
                orc_fq_mul(&lam, &lam, &t);
                line_eval(&l, &lam, &xt, &yt, q);
                orc_fq12_mul(f, f, &l);
                orc_fq_sqr(&x3, &lam);
                orc_fq_sub(&x3, &x3, &xt);
                orc_fq_sub(&x3, &x3, &xt);
                orc_fq_sub(&t, &xt, &x3);
                orc_fq_mul(&y3, &lam, &t);
                orc_fq_sub(&y3, &y3, &yt);
                xt = x3;
                yt = y3;
            }
        }
        if ((ORC_FR_MODULUS[i / 64] >> (i % 64)) & 1) {
            if (inf) {
                xt = p->x; yt = p->y; inf = 0;
            } else if (orc_fq_eq(&xt, &p->x)) {
                orc_fq_add(&t, &yt, &p->y);
                if (orc_fq_is_zero(&t)) {
                    inf = 1; /* vertical line: skipped */
                } else {     /* T = P: tangent (cannot happen mid-loop for ord(P) = r; kept for safety) */
                    orc_fq_sqr(&lam, &xt);
                    orc_fq_mul(&lam, &lam, &three);
                    orc_fq_inv(&t, &t);
                    orc_fq_mul(&lam, &lam, &t);
                    goto add_step;
                }
            } else {
                orc_fq_sub(&lam, &p->y, &yt);
                orc_fq_sub(&t, &p->x, &xt);
                orc_fq_inv(&t, &t);
                orc_fq_mul(&lam, &lam, &t);
            add_step:
                line_eval(&l, &lam, &xt, &yt, q);
                orc_fq12_mul(f, f, &l);
                orc_fq_sqr(&x3, &lam);
                orc_fq_sub(&x3, &x3, &xt);
                orc_fq_sub(&x3, &x3, &p->x);
                orc_fq_sub(&t, &xt, &x3);
                orc_fq_mul(&y3, &lam, &t);
                orc_fq_sub(&y3, &y3, &yt);
                xt = x3;
                yt = y3;
            }
        }
    }
}

void orc_pairing(orc_fq12 *o, const orc_g1a *p, const orc_g2a *q) {
    orc_fq12 f;
    miller(&f, p, q);
    fq12_pow_words(o, &f, FINAL_EXP, 44);
}

int orc_pairing_check(const orc_g1a *p, const orc_g2a *q, size_t n) {
    orc_fq12 acc, f;
    fq12_one(&acc);
    for (size_t i = 0; i < n; i++) {
        miller(&f, &p[i], &q[i]);
        orc_fq12_mul(&acc, &acc, &f);
    }
    fq12_pow_words(&f, &acc, FINAL_EXP, 44);
    return orc_fq12_is_one(&f);
}
