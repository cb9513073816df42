/*
 * oracle/poly.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see poly.h header).
 */
#include "poly.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ MSM */
void orc_msm_naive(orc_g1 *out, const orc_fr *scalars, const orc_g1a *bases, size_t n) {
    orc_g1 acc;
    orc_g1_identity(&acc);
    for (size_t i = 0; i < n; i++) {
        orc_g1 p, t;
        orc_g1_from_affine(&p, &bases[i]);
        orc_g1_mul(&t, &p, &scalars[i]);
        orc_g1_add(&acc, &acc, &t);
    }
    *out = acc;
}

/* halo2 arithmetic.rs `get_at`: c-bit digit number `segment` of the 32-byte LE repr */
static inline size_t get_at(size_t segment, size_t c, const uint8_t *bytes) {
    size_t skip_bits = segment * c, skip_bytes = skip_bits / 8;
    if (skip_bytes >= 32) return 0;
    uint64_t tmp = 0;
    for (size_t i = 0; i < 8 && skip_bytes + i < 32; i++) tmp |= (uint64_t)bytes[skip_bytes + i] << (8 * i);
    tmp >>= skip_bits - skip_bytes * 8;
    return (size_t)(tmp % ((uint64_t)1 << c));
}

static size_t msm_window(size_t n) {
    if (n < 4) return 1;
    if (n < 32) return 3;
    return (size_t)ceil(log((double)n));
}

/* halo2 `multiexp_serial`; accumulates into *acc like upstream */
static void multiexp_serial(orc_g1 *acc, const orc_fr *scalars, const orc_g1a *bases, size_t n) {
    uint8_t *repr = (uint8_t *)malloc(n * 32);
    for (size_t i = 0; i < n; i++) {
        uint64_t v[4];
        orc_fr_to_raw(v, &scalars[i]);
        memcpy(repr + 32 * i, v, 32); /* little-endian host: limbs are the LE byte repr */
    }
    size_t c = msm_window(n);
    size_t segments = 256 / c + 1;
    size_t nb = ((size_t)1 << c) - 1;
    orc_g1 *buckets = (orc_g1 *)malloc(nb * sizeof(orc_g1));
    for (size_t seg = segments; seg-- > 0;) {
        for (size_t i = 0; i < c; i++) orc_g1_double(acc, acc);
        for (size_t b = 0; b < nb; b++) orc_g1_identity(&buckets[b]);
        for (size_t i = 0; i < n; i++) {
            size_t d = get_at(seg, c, repr + 32 * i);
            if (d != 0) orc_g1_add_mixed(&buckets[d - 1], &buckets[d - 1], &bases[i]);
        }
        orc_g1 running;
        orc_g1_identity(&running);
        for (size_t b = nb; b-- > 0;) {
            orc_g1_add(&running, &buckets[b], &running);
            orc_g1_add(acc, acc, &running);
        }
    }
    free(buckets);
    free(repr);
}

void orc_msm(orc_g1 *out, const orc_fr *scalars, const orc_g1a *bases, size_t n) {
    orc_g1 acc;
    orc_g1_identity(&acc);
    multiexp_serial(&acc, scalars, bases, n);
    *out = acc;
}

void orc_msm_mt(orc_g1 *out, const orc_fr *scalars, const orc_g1a *bases, size_t n, int threads) {
    if (threads < 1) threads = 1;
    if (n <= (size_t)threads) { orc_msm(out, scalars, bases, n); return; }
    size_t chunk = n / (size_t)threads; /* best_multiexp: chunk = len / num_threads */
    size_t nchunks = (n + chunk - 1) / chunk;
    orc_g1 *res = (orc_g1 *)malloc(nchunks * sizeof(orc_g1));
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (long ci = 0; ci < (long)nchunks; ci++) {
        size_t lo = (size_t)ci * chunk, hi = lo + chunk > n ? n : lo + chunk;
        orc_g1_identity(&res[ci]);
        multiexp_serial(&res[ci], scalars + lo, bases + lo, hi - lo);
    }
    orc_g1 acc;
    orc_g1_identity(&acc);
    for (size_t ci = 0; ci < nchunks; ci++) orc_g1_add(&acc, &acc, &res[ci]);
    free(res);
    *out = acc;
}

/* ------------------------------------------------------------------ FFT */
static inline uint32_t bitreverse(uint32_t n, uint32_t l) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < l; i++) {
        r = (r << 1) | (n & 1);
        n >>= 1;
    }
    return r;
}

void orc_fft(orc_fr *a, const orc_fr *omega, uint32_t log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t k = 0; k < n; k++) {
        size_t rk = bitreverse((uint32_t)k, log_n);
        if (k < rk) {
            orc_fr t = a[rk];
            a[rk] = a[k];
            a[k] = t;
        }
    }
    if (n < 2) return;
    orc_fr *tw = (orc_fr *)malloc((n / 2) * sizeof(orc_fr));
    orc_fr w = ORC_FR_ONE;
    for (size_t i = 0; i < n / 2; i++) {
        tw[i] = w;
        orc_fr_mul(&w, &w, omega);
    }
    size_t chunk = 2, twiddle_chunk = n / 2;
    for (uint32_t s = 0; s < log_n; s++) {
        /* n/2 independent butterflies per stage (halo2 spreads them over rayon threads) */
        const size_t half = chunk / 2;
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (long bf = 0; bf < (long)(n / 2); bf++) {
            size_t blk = (size_t)bf / half, i = (size_t)bf % half;
            orc_fr *left = a + blk * chunk, *right = left + half;
            orc_fr t = right[i];
            if (i != 0) orc_fr_mul(&t, &t, &tw[i * twiddle_chunk]);
            orc_fr_sub(&right[i], &left[i], &t);
            orc_fr_add(&left[i], &left[i], &t);
        }
        chunk *= 2;
        twiddle_chunk /= 2;
    }
    free(tw);
}

void orc_dft_naive(orc_fr *out, const orc_fr *a, const orc_fr *omega, uint32_t log_n) {
    size_t n = (size_t)1 << log_n;
    orc_fr wk = ORC_FR_ONE; /* omega^k */
    for (size_t k = 0; k < n; k++) {
        orc_fr acc = ORC_FR_ZERO, w = ORC_FR_ONE;
        for (size_t j = 0; j < n; j++) {
            orc_fr t;
            orc_fr_mul(&t, &a[j], &w);
            orc_fr_add(&acc, &acc, &t);
            orc_fr_mul(&w, &w, &wk);
        }
        out[k] = acc;
        orc_fr_mul(&wk, &wk, omega);
    }
}

void orc_eval_poly(orc_fr *out, const orc_fr *coeffs, size_t n, const orc_fr *x) {
    orc_fr acc = ORC_FR_ZERO;
    for (size_t i = n; i-- > 0;) {
        orc_fr_mul(&acc, &acc, x);
        orc_fr_add(&acc, &acc, &coeffs[i]);
    }
    *out = acc;
}

/* halo2 `kate_division`: divide a(X) by (X - b), discarding the remainder. */
void orc_kate_division(orc_fr *q, const orc_fr *a, size_t n, const orc_fr *b) {
    if (n < 2) return;
    orc_fr nb, tmp = ORC_FR_ZERO;
    orc_fr_neg(&nb, b); /* upstream: b = -b; q_i = a_{i+1} - tmp; tmp = q_i * b */
    for (size_t i = n - 1; i-- > 0;) {
        orc_fr lead;
        orc_fr_sub(&lead, &a[i + 1], &tmp);
        q[i] = lead;
        orc_fr_mul(&tmp, &lead, &nb);
    }
}

/* ------------------------------------------------------------------ EvaluationDomain */
void orc_domain_new(orc_domain *d, uint32_t j, uint32_t k) {
    memset(d, 0, sizeof(*d));
    d->k = k;
    d->n = (uint64_t)1 << k;
    d->quotient_poly_degree = j - 1;
    uint32_t ek = k;
    while (((uint64_t)1 << ek) < d->n * d->quotient_poly_degree) ek++;
    d->extended_k = ek;
    d->extended_n = (uint64_t)1 << ek;
    orc_fr eo = ORC_FR_ROOT_OF_UNITY;
    for (uint32_t i = ek; i < 28; i++) orc_fr_sqr(&eo, &eo);
    d->extended_omega = eo;
    orc_fr_inv(&d->extended_omega_inv, &eo);
    orc_fr o = eo;
    for (uint32_t i = k; i < ek; i++) orc_fr_sqr(&o, &o);
    d->omega = o;
    orc_fr_inv(&d->omega_inv, &o);
    d->g_coset = ORC_FR_ZETA;
    orc_fr_sqr(&d->g_coset_inv, &ORC_FR_ZETA);
    orc_fr t;
    orc_fr_from_u64(&t, d->n);
    orc_fr_inv(&d->ifft_divisor, &t);
    d->barycentric_weight = d->ifft_divisor;
    orc_fr_from_u64(&t, d->extended_n);
    orc_fr_inv(&d->extended_ifft_divisor, &t);
    /* t_evaluations: ((zeta * ext_omega^i)^n - 1)^-1 for one period */
    d->t_len = (size_t)1 << (ek - k);
    d->t_evaluations = (orc_fr *)malloc(d->t_len * sizeof(orc_fr));
    orc_fr orig, step, cur;
    orc_fr_pow_u64(&orig, &ORC_FR_ZETA, d->n);
    orc_fr_pow_u64(&step, &eo, d->n);
    cur = orig;
    for (size_t i = 0; i < d->t_len; i++) {
        orc_fr_sub(&d->t_evaluations[i], &cur, &ORC_FR_ONE);
        orc_fr_mul(&cur, &cur, &step);
    }
    orc_fr_batch_inv(d->t_evaluations, d->t_len);
}

void orc_domain_free(orc_domain *d) {
    free(d->t_evaluations);
    d->t_evaluations = NULL;
}

static void ifft(orc_fr *a, const orc_fr *omega_inv, uint32_t log_n, const orc_fr *divisor) {
    orc_fft(a, omega_inv, log_n);
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) orc_fr_mul(&a[i], &a[i], divisor);
}

void orc_lagrange_to_coeff(const orc_domain *d, orc_fr *a) { ifft(a, &d->omega_inv, d->k, &d->ifft_divisor); }
void orc_coeff_to_lagrange(const orc_domain *d, orc_fr *a) { orc_fft(a, &d->omega, d->k); }

/* `distribute_powers_zeta`: a[i] *= zeta^(i mod 3) (into coset) or zeta^-(i mod 3) */
static void distribute_powers_zeta(const orc_domain *d, orc_fr *a, size_t len, int into_coset) {
    const orc_fr *p1 = into_coset ? &d->g_coset : &d->g_coset_inv;
    const orc_fr *p2 = into_coset ? &d->g_coset_inv : &d->g_coset;
    for (size_t i = 0; i < len; i++) {
        size_t m = i % 3;
        if (m == 1) orc_fr_mul(&a[i], &a[i], p1);
        else if (m == 2) orc_fr_mul(&a[i], &a[i], p2);
    }
}

void orc_coeff_to_extended(const orc_domain *d, orc_fr *out, const orc_fr *coeffs) {
    memcpy(out, coeffs, d->n * sizeof(orc_fr));
    memset(out + d->n, 0, (d->extended_n - d->n) * sizeof(orc_fr));
    distribute_powers_zeta(d, out, d->n, 1);
    orc_fft(out, &d->extended_omega, d->extended_k);
}

void orc_extended_to_coeff(const orc_domain *d, orc_fr *out, orc_fr *a) {
    ifft(a, &d->extended_omega_inv, d->extended_k, &d->extended_ifft_divisor);
    distribute_powers_zeta(d, a, d->extended_n, 0);
    memcpy(out, a, d->n * d->quotient_poly_degree * sizeof(orc_fr));
}

void orc_divide_by_vanishing(const orc_domain *d, orc_fr *a) {
    for (size_t i = 0; i < d->extended_n; i++) orc_fr_mul(&a[i], &a[i], &d->t_evaluations[i % d->t_len]);
}

void orc_rotate_omega(const orc_domain *d, orc_fr *out, const orc_fr *x, int32_t rotation) {
    orc_fr p;
    if (rotation >= 0) orc_fr_pow_u64(&p, &d->omega, (uint64_t)rotation);
    else orc_fr_pow_u64(&p, &d->omega_inv, (uint64_t)(-(int64_t)rotation));
    orc_fr_mul(out, x, &p);
}

/* ------------------------------------------------------------------ SRS */
#define FB_WINDOWS 32
#define FB_ENTRIES 255
static orc_g1a *fb_table = NULL; /* [32][255]: d * 2^(8w) * G */

static void fb_build(void) {
    orc_g1 *tmp = (orc_g1 *)malloc(FB_WINDOWS * FB_ENTRIES * sizeof(orc_g1));
    orc_g1 base;
    orc_g1_generator(&base);
    for (int w = 0; w < FB_WINDOWS; w++) {
        orc_g1 acc = base;
        for (int dgt = 0; dgt < FB_ENTRIES; dgt++) {
            tmp[w * FB_ENTRIES + dgt] = acc;
            orc_g1_add(&acc, &acc, &base);
        }
        base = acc; /* 256 * previous base */
    }
    orc_g1a *tab = (orc_g1a *)malloc(FB_WINDOWS * FB_ENTRIES * sizeof(orc_g1a));
    orc_g1_batch_to_affine(tab, tmp, FB_WINDOWS * FB_ENTRIES);
    free(tmp);
    fb_table = tab;
}

void orc_fixed_base_mul(orc_g1a *out, const orc_fr *scalars, size_t n) {
#pragma omp critical(orc_fb_build)
    {
        if (!fb_table) fb_build();
    }
    orc_g1 *proj = (orc_g1 *)malloc(n * sizeof(orc_g1));
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) {
        uint64_t v[4];
        orc_fr_to_raw(v, &scalars[i]);
        const uint8_t *bytes = (const uint8_t *)v;
        orc_g1 acc;
        orc_g1_identity(&acc);
        for (int w = 0; w < FB_WINDOWS; w++)
            if (bytes[w]) orc_g1_add_mixed(&acc, &acc, &fb_table[w * FB_ENTRIES + bytes[w] - 1]);
        proj[i] = acc;
    }
    orc_g1_batch_to_affine(out, proj, n);
    free(proj);
}

/* ParamsKZG::setup: g[i] = s^i G ; g_lagrange[i] = L_i(s) G with
 * L_i(s) = (s^n - 1)/n * omega^i / (s - omega^i) */
void orc_params_new(orc_params *p, uint32_t k, const orc_fr *s) {
    p->k = k;
    p->n = (uint64_t)1 << k;
    p->s = *s;
    orc_g2_generator(&p->g2);
    orc_g2a_mul(&p->s_g2, &p->g2, s);
    size_t n = (size_t)p->n;
    p->g = (orc_g1a *)malloc(n * sizeof(orc_g1a));
    p->g_lagrange = (orc_g1a *)malloc(n * sizeof(orc_g1a));
    orc_fr *sc = (orc_fr *)malloc(n * sizeof(orc_fr));
    orc_fr cur = ORC_FR_ONE;
    for (size_t i = 0; i < n; i++) {
        sc[i] = cur;
        orc_fr_mul(&cur, &cur, s);
    }
    orc_fixed_base_mul(p->g, sc, n);
    /* cur == s^n now */
    orc_fr root = ORC_FR_ROOT_OF_UNITY, n_inv, mult, t;
    for (uint32_t i = k; i < 28; i++) orc_fr_sqr(&root, &root);
    orc_fr_from_u64(&t, p->n);
    orc_fr_inv(&n_inv, &t);
    orc_fr_sub(&mult, &cur, &ORC_FR_ONE);
    orc_fr_mul(&mult, &mult, &n_inv);
    orc_fr *den = (orc_fr *)malloc(n * sizeof(orc_fr));
    orc_fr rp = ORC_FR_ONE;
    for (size_t i = 0; i < n; i++) {
        orc_fr_mul(&sc[i], &mult, &rp);
        orc_fr_sub(&den[i], s, &rp);
        orc_fr_mul(&rp, &rp, &root);
    }
    orc_fr_batch_inv(den, n);
    for (size_t i = 0; i < n; i++) orc_fr_mul(&sc[i], &sc[i], &den[i]);
    orc_fixed_base_mul(p->g_lagrange, sc, n);
    free(den);
    free(sc);
}

void orc_params_free(orc_params *p) {
    free(p->g);
    free(p->g_lagrange);
    p->g = p->g_lagrange = NULL;
}

static int commit_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_commit(const orc_params *p, orc_g1a *out, const orc_fr *coeffs) {
    orc_g1 r;
    orc_msm_mt(&r, coeffs, p->g, (size_t)p->n, commit_threads());
    orc_g1_to_affine(out, &r);
}

void orc_commit_lagrange(const orc_params *p, orc_g1a *out, const orc_fr *evals) {
    orc_g1 r;
    orc_msm_mt(&r, evals, p->g_lagrange, (size_t)p->n, commit_threads());
    orc_g1_to_affine(out, &r);
}
