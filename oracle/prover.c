/*
 * oracle/prover.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see prover.h for provenance and the
 * "parity unpinned" statement).  Straight-line, single-threaded where order matters; OpenMP only on
 * row loops so that it can double as the timed CPU baseline.
 */
#include "prover.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ randomness */
/* ChaCha20 block function (RFC 7539 section 2.3: 32-byte key, 32-bit block counter, 96-bit nonce). */
static inline uint32_t rotl32(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
#define ORC_QR(a, b, c, d)                                       \
    do {                                                         \
        a += b; d ^= a; d = rotl32(d, 16);                       \
        c += d; b ^= c; b = rotl32(b, 12);                       \
        a += b; d ^= a; d = rotl32(d, 8);                        \
        c += d; b ^= c; b = rotl32(b, 7);                        \
    } while (0)

void orc_chacha20_block(const uint8_t key[32], uint32_t counter, const uint32_t nonce[3], uint32_t out[16]) {
    uint32_t st[16], x[16];
    st[0] = 0x61707865u; st[1] = 0x3320646eu; st[2] = 0x79622d32u; st[3] = 0x6b206574u;
    for (int i = 0; i < 8; i++)
        st[4 + i] = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) | ((uint32_t)key[4 * i + 2] << 16) |
                    ((uint32_t)key[4 * i + 3] << 24);
    st[12] = counter;
    st[13] = nonce[0]; st[14] = nonce[1]; st[15] = nonce[2];
    memcpy(x, st, sizeof(x));
    for (int r = 0; r < 10; r++) {
        ORC_QR(x[0], x[4], x[8], x[12]);
        ORC_QR(x[1], x[5], x[9], x[13]);
        ORC_QR(x[2], x[6], x[10], x[14]);
        ORC_QR(x[3], x[7], x[11], x[15]);
        ORC_QR(x[0], x[5], x[10], x[15]);
        ORC_QR(x[1], x[6], x[11], x[12]);
        ORC_QR(x[2], x[7], x[8], x[13]);
        ORC_QR(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + st[i];
}

/* Blinding scalar (key, tag, index): ChaCha20 keystream with nonce = (tag, index_lo, index_hi), block counter =
 * attempt; each 64-byte block offers two 254-bit candidates (words 0-7, then 8-15, top word masked to 30 bits,
 * little-endian); the first one below r is taken (acceptance 0.76 per candidate). */
void orc_rand_fr(orc_fr *out, const uint8_t key[32], uint32_t tag, uint64_t index) {
    const uint32_t nonce[3] = {tag, (uint32_t)index, (uint32_t)(index >> 32)};
    uint64_t v[4];
    for (uint32_t attempt = 0;; attempt++) {
        uint32_t blk[16];
        orc_chacha20_block(key, attempt, nonce, blk);
        for (int half = 0; half < 2; half++) {
            const uint32_t *w = blk + 8 * half;
            for (int i = 0; i < 4; i++) v[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
            v[3] &= 0x3fffffffffffffffULL;
            int lt = 0;
            for (int i = 3; i >= 0; i--) {
                if (v[i] < ORC_FR_MODULUS[i]) { lt = 1; break; }
                if (v[i] > ORC_FR_MODULUS[i]) break;
            }
            if (lt) {
                orc_fr_from_raw(out, v);
                return;
            }
        }
    }
}

/* ------------------------------------------------------------------ EvmTranscript (snark-verifier) */
typedef struct {
    uint8_t *buf;  /* pending hash input  */
    size_t len, cap;
    uint8_t *out;  /* proof stream        */
    size_t olen, ocap;
    const uint8_t *in; /* reader mode     */
    size_t ilen, ipos;
    int overflow;
} transcript;

static void tr_init(transcript *t, uint8_t *out, size_t ocap) {
    memset(t, 0, sizeof(*t));
    t->cap = 1 << 16;
    t->buf = (uint8_t *)malloc(t->cap);
    t->out = out;
    t->ocap = ocap;
}
static void tr_free(transcript *t) { free(t->buf); }
static void tr_absorb(transcript *t, const uint8_t *b, size_t n) {
    if (t->len + n > t->cap) {
        t->cap = 2 * (t->len + n);
        t->buf = (uint8_t *)realloc(t->buf, t->cap);
    }
    memcpy(t->buf + t->len, b, n);
    t->len += n;
}
static void tr_emit(transcript *t, const uint8_t *b, size_t n) {
    if (!t->out) return;
    if (t->olen + n > t->ocap) { t->overflow = 1; return; }
    memcpy(t->out + t->olen, b, n);
    t->olen += n;
}
static void tr_common_scalar(transcript *t, const orc_fr *s) {
    uint8_t b[32];
    orc_fr_to_be_bytes(b, s);
    tr_absorb(t, b, 32);
}
/* identity points cannot be absorbed (EvmTranscript errors); callers never produce them */
static int tr_common_point(transcript *t, const orc_g1a *p) {
    if (orc_fq_is_zero(&p->x) && orc_fq_is_zero(&p->y)) return -1;
    uint8_t b[64];
    orc_fq_to_be_bytes(b, &p->x);
    orc_fq_to_be_bytes(b + 32, &p->y);
    tr_absorb(t, b, 64);
    return 0;
}
static int tr_write_point(transcript *t, const orc_g1a *p) {
    if (tr_common_point(t, p)) return -1;
    uint8_t b[64];
    orc_fq_to_be_bytes(b, &p->x);
    orc_fq_to_be_bytes(b + 32, &p->y);
    tr_emit(t, b, 64);
    return 0;
}
static void tr_write_scalar(transcript *t, const orc_fr *s) {
    tr_common_scalar(t, s);
    uint8_t b[32];
    orc_fr_to_be_bytes(b, s);
    tr_emit(t, b, 32);
}
/* squeeze: hash = keccak256(buf ++ (len == 32 ? [1] : [])); buf = hash; challenge = hash mod r */
static void tr_squeeze(transcript *t, orc_fr *c) {
    uint8_t h[32];
    if (t->len == 32) {
        uint8_t one = 1;
        tr_absorb(t, &one, 1);
    }
    orc_keccak256(t->buf, t->len, h);
    memcpy(t->buf, h, 32);
    t->len = 32;
    orc_fr_from_be_bytes_reduce(c, h);
}
static int tr_read_point(transcript *t, orc_g1a *p) {
    if (t->ipos + 64 > t->ilen) return -1;
    const uint8_t *b = t->in + t->ipos;
    t->ipos += 64;
    uint64_t v[4];
    for (int c = 0; c < 2; c++) {
        memset(v, 0, sizeof(v));
        for (int i = 0; i < 32; i++) v[(31 - i) / 8] |= (uint64_t)b[32 * c + i] << (8 * ((31 - i) % 8));
        /* coordinates must be canonical (< q) */
        int lt = 0;
        for (int i = 3; i >= 0; i--) {
            if (v[i] < ORC_FQ_MODULUS[i]) { lt = 1; break; }
            if (v[i] > ORC_FQ_MODULUS[i]) break;
        }
        if (!lt) return -1;
        orc_fq_from_raw(c == 0 ? &p->x : &p->y, v);
    }
    if (!orc_g1a_on_curve(p)) return -1;
    return tr_common_point(t, p);
}
static int tr_read_scalar(transcript *t, orc_fr *s) {
    if (t->ipos + 32 > t->ilen) return -1;
    const uint8_t *b = t->in + t->ipos;
    t->ipos += 32;
    uint64_t v[4] = {0, 0, 0, 0};
    for (int i = 0; i < 32; i++) v[(31 - i) / 8] |= (uint64_t)b[i] << (8 * ((31 - i) % 8));
    int lt = 0;
    for (int i = 3; i >= 0; i--) {
        if (v[i] < ORC_FR_MODULUS[i]) { lt = 1; break; }
        if (v[i] > ORC_FR_MODULUS[i]) break;
    }
    if (!lt) return -1;
    orc_fr_from_raw(s, v);
    tr_common_scalar(t, s);
    return 0;
}

/* ------------------------------------------------------------------ expression evaluation */
/* value of polynomial `p` at `row` of arrays of length `size`, rotations scaled by rot_scale: sum over its monomials of
 * coeff * product of the queried cells.  The monomials of a polynomial, their factors sorted by query index, form a trie
 * whose node is the product of the cells on its path -- a cell product that several monomials begin with is taken once (the
 * twelve gates of WnnCircuit: 414 products per row as a flat list, 135 as tries).  halo2's GraphEvaluator shares the nodes of
 * the expression graph in the same spirit (plonk/evaluation.rs); the field element is the same either way. */
#define PLAN_NONE 0xffffffffu
typedef struct {
    uint32_t n_nodes, n_terms;
    uint32_t *parent, *query;   /* [n_nodes]: node = parent's product (or 1) times cell(query) */
    uint32_t *term_node;        /* [n_terms]: the node a monomial ends at (PLAN_NONE: a constant) */
    uint8_t *term_kind;         /* 0: +1, 1: -1, 2: coeff */
    const zg_monomial **term_mono;
} poly_plan;

static void poly_plan_build(poly_plan *pl, const zg_circuit *cs, const zg_poly *p) {
    uint32_t cap = 1;
    for (uint32_t m = p->first; m < p->first + p->count; m++) cap += cs->monomials[m].n_factors;
    pl->parent = (uint32_t *)malloc(cap * sizeof(uint32_t));
    pl->query = (uint32_t *)malloc(cap * sizeof(uint32_t));
    pl->term_node = (uint32_t *)malloc((p->count ? p->count : 1) * sizeof(uint32_t));
    pl->term_kind = (uint8_t *)malloc(p->count ? p->count : 1);
    pl->term_mono = (const zg_monomial **)malloc((p->count ? p->count : 1) * sizeof(void *));
    pl->n_nodes = 0;
    pl->n_terms = p->count;
    orc_fr minus_one;
    orc_fr_neg(&minus_one, &ORC_FR_ONE);
    for (uint32_t t = 0; t < p->count; t++) {
        const zg_monomial *mo = &cs->monomials[p->first + t];
        uint32_t f[64], nf = mo->n_factors;
        for (uint32_t i = 0; i < nf; i++) { /* insertion sort by query index */
            uint32_t v = mo->factors[i], j = i;
            while (j > 0 && f[j - 1] > v) { f[j] = f[j - 1]; j--; }
            f[j] = v;
        }
        uint32_t node = PLAN_NONE;
        for (uint32_t i = 0; i < nf; i++) {
            uint32_t found = PLAN_NONE;
            for (uint32_t c = 0; c < pl->n_nodes; c++)
                if (pl->parent[c] == node && pl->query[c] == f[i]) { found = c; break; }
            if (found == PLAN_NONE) {
                found = pl->n_nodes++;
                pl->parent[found] = node;
                pl->query[found] = f[i];
            }
            node = found;
        }
        pl->term_node[t] = node;
        pl->term_mono[t] = mo;
        pl->term_kind[t] = memcmp(&mo->coeff, &ORC_FR_ONE, 32) == 0 ? 0 : memcmp(&mo->coeff, &minus_one, 32) == 0 ? 1 : 2;
    }
}
static void poly_plan_free(poly_plan *pl) {
    free(pl->parent); free(pl->query); free(pl->term_node); free(pl->term_kind); free((void *)pl->term_mono);
}

static void eval_plan_row(orc_fr *out, const zg_circuit *cs, const poly_plan *pl, const orc_fr *fixed, const orc_fr *advice,
                          const orc_fr *instance, size_t size, size_t row, int64_t rot_scale) {
    orc_fr val[pl->n_nodes ? pl->n_nodes : 1];
    for (uint32_t i = 0; i < pl->n_nodes; i++) {
        const zg_query *q = &cs->queries[pl->query[i]];
        int64_t idx = ((int64_t)row + (int64_t)q->rotation * rot_scale) % (int64_t)size;
        if (idx < 0) idx += (int64_t)size;
        const orc_fr *col = q->kind == ZG_FIXED ? fixed : q->kind == ZG_ADVICE ? advice : instance;
        const orc_fr *v = &col[(size_t)q->column * size + (size_t)idx];
        if (pl->parent[i] == PLAN_NONE) val[i] = *v;
        else orc_fr_mul(&val[i], &val[pl->parent[i]], v);
    }
    orc_fr acc = ORC_FR_ZERO, prod;
    for (uint32_t t = 0; t < pl->n_terms; t++) {
        const orc_fr *x = pl->term_node[t] == PLAN_NONE ? &ORC_FR_ONE : &val[pl->term_node[t]];
        if (pl->term_kind[t] == 0) orc_fr_add(&acc, &acc, x);
        else if (pl->term_kind[t] == 1) orc_fr_sub(&acc, &acc, x);
        else {
            memcpy(&prod, &pl->term_mono[t]->coeff, 32);
            if (pl->term_node[t] != PLAN_NONE) orc_fr_mul(&prod, &prod, x);
            orc_fr_add(&acc, &acc, &prod);
        }
    }
    *out = acc;
}

void orc_grand_product(orc_fr *z, const orc_fr *num, const orc_fr *den, const orc_fr *z0, size_t n) {
    orc_fr *inv = (orc_fr *)malloc(n * sizeof(orc_fr));
    memcpy(inv, den, n * sizeof(orc_fr));
    orc_fr_batch_inv(inv, n);
    z[0] = *z0;
    for (size_t i = 0; i + 1 < n; i++) {
        orc_fr t;
        orc_fr_mul(&t, &num[i], &inv[i]);
        orc_fr_mul(&z[i + 1], &z[i], &t);
    }
    free(inv);
}

static int cmp_raw(const void *a, const void *b) {
    const uint64_t *x = (const uint64_t *)a, *y = (const uint64_t *)b;
    for (int i = 3; i >= 0; i--) {
        if (x[i] < y[i]) return -1;
        if (x[i] > y[i]) return 1;
    }
    return 0;
}

/* lookup::prover::permute_expression_pair (without the blinding tail). in/out are Montgomery. */
static int permute_expression_pair(orc_fr *pin, orc_fr *ptab, const orc_fr *input, const orc_fr *table,
                                   size_t usable) {
    uint64_t(*a)[4] = malloc(usable * 32);
    uint64_t(*t)[4] = malloc(usable * 32);
    for (size_t i = 0; i < usable; i++) {
        orc_fr_to_raw(a[i], &input[i]);
        orc_fr_to_raw(t[i], &table[i]);
    }
    qsort(a, usable, 32, cmp_raw); /* permuted_input_expression.sort() : Fr::cmp = canonical order */
    qsort(t, usable, 32, cmp_raw); /* BTreeMap<value, count> in ascending key order */
    /* unique table values with counts */
    size_t nu = 0;
    uint32_t *cnt = (uint32_t *)calloc(usable ? usable : 1, sizeof(uint32_t));
    for (size_t i = 0; i < usable; i++) {
        if (i == 0 || cmp_raw(t[i], t[nu - 1]) != 0) {
            memcpy(t[nu], t[i], 32);
            cnt[nu] = 1;
            nu++;
        } else {
            cnt[nu - 1]++;
        }
    }
    size_t *repeated = (size_t *)malloc((usable ? usable : 1) * sizeof(size_t));
    size_t nrep = 0;
    uint64_t(*pt)[4] = calloc(usable ? usable : 1, 32);
    int status = 0;
    for (size_t row = 0; row < usable; row++) {
        if (row == 0 || cmp_raw(a[row], a[row - 1]) != 0) {
            memcpy(pt[row], a[row], 32);
            /* remove one instance of the value from the leftover map */
            size_t lo = 0, hi = nu;
            while (lo < hi) {
                size_t mid = (lo + hi) / 2;
                if (cmp_raw(t[mid], a[row]) < 0) lo = mid + 1;
                else hi = mid;
            }
            if (lo == nu || cmp_raw(t[lo], a[row]) != 0) { status = ZG_ERR_CONSTRAINT; break; }
            /* upstream asserts count > 0: an exhausted entry stays in the map with count 0 */
            if (cnt[lo] == 0) { status = ZG_ERR_CONSTRAINT; break; }
            cnt[lo]--;
        } else {
            repeated[nrep++] = row;
        }
    }
    if (status == 0) {
        for (size_t u = 0; u < nu; u++)
            for (uint32_t c = 0; c < cnt[u]; c++) {
                if (nrep == 0) { status = ZG_ERR_INVALID_ARG; break; }
                memcpy(pt[repeated[--nrep]], t[u], 32);
            }
        if (nrep != 0) status = ZG_ERR_INVALID_ARG;
    }
    if (status == 0)
        for (size_t i = 0; i < usable; i++) {
            orc_fr_from_raw(&pin[i], a[i]);
            orc_fr_from_raw(&ptab[i], pt[i]);
        }
    free(a); free(t); free(cnt); free(repeated); free(pt);
    return status;
}

void orc_trace_free(orc_trace *t) {
    free(t->h_ext); free(t->perm_z); free(t->lookup_z); free(t->permuted_input); free(t->permuted_table);
    free(t->h_pieces);
    memset(t, 0, sizeof(*t));
}

static size_t n_perm_sets(const zg_circuit *cs) {
    if (cs->n_perm_columns == 0) return 0;
    size_t chunk = cs->cs_degree - 2;
    return (cs->n_perm_columns + chunk - 1) / chunk;
}

size_t orc_proof_size(const zg_circuit *cs) {
    size_t sets = n_perm_sets(cs);
    size_t qpd = cs->cs_degree - 1;
    size_t points = cs->n_advice + 2 * cs->n_lookups + sets + cs->n_lookups + 1 + qpd;
    size_t scalars = cs->n_advice_queries + cs->n_fixed_queries + 1 + cs->n_perm_columns +
                     (sets ? 3 * sets - 1 : 0) + 5 * cs->n_lookups;
    /* GWC: one W per distinct opening point (at most: x, and one per distinct rotation) */
    size_t max_points = 2 + cs->n_advice_queries + cs->n_fixed_queries;
    return 64 * (points + max_points) + 32 * scalars;
}

typedef struct {
    orc_fr point;
    const orc_fr *poly; /* coefficient form, n entries */
    orc_fr eval;
} query;

/* ------------------------------------------------------------------ create_proof */
/* Wall-clock milliseconds the LAST orc_create_proof of this thread spent per phase, in the slots of the product's
 * zg_prover_phase_ms (0 instance + advice commitments, 1 permuted lookups, 2 products + random polynomial, 3 coefficient
 * forms + evaluate_h + h commitments, 4 evaluations, 5 GWC openings, 6 total): bench.py's cpu_baseline prints them
 * beside the GPU's. */
#include <time.h>
static __thread double g_phase_ms[8];
static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
void orc_last_phase_ms(double *out, size_t cap) {
    for (size_t i = 0; i < cap && i < 8; i++) out[i] = g_phase_ms[i];
}
#define PHASE_LAP(slot)                          \
    do {                                         \
        const double now_ = now_ms();            \
        g_phase_ms[slot] = now_ - t_prev_;       \
        t_prev_ = now_;                          \
    } while (0)

/* What keygen_pk stores in the ProvingKey beside the Lagrange values (halo2_proofs src/plonk/keygen.rs keygen_pk:
 * fixed_polys, fixed_cosets, permutation::ProvingKey {polys, cosets}, l0, l_last, l_active_row): a function of the key
 * alone.  orc_pk_derive fills pk->derived once; a key without it (derived == NULL) has the same material computed inside
 * every create_proof -- the bytes of the proof are the same either way (tests/test_oracle_prover.py). */
typedef struct {
    orc_fr *fix_poly, *sig_poly;          /* [F][n], [P][n] coefficient forms          */
    orc_fr *fix_cos, *sig_cos;            /* [F][en], [P][en] on the extended coset    */
    orc_fr *l0, *llast, *lactive;         /* [en]                                      */
} pk_derived;

static void transform_columns(const orc_domain *dom, orc_fr *cos, orc_fr *poly, size_t count, int lagrange_input);

static pk_derived *derive_key_material(const orc_pk *pk) {
    const zg_circuit *cs = pk->cs;
    const size_t n = (size_t)1 << cs->k, F = cs->n_fixed, P = cs->n_perm_columns, bf = cs->blinding_factors;
    orc_domain dom;
    orc_domain_new(&dom, cs->cs_degree, cs->k);
    const size_t en = (size_t)dom.extended_n;
    pk_derived *d = (pk_derived *)calloc(1, sizeof(pk_derived));
    d->fix_poly = (orc_fr *)malloc((F ? F : 1) * n * sizeof(orc_fr));
    d->sig_poly = (orc_fr *)malloc((P ? P : 1) * n * sizeof(orc_fr));
    d->fix_cos = (orc_fr *)malloc((F ? F : 1) * en * sizeof(orc_fr));
    d->sig_cos = (orc_fr *)malloc((P ? P : 1) * en * sizeof(orc_fr));
    memcpy(d->fix_poly, pk->fixed_values, F * n * sizeof(orc_fr));
    memcpy(d->sig_poly, pk->sigma_values, P * n * sizeof(orc_fr));
    transform_columns(&dom, d->fix_cos, d->fix_poly, F, 1);
    transform_columns(&dom, d->sig_cos, d->sig_poly, P, 1);
    /* l_0, l_last, l_active_row on the coset */
    orc_fr *l = (orc_fr *)calloc(3 * n, sizeof(orc_fr)), *lc = (orc_fr *)malloc(3 * en * sizeof(orc_fr));
    l[0] = ORC_FR_ONE;
    l[n + n - bf - 1] = ORC_FR_ONE;
    for (size_t j = 0; j < bf; j++) l[2 * n + n - 1 - j] = ORC_FR_ONE; /* l_blind */
    transform_columns(&dom, lc, l, 3, 1);
    d->l0 = (orc_fr *)malloc(en * sizeof(orc_fr));
    d->llast = (orc_fr *)malloc(en * sizeof(orc_fr));
    d->lactive = (orc_fr *)malloc(en * sizeof(orc_fr));
    memcpy(d->l0, lc, en * sizeof(orc_fr));
    memcpy(d->llast, lc + en, en * sizeof(orc_fr));
    for (size_t i = 0; i < en; i++) {
        orc_fr t;
        orc_fr_add(&t, &lc[en + i], &lc[2 * en + i]);
        orc_fr_sub(&d->lactive[i], &ORC_FR_ONE, &t);
    }
    free(l);
    free(lc);
    orc_domain_free(&dom);
    return d;
}
static void free_key_material(pk_derived *d) {
    if (!d) return;
    free(d->fix_poly); free(d->sig_poly); free(d->fix_cos); free(d->sig_cos);
    free(d->l0); free(d->llast); free(d->lactive);
    free(d);
}
void orc_pk_derive(orc_pk *pk) {
    if (!pk->derived) pk->derived = derive_key_material(pk);
}
void orc_pk_release(orc_pk *pk) {
    free_key_material((pk_derived *)pk->derived);
    pk->derived = NULL;
}

/* `count` columns of n values each to coefficient form in place (when lagrange_input) and on to the extended coset
 * (when cos != NULL): one column per thread -- the transforms of different columns share nothing (halo2 runs each
 * transform's butterflies over its rayon pool instead; the values are the same). */
static void transform_columns(const orc_domain *dom, orc_fr *cos, orc_fr *poly, size_t count, int lagrange_input) {
    const size_t n = (size_t)dom->n, en = (size_t)dom->extended_n;
#pragma omp parallel for schedule(dynamic, 1) if (count > 1)
    for (long c = 0; c < (long)count; c++) {
        if (lagrange_input) orc_lagrange_to_coeff(dom, poly + (size_t)c * n);
        if (cos) orc_coeff_to_extended(dom, cos + (size_t)c * en, poly + (size_t)c * n);
    }
}

int orc_create_proof(const orc_pk *pk, const orc_fr *advice_in, const orc_fr *instance_in, size_t instance_len,
                     const uint8_t seed[32], uint8_t *proof, size_t cap, size_t *proof_len, orc_trace *trace) {
    const zg_circuit *cs = pk->cs;
    const size_t n = (size_t)1 << cs->k;
    const size_t bf = cs->blinding_factors;
    const size_t usable = n - (bf + 1);
    const size_t A = cs->n_advice, F = cs->n_fixed, I = cs->n_instance, P = cs->n_perm_columns;
    const size_t NL = cs->n_lookups;
    if (instance_len > usable) return ZG_ERR_INVALID_ARG; /* Error::InstanceTooLarge */
    orc_domain dom;
    orc_domain_new(&dom, cs->cs_degree, cs->k);
    const size_t en = (size_t)dom.extended_n;
    const size_t qpd = dom.quotient_poly_degree;
    const int64_t rs = (int64_t)(en / n); /* rotation scale on the extended domain */
    int status = 0;
    const double t_start_ = now_ms();
    double t_prev_ = t_start_;
    /* the polynomials this proof evaluates row by row (gates; the lookups' input and table expressions), as tries */
    size_t n_lk_polys = 0;
    for (size_t l = 0; l < NL; l++) n_lk_polys += 2 * (size_t)cs->lookups[l].width;
    poly_plan *gate_plan = (poly_plan *)calloc(cs->n_gates ? cs->n_gates : 1, sizeof(poly_plan));
    poly_plan *lk_plan = (poly_plan *)calloc(n_lk_polys ? n_lk_polys : 1, sizeof(poly_plan));
    size_t *lk_plan_first = (size_t *)calloc(NL ? NL : 1, sizeof(size_t)); /* lookup l: inputs at [first + e], tables at [first + width + e] */
    for (uint32_t g = 0; g < cs->n_gates; g++) poly_plan_build(&gate_plan[g], cs, &cs->gates[g]);
    for (size_t l = 0, at = 0; l < NL; l++) {
        const zg_lookup *lk = &cs->lookups[l];
        lk_plan_first[l] = at;
        for (uint32_t e = 0; e < lk->width; e++) {
            poly_plan_build(&lk_plan[at + e], cs, &lk->inputs[e]);
            poly_plan_build(&lk_plan[at + lk->width + e], cs, &lk->tables[e]);
        }
        at += 2 * (size_t)lk->width;
    }

    transcript tr;
    tr_init(&tr, proof, cap);
    tr_common_scalar(&tr, &pk->vk_repr); /* vk.hash_into */

    /* ---- instance: values hashed (KZG: QUERY_INSTANCE = false), padded, iFFT */
    orc_fr *inst_val = (orc_fr *)calloc((I ? I : 1) * n, sizeof(orc_fr));
    for (size_t c = 0; c < I; c++)
        for (size_t i = 0; i < instance_len; i++) {
            tr_common_scalar(&tr, &instance_in[c * instance_len + i]);
            inst_val[c * n + i] = instance_in[c * instance_len + i];
        }
    orc_fr *inst_poly = (orc_fr *)malloc((I ? I : 1) * n * sizeof(orc_fr));
    memcpy(inst_poly, inst_val, I * n * sizeof(orc_fr));
    for (size_t c = 0; c < I; c++) orc_lagrange_to_coeff(&dom, inst_poly + c * n);

    /* ---- advice: blind the last bf+1 rows, commit in the Lagrange basis */
    orc_fr *adv_val = (orc_fr *)malloc((A ? A : 1) * n * sizeof(orc_fr));
    memcpy(adv_val, advice_in, A * n * sizeof(orc_fr));
    for (size_t c = 0; c < A; c++)
        for (size_t j = 0; j < bf + 1; j++)
            orc_rand_fr(&adv_val[c * n + usable + j], seed, ORC_TAG_ADVICE_BLIND, c * (bf + 1) + j);
    for (size_t c = 0; c < A; c++) {
        orc_g1a cm;
        orc_commit_lagrange(pk->params, &cm, adv_val + c * n);
        if (tr_write_point(&tr, &cm)) status = ZG_ERR_INVALID_ARG;
    }
    orc_fr theta;
    tr_squeeze(&tr, &theta);
    PHASE_LAP(0);

    /* ---- lookups: commit_permuted */
    orc_fr *cin = (orc_fr *)malloc((NL ? NL : 1) * n * sizeof(orc_fr));  /* compressed input  */
    orc_fr *ctab = (orc_fr *)malloc((NL ? NL : 1) * n * sizeof(orc_fr)); /* compressed table  */
    orc_fr *pin = (orc_fr *)malloc((NL ? NL : 1) * n * sizeof(orc_fr));  /* permuted input a' */
    orc_fr *ptab = (orc_fr *)malloc((NL ? NL : 1) * n * sizeof(orc_fr)); /* permuted table s' */
    for (size_t l = 0; l < NL && status == 0; l++) {
        const zg_lookup *lk = &cs->lookups[l];
#pragma omp parallel for schedule(static)
        for (long row = 0; row < (long)n; row++) {
            orc_fr ai = ORC_FR_ZERO, ti = ORC_FR_ZERO, v;
            for (uint32_t e = 0; e < lk->width; e++) {
                eval_plan_row(&v, cs, &lk_plan[lk_plan_first[l] + e], pk->fixed_values, adv_val, inst_val, n, (size_t)row, 1);
                orc_fr_mul(&ai, &ai, &theta);
                orc_fr_add(&ai, &ai, &v);
                eval_plan_row(&v, cs, &lk_plan[lk_plan_first[l] + lk->width + e], pk->fixed_values, adv_val, inst_val, n, (size_t)row, 1);
                orc_fr_mul(&ti, &ti, &theta);
                orc_fr_add(&ti, &ti, &v);
            }
            cin[l * n + row] = ai;
            ctab[l * n + row] = ti;
        }
        status = permute_expression_pair(pin + l * n, ptab + l * n, cin + l * n, ctab + l * n, usable);
        if (status) break;
        for (size_t j = 0; j < bf + 1; j++) {
            orc_rand_fr(&pin[l * n + usable + j], seed, ORC_TAG_PERMUTED_INPUT, l * (bf + 1) + j);
            orc_rand_fr(&ptab[l * n + usable + j], seed, ORC_TAG_PERMUTED_TABLE, l * (bf + 1) + j);
        }
        orc_g1a ci, ct;
        orc_commit_lagrange(pk->params, &ci, pin + l * n);
        orc_commit_lagrange(pk->params, &ct, ptab + l * n);
        if (tr_write_point(&tr, &ci) || tr_write_point(&tr, &ct)) status = ZG_ERR_INVALID_ARG;
    }
    if (status) goto fail_early;

    orc_fr beta, gamma;
    tr_squeeze(&tr, &beta);
    tr_squeeze(&tr, &gamma);
    PHASE_LAP(1);

    /* ---- permutation argument: permutation::prover::commit */
    const size_t chunk = cs->cs_degree - 2;
    const size_t sets = n_perm_sets(cs);
    orc_fr *pz = (orc_fr *)malloc((sets ? sets : 1) * n * sizeof(orc_fr));
    {
        orc_fr deltaomega = ORC_FR_ONE, last_z = ORC_FR_ONE;
        orc_fr *num = (orc_fr *)malloc(n * sizeof(orc_fr));
        orc_fr *den = (orc_fr *)malloc(n * sizeof(orc_fr));
        for (size_t s = 0; s < sets; s++) {
            size_t c0 = s * chunk, c1 = c0 + chunk > P ? P : c0 + chunk;
            for (size_t i = 0; i < n; i++) { num[i] = ORC_FR_ONE; den[i] = ORC_FR_ONE; }
            for (size_t c = c0; c < c1; c++) {
                const zg_query *col = &cs->perm_columns[c];
                const orc_fr *vals = (col->kind == ZG_FIXED ? pk->fixed_values
                                      : col->kind == ZG_ADVICE ? adv_val : inst_val) + (size_t)col->column * n;
                const orc_fr *sig = pk->sigma_values + c * n;
                orc_fr dw = deltaomega;
                for (size_t i = 0; i < n; i++) {
                    orc_fr t;
                    orc_fr_mul(&t, &beta, &sig[i]);
                    orc_fr_add(&t, &t, &gamma);
                    orc_fr_add(&t, &t, &vals[i]);
                    orc_fr_mul(&den[i], &den[i], &t);
                    orc_fr_mul(&t, &dw, &beta);
                    orc_fr_add(&t, &t, &gamma);
                    orc_fr_add(&t, &t, &vals[i]);
                    orc_fr_mul(&num[i], &num[i], &t);
                    orc_fr_mul(&dw, &dw, &dom.omega);
                }
                orc_fr_mul(&deltaomega, &deltaomega, &ORC_FR_DELTA);
            }
            orc_fr *z = pz + s * n;
            orc_grand_product(z, num, den, &last_z, n);
            for (size_t j = 0; j < bf; j++) orc_rand_fr(&z[n - bf + j], seed, ORC_TAG_PERM_Z, s * bf + j);
            last_z = z[n - (bf + 1)];
            orc_g1a cm;
            orc_commit_lagrange(pk->params, &cm, z);
            if (tr_write_point(&tr, &cm)) status = ZG_ERR_INVALID_ARG;
        }
        free(num);
        free(den);
    }

    /* ---- lookups: commit_product */
    orc_fr *lz = (orc_fr *)malloc((NL ? NL : 1) * n * sizeof(orc_fr));
    {
        orc_fr *num = (orc_fr *)malloc(n * sizeof(orc_fr));
        orc_fr *den = (orc_fr *)malloc(n * sizeof(orc_fr));
        for (size_t l = 0; l < NL; l++) {
            for (size_t i = 0; i < n; i++) {
                orc_fr a, b;
                orc_fr_add(&a, &pin[l * n + i], &beta);
                orc_fr_add(&b, &ptab[l * n + i], &gamma);
                orc_fr_mul(&den[i], &a, &b);
                orc_fr_add(&a, &cin[l * n + i], &beta);
                orc_fr_add(&b, &ctab[l * n + i], &gamma);
                orc_fr_mul(&num[i], &a, &b);
            }
            orc_fr *z = lz + l * n;
            orc_grand_product(z, num, den, &ORC_FR_ONE, n);
            for (size_t j = 0; j < bf; j++) orc_rand_fr(&z[n - bf + j], seed, ORC_TAG_LOOKUP_Z, l * bf + j);
            orc_g1a cm;
            orc_commit_lagrange(pk->params, &cm, z);
            if (tr_write_point(&tr, &cm)) status = ZG_ERR_INVALID_ARG;
        }
        free(num);
        free(den);
    }

    /* ---- vanishing::Argument::commit: random polynomial, coefficient basis */
    orc_fr *random_poly = (orc_fr *)malloc(n * sizeof(orc_fr));
    for (size_t i = 0; i < n; i++) orc_rand_fr(&random_poly[i], seed, ORC_TAG_RANDOM_POLY, i);
    {
        orc_g1a cm;
        orc_commit(pk->params, &cm, random_poly);
        if (tr_write_point(&tr, &cm)) status = ZG_ERR_INVALID_ARG;
    }
    orc_fr y;
    tr_squeeze(&tr, &y);
    PHASE_LAP(2);

    /* ---- coefficient forms */
    pk_derived *own = pk->derived ? NULL : derive_key_material(pk);
    const pk_derived *km = pk->derived ? (const pk_derived *)pk->derived : own;
    const orc_fr *fix_poly = km->fix_poly, *sig_poly = km->sig_poly;
    /* every column this proof made, side by side: advice | permutation z | lookup z | permuted inputs | permuted tables */
    const size_t n_own = A + sets + 3 * NL;
    orc_fr *own_poly = (orc_fr *)malloc((n_own ? n_own : 1) * n * sizeof(orc_fr));
    orc_fr *adv_poly = own_poly, *pz_poly = adv_poly + A * n, *lz_poly = pz_poly + sets * n;
    orc_fr *pin_poly = lz_poly + NL * n, *ptab_poly = pin_poly + NL * n;
    memcpy(adv_poly, adv_val, A * n * sizeof(orc_fr));
    memcpy(pz_poly, pz, sets * n * sizeof(orc_fr));
    memcpy(lz_poly, lz, NL * n * sizeof(orc_fr));
    memcpy(pin_poly, pin, NL * n * sizeof(orc_fr));
    memcpy(ptab_poly, ptab, NL * n * sizeof(orc_fr));
    /* ---- evaluate_h on the extended coset */
    orc_fr *own_cos = (orc_fr *)malloc((n_own ? n_own : 1) * en * sizeof(orc_fr));
    orc_fr *adv_cos = own_cos, *pz_cos = adv_cos + A * en, *lz_cos = pz_cos + sets * en;
    orc_fr *pin_cos = lz_cos + NL * en, *ptab_cos = pin_cos + NL * en;
    transform_columns(&dom, own_cos, own_poly, n_own, 1);
    orc_fr *inst_cos = (orc_fr *)malloc((I ? I : 1) * en * sizeof(orc_fr));
    transform_columns(&dom, inst_cos, inst_poly, I, 0);
    const orc_fr *fix_cos = km->fix_cos, *sig_cos = km->sig_cos;
    const orc_fr *l0 = km->l0, *llast = km->llast, *lactive = km->lactive;
    orc_fr *h = (orc_fr *)malloc(en * sizeof(orc_fr));
    {
        /* beta_term = extended_omega^idx, delta_start = beta * ZETA */
        orc_fr *eo_pow = (orc_fr *)malloc(en * sizeof(orc_fr));
        eo_pow[0] = ORC_FR_ONE;
        for (size_t i = 1; i < en; i++) orc_fr_mul(&eo_pow[i], &eo_pow[i - 1], &dom.extended_omega);
        orc_fr delta_start;
        orc_fr_mul(&delta_start, &beta, &ORC_FR_ZETA);
        const int64_t last_rot = -(int64_t)(bf + 1);
#pragma omp parallel for schedule(static)
        for (long ii = 0; ii < (long)en; ii++) {
            const size_t idx = (size_t)ii;
            const size_t r_next = (idx + (size_t)rs) % en;
            const size_t r_prev = (idx + en - (size_t)rs) % en;
            const size_t r_last = (size_t)(((int64_t)idx + last_rot * rs) % (int64_t)en + (int64_t)en) % en;
            orc_fr value = ORC_FR_ZERO, t, u;
            /* custom gates */
            for (uint32_t g = 0; g < cs->n_gates; g++) {
                eval_plan_row(&t, cs, &gate_plan[g], fix_cos, adv_cos, inst_cos, en, idx, rs);
                orc_fr_mul(&value, &value, &y);
                orc_fr_add(&value, &value, &t);
            }
            /* permutation argument */
            if (sets > 0) {
                const orc_fr *zf = pz_cos, *zl = pz_cos + (sets - 1) * en;
                orc_fr_sub(&t, &ORC_FR_ONE, &zf[idx]);
                orc_fr_mul(&t, &t, &l0[idx]);
                orc_fr_mul(&value, &value, &y);
                orc_fr_add(&value, &value, &t);
                orc_fr_sqr(&t, &zl[idx]);
                orc_fr_sub(&t, &t, &zl[idx]);
                orc_fr_mul(&t, &t, &llast[idx]);
                orc_fr_mul(&value, &value, &y);
                orc_fr_add(&value, &value, &t);
                for (size_t s = 1; s < sets; s++) {
                    orc_fr_sub(&t, &pz_cos[s * en + idx], &pz_cos[(s - 1) * en + r_last]);
                    orc_fr_mul(&t, &t, &l0[idx]);
                    orc_fr_mul(&value, &value, &y);
                    orc_fr_add(&value, &value, &t);
                }
                orc_fr current_delta;
                orc_fr_mul(&current_delta, &delta_start, &eo_pow[idx]);
                for (size_t s = 0; s < sets; s++) {
                    size_t c0 = s * chunk, c1 = c0 + chunk > P ? P : c0 + chunk;
                    orc_fr left = pz_cos[s * en + r_next], right = pz_cos[s * en + idx];
                    for (size_t c = c0; c < c1; c++) {
                        const zg_query *col = &cs->perm_columns[c];
                        const orc_fr *vals = (col->kind == ZG_FIXED ? fix_cos
                                              : col->kind == ZG_ADVICE ? adv_cos : inst_cos) + (size_t)col->column * en;
                        orc_fr_mul(&t, &beta, &sig_cos[c * en + idx]);
                        orc_fr_add(&t, &t, &vals[idx]);
                        orc_fr_add(&t, &t, &gamma);
                        orc_fr_mul(&left, &left, &t);
                        orc_fr_add(&u, &vals[idx], &current_delta);
                        orc_fr_add(&u, &u, &gamma);
                        orc_fr_mul(&right, &right, &u);
                        orc_fr_mul(&current_delta, &current_delta, &ORC_FR_DELTA);
                    }
                    orc_fr_sub(&t, &left, &right);
                    orc_fr_mul(&t, &t, &lactive[idx]);
                    orc_fr_mul(&value, &value, &y);
                    orc_fr_add(&value, &value, &t);
                }
            }
            /* lookups */
            for (size_t l = 0; l < NL; l++) {
                const zg_lookup *lk = &cs->lookups[l];
                orc_fr ai = ORC_FR_ZERO, ti = ORC_FR_ZERO, v;
                for (uint32_t e = 0; e < lk->width; e++) {
                    eval_plan_row(&v, cs, &lk_plan[lk_plan_first[l] + e], fix_cos, adv_cos, inst_cos, en, idx, rs);
                    orc_fr_mul(&ai, &ai, &theta);
                    orc_fr_add(&ai, &ai, &v);
                    eval_plan_row(&v, cs, &lk_plan[lk_plan_first[l] + lk->width + e], fix_cos, adv_cos, inst_cos, en, idx, rs);
                    orc_fr_mul(&ti, &ti, &theta);
                    orc_fr_add(&ti, &ti, &v);
                }
                const orc_fr *z = lz_cos + l * en, *ap = pin_cos + l * en, *sp = ptab_cos + l * en;
                /* l_0 (1 - z) */
                orc_fr_sub(&t, &ORC_FR_ONE, &z[idx]);
                orc_fr_mul(&t, &t, &l0[idx]);
                orc_fr_mul(&value, &value, &y);
                orc_fr_add(&value, &value, &t);
                /* l_last (z^2 - z) */
                orc_fr_sqr(&t, &z[idx]);
                orc_fr_sub(&t, &t, &z[idx]);
                orc_fr_mul(&t, &t, &llast[idx]);
                orc_fr_mul(&value, &value, &y);
                orc_fr_add(&value, &value, &t);
                /* l_active (z(wX)(a'+beta)(s'+gamma) - z(X)(a+beta)(s+gamma)) */
                orc_fr lft, rgt;
                orc_fr_add(&t, &ap[idx], &beta);
                orc_fr_add(&u, &sp[idx], &gamma);
                orc_fr_mul(&lft, &t, &u);
                orc_fr_mul(&lft, &lft, &z[r_next]);
                orc_fr_add(&t, &ai, &beta);
                orc_fr_add(&u, &ti, &gamma);
                orc_fr_mul(&rgt, &t, &u);
                orc_fr_mul(&rgt, &rgt, &z[idx]);
                orc_fr_sub(&t, &lft, &rgt);
                orc_fr_mul(&t, &t, &lactive[idx]);
                orc_fr_mul(&value, &value, &y);
                orc_fr_add(&value, &value, &t);
                /* l_0 (a' - s') */
                orc_fr ams;
                orc_fr_sub(&ams, &ap[idx], &sp[idx]);
                orc_fr_mul(&t, &ams, &l0[idx]);
                orc_fr_mul(&value, &value, &y);
                orc_fr_add(&value, &value, &t);
                /* l_active (a' - s')(a' - a'(w^-1 X)) */
                orc_fr_sub(&t, &ap[idx], &ap[r_prev]);
                orc_fr_mul(&t, &t, &ams);
                orc_fr_mul(&t, &t, &lactive[idx]);
                orc_fr_mul(&value, &value, &y);
                orc_fr_add(&value, &value, &t);
            }
            h[idx] = value;
        }
        free(eo_pow);
    }
    /* ---- vanishing::construct: divide by X^n - 1, back to coefficients, split, commit */
    orc_divide_by_vanishing(&dom, h);
    if (trace) { /* h on the extended coset AFTER the division (what the GPU kernel stores) */
        trace->h_ext = (orc_fr *)malloc(en * sizeof(orc_fr));
        memcpy(trace->h_ext, h, en * sizeof(orc_fr));
    }
    orc_fr *h_coeff = (orc_fr *)malloc(qpd * n * sizeof(orc_fr));
    orc_extended_to_coeff(&dom, h_coeff, h);
    for (size_t i = 0; i < qpd; i++) {
        orc_g1a cm;
        orc_commit(pk->params, &cm, h_coeff + i * n);
        if (tr_write_point(&tr, &cm)) status = ZG_ERR_INVALID_ARG;
    }
    orc_fr x, xn;
    tr_squeeze(&tr, &x);
    PHASE_LAP(3);
    orc_fr_pow_u64(&xn, &x, (uint64_t)n);

    /* ---- evaluations */
    size_t max_q = A + F + P + 8 * (sets + NL) + cs->n_advice_queries + cs->n_fixed_queries + 8;
    query *qs = (query *)malloc(max_q * sizeof(query));
    size_t nq = 0;
    orc_fr x_next, x_inv, x_last;
    orc_rotate_omega(&dom, &x_next, &x, 1);
    orc_rotate_omega(&dom, &x_inv, &x, -1);
    orc_rotate_omega(&dom, &x_last, &x, -(int32_t)(bf + 1));
#define ADD_QUERY(pt, pl)                                      \
    do {                                                       \
        qs[nq].point = (pt);                                   \
        qs[nq].poly = (pl);                                    \
        orc_eval_poly(&qs[nq].eval, (pl), n, &qs[nq].point);   \
        nq++;                                                  \
    } while (0)
    /* advice evals */
    for (uint32_t i = 0; i < cs->n_advice_queries; i++) {
        orc_fr pt;
        orc_rotate_omega(&dom, &pt, &x, cs->advice_queries[i].rotation);
        ADD_QUERY(pt, adv_poly + (size_t)cs->advice_queries[i].column * n);
        tr_write_scalar(&tr, &qs[nq - 1].eval);
    }
    size_t q_after_advice = nq;
    /* fixed evals (queries re-ordered below: transcript order != query order) */
    query *fq = (query *)malloc((cs->n_fixed_queries + 1) * sizeof(query));
    for (uint32_t i = 0; i < cs->n_fixed_queries; i++) {
        orc_rotate_omega(&dom, &fq[i].point, &x, cs->fixed_queries[i].rotation);
        fq[i].poly = fix_poly + (size_t)cs->fixed_queries[i].column * n;
        orc_eval_poly(&fq[i].eval, fq[i].poly, n, &fq[i].point);
        tr_write_scalar(&tr, &fq[i].eval);
    }
    /* vanishing.evaluate: h(X) = sum_i xn^i h_i(X); random_eval */
    orc_fr *h_poly = (orc_fr *)calloc(n, sizeof(orc_fr));
    for (size_t i = qpd; i-- > 0;)
        for (size_t j = 0; j < n; j++) {
            orc_fr_mul(&h_poly[j], &h_poly[j], &xn);
            orc_fr_add(&h_poly[j], &h_poly[j], &h_coeff[i * n + j]);
        }
    orc_fr random_eval;
    orc_eval_poly(&random_eval, random_poly, n, &x);
    tr_write_scalar(&tr, &random_eval);
    /* pk.permutation.evaluate: sigma polys at x */
    query *sq = (query *)malloc((P + 1) * sizeof(query));
    for (size_t c = 0; c < P; c++) {
        sq[c].point = x;
        sq[c].poly = sig_poly + c * n;
        orc_eval_poly(&sq[c].eval, sq[c].poly, n, &x);
        tr_write_scalar(&tr, &sq[c].eval);
    }
    /* permutation z evals */
    orc_fr *pz_eval = (orc_fr *)malloc((3 * sets + 1) * sizeof(orc_fr));
    for (size_t s = 0; s < sets; s++) {
        orc_eval_poly(&pz_eval[3 * s], pz_poly + s * n, n, &x);
        orc_eval_poly(&pz_eval[3 * s + 1], pz_poly + s * n, n, &x_next);
        tr_write_scalar(&tr, &pz_eval[3 * s]);
        tr_write_scalar(&tr, &pz_eval[3 * s + 1]);
        if (s + 1 < sets) {
            orc_eval_poly(&pz_eval[3 * s + 2], pz_poly + s * n, n, &x_last);
            tr_write_scalar(&tr, &pz_eval[3 * s + 2]);
        }
    }
    /* lookup evals */
    orc_fr *lk_eval = (orc_fr *)malloc((5 * NL + 1) * sizeof(orc_fr));
    for (size_t l = 0; l < NL; l++) {
        orc_eval_poly(&lk_eval[5 * l + 0], lz_poly + l * n, n, &x);
        orc_eval_poly(&lk_eval[5 * l + 1], lz_poly + l * n, n, &x_next);
        orc_eval_poly(&lk_eval[5 * l + 2], pin_poly + l * n, n, &x);
        orc_eval_poly(&lk_eval[5 * l + 3], pin_poly + l * n, n, &x_inv);
        orc_eval_poly(&lk_eval[5 * l + 4], ptab_poly + l * n, n, &x);
        for (int e = 0; e < 5; e++) tr_write_scalar(&tr, &lk_eval[5 * l + e]);
    }
    /* ---- opening queries in create_proof's order */
    (void)q_after_advice;
    for (size_t s = 0; s < sets; s++) { /* permutation.open: (x, z), (x_next, z) per set */
        qs[nq].point = x; qs[nq].poly = pz_poly + s * n; qs[nq].eval = pz_eval[3 * s]; nq++;
        qs[nq].point = x_next; qs[nq].poly = pz_poly + s * n; qs[nq].eval = pz_eval[3 * s + 1]; nq++;
    }
    for (size_t s = sets; s-- > 0;) { /* then (x_last, z) for sets.rev().skip(1) */
        if (s + 1 == sets) continue;
        qs[nq].point = x_last; qs[nq].poly = pz_poly + s * n; qs[nq].eval = pz_eval[3 * s + 2]; nq++;
    }
    for (size_t l = 0; l < NL; l++) {
        qs[nq].point = x; qs[nq].poly = lz_poly + l * n; qs[nq].eval = lk_eval[5 * l + 0]; nq++;
        qs[nq].point = x; qs[nq].poly = pin_poly + l * n; qs[nq].eval = lk_eval[5 * l + 2]; nq++;
        qs[nq].point = x; qs[nq].poly = ptab_poly + l * n; qs[nq].eval = lk_eval[5 * l + 4]; nq++;
        qs[nq].point = x_inv; qs[nq].poly = pin_poly + l * n; qs[nq].eval = lk_eval[5 * l + 3]; nq++;
        qs[nq].point = x_next; qs[nq].poly = lz_poly + l * n; qs[nq].eval = lk_eval[5 * l + 1]; nq++;
    }
    for (uint32_t i = 0; i < cs->n_fixed_queries; i++) qs[nq++] = fq[i];
    for (size_t c = 0; c < P; c++) qs[nq++] = sq[c];
    ADD_QUERY(x, h_poly);
    qs[nq].point = x; qs[nq].poly = random_poly; qs[nq].eval = random_eval; nq++;

    /* ---- ProverGWC::create_proof */
    orc_fr v;
    tr_squeeze(&tr, &v);
    PHASE_LAP(4);
    {
        int *done = (int *)calloc(nq, sizeof(int));
        orc_fr *batch = (orc_fr *)malloc(n * sizeof(orc_fr));
        orc_fr *wit = (orc_fr *)calloc(n, sizeof(orc_fr));
        for (size_t first = 0; first < nq; first++) {
            if (done[first]) continue;
            orc_fr z = qs[first].point, eval_batch = ORC_FR_ZERO;
            memset(batch, 0, n * sizeof(orc_fr));
            for (size_t j = first; j < nq; j++) {
                if (done[j] || !orc_fr_eq(&qs[j].point, &z)) continue;
                done[j] = 1;
                for (size_t i = 0; i < n; i++) {
                    orc_fr_mul(&batch[i], &batch[i], &v);
                    orc_fr_add(&batch[i], &batch[i], &qs[j].poly[i]);
                }
                orc_fr_mul(&eval_batch, &eval_batch, &v);
                orc_fr_add(&eval_batch, &eval_batch, &qs[j].eval);
            }
            orc_fr_sub(&batch[0], &batch[0], &eval_batch);
            memset(wit, 0, n * sizeof(orc_fr));
            orc_kate_division(wit, batch, n, &z);
            orc_g1a w;
            orc_commit(pk->params, &w, wit); /* wit[n-1] = 0: same as committing n-1 coefficients */
            if (tr_write_point(&tr, &w)) status = ZG_ERR_INVALID_ARG;
        }
        free(done); free(batch); free(wit);
    }
    if (tr.overflow) status = ZG_ERR_INVALID_ARG;
    if (proof_len) *proof_len = tr.olen;

    if (trace) {
        trace->theta = theta; trace->beta = beta; trace->gamma = gamma; trace->y = y; trace->x = x; trace->v = v;
        trace->n_sets = (uint32_t)sets;
        trace->perm_z = pz; pz = NULL;
        trace->lookup_z = lz; lz = NULL;
        trace->permuted_input = pin; pin = NULL;
        trace->permuted_table = ptab; ptab = NULL;
        trace->h_pieces = h_coeff; h_coeff = NULL;
    }
    PHASE_LAP(5);
    g_phase_ms[6] = t_prev_ - t_start_;
    free(qs); free(fq); free(sq); free(pz_eval); free(lk_eval); free(h_poly); free(h_coeff); free(h);
    free(inst_cos); free(own_cos); free(own_poly);
    free_key_material(own);
    free(random_poly); free(lz); free(pz);
fail_early:
    for (uint32_t g = 0; g < cs->n_gates; g++) poly_plan_free(&gate_plan[g]);
    for (size_t i = 0; i < n_lk_polys; i++) poly_plan_free(&lk_plan[i]);
    free(gate_plan); free(lk_plan); free(lk_plan_first);
    free(cin); free(ctab); free(pin); free(ptab);
    free(adv_val); free(inst_val); free(inst_poly);
    tr_free(&tr);
    orc_domain_free(&dom);
    return status;
}

/* ------------------------------------------------------------------ verify_proof */
typedef struct {
    orc_fr point;
    orc_g1 commitment;
    orc_fr eval;
} vquery;

static void scalar_mul_g(orc_g1 *out, const orc_fr *k) {
    orc_g1 g;
    orc_g1_generator(&g);
    orc_g1_mul(out, &g, k);
}

static int verify_impl(const orc_pk *pk, const orc_fr *instance_in, size_t instance_len, const uint8_t *proof,
                       size_t proof_len, int use_pairing);

/* Test-only shortcut: the opening equations are checked in G1 with the known toxic scalar. */
int orc_verify_proof(const orc_pk *pk, const orc_fr *instance_in, size_t instance_len, const uint8_t *proof,
                     size_t proof_len) {
    return verify_impl(pk, instance_in, instance_len, proof, proof_len, 0);
}

/* The public verification equation (gwc/verifier.rs + strategy.rs `SingleStrategy`): after all W_i are
 * read, u is squeezed and  e(sum u^i W_i, [s]_2) == e(sum u^i (z_i W_i + C_i - e_i G), [1]_2)  is decided
 * with the pairing; only g2 / s_g2 of the parameters are used. */
int orc_verify_proof_pairing(const orc_pk *pk, const orc_fr *instance_in, size_t instance_len, const uint8_t *proof,
                             size_t proof_len) {
    return verify_impl(pk, instance_in, instance_len, proof, proof_len, 1);
}

static int verify_impl(const orc_pk *pk, const orc_fr *instance_in, size_t instance_len, const uint8_t *proof,
                       size_t proof_len, int use_pairing) {
    const zg_circuit *cs = pk->cs;
    const size_t n = (size_t)1 << cs->k;
    const size_t bf = cs->blinding_factors;
    const size_t A = cs->n_advice, F = cs->n_fixed, I = cs->n_instance, P = cs->n_perm_columns;
    const size_t NL = cs->n_lookups;
    const size_t chunk = cs->cs_degree - 2, sets = n_perm_sets(cs);
    orc_domain dom;
    orc_domain_new(&dom, cs->cs_degree, cs->k);
    const size_t qpd = dom.quotient_poly_degree;
    transcript tr;
    tr_init(&tr, NULL, 0);
    tr.in = proof;
    tr.ilen = proof_len;
    int ok = 1, bad = 0;
    tr_common_scalar(&tr, &pk->vk_repr);
    for (size_t c = 0; c < I; c++)
        for (size_t i = 0; i < instance_len; i++) tr_common_scalar(&tr, &instance_in[c * instance_len + i]);

    orc_g1a *adv_c = (orc_g1a *)malloc((A + 1) * sizeof(orc_g1a));
    for (size_t c = 0; c < A; c++) bad |= tr_read_point(&tr, &adv_c[c]);
    orc_fr theta, beta, gamma, y, x, v;
    tr_squeeze(&tr, &theta);
    orc_g1a *pin_c = (orc_g1a *)malloc((NL + 1) * sizeof(orc_g1a));
    orc_g1a *ptab_c = (orc_g1a *)malloc((NL + 1) * sizeof(orc_g1a));
    for (size_t l = 0; l < NL; l++) {
        bad |= tr_read_point(&tr, &pin_c[l]);
        bad |= tr_read_point(&tr, &ptab_c[l]);
    }
    tr_squeeze(&tr, &beta);
    tr_squeeze(&tr, &gamma);
    orc_g1a *pz_c = (orc_g1a *)malloc((sets + 1) * sizeof(orc_g1a));
    for (size_t s = 0; s < sets; s++) bad |= tr_read_point(&tr, &pz_c[s]);
    orc_g1a *lz_c = (orc_g1a *)malloc((NL + 1) * sizeof(orc_g1a));
    for (size_t l = 0; l < NL; l++) bad |= tr_read_point(&tr, &lz_c[l]);
    orc_g1a random_c;
    bad |= tr_read_point(&tr, &random_c);
    tr_squeeze(&tr, &y);
    orc_g1a *h_c = (orc_g1a *)malloc((qpd + 1) * sizeof(orc_g1a));
    for (size_t i = 0; i < qpd; i++) bad |= tr_read_point(&tr, &h_c[i]);
    tr_squeeze(&tr, &x);
    orc_fr xn;
    orc_fr_pow_u64(&xn, &x, (uint64_t)n);

    orc_fr *adv_e = (orc_fr *)malloc((cs->n_advice_queries + 1) * sizeof(orc_fr));
    orc_fr *fix_e = (orc_fr *)malloc((cs->n_fixed_queries + 1) * sizeof(orc_fr));
    for (uint32_t i = 0; i < cs->n_advice_queries; i++) bad |= tr_read_scalar(&tr, &adv_e[i]);
    for (uint32_t i = 0; i < cs->n_fixed_queries; i++) bad |= tr_read_scalar(&tr, &fix_e[i]);
    orc_fr random_eval;
    bad |= tr_read_scalar(&tr, &random_eval);
    orc_fr *sig_e = (orc_fr *)malloc((P + 1) * sizeof(orc_fr));
    for (size_t c = 0; c < P; c++) bad |= tr_read_scalar(&tr, &sig_e[c]);
    orc_fr *pz_e = (orc_fr *)calloc(3 * sets + 1, sizeof(orc_fr));
    for (size_t s = 0; s < sets; s++) {
        bad |= tr_read_scalar(&tr, &pz_e[3 * s]);
        bad |= tr_read_scalar(&tr, &pz_e[3 * s + 1]);
        if (s + 1 < sets) bad |= tr_read_scalar(&tr, &pz_e[3 * s + 2]);
    }
    orc_fr *lk_e = (orc_fr *)malloc((5 * NL + 1) * sizeof(orc_fr));
    for (size_t l = 0; l < NL; l++)
        for (int e = 0; e < 5; e++) bad |= tr_read_scalar(&tr, &lk_e[5 * l + e]);
    if (bad) { ok = -1; goto out; }

    /* instance evals by interpolation == evaluation of the instance polynomial at omega^rot x */
    orc_fr *inst_poly = (orc_fr *)calloc((I ? I : 1) * n, sizeof(orc_fr));
    for (size_t c = 0; c < I; c++) {
        for (size_t i = 0; i < instance_len; i++) inst_poly[c * n + i] = instance_in[c * instance_len + i];
        orc_lagrange_to_coeff(&dom, inst_poly + c * n);
    }
    /* per-query evaluation table */
    orc_fr *qe = (orc_fr *)malloc((cs->n_queries + 1) * sizeof(orc_fr));
    for (uint32_t q = 0; q < cs->n_queries && ok == 1; q++) {
        const zg_query *qq = &cs->queries[q];
        int found = 0;
        if (qq->kind == ZG_ADVICE) {
            for (uint32_t i = 0; i < cs->n_advice_queries; i++)
                if (cs->advice_queries[i].column == qq->column && cs->advice_queries[i].rotation == qq->rotation) {
                    qe[q] = adv_e[i];
                    found = 1;
                }
        } else if (qq->kind == ZG_FIXED) {
            for (uint32_t i = 0; i < cs->n_fixed_queries; i++)
                if (cs->fixed_queries[i].column == qq->column && cs->fixed_queries[i].rotation == qq->rotation) {
                    qe[q] = fix_e[i];
                    found = 1;
                }
        } else {
            orc_fr pt;
            orc_rotate_omega(&dom, &pt, &x, qq->rotation);
            orc_eval_poly(&qe[q], inst_poly + (size_t)qq->column * n, n, &pt);
            found = 1;
        }
        if (!found) ok = -2;
    }
    if (ok != 1) { free(qe); free(inst_poly); goto out; }
#define EVAL_POLY(dst, pp)                                                    \
    do {                                                                      \
        orc_fr _acc = ORC_FR_ZERO;                                            \
        for (uint32_t _m = (pp)->first; _m < (pp)->first + (pp)->count; _m++) { \
            const zg_monomial *_mo = &cs->monomials[_m];                      \
            orc_fr _p;                                                        \
            memcpy(&_p, &_mo->coeff, 32);                                     \
            for (uint32_t _f = 0; _f < _mo->n_factors; _f++) orc_fr_mul(&_p, &_p, &qe[_mo->factors[_f]]); \
            orc_fr_add(&_acc, &_acc, &_p);                                    \
        }                                                                     \
        (dst) = _acc;                                                         \
    } while (0)

    /* l_0, l_last, l_blind at x: l_i(x) = (x^n - 1)/n * omega^i / (x - omega^i) */
    orc_fr l0, llast, lblind = ORC_FR_ZERO, lactive;
    {
        orc_fr num, t;
        orc_fr_sub(&num, &xn, &ORC_FR_ONE);
        orc_fr_mul(&num, &num, &dom.ifft_divisor);
        for (int64_t r = -(int64_t)(bf + 1); r <= 0; r++) {
            orc_fr wi, d, li;
            orc_rotate_omega(&dom, &wi, &ORC_FR_ONE, (int32_t)r);
            orc_fr_sub(&d, &x, &wi);
            orc_fr_inv(&d, &d);
            orc_fr_mul(&li, &num, &wi);
            orc_fr_mul(&li, &li, &d);
            if (r == -(int64_t)(bf + 1)) llast = li;
            else if (r == 0) l0 = li;
            else orc_fr_add(&lblind, &lblind, &li);
        }
        orc_fr_add(&t, &llast, &lblind);
        orc_fr_sub(&lactive, &ORC_FR_ONE, &t);
    }
    /* expected h(x): fold every constraint with y, divide by x^n - 1 */
    orc_fr acc = ORC_FR_ZERO, t, u;
#define FOLD(val)                     \
    do {                              \
        orc_fr_mul(&acc, &acc, &y);   \
        orc_fr_add(&acc, &acc, &(val)); \
    } while (0)
    for (uint32_t g = 0; g < cs->n_gates; g++) {
        EVAL_POLY(t, &cs->gates[g]);
        FOLD(t);
    }
    if (sets > 0) {
        orc_fr_sub(&t, &ORC_FR_ONE, &pz_e[0]);
        orc_fr_mul(&t, &t, &l0);
        FOLD(t);
        orc_fr_sqr(&t, &pz_e[3 * (sets - 1)]);
        orc_fr_sub(&t, &t, &pz_e[3 * (sets - 1)]);
        orc_fr_mul(&t, &t, &llast);
        FOLD(t);
        for (size_t s = 1; s < sets; s++) {
            orc_fr_sub(&t, &pz_e[3 * s], &pz_e[3 * (s - 1) + 2]);
            orc_fr_mul(&t, &t, &l0);
            FOLD(t);
        }
        for (size_t s = 0; s < sets; s++) {
            size_t c0 = s * chunk, c1 = c0 + chunk > P ? P : c0 + chunk;
            orc_fr left = pz_e[3 * s + 1], right = pz_e[3 * s], cd, dp;
            orc_fr_pow_u64(&dp, &ORC_FR_DELTA, (uint64_t)(s * chunk));
            orc_fr_mul(&cd, &beta, &x);
            orc_fr_mul(&cd, &cd, &dp);
            for (size_t c = c0; c < c1; c++) {
                /* column eval at Rotation::cur() */
                const zg_query *col = &cs->perm_columns[c];
                orc_fr ce;
                int found = 0;
                if (col->kind == ZG_ADVICE) {
                    for (uint32_t i = 0; i < cs->n_advice_queries; i++)
                        if (cs->advice_queries[i].column == col->column && cs->advice_queries[i].rotation == 0) { ce = adv_e[i]; found = 1; }
                } else if (col->kind == ZG_FIXED) {
                    for (uint32_t i = 0; i < cs->n_fixed_queries; i++)
                        if (cs->fixed_queries[i].column == col->column && cs->fixed_queries[i].rotation == 0) { ce = fix_e[i]; found = 1; }
                } else {
                    orc_eval_poly(&ce, inst_poly + (size_t)col->column * n, n, &x);
                    found = 1;
                }
                if (!found) { ok = -3; ce = ORC_FR_ZERO; }
                orc_fr_mul(&t, &beta, &sig_e[c]);
                orc_fr_add(&t, &t, &ce);
                orc_fr_add(&t, &t, &gamma);
                orc_fr_mul(&left, &left, &t);
                orc_fr_add(&u, &ce, &cd);
                orc_fr_add(&u, &u, &gamma);
                orc_fr_mul(&right, &right, &u);
                orc_fr_mul(&cd, &cd, &ORC_FR_DELTA);
            }
            orc_fr_sub(&t, &left, &right);
            orc_fr_mul(&t, &t, &lactive);
            FOLD(t);
        }
    }
    for (size_t l = 0; l < NL; l++) {
        const zg_lookup *lk = &cs->lookups[l];
        orc_fr ai = ORC_FR_ZERO, ti = ORC_FR_ZERO, vv;
        for (uint32_t e = 0; e < lk->width; e++) {
            EVAL_POLY(vv, &lk->inputs[e]);
            orc_fr_mul(&ai, &ai, &theta);
            orc_fr_add(&ai, &ai, &vv);
            EVAL_POLY(vv, &lk->tables[e]);
            orc_fr_mul(&ti, &ti, &theta);
            orc_fr_add(&ti, &ti, &vv);
        }
        const orc_fr *e5 = &lk_e[5 * l]; /* z, z_next, a', a'_inv, s' */
        orc_fr_sub(&t, &ORC_FR_ONE, &e5[0]);
        orc_fr_mul(&t, &t, &l0);
        FOLD(t);
        orc_fr_sqr(&t, &e5[0]);
        orc_fr_sub(&t, &t, &e5[0]);
        orc_fr_mul(&t, &t, &llast);
        FOLD(t);
        orc_fr lft, rgt;
        orc_fr_add(&t, &e5[2], &beta);
        orc_fr_add(&u, &e5[4], &gamma);
        orc_fr_mul(&lft, &t, &u);
        orc_fr_mul(&lft, &lft, &e5[1]);
        orc_fr_add(&t, &ai, &beta);
        orc_fr_add(&u, &ti, &gamma);
        orc_fr_mul(&rgt, &t, &u);
        orc_fr_mul(&rgt, &rgt, &e5[0]);
        orc_fr_sub(&t, &lft, &rgt);
        orc_fr_mul(&t, &t, &lactive);
        FOLD(t);
        orc_fr ams;
        orc_fr_sub(&ams, &e5[2], &e5[4]);
        orc_fr_mul(&t, &ams, &l0);
        FOLD(t);
        orc_fr_sub(&t, &e5[2], &e5[3]);
        orc_fr_mul(&t, &t, &ams);
        orc_fr_mul(&t, &t, &lactive);
        FOLD(t);
    }
    orc_fr expected_h;
    orc_fr_sub(&t, &xn, &ORC_FR_ONE);
    orc_fr_inv(&t, &t);
    orc_fr_mul(&expected_h, &acc, &t);

    /* h commitment = sum_i xn^i H_i */
    orc_g1 h_commit;
    orc_g1_identity(&h_commit);
    for (size_t i = qpd; i-- > 0;) {
        orc_g1 tmp;
        orc_g1_mul(&tmp, &h_commit, &xn);
        orc_g1_add_mixed(&h_commit, &tmp, &h_c[i]);
    }
    /* fixed / sigma commitments (the vk holds them; recomputed here from the pk values) */
    orc_g1a *fix_c = (orc_g1a *)malloc((F + 1) * sizeof(orc_g1a));
    orc_g1a *sig_c = (orc_g1a *)malloc((P + 1) * sizeof(orc_g1a));
    for (size_t c = 0; c < F; c++) orc_commit_lagrange(pk->params, &fix_c[c], pk->fixed_values + c * n);
    for (size_t c = 0; c < P; c++) orc_commit_lagrange(pk->params, &sig_c[c], pk->sigma_values + c * n);

    /* queries in the verifier's (= prover's) order */
    size_t max_q = cs->n_advice_queries + cs->n_fixed_queries + P + 3 * sets + 5 * NL + 4;
    vquery *qs = (vquery *)malloc(max_q * sizeof(vquery));
    size_t nq = 0;
    orc_fr x_next, x_inv, x_last;
    orc_rotate_omega(&dom, &x_next, &x, 1);
    orc_rotate_omega(&dom, &x_inv, &x, -1);
    orc_rotate_omega(&dom, &x_last, &x, -(int32_t)(bf + 1));
#define VQ(pt, aff, ev)                                  \
    do {                                                 \
        qs[nq].point = (pt);                             \
        orc_g1_from_affine(&qs[nq].commitment, (aff));   \
        qs[nq].eval = (ev);                              \
        nq++;                                            \
    } while (0)
    for (uint32_t i = 0; i < cs->n_advice_queries; i++) {
        orc_fr pt;
        orc_rotate_omega(&dom, &pt, &x, cs->advice_queries[i].rotation);
        VQ(pt, &adv_c[cs->advice_queries[i].column], adv_e[i]);
    }
    for (size_t s = 0; s < sets; s++) {
        VQ(x, &pz_c[s], pz_e[3 * s]);
        VQ(x_next, &pz_c[s], pz_e[3 * s + 1]);
    }
    for (size_t s = sets; s-- > 0;) {
        if (s + 1 == sets) continue;
        VQ(x_last, &pz_c[s], pz_e[3 * s + 2]);
    }
    for (size_t l = 0; l < NL; l++) {
        VQ(x, &lz_c[l], lk_e[5 * l + 0]);
        VQ(x, &pin_c[l], lk_e[5 * l + 2]);
        VQ(x, &ptab_c[l], lk_e[5 * l + 4]);
        VQ(x_inv, &pin_c[l], lk_e[5 * l + 3]);
        VQ(x_next, &lz_c[l], lk_e[5 * l + 1]);
    }
    for (uint32_t i = 0; i < cs->n_fixed_queries; i++) {
        orc_fr pt;
        orc_rotate_omega(&dom, &pt, &x, cs->fixed_queries[i].rotation);
        VQ(pt, &fix_c[cs->fixed_queries[i].column], fix_e[i]);
    }
    for (size_t c = 0; c < P; c++) VQ(x, &sig_c[c], sig_e[c]);
    qs[nq].point = x; qs[nq].commitment = h_commit; qs[nq].eval = expected_h; nq++;
    VQ(x, &random_c, random_eval);

    tr_squeeze(&tr, &v);
    {
        int *done = (int *)calloc(nq, sizeof(int));
        /* pairing mode: the per-point terms wait for u */
        orc_g1 *pw = (orc_g1 *)malloc(nq * sizeof(orc_g1)), *pr = (orc_g1 *)malloc(nq * sizeof(orc_g1));
        size_t nsets = 0;
        for (size_t first = 0; first < nq && ok == 1; first++) {
            if (done[first]) continue;
            orc_fr z = qs[first].point, eval_batch = ORC_FR_ZERO;
            orc_g1 cb;
            orc_g1_identity(&cb);
            for (size_t j = first; j < nq; j++) {
                if (done[j] || !orc_fr_eq(&qs[j].point, &z)) continue;
                done[j] = 1;
                orc_g1 tmp;
                orc_g1_mul(&tmp, &cb, &v);
                orc_g1_add(&cb, &tmp, &qs[j].commitment);
                orc_fr_mul(&eval_batch, &eval_batch, &v);
                orc_fr_add(&eval_batch, &eval_batch, &qs[j].eval);
            }
            orc_g1a w;
            if (tr_read_point(&tr, &w)) { ok = -1; break; }
            /* e(W, [s - z]_2) == e(C - eval*G, [1]_2)  <=>  (s - z) W == C - eval G */
            orc_g1 lhs, rhs, wj, eg;
            orc_g1_from_affine(&wj, &w);
            scalar_mul_g(&eg, &eval_batch);
            orc_g1_neg(&eg, &eg);
            orc_g1_add(&rhs, &cb, &eg);
            if (use_pairing) {
                orc_g1_mul(&lhs, &wj, &z);      /* z W + C - e G */
                orc_g1_add(&pr[nsets], &lhs, &rhs);
                pw[nsets++] = wj;
                continue;
            }
            orc_fr smz;
            orc_fr_sub(&smz, &pk->params->s, &z);
            orc_g1_mul(&lhs, &wj, &smz);
            if (!orc_g1_eq(&lhs, &rhs)) ok = 0;
        }
        if (use_pairing && ok == 1) {
            orc_fr u;
            tr_squeeze(&tr, &u);
            orc_g1 wsum, rsum, t;
            orc_g1_identity(&wsum);
            orc_g1_identity(&rsum);
            for (size_t i = nsets; i-- > 0;) { /* Horner in u */
                orc_g1_mul(&t, &wsum, &u);
                orc_g1_add(&wsum, &t, &pw[i]);
                orc_g1_mul(&t, &rsum, &u);
                orc_g1_add(&rsum, &t, &pr[i]);
            }
            orc_g1_neg(&rsum, &rsum);
            orc_g1a ps[2];
            orc_g2a qs2[2] = {pk->params->s_g2, pk->params->g2};
            orc_g1_to_affine(&ps[0], &wsum);
            orc_g1_to_affine(&ps[1], &rsum);
            if (!orc_pairing_check(ps, qs2, 2)) ok = 0;
        }
        free(pw); free(pr);
        free(done);
    }
    if (ok == 1 && tr.ipos != proof_len) ok = 0; /* trailing bytes */
    free(qs); free(fix_c); free(sig_c); free(qe); free(inst_poly);
out:
    free(adv_c); free(pin_c); free(ptab_c); free(pz_c); free(lz_c); free(h_c);
    free(adv_e); free(fix_e); free(sig_e); free(pz_e); free(lk_e);
    tr_free(&tr);
    orc_domain_free(&dom);
    return ok;
}
