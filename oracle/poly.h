/*
 * oracle/poly.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of `halo2_proofs::arithmetic::{best_multiexp, best_fft, eval_polynomial,
 * kate_division}`, `halo2_proofs::poly::EvaluationDomain` and `poly::kzg::commitment::ParamsKZG`
 * at tag v2023_04_20 (git dependency, reference Cargo.toml:21-25; source NOT in /root/reference --
 * restated from the published algorithm).  Reference call sites that reach them:
 *   create_proof        /root/reference/src/wnn.rs:242-259
 *   ParamsKZG::new(k)   /root/reference/benches/bench.rs:19, src/main.rs:232
 *   keygen_vk/keygen_pk /root/reference/src/wnn.rs:226-228
 * PARITY STATUS: parity unpinned (no golden vectors upstream of us); pinned here by algebraic
 * identities in tests/ (NTT vs O(n^2) DFT, MSM vs naive sum, iNTT(NTT(a)) = a, ...).
 */
#ifndef ZG_ORACLE_POLY_H
#define ZG_ORACLE_POLY_H

#include "bn254.h"
#include "pairing.h"

#ifdef __cplusplus
extern "C" {
#endif

/* OpenMP thread count used by the parallel loops below (CPU-baseline timing). */
void orc_set_threads(int n);

/* ---- arithmetic.rs ---- */
/* Naive sum_i s_i * P_i by double-and-add: the definition MSM implementations are checked against. */
void orc_msm_naive(orc_g1 *out, const orc_fr *scalars, const orc_g1a *bases, size_t n);
/* halo2 `multiexp_serial`: c = 3 if n < 32 else ceil(ln n); segments = 256/c + 1; buckets 2^c - 1;
 * summation by parts.  `best_multiexp` chunks this over rayon threads and sums the chunk results;
 * the group element is the same. */
void orc_msm(orc_g1 *out, const orc_fr *scalars, const orc_g1a *bases, size_t n);
/* Same algorithm, OpenMP over base chunks like best_multiexp does over rayon threads. */
void orc_msm_mt(orc_g1 *out, const orc_fr *scalars, const orc_g1a *bases, size_t n, int threads);

/* halo2 `best_fft` (serial branch): bit-reversal + radix-2 DIT, in place, natural order in/out. */
void orc_fft(orc_fr *a, const orc_fr *omega, uint32_t log_n);
/* O(n^2) definition out[k] = sum_j a[j] omega^(jk), for cross-checks. */
void orc_dft_naive(orc_fr *out, const orc_fr *a, const orc_fr *omega, uint32_t log_n);
/* `eval_polynomial`: Horner. */
void orc_eval_poly(orc_fr *out, const orc_fr *coeffs, size_t n, const orc_fr *x);
/* `kate_division`: quotient of a(X) by (X - b); q has n-1 entries. */
void orc_kate_division(orc_fr *q, const orc_fr *a, size_t n, const orc_fr *b);

/* ---- poly/domain.rs ---- */
typedef struct {
    uint32_t k, extended_k;
    uint64_t n, extended_n;
    uint32_t quotient_poly_degree;       /* j - 1 */
    orc_fr omega, omega_inv, extended_omega, extended_omega_inv;
    orc_fr g_coset, g_coset_inv;         /* ZETA, ZETA^2 */
    orc_fr ifft_divisor, extended_ifft_divisor;
    orc_fr barycentric_weight;
    orc_fr *t_evaluations;               /* 2^(extended_k-k) inverses of (zeta*ext_omega^i)^n - 1 */
    size_t t_len;
} orc_domain;

void orc_domain_new(orc_domain *d, uint32_t j /* cs.degree() */, uint32_t k);
void orc_domain_free(orc_domain *d);
void orc_lagrange_to_coeff(const orc_domain *d, orc_fr *a /* n, in place */);
void orc_coeff_to_lagrange(const orc_domain *d, orc_fr *a /* n, in place (not in halo2; test aid) */);
/* out has extended_n entries */
void orc_coeff_to_extended(const orc_domain *d, orc_fr *out, const orc_fr *coeffs /* n */);
/* a has extended_n entries and is clobbered; out gets n * quotient_poly_degree entries */
void orc_extended_to_coeff(const orc_domain *d, orc_fr *out, orc_fr *a);
void orc_divide_by_vanishing(const orc_domain *d, orc_fr *a /* extended_n */);
void orc_rotate_omega(const orc_domain *d, orc_fr *out, const orc_fr *x, int32_t rotation);

/* ---- poly/kzg/commitment.rs: ParamsKZG::new(k) with a caller-supplied toxic scalar ---- */
typedef struct {
    uint32_t k;
    uint64_t n;
    orc_g1a *g;          /* s^i * G                */
    orc_g1a *g_lagrange; /* L_i(s) * G             */
    orc_fr s;            /* kept for the test-only pairing-free verifier */
    orc_g2a g2, s_g2;    /* [1]_2, [s]_2: all the pairing verifier may use of s */
} orc_params;

void orc_params_new(orc_params *p, uint32_t k, const orc_fr *s);
void orc_params_free(orc_params *p);
void orc_commit(const orc_params *p, orc_g1a *out, const orc_fr *coeffs);          /* g          */
void orc_commit_lagrange(const orc_params *p, orc_g1a *out, const orc_fr *evals);  /* g_lagrange */

/* Fixed-base helper: out[i] = scalars[i] * G, affine (used to build the SRS). */
void orc_fixed_base_mul(orc_g1a *out, const orc_fr *scalars, size_t n);

#ifdef __cplusplus
}
#endif
#endif
