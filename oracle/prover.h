/*
 * oracle/prover.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of halo2_proofs::plonk::create_proof (v2023_04_20: src/plonk/prover.rs,
 * plonk/evaluation.rs, plonk/lookup/prover.rs, plonk/permutation/prover.rs, plonk/vanishing/prover.rs,
 * poly/kzg/multiopen/gwc/prover.rs) specialised the way zero_g calls it
 * (/root/reference/src/wnn.rs:242-259: KZGCommitmentScheme<Bn256>, ProverGWC, EvmTranscript, one
 * circuit instance, one phase, no user challenges), plus snark-verifier's EvmTranscript
 * (system/halo2/transcript/evm.rs at v2023_04_20).  Those crates are git dependencies whose source
 * is NOT in /root/reference (reference Cargo.toml:21-28,43): restated from the published algorithm.
 *
 * PARITY STATUS: "parity unpinned" w.r.t. real halo2 proof bytes.  The reference's own tests only
 * pin that a proof VERIFIES (src/lib.rs:10-33, test_cli.sh:62-82); this oracle is pinned the same way
 * by orc_verify below (the PLONK/GWC verification equations, with the pairing replaced by scalar
 * arithmetic on the known toxic scalar s of the test SRS) and by keccak256("").
 *
 * The circuit description types are the ABI's (include/zg_halo2.h): the boundary is shared, the
 * arithmetic below is independent of the product (4x64-bit limbs vs the product's 8x32).
 */
#ifndef ZG_ORACLE_PROVER_H
#define ZG_ORACLE_PROVER_H

#include "../include/zg_halo2.h"
#include "poly.h"

#ifdef __cplusplus
extern "C" {
#endif

void orc_keccak256(const uint8_t *data, size_t len, uint8_t out[32]);

/* Blinding scalar = pure function of (key, tag, index): ChaCha20 (RFC 7539) keystream under the caller's 32-byte
 * key, nonce = (tag, index), rejection sampling of a canonical value < r, returned in Montgomery form
 * (DESIGN.md "randomness").  Upstream draws from OsRng (/root/reference/src/wnn.rs:256); a caller that wants
 * the same hiding passes 32 fresh random bytes, tests pass fixed keys. */
enum {
    ORC_TAG_ADVICE_BLIND = 1,     /* index = column * (bf + 1) + j                    */
    ORC_TAG_PERMUTED_INPUT = 2,   /* index = lookup * (bf + 1) + j                    */
    ORC_TAG_PERMUTED_TABLE = 3,
    ORC_TAG_PERM_Z = 4,           /* index = set * bf + j                             */
    ORC_TAG_LOOKUP_Z = 5,         /* index = lookup * bf + j                          */
    ORC_TAG_RANDOM_POLY = 6       /* index = coefficient                              */
};
void orc_chacha20_block(const uint8_t key[32], uint32_t counter, const uint32_t nonce[3], uint32_t out[16]);
void orc_rand_fr(orc_fr *out, const uint8_t key[32], uint32_t tag, uint64_t index);

/* Proving key material, as keygen_pk leaves it (all host arrays, Lagrange values). */
typedef struct {
    const zg_circuit *cs;
    const orc_fr *fixed_values;   /* [n_fixed][n]        */
    const orc_fr *sigma_values;   /* [n_perm_columns][n] */
    const orc_params *params;
    orc_fr vk_repr;
    void *derived;                /* orc_pk_derive's; NULL = computed inside every create_proof */
} orc_pk;
/* The rest of what keygen_pk leaves in the ProvingKey (fixed / permutation polys and cosets, l0, l_last, l_active_row):
 * computed once per key.  create_proof gives the same bytes with or without it. */
void orc_pk_derive(orc_pk *pk);
void orc_pk_release(orc_pk *pk);

/* Intermediates kept for piecewise parity checks against the GPU path. */
typedef struct {
    orc_fr *h_ext;            /* [extended_n] h on the coset, after division by (X^n - 1) */
    orc_fr *perm_z;           /* [sets][n]    */
    orc_fr *lookup_z;         /* [lookups][n] */
    orc_fr *permuted_input;   /* [lookups][n] */
    orc_fr *permuted_table;   /* [lookups][n] */
    orc_fr *h_pieces;         /* [qpd * n]    */
    orc_fr theta, beta, gamma, y, x, v;
    uint32_t n_sets;
} orc_trace;
void orc_trace_free(orc_trace *t);

/* Returns 0, or ZG_ERR_CONSTRAINT when a lookup input is missing from its table
 * (plonk::Error::ConstraintSystemFailure), or ZG_ERR_INVALID_ARG.  advice: [n_advice][n], not
 * modified.  trace may be NULL. */
int orc_create_proof(const orc_pk *pk, const orc_fr *advice, const orc_fr *instance, size_t instance_len,
                     const uint8_t seed[32], uint8_t *proof, size_t cap, size_t *proof_len, orc_trace *trace);

/* plonk::verify_proof restated (VerifierGWC + SingleStrategy), with the final pairing
 * e(W, [s]_2 - z[1]_2) = e(C - v[1]_1, [1]_2) checked in G1 through the known toxic scalar:
 * (s - z) * W == C - v * G.  Only valid for SRS built by orc_params_new (tests).  Returns 1 if the
 * proof verifies, 0 if it does not, negative on malformed input. */
int orc_verify_proof(const orc_pk *pk, const orc_fr *instance, size_t instance_len, const uint8_t *proof,
                     size_t proof_len);
/* Same checks, with the opening equation decided by the BN254 pairing from g2 / s_g2 only (the public
 * verification equation of Wnn::verify_proof, /root/reference/src/wnn.rs:265-280). */
int orc_verify_proof_pairing(const orc_pk *pk, const orc_fr *instance, size_t instance_len, const uint8_t *proof,
                             size_t proof_len);

/* Milliseconds the calling thread's last orc_create_proof spent per phase, in the slots of zg_prover_phase_ms (0 advice,
 * 1 permuted lookups, 2 products, 3 coefficient forms + evaluate_h + h, 4 evaluations, 5 openings, 6 total). */
void orc_last_phase_ms(double *out, size_t cap);

/* Stand-alone pieces, used by the piecewise GPU parity tests. */
void orc_grand_product(orc_fr *z, const orc_fr *num, const orc_fr *den, const orc_fr *z0, size_t n);
size_t orc_proof_size(const zg_circuit *cs);

#ifdef __cplusplus
}
#endif
#endif
