// Device-side circuit image and the launch helpers of poly.hip (shared with prover.hip).
#pragma once

#include "common.h"

namespace zg {

struct alignas(16) DMono {
    Fe coeff;
    uint32_t n_factors;
    uint32_t coeff_is_one;
    uint32_t factors[ZG_MAX_FACTORS];
    uint32_t pad[2];
};
static_assert(sizeof(DMono) == 80, "DMono layout");

struct DLookup {
    uint32_t width;
    zg_poly inputs[ZG_MAX_LOOKUP_WIDTH];
    zg_poly tables[ZG_MAX_LOOKUP_WIDTH];
};

// circuit arrays in HBM (read through the scalar cache: every index is wave-uniform)
struct DevCircuit {
    const zg_query* queries;
    const DMono* monos;
    const zg_poly* gates;
    const DLookup* lookups;
    const zg_query* perm_cols;
    uint32_t n_gates, n_lookups, n_perm, chunk, n_sets;
};

// one family of column arrays [column][size]
struct Cols {
    const Fe* fixed;
    const Fe* advice;
    const Fe* instance;
    uint32_t log_size;
    int32_t rot_scale;
};

struct EvalHArgs {
    DevCircuit c;
    Cols cols;  // extended cosets
    const Fe* sigma_cos;
    const Fe* pz_cos;
    const Fe* lz_cos;
    const Fe* pin_cos;   // lookup l's permuted input / table cosets: pin_cos + l * perm_stride, ptab_cos + l * perm_stride
    const Fe* ptab_cos;
    size_t perm_stride;  // (2 * en when the two live interleaved in one batch, as the prover keeps them)
    const Fe* l0;
    const Fe* llast;
    const Fe* lactive;
    const Fe* ext_tw;  // extended_omega^i
    const Fe* t_eval;  // ((zeta*ext_omega^i)^n - 1)^-1, period t_len (power of two)
    uint32_t t_mask;
    int32_t last_rot;  // -(blinding_factors + 1)
    Fe y, beta, gamma, theta, delta_start, delta;
    Fe* h;
    // nine-limb evaluation (hat = true): every coset slab, l-polynomial, ext_tw and t_eval entry and every
    // constant above is in the 2^261 Montgomery form (x * 2^261 mod r, packed), and so are the monomial
    // coefficients in monos_hat; h comes out in that form too (extended_to_coeff_dev(unhat) undoes it)
    bool hat;
    const DMono* monos_hat;
    // gates with a cell common to all their monomials (zero_g's are selector * (...)) are given factored:
    // gates_hat[g] ranges over monos_hat entries with that cell removed, gate_common[g] is its query index
    // (0xffffffff: not factored, gates_hat[g] is the plain polynomial)
    const zg_poly* gates_hat;
    const uint32_t* gate_common;
    // the factor may be a polynomial in that cell (halo2's merged selectors: q * prod (u - q)):
    // gate g = U(cell) * gates_hat[g], U(x) = sum_{k=1..count} uni_coef[first + k - 1] x^k; count 0: U(x) = x
    const zg_poly* gate_uni;
    const Fe* uni_coef;
    // ... and when that cell is a fixed one, U(cell) is tabulated over the extended domain by poly_gate_factor:
    // gate_slab[g] = index of the gate's slab (en entries) in gate_slabs, 0xffffffff: evaluate U in the kernel
    const uint32_t* gate_slab;
    const Fe* gate_slabs;
};

// blinding scalar = f(seed, tag, index); identical to the oracle's definition
enum {
    TAG_ADVICE_BLIND = 1,
    TAG_PERMUTED_INPUT = 2,
    TAG_PERMUTED_TABLE = 3,
    TAG_PERM_Z = 4,
    TAG_LOOKUP_Z = 5,
    TAG_RANDOM_POLY = 6
};

// ---- launch helpers (all asynchronous on ctx->stream) ----
int poly_blind_rows(zg_ctx* ctx, Fe* base, size_t col_stride, uint32_t ncols, uint32_t row0, uint32_t nrows,
                    uint64_t seed, uint32_t tag);
int poly_random(zg_ctx* ctx, Fe* out, uint32_t n, uint64_t seed, uint32_t tag, Fe* out2 = nullptr);
int poly_random_and_blind(zg_ctx* ctx, Fe* out, Fe* out2, uint32_t n, uint64_t seed, uint32_t tag, Fe* base, size_t col_stride,
                          uint32_t ncols, uint32_t row0, uint32_t nrows, uint32_t blind_tag);  // out2: a second copy
int poly_lookup_compress(zg_ctx* ctx, const DevCircuit& c, const Cols& cols, const Fe& theta, Fe* cin, Fe* ctab,
                         uint32_t n, Fe* raw_in = nullptr, Fe* raw_tab = nullptr, uint32_t usable = 0);
int poly_blind_rows2(zg_ctx* ctx, Fe* base, size_t col_stride, uint32_t ncols0, uint32_t tag0, uint32_t ncols1, uint32_t tag1,
                     uint32_t row0, uint32_t nrows, uint64_t seed);
int poly_permuted_finish(zg_ctx* ctx, const Fe* raw_in, const Fe* raw_tab, Fe* perm, uint32_t n, uint32_t usable,
                         uint32_t nblind, uint32_t n_lookups, uint64_t seed, uint32_t tag_in, uint32_t tag_tab);
int poly_to_raw(zg_ctx* ctx, const Fe* in, Fe* out, size_t count);
int poly_from_raw(zg_ctx* ctx, const Fe* in, Fe* out, size_t count);
int poly_from_raw_rows(zg_ctx* ctx, const Fe* src, size_t src_stride, Fe* dst, size_t dst_stride, uint32_t rows,
                       uint32_t len);
int poly_lookup_terms(zg_ctx* ctx, const Fe* cin, const Fe* ctab, const Fe* pin, const Fe* ptab, size_t perm_stride,
                      const Fe& beta, const Fe& gamma, Fe* num, Fe* den, uint32_t n, uint32_t n_lookups);
int poly_perm_terms(zg_ctx* ctx, const DevCircuit& c, const Cols& cols, const Fe* sigma_val, const Fe* omega_tw,
                    const Fe& beta, const Fe& gamma, Fe* num, Fe* den, uint32_t n);
// z[b][0] = z0[b] (device array, or all ones if null), z[b][i+1] = z[b][i]*num[b][i]/den[b][i]; the first
// `chain` products are chained: product b starts from product b-1's value at row `last`.
size_t poly_grand_product_tmp_elems(uint32_t n, uint32_t batch);
int poly_grand_product(zg_ctx* ctx, const Fe* num, const Fe* den, const Fe* d_z0, Fe* z, Fe* tmp, uint32_t n,
                       uint32_t batch, uint32_t chain, uint32_t last);
int poly_evaluate_h(zg_ctx* ctx, const EvalHArgs& a, uint32_t en);
// out[i] = U(col[(i + rot_off) mod en]),  U(x) = sum_{k=1..count} coef[k-1] x^k; everything in the 2^261 form
int poly_gate_factor(zg_ctx* ctx, const Fe* col, uint32_t rot_off, uint32_t en, const Fe* coef, uint32_t count, Fe* out);
int poly_powers(zg_ctx* ctx, const Fe* points_host, uint32_t npoints, uint32_t n, Fe* d_pow);
int poly_dot(zg_ctx* ctx, const Fe* polys, size_t stride, uint32_t n, const uint32_t* d_poly_idx,
             const uint32_t* d_point_idx, const Fe* d_pow, uint32_t count, Fe* d_out);
// out[i] = sum_j horner in `v` over the polys listed (first listed = highest power), then out[0] -= sub
int poly_horner_combine(zg_ctx* ctx, const Fe* polys, size_t stride, const uint32_t* d_list, uint32_t count,
                        const Fe& v, const Fe& sub, Fe* out, uint32_t n);
constexpr uint32_t HC_MAX_SETS = 8;
constexpr uint32_t FESET_MAX = 8;
struct FeSet {  // a few field elements passed by value in kernel arguments (opening points)
    Fe v[FESET_MAX];
};
int poly_horner_combine_sets(zg_ctx* ctx, const Fe* polys, size_t stride, const uint32_t* d_lists, uint32_t list_stride,
                             const uint32_t* counts, const Fe* subs, uint32_t nsets, const Fe& v, Fe* out, size_t out_stride,
                             uint32_t n);
size_t poly_kate_tmp_elems(uint32_t n, uint32_t batch);
int poly_kate_division(zg_ctx* ctx, const Fe* a, size_t a_stride, const Fe* zs_host, Fe* q, size_t q_stride, Fe* tmp,
                       uint32_t n, uint32_t batch);
int poly_l_cosets_init(zg_ctx* ctx, Fe* l0, Fe* llast, Fe* lblind, uint32_t n, uint32_t bf);
int poly_lactive(zg_ctx* ctx, Fe* lactive, const Fe* llast, const Fe* lblind, uint32_t en, bool hat);
int poly_scale(zg_ctx* ctx, const Fe* in, Fe* out, size_t count, const Fe& factor);  // out[i] = in[i] * factor

// from sort.hip: lookup::prover::permute_expression_pair on the device
int poly_sort_pad(zg_ctx* ctx, Fe* keys, uint32_t n, uint32_t usable, uint32_t batch);
int poly_sort_keys(zg_ctx* ctx, Fe* keys, uint32_t n, uint32_t batch);
int poly_permute_pairs(zg_ctx* ctx, const Fe* a, const Fe* t, Fe* sprime, uint32_t n, uint32_t usable, uint32_t batch,
                       uint32_t* scratch_u32, Fe* scratch_fe, uint32_t* d_err);

// from ntt.hip
int ntt_batch_dev(zg_ctx* ctx, Fe* d_a, size_t stride, size_t batch, uint32_t log_n, const Fe& omega,
                  const Fe* divisor);
int ntt_batch_to_dev(zg_ctx* ctx, const Fe* d_in, Fe* d_out, size_t stride, size_t batch, uint32_t log_n, const Fe& omega,
                     const Fe* divisor);
int coeff_to_extended_dev(zg_ctx* ctx, const Fe* d_in, size_t in_stride, Fe* d_out, size_t out_stride,
                          size_t batch, uint32_t k, uint32_t ext_k, bool hat);
int extended_to_coeff_dev(zg_ctx* ctx, Fe* d_evals, uint32_t k, uint32_t ext_k, size_t out_len, Fe* d_out, bool unhat);
int coeff_to_coset_dev(zg_ctx* ctx, const Fe* d_in, size_t in_stride, uint32_t in_len, Fe* d_out, size_t out_stride,
                       size_t batch, uint32_t ext_k, bool hat, int zeta_pow);
int coset_to_coeff_dev(zg_ctx* ctx, Fe* d_evals, uint32_t ext_k, size_t out_len, Fe* d_out, bool unhat, int zeta_pow);
// split extended domain (prover.hip): the pieces of the interpolation between its two cosets
int poly_fold(zg_ctx* ctx, const Fe* a, uint32_t len, uint32_t parts, const Fe& e, Fe* out);  // out[r] = sum_q a[r + q*len] e^q
int poly_diff_scale(zg_ctx* ctx, const Fe* u, const Fe& cu, const Fe* v, const Fe& scale, Fe* out, uint32_t len);  // (u*cu - v)*scale
int poly_split_combine(zg_ctx* ctx, Fe* h, const Fe* b, uint32_t len, const Fe& c1, uint32_t hi_at);  // h[j] -= c1 b[j]; h[hi_at + j] = b[j]
// from msm.hip
int msm_batch_dev(zg_ctx* ctx, const zg_bases* bases, const Fe* d_scalars, size_t stride, size_t batch, size_t n,
                  XYZZ* d_out);
int msm_batch2_dev(zg_ctx* ctx, const zg_bases* bases, const zg_bases* bases_b, size_t split, const Fe* d_scalars,
                   size_t stride, size_t batch, size_t n, XYZZ* d_out);
int msm_batch3_dev(zg_ctx* ctx, const zg_bases* bases, const zg_bases* bases_b, size_t split, const Fe* d_scalars,
                   size_t stride, size_t batch, size_t n, XYZZ* d_out, uint64_t run_mask);
int bases_enable_runs(zg_ctx* ctx, zg_bases* b);  // running-sum table for the run form (idempotent)
int bases_register_dev(zg_ctx* ctx, const Affine* d_bases, size_t n, uint32_t window_bits, zg_bases** out);
void xyzz_batch_normalise(const XYZZ* in, size_t count, zg_g1* out);

}  // namespace zg
