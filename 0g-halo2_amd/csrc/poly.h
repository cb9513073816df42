// Device-side circuit image and the launch helpers of poly.hip (shared with prover.hip).
#pragma once

#include "common.h"

namespace zg {

struct alignas(16) DMono {
    Fe coeff;
    uint32_t n_factors;
    uint32_t coeff_is_one;  // 1: coefficient +1, 2: coefficient -1 (the nine-limb evaluation subtracts instead of multiplying)
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

// one family of column arrays [column][size]; in a lock-step batch proof b's advice / instance columns start
// adv_bs / inst_bs elements after proof b-1's (the fixed columns belong to the proving key: one copy)
struct Cols {
    const Fe* fixed;
    const Fe* advice;
    const Fe* instance;
    uint32_t log_size;
    int32_t rot_scale;
    size_t adv_bs, inst_bs;
};

// Per-proof scalars of a lock-step batch: one entry per proof in HBM, rewritten by the host at every transcript step
// (kernels read their proof's entry through the scalar cache: the proof index is workgroup-uniform).
constexpr uint32_t PC_MAX_POINTS = 16;
constexpr uint32_t EH_MAX_YPOW = 48;  // evaluate_h's grouped form: powers of y, one per term after the gates (+ 1)
struct alignas(16) ProofConst {
    uint32_t key[8];            // blinding key (ChaCha20), rand_fr
    Fe theta, beta, gamma;      // library (2^256 Montgomery) form: lookup compression, product terms
    Fe eh_y, eh_beta, eh_gamma, eh_theta;  // the form evaluate_h computes in (x 2^5 for the nine-limb kernel)
    Fe eh_delta_start[2];       // beta * zeta^zpow for zpow = 1, 2, same form
    Fe eh_ypow[EH_MAX_YPOW];    // y^j, same form (the weights of the permutation / lookup terms, EvalHArgs::n_terms)
    Fe xn;                      // x^n (vanishing::evaluate's Horner variable)
    Fe v;                       // GWC's v
    Fe points[PC_MAX_POINTS];   // opening points x * omega^rotation, by slot
    Fe subs[PC_MAX_POINTS];     // eval_batch of GWC point set s
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
    // lock-step batch: workgroup row blockIdx.y = proof b; its challenges are pc[b] (eh_* fields; delta_start is
    // eh_delta_start[zpow - 1]), its cosets start b * cos_bs after sigma_cos' siblings, its h at h + b * h_bs
    const ProofConst* pc;
    uint32_t zpow;
    size_t cos_bs, h_bs;
    Fe delta;
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
    // grouped form of the nine-limb evaluation: the n_terms permutation / lookup terms that follow the gates are
    // weighted by pc->eh_ypow and summed per l-polynomial (0: the plain Horner fold in y)
    uint32_t n_terms;
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
// Lock-step batches: `nb` proofs, proof b's scalars in pc[b], its arrays `*_bs` elements after proof b-1's.
// The two draws a proof starts with: the vanishing argument's random polynomial (n values, to out and out2) and
// the blinding rows [row0, row0 + nrows) of the ncols advice columns.
int poly_random_and_blind(zg_ctx* ctx, const ProofConst* pc, uint32_t nb, Fe* out, size_t out_bs, Fe* out2, size_t out2_bs,
                          uint32_t n, uint32_t tag, Fe* base, size_t base_bs, size_t col_stride, uint32_t ncols,
                          uint32_t row0, uint32_t nrows, uint32_t blind_tag);
// columns [0, ncols0) draw from tag0 (index c * nrows + j), the ncols1 columns after them from tag1 (index restarts)
int poly_blind_rows2(zg_ctx* ctx, const ProofConst* pc, uint32_t nb, Fe* base, size_t base_bs, size_t col_stride,
                     uint32_t ncols0, uint32_t tag0, uint32_t ncols1, uint32_t tag1, uint32_t row0, uint32_t nrows);
// lookup l of proof b: compressed columns (and the sort keys raw_in / raw_tab) at index b * n_lookups + l
int poly_lookup_compress(zg_ctx* ctx, const DevCircuit& c, const Cols& cols, const ProofConst* pc, uint32_t nb, Fe* cin,
                         Fe* ctab, uint32_t n, Fe* raw_in = nullptr, Fe* raw_tab = nullptr, uint32_t usable = 0);
// perm + b * perm_bs + 2l * n = a'_l, + n = s'_l
int poly_permuted_finish(zg_ctx* ctx, const ProofConst* pc, uint32_t nb, const Fe* raw_in, const Fe* raw_tab, Fe* perm,
                         size_t perm_bs, uint32_t n, uint32_t usable, uint32_t nblind, uint32_t n_lookups, uint32_t tag_in,
                         uint32_t tag_tab);
int poly_to_raw(zg_ctx* ctx, const Fe* in, Fe* out, size_t count);
int poly_from_raw(zg_ctx* ctx, const Fe* in, Fe* out, size_t count);
int poly_from_raw_rows(zg_ctx* ctx, const Fe* src, size_t src_stride, Fe* dst, size_t dst_stride, uint32_t rows,
                       uint32_t len);
// product q = b * per + j of the batch (per products per proof: `sets` permutation sets, then the lookups):
// num / den at q * n
int poly_lookup_terms(zg_ctx* ctx, const ProofConst* pc, uint32_t nb, const Fe* cin, const Fe* ctab, const Fe* pin,
                      const Fe* ptab, size_t perm_stride, size_t perm_bs, Fe* num, Fe* den, uint32_t per, uint32_t first,
                      uint32_t n, uint32_t n_lookups);
int poly_perm_terms(zg_ctx* ctx, const DevCircuit& c, const Cols& cols, const ProofConst* pc, uint32_t nb,
                    const Fe* sigma_val, const Fe* omega_tw, Fe* num, Fe* den, uint32_t per, uint32_t n);
// z[q][0] = z0[q] (device array, or all ones if null), z[q][i+1] = z[q][i]*num[q][i]/den[q][i] for the `batch`
// products q = g * per + j (per = 0: one group); within a group the first `chain` products are chained: product j
// starts from product j-1's value at row `last`.  num / den / tmp are flat ([q][n]); z of product q goes to
// z + g * z_outer + j * n (z_outer = 0: flat).
size_t poly_grand_product_tmp_elems(uint32_t n, uint32_t batch);
// half: 0 = the whole sequence; 1 = only the launches up to the totals, 2 = only what follows them (the latency form's
// host inversion of the totals -- a stream synchronisation -- and the apply launch): the two halves of one call, same arguments.
int poly_grand_product(zg_ctx* ctx, const Fe* num, const Fe* den, const Fe* d_z0, Fe* z, Fe* tmp, uint32_t n,
                       uint32_t batch, uint32_t chain, uint32_t last, uint32_t per = 0, size_t z_outer = 0, int half = 0);
// n_columns = advice + instance + fixed columns of the circuit, unit_share = the rows of EvaluationDomain's extended domain
// this launch stands for (both only shape the profile's byte charges)
int poly_evaluate_h(zg_ctx* ctx, const EvalHArgs& a, uint32_t en, uint32_t nb, uint32_t n_columns, double unit_share);
// out[i] = U(col[(i + rot_off) mod en]),  U(x) = sum_{k=1..count} coef[k-1] x^k; everything in the 2^261 form
int poly_gate_factor(zg_ctx* ctx, const Fe* col, uint32_t rot_off, uint32_t en, const Fe* coef, uint32_t count, Fe* out);
// Coefficient-form polynomials a kernel may be asked for by index: indices below nsh name the proving key's
// (one copy: sh + ix * n), the others proof b's (pp + b * pp_bs + (ix - nsh) * n).
struct PolySet {
    const Fe* sh;
    const Fe* pp;
    uint32_t nsh;
    size_t n;      // elements per polynomial
    size_t pp_bs;
};
// pw + b * pw_bs + s * n <- powers 0 .. n-1 of pc[b].points[s], s < npoints
int poly_powers(zg_ctx* ctx, const ProofConst* pc, uint32_t nb, uint32_t npoints, uint32_t n, Fe* d_pow, size_t pw_bs);
// out[b * out_bs + j] = <poly poly_idx[j] of proof b, powers row point_idx[j] of proof b>
int poly_dot(zg_ctx* ctx, const PolySet& polys, uint32_t nb, uint32_t n, const uint32_t* d_poly_idx,
             const uint32_t* d_point_idx, const Fe* d_pow, size_t pw_bs, uint32_t count, Fe* d_out, size_t out_bs,
             uint32_t distinct_polys = 0, uint32_t distinct_points = 0);  // (distinct counts: the profile's byte charge only)
// out[b][i] = Horner in pc[b].xn over the polys listed (first listed = highest power)
int poly_horner_combine_xn(zg_ctx* ctx, const PolySet& polys, const ProofConst* pc, uint32_t nb, const uint32_t* d_list,
                           uint32_t count, Fe* out, size_t out_bs, uint32_t n);
constexpr uint32_t HC_MAX_SETS = PC_MAX_POINTS;
constexpr uint32_t FESET_MAX = 8;
struct FeSet {  // a few field elements passed by value in kernel arguments
    Fe v[FESET_MAX];
};
// GWC: set s of proof b = Horner in pc[b].v over the polys of list s, minus pc[b].subs[s] at X^0;
// out + b * out_bs + s * out_stride
int poly_horner_combine_sets(zg_ctx* ctx, const PolySet& polys, const ProofConst* pc, uint32_t nb, const uint32_t* d_lists,
                             uint32_t list_stride, const uint32_t* counts, uint32_t nsets, Fe* out, size_t out_stride,
                             size_t out_bs, uint32_t n);
// kate_division of `nsets` polynomials per proof: a + b * a_bs + s * a_stride divided by (X - pc[b].points[slot[s]])
size_t poly_kate_tmp_elems(uint32_t n, uint32_t batch);
int poly_kate_division(zg_ctx* ctx, const ProofConst* pc, uint32_t nb, const uint32_t* slots, uint32_t nsets, const Fe* a,
                       size_t a_stride, size_t a_bs, Fe* q, size_t q_stride, size_t q_bs, Fe* tmp, uint32_t n);
int poly_l_cosets_init(zg_ctx* ctx, Fe* l0, Fe* llast, Fe* lblind, uint32_t n, uint32_t bf);
int poly_lactive(zg_ctx* ctx, Fe* lactive, const Fe* llast, const Fe* lblind, uint32_t en, bool hat);
int poly_scale(zg_ctx* ctx, const Fe* in, Fe* out, size_t count, const Fe& factor);  // out[i] = in[i] * factor

// from sort.hip: lookup::prover::permute_expression_pair on the device
int poly_sort_pad(zg_ctx* ctx, Fe* keys, uint32_t n, uint32_t usable, uint32_t batch);
int poly_sort_keys(zg_ctx* ctx, Fe* keys, uint32_t n, uint32_t batch);
int poly_permute_pairs(zg_ctx* ctx, const Fe* a, const Fe* t, Fe* sprime, uint32_t n, uint32_t usable, uint32_t batch,
                       uint32_t* scratch_u32, Fe* scratch_fe, uint32_t* d_err);

// from ntt.hip
// Batches laid out in groups (the same columns of several proofs): array v of a batch sits at
// base + (v / per) * outer + (v % per) * stride, on the input and on the output side; per = 0: flat (v * stride).
struct Grouping {
    uint32_t per = 0;
    size_t in_outer = 0, out_outer = 0;
};
int ntt_batch_dev(zg_ctx* ctx, Fe* d_a, size_t stride, size_t batch, uint32_t log_n, const Fe& omega,
                  const Fe* divisor);
int ntt_batch_to_dev(zg_ctx* ctx, const Fe* d_in, Fe* d_out, size_t stride, size_t batch, uint32_t log_n, const Fe& omega,
                     const Fe* divisor, const Grouping* grp = nullptr);
int coeff_to_extended_dev(zg_ctx* ctx, const Fe* d_in, size_t in_stride, Fe* d_out, size_t out_stride,
                          size_t batch, uint32_t k, uint32_t ext_k, bool hat);
int extended_to_coeff_dev(zg_ctx* ctx, Fe* d_evals, uint32_t k, uint32_t ext_k, size_t out_len, Fe* d_out, bool unhat);
int coeff_to_coset_dev(zg_ctx* ctx, const Fe* d_in, size_t in_stride, uint32_t in_len, Fe* d_out, size_t out_stride,
                       size_t batch, uint32_t ext_k, bool hat, int zeta_pow, const Grouping* grp = nullptr);
int coset_to_coeff_dev(zg_ctx* ctx, Fe* d_evals, uint32_t ext_k, size_t out_len, Fe* d_out, bool unhat, int zeta_pow,
                       size_t batch = 1, size_t in_stride = 0, size_t out_stride = 0);
// split extended domain (prover.hip): the pieces of the interpolation between its two cosets
// (each for nb proofs: proof b's arrays `*_bs` elements after proof b-1's)
int poly_fold(zg_ctx* ctx, uint32_t nb, const Fe* a, size_t a_bs, uint32_t len, uint32_t parts, const Fe& e, Fe* out,
              size_t out_bs);  // out[r] = sum_q a[r + q*len] e^q
int poly_diff_scale(zg_ctx* ctx, uint32_t nb, const Fe* u, size_t u_bs, const Fe& cu, const Fe* v, size_t v_bs, const Fe& scale,
                    Fe* out, size_t out_bs, uint32_t len);  // (u*cu - v)*scale
int poly_split_combine(zg_ctx* ctx, uint32_t nb, Fe* h, size_t h_bs, const Fe* b, size_t b_bs, uint32_t len, const Fe& c1,
                       uint32_t hi_at);  // h[j] -= c1 b[j]; h[hi_at + j] = b[j]
// from msm.hip
int msm_batch_dev(zg_ctx* ctx, const zg_bases* bases, const Fe* d_scalars, size_t stride, size_t batch, size_t n,
                  XYZZ* d_out);
int msm_batch2_dev(zg_ctx* ctx, const zg_bases* bases, const zg_bases* bases_b, size_t split, const Fe* d_scalars,
                   size_t stride, size_t batch, size_t n, XYZZ* d_out);
int msm_batch3_dev(zg_ctx* ctx, const zg_bases* bases, const zg_bases* bases_b, size_t split, const Fe* d_scalars,
                   size_t stride, size_t batch, size_t n, XYZZ* d_out, uint64_t run_mask);
// groups of `per` vectors: vector v at d_scalars + (v / per) * outer + (v % per) * stride; split / run_mask by v % per
int msm_batch4_dev(zg_ctx* ctx, const zg_bases* bases, const zg_bases* bases_b, size_t split, const Fe* d_scalars,
                   size_t stride, size_t per, size_t outer, size_t batch, size_t n, XYZZ* d_out, uint64_t run_mask,
                   uint32_t naf_width);
int bases_enable_runs(zg_ctx* ctx, zg_bases* b);  // running-sum table for the run form (idempotent)
// b->dense: one row per bit position, odd w-bit digits (strict: refuse a table made for another default width)
int bases_enable_naf(zg_ctx* ctx, zg_bases* b, uint32_t w, bool strict = false);
int bases_register_dev(zg_ctx* ctx, const Affine* d_bases, size_t n, uint32_t window_bits, zg_bases** out);
// digit tables of the latency form (every multiple of every window; window_bits 0 = from n and the free memory, which may
// decide on none); with_runs: for the running sums too (the set must have its running-sum table)
int bases_enable_full(zg_ctx* ctx, zg_bases* b, uint32_t window_bits, bool with_runs);
uint32_t default_full_bits(size_t n, double budget_bytes);  // 0 = no digit tables at this size / budget
void xyzz_batch_normalise(const XYZZ* in, size_t count, zg_g1* out);

}  // namespace zg
