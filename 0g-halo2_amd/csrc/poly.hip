// Polynomial-side kernels of create_proof for gfx950: blinding, lookup compression, grand products
// (lookup::prover::commit_product, permutation::prover::commit), Evaluator::evaluate_h, eval_polynomial,
// the GWC linear combinations and kate_division (halo2_proofs v2023_04_20 src/plonk/{evaluation,
// lookup/prover, permutation/prover, vanishing/prover}.rs, src/poly/kzg/multiopen/gwc/prover.rs,
// src/arithmetic.rs; reached from /root/reference/src/wnn.rs:242-259).
//
// All of it is row-parallel 254-bit modular arithmetic over HBM-resident columns: one lane per row,
// coalesced 32-B loads, wave-uniform circuit tables read through the scalar cache.  The two sequential
// recurrences upstream runs on one core -- the running products z[i+1] = z[i]*num/den and the
// synthetic division by (X - z) -- become block-level scans (1024 lanes x a strip each).
#include "poly.h"
#include "field9.h"

namespace zg {

__device__ __forceinline__ Fe ldg(const Fe* p) {
    Fe r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void stg(Fe* p, const Fe& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// ------------------------------------------------------------------ blinding scalars
// rand_fr(key, tag, index): ChaCha20 (RFC 7539: 32-byte key, 32-bit block counter, 96-bit nonce) keystream with
// nonce = (tag, index_lo, index_hi) and block counter = attempt; a 64-byte block offers two 254-bit candidates
// (words 0-7, then 8-15, top word masked to 30 bits); the first one below r is taken.  Upstream draws these
// scalars from OsRng (/root/reference/src/wnn.rs:256): a caller that passes 32 fresh random bytes per proof gets
// 256-bit keyed blinding; the test checker restates the same function for byte parity.
__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int sft) { return (x << sft) | (x >> (32 - sft)); }
#define ZG_QR(a, b, c, d)                       \
    do {                                        \
        a += b; d ^= a; d = rotl32(d, 16);      \
        c += d; b ^= c; b = rotl32(b, 12);      \
        a += b; d ^= a; d = rotl32(d, 8);       \
        c += d; b ^= c; b = rotl32(b, 7);       \
    } while (0)

__host__ __device__ inline Fe rand_fr(const uint32_t* key, uint32_t tag, uint64_t index) {
    Fe v;
    for (uint32_t attempt = 0;; attempt++) {
        uint32_t x0 = 0x61707865u, x1 = 0x3320646eu, x2 = 0x79622d32u, x3 = 0x6b206574u;
        uint32_t x4 = key[0], x5 = key[1], x6 = key[2], x7 = key[3], x8 = key[4], x9 = key[5], x10 = key[6], x11 = key[7];
        uint32_t x12 = attempt, x13 = tag, x14 = (uint32_t)index, x15 = (uint32_t)(index >> 32);
        for (int r = 0; r < 10; r++) {
            ZG_QR(x0, x4, x8, x12);
            ZG_QR(x1, x5, x9, x13);
            ZG_QR(x2, x6, x10, x14);
            ZG_QR(x3, x7, x11, x15);
            ZG_QR(x0, x5, x10, x15);
            ZG_QR(x1, x6, x11, x12);
            ZG_QR(x2, x7, x8, x13);
            ZG_QR(x3, x4, x9, x14);
        }
        for (int half = 0; half < 2; half++) {
            if (half == 0) {
                v.l[0] = x0 + 0x61707865u; v.l[1] = x1 + 0x3320646eu; v.l[2] = x2 + 0x79622d32u; v.l[3] = x3 + 0x6b206574u;
                v.l[4] = x4 + key[0]; v.l[5] = x5 + key[1]; v.l[6] = x6 + key[2]; v.l[7] = x7 + key[3];
            } else {
                v.l[0] = x8 + key[4]; v.l[1] = x9 + key[5]; v.l[2] = x10 + key[6]; v.l[3] = x11 + key[7];
                v.l[4] = x12 + attempt; v.l[5] = x13 + tag; v.l[6] = x14 + (uint32_t)index; v.l[7] = x15 + (uint32_t)(index >> 32);
            }
            v.l[7] &= 0x3fffffffu;
            bool lt = false;
            for (int i = 7; i >= 0; i--) {
                uint32_t p = FrParams::p(i);
                if (v.l[i] < p) { lt = true; break; }
                if (v.l[i] > p) break;
            }
            if (lt) return Fr::from_raw(v);
        }
    }
}

// proof b = blockIdx.y: columns [0, ncols0) draw from tag0 (index c * nrows + j), columns [ncols0, ncols) from tag1
// (index restarts)
__global__ void blind_rows_kernel(const ProofConst* __restrict__ pc, Fe* base, size_t base_bs, size_t col_stride,
                                  uint32_t ncols, uint32_t row0, uint32_t nrows, uint32_t tag0, uint32_t ncols0, uint32_t tag1) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = blockIdx.y;
    if (t >= ncols * nrows) return;
    uint32_t c = t / nrows, j = t % nrows;
    const bool second = c >= ncols0;
    stg(base + (size_t)b * base_bs + (size_t)c * col_stride + row0 + j,
        rand_fr(pc[b].key, second ? tag1 : tag0, (uint64_t)(second ? c - ncols0 : c) * nrows + j));
}

int poly_blind_rows2(zg_ctx* ctx, const ProofConst* pc, uint32_t nb, Fe* base, size_t base_bs, size_t col_stride,
                     uint32_t ncols0, uint32_t tag0, uint32_t ncols1, uint32_t tag1, uint32_t row0, uint32_t nrows) {
    uint32_t total = (ncols0 + ncols1) * nrows;
    if (total == 0 || nb == 0) return ZG_OK;
    ZG_LAUNCH(ctx, "blind_rows", 0, blind_rows_kernel, dim3((total + 63) / 64, nb), dim3(64), 0, pc, base, base_bs, col_stride,
              ncols0 + ncols1, row0, nrows, tag0, ncols0, tag1);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// The two draws a proof starts with, in one launch: the vanishing argument's random polynomial (n values,
// written to out and out2) and the blinding rows of the advice columns.  Proof b = blockIdx.y.
__global__ void random_and_blind_kernel(const ProofConst* __restrict__ pc, Fe* out, size_t out_bs, Fe* out2, size_t out2_bs,
                                        uint32_t n, uint32_t tag, Fe* base, size_t base_bs, size_t col_stride, uint32_t ncols,
                                        uint32_t row0, uint32_t nrows, uint32_t blind_tag) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = blockIdx.y;
    if (i < n) {
        const Fe v = rand_fr(pc[b].key, tag, i);
        stg(out + (size_t)b * out_bs + i, v);
        if (out2) stg(out2 + (size_t)b * out2_bs + i, v);
        return;
    }
    i -= n;
    if (i >= ncols * nrows) return;
    const uint32_t c = i / nrows, j = i % nrows;
    stg(base + (size_t)b * base_bs + (size_t)c * col_stride + row0 + j, rand_fr(pc[b].key, blind_tag, (uint64_t)c * nrows + j));
}
int poly_random_and_blind(zg_ctx* ctx, const ProofConst* pc, uint32_t nb, Fe* out, size_t out_bs, Fe* out2, size_t out2_bs,
                          uint32_t n, uint32_t tag, Fe* base, size_t base_bs, size_t col_stride, uint32_t ncols,
                          uint32_t row0, uint32_t nrows, uint32_t blind_tag) {
    const uint32_t total = n + ncols * nrows;
    if (nb == 0) return ZG_OK;
    ZG_LAUNCH(ctx, "random_poly", (double)nb * n * 32, random_and_blind_kernel, dim3((total + 255) / 256, nb), dim3(256), 0, pc,
              out, out_bs, out2, out2_bs, n, tag, base, base_bs, col_stride, ncols, row0, nrows, blind_tag);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// ------------------------------------------------------------------ expression interpreter
__device__ __forceinline__ Fe eval_poly(const DevCircuit& c, const Cols& cols, zg_poly p, uint32_t row) {
    const uint32_t mask = (1u << cols.log_size) - 1u;
    Fe acc = fe_zero();
    for (uint32_t m = p.first; m < p.first + p.count; m++) {
        const DMono* mo = c.monos + m;
        const uint32_t nf = mo->n_factors;
        Fe prod;
        uint32_t f = 0;
        if (mo->coeff_is_one == 1 && nf > 0) {
            const zg_query q = c.queries[mo->factors[0]];
            const Fe* base = q.kind == ZG_FIXED ? cols.fixed : q.kind == ZG_ADVICE ? cols.advice : cols.instance;
            uint32_t idx = (row + (uint32_t)(q.rotation * cols.rot_scale)) & mask;
            prod = ldg(base + ((size_t)q.column << cols.log_size) + idx);
            f = 1;
        } else {
            prod = mo->coeff;
        }
        for (; f < nf; f++) {
            const zg_query q = c.queries[mo->factors[f]];
            const Fe* base = q.kind == ZG_FIXED ? cols.fixed : q.kind == ZG_ADVICE ? cols.advice : cols.instance;
            uint32_t idx = (row + (uint32_t)(q.rotation * cols.rot_scale)) & mask;
            prod = Fr::mul(prod, ldg(base + ((size_t)q.column << cols.log_size) + idx));
        }
        acc = Fr::add(acc, prod);
    }
    return acc;
}

// lookup::Argument::commit_permuted `compress_expressions`: theta-fold of the input / table tuples
// raw_in / raw_tab (optional): the same values as canonical integers -- the sort keys of permute_expression_pair --
// with the all-ones sentinel on the rows from `usable` on (sorts last; real keys are < r < 2^254)
__device__ __forceinline__ Cols cols_of(const Cols& c, uint32_t b) {  // proof b's view of a batch's columns
    Cols r = c;
    r.advice += (size_t)b * c.adv_bs;
    r.instance += (size_t)b * c.inst_bs;
    return r;
}

__global__ __launch_bounds__(256) void lookup_compress_kernel(DevCircuit c, Cols cols_all, const ProofConst* __restrict__ pc,
                                                              Fe* cin, Fe* ctab, uint32_t n, Fe* raw_in, Fe* raw_tab,
                                                              uint32_t usable) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = blockIdx.z;
    uint32_t l = blockIdx.y;
    if (row >= n) return;
    const Cols cols = cols_of(cols_all, b);
    const Fe theta = pc[b].theta;
    const DLookup* lk = c.lookups + l;
    l += b * c.n_lookups;  // (outputs: lookup l of proof b)
    Fe ai = fe_zero(), ti = fe_zero();
    for (uint32_t e = 0; e < lk->width; e++) {
        ai = Fr::add(Fr::mul(ai, theta), eval_poly(c, cols, lk->inputs[e], row));
        ti = Fr::add(Fr::mul(ti, theta), eval_poly(c, cols, lk->tables[e], row));
    }
    stg(cin + (size_t)l * n + row, ai);
    stg(ctab + (size_t)l * n + row, ti);
    if (raw_in) {
        Fe ra, rt;
        if (row < usable) {
            ra = Fr::to_raw(ai);
            rt = Fr::to_raw(ti);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) ra.l[j] = rt.l[j] = 0xffffffffu;
        }
        stg(raw_in + (size_t)l * n + row, ra);
        stg(raw_tab + (size_t)l * n + row, rt);
    }
}

int poly_lookup_compress(zg_ctx* ctx, const DevCircuit& c, const Cols& cols, const ProofConst* pc, uint32_t nb, Fe* cin,
                         Fe* ctab, uint32_t n, Fe* raw_in, Fe* raw_tab, uint32_t usable) {
    if (c.n_lookups == 0 || nb == 0) return ZG_OK;
    ZG_LAUNCH(ctx, "lookup_compress", (double)nb * c.n_lookups * n * 64, lookup_compress_kernel,
              dim3((n + 255) / 256, c.n_lookups, nb), dim3(256), 0, c, cols, pc, cin, ctab, n, raw_in, raw_tab, usable);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

__global__ void to_raw_kernel(const Fe* in, Fe* out, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) stg(out + i, Fr::to_raw(ldg(in + i)));
}
__global__ void from_raw_kernel(const Fe* in, Fe* out, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) stg(out + i, Fr::from_raw(ldg(in + i)));
}
// rows of `len` elements: dst[r*dst_stride + i] = from_raw(src[r*src_stride + i])
__global__ void from_raw_rows_kernel(const Fe* src, size_t src_stride, Fe* dst, size_t dst_stride, uint32_t len) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (i < len) stg(dst + (size_t)r * dst_stride + i, Fr::from_raw(ldg(src + (size_t)r * src_stride + i)));
}
int poly_from_raw_rows(zg_ctx* ctx, const Fe* src, size_t src_stride, Fe* dst, size_t dst_stride, uint32_t rows,
                       uint32_t len) {
    if (!rows || !len) return ZG_OK;
    ZG_LAUNCH(ctx, "from_raw", (double)rows * len * 64, from_raw_rows_kernel, dim3((len + 255) / 256, rows), dim3(256), 0, src,
              src_stride, dst, dst_stride, len);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// lookup commit_permuted's last step in one launch: perm[2l] = a'_l, perm[2l+1] = s'_l (Montgomery form) on
// the usable rows [0, usable), blinding scalars on the rows after them (tag_in / tag_tab, index = l * nblind + j)
__global__ __launch_bounds__(256) void permuted_finish_kernel(const ProofConst* __restrict__ pc, const Fe* __restrict__ raw_in,
                                                              const Fe* __restrict__ raw_tab, Fe* __restrict__ perm, size_t perm_bs,
                                                              uint32_t n, uint32_t usable, uint32_t nblind, uint32_t tag_in,
                                                              uint32_t tag_tab) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y, b = blockIdx.z;
    if (i >= usable + nblind) return;
    Fe* a = perm + (size_t)b * perm_bs + (size_t)(2 * l) * n;
    Fe* t = a + n;
    const size_t src = (size_t)(b * gridDim.y + l) * n;  // (raw rows: lookup l of proof b)
    if (i < usable) {
        stg(a + i, Fr::from_raw(ldg(raw_in + src + i)));
        stg(t + i, Fr::from_raw(ldg(raw_tab + src + i)));
    } else {
        const uint64_t idx = (uint64_t)l * nblind + (i - usable);
        stg(a + i, rand_fr(pc[b].key, tag_in, idx));
        stg(t + i, rand_fr(pc[b].key, tag_tab, idx));
    }
}
int poly_permuted_finish(zg_ctx* ctx, const ProofConst* pc, uint32_t nb, const Fe* raw_in, const Fe* raw_tab, Fe* perm,
                         size_t perm_bs, uint32_t n, uint32_t usable, uint32_t nblind, uint32_t n_lookups, uint32_t tag_in,
                         uint32_t tag_tab) {
    if (!n_lookups || !nb) return ZG_OK;
    ZG_REQUIRE(usable + nblind <= n, ZG_ERR_INVALID_ARG, "poly_permuted_finish: %u + %u rows of %u", usable, nblind, n);
    ZG_LAUNCH(ctx, "permuted_finish", (double)nb * n_lookups * n * 128, permuted_finish_kernel,
              dim3((usable + nblind + 255) / 256, n_lookups, nb), dim3(256), 0, pc, raw_in, raw_tab, perm, perm_bs, n, usable,
              nblind, tag_in, tag_tab);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int poly_to_raw(zg_ctx* ctx, const Fe* in, Fe* out, size_t count) {
    if (!count) return ZG_OK;
    ZG_LAUNCH(ctx, "to_raw", (double)count * 64, to_raw_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, in, out, count);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}
int poly_from_raw(zg_ctx* ctx, const Fe* in, Fe* out, size_t count) {
    if (!count) return ZG_OK;
    ZG_LAUNCH(ctx, "from_raw", (double)count * 64, from_raw_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, in, out, count);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// ------------------------------------------------------------------ grand-product terms
// lookup commit_product: den = (a' + beta)(s' + gamma), num = (A + beta)(S + gamma)
// (lookup l = blockIdx.y of proof b = blockIdx.z: compressed columns at (b * n_lookups + l) * n, permuted columns at
//  b * perm_bs + l * perm_stride, outputs at product (b * per + first + l) * n)
__global__ __launch_bounds__(256) void lookup_terms_kernel(const ProofConst* __restrict__ pc, const Fe* cin, const Fe* ctab,
                                                           const Fe* pin, const Fe* ptab, size_t perm_stride, size_t perm_bs,
                                                           Fe* num, Fe* den, uint32_t per, uint32_t first, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t l = blockIdx.y, b = blockIdx.z;
    if (i >= n) return;
    const Fe beta = pc[b].beta, gamma = pc[b].gamma;
    const size_t ci = (size_t)(b * gridDim.y + l) * n + i, po = (size_t)b * perm_bs + (size_t)l * perm_stride + i;
    const size_t o = (size_t)(b * per + first + l) * n + i;
    stg(den + o, Fr::mul(Fr::add(ldg(pin + po), beta), Fr::add(ldg(ptab + po), gamma)));
    stg(num + o, Fr::mul(Fr::add(ldg(cin + ci), beta), Fr::add(ldg(ctab + ci), gamma)));
}

int poly_lookup_terms(zg_ctx* ctx, const ProofConst* pc, uint32_t nb, const Fe* cin, const Fe* ctab, const Fe* pin,
                      const Fe* ptab, size_t perm_stride, size_t perm_bs, Fe* num, Fe* den, uint32_t per, uint32_t first,
                      uint32_t n, uint32_t n_lookups) {
    if (!n_lookups || !nb) return ZG_OK;
    ZG_LAUNCH(ctx, "lookup_terms", (double)nb * n_lookups * n * 192, lookup_terms_kernel, dim3((n + 255) / 256, n_lookups, nb),
              dim3(256), 0, pc, cin, ctab, pin, ptab, perm_stride, perm_bs, num, den, per, first, n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// permutation commit, one set per blockIdx.y:
//   den = prod_c (v_c + beta*sigma_c + gamma),  num = prod_c (v_c + delta^c * omega^i * beta + gamma)
__global__ __launch_bounds__(256) void perm_terms_kernel(DevCircuit c, Cols cols_all, const ProofConst* __restrict__ pc,
                                                         const Fe* sigma_val, const Fe* omega_tw, Fe* num, Fe* den,
                                                         uint32_t per, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = blockIdx.y;
    const uint32_t b = blockIdx.z;
    if (i >= n) return;
    const Cols cols = cols_of(cols_all, b);
    const Fe beta = pc[b].beta, gamma = pc[b].gamma;
    uint32_t c0 = s * c.chunk, c1 = c0 + c.chunk;
    if (c1 > c.n_perm) c1 = c.n_perm;
    Fe d = Fr::one(), m = Fr::one();
    Fe dw = Fr::mul(Fr::mul(Fr::pow_u64(fr_delta(), c0), ldg(omega_tw + i)), beta);  // delta^c omega^i beta
    for (uint32_t col = c0; col < c1; col++) {
        const zg_query q = c.perm_cols[col];
        const Fe* base = q.kind == ZG_FIXED ? cols.fixed : q.kind == ZG_ADVICE ? cols.advice : cols.instance;
        Fe v = ldg(base + ((size_t)q.column << cols.log_size) + i);
        Fe sg = ldg(sigma_val + (size_t)col * n + i);
        d = Fr::mul(d, Fr::add(Fr::add(Fr::mul(beta, sg), gamma), v));
        m = Fr::mul(m, Fr::add(Fr::add(dw, gamma), v));
        dw = Fr::mul(dw, fr_delta());
    }
    stg(den + (size_t)(b * per + s) * n + i, d);
    stg(num + (size_t)(b * per + s) * n + i, m);
}

int poly_perm_terms(zg_ctx* ctx, const DevCircuit& c, const Cols& cols, const ProofConst* pc, uint32_t nb,
                    const Fe* sigma_val, const Fe* omega_tw, Fe* num, Fe* den, uint32_t per, uint32_t n) {
    if (!c.n_sets || !nb) return ZG_OK;
    ZG_LAUNCH(ctx, "perm_terms", nb * ((double)c.n_perm * n * 64 + (double)c.n_sets * n * 64), perm_terms_kernel,
              dim3((n + 255) / 256, c.n_sets, nb), dim3(256), 0, c, cols, pc, sigma_val, omega_tw, num, den, per, n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// ------------------------------------------------------------------ grand product
// z[0] = z0, z[i+1] = z[i] * num[i] / den[i], as z[i] = z0 * N(i-1) * D'(i) / T with
//   N(i)  = prod_{j<=i} num'_j          (prefix products)
//   D'(i) = prod_{j>=i} den'_j          (suffix products),  T = prod_j den'_j
// where a zero denominator counts as den' = 1, num' = 0 (halo2's BatchInvert leaves zeros alone, so
// that ratio is 0): ONE field inversion per product instead of one per row.  Strip form, two launches for ALL
// products of a batch, one workgroup of GP_LANES lanes per product, lane l owning rows [l S, (l + 1) S):
//   gp_strip_scan   the strip's num' product and its den' suffix products (kept in HBM, one per row), then prefix /
//                   suffix scans of the strip products over the lanes (Hillis-Steele through LDS), 1 / T, and the
//                   product's value at row `last` for z0 = 1 (feeds the chain);
//   gp_strip_apply  z_i = k_l * (num' prefix inside the strip) * (den' suffix inside the strip) with the lane constant
//                   k_l = z0 * chain / T * N(before the strip) * D'(after the strip): two products per row.
// About 6.5 products per row where the block-scan form (three launches) spent 21.  `chain` leading products of a group
// are chained like the permutation sets: z0 of product j is product j-1's value at row `last`.
// ---- latency form (a lone proof: zg_ctx_set_msm_latency / zg_prover_set_overlap(1)): block-local Hillis-Steele scans,
// coalesced accesses, three short launches -- more products per row (21), but a lone proof waits for dependent chains
// and for memory latency, not for issue slots: 41 us against the strip form's 170 us for the six products of a proof.
constexpr uint32_t GP_BLOCK = 256;

__global__ __launch_bounds__(GP_BLOCK) void gp_local_kernel(const Fe* __restrict__ num, const Fe* __restrict__ den,
                                                             Fe* __restrict__ locn, Fe* __restrict__ locd,
                                                             Fe* __restrict__ totn, Fe* __restrict__ totd, uint32_t n,
                                                             uint32_t nblk) {
    __shared__ Fe shn[GP_BLOCK], shd[GP_BLOCK];
    const uint32_t tid = threadIdx.x, blk = blockIdx.x, b = blockIdx.y;
    const uint32_t i = blk * GP_BLOCK + tid;
    Fe nv = Fr::one(), dv = Fr::one();
    if (i < n) {
        Fe d = ldg(den + (size_t)b * n + i);
        if (fe_is_zero(d)) {
            nv = fe_zero();
        } else {
            nv = ldg(num + (size_t)b * n + i);
            dv = d;
        }
    }
    shn[tid] = nv;
    shd[tid] = dv;
    __syncthreads();
    for (uint32_t off = 1; off < GP_BLOCK; off <<= 1) {
        Fe pn = Fr::one(), pd = Fr::one();
        const bool hn = tid >= off, hd = tid + off < GP_BLOCK;
        if (hn) pn = shn[tid - off];
        if (hd) pd = shd[tid + off];
        __syncthreads();
        if (hn) shn[tid] = Fr::mul(shn[tid], pn);
        if (hd) shd[tid] = Fr::mul(shd[tid], pd);
        __syncthreads();
    }
    if (i < n) {
        stg(locn + (size_t)b * n + i, shn[tid]);  // inclusive prefix inside the block
        stg(locd + (size_t)b * n + i, shd[tid]);  // inclusive suffix inside the block
    }
    if (tid == 0) {
        stg(totn + (size_t)b * nblk + blk, shn[GP_BLOCK - 1]);
        stg(totd + (size_t)b * nblk + blk, shd[0]);
    }
}

// per product: exclusive prefix of the block totals of num', exclusive suffix of those of den', 1/T,
// and the product's value at row `last` for z0 = 1 (feeds the chain)
__global__ __launch_bounds__(1024) void gp_totals_kernel(Fe* __restrict__ totn, Fe* __restrict__ totd,
                                                         const Fe* __restrict__ locn, const Fe* __restrict__ locd,
                                                         Fe* __restrict__ tinv, Fe* __restrict__ zlast, uint32_t n,
                                                         uint32_t nblk, uint32_t last, uint32_t defer_inverse) {
    __shared__ Fe shn[1024], shd[1024];
    const uint32_t tid = threadIdx.x, b = blockIdx.x;
    Fe* tn = totn + (size_t)b * nblk;
    Fe* td = totd + (size_t)b * nblk;
    Fe vn = Fr::one(), vd = Fr::one();
    if (tid < nblk) {
        vn = ldg(tn + tid);
        vd = ldg(td + tid);
    }
    shn[tid] = vn;
    shd[tid] = vd;
    __syncthreads();
    // (entries at and beyond nblk are ones: the scan only has to span the block totals)
    uint32_t span = 64;
    while (span < nblk) span <<= 1;
    for (uint32_t off = 1; off < span; off <<= 1) {
        Fe pn = Fr::one(), pd = Fr::one();
        const bool hn = tid >= off && tid < span, hd = tid + off < span;
        if (hn) pn = shn[tid - off];
        if (hd) pd = shd[tid + off];
        __syncthreads();
        if (hn) shn[tid] = Fr::mul(shn[tid], pn);
        if (hd) shd[tid] = Fr::mul(shd[tid], pd);
        __syncthreads();
    }
    // exclusive forms
    Fe en = tid > 0 ? shn[tid - 1] : Fr::one();
    Fe ed = tid + 1 < 1024 ? shd[tid + 1] : Fr::one();
    Fe total_d = shd[0];
    __syncthreads();
    if (tid < nblk) {
        stg(tn + tid, en);
        stg(td + tid, ed);
    }
    shn[tid] = en;
    shd[tid] = ed;
    __syncthreads();
    if (tid == 0) {
        // defer_inverse: the host inverts T (tinv[b] <- T, zlast[b] <- the product without 1/T); a lone lane
        // spends ~0.1 ms on it, the host a few microseconds -- worth a round trip when one proof is all there is
        Fe ti = defer_inverse ? total_d : Fr::inv(total_d);
        stg(tinv + b, ti);
        // z(last) for z0 = 1: N(last-1) * D'(last) / T
        Fe v = defer_inverse ? Fr::one() : ti;
        if (last < n) {
            uint32_t lb = last / GP_BLOCK;
            v = Fr::mul(v, Fr::mul(ldg(locd + (size_t)b * n + last), shd[lb]));
            if (last > 0) {
                uint32_t pb = (last - 1) / GP_BLOCK;
                v = Fr::mul(v, Fr::mul(ldg(locn + (size_t)b * n + last - 1), shn[pb]));
            }
        }
        stg(zlast + b, v);
    }
}

__global__ __launch_bounds__(GP_BLOCK) void gp_apply_kernel(const Fe* __restrict__ locn, const Fe* __restrict__ locd,
                                                            const Fe* __restrict__ totn, const Fe* __restrict__ totd,
                                                            const Fe* __restrict__ tinv, const Fe* __restrict__ zlast,
                                                            const Fe* __restrict__ z0, Fe* __restrict__ z, uint32_t n,
                                                            uint32_t nblk, uint32_t chain, FeSet host_tinv, uint32_t use_host,
                                                            uint32_t per, size_t z_outer) {
    const uint32_t tid = threadIdx.x, blk = blockIdx.x, b = blockIdx.y;
    const uint32_t i = blk * GP_BLOCK + tid;
    if (i >= n) return;
    const uint32_t j = b % per, g0 = b - j;  // product j of its group; the group's first product
    // start value: z0[b] (if given) times the chained last values of the group's products before this one
    // (use_host: 1/T arrives in the kernel arguments and zlast still lacks that factor; one group only)
    Fe c = z0 ? ldg(z0 + b) : Fr::one();
    if (j < chain)
        for (uint32_t s = 0; s < j; s++) {
            c = Fr::mul(c, ldg(zlast + g0 + s));
            if (use_host) c = Fr::mul(c, host_tinv.v[s]);
        }
    Fe v = Fr::mul(c, use_host ? host_tinv.v[b] : ldg(tinv + b));
    v = Fr::mul(v, Fr::mul(ldg(locd + (size_t)b * n + i), ldg(totd + (size_t)b * nblk + blk)));
    if (i > 0) {
        uint32_t pb = (i - 1) / GP_BLOCK;
        v = Fr::mul(v, Fr::mul(ldg(locn + (size_t)b * n + i - 1), ldg(totn + (size_t)b * nblk + pb)));
    }
    stg(z + (size_t)(b / per) * z_outer + (size_t)j * n + i, v);
}

// ---- throughput form
constexpr uint32_t GP_LANES = 1024;

__device__ __forceinline__ void gp_load(const Fe* num, const Fe* den, size_t at, Fe& nv, Fe& dv) {
    dv = ldg(den + at);
    if (fe_is_zero(dv)) {
        nv = fe_zero();
        dv = Fr::one();
    } else {
        nv = ldg(num + at);
    }
}

// aux per product: [lanes] N before the strip, [lanes] D' after the strip
__global__ __launch_bounds__(GP_LANES) void gp_strip_scan_kernel(const Fe* __restrict__ num, const Fe* __restrict__ den,
                                                                 Fe* __restrict__ locd, Fe* __restrict__ aux,
                                                                 Fe* __restrict__ tinv, Fe* __restrict__ zlast, uint32_t n,
                                                                 uint32_t lanes, uint32_t strip, uint32_t last,
                                                                 uint32_t defer_inverse) {
    __shared__ Fe shn[GP_LANES], shd[GP_LANES];
    __shared__ Fe s_n_last, s_d_last;
    const uint32_t tid = threadIdx.x, b = blockIdx.x;
    const size_t base = (size_t)b * n;
    const uint32_t i0 = tid * strip;
    const uint32_t i1 = tid < lanes ? (i0 + strip < n ? i0 + strip : n) : i0;
    // ascending: the strip's num' product (and its value up to row last - 1, if that row is here)
    Fe pn = Fr::one(), n_at_last = Fr::one();
    for (uint32_t i = i0; i < i1; i++) {
        Fe nv, dv;
        gp_load(num, den, base + i, nv, dv);
        pn = Fr::mul(pn, nv);
        if (i + 1 == last) n_at_last = pn;
    }
    // descending: den' suffix products inside the strip, one per row
    Fe sd = Fr::one(), d_at_last = Fr::one();
    for (uint32_t i = i1; i-- > i0;) {
        Fe dv = ldg(den + base + i);
        if (!fe_is_zero(dv)) sd = Fr::mul(sd, dv);
        stg(locd + base + i, sd);
        if (i == last) d_at_last = sd;
    }
    shn[tid] = pn;
    shd[tid] = sd;
    __syncthreads();
    for (uint32_t off = 1; off < GP_LANES; off <<= 1) {
        Fe a = Fr::one(), d = Fr::one();
        const bool hn = tid >= off, hd = tid + off < GP_LANES;
        if (hn) a = shn[tid - off];
        if (hd) d = shd[tid + off];
        __syncthreads();
        if (hn) shn[tid] = Fr::mul(shn[tid], a);
        if (hd) shd[tid] = Fr::mul(shd[tid], d);
        __syncthreads();
    }
    const Fe n_before = tid > 0 ? shn[tid - 1] : Fr::one();
    const Fe d_after = tid + 1 < GP_LANES ? shd[tid + 1] : Fr::one();
    if (tid < lanes) {
        stg(aux + (size_t)b * 2 * lanes + tid, n_before);
        stg(aux + (size_t)b * 2 * lanes + lanes + tid, d_after);
    }
    // the value at row `last` for z0 = 1: N(last - 1) * D'(last) / T, pieces from the lanes that own those rows
    if (tid == 0) {
        s_n_last = Fr::one();
        s_d_last = Fr::one();
    }
    __syncthreads();
    if (last < n) {
        if (last > 0 && last - 1 >= i0 && last - 1 < i1) s_n_last = Fr::mul(n_before, n_at_last);
        if (last >= i0 && last < i1) s_d_last = Fr::mul(d_after, d_at_last);
    }
    __syncthreads();
    if (tid == 0) {
        // defer_inverse: the host inverts T (tinv[b] <- T, zlast[b] <- the product without 1/T); a lone lane
        // spends ~0.1 ms on it, the host a few microseconds -- worth a round trip when one proof is all there is
        const Fe total_d = shd[0];
        const Fe ti = defer_inverse ? total_d : Fr::inv(total_d);
        stg(tinv + b, ti);
        Fe v = defer_inverse ? Fr::one() : ti;
        if (last < n) v = Fr::mul(v, Fr::mul(s_n_last, s_d_last));
        stg(zlast + b, v);
    }
}

__global__ __launch_bounds__(256) void gp_strip_apply_kernel(const Fe* __restrict__ num, const Fe* __restrict__ den,
                                                             const Fe* __restrict__ locd, const Fe* __restrict__ aux,
                                                             const Fe* __restrict__ tinv, const Fe* __restrict__ zlast,
                                                             const Fe* __restrict__ z0, Fe* __restrict__ z, uint32_t n,
                                                             uint32_t lanes, uint32_t strip, uint32_t chain, FeSet host_tinv,
                                                             uint32_t use_host, uint32_t per, size_t z_outer) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (l >= lanes) return;
    const uint32_t j = b % per, g0 = b - j;  // product j of its group; the group's first product
    // start value: z0[b] (if given) times the chained last values of the group's products before this one
    // (use_host: 1/T arrives in the kernel arguments and zlast still lacks that factor; one group only)
    Fe c = z0 ? ldg(z0 + b) : Fr::one();
    if (j < chain)
        for (uint32_t s = 0; s < j; s++) {
            c = Fr::mul(c, ldg(zlast + g0 + s));
            if (use_host) c = Fr::mul(c, host_tinv.v[s]);
        }
    Fe run = Fr::mul(c, use_host ? host_tinv.v[b] : ldg(tinv + b));
    run = Fr::mul(run, Fr::mul(ldg(aux + (size_t)b * 2 * lanes + l), ldg(aux + (size_t)b * 2 * lanes + lanes + l)));
    const size_t base = (size_t)b * n;
    Fe* zo = z + (size_t)(b / per) * z_outer + (size_t)j * n;
    const uint32_t i0 = l * strip, i1 = i0 + strip < n ? i0 + strip : n;
    for (uint32_t i = i0; i < i1; i++) {
        stg(zo + i, Fr::mul(run, ldg(locd + base + i)));  // k_l * N(strip start .. i-1) * D'(i .. strip end)
        Fe nv, dv;
        gp_load(num, den, base + i, nv, dv);
        run = Fr::mul(run, nv);
    }
}

// tmp: batch*n (den' suffixes) + 2*batch*GP_LANES (lane constants) + 2*batch elements
size_t poly_grand_product_tmp_elems(uint32_t n, uint32_t batch) {  // (room for either form)
    const size_t nblk = (n + GP_BLOCK - 1) / GP_BLOCK;
    const size_t strips = (size_t)batch * n + (size_t)2 * batch * GP_LANES + (size_t)2 * batch;
    const size_t blocks = (size_t)2 * batch * n + (size_t)2 * batch * nblk + (size_t)2 * batch;
    return strips > blocks ? strips : blocks;
}

// the latency form's launch sequence (tmp: 2*batch*n + 2*batch*nblk + 2*batch elements)
static int grand_product_blocks(zg_ctx* ctx, const Fe* num, const Fe* den, const Fe* d_z0, Fe* z, Fe* tmp, uint32_t n,
                                uint32_t batch, uint32_t chain, uint32_t last, uint32_t per, size_t z_outer, int half) {
    const uint32_t nblk = (n + GP_BLOCK - 1) / GP_BLOCK;
    ZG_REQUIRE(nblk <= 1024, ZG_ERR_UNSUPPORTED, "grand product: n=%u > 2^18 not built", n);
    Fe* locn = tmp;
    Fe* locd = locn + (size_t)batch * n;
    Fe* totn = locd + (size_t)batch * n;
    Fe* totd = totn + (size_t)batch * nblk;
    Fe* tinv = totd + (size_t)batch * nblk;
    Fe* zlast = tinv + batch;
    // the UNIT: one running product = num, den in, z out = 3 n * 32 B, carried by the apply kernel; each
    // kernel's own streams beside it (the block-local products cross HBM between the launches)
    const double bytes = (double)batch * n * 96;
    if (half != 2)
        ZG_LAUNCH(ctx, "grand_product_local", (double)batch * n * 128, gp_local_kernel, dim3(nblk, batch), dim3(GP_BLOCK), 0, num, den, locn,
                  locd, totn, totd, n, nblk);
    // latency configuration: the host inverts the totals (one shared inversion) between the two launches
    const bool host_inv = ctx->msm_pair && batch <= FESET_MAX && batch == per;
    // (... which the totals kernel writes straight into the context's pinned, mapped host buffer: no copy command between
    //  the kernel and the host's wait)
    Fe* h = nullptr;
    void* h_dev = nullptr;
    if (host_inv) {
        ZG_TRY(pinned_reserve(ctx, 4096));
        h = reinterpret_cast<Fe*>(ctx->pinned);
        ZG_HIP(hipHostGetDevicePointer(&h_dev, ctx->pinned, 0));
    }
    if (half != 2)
        ZG_LAUNCH(ctx, "grand_product_totals", (double)batch * nblk * 128, gp_totals_kernel, dim3(batch), dim3(1024), 0, totn, totd, locn, locd,
                  host_inv ? reinterpret_cast<Fe*>(h_dev) : tinv, zlast, n, nblk, last, host_inv ? 1u : 0u);
    if (half == 1) {
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    FeSet inv_set;
    memset(&inv_set, 0, sizeof(inv_set));
    if (host_inv) {
        ZG_HIP(hipStreamSynchronize(ctx->stream));
        // Montgomery's trick: prefix products, one inversion, walk back (a zero total inverts to zero, as Fr::inv does)
        Fe pre[FESET_MAX], acc = Fr::one();
        for (uint32_t b = 0; b < batch; b++) {
            pre[b] = acc;
            if (!fe_is_zero(h[b])) acc = Fr::mul(acc, h[b]);
        }
        acc = Fr::inv(acc);
        for (uint32_t b = batch; b-- > 0;) {
            if (fe_is_zero(h[b])) {
                inv_set.v[b] = fe_zero();
                continue;
            }
            inv_set.v[b] = Fr::mul(acc, pre[b]);
            acc = Fr::mul(acc, h[b]);
        }
    }
    ZG_LAUNCH_U(ctx, "grand_product_apply", bytes, bytes, gp_apply_kernel, dim3(nblk, batch), dim3(GP_BLOCK), 0, locn, locd, totn,
              totd, tinv, zlast, d_z0, z, n, nblk, chain, inv_set, host_inv ? 1u : 0u, per, z_outer);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}


int poly_grand_product(zg_ctx* ctx, const Fe* num, const Fe* den, const Fe* d_z0, Fe* z, Fe* tmp, uint32_t n,
                       uint32_t batch, uint32_t chain, uint32_t last, uint32_t per, size_t z_outer, int half) {
    if (!batch || !n) return ZG_OK;
    if (per == 0) {  // one group, results back to back
        per = batch;
        z_outer = 0;
    }
    ZG_REQUIRE(batch % per == 0, ZG_ERR_INVALID_ARG, "grand product: %u products in groups of %u", batch, per);
    if (batch == per) z_outer = 0;
    if (ctx->msm_pair && n <= (1u << 18)) return grand_product_blocks(ctx, num, den, d_z0, z, tmp, n, batch, chain, last, per, z_outer, half);
    const uint32_t lanes = n < GP_LANES ? n : GP_LANES;
    const uint32_t strip = (n + lanes - 1) / lanes;
    Fe* locd = tmp;
    Fe* aux = locd + (size_t)batch * n;
    Fe* tinv = aux + (size_t)2 * batch * GP_LANES;
    Fe* zlast = tinv + batch;
    const double bytes = (double)batch * n * 96;
    // latency configuration: the host inverts the totals (one shared inversion) between the two launches
    const bool host_inv = ctx->msm_pair && batch <= FESET_MAX && batch == per;
    if (half != 2)
        ZG_LAUNCH(ctx, "grand_product_scan", bytes, gp_strip_scan_kernel, dim3(batch), dim3(GP_LANES), 0, num, den, locd, aux, tinv,
                  zlast, n, lanes, strip, last, host_inv ? 1u : 0u);
    if (host_inv) ZG_TRY(pinned_reserve(ctx, 4096));
    Fe* h = reinterpret_cast<Fe*>(ctx->pinned);
    if (host_inv && half != 2) ZG_HIP(hipMemcpyAsync(h, tinv, batch * sizeof(Fe), hipMemcpyDeviceToHost, ctx->stream));
    if (half == 1) {
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    FeSet inv_set;
    memset(&inv_set, 0, sizeof(inv_set));
    if (host_inv) {
        ZG_HIP(hipStreamSynchronize(ctx->stream));
        // Montgomery's trick: prefix products, one inversion, walk back (a zero total inverts to zero, as Fr::inv does)
        Fe pre[FESET_MAX], acc = Fr::one();
        for (uint32_t b = 0; b < batch; b++) {
            pre[b] = acc;
            if (!fe_is_zero(h[b])) acc = Fr::mul(acc, h[b]);
        }
        acc = Fr::inv(acc);
        for (uint32_t b = batch; b-- > 0;) {
            if (fe_is_zero(h[b])) {
                inv_set.v[b] = fe_zero();
                continue;
            }
            inv_set.v[b] = Fr::mul(acc, pre[b]);
            acc = Fr::mul(acc, h[b]);
        }
    }
    ZG_LAUNCH_U(ctx, "grand_product_apply", (double)batch * n * 128, bytes, gp_strip_apply_kernel, dim3((lanes + 255) / 256, batch), dim3(256), 0, num, den,
              locd, aux, tinv, zlast, d_z0, z, n, lanes, strip, chain, inv_set, host_inv ? 1u : 0u, per, z_outer);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// ------------------------------------------------------------------ evaluate_h
// Evaluator::evaluate_h for one circuit instance: gates, permutation argument, lookup arguments folded
// by y on every point of the extended coset; the division by (X^n - 1) of vanishing::construct is
// fused into the store (t_eval has period 2^(ext_k - k)).
// One proof of a lock-step batch per grid row: its scalars, coset slabs and output (both kernels below).
struct EvalHProof {
    const ProofConst* pc;
    Cols cols;
    const Fe *pz_cos, *lz_cos, *pin_cos, *ptab_cos;
    Fe* h;
};
__device__ __forceinline__ EvalHProof evalh_proof(const EvalHArgs& a, uint32_t b) {
    EvalHProof r;
    r.pc = a.pc + b;
    r.cols = cols_of(a.cols, b);
    const size_t o = (size_t)b * a.cos_bs;
    r.pz_cos = a.pz_cos + o;
    r.lz_cos = a.lz_cos + o;
    r.pin_cos = a.pin_cos + o;
    r.ptab_cos = a.ptab_cos + o;
    r.h = a.h + (size_t)b * a.h_bs;
    return r;
}

__global__ __launch_bounds__(256) void evaluate_h_kernel(EvalHArgs a, uint32_t en) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= en) return;
    const EvalHProof pr = evalh_proof(a, blockIdx.y);
    const Fe y = pr.pc->eh_y, beta = pr.pc->eh_beta, gamma = pr.pc->eh_gamma, theta = pr.pc->eh_theta;
    const DevCircuit& c = a.c;
    const uint32_t mask = en - 1;
    const uint32_t rs = (uint32_t)a.cols.rot_scale;
    const uint32_t r_next = (idx + rs) & mask;
    const uint32_t r_prev = (idx - rs) & mask;
    const uint32_t r_last = (idx + (uint32_t)(a.last_rot * (int32_t)rs)) & mask;
    Fe value = fe_zero();
    for (uint32_t g = 0; g < c.n_gates; g++)
        value = Fr::add(Fr::mul(value, y), eval_poly(c, pr.cols, c.gates[g], idx));

    const Fe l0 = ldg(a.l0 + idx), llast = ldg(a.llast + idx), lactive = ldg(a.lactive + idx);
    if (c.n_sets > 0) {
        const Fe zf = ldg(pr.pz_cos + idx);
        const Fe zl = ldg(pr.pz_cos + (size_t)(c.n_sets - 1) * en + idx);
        value = Fr::add(Fr::mul(value, y), Fr::mul(Fr::sub(Fr::one(), zf), l0));
        value = Fr::add(Fr::mul(value, y), Fr::mul(Fr::sub(Fr::sqr(zl), zl), llast));
        for (uint32_t s = 1; s < c.n_sets; s++) {
            Fe t = Fr::sub(ldg(pr.pz_cos + (size_t)s * en + idx), ldg(pr.pz_cos + (size_t)(s - 1) * en + r_last));
            value = Fr::add(Fr::mul(value, y), Fr::mul(t, l0));
        }
        Fe current_delta = Fr::mul(pr.pc->eh_delta_start[a.zpow - 1], ldg(a.ext_tw + idx));
        for (uint32_t s = 0; s < c.n_sets; s++) {
            uint32_t c0 = s * c.chunk, c1 = c0 + c.chunk;
            if (c1 > c.n_perm) c1 = c.n_perm;
            Fe left = ldg(pr.pz_cos + (size_t)s * en + r_next);
            Fe right = ldg(pr.pz_cos + (size_t)s * en + idx);
            for (uint32_t col = c0; col < c1; col++) {
                const zg_query q = c.perm_cols[col];
                const Fe* base = q.kind == ZG_FIXED ? pr.cols.fixed : q.kind == ZG_ADVICE ? pr.cols.advice : pr.cols.instance;
                Fe v = ldg(base + ((size_t)q.column << a.cols.log_size) + idx);
                Fe sg = ldg(a.sigma_cos + (size_t)col * en + idx);
                left = Fr::mul(left, Fr::add(Fr::add(Fr::mul(beta, sg), v), gamma));
                right = Fr::mul(right, Fr::add(Fr::add(v, current_delta), gamma));
                current_delta = Fr::mul(current_delta, a.delta);
            }
            value = Fr::add(Fr::mul(value, y), Fr::mul(Fr::sub(left, right), lactive));
        }
    }
    for (uint32_t l = 0; l < c.n_lookups; l++) {
        const DLookup* lk = c.lookups + l;
        Fe ai = fe_zero(), ti = fe_zero();
        for (uint32_t e = 0; e < lk->width; e++) {
            ai = Fr::add(Fr::mul(ai, theta), eval_poly(c, pr.cols, lk->inputs[e], idx));
            ti = Fr::add(Fr::mul(ti, theta), eval_poly(c, pr.cols, lk->tables[e], idx));
        }
        const Fe* zc = pr.lz_cos + (size_t)l * en;
        const Fe* ap = pr.pin_cos + (size_t)l * a.perm_stride;
        const Fe* sp = pr.ptab_cos + (size_t)l * a.perm_stride;
        const Fe z = ldg(zc + idx), apv = ldg(ap + idx), spv = ldg(sp + idx);
        value = Fr::add(Fr::mul(value, y), Fr::mul(Fr::sub(Fr::one(), z), l0));
        value = Fr::add(Fr::mul(value, y), Fr::mul(Fr::sub(Fr::sqr(z), z), llast));
        Fe lft = Fr::mul(Fr::mul(Fr::add(apv, beta), Fr::add(spv, gamma)), ldg(zc + r_next));
        Fe rgt = Fr::mul(Fr::mul(Fr::add(ai, beta), Fr::add(ti, gamma)), z);
        value = Fr::add(Fr::mul(value, y), Fr::mul(Fr::sub(lft, rgt), lactive));
        Fe ams = Fr::sub(apv, spv);
        value = Fr::add(Fr::mul(value, y), Fr::mul(ams, l0));
        value = Fr::add(Fr::mul(value, y), Fr::mul(Fr::mul(ams, Fr::sub(apv, ldg(ap + r_prev))), lactive));
    }
    // divide_by_vanishing_poly
    value = Fr::mul(value, ldg(a.t_eval + (idx & a.t_mask)));
    stg(pr.h + idx, value);
}


// ---- the same evaluation on nine 29-bit limbs (field9.h).  Every operand is in the 2^261 Montgomery form,
// so Fr9::mul is the field product; sums are limb-wise and lazily normalised under one rule: both
// operands of a product have limb magnitudes < 2^29, except that ONE of them may be the sum of two
// normalised values (< 2^30).  A normalised value plus a product (the Horner step) therefore feeds the
// next product directly.
__device__ __forceinline__ F9 ld9(const Fe* p) { return f9_unpack(ldg(p)); }

__device__ __forceinline__ F9 eval_poly9(const DevCircuit& c, const DMono* monos, const Cols& cols, zg_poly p, uint32_t row) {
    const uint32_t mask = (1u << cols.log_size) - 1u;
    auto cell = [&](uint32_t qi) {
        const zg_query q = c.queries[qi];
        const Fe* base = q.kind == ZG_FIXED ? cols.fixed : q.kind == ZG_ADVICE ? cols.advice : cols.instance;
        const uint32_t idx = (row + (uint32_t)(q.rotation * cols.rot_scale)) & mask;
        return ld9(base + ((size_t)q.column << cols.log_size) + idx);
    };
    F9 acc;
#pragma unroll
    for (int i = 0; i < 9; i++) acc.l[i] = 0;
    uint32_t pending = 0;  // terms added since the last carry normalisation
    auto add = [&](const F9& t, bool neg) {  // acc +/- t  (a coefficient -1 is a subtraction, not a product)
        acc = neg ? f9_sub(acc, t) : f9_add(acc, t);
        if (++pending == 2) {  // (wave-uniform: the monomial list is)
            acc = f9_norm(acc);
            pending = 0;
        }
    };
    // A monomial is +/- L * R with R its last factor; the final products of two consecutive monomials share one
    // Montgomery reduction (Fr9::mul2).
    F9 hl, hr;          // a monomial waiting for its partner
    bool held = false, hneg = false;
    for (uint32_t m = p.first; m < p.first + p.count; m++) {
        const DMono* mo = monos + m;
        const uint32_t nf = mo->n_factors;
        const bool unit = mo->coeff_is_one != 0 && nf > 0;
        const bool neg = unit && mo->coeff_is_one == 2;
        const uint32_t terms = nf + (unit ? 0u : 1u);  // operands of the product
        if (terms == 1) {  // a bare cell or a bare constant
            add(unit ? cell(mo->factors[0]) : f9_unpack(mo->coeff), neg);
            continue;
        }
        uint32_t f = 0;
        F9 l = unit ? cell(mo->factors[f++]) : f9_unpack(mo->coeff);
        for (; f + 1 < nf; f++) l = Fr9::mul(l, cell(mo->factors[f]));
        const F9 r = cell(mo->factors[nf - 1]);
        if (held) {
            if (hneg == neg) add(Fr9::mul2<false>(hl, hr, l, r), neg);
            else add(Fr9::mul2<true>(hl, hr, l, r), hneg);  // +(hl hr - l r) or -(hl hr - l r)
            held = false;
        } else {
            hl = l;
            hr = r;
            hneg = neg;
            held = true;
        }
    }
    if (held) add(Fr9::mul(hl, hr), hneg);
    return f9_norm(acc);
}

// U(x) = x * (c_1 + x * (c_2 + ... )) for wave-uniform coefficients c_k = coef[k-1], k = 1..count
__device__ __forceinline__ F9 horner9(const F9& x, const Fe* coef, uint32_t count) {
    F9 u = f9_unpack(coef[count - 1]);
    for (uint32_t k = count - 1; k >= 1; k--) u = f9_add(Fr9::mul(u, x), f9_unpack(coef[k - 1]));
    return Fr9::mul(u, x);
}

__global__ __launch_bounds__(256) void gate_factor9_kernel(const Fe* __restrict__ col, uint32_t rot_off, uint32_t en,
                                                           const Fe* __restrict__ coef, uint32_t count, Fe* __restrict__ out) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= en) return;
    stg(out + idx, f9_reduce_pack<Fr9Params>(horner9(ld9(col + ((idx + rot_off) & (en - 1))), coef, count)));
}

int poly_gate_factor(zg_ctx* ctx, const Fe* col, uint32_t rot_off, uint32_t en, const Fe* coef, uint32_t count, Fe* out) {
    ZG_REQUIRE(count >= 1 && (en & (en - 1)) == 0, ZG_ERR_INVALID_ARG, "poly_gate_factor: %u coefficients, %u points", count, en);
    ZG_LAUNCH(ctx, "gate_factor", 2.0 * en * 32.0, gate_factor9_kernel, dim3((en + 255) / 256), dim3(256), 0, col, rot_off, en, coef,
              count, out);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// GROUPED: the terms after the gates are not folded one by one (value * y + term * l, two products each) but
// weighted and summed per l-polynomial,
//     h_num = G y^T + l0 * sum_i y^e_i a_i + llast * sum_i y^e_i b_i + lactive * sum_i y^e_i c_i,
// with the powers of y from pc->eh_ypow: one product per term (two terms per Montgomery reduction) + four at the
// end -- the same field element, since the arithmetic is exact.
template <bool GROUPED>
__global__ __launch_bounds__(256) void evaluate_h9_kernel(EvalHArgs a, uint32_t en) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= en) return;
    const EvalHProof pr = evalh_proof(a, blockIdx.y);
    const ProofConst* pc = pr.pc;  // (scalars are re-read where they are used: workgroup-uniform scalar loads)
    const DevCircuit& c = a.c;
    const uint32_t mask = en - 1;
    const uint32_t rs = (uint32_t)a.cols.rot_scale;
    const uint32_t r_next = (idx + rs) & mask;
    const uint32_t r_prev = (idx - rs) & mask;
    const uint32_t r_last = (idx + (uint32_t)(a.last_rot * (int32_t)rs)) & mask;
    // (constants and l-polynomial values are re-materialised at each use: a nine-limb value is 9 VGPRs, and
    // this kernel's occupancy is set by its register count)
    // value <- value * y + term   (term normalised; the sum has limbs < 2^30 and goes into the next product)
    auto fold = [&](const F9& value, const F9& term) { return f9_add(Fr9::mul(value, f9_unpack(pc->eh_y)), term); };
    // value <- value * y + u * v with ONE Montgomery reduction (value normalised, u and v with limbs < 2^29)
    auto fold2 = [&](const F9& value, const F9& u, const F9& v) { return Fr9::mul2<false>(value, f9_unpack(pc->eh_y), u, v); };
    F9 value;
#pragma unroll
    for (int i = 0; i < 9; i++) value.l[i] = 0;
    auto cell = [&](uint32_t qi) {
        const zg_query q = c.queries[qi];
        const Fe* base = q.kind == ZG_FIXED ? pr.cols.fixed : q.kind == ZG_ADVICE ? pr.cols.advice : pr.cols.instance;
        return ld9(base + ((size_t)q.column << a.cols.log_size) + ((idx + (uint32_t)(q.rotation * a.cols.rot_scale)) & mask));
    };
    for (uint32_t g = 0; g < c.n_gates; g++) {
        const F9 inner = eval_poly9(c, a.monos_hat, pr.cols, a.gates_hat[g], idx);
        const uint32_t common = a.gate_common[g];  // (wave-uniform)
        if (common != 0xffffffffu) {
            const uint32_t slab = a.gate_slab[g];
            const zg_poly uni = a.gate_uni[g];
            F9 u;
            if (slab != 0xffffffffu) {
                u = ld9(a.gate_slabs + (size_t)slab * en + idx);  // U(fixed cell), tabulated with the proving key
            } else if (uni.count) {
                u = horner9(cell(common), a.uni_coef + uni.first, uni.count);
            } else {
                u = cell(common);
            }
            value = fold2(f9_norm(value), u, inner);  // value * y + U(common) * inner, one reduction
        } else {
            value = fold(value, inner);
        }
    }
    value = f9_norm(value);  // from here on every term is a product: fold2 keeps value normalised

    auto l0 = [&]() { return ld9(a.l0 + idx); };
    auto llast = [&]() { return ld9(a.llast + idx); };
    auto lactive = [&]() { return ld9(a.lactive + idx); };
    auto one = [&]() { return Fr9Params::one(); };
    // theta-compression of a lookup's input and table expressions (the first expression enters as it is)
    auto compress = [&](const DLookup* lk, F9& ai, F9& ti) {
        ai = eval_poly9(c, a.monos_hat, pr.cols, lk->inputs[0], idx);
        ti = eval_poly9(c, a.monos_hat, pr.cols, lk->tables[0], idx);
        for (uint32_t e = 1; e < lk->width; e++) {
            ai = f9_add(Fr9::mul(ai, f9_unpack(pc->eh_theta)), eval_poly9(c, a.monos_hat, pr.cols, lk->inputs[e], idx));
            ti = f9_add(Fr9::mul(ti, f9_unpack(pc->eh_theta)), eval_poly9(c, a.monos_hat, pr.cols, lk->tables[e], idx));
        }
    };
    if (GROUPED) {
        // term i of the sequence (upstream's order) carries y^(T - 1 - i); `w` walks down from T - 1
        uint32_t w = a.n_terms;
        auto yw = [&](uint32_t e) { return f9_unpack(pc->eh_ypow[e]); };
        F9 s0, sl, sa;  // the sums that l0, llast and lactive multiply (normalised after every addition)
#pragma unroll
        for (int i = 0; i < 9; i++) s0.l[i] = sl.l[i] = sa.l[i] = 0;
        auto acc1 = [&](F9& s, uint32_t e, const F9& t) { s = f9_norm(f9_add(s, Fr9::mul(yw(e), t))); };
        auto acc2 = [&](F9& s, uint32_t e, const F9& t, uint32_t e2, const F9& t2) {
            s = f9_norm(f9_add(s, Fr9::mul2<false>(yw(e), t, yw(e2), t2)));
        };
        if (c.n_sets > 0) {
            const F9 zf = ld9(pr.pz_cos + idx);
            const F9 zl = ld9(pr.pz_cos + (size_t)(c.n_sets - 1) * en + idx);
            acc1(s0, w - 1, f9_sub(one(), zf));
            acc1(sl, w - 2, f9_sub(Fr9::sqr(zl), zl));
            w -= 2;
            for (uint32_t s = 1; s < c.n_sets; s++) {
                const F9 t = f9_sub(ld9(pr.pz_cos + (size_t)s * en + idx), ld9(pr.pz_cos + (size_t)(s - 1) * en + r_last));
                acc1(s0, --w, t);
            }
            F9 current_delta = Fr9::mul(f9_unpack(pc->eh_delta_start[a.zpow - 1]), ld9(a.ext_tw + idx));
            const F9 delta = f9_unpack(a.delta);
            for (uint32_t s = 0; s < c.n_sets; s++) {
                uint32_t c0 = s * c.chunk, c1 = c0 + c.chunk;
                if (c1 > c.n_perm) c1 = c.n_perm;
                F9 left = ld9(pr.pz_cos + (size_t)s * en + r_next);
                F9 right = ld9(pr.pz_cos + (size_t)s * en + idx);
                for (uint32_t col = c0; col < c1; col++) {
                    const zg_query q = c.perm_cols[col];
                    const Fe* base = q.kind == ZG_FIXED ? pr.cols.fixed : q.kind == ZG_ADVICE ? pr.cols.advice : pr.cols.instance;
                    const F9 v = ld9(base + ((size_t)q.column << a.cols.log_size) + idx);
                    const F9 sg = ld9(a.sigma_cos + (size_t)col * en + idx);
                    const F9 fl = f9_norm(f9_add(f9_add(Fr9::mul(f9_unpack(pc->eh_beta), sg), v), f9_unpack(pc->eh_gamma)));
                    const F9 fr = f9_norm(f9_add(f9_add(v, current_delta), f9_unpack(pc->eh_gamma)));
                    current_delta = Fr9::mul(current_delta, delta);
                    if (col + 1 < c1) {
                        left = Fr9::mul(left, fl);
                        right = Fr9::mul(right, fr);
                    } else {
                        left = Fr9::mul2<true>(left, fl, right, fr);
                    }
                }
                acc1(sa, --w, left);
            }
        }
        for (uint32_t l = 0; l < c.n_lookups; l++) {
            const DLookup* lk = c.lookups + l;
            F9 ai, ti;
            compress(lk, ai, ti);
            const Fe* zc = pr.lz_cos + (size_t)l * en;
            const Fe* ap = pr.pin_cos + (size_t)l * a.perm_stride;
            const Fe* sp = pr.ptab_cos + (size_t)l * a.perm_stride;
            const F9 z = ld9(zc + idx), apv = ld9(ap + idx), spv = ld9(sp + idx);
            const F9 lft = Fr9::mul(f9_add(apv, f9_unpack(pc->eh_beta)), f9_norm(f9_add(spv, f9_unpack(pc->eh_gamma))));
            const F9 rgt = Fr9::mul(f9_norm(f9_add(ai, f9_unpack(pc->eh_beta))), f9_norm(f9_add(ti, f9_unpack(pc->eh_gamma))));
            const F9 ams = f9_sub(apv, spv);
            // five terms: (1 - z) l0, (z^2 - z) llast, (lft z(wX) - rgt z) lactive, (a' - s') l0, (a' - s')(a' - a'(w^-1 X)) lactive
            acc2(s0, w - 1, f9_sub(one(), z), w - 4, ams);
            acc1(sl, w - 2, f9_sub(Fr9::sqr(z), z));
            acc2(sa, w - 3, Fr9::mul2<true>(lft, ld9(zc + r_next), rgt, z), w - 5, Fr9::mul(ams, f9_sub(apv, ld9(ap + r_prev))));
            w -= 5;
        }
        // h_num = value y^T + l0 s0 + llast sl + lactive sa
        value = f9_norm(f9_add(Fr9::mul2<false>(value, yw(a.n_terms), s0, l0()), Fr9::mul2<false>(sl, llast(), sa, lactive())));
    } else {
    if (c.n_sets > 0) {
        const F9 zf = ld9(pr.pz_cos + idx);
        const F9 zl = ld9(pr.pz_cos + (size_t)(c.n_sets - 1) * en + idx);
        value = fold2(value, f9_sub(one(), zf), l0());
        value = fold2(value, f9_sub(Fr9::sqr(zl), zl), llast());
        for (uint32_t s = 1; s < c.n_sets; s++) {
            const F9 t = f9_sub(ld9(pr.pz_cos + (size_t)s * en + idx), ld9(pr.pz_cos + (size_t)(s - 1) * en + r_last));
            value = fold2(value, t, l0());
        }
        F9 current_delta = Fr9::mul(f9_unpack(pc->eh_delta_start[a.zpow - 1]), ld9(a.ext_tw + idx));
        const F9 delta = f9_unpack(a.delta);
        for (uint32_t s = 0; s < c.n_sets; s++) {
            uint32_t c0 = s * c.chunk, c1 = c0 + c.chunk;
            if (c1 > c.n_perm) c1 = c.n_perm;
            F9 left = ld9(pr.pz_cos + (size_t)s * en + r_next);
            F9 right = ld9(pr.pz_cos + (size_t)s * en + idx);
            for (uint32_t col = c0; col < c1; col++) {
                const zg_query q = c.perm_cols[col];
                const Fe* base = q.kind == ZG_FIXED ? pr.cols.fixed : q.kind == ZG_ADVICE ? pr.cols.advice : pr.cols.instance;
                const F9 v = ld9(base + ((size_t)q.column << a.cols.log_size) + idx);
                const F9 sg = ld9(a.sigma_cos + (size_t)col * en + idx);
                // three-term sums: normalise before they enter a product
                const F9 fl = f9_norm(f9_add(f9_add(Fr9::mul(f9_unpack(pc->eh_beta), sg), v), f9_unpack(pc->eh_gamma)));
                const F9 fr = f9_norm(f9_add(f9_add(v, current_delta), f9_unpack(pc->eh_gamma)));
                current_delta = Fr9::mul(current_delta, delta);
                if (col + 1 < c1) {
                    left = Fr9::mul(left, fl);
                    right = Fr9::mul(right, fr);
                } else {  // last column of the set: left * fl - right * fr under one reduction
                    left = Fr9::mul2<true>(left, fl, right, fr);
                }
            }
            value = fold2(value, left, lactive());
        }
    }
    for (uint32_t l = 0; l < c.n_lookups; l++) {
        const DLookup* lk = c.lookups + l;
        F9 ai, ti;
        compress(lk, ai, ti);
        const Fe* zc = pr.lz_cos + (size_t)l * en;
        const Fe* ap = pr.pin_cos + (size_t)l * a.perm_stride;
        const Fe* sp = pr.ptab_cos + (size_t)l * a.perm_stride;
        const F9 z = ld9(zc + idx), apv = ld9(ap + idx), spv = ld9(sp + idx);
        value = fold2(value, f9_sub(one(), z), l0());
        value = fold2(value, f9_sub(Fr9::sqr(z), z), llast());
        // (x + beta)(y + gamma): one factor may stay a two-term sum, the other is normalised
        const F9 lft = Fr9::mul(f9_add(apv, f9_unpack(pc->eh_beta)), f9_norm(f9_add(spv, f9_unpack(pc->eh_gamma))));
        const F9 rgt = Fr9::mul(f9_norm(f9_add(ai, f9_unpack(pc->eh_beta))), f9_norm(f9_add(ti, f9_unpack(pc->eh_gamma))));
        value = fold2(value, Fr9::mul2<true>(lft, ld9(zc + r_next), rgt, z), lactive());
        const F9 ams = f9_sub(apv, spv);
        value = fold2(value, ams, l0());
        value = fold2(value, Fr9::mul(ams, f9_sub(apv, ld9(ap + r_prev))), lactive());
    }
    }
    // divide_by_vanishing_poly, then back to the canonical packed form (still x * 2^261)
    value = Fr9::mul(value, ld9(a.t_eval + (idx & a.t_mask)));
    stg(pr.h + idx, f9_reduce_pack<Fr9Params>(value));
}

int poly_evaluate_h(zg_ctx* ctx, const EvalHArgs& a, uint32_t en, uint32_t nb, uint32_t n_columns, double unit_share) {
    if (!nb) return ZG_OK;
    ZG_REQUIRE(a.pc != nullptr && (a.zpow == 1 || a.zpow == 2), ZG_ERR_INVALID_ARG, "evaluate_h: per-proof scalars missing");
    // algorithmic bytes: every input coset read once + h written, as SURVEY.md 8d counts them -- `n_columns` advice,
    // instance and fixed columns, 3 l-polynomials, sigma, the permutation products, 3 polynomials per lookup, h: ONE count
    // (round 3 charged 26 arrays here while bench.py counted 49).  own = over the `en` rows of this launch; unit = over the
    // 2^ext_k rows of EvaluationDomain's extended domain, of which this launch stands for `unit_share`.
    const DevCircuit& c = a.c;
    const double arrays = (double)n_columns + 3.0 + c.n_perm + c.n_sets + 3.0 * c.n_lookups + 1.0;
    const double unit = nb * arrays * unit_share * 32.0;
    ZG_REQUIRE(!a.hat || a.monos_hat != nullptr || (c.n_gates == 0 && c.n_lookups == 0), ZG_ERR_INVALID_ARG,
               "evaluate_h: the 2^261-form monomial table is missing");
    ZG_REQUIRE(!a.hat || c.n_gates == 0 || (a.gates_hat != nullptr && a.gate_common != nullptr && a.gate_uni != nullptr && a.uni_coef != nullptr && a.gate_slab != nullptr && a.gate_slabs != nullptr), ZG_ERR_INVALID_ARG,
               "evaluate_h: the factored gate table is missing");
    if (a.hat && a.n_terms)
        ZG_LAUNCH_U(ctx, "evaluate_h", nb * arrays * en * 32.0, unit, evaluate_h9_kernel<true>, dim3((en + 255) / 256, nb), dim3(256), 0, a, en);
    else if (a.hat)
        ZG_LAUNCH_U(ctx, "evaluate_h", nb * arrays * en * 32.0, unit, evaluate_h9_kernel<false>, dim3((en + 255) / 256, nb), dim3(256), 0, a, en);
    else
        ZG_LAUNCH_U(ctx, "evaluate_h", nb * arrays * en * 32.0, unit, evaluate_h_kernel, dim3((en + 255) / 256, nb), dim3(256), 0, a, en);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// ------------------------------------------------------------------ eval_polynomial
__device__ __forceinline__ const Fe* poly_of(const PolySet& ps, uint32_t ix, uint32_t b) {
    return ix < ps.nsh ? ps.sh + (size_t)ix * ps.n : ps.pp + (size_t)b * ps.pp_bs + (size_t)(ix - ps.nsh) * ps.n;
}

// pow[b][s][i] = x^i for x = pc[b].points[s]: lane computes x^(i0) by square-and-multiply, then a strip of CH products
// (16 in the throughput form; 4 for a lone proof, which waits for the dependent chain: 28 + 4 products instead of 28 + 16)
template <uint32_t CH>
__global__ __launch_bounds__(256) void powers_kernel(const ProofConst* __restrict__ pc, uint32_t n, Fe* __restrict__ pw,
                                                     size_t pw_bs) {
    uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * CH;
    if (i0 >= n) return;
    const uint32_t sl = blockIdx.y, b = blockIdx.z;
    Fe x = pc[b].points[sl];
    Fe cur = Fr::pow_u64(x, i0);
    Fe* out = pw + (size_t)b * pw_bs + (size_t)sl * n;
    for (uint32_t j = 0; j < CH && i0 + j < n; j++) {
        stg(out + i0 + j, cur);
        cur = Fr::mul(cur, x);
    }
}

int poly_powers(zg_ctx* ctx, const ProofConst* pc, uint32_t nb, uint32_t npoints, uint32_t n, Fe* d_pow, size_t pw_bs) {
    if (!npoints || !nb) return ZG_OK;
    ZG_REQUIRE(npoints <= PC_MAX_POINTS, ZG_ERR_UNSUPPORTED, "poly_powers: %u opening points (max %u)", npoints, PC_MAX_POINTS);
    if (ctx->msm_pair)
        ZG_LAUNCH(ctx, "powers", (double)nb * npoints * n * 32, powers_kernel<4>, dim3((n + 256 * 4 - 1) / (256 * 4), npoints, nb),
                  dim3(256), 0, pc, n, d_pow, pw_bs);
    else
        ZG_LAUNCH(ctx, "powers", (double)nb * npoints * n * 32, powers_kernel<16>, dim3((n + 256 * 16 - 1) / (256 * 16), npoints, nb),
                  dim3(256), 0, pc, n, d_pow, pw_bs);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// out[b][j] = <poly_j of proof b, pow_{point_j} of proof b> : one 1024-lane workgroup per (poly, point) pair (a few
// dozen pairs per proof: the wide workgroup is what keeps a lone proof from waiting on 64 products per lane)
constexpr uint32_t DOT_NT = 1024;
__global__ __launch_bounds__(DOT_NT) void dot_kernel(PolySet ps, uint32_t n, const uint32_t* __restrict__ poly_idx,
                                                     const uint32_t* __restrict__ point_idx, const Fe* __restrict__ pw,
                                                     size_t pw_bs, Fe* __restrict__ out, size_t out_bs) {
    __shared__ Fe sh[DOT_NT];
    const uint32_t j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const Fe* p = poly_of(ps, poly_idx[j], b);
    const Fe* w = pw + (size_t)b * pw_bs + (size_t)point_idx[j] * n;
    Fe acc = fe_zero();
    for (uint32_t i = tid; i < n; i += DOT_NT) acc = Fr::add(acc, Fr::mul(ldg(p + i), ldg(w + i)));
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t off = DOT_NT / 2; off > 0; off >>= 1) {
        if (tid < off) sh[tid] = Fr::add(sh[tid], sh[tid + off]);
        __syncthreads();
    }
    if (tid == 0) stg(out + (size_t)b * out_bs + j, sh[0]);
}

// The same dot product with ONE Montgomery reduction per lane (Dot9, field9.h): a term is its 81 multiply-adds, the
// columns are carried every third term, and the loads of three terms are in flight at once (a lone proof has one wave per
// SIMD here and waits on every dependent load).  Both operands are x * 2^256, so a lane's reduced sum is 2^251 * S; the
// factor 2^5 is put back by one product per (polynomial, point) pair at the end.  Same field element as dot_kernel.
// Workgroups of 1024 lanes for a lone proof (the latency form: a few dozen pairs have to fill the chip), of 256 in the
// throughput form, where a 1024-lane workgroup at this register count waits for a whole CU to drain between the other
// provers' kernels.
template <uint32_t DOT_NT>
__global__ __launch_bounds__(DOT_NT) void dot9_kernel(PolySet ps, uint32_t n, const uint32_t* __restrict__ poly_idx,
                                                      const uint32_t* __restrict__ point_idx, const Fe* __restrict__ pw,
                                                      size_t pw_bs, Fe* __restrict__ out, size_t out_bs) {
    __shared__ Fe sh[DOT_NT];
    const uint32_t j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const Fe* p = poly_of(ps, poly_idx[j], b);
    const Fe* w = pw + (size_t)b * pw_bs + (size_t)point_idx[j] * n;
    Dot9<Fr9Params> acc;
    acc.zero();
    uint32_t i = tid;
    for (; i + 2 * DOT_NT < n; i += 3 * DOT_NT) {  // (three terms: what 128 VGPRs hold next to the 17 columns)
        Fe pv[3], wv[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            pv[t] = ldg(p + i + t * DOT_NT);
            wv[t] = ldg(w + i + t * DOT_NT);
        }
#pragma unroll
        for (int t = 0; t < 3; t++) acc.mac(f9_unpack(pv[t]), f9_unpack(wv[t]));
        acc.carry();
    }
    for (; i < n; i += DOT_NT) {
        acc.mac(ld9(p + i), ld9(w + i));
        acc.carry();
    }
    sh[tid] = f9_reduce_pack<Fr9Params>(acc.reduce());
    __syncthreads();
    for (uint32_t off = DOT_NT / 2; off > 0; off >>= 1) {
        if (tid < off) sh[tid] = Fr::add(sh[tid], sh[tid + off]);
        __syncthreads();
    }
    if (tid == 0) stg(out + (size_t)b * out_bs + j, Fr::mul(sh[0], Fr9Params::c261_fe()));
}

int poly_dot(zg_ctx* ctx, const PolySet& polys, uint32_t nb, uint32_t n, const uint32_t* d_poly_idx,
             const uint32_t* d_point_idx, const Fe* d_pow, size_t pw_bs, uint32_t count, Fe* d_out, size_t out_bs,
             uint32_t distinct_polys, uint32_t distinct_points) {
    if (!count || !nb) return ZG_OK;
    // algorithmic bytes: a polynomial opened at several points is READ ONCE, and so is each point's power table (round 3
    // charged count * n * 64: more than the kernel moves -- its figure came out above the HBM peak)
    const double dot_bytes = (double)nb * ((double)(distinct_polys ? distinct_polys : count) + (distinct_points ? distinct_points : count)) * n * 32.0;
    // (a lane sums n / 256 terms at most: 2^12 at n = 2^20, inside Dot9's bound of 2^13 terms)
    if (knob(K_LAZY_DOT) != 0 && n <= (1u << 20) && ctx->msm_pair)
        ZG_LAUNCH(ctx, "eval_dot", dot_bytes, dot9_kernel<1024>, dim3(count, nb), dim3(1024), 0, polys, n, d_poly_idx,
                  d_point_idx, d_pow, pw_bs, d_out, out_bs);
    else if (knob(K_LAZY_DOT) != 0 && n <= (1u << 20))
        ZG_LAUNCH(ctx, "eval_dot", dot_bytes, dot9_kernel<256>, dim3(count, nb), dim3(256), 0, polys, n, d_poly_idx,
                  d_point_idx, d_pow, pw_bs, d_out, out_bs);
    else
    ZG_LAUNCH(ctx, "eval_dot", dot_bytes, dot_kernel, dim3(count, nb), dim3(DOT_NT), 0, polys, n, d_poly_idx,
              d_point_idx, d_pow, pw_bs, d_out, out_bs);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// ------------------------------------------------------------------ GWC
// out[b][i] = (...((p_0[i]) x + p_1[i]) x + ...) + p_{m-1}[i] with x = pc[b].xn (vanishing::evaluate's h(X))
__global__ __launch_bounds__(256) void horner_combine_kernel(PolySet ps, const ProofConst* __restrict__ pc,
                                                             const uint32_t* __restrict__ list, uint32_t count,
                                                             Fe* __restrict__ out, size_t out_bs, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = blockIdx.y;
    if (i >= n) return;
    const Fe v = pc[b].xn;
    Fe acc = fe_zero();
    for (uint32_t j = 0; j < count; j++) acc = Fr::add(Fr::mul(acc, v), ldg(poly_of(ps, list[j], b) + i));
    stg(out + (size_t)b * out_bs + i, acc);
}

// The same combinations as ONE dot product per coefficient (Dot9, field9.h): sum_j p_j[i] * v^(count - 1 - j) against a
// table of the powers of v in the 2^261 form -- made once per workgroup, lane e takes v^e -- so that a term costs 81
// multiply-adds and the Montgomery reduction is paid once per coefficient instead of once per term; the loads of three
// terms are in flight at once.  Horner's value exactly (the arithmetic is exact), SETS: v = pc[b].v and pc[b].subs[s]
// comes off the constant term (the multiopen argument's lists), else v = pc[b].xn (h(X) from its pieces).
struct HornerSets {
    uint32_t count[HC_MAX_SETS];
};
constexpr uint32_t HC9_MAX = 128;  // longest list the power table holds (longer ones take the Horner kernels)
template <bool SETS>
__global__ __launch_bounds__(256) void combine9_kernel(PolySet ps, const ProofConst* __restrict__ pc,
                                                       const uint32_t* __restrict__ lists, uint32_t list_stride, HornerSets sets,
                                                       Fe* __restrict__ out, size_t out_stride, size_t out_bs, uint32_t n) {
    __shared__ F9 vp[HC9_MAX];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y, b = blockIdx.z;
    const uint32_t count = sets.count[s];
    if (threadIdx.x < count) {
        const Fe v = SETS ? pc[b].v : pc[b].xn;
        vp[threadIdx.x] = f9_unpack(Fr::mul(Fr::pow_u64(v, threadIdx.x), Fr9Params::c261_fe()));
    }
    __syncthreads();
    if (i >= n) return;
    const uint32_t* list = lists + (size_t)s * list_stride;
    Dot9<Fr9Params> acc;
    acc.zero();
    uint32_t j = 0;
    for (; j + 3 <= count; j += 3) {
        const Fe p0 = ldg(poly_of(ps, list[j], b) + i), p1 = ldg(poly_of(ps, list[j + 1], b) + i),
                 p2 = ldg(poly_of(ps, list[j + 2], b) + i);
        acc.mac(f9_unpack(p0), vp[count - 1 - j]);
        acc.mac(f9_unpack(p1), vp[count - 2 - j]);
        acc.mac(f9_unpack(p2), vp[count - 3 - j]);
        acc.carry();
    }
    for (; j < count; j++) {
        acc.mac(ld9(poly_of(ps, list[j], b) + i), vp[count - 1 - j]);
        acc.carry();
    }
    Fe r = f9_reduce_pack<Fr9Params>(acc.reduce());
    if (SETS && i == 0) r = Fr::sub(r, pc[b].subs[s]);
    stg(out + (size_t)b * out_bs + (size_t)s * out_stride + i, r);
}

int poly_horner_combine_xn(zg_ctx* ctx, const PolySet& polys, const ProofConst* pc, uint32_t nb, const uint32_t* d_list,
                           uint32_t count, Fe* out, size_t out_bs, uint32_t n) {
    if (!nb) return ZG_OK;
    if (knob(K_LAZY_DOT) != 0 && count <= HC9_MAX) {
        HornerSets sets;
        memset(&sets, 0, sizeof(sets));
        sets.count[0] = count;
        ZG_LAUNCH(ctx, "horner_combine", (double)nb * (count + 1) * n * 32, combine9_kernel<false>, dim3((n + 255) / 256, 1, nb),
                  dim3(256), 0, polys, pc, d_list, 0u, sets, out, (size_t)0, out_bs, n);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    ZG_LAUNCH(ctx, "horner_combine", (double)nb * (count + 1) * n * 32, horner_combine_kernel, dim3((n + 255) / 256, nb), dim3(256),
              0, polys, pc, d_list, count, out, out_bs, n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// the same in pc[b].v for up to HC_MAX_SETS lists at once (GWC: one list per opening point): set s = blockIdx.y of
// proof b = blockIdx.z reads lists + s * list_stride and writes out + b * out_bs + s * out_stride, with
// pc[b].subs[s] taken off the constant term
__global__ __launch_bounds__(256) void horner_combine_sets_kernel(PolySet ps, const ProofConst* __restrict__ pc,
                                                                  const uint32_t* __restrict__ lists, uint32_t list_stride,
                                                                  HornerSets sets, Fe* __restrict__ out, size_t out_stride,
                                                                  size_t out_bs, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y, b = blockIdx.z;
    if (i >= n) return;
    const uint32_t* list = lists + (size_t)s * list_stride;
    const uint32_t count = sets.count[s];
    const Fe v = pc[b].v;
    Fe acc = fe_zero();
    for (uint32_t j = 0; j < count; j++) acc = Fr::add(Fr::mul(acc, v), ldg(poly_of(ps, list[j], b) + i));
    if (i == 0) acc = Fr::sub(acc, pc[b].subs[s]);
    stg(out + (size_t)b * out_bs + (size_t)s * out_stride + i, acc);
}

int poly_horner_combine_sets(zg_ctx* ctx, const PolySet& polys, const ProofConst* pc, uint32_t nb, const uint32_t* d_lists,
                             uint32_t list_stride, const uint32_t* counts, uint32_t nsets, Fe* out, size_t out_stride,
                             size_t out_bs, uint32_t n) {
    if (!nsets || !nb) return ZG_OK;
    ZG_REQUIRE(nsets <= HC_MAX_SETS, ZG_ERR_UNSUPPORTED, "poly_horner_combine_sets: %u sets", nsets);
    HornerSets sets;
    memset(&sets, 0, sizeof(sets));
    double total = 0;
    for (uint32_t s = 0; s < nsets; s++) {
        sets.count[s] = counts[s];
        total += counts[s] + 1;
    }
    bool lazy = knob(K_LAZY_DOT) != 0;
    for (uint32_t s = 0; s < nsets; s++) lazy = lazy && counts[s] <= HC9_MAX;
    if (lazy)
        ZG_LAUNCH(ctx, "horner_combine", nb * total * n * 32, combine9_kernel<true>, dim3((n + 255) / 256, nsets, nb), dim3(256), 0,
                  polys, pc, d_lists, list_stride, sets, out, out_stride, out_bs, n);
    else
    ZG_LAUNCH(ctx, "horner_combine", nb * total * n * 32, horner_combine_sets_kernel, dim3((n + 255) / 256, nsets, nb), dim3(256), 0,
              polys, pc, d_lists, list_stride, sets, out, out_stride, out_bs, n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// kate_division: q_i = a_{i+1} + z q_{i+1} (i = n-2 .. 0, q_{n-1} = 0) -- upstream's one-core recurrence.  Strip form,
// one workgroup per polynomial: lane l owns the strip [l S, (l + 1) S) of coefficients, S = n / lanes.
//   pass A   the strip's own contribution at its first element, H_l = sum_{i in strip} a_{i+1} z^(i - l S) (Horner,
//            S products);
//   scan     the true values at the strip starts obey the same recurrence with ratio z^S over the lanes,
//            Q_l = H_l + z^S Q_{l+1}: a weighted Hillis-Steele suffix scan through LDS (two products per step);
//   pass B   the recurrence again, now started from the true carry Q_{l+1}: every q_i of the strip, S products.
// About 3.3 products per coefficient where the block-scan form (three launches) spent ~30: 16 in its scan steps, up
// to 16 in the z^(256 - tid) of its apply pass.  Batched over (proof, point set): set s of proof b divides by
// X - pc[b].points[slot[s]].
constexpr uint32_t KD_LANES = 1024;
struct KdSlots {
    uint32_t slot[HC_MAX_SETS];
};

__global__ __launch_bounds__(KD_LANES) void kd_strip_kernel(const ProofConst* __restrict__ pc, KdSlots slots,
                                                            const Fe* __restrict__ a, size_t a_stride, size_t a_bs,
                                                            Fe* __restrict__ q, size_t q_stride, size_t q_bs, uint32_t n,
                                                            uint32_t lanes, uint32_t strip) {
    __shared__ Fe sh[KD_LANES];
    const uint32_t tid = threadIdx.x, s = blockIdx.x, b = blockIdx.y;
    const Fe* ap = a + (size_t)b * a_bs + (size_t)s * a_stride;
    Fe* qp = q + (size_t)b * q_bs + (size_t)s * q_stride;
    const Fe z = pc[b].points[slots.slot[s]];
    const uint32_t i0 = tid * strip;
    const uint32_t i1 = tid < lanes ? (i0 + strip < n ? i0 + strip : n) : i0;  // (lanes beyond the polynomial: empty strips)
    Fe acc = fe_zero();
    for (uint32_t i = i1; i-- > i0;) {
        acc = Fr::mul(z, acc);
        if (i + 1 < n) acc = Fr::add(acc, ldg(ap + i + 1));
    }
    sh[tid] = acc;
    __syncthreads();
    Fe w = Fr::pow_u64(z, strip);  // z^S, squared at every step
    for (uint32_t off = 1; off < KD_LANES; off <<= 1) {
        Fe t = fe_zero();
        const bool has = tid + off < KD_LANES;
        if (has) t = sh[tid + off];
        __syncthreads();
        if (has) sh[tid] = Fr::add(sh[tid], Fr::mul(w, t));
        w = Fr::sqr(w);
        __syncthreads();
    }
    acc = tid + 1 < KD_LANES ? sh[tid + 1] : fe_zero();  // q at the first coefficient of the strip above
    for (uint32_t i = i1; i-- > i0;) {
        acc = Fr::mul(z, acc);
        if (i + 1 < n) acc = Fr::add(acc, ldg(ap + i + 1));
        stg(qp + i, acc);
    }
}

// ---- latency form (a lone proof): block-local weighted scans, coalesced accesses, three short launches -- ~30 products
// per coefficient, but 70 us against the strip form's 130 us when four polynomials are all there is
constexpr uint32_t KD_BLOCK = 256;

__global__ __launch_bounds__(KD_BLOCK) void kd_local_kernel(const ProofConst* __restrict__ pc, KdSlots slots,
                                                            const Fe* __restrict__ a, size_t a_stride, size_t a_bs,
                                                            Fe* __restrict__ loc, Fe* __restrict__ heads, uint32_t n,
                                                            uint32_t nblk) {
    __shared__ Fe sh[KD_BLOCK];
    const uint32_t tid = threadIdx.x, blk = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
    const uint32_t q = b * gridDim.y + s;
    const uint32_t i = blk * KD_BLOCK + tid;
    const Fe* ap = a + (size_t)b * a_bs + (size_t)s * a_stride;
    Fe v = (i + 1 < n) ? ldg(ap + i + 1) : fe_zero();
    sh[tid] = v;
    __syncthreads();
    Fe w = pc[b].points[slots.slot[s]];  // z^off
    for (uint32_t off = 1; off < KD_BLOCK; off <<= 1) {
        Fe t = fe_zero();
        const bool has = tid + off < KD_BLOCK;
        if (has) t = sh[tid + off];
        __syncthreads();
        if (has) sh[tid] = Fr::add(sh[tid], Fr::mul(w, t));
        w = Fr::sqr(w);
        __syncthreads();
    }
    if (i < n) stg(loc + (size_t)q * n + i, sh[tid]);
    if (tid == 0) stg(heads + (size_t)q * nblk + blk, sh[0]);
}

// heads[blk] <- sum_{blk' > blk} heads[blk'] * (z^256)^(blk' - blk - 1): the carry entering block blk from above
__global__ __launch_bounds__(1024) void kd_heads_kernel(const ProofConst* __restrict__ pc, KdSlots slots, uint32_t nsets,
                                                        Fe* __restrict__ heads, uint32_t nblk) {
    __shared__ Fe sh[1024];
    const uint32_t tid = threadIdx.x, q = blockIdx.x;
    const uint32_t b = q / nsets, s = q % nsets;
    Fe* hp = heads + (size_t)q * nblk;
    sh[tid] = tid < nblk ? ldg(hp + tid) : fe_zero();
    __syncthreads();
    Fe w = Fr::pow_u64(pc[b].points[slots.slot[s]], KD_BLOCK);
    uint32_t span = 64;  // (entries at and beyond nblk are zero: the scan only has to span the block heads)
    while (span < nblk) span <<= 1;
    for (uint32_t off = 1; off < span; off <<= 1) {
        Fe t = fe_zero();
        const bool has = tid + off < span;
        if (has) t = sh[tid + off];
        __syncthreads();
        if (has) sh[tid] = Fr::add(sh[tid], Fr::mul(w, t));
        w = Fr::sqr(w);
        __syncthreads();
    }
    Fe carry = tid + 1 < span ? sh[tid + 1] : fe_zero();
    __syncthreads();
    if (tid < nblk) stg(hp + tid, carry);
}

__global__ __launch_bounds__(KD_BLOCK) void kd_apply_kernel(const ProofConst* __restrict__ pc, KdSlots slots,
                                                            const Fe* __restrict__ loc, const Fe* __restrict__ heads,
                                                            Fe* __restrict__ qo, size_t q_stride, size_t q_bs, uint32_t n,
                                                            uint32_t nblk) {
    const uint32_t tid = threadIdx.x, blk = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
    const uint32_t q = b * gridDim.y + s;
    const uint32_t i = blk * KD_BLOCK + tid;
    if (i >= n) return;
    Fe v = ldg(loc + (size_t)q * n + i);
    Fe carry = ldg(heads + (size_t)q * nblk + blk);
    if (!fe_is_zero(carry)) v = Fr::add(v, Fr::mul(Fr::pow_u64(pc[b].points[slots.slot[s]], KD_BLOCK - tid), carry));
    stg(qo + (size_t)b * q_bs + (size_t)s * q_stride + i, v);
}

size_t poly_kate_tmp_elems(uint32_t n, uint32_t batch) {  // (the latency form's scratch; the strip form needs none)
    size_t nblk = (n + KD_BLOCK - 1) / KD_BLOCK;
    return (size_t)batch * n + (size_t)batch * nblk + batch;
}

int poly_kate_division(zg_ctx* ctx, const ProofConst* pc, uint32_t nb, const uint32_t* slots, uint32_t nsets, const Fe* a,
                       size_t a_stride, size_t a_bs, Fe* q, size_t q_stride, size_t q_bs, Fe* tmp, uint32_t n) {
    if (!nb || !nsets || !n) return ZG_OK;
    ZG_REQUIRE(nsets <= HC_MAX_SETS, ZG_ERR_UNSUPPORTED, "kate division: %u point sets", nsets);
    KdSlots ks;
    memset(&ks, 0, sizeof(ks));
    for (uint32_t s = 0; s < nsets; s++) {
        ZG_REQUIRE(slots[s] < PC_MAX_POINTS, ZG_ERR_INVALID_ARG, "kate division: point slot %u", slots[s]);
        ks.slot[s] = slots[s];
    }
    if (ctx->msm_pair && n <= (1u << 18)) {  // latency form (tmp per poly_kate_tmp_elems(n, nb * nsets))
        const uint32_t nblk = (n + KD_BLOCK - 1) / KD_BLOCK;
        const uint32_t m = nb * nsets;
        Fe* loc = tmp;
        Fe* heads = loc + (size_t)m * n;
        const double bytes = (double)m * n * 64;
        ZG_LAUNCH(ctx, "kate_local", bytes, kd_local_kernel, dim3(nblk, nsets, nb), dim3(KD_BLOCK), 0, pc, ks, a, a_stride, a_bs, loc,
                  heads, n, nblk);
        // (as many lanes as block heads, a power of two from 64: at n = 2^14 one wave, whose barriers cost nothing)
        uint32_t hl = 64;
        while (hl < nblk) hl <<= 1;
        ZG_LAUNCH(ctx, "kate_heads", (double)m * nblk * 64, kd_heads_kernel, dim3(m), dim3(hl), 0, pc, ks, nsets, heads, nblk);
        ZG_LAUNCH(ctx, "kate_apply", bytes, kd_apply_kernel, dim3(nblk, nsets, nb), dim3(KD_BLOCK), 0, pc, ks, loc, heads, q, q_stride,
                  q_bs, n, nblk);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    const uint32_t lanes = n < KD_LANES ? n : KD_LANES;
    const uint32_t strip = (n + lanes - 1) / lanes;
    ZG_LAUNCH(ctx, "kate_division", (double)nb * nsets * n * 64, kd_strip_kernel, dim3(nsets, nb), dim3(KD_LANES), 0, pc, ks, a,
              a_stride, a_bs, q, q_stride, q_bs, n, lanes, strip);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// ------------------------------------------------------------------ l_0 / l_last / l_blind
__global__ void l_init_kernel(Fe* l0, Fe* llast, Fe* lblind, uint32_t n, uint32_t bf) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe one = Fr::one(), zero = fe_zero();
    stg(l0 + i, i == 0 ? one : zero);
    stg(llast + i, i == n - bf - 1 ? one : zero);
    stg(lblind + i, i >= n - bf ? one : zero);
}
int poly_l_cosets_init(zg_ctx* ctx, Fe* l0, Fe* llast, Fe* lblind, uint32_t n, uint32_t bf) {
    ZG_LAUNCH(ctx, "l_init", 0, l_init_kernel, dim3((n + 255) / 256), dim3(256), 0, l0, llast, lblind, n, bf);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}
__global__ void lactive_kernel(Fe* lactive, const Fe* llast, const Fe* lblind, uint32_t en, Fe one) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < en) stg(lactive + i, Fr::sub(one, Fr::add(ldg(llast + i), ldg(lblind + i))));
}
int poly_lactive(zg_ctx* ctx, Fe* lactive, const Fe* llast, const Fe* lblind, uint32_t en, bool hat) {
    const Fe one = hat ? Fr::mul(Fr::one(), Fr9Params::c261_fe()) : Fr::one();  // "1" in the slabs' form
    ZG_LAUNCH(ctx, "lactive", 0, lactive_kernel, dim3((en + 255) / 256), dim3(256), 0, lactive, llast, lblind, en, one);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

__global__ void scale_kernel(const Fe* in, Fe* out, size_t count, Fe factor) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) stg(out + i, Fr::mul(ldg(in + i), factor));
}
int poly_scale(zg_ctx* ctx, const Fe* in, Fe* out, size_t count, const Fe& factor) {
    if (!count) return ZG_OK;
    ZG_LAUNCH(ctx, "scale", (double)count * 64, scale_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, in, out,
              count, factor);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// ---- interpolation across the two cosets of the split extended domain (prover.hip); proof b = blockIdx.y
__global__ void fold_kernel(const Fe* a, size_t a_bs, uint32_t len, uint32_t parts, Fe e, Fe* out, size_t out_bs) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= len) return;
    a += (size_t)blockIdx.y * a_bs;
    Fe acc = ldg(a + (size_t)(parts - 1) * len + r);
    for (uint32_t q = parts - 1; q-- > 0;) acc = Fr::add(Fr::mul(acc, e), ldg(a + (size_t)q * len + r));
    stg(out + (size_t)blockIdx.y * out_bs + r, acc);
}
int poly_fold(zg_ctx* ctx, uint32_t nb, const Fe* a, size_t a_bs, uint32_t len, uint32_t parts, const Fe& e, Fe* out,
              size_t out_bs) {
    ZG_REQUIRE(parts >= 1 && len >= 1, ZG_ERR_INVALID_ARG, "poly_fold: %u parts of %u", parts, len);
    if (!nb) return ZG_OK;
    ZG_LAUNCH(ctx, "fold", (double)nb * (parts + 1) * len * 32, fold_kernel, dim3((len + 255) / 256, nb), dim3(256), 0, a, a_bs, len,
              parts, e, out, out_bs);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}
__global__ void diff_scale_kernel(const Fe* u, size_t u_bs, Fe cu, const Fe* v, size_t v_bs, Fe scale, Fe* out, size_t out_bs,
                                  uint32_t len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i < len)
        stg(out + (size_t)b * out_bs + i,
            Fr::mul(Fr::sub(Fr::mul(ldg(u + (size_t)b * u_bs + i), cu), ldg(v + (size_t)b * v_bs + i)), scale));
}
int poly_diff_scale(zg_ctx* ctx, uint32_t nb, const Fe* u, size_t u_bs, const Fe& cu, const Fe* v, size_t v_bs, const Fe& scale,
                    Fe* out, size_t out_bs, uint32_t len) {
    if (!nb) return ZG_OK;
    ZG_LAUNCH(ctx, "diff_scale", (double)nb * len * 96, diff_scale_kernel, dim3((len + 255) / 256, nb), dim3(256), 0, u, u_bs, cu, v,
              v_bs, scale, out, out_bs, len);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}
__global__ void split_combine_kernel(Fe* h, size_t h_bs, const Fe* bq, size_t b_bs, uint32_t len, Fe c1, uint32_t hi_at) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= len) return;
    h += (size_t)blockIdx.y * h_bs;
    const Fe bj = ldg(bq + (size_t)blockIdx.y * b_bs + j);
    stg(h + j, Fr::sub(ldg(h + j), Fr::mul(c1, bj)));
    stg(h + hi_at + j, bj);
}
int poly_split_combine(zg_ctx* ctx, uint32_t nb, Fe* h, size_t h_bs, const Fe* b, size_t b_bs, uint32_t len, const Fe& c1,
                       uint32_t hi_at) {
    if (!nb) return ZG_OK;
    ZG_LAUNCH(ctx, "split_combine", (double)nb * len * 128, split_combine_kernel, dim3((len + 255) / 256, nb), dim3(256), 0, h, h_bs,
              b, b_bs, len, c1, hi_at);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

}  // namespace zg
