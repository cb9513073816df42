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
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

// SplitMix64 stream keyed by (seed, tag, index); rejection-sample a canonical value < r
__host__ __device__ __forceinline__ Fe rand_fr(uint64_t seed, uint32_t tag, uint64_t index) {
    uint64_t st = mix64(seed + 0x9e3779b97f4a7c15ULL * (uint64_t)(tag + 1)) ^
                  mix64(index + 0xd1b54a32d192ed03ULL * (uint64_t)(tag + 1));
    Fe v;
    for (;;) {
        for (int i = 0; i < 4; i++) {
            st += 0x9e3779b97f4a7c15ULL;
            uint64_t w = mix64(st);
            v.l[2 * i] = (uint32_t)w;
            v.l[2 * i + 1] = (uint32_t)(w >> 32);
        }
        v.l[7] &= 0x3fffffffu;
        bool lt = false;
        for (int i = 7; i >= 0; i--) {
            uint32_t p = FrParams::p(i);
            if (v.l[i] < p) { lt = true; break; }
            if (v.l[i] > p) break;
        }
        if (lt) break;
    }
    return Fr::from_raw(v);
}

__global__ void blind_rows_kernel(Fe* base, size_t col_stride, uint32_t ncols, uint32_t row0, uint32_t nrows,
                                  uint64_t seed, uint32_t tag) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncols * nrows) return;
    uint32_t c = t / nrows, j = t % nrows;
    stg(base + (size_t)c * col_stride + row0 + j, rand_fr(seed, tag, (uint64_t)c * nrows + j));
}

__global__ void random_kernel(Fe* out, uint32_t n, uint64_t seed, uint32_t tag) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stg(out + i, rand_fr(seed, tag, i));
}

int poly_blind_rows(zg_ctx* ctx, Fe* base, size_t col_stride, uint32_t ncols, uint32_t row0, uint32_t nrows,
                    uint64_t seed, uint32_t tag) {
    uint32_t total = ncols * nrows;
    if (total == 0) return ZG_OK;
    ZG_LAUNCH(ctx, "blind_rows", 0, blind_rows_kernel, dim3((total + 63) / 64), dim3(64), 0, base, col_stride, ncols,
              row0, nrows, seed, tag);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int poly_random(zg_ctx* ctx, Fe* out, uint32_t n, uint64_t seed, uint32_t tag) {
    ZG_LAUNCH(ctx, "random_poly", (double)n * 32, random_kernel, dim3((n + 255) / 256), dim3(256), 0, out, n, seed, tag);
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
        if (mo->coeff_is_one && nf > 0) {
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
__global__ __launch_bounds__(256) void lookup_compress_kernel(DevCircuit c, Cols cols, Fe theta, Fe* cin, Fe* ctab,
                                                              uint32_t n) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t l = blockIdx.y;
    if (row >= n) return;
    const DLookup* lk = c.lookups + l;
    Fe ai = fe_zero(), ti = fe_zero();
    for (uint32_t e = 0; e < lk->width; e++) {
        ai = Fr::add(Fr::mul(ai, theta), eval_poly(c, cols, lk->inputs[e], row));
        ti = Fr::add(Fr::mul(ti, theta), eval_poly(c, cols, lk->tables[e], row));
    }
    stg(cin + (size_t)l * n + row, ai);
    stg(ctab + (size_t)l * n + row, ti);
}

int poly_lookup_compress(zg_ctx* ctx, const DevCircuit& c, const Cols& cols, const Fe& theta, Fe* cin, Fe* ctab,
                         uint32_t n) {
    if (c.n_lookups == 0) return ZG_OK;
    ZG_LAUNCH(ctx, "lookup_compress", (double)c.n_lookups * n * 64, lookup_compress_kernel,
              dim3((n + 255) / 256, c.n_lookups), dim3(256), 0, c, cols, theta, cin, ctab, n);
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
__global__ __launch_bounds__(256) void lookup_terms_kernel(const Fe* cin, const Fe* ctab, const Fe* pin, const Fe* ptab,
                                                           Fe beta, Fe gamma, Fe* num, Fe* den, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t o = (size_t)blockIdx.y * n + i;
    stg(den + o, Fr::mul(Fr::add(ldg(pin + o), beta), Fr::add(ldg(ptab + o), gamma)));
    stg(num + o, Fr::mul(Fr::add(ldg(cin + o), beta), Fr::add(ldg(ctab + o), gamma)));
}

int poly_lookup_terms(zg_ctx* ctx, const Fe* cin, const Fe* ctab, const Fe* pin, const Fe* ptab, const Fe& beta,
                      const Fe& gamma, Fe* num, Fe* den, uint32_t n, uint32_t n_lookups) {
    if (!n_lookups) return ZG_OK;
    ZG_LAUNCH(ctx, "lookup_terms", (double)n_lookups * n * 192, lookup_terms_kernel, dim3((n + 255) / 256, n_lookups),
              dim3(256), 0, cin, ctab, pin, ptab, beta, gamma, num, den, n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// permutation commit, one set per blockIdx.y:
//   den = prod_c (v_c + beta*sigma_c + gamma),  num = prod_c (v_c + delta^c * omega^i * beta + gamma)
__global__ __launch_bounds__(256) void perm_terms_kernel(DevCircuit c, Cols cols, const Fe* sigma_val,
                                                         const Fe* omega_tw, Fe beta, Fe gamma, Fe* num, Fe* den,
                                                         uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = blockIdx.y;
    if (i >= n) return;
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
    stg(den + (size_t)s * n + i, d);
    stg(num + (size_t)s * n + i, m);
}

int poly_perm_terms(zg_ctx* ctx, const DevCircuit& c, const Cols& cols, const Fe* sigma_val, const Fe* omega_tw,
                    const Fe& beta, const Fe& gamma, Fe* num, Fe* den, uint32_t n) {
    if (!c.n_sets) return ZG_OK;
    ZG_LAUNCH(ctx, "perm_terms", (double)c.n_perm * n * 64 + (double)c.n_sets * n * 64, perm_terms_kernel,
              dim3((n + 255) / 256, c.n_sets), dim3(256), 0, c, cols, sigma_val, omega_tw, beta, gamma, num, den, n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// ------------------------------------------------------------------ grand product
// z[0] = z0, z[i+1] = z[i] * num[i] / den[i].  One 1024-lane workgroup per product: every lane owns a
// strip; strip-local Montgomery batch inversion (one binary-Euclid inversion per lane), then a
// Hillis-Steele product scan across the lanes through LDS.  Zero denominators invert to zero, as
// halo2's BatchInvert leaves them.
__global__ __launch_bounds__(1024) void grand_product_kernel(const Fe* __restrict__ num, const Fe* __restrict__ den,
                                                             const Fe* const* __restrict__ z0p, Fe* __restrict__ z,
                                                             Fe* __restrict__ tmp, uint32_t n) {
    __shared__ Fe sh[1024];
    const uint32_t tid = threadIdx.x, b = blockIdx.x;
    num += (size_t)b * n; den += (size_t)b * n; z += (size_t)b * n; tmp += (size_t)b * n;
    const uint32_t L = (n + 1023) / 1024;
    uint32_t lo = tid * L, hi = lo + L;
    if (lo > n) lo = n;
    if (hi > n) hi = n;
    // forward: prefix products of the strip's denominators
    Fe p = Fr::one();
    for (uint32_t i = lo; i < hi; i++) {
        stg(tmp + i, p);
        Fe d = ldg(den + i);
        if (!fe_is_zero(d)) p = Fr::mul(p, d);
    }
    Fe acc = Fr::inv(p);
    // backward: inverse of each denominator, ratio = num/den (kept in tmp), strip product of ratios
    for (uint32_t i = hi; i-- > lo;) {
        Fe d = ldg(den + i);
        Fe inv = fe_zero();
        if (!fe_is_zero(d)) {
            inv = Fr::mul(acc, ldg(tmp + i));
            acc = Fr::mul(acc, d);
        }
        stg(tmp + i, Fr::mul(ldg(num + i), inv));
    }
    Fe r = Fr::one();
    for (uint32_t i = lo; i < hi; i++) r = Fr::mul(r, ldg(tmp + i));
    sh[tid] = r;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        Fe v = Fr::one();
        if (tid >= off) v = sh[tid - off];
        __syncthreads();
        if (tid >= off) sh[tid] = Fr::mul(sh[tid], v);
        __syncthreads();
    }
    Fe start = (z0p && z0p[b]) ? ldg(z0p[b]) : Fr::one();
    if (tid > 0) start = Fr::mul(start, sh[tid - 1]);
    for (uint32_t i = lo; i < hi; i++) {
        stg(z + i, start);
        start = Fr::mul(start, ldg(tmp + i));
    }
}

int poly_grand_product(zg_ctx* ctx, const Fe* num, const Fe* den, const Fe* const* d_z0, Fe* z, Fe* tmp, uint32_t n,
                       uint32_t batch) {
    if (!batch || !n) return ZG_OK;
    ZG_LAUNCH(ctx, "grand_product", (double)batch * n * 96, grand_product_kernel, dim3(batch), dim3(1024), 0, num, den,
              d_z0, z, tmp, n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// ------------------------------------------------------------------ evaluate_h
// Evaluator::evaluate_h for one circuit instance: gates, permutation argument, lookup arguments folded
// by y on every point of the extended coset; the division by (X^n - 1) of vanishing::construct is
// fused into the store (t_eval has period 2^(ext_k - k)).
__global__ __launch_bounds__(256) void evaluate_h_kernel(EvalHArgs a, uint32_t en) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= en) return;
    const DevCircuit& c = a.c;
    const uint32_t mask = en - 1;
    const uint32_t rs = (uint32_t)a.cols.rot_scale;
    const uint32_t r_next = (idx + rs) & mask;
    const uint32_t r_prev = (idx - rs) & mask;
    const uint32_t r_last = (idx + (uint32_t)(a.last_rot * (int32_t)rs)) & mask;
    Fe value = fe_zero();
    for (uint32_t g = 0; g < c.n_gates; g++)
        value = Fr::add(Fr::mul(value, a.y), eval_poly(c, a.cols, c.gates[g], idx));

    const Fe l0 = ldg(a.l0 + idx), llast = ldg(a.llast + idx), lactive = ldg(a.lactive + idx);
    if (c.n_sets > 0) {
        const Fe zf = ldg(a.pz_cos + idx);
        const Fe zl = ldg(a.pz_cos + (size_t)(c.n_sets - 1) * en + idx);
        value = Fr::add(Fr::mul(value, a.y), Fr::mul(Fr::sub(Fr::one(), zf), l0));
        value = Fr::add(Fr::mul(value, a.y), Fr::mul(Fr::sub(Fr::sqr(zl), zl), llast));
        for (uint32_t s = 1; s < c.n_sets; s++) {
            Fe t = Fr::sub(ldg(a.pz_cos + (size_t)s * en + idx), ldg(a.pz_cos + (size_t)(s - 1) * en + r_last));
            value = Fr::add(Fr::mul(value, a.y), Fr::mul(t, l0));
        }
        Fe current_delta = Fr::mul(a.delta_start, ldg(a.ext_tw + idx));
        for (uint32_t s = 0; s < c.n_sets; s++) {
            uint32_t c0 = s * c.chunk, c1 = c0 + c.chunk;
            if (c1 > c.n_perm) c1 = c.n_perm;
            Fe left = ldg(a.pz_cos + (size_t)s * en + r_next);
            Fe right = ldg(a.pz_cos + (size_t)s * en + idx);
            for (uint32_t col = c0; col < c1; col++) {
                const zg_query q = c.perm_cols[col];
                const Fe* base = q.kind == ZG_FIXED ? a.cols.fixed : q.kind == ZG_ADVICE ? a.cols.advice : a.cols.instance;
                Fe v = ldg(base + ((size_t)q.column << a.cols.log_size) + idx);
                Fe sg = ldg(a.sigma_cos + (size_t)col * en + idx);
                left = Fr::mul(left, Fr::add(Fr::add(Fr::mul(a.beta, sg), v), a.gamma));
                right = Fr::mul(right, Fr::add(Fr::add(v, current_delta), a.gamma));
                current_delta = Fr::mul(current_delta, a.delta);
            }
            value = Fr::add(Fr::mul(value, a.y), Fr::mul(Fr::sub(left, right), lactive));
        }
    }
    for (uint32_t l = 0; l < c.n_lookups; l++) {
        const DLookup* lk = c.lookups + l;
        Fe ai = fe_zero(), ti = fe_zero();
        for (uint32_t e = 0; e < lk->width; e++) {
            ai = Fr::add(Fr::mul(ai, a.theta), eval_poly(c, a.cols, lk->inputs[e], idx));
            ti = Fr::add(Fr::mul(ti, a.theta), eval_poly(c, a.cols, lk->tables[e], idx));
        }
        const Fe* zc = a.lz_cos + (size_t)l * en;
        const Fe* ap = a.pin_cos + (size_t)l * en;
        const Fe* sp = a.ptab_cos + (size_t)l * en;
        const Fe z = ldg(zc + idx), apv = ldg(ap + idx), spv = ldg(sp + idx);
        value = Fr::add(Fr::mul(value, a.y), Fr::mul(Fr::sub(Fr::one(), z), l0));
        value = Fr::add(Fr::mul(value, a.y), Fr::mul(Fr::sub(Fr::sqr(z), z), llast));
        Fe lft = Fr::mul(Fr::mul(Fr::add(apv, a.beta), Fr::add(spv, a.gamma)), ldg(zc + r_next));
        Fe rgt = Fr::mul(Fr::mul(Fr::add(ai, a.beta), Fr::add(ti, a.gamma)), z);
        value = Fr::add(Fr::mul(value, a.y), Fr::mul(Fr::sub(lft, rgt), lactive));
        Fe ams = Fr::sub(apv, spv);
        value = Fr::add(Fr::mul(value, a.y), Fr::mul(ams, l0));
        value = Fr::add(Fr::mul(value, a.y), Fr::mul(Fr::mul(ams, Fr::sub(apv, ldg(ap + r_prev))), lactive));
    }
    // divide_by_vanishing_poly
    value = Fr::mul(value, ldg(a.t_eval + (idx & a.t_mask)));
    stg(a.h + idx, value);
}

int poly_evaluate_h(zg_ctx* ctx, const EvalHArgs& a, uint32_t en) {
    // algorithmic bytes: every input coset read once + h written
    const DevCircuit& c = a.c;
    double arrays = 3.0 + c.n_perm + c.n_sets + 3.0 * c.n_lookups + 1.0;  // l-polys, sigma, z's, lookup polys, h
    ZG_LAUNCH(ctx, "evaluate_h", arrays * en * 32.0, evaluate_h_kernel, dim3((en + 255) / 256), dim3(256), 0, a, en);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// ------------------------------------------------------------------ eval_polynomial
// pow[p][i] = x_p^i : lane computes x^(i0) by square-and-multiply, then a strip of 16 products
__global__ __launch_bounds__(256) void powers_kernel(const Fe* __restrict__ points, uint32_t n, Fe* __restrict__ pw) {
    constexpr uint32_t CH = 16;
    uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * CH;
    if (i0 >= n) return;
    Fe x = ldg(points + blockIdx.y);
    Fe cur = Fr::pow_u64(x, i0);
    Fe* out = pw + (size_t)blockIdx.y * n;
    for (uint32_t j = 0; j < CH && i0 + j < n; j++) {
        stg(out + i0 + j, cur);
        cur = Fr::mul(cur, x);
    }
}

int poly_powers(zg_ctx* ctx, const Fe* points_host, uint32_t npoints, uint32_t n, Fe* d_pow) {
    if (!npoints) return ZG_OK;
    // the points ride at the head of the table's own allocation (caller reserves npoints extra entries)
    Fe* d_pts = d_pow + (size_t)npoints * n;
    ZG_HIP(hipMemcpyAsync(d_pts, points_host, npoints * sizeof(Fe), hipMemcpyHostToDevice, ctx->stream));
    ZG_LAUNCH(ctx, "powers", (double)npoints * n * 32, powers_kernel, dim3((n + 256 * 16 - 1) / (256 * 16), npoints),
              dim3(256), 0, d_pts, n, d_pow);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// out[j] = <poly_j, pow_{point_j}> : one workgroup per (poly, point) pair
__global__ __launch_bounds__(256) void dot_kernel(const Fe* __restrict__ polys, size_t stride, uint32_t n,
                                                  const uint32_t* __restrict__ poly_idx,
                                                  const uint32_t* __restrict__ point_idx, const Fe* __restrict__ pw,
                                                  Fe* __restrict__ out) {
    __shared__ Fe sh[256];
    const uint32_t j = blockIdx.x, tid = threadIdx.x;
    const Fe* p = polys + (size_t)poly_idx[j] * stride;
    const Fe* w = pw + (size_t)point_idx[j] * n;
    Fe acc = fe_zero();
    for (uint32_t i = tid; i < n; i += 256) acc = Fr::add(acc, Fr::mul(ldg(p + i), ldg(w + i)));
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t off = 128; off > 0; off >>= 1) {
        if (tid < off) sh[tid] = Fr::add(sh[tid], sh[tid + off]);
        __syncthreads();
    }
    if (tid == 0) stg(out + j, sh[0]);
}

int poly_dot(zg_ctx* ctx, const Fe* polys, size_t stride, uint32_t n, const uint32_t* d_poly_idx,
             const uint32_t* d_point_idx, const Fe* d_pow, uint32_t count, Fe* d_out) {
    if (!count) return ZG_OK;
    ZG_LAUNCH(ctx, "eval_dot", (double)count * n * 64, dot_kernel, dim3(count), dim3(256), 0, polys, stride, n, d_poly_idx,
              d_point_idx, d_pow, d_out);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// ------------------------------------------------------------------ GWC
// out[i] = (...((p_0[i]) v + p_1[i]) v + ...) + p_{m-1}[i];  out[0] -= sub
__global__ __launch_bounds__(256) void horner_combine_kernel(const Fe* __restrict__ polys, size_t stride,
                                                             const uint32_t* __restrict__ list, uint32_t count, Fe v,
                                                             Fe sub, Fe* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe acc = fe_zero();
    for (uint32_t j = 0; j < count; j++) acc = Fr::add(Fr::mul(acc, v), ldg(polys + (size_t)list[j] * stride + i));
    if (i == 0) acc = Fr::sub(acc, sub);
    stg(out + i, acc);
}

int poly_horner_combine(zg_ctx* ctx, const Fe* polys, size_t stride, const uint32_t* d_list, uint32_t count,
                        const Fe& v, const Fe& sub, Fe* out, uint32_t n) {
    ZG_LAUNCH(ctx, "horner_combine", (double)(count + 1) * n * 32, horner_combine_kernel, dim3((n + 255) / 256), dim3(256),
              0, polys, stride, d_list, count, v, sub, out, n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// kate_division: q_i = a_{i+1} + z q_{i+1}, i = n-2 .. 0, q_{n-1} = 0.  Strip-local recurrences joined by
// a weighted suffix scan across the 1024 lanes (weights z^(L*s)).
__global__ __launch_bounds__(1024) void kate_division_kernel(const Fe* __restrict__ a, uint32_t n, Fe z,
                                                             Fe* __restrict__ q) {
    __shared__ Fe sh[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t L = (n + 1023) / 1024;
    const uint32_t lo = tid * L, hi = lo + L;  // strip of q indices [lo, hi); a beyond n-1 counts as 0
    Fe loc = fe_zero();
    for (uint32_t i = hi; i-- > lo;) {
        Fe an = (i + 1 < n) ? ldg(a + i + 1) : fe_zero();
        loc = Fr::add(an, Fr::mul(z, loc));
    }
    sh[tid] = loc;  // q_lo assuming q_hi = 0
    __syncthreads();
    Fe w = Fr::pow_u64(z, L);  // z^L
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        Fe v = fe_zero();
        if (tid + off < 1024) v = sh[tid + off];
        __syncthreads();
        if (tid + off < 1024) sh[tid] = Fr::add(sh[tid], Fr::mul(w, v));
        w = Fr::sqr(w);
        __syncthreads();
    }
    Fe carry = (tid + 1 < 1024) ? sh[tid + 1] : fe_zero();  // true q at index hi
    for (uint32_t i = hi; i-- > lo;) {
        Fe an = (i + 1 < n) ? ldg(a + i + 1) : fe_zero();
        carry = Fr::add(an, Fr::mul(z, carry));
        if (i < n) stg(q + i, carry);
    }
}

int poly_kate_division(zg_ctx* ctx, const Fe* a, uint32_t n, const Fe& z, Fe* q) {
    ZG_LAUNCH(ctx, "kate_division", (double)n * 64, kate_division_kernel, dim3(1), dim3(1024), 0, a, n, z, q);
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
__global__ void lactive_kernel(Fe* lactive, const Fe* llast, const Fe* lblind, uint32_t en) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < en) stg(lactive + i, Fr::sub(Fr::one(), Fr::add(ldg(llast + i), ldg(lblind + i))));
}
int poly_lactive(zg_ctx* ctx, Fe* lactive, const Fe* llast, const Fe* lblind, uint32_t en) {
    ZG_LAUNCH(ctx, "lactive", 0, lactive_kernel, dim3((en + 255) / 256), dim3(256), 0, lactive, llast, lblind, en);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

}  // namespace zg
