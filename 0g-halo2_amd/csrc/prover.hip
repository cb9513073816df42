// create_proof for one circuit instance on one MI355X -- replaces
// halo2_proofs::plonk::create_proof::<KZGCommitmentScheme<Bn256>, ProverGWC, _, _, EvmTranscript, _>
// as Wnn::proof calls it (/root/reference/src/wnn.rs:232-262; upstream v2023_04_20 src/plonk/prover.rs).
//
// Everything between "advice columns assigned" and "proof bytes" stays in HBM: columns, coefficient
// forms, extended cosets, lookup/permutation products, h(X).  The host only sees what the Fiat-Shamir
// transcript needs -- commitments (one 128-B XYZZ point each), evaluations (32 B each).  Even
// lookup::prover::permute_expression_pair (a sort + BTreeMap walk upstream) runs on the device
// (sort.hip).  Work that does not depend on the next challenge (iNTTs, coset NTTs) runs on a side
// stream while the main stream works through the commitment MSM and the host hashes.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <memory>

#include "poly.h"
#include "field9.h"
#include "transcript.h"

namespace zg {

// ------------------------------------------------------------------ Keccak-256 (original padding)
static inline uint64_t rol64(uint64_t x, unsigned s) { return s ? (x << s) | (x >> (64 - s)) : x; }

static void keccak_f1600(uint64_t a[25]) {
    static const uint64_t rc[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    // rho offsets indexed [x + 5y]
    static const unsigned rho[25] = {0,  1,  62, 28, 27, 36, 44, 6,  55, 20, 3,  10, 43,
                                     25, 39, 41, 45, 15, 21, 8,  18, 2,  61, 56, 14};
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        // rho + pi: b[y, 2x+3y] = rot(a[x, y])
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[x + 5 * y], rho[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= rc[round];
    }
}

void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
    constexpr size_t rate = 136;
    uint64_t st[25] = {0};
    auto absorb = [&](const uint8_t* blk) {
        for (size_t i = 0; i < rate / 8; i++) {
            uint64_t w = 0;
            for (int j = 0; j < 8; j++) w |= (uint64_t)blk[8 * i + j] << (8 * j);
            st[i] ^= w;
        }
        keccak_f1600(st);
    };
    while (len >= rate) {
        absorb(data);
        data += rate;
        len -= rate;
    }
    uint8_t last[rate];
    memset(last, 0, rate);
    memcpy(last, data, len);
    last[len] ^= 0x01;
    last[rate - 1] ^= 0x80;
    absorb(last);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(st[i] >> (8 * j));
}

}  // namespace zg

using namespace zg;

// ------------------------------------------------------------------ prover object
struct zg_prover {
    zg_ctx* ctx = nullptr;
    uint32_t k = 0, ext_k = 0, cs_degree = 0, bf = 0, qpd = 0;
    uint32_t n = 0, en = 0, usable = 0;
    uint32_t F = 0, A = 0, I = 0, P = 0, NL = 0, sets = 0, chunk = 0;
    std::vector<zg_query> advice_queries, fixed_queries;
    DevCircuit dc{};
    std::vector<void*> owned;  // device allocations freed at destroy
    zg_bases *g = nullptr, *gl = nullptr;
    bool use_side = true;   // coefficient / coset forms on a side stream (latency) or inline (throughput)
    // evaluate_h on nine 29-bit limbs: the coset slabs, l-polynomials, t_eval and the monomial coefficients it
    // reads are kept in the 2^261 Montgomery form (x * 2^5 of the library form); ZG_EVALH9=0 turns it off
    bool hat = true;
    DMono* monos_hat = nullptr;
    zg_poly* gates_hat = nullptr;
    uint32_t* gate_common = nullptr;
    zg_poly* gate_uni = nullptr;
    Fe* uni_coef = nullptr;
    uint32_t* gate_slab = nullptr;  // per gate: index of its U(fixed cell) coset in gate_slabs, or 0xffffffff
    struct SlabJob { uint32_t gate, query, first, count; };
    std::vector<SlabJob> slab_jobs;  // filled when the gates are factored, run once the fixed cosets exist
    bool own_bases = true;  // false: tables shared with other provers of the same device
    Fe vk_repr{};
    Fe omega{}, omega_inv{}, ifft_div{};
    // pk-derived, resident
    Fe *fixed_val = nullptr, *sigma_val = nullptr, *omega_tw = nullptr;
    // The extended domain evaluate_h works on.  Either EvaluationDomain's own coset zeta * <omega_(2^ext_k)> (8n points
    // for degree 6), or -- split -- two cosets that together hold just the (degree - 1) * n points the quotient needs:
    // zeta * <omega_(m1 n)> and zeta^2 * <omega_(m2 n)>, m1 + m2 = degree - 1 (4n + n).  Every coset slab exists per part.
    struct Dom {
        uint32_t ek = 0, en = 0;
        int zpow = 1;  // the coset shift is zeta^zpow
        Fe *fixed_cos = nullptr, *sigma_cos = nullptr, *l0 = nullptr, *llast = nullptr, *lactive = nullptr,
           *gate_slabs = nullptr, *t_eval = nullptr, *ext_tw = nullptr;                                   // proving key
        Fe *adv_cos = nullptr, *inst_cos = nullptr, *pz_cos = nullptr, *lz_cos = nullptr, *perm_cos = nullptr,
           *h = nullptr;                                                                                   // per proof
    };
    Dom dom[3];               // [0]: the single coset; [1], [2]: the two parts of the split domain (when it applies)
    uint32_t nparts = 1;      // 1, or 3 when the split domain is prepared too
    bool last_split = false;  // which of the two the last proof used (zg_prover_fetch)
    Fe* split_tmp = nullptr;  // interpolation between the two parts: 3 * dom[2].en elements
    // coefficient-form slab [n_polys][n]
    Fe* polys = nullptr;
    uint32_t n_polys = 0;
    uint32_t ix_fixed = 0, ix_sigma = 0, ix_adv = 0, ix_inst = 0, ix_pz = 0, ix_lz = 0, ix_perm = 0, ix_random = 0,
             ix_hpiece = 0, ix_hpoly = 0;
    // per-proof buffers
    Fe *adv_val = nullptr, *inst_val = nullptr;
    Fe *cin = nullptr, *ctab = nullptr, *perm = nullptr /* [2NL][n]: a'_l, s'_l */, *zs = nullptr /* [sets+NL][n] */;
    Fe *num = nullptr, *den = nullptr, *tmp = nullptr, *pw = nullptr, *evals = nullptr, *wpoly = nullptr,
       *raw = nullptr, *sraw = nullptr, *sort_fe = nullptr;
    uint32_t *sort_u32 = nullptr, *d_err = nullptr;
    Fe *pin_c = nullptr, *ptab_c = nullptr;
    XYZZ* xyzz = nullptr;
    uint32_t* d_idx = nullptr;
    std::map<uint32_t*, std::vector<uint32_t>> uploaded_lists;  // what h2d_list left at each destination
    size_t inst_filled = 0;  // rows of inst_val that may be non-zero
    Fe* ktmp = nullptr;
    hipEvent_t ev = nullptr, ev_fork = nullptr, ev_join = nullptr;
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    size_t stage_off = 0;
    bool have_last = false;
    double phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace {

template <class T>
int dalloc(zg_prover* p, T** out, size_t count) {
    void* q = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
        set_error("zg_prover: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return ZG_ERR_OOM;
    }
    p->owned.push_back(q);
    *out = reinterpret_cast<T*>(q);
    return ZG_OK;
}

inline Fe to_fe(const zg_fr* s) {
    Fe r;
    memcpy(&r, s, 32);
    return r;
}

Fe rotate_omega(const zg_prover* p, const Fe& x, int32_t rot) {
    Fe w = rot >= 0 ? Fr::pow_u64(p->omega, (uint64_t)rot) : Fr::pow_u64(p->omega_inv, (uint64_t)(-(int64_t)rot));
    return Fr::mul(x, w);
}

// Small host->device transfers go through a pinned staging arena: hipMemcpyAsync from pageable memory
// blocks the calling thread until the stream has drained up to the copy, which serialises host and
// GPU inside a proof and throttles concurrent proof streams.  The arena is a bump allocator reset at
// the start of every proof; each region is written once per proof.
constexpr size_t PIN_RESULTS = 0;            // commitments D2H
constexpr size_t PIN_EVALS = 64 * 1024;      // evaluations D2H
constexpr size_t PIN_STAGE = 128 * 1024;     // H2D staging arena starts here
void* stage(zg_prover* p, const void* src, size_t bytes) {
    size_t off = (p->stage_off + 63) & ~size_t(63);
    if (off + bytes > p->pinned_cap - 4096) return nullptr;  // caller falls back to a direct copy
    void* dst = (char*)p->pinned + off;
    memcpy(dst, src, bytes);
    p->stage_off = off + bytes;
    return dst;
}
// Index lists that only depend on the circuit (which polynomial is opened where) come out the same for every
// proof: upload one when its content differs from what that destination already holds.
int h2d_list(zg_prover* p, uint32_t* d_dst, const std::vector<uint32_t>& list) {
    std::vector<uint32_t>& held = p->uploaded_lists[d_dst];
    if (held == list) return ZG_OK;
    const void* s = stage(p, list.data(), list.size() * 4);
    ZG_HIP(hipMemcpyAsync(d_dst, s ? s : list.data(), list.size() * 4, hipMemcpyHostToDevice, p->ctx->stream));
    if (!s) ZG_HIP(hipStreamSynchronize(p->ctx->stream));  // (the source was pageable memory of the caller)
    held = list;
    return ZG_OK;
}
int h2d(zg_prover* p, void* d_dst, const void* src, size_t bytes) {
    const void* s = stage(p, src, bytes);
    ZG_HIP(hipMemcpyAsync(d_dst, s ? s : src, bytes, hipMemcpyHostToDevice, p->ctx->stream));
    return ZG_OK;
}

// D2H of `count` XYZZ results behind the work already queued; returns after ONLY that copy is done
int fetch_points(zg_prover* p, size_t count, std::vector<Jac>& out) {
    zg_ctx* ctx = p->ctx;
    ZG_HIP(hipMemcpyAsync((char*)p->pinned + PIN_RESULTS, p->xyzz, count * sizeof(XYZZ), hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(hipEventRecord(p->ev, ctx->stream));
    return ZG_OK;
}
int wait_points(zg_prover* p, size_t count, std::vector<Jac>& out) {
    ZG_HIP(hipEventSynchronize(p->ev));
    out.resize(count);
    xyzz_batch_normalise((const XYZZ*)((char*)p->pinned + PIN_RESULTS), count, reinterpret_cast<zg_g1*>(out.data()));
    return ZG_OK;
}


// ---- evaluate_h's view of one gate: gate = U(cell f) * inner, f a query index present in every monomial.
// U(x) = x when f occurs exactly once per monomial (uc empty); otherwise the monomials are grouped by the
// power of f and, when every group is a scalar multiple of the lowest one, U(x) = sum_k uc[k-1] x^k --
// the shape halo2's selector compression leaves behind (selector -> q * prod_{u != t} (u - q)).  Failing
// that, one occurrence of f is split off and the rest stays expanded.  Coefficients are in the 2^261 form.
struct GateFactor {
    std::vector<DMono> inner;
    std::vector<Fe> uc;
    uint32_t cost = 0;  // field products per row
};

GateFactor factor_gate(const std::vector<DMono>& monos, zg_poly g, uint32_t f, const Fe& c261, bool tabulated) {
    auto strip = [&](const DMono& src, bool all) {  // src without one / every occurrence of f
        DMono d = src;
        uint32_t w = 0;
        bool dropped = false;
        for (uint32_t b = 0; b < src.n_factors; b++) {
            if (src.factors[b] == f && (all || !dropped)) { dropped = true; continue; }
            d.factors[w++] = src.factors[b];
        }
        for (uint32_t b = w; b < ZG_MAX_FACTORS; b++) d.factors[b] = 0;
        d.n_factors = w;
        return d;
    };
    auto power = [&](const DMono& d) { return (uint32_t)std::count(d.factors, d.factors + d.n_factors, f); };
    auto same_cells = [](const DMono& x, const DMono& y) {
        return x.n_factors == y.n_factors && std::equal(x.factors, x.factors + x.n_factors, y.factors);
    };
    auto cell_order = [](const DMono& x, const DMono& y) {
        return std::lexicographical_compare(x.factors, x.factors + x.n_factors, y.factors, y.factors + y.n_factors);
    };
    uint32_t pmin = ZG_MAX_FACTORS + 1, pmax = 0;
    for (uint32_t m = g.first; m < g.first + g.count; m++) {
        pmin = std::min(pmin, power(monos[m]));
        pmax = std::max(pmax, power(monos[m]));
    }
    GateFactor out;
    bool univariate = pmax > 1;
    if (univariate) {
        std::vector<std::vector<DMono>> by_power(pmax + 1);
        for (uint32_t m = g.first; m < g.first + g.count; m++) by_power[power(monos[m])].push_back(strip(monos[m], true));
        for (auto& grp : by_power) std::sort(grp.begin(), grp.end(), cell_order);
        std::vector<DMono>& base = by_power[pmin];
        const Fe b0_inv = Fr::inv(base[0].coeff);
        out.uc.assign(pmax, Fe{});
        out.uc[pmin - 1] = Fr::mul(Fr::one(), c261);
        for (uint32_t k = pmin + 1; k <= pmax && univariate; k++) {
            const auto& grp = by_power[k];
            if (grp.empty()) continue;
            univariate = grp.size() == base.size();
            for (size_t i = 0; i < grp.size() && univariate; i++)  // grp = ratio * base, term by term
                univariate = same_cells(grp[i], base[i]) &&
                             fe_eq(Fr::mul(grp[i].coeff, base[0].coeff), Fr::mul(grp[0].coeff, base[i].coeff));
            if (univariate) out.uc[k - 1] = Fr::mul(Fr::mul(grp[0].coeff, b0_inv), c261);
        }
        if (univariate) {
            // U * B = (d U) * (B / d) with d the most frequent coefficient of B: those monomials become
            // coefficient-free products again, as they were before the selector was substituted
            size_t best = 0, best_n = 0;
            for (size_t i = 0; i < base.size(); i++) {
                size_t cnt = 0;
                for (const DMono& o : base) cnt += fe_eq(o.coeff, base[i].coeff);
                if (cnt > best_n) best = i, best_n = cnt;
            }
            const Fe d = base[best].coeff, d_inv = Fr::inv(d);
            const Fe one_hat = Fr::mul(Fr::one(), c261);
            for (DMono& o : base) {
                o.coeff = Fr::mul(Fr::mul(o.coeff, d_inv), c261);  // (o / d) back in the 2^261 form
                o.coeff_is_one = fe_eq(o.coeff, one_hat) ? 1 : 0;
            }
            for (Fe& u : out.uc) u = Fr::mul(u, Fr::mul(d, Fr::inv(c261)));  // (d carries the 2^261 factor already)
            out.inner = std::move(base);
        }
    }
    if (!univariate) {
        out.uc.clear();
        for (uint32_t m = g.first; m < g.first + g.count; m++) out.inner.push_back(strip(monos[m], false));
    }
    for (const DMono& d : out.inner) {
        const uint32_t operands = d.n_factors + (d.coeff_is_one && d.n_factors ? 0u : 1u);
        out.cost += operands ? operands - 1 : 0;
    }
    if (!out.uc.empty() && !tabulated) out.cost += (uint32_t)out.uc.size();  // Horner in the kernel
    return out;
}

}  // namespace

extern "C" {

void zg_keccak256(const uint8_t* data, size_t len, uint8_t out[32]) { keccak256(data, len, out); }

size_t zg_prover_proof_size(const zg_prover* p) {
    if (!p) return 0;
    size_t points = p->A + 2 * p->NL + p->sets + p->NL + 1 + p->qpd;
    size_t scalars = p->advice_queries.size() + p->fixed_queries.size() + 1 + p->P + (p->sets ? 3 * p->sets - 1 : 0) +
                     5 * p->NL;
    size_t max_open = 2 + p->advice_queries.size() + p->fixed_queries.size();
    return 64 * (points + max_open) + 32 * scalars;
}

void zg_prover_destroy(zg_prover* p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    if (p->ctx->side) (void)hipStreamSynchronize(p->ctx->side->stream);
    for (void* q : p->owned) (void)hipFree(q);
    if (p->own_bases) {
        if (p->g) zg_bases_free(p->g);
        if (p->gl) zg_bases_free(p->gl);
    }
    if (p->ev) (void)hipEventDestroy(p->ev);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    if (p->pinned) (void)hipHostFree(p->pinned);
    delete p;
}

static int prover_create_impl(zg_ctx* ctx, const zg_circuit* cs, const zg_fr* fixed_values, const zg_fr* sigma_values,
                              const zg_g1_affine* g, const zg_g1_affine* g_lagrange, const zg_bases* shared_g,
                              const zg_bases* shared_gl, const zg_fr* vk_repr, zg_prover** out);

int zg_prover_create(zg_ctx* ctx, const zg_circuit* cs, const zg_fr* fixed_values, const zg_fr* sigma_values,
                     const zg_g1_affine* g, const zg_g1_affine* g_lagrange, const zg_fr* vk_repr, zg_prover** out) {
    ZG_REQUIRE(g && g_lagrange, ZG_ERR_INVALID_ARG, "zg_prover_create: null SRS");
    return prover_create_impl(ctx, cs, fixed_values, sigma_values, g, g_lagrange, nullptr, nullptr, vk_repr, out);
}

int zg_prover_create_shared(zg_ctx* ctx, const zg_circuit* cs, const zg_fr* fixed_values, const zg_fr* sigma_values,
                            const zg_bases* g, const zg_bases* g_lagrange, const zg_fr* vk_repr, zg_prover** out) {
    ZG_REQUIRE(g && g_lagrange, ZG_ERR_INVALID_ARG, "zg_prover_create_shared: null bases");
    return prover_create_impl(ctx, cs, fixed_values, sigma_values, nullptr, nullptr, g, g_lagrange, vk_repr, out);
}

static int prover_create_impl(zg_ctx* ctx, const zg_circuit* cs, const zg_fr* fixed_values, const zg_fr* sigma_values,
                              const zg_g1_affine* g, const zg_g1_affine* g_lagrange, const zg_bases* shared_g,
                              const zg_bases* shared_gl, const zg_fr* vk_repr, zg_prover** out) {
    ZG_REQUIRE(ctx && cs && vk_repr && out, ZG_ERR_INVALID_ARG, "zg_prover_create: null argument");
    ZG_REQUIRE(cs->n_fixed == 0 || fixed_values, ZG_ERR_INVALID_ARG, "zg_prover_create: fixed_values is null");
    ZG_REQUIRE(cs->n_perm_columns == 0 || sigma_values, ZG_ERR_INVALID_ARG, "zg_prover_create: sigma_values is null");
    ZG_REQUIRE(cs->cs_degree >= 3 && cs->cs_degree <= 9, ZG_ERR_UNSUPPORTED, "zg_prover_create: cs_degree %u", cs->cs_degree);
    ZG_REQUIRE(cs->k >= 4, ZG_ERR_UNSUPPORTED, "zg_prover_create: k=%u < 4", cs->k);
    ZG_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<zg_prover, void (*)(zg_prover*)> guard(new zg_prover(), zg_prover_destroy);
    zg_prover* p = guard.get();
    p->ctx = ctx;
    p->k = cs->k;
    p->n = 1u << cs->k;
    p->cs_degree = cs->cs_degree;
    p->bf = cs->blinding_factors;
    p->qpd = cs->cs_degree - 1;
    p->ext_k = cs->k;
    while ((1ull << p->ext_k) < (uint64_t)p->n * p->qpd) p->ext_k++;
    ZG_REQUIRE(p->ext_k <= 22, ZG_ERR_UNSUPPORTED, "zg_prover_create: extended domain 2^%u not built", p->ext_k);
    p->en = 1u << p->ext_k;
    ZG_REQUIRE(p->n > p->bf + 2, ZG_ERR_INVALID_ARG, "zg_prover_create: too few rows");
    p->usable = p->n - (p->bf + 1);
    p->F = cs->n_fixed; p->A = cs->n_advice; p->I = cs->n_instance; p->P = cs->n_perm_columns; p->NL = cs->n_lookups;
    p->chunk = cs->cs_degree - 2;
    p->sets = p->P ? (p->P + p->chunk - 1) / p->chunk : 0;
    p->advice_queries.assign(cs->advice_queries, cs->advice_queries + cs->n_advice_queries);
    p->fixed_queries.assign(cs->fixed_queries, cs->fixed_queries + cs->n_fixed_queries);
    p->vk_repr = to_fe(vk_repr);
    p->omega = host_domain_omega(p->k);
    p->omega_inv = Fr::inv(p->omega);
    p->ifft_div = Fr::inv(Fr::from_u64(p->n));
    const uint32_t n = p->n, en = p->en;
    hipStream_t st = ctx->stream;
    ZG_HIP(hipEventCreateWithFlags(&p->ev, hipEventDisableTiming));
    ZG_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    ZG_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    if (const char* e = getenv("ZG_EVALH9")) p->hat = atoi(e) != 0;
    if (p->use_side && !ctx->side) ZG_TRY(zg_ctx_create(ctx->device, &ctx->side));

    // ---- validate and upload the circuit tables
    for (uint32_t q = 0; q < cs->n_queries; q++) {
        const zg_query& qq = cs->queries[q];
        uint32_t lim = qq.kind == ZG_FIXED ? p->F : qq.kind == ZG_ADVICE ? p->A : qq.kind == ZG_INSTANCE ? p->I : 0;
        ZG_REQUIRE(qq.column < lim, ZG_ERR_INVALID_ARG, "zg_prover_create: query %u names column %u of kind %u", q,
                   qq.column, qq.kind);
    }
    std::vector<DMono> monos(cs->n_monomials);
    Fe one = Fr::one();
    for (uint32_t m = 0; m < cs->n_monomials; m++) {
        const zg_monomial& s = cs->monomials[m];
        ZG_REQUIRE(s.n_factors <= ZG_MAX_FACTORS, ZG_ERR_INVALID_ARG, "zg_prover_create: monomial %u has %u factors", m, s.n_factors);
        DMono d;
        memset(&d, 0, sizeof(d));
        memcpy(&d.coeff, &s.coeff, 32);
        d.n_factors = s.n_factors;
        d.coeff_is_one = fe_eq(d.coeff, one) ? 1 : 0;
        for (uint32_t f = 0; f < s.n_factors; f++) {
            ZG_REQUIRE(s.factors[f] < cs->n_queries, ZG_ERR_INVALID_ARG, "zg_prover_create: monomial %u factor out of range", m);
            d.factors[f] = s.factors[f];
        }
        monos[m] = d;
    }
    auto poly_ok = [&](const zg_poly& q) { return (uint64_t)q.first + q.count <= cs->n_monomials; };
    std::vector<DLookup> lks(cs->n_lookups);
    for (uint32_t l = 0; l < cs->n_lookups; l++) {
        const zg_lookup& s = cs->lookups[l];
        ZG_REQUIRE(s.width >= 1 && s.width <= ZG_MAX_LOOKUP_WIDTH, ZG_ERR_INVALID_ARG, "zg_prover_create: lookup %u width %u", l, s.width);
        lks[l].width = s.width;
        for (uint32_t e = 0; e < s.width; e++) {
            ZG_REQUIRE(poly_ok(s.inputs[e]) && poly_ok(s.tables[e]), ZG_ERR_INVALID_ARG, "zg_prover_create: lookup %u polynomial out of range", l);
            lks[l].inputs[e] = s.inputs[e];
            lks[l].tables[e] = s.tables[e];
        }
    }
    for (uint32_t gi = 0; gi < cs->n_gates; gi++)
        ZG_REQUIRE(poly_ok(cs->gates[gi]), ZG_ERR_INVALID_ARG, "zg_prover_create: gate %u out of range", gi);
    for (uint32_t c = 0; c < cs->n_perm_columns; c++) {
        const zg_query& qq = cs->perm_columns[c];
        uint32_t lim = qq.kind == ZG_FIXED ? p->F : qq.kind == ZG_ADVICE ? p->A : qq.kind == ZG_INSTANCE ? p->I : 0;
        ZG_REQUIRE(qq.column < lim, ZG_ERR_INVALID_ARG, "zg_prover_create: permutation column %u out of range", c);
    }
    zg_query* d_q; DMono* d_m; zg_poly* d_g; DLookup* d_l; zg_query* d_pc;
    ZG_TRY(dalloc(p, &d_q, cs->n_queries));
    ZG_TRY(dalloc(p, &d_m, cs->n_monomials));
    ZG_TRY(dalloc(p, &d_g, cs->n_gates));
    ZG_TRY(dalloc(p, &d_l, cs->n_lookups));
    ZG_TRY(dalloc(p, &d_pc, cs->n_perm_columns));
    if (cs->n_queries) ZG_HIP(hipMemcpyAsync(d_q, cs->queries, cs->n_queries * sizeof(zg_query), hipMemcpyHostToDevice, st));
    if (cs->n_monomials) ZG_HIP(hipMemcpyAsync(d_m, monos.data(), monos.size() * sizeof(DMono), hipMemcpyHostToDevice, st));
    if (cs->n_gates) ZG_HIP(hipMemcpyAsync(d_g, cs->gates, cs->n_gates * sizeof(zg_poly), hipMemcpyHostToDevice, st));
    if (cs->n_lookups) ZG_HIP(hipMemcpyAsync(d_l, lks.data(), lks.size() * sizeof(DLookup), hipMemcpyHostToDevice, st));
    if (cs->n_perm_columns) ZG_HIP(hipMemcpyAsync(d_pc, cs->perm_columns, cs->n_perm_columns * sizeof(zg_query), hipMemcpyHostToDevice, st));
    ZG_HIP(hipStreamSynchronize(st));  // the host vectors above go out of scope
    p->dc.queries = d_q; p->dc.monos = d_m; p->dc.gates = d_g; p->dc.lookups = d_l; p->dc.perm_cols = d_pc;
    p->dc.n_gates = cs->n_gates; p->dc.n_lookups = cs->n_lookups; p->dc.n_perm = p->P; p->dc.chunk = p->chunk;
    p->dc.n_sets = p->sets;
    if (p->hat) {  // evaluate_h's view: coefficients in the 2^261 form, gates factored by their common cell
        const Fe c261 = Fr9Params::c261_fe();
        for (auto& d : monos) d.coeff = Fr::mul(d.coeff, c261);
        std::vector<zg_poly> gates_hat(cs->n_gates);
        std::vector<uint32_t> common(cs->n_gates, 0xffffffffu);
        std::vector<zg_poly> gate_uni(cs->n_gates, zg_poly{0, 0});  // count 0: the factor is the cell itself
        std::vector<Fe> uni_coef;
        for (uint32_t gi = 0; gi < cs->n_gates; gi++) {
            const zg_poly g = cs->gates[gi];
            gates_hat[gi] = g;
            if (g.count < 2) continue;
            // Every query index present in all monomials of the gate is a candidate factor (q * (b^2 - b) has
            // two); the cheapest evaluation wins.  zero_g's gates are selector * (...), and a selector halo2
            // merged with others is a polynomial in its column: that one is tabulated with the proving key.
            GateFactor best;
            uint32_t f = 0xffffffffu;
            const DMono& first = monos[g.first];
            for (uint32_t a = 0; a < first.n_factors; a++) {
                const uint32_t cand = first.factors[a];
                if (a && cand == first.factors[a - 1]) continue;
                bool all = true;
                for (uint32_t m = g.first; m < g.first + g.count && all; m++)
                    all = std::find(monos[m].factors, monos[m].factors + monos[m].n_factors, cand) != monos[m].factors + monos[m].n_factors;
                if (!all) continue;
                GateFactor opt = factor_gate(monos, g, cand, c261, cs->queries[cand].kind == ZG_FIXED);
                if (f == 0xffffffffu || opt.cost < best.cost) best = std::move(opt), f = cand;
            }
            if (f == 0xffffffffu) continue;
            gates_hat[gi].first = (uint32_t)monos.size();
            gates_hat[gi].count = (uint32_t)best.inner.size();
            monos.insert(monos.end(), best.inner.begin(), best.inner.end());
            if (!best.uc.empty()) {
                gate_uni[gi].first = (uint32_t)uni_coef.size();
                gate_uni[gi].count = (uint32_t)best.uc.size();
                if (cs->queries[f].kind == ZG_FIXED) p->slab_jobs.push_back({gi, f, gate_uni[gi].first, gate_uni[gi].count});
                uni_coef.insert(uni_coef.end(), best.uc.begin(), best.uc.end());
            }
            common[gi] = f;
        }
        ZG_TRY(dalloc(p, &p->gate_uni, cs->n_gates ? cs->n_gates : 1));
        ZG_TRY(dalloc(p, &p->uni_coef, uni_coef.size() ? uni_coef.size() : 1));
        if (cs->n_gates) ZG_HIP(hipMemcpy(p->gate_uni, gate_uni.data(), cs->n_gates * sizeof(zg_poly), hipMemcpyHostToDevice));
        if (!uni_coef.empty()) ZG_HIP(hipMemcpy(p->uni_coef, uni_coef.data(), uni_coef.size() * sizeof(Fe), hipMemcpyHostToDevice));
        ZG_TRY(dalloc(p, &p->monos_hat, monos.size() ? monos.size() : 1));
        ZG_TRY(dalloc(p, &p->gates_hat, cs->n_gates ? cs->n_gates : 1));
        ZG_TRY(dalloc(p, &p->gate_common, cs->n_gates ? cs->n_gates : 1));
        if (!monos.empty()) ZG_HIP(hipMemcpy(p->monos_hat, monos.data(), monos.size() * sizeof(DMono), hipMemcpyHostToDevice));
        if (cs->n_gates) {
            ZG_HIP(hipMemcpy(p->gates_hat, gates_hat.data(), cs->n_gates * sizeof(zg_poly), hipMemcpyHostToDevice));
            ZG_HIP(hipMemcpy(p->gate_common, common.data(), cs->n_gates * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }

    // ---- SRS: upload + window tables, or tables shared with other provers on this device (read-only)
    if (shared_g) {
        ZG_REQUIRE(shared_g->ctx->device == ctx->device && shared_gl->ctx->device == ctx->device, ZG_ERR_INVALID_ARG,
                   "zg_prover_create_shared: bases live on another device");
        ZG_REQUIRE(shared_g->n == n && shared_gl->n == n && shared_g->c == shared_gl->c, ZG_ERR_INVALID_ARG,
                   "zg_prover_create_shared: bases do not match 2^k = %u points", n);
        p->g = const_cast<zg_bases*>(shared_g);
        p->gl = const_cast<zg_bases*>(shared_gl);
        p->own_bases = false;
    } else {
        WsScope ws(ctx);
        Affine* d = ws.get<Affine>(n);
        if (!d) return ZG_ERR_OOM;
        ZG_HIP(hipMemcpyAsync(d, g, (size_t)n * sizeof(Affine), hipMemcpyHostToDevice, st));
        ZG_TRY(bases_register_dev(ctx, d, n, 0, &p->g));
        ZG_HIP(hipMemcpyAsync(d, g_lagrange, (size_t)n * sizeof(Affine), hipMemcpyHostToDevice, st));
        ZG_TRY(bases_register_dev(ctx, d, n, 0, &p->gl));
    }
    static const bool run_form = !(getenv("ZG_MSM_RUNS") && atoi(getenv("ZG_MSM_RUNS")) == 0);  // A/B knob
    if (run_form && p->sets + p->NL > 0) ZG_TRY(bases_enable_runs(ctx, p->gl));

    // ---- slabs
    const uint32_t F = p->F, A = p->A, I = p->I, P = p->P, NL = p->NL, S = p->sets, Q = p->qpd;
    p->ix_fixed = 0; p->ix_sigma = F; p->ix_adv = F + P; p->ix_inst = p->ix_adv + A; p->ix_pz = p->ix_inst + I;
    p->ix_lz = p->ix_pz + S; p->ix_perm = p->ix_lz + NL; p->ix_random = p->ix_perm + 2 * NL;
    p->ix_hpiece = p->ix_random + 1; p->ix_hpoly = p->ix_hpiece + Q;
    p->n_polys = p->ix_hpoly + 1;
    ZG_TRY(dalloc(p, &p->polys, (size_t)p->n_polys * n));
    ZG_TRY(dalloc(p, &p->fixed_val, (size_t)F * n));
    ZG_TRY(dalloc(p, &p->sigma_val, (size_t)P * n));
    // parts of the extended domain
    {
        static const bool split_env = !(getenv("ZG_SPLIT_DOMAIN") && atoi(getenv("ZG_SPLIT_DOMAIN")) == 0);  // A/B knob
        uint32_t m1 = 1;
        while (m1 * 2 <= Q) m1 *= 2;
        const uint32_t m2 = Q - m1;
        const bool split = split_env && p->hat && m2 != 0 && (m2 & (m2 - 1)) == 0 && (m1 + m2) * n < en;
        auto log2u = [](uint32_t v) { uint32_t l = 0; while ((1u << l) < v) l++; return l; };
        // The single coset serves the latency configuration (a lone proof pays for the extra launches of the split
        // form in its h phase: 0.72 -> 0.93 ms), the split one the throughput configuration (-5 % ms/proof); both sets
        // of proving-key cosets are kept (+60 % of 0.2 GB per prover) and zg_prover_set_overlap picks.
        p->nparts = 1;
        p->dom[0].ek = p->ext_k; p->dom[0].en = en; p->dom[0].zpow = 1;
        if (split) {
            p->nparts = 3;
            p->dom[1].ek = p->k + log2u(m1); p->dom[1].en = n * m1; p->dom[1].zpow = 1;
            p->dom[2].ek = p->k + log2u(m2); p->dom[2].en = n * m2; p->dom[2].zpow = 2;
            ZG_TRY(dalloc(p, &p->split_tmp, (size_t)3 * p->dom[2].en));
        }
    }
    for (uint32_t di = 0; di < p->nparts; di++) {
        zg_prover::Dom& d = p->dom[di];
        ZG_TRY(dalloc(p, &d.fixed_cos, (size_t)F * d.en));
        ZG_TRY(dalloc(p, &d.sigma_cos, (size_t)P * d.en));
        ZG_TRY(dalloc(p, &d.l0, (size_t)d.en));
        ZG_TRY(dalloc(p, &d.llast, (size_t)d.en));
        ZG_TRY(dalloc(p, &d.lactive, (size_t)d.en));
        // (one block, in the order of the coefficient slab: advice, instance, permutation z, lookup z, a'/s' -- the
        //  split form transforms all of them in one batch)
        ZG_TRY(dalloc(p, &d.adv_cos, (size_t)(A + I + S + NL + 2 * NL) * d.en));
        d.inst_cos = d.adv_cos + (size_t)A * d.en;
        d.pz_cos = d.inst_cos + (size_t)I * d.en;
        d.lz_cos = d.pz_cos + (size_t)S * d.en;
        d.perm_cos = d.lz_cos + (size_t)NL * d.en;
        ZG_TRY(dalloc(p, &d.h, (size_t)d.en));
    }
    ZG_TRY(dalloc(p, &p->adv_val, (size_t)A * n));
    ZG_TRY(dalloc(p, &p->inst_val, (size_t)I * n));
    if (I) ZG_HIP(hipMemset(p->inst_val, 0, (size_t)I * n * 32));  // rows past the instance stay zero (prove refills only what it must)
    ZG_TRY(dalloc(p, &p->pin_c, (size_t)NL * n));
    ZG_TRY(dalloc(p, &p->ptab_c, (size_t)NL * n));
    ZG_TRY(dalloc(p, &p->cin, (size_t)2 * NL * n));  // compressed inputs, then compressed tables
    p->ctab = p->cin + (size_t)NL * n;
    ZG_TRY(dalloc(p, &p->perm, (size_t)(2 * NL + 1) * n));  // + the vanishing argument's random polynomial
    ZG_TRY(dalloc(p, &p->zs, (size_t)(S + NL + 1) * n));
    const uint32_t mb = S + NL;
    ZG_TRY(dalloc(p, &p->num, (size_t)mb * n));
    ZG_TRY(dalloc(p, &p->den, (size_t)mb * n));
    ZG_TRY(dalloc(p, &p->tmp, poly_grand_product_tmp_elems(n, mb)));
    ZG_TRY(dalloc(p, &p->raw, (size_t)2 * NL * n));
    ZG_TRY(dalloc(p, &p->sraw, (size_t)NL * n));
    ZG_TRY(dalloc(p, &p->sort_fe, (size_t)NL * n));
    ZG_TRY(dalloc(p, &p->sort_u32, (size_t)2 * NL * n + 3 * NL + 2));
    p->d_err = p->sort_u32 + (size_t)2 * NL * n + 2 * NL;  // behind permute_pairs' scratch: zeroed by the same fill
    ZG_REQUIRE(NL <= 60, ZG_ERR_UNSUPPORTED, "zg_prover_create: %u lookups", NL);
    const uint32_t max_points = 4 + (uint32_t)(p->advice_queries.size() + p->fixed_queries.size());
    ZG_TRY(dalloc(p, &p->pw, (size_t)max_points * n + max_points));
    const uint32_t max_evals = (uint32_t)(p->advice_queries.size() + p->fixed_queries.size()) + P + 3 * S + 5 * NL + 4;
    ZG_TRY(dalloc(p, &p->evals, max_evals));
    ZG_TRY(dalloc(p, &p->wpoly, (size_t)2 * max_points * n));
    ZG_TRY(dalloc(p, &p->xyzz, std::max<size_t>(std::max<size_t>(A, 2 * NL + 1), std::max<size_t>(S + NL + 1, std::max<size_t>(Q, max_points)))));
    ZG_TRY(dalloc(p, &p->d_idx, (size_t)4 * max_evals + 64 + (size_t)max_points * 512));
    ZG_TRY(dalloc(p, &p->ktmp, poly_kate_tmp_elems(n, max_points)));
    p->pinned_cap = 1u << 20;  // commitments (128 B each), evaluations, error flags
    ZG_HIP(hipHostMalloc(&p->pinned, p->pinned_cap, hipHostMallocDefault));

    // ---- keygen_pk's derived data: fixed / sigma polys + cosets, l_0 / l_last / l_active_row
    if (F) {
        ZG_HIP(hipMemcpyAsync(p->fixed_val, fixed_values, (size_t)F * n * 32, hipMemcpyHostToDevice, st));
        Fe* fp = p->polys + (size_t)p->ix_fixed * n;
        ZG_TRY(ntt_batch_to_dev(ctx, p->fixed_val, fp, n, F, p->k, p->omega_inv, &p->ifft_div));
        for (uint32_t di = 0; di < p->nparts; di++) {
            zg_prover::Dom& d = p->dom[di];
            ZG_TRY(coeff_to_coset_dev(ctx, fp, n, n, d.fixed_cos, d.en, F, d.ek, p->hat, d.zpow));
        }
    }
    if (p->hat) {
        // a gate factor that is a polynomial in a FIXED cell (a merged selector) does not depend on the
        // witness: its coset is part of the proving key here, as the unmerged selector's would have been
        std::vector<uint32_t> slab_of(cs->n_gates ? cs->n_gates : 1, 0xffffffffu);
        ZG_TRY(dalloc(p, &p->gate_slab, slab_of.size()));
        for (uint32_t di = 0; di < p->nparts; di++) {
            zg_prover::Dom& d = p->dom[di];
            ZG_TRY(dalloc(p, &d.gate_slabs, std::max<size_t>(1, p->slab_jobs.size() * (size_t)d.en)));
            for (size_t j = 0; j < p->slab_jobs.size(); j++) {
                const auto& job = p->slab_jobs[j];
                const zg_query q = cs->queries[job.query];
                ZG_TRY(poly_gate_factor(ctx, d.fixed_cos + (size_t)q.column * d.en, (uint32_t)(q.rotation * (int32_t)(d.en / n)),
                                        d.en, p->uni_coef + job.first, job.count, d.gate_slabs + j * (size_t)d.en));
                slab_of[job.gate] = (uint32_t)j;
            }
        }
        ZG_HIP(hipMemcpyAsync(p->gate_slab, slab_of.data(), slab_of.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        ZG_HIP(hipStreamSynchronize(st));
    }
    if (P) {
        ZG_HIP(hipMemcpyAsync(p->sigma_val, sigma_values, (size_t)P * n * 32, hipMemcpyHostToDevice, st));
        Fe* sp = p->polys + (size_t)p->ix_sigma * n;
        ZG_TRY(ntt_batch_to_dev(ctx, p->sigma_val, sp, n, P, p->k, p->omega_inv, &p->ifft_div));
        for (uint32_t di = 0; di < p->nparts; di++) {
            zg_prover::Dom& d = p->dom[di];
            ZG_TRY(coeff_to_coset_dev(ctx, sp, n, n, d.sigma_cos, d.en, P, d.ek, p->hat, d.zpow));
        }
    }
    {
        WsScope ws(ctx);
        Fe* t3 = ws.get<Fe>((size_t)3 * n);
        Fe* lblind = ws.get<Fe>(en);
        if (ws.failed) return ZG_ERR_OOM;
        ZG_TRY(poly_l_cosets_init(ctx, t3, t3 + n, t3 + 2 * n, n, p->bf));
        ZG_TRY(ntt_batch_dev(ctx, t3, n, 3, p->k, p->omega_inv, &p->ifft_div));
        for (uint32_t di = 0; di < p->nparts; di++) {
            zg_prover::Dom& d = p->dom[di];
            ZG_TRY(coeff_to_coset_dev(ctx, t3, n, n, d.l0, d.en, 1, d.ek, p->hat, d.zpow));
            ZG_TRY(coeff_to_coset_dev(ctx, t3 + n, n, n, d.llast, d.en, 1, d.ek, p->hat, d.zpow));
            ZG_TRY(coeff_to_coset_dev(ctx, t3 + 2 * n, n, n, lblind, d.en, 1, d.ek, p->hat, d.zpow));
            ZG_TRY(poly_lactive(ctx, d.lactive, d.llast, lblind, d.en, p->hat));
        }
        ZG_HIP(hipStreamSynchronize(st));
    }
    // t_evaluations of EvaluationDomain: ((shift * ext_omega^i)^n - 1)^-1, one period, per part of the domain
    ZG_TRY(get_twiddles(ctx, p->k, p->omega, &p->omega_tw));
    for (uint32_t di = 0; di < p->nparts; di++) {
        zg_prover::Dom& d = p->dom[di];
        uint32_t t_len = 1u << (d.ek - p->k);
        std::vector<Fe> te(t_len);
        Fe ext_omega = host_domain_omega(d.ek);
        const Fe shift = d.zpow == 1 ? fr_zeta() : Fr::sqr(fr_zeta());
        Fe cur = Fr::pow_u64(shift, n), step = Fr::pow_u64(ext_omega, n);
        for (uint32_t i = 0; i < t_len; i++) {
            te[i] = Fr::inv(Fr::sub(cur, Fr::one()));
            if (p->hat) te[i] = Fr::mul(te[i], Fr9Params::c261_fe());
            cur = Fr::mul(cur, step);
        }
        ZG_TRY(dalloc(p, &d.t_eval, t_len));
        ZG_HIP(hipMemcpy(d.t_eval, te.data(), t_len * sizeof(Fe), hipMemcpyHostToDevice));
        ZG_TRY(get_twiddles(ctx, d.ek, ext_omega, &d.ext_tw));
    }
    ZG_HIP(hipStreamSynchronize(st));
    *out = guard.release();
    return ZG_OK;
}

int zg_prover_prove_dev(zg_prover* p, void* d_advice, const zg_fr* instance, size_t instance_len, uint64_t seed,
                        uint8_t* proof, size_t proof_cap, size_t* proof_len) {
    ZG_REQUIRE(p && proof && proof_len && (d_advice || p->A == 0), ZG_ERR_INVALID_ARG, "zg_prover_prove: null argument");
    ZG_REQUIRE(p->I == 0 || instance || instance_len == 0, ZG_ERR_INVALID_ARG, "zg_prover_prove: instance is null");
    ZG_REQUIRE(instance_len <= p->usable, ZG_ERR_INVALID_ARG, "zg_prover_prove: instance too large (Error::InstanceTooLarge)");
    zg_ctx* ctx = p->ctx;
    ZG_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t n = p->n, k = p->k, ek = p->ext_k, bf = p->bf, usable = p->usable;
    const uint32_t F = p->F, A = p->A, I = p->I, P = p->P, NL = p->NL, S = p->sets, Q = p->qpd;
    (void)F;
    // extended-domain parts of this proof: the split pair in the throughput configuration, the single coset otherwise
    const bool split = p->nparts == 3 && !p->use_side;
    const uint32_t dlo = split ? 1u : 0u, dhi = split ? 3u : 1u;
    Fe* polys = p->polys;
    auto poly_at = [&](uint32_t ix) { return polys + (size_t)ix * n; };
    EvmTranscript tr;
    std::vector<Jac> pts;
    p->have_last = false;
    p->stage_off = PIN_STAGE;
    using clk = std::chrono::steady_clock;
    auto t_start = clk::now(), t_prev = t_start;
    auto lap = [&](int slot) {
        auto now = clk::now();
        p->phase_ms[slot] = std::chrono::duration<double, std::milli>(now - t_prev).count();
        t_prev = now;
    };

    // side stream: coefficient / coset forms of committed columns are computed there while the main
    // stream runs the commitment MSM (whose tail is a chain of dependent EC additions on a few CUs)
    // (p->use_side == false: everything stays on the main stream -- the throughput configuration, where
    // other proofs in flight fill the gaps and every extra HIP stream costs a hardware queue)
    zg_ctx* sx = p->use_side ? ctx->side : ctx;
    hipStream_t ss = sx->stream;
    auto fork = [&]() -> int {  // side stream continues after everything queued on the main stream so far
        if (!p->use_side) return ZG_OK;
        ZG_HIP(hipEventRecord(p->ev_fork, st));
        ZG_HIP(hipStreamWaitEvent(ss, p->ev_fork, 0));
        return ZG_OK;
    };
    auto join = [&]() -> int {  // main stream continues after everything queued on the side stream so far
        if (!p->use_side) return ZG_OK;
        ZG_HIP(hipEventRecord(p->ev_join, ss));
        ZG_HIP(hipStreamWaitEvent(st, p->ev_join, 0));
        return ZG_OK;
    };

    // ---- vk + instance values into the transcript; instance polynomial
    tr.common_scalar(p->vk_repr);
    // vanishing::Argument::commit's random polynomial depends on no challenge: generate it now and
    // commit it inside the permuted-lookup batch (coefficient basis `g` next to `g_lagrange` vectors)
    Fe* random_row = p->perm + (size_t)(2 * NL) * n;
    // (the same launch draws the blinding rows of the advice columns: commit_lagrange's input below)
    Fe* adv = reinterpret_cast<Fe*>(d_advice);
    ZG_TRY(poly_random_and_blind(ctx, random_row, poly_at(p->ix_random), n, seed, TAG_RANDOM_POLY, adv, n, A, usable, bf + 1,
                                 TAG_ADVICE_BLIND));
    if (I) {
        if (p->inst_filled > instance_len) ZG_HIP(hipMemsetAsync(p->inst_val, 0, (size_t)I * n * 32, st));  // (zeroed at create)
        p->inst_filled = instance_len;
        for (uint32_t c = 0; c < I; c++) {
            for (size_t i = 0; i < instance_len; i++) tr.common_scalar(to_fe(&instance[c * instance_len + i]));
            if (instance_len)
                ZG_TRY(h2d(p, p->inst_val + (size_t)c * n, instance + c * instance_len, instance_len * 32));
        }
    }

    // ---- advice: commit (Lagrange basis)
    ZG_TRY(fork());
    if (I) {
        ZG_TRY(ntt_batch_to_dev(sx, p->inst_val, poly_at(p->ix_inst), n, I, k, p->omega_inv, &p->ifft_div));
        for (uint32_t di = dlo; di < dhi && !split; di++)
            ZG_TRY(coeff_to_coset_dev(sx, poly_at(p->ix_inst), n, n, p->dom[di].inst_cos, p->dom[di].en, I, p->dom[di].ek, p->hat, p->dom[di].zpow));
    }
    if (A) {
        ZG_TRY(msm_batch_dev(ctx, p->gl, adv, n, A, n, p->xyzz));
        ZG_TRY(fetch_points(p, A, pts));
        ZG_TRY(ntt_batch_to_dev(sx, adv, poly_at(p->ix_adv), n, A, k, p->omega_inv, &p->ifft_div));
        for (uint32_t di = dlo; di < dhi && !split; di++)
            ZG_TRY(coeff_to_coset_dev(sx, poly_at(p->ix_adv), n, n, p->dom[di].adv_cos, p->dom[di].en, A, p->dom[di].ek, p->hat, p->dom[di].zpow));
        ZG_TRY(wait_points(p, A, pts));
        for (auto& q : pts) tr.write_point(q);
    }
    const Fe theta = tr.squeeze();
    lap(0);

    Cols base_cols;
    base_cols.fixed = p->fixed_val; base_cols.advice = adv; base_cols.instance = p->inst_val;
    base_cols.log_size = k; base_cols.rot_scale = 1;

    // ---- lookups: commit_permuted (+ the random polynomial's commitment)
    Jac random_commit;
    bool have_random = false;
    if (NL) {
        // permute_expression_pair on the device: canonical keys (written by the compression kernel itself, with
        // the sentinel padding), bitonic sort of inputs and tables, scan-based construction of s' (sort.hip).
        // raw rows [0,NL) = inputs -> a', [NL,2NL) = tables.
        ZG_TRY(poly_lookup_compress(ctx, p->dc, base_cols, theta, p->cin, p->ctab, n, p->raw, p->raw + (size_t)NL * n, usable));
        auto t_sort = clk::now();
        ZG_TRY(poly_sort_keys(ctx, p->raw, n, 2 * NL));
        ZG_TRY(poly_permute_pairs(ctx, p->raw, p->raw + (size_t)NL * n, p->sraw, n, usable, NL, p->sort_u32, p->sort_fe,
                                  p->d_err));
        // perm[2l] = a'_l, perm[2l+1] = s'_l (Montgomery form) on the usable rows, then the blinding tail
        // (blinding: a' rows get tag 2, s' rows tag 3, index = lookup * (bf+1) + j)
        ZG_TRY(poly_permuted_finish(ctx, p->raw, p->sraw, p->perm, n, usable, bf + 1, NL, seed, TAG_PERMUTED_INPUT,
                                    TAG_PERMUTED_TABLE));
        p->phase_ms[7] = std::chrono::duration<double, std::milli>(clk::now() - t_sort).count();
        ZG_TRY(fork());
        // (a' and s' are sorted: equal neighbours everywhere, so the run form leaves one entry per distinct value)
        const uint64_t sorted_runs = p->gl->run_table && 2 * NL < 64 ? (1ull << (2 * NL)) - 1ull : 0ull;
        ZG_TRY(msm_batch3_dev(ctx, p->gl, p->g, 2 * NL, p->perm, n, 2 * NL + 1, n, p->xyzz, sorted_runs));
        uint32_t* h_err = reinterpret_cast<uint32_t*>((char*)p->pinned + p->pinned_cap - 256);
        ZG_HIP(hipMemcpyAsync(h_err, p->d_err, NL * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        ZG_TRY(fetch_points(p, 2 * NL + 1, pts));
        ZG_TRY(ntt_batch_to_dev(sx, p->perm, poly_at(p->ix_perm), n, 2 * NL, k, p->omega_inv, &p->ifft_div));
        for (uint32_t di = dlo; di < dhi && !split; di++)
            ZG_TRY(coeff_to_coset_dev(sx, poly_at(p->ix_perm), n, n, p->dom[di].perm_cos, p->dom[di].en, 2 * NL, p->dom[di].ek, p->hat, p->dom[di].zpow));
        ZG_TRY(wait_points(p, 2 * NL + 1, pts));
        for (uint32_t l = 0; l < NL; l++)
            if (h_err[l]) {
                set_error("zg_prover_prove: lookup %u has an input outside its table (ConstraintSystemFailure)", l);
                (void)hipStreamSynchronize(ss);
                return ZG_ERR_CONSTRAINT;
            }
        for (uint32_t i = 0; i < 2 * NL; i++) tr.write_point(pts[i]);
        random_commit = pts[2 * NL];
        have_random = true;
    }
    const Fe beta = tr.squeeze();
    lap(1);
    const Fe gamma = tr.squeeze();

    // ---- permutation products (sets chained through z[n - bf - 1]) and lookup products
    Fe* pz = p->zs;
    if (S) ZG_TRY(poly_perm_terms(ctx, p->dc, base_cols, p->sigma_val, p->omega_tw, beta, gamma, p->num, p->den, n));
    // (a'_l / s'_l are interleaved in `perm`: two views with a stride of two columns)
    ZG_TRY(poly_lookup_terms(ctx, p->cin, p->ctab, p->perm, p->perm + n, (size_t)2 * n, beta, gamma, p->num + (size_t)S * n,
                             p->den + (size_t)S * n, n, NL));
    if (S + NL) {
        // all running products of the proof in one scan sequence; the S permutation sets are chained
        // through row n - bf - 1, the lookup products start from one
        ZG_TRY(poly_grand_product(ctx, p->num, p->den, nullptr, p->zs, p->tmp, n, S + NL, S, n - bf - 1));
        ZG_TRY(poly_blind_rows2(ctx, pz, n, S, TAG_PERM_Z, NL, TAG_LOOKUP_Z, n - bf, bf, seed));  // (lz follows pz)
        // The products stay constant wherever a row changes nothing (every padding row of the circuit): they are
        // committed in the run form, sum_i (z_i - z_{i+1}) Q_i over the running sums Q of g_lagrange.
        const uint64_t z_runs = p->gl->run_table && S + NL < 64 ? (1ull << (S + NL)) - 1ull : 0ull;
        ZG_TRY(fork());
        if (have_random) {
            ZG_TRY(msm_batch3_dev(ctx, p->gl, nullptr, S + NL, p->zs, n, S + NL, n, p->xyzz, z_runs));
        } else {  // no lookups: the random polynomial rides here instead (row S+NL of zs)
            ZG_HIP(hipMemcpyAsync(p->zs + (size_t)(S + NL) * n, random_row, (size_t)n * 32, hipMemcpyDeviceToDevice, st));
            ZG_TRY(msm_batch3_dev(ctx, p->gl, p->g, S + NL, p->zs, n, S + NL + 1, n, p->xyzz, z_runs));
        }
        const uint32_t npts = S + NL + (have_random ? 0 : 1);
        ZG_TRY(fetch_points(p, npts, pts));
        ZG_TRY(ntt_batch_to_dev(sx, p->zs, poly_at(p->ix_pz), n, S + NL, k, p->omega_inv, &p->ifft_div));
        for (uint32_t di = dlo; di < dhi && !split; di++) {
            const zg_prover::Dom& d = p->dom[di];
            if (S) ZG_TRY(coeff_to_coset_dev(sx, poly_at(p->ix_pz), n, n, d.pz_cos, d.en, S, d.ek, p->hat, d.zpow));
            if (NL) ZG_TRY(coeff_to_coset_dev(sx, poly_at(p->ix_lz), n, n, d.lz_cos, d.en, NL, d.ek, p->hat, d.zpow));
        }
        ZG_TRY(wait_points(p, npts, pts));
        for (uint32_t i = 0; i < S + NL; i++) tr.write_point(pts[i]);
        if (!have_random) {
            random_commit = pts[S + NL];
            have_random = true;
        }
    }
    if (!have_random) {  // neither lookups nor permutation: commit the random polynomial on its own
        ZG_TRY(msm_batch_dev(ctx, p->g, random_row, n, 1, n, p->xyzz));
        ZG_TRY(fetch_points(p, 1, pts));
        ZG_TRY(wait_points(p, 1, pts));
        random_commit = pts[0];
    }
    tr.write_point(random_commit);
    ZG_TRY(join());  // evaluate_h reads every coset the side stream produced
    const Fe y = tr.squeeze();
    lap(2);

    // (split form: nothing overlaps in the throughput configuration, so every witness polynomial goes to both cosets
    //  here, in one batch per coset, instead of phase by phase)
    if (split) {
        ZG_REQUIRE(p->ix_inst == p->ix_adv + A && p->ix_pz == p->ix_inst + I && p->ix_lz == p->ix_pz + S && p->ix_perm == p->ix_lz + NL,
                   ZG_ERR_INVALID_ARG, "zg_prover_prove: coefficient slab out of order");
        for (uint32_t di = dlo; di < dhi; di++)
            ZG_TRY(coeff_to_coset_dev(ctx, poly_at(p->ix_adv), n, n, p->dom[di].adv_cos, p->dom[di].en, A + I + S + NL + 2 * NL,
                                      p->dom[di].ek, p->hat, p->dom[di].zpow));
    }
    // ---- evaluate_h (+ division by X^n - 1) on every part of the extended domain, back to coefficients, h pieces
    for (uint32_t di = dlo; di < dhi; di++) {
        const zg_prover::Dom& d = p->dom[di];
        EvalHArgs a;
        memset(&a, 0, sizeof(a));
        a.c = p->dc;
        a.cols.fixed = d.fixed_cos; a.cols.advice = d.adv_cos; a.cols.instance = d.inst_cos;
        a.cols.log_size = d.ek; a.cols.rot_scale = (int32_t)(d.en / n);
        a.sigma_cos = d.sigma_cos; a.pz_cos = d.pz_cos; a.lz_cos = d.lz_cos;
        // perm_cos is interleaved, [2l] = a'_l and [2l+1] = s'_l: two views with a stride of two slabs
        a.pin_cos = d.perm_cos; a.ptab_cos = d.perm_cos + d.en; a.perm_stride = (size_t)2 * d.en;
        a.l0 = d.l0; a.llast = d.llast; a.lactive = d.lactive;
        a.ext_tw = p->hat ? d.ext_tw + d.en : d.ext_tw;  // (the twiddle table's second half is the 2^261 form)
        a.t_eval = d.t_eval; a.t_mask = (1u << (d.ek - k)) - 1;
        a.last_rot = -(int32_t)(bf + 1);
        a.y = y; a.beta = beta; a.gamma = gamma; a.theta = theta;
        a.delta_start = Fr::mul(beta, d.zpow == 1 ? fr_zeta() : Fr::sqr(fr_zeta())); a.delta = fr_delta();  // beta * coset shift
        a.hat = p->hat;
        a.monos_hat = p->monos_hat;
        a.gates_hat = p->gates_hat;
        a.gate_common = p->gate_common;
        a.gate_uni = p->gate_uni;
        a.uni_coef = p->uni_coef;
        a.gate_slab = p->gate_slab;
        a.gate_slabs = d.gate_slabs;
        if (p->hat) {
            const Fe c261 = Fr9Params::c261_fe();
            for (Fe* cst : {&a.y, &a.beta, &a.gamma, &a.theta, &a.delta_start, &a.delta}) *cst = Fr::mul(*cst, c261);
        }
        a.h = d.h;
        ZG_TRY(poly_evaluate_h(ctx, a, d.en));
    }
    p->have_last = true;
    p->last_split = split;
    if (!split) {
        ZG_TRY(extended_to_coeff_dev(ctx, p->dom[0].h, k, ek, (size_t)Q * n, poly_at(p->ix_hpiece), p->hat));
    } else {
        // h = A + (X^L1 - c1) B:  A (degree < L1) from the first coset, where X^L1 = c1 = shift1^L1;  B (degree < L2)
        // from the second, where X^L1 = c2 and X^L2 = e are constants too:  B = (h - A) / (c2 - c1) there, with A
        // folded modulo X^L2 - e before it is evaluated on those L2 points.
        const zg_prover::Dom &d1 = p->dom[1], &d2 = p->dom[2];
        const uint32_t L1 = d1.en, L2 = d2.en;
        const Fe zeta = fr_zeta(), zeta2 = Fr::sqr(zeta);
        const Fe c1 = Fr::pow_u64(zeta, L1), c2 = Fr::pow_u64(zeta2, L1), e = Fr::pow_u64(zeta2, L2);
        Fe* hp = poly_at(p->ix_hpiece);
        Fe *fold = p->split_tmp, *a2 = fold + L2, *bc = a2 + L2;
        ZG_TRY(coset_to_coeff_dev(ctx, d1.h, d1.ek, L1, hp, p->hat, 1));               // A, in place of the low pieces
        ZG_TRY(poly_fold(ctx, hp, L2, L1 / L2, e, fold));                               // A mod (X^L2 - e)
        ZG_TRY(coeff_to_coset_dev(ctx, fold, L2, L2, a2, L2, 1, d2.ek, false, 2));      // A on the second coset
        const Fe unhat = p->hat ? Fr::inv(Fr::from_u64(32)) : Fr::one();
        ZG_TRY(poly_diff_scale(ctx, d2.h, unhat, a2, Fr::inv(Fr::sub(c2, c1)), a2, L2));  // B on the second coset
        ZG_TRY(coset_to_coeff_dev(ctx, a2, d2.ek, L2, bc, false, 2));                   // B
        ZG_TRY(poly_split_combine(ctx, hp, bc, L2, c1, L1));                            // h = A - c1 B + X^L1 B
    }
    ZG_TRY(msm_batch_dev(ctx, p->g, poly_at(p->ix_hpiece), n, Q, n, p->xyzz));
    ZG_TRY(fetch_points(p, Q, pts));
    ZG_TRY(wait_points(p, Q, pts));
    for (auto& q : pts) tr.write_point(q);
    const Fe x = tr.squeeze();
    lap(3);
    const Fe xn = Fr::pow_u64(x, n);

    // ---- evaluations
    // distinct opening points, in any order (the powers table is indexed by slot)
    std::vector<int32_t> rots = {0, 1, -1, -(int32_t)(bf + 1)};
    auto rot_slot = [&](int32_t r) -> uint32_t {
        for (size_t i = 0; i < rots.size(); i++)
            if (rots[i] == r) return (uint32_t)i;
        rots.push_back(r);
        return (uint32_t)rots.size() - 1;
    };
    struct Q1 { uint32_t poly, slot; };
    std::vector<Q1> evq;  // evaluations in transcript order, then h_poly at x
    for (auto& q : p->advice_queries) evq.push_back({p->ix_adv + q.column, rot_slot(q.rotation)});
    const size_t e_fixed = evq.size();
    for (auto& q : p->fixed_queries) evq.push_back({p->ix_fixed + q.column, rot_slot(q.rotation)});
    const size_t e_random = evq.size();
    evq.push_back({p->ix_random, 0});
    const size_t e_sigma = evq.size();
    for (uint32_t c = 0; c < P; c++) evq.push_back({p->ix_sigma + c, 0});
    const size_t e_pz = evq.size();
    for (uint32_t s = 0; s < S; s++) {
        evq.push_back({p->ix_pz + s, 0});
        evq.push_back({p->ix_pz + s, 1});
        if (s + 1 < S) evq.push_back({p->ix_pz + s, 3});
    }
    const size_t e_lk = evq.size();
    for (uint32_t l = 0; l < NL; l++) {
        evq.push_back({p->ix_lz + l, 0});            // z(x)
        evq.push_back({p->ix_lz + l, 1});            // z(omega x)
        evq.push_back({p->ix_perm + 2 * l, 0});      // a'(x)
        evq.push_back({p->ix_perm + 2 * l, 2});      // a'(omega^-1 x)
        evq.push_back({p->ix_perm + 2 * l + 1, 0});  // s'(x)
    }
    const size_t e_written = evq.size();
    evq.push_back({p->ix_hpoly, 0});
    const size_t e_h = e_written;

    std::vector<Fe> points(rots.size());
    for (size_t i = 0; i < rots.size(); i++) points[i] = rotate_omega(p, x, rots[i]);
    // vanishing.evaluate: h(X) = sum_i xn^i h_i(X)
    {
        std::vector<uint32_t> list(Q);
        for (uint32_t i = 0; i < Q; i++) list[i] = p->ix_hpiece + (Q - 1 - i);
        uint32_t* dl = p->d_idx + (size_t)4 * (p->advice_queries.size() + p->fixed_queries.size() + P + 3 * S + 5 * NL + 4);
        ZG_TRY(h2d_list(p, dl, list));
        ZG_TRY(poly_horner_combine(ctx, polys, n, dl, Q, xn, fe_zero(), poly_at(p->ix_hpoly), n));
    }
    {
        const Fe* pts_pinned = (const Fe*)stage(p, points.data(), points.size() * sizeof(Fe));
        ZG_TRY(poly_powers(ctx, pts_pinned ? pts_pinned : points.data(), (uint32_t)points.size(), n, p->pw));
    }
    std::vector<uint32_t> idx(2 * evq.size());
    for (size_t i = 0; i < evq.size(); i++) {
        idx[i] = evq[i].poly;
        idx[evq.size() + i] = evq[i].slot;
    }
    ZG_TRY(h2d_list(p, p->d_idx, idx));
    ZG_TRY(poly_dot(ctx, polys, n, n, p->d_idx, p->d_idx + evq.size(), p->pw, (uint32_t)evq.size(), p->evals));
    ZG_REQUIRE(evq.size() * sizeof(Fe) <= PIN_STAGE - PIN_EVALS, ZG_ERR_UNSUPPORTED, "zg_prover_prove: too many evaluations");
    const Fe* ev = reinterpret_cast<const Fe*>((char*)p->pinned + PIN_EVALS);
    ZG_HIP(hipMemcpyAsync((void*)ev, p->evals, evq.size() * sizeof(Fe), hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    for (size_t i = 0; i < e_written; i++) tr.write_scalar(ev[i]);

    // ---- opening queries in create_proof's order: (poly, point slot, eval)
    struct OQ { uint32_t poly, slot; Fe eval; };
    std::vector<OQ> oq;
    for (size_t i = 0; i < e_fixed; i++) oq.push_back({evq[i].poly, evq[i].slot, ev[i]});
    {
        size_t e = e_pz;
        std::vector<size_t> e_last(S, 0), e_cur(S, 0), e_next(S, 0);
        for (uint32_t s = 0; s < S; s++) {
            e_cur[s] = e++;
            e_next[s] = e++;
            if (s + 1 < S) e_last[s] = e++;
        }
        for (uint32_t s = 0; s < S; s++) {
            oq.push_back({p->ix_pz + s, 0, ev[e_cur[s]]});
            oq.push_back({p->ix_pz + s, 1, ev[e_next[s]]});
        }
        for (uint32_t s = S; s-- > 0;) {
            if (s + 1 == S) continue;
            oq.push_back({p->ix_pz + s, 3, ev[e_last[s]]});
        }
    }
    for (uint32_t l = 0; l < NL; l++) {
        const Fe* e5 = &ev[e_lk + 5 * l];
        oq.push_back({p->ix_lz + l, 0, e5[0]});
        oq.push_back({p->ix_perm + 2 * l, 0, e5[2]});
        oq.push_back({p->ix_perm + 2 * l + 1, 0, e5[4]});
        oq.push_back({p->ix_perm + 2 * l, 2, e5[3]});
        oq.push_back({p->ix_lz + l, 1, e5[1]});
    }
    for (size_t i = e_fixed; i < e_random; i++) oq.push_back({evq[i].poly, evq[i].slot, ev[i]});
    for (uint32_t c = 0; c < P; c++) oq.push_back({p->ix_sigma + c, 0, ev[e_sigma + c]});
    oq.push_back({p->ix_hpoly, 0, ev[e_h]});
    oq.push_back({p->ix_random, 0, ev[e_random]});

    // ---- ProverGWC::create_proof
    const Fe v = tr.squeeze();
    lap(4);
    {
        std::vector<char> done(oq.size(), 0);
        uint32_t npts = 0;
        std::vector<uint32_t> lists, counts;  // list of point set s at lists[s * 512 ..]
        std::vector<Fe> open_points, eval_batches;
        for (size_t first = 0; first < oq.size(); first++) {
            if (done[first]) continue;
            const uint32_t slot = oq[first].slot;
            lists.resize((size_t)(npts + 1) * 512, 0);
            uint32_t cnt = 0;
            Fe eval_batch = fe_zero();
            for (size_t j = first; j < oq.size(); j++) {
                if (done[j] || oq[j].slot != slot) continue;
                done[j] = 1;
                ZG_REQUIRE(cnt < 512, ZG_ERR_UNSUPPORTED, "zg_prover_prove: more than 512 polynomials opened at one point");
                lists[(size_t)npts * 512 + cnt++] = oq[j].poly;
                eval_batch = Fr::add(Fr::mul(eval_batch, v), oq[j].eval);
            }
            counts.push_back(cnt);
            eval_batches.push_back(eval_batch);
            open_points.push_back(points[slot]);
            npts++;
        }
        // poly_batch of every point set in one launch: set s -> wpoly[2s]
        // (their own region of d_idx, behind the evaluation lists: nothing else writes there between proofs)
        uint32_t* d_lists = p->d_idx + (size_t)4 * (p->advice_queries.size() + p->fixed_queries.size() + P + 3 * S + 5 * NL + 4) + 64;
        ZG_TRY(h2d_list(p, d_lists, lists));
        for (uint32_t s0 = 0; s0 < npts; s0 += HC_MAX_SETS) {
            const uint32_t m = std::min<uint32_t>(HC_MAX_SETS, npts - s0);
            ZG_TRY(poly_horner_combine_sets(ctx, polys, n, d_lists + (size_t)s0 * 512, 512, counts.data() + s0,
                                            eval_batches.data() + s0, m, v, p->wpoly + (size_t)(2 * s0) * n, (size_t)2 * n, n));
        }
        // one batched kate_division: poly j at wpoly[2j], quotient at wpoly[2j+1]
        const Fe* op_pinned = (const Fe*)stage(p, open_points.data(), open_points.size() * sizeof(Fe));
        ZG_TRY(poly_kate_division(ctx, p->wpoly, (size_t)2 * n, op_pinned ? op_pinned : open_points.data(), p->wpoly + n,
                                  (size_t)2 * n, p->ktmp, n, npts));
        // the witness polynomials sit at odd slots: stride 2n
        ZG_TRY(msm_batch_dev(ctx, p->g, p->wpoly + n, (size_t)2 * n, npts, n, p->xyzz));
        ZG_TRY(fetch_points(p, npts, pts));
        ZG_TRY(wait_points(p, npts, pts));
        for (auto& q : pts) tr.write_point(q);
    }
    ZG_REQUIRE(!tr.failed, ZG_ERR_INVALID_ARG,
               "zg_prover_prove: a commitment is the identity point; EvmTranscript cannot absorb it");
    ZG_REQUIRE(tr.stream.size() <= proof_cap, ZG_ERR_INVALID_ARG, "zg_prover_prove: proof buffer too small (%zu > %zu)",
               tr.stream.size(), proof_cap);
    memcpy(proof, tr.stream.data(), tr.stream.size());
    *proof_len = tr.stream.size();
    lap(5);
    p->phase_ms[6] = std::chrono::duration<double, std::milli>(clk::now() - t_start).count();
    return ZG_OK;
}

int zg_prover_prove(zg_prover* p, const zg_fr* advice, const zg_fr* instance, size_t instance_len, uint64_t seed,
                    uint8_t* proof, size_t proof_cap, size_t* proof_len) {
    ZG_REQUIRE(p && (advice || p->A == 0), ZG_ERR_INVALID_ARG, "zg_prover_prove: null argument");
    ZG_HIP(hipSetDevice(p->ctx->device));
    if (p->A)
        ZG_HIP(hipMemcpyAsync(p->adv_val, advice, (size_t)p->A * p->n * 32, hipMemcpyHostToDevice, p->ctx->stream));
    return zg_prover_prove_dev(p, p->adv_val, instance, instance_len, seed, proof, proof_cap, proof_len);
}

int zg_prover_set_overlap(zg_prover* p, int enable) {
    ZG_REQUIRE(p, ZG_ERR_INVALID_ARG, "zg_prover_set_overlap: null prover");
    if (enable && !p->ctx->side) {
        ZG_HIP(hipSetDevice(p->ctx->device));
        ZG_TRY(zg_ctx_create(p->ctx->device, &p->ctx->side));
        p->ctx->side->profiling = p->ctx->profiling;
        p->ctx->side->prof_filter = p->ctx->prof_filter;
    }
    p->use_side = enable != 0;
    p->ctx->msm_pair = enable != 0;  // latency configuration: two lanes per addition in the MSM reduction
    return ZG_OK;
}

int zg_prover_phase_ms(const zg_prover* p, double* out, size_t cap) {
    ZG_REQUIRE(p && out, ZG_ERR_INVALID_ARG, "zg_prover_phase_ms: null argument");
    for (size_t i = 0; i < cap && i < 8; i++) out[i] = p->phase_ms[i];
    return ZG_OK;
}

int zg_prover_fetch(zg_prover* p, uint32_t what, uint32_t index, zg_fr* out, size_t cap_elems) {
    ZG_REQUIRE(p && out, ZG_ERR_INVALID_ARG, "zg_prover_fetch: null argument");
    ZG_REQUIRE(p->have_last, ZG_ERR_INVALID_ARG, "zg_prover_fetch: no proof has been produced yet");
    const Fe* src = nullptr;
    size_t count = 0;
    const size_t n = p->n;
    switch (what) {
        case 0: src = p->dom[0].h; count = p->en; break;
        case 1: ZG_REQUIRE(index < p->sets, ZG_ERR_INVALID_ARG, "zg_prover_fetch: set %u", index);
                src = p->zs + (size_t)index * n; count = n; break;
        case 2: ZG_REQUIRE(index < p->NL, ZG_ERR_INVALID_ARG, "zg_prover_fetch: lookup %u", index);
                src = p->zs + (size_t)(p->sets + index) * n; count = n; break;
        case 3: ZG_REQUIRE(index < p->NL, ZG_ERR_INVALID_ARG, "zg_prover_fetch: lookup %u", index);
                src = p->perm + (size_t)(2 * index) * n; count = n; break;
        case 4: ZG_REQUIRE(index < p->NL, ZG_ERR_INVALID_ARG, "zg_prover_fetch: lookup %u", index);
                src = p->perm + (size_t)(2 * index + 1) * n; count = n; break;
        case 5: src = p->polys + (size_t)p->ix_hpiece * n; count = (size_t)p->qpd * n; break;
        default: ZG_REQUIRE(false, ZG_ERR_INVALID_ARG, "zg_prover_fetch: unknown item %u", what);
    }
    ZG_REQUIRE(cap_elems >= count, ZG_ERR_INVALID_ARG, "zg_prover_fetch: need %zu elements", count);
    ZG_HIP(hipSetDevice(p->ctx->device));
    if (what == 0 && p->last_split) {  // split domain: h on EvaluationDomain's coset, from its coefficients
        WsScope ws(p->ctx);
        Fe* tmp = ws.get<Fe>(count);
        if (!tmp) return ZG_ERR_OOM;
        const Fe* hp = p->polys + (size_t)p->ix_hpiece * n;
        ZG_TRY(coeff_to_coset_dev(p->ctx, hp, (size_t)p->qpd * n, (uint32_t)(p->qpd * n), tmp, count, 1, p->ext_k, false, 1));
        ZG_HIP(hipStreamSynchronize(p->ctx->stream));
        ZG_HIP(hipMemcpy(out, tmp, count * 32, hipMemcpyDeviceToHost));
        return ZG_OK;
    }
    if (what == 0 && p->hat) {  // h on the coset is kept as x * 2^261: hand back the library form
        WsScope ws(p->ctx);
        Fe* tmp = ws.get<Fe>(count);
        if (!tmp) return ZG_ERR_OOM;
        ZG_TRY(poly_scale(p->ctx, src, tmp, count, Fr::inv(Fr::from_u64(32))));
        ZG_HIP(hipStreamSynchronize(p->ctx->stream));
        ZG_HIP(hipMemcpy(out, tmp, count * 32, hipMemcpyDeviceToHost));
        return ZG_OK;
    }
    ZG_HIP(hipMemcpy(out, src, count * 32, hipMemcpyDeviceToHost));
    return ZG_OK;
}

// ---- stand-alone building blocks (tests) ----
int zg_grand_product_dev(zg_ctx* ctx, const void* d_num, const void* d_den, const zg_fr* z0, size_t n, void* d_z) {
    ZG_REQUIRE(ctx && d_num && d_den && d_z && z0, ZG_ERR_INVALID_ARG, "zg_grand_product_dev: null argument");
    ZG_REQUIRE(n < (1u << 28), ZG_ERR_UNSUPPORTED, "zg_grand_product_dev: n too large");
    ZG_HIP(hipSetDevice(ctx->device));
    WsScope ws(ctx);
    Fe* tmp = ws.get<Fe>(poly_grand_product_tmp_elems((uint32_t)n, 1) + 1);
    if (ws.failed) return ZG_ERR_OOM;
    Fe* z0d = tmp + poly_grand_product_tmp_elems((uint32_t)n, 1);
    ZG_HIP(hipMemcpyAsync(z0d, z0, 32, hipMemcpyHostToDevice, ctx->stream));
    ZG_TRY(poly_grand_product(ctx, (const Fe*)d_num, (const Fe*)d_den, z0d, (Fe*)d_z, tmp, (uint32_t)n, 1, 0, 0));
    ZG_HIP(hipStreamSynchronize(ctx->stream));
    return ZG_OK;
}

int zg_eval_polys_dev(zg_ctx* ctx, const void* d_polys, size_t stride_elems, size_t n, const uint32_t* poly_index,
                      const zg_fr* points, size_t count, zg_fr* out) {
    ZG_REQUIRE(ctx && d_polys && poly_index && points && out, ZG_ERR_INVALID_ARG, "zg_eval_polys_dev: null argument");
    if (!count) return ZG_OK;
    ZG_HIP(hipSetDevice(ctx->device));
    WsScope ws(ctx);
    // every pair gets its own powers row (callers with shared points should use the prover)
    Fe* pw = ws.get<Fe>(count * n + count);
    uint32_t* di = ws.get<uint32_t>(2 * count);
    Fe* de = ws.get<Fe>(count);
    if (ws.failed) return ZG_ERR_OOM;
    std::vector<uint32_t> idx(2 * count);
    for (size_t i = 0; i < count; i++) {
        idx[i] = poly_index[i];
        idx[count + i] = (uint32_t)i;
    }
    ZG_TRY(poly_powers(ctx, (const Fe*)points, (uint32_t)count, (uint32_t)n, pw));
    ZG_HIP(hipMemcpyAsync(di, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    ZG_TRY(poly_dot(ctx, (const Fe*)d_polys, stride_elems, (uint32_t)n, di, di + count, pw, (uint32_t)count, de));
    ZG_HIP(hipMemcpyAsync(out, de, count * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(hipStreamSynchronize(ctx->stream));
    return ZG_OK;
}

int zg_kate_division_dev(zg_ctx* ctx, const void* d_a, size_t n, const zg_fr* z, void* d_q) {
    ZG_REQUIRE(ctx && d_a && z && d_q && n >= 1, ZG_ERR_INVALID_ARG, "zg_kate_division_dev: bad argument");
    ZG_HIP(hipSetDevice(ctx->device));
    ZG_REQUIRE(n < (1u << 28), ZG_ERR_UNSUPPORTED, "zg_kate_division_dev: n too large");
    WsScope ws(ctx);
    Fe* tmp = ws.get<Fe>(poly_kate_tmp_elems((uint32_t)n, 1));
    if (ws.failed) return ZG_ERR_OOM;
    Fe zz = to_fe(z);
    ZG_TRY(poly_kate_division(ctx, (const Fe*)d_a, n, &zz, (Fe*)d_q, n, tmp, (uint32_t)n, 1));
    ZG_HIP(hipStreamSynchronize(ctx->stream));
    return ZG_OK;
}

}  // extern "C"
