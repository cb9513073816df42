// create_proof for one circuit instance on one MI355X -- replaces
// halo2_proofs::plonk::create_proof::<KZGCommitmentScheme<Bn256>, ProverGWC, _, _, EvmTranscript, _>
// as Wnn::proof calls it (/root/reference/src/wnn.rs:232-262; upstream v2023_04_20 src/plonk/prover.rs).
//
// Proofs are made in lock-step batches: one launch sequence serves `nb` circuit instances at once (a single proof
// is the batch of one), each with its own Fiat-Shamir transcript on the host.
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
#include <dlfcn.h>
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

// ------------------------------------------------------------------ proving key on the device
// What keygen_pk derives, resident in HBM, shared (read-only) by every prover forked from the one that built it.
struct PkDev {
    int device = 0;
    uint32_t k = 0, ext_k = 0, cs_degree = 0, bf = 0, qpd = 0;
    uint32_t n = 0, en = 0, usable = 0;
    uint32_t F = 0, A = 0, I = 0, P = 0, NL = 0, sets = 0, chunk = 0;
    std::vector<zg_query> advice_queries, fixed_queries;
    DevCircuit dc{};
    std::vector<void*> owned;  // device allocations freed with the key
    // evaluate_h on nine 29-bit limbs: the coset slabs, l-polynomials, t_eval and the monomial coefficients it
    // reads are kept in the 2^261 Montgomery form (x * 2^5 of the library form); ZG_EVALH9=0 turns it off
    bool hat = true;
    bool grouped = true;  // the terms after the gates are weighted by powers of y and summed per l-polynomial (ZG_EVALH_GROUPED)
    DMono* monos_hat = nullptr;
    zg_poly* gates_hat = nullptr;
    uint32_t* gate_common = nullptr;
    zg_poly* gate_uni = nullptr;
    Fe* uni_coef = nullptr;
    uint32_t* gate_slab = nullptr;  // per gate: index of its U(fixed cell) coset in gate_slabs, or 0xffffffff
    struct SlabJob { uint32_t gate, query, first, count; };
    std::vector<SlabJob> slab_jobs;  // filled when the gates are factored, run once the fixed cosets exist
    Fe vk_repr{};
    Fe omega{}, omega_inv{}, ifft_div{};
    Fe *fixed_val = nullptr, *sigma_val = nullptr, *omega_tw = nullptr;
    Fe* sh_polys = nullptr;  // coefficient forms [F + P][n]: fixed, then sigma
    // The extended domain evaluate_h works on.  Either EvaluationDomain's own coset zeta * <omega_(2^ext_k)> (8n points
    // for degree 6), or -- split -- two cosets that together hold just the (degree - 1) * n points the quotient needs:
    // zeta * <omega_(m1 n)> and zeta^2 * <omega_(m2 n)>, m1 + m2 = degree - 1 (4n + n).  Every coset slab exists per part.
    struct Dom {
        uint32_t ek = 0, en = 0;
        int zpow = 1;  // the coset shift is zeta^zpow
        Fe *fixed_cos = nullptr, *sigma_cos = nullptr, *l0 = nullptr, *llast = nullptr, *lactive = nullptr,
           *gate_slabs = nullptr, *t_eval = nullptr, *ext_tw = nullptr;
    };
    Dom dom[3];           // [0]: the single coset; [1], [2]: the two parts of the split domain (when it applies)
    uint32_t nparts = 1;  // 1, or 3 when the split domain is prepared too
    ~PkDev() {
        (void)hipSetDevice(device);
        for (void* q : owned) (void)hipFree(q);
    }
};

// ------------------------------------------------------------------ prover object
// One context (stream + workspace), one proving key (possibly shared), `cap` proof slots: every per-proof buffer is
// [cap] x its single-proof size, proof-major, so that one launch serves every proof of a lock-step batch.
struct zg_prover {
    zg_ctx* ctx = nullptr;
    std::shared_ptr<PkDev> pk;
    zg_bases *g = nullptr, *gl = nullptr;
    // base tables the prover registered itself (zg_prover_create): owned jointly with its forks, freed with the last
    // of them; null when the caller registered the tables (zg_prover_create_shared) and keeps them alive
    struct OwnedBases {
        zg_bases *g = nullptr, *gl = nullptr;
        ~OwnedBases() {
            if (g) zg_bases_free(g);
            if (gl) zg_bases_free(gl);
        }
    };
    std::shared_ptr<OwnedBases> owned_bases;
    bool use_side = true;   // coefficient / coset forms on a side stream (latency) or inline (throughput)
    // what a LONE proof (latency form) borrows from the throughput form once the circuit is large enough for the work
    // to outweigh the launches (from k: K_LAT_SPLIT_K)
    bool lat_split = false;
    uint32_t naf_gl_w = 0;  // digit width of the run-form commitments' free-position form, 0 = windows (naf_gl_default)
    // point-range shard of the commitments (zg_prover_set_shard): this prover's base sets hold points
    // [shard_lo, shard_lo + shard_n) of the SRS; partial commitments of all ranks are exchanged and summed
    uint32_t shard_lo = 0, shard_n = 0, world = 1, rank = 0;
    zg_exchange_fn exchange = nullptr;
    void* exchange_user = nullptr;
    // ... or, with a communicator of the collective library (zg_prover_set_shard_rccl), all-gathered and summed on the
    // device: the phase's partial sums never visit the host before they are whole
    void* rccl_comm = nullptr;
    XYZZ* gathered = nullptr;  // [world][maxv * cap]
    size_t gathered_cap = 0;   // the slot count it is sized for (zg_prover_set_batch regrows it)
    // per-proof buffers, [cap] slots each (alloc_slots)
    uint32_t cap = 0;
    std::vector<void*> slot_owned;
    uint32_t npp = 0;  // per-proof coefficient polynomials: advice, instance, perm z, lookup z, a'/s', random, h pieces, h
    uint32_t ncos = 0; // per-proof coset slabs: advice, instance, perm z, lookup z, a'/s'
    // indices into the coefficient-polynomial space (PolySet: < nsh = F + P shared, the rest per proof)
    uint32_t ix_fixed = 0, ix_sigma = 0, ix_adv = 0, ix_inst = 0, ix_pz = 0, ix_lz = 0, ix_perm = 0, ix_random = 0,
             ix_hpiece = 0, ix_hpoly = 0, nsh = 0;
    Fe* pp = nullptr;  // [cap][npp][n]
    struct DomBuf {
        Fe *cos = nullptr /* [cap][ncos][en] */, *h = nullptr /* [cap][en] */;
    };
    DomBuf dbuf[3];
    Fe* split_tmp = nullptr;  // [cap][3 * dom[2].en]
    Fe *adv_val = nullptr /* [cap][A][n] */, *inst_val = nullptr /* [cap][I][n] */;
    Fe *cin = nullptr, *ctab = nullptr /* [cap * NL][n] each */, *perm = nullptr /* [cap][2NL + 1][n]: a'_l, s'_l, random */,
       *zs = nullptr /* [cap][S + NL + 1][n] */;
    Fe *num = nullptr, *den = nullptr, *tmp = nullptr, *pw = nullptr, *wpoly = nullptr, *raw = nullptr,
       *sraw = nullptr, *sort_fe = nullptr, *ktmp = nullptr;
    uint32_t *sort_u32 = nullptr, *d_err = nullptr;
    XYZZ* xyzz = nullptr;
    uint32_t maxv = 0, max_points = 0, max_evals = 0;
    ProofConst* d_pc = nullptr;
    std::vector<ProofConst> hpc;
    uint32_t* d_idx = nullptr;  // index lists (circuit only: the same for every proof)
    std::map<uint32_t*, std::vector<uint32_t>> uploaded_lists;  // what h2d_list left at each destination
    std::vector<size_t> inst_filled;  // per slot: rows of inst_val that may be non-zero
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_err = nullptr;
    // One event per WAIT of a proof (the five commitment phases and the evaluations): with the gate (below) the next phase's
    // launches -- and its commit's event record -- are queued before the host waits for this one, so they cannot share one.
    static constexpr int N_WAITS = 6;
    hipEvent_t evs[N_WAITS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // The gate of a lone proof (ZG_LAT_GATE, ProveBatch::run): a word of the pinned arena that gate_pull_kernel polls and
    // the host writes once the next challenge is staged -- the next phase is then already in the queue behind that kernel.
    // gate_word()[0] = the last gate opened (a sequence number), [16] = the gate that gave up waiting, if any.
    uint32_t gate_seq = 0;
    uint32_t* gate_word() const { return reinterpret_cast<uint32_t*>((char*)pinned + pinned_cap - 128); }
    // (what the last COMPLETED proof looked like -- ProveBatch::form_sig: a first proof in a form creates twiddle tables
    //  and workspace, with stream synchronisations the gate must not stand in front of; only a repeat is gated)
    uint64_t warm_sig = 0;
    uint64_t gate_stats[4] = {0, 0, 0, 0};  // zg_prover_gate_stats
    void* pinned = nullptr;
    void* pinned_dev = nullptr;  // the same memory as the device addresses it (hipHostGetDevicePointer)
    size_t pinned_cap = 0, pin_results = 0, pin_evals = 0, pin_stage = 0;
    size_t stage_off = 0;
    bool have_last = false;
    bool in_flight = false;   // a batch was started and did not reach its end (an error return): work may still be queued
    bool last_split = false;  // which extended domain the last proof used (zg_prover_fetch)
    uint32_t last_nb = 0;
    double phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace {

// A lone proof of a LARGER circuit is bound by work, not by launches: from this size on it takes the quotient from the
// split domain as the throughput form does (39n instead of 68n butterflies per polynomial, 5n instead of 8n evaluate_h
// rows), its coset transforms still phase by phase on the side stream.  gpurun_out/lone_forms3/4.txt -> profiles/r03/
// lone_forms.txt: k = 14 2.96 -> 3.10 ms (the interpolation's extra launches), k = 15 4.24 -> 4.19 (small) and 4.34 ->
// 4.33 (medium), k = 17 11.74 -> 10.39.  (The bit-position tables lose at every size for a lone proof -- k = 17: 11.57 ->
// 11.74 -- and stay a throughput-form choice.)
constexpr int LAT_SPLIT_K_DEFAULT = 15;

template <class T>
int dalloc_into(std::vector<void*>& owned, T** out, size_t count) {
    void* q = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
        set_error("zg_prover: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return ZG_ERR_OOM;
    }
    owned.push_back(q);
    *out = reinterpret_cast<T*>(q);
    return ZG_OK;
}

inline Fe to_fe(const zg_fr* s) {
    Fe r;
    memcpy(&r, s, 32);
    return r;
}

Fe rotate_omega(const PkDev& k, const Fe& x, int32_t rot) {
    Fe w = rot >= 0 ? Fr::pow_u64(k.omega, (uint64_t)rot) : Fr::pow_u64(k.omega_inv, (uint64_t)(-(int64_t)rot));
    return Fr::mul(x, w);
}

// ZG_LAT_GATE (ProveBatch::run).  OFF unless asked for: a kernel that waits for the host is only safe while every stream
// of the process has a hardware queue of its own (GPU_MAX_HW_QUEUES, default 4): packets of streams that share a queue
// run in order, so this prover's side stream -- which the host waits for before it opens the next gate -- may sit behind
// another stream's wait for work that itself sits behind the gate.  The kernel's time limit turns that cycle into a
// stall and a second, plain run of the proof (prove_batch_impl), never into a hang or a wrong proof.
constexpr int LAT_GATE_DEFAULT = 0;
constexpr int LAT_PULL_DEFAULT = 1;  // (ZG_LAT_PULL: a lone proof's small uploads by a one-wave kernel instead of a copy command)
// Small host->device transfers go through a pinned staging arena: hipMemcpyAsync from pageable memory
// blocks the calling thread until the stream has drained up to the copy, which serialises host and
// GPU inside a proof and throttles concurrent proof streams.  The arena is a bump allocator reset at
// the start of every batch; each region is written once per batch.
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
// (a lone proof's version of a small upload: ONE wave reads the staged bytes through the device's view of the pinned arena
//  and writes them where they go -- a kernel dispatch instead of a copy command behind every transcript step)
__global__ void pull_kernel(const uint4* __restrict__ src_mapped, uint4* __restrict__ dst, uint32_t n16) {
    for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src_mapped[i];
}
// The gate: ONE workgroup that waits until the host has written `seq` into the gate word (mapped, coherent host memory),
// then copies the scalars the host staged BEFORE that store to where the kernels read them -- wait and upload in one
// dispatch.  (hipStreamWaitValue32 + pull_kernel, the first version, were two: on this runtime the stream wait is itself
// a spinning kernel, __amd_rocclr_streamOpsWait.)  Every wave reaches the end: a gate nobody opens within `max_ticks` of
// the 100 MHz clock gives up, says so in *gave_up (the host then fails the proof) and leaves dst alone.
__global__ void gate_pull_kernel(const uint32_t* gate, uint32_t seq, uint32_t* gave_up, uint64_t max_ticks, const uint4* __restrict__ src_mapped,
                                 uint4* __restrict__ dst, uint32_t n16) {
    __shared__ uint32_t open;
    if (threadIdx.x == 0) {
        const uint64_t t0 = wall_clock64();
        uint32_t got = 1;
        while (__hip_atomic_load(gate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (wall_clock64() - t0 > max_ticks) {
                got = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!got) __hip_atomic_store(gave_up, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        open = got;
    }
    __syncthreads();
    if (!open) return;
    for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src_mapped[i];
}
constexpr uint64_t GATE_MAX_TICKS = 400000000ull;  // 4 s: a host thread that is merely descheduled comes back sooner
int h2d(zg_prover* p, void* d_dst, const void* src, size_t bytes) {
    const void* s = stage(p, src, bytes);
    const int pull = knob(K_LAT_PULL);
    if (s && p->use_side && (pull < 0 ? LAT_PULL_DEFAULT : pull) != 0 && bytes % 16 == 0 && bytes <= (1u << 16) && ((uintptr_t)d_dst & 15u) == 0) {
        const uint4* dev_view = reinterpret_cast<const uint4*>((const char*)p->pinned_dev + ((const char*)s - (const char*)p->pinned));
        ZG_LAUNCH(p->ctx, "pull", (double)bytes * 2, pull_kernel, dim3(1), dim3(256), 0, dev_view, (uint4*)d_dst, (uint32_t)(bytes / 16));
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    ZG_HIP(hipMemcpyAsync(d_dst, s ? s : src, bytes, hipMemcpyHostToDevice, p->ctx->stream));
    if (!s) ZG_HIP(hipStreamSynchronize(p->ctx->stream));
    return ZG_OK;
}
// the per-proof scalars of the batch, as the host holds them now, to the device (behind the work already queued)
int upload_consts(zg_prover* p, uint32_t nb) { return h2d(p, p->d_pc, p->hpc.data(), (size_t)nb * sizeof(ProofConst)); }

}  // namespace

// Host: out[i] = normalised sum over ranks r of the extended-Jacobian (X, Y, ZZ, ZZZ; 128 B) partial parts[r * count + i]
// -- the additions that follow the all-gather of a sharded commitment phase.
extern "C" int zg_xyzz_sum_ranks(const void* parts, size_t world, size_t count, zg_g1* out) {
    ZG_REQUIRE(out && (parts || count == 0) && world >= 1, ZG_ERR_INVALID_ARG, "zg_xyzz_sum_ranks: bad argument");
    const XYZZ* all = reinterpret_cast<const XYZZ*>(parts);
    std::vector<XYZZ> sum(count);
    for (size_t i = 0; i < count; i++) {
        XYZZ acc = all[i];
        for (size_t r = 1; r < world; r++) acc = xyzz_add(acc, all[r * count + i]);
        sum[i] = acc;
    }
    xyzz_batch_normalise(sum.data(), count, out);
    return ZG_OK;
}

namespace {

// the bit-position table of a base set for the throughput form's commitments of full-size scalars (the sorted a' / s'
// columns and the products in their run form, the random vectors), else the base set itself
// (a lone proof keeps the window tables: from the 0.27 GB bit-position table, which no cache holds, its few waves wait on
//  the gathers -- 3.27 against 3.01 ms with the same bucket count)
static const zg_bases* naf_of(const zg_prover* p, const zg_bases* b) {
    const zg_bases* d = bases_dense(b);
    return d && d->naf_w && !p->ctx->msm_pair ? d : b;
}

// Digit width of the free-position form for the run-form commitments against g_lagrange (the sorted a' / s' columns and
// the products: full-size coefficients wherever a row changes something, but half-empty vectors -- so the width that
// keeps the bucket count of the window form, c + 1: 254 / (c + 2) digits per coefficient instead of 255 / c windows.
// Same-box A/B at k = 14: 0.7122 -> 0.7060 ms/proof at 13; 12: 0.718, 14: 0.722; 15, the random vectors' width: 0.740.)
// ZG_MSM_NAF_GL = width, 0 = the window form; fixed when the prover is created (zg_prover::naf_gl_w).
static uint32_t naf_gl_default(const zg_bases* gl) {
    const int v = knob(K_MSM_NAF_GL);
    if (v >= 0) return v >= 3 && v <= 16 ? (uint32_t)v : 0u;
    const uint32_t w = gl->c + 1;
    return w >= 3 && w <= 16 ? w : 0u;
}
static uint32_t naf_gl_width(const zg_prover* p) { return p->naf_gl_w; }

static bool lone_split(const zg_prover* p, bool latency_form) {
    const int sk = knob(K_LAT_SPLIT_K);
    return latency_form && p->pk->k >= (uint32_t)(sk >= 0 ? sk : LAT_SPLIT_K_DEFAULT);
}

// the table of g for the all-random commitments (the quotient pieces, the opening quotients): its bit-position table in
// the throughput form, recoded at the width the table was made for
static const zg_bases* dense_g(const zg_prover* p) { return naf_of(p, p->g); }

// Commitments of a phase: one MSM launch sequence over `count` = groups x per scalar vectors (msm_batch4_dev), against
// this prover's point range of the base sets; the XYZZ results go to the host behind it.
// RCCL is bound at run time (dlopen): the library has no link-time dependency on it, and only a prover that was given a
// communicator ever asks for it.
typedef int (*rccl_all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
static rccl_all_gather_fn rccl_all_gather() {
    static rccl_all_gather_fn fn = []() -> rccl_all_gather_fn {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        return h ? (rccl_all_gather_fn)dlsym(h, "ncclAllGather") : nullptr;
    }();
    return fn;
}

// out[i] = sum over ranks r of parts[r * count + i] (extended Jacobian), one lane per commitment: world - 1 additions
__global__ void xyzz_sum_ranks_kernel(const XYZZ* __restrict__ parts, uint32_t world, uint32_t count, XYZZ* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    XYZZ acc = parts[i];
    for (uint32_t r = 1; r < world; r++) acc = xyzz_add(acc, parts[(size_t)r * count + i]);
    out[i] = acc;
}

int commit(zg_prover* p, const zg_bases* a, const zg_bases* b2, size_t split, const Fe* scalars, size_t stride, size_t per,
           size_t outer, size_t count, uint64_t run_mask, uint32_t naf_width = 0, int wait_ix = 0) {
    ZG_REQUIRE(count <= p->maxv * (size_t)p->cap, ZG_ERR_INVALID_ARG, "zg_prover: %zu commitments in one phase", count);
    // The last kernel of the MSM writes its sums straight into the pinned host buffer the transcript reads (the memory is
    // mapped into the device's address space): no copy command between the MSM and the host -- a command costs a lone
    // proof its dispatch latency on top of its run time, ~9 us per commitment phase.  Only the in-library RCCL exchange
    // keeps the sums on the device (all-gather and the additions follow on this stream) and copies the result out.
    XYZZ* h_out = (XYZZ*)((char*)p->pinned_dev + p->pin_results);
    ZG_TRY(msm_batch4_dev(p->ctx, a, b2, split, scalars + p->shard_lo, stride, per, outer, count, p->shard_n,
                          p->rccl_comm ? p->xyzz : h_out, run_mask, naf_width));
    if (p->rccl_comm) {  // ONE all-gather of the phase's partial sums over xGMI, then the additions, all on this stream
        rccl_all_gather_fn gather = rccl_all_gather();
        ZG_REQUIRE(gather != nullptr, ZG_ERR_UNSUPPORTED, "zg_prover: librccl.so could not be loaded");
        const int st = gather(p->xyzz, p->gathered, count * sizeof(XYZZ), /* ncclUint8 */ 1, p->rccl_comm, p->ctx->stream);
        ZG_REQUIRE(st == 0, ZG_ERR_HIP, "zg_prover: ncclAllGather failed with %d", st);
        ZG_LAUNCH(p->ctx, "xyzz_sum_ranks", (double)p->world * count * sizeof(XYZZ), xyzz_sum_ranks_kernel,
                  dim3((uint32_t)((count + 63) / 64)), dim3(64), 0, p->gathered, p->world, (uint32_t)count, h_out);
        ZG_HIP(hipGetLastError());
    }
    ZG_HIP(hipEventRecord(p->evs[wait_ix], p->ctx->stream));
    return ZG_OK;
}
// ... and waits for ONLY that launch sequence; with a sharded SRS the partial commitments of all ranks are exchanged
// (all-gather) and summed here -- EC addition is not a reduction operator of the collective library
int wait_points(zg_prover* p, size_t count, std::vector<Jac>& out, int wait_ix = 0) {
    // (hipEventSynchronize already polls: the events are created without hipEventBlockingSync.  A hand-written
    //  hipEventQuery loop in its place never saw the event complete on this runtime -- round 3, gpurun_out/r3_t3.log --
    //  and bought nothing: the wait was never a sleep.)
    ZG_HIP(hipEventSynchronize(p->evs[wait_ix]));
    const XYZZ* local = (const XYZZ*)((char*)p->pinned + p->pin_results);
    out.resize(count);
    if (p->world <= 1 || p->rccl_comm) {  // (whole sums already: a lone prover, or gathered and added on the device)
        xyzz_batch_normalise(local, count, reinterpret_cast<zg_g1*>(out.data()));
        return ZG_OK;
    }
    ZG_REQUIRE(p->exchange != nullptr, ZG_ERR_INVALID_ARG, "zg_prover: sharded prover without an exchange function");
    std::vector<XYZZ> all((size_t)p->world * count);
    const int st = p->exchange(p->exchange_user, local, count * sizeof(XYZZ), all.data());
    ZG_REQUIRE(st == 0, ZG_ERR_HIP, "zg_prover: the exchange function failed with %d", st);
    return zg_xyzz_sum_ranks(all.data(), p->world, count, reinterpret_cast<zg_g1*>(out.data()));
}

// the scalars evaluate_h reads, in the form it computes in (theta, beta, gamma already in c)
// terms of evaluate_h after the gates: permutation (l0, llast, one l0 per further set, one lactive per set) and five per
// lookup; 0 when they are too many for ProofConst::eh_ypow (or switched off): the kernel then folds in y term by term
uint32_t evalh_terms(const PkDev& pk) {
    const bool grouped = pk.grouped;
    const uint32_t t = (pk.sets ? 2 + (pk.sets - 1) + pk.sets : 0) + 5 * pk.NL;
    // (at most 40: the sums that multiply l0 / lactive then stay below the operand bound of Fr9::mul2)
    return grouped && pk.hat && t >= 1 && t <= 40 && t + 1 <= EH_MAX_YPOW ? t : 0;
}

void evalh_consts(ProofConst& c, const Fe& y, bool hat, uint32_t n_terms) {
    const Fe zeta = fr_zeta(), zeta2 = Fr::sqr(zeta);
    c.eh_y = y; c.eh_beta = c.beta; c.eh_gamma = c.gamma; c.eh_theta = c.theta;
    c.eh_delta_start[0] = Fr::mul(c.beta, zeta);   // beta * coset shift
    c.eh_delta_start[1] = Fr::mul(c.beta, zeta2);
    if (hat) {
        const Fe c261 = Fr9Params::c261_fe();
        for (Fe* cst : {&c.eh_y, &c.eh_beta, &c.eh_gamma, &c.eh_theta, &c.eh_delta_start[0], &c.eh_delta_start[1]})
            *cst = Fr::mul(*cst, c261);
        Fe pw = Fr::one();
        for (uint32_t j = 0; j <= n_terms && j < EH_MAX_YPOW; j++) {
            c.eh_ypow[j] = Fr::mul(pw, c261);
            pw = Fr::mul(pw, y);
        }
    }
}

// evaluate_h's arguments for part `di` of the extended domain: the proving key's slabs, the per-proof slabs of this
// prover's slots (proof b at b * cos_bs), scalars from d_pc
EvalHArgs evalh_args(const zg_prover* p, uint32_t di) {
    const PkDev& pk = *p->pk;
    const uint32_t n = pk.n, k = pk.k, bf = pk.bf, A = pk.A, I = pk.I, NL = pk.NL, S = pk.sets;
    const bool hat = pk.hat;
    const PkDev::Dom& d = pk.dom[di];
    const Fe* cos = p->dbuf[di].cos;
    EvalHArgs a;
    memset(&a, 0, sizeof(a));
    a.c = pk.dc;
    a.cols.fixed = d.fixed_cos; a.cols.advice = cos; a.cols.instance = cos + (size_t)A * d.en;
    a.cols.log_size = d.ek; a.cols.rot_scale = (int32_t)(d.en / n);
    a.cols.adv_bs = a.cols.inst_bs = (size_t)p->ncos * d.en;
    a.sigma_cos = d.sigma_cos;
    a.pz_cos = cos + (size_t)(A + I) * d.en;
    a.lz_cos = a.pz_cos + (size_t)S * d.en;
    // the a'/s' cosets are interleaved, [2l] = a'_l and [2l+1] = s'_l: two views with a stride of two slabs
    a.pin_cos = a.lz_cos + (size_t)NL * d.en; a.ptab_cos = a.pin_cos + d.en; a.perm_stride = (size_t)2 * d.en;
    a.l0 = d.l0; a.llast = d.llast; a.lactive = d.lactive;
    a.ext_tw = hat ? d.ext_tw + d.en : d.ext_tw;  // (the twiddle table's second half is the 2^261 form)
    a.t_eval = d.t_eval; a.t_mask = (1u << (d.ek - k)) - 1;
    a.last_rot = -(int32_t)(bf + 1);
    a.pc = p->d_pc; a.zpow = (uint32_t)d.zpow;
    a.cos_bs = (size_t)p->ncos * d.en; a.h_bs = d.en;
    a.delta = hat ? Fr::mul(fr_delta(), Fr9Params::c261_fe()) : fr_delta();
    a.hat = hat;
    a.monos_hat = pk.monos_hat;
    a.n_terms = evalh_terms(pk);
    a.gates_hat = pk.gates_hat;
    a.gate_common = pk.gate_common;
    a.gate_uni = pk.gate_uni;
    a.uni_coef = pk.uni_coef;
    a.gate_slab = pk.gate_slab;
    a.gate_slabs = d.gate_slabs;
    a.h = p->dbuf[di].h;
    return a;
}

void free_slots(zg_prover* p) {
    for (void* q : p->slot_owned) (void)hipFree(q);
    p->slot_owned.clear();
    if (p->pinned) (void)hipHostFree(p->pinned);
    p->pinned = nullptr;
    p->cap = 0;
    p->uploaded_lists.clear();
    p->warm_sig = 0;  // (new buffers: the next proof is a first proof again -- ProveBatch::gate_wanted)
}

int alloc_slots_impl(zg_prover* p, uint32_t cap);
// (re)allocates every per-proof buffer for `cap` proofs in flight; on failure the prover is left without slots
// (zg_prover_prove* then refuse with "0 slots") rather than with half of them
int alloc_slots(zg_prover* p, uint32_t cap) {
    const int st = alloc_slots_impl(p, cap);
    if (st != ZG_OK) free_slots(p);
    return st;
}
int alloc_slots_impl(zg_prover* p, uint32_t cap) {
    const PkDev& k = *p->pk;
    ZG_REQUIRE(cap >= 1 && cap <= 1024, ZG_ERR_INVALID_ARG, "zg_prover: batch of %u proofs", cap);
    free_slots(p);
    auto& own = p->slot_owned;
    const uint32_t n = k.n, F = k.F, A = k.A, I = k.I, P = k.P, NL = k.NL, S = k.sets, Q = k.qpd;
    (void)F;
    p->cap = cap;
    p->ncos = A + I + S + NL + 2 * NL;
    p->npp = p->ncos + 1 + Q + 1;
    p->nsh = k.F + P;
    p->ix_fixed = 0; p->ix_sigma = k.F; p->ix_adv = p->nsh; p->ix_inst = p->ix_adv + A; p->ix_pz = p->ix_inst + I;
    p->ix_lz = p->ix_pz + S; p->ix_perm = p->ix_lz + NL; p->ix_random = p->ix_perm + 2 * NL;
    p->ix_hpiece = p->ix_random + 1; p->ix_hpoly = p->ix_hpiece + Q;
    const size_t c = cap;
    ZG_TRY(dalloc_into(own, &p->pp, c * p->npp * n));
    for (uint32_t di = 0; di < k.nparts; di++) {
        // (one block per proof, in the order of the coefficient slab: advice, instance, permutation z, lookup z, a'/s'
        //  -- the split form transforms all of them in one batch)
        ZG_TRY(dalloc_into(own, &p->dbuf[di].cos, c * p->ncos * k.dom[di].en));
        ZG_TRY(dalloc_into(own, &p->dbuf[di].h, c * k.dom[di].en));
    }
    if (k.nparts == 3) ZG_TRY(dalloc_into(own, &p->split_tmp, c * 3 * k.dom[2].en));
    ZG_TRY(dalloc_into(own, &p->adv_val, c * A * n));
    ZG_TRY(dalloc_into(own, &p->inst_val, c * I * n));
    if (I) ZG_HIP(hipMemset(p->inst_val, 0, c * I * n * 32));  // rows past the instance stay zero (prove refills only what it must)
    p->inst_filled.assign(cap, 0);
    ZG_TRY(dalloc_into(own, &p->cin, c * 2 * NL * n));  // compressed inputs, then compressed tables
    p->ctab = p->cin + c * NL * n;
    ZG_TRY(dalloc_into(own, &p->perm, c * (2 * NL + 1) * n));  // + the vanishing argument's random polynomial
    ZG_TRY(dalloc_into(own, &p->zs, c * (S + NL + 1) * n));
    const uint32_t mb = S + NL;
    ZG_TRY(dalloc_into(own, &p->num, c * mb * n));
    ZG_TRY(dalloc_into(own, &p->den, c * mb * n));
    ZG_TRY(dalloc_into(own, &p->tmp, poly_grand_product_tmp_elems(n, cap * mb)));
    ZG_TRY(dalloc_into(own, &p->raw, c * 2 * NL * n));
    ZG_TRY(dalloc_into(own, &p->sraw, c * NL * n));
    ZG_TRY(dalloc_into(own, &p->sort_fe, c * NL * n));
    ZG_TRY(dalloc_into(own, &p->sort_u32, c * 2 * NL * n + c * 3 * NL + 2));
    p->max_points = 4 + (uint32_t)(k.advice_queries.size() + k.fixed_queries.size());
    if (p->max_points > PC_MAX_POINTS) p->max_points = PC_MAX_POINTS;
    ZG_TRY(dalloc_into(own, &p->pw, c * p->max_points * n));
    p->max_evals = (uint32_t)(k.advice_queries.size() + k.fixed_queries.size()) + P + 3 * S + 5 * NL + 4;
    ZG_TRY(dalloc_into(own, &p->wpoly, c * 2 * p->max_points * n));
    p->maxv = std::max<uint32_t>(std::max<uint32_t>(A, 2 * NL + 1), std::max<uint32_t>(S + NL + 1, std::max<uint32_t>(Q, p->max_points)));
    ZG_TRY(dalloc_into(own, &p->xyzz, c * p->maxv));
    ZG_TRY(dalloc_into(own, &p->d_idx, (size_t)4 * p->max_evals + 64 + (size_t)p->max_points * 512));
    ZG_TRY(dalloc_into(own, &p->ktmp, poly_kate_tmp_elems(n, cap * p->max_points)));
    ZG_TRY(dalloc_into(own, &p->d_pc, c));
    p->hpc.assign(cap, ProofConst{});
    // pinned: commitments D2H (128 B each), evaluations D2H, error flags, then the H2D staging arena
    p->pin_results = 0;
    p->pin_evals = (c * p->maxv * sizeof(XYZZ) + 4095) & ~size_t(4095);
    p->pin_stage = p->pin_evals + ((c * p->max_evals * sizeof(Fe) + c * NL * 4 + 4095) & ~size_t(4095));
    p->pinned_cap = p->pin_stage + (1u << 20) + c * 16 * sizeof(ProofConst) + c * (size_t)k.I * 4096;
    // Kernels store results here that the host reads right after an event.  COHERENT (fine-grained), said explicitly:
    // with any flag but the default the runtime takes the coherence of host memory from HIP_HOST_COHERENT, and an event's
    // default release is device scope, which promises nothing about non-coherent host memory (the non-coherent form with a
    // hipEventReleaseToSystem event measured the same: 2.17 against 2.18 ms for a lone k = 14 proof).
    ZG_HIP(hipHostMalloc(&p->pinned, p->pinned_cap, hipHostMallocMapped | hipHostMallocCoherent));
    ZG_HIP(hipHostGetDevicePointer(&p->pinned_dev, p->pinned, 0));
    memset((char*)p->pinned + p->pinned_cap - 4096, 0, 4096);  // (the margin behind the staging arena: zg_prover::gate_word)
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
            // (up to sign: a coefficient -1 costs the kernel a subtraction, like +1 an addition)
            size_t best = 0, best_n = 0;
            for (size_t i = 0; i < base.size(); i++) {
                size_t cnt = 0;
                const Fe neg_i = Fr::neg(base[i].coeff);
                for (const DMono& o : base) cnt += fe_eq(o.coeff, base[i].coeff) || fe_eq(o.coeff, neg_i);
                if (cnt > best_n) best = i, best_n = cnt;
            }
            const Fe d = base[best].coeff, d_inv = Fr::inv(d);
            const Fe one_hat = Fr::mul(Fr::one(), c261);
            for (DMono& o : base) {
                o.coeff = Fr::mul(Fr::mul(o.coeff, d_inv), c261);  // (o / d) back in the 2^261 form
                o.coeff_is_one = fe_eq(o.coeff, one_hat) ? 1 : fe_eq(o.coeff, Fr::neg(one_hat)) ? 2 : 0;
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

namespace zg {
ProverShape prover_shape(const zg_prover* p) {
    const PkDev& k = *p->pk;
    return ProverShape{p->ctx, k.device, k.k, k.A, k.I, k.usable, p->in_flight};
}
int prover_drain(zg_prover* p) {
    ZG_ENTER(p->ctx);
    ZG_HIP(hipStreamSynchronize(p->ctx->stream));
    if (p->ctx->side) ZG_HIP(hipStreamSynchronize(p->ctx->side->stream));
    return ZG_OK;
}
}  // namespace zg

extern "C" {

void zg_keccak256(const uint8_t* data, size_t len, uint8_t out[32]) { keccak256(data, len, out); }

size_t zg_prover_proof_size(const zg_prover* p) {
    if (!p) return 0;
    const PkDev& k = *p->pk;
    size_t points = k.A + 2 * k.NL + k.sets + k.NL + 1 + k.qpd;
    size_t scalars = k.advice_queries.size() + k.fixed_queries.size() + 1 + k.P + (k.sets ? 3 * k.sets - 1 : 0) + 5 * k.NL;
    size_t max_open = 2 + k.advice_queries.size() + k.fixed_queries.size();
    return 64 * (points + max_open) + 32 * scalars;
}

void zg_prover_destroy(zg_prover* p) {
    if (!p) return;
    {
        std::lock_guard<std::recursive_mutex> lock(p->ctx->mu);
        (void)hipSetDevice(p->ctx->device);
        (void)hipStreamSynchronize(p->ctx->stream);
        if (p->ctx->side) (void)hipStreamSynchronize(p->ctx->side->stream);
        free_slots(p);
        if (p->gathered) (void)hipFree(p->gathered);
        p->owned_bases.reset();  // (the tables go with the last prover that uses them)
        if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
        if (p->ev_join) (void)hipEventDestroy(p->ev_join);
        if (p->ev_err) (void)hipEventDestroy(p->ev_err);
        for (hipEvent_t e : p->evs)
            if (e) (void)hipEventDestroy(e);
        p->pk.reset();  // (the key's HBM goes with its last prover)
    }
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

static int prover_events(zg_prover* p) {
    ZG_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    ZG_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    ZG_HIP(hipEventCreateWithFlags(&p->ev_err, hipEventDisableTiming));
    for (hipEvent_t& e : p->evs) ZG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return ZG_OK;
}

static int prover_create_impl(zg_ctx* ctx, const zg_circuit* cs, const zg_fr* fixed_values, const zg_fr* sigma_values,
                              const zg_g1_affine* g, const zg_g1_affine* g_lagrange, const zg_bases* shared_g,
                              const zg_bases* shared_gl, const zg_fr* vk_repr, zg_prover** out) {
    ZG_REQUIRE(ctx && cs && vk_repr && out, ZG_ERR_INVALID_ARG, "zg_prover_create: null argument");
    ZG_REQUIRE(cs->n_fixed == 0 || fixed_values, ZG_ERR_INVALID_ARG, "zg_prover_create: fixed_values is null");
    ZG_REQUIRE(cs->n_perm_columns == 0 || sigma_values, ZG_ERR_INVALID_ARG, "zg_prover_create: sigma_values is null");
    ZG_REQUIRE(cs->cs_degree >= 3 && cs->cs_degree <= 9, ZG_ERR_UNSUPPORTED, "zg_prover_create: cs_degree %u", cs->cs_degree);
    ZG_REQUIRE(cs->k >= 4, ZG_ERR_UNSUPPORTED, "zg_prover_create: k=%u < 4", cs->k);
    ZG_ENTER(ctx);
    std::unique_ptr<zg_prover, void (*)(zg_prover*)> guard(new zg_prover(), zg_prover_destroy);
    zg_prover* p = guard.get();
    p->ctx = ctx;
    p->pk = std::make_shared<PkDev>();
    PkDev* pk = p->pk.get();
    pk->device = ctx->device;
    pk->k = cs->k;
    pk->n = 1u << cs->k;
    pk->cs_degree = cs->cs_degree;
    pk->bf = cs->blinding_factors;
    pk->qpd = cs->cs_degree - 1;
    pk->ext_k = cs->k;
    while ((1ull << pk->ext_k) < (uint64_t)pk->n * pk->qpd) pk->ext_k++;
    ZG_REQUIRE(pk->ext_k <= 22, ZG_ERR_UNSUPPORTED, "zg_prover_create: extended domain 2^%u not built", pk->ext_k);
    pk->en = 1u << pk->ext_k;
    ZG_REQUIRE(pk->n > pk->bf + 2, ZG_ERR_INVALID_ARG, "zg_prover_create: too few rows");
    pk->usable = pk->n - (pk->bf + 1);
    pk->F = cs->n_fixed; pk->A = cs->n_advice; pk->I = cs->n_instance; pk->P = cs->n_perm_columns; pk->NL = cs->n_lookups;
    pk->chunk = cs->cs_degree - 2;
    pk->sets = pk->P ? (pk->P + pk->chunk - 1) / pk->chunk : 0;
    pk->advice_queries.assign(cs->advice_queries, cs->advice_queries + cs->n_advice_queries);
    pk->fixed_queries.assign(cs->fixed_queries, cs->fixed_queries + cs->n_fixed_queries);
    pk->vk_repr = to_fe(vk_repr);
    pk->omega = host_domain_omega(pk->k);
    pk->omega_inv = Fr::inv(pk->omega);
    pk->ifft_div = Fr::inv(Fr::from_u64(pk->n));
    const uint32_t n = pk->n, en = pk->en;
    hipStream_t st = ctx->stream;
    ZG_TRY(prover_events(p));
    pk->hat = knob(K_EVALH9) != 0;
    pk->grouped = knob(K_EVALH_GROUPED) != 0;
    if (p->use_side && !ctx->side) ZG_TRY(zg_ctx_create(ctx->device, &ctx->side));
    auto dalloc = [&](auto** o, size_t count) { return dalloc_into(pk->owned, o, count); };

    // ---- validate and upload the circuit tables
    for (uint32_t q = 0; q < cs->n_queries; q++) {
        const zg_query& qq = cs->queries[q];
        uint32_t lim = qq.kind == ZG_FIXED ? pk->F : qq.kind == ZG_ADVICE ? pk->A : qq.kind == ZG_INSTANCE ? pk->I : 0;
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
        d.coeff_is_one = fe_eq(d.coeff, one) ? 1 : fe_eq(d.coeff, Fr::neg(one)) ? 2 : 0;
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
        uint32_t lim = qq.kind == ZG_FIXED ? pk->F : qq.kind == ZG_ADVICE ? pk->A : qq.kind == ZG_INSTANCE ? pk->I : 0;
        ZG_REQUIRE(qq.column < lim, ZG_ERR_INVALID_ARG, "zg_prover_create: permutation column %u out of range", c);
    }
    ZG_REQUIRE(cs->n_lookups <= 60, ZG_ERR_UNSUPPORTED, "zg_prover_create: %u lookups", cs->n_lookups);
    zg_query* d_q; DMono* d_m; zg_poly* d_g; DLookup* d_l; zg_query* d_pc;
    ZG_TRY(dalloc(&d_q, cs->n_queries));
    ZG_TRY(dalloc(&d_m, cs->n_monomials));
    ZG_TRY(dalloc(&d_g, cs->n_gates));
    ZG_TRY(dalloc(&d_l, cs->n_lookups));
    ZG_TRY(dalloc(&d_pc, cs->n_perm_columns));
    if (cs->n_queries) ZG_HIP(hipMemcpyAsync(d_q, cs->queries, cs->n_queries * sizeof(zg_query), hipMemcpyHostToDevice, st));
    if (cs->n_monomials) ZG_HIP(hipMemcpyAsync(d_m, monos.data(), monos.size() * sizeof(DMono), hipMemcpyHostToDevice, st));
    if (cs->n_gates) ZG_HIP(hipMemcpyAsync(d_g, cs->gates, cs->n_gates * sizeof(zg_poly), hipMemcpyHostToDevice, st));
    if (cs->n_lookups) ZG_HIP(hipMemcpyAsync(d_l, lks.data(), lks.size() * sizeof(DLookup), hipMemcpyHostToDevice, st));
    if (cs->n_perm_columns) ZG_HIP(hipMemcpyAsync(d_pc, cs->perm_columns, cs->n_perm_columns * sizeof(zg_query), hipMemcpyHostToDevice, st));
    ZG_HIP(hipStreamSynchronize(st));  // the host vectors above go out of scope
    pk->dc.queries = d_q; pk->dc.monos = d_m; pk->dc.gates = d_g; pk->dc.lookups = d_l; pk->dc.perm_cols = d_pc;
    pk->dc.n_gates = cs->n_gates; pk->dc.n_lookups = cs->n_lookups; pk->dc.n_perm = pk->P; pk->dc.chunk = pk->chunk;
    pk->dc.n_sets = pk->sets;
    if (pk->hat) {  // evaluate_h's view: coefficients in the 2^261 form, gates factored by their common cell
        // Invariant (a fault in round 1, gdb: evaluate_h9_kernel reading monos_hat[m].n_factors through a null table):
        // hat implies that monos_hat, gates_hat, gate_common, gate_uni, uni_coef and gate_slab are ALL allocated here,
        // whatever the circuit holds (no gates, no lookups); poly_evaluate_h refuses a launch without them.
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
                if (cs->queries[f].kind == ZG_FIXED) pk->slab_jobs.push_back({gi, f, gate_uni[gi].first, gate_uni[gi].count});
                uni_coef.insert(uni_coef.end(), best.uc.begin(), best.uc.end());
            }
            common[gi] = f;
        }
        ZG_TRY(dalloc(&pk->gate_uni, cs->n_gates ? cs->n_gates : 1));
        ZG_TRY(dalloc(&pk->uni_coef, uni_coef.size() ? uni_coef.size() : 1));
        if (cs->n_gates) ZG_HIP(hipMemcpy(pk->gate_uni, gate_uni.data(), cs->n_gates * sizeof(zg_poly), hipMemcpyHostToDevice));
        if (!uni_coef.empty()) ZG_HIP(hipMemcpy(pk->uni_coef, uni_coef.data(), uni_coef.size() * sizeof(Fe), hipMemcpyHostToDevice));
        ZG_TRY(dalloc(&pk->monos_hat, monos.size() ? monos.size() : 1));
        ZG_TRY(dalloc(&pk->gates_hat, cs->n_gates ? cs->n_gates : 1));
        ZG_TRY(dalloc(&pk->gate_common, cs->n_gates ? cs->n_gates : 1));
        if (!monos.empty()) ZG_HIP(hipMemcpy(pk->monos_hat, monos.data(), monos.size() * sizeof(DMono), hipMemcpyHostToDevice));
        if (cs->n_gates) {
            ZG_HIP(hipMemcpy(pk->gates_hat, gates_hat.data(), cs->n_gates * sizeof(zg_poly), hipMemcpyHostToDevice));
            ZG_HIP(hipMemcpy(pk->gate_common, common.data(), cs->n_gates * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }

    // ---- SRS: upload + window tables, or tables shared with other provers on this device (read-only)
    if (shared_g) {
        ZG_REQUIRE(shared_g->device == ctx->device && shared_gl->device == ctx->device, ZG_ERR_INVALID_ARG,
                   "zg_prover_create_shared: bases live on another device");
        ZG_REQUIRE(shared_g->n == shared_gl->n && shared_g->n <= n && shared_g->n >= 1 && shared_g->c == shared_gl->c, ZG_ERR_INVALID_ARG,
                   "zg_prover_create_shared: bases do not match 2^k = %u points", n);
        p->g = const_cast<zg_bases*>(shared_g);
        p->gl = const_cast<zg_bases*>(shared_gl);
    } else {
        p->owned_bases = std::make_shared<zg_prover::OwnedBases>();
        WsScope ws(ctx);
        Affine* d = ws.get<Affine>(n);
        if (!d) return ZG_ERR_OOM;
        ZG_HIP(hipMemcpyAsync(d, g, (size_t)n * sizeof(Affine), hipMemcpyHostToDevice, st));
        ZG_TRY(bases_register_dev(ctx, d, n, 0, &p->owned_bases->g));
        ZG_HIP(hipMemcpyAsync(d, g_lagrange, (size_t)n * sizeof(Affine), hipMemcpyHostToDevice, st));
        ZG_TRY(bases_register_dev(ctx, d, n, 0, &p->owned_bases->gl));
        p->g = p->owned_bases->g;
        p->gl = p->owned_bases->gl;
    }
    // (fewer points than 2^k: a point-range shard of the SRS; zg_prover_set_shard names the range before the first proof)
    p->shard_lo = 0;
    p->shard_n = (uint32_t)p->g->n;
    const bool run_form = knob(K_MSM_RUNS) != 0;
    if (run_form && pk->sets + pk->NL > 0) ZG_TRY(bases_enable_runs(ctx, p->gl));
    p->naf_gl_w = naf_gl_default(p->gl);
    {   // The quotient pieces and the opening quotients are vectors of random scalars, which fill every window.  With
        // 288 GB of HBM the base set gets a table with one row per BIT position (bases_enable_naf; 255 rows: 0.27 GB at
        // k = 14, 2.1 GB at k = 17) and a scalar is recoded into odd signed digits at FREE positions: 254 / (w + 1)
        // additions per scalar and only odd buckets -- 15.9 at w = 15 where 12-bit windows spend 22.  Same-box A/B at
        // k = 14: 0.7447 -> 0.7319 ms/proof against a larger-window table (w = 15; 14: 0.7355, 13: 0.744, 16: 0.753);
        // k = 17, w = 16: 5.97 -> 5.895.  ZG_MSM_NAF = digit width, 0 = the window table only.
        const int naf_env = knob(K_MSM_NAF);
        uint32_t lg = 0;
        while ((2u << lg) <= p->g->n) lg++;
        uint32_t nw = lg >= 16 ? 16u : lg >= 14 ? 15u : lg + 1 < 3 ? 3u : lg + 1;
        if (naf_env >= 0) nw = (uint32_t)naf_env;
        if (nw >= 3 && nw <= 16) {
            ZG_TRY(bases_enable_naf(ctx, p->g, nw));
            const zg_bases* gd = bases_dense(p->g);
            if (p->naf_gl_w && p->gl->run_table && gd && gd->naf_w) {
                // the same for g_lagrange and its running sums: the sorted columns and the products are committed in the
                // run form, whose coefficients s_i - s_{i+1} are full-size scalars wherever a row changes something
                ZG_TRY(bases_enable_naf(ctx, p->gl, gd->naf_w));
                if (zg_bases* gld = bases_dense(p->gl)) ZG_TRY(bases_enable_runs(ctx, gld));
            }
        }
    }

    // ---- proving-key slabs
    const uint32_t F = pk->F, P = pk->P, Q = pk->qpd;
    ZG_TRY(dalloc(&pk->sh_polys, (size_t)(F + P) * n));
    ZG_TRY(dalloc(&pk->fixed_val, (size_t)F * n));
    ZG_TRY(dalloc(&pk->sigma_val, (size_t)P * n));
    // parts of the extended domain
    {
        const bool split_env = knob(K_SPLIT_DOMAIN) != 0;
        uint32_t m1 = 1;
        while (m1 * 2 <= Q) m1 *= 2;
        const uint32_t m2 = Q - m1;
        const bool split = split_env && pk->hat && m2 != 0 && (m2 & (m2 - 1)) == 0 && (m1 + m2) * n < en;
        auto log2u = [](uint32_t v) { uint32_t l = 0; while ((1u << l) < v) l++; return l; };
        // The single coset serves the latency configuration (a lone proof pays for the extra launches of the split
        // form in its h phase: 0.72 -> 0.93 ms), the split one the throughput configuration (-5 % ms/proof); both sets
        // of proving-key cosets are kept (+60 % of 0.2 GB per key) and zg_prover_set_overlap picks.
        pk->nparts = 1;
        pk->dom[0].ek = pk->ext_k; pk->dom[0].en = en; pk->dom[0].zpow = 1;
        if (split) {
            pk->nparts = 3;
            pk->dom[1].ek = pk->k + log2u(m1); pk->dom[1].en = n * m1; pk->dom[1].zpow = 1;
            pk->dom[2].ek = pk->k + log2u(m2); pk->dom[2].en = n * m2; pk->dom[2].zpow = 2;
        }
    }
    for (uint32_t di = 0; di < pk->nparts; di++) {
        PkDev::Dom& d = pk->dom[di];
        ZG_TRY(dalloc(&d.fixed_cos, (size_t)F * d.en));
        ZG_TRY(dalloc(&d.sigma_cos, (size_t)P * d.en));
        ZG_TRY(dalloc(&d.l0, (size_t)d.en));
        ZG_TRY(dalloc(&d.llast, (size_t)d.en));
        ZG_TRY(dalloc(&d.lactive, (size_t)d.en));
    }

    // ---- keygen_pk's derived data: fixed / sigma polys + cosets, l_0 / l_last / l_active_row
    if (F) {
        ZG_HIP(hipMemcpyAsync(pk->fixed_val, fixed_values, (size_t)F * n * 32, hipMemcpyHostToDevice, st));
        Fe* fp = pk->sh_polys;
        ZG_TRY(ntt_batch_to_dev(ctx, pk->fixed_val, fp, n, F, pk->k, pk->omega_inv, &pk->ifft_div));
        for (uint32_t di = 0; di < pk->nparts; di++) {
            PkDev::Dom& d = pk->dom[di];
            ZG_TRY(coeff_to_coset_dev(ctx, fp, n, n, d.fixed_cos, d.en, F, d.ek, pk->hat, d.zpow));
        }
    }
    if (pk->hat) {
        // a gate factor that is a polynomial in a FIXED cell (a merged selector) does not depend on the
        // witness: its coset is part of the proving key here, as the unmerged selector's would have been
        std::vector<uint32_t> slab_of(cs->n_gates ? cs->n_gates : 1, 0xffffffffu);
        ZG_TRY(dalloc(&pk->gate_slab, slab_of.size()));
        for (uint32_t di = 0; di < pk->nparts; di++) {
            PkDev::Dom& d = pk->dom[di];
            ZG_TRY(dalloc(&d.gate_slabs, std::max<size_t>(1, pk->slab_jobs.size() * (size_t)d.en)));
            for (size_t j = 0; j < pk->slab_jobs.size(); j++) {
                const auto& job = pk->slab_jobs[j];
                const zg_query q = cs->queries[job.query];
                ZG_TRY(poly_gate_factor(ctx, d.fixed_cos + (size_t)q.column * d.en, (uint32_t)(q.rotation * (int32_t)(d.en / n)),
                                        d.en, pk->uni_coef + job.first, job.count, d.gate_slabs + j * (size_t)d.en));
                slab_of[job.gate] = (uint32_t)j;
            }
        }
        ZG_HIP(hipMemcpyAsync(pk->gate_slab, slab_of.data(), slab_of.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        ZG_HIP(hipStreamSynchronize(st));
    }
    if (P) {
        ZG_HIP(hipMemcpyAsync(pk->sigma_val, sigma_values, (size_t)P * n * 32, hipMemcpyHostToDevice, st));
        Fe* sp = pk->sh_polys + (size_t)F * n;
        ZG_TRY(ntt_batch_to_dev(ctx, pk->sigma_val, sp, n, P, pk->k, pk->omega_inv, &pk->ifft_div));
        for (uint32_t di = 0; di < pk->nparts; di++) {
            PkDev::Dom& d = pk->dom[di];
            ZG_TRY(coeff_to_coset_dev(ctx, sp, n, n, d.sigma_cos, d.en, P, d.ek, pk->hat, d.zpow));
        }
    }
    {
        WsScope ws(ctx);
        Fe* t3 = ws.get<Fe>((size_t)3 * n);
        Fe* lblind = ws.get<Fe>(en);
        if (ws.failed) return ZG_ERR_OOM;
        ZG_TRY(poly_l_cosets_init(ctx, t3, t3 + n, t3 + 2 * n, n, pk->bf));
        ZG_TRY(ntt_batch_dev(ctx, t3, n, 3, pk->k, pk->omega_inv, &pk->ifft_div));
        for (uint32_t di = 0; di < pk->nparts; di++) {
            PkDev::Dom& d = pk->dom[di];
            ZG_TRY(coeff_to_coset_dev(ctx, t3, n, n, d.l0, d.en, 1, d.ek, pk->hat, d.zpow));
            ZG_TRY(coeff_to_coset_dev(ctx, t3 + n, n, n, d.llast, d.en, 1, d.ek, pk->hat, d.zpow));
            ZG_TRY(coeff_to_coset_dev(ctx, t3 + 2 * n, n, n, lblind, d.en, 1, d.ek, pk->hat, d.zpow));
            ZG_TRY(poly_lactive(ctx, d.lactive, d.llast, lblind, d.en, pk->hat));
        }
        ZG_HIP(hipStreamSynchronize(st));
    }
    // t_evaluations of EvaluationDomain: ((shift * ext_omega^i)^n - 1)^-1, one period, per part of the domain
    ZG_TRY(get_twiddles(ctx, pk->k, pk->omega, &pk->omega_tw));
    for (uint32_t di = 0; di < pk->nparts; di++) {
        PkDev::Dom& d = pk->dom[di];
        uint32_t t_len = 1u << (d.ek - pk->k);
        std::vector<Fe> te(t_len);
        Fe ext_omega = host_domain_omega(d.ek);
        const Fe shift = d.zpow == 1 ? fr_zeta() : Fr::sqr(fr_zeta());
        Fe cur = Fr::pow_u64(shift, n), step = Fr::pow_u64(ext_omega, n);
        for (uint32_t i = 0; i < t_len; i++) {
            te[i] = Fr::inv(Fr::sub(cur, Fr::one()));
            if (pk->hat) te[i] = Fr::mul(te[i], Fr9Params::c261_fe());
            cur = Fr::mul(cur, step);
        }
        ZG_TRY(dalloc(&d.t_eval, t_len));
        ZG_HIP(hipMemcpy(d.t_eval, te.data(), t_len * sizeof(Fe), hipMemcpyHostToDevice));
        ZG_TRY(get_twiddles(ctx, d.ek, ext_omega, &d.ext_tw));
    }
    ZG_HIP(hipStreamSynchronize(st));
    p->lat_split = lone_split(p, p->use_side);  // (a prover starts in the latency form)
    ZG_TRY(alloc_slots(p, 1));
    *out = guard.release();
    return ZG_OK;
}

int zg_prover_set_batch(zg_prover* p, size_t max_batch) {
    ZG_REQUIRE(p, ZG_ERR_INVALID_ARG, "zg_prover_set_batch: null prover");
    ZG_ENTER(p->ctx);
    ZG_HIP(hipStreamSynchronize(p->ctx->stream));
    if (p->ctx->side) ZG_HIP(hipStreamSynchronize(p->ctx->side->stream));
    if (max_batch == p->cap) return ZG_OK;
    ZG_REQUIRE(max_batch >= 1 && max_batch <= 1024, ZG_ERR_INVALID_ARG, "zg_prover_set_batch: %zu proofs", max_batch);
    p->have_last = false;
    ZG_TRY(alloc_slots(p, (uint32_t)max_batch));
    if (p->rccl_comm && p->gathered_cap < p->cap) {  // the gather buffer follows the slot count
        if (p->gathered) (void)hipFree(p->gathered);
        p->gathered = nullptr;
        ZG_HIP(hipMalloc((void**)&p->gathered, (size_t)p->world * p->maxv * p->cap * sizeof(XYZZ)));
        p->gathered_cap = p->cap;
    }
    return ZG_OK;
}

size_t zg_prover_batch(const zg_prover* p) { return p ? p->cap : 0; }

void* zg_prover_advice_slot(zg_prover* p, size_t slot) {
    if (!p || slot >= p->cap) return nullptr;
    return p->adv_val + slot * (size_t)p->pk->A * p->pk->n;
}

int zg_prover_fork(const zg_prover* parent, zg_ctx* ctx, zg_prover** out) {
    ZG_REQUIRE(parent && ctx && out, ZG_ERR_INVALID_ARG, "zg_prover_fork: null argument");
    ZG_REQUIRE(ctx->device == parent->pk->device, ZG_ERR_INVALID_ARG, "zg_prover_fork: the context is on another device");
    ZG_ENTER(ctx);
    std::unique_ptr<zg_prover, void (*)(zg_prover*)> guard(new zg_prover(), zg_prover_destroy);
    zg_prover* p = guard.get();
    p->ctx = ctx;
    p->pk = parent->pk;
    p->g = parent->g;
    p->gl = parent->gl;
    p->owned_bases = parent->owned_bases;  // (joint ownership; null when the caller registered the tables)
    p->use_side = parent->use_side;
    p->lat_split = parent->lat_split;
    p->naf_gl_w = parent->naf_gl_w;
    p->shard_lo = parent->shard_lo; p->shard_n = parent->shard_n; p->world = parent->world; p->rank = parent->rank;
    p->exchange = parent->exchange; p->exchange_user = parent->exchange_user;
    // (a communicator serialises its collectives on ONE stream: a fork does not inherit it)
    ZG_TRY(prover_events(p));
    if (p->use_side && !ctx->side) ZG_TRY(zg_ctx_create(ctx->device, &ctx->side));
    ZG_TRY(alloc_slots(p, parent->cap ? parent->cap : 1));
    *out = guard.release();
    return ZG_OK;
}

int zg_prover_set_shard(zg_prover* p, uint32_t rank, uint32_t world, size_t first_point, zg_exchange_fn fn, void* user) {
    ZG_REQUIRE(p, ZG_ERR_INVALID_ARG, "zg_prover_set_shard: null prover");
    ZG_ENTER(p->ctx);
    ZG_REQUIRE(world >= 1 && rank < world, ZG_ERR_INVALID_ARG, "zg_prover_set_shard: rank %u of %u", rank, world);
    ZG_REQUIRE(world == 1 || fn != nullptr, ZG_ERR_INVALID_ARG, "zg_prover_set_shard: no exchange function");
    ZG_REQUIRE(first_point + p->g->n <= p->pk->n, ZG_ERR_INVALID_ARG, "zg_prover_set_shard: points [%zu, %zu) of %u", first_point,
               first_point + p->g->n, p->pk->n);
    ZG_REQUIRE(world > 1 || p->g->n == p->pk->n, ZG_ERR_INVALID_ARG, "zg_prover_set_shard: a lone prover needs all 2^k points");
    p->rank = rank;
    p->world = world;
    p->shard_lo = (uint32_t)first_point;
    p->shard_n = (uint32_t)p->g->n;
    p->exchange = fn;
    p->exchange_user = user;
    p->rccl_comm = nullptr;
    return ZG_OK;
}

int zg_xyzz_sum_ranks_dev(zg_ctx* ctx, const void* d_parts, size_t world, size_t count, void* d_out) {
    ZG_REQUIRE(ctx && d_out && (d_parts || count == 0) && world >= 1 && count < (1ull << 31), ZG_ERR_INVALID_ARG,
               "zg_xyzz_sum_ranks_dev: bad argument");
    if (count == 0) return ZG_OK;
    ZG_ENTER(ctx);
    ZG_LAUNCH(ctx, "xyzz_sum_ranks", (double)world * count * sizeof(XYZZ), xyzz_sum_ranks_kernel, dim3((uint32_t)((count + 63) / 64)),
              dim3(64), 0, (const XYZZ*)d_parts, (uint32_t)world, (uint32_t)count, (XYZZ*)d_out);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int zg_prover_set_shard_rccl(zg_prover* p, uint32_t rank, uint32_t world, size_t first_point, void* nccl_comm) {
    ZG_REQUIRE(p && nccl_comm, ZG_ERR_INVALID_ARG, "zg_prover_set_shard_rccl: null argument");
    ZG_ENTER(p->ctx);
    ZG_REQUIRE(world >= 1 && rank < world, ZG_ERR_INVALID_ARG, "zg_prover_set_shard_rccl: rank %u of %u", rank, world);
    ZG_REQUIRE(first_point + p->g->n <= p->pk->n, ZG_ERR_INVALID_ARG, "zg_prover_set_shard_rccl: points [%zu, %zu) of %u", first_point,
               first_point + p->g->n, p->pk->n);
    ZG_REQUIRE(world > 1 || p->g->n == p->pk->n, ZG_ERR_INVALID_ARG, "zg_prover_set_shard_rccl: a lone prover needs all 2^k points");
    ZG_REQUIRE(rccl_all_gather() != nullptr, ZG_ERR_UNSUPPORTED, "zg_prover_set_shard_rccl: librccl.so could not be loaded");
    if (p->gathered) (void)hipFree(p->gathered);
    p->gathered = nullptr;
    ZG_HIP(hipMalloc((void**)&p->gathered, (size_t)world * p->maxv * p->cap * sizeof(XYZZ)));
    p->gathered_cap = (size_t)p->cap;
    p->rank = rank;
    p->world = world;
    p->shard_lo = (uint32_t)first_point;
    p->shard_n = (uint32_t)p->g->n;
    p->exchange = nullptr;
    p->exchange_user = nullptr;
    p->rccl_comm = nccl_comm;
    return ZG_OK;
}

// create_proof for `nb` circuit instances in lock step: every kernel launch below serves all nb proofs (grid rows /
// vector groups per proof, scalars from d_pc[b]); the host keeps one transcript per proof and hands each its own
// challenges.  nb = 1 is the single-proof path (zg_prover_prove / _dev): there is no other.
// advice_host / advice_dev: per proof, one of them may be given (host columns are uploaded, foreign device columns
// copied into the proof's slot); both null = the slot already holds the columns (zg_prover_advice_slot).
// Host-side timeline of a batch (development aid, compiled in with -DZG_TICKS: `make EXTRA=-DZG_TICKS`): every ZG_TICK
// records a label and the time since the batch began; the list goes to stderr when the batch is done.
#ifdef ZG_TICKS
struct TickLog {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::vector<std::pair<const char*, double>> v;
    void tick(const char* l) { v.emplace_back(l, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count()); }
    ~TickLog() {
        double prev = 0;
        for (auto& e : v) {
            fprintf(stderr, "tick %9.1f us  +%7.1f  %s\n", e.second, e.second - prev, e.first);
            prev = e.second;
        }
    }
};
#define ZG_TICK(l) ticks.tick(l)
#else
#define ZG_TICK(l) ((void)0)
#endif

// One lock-step batch of create_proofs (upstream plonk/prover.rs, `create_proof`): the state its phases share, and ONE member
// function per phase -- round 3 had all of it in a single 510-line function.  A phase queues its kernels
// on the prover's stream(s), waits for what the transcript needs (commitments as points, evaluations), feeds every proof's
// EvmTranscript and leaves the next challenges in p->hpc[] / on the device (upload_consts).  Order and contents of the phases
// are upstream's:
//   load_inputs        instance + advice columns in place, blinding rows, the vanishing argument's random polynomial
//   commit_advice      advice commitments                                   -> theta
//   commit_permuted    lookup::commit_permuted (+ the random polynomial)    -> beta, gamma
//   commit_products    permutation::commit, lookup::commit_product          -> y
//   quotient           evaluate_h, h(X) in pieces, their commitments        -> x
//   evaluations        eval_polynomial at x omega^rot                       -> v
//   openings           ProverGWC::create_proof
//   finish             proof bytes and statuses out
// A nonzero return leaves p->in_flight set: the next batch drains the streams before it reuses the slots.
struct ProveBatch {
    using clk = std::chrono::steady_clock;
    // ---- the call
    zg_prover* p;
    const zg_fr* const* advice_host;
    void* const* advice_dev;
    const zg_fr* const* instance;
    size_t instance_len;
    const uint8_t* keys;
    uint8_t* const* proofs;
    size_t proof_cap;
    size_t* proof_lens;
    int* statuses;
    // ---- shapes
    const PkDev& pk;
    zg_ctx* ctx;
    hipStream_t st;
    uint32_t nb, n, k, ek, bf, usable, A, I, P, NL, S, Q;
    bool hat;
    bool split;         // extended-domain parts of this proof: the split pair in the throughput configuration, the single coset otherwise
    bool phase_cosets;  // the coset forms of a phase's columns: on the side stream while that phase's commitments run (latency
                        // form), or all at once before evaluate_h (throughput form)
    uint32_t dlo, dhi;
    size_t pp_bs, adv_bs, inst_bs, perm_bs, zs_bs, pw_bs, wp_bs;  // strides between consecutive proofs
    // side stream: coefficient / coset forms of committed columns are computed there while the main stream runs the
    // commitment MSM (p->use_side == false: everything stays on the main stream -- the throughput configuration, where other
    // proofs in flight fill the gaps and every extra HIP stream costs a hardware queue)
    zg_ctx* sx;
    hipStream_t ss;
    double parts_en, ext_unit;  // (profile charges: SURVEY.md 8d counts one coeff_to_extended as (n + 2^ext_k) * 32 B whatever parts it is computed on)
    // ---- state carried from phase to phase
    PolySet polys;
    std::vector<EvmTranscript> tr;
    std::vector<int> status;
    std::vector<Jac> pts;
    Fe* random_row = nullptr;  // (proof 0's; proof b's is perm_bs further)
    Fe* adv = nullptr;
    Cols base_cols;
    std::vector<Jac> random_commit;
    bool have_random = false;
    struct Q1 { uint32_t poly, slot; };
    std::vector<int32_t> rots;  // distinct opening points, in any order (the powers table is indexed by slot)
    std::vector<Q1> evq;
    size_t e_fixed = 0, e_random = 0, e_sigma = 0, e_pz = 0, e_lk = 0, e_written = 0, e_h = 0;
    uint32_t npoints = 0;
    uint32_t* d_hlist = nullptr;
    const Fe* ev_all = nullptr;
    uint32_t* h_err = nullptr;  // the lookups' error words on the host
    uint32_t prod_per = 0;      // commitments per proof of the products phase (the random polynomial rides there without lookups)
    uint32_t nsets = 0;         // opening point sets: list of set s at lists[s * 512 ..], its evaluation indices in list order
    std::vector<uint32_t> lists, counts, set_slot;
    std::vector<std::vector<size_t>> set_evs;
    // ---- the gate (run())
    enum Wait { W_ADVICE, W_PERMUTED, W_PRODUCTS, W_QUOTIENT, W_EVALS, W_GWC };
    bool gated = false, armed = false;
    bool no_gate = false;       // (the second run of a proof whose gate gave up)
    bool gate_gave_up = false;  // finish(): a gate kernel ran into its time limit -- the proof was made on stale scalars
    uint32_t gates_armed = 0;
    void* gate_slot = nullptr;
    zg_ctx::GateHold hold;  // what a blocking call inside a queued-ahead phase needs to let the gate go (common.h)
    clk::time_point t_start, t_prev;
#ifdef ZG_TICKS
    TickLog ticks;
#endif

    ProveBatch(zg_prover* p_, size_t count, const zg_fr* const* advice_host_, void* const* advice_dev_, const zg_fr* const* instance_,
               size_t instance_len_, const uint8_t* keys_, uint8_t* const* proofs_, size_t proof_cap_, size_t* proof_lens_, int* statuses_)
        : p(p_), advice_host(advice_host_), advice_dev(advice_dev_), instance(instance_), instance_len(instance_len_), keys(keys_),
          proofs(proofs_), proof_cap(proof_cap_), proof_lens(proof_lens_), statuses(statuses_), pk(*p_->pk), ctx(p_->ctx),
          st(p_->ctx->stream), nb((uint32_t)count), tr(count), status(count, ZG_OK) {
        n = pk.n; k = pk.k; ek = pk.ext_k; bf = pk.bf; usable = pk.usable;
        A = pk.A; I = pk.I; P = pk.P; NL = pk.NL; S = pk.sets; Q = pk.qpd;
        hat = pk.hat;
        split = pk.nparts == 3 && (!p->use_side || p->lat_split);
        phase_cosets = p->use_side || !split;
        dlo = split ? 1u : 0u;
        dhi = split ? 3u : 1u;
        pp_bs = (size_t)p->npp * n; adv_bs = (size_t)A * n; inst_bs = (size_t)I * n; perm_bs = (size_t)(2 * NL + 1) * n;
        zs_bs = (size_t)(S + NL + 1) * n; pw_bs = (size_t)p->max_points * n; wp_bs = (size_t)2 * p->max_points * n;
        polys.sh = pk.sh_polys; polys.pp = p->pp; polys.nsh = p->nsh; polys.n = n; polys.pp_bs = pp_bs;
        sx = p->use_side ? ctx->side : ctx;
        ss = sx->stream;
        parts_en = 0.0;
        for (uint32_t di = dlo; di < dhi; di++) parts_en += (double)pk.dom[di].en;
        ext_unit = ((double)n + (double)((size_t)1 << ek)) * 32.0;
        t_start = t_prev = clk::now();
    }

    Fe* pp_at(uint32_t ix) const { return p->pp + (size_t)(ix - p->nsh) * n; }  // proof 0's polynomial ix (>= nsh)
    void lap(int slot) {
        auto now = clk::now();
        p->phase_ms[slot] = std::chrono::duration<double, std::milli>(now - t_prev).count();
        t_prev = now;
    }
    int fork() {  // side stream continues after everything queued on the main stream so far
        if (!p->use_side) return ZG_OK;
        ZG_HIP(hipEventRecord(p->ev_fork, st));
        ZG_HIP(hipStreamWaitEvent(ss, p->ev_fork, 0));
        return ZG_OK;
    }
    int join() {  // main stream continues after everything queued on the side stream so far
        if (!p->use_side) return ZG_OK;
        ZG_HIP(hipEventRecord(p->ev_join, ss));
        ZG_HIP(hipStreamWaitEvent(st, p->ev_join, 0));
        return ZG_OK;
    }
    // two-level layouts of the transforms: `per` arrays per proof
    static Grouping grouping(uint32_t per, size_t in_outer, size_t out_outer) {
        Grouping g;
        g.per = per; g.in_outer = in_outer; g.out_outer = out_outer;
        return g;
    }
    // coefficient forms (ix0 .. ix0 + per) of every proof -> their slabs on each part of the extended domain
    int to_cosets(zg_ctx* c, uint32_t ix0, uint32_t per) {
        for (uint32_t di = dlo; di < dhi; di++) {
            const PkDev::Dom& d = pk.dom[di];
            const Grouping g = grouping(per, pp_bs, (size_t)p->ncos * d.en);
            c->unit_next = (double)nb * per * ext_unit * ((double)d.en / parts_en);
            ZG_TRY(coeff_to_coset_dev(c, pp_at(ix0), n, n, p->dbuf[di].cos + (size_t)(ix0 - p->ix_adv) * d.en, d.en, (size_t)nb * per,
                                      d.ek, hat, d.zpow, &g));
        }
        return ZG_OK;
    }

    // The phases of create_proof, each in two halves: X_queue() puts the phase's device work on the streams (it needs only
    // the circuit and what the device already holds -- every challenge reaches the kernels through d_pc), X_absorb() waits
    // for the phase's results, writes them into the transcripts and PUBLISHES the next challenge (publish(): hpc -> d_pc).
    //
    // Plain order: queue, absorb, queue, absorb ...  A lone proof with the gate (ZG_LAT_GATE) queues the NEXT phase before
    // it absorbs this one: arm() reserves a staging slot and puts gate_pull_kernel on the stream (it waits for the gate
    // word, then copies the slot to d_pc), the next phase's launches follow behind it, and publish() fills the slot and
    // opens the gate -- between the host having a challenge and the device using it stands one store, not a launch
    // sequence.  What follows the grand products' totals is never queued ahead: it starts with the HOST inverting them.
    int run() {
        const int st_run = run_phases();
        p->gate_stats[0] += gated ? 1 : 0;
        p->gate_stats[1] += gates_armed;
        p->gate_stats[3] += hold.yields;
        // An error that surfaced BEHIND a failed gate (a time limit, a yield) is the stale scalars' doing, not the caller's:
        // drain what was queued and let prove_batch_impl make the proof again in the plain order.
        if (st_run != ZG_OK && !gate_gave_up && gate_failed()) {
            open_gate();
            (void)hipStreamSynchronize(st);
            if (ctx->side) (void)hipStreamSynchronize(ctx->side->stream);
            p->in_flight = false;
            gate_gave_up = true;
        }
        if (gate_gave_up) p->gate_stats[2]++;
        if (st_run != ZG_OK) p->warm_sig = 0;  // (whatever way a proof failed: the next one is a first proof)
        return st_run;
    }
    int run_phases() {
        begin();
        ZG_TRY(load_inputs());
        gated = gate_wanted();
        ZG_TRY(advice_queue());
        if (!gated) {
            ZG_TRY(advice_absorb());
            ZG_TRY(permuted_queue());
            ZG_TRY(permuted_absorb());
            ZG_TRY(products_terms_queue());
            ZG_TRY(products_queue());
            ZG_TRY(products_absorb());
            ZG_TRY(evaluation_lists());
            ZG_TRY(quotient_queue());
            ZG_TRY(quotient_absorb());
            ZG_TRY(evaluations_queue());
            ZG_TRY(openings_lists());
            ZG_TRY(evaluations_absorb());
            ZG_TRY(openings_queue());
        } else {
            ZG_TRY(arm());
            ZG_TRY(permuted_queue());    // behind theta
            ZG_TRY(advice_absorb());
            ZG_TRY(arm());
            ZG_TRY(products_terms_queue());  // behind beta, gamma
            ZG_TRY(permuted_absorb());
            ZG_TRY(products_queue());  // (starts with the host's share of the grand products: never queued ahead)
            ZG_TRY(evaluation_lists());
            ZG_TRY(arm());
            ZG_TRY(quotient_queue());    // behind y
            ZG_TRY(products_absorb());
            ZG_TRY(arm());
            ZG_TRY(evaluations_queue());  // behind x
            ZG_TRY(quotient_absorb());
            ZG_TRY(openings_lists());
            ZG_TRY(arm());
            ZG_TRY(openings_queue());    // behind v
            ZG_TRY(evaluations_absorb());
        }
        ZG_TRY(openings_absorb());
        return finish();
    }

    // ---- the gate
    uint64_t form_sig() const {
        // (everything that shapes the proof's allocation requests and launch sequence: a proof whose signature differs from
        //  the last completed one's is a FIRST proof -- it may create tables and workspace, with synchronisations -- and is
        //  never gated: scheduling form, domain split, digit tables, batch, instance length, and the tuning generation,
        //  which every zg_tuning_set bumps)
        return 1u | (uint64_t)p->use_side << 1 | (uint64_t)split << 2 | (uint64_t)(p->g->full_table.load() != nullptr) << 3 |
               (uint64_t)(p->gl->full_table.load() != nullptr) << 4 | (uint64_t)ctx->msm_pair << 5 | (uint64_t)(nb & 0xFFu) << 8 |
               (uint64_t)(instance_len & 0xFFFFu) << 16 | (uint64_t)tuning_generation() << 32;
    }
    bool gate_wanted() {
        const int v = knob(K_LAT_GATE);
        if (no_gate || (v < 0 ? LAT_GATE_DEFAULT : v) == 0 || !p->use_side || nb != 1 || p->world > 1 || p->rccl_comm) return false;
        if (p->warm_sig != form_sig()) return false;  // (first-use allocations and their synchronisations are behind us)
        // (a runtime that completes every launch before it submits the next one would never reach publish())
        const bool serialising = runtime_serialises_launches();
        // room in the staging arena for the five gated uploads and the index lists behind them
        const size_t need = 5 * ((size_t)nb * sizeof(ProofConst) + 64) + (256u << 10);
        if (serialising || sizeof(ProofConst) % 16 != 0 || p->stage_off + need > p->pinned_cap - 4096) return false;
        p->gate_word()[16] = 0;  // (no gate kernel is in flight between proofs)
        return true;
    }
    int arm() {
        const size_t bytes = (size_t)nb * sizeof(ProofConst);
        const size_t off = (p->stage_off + 63) & ~size_t(63);
        // (gate_wanted() checked the arena's room: a phase queued ahead WITHOUT its gate would run on the previous challenge)
        ZG_REQUIRE(gated && !armed && off + bytes <= p->pinned_cap - 4096, ZG_ERR_INVALID_ARG, "zg_prover_prove: no room to arm the gate");
        gate_slot = (char*)p->pinned + off;
        p->stage_off = off + bytes;
        if (++p->gate_seq == 0) ++p->gate_seq;
        armed = true;  // (from here on somebody has to open it: ~ProveBatch)
        hold.word = p->gate_word();
        hold.seq = p->gate_seq;
        ctx->gate_hold = &hold;  // (... or gate_yield(), from a blocking call of the phase queued behind it)
        if (ctx->side) ctx->side->gate_hold = &hold;
        uint32_t* gate_dev = reinterpret_cast<uint32_t*>((char*)p->pinned_dev + p->pinned_cap - 128);
        const uint4* dev_view = reinterpret_cast<const uint4*>((const char*)p->pinned_dev + off);
        // (ZG_LAT_GATE=2, for the tests: the proof's first gate is never opened by publish() and gives up after 0.2 s)
        const bool lost = knob(K_LAT_GATE) == 2 && gates_armed == 0;
        gates_armed++;
        ZG_LAUNCH(ctx, "gate_pull", (double)bytes * 2, gate_pull_kernel, dim3(1), dim3(256), 0, gate_dev, lost ? p->gate_seq ^ 0x80000000u : p->gate_seq,
                  gate_dev + 16, lost ? GATE_MAX_TICKS / 20 : GATE_MAX_TICKS, dev_view, (uint4*)p->d_pc, (uint32_t)(bytes / 16));
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    void open_gate() {
        if (!armed) return;
        __atomic_store_n(p->gate_word(), p->gate_seq, __ATOMIC_SEQ_CST);
        armed = false;
        hold.word = nullptr;
    }
    // a gate kernel ran into its time limit, or a blocking call let one go early: phases ran on the previous challenge
    bool gate_failed() const { return gated && (p->gate_word()[16] != 0 || hold.yielded); }
    // the per-proof scalars as the host holds them now -> d_pc: through the armed gate, else behind the work already queued
    int publish() {
        if (!armed) return upload_consts(p, nb);
        memcpy(gate_slot, p->hpc.data(), (size_t)nb * sizeof(ProofConst));
        open_gate();  // (a gate that gate_yield() already opened copied the previous scalars: gate_failed(), the proof is re-made)
        return ZG_OK;
    }
    // (an error return between arm() and publish(): the queue must drain whatever it then computes)
    ~ProveBatch() {
        open_gate();
        if (ctx->gate_hold == &hold) ctx->gate_hold = nullptr;
        if (ctx->side && ctx->side->gate_hold == &hold) ctx->side->gate_hold = nullptr;
    }

    void begin() {
        p->have_last = false;
        if (p->in_flight) {  // the previous batch left through an error return: drain what it queued before its staging
            (void)hipStreamSynchronize(st);  // arena and slots are reused
            if (ctx->side) (void)hipStreamSynchronize(ctx->side->stream);
        }
        p->in_flight = true;
        p->stage_off = p->pin_stage;
    }

    int load_inputs() {
        // ---- advice columns into their slots
        for (uint32_t b = 0; b < nb; b++) {
            Fe* slot = p->adv_val + b * adv_bs;
            if (!A) break;
            if (advice_host && advice_host[b]) {
                ZG_HIP(hipMemcpyAsync(slot, advice_host[b], adv_bs * 32, hipMemcpyHostToDevice, st));
            } else if (advice_dev && advice_dev[b] && advice_dev[b] != (void*)slot) {
                ZG_HIP(hipMemcpyAsync(slot, advice_dev[b], adv_bs * 32, hipMemcpyDeviceToDevice, st));
            }
        }

        // ---- vk + instance values into the transcripts; instance polynomials
        for (uint32_t b = 0; b < nb; b++) {
            memset(&p->hpc[b], 0, sizeof(ProofConst));
            memcpy(p->hpc[b].key, keys + 32 * (size_t)b, 32);
            tr[b].common_scalar(pk.vk_repr);
        }
        ZG_TRY(upload_consts(p, nb));
        // vanishing::Argument::commit's random polynomial depends on no challenge: generate it now and
        // commit it inside the permuted-lookup batch (coefficient basis `g` next to `g_lagrange` vectors)
        random_row = p->perm + (size_t)(2 * NL) * n;  // (proof 0's; proof b's is perm_bs further)
        // (the same launch draws the blinding rows of the advice columns: commit_lagrange's input below)
        adv = p->adv_val;
        ZG_TRY(poly_random_and_blind(ctx, p->d_pc, nb, random_row, perm_bs, pp_at(p->ix_random), pp_bs, n, TAG_RANDOM_POLY, adv, adv_bs,
                                     n, A, usable, bf + 1, TAG_ADVICE_BLIND));
        if (I) {
            for (uint32_t b = 0; b < nb; b++) {
                Fe* iv = p->inst_val + b * inst_bs;
                if (p->inst_filled[b] > instance_len) ZG_HIP(hipMemsetAsync(iv, 0, inst_bs * 32, st));  // (zeroed at create)
                p->inst_filled[b] = instance_len;
                for (uint32_t c = 0; c < I; c++) {
                    const zg_fr* src = instance_len ? instance[b] + (size_t)c * instance_len : nullptr;
                    for (size_t i = 0; i < instance_len; i++) tr[b].common_scalar(to_fe(&src[i]));
                    if (instance_len) ZG_TRY(h2d(p, iv + (size_t)c * n, src, instance_len * 32));
                }
            }
        }
        base_cols.fixed = pk.fixed_val; base_cols.advice = adv; base_cols.instance = p->inst_val;
        base_cols.log_size = k; base_cols.rot_scale = 1;
        base_cols.adv_bs = adv_bs; base_cols.inst_bs = inst_bs;
        return ZG_OK;
    }

    int advice_queue() {
        // ---- advice: commit (Lagrange basis)
        ZG_TRY(fork());
        if (I) {
            const Grouping g = grouping(I, inst_bs, pp_bs);
            ZG_TRY(ntt_batch_to_dev(sx, p->inst_val, pp_at(p->ix_inst), n, (size_t)nb * I, k, pk.omega_inv, &pk.ifft_div, &g));
            if (phase_cosets) ZG_TRY(to_cosets(sx, p->ix_inst, I));
        }
        if (A) {
            ZG_TRY(commit(p, p->gl, nullptr, A, adv, n, A, adv_bs, (size_t)nb * A, 0, 0, W_ADVICE));
            const Grouping g = grouping(A, adv_bs, pp_bs);
            ZG_TRY(ntt_batch_to_dev(sx, adv, pp_at(p->ix_adv), n, (size_t)nb * A, k, pk.omega_inv, &pk.ifft_div, &g));
            if (phase_cosets) ZG_TRY(to_cosets(sx, p->ix_adv, A));
        }
        ZG_TICK("advice: queued");
        return ZG_OK;
    }
    int advice_absorb() {
        if (A) {
            ZG_TRY(wait_points(p, (size_t)nb * A, pts, W_ADVICE));
            ZG_TICK("advice: points on the host");
            for (uint32_t b = 0; b < nb; b++)
                for (uint32_t c = 0; c < A; c++) tr[b].write_point(pts[(size_t)b * A + c]);
        }
        for (uint32_t b = 0; b < nb; b++) p->hpc[b].theta = tr[b].squeeze();
        ZG_TICK("theta");
        ZG_TRY(publish());
        ZG_TICK("theta uploaded");
        lap(0);
        return ZG_OK;
    }

    int permuted_queue() {
        // ---- lookups: commit_permuted (+ the random polynomial's commitment)
        if (NL) {
            // permute_expression_pair on the device: canonical keys (written by the compression kernel itself, with
            // the sentinel padding), bitonic sort of inputs and tables, scan-based construction of s' (sort.hip).
            // raw rows [0, m) = inputs -> a', [m, 2m) = tables, m = nb * NL, row b * NL + l = lookup l of proof b.
            const uint32_t m = nb * NL;
            Fe *raw_in = p->raw, *raw_tab = p->raw + (size_t)m * n;
            uint32_t* d_err = p->sort_u32 + (size_t)2 * m * n + 2 * m;  // behind permute_pairs' scratch: zeroed by the same fill
            ZG_TRY(poly_lookup_compress(ctx, pk.dc, base_cols, p->d_pc, nb, p->cin, p->ctab, n, raw_in, raw_tab, usable));
            auto t_sort = clk::now();
            ZG_TRY(poly_sort_keys(ctx, p->raw, n, 2 * m));
            ZG_TRY(poly_permute_pairs(ctx, raw_in, raw_tab, p->sraw, n, usable, m, p->sort_u32, p->sort_fe, d_err));
            // perm[2l] = a'_l, perm[2l+1] = s'_l (Montgomery form) on the usable rows, then the blinding tail
            // (blinding: a' rows get tag 2, s' rows tag 3, index = lookup * (bf+1) + j)
            ZG_TRY(poly_permuted_finish(ctx, p->d_pc, nb, raw_in, p->sraw, p->perm, perm_bs, n, usable, bf + 1, NL, TAG_PERMUTED_INPUT,
                                        TAG_PERMUTED_TABLE));
            p->phase_ms[7] = std::chrono::duration<double, std::milli>(clk::now() - t_sort).count();
            ZG_TRY(fork());
            // the lookups' error words leave on the side stream, beside the commitments (an event of their own)
            h_err = reinterpret_cast<uint32_t*>((char*)p->pinned + p->pin_evals + (size_t)p->cap * p->max_evals * sizeof(Fe));
            ZG_HIP(hipMemcpyAsync(h_err, d_err, m * sizeof(uint32_t), hipMemcpyDeviceToHost, ss));
            ZG_HIP(hipEventRecord(p->ev_err, ss));
            // (a' and s' are sorted: equal neighbours everywhere, so the run form leaves one entry per distinct value)
            const zg_bases *cgl = naf_of(p, p->gl), *cg = cgl == p->gl ? p->g : naf_of(p, p->g);  // (both or neither)
            const uint64_t sorted_runs = cgl->run_table && 2 * NL < 64 ? (1ull << (2 * NL)) - 1ull : 0ull;
            ZG_TRY(commit(p, cgl, cg, 2 * NL, p->perm, n, 2 * NL + 1, perm_bs, (size_t)nb * (2 * NL + 1), sorted_runs, naf_gl_width(p), W_PERMUTED));
            {
                const Grouping g = grouping(2 * NL, perm_bs, pp_bs);
                ZG_TRY(ntt_batch_to_dev(sx, p->perm, pp_at(p->ix_perm), n, (size_t)nb * 2 * NL, k, pk.omega_inv, &pk.ifft_div, &g));
            }
            if (phase_cosets) ZG_TRY(to_cosets(sx, p->ix_perm, 2 * NL));
        }
        ZG_TICK("permuted: queued");
        return ZG_OK;
    }
    int permuted_absorb() {
        random_commit.resize(nb);
        have_random = false;
        if (NL) {
            ZG_TRY(wait_points(p, (size_t)nb * (2 * NL + 1), pts, W_PERMUTED));
            ZG_HIP(hipEventSynchronize(p->ev_err));
            ZG_TICK("permuted: points on the host");
            for (uint32_t b = 0; b < nb; b++) {
                for (uint32_t l = 0; l < NL; l++)
                    if (h_err[b * NL + l] && status[b] == ZG_OK) {
                        set_error("zg_prover_prove: lookup %u of proof %u has an input outside its table (ConstraintSystemFailure)", l, b);
                        status[b] = ZG_ERR_CONSTRAINT;
                    }
                const Jac* q = &pts[(size_t)b * (2 * NL + 1)];
                for (uint32_t i = 0; i < 2 * NL; i++) tr[b].write_point(q[i]);
                random_commit[b] = q[2 * NL];
            }
            have_random = true;
            if (nb == 1 && status[0] != ZG_OK && gate_failed()) return ZG_ERR_HIP;  // (error words from behind a failed gate: run() re-makes the proof)
            if (nb == 1 && status[0] != ZG_OK) {  // a lone proof stops here, as upstream's `?` does
                open_gate();  // (whatever was queued ahead runs out on stale scalars: nobody reads its results)
                (void)hipStreamSynchronize(ss);
                (void)hipStreamSynchronize(st);
                if (statuses) statuses[0] = status[0];
                proof_lens[0] = 0;
                p->in_flight = false;
                return status[0];
            }
        }
        for (uint32_t b = 0; b < nb; b++) {
            p->hpc[b].beta = tr[b].squeeze();
            p->hpc[b].gamma = tr[b].squeeze();
        }
        ZG_TICK("beta, gamma");
        ZG_TRY(publish());
        ZG_TICK("beta, gamma uploaded");
        lap(1);
        return ZG_OK;
    }

    // ---- permutation products (sets chained through z[n - bf - 1]) and lookup products
    // (in two parts: the terms and the running products up to their totals need beta and gamma only; what follows the
    //  totals begins, in the latency form, with the HOST inverting them -- it cannot be queued ahead of anything)
    int products_terms_queue() {
        const uint32_t mb = S + NL;
        if (S) ZG_TRY(poly_perm_terms(ctx, pk.dc, base_cols, p->d_pc, nb, pk.sigma_val, pk.omega_tw, p->num, p->den, mb, n));
        // (a'_l / s'_l are interleaved in `perm`: two views with a stride of two columns)
        ZG_TRY(poly_lookup_terms(ctx, p->d_pc, nb, p->cin, p->ctab, p->perm, p->perm + n, (size_t)2 * n, perm_bs, p->num, p->den, mb, S, n, NL));
        // all running products of the batch in one scan sequence; per proof the S permutation sets are chained
        // through row n - bf - 1, the lookup products start from one
        if (mb) ZG_TRY(poly_grand_product(ctx, p->num, p->den, nullptr, p->zs, p->tmp, n, nb * mb, S, n - bf - 1, mb, zs_bs, 1));
        ZG_TICK("product terms: queued");
        return ZG_OK;
    }
    int products_queue() {
        const uint32_t mb = S + NL;
        if (mb) {
            ZG_TRY(poly_grand_product(ctx, p->num, p->den, nullptr, p->zs, p->tmp, n, nb * mb, S, n - bf - 1, mb, zs_bs, 2));
            ZG_TRY(poly_blind_rows2(ctx, p->d_pc, nb, p->zs, zs_bs, n, S, TAG_PERM_Z, NL, TAG_LOOKUP_Z, n - bf, bf));  // (lz follows pz)
            // The products stay constant wherever a row changes nothing (every padding row of the circuit): they are
            // committed in the run form, sum_i (z_i - z_{i+1}) Q_i over the running sums Q of g_lagrange.
            const zg_bases *cgl = naf_of(p, p->gl), *cg = cgl == p->gl ? p->g : naf_of(p, p->g);
            const uint64_t z_runs = cgl->run_table && mb < 64 ? (1ull << mb) - 1ull : 0ull;
            ZG_TRY(fork());
            prod_per = mb;
            if (have_random) {
                ZG_TRY(commit(p, cgl, nullptr, mb, p->zs, n, mb, zs_bs, (size_t)nb * mb, z_runs, naf_gl_width(p), W_PRODUCTS));
            } else {  // no lookups: the random polynomial rides here instead (row mb of zs)
                for (uint32_t b = 0; b < nb; b++)
                    ZG_HIP(hipMemcpyAsync(p->zs + b * zs_bs + (size_t)mb * n, random_row + b * perm_bs, (size_t)n * 32, hipMemcpyDeviceToDevice, st));
                prod_per = mb + 1;
                ZG_TRY(commit(p, cgl, cg, mb, p->zs, n, prod_per, zs_bs, (size_t)nb * prod_per, z_runs, naf_gl_width(p), W_PRODUCTS));
            }
            {
                const Grouping g = grouping(mb, zs_bs, pp_bs);
                ZG_TRY(ntt_batch_to_dev(sx, p->zs, pp_at(p->ix_pz), n, (size_t)nb * mb, k, pk.omega_inv, &pk.ifft_div, &g));
            }
            if (phase_cosets) ZG_TRY(to_cosets(sx, p->ix_pz, mb));
        } else if (!have_random) {  // neither lookups nor permutation: commit the random polynomial on its own
            ZG_TRY(commit(p, p->g, nullptr, 1, random_row, n, 1, perm_bs, nb, 0, 0, W_PRODUCTS));
        }
        ZG_TICK("products: queued");
        return ZG_OK;
    }
    int products_absorb() {
        const uint32_t mb = S + NL;
        if (mb) {
            ZG_TRY(wait_points(p, (size_t)nb * prod_per, pts, W_PRODUCTS));
            ZG_TICK("products: points on the host");
            for (uint32_t b = 0; b < nb; b++) {
                const Jac* q = &pts[(size_t)b * prod_per];
                for (uint32_t i = 0; i < mb; i++) tr[b].write_point(q[i]);
                if (!have_random) random_commit[b] = q[mb];
            }
        } else if (!have_random) {
            ZG_TRY(wait_points(p, nb, pts, W_PRODUCTS));
            for (uint32_t b = 0; b < nb; b++) random_commit[b] = pts[b];
        }
        have_random = true;
        for (uint32_t b = 0; b < nb; b++) tr[b].write_point(random_commit[b]);
        for (uint32_t b = 0; b < nb; b++) evalh_consts(p->hpc[b], tr[b].squeeze(), hat, evalh_terms(pk));
        ZG_TICK("y");
        ZG_TRY(publish());
        ZG_TICK("y uploaded");
        lap(2);
        return ZG_OK;
    }

    int quotient_queue() {
        ZG_TRY(join());  // evaluate_h reads every coset the side stream produced
        // (throughput configuration: nothing overlaps, so every witness polynomial goes to its cosets here, in one batch per
        //  coset, instead of phase by phase)
        if (!phase_cosets) ZG_TRY(to_cosets(ctx, p->ix_adv, p->ncos));
        // ---- evaluate_h (+ division by X^n - 1) on every part of the extended domain, back to coefficients, h pieces
        for (uint32_t di = dlo; di < dhi; di++) {
            const EvalHArgs a = evalh_args(p, di);
            ZG_TRY(poly_evaluate_h(ctx, a, pk.dom[di].en, nb, A + I + pk.F, (double)((size_t)1 << ek) * ((double)pk.dom[di].en / parts_en)));
        }
        p->have_last = true;
        p->last_split = split;
        p->last_nb = nb;
        const double ext_inv_unit = (double)nb * 2.0 * (double)((size_t)1 << ek) * 32.0;  // (ext -> coeff, 2 * 8n * 32 B)
        if (!split) {
            ctx->unit_next = ext_inv_unit;
            ZG_TRY(coset_to_coeff_dev(ctx, p->dbuf[0].h, ek, (size_t)Q * n, pp_at(p->ix_hpiece), hat, 1, nb, pk.dom[0].en, pp_bs));
        } else {
            // h = A + (X^L1 - c1) B:  A (degree < L1) from the first coset, where X^L1 = c1 = shift1^L1;  B (degree < L2)
            // from the second, where X^L1 = c2 and X^L2 = e are constants too:  B = (h - A) / (c2 - c1) there, with A
            // folded modulo X^L2 - e before it is evaluated on those L2 points.
            const PkDev::Dom &d1 = pk.dom[1], &d2 = pk.dom[2];
            const uint32_t L1 = d1.en, L2 = d2.en;
            const Fe zeta = fr_zeta(), zeta2 = Fr::sqr(zeta);
            const Fe c1 = Fr::pow_u64(zeta, L1), c2 = Fr::pow_u64(zeta2, L1), e = Fr::pow_u64(zeta2, L2);
            Fe* hp = pp_at(p->ix_hpiece);
            const size_t tb = (size_t)3 * L2;
            Fe *fold = p->split_tmp, *a2 = fold + L2, *bc = a2 + L2;
            ctx->unit_next = ext_inv_unit;  // (the three transforms of the split form stand for ONE extended_to_coeff)
            ZG_TRY(coset_to_coeff_dev(ctx, p->dbuf[1].h, d1.ek, L1, hp, hat, 1, nb, L1, pp_bs));  // A, in place of the low pieces
            ZG_TRY(poly_fold(ctx, nb, hp, pp_bs, L2, L1 / L2, e, fold, tb));                       // A mod (X^L2 - e)
            ctx->unit_next = 0.0;
            ZG_TRY(coeff_to_coset_dev(ctx, fold, tb, L2, a2, tb, nb, d2.ek, false, 2));            // A on the second coset
            const Fe unhat = hat ? Fr::inv(Fr::from_u64(32)) : Fr::one();
            ZG_TRY(poly_diff_scale(ctx, nb, p->dbuf[2].h, L2, unhat, a2, tb, Fr::inv(Fr::sub(c2, c1)), a2, tb, L2));  // B on the second coset
            ctx->unit_next = 0.0;
            ZG_TRY(coset_to_coeff_dev(ctx, a2, d2.ek, L2, bc, false, 2, nb, tb, tb));              // B
            ZG_TRY(poly_split_combine(ctx, nb, hp, pp_bs, bc, tb, L2, c1, L1));                    // h = A - c1 B + X^L1 B
        }
        ctx->msm_dense_hint = true;  // (the quotient pieces are random vectors: every digit of every window is an addition)
        const int st_h = commit(p, dense_g(p), nullptr, Q, pp_at(p->ix_hpiece), n, Q, pp_bs, (size_t)nb * Q, 0, 0, W_QUOTIENT);
        ctx->msm_dense_hint = false;
        ZG_TRY(st_h);
        ZG_TICK("h: queued");
        return ZG_OK;
    }
    // (h's commitments, then x and the opening points: needs evaluation_lists())
    int quotient_absorb() {
        ZG_TRY(wait_points(p, (size_t)nb * Q, pts, W_QUOTIENT));
        ZG_TICK("h: points on the host");
        for (uint32_t b = 0; b < nb; b++)
            for (uint32_t i = 0; i < Q; i++) tr[b].write_point(pts[(size_t)b * Q + i]);
        for (uint32_t b = 0; b < nb; b++) {
            ProofConst& c = p->hpc[b];
            const Fe x = tr[b].squeeze();
            c.xn = Fr::pow_u64(x, n);
            for (uint32_t i = 0; i < npoints; i++) c.points[i] = rotate_omega(pk, x, rots[i]);
        }
        ZG_TICK("x");
        ZG_TRY(publish());
        ZG_TICK("x uploaded");
        lap(3);
        return ZG_OK;
    }

    // ---- evaluations: which polynomial is evaluated at which opening point (circuit only)
    int evaluation_lists() {
        // distinct opening points, in any order (the powers table is indexed by slot)
        rots = {0, 1, -1, -(int32_t)(bf + 1)};
        auto rot_slot = [&](int32_t r) -> uint32_t {
            for (size_t i = 0; i < rots.size(); i++)
                if (rots[i] == r) return (uint32_t)i;
            rots.push_back(r);
            return (uint32_t)rots.size() - 1;
        };
        evq.clear();  // evaluations in transcript order, then h_poly at x
        for (auto& q : pk.advice_queries) evq.push_back({p->ix_adv + q.column, rot_slot(q.rotation)});
        e_fixed = evq.size();
        for (auto& q : pk.fixed_queries) evq.push_back({p->ix_fixed + q.column, rot_slot(q.rotation)});
        e_random = evq.size();
        evq.push_back({p->ix_random, 0});
        e_sigma = evq.size();
        for (uint32_t c = 0; c < P; c++) evq.push_back({p->ix_sigma + c, 0});
        e_pz = evq.size();
        for (uint32_t s = 0; s < S; s++) {
            evq.push_back({p->ix_pz + s, 0});
            evq.push_back({p->ix_pz + s, 1});
            if (s + 1 < S) evq.push_back({p->ix_pz + s, 3});
        }
        e_lk = evq.size();
        for (uint32_t l = 0; l < NL; l++) {
            evq.push_back({p->ix_lz + l, 0});            // z(x)
            evq.push_back({p->ix_lz + l, 1});            // z(omega x)
            evq.push_back({p->ix_perm + 2 * l, 0});      // a'(x)
            evq.push_back({p->ix_perm + 2 * l, 2});      // a'(omega^-1 x)
            evq.push_back({p->ix_perm + 2 * l + 1, 0});  // s'(x)
        }
        e_written = evq.size();
        evq.push_back({p->ix_hpoly, 0});
        e_h = e_written;
        npoints = (uint32_t)rots.size();
        ZG_REQUIRE(npoints <= p->max_points, ZG_ERR_UNSUPPORTED, "zg_prover_prove: %u distinct rotations are queried (max %u)", npoints,
                   p->max_points);
        ZG_REQUIRE(evq.size() <= p->max_evals, ZG_ERR_UNSUPPORTED, "zg_prover_prove: too many evaluations");

        return ZG_OK;
    }
    int evaluations_queue() {
        // vanishing.evaluate: h(X) = sum_i xn^i h_i(X)
        d_hlist = p->d_idx + (size_t)4 * p->max_evals;
        {
            std::vector<uint32_t> list(Q);
            for (uint32_t i = 0; i < Q; i++) list[i] = p->ix_hpiece + (Q - 1 - i);
            ZG_TRY(h2d_list(p, d_hlist, list));
            ZG_TRY(poly_horner_combine_xn(ctx, polys, p->d_pc, nb, d_hlist, Q, pp_at(p->ix_hpoly), pp_bs, n));
        }
        ZG_TRY(poly_powers(ctx, p->d_pc, nb, npoints, n, p->pw, pw_bs));
        std::vector<uint32_t> idx(2 * evq.size());
        for (size_t i = 0; i < evq.size(); i++) {
            idx[i] = evq[i].poly;
            idx[evq.size() + i] = evq[i].slot;
        }
        ZG_TRY(h2d_list(p, p->d_idx, idx));
        // (the evaluations too are written where the host reads them: no copy command behind the kernel)
        uint32_t distinct_polys = 0;
        {
            std::vector<uint32_t> seen(idx.begin(), idx.begin() + evq.size());
            std::sort(seen.begin(), seen.end());
            distinct_polys = (uint32_t)(std::unique(seen.begin(), seen.end()) - seen.begin());
        }
        ZG_TRY(poly_dot(ctx, polys, nb, n, p->d_idx, p->d_idx + evq.size(), p->pw, pw_bs, (uint32_t)evq.size(),
                        reinterpret_cast<Fe*>((char*)p->pinned_dev + p->pin_evals), p->max_evals, distinct_polys, npoints));
        ev_all = reinterpret_cast<const Fe*>((char*)p->pinned + p->pin_evals);
        ZG_HIP(hipEventRecord(p->evs[W_EVALS], st));
        ZG_TICK("evals: queued");
        return ZG_OK;
    }
    // (the evaluations into the transcripts, then v and each point set's v-weighted evaluation: needs openings_lists())
    int evaluations_absorb() {
        ZG_HIP(hipEventSynchronize(p->evs[W_EVALS]));
        ZG_TICK("evals on the host");
        for (uint32_t b = 0; b < nb; b++) {
            const Fe* ev = ev_all + (size_t)b * p->max_evals;
            for (size_t i = 0; i < e_written; i++) tr[b].write_scalar(ev[i]);
            ProofConst& c = p->hpc[b];
            c.v = tr[b].squeeze();
            for (uint32_t s = 0; s < nsets; s++) {
                Fe eval_batch = fe_zero();
                for (size_t e : set_evs[s]) eval_batch = Fr::add(Fr::mul(eval_batch, c.v), ev[e]);
                c.subs[s] = eval_batch;
            }
        }
        ZG_TICK("v");
        ZG_TRY(publish());
        ZG_TICK("v uploaded");
        lap(4);
        return ZG_OK;
    }

    int openings_lists() {
        // ---- opening queries in create_proof's order: (poly, point slot, index of the evaluation)
        struct OQ { uint32_t poly, slot; size_t ev; };
        std::vector<OQ> oq;
        for (size_t i = 0; i < e_fixed; i++) oq.push_back({evq[i].poly, evq[i].slot, i});
        {
            size_t e = e_pz;
            std::vector<size_t> e_last(S, 0), e_cur(S, 0), e_next(S, 0);
            for (uint32_t s = 0; s < S; s++) {
                e_cur[s] = e++;
                e_next[s] = e++;
                if (s + 1 < S) e_last[s] = e++;
            }
            for (uint32_t s = 0; s < S; s++) {
                oq.push_back({p->ix_pz + s, 0, e_cur[s]});
                oq.push_back({p->ix_pz + s, 1, e_next[s]});
            }
            for (uint32_t s = S; s-- > 0;) {
                if (s + 1 == S) continue;
                oq.push_back({p->ix_pz + s, 3, e_last[s]});
            }
        }
        for (uint32_t l = 0; l < NL; l++) {
            const size_t e5 = e_lk + 5 * l;
            oq.push_back({p->ix_lz + l, 0, e5 + 0});
            oq.push_back({p->ix_perm + 2 * l, 0, e5 + 2});
            oq.push_back({p->ix_perm + 2 * l + 1, 0, e5 + 4});
            oq.push_back({p->ix_perm + 2 * l, 2, e5 + 3});
            oq.push_back({p->ix_lz + l, 1, e5 + 1});
        }
        for (size_t i = e_fixed; i < e_random; i++) oq.push_back({evq[i].poly, evq[i].slot, i});
        for (uint32_t c = 0; c < P; c++) oq.push_back({p->ix_sigma + c, 0, e_sigma + c});
        oq.push_back({p->ix_hpoly, 0, e_h});
        oq.push_back({p->ix_random, 0, e_random});

        // ---- ProverGWC::create_proof: the point sets (circuit only), then per proof its v-weighted evaluation batches
        nsets = 0;
        lists.clear(); counts.clear(); set_slot.clear(); set_evs.clear();
        {
            std::vector<char> done(oq.size(), 0);
            for (size_t first = 0; first < oq.size(); first++) {
                if (done[first]) continue;
                const uint32_t slot = oq[first].slot;
                lists.resize((size_t)(nsets + 1) * 512, 0);
                set_evs.emplace_back();
                uint32_t cnt = 0;
                for (size_t j = first; j < oq.size(); j++) {
                    if (done[j] || oq[j].slot != slot) continue;
                    done[j] = 1;
                    ZG_REQUIRE(cnt < 512, ZG_ERR_UNSUPPORTED, "zg_prover_prove: more than 512 polynomials opened at one point");
                    lists[(size_t)nsets * 512 + cnt++] = oq[j].poly;
                    set_evs.back().push_back(oq[j].ev);
                }
                counts.push_back(cnt);
                set_slot.push_back(slot);
                nsets++;
            }
        }
        ZG_REQUIRE(nsets <= HC_MAX_SETS, ZG_ERR_UNSUPPORTED, "zg_prover_prove: %u opening points", nsets);
        ZG_TICK("opening sets listed");
        return ZG_OK;
    }
    int openings_queue() {
        // poly_batch of every point set in one launch: set s -> wpoly[2s]
        // (their own region of d_idx, behind the evaluation lists: nothing else writes there between proofs)
        uint32_t* d_lists = d_hlist + 64;
        ZG_TRY(h2d_list(p, d_lists, lists));
        ZG_TRY(poly_horner_combine_sets(ctx, polys, p->d_pc, nb, d_lists, 512, counts.data(), nsets, p->wpoly, (size_t)2 * n, wp_bs, n));
        // one batched kate_division: poly s at wpoly[2s], quotient at wpoly[2s+1]
        ZG_TRY(poly_kate_division(ctx, p->d_pc, nb, set_slot.data(), nsets, p->wpoly, (size_t)2 * n, wp_bs, p->wpoly + n, (size_t)2 * n,
                                  wp_bs, p->ktmp, n));
        // the witness polynomials sit at odd slots: stride 2n
        ctx->msm_dense_hint = true;  // (so are the opening quotients)
        const int st_w = commit(p, dense_g(p), nullptr, nsets, p->wpoly + n, (size_t)2 * n, nsets, wp_bs, (size_t)nb * nsets, 0, 0, W_GWC);
        ctx->msm_dense_hint = false;
        ZG_TRY(st_w);
        ZG_TICK("gwc: queued");
        return ZG_OK;
    }
    int openings_absorb() {
        ZG_TRY(wait_points(p, (size_t)nb * nsets, pts, W_GWC));
        ZG_TICK("gwc: points on the host");
        for (uint32_t b = 0; b < nb; b++)
            for (uint32_t s = 0; s < nsets; s++) tr[b].write_point(pts[(size_t)b * nsets + s]);
        return ZG_OK;
    }

    int finish() {
        int first_bad = ZG_OK;
        if (gate_failed()) {  // (every gate kernel has ended: the last commitments came from behind them)
            set_error(hold.yielded ? "zg_prover_prove: a phase queued ahead of its challenge had to allocate or synchronise; its gate was let go (ZG_LAT_GATE)"
                                   : "zg_prover_prove: a phase waited more than 4 s for its challenge and ran without it (ZG_LAT_GATE)");
            gate_gave_up = true;  // (nothing is handed out: prove_batch_impl makes the proof again in the plain order)
            p->in_flight = false;
            return ZG_ERR_HIP;
        }
        for (uint32_t b = 0; b < nb; b++) {
            if (status[b] == ZG_OK && tr[b].failed) {
                set_error("zg_prover_prove: a commitment of proof %u is the identity point; EvmTranscript cannot absorb it", b);
                status[b] = ZG_ERR_INVALID_ARG;
            }
            if (status[b] == ZG_OK && tr[b].stream.size() > proof_cap) {
                set_error("zg_prover_prove: proof buffer too small (%zu > %zu)", tr[b].stream.size(), proof_cap);
                status[b] = ZG_ERR_INVALID_ARG;
            }
            if (status[b] == ZG_OK) {
                memcpy(proofs[b], tr[b].stream.data(), tr[b].stream.size());
                proof_lens[b] = tr[b].stream.size();
            } else {
                proof_lens[b] = 0;
                if (first_bad == ZG_OK) first_bad = status[b];
            }
            if (statuses) statuses[b] = status[b];
        }
        ZG_TICK("proof bytes out");
        lap(5);
        p->phase_ms[6] = std::chrono::duration<double, std::milli>(clk::now() - t_start).count();
        p->in_flight = false;
        p->warm_sig = first_bad == ZG_OK ? form_sig() : 0;
        return first_bad;
    }
};

static int prove_batch_impl(zg_prover* p, size_t count, const zg_fr* const* advice_host, void* const* advice_dev,
                            const zg_fr* const* instance, size_t instance_len, const uint8_t* keys /* [count][32] */,
                            uint8_t* const* proofs, size_t proof_cap, size_t* proof_lens, int* statuses) {
    ZG_REQUIRE(p && proofs && proof_lens && keys, ZG_ERR_INVALID_ARG, "zg_prover_prove: null argument");
    const PkDev& pk = *p->pk;
    ZG_REQUIRE(count >= 1 && count <= p->cap, ZG_ERR_INVALID_ARG, "zg_prover_prove: %zu proofs for %u slots (zg_prover_set_batch)",
               count, p->cap);
    ZG_REQUIRE(pk.I == 0 || instance || instance_len == 0, ZG_ERR_INVALID_ARG, "zg_prover_prove: instance is null");
    ZG_REQUIRE(instance_len <= pk.usable, ZG_ERR_INVALID_ARG, "zg_prover_prove: instance too large (Error::InstanceTooLarge)");
    ZG_REQUIRE(p->world > 1 || p->shard_n == pk.n, ZG_ERR_INVALID_ARG,
               "zg_prover_prove: the base sets hold %u of %u points and no shard was declared (zg_prover_set_shard)", p->shard_n, pk.n);
    ZG_ENTER(p->ctx);
    {
        ProveBatch job(p, count, advice_host, advice_dev, instance, instance_len, keys, proofs, proof_cap, proof_lens, statuses);
        const int st = job.run();
        if (!job.gate_gave_up) return st;
    }
    // A gate ran into its time limit (the host thread was away for seconds, or a stream that shares a hardware queue with
    // this prover's stood in the way -- LAT_GATE_DEFAULT): the phases behind it ran on the previous challenge.  The inputs
    // are where they were (the blinding rows are a function of the key): the same proof again, every phase after its challenge.
    ProveBatch again(p, count, advice_host, advice_dev, instance, instance_len, keys, proofs, proof_cap, proof_lens, statuses);
    again.no_gate = true;
    return again.run();
}

int zg_prover_prove_batch(zg_prover* p, size_t count, const zg_fr* const* advice, const zg_fr* const* instance,
                          size_t instance_len, const uint8_t* rng_keys, uint8_t* const* proofs, size_t proof_cap,
                          size_t* proof_lens, int* statuses) {
    return prove_batch_impl(p, count, advice, nullptr, instance, instance_len, rng_keys, proofs, proof_cap, proof_lens, statuses);
}

int zg_prover_prove_batch_dev(zg_prover* p, size_t count, void* const* d_advice, const zg_fr* const* instance,
                              size_t instance_len, const uint8_t* rng_keys, uint8_t* const* proofs, size_t proof_cap,
                              size_t* proof_lens, int* statuses) {
    return prove_batch_impl(p, count, nullptr, d_advice, instance, instance_len, rng_keys, proofs, proof_cap, proof_lens, statuses);
}

int zg_prover_prove_dev(zg_prover* p, void* d_advice, const zg_fr* instance, size_t instance_len, const uint8_t rng_key[32],
                        uint8_t* proof, size_t proof_cap, size_t* proof_len) {
    ZG_REQUIRE(p && proof && proof_len && rng_key && (d_advice || p->pk->A == 0), ZG_ERR_INVALID_ARG, "zg_prover_prove: null argument");
    void* adv[1] = {d_advice};
    const zg_fr* inst[1] = {instance};
    uint8_t* out[1] = {proof};
    return prove_batch_impl(p, 1, nullptr, adv, instance ? inst : nullptr, instance_len, rng_key, out, proof_cap, proof_len, nullptr);
}

int zg_prover_prove(zg_prover* p, const zg_fr* advice, const zg_fr* instance, size_t instance_len, const uint8_t rng_key[32],
                    uint8_t* proof, size_t proof_cap, size_t* proof_len) {
    ZG_REQUIRE(p && proof && proof_len && rng_key && (advice || p->pk->A == 0), ZG_ERR_INVALID_ARG, "zg_prover_prove: null argument");
    const zg_fr* adv[1] = {advice};
    const zg_fr* inst[1] = {instance};
    uint8_t* out[1] = {proof};
    return prove_batch_impl(p, 1, adv, nullptr, instance ? inst : nullptr, instance_len, rng_key, out, proof_cap, proof_len, nullptr);
}

int zg_prover_set_overlap(zg_prover* p, int enable) {
    ZG_REQUIRE(p, ZG_ERR_INVALID_ARG, "zg_prover_set_overlap: null prover");
    ZG_ENTER(p->ctx);
    if (enable && !p->ctx->side) {
        ZG_TRY(zg_ctx_create(p->ctx->device, &p->ctx->side));
        p->ctx->side->profiling = p->ctx->profiling;
        p->ctx->side->prof_filter = p->ctx->prof_filter;
    }
    p->use_side = enable != 0;
    p->ctx->msm_pair = enable != 0;  // latency configuration: two lanes per addition in the MSM reduction
    p->lat_split = lone_split(p, enable != 0);
    return ZG_OK;
}

// The latency form's digit tables (msm.hip: every multiple of every window -- no buckets) of this prover's base sets:
// g, g_lagrange and the running sums of g_lagrange.  EXPLICIT since round 4: tens of GB and seconds of build
// time are the caller's decision, never a side effect of the first proof.
int zg_prover_enable_digit_tables(zg_prover* p, uint64_t max_bytes, uint64_t* bytes_built) {
    ZG_REQUIRE(p, ZG_ERR_INVALID_ARG, "zg_prover_enable_digit_tables: null prover");
    zg_ctx* ctx = p->ctx;
    ZG_ENTER(ctx);
    if (bytes_built) *bytes_built = 0;
    ZG_REQUIRE(p->g->n == p->gl->n, ZG_ERR_INVALID_ARG, "zg_prover_enable_digit_tables: the base sets differ in length");
    uint32_t c = p->g->full_c ? p->g->full_c : p->gl->full_c;
    if (!c) {
        size_t free_b = 0, total_b = 0;
        ZG_HIP(hipMemGetInfo(&free_b, &total_b));
        // the library's own cap: a third of the card (90 GB of the MI355X's 288), and never more than what is free now
        // minus a reserve of 8 GiB per table build
        double budget = max_bytes ? (double)max_bytes : (double)total_b / 3.0;
        if (budget > 90e9 && !max_bytes) budget = 90e9;
        if (budget + 3.0 * 8.0 * 1073741824.0 > (double)free_b) budget = (double)free_b - 3.0 * 8.0 * 1073741824.0;
        c = budget > 0 ? default_full_bits(p->g->n, budget) : 0;
    }
    if (!c) return ZG_OK;  // (n too large, the knob says none, or no room inside the budget: the bucket form stays)
    ZG_TRY(bases_enable_full(ctx, p->g, c, false));
    ZG_TRY(bases_enable_full(ctx, p->gl, c, true));
    if (bytes_built) {
        const uint64_t one = (uint64_t)((255 + c - 1) / c) * (1ull << (c - 1)) * p->g->n * sizeof(Affine);
        *bytes_built = (p->g->full_table.load() ? one : 0) + (p->gl->full_table.load() ? one : 0) + (p->gl->full_run_table.load() ? one : 0);
    }
    return ZG_OK;
}

int zg_prover_phase_ms(const zg_prover* p, double* out, size_t cap) {
    ZG_REQUIRE(p && out, ZG_ERR_INVALID_ARG, "zg_prover_phase_ms: null argument");
    for (size_t i = 0; i < cap && i < 8; i++) out[i] = p->phase_ms[i];
    return ZG_OK;
}

int zg_prover_gate_stats(const zg_prover* p, uint64_t* out, size_t cap) {
    ZG_REQUIRE(p && out, ZG_ERR_INVALID_ARG, "zg_prover_gate_stats: null argument");
    for (size_t i = 0; i < cap && i < 4; i++) out[i] = p->gate_stats[i];
    return ZG_OK;
}

int zg_prover_fetch_slot(zg_prover* p, size_t slot, uint32_t what, uint32_t index, zg_fr* out, size_t cap_elems) {
    ZG_REQUIRE(p && out, ZG_ERR_INVALID_ARG, "zg_prover_fetch: null argument");
    ZG_ENTER(p->ctx);
    ZG_REQUIRE(p->have_last, ZG_ERR_INVALID_ARG, "zg_prover_fetch: no proof has been produced yet");
    ZG_REQUIRE(slot < p->last_nb, ZG_ERR_INVALID_ARG, "zg_prover_fetch: slot %zu of a batch of %u", slot, p->last_nb);
    const PkDev& pk = *p->pk;
    const Fe* src = nullptr;
    size_t count = 0;
    const size_t n = pk.n;
    const Fe* zs = p->zs + slot * (size_t)(pk.sets + pk.NL + 1) * n;
    const Fe* perm = p->perm + slot * (size_t)(2 * pk.NL + 1) * n;
    const Fe* hp = p->pp + slot * (size_t)p->npp * n + (size_t)(p->ix_hpiece - p->nsh) * n;
    switch (what) {
        case 0: src = p->dbuf[0].h + slot * (size_t)pk.en; count = pk.en; break;
        case 1: ZG_REQUIRE(index < pk.sets, ZG_ERR_INVALID_ARG, "zg_prover_fetch: set %u", index);
                src = zs + (size_t)index * n; count = n; break;
        case 2: ZG_REQUIRE(index < pk.NL, ZG_ERR_INVALID_ARG, "zg_prover_fetch: lookup %u", index);
                src = zs + (size_t)(pk.sets + index) * n; count = n; break;
        case 3: ZG_REQUIRE(index < pk.NL, ZG_ERR_INVALID_ARG, "zg_prover_fetch: lookup %u", index);
                src = perm + (size_t)(2 * index) * n; count = n; break;
        case 4: ZG_REQUIRE(index < pk.NL, ZG_ERR_INVALID_ARG, "zg_prover_fetch: lookup %u", index);
                src = perm + (size_t)(2 * index + 1) * n; count = n; break;
        case 5: src = hp; count = (size_t)pk.qpd * n; break;
        default: ZG_REQUIRE(false, ZG_ERR_INVALID_ARG, "zg_prover_fetch: unknown item %u", what);
    }
    ZG_REQUIRE(cap_elems >= count, ZG_ERR_INVALID_ARG, "zg_prover_fetch: need %zu elements", count);
    if (what == 0 && p->last_split) {  // split domain: h on EvaluationDomain's coset, from its coefficients
        WsScope ws(p->ctx);
        Fe* tmp = ws.get<Fe>(count);
        if (!tmp) return ZG_ERR_OOM;
        ZG_TRY(coeff_to_coset_dev(p->ctx, hp, (size_t)pk.qpd * n, (uint32_t)(pk.qpd * n), tmp, count, 1, pk.ext_k, false, 1));
        ZG_HIP(hipStreamSynchronize(p->ctx->stream));
        ZG_HIP(hipMemcpy(out, tmp, count * 32, hipMemcpyDeviceToHost));
        return ZG_OK;
    }
    if (what == 0 && pk.hat) {  // h on the coset is kept as x * 2^261: hand back the library form
        WsScope ws(p->ctx);
        Fe* tmp = ws.get<Fe>(count);
        if (!tmp) return ZG_ERR_OOM;
        ZG_TRY(poly_scale(p->ctx, src, tmp, count, Fr::inv(Fr::from_u64(32))));
        ZG_HIP(hipStreamSynchronize(p->ctx->stream));
        ZG_HIP(hipMemcpy(out, tmp, count * 32, hipMemcpyDeviceToHost));
        return ZG_OK;
    }
    ZG_HIP(hipStreamSynchronize(p->ctx->stream));
    ZG_HIP(hipMemcpy(out, src, count * 32, hipMemcpyDeviceToHost));
    return ZG_OK;
}

int zg_prover_fetch(zg_prover* p, uint32_t what, uint32_t index, zg_fr* out, size_t cap_elems) {
    return zg_prover_fetch_slot(p, 0, what, index, out, cap_elems);
}

// ---- stand-alone building blocks ----
// Evaluator::evaluate_h (+ vanishing::Argument::construct's division by X^n - 1) for ONE circuit instance over this
// prover's resident proving key: from the coefficient forms of the witness-side polynomials and the four challenges
// to h on EvaluationDomain's extended coset (2^ext_k values, library form).  Host pointers: this is the
// arithmetic-level entry; inside create_proof the same kernel runs on slabs that never leave HBM.
int zg_prover_evaluate_h(zg_prover* p, const zg_fr* advice_polys, const zg_fr* instance_polys, const zg_fr* perm_z_polys,
                         const zg_fr* lookup_z_polys, const zg_fr* permuted_polys, const zg_fr* theta, const zg_fr* beta,
                         const zg_fr* gamma, const zg_fr* y, zg_fr* h_out) {
    ZG_REQUIRE(p && theta && beta && gamma && y && h_out, ZG_ERR_INVALID_ARG, "zg_prover_evaluate_h: null argument");
    const PkDev& pk = *p->pk;
    ZG_REQUIRE((advice_polys || !pk.A) && (instance_polys || !pk.I) && (perm_z_polys || !pk.sets) &&
                   (lookup_z_polys || !pk.NL) && (permuted_polys || !pk.NL),
               ZG_ERR_INVALID_ARG, "zg_prover_evaluate_h: a polynomial family is missing");
    ZG_REQUIRE(p->cap >= 1, ZG_ERR_INVALID_ARG, "zg_prover_evaluate_h: the prover has no slot");
    zg_ctx* ctx = p->ctx;
    ZG_ENTER(ctx);
    hipStream_t st = ctx->stream;
    const uint32_t n = pk.n;
    p->have_last = false;
    // slot 0's coefficient slab, in its order: advice, instance, permutation z, lookup z, a'/s'
    struct Fam { const zg_fr* src; uint32_t ix, count; };
    const Fam fams[] = {{advice_polys, p->ix_adv, pk.A}, {instance_polys, p->ix_inst, pk.I}, {perm_z_polys, p->ix_pz, pk.sets},
                        {lookup_z_polys, p->ix_lz, pk.NL}, {permuted_polys, p->ix_perm, 2 * pk.NL}};
    for (const Fam& f : fams)
        if (f.count)
            ZG_HIP(hipMemcpyAsync(p->pp + (size_t)(f.ix - p->nsh) * n, f.src, (size_t)f.count * n * 32, hipMemcpyHostToDevice, st));
    ProofConst& c = p->hpc[0];
    memset(&c, 0, sizeof(c));
    c.theta = to_fe(theta); c.beta = to_fe(beta); c.gamma = to_fe(gamma);
    evalh_consts(c, to_fe(y), pk.hat, evalh_terms(pk));
    p->stage_off = p->pin_stage;
    ZG_TRY(upload_consts(p, 1));
    const PkDev::Dom& d = pk.dom[0];
    ZG_TRY(coeff_to_coset_dev(ctx, p->pp + (size_t)(p->ix_adv - p->nsh) * n, n, n, p->dbuf[0].cos, d.en, p->ncos, d.ek, pk.hat, d.zpow));
    const EvalHArgs a = evalh_args(p, 0);
    ZG_TRY(poly_evaluate_h(ctx, a, d.en, 1, pk.A + pk.I + pk.F, (double)d.en));
    WsScope ws(ctx);
    Fe* tmp = ws.get<Fe>(d.en);
    if (!tmp) return ZG_ERR_OOM;
    // (the nine-limb kernel leaves h as x * 2^261: hand back the library form)
    ZG_TRY(poly_scale(ctx, p->dbuf[0].h, tmp, d.en, pk.hat ? Fr::inv(Fr::from_u64(32)) : Fr::one()));
    ZG_HIP(hipMemcpyAsync(h_out, tmp, (size_t)d.en * 32, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    return ZG_OK;
}

int zg_grand_product_dev(zg_ctx* ctx, const void* d_num, const void* d_den, const zg_fr* z0, size_t n, void* d_z) {
    ZG_REQUIRE(ctx && d_num && d_den && d_z && z0, ZG_ERR_INVALID_ARG, "zg_grand_product_dev: null argument");
    ZG_REQUIRE(n < (1u << 28), ZG_ERR_UNSUPPORTED, "zg_grand_product_dev: n too large");
    ZG_ENTER(ctx);
    WsScope ws(ctx);
    Fe* tmp = ws.get<Fe>(poly_grand_product_tmp_elems((uint32_t)n, 1) + 1);
    if (ws.failed) return ZG_ERR_OOM;
    Fe* z0d = tmp + poly_grand_product_tmp_elems((uint32_t)n, 1);
    ZG_HIP(hipMemcpyAsync(z0d, z0, 32, hipMemcpyHostToDevice, ctx->stream));
    ZG_TRY(poly_grand_product(ctx, (const Fe*)d_num, (const Fe*)d_den, z0d, (Fe*)d_z, tmp, (uint32_t)n, 1, 0, 0));
    ZG_HIP(hipStreamSynchronize(ctx->stream));
    return ZG_OK;
}

// Host-pointer form of the same (lookup::prover::commit_product / permutation::prover::commit's running product):
// z[0] = z0, z[i+1] = z[i] * num[i] / den[i] with BatchInvert's rule for a zero denominator (ratio 0); z has n entries.
int zg_grand_product(zg_ctx* ctx, const zg_fr* num, const zg_fr* den, const zg_fr* z0, size_t n, zg_fr* z) {
    ZG_REQUIRE(ctx && num && den && z0 && z, ZG_ERR_INVALID_ARG, "zg_grand_product: null argument");
    ZG_REQUIRE(n >= 1 && n < (1u << 28), ZG_ERR_UNSUPPORTED, "zg_grand_product: n out of range");
    ZG_ENTER(ctx);
    WsScope ws(ctx);
    Fe* d = ws.get<Fe>(3 * n);
    if (ws.failed) return ZG_ERR_OOM;
    ZG_HIP(hipMemcpyAsync(d, num, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ZG_HIP(hipMemcpyAsync(d + n, den, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ZG_TRY(zg_grand_product_dev(ctx, d, d + n, z0, n, d + 2 * n));
    ZG_HIP(hipMemcpy(z, d + 2 * n, n * 32, hipMemcpyDeviceToHost));
    return ZG_OK;
}

int zg_eval_polys_dev(zg_ctx* ctx, const void* d_polys, size_t stride_elems, size_t n, const uint32_t* poly_index,
                      const zg_fr* points, size_t count, zg_fr* out) {
    ZG_REQUIRE(ctx && d_polys && poly_index && points && out, ZG_ERR_INVALID_ARG, "zg_eval_polys_dev: null argument");
    if (!count) return ZG_OK;
    ZG_REQUIRE(stride_elems == n || count == 0, ZG_ERR_UNSUPPORTED, "zg_eval_polys_dev: stride %zu != n %zu", stride_elems, n);
    ZG_ENTER(ctx);
    WsScope ws(ctx);
    // the pairs are served PC_MAX_POINTS at a time, every pair with a powers row of its own (callers with shared
    // points should use the prover)
    Fe* pw = ws.get<Fe>((size_t)PC_MAX_POINTS * n);
    uint32_t* di = ws.get<uint32_t>(2 * PC_MAX_POINTS);
    Fe* de = ws.get<Fe>(PC_MAX_POINTS);
    ProofConst* dpc = ws.get<ProofConst>(1);
    if (ws.failed) return ZG_ERR_OOM;
    PolySet ps;
    ps.sh = (const Fe*)d_polys; ps.pp = nullptr; ps.nsh = 0xffffffffu; ps.n = n; ps.pp_bs = 0;
    for (size_t c0 = 0; c0 < count; c0 += PC_MAX_POINTS) {
        const uint32_t m = (uint32_t)std::min<size_t>(PC_MAX_POINTS, count - c0);
        ProofConst hc;
        memset(&hc, 0, sizeof(hc));
        uint32_t idx[2 * PC_MAX_POINTS];
        for (uint32_t i = 0; i < m; i++) {
            hc.points[i] = to_fe(&points[c0 + i]);
            idx[i] = poly_index[c0 + i];
            idx[m + i] = i;
        }
        ZG_HIP(hipMemcpyAsync(dpc, &hc, sizeof(hc), hipMemcpyHostToDevice, ctx->stream));
        ZG_HIP(hipMemcpyAsync(di, idx, 2 * m * 4, hipMemcpyHostToDevice, ctx->stream));
        ZG_HIP(hipStreamSynchronize(ctx->stream));  // (hc and idx are stack memory)
        ZG_TRY(poly_powers(ctx, dpc, 1, m, (uint32_t)n, pw, 0));
        ZG_TRY(poly_dot(ctx, ps, 1, (uint32_t)n, di, di + m, pw, 0, m, de, 0));
        ZG_HIP(hipMemcpyAsync(out + c0, de, m * 32, hipMemcpyDeviceToHost, ctx->stream));
        ZG_HIP(hipStreamSynchronize(ctx->stream));
    }
    return ZG_OK;
}

int zg_kate_division_dev(zg_ctx* ctx, const void* d_a, size_t n, const zg_fr* z, void* d_q) {
    ZG_REQUIRE(ctx && d_a && z && d_q && n >= 1, ZG_ERR_INVALID_ARG, "zg_kate_division_dev: bad argument");
    ZG_ENTER(ctx);
    ZG_REQUIRE(n < (1u << 28), ZG_ERR_UNSUPPORTED, "zg_kate_division_dev: n too large");
    WsScope ws(ctx);
    Fe* tmp = ws.get<Fe>(poly_kate_tmp_elems((uint32_t)n, 1));
    ProofConst* dpc = ws.get<ProofConst>(1);
    if (ws.failed) return ZG_ERR_OOM;
    ProofConst hc;
    memset(&hc, 0, sizeof(hc));
    hc.points[0] = to_fe(z);
    ZG_HIP(hipMemcpyAsync(dpc, &hc, sizeof(hc), hipMemcpyHostToDevice, ctx->stream));
    ZG_HIP(hipStreamSynchronize(ctx->stream));
    const uint32_t slot0 = 0;
    ZG_TRY(poly_kate_division(ctx, dpc, 1, &slot0, 1, (const Fe*)d_a, n, 0, (Fe*)d_q, n, 0, tmp, (uint32_t)n));
    ZG_HIP(hipStreamSynchronize(ctx->stream));
    return ZG_OK;
}

}  // extern "C"
