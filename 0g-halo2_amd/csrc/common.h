// Shared internals of libzg_halo2: context object, workspace arena, error plumbing.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <array>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/zg_halo2.h"
#include "curve.h"

namespace zg {

void set_error(const char* fmt, ...);

#define ZG_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t _e = (call);                                                              \
        if (_e != hipSuccess) {                                                              \
            ::zg::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                       \
            return ZG_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

#define ZG_TRY(expr)              \
    do {                          \
        int _s = (expr);          \
        if (_s != ZG_OK) return _s; \
    } while (0)

// First statement of an entry point that works on a context: takes the context lock for the rest of the call
// and makes its device current on the calling thread.
#define ZG_ENTER(ctxp)                                            \
    std::lock_guard<std::recursive_mutex> _zg_lock((ctxp)->mu); \
    ZG_HIP(hipSetDevice((ctxp)->device))

#define ZG_REQUIRE(cond, status, ...)   \
    do {                                \
        if (!(cond)) {                  \
            ::zg::set_error(__VA_ARGS__); \
            return (status);            \
        }                               \
    } while (0)

static_assert(sizeof(Fe) == 32 && sizeof(zg_fr) == 32, "field element layout");
static_assert(sizeof(Affine) == 64 && sizeof(zg_g1_affine) == 64, "affine layout");
static_assert(sizeof(Jac) == 96 && sizeof(zg_g1) == 96, "jacobian layout");
static_assert(sizeof(XYZZ) == 128, "xyzz layout");

// Cached HBM blocks.  Every user runs on the context stream, so a block released by one call can
// be handed to the next without further synchronisation; hipMalloc happens only the first time a
// size is seen (288 GB of HBM: blocks are kept, never trimmed).
struct WsBlock {
    void* p = nullptr;
    size_t cap = 0;
    bool used = false;
};

struct TwiddleKey {
    uint32_t log_n;
    std::array<uint64_t, 4> omega;
    bool operator<(const TwiddleKey& o) const {
        if (log_n != o.log_n) return log_n < o.log_n;
        return omega < o.omega;
    }
};

}  // namespace zg

namespace zg {
// What the contexts of one device share: NTT twiddle tables (omega^i, i < 2^log_n, per (log_n, omega); they live in
// HBM until the last context of the device is destroyed) and the once-per-device kernel attributes.
struct DeviceState {
    std::mutex mu;
    std::map<TwiddleKey, Fe*> twiddles;
    int refs = 0;
    bool msm_attrs = false, ntt_attrs = false;  // hipFuncSetAttribute is per device; set under `mu`
};
DeviceState& device_state(int device);
}  // namespace zg

struct zg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    std::vector<zg::WsBlock> pool;
    // pinned host staging for small D2H results
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    // Every extern "C" entry that takes this context (or a prover / base set created on it) holds `mu` for the
    // whole call: calls on one context serialise, calls on different contexts run concurrently.  Recursive
    // because entry points are built from one another (zg_msm -> zg_msm_batch -> zg_msm_finish).
    std::recursive_mutex mu;
    // optional per-kernel HIP-event timing (zg_ctx_profile_*): events bracket every launch
    bool profiling = false;
    std::string prof_filter;  // when non-empty only launches of this kernel are bracketed
    struct ProfRec {
        const char* name;
        hipEvent_t e0, e1;
        double bytes;       // what THIS kernel's algorithm streams: every distinct input once, every output once
        double unit_bytes;  // SURVEY.md 8d's figure of the unit of work (one MSM, one transform, ...), charged ONCE per unit:
                            // on the kernel that carries the unit; 0 on the other kernels of its launch sequence
    };
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> event_pool;  // recycled by zg_ctx_profile_collect
    zg_ctx* side = nullptr;  // optional second stream + workspace pool (created on demand, same device)
    // MSM bucket reduction with two lanes per EC addition (latency) or one (throughput); see msm.hip
    bool msm_pair = true;
    // SURVEY-unit bytes of the NEXT transform plan run on this context (-1: the plan's own in + out; ntt.hip consumes it)
    double unit_next = -1.0;
    bool msm_dense_hint = false;  // set by a caller around an MSM whose vectors are all random (latency form: one lane per task)
    uint32_t* msm_tickets = nullptr;  // last-workgroup-done counters of the MSM reduction (msm.hip), zero between launches
    // While a prover's gate kernel (prover.hip, ZG_LAT_GATE) spins on this context's stream -- or on the main stream this side
    // context's stream follows -- the host must not WAIT for that stream before it has opened the gate.  Every path of the
    // library that can block behind a stream from inside a proof (a workspace block that has to be allocated, a pinned arena
    // that has to grow, a twiddle table made on first use, the MSM's ticket counters) calls gate_yield() first: it opens the
    // gate at once and marks the proof as made on stale scalars (the prover makes it again in the plain order) -- a missed
    // case costs one proof, never the gate's time limit.
    struct GateHold {
        uint32_t* word = nullptr;  // the gate word in mapped host memory; nullptr = no gate armed
        uint32_t seq = 0;
        bool yielded = false;
        uint32_t yields = 0;
    };
    GateHold* gate_hold = nullptr;
};

struct zg_bases {
    zg_ctx* ctx = nullptr;  // the context that registered the set (its stream built the tables)
    int device = 0;         // any context of this device may multiply against the tables (read-only)
    size_t n = 0;
    uint32_t c = 0;        // window bits
    uint32_t windows = 0;  // ceil(255 / c)
    zg::Affine* table = nullptr;  // [windows][n]: 2^(c*w) * P_i, affine
    // the same table for the running sums Q_i = P_0 + ... + P_i (built on demand by bases_enable_runs): a scalar
    // vector with long constant runs is multiplied as sum_i (s_i - s_{i+1}) Q_i, whose coefficients vanish inside runs
    zg::Affine* run_table = nullptr;
    // the same points with one table row per BIT position (built on demand by bases_enable_naf), for vectors of
    // full-size scalars recoded into odd digits at free positions
    // (written once, under `mu`, before any prover that uses the set exists; readers on other contexts load it with
    //  acquire order: bases_dense())
    std::atomic<zg_bases*> dense{nullptr};
    // > 0: this table holds 2^j * P_i for EVERY bit position j (c = 1, 255 rows) and is multiplied with odd signed
    // digits of naf_w bits at free positions (msm.hip msm_digits_naf_kernel); only a `dense` table is built that way
    uint32_t naf_w = 0;
    // Digit tables for the LATENCY form (bases_enable_full): EVERY multiple d * 2^(full_c w) * P_i, d = 1 .. 2^(full_c - 1),
    // of every window -- [full_windows][2^(full_c-1)][n] affine points (26 GB at n = 2^14, c = 11: what 288 GB of HBM
    // are for).  A lone MSM then needs no buckets at all: every signed digit names its summand, and the sum of n * W
    // gathered points is a flat accumulation + a tree (msm_accumulate_full / msm_tree kernels) -- no digit sort, no
    // bucket reduction.  Published like `dense` (complete before the pointer is stored).
    std::atomic<zg::Affine*> full_table{nullptr};
    std::atomic<zg::Affine*> full_run_table{nullptr};  // the same for the running sums (run-form commitments)
    uint32_t full_c = 0, full_windows = 0;
    std::mutex mu;
};

namespace zg {

inline zg_bases* bases_dense(const zg_bases* b) { return b ? b->dense.load(std::memory_order_acquire) : nullptr; }

void* ws_alloc(zg_ctx* ctx, size_t bytes);  // nullptr on failure (error set)
void ws_release(zg_ctx* ctx, void* p);

// RAII: all blocks taken through a scope go back to the pool when the call returns.
struct WsScope {
    zg_ctx* ctx;
    std::vector<void*> held;
    bool failed = false;
    explicit WsScope(zg_ctx* c) : ctx(c) {}
    ~WsScope() {
        for (void* p : held) ws_release(ctx, p);
    }
    template <class T>
    T* get(size_t count) {
        void* p = ws_alloc(ctx, count * sizeof(T) + 256);
        if (!p) {
            failed = true;
            return nullptr;
        }
        held.push_back(p);
        return reinterpret_cast<T*>(p);
    }
};
int pinned_reserve(zg_ctx* ctx, size_t bytes);
void gate_yield(zg_ctx* ctx);      // see zg_ctx::GateHold
uint32_t tuning_generation();      // bumped by every zg_tuning_set (a warm prover's next proof is a first proof again)

// Launch wrapper: when profiling is on, the dispatch carries a start and a stop event of its own
// (hipExtLaunchKernelGGL: the timestamps are the kernel's begin and end, as rocprofv3 reports them -- events
// recorded around the launch would add the queueing time of a busy stream) and `algo_bytes` (the
// ALGORITHMIC bytes this launch is charged with) is recorded beside them.
zg_ctx::ProfRec* prof_slot(zg_ctx* ctx, const char* name, double algo_bytes, double unit_bytes);  // nullptr: filtered out
// ZG_LAUNCH_U: a kernel with its own streamed bytes AND the SURVEY-unit bytes it carries (0: a stage kernel of a
// multi-kernel unit, or a kernel SURVEY.md 8d names no unit for); ZG_LAUNCH: a stage / unit-less kernel (unit bytes 0).
#define ZG_LAUNCH_U(ctx, name, bytes, unit, kernel, grid, block, lds, ...)                               \
    do {                                                                                                 \
        zg_ctx::ProfRec* _zg_r = (ctx)->profiling ? ::zg::prof_slot((ctx), (name), (double)(bytes), (double)(unit)) : nullptr; \
        if (_zg_r)                                                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, (ctx)->stream, _zg_r->e0, _zg_r->e1, 0, __VA_ARGS__); \
        else                                                                                             \
            hipLaunchKernelGGL(kernel, grid, block, lds, (ctx)->stream, __VA_ARGS__);                    \
    } while (0)
#define ZG_LAUNCH(ctx, name, bytes, kernel, grid, block, lds, ...) \
    ZG_LAUNCH_U(ctx, name, bytes, 0.0, kernel, grid, block, lds, __VA_ARGS__)

// twiddle table for (log_n, omega), created on first use
int get_twiddles(zg_ctx* ctx, uint32_t log_n, const Fe& omega, Fe** out);

// host-side constants
Fe host_domain_omega(uint32_t log_n);

// Tuning knobs (include/zg_halo2.h, "tuning").  Each starts from its ZG_* environment variable, read once, and can be
// changed at run time with zg_tuning_set -- tests walk all of them in ONE process.  knob() < 0 = the library's default.
// Knobs that shape resident data (window size, tables, the forms of a proving key) are read when that object is built;
// launch shapes (task sizes, reduction blocks) at every launch.
enum Knob : int {
    K_MSM_C,          // window bits of a base set registered with window_bits = 0
    K_MSM_K,          // points per accumulate task, throughput form (4..120)
    K_MSM_K_LAT,      // ... latency form
    K_MSM_RB,         // buckets per reduction block, latency form (64 / 128 / 256)
    K_MSM_LANES,      // lanes per EC addition in the latency reduction (2 / 4)
    K_MSM_STRIP,      // buckets per lane in the throughput reduction (2 / 4 / 8 / 16)
    K_MSM_NAF,        // digit width of the free-position form for the all-random commitments; 0 = window tables only
    K_MSM_NAF_GL,     // ... for the run-form commitments against g_lagrange; 0 = windows
    K_MSM_RUNS,       // 0 = no run form (summation by parts) for the sorted columns and the products
    K_EVALH_GROUPED,  // 0 = evaluate_h folds in y term by term
    K_EVALH9,         // 0 = evaluate_h on 8 x 32-bit limbs (implies the single extended coset)
    K_SPLIT_DOMAIN,   // 0 = EvaluationDomain's single extended coset in the throughput form too
    K_LAT_SPLIT_K,    // smallest k at which a LONE proof (latency form) takes the quotient from the split domain as well
    K_LAT_FULL_C,     // window bits of the latency form's digit tables (every multiple of every window); 0 = none
    K_LAT_FULL_K,     // summands per lane pair in the digit-table accumulation (4..120)
    K_LAZY_DOT,       // 0 = eval_polynomial and the multiopen combinations reduce after every term (dot_kernel, Horner)
    K_MSM_AFFINE,     // rounds of batched-affine pairwise additions before the bucket chains, throughput form (0..4)
    K_MSM_HEAVY,      // task partials above which a bucket is merged by msm_heavy instead of its strip's lane (1..64)
    K_LAT_PULL,       // 1 = a lone proof's per-phase scalars are pulled from pinned host memory by a one-wave kernel (no copy command)
    K_LAT_GATE,       // 1 = a lone proof queues each phase before the previous one's challenge exists, behind a gate word the host opens
    K_WITNESS_LDS,    // 0 = the witness program keeps every operand in HBM (witness.hip; read when a plan is made)
    K_MSM_TOPSPLIT,   // 0 = the free-position recoding leaves its last digit whatever bits remain (msm_digits_naf_kernel)
    K_NTT9,           // the transforms' butterflies: 0 = on 8 x 32-bit limbs (ntt_pass_kernel), 1 = on nine 29-bit ones; default: nine in the latency form
    K_COUNT
};
int knob(Knob k);
bool runtime_serialises_launches();  // AMD_SERIALIZE_KERNEL / HIP_LAUNCH_BLOCKING are set: no launch may wait for the host

// prover.hip: what a witness program has to match before it may write into a prover's advice slots (witness.hip)
struct ProverShape {
    zg_ctx* ctx;
    int device;
    uint32_t k, n_advice, n_instance, usable_rows;
    bool in_flight;  // a batch left through an error return: work may still be queued on the prover's streams
};
ProverShape prover_shape(const zg_prover* p);
int prover_drain(zg_prover* p);  // waits for everything queued on the prover's streams

}  // namespace zg
