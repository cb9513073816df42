// Context, workspace pool and error plumbing of libzg_halo2.
#include <cstdlib>

#include "common.h"

namespace zg {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

// ---- tuning knobs
static const char* const kKnobNames[K_COUNT] = {"ZG_MSM_C", "ZG_MSM_K", "ZG_MSM_K_LAT", "ZG_MSM_RB", "ZG_MSM_LANES", "ZG_MSM_STRIP",
                                                "ZG_MSM_NAF", "ZG_MSM_NAF_GL", "ZG_MSM_RUNS", "ZG_EVALH_GROUPED", "ZG_EVALH9",
                                                "ZG_SPLIT_DOMAIN", "ZG_LAT_SPLIT_K", "ZG_LAT_FULL_C", "ZG_LAT_FULL_K", "ZG_LAZY_DOT",
                                                "ZG_MSM_AFFINE", "ZG_MSM_HEAVY", "ZG_LAT_PULL", "ZG_LAT_GATE", "ZG_WITNESS_LDS", "ZG_MSM_TOPSPLIT", "ZG_NTT9"};
// (read beside the knobs, not knobs: settings of the HIP RUNTIME under which every launch completes before the next one is
//  submitted -- a stream that waits for the host, ZG_LAT_GATE, must not be started then)
static const char* const kRuntimeSerialising[2] = {"AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING"};
static std::atomic<int> g_knobs[K_COUNT];
static std::atomic<bool> g_runtime_serialises{false};
static std::once_flag g_knobs_once;
static void knobs_init() {
    for (int i = 0; i < K_COUNT + 2; i++) {
        const char* e = getenv(i < K_COUNT ? kKnobNames[i] : kRuntimeSerialising[i - K_COUNT]);  // the ONLY getenv of the library
        if (i < K_COUNT) g_knobs[i].store(e && *e ? atoi(e) : -1, std::memory_order_relaxed);
        else if (e && atoi(e)) g_runtime_serialises.store(true, std::memory_order_relaxed);
    }
}
int knob(Knob k) {
    std::call_once(g_knobs_once, knobs_init);
    return g_knobs[k].load(std::memory_order_relaxed);
}
bool runtime_serialises_launches() {
    std::call_once(g_knobs_once, knobs_init);
    return g_runtime_serialises.load(std::memory_order_relaxed);
}
static int knob_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < K_COUNT; i++)
        if (strcmp(name, kKnobNames[i]) == 0) return i;
    return -1;
}

void* ws_alloc(zg_ctx* ctx, size_t bytes) {
    bytes = (bytes + 4095) & ~size_t(4095);
    WsBlock* best = nullptr;
    for (auto& b : ctx->pool)
        if (!b.used && b.cap >= bytes && (!best || b.cap < best->cap)) best = &b;
    // do not hand a huge block to a tiny request when that would starve a later large one
    if (best && best->cap <= 4 * bytes + (1u << 20)) {
        best->used = true;
        return best->p;
    }
    void* p = nullptr;
    gate_yield(ctx);  // (an allocation may wait for the device)
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    ctx->pool.push_back(WsBlock{p, bytes, true});
    return p;
}

void ws_release(zg_ctx* ctx, void* p) {
    for (auto& b : ctx->pool)
        if (b.p == p) {
            b.used = false;
            return;
        }
}

int pinned_reserve(zg_ctx* ctx, size_t bytes) {
    if (ctx->pinned_cap >= bytes) return ZG_OK;
    gate_yield(ctx);
    if (ctx->pinned) {
        ZG_HIP(hipStreamSynchronize(ctx->stream));
        ZG_HIP(hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_cap = 0;
    }
    size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes;
    // (mapped + coherent: kernels may write their few result words straight into it -- no copy command behind them)
    ZG_HIP(hipHostMalloc(&ctx->pinned, cap, hipHostMallocMapped | hipHostMallocCoherent));
    ctx->pinned_cap = cap;
    return ZG_OK;
}

void gate_yield(zg_ctx* ctx) {
    zg_ctx::GateHold* h = ctx->gate_hold;
    if (!h || !h->word) return;
    h->yielded = true;
    h->yields++;
    __atomic_store_n(h->word, h->seq, __ATOMIC_SEQ_CST);
    h->word = nullptr;
}

static std::atomic<uint32_t> g_tuning_generation{0};
uint32_t tuning_generation() { return g_tuning_generation.load(std::memory_order_relaxed); }
void bump_tuning_generation() { g_tuning_generation.fetch_add(1, std::memory_order_relaxed); }

static hipEvent_t pool_event(zg_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

zg_ctx::ProfRec* prof_slot(zg_ctx* ctx, const char* name, double algo_bytes, double unit_bytes) {
    if (!ctx->prof_filter.empty() && ctx->prof_filter != name) return nullptr;
    zg_ctx::ProfRec r;
    r.name = name;
    r.bytes = algo_bytes;
    r.unit_bytes = unit_bytes;
    r.e0 = pool_event(ctx);
    r.e1 = pool_event(ctx);
    ctx->prof.push_back(r);
    return &ctx->prof.back();  // (valid until the next slot is taken: the launch macro uses it at once)
}

DeviceState& device_state(int device) {
    static std::mutex table_mu;
    static std::map<int, DeviceState*> table;  // entries are never removed: references stay valid
    std::lock_guard<std::mutex> lock(table_mu);
    DeviceState*& d = table[device];
    if (!d) d = new DeviceState();
    return *d;
}

Fe host_domain_omega(uint32_t log_n) {
    Fe w = fr_root_of_unity();
    for (uint32_t i = log_n; i < FR_S; i++) w = Fr::sqr(w);
    return w;
}

}  // namespace zg

using namespace zg;

extern "C" {

const char* zg_last_error(void) { return g_last_error.c_str(); }

const char* zg_version(void) { return "zg_halo2 0.3 (gfx950)"; }

int zg_tuning_set(const char* name, int value) {
    const int i = knob_index(name);
    ZG_REQUIRE(i >= 0, ZG_ERR_INVALID_ARG, "zg_tuning_set: unknown knob %s", name ? name : "(null)");
    (void)knob((Knob)i);  // (the environment is read first, so that it cannot overwrite this value later)
    g_knobs[i].store(value < 0 ? -1 : value, std::memory_order_relaxed);
    bump_tuning_generation();  // (a knob may change buffer sizes or forms: a warm prover's next proof is a first proof again)
    return ZG_OK;
}
int zg_tuning_get(const char* name, int* value) {
    const int i = knob_index(name);
    ZG_REQUIRE(i >= 0 && value, ZG_ERR_INVALID_ARG, "zg_tuning_get: unknown knob %s", name ? name : "(null)");
    *value = knob((Knob)i);
    return ZG_OK;
}
size_t zg_tuning_names(const char** out, size_t cap) {
    for (size_t i = 0; i < (size_t)K_COUNT && i < cap && out; i++) out[i] = kKnobNames[i];
    return (size_t)K_COUNT;
}

int zg_ctx_create(int device_id, zg_ctx** out) {
    ZG_REQUIRE(out != nullptr, ZG_ERR_INVALID_ARG, "zg_ctx_create: out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("zg_ctx_create: no HIP device visible (%s); this library has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return ZG_ERR_NO_DEVICE;
    }
    ZG_REQUIRE(device_id >= 0 && device_id < count, ZG_ERR_INVALID_ARG,
               "zg_ctx_create: device %d out of range (%d devices)", device_id, count);
    ZG_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    ZG_HIP(hipGetDeviceProperties(&prop, device_id));
    zg_ctx* ctx = new zg_ctx();
    ctx->device = device_id;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete ctx;
        return ZG_ERR_HIP;
    }
    {
        DeviceState& ds = device_state(device_id);
        std::lock_guard<std::mutex> lock(ds.mu);
        ds.refs++;
    }
    *out = ctx;
    return ZG_OK;
}

void zg_ctx_destroy(zg_ctx* ctx) {
    if (!ctx) return;
    if (ctx->side) {
        zg_ctx_destroy(ctx->side);
        ctx->side = nullptr;
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    {
        DeviceState& ds = device_state(ctx->device);
        std::lock_guard<std::mutex> lock(ds.mu);
        if (--ds.refs == 0) {  // last context of the device: the shared twiddle tables go with it
            for (auto& kv : ds.twiddles) (void)hipFree(kv.second);
            ds.twiddles.clear();
        }
    }
    for (auto& r : ctx->prof) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (auto& b : ctx->pool) (void)hipFree(b.p);
    if (ctx->msm_tickets) (void)hipFree(ctx->msm_tickets);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int zg_ctx_sync(zg_ctx* ctx) {
    ZG_REQUIRE(ctx != nullptr, ZG_ERR_INVALID_ARG, "zg_ctx_sync: ctx is null");
    ZG_ENTER(ctx);
    ZG_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->side) ZG_HIP(hipStreamSynchronize(ctx->side->stream));
    return ZG_OK;
}

int zg_ctx_trim(zg_ctx* ctx, uint64_t* freed_bytes) {
    ZG_REQUIRE(ctx != nullptr, ZG_ERR_INVALID_ARG, "zg_ctx_trim: ctx is null");
    ZG_ENTER(ctx);
    uint64_t freed = 0;
    for (zg_ctx* c : {ctx, ctx->side}) {
        if (!c) continue;
        ZG_HIP(hipStreamSynchronize(c->stream));
        std::vector<WsBlock> keep;
        for (auto& b : c->pool) {
            if (b.used) {
                keep.push_back(b);
            } else {
                (void)hipFree(b.p);
                freed += b.cap;
            }
        }
        c->pool.swap(keep);
    }
    if (freed_bytes) *freed_bytes = freed;
    return ZG_OK;
}

void* zg_ctx_stream(zg_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int zg_ctx_profile_enable(zg_ctx* ctx, int on) {
    ZG_REQUIRE(ctx != nullptr, ZG_ERR_INVALID_ARG, "zg_ctx_profile_enable: ctx is null");
    ZG_ENTER(ctx);
    ctx->profiling = on != 0;
    if (ctx->side) ctx->side->profiling = ctx->profiling;
    return ZG_OK;
}

int zg_ctx_profile_filter(zg_ctx* ctx, const char* kernel_name) {
    ZG_REQUIRE(ctx != nullptr, ZG_ERR_INVALID_ARG, "zg_ctx_profile_filter: ctx is null");
    ZG_ENTER(ctx);
    ctx->prof_filter = kernel_name ? kernel_name : "";
    if (ctx->side) ctx->side->prof_filter = ctx->prof_filter;
    return ZG_OK;
}

// Synchronises, folds the recorded launches into per-kernel totals and clears the log.
int zg_ctx_profile_collect(zg_ctx* ctx, zg_kernel_stat* out, size_t cap, size_t* count) {
    ZG_REQUIRE(ctx && count, ZG_ERR_INVALID_ARG, "zg_ctx_profile_collect: null argument");
    ZG_ENTER(ctx);
    ZG_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<zg_kernel_stat> acc;
    if (ctx->side) {  // fold the side stream's launches in
        ZG_HIP(hipStreamSynchronize(ctx->side->stream));
        for (auto& r : ctx->side->prof) ctx->prof.push_back(r);
        ctx->side->prof.clear();
    }
    for (auto& r : ctx->prof) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        ctx->event_pool.push_back(r.e0);
        ctx->event_pool.push_back(r.e1);
        zg_kernel_stat* s = nullptr;
        for (auto& a : acc)
            if (strcmp(a.name, r.name) == 0) s = &a;
        if (!s) {
            zg_kernel_stat n;
            memset(&n, 0, sizeof(n));
            strncpy(n.name, r.name, sizeof(n.name) - 1);
            acc.push_back(n);
            s = &acc.back();
        }
        s->launches += 1;
        s->total_ms += ms;
        s->algo_bytes += r.bytes;
        s->unit_bytes += r.unit_bytes;
    }
    ctx->prof.clear();
    *count = acc.size();
    if (out)
        for (size_t i = 0; i < acc.size() && i < cap; i++) out[i] = acc[i];
    return ZG_OK;
}

int zg_domain_omega(uint32_t log_n, zg_fr* omega, zg_fr* omega_inv) {
    ZG_REQUIRE(log_n <= FR_S, ZG_ERR_INVALID_ARG, "zg_domain_omega: log_n %u > 28", log_n);
    Fe w = host_domain_omega(log_n);
    if (omega) memcpy(omega, &w, 32);
    if (omega_inv) {
        Fe wi = Fr::inv(w);
        memcpy(omega_inv, &wi, 32);
    }
    return ZG_OK;
}

}  // extern "C"
