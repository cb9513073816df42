// Witness of a batch of images on the device: the advice columns create_proof starts from.
//
// Upstream's create_proof obtains them by running the circuit's synthesize under a WitnessCollection
// (halo2_proofs v2023_04_20 src/plonk/prover.rs; for zero_g that is WnnChip::predict,
// /root/reference/src/gadgets/wnn.rs:180-237, reached from Wnn::proof, /root/reference/src/wnn.rs:232-262): a host
// pass per image.  For a fixed circuit the layout does not depend on the input, so that pass can be recorded ONCE as a
// straight-line program over unsigned 256-bit integers -- image bytes in, one slot per distinct cell value out
// (harness/witness_tape.py records it from the chip code; a Rust caller records it from its own gadgets) -- and
// replayed here for any number of images: SURVEY.md 8f item 2, the term that otherwise bounds batched proving.
//
// Two kernels per batch.  `witness_run`: one workgroup per image walks the program level by level (operations of one
// dependency level are independent; a workgroup barrier separates the levels; ~10^2 levels, a few 10^4 operations).
// `witness_finish`: one lane per advice cell -- the slot its cell shows, reduced below r, into the Montgomery form, or
// zero for a cell the circuit leaves unassigned -- plus the instance values (class scores) for the host's transcript.
#include "common.h"
#include "field.h"

namespace zg {

enum : uint32_t {
    W_CONST, W_PIXEL, W_ADD, W_SUB, W_MUL, W_ADDI, W_RSUBI, W_MULI, W_SHRI, W_SHLI, W_ANDI, W_SHRV, W_GTI, W_GEI,
    W_EQI, W_DIVI, W_TABLE, W_OPS
};
constexpr uint32_t W_NO_SLOT = 0xFFFFFFFFu;

struct alignas(8) WOp {
    uint32_t op, a, b, pad;
    uint64_t imm;
};

struct alignas(16) U256 {
    uint64_t w[4];
};

__device__ __forceinline__ U256 u256_of(uint64_t v) { return U256{{v, 0, 0, 0}}; }

__device__ __forceinline__ U256 u256_add(const U256& a, const U256& b) {
    U256 r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint64_t s = a.w[i] + b.w[i];
        const uint64_t c1 = s < a.w[i];
        r.w[i] = s + c;
        c = c1 | (r.w[i] < s);
    }
    return r;
}

__device__ __forceinline__ U256 u256_sub(const U256& a, const U256& b) {
    U256 r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint64_t d = a.w[i] - b.w[i];
        const uint64_t b1 = a.w[i] < b.w[i];
        r.w[i] = d - br;
        br = b1 | (d < br);
    }
    return r;
}

__device__ __forceinline__ bool u256_ge(const U256& a, const U256& b) {
#pragma unroll
    for (int i = 3; i >= 0; i--) {
        if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
    }
    return true;
}

// low 256 bits of a * b
__device__ __forceinline__ U256 u256_mul(const U256& a, const U256& b) {
    U256 r{{0, 0, 0, 0}};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j + i < 4; j++) {
            const uint64_t lo = a.w[i] * b.w[j], hi = __umul64hi(a.w[i], b.w[j]);
            uint64_t s = r.w[i + j] + lo;
            uint64_t c = s < lo;
            s += carry;
            c += s < carry;
            r.w[i + j] = s;
            carry = hi + c;
        }
    }
    return r;
}

__device__ __forceinline__ U256 u256_shr(const U256& a, uint32_t s) {
    if (s >= 256) return U256{{0, 0, 0, 0}};
    U256 r;
    const uint32_t ws = s >> 6, bs = s & 63;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t lo = i + ws, hi = lo + 1;
        uint64_t v = lo < 4 ? a.w[lo] >> bs : 0;
        if (bs && hi < 4) v |= a.w[hi] << (64 - bs);
        r.w[i] = v;
    }
    return r;
}

__device__ __forceinline__ U256 u256_shl(const U256& a, uint32_t s) {
    if (s >= 256) return U256{{0, 0, 0, 0}};
    U256 r;
    const uint32_t ws = s >> 6, bs = s & 63;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int lo = i - (int)ws, lo1 = lo - 1;
        uint64_t v = lo >= 0 ? a.w[lo] << bs : 0;
        if (bs && lo1 >= 0) v |= a.w[lo1] >> (64 - bs);
        r.w[i] = v;
    }
    return r;
}

// a / d for a 64-bit divisor (d > 0): restoring division, one bit per step (a few dozen such operations per image)
__device__ __forceinline__ U256 u256_div64(const U256& a, uint64_t d) {
    U256 q{{0, 0, 0, 0}};
    uint64_t rem = 0;
    for (int bit = 255; bit >= 0; bit--) {
        const uint64_t top = rem >> 63;
        rem = (rem << 1) | ((a.w[bit >> 6] >> (bit & 63)) & 1);
        if (top || rem >= d) {
            rem -= d;
            q.w[bit >> 6] |= 1ull << (bit & 63);
        }
    }
    return q;
}

__device__ __forceinline__ U256 ld_u256(const U256* p) {
    const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p);
    const ulonglong2 a = q[0], b = q[1];
    return U256{{a.x, a.y, b.x, b.y}};
}
__device__ __forceinline__ void st_u256(U256* p, const U256& v) {
    ulonglong2* q = reinterpret_cast<ulonglong2*>(p);
    q[0] = make_ulonglong2(v.w[0], v.w[1]);
    q[1] = make_ulonglong2(v.w[2], v.w[3]);
}

constexpr uint32_t W_LANES = 1024;

__global__ __launch_bounds__(W_LANES) void witness_run_kernel(const WOp* __restrict__ ops, const uint32_t* __restrict__ level_start,
                                                              uint32_t n_levels, const U256* __restrict__ consts,
                                                              const uint64_t* __restrict__ table, uint32_t n_table,
                                                              const uint8_t* __restrict__ images, uint32_t image_bytes,
                                                              U256* slots_all, uint32_t n_ops) {
    const uint32_t img = blockIdx.x, tid = threadIdx.x;
    U256* slots = slots_all + (size_t)img * n_ops;
    const uint8_t* image = images + (size_t)img * image_bytes;
    for (uint32_t lv = 0; lv < n_levels; lv++) {
        const uint32_t i0 = level_start[lv], i1 = level_start[lv + 1];
        for (uint32_t i = i0 + tid; i < i1; i += W_LANES) {
            const WOp o = ops[i];
            const uint64_t imm = o.imm;
            U256 r;
            switch (o.op) {
                case W_CONST: r = ld_u256(consts + imm); break;   // (operand ranges are checked when the plan is made)
                case W_PIXEL: r = u256_of(image[imm]); break;
                case W_ADD: r = u256_add(ld_u256(slots + o.a), ld_u256(slots + o.b)); break;
                case W_SUB: r = u256_sub(ld_u256(slots + o.a), ld_u256(slots + o.b)); break;
                case W_MUL: r = u256_mul(ld_u256(slots + o.a), ld_u256(slots + o.b)); break;
                case W_ADDI: r = u256_add(ld_u256(slots + o.a), u256_of(imm)); break;
                case W_RSUBI: r = u256_sub(u256_of(imm), ld_u256(slots + o.a)); break;
                case W_MULI: r = u256_mul(ld_u256(slots + o.a), u256_of(imm)); break;
                case W_SHRI: r = u256_shr(ld_u256(slots + o.a), (uint32_t)imm); break;
                case W_SHLI: r = u256_shl(ld_u256(slots + o.a), (uint32_t)imm); break;
                case W_ANDI: r = u256_of(ld_u256(slots + o.a).w[0] & imm); break;
                case W_SHRV: {
                    const U256 s = ld_u256(slots + o.b);
                    const bool big = (s.w[1] | s.w[2] | s.w[3]) != 0 || s.w[0] >= 256;
                    r = big ? U256{{0, 0, 0, 0}} : u256_shr(ld_u256(slots + o.a), (uint32_t)s.w[0]);
                    break;
                }
                case W_GTI: {
                    const U256 a = ld_u256(slots + o.a);
                    r = u256_of(((a.w[1] | a.w[2] | a.w[3]) != 0 || a.w[0] > imm) ? 1 : 0);
                    break;
                }
                case W_GEI: {
                    const U256 a = ld_u256(slots + o.a);
                    r = u256_of(((a.w[1] | a.w[2] | a.w[3]) != 0 || a.w[0] >= imm) ? 1 : 0);
                    break;
                }
                case W_EQI: {
                    const U256 a = ld_u256(slots + o.a);
                    r = u256_of(((a.w[1] | a.w[2] | a.w[3]) == 0 && a.w[0] == imm) ? 1 : 0);
                    break;
                }
                case W_DIVI: r = u256_div64(ld_u256(slots + o.a), imm); break;
                case W_TABLE: {
                    // the index is data: an image the recorded program was not made for must not read outside the table
                    const U256 a = ld_u256(slots + o.a);
                    const uint64_t room = (uint64_t)n_table - imm;  // (imm < n_table: checked with the plan)
                    const bool ok = (a.w[1] | a.w[2] | a.w[3]) == 0 && a.w[0] < room;
                    r = u256_of(ok ? table[imm + a.w[0]] : 0);
                    break;
                }
                default: r = U256{{0, 0, 0, 0}}; break;
            }
            st_u256(slots + i, r);
        }
        __syncthreads();  // (workgroup-scope release/acquire of the global stores above)
    }
}

struct WPointers {
    Fe* advice[64];  // one lock-step batch at most per launch
};

// canonical integer < 2^256 -> the library's Montgomery form (values of an honest witness are < r already)
__device__ __forceinline__ Fe w_to_mont(const U256& v) {
    Fe x{{(uint32_t)v.w[0], (uint32_t)(v.w[0] >> 32), (uint32_t)v.w[1], (uint32_t)(v.w[1] >> 32),
          (uint32_t)v.w[2], (uint32_t)(v.w[2] >> 32), (uint32_t)v.w[3], (uint32_t)(v.w[3] >> 32)}};
#pragma unroll 1
    for (int it = 0; it < 5; it++) Fr::reduce_once(x);  // 2^256 / r < 6
    return Fr::mul(x, FrParams::r2());
}

__global__ __launch_bounds__(256) void witness_finish_kernel(const U256* __restrict__ slots_all, uint32_t n_ops,
                                                             const uint32_t* __restrict__ cell_slot, uint32_t n_cells,
                                                             const uint32_t* __restrict__ instance_slots, uint32_t n_instance,
                                                             WPointers out, Fe* __restrict__ instance_out) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x, img = blockIdx.y;
    const U256* slots = slots_all + (size_t)img * n_ops;
    if (c < n_cells) {
        const uint32_t s = cell_slot[c];
        Fe v = fe_zero();
        if (s != W_NO_SLOT) v = w_to_mont(ld_u256(slots + s));
        uint4* q = reinterpret_cast<uint4*>(out.advice[img] + c);
        q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
        q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    } else if (c - n_cells < n_instance) {
        const uint32_t j = c - n_cells;
        instance_out[(size_t)img * n_instance + j] = w_to_mont(ld_u256(slots + instance_slots[j]));
    }
}

}  // namespace zg

using namespace zg;

struct zg_witness_plan {
    zg_ctx* ctx = nullptr;
    WOp* ops = nullptr;
    uint32_t* level_start = nullptr;
    U256* consts = nullptr;
    uint64_t* table = nullptr;
    uint32_t* cell_slot = nullptr;
    uint32_t* instance_slots = nullptr;
    uint32_t n_ops = 0, n_levels = 0, n_consts = 0, n_table = 0, n_cells = 0, n_instance = 0, n_advice = 0, k = 0;
    size_t image_bytes = 0;
};

extern "C" {

int zg_witness_plan_create(zg_ctx* ctx, const zg_witness_op* ops, size_t n_ops, const uint32_t* level_start, size_t n_levels,
                           const uint64_t* consts, size_t n_consts, const uint64_t* table, size_t n_table,
                           const uint32_t* cell_slot, uint32_t n_advice, uint32_t k, const uint32_t* instance_slots,
                           size_t n_instance, size_t image_bytes, zg_witness_plan** out) {
    ZG_REQUIRE(ctx && ops && level_start && consts && table && cell_slot && out && (instance_slots || !n_instance),
               ZG_ERR_INVALID_ARG, "zg_witness_plan_create: null argument");
    ZG_REQUIRE(n_ops >= 1 && n_ops < (1u << 28) && n_levels >= 1 && n_levels <= n_ops && n_consts >= 1 && n_table >= 1 &&
                   n_table < (1ull << 32) && n_advice >= 1 && k >= 1 && k <= 24 && image_bytes >= 1,
               ZG_ERR_INVALID_ARG, "zg_witness_plan_create: sizes out of range");
    ZG_ENTER(ctx);
    // The program runs on the GPU unchecked, so everything static is checked here: levels partition the operations in
    // order, an operand is a slot of an EARLIER level (straight-line, no cycles), immediates index inside their pools.
    ZG_REQUIRE(level_start[0] == 0 && level_start[n_levels] == n_ops, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: levels do not cover the operations");
    // the WHOLE array, not only the levels that hold an operation: witness_run_kernel walks every level's
    // [level_start[l], level_start[l + 1]) and a trailing level beyond n_ops would read ops[] and write slots[] out of bounds
    for (size_t l = 0; l < n_levels; l++)
        ZG_REQUIRE(level_start[l] <= level_start[l + 1] && level_start[l + 1] <= n_ops, ZG_ERR_INVALID_ARG,
                   "zg_witness_plan_create: level %zu spans [%u, %u) of %zu operations", l, level_start[l], level_start[l + 1], n_ops);
    std::vector<WOp> dev_ops(n_ops);
    size_t lv = 0;
    for (size_t i = 0; i < n_ops; i++) {
        while (lv < n_levels && i >= level_start[lv + 1]) {
            ZG_REQUIRE(level_start[lv + 1] >= level_start[lv], ZG_ERR_INVALID_ARG, "zg_witness_plan_create: level starts decrease");
            lv++;
        }
        ZG_REQUIRE(lv < n_levels, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: operation %zu lies in no level", i);
        const zg_witness_op& o = ops[i];
        const uint64_t first_of_level = level_start[lv];
        ZG_REQUIRE(o.op < W_OPS, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: operation %zu has opcode %llu", i, (unsigned long long)o.op);
        const bool uses_a = o.op >= W_ADD, uses_b = o.op == W_ADD || o.op == W_SUB || o.op == W_MUL || o.op == W_SHRV;
        ZG_REQUIRE(!uses_a || o.a < first_of_level, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: operation %zu reads slot %llu of its own or a later level", i, (unsigned long long)o.a);
        ZG_REQUIRE(!uses_b || o.b < first_of_level, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: operation %zu reads slot %llu of its own or a later level", i, (unsigned long long)o.b);
        if (o.op == W_CONST) ZG_REQUIRE(o.imm < n_consts, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: constant %llu of %zu", (unsigned long long)o.imm, n_consts);
        if (o.op == W_PIXEL) ZG_REQUIRE(o.imm < image_bytes, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: pixel %llu of %zu", (unsigned long long)o.imm, image_bytes);
        if (o.op == W_TABLE) ZG_REQUIRE(o.imm < n_table, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: table base %llu of %zu", (unsigned long long)o.imm, n_table);
        if (o.op == W_DIVI) ZG_REQUIRE(o.imm != 0, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: division by zero in operation %zu", i);
        if (o.op == W_SHRI || o.op == W_SHLI) ZG_REQUIRE(o.imm < 256, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: shift by %llu", (unsigned long long)o.imm);
        dev_ops[i] = WOp{(uint32_t)o.op, uses_a ? (uint32_t)o.a : 0u, uses_b ? (uint32_t)o.b : 0u, 0u, o.imm};
    }
    const size_t n_cells = (size_t)n_advice << k;
    ZG_REQUIRE(n_cells < (1ull << 31), ZG_ERR_UNSUPPORTED, "zg_witness_plan_create: %zu advice cells", n_cells);
    for (size_t c = 0; c < n_cells; c++)
        ZG_REQUIRE(cell_slot[c] == W_NO_SLOT || cell_slot[c] < n_ops, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: cell %zu shows slot %u of %zu", c, cell_slot[c], n_ops);
    for (size_t j = 0; j < n_instance; j++)
        ZG_REQUIRE(instance_slots[j] < n_ops, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: instance %zu shows slot %u of %zu", j, instance_slots[j], n_ops);

    zg_witness_plan* p = new zg_witness_plan();
    p->ctx = ctx;
    p->n_ops = (uint32_t)n_ops; p->n_levels = (uint32_t)n_levels; p->n_consts = (uint32_t)n_consts; p->n_table = (uint32_t)n_table;
    p->n_cells = (uint32_t)n_cells; p->n_instance = (uint32_t)n_instance; p->n_advice = n_advice; p->k = k;
    p->image_bytes = image_bytes;
    auto up = [&](auto** dst, const void* src, size_t bytes) -> int {
        ZG_HIP(hipMalloc((void**)dst, bytes ? bytes : 1));
        if (bytes) ZG_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return ZG_OK;
    };
    int st = up(&p->ops, dev_ops.data(), n_ops * sizeof(WOp));
    if (st == ZG_OK) st = up(&p->level_start, level_start, (n_levels + 1) * sizeof(uint32_t));
    if (st == ZG_OK) st = up(&p->consts, consts, n_consts * sizeof(U256));
    if (st == ZG_OK) st = up(&p->table, table, n_table * sizeof(uint64_t));
    if (st == ZG_OK) st = up(&p->cell_slot, cell_slot, n_cells * sizeof(uint32_t));
    if (st == ZG_OK) st = up(&p->instance_slots, instance_slots, n_instance * sizeof(uint32_t));
    if (st != ZG_OK) {
        zg_witness_plan_destroy(p);
        return st;
    }
    *out = p;
    return ZG_OK;
}

void zg_witness_plan_destroy(zg_witness_plan* p) {
    if (!p) return;
    {
        std::lock_guard<std::recursive_mutex> lock(p->ctx->mu);
        (void)hipSetDevice(p->ctx->device);
        (void)hipStreamSynchronize(p->ctx->stream);
        for (void* q : {(void*)p->ops, (void*)p->level_start, (void*)p->consts, (void*)p->table, (void*)p->cell_slot, (void*)p->instance_slots})
            if (q) (void)hipFree(q);
    }
    delete p;
}

size_t zg_witness_plan_image_bytes(const zg_witness_plan* p) { return p ? p->image_bytes : 0; }
size_t zg_witness_plan_instance_len(const zg_witness_plan* p) { return p ? p->n_instance : 0; }

int zg_witness_run_dev(zg_witness_plan* p, const uint8_t* images, size_t count, void* const* d_advice, zg_fr* instance_out) {
    ZG_REQUIRE(p && images && d_advice && (instance_out || !p->n_instance), ZG_ERR_INVALID_ARG, "zg_witness_run_dev: null argument");
    ZG_REQUIRE(count <= 64, ZG_ERR_UNSUPPORTED, "zg_witness_run_dev: %zu images in one call (64 at most)", count);
    if (count == 0) return ZG_OK;
    zg_ctx* ctx = p->ctx;
    ZG_ENTER(ctx);
    WPointers ptrs;
    memset(&ptrs, 0, sizeof(ptrs));
    for (size_t i = 0; i < count; i++) {
        ZG_REQUIRE(d_advice[i] != nullptr, ZG_ERR_INVALID_ARG, "zg_witness_run_dev: advice slot %zu is null", i);
        ptrs.advice[i] = (Fe*)d_advice[i];
    }
    WsScope ws(ctx);
    U256* slots = ws.get<U256>(count * p->n_ops);
    uint8_t* d_img = ws.get<uint8_t>(count * p->image_bytes);
    Fe* d_inst = ws.get<Fe>(count * (p->n_instance ? p->n_instance : 1));
    if (ws.failed) return ZG_ERR_OOM;
    const size_t img_bytes = count * p->image_bytes, inst_bytes = count * p->n_instance * sizeof(Fe);
    ZG_TRY(pinned_reserve(ctx, img_bytes + inst_bytes + 64));
    memcpy(ctx->pinned, images, img_bytes);
    ZG_HIP(hipMemcpyAsync(d_img, ctx->pinned, img_bytes, hipMemcpyHostToDevice, ctx->stream));
    // algorithmic bytes: the image in, the advice columns and the instance values out
    const double bytes = (double)count * ((double)p->image_bytes + (double)p->n_cells * 32 + (double)p->n_instance * 32);
    ZG_LAUNCH(ctx, "witness_run", bytes, witness_run_kernel, dim3((uint32_t)count), dim3(W_LANES), 0, p->ops, p->level_start,
              p->n_levels, p->consts, p->table, p->n_table, d_img, (uint32_t)p->image_bytes, slots, p->n_ops);
    const uint32_t lanes = p->n_cells + p->n_instance;
    ZG_LAUNCH(ctx, "witness_finish", bytes, witness_finish_kernel, dim3((lanes + 255) / 256, (uint32_t)count), dim3(256), 0, slots,
              p->n_ops, p->cell_slot, p->n_cells, p->instance_slots, p->n_instance, ptrs, d_inst);
    ZG_HIP(hipGetLastError());
    if (p->n_instance) {
        void* h_inst = (char*)ctx->pinned + ((img_bytes + 63) & ~size_t(63));
        ZG_HIP(hipMemcpyAsync(h_inst, d_inst, inst_bytes, hipMemcpyDeviceToHost, ctx->stream));
        ZG_HIP(hipStreamSynchronize(ctx->stream));
        memcpy(instance_out, h_inst, inst_bytes);
    } else {
        ZG_HIP(hipStreamSynchronize(ctx->stream));  // (the staged images are reused by the next call)
    }
    return ZG_OK;
}

int zg_prover_prove_images(zg_prover* p, zg_witness_plan* plan, const uint8_t* images, size_t count, const uint8_t* rng_keys,
                           uint8_t* const* proofs, size_t proof_cap, size_t* proof_lens, zg_fr* outputs, int* statuses) {
    ZG_REQUIRE(p && plan && images && rng_keys && proofs && proof_lens && (outputs || !plan->n_instance), ZG_ERR_INVALID_ARG,
               "zg_prover_prove_images: null argument");
    ZG_REQUIRE(count >= 1 && count <= zg_prover_batch(p) && count <= 64, ZG_ERR_INVALID_ARG,
               "zg_prover_prove_images: %zu images for a prover of %zu slots (64 at most per call)", count, zg_prover_batch(p));
    void* slots[64];
    const zg_fr* inst[64];
    for (size_t b = 0; b < count; b++) {
        slots[b] = zg_prover_advice_slot(p, b);
        ZG_REQUIRE(slots[b] != nullptr, ZG_ERR_INVALID_ARG, "zg_prover_prove_images: the prover has no slot %zu", b);
        inst[b] = outputs + b * plan->n_instance;
    }
    // The program writes [n_advice][2^k] columns into the prover's slots: a plan recorded for another model (other k or
    // column count) would overrun them, so its shape must be the prover's circuit's, on the prover's device.
    const ProverShape sh = prover_shape(p);
    ZG_REQUIRE(plan->ctx->device == sh.device, ZG_ERR_INVALID_ARG, "zg_prover_prove_images: the plan lives on device %d, the prover on %d",
               plan->ctx->device, sh.device);
    ZG_REQUIRE(plan->k == sh.k && plan->n_advice == sh.n_advice, ZG_ERR_INVALID_ARG,
               "zg_prover_prove_images: the plan writes %u advice columns of 2^%u rows, the prover's circuit has %u of 2^%u", plan->n_advice,
               plan->k, sh.n_advice, sh.k);
    ZG_REQUIRE((plan->n_instance == 0 || sh.n_instance == 1) && plan->n_instance <= sh.usable_rows, ZG_ERR_INVALID_ARG,
               "zg_prover_prove_images: the plan yields %u instance values for a circuit with %u instance column(s) of %u usable rows",
               plan->n_instance, sh.n_instance, sh.usable_rows);
    // The witness runs on the plan's stream, the proofs on the prover's: when those differ (or a batch left through an
    // error return and its kernels may still read the slots) the prover's streams are drained before the slots are rewritten.
    if (sh.in_flight || plan->ctx != sh.ctx) ZG_TRY(prover_drain(p));
    ZG_TRY(zg_witness_run_dev(plan, images, count, slots, outputs));
    return zg_prover_prove_batch_dev(p, count, nullptr, inst, plan->n_instance, rng_keys, proofs, proof_cap, proof_lens, statuses);
}

}  // extern "C"
