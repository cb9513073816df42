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
// The LIVE values of the program stay in LDS (round 5): the plan is "register-allocated" when it is made -- every value
// gets its last-use level, a sound interval bound (pixels 0..255, table words < 2^64) decides whether it can exceed 64 bits,
// and a linear scan hands out 8-byte and 32-byte LDS cells that are recycled once a value's last consumer has run -- so an
// operand costs an LDS read instead of an HBM round trip and the barrier between two levels waits for LDS only (a level is
// ~0.4 us instead of ~2.5 us: three dependent global round trips -- operation, operands, store acknowledged -- became none;
// the next level's operations are prefetched while this one computes).  Every result is still stored to its slot in HBM --
// `witness_finish` reads the cells' values from there -- but nobody waits for those stores.  What does not fit the 160 KB
// (the medium and large models' live sets) keeps its operands in HBM, and only the levels that read such operands pay the
// global barrier.  ZG_WITNESS_LDS = 0 keeps every operand in HBM (the round-2 kernel's behaviour, for A/B).
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

// a / b: an operand -- a slot index in HBM, or (W_REF_LDS set) an LDS cell: W_REF_WIDE = a 32-byte cell, else an 8-byte cell.
// dst: the LDS cell the result is ALSO written to (W_NO_CELL: none; W_REF_WIDE as above); the slot in HBM is always written.
constexpr uint32_t W_REF_LDS = 0x80000000u, W_REF_WIDE = 0x40000000u, W_REF_MASK = 0x3FFFFFFFu, W_NO_CELL = 0xFFFFFFFFu;
struct alignas(8) WOp {
    uint32_t op, a, b, dst;
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

// (shifts and the division index the four words with SELECTS, never with a run-time subscript: an array indexed by a register
//  lives in scratch memory, and every scratch access is a vector-memory operation the kernel would wait for like a global one)
__device__ __forceinline__ U256 u256_shr(const U256& a, uint32_t s) {
    if (s >= 256) return U256{{0, 0, 0, 0}};
    const uint32_t ws = s >> 6, bs = s & 63;
    uint64_t w0 = a.w[0], w1 = a.w[1], w2 = a.w[2], w3 = a.w[3];
    if (ws & 1) { w0 = w1; w1 = w2; w2 = w3; w3 = 0; }
    if (ws & 2) { w0 = w2; w1 = w3; w2 = 0; w3 = 0; }
    if (bs) {
        w0 = (w0 >> bs) | (w1 << (64 - bs));
        w1 = (w1 >> bs) | (w2 << (64 - bs));
        w2 = (w2 >> bs) | (w3 << (64 - bs));
        w3 >>= bs;
    }
    return U256{{w0, w1, w2, w3}};
}

__device__ __forceinline__ U256 u256_shl(const U256& a, uint32_t s) {
    if (s >= 256) return U256{{0, 0, 0, 0}};
    const uint32_t ws = s >> 6, bs = s & 63;
    uint64_t w0 = a.w[0], w1 = a.w[1], w2 = a.w[2], w3 = a.w[3];
    if (ws & 1) { w3 = w2; w2 = w1; w1 = w0; w0 = 0; }
    if (ws & 2) { w3 = w1; w2 = w0; w1 = 0; w0 = 0; }
    if (bs) {
        w3 = (w3 << bs) | (w2 >> (64 - bs));
        w2 = (w2 << bs) | (w1 >> (64 - bs));
        w1 = (w1 << bs) | (w0 >> (64 - bs));
        w0 <<= bs;
    }
    return U256{{w0, w1, w2, w3}};
}

// a / d for a 64-bit divisor (d > 0): restoring division, one bit per step, word by word from the top (a few dozen such
// operations per image)
__device__ __forceinline__ U256 u256_div64(const U256& a, uint64_t d) {
    uint64_t q[4] = {0, 0, 0, 0};
    uint64_t rem = 0;
#pragma unroll
    for (int wi = 3; wi >= 0; wi--) {
        const uint64_t x = a.w[wi];
        uint64_t qw = 0;
#pragma unroll 1
        for (int bit = 63; bit >= 0; bit--) {
            const uint64_t top = rem >> 63;
            rem = (rem << 1) | ((x >> bit) & 1);
            qw <<= 1;
            if (top || rem >= d) {
                rem -= d;
                qw |= 1;
            }
        }
        q[wi] = qw;
    }
    return U256{{q[0], q[1], q[2], q[3]}};
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

// A value that came from global memory is waited for WHERE IT WAS LOADED: the compiler tracks pending loads per register and,
// where branches meet, waits for whatever any of them left pending -- with vmcnt(0), since loads and stores share one counter on
// gfx9 -- so a load left pending in ONE case of the interpreter's switch would make EVERY operation wait for the stores and
// prefetches in flight.  The empty asm uses the registers, so the wait lands here.
__device__ __forceinline__ void w_settle(U256& v) { asm volatile("" : "+v"(v.w[0]), "+v"(v.w[1]), "+v"(v.w[2]), "+v"(v.w[3])); }
__device__ __forceinline__ void w_settle(uint64_t& v) { asm volatile("" : "+v"(v)); }

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

// one operation on operands already fetched (the same semantics as witness_run_kernel's switch, operand for operand)
__device__ __forceinline__ U256 w_eval(uint32_t op, uint64_t imm, const U256& a, const U256& b, const uint8_t* image,
                                       const U256* __restrict__ consts, const uint64_t* __restrict__ table, uint32_t n_table) {
    switch (op) {
        case W_CONST: {
            U256 c = ld_u256(consts + imm);
            w_settle(c);
            return c;
        }
        case W_PIXEL: return u256_of(image[imm]);
        case W_ADD: return u256_add(a, b);
        case W_SUB: return u256_sub(a, b);
        case W_MUL: return u256_mul(a, b);
        case W_ADDI: return u256_add(a, u256_of(imm));
        case W_RSUBI: return u256_sub(u256_of(imm), a);
        case W_MULI: return u256_mul(a, u256_of(imm));
        case W_SHRI: return u256_shr(a, (uint32_t)imm);
        case W_SHLI: return u256_shl(a, (uint32_t)imm);
        case W_ANDI: return u256_of(a.w[0] & imm);
        case W_SHRV: {
            const bool big = (b.w[1] | b.w[2] | b.w[3]) != 0 || b.w[0] >= 256;
            return big ? U256{{0, 0, 0, 0}} : u256_shr(a, (uint32_t)b.w[0]);
        }
        case W_GTI: return u256_of(((a.w[1] | a.w[2] | a.w[3]) != 0 || a.w[0] > imm) ? 1 : 0);
        case W_GEI: return u256_of(((a.w[1] | a.w[2] | a.w[3]) != 0 || a.w[0] >= imm) ? 1 : 0);
        case W_EQI: return u256_of(((a.w[1] | a.w[2] | a.w[3]) == 0 && a.w[0] == imm) ? 1 : 0);
        case W_DIVI: return u256_div64(a, imm);
        case W_TABLE: {
            // the index is data: an image the recorded program was not made for must not read outside the table
            const uint64_t room = (uint64_t)n_table - imm;  // (imm < n_table: checked with the plan)
            const bool ok = (a.w[1] | a.w[2] | a.w[3]) == 0 && a.w[0] < room;
            uint64_t t = 0;
            if (ok) {
                t = table[imm + a.w[0]];
                w_settle(t);
            }
            return u256_of(t);
        }
        default: return U256{{0, 0, 0, 0}};
    }
}

// An operation whose operands and result are all below 2^64 by the plan's interval bounds (W_NARROW in WOp::op) runs on
// 64-bit integers: an addition is two instructions instead of a four-word carry chain, a product four instead of ~200.
constexpr uint32_t W_NARROW = 0x100u;
__device__ __forceinline__ uint64_t w_eval64(uint32_t op, uint64_t imm, uint64_t a, uint64_t b, const uint8_t* image,
                                             const U256* __restrict__ consts, const uint64_t* __restrict__ table, uint32_t n_table) {
    switch (op) {
        case W_CONST: {
            uint64_t c = consts[imm].w[0];
            w_settle(c);
            return c;
        }
        case W_PIXEL: return image[imm];
        case W_ADD: return a + b;
        case W_SUB: return a - b;
        case W_MUL: return a * b;
        case W_ADDI: return a + imm;
        case W_RSUBI: return imm - a;
        case W_MULI: return a * imm;
        case W_SHRI: return imm >= 64 ? 0 : a >> imm;
        case W_SHLI: return imm >= 64 ? 0 : a << imm;
        case W_ANDI: return a & imm;
        case W_SHRV: return b >= 64 ? 0 : a >> b;
        case W_GTI: return a > imm ? 1 : 0;
        case W_GEI: return a >= imm ? 1 : 0;
        case W_EQI: return a == imm ? 1 : 0;
        case W_DIVI: return a / imm;
        case W_TABLE: {
            const uint64_t room = (uint64_t)n_table - imm;
            uint64_t t = 0;
            if (a < room) {
                t = table[imm + a];
                w_settle(t);
            }
            return t;
        }
        default: return 0;
    }
}

#ifdef ZG_WITNESS_TRACE
__device__ uint64_t zg_w_trace[4096];
#endif
// The same walk with the live values in LDS (see the head of this file), as a stream of EPOCHS: the operations are consumed
// one workgroup's worth at a time in program order (which is level order), lane t of epoch e owns operation e * lanes + t and
// holds it in registers -- fetched an epoch ahead --, runs it when its level comes up (a level may end inside an epoch:
// barrier; or span several epochs: no barrier in between), keeps the result in registers as well and stores it to its slot
// in HBM at the END of the epoch.  On gfx9 loads and stores share one completion counter and the compiler can only wait for
// all of them at once: with the store issued right AFTER the one wait of the epoch, everything that wait covers is old.
// LDS: cap_w 32-byte cells, cap_n 8-byte cells, the image, the level table (bit 31 of an entry: that level reads an operand
// from HBM -- the barrier in front of it waits for the stores, too; a result some operation reads from HBM is stored at once,
// W_STORE_NOW in dst).  ALL_LDS: no operand of the program lives in HBM (the tiny model's case).
// Dispatch: the operations of a level are sorted by opcode, so a wave mostly holds ONE opcode: then the interpreter's switch
// runs on a scalar (uniform branches, no exec-mask juggling); mixed waves take the per-lane switch.
constexpr uint32_t W_STORE_NOW = 0x20000000u;  // (in WOp::dst beside W_REF_WIDE; the cell index keeps the low 29 bits)
constexpr uint32_t W_CELL_MASK = 0x1FFFFFFFu;
constexpr uint32_t W_LEVEL_HBM = 0x80000000u;  // (in the LDS level table: the level reads operands from HBM)

template <bool ALL_LDS>
__global__ __launch_bounds__(W_LANES) void witness_run_lds_kernel(const WOp* __restrict__ ops, const uint32_t* __restrict__ level_start,
                                                                  const uint8_t* __restrict__ level_flags, uint32_t n_levels,
                                                                  const U256* __restrict__ consts, const uint64_t* __restrict__ table,
                                                                  uint32_t n_table, const uint8_t* __restrict__ images, uint32_t image_bytes,
                                                                  U256* slots_all, uint32_t n_ops, uint32_t cap_n, uint32_t cap_w) {
    const uint32_t W_LANES = blockDim.x;  // (the plan picks the workgroup: small programs leave fewer waves idling at every barrier)
    extern __shared__ __align__(16) unsigned char w_smem[];
    U256* wid = reinterpret_cast<U256*>(w_smem);
    uint64_t* nar = reinterpret_cast<uint64_t*>(wid + cap_w);
    uint8_t* img = reinterpret_cast<uint8_t*>(nar + cap_n);
    uint32_t* ls = reinterpret_cast<uint32_t*>(img + ((image_bytes + 15u) & ~15u));  // [n_levels + 3]: bounds | W_LEVEL_HBM
    const uint32_t blk = blockIdx.x, tid = threadIdx.x;
    U256* slots = slots_all + (size_t)blk * n_ops;
    const uint8_t* image = images + (size_t)blk * image_bytes;
    for (uint32_t j = tid; j < image_bytes; j += W_LANES) img[j] = image[j];
    for (uint32_t j = tid; j <= n_levels + 2; j += W_LANES)
        ls[j] = j <= n_levels ? (level_start[j] | ((level_flags[j] & 1) ? W_LEVEL_HBM : 0u)) : n_ops;
    // two operation registers, used in turn: epoch e runs out of one while the other holds epoch e + 1's (requested an epoch
    // earlier) -- two NAMED buffers and a loop unrolled by two, so that a load lands in the register it is used from
    WOp op_a{}, op_b{};
    if (tid < n_ops) op_a = ops[tid];
    if (tid + W_LANES < n_ops) op_b = ops[tid + W_LANES];
    __syncthreads();
    auto fetch = [&](uint32_t ref) -> U256 {
        if (ALL_LDS || (ref & W_REF_LDS)) {
            const uint32_t c = ref & W_CELL_MASK;
            if (ref & W_REF_WIDE) {
                const ulonglong2* q = reinterpret_cast<const ulonglong2*>(wid + c);
                const ulonglong2 x = q[0], y = q[1];
                return U256{{x.x, x.y, y.x, y.y}};
            }
            uint64_t v = nar[c];
            if (!ALL_LDS) asm volatile("" : "+v"(v));  // (see fetch64)
            return u256_of(v);
        }
        U256 g = ld_u256(slots + ref);
        w_settle(g);
        return g;
    };
    auto fetch64 = [&](uint32_t ref) -> uint64_t {  // (an operand the plan knows to be below 2^64: an 8-byte cell, or word 0 of its slot)
        if (ALL_LDS) return nar[ref & W_CELL_MASK];
        uint64_t v;  // (two loads in two branches: a pointer picked from both address spaces would make it a FLAT load, which
        if (ref & W_REF_LDS) {                            //  counts as LDS and vector memory at once and turns every wait into "all";
            v = nar[ref & W_CELL_MASK];                   //  the empty asm keeps the compiler from merging the branches again)
            asm volatile("" : "+v"(v));
        } else {
            v = slots[ref].w[0];
            w_settle(v);
        }
        return v;
    };
    uint32_t lv = 0;
    uint32_t l_end_f = ls[1], l_next_f = ls[2];  // this level's end (with the NEXT level's HBM flag: it sits in entry lv + 1), the next one's
    const uint32_t n_epochs = (n_ops + W_LANES - 1) / W_LANES;
#ifdef ZG_WITNESS_TRACE
    uint32_t tr_n = 0;
#define W_TRACE(tag) do { if (tid == 0 && blk == 0 && tr_n < 4000) { zg_w_trace[tr_n++] = ((uint64_t)(tag) << 56) | (__builtin_readcyclecounter() & 0xFFFFFFFFFFFFFFull); } } while (0)
#else
#define W_TRACE(tag) do {} while (0)
#endif
    auto epoch = [&](const uint32_t e, WOp& o, WOp& other) {
        const uint32_t idx = e * W_LANES + tid;
        const uint32_t e_end = (e + 1) * W_LANES < n_ops ? (e + 1) * W_LANES : n_ops;
        bool pending = idx < n_ops, deferred = false;
        U256 r{{0, 0, 0, 0}};
        while (lv < n_levels) {
            const uint32_t l_end = l_end_f & ~W_LEVEL_HBM;  // (uniform: every lane walks the same levels)
            const uint32_t l_after = ls[lv + 3];            // (requested now, needed after the barrier)
            W_TRACE(1);
            if (pending && idx < l_end) {
                pending = false;
                const uint32_t op_u = __builtin_amdgcn_readfirstlane(o.op);
                if ((op_u & W_NARROW) && __builtin_amdgcn_ballot_w64(o.op != op_u) == 0) {
                    // one narrow operation for every active lane of the wave: the switch runs on a scalar
                    const uint32_t opc = op_u & 0xFFu;
                    const bool ua = opc >= W_ADD, ub = opc == W_ADD || opc == W_SUB || opc == W_MUL || opc == W_SHRV;
                    uint64_t a = 0, b = 0;
                    if (ua) a = fetch64(o.a);
                    if (ub) b = fetch64(o.b);
                    r = u256_of(w_eval64(opc, o.imm, a, b, img, consts, table, n_table));
                } else {
                    const uint32_t opc = o.op & 0xFFu;
                    const bool ua = opc >= W_ADD, ub = opc == W_ADD || opc == W_SUB || opc == W_MUL || opc == W_SHRV;
                    if (o.op & W_NARROW) {
                        const uint64_t a = ua ? fetch64(o.a) : 0, b = ub ? fetch64(o.b) : 0;
                        r = u256_of(w_eval64(opc, o.imm, a, b, img, consts, table, n_table));
                    } else {
                        U256 a{{0, 0, 0, 0}}, b{{0, 0, 0, 0}};
                        if (ua) a = fetch(o.a);
                        if (ub) b = fetch(o.b);
                        r = w_eval(opc, o.imm, a, b, img, consts, table, n_table);
                    }
                }
                if (o.dst != W_NO_CELL) {
                    if (!ALL_LDS && (o.dst & W_STORE_NOW)) {
                        st_u256(slots + idx, r);  // (an operation of a later level reads this one from HBM)
                    } else {
                        deferred = true;
                        const uint32_t c = o.dst & W_CELL_MASK;
                        if (o.dst & W_REF_WIDE) {
                            ulonglong2* q = reinterpret_cast<ulonglong2*>(wid + c);
                            q[0] = make_ulonglong2(r.w[0], r.w[1]);
                            q[1] = make_ulonglong2(r.w[2], r.w[3]);
                        } else {
                            nar[c] = r.w[0];  // (the plan's interval bound says the value is below 2^64 for EVERY image)
                        }
                    }
                } else {
                    deferred = true;  // nobody reads it in this kernel: witness_finish does, from HBM
                }
            }
            W_TRACE(2);
            if (l_end > e_end) break;  // the level goes on in the next epoch: no barrier
            // level lv ends inside this epoch
            lv++;
            if (!ALL_LDS && lv < n_levels && (l_end_f & W_LEVEL_HBM)) {
                __syncthreads();  // (workgroup-scope release / acquire of the global stores: the next level reads operands from HBM)
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS only: stores to the slots stay in flight
            }
            l_end_f = l_next_f;
            l_next_f = l_after;
            W_TRACE(3);
            if (l_end == e_end) break;
        }
        W_TRACE(4);
        // ---- the epoch's one wait: the OTHER buffer (requested an epoch ago) and, with it, the stores of the previous boundary
        asm volatile("" ::"v"(other.op), "v"(other.a), "v"(other.b), "v"(other.dst), "v"(other.imm) : "memory");
        if (deferred) st_u256(slots + idx, r);
        W_TRACE(5);
        if (idx + 2 * W_LANES < n_ops) o = ops[idx + 2 * W_LANES];  // this buffer's next turn is two epochs away
        W_TRACE(6);
    };
    for (uint32_t e = 0; e < n_epochs; e += 2) {
        epoch(e, op_a, op_b);
        if (e + 1 < n_epochs) epoch(e + 1, op_b, op_a);
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
    // the LDS form (witness_run_lds_kernel): ops carry cell references, level_flags names the levels that read from HBM
    bool lds = false;
    uint8_t* level_flags = nullptr;
    uint32_t cap_n = 0, cap_w = 0, lanes = 1024;
    size_t lds_bytes = 0;
    uint64_t info[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // zg_witness_plan_info
};

namespace {

// LDS the live values may take (of the CU's 160 KB; the image and some slack stay outside)
constexpr size_t W_LDS_BUDGET = 150u << 10;

// "Register allocation" of a recorded program (head of this file).  dev_ops come in with plain slot indices and leave with
// cell references; flags[l] |= 1 where level l still reads an operand from HBM.
struct WAllocation {
    uint32_t cap_n = 0, cap_w = 0;
    uint64_t in_lds = 0, in_hbm = 0, wide_values = 0, max_live_n = 0, max_live_w = 0, hbm_levels = 0, narrow_ops = 0;
};

WAllocation witness_allocate(std::vector<WOp>& dev_ops, const uint32_t* level_start, size_t n_levels, const uint64_t* consts,
                             std::vector<uint8_t>& flags, size_t budget) {
    typedef unsigned __int128 u128;
    const size_t n = dev_ops.size();
    WAllocation out;
    // ---- sound interval bounds: may the value exceed 64 bits for ANY image (pixels 0..255) and ANY table content?
    std::vector<uint64_t> lo(n), hi(n);
    std::vector<uint8_t> wide(n, 0), src_wide(n, 0);
    for (size_t i = 0; i < n; i++) {
        const WOp& o = dev_ops[i];
        const bool ua = o.op >= W_ADD, ub = o.op == W_ADD || o.op == W_SUB || o.op == W_MUL || o.op == W_SHRV;
        const bool wa = ua && wide[o.a], wb = ub && wide[o.b];
        const uint64_t la = ua ? lo[o.a] : 0, ha = ua ? hi[o.a] : 0, lb = ub ? lo[o.b] : 0, hb = ub ? hi[o.b] : 0, imm = o.imm;
        bool w = false;
        uint64_t l = 0, h = 0;
        switch (o.op) {
            case W_CONST: {
                const uint64_t* c = consts + 4 * imm;
                if (c[1] | c[2] | c[3]) w = true; else l = h = c[0];
                break;
            }
            case W_PIXEL: l = 0; h = 255; break;
            case W_ADD: {
                const u128 t = (u128)ha + hb;
                if (wa || wb || (t >> 64)) w = true; else { h = (uint64_t)t; l = la + lb; }
                break;
            }
            case W_SUB: if (wa || wb || la < hb) w = true; else { l = la - hb; h = ha - lb; } break;  // (may wrap below zero otherwise)
            case W_MUL: {
                const u128 t = (u128)ha * hb;
                if (wa || wb || (t >> 64)) w = true; else { h = (uint64_t)t; l = la * lb; }
                break;
            }
            case W_ADDI: {
                const u128 t = (u128)ha + imm;
                if (wa || (t >> 64)) w = true; else { h = (uint64_t)t; l = la + imm; }
                break;
            }
            case W_RSUBI: if (wa || imm < ha) w = true; else { l = imm - ha; h = imm - la; } break;
            case W_MULI: {
                const u128 t = (u128)ha * imm;
                if (wa || (t >> 64)) w = true; else { h = (uint64_t)t; l = la * imm; }
                break;
            }
            case W_SHRI:
                if (wa) { w = imm < 192; l = 0; h = ~0ull; }  // (a < 2^256: from 192 positions on the result is below 2^64)
                else { l = imm >= 64 ? 0 : la >> imm; h = imm >= 64 ? 0 : ha >> imm; }
                break;
            case W_SHLI:
                if (wa) w = true;
                else if (imm >= 64) { w = ha != 0; l = h = 0; }
                else if (imm && (ha >> (64 - imm))) w = true;
                else { h = ha << imm; l = la << imm; }
                break;
            case W_ANDI: l = 0; h = wa ? imm : (ha < imm ? ha : imm); break;  // (w[0] & imm)
            case W_SHRV: if (wa) w = true; else { l = 0; h = ha; } break;      // (a >> anything <= a)
            case W_GTI: case W_GEI: case W_EQI: l = 0; h = 1; break;
            case W_DIVI: if (wa) w = true; else { l = la / imm; h = ha / imm; } break;
            case W_TABLE: l = 0; h = ~0ull; break;
            default: w = true; break;
        }
        src_wide[i] = (uint8_t)((wa ? 1 : 0) | (wb ? 2 : 0));
        wide[i] = w ? 1 : 0;
        lo[i] = w ? 0 : l;
        hi[i] = w ? ~0ull : h;
        out.wide_values += w ? 1 : 0;
    }
    // ---- levels and last uses
    std::vector<uint32_t> level_of(n);
    for (size_t l = 0; l < n_levels; l++)
        for (uint32_t i = level_start[l]; i < level_start[l + 1]; i++) level_of[i] = (uint32_t)l;
    std::vector<int64_t> last(n, -1);
    for (size_t i = 0; i < n; i++) {
        const WOp& o = dev_ops[i];
        const bool ua = o.op >= W_ADD, ub = o.op == W_ADD || o.op == W_SUB || o.op == W_MUL || o.op == W_SHRV;
        if (ua && last[o.a] < (int64_t)level_of[i]) last[o.a] = level_of[i];
        if (ub && last[o.b] < (int64_t)level_of[i]) last[o.b] = level_of[i];
    }
    // ---- how many values of each class are alive at once (a value holds its cell from its level to its last consumer's)
    std::vector<int64_t> dn(n_levels + 2, 0), dw(n_levels + 2, 0);
    for (size_t i = 0; i < n; i++)
        if (last[i] >= 0) {
            auto& d = wide[i] ? dw : dn;
            d[level_of[i]] += 1;
            d[(size_t)last[i] + 1] -= 1;
        }
    int64_t cn = 0, cw = 0;
    for (size_t l = 0; l <= n_levels; l++) {
        cn += dn[l]; cw += dw[l];
        if ((uint64_t)cn > out.max_live_n) out.max_live_n = (uint64_t)cn;
        if ((uint64_t)cw > out.max_live_w) out.max_live_w = (uint64_t)cw;
    }
    uint64_t cap_w = out.max_live_w, cap_n = out.max_live_n;
    if (32 * cap_w + 8 * cap_n > budget) {  // not everything fits: a quarter of the room for the wide class at most, the rest narrow
        if (cap_w > budget / 4 / 32) cap_w = budget / 4 / 32;
        cap_n = (budget - 32 * cap_w) / 8;
        if (cap_n > out.max_live_n) {
            cap_n = out.max_live_n;
            cap_w = (budget - 8 * cap_n) / 32;
            if (cap_w > out.max_live_w) cap_w = out.max_live_w;
        }
    }
    out.cap_n = (uint32_t)cap_n;
    out.cap_w = (uint32_t)cap_w;
    // ---- linear scan over the levels: a cell is free again from the level AFTER its value's last consumer
    std::vector<std::vector<uint32_t>> by_last(n_levels);
    for (size_t i = 0; i < n; i++)
        if (last[i] >= 0) by_last[(size_t)last[i]].push_back((uint32_t)i);
    std::vector<uint32_t> cell(n, W_NO_CELL), free_n, free_w;
    uint32_t next_n = 0, next_w = 0;
    for (size_t l = 0; l < n_levels; l++) {
        if (l > 0)
            for (uint32_t s_ : by_last[l - 1])
                if (cell[s_] != W_NO_CELL) (cell[s_] & W_REF_WIDE ? free_w : free_n).push_back(cell[s_] & W_REF_MASK);
        for (uint32_t i = level_start[l]; i < level_start[l + 1]; i++) {
            if (last[i] < 0) continue;  // nobody reads it: the slot in HBM is all it needs
            auto& fr = wide[i] ? free_w : free_n;
            uint32_t& next = wide[i] ? next_w : next_n;
            const uint32_t cap = wide[i] ? out.cap_w : out.cap_n;
            uint32_t c = W_NO_CELL;
            if (!fr.empty()) { c = fr.back(); fr.pop_back(); }
            else if (next < cap) c = next++;
            if (c != W_NO_CELL) { cell[i] = c | (wide[i] ? W_REF_WIDE : 0u); out.in_lds++; }
            else out.in_hbm++;
        }
    }
    // ---- rewrite the operations
    flags.assign(n_levels + 1, 0);
    for (size_t i = 0; i < n; i++) {
        WOp& o = dev_ops[i];
        const bool ua = o.op >= W_ADD, ub = o.op == W_ADD || o.op == W_SUB || o.op == W_MUL || o.op == W_SHRV;
        bool from_hbm = false;
        if (ua) { if (cell[o.a] != W_NO_CELL) o.a = W_REF_LDS | cell[o.a]; else from_hbm = true; }
        if (ub) { if (cell[o.b] != W_NO_CELL) o.b = W_REF_LDS | cell[o.b]; else from_hbm = true; }
        // dst: the LDS cell; or "no cell, but an operation reads it from HBM: store at once" (cell index all ones); or nothing
        o.dst = cell[i] != W_NO_CELL ? cell[i] : (last[i] >= 0 ? (W_STORE_NOW | W_CELL_MASK) : W_NO_CELL);
        if (!wide[i] && !(ua && src_wide[i] & 1) && !(ub && src_wide[i] & 2)) { o.op |= W_NARROW; out.narrow_ops++; }
        if (from_hbm) flags[level_of[i]] |= 1;
    }
    for (size_t l = 0; l < n_levels; l++) out.hbm_levels += flags[l] & 1;
    return out;
}

}  // namespace

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
        dev_ops[i] = WOp{(uint32_t)o.op, uses_a ? (uint32_t)o.a : 0u, uses_b ? (uint32_t)o.b : 0u, W_NO_CELL, o.imm};
    }
    const size_t n_cells = (size_t)n_advice << k;
    ZG_REQUIRE(n_cells < (1ull << 31), ZG_ERR_UNSUPPORTED, "zg_witness_plan_create: %zu advice cells", n_cells);
    for (size_t c = 0; c < n_cells; c++)
        ZG_REQUIRE(cell_slot[c] == W_NO_SLOT || cell_slot[c] < n_ops, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: cell %zu shows slot %u of %zu", c, cell_slot[c], n_ops);
    for (size_t j = 0; j < n_instance; j++)
        ZG_REQUIRE(instance_slots[j] < n_ops, ZG_ERR_INVALID_ARG, "zg_witness_plan_create: instance %zu shows slot %u of %zu", j, instance_slots[j], n_ops);

    // the LDS form (ZG_WITNESS_LDS, default on): live values in LDS cells, recycled after their last consumer
    std::vector<uint8_t> level_flags(n_levels + 1, 1);
    WAllocation al;
    // (ZG_WITNESS_LDS: 0 = every operand in HBM, 1 / unset = the whole budget, n >= 2 = n KB of cells: tests force spills with it)
    const int lds_knob = knob(K_WITNESS_LDS);
    // (beside the cells: the image, the level bounds and the level flags; a program whose image or level list would take
    //  most of the LDS by itself keeps its operands in HBM)
    const size_t img_room = ((image_bytes + 15) & ~size_t(15)) + (((n_levels + 3) * 4 + 15) & ~size_t(15));
    const bool lds = lds_knob != 0 && img_room + (32u << 10) < W_LDS_BUDGET;
    if (lds) {
        size_t budget = W_LDS_BUDGET - img_room;
        if (lds_knob >= 2 && ((size_t)lds_knob << 10) < budget) budget = (size_t)lds_knob << 10;
        al = witness_allocate(dev_ops, level_start, n_levels, consts, level_flags, budget);
    }

    zg_witness_plan* p = new zg_witness_plan();
    p->ctx = ctx;
    p->lds = lds;
    p->cap_n = al.cap_n; p->cap_w = al.cap_w;
    p->lds_bytes = (size_t)32 * al.cap_w + (size_t)8 * al.cap_n + img_room;
    p->info[0] = lds ? p->lds_bytes : 0; p->info[1] = al.cap_n; p->info[2] = al.cap_w; p->info[3] = al.in_lds; p->info[4] = al.in_hbm;
    p->info[5] = lds ? al.hbm_levels : n_levels; p->info[6] = al.wide_values; p->info[7] = n_levels;
    // the workgroup: twice the mean level, between 4 and 16 waves (idle waves still walk every level and meet every barrier)
    {
        const size_t mean2 = 2 * n_ops / (n_levels ? n_levels : 1);
        uint32_t lanes = 256;
        while (lanes < 1024 && lanes < mean2) lanes *= 2;
        p->lanes = lds ? lanes : 1024;
    }
    p->info[8] = al.narrow_ops; p->info[9] = p->lanes;
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
    if (st == ZG_OK) st = up(&p->level_flags, level_flags.data(), level_flags.size());
    if (st == ZG_OK && lds &&
        (hipFuncSetAttribute((const void*)witness_run_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
         hipFuncSetAttribute((const void*)witness_run_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)) {
        set_error("zg_witness_plan_create: the device refuses 160 KB of dynamic LDS");
        st = ZG_ERR_HIP;
    }
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
        for (void* q : {(void*)p->ops, (void*)p->level_start, (void*)p->consts, (void*)p->table, (void*)p->cell_slot, (void*)p->instance_slots,
                        (void*)p->level_flags})
            if (q) (void)hipFree(q);
    }
    delete p;
}

size_t zg_witness_plan_image_bytes(const zg_witness_plan* p) { return p ? p->image_bytes : 0; }
size_t zg_witness_plan_instance_len(const zg_witness_plan* p) { return p ? p->n_instance : 0; }
int zg_witness_plan_info(const zg_witness_plan* p, uint64_t* out, size_t cap) {
    ZG_REQUIRE(p && out, ZG_ERR_INVALID_ARG, "zg_witness_plan_info: null argument");
    for (size_t i = 0; i < cap && i < 10; i++) out[i] = p->info[i];
    return ZG_OK;
}

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
    if (p->lds && p->info[4] == 0)  // (no consumed value was left in HBM: the variant without the HBM operand paths)
        ZG_LAUNCH(ctx, "witness_run", bytes, witness_run_lds_kernel<true>, dim3((uint32_t)count), dim3(p->lanes), p->lds_bytes, p->ops, p->level_start,
                  p->level_flags, p->n_levels, p->consts, p->table, p->n_table, d_img, (uint32_t)p->image_bytes, slots, p->n_ops, p->cap_n, p->cap_w);
    else if (p->lds)
        ZG_LAUNCH(ctx, "witness_run", bytes, witness_run_lds_kernel<false>, dim3((uint32_t)count), dim3(p->lanes), p->lds_bytes, p->ops, p->level_start,
                  p->level_flags, p->n_levels, p->consts, p->table, p->n_table, d_img, (uint32_t)p->image_bytes, slots, p->n_ops, p->cap_n, p->cap_w);
    else
        ZG_LAUNCH(ctx, "witness_run", bytes, witness_run_kernel, dim3((uint32_t)count), dim3(W_LANES), 0, p->ops, p->level_start,
                  p->n_levels, p->consts, p->table, p->n_table, d_img, (uint32_t)p->image_bytes, slots, p->n_ops);
    const uint32_t lanes = p->n_cells + p->n_instance;
    ZG_LAUNCH(ctx, "witness_finish", bytes, witness_finish_kernel, dim3((lanes + 255) / 256, (uint32_t)count), dim3(256), 0, slots,
              p->n_ops, p->cell_slot, p->n_cells, p->instance_slots, p->n_instance, ptrs, d_inst);
    ZG_HIP(hipGetLastError());
#ifdef ZG_WITNESS_TRACE
    if (p->lds) {
        static int dumped = 0;
        (void)hipStreamSynchronize(ctx->stream);
        std::vector<uint64_t> tr(4096);
        (void)hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(zg_w_trace), 4096 * 8);
        if (dumped++ == 5) {
            uint64_t prev = 0;
            for (int i = 0; i < 4000 && tr[i]; i++) {
                const uint64_t t = tr[i] & 0xFFFFFFFFFFFFFFull;
                fprintf(stderr, "T %d %llu\n", (int)(tr[i] >> 56), (unsigned long long)(prev ? t - prev : 0));
                prev = t;
            }
        }
        std::vector<uint64_t> z(4096, 0);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(zg_w_trace), z.data(), 4096 * 8);
    }
#endif
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
