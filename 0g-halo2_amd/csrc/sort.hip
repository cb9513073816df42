// lookup::prover::permute_expression_pair on the device (halo2_proofs v2023_04_20
// src/plonk/lookup/prover.rs; upstream sorts with Vec::sort + a BTreeMap on one core).
//
//   a' = the compressed input column's usable rows, sorted by canonical value (Fr: Ord);
//   s' = at the first row of every distinct a' value that same value (one instance is taken out of
//        the table multiset), at the repeated rows the left-over table values: ascending left-overs
//        go to the repeated rows taken from the END (upstream pops a Vec of repeated rows).
//
// Keys are canonical 256-bit integers.  Sorting is a bitonic network: 1024-key chunks entirely in
// LDS, global compare-exchange passes only for strides >= 1024.  The rest is flags + prefix sums:
// every distinct a' value binary-searches the sorted table and consumes that run's first element;
// unconsumed table entries and repeated rows are compacted with scans and matched in reverse.
#include "poly.h"

namespace zg {

__device__ __forceinline__ Fe ldk(const Fe* p) {
    Fe r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void stk(Fe* p, const Fe& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// canonical integers: -1 / 0 / +1
__device__ __forceinline__ int key_cmp(const Fe& a, const Fe& b) {
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (a.l[i] < b.l[i]) return -1;
        if (a.l[i] > b.l[i]) return 1;
    }
    return 0;
}

constexpr uint32_t SORT_CH = 1024;  // keys per LDS chunk (32 KB)
constexpr uint32_t SORT_NT = 512;

// keys[b][i] for i >= usable <- all-ones sentinel (sorts last; real keys are < r < 2^254)
__global__ void sort_pad_kernel(Fe* keys, uint32_t n, uint32_t usable) {
    uint32_t i = usable + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe s;
#pragma unroll
    for (int j = 0; j < 8; j++) s.l[j] = 0xffffffffu;
    stk(keys + (size_t)blockIdx.y * n + i, s);
}

__device__ __forceinline__ void lds_cmpx(Fe* sh, uint32_t i, uint32_t j, bool ascending) {
    Fe a = sh[i], b = sh[j];
    const bool swap = (key_cmp(a, b) > 0) == ascending;
    if (swap) {
        sh[i] = b;
        sh[j] = a;
    }
}

// Stage A: every 1024-key chunk becomes a sorted run, direction alternating with the chunk's
// position in the size-2048 bitonic pattern (bit `SORT_CH` of the global index).
__global__ __launch_bounds__(SORT_NT) void sort_local_kernel(Fe* keys, uint32_t n) {
    __shared__ Fe sh[SORT_CH];
    Fe* base = keys + (size_t)blockIdx.y * n + (size_t)blockIdx.x * SORT_CH;
    const uint32_t g0 = blockIdx.x * SORT_CH;
    const uint32_t len = n < SORT_CH ? n : SORT_CH;
    for (uint32_t e = threadIdx.x; e < len; e += SORT_NT) sh[e] = ldk(base + e);
    __syncthreads();
    for (uint32_t size = 2; size <= len; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < len / 2; t += SORT_NT) {
                uint32_t i = 2 * t - (t & (stride - 1));  // index with bit `stride` clear
                uint32_t j = i + stride;
                bool asc = ((g0 + i) & size) == 0;
                lds_cmpx(sh, i, j, asc);
            }
            __syncthreads();
        }
    }
    for (uint32_t e = threadIdx.x; e < len; e += SORT_NT) stk(base + e, sh[e]);
}

// Stage B: one global compare-exchange pass of the bitonic network (stride >= SORT_CH)
__global__ __launch_bounds__(256) void sort_global_kernel(Fe* keys, uint32_t n, uint32_t size, uint32_t stride) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n / 2) return;
    Fe* base = keys + (size_t)blockIdx.y * n;
    uint32_t i = 2 * t - (t & (stride - 1));
    uint32_t j = i + stride;
    Fe a = ldk(base + i), b = ldk(base + j);
    const bool asc = (i & size) == 0;
    if ((key_cmp(a, b) > 0) == asc) {
        stk(base + i, b);
        stk(base + j, a);
    }
}

// Stage B, all of a level's global strides (size/2 .. SORT_CH) in one launch: the keys that meet through
// those strides differ only in the `nbits` index bits [log2 SORT_CH, log2 size); a workgroup takes all 2^nbits
// values of those bits for a run of LW = SORT_CH >> nbits consecutive low indices (2^nbits coalesced
// segments), so the strides become LDS strides LW << p.
__global__ __launch_bounds__(SORT_NT) void sort_global_fused_kernel(Fe* keys, uint32_t n, uint32_t size, uint32_t nbits) {
    __shared__ Fe sh[SORT_CH];
    const uint32_t lw_len = SORT_CH >> nbits;
    const uint32_t top = blockIdx.x >> nbits, r = blockIdx.x & ((1u << nbits) - 1u);
    Fe* base = keys + (size_t)blockIdx.y * n + (size_t)top * size + (size_t)r * lw_len;
    for (uint32_t e = threadIdx.x; e < SORT_CH; e += SORT_NT) {
        const uint32_t hb = e / lw_len, lw = e % lw_len;
        sh[e] = ldk(base + (size_t)hb * SORT_CH + lw);
    }
    __syncthreads();
    const bool asc = (top & 1u) == 0;  // bit `size` of the global index
    for (uint32_t stride = SORT_CH >> 1; stride >= lw_len; stride >>= 1) {
        for (uint32_t t = threadIdx.x; t < SORT_CH / 2; t += SORT_NT) {
            uint32_t i = 2 * t - (t & (stride - 1));
            lds_cmpx(sh, i, i + stride, asc);
        }
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < SORT_CH; e += SORT_NT) {
        const uint32_t hb = e / lw_len, lw = e % lw_len;
        stk(base + (size_t)hb * SORT_CH + lw, sh[e]);
    }
}

// Stage C: the remaining strides (< SORT_CH) of merge level `size`, chunk-local in LDS
__global__ __launch_bounds__(SORT_NT) void sort_merge_local_kernel(Fe* keys, uint32_t n, uint32_t size) {
    __shared__ Fe sh[SORT_CH];
    Fe* base = keys + (size_t)blockIdx.y * n + (size_t)blockIdx.x * SORT_CH;
    const uint32_t g0 = blockIdx.x * SORT_CH;
    for (uint32_t e = threadIdx.x; e < SORT_CH; e += SORT_NT) sh[e] = ldk(base + e);
    __syncthreads();
    const bool asc = (g0 & size) == 0;  // the whole chunk lies in one half of the size-`size` pattern
    for (uint32_t stride = SORT_CH >> 1; stride > 0; stride >>= 1) {
        for (uint32_t t = threadIdx.x; t < SORT_CH / 2; t += SORT_NT) {
            uint32_t i = 2 * t - (t & (stride - 1));
            lds_cmpx(sh, i, i + stride, asc);
        }
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < SORT_CH; e += SORT_NT) stk(base + e, sh[e]);
}

// keys: [batch][n], n a power of two; ascending by canonical value, in place
int poly_sort_keys(zg_ctx* ctx, Fe* keys, uint32_t n, uint32_t batch) {
    if (!batch || n < 2) return ZG_OK;
    ZG_REQUIRE((n & (n - 1)) == 0, ZG_ERR_INVALID_ARG, "sort: n=%u is not a power of two", n);
    const uint32_t chunks = n <= SORT_CH ? 1 : n / SORT_CH;
    const double bytes = (double)batch * n * 64;
    ZG_LAUNCH(ctx, "sort_local", bytes, sort_local_kernel, dim3(chunks, batch), dim3(SORT_NT), 0, keys, n);
    for (uint32_t size = 2 * SORT_CH; size <= n; size <<= 1) {
        uint32_t nbits = 0;
        while ((SORT_CH << nbits) < size) nbits++;
        if (nbits <= 10) {  // (SORT_CH >> nbits >= 1)
            ZG_LAUNCH(ctx, "sort_global", bytes, sort_global_fused_kernel, dim3(chunks, batch), dim3(SORT_NT), 0, keys, n, size,
                      nbits);
        } else {
            for (uint32_t stride = size >> 1; stride >= SORT_CH; stride >>= 1)
                ZG_LAUNCH(ctx, "sort_global", bytes, sort_global_kernel, dim3((n / 2 + 255) / 256, batch), dim3(256), 0, keys, n,
                          size, stride);
        }
        ZG_LAUNCH(ctx, "sort_merge_local", bytes, sort_merge_local_kernel, dim3(chunks, batch), dim3(SORT_NT), 0, keys, n,
                  size);
    }
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// ------------------------------------------------------------------ building s'
// first-occurrence flags of a', and one consumed table entry per distinct a' value
__global__ __launch_bounds__(256) void pp_flags_kernel(const Fe* __restrict__ a, const Fe* __restrict__ t, uint32_t n,
                                                       uint32_t usable, uint32_t* __restrict__ repeated,
                                                       uint32_t* __restrict__ consumed, uint32_t* __restrict__ err) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= usable) return;
    const Fe* ab = a + (size_t)b * n;
    const Fe* tb = t + (size_t)b * n;
    Fe v = ldk(ab + i);
    bool first = i == 0 || key_cmp(ldk(ab + i - 1), v) != 0;
    repeated[(size_t)b * n + i] = first ? 0u : 1u;
    if (!first) return;
    // lower_bound of v in the sorted table: the first element of v's run
    uint32_t lo = 0, hi = usable;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (key_cmp(ldk(tb + mid), v) < 0) lo = mid + 1;
        else hi = mid;
    }
    if (lo == usable || key_cmp(ldk(tb + lo), v) != 0) {
        atomicOr(err + b, 1u);  // input value missing from the table: ConstraintSystemFailure
        return;
    }
    consumed[(size_t)b * n + lo] = 1u;
}

// Exclusive prefix sums of `repeated` and of `!consumed` over the usable rows, in two levels (one workgroup per vector
// walking 2^17 rows in per-lane strips took 0.37 ms at k = 17 -- uncoalesced, dependent loads; round 3): every tile of 1024
// rows is scanned by a workgroup of its own (coalesced in, coalesced out: the flag in the top bit, the TILE-LOCAL
// exclusive prefix below it), a second kernel scans the tile totals per vector, and the two readers below add their
// tile's offset.
constexpr uint32_t PP_TILE = 1024;

__global__ __launch_bounds__(PP_TILE) void pp_scan_local_kernel(uint32_t* __restrict__ repeated, uint32_t* __restrict__ consumed,
                                                                uint32_t n, uint32_t usable, uint32_t ntile,
                                                                uint32_t* __restrict__ toff) {
    __shared__ uint32_t sr[PP_TILE], sl[PP_TILE];
    const uint32_t tid = threadIdx.x, tile = blockIdx.x, b = blockIdx.y;
    const uint32_t i = tile * PP_TILE + tid;
    uint32_t* rp = repeated + (size_t)b * n;
    uint32_t* cp = consumed + (size_t)b * n;
    const uint32_t fr = i < usable ? rp[i] : 0u, fl = i < usable ? 1u - cp[i] : 0u;
    sr[tid] = fr;
    sl[tid] = fl;
    __syncthreads();
    for (uint32_t o = 1; o < PP_TILE; o <<= 1) {
        uint32_t vr = 0, vl = 0;
        if (tid >= o) {
            vr = sr[tid - o];
            vl = sl[tid - o];
        }
        __syncthreads();
        sr[tid] += vr;
        sl[tid] += vl;
        __syncthreads();
    }
    if (i < usable) {
        rp[i] = (sr[tid] - fr) | (fr << 31);
        cp[i] = (sl[tid] - fl) | (fl << 31);
    }
    if (tid == PP_TILE - 1) {
        toff[((size_t)b * ntile + tile) * 2] = sr[tid];
        toff[((size_t)b * ntile + tile) * 2 + 1] = sl[tid];
    }
}

// tile totals -> exclusive tile offsets (in place) and the vector's totals; ntile <= 1024
__global__ __launch_bounds__(PP_TILE) void pp_scan_tiles_kernel(uint32_t* __restrict__ toff, uint32_t ntile, uint32_t* __restrict__ totals) {
    __shared__ uint32_t sr[PP_TILE], sl[PP_TILE];
    const uint32_t tid = threadIdx.x, b = blockIdx.x;
    uint32_t* tb = toff + (size_t)b * ntile * 2;
    const uint32_t r = tid < ntile ? tb[2 * tid] : 0u, l = tid < ntile ? tb[2 * tid + 1] : 0u;
    sr[tid] = r;
    sl[tid] = l;
    __syncthreads();
    for (uint32_t o = 1; o < PP_TILE; o <<= 1) {
        uint32_t vr = 0, vl = 0;
        if (tid >= o) {
            vr = sr[tid - o];
            vl = sl[tid - o];
        }
        __syncthreads();
        sr[tid] += vr;
        sl[tid] += vl;
        __syncthreads();
    }
    if (tid < ntile) {
        tb[2 * tid] = sr[tid] - r;
        tb[2 * tid + 1] = sl[tid] - l;
    }
    if (tid == PP_TILE - 1) {
        totals[2 * b] = sr[tid];
        totals[2 * b + 1] = sl[tid];
    }
}

// left-over table entries, ascending
__global__ __launch_bounds__(256) void pp_leftover_kernel(const Fe* __restrict__ t, const uint32_t* __restrict__ consumed,
                                                          uint32_t n, uint32_t usable, Fe* __restrict__ leftover,
                                                          const uint32_t* __restrict__ toff, uint32_t ntile) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (j >= usable) return;
    uint32_t c = consumed[(size_t)b * n + j];
    if (c >> 31) {
        const uint32_t at = (c & 0x7fffffffu) + toff[((size_t)b * ntile + j / PP_TILE) * 2 + 1];
        stk(leftover + (size_t)b * n + at, ldk(t + (size_t)b * n + j));
    }
}

__global__ __launch_bounds__(256) void pp_build_kernel(const Fe* __restrict__ a, const uint32_t* __restrict__ repeated,
                                                       const Fe* __restrict__ leftover, const uint32_t* __restrict__ totals,
                                                       uint32_t n, uint32_t usable, Fe* __restrict__ sprime,
                                                       uint32_t* __restrict__ err, const uint32_t* __restrict__ toff, uint32_t ntile) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= usable) return;
    const uint32_t m_rep = totals[2 * b], m_left = totals[2 * b + 1];
    if (m_rep != m_left) {
        if (i == 0) atomicOr(err + b, 2u);
        return;
    }
    uint32_t r = repeated[(size_t)b * n + i];
    Fe v;
    if (r >> 31) v = ldk(leftover + (size_t)b * n + (m_rep - 1 - ((r & 0x7fffffffu) + toff[((size_t)b * ntile + i / PP_TILE) * 2])));
    else v = ldk(a + (size_t)b * n + i);
    stk(sprime + (size_t)b * n + i, v);
}

// a: [batch][n] sorted inputs, t: [batch][n] sorted tables (sentinel-padded beyond `usable`);
// sprime: [batch][n] receives s' on rows < usable.  scratch_u32: 2*batch*n + 2*batch words;
// scratch_fe: batch*n elements.  d_err: batch words, zeroed here (in the same fill as the scratch when it
// is scratch_u32 + 2*batch*n + 2*batch); bit 0 = input missing from table.
int poly_permute_pairs(zg_ctx* ctx, const Fe* a, const Fe* t, Fe* sprime, uint32_t n, uint32_t usable, uint32_t batch,
                       uint32_t* scratch_u32, Fe* scratch_fe, uint32_t* d_err) {
    if (!batch) return ZG_OK;
    uint32_t* repeated = scratch_u32;
    uint32_t* consumed = repeated + (size_t)batch * n;
    uint32_t* totals = consumed + (size_t)batch * n;
    if (d_err == totals + 2 * batch) {  // the caller keeps the error words behind the scratch: one fill for all of it
        ZG_HIP(hipMemsetAsync(consumed, 0, ((size_t)batch * n + 3 * batch) * 4, ctx->stream));
    } else {
        ZG_HIP(hipMemsetAsync(consumed, 0, (size_t)batch * n * 4, ctx->stream));
        ZG_HIP(hipMemsetAsync(d_err, 0, (size_t)batch * 4, ctx->stream));
    }
    dim3 g((usable + 255) / 256, batch);
    // what each kernel streams per row AT LEAST (round 3 charged all five n * 96: the scans touch 4-byte flags only; the
    // leftover list is written / read for the unconsumed rows only and is left out of the charge)
    const double rows = (double)batch * usable;
    ZG_LAUNCH(ctx, "permute_flags", rows * 72, pp_flags_kernel, g, dim3(256), 0, a, t, n, usable, repeated, consumed, d_err);
    const uint32_t ntile = (usable + PP_TILE - 1) / PP_TILE;
    ZG_REQUIRE(ntile <= PP_TILE, ZG_ERR_UNSUPPORTED, "permute_pairs: %u rows", usable);
    WsScope ws(ctx);
    uint32_t* toff = ws.get<uint32_t>((size_t)2 * batch * ntile);
    if (!toff) return ZG_ERR_OOM;
    ZG_LAUNCH(ctx, "permute_scan", rows * 8 + (double)batch * ntile * 8, pp_scan_local_kernel, dim3(ntile, batch), dim3(PP_TILE), 0, repeated, consumed, n, usable, ntile, toff);
    ZG_LAUNCH(ctx, "permute_scan", (double)batch * ntile * 16, pp_scan_tiles_kernel, dim3(batch), dim3(PP_TILE), 0, toff, ntile, totals);
    ZG_LAUNCH(ctx, "permute_leftover", rows * 36, pp_leftover_kernel, g, dim3(256), 0, t, consumed, n, usable, scratch_fe, toff, ntile);
    ZG_LAUNCH(ctx, "permute_build", rows * 68, pp_build_kernel, g, dim3(256), 0, a, repeated, scratch_fe, totals, n, usable, sprime,
              d_err, toff, ntile);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int poly_sort_pad(zg_ctx* ctx, Fe* keys, uint32_t n, uint32_t usable, uint32_t batch) {
    if (!batch || usable >= n) return ZG_OK;
    ZG_LAUNCH(ctx, "sort_pad", 0, sort_pad_kernel, dim3((n - usable + 63) / 64, batch), dim3(64), 0, keys, n, usable);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

}  // namespace zg
