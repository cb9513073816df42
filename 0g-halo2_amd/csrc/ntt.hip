// Radix-2 NTT / iNTT over BN254 Fr for gfx950 -- replaces halo2_proofs::arithmetic::best_fft and the
// EvaluationDomain wrappers lagrange_to_coeff / coeff_to_extended / extended_to_coeff
// (halo2_proofs v2023_04_20 src/arithmetic.rs, src/poly/domain.rs; reached from create_proof,
// reference call site /root/reference/src/wnn.rs:242-259).
//
// Structure (MI355X-first, not halo2's recursive butterfly):
//   N = N1 * N2.  Pass 1 ("cols"): each workgroup stages an [N1][C] tile of C adjacent columns in
//   LDS, runs the size-N1 DIF butterflies there, multiplies by the inter-pass twiddle
//   omega^(j2*k1) and writes rows of C contiguous elements.  Pass 2 ("rows"): each workgroup stages
//   R contiguous rows of N2, runs the size-N2 butterflies in LDS and writes X[k1 + N1*k2] so that R
//   neighbouring k1 form one contiguous segment.  Natural order in, natural order out, the
//   bit-reversal is absorbed into the LDS read index of the store loop.  Twiddles omega^i come
//   from one HBM table per (log_n, omega) (coalesced 32-B loads, L2 resident), the per-tile
//   sub-transform twiddles are staged in LDS.  Zero padding, the zeta^(i mod 3) coset scaling, the
//   ifft divisor and the truncation of extended_to_coeff are fused into the first load / last
//   store so a polynomial crosses HBM exactly twice per transform.
// The kernel is integer-ALU bound (one 254-bit Montgomery product per butterfly), not MFMA work.
//
// A nine-limb back end (field9.h; 36-byte lazily reduced elements in LDS, 18 % fewer instructions per butterfly) was
// built, bit-exact, and measured on par at best (1.56 vs 1.56 ms/proof; 1.75 vs 1.65 with 4-byte LDS accesses): the
// transform is bound by its LDS round trips and barriers before its VALU work.  It was removed in round 3 with its
// ZG_NTT9 switch (git history has it); the twiddle tables keep their second half, omega^i * 2^5, which evaluate_h reads.
#include "poly.h"
#include "field9.h"

namespace zg {

struct NttArgs {
    const Fe* in;
    Fe* out;
    const Fe* tw;       // omega^i, i < N; entries [N, 2N) hold omega^i * 2^5 (the 2^261 Montgomery form)
    const Fe* tw_m;     // (nine-limb pass) the same table for the pass's sub-transform: omega_M^t, t < M, then omega_M^t * 2^5
    uint32_t tw_shift;  // (nine-limb pass) LDS holds the sub-transform twiddles whose index is a multiple of 2^tw_shift; the
                        // stages before that read theirs from tw_m (contiguous, L1-resident)
    size_t in_stride;   // elements between consecutive batch arrays
    size_t out_stride;
    // groups (poly.h Grouping): array v of the batch sits at (v / per) * outer + (v % per) * stride; a flat side has
    // per = 0xffffffff (v / per = 0)
    uint32_t in_per, out_per;
    size_t in_outer, out_outer;
    uint32_t log_n, log_n1, log_n2;
    uint32_t in_len;    // FIRST pass: input entries beyond in_len read as zero
    uint32_t out_len;   // LAST pass: entries >= out_len are not written
    uint32_t coset_in;  // FIRST pass: 1 = multiply entry j by zin[j % 3] for j % 3 != 0; 2 = all three (zin0 too)
    uint32_t coset_out; // LAST pass: multiply entry k by zout[k % 3]
    uint32_t scale_out; // LAST pass: multiply by `scale`
    Fe zin0, zin1, zin2, zout1, zout2, scale;
};

__device__ __forceinline__ uint32_t bitrev(uint32_t v, uint32_t bits) {
    return bits == 0 ? 0u : (__brev(v) >> (32 - bits));
}

__device__ __forceinline__ Fe ld_fe(const Fe* p) {
    Fe r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}

__device__ __forceinline__ void st_fe(Fe* p, const Fe& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// One pass over LDS tiles of T = 2^LOG_T elements with T/4 threads.
//   COLS : tile = [M = N1][cnt = T/N1 columns], sub-transform along the column (stride cnt)
//   !COLS: tile = [cnt = min(T/N2, N1) rows][M = N2], sub-transform along the row
//   FIRST: this pass reads the caller's input (zero padding / coset scaling apply)
// The rows pass is always the last one (output scaling / truncation apply there).
template <int LOG_T, bool COLS, bool FIRST>
__global__ __launch_bounds__(1 << (LOG_T - 2)) void ntt_pass_kernel(NttArgs a) {
    constexpr uint32_t T = 1u << LOG_T;
    constexpr uint32_t NT = T / 4;
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS layout: an element is two 16-byte halves in two separate arrays (a 32-byte stride puts lanes i
    // and i+4 of a b128 access on the same banks), and its slot is its index with the higher 3-bit groups
    // XOR-folded into the low three bits -- consecutive indices and every power-of-two stride (late
    // butterfly stages, the bit-reversed reads of the store loop, twiddle strides) then fall on eight
    // different bank quads.  Measured before: a third of the kernel's busy cycles were LDS bank conflicts.
    uint4* XL = reinterpret_cast<uint4*>(smem);
    uint4* XH = XL + T;
    uint4* TL = XH + T;
    uint4* TH = TL + (1u << ((COLS ? a.log_n1 : a.log_n2) > 0 ? (COLS ? a.log_n1 : a.log_n2) - 1 : 0));
    auto slot = [](uint32_t i) { return i ^ ((i >> 3) & 7u) ^ ((i >> 6) & 7u) ^ ((i >> 9) & 7u); };
    auto ldx = [&](uint32_t i) {
        const uint32_t s = slot(i);
        const uint4 lo = XL[s], hi = XH[s];
        return Fe{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    };
    auto stx = [&](uint32_t i, const Fe& v) {
        const uint32_t s = slot(i);
        XL[s] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
        XH[s] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    };
    auto ldt = [&](uint32_t i) {
        const uint32_t s = slot(i);
        const uint4 lo = TL[s], hi = TH[s];
        return Fe{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    };

    const uint32_t tid = threadIdx.x;
    const uint32_t log_m = COLS ? a.log_n1 : a.log_n2;
    const uint32_t M = 1u << log_m;
    const uint32_t N1 = 1u << a.log_n1, N2 = 1u << a.log_n2;
    uint32_t log_cnt = LOG_T - log_m;
    if (!COLS && log_cnt > a.log_n1) log_cnt = a.log_n1;
    const uint32_t cnt = 1u << log_cnt;
    const uint32_t tile = M << log_cnt;  // elements actually used (<= T)

    // XCD-aware remap: consecutive tiles (which share 128-B lines when cnt*32 B < 128 B) go to
    // the same XCD, i.e. the same L2, under the round-robin workgroup dispatch.
    uint32_t nblk = gridDim.x, bid = blockIdx.x;
    if ((nblk & 7u) == 0) bid = (bid & 7u) * (nblk >> 3) + (bid >> 3);
    const uint32_t base = bid << log_cnt;  // first column (COLS) or first row (!COLS)

    const Fe* in = a.in + (size_t)(blockIdx.y / a.in_per) * a.in_outer + (size_t)(blockIdx.y % a.in_per) * a.in_stride;
    Fe* out = a.out + (size_t)(blockIdx.y / a.out_per) * a.out_outer + (size_t)(blockIdx.y % a.out_per) * a.out_stride;

    // stage the sub-transform twiddles omega_M^t = omega^(t * N/M)
    {
        const uint32_t tw_shift = a.log_n - log_m;
        for (uint32_t t = tid; t < M / 2; t += NT) {
            const Fe w = ld_fe(a.tw + ((size_t)t << tw_shift));
            const uint32_t s = slot(t);
            TL[s] = make_uint4(w.l[0], w.l[1], w.l[2], w.l[3]);
            TH[s] = make_uint4(w.l[4], w.l[5], w.l[6], w.l[7]);
        }
    }

    // load the tile
    for (uint32_t e = tid; e < tile; e += NT) {
        uint32_t g, li;
        if (COLS) {
            uint32_t c = e & (cnt - 1), j1 = e >> log_cnt;
            g = j1 * N2 + base + c;
            li = e;  // [j1][c]
        } else {
            uint32_t j2 = e & (M - 1), r = e >> log_m;
            g = (base + r) * N2 + j2;
            li = e;  // [r][j2]
        }
        Fe v;
        if (FIRST) {
            if (g < a.in_len) {
                v = ld_fe(in + g);
                if (a.coset_in) {
                    uint32_t m3 = g % 3u;
                    if (m3 == 1) v = Fr::mul(v, a.zin1);
                    else if (m3 == 2) v = Fr::mul(v, a.zin2);
                    else if (a.coset_in == 2) v = Fr::mul(v, a.zin0);
                }
            } else {
                v = fe_zero();
            }
        } else {
            v = ld_fe(in + g);
        }
        stx(li, v);
    }
    __syncthreads();

    // DIF butterflies: natural order in, bit-reversed order out.  Two stages at a time in registers (a
    // radix-2^2 group of four elements per lane: the same four products, half the LDS round trips and
    // barriers), one radix-2 stage at the end when the count is odd.
    auto at = [&](uint32_t sub, uint32_t e) { return COLS ? (e << log_cnt) + sub : (sub << log_m) + e; };
    uint32_t st = 0;
    // Zero padding (coeff_to_extended: n coefficients in an 8n-point transform): rows j1 >= nz_rows of every
    // column are zero.  While they start within the first quarter, the first radix-2^2 group of a column has
    // one non-zero input at most -- three products instead of four where there is one, none where there is none.
    if (FIRST && COLS && log_m >= 2) {
        const uint32_t nz_rows = (a.in_len + N2 - 1) >> a.log_n2;
        const uint32_t quarter = M >> 2, half = M >> 1;
        if (nz_rows <= quarter) {
            for (uint32_t g = tid; g < tile / 4; g += NT) {
                const uint32_t sub = g & (cnt - 1), i = g >> log_cnt;  // (one block at this stage: e0 = i)
                if (i >= nz_rows) continue;  // four zeros in, four zeros out: the tile already holds them
                const uint32_t i0 = at(sub, i), i1 = at(sub, i + quarter), i2 = at(sub, i + half), i3 = at(sub, i + half + quarter);
                const Fe x0 = ldx(i0);
                Fe a2 = x0;
                if (i != 0) a2 = Fr::mul(x0, ldt(i));
                stx(i2, a2);
                if (i != 0) {
                    const Fe w = ldt(i << 1);
                    stx(i1, Fr::mul(x0, w));
                    stx(i3, Fr::mul(a2, w));
                } else {
                    stx(i1, x0);
                    stx(i3, a2);
                }
            }
            __syncthreads();
            st = 2;
        }
    }
    for (; st + 1 < log_m; st += 2) {
        const uint32_t log_half = log_m - st - 1, log_q = log_half - 1;
        const uint32_t half = 1u << log_half, quarter = 1u << log_q;
        for (uint32_t g = tid; g < tile / 4; g += NT) {
            uint32_t sub, b;
            if (COLS) {
                sub = g & (cnt - 1);
                b = g >> log_cnt;
            } else {
                b = g & (M / 4 - 1);
                sub = g >> (log_m - 2);
            }
            const uint32_t blk = b >> log_q, i = b & (quarter - 1);
            const uint32_t e0 = (blk << (log_half + 1)) + i;
            const uint32_t i0 = at(sub, e0), i1 = at(sub, e0 + quarter), i2 = at(sub, e0 + half), i3 = at(sub, e0 + half + quarter);
            const Fe x0 = ldx(i0), x1 = ldx(i1), x2 = ldx(i2), x3 = ldx(i3);
            // stage st: (x0, x2) and (x1, x3), twiddles omega_M^(j << st) for j = i and i + quarter
            const Fe a0 = Fr::add(x0, x2), a1 = Fr::add(x1, x3);
            Fe a2 = Fr::sub(x0, x2), a3 = Fr::sub(x1, x3);
            const uint32_t t0 = i << st;
            if (t0 != 0) a2 = Fr::mul(a2, ldt(t0));
            a3 = Fr::mul(a3, ldt((i + quarter) << st));
            // stage st + 1: (a0, a1) and (a2, a3), twiddle omega_M^(i << (st + 1)) for both
            const uint32_t t1 = i << (st + 1);
            stx(i0, Fr::add(a0, a1));
            stx(i2, Fr::add(a2, a3));
            Fe b1 = Fr::sub(a0, a1), b3 = Fr::sub(a2, a3);
            if (t1 != 0) {
                const Fe w = ldt(t1);
                b1 = Fr::mul(b1, w);
                b3 = Fr::mul(b3, w);
            }
            stx(i1, b1);
            stx(i3, b3);
        }
        __syncthreads();
    }
    const uint32_t nbf = tile / 2;
    for (; st < log_m; st++) {
        const uint32_t log_half = log_m - st - 1;
        const uint32_t half = 1u << log_half;
        for (uint32_t bf = tid; bf < nbf; bf += NT) {
            uint32_t s, b;
            if (COLS) {
                s = bf & (cnt - 1);
                b = bf >> log_cnt;
            } else {
                b = bf & (M / 2 - 1);
                s = bf >> (log_m - 1);
            }
            uint32_t blk = b >> log_half, i = b & (half - 1);
            uint32_t lo = (blk << (log_half + 1)) + i, hi = lo + half;
            uint32_t ilo = at(s, lo), ihi = at(s, hi);
            Fe u = ldx(ilo), v = ldx(ihi);
            stx(ilo, Fr::add(u, v));
            Fe d = Fr::sub(u, v);
            uint32_t twi = i << st;
            if (twi != 0) d = Fr::mul(d, ldt(twi));
            stx(ihi, d);
        }
        __syncthreads();
    }

    // store
    if (COLS) {
        for (uint32_t e = tid; e < tile; e += NT) {
            uint32_t c = e & (cnt - 1), pos = e >> log_cnt;
            uint32_t k1 = bitrev(pos, log_m);
            uint32_t j2 = base + c;
            Fe v = ldx(e);
            uint32_t ti = j2 * k1;  // < N
            if (ti != 0) v = Fr::mul(v, ld_fe(a.tw + ti));
            st_fe(out + (size_t)k1 * N2 + j2, v);
        }
    } else {
        for (uint32_t e = tid; e < tile; e += NT) {
            uint32_t r = e & (cnt - 1), k2 = e >> log_cnt;
            uint32_t pos = bitrev(k2, log_m);
            uint32_t k = (base + r) + N1 * k2;
            if (k >= a.out_len) continue;
            Fe v = ldx((r << log_m) + pos);
            if (a.scale_out) v = Fr::mul(v, a.scale);
            if (a.coset_out) {
                uint32_t m3 = k % 3u;
                if (m3 == 1) v = Fr::mul(v, a.zout1);
                else if (m3 == 2) v = Fr::mul(v, a.zout2);
            }
            st_fe(out + k, v);
        }
    }
}

// ---- the same pass on nine 29-bit limbs (field9.h; ZG_NTT9: the default of the latency form).  Tiling, indices, the XOR-folded LDS slots and
// the fused first-load / last-store work are ntt_pass_kernel's; what differs is the arithmetic:
//   * an element is 36 bytes in LDS (limbs 0..3 and 4..7 as two 16-byte words, limb 8 in a third array);
//   * the data stay in the library's x * 2^256 form and every twiddle comes from the table's second half
//     (omega^i * 2^5, i.e. the 2^261 form as an integer): Fr9::mul(x * 2^256, w * 2^261) = x w * 2^256 -- no conversion;
//   * sums and differences are limb-wise, a sum is carried back to 29-bit limbs (f9_norm) before it is stored or subtracted,
//     the difference of two normalised values enters its product as it is, and EVERY twiddle is multiplied (omega^0 too:
//     the branch of ntt_pass_kernel saved nothing -- a wave takes it for the one lane that has i = 0);
//   * bounds.  A product by a canonical twiddle of an operand below 2^261 lies in (-p, 2p).  An element that is only ever
//     SUMMED quadruples per radix-2^2 group: from 2p to 8p, 32p, 128p -- under the 2^261 = 169 p that Fr9::mul's top limb
//     allows for the differences taken from it only while it is <= 84 p.  So the sum of every THIRD group is multiplied
//     by one (2^261 mod r: a Montgomery reduction back into (-p, 2p)) -- one product in twelve on top.  Where every lane's
//     twiddle is omega^0 (the last group of an even stage count: three of its four; the last stage of an odd count) nothing
//     is multiplied, as in ntt_pass_kernel, and the results leave at most 4 x (2 x) their inputs' size: with the reduction
//     in the third group that is 64 p after ten stages and 128 p after eleven, under what the pass's closing product or
//     f9_reduce_pack (2^263) take;
//   * what leaves a pass is canonical (f9_reduce_pack: magnitude < 2^263), so the bytes are ntt_pass_kernel's.
// 4 x 232 + 3 x 24 + 8 x 9 instructions per group of four against 4 x 340 + 8 x 28.
template <int LOG_T, bool COLS, bool FIRST>
__global__ __launch_bounds__(1 << (LOG_T - 2)) void ntt9_pass_kernel(NttArgs a) {
    constexpr uint32_t T = 1u << LOG_T;
    constexpr uint32_t NT = T / 4;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t log_m = COLS ? a.log_n1 : a.log_n2;
    const uint32_t M = 1u << log_m;
    const uint32_t S = a.tw_shift;
    const uint32_t TWN = (M / 2) >> S ? (M / 2) >> S : 1;
    uint4* XL = reinterpret_cast<uint4*>(smem);
    uint4* XH = XL + T;
    uint4* TL = XH + T;
    uint4* TH = TL + TWN;
    int32_t* X8 = reinterpret_cast<int32_t*>(TH + TWN);
    int32_t* T8 = X8 + T;
    auto slot = [](uint32_t i) { return i ^ ((i >> 3) & 7u) ^ ((i >> 6) & 7u) ^ ((i >> 9) & 7u); };
    auto ld3 = [&](const uint4* L, const uint4* H, const int32_t* E, uint32_t i) {
        const uint32_t s = slot(i);
        const uint4 lo = L[s], hi = H[s];
        F9 r;
        r.l[0] = (int32_t)lo.x; r.l[1] = (int32_t)lo.y; r.l[2] = (int32_t)lo.z; r.l[3] = (int32_t)lo.w;
        r.l[4] = (int32_t)hi.x; r.l[5] = (int32_t)hi.y; r.l[6] = (int32_t)hi.z; r.l[7] = (int32_t)hi.w;
        r.l[8] = E[s];
        return r;
    };
    auto st3 = [&](uint4* L, uint4* H, int32_t* E, uint32_t i, const F9& v) {
        const uint32_t s = slot(i);
        L[s] = make_uint4((uint32_t)v.l[0], (uint32_t)v.l[1], (uint32_t)v.l[2], (uint32_t)v.l[3]);
        H[s] = make_uint4((uint32_t)v.l[4], (uint32_t)v.l[5], (uint32_t)v.l[6], (uint32_t)v.l[7]);
        E[s] = v.l[8];
    };
    auto ldx = [&](uint32_t i) { return ld3(XL, XH, X8, i); };
    auto stx = [&](uint32_t i, const F9& v) { st3(XL, XH, X8, i, v); };
    const Fe* twm9 = a.tw_m + M;
    // twiddle omega_M^i of stage st (i is a multiple of 2^st; st is uniform)
    auto ldt = [&](uint32_t i, uint32_t st) { return st >= S ? ld3(TL, TH, T8, i >> S) : f9_unpack(ld_fe(twm9 + i)); };

    const uint32_t tid = threadIdx.x;
    const uint32_t N1 = 1u << a.log_n1, N2 = 1u << a.log_n2;
    uint32_t log_cnt = LOG_T - log_m;
    if (!COLS && log_cnt > a.log_n1) log_cnt = a.log_n1;
    const uint32_t cnt = 1u << log_cnt;
    const uint32_t tile = M << log_cnt;

    uint32_t nblk = gridDim.x, bid = blockIdx.x;
    if ((nblk & 7u) == 0) bid = (bid & 7u) * (nblk >> 3) + (bid >> 3);
    const uint32_t base = bid << log_cnt;

    const Fe* in = a.in + (size_t)(blockIdx.y / a.in_per) * a.in_outer + (size_t)(blockIdx.y % a.in_per) * a.in_stride;
    Fe* out = a.out + (size_t)(blockIdx.y / a.out_per) * a.out_outer + (size_t)(blockIdx.y % a.out_per) * a.out_stride;
    const Fe* tw9 = a.tw + ((size_t)1 << a.log_n);  // omega^i * 2^5

    for (uint32_t t = tid; t < (M / 2) >> S; t += NT) st3(TL, TH, T8, t, f9_unpack(ld_fe(twm9 + (t << S))));
    for (uint32_t e = tid; e < tile; e += NT) {
        uint32_t g;
        if (COLS) {
            uint32_t c = e & (cnt - 1), j1 = e >> log_cnt;
            g = j1 * N2 + base + c;
        } else {
            uint32_t j2 = e & (M - 1), r = e >> log_m;
            g = (base + r) * N2 + j2;
        }
        F9 v;
        if (FIRST) {
            if (g < a.in_len) {
                v = f9_unpack(ld_fe(in + g));
                if (a.coset_in) {  // (zin*, zout*, scale arrive in the 2^261 form on this path: launch_passes)
                    uint32_t m3 = g % 3u;
                    if (m3 == 1) v = Fr9::mul(v, f9_unpack(a.zin1));
                    else if (m3 == 2) v = Fr9::mul(v, f9_unpack(a.zin2));
                    else if (a.coset_in == 2) v = Fr9::mul(v, f9_unpack(a.zin0));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 9; i++) v.l[i] = 0;
            }
        } else {
            v = f9_unpack(ld_fe(in + g));
        }
        stx(e, v);
    }
    __syncthreads();

    auto at = [&](uint32_t sub, uint32_t e) { return COLS ? (e << log_cnt) + sub : (sub << log_m) + e; };
    uint32_t st = 0, grp = 0;  // grp: radix-2^2 groups an only-summed element has been through since its last reduction
    if (FIRST && COLS && log_m >= 2) {  // (zero padding: ntt_pass_kernel)
        const uint32_t nz_rows = (a.in_len + N2 - 1) >> a.log_n2;
        const uint32_t quarter = M >> 2, half = M >> 1;
        if (nz_rows <= quarter) {
            for (uint32_t g = tid; g < tile / 4; g += NT) {
                const uint32_t sub = g & (cnt - 1), i = g >> log_cnt;
                if (i >= nz_rows) continue;
                const uint32_t i0 = at(sub, i), i1 = at(sub, i + quarter), i2 = at(sub, i + half), i3 = at(sub, i + half + quarter);
                const F9 x0 = ldx(i0);
                const F9 a2 = Fr9::mul(x0, ldt(i, 0));
                const F9 w = ldt(i << 1, 1);
                stx(i2, a2);
                stx(i1, Fr9::mul(x0, w));
                stx(i3, Fr9::mul(a2, w));
            }
            __syncthreads();
            st = 2;  // (x0 itself stays where it is: nothing was summed)
        }
    }
    for (; st + 1 < log_m; st += 2) {
        const uint32_t log_half = log_m - st - 1, log_q = log_half - 1;
        const uint32_t half = 1u << log_half, quarter = 1u << log_q;
        const bool reduce = ++grp == 3;  // (uniform)
        if (reduce) grp = 0;
        const bool last = quarter == 1;  // (uniform) i = 0 for every lane: three of the four twiddles are omega^0
        for (uint32_t g = tid; g < tile / 4; g += NT) {
            uint32_t sub, b;
            if (COLS) {
                sub = g & (cnt - 1);
                b = g >> log_cnt;
            } else {
                b = g & (M / 4 - 1);
                sub = g >> (log_m - 2);
            }
            const uint32_t blk = b >> log_q, i = b & (quarter - 1);
            const uint32_t e0 = (blk << (log_half + 1)) + i;
            const uint32_t i0 = at(sub, e0), i1 = at(sub, e0 + quarter), i2 = at(sub, e0 + half), i3 = at(sub, e0 + half + quarter);
            const F9 x0 = ldx(i0), x1 = ldx(i1), x2 = ldx(i2), x3 = ldx(i3);
            const F9 a0 = f9_add(x0, x2), a1 = f9_add(x1, x3);
            const F9 a3 = Fr9::mul(f9_sub(x1, x3), ldt((i + quarter) << st, st));
            if (last) {  // the sub-transform's last two stages: its results leave at most four times their inputs' size
                const F9 d0 = f9_sub(x0, x2);
                stx(i0, f9_norm(f9_add(a0, a1)));
                stx(i1, f9_norm(f9_sub(a0, a1)));
                stx(i2, f9_norm(f9_add(d0, a3)));
                stx(i3, f9_norm(f9_sub(d0, a3)));
                continue;
            }
            const F9 a2 = Fr9::mul(f9_sub(x0, x2), ldt(i << st, st));
            F9 s0 = f9_norm(f9_add(a0, a1));
            if (reduce) s0 = Fr9::mul(s0, Fr9Params::one());
            stx(i0, s0);
            stx(i2, f9_norm(f9_add(a2, a3)));
            const F9 w = ldt(i << (st + 1), st + 1);
            stx(i1, Fr9::mul(f9_norm(f9_sub(a0, a1)), w));
            stx(i3, Fr9::mul(f9_sub(a2, a3), w));
        }
        __syncthreads();
    }
    const uint32_t nbf = tile / 2;
    for (; st < log_m; st++) {  // (the last stage of an odd count: half = 1, every twiddle is omega^0)
        for (uint32_t bf = tid; bf < nbf; bf += NT) {
            uint32_t s, b;
            if (COLS) {
                s = bf & (cnt - 1);
                b = bf >> log_cnt;
            } else {
                b = bf & (M / 2 - 1);
                s = bf >> (log_m - 1);
            }
            const uint32_t ilo = at(s, b << 1), ihi = at(s, (b << 1) + 1);
            const F9 u = ldx(ilo), v = ldx(ihi);
            stx(ilo, f9_norm(f9_add(u, v)));
            stx(ihi, f9_norm(f9_sub(u, v)));
        }
        __syncthreads();
    }

    if (COLS) {
        for (uint32_t e = tid; e < tile; e += NT) {
            uint32_t c = e & (cnt - 1), pos = e >> log_cnt;
            uint32_t k1 = bitrev(pos, log_m);
            uint32_t j2 = base + c;
            const F9 v = Fr9::mul(ldx(e), f9_unpack(ld_fe(tw9 + j2 * k1)));  // (j2 * k1 < N)
            st_fe(out + (size_t)k1 * N2 + j2, f9_reduce_pack<Fr9Params>(v));
        }
    } else {
        for (uint32_t e = tid; e < tile; e += NT) {
            uint32_t r = e & (cnt - 1), k2 = e >> log_cnt;
            uint32_t pos = bitrev(k2, log_m);
            uint32_t k = (base + r) + N1 * k2;
            if (k >= a.out_len) continue;
            F9 v = ldx((r << log_m) + pos);
            if (a.scale_out) v = Fr9::mul(v, f9_unpack(a.scale));
            if (a.coset_out) {
                uint32_t m3 = k % 3u;
                if (m3 == 1) v = Fr9::mul(v, f9_unpack(a.zout1));
                else if (m3 == 2) v = Fr9::mul(v, f9_unpack(a.zout2));
            }
            st_fe(out + k, f9_reduce_pack<Fr9Params>(v));
        }
    }
}

// tw[i] = omega^i, tw[n + i] = omega^i in the 2^261 Montgomery form (nine-limb butterflies), i < n
__global__ void twiddle_kernel(Fe* tw, Fe omega, uint32_t n) {
    constexpr uint32_t CH = 16;
    uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * CH;
    if (i0 >= n) return;
    const Fe c261 = Fr9Params::c261_fe();
    Fe cur = Fr::pow_u64(omega, i0);
    for (uint32_t j = 0; j < CH && i0 + j < n; j++) {
        st_fe(tw + i0 + j, cur);
        st_fe(tw + n + i0 + j, Fr::mul(cur, c261));
        cur = Fr::mul(cur, omega);
    }
}

// One table per (device, log_n, omega), shared by every context of the device.  The table is complete (the
// creating stream is drained) before its pointer is published, so that another context's stream may read it.
int get_twiddles(zg_ctx* ctx, uint32_t log_n, const Fe& omega, Fe** out) {
    TwiddleKey key;
    key.log_n = log_n;
    memcpy(key.omega.data(), &omega, 32);
    DeviceState& ds = device_state(ctx->device);
    std::lock_guard<std::mutex> lock(ds.mu);
    auto it = ds.twiddles.find(key);
    if (it != ds.twiddles.end()) {
        *out = it->second;
        return ZG_OK;
    }
    uint32_t n = 1u << log_n;
    Fe* tw = nullptr;
    gate_yield(ctx);  // (a new table: an allocation and a stream synchronisation)
    ZG_HIP(hipMalloc(&tw, (size_t)2 * n * sizeof(Fe)));
    uint32_t threads = 256, per = 16;
    uint32_t blocks = (n + threads * per - 1) / (threads * per);
    hipLaunchKernelGGL(twiddle_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, tw, omega, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(tw);
        set_error("get_twiddles: %s", hipGetErrorString(e));
        return ZG_ERR_HIP;
    }
    ds.twiddles[key] = tw;
    *out = tw;
    return ZG_OK;
}

struct NttPlan {
    const Fe* in;
    size_t in_stride;
    Fe* out;
    size_t out_stride;
    size_t batch;
    uint32_t log_n;
    Fe omega;
    uint32_t in_len, out_len;
    Grouping grp;           // how the caller's arrays are laid out (the workspace between the passes is flat)
    uint32_t coset_in = 0;  // 0 none, 1 zeta^(j%3) on j%3 != 0, 2 all entries (zin0 too)
    bool coset_out = false, scale_out = false;
    Fe zin0, zin1, zin2, zout1, zout2, scale;
};

// (nine-limb pass) the sub-transform's own twiddle table -- omega_M = omega^(N / M), contiguous -- and how much of it the pass
// keeps in LDS: a tile of T nine-limb elements is 36 T bytes, and what is left of the CU's 160 KB at four workgroups of
// 2^10 elements (two of 2^11) holds 113 (227) twiddles; the stages whose twiddles do not fit read them from the table.
static int sub_twiddles(zg_ctx* ctx, const NttPlan& p, uint32_t log_m, uint32_t log_t, const Fe** table, uint32_t* shift) {
    Fe om = p.omega;
    for (uint32_t i = log_m; i < p.log_n; i++) om = Fr::sqr(om);
    Fe* t = nullptr;
    ZG_TRY(get_twiddles(ctx, log_m, om, &t));
    *table = t;
    const uint32_t room = log_t == 10 ? 113u : 227u;
    uint32_t s = 0;
    while ((((1u << log_m) / 2) >> s) > room) s++;
    *shift = s;
    return ZG_OK;
}
static size_t nine_lds_bytes(uint32_t T, uint32_t log_m, uint32_t shift) {
    const uint32_t twn = ((1u << log_m) / 2) >> shift;
    return (size_t)(T + (twn ? twn : 1)) * 36;
}

template <int LOG_T, bool NINE>
static int launch_passes(zg_ctx* ctx, const NttPlan& p, const Fe* tw, Fe* tmp, size_t tmp_stride) {
    constexpr uint32_t T = 1u << LOG_T;
    constexpr size_t ELEM = NINE ? 36 : sizeof(Fe);
    NttArgs a;
    memset(&a, 0, sizeof(a));
    a.tw = tw;
    a.log_n = p.log_n;
    a.in_len = p.in_len;
    a.out_len = p.out_len;
    a.coset_in = p.coset_in;
    a.coset_out = p.coset_out;
    a.scale_out = p.scale_out;
    a.zin0 = p.zin0; a.zin1 = p.zin1; a.zin2 = p.zin2; a.zout1 = p.zout1; a.zout2 = p.zout2; a.scale = p.scale;
    if (NINE) {  // the nine-limb pass multiplies x * 2^256 by constants in the 2^261 form (c * 2^5 here)
        const Fe c261 = Fr9Params::c261_fe();
        for (Fe* c : {&a.zin0, &a.zin1, &a.zin2, &a.zout1, &a.zout2, &a.scale}) *c = Fr::mul(*c, c261);
    }
    const uint32_t N = 1u << p.log_n;
    const uint32_t gper = p.grp.per ? p.grp.per : 0xffffffffu;
    // algorithmic bytes.  The UNIT is one transform: input entries read + output entries written -- or what
    // the caller says this plan stands for (zg_ctx::unit_next: the split extended domain computes EvaluationDomain's n -> 8n
    // transform as n -> 4n and n -> n), carried by the LAST pass.  Each pass is charged what it streams itself: the first
    // reads the input and writes the N-point workspace, the second reads that and writes the output.
    const double pass_bytes = (double)p.batch * ((double)p.in_len + (double)p.out_len) * 32.0;
    const double unit_bytes = ctx->unit_next >= 0.0 ? ctx->unit_next : pass_bytes;
    ctx->unit_next = -1.0;
    const double pts = (double)p.batch * (double)(1u << p.log_n) * 32.0;
    dim3 block(T / 4);
    const bool single = p.log_n <= (uint32_t)LOG_T;
    if (single) {
        // one workgroup holds the whole transform: single rows pass, in-place safe
        a.log_n1 = 0;
        a.log_n2 = p.log_n;
        a.in = p.in; a.in_stride = p.in_stride;
        a.out = p.out; a.out_stride = p.out_stride;
        a.in_per = a.out_per = gper; a.in_outer = p.grp.in_outer; a.out_outer = p.grp.out_outer;
        size_t lds = (size_t)(T + (N > 1 ? N / 2 : 1)) * ELEM;
        if (NINE) {
            ZG_TRY(sub_twiddles(ctx, p, a.log_n2, LOG_T, &a.tw_m, &a.tw_shift));
            lds = nine_lds_bytes(T, a.log_n2, a.tw_shift);
        }
        if (NINE)
            ZG_LAUNCH_U(ctx, "ntt_single", pass_bytes, unit_bytes, (ntt9_pass_kernel<LOG_T, false, true>), dim3(1, (uint32_t)p.batch), block, lds, a);
        else
            ZG_LAUNCH_U(ctx, "ntt_single", pass_bytes, unit_bytes, (ntt_pass_kernel<LOG_T, false, true>), dim3(1, (uint32_t)p.batch), block, lds, a);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    a.log_n1 = p.log_n / 2;
    a.log_n2 = p.log_n - a.log_n1;
    const uint32_t N1 = 1u << a.log_n1, N2 = 1u << a.log_n2;
    {   // pass 1: in -> tmp
        a.in = p.in; a.in_stride = p.in_stride;
        a.out = tmp; a.out_stride = tmp_stride;
        a.in_per = gper; a.in_outer = p.grp.in_outer;
        a.out_per = 0xffffffffu; a.out_outer = 0;
        uint32_t cnt = T / N1;
        size_t lds = (size_t)(T + (N1 > 1 ? N1 / 2 : 1)) * ELEM;
        if (NINE) {
            ZG_TRY(sub_twiddles(ctx, p, a.log_n1, LOG_T, &a.tw_m, &a.tw_shift));
            lds = nine_lds_bytes(T, a.log_n1, a.tw_shift);
        }
        if (NINE)
            ZG_LAUNCH(ctx, "ntt_cols", (double)p.batch * (double)p.in_len * 32.0 + pts, (ntt9_pass_kernel<LOG_T, true, true>),
                      dim3(N2 / cnt, (uint32_t)p.batch), block, lds, a);
        else
            ZG_LAUNCH(ctx, "ntt_cols", (double)p.batch * (double)p.in_len * 32.0 + pts, (ntt_pass_kernel<LOG_T, true, true>),
                      dim3(N2 / cnt, (uint32_t)p.batch), block, lds, a);
        ZG_HIP(hipGetLastError());
    }
    {   // pass 2: tmp -> out
        a.in = tmp; a.in_stride = tmp_stride;
        a.out = p.out; a.out_stride = p.out_stride;
        a.in_per = 0xffffffffu; a.in_outer = 0;
        a.out_per = gper; a.out_outer = p.grp.out_outer;
        uint32_t cnt = T / N2;
        if (cnt > N1) cnt = N1;
        size_t lds = (size_t)(T + (N2 > 1 ? N2 / 2 : 1)) * ELEM;
        if (NINE) {
            ZG_TRY(sub_twiddles(ctx, p, a.log_n2, LOG_T, &a.tw_m, &a.tw_shift));
            lds = nine_lds_bytes(T, a.log_n2, a.tw_shift);
        }
        if (NINE)
            ZG_LAUNCH_U(ctx, "ntt_rows", pts + (double)p.batch * (double)p.out_len * 32.0, unit_bytes, (ntt9_pass_kernel<LOG_T, false, false>),
                        dim3(N1 / cnt, (uint32_t)p.batch), block, lds, a);
        else
            ZG_LAUNCH_U(ctx, "ntt_rows", pts + (double)p.batch * (double)p.out_len * 32.0, unit_bytes, (ntt_pass_kernel<LOG_T, false, false>),
                        dim3(N1 / cnt, (uint32_t)p.batch), block, lds, a);
        ZG_HIP(hipGetLastError());
    }
    return ZG_OK;
}

// Runs the plan on the context stream.  `tmp` must hold batch * 2^log_n elements whenever the
// transform takes two passes (see ntt_needs_tmp).
static uint32_t ntt_log_t(uint32_t log_n) { return log_n <= 16 ? 10u : 11u; }
bool ntt_needs_tmp(uint32_t log_n) { return log_n > ntt_log_t(log_n); }

int ntt_run(zg_ctx* ctx, const NttPlan& p, Fe* tmp, size_t tmp_stride) {
    ZG_REQUIRE(p.log_n <= 22, ZG_ERR_UNSUPPORTED, "ntt: log_n %u > 22 not built", p.log_n);
    if (p.batch == 0) return ZG_OK;
    Fe* tw = nullptr;
    ZG_TRY(get_twiddles(ctx, p.log_n, p.omega, &tw));
    // nine-limb butterflies in the latency form (a lone proof: its transforms finish 7-15 % sooner), 8 x 32-bit ones in the
    // throughput form (under twelve provers the nine-limb pass measured +0.4 ... +0.6 % ms/proof: 127 VGPRs instead of 76, the CU's
    // LDS full at four workgroups, 27 % more multiply-adds and a 0.7 % lower clock -- DESIGN.md section 5); ZG_NTT9 = 0 / 1 forces one
    const int k9 = knob(K_NTT9);
    const bool nine = k9 < 0 ? ctx->msm_pair : k9 != 0;
    if (ntt_log_t(p.log_n) == 10) return nine ? launch_passes<10, true>(ctx, p, tw, tmp, tmp_stride) : launch_passes<10, false>(ctx, p, tw, tmp, tmp_stride);
    return nine ? launch_passes<11, true>(ctx, p, tw, tmp, tmp_stride) : launch_passes<11, false>(ctx, p, tw, tmp, tmp_stride);
}

static int ensure_lds_attr(zg_ctx* ctx) {
    DeviceState& ds = device_state(ctx->device);
    std::lock_guard<std::mutex> lock(ds.mu);
    if (ds.ntt_attrs) return ZG_OK;
    // tiles above 64 KB need the opt-in dynamic LDS limit
    const int big = 160 * 1024;
    ZG_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<11, true, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, big));
    ZG_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<11, false, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, big));
    ZG_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<11, false, false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, big));
    ZG_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<10, true, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, big));
    ZG_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<10, false, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, big));
    ZG_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<10, false, false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, big));
    ZG_HIP(hipFuncSetAttribute((const void*)ntt9_pass_kernel<11, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    ZG_HIP(hipFuncSetAttribute((const void*)ntt9_pass_kernel<11, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    ZG_HIP(hipFuncSetAttribute((const void*)ntt9_pass_kernel<11, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    ZG_HIP(hipFuncSetAttribute((const void*)ntt9_pass_kernel<10, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    ZG_HIP(hipFuncSetAttribute((const void*)ntt9_pass_kernel<10, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    ZG_HIP(hipFuncSetAttribute((const void*)ntt9_pass_kernel<10, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    ds.ntt_attrs = true;
    return ZG_OK;
}

// Device-resident batched transform, in place on d_a (through a workspace copy when two passes
// are needed).
// d_out <- NTT(d_in), both [batch][stride]; d_out may be d_in.  (The first pass reads d_in, the last one writes
// d_out: a caller that needs the input afterwards saves itself a copy.)
int ntt_batch_to_dev(zg_ctx* ctx, const Fe* d_in, Fe* d_out, size_t stride, size_t batch, uint32_t log_n, const Fe& omega,
                     const Fe* divisor, const Grouping* grp) {
    ZG_TRY(ensure_lds_attr(ctx));
    WsScope ws(ctx);
    NttPlan p;
    if (grp) p.grp = *grp;
    p.in = d_in; p.in_stride = stride;
    p.out = d_out; p.out_stride = stride;
    p.batch = batch;
    p.log_n = log_n;
    p.omega = omega;
    p.in_len = p.out_len = 1u << log_n;
    if (divisor) {
        p.scale_out = true;
        p.scale = *divisor;
    }
    Fe* tmp = nullptr;
    size_t n = (size_t)1 << log_n;
    if (ntt_needs_tmp(log_n)) {
        tmp = ws.get<Fe>(batch * n);
        if (!tmp) return ZG_ERR_OOM;
    }
    return ntt_run(ctx, p, tmp, n);
}

int ntt_batch_dev(zg_ctx* ctx, Fe* d_a, size_t stride, size_t batch, uint32_t log_n, const Fe& omega,
                  const Fe* divisor) {
    return ntt_batch_to_dev(ctx, d_a, d_a, stride, batch, log_n, omega, divisor);
}


// hat: the evaluations come out multiplied by 2^5, i.e. in the 2^261 Montgomery form evaluate_h's
// nine-limb arithmetic works in (the factor rides on the coset constants: one extra product for the
// entries with j % 3 == 0, n of the 8n loaded)
int coeff_to_extended_dev(zg_ctx* ctx, const Fe* d_in, size_t in_stride, Fe* d_out, size_t out_stride,
                          size_t batch, uint32_t k, uint32_t ext_k, bool hat) {
    return coeff_to_coset_dev(ctx, d_in, in_stride, 1u << k, d_out, out_stride, batch, ext_k, hat, 1, nullptr);
}

// The general form: `in_len` coefficients (<= 2^ext_k) evaluated on the coset zeta^zeta_pow * <omega_(2^ext_k)>
// (zeta_pow = 1: EvaluationDomain's own coset; 2: the second coset of the prover's split extended domain).
int coeff_to_coset_dev(zg_ctx* ctx, const Fe* d_in, size_t in_stride, uint32_t in_len, Fe* d_out, size_t out_stride,
                       size_t batch, uint32_t ext_k, bool hat, int zeta_pow, const Grouping* grp) {
    ZG_REQUIRE(zeta_pow == 1 || zeta_pow == 2, ZG_ERR_INVALID_ARG, "coeff_to_coset: zeta power %d", zeta_pow);
    ZG_REQUIRE(in_len <= (1u << ext_k), ZG_ERR_INVALID_ARG, "coeff_to_coset: %u coefficients for 2^%u points", in_len, ext_k);
    ZG_TRY(ensure_lds_attr(ctx));
    WsScope ws(ctx);
    NttPlan p;
    if (grp) p.grp = *grp;
    p.in = d_in; p.in_stride = in_stride;
    p.out = d_out; p.out_stride = out_stride;
    p.batch = batch;
    p.log_n = ext_k;
    p.omega = host_domain_omega(ext_k);
    p.in_len = in_len;
    p.out_len = 1u << ext_k;
    p.coset_in = 1;
    p.zin0 = Fr::one();
    p.zin1 = zeta_pow == 1 ? fr_zeta() : Fr::sqr(fr_zeta());   // shift^1
    p.zin2 = zeta_pow == 1 ? Fr::sqr(fr_zeta()) : fr_zeta();   // shift^2 (zeta^4 = zeta)
    if (hat) {
        const Fe c32 = Fr::from_u64(32);
        p.coset_in = 2;
        p.zin0 = c32;
        p.zin1 = Fr::mul(p.zin1, c32);
        p.zin2 = Fr::mul(p.zin2, c32);
    }
    size_t n = (size_t)1 << ext_k;
    Fe* tmp = nullptr;
    if (ntt_needs_tmp(ext_k)) {
        tmp = ws.get<Fe>(batch * n);
        if (!tmp) return ZG_ERR_OOM;
    } else if (d_in == d_out) {
        // single pass is in-place safe
    }
    return ntt_run(ctx, p, tmp, n);
}

// unhat: the input is in the 2^261 form (see coeff_to_extended_dev); the output scale takes the 2^-5
int extended_to_coeff_dev(zg_ctx* ctx, Fe* d_evals, uint32_t k, uint32_t ext_k, size_t out_len,
                          Fe* d_out, bool unhat) {
    (void)k;
    return coset_to_coeff_dev(ctx, d_evals, ext_k, out_len, d_out, unhat, 1, 1, 0, 0);
}

// `batch` arrays: evaluations at d_evals + b * in_stride (in_stride = 0: 2^ext_k), coefficients to d_out + b * out_stride
int coset_to_coeff_dev(zg_ctx* ctx, Fe* d_evals, uint32_t ext_k, size_t out_len, Fe* d_out, bool unhat, int zeta_pow,
                       size_t batch, size_t in_stride, size_t out_stride) {
    ZG_REQUIRE(zeta_pow == 1 || zeta_pow == 2, ZG_ERR_INVALID_ARG, "coset_to_coeff: zeta power %d", zeta_pow);
    ZG_TRY(ensure_lds_attr(ctx));
    WsScope ws(ctx);
    size_t n = (size_t)1 << ext_k;
    ZG_REQUIRE(out_len <= n, ZG_ERR_INVALID_ARG, "extended_to_coeff: out_len %zu > 2^%u", out_len, ext_k);
    NttPlan p;
    p.in = d_evals; p.in_stride = in_stride ? in_stride : n;
    p.out = d_out; p.out_stride = out_stride ? out_stride : out_len;
    p.batch = batch;
    p.log_n = ext_k;
    p.omega = Fr::inv(host_domain_omega(ext_k));
    p.in_len = (uint32_t)n;
    p.out_len = (uint32_t)out_len;
    p.scale_out = true;
    p.scale = Fr::inv(Fr::from_u64((uint64_t)n * (unhat ? 32u : 1u)));
    p.coset_out = true;
    p.zout1 = zeta_pow == 1 ? Fr::sqr(fr_zeta()) : fr_zeta();  // shift^-1
    p.zout2 = zeta_pow == 1 ? fr_zeta() : Fr::sqr(fr_zeta());  // shift^-2
    Fe* tmp = nullptr;
    if (ntt_needs_tmp(ext_k)) {
        tmp = ws.get<Fe>(batch * n);
        if (!tmp) return ZG_ERR_OOM;
    } else if (d_out != d_evals) {
        // single pass reads everything before it writes: fine for distinct or equal buffers
    }
    return ntt_run(ctx, p, tmp, n);
}

}  // namespace zg

using namespace zg;

static inline Fe to_fe(const zg_fr* p) {
    Fe r;
    memcpy(&r, p, 32);
    return r;
}

extern "C" {

int zg_ntt_batch_dev(zg_ctx* ctx, void* d_a, size_t stride_elems, size_t batch, uint32_t log_n,
                     const zg_fr* omega, const zg_fr* divisor) {
    ZG_REQUIRE(ctx && d_a && omega, ZG_ERR_INVALID_ARG, "zg_ntt_batch_dev: null argument");
    ZG_REQUIRE(stride_elems >= ((size_t)1 << log_n) || batch <= 1, ZG_ERR_INVALID_ARG,
               "zg_ntt_batch_dev: stride %zu < 2^%u", stride_elems, log_n);
    ZG_ENTER(ctx);
    Fe om = to_fe(omega), dv;
    if (divisor) dv = to_fe(divisor);
    return ntt_batch_dev(ctx, (Fe*)d_a, stride_elems, batch, log_n, om, divisor ? &dv : nullptr);
}

int zg_intt_batch(zg_ctx* ctx, zg_fr* const* a, size_t batch, uint32_t log_n, const zg_fr* omega_inv,
                  const zg_fr* divisor) {
    ZG_REQUIRE(ctx && omega_inv && (a || batch == 0), ZG_ERR_INVALID_ARG, "zg_ntt: null argument");
    ZG_REQUIRE(log_n <= 22, ZG_ERR_UNSUPPORTED, "zg_ntt: log_n %u > 22 not built", log_n);
    if (batch == 0) return ZG_OK;
    ZG_ENTER(ctx);
    WsScope ws(ctx);
    size_t n = (size_t)1 << log_n;
    Fe* d = ws.get<Fe>(batch * n);
    if (!d) return ZG_ERR_OOM;
    for (size_t b = 0; b < batch; b++) {
        ZG_REQUIRE(a[b] != nullptr, ZG_ERR_INVALID_ARG, "zg_ntt: a[%zu] is null", b);
        ZG_HIP(hipMemcpyAsync(d + b * n, a[b], n * 32, hipMemcpyHostToDevice, ctx->stream));
    }
    Fe om = to_fe(omega_inv), dv;
    if (divisor) dv = to_fe(divisor);
    ZG_TRY(ntt_batch_dev(ctx, d, n, batch, log_n, om, divisor ? &dv : nullptr));
    for (size_t b = 0; b < batch; b++)
        ZG_HIP(hipMemcpyAsync(a[b], d + b * n, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(hipStreamSynchronize(ctx->stream));
    return ZG_OK;
}

int zg_ntt_batch(zg_ctx* ctx, zg_fr* const* a, size_t batch, uint32_t log_n, const zg_fr* omega) {
    return zg_intt_batch(ctx, a, batch, log_n, omega, nullptr);
}

int zg_ntt(zg_ctx* ctx, zg_fr* a, uint32_t log_n, const zg_fr* omega) {
    zg_fr* arr[1] = {a};
    return zg_intt_batch(ctx, arr, 1, log_n, omega, nullptr);
}

int zg_intt(zg_ctx* ctx, zg_fr* a, uint32_t log_n, const zg_fr* omega_inv, const zg_fr* divisor) {
    zg_fr* arr[1] = {a};
    return zg_intt_batch(ctx, arr, 1, log_n, omega_inv, divisor);
}

int zg_coeff_to_extended_batch_dev(zg_ctx* ctx, const void* d_coeffs, size_t in_stride_elems,
                                   void* d_out, size_t out_stride_elems, size_t batch, uint32_t k,
                                   uint32_t ext_k) {
    ZG_REQUIRE(ctx && d_coeffs && d_out, ZG_ERR_INVALID_ARG, "zg_coeff_to_extended: null argument");
    ZG_REQUIRE(k <= ext_k && ext_k <= 22, ZG_ERR_UNSUPPORTED, "zg_coeff_to_extended: k=%u ext_k=%u", k, ext_k);
    ZG_ENTER(ctx);
    return coeff_to_extended_dev(ctx, (const Fe*)d_coeffs, in_stride_elems, (Fe*)d_out,
                                 out_stride_elems, batch, k, ext_k, false);
}

int zg_coeff_to_extended(zg_ctx* ctx, const zg_fr* coeffs, uint32_t k, uint32_t ext_k, zg_fr* out) {
    ZG_REQUIRE(ctx && coeffs && out, ZG_ERR_INVALID_ARG, "zg_coeff_to_extended: null argument");
    ZG_REQUIRE(k <= ext_k && ext_k <= 22, ZG_ERR_UNSUPPORTED, "zg_coeff_to_extended: k=%u ext_k=%u", k, ext_k);
    ZG_ENTER(ctx);
    WsScope ws(ctx);
    size_t n = (size_t)1 << k, en = (size_t)1 << ext_k;
    Fe* din = ws.get<Fe>(n);
    Fe* dout = ws.get<Fe>(en);
    if (!din || !dout) return ZG_ERR_OOM;
    ZG_HIP(hipMemcpyAsync(din, coeffs, n * 32, hipMemcpyHostToDevice, ctx->stream));
    ZG_TRY(coeff_to_extended_dev(ctx, din, n, dout, en, 1, k, ext_k, false));
    ZG_HIP(hipMemcpyAsync(out, dout, en * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(hipStreamSynchronize(ctx->stream));
    return ZG_OK;
}

int zg_extended_to_coeff_dev(zg_ctx* ctx, void* d_evals, uint32_t k, uint32_t ext_k, size_t out_len,
                             void* d_out) {
    ZG_REQUIRE(ctx && d_evals && d_out, ZG_ERR_INVALID_ARG, "zg_extended_to_coeff: null argument");
    ZG_REQUIRE(k <= ext_k && ext_k <= 22, ZG_ERR_UNSUPPORTED, "zg_extended_to_coeff: k=%u ext_k=%u", k, ext_k);
    ZG_ENTER(ctx);
    return extended_to_coeff_dev(ctx, (Fe*)d_evals, k, ext_k, out_len, (Fe*)d_out, false);
}

int zg_extended_to_coeff(zg_ctx* ctx, zg_fr* evals, uint32_t k, uint32_t ext_k, size_t out_len,
                         zg_fr* out) {
    ZG_REQUIRE(ctx && evals && out, ZG_ERR_INVALID_ARG, "zg_extended_to_coeff: null argument");
    ZG_REQUIRE(k <= ext_k && ext_k <= 22, ZG_ERR_UNSUPPORTED, "zg_extended_to_coeff: k=%u ext_k=%u", k, ext_k);
    ZG_ENTER(ctx);
    WsScope ws(ctx);
    size_t en = (size_t)1 << ext_k;
    ZG_REQUIRE(out_len <= en, ZG_ERR_INVALID_ARG, "zg_extended_to_coeff: out_len too large");
    Fe* din = ws.get<Fe>(en);
    Fe* dout = ws.get<Fe>(out_len ? out_len : 1);
    if (!din || !dout) return ZG_ERR_OOM;
    ZG_HIP(hipMemcpyAsync(din, evals, en * 32, hipMemcpyHostToDevice, ctx->stream));
    ZG_TRY(extended_to_coeff_dev(ctx, din, k, ext_k, out_len, dout, false));
    ZG_HIP(hipMemcpyAsync(out, dout, out_len * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(hipStreamSynchronize(ctx->stream));
    return ZG_OK;
}

}  // extern "C"
