// Nine 29-bit limbs: the register form of Fq used inside the MSM bucket accumulation (device only).
//
// Why: with 32-bit limbs a column of partial products overflows a 64-bit accumulator after one
// product, so every v_mad_u64_u32 drags a v_addc_co_u32 behind it (128 + 128 per Montgomery product).
// With 29-bit limbs a whole column (9 data products + 9 reduction products < 2^62.2) fits the 64-bit
// accumulator: every partial product is ONE v_mad_i64_i32 and there is no carry word -- 162 mads and
// ~70 other instructions per product instead of ~340 (measured: 167 vs 134 G products/s,
// tools/mont29_probe.hip).
//
// Representation: value = sum l[i] * 2^(29 i), limbs SIGNED.  "Normalised": l[0..7] in [0, 2^29),
// l[8] a small signed remainder.  Elements are residues mod p, NOT kept canonical: the Montgomery
// radix is 2^261 >> p, so a product of values of magnitude < 2^258 lands in (-2^256, 2^256 + p) with
// no final subtraction; additions and subtractions are plain limb-wise (no carries, no bias), and the
// only rule is that an operand of mul9 has limb magnitudes < 2^29 -- true for normalised values and
// for the difference of two normalised values (sums need norm9 first).
// Montgomery form here is x * 2^261; conversion from / to the library's x * 2^256 form is one product
// by a constant (k266 / k256).  Memory keeps the packed 8 x 32-bit canonical form.
#pragma once

#include "curve.h"
#include "field.h"

namespace zg {

struct F9 {
    int32_t l[9];
};

constexpr int32_t MASK29 = (1 << 29) - 1;

struct Fq9Params {
    static __device__ __forceinline__ int32_t p(int i) {
        constexpr int32_t P[9] = {0x187cfd47, 0x010460b6, 0x1c72a34f, 0x02d522d0, 0x1585d978,
                                  0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e};
        return P[i];
    }
    static constexpr uint32_t INV29 = 0x04866389u;  // -p^-1 mod 2^29
    static __device__ __forceinline__ F9 one() {    // 2^261 mod p
        return F9{{0x157ccc21, 0x141c2758, 0x185230d3, 0x014c0419, 0x0aa36fb9, 0x1d4240ce, 0x11d54c07, 0x052ac7a8, 0x000dc836}};
    }
    static __device__ __forceinline__ F9 k256() {   // 2^256 mod p: mul9(x * 2^261, k256) = x * 2^256
        return F9{{0x058f0d9d, 0x1aea1c6e, 0x11c2cf74, 0x11d651eb, 0x1462c0a7, 0x11b7bc3c, 0x1cbd99ba, 0x183340fb, 0x000e0a77}};
    }
    // 2^261 mod p as an 8 x 32-bit integer: Fq::mul(X, c261) = X * 2^5, i.e. x*2^256 -> x*2^261
    static ZG_HD Fe c261_fe() {
        return Fe{{0x157ccc21u, 0x4e8384ebu, 0x0ce148c3u, 0xfb90a602u, 0x819caa36u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u}};
    }
    // 2^271 mod p: Fq::inv of the packed form of x * 2^261 returns x^-1 * 2^251; mul9(that, k271) = x^-1 * 2^261
    static __device__ __forceinline__ F9 k271() {
        return F9{{0x1d1c9c4b, 0x08a372ee, 0x1273abad, 0x17c9d397, 0x1698b0a7, 0x09c89e50, 0x177e12ab, 0x185f3518, 0x001ed378}};
    }
};

struct Fr9Params {
    static __device__ __forceinline__ int32_t p(int i) {
        constexpr int32_t P[9] = {0x10000001, 0x1f0fac9f, 0x0e5c2450, 0x07d090f3, 0x1585d283,
                                  0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e};
        return P[i];
    }
    static constexpr uint32_t INV29 = 0x0fffffffu;
    static __device__ __forceinline__ F9 one() {  // 2^261 mod r
        return F9{{0x0fffff57, 0x1ea70ab4, 0x052c068b, 0x17504f49, 0x0aa8075b, 0x1d4240ce, 0x11d54c07, 0x052ac7a8, 0x000dc836}};
    }
    // 2^261 mod r as an 8 x 32-bit integer: Fr::mul(X, c261) = X * 2^5, i.e. x*2^256 -> x*2^261
    static ZG_HD Fe c261_fe() {
        return Fe{{0x8fffff57u, 0x2fd4e156u, 0xa494b01au, 0x75bba827u, 0x819caa80u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u}};
    }
};

// 8 x 32-bit packed (canonical, < 2^256) -> normalised limbs
__device__ __forceinline__ F9 f9_unpack(const Fe& a) {
    F9 o;
    o.l[0] = (int32_t)(a.l[0] & (uint32_t)MASK29);
#pragma unroll
    for (int i = 1; i < 8; i++) {
        const int bit = 29 * i, w = bit / 32, s = bit % 32;
        // (two 32-bit shifts, not a 64-bit funnel: on kernel-argument operands the latter spills them to scratch)
        o.l[i] = (int32_t)(((a.l[w] >> s) | (a.l[w + 1] << (32 - s))) & (uint32_t)MASK29);
    }
    o.l[8] = (int32_t)(a.l[7] >> 8);
    return o;
}

// canonical normalised limbs (all in [0, 2^29), value < 2^256) -> packed
__device__ __forceinline__ Fe f9_pack(const F9& a) {
    Fe o;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int bit = 32 * w, i = bit / 29, s = bit % 29;  // word w starts inside limb i at bit s
        uint32_t v = (uint32_t)a.l[i] >> s;
        v |= (uint32_t)a.l[i + 1] << (29 - s);
        if (29 - s + 29 < 32 && i + 2 < 9) v |= (uint32_t)a.l[i + 2] << (58 - s);
        o.l[w] = v;
    }
    return o;
}

// carry propagation: l[0..7] into [0, 2^29), l[8] takes the signed remainder (value unchanged)
__device__ __forceinline__ F9 f9_norm(const F9& a) {
    F9 o;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int32_t v = a.l[i] + c;
        o.l[i] = v & MASK29;
        c = v >> 29;
    }
    o.l[8] = a.l[8] + c;
    return o;
}

__device__ __forceinline__ F9 f9_add(const F9& a, const F9& b) {
    F9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = a.l[i] + b.l[i];
    return o;
}
__device__ __forceinline__ F9 f9_sub(const F9& a, const F9& b) {
    F9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = a.l[i] - b.l[i];
    return o;
}
__device__ __forceinline__ F9 f9_neg(const F9& a) {
    F9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = -a.l[i];
    return o;
}
__device__ __forceinline__ bool f9_limbs_zero(const F9& a) {
    int32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.l[i];
    return o == 0;
}

template <class P>
struct Field9 {
    // a * b * 2^-261 mod p.  Operand limb magnitudes < 2^29 (one of them may reach 2^30); operand
    // values of magnitude < 2^258.  Result normalised, value in (-2^256, 2^256 + p).
    static __device__ __forceinline__ F9 mul(const F9& a, const F9& b) {
        int64_t acc = 0;
        int32_t m[9];
        F9 r;
#pragma unroll
        for (int k = 0; k < 17; k++) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (j >= 0 && j < 9) acc += (int64_t)a.l[i] * (int64_t)b.l[j];
            }
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (i < k && j >= 0 && j < 9) acc += (int64_t)m[i] * (int64_t)P::p(j);
            }
            if (k < 9) {
                m[k] = (int32_t)(((uint32_t)acc * P::INV29) & (uint32_t)MASK29);
                acc += (int64_t)m[k] * (int64_t)P::p(0);
                acc >>= 29;
            } else {
                r.l[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)MASK29);
                acc >>= 29;
            }
        }
        r.l[8] = (int32_t)acc;
        return r;
    }
    // (a*b + c*d) * 2^-261 (SUB: a*b - c*d) with ONE Montgomery reduction: 81 + 81 + 81 mads instead of
    // 2 * 162.  All four operands need limb magnitudes < 2^29 (a column then holds 27 * 2^58 < 2^63).
    template <bool SUB>
    static __device__ __forceinline__ F9 mul2(const F9& a, const F9& b, const F9& c, const F9& d) {
        int64_t acc = 0;
        int32_t m[9];
        F9 r;
#pragma unroll
        for (int k = 0; k < 17; k++) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (j >= 0 && j < 9) {
                    acc += (int64_t)a.l[i] * (int64_t)b.l[j];
                    if (SUB) acc -= (int64_t)c.l[i] * (int64_t)d.l[j];
                    else acc += (int64_t)c.l[i] * (int64_t)d.l[j];
                }
            }
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (i < k && j >= 0 && j < 9) acc += (int64_t)m[i] * (int64_t)P::p(j);
            }
            if (k < 9) {
                m[k] = (int32_t)(((uint32_t)acc * P::INV29) & (uint32_t)MASK29);
                acc += (int64_t)m[k] * (int64_t)P::p(0);
                acc >>= 29;
            } else {
                r.l[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)MASK29);
                acc >>= 29;
            }
        }
        r.l[8] = (int32_t)acc;
        return r;
    }

    // a * a * 2^-261: the 36 cross products are taken once against the doubled operand (limbs < 2^30,
    // which the accumulator bound allows on one side) -- 45 + 81 mads instead of 81 + 81.
    static __device__ __forceinline__ F9 sqr(const F9& a) {
        int64_t acc = 0;
        int32_t m[9], d[9];
        F9 r;
#pragma unroll
        for (int i = 0; i < 9; i++) d[i] = a.l[i] + a.l[i];
#pragma unroll
        for (int k = 0; k < 17; k++) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (j >= 0 && j < 9 && i < j) acc += (int64_t)d[i] * (int64_t)a.l[j];
                if (j == i) acc += (int64_t)a.l[i] * (int64_t)a.l[i];
            }
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (i < k && j >= 0 && j < 9) acc += (int64_t)m[i] * (int64_t)P::p(j);
            }
            if (k < 9) {
                m[k] = (int32_t)(((uint32_t)acc * P::INV29) & (uint32_t)MASK29);
                acc += (int64_t)m[k] * (int64_t)P::p(0);
                acc >>= 29;
            } else {
                r.l[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)MASK29);
                acc >>= 29;
            }
        }
        r.l[8] = (int32_t)acc;
        return r;
    }

    // value in (-2p, 3p), any limbs -> the canonical representative in [0, p), normalised
    static __device__ F9 canon(const F9& x) {
        F9 a = f9_norm(x);
        F9 pp;
#pragma unroll
        for (int i = 0; i < 9; i++) pp.l[i] = P::p(i);
#pragma unroll 1
        for (int it = 0; it < 2; it++)
            if (a.l[8] < 0) a = f9_norm(f9_add(a, pp));
#pragma unroll 1
        for (int it = 0; it < 2; it++) {
            F9 t = f9_norm(f9_sub(a, pp));
            if (t.l[8] >= 0) a = t;
        }
        return a;
    }

    // exact test x == 0 (mod p) for a NORMALISED value in [0, 8p): the low limb filters, the rare hit is
    // settled by subtraction
    static __device__ bool is_zero_mod_p(const F9& a) {
        bool cand = false;
        uint32_t lo = 0;  // low limb of j*p
#pragma unroll
        for (int j = 0; j < 8; j++) {
            cand |= (uint32_t)a.l[0] == lo;
            lo = (lo + (uint32_t)P::p(0)) & (uint32_t)MASK29;
        }
        if (!cand) return false;
        F9 t = a;
        F9 pp;
#pragma unroll
        for (int i = 0; i < 9; i++) pp.l[i] = P::p(i);
#pragma unroll 1
        for (int j = 0; j < 8; j++) {
            if (f9_limbs_zero(t)) return true;
            t = f9_norm(f9_sub(t, pp));
            if (t.l[8] < 0) return false;
        }
        return false;
    }
};

using Fq9 = Field9<Fq9Params>;
using Fr9 = Field9<Fr9Params>;

// sum_t a_t * b_t with ONE Montgomery reduction at the end (dot products: eval_polynomial against a power table, the
// linear combinations of the multiopen argument).  The 17 product columns are kept as they are -- a term costs its 81
// multiply-adds and nothing else -- and carried back to 29 bits every third term: operands are unpacked canonical
// values (limbs in [0, 2^29)), so a column grows by at most 9 * 2^58 per term and 3 * 9 * 2^58 + 2^35 < 2^63.
template <class P>
struct Dot9 {
    int64_t c[17];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < 17; k++) c[k] = 0;
    }
    __device__ __forceinline__ void mac(const F9& a, const F9& b) {
#pragma unroll
        for (int k = 0; k < 17; k++) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (j >= 0 && j < 9) c[k] += (int64_t)a.l[i] * (int64_t)b.l[j];
            }
        }
    }
    // columns 0..15 back into [0, 2^29); the top column takes what is left (value unchanged)
    __device__ __forceinline__ void carry() {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            c[k + 1] += c[k] >> 29;
            c[k] &= (int64_t)MASK29;
        }
    }
    // (sum) * 2^-261 mod p, normalised, value in [0, sum / 2^261 + p).  Call after carry(); the sum must stay below
    // 2^521 (2^13 terms of canonical operands of a 254-bit field), which keeps the result under f9_reduce_pack's 2^263.
    __device__ __forceinline__ F9 reduce() const {
        int64_t acc = 0;
        int32_t m[9];
        F9 r;
#pragma unroll
        for (int k = 0; k < 17; k++) {
            acc += c[k];
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (i < k && j >= 0 && j < 9) acc += (int64_t)m[i] * (int64_t)P::p(j);
            }
            if (k < 9) {
                m[k] = (int32_t)(((uint32_t)acc * P::INV29) & (uint32_t)MASK29);
                acc += (int64_t)m[k] * (int64_t)P::p(0);
                acc >>= 29;
            } else {
                r.l[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)MASK29);
                acc >>= 29;
            }
        }
        r.l[8] = (int32_t)acc;
        return r;
    }
};

// Normalised value of magnitude < 2^263 -> canonical packed form.  The quotient by p is estimated from
// the top limb in float (2^232 / p = 3.1531751e-7 for both BN254 moduli): the float product is within
// 8e-5 of v/p and the dropped low limbs add less than 4e-7, so floor(estimate - 1e-4) is floor(v/p) or
// one less -- the remainder lies in [0, 2p) and one conditional subtraction finishes.
template <class P>
__device__ __forceinline__ Fe f9_reduce_pack(const F9& v) {
    const int32_t q = (int32_t)floorf((float)v.l[8] * 3.153175148504101e-07f - 1.0e-4f);
    F9 a, t;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int64_t w = (int64_t)v.l[i] - (int64_t)q * (int64_t)P::p(i) + c;
        a.l[i] = (int32_t)((uint32_t)w & (uint32_t)MASK29);
        c = w >> 29;
    }
    a.l[8] = (int32_t)((int64_t)v.l[8] - (int64_t)q * (int64_t)P::p(8) + c);
    int32_t b = 0;  // t = a - p, normalised
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int32_t w = a.l[i] - P::p(i) + b;
        t.l[i] = w & MASK29;
        b = w >> 29;
    }
    t.l[8] = a.l[8] - P::p(8) + b;
    const bool ge = t.l[8] >= 0;
#pragma unroll
    for (int i = 0; i < 9; i++) a.l[i] = ge ? t.l[i] : a.l[i];
    return f9_pack(a);
}

// XYZZ point in the nine-limb form (coordinates x * 2^261, normalised); identity <=> zz limbs all zero
struct alignas(16) XYZZ9 {
    F9 x, y, zz, zzz;
};
static_assert(sizeof(XYZZ9) == 144, "XYZZ9 layout");

__device__ __forceinline__ XYZZ9 xyzz9_identity() {
    XYZZ9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.x.l[i] = r.y.l[i] = r.zz.l[i] = r.zzz.l[i] = 0;
    return r;
}
__device__ __forceinline__ bool xyzz9_is_identity(const XYZZ9& a) { return f9_limbs_zero(a.zz); }

__device__ __forceinline__ XYZZ9 ld_xyzz9(const XYZZ9* p) {
    XYZZ9 r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = q[i];
    return r;
}
__device__ __forceinline__ void st_xyzz9(XYZZ9* p, const XYZZ9& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    const uint4* s = reinterpret_cast<const uint4*>(&v);
#pragma unroll
    for (int i = 0; i < 9; i++) q[i] = s[i];
}

// acc + q, q affine in the nine-limb 2^261 form (madd-2008-s, same case analysis as xyzz_madd).
// `inf` is the accumulator's identity flag.  All coordinates in and out are normalised.
__device__ __forceinline__ void xyzz9_madd(XYZZ9& a, bool& inf, const F9& qx, const F9& qy) {
    if (inf) {
        a.x = qx;
        a.y = f9_norm(qy);  // (a negated y arrives with negative limbs)
        a.zz = Fq9Params::one();
        a.zzz = Fq9Params::one();
        inf = false;
        return;
    }
    const F9 u2 = Fq9::mul(qx, a.zz);
    const F9 s2 = Fq9::mul(qy, a.zzz);
    const F9 p = f9_sub(u2, a.x);  // limb magnitudes < 2^29: both operands normalised
    const F9 r = f9_sub(s2, a.y);
    const F9 pp = Fq9::sqr(p);
    // p == 0 (mod q)  <=>  pp == 0 (mod q); pp is normalised and lies in [0, 2^256 + q)
    if (__builtin_expect(pp.l[8] >= 0 && Fq9::is_zero_mod_p(pp), 0)) {
        const F9 rr = Fq9::sqr(r);
        if (!Fq9::is_zero_mod_p(rr)) {  // q = -acc
            inf = true;
            return;
        }
        // q = acc: dbl-2008-s-1 on the affine point (rare: every operand normalised on the way)
        const F9 u = f9_norm(f9_add(qy, qy));
        const F9 v = Fq9::sqr(u);
        const F9 w = Fq9::mul(u, v);
        const F9 s = Fq9::mul(qx, v);
        const F9 x2 = Fq9::sqr(qx);
        const F9 m = f9_norm(f9_add(f9_add(x2, x2), x2));
        a.x = f9_norm(f9_sub(f9_sub(Fq9::sqr(m), s), s));
        a.y = f9_norm(f9_sub(Fq9::mul(m, f9_norm(f9_sub(s, a.x))), Fq9::mul(w, f9_norm(qy))));
        a.zz = v;
        a.zzz = w;
        return;
    }
    const F9 ppp = Fq9::mul(p, pp);
    const F9 qq = Fq9::mul(a.x, pp);
    const F9 x3 = f9_norm(f9_sub(f9_sub(f9_sub(Fq9::sqr(r), ppp), qq), qq));
    const F9 t = f9_sub(qq, x3);
    a.y = Fq9::mul2<true>(r, t, a.y, ppp);  // R (Q - X3) - Y1 PPP, one reduction
    a.x = x3;
    a.zz = Fq9::mul(a.zz, pp);
    a.zzz = Fq9::mul(a.zzz, ppp);
}

// dbl-2008-s-1 in the nine-limb form (only reached when an addition meets two equal points)
__device__ __noinline__ XYZZ9 xyzz9_dbl(const XYZZ9& a) {
    if (xyzz9_is_identity(a)) return a;
    XYZZ9 o;
    const F9 u = f9_norm(f9_add(a.y, a.y));
    const F9 v = Fq9::sqr(u);
    const F9 w = Fq9::mul(u, v);
    const F9 s = Fq9::mul(a.x, v);
    const F9 xx = Fq9::sqr(a.x);
    const F9 m = f9_norm(f9_add(f9_add(xx, xx), xx));
    o.x = f9_norm(f9_sub(f9_sub(Fq9::sqr(m), s), s));
    o.y = f9_norm(f9_sub(Fq9::mul(m, f9_norm(f9_sub(s, o.x))), Fq9::mul(w, a.y)));
    o.zz = Fq9::mul(v, a.zz);
    o.zzz = Fq9::mul(w, a.zzz);
    return o;
}

// add-2008-s in the nine-limb form, same case analysis as xyzz_add.  Inputs and output normalised.
__device__ __forceinline__ XYZZ9 xyzz9_add(const XYZZ9& a, const XYZZ9& b) {
    if (xyzz9_is_identity(a)) return b;
    if (xyzz9_is_identity(b)) return a;
    const F9 u1 = Fq9::mul(a.x, b.zz);
    const F9 u2 = Fq9::mul(b.x, a.zz);
    const F9 s1 = Fq9::mul(a.y, b.zzz);
    const F9 s2 = Fq9::mul(b.y, a.zzz);
    const F9 p = f9_sub(u2, u1);
    const F9 r = f9_sub(s2, s1);
    const F9 pp = Fq9::sqr(p);
    if (__builtin_expect(pp.l[8] >= 0 && Fq9::is_zero_mod_p(pp), 0)) {
        if (Fq9::is_zero_mod_p(Fq9::sqr(r))) return xyzz9_dbl(a);
        return xyzz9_identity();
    }
    const F9 ppp = Fq9::mul(p, pp);
    const F9 qq = Fq9::mul(u1, pp);
    XYZZ9 o;
    o.x = f9_norm(f9_sub(f9_sub(f9_sub(Fq9::sqr(r), ppp), qq), qq));
    o.y = Fq9::mul2<true>(r, f9_sub(qq, o.x), s1, ppp);
    o.zz = Fq9::mul(Fq9::mul(a.zz, b.zz), pp);
    o.zzz = Fq9::mul(Fq9::mul(a.zzz, b.zzz), ppp);
    return o;
}


// ---- two lanes per addition -------------------------------------------------------------------------
// The bucket reduction of the MSM is a chain of DEPENDENT additions on a nearly idle chip, so its
// wall time is the latency of one addition times the depth.  Lanes 2i ("A") and 2i+1 ("B") of a wave
// share one addition: seven of the fourteen products each, intermediate values swapped with DPP
// quad_perm [1,0,3,2] moves (45 moves per addition), results written per coordinate by the lane that
// holds them (A: zz; B: x, y, zzz).  Same case analysis as xyzz9_add; the equal-x case falls back to
// lane A doing the generic addition alone.
__device__ __forceinline__ int32_t dpp_swap1(int32_t v) {
    int32_t r = __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
    // keep the move a move: folded into a following subtraction (GCN DPP combine) the operands came out
    // swapped -- dpp(h) - k instead of h - dpp(k) -- on this toolchain
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ F9 f9_swap(const F9& a) {
    F9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = dpp_swap1(a.l[i]);
    return o;
}
__device__ __forceinline__ F9 f9_sel(bool c, const F9& a, const F9& b) {
    F9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = c ? a.l[i] : b.l[i];
    return o;
}
__device__ __forceinline__ F9 ld_f9(const F9* p) {
    F9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = p->l[i];
    return r;
}
__device__ __forceinline__ void st_f9(F9* p, const F9& v) {
#pragma unroll
    for (int i = 0; i < 9; i++) p->l[i] = v.l[i];
}

// What one lane holds of a sum and which coordinates it has to write (bit 0 x, 1 y, 2 zz, 3 zzz).
struct XSum {
    XYZZ9 r;
    uint32_t wm;
};

template <bool PAIR>
__device__ __forceinline__ void xstore(XYZZ9* dst, const XSum& s) {
    if (!PAIR) {
        st_xyzz9(dst, s.r);
        return;
    }
    if (s.wm & 1u) st_f9(&dst->x, s.r.x);
    if (s.wm & 2u) st_f9(&dst->y, s.r.y);
    if (s.wm & 4u) st_f9(&dst->zz, s.r.zz);
    if (s.wm & 8u) st_f9(&dst->zzz, s.r.zzz);
    // the partner lane reads these coordinates next: keep the compiler from carrying stale copies of
    // what the other lane writes (wavefront scope: both lanes run in lockstep, no hardware wait needed)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// *pa + *pb.  PAIR: called by both lanes of a pair with the same pointers, role = lane parity.
template <bool PAIR>
__device__ __forceinline__ XSum xadd(const XYZZ9* pa, const XYZZ9* pb, uint32_t role) {
    XSum s;
    if (!PAIR) {
        s.r = xyzz9_add(ld_xyzz9(pa), ld_xyzz9(pb));
        s.wm = 0xFu;
        return s;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const bool A = role == 0;
    const XYZZ9* own = A ? pa : pb;
    const XYZZ9* oth = A ? pb : pa;
    const F9 mx = ld_f9(&own->x), my = ld_f9(&own->y);
    const F9 mz = ld_f9(&oth->zz), mzzz = ld_f9(&oth->zzz);
    const F9 e1 = ld_f9(A ? &own->zz : &own->zzz);  // A: ZZ1, B: ZZZ2
    // identity flags as both lanes see them (identity points are all-zero)
    const bool a_inf = A ? f9_limbs_zero(e1) : f9_limbs_zero(mz);
    const bool b_inf = A ? f9_limbs_zero(mz) : f9_limbs_zero(e1);
    if (a_inf || b_inf) {
        if (a_inf) {  // result = b: A holds b.zz, B holds b.x, b.y, b.zzz
            s.r.x = mx; s.r.y = my; s.r.zz = mz; s.r.zzz = e1;
            s.wm = A ? 4u : 11u;
        } else {      // result = a: A holds a.x, a.y, a.zz, B holds a.zzz
            s.r.x = mx; s.r.y = my; s.r.zz = e1; s.r.zzz = mzzz;
            s.wm = A ? 7u : 8u;
        }
        return s;
    }
    const F9 u = Fq9::mul(mx, mz);      // A: U1 = X1 ZZ2      B: U2 = X2 ZZ1
    const F9 sv = Fq9::mul(my, mzzz);   // A: S1 = Y1 ZZZ2     B: S2 = Y2 ZZZ1
    const F9 e = Fq9::mul(e1, A ? mz : mzzz);  // A: ZZ1 ZZ2   B: ZZZ2 ZZZ1
    const F9 ou = f9_swap(u), os = f9_swap(sv);
    const F9 p = A ? f9_sub(ou, u) : f9_sub(u, ou);    // U2 - U1
    const F9 r = A ? f9_sub(os, sv) : f9_sub(sv, os);  // S2 - S1
    const F9 f = Fq9::sqr(f9_sel(A, p, r));            // A: PP      B: RR
    const F9 of = f9_swap(f);
    const F9 pp = f9_sel(A, f, of), rr = f9_sel(A, of, f);
    // equal x (doubling or inverse): decided by A on PP, told to B, settled by A alone
    int32_t rare = (A && pp.l[8] >= 0 && Fq9::is_zero_mod_p(pp)) ? 1 : 0;
    rare |= dpp_swap1(rare);
    if (__builtin_expect(rare != 0, 0)) {
        s.wm = 0;
        if (A) {
            s.r = xyzz9_add(ld_xyzz9(pa), ld_xyzz9(pb));
            s.wm = 0xFu;
        }
        return s;
    }
    const F9 u1 = f9_sel(A, u, ou);
    const F9 g = Fq9::mul(f9_sel(A, p, u1), pp);       // A: PPP = P PP    B: Q = U1 PP
    const F9 og = f9_swap(g);
    const F9 ppp = f9_sel(A, g, og), qq = f9_sel(A, og, g);
    const F9 x3 = f9_norm(f9_sub(f9_sub(f9_sub(rr, ppp), qq), qq));
    const F9 h = Fq9::mul(f9_sel(A, e, r), f9_sel(A, pp, f9_sub(qq, x3)));  // A: ZZ3 = ZZ12 PP   B: R (Q - X3)
    const F9 k = Fq9::mul(f9_sel(A, sv, e), ppp);                            // A: S1 PPP          B: ZZZ3 = ZZZ12 PPP
    const F9 ok = f9_swap(k);
    s.r.x = x3;
    s.r.y = f9_norm(f9_sub(h, ok));  // (meaningful on B)
    s.r.zz = h;                      // (meaningful on A)
    s.r.zzz = k;                     // (meaningful on B)
    s.wm = A ? 4u : 11u;
    return s;
}

// ---- *pa + *pb by FOUR lanes (a quad, role = lane & 3): four rounds of one product per lane instead of the
// pair's seven,
//      round 1   U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1
//      round 2   ZZ1 ZZ2     | PP = P^2    | ZZZ1 ZZZ2    | RR = R^2
//      round 3   Q = U1 PP   | PPP = P PP  | ZZ3          | -
//      round 4   -           | S1 PPP      | ZZZ3         | R (Q - X3)
// with quad_perm exchanges / broadcasts in between; lane 3 ends up with x and y, lane 2 with zz and zzz.
// Every lane runs the same instruction stream (squares go through the general product: a divergent
// branch would cost more than it saves).  Identity operands and equal x are settled as in xadd<true>.
template <int CTRL>
__device__ __forceinline__ int32_t dpp_quad(int32_t v) {
    int32_t r = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
    asm volatile("" : "+v"(r));  // (see dpp_swap1)
    return r;
}
template <int CTRL>
__device__ __forceinline__ F9 f9_quad(const F9& a) {
    F9 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = dpp_quad<CTRL>(a.l[i]);
    return o;
}
constexpr int QUAD_SWAP1 = 0xB1, QUAD_SWAP2 = 0x4E, QUAD_B0 = 0x00, QUAD_B1 = 0x55, QUAD_B2 = 0xAA;

__device__ __forceinline__ XSum xadd4(const XYZZ9* pa, const XYZZ9* pb, uint32_t role) {
    XSum s;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const bool first = (role & 1u) == 0;  // lanes 0, 2 start from a's coordinate, lanes 1, 3 from b's
    const bool ypart = role >= 2;
    const XYZZ9* own = first ? pa : pb;
    const XYZZ9* oth = first ? pb : pa;
    const F9 m = ld_f9(ypart ? &own->y : &own->x);      // X1 | X2 | Y1 | Y2
    const F9 mz = ld_f9(ypart ? &oth->zzz : &oth->zz);  // ZZ2 | ZZ1 | ZZZ2 | ZZZ1
    const F9 e1 = ld_f9(ypart ? &pa->zzz : &pa->zz);    // ZZ1 | . | ZZZ1 | .
    // identity flags from lane 0 (it holds ZZ1 and ZZ2; identity points are all-zero)
    int32_t fl = (f9_limbs_zero(e1) ? 1 : 0) | (f9_limbs_zero(mz) ? 2 : 0);
    fl = dpp_quad<QUAD_B0>(fl);
    if (fl) {  // result = the other operand (b when a is the identity), copied by the two writer lanes
        const XYZZ9* src = (fl & 1) ? pb : pa;
        s.wm = 0;
        if (role == 3) { s.r.x = ld_f9(&src->x); s.r.y = ld_f9(&src->y); s.wm = 3u; }
        if (role == 2) { s.r.zz = ld_f9(&src->zz); s.r.zzz = ld_f9(&src->zzz); s.wm = 12u; }
        return s;
    }
    const F9 t1 = Fq9::mul(m, mz);                       // U1 | U2 | S1 | S2
    const F9 o1 = f9_quad<QUAD_SWAP1>(t1);
    const F9 d = first ? f9_sub(o1, t1) : f9_sub(t1, o1);  // P | P | R | R
    const F9 t2 = Fq9::mul(f9_sel(first, e1, d), f9_sel(first, mz, d));  // ZZ1 ZZ2 | PP | ZZZ1 ZZZ2 | RR
    int32_t rare = (role == 1 && t2.l[8] >= 0 && Fq9::is_zero_mod_p(t2)) ? 1 : 0;
    rare = dpp_quad<QUAD_B1>(rare);
    if (__builtin_expect(rare != 0, 0)) {  // equal x: lane 0 settles it alone
        s.wm = 0;
        if (role == 0) {
            s.r = xyzz9_add(ld_xyzz9(pa), ld_xyzz9(pb));
            s.wm = 0xFu;
        }
        return s;
    }
    const F9 pp = f9_quad<QUAD_B1>(t2);
    const F9 z12 = f9_quad<QUAD_SWAP2>(t2);              // lane 2: ZZ1 ZZ2
    const F9 t3 = Fq9::mul(role == 0 ? t1 : role == 2 ? z12 : d, pp);  // Q | PPP | ZZ3 | .
    const F9 ppp = f9_quad<QUAD_B1>(t3), qq = f9_quad<QUAD_B0>(t3);
    const F9 x3 = f9_norm(f9_sub(f9_sub(f9_sub(t2, ppp), qq), qq));   // lane 3: RR - PPP - 2Q
    const F9 tt = f9_sub(qq, x3);
    const F9 s1 = f9_quad<QUAD_B2>(t1);
    const F9 t4 = Fq9::mul(role == 3 ? d : role == 2 ? t2 : s1, role == 3 ? tt : ppp);  // . | S1 PPP | ZZZ3 | R (Q - X3)
    const F9 o4 = f9_quad<QUAD_B1>(t4);
    s.r.x = x3;
    s.r.y = f9_norm(f9_sub(t4, o4));  // (meaningful on lane 3)
    s.r.zz = t3;                       // (meaningful on lane 2)
    s.r.zzz = t4;
    s.wm = role == 3 ? 3u : role == 2 ? 12u : 0u;
    return s;
}

// L lanes per addition: 1 (registers), 2 (xadd<true>) or 4 (xadd4); the result is written with xstore<L > 1>.
template <int L>
__device__ __forceinline__ XSum xaddl(const XYZZ9* pa, const XYZZ9* pb, uint32_t role) {
    if constexpr (L == 4) return xadd4(pa, pb, role);
    else return xadd<L == 2>(pa, pb, role);
}

// ---- mixed addition acc += (qx, qy) by a pair of lanes, the running sum split between them for the whole
// chain:   lane A: m = X, z = ZZ, w = ZZZ        lane B: m = Y, z = ZZZ.
// Five products per lane instead of ten (madd-2008-s): U2 | S2,  PP | RR,  Q | PPP,  ZZ3 | Y1 PPP,
// ZZZ3 | R (Q - X3); five quad_perm exchanges.  Both lanes pass the same qx, qy (already sign-adjusted) and
// carry the same `inf`.  Equal x (doubling / inverse) is settled by both lanes running the one-lane
// formula on the reassembled point.
struct PairAcc {
    F9 m, z, w;
};
__device__ __forceinline__ void xmadd_pair(PairAcc& a, bool& inf, const F9& qx, const F9& qy, bool A) {
    if (inf) {
        a.m = A ? qx : f9_norm(qy);  // (a negated y arrives with negative limbs)
        a.z = Fq9Params::one();
        a.w = Fq9Params::one();
        inf = false;
        return;
    }
    const F9 v = Fq9::mul(f9_sel(A, qx, qy), a.z);   // A: U2 = qx ZZ1      B: S2 = qy ZZZ1
    const F9 d = f9_sub(v, a.m);                     // A: P                B: R
    const F9 f = Fq9::sqr(d);                        // A: PP               B: RR
    const F9 od = f9_swap(d), of = f9_swap(f);       // A: R, RR            B: P, PP
    int32_t rare = (A && f.l[8] >= 0 && Fq9::is_zero_mod_p(f)) ? 1 : 0;
    rare |= dpp_swap1(rare);
    if (__builtin_expect(rare != 0, 0)) {
        const F9 om = f9_swap(a.m), oz = f9_swap(a.z);
        XYZZ9 full;
        full.x = f9_sel(A, a.m, om);
        full.y = f9_sel(A, om, a.m);
        full.zz = f9_sel(A, a.z, oz);
        full.zzz = f9_sel(A, a.w, a.z);
        xyzz9_madd(full, inf, qx, qy);
        a.m = f9_sel(A, full.x, full.y);
        a.z = f9_sel(A, full.zz, full.zzz);
        a.w = full.zzz;
        return;
    }
    const F9 g = Fq9::mul(f9_sel(A, a.m, od), f9_sel(A, f, of));  // A: Q = X1 PP      B: PPP = P PP
    const F9 og = f9_swap(g);                                     // A: PPP            B: Q
    const F9 x3 = f9_norm(f9_sub(f9_sub(f9_sub(of, og), g), g));  // A: RR - PPP - 2Q
    const F9 ot = f9_swap(f9_sub(g, x3));                         //                   B: T = Q - X3
    const F9 h = Fq9::mul(f9_sel(A, a.z, a.m), f9_sel(A, f, g));  // A: ZZ3 = ZZ1 PP   B: Y1 PPP
    const F9 k = Fq9::mul(f9_sel(A, a.w, d), f9_sel(A, og, ot));  // A: ZZZ3 = ZZZ1 PPP   B: R T
    const F9 ok = f9_swap(k);                                     //                   B: ZZZ3
    a.m = f9_sel(A, x3, f9_norm(f9_sub(k, h)));
    a.z = f9_sel(A, h, ok);
    a.w = k;
}

// nine-limb 2^261 form -> the library's packed XYZZ (coordinates x * 2^256, canonical)
__device__ __forceinline__ XYZZ xyzz9_to_xyzz(const XYZZ9& a, bool inf) {
    if (inf || xyzz9_is_identity(a)) return xyzz_identity();
    XYZZ o;
    const F9 k = Fq9Params::k256();
    o.x = f9_pack(Fq9::canon(Fq9::mul(a.x, k)));
    o.y = f9_pack(Fq9::canon(Fq9::mul(a.y, k)));
    o.zz = f9_pack(Fq9::canon(Fq9::mul(a.zz, k)));
    o.zzz = f9_pack(Fq9::canon(Fq9::mul(a.zzz, k)));
    return o;
}

}  // namespace zg
