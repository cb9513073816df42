// BN254 prime-field arithmetic for gfx950 (and the host side of the same library).
//
// Replaces halo2curves 0.3.3 `bn256::{Fr, Fq}` (reference import: /root/reference/src/wnn.rs:18)
// on the device.  Memory format is the reference's: 4 x u64 little-endian limbs in Montgomery form
// (R = 2^256); in registers an element is 8 x u32 limbs, because the CDNA4 VALU multiplies
// 32 x 32 -> 64 (v_mad_u64_u32) and has no 64-bit vector multiplier.  No MFMA: this is modular
// integer arithmetic, not a dense contraction.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZG_HD __host__ __device__ __forceinline__

namespace zg {

struct alignas(16) Fe {
    uint32_t l[8];
};

// ---- per-field constants --------------------------------------------------------------------
struct FrParams {
    static ZG_HD uint32_t p(int i) {
        constexpr uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                   0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return P[i];
    }
    static constexpr uint32_t P_TOP = 0x30644e72u;  // = p(7): the modulus is below 2^254 (msm.hip relies on it)
    static constexpr uint32_t INV = 0xefffffffu;  // -p^-1 mod 2^32
    static constexpr bool IS_FR = true;
    static ZG_HD Fe one() {  // R mod p
        return Fe{{0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu,
                   0x9a07df2fu, 0x0e0a77c1u}};
    }
    static ZG_HD Fe r2() {  // R^2 mod p
        return Fe{{0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du,
                   0x7f4e44a5u, 0x0216d0b1u}};
    }
};

struct FqParams {
    static ZG_HD uint32_t p(int i) {
        constexpr uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                   0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return P[i];
    }
    static constexpr uint32_t INV = 0xe4866389u;
    static constexpr bool IS_FR = false;
    static ZG_HD Fe one() {
        return Fe{{0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu,
                   0x9a07df2fu, 0x0e0a77c1u}};
    }
    static ZG_HD Fe r2() {
        return Fe{{0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu,
                   0xcab8351fu, 0x06d89f71u}};
    }
};

// ---- generic helpers ------------------------------------------------------------------------
ZG_HD Fe fe_zero() { return Fe{{0, 0, 0, 0, 0, 0, 0, 0}}; }

ZG_HD bool fe_is_zero(const Fe& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.l[i];
    return o == 0;
}

ZG_HD bool fe_eq(const Fe& a, const Fe& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

// a + b with carry out.  Device: add-with-carry builtins, which select to one v_add_co_u32 +
// seven v_addc_co_u32 (the 64-bit accumulator form compiles to v_lshl_add_u64 + moves, ~4x longer).
ZG_HD uint32_t add8(uint32_t* o, const uint32_t* a, const uint32_t* b) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned int c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = __builtin_addc(a[i], b[i], c, &c);
    return c;
#else
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a[i] + b[i];
        o[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
#endif
}

// a - b with borrow out (1 = borrowed)
ZG_HD uint32_t sub8(uint32_t* o, const uint32_t* a, const uint32_t* b) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned int c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = __builtin_subc(a[i], b[i], c, &c);
    return c;
#else
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a[i] - (int64_t)b[i];
        o[i] = (uint32_t)c;
        c >>= 32;  // arithmetic shift: 0 or -1
    }
    return (uint32_t)(c & 1);
#endif
}

template <class P>
struct Field {
    static ZG_HD Fe zero() { return fe_zero(); }
    static ZG_HD Fe one() { return P::one(); }

    // conditional subtract: a in [0, 2p) -> [0, p)
    static ZG_HD void reduce_once(Fe& a) {
        uint32_t pm[8], t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) pm[i] = P::p(i);
        uint32_t borrow = sub8(t, a.l, pm);
        if (!borrow) {
#pragma unroll
            for (int i = 0; i < 8; i++) a.l[i] = t[i];
        }
    }

    static ZG_HD Fe add(const Fe& a, const Fe& b) {
        Fe o;
        add8(o.l, a.l, b.l);  // p < 2^254: no carry out
        reduce_once(o);
        return o;
    }

    static ZG_HD Fe sub(const Fe& a, const Fe& b) {
        Fe o;
        uint32_t borrow = sub8(o.l, a.l, b.l);
        if (borrow) {
            uint32_t pm[8];
#pragma unroll
            for (int i = 0; i < 8; i++) pm[i] = P::p(i);
            add8(o.l, o.l, pm);
        }
        return o;
    }

    static ZG_HD Fe neg(const Fe& a) {
        if (fe_is_zero(a)) return a;
        Fe o;
        uint32_t pm[8];
#pragma unroll
        for (int i = 0; i < 8; i++) pm[i] = P::p(i);
        sub8(o.l, pm, a.l);
        return o;
    }

    static ZG_HD Fe dbl(const Fe& a) { return add(a, a); }

    // Montgomery product a*b*R^-1 mod p.
    //   device: product scanning over 16 columns, every partial product one v_mad_u64_u32 with its
    //           carry caught by one v_addc_co_u32 (generated asm, tools/gen_mont_mul.py);
    //   host  : CIOS with fused multiply/reduce rows (the modulus leaves two spare top bits, so the
    //           running value never needs a 10th word).
    static ZG_HD Fe mul(const Fe& a, const Fe& b) {
#if defined(__HIP_DEVICE_COMPILE__)
        uint64_t acc = 0;
        uint32_t ovf = 0;
        uint32_t m[8], r[8];
#include "mont_mul_gfx950.inc"
        Fe o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.l[i] = r[i];
        // the column sums keep the value below 2p: one conditional subtraction
        reduce_once(o);
        return o;
#else
        // four 64-bit limbs, CIOS over unsigned __int128 (the transcript's host arithmetic -- normalising commitments,
        // opening-point powers, evaluation batches -- sits between the phases of every proof)
        typedef unsigned __int128 u128;
        uint64_t A[4], B[4], M[4], t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            A[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
            B[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
            M[i] = (uint64_t)P::p(2 * i) | ((uint64_t)P::p(2 * i + 1) << 32);
        }
        // -p^-1 mod 2^64 from the 32-bit constant (one Newton step doubles the valid bits)
        const uint64_t ninv32 = P::INV;
        const uint64_t inv64 = ninv32 * (2 + M[0] * ninv32);  // (-x)(2 + m(-x)) = -(x(2 - mx)) for x = p^-1 mod 2^32
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)A[j] * B[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (uint64_t)c;
            t[5] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * inv64;
            c = (u128)m * M[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; j++) {
                c += (u128)m * M[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = t[5] + (uint64_t)(c >> 64);
        }
        Fe o;
        for (int i = 0; i < 4; i++) {
            o.l[2 * i] = (uint32_t)t[i];
            o.l[2 * i + 1] = (uint32_t)(t[i] >> 32);
        }
        reduce_once(o);
        return o;
#endif
    }

    static ZG_HD Fe sqr(const Fe& a) { return mul(a, a); }

    static ZG_HD Fe from_raw(const Fe& a) { return mul(a, P::r2()); }  // canonical -> Montgomery

    static ZG_HD Fe to_raw(const Fe& a) {  // Montgomery -> canonical integer
        Fe one_raw = fe_zero();
        one_raw.l[0] = 1;
        return mul(a, one_raw);
    }

    static ZG_HD Fe from_u64(uint64_t v) {
        Fe t = fe_zero();
        t.l[0] = (uint32_t)v;
        t.l[1] = (uint32_t)(v >> 32);
        return from_raw(t);
    }

    // a^e for a 256-bit exponent given as 8 LE u32 limbs (left-to-right square and multiply)
    static ZG_HD Fe pow(const Fe& a, const uint32_t e[8]) {
        Fe res = one();
        for (int i = 7; i >= 0; i--)
            for (int b = 31; b >= 0; b--) {
                res = sqr(res);
                if ((e[i] >> b) & 1) res = mul(res, a);
            }
        return res;
    }

    static ZG_HD Fe pow_u64(const Fe& a, uint64_t e) {
        Fe res = one();
        bool started = false;
        for (int b = 63; b >= 0; b--) {
            if (started) res = sqr(res);
            if ((e >> b) & 1) {
                res = mul(res, a);
                started = true;
            }
        }
        return res;
    }

    // Fermat inversion a^(p-2); 0 -> 0.  ~380 dependent products: use inv() unless a constant-time
    // chain is wanted.
    static ZG_HD Fe inv_fermat(const Fe& a) {
        uint32_t e[8];
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = P::p(i);
        e[0] -= 2;
        return pow(a, e);
    }

    // Inversion by the binary extended Euclidean algorithm on the canonical integer (HAC 14.61):
    // shifts, adds and subtracts only -- an order of magnitude shorter than the Fermat chain when a
    // single lane has to do it (grand-product denominators, affine normalisation).  0 -> 0.
    static ZG_HD Fe inv(const Fe& a) {
        if (fe_is_zero(a)) return a;
#if !defined(__HIP_DEVICE_COMPILE__)
        // Host: the same algorithm on four 64-bit limbs (the inversion behind every commitment phase's affine
        // normalisation sits on the critical path of a proof: 21 us with 32-bit limbs, 4 us this way).
        {
            typedef unsigned __int128 u128;
            uint64_t pm4[4], u[4], v[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
            for (int i = 0; i < 4; i++) {
                pm4[i] = (uint64_t)P::p(2 * i) | ((uint64_t)P::p(2 * i + 1) << 32);
                u[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
                v[i] = pm4[i];
            }
            auto add4 = [](uint64_t* o, const uint64_t* x, const uint64_t* y) {
                u128 c = 0;
                for (int i = 0; i < 4; i++) {
                    c += (u128)x[i] + y[i];
                    o[i] = (uint64_t)c;
                    c >>= 64;
                }
                return (uint64_t)c;
            };
            auto sub4 = [](uint64_t* o, const uint64_t* x, const uint64_t* y) {  // returns the borrow
                uint64_t b = 0;
                for (int i = 0; i < 4; i++) {
                    const uint64_t d = x[i] - y[i], b1 = x[i] < y[i];
                    o[i] = d - b;
                    b = b1 | (d < b);
                }
                return b;
            };
            auto shr1 = [](uint64_t* w, uint64_t top) {
                for (int i = 0; i < 3; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 63);
                w[3] = (w[3] >> 1) | (top << 63);
            };
            auto is_one = [](const uint64_t* w) { return ((w[0] ^ 1ull) | w[1] | w[2] | w[3]) == 0; };
            auto geq = [](const uint64_t* x, const uint64_t* y) {
                for (int i = 3; i >= 0; i--)
                    if (x[i] != y[i]) return x[i] > y[i];
                return true;
            };
            auto halve_mod = [&](uint64_t* w) {
                uint64_t carry = 0;
                if (w[0] & 1ull) carry = add4(w, w, pm4);
                shr1(w, carry);
            };
            auto sub_mod = [&](uint64_t* x, const uint64_t* y) {
                if (sub4(x, x, y)) add4(x, x, pm4);
            };
            int budget = 1100;  // (see the device loop below: ends for every input)
            while (!is_one(u) && !is_one(v) && budget > 0) {
                while ((u[0] & 1ull) == 0 && --budget > 0) { shr1(u, 0); halve_mod(x1); }
                while ((v[0] & 1ull) == 0 && --budget > 0) { shr1(v, 0); halve_mod(x2); }
                --budget;
                if (geq(u, v)) { sub4(u, u, v); sub_mod(x1, x2); }
                else { sub4(v, v, u); sub_mod(x2, x1); }
            }
            // budget exhausted (a non-canonical multiple of p, a corrupted modulus): 0, as for a = 0 -- never a value that
            // could pass for an inverse
            if (!is_one(u) && !is_one(v)) return fe_zero();
            const uint64_t* res = is_one(u) ? x1 : x2;
            Fe r;
            for (int i = 0; i < 4; i++) {
                r.l[2 * i] = (uint32_t)res[i];
                r.l[2 * i + 1] = (uint32_t)(res[i] >> 32);
            }
            return mul(mul(r, P::r2()), P::r2());
        }
#endif
        uint32_t pm[8];
#pragma unroll
        for (int i = 0; i < 8; i++) pm[i] = P::p(i);
        // u = a*R (the Montgomery residue taken as an integer), v = p; invariants:
        //   x1 * (aR) == u (mod p),  x2 * (aR) == v (mod p)
        uint32_t u[8], v[8], x1[8], x2[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            u[i] = a.l[i];
            v[i] = pm[i];
            x1[i] = 0;
            x2[i] = 0;
        }
        x1[0] = 1;
        auto is_one = [](const uint32_t* w) {
            uint32_t o = w[0] ^ 1u;
#pragma unroll
            for (int i = 1; i < 8; i++) o |= w[i];
            return o == 0;
        };
        auto shr1 = [](uint32_t* w, uint32_t top) {
#pragma unroll
            for (int i = 0; i < 7; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 31);
            w[7] = (w[7] >> 1) | (top << 31);
        };
        auto halve_mod = [&](uint32_t* w) {  // w <- w/2 mod p
            uint32_t carry = 0;
            if (w[0] & 1u) carry = add8(w, w, pm);
            shr1(w, carry);
        };
        auto geq = [](const uint32_t* x, const uint32_t* y) {
            for (int i = 7; i >= 0; i--) {
                if (x[i] > y[i]) return true;
                if (x[i] < y[i]) return false;
            }
            return true;
        };
        auto sub_mod = [&](uint32_t* x, const uint32_t* y) {  // x <- x - y mod p
            uint32_t borrow = sub8(x, x, y);
            if (borrow) add8(x, x, pm);
        };
        // Every step removes a bit of u or v: 512 halvings end the loop for any input coprime to p.  The budget makes
        // the loop end for EVERY input (a multiple of p, a wrong modulus): a wave that never finishes takes the GPU
        // with it -- round 3 lost three runs to this loop when an edit dropped the initialisation of pm above.
        int budget = 1100;
        while (!is_one(u) && !is_one(v) && budget > 0) {
            while ((u[0] & 1u) == 0 && --budget > 0) {
                shr1(u, 0);
                halve_mod(x1);
            }
            while ((v[0] & 1u) == 0 && --budget > 0) {
                shr1(v, 0);
                halve_mod(x2);
            }
            --budget;
            if (geq(u, v)) {
                sub8(u, u, v);
                sub_mod(x1, x2);
            } else {
                sub8(v, v, u);
                sub_mod(x2, x1);
            }
        }
        Fe r;
        if (!is_one(u) && !is_one(v)) return fe_zero();  // (budget exhausted: 0, never a would-be inverse)
        const uint32_t* res = is_one(u) ? x1 : x2;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = res[i];
        // r = (aR)^-1 = a^-1 R^-1 as an integer; two Montgomery products by R^2 give a^-1 R
        return mul(mul(r, P::r2()), P::r2());
    }
};

using Fr = Field<FrParams>;
using Fq = Field<FqParams>;

// halo2curves bn256::Fr constants, Montgomery form (tests re-derive them with Python integers)
ZG_HD Fe fr_root_of_unity() {  // 7^((r-1)/2^28)
    return Fe{{0xb639feb8u, 0x9632c7c5u, 0x0d0ff299u, 0x985ce340u, 0x01b0ecd8u, 0xb2dd8800u,
               0x6d98ce29u, 0x1d69070du}};
}
ZG_HD Fe fr_delta() {  // 7^(2^28)
    return Fe{{0xefd78855u, 0x9a0c322bu, 0x249b563cu, 0x46e82d14u, 0xe0b0b7a7u, 0x5983a663u,
               0xaaa111adu, 0x22ab452bu}};
}
ZG_HD Fe fr_zeta() {  // 7^((r-1)/3): the coset generator of EvaluationDomain
    return Fe{{0x4a0329b3u, 0x93e7cedeu, 0x7a96c167u, 0x7d4fdca7u, 0xb19a750au, 0x8be4ba08u,
               0xa5661c25u, 0x1cbd5653u}};
}
constexpr uint32_t FR_S = 28;  // two-adicity

}  // namespace zg
