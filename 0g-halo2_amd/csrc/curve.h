// BN254 G1 (y^2 = x^3 + 3 over Fq) group law for gfx950 and the host side of the library.
//
// Replaces halo2curves 0.3.3 `bn256::{G1, G1Affine}` (reference import /root/reference/src/wnn.rs:18)
// on the device.  Memory formats are the reference's: G1Affine = {x, y} 64 B with (0,0) = identity,
// G1 = Jacobian {x, y, z} 96 B with z = 0 = identity.  Bucket accumulators use extended Jacobian
// "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): a mixed add is 8M+2S instead of 7M+4S
// and a full add 12M+2S instead of 11M+5S, which matters because every field product is ~130
// v_mad_u64_u32 on the CDNA4 VALU.
#pragma once

#include "field.h"

namespace zg {

struct alignas(16) Affine {
    Fe x, y;
};
struct alignas(16) Jac {
    Fe x, y, z;
};
struct alignas(16) XYZZ {
    Fe x, y, zz, zzz;
};

ZG_HD bool affine_is_identity(const Affine& p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
ZG_HD bool xyzz_is_identity(const XYZZ& p) { return fe_is_zero(p.zz); }
ZG_HD bool jac_is_identity(const Jac& p) { return fe_is_zero(p.z); }

ZG_HD XYZZ xyzz_identity() {
    XYZZ r;
    r.x = fe_zero();
    r.y = Fq::one();
    r.zz = fe_zero();
    r.zzz = fe_zero();
    return r;
}

ZG_HD XYZZ xyzz_from_affine(const Affine& p) {
    if (affine_is_identity(p)) return xyzz_identity();
    XYZZ r;
    r.x = p.x;
    r.y = p.y;
    r.zz = Fq::one();
    r.zzz = Fq::one();
    return r;
}

ZG_HD Affine affine_neg(const Affine& p) {
    Affine r;
    r.x = p.x;
    r.y = Fq::neg(p.y);
    return r;
}

// mdbl-2008-s-1: 2 * (affine point), a = 0
ZG_HD XYZZ xyzz_dbl_affine(const Affine& p) {
    if (affine_is_identity(p)) return xyzz_identity();
    XYZZ r;
    Fe u = Fq::dbl(p.y);
    Fe v = Fq::sqr(u);
    Fe w = Fq::mul(u, v);
    Fe s = Fq::mul(p.x, v);
    Fe xx = Fq::sqr(p.x);
    Fe m = Fq::add(Fq::dbl(xx), xx);
    r.x = Fq::sub(Fq::sqr(m), Fq::dbl(s));
    r.y = Fq::sub(Fq::mul(m, Fq::sub(s, r.x)), Fq::mul(w, p.y));
    r.zz = v;
    r.zzz = w;
    return r;
}

// dbl-2008-s-1, a = 0
ZG_HD XYZZ xyzz_dbl(const XYZZ& p) {
    if (xyzz_is_identity(p)) return p;
    XYZZ r;
    Fe u = Fq::dbl(p.y);
    Fe v = Fq::sqr(u);
    Fe w = Fq::mul(u, v);
    Fe s = Fq::mul(p.x, v);
    Fe xx = Fq::sqr(p.x);
    Fe m = Fq::add(Fq::dbl(xx), xx);
    r.x = Fq::sub(Fq::sqr(m), Fq::dbl(s));
    r.y = Fq::sub(Fq::mul(m, Fq::sub(s, r.x)), Fq::mul(w, p.y));
    r.zz = Fq::mul(v, p.zz);
    r.zzz = Fq::mul(w, p.zzz);
    return r;
}

// madd-2008-s: acc + (affine q)
ZG_HD XYZZ xyzz_madd(const XYZZ& a, const Affine& q) {
    if (affine_is_identity(q)) return a;
    if (xyzz_is_identity(a)) return xyzz_from_affine(q);
    Fe u2 = Fq::mul(q.x, a.zz);
    Fe s2 = Fq::mul(q.y, a.zzz);
    Fe p = Fq::sub(u2, a.x);
    Fe r = Fq::sub(s2, a.y);
    if (fe_is_zero(p)) {
        if (fe_is_zero(r)) return xyzz_dbl_affine(q);
        return xyzz_identity();
    }
    Fe pp = Fq::sqr(p);
    Fe ppp = Fq::mul(p, pp);
    Fe qq = Fq::mul(a.x, pp);
    XYZZ o;
    o.x = Fq::sub(Fq::sub(Fq::sqr(r), ppp), Fq::dbl(qq));
    o.y = Fq::sub(Fq::mul(r, Fq::sub(qq, o.x)), Fq::mul(a.y, ppp));
    o.zz = Fq::mul(a.zz, pp);
    o.zzz = Fq::mul(a.zzz, ppp);
    return o;
}

// add-2008-s
ZG_HD XYZZ xyzz_add(const XYZZ& a, const XYZZ& b) {
    if (xyzz_is_identity(a)) return b;
    if (xyzz_is_identity(b)) return a;
    Fe u1 = Fq::mul(a.x, b.zz);
    Fe u2 = Fq::mul(b.x, a.zz);
    Fe s1 = Fq::mul(a.y, b.zzz);
    Fe s2 = Fq::mul(b.y, a.zzz);
    Fe p = Fq::sub(u2, u1);
    Fe r = Fq::sub(s2, s1);
    if (fe_is_zero(p)) {
        if (fe_is_zero(r)) return xyzz_dbl(a);
        return xyzz_identity();
    }
    Fe pp = Fq::sqr(p);
    Fe ppp = Fq::mul(p, pp);
    Fe qq = Fq::mul(u1, pp);
    XYZZ o;
    o.x = Fq::sub(Fq::sub(Fq::sqr(r), ppp), Fq::dbl(qq));
    o.y = Fq::sub(Fq::mul(r, Fq::sub(qq, o.x)), Fq::mul(s1, ppp));
    o.zz = Fq::mul(Fq::mul(a.zz, b.zz), pp);
    o.zzz = Fq::mul(Fq::mul(a.zzz, b.zzz), ppp);
    return o;
}

ZG_HD XYZZ xyzz_neg(const XYZZ& a) {
    XYZZ o = a;
    o.y = Fq::neg(a.y);
    return o;
}

// XYZZ -> affine (one inversion); identity -> (0,0)
ZG_HD Affine xyzz_to_affine(const XYZZ& a) {
    Affine o;
    if (xyzz_is_identity(a)) {
        o.x = fe_zero();
        o.y = fe_zero();
        return o;
    }
    Fe i = Fq::inv(Fq::mul(a.zz, a.zzz));  // 1/(ZZ*ZZZ)
    o.x = Fq::mul(a.x, Fq::mul(i, a.zzz)); // X/ZZ
    o.y = Fq::mul(a.y, Fq::mul(i, a.zz));  // Y/ZZZ
    return o;
}

ZG_HD Jac jac_from_affine(const Affine& p) {
    Jac o;
    if (affine_is_identity(p)) {
        o.x = fe_zero();
        o.y = Fq::one();
        o.z = fe_zero();
        return o;
    }
    o.x = p.x;
    o.y = p.y;
    o.z = Fq::one();
    return o;
}

ZG_HD Affine jac_to_affine(const Jac& p) {
    Affine o;
    if (jac_is_identity(p)) {
        o.x = fe_zero();
        o.y = fe_zero();
        return o;
    }
    Fe zi = Fq::inv(p.z);
    Fe zi2 = Fq::sqr(zi);
    o.x = Fq::mul(p.x, zi2);
    o.y = Fq::mul(p.y, Fq::mul(zi2, zi));
    return o;
}

ZG_HD bool affine_on_curve(const Affine& p) {
    if (affine_is_identity(p)) return true;
    Fe lhs = Fq::sqr(p.y);
    Fe rhs = Fq::add(Fq::mul(Fq::sqr(p.x), p.x), Fq::from_u64(3));
    return fe_eq(lhs, rhs);
}

// k * P for a canonical (non-Montgomery) 256-bit scalar given as LE u32 limbs; double-and-add
ZG_HD XYZZ xyzz_mul_raw(const Affine& p, const uint32_t k[8]) {
    XYZZ acc = xyzz_identity();
    for (int i = 7; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            acc = xyzz_dbl(acc);
            if ((k[i] >> b) & 1) acc = xyzz_madd(acc, p);
        }
    return acc;
}

}  // namespace zg
