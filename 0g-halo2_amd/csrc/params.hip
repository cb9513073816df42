// KZG structured reference string on the GPU -- replaces ParamsKZG::<Bn256>::new(k)
// (halo2_proofs v2023_04_20 src/poly/kzg/commitment.rs `ParamsKZG::setup`; reference call sites
// /root/reference/benches/bench.rs:19 and src/main.rs:232).  Upstream draws the toxic scalar s from
// OsRng; here the caller passes it so that runs are reproducible.
//   g[i]          = s^i * G
//   g_lagrange[i] = L_i(s) * G,  L_i(s) = (s^n - 1)/n * omega^i / (s - omega^i)
// One lane per point: 254-step double-and-add in XYZZ, one Fermat inversion to go affine.  This is a
// one-off (outside the timed region of the reference's bench, benches/bench.rs:30-36).
#include "common.h"

namespace zg {

__device__ __forceinline__ void st_affine(Affine* p, const Affine& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.x.l[0], v.x.l[1], v.x.l[2], v.x.l[3]);
    q[1] = make_uint4(v.x.l[4], v.x.l[5], v.x.l[6], v.x.l[7]);
    q[2] = make_uint4(v.y.l[0], v.y.l[1], v.y.l[2], v.y.l[3]);
    q[3] = make_uint4(v.y.l[4], v.y.l[5], v.y.l[6], v.y.l[7]);
}

__global__ void srs_kernel(Affine* __restrict__ g, Affine* __restrict__ gl, Fe s, Fe omega, Fe mult,
                           uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine gen;
    gen.x = Fq::from_u64(1);
    gen.y = Fq::from_u64(2);
    Fe si = Fr::pow_u64(s, i);
    Fe raw = Fr::to_raw(si);
    st_affine(g + i, xyzz_to_affine(xyzz_mul_raw(gen, raw.l)));
    Fe wi = Fr::pow_u64(omega, i);
    Fe li = Fr::mul(Fr::mul(mult, wi), Fr::inv(Fr::sub(s, wi)));
    raw = Fr::to_raw(li);
    st_affine(gl + i, xyzz_to_affine(xyzz_mul_raw(gen, raw.l)));
}

}  // namespace zg

using namespace zg;

extern "C" int zg_params_new_dev(zg_ctx* ctx, uint32_t k, const zg_fr* s, void* d_g, void* d_g_lagrange) {
    ZG_REQUIRE(ctx && s && d_g && d_g_lagrange, ZG_ERR_INVALID_ARG, "zg_params_new_dev: null argument");
    ZG_REQUIRE(k <= 24, ZG_ERR_UNSUPPORTED, "zg_params_new_dev: k=%u > 24", k);
    ZG_ENTER(ctx);
    Fe sv;
    memcpy(&sv, s, 32);
    uint32_t n = 1u << k;
    Fe omega = host_domain_omega(k);
    Fe mult = Fr::mul(Fr::sub(Fr::pow_u64(sv, n), Fr::one()), Fr::inv(Fr::from_u64(n)));
    ZG_LAUNCH(ctx, "srs", 0, srs_kernel, dim3((n + 63) / 64), dim3(64), 0, (Affine*)d_g, (Affine*)d_g_lagrange,
              sv, omega, mult, n);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

extern "C" int zg_params_new(zg_ctx* ctx, uint32_t k, const zg_fr* s, zg_g1_affine* g, zg_g1_affine* g_lagrange) {
    ZG_REQUIRE(ctx && s && g && g_lagrange, ZG_ERR_INVALID_ARG, "zg_params_new: null argument");
    ZG_REQUIRE(k <= 24, ZG_ERR_UNSUPPORTED, "zg_params_new: k=%u > 24", k);
    ZG_ENTER(ctx);
    WsScope ws(ctx);
    size_t n = (size_t)1 << k;
    Affine* dg = ws.get<Affine>(n);
    Affine* dl = ws.get<Affine>(n);
    if (ws.failed) return ZG_ERR_OOM;
    ZG_TRY(zg_params_new_dev(ctx, k, s, dg, dl));
    ZG_HIP(hipMemcpyAsync(g, dg, n * sizeof(Affine), hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(hipMemcpyAsync(g_lagrange, dl, n * sizeof(Affine), hipMemcpyDeviceToHost, ctx->stream));
    ZG_HIP(hipStreamSynchronize(ctx->stream));
    return ZG_OK;
}
