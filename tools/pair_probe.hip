// Probe: two- and four-lane additions (field9.h xadd<true>, xadd4) against the one-lane xyzz9_add on the same inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "curve.h"
#include "field9.h"
using namespace zg;

__device__ XYZZ9 mulG(uint32_t k) {  // k*G by repeated mixed addition (k small)
    const Fe c261 = Fq9Params::c261_fe();
    const F9 gx = f9_unpack(Fq::mul(Fq::from_u64(1), c261)), gy = f9_unpack(Fq::mul(Fq::from_u64(2), c261));
    XYZZ9 acc;
    bool inf = true;
    for (uint32_t i = 0; i < k; i++) xyzz9_madd(acc, inf, gx, gy);
    return inf ? xyzz9_identity() : acc;
}

template <int L>
__global__ void probe(uint32_t* bad, XYZZ* outp, XYZZ* outs, int mode) {
    __shared__ XYZZ9 sa[128], sb[128], sd[128];
    const uint32_t j = threadIdx.x / L, role = threadIdx.x % L;
    if (role == 0) {
        uint32_t ka = j + 2, kb = 3 * j + 5;
        if (mode == 1 && j % 7 == 0) ka = 0;            // a identity
        if (mode == 1 && j % 11 == 0) kb = 0;           // b identity
        if (mode == 2 && j % 5 == 0) kb = ka;           // doubling
        sa[j] = mulG(ka);
        sb[j] = mulG(kb);
    }
    __syncthreads();
    XSum s = xaddl<L>(&sa[j], &sb[j], role);
    xstore<true>(&sd[j], s);
    __syncthreads();
    if (role == 0) {
        XYZZ9 ref = xyzz9_add(sa[j], sb[j]);
        XYZZ p = xyzz9_to_xyzz(sd[j], false), q = xyzz9_to_xyzz(ref, false);
        outp[j] = p; outs[j] = q;
        // compare as affine-equivalent: x1*zz2 == x2*zz1 etc. (representations must even be identical here)
        bool eq = fe_eq(p.x, q.x) && fe_eq(p.y, q.y) && fe_eq(p.zz, q.zz) && fe_eq(p.zzz, q.zzz);
        if (!eq) atomicAdd(bad, 1u);
    }
}

int main() {
    uint32_t* d_bad; XYZZ *dp, *ds;
    hipMalloc(&d_bad, 4); hipMalloc(&dp, sizeof(XYZZ) * 128); hipMalloc(&ds, sizeof(XYZZ) * 128);
    for (int lanes = 2; lanes <= 4; lanes += 2)
        for (int mode = 0; mode < 3; mode++) {
            hipMemset(d_bad, 0, 4);
            if (lanes == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(256), 0, 0, d_bad, dp, ds, mode);
            else hipLaunchKernelGGL(probe<4>, dim3(1), dim3(512), 0, 0, d_bad, dp, ds, mode);
            uint32_t bad = 0;
            hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
            XYZZ hp[4], hs[4];
            hipMemcpy(hp, dp, sizeof(hp), hipMemcpyDeviceToHost);
            hipMemcpy(hs, ds, sizeof(hs), hipMemcpyDeviceToHost);
            printf("%d lanes, mode %d: %u of 128 differ; slot 1 x=%08x.. y=%08x.. zz=%08x.. zzz=%08x.. | single x=%08x.. y=%08x.. zz=%08x.. zzz=%08x..\n", lanes,
                   mode, bad, hp[1].x.l[0], hp[1].y.l[0], hp[1].zz.l[0], hp[1].zzz.l[0], hs[1].x.l[0], hs[1].y.l[0], hs[1].zz.l[0],
                   hs[1].zzz.l[0]);
        }
    return 0;
}
