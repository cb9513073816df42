// Probe: latency of a chain of dependent nine-limb products (field9.h Fq9::mul) for 1..4 waves per SIMD, and of a
// chain of dependent EC additions through LDS with 1 / 2 / 4 lanes per addition (the MSM reduction's inner step).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "curve.h"
#include "field9.h"
using namespace zg;

__global__ void chain(F9* out, int iters) {
    F9 a = Fq9Params::one(), b = Fq9Params::k256();
    a.l[0] += threadIdx.x;
    for (int i = 0; i < iters; i++) a = Fq9::mul(a, b);
    out[threadIdx.x] = a;
}

template <int L, bool BARRIER>
__global__ void add_chain(XYZZ* out, int iters) {
    __shared__ XYZZ9 sh[128], inc[128];
    const uint32_t j = threadIdx.x / L, role = threadIdx.x % L;
    if (role == 0) {
        const Fe c261 = Fq9Params::c261_fe();
        const F9 gx = f9_unpack(Fq::mul(Fq::from_u64(1), c261)), gy = f9_unpack(Fq::mul(Fq::from_u64(2), c261));
        XYZZ9 a;
        bool inf = true;
        for (uint32_t i = 0; i < j % 5 + 2; i++) xyzz9_madd(a, inf, gx, gy);
        sh[j] = a;
        xyzz9_madd(a, inf, gx, gy);
        inc[j] = xyzz9_add(a, a);
    }
    __syncthreads();
    for (int i = 0; i < iters; i++) {
        if constexpr (L == 1) {
            sh[j] = xyzz9_add(sh[j], inc[j]);
        } else {
            XSum s = xaddl<L>(&sh[j], &inc[j], role);
            if (BARRIER) __syncthreads();
            xstore<true>(&sh[j], s);
        }
        if (BARRIER) __syncthreads();
    }
    if (role == 0 && blockIdx.x == 0) out[j] = xyzz9_to_xyzz(sh[j], false);
}

template <int L, bool BARRIER>
static void time_adds(XYZZ* d, hipEvent_t e0, hipEvent_t e1, int slots, int blocks = 1, size_t lds = 0) {
    const int iters = 2000;
    hipLaunchKernelGGL((add_chain<L, BARRIER>), dim3(blocks), dim3(slots * L), lds, 0, d, 10);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((add_chain<L, BARRIER>), dim3(blocks), dim3(slots * L), lds, 0, d, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%d lane(s) per addition, %3d slots (%4d threads) x %3d workgroups%s%s: %.2f us per dependent addition\n", L, slots,
           slots * L, blocks, BARRIER ? ", two barriers per step" : "", lds ? ", CU-exclusive LDS" : "", ms * 1e3 / iters);
}

int main() {
    F9* d;
    if (hipMalloc(&d, sizeof(F9) * 1024) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(chain, dim3(1), dim3(threads), 0, 0, d, 100);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(chain, dim3(1), dim3(threads), 0, 0, d, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%4d threads on one CU (%d wave(s) per SIMD): %.3f us per dependent product\n", threads, (threads + 255) / 256,
               ms * 1e3 / iters);
    }
    XYZZ* dx;
    if (hipMalloc(&dx, sizeof(XYZZ) * 128) != hipSuccess) return 1;
    time_adds<1, false>(dx, e0, e1, 128);
    time_adds<2, false>(dx, e0, e1, 128);
    time_adds<4, false>(dx, e0, e1, 64);
    time_adds<4, false>(dx, e0, e1, 128);
    time_adds<2, true>(dx, e0, e1, 128);
    time_adds<4, true>(dx, e0, e1, 64);
    for (int blocks : {64, 128, 192, 256, 384})
        time_adds<2, true>(dx, e0, e1, 128, blocks);
    time_adds<2, true>(dx, e0, e1, 128, 192, 64 << 10);
    time_adds<4, true>(dx, e0, e1, 64, 384);
    return 0;
}
