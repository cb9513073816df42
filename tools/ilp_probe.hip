// Probe: a nine-limb Montgomery product whose partial products do not all hang on ONE accumulator (field9.h Field9::mul: 162
// multiply-adds on a single 64-bit chain).  Here the 81 data products go to 17 column accumulators (independent of the
// reduction), and each quotient digit m[i] is spread over its nine columns as soon as it exists (operand scanning): the only
// serial chain left is column -> m[i] -> carry.  Latency of a chain of dependent products at 1..4 waves per SIMD, and the
// throughput of independent products, for both forms.      hipcc -O3 --offload-arch=gfx950 -I0g-halo2_amd/csrc -Iinclude tools/ilp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "curve.h"
#include "field9.h"
using namespace zg;

template <class P>
__device__ __forceinline__ F9 mul_cols(const F9& a, const F9& b) {
    int64_t c[18];
#pragma unroll
    for (int k = 0; k < 18; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (int64_t)a.l[i] * (int64_t)b.l[j];
    F9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t m = (int32_t)(((uint32_t)c[i] * P::INV29) & (uint32_t)MASK29);
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (int64_t)m * (int64_t)P::p(j);
        c[i + 1] += c[i] >> 29;  // (the low 29 bits are zero)
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        r.l[k - 9] = (int32_t)((uint32_t)c[k] & (uint32_t)MASK29);
        c[k + 1] += c[k] >> 29;
    }
    r.l[8] = (int32_t)c[17];
    return r;
}

template <int FORM>
__global__ void chain(F9* out, int iters) {
    F9 a = Fq9Params::one(), b = Fq9Params::k256();
    a.l[0] += threadIdx.x;
    for (int i = 0; i < iters; i++) a = FORM ? mul_cols<Fq9Params>(a, b) : Fq9::mul(a, b);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

template <int FORM>
__global__ void check(F9* out) {
    F9 a = Fq9Params::one(), b = Fq9Params::k256();
    a.l[0] += threadIdx.x * 977 + 5;
    b.l[3] -= threadIdx.x;
    for (int i = 0; i < 50; i++) {
        const F9 t = FORM ? mul_cols<Fq9Params>(a, b) : Fq9::mul(a, b);
        b = a;
        a = t;
    }
    out[threadIdx.x] = Fq9::canon(a);
}

template <int FORM>
static void run(F9* d, hipEvent_t e0, hipEvent_t e1, int blocks, int threads, const char* what) {
    const int iters = 20000;
    hipLaunchKernelGGL(chain<FORM>, dim3(blocks), dim3(threads), 0, 0, d, 100);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(chain<FORM>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %4d workgroup(s) x %4d threads (%d wave(s) per SIMD): %.3f us per dependent product, %.1f G products/s\n", what, blocks,
           threads, (threads + 255) / 256, ms * 1e3 / iters, (double)blocks * threads * iters / (ms * 1e-3) / 1e9);
}

int main() {
    F9* d;
    if (hipMalloc(&d, sizeof(F9) * 1024 * 1024) != hipSuccess) return 1;
    F9 h0[64], h1[64];
    hipLaunchKernelGGL(check<0>, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h0, d, sizeof(h0), hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(check<1>, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h1, d, sizeof(h1), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++)
        for (int k = 0; k < 9; k++) bad += h0[i].l[k] != h1[i].l[k];
    printf("column form == chain form on 64 lanes x 50 products: %s\n", bad ? "NO" : "yes");
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int threads : {64, 256, 512, 768, 1024}) {
        run<0>(d, e0, e1, 1, threads, "one accumulator  ");
        run<1>(d, e0, e1, 1, threads, "column accumulators");
    }
    for (int threads : {256, 512, 768}) {
        run<0>(d, e0, e1, 256, threads, "one accumulator  , whole chip");
        run<1>(d, e0, e1, 256, threads, "column accumulators, whole chip");
    }
    for (int rep = 0; rep < 3; rep++) {  // (one wave per SIMD on every CU, alternating: is the difference above the order of the runs?)
        run<1>(d, e0, e1, 256, 256, "column accumulators, whole chip");
        run<0>(d, e0, e1, 256, 256, "one accumulator  , whole chip");
    }
    return bad;
}
