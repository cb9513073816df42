// Probe: Montgomery product with nine 29-bit limbs (R = 2^261) against the production 8 x 32-bit one.
// With 29-bit limbs a column of 18 partial products fits a 64-bit accumulator (18 * 2^58 < 2^64), so
// every partial product is ONE v_mad_u64_u32 and no carry word is needed (the 32-bit-limb form pays one
// v_addc_co_u32 per product).  Prints the achieved products/s for both and one result for a host check.
//   hipcc -O3 --offload-arch=gfx950 -I0g-halo2_amd/csrc -Iinclude -o tools/mont29_probe tools/mont29_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "field.h"

using namespace zg;

struct F9 { uint32_t l[9]; };

// Fr modulus in 29-bit limbs, -p^-1 mod 2^29
__device__ __constant__ uint32_t P9c[9];
constexpr uint32_t MASK29 = (1u << 29) - 1;

template <int DUMMY>
__device__ __forceinline__ F9 mul9(const F9& a, const F9& b, const uint32_t (&p)[9], uint32_t inv29) {
    uint64_t acc = 0;
    uint32_t m[9];
    F9 r;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j >= 0 && j < 9) acc += (uint64_t)a.l[i] * b.l[j];
        }
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (i < k && j >= 0 && j < 9 && i < 9) acc += (uint64_t)m[i] * p[j];
        }
        if (k < 9) {
            m[k] = ((uint32_t)acc * inv29) & MASK29;
            acc += (uint64_t)m[k] * p[0];
            acc >>= 29;
        } else {
            r.l[k - 9] = (uint32_t)acc & MASK29;
            acc >>= 29;
        }
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

__global__ void mul9_probe(F9* out, F9 x, F9 y, int iters, uint32_t inv29) {
    uint32_t p[9];
#pragma unroll
    for (int i = 0; i < 9; i++) p[i] = P9c[i];
    F9 a = x, b = y;
    a.l[0] ^= threadIdx.x & 1;
    for (int i = 0; i < iters; i++) {
        a = mul9<0>(a, b, p, inv29);
        b = mul9<0>(b, a, p, inv29);
    }
    if (a.l[0] == 0x12345 && b.l[1] == 0x54321) out[blockIdx.x * blockDim.x + threadIdx.x] = a;  // keep alive
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = a; out[1] = b; }
}

__global__ void mul8_probe(Fe* out, Fe x, Fe y, int iters) {
    Fe a = x, b = y;
    a.l[0] ^= threadIdx.x & 1;
    for (int i = 0; i < iters; i++) {
        a = Fr::mul(a, b);
        b = Fr::mul(b, a);
    }
    if (a.l[0] == 0x12345 && b.l[1] == 0x54321) out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int main() {
    const unsigned __int128 one = 1;
    (void)one;
    // modulus limbs (29-bit) from the 32-bit limbs
    uint32_t p32[8];
    for (int i = 0; i < 8; i++) p32[i] = FrParams::p(i);
    auto bits = [&](int lo, int n) {  // n <= 32 bits of p starting at bit lo
        uint64_t v = 0;
        for (int b = 0; b < n; b++) {
            int pos = lo + b;
            if (pos < 256) v |= (uint64_t)((p32[pos / 32] >> (pos % 32)) & 1) << b;
        }
        return (uint32_t)v;
    };
    uint32_t p9[9];
    for (int i = 0; i < 9; i++) p9[i] = bits(29 * i, 29);
    // inv29 = -p^-1 mod 2^29 (Newton)
    uint32_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - p9[0] * inv;
    uint32_t inv29 = (0u - inv) & MASK29;
    hipMemcpyToSymbol(HIP_SYMBOL(P9c), p9, sizeof(p9));
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    F9* d9;
    Fe* d8;
    hipMalloc(&d9, sizeof(F9) * 256 * cus * 8);
    hipMalloc(&d8, sizeof(Fe) * 256 * cus * 8);
    F9 x9, y9;
    Fe x8, y8;
    for (int i = 0; i < 9; i++) { x9.l[i] = (0x1234567u * (i + 1)) & MASK29; y9.l[i] = (0x7654321u * (i + 3)) & MASK29; }
    x9.l[8] &= 0xffff; y9.l[8] &= 0xffff;
    for (int i = 0; i < 8; i++) { x8.l[i] = 0x12345678u * (i + 1); y8.l[i] = 0x87654321u * (i + 3); }
    x8.l[7] &= 0x0fffffff; y8.l[7] &= 0x0fffffff;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int wg = 1; wg <= 8; wg *= 2) {
        int blocks = cus * wg;
        float ms9, ms8;
        hipLaunchKernelGGL(mul9_probe, dim3(blocks), dim3(256), 0, 0, d9, x9, y9, 10, inv29);
        hipEventRecord(e0);
        hipLaunchKernelGGL(mul9_probe, dim3(blocks), dim3(256), 0, 0, d9, x9, y9, iters, inv29);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms9, e0, e1);
        hipLaunchKernelGGL(mul8_probe, dim3(blocks), dim3(256), 0, 0, d8, x8, y8, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(mul8_probe, dim3(blocks), dim3(256), 0, 0, d8, x8, y8, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms8, e0, e1);
        double n = (double)blocks * 256 * iters * 2;
        printf("%d wg/CU: 9x29-bit %8.3f ms %7.1f Gmul/s | 8x32-bit %8.3f ms %7.1f Gmul/s\n", wg, ms9, n / ms9 / 1e6, ms8,
               n / ms8 / 1e6);
    }
    // one product for a host-side check: out[0] = chain value; re-run with iters = 1 so that it is x*y, y*(x*y)
    hipLaunchKernelGGL(mul9_probe, dim3(1), dim3(64), 0, 0, d9, x9, y9, 1, inv29);
    F9 h[2];
    hipMemcpy(h, d9, sizeof(h), hipMemcpyDeviceToHost);
    printf("x9 =");
    for (int i = 0; i < 9; i++) printf(" %u", x9.l[i]);
    printf("\ny9 =");
    for (int i = 0; i < 9; i++) printf(" %u", y9.l[i]);
    printf("\na  =");
    for (int i = 0; i < 9; i++) printf(" %u", h[0].l[i]);
    printf("\nb  =");
    for (int i = 0; i < 9; i++) printf(" %u", h[1].l[i]);
    printf("\n");
    return 0;
}
