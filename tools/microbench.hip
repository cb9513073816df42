// Instruction-throughput probe for the integer field arithmetic on gfx950 (not part of the library).
// Prints wave-instruction issue cost (cycles per instruction per SIMD, assuming 2.4 GHz) for the ops a
// 254-bit Montgomery product is built from, and the achieved Fr::mul rate at several occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../0g-halo2_amd/csrc/curve.h"
using namespace zg;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2000;

template <int OP>
__global__ void probe(uint32_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t acc[8];
    double d[8];
    for (int i = 0; i < 8; i++) { acc[i] = a + i; d[i] = (double)(a + i); }
    double db = (double)b * 1e-9 + 1.0;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
            if (OP == 1) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[i] = lo; }
            if (OP == 2) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[i] = lo; }
            if (OP == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
            if (OP == 4) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(lo) : "v"(b) : "vcc"); acc[i] = lo; }
            if (OP == 5) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(db));
            if (OP == 6) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(lo) : "v"(b)); acc[i] = lo; }
            if (OP == 7) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mov_b32 %0, %1" : "+v"(lo) : "v"(b)); acc[i] = lo; }
            if (OP == 8) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[i] = lo; }
            if (OP == 10) { long long sa = (long long)acc[i]; asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(sa) : "v"(a), "v"(b) : "vcc"); acc[i] = (uint64_t)sa; }
            if (OP == 9) { uint32_t lo = (uint32_t)acc[i]; asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo) : "v"(b) : "vcc"); acc[i] = lo; }
        }
    }
    uint64_t s = 0;
    double ds = 0;
    for (int i = 0; i < 8; i++) { s += acc[i]; ds += d[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s + (uint32_t)ds;
}

__global__ void modmul_probe(Fe* out, Fe x, Fe y, int iters) {
    Fe a = x, b = y;
    a.l[0] ^= threadIdx.x;
    for (int i = 0; i < iters; i++) {
        a = Fr::mul(a, b);
        b = Fr::mul(b, a);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = Fr::add(a, b);
}

__global__ void madd_probe(XYZZ* out, Affine p, int iters) {
    XYZZ acc = xyzz_from_affine(p);
    acc = xyzz_dbl(acc);
    Affine q = p;
    for (int i = 0; i < iters; i++) {
        q.x.l[0] ^= 0;  // same point; acc changes every step so the generic branch is taken
        acc = xyzz_madd(acc, q);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
static int run(const char* name, uint32_t* d_out) {
    const int blocks = 256 * 4, threads = 256;  // 16 waves per CU = 4 per SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 7u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 9u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double wave_instr = (double)blocks * (threads / 64) * ITERS * 8;
    double per_simd_per_s = wave_instr / (ms * 1e-3) / (256.0 * 4);
    printf("%-22s %8.3f ms  %7.2f cycles/wave-instr/SIMD @2.4GHz  (%.1f Gop/s lanes)\n", name, ms,
           2.4e9 / per_simd_per_s, wave_instr * 64 / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    uint32_t* d_out;
    CK(hipMalloc(&d_out, 256 * 4 * 256 * 64 * sizeof(uint32_t)));
    run<0>("v_mad_u64_u32", d_out);
    run<10>("v_mad_i64_i32", d_out);
    run<1>("v_mul_lo_u32", d_out);
    run<2>("v_mul_hi_u32", d_out);
    run<3>("v_lshl_add_u64", d_out);
    run<4>("v_add_co_u32", d_out);
    run<9>("v_addc_co_u32", d_out);
    run<5>("v_fma_f64", d_out);
    run<6>("v_mad_u32_u24", d_out);
    run<8>("v_mul_hi_u32_u24", d_out);
    run<7>("v_mov_b32", d_out);

    Fe* d_fe;
    CK(hipMalloc(&d_fe, (size_t)256 * 32 * 256 * sizeof(Fe)));
    Fe x = fr_root_of_unity(), y = fr_delta();
    const int iters = 500;
    for (int wg_per_cu : {1, 2, 4, 8}) {
        int blocks = 256 * wg_per_cu;
        hipLaunchKernelGGL(modmul_probe, dim3(blocks), dim3(256), 0, 0, d_fe, x, y, 10);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(modmul_probe, dim3(blocks), dim3(256), 0, 0, d_fe, x, y, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double muls = (double)blocks * 256 * iters * 2;
        printf("Fr::mul  %d wg/CU (%2d waves/SIMD): %8.3f ms  %.1f Gmodmul/s  (one dependent mul = %.0f ns)\n",
               wg_per_cu, wg_per_cu, ms, muls / (ms * 1e-3) / 1e9, ms * 1e6 / (iters * 2));
    }
    XYZZ* d_p;
    CK(hipMalloc(&d_p, (size_t)256 * 8 * 256 * sizeof(XYZZ)));
    Affine g; g.x = Fq::from_u64(1); g.y = Fq::from_u64(2);
    for (int wg_per_cu : {1, 2, 4}) {
        int blocks = 256 * wg_per_cu;
        hipLaunchKernelGGL(madd_probe, dim3(blocks), dim3(256), 0, 0, d_p, g, 4);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(madd_probe, dim3(blocks), dim3(256), 0, 0, d_p, g, 200);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double adds = (double)blocks * 256 * 200;
        printf("xyzz_madd %d wg/CU: %8.3f ms  %.2f Gadd/s  (one dependent madd = %.2f us)\n", wg_per_cu, ms,
               adds / (ms * 1e-3) / 1e9, ms * 1e3 / 200);
    }
    return 0;
}
