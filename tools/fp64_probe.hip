// Gated experiment: a 254-bit Montgomery product on the FP64 pipe -- five 52-bit limbs held as
// doubles, every 104-bit partial product split by two fused multiply-adds (Emmart & Weems' "dual FMA": hi = fma(a, b, C1)
// pins the exponent so that the mantissa holds the product's high half, lo = fma(a, b, C2 - hi) is the exact low half),
// the halves summed as 64-bit integers on the doubles' bit patterns -- against the production nine-limb product
// (field9.h: 162 v_mad_i64_i32 + ~70).  Gate: >= 1.2 x products/s at 3 waves per SIMD.
//
//   per partial product: 2 v_fma_f64 + 1 v_add_f64 (C2 - hi) + 2 64-bit integer additions        = 5 instructions
//   50 partial products (25 a*b + 25 m*q) + 5 x (m_i = low52(T * q')) + column carries + repacking
// so ~300 instructions against ~232: the FP64 form can only win if v_fma_f64 issues faster than v_mad_i64_i32.
//
//   hipcc -O3 --offload-arch=gfx950 -I0g-halo2_amd/csrc -Iinclude -o tools/fp64_probe.bin tools/fp64_probe.hip
// Prints a JSON object: rates at 1..8 workgroups per CU for both forms, and one product (inputs, output) that
// tools/fp64_probe_check.py verifies with Python integers (a * b * 2^-260 mod q).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "field9.h"

using namespace zg;

struct D5 {
    double l[5];  // integer-valued limbs, |l| < 2^52; value = sum l[i] 2^(52 i), a residue mod q (not canonical)
};

// q (BN254 base field) in 52-bit limbs, and -q^-1 mod 2^52
__device__ __constant__ double Q5[5];
__device__ __constant__ double QINV52;

// The kernels run with the FP64 rounding mode set to round-toward-zero (MODE.FP_ROUND[3:2] = 3), as Emmart's scheme
// asks: fma(a, b, 2^104) is then 2^104 + floor(ab / 2^52) * 2^52 exactly, and every quantity below is unsigned.
constexpr double C1 = 0x1p104;              // exponent pin of the high half: ulp 2^52
constexpr double C2 = 0x1p104 + 0x1p52;     // the low half comes out as 2^52 + (ab mod 2^52): exponent pinned again
constexpr int64_t HI_BIAS = (int64_t)0x467ull << 52;  // bit pattern of 2^104
constexpr int64_t LO_BIAS = (int64_t)0x433ull << 52;  // bit pattern of 2^52

__device__ __forceinline__ void round_toward_zero_f64() {
    // (inline asm: given the builtin, the compiler's mode-register pass puts the default mode back before the first
    //  FP64 instruction)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
}

// a * b (0 <= a, b < 2^52, integers): hi = floor(ab / 2^52), lo = ab mod 2^52; both as the raw bit patterns of doubles
// with pinned exponents (the biases are taken off once per column, not per product)
__device__ __forceinline__ void mul_hl(double a, double b, int64_t& hi, int64_t& lo) {
    const double h = __builtin_fma(a, b, C1);
    const double l = __builtin_fma(a, b, C2 - h);
    hi += __double_as_longlong(h);
    lo += __double_as_longlong(l);
}

// integer in [0, 2^52) held in an int64 -> the same value as a double (exact)
__device__ __forceinline__ double to_double52(int64_t v) {
    return __longlong_as_double(v | LO_BIAS) - 0x1p52;
}

// a * b * 2^-260 mod q.  Operands: limbs in [0, 2^52), values below 2^256.  Result: limbs in [0, 2^52), value in
// [0, 2^252 + q): no final subtraction (the radix 2^260 leaves the room).
__device__ __forceinline__ D5 mul5(const D5& a, const D5& b) {
    int64_t col[11];
#pragma unroll
    for (int c = 0; c < 11; c++) col[c] = 0;
    // ---- schoolbook a * b: column c takes the low halves of i + j = c and the high halves of i + j = c - 1
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) mul_hl(a.l[i], b.l[j], col[i + j + 1], col[i + j]);
    // take the exponent patterns off: column c holds n_lo(c) low halves and n_hi(c) high halves
#pragma unroll
    for (int c = 0; c < 10; c++) {
        const int n_lo = c < 5 ? c + 1 : 9 - c;  // pairs with i + j = c   (0 for c = 9)
        const int n_hi = c == 0 ? 0 : (c - 1 < 5 ? c : 10 - c);  // pairs with i + j = c - 1
        col[c] -= (int64_t)((uint64_t)n_lo * (uint64_t)LO_BIAS + (uint64_t)n_hi * (uint64_t)HI_BIAS);
    }
    // ---- Montgomery reduction, one 52-bit word at a time: m = T * (-q^-1) mod 2^52 (signed), T += m * q
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int64_t T = col[i] + carry;
        const int64_t u = T & (((int64_t)1 << 52) - 1);            // T mod 2^52, in [0, 2^52)
        const double ud = to_double52(u);
        const double h = __builtin_fma(ud, QINV52, C1);
        const double m = __builtin_fma(ud, QINV52, C2 - h) - 0x1p52;  // u * q' mod 2^52, in [0, 2^52)
        int64_t lo0 = 0, hi0 = 0;
        mul_hl(m, Q5[0], hi0, lo0);
        // T + lo(m q_0) == 0 (mod 2^52): only its carry survives
        carry = (T + (lo0 - LO_BIAS)) >> 52;
        col[i + 1] += hi0 - HI_BIAS;
#pragma unroll
        for (int j = 1; j < 5; j++) {
            int64_t lo = 0, hi = 0;
            mul_hl(m, Q5[j], hi, lo);
            col[i + j] += lo - LO_BIAS;
            col[i + j + 1] += hi - HI_BIAS;
        }
    }
    // ---- result = columns 5..9 (+ the last carry), carried into 52-bit limbs, back to doubles
    D5 r;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int64_t T = col[5 + j] + carry;
        r.l[j] = to_double52(T & (((int64_t)1 << 52) - 1));
        carry = T >> 52;
    }
    r.l[4] = to_double52(col[9] + carry);  // (< 2^46)
    return r;
}

__global__ void mul5_probe(D5* out, D5 x, D5 y, int iters) {
    round_toward_zero_f64();
    D5 a = x, b = y;
    a.l[0] += (double)(threadIdx.x & 1);
    for (int i = 0; i < iters; i++) {
        a = mul5(a, b);
        b = mul5(b, a);
    }
    if (a.l[0] == 12345.0 && b.l[1] == 54321.0) out[blockIdx.x * blockDim.x + threadIdx.x] = a;  // keep alive
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = a;
        out[1] = b;
    }
}

__global__ void mul9_probe(F9* out, F9 x, F9 y, int iters) {
    F9 a = x, b = y;
    a.l[0] ^= (int32_t)(threadIdx.x & 1);
    for (int i = 0; i < iters; i++) {
        a = Fq9::mul(a, b);
        b = Fq9::mul(b, a);
    }
    if (a.l[0] == 0x12345 && b.l[1] == 0x54321) out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int main() {
    // q in 52-bit limbs from its 32-bit limbs
    uint32_t p32[8];
    for (int i = 0; i < 8; i++) p32[i] = FqParams::p(i);
    auto bits = [&](int lo, int n) {
        uint64_t v = 0;
        for (int b = 0; b < n; b++) {
            const int pos = lo + b;
            if (pos < 256) v |= (uint64_t)((p32[pos / 32] >> (pos % 32)) & 1) << b;
        }
        return v;
    };
    double q5[5];
    uint64_t q5i[5];
    for (int i = 0; i < 5; i++) {
        q5i[i] = bits(52 * i, 52);
        q5[i] = (double)q5i[i];
    }
    uint64_t inv = 1;  // q^-1 mod 2^64 by Newton, then -q^-1 mod 2^52
    for (int i = 0; i < 7; i++) inv *= 2 - q5i[0] * inv;
    const uint64_t qinv = (0 - inv) & ((1ull << 52) - 1);
    const double qinv_d = (double)qinv;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(Q5), q5, sizeof(q5));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(QINV52), &qinv_d, sizeof(double));
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    D5* d5;
    F9* d9;
    (void)hipMalloc(&d5, sizeof(D5) * 256 * cus * 8);
    (void)hipMalloc(&d9, sizeof(F9) * 256 * cus * 8);
    D5 x5, y5;
    uint64_t xi[5], yi[5];
    for (int i = 0; i < 5; i++) {
        xi[i] = (0x123456789abcdull * (i + 1) + 0x1111) & ((1ull << 52) - 1);
        yi[i] = (0xfedcba9876543ull * (i + 3) + 0x2222) & ((1ull << 52) - 1);
    }
    xi[4] &= (1ull << 44) - 1;  // values below 2^252
    yi[4] &= (1ull << 44) - 1;
    for (int i = 0; i < 5; i++) {
        x5.l[i] = (double)xi[i];
        y5.l[i] = (double)yi[i];
    }
    F9 x9, y9;
    for (int i = 0; i < 9; i++) {
        x9.l[i] = (int32_t)((0x1234567u * (i + 1)) & (uint32_t)MASK29);
        y9.l[i] = (int32_t)((0x7654321u * (i + 3)) & (uint32_t)MASK29);
    }
    x9.l[8] &= 0xffff;
    y9.l[8] &= 0xffff;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 2000;
    printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"rates\": [", prop.gcnArchName, cus, iters);
    bool first = true;
    for (int wg : {1, 2, 3, 4, 6, 8}) {
        const int blocks = cus * wg;
        float ms5, ms9;
        hipLaunchKernelGGL(mul5_probe, dim3(blocks), dim3(256), 0, 0, d5, x5, y5, 10);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(mul5_probe, dim3(blocks), dim3(256), 0, 0, d5, x5, y5, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms5, e0, e1);
        hipLaunchKernelGGL(mul9_probe, dim3(blocks), dim3(256), 0, 0, d9, x9, y9, 10);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(mul9_probe, dim3(blocks), dim3(256), 0, 0, d9, x9, y9, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms9, e0, e1);
        const double n = (double)blocks * 256 * iters * 2;
        printf("%s\n  {\"waves_per_simd\": %d, \"fp64_5x52_ms\": %.3f, \"fp64_5x52_Gmul_s\": %.1f, \"int_9x29_ms\": %.3f, "
               "\"int_9x29_Gmul_s\": %.1f, \"ratio\": %.3f}",
               first ? "" : ",", wg, ms5, n / ms5 / 1e6, ms9, n / ms9 / 1e6, ms9 / ms5);
        first = false;
    }
    printf("],\n");
    // one product for the host check: a = x * y * 2^-260, b = y * a * 2^-260 (mod q)
    hipLaunchKernelGGL(mul5_probe, dim3(1), dim3(64), 0, 0, d5, x5, y5, 1);
    D5 h[2];
    (void)hipMemcpy(h, d5, sizeof(h), hipMemcpyDeviceToHost);
    auto dump = [](const char* name, const D5& v, const char* end) {
        printf(" \"%s\": [", name);
        for (int i = 0; i < 5; i++) printf("%s%.0f", i ? ", " : "", v.l[i]);
        printf("]%s\n", end);
    };
    D5 xin = x5;
    xin.l[0] += 0.0;  // (lane 0: threadIdx & 1 == 0)
    dump("x", xin, ",");
    dump("y", y5, ",");
    dump("a", h[0], ",");
    dump("b", h[1], "");
    printf("}\n");
    return 0;
}
