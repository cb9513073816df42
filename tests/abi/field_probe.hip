// Test program (not product code): Field::inv of csrc/field.h on inputs no valid caller passes -- non-canonical
// multiples of the modulus -- on the host path ("host") and on the device path ("device").  The binary extended Euclid
// never reaches 1 for them; since round 3 the loops carry an iteration budget, since round 4 an exhausted budget
// returns 0 (it used to return whichever cofactor was current, as if it were the inverse).
// Usage: field_probe host|device   -> prints one line per case and "OK" / "FAIL".
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "../../0g-halo2_amd/csrc/field.h"

using namespace zg;

struct Case { Fe in; Fe out_fr; Fe out_fq; };

__global__ void inv_kernel(Case* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    c[i].out_fr = Fr::inv(c[i].in);
    c[i].out_fq = Fq::inv(c[i].in);
}

template <class P> static Fe multiple_of_p(uint32_t m) {
    Fe r = fe_zero();
    uint64_t carry = 0;
    for (int i = 0; i < 8; i++) {
        carry += (uint64_t)P::p(i) * m;
        r.l[i] = (uint32_t)carry;
        carry >>= 32;
    }
    return r;
}

int main(int argc, char** argv) {
    const bool device = argc > 1 && !strcmp(argv[1], "device");
    // cases: 0, one (Montgomery), a generic residue, p_r, 2 p_r, p_q, 2 p_q, 3 p_q (< 2^256)
    Case cs[8];
    memset(cs, 0, sizeof(cs));
    cs[0].in = fe_zero();
    cs[1].in = Fr::one();
    cs[2].in = Fe{{0x12345678u, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u, 0x11111111u, 0x22222222u, 0x33333333u, 0x04444444u}};
    cs[3].in = multiple_of_p<FrParams>(1);
    cs[4].in = multiple_of_p<FrParams>(2);
    cs[5].in = multiple_of_p<FqParams>(1);
    cs[6].in = multiple_of_p<FqParams>(2);
    cs[7].in = multiple_of_p<FqParams>(3);
    const int n = 8;
    if (device) {
        Case* d = nullptr;
        if (hipMalloc(&d, sizeof(cs)) != hipSuccess) { printf("FAIL hipMalloc\n"); return 2; }
        (void)hipMemcpy(d, cs, sizeof(cs), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(inv_kernel, dim3(1), dim3(64), 0, 0, d, n);
        if (hipDeviceSynchronize() != hipSuccess) { printf("FAIL kernel\n"); return 2; }
        (void)hipMemcpy(cs, d, sizeof(cs), hipMemcpyDeviceToHost);
        (void)hipFree(d);
    } else {
        for (int i = 0; i < n; i++) {
            cs[i].out_fr = Fr::inv(cs[i].in);
            cs[i].out_fq = Fq::inv(cs[i].in);
        }
    }
    bool ok = true;
    auto check = [&](const char* what, bool cond) {
        printf("%-44s %s\n", what, cond ? "ok" : "WRONG");
        ok = ok && cond;
    };
    check("inv(0) == 0 (Fr, Fq)", fe_is_zero(cs[0].out_fr) && fe_is_zero(cs[0].out_fq));
    check("inv(1) == 1 (Fr)", fe_eq(cs[1].out_fr, Fr::one()));
    check("x * inv(x) == 1 (Fr)", fe_eq(Fr::mul(cs[2].in, cs[2].out_fr), Fr::one()));
    check("x * inv(x) == 1 (Fq)", fe_eq(Fq::mul(cs[2].in, cs[2].out_fq), Fq::one()));
    check("Fr::inv(r) terminates with 0", fe_is_zero(cs[3].out_fr));
    check("Fr::inv(2r) terminates with 0", fe_is_zero(cs[4].out_fr));
    check("Fq::inv(q) terminates with 0", fe_is_zero(cs[5].out_fq));
    check("Fq::inv(2q) terminates with 0", fe_is_zero(cs[6].out_fq));
    check("Fq::inv(3q) terminates with 0", fe_is_zero(cs[7].out_fq));
    printf(ok ? "OK\n" : "FAIL\n");
    return ok ? 0 : 1;
}
