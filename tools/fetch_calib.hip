// Calibration of rocprofv3's FETCH_SIZE on gfx950 for THIS library's access shapes (MI355X_MICROARCH.md, HBM section:
// the counter reports half the bytes of 16-B-per-lane coalesced streaming reads; "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Three kernels over a 2 GiB table (beyond L2 and the
// 256 MiB Infinity Cache), each reading a known number of bytes exactly once:
//   stream16   every lane 16 B, consecutive lanes consecutive addresses (the NTT / evaluate_h column loads)
//   gather64   every lane ONE random 64-byte record as four dwordx4 loads (msm_accumulate's table points: x, y)
//   gather32   every lane ONE random 32-byte record as two dwordx4 loads (a lone field element)
//     hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/fetch_calib.bin
//     rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./tools/fetch_calib.bin
// Prints the true byte counts; tools/install_r03.py divides the counter by them.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void stream16(const uint4* __restrict__ t, size_t n16, uint4* __restrict__ sink) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = t[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u) sink[0] = acc;
}

template <int WORDS>  // record = WORDS x 16 B
__global__ void gather(const uint4* __restrict__ t, uint32_t log_records, uint4* __restrict__ sink) {
    const uint32_t records = 1u << log_records;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < records; i += gridDim.x * blockDim.x) {
        // an odd multiplier is a bijection on [0, 2^log_records): every record is read exactly once, and neighbouring
        // lanes land ~2.6 * 10^9 records apart (mod the table): no two lanes of a wave share a 128-byte line
        const uint32_t r = (i * 2654435761u + 12345u) & (records - 1);
        const uint4* p = t + (size_t)r * WORDS;
#pragma unroll
        for (int w = 0; w < WORDS; w++) {
            const uint4 v = p[w];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    if (acc.x == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    uint4 *t = nullptr, *sink = nullptr;
    CK(hipMalloc(&t, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(t, 1, bytes));
    CK(hipDeviceSynchronize());
    const dim3 grid(256 * 8), block(256);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(stream16, grid, block, 0, 0, t, bytes / 16, sink);
        hipLaunchKernelGGL(gather<4>, grid, block, 0, 0, t, 25u, sink);  // 2^25 records of 64 B = 2 GiB
        hipLaunchKernelGGL(gather<2>, grid, block, 0, 0, t, 26u, sink);  // 2^26 records of 32 B = 2 GiB
        CK(hipDeviceSynchronize());
    }
    printf("{\"true_bytes\": {\"stream16\": %zu, \"gather<4>\": %zu, \"gather<2>\": %zu}}\n", bytes, bytes, bytes);
    return 0;
}
