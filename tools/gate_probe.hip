// Probe for a lone proof's round trips: what does it cost to have the NEXT phase's work queued behind a stream-ordered wait
// on a host-written flag (hipStreamWaitValue32) instead of queueing it after the host has the challenge?
//   hipcc -O3 --offload-arch=gfx950 tools/gate_probe.hip -o tools/gate_probe.bin && ./tools/gate_probe.bin
// Measures, per variant, the time from "the host has the data" to "a dependent kernel's result is back on the host":
//   A  copy (2.9 KB H2D, pinned) + kernel queued AFTER the data is ready                     (what the prover does)
//   B  wait-value + copy + kernel queued BEFORE, the host then writes the staging slot and the flag   (the gate)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);        \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

__global__ void consume(const uint32_t* consts, uint32_t* out_host) {
    // reads what the copy delivered, writes straight into mapped host memory (as the prover's last kernels do)
    if (threadIdx.x == 0) {
        out_host[1] = consts[0];
        __threadfence_system();
        out_host[0] = consts[1];
    }
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    uint32_t *d_consts, *h_stage, *h_out, *h_out_dev, *sig = nullptr;
    CK(hipMalloc(&d_consts, 4096));
    CK(hipHostMalloc(&h_stage, 4096, hipHostMallocDefault));
    CK(hipHostMalloc(&h_out, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&h_out_dev, h_out, 0));
    hipError_t es = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(signal) -> %s, ptr %p\n", hipGetErrorString(es), (void*)sig);
    using clk = std::chrono::steady_clock;
    const int reps = 200;
    double sum_a = 0, sum_b = 0;
    volatile uint32_t* vout = h_out;
    for (int r = 1; r <= reps; r++) {
        // ---- A
        vout[0] = 0;
        CK(hipStreamSynchronize(st));
        auto t0 = clk::now();
        h_stage[0] = 7;
        h_stage[1] = (uint32_t)r;
        CK(hipMemcpyAsync(d_consts, h_stage, 2944, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(consume, dim3(1), dim3(64), 0, st, d_consts, h_out_dev);
        while (vout[0] != (uint32_t)r) {
        }
        sum_a += std::chrono::duration<double, std::micro>(clk::now() - t0).count();
    }
    printf("A  queue after the data is ready:            %.1f us per round trip\n", sum_a / reps);
    if (can && es == hipSuccess && sig) {
        volatile uint32_t* vsig = sig;
        *vsig = 0;
        CK(hipStreamSynchronize(st));
        for (int r = 1; r <= reps; r++) {
            vout[0] = 0;
            CK(hipStreamSynchronize(st));
            // queued ahead of the data
            CK(hipStreamWaitValue32(st, sig, (uint32_t)r, hipStreamWaitValueGte, 0xffffffffu));
            CK(hipMemcpyAsync(d_consts, h_stage, 2944, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(consume, dim3(1), dim3(64), 0, st, d_consts, h_out_dev);
            // (the host "computes the challenge" for a while: the queue has long been drained up to the gate)
            auto spin = clk::now();
            while (std::chrono::duration<double, std::micro>(clk::now() - spin).count() < 50.0) {
            }
            auto t0 = clk::now();
            h_stage[0] = 7;
            h_stage[1] = (uint32_t)r;
            __sync_synchronize();
            *vsig = (uint32_t)r;
            int budget = 200000000;
            while (vout[0] != (uint32_t)r && --budget > 0) {
            }
            if (budget <= 0) {
                fprintf(stderr, "B: the gate never opened (round %d)\n", r);
                return 2;
            }
            sum_b += std::chrono::duration<double, std::micro>(clk::now() - t0).count();
        }
        printf("B  queued behind a host-written gate:         %.1f us per round trip\n", sum_b / reps);
    } else {
        printf("B  not available on this device\n");
    }
    return 0;
}
