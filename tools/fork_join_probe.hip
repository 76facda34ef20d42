// What a fork + join between two streams of one process costs on this part, against the same kernels in one stream (tools only)
// build: hipcc --offload-arch=gfx950 -O3 -o ab_libs/fork_join_probe tools/fork_join_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int *sink) {
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) {
    }
    if (cycles < 0) *sink = 1;
}
int main() {
    int *sink;
    hipMalloc(&sink, 4);
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipEvent_t e1, e2;
    hipEventCreateWithFlags(&e1, hipEventDisableTiming);
    hipEventCreateWithFlags(&e2, hipEventDisableTiming);
    const int N = 200;
    for (long long body : {1000LL, 10000LL}) {   // ticks of the 100 MHz counter: 10 us, 100 us
        for (int mode = 0; mode < 2; mode++) {
            for (int rep = 0; rep < 3; rep++) {
                hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; i++) {
                    hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, 100LL, sink);
                    if (mode == 0) {
                        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, body, sink);
                        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, body, sink);
                    } else {
                        hipEventRecord(e1, a);
                        hipStreamWaitEvent(b, e1, 0);
                        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, b, body, sink);
                        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, body, sink);
                        hipEventRecord(e2, b);
                        hipStreamWaitEvent(a, e2, 0);
                    }
                    hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, 100LL, sink);
                }
                hipDeviceSynchronize();
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
                printf("bodies of %lld ticks, %s: %.1f us per round\n", body, mode == 0 ? "one stream" : "fork + join", us);
            }
        }
    }
    return 0;
}
