// Does hipExtAnyOrderLaunch let two kernels of ONE stream run side by side on this part?  (tools only)
// build: hipcc --offload-arch=gfx950 -O3 -o ab_libs/any_order_probe tools/any_order_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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
    hipStream_t st;
    hipStreamCreate(&st);
    const long long cyc = 20000000;   // 100 MHz counter ticks?  measured below
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, cyc / 100, sink);   // a short kernel in front (the barrier's anchor)
            hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, cyc, sink);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, cyc, sink);
            else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, sink);
            hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, cyc / 100, sink);
            hipStreamSynchronize(st);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("%s: %.3f ms\n", mode == 0 ? "two kernels, in order" : (mode == 1 ? "second with hipExtAnyOrderLaunch" : "one kernel"), ms);
        }
    }
    return 0;
}
