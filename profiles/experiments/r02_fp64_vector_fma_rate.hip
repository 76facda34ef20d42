// FP64 FMA issue-rate calibration: NACC independent accumulators per lane, W waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int NACC>
__global__ __launch_bounds__(256) void fma_kernel(double *out, int iters, double a0, double b0) {
    double acc[NACC];
    double a[6], b[6];
    for (int i = 0; i < 6; i++) { a[i] = a0 + threadIdx.x * 1e-9 + i; b[i] = b0 + i * 0.5; }
    for (int i = 0; i < NACC; i++) acc[i] = i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_fma(a[i % 6], b[(i / 6) % 6], acc[i]);
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int lds_pad) {
    double *out; hipMalloc(&out, (size_t) blocks * 256 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *) fma_kernel<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fma_kernel<NACC>, dim3(blocks), dim3(256), lds_pad, 0, out, iters, 1.0, 2.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * NACC * iters * 256.0 * blocks;
    printf("NACC %d blocks %d ldspad %d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks, lds_pad, ms, flop / ms / 1e9);
    hipFree(out);
}
int main() {
    // lds_pad limits workgroups per CU: 160 KB -> 1 (1 wave/SIMD), 80 KB -> 2, 40 KB -> 4, 0 -> 8
    for (int pad : {160 * 1024 - 64, 80 * 1024 - 64, 40 * 1024 - 64, 0}) {
        run<36>(256 * 8 * 4, pad);
        run<12>(256 * 8 * 4, pad);
    }
    return 0;
}
