#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0) {
    v4d D[NT];
    for (int i = 0; i < NT; i++) D[i] = v4d{0, 0, 0, 0};
    double x0 = a0 + threadIdx.x, x1 = a0 * 2 + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NT; i++) D[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, D[i], 0, 0, 0);
        asm volatile("" : "+v"(x0), "+v"(x1));
    }
    double s = 0;
    for (int i = 0; i < NT; i++) s += D[i][0] + D[i][1] + D[i][2] + D[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NT>
void run(int pad) {
    const int blocks = 256 * 16, iters = 500;
    double *out; (void) hipMalloc(&out, (size_t) blocks * 256 * 8);
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    (void) hipFuncSetAttribute((const void *) k<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void) hipEventRecord(e0);
        hipLaunchKernelGGL(k<NT>, dim3(blocks), dim3(256), pad, 0, out, iters, 1.0);
        (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
        (void) hipEventElapsedTime(&ms, e0, e1);
    }
    const double n_mfma = (double) NT * iters * 4.0 * blocks;          // wave-level MFMAs
    const double cyc = ms * 1e-3 * 2.4e9 * 1024.0 / n_mfma;             // SIMD-cycles per MFMA at 2.4 GHz
    printf("tiles %d ldspad %d: %.3f ms  %.1f TFLOP/s  %.1f cycles/MFMA/SIMD\n", NT, pad, ms, n_mfma * 2048.0 / ms / 1e9, cyc);
    (void) hipFree(out);
}
int main() {
    for (int pad : {160 * 1024 - 64, 80 * 1024 - 64, 0}) { run<6>(pad); run<2>(pad); run<1>(pad); }
    return 0;
}
