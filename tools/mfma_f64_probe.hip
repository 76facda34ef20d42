// FP64 matrix-core probe for gfx950 (tools only, not part of libecal.so):
//   * lane layouts of v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64, checked against a host product
//   * issue cost of each against v_fma_f64, one wave per SIMD
//   * whether a wave of v_fma_f64 and a wave of MFMAs on the SAME SIMD run beside each other
// build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_f64_probe tools/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

// ---- layouts ----
__global__ void layout16(const double *A, const double *B, double *D) {   // A[16][4], B[4][16] row-major, D[16][16]
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];       // A[i][k]: i = lane & 15, k = lane >> 4
    const double b = B[(l >> 4) * 16 + (l & 15)];      // B[k][j]: j = lane & 15, k = lane >> 4
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];   // row = (lane >> 4) + 4 r, col = lane & 15
}
// four 4x4 blocks: hypothesis A[blk][i][k]: i = lane & 3, k = (lane >> 2) & 3 ?  The probe writes out the raw lanes and the host
// finds which (i, k, blk) assignment reproduces them.
__global__ void layout4(const double *a_in, const double *b_in, double *d_out) {
    const int l = threadIdx.x;
    double c = 0;
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(a_in[l], b_in[l], c, 0, 0, 0);
    d_out[l] = c;
}

// ---- timing ----
template <int MODE>   // 0: v_fma_f64 x 32 per round; 1: mfma 16x16x4 x 8 per round; 2: mfma 4x4x4 x 16 per round
__device__ __forceinline__ double work(int rounds, double seed) {
    if (MODE == 0) {
        double acc[32];
#pragma unroll
        for (int i = 0; i < 32; i++) acc[i] = seed + i;
        const double x = seed * 0.5, y = seed * 0.25;
        for (int r = 0; r < rounds; r++) {
#pragma unroll
            for (int i = 0; i < 32; i++) acc[i] = __builtin_fma(acc[i], x, y);
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < 32; i++) s += acc[i];
        return s;
    } else if (MODE == 1) {
        d4 c[8];
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = d4{seed, seed, seed, seed};
        const double a = seed * 0.5, b = seed * 0.25;
        for (int r = 0; r < rounds; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
        return s;
    } else {
        double c[16];
#pragma unroll
        for (int i = 0; i < 16; i++) c[i] = seed + i;
        const double a = seed * 0.5, b = seed * 0.25;
        for (int r = 0; r < rounds; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) s += c[i];
        return s;
    }
}

// waves [0, 4) of a 512-thread workgroup run MODE_A, waves [4, 8) MODE_B (-1: that half idles): one of each on every SIMD.
// out[2 * blockIdx + half] = clocks of wave 0 of that half.
template <int MODE_A, int MODE_B>
__global__ __launch_bounds__(512, 1) void beside(int rounds_a, int rounds_b, double seed, long long *clk, double *sink) {
    const int wave = threadIdx.x >> 6;
    const bool first = wave < 4;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    double s = 0;
    if (first) {
        if (MODE_A >= 0) s = work<MODE_A < 0 ? 0 : MODE_A>(rounds_a, seed);
    } else {
        if (MODE_B >= 0) s = work<MODE_B < 0 ? 0 : MODE_B>(rounds_b, seed);
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 255) == 0) clk[2 * blockIdx.x + (first ? 0 : 1)] = t1 - t0;
    if (s == 12345.678) sink[0] = s;
}

// one wave issuing both: 16 v_fma_f64 and 4 mfma 16x16x4 per round, interleaved by the compiler's own schedule
__global__ __launch_bounds__(256, 1) void same_wave(int rounds, double seed, long long *clk, double *sink) {
    double acc[16];
    d4 c[4];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = seed + i;
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = d4{seed, seed, seed, seed};
    const double x = seed * 0.5, y = seed * 0.25;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; k++) acc[4 * i + k] = __builtin_fma(acc[4 * i + k], x, y);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if ((threadIdx.x & 255) == 0) clk[blockIdx.x] = t1 - t0;
    if (s == 12345.678) sink[0] = s;
}

template <int MA, int MB>
static void run_beside(const char *what, int ra, int rb, int per_a, int per_b) {
    long long *clk;
    double *sink;
    const int nb = 256;
    CK(hipMalloc(&clk, 2 * nb * sizeof(long long)));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(clk, 0, 2 * nb * sizeof(long long)));
    for (int rep = 0; rep < 2; rep++) beside<MA, MB><<<nb, 512>>>(ra, rb, 1.0 + 1e-9, clk, sink);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(2 * nb);
    CK(hipMemcpy(h.data(), clk, 2 * nb * sizeof(long long), hipMemcpyDeviceToHost));
    double sa = 0, sb = 0;
    for (int i = 0; i < nb; i++) {
        sa += (double) h[2 * i];
        sb += (double) h[2 * i + 1];
    }
    sa /= nb;
    sb /= nb;
    printf("%-58s", what);
    if (MA >= 0) printf(" first half %9.0f clk = %6.2f per instruction", sa, sa / ((double) ra * per_a));
    if (MB >= 0) printf(" | second half %9.0f clk = %6.2f per instruction", sb, sb / ((double) rb * per_b));
    printf("\n");
    CK(hipFree(clk));
    CK(hipFree(sink));
}

int main() {
    // layouts
    {
        std::vector<double> A(64), B(64), D(256), ref(256, 0.0);
        for (int i = 0; i < 64; i++) {
            A[i] = (double) ((i * 37 + 11) % 23) - 7.0;
            B[i] = (double) ((i * 53 + 5) % 19) - 4.0;
        }
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++)
                for (int k = 0; k < 4; k++) ref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
        double *dA, *dB, *dD;
        CK(hipMalloc(&dA, 64 * 8));
        CK(hipMalloc(&dB, 64 * 8));
        CK(hipMalloc(&dD, 256 * 8));
        CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
        layout16<<<1, 64>>>(dA, dB, dD);
        CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 256; i++) bad += D[i] != ref[i];
        printf("16x16x4: A[i][k] at lane i + 16 k, B[k][j] at lane j + 16 k, D[(lane >> 4) + 4 r][lane & 15]: %d of 256 entries differ\n", bad);
        // 4x4x4_4b: raw lanes; try the assignments
        std::vector<double> a(64), b(64), d(64);
        for (int i = 0; i < 64; i++) {
            a[i] = (double) ((i * 29 + 3) % 31) - 9.0;
            b[i] = (double) ((i * 41 + 7) % 17) - 5.0;
        }
        CK(hipMemcpy(dA, a.data(), 64 * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, b.data(), 64 * 8, hipMemcpyHostToDevice));
        layout4<<<1, 64>>>(dA, dB, dD);
        CK(hipMemcpy(d.data(), dD, 64 * 8, hipMemcpyDeviceToHost));
        // hypotheses: lane = f(blk, x, y) with the three 2-bit fields in any order, for A (i, k), B (k, j), D (i, j)
        const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
        auto lane_of = [&](const int *p, int blk, int x, int y) {   // fields: 0 = blk, 1 = x, 2 = y placed at bit positions 2*p[.]
            return (blk << (2 * p[0])) | (x << (2 * p[1])) | (y << (2 * p[2]));
        };
        int found = 0;
        for (int pa = 0; pa < 6; pa++)
            for (int pb = 0; pb < 6; pb++)
                for (int pd = 0; pd < 6; pd++) {
                    int ok = 1;
                    for (int blk = 0; blk < 4 && ok; blk++)
                        for (int i = 0; i < 4 && ok; i++)
                            for (int j = 0; j < 4 && ok; j++) {
                                double s = 0;
                                for (int k = 0; k < 4; k++) s += a[lane_of(perms[pa], blk, i, k)] * b[lane_of(perms[pb], blk, k, j)];
                                if (d[lane_of(perms[pd], blk, i, j)] != s) ok = 0;
                            }
                    if (ok) {
                        found++;
                        printf("4x4x4_4b: A[blk][i][k] lane bits (blk,i,k) at 2*(%d,%d,%d); B[blk][k][j] (blk,k,j) at 2*(%d,%d,%d); D[blk][i][j] (blk,i,j) at 2*(%d,%d,%d)\n",
                               perms[pa][0], perms[pa][1], perms[pa][2], perms[pb][0], perms[pb][1], perms[pb][2], perms[pd][0], perms[pd][1], perms[pd][2]);
                    }
                }
        if (!found) {
            printf("4x4x4_4b: no field assignment matches; raw lanes:\n");
            for (int i = 0; i < 64; i++) printf("%g%c", d[i], (i & 15) == 15 ? '\n' : ' ');
        }
    }
    const int R = 4000;
    run_beside<0, -1>("v_fma_f64 alone (one wave per SIMD)", R, 0, 32, 1);
    run_beside<1, -1>("mfma_f64_16x16x4 alone", R, 0, 8, 1);
    run_beside<2, -1>("mfma_f64_4x4x4_4b alone", R, 0, 16, 1);
    run_beside<0, 0>("v_fma_f64 beside v_fma_f64 (two waves per SIMD)", R, R, 32, 32);
    run_beside<1, 1>("mfma 16x16x4 beside mfma 16x16x4", R, R, 8, 8);
    run_beside<2, 2>("mfma 4x4x4 beside mfma 4x4x4", R, R, 16, 16);
    run_beside<0, 1>("v_fma_f64 beside mfma 16x16x4 (about equal alone times)", R, R / 4, 32, 8);
    run_beside<0, 1>("v_fma_f64 beside mfma 16x16x4 (mfma half twice as long)", R, R / 2, 32, 8);
    run_beside<0, 2>("v_fma_f64 beside mfma 4x4x4", R, R / 2, 32, 16);
    {
        long long *clk;
        double *sink;
        CK(hipMalloc(&clk, 256 * sizeof(long long)));
        CK(hipMalloc(&sink, 8));
        for (int rep = 0; rep < 2; rep++) same_wave<<<256, 256>>>(R, 1.0 + 1e-9, clk, sink);
        CK(hipDeviceSynchronize());
        std::vector<long long> h(256);
        CK(hipMemcpy(h.data(), clk, 256 * sizeof(long long), hipMemcpyDeviceToHost));
        double s = 0;
        for (int i = 0; i < 256; i++) s += (double) h[i];
        s /= 256;
        printf("one wave, 4 mfma 16x16x4 + 16 v_fma_f64 per round: %.0f clk = %.1f per round\n", s, s / R);
    }
    return 0;
}
