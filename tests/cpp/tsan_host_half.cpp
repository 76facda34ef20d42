// ThreadSanitizer build of the host half of the LM step (CPU only: `make -C eventcalib_amd/csrc tsan`, run by
// tests/test_tsan_host_half.py).  What runs under the sanitizer is the product's own code, included as ecal_solver.hip includes
// it — HostPool (epochs, nudges, spin-then-sleep), the partitioned solves on the pool, and the host tasks of the streamed
// evaluation (arrow_streamed_tasks) — with a host thread in the kernel's role: it delivers the accumulation buffer group by
// group into the buffer the tasks read and raises the flags with release stores, as ne_publish_progress does from the GPU.
// Results are checked against the sequential solve; any sanitizer report fails the test (the runner greps stderr).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <random>
#include <thread>
#include <vector>
#include <sched.h>
#include <time.h>

#include "arrow_layout.hpp"
using namespace ecal;
namespace {
#include "arrow_host.hpp"

// J^T J and J^T r of a Jacobian with the solver's sparsity (every row touches the 9 intrinsics and 4 consecutive control
// points), written straight into the accumulation buffer's layout (arrow_layout.hpp)
std::vector<double> random_system(uint32_t n_cp, unsigned seed, int rows_per_span = 5) {
    std::mt19937_64 rng(seed);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> acc(ACC_HEAD + ACC_PER_CP * (size_t) n_cp, 0.0);
    const double colscale[6] = {1.0, 2.0, 0.5, 30.0, 20.0, 10.0};
    acc[0] = 1.0;
    for (uint32_t span = 3; span < n_cp; span++) {
        const uint32_t c0 = span - 3;
        for (int row = 0; row < rows_per_span; row++) {
            double J[33];
            for (int i = 0; i < 9; i++) J[i] = 0.3 * nd(rng);
            for (int i = 0; i < 24; i++) J[9 + i] = colscale[i % 6] * nd(rng);
            const double r = nd(rng);
            for (int i = 0; i < 9; i++) {
                acc[1 + i] += J[i] * r;
                for (int j = i; j < 9; j++) acc[10 + 9 * i + j] += J[i] * J[j];
            }
            for (int a = 0; a < 4; a++) {
                double *rec = acc.data() + ACC_HEAD + ACC_PER_CP * (size_t) (c0 + a);
                for (int k = 0; k < 6; k++) {
                    rec[k] += J[9 + 6 * a + k] * r;
                    for (int j = 0; j < 9; j++) rec[6 + 9 * k + j] += J[9 + 6 * a + k] * J[j];
                }
                for (int d = 0; a + d < 4; d++)
                    for (int ka = 0; ka < 6; ka++)
                        for (int kb = 0; kb < 6; kb++) {
                            if (d == 0 && kb < ka) continue;
                            rec[60 + 36 * d + 6 * ka + kb] += J[9 + 6 * a + ka] * J[9 + 6 * (a + d) + kb];
                        }
            }
        }
    }
    return acc;
}

double max_abs_diff(const std::vector<double> &a, const std::vector<double> &b) {
    double m = 0;
    for (size_t i = 0; i < a.size(); i++) m = std::max(m, std::fabs(a[i] - b[i]));
    return m;
}
double max_abs(const std::vector<double> &a) {
    double m = 0;
    for (double v : a) m = std::max(m, std::fabs(v));
    return m;
}

int fails = 0;
#define CHECK(cond, ...)                         \
    do {                                         \
        if (!(cond)) {                           \
            fails++;                             \
            fprintf(stderr, "FAILED: " __VA_ARGS__); \
            fprintf(stderr, "\n");               \
        }                                        \
    } while (0)

// one streamed evaluation + the finish of the linear solve, the producer on a thread of its own
// order: the order in which the groups finish; stop_after >= 0: the producer delivers that many groups and is then gone
bool streamed_solve(HostPool &pool, uint32_t n_cp, int P, const std::vector<double> &acc_ref, const std::vector<double> &scale, double radius,
                    uint32_t epoch, std::vector<uint32_t> &flag, std::vector<double> &delta, unsigned seed, int stop_after = -1) {
    const size_t nc = 6 * (size_t) n_cp, nt = nc + 9;
    std::vector<uint32_t> first, num;
    arrow_partition_stream(n_cp, P, first, num);
    std::vector<uint32_t> cut(P + 1), init(2 * NE_MAX_GROUPS, 0);
    for (int g = 0; g < P; g++) cut[g] = first[g];
    cut[P] = n_cp;
    for (int g = 0; g < P; g++) init[g] = 1;
    for (int b = 0; b + 1 < P; b++) init[NE_MAX_GROUPS + b] = 2;
    std::vector<double> acc(acc_ref.size(), -7.0);   // (stale values everywhere: a task that reads before its flag is up computes garbage)
    std::atomic<bool> gone{false};
    std::thread producer([&] {
        std::mt19937 rng(seed);
        std::vector<int> order(P);
        for (int g = 0; g < P; g++) order[g] = g;
        for (int g = 0; g + 1 < P; g++)   // mostly in order, neighbours swapped now and then (the kernel's groups finish roughly in order)
            if (rng() % 3u == 0) std::swap(order[g], order[g + 1]);
        std::vector<int> left(P, 2);
        auto copy = [&](uint32_t lo, uint32_t hi) {
            memcpy(acc.data() + ACC_HEAD + ACC_PER_CP * (size_t) lo, acc_ref.data() + ACC_HEAD + ACC_PER_CP * (size_t) lo,
                   ACC_PER_CP * (size_t) (hi - lo) * sizeof(double));
        };
        for (int k = 0; k < P; k++) {
            if (stop_after >= 0 && k >= stop_after) break;
            const int g = order[k];
            if (rng() % 2u) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 300u));
            copy(cut[g], g + 1 < P ? cut[g + 1] - 3 : cut[g + 1]);
            bool sep_l = false, sep_r = false;
            if (g > 0 && --left[g - 1] == 0) sep_l = true, copy(cut[g] - 3, cut[g]);
            if (g + 1 < P && --left[g] == 0) sep_r = true, copy(cut[g + 1] - 3, cut[g + 1]);
            __atomic_store_n(&flag[g], epoch, __ATOMIC_RELEASE);
            if (sep_l) __atomic_store_n(&flag[NE_MAX_GROUPS + g - 1], epoch, __ATOMIC_RELEASE);
            if (sep_r) __atomic_store_n(&flag[NE_MAX_GROUPS + g], epoch, __ATOMIC_RELEASE);
        }
        memcpy(acc.data(), acc_ref.data(), ACC_HEAD * sizeof(double));   // the head: read by the caller after the producer is joined
        gone.store(true);
    });
    ArrowSystem An;
    ArrowWorkspace ws;
    ArrowParts parts;
    std::vector<double> dd(nt);
    unpack_alloc(n_cp, An);
    arrow_parts_setup(An.nc, P, ws, parts, true);
    StreamedSource src;
    src.init = init.data();
    src.cut = cut.data();
    src.flag = flag.data();
    src.epoch = epoch;
    src.producer_gone = [&]() -> bool { return gone.load(); };
    bool reduced_ok = false;
    const bool delivered = arrow_streamed_tasks(&pool, P, src, acc.data(), An, true, radius, scale.data(), dd.data(), 1e-6, 1e32, ws, parts, &reduced_ok,
                                                nullptr, nullptr, std::chrono::steady_clock::now());
    producer.join();   // (= the stream synchronisation of the real evaluation)
    if (!delivered) return false;
    unpack_head(acc.data(), An);
    for (size_t i = nc; i < nt; i++) {
        const double h = An.corner[10 * (i - nc)] * scale[i] * scale[i];
        dd[i] = std::min(std::max(h, 1e-6), 1e32) / radius;
    }
    CHECK(reduced_ok, "streamed: a separator was not eliminated");
    if (!reduced_ok || !arrow_reduced_end(An, scale.data(), dd.data(), parts)) return false;
    std::vector<double> y;
    arrow_parts_backsub(An.nc, y, ws, parts, &pool, P);
    delta.resize(nt);
    for (size_t i = 0; i < nt; i++) delta[i] = y[i] * scale[i];
    return true;
}

}  // namespace

int main(int argc, char **argv) {
    const int scale_down = argc > 1 ? atoi(argv[1]) : 1;   // (a quicker run for interactive use)
    // 1. the pool alone: tasks handed out among parked threads that poll, sleep and are nudged awake
    for (int workers : {0, 1, 3, 7}) {
        const int bad = host_pool_selftest(workers, 400 / scale_down);
        CHECK(bad == 0, "host_pool_selftest(%d workers): %d violations", workers, bad);
    }
    // 2. the partitioned solves on the pool == the sequential routine
    for (uint32_t n_cp : {45u, 131u}) {
        const std::vector<double> acc = random_system(n_cp, n_cp);
        const size_t nt = 9 + 6 * (size_t) n_cp;
        ArrowSystem A;
        unpack(acc.data(), n_cp, A);
        std::vector<double> scale(nt);
        for (size_t i = 0; i < nt - 9; i++) scale[i] = 1.0 / (1.0 + std::sqrt(A.band[i * BW]));
        for (int i = 0; i < 9; i++) scale[nt - 9 + i] = 1.0 / (1.0 + std::sqrt(A.corner[10 * i]));
        for (double radius : {1e4, 3.0}) {
            std::vector<double> seq(nt), d(nt);
            int fail = -1;
            CHECK(arrow_debug_solve_host(n_cp, acc.data(), scale.data(), radius, 1e-6, 1e32, seq.data(), &fail, 0, 0) == 0 && fail == 0, "sequential solve");
            for (int parts : {3, 6})
                for (int mode : {2, 3})
                    for (int workers : {1, 5}) {
                        CHECK(arrow_debug_solve_host(n_cp, acc.data(), scale.data(), radius, 1e-6, 1e32, d.data(), &fail, mode, parts, workers) == 0 && fail == 0,
                              "partitioned solve, n_cp %u parts %d mode %d", n_cp, parts, mode);
                        CHECK(max_abs_diff(d, seq) <= 1e-9 * max_abs(seq), "partitioned != sequential: n_cp %u parts %d mode %d workers %d: %g of %g", n_cp,
                              parts, mode, workers, max_abs_diff(d, seq), max_abs(seq));
                    }
            {   // the step's quadratic forms on the pool == on one thread (fixed ranges, partial sums added in range order)
                HostPool pool(3);
                double g1, h1, g2, h2;
                quad_forms(A, seq, &g1, &h1);
                quad_forms(A, seq, &g2, &h2, false, &pool, 12);
                CHECK(std::fabs(g1 - g2) <= 1e-12 * std::fabs(g1) + 1e-300 && std::fabs(h1 - h2) <= 1e-12 * std::fabs(h1) + 1e-300, "quad_forms on the pool");
            }
            // 3. the streamed evaluation's host tasks against a producer thread, several evaluations on the same flags (epochs)
            for (int workers : {2, 6}) {
                HostPool pool(workers);
                std::vector<uint32_t> flag(2 * NE_MAX_GROUPS, 0);
                uint32_t epoch = 0;
                for (int P : {3, 6}) {
                    if ((uint32_t) (7 * P) > n_cp) continue;
                    for (int rep = 0; rep < 3 / std::min(scale_down, 3) + 0; rep++) {
                        std::vector<double> ds;
                        const bool ok = streamed_solve(pool, n_cp, P, acc, scale, radius, ++epoch, flag, ds, 1000u * n_cp + 10u * P + rep);
                        CHECK(ok, "streamed solve failed: n_cp %u P %d workers %d", n_cp, P, workers);
                        if (ok)
                            CHECK(max_abs_diff(ds, seq) <= 1e-9 * max_abs(seq), "streamed != sequential: n_cp %u P %d workers %d: %g of %g", n_cp, P, workers,
                                  max_abs_diff(ds, seq), max_abs(seq));
                    }
                }
            }
        }
    }
    // 4. a producer that stops half way: every waiter gives up (two seconds), nothing hangs, the caller is told
    {
        const uint32_t n_cp = 131;
        const std::vector<double> acc = random_system(n_cp, 5);
        std::vector<double> scale(9 + 6 * (size_t) n_cp, 1.0), ds;
        HostPool pool(4);
        std::vector<uint32_t> flag(2 * NE_MAX_GROUPS, 0);
        const auto t0 = std::chrono::steady_clock::now();
        const bool ok = streamed_solve(pool, n_cp, 6, acc, scale, 1e4, 1u, flag, ds, 77u, 3);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        CHECK(!ok, "a producer that stopped half way was not noticed");
        CHECK(dt < 20.0, "giving up took %.1f s", dt);
        // ... and the next evaluation on the same pool and flags (a new epoch) is whole again
        std::vector<double> seq(scale.size());
        int fail = -1;
        CHECK(arrow_debug_solve_host(n_cp, acc.data(), scale.data(), 1e4, 1e-6, 1e32, seq.data(), &fail, 0, 0) == 0 && fail == 0, "sequential solve");
        const bool ok2 = streamed_solve(pool, n_cp, 6, acc, scale, 1e4, 2u, flag, ds, 78u);
        CHECK(ok2 && max_abs_diff(ds, seq) <= 1e-9 * max_abs(seq), "the evaluation after a failed one");
    }
    if (fails) {
        fprintf(stderr, "tsan_host_half: %d check(s) failed\n", fails);
        return 1;
    }
    printf("tsan_host_half: ok\n");
    return 0;
}
