// The host half of the LM step (EventCalibSpline.cpp:238-247: the linear algebra Ceres' SPARSE_NORMAL_CHOLESKY does there): the
// banded-arrow system unpacked from the kernel's accumulation buffer, the sequential solve, the partitioned solves on a
// worker pool (arrow_host_parts.hpp), the host tasks of the streamed evaluation, and the bodies of the debug hooks.  Plain C++ —
// no HIP in here: ecal_solver.hip includes it inside its anonymous namespace, and tests/cpp/tsan_host_half.cpp compiles it
// with g++ -fsanitize=thread (make tsan) to run the pool, the partitioned solves and the streamed tasks under ThreadSanitizer
// against a host thread that plays the kernel's part.
// Included inside the including file's anonymous namespace, after `using namespace ecal` and these file-scope includes:
// <algorithm> <atomic> <chrono> <cmath> <condition_variable> <cstdio> <cstdlib> <cstring> <functional> <memory> <mutex> <thread>
// <vector> <sched.h> <time.h> "arrow_layout.hpp"
#pragma once

constexpr int BW = 24;  // scalar half-bandwidth + 1 of the control-point part (4 blocks of 6)

struct ArrowSystem {
    size_t nc = 0;                // 6 * n_cp
    std::vector<double> band;     // [nc][BW]: band[i][k] = A(i, i-k), lower band
    std::vector<double> border;   // [nc][9]:  A(i, intr j)
    double corner[81];            // A(intr, intr)
    std::vector<double> gc;       // [nc]
    double gi[9];
};

// unpack the accumulation buffer (upper blocks) into the symmetric arrow system
void unpack_head(const double *acc, ArrowSystem &A) {
    for (int i = 0; i < 9; i++) {
        A.gi[i] = acc[1 + i];
        for (int j = i; j < 9; j++) A.corner[9 * i + j] = A.corner[9 * j + i] = acc[10 + 9 * i + j];
    }
}
// rows of control points [r_lo, r_hi): the band row of control point r = blocks (c, r) for c = r - 3 .. r, read from the column
// owners' records — a range writes its own rows only, so ranges may run on different threads
void unpack_rows(const double *acc, ArrowSystem &A, uint32_t r_lo, uint32_t r_hi) {
    for (uint32_t r = r_lo; r < r_hi; r++) {
        const double *b = acc + ACC_HEAD + ACC_PER_CP * (size_t) r;
        for (int k = 0; k < 6; k++) {
            A.gc[6 * r + k] = b[k];
            for (int j = 0; j < 9; j++) A.border[(6 * (size_t) r + k) * 9 + j] = b[6 + 9 * k + j];
            double *row = &A.band[(6 * (size_t) r + k) * BW];
            for (int q = 0; q < BW; q++) row[q] = 0.0;
        }
        for (uint32_t d = 0; d < 4 && d <= r; d++) {
            const uint32_t c = r - d;   // column owner: block (c, c + d) of its record
            const double *blk = acc + ACC_HEAD + ACC_PER_CP * (size_t) c + 60 + 36 * d;
            for (int ka = 0; ka < 6; ka++)
                for (int kb = 0; kb < 6; kb++) {
                    if (d == 0 && kb < ka) continue;  // diagonal block: upper stored
                    const size_t row = 6 * (size_t) r + kb, col = 6 * (size_t) c + ka;  // row >= col
                    A.band[row * BW + (row - col)] = blk[6 * ka + kb];
                }
        }
    }
}
void unpack_alloc(uint32_t n_cp, ArrowSystem &A) {
    A.nc = 6 * (size_t) n_cp;
    A.band.resize(A.nc * BW);
    A.border.resize(A.nc * 9);
    A.gc.resize(A.nc);
}
void unpack(const double *acc, uint32_t n_cp, ArrowSystem &A) {
    unpack_alloc(n_cp, A);
    unpack_head(acc, A);
    unpack_rows(acc, A, 0, n_cp);
}

// Solve (S A S + diag(dd)) y = -S g for the arrow system; returns false if not positive definite.
// scale: Jacobi column scaling S (nc + 9); dd: LM diagonal added to the scaled system (nc + 9).
// ws: caller-owned workspace (reused across iterations).  The border and the right-hand side are kept
// as 10 contiguous columns so that the substitution loops vectorise.
struct ArrowWorkspace {
    std::vector<double> L, Z;  // [nc][BW] factor; [nc][10] = L^-1 [border | rhs]
    double G[100];             // [Zb z]^T [Zb z]
    std::function<bool(double *)> reduce_G;  // distributed mode: sums G over ranks in place; false = some rank failed
};

__attribute__((target("avx2,fma"))) bool solve_arrow(const ArrowSystem &A, const std::vector<double> &scale, const std::vector<double> &dd,
                 std::vector<double> &y, ArrowWorkspace &ws) {
    const size_t nc = A.nc;
    ws.L.resize(nc * BW);
    ws.Z.resize(nc * 10);
    double *__restrict__ L = ws.L.data();
    double *__restrict__ Z = ws.Z.data();
    const double *__restrict__ sc = scale.data();
    for (size_t i = 0; i < nc; i++) {
        const double si = sc[i];
        const size_t kmax = std::min<size_t>(BW - 1, i);
        for (size_t k = 0; k <= kmax; k++) L[i * BW + k] = A.band[i * BW + k] * si * sc[i - k];
        for (size_t k = kmax + 1; k < (size_t) BW; k++) L[i * BW + k] = 0.0;
        L[i * BW] += dd[i];
        for (int j = 0; j < 9; j++) Z[i * 10 + j] = A.border[i * 9 + j] * si * sc[nc + j];
        Z[i * 10 + 9] = -A.gc[i] * si;
    }
    // banded Cholesky, right-looking: once column j is final, its rank-1 update goes into the (at most BW-1) rows
    // below it and into their 10 border / right-hand-side columns.  The inner loops run over contiguous pieces of a
    // row's band and of a small copy of the column, with fixed short trip counts: they vectorise (this function is
    // compiled for AVX2 + FMA), which the row-wise dot-product form did not.
    bool pd = true;
    for (size_t j = 0; j < nc; j++) {
        double d = L[j * BW];
        if (!(d > 0.0)) {
            pd = false;
            break;
        }
        d = std::sqrt(d);
        const double inv = 1.0 / d;
        L[j * BW] = d;
        double *__restrict__ Zj = Z + j * 10;
        for (int c = 0; c < 10; c++) Zj[c] *= inv;
        // the band is a BLOCK band (4 blocks of 6): column j of block column J reaches down to row 6 (J + 3) + 5 only, and
        // Cholesky fill stays inside that envelope — the rows beyond hold exact zeros
        const int rmax = (int) std::min<size_t>(BW - 1 - j % 6, nc - 1 - j);
        double col[BW];  // col[r] = L(j + r, j)
        for (int r = 1; r <= rmax; r++) {
            col[r] = L[(j + r) * BW + r] * inv;
            L[(j + r) * BW + r] = col[r];
        }
        for (int r = 1; r <= rmax; r++) {
            const double lr = col[r];
            double *__restrict__ Lr = L + (j + r) * BW;  // row j + r: entry (j + r, j + c) sits at band offset r - c
            for (int c = 1; c <= r; c++) Lr[r - c] -= lr * col[c];
            double *__restrict__ Zr = Z + (j + r) * 10;
            for (int c = 0; c < 10; c++) Zr[c] -= lr * Zj[c];
        }
    }
    // Schur complement on the 9 intrinsics: S = C - Zb^T Zb, b = -g_i - Zb^T z.  G = [Zb z]^T [Zb z] is a sum over the
    // control-point rows: with the segments sharded over ranks it is the one thing the linear solve has to all-reduce.
    double S[81], bvec[9];
    double *G = ws.G;
    for (int i = 0; i < 100; i++) G[i] = 0.0;
    if (!pd) {
        if (ws.reduce_G) {  // every rank must take part in the collective
            G[99] = std::numeric_limits<double>::quiet_NaN();
            (void) ws.reduce_G(G);
        }
        return false;
    }
    for (size_t r = 0; r < nc; r++) {
        const double *__restrict__ z = Z + r * 10;
        for (int i = 0; i < 10; i++) {
            const double zi = z[i];
            for (int j = i; j < 10; j++) G[10 * i + j] += zi * z[j];
        }
    }
    if (ws.reduce_G && !ws.reduce_G(G)) return false;  // sum over ranks (also carries "some rank failed")
    for (int i = 0; i < 9; i++) {
        for (int j = 0; j < 9; j++) {
            double v = A.corner[9 * i + j] * sc[nc + i] * sc[nc + j] - (i <= j ? G[10 * i + j] : G[10 * j + i]);
            if (i == j) v += dd[nc + i];
            S[9 * i + j] = v;
        }
        bvec[i] = -A.gi[i] * sc[nc + i] - G[10 * i + 9];
    }
    for (int i = 0; i < 9; i++) {  // dense Cholesky 9x9
        for (int j = 0; j <= i; j++) {
            double v = S[9 * i + j];
            for (int k = 0; k < j; k++) v -= S[9 * i + k] * S[9 * j + k];
            if (i == j) {
                if (!(v > 0.0)) return false;
                S[9 * i + i] = std::sqrt(v);
            } else {
                S[9 * i + j] = v / S[9 * j + j];
            }
        }
    }
    double yi[9];
    for (int i = 0; i < 9; i++) {
        double v = bvec[i];
        for (int k = 0; k < i; k++) v -= S[9 * i + k] * yi[k];
        yi[i] = v / S[9 * i + i];
    }
    for (int i = 8; i >= 0; i--) {
        double v = yi[i];
        for (int k = i + 1; k < 9; k++) v -= S[9 * k + i] * yi[k];
        yi[i] = v / S[9 * i + i];
    }
    // back substitution for the control points: L^T y_c = z - Zb yi
    y.assign(nc + 9, 0.0);
    for (int j = 0; j < 9; j++) y[nc + j] = yi[j];
    double *__restrict__ yc = y.data();
    for (size_t ii = nc; ii-- > 0;) {
        const double *__restrict__ z = Z + ii * 10;
        double v = z[9];
        for (int j = 0; j < 9; j++) v -= z[j] * yi[j];
        const int kmax = (int) std::min<size_t>(BW - 1 - ii % 6, nc - 1 - ii);
        for (int k = 1; k <= kmax; k++) v -= L[(ii + k) * BW + k] * yc[ii + k];
        yc[ii] = v / L[ii * BW];
    }
    return true;
}

#include "arrow_host_parts.hpp"

// y^T A y and g^T y on the unscaled system (for the model cost change)
// skip_shared: distributed mode, ranks other than 0 — the intrinsics-only terms are counted once
void quad_forms_rows(const ArrowSystem &A, const std::vector<double> &d, size_t lo, size_t hi, double *gTd, double *dHd) {
    const size_t nc = A.nc;
    double g = 0, h = 0;
    for (size_t i = lo; i < hi; i++) {
        g += A.gc[i] * d[i];
        double row = A.band[i * BW] * d[i];
        const int kmax = std::min<size_t>(BW - 1, i);
        for (int k = 1; k <= kmax; k++) row += 2.0 * A.band[i * BW + k] * d[i - k];
        h += d[i] * row;
        for (int j = 0; j < 9; j++) h += 2.0 * d[i] * A.border[i * 9 + j] * d[nc + j];
    }
    *gTd = g;
    *dHd = h;
}
void quad_forms(const ArrowSystem &A, const std::vector<double> &d, double *gTd, double *dHd, bool skip_shared = false, HostPool *pool = nullptr,
                int tasks = 1) {
    const size_t nc = A.nc;
    double g = 0, h = 0;
    if (pool && tasks > 1) {   // fixed ranges, partial sums added in range order: the result does not depend on the thread count
        std::vector<double> part(2 * (size_t) tasks, 0.0);
        const size_t per = (nc + tasks - 1) / tasks;
        pool->run(tasks, [&](int t) { quad_forms_rows(A, d, std::min(nc, t * per), std::min(nc, (t + 1) * per), &part[2 * t], &part[2 * t + 1]); });
        for (int t = 0; t < tasks; t++) {
            g += part[2 * t];
            h += part[2 * t + 1];
        }
    } else {
        quad_forms_rows(A, d, 0, nc, &g, &h);
    }
    for (int i = 0; i < 9 && !skip_shared; i++) {
        g += A.gi[i] * d[nc + i];
        for (int j = 0; j < 9; j++) h += d[nc + i] * A.corner[9 * i + j] * d[nc + j];
    }
    *gTd = g;
    *dHd = h;
}

// ---- the host tasks of a streamed evaluation -------------------------------------------------------------------------
// What the host knows of its producer (the normal-equation kernel, NeProgress in ecal_solver.hip; a host thread in the
// ThreadSanitizer build): group g's records [cut[g], cut[g+1] - 3) (the last group: to cut[P]) are in `acc` once flag[g] carries
// this evaluation's number, the three records of the separator behind group b once flag[NE_MAX_GROUPS + b] does; init[..] == 0:
// nothing ever arrives for that item (its records are zero).  producer_gone: asked by a waiter after two seconds without its
// flag — true = the producer has stopped (finished or failed) and the flag will not come.
struct StreamedSource {
    const uint32_t *init = nullptr;   // [2 * NE_MAX_GROUPS]
    const uint32_t *cut = nullptr;    // [P + 1]
    const uint32_t *flag = nullptr;   // [2 * NE_MAX_GROUPS], written by the producer with release semantics
    uint32_t epoch = 0;
    std::function<bool()> producer_gone;
};

// Tasks 0 .. P-1: an interior each (taken in this order: a thread that waits, waits for the producer alone) — wait for the
// group's flag, unpack its rows, and with `factor` add the LM diagonal of radius r_fact (dd_next) and factorise it; task P, with
// `factor`: the separators of the reduced system one after the other, each as soon as the interiors beside it are done.
// Returns false when the producer failed to deliver (then nothing in An / ws / parts is to be used); *reduced_ok: every
// separator eliminated.  tl_arrive / tl_done ([P + 1], may be null): seconds since t_origin at which an interior's records were
// in / its task was done.
inline bool arrow_streamed_tasks(HostPool *pool, int P, const StreamedSource &src, const double *acc, ArrowSystem &An, bool factor, double r_fact,
                                 const double *scale, double *dd_next, double min_diag, double max_diag, ArrowWorkspace &ws, ArrowParts &parts,
                                 bool *reduced_ok, double *tl_arrive, double *tl_done, std::chrono::steady_clock::time_point t_origin) {
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    *reduced_ok = false;
    std::atomic<bool> failed{false};
    std::atomic<bool> good_reduced{false};
    std::unique_ptr<std::atomic<int>[]> done(new std::atomic<int>[P]);
    for (int p = 0; p < P; p++) done[p].store(0);
    const uint32_t *flag = src.flag;
    const uint32_t epoch = src.epoch;
    auto up = [&](uint32_t idx) { return __atomic_load_n(&flag[idx], __ATOMIC_ACQUIRE) == epoch; };
    // A thread polls its flag only when its turn is near (the flag four groups earlier is up); before that it naps: sixteen
    // polling threads for the length of a kernel are the machine's whole CPU allowance on a 16-CPU cgroup, and a throttled
    // process loses milliseconds (measured: 290 -> 230 iterations/s in runs that hit the quota).
    auto nap = [] {
        timespec ts{0, 20000};
        nanosleep(&ts, nullptr);
    };
    // a flag that does not come: after two seconds of waiting the producer is asked — still busy: wait on (a long kernel);
    // gone with the flag down: it has failed to deliver, every waiter gives up
    auto overdue = [&](std::chrono::steady_clock::time_point &t0) -> bool {
        if (failed.load()) return true;
        if (secs(t0, now()) < 2.0) return false;
        t0 = now();
        return src.producer_gone ? src.producer_gone() : false;
    };
    auto wait_for = [&](uint32_t idx, int gate) {
        auto t0 = now();
        if (gate >= 0)
            while (!up((uint32_t) gate) && !up(idx)) {
                nap();
                if (overdue(t0) && !up(idx)) {
                    failed.store(true);
                    return;
                }
            }
        for (uint32_t spin = 0; !up(idx); spin++) {
            for (int i = 0; i < 16; i++) __builtin_ia32_pause();
            if ((spin & 0xFFFu) == 0xFFFu && overdue(t0) && !up(idx)) {
                failed.store(true);
                return;
            }
        }
    };
    pool->run(factor ? P + 1 : P, [&](int p) {
        if (p == P) {
            bool good = true;
            for (int sp = 0; sp + 1 < P && good; sp++) {
                for (uint32_t spin = 0; !(done[sp].load(std::memory_order_acquire) && done[sp + 1].load(std::memory_order_acquire)); spin++) {
                    if (sp + 8 < P) nap();   // (the separators of the spline's last stretch are the ones to be prompt about)
                    else
                        for (int i = 0; i < 16; i++) __builtin_ia32_pause();
                    if ((spin & 0xFFu) == 0xFFu && failed.load()) return;
                }
                if (failed.load()) return;
                // the separator's own records are in (the interior behind it waited for them): its rows, its diagonal
                unpack_rows(acc, An, src.cut[sp + 1] - 3, src.cut[sp + 1]);
                for (size_t i = 6 * (size_t) (src.cut[sp + 1] - 3); i < 6 * (size_t) src.cut[sp + 1]; i++) {
                    const double h = An.band[i * BW] * scale[i] * scale[i];
                    dd_next[i] = std::min(std::max(h, min_diag), max_diag) / r_fact;
                }
                good = arrow_reduced_separator(An, scale, dd_next, parts, sp);
            }
            good_reduced.store(good);
            if (tl_done) tl_done[P] = secs(t_origin, now());
            return;
        }
        if (src.init[p]) wait_for((uint32_t) p, p >= 4 && src.init[p - 4] ? p - 4 : -1);
        if (p > 0 && src.init[NE_MAX_GROUPS + p - 1]) wait_for((uint32_t) (NE_MAX_GROUPS + p - 1), -1);
        if (p == P - 1 && factor) pool->nudge();   // the end is near: the threads that finished early are wanted for the back-substitution
        if (tl_arrive) tl_arrive[p] = secs(t_origin, now());
        if (!failed.load()) {
            // rows of the interior and of the separator behind it (whose own records are not complete yet: those rows are
            // unpacked again later; what the factorisation reads of them comes from the interior's records).  The interior's
            // task writes the rows [cut[p], cut[p+1] - 3) for good and is the only reader of the three rows behind them until
            // done[p] is up; the separator task rewrites those three after done[p] and done[p+1].
            unpack_rows(acc, An, src.cut[p], src.cut[p + 1]);
            if (factor) {
                for (size_t i = parts.a[p]; i < parts.a[p] + parts.n[p]; i++) {
                    const double h = An.band[i * BW] * scale[i] * scale[i];
                    dd_next[i] = std::min(std::max(h, min_diag), max_diag) / r_fact;
                }
                arrow_part_factor(An, scale, dd_next, ws, parts, p);
            }
        }
        if (tl_done) tl_done[p] = secs(t_origin, now());
        done[p].store(1, std::memory_order_release);
    });
    *reduced_ok = good_reduced.load();
    return !failed.load();
}

// ---- bodies of the debug hooks (ecal_solver.hip exports them; the ThreadSanitizer build calls them directly) ----------
// modes: 0 the sequential routine, 2 the partitioned one on the even partition, 3 on the streamed evaluation's partition
inline int arrow_debug_solve_host(uint32_t n_cp, const double *accum, const double *scale, double radius, double min_diag, double max_diag,
                                  double *delta_out, int *fail_out, int mode, int parts_wanted, int pool_workers = 3) {
    if (!accum || !scale || !delta_out || !fail_out || n_cp < 1 || (mode != 0 && mode != 2 && mode != 3)) return -1;
    const size_t nc = 6 * (size_t) n_cp, nt = nc + 9;
    ArrowSystem A;
    ArrowWorkspace ws;
    unpack(accum, n_cp, A);
    std::vector<double> sc(scale, scale + nt), dd(nt), y;
    for (size_t i = 0; i < nt; i++) {
        const double h = (i < nc ? A.band[i * BW] : A.corner[10 * (i - nc)]) * sc[i] * sc[i];
        dd[i] = std::min(std::max(h, min_diag), max_diag) / radius;
    }
    bool ok;
    if (mode == 2 || mode == 3) {
        ArrowParts parts;
        int P = parts_wanted > 0 ? parts_wanted : arrow_parts_for(n_cp);
        if (P < 2) P = 4;
        if ((uint32_t) (7 * P) > n_cp) return -6;   // ECAL_ERR_RANGE
        HostPool pool(pool_workers);
        ok = solve_arrow_parts(A, sc, dd, y, ws, parts, &pool, P, -1, nullptr, mode == 3);
    } else {
        ok = solve_arrow(A, sc, dd, y, ws);
    }
    *fail_out = ok ? 0 : 1;
    if (ok)
        for (size_t i = 0; i < nt; i++) delta_out[i] = y[i] * sc[i];
    return 0;
}

// the solve's worker pool — `rounds` runs of 1 .. 40 tasks on `workers` threads, with pauses long enough for the workers to go to
// sleep now and then, nudges from inside tasks (as the streamed evaluation's last interior does) and from the caller; every task
// of every run must have run exactly once, in a run of its own.  Returns the number of violations.
inline int host_pool_selftest(int workers, int rounds) {
    if (workers < 0 || workers > 64 || rounds < 1) return -1;
    HostPool pool(workers, 50);
    std::vector<std::atomic<int>> hits(64);
    int bad = 0;
    uint64_t rng = 88172645463325252ull;
    auto next = [&]() {
        rng ^= rng << 13;
        rng ^= rng >> 7;
        rng ^= rng << 17;
        return rng;
    };
    std::atomic<int> in_run{0};
    for (int r = 0; r < rounds; r++) {
        const int n = 1 + (int) (next() % 40u);
        for (auto &h : hits) h.store(0);
        const int nudger = (int) (next() % (uint64_t) n);
        if (next() % 4u == 0) std::this_thread::sleep_for(std::chrono::microseconds(120));   // (workers asleep by now)
        if (next() % 3u == 0) pool.nudge();
        in_run.store(r + 1);
        pool.run(n, [&](int t) {
            if (in_run.load() != r + 1) hits[63].fetch_add(1000);   // a task of another run
            if (t == nudger) pool.nudge();
            volatile double x = 0;
            for (int i = 0; i < 200 + (t * 37) % 500; i++) x = x + i;
            hits[t].fetch_add(1);
        });
        for (int t = 0; t < 64; t++) bad += hits[t].load() != (t < n ? 1 : 0);
    }
    return bad;
}
