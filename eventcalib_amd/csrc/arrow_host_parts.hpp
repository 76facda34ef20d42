// Host linear solve of the LM step on several cores: the banded-arrow system of ecal_solver.hip cut into P interiors
// separated by 3-control-point separators (the band couples a control point with the next three, so two interiors never
// touch).  Every interior is factorised on its own thread — banded Cholesky as solve_arrow does it, with 46 right-hand
// columns instead of 10: [left separator 18 | intrinsics 9 | rhs | right separator 18] — and contributes a 46 x 46 Gram
// block to the reduced system over the separators and the intrinsics (block tridiagonal + dense border, a few hundred
// unknowns), which one thread solves; the interiors are then back-substituted in parallel again.
// Same algebra as arrow_device.hpp (the device form), same result as solve_arrow up to summation order.
// Included by ecal_solver.hip inside its anonymous namespace (uses ArrowSystem, ArrowWorkspace, BW).
// (<atomic>, <condition_variable>, <mutex>, <thread> are included by ecal_solver.hip at file scope)
#pragma once

// A few parked worker threads; run(n, fn) hands out tasks 0 .. n-1 (the caller works too) and returns when all are done.
class HostPool {
public:
    explicit HostPool(int workers, int spin_us = 200) : spin_us_(spin_us) {
        if (const char *e = getenv("ECAL_HOST_POOL_SPIN_US")) spin_us_ = atoi(e);   // debug switch
        for (int i = 0; i < workers; i++) th_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            epoch_.fetch_add(1);
        }
        cv_start_.notify_all();
        for (auto &t : th_) t.join();
    }
    int workers() const { return (int) th_.size(); }
    void run(int n_tasks, const std::function<void(int)> &fn) {
        if (th_.empty() || n_tasks <= 1) {
            for (int t = 0; t < n_tasks; t++) fn(t);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            n_tasks_ = n_tasks;
            next_.store(0);
            pending_ = (int) th_.size();
            epoch_.fetch_add(1);
        }
        cv_start_.notify_all();
        for (int t; (t = next_.fetch_add(1)) < n_tasks;) fn(t);
        std::unique_lock<std::mutex> g(m_);
        cv_done_.wait(g, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            // the pool's calls of one linear solve follow each other within microseconds: poll the counter that long before going
            // to sleep on the condition variable (polling through a whole Jacobian evaluation, 6 ms, measured no faster)
            {
                const auto t0 = std::chrono::steady_clock::now();
                while (epoch_.load(std::memory_order_acquire) == seen) {
                    for (int i = 0; i < 64; i++) __builtin_ia32_pause();
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_)) break;
                }
            }
            {
                std::unique_lock<std::mutex> g(m_);
                cv_start_.wait(g, [&] { return epoch_.load() != seen; });
                seen = epoch_.load();
                if (stop_) return;
            }
            const std::function<void(int)> *fn = fn_;
            const int n = n_tasks_;
            for (int t; (t = next_.fetch_add(1)) < n;) (*fn)(t);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0) cv_done_.notify_one();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_start_, cv_done_;
    const std::function<void(int)> *fn_ = nullptr;
    std::atomic<int> next_{0};
    std::atomic<uint64_t> epoch_{0};
    int n_tasks_ = 0, pending_ = 0, spin_us_ = 200;
    bool stop_ = false;
};

constexpr int APW = 18;               // scalars of a separator (3 control points)
constexpr int APZ = 2 * APW + 10;     // right-hand columns of an interior: left | intrinsics 9 | rhs | right
constexpr int AP_INTR = APW, AP_RHS = APW + 9, AP_RIGHT = APW + 10;

struct ArrowParts {
    int P = 1;
    std::vector<size_t> a, n;        // first row and number of rows of interior p (scalars)
    std::vector<double> Z;           // [nc][APZ] (rows of the separators unused)
    std::vector<double> G;           // [P][APZ * APZ]
    std::vector<double> R, rhs;      // reduced system, dense [NR][NR] (lower) and [NR]
    std::vector<int> lo;             // skyline of the reduced system: first non-zero column of row i
    std::vector<char> ok;            // per interior: positive definite
};

// P interiors for n_cp control points: 1 (the sequential routine) for small problems; fixed by the problem size alone, so the
// result does not depend on the machine's thread count
inline int arrow_parts_for(uint32_t n_cp) {
    if (const char *e = getenv("ECAL_HOST_ARROW_PARTS")) return std::max(1, atoi(e));   // tests: partitions on small problems
    if (n_cp < 512) return 1;
    return (int) std::min<uint32_t>(16u, n_cp / 96u);
}

__attribute__((target("avx2,fma"))) inline void arrow_part_factor(const ArrowSystem &A, const double *__restrict__ sc,
                                                                  const double *__restrict__ dd, ArrowWorkspace &ws, ArrowParts &pt, int p) {
    const size_t nc = A.nc, a = pt.a[p], n = pt.n[p];
    const bool has_left = p > 0, has_right = p + 1 < pt.P;
    double *__restrict__ L = ws.L.data() + a * BW;
    double *__restrict__ Z = pt.Z.data() + a * APZ;
    // scaled entries: band columns before the interior's first row belong to the left separator
    for (size_t i = 0; i < n; i++) {
        const size_t gi = a + i;
        const double si = sc[gi];
        double *__restrict__ Li = L + i * BW, *__restrict__ Zi = Z + i * APZ;
        for (int c = 0; c < APZ; c++) Zi[c] = 0.0;
        for (size_t k = 0; k < (size_t) BW; k++) {
            double v = 0.0;
            if (k <= gi) {
                v = A.band[gi * BW + k] * si * sc[gi - k];
                if (k > i) {   // column gi - k < a: a separator scalar (has_left, and k - i <= APW by the band's reach)
                    Zi[APW - (k - i)] = v;
                    v = 0.0;
                }
            }
            Li[k] = v;
        }
        Li[0] += dd[gi];
        for (int j = 0; j < 9; j++) Zi[AP_INTR + j] = A.border[gi * 9 + j] * si * sc[nc + j];
        Zi[AP_RHS] = -A.gc[gi] * si;
    }
    (void) has_left;
    if (has_right) {   // rows of the right separator reach back into the interior
        const size_t s0 = a + n;
        for (size_t c = 0; c < (size_t) APW; c++) {
            const size_t gc = s0 + c;
            for (size_t k = c + 1; k < (size_t) BW && k <= gc; k++) {
                const size_t r = gc - k;   // < s0
                if (r < a) break;
                Z[(r - a) * APZ + AP_RIGHT + c] = A.band[gc * BW + k] * sc[gc] * sc[r];
            }
        }
    }
    // banded Cholesky, right-looking (as solve_arrow); the right separator's columns are zero above the last APW rows
    const size_t right_from = has_right ? (n > (size_t) APW ? n - APW : 0) : n;
    bool pd = true;
    for (size_t j = 0; j < n; j++) {
        double d = L[j * BW];
        if (!(d > 0.0)) {
            pd = false;
            break;
        }
        d = std::sqrt(d);
        const double inv = 1.0 / d;
        L[j * BW] = d;
        double *__restrict__ Zj = Z + j * APZ;
        const int zc = j >= right_from ? APZ : AP_RIGHT;
        for (int c = 0; c < zc; c++) Zj[c] *= inv;
        const int rmax = (int) std::min<size_t>(BW - 1 - (a + j) % 6, n - 1 - j);
        double col[BW];
        for (int r = 1; r <= rmax; r++) {
            col[r] = L[(j + r) * BW + r] * inv;
            L[(j + r) * BW + r] = col[r];
        }
        for (int r = 1; r <= rmax; r++) {
            const double lr = col[r];
            double *__restrict__ Lr = L + (j + r) * BW;
            for (int c = 1; c <= r; c++) Lr[r - c] -= lr * col[c];
            double *__restrict__ Zr = Z + (j + r) * APZ;
            for (int c = 0; c < zc; c++) Zr[c] -= lr * Zj[c];
        }
    }
    pt.ok[p] = pd ? 1 : 0;
    double *__restrict__ G = pt.G.data() + (size_t) p * APZ * APZ;
    for (int i = 0; i < APZ * APZ; i++) G[i] = 0.0;
    if (!pd) return;
    for (size_t r = 0; r < n; r++) {
        const double *__restrict__ z = Z + r * APZ;
        const int zc = r >= right_from ? APZ : AP_RIGHT;
        for (int i = 0; i < zc; i++) {
            const double zi = z[i];
            double *__restrict__ Gi = G + (size_t) i * APZ;
            for (int j = i; j < zc; j++) Gi[j] += zi * z[j];
        }
    }
}

__attribute__((target("avx2,fma"))) inline void arrow_part_backsub(const ArrowWorkspace &ws, const ArrowParts &pt, int p, const double *yr,
                                                                   double *__restrict__ y) {
    const size_t a = pt.a[p], n = pt.n[p];
    const bool has_right = p + 1 < pt.P;
    const double *__restrict__ L = ws.L.data() + a * BW;
    const double *__restrict__ Z = pt.Z.data() + a * APZ;
    const int NS = APW * (pt.P - 1);
    // the reduced unknowns this interior sees, in its own column order
    double u[APZ];
    for (int c = 0; c < APW; c++) u[c] = p > 0 ? yr[APW * (p - 1) + c] : 0.0;
    for (int j = 0; j < 9; j++) u[AP_INTR + j] = yr[NS + j];
    u[AP_RHS] = 0.0;
    for (int c = 0; c < APW; c++) u[AP_RIGHT + c] = has_right ? yr[APW * p + c] : 0.0;
    const size_t right_from = has_right ? (n > (size_t) APW ? n - APW : 0) : n;
    double *__restrict__ yc = y + a;
    for (size_t ii = n; ii-- > 0;) {
        const double *__restrict__ z = Z + ii * APZ;
        double v = z[AP_RHS];
        const int zc = ii >= right_from ? APZ : AP_RIGHT;
        for (int c = 0; c < zc; c++) v -= z[c] * u[c];   // (u[AP_RHS] = 0)
        const int kmax = (int) std::min<size_t>(BW - 1 - (a + ii) % 6, n - 1 - ii);
        for (int k = 1; k <= kmax; k++) v -= L[(ii + k) * BW + k] * yc[ii + k];
        yc[ii] = v / L[ii * BW];
    }
}

// interiors of (nearly) equal size, separators of 3 control points between them: first control point and number of control
// points of interior p.  The ONE place the cut is defined: the time-sharded multi-GPU mode cuts the residuals at the same
// control points (ecal_solver_time_shard_cuts).
inline void arrow_partition(uint32_t n_cp, int P, std::vector<uint32_t> &first_cp, std::vector<uint32_t> &num_cp) {
    first_cp.resize(P);
    num_cp.resize(P);
    const uint32_t inner = n_cp - 3u * (uint32_t) (P - 1);
    uint32_t at = 0;
    for (int p = 0; p < P; p++) {
        const uint32_t m = inner / P + ((uint32_t) p < inner % P ? 1u : 0u);
        first_cp[p] = at;
        num_cp[p] = m;
        at += m + 3u;
    }
}

// (S A S + diag(dd)) y = -S g with P interiors on the pool's threads; false if not positive definite.
// only_part >= 0 (time-sharded ranks: one interior per rank): this process factorises and back-substitutes that interior only;
// `exchange` sums pt.G (all P blocks; the other ranks' are zero here) and pt.ok over the ranks in between.  A holds this rank's
// rows, the separators' and the intrinsics' rows summed over the ranks; y comes back with this interior, every separator and
// the intrinsics filled in.
inline bool solve_arrow_parts(const ArrowSystem &A, const std::vector<double> &scale, const std::vector<double> &dd, std::vector<double> &y,
                              ArrowWorkspace &ws, ArrowParts &pt, HostPool *pool, int P, int only_part = -1,
                              const std::function<bool(ArrowParts &)> *exchange = nullptr) {
    const size_t nc = A.nc;
    const uint32_t n_cp = (uint32_t) (nc / 6);
    const double *sc = scale.data();
    pt.P = P;
    pt.a.resize(P);
    pt.n.resize(P);
    {
        std::vector<uint32_t> f, m;
        arrow_partition(n_cp, P, f, m);
        for (int p = 0; p < P; p++) {
            pt.a[p] = 6 * (size_t) f[p];
            pt.n[p] = 6 * (size_t) m[p];
        }
    }
    ws.L.resize(nc * BW);
    pt.Z.resize(nc * APZ);
    pt.G.assign((size_t) P * APZ * APZ, 0.0);
    pt.ok.assign(P, 0);
    if (only_part >= 0) {
        arrow_part_factor(A, sc, dd.data(), ws, pt, only_part);
        if (!exchange || !(*exchange)(pt)) return false;
    } else if (pool) {
        pool->run(P, [&](int p) { arrow_part_factor(A, sc, dd.data(), ws, pt, p); });
    } else {
        for (int p = 0; p < P; p++) arrow_part_factor(A, sc, dd.data(), ws, pt, p);
    }
    for (int p = 0; p < P; p++)
        if (!pt.ok[p]) return false;
    // reduced system over [separator 0 .. separator P-2 | intrinsics]: block tridiagonal + dense border, skyline Cholesky
    const int NS = APW * (P - 1), NR = NS + 9;
    pt.R.assign((size_t) NR * NR, 0.0);
    pt.rhs.assign(NR, 0.0);
    pt.lo.resize(NR);
    double *R = pt.R.data();
    for (int s = 0; s + 1 < P; s++) {
        const size_t g0 = pt.a[s] + pt.n[s];   // first scalar of separator s
        for (int i = 0; i < APW; i++) {
            const size_t gi = g0 + i;
            const int ri = APW * s + i;
            pt.lo[ri] = s > 0 ? APW * (s - 1) : 0;
            for (int j = 0; j <= i; j++) R[(size_t) ri * NR + APW * s + j] = A.band[gi * BW + (i - j)] * sc[gi] * sc[g0 + j];
            R[(size_t) ri * NR + ri] += dd[gi];
            for (int j = 0; j < 9; j++) R[(size_t) (NS + j) * NR + ri] = A.border[gi * 9 + j] * sc[gi] * sc[nc + j];
            pt.rhs[ri] = -A.gc[gi] * sc[gi];
        }
    }
    for (int i = 0; i < 9; i++) {
        pt.lo[NS + i] = 0;
        for (int j = 0; j <= i; j++) R[(size_t) (NS + i) * NR + NS + j] = A.corner[9 * i + j] * sc[nc + i] * sc[nc + j];
        R[(size_t) (NS + i) * NR + NS + i] += dd[nc + i];
        pt.rhs[NS + i] = -A.gi[i] * sc[nc + i];
    }
    for (int p = 0; p < P; p++) {   // minus the interiors' Gram blocks (upper triangles stored)
        const double *G = pt.G.data() + (size_t) p * APZ * APZ;
        int idx[APZ];   // reduced index of the interior's column, -1: none
        for (int c = 0; c < APW; c++) idx[c] = p > 0 ? APW * (p - 1) + c : -1;
        for (int j = 0; j < 9; j++) idx[AP_INTR + j] = NS + j;
        idx[AP_RHS] = -1;
        for (int c = 0; c < APW; c++) idx[AP_RIGHT + c] = p + 1 < P ? APW * p + c : -1;
        for (int i = 0; i < APZ; i++) {
            if (idx[i] < 0) continue;
            for (int j = 0; j < APZ; j++) {
                if (idx[j] < 0 || idx[j] > idx[i]) continue;   // lower triangle of R
                R[(size_t) idx[i] * NR + idx[j]] -= i <= j ? G[(size_t) i * APZ + j] : G[(size_t) j * APZ + i];
            }
            pt.rhs[idx[i]] -= i <= AP_RHS ? G[(size_t) i * APZ + AP_RHS] : G[(size_t) AP_RHS * APZ + i];
        }
    }
    for (int i = 0; i < NR; i++) {   // skyline Cholesky: row i starts at column lo[i]
        double *Ri = R + (size_t) i * NR;
        for (int j = pt.lo[i]; j <= i; j++) {
            const double *Rj = R + (size_t) j * NR;
            double v = Ri[j];
            for (int k = std::max(pt.lo[i], pt.lo[j]); k < j; k++) v -= Ri[k] * Rj[k];
            if (i == j) {
                if (!(v > 0.0)) return false;
                Ri[i] = std::sqrt(v);
            } else {
                Ri[j] = v / Rj[j];
            }
        }
    }
    std::vector<double> yr(NR);
    for (int i = 0; i < NR; i++) {
        double v = pt.rhs[i];
        for (int k = pt.lo[i]; k < i; k++) v -= R[(size_t) i * NR + k] * yr[k];
        yr[i] = v / R[(size_t) i * NR + i];
    }
    for (int i = NR - 1; i >= 0; i--) {
        double v = yr[i];
        for (int k = i + 1; k < NR; k++)
            if (pt.lo[k] <= i) v -= R[(size_t) k * NR + i] * yr[k];
        yr[i] = v / R[(size_t) i * NR + i];
    }
    y.assign(nc + 9, 0.0);
    for (int j = 0; j < 9; j++) y[nc + j] = yr[NS + j];
    for (int s = 0; s + 1 < P; s++)
        for (int i = 0; i < APW; i++) y[pt.a[s] + pt.n[s] + i] = yr[APW * s + i];
    if (only_part >= 0) arrow_part_backsub(ws, pt, only_part, yr.data(), y.data());
    else if (pool) pool->run(P, [&](int p) { arrow_part_backsub(ws, pt, p, yr.data(), y.data()); });
    else
        for (int p = 0; p < P; p++) arrow_part_backsub(ws, pt, p, yr.data(), y.data());
    return true;
}
