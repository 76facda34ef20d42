// Host linear solve of the LM step on several cores: the banded-arrow system of ecal_solver.hip cut into P interiors
// separated by 3-control-point separators (the band couples a control point with the next three, so two interiors never
// touch).  Every interior is factorised on its own thread — banded Cholesky as solve_arrow does it, with 46 right-hand
// columns instead of 10: [left separator 18 | intrinsics 9 | rhs | right separator 18] — and contributes a 46 x 46 Gram
// block to the reduced system over the separators and the intrinsics (block tridiagonal + dense border, a few hundred
// unknowns), which one thread solves; the interiors are then back-substituted in parallel again.
// Same result as solve_arrow up to summation order.  (A device form of the same algebra — interiors in LDS, one workgroup each — ran
// 2.9 ms per solve against this file's 0.6 ms on the benchmark problem: profiles/experiments/r06_device_linear_solve.patch.)
// Included by ecal_solver.hip inside its anonymous namespace (uses ArrowSystem, ArrowWorkspace, BW).
// (<atomic>, <condition_variable>, <mutex>, <thread>, <sched.h> are included by ecal_solver.hip at file scope)
#pragma once

// CPUs this process may keep busy: its affinity mask ∩ the cgroup's CPU quota (cpu.max of cgroup v2, cpu.cfs_quota_us of v1),
// shared among the ranks of the node (LOCAL_WORLD_SIZE, as torch.distributed.run and the bench's launcher set it) — not
// std::thread::hardware_concurrency(): the GPU box shows 256 hardware threads behind a quota of 16 CPUs, and eight ranks
// with fifteen workers each would be 120 polling threads on 16 CPUs (a throttled cgroup loses milliseconds per iteration).
// ECAL_HOST_THREADS overrides (tests, odd launchers).  quota_out (may be null): the node-wide figure before the division.
inline int host_usable_cpus(int *quota_out = nullptr) {
    long n = (long) std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = std::min<long>(n, CPU_COUNT(&set));
    auto quota_of = [](const char *path_max, const char *path_q, const char *path_p) -> long {
        if (path_max) {
            if (FILE *f = fopen(path_max, "r")) {   // "max 100000" or "<quota> <period>"
                char q[64] = {0};
                long period = 0;
                const int got = fscanf(f, "%63s %ld", q, &period);
                fclose(f);
                if (got == 2 && period > 0 && q[0] >= '0' && q[0] <= '9') return (atol(q) + period - 1) / period;
            }
            return 0;
        }
        long quota = -1, period = 0;
        if (FILE *f = fopen(path_q, "r")) {
            if (fscanf(f, "%ld", &quota) != 1) quota = -1;
            fclose(f);
        }
        if (FILE *f = fopen(path_p, "r")) {
            if (fscanf(f, "%ld", &period) != 1) period = 0;
            fclose(f);
        }
        return quota > 0 && period > 0 ? (quota + period - 1) / period : 0;
    };
    for (long q : {quota_of("/sys/fs/cgroup/cpu.max", nullptr, nullptr),
                   quota_of(nullptr, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"),
                   quota_of(nullptr, "/sys/fs/cgroup/cpu,cpuacct/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu,cpuacct/cpu.cfs_period_us")})
        if (q > 0) n = std::min(n, q);
    if (quota_out) *quota_out = (int) n;
    long ranks = 1;
    if (const char *e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1L, atol(e));
    n = std::max(1L, n / ranks);
    if (const char *e = getenv("ECAL_HOST_THREADS"))
        if (atol(e) >= 1) n = atol(e);
    return (int) n;
}

// A few parked worker threads; run(n, fn) hands out tasks 0 .. n-1 (the caller works too) and returns when all are done.
class HostPool {
public:
    explicit HostPool(int workers, int spin_us = 400) : spin_us_(spin_us) {
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
    // workers that have gone to sleep come back to polling (for spin_us): called a little before a run() that is on the caller's
    // critical path — waking a sleeping thread takes 50 - 150 us, polling ones pick a task up in a microsecond or two
    void nudge() {
        {
            std::lock_guard<std::mutex> g(m_);
            nudge_.fetch_add(1);
        }
        cv_start_.notify_all();
    }
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
        uint64_t seen = 0, seen_nudge = 0;
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
                cv_start_.wait(g, [&] { return epoch_.load() != seen || nudge_.load() != seen_nudge; });
                seen_nudge = nudge_.load();
                if (epoch_.load() == seen) continue;   // nudged: poll again
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
    std::atomic<uint64_t> epoch_{0}, nudge_{0};
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
    std::vector<double> R, rhs, yr;  // reduced system, dense [NR][NR] (lower), [NR], and its solution
    int NS = 0, NR = 0;              // scalars of all separators; + the 9 intrinsics
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

#ifdef ECAL_ARROW_PROF
static unsigned long long g_arrow_prof[8];
#endif
__attribute__((target("avx2,fma"))) inline void arrow_part_factor(const ArrowSystem &A, const double *__restrict__ sc,
                                                                  const double *__restrict__ dd, ArrowWorkspace &ws, ArrowParts &pt, int p) {
    const size_t nc = A.nc, a = pt.a[p], n = pt.n[p];
    const bool has_left = p > 0, has_right = p + 1 < pt.P;
    double *__restrict__ L = ws.L.data() + a * BW;
    double *__restrict__ Z = pt.Z.data() + a * APZ;
#ifdef ECAL_ARROW_PROF   // profiling builds: cycles per phase of this routine (tools), printed by the timing hook
#define AP_MARK(k) { const unsigned long long t_ = __builtin_ia32_rdtsc(); g_arrow_prof[k] += t_ - ap_t; ap_t = t_; }
    unsigned long long ap_t = __builtin_ia32_rdtsc();
#else
#define AP_MARK(k)
#endif
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
#ifdef ECAL_ARROW_UNBLOCKED   // A/B builds: the column-by-column form (profiles/r04_notes.md)
    const size_t right_from_u = has_right ? (n > (size_t) APW ? n - APW : 0) : n;
    bool pd_u = true;
    for (size_t j = 0; j < n; j++) {
        double d = L[j * BW];
        if (!(d > 0.0)) {
            pd_u = false;
            break;
        }
        d = std::sqrt(d);
        const double inv = 1.0 / d;
        L[j * BW] = d;
        double *__restrict__ Zj = Z + j * APZ;
        const int zc = j >= right_from_u ? APZ : AP_RIGHT;
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
    pt.ok[p] = pd_u ? 1 : 0;
    double *__restrict__ Gu = pt.G.data() + (size_t) p * APZ * APZ;
    for (int i = 0; i < APZ * APZ; i++) Gu[i] = 0.0;
    if (!pd_u) return;
    for (size_t r = 0; r < n; r++) {
        const double *__restrict__ z = Z + r * APZ;
        const int zc = r >= right_from_u ? APZ : AP_RIGHT;
        for (int i = 0; i < zc; i++) {
            const double zi = z[i];
            double *__restrict__ Gi = Gu + (size_t) i * APZ;
            for (int j = i; j < zc; j++) Gi[j] += zi * z[j];
        }
    }
    return;
#endif
    AP_MARK(0)
    // banded Cholesky, right-looking, one control point (6 columns) at a time: the block's own 6 x 6 factor and its rows of Z;
    // the panel of the (up to) 18 rows below it — every one of them is inside the band of all six columns —; then the rank-6
    // update of those rows' band entries and Z rows, and the block's share of the Gram matrix Z^T Z.  (Column by column, as
    // solve_arrow does it, every row of Z is read and written once per column: 2 memory operations per multiply-add; here
    // once per six.)  The right separator's columns of Z are zero above the last APW rows.
    const size_t right_from = has_right ? (n > (size_t) APW ? n - APW : 0) : n;
    bool pd = true;
    double *__restrict__ G = pt.G.data() + (size_t) p * APZ * APZ;
    for (int i = 0; i < APZ * APZ; i++) G[i] = 0.0;
    for (size_t j0 = 0; j0 < n && pd; j0 += 6) {
        const int zc = j0 >= right_from ? APZ : AP_RIGHT;
        double *__restrict__ Zb = Z + j0 * APZ;
        // 1. the diagonal block, column by column inside the block
        for (int j = 0; j < 6; j++) {
            double d = L[(j0 + j) * BW];
            if (!(d > 0.0)) {
                pd = false;
                break;
            }
            d = std::sqrt(d);
            const double inv = 1.0 / d;
            L[(j0 + j) * BW] = d;
            double *__restrict__ Zj = Zb + j * APZ;
            for (int c = 0; c < zc; c++) Zj[c] *= inv;
            double col[6];
            for (int r = j + 1; r < 6; r++) {
                col[r] = L[(j0 + r) * BW + (r - j)] * inv;
                L[(j0 + r) * BW + (r - j)] = col[r];
            }
            for (int r = j + 1; r < 6; r++) {
                const double lr = col[r];
                double *__restrict__ Lr = L + (j0 + r) * BW;
                for (int c = j + 1; c <= r; c++) Lr[r - c] -= lr * col[c];
                double *__restrict__ Zr = Zb + r * APZ;
                for (int c = 0; c < zc; c++) Zr[c] -= lr * Zj[c];
            }
        }
        if (!pd) break;
        AP_MARK(1)
        // the block's rows of Z are final: their share of Z^T Z (upper triangle)
        {
            const double *__restrict__ z0 = Zb, *__restrict__ z1 = Zb + APZ, *__restrict__ z2 = Zb + 2 * APZ, *__restrict__ z3 = Zb + 3 * APZ,
                         *__restrict__ z4 = Zb + 4 * APZ, *__restrict__ z5 = Zb + 5 * APZ;
            for (int i = 0; i < zc; i++) {
                const double a0 = z0[i], a1 = z1[i], a2 = z2[i], a3 = z3[i], a4 = z4[i], a5 = z5[i];
                double *__restrict__ Gi = G + (size_t) i * APZ;
                for (int j = i; j < zc; j++) Gi[j] += (a0 * z0[j] + a1 * z1[j] + a2 * z2[j]) + (a3 * z3[j] + a4 * z4[j] + a5 * z5[j]);
            }
        }
        AP_MARK(2)
        const int nq = (int) std::min<size_t>(18, n - (j0 + 6));
        if (nq <= 0) continue;
        // 2. the panel: rows j0 + 6 + q, entries (row, j0 + j) at band index 6 + q - j; solved against the block's factor
        double Pq[18][6], PT[6][20];
        for (int q = 0; q < nq; q++) {
            double *__restrict__ Lr = L + (j0 + 6 + q) * BW;
            for (int j = 0; j < 6; j++) {
                double v = Lr[6 + q - j];
                const double *__restrict__ Lj = L + (j0 + j) * BW;
                for (int jj = 0; jj < j; jj++) v -= Pq[q][jj] * Lj[j - jj];
                v /= Lj[0];
                Pq[q][j] = v;
                PT[j][q] = v;
                Lr[6 + q - j] = v;
            }
        }
        for (int j = 0; j < 6; j++)
            for (int q = nq; q < 20; q++) PT[j][q] = 0.0;
        AP_MARK(3)
        // 3. the rows below: band entries (row q, row q') for q' <= q, and their rows of Z
        for (int q = 0; q < nq; q++) {
            const double l0 = Pq[q][0], l1 = Pq[q][1], l2 = Pq[q][2], l3 = Pq[q][3], l4 = Pq[q][4], l5 = Pq[q][5];
            double t[20];
            for (int qq = 0; qq < 20; qq++)
                t[qq] = (l0 * PT[0][qq] + l1 * PT[1][qq] + l2 * PT[2][qq]) + (l3 * PT[3][qq] + l4 * PT[4][qq] + l5 * PT[5][qq]);
            double *__restrict__ Lr = L + (j0 + 6 + q) * BW;
            for (int qq = 0; qq <= q; qq++) Lr[q - qq] -= t[qq];
            double *__restrict__ Zr = Zb + (size_t) (6 + q) * APZ;
            const double *__restrict__ z0 = Zb, *__restrict__ z1 = Zb + APZ, *__restrict__ z2 = Zb + 2 * APZ, *__restrict__ z3 = Zb + 3 * APZ,
                         *__restrict__ z4 = Zb + 4 * APZ, *__restrict__ z5 = Zb + 5 * APZ;
            for (int c = 0; c < zc; c++) Zr[c] -= (l0 * z0[c] + l1 * z1[c] + l2 * z2[c]) + (l3 * z3[c] + l4 * z4[c] + l5 * z5[c]);
        }
        AP_MARK(4)
    }
#undef AP_MARK
    pt.ok[p] = pd ? 1 : 0;
    if (!pd)
        for (int i = 0; i < APZ * APZ; i++) G[i] = 0.0;
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
    // a row is a chain of ~70 dependent multiply-adds as written; four partial sums (the value of the row before enters last)
    for (size_t ii = n; ii-- > 0;) {
        const double *__restrict__ z = Z + ii * APZ;
        const int zc = ii >= right_from ? APZ : AP_RIGHT;
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        int c = 0;
        for (; c + 3 < zc; c += 4) {   // (u[AP_RHS] = 0)
            v0 += z[c] * u[c];
            v1 += z[c + 1] * u[c + 1];
            v2 += z[c + 2] * u[c + 2];
            v3 += z[c + 3] * u[c + 3];
        }
        for (; c < zc; c++) v0 += z[c] * u[c];
        const int kmax = (int) std::min<size_t>(BW - 1 - (a + ii) % 6, n - 1 - ii);
        int k = kmax;
        for (; k >= 4; k -= 4) {
            v0 += L[(ii + k) * BW + k] * yc[ii + k];
            v1 += L[(ii + k - 1) * BW + k - 1] * yc[ii + k - 1];
            v2 += L[(ii + k - 2) * BW + k - 2] * yc[ii + k - 2];
            v3 += L[(ii + k - 3) * BW + k - 3] * yc[ii + k - 3];
        }
        double tail = 0;
        for (; k >= 1; k--) tail += L[(ii + k) * BW + k] * yc[ii + k];
        yc[ii] = (z[AP_RHS] - ((v0 + v1) + (v2 + v3)) - tail) / L[ii * BW];
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

// The partition of the streamed evaluation (ecal_solver_solve): the interiors arrive in time order while the kernel runs and
// every one is factorised on its own thread on arrival, so what counts is the END — the interiors shrink geometrically towards
// the end of the spline (each is done about when the next, smaller one arrives: factor 1.54 = 1 + kernel time per control
// point / factorisation time per control point, measured), the last one is a few control points; the front of the spline is cut
// into equal interiors.  Fixed by (n_cp, P) alone, as arrow_partition.
inline void arrow_partition_stream(uint32_t n_cp, int P, std::vector<uint32_t> &first_cp, std::vector<uint32_t> &num_cp) {
    first_cp.resize(P);
    num_cp.resize(P);
    const uint32_t inner = n_cp - 3u * (uint32_t) (P - 1);
    std::vector<uint32_t> size(P, 0);
    uint32_t rem = inner;
    int k = P;   // interiors 0 .. k-1 still to size
    double g = std::max(6.0, 0.006 * inner);
    while (k > 1) {
        const uint32_t gi = (uint32_t) std::ceil(g);
        if ((uint64_t) gi * (uint32_t) k >= rem) break;   // as large as an equal share of what is left: the equal part starts here
        size[k - 1] = gi;
        rem -= gi;
        k--;
        g *= 1.54;
    }
    for (int p = 0; p < k; p++) size[p] = rem / (uint32_t) k + ((uint32_t) p < rem % (uint32_t) k ? 1u : 0u);
    uint32_t at = 0;
    for (int p = 0; p < P; p++) {
        first_cp[p] = at;
        num_cp[p] = size[p];
        at += size[p] + 3u;
    }
}

// The steps of solve_arrow_parts, separately for the streamed evaluation (ecal_solver_solve), which runs arrow_part_factor for
// an interior as soon as the GPU has delivered its rows and eliminates a separator as soon as the interiors on both sides of
// it are factorised: the partition and the buffers …
inline void arrow_parts_setup(size_t nc, int P, ArrowWorkspace &ws, ArrowParts &pt, bool stream_partition = false) {
    const uint32_t n_cp = (uint32_t) (nc / 6);
    pt.P = P;
    pt.a.resize(P);
    pt.n.resize(P);
    {
        std::vector<uint32_t> f, m;
        if (stream_partition) arrow_partition_stream(n_cp, P, f, m);
        else arrow_partition(n_cp, P, f, m);
        for (int p = 0; p < P; p++) {
            pt.a[p] = 6 * (size_t) f[p];
            pt.n[p] = 6 * (size_t) m[p];
        }
    }
    ws.L.resize(nc * BW);
    pt.Z.resize(nc * APZ);
    pt.G.assign((size_t) P * APZ * APZ, 0.0);
    pt.ok.assign(P, 0);
    // reduced system over [separator 0 .. separator P-2 | intrinsics]: block tridiagonal + dense border, skyline Cholesky
    pt.NS = APW * (P - 1);
    pt.NR = pt.NS + 9;
    pt.R.assign((size_t) pt.NR * pt.NR, 0.0);
    pt.rhs.assign(pt.NR, 0.0);
    pt.yr.assign(pt.NR, 0.0);
    pt.lo.resize(pt.NR);
    for (int s = 0; s + 1 < P; s++)
        for (int i = 0; i < APW; i++) pt.lo[APW * s + i] = s > 0 ? APW * (s - 1) : 0;
    for (int i = 0; i < 9; i++) pt.lo[pt.NS + i] = 0;
}

// minus interior p's Gram block, the entries of one stage: stage s = rows of separator s and the intrinsics' rows in the columns
// of separator s; stage -1 = the intrinsics' own 9 x 9 corner
inline void arrow_reduced_apply(ArrowParts &pt, int p, int stage) {
    const int P = pt.P, NS = pt.NS, NR = pt.NR;
    double *R = pt.R.data();
    const double *G = pt.G.data() + (size_t) p * APZ * APZ;
    if (stage < 0) {   // the corner: the interior's intrinsics x intrinsics block (upper triangle stored) and its right-hand side
        for (int i = 0; i < 9; i++) {
            for (int j = 0; j <= i; j++) R[(size_t) (NS + i) * NR + NS + j] -= G[(size_t) (AP_INTR + j) * APZ + AP_INTR + i];
            pt.rhs[NS + i] -= G[(size_t) (AP_INTR + i) * APZ + AP_RHS];
        }
        return;
    }
    // stage s sees interior p = s as the separator's LEFT neighbour (the separator is the interior's right one: columns
    // AP_RIGHT ..) and interior p = s + 1 as its right neighbour (the interior's left columns 0 ..); G's upper triangle is stored
    (void) P;
    const int s0 = APW * stage;
    if (p == stage) {
        for (int i = 0; i < APW; i++) {
            double *Rr = R + (size_t) (s0 + i) * NR;
            if (p > 0)   // the interior's left separator = separator s - 1: the block that couples the two separators
                for (int j = 0; j < APW; j++) Rr[s0 - APW + j] -= G[(size_t) j * APZ + AP_RIGHT + i];
            for (int j = 0; j <= i; j++) Rr[s0 + j] -= G[(size_t) (AP_RIGHT + j) * APZ + AP_RIGHT + i];
            for (int j = 0; j < 9; j++) R[(size_t) (NS + j) * NR + s0 + i] -= G[(size_t) (AP_INTR + j) * APZ + AP_RIGHT + i];
            pt.rhs[s0 + i] -= G[(size_t) AP_RHS * APZ + AP_RIGHT + i];
        }
    } else {   // p == stage + 1
        for (int i = 0; i < APW; i++) {
            double *Rr = R + (size_t) (s0 + i) * NR;
            for (int j = 0; j <= i; j++) Rr[s0 + j] -= G[(size_t) j * APZ + i];
            for (int j = 0; j < 9; j++) R[(size_t) (NS + j) * NR + s0 + i] -= G[(size_t) i * APZ + AP_INTR + j];
            pt.rhs[s0 + i] -= G[(size_t) i * APZ + AP_RHS];
        }
    }
}

// separator s of the reduced system (the interiors s and s + 1 factorised, the separator's rows of A, sc and dd in place):
// its rows assembled and factorised, the intrinsics' rows in its columns, the forward substitution of its rows
inline bool arrow_reduced_separator(const ArrowSystem &A, const double *sc, const double *dd, ArrowParts &pt, int s) {
    const size_t nc = A.nc;
    const int NS = pt.NS, NR = pt.NR;
    double *R = pt.R.data();
    if (!pt.ok[s] || !pt.ok[s + 1]) return false;
    const size_t g0 = pt.a[s] + pt.n[s];   // first scalar of separator s
    for (int i = 0; i < APW; i++) {
        const size_t gi = g0 + i;
        const int ri = APW * s + i;
        for (int j = 0; j <= i; j++) R[(size_t) ri * NR + APW * s + j] = A.band[gi * BW + (i - j)] * sc[gi] * sc[g0 + j];
        R[(size_t) ri * NR + ri] += dd[gi];
        for (int j = 0; j < 9; j++) R[(size_t) (NS + j) * NR + ri] = A.border[gi * 9 + j] * sc[gi] * sc[nc + j];
        pt.rhs[ri] = -A.gc[gi] * sc[gi];
    }
    arrow_reduced_apply(pt, s, s);
    arrow_reduced_apply(pt, s + 1, s);
    const int c0 = APW * s, c1 = APW * (s + 1);
    for (int i = c0; i < c1; i++) {   // skyline Cholesky: row i starts at column lo[i]
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
    for (int i = NS; i < NR; i++) {   // the intrinsics' rows, columns of this separator
        double *Ri = R + (size_t) i * NR;
        for (int j = c0; j < c1; j++) {
            const double *Rj = R + (size_t) j * NR;
            double v = Ri[j];
            for (int k = pt.lo[j]; k < j; k++) v -= Ri[k] * Rj[k];
            Ri[j] = v / Rj[j];
        }
    }
    for (int i = c0; i < c1; i++) {
        double v = pt.rhs[i];
        for (int k = pt.lo[i]; k < i; k++) v -= R[(size_t) i * NR + k] * pt.yr[k];
        pt.yr[i] = v / R[(size_t) i * NR + i];
    }
    return true;
}

// … every separator done: the intrinsics' corner, the rest of the forward substitution, the backward substitution (pt.yr)
inline bool arrow_reduced_end(const ArrowSystem &A, const double *sc, const double *dd, ArrowParts &pt) {
    const size_t nc = A.nc;
    const int P = pt.P, NS = pt.NS, NR = pt.NR;
    double *R = pt.R.data();
    for (int i = 0; i < 9; i++) {
        for (int j = 0; j <= i; j++) R[(size_t) (NS + i) * NR + NS + j] = A.corner[9 * i + j] * sc[nc + i] * sc[nc + j];
        R[(size_t) (NS + i) * NR + NS + i] += dd[nc + i];
        pt.rhs[NS + i] = -A.gi[i] * sc[nc + i];
    }
    for (int p = 0; p < P; p++) {
        if (!pt.ok[p]) return false;
        arrow_reduced_apply(pt, p, -1);
    }
    for (int i = NS; i < NR; i++) {
        double *Ri = R + (size_t) i * NR;
        for (int j = NS; j <= i; j++) {
            const double *Rj = R + (size_t) j * NR;
            double v = Ri[j];
            for (int k = 0; k < j; k++) v -= Ri[k] * Rj[k];
            if (i == j) {
                if (!(v > 0.0)) return false;
                Ri[i] = std::sqrt(v);
            } else {
                Ri[j] = v / Rj[j];
            }
        }
        double v = pt.rhs[i];
        for (int k = 0; k < i; k++) v -= Ri[k] * pt.yr[k];
        pt.yr[i] = v / Ri[i];
    }
    // backward substitution, row-wise (a solved unknown is taken out of the rows above it: contiguous reads of its own row)
    double *yr = pt.yr.data();
    for (int i = NR - 1; i >= 0; i--) {
        const double *Ri = R + (size_t) i * NR;
        const double v = yr[i] / Ri[i];
        yr[i] = v;
        for (int k = pt.lo[i]; k < i; k++) yr[k] -= Ri[k] * v;
    }
    return true;
}

// … and the solution put together: separators and intrinsics from pt.yr, the interiors back-substituted in parallel
inline void arrow_parts_backsub(size_t nc, std::vector<double> &y, ArrowWorkspace &ws, ArrowParts &pt, HostPool *pool, int P, int only_part = -1) {
    const int NS = pt.NS;
    const double *yr = pt.yr.data();
    y.assign(nc + 9, 0.0);
    for (int j = 0; j < 9; j++) y[nc + j] = yr[NS + j];
    for (int s = 0; s + 1 < P; s++)
        for (int i = 0; i < APW; i++) y[pt.a[s] + pt.n[s] + i] = yr[APW * s + i];
    if (only_part >= 0) arrow_part_backsub(ws, pt, only_part, yr, y.data());
    else if (pool) pool->run(P, [&](int p) { arrow_part_backsub(ws, pt, p, yr, y.data()); });
    else
        for (int p = 0; p < P; p++) arrow_part_backsub(ws, pt, p, yr, y.data());
}

// (S A S + diag(dd)) y = -S g with P interiors on the pool's threads; false if not positive definite.
// only_part >= 0 (time-sharded ranks: one interior per rank): this process factorises and back-substitutes that interior only;
// `exchange` sums pt.G (all P blocks; the other ranks' are zero here) and pt.ok over the ranks in between.  A holds this rank's
// rows, the separators' and the intrinsics' rows summed over the ranks; y comes back with this interior, every separator and
// the intrinsics filled in.
inline bool solve_arrow_parts(const ArrowSystem &A, const std::vector<double> &scale, const std::vector<double> &dd, std::vector<double> &y,
                              ArrowWorkspace &ws, ArrowParts &pt, HostPool *pool, int P, int only_part = -1,
                              const std::function<bool(ArrowParts &)> *exchange = nullptr, bool stream_partition = false) {
    const double *sc = scale.data();
    arrow_parts_setup(A.nc, P, ws, pt, stream_partition);
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
    for (int s = 0; s + 1 < P; s++)
        if (!arrow_reduced_separator(A, sc, dd.data(), pt, s)) return false;
    if (!arrow_reduced_end(A, sc, dd.data(), pt)) return false;
    arrow_parts_backsub(A.nc, y, ws, pt, pool, P, only_part);
    return true;
}
