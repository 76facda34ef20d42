// TEST INFRASTRUCTURE ONLY — CPU oracle for the per-slice circle-candidate extraction (see
// dbscan_oracle.cpp for the rules: only tests/, smoke() and bench.py's cpu_baseline may use this).
//
// Restates, from modules/camera_calibration/event_camera_calib/src/CirclesEventFrame.cpp:
//   :16-33   circleRadiusThreshold_ (max plausible circle radius in pixels, x1.5)
//   :66-72   DBSCAN::Run on the positive then on the negative pixel set
//   :89-117  clusters with fewer than clusterMinSample members are erased, the rest renumbered
//   :127-129 fail when either polarity keeps fewer than rows*cols clusters
//   :136-147 representative of a cluster = its std::nth_element median by Vector2d::norm()
//   :283-311 fitCircle == 0 path: mutual nearest representatives (+ -> -, - -> +), distance gate
//            4*thr^2, circle = (midpoint, half distance), mean |dist - r| / r < 10 / r
//   :361-415 fitCircle(): algebraic least-squares circle through two clusters (3x3 system solved
//            with partial-pivot LU, as Eigen's Matrix3d::lu())   [used by the fitCircle == 1 path]
// The in-cluster element order that feeds nth_element is the reference's BFS order (from
// oracle_dbscan's `members`), and std::nth_element is libstdc++'s, so the representative is the
// reference's even when several members tie in norm.
//
// NOT restated here: cv::findCirclesGrid (vendored OpenCV, :332-336) — the ordering of the
// candidates into the 4x9 grid is a "next" row of SURVEY §8f.
// Third-party behaviour this file has to assume (parity unpinned): nanoflann's 1-NN tie-break
// (:284,:289) — restated as "smallest index among equal squared distances".

#include <cstdint>
#include <cstddef>
#include <cmath>
#include <vector>
#include <algorithm>
#include <limits>

extern "C" int oracle_dbscan(const double *xy, uint32_t n, double eps, uint32_t minpts, int32_t *labels,
                             uint32_t *n_clusters, uint32_t *members, uint32_t *member_off);

namespace {

struct Pt {
    double x, y;
};

inline double norm2(const Pt &p) { return std::sqrt(p.x * p.x + p.y * p.y); }  // Vector2d::norm()

struct Side {
    std::vector<Pt> pts;
    std::vector<std::vector<uint32_t>> clusters;  // kept clusters, reference element order
    std::vector<uint32_t> centres;                // representative pid per kept cluster
    std::vector<int32_t> kept_label;              // per point: renumbered cluster or -1
    std::vector<int32_t> raw_label;               // per point: index into Run()'s Clusters or -1 (Noise)
    uint32_t n_raw = 0;                           // Clusters.size() before the filter
};

void run_side(Side &s, const double *xy, uint32_t n, double eps, uint32_t minpts, uint32_t cluster_min) {
    s.pts.resize(n);
    for (uint32_t i = 0; i < n; i++) s.pts[i] = Pt{xy[2 * i], xy[2 * i + 1]};
    s.kept_label.assign(n, -1);
    if (n == 0) return;
    std::vector<int32_t> labels(n);
    std::vector<uint32_t> members(n), off(n + 2);
    uint32_t nc = 0;
    oracle_dbscan(xy, n, eps, minpts, labels.data(), &nc, members.data(), off.data());
    s.raw_label = labels;
    s.n_raw = nc;
    for (uint32_t c = 0; c < nc; c++) {
        if (off[c + 1] - off[c] < cluster_min) continue;  // :91, :106
        s.clusters.emplace_back(members.begin() + off[c], members.begin() + off[c + 1]);
        for (uint32_t v : s.clusters.back()) s.kept_label[v] = (int32_t) s.clusters.size() - 1;
    }
}

// returns true if some cluster has another member with exactly the representative's norm (then the
// representative depends on the in-cluster order — SURVEY A.6: 0.58 % of clusters)
// tie_out (optional): per kept cluster, 1 if it has such a tie.  rep_override (optional): per kept cluster a member pid to
// use instead (0xFFFFFFFF: keep the reference's) — the parity tests hand in the build's choice among the equal-norm
// members of a tied cluster and then require everything downstream to be identical.
bool medians(Side &s, uint32_t *tie_out = nullptr, const uint32_t *rep_override = nullptr) {
    bool tie = false;
    auto less = [&](uint32_t a, uint32_t b) { return norm2(s.pts[a]) < norm2(s.pts[b]); };
    size_t ci = 0;
    for (auto &c : s.clusters) {
        std::nth_element(c.begin(), c.begin() + c.size() / 2, c.end(), less);  // :141, :145
        uint32_t rep = c[c.size() / 2];
        bool t = false;
        for (uint32_t v : c)
            if (v != rep && norm2(s.pts[v]) == norm2(s.pts[rep])) t = true;
        if (rep_override && rep_override[ci] != 0xFFFFFFFFu) rep = rep_override[ci];
        s.centres.push_back(rep);
        if (tie_out) tie_out[ci] = t ? 1u : 0u;
        tie = tie || t;
        ci++;
    }
    return tie;
}

// 1-NN among centres (nanoflann metric_L2_Simple: sum of squared differences, dim 0 then 1)
size_t nearest(const Side &s, const Pt &q, double *d2_out) {
    size_t best = 0;
    double bd = std::numeric_limits<double>::max();
    for (size_t i = 0; i < s.centres.size(); i++) {
        const Pt &c = s.pts[s.centres[i]];
        const double dx = q.x - c.x, dy = q.y - c.y;
        const double d = dx * dx + dy * dy;
        if (d < bd) {
            bd = d;
            best = i;
        }
    }
    *d2_out = bd;
    return best;
}

// k nearest centres, ascending squared distance (ties: smaller index first — nanoflann unpinned)
void knn(const Side &s, const Pt &q, size_t k, std::vector<size_t> &idx, std::vector<double> &d2) {
    std::vector<std::pair<double, size_t>> all;
    for (size_t i = 0; i < s.centres.size(); i++) {
        const Pt &c = s.pts[s.centres[i]];
        const double dx = q.x - c.x, dy = q.y - c.y;
        all.emplace_back(dx * dx + dy * dy, i);
    }
    std::stable_sort(all.begin(), all.end(), [](auto &a, auto &b) { return a.first < b.first; });
    idx.assign(k, 0);
    d2.assign(k, 0.0);
    for (size_t i = 0; i < k && i < all.size(); i++) {
        idx[i] = all[i].second;
        d2[i] = all[i].first;
    }
}

struct Fit {
    double err, radius, cx, cy;
};

extern "C" int oracle_fit_circle(const double *a_xy, uint32_t na, const double *b_xy, uint32_t nb, double *centre_xy,
                                 double *radius);

// fitCircle + the error of :202-219 for the pair (+ cluster pc, - cluster nc)
Fit fit_pair(const Side &P, size_t pc, const Side &N, size_t nc, double thr) {
    std::vector<double> a, b;
    for (uint32_t e : P.clusters[pc]) {
        a.push_back(P.pts[e].x);
        a.push_back(P.pts[e].y);
    }
    for (uint32_t e : N.clusters[nc]) {
        b.push_back(N.pts[e].x);
        b.push_back(N.pts[e].y);
    }
    double c[2], r;
    oracle_fit_circle(a.data(), (uint32_t) P.clusters[pc].size(), b.data(), (uint32_t) N.clusters[nc].size(), c, &r);
    Fit f{0.0, r, c[0], c[1]};
    const Pt pr = P.pts[P.centres[pc]], nr = N.pts[N.centres[nc]];
    const double ax = pr.x - nr.x, ay = pr.y - nr.y;
    const double approx = std::sqrt(ax * ax + ay * ay) / 2;
    if (r > thr || r > 2 * approx) {  // also false for a NaN radius, as in the reference
        f.err = std::numeric_limits<double>::max();
        return f;
    }
    for (uint32_t e : P.clusters[pc]) {
        const double ex = P.pts[e].x - c[0], ey = P.pts[e].y - c[1];
        f.err += std::abs(std::sqrt(ex * ex + ey * ey) - r);
    }
    for (uint32_t e : N.clusters[nc]) {
        const double ex = N.pts[e].x - c[0], ey = N.pts[e].y - c[1];
        f.err += std::abs(std::sqrt(ex * ex + ey * ey) - r);
    }
    f.err /= (P.clusters[pc].size() + N.clusters[nc].size()) * r;
    return f;
}

}  // namespace

extern "C" {

// CirclesEventFrame.cpp:16-33.  width/height are doubles in the reference (real division).
double oracle_circle_radius_threshold(double width, double height, int rows, int cols, int asymmetric,
                                      double square, double radius) {
    const double lo = std::min(width, height), hi = std::max(width, height);
    const int a = asymmetric ? std::max(rows, 2 * cols) : std::max(rows, cols);
    const int b = asymmetric ? std::min(rows, 2 * cols) : std::min(rows, cols);
    return std::min(hi / a, lo / b) / square * radius * 1.5;
}

// extractFeatures() up to the candidate list, fitCircle == 0 path.
// Outputs: info[4] = {n_candidates, kept + clusters, kept - clusters, status (0 ok, 1 too few
// clusters :127-129 or an empty polarity :62-64; +2 = some cluster has a norm tie at its median)}; cand_pair[2j..] = (+ cluster, - cluster) in
// kept numbering; cand_xyr[3j..] = centre x, centre y, radius; kept_pos/kept_neg per point;
// rep_pos/rep_neg = representative pid per kept cluster (sized n_pos / n_neg).
int oracle_extract_candidates_mode(const double *pos_xy, uint32_t n_pos, const double *neg_xy, uint32_t n_neg,
                                   double eps, uint32_t minpts, uint32_t cluster_min, uint32_t need_clusters,
                                   double radius_thr, int fit_circle, uint32_t knn_num, uint32_t *info,
                                   uint32_t *cand_pair, double *cand_xyr, int32_t *kept_pos, int32_t *kept_neg,
                                   uint32_t *rep_pos, uint32_t *rep_neg);

int oracle_extract_candidates(const double *pos_xy, uint32_t n_pos, const double *neg_xy, uint32_t n_neg, double eps,
                              uint32_t minpts, uint32_t cluster_min, uint32_t need_clusters, double radius_thr,
                              uint32_t *info, uint32_t *cand_pair, double *cand_xyr, int32_t *kept_pos,
                              int32_t *kept_neg, uint32_t *rep_pos, uint32_t *rep_neg) {
    return oracle_extract_candidates_mode(pos_xy, n_pos, neg_xy, n_neg, eps, minpts, cluster_min, need_clusters,
                                          radius_thr, 0, 1, info, cand_pair, cand_xyr, kept_pos, kept_neg, rep_pos,
                                          rep_neg);
}

int oracle_extract_candidates_override(const double *pos_xy, uint32_t n_pos, const double *neg_xy, uint32_t n_neg,
                                       double eps, uint32_t minpts, uint32_t cluster_min, uint32_t need_clusters,
                                       double radius_thr, int fit_circle, uint32_t knn_num, uint32_t *info,
                                       uint32_t *cand_pair, double *cand_xyr, int32_t *kept_pos, int32_t *kept_neg,
                                       uint32_t *rep_pos, uint32_t *rep_neg, uint32_t *tie_pos, uint32_t *tie_neg,
                                       const uint32_t *override_pos, const uint32_t *override_neg);

int oracle_extract_candidates_full(const double *pos_xy, uint32_t n_pos, const double *neg_xy, uint32_t n_neg,
                                   double eps, uint32_t minpts, uint32_t cluster_min, uint32_t need_clusters,
                                   double radius_thr, int fit_circle, uint32_t knn_num, uint32_t *info,
                                   uint32_t *cand_pair, double *cand_xyr, int32_t *kept_pos, int32_t *kept_neg,
                                   uint32_t *rep_pos, uint32_t *rep_neg, uint32_t *tie_pos, uint32_t *tie_neg,
                                   const uint32_t *override_pos, const uint32_t *override_neg, int32_t *raw_pos,
                                   int32_t *raw_neg, uint32_t *n_raw);

// fit_circle == 0: the :283-311 path; fit_circle != 0: the :180-281 path with knn_num neighbours.
int oracle_extract_candidates_mode(const double *pos_xy, uint32_t n_pos, const double *neg_xy, uint32_t n_neg,
                                   double eps, uint32_t minpts, uint32_t cluster_min, uint32_t need_clusters,
                                   double radius_thr, int fit_circle, uint32_t knn_num, uint32_t *info,
                                   uint32_t *cand_pair, double *cand_xyr, int32_t *kept_pos, int32_t *kept_neg,
                                   uint32_t *rep_pos, uint32_t *rep_neg) {
    return oracle_extract_candidates_override(pos_xy, n_pos, neg_xy, n_neg, eps, minpts, cluster_min, need_clusters, radius_thr,
                                              fit_circle, knn_num, info, cand_pair, cand_xyr, kept_pos, kept_neg, rep_pos, rep_neg,
                                              nullptr, nullptr, nullptr, nullptr);
}

// The same with per-cluster tie flags out (tie_pos / tie_neg, sized like rep_pos / rep_neg, optional) and representatives
// handed in for chosen clusters (override_pos / override_neg, optional; see medians()).
int oracle_extract_candidates_override(const double *pos_xy, uint32_t n_pos, const double *neg_xy, uint32_t n_neg,
                                       double eps, uint32_t minpts, uint32_t cluster_min, uint32_t need_clusters,
                                       double radius_thr, int fit_circle, uint32_t knn_num, uint32_t *info,
                                       uint32_t *cand_pair, double *cand_xyr, int32_t *kept_pos, int32_t *kept_neg,
                                       uint32_t *rep_pos, uint32_t *rep_neg, uint32_t *tie_pos, uint32_t *tie_neg,
                                       const uint32_t *override_pos, const uint32_t *override_neg) {
    return oracle_extract_candidates_full(pos_xy, n_pos, neg_xy, n_neg, eps, minpts, cluster_min, need_clusters, radius_thr,
                                          fit_circle, knn_num, info, cand_pair, cand_xyr, kept_pos, kept_neg, rep_pos, rep_neg,
                                          tie_pos, tie_neg, override_pos, override_neg, nullptr, nullptr, nullptr);
}

// The same, and Run()'s own output per polarity as well (raw_pos / raw_neg: index into Clusters or -1 per point;
// n_raw[2]: Clusters.size() of + and -), so that one pass over a window gives the labels AND the candidates.  With the raw
// outputs asked for, a polarity is clustered even when the other one is empty (the reference returns before any
// clustering then, :62-64: info says so as before; the labels are what Run() gives on that set alone).
int oracle_extract_candidates_full(const double *pos_xy, uint32_t n_pos, const double *neg_xy, uint32_t n_neg,
                                   double eps, uint32_t minpts, uint32_t cluster_min, uint32_t need_clusters,
                                   double radius_thr, int fit_circle, uint32_t knn_num, uint32_t *info,
                                   uint32_t *cand_pair, double *cand_xyr, int32_t *kept_pos, int32_t *kept_neg,
                                   uint32_t *rep_pos, uint32_t *rep_neg, uint32_t *tie_pos, uint32_t *tie_neg,
                                   const uint32_t *override_pos, const uint32_t *override_neg, int32_t *raw_pos,
                                   int32_t *raw_neg, uint32_t *n_raw) {
    info[0] = info[1] = info[2] = 0;
    info[3] = 1;
    for (uint32_t i = 0; i < n_pos; i++) kept_pos[i] = -1;
    for (uint32_t i = 0; i < n_neg; i++) kept_neg[i] = -1;
    if (n_raw) n_raw[0] = n_raw[1] = 0;
    Side P, N;
    if (n_pos == 0 || n_neg == 0) {  // :62-64
        if (raw_pos && n_pos) {
            run_side(P, pos_xy, n_pos, eps, minpts, cluster_min);
            for (uint32_t i = 0; i < n_pos; i++) raw_pos[i] = P.raw_label[i];
            if (n_raw) n_raw[0] = P.n_raw;
        }
        if (raw_neg && n_neg) {
            run_side(N, neg_xy, n_neg, eps, minpts, cluster_min);
            for (uint32_t i = 0; i < n_neg; i++) raw_neg[i] = N.raw_label[i];
            if (n_raw) n_raw[1] = N.n_raw;
        }
        return 0;
    }
    run_side(P, pos_xy, n_pos, eps, minpts, cluster_min);
    run_side(N, neg_xy, n_neg, eps, minpts, cluster_min);
    if (raw_pos)
        for (uint32_t i = 0; i < n_pos; i++) raw_pos[i] = P.raw_label[i];
    if (raw_neg)
        for (uint32_t i = 0; i < n_neg; i++) raw_neg[i] = N.raw_label[i];
    if (n_raw) {
        n_raw[0] = P.n_raw;
        n_raw[1] = N.n_raw;
    }
    for (uint32_t i = 0; i < n_pos; i++) kept_pos[i] = P.kept_label[i];
    for (uint32_t i = 0; i < n_neg; i++) kept_neg[i] = N.kept_label[i];
    info[1] = (uint32_t) P.clusters.size();
    info[2] = (uint32_t) N.clusters.size();
    if (P.clusters.size() < need_clusters || N.clusters.size() < need_clusters) return 0;  // :127-129
    info[3] = 0;
    const bool tp = medians(P, tie_pos, override_pos);
    const bool tn = medians(N, tie_neg, override_neg);
    if (tp || tn) info[3] |= 2u;  // bit 1: representative is order dependent in some cluster
    for (size_t i = 0; i < P.centres.size(); i++) rep_pos[i] = P.centres[i];
    for (size_t i = 0; i < N.centres.size(); i++) rep_neg[i] = N.centres[i];
    uint32_t nc = 0;
    if (fit_circle) {  // :180-281
        const size_t K = knn_num;
        const double gate = 4 * radius_thr * radius_thr;
        std::vector<size_t> n_idx, p_idx;
        std::vector<double> d2;
        for (size_t pi = 0; pi < P.centres.size(); pi++) {
            size_t real = K;
            knn(N, P.pts[P.centres[pi]], K, n_idx, d2);
            for (size_t oi = 0; oi < K; oi++)
                if (d2[oi] > d2[0] * 4 || d2[oi] > gate) {
                    real = oi;
                    break;
                }
            if (real == 0) continue;
            std::vector<Fit> fits(real);
            for (size_t j = 0; j < real; j++) fits[j] = fit_pair(P, pi, N, n_idx[j], radius_thr);
            size_t nmin = 0;
            for (size_t j = 1; j < real; j++)
                if (fits[j].err < fits[nmin].err) nmin = j;  // std::min_element: first minimum
            if (!(fits[nmin].err < 2 / fits[nmin].radius)) continue;
            real = K;
            knn(P, N.pts[N.centres[n_idx[nmin]]], K, p_idx, d2);
            for (size_t oi = 0; oi < K; oi++)
                if (d2[oi] > d2[0] * 4 || d2[oi] > gate) {
                    real = oi;
                    break;
                }
            if (real == 0) continue;
            std::vector<Fit> back(real);
            for (size_t i = 0; i < real; i++) back[i] = fit_pair(P, p_idx[i], N, n_idx[nmin], radius_thr);
            size_t pmin = 0;
            for (size_t i = 1; i < real; i++)
                if (back[i].err < back[pmin].err) pmin = i;
            if (p_idx[pmin] == pi) {
                cand_pair[2 * nc] = (uint32_t) pi;
                cand_pair[2 * nc + 1] = (uint32_t) n_idx[nmin];
                cand_xyr[3 * nc] = back[pmin].cx;
                cand_xyr[3 * nc + 1] = back[pmin].cy;
                cand_xyr[3 * nc + 2] = back[pmin].radius;
                nc++;
            }
        }
        info[0] = nc;
        return 0;
    }
    for (size_t pi = 0; pi < P.centres.size(); pi++) {  // :283-311
        double d2;
        const Pt pc = P.pts[P.centres[pi]];
        const size_t ni = nearest(N, pc, &d2);
        if (d2 > 4 * radius_thr * radius_thr) continue;
        const Pt ncn = N.pts[N.centres[ni]];
        const size_t back = nearest(P, ncn, &d2);
        if (back != pi) continue;
        const Pt centre{(pc.x + ncn.x) / 2, (pc.y + ncn.y) / 2};
        const double ddx = pc.x - ncn.x, ddy = pc.y - ncn.y;
        const double r = std::sqrt(ddx * ddx + ddy * ddy) / 2;
        double fit = 0;
        for (uint32_t e : P.clusters[pi]) {
            const double ex = P.pts[e].x - centre.x, ey = P.pts[e].y - centre.y;
            fit += std::abs(std::sqrt(ex * ex + ey * ey) - r);
        }
        for (uint32_t e : N.clusters[ni]) {
            const double ex = N.pts[e].x - centre.x, ey = N.pts[e].y - centre.y;
            fit += std::abs(std::sqrt(ex * ex + ey * ey) - r);
        }
        fit /= (P.clusters[pi].size() + N.clusters[ni].size()) * r;
        if (fit < 10 / r) {
            cand_pair[2 * nc] = (uint32_t) pi;
            cand_pair[2 * nc + 1] = (uint32_t) ni;
            cand_xyr[3 * nc] = centre.x;
            cand_xyr[3 * nc + 1] = centre.y;
            cand_xyr[3 * nc + 2] = r;
            nc++;
        }
    }
    info[0] = nc;
    return 0;
}

// fitCircle() :361-415 — algebraic circle through the union of two point lists (summation order:
// first list then second, as given).  Solves with partial-pivot LU like Eigen's Matrix3d::lu().
int oracle_fit_circle(const double *a_xy, uint32_t na, const double *b_xy, uint32_t nb, double *centre_xy,
                      double *radius) {
    double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0, sxxx = 0, syyy = 0, sxyy = 0, sxxy = 0;
    auto add = [&](double x, double y) {
        sx += x;
        sy += y;
        const double xx = x * x, yy = y * y, xy = x * y;
        sxx += xx;
        syy += yy;
        sxy += xy;
        sxxx += xx * x;
        syyy += yy * y;
        sxyy += xy * y;
        sxxy += x * xy;
    };
    for (uint32_t i = 0; i < na; i++) add(a_xy[2 * i], a_xy[2 * i + 1]);
    for (uint32_t i = 0; i < nb; i++) add(b_xy[2 * i], b_xy[2 * i + 1]);
    double A[3][4] = {{2 * sx, 2 * sy, (double) (na + nb), sxx + syy},
                      {2 * sxx, 2 * sxy, sx, sxxx + sxyy},
                      {2 * sxy, 2 * syy, sy, sxxy + syyy}};
    for (int c = 0; c < 3; c++) {
        int piv = c;
        for (int r = c + 1; r < 3; r++)
            if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
        if (piv != c)
            for (int k = 0; k < 4; k++) std::swap(A[c][k], A[piv][k]);
        for (int r = c + 1; r < 3; r++) {
            const double f = A[r][c] / A[c][c];
            for (int k = c; k < 4; k++) A[r][k] -= f * A[c][k];
        }
    }
    double x[3];
    for (int r = 2; r >= 0; r--) {
        double v = A[r][3];
        for (int k = r + 1; k < 3; k++) v -= A[r][k] * x[k];
        x[r] = v / A[r][r];
    }
    centre_xy[0] = x[0];
    centre_xy[1] = x[1];
    *radius = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2]);
    return 0;
}

}  // extern "C"
