// TEST INFRASTRUCTURE ONLY — the CPU baseline loop: for every window, the EventFrame constructor
// followed by extractFeatures up to the candidate circles (two DBSCAN::Run, cluster filter, medians,
// pairing), i.e. the body of the reference's worker loop (event_camera_calib/test/
// eventCameraCalib.cpp:49-56 -> EventFrame.cpp:10-36 -> CirclesEventFrame.cpp:61-312) on the
// oracle restatements, without the findCirclesGrid call.  Single thread.  Used by bench.py's
// cpu_baseline leg and by the parity tests as a batch checker.
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>
#include <cstddef>

extern "C" {
int oracle_window_bounds(const uint8_t *rec, uint64_t n, double t0, double t1, uint64_t *lo, uint64_t *hi);
int oracle_event_frame_ref(const uint8_t *rec, uint64_t lo, uint64_t hi, double *xy_out, uint32_t *n_pos, uint32_t *n_neg,
                           int32_t *event_point);
int oracle_extract_candidates(const double *pos_xy, uint32_t n_pos, const double *neg_xy, uint32_t n_neg, double eps,
                              uint32_t minpts, uint32_t cluster_min, uint32_t need_clusters, double radius_thr,
                              uint32_t *info, uint32_t *cand_pair, double *cand_xyr, int32_t *kept_pos,
                              int32_t *kept_neg, uint32_t *rep_pos, uint32_t *rep_neg);

int oracle_extract_candidates_full(const double *pos_xy, uint32_t n_pos, const double *neg_xy, uint32_t n_neg,
                                   double eps, uint32_t minpts, uint32_t cluster_min, uint32_t need_clusters,
                                   double radius_thr, int fit_circle, uint32_t knn_num, uint32_t *info,
                                   uint32_t *cand_pair, double *cand_xyr, int32_t *kept_pos, int32_t *kept_neg,
                                   uint32_t *rep_pos, uint32_t *rep_neg, uint32_t *tie_pos, uint32_t *tie_neg,
                                   const uint32_t *override_pos, const uint32_t *override_neg, int32_t *raw_pos,
                                   int32_t *raw_neg, uint32_t *n_raw);

// Returns the number of events covered by the windows; *n_clusters_total accumulates cluster counts
// (so the work cannot be optimised away); labels are discarded.
uint64_t oracle_detect_windows(const uint8_t *rec, uint64_t n, const double *t0, const double *t1, uint32_t S,
                               double eps, uint32_t minpts, uint32_t cluster_min, uint32_t need_clusters,
                               double radius_thr, uint64_t *n_clusters_total) {
    uint64_t events = 0, clusters = 0;
    std::vector<double> xy, cxyr;
    std::vector<int32_t> ep, kp, kn;
    std::vector<uint32_t> pair, rp, rn;
    for (uint32_t s = 0; s < S; s++) {
        uint64_t lo, hi;
        oracle_window_bounds(rec, n, t0[s], t1[s], &lo, &hi);
        const uint64_t m = hi - lo;
        if (m == 0) continue;
        xy.resize(2 * m);
        ep.resize(m);
        kp.resize(m);
        kn.resize(m);
        rp.resize(m);
        rn.resize(m);
        pair.resize(2 * m);
        cxyr.resize(3 * m);
        uint32_t np = 0, nn = 0, info[4];
        oracle_event_frame_ref(rec, lo, hi, xy.data(), &np, &nn, ep.data());   // the reference's containers and order
        oracle_extract_candidates(xy.data(), np, xy.data() + 2 * (std::size_t) np, nn, eps, minpts, cluster_min,
                                  need_clusters, radius_thr, info, pair.data(), cxyr.data(), kp.data(), kn.data(),
                                  rp.data(), rn.data());
        clusters += info[1] + info[2];
        events += m;
    }
    *n_clusters_total = clusters;
    return events;
}

// The reference driver's threading (event_camera_calib/test/eventCameraCalib.cpp:172-190): T = hardware threads - 2
// workers, the time range cut into 5 T pieces, every worker runs whole pieces.  Here the tiled windows are cut into
// 5 T contiguous pieces handed out from an atomic counter (the reference hands piece k to thread k % T).
uint64_t oracle_detect_windows_mt(const uint8_t *rec, uint64_t n, const double *t0, const double *t1, uint32_t S, double eps,
                                  uint32_t minpts, uint32_t cluster_min, uint32_t need_clusters, double radius_thr,
                                  uint32_t n_threads, uint64_t *n_clusters_total) {
    if (n_threads < 1) n_threads = 1;
    const uint32_t pieces = 5 * n_threads;
    std::atomic<uint32_t> next{0};
    std::atomic<uint64_t> events{0}, clusters{0};
    auto work = [&]() {
        for (;;) {
            const uint32_t k = next.fetch_add(1);
            if (k >= pieces) return;
            const uint32_t lo = (uint32_t) ((uint64_t) S * k / pieces), hi = (uint32_t) ((uint64_t) S * (k + 1) / pieces);
            if (hi == lo) continue;
            uint64_t c = 0;
            const uint64_t e = oracle_detect_windows(rec, n, t0 + lo, t1 + lo, hi - lo, eps, minpts, cluster_min, need_clusters,
                                                     radius_thr, &c);
            events += e;
            clusters += c;
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < n_threads; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    *n_clusters_total = clusters.load();
    return events.load();
}
// The whole result of every window, laid out as the device pipeline lays it out ("slots"): window s owns
// the slots [win_base[s], win_base[s] + events of the window) (win_base = exclusive scan of the windows' event counts,
// handed in, S + 1 entries); its positive points take the first seg_cnt[2s] slots, the negative ones the next
// seg_cnt[2s+1].  Per slot: xy (EventFrame.cpp:34-35, the reference's container order), labels (DBSCAN::Run),
// kept_labels (CirclesEventFrame.cpp:89-117); rep per kept cluster from the polarity's first slot on; cand_pair /
// cand_xyr per candidate from the window's first slot on; event_point per event of the window.  def_* are byte masks
// saying which slots the reference's semantics define: def_pts (a point lives there), def_kept (… and the window
// reached the cluster filter, i.e. both polarities non-empty), def_rep, def_cand (status 0 only).  win_info[s] =
// {candidates, kept +, kept -, status (bit 0 only)}; tie[s] = 1 when a median of the window is order dependent.
// Threads as oracle_detect_windows_mt.  Returns the events covered, or (uint64_t) -1 when a window does not fit its slots.
uint64_t oracle_detect_windows_full_mt(const uint8_t *rec, uint64_t n, const double *t0, const double *t1, uint32_t S, double eps,
                                       uint32_t minpts, uint32_t cluster_min, uint32_t need_clusters, double radius_thr,
                                       int fit_circle, uint32_t knn_num, uint32_t n_threads, const uint64_t *win_base,
                                       uint64_t *win_lo, uint64_t *win_hi, uint32_t *seg_cnt, uint32_t *n_clusters,
                                       uint32_t *win_info, uint8_t *tie, double *xy, int32_t *event_point, int32_t *labels,
                                       int32_t *kept_labels, uint32_t *rep, uint32_t *cand_pair, double *cand_xyr,
                                       uint8_t *def_pts, uint8_t *def_kept, uint8_t *def_rep, uint8_t *def_cand) {
    if (n_threads < 1) n_threads = 1;
    std::atomic<uint32_t> next{0};
    std::atomic<uint64_t> events{0};
    std::atomic<int> bad{0};
    const uint32_t grain = 16;
    auto work = [&]() {
        std::vector<double> wxy, cxyr;
        std::vector<int32_t> ep, kp, kn, lp, ln;
        std::vector<uint32_t> pair, rp, rn, tp, tn;
        for (;;) {
            const uint32_t first = next.fetch_add(grain);
            if (first >= S) return;
            for (uint32_t s = first; s < S && s < first + grain; s++) {
                uint64_t lo, hi;
                oracle_window_bounds(rec, n, t0[s], t1[s], &lo, &hi);
                win_lo[s] = lo;
                win_hi[s] = hi;
                const uint64_t m = hi - lo, base = win_base[s];
                seg_cnt[2 * s] = seg_cnt[2 * s + 1] = 0;
                n_clusters[2 * s] = n_clusters[2 * s + 1] = 0;
                win_info[4 * s] = win_info[4 * s + 1] = win_info[4 * s + 2] = 0;
                win_info[4 * s + 3] = 1;
                tie[s] = 0;
                if (m == 0) continue;
                if (base + m > win_base[s + 1]) {
                    bad = 1;
                    continue;
                }
                const std::size_t mm = (std::size_t) m;
                wxy.resize(2 * mm);
                ep.resize(mm);
                kp.resize(mm);
                kn.resize(mm);
                lp.resize(mm);
                ln.resize(mm);
                rp.resize(mm);
                rn.resize(mm);
                tp.resize(mm);
                tn.resize(mm);
                pair.resize(2 * mm);
                cxyr.resize(3 * mm);
                uint32_t np = 0, nn = 0, info[4], nraw[2];
                oracle_event_frame_ref(rec, lo, hi, wxy.data(), &np, &nn, ep.data());
                oracle_extract_candidates_full(wxy.data(), np, wxy.data() + 2 * (std::size_t) np, nn, eps, minpts, cluster_min,
                                               need_clusters, radius_thr, fit_circle, knn_num, info, pair.data(), cxyr.data(),
                                               kp.data(), kn.data(), rp.data(), rn.data(), tp.data(), tn.data(), nullptr, nullptr,
                                               lp.data(), ln.data(), nraw);
                seg_cnt[2 * s] = np;
                seg_cnt[2 * s + 1] = nn;
                n_clusters[2 * s] = nraw[0];
                n_clusters[2 * s + 1] = nraw[1];
                win_info[4 * s] = info[0];
                win_info[4 * s + 1] = info[1];
                win_info[4 * s + 2] = info[2];
                win_info[4 * s + 3] = info[3] & 1u;
                tie[s] = (info[3] & 2u) ? 1 : 0;
                const bool both = np > 0 && nn > 0, ok = (info[3] & 1u) == 0;
                for (std::size_t i = 0; i < mm; i++) event_point[base + i] = ep[i];
                for (uint32_t i = 0; i < np + nn; i++) {
                    const std::size_t d = (std::size_t) base + i;
                    xy[2 * d] = wxy[2 * (std::size_t) i];
                    xy[2 * d + 1] = wxy[2 * (std::size_t) i + 1];
                    labels[d] = i < np ? lp[i] : ln[i - np];
                    kept_labels[d] = i < np ? kp[i] : kn[i - np];
                    def_pts[d] = 1;
                    def_kept[d] = both ? 1 : 0;
                }
                if (ok) {
                    for (uint32_t c = 0; c < info[1]; c++) {
                        rep[base + c] = rp[c];
                        def_rep[base + c] = 1;
                    }
                    for (uint32_t c = 0; c < info[2]; c++) {
                        rep[base + np + c] = rn[c];
                        def_rep[base + np + c] = 1;
                    }
                    for (uint32_t c = 0; c < info[0]; c++) {
                        cand_pair[2 * (base + c)] = pair[2 * c];
                        cand_pair[2 * (base + c) + 1] = pair[2 * c + 1];
                        cand_xyr[3 * (base + c)] = cxyr[3 * c];
                        cand_xyr[3 * (base + c) + 1] = cxyr[3 * c + 1];
                        cand_xyr[3 * (base + c) + 2] = cxyr[3 * c + 2];
                        def_cand[base + c] = 1;
                    }
                }
                events += m;
            }
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < n_threads; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    if (bad.load()) return (uint64_t) -1;
    return events.load();
}
}
