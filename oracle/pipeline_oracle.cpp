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
}
