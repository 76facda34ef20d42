// TEST INFRASTRUCTURE ONLY — the CPU baseline loop: for every window, the EventFrame constructor
// followed by one DBSCAN::Run per polarity, i.e. the body of the reference's worker loop
// (event_camera_calib/test/eventCameraCalib.cpp:49-56 -> EventFrame.cpp:10-36 ->
// CirclesEventFrame.cpp:66-72) on the oracle restatements.  Single thread.  Used by bench.py's
// cpu_baseline leg and by the parity tests as a batch checker.
#include <cstdint>
#include <vector>
#include <cstddef>

extern "C" {
int oracle_window_bounds(const uint8_t *rec, uint64_t n, double t0, double t1, uint64_t *lo, uint64_t *hi);
int oracle_event_frame(const uint8_t *rec, uint64_t lo, uint64_t hi, double *xy_out, uint32_t *n_pos, uint32_t *n_neg,
                       int32_t *event_point);
int oracle_dbscan(const double *xy, uint32_t n, double eps, uint32_t minpts, int32_t *labels, uint32_t *n_clusters,
                  uint32_t *members, uint32_t *member_off);

// Returns the number of events covered by the windows; *n_clusters_total accumulates cluster counts
// (so the work cannot be optimised away); labels are discarded.
uint64_t oracle_detect_windows(const uint8_t *rec, uint64_t n, const double *t0, const double *t1, uint32_t S,
                               double eps, uint32_t minpts, uint64_t *n_clusters_total) {
    uint64_t events = 0, clusters = 0;
    std::vector<double> xy;
    std::vector<int32_t> ep, labels;
    for (uint32_t s = 0; s < S; s++) {
        uint64_t lo, hi;
        oracle_window_bounds(rec, n, t0[s], t1[s], &lo, &hi);
        const uint64_t m = hi - lo;
        if (m == 0) continue;
        xy.resize(2 * m);
        ep.resize(m);
        labels.resize(m);
        uint32_t np = 0, nn = 0, nc = 0;
        oracle_event_frame(rec, lo, hi, xy.data(), &np, &nn, ep.data());
        if (np) {
            oracle_dbscan(xy.data(), np, eps, minpts, labels.data(), &nc, nullptr, nullptr);
            clusters += nc;
        }
        if (nn) {
            oracle_dbscan(xy.data() + 2 * (std::size_t) np, nn, eps, minpts, labels.data(), &nc, nullptr, nullptr);
            clusters += nc;
        }
        events += m;
    }
    *n_clusters_total = clusters;
    return events;
}
}
