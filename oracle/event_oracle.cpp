// TEST INFRASTRUCTURE ONLY — CPU oracle for the ingest + time-slicing leg (see dbscan_oracle.cpp
// for the rules: only tests/, smoke() and bench.py's cpu_baseline may use this).
//
// Restates (paths relative to the reference tree):
//   * modules/camera_calibration/event/include/opengv2/event/Event.hpp:41-47 — one record is
//     f64 t, f64 x, f64 y, u8 polarity, packed (25 bytes, no padding, little endian).
//   * modules/camera_calibration/event_camera_calib/test/eventCameraCalib.cpp:154-163 — events go
//     into a std::multimap<double, Event_loc_pol>: time-ordered, equal keys keep insertion order.
//   * modules/camera_calibration/event/src/EventFrame.cpp:10-36 — a frame takes the events with
//     lower_bound(t0) <= it < upper_bound(t1) (both ends inclusive), builds one set of unique
//     pixel locations per polarity (equality = operator== on the two doubles), then erases every
//     location present in BOTH sets.
//
// Deliberate, documented difference: the reference emits positiveEvents_/negativeEvents_ in
// libstdc++ unordered_set iteration order (an artefact of EigenMatrixHash, utility.hpp:38-51, and
// of the bucket count).  The build defines a canonical order instead — ascending index of the
// pixel's FIRST occurrence with that polarity inside the window — and this oracle emits that
// order, so DBSCAN pids are comparable 1:1.  Parity status of this file:
// unpinned against a reference run (EventFrame.cpp needs Eigen + OpenCV, absent here); pinned
// only against hand-written known-answer cases in tests/test_oracle_events.py.

#include <cstdint>
#include <cstring>
#include <map>
#include <vector>
#include <algorithm>

namespace {

struct Ev {
    double t, x, y;
    uint8_t p;
};

inline Ev read_record(const uint8_t *rec, uint64_t i) {
    Ev e;
    const uint8_t *r = rec + 25 * i;
    std::memcpy(&e.t, r, 8);
    std::memcpy(&e.x, r + 8, 8);
    std::memcpy(&e.y, r + 16, 8);
    e.p = r[24];
    return e;
}

struct KeyLess {  // strict weak order consistent with operator== on non-NaN doubles (-0.0 == 0.0)
    bool operator()(const std::pair<double, double> &a, const std::pair<double, double> &b) const {
        if (a.first < b.first) return true;
        if (b.first < a.first) return false;
        return a.second < b.second;
    }
};

}  // namespace

extern "C" {

// [lo, hi) = index range of the events with t0 <= t <= t1 in a time-sorted record array
// (multimap lower_bound / upper_bound, EventFrame.cpp:14-15).  Returns 0, or -5 if unsorted.
int oracle_window_bounds(const uint8_t *rec, uint64_t n, double t0, double t1, uint64_t *lo, uint64_t *hi) {
    uint64_t a = 0, b = n;
    while (a < b) {  // first index with t >= t0
        uint64_t m = (a + b) / 2;
        if (read_record(rec, m).t < t0) a = m + 1; else b = m;
    }
    *lo = a;
    b = n;
    while (a < b) {  // first index with t > t1
        uint64_t m = (a + b) / 2;
        if (read_record(rec, m).t <= t1) a = m + 1; else b = m;
    }
    *hi = a;
    return 0;
}

int oracle_check_sorted(const uint8_t *rec, uint64_t n) {
    for (uint64_t i = 1; i < n; i++)
        if (read_record(rec, i).t < read_record(rec, i - 1).t) return -5;
    return 0;
}

// EventFrame constructor on the events [lo, hi).  Outputs (caller sizes everything hi-lo):
//   xy_out      : n_pos positive locations then n_neg negative locations, canonical order
//   event_point : per event in [lo,hi): index of its pixel inside its polarity's list, or -1 when
//                 the pixel was erased because both polarities fired there
// Polarity: the reference reads the byte into a bool (Event.hpp:45); any non-zero byte is positive.
int oracle_event_frame(const uint8_t *rec, uint64_t lo, uint64_t hi, double *xy_out, uint32_t *n_pos,
                       uint32_t *n_neg, int32_t *event_point) {
    typedef std::pair<double, double> Key;
    std::map<Key, uint64_t, KeyLess> first_pos, first_neg;  // pixel -> first event index
    for (uint64_t i = lo; i < hi; i++) {
        Ev e = read_record(rec, i);
        Key k(e.x, e.y);
        if (e.p) first_pos.emplace(k, i); else first_neg.emplace(k, i);
    }
    // erase locations present in both (EventFrame.cpp:24-32)
    for (auto it = first_pos.begin(); it != first_pos.end();) {
        auto f = first_neg.find(it->first);
        if (f == first_neg.end()) {
            ++it;
        } else {
            first_neg.erase(f);
            it = first_pos.erase(it);
        }
    }
    std::vector<std::pair<uint64_t, Key>> P, N;
    for (auto &kv : first_pos) P.emplace_back(kv.second, kv.first);
    for (auto &kv : first_neg) N.emplace_back(kv.second, kv.first);
    std::sort(P.begin(), P.end(), [](auto &a, auto &b) { return a.first < b.first; });
    std::sort(N.begin(), N.end(), [](auto &a, auto &b) { return a.first < b.first; });
    std::map<Key, int32_t, KeyLess> idx_pos, idx_neg;
    for (size_t i = 0; i < P.size(); i++) {
        xy_out[2 * i] = P[i].second.first;
        xy_out[2 * i + 1] = P[i].second.second;
        idx_pos[P[i].second] = (int32_t) i;
    }
    for (size_t i = 0; i < N.size(); i++) {
        xy_out[2 * (P.size() + i)] = N[i].second.first;
        xy_out[2 * (P.size() + i) + 1] = N[i].second.second;
        idx_neg[N[i].second] = (int32_t) i;
    }
    *n_pos = (uint32_t) P.size();
    *n_neg = (uint32_t) N.size();
    for (uint64_t i = lo; i < hi; i++) {
        Ev e = read_record(rec, i);
        Key k(e.x, e.y);
        auto &m = e.p ? idx_pos : idx_neg;
        auto f = m.find(k);
        event_point[i - lo] = (f == m.end()) ? -1 : f->second;
    }
    return 0;
}

}  // extern "C"
