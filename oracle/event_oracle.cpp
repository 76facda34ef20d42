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
// Two element orders are provided:
//   * oracle_event_frame_ref — THE REFERENCE'S ORDER: positiveEvents_/negativeEvents_ are emitted in
//     the iteration order of a real std::unordered_set whose hash restates EigenMatrixHash
//     (core/utility/include/opengv2/utility/utility.hpp:38-51: boost-style hash_combine of
//     std::hash<double> over the two coordinates), built and erased exactly as EventFrame.cpp:12-32
//     does.  Exact by construction on the toolchain it is compiled with; pinned to libstdc++ of
//     g++ 11.4 (GLIBCXX_3.4.29, the image's) — the order is an artefact of that library's
//     _Hashtable (prime bucket counts, insert-at-bucket-front, rehash walk), see
//     oracle_event_frame_model below for the restated rules.  DBSCAN labels depend on this order
//     (insertion-order kd-tree, kdtree.cpp:128-131,169), so `.bin`-level parity needs it.
//   * oracle_event_frame — the build's CANONICAL order (ascending index of the pixel's FIRST
//     occurrence with that polarity inside the window), the order=canonical mode of the slicer.
// Parity status of this file: EventFrame.cpp itself is unbuildable here (needs Eigen + OpenCV);
// the container semantics are not restated but executed (std::unordered_set, std::hash<double>);
// the surrounding loop is pinned by the hand-written known-answer cases in
// tests/test_oracle_events.py and the committed fixtures tests/golden/eventframe_order_*.npz.

#include <cstdint>
#include <cstring>
#include <map>
#include <vector>
#include <algorithm>
#include <array>
#include <functional>
#include <unordered_set>
#include <unordered_map>

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

// ---- reference order (EventFrame.cpp:10-36 with the real container) -------------------------------------------------
namespace {

typedef std::array<double, 2> Px;

// utility.hpp:38-51 for a Vector2d: size() == 2, data() = {x, y}.  (0x9e3779b9 is an unsigned int literal; the sum is
// formed in size_t.)  operator() is deliberately NOT noexcept, as the reference's: libstdc++ then caches hash codes.
struct RefPixelHash {
    std::size_t operator()(const Px &m) const {
        std::size_t seed = 0;
        for (std::size_t i = 0; i < 2; ++i) seed ^= std::hash<double>()(m[i]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return seed;
    }
};
struct RefPixelEq {  // std::equal_to<> -> Eigen operator== -> cwiseEqual().all()
    bool operator()(const Px &a, const Px &b) const { return a[0] == b[0] && a[1] == b[1]; }
};
typedef std::unordered_set<Px, RefPixelHash, RefPixelEq> RefSet;

// libstdc++'s 64-bit std::hash<double> restated: 0 for +-0.0, else _Hash_bytes(&v, 8, 0xc70f6907) — the Murmur-style
// mix of libstdc++-v3/libsupc++/hash_bytes.cc.  tests/test_oracle_events.py compares it with the real std::hash.
inline uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }
inline uint64_t hash_double_restated(double v) {
    if (v == 0.0) return 0;
    const uint64_t mul = (0xc6a4a793ull << 32) + 0x5bd1e995ull;
    uint64_t bits;
    std::memcpy(&bits, &v, 8);
    uint64_t h = 0xc70f6907ull ^ (8 * mul);
    const uint64_t data = shift_mix(bits * mul) * mul;
    h ^= data;
    h *= mul;
    h = shift_mix(h) * mul;
    h = shift_mix(h);
    return h;
}
inline uint64_t pixel_hash_restated(double x, double y) {
    uint64_t seed = 0;
    seed ^= hash_double_restated(x) + 0x9e3779b9ull + (seed << 6) + (seed >> 2);
    seed ^= hash_double_restated(y) + 0x9e3779b9ull + (seed << 6) + (seed >> 2);
    return seed;
}

}  // namespace

extern "C" {

uint64_t oracle_pixel_hash(double x, double y) { return RefPixelHash()(Px{x, y}); }
uint64_t oracle_pixel_hash_restated(double x, double y) { return pixel_hash_restated(x, y); }

// bucket_count() of a real unordered_set after each of n insertions of distinct keys (out[k] = count after k+1 keys)
void oracle_bucket_counts(uint32_t n, uint64_t *out) {
    std::unordered_set<uint64_t> s;
    for (uint32_t k = 0; k < n; k++) {
        s.insert(k);
        out[k] = s.bucket_count();
    }
}
// the growth steps themselves, straight from the library's policy object: next bucket count for a request of n
uint64_t oracle_next_bkt(uint64_t n) {
    std::__detail::_Prime_rehash_policy pol;
    return pol._M_next_bkt(n);
}

// EventFrame constructor on the events [lo, hi), the reference's element order.  Same outputs as oracle_event_frame.
int oracle_event_frame_ref(const uint8_t *rec, uint64_t lo, uint64_t hi, double *xy_out, uint32_t *n_pos,
                           uint32_t *n_neg, int32_t *event_point) {
    RefSet positiveEvents, negativeEvents;
    for (uint64_t i = lo; i < hi; i++) {  // EventFrame.cpp:14-21
        Ev e = read_record(rec, i);
        if (e.p) positiveEvents.insert(Px{e.x, e.y}); else negativeEvents.insert(Px{e.x, e.y});
    }
    for (auto itr = positiveEvents.begin(); itr != positiveEvents.end();) {  // EventFrame.cpp:24-32
        auto found = negativeEvents.find(*itr);
        if (found == negativeEvents.end()) {
            itr++;
        } else {
            negativeEvents.erase(found);
            itr = positiveEvents.erase(itr);
        }
    }
    std::unordered_map<Px, int32_t, RefPixelHash, RefPixelEq> idx_pos, idx_neg;
    uint32_t k = 0;
    for (const Px &p : positiveEvents) {  // EventFrame.cpp:34
        xy_out[2 * k] = p[0];
        xy_out[2 * k + 1] = p[1];
        idx_pos[p] = (int32_t) k++;
    }
    *n_pos = k;
    uint32_t j = 0;
    for (const Px &p : negativeEvents) {  // EventFrame.cpp:35
        xy_out[2 * (k + j)] = p[0];
        xy_out[2 * (k + j) + 1] = p[1];
        idx_neg[p] = (int32_t) j++;
    }
    *n_neg = j;
    for (uint64_t i = lo; i < hi; i++) {
        Ev e = read_record(rec, i);
        auto &m = e.p ? idx_pos : idx_neg;
        auto f = m.find(Px{e.x, e.y});
        event_point[i - lo] = (f == m.end()) ? -1 : f->second;
    }
    return 0;
}

// The same order from the RULES the HIP slicer follows (no std container): what libstdc++'s _Hashtable does to the
// singly linked node list, restated.
//   * unique keys in first-occurrence order k = 0, 1, ... ; hash h_k = pixel_hash_restated.
//   * bucket counts: 13 for the first 13 keys; whenever key number B+1 arrives with B buckets, the table is first
//     rehashed to the next listed prime >= 2 B (29, 59, 127, 257, 541, 1109, 2357, 5087, ...).
//   * insert with B buckets: if the list holds a node of bucket h % B the new node goes in FRONT of that bucket's run,
//     else to the front of the whole list.  Rehash: walk the list in order and re-insert every node by the same rule.
//   * hence per epoch (constant B): sequence = old list order ++ new keys in arrival order; new list = the sequence
//     sorted by (first appearance of the element's bucket in the sequence, own position) and then REVERSED.
//   * erase keeps the relative order of what stays.
// bucket_steps: the growth list (oracle_next_bkt), n_steps entries starting with 13.
int oracle_event_frame_model(const uint8_t *rec, uint64_t lo, uint64_t hi, const uint64_t *bucket_steps, uint32_t n_steps,
                             double *xy_out, uint32_t *n_pos, uint32_t *n_neg, int32_t *event_point) {
    const uint64_t n = hi - lo;
    std::vector<int64_t> rep(n, -1);  // per event: index (in uniq[pol]) of its pixel's first occurrence
    struct U { double x, y; uint64_t h; bool erased; };
    std::vector<U> uniq[2];
    for (uint64_t i = 0; i < n; i++) {
        Ev e = read_record(rec, lo + i);
        auto &u = uniq[e.p ? 1 : 0];
        int64_t f = -1;
        for (size_t q = 0; q < u.size(); q++)
            if (u[q].x == e.x && u[q].y == e.y) { f = (int64_t) q; break; }
        if (f < 0) {
            f = (int64_t) u.size();
            u.push_back(U{e.x, e.y, pixel_hash_restated(e.x, e.y), false});
        }
        rep[i] = f;
    }
    for (auto &a : uniq[1])
        for (auto &b : uniq[0])
            if (a.x == b.x && a.y == b.y) a.erased = b.erased = true;
    std::vector<int32_t> final_idx[2];
    uint32_t counts[2] = {0, 0};
    for (int pol = 1; pol >= 0; pol--) {
        auto &u = uniq[pol];
        const size_t m = u.size();
        std::vector<uint32_t> list;  // current node order (indices into u)
        size_t done = 0;
        for (uint32_t e = 0; done < m; e++) {
            if (e >= n_steps) return -4;
            const uint64_t B = bucket_steps[e];
            const size_t upto = std::min<size_t>(m, (size_t) B);
            std::vector<uint32_t> seq(list);
            for (size_t k = done; k < upto; k++) seq.push_back((uint32_t) k);
            std::unordered_map<uint64_t, size_t> first;  // bucket -> first position in seq
            for (size_t q = 0; q < seq.size(); q++) first.emplace(u[seq[q]].h % B, q);
            std::vector<std::pair<std::pair<size_t, size_t>, uint32_t>> keyed;
            for (size_t q = 0; q < seq.size(); q++) keyed.push_back({{first[u[seq[q]].h % B], q}, seq[q]});
            std::sort(keyed.begin(), keyed.end());
            list.clear();
            for (size_t q = keyed.size(); q-- > 0;) list.push_back(keyed[q].second);
            done = upto;
        }
        final_idx[pol].assign(m, -1);
        uint32_t c = 0;
        double *out = xy_out + (pol == 1 ? 0 : 2 * (size_t) counts[1]);
        for (uint32_t id : list) {
            if (u[id].erased) continue;
            out[2 * c] = u[id].x;
            out[2 * c + 1] = u[id].y;
            final_idx[pol][id] = (int32_t) c++;
        }
        counts[pol] = c;
    }
    *n_pos = counts[1];
    *n_neg = counts[0];
    for (uint64_t i = 0; i < n; i++) {
        Ev e = read_record(rec, lo + i);
        event_point[i] = final_idx[e.p ? 1 : 0][rep[i]];
    }
    return 0;
}

}  // extern "C"
