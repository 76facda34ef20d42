// TEST INFRASTRUCTURE ONLY — CPU oracle for the reference driver's adaptive windowing loop and keyframe gate (see
// dbscan_oracle.cpp for the rules: only tests/, smoke() and bench.py's cpu_baseline may use this).
//
// Restates (paths relative to the reference tree):
//   * modules/camera_calibration/event_camera_calib/test/eventCameraCalib.cpp:168-179 — the time range is cut into
//     pieceNum pieces [endTime - step (k+1), endTime - step k), k = 0 .. pieceNum-1 (piece 0 is the LAST in time);
//   * :34-97 MultiProcess::process — a worker pops pieces from the BACK of the vector (so piece pieceNum-1, the earliest in
//     time, is taken first) and runs, per piece, the window loop: duration = (first, first + len); while duration.second <
//     piece end: build the frame, extractFeatures(); on success the body frame's time stamp is the window's middle and
//     tracking->process() decides; accepted -> the window jumps behind itself by frameGap; otherwise, and when the
//     extraction fails, the window slides by one step (when it holds more than FrameEventNumThreshold events or is longer
//     than 3 len) or grows by one step.  The floating-point operations are the reference's, in its order;
//   * modules/core/tracking/src/TrackingBase.cpp:16-46 — the very first frame that reaches process() initialises the map
//     and is accepted; every later one goes through track();
//   * modules/camera_calibration/event_camera_calib/src/EventCalibIni.cpp:23-97 EventCalibIni::track — reference frame =
//     keyframes().lower_bound(time stamp), else the last keyframe; per pattern row the direction (B, -A) of the total
//     least squares line through the row's circle centres (right singular vector of [x y 1] for the smallest singular
//     value), oriented from the row's first to its last circle; theta_i = acos of the normalised dot product of the two
//     frames' directions; accepted iff  nth_element-median(theta) / |time distance| < (5e-4 pi) / MotionTimeStep.
//
// extractFeatures() itself (EventFrame + DBSCAN + pairing + cv::findCirclesGrid) is NOT part of this file: the caller
// supplies it as a callback per window, so that the loop and the gate are checked on their own.  Two gate modes:
//   mode 0  "own piece": the map a window is checked against holds the keyframes of ITS OWN piece only, and the first
//           successful window of every piece is accepted like the reference's very first frame — the deterministic policy of
//           the build (the reference's result depends on its thread schedule: track() reads a map that other workers are
//           inserting into, EventCalibIni.cpp:26-36);
//   mode 1  "reference, one worker": ONE map shared by all pieces, pieces processed one after the other in the reference's
//           pop_back order — what the reference computes when a single worker thread runs (threadNum = 1).
// Parity status: unpinned against a reference run (needs Eigen + OpenCV); Eigen's JacobiSVD is replaced by the
// eigenvectors of the 3x3 Gram matrix (same right singular vectors); std::nth_element is libstdc++'s.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

namespace {

struct Key {
    double time, d0, d1;
    int events;
    std::vector<double> feat;  // [rows*cols][3]
};

// eigenvector of the symmetric 3x3 matrix M for its smallest eigenvalue (cyclic Jacobi rotations)
void smallest_eigenvector(double M[3][3], double v[3]) {
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = M[0][1] * M[0][1] + M[0][2] * M[0][2] + M[1][2] * M[1][2];
        if (off < 1e-300) break;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 3; q++) {
                if (M[p][q] == 0.0) continue;
                const double theta = (M[q][q] - M[p][p]) / (2 * M[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 3; k++) {
                    const double a = M[k][p], b = M[k][q];
                    M[k][p] = c * a - s * b;
                    M[k][q] = s * a + c * b;
                }
                for (int k = 0; k < 3; k++) {
                    const double a = M[p][k], b = M[q][k];
                    M[p][k] = c * a - s * b;
                    M[q][k] = s * a + c * b;
                }
                for (int k = 0; k < 3; k++) {
                    const double a = V[k][p], b = V[k][q];
                    V[k][p] = c * a - s * b;
                    V[k][q] = s * a + c * b;
                }
            }
    }
    int m = 0;
    for (int a = 1; a < 3; a++)
        if (M[a][a] < M[m][m]) m = a;
    for (int k = 0; k < 3; k++) v[k] = V[k][m];
}

// EventCalibIni.cpp:42-58 for one row of one frame
void row_direction(const double *feat, int width, double dir[2]) {
    double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int j = 0; j < width; j++) {
        const double r[3] = {feat[3 * j], feat[3 * j + 1], 1.0};
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) G[a][b] += r[a] * r[b];
    }
    double v[3];
    smallest_eigenvector(G, v);
    dir[0] = v[1];   // (B, -A): svd.matrixV()(1, 2), -svd.matrixV()(0, 2)
    dir[1] = -v[0];
    const double dx = feat[3 * (width - 1)] - feat[0], dy = feat[3 * (width - 1) + 1] - feat[1];
    if (dir[0] * dx + dir[1] * dy < 0) {
        dir[0] = -dir[0];
        dir[1] = -dir[1];
    }
}

bool track_gate(const Key &ref, const Key &cur, int rows, int cols, double motion_time_step) {
    const double duration = std::fabs(cur.time - ref.time);
    std::vector<double> theta;
    for (int i = 0; i < rows; i++) {
        double a[2], b[2];
        row_direction(ref.feat.data() + 3 * (size_t) i * cols, cols, a);
        row_direction(cur.feat.data() + 3 * (size_t) i * cols, cols, b);
        const double c = (a[0] * b[0] + a[1] * b[1]) / (std::sqrt(a[0] * a[0] + a[1] * a[1]) * std::sqrt(b[0] * b[0] + b[1] * b[1]));
        theta.push_back(std::acos(c));   // (EventCalibIni.cpp:75: no clamp — a cosine rounded above 1 gives NaN)
    }
    // (a NaN among the angles: the standard leaves the result unspecified; this IS libstdc++'s std::nth_element, the function the
    // reference calls, so whatever it leaves at the position is the reference's behaviour on this toolchain)
    std::nth_element(theta.begin(), theta.begin() + theta.size() / 2, theta.end());
    return theta[theta.size() / 2] / duration < (5e-4 * M_PI) / motion_time_step;
}

}  // namespace

extern "C" {

// returns 1 when extractFeatures() succeeds on the window [t0, t1]; *events_num = EventFrame::eventsNum();
// features[rows*cols][3] = ordered circles (x, y, radius), written on success
typedef int (*oracle_detect_fn)(void *user, double t0, double t1, int *events_num, double *features);

// Outputs sorted by time stamp (the map's order).  Returns the number of keyframes (also when it exceeds max_keyframes:
// then only the first max_keyframes are written), or -1 on bad arguments.  *windows_evaluated = calls of detect.
int64_t oracle_policy_run(oracle_detect_fn detect, void *user, double start_time, double end_time, int piece_num,
                          double motion_time_step, int frame_event_num_threshold, int rows, int cols, int mode,
                          uint32_t max_keyframes, double *kf_time, double *kf_duration, int32_t *kf_events, double *kf_features,
                          uint64_t *windows_evaluated) {
    if (!detect || piece_num < 1 || rows < 1 || cols < 1) return -1;
    const double len = 3 * motion_time_step, frameGap = 5 * motion_time_step;  // eventCameraCalib.cpp:169-170
    const double step = (end_time - start_time) / piece_num;                  // :174
    std::vector<std::pair<double, double>> timeBoundSet;
    for (int k = 0; k < piece_num; ++k) timeBoundSet.emplace_back(end_time - step * (k + 1), end_time - step * k);  // :177-179
    std::map<double, Key> shared_map;             // mode 1: MapBase::keyframes(), time-keyed
    bool initialised = false;                     // TrackingBase::state
    std::vector<Key> all;
    uint64_t evaluated = 0;
    const size_t M = (size_t) rows * cols;
    while (!timeBoundSet.empty()) {
        const std::pair<double, double> timeBound = timeBoundSet.back();  // :40-41
        timeBoundSet.pop_back();
        std::map<double, Key> own_map;            // mode 0
        std::map<double, Key> &map = mode == 0 ? own_map : shared_map;
        bool &init = initialised;
        if (mode == 0) init = false;
        std::pair<double, double> duration(timeBound.first, timeBound.first + len);  // :49
        while (duration.second < timeBound.second) {                                 // :50
            Key cur;
            cur.feat.assign(3 * M, 0.0);
            int events_num = 0;
            evaluated++;
            const bool found = detect(user, duration.first, duration.second, &events_num, cur.feat.data()) != 0;
            bool accepted = false;
            if (found) {
                cur.time = (duration.first + duration.second) / 2;  // :58
                cur.d0 = duration.first;
                cur.d1 = duration.second;
                cur.events = events_num;
                if (!init) {  // TrackingBase.cpp:18-27 -> initialization(): addFrame, true
                    init = true;
                    accepted = true;
                } else {      // EventCalibIni.cpp:26-36
                    auto itr = map.lower_bound(cur.time);
                    const Key &ref = itr != map.end() ? itr->second : map.rbegin()->second;
                    accepted = track_gate(ref, cur, rows, cols, motion_time_step);
                }
                if (accepted) map[cur.time] = cur;   // MapBase::addFrame: keyed by time stamp
            }
            if (accepted) {  // :60-62
                duration.first = duration.second + frameGap;
                duration.second = duration.first + len;
            } else if (events_num > frame_event_num_threshold || (duration.second - duration.first) > 3 * len) {  // :67-69, :75-77
                duration.first += motion_time_step;
                duration.second = duration.first + len;
            } else {  // :70-71, :78-79
                duration.second += motion_time_step;
            }
        }
        if (mode == 0)
            for (auto &kv : own_map) all.push_back(kv.second);
    }
    if (mode != 0)
        for (auto &kv : shared_map) all.push_back(kv.second);
    std::sort(all.begin(), all.end(), [](const Key &a, const Key &b) { return a.time < b.time; });
    const size_t K = all.size();
    for (size_t k = 0; k < K && k < max_keyframes; k++) {
        kf_time[k] = all[k].time;
        kf_duration[2 * k] = all[k].d0;
        kf_duration[2 * k + 1] = all[k].d1;
        kf_events[k] = all[k].events;
        std::memcpy(kf_features + 3 * M * k, all[k].feat.data(), 3 * M * sizeof(double));
    }
    if (windows_evaluated) *windows_evaluated = evaluated;
    return (int64_t) K;
}

// libstdc++'s own std::nth_element on doubles (operator<), for pinning the product's restatement of it
void oracle_nth_element_f64(double *a, uint32_t n, uint32_t nth) {
    if (a && nth < n) std::nth_element(a, a + nth, a + n);
}

// An input on which std::nth_element(a, a + nth, a + n) runs out of its depth limit 2 lg n and takes its heap-select branch:
// McIlroy's adversary ("A killer adversary for quicksort", 1999) played against the library itself — the values are decided
// while the library compares them, so that every pivot turns out to be nearly the smallest element left.  out[n] = the values.
void oracle_nth_killer(uint32_t n, uint32_t nth, double *out) {
    std::vector<int> val(n), idx(n);
    const int gas = (int) n;
    int nsolid = 0, candidate = 0;
    for (uint32_t i = 0; i < n; i++) {
        val[i] = gas;
        idx[i] = (int) i;
    }
    auto cmp = [&](int x, int y) {
        if (val[x] == gas && val[y] == gas) {
            if (x == candidate) val[x] = nsolid++;
            else val[y] = nsolid++;
        }
        if (val[x] == gas) candidate = x;
        else if (val[y] == gas) candidate = y;
        return val[x] < val[y];
    };
    if (nth < n) std::nth_element(idx.begin(), idx.begin() + nth, idx.end(), cmp);
    for (uint32_t i = 0; i < n; i++) out[i] = (double) (val[i] == gas ? nsolid++ : val[i]);
}

// the gate alone, for known-answer tests: 1 = accepted
int oracle_track_gate(const double *ref_feat, double ref_time, const double *cur_feat, double cur_time, int rows, int cols,
                      double motion_time_step) {
    Key a, b;
    a.time = ref_time;
    b.time = cur_time;
    a.feat.assign(ref_feat, ref_feat + 3 * (size_t) rows * cols);
    b.feat.assign(cur_feat, cur_feat + 3 * (size_t) rows * cols);
    return track_gate(a, b, rows, cols, motion_time_step) ? 1 : 0;
}

}  // extern "C"
