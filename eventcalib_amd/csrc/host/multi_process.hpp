// Counterpart of the reference driver's per-piece adaptive windowing loop and keyframe gate
//   MultiProcess::process          event_camera_calib/test/eventCameraCalib.cpp:34-97
//   piece construction             event_camera_calib/test/eventCameraCalib.cpp:168-179
//   EventCalibIni::track           event_camera_calib/src/EventCalibIni.cpp:23-97
// on top of libecal.so.  The reference runs one worker thread per time piece and one frame object per
// window; here every piece advances in lockstep and each step is ONE batched GPU call over the current
// window of every active piece (detect_windows), so the data-dependent control flow costs a handful of
// launches per step instead of a kernel per window.
//
// Deterministic policy (the reference's result depends on its thread schedule: track() consults a map that
// other workers are inserting into, EventCalibIni.cpp:26-36): a window is gated against the previous
// keyframe of ITS OWN piece; the first successful window of a piece is accepted as the reference accepts
// its very first frame (TrackingBase::process -> initialization).
#ifndef ECAL_HOST_MULTI_PROCESS_HPP_
#define ECAL_HOST_MULTI_PROCESS_HPP_

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>

#include "circles_event_frame.hpp"

namespace opengv2 {

struct KeyFrame {
    double timeStamp;                        // (duration.first + duration.second) / 2, eventCameraCalib.cpp:58
    std::pair<double, double> duration;
    int eventsNum;
    std::vector<CirclesEventFrame::CalibCircle> features;  // grid order
};

namespace detail {
// direction (B, -A) of the total-least-squares line A x + B y + C = 0 through the points: the right
// singular vector of [x y 1] with the smallest singular value (EventCalibIni.cpp:46-57), oriented from
// the first to the last point
inline Vector2d row_direction(const CirclesEventFrame::CalibCircle *p, int width) {
    double M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int j = 0; j < width; j++) {
        const double r[3] = {p[j].location[0], p[j].location[1], 1.0};
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) M[a][b] += r[a] * r[b];
    }
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; sweep++) {  // cyclic Jacobi on the 3x3 Gram matrix
        double off = 0;
        for (int a = 0; a < 3; a++)
            for (int b = a + 1; b < 3; b++) off += M[a][b] * M[a][b];
        if (off < 1e-300) break;
        for (int a = 0; a < 3; a++)
            for (int b = a + 1; b < 3; b++) {
                if (M[a][b] == 0.0) continue;
                const double th = 0.5 * std::atan2(2 * M[a][b], M[b][b] - M[a][a]);
                const double c = std::cos(th), s = std::sin(th);
                for (int k = 0; k < 3; k++) {
                    const double mka = M[k][a], mkb = M[k][b];
                    M[k][a] = c * mka - s * mkb;
                    M[k][b] = s * mka + c * mkb;
                }
                for (int k = 0; k < 3; k++) {
                    const double mak = M[a][k], mbk = M[b][k];
                    M[a][k] = c * mak - s * mbk;
                    M[b][k] = s * mak + c * mbk;
                }
                for (int k = 0; k < 3; k++) {
                    const double vka = V[k][a], vkb = V[k][b];
                    V[k][a] = c * vka - s * vkb;
                    V[k][b] = s * vka + c * vkb;
                }
            }
    }
    int m = 0;
    for (int a = 1; a < 3; a++)
        if (M[a][a] < M[m][m]) m = a;
    Vector2d dir{{V[1][m], -V[0][m]}};
    const double dx = p[width - 1].location[0] - p[0].location[0], dy = p[width - 1].location[1] - p[0].location[1];
    if (dir[0] * dx + dir[1] * dy < 0) {
        dir[0] = -dir[0];
        dir[1] = -dir[1];
    }
    return dir;
}

// EventCalibIni::track's test: median angle between corresponding pattern rows of the two frames,
// divided by their time distance, below (5e-4 pi) / MotionTimeStep rad/s
inline bool orientation_gate(const KeyFrame &ref, const KeyFrame &cur, int rows, int cols, double motionTimeStep) {
    const double duration = std::fabs(cur.timeStamp - ref.timeStamp);
    std::vector<double> theta;
    for (int i = 0; i < rows; i++) {
        const Vector2d a = row_direction(&ref.features[i * cols], cols), b = row_direction(&cur.features[i * cols], cols);
        const double c = (a[0] * b[0] + a[1] * b[1]) / (std::sqrt(a[0] * a[0] + a[1] * a[1]) * std::sqrt(b[0] * b[0] + b[1] * b[1]));   // Eigen's norm(), not hypot
        theta.push_back(std::acos(c));   // (not clamped, as the reference: a cosine rounded above 1 gives NaN)
    }
    // (with a NaN among the angles the result is whatever the library's loops leave at the position — on libstdc++ the very
    // function the reference calls; the device policy restates those loops, ref_nth_element.hpp)
    std::nth_element(theta.begin(), theta.begin() + theta.size() / 2, theta.end());
    return theta[theta.size() / 2] / duration < (5e-4 * M_PI) / motionTimeStep;
}
}  // namespace detail

// The driver's keyframe search over [startTime, endTime]: pieceNum pieces (the reference uses
// 5 * (hardware_concurrency - 2)), len = 3 steps, frameGap = 5 steps (eventCameraCalib.cpp:168-179).
inline std::vector<KeyFrame> detect_keyframes(EventContainer &container, CirclePatternParameters::Ptr pattern,
                                              const CirclesEventFrame::Params &params, double motionTimeStep,
                                              int frameEventNumThreshold, int pieceNum, double startTime, double endTime) {
    const double len = 3 * motionTimeStep, frameGap = 5 * motionTimeStep;
    const double step = (endTime - startTime) / pieceNum;
    struct Piece {
        std::pair<double, double> bound, duration;
        bool active;
        std::vector<KeyFrame> keys;
    };
    std::vector<Piece> pieces(pieceNum);
    for (int k = 0; k < pieceNum; k++) {
        pieces[k].bound = {endTime - step * (k + 1), endTime - step * k};
        pieces[k].duration = {pieces[k].bound.first, pieces[k].bound.first + len};
        pieces[k].active = pieces[k].duration.second < pieces[k].bound.second;
    }
    ecal_detect_params prm;
    prm.dbscan_eps = params.dbscan_eps;
    prm.dbscan_min_samples = (uint32_t) params.dbscan_startMinSample;
    prm.cluster_min_sample = (uint32_t) params.clusterMinSample;
    prm.need_clusters = (uint32_t) (pattern->rows * pattern->cols);
    prm.circle_radius_threshold = ecal_circle_radius_threshold(container.cameraSize[0], container.cameraSize[1], pattern->rows,
                                                               pattern->cols, pattern->isAsymmetric, pattern->squareSize,
                                                               pattern->circleRadius);
    prm.fit_circle = params.fitCircle ? 1 : 0;
    prm.knn_num = (uint32_t) params.knn_num;
    prm.rows = (uint32_t) pattern->rows;
    prm.cols = (uint32_t) pattern->cols;
    std::vector<std::pair<double, double>> batch;
    std::vector<int> owner;
    std::vector<FrameDetection> det;
    for (;;) {
        batch.clear();
        owner.clear();
        for (int k = 0; k < pieceNum; k++)
            if (pieces[k].active) {
                batch.push_back(pieces[k].duration);
                owner.push_back(k);
            }
        if (batch.empty()) break;
        detect_windows(container, batch, prm, det);
        for (size_t b = 0; b < batch.size(); b++) {
            Piece &pc = pieces[owner[b]];
            const FrameDetection &d = det[b];
            const int events_num = (int) (d.positive.size() + d.negative.size());  // EventFrame::eventsNum()
            bool accepted = false;
            if (d.status == 0 && d.gridFound) {  // extractFeatures() == true
                KeyFrame kf;
                kf.timeStamp = (pc.duration.first + pc.duration.second) / 2;
                kf.duration = pc.duration;
                kf.eventsNum = events_num;
                for (size_t idx : d.orderIdxs)
                    kf.features.push_back(CirclesEventFrame::CalibCircle{d.candidateCenters[idx], d.candidatesRadius[idx]});
                if (pc.keys.empty() || detail::orientation_gate(pc.keys.back(), kf, pattern->rows, pattern->cols, motionTimeStep)) {
                    pc.keys.push_back(std::move(kf));
                    accepted = true;
                }
            }
            if (accepted) {  // eventCameraCalib.cpp:61-62
                pc.duration.first = pc.duration.second + frameGap;
                pc.duration.second = pc.duration.first + len;
            } else if (events_num > frameEventNumThreshold || (pc.duration.second - pc.duration.first) > 3 * len) {  // :67-69, :75-77
                pc.duration.first += motionTimeStep;
                pc.duration.second = pc.duration.first + len;
            } else {  // :70-71, :78-79
                pc.duration.second += motionTimeStep;
            }
            pc.active = pc.duration.second < pc.bound.second;  // :50
        }
    }
    std::vector<KeyFrame> all;
    for (auto &pc : pieces)
        for (auto &k : pc.keys) all.push_back(std::move(k));
    std::sort(all.begin(), all.end(), [](const KeyFrame &a, const KeyFrame &b) { return a.timeStamp < b.timeStamp; });
    return all;
}

// The same search with the policy ON THE DEVICE (ecal_detect_keyframes): the passes are enqueued back to back, nothing
// but a 4-byte counter crosses PCIe until the keyframes come back.  Same keyframes as detect_keyframes above.
// gateMode: ECAL_GATE_OWN_PIECE (the deterministic own-piece gate of detect_keyframes above) or ECAL_GATE_SHARED_MAP (the
// reference run with one worker thread: one keyframe map for all pieces, TrackingBase.cpp:16-46 + EventCalibIni.cpp:26-36).
inline std::vector<KeyFrame> detect_keyframes_device(EventContainer &container, CirclePatternParameters::Ptr pattern,
                                                     const CirclesEventFrame::Params &params, double motionTimeStep,
                                                     int frameEventNumThreshold, int pieceNum, double startTime, double endTime,
                                                     int gateMode = ECAL_GATE_SHARED_MAP, int pieceFirst = 0, int pieceCount = 0) {
    // (pieceCount != 0: only the pieces pieceFirst .. pieceFirst + pieceCount - 1, with the bounds they have in the whole run —
    // the cut of one search over processes / GPUs; own-piece gate only)
    ecal_detect_params prm;
    prm.dbscan_eps = params.dbscan_eps;
    prm.dbscan_min_samples = (uint32_t) params.dbscan_startMinSample;
    prm.cluster_min_sample = (uint32_t) params.clusterMinSample;
    prm.need_clusters = (uint32_t) (pattern->rows * pattern->cols);
    prm.circle_radius_threshold = ecal_circle_radius_threshold(container.cameraSize[0], container.cameraSize[1], pattern->rows,
                                                               pattern->cols, pattern->isAsymmetric, pattern->squareSize,
                                                               pattern->circleRadius);
    prm.fit_circle = params.fitCircle ? 1 : 0;
    prm.knn_num = (uint32_t) params.knn_num;
    prm.rows = (uint32_t) pattern->rows;
    prm.cols = (uint32_t) pattern->cols;
    ecal_adaptive_params ap;
    ap.motion_time_step = motionTimeStep;
    ap.frame_event_num_threshold = (uint32_t) frameEventNumThreshold;
    ap.piece_num = (uint32_t) pieceNum;
    ap.start_time = startTime;
    ap.end_time = endTime;
    ap.max_passes = 0;
    ap.check_every = 0;
    ap.gate_mode = gateMode;
    ap.piece_first = (uint32_t) pieceFirst;
    ap.piece_count = (uint32_t) pieceCount;
    const ecal_stream *es = container.device();
    const uint64_t n = ecal_stream_size(es);
    const size_t M = (size_t) prm.rows * prm.cols;
    const double span = std::max(endTime - startTime, 1e-9);
    // a pass covers a chain of windows per piece: the library's own estimate, doubled on ECAL_ERR_RANGE
    const uint64_t cap_max = std::min<uint64_t>(0xFFFFFFC0ull, 6 * n + 4096);   // (the hint's own maximum: the windows of one slot index are disjoint)
    uint64_t cap = ecal_detect_keyframes_cap_hint_dev(ecal_host::thread_ctx(), ecal_stream_data(es), n, &ap);
    if (cap == 0) throw std::runtime_error("ecal_detect_keyframes_cap_hint_dev: invalid parameters");
    uint32_t max_keys = (uint32_t) (span / (8 * motionTimeStep)) + (uint32_t) pieceNum + 64;
    std::vector<double> t, d, f;
    std::vector<int32_t> e;
    uint32_t K = 0;
    for (;;) {
        t.resize(max_keys);
        d.resize(2 * (size_t) max_keys);
        e.resize(max_keys);
        f.resize(3 * M * max_keys);
        const int rc = ecal_detect_keyframes(ecal_host::thread_ctx(), ecal_stream_data(es), n, &ap, &prm, (uint32_t) cap, max_keys,
                                             t.data(), d.data(), e.data(), f.data(), &K, nullptr, nullptr);
        if (rc == ECAL_OK) break;
        // double only the quantity that was short: K comes back with the needed keyframe count when that was it
        const bool keys_short = rc == ECAL_ERR_RANGE && K > max_keys;
        if (rc != ECAL_ERR_RANGE || (keys_short ? max_keys > (1u << 30) : cap >= cap_max))
            throw std::runtime_error(std::string("ecal_detect_keyframes: ") + ecal_strerror(rc) + " — " +
                                     ecal_last_error(ecal_host::thread_ctx()));
        if (keys_short) max_keys = std::max(2 * max_keys, K + 64);
        else cap = std::min<uint64_t>(cap_max, 2 * cap);
    }
    std::vector<KeyFrame> all(K);
    for (uint32_t k = 0; k < K; k++) {
        all[k].timeStamp = t[k];
        all[k].duration = {d[2 * k], d[2 * k + 1]};
        all[k].eventsNum = e[k];
        for (size_t c = 0; c < M; c++) {
            const double *p = f.data() + 3 * (k * M + c);
            all[k].features.push_back(CirclesEventFrame::CalibCircle{Vector2d{{p[0], p[1]}}, p[2]});
        }
    }
    return all;
}

}  // namespace opengv2

#endif  // ECAL_HOST_MULTI_PROCESS_HPP_
