// Host-side counterparts of the reference's event module (modules/camera_calibration/event) on top
// of libecal.so: same class names and member meaning, Eigen/OpenCV-free.
//   opengv2::Event          event/include/opengv2/event/Event.hpp
//   opengv2::EventStream    event/include/opengv2/event/EventStream.hpp, event/src/EventStream.cpp
//   opengv2::EventContainer event/include/opengv2/event/EventContainer.hpp  (device-resident here)
//   opengv2::EventFrame     event/include/opengv2/event/EventFrame.hpp, event/src/EventFrame.cpp
#ifndef ECAL_HOST_EVENT_HPP_
#define ECAL_HOST_EVENT_HPP_

#include <array>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/ecal.h"
#include "dbscan.h"

namespace opengv2 {

using Vector2d = std::array<double, 2>;  // stands in for Eigen::Vector2d: operator[] and contiguous

class Event {
public:
    Event() : timeStamp_(0), location_{{0, 0}}, polarity_(false) {}
    Event(double timeStamp, const Vector2d &location, bool polarity)
        : timeStamp_(timeStamp), location_(location), polarity_(polarity) {}
    double timeStamp() const noexcept { return timeStamp_; }
    const Vector2d &location() const noexcept { return location_; }  // (x, y)
    bool polarity() const noexcept { return polarity_; }             // true = positive

    static constexpr size_t kRecordBytes = 25;  // f64 t, f64 x, f64 y, u8 polarity — packed
    static Event unpack(const uint8_t *r) {
        Event e;
        std::memcpy(&e.timeStamp_, r, 8);
        std::memcpy(&e.location_[0], r + 8, 8);
        std::memcpy(&e.location_[1], r + 16, 8);
        e.polarity_ = r[24] != 0;
        return e;
    }

private:
    double timeStamp_;
    Vector2d location_;
    bool polarity_;
};

// Single-pass reader of the reference's .bin format.  Throws std::invalid_argument when the file
// does not exist, like the reference constructor (EventStream.cpp:10-17).
class EventStream {
public:
    explicit EventStream(const std::string &binFilePath) : is_(binFilePath, std::ifstream::binary | std::ifstream::in) {
        if (!is_.is_open()) throw std::invalid_argument("No such file: " + binFilePath);
        advance();
    }
    void close() {
        if (is_.is_open()) is_.close();
        end_ = true;
    }
    bool isEnd() const noexcept { return end_; }
    const Event &current() const noexcept { return cur_; }
    const uint8_t *currentRecord() const noexcept { return rec_; }
    void next() { advance(); }

    // text -> .bin converter with the reference's rules (EventStream.cpp:25-67): lines
    // "stamp x y polarity"; t = (stamp - base) * magnitude; negative t dropped; stops after endTime.
    static long long txt2bin(const std::string &txtFilePath, double timeMagnitude = 1e-6,
                             long long timeBase_in = std::numeric_limits<long long>::min(),
                             long long endTime_in = std::numeric_limits<long long>::min()) {
        std::ifstream is(txtFilePath);
        if (!is.is_open()) throw std::invalid_argument("No such file: " + txtFilePath);
        const auto dot = txtFilePath.find_last_of('.');
        std::ofstream os(txtFilePath.substr(0, dot) + ".bin", std::ofstream::binary | std::ofstream::trunc);
        long long counter = 0, stamp = 0, base = timeBase_in;
        double x = 0, y = 0;
        bool pol = false;
        while (is.good()) {
            is >> stamp >> x >> y >> pol;
            if (counter == 0 && timeBase_in == std::numeric_limits<long long>::min()) base = stamp;
            if (endTime_in != std::numeric_limits<long long>::min() && stamp > endTime_in) break;
            const double t = (stamp - base) * timeMagnitude;
            if (t < 0) continue;
            os.write((const char *) &t, 8);
            os.write((const char *) &x, 8);
            os.write((const char *) &y, 8);
            os.write((const char *) &pol, 1);
            counter++;
        }
        return counter;
    }

private:
    void advance() {
        is_.read((char *) rec_, Event::kRecordBytes);
        if (is_.gcount() != (std::streamsize) Event::kRecordBytes) {
            end_ = true;
            return;
        }
        cur_ = Event::unpack(rec_);
    }
    std::ifstream is_;
    Event cur_;
    uint8_t rec_[Event::kRecordBytes];
    bool end_ = false;
};

// The reference keeps a std::multimap<double, Event_loc_pol> on the host; here the time-ordered
// records live in HBM (one upload), and frames are cut out of them on the GPU.
struct EventContainer {
    typedef std::shared_ptr<EventContainer> Ptr;

    void emplace(const Event &e) {  // append in time order (the file order of a .bin)
        uint8_t r[Event::kRecordBytes];
        const double t = e.timeStamp();
        std::memcpy(r, &t, 8);
        std::memcpy(r + 8, &e.location()[0], 8);
        std::memcpy(r + 16, &e.location()[1], 8);
        r[24] = e.polarity() ? 1 : 0;
        if (fromFile_) throw std::logic_error("EventContainer: emplace after loadFile (the records live in HBM only)");
        records.insert(records.end(), r, r + Event::kRecordBytes);
        if (stream_) release();
    }
    // The whole reading loop of eventCameraCalib.cpp:154-163 in one call (ecal_stream_create_from_file): the records with
    // timeStamp >= startTime — up to the first one with timeStamp >= endTime when customEnd — go from the file straight
    // to HBM (chunked reads overlapped with the upload); `records` stays empty.
    void loadFile(const std::string &binFilePath, double startTime, bool customEnd = false, double endTime = 0.0) {
        release();
        records.clear();
        const int rc = ecal_stream_create_from_file(ecal_host::thread_ctx(), binFilePath.c_str(), startTime, customEnd ? 1 : 0, endTime,
                                                    &stream_);
        if (rc == ECAL_ERR_INVALID) throw std::invalid_argument("No such file: " + binFilePath);
        if (rc != ECAL_OK)
            throw std::runtime_error(std::string("ecal_stream_create_from_file: ") + ecal_strerror(rc) + " — " +
                                     ecal_last_error(ecal_host::thread_ctx()));
        if (ecal_stream_times(stream_, &devFirst_, &devLast_) != ECAL_OK) throw std::runtime_error("ecal_stream_times");
        fromFile_ = true;
    }
    size_t size() const { return fromFile_ ? (size_t) ecal_stream_size(stream_) : records.size() / Event::kRecordBytes; }
    double firstTime() const { return fromFile_ ? devFirst_ : Event::unpack(records.data()).timeStamp(); }
    double lastTime() const {
        return fromFile_ ? devLast_ : Event::unpack(records.data() + records.size() - Event::kRecordBytes).timeStamp();
    }

    // device image, uploaded lazily; throws if the records are not in time order
    const ecal_stream *device() {
        if (!stream_) {
            const int rc = ecal_stream_create(ecal_host::thread_ctx(), records.data(), size(), &stream_);
            if (rc != ECAL_OK)
                throw std::runtime_error(std::string("ecal_stream_create: ") + ecal_strerror(rc) + " — " +
                                         ecal_last_error(ecal_host::thread_ctx()));
        }
        return stream_;
    }
    void release() {
        if (stream_) ecal_stream_destroy(stream_);
        stream_ = nullptr;
        fromFile_ = false;
    }
    ~EventContainer() { release(); }

    std::vector<uint8_t> records;  // packed 25-byte records
    Vector2d cameraSize{{346, 260}};  // (width, height) of the sensor (CameraBase::size())

private:
    ecal_stream *stream_ = nullptr;
    bool fromFile_ = false;
    double devFirst_ = 0, devLast_ = 0;
};

// Result of the GPU pass over one window (shared by EventFrame and CirclesEventFrame)
struct FrameDetection {
    std::vector<Vector2d> positive, negative;  // positiveEvents_ / negativeEvents_, canonical order
    std::vector<int32_t> labelsPos, labelsNeg;  // DBSCAN labels (index into Clusters, -1 = Noise)
    std::vector<int32_t> keptPos, keptNeg;      // after the clusterMinSample filter
    uint32_t nClustersPos = 0, nClustersNeg = 0, keptClustersPos = 0, keptClustersNeg = 0, status = 1;
    bool tieFallback = false;   // ECAL_WIN_TIE_FALLBACK: a tied median of this window is not guaranteed to be the reference's pick
    std::vector<std::pair<size_t, size_t>> candidates;  // (+ cluster, - cluster), kept numbering
    std::vector<Vector2d> candidateCenters;
    std::vector<double> candidatesRadius;
    bool gridFound = false;            // cv::findCirclesGrid's isFound (CirclesEventFrame.cpp:332-336)
    std::vector<size_t> orderIdxs;     // candidate index per grid position i*cols + j (:343-348)
};

inline void detect_windows(EventContainer &c, const std::vector<std::pair<double, double>> &durations,
                           const ecal_detect_params &prm, std::vector<FrameDetection> &out) {
    const uint32_t S = (uint32_t) durations.size();
    out.assign(S, FrameDetection());
    if (S == 0) return;
    std::vector<double> t0(S), t1(S);
    for (uint32_t s = 0; s < S; s++) {
        t0[s] = durations[s].first;
        t1[s] = durations[s].second;
    }
    ecal_ctx *ctx = ecal_host::thread_ctx();
    // capacity: windows may overlap, so count the covered events first (bounds only, cheap)
    std::vector<uint32_t> base(S + 1);
    ecal_detect_result probe;
    std::memset(&probe, 0, sizeof(probe));
    probe.win_base = base.data();
    size_t cap = c.size();
    std::vector<double> xy;
    const uint32_t M = prm.rows * prm.cols;
    std::vector<int32_t> gorder((size_t) S * (M ? M : 1));
    std::vector<uint32_t> gfound(S, 0);
    std::vector<uint32_t> seg_off(2 * S), seg_cnt(2 * S), ncl(2 * S), info(4 * S), pair;
    std::vector<int32_t> labels, kept;
    std::vector<double> xyr;
    for (int attempt = 0; attempt < 2; attempt++) {
        xy.resize(2 * cap);
        labels.resize(cap);
        kept.resize(cap);
        pair.resize(2 * cap);
        xyr.resize(3 * cap);
        ecal_detect_result r;
        std::memset(&r, 0, sizeof(r));
        r.win_base = base.data();
        r.xy = xy.data();
        r.seg_off = seg_off.data();
        r.seg_cnt = seg_cnt.data();
        r.labels = labels.data();
        r.n_clusters = ncl.data();
        r.kept_labels = kept.data();
        r.win_info = info.data();
        r.cand_pair = pair.data();
        r.cand_xyr = xyr.data();
        r.grid_order = gorder.data();
        r.grid_found = gfound.data();
        const int rc = ecal_detect_batch(ctx, c.device(), t0.data(), t1.data(), S, &prm, (uint32_t) cap, &r);
        if (rc == ECAL_OK) break;
        if (rc == ECAL_ERR_RANGE && attempt == 0) {  // overlapping windows cover more slots than events
            cap = (size_t) base[S] + 16;
            continue;
        }
        throw std::runtime_error(std::string("ecal_detect_batch: ") + ecal_strerror(rc) + " — " + ecal_last_error(ctx));
    }
    for (uint32_t s = 0; s < S; s++) {
        FrameDetection &f = out[s];
        const uint32_t op = seg_off[2 * s], np = seg_cnt[2 * s], on = seg_off[2 * s + 1], nn = seg_cnt[2 * s + 1];
        f.positive.resize(np);
        f.negative.resize(nn);
        for (uint32_t i = 0; i < np; i++) f.positive[i] = Vector2d{{xy[2 * (op + i)], xy[2 * (op + i) + 1]}};
        for (uint32_t i = 0; i < nn; i++) f.negative[i] = Vector2d{{xy[2 * (on + i)], xy[2 * (on + i) + 1]}};
        f.labelsPos.assign(labels.begin() + op, labels.begin() + op + np);
        f.labelsNeg.assign(labels.begin() + on, labels.begin() + on + nn);
        f.keptPos.assign(kept.begin() + op, kept.begin() + op + np);
        f.keptNeg.assign(kept.begin() + on, kept.begin() + on + nn);
        f.nClustersPos = ncl[2 * s];
        f.nClustersNeg = ncl[2 * s + 1];
        f.keptClustersPos = info[4 * s + 1];
        f.keptClustersNeg = info[4 * s + 2];
        f.status = ECAL_WIN_STATUS(info[4 * s + 3]);
        f.tieFallback = (info[4 * s + 3] & ECAL_WIN_TIE_FALLBACK) != 0;
        for (uint32_t j = 0; j < info[4 * s]; j++) {
            f.candidates.emplace_back(pair[2 * (op + j)], pair[2 * (op + j) + 1]);
            f.candidateCenters.push_back(Vector2d{{xyr[3 * (op + j)], xyr[3 * (op + j) + 1]}});
            f.candidatesRadius.push_back(xyr[3 * (op + j) + 2]);
        }
        f.gridFound = M > 0 && gfound[s] != 0;
        if (f.gridFound)
            for (uint32_t m = 0; m < M; m++) f.orderIdxs.push_back((size_t) gorder[(size_t) s * M + m]);
    }
}

// EventFrame: events with duration.first <= t <= duration.second, per-polarity unique pixel sets,
// pixels that fired with both polarities removed (EventFrame.cpp:10-36).
class EventFrame {
public:
    EventFrame(EventContainer::Ptr container, const std::pair<double, double> &duration)
        : container_(std::move(container)), duration_(duration) {}
    virtual ~EventFrame() {}
    int eventsNum() {
        ensure();
        return (int) (det_.positive.size() + det_.negative.size());
    }
    void releaseEventSet() {
        det_.positive.clear();
        det_.negative.clear();
    }
    const std::vector<Vector2d> &positiveEvents() {
        ensure();
        return det_.positive;
    }
    const std::vector<Vector2d> &negativeEvents() {
        ensure();
        return det_.negative;
    }

protected:
    virtual ecal_detect_params params() const {
        ecal_detect_params p;
        p.dbscan_eps = 4;
        p.dbscan_min_samples = 2;
        p.cluster_min_sample = 5;
        p.need_clusters = 36;
        p.circle_radius_threshold = ecal_circle_radius_threshold(346, 260, 9, 4, 1, 5.5, 1.75);
        p.fit_circle = 0;
        p.knn_num = 3;
        p.rows = 0;
        p.cols = 0;
        return p;
    }
    void ensure() {
        if (done_) return;
        std::vector<FrameDetection> out;
        detect_windows(*container_, {duration_}, params(), out);
        det_ = std::move(out[0]);
        done_ = true;
    }
    EventContainer::Ptr container_;
    std::pair<double, double> duration_;
    FrameDetection det_;
    bool done_ = false;
};

}  // namespace opengv2

#endif  // ECAL_HOST_EVENT_HPP_
